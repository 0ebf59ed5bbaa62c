#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite database.

usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/xxx.txt]
Equivalent to rocprofv3's kernel stats table; used because this rocprofv3 build writes rocpd databases by default.
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    rows = c.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name").fetchall() \
        if "name" in cols else []
    if not rows:
        rows = c.execute(
            "select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
            "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name").fetchall()
    total = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel")
    for name, n, t, mn, mx in rows:
        print(f"{n:7d} {t / 1e6:10.3f} {t / n / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * t / total:6.2f}  {short(name)}")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


def pmc(path):
    """per-kernel average of every collected counter (view counters_collection): `python tools/rocpd_stats.py --pmc x.db`"""
    c = sqlite3.connect(path)
    # one kernel instance serves several shapes (conv1 of the static and of the gripper camera): its launches are told apart by their
    # LDS allocation and grid
    rows = c.execute("select kernel_name, counter_name, count(*), sum(value), lds_block_size, grid_size from counters_collection "
                     "group by kernel_name, counter_name, lds_block_size, grid_size").fetchall()
    rows.sort(key=lambda r: -r[3])
    print(f"{'calls':>7} {'avg_value':>16} {'counter':>14} {'lds':>7} {'grid':>9}  kernel")
    for name, ctr, n, v, lds, grid in rows:
        print(f"{n:7d} {v / n:16.1f} {ctr:>14} {lds:7d} {grid:9d}  {short(name)}")


def mfma(path):
    """MFMA utilisation per kernel from one pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE:
         util %   = 100 x SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 256 CUs x 4 SIMDs)     (the gfx94x MfmaUtil formula rocprofv3 falls back to)
         TFLOP/s  = 512 x SQ_INSTS_VALU_MFMA_MOPS_BF16 / (cycles / 2.4 GHz)           (counter unit: 512 FLOP)
       with cycles = GRBM_GUI_ACTIVE / 8: rocprofv3 reports the counter summed over the 8 XCDs (checked against kernel durations: the
       conv3 forward launch shows 2.66 M "active" = 8 x 332 k cycles = 138 us, its traced duration under the counter pass).  A dispatch
       under --pmc carries ~10 us of serialisation, so small kernels read low; the big kernels' numbers match FLOPs / traced time.
       `python tools/rocpd_stats.py --mfma x.db`"""
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    per = {}
    for name, ctr, n, v in rows:
        d = per.setdefault(name, {"n": 0})
        d[ctr] = v
        d["n"] = max(d["n"], n)
    tot_busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in per.values())
    tot_act = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in per.values()) / 8.0
    tot_mops = sum(d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) for d in per.values())
    print(f"# all kernels: MFMA busy {tot_busy:.3e} cycles over {tot_act:.3e} active cycles (GRBM_GUI_ACTIVE / 8 XCDs) -> utilisation {100 * tot_busy / max(tot_act * 1024, 1):.2f} % of the "
          f"1024 SIMD matrix pipes; {512 * tot_mops / 1e12:.3f} TFLOP counted ({512 * tot_mops / max(tot_act / 2.4e9, 1e-12) / 1e12:.1f} TFLOP/s at 2.4 GHz)")
    print(f"{'calls':>7} {'cycles/launch':>18} {'mfma_busy/launch':>17} {'mfma_util_%':>11} {'bf16_TFLOP/s':>12} {'sq_busy/launch':>15}  kernel")
    for name, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        n, act, busy = d["n"], d.get("GRBM_GUI_ACTIVE", 0) / 8.0, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        mops, sqb = d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0), d.get("SQ_BUSY_CYCLES", 0)
        print(f"{n:7d} {act / n:18.0f} {busy / n:17.0f} {100 * busy / max(act * 1024, 1):11.2f} {512 * mops / max(act / 2.4e9, 1e-12) / 1e12:12.1f} {sqb / n:15.0f}  {short(name)}")


def listing(path, pattern):
    """every dispatch whose kernel name contains `pattern`, in launch order: `python tools/rocpd_stats.py --list PATTERN x.db`"""
    c = sqlite3.connect(path)
    rows = c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    for name, st, en, grid in rows:
        if pattern in name:
            print(f"{(en - st) / 1e3:10.2f} us  grid {grid:8d}  {short(name)[:80]}")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--list":
        listing(sys.argv[3], sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--mfma":
        mfma(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--pmc":
        pmc(sys.argv[2])
    else:
        main(sys.argv[1])
