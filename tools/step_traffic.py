"""Counted HBM traffic of one training step: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every kernel of the two --pmc passes
(gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md HBM section), divided by the steps the passes ran.

usage: python tools/step_traffic.py                      the newest committed pair profiles/r*_pmc_fetch_size.txt / _write_size.txt
       python tools/step_traffic.py <tag>                 that pair, e.g. r06_a
       python tools/step_traffic.py <fetch.db> <write.db> [steps]   the rocpd databases of the two passes themselves
The number of steps is the number of optimizer (adam_kernel) launches in the pass when it is not given."""
import glob
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def from_db(path, ctr):
    c = sqlite3.connect(path)
    tot = c.execute("select sum(value) from counters_collection where counter_name = ?", (ctr,)).fetchone()[0] or 0.0
    n = c.execute("select count(*) from counters_collection where counter_name = ? and kernel_name like '%adam_kernel%'", (ctr,)).fetchone()[0]
    return tot, n


def from_txt(path):
    """the committed per-kernel summaries: `calls  avg_value  counter  lds  grid  kernel` -> (sum of calls x avg in KB, optimizer launches, rows)"""
    tot, steps, rows = 0.0, 0, []
    for line in open(path).read().splitlines():
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+(\w+)\s+(\d+)\s+(\d+)\s+(.*)", line)
        if not m:
            continue
        n, avg, name = int(m.group(1)), float(m.group(2)), m.group(6).strip()
        tot += n * avg
        rows.append((n * avg, n, name))
        if "adam_kernel" in name:
            steps = max(steps, n)
    return tot, steps, rows


def main(argv):
    if len(argv) >= 2 and argv[0].endswith(".db"):
        (f, nf), (w, nw) = from_db(argv[0], "FETCH_SIZE"), from_db(argv[1], "WRITE_SIZE")
        steps = float(argv[2]) if len(argv) > 2 else float(max(nf, nw, 1))
        src = f"{argv[0]} + {argv[1]}"
        top = []
    else:
        if argv:
            tag = argv[0]
        else:
            pairs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_size.txt")), key=os.path.getmtime)
            if not pairs:
                raise SystemExit("no profiles/r*_pmc_fetch_size.txt; give <tag> or the two rocpd databases (see the module docstring)")
            tag = os.path.basename(pairs[-1])[:-len("_pmc_fetch_size.txt")]
        fp, wp = (os.path.join(ROOT, "profiles", f"{tag}_pmc_{k}.txt") for k in ("fetch_size", "write_size"))
        (f, nf, fr), (w, nw, wr) = from_txt(fp), from_txt(wp)
        steps = float(max(nf, nw, 1))
        src = f"profiles/{tag}_pmc_fetch_size.txt + _write_size.txt"
        wmap = {name: v for v, _, name in wr}
        top = sorted(((2 * v + wmap.get(name, 0.0)) * 1024 / steps, name) for v, _, name in fr)[::-1][:12]
    print(f"counted HBM traffic: {(2 * f + w) * 1024 / steps / 1e9:.2f} GB per step (fetch {2 * f * 1024 / steps / 1e9:.2f}, write {w * 1024 / steps / 1e9:.2f}; "
          f"{steps:.0f} steps, set-up kernels included; {src})")
    for b, name in top:
        print(f"  {b / 1e6:9.1f} MB per step  {name[:130]}")


if __name__ == "__main__":
    main(sys.argv[1:])
