#!/usr/bin/env python3
"""One REPLAYED step's kernels in start order from a rocprofv3 kernel trace (rocpd SQLite): start offset, duration, hardware queue, and how
many kernels were running when each one started — the picture of what the forked branches of the captured graph overlap (round 6; the eager
tools/step_timeline.py serialises everything on one stream and cannot show it).

usage:  rocprofv3 --kernel-trace -d gpurun_out/pg -o g -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary
        python tools/graph_timeline.py $(find gpurun_out/pg -name "*.db" | head -1) [step index from the end, default 2]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def main(path, back=2):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info('rocpd_kernel_dispatch')")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = c.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) < back + 1:
        raise SystemExit(f"only {len(adam)} optimizer launches in the trace")
    lo, hi = adam[-back - 1] + 1, adam[-back] + 1
    # the derived-copy launches behind Adam belong to the step that ends there
    while hi < len(rows) and ("derive_copies" in rows[hi][0] or "gather_chunks" in rows[hi][0]):
        hi += 1
    while lo < len(rows) and ("derive_copies" in rows[lo][0] or "gather_chunks" in rows[lo][0]):
        lo += 1
    step = rows[lo:hi]
    t0 = step[0][1]
    qs = {q: i for i, q in enumerate(sorted({r[3] for r in step}))}
    busy = 0.0
    last_end = t0
    union = 0.0
    print(f"# {len(step)} kernels, {len(qs)} hardware queues; columns: start offset us | duration us | queue | kernels already running | kernel")
    for i, (name, st, en, q) in enumerate(step):
        running = sum(1 for (_, s2, e2, _) in step[:i] if e2 > st)
        print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{qs[q]}  {running:2d}  {short(name)}")
        busy += en - st
        if en > last_end:
            union += en - max(st, last_end)
            last_end = en
    span = (max(r[2] for r in step) - t0) / 1e3
    print(f"# step span {span:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, time with at least one kernel running {union / 1e3:.1f} us, "
          f"idle inside the span {span - union / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
