"""Reconstructs one steady-state step from a rocprofv3 kernel trace of bench.py:

    python tools/step_timeline.py gpurun_out/<dir>/<name>_kernel_trace.csv [step_index_from_end]

A step is the window between two consecutive radam_kernel dispatches.  Prints, per HIP stream, busy time and the largest
gaps on the main stream (the stream radam runs on) with the kernels either side -- the idle the in-library tracer books as
"other" -- and the per-kernel time on the main stream.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"])))
    rows.sort()
    radam = [i for i, r in enumerate(rows) if "radam_kernel" in r[3]]
    if len(radam) < back + 1:
        raise SystemExit("not enough steps in the trace")
    lo, hi = radam[-back - 1], radam[-back]
    step = rows[lo + 1:hi + 1]
    t0, t1 = rows[lo][1], rows[hi][1]
    main_stream = rows[hi][2]
    print(f"step window {(t1 - t0) / 1e3:.1f} us, {len(step)} dispatches, main stream {main_stream}")
    by_stream = defaultdict(list)
    for r in step:
        by_stream[r[2]].append(r)
    for s, rs in sorted(by_stream.items()):
        busy = sum(e - b for b, e, _, _ in rs)
        print(f"  stream {s}: {len(rs):4d} dispatches, busy {busy / 1e3:8.1f} us")
    ms = by_stream[main_stream]
    gaps = []
    prev_end, prev_name = t0, "radam_kernel(prev)"
    for b, e, _, n in ms:
        gaps.append((b - prev_end, prev_name, n, (b - t0) / 1e3))
        prev_end, prev_name = max(prev_end, e), n
    tot_gap = sum(max(g[0], 0) for g in gaps)
    print(f"  main-stream idle {tot_gap / 1e3:.1f} us in {len(gaps)} gaps; mean {tot_gap / len(gaps) / 1e3:.2f} us")
    print("  largest gaps (us, at, after -> before):")
    for g in sorted(gaps, reverse=True)[:25]:
        print(f"    {g[0] / 1e3:7.1f}  @{g[3]:8.1f}  {g[1]}  ->  {g[2]}")
    agg = defaultdict(lambda: [0, 0])
    for b, e, _, n in ms:
        agg[n][0] += e - b
        agg[n][1] += 1
    print("  main-stream kernels:")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"    {t / 1e3:8.1f} us  x{c:3d}  {n}")
    # gap histogram
    hist = defaultdict(int)
    for g in gaps:
        k = 0 if g[0] < 1000 else 1 if g[0] < 2000 else 2 if g[0] < 5000 else 3 if g[0] < 10000 else 4
        hist[k] += 1
    print("  gap histogram (<1, 1-2, 2-5, 5-10, >10 us):", [hist[k] for k in range(5)])


if __name__ == "__main__":
    main()
