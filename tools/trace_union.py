#!/usr/bin/env python3
"""How busy was the GPU?  Reads a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv) and prints, for the last FRAC of the trace (the
steady state): the span, the UNION of all kernel intervals (time with at least one kernel resident), the time with >= 2 kernels
resident, the idle gaps (count, total, the largest) and the sum of kernel durations.  span - union = time the device waited for the host
(or for an event): a host-bound step shows it, a device-bound one does not.
With a third argument (a kernel-name substring that occurs ONCE per step, e.g. radam_kernel) the same figures are printed per step --
the interval between two consecutive ends of that kernel -- for the last few steps, so warm-up and the profiler's start-up stay out.
usage: python tools/trace_union.py <kernel_trace.csv> [frac=0.5] [delimiter]"""
import csv
import sys

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows, queue = [], {}
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        queue[(rows[-1][0], rows[-1][1])] = r.get("Queue_Id", "")
rows.sort()
if len(sys.argv) > 3:
    ends = [e for _, e, n in rows if sys.argv[3] in n]
    if len(sys.argv) > 4:                    # a fourth argument: dump the last step's kernels (start, duration, queue, name) to that file
        a, b = ends[-2], ends[-1]
        import re
        with open(sys.argv[4], "w") as f:
            for s_, e_, n in rows:
                if e_ > a and s_ < b:
                    short = re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]
                    f.write(f"{(s_ - a) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f} q{queue[(s_, e_)]:>3s} {short}\n")
    for a, b in list(zip(ends, ends[1:]))[-8:]:
        ks = [(max(s, a), min(e, b)) for s, e, _ in rows if e > a and s < b]
        ev = sorted([(s, 1) for s, _ in ks] + [(e, -1) for _, e in ks])
        depth, last, busy, busy2 = 0, a, 0, 0
        gaps = []
        for t, d in ev:
            if depth >= 1:
                busy += t - last
            elif t - last > 0:
                gaps.append(t - last)
            if depth >= 2:
                busy2 += t - last
            depth += d
            last = t
        gaps.sort(reverse=True)
        print(f"step {(b - a) / 1e6:7.3f} ms  kernels {len(ks):4d}  union {busy / 1e6:6.3f} ms ({busy / (b - a):5.1%})  >=2 resident {busy2 / 1e6:6.3f} ms"
              f"  sum {sum(e - s for s, e in ks) / 1e6:6.3f} ms  idle {sum(gaps) / 1e6:6.3f} ms in {len(gaps)} gaps, largest {[round(g / 1e3, 1) for g in gaps[:5]]} us")
    sys.exit(0)
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
lo = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= lo]
span = max(e for _, e, _ in rows) - rows[0][0]
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, busy1, busy2, gaps = 0, ev[0][0], 0, 0, []
for t, d in ev:
    if depth >= 1:
        busy1 += t - last
    elif t > last:
        gaps.append(t - last)
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
tot = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  union {busy1 / 1e6:.3f} ms ({busy1 / span:.1%})  >=2 resident {busy2 / 1e6:.3f} ms ({busy2 / span:.1%})"
      f"  sum of durations {tot / 1e6:.3f} ms")
gaps.sort(reverse=True)
print(f"idle gaps {len(gaps)}  total {sum(gaps) / 1e6:.3f} ms  largest (us): {[round(g / 1e3, 1) for g in gaps[:12]]}")
big = sum(g for g in gaps if g > 5000)
print(f"gaps > 5 us: {sum(1 for g in gaps if g > 5000)} totalling {big / 1e6:.3f} ms; gaps <= 5 us: {sum(1 for g in gaps if g <= 5000)} totalling {(sum(gaps) - big) / 1e6:.3f} ms")
