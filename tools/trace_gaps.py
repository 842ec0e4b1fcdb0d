"""Idle time between consecutive kernels of a traced run (rocprofv3 --kernel-trace CSV): per step (delimited by a marker
kernel, default adam_k) the wall time, the busy time (union of kernel intervals) and the largest gaps with the kernels
on either side.    python tools/trace_gaps.py <kernel_trace.csv> [marker] [step-index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_k"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
ends = [i for i, e in enumerate(ev) if marker in e[2]]
a, b = ends[which - 1] + 1, ends[which] + 1
step = ev[a:b]
wall = step[-1][1] - ev[a - 1][1]
busy, gaps, cur = 0, [], ev[a - 1][1]
for s, e, n in step:
    if s > cur:
        gaps.append((s - cur, n))
    busy += max(0, e - max(s, cur))
    cur = max(cur, e)
print("step of %d kernels: wall %.1f us, busy %.1f us, idle %.1f us in %d gaps" % (len(step), wall / 1e3, busy / 1e3, (wall - busy) / 1e3, len(gaps)))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
prev = ev[a - 1]
for s, e, n in step:
    g = s - prev[1]
    if g > 1500:
        print("  %6.1f us before %-60s (after %s)" % (g / 1e3, short(n), short(prev[2])))
    if e > prev[1]:
        prev = (s, e, n)
