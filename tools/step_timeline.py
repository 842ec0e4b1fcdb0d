"""Busy / idle account of the optimisation steps in a rocprofv3 kernel trace (tools/kstats.sh <dir> tools/prof_step.py ...):
python tools/step_timeline.py gpurun_out/<dir>/p_kernel_trace.csv <first kernel of a step (substring)> [skip steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
# a step = from one occurrence of the key kernel to the next one (first occurrence per step: de-duplicate close repeats)
steps = []
for a, b in zip(starts, starts[1:]):
    steps.append(rows[a:b])
steps = steps[skip:]
if not steps:
    sys.exit("no steps found")
tot = busy = 0
per = {}
cnt = {}
for st in steps:
    t0, t1 = int(st[0]["Start_Timestamp"]), int(st[-1]["End_Timestamp"])
    end = t0
    b = 0
    for r in st:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > end:
            b += e - max(s, end)
            end = e
        per[r["Kernel_Name"][:80]] = per.get(r["Kernel_Name"][:80], 0) + e - s
        cnt[r["Kernel_Name"][:80]] = cnt.get(r["Kernel_Name"][:80], 0) + 1
    tot += t1 - t0
    busy += b
n = len(steps)
print("%d steps: %.1f us from first launch to last end, GPU busy %.1f us (%.1f %% idle); %d launches per step" %
      (n, tot / n / 1e3, busy / n / 1e3, 100 * (1 - busy / tot), sum(len(s) for s in steps) / n))
small = 0.
for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
    each = v / cnt[k] / 1e3
    if each < 10.:
        small += cnt[k] / n
    print("  %-80s %9.1f us/step  %5.2f launches/step  %8.1f us each" % (k, v / n / 1e3, cnt[k] / n, each))
print("launches under 10 us: %.1f per step" % small)
