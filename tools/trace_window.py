"""Kernels around the n-th launch matching a substring in a rocprofv3 --kernel-trace CSV (start order, with queue ids):
python tools/trace_window.py <kernel_trace.csv> <substring> [n] [before] [after]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
sub = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
before = int(sys.argv[4]) if len(sys.argv) > 4 else 12; after = int(sys.argv[5]) if len(sys.argv) > 5 else 12
idx = [i for i, e in enumerate(ev) if sub in e[2]]
i0 = idx[n]
t0 = ev[i0][0]
for s, e, name, q, st in ev[max(0, i0 - before):i0 + after]:
    print("%9.1f .. %9.1f us  q%s s%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, st, name))
