#!/bin/bash
# rocprofv3 kernel trace of a python tool, per-kernel totals printed:  bash tools/kstats.sh <outdir> <script> [args...]
# (environment variables such as GNF_MONO_SHAPE are inherited; the program after `--` is python3 itself)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$@" > "$out.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f: sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in list(csv.DictReader(open(f[0])))[:int(16)]:
    print("%-90s calls %5s  total %12s ns  avg %10.1f us" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]) / 1e3))
PY
