#!/bin/bash
# Per-kernel time of one precision tier on the single-stream schedule (run on the GPU box from the repo root):
# tools/kstats.sh TAG PRECISION [extra bench flags]  ->  top kernels on stdout, ms per get_reward step
TAG=$1; PREC=$2; shift 2
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --precision $PREC --steps 2 --warmup 1 --overlap 0 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv")
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 3e6
print("all kernels: %.1f ms/step" % tot)
for r in rows[:16]:
    print("%-78s %5s calls %8.2f ms/step %9.1f us" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 3e6, float(r["AverageNs"]) / 1e3))
PY
find gpurun_out/prof_$TAG -type f ! -name "*kernel_stats.csv" -delete
