#!/bin/bash
# Per-dispatch kernel trace of the two-stream schedule (one warm-up step + one step) for tools/analyze_overlap.py.
# usage (GPU box, repo root): tools/trace_overlap.sh TAG [extra bench.py args]
set -e
TAG=${1:-r3}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace2_${TAG} -- python3 $ROOT/bench.py --steps 1 --warmup 1 --overlap 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --no-single-stream-leg "$@" > $OUT/trace2_${TAG}.log 2>&1
cd $ROOT
find $OUT/trace2_${TAG} -name "*kernel_trace.csv" | head -1
