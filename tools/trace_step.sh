#!/bin/bash
# Per-dispatch kernel trace of the single-stream schedule (one warm-up step + one step), for per-step analysis with
# tools/analyze_trace.py.  usage (GPU box, repo root): tools/trace_step.sh TAG [extra bench.py args]
set -e
TAG=${1:-r2}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_${TAG} -- python3 $ROOT/bench.py --steps 1 --warmup 1 --overlap 0 --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --no-single-stream-leg "$@" > $OUT/trace_${TAG}.log 2>&1
cd $ROOT
find $OUT/trace_${TAG} -name "*kernel_trace.csv" | head -1
