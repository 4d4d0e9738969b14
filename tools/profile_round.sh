#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   kernel-trace stats of `bench.py --overlap 0` (kernels in isolation) and of the default two-stream
#   schedule, and separate FETCH_SIZE / WRITE_SIZE / matrix-core busy passes over one single-stream get_reward.
# usage: tools/profile_round.sh TAG [PRECISION]   ->  gpurun_out/prof_TAG_{serial,overlap,fetch,write,mfma}
set -e
TAG=${1:-r1}
PREC=${2:-fp16}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --precision $PREC"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_serial -- python3 $B --steps 2 --warmup 1 --overlap 0 > $OUT/prof_${TAG}_serial.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_overlap -- python3 $B --steps 2 --warmup 1 --no-single-stream-leg > $OUT/prof_${TAG}_overlap.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $B --steps 1 --warmup 0 --overlap 0 > $OUT/prof_${TAG}_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $B --steps 1 --warmup 0 --overlap 0 > $OUT/prof_${TAG}_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_${TAG}_mfma -- python3 $B --steps 1 --warmup 0 --overlap 0 > $OUT/prof_${TAG}_mfma.log 2>&1
cd $ROOT
# keep only the small summary files (the traces are tens of MB)
find $OUT/prof_${TAG}_serial $OUT/prof_${TAG}_overlap -type f ! -name "*kernel_stats.csv" -delete
LABEL="C2 (d512 N6 h8, B64 R16 T50 V32000), precision tier $PREC"
python3 tools/summarize_profile.py ${TAG} $OUT/prof_${TAG}_serial $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write --mfma=$OUT/prof_${TAG}_mfma "--label=3 get_reward steps (1 warm-up + 2 timed) on the single-stream schedule; $LABEL"
python3 tools/summarize_profile.py ${TAG}_two_stream $OUT/prof_${TAG}_overlap "--label=3 get_reward steps (1 warm-up + 2 timed) on the two-stream schedule; $LABEL"
mkdir -p $OUT/profiles_${TAG} && cp profiles/${TAG}_* $OUT/profiles_${TAG}/
echo profile_round done
