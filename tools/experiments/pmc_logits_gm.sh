#!/bin/bash
# FETCH_SIZE of the logits GEMM for two tile-group sizes (one single-stream get_reward each)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --steps 1 --warmup 0 --overlap 0"
for gm in 2 8; do
  export NMTGAN_GEMM_GM=$gm
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_gm$gm -- python3 $B > $OUT/pmc_gm$gm.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for gm in (2, 8):
    f = glob.glob('gpurun_out/pmc_gm%d/*/*counter_collection.csv' % gm)[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE': continue
        k = r['Kernel_Name'].split('(')[0]
        tot[k] += float(r['Counter_Value']); n[k] += 1
    for k in tot:
        if 'Li100E' in k or ', 100>' in k or '100>' in k:
            print('GM=%d %s launches %d  FETCH x2 = %.1f GB' % (gm, k[:60], n[k], 2 * tot[k] * 1024 / 1e9))
    print('GM=%d total FETCH x2 = %.1f GB' % (gm, 2 * sum(tot.values()) * 1024 / 1e9))
PY
