# two-stream schedule: rows that must have finished before the discriminator stream is handed a batch
set -e
PREC=${1:-fp16}
for v in 6144 12288 24576 32768 36864 40960 45056 6144; do
  timeout -k 10 250 python bench.py --precision $PREC --steps 6 --warmup 2 --aux-min-rows $v --no-cpu-baseline --no-roofline --no-fp32-leg --no-tier-legs --no-call-leg --no-single-stream-leg > gpurun_out/ho_${PREC}_$v.json 2>gpurun_out/ho_${PREC}_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/ho_${PREC}_$v.json').read().strip().splitlines()[-1])
print("$PREC aux_min_rows $v: two-stream %.2f ms" % d['ms_per_step'], flush=True)
PY
done
