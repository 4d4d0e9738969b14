# A/B of the row cut of LINEAR problems (NMTGAN_GEMM_ROWSPLIT), both throughput tiers, same box
set -e
for prec in fp16 fp16x3; do
for v in 0 1 0 1; do
  NMTGAN_GEMM_ROWSPLIT=$v timeout -k 10 250 python bench.py --precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-tier-legs --no-call-leg > gpurun_out/rowsplit_${prec}_$v.json 2>gpurun_out/rowsplit_${prec}_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/rowsplit_${prec}_$v.json').read().strip().splitlines()[-1])
r=d['roofline_all']
print("$prec rowsplit $v: two-stream %.2f single %.2f | ffn %.2f proj %.2f ms" % (d['ms_per_step'], d.get('single_stream_ms_per_step') or 0, r['gemm_ffn']['ms'], r['gemm_proj']['ms']), flush=True)
PY
done
done
