# two-stream knobs of bench.py: workgroup cap of the auxiliary stream's persistent GEMMs x rows per hand-over batch
for w in 128 176 224 256; do for r in 4096 6144 10240; do
  echo -n "wgs $w minrows $r: "
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --precision ${1:-fp16} --no-tier-legs --no-call-leg --no-cpu-baseline --no-roofline --no-single-stream-leg --aux-wgs $w --aux-min-rows $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))"
done; done
