#!/bin/bash
# Sweep of the two-stream schedule's knobs (workgroup cap of the discriminator GEMMs while decode steps run, rows per
# hand-over) on the benchmark workload.  usage (GPU box): tools/sweep_aux.sh "WGS..." "ROWS..."
for w in ${1:-160 192 224 256}; do for m in ${2:-4096}; do
  r=$(NMTGAN_AUX_WGS=$w NMTGAN_AUX_MIN_ROWS=$m timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --no-single-stream-leg 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],1))")
  echo "aux_wgs=$w min_rows=$m ms=$r"
done; done
