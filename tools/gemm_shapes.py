#!/usr/bin/env python
"""Aggregates the per-call lines NMTGAN_TRACE_GEMM=1 prints (shape, microseconds, TFLOP/s) by (epilogue, N, K, ldw) and
by ranges of M.  usage: NMTGAN_TRACE_GEMM=1 python bench.py --steps 1 --warmup 0 --overlap 0 ... 2> log; tools/gemm_shapes.py log"""
import re
import sys
from collections import defaultdict

pat = re.compile(r"GEMM epi=(\d+) M=(\d+) N=(\d+) K=(\d+) a_group=(-?\d+) ldw=(-?\d+) us=([\d.]+) tflops=([\d.]+)")
agg = defaultdict(lambda: [0, 0.0, 0.0])
for line in open(sys.argv[1]):
    m = pat.search(line)
    if not m:
        continue
    epi, M, N, K, ag, ldw = (int(m.group(i)) for i in range(1, 7))
    us = float(m.group(7))
    mb = "M<4k" if M < 4096 else ("M<16k" if M < 16384 else ("M<64k" if M < 65536 else "M>=64k"))
    key = (epi, N, K, "conv" if ag > 0 else "dense", mb)
    a = agg[key]
    a[0] += 1
    a[1] += us
    a[2] += 2.0 * M * N * K
tot = sum(a[1] for a in agg.values())
print("total %.1f ms over %d calls" % (tot / 1e3, sum(a[0] for a in agg.values())))
for key, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.2f ms %5d calls %7.1f us/call %7.1f TF/s  epi=%d N=%-5d K=%-6d %s %s" % (us / 1e3, n, us / n, fl / us / 1e6, *key))
