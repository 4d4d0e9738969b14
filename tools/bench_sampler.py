"""Micro-benchmark of nmtgan_sample_rows (HIP events): rows x V fp32 logits far larger than the MALL."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nmt_gan_amd import _lib as L
dev = torch.device("cuda:0")
V = 32000
for M in (4096, 16384, 49152):
    lg = torch.randn(M, V, device=dev)
    u = torch.rand(M, device=dev)
    out = torch.empty(M, dtype=torch.int32, device=dev)
    a = L.SampleArgs()
    a.logits, a.ld, a.V, a.M, a.u, a.out = lg.data_ptr(), V, V, M, u.data_ptr(), out.data_ptr()
    st = L.stream_ptr()
    for _ in range(2):
        L.check(L.lib.nmtgan_sample_rows(C.byref(a), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.check(L.lib.nmtgan_sample_rows(C.byref(a), st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("sampler M=%6d  %8.3f ms  %7.1f GB/s" % (M, ms, M * (4.0 * V + 8) / ms / 1e6), flush=True)
    del lg
