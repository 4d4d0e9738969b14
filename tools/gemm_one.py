"""Run one plain GEMM shape through nmtgan_gemm repeatedly (for rocprofv3 / PMC passes).
usage: gemm_one.py M N K [iters]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nmt_gan_amd import _lib as L

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).bfloat16()
W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).bfloat16()
if os.environ.get('ZERO'):
    A.zero_(); W.zero_()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(N, device=dev)
a = L.GemmArgs()
a.op_dtype, a.epilogue = L.NMTGAN_BF16, L.EPI_LINEAR
a.A, a.Wt, a.M, a.N, a.K, a.lda = A.data_ptr(), W.data_ptr(), M, N, K, K
a.bias, a.n_seg, a.seg_width = bias.data_ptr(), 1, N
a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, 0, 1.0
st = L.stream_ptr()
for _ in range(2):
    L.check(L.lib.nmtgan_gemm(C.byref(a), st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    L.check(L.lib.nmtgan_gemm(C.byref(a), st))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
ref = (A[:256].float() @ W.float().T)
err = (out[:256].float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
print("M%d N%d K%d  %.3f ms  %.1f TFLOP/s  rel err (first 256 rows) %.2e" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err), flush=True)
