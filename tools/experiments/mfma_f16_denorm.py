import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch
from nmt_gan_amd import _lib as L
dev = torch.device("cuda:0")
M, N, K = 128, 128, 64
A = torch.ones(M, K, device=dev, dtype=torch.float16)
for val in (1e-5, 3e-6, 2.4e-5, 1e-7):
    W = torch.full((N, K), val, device=dev, dtype=torch.float16)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    bias = torch.zeros(N, device=dev)
    a = L.GemmArgs()
    a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = L.NMTGAN_F16, L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, K, K
    a.bias, a.n_seg, a.seg_width = bias.data_ptr(), 1, N
    a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, 1, 1.0
    L.check(L.lib.nmtgan_gemm(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    print("w=%g stored %g: sum over K=64 -> %g (expected %g)" % (val, float(W[0, 0]), float(out[0, 0]), float(W[0, 0]) * K))
