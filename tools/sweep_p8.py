"""Sweep of the two persistent GEMM kernels over the row counts of the rollout steps (run once per kernel choice:
NMTGAN_GEMM_P8=0 -> 256x128 tiles, NMTGAN_GEMM_P8=2 -> 256x256 tiles, unset -> the dispatch heuristic).
usage: python tools/sweep_p8.py [fp16|bf16]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nmt_gan_amd import _lib as L

dev = torch.device("cuda:0")
OP = sys.argv[1] if len(sys.argv) > 1 else "fp16"
TDT, OPC = L.op_dtype_of(OP)


def run(M, N, K, out_f32, iters=8):
    g = torch.Generator(device=dev).manual_seed(1)
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(TDT)
    W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(TDT)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else TDT)
    a = L.GemmArgs()
    a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = OPC, L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, K, K
    a.bias, a.n_seg, a.seg_width = bias.data_ptr(), 1, N
    a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, out_f32, 1.0
    st = L.stream_ptr()
    for _ in range(2):
        L.check(L.lib.nmtgan_gemm(C.byref(a), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.check(L.lib.nmtgan_gemm(C.byref(a), st))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    shapes = [("qkv", 1536, 512, 0), ("ffn1", 2048, 512, 0), ("ffn2", 512, 2048, 1), ("cq", 512, 512, 0)]
    for name, N, K, f32 in shapes:
        print(name, " ".join("%d:%.1f" % (M, run(M, N, K, f32)) for M in range(2048, 50177, 2048)), flush=True)
