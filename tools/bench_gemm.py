"""Micro-benchmark of nmtgan_gemm on the shapes of the C2 workload (HIP events, random data), with torch.matmul
(hipBLASLt / rocBLAS) on the same box as a YARDSTICK for the dense shapes - a measuring stick only, the library
never calls it.  usage: python tools/bench_gemm.py [bf16|fp16] [k512|addln|addln-sweep]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nmt_gan_amd import _lib as L

dev = torch.device("cuda:0")
OP = sys.argv[1] if len(sys.argv) > 1 else "fp16"
TDT, OPC = L.op_dtype_of(OP)
SHAPES = [  # name, M, N, K, conv(P,T,E) or None, out_f32
    ("square 4096^3", 4096, 4096, 4096, None, 0),
    ("square 8192^3", 8192, 8192, 8192, None, 0),
    ("qkv    M49152 N1536 K512", 49152, 1536, 512, None, 0),
    ("proj   M49152 N512  K512", 49152, 512, 512, None, 1),
    ("ffn1   M49152 N2048 K512", 49152, 2048, 512, None, 0),
    ("ffn2   M49152 N512  K2048", 49152, 512, 2048, None, 1),
    ("proj   M8192  N512  K512", 8192, 512, 512, None, 1),
    ("proj   M1024  N512  K512", 1024, 512, 512, None, 1),
    ("logits M4096  N32000 K512", 4096, 32000, 512, None, 1),
    ("logits M16384 N32000 K512", 16384, 32000, 512, None, 1),
    ("conv w1  S4096", 4096 * 50, 110, 512, (50, 50, 512), 1),
    ("conv w5  S4096", 4096 * 46, 150, 5 * 512, (46, 50, 512), 1),
    ("conv w9  S4096", 4096 * 42, 190, 9 * 512, (42, 50, 512), 1),
    ("conv w17 S4096", 4096 * 34, 270, 17 * 512, (34, 50, 512), 1),
    ("conv w21 S4096", 4096 * 30, 310, 21 * 512, (30, 50, 512), 1),
    ("conv w29 S4096", 4096 * 22, 390, 29 * 512, (22, 50, 512), 1),
    ("conv w45 S4096", 4096 * 6, 550, 45 * 512, (6, 50, 512), 1),
    ("conv w13 S4096", 4096 * 38, 230, 13 * 512, (38, 50, 512), 1),
    ("conv w25 S4096", 4096 * 26, 350, 25 * 512, (26, 50, 512), 1),
    ("conv w37 S4096", 4096 * 14, 470, 37 * 512, (14, 50, 512), 1),
    ("conv w49 S4096", 4096 * 2, 590, 49 * 512, (2, 50, 512), 1),
    ("highway M4096 N4608 K4608", 4096, 4608, 4608, None, 1),
    # bucket GEMMs of the partial-window experiment: 8 windows per rollout, first taps skipped, raw fp32 out
    ("bucket w25 k17 S4096x8", 4096 * 8, 350, 17 * 512, (8, 50, 512), 2),
    ("bucket w25 k9  S4096x8", 4096 * 8, 350, 9 * 512, (8, 50, 512), 2),
    ("bucket w25 k1  S4096x8", 4096 * 8, 350, 1 * 512, (8, 50, 512), 2),
    ("bucket w41 k33 S4096x8", 4096 * 8, 510, 33 * 512, (8, 50, 512), 2),
    ("bucket w41 k17 S4096x8", 4096 * 8, 510, 17 * 512, (8, 50, 512), 2),
    ("full   w25 S4096x26", 4096 * 26, 350, 25 * 512, (26, 50, 512), 1),
    ("full   w41 S4096x10", 4096 * 10, 510, 41 * 512, (10, 50, 512), 1),
]


def run(name, M, N, K, conv, out_f32, iters=10):
    g = torch.Generator(device=dev).manual_seed(1)
    a = L.GemmArgs()
    a.op_dtype = OPC
    if conv:
        P, T, E = conv
        S = M // P
        A = (torch.rand(S * T, E, device=dev, generator=g) * 2 - 1).to(TDT)
        a.lda, a.a_group, a.a_group_stride = E, P, T * E
    else:
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(TDT)
        a.lda = K
    W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(TDT)
    a.A, a.Wt, a.M, a.N, a.K = A.data_ptr(), W.data_ptr(), M, N, K
    keep = []
    if conv and out_f32 == 2:          # window rows, LINEAR epilogue, raw fp32 out
        a.epilogue = L.EPI_LINEAR
        bias = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
        a.bias, a.n_seg, a.seg_width = bias.data_ptr(), 1, N
        a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, 1, 1.0
        keep += [bias, out]
    elif conv:
        a.epilogue = L.EPI_CONVMAX
        sc = torch.ones(N, device=dev); sh = torch.zeros(N, device=dev)
        pooled = torch.zeros(M // conv[0], 4608, device=dev)
        a.scale, a.shift, a.pooled, a.ld_pool = sc.data_ptr(), sh.data_ptr(), pooled.data_ptr(), 4608
        keep += [sc, sh, pooled]
    else:
        a.epilogue = L.EPI_LINEAR
        bias = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else TDT)
        a.bias, a.n_seg, a.seg_width = bias.data_ptr(), 1, N
        a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, out_f32, 1.0
        keep += [bias, out]
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
    ref = ""
    if not conv:                           # same product through torch.matmul (16-bit output) as a yardstick
        for _ in range(2):
            torch.matmul(A, W.t())
        e0.record()
        for _ in range(iters):
            torch.matmul(A, W.t())
        e1.record()
        torch.cuda.synchronize()
        ms_t = e0.elapsed_time(e1) / iters
        ref = "   torch.matmul %8.3f ms %7.1f TFLOP/s" % (ms_t, 2.0 * M * N * K / ms_t / 1e9)
    print("%-28s %8.3f ms  %7.1f TFLOP/s%s" % (name, ms, 2.0 * M * N * K / ms / 1e9, ref), flush=True)


def run_addln(M, K, iters=10):
    """projection + residual + LayerNorm (N = 512): the fused epilogue against GEMM (fp32 out) + add_layernorm"""
    N = 512
    g = torch.Generator(device=dev).manual_seed(2)
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(TDT)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * (3.0 / K) ** 0.5).to(TDT)
    bias = torch.zeros(N, device=dev); gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
    x = torch.randn(M, N, device=dev, generator=g); y = torch.empty(M, N, device=dev); xo = torch.empty(M, N, device=dev, dtype=TDT)
    f = L.GemmArgs()
    f.op_dtype, f.epilogue, f.A, f.Wt, f.M, f.N, f.K, f.lda = OPC, L.EPI_ADDLN, A.data_ptr(), W.data_ptr(), M, N, K, K
    f.bias, f.ln_x, f.ln_gamma, f.ln_beta, f.ln_out = bias.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), xo.data_ptr()
    a = L.GemmArgs()
    a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = OPC, L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, K, K
    a.bias, a.n_seg, a.seg_width = bias.data_ptr(), 1, N
    a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = y.data_ptr(), N, 1, 1.0
    n = L.AddLnArgs()
    n.x, n.y, n.gamma, n.beta, n.M, n.d = x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, N
    n.out_f32, n.out_op, n.op_dtype = x.data_ptr(), xo.data_ptr(), OPC
    st = L.stream_ptr()

    def fused():
        L.check(L.lib.nmtgan_gemm(C.byref(f), st))

    def split():
        L.check(L.lib.nmtgan_gemm(C.byref(a), st))
        L.check(L.lib.nmtgan_add_layernorm(C.byref(n), st))

    ms = []
    for fn in (fused, split):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / iters)
    gb = M * N * (10 + 2.0 * K / N) / 1e9        # x in/out fp32 + op-dtype copy + A
    print("addln  M%-6d K%-5d fused %8.3f ms (%6.2f TB/s of its 10 B/elt + A)   gemm + add_layernorm %8.3f ms"
          % (M, K, ms[0], gb / ms[0], ms[1]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "addln":
        for M in (1024, 4096, 8192, 16384, 32768, 49152):
            for K in (512, 2048):
                run_addln(M, K)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "addln-sweep":       # the row counts of the rollout steps, K = 512
        for M in range(2048, 50177, 2048):
            run_addln(M, 512)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "k512":              # the K = 512 projection family over the rollout row counts,
        for M in (49152, 24576, 8192, 2048):                     # each beside the torch.matmul yardstick (16-bit output)
            run("qkv    M%-5d N1536 K512" % M, M, 1536, 512, None, 0)
            run("cq     M%-5d N512  K512" % M, M, 512, 512, None, 0)
            run("o      M%-5d N512  K512 (fp32 out)" % M, M, 512, 512, None, 1)
            run("ffn1   M%-5d N2048 K512" % M, M, 2048, 512, None, 0)
            run("ffn2   M%-5d N512  K2048 (fp32 out)" % M, M, 512, 2048, None, 1)
        sys.exit(0)
    for s in SHAPES:
        run(*s)
