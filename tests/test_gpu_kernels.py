"""GPU parity of the individual gfx950 kernels, called through the C ABI (ctypes), against
fp64 restatements.  Integer outputs are bit-exact; fp32-operand paths within 1e-5, bf16-operand
paths within the bf16 input-rounding bound stated at each test."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from nmt_gan_amd import _lib
    return _lib


_TD = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def _code(L, op):
    return {"bf16": L.NMTGAN_BF16, "fp16": L.NMTGAN_F16, "fp32": L.NMTGAN_F32}[op]


def _gemm(L, A, Wt, bias=None, relu=0, op="bf16", a_group=0, a_group_stride=0, lda=None, M=None, K=None,
          segs=None, conv=None):
    """Run nmtgan_gemm.  A, Wt torch tensors already in op dtype on the GPU."""
    a = L.GemmArgs()
    a.op_dtype = _code(L, op)
    a.A, a.Wt = A.data_ptr(), Wt.data_ptr()
    N, Kw = Wt.shape
    a.M = A.shape[0] if M is None else M
    a.N, a.K = N, Kw if K is None else K
    a.lda = A.stride(0) if lda is None else lda
    a.a_group, a.a_group_stride = a_group, a_group_stride
    outs = []
    if conv is None:
        a.epilogue = L.EPI_LINEAR
        a.bias = None if bias is None else bias.data_ptr()
        a.relu = relu
        segs = segs or [(N, True, 1.0)]
        a.n_seg = len(segs)
        a.seg_width = segs[0][0]
        for i, (w, is_f32, sc) in enumerate(segs):
            o = torch.full((a.M, w), float("nan"), device=A.device, dtype=torch.float32 if is_f32 else A.dtype)
            outs.append(o)
            a.out[i], a.ldo[i], a.out_f32[i], a.seg_scale[i] = o.data_ptr(), w, 1 if is_f32 else 0, sc
    else:
        scale, shift, pooled = conv
        a.epilogue = L.EPI_CONVMAX
        a.scale, a.shift, a.pooled, a.ld_pool = scale.data_ptr(), shift.data_ptr(), pooled.data_ptr(), pooled.stride(0)
        outs = [pooled]
    L.check(L.lib.nmtgan_gemm(C.byref(a), L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("op", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 512), (1, 96, 2048), (257, 1536, 64), (1000, 130, 1088)])
def test_gemm_linear(cuda, op, M, N, K):
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    dt = _TD[op]
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dt).to(cuda)
    # asymmetric weights catch a transposed C write (cdna guide section 3)
    Wt = (torch.rand(N, K, generator=g) * 2 - 1 + torch.arange(N)[:, None] * 1e-3).to(dt).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    (out,) = _gemm(L, A, Wt, bias=bias, relu=1, op=op)
    ref = torch.relu(A.double() @ Wt.double().T + bias.double())
    err = (out.double() - ref).abs().max().item()
    # operands are exactly representable in the op dtype; only fp32 accumulation order differs
    tol = 2e-6 * K ** 0.5 * (K / 3) ** 0.5 + 1e-5
    assert err < tol, (err, tol)


@pytest.mark.parametrize("op", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("M,N,K,relu", [(10000, 2304, 768, 1), (70000, 512, 512, 0), (9000, 1300, 1024, 0)])
def test_gemm_persistent_many_tiles(cuda, op, M, N, K, relu):
    """Shapes with more output tiles than CUs: every workgroup of the persistent kernels walks several
    tiles (cross-tile prefetch, epilogue under the next tile's loads, ragged last M/N tiles).  Checked in
    full against a float64-accumulated torch product of the same operands."""
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dt = _TD[op]
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dt).to(cuda)
    Wt = (torch.rand(N, K, generator=g) * 2 - 1 + torch.arange(N)[:, None] * 1e-4).to(dt).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    (out,) = _gemm(L, A, Wt, bias=bias, relu=relu, op=op)
    worst = 0.0
    for r0 in range(0, M, 4096):
        ref = A[r0:r0 + 4096].double() @ Wt.double().T + bias.double()
        if relu:
            ref = torch.relu(ref)
        worst = max(worst, (out[r0:r0 + 4096].double() - ref).abs().max().item())
    tol = 2e-6 * K ** 0.5 * (K / 3) ** 0.5 + 1e-5
    assert worst < tol, (worst, tol)


@pytest.mark.parametrize("op", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,f32_out,relu", [(49152, 512, 2048, True, 0), (40960, 512, 2048, True, 0),
                                                 (45056, 2048, 512, False, 1), (33792, 512, 1536, True, 0)])
def test_gemm_row_cut_same_bits(cuda, op, M, N, K, f32_out, relu):
    """Row counts at which the kernel choice cuts a LINEAR problem by rows (full rounds of 256x256 tiles first, the
    remainder on its own): every row must carry the bits it gets when its 8192-row block is a problem of its own - the
    cut moves operand and output pointers and may hand the two parts to different kernels - and the whole result is
    checked against a float64 product."""
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dt = _TD[op]
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dt).to(cuda)
    Wt = (torch.rand(N, K, generator=g) * 2 - 1 + torch.arange(N)[:, None] * 1e-4).to(dt).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    (out,) = _gemm(L, A, Wt, bias=bias, relu=relu, op=op, segs=[(N, f32_out, 1.0)])
    worst, scale = 0.0, 0.0
    for r0 in range(0, M, 8192):
        blk = A[r0:r0 + 8192]
        (part,) = _gemm(L, blk, Wt, bias=bias, relu=relu, op=op, segs=[(N, f32_out, 1.0)])
        assert torch.equal(part, out[r0:r0 + 8192]), r0
        ref = blk.double() @ Wt.double().T + bias.double()
        if relu:
            ref = torch.relu(ref)
        worst = max(worst, (part.double() - ref).abs().max().item())
        scale = max(scale, ref.abs().max().item())
    # fp32 outputs: accumulation order only; 16-bit outputs: one rounding of the stored value (2^-8 covers bf16)
    tol = 2e-6 * K ** 0.5 * (K / 3) ** 0.5 + 1e-5 if f32_out else scale * 2.0 ** -8
    assert worst < tol, (worst, tol)


@pytest.mark.parametrize("op", ["bf16", "fp16"])
def test_gemm_segments_and_scale(cuda, op):
    L = _lib()
    g = torch.Generator().manual_seed(3)
    M, d = 70, 64
    A = torch.randn(M, d, generator=g).to(_TD[op]).to(cuda)
    Wt = (torch.randn(3 * d, d, generator=g) * 0.2).to(_TD[op]).to(cuda)
    bias = torch.randn(3 * d, generator=g).to(cuda)
    q, k, v = _gemm(L, A, Wt, bias=bias, op=op, segs=[(d, True, 0.25), (d, False, 1.0), (d, False, 1.0)])
    ref = A.double() @ Wt.double().T + bias.double()
    tol = 3e-2 if op == "bf16" else 4e-3                           # output rounding: 2^-9 / 2^-12 relative
    assert (q.double() - ref[:, :d] * 0.25).abs().max() < 1e-4
    assert (k.double() - ref[:, d:2 * d]).abs().max() < tol
    assert (v.double() - ref[:, 2 * d:]).abs().max() < tol


@pytest.mark.parametrize("M,N,K", [(300, 1536, 512), (70000, 512, 512), (257, 2048, 512)])
def test_gemm_fp16_outputs_saturate(cuda, M, N, K):
    """fp16 output segments beyond the half range are stored as +-65504, never as inf (the wide-store epilogues
    of both persistent kernels and the small-shape kernel); values inside the range round to nearest."""
    L = _lib()
    g = torch.Generator().manual_seed(M + N)
    A = (torch.rand(M, K, generator=g) * 2 - 1).half().to(cuda)
    Wt = (torch.rand(N, K, generator=g) * 2 - 1).half().to(cuda)
    bias = torch.zeros(N)
    bias[::7] = 7.0e4                                              # columns pushed past +65504
    bias[3::7] = -9.0e4                                            # ... and past -65504
    bias = bias.to(cuda)
    (out,) = _gemm(L, A, Wt, bias=bias, op="fp16", segs=[(N, False, 1.0)])
    assert out.dtype == torch.float16 and bool(torch.isfinite(out).all())
    ref = (A.double() @ Wt.double().T + bias.double()).clamp(-65504.0, 65504.0)
    assert float((out[:, ::7].double() - 65504.0).abs().max()) == 0.0
    assert float((out[:, 3::7].double() + 65504.0).abs().max()) == 0.0
    inside = ref.abs() < 6.0e4
    err = ((out.double() - ref).abs() / ref.abs().clamp_min(1.0))[inside].max().item()
    assert err < 1e-3, err


@pytest.mark.parametrize("op", ["fp16", "bf16", "fp32"])
def test_gemm_convmax(cuda, op):
    """width-w VALID conv + folded BN + ReLU + max over time (cnn_discriminator.py:94-105, 469)."""
    L = _lib()
    g = torch.Generator().manual_seed(11)
    dt = _TD[op]
    S, T, E = 37, 12, 64
    e = (torch.rand(S, T, E, generator=g) * 2 - 1).to(dt).to(cuda)
    col = 0
    Fp = 512
    pooled = torch.zeros(S, Fp, device=cuda)
    refs = []
    for w, n in [(1, 110), (5, 150), (9, 190)]:
        P = T - w + 1
        Wt = (torch.randn(n, w * E, generator=g) * 0.1).to(dt).to(cuda)
        scale = (1 + 0.3 * torch.randn(n, generator=g)).to(cuda)
        shift = (0.3 * torch.randn(n, generator=g)).to(cuda)
        _gemm(L, e.view(S * T, E), Wt, op=op, a_group=P, a_group_stride=T * E, lda=E, M=S * P, K=w * E,
              conv=(scale, shift, pooled[:, col:]))
        win = e.double().unfold(1, w, 1).permute(0, 1, 3, 2).reshape(S, P, w * E)
        c = win @ Wt.double().T * scale.double() + shift.double()
        refs.append(torch.relu(c).max(dim=1).values)
        col += n
    ref = torch.cat(refs, dim=1)
    err = (pooled[:, :col].double() - ref).abs().max().item()
    assert err < 2e-4, err
    assert pooled[:, col:].abs().max().item() == 0.0


@pytest.mark.parametrize("op", ["fp16", "fp32"])
def test_gemm_grouped_equals_single_launches(cuda, op):
    """nmtgan_gemm_grouped: conv problems of different widths, row counts and window ranges in ONE persistent launch
    give the bits of one nmtgan_gemm call each (same k order per output element); with the additive term (the part of
    a window's sum that comes from a table) they match the fp64 restatement; LINEAR problems with an fp32 output or one
    in the operand type too."""
    L = _lib()
    g = torch.Generator().manual_seed(21)
    dt = _TD[op]
    T, E, Fp = 20, 64, 1024
    probs, keep = [], []
    pooled_g = {}
    for (S, w, n, p0) in [(37, 5, 150, 0), (300, 9, 190, 3), (5, 17, 270, 0), (64, 13, 230, 2), (1200, 2, 110, 0)]:
        e = (torch.rand(S, T, E, generator=g) * 2 - 1).to(dt).to(cuda)
        Wt = (torch.randn(n, w * E, generator=g) * 0.1).to(dt).to(cuda)
        scale = (1 + 0.3 * torch.randn(n, generator=g)).to(cuda)
        shift = (0.3 * torch.randn(n, generator=g)).to(cuda)
        P = T - w + 1 - p0
        outs = []
        for _ in range(2):                                          # [0] single launch, [1] grouped
            pooled = torch.zeros(S, Fp, device=cuda)
            a = L.GemmArgs()
            a.op_dtype, a.epilogue = _code(L, op), L.EPI_CONVMAX
            a.A, a.Wt = e.data_ptr() + p0 * E * e.element_size(), Wt.data_ptr()
            a.M, a.N, a.K, a.lda, a.a_group, a.a_group_stride = S * P, n, w * E, E, P, T * E
            a.scale, a.shift, a.pooled, a.ld_pool = scale.data_ptr(), shift.data_ptr(), pooled.data_ptr(), Fp
            outs.append((a, pooled))
        L.check(L.lib.nmtgan_gemm(C.byref(outs[0][0]), L.stream_ptr()), "gemm")
        probs.append(outs[1][0])
        keep.append((e, Wt, scale, shift, outs))
    arr = (L.GemmArgs * len(probs))(*probs)
    L.check(L.lib.nmtgan_gemm_grouped(arr, len(probs), L.stream_ptr()), "gemm_grouped")
    torch.cuda.synchronize()
    for e, Wt, scale, shift, outs in keep:
        assert float(outs[0][1].abs().max()) > 0
        assert torch.equal(outs[0][1], outs[1][1])
    # additive term: windows p0 .. p0 + nq - 1 of S sentences, taps from k0 on multiplied, the rest from a table
    S, w, n, nq, p0, k0, nblk = 90, 9, 190, 3, 4, 2, 7
    e = (torch.rand(S, T, E, generator=g) * 2 - 1).to(dt).to(cuda)
    Wt = (torch.randn(n, w * E, generator=g) * 0.1).to(dt).to(cuda)
    scale = (1 + 0.3 * torch.randn(n, generator=g)).to(cuda)
    shift = (0.3 * torch.randn(n, generator=g)).to(cuda)
    Pw = T - w + 1
    add = torch.randn(nblk, Pw, n, generator=g).to(cuda)
    blk = torch.randint(0, nblk, (S,), generator=g, dtype=torch.int32).to(cuda)
    pooled = torch.zeros(S, Fp, device=cuda)
    a = L.GemmArgs()
    a.op_dtype, a.epilogue = _code(L, op), L.EPI_CONVMAX
    a.A = e.data_ptr() + (p0 + k0) * E * e.element_size()
    a.Wt = Wt.data_ptr() + k0 * E * Wt.element_size()
    a.M, a.N, a.K, a.ldw, a.lda, a.a_group, a.a_group_stride = S * nq, n, (w - k0) * E, w * E, E, nq, T * E
    a.scale, a.shift, a.pooled, a.ld_pool = scale.data_ptr(), shift.data_ptr(), pooled.data_ptr(), Fp
    a.add, a.add_blk, a.add_rows, a.add_row0 = add.data_ptr(), blk.data_ptr(), Pw, p0
    L.check(L.lib.nmtgan_gemm_grouped(C.byref(a), 1, L.stream_ptr()), "gemm_grouped add")
    torch.cuda.synchronize()
    win = e.double().unfold(1, w, 1).permute(0, 1, 3, 2)[:, p0:p0 + nq, k0:].reshape(S, nq, (w - k0) * E)
    c = win @ Wt.double().view(n, w, E)[:, k0:].reshape(n, -1).T + add.double()[blk.long(), p0:p0 + nq]
    ref = torch.relu(c * scale.double() + shift.double()).max(dim=1).values
    assert (pooled[:, :n].double() - ref).abs().max().item() < 2e-4
    # LINEAR problems with an fp32 output
    lin, keepl = [], []
    for (M, N, K) in [(300, 200, 512), (70, 130, 1024), (1000, 64, 256)]:
        A = (torch.rand(M, K, generator=g) * 2 - 1).to(dt).to(cuda)
        W = (torch.rand(N, K, generator=g) * 2 - 1).to(dt).to(cuda)
        single = _gemm(L, A, W, op=op)[0]
        out = torch.full((M, N), float("nan"), device=cuda)
        a = L.GemmArgs()
        a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = _code(L, op), L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, K, K
        a.n_seg, a.seg_width = 1, N
        a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, 1, 1.0
        lin.append(a)
        keepl.append((A, W, single, out))
    arr = (L.GemmArgs * len(lin))(*lin)
    L.check(L.lib.nmtgan_gemm_grouped(arr, len(lin), L.stream_ptr()), "gemm_grouped linear")
    torch.cuda.synchronize()
    for A, W, single, out in keepl:
        assert torch.equal(single, out)
    # ... and with an output in the operand type, into a wider row (ldo > N)
    lin, keepo = [], []
    for (A, W, single, _) in keepl:
        M, N = single.shape
        single_op = _gemm(L, A, W, op=op, segs=[(N, False, 1.0)])[0]
        out = torch.zeros((M, N + 24), device=cuda, dtype=dt)
        a = L.GemmArgs()
        a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = _code(L, op), L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, A.shape[1], A.shape[1]
        a.n_seg, a.seg_width = 1, N
        a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N + 24, 0, 1.0
        lin.append(a)
        keepo.append((single_op, out))
    arr = (L.GemmArgs * len(lin))(*lin)
    L.check(L.lib.nmtgan_gemm_grouped(arr, len(lin), L.stream_ptr()), "gemm_grouped linear op-dtype output")
    torch.cuda.synchronize()
    for single_op, out in keepo:
        N = single_op.shape[1]
        assert torch.equal(single_op, out[:, :N]) and float(out[:, N:].abs().max()) == 0.0


def test_gemm_grouped_many_problems(cuda):
    """More problems than one grouped launch holds (24): 31 conv products with random row counts, widths, window ranges
    and filter counts - single tiles, ragged M and N, k-loops of 4 to 40 tiles - against one nmtgan_gemm call each."""
    L = _lib()
    rng = np.random.default_rng(5)
    g = torch.Generator().manual_seed(9)
    T, E, Fp = 24, 64, 768
    e = (torch.rand(700, T, E, generator=g) * 2 - 1).half().to(cuda)
    probs, singles, groupeds, keep = [], [], [], []
    for i in range(31):
        w = int(rng.integers(4, 21))
        S = int(rng.choice([1, 3, 40, 129, 300, 700]))
        n = int(rng.integers(1, 700))
        p0 = int(rng.integers(0, T - w + 1))
        P = int(rng.integers(1, T - w + 2 - p0))
        Wt = (torch.randn(n, w * E, generator=g) * 0.1).half().to(cuda)
        scale = (1 + 0.3 * torch.randn(n, generator=g)).to(cuda)
        shift = (0.3 * torch.randn(n, generator=g)).to(cuda)
        pair = []
        for _ in range(2):
            pooled = torch.zeros(S, Fp, device=cuda)
            a = L.GemmArgs()
            a.op_dtype, a.epilogue = L.NMTGAN_F16, L.EPI_CONVMAX
            a.A, a.Wt = e.data_ptr() + p0 * E * 2, Wt.data_ptr()
            a.M, a.N, a.K, a.lda, a.a_group, a.a_group_stride = S * P, n, w * E, E, P, T * E
            a.scale, a.shift, a.pooled, a.ld_pool = scale.data_ptr(), shift.data_ptr(), pooled.data_ptr(), Fp
            pair.append((a, pooled))
        L.check(L.lib.nmtgan_gemm(C.byref(pair[0][0]), L.stream_ptr()), "gemm")
        probs.append(pair[1][0])
        singles.append(pair[0][1])
        groupeds.append(pair[1][1])
        keep.append((Wt, scale, shift))
    arr = (L.GemmArgs * len(probs))(*probs)
    L.check(L.lib.nmtgan_gemm_grouped(arr, len(probs), L.stream_ptr()), "gemm_grouped")
    torch.cuda.synchronize()
    for i, (a_, b_) in enumerate(zip(singles, groupeds)):
        assert torch.equal(a_, b_), i
    assert sum(float(x.abs().max()) > 0 for x in singles) >= 25
    # argument errors are reported before anything is launched
    bad = (L.GemmArgs * 2)(probs[0], probs[1])
    bad[1].op_dtype = L.NMTGAN_F32
    assert L.lib.nmtgan_gemm_grouped(bad, 2, L.stream_ptr()) != 0 and b"share op dtype" in L.lib.nmtgan_last_error()
    assert L.lib.nmtgan_gemm_grouped(None, 0, L.stream_ptr()) == 0


@pytest.mark.parametrize("d", [64, 128, 512, 1024])
def test_add_layernorm(cuda, d):
    L = _lib()
    g = torch.Generator().manual_seed(d)
    M = 77
    x = torch.randn(M, d, generator=g).to(cuda)
    y = torch.randn(M, d, generator=g).to(cuda)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(cuda)
    beta = (0.1 * torch.randn(d, generator=g)).to(cuda)
    zero = torch.zeros(M, dtype=torch.uint8, device=cuda)
    zero[5] = 1
    of = torch.empty(M, d, device=cuda)
    oo = torch.empty(M, d, device=cuda, dtype=torch.bfloat16)
    a = L.AddLnArgs()
    a.x, a.y, a.gamma, a.beta, a.M, a.d = x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, d
    a.out_f32, a.out_op, a.op_dtype, a.zero_row = of.data_ptr(), oo.data_ptr(), L.NMTGAN_BF16, zero.data_ptr()
    L.check(L.lib.nmtgan_add_layernorm(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    s = (x + y).double()
    mean = s.mean(-1, keepdim=True)
    var = ((s - mean) ** 2).mean(-1, keepdim=True)
    ref = (s - mean) * torch.rsqrt(var + 1e-6) * gamma.double() + beta.double()
    ref[5] = 0
    assert (of.double() - ref).abs().max().item() < 2e-5
    assert torch.equal(oo, of.bfloat16())


def _gemm_addln(L, A, Wt, bias, x, gamma, beta, op, M=None):
    """nmtgan_gemm with the ADDLN epilogue: returns (x rewritten in place, its op-dtype copy)."""
    a = L.GemmArgs()
    dt, a.op_dtype = L.op_dtype_of(op)
    a.epilogue = L.EPI_ADDLN
    a.A, a.Wt = A.data_ptr(), Wt.data_ptr()
    a.M = A.shape[0] if M is None else M
    a.N, a.K = Wt.shape
    a.lda = A.stride(0)
    a.bias = None if bias is None else bias.data_ptr()
    xo = torch.full((a.M, a.N), float("nan"), device=A.device, dtype=dt)
    a.ln_x, a.ln_gamma, a.ln_beta, a.ln_out = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), xo.data_ptr()
    L.check(L.lib.nmtgan_gemm(C.byref(a), L.stream_ptr()), "gemm addln")
    torch.cuda.synchronize()
    return x, xo


@pytest.mark.parametrize("op", ["fp16", "bf16"])
@pytest.mark.parametrize("M,K", [(1, 512), (63, 512), (64, 2048), (200, 128), (4113, 512), (40000, 512), (17000, 2048)])
def test_gemm_addln_fused(cuda, op, M, K):
    """Projection + residual + LayerNorm in one kernel (N = 512) against the float64 formula of model.py:725-739 on the
    same 16-bit operands; rows past M untouched; a row's result does not depend on how many rows the call has."""
    L = _lib()
    N = 512
    g = torch.Generator(device="cpu").manual_seed(M + K)
    dt, _ = L.op_dtype_of(op)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dt).to(cuda)
    Wt = ((torch.rand(N, K, generator=g) * 2 - 1) * (3.0 / K) ** 0.5 + torch.arange(N)[:, None] * 1e-4).to(dt).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).to(cuda)
    beta = (0.1 * torch.randn(N, generator=g)).to(cuda)
    x0 = torch.randn(M + 3, N, generator=g).to(cuda)
    x = x0.clone()
    _, xo = _gemm_addln(L, A, Wt, bias, x, gamma, beta, op, M=M)
    v = x0[:M].double() + A.double() @ Wt.double().T + bias.double()
    mean = v.mean(1, keepdim=True)
    var = ((v - mean) ** 2).mean(1, keepdim=True)
    ref = (v - mean) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()
    err = (x[:M].double() - ref).abs().max().item()
    assert err < 2e-5, err
    assert torch.equal(x[M:], x0[M:])                       # rows past M are not touched
    assert torch.equal(xo, x[:M].to(dt))                    # the op-dtype copy is the rounded fp32 output
    if M >= 200:                                            # batching independence: the first rows alone give the same bits
        x2 = x0[:150].clone()
        _gemm_addln(L, A[:150].contiguous(), Wt, bias, x2, gamma, beta, op)
        assert torch.equal(x2, x[:150])


@pytest.mark.parametrize("M,N,K", [(300, 512, 512), (4097, 1536, 512), (1000, 512, 2048)])
def test_split_precision_gemm(cuda, M, N, K):
    """fp32 GEMM as ONE fp16 product over hi/lo operand splits: nmtgan_split3 gives [A_hi | A_lo | A_hi] exactly as the
    host formula does, and nmtgan_gemm over 3K against [W_hi | W_hi | W_lo] matches the float64 product of the fp32
    operands to fp32-GEMM accuracy (plain fp16 operands: ~5e-4 at these sizes)."""
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * 1.3).to(cuda)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(cuda)
    A3 = torch.empty(M, 3 * K, device=cuda, dtype=torch.float16)
    L.check(L.lib.nmtgan_split3(A.data_ptr(), K, M, K, A3.data_ptr(), L.stream_ptr()), "split3")
    torch.cuda.synchronize()
    hi = A.half()
    lo = (A - hi.float()).half()
    assert torch.equal(A3, torch.cat([hi, lo, hi], dim=1))
    W3 = L.split3_weights(W)
    a = L.GemmArgs()
    a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = L.NMTGAN_F16, L.EPI_LINEAR, A3.data_ptr(), W3.data_ptr(), M, N, 3 * K, 3 * K
    out = torch.empty(M, N, device=cuda)
    a.n_seg, a.seg_width = 1, N
    a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), N, 1, 1.0
    L.check(L.lib.nmtgan_gemm(C.byref(a), L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    plain = float(((hi.double() @ W.half().double().T) - ref).abs().max() / ref.abs().max())
    print("split-precision GEMM rel err %.3g (plain fp16 operands %.3g)" % (err, plain))
    assert err < 3e-6, err
    # values beyond the fp16 range: hi is clamped to +-65504 and lo carries the rest (exact up to twice the range)
    big = torch.tensor([[1.0e5, -7.0e4, 65504.0, -131000.0, 3.0, 0.0, -65520.0, 1.25e5]], device=cuda)
    img = torch.empty(1, 24, device=cuda, dtype=torch.float16)
    L.check(L.lib.nmtgan_split3(big.data_ptr(), 8, 1, 8, img.data_ptr(), L.stream_ptr()), "split3")
    torch.cuda.synchronize()
    assert torch.isfinite(img.float()).all()
    assert torch.equal(img[:, :8], img[:, 16:])
    back = img[:, :8].float() + img[:, 8:16].float()
    assert ((back - big).abs() <= big.abs() * 2.0 ** -11).all(), (back, big)


def _attn_ref(q, keys, vals, heads):
    """fp64 softmax attention of one row: q [d], keys/vals [n, d]."""
    d = q.shape[0]
    dk = d // heads
    out = torch.zeros(d, dtype=torch.float64)
    for h in range(heads):
        sl = slice(h * dk, (h + 1) * dk)
        s = keys[:, sl] @ q[sl]
        w = torch.softmax(s, 0)
        out[sl] = w @ vals[:, sl]
    return out


@pytest.mark.parametrize("q_op", [0, 1])
@pytest.mark.parametrize("kv", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("d,heads", [(64, 4), (256, 4), (512, 8), (1024, 16)])
def test_attn_rows_two_segments(cuda, kv, d, heads, q_op):
    """shared prefix (per-row block + count, with a padding mask) + private step-major suffix;
    q in fp32 or (q_op) in the K/V dtype as the QKV GEMM writes it."""
    L = _lib()
    g = torch.Generator().manual_seed(d + heads)
    dt = _TD[kv]
    nblk, len1, M, nsteps = 3, 9, 10, 5
    q = (torch.randn(M, d, generator=g) * 0.3).to(cuda)
    if q_op:
        q = q.to(dt)
    k1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    v1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    blk = torch.randint(0, nblk, (M,), generator=g, dtype=torch.int32).to(cuda)
    cnt = torch.randint(1, len1 + 1, (M,), generator=g, dtype=torch.int32).to(cuda)
    mask = torch.zeros(nblk, len1, dtype=torch.uint8)
    mask[1, 3] = 1
    mask[2, 0] = 1
    mask = mask.to(cuda)
    live = [M, M - 2, M - 2, M - 5, M - 6]
    step_off = torch.tensor(np.concatenate([[0], np.cumsum(live)]).astype(np.int64)).to(cuda)
    k2 = torch.randn(int(sum(live)), d, generator=g).to(dt).to(cuda)
    v2 = torch.randn(int(sum(live)), d, generator=g).to(dt).to(cuda)
    Mlive = live[-1]          # rows alive at the last step attend to all 5 private keys
    out = torch.empty(Mlive, d, device=cuda, dtype=dt)
    a = L.AttnArgs()
    a.q, a.M, a.d, a.heads, a.q_op_dtype = q.data_ptr(), Mlive, d, heads, q_op
    a.kv_dtype = _code(L, kv)
    a.k1, a.v1, a.len1 = k1.data_ptr(), v1.data_ptr(), len1
    a.seg1_blk, a.seg1_cnt, a.mask1 = blk.data_ptr(), cnt.data_ptr(), mask.data_ptr()
    a.k2, a.v2, a.step_off, a.n_steps2 = k2.data_ptr(), v2.data_ptr(), step_off.data_ptr(), nsteps
    a.out = out.data_ptr()
    L.check(L.lib.nmtgan_attn_rows(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    so = step_off.cpu().numpy()
    for r in range(Mlive):
        b, c = int(blk[r]), int(cnt[r])
        keep = [j for j in range(c) if not int(mask[b, j])]
        ks = [k1[b, j] for j in keep] + [k2[so[s] + r] for s in range(nsteps)]
        vs = [v1[b, j] for j in keep] + [v2[so[s] + r] for s in range(nsteps)]
        ref = _attn_ref(q[r].double().cpu(), torch.stack(ks).double().cpu(), torch.stack(vs).double().cpu(), heads)
        err = (out[r].double().cpu() - ref).abs().max().item()
        assert err < {"bf16": 2e-2, "fp16": 3e-3, "fp32": 2e-5}[kv], (r, err)   # 16-bit: output rounding only


@pytest.mark.parametrize("kv,q_op", [("fp16", 1), ("fp16", 0), ("bf16", 1), ("fp32", 0)])
@pytest.mark.parametrize("d,heads", [(512, 8), (1024, 16)])
def test_attn_rows_mask_free_kernels(cuda, kv, q_op, d, heads):
    """Calls without a padding mask take kernels of their own at d = 512 / 1024 (the rollout steps' self-attention), checked
    against a float64 softmax for every mix of prefix and suffix lengths around the batch size of four keys, and against
    the general kernels, which an all-zero mask selects.  The two forms round differently in the last place - the 16-bit
    kernel batches each segment on its own, so a prefix length that is no multiple of four moves the rescaling points of
    the online softmax, and the branch-free fp32 kernel lets the compiler fuse l * alpha + w where the general kernel's
    branches keep the two apart - so they are compared at one rounding step of the stored value; the fp32 kernel also as
    the split-precision operand image (hi + lo)."""
    L = _lib()
    g = torch.Generator().manual_seed(d + heads + q_op)
    dt = _TD[kv]
    nblk, len1, M, nsteps = 5, 11, 37, 6
    q = (torch.randn(M, d, generator=g) * 0.3).to(cuda)
    if q_op:
        q = q.to(dt)
    k1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    v1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    blk = torch.randint(0, nblk, (M,), generator=g, dtype=torch.int32).to(cuda)
    cnt = torch.randint(1, len1 + 1, (M,), generator=g, dtype=torch.int32).to(cuda)
    zero_mask = torch.zeros(nblk, len1, dtype=torch.uint8, device=cuda)
    live = [M, M - 1, M - 1, M - 4, M - 9, M - 9]
    step_off = torch.tensor(np.concatenate([[0], np.cumsum(live)]).astype(np.int64)).to(cuda)
    k2 = torch.randn(int(sum(live)), d, generator=g).to(dt).to(cuda)
    v2 = torch.randn(int(sum(live)), d, generator=g).to(dt).to(cuda)
    so = step_off.cpu().numpy()

    def run(ns, mask, split):
        rows = live[ns - 1] if ns > 0 else M
        out = torch.full((rows, 3 * d if split else d), float("nan"), device=cuda, dtype=torch.float16 if split else dt)
        a = L.AttnArgs()
        a.q, a.M, a.d, a.heads, a.q_op_dtype = q.data_ptr(), rows, d, heads, q_op
        a.kv_dtype = _code(L, kv)
        a.k1, a.v1, a.len1 = k1.data_ptr(), v1.data_ptr(), len1
        a.seg1_blk, a.seg1_cnt = blk.data_ptr(), cnt.data_ptr()
        a.mask1 = None if mask is None else mask.data_ptr()
        if ns > 0:
            a.k2, a.v2, a.step_off, a.n_steps2 = k2.data_ptr(), v2.data_ptr(), step_off.data_ptr(), ns
        a.out, a.out_split = out.data_ptr(), int(split)
        L.check(L.lib.nmtgan_attn_rows(C.byref(a), L.stream_ptr()))
        torch.cuda.synchronize()
        return out

    for ns in range(0, nsteps + 1):
        lean, general = run(ns, None, False), run(ns, zero_mask, False)
        gap = (lean.double() - general.double()).abs().max().item()
        assert gap <= {"bf16": 2.0 ** -6, "fp16": 2.0 ** -9, "fp32": 2e-6}[kv], (ns, gap)   # one rounding step of the stored value
        if kv == "fp32":
            img_l, img_g = run(ns, None, True).double(), run(ns, zero_mask, True).double()
            assert torch.equal(img_l[:, :d], img_l[:, 2 * d:]) and torch.equal(img_g[:, :d], img_g[:, 2 * d:])
            assert ((img_l[:, :d] + img_l[:, d:2 * d]) - lean.double()).abs().max().item() <= 2.0 ** -20
            assert ((img_l[:, :d] + img_l[:, d:2 * d]) - (img_g[:, :d] + img_g[:, d:2 * d])).abs().max().item() <= 2e-6
        for r in range(0, lean.shape[0], 5):
            b, c = int(blk[r]), int(cnt[r])
            ks = [k1[b, j] for j in range(c)] + [k2[so[t] + r] for t in range(ns)]
            vs = [v1[b, j] for j in range(c)] + [v2[so[t] + r] for t in range(ns)]
            ref = _attn_ref(q[r].double().cpu(), torch.stack(ks).double().cpu(), torch.stack(vs).double().cpu(), heads)
            err = (lean[r].double().cpu() - ref).abs().max().item()
            assert err < {"bf16": 2e-2, "fp16": 3e-3, "fp32": 2e-5}[kv], (ns, r, err)


def test_attn_rows_causal(cuda):
    L = _lib()
    g = torch.Generator().manual_seed(5)
    B, T, d, heads = 3, 7, 128, 4
    q = (torch.randn(B * T, d, generator=g) * 0.3).to(cuda)
    k = torch.randn(B, T, d, generator=g).to(cuda)
    v = torch.randn(B, T, d, generator=g).to(cuda)
    out = torch.empty(B * T, d, device=cuda)
    a = L.AttnArgs()
    a.q, a.M, a.d, a.heads, a.kv_dtype = q.data_ptr(), B * T, d, heads, L.NMTGAN_F32
    a.k1, a.v1, a.len1, a.rows_per_blk, a.causal = k.data_ptr(), v.data_ptr(), T, T, 1
    a.out = out.data_ptr()
    L.check(L.lib.nmtgan_attn_rows(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    for r in range(B * T):
        b, i = divmod(r, T)
        ref = _attn_ref(q[r].double().cpu(), k[b, :i + 1].double().cpu(), v[b, :i + 1].double().cpu(), heads)
        assert (out[r].double().cpu() - ref).abs().max().item() < 2e-5


def _sample(L, logits, u, row_map=None, u_off=None, u_add=0, greedy=0, M=None):
    M = logits.shape[0] if M is None else M
    out = torch.full((M,), -1, dtype=torch.int32, device=logits.device)
    a = L.SampleArgs()
    a.logits, a.ld, a.V, a.M = logits.data_ptr(), logits.stride(0), logits.shape[1], M
    a.row_map = None if row_map is None else row_map.data_ptr()
    a.u = None if u is None else u.data_ptr()
    a.u_off = None if u_off is None else u_off.data_ptr()
    a.u_add, a.greedy, a.out = u_add, greedy, out.data_ptr()
    L.check(L.lib.nmtgan_sample_rows(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("V", [512, 32000, 34432])
def test_sampler_matches_cdf_oracle(cuda, V):
    """Known-answer test of the sampler against the fp64 CDF oracle, including peaked
    rows, a row with one dominant logit, -inf entries and u at 0 / just below 1."""
    from oracle import nmtgan_oracle as O
    L = _lib()
    rng = np.random.default_rng(V)
    M = 48
    lg = (rng.standard_normal((M, V)) * rng.choice([0.5, 2.0, 8.0], size=(M, 1))).astype(np.float32)
    lg[3, 17] = 60.0
    lg[4, :] = -3.0
    lg[5, : V // 2] = -np.inf
    u = rng.random(M, dtype=np.float32)
    u[0], u[1], u[2] = 0.0, np.float32(1.0 - 2 ** -24), 0.5
    got = _sample(L, torch.from_numpy(lg).to(cuda), torch.from_numpy(u).to(cuda))
    want = np.array([O.multinomial_cdf(lg[r], u[r]) for r in range(M)])
    bad = np.nonzero(got != want)[0]
    # any mismatch must be a stated near-tie: u*total within 1e-6 (relative) of a CDF boundary
    for r in bad:
        l = lg[r].astype(np.float64)
        c = np.cumsum(np.exp(l - l.max()))
        t = np.float64(u[r]) * c[-1]
        assert abs(int(got[r]) - int(want[r])) <= 1 or np.isinf(lg[r, min(got[r], want[r]) + 1:max(got[r], want[r])]).all()
        assert np.abs(c[[got[r], want[r]]] - t).min() < 1e-6 * c[-1], (r, got[r], want[r])
    assert len(bad) <= 1, bad
    assert got[3] == 17 or u[3] < 1e-20


def test_sampler_indirection_and_greedy(cuda):
    from oracle import nmtgan_oracle as O
    L = _lib()
    rng = np.random.default_rng(9)
    V, Mrows, M = 1024, 6, 20
    lg = rng.standard_normal((Mrows, V)).astype(np.float32) * 3
    ubig = rng.random(500, dtype=np.float32)
    row_map = rng.integers(0, Mrows, M).astype(np.int32)
    u_off = rng.integers(0, 400, M).astype(np.int64)
    got = _sample(L, torch.from_numpy(lg).to(cuda), torch.from_numpy(ubig).to(cuda),
                  row_map=torch.from_numpy(row_map).to(cuda), u_off=torch.from_numpy(u_off).to(cuda), u_add=7, M=M)
    want = np.array([O.multinomial_cdf(lg[row_map[r]], ubig[u_off[r] + 7]) for r in range(M)])
    assert np.array_equal(got, want)
    lg[2, 100] = lg[2, 900] = 50.0           # tie: first index wins (np.argmax rule)
    got = _sample(L, torch.from_numpy(lg).to(cuda), None, greedy=1)
    assert np.array_equal(got, lg.argmax(axis=1))


@pytest.mark.parametrize("op,M,V,d", [("bf16", 300, 32000, 512), ("bf16", 70, 4096, 256), ("fp32", 129, 2048, 128),
                                      ("bf16", 1, 32000, 512), ("fp16", 300, 32000, 512), ("fp16", 33, 4096, 256)])
def test_logits_sample_fused(cuda, op, M, V, d):
    """Fused logits + sampling (no logits in HBM) against the fp64 CDF oracle applied to the exact
    logits x . w^T of the same (op-dtype-representable) operands.  A mismatch is accepted only as a
    near-tie: u*total within 1e-5 (relative) of the CDF boundary between the two answers."""
    from oracle import nmtgan_oracle as O
    L = _lib()
    rng = np.random.default_rng(M + V)
    dt, opc = L.op_dtype_of(op)
    x = torch.from_numpy(rng.standard_normal((M, d)).astype(np.float32)).to(dt)
    w = torch.from_numpy((rng.standard_normal((V, d)) * rng.choice([0.02, 0.1, 0.4], size=(1, 1))).astype(np.float32)).to(dt)
    if M > 3:
        w[11] = x[2] * 0.5                                  # one dominant token for row 2
    ubig = rng.random(M + 50, dtype=np.float32)
    u_off = rng.permutation(M).astype(np.int64)
    ubig[u_off[0] + 7] = 0.0
    if M > 1:
        ubig[u_off[1] + 7] = np.float32(1.0 - 2 ** -24)
    xg, wg, ug, og = x.to(cuda), w.to(cuda), torch.from_numpy(ubig).to(cuda), torch.from_numpy(u_off).to(cuda)
    out = torch.full((M,), -1, dtype=torch.int32, device=cuda)
    nb = L.lib.nmtgan_logits_sample_workspace_bytes(M, V)
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    L.check(L.lib.nmtgan_logits_sample(opc, xg.data_ptr(), wg.data_ptr(), M, V, d, ug.data_ptr(), og.data_ptr(), 7,
                                       out.data_ptr(), ws.data_ptr(), nb, L.stream_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    lg = (x.double() @ w.double().T).numpy()
    want = np.array([O.multinomial_cdf(lg[r].astype(np.float32), ubig[u_off[r] + 7]) for r in range(M)])
    bad = np.nonzero(got != want)[0]
    for r in bad:
        l = lg[r]
        c = np.cumsum(np.exp(l - l.max()))
        t = np.float64(ubig[u_off[r] + 7]) * c[-1]
        lo, hi = sorted((int(got[r]), int(want[r])))
        assert np.abs(c[lo:hi] - t).min() < 1e-5 * c[-1], (r, got[r], want[r])
    print("fused sampler mismatches (near ties): %d / %d" % (len(bad), M))
    assert len(bad) <= max(1, M // 50), bad
    assert (got >= 0).all() and (got < V).all()
    # the unsupported shapes are refused, not silently mis-sampled
    assert L.lib.nmtgan_logits_sample(opc, xg.data_ptr(), wg.data_ptr(), M, V - 1, d, ug.data_ptr(), og.data_ptr(), 7,
                                      out.data_ptr(), ws.data_ptr(), nb, L.stream_ptr()) != 0


def test_reward_reduce(cuda):
    L = _lib()
    rng = np.random.default_rng(2)
    B, T, R = 3, 6, 4
    mask = np.array([[1, 1, 1, 1, 0, 0], [1] * 6, [1, 0, 0, 0, 0, 0]], np.float32)
    from nmt_gan_amd.share_function import build_rollout_plan
    plan = build_rollout_plan(mask, R)
    p = rng.random(plan.n_rows).astype(np.float32)
    pb = rng.random(B).astype(np.float32)
    out = torch.empty(B, T, device=cuda)
    p_d, pb_d = torch.from_numpy(p).to(cuda), torch.from_numpy(pb).to(cuda)
    gf_d, mask_d = torch.from_numpy(plan.grp_first).to(cuda), torch.from_numpy(mask).to(cuda)
    for use_bias, bias in [(0, 0.0), (1, 0.6)]:
        L.check(L.lib.nmtgan_reward_reduce(p_d.data_ptr(), pb_d.data_ptr(), gf_d.data_ptr(), mask_d.data_ptr(),
                                           B, T, R, use_bias, bias, out.data_ptr(), L.stream_ptr()))
        torch.cuda.synchronize()
        want = np.zeros((B, T), np.float32)
        for b in range(B):
            for t in range(T):
                q = np.float32(0)
                first = plan.grp_first[b, t] if t < T - 1 else -1
                for r in range(R):
                    q = np.float32(q + (p[first + r] if first >= 0 else pb[b]))
                if use_bias:
                    q = np.float32(q - np.float32(bias * R))
                want[b, t] = np.float32(np.float32(q * mask[b, t]) / np.float32(R))
        assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("kv", ["bf16", "fp16"])
@pytest.mark.parametrize("q_bf16", [0, 1])
@pytest.mark.parametrize("len1,Lp", [(50, 64), (80, 96)])
@pytest.mark.parametrize("d,heads,R", [(128, 2, 16), (512, 8, 16), (1024, 16, 20), (128, 4, 3), (256, 8, 32), (256, 2, 8)])
def test_attn_group_final_and_partial(cuda, d, heads, R, len1, Lp, q_bf16, kv):
    """MFMA group attention (shared keys, bf16) in both output modes: normalised output (cross
    attention with a padding mask) and running state continued by attn_rows over private keys."""
    L = _lib()
    g = torch.Generator().manual_seed(d + R)
    nblk = 4
    G = 7                                   # groups
    M = G * R
    q = (torch.randn(M, d, generator=g) * 0.3).to(cuda)
    dt = _TD[kv]
    if q_bf16:
        q = q.to(dt)
    k1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    v1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    vt = torch.full((nblk, d, Lp), float("nan"), dtype=dt, device=cuda)
    L.check(L.lib.nmtgan_transpose_v(v1.data_ptr(), vt.data_ptr(), nblk, len1, Lp, d, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(vt[:, :, :len1], v1.transpose(1, 2)) and vt[:, :, len1:].abs().max().item() == 0
    gblk = torch.randint(0, nblk, (G,), generator=g, dtype=torch.int32)
    gcnt = torch.tensor([1, 2, 31, 32, 33, len1, 65 if len1 > 65 else 17], dtype=torch.int32)
    blk = gblk.repeat_interleave(R).to(cuda)
    cnt = gcnt.repeat_interleave(R).to(cuda)
    mask = torch.zeros(nblk, len1, dtype=torch.uint8)
    mask[:, len1 - 10:] = 1
    mask[1, 0] = 1
    mask_d = mask.to(cuda)

    def ref_row(r, use_mask, extra_k=None, extra_v=None):
        b, c = int(blk[r]), int(cnt[r])
        keep = [j for j in range(c) if not (use_mask and int(mask[b, j]))]
        ks, vs = [k1[b, j] for j in keep], [v1[b, j] for j in keep]
        if extra_k is not None:
            ks += extra_k
            vs += extra_v
        return _attn_ref(q[r].double().cpu(), torch.stack(ks).double().cpu(), torch.stack(vs).double().cpu(), heads)

    # --- final mode with mask (counts = full length, like cross attention)
    out = torch.full((M, d), float("nan"), dtype=dt, device=cuda)
    tol = (3e-2 if d // heads <= 64 else 5e-2) if kv == "bf16" else 6e-3    # q / p / output rounding of the operand type
    a = L.AttnGroupArgs()
    a.kv_dtype = _code(L, kv)
    a.q, a.M, a.d, a.heads, a.rows_per_group, a.q_bf16 = q.data_ptr(), M, d, heads, R, q_bf16
    a.k, a.vt, a.len, a.len_padded = k1.data_ptr(), vt.data_ptr(), len1, Lp
    a.row_blk, a.row_cnt, a.mask, a.out = blk.data_ptr(), cnt.data_ptr(), mask_d.data_ptr(), out.data_ptr()
    rows_ok = [r for r in range(M) if any(not int(mask[int(blk[r]), j]) for j in range(int(cnt[r])))]
    L.check(L.lib.nmtgan_attn_group(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    for r in rows_ok[::3]:
        err = (out[r].double().cpu() - ref_row(r, True)).abs().max().item()
        assert err < tol, (r, err)
    # --- partial mode, continued over 3 private keys by attn_rows
    o_part = torch.empty(M, d, device=cuda)
    m_part = torch.empty(M, heads, device=cuda)
    l_part = torch.empty(M, heads, device=cuda)
    a.mask, a.out = None, None
    a.o_part, a.m_part, a.l_part = o_part.data_ptr(), m_part.data_ptr(), l_part.data_ptr()
    L.check(L.lib.nmtgan_attn_group(C.byref(a), L.stream_ptr()))
    nsteps = 3
    step_off = torch.tensor([0, M, 2 * M, 3 * M], dtype=torch.int64, device=cuda)
    k2 = torch.randn(nsteps * M, d, generator=g).to(dt).to(cuda)
    v2 = torch.randn(nsteps * M, d, generator=g).to(dt).to(cuda)
    out2 = torch.empty(M, d, dtype=dt, device=cuda)
    b = L.AttnArgs()
    b.q, b.M, b.d, b.heads, b.kv_dtype, b.q_op_dtype = q.data_ptr(), M, d, heads, _code(L, kv), q_bf16
    b.k2, b.v2, b.step_off, b.n_steps2 = k2.data_ptr(), v2.data_ptr(), step_off.data_ptr(), nsteps
    b.o_init, b.m_init, b.l_init, b.out = o_part.data_ptr(), m_part.data_ptr(), l_part.data_ptr(), out2.data_ptr()
    L.check(L.lib.nmtgan_attn_rows(C.byref(b), L.stream_ptr()))
    torch.cuda.synchronize()
    for r in range(0, M, 5):
        ek = [k2[s * M + r] for s in range(nsteps)]
        ev = [v2[s * M + r] for s in range(nsteps)]
        err = (out2[r].double().cpu() - ref_row(r, False, ek, ev)).abs().max().item()
        assert err < tol, (r, err)


@pytest.mark.parametrize("kv", ["bf16", "fp16"])
@pytest.mark.parametrize("d,heads,R", [(512, 8, 16), (256, 4, 4), (128, 2, 16)])
def test_attn_group_slot_table(cuda, d, heads, R, kv):
    """Slot form: 32 // R groups of one sentence per workgroup (per-slot key counts, dead slots, a
    partially filled entry) must give exactly what the one-group-per-workgroup form gives."""
    L = _lib()
    g = torch.Generator().manual_seed(d * R)
    nblk, len1, Lp = 3, 50, 64
    slots = 32 // R
    per_blk = 2 * slots + 1                       # groups per sentence: two full entries + one partial
    G = nblk * per_blk
    M = G * R
    dt = _TD[kv]
    q = (torch.randn(M, d, generator=g) * 0.3).to(dt).to(cuda)
    k1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    v1 = torch.randn(nblk, len1, d, generator=g).to(dt).to(cuda)
    vt = torch.zeros((nblk, d, Lp), dtype=dt, device=cuda)
    L.check(L.lib.nmtgan_transpose_v(v1.data_ptr(), vt.data_ptr(), nblk, len1, Lp, d, L.stream_ptr()))
    gblk = torch.arange(nblk, dtype=torch.int32).repeat_interleave(per_blk)
    gcnt = torch.randint(1, len1 + 1, (G,), generator=g, dtype=torch.int32)
    gcnt[0], gcnt[1] = 3, 40                      # one entry straddling the 32-key tile boundary
    blk = gblk.repeat_interleave(R).to(cuda)
    cnt = gcnt.repeat_interleave(R).to(cuda)
    mask = torch.zeros(nblk, len1, dtype=torch.uint8)
    mask[:, 45:] = 1
    mask[2, 0] = 1
    mask_d = mask.to(cuda)
    ents = []
    for b in range(nblk):
        gs = [b * per_blk + i for i in range(per_blk)]
        for i in range(0, per_blk, slots):
            part = gs[i:i + slots]
            ents.append([b, 0] + [x * R for x in part] + [-1] * (slots - len(part))
                        + [int(gcnt[x]) for x in part] + [0] * (slots - len(part)))
    table = torch.tensor(ents, dtype=torch.int32).to(cuda)
    M_live = M - 2 * R                            # the last two groups are dead (row0 >= M)

    def run(use_table, final):
        a = L.AttnGroupArgs()
        a.q, a.M, a.d, a.heads, a.rows_per_group, a.q_bf16 = q.data_ptr(), M_live, d, heads, R, 1
        a.kv_dtype = _code(L, kv)
        a.k, a.vt, a.len, a.len_padded = k1.data_ptr(), vt.data_ptr(), len1, Lp
        a.row_blk = blk.data_ptr()
        out = torch.zeros((M, d), dtype=dt, device=cuda)
        o_part = torch.zeros(M, d, device=cuda)
        m_part = torch.zeros(M, heads, device=cuda)
        l_part = torch.zeros(M, heads, device=cuda)
        if final:                                 # cross attention: every key, padding mask
            a.mask, a.out = mask_d.data_ptr(), out.data_ptr()
        else:
            a.row_cnt = cnt.data_ptr()
            a.o_part, a.m_part, a.l_part = o_part.data_ptr(), m_part.data_ptr(), l_part.data_ptr()
        if use_table:
            a.slot_table, a.slots, a.n_entries = table.data_ptr(), slots, table.shape[0]
        L.check(L.lib.nmtgan_attn_group(C.byref(a), L.stream_ptr()))
        torch.cuda.synchronize()
        return out, o_part, m_part, l_part

    for final in (True, False):
        want = run(False, final)
        got = run(True, final)
        for w_, g_ in zip(want, got):
            assert torch.equal(w_[:M_live], g_[:M_live])
            assert g_[M_live:].abs().max().item() == 0          # dead slots are not written


@pytest.mark.parametrize("len1,Lp", [(50, 64), (80, 96)])
@pytest.mark.parametrize("d,heads,R", [(128, 4, 16), (512, 8, 16), (128, 2, 3), (256, 8, 32), (256, 4, 8)])
def test_attn_group_f32(cuda, d, heads, R, len1, Lp):
    """fp32 form of the group attention (fp32 K / V^T / q, fp32 MFMA, fp32 output): normalised output
    under a padding mask, the running state continued by attn_rows over private fp32 keys, and the slot
    table giving exactly the one-group-per-workgroup result.  Tolerance 2e-5: fp32 products and sums."""
    L = _lib()
    g = torch.Generator().manual_seed(3 * d + R)
    nblk = 3
    slots = 32 // R
    per_blk = 2 * slots + 1
    G = nblk * per_blk
    M = G * R
    q = (torch.randn(M, d, generator=g) * 0.3).to(cuda)
    k1 = torch.randn(nblk, len1, d, generator=g).to(cuda)
    v1 = torch.randn(nblk, len1, d, generator=g).to(cuda)
    vt = torch.full((nblk, d, Lp), float("nan"), device=cuda)
    L.check(L.lib.nmtgan_transpose_v_f32(v1.data_ptr(), vt.data_ptr(), nblk, len1, Lp, d, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(vt[:, :, :len1], v1.transpose(1, 2)) and vt[:, :, len1:].abs().max().item() == 0
    gblk = torch.arange(nblk, dtype=torch.int32).repeat_interleave(per_blk)
    gcnt = torch.randint(1, len1 + 1, (G,), generator=g, dtype=torch.int32)
    gcnt[0], gcnt[1], gcnt[2] = 1, 33, len1
    blk = gblk.repeat_interleave(R).to(cuda)
    cnt = gcnt.repeat_interleave(R).to(cuda)
    mask = torch.zeros(nblk, len1, dtype=torch.uint8)
    mask[:, len1 - 10:] = 1
    mask[1, 0] = 1
    mask_d = mask.to(cuda)
    ents = []
    for b in range(nblk):
        gs = [b * per_blk + i for i in range(per_blk)]
        for i in range(0, per_blk, slots):
            part = gs[i:i + slots]
            ents.append([b, 0] + [x * R for x in part] + [-1] * (slots - len(part))
                        + [int(gcnt[x]) for x in part] + [0] * (slots - len(part)))
    table = torch.tensor(ents, dtype=torch.int32).to(cuda)

    def run(use_table, final):
        a = L.AttnGroupArgs()
        a.q, a.M, a.d, a.heads, a.rows_per_group, a.kv_dtype = q.data_ptr(), M, d, heads, R, L.NMTGAN_F32
        a.k, a.vt, a.len, a.len_padded = k1.data_ptr(), vt.data_ptr(), len1, Lp
        a.row_blk = blk.data_ptr()
        out = torch.zeros(M, d, device=cuda)
        o_part, m_part, l_part = torch.zeros(M, d, device=cuda), torch.zeros(M, heads, device=cuda), torch.zeros(M, heads, device=cuda)
        if final:
            a.mask, a.out = mask_d.data_ptr(), out.data_ptr()
        else:
            a.row_cnt = cnt.data_ptr()
            a.o_part, a.m_part, a.l_part = o_part.data_ptr(), m_part.data_ptr(), l_part.data_ptr()
        if use_table:
            a.slot_table, a.slots, a.n_entries = table.data_ptr(), slots, table.shape[0]
        L.check(L.lib.nmtgan_attn_group(C.byref(a), L.stream_ptr()))
        torch.cuda.synchronize()
        return out, o_part, m_part, l_part

    def ref_row(r, final, extra_k=None, extra_v=None):
        b = int(blk[r])
        keep = [j for j in range(len1) if not int(mask[b, j])] if final else list(range(int(cnt[r])))
        ks, vs = [k1[b, j] for j in keep], [v1[b, j] for j in keep]
        if extra_k is not None:
            ks += extra_k
            vs += extra_v
        return _attn_ref(q[r].double().cpu(), torch.stack(ks).double().cpu(), torch.stack(vs).double().cpu(), heads)

    out = run(False, True)[0]
    for r in range(0, M, 3):
        err = (out[r].double().cpu() - ref_row(r, True)).abs().max().item()
        assert err < 2e-5, (r, err)
    part = run(False, False)
    nsteps = 3
    step_off = torch.tensor([0, M, 2 * M, 3 * M], dtype=torch.int64, device=cuda)
    k2 = torch.randn(nsteps * M, d, generator=g).to(cuda)
    v2 = torch.randn(nsteps * M, d, generator=g).to(cuda)
    out2 = torch.empty(M, d, device=cuda)
    b = L.AttnArgs()
    b.q, b.M, b.d, b.heads, b.kv_dtype = q.data_ptr(), M, d, heads, L.NMTGAN_F32
    b.k2, b.v2, b.step_off, b.n_steps2 = k2.data_ptr(), v2.data_ptr(), step_off.data_ptr(), nsteps
    b.o_init, b.m_init, b.l_init, b.out = part[1].data_ptr(), part[2].data_ptr(), part[3].data_ptr(), out2.data_ptr()
    L.check(L.lib.nmtgan_attn_rows(C.byref(b), L.stream_ptr()))
    torch.cuda.synchronize()
    for r in range(0, M, 5):
        ek = [k2[s * M + r] for s in range(nsteps)]
        ev = [v2[s * M + r] for s in range(nsteps)]
        err = (out2[r].double().cpu() - ref_row(r, False, ek, ev)).abs().max().item()
        assert err < 2e-5, (r, err)
    for final in (True, False):
        want, got = run(False, final), run(True, final)
        for w_, g_ in zip(want, got):
            assert torch.equal(w_, g_)
    # head dims the fp32 kernel does not serve are refused
    a = L.AttnGroupArgs()
    a.q, a.M, a.d, a.heads, a.rows_per_group, a.kv_dtype = q.data_ptr(), M, d, 1, R, L.NMTGAN_F32
    a.k, a.vt, a.len, a.len_padded, a.row_blk, a.out = k1.data_ptr(), vt.data_ptr(), len1, Lp, blk.data_ptr(), out.data_ptr()
    assert L.lib.nmtgan_attn_group(C.byref(a), L.stream_ptr()) != 0


def _split3_ref(x):
    """[hi | lo | hi] fp16 image of an fp32 matrix, the host formula of nmtgan_split3."""
    hi = x.half()
    lo = (x - hi.float()).half()
    return torch.cat([hi, lo, hi], dim=1)


@pytest.mark.parametrize("M,N,K,relu,bias", [(300, 512, 512, 0, 1), (4097, 2048, 512, 1, 1), (70000, 2048, 1536, 1, 1),
                                             (16, 2048, 1536, 1, 0), (2600, 512, 6144, 0, 1)])
def test_gemm_split_image_output(cuda, M, N, K, relu, bias):
    """out_f32 = 2: the GEMM writes its result as the [hi | lo | hi] operand image of the next split-precision GEMM.
    Bit-identical to nmtgan_split3 of the same GEMM's fp32 output, on every kernel the shapes dispatch to (the
    double-buffered 128x128, the persistent 256x128 and 256x256 kernels)."""
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).half().to(cuda)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(cuda)
    b = torch.randn(N, generator=g).to(cuda) if bias else None

    def run(mode):
        a = L.GemmArgs()
        a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = L.NMTGAN_F16, L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, K, K
        a.bias, a.relu, a.n_seg, a.seg_width = (b.data_ptr() if bias else None), relu, 1, N
        out = torch.full((M, 3 * N), float("nan"), device=cuda, dtype=torch.float16) if mode == 2 else torch.empty(M, N, device=cuda)
        a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = out.data_ptr(), (3 * N if mode == 2 else N), mode, 1.0
        L.check(L.lib.nmtgan_gemm(C.byref(a), L.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        return out

    f32 = run(1)
    img = run(2)
    assert torch.equal(img, _split3_ref(f32))
    # two segments: fp32 and an image
    if N != 512:
        return
    a = L.GemmArgs()
    w = N // 2
    a.op_dtype, a.epilogue, a.A, a.Wt, a.M, a.N, a.K, a.lda = L.NMTGAN_F16, L.EPI_LINEAR, A.data_ptr(), W.data_ptr(), M, N, K, K
    a.bias, a.relu, a.n_seg, a.seg_width = (b.data_ptr() if bias else None), relu, 2, w
    o0 = torch.empty(M, w, device=cuda)
    o1 = torch.full((M, 3 * w), float("nan"), device=cuda, dtype=torch.float16)
    a.out[0], a.ldo[0], a.out_f32[0], a.seg_scale[0] = o0.data_ptr(), w, 1, 1.0
    a.out[1], a.ldo[1], a.out_f32[1], a.seg_scale[1] = o1.data_ptr(), 3 * w, 2, 1.0
    L.check(L.lib.nmtgan_gemm(C.byref(a), L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(o0, f32[:, :w]) and torch.equal(o1, _split3_ref(f32[:, w:].contiguous()))


@pytest.mark.parametrize("d,heads", [(64, 4), (256, 4), (512, 8), (1024, 16)])
def test_split_image_producers(cuda, d, heads):
    """embed, add_layernorm, attn_rows (fp32) and attn_group (fp32) writing the split-precision operand image of
    their output: bit-identical to nmtgan_split3 applied to the fp32 output of the same call."""
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(d)
    M, V, P = 203, 97, 60
    # embed
    table, timing = torch.randn(V, d, generator=g).to(cuda), torch.randn(P, d, generator=g).to(cuda)
    tok = torch.randint(0, V, (M,), generator=g, dtype=torch.int32).to(cuda)
    e = L.EmbedArgs()
    e.table, e.timing, e.vocab, e.max_pos, e.tok, e.tok_ld, e.pos_mod, e.mult = table.data_ptr(), timing.data_ptr(), V, P, tok.data_ptr(), 1, P, 3.0
    xf = torch.empty(M, d, device=cuda)
    img = torch.full((M, 3 * d), float("nan"), device=cuda, dtype=torch.float16)
    e.M, e.d, e.out_f32, e.op_dtype, e.out_split = M, d, xf.data_ptr(), L.NMTGAN_F32, img.data_ptr()
    L.check(L.lib.nmtgan_embed(C.byref(e), L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(img, _split3_ref(xf))
    # add_layernorm, in place on x with the op-dtype copy aliased to it (the fp32 tiers' layout)
    y = torch.randn(M, d, generator=g).to(cuda)
    gam, bet = torch.randn(d, generator=g).to(cuda), torch.randn(d, generator=g).to(cuda)
    a = L.AddLnArgs()
    a.x, a.y, a.gamma, a.beta, a.M, a.d = xf.data_ptr(), y.data_ptr(), gam.data_ptr(), bet.data_ptr(), M, d
    a.out_f32, a.out_op, a.op_dtype, a.out_split = xf.data_ptr(), xf.data_ptr(), L.NMTGAN_F32, img.data_ptr()
    want = torch.nn.functional.layer_norm((xf + y).double(), (d,), gam.double(), bet.double(), 1e-6)
    L.check(L.lib.nmtgan_add_layernorm(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    assert (xf.double() - want).abs().max().item() < 1e-4
    assert torch.equal(img, _split3_ref(xf))
    # attn_rows, fp32 keys: private segment of 5 steps
    nsteps = 5
    q = (torch.randn(M, d, generator=g) * 0.3).to(cuda)
    k2, v2 = torch.randn(nsteps * M, d, generator=g).to(cuda), torch.randn(nsteps * M, d, generator=g).to(cuda)
    step_off = (torch.arange(nsteps + 1, dtype=torch.int64) * M).to(cuda)
    outs = []
    for split in (0, 1):
        o = torch.full((M, 3 * d), float("nan"), device=cuda, dtype=torch.float16) if split else torch.empty(M, d, device=cuda)
        b = L.AttnArgs()
        b.q, b.M, b.d, b.heads, b.kv_dtype, b.out_split = q.data_ptr(), M, d, heads, L.NMTGAN_F32, split
        b.k2, b.v2, b.step_off, b.n_steps2, b.out = k2.data_ptr(), v2.data_ptr(), step_off.data_ptr(), nsteps, o.data_ptr()
        L.check(L.lib.nmtgan_attn_rows(C.byref(b), L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[1], _split3_ref(outs[0]))
    b.kv_dtype, b.out_split = L.NMTGAN_F16, 1                  # only the fp32 form has the image output
    assert L.lib.nmtgan_attn_rows(C.byref(b), L.stream_ptr()) != 0
    # attn_group, fp32, normalised output
    if d // heads not in (32, 64):
        return
    R, nblk, len1, Lp = 16, 3, 50, 64
    Mg = 12 * R
    qg = (torch.randn(Mg, d, generator=g) * 0.3).to(cuda)
    k1, v1 = torch.randn(nblk, len1, d, generator=g).to(cuda), torch.randn(nblk, len1, d, generator=g).to(cuda)
    vt = torch.empty(nblk, d, Lp, device=cuda)
    L.check(L.lib.nmtgan_transpose_v_f32(v1.data_ptr(), vt.data_ptr(), nblk, len1, Lp, d, L.stream_ptr()))
    blk = torch.randint(0, nblk, (12,), generator=g, dtype=torch.int32).repeat_interleave(R).to(cuda)
    mask = torch.zeros(nblk, len1, dtype=torch.uint8)
    mask[:, 44:] = 1
    mask_d = mask.to(cuda)
    outs = []
    for split in (0, 1):
        o = torch.full((Mg, 3 * d), float("nan"), device=cuda, dtype=torch.float16) if split else torch.empty(Mg, d, device=cuda)
        ga = L.AttnGroupArgs()
        ga.q, ga.M, ga.d, ga.heads, ga.rows_per_group, ga.kv_dtype = qg.data_ptr(), Mg, d, heads, R, L.NMTGAN_F32
        ga.k, ga.vt, ga.len, ga.len_padded = k1.data_ptr(), vt.data_ptr(), len1, Lp
        ga.row_blk, ga.mask, ga.out, ga.out_split = blk.data_ptr(), mask_d.data_ptr(), o.data_ptr(), split
        L.check(L.lib.nmtgan_attn_group(C.byref(ga), L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[1], _split3_ref(outs[0]))
