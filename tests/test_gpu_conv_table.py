"""Table form of the discriminator's conv + BatchNorm + ReLU + max-over-time (cnn_discriminator.py:88-109, 461-477)
through its C-ABI building blocks: nmtgan_conv_table_build against the fp64 product of the embedding table with every
filter tap, nmtgan_conv_table_prefix / nmtgan_conv_table_rows on given tables against a host restatement that adds the
same table entries in the same order - partial sums bit for bit, features to the last fused multiply-add - and the
whole chain against the reference-form conv + max of the oracle on the same rollouts."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import disc_weights, make_gpu_pair, tiny_cfgs
from oracle import nmtgan_oracle as O

pytestmark = pytest.mark.gpu

_TD = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}


def _lib():
    from nmt_gan_amd import _lib as L
    return L


def _rows(rng, y, lens, R, V):
    """Rollout rows (b, g, r): tokens 0..g-1 of y[b], then own tokens up to len_b, zeros behind (model.py:307)."""
    B, T = y.shape
    ids, rb, rg = [], [], []
    for b in range(B):
        for g in range(1, int(lens[b])):
            for _ in range(R):
                row = y[b].copy()
                row[g:int(lens[b])] = rng.integers(1, V, int(lens[b]) - g)
                ids.append(row)
                rb.append(b)
                rg.append(g)
    return np.asarray(ids, np.int32), np.asarray(rb, np.int32), np.asarray(rg, np.int32)


def _table_struct(L, dev, T, V, Fp, widths, nfs, nfps, dtype_code, tabs, S, scale, shift, wt=None):
    t = L.ConvTable()
    t.n_widths, t.T, t.Fp, t.vocab, t.dtype = len(widths), T, Fp, V, dtype_code
    for i, w in enumerate(widths):
        t.tab[i], t.S[i] = tabs[i].data_ptr(), S[i].data_ptr()
        t.scale[i], t.shift[i] = scale[i].data_ptr(), shift[i].data_ptr()
        t.w[i], t.nf[i], t.nfp[i] = w, nfs[i], nfps[i]
        t.wt[i] = wt[i].data_ptr() if wt is not None else None
    return t


@pytest.mark.parametrize("entries", ["fp32", "fp16", "bf16"])
def test_conv_table_kernels_against_host_restatement(cuda, entries):
    """Given tables (random entries, pad columns poisoned with NaN): the partial sums S are the fp32 roundings of the
    fp64 (fp32 entries) / sequential fp32 (16-bit entries) tap sums - bit for bit -, the window values and the pooled
    features those of the host loop up to the rounding of one multiply-add."""
    L = _lib()
    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(5)
    T, V, B, R = 14, 97, 5, 3
    widths, nfs = [1, 3, 6, 14], [10, 22, 18, 7]
    es = 4 if entries == "fp32" else 2
    nfps = [(n * es + 127) // 128 * 128 // es for n in nfs]
    F = sum(nfs)
    Fp = (F + 63) // 64 * 64
    dt = _TD[entries]
    y = rng.integers(1, V, (B, T)).astype(np.int32)
    lens = np.array([T, 9, 2, T, 5])
    for b in range(B):
        y[b, lens[b]:] = 0
    ids, rb, rg = _rows(rng, y, lens, R, V)
    n_rows = ids.shape[0]
    tabs, S, scale, shift, tabs_h = [], [], [], [], []
    for w, n, npad in zip(widths, nfs, nfps):
        tb = (torch.rand(V, w, npad, generator=g) * 2 - 1).to(dt)
        tb[:, :, n:] = float("nan")                                  # pad columns are never used
        tabs_h.append(tb[:, :, :n].float().numpy())
        tabs.append(tb.to(cuda))
        S.append(torch.full((B * (T - w + 1) * max(w - 1, 1) * npad,), float("nan"), device=cuda))
        scale.append((1 + 0.3 * torch.randn(n, generator=g)).to(cuda))
        shift.append((0.3 * torch.randn(n, generator=g)).to(cuda))
    t = _table_struct(L, cuda, T, V, Fp, widths, nfs, nfps, L.op_dtype_of(entries)[1], tabs, S, scale, shift)
    y_d, ids_d = torch.from_numpy(y).to(cuda), torch.from_numpy(ids).to(cuda)
    rb_d, rg_d = torch.from_numpy(rb).to(cuda), torch.from_numpy(rg).to(cuda)
    win = torch.full((B * T, Fp), -1.0, device=cuda)
    L.check(L.lib.nmtgan_conv_table_prefix(C.byref(t), y_d.data_ptr(), B, win.data_ptr(), L.stream_ptr()), "prefix")
    # host restatement
    acc_t = np.float64 if entries == "fp32" else np.float32
    win_ref = np.full((B * T, Fp), -1.0, np.float32)
    S_ref, col = [], 0
    for i, (w, n, npad) in enumerate(zip(widths, nfs, nfps)):
        Pw = T - w + 1
        s_i = np.zeros((B, Pw, max(w - 1, 1), n), np.float32)
        sc, sh = scale[i].cpu().numpy(), shift[i].cpu().numpy()
        for b in range(B):
            for p in range(Pw):
                a = np.zeros(n, acc_t)
                for j in range(w):
                    a = a + tabs_h[i][y[b, p + j], j].astype(acc_t)
                    if j + 1 < w:
                        s_i[b, p, j] = a.astype(np.float32)
                win_ref[b * T + p, col:col + n] = np.maximum(a.astype(np.float32) * sc + sh, 0)
        S_ref.append(s_i)
        col += n
    torch.cuda.synchronize()
    for i, (w, n, npad) in enumerate(zip(widths, nfs, nfps)):
        if w > 1:
            got = S[i].view(B, T - w + 1, w - 1, npad)[..., :n].cpu().numpy()
            assert np.array_equal(got, S_ref[i]), (entries, w)
    wv = win.cpu().numpy()
    assert np.allclose(wv[:, :F], win_ref[:, :F], rtol=0, atol=2e-6 * max(1.0, np.abs(win_ref[:, :F]).max()))
    assert np.all(wv[:, F:] == -1.0)                                # columns beyond the filters untouched
    # rollout rows: pooled starts from the max over the windows that end inside the prefix
    pooled0 = np.zeros((n_rows, Fp), np.float32)
    col = 0
    for i, (w, n) in enumerate(zip(widths, nfs)):
        for r in range(n_rows):
            for p in range(0, rg[r] - w + 1):
                pooled0[r, col:col + n] = np.maximum(pooled0[r, col:col + n], win_ref[rb[r] * T + p, col:col + n])
        col += n
    pooled = torch.from_numpy(pooled0).to(cuda)
    L.check(L.lib.nmtgan_conv_table_rows(C.byref(t), ids_d.data_ptr(), rb_d.data_ptr(), rg_d.data_ptr(), n_rows,
                                         pooled.data_ptr(), L.stream_ptr()), "rows")
    torch.cuda.synchronize()
    want = pooled0.copy()
    full = np.zeros((n_rows, Fp), np.float32)                       # every window from the row's own tokens, no reuse
    col = 0
    for i, (w, n) in enumerate(zip(widths, nfs)):
        sc, sh = scale[i].cpu().numpy(), shift[i].cpu().numpy()
        for r in range(n_rows):
            b, gq = rb[r], rg[r]
            for p in range(T - w + 1):
                af = np.zeros(n, np.float64)
                for j in range(w):
                    af += tabs_h[i][ids[r, p + j], j]
                full[r, col:col + n] = np.maximum(full[r, col:col + n], np.maximum(af.astype(np.float32) * sc + sh, 0))
                if p < gq - w + 1:
                    continue
                j0 = max(0, gq - p)
                a = S_ref[i][b, p, j0 - 1].astype(acc_t) if j0 > 0 else np.zeros(n, acc_t)
                for j in range(j0, w):
                    a = a + tabs_h[i][ids[r, p + j], j].astype(acc_t)
                want[r, col:col + n] = np.maximum(want[r, col:col + n], a.astype(np.float32) * sc + sh)
        col += n
    got = pooled.cpu().numpy()
    tol = 2e-6 * max(1.0, np.abs(want).max())
    assert np.abs(got[:, :F] - want[:, :F]).max() < tol
    assert np.all(got[:, F:] == 0)
    # ... and the three reuses together are the plain conv + max of the row (fp32 rounding of the sums for 16-bit entries)
    assert np.abs(got[:, :F] - full[:, :F]).max() < (2e-6 if entries == "fp32" else 2e-5) * max(1.0, np.abs(full).max())
    # argument errors
    t.nfp[1] = nfs[1] - 1
    assert L.lib.nmtgan_conv_table_rows(C.byref(t), ids_d.data_ptr(), rb_d.data_ptr(), rg_d.data_ptr(), n_rows,
                                        pooled.data_ptr(), L.stream_ptr()) != 0
    assert b"bad width entry" in L.lib.nmtgan_last_error()


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-6), ("fp16x3", 2e-6), ("fp16", 2e-3), ("bf16", 1.5e-2)])
def test_conv_table_build_and_chain_against_oracle(cuda, precision, tol):
    """nmtgan_conv_table_build on a packed discriminator: every entry against the fp64 product E[v] . W_w[j] of the
    fp32 master weights (tolerance = the tier's operand / entry rounding), pad columns zero; then prefix pass, prefix max
    and the gather over rollouts give the pooled features of the oracle's target tower (cnn_discriminator.py:461-477)."""
    L = _lib()
    T, R = 12, 2
    gcfg, dcfg = tiny_cfgs(T=T)
    Wg, Wd = O.init_generator_weights(gcfg, 0), disc_weights(dcfg, 1)
    _, disc = make_gpu_pair(gcfg, dcfg, Wg, Wd, precision, T)
    d = disc.c_struct
    V, E = dcfg.vocab_size, dcfg.dim_word
    widths, nfs = list(dcfg.filter_sizes), list(dcfg.num_filters)
    split = precision == "fp16x3"
    entries = "fp32" if precision in ("fp32", "fp16x3") else precision
    es = 4 if entries == "fp32" else 2
    ops = 4 if precision == "fp32" else 2                                # operand bytes
    nfps = [(n * es + 127) // 128 * 128 // es for n in nfs]
    Fp = d.Fp
    rng = np.random.default_rng(8)
    B = 4
    y = rng.integers(1, V, (B, T)).astype(np.int32)
    lens = np.array([T, 7, T, 3])
    for b in range(B):
        y[b, lens[b]:] = 0
    ids, rb, rg = _rows(rng, y, lens, R, V)
    n_rows = ids.shape[0]
    tabs = [torch.full((V, w, npad), float("nan"), device=cuda, dtype=_TD[entries]) for w, npad in zip(widths, nfps)]
    S = [torch.zeros(B * (T - w + 1) * max(w - 1, 1) * npad, device=cuda) for w, npad in zip(widths, nfps)]
    wt = [torch.empty(w * npad * E * (3 if split else 1) * ops, dtype=torch.uint8, device=cuda) for w, npad in zip(widths, nfps)]
    t = L.ConvTable()
    t.n_widths, t.T, t.Fp, t.vocab, t.dtype = len(widths), T, Fp, V, L.op_dtype_of(entries)[1]
    for i, w in enumerate(widths):
        t.tab[i], t.S[i], t.wt[i] = tabs[i].data_ptr(), S[i].data_ptr(), wt[i].data_ptr()
        t.scale[i], t.shift[i] = d.tgt.conv_scale[i], d.tgt.conv_shift[i]
        t.w[i], t.nf[i], t.nfp[i] = w, nfs[i], nfps[i]
    L.check(L.lib.nmtgan_conv_table_build(C.byref(d), C.byref(t), L.stream_ptr()), "build")
    torch.cuda.synchronize()
    emb = torch.from_numpy(Wd["discnn/model_self/embeddingtable"]).double()
    for i, (w, n, npad) in enumerate(zip(widths, nfs, nfps)):
        Wc = torch.from_numpy(Wd["discnn/conv_maxpool-%d/self_model/W" % w]).double()[:, :, 0, :]      # [w][E][n]
        ref = torch.einsum("ve,jen->vjn", emb, Wc)
        got = tabs[i].double().cpu()
        assert float((got[:, :, :n] - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), (precision, w)
        assert float(got[:, :, n:].abs().max()) == 0.0 if npad > n else True
    # the chain on rollouts: prefix pass -> prefix max -> rows -> (facade) highway + head
    y_d, ids_d = torch.from_numpy(y).to(cuda), torch.from_numpy(ids).to(cuda)
    rb_d, rg_d = torch.from_numpy(rb).to(cuda), torch.from_numpy(rg).to(cuda)
    win = torch.zeros((B * T, Fp), device=cuda)
    L.check(L.lib.nmtgan_conv_table_prefix(C.byref(t), y_d.data_ptr(), B, win.data_ptr(), L.stream_ptr()), "prefix")
    torch.cuda.synchronize()
    wv = win.view(B, T, Fp)
    pooled = torch.zeros((n_rows, Fp), device=cuda)
    col = 0
    for w, n in zip(widths, nfs):
        for r in range(n_rows):
            if rg[r] - w + 1 > 0:
                pooled[r, col:col + n] = wv[rb[r], :rg[r] - w + 1, col:col + n].max(dim=0).values
        col += n
    L.check(L.lib.nmtgan_conv_table_rows(C.byref(t), ids_d.data_ptr(), rb_d.data_ptr(), rg_d.data_ptr(), n_rows,
                                         pooled.data_ptr(), L.stream_ptr()), "rows")
    torch.cuda.synchronize()
    # pooled features of the oracle's target tower on the same rows (before the highway layer)
    Wd64 = O.to_torch(Wd)
    e = Wd64["discnn/model_self/embeddingtable"][torch.from_numpy(ids).long()]                      # [rows, T, E]
    col = 0
    F = sum(nfs)
    ref_pool = torch.zeros((n_rows, F), dtype=torch.float64)
    for w, n in zip(widths, nfs):
        ref_pool[:, col:col + n] = O._conv_bn_relu_max(e, Wd64, "discnn/conv_maxpool-%d" % w)
        col += n
    err = float((pooled[:, :F].double().cpu() - ref_pool).abs().max() / ref_pool.abs().max())
    print("%s table chain: pooled features rel err %.3g" % (precision, err))
    assert err < {"fp32": 1e-5, "fp16x3": 1e-5, "fp16": 3e-3, "bf16": 2e-2}[precision]


@pytest.mark.parametrize("case", [
    dict(T=8, widths=[1, 2, 8], nfs=[5, 33, 64], B=3, R=2, V=60, lens=[8, 1, 5]),
    dict(T=17, widths=[3, 4, 16, 17], nfs=[130, 7, 48, 19], B=4, R=3, V=300, lens=[17, 17, 2, 9]),
    dict(T=30, widths=[1, 7, 13, 29], nfs=[110, 170, 230, 390], B=5, R=2, V=128, lens=[30, 12, 30, 3, 21]),
])
def test_get_reward_table_form_equals_gemm_form_on_odd_geometries(cuda, case):
    """Whole get_reward calls in the fp32 tier with the table form forced on against the conv GEMMs and the oracle:
    filter widths 1, 2, T - 1 and T, filter counts that are no multiple of 4, ragged samples down to one token, a
    vocabulary that is no power of two (the generator wants a multiple of 4).  Same rollouts; P(real) of every rollout within fp32 rounding; rewards within
    1e-5 of the oracle's."""
    T, B, R, V = case["T"], case["B"], case["R"], case["V"]
    gcfg = O.GenConfig(hidden_units=64, num_blocks=2, num_heads=4, src_vocab_size=V, dst_vocab_size=V,
                       scale_embedding=True, shared_embedding=False)
    dcfg = O.DisConfig(max_len=T, dim_word=64, vocab_size=V, vocab_size_s=V, filter_sizes=list(case["widths"]),
                       num_filters=list(case["nfs"]))
    Wg, Wd = O.init_generator_weights(gcfg, 3), disc_weights(dcfg, 4)
    rng = np.random.default_rng(T)
    x = rng.integers(4, V, (B, T)).astype(np.int32)
    y = rng.integers(4, V, (B, T)).astype(np.int32)
    for b, n in enumerate(case["lens"]):
        if n < T:
            y[b, n - 1] = O.EOS_ID
    yd, mask = O.deal_generated_samples(y)
    xp = O.extend_sentence_to_maxlen(x, T)
    u = rng.random((R, T - 1, B, T), dtype=np.float32)
    ref = O.get_reward(O.to_torch(Wg), gcfg, O.to_torch(Wd), dcfg, x, xp, yd, mask, R, T, 0.6, u, mode="incremental")
    m, disc = make_gpu_pair(gcfg, dcfg, Wg, Wd, "fp32", T)
    out = {}
    for form in (1, -1):
        m.conv_table = form
        r = m.get_reward(x, xp, yd, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u, return_details=True)
        out[form] = (r, m.last_details)
    n = out[1][1]["plan"].n_rows
    assert np.array_equal(out[1][1]["rollouts"], out[-1][1]["rollouts"])
    dp = float(np.abs(out[1][1]["p_rows"][:n] - out[-1][1]["p_rows"][:n]).max()) if n else 0.0
    print("T=%d: %d rollouts, max |dP(real)| table vs GEMM form %.3g" % (T, n, dp))
    assert dp < 5e-6
    for form in (1, -1):
        assert np.abs(out[form][0] - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), form
