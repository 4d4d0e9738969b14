"""CPU tests of the host side: reference-named helpers, the rollout plan, the C-ABI library
(loads and exports every symbol include/nmtgan.h declares; no compute without a GPU), argument
validation of the entry points, and the N>1 sharding/gather path under gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from nmt_gan_amd import _lib
    header = open(os.path.join(ROOT, "include", "nmtgan.h")).read()
    declared = set(re.findall(r"\b(nmtgan_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libnmtgan_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.lib.nmtgan_version() == 1


def test_ctypes_struct_sizes_match_header_layout():
    """Guards the hand-mirrored ctypes structs: sizes follow from the C layout rules."""
    from nmt_gan_amd import _lib
    assert C.sizeof(_lib.Dense) == 16 and C.sizeof(_lib.Ln) == 16
    assert C.sizeof(_lib.EncBlock) == 6 * 16 and C.sizeof(_lib.DecBlock) == 10 * 16
    # 7 ints+pad, 4 ptrs, max_pos+pad, the blocks, gemm_split+pad
    assert C.sizeof(_lib.Generator) == 7 * 4 + 4 + 4 * 8 + 8 + 16 * (96 + 160) + 8
    assert C.sizeof(_lib.DiscTower) == 8 + 3 * 32 * 8 + 16
    assert C.sizeof(_lib.GemmArgs) % 8 == 0 and C.sizeof(_lib.RewardPlan) % 8 == 0


def test_ctypes_structs_match_the_c_header(tmp_path):
    """Every hand-mirrored ctypes struct has the size the C compiler gives the struct of include/nmtgan.h,
    and its last field sits at the same offset (gcc on the header itself)."""
    from nmt_gan_amd import _lib
    pairs = {"nmtgan_gemm_args": (_lib.GemmArgs, "add_row0"), "nmtgan_embed_args": (_lib.EmbedArgs, "out_split"),
             "nmtgan_addln_args": (_lib.AddLnArgs, "out_split"), "nmtgan_attn_args": (_lib.AttnArgs, "out_split"),
             "nmtgan_attn_group_args": (_lib.AttnGroupArgs, "out_split"), "nmtgan_sample_args": (_lib.SampleArgs, "out"),
             "nmtgan_generator": (_lib.Generator, "gemm_split"), "nmtgan_discriminator": (_lib.Discriminator, "gemm_split"),
             "nmtgan_reward_plan": (_lib.RewardPlan, "conv_table_max_bytes"),
             "nmtgan_conv_table": (_lib.ConvTable, "wt"),
             "nmtgan_disc_tower_params": (_lib.DiscTowerParams, "hw_tr"), "nmtgan_disc_params": (_lib.DiscParams, "head_b")}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "nmtgan.h"', 'int main(void) {']
    for cname, (_, last) in pairs.items():
        src.append('  printf("%s %%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (cname, cname, cname, last))
    src += ['  return 0;', '}']
    cfile = tmp_path / "abi.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "abi"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-I", inc, str(cfile), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.strip().splitlines():
        cname, size, off = line.split()
        cls, last = pairs[cname]
        assert C.sizeof(cls) == int(size), (cname, C.sizeof(cls), size)
        assert getattr(cls, last).offset == int(off), (cname, last)


def test_entry_points_validate_before_launch():
    """Host-side argument checks fail with a message and never touch the GPU."""
    from nmt_gan_amd import _lib
    a = _lib.GemmArgs()
    a.op_dtype, a.epilogue, a.M, a.N, a.K = 0, 0, 4, 4, 3          # K*2 bytes not a multiple of 16
    a.A = a.Wt = 4096
    a.lda = 8
    rc = _lib.lib.nmtgan_gemm(C.byref(a), None)
    assert rc != 0 and b"multiple of 16" in _lib.lib.nmtgan_last_error()
    s = _lib.SampleArgs()
    s.M, s.V, s.logits, s.out, s.u = 1, 30, 4096, 4096, 4096       # V % 4 != 0
    s.ld = 32
    assert _lib.lib.nmtgan_sample_rows(C.byref(s), None) != 0
    t = _lib.AttnArgs()
    t.M, t.d, t.heads = 2, 96, 4                                   # d % 64 != 0
    assert _lib.lib.nmtgan_attn_rows(C.byref(t), None) != 0
    g = _lib.Generator()                                           # update step: shapes are checked before any launch
    g.op_dtype, g.d, g.heads, g.n_blocks, g.vocab_src, g.vocab_dst, g.max_pos = 0, 64, 4, 2, 512, 508, 256
    assert _lib.lib.nmtgan_pg_workspace_bytes(C.byref(g), 4, 8, 8) == 0
    assert b"multiple of 8" in _lib.lib.nmtgan_last_error()
    g.vocab_dst = 512
    assert _lib.lib.nmtgan_pg_workspace_bytes(C.byref(g), 4, 8, 300) == 0      # T beyond the timing table
    assert _lib.lib.nmtgan_pg_workspace_bytes(C.byref(g), 4, 8, 8) > 0
    gr = _lib.Generator()                                                       # two of three gradient tables aliased
    gr.emb_src, gr.emb_dst, gr.softmax_w = 4096, 8192, 8192
    assert _lib.lib.nmtgan_pg_grads(C.byref(g), C.byref(gr), None, None, None, None, None, 4, 8, 8, None, None, None, 0,
                                    None) != 0
    assert b"shared_embedding" in _lib.lib.nmtgan_last_error()
    assert _lib.lib.nmtgan_pg_workspace_bytes(C.byref(g), 4, 200, 200) == 0    # LDS budget of the attention backward
    assert _lib.lib.nmtgan_rmsprop_update(None, None, None, None, 16, 1e-5, 0.9, 0.0, 1e-10, 1.0, 0.0, None, 0, None) != 0
    assert _lib.lib.nmtgan_adadelta_update(4096, 4096, 4096, 4096, 16, 1e-3, 1.5, 1e-8, 1.0, 0.0, None, 0, None) != 0
    assert _lib.lib.nmtgan_adam_update(4096, None, None, 4096, 16, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, 0.0, None, 0, None) != 0
    assert _lib.lib.nmtgan_adam_update(4096, 4096, 4096, 4096, 16, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, 0.0, None, 0, None) != 0
    assert _lib.lib.nmtgan_sgd_update(None, None, 16, 1e-4, 1.0, 0.0, None, 0, None) != 0
    dd = _lib.Discriminator()                                      # discriminator training step
    dd.op_dtype, dd.max_len, dd.dim_word, dd.vocab, dd.vocab_s, dd.n_widths, dd.F, dd.Fp = 0, 12, 64, 512, 512, 2, 250, 256
    dd.widths[0], dd.widths[1], dd.n_filters[0], dd.n_filters[1] = 1, 5, 110, 150
    assert _lib.lib.nmtgan_disc_train_workspace_bytes(C.byref(dd), 4) == 0      # F != sum of the filter counts
    assert b"filter counts" in _lib.lib.nmtgan_last_error()
    dd.F = 260
    dd.Fp = 320
    assert _lib.lib.nmtgan_disc_train_workspace_bytes(C.byref(dd), 4) > 0
    assert _lib.lib.nmtgan_disc_train_workspace_bytes(C.byref(dd), 0) == 0
    assert _lib.lib.nmtgan_clip_values(None, 8, -1.0, 1.0, None, 0, None) != 0
    assert _lib.lib.nmtgan_disc_fold_bn(None, None, None, None, None, 4, None, None, None) != 0
    with pytest.raises(_lib.NmtGanError):
        _lib.check(-2, "x")


def test_tier_pairing_is_checked_on_both_sides():
    """A split-precision generator (fp32 storage, fp16 products) scores only with a 16-bit discriminator: the
    library's plan check refuses an fp32 one before anything is sized or launched, and the facade refuses the same
    pairs with a Python exception."""
    from nmt_gan_amd import _lib
    from nmt_gan_amd.model import check_pairing
    g = _lib.Generator()
    g.op_dtype, g.d, g.heads, g.n_blocks, g.vocab_src, g.vocab_dst, g.max_pos = _lib.NMTGAN_F32, 64, 4, 2, 512, 512, 256
    g.emb_src = g.emb_dst = g.softmax_w = g.timing = 4096
    g.gemm_split = 1
    d = _lib.Discriminator()
    d.op_dtype, d.max_len, d.dim_word, d.vocab, d.vocab_s, d.n_widths, d.F, d.Fp = _lib.NMTGAN_F32, 12, 64, 512, 512, 2, 260, 320
    d.widths[0], d.widths[1], d.n_filters[0], d.n_filters[1] = 1, 5, 110, 150
    live = (C.c_int32 * 1)(0)
    off = (C.c_int64 * 1)(0)
    p = _lib.RewardPlan()
    p.B, p.Ls, p.T, p.rollnum, p.n_rows, p.n_steps = 2, 12, 12, 2, 0, 0
    p.live, p.step_off_host = C.addressof(live), C.addressof(off)
    assert _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p)) == 0
    assert b"16-bit discriminator" in _lib.lib.nmtgan_last_error()
    d.op_dtype = _lib.NMTGAN_F16                                   # plain fp16 and the split form are both accepted
    assert _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p)) > 0
    d.gemm_split = 1
    assert _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p)) > 0
    d.op_dtype = _lib.NMTGAN_BF16                                  # the split form is made of fp16 products
    assert _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p)) == 0
    assert b"gemm_split needs op_dtype NMTGAN_F16" in _lib.lib.nmtgan_last_error()
    d.gemm_split = 0
    g.gemm_split, g.op_dtype = 0, _lib.NMTGAN_F16                  # every other tier packs both networks alike
    assert _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p)) == 0
    assert b"same op dtype" in _lib.lib.nmtgan_last_error()
    for gen, disc in (("fp16x3", "fp32"), ("fp16", "fp32"), ("fp32", "fp16"), ("bf16", "fp16x3")):
        with pytest.raises(ValueError):
            check_pairing(gen, disc)
    for gen, disc in (("fp16x3", "fp16x3"), ("fp16x3", "fp16"), ("fp16", "fp16x3"), ("fp32", "fp32"), ("bf16", "bf16")):
        check_pairing(gen, disc)


def test_conv_form_is_chosen_once_per_batch_and_sized_by_the_library():
    """The table form of the rollouts' conv: the facade picks the form from the WHOLE batch (so the shards of a job and
    the unsharded call agree), the library's workspace query grows by the tables exactly when the form is on, and a cap
    below the tables' size switches it off."""
    import types
    from nmt_gan_amd import _lib
    from nmt_gan_amd.model import Model
    T, R, V = 50, 16, 32000
    disc = types.SimpleNamespace(precision="fp16", vocab_size=V)
    me = types.SimpleNamespace(conv_table=0)
    full = np.ones((64, T), np.float32)
    assert Model.conv_form(me, full, R, T, disc) == 1                      # 50,176 rollouts x 50 tokens over 32,000 entries
    assert Model.conv_form(me, full[:2], R, T, disc) == -1                 # 1,568 rollouts: below 4 tokens per entry
    assert Model.conv_form(me, np.zeros((64, T), np.float32), R, T, disc) == -1
    assert Model.conv_form(me, full, R, T, types.SimpleNamespace(precision="fp32", vocab_size=V)) == -1
    me.conv_table = -1
    assert Model.conv_form(me, full, R, T, disc) == -1
    me.conv_table = 1
    assert Model.conv_form(me, full[:1], R, T, disc) == 1
    # library side: workspace with the tables on / off / capped
    g = _lib.Generator()
    g.op_dtype, g.d, g.heads, g.n_blocks, g.vocab_src, g.vocab_dst, g.max_pos = _lib.NMTGAN_F16, 64, 4, 2, 512, 512, 256
    g.emb_src = g.emb_dst = g.softmax_w = g.timing = 4096
    d = _lib.Discriminator()
    d.op_dtype, d.max_len, d.dim_word, d.vocab, d.vocab_s, d.n_widths, d.F, d.Fp = _lib.NMTGAN_F16, 12, 64, 512, 512, 2, 260, 320
    d.widths[0], d.widths[1], d.n_filters[0], d.n_filters[1] = 1, 5, 110, 150
    n_rows = 8
    live = (C.c_int32 * 1)(n_rows)
    off = (C.c_int64 * 1)(0)
    row_g = (C.c_int32 * n_rows)(*([3] * n_rows))
    p = _lib.RewardPlan()
    p.B, p.Ls, p.T, p.rollnum, p.n_rows, p.n_steps = 2, 12, 12, 2, n_rows, 0
    p.live, p.step_off_host, p.row_g_host = C.addressof(live), C.addressof(off), C.addressof(row_g)
    size = lambda: _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p))
    p.conv_bucket_taps = -1                                                # (the GEMM form's own tables off: sizes comparable)
    p.conv_table = -1
    off_bytes = size()
    p.conv_table = 0
    assert size() == off_bytes                                             # 8 rollouts do not repay a pass over 512 entries
    p.conv_table = 1
    tables = 512 * (1 * 128 + 5 * 192) * 2                                 # vocab x sum(w x filters padded to 64) x fp16
    assert size() >= off_bytes + tables - 4096 and size() < off_bytes + 2 * tables
    d.gemm_split = 1                                                       # split form: fp32 entries, rows padded to 32
    on3 = size()
    p.conv_table = -1
    assert on3 >= size() + 512 * (1 * 128 + 5 * 160) * 4 - 4096
    p.conv_table, p.conv_table_max_bytes = 1, 1000
    assert size() == _lib.lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(d), C.byref(p))
    p.conv_table = -1
    capped_off = size()
    p.conv_table = 1
    assert size() == capped_off                                            # the cap wins over conv_table = 1


def test_models_are_freed_by_reference_counting():
    """A Model / DisCNN and its reference-named `saver` form no cycle: the last reference gone, the object (and with it
    tens of GiB of device workspace at benchmark size) is released at once, not at the next cyclic collection."""
    import gc
    import weakref
    import nmt_gan_amd as N
    from nmt_gan_amd.utils import AttrDict as Cfg
    mcfg = Cfg(hidden_units=64, num_blocks=2, num_heads=4, src_vocab_size=512, dst_vocab_size=512, scale_embedding=True,
               residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
    gc.collect()
    gc.disable()
    try:
        m = N.Model(mcfg, precision="fp16")
        d = N.DisCNN(None, 12, 2, 512, 4, 64, [1, 5], [110, 150], vocab_size_s=512, precision="fp16")
        saver = m.saver
        rm, rd = weakref.ref(m), weakref.ref(d)
        del m, d
        assert rm() is None and rd() is None
        with pytest.raises(RuntimeError):
            saver.save(None, "/tmp/never_written")
    finally:
        gc.enable()


def test_split_packers_reassemble_the_fp32_values():
    """The three split layouts (activation rows [hi | lo | hi], weight rows [hi | hi | lo], filters per tap) hold
    hi + lo = the fp32 value to 2^-22 relative, and the per-tap interleave keeps every tap next to its own lo part."""
    import torch
    from nmt_gan_amd import _lib
    g = torch.Generator().manual_seed(0)
    a = torch.rand((6, 16), generator=g) * 2 - 1
    img = _lib.split3_acts(a).float()
    assert img.shape == (6, 48)
    assert torch.equal(img[:, :16], img[:, 32:]) and float((img[:, :16] + img[:, 16:32] - a).abs().max()) < 2 ** -21
    w = torch.randn((5, 3 * 8), generator=g) * 0.1                  # 5 filters, 3 taps of E = 8
    wi = _lib.split3_conv_weights(w, 3).float().view(5, 3, 3, 8)
    assert torch.equal(wi[:, :, 0], wi[:, :, 1])
    assert float((wi[:, :, 0] + wi[:, :, 2] - w.view(5, 3, 8)).abs().max()) < 2 ** -24
    # the split product over the interleaved layouts equals the fp32 product to fp32 rounding
    e = torch.rand((7, 3, 8), generator=g) * 2 - 1                  # 7 windows of 3 tokens
    ei = _lib.split3_acts(e.view(21, 8)).float().view(7, 3 * 24)
    got = ei.double() @ wi.reshape(5, 72).double().T
    want = e.view(7, 24).double() @ w.double().T
    assert float((got - want).abs().max()) < 1e-6


def test_reference_named_helpers():
    import nmt_gan_amd as N
    y = np.array([[5, 6, 3, 7, 3], [4, 4, 4, 4, 4], [3, 9, 9, 9, 9]], np.int32)
    yd, m = N.deal_generated_samples(y, {"</S>": 3})
    assert np.array_equal(yd, [[5, 6, 0, 0, 0], [4, 4, 4, 4, 4], [0, 0, 0, 0, 0]])
    assert np.array_equal(m, [[1, 1, 0, 0, 0], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]])
    assert yd.dtype == np.int32 and m.dtype == np.float32
    assert np.array_equal(N.extend_sentence_to_maxlen([[1, 2], [3]], 4), [[1, 2, 0, 0], [3, 0, 0, 0]])
    with pytest.raises(ValueError):
        N.extend_sentence_to_maxlen([[1, 2, 3]], 2)
    # same table as the oracle's restatement of add_timing_signal_1d
    from oracle import nmtgan_oracle as O
    assert np.array_equal(N.timing_signal(50, 64), O.timing_signal(50, 64))
    # split_tensor boundaries (model.py:751-753), uneven shards allowed
    assert N.split_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert N.split_bounds(64, 8) == list(range(0, 65, 8))
    cfg = N.AttrDict({"a": 1, "train": {"shared_embedding": False}})
    assert cfg.a == 1 and cfg.train.shared_embedding is False


def test_prepare_gan_dis_data(tmp_path):
    """share_function.py:17-36: num aligned (source, target) pairs, shuffled jointly; intermediates removed."""
    import nmt_gan_amd as N
    src = ["s%d a b" % i for i in range(20)]
    dst = ["t%d c d" % i for i in range(20)]
    fs, ft = tmp_path / "train.src", tmp_path / "train.dst"
    fs.write_text("\n".join(src) + "\n")
    ft.write_text("\n".join(dst) + "\n")
    out_s, out_p = tmp_path / "gan.src", tmp_path / "gan.pos"
    assert N.prepare_gan_dis_data(str(fs), str(ft), str(out_s), str(out_p), num=7, reshuf=True, seed=3) == 7
    got_s, got_p = out_s.read_text().splitlines(), out_p.read_text().splitlines()
    assert len(got_s) == len(got_p) == 7 and len(set(got_s)) == 7
    assert all(a.split()[0][1:] == b.split()[0][1:] for a, b in zip(got_s, got_p))       # pairs stay aligned
    assert got_s != src[:7]                                                               # and were shuffled
    assert N.prepare_gan_dis_data(str(fs), str(ft), str(out_s), str(out_p), num=None, reshuf=False) == 20
    assert out_s.read_text().splitlines() == src and out_p.read_text().splitlines() == dst
    assert N.prepare_gan_dis_data(str(fs), str(ft), str(out_s), str(out_p), num=99, reshuf=True) == 20
    assert sorted(out_s.read_text().splitlines()) == sorted(src)
    ft.write_text("\n".join(dst[:5]) + "\n")
    with pytest.raises(ValueError):
        N.prepare_gan_dis_data(str(fs), str(ft), str(out_s), str(out_p))
    assert not (tmp_path / "train.src.shuf").exists()


def test_rollout_plan_invariants():
    import nmt_gan_amd as N
    rng = np.random.default_rng(0)
    B, T, R = 7, 11, 3
    lens = np.array([11, 0, 1, 2, 6, 11, 4])
    mask = (np.arange(T)[None] < lens[:, None]).astype(np.float32)
    p = N.build_rollout_plan(mask, R, b_offset=20, B_global=64)
    # one row per (b, r, g) with 1 <= g < len_b
    want = sum(max(int(l) - 1, 0) for l in lens) * R
    assert p.n_rows == want == p.live[0]
    steps = lens[p.row_b] - 1 - p.row_g
    assert (np.diff(steps) <= 0).all()                              # sorted by remaining steps
    for s in range(p.n_steps + 1):
        assert p.live[s] == (steps >= s).sum()
        assert (steps[:p.live[s]] >= s).all()                       # live rows are a prefix
    assert p.n_steps == lens.max() - 2
    assert np.array_equal(p.step_off, np.concatenate([[0], np.cumsum(p.live[1:])]))
    # groups of R adjacent rows, r = 0..R-1 in order; grp_first addresses them
    assert np.array_equal(p.row_r.reshape(-1, R), np.tile(np.arange(R), (p.n_rows // R, 1)))
    for b in range(B):
        for g in range(1, T):
            f = p.grp_first[b, g - 1]
            if g < lens[b]:
                assert (p.row_b[f:f + R] == b).all() and (p.row_g[f:f + R] == g).all()
            else:
                assert f == -1
    # uniforms are addressed by GLOBAL sentence id: u[r, g-1, b+20, g] of a [R, T-1, 64, T] tensor
    u = np.arange(R * (T - 1) * 64 * T).reshape(R, T - 1, 64, T)
    assert np.array_equal(u.reshape(-1)[p.row_uoff], u[p.row_r, p.row_g - 1, p.row_b + 20, p.row_g])
    with pytest.raises(ValueError):
        bad = mask.copy()
        bad[0, 3] = 0
        N.build_rollout_plan(bad, R)
    # attention slot table: 32 // R groups of ONE sentence per entry, every group exactly once, key count
    # g + 1, entries sorted by their first row so the entries live at step s are the prefix [0, grp_live[s])
    S = p.grp_slots
    assert S == 32 // R and p.grp_table.shape[1] == 2 + 2 * S
    seen = []
    for e in p.grp_table:
        b, rows, cnts = int(e[0]), e[2:2 + S], e[2 + S:]
        assert rows[0] >= 0
        for r0, c in zip(rows, cnts):
            if r0 >= 0:
                assert p.row_b[r0] == b and p.row_g[r0] + 1 == c and r0 % R == 0
                seen.append(int(r0))
        live_rows = rows[rows >= 0]
        assert (np.diff(live_rows) > 0).all()                       # slot 0 = smallest g = lives longest
    assert sorted(seen) == list(range(0, p.n_rows, R))
    assert (np.diff(p.grp_table[:, 2]) > 0).all()
    for s in range(p.n_steps + 1):
        n = p.grp_live[s]
        assert (p.grp_table[:n, 2] < p.live[s]).all() and (p.grp_table[n:, 2] >= p.live[s]).all()
    big = N.build_rollout_plan(mask, 40)                            # rollnum > 32: no table, plain form
    assert big.grp_slots == 0 and big.grp_table.shape[0] == 0


def test_roll_shard_pieces_cover_every_unit_once():
    """(sentence, roll) units for B < world: every unit on exactly one rank, contiguous, sentence-major."""
    from nmt_gan_amd.share_function import roll_shard_pieces
    for B, R, world in ((5, 16, 8), (1, 4, 2), (3, 4, 8), (1, 2, 8), (7, 3, 8)):
        seen = []
        for rank in range(world):
            u0, u1, pieces = roll_shard_pieces(B, R, world, rank)
            units = [b * R + r for b, r0, r1 in pieces for r in range(r0, r1)]
            assert units == list(range(u0, u1))
            assert all(0 <= r0 < r1 <= R and 0 <= b < B for b, r0, r1 in pieces)
            seen += units
        assert seen == list(range(B * R))


_GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from nmt_gan_amd.model import gather_rows
from nmt_gan_amd.share_function import split_bounds, build_rollout_plan
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
B, T, R = 5, 6, 2                                  # uneven shards: [0,2,5]
bounds = split_bounds(B, 2)
assert bounds == [0, 2, 5]
full = np.arange(B * T, dtype=np.float32).reshape(B, T)
local = torch.from_numpy(full[bounds[rank]:bounds[rank + 1]].copy())
out = gather_rows(local, bounds, torch.device("cpu"))
assert out.shape == (B, T) and np.array_equal(out.numpy(), full), out
# sharded plans draw the same uniforms as the unsharded one
mask = np.ones((B, T), np.float32); mask[3, 4:] = 0
whole = build_rollout_plan(mask, R)
part = build_rollout_plan(mask[bounds[rank]:bounds[rank + 1]], R, b_offset=bounds[rank], B_global=B)
key = lambda p, off: {(int(b) + off, int(r), int(g)): int(u) for b, r, g, u in zip(p.row_b, p.row_r, p.row_g, p.row_uoff)}
kw, kp = key(whole, 0), key(part, bounds[rank])
assert all(kw[k] == v for k, v in kp.items()) and len(kp) == part.n_rows
# roll-level spill (fewer sentences than ranks): (sentence, roll) units, uniforms by global (b, r)
from nmt_gan_amd.share_function import roll_shard_pieces
B1, R1 = 1, 5
mask1 = np.ones((B1, T), np.float32)
whole1 = key(build_rollout_plan(mask1, R1), 0)
u0, u1, pieces = roll_shard_pieces(B1, R1, 2, rank)
assert (u0, u1) == ((0, 2) if rank == 0 else (2, 5)) and pieces == [(0, u0, u1)]
got = {}
for b, r0, r1 in pieces:
    pp = build_rollout_plan(mask1[b:b + 1], r1 - r0, b_offset=b, B_global=B1, r_offset=r0)
    got.update({(int(bb) + b, int(r) + r0, int(g)): int(uu) for bb, r, g, uu in zip(pp.row_b, pp.row_r, pp.row_g, pp.row_uoff)})
assert all(whole1[k] == v for k, v in got.items()) and len(got) == (u1 - u0) * (T - 1)
cache = {}
blk = torch.full((u1 - u0, T), float(rank + 1))
for _ in range(2):                                 # second call re-uses the cached send/receive buffers
    allp = gather_rows(blk, split_bounds(B1 * R1, 2), torch.device("cpu"), cache)
    assert allp.shape == (5, T) and allp[:2].eq(1).all() and allp[2:].eq(2).all()
assert len(cache) == 1
# even shards: the gathered rows could alias the cached receive buffer; by default they do not (two successive
# results own their storage), copy=False hands out the view for callers that consume it at once
ev = torch.full((2, T), float(rank))
r1 = gather_rows(ev, [0, 2, 4], torch.device("cpu"), cache)
r2 = gather_rows(ev + 10, [0, 2, 4], torch.device("cpu"), cache)
assert r1.data_ptr() != r2.data_ptr() and r1[:2].eq(0).all() and r1[2:].eq(1).all() and r2[2:].eq(11).all()
v1 = gather_rows(ev, [0, 2, 4], torch.device("cpu"), cache, copy=False)
v2 = gather_rows(ev + 10, [0, 2, 4], torch.device("cpu"), cache, copy=False)
assert v1.data_ptr() == v2.data_ptr() and v1[2:].eq(11).all()          # the view follows the buffer
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_sharding_and_gather(tmp_path):
    """The N>1 path (sentence sharding with split_tensor bounds + all-gather of reward rows) under
    gloo, world_size 2, on CPU; RCCL uses the same code with backend nccl."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    port = str(29500 + os.getpid() % 2000)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_bench_tier_flags_come_from_the_tracked_accuracy_records():
    """bench.py's `tiers` block reads the accuracy every tier was measured at from the records the repository tracks
    (profiles/r3_precision_c2_<tier>_<vehicle>.json, written by tests/test_gpu_precision_c2.py on an MI355X): the
    split-precision tier meets every clause of the stated tolerance on both vehicles, the fp16 tier the logits clause
    only, and the worst vehicle decides."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    x3 = bench.tier_accuracy("fp16x3")
    assert x3 is not None and set(x3["vehicles"]) == {"reference-init", "trained-like"}
    assert x3["meets_north_star"] == {"logits": True, "tokens": True, "rewards": True, "all": True}
    assert x3["logits_rel"] <= 1e-5 and x3["token_agreement"] >= 0.995 and x3["reward_rel"] <= 1e-3 and x3["p_real_abs"] <= 1e-3
    f16 = bench.tier_accuracy("fp16")
    assert f16["meets_north_star"]["logits"] and not f16["meets_north_star"]["tokens"] and not f16["meets_north_star"]["all"]
    f32 = bench.tier_accuracy("fp32")
    assert f32["tolerance"]["logits_rel"] == 1e-5 and f32["meets_north_star"]["logits"] and f32["meets_north_star"]["tokens"]
    assert bench.tier_accuracy("no-such-tier") is None


def test_bench_table_gather_bytes_match_a_direct_count():
    """bench.py prices the table form of the conv by the bytes it has to gather; the closed loop over give points in
    bench.table_gather_bytes equals a direct walk over every (rollout, width, window, tap)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    T, widths, nfs = 11, [1, 3, 4, 10, 11], [7, 12, 5, 9, 20]
    row_g = np.array([1, 1, 2, 5, 5, 5, 9, 10], np.int64)
    for tes in (2, 4):
        want = 0
        for g in row_g:
            for w, n in zip(widths, nfs):
                for q in range(T - w + 1):
                    if q + w <= g:
                        continue                                   # window inside the prefix: from the prefix max
                    own = sum(1 for j in range(w) if q + j >= g)   # taps that hold the rollout's own tokens
                    want += own * n * tes + (n * 4 if q < g else 0)  # + one fp32 partial-sum row when it straddles
        assert bench.table_gather_bytes(T, widths, nfs, row_g, tes) == want
