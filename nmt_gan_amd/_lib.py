"""ctypes binding of libnmtgan_hip.so (include/nmtgan.h).

The HIP library is the product path: if it is missing or does not load, importing this module
raises - there is no CPU or eager-PyTorch fallback anywhere in nmt_gan_amd.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMTGAN_LIB") or os.path.join(_HERE, "libnmtgan_hip.so")   # NMTGAN_LIB: another build of the same ABI

NMTGAN_BF16, NMTGAN_F32, NMTGAN_F16 = 0, 1, 2
EPI_LINEAR, EPI_CONVMAX, EPI_ADDLN = 0, 1, 2
MAX_BLOCKS, MAX_WIDTHS = 16, 32

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t


class GemmArgs(C.Structure):
    _fields_ = [("op_dtype", i32), ("epilogue", i32), ("A", vp), ("Wt", vp), ("M", i32), ("N", i32), ("K", i32),
                ("lda", i64), ("a_group", i32), ("a_group_stride", i64),
                ("bias", vp), ("relu", i32), ("n_seg", i32), ("seg_width", i32),
                ("out", vp * 3), ("ldo", i64 * 3), ("out_f32", i32 * 3), ("seg_scale", f32 * 3),
                ("scale", vp), ("shift", vp), ("pooled", vp), ("ld_pool", i64), ("ldw", i64),
                ("ln_x", vp), ("ln_gamma", vp), ("ln_beta", vp), ("ln_out", vp),
                ("add", vp), ("add_blk", vp), ("add_rows", i32), ("add_row0", i32)]


class EmbedArgs(C.Structure):
    _fields_ = [("table", vp), ("timing", vp), ("vocab", i32), ("max_pos", i32),
                ("tok", vp), ("tok_ld", i64), ("tok_col", vp), ("tok_col_add", i32),
                ("pos", vp), ("pos_mod", i32), ("pos_add", i32), ("mult", f32), ("M", i32), ("d", i32),
                ("out_f32", vp), ("out_op", vp), ("op_dtype", i32), ("out_split", vp)]


class AddLnArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("gamma", vp), ("beta", vp), ("M", i32), ("d", i32),
                ("out_f32", vp), ("out_op", vp), ("op_dtype", i32), ("zero_row", vp),
                ("out_split", vp)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("M", i32), ("d", i32), ("heads", i32), ("kv_dtype", i32),
                ("k1", vp), ("v1", vp), ("len1", i32),
                ("seg1_blk", vp), ("seg1_cnt", vp), ("rows_per_blk", i32), ("causal", i32), ("mask1", vp),
                ("k2", vp), ("v2", vp), ("step_off", vp), ("n_steps2", i32), ("out", vp),
                ("o_init", vp), ("m_init", vp), ("l_init", vp), ("q_op_dtype", i32),
                ("out_split", i32)]


class AttnGroupArgs(C.Structure):
    _fields_ = [("q", vp), ("M", i32), ("d", i32), ("heads", i32), ("rows_per_group", i32),
                ("k", vp), ("vt", vp), ("len", i32), ("len_padded", i32),
                ("row_blk", vp), ("row_cnt", vp), ("mask", vp), ("out", vp),
                ("o_part", vp), ("m_part", vp), ("l_part", vp), ("q_bf16", i32),
                ("slot_table", vp), ("slots", i32), ("n_entries", i32), ("kv_dtype", i32),
                ("out_split", i32)]


class SampleArgs(C.Structure):
    _fields_ = [("logits", vp), ("ld", i64), ("V", i32), ("M", i32), ("row_map", vp),
                ("u", vp), ("u_off", vp), ("u_add", i64), ("greedy", i32), ("out", vp)]


class Dense(C.Structure):
    _fields_ = [("w", vp), ("b", vp)]


class Ln(C.Structure):
    _fields_ = [("g", vp), ("b", vp)]


class EncBlock(C.Structure):
    _fields_ = [("qkv", Dense), ("o", Dense), ("ffn1", Dense), ("ffn2", Dense), ("ln0", Ln), ("ln1", Ln)]


class DecBlock(C.Structure):
    _fields_ = [("qkv", Dense), ("o", Dense), ("cq", Dense), ("ckv", Dense), ("co", Dense),
                ("ffn1", Dense), ("ffn2", Dense), ("ln0", Ln), ("ln1", Ln), ("ln2", Ln)]


class Generator(C.Structure):
    _fields_ = [("op_dtype", i32), ("d", i32), ("heads", i32), ("n_blocks", i32), ("vocab_src", i32),
                ("vocab_dst", i32), ("emb_mult", f32), ("emb_src", vp), ("emb_dst", vp), ("softmax_w", vp),
                ("timing", vp), ("max_pos", i32),
                ("enc", EncBlock * MAX_BLOCKS), ("dec", DecBlock * MAX_BLOCKS), ("gemm_split", i32)]


class DiscTower(C.Structure):
    _fields_ = [("emb", vp), ("conv_w", vp * MAX_WIDTHS), ("conv_scale", vp * MAX_WIDTHS),
                ("conv_shift", vp * MAX_WIDTHS), ("hw_out", vp), ("hw_tr", vp)]


class Discriminator(C.Structure):
    _fields_ = [("op_dtype", i32), ("max_len", i32), ("dim_word", i32), ("vocab", i32), ("vocab_s", i32),
                ("n_widths", i32), ("widths", i32 * MAX_WIDTHS), ("n_filters", i32 * MAX_WIDTHS),
                ("F", i32), ("Fp", i32), ("tgt", DiscTower), ("src", DiscTower), ("head_w", vp), ("head_b", vp),
                ("gemm_split", i32)]


class DiscTowerParams(C.Structure):
    _fields_ = [("emb", vp), ("conv_w", vp * MAX_WIDTHS), ("conv_b", vp), ("gamma", vp), ("beta", vp),
                ("moving_mean", vp), ("moving_var", vp), ("hw_out", vp), ("hw_tr", vp)]


class DiscParams(C.Structure):
    _fields_ = [("tgt", DiscTowerParams), ("src", DiscTowerParams), ("head_w", vp), ("head_b", vp)]


class ConvTable(C.Structure):
    _fields_ = [("n_widths", i32), ("T", i32), ("Fp", i32), ("vocab", i32), ("dtype", i32),
                ("tab", vp * MAX_WIDTHS), ("S", vp * MAX_WIDTHS), ("scale", vp * MAX_WIDTHS), ("shift", vp * MAX_WIDTHS),
                ("w", i32 * MAX_WIDTHS), ("nf", i32 * MAX_WIDTHS), ("nfp", i32 * MAX_WIDTHS), ("wt", vp * MAX_WIDTHS)]


class RewardPlan(C.Structure):
    _fields_ = [("B", i32), ("Ls", i32), ("T", i32), ("rollnum", i32), ("n_rows", i32), ("n_steps", i32),
                ("live", vp), ("step_off_host", vp), ("step_off", vp),
                ("x", vp), ("x_pad", vp), ("x_pad_to_T", vp), ("y", vp), ("dec_in", vp), ("mask", vp),
                ("row_b", vp), ("row_g", vp), ("row_gp1", vp), ("row_lrow", vp), ("row_uoff", vp),
                ("grp_first", vp), ("uniforms", vp), ("use_bias", i32), ("bias_num", f64),
                ("logits_chunk_rows", i32), ("disc_chunk_rows", i32),
                ("rewards", vp), ("rollouts", vp), ("p_rows", vp), ("p_base", vp), ("prefix_logits", vp),
                ("row_g_host", vp), ("grp_table", vp), ("grp_slots", i32), ("grp_live", vp), ("aux_stream", vp), ("aux_gemm_wgs", i32), ("aux_min_rows", i32), ("conv_bucket_taps", i32),
                ("conv_table", i32), ("conv_table_max_bytes", i64)]


EXPORTS = {
    # name: (restype, argtypes)
    "nmtgan_version": (i32, []),
    "nmtgan_last_error": (C.c_char_p, []),
    "nmtgan_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "nmtgan_gemm_grouped": (i32, [C.POINTER(GemmArgs), i32, vp]),
    "nmtgan_conv_table_build": (i32, [C.POINTER(Discriminator), C.POINTER(ConvTable), vp]),
    "nmtgan_conv_table_prefix": (i32, [C.POINTER(ConvTable), vp, i32, vp, vp]),
    "nmtgan_conv_table_rows": (i32, [C.POINTER(ConvTable), vp, vp, vp, i32, vp, vp]),
    "nmtgan_embed": (i32, [C.POINTER(EmbedArgs), vp]),
    "nmtgan_add_layernorm": (i32, [C.POINTER(AddLnArgs), vp]),
    "nmtgan_split3": (i32, [vp, i64, i32, i32, vp, vp]),
    "nmtgan_attn_rows": (i32, [C.POINTER(AttnArgs), vp]),
    "nmtgan_sample_rows": (i32, [C.POINTER(SampleArgs), vp]),
    "nmtgan_logits_sample_workspace_bytes": (sz, [i32, i32]),
    "nmtgan_logits_sample": (i32, [i32, vp, vp, i32, i32, i32, vp, vp, i64, vp, vp, sz, vp]),
    "nmtgan_logits_sample_static": (i32, [i32, vp, vp, i32, i32, i32, vp, vp, i64, vp, vp]),
    "nmtgan_attn_group": (i32, [C.POINTER(AttnGroupArgs), vp]),
    "nmtgan_transpose_v": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nmtgan_transpose_v_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nmtgan_disc_workspace_bytes": (sz, [C.POINTER(Discriminator), i32]),
    "nmtgan_disc_features": (i32, [C.POINTER(Discriminator), i32, vp, i32, vp, i32, vp, sz, vp]),
    "nmtgan_disc_head": (i32, [C.POINTER(Discriminator), vp, vp, vp, i32, vp, vp, vp]),
    "nmtgan_reward_reduce": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f64, vp, vp]),
    "nmtgan_rollout_reward_workspace_bytes": (sz, [C.POINTER(Generator), C.POINTER(Discriminator), C.POINTER(RewardPlan)]),
    "nmtgan_rollout_reward": (i32, [C.POINTER(Generator), C.POINTER(Discriminator), C.POINTER(RewardPlan), vp, sz, vp]),
    "nmtgan_generate_workspace_bytes": (sz, [C.POINTER(Generator), i32, i32, i32]),
    "nmtgan_generate_greedy": (i32, [C.POINTER(Generator), vp, vp, i32, i32, i32, vp, vp, sz, vp]),
    "nmtgan_teacher_forced_workspace_bytes": (sz, [C.POINTER(Generator), i32, i32, i32]),
    "nmtgan_teacher_forced_logits": (i32, [C.POINTER(Generator), vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]),
    "nmtgan_pg_workspace_bytes": (sz, [C.POINTER(Generator), i32, i32, i32]),
    "nmtgan_pg_grads": (i32, [C.POINTER(Generator), C.POINTER(Generator), vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp,
                              sz, vp]),
    "nmtgan_rmsprop_update": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, i32, vp]),
    "nmtgan_adadelta_update": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, i32, vp]),
    "nmtgan_disc_train_workspace_bytes": (sz, [C.POINTER(Discriminator), i32]),
    "nmtgan_disc_train_grads": (i32, [C.POINTER(Discriminator), C.POINTER(DiscParams), C.POINTER(DiscParams), vp, vp, vp, i32,
                                      vp, vp, vp, sz, vp]),
    "nmtgan_disc_fold_bn": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "nmtgan_clip_values": (i32, [vp, i64, f32, f32, vp, i32, vp]),
    "nmtgan_sgd_update": (i32, [vp, vp, i64, f32, f32, f32, vp, i32, vp]),
    "nmtgan_adam_update": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, f32, vp, i32, vp]),
    "nmtgan_profile_classes": (i32, []),
    "nmtgan_profile_start": (i32, []),
    "nmtgan_profile_stop": (i32, [C.POINTER(f64), C.POINTER(i64), i32]),
}

PROFILE_CLASSES = ["other", "gemm_proj", "gemm_ffn", "gemm_logits", "gemm_conv", "gemm_highway", "attn_self_step",
                   "attn_cross_step", "attn_prefill", "sampler", "add_layernorm", "embed", "disc_small",
                   "attn_self_private", "sampler_rows", "conv_table"]


class NmtGanError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nmt_gan_amd: %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C nmt_gan_amd/csrc`). There is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def op_dtype_of(precision: str):
    """(torch dtype, NMTGAN_* code) of the MFMA operand type of a precision tier: 'fp16' (IEEE half operands, the
    throughput tier), 'bf16' (same rate, 8x coarser operand rounding, fp32 range) or 'fp32' (exact, the parity tier)."""
    import torch
    table = {"fp16": (torch.float16, NMTGAN_F16), "bf16": (torch.bfloat16, NMTGAN_BF16),
             "fp32": (torch.float32, NMTGAN_F32)}
    if precision not in table:
        raise ValueError("precision must be 'fp16', 'bf16' or 'fp32'")
    return table[precision]


SPLIT_PRECISION = "fp16x3"       # generator tier: fp32 storage, every GEMM as one fp16 MFMA product over [hi | lo] splits


def split3_weights(w):
    """fp32 [N, K] (torch, any device) -> fp16 [N, 3K] = [W_hi | W_hi | W_lo], W_hi = fp16(W), W_lo = fp16(W - W_hi):
    the weight side of nmtgan_generator.gemm_split (the activation side is [A_hi | A_lo | A_hi])."""
    import torch
    w = w.to(torch.float32)
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    return torch.cat([hi, hi, lo], dim=1).contiguous()


def split3_acts(a):
    """fp32 [M, K] (torch) -> fp16 [M, 3K] = [A_hi | A_lo | A_hi]: the activation side of a split-precision product
    (what nmtgan_split3 writes); used for tables that are gathered into a GEMM operand (the discriminator's embeddings)."""
    import torch
    a = a.to(torch.float32)
    hi = a.to(torch.float16)
    lo = (a - hi.to(torch.float32)).to(torch.float16)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def split3_conv_weights(wt, taps):
    """Conv filter bank Wt[n, taps*E] fp32 (torch) -> fp16 [n, taps*3E], every tap as [W_hi | W_hi | W_lo]: the weight
    side of nmtgan_discriminator.gemm_split, interleaved per tap so that it matches a sentence image whose tokens are
    [e_hi | e_lo | e_hi] rows."""
    n, k = wt.shape
    e = k // taps
    return split3_weights(wt.reshape(n * taps, e)).reshape(n, taps * 3 * e).contiguous()


def check(rc: int, what: str = "nmtgan"):
    if rc != 0:
        raise NmtGanError("%s failed (%d): %s" % (what, rc, lib.nmtgan_last_error().decode("utf-8", "replace")))


def ptr(t):
    """Device (or host) address of a torch tensor / None -> c_void_p value."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
