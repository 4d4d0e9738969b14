"""CPU oracle for the NMT_GAN rollout-reward path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch-CPU / numpy, fp64 or fp32) of the reference
algorithm for the path BASELINE.json:north_star names.  It is the checker for the
HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  The product package (nmt_gan_amd/) never imports, calls or falls back
to anything in oracle/.

PARITY UNPINNED: the reference (Python 2 + TensorFlow 1.2) cannot be run here, it
ships no tests, fixtures or golden vectors, and every arithmetic
kernel lives in TensorFlow (`tensorflow>=1.2.0`, requirements.txt:3), which is not
under /root/reference.  The TF-kernel semantics restated here that the reference
source does not show (recalled from upstream TF 1.x, unverifiable offline):
  * CPU `tf.multinomial`: fp64 CDF of exp(logit - max) in index order,
    std::upper_bound(u * total)           -> `multinomial_cdf`
  * contrib `batch_norm` inference form, epsilon=0.001 -> `_conv_bn_relu_max`
  * `tf.layers.conv2d` has a bias (use_bias=True)      -> `_dense`
The goldens under tests/golden/ were produced by THIS file (fp64) and pin it
against regressions only.  The hand-written operators below (layer norm, attention,
conv + batch norm + pool, feed-forward / highway blocks, CDF sampler, timing signal,
training-mode batch norm, cross-entropy, RMSProp / Adadelta / Adam) are checked
against PyTorch's and the transformers library's own implementations of the same
standard operators in tests/test_oracle_crosscheck_cpu.py - a guard against slips in
the formulas, not a substitute for the reference's outputs.

Each function cites the reference file:line it follows (paths relative to
/root/reference).  Two decoding modes are provided:
  * structure-faithful ("full"): the reference's loop nest - full T-position
    re-decode per sampled token, all-position logits, encoder re-run per rollout
    call, source tower recomputed per discriminator call (model.py:224-245,
    291-346).  Used for the CPU timing baseline.
  * incremental: KV-cached single-position decoding; proves
    the restructuring used on the GPU is result-identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

PAD_ID, UNK_ID, BOS_ID, EOS_ID = 0, 1, 2, 3      # utils.py:39, 269, 287; model.py:764-766
LN_EPS = 1e-6                                     # common_layers.py:359
BN_EPS = 1e-3                                     # contrib batch_norm default epsilon
NEG_INF_BIAS = -1e9                               # common_attention.py:188, 201


# --------------------------------------------------------------------------- config
@dataclass
class GenConfig:
    """Hot-path keys of configs/config_gan_train.yaml:4-18,37."""
    hidden_units: int = 512
    num_blocks: int = 6
    num_heads: int = 8
    src_vocab_size: int = 32000
    dst_vocab_size: int = 32000
    scale_embedding: bool = True
    shared_embedding: bool = False


@dataclass
class DisConfig:
    """DisCNN geometry as built by gan_train.py:47-59."""
    max_len: int = 50
    dim_word: int = 512
    vocab_size: int = 32000
    vocab_size_s: int = 32000
    filter_sizes: List[int] = field(default_factory=list)
    num_filters: List[int] = field(default_factory=list)

    def __post_init__(self):
        if not self.filter_sizes:
            self.filter_sizes = list(range(1, self.max_len, 4))          # gan_train.py:47
        if not self.num_filters:
            self.num_filters = [100 + i * 10 for i in self.filter_sizes]  # gan_train.py:48

    @property
    def num_filters_total(self) -> int:
        return int(sum(self.num_filters))


# --------------------------------------------------------------------------- weights
def _glorot_uniform(rng, fan_in, fan_out, shape):
    lim = math.sqrt(6.0 / (fan_in + fan_out))      # variance_scaling(1, fan_avg, uniform): model.py:63
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():                               # tf.truncated_normal: redraw beyond 2 sigma
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


def init_generator_weights(cfg: GenConfig, seed: int = 0, randomize_affine: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic generator weights with the reference initialisers.

    Names are the TF variable names of model.py:496,514,555,575,773-819 with the conv
    kernels squeezed from [1,1,din,dout] to [din,dout].  `randomize_affine` also
    perturbs biases / LayerNorm scale+bias (zero/one in a fresh reference model) so
    that tests exercise them like a trained checkpoint would.
    """
    rng = np.random.default_rng(seed)
    d, V_s, V_t = cfg.hidden_units, cfg.src_vocab_size, cfg.dst_vocab_size
    W: Dict[str, np.ndarray] = {}
    emb_std = 512 ** -0.5                          # model.py:769 (hard-coded, quirk B.3)
    if cfg.shared_embedding:
        W["embedding/shared/kernel"] = (rng.standard_normal((V_t, d)) * emb_std).astype(np.float32)
    else:
        W["embedding/src_embedding/kernel"] = (rng.standard_normal((V_s, d)) * emb_std).astype(np.float32)
        W["embedding/dst_embedding/kernel"] = (rng.standard_normal((V_t, d)) * emb_std).astype(np.float32)
        W["embedding/softmax/kernel"] = (rng.standard_normal((V_t, d)) * emb_std).astype(np.float32)

    def dense(name, din, dout):
        W[name + "/kernel"] = _glorot_uniform(rng, din, dout, (din, dout))
        W[name + "/bias"] = (rng.standard_normal(dout) * 0.02).astype(np.float32) if randomize_affine \
            else np.zeros(dout, np.float32)

    def lnorm(name):
        if randomize_affine:
            W[name + "/layer_norm_scale"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
            W[name + "/layer_norm_bias"] = (0.1 * rng.standard_normal(d)).astype(np.float32)
        else:
            W[name + "/layer_norm_scale"] = np.ones(d, np.float32)
            W[name + "/layer_norm_bias"] = np.zeros(d, np.float32)

    for i in range(cfg.num_blocks):
        p = "encoder/block_%d/" % i
        dense(p + "encoder_self_attention/qkv_transform_single", d, 3 * d)
        dense(p + "encoder_self_attention/output_transform_single", d, d)
        dense(p + "conv_hidden_relu/conv1_single", d, 4 * d)
        dense(p + "conv_hidden_relu/conv2_single", 4 * d, d)
        lnorm(p + "layer_norm")
        lnorm(p + "layer_norm_1")
    for i in range(cfg.num_blocks):
        p = "decoder/block_%d/" % i
        dense(p + "decoder_self_attention/qkv_transform_single", d, 3 * d)
        dense(p + "decoder_self_attention/output_transform_single", d, d)
        dense(p + "decoder_vanilla_attention/q_transform_single", d, d)
        dense(p + "decoder_vanilla_attention/kv_transform_single", d, 2 * d)
        dense(p + "decoder_vanilla_attention/output_transform_single", d, d)
        dense(p + "conv_hidden_relu/conv1_single", d, 4 * d)
        dense(p + "conv_hidden_relu/conv2_single", 4 * d, d)
        lnorm(p + "layer_norm")
        lnorm(p + "layer_norm_1")
        lnorm(p + "layer_norm_2")
    return W


def init_disc_weights(cfg: DisConfig, seed: int = 1, randomize_bn: bool = True,
                      clip_unit: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic discriminator weights (cnn_discriminator.py:80-85, 187-191, 43-47, 24-32).

    BN moving stats are randomised (mean~N(0,0.1), var~U[0.5,1.5], gamma with a few
    negative entries) so inference BN is not a no-op.  `clip_unit` applies
    the reference's one-off clip of every D variable to [-1,1]
    (cnn_discriminator.py:583-591, 690-691).
    """
    rng = np.random.default_rng(seed)
    E, F = cfg.dim_word, cfg.num_filters_total
    W: Dict[str, np.ndarray] = {}
    s = "discnn/"
    W[s + "model_self/embeddingtable"] = rng.uniform(-1, 1, (cfg.vocab_size, E)).astype(np.float32)
    W[s + "model_self/embeddingtable_s"] = rng.uniform(-1, 1, (cfg.vocab_size_s, E)).astype(np.float32)
    W[s + "model_self/W"] = _trunc_normal(rng, (2 * F, 2), 0.1)
    W[s + "model_self/b"] = np.full(2, 0.1, np.float32)
    for tower in ("conv_maxpool-%d", "conv_s_maxpool-%d"):
        for w, n in zip(cfg.filter_sizes, cfg.num_filters):
            p = s + (tower % w) + "/"
            W[p + "self_model/W"] = _trunc_normal(rng, (w, E, 1, n), 0.1)
            W[p + "self_model/b"] = np.full(n, 0.1, np.float32)
            if randomize_bn:
                gamma = (1.0 + 0.2 * rng.standard_normal(n)).astype(np.float32)
                gamma[:: max(1, n // 7)] *= -1.0
                W[p + "conv_batch_norm/bn/gamma"] = gamma
                W[p + "conv_batch_norm/bn/beta"] = (0.2 * rng.standard_normal(n)).astype(np.float32)
                W[p + "conv_batch_norm/bn/moving_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
                W[p + "conv_batch_norm/bn/moving_variance"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
            else:
                W[p + "conv_batch_norm/bn/gamma"] = np.ones(n, np.float32)
                W[p + "conv_batch_norm/bn/beta"] = np.zeros(n, np.float32)
                W[p + "conv_batch_norm/bn/moving_mean"] = np.zeros(n, np.float32)
                W[p + "conv_batch_norm/bn/moving_variance"] = np.ones(n, np.float32)
    for nm in ("output_lin_0", "transform_lin_0", "output_s_lin_0", "transform_s_lin_0"):
        W[s + nm + "/weights"] = _glorot_uniform(rng, F, F, (F, F))   # tf.get_variable default init
    if clip_unit:
        for k in W:
            np.clip(W[k], -1.0, 1.0, out=W[k])
    return W


def to_torch(W: Dict[str, np.ndarray], dtype=torch.float64) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in W.items()}


# --------------------------------------------------------------------------- host helpers
def deal_generated_samples(y_sample: np.ndarray, eos_id: int = EOS_ID):
    """share_function.py:269-289: zero from the first </S> on; mask = 1 before it."""
    y_sample = np.asarray(y_sample)
    n, L = y_sample.shape
    y = np.zeros((n, L), np.int32)
    m = np.zeros((n, L), np.float32)
    for i, row in enumerate(y_sample):
        lst = row.tolist()
        first = lst.index(eos_id) if eos_id in lst else len(lst)
        y[i, :first] = row[:first]
        m[i, :first] = 1.0
    return y, m


def extend_sentence_to_maxlen(seqs, maxlen: int = 50):
    """share_function.py:240-245 (raises if a sentence is longer than maxlen)."""
    x = np.zeros((len(seqs), maxlen), np.int32)
    for i, s in enumerate(seqs):
        x[i, :len(s)] = s
    return x


def timing_signal(length: int, channels: int) -> np.ndarray:
    """common_attention.py:55-67, evaluated in fp32 exactly as the TF graph does."""
    position = np.arange(length, dtype=np.float32)
    nt = channels // 2
    inc = np.float32(math.log(1.0e4 / 1.0)) / (np.float32(nt) - np.float32(1.0))
    inv = (np.float32(1.0) * np.exp(np.arange(nt, dtype=np.float32) * -inc)).astype(np.float32)
    scaled = position[:, None] * inv[None, :]
    sig = np.concatenate([np.sin(scaled), np.cos(scaled)], axis=1).astype(np.float32)
    if channels % 2:
        sig = np.pad(sig, [[0, 0], [0, 1]])
    return sig


def multinomial_cdf(logits_row: np.ndarray, u) -> int:
    """tf.multinomial CPU kernel semantics (model.py:235-241), injected uniform u.

    c_k = sum_{j<=k} exp(l_j - max) in fp64, index order; token = min{k: c_k > u*c_{V-1}},
    clamped to V-1 (quirk B.10).
    """
    l = np.asarray(logits_row, dtype=np.float64)
    c = np.cumsum(np.exp(l - l.max()))
    k = int(np.searchsorted(c, np.float64(u) * c[-1], side="right"))
    return min(k, l.shape[0] - 1)


# --------------------------------------------------------------------------- generator math
def _dense(x, W, name):
    """1x1 conv == dense + bias (common_layers.py:229-290; tf.layers.conv2d use_bias=True)."""
    return x @ W[name + "/kernel"] + W[name + "/bias"]


def _layer_norm(x, W, name):
    """common_layers.py:336-341, 359-374 (biased variance, eps inside rsqrt)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + LN_EPS) * W[name + "/layer_norm_scale"] + W[name + "/layer_norm_bias"]


def _split_heads(x, h):      # common_attention.py:242-252
    B, L, D = x.shape
    return x.view(B, L, h, D // h).permute(0, 2, 1, 3)


def _attend(q, k, v, bias, h):
    """common_attention.py:389-396, 323-331."""
    dk = q.shape[-1] // h
    q = _split_heads(q, h) * (dk ** -0.5)
    k = _split_heads(k, h)
    v = _split_heads(v, h)
    logits = q @ k.transpose(-1, -2)
    if bias is not None:
        logits = logits + bias
    w = torch.softmax(logits, dim=-1)
    o = w @ v
    B, _, L, _ = o.shape
    return o.permute(0, 2, 1, 3).reshape(B, L, -1)            # combine_heads :255-264


def _self_mha(x, bias, W, prefix, h):
    qkv = _dense(x, W, prefix + "/qkv_transform_single")     # common_attention.py:372-379
    d = x.shape[-1]
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    return _dense(_attend(q, k, v, bias, h), W, prefix + "/output_transform_single")


def _cross_mha(x, mem, bias, W, prefix, h):
    q = _dense(x, W, prefix + "/q_transform_single")         # common_attention.py:381-388
    kv = _dense(mem, W, prefix + "/kv_transform_single")
    d = x.shape[-1]
    k, v = kv[..., :d], kv[..., d:]
    return _dense(_attend(q, k, v, bias, h), W, prefix + "/output_transform_single")


def _ffn(x, W, prefix):
    """common_layers.py:1078-1109 with hidden 4d (model.py:533-537)."""
    return _dense(torch.relu(_dense(x, W, prefix + "/conv1_single")), W, prefix + "/conv2_single")


def _emb_name(cfg, which):
    return "embedding/shared/kernel" if cfg.shared_embedding else "embedding/%s/kernel" % which


def encoder(W, cfg: GenConfig, x: torch.Tensor) -> torch.Tensor:
    """model.py:486-543.  x: int64 [B, Ls]."""
    dt = W[_emb_name(cfg, "src_embedding")].dtype
    pad = x.eq(PAD_ID)
    mult = cfg.hidden_units ** 0.5 if cfg.scale_embedding else 1.0
    h = W[_emb_name(cfg, "src_embedding")][x]
    if mult != 1.0:
        h = h * torch.tensor(np.float32(mult)).to(dt)
    h = h + torch.from_numpy(timing_signal(x.shape[1], cfg.hidden_units)).to(dt)
    bias = (pad.to(dt) * NEG_INF_BIAS)[:, None, None, :]     # common_attention.py:192-202
    for i in range(cfg.num_blocks):
        p = "encoder/block_%d/" % i
        h = _layer_norm(h + _self_mha(h, bias, W, p + "encoder_self_attention", cfg.num_heads), W, p + "layer_norm")
        h = _layer_norm(h + _ffn(h, W, p + "conv_hidden_relu"), W, p + "layer_norm_1")
    return h * (~pad).to(dt)[..., None]                      # model.py:542


def shift_right(y: torch.Tensor) -> torch.Tensor:
    """model.py:764-766."""
    return torch.cat([torch.full_like(y[:, :1], BOS_ID), y[:, :-1]], dim=1)


def decoder(W, cfg: GenConfig, dec_in: torch.Tensor, enc_out: torch.Tensor) -> torch.Tensor:
    """model.py:545-618.  dec_in: int64 [B, T] (already shifted right)."""
    dt = enc_out.dtype
    enc_pad = enc_out.abs().sum(-1).eq(0.0)                  # model.py:546
    enc_bias = (enc_pad.to(dt) * NEG_INF_BIAS)[:, None, None, :]
    mult = cfg.hidden_units ** 0.5 if cfg.scale_embedding else 1.0
    h = W[_emb_name(cfg, "dst_embedding")][dec_in]
    if mult != 1.0:
        h = h * torch.tensor(np.float32(mult)).to(dt)
    T = dec_in.shape[1]
    h = h + (h + torch.from_numpy(timing_signal(T, cfg.hidden_units)).to(dt))   # model.py:565 (quirk B.2)
    tri = torch.tril(torch.ones(T, T, dtype=dt))
    self_bias = (NEG_INF_BIAS * (1.0 - tri))[None, None]      # common_attention.py:178-189
    for i in range(cfg.num_blocks):
        p = "decoder/block_%d/" % i
        h = _layer_norm(h + _self_mha(h, self_bias, W, p + "decoder_self_attention", cfg.num_heads), W, p + "layer_norm")
        h = _layer_norm(h + _cross_mha(h, enc_out, enc_bias, W, p + "decoder_vanilla_attention", cfg.num_heads),
                        W, p + "layer_norm_1")
        h = _layer_norm(h + _ffn(h, W, p + "conv_hidden_relu"), W, p + "layer_norm_2")
    return h


def top(W, cfg: GenConfig, body: torch.Tensor) -> torch.Tensor:
    """model.py:804-820: logits = h . E_softmax^T, no bias."""
    return body @ W[_emb_name(cfg, "softmax")].t()


# --------------------------------------------------------------------------- rollout
def rollout_full(W, cfg, x, y, give_num: int, u_rg: np.ndarray, max_len: int) -> np.ndarray:
    """model.py:203-266, one sess.run of roll_y_sample, structure-faithful.

    u_rg: [B, T] uniforms for this (roll, give_num); entry t is consumed when position t is
    sampled.  Encoder re-run per call (model.py:217-222); per step a full T-position decoder
    pass and all-position logits, one position used (model.py:224-233).
    """
    B = x.shape[0]
    enc = encoder(W, cfg, x)
    given = torch.zeros(B, max_len, dtype=torch.int64)
    given[:, :give_num] = y[:, :give_num]
    for t in range(give_num, max_len):
        logits = top(W, cfg, decoder(W, cfg, shift_right(given), enc))[:, t, :]
        lg = logits.numpy()
        for b in range(B):
            given[b, t] = multinomial_cdf(lg[b], u_rg[b, t])
        # model.py:242-243: positions > t are zero already
    return given.numpy().astype(np.int32)


class IncrementalDecoder:
    """KV-cached single-position decoder: same math as `decoder`, one position
    per call; caches self K/V per block and the per-sentence cross K/V."""

    def __init__(self, W, cfg: GenConfig, enc_out: torch.Tensor):
        self.W, self.cfg = W, cfg
        self.dt = enc_out.dtype
        d = cfg.hidden_units
        enc_pad = enc_out.abs().sum(-1).eq(0.0)
        self.enc_bias = (enc_pad.to(self.dt) * NEG_INF_BIAS)[:, None, None, :]
        self.cross_k, self.cross_v = [], []
        for i in range(cfg.num_blocks):
            kv = _dense(enc_out, W, "decoder/block_%d/decoder_vanilla_attention/kv_transform_single" % i)
            self.cross_k.append(kv[..., :d])
            self.cross_v.append(kv[..., d:])
        self.self_k = [None] * cfg.num_blocks
        self.self_v = [None] * cfg.num_blocks
        self.pos = 0
        self.sig = None

    def step(self, tok: torch.Tensor) -> torch.Tensor:
        """tok: int64 [B] decoder INPUT token at position self.pos -> hidden [B, d]."""
        W, cfg, dt = self.W, self.cfg, self.dt
        d, h = cfg.hidden_units, cfg.num_heads
        mult = d ** 0.5 if cfg.scale_embedding else 1.0
        e = W[_emb_name(cfg, "dst_embedding")][tok]
        if mult != 1.0:
            e = e * torch.tensor(np.float32(mult)).to(dt)
        if self.sig is None or self.sig.shape[0] <= self.pos:
            self.sig = torch.from_numpy(timing_signal(max(64, 2 * (self.pos + 1)), d)).to(dt)
        x = (e + (e + self.sig[self.pos]))[:, None, :]
        for i in range(cfg.num_blocks):
            p = "decoder/block_%d/" % i
            qkv = _dense(x, W, p + "decoder_self_attention/qkv_transform_single")
            q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            self.self_k[i] = k if self.self_k[i] is None else torch.cat([self.self_k[i], k], 1)
            self.self_v[i] = v if self.self_v[i] is None else torch.cat([self.self_v[i], v], 1)
            a = _dense(_attend(q, self.self_k[i], self.self_v[i], None, h), W,
                       p + "decoder_self_attention/output_transform_single")
            x = _layer_norm(x + a, W, p + "layer_norm")
            q = _dense(x, W, p + "decoder_vanilla_attention/q_transform_single")
            a = _dense(_attend(q, self.cross_k[i], self.cross_v[i], self.enc_bias, h), W,
                       p + "decoder_vanilla_attention/output_transform_single")
            x = _layer_norm(x + a, W, p + "layer_norm_1")
            x = _layer_norm(x + _ffn(x, W, p + "conv_hidden_relu"), W, p + "layer_norm_2")
        self.pos += 1
        return x[:, 0, :]


def rollout_incremental(W, cfg, x, y, give_num: int, u_rg: np.ndarray, max_len: int,
                        enc: Optional[torch.Tensor] = None) -> np.ndarray:
    """Same result as `rollout_full` (up to fp summation order) with a KV cache."""
    B = x.shape[0]
    if enc is None:
        enc = encoder(W, cfg, x)
    dec = IncrementalDecoder(W, cfg, enc)
    given = torch.zeros(B, max_len, dtype=torch.int64)
    given[:, :give_num] = y[:, :give_num]
    dec_in = shift_right(given)
    hidden = None
    for p in range(give_num + 1):                            # teacher-forced prefix, positions 0..g
        hidden = dec.step(dec_in[:, p])
    for t in range(give_num, max_len):
        lg = top(W, cfg, hidden).numpy()
        for b in range(B):
            given[b, t] = multinomial_cdf(lg[b], u_rg[b, t])
        if t + 1 < max_len:
            hidden = dec.step(given[:, t])
    return given.numpy().astype(np.int32)


def teacher_forced_logits(W, cfg, x, y_full) -> torch.Tensor:
    """Logits [B, T, V] of decoder(shift_right(y_full)) - used to check sampled tokens under
    forced-identical prefixes."""
    enc = encoder(W, cfg, x)
    return top(W, cfg, decoder(W, cfg, shift_right(y_full), enc))


def generate_step(W, cfg, x, max_len: int) -> np.ndarray:
    """model.py:140-176, 268-271: greedy decode, argmax (first max), no early stop."""
    B = x.shape[0]
    enc = encoder(W, cfg, x)
    cur = torch.zeros(B, max_len, dtype=torch.int64)
    for i in range(max_len):
        logits = top(W, cfg, decoder(W, cfg, shift_right(cur), enc))[:, i, :]
        cur[:, i] = torch.from_numpy(np.argmax(torch.softmax(logits, -1).numpy(), axis=1))
    return cur.numpy().astype(np.int32)


# --------------------------------------------------------------------------- policy-gradient step
def pg_loss(W, cfg: GenConfig, x, y, reward) -> torch.Tensor:
    """model.py:649-666 gan_output on decoder(shift_right(Y)) (model.py:178-184):
    g_loss = -sum_{b,t} softmax(logits)[b, t, Y[b,t]] * reward[b,t] / B   (probabilities, not log-probs)."""
    dt = W[_emb_name(cfg, "softmax")].dtype
    xt = torch.as_tensor(np.asarray(x), dtype=torch.int64)
    yt = torch.as_tensor(np.asarray(y), dtype=torch.int64)
    enc = encoder(W, cfg, xt)
    logits = top(W, cfg, decoder(W, cfg, shift_right(yt), enc))
    p_y = torch.softmax(logits, -1).gather(-1, yt[..., None])[..., 0]
    return -(p_y * torch.as_tensor(np.asarray(reward)).to(dt)).sum() / xt.shape[0]


def pg_loss_and_grads(Wnp: Dict[str, np.ndarray], cfg: GenConfig, x, y, reward, dtype=torch.float64):
    """optimizer.compute_gradients(g_loss) over every trainable variable (model.py:181): autograd through
    this file's encoder/decoder/top.  Returns (loss, {name: grad ndarray in the variable's shape})."""
    W = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).requires_grad_(True) for k, v in Wnp.items()}
    loss = pg_loss(W, cfg, x, y, reward)
    loss.backward()
    return float(loss.item()), {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()
                                for k, v in W.items()}


def rmsprop_apply(W: Dict[str, np.ndarray], G: Dict[str, np.ndarray], ms: Dict[str, np.ndarray], lr: float,
                  decay: float = 0.9, momentum: float = 0.0, epsilon: float = 1e-10, mom=None):
    """tf.train.RMSPropOptimizer(learning_rate) defaults (model.py:119; decay 0.9, momentum 0, epsilon 1e-10,
    `rms` slot initialised to ones), dense ApplyRMSProp as recalled from TF 1.x (unverifiable offline):
    ms += (g*g - ms) * (1 - decay);  mom = mom * momentum + lr * g / sqrt(ms + epsilon);  var -= mom.
    average_gradients (model.py:689-722) densifies the embedding gradients (tf.expand_dims on IndexedSlices), so
    every row of every table takes the dense update.  In place on W and ms (and mom when given)."""
    for k in W:
        g = np.asarray(G[k], W[k].dtype)
        ms[k] += (g * g - ms[k]) * (1.0 - decay)
        step = lr * g / np.sqrt(ms[k] + epsilon)
        if mom is not None:
            mom[k][...] = mom[k] * momentum + step
            step = mom[k]
        W[k] -= step.astype(W[k].dtype)


def adadelta_apply(W, G, accum, accum_update, lr: float = 0.001, rho: float = 0.95, epsilon: float = 1e-8):
    """tf.train.AdadeltaOptimizer() defaults (model.py:115-116), dense ApplyAdadelta as recalled from TF 1.x
    (unverifiable offline; both slots start at zero).  In place."""
    for k in W:
        g = np.asarray(G[k], W[k].dtype)
        accum[k][...] = accum[k] * rho + g * g * (1.0 - rho)
        upd = np.sqrt(accum_update[k] + epsilon) / np.sqrt(accum[k] + epsilon) * g
        accum_update[k][...] = accum_update[k] * rho + upd * upd * (1.0 - rho)
        W[k] -= lr * upd


# --------------------------------------------------------------------------- discriminator
def _conv_bn_relu_max(e, Wd, scope):
    """cnn_layer.conv_op + max_pool (cnn_discriminator.py:88-109, 22-33, 469).  e: [B,T,E]."""
    Wc = Wd[scope + "/self_model/W"]                                  # [w, E, 1, n]
    w, E, _, n = Wc.shape
    win = e.unfold(1, w, 1).permute(0, 1, 3, 2)                      # [B, T-w+1, w, E]
    c = win.reshape(win.shape[0], win.shape[1], w * E) @ Wc.reshape(w * E, n)
    c = c + Wd[scope + "/self_model/b"]                               # bias_add :100
    bn = scope + "/conv_batch_norm/bn/"
    inv = torch.rsqrt(Wd[bn + "moving_variance"] + BN_EPS) * Wd[bn + "gamma"]
    c = c * inv + (Wd[bn + "beta"] - Wd[bn + "moving_mean"] * inv)    # batch_norm(is_training=False)
    return torch.relu(c).max(dim=1).values                            # relu :105, max over time :469


def _highway(H, Wd, out_name, tr_name):
    """highway(...,layer_size=1,bias=0) (cnn_discriminator.py:50-70; called at 481, 501)."""
    o = torch.relu(H @ Wd[out_name + "/weights"])
    t = torch.sigmoid(H @ Wd[tr_name + "/weights"] + 0.0)
    return t * o + (1.0 - t) * H


def disc_tower(Wd, dcfg: DisConfig, ids: torch.Tensor, source: bool) -> torch.Tensor:
    """One tower of build_discriminator_body (cnn_discriminator.py:454-482 / 485-502). ids [B,T]."""
    s = "discnn/"
    emb = Wd[s + ("model_self/embeddingtable_s" if source else "model_self/embeddingtable")]
    e = emb[ids]
    fmt = "conv_s_maxpool-%d" if source else "conv_maxpool-%d"
    pooled = [_conv_bn_relu_max(e, Wd, s + fmt % w) for w in dcfg.filter_sizes]
    H = torch.cat(pooled, dim=1)                                      # :473-477
    if source:
        return _highway(H, Wd, s + "output_s_lin_0", s + "transform_s_lin_0")
    return _highway(H, Wd, s + "output_lin_0", s + "transform_lin_0")


def disc_forward(Wd, dcfg: DisConfig, input_x_TB: np.ndarray, input_xs_TB: np.ndarray,
                 src_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sess.run(dis_ypred_for_auc, {dis_input_x:[T,B], dis_input_xs:[T,B], keep:1.0})
    (cnn_discriminator.py:529-558, 440-522).  Returns softmax probs [B, 2]."""
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(input_x_TB).T)).long()
    ht = disc_tower(Wd, dcfg, x, source=False)
    if src_feat is None:
        xs = torch.from_numpy(np.ascontiguousarray(np.asarray(input_xs_TB).T)).long()
        src_feat = disc_tower(Wd, dcfg, xs, source=True)
    hc = torch.cat([ht, src_feat], dim=1)                             # :504
    scores = hc @ Wd["discnn/model_self/W"] + Wd["discnn/model_self/b"]   # :506
    return torch.softmax(scores, dim=1)                               # :508


# --------------------------------------------------------------------------- discriminator training step
BN_DECAY = 0.9                                    # conv_batch_norm(decay=0.9), cnn_discriminator.py:22
DIS_TRAINABLE_SKIP = ("moving_mean", "moving_variance")


def _conv_bn_train_relu_max(e, Wd, scope, stats):
    """cnn_layer.conv_op(is_train=True) + max_pool (cnn_discriminator.py:88-109, 22-33, 282-283): contrib batch_norm
    in training mode normalises with the BATCH moments over (batch, time) per filter (tf.nn.moments: biased
    variance, epsilon 1e-3) and - updates_collections=None - moves the moving statistics in place:
    moving = moving * decay + batch * (1 - decay).  `stats[scope]` receives (batch mean, batch variance)."""
    Wc = Wd[scope + "/self_model/W"]
    w, E, _, n = Wc.shape
    win = e.unfold(1, w, 1).permute(0, 1, 3, 2)
    c = win.reshape(win.shape[0], win.shape[1], w * E) @ Wc.reshape(w * E, n) + Wd[scope + "/self_model/b"]
    mean = c.mean(dim=(0, 1))
    var = ((c - mean) ** 2).mean(dim=(0, 1))
    stats[scope] = (mean.detach(), var.detach())
    bn = scope + "/conv_batch_norm/bn/"
    c = (c - mean) * torch.rsqrt(var + BN_EPS) * Wd[bn + "gamma"] + Wd[bn + "beta"]
    return torch.relu(c).max(dim=1).values


def disc_train_loss(Wd, dcfg: DisConfig, x_TB, xs_TB, y_2B):
    """DisCNN.build_model (cnn_discriminator.py:248-345) for one tower, dropout keep 1.0 (train() feeds
    drop_prob = 1.0, :640): returns (per-example softmax-CE losses [B], scores [B,2], batch BN stats)."""
    stats = {}
    feats = []
    for source, ids_TB in ((False, x_TB), (True, xs_TB)):
        ids = torch.from_numpy(np.ascontiguousarray(np.asarray(ids_TB).T)).long()
        emb = Wd["discnn/model_self/" + ("embeddingtable_s" if source else "embeddingtable")]
        e = emb[ids]
        fmt = "discnn/conv_s_maxpool-%d" if source else "discnn/conv_maxpool-%d"
        H = torch.cat([_conv_bn_train_relu_max(e, Wd, fmt % w, stats) for w in dcfg.filter_sizes], dim=1)
        feats.append(_highway(H, Wd, "discnn/output_s_lin_0" if source else "discnn/output_lin_0",
                              "discnn/transform_s_lin_0" if source else "discnn/transform_lin_0"))
    scores = torch.cat(feats, dim=1) @ Wd["discnn/model_self/W"] + Wd["discnn/model_self/b"]
    y = torch.from_numpy(np.ascontiguousarray(np.asarray(y_2B, np.float64).T)).to(scores.dtype)
    losses = -(y * torch.log_softmax(scores, dim=1)).sum(dim=1)       # softmax_cross_entropy_with_logits :323
    return losses, scores, stats


def disc_train_grads(Wnp: Dict[str, np.ndarray], dcfg: DisConfig, x_TB, xs_TB, y_2B, dtype=torch.float64):
    """optimizer.compute_gradients(losses, params) (cnn_discriminator.py:334): `losses` is the per-example vector, so
    the gradient is that of its SUM; the reported loss is its mean (:343) and accuracy the argmax match rate (:326).
    Returns (loss, accuracy, {trainable name: grad}, {bn scope: (batch mean, batch var)})."""
    W = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in Wnp.items()}
    for k, v in W.items():
        if not k.endswith(DIS_TRAINABLE_SKIP):
            v.requires_grad_(True)
    losses, scores, stats = disc_train_loss(W, dcfg, x_TB, xs_TB, y_2B)
    losses.sum().backward()
    acc = float((scores.argmax(1).numpy() == np.asarray(y_2B).argmax(0)).mean())
    G = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()
         for k, v in W.items() if v.requires_grad}
    return float(losses.mean().item()), acc, G, {k: (m.numpy(), v.numpy()) for k, (m, v) in stats.items()}


def adam_apply(W, G, m, v, t: int, lr: float = 0.001, beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-8):
    """tf.train.AdamOptimizer() defaults (cnn_discriminator.py:153-155), dense ApplyAdam as recalled from TF 1.x:
    lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t); m, v moving averages from 0; var -= lr_t m / (sqrt(v) + epsilon)."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k in W:
        g = np.asarray(G[k], W[k].dtype)
        m[k][...] = beta1 * m[k] + (1.0 - beta1) * g
        v[k][...] = beta2 * v[k] + (1.0 - beta2) * g * g
        W[k] -= lr_t * m[k] / (np.sqrt(v[k]) + epsilon)


def disc_train_step(W: Dict[str, np.ndarray], dcfg: DisConfig, x_TB, xs_TB, y_2B, ms: Dict[str, np.ndarray], lr=1e-4,
                    first_step=False, adadelta=None, adam=None, sgd=False):
    """One iteration of DisCNN.train (cnn_discriminator.py:561-591, 664-691), single tower, optimizer 'rmsprop'
    (tf.train.RMSPropOptimizer(0.0001), :162-163): gradients clipped to [-1, 1] by value
    (average_clip_gradient_by_value), RMSProp, BN moving statistics moved by the forward pass, and after the FIRST
    step every trainable variable clipped to [-1, 1] (:690-691).  In place on W and ms; returns (loss, accuracy)."""
    loss, acc, G, stats = disc_train_grads(W, dcfg, x_TB, xs_TB, y_2B)
    G = {k: np.clip(g, -1.0, 1.0) for k, g in G.items()}
    train = {k: W[k] for k in G}
    if adadelta is not None:      # (accum, accum_update): AdadeltaOptimizer(learning_rate=1., rho=0.95, epsilon=1e-6), :156-158
        adadelta_apply(train, G, adadelta[0], adadelta[1], lr=1.0, rho=0.95, epsilon=1e-6)
    elif adam is not None:        # (m, v, t): AdamOptimizer(), :153-155
        adam_apply(train, G, adam[0], adam[1], adam[2])
    elif sgd:                     # GradientDescentOptimizer(0.0001), :159-161
        for k in train:
            train[k] -= lr * G[k]
    else:
        rmsprop_apply(train, G, ms, lr)
    for scope, (mean, var) in stats.items():
        bn = scope + "/conv_batch_norm/bn/"
        W[bn + "moving_mean"][...] = W[bn + "moving_mean"] * BN_DECAY + mean * (1.0 - BN_DECAY)
        W[bn + "moving_variance"][...] = W[bn + "moving_variance"] * BN_DECAY + var * (1.0 - BN_DECAY)
    if first_step:
        for k in G:
            np.clip(W[k], -1.0, 1.0, out=W[k])
    return loss, acc


# --------------------------------------------------------------------------- get_reward
def get_reward(W, cfg, Wd, dcfg, x, x_to_maxlen, y_sample, y_sample_mask, rollnum, max_len,
               bias_num, uniforms: np.ndarray, mode: str = "full",
               return_rollouts: bool = False, pairs: Optional[Sequence] = None):
    """model.py:291-346 with injected uniforms[R, T-1, B, T] (index (roll, g-1, b, t)).

    mode "full": structure-faithful loop nest (encoder + source tower recomputed per call).
    mode "incremental": KV-cached decoding, encoder and source tower computed once.
    `pairs` (timing only): restrict the double loop to the given (roll, give_num) pairs.
    Returns fp32 rewards [B, T] (and the masked rollouts [R, T-1, B, T] if asked).
    """
    x_t = torch.from_numpy(np.asarray(x)).long()
    y_t = torch.from_numpy(np.asarray(y_sample)).long()
    mask = np.asarray(y_sample_mask, np.float32)
    B, T = y_t.shape
    assert T == max_len
    x_pad_TB = np.transpose(np.asarray(x_to_maxlen))                  # model.py:294
    enc = src_feat = None
    if mode == "incremental":
        enc = encoder(W, cfg, x_t)
        xs = torch.from_numpy(np.ascontiguousarray(x_pad_TB.T)).long()
        src_feat = disc_tower(Wd, dcfg, xs, source=True)
    rewards: List[np.ndarray] = []
    rolls = np.zeros((rollnum, max_len - 1, B, T), np.int32) if return_rollouts else None
    for i in range(rollnum):                                          # model.py:296
        for give_num in range(1, max_len):                            # :297
            if pairs is not None and (i, give_num) not in pairs:
                if i == 0:
                    rewards.append(np.zeros(B, np.float32))
                continue
            if mode == "full":
                out = rollout_full(W, cfg, x_t, y_t, give_num, uniforms[i, give_num - 1], max_len)
            else:
                out = rollout_incremental(W, cfg, x_t, y_t, give_num, uniforms[i, give_num - 1], max_len, enc=enc)
            out = (out * mask).astype(np.int32)                       # :307
            if rolls is not None:
                rolls[i, give_num - 1] = out
            ypred = disc_forward(Wd, dcfg, out.T, x_pad_TB, src_feat=src_feat)[:, 1].numpy().astype(np.float32)  # :311-315
            if i == 0:
                rewards.append(ypred)
            else:
                rewards[give_num - 1] = rewards[give_num - 1] + ypred   # :319
        ypred = disc_forward(Wd, dcfg, np.asarray(y_sample).T, x_pad_TB, src_feat=src_feat)[:, 1].numpy().astype(np.float32)
        if i == 0:
            rewards.append(ypred)                                     # :330
        else:
            rewards[max_len - 1] = rewards[max_len - 1] + ypred       # :332
    r = np.transpose(np.array(rewards, dtype=np.float32))             # :334
    if bias_num is None:
        r = r * mask
        r = r / np.float32(1.0 * rollnum)
    else:
        bias = np.zeros_like(r)
        bias += bias_num * rollnum                                    # :340-341
        r = (r - bias) * mask                                         # :342-344
        r = r / np.float32(1.0 * rollnum)                             # :345
    r = r.astype(np.float32)
    return (r, rolls) if return_rollouts else r


# --------------------------------------------------------------------------- synthetic inputs
def synthetic_batch(B: int, T: int, Vs: int, Vt: int, seed: int = 1234, eos_rows: Sequence[int] = (),
                    short_src_rows: Sequence[int] = ()):
    """Synthetic benchmark inputs: x = T-1 ids in [4,V) + </S>; y_sample ids in [4,V), no EOS
    (all rows full length) except `eos_rows`, which get an early </S>; `short_src_rows` get a
    shorter, zero-padded source."""
    rng = np.random.default_rng(seed)
    x = rng.integers(4, Vs, size=(B, T), dtype=np.int64).astype(np.int32)
    x[:, T - 1] = EOS_ID
    for b in short_src_rows:
        L = int(rng.integers(2, T - 1))
        x[b, L - 1] = EOS_ID
        x[b, L:] = 0
    y = rng.integers(4, Vt, size=(B, T), dtype=np.int64).astype(np.int32)
    for b in eos_rows:
        y[b, int(rng.integers(2, T - 1))] = EOS_ID
    return x, y
