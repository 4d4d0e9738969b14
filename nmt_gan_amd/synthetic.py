"""Synthetic weights and inputs of the benchmark shapes: random
initialisation with the reference's initialisers, TF variable names.  There is no network for
datasets or checkpoints; bench.py and smoke() say `data: synthetic`."""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np

from .share_function import EOS_ID


def _glorot(rng, din, dout):
    lim = math.sqrt(6.0 / (din + dout))            # variance_scaling(1, fan_avg, uniform), model.py:63
    return rng.uniform(-lim, lim, size=(din, dout)).astype(np.float32)


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(size=shape, dtype=np.float32)
    bad = np.abs(out) > 2.0
    while bad.any():                               # tf.truncated_normal redraws beyond 2 sigma
        out[bad] = rng.standard_normal(size=int(bad.sum()), dtype=np.float32)
        bad = np.abs(out) > 2.0
    return out * np.float32(std)


def generator_weights(hidden_units, num_blocks, src_vocab_size, dst_vocab_size, shared_embedding=False,
                      seed=0) -> Dict[str, np.ndarray]:
    """model.py:63, 769, 773-819: embeddings N(0, 512^-1), Glorot-uniform kernels, zero biases,
    LayerNorm scale 1 / bias 0 (a freshly initialised reference model)."""
    rng = np.random.default_rng(seed)
    d = hidden_units
    W = {}
    std = np.float32(512 ** -0.5)
    if shared_embedding:
        W["embedding/shared/kernel"] = rng.standard_normal((dst_vocab_size, d), dtype=np.float32) * std
    else:
        W["embedding/src_embedding/kernel"] = rng.standard_normal((src_vocab_size, d), dtype=np.float32) * std
        W["embedding/dst_embedding/kernel"] = rng.standard_normal((dst_vocab_size, d), dtype=np.float32) * std
        W["embedding/softmax/kernel"] = rng.standard_normal((dst_vocab_size, d), dtype=np.float32) * std

    def dense(name, din, dout):
        W[name + "/kernel"] = _glorot(rng, din, dout)
        W[name + "/bias"] = np.zeros(dout, np.float32)

    def lnorm(name):
        W[name + "/layer_norm_scale"] = np.ones(d, np.float32)
        W[name + "/layer_norm_bias"] = np.zeros(d, np.float32)

    for i in range(num_blocks):
        p = "encoder/block_%d/" % i
        dense(p + "encoder_self_attention/qkv_transform_single", d, 3 * d)
        dense(p + "encoder_self_attention/output_transform_single", d, d)
        dense(p + "conv_hidden_relu/conv1_single", d, 4 * d)
        dense(p + "conv_hidden_relu/conv2_single", 4 * d, d)
        lnorm(p + "layer_norm")
        lnorm(p + "layer_norm_1")
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


def disc_filters(max_len):
    """gan_train.py:47-48."""
    widths = list(range(1, max_len, 4))
    return widths, [100 + 10 * w for w in widths]


def discriminator_weights(max_len, dim_word, vocab_size, vocab_size_s, seed=1, scope="discnn",
                          calibrated=False, head_scale=0.05) -> Dict[str, np.ndarray]:
    """cnn_discriminator.py:80-85, 187-191; BN moving stats randomised (mean~N(0,0.1), var~U[0.5,1.5])
    so inference BN is not a no-op; every variable clipped to [-1,1] as the reference does once at its first
    training step (cnn_discriminator.py:583-591).
    calibrated=True gives the statistics a TRAINED discriminator has instead: the moving mean / variance of every
    filter are those of its own pre-activations over random sentences (mean = conv bias, variance = sum(W^2)/3 for
    U(-1,1) embeddings), so BatchNorm outputs are O(1) as after training (the random statistics above leave them at
    up to +-30 for the wide filters and saturate the 2-way head), and the head is shrunk by `head_scale` so that
    P(real) spreads over (0, 1).  Accuracy tests use it; throughput does not depend on the values."""
    rng = np.random.default_rng(seed)
    widths, nfs = disc_filters(max_len)
    E, F = dim_word, sum(nfs)
    s = scope + "/"
    W = {}
    W[s + "model_self/embeddingtable"] = rng.uniform(-1, 1, (vocab_size, E)).astype(np.float32)
    W[s + "model_self/embeddingtable_s"] = rng.uniform(-1, 1, (vocab_size_s, E)).astype(np.float32)
    W[s + "model_self/W"] = _trunc_normal(rng, (2 * F, 2), 0.1)
    W[s + "model_self/b"] = np.full(2, 0.1, np.float32)
    for fmt in ("conv_maxpool-%d", "conv_s_maxpool-%d"):
        for w, n in zip(widths, nfs):
            p = s + fmt % w + "/"
            W[p + "self_model/W"] = _trunc_normal(rng, (w, E, 1, n), 0.1)
            W[p + "self_model/b"] = np.full(n, 0.1, np.float32)
            W[p + "conv_batch_norm/bn/gamma"] = np.ones(n, np.float32)
            W[p + "conv_batch_norm/bn/beta"] = np.zeros(n, np.float32)
            W[p + "conv_batch_norm/bn/moving_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
            W[p + "conv_batch_norm/bn/moving_variance"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    for nm in ("output_lin_0", "transform_lin_0", "output_s_lin_0", "transform_s_lin_0"):
        W[s + nm + "/weights"] = _glorot(rng, F, F)
    for k in W:
        np.clip(W[k], -1.0, 1.0, out=W[k])
    if calibrated:
        for fmt in ("conv_maxpool-%d", "conv_s_maxpool-%d"):
            for w in widths:
                p = s + fmt % w + "/"
                Wc = W[p + "self_model/W"].astype(np.float64)
                W[p + "conv_batch_norm/bn/moving_mean"] = W[p + "self_model/b"].copy()
                W[p + "conv_batch_norm/bn/moving_variance"] = ((Wc ** 2).sum(axis=(0, 1, 2)) / 3.0).astype(np.float32)
        W[s + "model_self/W"] = (W[s + "model_self/W"] * np.float32(head_scale)).astype(np.float32)
    return W


def synthetic_batch(B, T, src_vocab_size, dst_vocab_size, seed=1234, eos_rows: Sequence[int] = ()):
    """x: T-1 ids in [4,V) + </S>; y_sample: ids in [4,V), no </S> (all rows full length, the worst
    case "len T") except `eos_rows`."""
    rng = np.random.default_rng(seed)
    x = rng.integers(4, src_vocab_size, size=(B, T)).astype(np.int32)
    x[:, T - 1] = EOS_ID
    y = rng.integers(4, dst_vocab_size, size=(B, T)).astype(np.int32)
    for b in eos_rows:
        y[b, int(rng.integers(2, T - 1))] = EOS_ID
    return x, y
