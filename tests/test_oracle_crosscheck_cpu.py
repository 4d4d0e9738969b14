"""The oracle's building blocks against an independent implementation of the same standard operators: PyTorch's own
layer_norm, scaled_dot_product_attention, conv2d + batch_norm, linear layers, Adam and Adadelta.  The reference cannot
run here (Python 2 + TensorFlow 1.2) and ships no fixtures, so nothing pins the oracle to the reference itself; this
file at least pins every operator the oracle restates by hand to a second code base that was written without it -
a slip in a hand-written softmax, variance, window layout or optimizer formula shows up here."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import nmtgan_oracle as O

torch.manual_seed(0)


def test_layer_norm_is_the_library_layer_norm_with_eps_1e6():
    x = torch.randn(5, 7, 64, dtype=torch.float64) * 3 + 1
    W = {"n/layer_norm_scale": torch.randn(64, dtype=torch.float64), "n/layer_norm_bias": torch.randn(64, dtype=torch.float64)}
    want = F.layer_norm(x, (64,), W["n/layer_norm_scale"], W["n/layer_norm_bias"], eps=1e-6)       # common_layers.py:359-374
    assert torch.allclose(O._layer_norm(x, W, "n"), want, rtol=0, atol=1e-12)


def test_attention_is_scaled_dot_product_attention_over_split_heads():
    """common_attention.py:294-331, 389-396: q scaled by dk^-0.5, additive bias, softmax over keys, heads recombined -
    with a causal bias, with a padding bias, and without any."""
    B, Lq, Lk, d, h = 3, 6, 9, 32, 4
    q, k, v = (torch.randn(B, L, d, dtype=torch.float64) for L in (Lq, Lk, Lk))
    pad = torch.zeros(B, 1, 1, Lk, dtype=torch.float64)
    pad[1, ..., 7:] = -1e9
    pad[2, ..., 3:] = -1e9
    causal = torch.full((Lq, Lq), -1e9, dtype=torch.float64).triu(1)[None, None]
    split = lambda t: t.view(B, t.shape[1], h, d // h).transpose(1, 2)
    for bias, kk, vv in ((None, k, v), (pad, k, v), (causal, k[:, :Lq], v[:, :Lq])):
        got = O._attend(q, kk, vv, bias, h)
        ref = F.scaled_dot_product_attention(split(q), split(kk), split(vv), attn_mask=bias, scale=(d // h) ** -0.5)
        assert torch.allclose(got, ref.transpose(1, 2).reshape(B, Lq, d), rtol=0, atol=1e-12)


def test_conv_tower_is_conv2d_batchnorm_relu_maxpool():
    """cnn_discriminator.py:88-109, 22-33, 469: VALID conv over [T, E] windows of width w, bias, inference BatchNorm
    (eps 1e-3), ReLU, max over time - against F.conv2d / F.batch_norm on the NCHW image."""
    dcfg = O.DisConfig(max_len=13, dim_word=16, vocab_size=50, vocab_size_s=50, filter_sizes=[1, 4, 13], num_filters=[6, 9, 5])
    Wd = O.to_torch(O.init_disc_weights(dcfg, 3))
    e = torch.randn(4, 13, 16, dtype=torch.float64)
    for w in dcfg.filter_sizes:
        s = "discnn/conv_maxpool-%d" % w
        got = O._conv_bn_relu_max(e, Wd, s)
        kern = Wd[s + "/self_model/W"].permute(3, 2, 0, 1)                    # [w, E, 1, n] -> [n, 1, w, E]
        c = F.conv2d(e[:, None], kern, Wd[s + "/self_model/b"])              # [B, n, T-w+1, 1]
        bn = s + "/conv_batch_norm/bn/"
        c = F.batch_norm(c, Wd[bn + "moving_mean"], Wd[bn + "moving_variance"], Wd[bn + "gamma"], Wd[bn + "beta"],
                         training=False, eps=1e-3)
        want = torch.relu(c).amax(dim=(2, 3))
        assert torch.allclose(got, want, rtol=0, atol=1e-12), w


def test_dense_ffn_and_highway_are_linear_layers():
    x = torch.randn(3, 5, 8, dtype=torch.float64)
    W = {"f/conv1_single/kernel": torch.randn(8, 32, dtype=torch.float64), "f/conv1_single/bias": torch.randn(32, dtype=torch.float64),
         "f/conv2_single/kernel": torch.randn(32, 8, dtype=torch.float64), "f/conv2_single/bias": torch.randn(8, dtype=torch.float64)}
    want = F.linear(torch.relu(F.linear(x, W["f/conv1_single/kernel"].T, W["f/conv1_single/bias"])),
                    W["f/conv2_single/kernel"].T, W["f/conv2_single/bias"])                       # common_layers.py:1078-1109
    assert torch.allclose(O._ffn(x, W, "f"), want, rtol=0, atol=1e-12)
    H = torch.randn(6, 10, dtype=torch.float64)
    Wh = {"o/weights": torch.randn(10, 10, dtype=torch.float64), "t/weights": torch.randn(10, 10, dtype=torch.float64)}
    t = torch.sigmoid(F.linear(H, Wh["t/weights"].T))
    want = t * torch.relu(F.linear(H, Wh["o/weights"].T)) + (1 - t) * H                            # cnn_discriminator.py:50-59
    assert torch.allclose(O._highway(H, Wh, "o", "t"), want, rtol=0, atol=1e-12)


def test_cdf_sampler_is_searchsorted_on_the_cumulative_softmax():
    """tf.multinomial's CPU kernel as the oracle restates it (model.py:235-241): first index whose fp64 CDF exceeds
    u * total - the same bucket torch.searchsorted finds on the normalised cumulative softmax."""
    rng = np.random.default_rng(2)
    for _ in range(200):
        logits = (rng.standard_normal(97) * 3).astype(np.float32)
        u = float(rng.random())
        cdf = torch.cumsum(torch.softmax(torch.from_numpy(logits).double(), 0), 0)
        want = int(torch.searchsorted(cdf, torch.tensor(u, dtype=torch.float64), right=True).clamp(max=96))
        assert O.multinomial_cdf(logits, np.float32(u)) == want or abs(float(cdf[min(want, 96)]) - u) < 1e-9


def _run_torch_opt(opt_cls, W0, grads, **kw):
    p = {k: torch.nn.Parameter(torch.from_numpy(v.copy())) for k, v in W0.items()}
    opt = opt_cls(list(p.values()), **kw)
    for G in grads:
        for k in p:
            p[k].grad = torch.from_numpy(G[k].copy())
        opt.step()
    return {k: v.detach().numpy() for k, v in p.items()}


def test_adam_and_adadelta_steps_are_the_library_optimizers():
    """The update rules the oracle restates from TF 1.x (cnn_discriminator.py:153-155 AdamOptimizer(), model.py:115-116
    AdadeltaOptimizer()) against torch.optim over three steps.  TF folds Adam's bias corrections into the step size and
    keeps epsilon outside them (epsilon-hat form): the two forms differ by a factor 1 + O(epsilon / |g| sqrt(1 - beta2^t)),
    1e-5 of a step of 1e-3 here."""
    rng = np.random.default_rng(5)
    W0 = {"a": rng.standard_normal((4, 3)), "b": rng.standard_normal(7)}
    grads = [{k: rng.standard_normal(v.shape) for k, v in W0.items()} for _ in range(3)]
    W = {k: v.copy() for k, v in W0.items()}
    m = {k: np.zeros_like(v) for k, v in W.items()}
    v_ = {k: np.zeros_like(v) for k, v in W.items()}
    for t, G in enumerate(grads, 1):
        O.adam_apply(W, G, m, v_, t)
    ref = _run_torch_opt(torch.optim.Adam, W0, grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for k in W:
        assert np.allclose(W[k], ref[k], rtol=0, atol=2e-7)
        assert np.abs(W[k] - W0[k]).max() > 2e-3               # three full steps were taken
    W = {k: v.copy() for k, v in W0.items()}
    acc = {k: np.zeros_like(v) for k, v in W.items()}
    upd = {k: np.zeros_like(v) for k, v in W.items()}
    for G in grads:
        O.adadelta_apply(W, G, acc, upd, lr=1.0, rho=0.95, epsilon=1e-6)
    ref = _run_torch_opt(torch.optim.Adadelta, W0, grads, lr=1.0, rho=0.95, eps=1e-6)
    for k in W:
        assert np.allclose(W[k], ref[k], rtol=0, atol=1e-12)


def test_timing_signal_is_the_tensor2tensor_layout_other_libraries_implement():
    """common_attention.py:57-93 (get_timing_signal_1d: log-spaced timescales, all sines then all cosines): the
    `transformers` library ships the same layout for fairseq-style models and says so in its source ("matches the
    implementation in tensor2tensor"); it computes in fp32."""
    import pytest
    m2m = pytest.importorskip("transformers.models.m2m_100.modeling_m2m_100")
    for length, channels in ((50, 64), (80, 512)):
        ref = m2m.M2M100SinusoidalPositionalEmbedding.get_embedding(length, channels, None).double().numpy()
        assert np.abs(O.timing_signal(length, channels) - ref).max() < 5e-5      # fp32 sines of arguments up to 79


def test_rmsprop_step_is_the_library_optimizer_started_from_ones():
    """tf.train.RMSPropOptimizer(lr) as the oracle restates it (model.py:119: decay 0.9, momentum 0, epsilon 1e-10 inside
    the square root, mean-square slot initialised to ONES) against torch.optim.RMSprop with its slot preset to ones;
    the two place epsilon differently (inside / outside the root), 1e-10 of a mean square that never falls below 0.7 here."""
    rng = np.random.default_rng(6)
    W0 = {"a": rng.standard_normal((5, 2)), "b": rng.standard_normal(3)}
    grads = [{k: rng.standard_normal(v.shape) for k, v in W0.items()} for _ in range(3)]
    W = {k: v.copy() for k, v in W0.items()}
    ms = {k: np.ones_like(v) for k, v in W.items()}
    for G in grads:
        O.rmsprop_apply(W, G, ms, lr=1e-3)
    p = {k: torch.nn.Parameter(torch.from_numpy(v.copy())) for k, v in W0.items()}
    opt = torch.optim.RMSprop(list(p.values()), lr=1e-3, alpha=0.9, eps=1e-10, momentum=0.0, centered=False)
    for q in p.values():
        opt.state[q] = {"step": torch.tensor(0.0), "square_avg": torch.ones_like(q)}
    for G in grads:
        for k in p:
            p[k].grad = torch.from_numpy(G[k].copy())
        opt.step()
    for k in W:
        assert np.allclose(W[k], p[k].detach().numpy(), rtol=0, atol=1e-12)
        assert np.abs(W[k] - W0[k]).max() > 1e-3


def test_training_mode_conv_tower_and_loss_are_the_library_ops():
    """The discriminator's training graph as the oracle restates it (cnn_discriminator.py:88-109, 282-283, 323):
    batch-moment BatchNorm over (batch, time) with a biased variance and eps 1e-3 - F.batch_norm(training=True) on the
    NCHW conv output, whose running-statistics update also reports the batch mean - and the per-example softmax
    cross-entropy against one-hot labels - F.cross_entropy(reduction='none')."""
    dcfg = O.DisConfig(max_len=11, dim_word=8, vocab_size=40, vocab_size_s=40, filter_sizes=[2, 5], num_filters=[6, 4])
    Wd = O.to_torch(O.init_disc_weights(dcfg, 9))
    e = torch.randn(7, 11, 8, dtype=torch.float64)
    for w in dcfg.filter_sizes:
        s = "discnn/conv_maxpool-%d" % w
        stats = {}
        got = O._conv_bn_train_relu_max(e, Wd, s, stats)
        c = F.conv2d(e[:, None], Wd[s + "/self_model/W"].permute(3, 2, 0, 1), Wd[s + "/self_model/b"])
        bn = s + "/conv_batch_norm/bn/"
        run_mean, run_var = torch.zeros(c.shape[1], dtype=torch.float64), torch.ones(c.shape[1], dtype=torch.float64)
        out = F.batch_norm(c, run_mean, run_var, Wd[bn + "gamma"], Wd[bn + "beta"], training=True, momentum=1.0, eps=1e-3)
        assert torch.allclose(got, torch.relu(out).amax(dim=(2, 3)), rtol=0, atol=1e-12)
        assert torch.allclose(stats[s][0], run_mean, rtol=0, atol=1e-12)        # momentum 1: running mean = batch mean
    rng = np.random.default_rng(1)
    x = rng.integers(0, 40, (11, 5))
    xs = rng.integers(0, 40, (11, 5))
    labels = rng.integers(0, 2, 5)
    y = np.eye(2)[labels].T                                                    # [2, B] one-hot, as train() feeds it
    losses, scores, _ = O.disc_train_loss(Wd, dcfg, x, xs, y)
    assert torch.allclose(losses, F.cross_entropy(scores, torch.from_numpy(labels), reduction="none"), rtol=0, atol=1e-12)
