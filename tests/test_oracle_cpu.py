"""CPU tests of the oracle (test infrastructure): it must reproduce its committed golden vectors,
its two decoding modes must agree, and its host helpers must behave like the reference's.
The reference holds no fixtures for this path: parity vs TensorFlow is UNPINNED."""
import os

import numpy as np
import pytest
import torch

from helpers import disc_weights, tiny_cfgs
from oracle import nmtgan_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_v1.npz")


@pytest.fixture(scope="module")
def gold():
    with np.load(GOLD, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def models(gold):
    gcfg, dcfg = tiny_cfgs(T=int(gold["T"]))
    W = O.to_torch(O.init_generator_weights(gcfg, int(gold["gen_seed"])))
    Wd = O.to_torch(disc_weights(dcfg, int(gold["disc_seed"])))
    return gcfg, dcfg, W, Wd


def test_oracle_reproduces_golden_rewards_and_rollouts(gold, models):
    gcfg, dcfg, W, Wd = models
    T, R = int(gold["T"]), int(gold["R"])
    r, rolls = O.get_reward(W, gcfg, Wd, dcfg, gold["x"], gold["xp"], gold["y"], gold["mask"], R, T, 0.6, gold["u"],
                            mode="incremental", return_rollouts=True)
    assert np.array_equal(rolls, gold["rollouts"])
    assert np.abs(r - gold["rewards_bias06"]).max() < 1e-6
    r2 = O.get_reward(W, gcfg, Wd, dcfg, gold["x"], gold["xp"], gold["y"], gold["mask"], R, T, None, gold["u"],
                      mode="incremental")
    assert np.abs(r2 - gold["rewards_nobias"]).max() < 1e-6
    # bias relation of model.py:336-345
    assert np.abs((r2 - 0.6) * gold["mask"] - r).max() < 1e-6


def test_structure_faithful_equals_incremental(gold, models):
    """KV-cached single-position decoding is result-identical to the reference's full
    re-decode per sampled token (one roll, three prefix lengths)."""
    gcfg, dcfg, W, Wd = models
    T = int(gold["T"])
    x = torch.from_numpy(gold["x"]).long()
    y = torch.from_numpy(gold["y"]).long()
    for g in (1, T // 2, T - 1):
        a = O.rollout_full(W, gcfg, x, y, g, gold["u"][0, g - 1], T)
        b = O.rollout_incremental(W, gcfg, x, y, g, gold["u"][0, g - 1], T)
        assert np.array_equal(a, b)
        assert np.array_equal(a[:, :g], gold["y"][:, :g])
        assert np.array_equal((a * gold["mask"]).astype(np.int32), gold["rollouts"][0, g - 1])


def test_oracle_reproduces_golden_logits_greedy_disc_sampler(gold, models):
    gcfg, dcfg, W, Wd = models
    T = int(gold["T"])
    lg = O.teacher_forced_logits(W, gcfg, torch.from_numpy(gold["x"]).long(), torch.from_numpy(gold["y"]).long()).numpy()
    assert np.abs(lg - gold["tf_logits"]).max() < 1e-5
    assert np.array_equal(O.generate_step(W, gcfg, torch.from_numpy(gold["x"]).long(), T), gold["greedy"])
    p = O.disc_forward(Wd, dcfg, gold["disc_x"], gold["disc_xs"]).numpy()
    assert np.abs(p - gold["disc_probs"]).max() < 1e-9
    ids = [O.multinomial_cdf(gold["samp_logits"][i], gold["samp_u"][i]) for i in range(8)]
    assert np.array_equal(ids, gold["samp_ids"])


def test_fp32_oracle_close_to_fp64(gold, models):
    """The timing baseline runs the same restatement in fp32; it must stay within 1e-4 of fp64."""
    gcfg, dcfg, W, Wd = models
    W32 = {k: v.float() for k, v in W.items()}
    lg = O.teacher_forced_logits(W32, gcfg, torch.from_numpy(gold["x"]).long(), torch.from_numpy(gold["y"]).long()).numpy()
    assert np.abs(lg - gold["tf_logits"]).max() < 1e-4 * np.abs(gold["tf_logits"]).max() + 1e-5


def test_multinomial_cdf_semantics():
    """TF CPU multinomial: upper_bound of u*total in the fp64 CDF; clamp at V-1 (quirk B.10)."""
    l = np.log(np.array([0.1, 0.2, 0.3, 0.4]))
    assert O.multinomial_cdf(l, 0.0) == 0
    assert O.multinomial_cdf(l, 0.0999) == 0
    assert O.multinomial_cdf(l, 0.1001) == 1
    assert O.multinomial_cdf(l, 0.5999) == 2
    assert O.multinomial_cdf(l, 0.6001) == 3
    assert O.multinomial_cdf(l, 1.0) == 3                  # clamped
    assert O.multinomial_cdf(np.array([0.0, -np.inf, 0.0]), 0.5) in (0, 2)
    assert O.multinomial_cdf(np.array([-np.inf, 5.0, -np.inf]), 0.3) == 1


def test_deal_generated_samples_and_extend():
    y = np.array([[5, 6, 3, 7, 3], [4, 4, 4, 4, 4], [3, 9, 9, 9, 9]], np.int32)
    yd, m = O.deal_generated_samples(y)
    assert np.array_equal(yd, [[5, 6, 0, 0, 0], [4, 4, 4, 4, 4], [0, 0, 0, 0, 0]])
    assert np.array_equal(m, [[1, 1, 0, 0, 0], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]])
    assert yd.dtype == np.int32 and m.dtype == np.float32
    xp = O.extend_sentence_to_maxlen([[1, 2], [3]], 4)
    assert np.array_equal(xp, [[1, 2, 0, 0], [3, 0, 0, 0]])
    with pytest.raises(ValueError):
        O.extend_sentence_to_maxlen([[1, 2, 3]], 2)


def test_timing_signal_layout():
    """Sines in the first half of the channels, cosines in the second (common_attention.py:65)."""
    s = O.timing_signal(5, 8)
    assert s.shape == (5, 8) and s.dtype == np.float32
    assert np.allclose(s[0], [0, 0, 0, 0, 1, 1, 1, 1])
    assert np.isclose(s[3, 0], np.sin(3.0)) and np.isclose(s[3, 4], np.cos(3.0))
    assert np.isclose(s[2, 3], np.sin(2.0 * 1e-4), atol=1e-7)      # slowest timescale = 1e4


def test_decoder_doubles_embedding(models):
    """Quirk B.2 (model.py:565): the decoder input is 2*emb*sqrt(d) + timing."""
    gcfg, dcfg, W, Wd = models
    W0 = {k: v.clone() for k, v in W.items()}
    enc = O.encoder(W0, gcfg, torch.tensor([[5, 6, 3]]))
    dec = O.IncrementalDecoder(W0, gcfg, enc)
    tok = torch.tensor([7])
    e = W0["embedding/dst_embedding/kernel"][tok] * torch.tensor(np.float32(gcfg.hidden_units ** 0.5)).double()
    sig = torch.from_numpy(O.timing_signal(4, gcfg.hidden_units)).double()
    # reproduce the first sub-layer input by hand and compare the QKV projection of block 0
    x0 = 2 * e + sig[0]
    qkv = x0 @ W0["decoder/block_0/decoder_self_attention/qkv_transform_single/kernel"] + \
        W0["decoder/block_0/decoder_self_attention/qkv_transform_single/bias"]
    dec.step(tok)
    d = gcfg.hidden_units
    assert torch.allclose(dec.self_k[0][:, 0], qkv[:, d:2 * d], atol=1e-12)
