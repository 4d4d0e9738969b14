"""Discriminator training step (reference DisCNN.train / build_model / build_train_model,
cnn_discriminator.py:248-345, 561-591, 664-691) through the C ABI against the oracle's train-mode restatement
(autograd in fp64 + gradient clip + RMSProp + moving-statistics update).  Parity unpinned like the rest of the
path (no TensorFlow here); tolerances are written in each test."""
import numpy as np
import pytest

from helpers import disc_weights, rel_err, tiny_cfgs
from oracle import nmtgan_oracle as O

pytestmark = pytest.mark.gpu
T = 12


def _case(B=6, seed=3):
    _, dcfg = tiny_cfgs(T=T)
    Wd = disc_weights(dcfg, 1, head_scale=0.3)
    rng = np.random.default_rng(seed)
    x = rng.integers(0, dcfg.vocab_size, (T, B)).astype(np.int32)
    xs = rng.integers(0, dcfg.vocab_size_s, (T, B)).astype(np.int32)
    x[7:, 1] = 0                                          # padded target / source tails as the iterators produce them
    xs[9:, 4] = 0
    y = np.zeros((2, B), np.float32)
    y[rng.integers(0, 2, B), np.arange(B)] = 1.0
    return dcfg, Wd, x, xs, y


def _disc(dcfg, Wd, precision):
    import nmt_gan_amd as N
    d = N.DisCNN(None, T, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
                 vocab_size_s=dcfg.vocab_size_s, precision=precision, optimizer='rmsprop')
    d.load_weights(Wd)
    return d


def test_fp32_loss_accuracy_and_gradients_match_autograd(cuda):
    """fp32 tier: loss within 1e-5 relative, accuracy equal, every gradient tensor (before clipping) within 2e-4 at
    tensor scale of the fp64 autograd gradient of sum(losses) with BatchNorm in batch-statistics mode."""
    dcfg, Wd, x, xs, y = _case()
    ref_loss, ref_acc, ref_g, _ = O.disc_train_grads(Wd, dcfg, x, xs, y)
    d = _disc(dcfg, Wd, "fp32")
    loss, acc, grads = d.train_step(x, y, xs, return_grads=True)
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss) and acc == pytest.approx(ref_acc)
    assert set(grads) == set(ref_g)
    for k in ref_g:
        assert grads[k].shape == ref_g[k].shape, k
    worst = max((rel_err(grads[k], ref_g[k]), k) for k in ref_g if np.abs(ref_g[k]).max() > 1e-12)
    assert worst[0] < 2e-4, worst
    assert max(np.abs(g).max() for g in ref_g.values()) > 1.0          # the [-1, 1] clip is exercised by this case


def test_fp32_two_steps_match_oracle(cuda):
    """Two iterations (clip by value, RMSProp 1e-4, moving statistics, weight clip after the first): the change of
    every variable within 2e-3 (tensor scale) of the oracle's plus fp32 rounding of the stored value; losses 1e-4."""
    dcfg, Wd, x, xs, y = _case()
    W = {k: np.array(v, np.float64) for k, v in Wd.items()}
    W["discnn/model_self/embeddingtable"][5, :4] = 1.7      # outside [-1, 1]: the one-off weight clip must act
    Wd = {k: v.astype(np.float32) for k, v in W.items()}
    ms = {k: np.ones_like(v) for k, v in W.items() if not k.endswith(O.DIS_TRAINABLE_SKIP)}
    ref = [O.disc_train_step(W, dcfg, x, xs, y, ms, first_step=(i == 0)) for i in range(2)]
    d = _disc(dcfg, Wd, "fp32")
    got = [d.train_step(x, y, xs) for _ in range(2)]
    for (l, a), (rl, ra) in zip(got, ref):
        assert abs(l - rl) <= 1e-4 * abs(rl) and a == pytest.approx(ra)
    after = d.get_weights()
    assert set(after) == set(W)
    for k in W:
        d_ref = W[k] - np.asarray(Wd[k], np.float64)
        d_got = after[k].astype(np.float64) - np.asarray(Wd[k], np.float64)
        tol = 2e-3 * np.abs(d_ref).max() + 3 * np.finfo(np.float32).eps * max(np.abs(Wd[k]).max(), 1.0)
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)
    assert after["discnn/model_self/embeddingtable"].max() <= 1.0


def test_bf16_gradients_close_to_autograd(cuda):
    """bf16 operands: loss within 2e-2; per tensor cosine similarity > 0.98 and L2 error < 0.2 (the arg-max window of
    a filter can move when two windows tie within bf16 rounding, which re-routes that filter's gradient)."""
    dcfg, Wd, x, xs, y = _case()
    ref_loss, _, ref_g, _ = O.disc_train_grads(Wd, dcfg, x, xs, y)
    d = _disc(dcfg, Wd, "bf16")
    loss, _, grads = d.train_step(x, y, xs, return_grads=True)
    assert abs(loss - ref_loss) <= 2e-2 * abs(ref_loss)
    for k in ref_g:
        a, b = grads[k].astype(np.float64).ravel(), ref_g[k].ravel()
        if np.abs(b).max() < 1e-12:
            continue
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b))
        l2 = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert cos > 0.98 and l2 < 0.2, (k, cos, l2)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_trained_weights_drive_inference_and_survive_a_checkpoint(cuda, tmp_path, precision):
    """After train_step the scoring path (dis_ypred_for_auc) runs on the UPDATED filters, moving statistics and
    head: same probabilities (1e-5) as a fresh DisCNN restored from the bundle save() wrote."""
    import nmt_gan_amd as N
    dcfg, Wd, x, xs, y = _case()
    d = _disc(dcfg, Wd, precision)
    p0 = d.run_ypred_for_auc(x, xs, 1.0)
    for _ in range(3):
        d.train_step(x, y, xs)
    p1 = d.run_ypred_for_auc(x, xs, 1.0)
    d.save(str(tmp_path / "disc" / "discriminator"))
    d2 = N.DisCNN(None, T, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
                  vocab_size_s=dcfg.vocab_size_s, precision=precision, reload=True,
                  saveto=str(tmp_path / "disc" / "discriminator"))
    p2 = d2.run_ypred_for_auc(x, xs, 1.0)
    assert np.abs(p1 - p2).max() < 1e-5
    assert np.abs(p1 - p0).max() > 1e-4                  # training moved the scores
    assert np.isfinite(p1).all()
