"""Discriminator training step ( reference DisCNN.train / build_model / build_train_model,
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


def _disc(dcfg, Wd, precision, optimizer='rmsprop'):
    import nmt_gan_amd as N
    d = N.DisCNN(None, T, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
                 vocab_size_s=dcfg.vocab_size_s, precision=precision, optimizer=optimizer)
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


def test_fp32_two_adadelta_steps_match_oracle(cuda):
    """The constructor's default optimizer, AdadeltaOptimizer(learning_rate=1., rho=0.95, epsilon=1e-6)
    (cnn_discriminator.py:156-158): two steps against the oracle, tolerances as for RMSProp."""
    dcfg, Wd, x, xs, y = _case()
    W = {k: np.array(v, np.float64) for k, v in Wd.items()}
    slots = tuple({k: np.zeros_like(v) for k, v in W.items() if not k.endswith(O.DIS_TRAINABLE_SKIP)} for _ in range(2))
    ref = [O.disc_train_step(W, dcfg, x, xs, y, None, first_step=(i == 0), adadelta=slots) for i in range(2)]
    d = _disc(dcfg, Wd, "fp32", optimizer='adadelta')
    got = [d.train_step(x, y, xs) for _ in range(2)]
    for (l, a), (rl, ra) in zip(got, ref):
        assert abs(l - rl) <= 2e-4 * abs(rl) and a == pytest.approx(ra)
    after = d.get_weights()
    for k in W:
        d_ref = W[k] - np.asarray(Wd[k], np.float64)
        d_got = after[k].astype(np.float64) - np.asarray(Wd[k], np.float64)
        tol = 3e-3 * np.abs(d_ref).max() + 3 * np.finfo(np.float32).eps * max(np.abs(Wd[k]).max(), 1.0)
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)
    with pytest.raises(ValueError):
        _disc(dcfg, Wd, "fp32", optimizer='momentum').train_step(x, y, xs)


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_fp32_two_adam_and_sgd_steps_match_oracle(cuda, optimizer):
    """The remaining dis_optimizer choices (cnn_discriminator.py:153-161): AdamOptimizer() and
    GradientDescentOptimizer(0.0001), two steps against the oracle.  The conv biases are left out of the Adam
    comparison: BatchNorm removes the mean, so their true gradient is 0 and what arrives is rounding noise (1e-14 in
    the fp64 oracle, 1e-7 in fp32), which Adam's m / (sqrt(v) + 1e-8) turns into a full-size step in fp32."""
    dcfg, Wd, x, xs, y = _case()
    W = {k: np.array(v, np.float64) for k, v in Wd.items()}
    tr = [k for k in W if not k.endswith(O.DIS_TRAINABLE_SKIP)]
    m, v = {k: np.zeros_like(W[k]) for k in tr}, {k: np.zeros_like(W[k]) for k in tr}
    for i in range(2):
        O.disc_train_step(W, dcfg, x, xs, y, None, first_step=(i == 0), adam=(m, v, i + 1) if optimizer == "adam" else None,
                          sgd=optimizer == "sgd")
    d = _disc(dcfg, Wd, "fp32", optimizer=optimizer)
    for _ in range(2):
        d.train_step(x, y, xs)
    after = d.get_weights()
    for k in W:
        if optimizer == "adam" and k.endswith("self_model/b"):
            continue
        d_ref = W[k] - np.asarray(Wd[k], np.float64)
        d_got = after[k].astype(np.float64) - np.asarray(Wd[k], np.float64)
        # Adam divides by sqrt(v): an element whose gradient is 1e-4 of the tensor's largest carries the tensor-scale
        # gradient error (1e-4) at full size into its step, hence 2e-2 of the largest step for Adam, 3e-3 for SGD
        tol = (2e-2 if optimizer == "adam" else 3e-3) * np.abs(d_ref).max() + \
            3 * np.finfo(np.float32).eps * max(np.abs(Wd[k]).max(), 1.0)
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)


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


def test_reference_geometry_gradient_is_the_directional_derivative(cuda):
    """Size-independent property at the reference geometry (max_len 50, dim_word 512, 13 widths 1..49 with
    110..590 filters, F = 4550, V = 32000), fp32 tier, 6 examples: along u = g/||g|| the symmetric difference of
    the summed loss equals ||g|| within 3e-2 (the network is piecewise smooth: ReLU, max over time).  Learning rate 0,
    so train_step leaves the trainable variables where they are."""
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    Tn, E, V, B = 50, 512, 32000, 6
    widths, nfs = synthetic.disc_filters(Tn)
    Wd = synthetic.discriminator_weights(Tn, E, V, V, seed=1)
    Wd["discnn/model_self/W"] = (Wd["discnn/model_self/W"] * np.float32(0.02)).astype(np.float32)   # unsaturated head
    rng = np.random.default_rng(9)
    x = rng.integers(4, V, (Tn, B)).astype(np.int32)
    xs = rng.integers(4, V, (Tn, B)).astype(np.int32)
    x[30:, 2] = 0
    y = np.zeros((2, B), np.float32)
    y[rng.integers(0, 2, B), np.arange(B)] = 1.0

    def run(W, want_grads=False):
        d = N.DisCNN(None, Tn, 2, V, B, E, widths, nfs, vocab_size_s=V, precision="fp32", optimizer='rmsprop')
        d.learning_rate = 0.0
        d.load_weights(W)
        return d.train_step(x, y, xs, return_grads=want_grads)

    L0, _, g = run(Wd, True)
    norm = float(np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values())))
    assert np.isfinite(L0) and norm > 0
    eps = 0.01 * abs(L0) * B / norm
    plus, minus = dict(Wd), dict(Wd)
    for k in g:
        step = (eps * g[k].astype(np.float64) / norm).reshape(np.asarray(Wd[k]).shape)
        plus[k] = (np.asarray(Wd[k], np.float64) + step).astype(np.float32)
        minus[k] = (np.asarray(Wd[k], np.float64) - step).astype(np.float32)
    slope = (run(plus)[0] - run(minus)[0]) * B / (2 * eps)
    assert abs(slope - norm) < 3e-2 * norm, (slope, norm, L0, eps)


_TWO_RANK_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
z = np.load(sys.argv[4], allow_pickle=False)
from helpers import tiny_cfgs
import nmt_gan_amd as N
T = int(z["T"])
_, dcfg = tiny_cfgs(T=T)
d = N.DisCNN(None, T, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
             vocab_size_s=dcfg.vocab_size_s, precision="fp32", optimizer='rmsprop')
d.load_weights({k[2:]: z[k] for k in z.files if k.startswith("W:")})
loss, acc = d.train_step(z["x"], z["y"], z["xs"])                    # every rank passes the WHOLE batch
np.savez(sys.argv[5] + ".%d.npz" % dist.get_rank(), loss=np.float64(loss), acc=np.float64(acc), **d.get_weights())
dist.barrier(); dist.destroy_process_group()
print("rank ok")
'''


def test_two_rank_step_is_the_tower_average(cuda, tmp_path):
    """N > 1 path (2 processes, gloo, both on cuda:0): the batch is cut in two like numpy.split
    (cnn_discriminator.py:654-656), each tower normalises BatchNorm with its own half, gradients are averaged, then
    clipped (average_clip_gradient_by_value) and applied.  Trainable variables end identical on both ranks and equal
    to the oracle's step on the averaged fp64 gradients; loss and accuracy are the tower means."""
    import os
    import subprocess
    import sys
    dcfg, Wd, x, xs, y = _case(B=8)
    halves = [O.disc_train_grads(Wd, dcfg, x[:, r * 4:(r + 1) * 4], xs[:, r * 4:(r + 1) * 4], y[:, r * 4:(r + 1) * 4])
              for r in range(2)]
    W = {k: np.array(v, np.float64) for k, v in Wd.items()}
    G = {k: np.clip(0.5 * (halves[0][2][k] + halves[1][2][k]), -1.0, 1.0) for k in halves[0][2]}
    train = {k: W[k] for k in G}
    O.rmsprop_apply(train, G, {k: np.ones_like(v) for k, v in train.items()}, 1e-4)
    for k in G:
        np.clip(W[k], -1.0, 1.0, out=W[k])
    inp = str(tmp_path / "in.npz")
    np.savez(inp, x=x, xs=xs, y=y, T=T, **{"W:" + k: v for k, v in Wd.items()})
    script = tmp_path / "worker.py"
    script.write_text(_TWO_RANK_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(33500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), root, port, str(r), inp, str(tmp_path / "out")],
                              env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "rank ok" in o, o
    got = [dict(np.load(str(tmp_path / "out") + ".%d.npz" % r, allow_pickle=False)) for r in range(2)]
    want_loss, want_acc = 0.5 * (halves[0][0] + halves[1][0]), 0.5 * (halves[0][1] + halves[1][1])
    for r in range(2):
        assert abs(float(got[r]["loss"]) - want_loss) <= 1e-5 * abs(want_loss)
        assert float(got[r]["acc"]) == pytest.approx(want_acc)
    for k in G:
        assert np.array_equal(got[0][k], got[1][k]), k
        d_ref = W[k] - np.asarray(Wd[k], np.float64)
        d_got = got[0][k].astype(np.float64) - np.asarray(Wd[k], np.float64)
        tol = 2e-3 * np.abs(d_ref).max() + 3 * np.finfo(np.float32).eps * max(np.abs(Wd[k]).max(), 1.0)
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)
