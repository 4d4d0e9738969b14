"""Policy-gradient update of the generator ( reference Model.generate_step_and_update,
model.py:273-276, 178-190, 649-666) through the C ABI against the oracle's autograd (fp64) + RMSProp
restatement.  Parity unpinned like the rest of the path (no TensorFlow here): the oracle is this repo's
restatement; tolerances are written in each test."""
import numpy as np
import pytest

from helpers import Cfg, disc_weights, facade_config, make_gpu_pair, rel_err, tiny_cfgs
from oracle import nmtgan_oracle as O

pytestmark = pytest.mark.gpu


def _case(T=12, B=5, seed=11, reward_scale=1.0):
    gcfg, dcfg = tiny_cfgs(T=T)
    Wg = O.init_generator_weights(gcfg, 0)
    x, y = O.synthetic_batch(B, T, gcfg.src_vocab_size, gcfg.dst_vocab_size, seed=seed,
                             eos_rows=tuple(b for b in (1, 3) if b < B), short_src_rows=(2,))
    yd, mask = O.deal_generated_samples(y)
    rng = np.random.default_rng(seed)
    reward = ((rng.random((B, T)).astype(np.float32) - np.float32(0.6)) * mask * np.float32(reward_scale)).astype(np.float32)
    return gcfg, dcfg, Wg, x, yd, mask, reward


def _model(gcfg, Wg, precision, T, lr=1e-5):
    import nmt_gan_amd as N
    cfg = facade_config(gcfg)
    cfg["generator"] = Cfg(learning_rate=lr)
    m = N.Model(cfg, precision=precision)
    m.build_train_model()
    m.build_generate(T, optimizer='rmsprop')
    m.build_rollout_generate(T)
    m.init_and_restore(weights=Wg)
    return m


def test_fp32_loss_and_gradients_match_autograd(cuda):
    """fp32 tier: loss within 1e-5 relative, every gradient tensor within 1e-4 at tensor scale
    (||dgpu - dref||_inf / ||dref||_inf) of the fp64 autograd gradient."""
    gcfg, _, Wg, x, y, mask, reward = _case()
    ref_loss, ref_g = O.pg_loss_and_grads(Wg, gcfg, x, y, reward)
    m = _model(gcfg, Wg, "fp32", y.shape[1])
    loss, grads = m.generate_step_and_update(x, y, reward, return_grads=True)
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss)
    assert set(grads) == set(ref_g)
    worst = max((rel_err(grads[k], ref_g[k]), k) for k in ref_g if np.abs(ref_g[k]).max() > 0)
    assert worst[0] < 1e-4, worst
    for k in ref_g:                                     # variables the loss cannot reach stay at exactly zero
        if np.abs(ref_g[k]).max() == 0:
            assert np.abs(grads[k]).max() == 0, k
    # rows of the embedding tables no token touched have zero gradient (dense gradient of tf.gather)
    used = np.unique(x)
    untouched = np.setdiff1d(np.arange(gcfg.src_vocab_size), used)
    assert np.abs(grads["embedding/src_embedding/kernel"][untouched]).max() == 0


def test_fp32_two_updates_match_oracle_rmsprop(cuda):
    """Two RMSProp steps (lr raised so the step is far above fp32 rounding of the weights): the weight change
    of every tensor within 2e-3 (tensor scale) of the oracle's plus the fp32 rounding of the stored weight, the
    second loss within 1e-4 relative."""
    lr = 1e-2
    gcfg, _, Wg, x, y, mask, reward = _case(reward_scale=50.0)
    W = {k: np.array(v, np.float64) for k, v in Wg.items()}
    ms = {k: np.ones_like(v) for k, v in W.items()}
    ref_losses = []
    for _ in range(2):
        l, G = O.pg_loss_and_grads(W, gcfg, x, y, reward)
        ref_losses.append(l)
        O.rmsprop_apply(W, G, ms, lr)
    m = _model(gcfg, Wg, "fp32", y.shape[1], lr=lr)
    losses = [m.generate_step_and_update(x, y, reward) for _ in range(2)]
    assert abs(losses[0] - ref_losses[0]) <= 1e-5 * abs(ref_losses[0])
    assert abs(losses[1] - ref_losses[1]) <= 1e-4 * abs(ref_losses[1])
    assert abs(ref_losses[1] - ref_losses[0]) > 1e-3 * abs(ref_losses[0])      # the step is visible
    got = m.get_weights()
    for k in W:
        d_ref = W[k] - np.asarray(Wg[k], np.float64)
        d_got = got[k].astype(np.float64) - np.asarray(Wg[k], np.float64)
        assert got[k].shape == np.asarray(Wg[k]).shape
        # two fp32 roundings of the stored weight (one per step) ride on top of the 2e-3 of the step itself
        tol = 2e-3 * np.abs(d_ref).max() + 2 * np.finfo(np.float32).eps * np.abs(Wg[k]).max()
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)


def test_bf16_gradients_close_to_autograd(cuda):
    """bf16 operands (fp32 accumulation, fp32 gradients): loss within 2e-2 relative; per tensor the gradient
    direction agrees (cosine similarity > 0.99) and ||dgpu - dref||_2 < 0.15 ||dref||_2 - operand rounding (2^-9)
    compounds over 18 sub-layers forward and 18 backward, and a hidden unit whose pre-activation rounds across zero
    flips its ReLU mask, which moves single elements of the FFN gradients by tens of percent (60 rows only)."""
    gcfg, _, Wg, x, y, mask, reward = _case()
    ref_loss, ref_g = O.pg_loss_and_grads(Wg, gcfg, x, y, reward)
    m = _model(gcfg, Wg, "bf16", y.shape[1])
    loss, grads = m.generate_step_and_update(x, y, reward, return_grads=True)
    assert abs(loss - ref_loss) <= 2e-2 * abs(ref_loss)
    for k in ref_g:
        a, b = grads[k].astype(np.float64).ravel(), ref_g[k].ravel()
        if np.abs(b).max() == 0:
            continue
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b))
        l2 = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert cos > 0.99 and l2 < 0.15, (k, cos, l2)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_updated_weights_drive_the_rollout_path_and_survive_a_checkpoint(cuda, tmp_path, precision):
    """gan_train.py:102-118 in sequence: get_reward -> generate_step_and_update -> saver.save, then the next
    iteration's get_reward runs on the UPDATED weights: identical to a fresh model restored from the bundle."""
    T, R = 10, 2
    gcfg, dcfg, Wg, x, y, mask, _ = _case(T=T, B=3)
    Wd = disc_weights(dcfg, 1)
    m, disc = make_gpu_pair(gcfg, dcfg, Wg, Wd, precision, T)
    m.config["generator"] = Cfg(learning_rate=1e-2)
    xp = O.extend_sentence_to_maxlen(x, T)
    u = np.random.default_rng(5).random((R, T - 1, 3, T), dtype=np.float32)
    r0 = m.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u)
    loss = m.generate_step_and_update(x, y, r0 * np.float32(50.0))
    assert np.isfinite(loss)
    r1 = m.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u)
    m.save(str(tmp_path / "gen" / "model_step_1"))
    m2, _ = make_gpu_pair(gcfg, dcfg, Wg, Wd, precision, T)
    m2.init_and_restore(modelFile=str(tmp_path / "gen" / "model_step_1"))
    r2 = m2.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u)
    assert np.array_equal(r1, r2)
    assert np.array_equal(m.generate_step(x), m2.generate_step(x))
    assert not np.array_equal(r0, r1)                  # the update moved the policy


def test_update_rejects_what_it_does_not_support(cuda):
    gcfg, _, Wg, x, y, mask, reward = _case()
    m = _model(gcfg, Wg, "fp32", y.shape[1])
    with pytest.raises(ValueError):
        m.generate_step_and_update(x, y, reward[:, :-1])


def test_fp32_shared_embedding_update_matches_oracle(cuda):
    """train.shared_embedding: True (model.py:777-818): one table serves as source embedding, target embedding and
    softmax matrix; its gradient is the sum of the three contributions.  Loss 1e-5, gradients 1e-4, one RMSProp step."""
    import nmt_gan_amd as N
    gcfg, _ = tiny_cfgs(T=12, shared=True)
    Wg = O.init_generator_weights(gcfg, 0)
    assert "embedding/shared/kernel" in Wg
    _, _, _, x, y, mask, reward = _case()
    ref_loss, ref_g = O.pg_loss_and_grads(Wg, gcfg, x, y, reward)
    cfg = facade_config(gcfg)
    cfg["generator"] = Cfg(learning_rate=1e-5)
    m = N.Model(cfg, precision="fp32")
    m.build_generate(y.shape[1])
    m.init_and_restore(weights=Wg)
    loss, grads = m.generate_step_and_update(x, y, reward, return_grads=True)
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss)
    assert set(grads) == set(ref_g)
    worst = max((rel_err(grads[k], ref_g[k]), k) for k in ref_g if np.abs(ref_g[k]).max() > 0)
    assert worst[0] < 1e-4, worst
    assert np.isfinite(m.generate_step_and_update(x, y, reward))        # and the updated shared table still drives the path
    assert m.generate_step(x).shape == y.shape


def test_fp32_two_adadelta_updates_match_oracle(cuda):
    """build_generate(optimizer='adadelta') (model.py:115-116): two steps of tf.train.AdadeltaOptimizer() defaults
    against the oracle's restatement, same tolerances as the RMSProp test."""
    gcfg, _, Wg, x, y, mask, reward = _case(reward_scale=50.0)
    W = {k: np.array(v, np.float64) for k, v in Wg.items()}
    acc, acc_u = {k: np.zeros_like(v) for k, v in W.items()}, {k: np.zeros_like(v) for k, v in W.items()}
    ref_losses = []
    for _ in range(2):
        l, G = O.pg_loss_and_grads(W, gcfg, x, y, reward)
        ref_losses.append(l)
        O.adadelta_apply(W, G, acc, acc_u)
    m = _model(gcfg, Wg, "fp32", y.shape[1])
    m.build_generate(y.shape[1], optimizer='adadelta')
    losses = [m.generate_step_and_update(x, y, reward) for _ in range(2)]
    assert abs(losses[1] - ref_losses[1]) <= 1e-4 * abs(ref_losses[1])
    got = m.get_weights()
    for k in W:
        d_ref = W[k] - np.asarray(Wg[k], np.float64)
        d_got = got[k].astype(np.float64) - np.asarray(Wg[k], np.float64)
        tol = 2e-3 * np.abs(d_ref).max() + 2 * np.finfo(np.float32).eps * np.abs(Wg[k]).max()
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)


def test_production_width_gradient_is_the_directional_derivative(cuda):
    """Size-independent property at the production width (d=512, 8 heads, V=32000, T=50; 2 blocks, 4 sentences to
    keep it quick), fp32 tier: along u = g/||g|| the symmetric difference (L(W + eps u) - L(W - eps u)) / (2 eps)
    equals ||g|| within 2e-2.  The learning rate is 0, so the update leaves the weights where they are."""
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    d, blocks, heads, V, T, B = 512, 2, 8, 32000, 50, 4
    Wg = synthetic.generator_weights(d, blocks, V, V, seed=0)
    cfg = Cfg(hidden_units=d, num_blocks=blocks, num_heads=heads, src_vocab_size=V, dst_vocab_size=V, scale_embedding=True,
              residual_dropout_rate=0.0, train=Cfg(shared_embedding=False), generator=Cfg(learning_rate=0.0))
    x, y = synthetic.synthetic_batch(B, T, V, V, seed=3)
    y[1, 20] = 3
    x[2, 30:] = 0
    x[2, 29] = 3
    y, mask = N.deal_generated_samples(y)
    reward = ((np.random.default_rng(3).random((B, T)).astype(np.float32) - np.float32(0.4)) * mask).astype(np.float32)

    def loss_at(W, want_grads=False):
        m = N.Model(cfg, precision="fp32")
        m.build_generate(T)
        m.init_and_restore(weights=W)
        out = m.generate_step_and_update(x, y, reward, return_grads=want_grads)
        if want_grads:
            after = m.get_weights()
            assert all(np.array_equal(after[k], np.asarray(W[k], np.float32).reshape(after[k].shape)) for k in after)
        return out

    L0, g = loss_at(Wg, True)
    norm = float(np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values())))
    assert np.isfinite(L0) and norm > 0
    eps = 0.02 * abs(L0) / norm
    plus = {k: (np.asarray(Wg[k], np.float64) + eps * g[k].reshape(np.asarray(Wg[k]).shape) / norm).astype(np.float32) for k in g}
    minus = {k: (np.asarray(Wg[k], np.float64) - eps * g[k].reshape(np.asarray(Wg[k]).shape) / norm).astype(np.float32) for k in g}
    slope = (loss_at(plus) - loss_at(minus)) / (2 * eps)
    assert abs(slope - norm) < 2e-2 * norm, (slope, norm, L0, eps)


_TWO_RANK_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
z = np.load(sys.argv[4], allow_pickle=False)
from helpers import Cfg, facade_config, tiny_cfgs
import nmt_gan_amd as N
gcfg, _ = tiny_cfgs(T=int(z["T"]))
cfg = facade_config(gcfg); cfg["generator"] = Cfg(learning_rate=float(z["lr"]))
m = N.Model(cfg, precision="fp32")
m.build_generate(int(z["T"]))
m.init_and_restore(weights={k[2:]: z[k] for k in z.files if k.startswith("W:")})
loss = m.generate_step_and_update(z["x"], z["y"], z["reward"])        # every rank passes the WHOLE batch
np.savez(sys.argv[5] + ".%d.npz" % dist.get_rank(), loss=np.float64(loss), **m.get_weights())
dist.barrier(); dist.destroy_process_group()
print("rank ok")
'''


def test_two_rank_update_is_the_tower_average(cuda, tmp_path):
    """N > 1 path of the update (2 processes, gloo, both on cuda:0; RCCL runs the same code with backend nccl):
    sentences split with split_tensor bounds, each rank's loss normalised by its own batch, gradients averaged
    (model.py:121-123, 178-187, 689-722).  Both ranks end with the same weights = the oracle's RMSProp step on the
    averaged fp64 gradients, and report the mean of the two tower losses."""
    import os
    import subprocess
    import sys
    from nmt_gan_amd.share_function import split_bounds
    lr = 1e-2
    gcfg, _, Wg, x, y, mask, reward = _case(reward_scale=50.0)
    bounds = split_bounds(x.shape[0], 2)
    parts = [O.pg_loss_and_grads(Wg, gcfg, x[bounds[r]:bounds[r + 1]], y[bounds[r]:bounds[r + 1]],
                                 reward[bounds[r]:bounds[r + 1]]) for r in range(2)]
    W = {k: np.array(v, np.float64) for k, v in Wg.items()}
    G = {k: 0.5 * (parts[0][1][k] + parts[1][1][k]) for k in W}
    O.rmsprop_apply(W, G, {k: np.ones_like(v) for k, v in W.items()}, lr)
    inp = str(tmp_path / "in.npz")
    np.savez(inp, x=x, y=y, reward=reward, T=y.shape[1], lr=lr, **{"W:" + k: v for k, v in Wg.items()})
    script = tmp_path / "worker.py"
    script.write_text(_TWO_RANK_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), root, port, str(r), inp, str(tmp_path / "out")],
                              env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "rank ok" in o, o
    got = [dict(np.load(str(tmp_path / "out") + ".%d.npz" % r, allow_pickle=False)) for r in range(2)]
    want_loss = 0.5 * (parts[0][0] + parts[1][0])
    for r in range(2):
        assert abs(float(got[r]["loss"]) - want_loss) <= 1e-5 * abs(want_loss)
    for k in W:
        assert np.array_equal(got[0][k], got[1][k]), k
        d_ref = W[k] - np.asarray(Wg[k], np.float64)
        d_got = got[0][k].astype(np.float64) - np.asarray(Wg[k], np.float64)
        tol = 2e-3 * np.abs(d_ref).max() + 2 * np.finfo(np.float32).eps * np.abs(Wg[k]).max()
        assert np.abs(d_got - d_ref).max() <= tol, (k, np.abs(d_got - d_ref).max(), tol)


def test_production_size_bf16_loss_equals_the_teacher_forced_probabilities(cuda):
    """C2 generator (d=512, 6 blocks, 8 heads, V=32000, T=50), bf16 tier, 8 sentences: the loss the update step
    reports equals -sum p(y_t) reward_t / B recomputed on the host from nmtgan_teacher_forced_logits (the parity-tested
    forward of the rollout path) within 2e-3, the step leaves finite weights, and a second step with the same batch
    lowers the loss (rewards positive: the update raises p(y); learning rate and rewards are scaled up so that the
    step clears the bf16 resolution of the operand copies)."""
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    d, blocks, heads, V, T, B = 512, 6, 8, 32000, 50, 8
    Wg = synthetic.generator_weights(d, blocks, V, V, seed=0)
    cfg = Cfg(hidden_units=d, num_blocks=blocks, num_heads=heads, src_vocab_size=V, dst_vocab_size=V, scale_embedding=True,
              residual_dropout_rate=0.0, train=Cfg(shared_embedding=False), generator=Cfg(learning_rate=5e-2))
    m = N.Model(cfg, precision="bf16")
    m.build_generate(T)
    m.init_and_restore(weights=Wg)
    x, y = synthetic.synthetic_batch(B, T, V, V, seed=5)
    y[3, 17] = 3
    y, mask = N.deal_generated_samples(y)
    reward = (np.random.default_rng(5).random((B, T)).astype(np.float32) * mask * np.float32(1000.0)).astype(np.float32)
    logits = m.teacher_forced_logits(x, y).astype(np.float64)
    p = np.exp(logits - logits.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    p_y = np.take_along_axis(p, y[..., None].astype(np.int64), -1)[..., 0]
    want = -(p_y * reward).sum() / B
    l0 = m.generate_step_and_update(x, y, reward)
    assert abs(l0 - want) <= 2e-3 * abs(want), (l0, want)
    l1 = m.generate_step_and_update(x, y, reward)
    assert l1 < l0 < 0
    assert all(np.isfinite(v).all() for v in m.get_weights().values())
