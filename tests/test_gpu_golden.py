"""GPU path against the committed golden vectors (tests/golden/tiny_v1.npz, produced by the fp64
oracle - see make_goldens.py).  fp32 tier: ids bit-exact, floats within 1e-5."""
import os

import numpy as np
import pytest

from helpers import disc_weights, make_gpu_pair, tiny_cfgs
from oracle import nmtgan_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_v1.npz")


def test_fp32_tier_reproduces_goldens(cuda):
    with np.load(GOLD, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    T, R = int(g["T"]), int(g["R"])
    gcfg, dcfg = tiny_cfgs(T=T)
    Wg, Wd = O.init_generator_weights(gcfg, int(g["gen_seed"])), disc_weights(dcfg, int(g["disc_seed"]))
    m, disc = make_gpu_pair(gcfg, dcfg, Wg, Wd, "fp32", T)
    r = m.get_reward(g["x"], g["xp"], g["y"], g["mask"], R, disc, max_len=T, bias_num=0.6, uniforms=g["u"],
                     return_details=True)
    plan, det = m.last_details["plan"], m.last_details
    assert np.array_equal(det["rollouts"][:plan.n_rows], g["rollouts"][plan.row_r, plan.row_g - 1, plan.row_b])
    assert np.abs(r - g["rewards_bias06"]).max() < 1e-5
    r2 = m.get_reward(g["x"], g["xp"], g["y"], g["mask"], R, disc, max_len=T, bias_num=None, uniforms=g["u"])
    assert np.abs(r2 - g["rewards_nobias"]).max() < 1e-5
    assert np.abs(m.teacher_forced_logits(g["x"], g["y"]) - g["tf_logits"]).max() < 1e-5 * np.abs(g["tf_logits"]).max()
    assert np.array_equal(m.generate_step(g["x"]), g["greedy"])
    assert np.abs(disc.run_ypred_for_auc(g["disc_x"], g["disc_xs"], 1.0) - g["disc_probs"]).max() < 1e-5


def test_sampler_reproduces_golden_ids(cuda):
    import ctypes as C
    import torch
    from nmt_gan_amd import _lib as L
    with np.load(GOLD, allow_pickle=False) as z:
        lg, u, ids = z["samp_logits"], z["samp_u"], z["samp_ids"]
    lgd, ud = torch.from_numpy(lg).to(cuda), torch.from_numpy(u).to(cuda)
    out = torch.empty(lg.shape[0], dtype=torch.int32, device=cuda)
    a = L.SampleArgs()
    a.logits, a.ld, a.V, a.M, a.u, a.out = lgd.data_ptr(), lg.shape[1], lg.shape[1], lg.shape[0], ud.data_ptr(), out.data_ptr()
    L.check(L.lib.nmtgan_sample_rows(C.byref(a), L.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ids)


def test_same_uniforms_same_rewards_any_chunking(cuda):
    """Chunk sizes of the logits/sampler and discriminator stages are performance knobs only."""
    with np.load(GOLD, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    T, R = int(g["T"]), int(g["R"])
    gcfg, dcfg = tiny_cfgs(T=T)
    Wg, Wd = O.init_generator_weights(gcfg, 0), disc_weights(dcfg, 1)
    m, disc = make_gpu_pair(gcfg, dcfg, Wg, Wd, "bf16", T)
    a = m.get_reward(g["x"], g["xp"], g["y"], g["mask"], R, disc, max_len=T, bias_num=0.6, uniforms=g["u"])
    m.logits_chunk_rows, m.disc_chunk_rows = 5, 3
    m._ws = None
    b = m.get_reward(g["x"], g["xp"], g["y"], g["mask"], R, disc, max_len=T, bias_num=0.6, uniforms=g["u"])
    assert np.array_equal(a, b)               # a row's score does not depend on which rows share its batch
    # and a sentence-sharded evaluation (what each rank of an N-GPU job computes) gives the same rows
    import torch
    u = torch.from_numpy(g["u"]).to(cuda)
    B = g["x"].shape[0]
    parts = []
    for b0, b1 in ((0, 1), (1, B)):
        parts.append(m._reward_local(g["x"][b0:b1], g["xp"][b0:b1], g["y"][b0:b1], g["mask"][b0:b1], R, disc, T, 0.6,
                                     u, b_offset=b0, B_global=B, return_details=False).cpu().numpy())
    assert np.array_equal(np.concatenate(parts), a)


def test_restore_from_tf_checkpoints_reproduces_goldens(cuda, tmp_path):
    """Row f-2: weights arrive through TF V2 bundles (Model.init_and_restore(modelFile) / latest checkpoint of
    config.train.logdir, DisCNN(reload=True, saveto=...)) instead of dicts; the bundle also carries variables
    the path does not use (optimizer slots), which the restore must skip."""
    import nmt_gan_amd as N
    from helpers import Cfg, facade_config
    from nmt_gan_amd import tf_checkpoint as ck
    with np.load(GOLD, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    T, R = int(g["T"]), int(g["R"])
    gcfg, dcfg = tiny_cfgs(T=T)
    Wg, Wd = O.init_generator_weights(gcfg, int(g["gen_seed"])), disc_weights(dcfg, int(g["disc_seed"]))
    extra = {"global_step": np.int64(7), "embedding/softmax/kernel/RMSProp": np.ones((3, 3), np.float32)}
    gdir = tmp_path / "gen_logdir"
    ck.save_checkpoint(str(gdir / "model_epoch_0_step_7"), dict(Wg, **extra))
    ck.save_checkpoint(str(tmp_path / "disc" / "discriminator"), dict(Wd, **{"discnn/global_step": np.int64(3)}))
    cfg = facade_config(gcfg)
    cfg["train"] = Cfg(shared_embedding=gcfg.shared_embedding, logdir=str(gdir))
    rewards = []
    for model_file in (str(gdir / "model_epoch_0_step_7"), None):       # explicit prefix, then latest_checkpoint(logdir)
        m = N.Model(cfg, precision="fp32")
        m.build_train_model()
        m.build_generate(T)
        m.build_rollout_generate(T)
        m.init_and_restore(modelFile=model_file)
        disc = N.DisCNN(None, T, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
                        vocab_size_s=dcfg.vocab_size_s, precision="fp32", reload=True,
                        saveto=str(tmp_path / "disc" / "discriminator"))
        rewards.append(m.get_reward(g["x"], g["xp"], g["y"], g["mask"], R, disc, max_len=T, bias_num=0.6,
                                    uniforms=g["u"]))
        assert np.array_equal(m.generate_step(g["x"]), g["greedy"])
    assert np.abs(rewards[0] - g["rewards_bias06"]).max() < 1e-5
    assert np.array_equal(rewards[0], rewards[1])
    with pytest.raises(Exception):
        N.DisCNN(None, T, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
                 vocab_size_s=dcfg.vocab_size_s, reload=True, saveto=str(tmp_path / "missing"))
