"""Regenerates tests/golden/tiny_v1.npz from the fp64 CPU oracle (oracle/nmtgan_oracle.py).

The reference ships no tests, fixtures or golden vectors and cannot be run here (Python 2 +
TensorFlow 1.2), so these vectors pin the ORACLE against regressions and give the
GPU tests a fixed target; they are NOT outputs of the reference ("parity unpinned").
Weights are not stored: they are regenerated from the recorded seeds by oracle.init_*_weights
(numpy PCG64 streams are stable across numpy versions).

    python tests/golden/make_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import disc_weights, tiny_cfgs  # noqa: E402
from oracle import nmtgan_oracle as O  # noqa: E402

T, B, R = 8, 4, 2
GEN_SEED, DISC_SEED = 0, 1


def main():
    gcfg, dcfg = tiny_cfgs(T=T)
    W = O.to_torch(O.init_generator_weights(gcfg, GEN_SEED))
    Wd = O.to_torch(disc_weights(dcfg, DISC_SEED))
    x, y = O.synthetic_batch(B, T, gcfg.src_vocab_size, gcfg.dst_vocab_size, seed=77, eos_rows=(1,), short_src_rows=(2,))
    yd, mask = O.deal_generated_samples(y)
    xp = O.extend_sentence_to_maxlen(x, T)
    u = np.random.default_rng(4321).random((R, T - 1, B, T), dtype=np.float32)
    out = dict(T=T, B=B, R=R, gen_seed=GEN_SEED, disc_seed=DISC_SEED, x=x, y_raw=y, y=yd, mask=mask, xp=xp, u=u)
    r_full, rolls = O.get_reward(W, gcfg, Wd, dcfg, x, xp, yd, mask, R, T, 0.6, u, mode="full", return_rollouts=True)
    r_inc, rolls_inc = O.get_reward(W, gcfg, Wd, dcfg, x, xp, yd, mask, R, T, 0.6, u, mode="incremental",
                                    return_rollouts=True)
    assert np.array_equal(rolls, rolls_inc) and np.abs(r_full - r_inc).max() < 1e-6
    out["rewards_bias06"] = r_full
    out["rollouts"] = rolls
    out["rewards_nobias"] = O.get_reward(W, gcfg, Wd, dcfg, x, xp, yd, mask, R, T, None, u, mode="incremental")
    out["tf_logits"] = O.teacher_forced_logits(W, gcfg, torch.from_numpy(x).long(),
                                               torch.from_numpy(yd).long()).numpy().astype(np.float32)
    out["greedy"] = O.generate_step(W, gcfg, torch.from_numpy(x).long(), T)
    rng = np.random.default_rng(5)
    dx = rng.integers(0, dcfg.vocab_size, (T, 9)).astype(np.int32)
    dxs = rng.integers(0, dcfg.vocab_size_s, (T, 9)).astype(np.int32)
    out["disc_x"], out["disc_xs"] = dx, dxs
    out["disc_probs"] = O.disc_forward(Wd, dcfg, dx, dxs).numpy()
    # sampler known-answer vectors (incl. a dominant logit, a flat row, -inf entries, u = 0 and u -> 1)
    V = 4096
    lg = (rng.standard_normal((8, V)) * np.array([0.5, 1, 2, 4, 8, 1, 1, 1])[:, None]).astype(np.float32)
    lg[5, 1234] = 40.0
    lg[6, :] = 0.25
    lg[7, :2048] = -np.inf
    su = rng.random(8, dtype=np.float32)
    su[0], su[1] = 0.0, np.float32(1.0 - 2 ** -24)
    out["samp_logits"], out["samp_u"] = lg, su
    out["samp_ids"] = np.array([O.multinomial_cdf(lg[i], su[i]) for i in range(8)], np.int32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
