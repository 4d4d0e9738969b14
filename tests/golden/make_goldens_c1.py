"""Regenerates tests/golden/c1_v1.npz: BASELINE.json configs[0] (transformer-small d256 N2 h4 + CNN discriminator,
batch 8, rollnum 4, len 30, vocabulary 32000) from the fp64 CPU oracle (oracle/nmtgan_oracle.py), incremental mode
(identical to the structure-faithful mode: tests/test_oracle_cpu.py).  Like tiny_v1.npz these vectors are outputs of
this repository's oracle, not of the reference, which cannot run here ("parity unpinned").  Weights are regenerated
from the recorded seeds.  Takes a few minutes on 8 cores.

    python tests/golden/make_goldens_c1.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import disc_weights, tiny_cfgs  # noqa: E402
from oracle import nmtgan_oracle as O  # noqa: E402

D, BLOCKS, HEADS, V, E, T, B, R = 256, 2, 4, 32000, 256, 30, 8, 4
GEN_SEED, DISC_SEED = 0, 1


def main():
    gcfg, dcfg = tiny_cfgs(d=D, blocks=BLOCKS, heads=HEADS, Vs=V, Vt=V, T=T, E=E)
    W = O.to_torch(O.init_generator_weights(gcfg, GEN_SEED))
    Wd = O.to_torch(disc_weights(dcfg, DISC_SEED))
    x, y = O.synthetic_batch(B, T, V, V, seed=31, eos_rows=(2, 5))
    yd, mask = O.deal_generated_samples(y)
    xp = O.extend_sentence_to_maxlen(x, T)
    u = np.random.default_rng(977).random((R, T - 1, B, T), dtype=np.float32)
    t0 = time.time()
    rewards, rolls = O.get_reward(W, gcfg, Wd, dcfg, x, xp, yd, mask, R, T, 0.6, u, mode="incremental",
                                  return_rollouts=True)
    print("oracle get_reward: %.1f s" % (time.time() - t0))
    greedy = O.generate_step(W, gcfg, torch.from_numpy(x).long(), T)
    out = dict(D=D, BLOCKS=BLOCKS, HEADS=HEADS, V=V, E=E, T=T, B=B, R=R, gen_seed=GEN_SEED, disc_seed=DISC_SEED,
               x=x, y=yd, mask=mask, xp=xp, u=u, rewards_bias06=rewards.astype(np.float32),
               rollouts=rolls.astype(np.int32), greedy=greedy.astype(np.int32))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
