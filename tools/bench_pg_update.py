#!/usr/bin/env python
"""Times Model.generate_step_and_update (row f-3: teacher-forced forward, policy-gradient loss, backward,
RMSProp) on the C2 generator with a synthetic batch, beside one get_reward of the same batch.

    python tools/bench_pg_update.py [--batch 64] [--steps 5] [--precision bf16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Cfg(dict):
    __getattr__ = dict.get


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--V", type=int, default=32000)
    a = ap.parse_args()
    import torch
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    d, T, V, B, Nb = a.d, a.T, a.V, a.batch, a.blocks
    Wg = synthetic.generator_weights(d, Nb, V, V, seed=0)
    cfg = Cfg(hidden_units=d, num_blocks=Nb, num_heads=a.heads, src_vocab_size=V, dst_vocab_size=V, scale_embedding=True,
              residual_dropout_rate=0.0, train=Cfg(shared_embedding=False), generator=Cfg(learning_rate=1e-5))
    m = N.Model(cfg, precision=a.precision)
    m.build_train_model()
    m.build_generate(T)
    m.build_rollout_generate(T)
    m.init_and_restore(weights=Wg)
    x, y = synthetic.synthetic_batch(B, T, V, V, seed=1234)
    y, mask = N.deal_generated_samples(y)
    reward = ((np.random.default_rng(0).random((B, T)).astype(np.float32) - np.float32(0.6)) * mask).astype(np.float32)
    losses = [m.generate_step_and_update(x, y, reward)]          # warm-up (allocates the flat buffers + workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(m.generate_step_and_update(x, y, reward))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    rows = B * T + B * x.shape[1]
    # forward GEMM flops: decoder 28 d^2 N per row + logits 2 d V, encoder 24 d^2 N per row (+ cross K/V 4 d^2 N);
    # backward = 2x (dX and dW)
    fwd = (28 * d * d * Nb + 2 * d * V) * B * T + (24 * d * d * Nb + 4 * d * d * Nb) * B * x.shape[1]
    print(json.dumps({"what": "generate_step_and_update (forward + backward + RMSProp), facade-inclusive",
                      "precision": a.precision, "batch": B, "T": T, "rows": rows, "ms_per_update": round(ms, 2),
                      "sentences_per_s": round(B / (ms * 1e-3), 1), "gemm_TFLOP_per_update": round(3 * fwd / 1e12, 3),
                      "gemm_TFLOPs_achieved": round(3 * fwd / (ms * 1e-3) / 1e12, 1),
                      "params": int(m._train["n"]), "losses": [round(float(l), 8) for l in losses[:3]]}))


if __name__ == "__main__":
    main()
