#!/usr/bin/env python
"""Times DisCNN.train_step (row f-4) at the reference geometry (T=50, E=512, 13 widths, F=4550) on a synthetic batch.

    python tools/bench_disc_train.py [--batch 40] [--steps 5] [--precision bf16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=40)           # config_gan_train.yaml: dis_batch_size 40
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    import torch
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    T, E, V, B = 50, 512, 32000, a.batch
    widths, nfs = synthetic.disc_filters(T)
    d = N.DisCNN(None, T, 2, V, B, E, widths, nfs, vocab_size_s=V, precision=a.precision, optimizer='rmsprop')
    d.load_weights(synthetic.discriminator_weights(T, E, V, V, seed=1))
    rng = np.random.default_rng(0)
    x = rng.integers(4, V, (T, B)).astype(np.int32)
    xs = rng.integers(4, V, (T, B)).astype(np.int32)
    y = np.zeros((2, B), np.float32)
    y[rng.integers(0, 2, B), np.arange(B)] = 1.0
    out = [d.train_step(x, y, xs)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out.append(d.train_step(x, y, xs))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    conv = sum(2 * (T - w + 1) * w * E * n for w, n in zip(widths, nfs))
    F = sum(nfs)
    flop = 3 * 2 * B * (conv + 4 * F * F)                       # forward + dW + dX, both towers
    print(json.dumps({"what": "DisCNN.train_step (forward, backward, clip, RMSProp, refold), facade-inclusive",
                      "precision": a.precision, "batch": B, "ms_per_step": round(ms, 2),
                      "examples_per_s": round(B / (ms * 1e-3), 1), "gemm_TFLOP_per_step": round(flop / 1e12, 3),
                      "gemm_TFLOPs_achieved": round(flop / (ms * 1e-3) / 1e12, 1), "params": int(d._train["n"]),
                      "loss_acc": [[round(float(v), 5) for v in o] for o in out[:4]]}))


if __name__ == "__main__":
    main()
