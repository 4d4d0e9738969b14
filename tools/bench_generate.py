#!/usr/bin/env python
"""Times Model.generate_step (row f-1: greedy KV-cached decode, model.py:140-176, 268-271) on the C2 generator.

    python tools/bench_generate.py [--batch 64] [--steps 5] [--precision bf16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Cfg(dict):
    __getattr__ = dict.get


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    import torch
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    d, T, V, B, Nb = 512, 50, 32000, a.batch, 6
    cfg = Cfg(hidden_units=d, num_blocks=Nb, num_heads=8, src_vocab_size=V, dst_vocab_size=V, scale_embedding=True,
              residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
    m = N.Model(cfg, precision=a.precision)
    m.build_generate(T)
    m.init_and_restore(weights=synthetic.generator_weights(d, Nb, V, V, seed=0))
    x, _ = synthetic.synthetic_batch(B, T, V, V, seed=1234)
    m.generate_step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = m.generate_step(x)
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print(json.dumps({"what": "generate_step (greedy, facade-inclusive: H2D ids, D2H result)", "precision": a.precision,
                      "batch": B, "T": T, "ms_per_call": round(ms, 2), "sentences_per_s": round(B / (ms * 1e-3), 1),
                      "tokens_per_s": round(B * T / (ms * 1e-3))}))


if __name__ == "__main__":
    main()
