#!/usr/bin/env python
"""One generator iteration of gan_train.py:87-118 plus a discriminator fine-tuning step (gan_train.py:147-161) on
synthetic data, phase by phase with wall-clock times - every phase runs on the MI355X path.

    python tools/gan_step.py [--config c2] [--iters 2] [--precision bf16]

    x, y_ground            synthetic batch
    y_sample               generator.generate_step(x)                                  gan_train.py:91
    y, mask                deal_generated_samples(y_sample)                            gan_train.py:93
    rewards                generator.get_reward(x, x_to_maxlen, y, mask, rollnum, D)   gan_train.py:102   <- the hot path
    loss                   generator.generate_step_and_update(x, y, rewards)           gan_train.py:112
    P(real) of both        D scoring of generated vs. reference batch (BASELINE config 5)
    D loss, accuracy       discriminator.train_step(generated + reference, labels)     cnn_discriminator.py:664
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
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    import torch
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    cfg = dict(bench.CONFIGS[a.config])
    if a.batch:
        cfg["batch"] = a.batch
    d, T, V, R, B = cfg["d"], cfg["T"], cfg["V"], cfg["rollnum"], cfg["batch"]
    mcfg = bench.Cfg(hidden_units=d, num_blocks=cfg["blocks"], num_heads=cfg["heads"], src_vocab_size=V, dst_vocab_size=V,
                     scale_embedding=True, residual_dropout_rate=0.0, train=bench.Cfg(shared_embedding=False),
                     generator=bench.Cfg(learning_rate=1e-5))
    gen = N.Model(mcfg, precision=a.precision)
    gen.build_train_model()
    gen.build_generate(T, optimizer='rmsprop')
    gen.build_rollout_generate(T)
    gen.init_and_restore(weights=synthetic.generator_weights(d, cfg["blocks"], V, V, seed=0))
    widths, nfs = synthetic.disc_filters(T)
    disc = N.DisCNN(None, T, 2, V, B, cfg["E"], widths, nfs, vocab_size_s=V, precision=a.precision, optimizer='rmsprop')
    disc.load_weights(synthetic.discriminator_weights(T, cfg["E"], V, V, seed=1))
    x, y_ground = synthetic.synthetic_batch(B, T, V, V, seed=1234)
    phases = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        phases.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))
        return out

    for it in range(a.iters):
        y_sample = timed("generate_step", lambda: gen.generate_step(x))
        y, mask = N.deal_generated_samples(y_sample)
        xp = N.extend_sentence_to_maxlen(x, T)
        rewards = timed("get_reward", lambda: gen.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, seed=it))
        g_loss = timed("generate_step_and_update", lambda: gen.generate_step_and_update(x, y, rewards))
        yg, _ = N.deal_generated_samples(y_ground)
        p_fake = timed("score_generated", lambda: disc.run_ypred_for_auc(y.T, xp.T, 1.0))
        p_real = timed("score_reference", lambda: disc.run_ypred_for_auc(yg.T, xp.T, 1.0))
        dx = np.concatenate([y, yg], axis=0).T                       # [T, 2B]: negatives then positives
        dxs = np.concatenate([xp, xp], axis=0).T
        lab = np.zeros((2, 2 * B), np.float32)
        lab[0, :B] = 1.0                                             # [1, 0] = generated, [0, 1] = human (column 1 = P(real))
        lab[1, B:] = 1.0
        d_loss, d_acc = timed("disc_train_step", lambda: disc.train_step(dx, lab, dxs))
        print("iter %d: g_loss %.6g  mean reward %.4f  P(real) generated %.4f reference %.4f  d_loss %.4f d_acc %.3f"
              % (it, g_loss, float(rewards.mean()), float(p_fake[:, 1].mean()), float(p_real[:, 1].mean()), d_loss, d_acc),
              flush=True)
    last = {k: round(v[-1], 2) for k, v in phases.items()}
    print(json.dumps({"what": "one GAN iteration, facade-inclusive wall-clock per phase (last iteration)", "config": cfg["name"],
                      "precision": a.precision, "ms": last, "total_ms": round(sum(last.values()), 1)}))


if __name__ == "__main__":
    main()
