"""One get_reward as K sentence sub-batches on their own streams (each with its own workspace and auxiliary stream),
enqueued from K host threads, against the unsplit call.  usage: python tools/experiments/try_split.py [B] [K]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nmt_gan_amd as N
from nmt_gan_amd import synthetic
from nmt_gan_amd.share_function import split_bounds
from bench import CONFIGS, Cfg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
cfg = dict(CONFIGS["c2"])
d, T, V, R = cfg["d"], cfg["T"], cfg["V"], cfg["rollnum"]
Wg = synthetic.generator_weights(d, cfg["blocks"], V, V, seed=0)
Wd = synthetic.discriminator_weights(T, cfg["E"], V, V, seed=1)
widths, nfs = synthetic.disc_filters(T)
mcfg = Cfg(hidden_units=d, num_blocks=cfg["blocks"], num_heads=cfg["heads"], src_vocab_size=V, dst_vocab_size=V,
           scale_embedding=True, residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
model = N.Model(mcfg, precision="fp16", device=str(dev))
model.build_train_model(); model.build_generate(T); model.build_rollout_generate(T)
model.init_and_restore(weights=Wg)
disc = N.DisCNN(None, T, 2, V, B, cfg["E"], widths, nfs, vocab_size_s=V, precision="fp16", device=str(dev))
disc.load_weights(Wd)
x, y = synthetic.synthetic_batch(B, T, V, V, seed=1234)
y, mask = N.deal_generated_samples(y)
xp = N.extend_sentence_to_maxlen(x, T)
u = model.make_uniforms(R, T, B, seed=2024)

MODE = os.environ.get("SPLIT_MODE", "both")          # whole | parts | both: a HIP stream keeps the hardware queue it was
whole = model.prepare_reward(x, xp, y, mask, R, disc, T, 0.6, u) if MODE != "parts" else None   # created on (4 per process)
bounds = split_bounds(B, K)
parts = []
for i in range(K):
    b0, b1 = bounds[i], bounds[i + 1]
    st = None if i == 0 else torch.cuda.Stream(device=dev)     # the first sub-batch on the default stream (its own queue)
    parts.append(model.prepare_reward(x[b0:b1], xp[b0:b1], y[b0:b1], mask[b0:b1], R, disc, T, 0.6, u, b_offset=b0, B_global=B,
                                      own_workspace=True, stream=st))


def run_whole():
    return whole.launch()


def run_parts():
    th = [threading.Thread(target=p.launch) for p in parts]
    for t in th: t.start()
    for t in th: t.join()


def timed(fn, n=3):
    fn(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3


if MODE == "parts":
    print("B=%d: %d sub-batches on their own streams %.1f ms" % (B, K, timed(run_parts)), flush=True)
elif MODE == "whole":
    print("B=%d: whole %.1f ms" % (B, timed(run_whole)), flush=True)
else:
    t_whole = timed(run_whole)
    ref = whole.tensors["rewards"].clone()
    t_parts = timed(run_parts)
    got = torch.cat([p.tensors["rewards"] for p in parts])
    print("B=%d: whole %.1f ms, %d sub-batches on their own streams %.1f ms; identical rewards: %s"
          % (B, t_whole, K, t_parts, bool(torch.equal(ref, got))), flush=True)
