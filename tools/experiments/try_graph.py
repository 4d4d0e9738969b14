"""Experiment: one get_reward (C2) as plain stream launches vs one HIP-graph replay."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import nmt_gan_amd as N
from nmt_gan_amd import synthetic
import bench

cfg = dict(bench.CONFIGS["c2"])
d, T, V, R, B = cfg["d"], cfg["T"], cfg["V"], cfg["rollnum"], cfg["batch"]
Wg = synthetic.generator_weights(d, cfg["blocks"], V, V, seed=0)
Wd = synthetic.discriminator_weights(T, cfg["E"], V, V, seed=1)
widths, nfs = synthetic.disc_filters(T)
mcfg = bench.Cfg(hidden_units=d, num_blocks=cfg["blocks"], num_heads=cfg["heads"], src_vocab_size=V, dst_vocab_size=V,
                 scale_embedding=True, residual_dropout_rate=0.0, train=bench.Cfg(shared_embedding=False))
model = N.Model(mcfg, precision="bf16")
model.build_generate(T); model.build_rollout_generate(T); model.init_and_restore(weights=Wg)
disc = N.DisCNN(None, T, 2, V, B, cfg["E"], widths, nfs, vocab_size_s=V, precision="bf16")
disc.load_weights(Wd)
x, y = synthetic.synthetic_batch(B, T, V, V, seed=1234)
y, mask = N.deal_generated_samples(y)
xp = N.extend_sentence_to_maxlen(x, T)
u = model.make_uniforms(R, T, B, seed=2024)
for overlap in (True, False):
    model.overlap = overlap
    prep = model.prepare_reward(x, xp, y, mask, R, disc, T, 0.6, u)
    ref = prep.launch().clone(); torch.cuda.synchronize()
    def timeit(fn, n=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    plain = timeit(prep.launch)
    t0 = time.perf_counter(); prep.capture(); torch.cuda.synchronize(); cap = time.perf_counter() - t0
    graph = timeit(prep.replay)
    same = bool(torch.equal(prep.replay(), ref)); torch.cuda.synchronize()
    print("overlap=%d plain %.1f ms  graph %.1f ms  (capture %.2f s)  identical=%s" % (overlap, plain, graph, cap, same), flush=True)
