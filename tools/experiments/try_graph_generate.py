"""Experiment: greedy generate_step as plain launches vs a HIP-graph replay (C2 generator)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nmt_gan_amd as N
from nmt_gan_amd import synthetic, _lib
from nmt_gan_amd._lib import lib, ptr, check
class Cfg(dict): __getattr__ = dict.get
d, T, V, Nb = 512, 50, 32000, 6
cfg = Cfg(hidden_units=d, num_blocks=Nb, num_heads=8, src_vocab_size=V, dst_vocab_size=V, scale_embedding=True,
          residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
m = N.Model(cfg, precision="bf16"); m.build_generate(T)
m.init_and_restore(weights=synthetic.generator_weights(d, Nb, V, V, seed=0))
for B in (64, 512):
    x, _ = synthetic.synthetic_batch(B, T, V, V, seed=1234)
    ref = m.generate_step(x)
    xd = m._dev_i32(x); pad = (xd == 0).to(torch.uint8).contiguous()
    y = torch.empty((B, T), dtype=torch.int32, device=m.device)
    g = m.c_struct
    need = lib.nmtgan_generate_workspace_bytes(C.byref(g), B, x.shape[1], T)
    ws = torch.empty(need, dtype=torch.uint8, device=m.device)
    def launch():
        check(lib.nmtgan_generate_greedy(C.byref(g), ptr(xd), ptr(pad), B, x.shape[1], T, ptr(y), ptr(ws), ws.numel(), _lib.stream_ptr()))
    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
    plain = timeit(launch)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="relaxed"):
        launch()
    graph = timeit(gr.replay)
    y.zero_(); gr.replay(); torch.cuda.synchronize()
    print("B=%d plain %.1f ms  graph %.1f ms  identical=%s" % (B, plain, graph, np.array_equal(y.cpu().numpy(), ref)), flush=True)
