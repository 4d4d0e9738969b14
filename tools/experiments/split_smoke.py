import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from helpers import make_gpu_pair, tiny_cfgs, disc_weights
from oracle import nmtgan_oracle as O
import nmt_gan_amd as N
T, B, R = 10, 4, 2
gcfg, dcfg = tiny_cfgs(T=T)
Wg, Wd = O.init_generator_weights(gcfg, 0), disc_weights(dcfg, 1)
x, yd = O.synthetic_batch(B, T, gcfg.src_vocab_size, gcfg.dst_vocab_size, seed=4, short_src_rows=(1,))
y, mask = N.deal_generated_samples(yd)
xp = N.extend_sentence_to_maxlen(x, T)
u = np.random.default_rng(0).random((R, T - 1, B, T), dtype=np.float32)
ref = O.get_reward(O.to_torch(Wg), gcfg, O.to_torch(Wd), dcfg, x, xp, y, mask, R, T, 0.6, u, mode="incremental")
lg_ref = O.teacher_forced_logits(O.to_torch(Wg), gcfg, torch.from_numpy(x).long(), torch.from_numpy(y).long()).numpy()
for prec in ("fp32", "fp16x3", "fp16"):
    m, d = make_gpu_pair(gcfg, dcfg, Wg, Wd, prec, T)
    lg = m.teacher_forced_logits(x, y)
    got = m.get_reward(x, xp, y, mask, R, d, max_len=T, bias_num=0.6, uniforms=u)
    gen = m.generate_step(x)
    print(prec, "logits rel err %.3g" % (np.abs(lg - lg_ref).max() / np.abs(lg_ref).max()), "reward max diff %.3g" % np.abs(got - ref).max(), gen[0][:6], flush=True)
