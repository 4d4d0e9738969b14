"""Diagnostic for the discriminator's partial-window reuse at C2 size (run on a GPU box: python tools/check_conv_reuse.py
[bf16|fp32]): P(real) of every rollout with the reuse on/off on both schedules, the rows that differ most checked
against the fp64 CPU oracle, and the tier's accuracy on 48 random rows.  Test infrastructure (imports oracle/)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import nmt_gan_amd as N
from nmt_gan_amd import synthetic
from nmt_gan_amd.utils import AttrDict as Cfg
D, BLOCKS, HEADS, V, E, T, R, B = 512, 6, 8, 32000, 512, 50, 16, 64
Wg = synthetic.generator_weights(D, BLOCKS, V, V, seed=0)
Wd = synthetic.discriminator_weights(T, E, V, V, seed=1)
widths, nfs = synthetic.disc_filters(T)
mcfg = Cfg(hidden_units=D, num_blocks=BLOCKS, num_heads=HEADS, src_vocab_size=V, dst_vocab_size=V,
           scale_embedding=True, residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = N.Model(mcfg, precision=prec); model.build_train_model(); model.build_generate(T); model.build_rollout_generate(T)
model.init_and_restore(weights=Wg)
disc = N.DisCNN(None, T, 2, V, B, E, widths, nfs, vocab_size_s=V, precision=prec); disc.load_weights(Wd)
x, y = synthetic.synthetic_batch(B, T, V, V, seed=1234)
y[3, 20] = 3; y[7, 2] = 3
y, mask = N.deal_generated_samples(y); xp = N.extend_sentence_to_maxlen(x, T)
u = model.make_uniforms(R, T, B, seed=2024)
res = {}
for name, ov, bk in (("serial_off", False, -1), ("serial_on", False, 0), ("two_on", True, 0), ("two_off", True, -1)):
    model.overlap, model.conv_bucket_taps = ov, bk
    r = model.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u, return_details=True)
    res[name] = (r, model.last_details["p_rows"].copy(), model.last_details["rollouts"].copy())
plan = model.last_details["plan"]
base = res["serial_off"]
for k, v in res.items():
    dp = np.abs(v[1] - base[1])
    print(k, "rollouts equal", np.array_equal(v[2], base[2]), "max dP", dp.max(), "mean dP", dp.mean(), "frac>1e-5", (dp > 1e-5).mean())
    if dp.max() > 1e-5:
        i = int(dp.argmax()); print("   worst row", i, "b", plan.row_b[i], "g", plan.row_g[i], v[1][i], base[1][i])
        bad = np.where(dp > 1e-4)[0]; print("   g of bad rows:", np.bincount(plan.row_g[bad], minlength=T)[:T])

# oracle check of the worst rows (fp64 CPU discriminator)
from oracle import nmtgan_oracle as O
on, off = res["serial_on"], res["serial_off"]
dp = np.abs(on[1] - off[1])
idx = np.argsort(-dp)[:6]
dcfg = O.DisConfig(max_len=T, dim_word=E, vocab_size=V, vocab_size_s=V)
Wd64 = O.to_torch(Wd)
pref = O.disc_forward(Wd64, dcfg, off[2][idx].T, xp[plan.row_b[idx]].T).numpy()[:, 1]
for k, i in enumerate(idx):
    print("row", i, "b", plan.row_b[i], "g", plan.row_g[i], "oracle %.6f off %.6f on %.6f" % (pref[k], off[1][i], on[1][i]))

# accuracy of the tier on random rows
rng = np.random.default_rng(0)
ridx = rng.choice(plan.n_rows, 48, replace=False)
pr = O.disc_forward(Wd64, dcfg, off[2][ridx].T, xp[plan.row_b[ridx]].T).numpy()[:, 1]
for nm in ("serial_off", "serial_on"):
    e = np.abs(res[nm][1][ridx] - pr)
    print("acc", prec, nm, "48 random rows |P - oracle|: mean %.3g median %.3g max %.3g" % (e.mean(), np.median(e), e.max()))
