"""Accuracy of the 16-bit tiers where the benchmark runs: BASELINE.json configs[1] (d512 N6 h8, V = 32000,
T = 50, rollnum 16), measured against the fp32 tier, which is itself checked here against the fp64 oracle at the
real vocabulary (and by the tiny-config tests at 1e-5 / identical tokens).

For each 16-bit tier and test vehicle: one get_reward on 8 sentences, then 512 of ITS rollouts are replayed
teacher-forced through the fp32 tier and through the tier itself (forced-identical prefixes), and
  * logits: tensor-scale relative error ||tier - fp32||_inf / ||fp32||_inf over the 512 x 50 x 32000 logits;
  * tokens: every sampled position of those rollouts is re-drawn from the fp32 logits with the SAME uniform
    (fp64 CDF in index order, first index above u * total - the oracle's multinomial_cdf on the device) and
    compared with the token the tier drew; every flip must be a STATED NEAR-TIE: u lies within 2 max|dlogit| of the
    normalised CDF boundary between the two tokens (a logit perturbation of size e moves a normalised CDF
    boundary by at most exp(2e) - 1);
  * P(real): the tier's discriminator scores of its rollouts against the fp32 tier's scores of the same rollouts,
    and the [B, T] rewards rebuilt from both.
Two vehicles, because token agreement under CDF inversion is set by the DISTRIBUTION, not only by the logit error:
  "reference-init"  the reference's initialisers (softmax table N(0, 1/512): near-uniform over 32000 tokens, the
                    worst case - a boundary moves by ~e sqrt(k)/V against bucket widths of 1/V, so a relative logit
                    error of 1e-3, the north star's own tolerance, already flips ~9 % of the draws; discriminator with
                    random BatchNorm statistics, whose 2-way score has |s| ~ 100 and turns 3e-4 relative into 3e-2 of P);
  "trained-like"    softmax table x 5 (entropy of a trained NMT softmax, a few nats) and a discriminator whose
                    BatchNorm statistics match its own activations, as after training.
The bounds asserted are the measured values with margin; north_star asks 1e-3 relative and 99.5 % agreement."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nmtgan_oracle as O

pytestmark = pytest.mark.gpu

D, BLOCKS, HEADS, V, E, T, R, B = 512, 6, 8, 32000, 512, 50, 16, 8
N_REPLAY = 512
PEAK = 5.0

# measured on MI355X; asserted with margin.  Measured: fp16 logits 9.0e-4 (both vehicles), bf16 7.4e-3; token agreement
# fp16 90.7 % / 91.7 %, bf16 66.4 % / 81.4 % (reference-init / trained-like); max |dP(real)| fp16 2.7e-2 / 1.6e-4, bf16
# 2.0e-1 / 1.3e-3; rewards (inf-norm, relative) fp16 4.4e-3 / 4.7e-4, bf16 3.9e-2 / 5.2e-3 (conv GEMMs; with the table form
# of the rollouts' conv, the default at this size: max |dP(real)| fp16 2.9e-2 / 1.8e-4, bf16 1.9e-1 / 1.5e-3, rewards fp16
# 6.2e-3 / 5.7e-4, bf16 2.1e-2 / 4.7e-3; fp16x3 3.0e-4 / 1.6e-6 and 5.6e-5 / 5.1e-6).
# Against the north star (1e-3 logits and rewards, 99.5 % tokens): fp16 meets the logits bound on both vehicles and the
# reward bound with a discriminator whose BatchNorm statistics fit its activations; NO 16-bit tier reaches 99.5 % token
# agreement at V = 32000 - at a logit error of 1e-3 about one draw in eleven is a near-tie there - only the fp32 tier and
# the split-precision tier fp16x3 do (measured: logits 2.7e-6 / 2.5e-6, token agreement 100 % / 99.64 %; with the
# split-precision discriminator max |dP(real)| 2.4e-4 / 1.8e-6 and rewards 4e-5 / 7e-6 - with the fp16 discriminator it
# had 2.7e-2 / 1.7e-4 and 4.7e-3 / 4.7e-4, tools/error_budget_disc.py shows every stage of the fp16 form alone breaks 1e-3).
BOUNDS = {
    ("fp16", "reference-init"): dict(logits_rel=1.2e-3, agree=0.89, p_abs=4e-2, reward_rel=8e-3),
    ("bf16", "reference-init"): dict(logits_rel=1.0e-2, agree=0.62, p_abs=3e-1, reward_rel=6e-2),
    ("fp16", "trained-like"): dict(logits_rel=1.2e-3, agree=0.90, p_abs=4e-4, reward_rel=1e-3),
    ("bf16", "trained-like"): dict(logits_rel=1.0e-2, agree=0.79, p_abs=3e-3, reward_rel=1e-2),
    # split-precision generator AND discriminator (fp16 products over hi/lo operand splits, fp32 accumulation and
    # storage): every bound is the north star's - 1e-3 on logits, P(real) and rewards, 99.5 % of the tokens
    ("fp16x3", "reference-init"): dict(logits_rel=1.0e-5, agree=0.995, p_abs=1e-3, reward_rel=1e-3),
    ("fp16x3", "trained-like"): dict(logits_rel=1.0e-5, agree=0.995, p_abs=1e-3, reward_rel=1e-3),
}


@pytest.fixture(scope="module")
def world(cuda):
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    from nmt_gan_amd.utils import AttrDict as Cfg
    widths, nfs = synthetic.disc_filters(T)
    mcfg = Cfg(hidden_units=D, num_blocks=BLOCKS, num_heads=HEADS, src_vocab_size=V, dst_vocab_size=V,
               scale_embedding=True, residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
    Wg = {"reference-init": synthetic.generator_weights(D, BLOCKS, V, V, seed=0)}
    Wg["trained-like"] = dict(Wg["reference-init"])
    Wg["trained-like"]["embedding/softmax/kernel"] = Wg["reference-init"]["embedding/softmax/kernel"] * np.float32(PEAK)
    Wd = {"reference-init": synthetic.discriminator_weights(T, E, V, V, seed=1),
          "trained-like": synthetic.discriminator_weights(T, E, V, V, seed=1, calibrated=True)}
    cache = {}

    def pair(precision, vehicle):
        if (precision, vehicle) not in cache:
            if precision != "fp32":                        # one 16-bit pair alive at a time, its workspace (25-49 GiB
                for k in [k for k in cache if k[0] != "fp32"]:     # with the conv tables) returned before the next
                    del cache[k]
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            m = N.Model(mcfg, precision=precision)
            m.build_rollout_generate(T)
            m.init_and_restore(weights=Wg[vehicle])
            d = N.DisCNN(None, T, 2, V, B, E, widths, nfs, vocab_size_s=V, precision=precision)
            d.load_weights(Wd[vehicle])
            cache[(precision, vehicle)] = (m, d)
        return cache[(precision, vehicle)]

    x, y = synthetic.synthetic_batch(B, T, V, V, seed=77)
    y[2, 31] = 3                                          # one early </S>
    y, mask = N.deal_generated_samples(y)
    xp = N.extend_sentence_to_maxlen(x, T)
    u = torch.rand((R, T - 1, B, T), generator=torch.Generator().manual_seed(99)).numpy()
    return dict(N=N, pair=pair, Wg=Wg, Wd=Wd, x=x, y=y, mask=mask, xp=xp, u=u)


def _cdf(logits):
    l = logits.double()
    c = torch.cumsum(torch.exp(l - l.max(dim=1, keepdim=True).values), dim=1)
    return c


def _cdf_tokens(logits, u):
    """tf.multinomial's CPU semantics on the device: fp64 exp(l - max), running sum in index order, first index
    whose CDF exceeds u * total, clamped (restates oracle/nmtgan_oracle.py multinomial_cdf)."""
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    for i in range(0, logits.shape[0], 2048):
        c = _cdf(logits[i:i + 2048])
        tgt = u[i:i + 2048].double() * c[:, -1]
        out[i:i + 2048] = torch.searchsorted(c, tgt[:, None], right=True)[:, 0].clamp(max=logits.shape[1] - 1)
    return out


def test_fp32_tier_matches_fp64_oracle_at_full_vocabulary(world):
    """The yardstick of this file pinned at V = 32000: fp32-tier teacher-forced logits of 12 sentences against the
    fp64 oracle (1e-5 at tensor scale), and every position re-drawn from both with the same uniform."""
    w = world
    m32, _ = w["pair"]("fp32", "reference-init")
    rng = np.random.default_rng(11)
    n = 12
    x = w["x"][rng.integers(0, B, n)]
    y = rng.integers(4, V, (n, T)).astype(np.int32)
    got = m32.teacher_forced_logits(x, y, device_out=True)
    W64 = O.to_torch(w["Wg"]["reference-init"])
    gcfg = O.GenConfig(hidden_units=D, num_blocks=BLOCKS, num_heads=HEADS, src_vocab_size=V, dst_vocab_size=V,
                       scale_embedding=True, shared_embedding=False)
    ref = O.teacher_forced_logits(W64, gcfg, torch.from_numpy(x).long(), torch.from_numpy(y).long())
    ref_d = ref.to(got.device)
    rel = float((got.double() - ref_d).abs().max() / ref_d.abs().max())
    uu = torch.rand(n * T, generator=torch.Generator().manual_seed(5)).to(got.device)
    a = _cdf_tokens(got.reshape(n * T, V), uu)
    b = _cdf_tokens(ref_d.reshape(n * T, V), uu)
    agree = float((a == b).float().mean())
    print("fp32 tier vs fp64 oracle at V=32000: logits rel %.3g, token agreement %.4f of %d" % (rel, agree, n * T))
    assert rel < 1e-5, rel
    assert agree >= 0.995, agree
    # discriminator of the tier: P(real) of 16 of its own rollouts + the 8 samples against the fp64 oracle, on both
    # vehicles; the rewards are means of these P(real) minus a constant, so their error is bounded by it
    dcfg = O.DisConfig(max_len=T, dim_word=E, vocab_size=V, vocab_size_s=V)
    p_err, r_scale = 0.0, 1.0
    for vehicle in ("reference-init", "trained-like"):
        m32v, d32v = w["pair"]("fp32", vehicle)
        r32 = m32v.get_reward(w["x"], w["xp"], w["y"], w["mask"], R, d32v, max_len=T, bias_num=0.6, uniforms=w["u"],
                              return_details=True)
        det = m32v.last_details
        plan = det["plan"]
        idx = np.random.default_rng(4).choice(plan.n_rows, 16, replace=False)
        Wd64 = O.to_torch(w["Wd"][vehicle])
        p = O.disc_forward(Wd64, dcfg, det["rollouts"][idx].T, w["xp"][plan.row_b[idx]].T).numpy()[:, 1]
        pb = O.disc_forward(Wd64, dcfg, w["y"].T, w["xp"].T).numpy()[:, 1]
        e = max(float(np.abs(p - det["p_rows"][idx]).max()), float(np.abs(pb - det["p_base"]).max()))
        print("fp32 tier P(real) vs fp64 oracle (%s): %.3g" % (vehicle, e))
        if e / float(np.abs(r32).max()) > p_err / r_scale:
            p_err, r_scale = e, float(np.abs(r32).max())
    # measured 2.3e-5 (reference-init: a 2-way score of |s| ~ 100 carries the fp32 rounding of its own 25,000-term conv
    # sums, ~1e-7 |s| per feature - any fp32 evaluation, the reference's included, sits this far from fp64) / 1e-6
    assert p_err < 1e-4, p_err
    res = dict(precision="fp32", vehicle="fp64-oracle", logits_rel_inf=rel, token_agreement=agree, tokens_compared=n * T,
               p_real_abs_max=p_err, reward_rel_inf=p_err / r_scale,
               reward_note="bound: rewards are means of P(real) minus a constant, so |d reward| <= max |dP(real)|; "
                           "divided by the largest |reward| of the call.  Worst vehicle (reference-init, |score| ~ 100): "
                           "fp32 summation-order rounding of the score itself; 1e-6 on the trained-like vehicle",
               meets_north_star=dict(logits=rel <= 1e-5, tokens=agree >= 0.995, rewards=p_err / r_scale <= 1e-5),
               north_star=dict(logits_rel=1e-5, token_agreement=0.995, reward_rel=1e-5))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for out_dir in (os.path.join(root, "profiles"), os.path.join(root, "gpurun_out")):
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r3_precision_c2_fp32_fp64-oracle.json"), "w") as f:
            json.dump(res, f, indent=1)


@pytest.mark.parametrize("vehicle", ["reference-init", "trained-like"])
@pytest.mark.parametrize("precision", ["fp16", "bf16", "fp16x3"])
def test_tier_accuracy_at_benchmark_shape(world, precision, vehicle):
    w = world
    m32, d32 = w["pair"]("fp32", vehicle)
    m, d = w["pair"](precision, vehicle)
    dev = m.device
    r_tier = m.get_reward(w["x"], w["xp"], w["y"], w["mask"], R, d, max_len=T, bias_num=0.6, uniforms=w["u"],
                          return_details=True)
    det = m.last_details
    plan = det["plan"]
    rolls = det["rollouts"][:plan.n_rows]
    rng = np.random.default_rng(3)
    idx = np.sort(rng.choice(plan.n_rows, N_REPLAY, replace=False))
    rb, rr, rg = plan.row_b[idx], plan.row_r[idx], plan.row_g[idx]
    xr, yr = w["x"][rb], rolls[idx]
    ref = m32.teacher_forced_logits(xr, yr, device_out=True)                  # [512, T, V] fp32 tier
    got = m.teacher_forced_logits(xr, yr, device_out=True)
    scale = float(ref.abs().max())
    dmax = float((got - ref).abs().max())
    logits_rel = dmax / scale
    rms_rel = float((got - ref).pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
    del got
    # sampled positions of rollout (b, r, g): t = g .. len_b - 1, uniform u[r, g-1, b, t]
    lens = plan.lens[rb]
    tt = np.arange(T)[None, :]
    ii, ti = np.nonzero((tt >= rg[:, None]) & (tt < lens[:, None]))
    u_dev = torch.as_tensor(w["u"]).to(dev)
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a)).long().to(dev)
    uu = u_dev[dv(rr[ii]), dv(rg[ii] - 1), dv(rb[ii]), dv(ti)]
    rows = ref[dv(ii), dv(ti)]                                                # [n_tok, V] fp32-tier logits
    want = _cdf_tokens(rows, uu)
    have = dv(yr[ii, ti])
    flip = torch.nonzero(want != have)[:, 0]
    agree = 1.0 - flip.numel() / have.numel()
    # entropy of the sampling distribution (nats) and the near-tie margin of every flip
    p = torch.softmax(rows[:512].double(), dim=1)
    entropy = float(-(p * torch.log(p.clamp_min(1e-300))).sum(1).mean())
    worst_margin = 0.0
    for i in range(0, flip.numel(), 1024):
        f = flip[i:i + 1024]
        c = _cdf(rows[f])
        c = c / c[:, -1:]
        lo, hi = torch.minimum(want[f], have[f]), torch.maximum(want[f], have[f])
        # boundaries between the two tokens are c[lo .. hi-1]; u is nearest to one of the two outermost
        gap = torch.minimum((uu[f].double() - c.gather(1, lo[:, None])[:, 0]).abs(),
                            (uu[f].double() - c.gather(1, (hi - 1)[:, None])[:, 0]).abs())
        worst_margin = max(worst_margin, float(gap.max()))
    del rows, ref
    # discriminator: the tier's P(real) of its own rollouts vs the fp32 tier's P(real) of the same rollouts
    p32 = d32.run_ypred_for_auc(rolls.T, w["xp"][plan.row_b].T)[:, 1]
    pb32 = d32.run_ypred_for_auc(w["y"].T, w["xp"].T)[:, 1]
    dp = np.abs(p32 - det["p_rows"][:plan.n_rows])
    # rewards rebuilt from the fp32 scores in the reference's order (model.py:317-345)
    q = np.zeros((B, T), np.float32)
    for r_ in range(R):
        add = np.repeat(pb32[:, None], T, axis=1).astype(np.float32)
        live = plan.grp_first >= 0
        add[:, :T - 1][live] = p32[plan.grp_first[live] + r_]
        q += add
    r32 = ((q - np.float32(0.6 * R)) * w["mask"]) / np.float32(R)
    reward_rel = float(np.abs(r_tier - r32).max() / np.abs(r32).max())
    res = dict(precision=precision, vehicle=vehicle, logits_rel_inf=logits_rel, logits_rel_rms=rms_rel,
               logits_abs_max_err=dmax, logits_scale=scale, softmax_entropy_nats=entropy, token_agreement=agree,
               tokens_compared=int(have.numel()), flips=int(flip.numel()), worst_flip_margin=worst_margin,
               p_real_abs_max=float(dp.max()), p_real_abs_mean=float(dp.mean()),
               p_real_spread=[float(p32.min()), float(np.median(p32)), float(p32.max())], reward_rel_inf=reward_rel)
    bnd = BOUNDS[(precision, vehicle)]
    res["asserted_bounds"] = bnd
    res["north_star"] = dict(logits_rel=1e-3, token_agreement=0.995, reward_rel=1e-3)
    res["meets_north_star"] = dict(logits=logits_rel <= 1e-3, tokens=agree >= 0.995, rewards=reward_rel <= 1e-3)
    print("C2 accuracy vs fp32 tier:", json.dumps(res))
    # the record bench.py's `tiers` block reads: written next to the other committed summaries (and under gpurun_out/,
    # the directory a GPU run hands back)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for out_dir in (os.path.join(root, "profiles"), os.path.join(root, "gpurun_out")):
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r3_precision_c2_%s_%s.json" % (precision, vehicle)), "w") as f:
            json.dump(res, f, indent=1)
    assert worst_margin <= float(np.expm1(2.0 * dmax)) + 1e-6, res          # every flip is a stated near-tie
    assert logits_rel <= bnd["logits_rel"], res
    assert agree >= bnd["agree"], res
    assert dp.max() <= bnd["p_abs"], res
    assert reward_rel <= bnd["reward_rel"], res
