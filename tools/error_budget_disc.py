"""Error budget of the discriminator's scoring path at the benchmark shape (T = 50, E = 512, 13 widths 1..49,
F = 4550, V = 32000), by ablation from the split-precision form (DisCNN(precision="fp16x3")): ONE stage at a time is
packed without the lo halves of its operand - which leaves exactly the fp16 tier's rounding of that operand - and
P(real) of the same rollouts is compared with the fp32 discriminator's (cnn_discriminator.py:440-522 computes in fp32).
Stages: the sentence image (embedding rows), the conv filters by width group, the pooled features entering the highway
layer, the highway matrices; the 2-way head is fp32 in every tier.  Rewards are rebuilt from both score sets in the
reference's order (model.py:317-345, bias 0.6) and compared at tensor scale.
Both test vehicles: "reference-init" (random BatchNorm statistics, |score| ~ 100) and "trained-like" (statistics that
fit the activations, O(1) scores).
usage (GPU box): python tools/error_budget_disc.py   -> a table on stdout + gpurun_out/error_budget_disc.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

D, BLOCKS, HEADS, V, E, T, R, B = 512, 6, 8, 32000, 512, 50, 16, 8


def main():
    import torch
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    from nmt_gan_amd.utils import AttrDict as Cfg
    widths, nfs = synthetic.disc_filters(T)
    mcfg = Cfg(hidden_units=D, num_blocks=BLOCKS, num_heads=HEADS, src_vocab_size=V, dst_vocab_size=V,
               scale_embedding=True, residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
    x, y = synthetic.synthetic_batch(B, T, V, V, seed=77)
    y[2, 31] = 3
    y, mask = N.deal_generated_samples(y)
    xp = N.extend_sentence_to_maxlen(x, T)
    u = torch.rand((R, T - 1, B, T), generator=torch.Generator().manual_seed(99)).numpy()

    def disc(precision, Wd, ablate=None):
        d = N.DisCNN(None, T, 2, V, B, E, widths, nfs, vocab_size_s=V, precision=precision)
        d.split_ablate = ablate or {}
        return d.load_weights(Wd)

    # the rollouts every discriminator scores: one get_reward of the fp16 tier (their content does not matter here)
    m = N.Model(mcfg, precision="fp16")
    m.build_rollout_generate(T)
    m.init_and_restore(weights=synthetic.generator_weights(D, BLOCKS, V, V, seed=0))
    m.get_reward(x, xp, y, mask, R, disc("fp16", synthetic.discriminator_weights(T, E, V, V, seed=1)), max_len=T,
                 bias_num=0.6, uniforms=u, return_details=True)
    plan = m.last_details["plan"]
    rolls = m.last_details["rollouts"][:plan.n_rows].copy()
    del m
    src = xp[plan.row_b]

    def scores(d):
        return d.run_ypred_for_auc(rolls.T, src.T)[:, 1], d.run_ypred_for_auc(y.T, xp.T)[:, 1]

    def rewards(p, pb):
        q = np.zeros((B, T), np.float32)
        live = plan.grp_first >= 0
        for r_ in range(R):
            add = np.repeat(pb[:, None], T, axis=1).astype(np.float32)
            add[:, :T - 1][live] = p[plan.grp_first[live] + r_]
            q += add
        return ((q - np.float32(0.6 * R)) * mask) / np.float32(R)

    narrow, mid, wide = [w for w in widths if w <= 13], [w for w in widths if 13 < w <= 33], [w for w in widths if w > 33]
    rows = [("fp16x3", None, "nothing lowered (the split form as shipped)"),
            ("fp16x3", {"emb": True}, "sentence image (embedding rows) in fp16"),
            ("fp16x3", {"conv": narrow}, "conv filters of widths %s in fp16" % narrow),
            ("fp16x3", {"conv": mid}, "conv filters of widths %s in fp16" % mid),
            ("fp16x3", {"conv": wide}, "conv filters of widths %s in fp16" % wide),
            ("fp16x3", {"conv": list(widths)}, "all conv filters in fp16"),
            ("fp16x3", {"pooled": True}, "pooled features entering the highway layer in fp16"),
            ("fp16x3", {"hw": True}, "highway matrices in fp16"),
            ("fp16x3", {"pooled": True, "hw": True}, "highway layer (features and matrices) in fp16"),
            ("fp16x3", {"emb": True, "conv": list(widths)}, "conv operands (image and filters) in fp16, highway split"),
            ("fp16x3", {"emb": True, "conv": list(widths), "pooled": True, "hw": True}, "every stage lowered"),
            ("fp16", None, "the fp16 discriminator"),
            ("bf16", None, "the bf16 discriminator")]
    results = []
    for vehicle in ("reference-init", "trained-like"):
        Wd = synthetic.discriminator_weights(T, E, V, V, seed=1, calibrated=vehicle == "trained-like")
        d32 = disc("fp32", Wd)
        p32, pb32 = scores(d32)
        r32 = rewards(p32, pb32)
        del d32
        for precision, ablate, what in rows:
            d = disc(precision, Wd, ablate)
            p, pb = scores(d)
            del d
            dp = np.abs(p.astype(np.float64) - p32)
            res = dict(vehicle=vehicle, precision=precision, lowered=what, ablate=ablate, rollouts=int(plan.n_rows),
                       p_real_abs_max=float(max(dp.max(), np.abs(pb - pb32).max())), p_real_abs_mean=float(dp.mean()),
                       reward_rel_inf=float(np.abs(rewards(p, pb) - r32).max() / np.abs(r32).max()))
            results.append(res)
            print("%-14s %-7s max|dP| %.3g  mean|dP| %.3g  rewards rel %.3g   %s"
                  % (vehicle, precision, res["p_real_abs_max"], res["p_real_abs_mean"], res["reward_rel_inf"], what), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "error_budget_disc.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
