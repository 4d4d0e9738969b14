#!/usr/bin/env python
"""bench.py - rollout-scored sentences/sec of the MI355X rollout-reward path.

A "step" is one Model.get_reward pass (reference model.py:291-346) over one synthetic batch whose
inputs are already resident in HBM: encoder, teacher-forced prefix pass, rollnum x (T-1) Monte-Carlo
rollouts per sentence (KV-cached lock-step decoding + CDF-inversion sampling with injected
uniforms), discriminator scoring of every rollout, per-prefix reward averaging, and - for N > 1 -
the RCCL all-gather of the reward rows.  Weak scaling: every rank (one process per GPU) owns
`batch` sentences; the global batch is batch x N and uniforms are indexed by global sentence id.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (fields: this docstring and the argument help).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case (parity-test sized)
    "c1": dict(name="transformer-small d256 N2 h4, B8 R4 T30 V32k", d=256, blocks=2, heads=4, batch=8, rollnum=4,
               T=30, V=32000, E=256),
    # BASELINE.json configs[1]: the configuration the metric is quoted on (default)
    "c2": dict(name="transformer-base d512 N6 h8, B64 R16 T50 V32k", d=512, blocks=6, heads=8, batch=64, rollnum=16,
               T=50, V=32000, E=512),
    # BASELINE.json configs[3] per-GPU share (d=1024, 16 heads, R20, T80)
    "c4": dict(name="transformer-big d1024 N6 h16, B8 R20 T80 V32k", d=1024, blocks=6, heads=16, batch=8, rollnum=20,
               T=80, V=32000, E=512),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# dense peaks, MI355X_MICROARCH.md; fp16x3 spends three fp16 MFMA flops per algorithmic flop
MFMA_PEAK_TF = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3, "fp16x3": 2500.0 / 3}


class Cfg(dict):
    __getattr__ = dict.get


def table_gather_bytes(T, widths, nf, row_g, tes):
    """Bytes the table form of the rollouts' conv has to read for ONE get_reward: for a rollout with give point g and a
    filter of width w, every window q >= g - w + 1 adds its w - max(0, g - q) own taps (rows of n_w table entries of
    `tes` bytes) to one fp32 partial-sum row taken from the pass over y_sample when the window starts inside the prefix."""
    per_g = np.zeros(T + 1, np.int64)
    for g in range(1, T):
        for w, n in zip(widths, nf):
            q = np.arange(max(0, g - w + 1), T - w + 1)
            pre = np.maximum(0, g - q)
            per_g[g] += int(((w - pre) * n * tes + (pre > 0) * n * 4).sum())
    return int(per_g[np.asarray(row_g, np.int64)].sum())


def algorithmic_work(cfg, plan, B, Ls, es, table_es=0):
    """Algorithmic FLOPs / bytes per kernel class for ONE get_reward (table_es > 0: the rollouts' conv runs in its
    table form with entries of that many bytes)."""
    d, N, T, V, E, R = cfg["d"], cfg["blocks"], cfg["T"], cfg["V"], cfg["E"], cfg["rollnum"]
    live = plan.live.astype(np.int64)
    step_rows = int(live[1:].sum())
    dec_rows = step_rows + B * T                       # rollout steps + teacher-forced pass
    enc_rows = B * Ls
    widths = list(range(1, T, 4))
    nf = [100 + 10 * w for w in widths]
    F = sum(nf)
    conv_per_row = sum(2 * (T - w + 1) * w * E * n for w, n in zip(widths, nf))
    scored = plan.n_rows + 2 * B                       # rollouts + y_sample + source tower
    kv_entry = 2 * d * es                              # one K + one V vector
    # sharing-aware KV bytes of the rollout self-attention: per (group, step, block) the shared prefix
    # once (g+1 entries) + every row's private suffix (s entries each)
    g_of_group = plan.row_g[::R].astype(np.int64)
    steps_of_group = (plan.lens[plan.row_b[::R]] - 1 - g_of_group)
    self_entries = priv_entries = 0
    for s in range(1, plan.n_steps + 1):
        alive = steps_of_group >= s
        self_entries += int((g_of_group[alive] + 1).sum()) + int(alive.sum()) * R * s
        priv_entries += int(alive.sum()) * R * s
    cross_entries = sum(int(np.unique(plan.row_b[:live[s]]).size) * Ls for s in range(1, plan.n_steps + 1))
    conv_flops = conv_per_row * scored
    extra = {}
    if table_es:
        # the GEMMs left in the class: the per-(token, tap) tables of the whole vocabulary + y_sample and source towers
        conv_flops = 2 * V * E * sum(w * n for w, n in zip(widths, nf)) + conv_per_row * 2 * B
        extra["conv_table"] = ("hbm", table_gather_bytes(T, widths, nf, plan.row_g, table_es))
    return {
        **extra,
        "gemm_proj": ("mfma", (12 * d * d * N) * dec_rows + (8 * d * d * N + 4 * d * d * N) * enc_rows),
        "gemm_ffn": ("mfma", (16 * d * d * N) * (dec_rows + enc_rows)),
        "gemm_logits": ("mfma", 2 * d * V * dec_rows),
        "gemm_conv": ("mfma", conv_flops),
        "gemm_highway": ("mfma", 4 * F * F * scored),
        # first token of every rollout: nmtgan_sample_rows over a prefix-logits row (the contract's 4V + 8 bytes per
        # sampled token; the rollnum rollouts of a prefix share the row, so all but 1/rollnum of it comes from L2)
        "sampler_rows": ("hbm", plan.n_rows * (4 * V + 8)),
        # rollout steps (fused logits + sampling): the per-tile stats row (8 B per 64 vocabulary entries), the
        # hidden state and the bookkeeping words - what is left of the sampler once the logits never reach HBM
        "sampler": ("hbm", step_rows * (8 * V // 64 + d * es + 40)),
        "attn_self_step": ("hbm", N * kv_entry * self_entries),
        # the private-suffix pass alone (one wave per row): its K/V entries + the row's q in and output out
        "attn_self_private": ("hbm", N * (kv_entry * priv_entries + 2 * d * es * step_rows)),
        "attn_cross_step": ("hbm", N * kv_entry * cross_entries),
    }


PROFILE_TAG = {"fp16": "r3_c2_fp16", "bf16": "r1_c2_bf16", "fp16x3": "r3_c2_fp16x3"}     # committed PMC summaries under profiles/
# kernel classes -> kernels of the PMC summary (template argument 2 of the GEMM kernels is the epilogue; the element
# type prints as `unsigned short` for bf16 and `_Float16` for fp16)
TRAFFIC_KERNELS = {
    # 1 = conv + max (single-problem kernels), 101 = the grouped kernel (conv + max with the table term of the
    # partial-window reuse): together every conv product of the discriminator
    "gemm_conv": r"gemm_\w+_kernel<(unsigned short|_Float16), (1|101)[,>]",
    "gemm_logits": r"gemm_\w+_kernel<(unsigned short|_Float16), (2|100)[,>]",     # 2 in the bf16 summary, made by an earlier build
    # (the split-precision tier's fp32 row kernel also serves its 24 encoder / prefill launches, 8 % of the launches)
    # the rollout steps' self-attention: one per-row launch over shared prefix + private suffix
    "attn_self_step": r"attn_rows_(bf16|16s?)_kernel|attn_rows_kernel<float",
    "add_layernorm": r"addln_kernel",
    "sampler_rows": r"sample_kernel",
    "conv_table": r"table_conv_kernel",
}


def _profile_file(precision, kind):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                        "%s_%s.json" % (PROFILE_TAG.get(precision, "none"), kind))


def committed_traffic(cls, config, precision):
    """HBM bytes per launch of a kernel class from the committed PMC summary (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes over one get_reward of THIS workload, gfx950 corrections applied by
    tools/summarize_profile.py).  PMC counters cannot be collected inside the timed run, so this is the
    per-launch traffic of the same command measured beforehand; None when no summary matches."""
    import re
    path = _profile_file(precision, "hbm_traffic")
    if config != "c2" or cls not in TRAFFIC_KERNELS or not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    hit = [v for k, v in ks.items() if re.search(TRAFFIC_KERNELS[cls], k)]
    n = sum(v["launches"] for v in hit)
    return round(sum(v["read_GB"] + v["write_GB"] for v in hit) * 1e9 / n) if n else None


def committed_mfma_busy(cls, config, precision):
    """Matrix-core busy fraction of a GEMM class from the committed PMC summary (a separate rocprofv3 --pmc
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over one get_reward of THIS workload; tools/summarize_profile.py):
    busy cycles / (1024 SIMDs x elapsed cycles) - utilisation at the clock the kernel ran at, where `frac` prices
    the same kernel against the 2.4 GHz spec peak.  None when no summary matches."""
    import re
    path = _profile_file(precision, "mfma_util")
    if config != "c2" or cls not in TRAFFIC_KERNELS or not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    hit = [v for k, v in ks.items() if re.search(TRAFFIC_KERNELS[cls], k)]
    cyc = sum(v["elapsed_cycles"] for v in hit)
    if not cyc:
        return None
    busy = sum(v["mfma_busy_cycles"] for v in hit)
    out = {"mfma_busy_frac": round(busy / (1024.0 * cyc), 4)}
    ns = sum(v["elapsed_cycles"] / v["effective_clock_GHz"] for v in hit if v.get("effective_clock_GHz"))
    if ns:                                 # executed flop / dense peak at the 2.4 GHz spec clock, and the clock held
        out["executed_frac_of_peak"] = round(busy / (1024.0 * ns * 2.4), 4)
        out["effective_clock_GHz"] = round(cyc / ns, 3)
    return out


def committed_top_kernel(config, precision):
    """(kernel name, its class) of the kernel with the largest total time in the committed rocprofv3 kernel-trace
    summary of this workload (profiles/<tag>_kernel_stats.csv, single-stream schedule); class None when the kernel
    serves several classes (the LINEAR GEMM kernels run projections, feed-forward layers and the table build)."""
    import csv
    import re
    path = _profile_file(precision, "kernel_stats")[:-5] + ".csv"
    if config != "c2" or not os.path.exists(path):
        return None, None
    rows = [r for r in csv.reader(l for l in open(path) if not l.startswith("#"))][1:]
    if not rows:
        return None, None
    name = max(rows, key=lambda r: float(r[2]))[0].replace(";", ",")
    for cls, pat in TRAFFIC_KERNELS.items():
        if re.search(pat, name):
            return name, cls
    return name, None


NORTH_STAR = {"logits_rel": 1e-3, "token_agreement": 0.995, "reward_rel": 1e-3}


def tier_accuracy(tier):
    """Measured accuracy of a precision tier at the benchmark shape, from the records tests/test_gpu_precision_c2.py
    writes and the repository tracks (profiles/r3_precision_c2_<tier>_<vehicle>.json): the WORST of the two test
    vehicles per quantity, and whether it meets each clause of the stated tolerance (1e-3 relative on logits and
    rewards, >= 99.5 % token agreement).  16-bit and split tiers are measured against the fp32 tier on 512 replayed
    rollouts; the fp32 tier against the fp64 oracle.  None when no record is committed."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    recs = []
    for vehicle in ("reference-init", "trained-like", "fp64-oracle"):
        path = os.path.join(root, "r3_precision_c2_%s_%s.json" % (tier, vehicle))
        if os.path.exists(path):
            recs.append((vehicle, json.load(open(path))))
    if not recs:
        return None
    worst = lambda key, pick: pick(r[key] for _, r in recs if r.get(key) is not None)
    out = {"logits_rel": worst("logits_rel_inf", max), "token_agreement": worst("token_agreement", min),
           "reward_rel": worst("reward_rel_inf", max), "p_real_abs": worst("p_real_abs_max", max),
           "vehicles": [v for v, _ in recs],
           "source": "profiles/r3_precision_c2_%s_{%s}.json" % (tier, ",".join(v for v, _ in recs))}
    tol = recs[0][1].get("north_star", NORTH_STAR)          # the fp32 tier's clause is 1e-5, the others' 1e-3
    out["tolerance"] = tol
    out["meets_north_star"] = {"logits": out["logits_rel"] <= tol["logits_rel"],
                               "tokens": out["token_agreement"] >= tol["token_agreement"],
                               "rewards": out["reward_rel"] <= tol["reward_rel"]}
    out["meets_north_star"]["all"] = all(out["meets_north_star"].values())
    return out


def cpu_baseline(cfg, Wg, Wd, x, y, seed_u):
    """Reference-structured CPU path (the oracle's structure-faithful torch-CPU fp32 restatement: full
    re-decode per sampled token, all-position logits, encoder + source tower per call) timed on the
    host cores on a BOUNDED sample, extrapolated by exact (roll, give_num) counts."""
    import torch
    from oracle import nmtgan_oracle as O
    T, R = cfg["T"], cfg["rollnum"]
    Bc = 4
    gcfg = O.GenConfig(hidden_units=cfg["d"], num_blocks=cfg["blocks"], num_heads=cfg["heads"],
                       src_vocab_size=cfg["V"], dst_vocab_size=cfg["V"])
    dcfg = O.DisConfig(max_len=T, dim_word=cfg["E"], vocab_size=cfg["V"], vocab_size_s=cfg["V"])
    W32, Wd32 = O.to_torch(Wg, torch.float32), O.to_torch(Wd, torch.float32)
    xs, ys = x[:Bc], y[:Bc]
    mask = np.ones_like(ys, dtype=np.float32)
    xp = O.extend_sentence_to_maxlen(xs, T)
    u = np.random.default_rng(seed_u).random((1, T - 1, Bc, T), dtype=np.float32)
    stride = 3 if cfg["d"] >= 512 else 2
    gs = list(range(T - 1, 0, -stride))                # stratified prefixes: 1, 1+stride, ... sampled steps (~10-30 s)
    pts = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for g in gs:
            t0 = time.perf_counter()
            O.get_reward(W32, gcfg, Wd32, dcfg, xs, xp, ys, mask, 1, T, 0.6, u, mode="full", pairs={(0, g)})
            pts.append((T - g, time.perf_counter() - t0))
    elapsed = time.perf_counter() - t_all
    # each timed call = 1 rollout call of (T-g) sampled steps + 2 discriminator calls (the pair's and the
    # per-roll y_sample one); fit t = a + b*(T-g) and sum over g = 1..T-1 for one roll
    n = np.array([p[0] for p in pts], np.float64)
    t = np.array([p[1] for p in pts], np.float64)
    b, a = np.polyfit(n, t, 1) if len(pts) > 1 else (t[0] / n[0], 0.0)
    per_roll = sum(a + b * (T - g) for g in range(1, T))
    total = R * per_roll
    # the same sample through the oracle's KV-cached (incremental) decoding: separates what the restructuring
    # buys on the same cores from what the GPU buys
    pts_i = []
    t_inc = time.perf_counter()
    with torch.no_grad():
        for g in gs:
            t0 = time.perf_counter()
            O.get_reward(W32, gcfg, Wd32, dcfg, xs, xp, ys, mask, 1, T, 0.6, u, mode="incremental", pairs={(0, g)})
            pts_i.append((T - g, time.perf_counter() - t0))
    elapsed_i = time.perf_counter() - t_inc
    ni = np.array([p[0] for p in pts_i], np.float64)
    ti = np.array([p[1] for p in pts_i], np.float64)
    bi, ai = np.polyfit(ni, ti, 1) if len(pts_i) > 1 else (ti[0] / ni[0], 0.0)
    total_i = R * sum(ai + bi * (T - g) for g in range(1, T))
    return {"value": Bc / total, "unit": "rollout-scored sentences/s", "cores": torch.get_num_threads(),
            "incremental_value": Bc / total_i,
            "incremental_sample": "the same pairs through the oracle's KV-cached decoding (%.1f s measured; each timed "
                                  "call re-runs the encoder and the source tower once)" % elapsed_i,
            "kind": "port",
            "sample": "oracle structure-faithful fp32 torch-CPU, %d sentences, 1 roll, give_num in %s "
                      "(%.1f s measured), linear fit in sampled steps, extrapolated to rollnum %d x %d prefixes"
                      % (Bc, gs, elapsed, R, T - 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32", "fp16x3"],
                    help="tier of the headline `value`: fp16 (default: the 16-bit throughput tier; meets the 1e-3 logits "
                         "tolerance, not the 99.5 %% token clause at V = 32000 - see `tiers`), bf16, fp32 (exact), fp16x3 "
                         "(generator and discriminator products as fp16 MFMA over hi/lo operand splits, fp32 storage: "
                         "the fastest tier that meets every clause of the tolerance)")
    ap.add_argument("--no-fp32-leg", "--no-tier-legs", dest="no_fp32_leg", action="store_true",
                    help="skip timing the same workload on the other tiers (fp16x3: same steps/warmup; fp32: <= 3 steps)")
    ap.add_argument("--no-call-leg", action="store_true", help="skip timing whole Model.get_reward calls (host half included)")
    ap.add_argument("--batch", type=int, default=None, help="sentences per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--logits-chunk", type=int, default=None)
    ap.add_argument("--disc-chunk", type=int, default=None)
    ap.add_argument("--overlap", type=int, default=1, choices=[0, 1],
                    help="1: two-stream schedule (discriminator scoring of finished rollouts overlaps the decode steps)")
    ap.add_argument("--no-single-stream-leg", action="store_true",
                    help="skip the extra timing of the same workload on the single-stream schedule")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 rehearsal on a box with fewer GPUs than ranks: gloo backend, ranks share the visible GPUs "
                         "(device = local_rank %% device_count).  Exercises the sharding / gather / timing code of the "
                         "multi-GPU run; the JSON line is marked \"rehearsal\": true and is not a measurement")
    ap.add_argument("--conv-table", type=int, default=0, choices=[-1, 0, 1],
                    help="table form of the rollouts' discriminator conv: 0 library default (on for the 16-bit and split "
                         "tiers; the tables are rebuilt inside every step), 1 on for every tier, -1 off (conv GEMMs)")
    ap.add_argument("--aux-wgs", type=int, default=0)
    ap.add_argument("--aux-min-rows", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if args.rehearse:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and args.rehearse:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI

    import nmt_gan_amd as N
    from nmt_gan_amd import _lib, synthetic
    from nmt_gan_amd.model import gather_rows
    from nmt_gan_amd.share_function import split_bounds

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    d, T, V, R, Bl = cfg["d"], cfg["T"], cfg["V"], cfg["rollnum"], cfg["batch"]
    Bg = Bl * world
    Wg = synthetic.generator_weights(d, cfg["blocks"], V, V, seed=0)
    Wd = synthetic.discriminator_weights(T, cfg["E"], V, V, seed=1)
    widths, nfs = synthetic.disc_filters(T)
    mcfg = Cfg(hidden_units=d, num_blocks=cfg["blocks"], num_heads=cfg["heads"], src_vocab_size=V, dst_vocab_size=V,
               scale_embedding=True, residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
    model = N.Model(mcfg, precision=args.precision, device=str(dev))
    model.build_train_model()
    model.build_generate(T)
    model.build_rollout_generate(T)
    model.init_and_restore(weights=Wg)
    disc = N.DisCNN(None, T, 2, V, Bl, cfg["E"], widths, nfs, vocab_size_s=V, precision=args.precision, device=str(dev))
    disc.load_weights(Wd)
    if args.logits_chunk:
        model.logits_chunk_rows = args.logits_chunk
    if args.disc_chunk:
        model.disc_chunk_rows = args.disc_chunk
    model.overlap = bool(args.overlap)
    model.aux_gemm_wgs, model.aux_min_rows = args.aux_wgs, args.aux_min_rows
    model.conv_table = args.conv_table
    table_on = os.environ.get("NMTGAN_CONV_TABLE") != "0" and (args.conv_table > 0 or (
        args.conv_table == 0 and args.precision != "fp32" and Bl * R * (T - 1) * T >= 4 * V))   # the library's default rule

    x, y = synthetic.synthetic_batch(Bg, T, V, V, seed=1234)         # no </S>: every row full length T
    y, mask = N.deal_generated_samples(y)
    xp = N.extend_sentence_to_maxlen(x, T)
    bounds = split_bounds(Bg, world)
    b0, b1 = bounds[rank], bounds[rank + 1]
    u = model.make_uniforms(R, T, Bg, seed=2024)                      # Philox, resident in HBM
    prep = model.prepare_reward(x[b0:b1], xp[b0:b1], y[b0:b1], mask[b0:b1], R, disc, T, 0.6, u, b_offset=b0,
                                B_global=Bg)

    gather_cache = {}                     # send / receive buffers of the all-gather, kept between steps

    def step():
        r = prep.launch()
        return gather_rows(r, bounds, dev, gather_cache, copy=False) if world > 1 else r

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert bool(torch.isfinite(out).all())

    # the same workload on the single-stream schedule (the discriminator after the last decode step): the
    # schedule the per-kernel roofline leg below and the committed rocprof summaries are taken on
    single_ms = None
    if args.overlap and world == 1 and not args.no_single_stream_leg:
        prep1 = prep.with_schedule(overlap=False)
        prep1.launch()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            prep1.launch()
        sync()
        single_ms = 1e3 * (time.perf_counter() - t1) / args.steps

    result = {
        "metric": "rollout-scored sentences/sec (rollnum=16, len 50) at 1/2/4/8 MI355X" if args.config == "c2"
        else "rollout-scored sentences/sec (rollnum=%d, len %d)" % (R, T),
        "value": Bg * args.steps / elapsed, "unit": "sentences/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": cfg["name"], "config_id": args.config, "global_batch": Bg, "batch_per_gpu": Bl,
                   "rollnum": R, "max_len": T, "vocab": V, "rollouts_per_step": int(prep.plan.n_rows) * world,
                   "sampled_tokens_per_step": int(prep.plan.n_rows + prep.plan.live[1:].sum()) * world,
                   "parallelism": "sentence-sharded x%d, RCCL all-gather of rewards" % world if world > 1 else "single GPU",
                   "workspace_GiB": round(prep.workspace_bytes / 2 ** 30, 2),
                   "disc_conv": "table form: per-(token, tap) filter responses of the whole vocabulary rebuilt inside every "
                                "step (the discriminator changes between get_reward calls in gan_train.py), windows of the "
                                "rollouts gathered from them" if table_on else "conv GEMMs",
                   "schedule": "two-stream: discriminator scoring of finished rollouts overlaps the remaining decode "
                               "steps" if args.overlap else "single stream"},
    }
    if single_ms is not None:
        result["single_stream_ms_per_step"] = round(single_ms, 2)
    result["value_is"] = "device-side rate: inputs, plan tables and uniforms resident in HBM, launch() only enqueues"

    # whole Model.get_reward calls as gan_train.py makes them (numpy in, numpy out): host-side plan, ~15 H2D uploads,
    # fresh Philox uniforms, the device work, the final synchronise and the D2H copy of the rewards
    if rank == 0 and world == 1 and not args.no_call_leg:
        model.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6)
        t2 = time.perf_counter()
        for _ in range(args.steps):
            model.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6)
        result["get_reward_call_ms"] = round(1e3 * (time.perf_counter() - t2) / args.steps, 2)

    # Every tier on the same workload, timed like the headline (fp16x3: same steps and warm-up; fp32: at most 3 steps
    # after 1 warm-up - it runs ~1.7 s per step - stated in its entry), with the accuracy each tier was MEASURED at
    # (tracked records of tests/test_gpu_precision_c2.py) and which clauses of the tolerance it meets.
    if rank == 0 and world == 1 and not args.no_fp32_leg:
        def tier_leg(precision, steps, warmup):
            mt = N.Model(mcfg, precision=precision, device=str(dev))
            mt.build_rollout_generate(T)
            mt.init_and_restore(weights=Wg)
            mt.overlap, mt.conv_table = model.overlap, model.conv_table
            dt = N.DisCNN(None, T, 2, V, Bl, cfg["E"], widths, nfs, vocab_size_s=V, precision=precision, device=str(dev))
            dt.load_weights(Wd)
            pt = mt.prepare_reward(x[b0:b1], xp[b0:b1], y[b0:b1], mask[b0:b1], R, dt, T, 0.6, u, b_offset=b0, B_global=Bg)
            for _ in range(warmup):
                pt.launch()
            torch.cuda.synchronize(dev)
            tt0 = time.perf_counter()
            for _ in range(steps):
                pt.launch()
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - tt0) / steps
            del pt, mt, dt
            return ms
        tiers = {}
        for precision in ("fp16", "fp16x3", "fp32"):
            if precision == args.precision:
                ms, st_, wu_ = 1e3 * elapsed / args.steps, args.steps, args.warmup
            elif precision == "fp32":
                st_, wu_ = min(args.steps, 3), 1
                ms = tier_leg(precision, st_, wu_)
            else:
                st_, wu_ = args.steps, args.warmup
                ms = tier_leg(precision, st_, wu_)
            entry = {"ms_per_step": round(ms, 2), "sentences_per_s": round(Bl * 1e3 / ms, 2), "steps": st_, "warmup": wu_}
            acc = tier_accuracy(precision) if args.config == "c2" else None
            if acc:
                entry.update(acc)
            tiers[precision] = entry
        result["tiers"] = tiers
        result["north_star_tolerance"] = NORTH_STAR
        ok = [t for t in ("fp16", "fp16x3", "fp32") if tiers[t].get("meets_north_star", {}).get("all")]
        if ok:
            best = max(ok, key=lambda t: tiers[t]["sentences_per_s"])
            result["compliant_tier"] = best
            result["compliant_tier_value"] = tiers[best]["sentences_per_s"]
        hm = tiers.get(args.precision, {}).get("meets_north_star")
        if hm is not None:
            result["value_meets_north_star"] = hm
        # (flat field names that readers of earlier versions of this line look for)
        result["fp32_tier_ms_per_step"] = tiers["fp32"]["ms_per_step"]
        result["fp16x3_tier_ms_per_step"] = tiers["fp16x3"]["ms_per_step"]
        result["fp16x3_tier_sentences_per_s"] = tiers["fp16x3"]["sentences_per_s"]
    if args.rehearse:
        result["rehearsal"] = True

    if rank == 0 and world == 1 and not args.no_roofline:
        ncls = _lib.lib.nmtgan_profile_classes()
        ms = (C.c_double * ncls)()
        cnt = (C.c_int64 * ncls)()
        prof_prep = prep.with_schedule(overlap=False) if args.overlap else prep
        _lib.lib.nmtgan_profile_start()
        prof_prep.launch()
        _lib.check(_lib.lib.nmtgan_profile_stop(ms, cnt, ncls), "profile_stop")
        table_es = 0 if not table_on else (2 if args.precision in ("fp16", "bf16") else 4)
        work = algorithmic_work(cfg, prep.plan, Bl, x.shape[1], 4 if args.precision in ("fp32", "fp16x3") else 2, table_es)
        rows = {}
        names = list(_lib.PROFILE_CLASSES)
        i_self, i_priv = names.index("attn_self_step"), names.index("attn_self_private")
        profiled_ms = float(sum(ms))
        priv_ms, priv_cnt = ms[i_priv], cnt[i_priv]
        ms[i_self] += priv_ms                      # the contract's decode self-attention = shared + private pass
        cnt[i_self] += priv_cnt
        for i, name in enumerate(names):
            if cnt[i] == 0:
                continue
            row = {"launches": int(cnt[i]), "ms": round(ms[i], 3), "avg_us": round(1e3 * ms[i] / cnt[i], 2)}
            if name in work:
                bound, amount = work[name]
                if bound == "mfma":
                    row.update(bound="mfma", achieved=round(amount / (ms[i] * 1e-3) / 1e12, 2),
                               peak=MFMA_PEAK_TF[args.precision], unit="TFLOP/s")
                else:
                    row.update(bound="hbm", achieved=round(amount / (ms[i] * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                               unit="GB/s")
                row["frac"] = round(row["achieved"] / row["peak"], 4)
                row["algorithmic_per_step"] = amount
            tr = committed_traffic(name, args.config, args.precision)
            if tr is not None:                     # PMC bytes actually moved (committed summary) over the live time
                row["traffic_per_launch"] = tr
                row["traffic_GBps"] = round(tr * row["launches"] / (ms[i] * 1e-3) / 1e9, 1)
                if row.get("bound") == "hbm":      # HBM bytes moved / algorithmic bytes: > 1 = re-reads, < 1 = cache hits
                    row["traffic_over_algorithmic"] = round(tr * row["launches"] / row["algorithmic_per_step"], 3)
            if name == "sampler_rows" and tr is not None:
                # the rollnum rollouts of a prefix share one logits row, so most of the contract's 4V + 8 bytes per
                # token are L2 hits: price the kernel against HBM by the bytes the fabric actually moved
                row["algorithmic_GBps"] = row["achieved"]
                row["achieved"] = row["traffic_GBps"]
                row["frac"] = round(row["achieved"] / row["peak"], 4)
            if row.get("bound") == "mfma":
                mb = committed_mfma_busy(name, args.config, args.precision)
                if mb is not None:                 # PMC matrix-core counters (committed summary)
                    row.update(mb)
            rows[name] = row
        # the dominant KERNEL: the one with the largest total time in the committed rocprofv3 kernel summary of this
        # workload, priced through its class; when that kernel serves several classes (or no summary is committed),
        # the class with the largest summed launch time
        top_kernel, top = committed_top_kernel(args.config, args.precision)
        if top is None or "bound" not in rows.get(top, {}):
            top_kernel, top = None, max((k for k in rows if "bound" in rows[k]), key=lambda k: rows[k]["ms"])
        r = rows[top]
        result["roofline"] = {"kernel": top, "rocprof_kernel": top_kernel, "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"],
                              "unit": r["unit"], "frac": r["frac"],
                              "traffic": committed_traffic(top, args.config, args.precision),
                              "mfma_busy_frac": r.get("mfma_busy_frac"),
                              "algorithmic_per_launch": round(r["algorithmic_per_step"] / r["launches"]),
                              "launches": r["launches"], "avg_launch_us": r["avg_us"],
                              "note": "dominant kernel = largest total time in the committed rocprofv3 kernel summary, priced "
                                      "through its launch class (roofline_all has every class); "
                                      "HIP events around every launch of one extra (untimed) step on the SINGLE-STREAM "
                                      "schedule (kernels in isolation; same schedule as `bench.py --overlap 0`, whose "
                                      "rocprofv3 --kernel-trace --stats summary is committed under profiles/); "
                                      "achieved = algorithmic work of the class / summed launch time (a launch = one library GEMM "
                                      "call; a conv call split into 256-column panels + remainder shows as two kernels in "
                                      "rocprof, same total); traffic = HBM bytes per launch from the committed PMC summary "
                                      "profiles/" + PROFILE_TAG.get(args.precision, "none") + "_hbm_traffic.json (for gemm_conv: "
                                      "the single-problem conv+max kernels and the grouped kernel that runs the smaller and the "
                                      "partial-window products); mfma_busy_frac = matrix-core busy cycles / elapsed cycles at the clock the "
                                      "kernels held (profiles/..._mfma_util.json)"}
        result["roofline_all"] = rows
        result["profiled_step_ms"] = round(profiled_ms, 2)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, Wg, Wd, x, y, 99)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
