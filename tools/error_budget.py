"""Error budget of the 16-bit tiers at the benchmark shape (d512 N6 h8, V = 32000, T = 50), by ablation from the
split-precision tier: one component at a time is given the fp16 tier's precision (NMTGAN_SPLIT_ABLATE, read once per
process - hence one child process per row) and the teacher-forced logits are compared with the fp32 tier's on the
same 96 sentences; every position is re-drawn from both with the same uniform (CDF inversion, fp64 CDF).
usage (GPU box): python tools/error_budget.py            -> a table on stdout + gpurun_out/error_budget.json"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

D, BLOCKS, HEADS, V, T, NS = 512, 6, 8, 32000, 50, 96
ROWS = [(0, "fp16x3", "nothing lowered (the tier as shipped)"),
        (4, "fp16x3", "logits GEMM operands in fp16"),
        (2, "fp16x3", "feed-forward GEMM operands in fp16"),
        (1, "fp16x3", "projection GEMM operands in fp16 (qkv, o, cross q / kv / o)"),
        (8, "fp16x3", "attention inputs q, K, V rounded to fp16 (the fp16 cache)"),
        (15, "fp16x3", "all four lowered (fp32 residual stream and attention arithmetic kept)"),
        (0, "fp16", "the fp16 tier"),
        (0, "bf16", "the bf16 tier")]


def child(precision, vehicle):
    import numpy as np
    import torch
    import nmt_gan_amd as N
    from nmt_gan_amd import synthetic
    from nmt_gan_amd.utils import AttrDict as Cfg
    mcfg = Cfg(hidden_units=D, num_blocks=BLOCKS, num_heads=HEADS, src_vocab_size=V, dst_vocab_size=V,
               scale_embedding=True, residual_dropout_rate=0.0, train=Cfg(shared_embedding=False))
    Wg = synthetic.generator_weights(D, BLOCKS, V, V, seed=0)
    if vehicle == "trained-like":
        Wg["embedding/softmax/kernel"] = Wg["embedding/softmax/kernel"] * np.float32(5.0)
    x, y = synthetic.synthetic_batch(NS, T, V, V, seed=77)
    out = {}
    logits = {}
    for prec in ("fp32", precision):
        m = N.Model(mcfg, precision=prec)
        m.build_rollout_generate(T)
        m.init_and_restore(weights=Wg)
        logits[prec] = m.teacher_forced_logits(x, y, device_out=True).reshape(NS * T, V)
        del m
    ref, got = logits["fp32"], logits[precision]
    out["logits_rel_inf"] = float((got - ref).abs().max() / ref.abs().max())
    out["logits_rel_rms"] = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    u = torch.rand(NS * T, generator=torch.Generator().manual_seed(5)).to(ref.device)

    def draw(lg):
        tok = torch.empty(lg.shape[0], dtype=torch.int64, device=lg.device)
        for i in range(0, lg.shape[0], 1024):
            l = lg[i:i + 1024].double()
            c = torch.cumsum(torch.exp(l - l.max(dim=1, keepdim=True).values), dim=1)
            tgt = u[i:i + 1024].double() * c[:, -1]
            tok[i:i + 1024] = torch.searchsorted(c, tgt[:, None], right=True)[:, 0].clamp(max=V - 1)
        return tok

    out["token_agreement"] = float((draw(ref) == draw(got)).double().mean())
    out["tokens"] = NS * T
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3])
    results = []
    for vehicle in ("trained-like", "reference-init"):
        for mask, precision, what in ROWS:
            env = dict(os.environ, NMTGAN_SPLIT_ABLATE=str(mask))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", precision, vehicle], env=env,
                               capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                print(p.stdout[-2000:], p.stderr[-2000:])
                raise SystemExit("child failed: %s mask %d" % (precision, mask))
            r = json.loads(line[0][7:])
            r.update(vehicle=vehicle, precision=precision, mask=mask, lowered=what)
            results.append(r)
            print("%-14s %-7s mask %2d  logits rel inf %.3g  rms %.3g  token agreement %.4f   %s"
                  % (vehicle, precision, mask, r["logits_rel_inf"], r["logits_rel_rms"], r["token_agreement"], what), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "error_budget.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
