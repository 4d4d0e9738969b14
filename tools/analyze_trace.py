#!/usr/bin/env python
"""Per-step view of a rocprofv3 --kernel-trace CSV of one get_reward (tools/trace_step.sh): the dispatches of the LAST
step are cut at the last embed-of-the-encoder boundary, kernels are shortened to a class name, and the private-suffix
attention launches / GEMMs are listed per rollout step.  usage: tools/analyze_trace.py CSV [--kernel SUBSTR | --steps | --kinds]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("nmtgan::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<.*>)?", name)
    base = m.group(1)
    targs = m.group(2) or ""
    targs = targs.replace("unsigned short", "bf16").replace("_Float16", "f16")
    return base + targs.split("(")[0]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows]
    # last step: starts at the second-to-last... find embed kernels; a step = [encoder embed ... reward_reduce]
    ends = [i for i, x in enumerate(t) if "reward_reduce_kernel" in x[0]]
    if len(ends) >= 2:
        t = t[ends[-2] + 1: ends[-1] + 1]
    tot = defaultdict(lambda: [0, 0.0])
    for n, s, e, _ in t:
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e3
    wall = (t[-1][2] - t[0][1]) / 1e3
    busy = sum(v[1] for v in tot.values())
    print("step wall %.1f us, kernel busy %.1f us, %d dispatches" % (wall, busy, len(t)))
    for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%9.1f us %6d x %8.1f  %s" % (us, c, us / c, n[:150]))
    if len(sys.argv) > 3 and sys.argv[2] == "--kernel":
        sub = sys.argv[3]
        print("--- dispatches matching", sub)
        for i, (n, s, e, r) in enumerate(t):
            if sub in n:
                print("%6d %9.1f us grid %s wg %s  %s" % (i, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"), n[:100]))


def per_step():
    """--steps: wall / busy time and dispatch count of every rollout step (cut at the step's embed kernel), with the
    rows of the step read off the embed grid (4 rows per 256-thread workgroup)"""
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows]
    ends = [i for i, x in enumerate(t) if "reward_reduce_kernel" in x[0]]
    if len(ends) >= 2:
        t = t[ends[-2] + 1: ends[-1] + 1]
    cuts = [i for i, x in enumerate(t) if "embed_kernel" in x[0] and "disc_embed" not in x[0]]
    cuts.append(len(t))
    print("  seg   rows~   wall_us   busy_us  launches  gap_us/launch")
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = t[a:b]
        wall = ((t[b][1] if b < len(t) else seg[-1][2]) - seg[0][1]) / 1e3
        busy = sum(e - s for _, s, e, _ in seg) / 1e3
        g = int(seg[0][3].get("Grid_Size_X", seg[0][3].get("Grid_Size", 0))) // 256 * 4
        print("%5d %7d %9.1f %9.1f %8d %10.2f" % (a, g, wall, busy, len(seg), (wall - busy) / len(seg)))


def per_step_kinds():
    """--kinds: per rollout step, the time of each kind of launch summed over the decoder blocks (a step of the fused
    16-bit path is embed, 9 launches per block in a fixed order, the logits GEMM and six sampler launches)."""
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows]
    ends = [i for i, x in enumerate(t) if "reward_reduce_kernel" in x[0]]
    if len(ends) >= 2:
        t = t[ends[-2] + 1: ends[-1] + 1]
    cuts = [i for i, x in enumerate(t) if "embed_kernel" in x[0] and "disc_embed" not in x[0]]
    cuts.append(len(t))
    kinds = ["qkv", "self", "o+ln", "cq", "cross", "o+ln", "ffn1", "ffn2", "addln"]
    print("microseconds per step, summed over the decoder blocks")
    print(" rows keys " + " ".join("%6s" % k for k in kinds) + " logits sample   step")
    keys = 0
    lens = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    step_len = max(set(lens), key=lens.count)            # the rollout steps all have the same number of launches
    n_blocks = (step_len - 8) // 9
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = t[a:b]
        if (len(seg) != step_len and not (b == len(t) and len(seg) > step_len)) or "gemm" not in seg[1][0]:
            continue                    # encoder / prefill segments
        seg = seg[:step_len]            # the last step's segment also holds the discriminator tail
        keys += 1
        d = [(e - s) / 1e3 for _, s, e, _ in seg]
        g = int(seg[0][3].get("Grid_Size_X", seg[0][3].get("Grid_Size", 0))) // 256 * 4
        lay = [sum(d[1 + 9 * l + k] for l in range(n_blocks)) for k in range(9)]
        lg = 1 + 9 * n_blocks
        print("%5d %4d " % (g, keys) + " ".join("%6.0f" % v for v in lay) + " %6.0f %6.0f %6.0f" % (d[lg], sum(d[lg + 1:]), sum(d)))


if __name__ == "__main__":
    if "--kinds" in sys.argv:
        per_step_kinds()
    elif "--steps" in sys.argv:
        per_step()
    else:
        main()
