"""Condense rocprofv3 outputs into the small files committed under profiles/.

usage: summarize_profile.py TAG STATS_DIR [FETCH_DIR WRITE_DIR] [--mfma=MFMA_DIR]
  STATS_DIR : output dir of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`
  FETCH_DIR / WRITE_DIR : output dirs of separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over ONE
  get_reward (bench.py --steps 1 --warmup 0 --no-roofline).
Writes profiles/TAG_kernel_stats.csv (per-kernel calls / total / average) and, with the PMC dirs,
profiles/TAG_hbm_traffic.json: per kernel HBM bytes read / written per launch.  FETCH_SIZE and WRITE_SIZE
are reported in KiB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, hence the x2
(MI355X_MICROARCH.md, HBM / rocprofv3 section).
  --mfma=MFMA_DIR : output dir of a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass over one get_reward ->
  profiles/TAG_mfma_util.json: per kernel the matrix-core busy fraction = SQ_VALU_MFMA_BUSY_CYCLES (summed over
  the chip's 1024 SIMDs; 32 cycles per v_mfma_f32_32x32x16_bf16) / (1024 x elapsed cycles), elapsed cycles =
  GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs), and the MFMA rate the busy cycles stand for at the clock
  the kernel actually ran at (busy / 32 x 32768 flop for the bf16 32x32x16 form)."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def short(name):
    """Kernel name without return type, namespace and argument list; the Itanium names rocprofv3 leaves mangled
    (its demangler does not know _Float16 template arguments) are decoded far enough to read: name<type, ints>."""
    import re
    m = re.match(r"_ZN6nmtgan(\d+)", name)
    if m:
        n = int(m.group(1))
        ident, rest = name[m.end():m.end() + n], name[m.end() + n:]
        args = []
        if rest.startswith("I"):
            rest = rest[1:]
            while rest and not rest.startswith("E"):
                for pat, rep in (("DF16_", "_Float16"), ("t", "unsigned short"), ("f", "float")):
                    if rest.startswith(pat):
                        args.append(rep)
                        rest = rest[len(pat):]
                        break
                else:
                    lit = re.match(r"L[ib](\d+)E", rest)
                    if not lit:
                        break
                    args.append(lit.group(1))
                    rest = rest[lit.end():]
        return ident + ("<" + ", ".join(args) + ">" if args else "")
    name = name.replace("void ", "").replace("nmtgan::", "")
    return name.split("(")[0]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag, stats_dir = args[0], args[1]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(find(stats_dir, "kernel_stats.csv"))))
    label = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--label=")), "")
    with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w") as f:
        if label:
            f.write("# %s\n" % label)
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for r in rows:
            f.write("%s,%s,%.3f,%.2f,%s\n" % (short(r["Name"]).replace(",", ";"), r["Calls"],
                                             float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                             r["Percentage"]))
    print("wrote", tag + "_kernel_stats.csv", "(%d kernels)" % len(rows))
    if len(args) >= 4:
        agg = collections.defaultdict(lambda: {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
        for key, d, mult in (("read_bytes", args[2], 2.0), ("write_bytes", args[3], 1.0)):
            seen = collections.Counter()
            for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
                k = short(r["Kernel_Name"])
                agg[k][key] += float(r["Counter_Value"]) * 1024.0 * mult
                seen[k] += 1
            for k, n in seen.items():
                agg[k]["launches"] = n
        out = {}
        for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"])):
            n = max(v["launches"], 1)
            out[k] = {"launches": v["launches"], "read_GB": round(v["read_bytes"] / 1e9, 3),
                      "write_GB": round(v["write_bytes"] / 1e9, 3),
                      "bytes_per_launch": round((v["read_bytes"] + v["write_bytes"]) / n)}
        json.dump({"note": "one get_reward (%s); FETCH_SIZE x2 (gfx950) and WRITE_SIZE, KiB units" % (label or tag),
                   "total_GB": round(sum(v["read_GB"] + v["write_GB"] for v in out.values()), 1), "kernels": out},
                  open(os.path.join(out_dir, tag + "_hbm_traffic.json"), "w"), indent=1)
        print("wrote", tag + "_hbm_traffic.json")


def mfma_summary(tag, d, out_dir):
    agg = collections.defaultdict(lambda: {"launches": 0, "busy": 0.0, "gui": 0.0, "ns": 0.0})
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            agg[k]["busy"] += float(r["Counter_Value"])
            agg[k]["launches"] += 1
            agg[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            agg[k]["gui"] += float(r["Counter_Value"])
    out = {}
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["busy"]):
        if v["busy"] <= 0 or v["gui"] <= 0:
            continue
        cycles = v["gui"] / 8.0
        out[k] = {"launches": v["launches"], "mfma_busy_cycles": round(v["busy"]), "elapsed_cycles": round(cycles),
                  "mfma_busy_frac": round(v["busy"] / (1024.0 * cycles), 4),
                  "mfma_TFLOP_16bit": round(v["busy"] * 1024.0 / 1e12, 3),      # 1024 flop per busy cycle, either MFMA shape
                  # clock the kernels held (elapsed cycles / dispatch time; reads high on dispatches under ~0.3 ms) and
                  # the busy fraction re-priced against the 2.4 GHz spec clock = executed flop / dense peak
                  "effective_clock_GHz": round(cycles / v["ns"], 3) if v["ns"] > 0 else None,
                  "executed_frac_of_2p4GHz_peak": round(v["busy"] / (1024.0 * v["ns"] * 2.4), 4) if v["ns"] > 0 else None}
    json.dump({"note": "one get_reward (" + tag + ", single-stream schedule); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                       "(1024 SIMDs x GRBM_GUI_ACTIVE / 8); 1.0 = every matrix core issuing back to back = the dense "
                       "16-bit peak AT THE CLOCK THE KERNEL RAN AT (DVFS holds these GEMMs well under 2.4 GHz)",
               "kernels": out}, open(os.path.join(out_dir, tag + "_mfma_util.json"), "w"), indent=1)
    print("wrote", tag + "_mfma_util.json")


if __name__ == "__main__":
    for a in sys.argv[1:]:
        if a.startswith("--mfma="):
            mfma_summary([b for b in sys.argv[1:] if not b.startswith("--")][0], a.split("=", 1)[1],
                         os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    main()
