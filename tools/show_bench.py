"""Compact view of bench.py JSON lines: python tools/show_bench.py LOG [LOG ...]"""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        print("%s: %.1f ms/step  %.1f %s" % (path, d["ms_per_step"], d["value"], d["unit"]))
        for k, r in d.get("roofline_all", {}).items():
            extra = "  %8.1f %s  frac %.3f" % (r["achieved"], r["unit"], r["frac"]) if "frac" in r else ""
            print("   %-16s %5d launches %8.2f ms  avg %8.1f us%s" % (k, r["launches"], r["ms"], r["avg_us"], extra))
        if "profiled_step_ms" in d:
            print("   profiled step (sum of launches): %.1f ms" % d["profiled_step_ms"])
