#!/usr/bin/env python
"""Two-stream timeline of one get_reward from a rocprofv3 --kernel-trace CSV (bench.py --overlap 1 --steps 1 --warmup 1):
per hardware queue the busy time, first / last dispatch, and per 10 ms bin the fraction of the bin each queue has a kernel
in flight; then the tail (what still runs after the last decode-step kernel).  usage: tools/analyze_overlap.py CSV"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    ends = [i for i, n in enumerate(names) if "reward_reduce_kernel" in n]
    if len(ends) >= 2:
        rows = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = int(rows[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    per_q = defaultdict(list)
    for r in rows:
        per_q[r[qkey]].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"]))
    print("step wall %.1f ms, %d dispatches, queues: %s" % ((t1 - t0) / 1e6, len(rows), sorted(per_q)))
    nb = (t1 - t0) // 10_000_000 + 1
    for q, ds in sorted(per_q.items()):
        busy = sum(e - s for s, e, _ in ds)
        print("queue %s: %d dispatches, busy %.1f ms, first %.1f ms, last end %.1f ms" % (q, len(ds), busy / 1e6, ds[0][0] / 1e6, max(e for _, e, _ in ds) / 1e6))
        bins = [0] * nb
        for s, e, _ in ds:
            b = s // 10_000_000
            while s < e:
                nxt = min(e, (b + 1) * 10_000_000)
                bins[b] += nxt - s
                s = nxt
                b += 1
        print("   busy %% per 10 ms: " + " ".join("%3d" % (100 * x // 10_000_000) for x in bins))
        top = defaultdict(float)
        for s, e, n in ds:
            top[n.split("(")[0].replace("void ", "").replace("nmtgan::", "")[:60]] += (e - s) / 1e6
        for n, ms in sorted(top.items(), key=lambda kv: -kv[1])[:6]:
            print("   %8.1f ms  %s" % (ms, n))


if __name__ == "__main__":
    main()
