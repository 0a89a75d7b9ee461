#!/usr/bin/env python3
"""Per-kernel totals of the test-time frames in a rocprofv3 kernel trace of tools/render_bench.py: everything launched after the
last optimizer launch of the training phase, divided by the number of frames.
usage: render_trace.py <kernel_trace.csv> <frames>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_adam = max(i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"])
tail = rows[last_adam + 1:]
first_march = min(i for i, r in enumerate(tail) if "march_test_kernel" in r["Kernel_Name"])


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("at::native::", "torch:")
    return n[:70]


agg = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += d
rounds = agg["march_test_kernel"][0]
tot = sum(v[1] for v in agg.values())
print(f"# {frames} frames, {rounds} rounds, {len(tail)} launches; kernel time per frame {tot / frames / 1e3:.2f} ms "
      f"(includes the ray generation of the bench between frames)")
print(f"{'kernel':72s} {'calls/frame':>11s} {'ms/frame':>9s} {'avg us':>8s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:72s} {v[0] / frames:11.1f} {v[1] / frames / 1e3:9.3f} {v[1] / v[0]:8.1f}")
