#!/usr/bin/env python3
"""One round of the test-time loop, kernel by kernel, from a rocprofv3 kernel trace of tools/render_bench.py.
usage: render_round_timeline.py <kernel_trace.csv> [round index counted from the end, default 20]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "march_test_kernel" in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n).replace("at::native::", "torch:")
    return n[:60]


t0 = int(rows[a]["Start_Timestamp"])
print(f"# round of {b - a} launches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from march to march")
print(" start_us   dur_us   q  kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r.get('Queue_Id', '?'):>2}  {short(r['Kernel_Name'])}")
