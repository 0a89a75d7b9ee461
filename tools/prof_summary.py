#!/usr/bin/env python3
"""Condenses a rocprofv3 `*_kernel_stats.csv` into a short table (profiles/ keeps these)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*?>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("at::native"):
        inner = re.search(r"(CUDAFunctor\w*|FillFunctor|\w+Functor|\w+_kernel_cuda|NormTwoOps|MeanOps|sum_functor|CatArray\w+|scan\w+)", name)
        s = "torch:" + (inner.group(1) if inner else s.split("::")[-1])
    return s[:70]


def main(path, steps=None):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"# {path}\n# total kernel time {tot/1e6:.2f} ms" + (f" = {tot/1e6/steps:.3f} ms/step over {steps} steps" if steps else ""))
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
    for r in rows[:45]:
        print(f"{short(r['Name']):72s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:9.3f} "
              f"{float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
