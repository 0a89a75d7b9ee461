#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files."""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*?>)?)", name)
    return (m.group(1) if m else name)[:60]


def main(pattern, like):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if like and like not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k in sorted(agg):
        print(k, {c: (round(v / cnt[k][c], 1), cnt[k][c]) for c, v in agg[k].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
