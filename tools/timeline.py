#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 `*_kernel_trace.csv`: picks one training step near the end of
the run (steps are delimited by clip_coef_kernel launches, one per optimizer step) and prints every kernel with its start offset,
duration, queue and the idle gap on the device before it.  Also prints the busy/idle split of the step.

usage: timeline.py <kernel_trace.csv> [steps_from_end=3]
       timeline.py <kernel_trace.csv> region <K> [skip]   per-kernel time over K steps ending `skip` steps before the end
                                                          (bench.py: 40 timed steps, then 8 untimed ones with all kernels bracketed)
       timeline.py <kernel_trace.csv> update [skip]  the last step that contains a density-grid update (packbits), at least
                                                     `skip` steps before the end (bench.py's trailing steps are bracketed
                                                     with events and the very last one follows a device synchronisation:
                                                     skip 10 lands in the timed region)
       timeline.py <kernel_trace.csv> steps <K> [skip]   wall time of each of K steps (update steps marked)
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*?>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("at::native"):
        inner = re.search(r"(CUDAFunctor\w*|FillFunctor|\w+Functor|\w+_kernel_cuda|NormTwoOps|MeanOps|sum_functor|"
                          r"CatArray\w+|scan\w+|reduce_kernel|index\w+)", name)
        s = "torch:" + (inner.group(1) if inner else s.split("::")[-1])
    return s[:58]


STEP_MARKS = ("clip_coef_kernel", "clip_decide_kernel")   # one of them per optimizer step (exact norm / norm bound)


def is_mark(name):
    return any(m in name for m in STEP_MARKS)


def steps(path, k, skip=0):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if is_mark(r[2])]
    plain, upd = [], []
    for j in range(len(marks) - 1 - skip - k, len(marks) - 1 - skip):
        lo, hi = marks[j], marks[j + 1]
        wall = (rows[hi][0] - rows[lo][0]) / 1e3
        u = any("packbits" in r[2] for r in rows[lo:hi])
        (upd if u else plain).append(wall)
        print(f"step {j - len(marks) + 1:4d}: {wall:8.1f} us{'   density-grid update' if u else ''}")
    if plain and upd:
        print(f"# plain steps {sum(plain)/len(plain):.1f} us (n={len(plain)}), update steps {sum(upd)/len(upd):.1f} us (n={len(upd)}): "
              f"+{sum(upd)/len(upd) - sum(plain)/len(plain):.1f} us per update, "
              f"+{(sum(upd)/len(upd) - sum(plain)/len(plain)) * len(upd) / (len(plain) + len(upd)):.1f} us per step")


def main(path, back=3, with_update=False, skip=2):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if is_mark(r[2])]
    if len(adam) < back + 2:
        raise SystemExit("not enough steps in the trace")
    if with_update:
        back = next(b for b in range(max(2, skip), len(adam) - 1)
                    if any("packbits" in r[2] for r in rows[adam[-b - 1]:adam[-b]]))
    lo, hi = adam[-back - 1], adam[-back]          # [optimizer of step k-1 ... optimizer of step k)
    step = rows[lo:hi]
    t0 = step[0][0]
    wall = rows[hi][0] - t0
    print(f"# step of {len(step)} launches, wall (clip_coef start -> next clip_coef start) {wall/1e3:.1f} us")
    print(f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'q':>3s}  kernel")
    busy_end = t0
    idle = 0
    for s, e, n, q in step:
        gap = max(0, s - busy_end)
        idle += gap
        print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} {gap/1e3:7.1f} {q:>3s}  {short(n)}")
        busy_end = max(busy_end, e)
    print(f"# device idle inside the step: {idle/1e3:.1f} us of {wall/1e3:.1f} us")


def region(path, k, skip=0):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if is_mark(r[2])]
    # the timed region ends with the optimizer of its last step: take the K steps before the last mark
    lo, hi = marks[-k - 1 - skip], marks[-1 - skip]
    sel = rows[lo:hi]
    wall = rows[hi][0] - rows[lo][0]
    agg = {}
    for s, e, n in sel:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += e - s
        a[1] += 1
    tot = sum(a[0] for a in agg.values())
    print(f"# timed region only ({k} steps, {skip} trailing steps skipped): wall {wall/k/1e6:.3f} ms/step, sum of kernel durations "
          f"{tot/k/1e6:.3f} ms/step (streams overlap), {len(sel)/k:.0f} launches/step")
    for name, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
        print(f"{name:60s} {ns/k/1e6:8.3f} ms/step {c/k:6.1f} calls/step {ns/c/1e3:9.1f} us/call")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "region":
        region(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif len(sys.argv) > 2 and sys.argv[2] == "update":
        main(sys.argv[1], with_update=True, skip=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    elif len(sys.argv) > 3 and sys.argv[2] == "steps":
        steps(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
