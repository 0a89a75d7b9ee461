#!/usr/bin/env python3
"""HBM traffic of the hash-grid kernels from rocprofv3 --pmc passes over bench.py.

usage: pmc_traffic.py <dir with FETCH pass> <dir with WRITE pass> <steps in timed region> <avg samples>
                      [<dir with the TCC_EA0_ATOMIC_sum pass> [<parameters swept by Adam per step>]]
Takes the LAST steps x launches-per-step dispatches of each kernel (the timed region of bench.py) and prints
JSON with per-launch and per-sample (Adam: per-parameter) bytes.  FETCH_SIZE / WRITE_SIZE are in KiB.
"""
import csv
import glob
import json
import sys


def last(dirpath, counter, like, n):
    vals = []
    for f in glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and like in r["Kernel_Name"]:
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals.sort()
    vals = [v for _, v in vals[-n:]]
    return sum(vals) / max(len(vals), 1)


def main():
    fdir, wdir, steps, samples = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    out = {}
    for name, like, per_step in (("grid_bwd_param", "grid_bwd_param", 2), ("grid_fwd", "grid_fwd", 2),
                                 ("grid_bwd_input", "grid_bwd_input", 1)):
        f = last(fdir, "FETCH_SIZE", like, steps * per_step) * 1024
        w = last(wdir, "WRITE_SIZE", like, steps * per_step) * 1024
        out[name] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                     "fetch_bytes_per_sample": f / samples, "write_bytes_per_sample": w / samples}
    if len(sys.argv) > 5:   # memory-side atomic requests of the scatter (one per 64-byte line per wave-instruction)
        a = last(sys.argv[5], "TCC_EA0_ATOMIC_sum", "grid_bwd_param", steps * 2)
        out["grid_bwd_param"]["atomic_requests_per_launch"] = a
        out["grid_bwd_param"]["atomic_requests_per_sample"] = a / samples
    if len(sys.argv) > 6:   # clip + Adam sweep: two launches per step ([density table | MLPs], colour table)
        params = float(sys.argv[6])
        f = last(fdir, "FETCH_SIZE", "adam_kernel", steps * 2) * 1024 * 2
        w = last(wdir, "WRITE_SIZE", "adam_kernel", steps * 2) * 1024 * 2
        out["adam_step"] = {"fetch_bytes_per_step": f, "write_bytes_per_step": w,
                            "fetch_bytes_per_param": f / params, "write_bytes_per_param": w / params}
    out["note"] = ("FETCH_SIZE uncorrected (gfx950 reports 1/2 of wide coalesced reads; 4-byte-per-lane and "
                   "gather accesses are uncalibrated); WRITE_SIZE is exact for fp32 atomics (MI355X_MICROARCH.md)")
    out["avg_samples_per_launch"] = samples
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
