#!/usr/bin/env python3
"""Benchmark of the instant-NGP training hot path on MI355X.

A "step" is one full training step of the reference's schedule (train.py:268-307 + optimizer):
occupancy-grid update every 16 steps, ray/AABB + marcher, hash-grid + MLP field with analytic
normals, compositing, NeRFLoss (rgb + opacity + distortion), backward, gradient all-reduce
(N > 1), clip-by-norm + Adam.  Workload = BASELINE.json configs[1]: lego-like scene (synthetic
"S-lego-proxy", no dataset offline), 800x800 images, 8192 rays per batch PER GPU, L=16 F=8 hash
grids with T=2^19 (density) / 2^21 (colour), fp32.  Data parallel = weak scaling (each rank draws
its own 8192 rays).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed inside the timed region) and, at N=1, `cpu_baseline` (the CPU oracle's noCUDA
render path on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix (MFMA) peak
BYTES_PER_SAMPLE = {        # SURVEY.md §8(d), fp32, L=16, F=8 (per encoder, per sample)
    "grid_fwd": 4096 + 512,         # 8 corners x 16 levels x 32 B gathered + 512 B written
    "grid_bwd_param": 4096 + 512,   # 4096 B of atomic adds + 512 B of dL_dy read
    "grid_bwd_input": 4096 + 512 + 12,
}


N_ARG = {"grid_fwd": 0, "grid_bwd_param": 1, "grid_bwd_input": 1}  # position of `n` among the int arguments


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--rays", type=int, default=8192, help="rays per batch per GPU")
    ap.add_argument("--analytic-init", action="store_true",
                    help="initialise the occupancy grid from the analytic scene instead of the reference's "
                         "schedule (256 warm-up steps over all cells, then sampled updates)")
    ap.add_argument("--pretrain", type=int, default=600,
                    help="untimed training steps run during setup so that the occupancy grid and the field are in "
                         "the steady-state regime the reference spends >98 %% of its 20k steps in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-march-ahead", action="store_true", help="march every batch inside its own step")
    ap.add_argument("--step-times", action="store_true", help="print the host time of every timed step to stderr")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-samples", type=int, default=64)
    return ap.parse_args()


def cpu_baseline(model, scene, n_rays, n_samples):
    """Times the CPU oracle's restatement of rendering_noCUDA.render (forward rendering of n_rays
    rays x n_samples dense samples through the CPU hash-grid + MLP field) on the host cores."""
    import oracle
    from oracle import nocuda
    from oracle.field import CpuNGP
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
             if k.endswith("params") or k.startswith("xyz_net")}
    field = CpuNGP(state, scale=model.scale)
    g = torch.Generator(device=scene.device).manual_seed(7)
    img, pix = scene.sample_batch(n_rays, generator=g)
    o, d = scene.rays(img, pix)
    o, d = o.cpu().numpy(), d.cpu().numpy()
    ref = nocuda.render([field, field], o, d, [n_samples])  # warm-up (page-in, OpenMP pool)
    # matched-PSNR check of the north star: the same rays AND sample depths through the HIP field +
    # compositor (render_dense) vs this CPU path, both scored against the scene's ground truth
    from ngp_amd.rendering import render_dense
    dev = scene.device
    o_t, d_t = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    gpu = render_dense(model, o_t, d_t, torch.from_numpy(ref["z_vals0"]).to(dev))
    gt, _ = scene.ground_truth(o_t, d_t, n_quad=512)
    gt = gt.cpu().numpy()

    def _psnr(a, b):
        return float(-10 * np.log10(np.mean((a - b) ** 2)))
    rgb_gpu = gpu["rgb"].cpu().numpy()
    hit = np.isfinite(ref["rgb0"]).all(-1)   # rays that miss the scene box have near == far -> 0/0 in
    ref["rgb0"], rgb_gpu, gt = ref["rgb0"][hit], rgb_gpu[hit], gt[hit]   # rendering_noCUDA.py:146 (both paths)
    match = {"psnr_cpu_path": _psnr(ref["rgb0"], gt), "psnr_hip_same_samples": _psnr(rgb_gpu, gt),
             "psnr_hip_vs_cpu_path": _psnr(rgb_gpu, ref["rgb0"])}
    match["psnr_delta"] = match["psnr_hip_same_samples"] - match["psnr_cpu_path"]
    match["rays_compared"] = int(hit.sum())
    reps, t0 = 0, time.perf_counter()
    while True:
        nocuda.render([field, field], o, d, [n_samples])
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 50:
            break
    return {
        "value": n_rays * reps / el, "unit": "rays/s", "cores": oracle.num_threads(), "kind": "port",
        "samples_per_s": n_rays * n_samples * reps / el,
        "sample": f"{reps}x forward render of {n_rays} rays x {n_samples} dense samples "
                  f"(oracle restatement of rendering_noCUDA.render + CPU hash-grid/MLP field, {el:.1f}s)",
        "parity": match,
    }


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for several GPUs without a launcher: start one rank per GPU as child processes (nothing has
        # touched the GPU in this process yet) and pass their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # NGP_DIST_BACKEND=gloo + several ranks on one GPU rehearses the N>1 path on a 1-GPU box
    backend = os.environ.get("NGP_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NGP_FORCE_SHARDED=1: one rank, but with a process group and the sharded-optimizer path, so that
    # every RCCL call of the N>1 path is exercised on a single-GPU box (a rehearsal, not a bench line)
    solo_group = world == 1 and bool(os.environ.get("NGP_FORCE_SHARDED"))
    if solo_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or solo_group:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import ngp_amd
    from ngp_amd import _lib
    from ngp_amd.networks import NGP
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer, shard_seed
    from ngp_amd.metrics import psnr

    seed = 20220806
    torch.manual_seed(seed)  # identical initial weights on every rank
    np.random.seed(seed)
    model = NGP(scale=0.5).to(dev)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    model.grid_rng = torch.Generator(device=dev).manual_seed(seed)  # same grid updates on all ranks

    scene = LegoProxy(n_images=100, img_wh=(800, 800), device=dev, seed=seed)
    trainer = NGPTrainer(model, lr=1e-2, num_epochs=20, steps_per_epoch=1000, exp_step_factor=0.0)
    if args.analytic_init:
        model.density_grid.copy_(scene.occupancy_from_analytic(model))
        ngp_amd.vren.packbits(model.density_grid.view(-1), 0.5, model.density_bitfield)
        trainer.global_step = 1024  # past the all-cells warm-up; grid keeps updating every 16 steps
        trainer.warmup_steps = 0
    trainer.broadcast_state(0)

    ray_gen = torch.Generator(device=dev).manual_seed(shard_seed(seed, rank))

    def next_batch():
        img, pix = scene.sample_batch(args.rays, generator=ray_gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=256)
        return o, d, gt

    batches = [next_batch() for _ in range(min(args.warmup + args.steps, 64))]

    def run(k, i0, feed_next_run):
        # the loader is one batch ahead (as the reference's DataLoader workers are): the trainer
        # marches batch i+1 under step i's backward.  Exactly k batches are marched per call of
        # run(): the warm-up's last step marches the timed region's first batch, the timed
        # region's last step marches nothing.
        tot_samples = torch.zeros((), dtype=torch.int64, device=dev)
        last = None
        stamps = [time.perf_counter()]
        for i in range(k):
            o, d, gt = batches[(i0 + i) % len(batches)]
            nxt = batches[(i0 + i + 1) % len(batches)][:2] if (i + 1 < k or feed_next_run) else None
            loss, res = trainer.step(o, d, gt, next_rays=None if args.no_march_ahead else nxt)
            tot_samples += res["total_samples"]
            last = (loss, res, gt)
            stamps.append(time.perf_counter())
        if args.step_times and not feed_next_run and rank == 0:
            import gc
            dt = [round((b - a) * 1e3, 2) for a, b in zip(stamps, stamps[1:])]
            print("step host ms:", dt, "gc:", gc.get_count(), gc.get_stats()[-1], file=sys.stderr)
        return tot_samples, last

    for _ in range(args.pretrain):  # setup: fresh rays every step, not part of warm-up or timing
        o, d, gt = next_batch()
        trainer.step(o, d, gt)
    # the interpreter's full collections walk every live object (modules, the 64 ray batches, ...): ~60 ms,
    # once every few hundred steps, with the GPU draining meanwhile.  Objects alive now live for the whole
    # run: move them out of the collector's sight (what a long-running training loop does as well).
    import gc
    gc.collect()
    gc.freeze()
    run(args.warmup, 0, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # HIP events around EVERY kernel of the step cost 2.4 % of the step (measured A/B): inside the timed region
    # only the dominant kernel is bracketed (the roofline's live measurement); the other kernels' figures come
    # from a short untimed pass right after it
    DOM = "grid_bwd_param"
    prof_keys = ("grid_fwd", "grid_bwd_input", "adam_step", "linear_fwd", "linear_bwd_input",
                 "linear_bwd_weight", "mlp_bwd_input", "mlp_bwd_weight", "mlp2_fwd")
    _lib.PROFILE = {DOM: []}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot_samples, last = run(args.steps, args.warmup, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof_dom, _lib.PROFILE = _lib.PROFILE, {k: [] for k in prof_keys}
    post_steps = min(8, args.steps)
    run(post_steps, args.warmup + args.steps, False)
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    prof[DOM] = prof_dom[DOM]
    steps_of = lambda name: args.steps if name == DOM else post_steps

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    s = tot_samples.clone()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_samples = int(s.item())

    if rank == 0:
        loss, res, gt = last
        rays_total = args.rays * world * args.steps
        # per-kernel device time over the timed region (HIP events on the launch stream)
        kern = {}
        for name, evs in prof.items():
            if not evs:
                continue
            ms = [e0.elapsed_time(e1) for e0, e1, _ in evs]
            if name.startswith("grid"):
                # samples processed by each launch = the int64 `n` argument
                ns = [a[N_ARG[name]] for _, _, a in evs]
                gbs = [BYTES_PER_SAMPLE[name] * n / (m * 1e-3) / 1e9 for n, m in zip(ns, ms) if m > 0]
                kern[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                              "avg_samples": sum(ns) / len(ns), "GBps": sum(gbs) / max(len(gbs), 1)}
            elif name.startswith("linear") or name.startswith("mlp"):
                # (.., n, n_in, n_out, ..) are int arguments 2,3,4 of the three linear_* entry points and
                # (n, n_in, H) arguments 5,6,7 of the fused 2-layer entry points (mlp2_fwd, mlp_bwd_*); only the MFMA-tiled
                # launches count (the 1..16-wide heads run on the VALU "skinny" kernels)
                i0 = 5 if name.startswith("mlp") else 2
                sel = [(m, 2.0 * a[i0] * a[i0 + 1] * a[i0 + 2]) for m, (_, _, a) in zip(ms, evs)
                       if min(a[i0 + 1], a[i0 + 2]) >= 32 and m > 0]
                if sel:
                    t_ms, fl = sum(m for m, _ in sel), sum(f for _, f in sel)
                    kern[name] = {"launches": len(sel), "total_ms": t_ms, "avg_ms": t_ms / len(sel),
                                  "TFLOPs": fl / (t_ms * 1e-3) / 1e12}
            elif name == "adam_step":
                # 4 streams read + 4 written per parameter (p, g, m, v; the gradient is zeroed): 32 B each;
                # the parameter count is the first int argument
                tot_b = sum(32.0 * a[0] for _, _, a in evs)
                kern[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                              "GBps": tot_b / (sum(ms) * 1e-3) / 1e9, "frac_of_hbm_peak": tot_b / (sum(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            else:
                kern[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms)}
        for name in kern:
            kern[name]["steps"] = steps_of(name)
        grid_names = [k for k in kern if k.startswith("grid")]
        heaviest = max(grid_names, key=lambda k: kern[k]["total_ms"] / kern[k]["steps"])
        dom = DOM   # the kernel bracketed live in the timed region; `heaviest` is reported should another one out-weigh it
        # HBM traffic per launch from the committed rocprofv3 --pmc passes over this same command
        # (profiles/r01_pmc_traffic.json, tools/pmc_traffic.py): WRITE_SIZE is exact for fp32 atomics,
        # FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B; calibrated here on the known
        # 512 B/sample dL_dy stream, which it reports as 268 B).
        traffic = None
        try:
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            pmc = pmc_all[dom]
            traffic = (2 * pmc["fetch_bytes_per_sample"] + pmc["write_bytes_per_sample"]) * kern[dom]["avg_samples"]
            # achieved HBM rate of every hash-grid kernel from the same PMC passes (FETCH_SIZE as reported, i.e.
            # a lower bound for the gathers, + WRITE_SIZE) over the launch time measured in this run
            for name in grid_names:
                b = (pmc_all[name]["fetch_bytes_per_sample"] + pmc_all[name]["write_bytes_per_sample"]) * kern[name]["avg_samples"]
                kern[name]["hbm_bytes_pmc"] = b
                kern[name]["hbm_GBps_pmc"] = b / (kern[name]["avg_ms"] * 1e-3) / 1e9
        except Exception:
            pass
        roofline = {
            "kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": kern[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
            "avg_launch_ms": kern[dom]["avg_ms"], "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE[dom],
            "heaviest_grid_kernel_per_step": heaviest,
            "note": "achieved = algorithmic bytes (4096 B of fp32 atomic adds + 512 B read per sample) / launch time; "
                    "the kernel is bound by the memory-side atomic request rate (~22 G requests/s measured, "
                    "~1.3 TB/s of added bytes in ideal shapes per MI355X_MICROARCH.md), not by the 8 TB/s used for "
                    "frac; run merging with corner carry-over + zero skipping + x-pair coalescing cut the real traffic to `traffic` bytes"
                    if dom == "grid_bwd_param" else "",
        }
        # MFMA utilisation of the MLP products against the dense f32 MFMA peak of gfx950
        lin = [kern[k] for k in kern if k.startswith("linear") or k.startswith("mlp")]
        mlp = None
        if lin:
            fl = sum(k["TFLOPs"] * k["total_ms"] for k in lin)
            t_ms = sum(k["total_ms"] for k in lin)
            mlp = {"bound": "mfma", "achieved": fl / t_ms, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": fl / t_ms / MFMA_F32_PEAK_TFLOPS, "ms_per_step": t_ms / post_steps,
                   "note": "all MFMA-tiled linear_* / mlp2_fwd / mlp_bwd_* launches of the step, fp32 operands, "
                           "v_mfma_f32_32x32x2_f32; HIP events over the untimed steps right after the timed region"}
        out = {
            "metric": "train rays/sec", "value": rays_total / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "samples_per_s": total_samples / elapsed,
            "samples_per_ray": total_samples / rays_total,
            "train_psnr_last_batch": float(psnr(res["rgb"].detach(), gt)),
            "loss": float(loss),
            "config": {"workload": "NeRF-Synthetic-lego-like (S-lego-proxy analytic scene), 800x800, "
                                   f"{args.rays} rays/batch/GPU, L=16 F=8 hashgrid T=2^19 (sigma) + 2^21 (rgb), fp32, "
                                   "full train step incl. density-grid update/16 steps, NeRFLoss, clip+Adam",
                       "rays_per_gpu": args.rays, "global_rays": args.rays * world,
                       "occupancy": "analytic init" if args.analytic_init else "reference schedule from step 0",
                       "pretrain_steps": args.pretrain,
                       "parallelism": f"ray-batch dp{world}"},
            "roofline": roofline,
            "mlp_mfma": mlp,
            "kernels": kern,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, scene, args.cpu_rays, args.cpu_samples)
        print(json.dumps(out))
    if world > 1 or solo_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
