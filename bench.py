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
ADAM_BYTES_PER_PARAM = 28   # p, g, m, v read + p, m, v written (the gradient zero-fill is skipped
                            # for pieces that are zero already, so it is not counted as algorithmic)
ATOMIC_REQ_PEAK = 20.0e9    # memory-side 64-byte atomic requests/s (tools/atomic_shapes.hip, profiles/r02_atomic_shapes.txt)


N_ARG = {"grid_fwd": 0, "grid_bwd_param": 1, "grid_bwd_input": 1}  # position of `n` among the int arguments
SCATTER_CALLS = ("grid_bwd_param", "grid_bwd_param_scaled")   # entry points of the table scatter (same kernel)
ADAM_CALLS = ("adam_step", "adam_step_width")                   # entry points of the clip + Adam sweep (same kernel)


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
    ap.add_argument("--windows", type=int, default=5,
                    help="further untimed-by-contract windows of --steps steps after the timed region (spread of the figure)")
    ap.add_argument("--no-solo", action="store_true", help="skip the cache-cold solo replays of the captured launches")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-samples", type=int, default=64)
    return ap.parse_args()


def _host_cpu():
    """(model string, physical cores, hardware threads, CPU quota of the container or None) of the host"""
    model, cores, threads = "unknown", set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                threads += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))      # a container may see fewer CPUs than the machine has
    except (AttributeError, OSError):
        allowed = threads or 1
    n_phys = len(cores) or allowed
    # a container's CPU-time quota (cgroup v2 cpu.max / v1 cfs quota): more runnable threads than that only get throttled
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return model, min(n_phys, allowed), allowed, quota


def cpu_baseline(model, scene, n_rays, n_samples, crop=200):
    """Times the CPU oracle's restatement of rendering_noCUDA.render (forward rendering of n_rays rays x n_samples
    dense samples through the CPU hash-grid + MLP field) on the host cores, BASELINE.json configs[0]: rays drawn from
    the centred `crop` x `crop` window of the training images, 1024 rays per batch.  Three ways: the field as one C call
    over all points with OpenMP over the points on one thread per PHYSICAL core (`value`); the same on one thread; and
    the layered restatement of rounds 1-2 (per-layer C calls with numpy glue between them, which the interpreter
    serialises)."""
    import oracle
    from oracle import nocuda
    from oracle.field import CpuNGP
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
             if k.endswith("params") or k.startswith("xyz_net")}
    field = CpuNGP(state, scale=model.scale)
    g = torch.Generator(device=scene.device).manual_seed(7)
    w, h = scene.img_wh
    crop = min(crop, w, h)
    img = torch.randint(scene.poses.shape[0], (n_rays,), device=scene.device, generator=g)
    uu = torch.randint(crop, (n_rays,), device=scene.device, generator=g) + (w - crop) // 2
    vv = torch.randint(crop, (n_rays,), device=scene.device, generator=g) + (h - crop) // 2
    o, d = scene.rays(img, vv * w + uu)
    o, d = o.cpu().numpy(), d.cpu().numpy()
    t_u = np.random.default_rng(20220806).random(n_rays, dtype=np.float32)   # the U[0,1) draws of rendering_noCUDA.py:139
    ref = nocuda.render([field, field], o, d, [n_samples], t_rand_u=t_u)  # warm-up (page-in, OpenMP pool)
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
    rgb_ref, rgb_gpu, gt = ref["rgb0"][hit], rgb_gpu[hit], gt[hit]   # rendering_noCUDA.py:146 (both paths)
    match = {"psnr_cpu_path": _psnr(rgb_ref, gt), "psnr_hip_same_samples": _psnr(rgb_gpu, gt),
             "psnr_hip_vs_cpu_path": _psnr(rgb_gpu, rgb_ref)}
    match["psnr_delta"] = match["psnr_hip_same_samples"] - match["psnr_cpu_path"]
    match["rays_compared"] = int(hit.sum())

    def timed(fn, budget_s, max_reps):
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= max_reps:
                return reps, el
    cpu_model, phys, hw_threads, quota = _host_cpu()
    all_threads = oracle.num_threads()
    whole = lambda: nocuda.render([field, field], o, d, [n_samples], t_rand_u=t_u)
    reps_omp, el_omp = timed(whole, 4.0, 50)

    # the same path with the field (two hash-grid gathers, d sigma/dx, the four MLPs) as ONE C call over all points,
    # OpenMP over the points, on one thread per physical core: no interpreter work between the layers
    class FusedField:
        center, half_size = field.center, field.half_size

        def __call__(self, xyzs, dirs, embed_a=None):
            return oracle.field_forward(field, xyzs, dirs) + (None,)
    ff = FusedField()
    fused = lambda: nocuda.render([ff, ff], o, d, [n_samples], t_rand_u=t_u)
    workers = max(1, phys if quota is None else min(phys, int(quota)))   # cores this process can actually run on
    oracle.set_num_threads(workers)
    out = fused()["rgb0"]                     # warm-up, and the fused field must reproduce the layered one
    same = bool(np.allclose(out[hit], rgb_ref, rtol=1e-4, atol=1e-5))
    reps, el = timed(fused, 10.0, 400)
    # the same sample on ONE thread as well (a scalar port's figure), a few seconds of it
    oracle.set_num_threads(1)
    reps1, el1 = timed(fused, 6.0, 10)
    oracle.set_num_threads(all_threads)
    return {
        "value": n_rays * reps / el, "unit": "rays/s", "cores": workers, "kind": "port",
        "threads": workers, "physical_cores": phys, "hardware_threads": hw_threads, "container_cpu_quota": quota,
        "samples_per_s": n_rays * n_samples * reps / el,
        "sample": f"{reps}x forward render of {n_rays} rays (centred {crop}x{crop} crop, configs[0]) x {n_samples} dense samples "
                  f"(oracle restatement of rendering_noCUDA.render; CPU hash-grid/MLP field as one C call, OpenMP over the "
                  f"points, {workers} threads = one per core this container may use: {phys} physical cores, CPU quota {quota}), {el:.1f}s; "
                  f"reproduces the layered field: {same}",
        "cpu_model": cpu_model,
        "layered_numpy_field": {"value": n_rays * reps_omp / el_omp, "unit": "rays/s", "threads": all_threads,
                                "sample": f"{reps_omp}x the same batch with the field as per-layer C calls + numpy glue "
                                          f"(OpenMP inside the C calls only; rounds 1-2's baseline) ({el_omp:.1f}s)"},
        "one_thread": {"value": n_rays * reps1 / el1, "unit": "rays/s", "cores": 1,
                       "sample": f"{reps1}x the same fused render on 1 thread ({el1:.1f}s)"},
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

    tuning = trainer.adam_width is None and not trainer.sharded
    if tuning:
        trainer.adam_tune = (1 << 60, trainer.adam_tune[1])   # not during the setup below: its loop renders fresh ground truth every step
    for _ in range(args.pretrain):  # setup: fresh rays every step, not part of warm-up or timing
        o, d, gt = next_batch()
        trainer.step(o, d, gt)
    if tuning:
        # the trainer times its two Adam launch widths in windows of 16 steps and keeps the faster one (NGPTrainer.__init__);
        # a training run does that once, at steps 320-448 of its own loop — here it happens in THIS loop's regime (resident
        # batches, march-ahead), still as untimed setup, before the warm-up and the timed region
        first = -(-trainer.global_step // trainer.update_interval) * trainer.update_interval
        trainer.adam_tune = (first, trainer.adam_tune[1])
        n_tune = first - trainer.global_step + 2 * trainer.adam_tune[1] * trainer.update_interval + 4
        run(n_tune, 0, True)
        torch.cuda.synchronize()
        run(4, n_tune, True)   # (the measurement is read once its last event has completed)
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
    # HIP events around EVERY library call of the step cost 2.4 % of the step (measured A/B): inside the timed region only
    # the candidates for "dominant kernel" are bracketed — the two hash-grid gathers (3 launches per step), the table
    # scatter (2) and the clip + Adam sweep (2): 7 of ~55 launches, 14 events per step.  Whichever took most device time per
    # step in the timed region is the `roofline` kernel; the MLP products' figures come from a short untimed pass after it.
    LIVE = ("grid_fwd", "grid_bwd_input", "grid_bwd_param", "adam_step")   # "grid_bwd_param" collects both scatter entry points
    prof_keys = ("linear_fwd", "linear_bwd_input", "linear_bwd_weight", "mlp_bwd_input", "mlp_bwd_weight", "mlp2_fwd",
                 "mlp2_fwd_dact", "sumsq")   # (mlp2_fwd_dact = the density head's forward: the same kernel, one more output)
    _lib.PROFILE = {k: [] for k in LIVE + SCATTER_CALLS + ADAM_CALLS}
    step_at_start = trainer.global_step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot_samples, last = run(args.steps, args.warmup, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof_live, _lib.PROFILE = _lib.PROFILE, {k: [] for k in prof_keys}
    updates_in_region = sum(1 for g in range(step_at_start, step_at_start + args.steps) if g % trainer.update_interval == 0)
    post_steps = min(8, args.steps)
    trainer.buckets.trace = []
    run(post_steps, args.warmup + args.steps, False)
    torch.cuda.synchronize()
    comm_trace, trainer.buckets.trace = trainer.buckets.trace, None
    prof, _lib.PROFILE = _lib.PROFILE, None      # the windows and replays below are timed on their own
    # the timed region is short (K steps of ~4 ms): further windows of the same K steps, each between two device
    # synchronisations, say how the region's figure sits in the run-to-run spread (reported, never the `value`)
    windows = []
    pos = args.warmup + args.steps + post_steps
    for _ in range(args.windows):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tw = time.perf_counter()
        run(args.steps, pos, False)
        torch.cuda.synchronize()
        windows.append((time.perf_counter() - tw) / args.steps * 1e3)
        pos += args.steps
    # SOLO rates: one more untimed step with the arguments of the gather / scatter / MLP launches captured, then every one
    # of them replayed alone on an idle device.  In the step these kernels share the memory system with whatever runs on
    # the other streams (the scatters run beside the MLP weight products, the density gather beside the Adam sweep), so
    # their in-step durations say how the schedule shares the device, the solo ones what the kernel itself does.  Every
    # repetition is timed on its own behind a 1 GiB fill: tables (174 / 588 MB) and activations start in HBM, not in the
    # 256 MiB Infinity Cache (back-to-back replays of one launch read above the HBM peak).
    solo_keys = SCATTER_CALLS + ("mlp_bwd_input", "mlp_bwd_weight", "mlp2_fwd", "mlp2_fwd_dact", "grid_fwd", "grid_bwd_input")
    solo = {}
    if world == 1 and not args.no_solo:
        extra = 0
        if trainer.global_step % trainer.update_interval == 0:   # not a step that starts with an occupancy update
            run(1, pos, False)
            extra = 1
        _lib.CAPTURE = {k: [] for k in solo_keys}
        run(1, pos + extra, False)
        trainer.wait()
        torch.cuda.synchronize()
        cap, _lib.CAPTURE = _lib.CAPTURE, None
        flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB
        for name, calls in cap.items():
            for a in calls:
                _lib.call(name, *a)               # once untimed (code / TLB warm)
                reps = []
                for _ in range(3):
                    flush.fill_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _lib.call(name, *a)
                    e1.record()
                    torch.cuda.synchronize()
                    reps.append(e0.elapsed_time(e1))
                key = "grid_bwd_param" if name in SCATTER_CALLS else ("mlp2_fwd" if name == "mlp2_fwd_dact" else name)
                solo.setdefault(key, []).append((sorted(reps)[1], tuple(x for x in a if isinstance(x, int))))
        del flush
    # multi-GPU readiness: what every rank sent through the backend per step, and behind which HIP stream
    main_stream = torch.cuda.current_stream().cuda_stream
    per_step = {}
    for ev in comm_trace:
        key = (ev["op"], ev["bucket"], ev["stream"])
        acc = per_step.setdefault(key, [0.0, 0.0, 0])
        acc[0] += ev["bytes"] / post_steps
        if "events" in ev:
            acc[1] += ev["events"][0].elapsed_time(ev["events"][1])
            acc[2] += 1
    comm = {"backend": (dist.get_backend() if dist.is_initialized() else None), "world_size": world,
            "sharded_optimizer": bool(trainer.sharded),
            "per_step": [{"op": op, "bucket": b, "MB": round(v[0] / 1e6, 3),
                          "ms": (round(v[1] / v[2], 4) if v[2] else None),
                          # payload / time, and the bus bandwidth of a ring / direct reduce-scatter or all-gather of that payload
                          "GBps": (round(v[0] * post_steps / v[2] / (v[1] / v[2] * 1e-3) / 1e9, 2) if v[2] and v[1] > 0 else None),
                          "bus_GBps": (round((world - 1) / max(world, 1) * v[0] * post_steps / v[2] / (v[1] / v[2] * 1e-3) / 1e9, 2)
                                       if v[2] and v[1] > 0 else None),
                          "enqueued_behind": "main stream" if st == main_stream else f"side stream {st:#x}"}
                         for (op, b, st), v in sorted(per_step.items())],
            "note": "ms: HIP events on the issuing stream around each collective in the traced steps behind the timed region, "
                    "where every collective is waited for at once (its own duration, not the overlapped schedule)"}
    if world > 1 or solo_group:
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else "n/a"
        except Exception:
            rccl = "unknown"
        print(f"[rank {rank}] {comm['backend']} world size {dist.get_world_size()} (RCCL {rccl}), "
              f"sharded optimizer {trainer.sharded}; per step: " +
              "; ".join(f"{c['op']} bucket {c['bucket']} {c['MB']} MB {c['ms']} ms ({c['bus_GBps']} GB/s bus) behind {c['enqueued_behind']}"
                        for c in comm["per_step"]),
              file=sys.stderr, flush=True)
    prof.update(prof_live)
    # the field calls the scaled entry point, other callers the plain one: one kernel, one entry in the tables
    prof["grid_bwd_param"] = [ev for k in SCATTER_CALLS for ev in prof.pop(k, [])]
    prof["adam_step"] = [ev for k in ADAM_CALLS for ev in prof.pop(k, [])]
    prof["mlp2_fwd"] = prof.get("mlp2_fwd", []) + prof.pop("mlp2_fwd_dact", [])
    steps_of = lambda name: args.steps if name in LIVE else post_steps

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
                # achieved rate = sum of algorithmic bytes over sum of launch time (NOT the mean of per-launch
                # ratios: the colour-table and density-table launches differ in duration)
                kern[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                              "avg_samples": sum(ns) / len(ns),
                              "algorithmic_bytes": float(BYTES_PER_SAMPLE[name]) * sum(ns),
                              "GBps": BYTES_PER_SAMPLE[name] * sum(ns) / (sum(ms) * 1e-3) / 1e9}
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
                # p, g, m, v read + p, m, v written per parameter = 28 B; the parameter count is
                # the first int argument
                tot_b = sum(float(ADAM_BYTES_PER_PARAM) * a[0] for _, _, a in evs)
                kern[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                              "avg_params": sum(a[0] for _, _, a in evs) / len(evs), "algorithmic_bytes": tot_b,
                              "GBps": tot_b / (sum(ms) * 1e-3) / 1e9, "frac_of_hbm_peak": tot_b / (sum(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            else:
                kern[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms)}
        for name, runs in solo.items():
            if name not in kern or not runs:
                continue
            ms = [m for m, _ in runs]
            kern[name]["solo_avg_ms"] = sum(ms) / len(ms)
            if name.startswith("grid"):
                ns = [a[N_ARG[name]] for _, a in runs]
                kern[name]["solo_GBps"] = BYTES_PER_SAMPLE[name] * sum(ns) / (sum(ms) * 1e-3) / 1e9
                if kern[name]["solo_GBps"] > HBM_PEAK_GBS:
                    kern[name]["solo_note"] = ("algorithmic bytes (SURVEY §8(d): every corner row of every sample) per second; above the "
                                               "HBM peak because runs of samples share rows on chip — not an HBM rate, see hbm_bytes_pmc")
            else:
                sel = [(m, 2.0 * a[5] * a[6] * a[7]) for m, a in runs if min(a[6], a[7]) >= 32]
                if sel:
                    kern[name]["solo_TFLOPs"] = sum(f for _, f in sel) / (sum(m for m, _ in sel) * 1e-3) / 1e12
                    kern[name]["solo_ms_mfma"] = sum(m for m, _ in sel)
                    kern[name]["solo_flop_mfma"] = sum(f for _, f in sel)
        for name in kern:
            kern[name]["steps"] = steps_of(name)
            kern[name]["ms_per_step"] = kern[name]["total_ms"] / kern[name]["steps"]
        grid_names = [k for k in kern if k.startswith("grid")]
        # dominant kernel = most device time per step among the kernels bracketed INSIDE the timed region (the gathers,
        # the scatter, the clip + Adam sweep); `heaviest_kernel_per_step` is the same maximum over every kernel measured
        # (the MLP products come from the post-region pass) — the two must agree, and the line says so
        live = [k for k in LIVE if k in kern]
        dom = max(live, key=lambda k: kern[k]["ms_per_step"])
        heaviest = max(kern, key=lambda k: kern[k]["ms_per_step"])
        # HBM traffic per launch from the committed rocprofv3 --pmc passes over this same command
        # (profiles/rNN_pmc_traffic.json, tools/pmc_traffic.py), ONE definition everywhere: 2 x FETCH_SIZE + WRITE_SIZE
        # (MI355X_MICROARCH.md §HBM: gfx950 tallies 128-byte read requests at 64 bytes; WRITE_SIZE is exact for fp32
        # atomics and 16-byte streaming stores).
        pmc_all = {}
        for fn in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pmc_all = json.load(open(os.path.join(ROOT, "profiles", fn)))
                pmc_all["_file"] = fn
                break
            except Exception:
                continue

        def traffic_of(name):
            pmc = pmc_all.get(name)
            if not pmc:
                return None
            if "fetch_bytes_per_sample" in pmc:
                return (2 * pmc["fetch_bytes_per_sample"] + pmc["write_bytes_per_sample"]) * kern[name]["avg_samples"]
            if "fetch_bytes_per_param" in pmc:
                return (2 * pmc["fetch_bytes_per_param"] + pmc["write_bytes_per_param"]) * kern[name]["avg_params"]
            return None

        for name in grid_names + ["adam_step"]:
            b_ = traffic_of(name) if name in kern else None
            if b_:
                kern[name]["hbm_bytes_pmc"] = b_
                kern[name]["hbm_GBps_pmc"] = b_ / (kern[name]["avg_ms"] * 1e-3) / 1e9

        NOTES = {
            "grid_fwd": "hash-grid gather, both tables: 16 levels x 8 corners x 32 B read + 512 B written per sample (SURVEY §8(d)); "
                        "a run of consecutive samples in one cell loads its corners once (run-leader kernel), and what binds the "
                        "kernel is the number of 64-byte L1 misses a CU keeps in flight (profiles/r03_gather_pmc.txt), not HBM bytes",
            "grid_bwd_input": "the same gather on the density table with the upstream gradient (4096 + 512 + 12 B per sample)",
            "grid_bwd_param": "4096 B of fp32 atomic adds + 512 B read per sample; the binding resource is the memory-side "
                              "atomic REQUEST rate (one request per 64-byte line a wave-instruction touches, 20 G/s "
                              "measured: profiles/r02_atomic_shapes.txt), see `atomic_requests`; line-aligned run merging + "
                              "zero skipping cut the real traffic to `traffic` bytes per launch",
            "adam_step": "clip + Adam sweep over all 200 M parameters: p, g, m, v read, p, m, v written; the next step's density gather runs beside it",
        }

        def roof(name):
            k = kern[name]
            r = {"kernel": name, "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": k["GBps"] / HBM_PEAK_GBS, "traffic": traffic_of(name), "traffic_source": pmc_all.get("_file"),
                 "avg_launch_ms": k["avg_ms"], "ms_per_step": k["ms_per_step"], "launches": k["launches"],
                 "algorithmic_bytes_per_launch": k["algorithmic_bytes"] / k["launches"],
                 "formula": "achieved = sum over the timed region's launches of algorithmic bytes / sum of their HIP-event durations",
                 "note": NOTES[name]}
            if "solo_GBps" in k:
                fr = k["solo_GBps"] / HBM_PEAK_GBS
                r["solo"] = {"achieved": k["solo_GBps"], "frac": fr, "avg_launch_ms": k["solo_avg_ms"],
                             "note": "the same launches (arguments captured from one more step) replayed alone, each repetition "
                                     "behind a 1 GiB fill (tables and activations start in HBM, not in the Infinity Cache)"}
                if fr > 1.0:     # algorithmic bytes above the HBM peak = the bytes did not come from HBM: not a roofline figure
                    r["solo"] = {"invalid": "algorithmic rate above the HBM peak (cache-resident operands)", "achieved": k["solo_GBps"]}
            if name == "adam_step":
                r["algorithmic_bytes_per_param"] = ADAM_BYTES_PER_PARAM
            else:
                r["algorithmic_bytes_per_sample"] = BYTES_PER_SAMPLE[name]
            if name == "grid_bwd_param":
                req = pmc_all.get(name, {}).get("atomic_requests_per_sample")
                if req:
                    rate = req * k["avg_samples"] / (k["avg_ms"] * 1e-3)
                    r["atomic_requests"] = {"per_sample": req, "per_launch": req * k["avg_samples"], "rate_per_s": rate,
                                            "peak_per_s": ATOMIC_REQ_PEAK, "frac": rate / ATOMIC_REQ_PEAK}
            return r
        roofline = roof(dom)
        roofline["heaviest_kernel_per_step"] = heaviest
        roofline["other_candidates"] = [roof(k) for k in sorted(live, key=lambda k: -kern[k]["ms_per_step"]) if k != dom]
        # MFMA utilisation of the MLP products against the dense f32 MFMA peak of gfx950
        lin = [kern[k] for k in kern if k.startswith("linear") or k.startswith("mlp")]
        mlp = None
        if lin:
            fl = sum(k["TFLOPs"] * k["total_ms"] for k in lin)
            t_ms = sum(k["total_ms"] for k in lin)
            mlp = {"bound": "mfma", "achieved": fl / t_ms, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": fl / t_ms / MFMA_F32_PEAK_TFLOPS, "ms_per_step": t_ms / post_steps,
                   "note": "all MFMA-tiled linear_* / mlp2_fwd / mlp_bwd_* launches of the step, fp32 operands, "
                           "v_mfma_f32_32x32x2_f32; HIP events over the untimed steps right after the timed region "
                           "(in-step: the weight products run beside the two table scatters on purpose)"}
            sfl = sum(k.get("solo_flop_mfma", 0.0) for k in lin)
            sms = sum(k.get("solo_ms_mfma", 0.0) for k in lin)
            if sms > 0:
                mlp["solo"] = {"achieved": sfl / (sms * 1e-3) / 1e12, "frac": sfl / (sms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                               "ms_per_step": sms, "note": "the same launches replayed alone on an idle device"}
        # the whole step against its byte floor: 9,216 B/sample for the two gathers, 9,216 for the two scatters,
        # 4,620 for the analytic-normal gather, 28 B per parameter for clip + Adam (one rank's share of the samples and, when the
        # optimizer is sharded, of the parameters)
        n_param = sum(p.numel() for p in model.parameters())
        step_bytes = 23052.0 * (total_samples / world) / args.steps + 28.0 * n_param / (world if world > 1 else 1)
        step_level = {"algorithmic_bytes_per_step": step_bytes, "GBps": step_bytes / (elapsed / args.steps) / 1e9,
                      "frac_of_hbm_peak": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                      "note": "23,052 B per sample (hash-grid gathers, scatters, analytic-normal gather: SURVEY 8(d)) + 28 B per "
                              "parameter (clip + Adam) over ms_per_step; activations (MLP inputs / hidden layers, ~8 KB per "
                              "sample) are not in the floor"}
        out = {
            "metric": "train rays/sec", "value": rays_total / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "ms_per_step_median_of_windows": (sorted(windows)[len(windows) // 2] if windows else None),
            "ms_per_step_windows": [round(v, 4) for v in windows],
            "grid_update_steps_in_timed_region": updates_in_region,
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
                       "pipelining": "steady state: batch i+1 is marched on a side stream under step i, so the timed "
                                     f"region marches {args.steps - 1} of its {args.steps} batches (the first was marched "
                                     "under the last warm-up step) and runs every other stage of all of them",
                       "adam_sweep_workgroups": trainer.adam_width,   # chosen by the trainer's own measurement (None: still measuring / sharded)
                       "adam_sweep_ms_per_window": getattr(trainer, "adam_tune_ms", None),
                       "parallelism": f"ray-batch dp{world}"},
            "roofline": roofline,
            "step_level": step_level,
            "mlp_mfma": mlp,
            "comm": comm,
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
