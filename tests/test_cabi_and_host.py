"""CPU-only checks of the drop-in boundary and the host logic: the C-ABI library loads and exports
exactly what include/ngp_hip.h declares, the Python surface mirrors the reference's names, the
product refuses to run without a GPU (no CPU fallback), trainer schedule / gradient buckets."""
import ctypes
import inspect
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(ngp):
    lib = ngp._lib.load()
    protos = ngp._lib.PROTOS
    assert len(protos) >= 35
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/ngp_hip.h but missing from libngp_hip.so"
    # nothing exported under the ngp_ prefix that the header does not declare
    out = subprocess.check_output(["nm", "-D", "--defined-only", ngp._lib.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T ngp_" in l}
    assert exported == set(protos), exported ^ set(protos)
    assert lib.ngp_version().startswith(b"ngp_hip")


def test_vren_surface_matches_reference_binding(ngp):
    # the 15 names of models/csrc/binding.cpp:323-342 with the reference's argument lists
    expected = {
        "ray_aabb_intersect": ["rays_o", "rays_d", "centers", "half_sizes", "max_hits"],
        "ray_sphere_intersect": ["rays_o", "rays_d", "centers", "radii", "max_hits"],
        "morton3D": ["coords"],
        "morton3D_invert": ["indices"],
        "packbits": ["density_grid", "density_threshold", "density_bitfield"],
        "raymarching_train": ["rays_o", "rays_d", "hits_t", "density_bitfield", "cascades", "scale",
                              "exp_step_factor", "noise", "grid_size", "max_samples"],
        "raymarching_test": ["rays_o", "rays_d", "hits_t", "alive_indices", "density_bitfield", "cascades", "scale",
                             "exp_step_factor", "grid_size", "max_samples", "N_samples"],
        "composite_alpha_fw": ["sigmas", "deltas", "rays_a", "T_threshold"],
        "composite_train_fw": ["sigmas", "rgbs", "normals_pred", "sems", "deltas", "ts", "rays_a", "T_threshold",
                               "classes"],
        "composite_train_bw": ["dL_dopacity", "dL_ddepth", "dL_drgb", "dL_dnormal_pred", "dL_dsem", "dL_dws",
                               "sigmas", "rgbs", "normals_pred", "ws", "deltas", "ts", "rays_a", "opacity", "depth",
                               "rgb", "normal_pred", "T_threshold", "classes"],
        "composite_test_fw": ["sigmas", "rgbs", "normals", "normals_raw", "sems", "deltas", "ts", "hits_t",
                              "alive_indices", "T_threshold", "classes", "N_eff_samples", "opacity", "depth", "rgb",
                              "normal", "normal_raw", "sem"],
        "composite_refloss_fw": ["sigmas", "normals_diff", "normals_ori", "deltas", "ts", "rays_a", "T_threshold"],
        "composite_refloss_bw": ["dL_dloss_o", "dL_dloss_p", "sigmas", "normals_diff", "normals_ori", "deltas", "ts",
                                 "rays_a", "loss_o", "loss_p", "T_threshold"],
        "distortion_loss_fw": ["ws", "deltas", "ts", "rays_a"],
        "distortion_loss_bw": ["dL_dloss", "ws_inclusive_scan", "wts_inclusive_scan", "ws", "deltas", "ts", "rays_a"],
    }
    for name, args in expected.items():
        fn = getattr(ngp.vren, name)
        assert list(inspect.signature(fn).parameters) == args, name


def test_operator_surface_names(ngp):
    cf = ngp.custom_functions
    for name in ("RayAABBIntersector", "RaySphereIntersector", "RayMarcher", "VolumeRenderer", "RefLoss", "TruncExp",
                 "ReLU", "TruncTanh", "sample_pdf", "raw2outputs"):
        assert hasattr(cf, name)
    assert list(inspect.signature(ngp.rendering.render).parameters)[:3] == ["model", "rays_o", "rays_d"]
    assert ngp.rendering.MAX_SAMPLES == 1024 and ngp.rendering.NEAR_DISTANCE == 0.01
    for name in ("Encoding", "Network", "NetworkWithInputEncoding"):
        assert hasattr(ngp.tinycudann, name)
    assert hasattr(ngp.torch_scatter, "segment_csr")


def test_no_cpu_fallback(ngp):
    """CHECK_INPUT semantics (models/csrc/include/utils.h:4-6): CPU tensors are rejected, loudly."""
    o = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ngp.vren.ray_aabb_intersect(o, o, torch.zeros(1, 3), torch.ones(1, 3), 1)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ngp.tinycudann.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})(o)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ngp.tinycudann.Network(16, 3, {"otype": "CutlassMLP", "n_neurons": 32, "n_hidden_layers": 1})(torch.zeros(4, 16))
    src = open(os.path.join(ROOT, "instant-ngp-pp_amd", "_lib.py")).read()
    for mod in os.listdir(os.path.join(ROOT, "instant-ngp-pp_amd")):
        if mod.endswith(".py"):
            txt = open(os.path.join(ROOT, "instant-ngp-pp_amd", mod)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, f"{mod} must not use the oracle"
    assert "CDLL" in src


def test_grid_layout_matches_oracle_and_reference_counts(ngp):
    """Level geometry of the two encoders of models/networks.py:36-76 at scale 0.5."""
    b = float(np.exp(np.log(2048 * 0.5 / 16) / 15))
    for log2_T in (19, 21):
        d = ngp._lib.GridDesc()
        n = ngp._lib.call_host("grid_layout", 16, 8, log2_T, 16, b, d)
        od, on = oracle.grid_layout(16, 8, log2_T, 16, b)
        assert n == on
        assert list(d.offsets)[:17] == list(od.offsets)[:17]
        assert list(d.resolution)[:16] == list(od.resolution)[:16]
        res = list(d.resolution)[:16]
        # fp32 evaluation of 16*b^l - 1 (as tcnn does) lands a hair above the integers at l = 5, 10, 15,
        # so those levels get one more cell than the real-valued formula (64 -> 65, 1024 -> 1025)
        assert res[0] == 16 and res[-1] in (1024, 1025) and all(a < c for a, c in zip(res, res[1:]))
        # every level is either dense (>= res^3 rows, multiple of 8) or capped at 2^log2_T
        for l in range(16):
            size = d.offsets[l + 1] - d.offsets[l]
            assert size % 8 == 0 and (size == 2 ** log2_T or size >= res[l] ** 3)
    with pytest.raises(ValueError):
        ngp.tinycudann.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 3})


def test_ngp_state_dict_keys_and_sizes(ngp):
    """State-dict layout of the reference's NGP (SURVEY.md §5 'Checkpoint'): flat tcnn `.params`."""
    m = ngp.networks.NGP(scale=0.5)
    sd = m.state_dict()
    for k in ("center", "xyz_min", "xyz_max", "half_size", "density_bitfield", "xyz_encoder.params",
              "xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_encoder.params",
              "rgb_net.params", "norm_pred_header.params", "semantic_header.params"):
        assert k in sd, k
    assert sd["rgb_net.params"].numel() == 20480              # 144x128 + 128x16 (§8 M2)
    assert sd["norm_pred_header.params"].numel() == 4608      # 128x32 + 32x16 (§8 M3)
    assert sd["semantic_header.params"].numel() == 4608
    assert sd["density_bitfield"].numel() == 128 ** 3 // 8 and m.cascades == 1
    assert ngp.networks.NGP(scale=8.0).cascades == 5 and ngp.networks.NGP(scale=16.0).cascades == 6
    e = ngp.networks.NGP(scale=0.5, embed_a=True, embed_a_len=8)
    assert e.rgb_net.padded_in == 160                          # 144 + 8 padded to a multiple of 16


def test_trainer_lr_schedule_matches_torch_cosine(ngp):
    from ngp_amd.trainer import NGPTrainer
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-2)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 20, 2e-2 / 30)
    t = NGPTrainer.__new__(NGPTrainer)
    t.base_lr, t.num_epochs = 2e-2, 20
    for epoch in range(21):
        assert math.isclose(t.lr_at(epoch), opt.param_groups[0]["lr"], rel_tol=1e-6), epoch
        opt.step()
        sch.step()


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import ngp_amd  # noqa: F401
    from ngp_amd.trainer import GradBuckets, shard_seed
    g = torch.Generator().manual_seed(shard_seed(20220806, rank))
    flat = torch.randn(1000, generator=g)
    mine = flat.clone()
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    total = sum(gathered)
    b = GradBuckets(flat, [0, 600, 1000])
    # sharded path: each rank ends up with its slice of the summed bucket in its own buffer ...
    shards = [torch.zeros(300), torch.zeros(200)]
    b.reduce_scatter_bucket(0, shards[0])   # fired from the rgb encoder's backward in the trainer
    b.reduce_scatter_bucket(1, shards[1])   # fired after backward
    b.wait()
    ok = all(torch.allclose(shards[i], total[slice(*b.shard_range(i))], atol=1e-6) for i in range(2))
    # ... updates it, and the all-gather rebuilds the full parameter vector on every rank
    params = torch.zeros(1000)
    for i in range(2):
        b.all_gather_bucket(i, params, shards[i] * 0.5)
    b.wait()
    ok = ok and torch.allclose(params, total * 0.5, atol=1e-6)
    # detached gathers (the trainer hands these to the model, which waits where it first reads
    # the parameters): small bucket first, then the large one
    params2 = torch.zeros(1000)
    w1 = b.all_gather_bucket(1, params2, shards[1] * 0.25, detach=True)
    w0 = b.all_gather_bucket(0, params2, shards[0] * 0.25, detach=True)
    ok = ok and not b.works
    w1.wait()
    w0.wait()
    ok = ok and torch.allclose(params2, total * 0.25, atol=1e-6)
    # plain all-reduce mode
    flat2 = mine.clone()
    b2 = GradBuckets(flat2, [0, 600, 1000])
    b2.reduce_bucket(0)
    b2.reduce_bucket(1)
    b2.wait()
    ok = ok and torch.allclose(flat2, total, atol=1e-6)
    q.put((rank, bool(ok), float(mine[0])))
    dist.destroy_process_group()


def test_grad_buckets_allreduce_gloo_world2():
    """N>1 path on CPU: two ranks with different (seed+rank) gradients end up with the same summed
    buckets; the averaging itself is folded into the clip coefficient (1/world)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] != res[1][2]  # ranks really drew different data


def test_empty_and_negative_batches_through_the_c_abi(ngp):
    """include/ngp_hip.h: "empty batches (n == 0) are valid and return NGP_OK before any pointer is looked at", and a
    negative size is NGP_EINVAL — for every entry point that takes a batch size.  Runs without a GPU: neither case may
    reach a launch (all data pointers are NULL here)."""
    import ctypes as C
    _lib = ngp._lib
    lib = _lib.load()
    size_args = {"n", "n_rays", "n_alive", "count", "n_seg", "n_bytes"}
    desc = _lib.GridDesc()
    assert _lib.call_host("grid_layout", 16, 8, 19, 16, 1.3195079, desc) > 0
    small = {"min_samples": 1, "max_samples_total": 1024, "degree": 4, "max_hits": 1, "n_voxels": 1, "n_spheres": 1, "cascades": 1, "n_samples": 1, "classes": 1, "width": 1,
             "cols": 1, "grid_size": 128, "max_samples": 1024, "n_in": 16, "n_out": 1, "H": 32, "step": 1}
    # entry points whose empty call needs more than NULLs (an output scalar, a counter) are exercised on the GPU instead
    needs_outputs = {"ngp_density_grid_ema_threshold", "ngp_raymarching_train", "ngp_nerf_loss", "ngp_sumsq", "ngp_sumsq_if",
                     "ngp_row_norm_sum", "ngp_live_rows", "ngp_test_round_begin"}   # (n_rays there is the frame's ray count)
    checked = 0
    for name, (_, args) in _lib.PROTOS.items():
        names = [a for _, a in args]
        if not (set(names) & size_args) or names[-1] != "stream":
            continue

        def build(size):
            vals = []
            for t, a in args:
                if a in size_args:
                    vals.append(size)
                elif a == "desc":
                    vals.append(C.addressof(desc))
                elif t is C.c_void_p:
                    vals.append(None)
                elif t in (C.c_float, C.c_double):
                    vals.append(1.0)
                elif a.startswith("ld"):
                    vals.append(128)
                else:
                    vals.append(small.get(a, 0))
            return vals
        assert getattr(lib, name)(*build(-1)) == -22, name
        if name not in needs_outputs:
            assert getattr(lib, name)(*build(0)) == 0, name
        checked += 1
    assert checked >= 40


# ---------------------------------------------------------------------------- inline-assembly loads of the product build
_ASM_BAD = """
_Z10fake_kernelv:
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v[2:3], off offset:0
\t;;#ASMEND
\tglobal_store_dword v[4:5], v6, off
{between}
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\tv_add_f32_e32 v20, v10, v11
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
"""


def _run_checker(path, want):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py"), path, want],
                          capture_output=True, text=True)


def test_asm_load_checker_detects_violations(tmp_path):
    """the checker itself: a use, a copy or an overwrite of a load destination before the covering wait is a finding;
    the same stream without it is clean; a wait that does not cover the load (vmcnt too large) is a finding"""
    cases = {"clean": ("\tv_mul_f32_e32 v30, v31, v32", 0), "read": ("\tv_mov_b32_e32 v40, v11", 1),
             "overwrite": ("\tv_mov_b32_e32 v12, 0", 1), "spill": ("\tscratch_store_dwordx4 off, v[10:13], off", 1)}
    for name, (between, rc) in cases.items():
        p = tmp_path / f"{name}.s"
        p.write_text(_ASM_BAD.format(between=between))
        r = _run_checker(str(p), "fake_kernel")
        assert r.returncode == rc, (name, r.stdout)
    p = tmp_path / "uncovered.s"
    p.write_text(_ASM_BAD.format(between="").replace("vmcnt(1)", "vmcnt(2)"))
    assert _run_checker(str(p), "fake_kernel").returncode == 1


def test_product_build_asm_loads_are_covered(ngp, tmp_path):
    """mlp_stream_fwd_kernel / mlp_stream_dgrad_kernel prefetch their rows with inline-assembly loads and hand-placed
    s_waitcnt vmcnt(N): compile the PRODUCT source with the PRODUCT flags to ISA and check that no instruction touches a
    load destination before the wait that covers it (a compiler update that re-schedules around the asm statements must
    fail here, on the CPU, not fault on the GPU)."""
    from ngp_amd import build
    if not build.have_hipcc():
        pytest.skip("hipcc not available")
    flags = [f for f in build.COMMON if f != "-fPIC"]
    out = tmp_path / "mlp_kernels.s"
    subprocess.check_call([build._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", str(out),
                                                      os.path.join(build.CSRC, "mlp_kernels.hip")],
                          stderr=subprocess.DEVNULL)
    r = _run_checker(str(out), "mlp_stream_")
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [l for l in r.stdout.splitlines() if "loop of" in l]
    with_loads = [l for l in lines if " 0 asm loads" not in l]
    assert any("mlp_stream_fwd_kernel" in l for l in with_loads) and any("mlp_stream_dgrad_kernel" in l for l in with_loads), \
        "the checker found no inline-assembly loads in the streaming kernels: its parsing no longer matches the ISA listing"


def test_product_library_reads_no_environment(ngp):
    """include/ngp_hip.h: "the product build reads NO environment variable" — the shared object must not even import
    getenv (A/B switches are compiled in with -DNGP_AB_VARIANTS only)."""
    if os.environ.get("NGP_AB_VARIANTS"):
        pytest.skip("A/B build")
    ngp._lib.load()
    out = subprocess.check_output(["nm", "-D", ngp._lib.LIB_PATH], text=True)
    assert "getenv" not in out


def test_adam_width_measurement_state_machine(ngp, monkeypatch):
    """NGPTrainer._adam_width_now without a GPU (events faked): windows of `update_interval` steps alternate between the
    candidates, one event per window boundary, the faster width stays — the default keeps its place on a tie — and a run
    that resumes past the start begins at the next window boundary"""
    import ngp_amd  # noqa: F401
    from ngp_amd.trainer import NGPTrainer

    clock = {"t": 0.0, "cost": {512: 1.0, 256: 1.0}, "width": 512}

    class FakeEvent:
        def __init__(self, enable_timing=False):
            self.t = None

        def record(self, stream=None):
            self.t = clock["t"]

        def query(self):
            return True

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)

    def drive(first_step, n_steps, start, rounds, cost):
        tr = NGPTrainer.__new__(NGPTrainer)
        tr.adam_width, tr.adam_candidates, tr.adam_tune = None, (512, 256), (start, rounds)
        tr._tune_events, tr.update_interval = [], 16
        clock["t"], clock["cost"] = 0.0, cost
        seen = []
        for gs in range(first_step, first_step + n_steps):
            tr.global_step = gs
            w = tr._adam_width_now(None)
            seen.append(w)
            clock["t"] += cost[w]          # a step with this width takes this long
        return tr, seen

    tr, seen = drive(1, 200, 32, 2, {512: 1.0, 256: 0.9})
    assert seen[:32] == [512] * 32                                   # before the measurement: the default
    assert seen[32:48] == [512] * 16 and seen[48:64] == [256] * 16 and seen[64:80] == [512] * 16 and seen[80:96] == [256] * 16
    assert tr.adam_width == 256 and seen[-1] == 256
    assert abs(tr.adam_tune_ms[512] - 16.0) < 1e-9 and abs(tr.adam_tune_ms[256] - 14.4) < 1e-9
    tr, seen = drive(1, 200, 32, 2, {512: 1.0, 256: 0.995})           # within the windows' own spread: the default stays
    assert tr.adam_width == 512
    tr, seen = drive(1, 200, 32, 2, {512: 0.9, 256: 1.0})
    assert tr.adam_width == 512
    tr, seen = drive(5004, 300, 320, 4, {512: 1.0, 256: 0.8})         # resumed at step 5003: starts at the next multiple of 16
    assert tr.adam_tune[0] == 5008 and tr.adam_width == 256
    tr = NGPTrainer.__new__(NGPTrainer)                               # a fixed width is never measured
    tr.adam_width = 384
    assert tr._adam_width_now(None) == 384
