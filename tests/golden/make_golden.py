#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN pure-torch functions.

Runs only in the build container (needs /root/reference); the GPU box uses the committed
.npz files.  The reference's `models.custom_functions` / `models.rendering_noCUDA` import
`vren` and `torch_scatter` (absent here: CUDA extension / un-vendored dependency) and call
`.cuda()`; they are imported with empty stand-in modules for those two names and an identity
`Tensor.cuda`, which leaves every function exercised here (pure torch) untouched.
`RayAABBIntersector` (a vren call) is replaced by a pure-torch slab test for the
rendering_noCUDA.render case; its outputs are stored in the fixture as inputs.

G6 / G7 go further and run the reference's own NGP class and render() on the CPU: `tinycudann` is
the pure-torch stand-in tcnn_cpu_shim.py (asserted equal to the C oracle's encoder), `vren` is
filled with the C oracle's restatements (install_oracle_vren), so the fixtures pin the reference's
PYTHON wiring of the field and of the renderer; the primitives themselves are pinned separately
(known-answer tests of the oracle, bit / tolerance parity of the HIP kernels against it).

Fixtures hold numbers only (inputs and the reference's outputs).  Seed 20220806 = the
reference's own seed (train.py:402).  G1-G7, G9, G11 regenerate byte-identically; G8 / G10 contain
gradients accumulated by multi-threaded CPU index_add and differ in the last bits from run to run
(the committed files are the ones the tests were run against).  A full run takes about 5 minutes.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20220806


def import_reference():
    sys.path.insert(0, REF)
    for name in ("vren", "torch_scatter"):
        sys.modules[name] = types.ModuleType(name)

    def segment_csr(src, indptr):
        return torch.stack([src[indptr[i]:indptr[i + 1]].sum(0) for i in range(len(indptr) - 1)])

    sys.modules["torch_scatter"].segment_csr = segment_csr
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models import custom_functions as cf
    from models import rendering_noCUDA as rn
    return cf, rn


def npz(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                   for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in arrs.items()})


def g1_raw2outputs(cf):
    torch.manual_seed(SEED)
    cases = {}
    idx = 0
    for (R, S) in [(1, 1), (4, 1), (4, 8), (257, 8), (4, 64), (2, 1024)]:
        for C in (0, 7, 10):
            for kind in ("zero", "uniform", "huge"):
                if kind != "uniform" and (R, S) not in [(4, 8), (2, 1024)]:
                    continue
                raw = torch.rand(R, S, 10 + C)
                if kind == "zero":
                    raw[..., 0] = 0
                elif kind == "uniform":
                    raw[..., 0] = raw[..., 0] * 50
                else:
                    raw[..., 0] = raw[..., 0] * 1e6
                z = torch.sort(torch.rand(R, S) * 3 + 0.05, -1)[0]
                d = torch.randn(R, 3)
                outs = cf.raw2outputs(raw, z, d, classes=C)
                p = f"c{idx}_"
                cases[p + "raw"], cases[p + "z"], cases[p + "d"] = raw, z, d
                cases[p + "classes"] = np.int64(C)
                for n, o in zip(("opacity", "rgb", "normal_raw", "normal_pred", "sem", "ws", "depth"), outs):
                    cases[p + n] = o
                idx += 1
    cases["n_cases"] = np.int64(idx)
    npz("g1_raw2outputs.npz", **cases)


def g2_sample_pdf(cf):
    torch.manual_seed(SEED + 1)
    cases = {}
    idx = 0
    for (R, nb, ns) in [(1, 8, 16), (5, 63, 64), (3, 63, 128), (4, 30, 7)]:
        bins = torch.sort(torch.rand(R, nb + 1) * 4, -1)[0]
        w = torch.rand(R, nb)
        w[0] = 0  # a zero-weight row
        if R > 2:
            w[2, : nb // 2] = 0
        out = cf.sample_pdf(bins, w, ns, det=True)
        p = f"c{idx}_"
        cases[p + "bins"], cases[p + "w"], cases[p + "n"], cases[p + "out"] = bins, w, np.int64(ns), out
        idx += 1
    cases["n_cases"] = np.int64(idx)
    npz("g2_sample_pdf.npz", **cases)


class FakeField(torch.nn.Module):
    """Analytic field; tests/test_oracle_golden.py carries the same formulas in numpy."""

    def __init__(self, classes, last):
        super().__init__()
        self.classes, self.last = classes, last
        self.register_buffer("center", torch.zeros(1, 3))
        self.register_buffer("half_size", torch.ones(1, 3) * 0.5)
        self.register_buffer("M", torch.linspace(-1, 1, 3 * classes).reshape(3, classes))

    def forward(self, x, d, embed_a, **kw):
        r2 = (x * x).sum(-1)
        sig = 40 * torch.exp(-r2 / 0.05)
        rgb = 0.5 + 0.5 * torch.sin(8 * x) * (0.5 + 0.5 * embed_a[:, :1])
        sems = torch.softmax(x @ self.M, -1)
        if not self.last:
            return sig, rgb, sems
        n_raw = -x / torch.sqrt(r2 + 1e-6)[:, None]
        n_pred = 0.5 * n_raw + 0.1
        return sig, rgb, n_raw, n_pred, sems, None


def torch_aabb(rays_o, rays_d, center, half_size):
    inv = 1.0 / rays_d
    a = (center - half_size - rays_o) * inv
    b = (center + half_size - rays_o) * inv
    t1 = torch.minimum(a, b).max(-1)[0]
    t2 = torch.maximum(a, b).min(-1)[0]
    miss = t1 > t2
    t1 = torch.where(miss, -torch.ones_like(t1), t1)
    t2 = torch.where(miss, -torch.ones_like(t2), t2)
    hit = t2 > 0
    out = -torch.ones(len(rays_o), 1, 2)
    out[hit, 0, 0] = t1[hit].clamp(min=0)
    out[hit, 0, 1] = t2[hit]
    return hit.int(), out, torch.where(hit, 0, -1)[:, None]


def g3_render(cf, rn):
    C = 7
    torch.manual_seed(SEED + 2)
    n = 96
    # cameras on a sphere of radius 1.5 looking roughly at the origin (all rays hit the box)
    o = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1) * 1.5
    tgt = (torch.rand(n, 3) - 0.5) * 0.6
    d = torch.nn.functional.normalize(tgt - o, dim=-1) * (1.0 + 0.2 * torch.rand(n, 1))

    class Intersector:
        @staticmethod
        def apply(rays_o, rays_d, center, half_size, max_hits):
            return torch_aabb(rays_o, rays_d, center, half_size)

    rn.RayAABBIntersector = Intersector
    cases = {"rays_o": o, "rays_d": d, "classes": np.int64(C)}
    for tag, samples in (("a", [64]), ("b", [32, 64])):
        emb = [torch.rand(n, 4) for _ in samples]
        kwargs = {"samples": samples, "num_classes": C, "test_time": False}
        for i, e in enumerate(emb):
            kwargs[f"embedding_a{i}"] = e
        models = [FakeField(C, last=False), FakeField(C, last=True)]
        torch.manual_seed(SEED + 3)
        u = torch.rand(n)
        torch.manual_seed(SEED + 3)
        res = rn.render(models, o, d, **kwargs)
        cases[tag + "_samples"] = np.asarray(samples, np.int64)
        cases[tag + "_t_rand_u"] = u
        for i, e in enumerate(emb):
            cases[f"{tag}_emb{i}"] = e
        for k, v in res.items():
            if torch.is_tensor(v):
                cases[f"{tag}_{k}"] = v
    npz("g3_render_nocuda.npz", **cases)


def g4_activations(cf):
    torch.manual_seed(SEED + 4)
    x = torch.cat([torch.randn(200) * 3, torch.tensor([-20.0, -7.0, 0.0, 7.0, 15.0, 20.0])])
    g = torch.randn_like(x)
    cases = {"x": x, "g": g}
    for name, fn in (("trunc_exp", cf.TruncExp), ("relu", cf.ReLU), ("trunc_tanh", cf.TruncTanh)):
        xi = x.clone().requires_grad_(True)
        y = fn.apply(xi)
        y.backward(g)
        cases[name + "_y"], cases[name + "_dx"] = y.detach(), xi.grad
    npz("g4_activations.npz", **cases)


def g5_raymarcher_bw(cf):
    torch.manual_seed(SEED + 5)
    counts = torch.tensor([3, 0, 5, 1, 0, 7, 2])
    starts = torch.cumsum(counts, 0) - counts
    rays_a = torch.stack([torch.arange(len(counts)), starts, counts], 1).long()
    N = int(counts.sum())
    ts = torch.rand(N)
    dxyz, ddirs = torch.randn(N, 3), torch.randn(N, 3)

    class Ctx:
        saved_tensors = (rays_a, ts)

    out = cf.RayMarcher.backward(Ctx, None, dxyz, ddirs, None, None, None)
    npz("g5_raymarcher_bw.npz", rays_a=rays_a, ts=ts, dL_dxyzs=dxyz, dL_ddirs=ddirs,
        dL_drays_o=out[0], dL_drays_d=out[1])


def table_rule(n, amp=0.6):
    """deterministic pseudo-random table of n floats in [-amp/2, amp/2): value_i = frac(i * phi32) - 0.5,
    reproduced by the tests instead of storing 200 M numbers"""
    i = np.arange(n, dtype=np.uint64)
    return (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 4294967296.0 - 0.5).astype(
        np.float32) * np.float32(amp)


def g6_ngp_field():
    """The reference's OWN models/networks.py::NGP (forward / forward_test / density) evaluated on the
    CPU: `tinycudann` is the pure-torch stand-in of tcnn_cpu_shim.py (checked against the C oracle
    below), `vren` an empty module (the field's forward does not touch it).  Pins the PYTHON wiring of
    the field: input normalisation, d(sigma)/dx normals through autograd, -F.normalize, head inputs,
    SH of (normalize(d)+1)/2, column order of rgb_net's input, activations, forward_test's swap."""
    sys.path.insert(0, OUT)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import tcnn_cpu_shim
    import oracle
    sys.modules["tinycudann"] = tcnn_cpu_shim
    from models import networks as ref_net
    cases = {}
    for tag, kw in (("a", {"scale": 0.5}), ("b", {"scale": 8.0, "embed_a": True, "embed_a_len": 8})):
        torch.manual_seed(SEED + 6)
        model = ref_net.NGP(**kw)
        g = np.random.default_rng(SEED + 7)
        with torch.no_grad():
            model.xyz_encoder.params.copy_(torch.from_numpy(table_rule(model.xyz_encoder.params.numel())))
            model.rgb_encoder.params.copy_(torch.from_numpy(table_rule(model.rgb_encoder.params.numel())))
            for name, p in model.named_parameters():
                if name.startswith("xyz_net") or name in ("rgb_net.params", "norm_pred_header.params",
                                                          "semantic_header.params"):
                    p.copy_(torch.from_numpy((g.standard_normal(p.shape) * 0.15).astype(np.float32)))
                    cases[f"{tag}_{name}"] = p.detach().clone()
        # the stand-in encoder must be the C oracle's encoder (same level rule, index rule, layout)
        desc, n_params = oracle.grid_layout(16, 8, 19, 16, float(np.exp(np.log(2048 * kw["scale"] / 16) / 15)))
        assert n_params == model.xyz_encoder.params.numel()
        xt = g.random((64, 3)).astype(np.float32)
        a = model.xyz_encoder(torch.from_numpy(xt)).detach().numpy()
        b = oracle.grid_fwd(desc, model.xyz_encoder.params.detach().numpy(), xt)
        assert np.abs(a - b).max() < 2e-6 * max(1.0, np.abs(b).max()), np.abs(a - b).max()

        n = 192
        s = kw["scale"]
        x = ((g.random((n, 3)) - 0.5) * 2 * s * 0.95).astype(np.float32)
        x[: n // 2] *= 0.3 / s if s > 1 else 1.0
        d = g.standard_normal((n, 3)).astype(np.float32)
        kwargs = {}
        if kw.get("embed_a"):
            emb = g.standard_normal((n, 8)).astype(np.float32)
            kwargs["embedding_a"] = torch.from_numpy(emb)
            cases[f"{tag}_embedding_a"] = emb
        cases[f"{tag}_x"], cases[f"{tag}_d"], cases[f"{tag}_scale"] = x, d, np.float64(s)
        outs = model(torch.from_numpy(x.copy()), torch.from_numpy(d), **kwargs)
        for k, v in zip(("sigmas", "rgbs", "normals_raw", "normals_pred", "semantic"), outs):
            cases[f"{tag}_fwd_{k}"] = v.detach()
        outs = model.forward_test(torch.from_numpy(x.copy()), torch.from_numpy(d), **kwargs)
        for k, v in zip(("sigmas", "rgbs", "normals_pred", "normals_raw", "semantic"), outs):
            cases[f"{tag}_test_{k}"] = v.detach()
        with torch.no_grad():
            cases[f"{tag}_density"] = model.density(torch.from_numpy(x.copy()))
    # skybox branch (networks.py:128-148, 284-291): SH degree 3 of the normalised direction -> 32 -> 3, sigmoid
    torch.manual_seed(SEED + 6)
    model = ref_net.NGP(scale=0.5, use_skybox=True)
    with torch.no_grad():
        p = model.skybox_rgb_net.params
        p.copy_(torch.from_numpy((g.standard_normal(p.shape) * 0.3).astype(np.float32)))
        cases["sky_params"] = p.detach().clone()
        dsky = g.standard_normal((100, 3)).astype(np.float32)
        cases["sky_d"] = dsky
        cases["sky_rgb"] = model.forward_skybox(torch.from_numpy(dsky))
    npz("g6_ngp_field.npz", **cases)


def install_oracle_vren(oracle):
    """fills the (so far empty) stand-in `vren` module with the CPU oracle's restatements, in the
    reference's calling convention (torch tensors in / out, in-place updates where the CUDA extension
    updates in place), so that the reference's own models/rendering.py can run on the CPU"""
    v = sys.modules["vren"]
    T = torch.from_numpy
    A = lambda t: t.detach().contiguous().numpy()

    def ray_aabb_intersect(o, d, c, h, max_hits):
        return [T(x) for x in oracle.ray_aabb_intersect(A(o), A(d), A(c), A(h), max_hits)]

    def raymarching_train(o, d, hits_t, bits, cascades, scale, esf, noise, G, max_samples):
        return [T(x) for x in oracle.raymarching_train(A(o), A(d), A(hits_t), A(bits), cascades, scale, esf, A(noise),
                                                       G, max_samples)]

    def raymarching_test(o, d, hits_t, alive, bits, cascades, scale, esf, G, max_samples, n_samples):
        assert hits_t.is_contiguous()
        return [T(x) for x in oracle.raymarching_test(A(o), A(d), hits_t.numpy(), A(alive), A(bits), cascades, scale,
                                                      esf, G, max_samples, n_samples)]

    def composite_train_fw(sig, rgbs, nrm, sems, deltas, ts, rays_a, thr, classes):
        return [T(x) for x in oracle.composite_train_fw(A(sig), A(rgbs), A(nrm), A(sems), A(deltas), A(ts), A(rays_a),
                                                        thr, classes)]

    def composite_test_fw(sig, rgbs, nrm, nrm_raw, sems, deltas, ts, hits_t, alive, thr, classes, n_eff, opacity, depth,
                          rgb, normal, normal_raw, sem):
        for t in (alive, opacity, depth, rgb, normal, normal_raw, sem):
            assert t.is_contiguous()
        oracle.composite_test_fw(A(sig), A(rgbs), A(nrm), A(nrm_raw), A(sems), A(deltas), A(ts), A(hits_t), alive.numpy(),
                                 thr, classes, A(n_eff), opacity.numpy(), depth.numpy(), rgb.numpy(), normal.numpy(),
                                 normal_raw.numpy(), sem.numpy())

    def composite_refloss_fw(sig, ndiff, nori, deltas, ts, rays_a, thr):
        return [T(x) for x in oracle.composite_refloss_fw(A(sig), A(ndiff), A(nori), A(deltas), A(ts), A(rays_a), thr)]

    def composite_train_bw(dO, dD, dRGB, dN, dS, dWs, sig, rgbs, nrm, ws, deltas, ts, rays_a, opacity, depth, rgb, normal,
                           thr, classes):
        return [T(x) for x in oracle.composite_train_bw(A(dO), A(dD), A(dRGB), A(dN), A(dS), A(dWs), A(sig), A(rgbs),
                                                        A(nrm), A(ws), A(deltas), A(ts), A(rays_a), A(opacity), A(depth),
                                                        A(rgb), A(normal), thr, classes)]

    def composite_refloss_bw(dLo, dLp, sig, ndiff, nori, deltas, ts, rays_a, lo, lp, thr):
        return [T(x) for x in oracle.composite_refloss_bw(A(dLo), A(dLp), A(sig), A(ndiff), A(nori), A(deltas), A(ts),
                                                          A(rays_a), A(lo), A(lp), thr)]

    def distortion_loss_fw(ws, deltas, ts, rays_a):
        return [T(x) for x in oracle.distortion_loss_fw(A(ws), A(deltas), A(ts), A(rays_a))]

    def distortion_loss_bw(dL, wi, wti, ws, deltas, ts, rays_a):
        return T(oracle.distortion_loss_bw(A(dL), A(wi), A(wti), A(ws), A(deltas), A(ts), A(rays_a)))

    for f in (ray_aabb_intersect, raymarching_train, raymarching_test, composite_train_fw, composite_test_fw,
              composite_refloss_fw, composite_train_bw, composite_refloss_bw, distortion_loss_fw, distortion_loss_bw):
        setattr(v, f.__name__, f)


def g7_render_paths():
    """The reference's OWN models/rendering.py::render — train path and test path — run on the CPU:
    the field is the reference's NGP class on the pure-torch tinycudann stand-in (as in G6), the vren
    entry points are the C oracle's restatements (each of them bit-checked against the HIP kernels
    by the GPU tests).  Pins the PYTHON glue of the renderer: AABB + near clamp, marcher call and
    trimming, per-sample kwargs, compositor call, background, RefLoss inputs, and at test time the
    whole progressive loop (samples per round, alive list, in-place hits_t, normalisation, argmax)."""
    sys.path.insert(0, OUT)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import tcnn_cpu_shim
    import oracle
    sys.modules["tinycudann"] = tcnn_cpu_shim
    install_oracle_vren(oracle)
    from models import networks as ref_net
    from models import rendering as ref_render
    torch.manual_seed(SEED + 8)
    g = np.random.default_rng(SEED + 9)
    model = ref_net.NGP(scale=0.5)
    cases = {}
    with torch.no_grad():
        model.xyz_encoder.params.copy_(torch.from_numpy(table_rule(model.xyz_encoder.params.numel())))
        model.rgb_encoder.params.copy_(torch.from_numpy(table_rule(model.rgb_encoder.params.numel())))
        for name, p in model.named_parameters():
            if name.startswith("xyz_net") or name in ("rgb_net.params", "norm_pred_header.params",
                                                      "semantic_header.params"):
                p.copy_(torch.from_numpy((g.standard_normal(p.shape) * 0.15).astype(np.float32)))
                if name == "xyz_net.2.bias":
                    p.fill_(30.0)       # dense medium: rays saturate and terminate inside the occupied ball
                cases[name] = p.detach().clone()
        # occupancy: a ball of radius 0.3 plus a slab, packed through morton order like update_density_grid does
        G = model.grid_size
        c = np.stack(np.meshgrid(*[np.arange(G, dtype=np.int32)] * 3, indexing="ij"), -1).reshape(-1, 3)
        xyz = (c.astype(np.float32) + 0.5) / G - 0.5
        occ = ((xyz ** 2).sum(-1) < 0.3 ** 2) | ((np.abs(xyz[:, 2] + 0.38) < 0.03) & (np.abs(xyz[:, 0]) < 0.4))
        grid = np.zeros(G ** 3, np.float32)
        grid[oracle.morton3D(c)] = occ.astype(np.float32)
        bits = oracle.packbits(grid, 0.5)
        model.density_bitfield.copy_(torch.from_numpy(bits))
    cases["density_bitfield"] = bits
    n = 64
    o = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1) * 1.5
    tgt = (torch.rand(n, 3) - 0.5) * 0.9
    tgt[-6:] = torch.tensor([3.0, 3.0, 3.0])          # a few rays that miss the box
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    noise = torch.rand(n)
    cases["rays_o"], cases["rays_d"], cases["noise"] = o, d, noise
    real_rand_like = torch.rand_like
    torch.rand_like = lambda t, *a, **k: noise.clone()   # the one random draw of the train path (custom_functions.py:84)
    try:
        res = ref_render.render(model, o, d, exp_step_factor=0.0, num_classes=7)
    finally:
        torch.rand_like = real_rand_like
    for k, v in res.items():
        cases["train_" + k] = v.detach() if torch.is_tensor(v) else np.asarray(v)
    with torch.no_grad():
        res = ref_render.render(model, o, d, test_time=True, exp_step_factor=0.0, num_classes=7, T_threshold=1e-2)
    for k, v in res.items():
        cases["test_" + k] = v.detach() if torch.is_tensor(v) else np.asarray(v)
    npz("g7_render_paths.npz", **cases)

    # ---- G8: one whole training step of the reference: render -> NeRFLoss (losses.py) -> sum of term
    # means (train.py:307) -> backward through the reference's autograd Functions -> gradient clipping at
    # 50 -> torch.optim.Adam(lr, eps=1e-8) (train.py:244,435).  Same model / rays / noise as G7.
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_losses", REF + "/losses.py")
    ref_losses = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_losses)
    step = {k: v for k, v in cases.items() if not (k.startswith("train_") or k.startswith("test_"))}
    gt = torch.rand(n, 3)
    step["rgb_gt"] = gt
    torch.rand_like = lambda t, *a, **k: noise.clone()
    try:
        res = ref_render.render(model, o, d, exp_step_factor=0.0, num_classes=7)
    finally:
        torch.rand_like = real_rand_like
    loss_d = ref_losses.NeRFLoss()(res, {"rgb": gt})
    loss = sum(lo.mean() for lo in loss_d.values())
    for k, v in loss_d.items():
        step["loss_" + k] = v.mean().detach()
    step["loss"] = loss.detach()
    loss.backward()
    small = ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_net.params",
             "norm_pred_header.params", "semantic_header.params")
    named = dict(model.named_parameters())
    for k in small:
        step["grad_" + k] = named[k].grad.clone()
    for k in ("xyz_encoder.params", "rgb_encoder.params"):
        gr = named[k].grad.numpy()
        idx = np.argpartition(np.abs(gr), -4096)[-4096:]            # the 4096 largest entries ...
        idx = np.concatenate([idx, (np.arange(4096, dtype=np.int64) * 7919) % gr.size])   # ... and 4096 fixed ones
        step["grad_idx_" + k], step["grad_val_" + k] = idx, gr[idx]
        step["grad_l2_" + k], step["grad_l1_" + k] = np.float64(np.sqrt((gr.astype(np.float64) ** 2).sum())), np.float64(
            np.abs(gr).sum(dtype=np.float64))
    params = [p for p in model.parameters()]
    step["grad_norm"] = torch.nn.utils.clip_grad_norm_(params, 50.0)
    opt = torch.optim.Adam(params, 1e-2, eps=1e-8)
    opt.step()
    for k in small:
        step["new_" + k] = named[k].detach().clone()
    for k in ("xyz_encoder.params", "rgb_encoder.params"):
        step["new_val_" + k] = named[k].detach().numpy()[step["grad_idx_" + k]]
    npz("g8_train_step.npz", **step)

    # ---- G10: the --normal_ref recipe (losses.py:107-109): Ro / Rp enter the loss, so the gradient runs
    # through RefLoss.backward and through normals_raw = -normalize(d sigma / dx), i.e. through the
    # DOUBLE backward of the density encoder and MLP (networks.py:186-196, create_graph=True)
    model.zero_grad(set_to_none=True)
    with torch.no_grad():     # back to the parameters the fixture documents (the Adam step above moved them)
        model.xyz_encoder.params.copy_(torch.from_numpy(table_rule(model.xyz_encoder.params.numel())))
        model.rgb_encoder.params.copy_(torch.from_numpy(table_rule(model.rgb_encoder.params.numel())))
        for k in small:
            named[k].copy_(step[k])
    ref = {k: step[k] for k in small + ("density_bitfield", "rays_o", "rays_d", "noise", "rgb_gt")}
    torch.rand_like = lambda t, *a, **k: noise.clone()
    try:
        res = ref_render.render(model, o, d, exp_step_factor=0.0, num_classes=7)
    finally:
        torch.rand_like = real_rand_like
    loss_d = ref_losses.NeRFLoss()(res, {"rgb": gt}, normal_ref=True)
    loss = sum(lo.mean() for lo in loss_d.values())
    for k, v in loss_d.items():
        ref["loss_" + k] = v.mean().detach()
    ref["loss"] = loss.detach()
    loss.backward()
    for k in small:
        ref["grad_" + k] = named[k].grad.clone()
    for k in ("xyz_encoder.params", "rgb_encoder.params"):
        gr = named[k].grad.numpy()
        idx = np.argpartition(np.abs(gr), -4096)[-4096:]
        ref["grad_idx_" + k], ref["grad_val_" + k] = idx, gr[idx]
        ref["grad_l2_" + k] = np.float64(np.sqrt((gr.astype(np.float64) ** 2).sum()))
    npz("g10_normal_ref_step.npz", **ref)


def noise_rule(shape, call_index):
    """deterministic stand-in for torch.rand_like in update_density_grid: u_i = frac((i + 1 + 7919 call) * phi)"""
    n = int(np.prod(shape))
    i = np.arange(1, n + 1, dtype=np.float64) + 7919.0 * call_index
    return torch.from_numpy(np.mod(i * 0.6180339887498949, 1.0).astype(np.float32).reshape(shape))


def g9_density_grid_update():
    """The reference's OWN NGP.update_density_grid (networks.py:379-408, warm-up branch: every cell) run
    twice on the CPU — cell centres, jitter, density(), EMA with decay, mean threshold, packbits in
    morton order (vren.morton3D / packbits = the C oracle).  The jitter draws are replaced by
    noise_rule so that the GPU test can feed the same numbers."""
    sys.path.insert(0, OUT)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import tcnn_cpu_shim
    import oracle
    sys.modules["tinycudann"] = tcnn_cpu_shim
    v = sys.modules["vren"]
    v.morton3D = lambda c: torch.from_numpy(oracle.morton3D(c.numpy()))
    v.morton3D_invert = lambda i: torch.from_numpy(oracle.morton3D_invert(i.numpy()))

    def packbits(grid, thr, bitfield):
        bitfield.copy_(torch.from_numpy(oracle.packbits(grid.detach().contiguous().numpy().reshape(-1), float(thr))))
    v.packbits = packbits
    from models import networks as ref_net
    torch.manual_seed(SEED + 10)
    g = np.random.default_rng(SEED + 11)
    model = ref_net.NGP(scale=0.5)
    cases = {}
    with torch.no_grad():
        model.xyz_encoder.params.copy_(torch.from_numpy(table_rule(model.xyz_encoder.params.numel())))
        for name, p in model.named_parameters():
            if name.startswith("xyz_net"):
                p.copy_(torch.from_numpy((g.standard_normal(p.shape) * 0.15).astype(np.float32)))
                if name == "xyz_net.2.bias":
                    p.fill_(2.0)
                cases[name] = p.detach().clone()
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3))
    model.register_buffer("grid_coords", torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32)] * 3,
                                                                    indexing="ij"), -1).reshape(-1, 3))
    real = torch.rand_like
    thr = 0.01 * 1024 / 3 ** 0.5
    try:
        for call in range(2):
            torch.rand_like = lambda t, *a, _c=call, **k: noise_rule(tuple(t.shape), _c)
            with torch.no_grad():
                model.update_density_grid(thr, warmup=True)
            dg = model.density_grid.numpy()
            cases[f"u{call}_grid_sub"] = dg[:, ::257].copy()
            cases[f"u{call}_mean_pos"] = np.float64(dg[dg > 0].mean())
            cases[f"u{call}_bitfield"] = model.density_bitfield.numpy().copy()
    finally:
        torch.rand_like = real
    cases["density_threshold"] = np.float64(thr)
    npz("g9_density_grid_update.npz", **cases)

    # ---- G11: mark_invisible_cells (networks.py:336-377) on a few cameras, two of them inside the box so
    # that the "too near to a camera" rule fires; scale 0.5 (one cascade) and scale 2 (three cascades)
    vis = {}
    for tag, scale in (("a", 0.5), ("b", 2.0)):
        torch.manual_seed(SEED + 12)
        m = ref_net.NGP(scale=scale)
        G = m.grid_size
        m.register_buffer("density_grid", torch.zeros(m.cascades, G ** 3))
        m.register_buffer("grid_coords", torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32)] * 3,
                                                                    indexing="ij"), -1).reshape(-1, 3))
        n_cam = 7
        pos = torch.nn.functional.normalize(torch.randn(n_cam, 3), dim=-1) * 1.5
        pos[-2:] *= 0.15                                   # cameras inside the scene box
        fwd = -torch.nn.functional.normalize(pos, dim=-1)
        up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
        right = torch.nn.functional.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
        down = torch.cross(fwd, right, dim=-1)
        poses = torch.stack([right, down, fwd, pos], -1)   # (n,3,4) c2w, [right down front]
        K = torch.tensor([[300.0, 0, 100.0], [0, 300.0, 100.0], [0, 0, 1.0]])
        m.mark_invisible_cells(K, poses, (200, 200))
        vis[tag + "_scale"], vis[tag + "_poses"], vis[tag + "_K"] = np.float64(scale), poses, K
        vis[tag + "_invisible_bits"] = np.packbits((m.density_grid.numpy() < 0).reshape(-1))
        vis[tag + "_count_sub"] = m.count_grid.numpy()[:, ::101].copy()
    npz("g11_invisible_cells.npz", **vis)


def g13_tonemapper():
    """M4 (networks.py:150-163, 229-240): the reference's OWN NGP(rgb_act='None') — rgb_net without output
    activation, log-radiance -> TruncExp with output_radiance=True, else the three per-channel tone-mapper
    networks (1 -> 64 -> 1, sigmoid).  networks.py calls self.log_radiance_to_rgb without defining it
    (SURVEY.md Appendix C); the method body exists in models/networks_noCUDA.py:238-259 and is bound onto
    the class here, unchanged (that module imports the non-existent models/rendering_old for one
    constant, which is stubbed).  Recorded: forward / forward_test with output_radiance, with the
    tone-mappers at unit exposure, and with a per-sample `exposure`."""
    sys.path.insert(0, OUT)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import tcnn_cpu_shim
    sys.modules["tinycudann"] = tcnn_cpu_shim
    stub = types.ModuleType("models.rendering_old")
    stub.NEAR_DISTANCE = 0.01
    sys.modules["models.rendering_old"] = stub
    from models import networks as ref_net
    from models import networks_noCUDA as ref_nocuda
    ref_net.NGP.log_radiance_to_rgb = ref_nocuda.NGP.log_radiance_to_rgb
    torch.manual_seed(SEED + 13)
    g = np.random.default_rng(SEED + 14)
    model = ref_net.NGP(scale=0.5, rgb_act='None')
    cases = {}
    with torch.no_grad():
        model.xyz_encoder.params.copy_(torch.from_numpy(table_rule(model.xyz_encoder.params.numel())))
        model.rgb_encoder.params.copy_(torch.from_numpy(table_rule(model.rgb_encoder.params.numel())))
        for name, p in model.named_parameters():
            if name.startswith("xyz_net") or name in ("rgb_net.params", "norm_pred_header.params",
                                                      "semantic_header.params") or name.startswith("tonemapper_net"):
                amp = 0.6 if name.startswith("tonemapper_net") else 0.15
                p.copy_(torch.from_numpy((g.standard_normal(p.shape) * amp).astype(np.float32)))
                cases[name] = p.detach().clone()
    n = 160
    x = ((g.random((n, 3)) - 0.5) * 0.95).astype(np.float32)
    d = g.standard_normal((n, 3)).astype(np.float32)
    exposure = (0.25 + 3.0 * g.random((n, 1))).astype(np.float32)
    cases["x"], cases["d"], cases["exposure"] = x, d, exposure
    X, D = torch.from_numpy(x), torch.from_numpy(d)
    for tag, kw in (("radiance", {"output_radiance": True}), ("ldr", {}), ("ldr_exposure", {"exposure": torch.from_numpy(exposure)})):
        outs = model(X.clone(), D, **kw)
        cases[f"fwd_{tag}_rgbs"], cases[f"fwd_{tag}_sigmas"] = outs[1].detach(), outs[0].detach()
        outs = model.forward_test(X.clone(), D, **kw)
        cases[f"test_{tag}_rgbs"] = outs[1].detach()
    # the unit-exposure regulariser of train.py:301-306 evaluates the tone-mappers at zero log-radiance
    cases["unit_exposure_rgb"] = model.log_radiance_to_rgb(torch.zeros(1, 3), exposure=torch.ones(1, 1)).detach()
    npz("g13_tonemapper.npz", **cases)


def g14_loss_terms():
    """The reference's OWN losses.py::NeRFLoss with every optional term switched on (losses.py:107-132:
    normal_ref, normal_mono, semantic + sky_depth, depth_mono) on synthetic per-ray results / targets, and
    compute_scale_and_shift; vren's distortion entry points = the C oracle.  Values and the gradients of
    sum(term.mean()) (train.py:307) w.r.t. every differentiable result are recorded."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import oracle
    install_oracle_vren(oracle)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_losses", REF + "/losses.py")
    ref_losses = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_losses)
    torch.manual_seed(SEED + 15)
    nr, C = 200, 7
    counts = torch.randint(0, 12, (nr,))
    counts[::17] = 0
    starts = torch.cumsum(counts, 0) - counts
    rays_a = torch.stack([torch.arange(nr), starts, counts], 1).long()
    N = int(counts.sum())
    res = {
        "rgb": torch.rand(nr, 3), "opacity": torch.rand(nr) * 0.98 + 0.01, "depth": torch.rand(nr) * 4,
        "normal_pred": torch.randn(nr, 3), "semantic": torch.randn(nr, C),
        "ws": torch.rand(N) * 0.2, "deltas": torch.rand(N) * 0.01 + 1e-3, "ts": torch.sort(torch.rand(N) * 3)[0],
        "rays_a": rays_a, "Ro": torch.rand(nr), "Rp": torch.rand(nr, 3),
    }
    label = torch.randint(0, C, (nr,))
    label[::9] = 4            # sky
    label[5::23] = 256        # ignore_index
    depth_gt = torch.rand(nr) * 60
    depth_gt[::7] = 0         # invalid depth
    tgt = {"rgb": torch.rand(nr, 3), "normal": torch.randn(nr, 3), "label": label, "depth": depth_gt}
    diff = ("rgb", "opacity", "depth", "normal_pred", "semantic", "ws", "Ro", "Rp")
    for k in diff:
        res[k].requires_grad_(True)
    kw = dict(normal_ref=True, normal_mono=True, semantic=True, depth_mono=True, scale=0.5)
    loss_d = ref_losses.NeRFLoss()(res, tgt, **kw)
    loss = sum(lo.mean() for lo in loss_d.values())
    loss.backward()
    cases = {"in_" + k: v.detach() for k, v in res.items()}
    cases.update({"tgt_" + k: v for k, v in tgt.items()})
    cases.update({"term_" + k: v.detach() for k, v in loss_d.items()})
    cases["loss"] = loss.detach()
    cases.update({"grad_" + k: res[k].grad for k in diff})
    valid = depth_gt / 25 > 0
    sc, sh = ref_losses.compute_scale_and_shift(res["depth"][valid].detach(), (depth_gt / 25)[valid])
    cases["scale_shift"] = torch.stack([sc, sh])
    cases["scene_scale"] = np.float64(kw["scale"])
    npz("g14_loss_terms.npz", **cases)


if __name__ == "__main__":
    cf, rn = import_reference()
    only = set(sys.argv[1:])     # e.g. `make_golden.py g13 g14` regenerates just those fixtures
    if only:
        for name in sorted(only):
            {"g13": g13_tonemapper, "g14": g14_loss_terms}[name]()
        raise SystemExit(0)
    g1_raw2outputs(cf)
    g2_sample_pdf(cf)
    g3_render(cf, rn)
    g4_activations(cf)
    g5_raymarcher_bw(cf)
    g6_ngp_field()
    g7_render_paths()
    g9_density_grid_update()
    g13_tonemapper()
    g14_loss_terms()
