"""Worker of test_gpu_parity.py::test_two_rank_sharded_training_matches_single_rank — launched with
torch.distributed.run, 2 ranks on ONE GPU over gloo (the RCCL path needs one GPU per rank; the
collective pattern, the sharded optimizer and the 1/world folding are the same code) — and of
test_one_rank_rccl_sharded_training_matches_unsharded: ONE rank over RCCL (NGP_DIST_BACKEND=nccl,
NGP_FORCE_SHARDED=1), which sends the same reduce-scatter / all-reduce / all-gather calls through the
backend the multi-GPU bench uses.

Every rank trains STEPS steps on its own rays with the sharded optimizer (reduce-scatter of the
gradient, clip + Adam on the rank's slice, all-gather of the parameters).  Checks:
  * all ranks hold bit-identical parameters afterwards;
  * rank 0 repeats the run alone on the CONCATENATED ray batches (unsharded optimizer): DDP's
    average of per-rank mean losses over equal batches is the mean over the union, so the
    parameters must agree up to summation order.
Prints DP_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import ngp_amd  # noqa: F401
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

STEPS, RAYS = 5, 2048


def build(dev):
    torch.manual_seed(7)
    model = NGP(scale=0.5).to(dev)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    model.grid_rng = torch.Generator(device=dev).manual_seed(11)
    return model


def batches(scene, rank, dev):
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    out = []
    for _ in range(STEPS):
        img, pix = scene.sample_batch(RAYS, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=64)
        out.append((o, d, gt))
    return out


def run(trainer, data):
    # the marcher's jitter is the only random draw of a step: pin it to 0.5 for every ray so that a ray is
    # marched identically whichever batch (rank-local or union) it sits in
    torch.rand_like = lambda t, *a, **k: torch.full_like(t, 0.5)
    losses = []
    for o, d, gt in data:
        losses.append(float(trainer.step(o, d, gt)[0]))
    trainer.wait()
    torch.cuda.synchronize()
    return losses


def check_bucket0_order(tr, model, batch, cdev):
    """Bucket 0's reduce-scatter is fired from the colour encoder's backward, on the SIDE stream the colour
    scatter was forked to (networks._FieldFn.backward): it must see the complete colour-table gradient.  One
    backward with the hook (sharded path as trained), one with the hook off into a zeroed buffer; the rank's
    slice of the summed hook-less gradients must equal what the hooked reduce-scatter delivered."""
    from ngp_amd.losses import nerf_loss_and_grads
    from ngp_amd.rendering import render
    o, d, gt = batch
    tr.wait()
    world = dist.get_world_size()
    b0 = tr.buckets.bounds[1]
    a, b = tr.shards[0]

    def backward_only():
        res = render(model, o, d, exp_step_factor=0.0, num_classes=7)
        terms, (d_rgb, d_op, d_ws) = nerf_loss_and_grads(res["rgb"], res["opacity"], res["ws"], res["deltas"], res["ts"],
                                                        res["rays_a"], gt, 2e-4, 3e-4)
        torch.autograd.backward([res["rgb"], res["opacity"], res["ws"]], [d_rgb, d_op, d_ws])

    assert tr.hooked0
    tr.flat_grad.zero_()
    torch.cuda.synchronize()
    backward_only()                       # fires reduce_scatter_bucket(0, grad_shard[0]) from the encoder hook
    tr.buckets.wait()
    torch.cuda.synchronize()
    hooked = tr.grad_shard[0].clone()
    hook, model.rgb_encoder.on_grad_ready = model.rgb_encoder.on_grad_ready, None
    tr.flat_grad.zero_()
    torch.cuda.synchronize()
    backward_only()
    torch.cuda.synchronize()
    model.rgb_encoder.on_grad_ready = hook
    full = tr.flat_grad[:b0].to(cdev).clone()
    dist.all_reduce(full)
    mine = full[a:b].to(hooked.device)
    tr.flat_grad.zero_()
    scale = float(mine.abs().max())
    err = float((mine - hooked).abs().max())
    nz = int((mine != 0).sum())
    print(f"rank{dist.get_rank()}: bucket-0 reduce-scatter vs complete colour-table gradient: max |diff| {err:.3e} "
          f"(max |g| {scale:.3e}, {nz} non-zero entries in the slice, world {world})", flush=True)
    return nz > 1000 and err <= 2e-5 * scale


def check_sharded_checkpoint(tr, model, batch, dev, rank, cdev):
    """a checkpoint loaded into a sharded-optimizer trainer must be what the next step starts from
    (NGPTrainer.load_ckpt -> sync_shards); a plain copy into the flat buffer would be overwritten by the stale
    master slices at the first all-gather."""
    import tempfile
    from ngp_amd import ckpt
    small = ("xyz_net.0.weight", "xyz_net.2.weight", "rgb_net.params")
    path = os.path.join(tempfile.gettempdir(), f"ngp_dp_ckpt_{os.environ.get('MASTER_PORT', '0')}.ckpt")
    if rank == 0:
        src = build(dev)
        with torch.no_grad():
            src.density_bitfield.copy_(model.density_bitfield)   # a slim checkpoint carries the occupancy bitfield
            for k, p in src.named_parameters():
                if k in small:
                    p.fill_(0.37)
        ckpt.save_ckpt(src, path)
        torch.save({"state_dict": ckpt.slim_ckpt(path)}, path)
        del src
    dist.barrier()
    tr.load_ckpt(path)
    o, d, gt = batch
    tr.step(o, d, gt)
    tr.wait()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    worst = max(float((named[k] - 0.37).abs().max()) for k in small)
    print(f"rank{rank}: after load_ckpt + one sharded step the MLP weights are within {worst:.3e} of the checkpoint "
          f"(lr 1e-2)", flush=True)
    dist.barrier()
    if rank == 0:
        os.remove(path)
    return worst <= 1.01e-2


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if os.environ.get("NGP_DIST_BACKEND", "gloo") == "nccl":   # one rank, RCCL: NGP_FORCE_SHARDED=1 rehearsal
        dist.init_process_group("nccl", device_id=dev)
        cdev = dev            # RCCL moves device tensors only
    else:
        dist.init_process_group("gloo")
        cdev = torch.device("cpu")
    rank, world = dist.get_rank(), dist.get_world_size()
    scene = LegoProxy(n_images=10, img_wh=(100, 100), device=dev)
    model = build(dev)
    tr = NGPTrainer(model, lr=1e-2)
    assert tr.sharded and tr.buckets.world == world
    tr.broadcast_state(0)
    mine = batches(scene, rank, dev)
    losses = run(tr, mine)
    assert all(l == l and l < 10 for l in losses), losses
    flat = tr.flat_param.detach().to(cdev)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for r in range(1, world):
        assert torch.equal(gathered[0], gathered[r]), f"rank {r} diverged: {(gathered[0]-gathered[r]).abs().max()}"
    lt = torch.tensor(losses, dtype=torch.float64, device=cdev)
    all_losses = [torch.empty_like(lt) for _ in range(world)]
    dist.all_gather(all_losses, lt)
    mean_losses = torch.stack(all_losses).mean(0).cpu()            # DDP: the mean over ranks of the per-rank mean losses
    small = ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_net.params")
    sharded_params = {k: v.detach().clone() for k, v in model.named_parameters() if k in small}
    ok = True
    ok = check_bucket0_order(tr, model, mine[0], cdev) and ok
    ok = check_sharded_checkpoint(tr, model, mine[0], dev, rank, cdev) and ok
    sub = dist.new_group([0])          # (collective) 1-rank subgroup for the unsharded comparison run on rank 0
    if rank == 0:
        del tr
        model1 = build(dev)
        tr1 = NGPTrainer(model1, lr=1e-2, group=sub, force_sharded=False)
        assert not tr1.sharded
        others = [batches(scene, r, dev) for r in range(world)]
        union = [(torch.cat([others[r][i][0] for r in range(world)]), torch.cat([others[r][i][1] for r in range(world)]),
                  torch.cat([others[r][i][2] for r in range(world)])) for i in range(STEPS)]
        losses1 = torch.tensor(run(tr1, union), dtype=torch.float64)
        rel_loss = float(((mean_losses - losses1).abs() / losses1).max())
        worst = 0.0
        for k, v in model1.named_parameters():
            if k in small:
                # a parameter moves by ~lr per Adam step: compare the displacement from the common start
                d = float((sharded_params[k] - v.detach()).abs().mean())
                worst = max(worst, d)
        print(f"rank0: max relative loss difference over the steps {rel_loss:.2e}; worst mean |param difference| {worst:.2e} "
              f"(lr = 1e-2, {STEPS} steps)", flush=True)
        ok = rel_loss < 2e-3 and worst < 2e-3
    flag = torch.tensor([1 if ok else 0], device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        print("DP_OK" if int(flag) == 1 else "DP_FAIL", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
