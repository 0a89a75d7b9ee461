#!/usr/bin/env python3
"""Trains on a scene directory with the reference's recipe and evaluates the test split (what
`python train.py --root_dir ... --dataset_name nerf|nsvf|colmap|tnt|nerfpp` + validation do in the
reference: train.py:82-392).  GPU only.

  python tools/train_dataset.py --root_dir /data/nerf_synthetic/lego --num_epochs 20
  python tools/train_dataset.py --root_dir /data/tnt/Playground --dataset_name tnt --scale 8 --exp_step_factor 0.00390625
  python tools/train_dataset.py --make_proxy /tmp/proxy --downsample 0.25 --num_epochs 2   # no dataset at hand
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd  # noqa: F401
from ngp_amd import ckpt
from ngp_amd.datasets import dataset_dict, get_rays, write_synthetic_dataset
from ngp_amd.metrics import psnr
from ngp_amd.networks import NGP
from ngp_amd.rendering import render
from ngp_amd.trainer import NGPTrainer


def build_model(scale, device):
    model = NGP(scale=scale).to(device)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=device))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=device)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    return model


def train(model, train_set, num_epochs, steps_per_epoch, batch_size, lr, log_every=0, exp_step_factor=0.0,
          render_kwargs=None):
    """the reference's schedule (NGPTrainer) fed by the dataset's own sampler, one batch ahead"""
    train_set.batch_size = batch_size
    trainer = NGPTrainer(model, lr=lr, num_epochs=num_epochs, steps_per_epoch=steps_per_epoch,
                         exp_step_factor=exp_step_factor, render_kwargs=render_kwargs)

    def next_batch():
        s = train_set[0]
        o, d = train_set.batch_rays(s)
        return o.contiguous(), d.contiguous(), s["rgb"].contiguous()

    import gc
    gc.collect()
    gc.freeze()   # model, dataset and trainer live for the whole run: keep full collections cheap
    cur = next_batch()
    total = num_epochs * steps_per_epoch
    t0 = time.perf_counter()
    for i in range(total):
        nxt = next_batch() if i + 1 < total else None
        loss, res = trainer.step(*cur, next_rays=None if nxt is None else nxt[:2])
        if log_every and (i + 1) % log_every == 0:
            torch.cuda.synchronize()
            print(json.dumps({"step": i + 1, "loss": float(loss), "train_psnr": float(psnr(res["rgb"].detach(), cur[2])),
                              "rays_per_s": batch_size * (i + 1) / (time.perf_counter() - t0)}), flush=True)
        cur = nxt
    trainer.wait()
    return trainer


@torch.no_grad()
def evaluate(model, test_set, chunk=131072, save_dir=None, exp_step_factor=0.0):
    """per-image PSNR of the test split through render(test_time=True) (train.py:347-392)"""
    w, h = test_set.img_wh
    out = []
    for i in range(len(test_set)):
        s = test_set[i]
        o, d = get_rays(test_set.directions, s["pose"])
        o, d = o.contiguous(), d.contiguous()
        rgb = torch.cat([render(model, o[j:j + chunk], d[j:j + chunk], test_time=True, T_threshold=1e-2,
                                exp_step_factor=exp_step_factor)["rgb"]
                         for j in range(0, o.shape[0], chunk)], 0).clamp(0, 1)
        out.append(float(psnr(rgb, s["rgb"])))
        if save_dir:
            from PIL import Image
            os.makedirs(save_dir, exist_ok=True)
            Image.fromarray((rgb.reshape(h, w, 3).cpu().numpy() * 255 + 0.5).astype("uint8")).save(
                os.path.join(save_dir, f"{i:03d}.png"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_dir")
    ap.add_argument("--dataset_name", default="nerf", choices=sorted(dataset_dict))
    ap.add_argument("--make_proxy", help="write the analytic lego-proxy scene to this directory first and train on it")
    ap.add_argument("--downsample", type=float, default=1.0)
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--exp_step_factor", type=float, default=0.0, help="1/256 for unbounded scenes (opt.py)")
    ap.add_argument("--random_bg", action="store_true")
    ap.add_argument("--batch_size", type=int, default=8192)
    ap.add_argument("--num_epochs", type=int, default=20)
    ap.add_argument("--steps_per_epoch", type=int, default=1000)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--save_dir")
    ap.add_argument("--ckpt_path")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(20220806)
    root = args.root_dir
    if args.make_proxy:
        from ngp_amd.synthetic import LegoProxy
        wh = int(800 * args.downsample)
        scene = LegoProxy(n_images=108, img_wh=(wh, wh), device=dev)
        root = write_synthetic_dataset(args.make_proxy, scene, n_train=100, n_test=8, rgba=False)
    loader = dataset_dict[args.dataset_name]
    train_set = loader(root, "train", args.downsample, device=dev)
    test_set = loader(root, "test", args.downsample, device=dev)
    model = build_model(args.scale, dev)
    t0 = time.perf_counter()
    train(model, train_set, args.num_epochs, args.steps_per_epoch, args.batch_size, args.lr, log_every=500,
          exp_step_factor=args.exp_step_factor, render_kwargs={"random_bg": True} if args.random_bg else None)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    psnrs = evaluate(model, test_set, save_dir=args.save_dir, exp_step_factor=args.exp_step_factor)
    if args.ckpt_path:
        ckpt.save_ckpt(model, args.ckpt_path)
    print(json.dumps({"train_s": t_train, "test_psnr_mean": sum(psnrs) / len(psnrs), "test_psnr": psnrs,
                      "steps": args.num_epochs * args.steps_per_epoch, "img_wh": train_set.img_wh}))


if __name__ == "__main__":
    main()
