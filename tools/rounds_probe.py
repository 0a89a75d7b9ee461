import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ngp_amd
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer
from ngp_amd import rendering
dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
scene = LegoProxy(img_wh=(800, 800), device=dev)
tr = NGPTrainer(model, lr=1e-2)
gen = torch.Generator(device=dev).manual_seed(1)
for i in range(600):
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=128)
    tr.step(o, d, gt)
tr.wait(); torch.cuda.synchronize()
n = 800 * 800
pix = torch.arange(n, device=dev)
img = torch.full((n,), 1, dtype=torch.long, device=dev)
o, d = scene.rays(img, pix)
# instrument the host loop: wrap call("raymarching_test")
log = []
orig = rendering.call
def spy(name, *a):
    if name == "raymarching_test":
        torch.cuda.synchronize()
        log.append([time.perf_counter(), a[11], a[10]])   # N_alive, N_samples  (positional ints)
    return orig(name, *a)
rendering.call = spy
with torch.no_grad():
    out = rendering.render(model, o, d, test_time=True, exp_step_factor=0.0, T_threshold=1e-2, host_test_loop=True)
torch.cuda.synchronize()
tend = time.perf_counter()
print("rounds", len(log))
for i, (t, na, ns) in enumerate(log):
    t1 = log[i + 1][0] if i + 1 < len(log) else tend
    print(i, "N_alive", na, "N_samples", ns, "rows", na * ns, "ms %.3f" % ((t1 - t) * 1e3))
