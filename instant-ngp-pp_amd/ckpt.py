"""Checkpoint helpers with the reference's layout (utils.py:7-42): Lightning-style
`{'state_dict': {'model.<key>': tensor, ...}}` files, the model's keys under a `model.` prefix,
tcnn modules as flat `*.params` vectors.  Loading uses weights_only=True (no unpickling of code)."""
import torch


def extract_model_state_dict(ckpt_path, model_name='model', prefixes_to_ignore=()):
    checkpoint = torch.load(ckpt_path, map_location='cpu', weights_only=True)
    if 'state_dict' in checkpoint:  # pytorch-lightning checkpoint
        checkpoint = checkpoint['state_dict']
    out = {}
    for k, v in checkpoint.items():
        if not k.startswith(model_name):
            continue
        k = k[len(model_name) + 1:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k] = v
    return out


def load_ckpt(model, ckpt_path, model_name='model', prefixes_to_ignore=()):
    """in-place update of `model` from a (slim) checkpoint; parameter storage (e.g. the trainer's
    flat buffer views) is preserved because values are copied, not rebound.  Strict like the reference's
    load_state_dict (utils.py:24-29): unknown keys and shape mismatches raise.  With a sharded-optimizer
    trainer use NGPTrainer.load_ckpt (it refreshes the optimizer's master slices afterwards)."""
    if not ckpt_path:
        return
    state = extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore)
    own = model.state_dict()
    missing = [k for k in state if k not in own]
    if missing:
        raise KeyError(f"checkpoint keys not in the model: {missing[:5]}")
    bad = [(k, tuple(v.shape), tuple(own[k].shape)) for k, v in state.items() if tuple(v.shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError("size mismatch for " + ", ".join(f"{k}: checkpoint {a} vs model {b}" for k, a, b in bad[:5]))
    with torch.no_grad():
        for k, v in state.items():
            own[k].copy_(v.to(own[k].device))


def save_ckpt(model, path, extra=None):
    """writes {'state_dict': {'model.<key>': ...}} like Lightning's ModelCheckpoint(save_weights_only)"""
    sd = {f"model.{k}": v.detach().cpu() for k, v in model.state_dict().items()}
    if extra:
        sd.update(extra)
    torch.save({'state_dict': sd}, path)


def slim_ckpt(ckpt_path, save_poses=False):
    """drops what inference does not need (utils.py:32-42)"""
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=True)
    keys_to_pop = ['directions', 'model.density_grid', 'model.grid_coords']
    if not save_poses:
        keys_to_pop += ['poses']
    keys_to_pop += [k for k in ckpt['state_dict'] if k.startswith('val_lpips')]
    for k in keys_to_pop:
        ckpt['state_dict'].pop(k, None)
    return ckpt['state_dict']
