"""Length and sampling shared by every loader (reference: datasets/base.py:5-64)."""
import torch


class BaseDataset(torch.utils.data.Dataset):
    """A training item is one random batch of (image, pixel) pairs — the loop draws 1000 of them per
    epoch; a test item is one whole image.  Per-ray supervision tensors are optional attributes:
    `rays` (N_img, h*w, 3|4), `labels`, `depths_2d`, `normals`, all indexed [image, pixel].

    Unlike the reference (numpy RNG on the host, tensors on the CPU) everything may live on the
    GPU (`device=`): indices come from torch.randint on the tensors' device, so a batch costs two
    small launches and no host round trip."""

    def __init__(self, root_dir, split='train', downsample=1.0):
        self.root_dir = root_dir
        self.split = split
        self.downsample = downsample
        self.batch_size = 8192
        self.ray_sampling_strategy = 'all_images'

    def read_intrinsics(self):
        raise NotImplementedError

    def __len__(self):
        return 1000 if self.split.startswith('train') else len(self.poses)

    def _extras(self):
        return [(key, getattr(self, attr)) for key, attr in (('label', 'labels'), ('depth', 'depths_2d'), ('normal', 'normals'))
                if hasattr(self, attr)]

    def __getitem__(self, idx):
        if self.split.startswith('train'):
            dev = self.rays.device
            n_img = len(self.poses)
            if self.ray_sampling_strategy == 'all_images':
                img_idxs = torch.randint(n_img, (self.batch_size,), device=dev)
            else:  # 'same_image'
                img_idxs = torch.randint(n_img, (1,), device=dev).expand(self.batch_size).contiguous()
            w, h = self.img_wh
            pix_idxs = torch.randint(w * h, (self.batch_size,), device=dev)
            rays = self.rays[img_idxs, pix_idxs]
            # the reference names these the other way round (u = pix // w is the row), kept as is
            uv = torch.stack([pix_idxs // w, pix_idxs % w], -1)
            sample = {'img_idxs': img_idxs, 'pix_idxs': pix_idxs, 'uv': uv, 'rgb': rays[:, :3]}
            for key, src in self._extras():
                sample[key] = src[img_idxs.to(src.device), pix_idxs.to(src.device)]
            if self.rays.shape[-1] == 4:  # HDR-NeRF data
                sample['exposure'] = rays[:, 3:]
        else:
            sample = {'pose': self.poses[idx], 'img_idxs': idx}
            if len(self.rays) > 0:  # ground truth available
                rays = self.rays[idx]
                sample['rgb'] = rays[:, :3]
                for key, src in self._extras():
                    if key != 'normal':
                        sample[key] = src[idx]
                if rays.shape[1] == 4:
                    sample['exposure'] = rays[0, 3]  # one exposure per image
        return sample

    def batch_rays(self, sample):
        """(rays_o, rays_d) of a train sample, as NeRFSystem.forward does (train.py:136-155)"""
        from .ray_utils import get_rays
        dev = self.directions.device
        return get_rays(self.directions[sample['pix_idxs'].to(dev)], self.poses[sample['img_idxs'].to(dev)])

    def get_path_rays(self, c2w_list):
        """{i: (h*w, 6) [origin | direction]} for a list of poses (tnt.py:254-266, colmap.py:238-247)"""
        from .ray_utils import get_rays
        rays = {}
        for i, pose in enumerate(c2w_list):
            c2w = torch.as_tensor(pose, dtype=torch.float32)[:3].to(self.directions.device)
            rays[i] = torch.cat(get_rays(self.directions, c2w), 1).cpu()
        return rays
