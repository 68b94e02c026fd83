"""Dataset glue of the Instant-NGP path, device-resident.

Mirrors, for Blender-format scenes, what the reference spreads over
  /root/reference/xrnerf/datasets/load_data/load_blender.py:32-89   (load_blender_data)
  /root/reference/xrnerf/datasets/load_data/load.py:51-68,177-190    (load_data, blender branch)
  /root/reference/xrnerf/datasets/hashnerf_dataset.py:13-150         (HashNerfDataset)
  /root/reference/xrnerf/datasets/load_data/get_rays.py:72-98        (load_rays_hash)
  /root/reference/xrnerf/datasets/pipelines/create.py:153-191, augment.py:290-317 (HashBatchSample, RandomBGColor)
with two MI355X-first changes: the [N*H*W, 11] ray table (o3, d3, rgba4, img_id) is generated ON THE DEVICE
with xr_gen_rays and stays there (the reference builds it with numpy, shuffles it on the host and slices +
H2D-copies a batch every iteration), and a training batch is one kernel launch (xr_make_batch_series).

PNG decoding uses PIL (the reference: imageio; cv2.INTER_AREA for half_res -- for the exact factor 2 that is
the mean of each 2x2 block, which is what `_half_res` computes).
"""
import json
import os

import numpy as np
import torch

from . import ops, synthetic


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def pose_spherical(theta, phi, radius):
    """load_blender.py:22-30 (float32 chain like the reference's torch.Tensor maths)"""
    t = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float32)
    p = np.float32(phi / 180. * np.pi)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]], np.float32)
    th = np.float32(theta / 180. * np.pi)
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    return flip @ (rt @ (rp @ t))


def _half_res(imgs):
    n, H, W, c = imgs.shape
    return imgs[:, :H // 2 * 2, :W // 2 * 2].reshape(n, H // 2, 2, W // 2, 2, c).mean((2, 4))


def load_blender_data(basedir, half_res=False, testskip=1):
    """-> imgs [N,H,W,4] f32 in [0,1], poses [N,4,4] f32, render_poses [40,4,4] f32, [H, W, focal], i_split
    (same contract and split order train/val/test as load_blender.py:32-89)."""
    splits = ['train', 'val', 'test']
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, 'transforms_{}.json'.format(s)), 'r') as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, counts = [], [], [0]
    for s in splits:
        meta = metas[s]
        skip = 1 if (s == 'train' or testskip == 0) else testskip
        imgs, poses = [], []
        for frame in meta['frames'][::skip]:
            imgs.append(_imread(os.path.join(basedir, frame['file_path'] + '.png')))
            poses.append(np.array(frame['transform_matrix']))
        imgs = (np.array(imgs) / 255.).astype(np.float32)          # keep all 4 channels (RGBA)
        poses = np.array(poses).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        all_imgs.append(imgs)
        all_poses.append(poses)
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    camera_angle_x = float(meta['camera_angle_x'])
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    render_poses = np.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        imgs = _half_res(imgs)
    return imgs, poses, render_poses, [H, W, focal], i_split


class DeviceRayTable:
    """All training rays of a set of posed RGBA images, resident in HBM: [N*H*W, 11] = (o3, d3, rgba4, img_id),
    shuffled once (hashnerf_dataset.py:43-45).  Serves the sampler hand-off (`get_alldata`, `get_info`:
    PassDatasetHook), the batch-size feedback (`set_batchsize`: ModifyBatchsizeHook) and batches
    (`next_batch`: HashBatchSample + RandomBGColor in one launch)."""

    def __init__(self, device, poses_ngp, images, H, W, focal, seed=1, shuffle=True, aabb_scale=1):
        self.device, self.H, self.W, self.focal = device, int(H), int(W), float(focal)
        # the reference's HashNerfDataset hard-codes aabb_scale = 1 (hashnerf_dataset.py:57-60); its sampler and kernels
        # take any power of two up to 128 (ngp_grid_sampler.py:83-85: max_cascade = log2(aabb_scale)) -- BASELINE config #4
        # (unbounded forward-facing scene) uses 16
        self.aabb_scale = int(aabb_scale)
        self.poses = np.ascontiguousarray(poses_ngp, dtype=np.float32)          # [n, 4, 3] NGP space
        self.n_img = self.poses.shape[0]
        rows = []
        for k in range(self.n_img):
            o, d = ops.gen_rays(self.poses[k], self.H, self.W, self.focal, self.focal, 0.5 * self.W, 0.5 * self.H, device=device)
            rgba = images(k, o, d) if callable(images) else torch.as_tensor(
                np.ascontiguousarray(images[k], dtype=np.float32)).to(device).reshape(-1, 4)
            ids = torch.full((o.shape[0], 1), float(k), dtype=torch.float32, device=device)
            rows.append(torch.cat([o, d, rgba, ids], 1))
        self.rays_rgb = torch.cat(rows, 0)
        if shuffle:
            g = torch.Generator(device='cpu').manual_seed(seed)
            perm = torch.randperm(self.rays_rgb.shape[0], generator=g).to(device)
            self.rays_rgb = self.rays_rgb[perm].contiguous()
        self.cur_i = 0
        self.N_rand = 4096
        self.batches_drawn = 0

    def get_alldata(self):                  # hashnerf_dataset.py:55-73
        aabb_scale = self.aabb_scale
        return {'aabb_scale': aabb_scale, 'aabb_range': (0.5 - aabb_scale / 2, 0.5 + aabb_scale / 2),
                'poses': self.poses, 'focal': np.ones((self.n_img, 2), dtype=float) * self.focal,
                'metadata': synthetic.metadata_rows(self.n_img, self.focal)}

    def get_info(self):                     # hashnerf_dataset.py:75-86
        K = np.array([[self.focal, 0, 0.5 * self.W], [0, self.focal, 0.5 * self.H], [0, 0, 1]])
        return {'H': self.H, 'W': self.W, 'focal': self.focal, 'K': K, 'hwf': [self.H, self.W, self.focal]}

    def set_batchsize(self, bs):            # ModifyBatchsizeHook (core/hooks/hash_hook.py:33-42)
        self.N_rand = int(bs)

    def next_batch(self, out=None):
        """HashBatchSample + RandomBGColor (pipelines/create.py:153-191, augment.py:290-317), on device, one launch.
        `out`: caller-owned buffers (ops.make_batch_buffers) the batch tensors become views of."""
        n = min(self.N_rand, self.rays_rgb.shape[0])          # a scene smaller than the batch: every ray, every step
        if self.cur_i + n > self.rays_rgb.shape[0]:
            self.cur_i = 0
        batch = ops.make_batch(self.rays_rgb[self.cur_i:self.cur_i + n], n, self.batches_drawn, out=out)
        self.cur_i += n
        self.batches_drawn += 1
        return batch


class HashNerfDataset(DeviceRayTable):
    """The reference's HashNerfDataset for a Blender-format scene directory, device-resident.

    cfg keys (configs/instant_ngp/nerf_blender_local01.py:129-160): datadir, half_res, testskip, white_bkgd,
    load_alpha, N_rand_per_sampler, mode ('train' | 'val' | 'test'), val_n.  Image order = (val, train) like
    hashnerf_dataset.py:33-35; poses -> NGP space with correct_pose [1,-1,-1], scale 0.33, offset 0.5 (:36-40)."""

    def __init__(self, cfg, pipeline=None, device=None, seed=1):
        cfg = dict(cfg)
        self.cfg, self.mode = cfg, cfg.get('mode', 'train')
        self.val_n = int(cfg.get('val_n', 1))
        device = device or torch.device('cuda:0')
        images, poses, render_poses, hwf, i_split = load_blender_data(cfg['datadir'], bool(cfg.get('half_res', False)),
                                                                      int(cfg.get('testskip', 1)))
        self.near, self.far = 2., 6.
        if cfg.get('white_bkgd', False):                     # load.py:61-63
            images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:])
        elif not cfg.get('load_alpha', True):
            images = images[..., :3]
        if images.shape[3] == 3:                             # check_img (hashnerf_dataset.py:19-23)
            images = np.concatenate([images, np.ones(list(images.shape[:3]) + [1], images.dtype)], 3)
        i_train, i_val, i_test = i_split
        self.i_train, self.i_val, self.i_test = i_train, i_val, i_test
        i_index = np.concatenate((i_val, i_train))
        images, poses = images[i_index], poses[i_index]
        self.images = images.astype(np.float32)
        self.render_poses = synthetic.poses_nerf2ngp(render_poses)
        poses_ngp = synthetic.poses_nerf2ngp(poses)
        H, W, focal = hwf
        if self.mode == 'test':                              # :47-54: black RGBA targets at the render poses
            blank = np.zeros((self.render_poses.shape[0], H, W, 4), np.float32)
            super().__init__(device, self.render_poses, blank, H, W, focal, seed=seed, shuffle=False)
            self.n_render = self.render_poses.shape[0]
        else:
            super().__init__(device, poses_ngp, self.images, H, W, focal, seed=seed, shuffle=(self.mode == 'train'))
        self.N_rand = int(cfg.get('N_rand_per_sampler', 4096))

    def get_info(self):
        info = super().get_info()
        info.update({'near': self.near, 'far': self.far})
        return info

    def fetch_val_data(self):               # _fetch_val_data (:99-103): ngp validates on the first val_n images
        return {'poses': self.poses[:self.val_n], 'images': self.images[:self.val_n]}

    def fetch_test_data(self, idx):         # _fetch_test_data (:105-117), rows of the unshuffled table
        n_pixel = self.H * self.W
        rows = self.rays_rgb[idx * n_pixel:(idx + 1) * n_pixel]
        return {'pose': self.render_poses[idx], 'rays_o': rows[:, :3], 'rays_d': rows[:, 3:6],
                'img_ids': torch.full((n_pixel, 1), float(idx), dtype=torch.float32, device=rows.device),
                'src_shape': np.array([self.H, self.W, 3]), 'idx': idx}

    def __len__(self):                      # :131-139
        if self.mode == 'train':
            return self.rays_rgb.shape[0] // int(self.cfg.get('N_rand_per_sampler', 4096)) * 4
        return 1 if self.mode == 'val' else self.n_render


# ------------------------------------------------------------------------------------------ LLFF (forward-facing) scenes
def _unit(v):
    return v / np.linalg.norm(v)


def _look_at(z, up, pos):
    """camera frame with optical axis z (viewmatrix, load_llff.py:131-137): columns [x, y, z, pos]"""
    z = _unit(z)
    x = _unit(np.cross(up, z))
    return np.stack([x, _unit(np.cross(z, x)), z, pos], 1)


def _average_pose(poses):
    """poses_avg (load_llff.py:145-155): mean position, summed z / y axes, hwf column of the first pose"""
    frame = _look_at(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], 1)


def _recenter(poses):
    """recenter_poses (load_llff.py:174-187): express every pose in the average pose's frame"""
    out = poses.copy()
    bottom = np.array([[0, 0, 0, 1.]])
    avg = np.concatenate([_average_pose(poses)[:3, :4], bottom], 0)
    full = np.concatenate([poses[:, :3, :4], np.tile(bottom[None], (poses.shape[0], 1, 1))], 1)
    out[:, :3, :4] = (np.linalg.inv(avg) @ full)[:, :3, :4]
    return out


def _spiral_path(c2w, up, rads, focal, zdelta, zrate, rots, n):
    """render_path_spiral (load_llff.py:158-171)"""
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    out = []
    for theta in np.linspace(0., 2. * np.pi * rots, n + 1)[:-1]:
        c = c2w[:3, :4] @ (np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = _unit(c - c2w[:3, :4] @ np.array([0, 0, -focal, 1.]))
        out.append(np.concatenate([_look_at(z, up, c), hwf], 1))
    return out


def _spherify(poses, bds):
    """spherify_poses (load_llff.py:193-265): re-centre on the point closest to all optical axes, unit mean radius,
    circular render path of 120 views"""
    def to44(p):
        return np.concatenate([p, np.tile(np.array([[[0, 0, 0, 1.]]]), (p.shape[0], 1, 1))], 1)
    d, o = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - d * np.transpose(d, [0, 2, 1])
    b = -A @ o
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    v0 = _unit((poses[:, :3, 3] - center).mean(0))
    v1 = _unit(np.cross([.1, .2, .3], v0))
    v2 = _unit(np.cross(v0, v1))
    c2w = np.stack([v1, v2, v0, center], 1)
    reset = np.linalg.inv(to44(c2w[None])) @ to44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    new = []
    for th in np.linspace(0., 2. * np.pi, 120):
        cam = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        z = _unit(cam)
        x = _unit(np.cross(z, np.array([0, 0, -1.])))
        new.append(np.stack([x, _unit(np.cross(z, x)), z, cam], 1))
    new = np.stack(new, 0)
    hwf = poses[0, :3, -1:]
    new = np.concatenate([new, np.broadcast_to(hwf, new[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, new, bds


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """-> images [N,H,W,3] f32, poses [N,3,5] f32 (hwf in the last column), bds [N,2] f32, render_poses f32, i_test
    Same contract as load_llff_data / _load_data (load_llff.py:67-128,268-349) for scenes whose down-scaled image folder
    (`images_<factor>`) already exists -- the reference shells out to ImageMagick's `mogrify` to create it otherwise; that
    is left to the user (no subprocess here)."""
    arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])          # [3,5,N]
    bds = arr[:, -2:].transpose([1, 0])
    sfx = '' if factor is None else '_{}'.format(factor)
    imgdir = os.path.join(basedir, 'images' + sfx)
    if not os.path.isdir(imgdir):
        raise FileNotFoundError('%s does not exist: create the down-scaled images first (the reference runs `mogrify '
                                '-resize %s%%` over a copy of images/)' % (imgdir, 100. / (factor or 1)))
    files = [os.path.join(imgdir, f) for f in sorted(os.listdir(imgdir)) if f.endswith(('JPG', 'jpg', 'png'))]
    if poses.shape[-1] != len(files):
        raise ValueError('Mismatch between imgs {} and poses {}'.format(len(files), poses.shape[-1]))
    imgs = [_imread(f)[..., :3] / 255. for f in files]
    sh = imgs[0].shape
    poses[:2, 4, :] = np.array(sh[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / (factor or 1)
    imgs = np.stack(imgs, -1)
    # [-y, x, z] column order -> [x, y, z] (load_llff.py:280-285), variable axis first
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = _recenter(poses)
    if spherify:
        poses, render_poses, bds = _spherify(poses, bds)
    else:
        c2w = _average_pose(poses)
        up = _unit(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)
        zdelta = close_depth * .2
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            c2w[:3, 3] = c2w[:3, 3] + (-close_depth * .1) * c2w[:3, 2]
            rads[2] = 0.
            n_rots, n_views = 1, n_views // 2
        render_poses = _spiral_path(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, n=n_views)
    render_poses = np.array(render_poses).astype(np.float32)
    c2w = _average_pose(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test
