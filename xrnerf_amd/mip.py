"""Registry entries of the Mip-NeRF path (BASELINE config #3, configs/mipnerf/mipnerf_multiscale.py):
`MipNerfNetwork`, `MipNerfEmbedder`, `MipNerfRender` (+ `sample_along_rays` / `resample_along_rays` / `get_z_vals`),
constructor signatures, `data` dict keys and numerics of
  /root/reference/xrnerf/models/networks/mipnerf.py:14-117, networks/utils/mip.py:134-176,
  embedders/mipnerf_embedder.py:12-99, renders/mipnerf_render.py:11-33 (+ renders/nerf_render.py:45-98),
  datasets/pipelines/create.py:486-531 (GetZvals).

Where the reference strings ~60 elementwise / reduction launches per level between the data loader and the
8x256 MLP, every stage here is ONE HIP launch through the C-ABI (xrnerf_amd/csrc/xr_mip.hip): GetZvals, cast_rays
+ integrated positional encoding + view encoding + concat, the renderer (forward and backward), the resampler.  The
MLP itself is `NerfMLP` (xrnerf_amd/vanilla.py: nn.Linear = rocBLAS GEMMs).  No CPU path: host tensors raise.
"""
import torch
from torch import nn

from . import builder, ops
from .builder import EMBEDDERS, NETWORKS, RENDERS
from .networks import get_dist_info, mse2psnr, unfold_batching
from .vanilla import NerfNetwork, merge_ret


class MipSamples:
    """what `sample_along_rays` leaves in data['samples']: the frustum description instead of materialised
    (means, covs) -- the encoder builds the gaussians in LDS.  Unpacks like the reference's tuple when someone
    asks for the gaussians themselves (`means, covs = data['samples']`, same formulas in torch on the device)."""

    def __init__(self, z_vals, rays_o, rays_d, radii, ray_shape):
        assert ray_shape in ('cone', 'cylinder')
        self.z_vals, self.rays_o, self.rays_d, self.radii, self.ray_shape = z_vals, rays_o, rays_d, radii, ray_shape

    def __iter__(self):
        z, o, d, r = self.z_vals, self.rays_o, self.rays_d, self.radii.reshape(-1, 1)
        t0, t1 = z[..., :-1], z[..., 1:]
        if self.ray_shape == 'cone':            # mip.py:92-110 (stable form)
            mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
            t_mean = mu + (2 * mu * hw**2) / (3 * mu**2 + hw**2)
            t_var = (hw**2) / 3 - (4 / 15) * ((hw**4 * (12 * mu**2 - hw**2)) / (3 * mu**2 + hw**2)**2)
            r_var = r**2 * ((mu**2) / 4 + (5 / 12) * hw**2 - 4 / 15 * (hw**4) / (3 * mu**2 + hw**2))
        else:                                   # mip.py:113-118
            t_mean, r_var, t_var = (t0 + t1) / 2, (r**2 / 4).expand_as(t0), (t1 - t0)**2 / 12
        d2 = d**2
        mag = torch.clamp_min(torch.sum(d2, -1, keepdim=True), 1e-10)
        means = d[..., None, :] * t_mean[..., None] + o[..., None, :]
        covs = t_var[..., None] * d2[..., None, :] + r_var[..., None] * (1 - d2 / mag)[..., None, :]
        return iter((means, covs))


def get_z_vals(data, N_samples, lindisp=False, randomized=False, z_rand=None):
    """GetZvals (create.py:486-531) on the device: data['near'], data['far'] [R,1] -> data['z_vals'] [R,N_samples]"""
    near = data['near']
    if randomized and z_rand is None:
        z_rand = torch.rand((near.reshape(-1).shape[0], N_samples), device=near.device)
    data['z_vals'] = ops.mip_zvals(near, data['far'], N_samples, lindisp, z_rand if randomized else None)
    return data


def sample_along_rays(data, ray_shape):
    """mip.py:134-148"""
    data['samples'] = MipSamples(data['z_vals'], data['rays_o'], data['rays_d'], data['radii'], ray_shape)
    return data


def resample_along_rays(data, randomized, ray_shape, resample_padding, rand=None):
    """mip.py:151-176: new z_vals from the previous level's weights (no gradient, as the reference detaches)"""
    z = data['z_vals']
    if randomized and rand is None:
        rand = torch.rand(z.shape, device=z.device)
    new_z = ops.mip_resample(z, data['weights'], resample_padding, rand if randomized else None)
    data['z_vals'] = new_z
    data['samples'] = MipSamples(new_z, data['rays_o'], data['rays_d'], data['radii'], ray_shape)
    return data


@EMBEDDERS.register_module()
class MipNerfEmbedder(nn.Module):
    """integrated positional encoding of the frustum gaussians + positional encoding of the view directions
    (mipnerf_embedder.py:12-99; diag covariances only, like every config of the reference)"""

    def __init__(self, min_deg_point, max_deg_point, min_deg_view, max_deg_view, input_ch=3, use_viewdirs=False,
                 diag=True, append_identity=True):
        super().__init__()
        if not diag:
            raise NotImplementedError('full-covariance IPE: no reference config uses it')
        if input_ch != 3:
            raise NotImplementedError('input_ch must be 3')
        self.min_deg, self.max_deg = int(min_deg_point), int(max_deg_point)
        self.min_deg_view, self.max_deg_view = int(min_deg_view), int(max_deg_view)
        self.use_viewdirs, self.diag, self.append_identity, self.input_ch = use_viewdirs, diag, append_identity, input_ch

    def get_embed_ch(self):
        d = self.input_ch
        ch_ipe = 2 * d * (self.max_deg - self.min_deg)
        ch_pe = 2 * d * (self.max_deg_view - self.min_deg_view) + (d if self.append_identity else 0)
        return ch_ipe, ch_pe

    def forward(self, data):
        s = data['samples']
        args = (self.min_deg, self.max_deg, self.min_deg_view, self.max_deg_view, self.append_identity)
        if isinstance(s, MipSamples):
            R, S = s.z_vals.shape[0], s.z_vals.shape[1] - 1
            data['embedded'] = ops.mip_encode(s.rays_o, s.rays_d, data['viewdirs'], s.radii, s.z_vals, *args,
                                              ray_shape=s.ray_shape)
        else:                                   # the reference's (means, covs) tuple
            means, covs = s
            R, S = means.shape[:2]
            data['embedded'] = ops.mip_encode_gaussians(means, covs, data['viewdirs'], *args)
        data['unflatten_shape'] = torch.Size((R, S))
        return data


class _MipRenderFn(torch.autograd.Function):
    """renderer forward / backward as one launch each; gradients flow from the colours only (the reference's losses,
    networks/mipnerf.py:52-60; `weights` feed the detached resampler)"""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, density_bias, rgb_padding, white_bkgd, act):
        rgb, dist, acc, w = ops.mip_render_forward(raw, z_vals, rays_d, density_bias, rgb_padding, white_bkgd, act)
        ctx.save_for_backward(raw, z_vals, rays_d)
        ctx.cfg = (density_bias, rgb_padding, white_bkgd, act)
        ctx.mark_non_differentiable(dist, acc, w)
        return rgb, dist, acc, w

    @staticmethod
    def backward(ctx, g_rgb, g_dist, g_acc, g_w):
        raw, z_vals, rays_d = ctx.saved_tensors
        return (ops.mip_render_backward(raw, z_vals, rays_d, g_rgb.contiguous(), *ctx.cfg), None, None, None, None,
                None, None)


@RENDERS.register_module()
class MipNerfRender(nn.Module):
    """NerfRender.forward with MipNerfRender's weights / distance map (nerf_render.py:45-98, mipnerf_render.py:12-33)"""

    def __init__(self, white_bkgd=False, raw_noise_std=0, rgb_padding=0, density_bias=0, density_activation='relu',
                 **kwarg):
        super().__init__()
        if density_activation not in ('softplus', 'relu'):
            raise NotImplementedError
        self.white_bkgd, self.raw_noise_std = white_bkgd, raw_noise_std
        self.rgb_padding, self.density_bias, self.density_activation = rgb_padding, density_bias, density_activation

    def forward(self, data, is_test=False):
        raw, z_vals = data['raw'], data['z_vals']
        if raw.shape[1] != z_vals.shape[1] - 1:
            raise ValueError('MipNerfRender expects interval edges: z_vals [R, S+1] for raw [R, S, 4]')
        noise_std = 0 if is_test else self.raw_noise_std
        if noise_std > 0.:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * noise_std
            raw = torch.cat([raw[..., :3], (raw[..., 3] + noise)[..., None]], -1)
        rgb, dist, acc, w = _MipRenderFn.apply(raw, z_vals, data['rays_d'], float(self.density_bias),
                                               float(self.rgb_padding), bool(self.white_bkgd), self.density_activation)
        data['weights'] = w
        return data, {'rgb': rgb, 'disp': dist, 'acc': acc}


@NETWORKS.register_module()
class MipNerfNetwork(NerfNetwork):
    """networks/mipnerf.py:14-117: num_levels passes of sample/resample -> mlp -> render with ONE shared MLP"""

    def __init__(self, cfg, mlp=None, render=None):
        super().__init__(cfg, mlp=mlp, render=render)
        cfg = builder.ConfigDict.wrap(dict(cfg))
        self.num_levels = cfg.num_levels
        self.resample_padding = cfg.resample_padding
        self.ray_shape = cfg.ray_shape
        self.use_multiscale = cfg.use_multiscale
        self.coarse_loss_mult = cfg.coarse_loss_mult

    def forward(self, data, is_test=False):
        randomized = not is_test
        ret = {}
        for i_level in range(self.num_levels):
            if i_level == 0:
                data = sample_along_rays(data, self.ray_shape)
            else:
                data = resample_along_rays(data, randomized, self.ray_shape, self.resample_padding)
            data, temp_ret = self.render(self.mlp(data), is_test)
            ret = temp_ret if not ret else merge_ret(ret, temp_ret)
        return ret

    def train_step(self, data, optimizer, **kwargs):
        for k in data:
            data[k] = unfold_batching(data[k])
        ret = self.forward(data, is_test=False)
        if 'lossmult' in data:
            mask = torch.broadcast_to(data['lossmult'], ret['rgb'].shape)
        else:
            mask = torch.ones_like(ret['rgb'])
        msum = mask.sum()
        loss_fine = (mask * (ret['rgb'] - data['target_s'])**2).sum() / msum
        loss_coarse = (mask * (ret['coarse_rgb'] - data['target_s'])**2).sum() / msum
        loss = loss_fine + self.coarse_loss_mult * loss_coarse
        psnr = mse2psnr(loss_fine)
        # the reference calls .item() four times here (four stream drains per iteration); the scalars are logged
        # lazily instead: float() them when a logger actually needs the numbers
        log_vars = {'loss': loss.detach(), 'loss_fine': loss_fine.detach(), 'loss_coarse': loss_coarse.detach(),
                    'psnr': psnr.detach()}
        return {'loss': loss, 'log_vars': log_vars, 'num_samples': ret['rgb'].shape[0]}

    def evaluate_once(self, data, **kwargs):
        """networks/mipnerf.py:100-117: one [1,H,W,*] image dict -> rgb / gt / distance maps"""
        H, W = data['image'].shape[1:3]
        idx = int(data.pop('idx').item())
        for key in data.keys():
            data[key] = data[key].squeeze(0).reshape(H * W, -1).to(torch.float32)
        with torch.no_grad():
            ret = self.batchify_forward(data, is_test=True)
        rgb = ret['rgb'].reshape((H, W, -1)).cpu().numpy()
        disp = ret['disp'].reshape((H, W, -1)).cpu().numpy()
        image = data['image'].reshape((H, W, -1)).cpu().numpy()
        return rgb, image, disp, idx

    def val_step(self, data, optimizer=None, **kwargs):
        if not self.use_multiscale:                  # networks/mipnerf.py:76-78 (configs/mipnerf/mipnerf_blender.py)
            return super().val_step(data, **kwargs)
        rank, _ = get_dist_info()
        if rank != 0:
            return {}
        rgb, image, disp, idx = self.evaluate_once(data, **kwargs)
        if self.phase == 'test':
            return {'rgb': rgb, 'gt_img': image, 'disp': disp, 'idx': idx}
        return {'rgbs': [rgb], 'gt_imgs': [image], 'disps': [disp]}


# ------------------------------------------------------------------ synthetic multiscale rays (bench / smoke / tests)
def synthetic_multiscale_rays(n_rays, device, seed=0, H=800, W=800, focal=1111.111, n_scales=4):
    """Lego-shaped multiscale rays like load_rays_multiscale (datasets/load_data/get_rays.py:100-152) produces:
    un-normalised directions K^-1 [x+.5, y+.5, 1] rotated to the world, radii = pixel footprint * 2/sqrt(12),
    lossmult = 4^scale, near/far 2/6, random cameras on the radius-4 hemisphere, random pixels."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    scale = torch.randint(0, n_scales, (n_rays,), generator=g)
    f = focal / (2.0 ** scale)
    w, h = (W / (2.0 ** scale)).floor(), (H / (2.0 ** scale)).floor()
    px = (torch.rand(n_rays, generator=g) * w).floor() + 0.5
    py = (torch.rand(n_rays, generator=g) * h).floor() + 0.5
    cam_dir = torch.stack([(px - 0.5 * w) / f, -(py - 0.5 * h) / f, -torch.ones(n_rays)], -1)
    theta = torch.rand(n_rays, generator=g) * 6.2831853
    phi = torch.rand(n_rays, generator=g) * 1.2 + 0.1
    cam = 4.0 * torch.stack([phi.cos() * theta.cos(), phi.cos() * theta.sin(), phi.sin()], -1)
    fwd = -cam / cam.norm(dim=-1, keepdim=True)
    right = torch.linalg.cross(fwd, torch.tensor([0., 0., 1.]).expand_as(fwd))
    right = right / right.norm(dim=-1, keepdim=True)
    up = torch.linalg.cross(right, fwd)
    rays_d = cam_dir[:, :1] * right + cam_dir[:, 1:2] * up - cam_dir[:, 2:3] * fwd
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    radii = (1.0 / f * 2 / 12 ** 0.5)[:, None]            # |d(x+1) - d(x)| = 1/f for these un-normalised directions
    data = dict(rays_o=cam, rays_d=rays_d, viewdirs=viewdirs, radii=radii, lossmult=(4.0 ** scale)[:, None],
                near=torch.full((n_rays, 1), 2.0), far=torch.full((n_rays, 1), 6.0),
                target_s=torch.rand(n_rays, 3, generator=g))
    return {k: v.to(torch.float32).contiguous().to(device) for k, v in data.items()}
