"""Registry entries of the vanilla-NeRF path (BASELINE config #1, configs/nerf/nerf_blender_base01.py):
`NerfNetwork`, `NerfMLP`, `BaseEmbedder`, `NerfRender`, plus stratified / hierarchical sampling.

This is the reference's only renderer that runs on a CPU as shipped and is
pure PyTorch there as well; it is plumbing for the registry contract and the CPU-runnable parity
case, not a hot path: plain tensor code, no kernels.  Constructor signatures, `data` dict keys,
sub-module / parameter names (`pts_linears.N`, `views_linears.0`, `feature_linear`, `alpha_linear`,
`rgb_linear`) and numerics follow
  /root/reference/xrnerf/models/embedders/base.py:8-77, mlps/nerf_mlp.py:11-94,
  renders/nerf_render.py:10-98, networks/utils/hierarchical_sample.py:6-53, networks/nerf.py:16-92
  and datasets/pipelines/create.py:486-531,577-601, augment.py:261-283.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import builder
from .builder import EMBEDDERS, MLPS, NETWORKS, RENDERS
from .networks import BaseNerfNetwork, img2mse, mse2psnr, unfold_batching


# ------------------------------------------------------------------ positional encoding
@EMBEDDERS.register_module()
class BaseEmbedder(nn.Module):
    """gamma(p) = (p, sin(2^k p), cos(2^k p))_{k<multires}; directions with multires_dirs"""

    def __init__(self, i_embed=0, multires=10, multires_dirs=4, input_ch=3, **kwargs):
        super().__init__()
        self.input_ch = input_ch
        if i_embed == -1:
            self.freqs, self.freqs_dirs = None, None
            self.embed_ch = self.embed_ch_dirs = input_ch
        else:
            self.freqs = [float(2.0 ** k) for k in range(multires)]          # 2**linspace(0, L-1, L)
            self.freqs_dirs = [float(2.0 ** k) for k in range(multires_dirs)]
            self.embed_ch = input_ch * (1 + 2 * multires)
            self.embed_ch_dirs = input_ch * (1 + 2 * multires_dirs)

    def get_embed_ch(self):
        return self.embed_ch, self.embed_ch_dirs

    @staticmethod
    def run_embed(x, freqs):
        if freqs is None:
            return x
        parts = [x]
        for f in freqs:
            parts.append(torch.sin(x * f))
            parts.append(torch.cos(x * f))
        return torch.cat(parts, -1)

    def forward(self, data):
        pts, viewdirs = data['pts'], data['viewdirs']
        data['unflatten_shape'] = pts.shape[:-1]
        if pts.dim() > viewdirs.dim():
            viewdirs = viewdirs[:, None].expand(pts.shape)
        e_pts = self.run_embed(pts.reshape(-1, pts.shape[-1]), self.freqs)
        e_dir = self.run_embed(viewdirs.reshape(-1, viewdirs.shape[-1]), self.freqs_dirs)
        data['embedded'] = torch.cat([e_pts, e_dir], -1)
        return data


# ------------------------------------------------------------------ 8 x 256 MLP with a skip and a view branch
@MLPS.register_module()
class NerfMLP(nn.Module):
    def __init__(self, skips=[4], netdepth=8, netwidth=256, output_ch=4, use_viewdirs=True, netchunk=1024 * 32,
                 embedder=None, **kwarg):
        super().__init__()
        self.skips = list(skips)
        self.chunk = netchunk
        self.use_viewdirs = use_viewdirs
        self.embedder = builder.build_embedder(embedder)
        D, W = netdepth, netwidth
        self.input_ch, self.input_ch_dirs = self.embedder.get_embed_ch()
        layers = [nn.Linear(self.input_ch, W)]
        for i in range(D - 1):
            layers.append(nn.Linear(W + self.input_ch if i in self.skips else W, W))
        self.pts_linears = nn.ModuleList(layers)
        if use_viewdirs:
            self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_dirs + W, W // 2)])
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    def _device_graph_ok(self, x):
        W = self.pts_linears[0].weight.shape[0]
        from . import ops
        return (ops._on_device(x) and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0 and self.use_viewdirs and len(self.views_linears) == 1
                and self.input_ch % 4 == 0 and W % 4 == 0 and (W // 2) % 4 == 0 and (len(self.pts_linears) - 1) not in self.skips
                and all(l.weight.shape[0] == W for l in self.pts_linears))

    def run_mlp(self, x):
        if self._device_graph_ok(x):
            # device: the whole MLP as ONE autograd node over the linear kernels, every activation in a buffer of its own layout
            # (_NerfMlpFn below): no concatenations, splits, paddings or per-layer autograd glue between the products
            params = []
            for layer in self.pts_linears:
                params += [layer.weight, layer.bias]
            v = self.views_linears[0]
            params += [v.weight, v.bias, self.feature_linear.weight, self.feature_linear.bias, self.alpha_linear.weight, self.alpha_linear.bias,
                       self.rgb_linear.weight, self.rgb_linear.bias]
            return _NerfMlpFn.apply(x, tuple(self.skips), self.input_ch, self.input_ch_dirs, *params)
        x_pts, x_dir = torch.split(x, [self.input_ch, self.input_ch_dirs], dim=-1)
        if x.is_cuda:
            # device: linear + bias + relu (and their gradients) on the fp32-MFMA kernels (xrnerf_amd/linear.py); same
            # maths, summation order aside.  alpha and feature heads share one product (rows of one weight matrix).
            from .linear import linear_act_padded as lin
            x_pts = x_pts.contiguous()
        else:
            def lin(t, w, b, relu=False):
                y = F.linear(t, w, b)
                return F.relu(y) if relu else y
        h = x_pts
        for i, layer in enumerate(self.pts_linears):
            h = lin(h, layer.weight, layer.bias, True)
            if i in self.skips:
                h = torch.cat([x_pts, h], -1)
        if not self.use_viewdirs:
            return lin(h, self.output_linear.weight, self.output_linear.bias)
        if x.is_cuda:
            W = self.feature_linear.weight.shape[0]
            both = lin(h, torch.cat([self.feature_linear.weight, self.alpha_linear.weight], 0),
                       torch.cat([self.feature_linear.bias, self.alpha_linear.bias], 0))
            feature, alpha = both[:, :W], both[:, W:W + 1]
        else:
            alpha = self.alpha_linear(h)
            feature = self.feature_linear(h)
        h = torch.cat([feature, x_dir], -1)
        for layer in self.views_linears:
            h = lin(h, layer.weight, layer.bias, True)
        return torch.cat([lin(h, self.rgb_linear.weight, self.rgb_linear.bias), alpha], -1)

    def batchify_run_mlp(self, x):
        # netchunk only bounds the reference's activation memory; on the device the whole batch goes through in one piece
        # (same rows, same maths; 288 GB of HBM, and the GEMM tiles fill the chip better)
        if self.chunk is None or x.is_cuda:
            return self.run_mlp(x)
        return torch.cat([self.run_mlp(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)], 0)

    def forward(self, data):
        data = self.embedder(data)
        out = self.batchify_run_mlp(data['embedded'])
        data['raw'] = out.reshape(list(data['unflatten_shape']) + [out.shape[-1]])
        del data['unflatten_shape']
        return data


class _NerfMlpFn(torch.autograd.Function):
    """NerfMLP.run_mlp on the device (nerf_mlp.py:62-94) as one autograd node.  Same products, same order of the concatenated inputs; what is
    gone is the glue around them (19 % of the Mip-NeRF step's GPU time in round 5: cat / split / pad copies, masked copies, two reductions per
    layer, per-layer autograd nodes):
      * the skip connection's [x_pts | h] is a buffer the skip layer writes its output INTO (xr_linear_forward with an output row stride);
        the gradient of that output is the same column range of the next layer's input gradient (row stride on dy and the relu mask);
      * feature and alpha heads are one product whose output lands in the view layer's input buffer [feature | alpha 0 0 0 | dir | 0]; the
        view layer's weight has zero columns under alpha and the padding (exact: the products are w * 0), so nothing is concatenated;
      * the rgb head has a fourth, zero output row; the node's output [rgb | alpha] is that product with alpha copied into column 3;
      * every layer's weight and bias gradient come from one launch and one reduction over the M ranges (ops.linear_backward_weight_bias).
    x: the embedded batch [M, input_ch + input_ch_dirs], rows 16-byte aligned (ops.mip_encode pads its rows for this; otherwise one copy)."""

    @staticmethod
    def forward(ctx, x, skips, ic, idr, *params):
        from . import ops
        D = (len(params) - 8) // 2
        vw, vb, fw, fb, aw, ab, rw, rb = [p.detach() for p in params[2 * D:]]
        pts = [p.detach() for p in params[:2 * D]]
        xr, _ = ops._rows(x.detach())
        M, W, W2 = xr.shape[0], pts[0].shape[0], vw.shape[0]
        x_pts, x_dir = xr[:, :ic], xr[:, ic:ic + idr]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=xr.device)
        acts, h = [], x_pts
        for i in range(D):
            if i in skips:
                cat = new(M, ic + W)
                cat[:, :ic].copy_(x_pts)
                y = ops.linear_forward(h, pts[2 * i], pts[2 * i + 1], True, out=cat[:, ic:])
                acts.append((h, y))
                h = cat
            else:
                y = ops.linear_forward(h, pts[2 * i], pts[2 * i + 1], True)
                acts.append((h, y))
                h = y
        o_dir = W + 4
        KV = o_dir + (idr + 3) // 4 * 4
        V = new(M, KV)
        wb = torch.cat([fw, aw, fw.new_zeros((3, W))], 0)
        bb = torch.cat([fb, ab, fb.new_zeros((3,))], 0)
        ops.linear_forward(h, wb, bb, False, out=V[:, :o_dir])
        V[:, o_dir:o_dir + idr].copy_(x_dir)
        if KV > o_dir + idr:
            V[:, o_dir + idr:].zero_()
        wv2 = vw.new_zeros((W2, KV))
        wv2[:, :W] = vw[:, :W]
        wv2[:, o_dir:o_dir + idr] = vw[:, W:]
        hv = ops.linear_forward(V, wv2, vb, True)
        wr2 = torch.cat([rw, rw.new_zeros((1, W2))], 0)
        br2 = torch.cat([rb, rb.new_zeros((1,))], 0)
        raw = ops.linear_forward(hv, wr2, br2, False)
        raw[:, 3].copy_(V[:, W])
        ctx.cfg = (tuple(skips), ic, idr, D, W, W2, o_dir, KV)
        ctx.acts, ctx.h_last, ctx.V, ctx.hv = acts, h, V, hv
        ctx.w = (pts, wb, wv2, wr2)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        from . import ops
        skips, ic, idr, D, W, W2, o_dir, KV = ctx.cfg
        pts, wb, wv2, wr2 = ctx.w
        acts, V, hv = ctx.acts, ctx.V, ctx.hv
        d_raw = d_raw.contiguous()
        dwr2, dbr2 = ops.linear_backward_weight_bias(d_raw, None, hv)
        dhv = ops.linear_backward_input(d_raw, None, wr2)
        dwv2, dbv = ops.linear_backward_weight_bias(dhv, hv, V)
        dV = ops.linear_backward_input(dhv, hv, wv2)
        dV[:, W].copy_(d_raw[:, 3])                          # alpha's gradient joins the feature gradient: one product for both heads
        dyb = dV[:, :o_dir]
        dwb, dbb = ops.linear_backward_weight_bias(dyb, None, ctx.h_last)
        dh = ops.linear_backward_input(dyb, None, wb)
        grads = [None] * (2 * D)
        for i in range(D - 1, -1, -1):
            xin, y = acts[i]
            dy = dh[:, ic:] if i in skips else dh
            grads[2 * i], grads[2 * i + 1] = ops.linear_backward_weight_bias(dy, y, xin)
            if i > 0:
                dh = ops.linear_backward_input(dy, y, pts[2 * i])
        dvw = torch.cat([dwv2[:, :W], dwv2[:, o_dir:o_dir + idr]], 1)
        ctx.acts = ctx.h_last = ctx.V = ctx.hv = None
        return (None, None, None, None) + tuple(grads) + (dvw, dbv, dwb[:W], dbb[:W], dwb[W:W + 1], dbb[W:W + 1], dwr2[:3], dbr2[:3])


# ------------------------------------------------------------------ classic volume rendering
@RENDERS.register_module()
class NerfRender(nn.Module):
    def __init__(self, white_bkgd=False, raw_noise_std=0, rgb_padding=0, density_bias=0, density_activation='relu',
                 **kwarg):
        super().__init__()
        self.white_bkgd, self.raw_noise_std = white_bkgd, raw_noise_std
        self.rgb_padding, self.density_bias = rgb_padding, density_bias
        if density_activation == 'softplus':
            self.density_activation = F.softplus
        elif density_activation == 'relu':
            self.density_activation = F.relu
        else:
            raise NotImplementedError

    @staticmethod
    def get_weights(density_delta):
        """w_k = alpha_k * prod_{j<k} (1 - alpha_j + 1e-10)"""
        alpha = 1 - torch.exp(-density_delta)
        ones = torch.ones((alpha.shape[0], 1), device=alpha.device, dtype=alpha.dtype)
        trans = torch.cumprod(torch.cat([ones, 1. - alpha + 1e-10], -1), -1)[:, :-1]
        return alpha * trans

    @staticmethod
    def get_disp_map(weights, z_vals):
        depth = torch.sum(weights * z_vals, -1)
        return 1. / torch.max(1e-10 * torch.ones_like(depth), depth / torch.sum(weights, -1))

    def forward(self, data, is_test=False):
        raw, z_vals, rays_d = data['raw'], data['z_vals'], data['rays_d']
        noise_std = 0 if is_test else self.raw_noise_std
        if (raw.is_cuda and not (torch.is_grad_enabled() and raw.requires_grad) and noise_std == 0
                and self.density_activation is F.relu and self.rgb_padding == 0 and self.density_bias == 0
                and raw.dim() == 3 and raw.shape[-1] == 4 and tuple(z_vals.shape) == tuple(raw.shape[:2])):
            # inference on the device: one launch (xr_nerf_render_forward) instead of ~25 tensor ops
            from . import ops
            rgb_map, disp, acc, weights = ops.nerf_render_forward(raw, z_vals, rays_d, self.white_bkgd)
            data['weights'] = weights
            return data, {'rgb': rgb_map, 'disp': disp, 'acc': acc}
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        if dists.shape[1] != raw.shape[1]:          # z_vals are sample positions, not interval edges
            far = torch.full_like(dists[..., :1], 1e10)
            dists = torch.cat([dists, far], -1)
        dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
        rgb = torch.sigmoid(raw[..., :3]) * (1 + 2 * self.rgb_padding) - self.rgb_padding
        noise = torch.randn(raw[..., 3].shape, device=raw.device) * noise_std if noise_std > 0. else 0.
        weights = self.get_weights(self.density_activation(raw[..., 3] + noise + self.density_bias) * dists)
        rgb_map = torch.sum(weights[..., None] * rgb, -2)
        acc_map = torch.sum(weights, -1)
        ret = {'rgb': rgb_map + (1. - acc_map[..., None]) if self.white_bkgd else rgb_map,
               'disp': self.get_disp_map(weights, z_vals), 'acc': acc_map}
        data['weights'] = weights
        return data, ret


# ------------------------------------------------------------------ sampling along rays
def get_z_vals(rays_o, near, far, n_samples=64, lindisp=False, randomized=False):
    """GetZvals (datasets/pipelines/create.py:486-531)"""
    t = torch.linspace(0., 1., steps=n_samples, device=rays_o.device)
    z = near * (1. - t) + far * t if not lindisp else 1. / (1. / near * (1. - t) + 1. / far * t)
    if randomized:
        return perturb_z_vals(z.expand(list(rays_o.shape[:-1]) + [n_samples]))
    return z.expand(list(rays_o.shape[:-1]) + [n_samples])


def perturb_z_vals(z_vals, t_rand=None):
    """PerturbZvals (datasets/pipelines/augment.py:261-283): one uniform draw inside each sample's bin"""
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    if t_rand is None:
        t_rand = torch.rand(z_vals.shape, device=z_vals.device)
    return lower + (upper - lower) * t_rand


def get_pts(rays_o, rays_d, z_vals):
    """GetPts (create.py:577-601)"""
    return rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]


def sample_pdf(data, N_samples, is_perturb=False, is_test=False, u=None):
    """hierarchical re-sampling from the coarse weights (networks/utils/hierarchical_sample.py:6-53)"""
    z_vals, rays_o, rays_d = data['z_vals'], data['rays_o'], data['rays_d']
    weights = data['weights'][..., 1:-1] + 1e-5
    det = True if is_test else (not is_perturb)
    bins = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0., 1., steps=N_samples).expand(list(cdf.shape[:-1]) + [N_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [N_samples])
    u = u.to(cdf.device).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_lo, bin_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    z_samples = (bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)).detach()
    z_all, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    data['pts'], data['z_vals'] = get_pts(rays_o, rays_d, z_all), z_all
    return data


def merge_ret(ret, fine_ret):
    for k in ('rgb', 'disp', 'acc'):
        ret['coarse_' + k] = ret[k]
        ret[k] = fine_ret[k]
    return ret


# ------------------------------------------------------------------ coarse + fine network
@NETWORKS.register_module()
class NerfNetwork(BaseNerfNetwork):
    def __init__(self, cfg, mlp=None, mlp_fine=None, render=None):
        super().__init__()
        cfg = builder.ConfigDict.wrap(dict(cfg))
        self.phase = cfg.get('phase', 'train')
        if 'chunk' in cfg: self.chunk = cfg.chunk
        if 'bs_data' in cfg: self.bs_data = cfg.bs_data
        if 'is_perturb' in cfg: self.is_perturb = cfg.is_perturb
        if 'N_importance' in cfg: self.N_importance = cfg.N_importance
        if mlp is not None: self.mlp = builder.build_mlp(mlp)
        if mlp_fine is not None: self.mlp_fine = builder.build_mlp(mlp_fine)
        if render is not None: self.render = builder.build_render(render)

    def forward(self, data, is_test=False):
        data, ret = self.render(self.mlp(data), is_test)
        if self.N_importance > 0:
            data = sample_pdf(data, self.N_importance, self.is_perturb, is_test)
            _, fine_ret = self.render(self.mlp_fine(data), is_test)
            ret = merge_ret(ret, fine_ret)
        return ret

    def batchify_forward(self, data, is_test=False):
        N = data[self.bs_data].shape[0]
        all_ret = {}
        for i in range(0, N, self.chunk):
            chunk = {k: (v[i:i + self.chunk] if v.shape[0] == N else v) for k, v in data.items()}
            for k, v in self.forward(chunk, is_test).items():
                all_ret.setdefault(k, []).append(v)
        return {k: torch.cat(v, 0) for k, v in all_ret.items()}

    def train_step(self, data, optimizer, **kwargs):
        for k in data:
            data[k] = unfold_batching(data[k])
        ret = self.forward(data, is_test=False)
        loss = img2mse(ret['rgb'], data['target_s'])
        psnr = mse2psnr(loss)
        if 'coarse_rgb' in ret:
            loss = loss + img2mse(ret['coarse_rgb'], data['target_s'])
        return {'loss': loss, 'log_vars': {'loss': loss.item(), 'psnr': psnr.item()}, 'num_samples': ret['rgb'].shape[0]}

    def val_step(self, data, optimizer=None, **kwargs):
        """networks/nerf.py:93-142: render the validation poses (timed per frame, pipeline included) and the spiral
        poses through `val_pipeline` (set by the validation hook) in `chunk`-sized pieces; rank 0 only"""
        import time
        from .networks import get_dist_info, recover_shape, unfold_batching
        if self.phase == 'test':
            return self.test_step(data, **kwargs)
        rank, _ = get_dist_info()
        if rank != 0:
            return {}
        for k in data:
            data[k] = unfold_batching(data[k])
        poses, images, spiral_poses = data['poses'], data['images'], data['spiral_poses']
        rgbs, disps, gt_imgs, elapsed = [], [], [], []
        with torch.no_grad():
            for i in range(poses.shape[0]):
                start = time.time()
                frame = self.val_pipeline({'pose': poses[i]})
                ret = self.batchify_forward(frame, is_test=True)
                rgb = recover_shape(ret['rgb'], frame['src_shape']).cpu().numpy()     # the read-back ends the frame
                elapsed.append(time.time() - start)
                rgbs.append(rgb)
                disps.append(recover_shape(ret['disp'], frame['src_shape']).cpu().numpy())
                gt_imgs.append(images[i].cpu().numpy())
            spiral_rgbs, spiral_disps = [], []
            for i in range(spiral_poses.shape[0]):
                frame = self.val_pipeline({'pose': spiral_poses[i]})
                ret = self.batchify_forward(frame, is_test=True)
                spiral_rgbs.append(recover_shape(ret['rgb'], frame['src_shape']).cpu().numpy())
                spiral_disps.append(recover_shape(ret['disp'], frame['src_shape']).cpu().numpy())
        return {'spiral_rgbs': spiral_rgbs, 'spiral_disps': spiral_disps, 'rgbs': rgbs, 'disps': disps, 'gt_imgs': gt_imgs,
                'elapsed_time': elapsed}

    def test_step(self, data, **kwargs):
        """networks/nerf.py:144-168 (the runner only knows train_step / val_step: val_step + phase == 'test')"""
        from .networks import get_dist_info, recover_shape, unfold_batching
        rank, _ = get_dist_info()
        if rank != 0:
            return {}
        for k in data:
            data[k] = unfold_batching(data[k])
        image, idx = data['image'], data['idx'].item()
        with torch.no_grad():
            frame = self.val_pipeline({'pose': data['pose']})
            ret = self.batchify_forward(frame, is_test=True)
        rgb = recover_shape(ret['rgb'], frame['src_shape']).cpu().numpy()
        return {'rgb': rgb, 'gt_img': image.cpu().numpy(), 'idx': idx}

    def set_val_pipeline(self, func):
        self.val_pipeline = func
