"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's Mip-NeRF sampling / encoding /
rendering path (BASELINE config #3), fp32 arithmetic in the reference's
operation order.  Pinned against the reference's OWN torch functions: tests/golden/ref_mipnerf.npz
(made by tests/golden/make_golden_mip.py with /root/reference imported) and, live, when
/root/reference exists (tests/test_mip_oracle_pinning.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference (paths relative to /root/reference/):
  xrnerf/datasets/pipelines/create.py:486-531          GetZvals
  xrnerf/models/networks/utils/mip.py:7-62             sorted_piecewise_constant_pdf
  xrnerf/models/networks/utils/mip.py:65-148           lift_gaussian / conical_frustum_to_gaussian / cylinder / cast_rays
  xrnerf/models/networks/utils/mip.py:151-176          resample_along_rays
  xrnerf/models/embedders/mipnerf_embedder.py:34-99    expected_sin / integrated_pos_enc / pos_enc / forward
  xrnerf/models/renders/nerf_render.py:45-98           NerfRender.forward
  xrnerf/models/renders/mipnerf_render.py:12-33        MipNerfRender.get_disp_map / get_weights
"""
import numpy as np

F = np.float32
EPS32 = F(np.finfo(np.float32).eps)
HALF_PI = F(0.5) * F(np.pi)          # 0.5 * torch.tensor(math.pi): fp32 pi halved


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- GetZvals (create.py:502-531)
def z_vals(near, far, n, lindisp=False, z_rand=None):
    """near, far [R,1]; z_rand [R,n] uniform draws or None (not randomized) -> [R,n]"""
    near, far = _f(near), _f(far)
    t = np.linspace(0., 1., n).astype(np.float32)
    if not lindisp:
        z = near * (F(1.) - t) + far * t
    else:
        z = F(1.) / (F(1.) / near * (F(1.) - t) + F(1.) / far * t)
    z = np.broadcast_to(z, (near.shape[0], n)).astype(np.float32)
    if z_rand is not None:
        mids = F(0.5) * (z[..., 1:] + z[..., :-1])
        upper = np.concatenate([mids, z[..., -1:]], -1)
        lower = np.concatenate([z[..., :1], mids], -1)
        z = lower + (upper - lower) * _f(z_rand)
    return _f(z)


# ---------------------------------------------------------------- mip.py:65-148
def lift_gaussian(d, t_mean, t_var, r_var):
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = np.maximum(F(1e-10), np.sum(d ** 2, axis=-1, keepdims=True, dtype=np.float32))
    d_outer_diag = d ** 2
    null_outer_diag = F(1) - d_outer_diag / d_mag_sq
    t_cov_diag = t_var[..., None] * d_outer_diag[..., None, :]
    xy_cov_diag = r_var[..., None] * null_outer_diag[..., None, :]
    return _f(mean), _f(t_cov_diag + xy_cov_diag)


def conical_frustum_to_gaussian(d, t0, t1, base_radius):
    mu = (t0 + t1) / F(2)
    hw = (t1 - t0) / F(2)
    t_mean = mu + (F(2) * mu * hw ** 2) / (F(3) * mu ** 2 + hw ** 2)
    t_var = (hw ** 2) / F(3) - F(4 / 15) * ((hw ** 4 * (F(12) * mu ** 2 - hw ** 2)) / (F(3) * mu ** 2 + hw ** 2) ** 2)
    r_var = base_radius ** 2 * ((mu ** 2) / F(4) + F(5 / 12) * hw ** 2 - F(4 / 15) * (hw ** 4) / (F(3) * mu ** 2 + hw ** 2))
    return lift_gaussian(d, _f(t_mean), _f(t_var), _f(r_var))


def cylinder_to_gaussian(d, t0, t1, radius):
    t_mean = (t0 + t1) / F(2)
    r_var = radius ** 2 / F(4)
    t_var = (t1 - t0) ** 2 / F(12)
    return lift_gaussian(d, _f(t_mean), _f(t_var), _f(np.broadcast_to(r_var, t_mean.shape)))


def cast_rays(z, origins, directions, radii, ray_shape='cone'):
    """z [R,S+1], origins/directions [R,3], radii [R,1] -> means, covs [R,S,3]"""
    z, origins, directions, radii = _f(z), _f(origins), _f(directions), _f(radii)
    t0, t1 = z[..., :-1], z[..., 1:]
    fn = conical_frustum_to_gaussian if ray_shape == 'cone' else cylinder_to_gaussian
    means, covs = fn(directions, t0, t1, radii)
    return _f(means + origins[..., None, :]), covs


# ---------------------------------------------------------------- mipnerf_embedder.py:34-99
def integrated_pos_enc(means, covs, min_deg, max_deg):
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype=np.float32)
    shape = list(means.shape[:-1]) + [-1]
    y = np.reshape(means[..., None, :] * scales[:, None], shape)
    y_var = np.reshape(covs[..., None, :] * scales[:, None] ** 2, shape)
    x = np.concatenate([y, y + HALF_PI], -1)
    x_var = np.concatenate([y_var, y_var], -1)
    return _f(np.exp(F(-0.5) * x_var) * np.sin(x))


def pos_enc(x, min_deg, max_deg, append_identity=True):
    x = _f(x)
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype=np.float32)
    xb = np.reshape(x[..., None, :] * scales[:, None], list(x.shape[:-1]) + [-1])
    four = np.sin(np.concatenate([xb, xb + HALF_PI], -1))
    return _f(np.concatenate([x, four], -1) if append_identity else four)


def embed(z, origins, directions, viewdirs, radii, min_deg=0, max_deg=16, min_deg_view=0, max_deg_view=4,
          append_identity=True, ray_shape='cone'):
    """MipNerfEmbedder.forward on sample_along_rays' output: -> [R*S, 6*(max-min) + 6*(maxv-minv) (+3)]"""
    means, covs = cast_rays(z, origins, directions, radii, ray_shape)
    ipe = integrated_pos_enc(means, covs, min_deg, max_deg)
    pe = pos_enc(viewdirs, min_deg_view, max_deg_view, append_identity)
    R, S = ipe.shape[:2]
    pe = np.broadcast_to(pe[:, None, :], (R, S, pe.shape[-1]))
    return _f(np.concatenate([ipe, pe], -1).reshape(R * S, -1))


# ---------------------------------------------------------------- mip.py:7-62, 151-176
def blurred_weights(weights, resample_padding):
    w = _f(weights)
    pad = np.concatenate([w[..., :1], w, w[..., -1:]], -1)
    wmax = np.maximum(pad[..., :-1], pad[..., 1:])
    return _f(F(0.5) * (wmax[..., :-1] + wmax[..., 1:]) + F(resample_padding))


def sorted_piecewise_constant_pdf(bins, weights, num_samples, rand=None):
    """bins [R,n+1], weights [R,n]; rand [R,num_samples] uniform draws (randomized) or None"""
    bins, weights = _f(bins), _f(weights).copy()
    eps = F(1e-5)
    weight_sum = np.sum(weights, axis=-1, keepdims=True, dtype=np.float32)
    padding = np.maximum(F(0), eps - weight_sum)
    weights += padding / F(weights.shape[-1])
    weight_sum = weight_sum + padding
    pdf = weights / weight_sum
    # torch.cumsum on the CPU accumulates fp32 inputs in double (at::acc_type<float, false>)
    cdf = np.minimum(F(1), np.cumsum(pdf[..., :-1].astype(np.float64), axis=-1).astype(np.float32))
    lead = list(cdf.shape[:-1]) + [1]                   # (one interval: cdf is empty here, the result is [0, 1])
    cdf = np.concatenate([np.zeros(lead, np.float32), cdf, np.ones(lead, np.float32)], -1)
    if rand is not None:
        s = F(1 / num_samples)                           # python float 1/n, promoted to fp32 by the tensor op
        u = np.arange(num_samples).astype(np.float32) * s
        u = u + _f(rand) * F(np.float64(1 / num_samples) - np.float64(EPS32))
        u = np.minimum(u, F(1. - np.float64(EPS32)))
    else:
        u = np.linspace(0., 1. - np.float64(EPS32), num_samples).astype(np.float32)
        u = np.broadcast_to(u, (cdf.shape[0], num_samples))
    u = _f(u)
    mask = u[..., None, :] >= cdf[..., :, None]

    def find_interval(x):
        x0 = np.max(np.where(mask, x[..., None], x[..., :1, None]), -2)
        x1 = np.min(np.where(~mask, x[..., None], x[..., -1:, None]), -2)
        return x0, x1

    b0, b1 = find_interval(bins)
    c0, c1 = find_interval(cdf)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = (u - c0) / (c1 - c0)
    t = np.clip(np.nan_to_num(t, nan=0.0), 0, 1).astype(np.float32)
    return _f(b0 + t * (b1 - b0))


def resample(z, weights, resample_padding, rand=None):
    """resample_along_rays' new z_vals: [R,S+1]"""
    return sorted_piecewise_constant_pdf(z, blurred_weights(weights, resample_padding), np.asarray(z).shape[-1], rand)


# ---------------------------------------------------------------- nerf_render.py:45-98 + mipnerf_render.py
def softplus(x):
    x = _f(x)
    with np.errstate(over='ignore'):
        return _f(np.where(x > F(20), x, np.log1p(np.exp(np.minimum(x, F(20))))))


def render(raw, z, rays_d, density_bias=-1., rgb_padding=0.001, white_bkgd=True, activation='softplus'):
    """raw [R,S,4], z [R,S+1], rays_d [R,3] -> rgb [R,3], distance ('disp') [R], acc [R], weights [R,S]"""
    raw, z, rays_d = _f(raw), _f(z), _f(rays_d)
    dists = (z[..., 1:] - z[..., :-1]) * np.sqrt(np.sum(rays_d ** 2, -1, dtype=np.float32))[..., None]
    rgb = F(1) / (F(1) + np.exp(-raw[..., :3]))
    rgb = rgb * F(1 + 2 * rgb_padding) - F(rgb_padding)
    x = raw[..., 3] + F(density_bias)
    dd = _f((softplus(x) if activation == 'softplus' else np.maximum(x, F(0))) * dists)
    alpha = F(1) - np.exp(-dd)
    csum = np.cumsum(dd[..., :-1].astype(np.float64), axis=-1).astype(np.float32)   # CPU torch.cumsum: double accumulate
    weights = _f(alpha * np.exp(-np.concatenate([np.zeros_like(dd[..., :1]), csum], -1)))
    rgb_map = np.sum(weights[..., None] * rgb, -2, dtype=np.float32)
    acc = np.sum(weights, -1, dtype=np.float32)
    z_mids = F(0.5) * (z[..., :-1] + z[..., 1:])
    depth = np.sum(weights * z_mids, -1, dtype=np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        q = depth / acc
    q = np.where(np.isnan(q), F(np.inf), q)
    q = np.where(np.isposinf(q), np.finfo(np.float32).max, q)       # nan_to_num(x, nan=inf): +inf -> max float
    q = np.where(np.isneginf(q), np.finfo(np.float32).min, q)
    disp = np.maximum(np.minimum(q, z[:, -1]), z[:, 0])
    if white_bkgd:
        rgb_map = rgb_map + (F(1) - acc[..., None])
    return _f(rgb_map), _f(disp), _f(acc), weights


def render_bwd(raw, z, rays_d, grad_rgb, density_bias=-1., rgb_padding=0.001, white_bkgd=True, activation='softplus'):
    """dL/draw [R,S,4] for a loss that depends on the rendered rgb only (float64 analytic adjoint)"""
    raw, z, rays_d, g = (np.asarray(a, np.float64) for a in (raw, z, rays_d, grad_rgb))
    dists = (z[..., 1:] - z[..., :-1]) * np.sqrt(np.sum(rays_d ** 2, -1))[..., None]
    s = 1 / (1 + np.exp(-raw[..., :3]))
    c = s * (1 + 2 * rgb_padding) - rgb_padding
    x = raw[..., 3] + density_bias
    if activation == 'softplus':
        sp = np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))
        dsp = np.where(x > 20, 1.0, 1 / (1 + np.exp(-x)))
    else:
        sp, dsp = np.maximum(x, 0), (x > 0).astype(np.float64)
    dd = sp * dists
    T = np.exp(-np.concatenate([np.zeros_like(dd[..., :1]), np.cumsum(dd[..., :-1], -1)], -1))
    w = (1 - np.exp(-dd)) * T
    cp = c - (1.0 if white_bkgd else 0.0)
    gc = np.sum(g[:, None, :] * cp, -1)                       # [R,S]  sum_ch g_ch c'_k,ch
    wg = w * gc
    suffix = np.cumsum(wg[..., ::-1], -1)[..., ::-1] - wg     # sum_{i>k} w_i gc_i
    d_dd = T * np.exp(-dd) * gc - suffix
    out = np.zeros_like(raw)
    out[..., 3] = d_dd * dists * dsp
    out[..., :3] = g[:, None, :] * w[..., None] * s * (1 - s) * (1 + 2 * rgb_padding)
    return out.astype(np.float32)


# ---------------------------------------------------------------- the same path in pure PyTorch (autograd-capable)
# Used as bench.py's cpu_baseline leg for config #3 (the reference's own code IS pure PyTorch but cannot travel to the
# GPU box) and as a second statement of the maths in the tests.  Written with searchsorted / cumsum rather than the
# reference's O(n^2) mask construction; pinned against the same fixture (tests/test_mip_oracle_pinning.py).
def torch_embed(z, o, d, vd, radii, min_deg=0, max_deg=16, min_deg_view=0, max_deg_view=4, append_identity=True):
    import torch
    t0, t1 = z[..., :-1], z[..., 1:]
    mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
    den = 3 * mu**2 + hw**2
    t_mean = mu + (2 * mu * hw**2) / den
    t_var = hw**2 / 3 - (4 / 15) * ((hw**4 * (12 * mu**2 - hw**2)) / den**2)
    r_var = radii**2 * (mu**2 / 4 + (5 / 12) * hw**2 - (4 / 15) * hw**4 / den)
    d2 = d * d
    mag = d2.sum(-1, keepdim=True).clamp_min(1e-10)
    mean = d[:, None, :] * t_mean[..., None] + o[:, None, :]
    cov = t_var[..., None] * d2[:, None, :] + r_var[..., None] * (1 - d2 / mag)[:, None, :]
    sc = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)], dtype=z.dtype)
    y = (mean[..., None, :] * sc[:, None]).flatten(-2)
    yv = (cov[..., None, :] * sc[:, None] ** 2).flatten(-2)
    ipe = torch.exp(-0.5 * torch.cat([yv, yv], -1)) * torch.sin(torch.cat([y, y + float(HALF_PI)], -1))
    sv = torch.tensor([2.0 ** i for i in range(min_deg_view, max_deg_view)], dtype=z.dtype)
    xb = (vd[..., None, :] * sv[:, None]).flatten(-2)
    pe = torch.sin(torch.cat([xb, xb + float(HALF_PI)], -1))
    if append_identity:
        pe = torch.cat([vd, pe], -1)
    R, S = ipe.shape[:2]
    return torch.cat([ipe, pe[:, None, :].expand(R, S, pe.shape[-1])], -1).reshape(R * S, -1)


def torch_render(raw, z, rays_d, density_bias=-1., rgb_padding=0.001, white_bkgd=True, activation='softplus'):
    import torch
    import torch.nn.functional as Fn
    dists = (z[..., 1:] - z[..., :-1]) * rays_d.norm(dim=-1, keepdim=True)
    rgb = torch.sigmoid(raw[..., :3]) * (1 + 2 * rgb_padding) - rgb_padding
    x = raw[..., 3] + density_bias
    dd = (Fn.softplus(x) if activation == 'softplus' else Fn.relu(x)) * dists
    before = torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], -1)], -1)
    w = (1 - torch.exp(-dd)) * torch.exp(-before)
    acc = w.sum(-1)
    col = (w[..., None] * rgb).sum(-2)
    depth = (w * (0.5 * (z[..., :-1] + z[..., 1:]))).sum(-1)
    dist = torch.maximum(torch.minimum(torch.nan_to_num(depth / acc, float('inf')), z[:, -1]), z[:, 0])
    return (col + (1 - acc[..., None]) if white_bkgd else col), dist, acc, w


def torch_resample(z, weights, resample_padding, rand=None):
    import torch
    w = weights.detach()
    pad = torch.cat([w[..., :1], w, w[..., -1:]], -1)
    wmax = torch.maximum(pad[..., :-1], pad[..., 1:])
    wb = 0.5 * (wmax[..., :-1] + wmax[..., 1:]) + resample_padding
    n_z = z.shape[-1]
    ws = wb.sum(-1, keepdim=True)
    padding = (1e-5 - ws).clamp_min(0)
    wb = wb + padding / wb.shape[-1]
    pdf = wb / (ws + padding)
    cdf = torch.cumsum(pdf[..., :-1], -1).clamp_max(1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf, torch.ones_like(cdf[..., :1])], -1)
    eps = float(EPS32)
    if rand is not None:
        u = torch.arange(n_z, dtype=z.dtype) * (1 / n_z) + rand * (1 / n_z - eps)
        u = u.clamp_max(1. - eps)
    else:
        u = torch.linspace(0., 1. - eps, n_z).expand(z.shape[0], n_z)
    i0 = (torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True) - 1).clamp(0, n_z - 1)
    i1 = (i0 + 1).clamp_max(n_z - 1)
    c0, c1, b0, b1 = cdf.gather(-1, i0), cdf.gather(-1, i1), z.gather(-1, i0), z.gather(-1, i1)
    t = torch.nan_to_num((u - c0) / (c1 - c0), 0.0).clamp(0, 1)
    return (b0 + t * (b1 - b0)).detach()


def torch_train_step(mlp, data, num_levels=2, resample_padding=0.01, coarse_loss_mult=0.1, render_kw=None, rand=None):
    """networks/mipnerf.py:24-60 with `mlp.run_mlp` (xrnerf_amd.vanilla.NerfMLP or the reference's) on any device"""
    import torch
    render_kw = render_kw or {}
    z, losses, w = data['z_vals'], [], None
    mask = torch.broadcast_to(data['lossmult'], data['target_s'].shape)
    for level in range(num_levels):
        if level > 0:
            r = rand if rand is not None else torch.rand(z.shape)
            z = torch_resample(z, w, resample_padding, r)
        e = torch_embed(z, data['rays_o'], data['rays_d'], data['viewdirs'], data['radii'])
        raw = mlp.batchify_run_mlp(e).reshape(z.shape[0], z.shape[1] - 1, 4)
        rgb, _, _, w = torch_render(raw, z, data['rays_d'], **render_kw)
        losses.append((mask * (rgb - data['target_s']) ** 2).sum() / mask.sum())
    return losses[-1] + coarse_loss_mult * sum(losses[:-1]), losses
