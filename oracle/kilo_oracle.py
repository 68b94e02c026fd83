"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's KiloNeRF rendering path (BASELINE config #5,
SURVEY.md section 8f row 4): sample -> network assignment / occupancy filter / grouping, local coordinates,
Fourier features, the per-network tiny MLP, classic NeRF compositing.

The reference's production path calls an external, un-vendored CUDA library (`kilonerf_cuda`: global_to_local,
compute_fourier_features, multimatmul_magma_grouped_static); this oracle follows the IN-TREE PyTorch statements of
the same operations, which the reference itself uses in the distillation phase that produces the very weights the
fast path consumes -- so they define the semantics -- and is pinned against them (tests/golden/ref_kilonerf.npz from
tests/golden/make_golden_kilo.py, and live in tests/test_kilo_oracle_pinning.py):
  xrnerf/models/networks/utils/transforms.py:35-45      convert_to_local_coords_multi   (<-> global_to_local)
  xrnerf/models/networks/utils/transforms.py:57-151     reorder_points_and_dirs
  xrnerf/models/embedders/kilonerf_fourier_embedder.py:33-52   Fourier features, 'pytorch' implementation
  xrnerf/models/mlps/multi_modules.py:590-668           MultiNetwork.forward (late_feed_direction, relu)
  xrnerf/models/mlps/multi_modules.py:160-195           naive_multimatmul* (<-> multimatmul_magma_grouped_static)
  xrnerf/models/mlps/kilonerf_mlp.py:138-190            KiloNerfMLP.forward (scatter back, zeros elsewhere)
  xrnerf/models/renders/nerf_render.py:30-98            NerfRender.forward (relu density, cumprod weights)
  xrnerf/datasets/pipelines/create.py:577-601           GetPts

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

F = np.float32


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def get_pts(rays_o, rays_d, z_vals):
    """create.py:588-597: o + d * z (fp32 multiply, then add)"""
    return _f(_f(rays_o)[:, None, :] + _f(rays_d)[:, None, :] * _f(z_vals)[:, :, None])


def assign(pts, gmin, gmax, fixed_res, res, occupancy, num_networks):
    """transforms.py:69-120 -> (network index per sample [n] int64, active mask [n] bool)
    pts [n,3]; gmin/gmax [3]; occupancy: flat bool grid of resolution `res` or None"""
    pts, gmin, gmax = _f(pts), _f(gmin), _f(gmax)
    fixed = np.asarray(fixed_res, np.int64)
    size = gmax - gmin
    voxel = (size / fixed.astype(np.float32)).astype(np.float32)
    idx3 = ((pts - gmin) / voxel).astype(np.int64)                   # truncation toward zero like tensor.to(long)
    strides = np.array([fixed[2] * fixed[1], fixed[2], 1], np.int64)
    net = (idx3 * strides).sum(1)
    eps = F(0.001)
    active = np.logical_and((pts > gmin + eps).all(1), (pts < gmax - eps).all(1))
    if occupancy is not None:
        r = np.asarray(res, np.int64)
        ovox = (size / r.astype(np.float32)).astype(np.float32)
        oi = ((pts - gmin) / ovox).astype(np.int64)
        oi = np.minimum(np.maximum(oi, 0), r - 1)
        oflat = (oi * np.array([r[2] * r[1], r[2], 1], np.int64)).sum(1)
        active = np.logical_and(active, np.asarray(occupancy).reshape(-1)[oflat].astype(bool))
    active = np.logical_and(active, np.logical_and(net >= 0, net < num_networks))
    return net, active


def group(net, active, num_networks):
    """transforms.py:118-141: indices of the active samples sorted by network (STABLE here; the reference's
    torch.sort is not, which only permutes rows inside a network's segment) + batch_size_per_network"""
    idx = np.nonzero(active)[0]
    order = idx[np.argsort(net[idx], kind='stable')]
    counts = np.bincount(net[idx], minlength=num_networks).astype(np.int64)
    return order, counts


def to_local(p, dmin, dmax):
    """transforms.py:35-45: 2 * (p - min) / (max - min) - 1"""
    p, dmin, dmax = _f(p), _f(dmin), _f(dmax)
    return _f(F(2) * (p - dmin) / (dmax - dmin) - F(1))


def fourier(x, num_frequencies):
    """kilonerf_fourier_embedder.py:33-52: per input channel [x, cos(x f_0..f_{F-1}), sin(x f_0..f_{F-1})], f_k = 2^k"""
    x = _f(x)
    f = (2.0 ** np.arange(num_frequencies)).astype(np.float32)
    xf = x[..., None] * f
    out = np.concatenate([x[..., None], np.cos(xf), np.sin(xf)], -1)
    return _f(out.reshape(x.shape[0], -1))


class TinyNets:
    """weights of the distilled multi network in the reference's `multimatmul` layout (kilonerf_mlp.py:96-127):
    weight [N, in, out], bias [N, out] per layer of MultiNetwork (late_feed_direction, no refeed)"""

    def __init__(self, pts_w, pts_b, alpha_w, alpha_b, feat_w, feat_b, dir_w, dir_b, rgb_w, rgb_b):
        self.pts_w, self.pts_b = [_f(w) for w in pts_w], [_f(b) for b in pts_b]
        self.alpha_w, self.alpha_b, self.feat_w, self.feat_b = _f(alpha_w), _f(alpha_b), _f(feat_w), _f(feat_b)
        self.dir_w, self.dir_b, self.rgb_w, self.rgb_b = _f(dir_w), _f(dir_b), _f(rgb_w), _f(rgb_b)
        self.num_networks = self.alpha_w.shape[0]

    @staticmethod
    def random(num_networks, rng, pos_ch=63, dir_ch=27, hidden=32, n_hidden=2, dir_hidden=32, scale=1.0):
        def lin(i, o):
            b = scale / np.sqrt(i)
            return rng.uniform(-b, b, (num_networks, i, o)).astype(np.float32), rng.uniform(-b, b, (num_networks, o)).astype(np.float32)
        pw, pb = [], []
        for l in range(n_hidden):
            w, b = lin(pos_ch if l == 0 else hidden, hidden)
            pw.append(w); pb.append(b)
        aw, ab = lin(hidden, 1)
        fw, fb = lin(hidden, hidden)
        dw, db = lin(hidden + dir_ch, dir_hidden)
        rw, rb = lin(dir_hidden, 3)
        return TinyNets(pw, pb, aw, ab, fw, fb, dw, db, rw, rb)

    def forward(self, n, pos_emb, dir_emb):
        """MultiNetwork.forward for rows that all belong to network n (multi_modules.py:590-668)"""
        h = pos_emb
        for w, b in zip(self.pts_w, self.pts_b):
            h = np.maximum(h @ w[n] + b[n], 0)
        alpha = h @ self.alpha_w[n] + self.alpha_b[n]
        feat = h @ self.feat_w[n] + self.feat_b[n]
        h = np.maximum(np.concatenate([feat, dir_emb], -1) @ self.dir_w[n] + self.dir_b[n], 0)
        rgb = h @ self.rgb_w[n] + self.rgb_b[n]
        return np.concatenate([rgb, alpha], -1).astype(np.float32)


def mlp_raw(rays_o, rays_d, viewdirs, z_vals, gmin, gmax, fixed_res, res, occupancy, dmins, dmaxs, nets,
            pos_freqs=10, dir_freqs=4, pts=None):
    """KiloNerfMLP.forward (kilonerf_mlp.py:138-190): raw [R,S,4], zeros where no network is evaluated.
    Also returns (net, active, counts) for the integer-exact checks."""
    R, S = np.asarray(z_vals).shape
    if pts is None:
        pts = get_pts(rays_o, rays_d, z_vals)
    flat = _f(pts).reshape(-1, 3)
    net, active = assign(flat, gmin, gmax, fixed_res, res, occupancy, nets.num_networks)
    order, counts = group(net, active, nets.num_networks)
    raw = np.zeros((R * S, 4), np.float32)
    dirs = np.repeat(_f(viewdirs), S, axis=0)
    start = 0
    for n in np.nonzero(counts)[0]:
        rows = order[start:start + counts[n]]
        start += counts[n]
        local = to_local(flat[rows], _f(dmins)[n], _f(dmaxs)[n])
        raw[rows] = nets.forward(n, fourier(local, pos_freqs), fourier(dirs[rows], dir_freqs))
    return raw.reshape(R, S, 4), net, active, counts


def nerf_render(raw, z_vals, rays_d, white_bkgd=True):
    """NerfRender.forward, raw_noise_std = 0 (nerf_render.py:45-98): z_vals are SAMPLE positions [R,S]
    -> rgb [R,3], disp [R], acc [R], weights [R,S]"""
    raw, z, rays_d = _f(raw), _f(z_vals), _f(rays_d)
    dists = np.concatenate([z[..., 1:] - z[..., :-1], np.full_like(z[..., :1], 1e10)], -1)
    dists = dists * np.sqrt(np.sum(rays_d ** 2, -1, dtype=np.float32))[..., None]
    rgb = F(1) / (F(1) + np.exp(-raw[..., :3]))
    dd = np.maximum(raw[..., 3], F(0)) * dists
    alpha = F(1) - np.exp(-dd)
    # torch.cumprod on the CPU accumulates fp32 in double (at::acc_type<float, false>)
    trans = np.cumprod(np.concatenate([np.ones_like(alpha[..., :1]), F(1) - alpha + F(1e-10)], -1).astype(np.float64),
                       -1)[..., :-1].astype(np.float32)
    w = _f(alpha * trans)
    rgb_map = np.sum(w[..., None] * rgb, -2, dtype=np.float32)
    acc = np.sum(w, -1, dtype=np.float32)
    depth = np.sum(w * z, -1, dtype=np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        disp = F(1) / np.maximum(F(1e-10), depth / acc)          # max(1e-10, nan) = nan in torch as well
    if white_bkgd:
        rgb_map = rgb_map + (F(1) - acc[..., None])
    return _f(rgb_map), _f(disp), _f(acc), w


# ---------------------------------------------------------------- gradients of the tiny MLPs (fine-tuning)
def tiny_backward(nets, n, pos_emb, dir_emb, d_raw):
    """dL/d(parameters of network n) for rows (pos_emb [m,P], dir_emb [m,D]) with dL/draw [m,4] = (rgb3, alpha); float64.
    MultiNetwork.forward's adjoint (multi_modules.py:590-668; AddMultiMatMul.backward :215-236 = the same products).
    -> dict name -> gradient in the `multimatmul` layout ([in, out] weights, [out] biases)"""
    f = lambda a: np.asarray(a, np.float64)
    x, dv, g = f(pos_emb), f(dir_emb), f(d_raw)
    acts, h = [], x
    for w, b in zip(nets.pts_w, nets.pts_b):
        pre = h @ f(w[n]) + f(b[n])
        acts.append((h, pre))
        h = np.maximum(pre, 0)
    feat = h @ f(nets.feat_w[n]) + f(nets.feat_b[n])
    din = np.concatenate([feat, dv], -1)
    pre_d = din @ f(nets.dir_w[n]) + f(nets.dir_b[n])
    hd = np.maximum(pre_d, 0)
    out = {}
    g_rgb, g_a = g[:, :3], g[:, 3:4]
    out['rgb_linear.weight'], out['rgb_linear.bias'] = hd.T @ g_rgb, g_rgb.sum(0)
    d_hd = (g_rgb @ f(nets.rgb_w[n]).T) * (pre_d > 0)
    out['direction_layer.weight'], out['direction_layer.bias'] = din.T @ d_hd, d_hd.sum(0)
    d_feat = d_hd @ f(nets.dir_w[n])[:feat.shape[1]].T
    out['feature_linear.weight'], out['feature_linear.bias'] = h.T @ d_feat, d_feat.sum(0)
    out['alpha_linear.weight'], out['alpha_linear.bias'] = h.T @ g_a, g_a.sum(0)
    d_h = d_feat @ f(nets.feat_w[n]).T + g_a @ f(nets.alpha_w[n]).T
    for l in reversed(range(len(nets.pts_w))):
        hin, pre = acts[l]
        d_pre = d_h * (pre > 0)
        out['pts_linears.%d.weight' % l], out['pts_linears.%d.bias' % l] = hin.T @ d_pre, d_pre.sum(0)
        d_h = d_pre @ f(nets.pts_w[l][n]).T
    return out


def mlp_raw_backward(d_raw, rays_o, rays_d, viewdirs, z_vals, gmin, gmax, fixed_res, res, occupancy, dmins, dmaxs, nets,
                     pos_freqs=10, dir_freqs=4, pts=None):
    """gradients of every network's parameters for dL/draw [R,S,4] (rows without a network do not contribute):
    dict name -> [N, ...] float64 arrays in the `multimatmul` layout"""
    R, S = np.asarray(z_vals).shape
    if pts is None:
        pts = get_pts(rays_o, rays_d, z_vals)
    flat = _f(pts).reshape(-1, 3)
    net, active = assign(flat, gmin, gmax, fixed_res, res, occupancy, nets.num_networks)
    order, counts = group(net, active, nets.num_networks)
    dirs = np.repeat(_f(viewdirs), S, axis=0)
    g = np.asarray(d_raw, np.float64).reshape(-1, 4)
    N = nets.num_networks
    shapes = {'rgb_linear.weight': nets.rgb_w.shape, 'rgb_linear.bias': nets.rgb_b.shape,
              'direction_layer.weight': nets.dir_w.shape, 'direction_layer.bias': nets.dir_b.shape,
              'feature_linear.weight': nets.feat_w.shape, 'feature_linear.bias': nets.feat_b.shape,
              'alpha_linear.weight': nets.alpha_w.shape, 'alpha_linear.bias': nets.alpha_b.shape}
    for l, (w, b) in enumerate(zip(nets.pts_w, nets.pts_b)):
        shapes['pts_linears.%d.weight' % l], shapes['pts_linears.%d.bias' % l] = w.shape, b.shape
    grads = {k: np.zeros(v, np.float64) for k, v in shapes.items()}
    start = 0
    for n in np.nonzero(counts)[0]:
        rows = order[start:start + counts[n]]
        start += counts[n]
        local = to_local(flat[rows], _f(dmins)[n], _f(dmaxs)[n])
        for k, v in tiny_backward(nets, n, fourier(local, pos_freqs), fourier(dirs[rows], dir_freqs), g[rows]).items():
            grads[k][n] = v.reshape(grads[k][n].shape)
    return grads
