"""TEST INFRASTRUCTURE ONLY.  numpy/ctypes front-end of the CPU oracle.

`port`  = oracle/libngp_oracle.so  (our C restatement, oracle/ngp_oracle.c; travels everywhere)
`ref`   = oracle/_ref/libref_raymarch.so (the reference's own kernels compiled for the CPU; only
          where /root/reference exists or the prebuilt .so travelled with the snapshot)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build as _build  # noqa: E402

G3 = 128 ** 3
MIN_STEP = np.float32(1.73205080757) / np.float32(1024)
MAX_WARP_STEP = MIN_STEP * np.float32(128)

_port = None
_ref = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def port():
    global _port
    if _port is None:
        _port = C.CDLL(_build.build_port())
        _port.xo_density_mean.restype = C.c_float
        _port.xo_huber_loss_grad.restype = C.c_float
        _port.xo_morton3d_api.restype = C.c_uint32
        _port.xo_morton3d_invert_api.restype = C.c_uint32
    return _port


def ref():
    """The reference-kernel library, or None when unavailable."""
    global _ref
    if _ref is None:
        path = _build.build_ref()
        if path is None:
            return None
        _ref = C.CDLL(path)
    return _ref


def have_ref():
    return ref() is not None


# ------------------------------------------------------------------ PCG32
def pcg32_host_state(ncalls, seed=9121):
    s, i = C.c_uint64(), C.c_uint64()
    port().xo_pcg32_host_state(C.c_uint64(seed), C.c_uint64(ncalls), C.byref(s), C.byref(i))
    return s.value, i.value


def pcg32_probe(seed, advance_by, lib=None):
    lib = lib or port()
    fn = lib.xo_pcg32_probe if lib is port() else lib.ref_pcg32_probe
    s, i = C.c_uint64(), C.c_uint64()
    u5 = np.zeros(5, np.uint32)
    f3 = np.zeros(3, np.float32)
    fn(C.c_uint64(seed), C.c_uint64(advance_by), C.byref(s), C.byref(i), _p(u5), _p(f3))
    return s.value, i.value, u5, f3


# ------------------------------------------------------------------ K1
def rays_sampler(rays_o, rays_d, bitfield, aabb=(0.0, 1.0), near=0.05, cone=1.0 / 256, max_samples=None,
                 rng_calls=0, backend='port', img_ids=None, metadata=None, xforms=None):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    if max_samples is None:
        max_samples = n * 1024
    coords = np.zeros((max_samples, 7), np.float32)
    index = np.zeros((n, 1), np.int32)
    numsteps = np.zeros((n, 2), np.int32)
    counter = np.zeros(2, np.int32)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    if backend == 'port':
        st, inc = pcg32_host_state(rng_calls)
        port().xo_rays_sampler(_p(rays_o), _p(rays_d), _p(bitfield), C.c_int(n), C.c_float(aabb[0]),
                               C.c_float(aabb[1]), C.c_float(near), C.c_float(cone),
                               C.c_uint32(max_samples), C.c_uint64(st), C.c_uint64(inc), _p(coords),
                               _p(index), _p(numsteps), _p(counter))
    else:
        n_img = 1
        if img_ids is None:
            img_ids = np.zeros((n, 1), np.int32)
        if metadata is None:
            metadata = np.zeros((n_img, 11), np.float32)
        if xforms is None:
            xforms = np.zeros((n_img, 4, 3), np.float32)
        img_ids, metadata, xforms = _i32(img_ids), _f32(metadata), _f32(xforms)
        ref().ref_rays_sampler(_p(rays_o), _p(rays_d), _p(bitfield), _p(metadata), _p(img_ids), _p(xforms),
                               C.c_int(n), C.c_int(metadata.shape[0]), C.c_float(aabb[0]), C.c_float(aabb[1]),
                               C.c_float(near), C.c_float(cone), C.c_int(max_samples), C.c_uint64(rng_calls),
                               _p(coords), _p(index), _p(numsteps), _p(counter))
    return coords, index, numsteps, counter


# ------------------------------------------------------------------ K2
def compacted_coord(coords, numsteps, max_compacted, network_output=None, backend='port',
                    rgb_act=2, density_act=3, aabb=(0.0, 1.0)):
    coords, numsteps = _f32(coords), _i32(numsteps)
    n = numsteps.shape[0]
    out = np.zeros((max_compacted, 7), np.float32)
    nc = np.zeros((n, 2), np.int32)
    rc = np.zeros(1, np.int32)
    sc = np.zeros(1, np.int32)
    if backend == 'port':
        port().xo_compacted_coord(_p(coords), _p(numsteps), C.c_int(n), C.c_uint32(max_compacted), _p(out),
                                  _p(nc), _p(rc), _p(sc))
    else:
        if network_output is None:
            network_output = np.zeros((coords.shape[0], 4), np.float32)
        network_output = _f32(network_output)
        ref().ref_compacted_coord(_p(network_output), _p(coords), _p(numsteps), C.c_int(n),
                                  C.c_int(coords.shape[0]), C.c_int(max_compacted), C.c_int(rgb_act),
                                  C.c_int(density_act), C.c_float(aabb[0]), C.c_float(aabb[1]), _p(out),
                                  _p(nc), _p(rc), _p(sc))
    return out, nc, rc, sc


# ------------------------------------------------------------------ K3/K4/K5
def calc_rgb_forward(raw, coords, numsteps, numsteps_c, bg, rgb_act=2, density_act=3, backend='port',
                     aabb=(0.0, 1.0)):
    raw, coords, numsteps, numsteps_c, bg = _f32(raw), _f32(coords), _i32(numsteps), _i32(numsteps_c), _f32(bg)
    n = numsteps.shape[0]
    out = np.zeros((n, 3), np.float32)
    if backend == 'port':
        port().xo_calc_rgb_forward(_p(raw), _p(coords), _p(numsteps), _p(numsteps_c), _p(bg), C.c_int(n),
                                   C.c_int(rgb_act), C.c_int(density_act), _p(out))
    else:
        ref().ref_calc_rgb_forward(_p(raw), _p(coords), _p(numsteps), _p(numsteps_c), _p(bg), C.c_int(n),
                                   C.c_int(raw.shape[0]), C.c_int(rgb_act), C.c_int(density_act),
                                   C.c_float(aabb[0]), C.c_float(aabb[1]), _p(out))
    return out


def calc_rgb_backward(raw, numsteps_c, coords, grad_rgb, rgb_final, density_grid_mean, rgb_act=2,
                      density_act=3, backend='port', aabb=(0.0, 1.0)):
    raw, coords, numsteps_c = _f32(raw), _f32(coords), _i32(numsteps_c)
    grad_rgb, rgb_final = _f32(grad_rgb), _f32(rgb_final)
    mean = _f32(np.atleast_1d(density_grid_mean))
    n = numsteps_c.shape[0]
    out = np.zeros_like(raw)
    if backend == 'port':
        port().xo_calc_rgb_backward(_p(raw), _p(numsteps_c), _p(coords), _p(grad_rgb), _p(rgb_final), _p(mean),
                                    C.c_int(n), C.c_int(rgb_act), C.c_int(density_act), _p(out))
    else:
        ref().ref_calc_rgb_backward(_p(raw), _p(numsteps_c), _p(coords), _p(grad_rgb), _p(rgb_final), _p(mean),
                                    C.c_int(n), C.c_int(raw.shape[0]), C.c_int(rgb_act), C.c_int(density_act),
                                    C.c_float(aabb[0]), C.c_float(aabb[1]), _p(out))
    return out


def calc_rgb_inference(raw, coords, numsteps, bg3, rgb_act=2, density_act=3, backend='port', aabb=(0.0, 1.0)):
    raw, coords, numsteps, bg3 = _f32(raw), _f32(coords), _i32(numsteps), _f32(bg3)
    n = numsteps.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    alpha = np.zeros((n, 1), np.float32)
    if backend == 'port':
        port().xo_calc_rgb_inference(_p(raw), _p(coords), _p(numsteps), _p(bg3), C.c_int(n), C.c_int(rgb_act),
                                     C.c_int(density_act), _p(rgb), _p(alpha))
    else:
        ref().ref_calc_rgb_inference(_p(raw), _p(coords), _p(numsteps), _p(bg3), C.c_int(n),
                                     C.c_int(raw.shape[0]), C.c_int(rgb_act), C.c_int(density_act),
                                     C.c_float(aabb[0]), C.c_float(aabb[1]), _p(rgb), _p(alpha))
    return rgb, alpha


# ------------------------------------------------------------------ K6..K11
def generate_grid_samples(grid, ema_step, n_elements, max_cascade, thresh, aabb=(0.0, 1.0), rng_calls=0,
                          backend='port'):
    grid = _f32(grid)
    pos = np.zeros((n_elements, 3), np.float32)
    idx = np.zeros(n_elements, np.int32)
    if backend == 'port':
        st, inc = pcg32_host_state(rng_calls)
        port().xo_generate_grid_samples(_p(grid), C.c_uint32(ema_step), C.c_uint32(n_elements),
                                        C.c_uint32(max_cascade + 1), C.c_float(thresh), C.c_float(aabb[0]),
                                        C.c_float(aabb[1]), C.c_uint64(st), C.c_uint64(inc), _p(pos), _p(idx))
    else:
        ref().ref_generate_grid_samples(_p(grid), C.c_int(ema_step), C.c_int(n_elements), C.c_int(max_cascade),
                                        C.c_float(thresh), C.c_float(aabb[0]), C.c_float(aabb[1]),
                                        C.c_uint64(rng_calls), _p(pos), _p(idx))
    return pos, idx


def mark_untrained(focal, xforms, n_elements, res, backend='port'):
    focal, xforms = _f32(focal), _f32(xforms)
    grid = np.zeros(n_elements, np.float32)
    n_img = xforms.shape[0]
    fn = port().xo_mark_untrained if backend == 'port' else ref().ref_mark_untrained
    if backend == 'port':
        fn(_p(focal), _p(xforms), C.c_uint32(n_elements), C.c_int(n_img), C.c_int(res[0]), C.c_int(res[1]), _p(grid))
    else:
        fn(_p(focal), _p(xforms), C.c_int(n_elements), C.c_int(n_img), C.c_int(res[0]), C.c_int(res[1]), _p(grid))
    return grid


def splat(mlp_out, indices, grid_tmp, padded_width=1, backend='port'):
    mlp_out, indices = _f32(mlp_out), _i32(indices)
    grid_tmp = _f32(grid_tmp).copy()
    n = indices.shape[0]
    if backend == 'port':
        port().xo_splat(_p(mlp_out), _p(indices), C.c_int(padded_width), C.c_uint32(n), _p(grid_tmp))
    else:
        ref().ref_splat(_p(mlp_out), _p(indices), C.c_int(padded_width), C.c_int(n), _p(grid_tmp))
    return grid_tmp


def ema(grid_tmp, grid, decay=0.95, backend='port'):
    grid_tmp = _f32(grid_tmp)
    grid = _f32(grid).copy()
    n = grid.shape[0]
    if backend == 'port':
        port().xo_ema(_p(grid_tmp), C.c_uint32(n), C.c_float(decay), _p(grid))
    else:
        ref().ref_ema(_p(grid_tmp), C.c_int(n), C.c_float(decay), _p(grid))
    return grid


def density_mean(grid):
    grid = _f32(grid)
    return np.float32(port().xo_density_mean(_p(grid)))


def bitfield_given_mean(grid, mean, backend='port'):
    grid = _f32(grid)
    bf = np.zeros(G3, np.uint8)
    if backend == 'port':
        port().xo_bitfield(_p(grid), C.c_float(mean), _p(bf))
    else:
        m = np.array([mean], np.float32)
        ref().ref_bitfield_given_mean(_p(grid), _p(m), _p(bf))
    return bf


def update_bitfield_ref(grid):
    """reference update_bitfield_api end to end (CPU-shim serial mean)."""
    grid = _f32(grid)
    mean = np.zeros(16384, np.float32)
    bf = np.zeros(G3, np.uint8)
    ref().ref_update_bitfield(_p(grid), _p(mean), _p(bf))
    return mean, bf


# ------------------------------------------------------------------ tcnn half (parity unpinned)
class GridMeta:
    def __init__(self, n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=None):
        if per_level_scale is None:
            per_level_scale = float(np.exp2(np.log2(2048 * 1 / 16) / (16 - 1)))
        self.n_levels = n_levels
        self.scale = np.zeros(n_levels, np.float32)
        self.resolution = np.zeros(n_levels, np.uint32)
        self.offset = np.zeros(n_levels + 1, np.uint32)
        port().xo_hashgrid_meta(C.c_int(n_levels), C.c_int(log2_hashmap_size), C.c_int(base_resolution),
                                C.c_double(per_level_scale), _p(self.scale), _p(self.resolution), _p(self.offset))
        self.n_params = int(self.offset[-1]) * 2


def hashgrid_fwd(params, x, meta):
    params, x = _f32(params), _f32(x)
    n = x.shape[0]
    y = np.zeros((n, 2 * meta.n_levels), np.float32)
    port().xo_hashgrid_fwd(_p(params), _p(x), C.c_int(n), C.c_int(meta.n_levels), _p(meta.scale),
                           _p(meta.resolution), _p(meta.offset), _p(y))
    return y


def hashgrid_bwd(x, dy, meta):
    x, dy = _f32(x), _f32(dy)
    g = np.zeros(meta.n_params, np.float32)
    port().xo_hashgrid_bwd(_p(x), _p(dy), C.c_int(x.shape[0]), C.c_int(meta.n_levels), _p(meta.scale),
                           _p(meta.resolution), _p(meta.offset), _p(g))
    return g


def sh4(dirs):
    dirs = _f32(dirs)
    out = np.zeros((dirs.shape[0], 16), np.float32)
    port().xo_sh4(_p(dirs), C.c_int(dirs.shape[0]), _p(out))
    return out


def mlp_fwd(W, x, in_pad, width, n_hidden, out_pad, want_acts=False):
    W, x = _f32(W), _f32(x)
    n = x.shape[0]
    y = np.zeros((n, out_pad), np.float32)
    acts = np.zeros((n_hidden, n, width), np.float32) if want_acts else None
    port().xo_mlp_fwd(_p(W), _p(x), C.c_int(n), C.c_int(in_pad), C.c_int(width), C.c_int(n_hidden),
                      C.c_int(out_pad), _p(y), _p(acts) if want_acts else None)
    return (y, acts) if want_acts else y


def mlp_bwd(W, x, acts, dy, in_pad, width, n_hidden, out_pad):
    W, x, acts, dy = _f32(W), _f32(x), _f32(acts), _f32(dy)
    n = x.shape[0]
    dW = np.zeros_like(W)
    dx = np.zeros((n, in_pad), np.float32)
    port().xo_mlp_bwd(_p(W), _p(x), _p(acts), _p(dy), C.c_int(n), C.c_int(in_pad), C.c_int(width),
                      C.c_int(n_hidden), C.c_int(out_pad), _p(dW), _p(dx))
    return dW, dx


def nerf_mlp_fwd(table, Wd, Wc, pts, dirs, meta, n_hidden_d=1, n_hidden_c=2, pad_value=1.0):
    table, Wd, Wc, pts = _f32(table), _f32(Wd), _f32(Wc), _f32(pts)
    n = pts.shape[0]
    raw = np.zeros((n, 4), np.float32)
    dp = None
    if dirs is not None:
        dirs = _f32(dirs)
        dp = _p(dirs)
    port().xo_nerf_mlp_fwd(_p(table), _p(Wd), _p(Wc), _p(pts), dp, C.c_int(n), C.c_int(meta.n_levels),
                           _p(meta.scale), _p(meta.resolution), _p(meta.offset), C.c_int(n_hidden_d),
                           C.c_int(n_hidden_c), C.c_float(pad_value), _p(raw))
    return raw


def nerf_mlp_kink_margin(table, Wd, Wc, pts, dirs, meta, n_hidden_d=1, n_hidden_c=2, pad_value=1.0):
    """per sample: the smallest |pre-activation| / sum |w x| over the hidden units of both networks (xo_nerf_mlp_kink_margin)"""
    table, Wd, Wc, pts, dirs = _f32(table), _f32(Wd), _f32(Wc), _f32(pts), _f32(dirs)
    n = pts.shape[0]
    margin = np.zeros((n,), np.float32)
    port().xo_nerf_mlp_kink_margin(_p(table), _p(Wd), _p(Wc), _p(pts), _p(dirs), C.c_int(n), C.c_int(meta.n_levels),
                                   _p(meta.scale), _p(meta.resolution), _p(meta.offset), C.c_int(n_hidden_d),
                                   C.c_int(n_hidden_c), C.c_float(pad_value), _p(margin))
    return margin


def nerf_mlp_bwd(table, Wd, Wc, pts, dirs, draw, meta, n_hidden_d=1, n_hidden_c=2, pad_value=1.0):
    table, Wd, Wc, pts, dirs, draw = _f32(table), _f32(Wd), _f32(Wc), _f32(pts), _f32(dirs), _f32(draw)
    gt, gd, gc = np.zeros_like(table), np.zeros_like(Wd), np.zeros_like(Wc)
    port().xo_nerf_mlp_bwd(_p(table), _p(Wd), _p(Wc), _p(pts), _p(dirs), _p(draw), C.c_int(pts.shape[0]),
                           C.c_int(meta.n_levels), _p(meta.scale), _p(meta.resolution), _p(meta.offset),
                           C.c_int(n_hidden_d), C.c_int(n_hidden_c), C.c_float(pad_value), _p(gt), _p(gd), _p(gc))
    return gt, gd, gc


# ------------------------------------------------------------------ ray generation, loss, optimiser
def gen_rays(pose43, H, W, fx, fy, cx, cy, row0=0, nrows=None):
    pose43 = _f32(pose43)
    nrows = H - row0 if nrows is None else nrows
    o = np.zeros((nrows * W, 3), np.float32)
    d = np.zeros((nrows * W, 3), np.float32)
    port().xo_gen_rays(_p(pose43), C.c_int(H), C.c_int(W), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                       C.c_float(cy), C.c_int(row0), C.c_int(nrows), _p(o), _p(d))
    return o, d


def huber_loss_grad(rgb, target, delta=0.1, scale=5.0):
    rgb, target = _f32(rgb), _f32(target)
    g = np.zeros_like(rgb)
    loss = port().xo_huber_loss_grad(_p(rgb), _p(target), C.c_int(rgb.size), C.c_float(delta), C.c_float(scale), _p(g))
    return np.float32(loss), g


def adam(p, g, m, v, step, lr=1e-2, b1=0.9, b2=0.99, eps=1e-15, wd=1e-6):
    """in place on p, m, v (float32 contiguous)."""
    assert p.dtype == np.float32 and p.flags.c_contiguous
    port().xo_adam(_p(p), _p(_f32(g)), _p(m), _p(v), C.c_size_t(p.size), C.c_int(step), C.c_float(lr),
                   C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(wd))


def set_threads(n):
    port().xo_set_threads(C.c_int(n))
