"""GPU parity (through the C-ABI) of K1..K11 against the CPU oracle on identical inputs.
Bar: bit-exact sample indices / counts / positions; fp32 compositor within 1e-4 abs (north_star)."""
import numpy as np
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def edge_rays():
    """rays the reference's slab test / DDA treat specially: axis aligned (zero components ->
    inf/NaN in 1/d), missing the box, starting inside the box, grazing a face, degenerate."""
    o = [[0.5, 0.5, -1.0], [0.5, 0.5, -1.0], [-1.0, 0.5, 0.5], [0.5, 2.0, 0.5], [0.5, 0.5, 0.5], [0.5, 0.5, 0.5],
         [3.0, 3.0, 3.0], [0.0, 0.5, -1.0], [1.0, 1.0, -1.0], [0.45, 0.55, -2.0], [0.5, 0.5, -1.0], [0.31, 0.4, 0.35]]
    d = [[0.0, 0.0, 1.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.577, 0.577, 0.577], [0.0, 1.0, 0.0],
         [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 1.0], [0.01, -0.01, 0.9999], [1e-9, 1e-9, 1.0], [-0.6, 0.0, 0.8]]
    return np.array(o, np.float32), np.array(d, np.float32)


def run_k1(ops, dev, o, d, bf, max_samples, calls, wide=False):
    c, ri, ns, cnt = ops.rays_sampler(T(o, dev), T(d, dev), T(bf, dev), (0.0, 1.0), 0.05, 1.0 / 256, max_samples, calls, wide=wide)
    torch.cuda.synchronize()
    return c.cpu().numpy(), ri.cpu().numpy(), ns.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize('n_rays,calls', [(4096, 0), (4096, 1), (1000, 3), (70000, 0)])
def test_k1_bit_exact(O, lego, dev, n_rays, calls):
    from xrnerf_amd import ops, synthetic as S
    o, d, _ = S.training_rays(lego['poses'], n_rays, seed=11 + calls)
    eo, ed = edge_rays()
    o, d = np.concatenate([eo, o]), np.concatenate([ed, d])
    n = o.shape[0]
    rc, ri, rn, rcnt = O.rays_sampler(o, d, lego['bitfield'], rng_calls=calls, max_samples=n * 64)
    # wide: XR_K1_WIDE, the count pass with 8 lanes per ray (launches of up to 32 768 rays; the 70 000-ray case stays on one ray per lane)
    for wide in (False, True):
        gc, gi, gn, gcnt = run_k1(ops, dev, o, d, lego['bitfield'], n * 64, calls, wide=wide)
        assert np.array_equal(gcnt, rcnt), (gcnt, rcnt, wide)
        assert np.array_equal(gn, rn), wide
        assert np.array_equal(gi, ri), wide
        S_ = int(rcnt[1])
        assert S_ > n   # the scene is actually hit
        assert np.array_equal(bits(gc[:S_]), bits(rc[:S_])), wide


def test_k1_multi_cascade_bit_exact(O, dev):
    """aabb_scale = 16, five active cascades (the reference-generated fixture of tests/golden/make_golden_cascades.py)"""
    import os, sys
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, G)
    from make_golden_cascades import cascade_inputs
    from xrnerf_amd import ops
    g = np.load(os.path.join(G, 'ref_raymarch_cascades.npz'))
    grid, o, d, aabb = cascade_inputs()
    bf = O.bitfield_given_mean(grid, np.float32(0.5))
    for wide in (False, True):
        c, ri, ns, cnt = ops.rays_sampler(T(o, dev), T(d, dev), T(bf, dev), aabb, 0.05, 1 / 256, o.shape[0] * 1024, 0, wide=wide)
        assert np.array_equal(cnt.cpu().numpy(), g['counter']), wide
        assert np.array_equal(ns.cpu().numpy(), g['numsteps']) and np.array_equal(ri.cpu().numpy(), g['index']), wide
        s = int(cnt[1])
        assert np.array_equal(bits(c[:s].cpu().numpy()), bits(g['coords'])), wide


def test_k1_overflow_and_k2_clip(O, lego, dev):
    """max_samples smaller than the demand: overflowing rays get (0, base) and no index (ray_sampler.cu:76-82);
    K2 clips at max_compacted (compacted_coord.cu:63-64)."""
    from xrnerf_amd import ops, synthetic as S
    o, d, _ = S.training_rays(lego['poses'], 3000, seed=5)
    full = O.rays_sampler(o, d, lego['bitfield'])
    total = int(full[3][1])
    for cap in (total // 2, total - 1, total, 7):
        rc, ri, rn, rcnt = O.rays_sampler(o, d, lego['bitfield'], max_samples=cap)
        gc, gi, gn, gcnt = run_k1(ops, dev, o, d, lego['bitfield'], cap, 0)
        assert np.array_equal(gcnt, rcnt) and np.array_equal(gn, rn) and np.array_equal(gi, ri)
        valid = rn[:, 0] > 0
        for i in np.nonzero(valid)[0][:200]:
            b, k = rn[i, 1], rn[i, 0]
            assert np.array_equal(bits(gc[b:b + k]), bits(rc[b:b + k]))
    # K2 on the un-truncated march
    rc, ri, rn, rcnt = full
    for cap in (1 << 18, total // 3, 5):
        oc, onc, orc, osc = O.compacted_coord(rc[:total], rn, cap)
        gc, gnc, grc, gsc = ops.compacted_coord(T(rc[:total], dev), T(rn, dev), cap)
        torch.cuda.synchronize()
        assert np.array_equal(gnc.cpu().numpy(), onc)
        assert int(grc.item()) == int(orc[0]) and int(gsc.item()) == int(osc[0])
        kept = min(cap, total)
        assert np.array_equal(bits(gc.cpu().numpy()[:kept]), bits(oc[:kept]))


def test_window_march_equals_one_launch_per_iteration(lego, dev):
    """xr_ngp_window_march: the batches and marches of several iterations as ONE series of launches (blockIdx.y = iteration) are bit for
    bit the batches, samples, counters and clipped counts of one xr_make_batch_series(1) / xr_rays_sampler / xr_clip_numsteps sequence per
    iteration with consecutive RNG call indices -- a batch size that is not a multiple of the 256-ray block, the table cursor wrapping
    inside the series, a launch whose samples overflow its buffer, a series that starts in the middle of the window."""
    from xrnerf_amd import ops, synthetic as S
    n, n_table = 300, 1000
    o, d, _ = S.training_rays(lego['poses'], n_table, seed=3)
    rng = np.random.default_rng(0)
    table = np.concatenate([o, d, rng.uniform(0, 1, (n_table, 4)), rng.integers(0, 20, (n_table, 1))], 1).astype(np.float32)
    tt, bf = T(table, dev), T(lego['bitfield'], dev)
    planes = dev.type == 'cuda'
    for max_samples, first, chunks, ready in ((n * 64, 3, 5, 0), (2500, 0, 4, 0), (n * 64, 11, 5, 1)):
        win = ops.MarchWindow(dev, 384, max(max_samples, 4096), planes=planes)
        cur0, b0, k0, clip = 450, 7, 21, 2000
        if ready:          # the first chunk's batch is already in place (drawn by the caller)
            ops.make_batch(tt[cur0:cur0 + n], n, b0, out=win.batch_out(first))
            end = ops.ngp_window_march(win, first, chunks, 1, n, tt, cur0 + n, b0 + 1, bf, (0.0, 1.0), 0.05, 1.0 / 256, max_samples, k0, clip)
        else:
            end = ops.ngp_window_march(win, first, chunks, 0, n, tt, cur0, b0, bf, (0.0, 1.0), 0.05, 1.0 / 256, max_samples, k0, clip)
        torch.cuda.synchronize()
        cur, overflowed = cur0, 0
        for j in range(chunks):
            if cur + n > n_table:
                cur = 0
            b = ops.make_batch(tt[cur:cur + n], n, b0 + j)
            cur += n
            c, ri, ns, cnt = ops.rays_sampler(b['rays_o'], b['rays_d'], bf, (0.0, 1.0), 0.05, 1.0 / 256, max_samples, k0 + j)
            cl, nv = ops.clip_numsteps(ns, cnt, clip)
            torch.cuda.synchronize()
            w = win.batch(first + j, n)
            for key in b:
                assert torch.equal(w[key], b[key]), (key, j)
            s = min(int(cnt[1]), max_samples)
            overflowed += int(cnt[1]) > max_samples
            assert torch.equal(win.counter2[first + j], cnt) and torch.equal(win.numsteps[first + j, :n], ns), j
            assert torch.equal(win.rays_index[first + j, :n], ri) and torch.equal(win.clipped[first + j, :n], cl), j
            assert torch.equal(win.n_valid[first + j], nv), j
            assert s > 0
            # rows of rays that were dropped for overflow are not written by either form: compare the rows the counts name
            keep = ns[:, 0] > 0
            for i in torch.nonzero(keep)[:, 0].tolist()[:400]:
                b_, k_ = int(ns[i, 1]), int(ns[i, 0])
                assert torch.equal(win.coords[first + j, b_:b_ + k_].view(torch.int32), c[b_:b_ + k_].view(torch.int32)), (j, i)
                if win.xyz is not None:
                    assert torch.equal(win.xyz[first + j][:, b_:b_ + k_].t().contiguous().view(torch.int32), c[b_:b_ + k_, :3].contiguous().view(torch.int32))
            if win.pinned is not None:
                assert win.pinned[first + j].tolist() == cnt.tolist()
        assert end == cur
        assert (overflowed > 0) == (max_samples == 2500)


def make_samples(O, lego, n_rays, seed):
    from xrnerf_amd import synthetic as S
    o, d, _ = S.training_rays(lego['poses'], n_rays, seed=seed)
    c, ri, ns, cnt = O.rays_sampler(o, d, lego['bitfield'])
    total = int(cnt[1])
    rng = np.random.default_rng(seed)
    raw = rng.normal(0, 1.5, (total, 4)).astype(np.float32)
    raw[:, 3] = rng.normal(2.0, 3.0, total)   # exp density from ~0 to large
    return c[:total].copy(), ns, raw, rng


@pytest.mark.parametrize('rgb_act,density_act', [(2, 3), (3, 3), (1, 1), (0, 2)])
def test_compositor_fwd_bwd_inference(O, lego, dev, rgb_act, density_act):
    from xrnerf_amd import ops
    coords, ns, raw, rng = make_samples(O, lego, 3000, 21)
    n = ns.shape[0]
    if density_act != 3:
        raw[:, 3] = np.abs(raw[:, 3])
    bg = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    # a clipped "compacted" view: half of the rays lose their tail (then no bg term, calc_rgb.cu:61-64)
    nsc = ns.copy()
    cut = rng.uniform(0, 1, n) < 0.3
    nsc[cut, 0] = nsc[cut, 0] // 2
    ref = O.calc_rgb_forward(raw, coords, ns, nsc, bg, rgb_act, density_act)
    got = ops.calc_rgb_forward(T(raw, dev), T(coords, dev), T(ns, dev), T(nsc, dev), T(bg, dev), rgb_act, density_act)
    tol = 1e-4 * max(1.0, np.abs(ref).max())   # 1e-4 abs for the config's logistic rgb; relative for exp (values ~1e4)
    assert np.abs(got.cpu().numpy() - ref).max() <= tol
    # backward
    grad = rng.normal(0, 1, (n, 3)).astype(np.float32)
    for mean in (0.001, 0.5):   # toggles the L1 density regulariser (calc_rgb.cu:104)
        rb = O.calc_rgb_backward(raw, nsc, coords, grad, ref, mean, rgb_act, density_act)
        gb = ops.calc_rgb_backward(T(raw, dev), T(nsc, dev), T(coords, dev), T(grad, dev), T(ref, dev),
                                   T(np.array([mean], np.float32), dev), rgb_act, density_act).cpu().numpy()
        err = np.abs(gb - rb).max()
        # exp rgb activation: gradients scale with exp(raw) ~ 1e4 -> relative bound
        assert err <= (1e-4 if rgb_act != 3 else 2e-4) * max(1.0, np.abs(rb).max()), err
    # inference
    rr, ra = O.calc_rgb_inference(raw, coords, ns, [0.2, 0.5, 0.9], rgb_act, density_act)
    gr, ga = ops.calc_rgb_inference(T(raw, dev), T(coords, dev), T(ns, dev), [0.2, 0.5, 0.9], rgb_act, density_act)
    assert np.abs(gr.cpu().numpy() - rr).max() <= 1e-4 * max(1.0, np.abs(rr).max()) and np.abs(ga.cpu().numpy() - ra).max() <= 1e-4


def test_compositor_zero_sample_rays(O, dev):
    from xrnerf_amd import ops
    ns = np.zeros((130, 2), np.int32)
    raw = np.zeros((4, 4), np.float32); coords = np.zeros((4, 7), np.float32)
    bg = np.random.default_rng(0).uniform(0, 1, (130, 3)).astype(np.float32)
    got = ops.calc_rgb_forward(T(raw, dev), T(coords, dev), T(ns, dev), T(ns, dev), T(bg, dev), 2, 3).cpu().numpy()
    assert np.array_equal(got, bg)
    r, a = ops.calc_rgb_inference(T(raw, dev), T(coords, dev), T(ns, dev), [1, 0, 1], 2, 3)
    assert np.array_equal(r.cpu().numpy(), np.tile(np.array([1, 0, 1], np.float32), (130, 1))) and float(a.abs().max()) == 0


def test_k6_grid_samples_bit_exact(O, lego, dev):
    from xrnerf_amd import ops
    rng = np.random.default_rng(3)
    grid = (lego['grid'] * rng.uniform(0.0, 0.05, lego['grid'].shape)).astype(np.float32)
    grid[rng.uniform(0, 1, grid.shape) < 0.1] = -1.0
    g = T(grid, dev)
    for (n, step, casc, thr, calls, aabb) in [(100000, 0, 1, -0.01, 0, (0.0, 1.0)), (100000, 7, 1, 0.01, 1, (0.0, 1.0)),
                                              (65536, 3, 5, 0.01, 4, (0.0, 1.0)), (65536, 3, 5, 0.01, 4, (-7.5, 8.5))]:
        rp, ri = O.generate_grid_samples(grid, step, n, casc - 1, thr, aabb=aabb, rng_calls=calls)
        gp, gi = ops.generate_grid_samples(g, step, n, casc, thr, aabb, calls)
        assert np.array_equal(gi.cpu().numpy(), ri)
        assert np.array_equal(bits(gp.cpu().numpy()), bits(rp))


def test_k7_mark_untrained(O, lego, dev):
    from xrnerf_amd import ops, synthetic as S
    poses = lego['poses'][:7]
    focal = np.full((7, 2), S.LEGO_FOCAL, np.float32)
    n = 2 * 128 ** 3
    ref = O.mark_untrained(focal, poses, n, (800, 800))
    got = ops.mark_untrained_density_grid(T(focal, dev), T(poses, dev), n, (800, 800)).cpu().numpy()
    assert np.array_equal(got, ref)
    assert 0 < (ref < 0).sum() < n


def test_k8_k9_k10_k11(O, lego, dev):
    from xrnerf_amd import ops
    rng = np.random.default_rng(8)
    n = 200000
    idx = rng.integers(0, 128 ** 3, n).astype(np.int32)
    idx[:1000] = idx[0]   # heavy collisions on one cell
    mlp = rng.normal(0, 2, (n, 1)).astype(np.float32)
    tmp0 = np.zeros(8 * 128 ** 3, np.float32)
    ref_tmp = O.splat(mlp, idx, tmp0)
    got_tmp = ops.splat_grid_samples(T(mlp, dev), T(idx, dev), 1, n, T(tmp0, dev)).cpu().numpy()
    assert np.allclose(got_tmp, ref_tmp, rtol=2e-6, atol=0)
    grid = (lego['grid'] * 0.02).astype(np.float32)
    grid[rng.uniform(0, 1, grid.shape) < 0.05] = -1.0
    ref_grid = O.ema(ref_tmp, grid)
    got_grid = ops.ema_grid_samples(T(ref_tmp, dev), grid.size, 0.95, T(grid, dev)).cpu().numpy()
    assert np.array_equal(bits(got_grid), bits(ref_grid))
    # K10 mean (order of summation differs -> relative tolerance) and K11 given the SAME mean: bit exact
    mean = torch.zeros(16384, dtype=torch.float32, device=dev)
    bf = torch.zeros(128 ** 3, dtype=torch.uint8, device=dev)
    ops.update_bitfield(T(ref_grid, dev), mean, bf)
    # the reference's own value depends on its float-atomic order; the CPU restatement sums 2M terms
    # serially in fp32 (error ~1e-3 rel), the HIP tree is compared with the exact (fp64) mean
    exact = np.maximum(ref_grid[:128 ** 3].astype(np.float64), 0).sum() / 128 ** 3
    assert abs(float(mean[0]) - exact) <= 1e-5 * exact
    assert abs(float(mean[0]) - O.density_mean(ref_grid)) <= 5e-3 * exact
    for m in (float(mean[0]), 0.5, 1e-5):
        mt = torch.tensor([m], dtype=torch.float32, device=dev)
        got_bf = ops.bitfield_from_mean(T(ref_grid, dev), mt, torch.zeros_like(bf)).cpu().numpy()
        assert np.array_equal(got_bf, O.bitfield_given_mean(ref_grid, np.float32(m)))
    # the end-to-end call agrees with K11 at its own mean, and is reproducible
    assert np.array_equal(bf.cpu().numpy(), O.bitfield_given_mean(ref_grid, np.float32(float(mean[0]))))
    mean2 = torch.zeros_like(mean); bf2 = torch.zeros_like(bf)
    ops.update_bitfield(T(ref_grid, dev), mean2, bf2)
    assert float(mean2[0]) == float(mean[0]) and torch.equal(bf, bf2)


@pytest.mark.parametrize('n_casc', [1, 8, 3])
def test_refresh_tail_in_three_launches(O, lego, dev, n_casc):
    """xr_ema_update_bitfield (K9 + K10's partial sums in one pass, the bits kernel folding the partials itself and emitting every
    cascade's own pool bytes from a wave ballot, the seven dependent max-pool launches as one chain kernel) == xr_ema_grid_samples
    followed by xr_update_bitfield: grid, mean and bitfield bit for bit -- and the oracle's K9 / K11 on the same inputs.  Densities in
    every cascade, so that what is pooled into a cascade from below meets bits of its own."""
    from xrnerf_amd import ops
    rng = np.random.default_rng(90 + n_casc)
    cells = 128 ** 3
    grid = (lego['grid'] * 0.02).astype(np.float32)
    grid[rng.uniform(0, 1, grid.shape) < 0.05] = -1.0
    for c in range(1, 8):                                   # sparse own bits in the outer cascades, some inside the pooled centre
        k = rng.integers(0, cells, 3000)
        grid[c * cells + k] = rng.uniform(0.0, 0.05, k.size).astype(np.float32)
    n_used = n_casc * cells
    tmp = np.zeros_like(grid)
    k = rng.integers(0, n_used, 400000)
    tmp[k] = rng.uniform(0, 0.2, k.size).astype(np.float32)
    # the two entry points of the reference's order
    g_a = T(grid.copy(), dev)                               # (copies: on the host build T() aliases the numpy array)
    ops.ema_grid_samples(T(tmp.copy(), dev), n_used, 0.95, g_a)
    mean_a = torch.zeros(16384, dtype=torch.float32, device=dev); bf_a = torch.full((cells,), 0x55, dtype=torch.uint8, device=dev)
    ops.update_bitfield(g_a, mean_a, bf_a)
    # one entry point, three launches
    g_b = T(grid.copy(), dev)
    mean_b = torch.zeros_like(mean_a); bf_b = torch.full((cells,), 0xaa, dtype=torch.uint8, device=dev)
    ops.ema_update_bitfield(T(tmp.copy(), dev), n_used, 0.95, g_b, mean_b, bf_b)
    assert np.array_equal(bits(g_b.cpu().numpy()), bits(g_a.cpu().numpy()))
    assert float(mean_b[0]) == float(mean_a[0])
    assert torch.equal(bf_b, bf_a)
    # against the oracle: K9 on the cells in use, K11 at the same mean
    ref_grid = grid.copy()
    ref_grid[:n_used] = O.ema(tmp[:n_used].copy(), grid[:n_used].copy())
    assert np.array_equal(bits(g_b.cpu().numpy()), bits(ref_grid))
    exact = np.maximum(ref_grid[:cells].astype(np.float64), 0).sum() / cells
    assert abs(float(mean_b[0]) - exact) <= 1e-5 * exact
    ref_bf = O.bitfield_given_mean(ref_grid, np.float32(float(mean_b[0])))
    assert np.array_equal(bf_b.cpu().numpy(), ref_bf)
    assert ref_bf[cells // 8:].any() and ref_bf[7 * cells // 8:].any()          # something did reach the outermost cascade


def test_gen_rays_huber_adam(O, lego, dev):
    from xrnerf_amd import ops, synthetic as S
    pose = lego['poses'][3]
    f = np.float32(S.LEGO_FOCAL)
    ro, rd = O.gen_rays(pose, 800, 800, f, f, 400.0, 400.0, row0=100, nrows=50)
    go, gd = ops.gen_rays(pose, 800, 800, float(f), float(f), 400.0, 400.0, row0=100, nrows=50, device=dev)
    assert np.array_equal(bits(go.cpu().numpy()), bits(ro))
    assert np.abs(gd.cpu().numpy() - rd).max() <= 2.4e-7   # <= 2 ulp of a unit vector component
    rng = np.random.default_rng(1)
    rgb = rng.uniform(0, 1, (5000, 3)).astype(np.float32); tgt = rng.uniform(0, 1, (5000, 3)).astype(np.float32)
    rl, rg = O.huber_loss_grad(rgb, tgt)
    gl, gg = ops.huber_loss_grad(T(rgb, dev), T(tgt, dev))
    assert abs(float(gl) - rl) <= 1e-5 * rl and np.abs(gg.cpu().numpy() - rg).max() <= 1e-6
    n = 100003
    p = rng.normal(0, 1, n).astype(np.float32); g = rng.normal(0, 1e-2, n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    tp, tm, tv = T(p, dev), T(m, dev), T(v, dev)
    for step in (1, 2, 3):
        O.adam(p, g, m, v, step)
        ops.adam_step(tp, T(g, dev), tm, tv, step)
    assert np.abs(tp.cpu().numpy() - p).max() <= 1e-5 and np.abs(tv.cpu().numpy() - v).max() <= 1e-9


def test_adam_multi_grad_scale_equals_scaling_pass_then_adam(O, dev):
    """xr_adam_step_multi(grad_scale = 1/world) -- what the data-parallel trainer runs on the all-reduced SUM of the gradients --
    against xr_scale_multi(1/world) followed by the same update: parameters, moments and EMA bit for bit, gradients untouched;
    and against the oracle's Adam on the averaged gradients"""
    from xrnerf_amd import ops
    rng = np.random.default_rng(5)
    sizes = (100003, 3072, 7168)
    mk = lambda sc: [rng.normal(0, sc, n).astype(np.float32) for n in sizes]
    p0, g0 = mk(1.0), mk(3e-2)
    for world in (2, 3, 8):
        fac = 1.0 / world
        # (.clone(): on the host-emulated device T() shares the numpy buffer)
        A = dict(p=[T(a, dev).clone() for a in p0], g=[T(a, dev).clone() for a in g0], m=[T(np.zeros_like(a), dev) for a in p0],
                 v=[T(np.zeros_like(a), dev) for a in p0], e=[T(a, dev).clone() for a in p0])
        B = {k: [t.clone() for t in v] for k, v in A.items()}
        ref = [(a.copy(), np.zeros_like(a), np.zeros_like(a)) for a in p0]
        for step in (1, 2, 3):
            ops.adam_step_multi(A['p'], A['g'], A['m'], A['v'], step, emas=A['e'], ema_momentum=0.05, grad_scale=fac)
            gs = [t.clone() for t in B['g']]
            ops.scale_multi(gs, None, fac)
            ops.adam_step_multi(B['p'], gs, B['m'], B['v'], step, emas=B['e'], ema_momentum=0.05)
            for (rp, rm, rv), g in zip(ref, g0):
                O.adam(rp, (g * np.float32(fac)).astype(np.float32), rm, rv, step)
        for k in 'pmve':
            for a, b in zip(A[k], B[k]):
                assert torch.equal(a, b), (world, k)
        for a, g in zip(A['g'], g0):
            assert np.array_equal(a.cpu().numpy(), g)
        for a, (rp, _, _) in zip(A['p'], ref):
            assert np.abs(a.cpu().numpy() - rp).max() <= 1e-5


@pytest.mark.parametrize('n_rays', [3000, 1001])
def test_fused_compositor_train_equals_k3_huber_k4(O, lego, dev, n_rays):
    """xr_composite_train (K3 + 5*Huber + masked MSE + K4 in one launch) against the three separate entry points on marched
    Lego samples incl. clipped tails and rays without samples: rgb and dL/draw bit for bit, the two loss scalars to 1e-6"""
    from xrnerf_amd import ops, synthetic as S
    rng = np.random.default_rng(21)
    o, d, _ = S.training_rays(lego['poses'], n_rays, seed=9)           # (1001: the last workgroups of both kernels are partly empty)
    coords, _, ns, cnt = ops.rays_sampler(T(o, dev), T(d, dev), T(lego['bitfield'], dev), (0., 1.), 0.05, 1 / 256, n_rays * 64, 0)
    total = int(cnt[1])
    cap = int(total * 0.8)                                            # clip: some rays lose their tail, some everything
    nsc, _ = ops.clip_numsteps(ns, cnt, cap)
    raw = T(rng.normal(0, 1.5, (cap, 4)).astype(np.float32), dev)
    bg = T(rng.uniform(0, 1, (n_rays, 3)).astype(np.float32), dev)
    tgt = T(rng.uniform(0, 1, (n_rays, 3)).astype(np.float32), dev)
    alpha = T((rng.uniform(0, 1, (n_rays, 1)) > 0.3).astype(np.float32), dev)
    mean = torch.tensor([0.005] + [0.0] * 15, dtype=torch.float32, device=dev)
    c = coords[:cap].contiguous()
    dead_seen = False
    for ra, da in ((2, 3), (3, 1)):
        rgb_a = ops.calc_rgb_forward(raw, c, ns, nsc, bg, ra, da)
        lm_a, grad = ops.huber_loss_grad_mse(rgb_a, tgt, alpha, 0.1, 5.0)
        draw_a = ops.calc_rgb_backward(raw, nsc, c, grad, rgb_a, mean, ra, da)
        lm_b = torch.zeros(2, dtype=torch.float32, device=dev)
        draw_b = torch.zeros_like(raw)
        rgb_b = ops.composite_train(raw, c, ns, nsc, bg, tgt, alpha, mean, ra, da, lm_b, draw_b)
        assert torch.equal(rgb_a, rgb_b)
        assert torch.equal(draw_a, draw_b)
        assert torch.allclose(lm_a, lm_b, rtol=1e-5, atol=0)
        # the same launch counting the live rows per 1024-row segment (live_seg_count) + the ranking pass alone
        # (seg_counts_ready) = the two-pass list: same rows, same count; some rows are dead (exact zeros behind T == 0)
        raw_dead = raw.clone()
        raw_dead[::3, 3] = 200.0                                     # opaque samples: everything behind them has T == 0
        draw_c, draw_d = torch.zeros_like(raw), torch.zeros_like(raw)
        lm_c = torch.zeros(2, dtype=torch.float32, device=dev)
        ops.composite_train(raw_dead, c, ns, nsc, bg, tgt, alpha, mean, ra, da, lm_c, draw_c)
        rows_2, n_2 = ops.live_rows(draw_c, cap)
        rows_2, n_2 = rows_2.clone(), n_2.clone()
        lm_c.zero_()
        seg = torch.zeros(ops.live_segments(cap), dtype=torch.int32, device=dev)
        ops.composite_train(raw_dead, c, ns, nsc, bg, tgt, alpha, mean, ra, da, lm_c, draw_d, live_seg=seg)
        rows_1, n_1 = ops.live_rows(draw_d, cap, seg_counts=seg)
        live = (draw_c != 0).any(1)
        assert seg.tolist() == [int(live[k:k + 1024].sum()) for k in range(0, cap, 1024)]
        assert torch.equal(draw_c, draw_d)
        nl = int(n_2[0])
        assert nl == int(n_1[0]) == int((draw_c != 0).any(1).sum()) and 0 < nl <= cap
        dead_seen = dead_seen or nl < cap
        assert torch.equal(rows_1[:nl], rows_2[:nl])
        # the wave-per-ray form (no loss accumulator handed in): another association of the same products; loss scalars from
        # xr_train_loss_scalars; its own live-row counts are exact for its own rows
        draw_w = torch.zeros_like(raw)
        seg_w = torch.zeros(ops.live_segments(cap), dtype=torch.int32, device=dev)
        rgb_w = ops.composite_train(raw, c, ns, nsc, bg, tgt, alpha, mean, ra, da, None, draw_w)
        assert float((rgb_w - rgb_a).abs().max()) <= 2e-6
        assert float((draw_w - draw_a).abs().max()) <= 2e-5 * float(draw_a.abs().max())
        lm_w = ops.train_loss_scalars(rgb_w, tgt, alpha, 0.1, 5.0)
        assert torch.allclose(lm_w, lm_a, rtol=1e-5, atol=0)
        # (opaque samples: the suffix colour behind them is a difference of nearly equal sums times exp(15) -- only the zero
        # pattern and the counts are compared on that input)
        draw_w.zero_()
        ops.composite_train(raw_dead, c, ns, nsc, bg, tgt, alpha, mean, ra, da, None, draw_w, live_seg=seg_w)
        live_w = (draw_w != 0).any(1)
        assert seg_w.tolist() == [int(live_w[k:k + 1024].sum()) for k in range(0, cap, 1024)]
        assert not bool((live_w & ~live).any())   # exact zeros (T == 0 behind an opaque sample) of the 16-lane form are zeros here too
    assert int((nsc[:, 0] == 0).sum()) > 0 and int((nsc[:, 0] < ns[:, 0]).sum()) > 0
    assert int(nsc[:, 0].max()) > 64                                  # chunks longer than the in-register fast path
    assert dead_seen
