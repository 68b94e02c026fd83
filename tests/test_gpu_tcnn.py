"""GPU parity of the tiny-cuda-nn half (hash grid, SH-4, fused MLP fwd/bwd) against the fp32 CPU
restatement.  PARITY UNPINNED vs the reference (tcnn is not in its tree): these pin the HIP kernels
to OUR statement of the published algorithm.  Tolerance: 1e-4 abs fp32 on raw (north_star)."""
import numpy as np
import pytest
import torch

from conftest import grad_close

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_grid_meta_matches_oracle(O, dev):
    from xrnerf_amd import ops
    a, b = ops.GridMeta(), O.GridMeta()
    assert np.array_equal(a.scale, b.scale) and np.array_equal(a.resolution, b.resolution)
    assert np.array_equal(a.offset, b.offset) and a.n_params == 12196240


@pytest.mark.parametrize('n', [1, 31, 4096, 50001])
def test_hashgrid_fwd_bwd(O, dev, n):
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(n)
    table = S.hash_table(meta.n_params, scale=1.0)
    x = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    x[0] = [0.0, 1.0, 0.5]   # corners of the domain (index wrap-around at res)
    ref = O.hashgrid_fwd(table, x, om)
    tt = T(table, dev)
    enc_t = ops.hashgrid_fwd(tt, T(x, dev), meta)
    got = enc_t[:, :n].t().cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-6
    # strided positions: the leading 3 columns of [n,7] coordinate rows
    rows = np.zeros((n, 7), np.float32); rows[:, :3] = x; rows[:, 3:] = 7.0
    enc2 = ops.hashgrid_fwd(tt, T(rows, dev)[:, :3], meta)
    assert torch.equal(enc2[:, :n], enc_t[:, :n])
    # backward
    dy = rng.normal(0, 1, (n, 32)).astype(np.float32)
    ref_g = O.hashgrid_bwd(x, dy, om)
    ld = enc_t.shape[1]
    dt = torch.zeros((32, ld), dtype=torch.float32, device=dev); dt[:, :n] = T(dy, dev).t()
    g = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
    ops.hashgrid_bwd(T(x, dev), dt.contiguous(), meta, g)
    err = np.abs(g.cpu().numpy() - ref_g).max()
    assert err <= 1e-4 * max(1.0, np.abs(ref_g).max()), err
    # the atomic scatter (no workspace) and the LDS-partition scan (n >= 16384, hashed levels) agree
    g1 = torch.zeros_like(g)
    ops.hashgrid_bwd(T(x, dev), dt.contiguous(), meta, g1, use_workspace=False)
    assert float((g1 - g).abs().max()) <= 1e-5 * max(1.0, float(g.abs().max()))
    assert np.abs(g1.cpu().numpy() - ref_g).max() <= 1e-4 * max(1.0, np.abs(ref_g).max())
    # level-range launches (data-parallel bucketing: fine half, then coarse half) write exactly their levels
    g2 = torch.zeros_like(g)
    ops.hashgrid_bwd(T(x, dev), dt.contiguous(), meta, g2, levels=(8, 16))
    cut = 2 * int(meta.offset[8])
    assert float(g2[:cut].abs().max()) == 0.0
    assert float((g2[cut:] - g[cut:]).abs().max()) <= 1e-5 * max(1.0, float(g.abs().max()))
    ops.hashgrid_bwd(T(x, dev), dt.contiguous(), meta, g2, levels=(0, 8))
    assert float((g2 - g).abs().max()) <= 1e-5 * max(1.0, float(g.abs().max()))


def test_sh4(O, dev):
    from xrnerf_amd import ops
    d = np.random.default_rng(0).uniform(0, 1, (1000, 3)).astype(np.float32)
    assert np.abs(ops.sh4(T(d, dev)).cpu().numpy() - O.sh4(d)).max() <= 1e-6


def nets(S):
    wd = S.mlp_weights(32, 64, 1, 16, seed=4)
    wc = S.mlp_weights(32, 64, 2, 16, seed=5)
    return wd, wc


@pytest.fixture(params=['f16x2', 'bf16x3', 'mfma'])
def f32_forward(request):
    """the forwards of the fp32 mode: f16x2 = 2-way fp16 operand split on the fp16 matrix cores (default), bf16x3 = exact 3-way bf16
    split, mfma = fp32 MFMA"""
    from xrnerf_amd import ops
    old = ops.f32_forward()
    ops.set_f32_forward(request.param)
    yield request.param
    ops.set_f32_forward(old)


@pytest.mark.parametrize('n', [1, 32, 33, 1000, 40000])
def test_nerf_mlp_fwd(O, dev, n, f32_forward):
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(n)
    table = S.hash_table(meta.n_params, scale=0.5)
    wd, wc = nets(S)
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    ref = O.nerf_mlp_fwd(table, wd, wc, pts, dirs, om)
    enc_t = ops.hashgrid_fwd(T(table, dev), T(pts, dev), meta)
    raw = ops.nerf_mlp_fwd(enc_t, T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2).cpu().numpy()
    err = np.abs(raw - ref).max()
    assert err <= 1e-4, err
    # density only (run_density)
    rd = ops.nerf_mlp_fwd(enc_t, None, n, T(wd, dev), None, 1, 2).cpu().numpy()
    assert np.abs(rd[:, 3] - ref[:, 3]).max() <= 1e-4


@pytest.mark.parametrize('n,n_valid', [(1, None), (33, None), (8191, 8000), (70001, None), (5000, 0)])
def test_nerf_mlp_fwd_split_operands_equal_fp32_mfma(dev, n, n_valid):
    """xr_nerf_mlp_fwd in the XR_MLP_BF16X3 and XR_MLP_F16X2 arithmetics against XR_MLP_F32 on the same inputs: activations spanning 1e-6 .. 1e2, weights of
    mixed magnitude, a device-side row count, row-indirect directions.  Both are fp32-accurate evaluations of the same sums,
    so they agree to a few ulp of the largest partial sum (here: 3e-6 of max|raw|), far inside the 1e-4 parity bar."""
    from xrnerf_amd import ops, synthetic as S
    rng = np.random.default_rng(7 * n + 1)
    wd, wc = nets(S)
    wd = (wd * rng.choice([0.01, 1.0, 4.0], wd.shape)).astype(np.float32)
    ld = (n + 63) // 64 * 64
    enc = np.zeros((32, ld), np.float32)
    enc[:, :n] = (rng.normal(0, 1, (32, n)) * np.exp(rng.uniform(np.log(1e-6), np.log(1e1), (1, n)))).astype(np.float32)
    dirs = rng.uniform(0, 1, (n + 7, 3)).astype(np.float32)
    rows = rng.permutation(n + 7)[:n].astype(np.int32)
    n_dev = None if n_valid is None else torch.tensor([n_valid], dtype=torch.int32, device=dev)
    out = {}
    old = ops.f32_forward()
    try:
        for kind in ('mfma', 'bf16x3', 'f16x2'):
            ops.set_f32_forward(kind)
            raw = torch.full((n, 4), 7.0, dtype=torch.float32, device=dev)
            ops.nerf_mlp_fwd(T(enc, dev), T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2, raw=raw, n_dev=n_dev, rows=T(rows, dev))
            dens = torch.full((n, 4), 7.0, dtype=torch.float32, device=dev)
            ops.nerf_mlp_fwd(T(enc, dev), None, n, T(wd, dev), None, 1, 2, raw=dens, n_dev=n_dev)
            out[kind] = (raw.cpu().numpy(), dens.cpu().numpy())
    finally:
        ops.set_f32_forward(old)
    m = n if n_valid is None else n_valid
    for kind in ('bf16x3', 'f16x2'):
        for a, b in zip(out['mfma'], out[kind]):
            assert np.all(a[m:] == 7.0) and np.all(b[m:] == 7.0)                   # rows behind the device-side count untouched
            if m:
                scale = np.abs(a[:m]).max()
                assert np.isfinite(b[:m]).all() and np.abs(a[:m] - b[:m]).max() <= 3e-6 * scale, (kind, np.abs(a[:m] - b[:m]).max(), scale)
        if m:
            assert np.array_equal(out[kind][0][:m, 3], out[kind][1][:m, 3])   # sigma: same arithmetic with and without the color net


def _chain_oracle(O, wd, wc, enc, dirs, sat=False, draw=None):
    """the (1, 2) network over explicit encoded features through the oracle's layer functions (xo_mlp_fwd / xo_mlp_bwd); sat: with
    XR_MLP_F16X2's saturation applied to the operands (features x 16, colour inputs -- hidden activations are checked, not clamped)"""
    enc = np.ascontiguousarray(enc, np.float32)
    if sat:
        enc = np.clip(enc * 16.0, -65504.0, 65504.0).astype(np.float32) / np.float32(16.0)
    dout, actd = O.mlp_fwd(wd, enc, 32, 64, 1, 16, want_acts=True)
    cin = np.concatenate([dout[:, 1:16], O.sh4(dirs), np.ones((enc.shape[0], 1), np.float32)], 1)
    if sat:
        cin = np.clip(cin, -65504.0, 65504.0)
    cout, actc = O.mlp_fwd(wc, cin, 32, 64, 2, 16, want_acts=True)
    raw = np.concatenate([cout[:, :3], dout[:, :1]], 1)
    if draw is None:
        return raw, max(float(actd.max()), float(actc.max()))
    dyc = np.zeros((enc.shape[0], 16), np.float32); dyc[:, :3] = draw[:, :3]
    gwc, dcin = O.mlp_bwd(wc, cin, actc, dyc, 32, 64, 2, 16)
    dyd = np.zeros((enc.shape[0], 16), np.float32); dyd[:, 0] = draw[:, 3]; dyd[:, 1:16] = dcin[:, :15]
    gwd, denc = O.mlp_bwd(wd, enc, actd, dyd, 32, 64, 1, 16)
    return raw, gwd, gwc, denc


def test_default_forward_and_backward_at_the_edge_of_the_fp16_range(O, dev):
    """XR_MLP_F16X2 (the default forward) and the h2f backward at the documented boundary: hash-grid features up to +-4000 (x 2^4 = 64000 <
    65504) and hidden activations up to ~6e4 are carried like any other value -- forward within 3e-6 of max|raw| of the oracle, gradients
    at the plain bar -- and the range word stays 0.  Beyond the boundary (features of 1e4, activations above 65504) the operands
    saturate: every output is finite, equal to the oracle's evaluation of the saturated operands where only the inputs saturate, and the
    range word counts the event (`mlp_range_events` in the bench line reads the same word)."""
    from xrnerf_amd import ops, synthetic as S
    assert ops.f32_forward() == 'f16x2'
    rng = np.random.default_rng(65504)
    n = 4096
    wd, wc = nets(S)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)

    def run(enc, wd_, wc_, backward):
        ops.mlp_range_events(dev, reset=True)
        enc_t = T(np.ascontiguousarray(enc.T), dev)
        raw = ops.nerf_mlp_fwd(enc_t, T(dirs, dev), n, T(wd_, dev), T(wc_, dev), 1, 2).cpu().numpy()
        ev = ops.mlp_range_events(dev, reset=True)
        if not backward:
            return raw, ev
        g_wd = torch.zeros(wd_.size, dtype=torch.float32, device=dev); g_wc = torch.zeros(wc_.size, dtype=torch.float32, device=dev)
        denc_t = ops.nerf_mlp_bwd(enc_t, T(dirs, dev), n, T(wd_, dev), T(wc_, dev), 1, 2, T(draw, dev), g_wd, g_wc)
        return raw, ev, g_wd.cpu().numpy(), g_wc.cpu().numpy(), denc_t[:, :n].t().cpu().numpy()

    # (a) at the boundary, inside: features +-4000, first-layer weights scaled so that the hidden activations reach ~3.6e4
    enc = rng.uniform(-4000, 4000, (n, 32)).astype(np.float32)
    enc[0, :] = 4000.0; enc[1, :] = -4000.0
    # hidden activations to ~3.6e4; the density outputs (the colour net's inputs) ~8.5e3, the colour net's own activations ~6e2.  (The
    # weights stay O(0.01 .. 1): the split carries an operand below 2^-3 to 2^-25 ABSOLUTE -- its low part is an fp16 subnormal -- so
    # weights of 1e-5 against activations of 1e4 would be a test of that documented floor, not of the range.)
    wd_a = wd.copy(); wd_a[:2048] *= 4.0; wd_a[2048:] *= 0.25
    wc = wc.copy(); wc[:2048] *= 0.125
    ref, gwd, gwc, denc = _chain_oracle(O, wd_a, wc, enc, dirs, draw=draw)
    _, hmax = _chain_oracle(O, wd_a, wc, enc, dirs)
    assert 2e4 < hmax < 65504.0, hmax
    raw, ev, g_wd, g_wc, d_enc = run(enc, wd_a, wc, True)
    assert ev == 0
    scale = np.abs(ref).max()
    assert np.isfinite(raw).all() and np.abs(raw - ref).max() <= 3e-6 * scale, (np.abs(raw - ref).max(), scale)
    for name, got, want in (('wd', g_wd, gwd), ('wc', g_wc, gwc), ('denc', d_enc, denc)):
        assert np.isfinite(got).all(), name
        grad_close(got / max(np.abs(want).max(), 1e-30), want / max(np.abs(want).max(), 1e-30), name)
    # (b) features beyond 4094: saturated to 65504 / 16 on the way in, counted, == the oracle on the saturated features
    enc_b = enc.copy(); enc_b[5, :] = 1e4; enc_b[6, 3] = -3e5
    ref_b, _ = _chain_oracle(O, wd_a, wc, enc_b, dirs, sat=True)
    raw_b, ev_b = run(enc_b, wd_a, wc, False)
    assert ev_b >= 1 and np.isfinite(raw_b).all()
    assert np.abs(raw_b - ref_b).max() <= 3e-6 * np.abs(ref_b).max(), (np.abs(raw_b - ref_b).max(), np.abs(ref_b).max())
    # (c) hidden activations beyond 65504 (a first layer twice as large on the same features): finite everywhere, counted; forward and backward
    wd_c = wd.copy(); wd_c[:2048] *= 8.0; wd_c[2048:] *= 0.25
    _, hmax_c = _chain_oracle(O, wd_c, wc, enc, dirs)
    assert hmax_c > 65504.0, hmax_c
    raw_c, ev_c, g_wd_c, g_wc_c, d_enc_c = run(enc, wd_c, wc, True)
    assert ev_c >= 1
    assert np.isfinite(raw_c).all() and np.isfinite(g_wd_c).all() and np.isfinite(g_wc_c).all() and np.isfinite(d_enc_c).all()
    ok = _chain_oracle(O, wd_c, wc, enc, dirs)[0]
    rows = np.abs(ok).max(1) > 0                                       # (every row; the saturated ones differ, boundedly)
    assert np.abs(raw_c[rows]).max() <= 2.0 * np.abs(ok).max()
    # (d) weights beyond the range are saturated and counted as well
    wd_d = wd_a.copy(); wd_d[7] = 1e5
    raw_d, ev_d = run(enc * 1e-6, wd_d, wc, False)
    assert ev_d >= 1 and np.isfinite(raw_d).all()
    # the fp32 MFMA forward is the escape: no saturation, no count
    ops.set_f32_forward('mfma')
    try:
        raw_e, ev_e = run(enc_b, wd_a, wc, False)
    finally:
        ops.set_f32_forward('f16x2')
    ref_e, _ = _chain_oracle(O, wd_a, wc, enc_b, dirs)
    assert ev_e == 0 and np.abs(raw_e - ref_e).max() <= 3e-6 * np.abs(ref_e).max()


@pytest.mark.parametrize('n', [5000, 40000])
def test_default_backward_on_kink_free_rows_at_the_plain_bar(O, dev, n, monkeypatch):
    """The DEFAULT backward (recompute on split fp16 operands, XR_MLP_BWD_DW=h2f) against the oracle at the bar of the exact modes
    (1e-3 * max, no kink allowance): the oracle reports per sample how close its nearest hidden pre-activation comes to zero
    (xo_nerf_mlp_kink_margin: |z| / sum |w x|); samples within 1e-5 -- where two correct evaluations may disagree about relu'(z) --
    get a zero dL/d(raw) on both sides, everything else is compared in full."""
    from xrnerf_amd import ops, synthetic as S
    monkeypatch.delenv('XR_MLP_BWD_DW', raising=False)
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(n + 11)
    table = S.hash_table(meta.n_params, scale=0.5)
    wd, wc = nets(S)
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)
    margin = O.nerf_mlp_kink_margin(table, wd, wc, pts, dirs, om)
    near = margin < 1e-5
    assert near.mean() < 0.02, near.mean()
    draw[near] = 0.0
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, draw, om)
    tt, tp = T(table, dev), T(pts, dev)
    enc_t = ops.hashgrid_fwd(tt, tp, meta)
    g_wd = torch.zeros(wd.size, dtype=torch.float32, device=dev)
    g_wc = torch.zeros(wc.size, dtype=torch.float32, device=dev)
    denc_t = ops.nerf_mlp_bwd(enc_t, T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2, T(draw, dev), g_wd, g_wc)
    g_t = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
    ops.hashgrid_bwd(tp, denc_t, meta, g_t)
    for name, got, ref in (('wd', g_wd, gd), ('wc', g_wc, gc), ('table', g_t, gt)):
        grad_close(got.cpu().numpy(), ref, name, kinks=False)


def test_nerf_mlp_fwd_asymmetric_weights(O, dev, f32_forward):
    """transpose / permutation detector: one-hot weights so that each output picks a known input."""
    from xrnerf_amd import ops
    n = 64
    rng = np.random.default_rng(0)
    enc = rng.uniform(0.1, 1, (n, 32)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    wd = np.zeros(64 * 32 + 16 * 64, np.float32)
    W0 = wd[:2048].reshape(64, 32); W1 = wd[2048:].reshape(16, 64)
    for o in range(64): W0[o, (o * 5 + 3) % 32] = 1.0 + o * 0.01
    for o in range(16): W1[o, (o * 7 + 1) % 64] = 1.0 + o * 0.1
    wc = np.zeros(64 * 32 + 64 * 64 + 16 * 64, np.float32)
    C0 = wc[:2048].reshape(64, 32); C1 = wc[2048:2048 + 4096].reshape(64, 64); C2 = wc[6144:].reshape(16, 64)
    for o in range(64): C0[o, (o * 3 + 2) % 32] = 0.5 + o * 0.01
    for o in range(64): C1[o, (o * 11 + 5) % 64] = 1.0 - o * 0.005
    for o in range(16): C2[o, (o * 13 + 7) % 64] = 1.0 + o * 0.2
    h = np.maximum(enc @ W0.T, 0); dout = h @ W1.T
    cin = np.concatenate([dout[:, 1:], O.sh4(dirs), np.ones((n, 1), np.float32)], 1)
    c = np.maximum(cin @ C0.T, 0); c = np.maximum(c @ C1.T, 0); cout = c @ C2.T
    ref = np.concatenate([cout[:, :3], dout[:, :1]], 1)
    enc_t = T(enc, dev).t().contiguous()
    raw = ops.nerf_mlp_fwd(enc_t, T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2).cpu().numpy()
    assert np.abs(raw - ref).max() <= 1e-4


@pytest.mark.parametrize('n', [32, 100, 5000, 40001])
def test_nerf_mlp_bwd(O, dev, n):
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(n + 1)
    table = S.hash_table(meta.n_params, scale=0.5)
    wd, wc = nets(S)
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, draw, om)
    tt, tp = T(table, dev), T(pts, dev)
    enc_t = ops.hashgrid_fwd(tt, tp, meta)
    g_wd = torch.zeros(wd.size, dtype=torch.float32, device=dev)
    g_wc = torch.zeros(wc.size, dtype=torch.float32, device=dev)
    denc_t = ops.nerf_mlp_bwd(enc_t, T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2, T(draw, dev), g_wd, g_wc)
    g_t = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
    ops.hashgrid_bwd(tp, denc_t, meta, g_t)
    for name, got, ref in (('wd', g_wd, gd), ('wc', g_wc, gc), ('table', g_t, gt)):
        # both sides sum n fp32 terms in different orders (the oracle serially): ~sqrt(n)*2^-24 relative
        grad_close(got.cpu().numpy(), ref, name)


@pytest.mark.parametrize('arith', ['f32', 'b2', 'b2x', 'h2f'])
def test_nerf_mlp_bwd_arithmetic_modes(O, dev, arith, monkeypatch):
    """XR_MLP_BWD_DW: the backward with every product on the fp32 MFMA, with the dW products on the bf16 matrix cores (2-way
    operand split), and (the default) with the dX chain there too -- each against the oracle at the same 1e-3 * max bar"""
    monkeypatch.setenv('XR_MLP_BWD_DW', 'bf16')                 # not a mode: an error, not a silent default
    with pytest.raises(Exception, match='XR_MLP_BWD_DW'):
        test_nerf_mlp_bwd(O, dev, 32)
    monkeypatch.setenv('XR_MLP_BWD_DW', arith)
    test_nerf_mlp_bwd(O, dev, 5000)
    test_nerf_mlp_bwd_live_rows(O, dev, 5000, 4100, 'f32')
    test_nerf_mlp_bwd_live_rows(O, dev, 31, None, 'f32')


@pytest.mark.parametrize('n', [3000])
def test_nerf_mlp_bwd_split_recompute_differs_by_relu_kinks_only(dev, n, monkeypatch):
    """XR_MLP_BWD_DW=b2f also recomputes the forward with 2-way split operands (2^-16 relative).  A hidden unit whose
    pre-activation is that close to zero can land on the other side of its ReLU than in the fp32 recompute; apart from those
    unit-samples the two backward passes agree closely: per sample the encoding gradient is either within 1e-3 of its scale or
    the sample is one of a few with a flipped unit, and the weight gradients agree in the 2-norm."""
    from xrnerf_amd import ops, synthetic as S
    rng = np.random.default_rng(n)
    wd, wc = nets(S)
    enc_t = T(rng.normal(0, 0.5, (32, n)).astype(np.float32), dev)
    dirs = T(rng.uniform(0, 1, (n, 3)).astype(np.float32), dev)
    draw = T(rng.normal(0, 1, (n, 4)).astype(np.float32), dev)
    out = {}
    for arith in ('b2x', 'b2f'):
        monkeypatch.setenv('XR_MLP_BWD_DW', arith)
        g_wd = torch.zeros(wd.size, dtype=torch.float32, device=dev_of(enc_t))
        g_wc = torch.zeros(wc.size, dtype=torch.float32, device=dev_of(enc_t))
        denc_t = ops.nerf_mlp_bwd(enc_t, dirs, n, T(wd, dev), T(wc, dev), 1, 2, draw, g_wd, g_wc)
        out[arith] = (denc_t[:, :n].cpu().numpy().astype(np.float64), g_wd.cpu().numpy().astype(np.float64), g_wc.cpu().numpy().astype(np.float64))
    (da, wda, wca), (db, wdb, wcb) = out['b2x'], out['b2f']
    row_err = np.abs(da - db).max(0) / max(np.abs(da).max(0).mean(), 1e-30)
    assert np.median(row_err) <= 1e-4, np.median(row_err)
    assert (row_err > 1e-3).mean() <= 0.05, (row_err > 1e-3).mean()          # samples with a flipped unit
    for a, b in ((wda, wdb), (wca, wcb)):
        assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(a), np.linalg.norm(a - b) / np.linalg.norm(a)


def dev_of(t):
    return t.device


@pytest.mark.parametrize('path', ['streamed', 'layered'])
@pytest.mark.parametrize('nhd,nhc,n,n_valid', [(5, 5, 300, None), (5, 5, 5000, 4100), (3, 4, 257, None), (2, 2, 100, None), (1, 1, 33, None),
                                               (8, 8, 700, 650), (2, 1, 1000, None)])
def test_deeper_topologies_against_the_oracle(O, dev, nhd, nhc, n, n_valid, path, monkeypatch):
    """tiny-cuda-nn's own default depth (5 hidden layers: what the reference's unchanged config builds if tcnn ignores its
    `num_layers` key) and any other depth: forward and backward against the oracle through the same ops entry
    points -- on the STREAMED fused kernels (k_nerf_mlp_fwd_deep / _bwd_deep: the layers' weights pass through LDS, round 5) and on
    the layer-by-layer path over the linear kernels (the independent statement of rounds 3-4)"""
    from xrnerf_amd import ops, synthetic as S
    if path == 'layered':
        monkeypatch.setattr(ops, '_FUSED_FWD', ((1, 2),))
        monkeypatch.setattr(ops, '_FUSED_BWD', ((1, 2),))
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(nhd * 10 + nhc + n)
    table = S.hash_table(meta.n_params, scale=0.5)
    wd, wc = S.mlp_weights(32, 64, nhd, 16, 4), S.mlp_weights(32, 64, nhc, 16, 5)
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)
    nv = n if n_valid is None else n_valid
    draw_ref = draw.copy(); draw_ref[nv:] = 0
    ref = O.nerf_mlp_fwd(table, wd, wc, pts, dirs, om, nhd, nhc)
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, draw_ref, om, nhd, nhc)
    tt, tp, td = T(table, dev), T(pts, dev), T(dirs, dev)
    n_dev = None if n_valid is None else torch.tensor([n_valid], dtype=torch.int32, device=dev)
    enc_t = ops.hashgrid_fwd(tt, tp, meta)
    raw = ops.nerf_mlp_fwd(enc_t, td, n, T(wd, dev), T(wc, dev), nhd, nhc, n_dev=n_dev)
    assert np.abs(raw.cpu().numpy()[:nv] - ref[:nv]).max() <= 1e-4
    dens = ops.nerf_mlp_fwd(enc_t, None, n, T(wd, dev), None, nhd, nhc)
    assert np.abs(dens.cpu().numpy()[:, 3] - ref[:, 3]).max() <= 1e-4 and float(dens[:, :3].abs().max()) == 0.0
    g_wd = torch.zeros(wd.size, dtype=torch.float32, device=dev)
    g_wc = torch.zeros(wc.size, dtype=torch.float32, device=dev)
    denc_t = ops.nerf_mlp_bwd(enc_t, td, n, T(wd, dev), T(wc, dev), nhd, nhc, T(draw, dev), g_wd, g_wc, n_dev=n_dev)
    g_t = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
    ops.hashgrid_bwd(tp, denc_t, meta, g_t, n_dev=n_dev)
    for name, got, refg in (('wd', g_wd, gd), ('wc', g_wc, gc), ('table', g_t, gt)):
        grad_close(got.cpu().numpy(), refg, name, kinks=True)


def test_streamed_kernels_equal_the_layer_by_layer_path_at_full_size(dev, monkeypatch):
    """tcnn's default 5 + 5 hidden layers at the training step's size: 2^18 rows of which 2.3e5 are valid, the rows behind the count
    holding NaN bit patterns in the encoded features and in dL/d(raw), half of the valid rows dead (exactly zero gradient) and the
    backward on the live-row list.  The streamed fused kernels against the layer-by-layer path (fp32 linear kernels, activations
    through HBM), which the oracle pins at small sizes: forward, dL/d(encoding) and both weight gradients."""
    from xrnerf_amd import ops, synthetic as S
    n, nv, nhd, nhc = 1 << 18, 230000, 5, 5
    g = torch.Generator(device='cpu').manual_seed(7)
    wd, wc = T(S.mlp_weights(32, 64, nhd, 16, 4), dev), T(S.mlp_weights(32, 64, nhc, 16, 5), dev)
    enc_t = (torch.randn((32, n), generator=g) * 0.5).to(dev)
    dirs = torch.rand((n, 3), generator=g).to(dev)
    draw = torch.randn((n, 4), generator=g).to(dev)
    draw[torch.rand((n,), generator=g).to(dev) < 0.5] = 0.0                # dead rows, as the compositor leaves them
    n_dev = torch.tensor([nv], dtype=torch.int32, device=dev)
    enc_t[:, nv:] = float('nan')
    draw[nv:] = float('nan')
    out = {}
    for kind in ('streamed', 'streamed_live', 'layered'):
        if kind == 'layered':
            monkeypatch.setattr(ops, '_FUSED_FWD', ((1, 2),))
            monkeypatch.setattr(ops, '_FUSED_BWD', ((1, 2),))
        raw = ops.nerf_mlp_fwd(enc_t, dirs, n, wd, wc, nhd, nhc, n_dev=n_dev)
        g_wd, g_wc = torch.zeros_like(wd), torch.zeros_like(wc)
        live = ops.live_rows(draw, n, n_dev=n_dev) if kind == 'streamed_live' else None
        denc_t = torch.zeros_like(enc_t)
        ops.nerf_mlp_bwd(enc_t, dirs, n, wd, wc, nhd, nhc, draw, g_wd, g_wc, denc_t=denc_t, n_dev=n_dev, live=live)
        out[kind] = (raw[:nv].clone(), g_wd, g_wc, denc_t[:, :nv].clone())
    # Both sides recompute the forward to fp32 rounding accuracy in different summation orders, so a hidden unit whose pre-activation is
    # within an ulp or two of zero can sit on the other side of its ReLU (~1e2 of the 1.5e8 unit-samples here).  Such a flip is one
    # sample's whole contribution to a weight row and one sample's encoding gradient: the weight gradients are compared in the 2-norm
    # (and loosely entry by entry), the encoding gradient row by row.
    for kind in ('streamed', 'streamed_live'):
        raw, g_wd, g_wc, denc = out[kind]
        lraw, lg_wd, lg_wc, ldenc = out['layered']
        for t in (raw, g_wd, g_wc, denc, lraw, lg_wd, lg_wc, ldenc):
            assert bool(torch.isfinite(t).all()), kind
        assert float((raw - lraw).abs().max()) <= 2e-5 * max(1.0, float(lraw.abs().max())), kind
        for name, a, b in (('g_wd', g_wd, lg_wd), ('g_wc', g_wc, lg_wc)):
            a, b = a.double(), b.double()
            assert float((a - b).norm()) <= 3e-3 * float(b.norm()), (kind, name, float((a - b).norm()) / float(b.norm()))
            assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max()), (kind, name)
        row_err = (denc - ldenc).abs().amax(0) / float(ldenc.abs().amax(0).mean())
        assert float(row_err.median()) <= 1e-4 and float((row_err > 1e-3).float().mean()) <= 0.01, (kind, float(row_err.median()), float((row_err > 1e-3).float().mean()))
    # the live-row list changes nothing: same kernel, same rows in the same order within a tile batch... up to the batches' composition
    a, b = out['streamed'], out['streamed_live']
    assert float((a[3] - b[3]).abs().max()) <= 1e-6 * float(a[3].abs().max()) and float((a[1] - b[1]).abs().max()) <= 1e-4 * float(a[1].abs().max())


def test_layer_by_layer_path_equals_the_fused_kernels_at_full_size(dev, monkeypatch):
    """The layer-by-layer path (what any topology but (1, 2) trains on, e.g. tcnn's default 5 + 5) at the training step's size: 2^18
    rows of which 2.3e5 are valid, the rows behind the count holding NaN bit patterns in the encoded features and in dL/d(raw) (what
    fresh memory may hold).  On the (1, 2) topology both paths exist: forward and all three gradients of the layered path against the
    fused kernels' (fp32 MFMA forward and backward, the parity arithmetic), which the oracle pins at small sizes."""
    from xrnerf_amd import ops, synthetic as S
    monkeypatch.setenv('XR_MLP_BWD_DW', 'f32')
    n, nv = 1 << 18, 230000
    meta = ops.GridMeta()
    g = torch.Generator(device='cpu').manual_seed(5)
    table = T(S.hash_table(meta.n_params, scale=0.5), dev)
    wd, wc = T(S.mlp_weights(32, 64, 1, 16, 4), dev), T(S.mlp_weights(32, 64, 2, 16, 5), dev)
    pts = torch.rand((n, 3), generator=g).to(dev)
    dirs = torch.rand((n, 3), generator=g).to(dev)
    draw = torch.randn((n, 4), generator=g).to(dev)
    draw[torch.rand((n,), generator=g).to(dev) < 0.5] = 0.0                # dead rows, as the compositor leaves them
    n_dev = torch.tensor([nv], dtype=torch.int32, device=dev)
    enc_t = ops.hashgrid_fwd(table, pts, meta)
    enc_t[:, nv:] = float('nan')
    draw[nv:] = float('nan')
    old_f = ops.f32_forward()
    ops.set_f32_forward('mfma')
    try:
        out = {}
        for kind in ('fused', 'layered'):
            monkeypatch.setattr(ops, '_FUSED_FWD', ((1, 2),) if kind == 'fused' else ())
            monkeypatch.setattr(ops, '_FUSED_BWD', ((1, 2),) if kind == 'fused' else ())
            raw = ops.nerf_mlp_fwd(enc_t, dirs, n, wd, wc, 1, 2, n_dev=n_dev)
            g_wd, g_wc = torch.zeros_like(wd), torch.zeros_like(wc)
            denc_t = ops.nerf_mlp_bwd(enc_t, dirs, n, wd, wc, 1, 2, draw, g_wd, g_wc, n_dev=n_dev)
            g_t = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
            ops.hashgrid_bwd(pts, denc_t, meta, g_t, n_dev=n_dev)
            out[kind] = (raw[:nv].clone(), g_wd, g_wc, g_t, denc_t[:, :nv].clone())
    finally:
        ops.set_f32_forward(old_f)
    for name, a, b in zip(('raw', 'g_wd', 'g_wc', 'g_table', 'denc'), out['layered'], out['fused']):
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), name
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert scale > 0 and err <= (1e-5 if name == 'raw' else 2e-5) * max(1.0, scale), (name, err, scale)


def sparse_draw(rng, n, dead_fraction=0.55):
    """dL/d(raw) the way the compositor produces it: runs of exactly-zero rows (samples behind an opaque surface), a few
    isolated zero rows, rows with a single non-zero component, -0.0 entries"""
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)
    pos = 0
    while pos < n:
        run = int(rng.integers(1, 90))
        if rng.uniform() < dead_fraction:
            draw[pos:pos + run] = 0.0
        pos += run
    draw[rng.integers(0, n, max(1, n // 50))] = 0.0
    one = rng.integers(0, n, max(1, n // 50))
    draw[one] = 0.0
    draw[one, rng.integers(0, 4, one.size)] = 0.5
    draw[rng.integers(0, n, max(1, n // 100)), 1] = -0.0
    return draw


@pytest.mark.parametrize('n,n_valid', [(31, None), (100, None), (5000, 4100), (40001, None), (3000, 0)])
@pytest.mark.parametrize('precision', ['f32', 'f16'])
def test_nerf_mlp_bwd_live_rows(O, dev, n, n_valid, precision):
    """the backward runs on the compaction of the rows with a non-zero dL/d(raw) (xr_mlp.hip: k_live_count / k_live_fill):
    same dW, and an exactly-zero dL/d(encoding) for the dead rows, as the run over every row (XR_MLP_LIVE=0 is that run;
    here the check is against the oracle and, for dead rows, exact)"""
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(7 * n + 1)
    table = S.hash_table(meta.n_params, scale=0.5)
    wd, wc = nets(S)
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = sparse_draw(rng, n)
    if n == 31:
        draw[:] = 0.0; draw[17, 3] = 1.0                   # one live row in the whole launch
    if precision == 'f16':
        draw *= 1e-2                                       # the magnitude the loss-scaled fp16 chain is built for
    nv = n if n_valid is None else n_valid
    dref = draw.copy(); dref[nv:] = 0.0
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, dref, om)
    tt, tp = T(table, dev), T(pts, dev)
    enc_t = ops.hashgrid_fwd(tt, tp, meta)
    g_wd = torch.zeros(wd.size, dtype=torch.float32, device=dev)
    g_wc = torch.zeros(wc.size, dtype=torch.float32, device=dev)
    n_dev = None if n_valid is None else torch.tensor([nv], dtype=torch.int32, device=dev)
    old = ops.precision()
    ops.set_precision(precision)
    try:
        denc_t = ops.nerf_mlp_bwd(enc_t, T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2, T(draw, dev), g_wd, g_wc, n_dev=n_dev)
    finally:
        ops.set_precision(old)
    d = denc_t.cpu().numpy()[:, :nv]
    dead = ~(dref[:nv] != 0).any(1)
    assert dead.sum() > 0 or nv == 0
    assert not d[:, dead].any(), 'dead rows must get an exactly-zero encoding gradient'
    if nv == 0:
        assert not g_wd.cpu().numpy().any() and not g_wc.cpu().numpy().any()
        return
    g_t = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
    ops.hashgrid_bwd(tp, denc_t, meta, g_t, n_dev=n_dev)
    tol = 1e-3 if precision == 'f32' else 3e-2             # fp16 operands: tests/f16_reference.py has the tight model
    for name, got, ref in (('wd', g_wd, gd), ('wc', g_wc, gc), ('table', g_t, gt)):
        err = np.abs(got.cpu().numpy() - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), (name, err, np.abs(ref).max())


@pytest.mark.parametrize('n,n_valid', [(100, None), (20000, 17000), (40001, None), (3000, 0)])
def test_shared_live_row_list_through_backward_and_scatter(O, dev, n, n_valid):
    """the training step's arrangement: ONE list (ops.live_rows) drives the MLP backward and the table scatter, rows outside
    it are never read or written (denc_t is poisoned with NaN to prove it); n >= 16384 takes the binned scatter"""
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(11 * n + 3)
    table = S.hash_table(meta.n_params, scale=0.5)
    wd, wc = nets(S)
    # ray-like positions: runs of consecutive rows walk through the volume (the dense levels' run-length merging sees runs)
    pts = np.clip(np.cumsum(rng.normal(0, 0.004, (n, 3)), 0) % 1.0, 0, 1).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = sparse_draw(rng, n)
    nv = n if n_valid is None else n_valid
    dref = draw.copy(); dref[nv:] = 0.0
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, dref, om)
    tt, tp, tdraw = T(table, dev), T(pts, dev), T(draw, dev)
    enc_t = ops.hashgrid_fwd(tt, tp, meta)
    g_wd = torch.zeros(wd.size, dtype=torch.float32, device=dev)
    g_wc = torch.zeros(wc.size, dtype=torch.float32, device=dev)
    n_dev = None if n_valid is None else torch.tensor([nv], dtype=torch.int32, device=dev)
    live = ops.live_rows(tdraw, n, n_dev=n_dev)
    rows, n_live = live[0].cpu().numpy(), int(live[1].cpu().numpy()[0])
    want = np.nonzero((dref != 0).any(1))[0]
    assert n_live == want.size and np.array_equal(rows[:n_live], want)          # stable compaction
    denc_t = torch.full_like(enc_t, float('nan'))
    ops.nerf_mlp_bwd(enc_t, T(dirs, dev), n, T(wd, dev), T(wc, dev), 1, 2, tdraw, g_wd, g_wc, denc_t=denc_t, n_dev=n_dev, live=live)
    d = denc_t.cpu().numpy()
    assert np.isfinite(d[:, want]).all() and np.isnan(np.delete(d[:, :n], want, axis=1)).all()
    g_t = torch.zeros(meta.n_params, dtype=torch.float32, device=dev)
    ops.hashgrid_bwd(tp, denc_t, meta, g_t, live=live)
    for name, got, ref in (('wd', g_wd, gd), ('wc', g_wc, gc), ('table', g_t, gt)):
        grad_close(got.cpu().numpy(), ref, name)


def test_tcnn_module_surface_runs_the_reference_mlp_recipe(O, dev):
    """xrnerf_amd.tcnn.{Encoding,Network} used exactly as xrnerf/models/mlps/hashnerf_mlp.py:34-45,55-79 uses
    tinycudann: separate modules, row-major tensors, torch.cat in between, autograd end to end -- and the result
    equals the fused HashNerfMLP path and the oracle."""
    from xrnerf_amd import tcnn, ops, synthetic as S
    from xrnerf_amd.mlps import get_per_level_scale
    torch.manual_seed(0)
    emb = tcnn.Encoding(3, dict(otype='HashGrid', n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                                base_resolution=16, interpolation='Linear', per_level_scale=get_per_level_scale(1))).to(dev)
    sh = tcnn.Encoding(3, dict(otype='SphericalHarmonics', degree=4)).to(dev)
    dnet = tcnn.Network(emb.n_output_dims, 16, dict(otype='FullyFusedMLP', activation='ReLU', output_activation='None',
                                                    n_neurons=64, num_layers=1)).to(dev)
    cnet = tcnn.Network(sh.n_output_dims + 16 - 1, 3, dict(otype='FullyFusedMLP', activation='ReLU',
                                                           output_activation='None', n_neurons=64, num_layers=2)).to(dev)
    with torch.no_grad():
        emb.params.copy_(T(S.hash_table(emb.meta.n_params, scale=0.5), dev))
    n = 3001
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32); dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)
    # --- the reference's run_mlp, line for line in spirit
    density_out = dnet(emb(T(pts, dev)))
    color_out = cnet(torch.cat([density_out[..., 1:], sh(T(dirs, dev))], dim=-1))
    outputs = torch.cat([color_out, density_out[..., :1]], -1)
    assert outputs.shape == (n, 4)
    (outputs * T(draw, dev)).sum().backward()
    om = O.GridMeta()
    table, wd, wc = emb.params.detach().cpu().numpy(), dnet.params.detach().cpu().numpy(), cnet.params.detach().cpu().numpy()
    ref = O.nerf_mlp_fwd(table, wd, wc, pts, dirs, om)
    assert np.abs(outputs.detach().cpu().numpy() - ref).max() <= 1e-4
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, draw, om)
    for got, want in ((emb.params.grad, gt), (dnet.params.grad, gd), (cnet.params.grad, gc)):
        grad_close(got.cpu().numpy(), want)
    # 3 hidden layers forward (strict-default-like depth) against the oracle's generic MLP
    net3 = tcnn.Network(32, 16, dict(otype='FullyFusedMLP', n_neurons=64, n_hidden_layers=3)).to(dev)
    x = rng.normal(0, 0.5, (500, 32)).astype(np.float32)
    y = net3(T(x, dev)).detach().cpu().numpy()
    assert np.abs(y - O.mlp_fwd(net3.params.detach().cpu().numpy(), x, 32, 64, 3, 16)).max() <= 1e-4


@pytest.mark.parametrize('n', [32, 100, 5000, 40001])
def test_nerf_mlp_reference_precision_mode(dev, n):
    """fp16-MFMA mode (tiny-cuda-nn's own arithmetic: fp16 weights / activations, fp32 accumulation) against the numpy
    statement with the same rounding points (tests/f16_reference.py).  A hidden activation that lands on an fp16 rounding
    boundary may round the other way when the fp32 sum is taken in another order: 1 fp16 ulp on that activation, hence the
    tolerances (raw: 5e-3 abs at |raw| ~ 2; gradients: 2e-3 of the largest entry).  Also: distance to the fp32 mode."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import f16_reference as R
    from xrnerf_amd import ops
    rng = np.random.default_rng(n)
    enc = rng.normal(0, 0.5, (n, 32)).astype(np.float32)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    wd = rng.uniform(-0.4, 0.4, 3072).astype(np.float32)
    wc = rng.uniform(-0.3, 0.3, 7168).astype(np.float32)
    draw = rng.normal(0, 1e-2, (n, 4)).astype(np.float32)
    draw[rng.uniform(0, 1, n) < 0.2] = 0
    ld = (n + 63) // 64 * 64
    enc_t = torch.zeros((32, ld), dtype=torch.float32, device=dev)
    enc_t[:, :n] = T(enc, dev).t()
    td, twd, twc = T(dirs, dev), T(wd, dev), T(wc, dev)
    prev = ops.precision()
    try:
        ops.set_precision('f16')
        raw = ops.nerf_mlp_fwd(enc_t, td, n, twd, twc, 1, 2).cpu().numpy()
        rawd = ops.nerf_mlp_fwd(enc_t, None, n, twd, None, 1, 2).cpu().numpy()
        gwd = torch.zeros(3072, dtype=torch.float32, device=dev)
        gwc = torch.zeros(7168, dtype=torch.float32, device=dev)
        denc_t = ops.nerf_mlp_bwd(enc_t, td, n, twd, twc, 1, 2, T(draw, dev), gwd, gwc).cpu().numpy()
        ops.set_precision('f32')
        raw32 = ops.nerf_mlp_fwd(enc_t, td, n, twd, twc, 1, 2).cpu().numpy()
    finally:
        ops.set_precision(prev)
    ref = R.forward(enc, dirs, wd, wc)
    de, rwd, rwc = R.backward(enc, dirs, wd, wc, draw)
    assert np.abs(raw - ref).max() <= 5e-3, np.abs(raw - ref).max()
    assert np.abs(rawd[:, 3] - ref[:, 3]).max() <= 5e-3
    assert np.abs(raw - raw32).max() <= 2e-2, np.abs(raw - raw32).max()          # what the reference's precision costs
    for name, got, want in (('denc', denc_t[:, :n].T, de), ('wd', gwd.cpu().numpy(), rwd), ('wc', gwc.cpu().numpy(), rwc)):
        err, big = np.abs(got - want), np.abs(want).max()
        # a pre-activation within summation error of zero flips its ReLU mask: ONE sample's gradient through ONE neuron changes
        # entirely -- its 32 feature gradients move by a few % of the largest, and a row / column of the weight gradients by
        # one |g h| term (measured on the MI355X: none at n = 5000 / 8193 / 33000, one or two at 32768 / 40001 / 100000; the
        # colour output layer, which has no mask behind it, always agrees to 1e-4).  Bounded and rare:
        assert np.quantile(err, 0.99) <= 2e-3 * big and err.max() <= 0.2 * big, (name, err.max(), np.quantile(err, 0.99), big)


@pytest.mark.parametrize('n', [1000, 40001])
def test_density_query_with_the_splat_inside_equals_query_plus_splat(dev, n, f32_forward):
    """xr_nerf_density_splat (K9's density query with K8 in the forward kernel's epilogue) against xr_nerf_mlp_fwd (density only) +
    xr_splat_grid_samples: the same order-free maxima, cell for cell and bit for bit, heavy collisions included; in the fp16 mode too"""
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta()
    rng = np.random.default_rng(n)
    table = T(S.hash_table(meta.n_params, scale=1.0), dev)
    wd = T(S.mlp_weights(32, 64, 1, 16, seed=4) * 3.0, dev)
    pts = T(rng.uniform(0, 1, (n, 3)).astype(np.float32), dev)
    idx = rng.integers(0, 128 ** 3, n).astype(np.int32)
    idx[:200] = idx[0]
    idx_d = T(idx, dev)
    modes = ['f32'] + (['f16'] if f32_forward == 'mfma' else [])
    for mode in modes:
        old = ops.precision()
        ops.set_precision(mode)
        try:
            enc_t = ops.hashgrid_fwd(table, pts, meta)
            raw = ops.nerf_mlp_fwd(enc_t, None, n, wd, None, 1, 2)
            a = torch.zeros(128 ** 3, dtype=torch.float32, device=dev)
            ops.splat_grid_samples(raw[:, 3:4], idx_d, raw.stride(0), n, a)
            b = torch.zeros_like(a)
            ops.nerf_density_splat(enc_t, n, wd, 1, 2, idx_d, b)
        finally:
            ops.set_precision(old)
        assert torch.equal(a, b) and float(a.max()) > 0
        assert int((a > 0).sum()) == len(np.unique(idx))


@pytest.mark.gpu
@pytest.mark.parametrize('n,fused', [(50001, False), (200000, False), (200000, True)])
def test_binned_scatter_is_the_same_bits_every_launch_and_keeps_small_gradients(O, dev, n, fused):
    """The LDS sums are 64-bit fixed point since round 6 (xr_scatter.hip, S3_FIX): integer additions, so a launch's result does not
    depend on the schedule -- two launches agree bit for bit (the fp64 sums of rounds 3-5 agreed up to one ulp, most of the time) --
    and the quantum, 2^-41 of a level's largest gradient at these sizes, is far below fp32's own resolution of entries six decades
    under that maximum."""
    from xrnerf_amd import ops, synthetic as S
    meta = ops.GridMeta(); om = O.GridMeta()
    rng = np.random.default_rng(7 + n)
    # ray-shaped positions (runs of samples inside one coarse cell: the run-length levels and the register merge see real runs)
    o = rng.uniform(0.2, 0.8, (n // 16 + 1, 1, 3)); d = rng.normal(0, 1, (n // 16 + 1, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = np.clip(o + d * (np.arange(16)[None, :, None] * 0.004), 0, 1).reshape(-1, 3)[:n].astype(np.float32)
    dy = (rng.normal(0, 1, (n, 32)) * 10.0 ** rng.uniform(-6, 0, (n, 1))).astype(np.float32)     # six decades inside one launch
    ld = (n + 63) // 64 * 64
    dt = torch.zeros((32, ld), dtype=torch.float32, device=dev); dt[:, :n] = T(dy, dev).t()
    xt = T(x, dev)

    def launch():
        if not fused:
            g = torch.full((meta.n_params,), 3.0, dtype=torch.float32, device=dev)
            ops.hashgrid_bwd(xt, dt, meta, g, overwrite=True)
            return [g]
        torch.manual_seed(3)
        p = torch.randn(meta.n_params, device=dev) * 1e-2
        m = torch.randn(meta.n_params, device=dev) * 1e-4; v = torch.rand(meta.n_params, device=dev) * 1e-8; ema = p.clone()
        ops.hashgrid_bwd_adam(xt, dt, meta, ops.adam_fuse(p, m, v, ema, 7, 1e-2, 0.9, 0.99, 1e-15, 1e-6, 0.05))
        return [p, m, v, ema]
    a = launch()
    for _ in range(3):
        b = launch()
        assert all(torch.equal(s, t) for s, t in zip(a, b))
    if fused:
        return
    ref = O.hashgrid_bwd(x, dy, om).astype(np.float64)
    g = a[0].cpu().numpy().astype(np.float64)
    top = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 1e-4 * top
    # entries down to 1e-7 of the largest one keep a relative accuracy the oracle's own fp32 products allow
    big = np.abs(ref) >= 1e-7 * top
    rel = np.abs(g - ref)[big] / np.abs(ref)[big]
    assert big.sum() > 1000 and np.quantile(rel, 0.99) <= 1e-5, float(np.quantile(rel, 0.99))     # (the rest: sums that cancel)


@pytest.mark.gpu
def test_binned_scatter_fixed_point_scale_follows_the_gradients_magnitude(O, dev):
    """The fixed-point scale is chosen per level from the launch's largest gradient: gradients of 1e-30 and of 1e+30 give the gradients of
    order 1 times that factor (no underflow to zero, no overflow), and a non-finite gradient makes the entries of ITS level NaN and leaves the
    other levels alone."""
    from xrnerf_amd import ops
    meta, om = ops.GridMeta(), O.GridMeta()
    n = 40000
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dy = rng.normal(0, 1, (n, 32)).astype(np.float32)
    ref = O.hashgrid_bwd(x, dy, om).astype(np.float64)
    ld = (n + 63) // 64 * 64
    xt = T(x, dev)

    def run(d):
        dt = torch.zeros((32, ld), dtype=torch.float32, device=dev); dt[:, :n] = T(d, dev).t()
        g = torch.full((meta.n_params,), 2.0, dtype=torch.float32, device=dev)
        ops.hashgrid_bwd(xt, dt, meta, g, overwrite=True)
        return g.cpu().numpy().astype(np.float64)
    for f in (1e-30, 1.0, 1e30):
        g = run((dy.astype(np.float64) * f).astype(np.float32))
        assert np.isfinite(g).all()
        assert np.abs(g / f - ref).max() <= 1e-4 * np.abs(ref).max(), f
    bad = dy.copy()
    bad[123, 2 * 9] = np.inf                                    # feature 0 of level 9
    g = run(bad)
    lo, hi = 2 * int(meta.offset[9]), 2 * int(meta.offset[10])
    assert np.isnan(g[lo:hi]).all()
    rest = np.concatenate([g[:lo], g[hi:]]); rref = np.concatenate([ref[:lo], ref[hi:]])
    assert np.isfinite(rest).all() and np.abs(rest - rref).max() <= 1e-4 * np.abs(ref).max()
