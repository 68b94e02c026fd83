"""The KiloNeRF kernels (xrnerf_amd/csrc/xr_kilo.hip: assignment, counting sort, the fp32-MFMA tiny-MLP forward AND
backward, NerfRender, the fused sparse frame call with per-ray spans) and the fp32-MFMA linear kernels (xr_gemm.hip) --
the SAME sources the GPU library is built from -- executed on the host by the HIP-on-CPU shim (tests/hip_emu), against the
oracles and the reference fixtures.  v_mfma_f32_32x32x2_f32 is emulated with its documented operand / accumulator lane
layout, so the register-resident layer chaining, the LDS-transposed outer products and the tile epilogues are exercised
for real.  tests/test_gpu_kilo.py / test_gpu_linear.py hold the same checks on the MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'hip_emu'))
G = os.path.join(ROOT, 'tests', 'golden')
LAYERS = ['pts_linears.0', 'pts_linears.1', 'alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear']


@pytest.fixture(scope='module')
def E():
    import emulib
    return emulib


@pytest.fixture(scope='module')
def K():
    import kilo_oracle
    return kilo_oracle


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(G, 'ref_kilonerf.npz'))


def scene(gold):
    import torch
    from xrnerf_amd import kilo
    sd = {}
    for nm in LAYERS:
        sd[nm + '.weight'], sd[nm + '.bias'] = torch.tensor(gold['w.' + nm]), torch.tensor(gold['b.' + nm])
    mn = kilo.MultiNetwork(24, 63, 27)
    mn.load_state_dict(sd)
    return mn


def kilo_args(E, gold, packed):
    f3, i3 = (C.c_float * 3), (C.c_int32 * 3)
    res = [int(v) for v in gold['res']]
    occ = np.ascontiguousarray(gold['occupancy'].astype(np.uint8))
    return dict(gmin=f3(*gold['gmin']), gmax=f3(*gold['gmax']), fixed=i3(*[r // 16 for r in res]), res=i3(*res), occ=occ,
                dmins=E.f32(gold['domain_mins']), dmaxs=E.f32(gold['domain_maxs']), params=packed)


def nets_from(K, g):
    return K.TinyNets([g['w.pts_linears.0'], g['w.pts_linears.1']], [g['b.pts_linears.0'], g['b.pts_linears.1']],
                      g['w.alpha_linear'], g['b.alpha_linear'], g['w.feature_linear'], g['b.feature_linear'],
                      g['w.direction_layer'], g['b.direction_layer'], g['w.rgb_linear'], g['b.rgb_linear'])


def test_kilo_forward_backward_and_render_on_the_reference_fixture(E, K, gold):
    from xrnerf_amd import kilo
    L = E.lib('xr_kilo')
    L.xr_kilo_workspace_bytes.restype = C.c_size_t
    mn = scene(gold)
    packed = E.aligned(tuple(mn.packed().shape), fill=mn.packed().numpy())
    a = kilo_args(E, gold, packed)
    R, S = gold['z_vals'].shape
    pts, vd = E.aligned((R, S, 3), fill=gold['pts']), E.f32(gold['viewdirs'])
    raw = E.aligned((R, S, 4), fill=7.0)
    counts = np.zeros(24, np.uint32)
    ws = E.aligned((int(L.xr_kilo_workspace_bytes(C.c_uint64(R * S), 24)),), np.uint8)
    common = (R, S, a['gmin'], a['gmax'], a['fixed'], a['res'], E.p(a['occ']), E.p(a['dmins']), E.p(a['dmaxs']), E.p(packed),
              packed.shape[1], 24, 10, 4, 2)
    E.check(L.xr_kilo_mlp_forward(E.p(pts), None, None, None, E.p(vd), *common, E.p(raw), E.p(counts), E.p(ws), C.c_size_t(ws.size), None), L)
    assert np.array_equal(counts.astype(np.int64), gold['batch_size_per_network'])
    active = np.zeros(R * S, bool); active[gold['active_samples']] = True
    assert np.array_equal(raw.reshape(-1, 4)[~active], np.zeros((int((~active).sum()), 4), np.float32))
    assert np.abs(raw - gold['raw']).max() <= 1e-4
    # sample positions evaluated on the fly from (o, d, z): same rows
    raw2 = E.aligned((R, S, 4))
    E.check(L.xr_kilo_mlp_forward(None, E.p(E.f32(gold['rays_o'])), E.p(E.f32(gold['rays_d'])), E.p(E.f32(gold['z_vals'])), E.p(vd), *common,
                                  E.p(raw2), None, E.p(ws), C.c_size_t(ws.size), None), L)
    assert np.array_equal(raw2, raw)
    # fine-tuning gradients vs the oracle adjoint (pinned to autograd through the reference's MultiNetwork)
    rng = np.random.default_rng(21)
    draw = E.aligned((R, S, 4), fill=rng.normal(0, 1, (R, S, 4)))
    grad = E.aligned(packed.shape)
    E.check(L.xr_kilo_mlp_backward(E.p(pts), None, None, None, E.p(vd), *common, E.p(draw), E.p(grad), 0, E.p(ws), C.c_size_t(ws.size), None), L)
    import torch
    got = kilo.MultiNetwork.unpack_like(torch.from_numpy(grad.copy()), mn.ordered_parameters())
    ref = K.mlp_raw_backward(draw, None, None, gold['viewdirs'], gold['z_vals'], gold['gmin'], gold['gmax'], [r // 16 for r in gold['res']],
                             gold['res'], gold['occupancy'], gold['domain_mins'], gold['domain_maxs'], nets_from(K, gold), pts=gold['pts'])
    for (name, _), t in zip(mn.named_parameters(), got):
        assert np.abs(t.numpy().astype(np.float64) - ref[name]).max() <= 1e-4 * max(1.0, np.abs(ref[name]).max()), name
    # reuse_assignment: a forward on the same samples leaves the assignment / order / segments in the workspace; the backward that skips
    # its own three preparation launches gives the same gradient blocks bit for bit
    E.check(L.xr_kilo_mlp_forward(E.p(pts), None, None, None, E.p(vd), *common, E.p(raw), None, E.p(ws), C.c_size_t(ws.size), None), L)
    grad2 = E.aligned(packed.shape)
    E.check(L.xr_kilo_mlp_backward(E.p(pts), None, None, None, E.p(vd), *common, E.p(draw), E.p(grad2), 1, E.p(ws), C.c_size_t(ws.size), None), L)
    assert np.array_equal(grad2, grad)
    # the parameter tensors <-> the packed blocks in one launch each (xr_kilo_pack_params / xr_kilo_unpack_grads) == MultiNetwork.pack / unpack_like
    params = [np.ascontiguousarray(t.detach().numpy()) for t in mn.ordered_parameters()]
    arr = (C.c_void_p * len(params))(*[a.ctypes.data for a in params])
    N, stride = packed.shape
    blocks = E.aligned((N, stride), fill=7.0)
    E.check(L.xr_kilo_pack_params(arr, N, 10, 4, 2, E.p(blocks), stride, None), L)
    assert np.array_equal(blocks, packed)
    outs = [np.full(a.shape, -5.0, np.float32) for a in params]
    oarr = (C.c_void_p * len(outs))(*[a.ctypes.data for a in outs])
    gcopy = E.aligned(packed.shape, fill=grad)
    E.check(L.xr_kilo_unpack_grads(E.p(gcopy), stride, N, 10, 4, 2, oarr, 1, None), L)
    for a, t in zip(outs, got):
        assert np.array_equal(a, t.numpy())
    assert not gcopy.any()                                        # left zero-filled for the next accumulation
    # NerfRender on the reference's raw
    rgb, disp, acc, w = np.zeros((R, 3), np.float32), np.zeros(R, np.float32), np.zeros(R, np.float32), np.zeros((R, S), np.float32)
    graw = E.aligned((R, S, 4), fill=gold['raw'])
    E.check(L.xr_nerf_render_forward(E.p(graw), E.p(E.f32(gold['z_vals'])), E.p(E.f32(gold['rays_d'])), R, S, 1, E.p(rgb), E.p(disp), E.p(acc),
                                     E.p(w), None), L)
    assert np.abs(w - gold['weights']).max() <= 2e-6 and np.abs(rgb - gold['rgb']).max() <= 5e-6
    ok = np.isfinite(gold['disp'])
    assert np.array_equal(ok, np.isfinite(disp)) and np.abs(disp[ok] - gold['disp'][ok]).max() <= 1e-5 * max(1.0, np.abs(gold['disp'][ok]).max())


@pytest.mark.parametrize('lindisp', [0, 1])
def test_fused_frame_call_equals_the_dense_path_on_the_host(E, gold, lindisp):
    """xr_kilo_render_rays (z on the fly, per-ray spans, rows without a network neither written nor read) against
    xr_mip_zvals -> xr_kilo_mlp_forward -> xr_nerf_render_forward, bit for bit, incl. axis-parallel and missing rays"""
    L, Lm = E.lib('xr_kilo'), E.lib('xr_mip')
    L.xr_kilo_workspace_bytes.restype = L.xr_kilo_render_workspace_bytes.restype = C.c_size_t
    mn = scene(gold)
    packed = E.aligned(tuple(mn.packed().shape), fill=mn.packed().numpy())
    a = kilo_args(E, gold, packed)
    rng = np.random.default_rng(3)
    R, S = 70, 150
    o = rng.normal(0, 1, (R, 3)); o = 3.2 * o / np.linalg.norm(o, axis=-1, keepdims=True)
    d = rng.uniform(-0.9, 0.9, (R, 3)) - o
    o[:6] = [[-4, 0.2, 0.1], [0.3, -4, 0.0], [0.1, 0.2, -4], [0.0, 0.0, 0.0], [5, 5, 5], [-4, 1.399, 0.1]]
    d[:6] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [0.3, -0.2, 0.9], [1, 0, 0], [1, 1e-4, 0]]
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o, d = E.f32(o), E.f32(d)
    near, far = np.full(R, 0.5, np.float32), np.full(R, 7.0, np.float32)
    common = (a['gmin'], a['gmax'], a['fixed'], a['res'], E.p(a['occ']), E.p(a['dmins']), E.p(a['dmaxs']), E.p(packed), packed.shape[1],
              24, 10, 4, 2)
    rgb, disp, acc = np.zeros((R, 3), np.float32), np.zeros(R, np.float32), np.zeros(R, np.float32)
    ws = E.aligned((int(L.xr_kilo_render_workspace_bytes(R, S, 24)),), np.uint8, fill=0xA5)      # garbage: unwritten rows must not matter
    E.check(L.xr_kilo_render_rays(E.p(o), E.p(d), E.p(d), E.p(near), E.p(far), R, S, lindisp, *common, 1, E.p(rgb), E.p(disp), E.p(acc), E.p(ws),
                                  C.c_size_t(ws.size), None), L)
    z = np.zeros((R, S), np.float32)
    E.check(Lm.xr_mip_zvals(E.p(near), E.p(far), R, S, lindisp, None, E.p(z), None), Lm)
    raw = E.aligned((R, S, 4))
    ws2 = E.aligned((int(L.xr_kilo_workspace_bytes(C.c_uint64(R * S), 24)),), np.uint8)
    E.check(L.xr_kilo_mlp_forward(None, E.p(o), E.p(d), E.p(z), E.p(d), R, S, *common, E.p(raw), None, E.p(ws2), C.c_size_t(ws2.size), None), L)
    rgb2, disp2, acc2, w2 = np.zeros((R, 3), np.float32), np.zeros(R, np.float32), np.zeros(R, np.float32), np.zeros((R, S), np.float32)
    E.check(L.xr_nerf_render_forward(E.p(raw), E.p(z), E.p(d), R, S, 1, E.p(rgb2), E.p(disp2), E.p(acc2), E.p(w2), None), L)
    assert (acc2 > 0).mean() > (0.2 if not lindisp else 0.05)          # sampling in disparity puts fewer samples in the box
    assert np.array_equal(rgb, rgb2) and np.array_equal(acc, acc2) and np.array_equal(np.nan_to_num(disp, nan=-1), np.nan_to_num(disp2, nan=-1))


@pytest.fixture(params=['split2', 'bf16x3', 'bf16x3all', 'mfma'])
def gemm_kernel(request, monkeypatch):
    """the kernels behind the linear entry points: fp32 results on the bf16 MFMA (3-way operand split) for the forward product only (default) / for all three products / fp32 MFMA throughout"""
    monkeypatch.setenv('XR_GEMM_F32', request.param)
    return request.param


@pytest.mark.parametrize('M,N,K', [(300, 256, 96), (257, 128, 284), (130, 260, 256), (5, 4, 8), (1000, 132, 36), (200, 384, 256), (321, 512, 132)])
def test_linear_kernels_on_the_host(E, M, N, K, gemm_kernel):
    L = E.lib('xr_gemm')
    rng = np.random.default_rng(M + N + K)
    x, w, b = E.aligned((M, K), fill=rng.normal(0, 1, (M, K))), E.aligned((N, K), fill=rng.normal(0, 1, (N, K)) / K ** 0.5), E.f32(rng.normal(0, 1, N))
    dy = E.aligned((M, N), fill=rng.normal(0, 1, (M, N)))
    y = E.aligned((M, N))
    E.check(L.xr_linear_forward(E.p(x), 0, E.p(w), E.p(b), M, N, K, 1, E.p(y), 0, None), L)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert np.abs(y - np.maximum(ref, 0)).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    dym = dy.astype(np.float64) * (y > 0)
    dx = E.aligned((M, K))
    E.check(L.xr_linear_backward_input(E.p(dy), 0, E.p(y), E.p(w), 0, M, N, K, E.p(dx), None), L)
    assert np.abs(dx - dym @ w.astype(np.float64)).max() <= 5e-5 * max(1.0, np.abs(dym @ w).max())
    splits = int(L.xr_linear_backward_splits(M, N, K))
    part = E.aligned((splits, N, K))
    E.check(L.xr_linear_backward_weight(E.p(dy), 0, E.p(y), E.p(x), 0, M, N, K, splits, E.p(part), None, C.c_size_t(0), None), L)
    rw = dym.T @ x.astype(np.float64)
    assert np.abs(part.sum(0) - rw).max() <= 5e-5 * max(1.0, np.abs(rw).max())
    bs = int(L.xr_linear_backward_splits(M, 0, 0))
    pb = E.aligned((bs, N))
    E.check(L.xr_linear_backward_bias(E.p(dy), E.p(y), M, N, bs, E.p(pb), None), L)
    assert np.abs(pb.sum(0) - dym.sum(0)).max() <= 5e-5 * max(1.0, np.abs(dym.sum(0)).max())
    # the input gradient with the weight handed over transposed (the forward's kernel), and weight + bias gradient in one launch
    wt = E.aligned((K, N), fill=np.ascontiguousarray(w.T))
    dx2 = E.aligned((M, K))
    E.check(L.xr_linear_backward_input(E.p(dy), 0, E.p(y), E.p(wt), 1, M, N, K, E.p(dx2), None), L)
    assert np.abs(dx2 - dym @ w.astype(np.float64)).max() <= 5e-5 * max(1.0, np.abs(dym @ w).max())
    part2, pb2 = E.aligned((splits, N, K)), E.aligned((splits, N))
    E.check(L.xr_linear_backward_weight(E.p(dy), 0, E.p(y), E.p(x), 0, M, N, K, splits, E.p(part2), E.p(pb2), C.c_size_t(0), None), L)
    assert np.array_equal(part2, part)
    assert np.abs(pb2.sum(0) - dym.sum(0)).max() <= 5e-5 * max(1.0, np.abs(dym.sum(0)).max())
    # ---- row strides: the same operands as column ranges of wider buffers (the skip connection's [x | h], the view layer's input)
    # give the same bits -- input x at columns [8, 8 + K) of a [M, K + 12] buffer, output / mask / dy at columns [4, 4 + N) of [M, N + 8]
    if True:
        xw = E.aligned((M, K + 12), fill=7.0); xw[:, 8:8 + K] = x
        yw = E.aligned((M, N + 8), fill=-3.0)
        voidp = lambda a, col: C.c_void_p(a.ctypes.data + 4 * col)
        E.check(L.xr_linear_forward(voidp(xw, 8), K + 12, E.p(w), E.p(b), M, N, K, 1, voidp(yw, 4), N + 8, None), L)
        assert np.array_equal(yw[:, 4:4 + N], y) and np.all(yw[:, :4] == -3.0) and np.all(yw[:, 4 + N:] == -3.0)
        dyw = E.aligned((M, N + 8), fill=5.0); dyw[:, 4:4 + N] = dy
        dx3 = E.aligned((M, K))
        E.check(L.xr_linear_backward_input(voidp(dyw, 4), N + 8, voidp(yw, 4), E.p(wt), 1, M, N, K, E.p(dx3), None), L)
        assert np.array_equal(dx3, dx2)
        # weight and bias partials in ONE [splits, N K + N] buffer
        joint = E.aligned((splits, N * K + N))
        E.check(L.xr_linear_backward_weight(voidp(dyw, 4), N + 8, voidp(yw, 4), voidp(xw, 8), K + 12, M, N, K, splits, E.p(joint),
                                            C.c_void_p(joint.ctypes.data + 4 * N * K), C.c_size_t(N * K + N), None), L)
        assert np.array_equal(joint[:, :N * K].reshape(splits, N, K), part2) and np.array_equal(joint[:, N * K:], pb2)
