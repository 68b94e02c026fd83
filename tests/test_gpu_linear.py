"""fp32-MFMA linear layers (xr_linear_*) against fp64 matmuls of the same operands: forward with bias / relu, input
gradient and weight gradient with the relu mask, ragged M / N / K (partial tiles, multi-split weight gradient), and the
autograd node against torch's own linear + relu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['split2', 'bf16x3', 'bf16x3all', 'mfma'])
def gemm_kernel(request, monkeypatch):
    """the kernels behind the linear entry points: the row-major products on 2-way split operands (fp16 parts forward, bf16 parts for gradients: the default) / on exact 3-way bf16 operands / that for all three products / fp32 MFMA throughout"""
    monkeypatch.setenv('XR_GEMM_F32', request.param)
    return request.param


@pytest.mark.parametrize('M,N,K', [(1000, 256, 96), (4096, 256, 352), (333, 128, 256), (70000, 256, 256), (5, 4, 8), (129, 132, 36)])
def test_kernels_against_fp64(dev, M, N, K, gemm_kernel):
    from xrnerf_amd import ops
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    ref = x.double() @ w.double().t() + b.double()
    y = ops.linear_forward(xd, wd, bd, False).cpu()
    assert (y.double() - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    yr = ops.linear_forward(xd, wd, bd, True)
    assert (yr.cpu().double() - ref.clamp_min(0)).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert (ops.linear_forward(xd, wd, None, False).cpu().double() - (ref - b.double())).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    dym = dy.double() * (ref > 0)
    # mask from the kernel's own relu output (elements within rounding of 0 may differ from the fp64 sign: exclude them)
    safe = (ref.abs() > 1e-4)
    dym_k = dy.double() * (yr.cpu() > 0)
    dx = ops.linear_backward_input(dyd, yr, wd).cpu().double()
    assert (dx - dym_k @ w.double()).abs().max() <= 5e-5 * max(1.0, float((dym_k @ w.double()).abs().max()))
    dw = ops.linear_backward_weight(dyd, yr, xd).cpu().double()
    rw = dym_k.t() @ x.double()
    assert (dw - rw).abs().max() <= 5e-5 * max(1.0, float(rw.abs().max()))
    assert ((yr.cpu() > 0) == (ref > 0))[safe].all()
    db = ops.linear_backward_bias(dyd, yr).cpu().double()
    assert (db - dym_k.sum(0)).abs().max() <= 5e-5 * max(1.0, float(dym_k.sum(0).abs().max()))
    assert (ops.linear_backward_bias(dyd, None).cpu().double() - dy.double().sum(0)).abs().max() <= 5e-5 * max(1.0, float(dy.double().sum(0).abs().max()))
    # weight and bias gradient from one launch (the column sums ride on the weight-gradient product's operand panels)
    dw2, db2 = ops.linear_backward_weight_bias(dyd, yr, xd)
    assert torch.equal(dw2.cpu().double(), dw)
    assert (db2.cpu().double() - dym_k.sum(0)).abs().max() <= 5e-5 * max(1.0, float(dym_k.sum(0).abs().max()))
    # no mask
    dx0 = ops.linear_backward_input(dyd, None, wd).cpu().double()
    assert (dx0 - dy.double() @ w.double()).abs().max() <= 5e-5 * max(1.0, float((dy.double() @ w.double()).abs().max()))


def test_autograd_node_equals_torch_linear(dev):
    from xrnerf_amd.linear import linear_act
    torch.manual_seed(0)
    x = torch.randn(3000, 352, device=dev, requires_grad=True)
    w = (torch.randn(256, 352, device=dev) / 18).requires_grad_(True)
    b = torch.randn(256, device=dev, requires_grad=True)
    g = torch.randn(3000, 256, device=dev)
    y = linear_act(x, w, b, True)
    y.backward(g)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    y2 = torch.relu(torch.nn.functional.linear(x, w, b))
    y2.backward(g)
    assert (y - y2).abs().max() <= 1e-4
    for a, r in zip(got, (x.grad, w.grad, b.grad)):
        assert (a - r).abs().max() <= 1e-4 * max(1.0, float(r.abs().max()))
    # shapes the kernel does not take fall through to torch (283-wide view layer, 3-wide head)
    x2 = torch.randn(100, 283, device=dev)
    w2 = torch.randn(128, 283, device=dev)
    assert torch.equal(linear_act(x2, w2, None, True), torch.relu(torch.nn.functional.linear(x2, w2)))


@pytest.mark.parametrize('M,padded_rows', [(3000, True), (257, False)])
def test_whole_mlp_as_one_autograd_node_equals_the_layer_by_layer_graph(dev, M, padded_rows):
    """NerfMLP.run_mlp on the device is ONE autograd node (vanilla._NerfMlpFn): the skip layer writes into the [x | h] buffer, feature and
    alpha heads land in the view layer's input buffer, gradients come from column ranges of the next layer's input gradient -- no cat /
    split / pad kernels.  Output and every parameter gradient against the same module evaluated layer by layer in float64 on the host
    (the reference's graph, nerf_mlp.py:62-94), for the Mip-NeRF widths (96 + 27 inputs, skip at 4)."""
    import copy
    from xrnerf_amd.vanilla import NerfMLP
    torch.manual_seed(5)
    emb = dict(type='MipNerfEmbedder', min_deg_point=0, max_deg_point=16, min_deg_view=0, max_deg_view=4, use_viewdirs=True, append_identity=True)
    import xrnerf_amd.mip  # noqa: F401  (registers the embedder)
    mlp = NerfMLP(skips=[4], netdepth=8, netwidth=256, use_viewdirs=True, embedder=emb).to(dev)
    assert (mlp.input_ch, mlp.input_ch_dirs) == (96, 27)
    ref = copy.deepcopy(mlp).double().cpu()
    x64 = torch.randn(M, 123, dtype=torch.float64)
    if padded_rows:                                          # what ops.mip_encode hands over: rows padded to 124 floats
        x = torch.empty((M, 124), device=dev)[:, :123]
        x.copy_(x64.float())
    else:
        x = x64.float().to(dev)                              # 123-float rows: one aligned copy inside the node
    assert mlp._device_graph_ok(x)
    g = torch.randn(M, 4, dtype=torch.float64)
    out = mlp.run_mlp(x)
    assert out.grad_fn is not None and type(out.grad_fn).__name__ == '_NerfMlpFnBackward'
    out.backward(g.float().to(dev))
    want = ref.run_mlp(x64)
    want.backward(g)
    assert (out.detach().cpu().double() - want.detach()).abs().max() <= 1e-4 * max(1.0, float(want.abs().max()))
    # (gradients: conftest.grad_close -- a hidden unit whose pre-activation lies within the forward's rounding of zero may sit on the other
    # side of its ReLU than in the float64 graph; 6 M pre-activations here, a handful do, each moving single entries by one sample's term)
    from conftest import grad_close
    for (name, p), (_, q) in zip(mlp.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, name
        grad_close(p.grad.cpu().numpy(), q.grad.numpy(), name, kinks=True)


def test_vanilla_nerf_config1_on_the_device_equals_the_host_path(dev):
    """BASELINE config #1's model dict (8x256 MLPs, 64 coarse + 128 fine samples): the device path (fp32-MFMA linear
    layers incl. the padded narrow heads and the shared alpha/feature product, fused NerfRender at inference) against the
    pure-PyTorch host path with the same weights -- test-mode outputs and one training step's loss and gradients"""
    import copy
    import json
    import os
    import xrnerf_amd
    from xrnerf_amd import vanilla
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    cfg = json.load(open(os.path.join(G, 'ngp_model_cfg.json')))['vanilla_model']
    torch.manual_seed(0)
    host = xrnerf_amd.build_network(cfg)
    device = copy.deepcopy(host).to(dev)
    n = 512
    rays_o = torch.tensor([[0., 0., 4.]]).repeat(n, 1) + torch.randn(n, 3) * 0.05
    rays_d = torch.nn.functional.normalize(torch.randn(n, 3) * 0.15 - torch.tensor([0., 0., 1.]), dim=-1)
    z = vanilla.get_z_vals(rays_o, 2., 6., 64)
    tgt = torch.rand(n, 3)

    def data(d):
        return {'rays_o': rays_o.to(d), 'rays_d': rays_d.to(d), 'viewdirs': rays_d.to(d), 'z_vals': z.to(d),
                'pts': vanilla.get_pts(rays_o, rays_d, z).to(d)}
    with torch.no_grad():
        a, b = host.forward(data('cpu'), is_test=True), device.forward(data(dev), is_test=True)
    for k in ('rgb', 'coarse_rgb', 'acc', 'coarse_acc'):
        assert (a[k] - b[k].cpu()).abs().max() <= 1e-4, k
    host.is_perturb = device.is_perturb = False                      # hierarchical sampling without random draws
    outs = []
    for net, d in ((host, 'cpu'), (device, dev)):
        batch = {k: v[None] for k, v in data(d).items()}
        batch['target_s'] = tgt.to(d)[None]
        out = net.train_step(batch, None)
        out['loss'].backward()
        outs.append(float(out['loss']))
    assert abs(outs[0] - outs[1]) <= 1e-5 * max(1.0, abs(outs[0]))
    hp, dp = dict(host.named_parameters()), dict(device.named_parameters())
    for name in ('mlp.pts_linears.0.weight', 'mlp.pts_linears.5.weight', 'mlp.alpha_linear.weight', 'mlp.alpha_linear.bias',
                 'mlp.views_linears.0.weight', 'mlp.rgb_linear.weight', 'mlp.rgb_linear.bias', 'mlp_fine.feature_linear.weight',
                 'mlp_fine.pts_linears.7.bias'):
        r, g = hp[name].grad, dp[name].grad.cpu()
        # bias / weight gradients are sums of ~1e5 signed terms: the summation order shows at the 1e-6 level
        assert (r - g).abs().max() <= 1e-3 * float(r.abs().max()) + 3e-6, name


@pytest.mark.parametrize('nb,n,stride', [(1, 100, 100), (7, 65, 80), (256, 65792, 65792), (300, 4097, 4100)])
def test_sum_partials_is_the_fixed_order_sum(dev, nb, n, stride):
    """xr_sum_partials (what finishes the split weight / bias gradients): out[j] = sum_b partials[b][j] for j < n, columns behind n untouched
    by the read, the same bits every launch, within fp32 summation error of the float64 sum"""
    from xrnerf_amd import ops
    torch.manual_seed(nb + n)
    buf = torch.randn((nb, stride), dtype=torch.float32, device=dev)
    buf[:, n:] = float('nan')                                  # padding columns must never be read
    part = buf[:, :n] if stride > n else buf
    outs = []
    for _ in range(2):
        out = torch.empty((n,), dtype=torch.float32, device=dev)
        from xrnerf_amd import _lib
        _lib.check(_lib.load().xr_sum_partials(ops._ptr(buf), nb, stride, n, ops._ptr(out), ops._stream()), 'xr_sum_partials')
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    want = part.double().sum(0)
    assert float((outs[0].double() - want).abs().max()) <= 1e-6 * nb ** 0.5 * max(1.0, float(part.abs().max()))
    if stride == n:
        assert torch.equal(ops.sum_partials(buf), outs[0])
