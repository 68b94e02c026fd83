"""`tinycudann`-shaped module surface: `Encoding(n_input_dims, encoding_config)` and
`Network(n_input_dims, n_output_dims, network_config)` as `nn.Module`s with one flat fp32 `params` each and an
`n_output_dims` attribute -- exactly what /root/reference/xrnerf/models/mlps/hashnerf_mlp.py:12,36-45,55-111 uses
of the (un-vendored) tiny-cuda-nn bindings, so that file runs unchanged with

    import sys, xrnerf_amd.tcnn
    sys.modules['tinycudann'] = xrnerf_amd.tcnn

Row-major [N, C] tensors in and out, autograd through the HIP kernels.  This is the compatibility surface (it
pays a transpose to / from the feature-major layout the kernels like); `xrnerf_amd.mlps.HashNerfMLP` is the
fused fast path.  fp32 parameters and outputs (tcnn would hand back fp16).
"""
import ctypes as C

import torch
from torch import nn

from . import _lib, ops
from .mlps import _hidden_layers


class _GridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, x, meta):
        x = x.detach()
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            x = x.float().contiguous()
        enc_t = ops.hashgrid_fwd(params, x, meta)
        ctx.save_for_backward(x)
        ctx.meta, ctx.ld = meta, enc_t.shape[1]
        return enc_t[:, :x.shape[0]].t()             # [N, 2L] view of the feature-major buffer

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n = x.shape[0]
        denc_t = torch.zeros((ctx.meta.n_output_dims, ctx.ld), dtype=torch.float32, device=x.device)
        denc_t[:, :n] = dy.t()
        g = torch.zeros(ctx.meta.n_params, dtype=torch.float32, device=x.device)
        ops.hashgrid_bwd(x, denc_t, ctx.meta, g)
        return g, None, None


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        super().__init__()
        c = dict(encoding_config)
        self.n_input_dims = n_input_dims
        self.otype = c.get('otype')
        if self.otype == 'HashGrid':
            assert n_input_dims == 3 and int(c.get('n_features_per_level', 2)) == 2
            self.meta = ops.GridMeta(int(c.get('n_levels', 16)), int(c.get('log2_hashmap_size', 19)),
                                     int(c.get('base_resolution', 16)), float(c.get('per_level_scale', 2.0)))
            self.n_output_dims = self.meta.n_output_dims
            g = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter(torch.empty(self.meta.n_params).uniform_(-1e-4, 1e-4, generator=g))
        elif self.otype == 'SphericalHarmonics':
            assert n_input_dims == 3 and int(c.get('degree', 4)) == 4
            self.n_output_dims = 16
            self.params = nn.Parameter(torch.zeros(0))
        else:
            raise NotImplementedError('encoding otype %r (HashGrid and SphericalHarmonics are on the Instant-NGP path)' % self.otype)

    def forward(self, x):
        if self.otype == 'HashGrid':
            return _GridFn.apply(self.params, x, self.meta)
        x = x.detach()
        return ops.sh4(x if (x.dtype == torch.float32 and x.stride(-1) == 1) else x.float().contiguous())


def _mlp_call(fn, *a):
    _lib.check(fn(*a), fn.__name__)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, x, net):
        L = _lib.load()
        x = x if x.dtype == torch.float32 else x.float()
        n = x.shape[0]
        y = torch.empty((n, 16), dtype=torch.float32, device=x.device)
        if not ops._on_device(x):
            raise _lib.XrError('xrnerf_amd.tcnn.Network needs ROCm device tensors: there is no CPU fallback')
        _lib.check(L.xr_mlp_fwd(C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), net.n_input_dims, net.pad_value, n,
                                ops._ptr(params), net.n_hidden, ops._ptr(y), ops._stream()), 'xr_mlp_fwd')
        ctx.save_for_backward(params, x)
        ctx.net = net
        return y[:, :net.n_output_dims]

    @staticmethod
    def backward(ctx, dy):
        L = _lib.load()
        params, x = ctx.saved_tensors
        net, n = ctx.net, x.shape[0]
        dyp = torch.zeros((n, 16), dtype=torch.float32, device=x.device)
        dyp[:, :net.n_output_dims] = dy
        dx = torch.empty((n, net.n_input_dims), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        gw = torch.zeros_like(params)
        ws = ops._ws(x.device, L.xr_mlp_bwd_workspace_bytes(net.n_hidden), 'mlp_generic')
        _lib.check(L.xr_mlp_bwd(C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), net.n_input_dims, net.pad_value, n,
                                ops._ptr(params), net.n_hidden, ops._ptr(dyp), ops._ptr(dx), ops._ptr(gw), ops._ptr(ws),
                                ws.numel(), ops._stream()), 'xr_mlp_bwd')
        return gw, dx, None


class Network(nn.Module):
    """FullyFusedMLP: bias-free, 64 wide, ReLU hidden, linear output; input padded to 32 with ones (tcnn pads its
    Identity-encoded input with 1), output padded to 16."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        c = dict(network_config)
        assert c.get('otype', 'FullyFusedMLP') == 'FullyFusedMLP' and c.get('activation', 'ReLU') == 'ReLU'
        assert c.get('output_activation', 'None') == 'None' and int(c.get('n_neurons', 64)) == 64
        assert 1 <= n_input_dims <= 32 and 1 <= n_output_dims <= 16
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.n_hidden = _hidden_layers(c)
        self.pad_value = 1.0
        dims = [32] + [64] * self.n_hidden + [16]
        g = torch.Generator().manual_seed(seed)
        ws = []
        for a, b in zip(dims[:-1], dims[1:]):
            lim = (6.0 / (a + b)) ** 0.5
            ws.append(torch.empty(b * a).uniform_(-lim, lim, generator=g))
        self.params = nn.Parameter(torch.cat(ws))

    def forward(self, x):
        if self.n_hidden > 2:
            # deeper than the fused single-network kernels (forward 1..3 hidden layers, backward 1..2): layer by layer on the
            # linear kernels (ops._layered_net) -- tiny-cuda-nn's own default depth, 5, takes this path
            x = x if x.dtype == torch.float32 else x.float()
            if x.shape[1] < 32:
                x = torch.cat([x, torch.full((x.shape[0], 32 - x.shape[1]), self.pad_value, dtype=torch.float32, device=x.device)], 1)
            return ops._layered_net(x.contiguous(), self.params, self.n_hidden)[:, :self.n_output_dims]
        return _NetFn.apply(self.params, x, self)
