"""`HashNerfMLP`: hash-grid + SH-4 + two tiny MLPs, the registered type of
/root/reference/xrnerf/models/mlps/hashnerf_mlp.py:23-111, on the MI355X kernels.

The reference builds four tinycudann modules (`embedder_pos`, `embedder_dir`, `density_net`,
`color_net`), each exposing ONE flat fp32 `params` tensor; the same sub-module and parameter
names are kept so `state_dict()` keys match (`mlp.embedder_pos.params`, ...).  tinycudann itself is
not used: encode -> fused MLP runs through libxrnerf_mi355.so.
"""
import os

import numpy as np
import torch
from torch import nn

from . import ops
from .builder import MLPS


def get_per_level_scale(bound):
    # hashnerf_mlp.py:17-20 (the reference passes the literal 1, not `bound`, at :35)
    return float(np.exp2(np.log2(2048 * bound / 16) / (16 - 1)))


def _hidden_layers(network_config):
    """tcnn reads `n_hidden_layers`; the reference config writes `num_layers`
    (configs/instant_ngp/nerf_blender_local01.py:106-124).  Honour n_hidden_layers, else num_layers
    (author intent = Instant-NGP paper); XRNERF_TCNN_STRICT_DEFAULTS=1 emulates tcnn silently
    ignoring the unknown key (its default is 5).  SURVEY.md section 2c."""
    if 'n_hidden_layers' in network_config:
        return int(network_config['n_hidden_layers'])
    if os.environ.get('XRNERF_TCNN_STRICT_DEFAULTS') == '1':
        return 5
    return int(network_config.get('num_layers', 2))


class HashGridEncoding(nn.Module):
    """tcnn.Encoding(otype='HashGrid') stand-in: owns the table as one flat `params`."""

    def __init__(self, n_input_dims=3, encoding_config=None):
        super().__init__()
        c = dict(encoding_config or {})
        assert n_input_dims == 3 and c.get('otype', 'HashGrid') == 'HashGrid'
        assert int(c.get('n_features_per_level', 2)) == 2 and c.get('interpolation', 'Linear') == 'Linear'
        self.meta = ops.GridMeta(int(c.get('n_levels', 16)), int(c.get('log2_hashmap_size', 19)),
                                 int(c.get('base_resolution', 16)), float(c.get('per_level_scale', 2.0)))
        self.n_output_dims = self.meta.n_output_dims
        p = torch.empty(self.meta.n_params, dtype=torch.float32).uniform_(-1e-4, 1e-4)  # tcnn grid init
        self.params = nn.Parameter(p)


class SHEncoding(nn.Module):
    """tcnn.Encoding(otype='SphericalHarmonics', degree=4): no parameters (empty `params`)."""

    def __init__(self, n_input_dims=3, encoding_config=None):
        super().__init__()
        c = dict(encoding_config or {})
        assert n_input_dims == 3 and c.get('otype') == 'SphericalHarmonics' and int(c.get('degree', 4)) == 4
        self.n_output_dims = 16
        self.params = nn.Parameter(torch.zeros(0, dtype=torch.float32))

    def forward(self, dirs):
        return ops.sh4(dirs)


class FusedMLPParams(nn.Module):
    """tcnn.Network(otype='FullyFusedMLP') stand-in: bias-free, ReLU hidden, linear out, width 64,
    flat `params` = row-major [out,in] matrices in layer order, in/out padded to 16."""

    def __init__(self, n_input_dims, n_output_dims, network_config=None):
        super().__init__()
        c = dict(network_config or {})
        assert c.get('otype', 'FullyFusedMLP') == 'FullyFusedMLP' and c.get('activation', 'ReLU') == 'ReLU'
        assert c.get('output_activation', 'None') == 'None' and int(c.get('n_neurons', 64)) == 64
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.in_pad = (self.n_input_dims + 15) // 16 * 16
        self.out_pad = (self.n_output_dims + 15) // 16 * 16
        self.width = 64
        self.n_hidden = _hidden_layers(c)
        dims = [self.in_pad] + [self.width] * self.n_hidden + [self.out_pad]
        ws = []
        for a, b in zip(dims[:-1], dims[1:]):   # tcnn: xavier uniform per matrix
            lim = float(np.sqrt(6.0 / (a + b)))
            ws.append(torch.empty(b * a, dtype=torch.float32).uniform_(-lim, lim))
        self.params = nn.Parameter(torch.cat(ws))


class _NerfMLPFn(torch.autograd.Function):
    """encode -> density_net -> (SH, color_net) -> raw [n,4]; backward recomputes activations."""

    @staticmethod
    def forward(ctx, table, wd, wc, pts, dirs, mlp, n_dev, planes=None):
        n = pts.shape[0]
        # `planes` [3, n]: the same positions as three planes when the sampler has them (coalesced loads in the gather)
        enc_t = ops.hashgrid_fwd(table, planes if planes is not None else pts, mlp.embedder_pos.meta, n_dev=n_dev)
        raw = ops.nerf_mlp_fwd(enc_t, dirs, n, wd, wc, mlp.density_net.n_hidden, mlp.color_net.n_hidden, mlp.pad_value,
                               n_dev=n_dev)
        ctx.save_for_backward(table, wd, wc, pts, dirs, enc_t)
        ctx.mlp, ctx.n_dev = mlp, n_dev
        return raw

    @staticmethod
    def backward(ctx, draw):
        table, wd, wc, pts, dirs, enc_t = ctx.saved_tensors
        mlp = ctx.mlp
        n = pts.shape[0]
        draw = draw.contiguous()
        g_wd, g_wc = torch.zeros_like(wd), torch.zeros_like(wc)
        denc_t = ops.nerf_mlp_bwd(enc_t, dirs, n, wd, wc, mlp.density_net.n_hidden, mlp.color_net.n_hidden, draw,
                                  g_wd, g_wc, mlp.pad_value, n_dev=ctx.n_dev)
        g_table = torch.zeros_like(table)
        ops.hashgrid_bwd(pts, denc_t, mlp.embedder_pos.meta, g_table, n_dev=ctx.n_dev)
        return g_table, g_wd, g_wc, None, None, None, None, None


@MLPS.register_module()
class HashNerfMLP(nn.Module):
    def __init__(self, bound=1, embedder_pos=None, embedder_dir=None, density_net=None, color_net=None, **kwarg):
        super().__init__()
        embedder_pos = {k: (dict(v) if isinstance(v, dict) else v) for k, v in dict(embedder_pos).items()}
        embedder_pos['encoding_config']['per_level_scale'] = get_per_level_scale(1)   # hashnerf_mlp.py:34-35
        self.embedder_pos = HashGridEncoding(**embedder_pos)
        self.embedder_dir = SHEncoding(**dict(embedder_dir))
        density_net = dict(density_net)
        density_net['n_input_dims'] = self.embedder_pos.n_output_dims
        self.density_net = FusedMLPParams(**density_net)
        color_net = dict(color_net)
        color_net['n_input_dims'] = self.embedder_dir.n_output_dims + density_net['n_output_dims'] - 1   # :43-44
        self.color_net = FusedMLPParams(**color_net)
        assert self.embedder_pos.n_output_dims == 32 and self.density_net.out_pad == 16 and self.color_net.in_pad == 32
        # tcnn pads the (Identity-encoded) 31-wide color input to 32 with ones
        self.pad_value = 1.0

    def forward(self, data):
        unflatten_shape = data['pts'].shape[:-1]
        outputs_flat = self.run_mlp(data)
        data['raw'] = torch.reshape(outputs_flat, list(unflatten_shape) + [outputs_flat.shape[-1]])
        return data

    @staticmethod
    def _rows(x):
        """[..., 3] fp32 -> 2-D view with unit column stride (column slices of [S,7] rows stay in place)."""
        x = x.detach()
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1])
        if x.dtype != torch.float32:
            x = x.float()
        if x.stride(1) != 1 or x.stride(0) < 3:
            x = x.contiguous()
        return x

    def run_mlp(self, data):
        pts = self._rows(data['pts'])
        viewdirs = data['viewdirs']
        if len(data['pts'].shape) > len(viewdirs.shape):
            viewdirs = viewdirs[:, None].expand(data['pts'].shape)
        dirs = self._rows(viewdirs)
        if pts.shape[0] == 0:
            return torch.zeros((0, 4), dtype=torch.float32, device=pts.device)
        # `n_valid_dev` (optional, device int32[1]): rows past it are padding of a fixed-size sample buffer
        # (the reference pads its compacted buffer to target_batch_size rows with zeros and evaluates them)
        planes = data.get('pts_planes')
        if planes is not None and (planes.dim() != 2 or planes.shape[0] != 3 or planes.shape[1] != pts.shape[0] or pts.shape[0] == 3):
            planes = None
        if not torch.is_grad_enabled() and ops._on_device(pts) and torch.is_tensor(pts) and pts.is_cuda:
            # inference (frames): no autograd node; the encoded features (consumed inside this call) live in a grow-only persistent
            # buffer (ops._buf).  The RESULT is a fresh tensor unless the caller opts in with data['reuse_buffers'] (HashNerfNetwork's
            # own forward does: its renderer consumes `raw` right away): then it is a view of this module's persistent buffer, valid
            # until this module's next such call -- a caller that keeps two results would otherwise get aliased data.
            n = pts.shape[0]
            ld = (n + 63) // 64 * 64
            nhd, nhc = self.density_net.n_hidden, self.color_net.n_hidden
            n_dev = data.get('n_valid_dev')
            enc_t = ops._buf(pts.device, (self.embedder_pos.meta.n_output_dims, ld), 'infer_enc')
            raw = ops._buf(pts.device, (n, 4), 'infer_raw_%x' % id(self)) if data.get('reuse_buffers') else \
                torch.empty((n, 4), dtype=torch.float32, device=pts.device)
            ops.hashgrid_fwd(self.embedder_pos.params.detach(), planes if planes is not None else pts, self.embedder_pos.meta, enc_t=enc_t,
                             ld=ld, n_dev=n_dev)
            return ops.nerf_mlp_fwd(enc_t, dirs, n, self.density_net.params.detach(), self.color_net.params.detach(), nhd, nhc,
                                    self.pad_value, raw=raw, n_dev=n_dev)
        return _NerfMLPFn.apply(self.embedder_pos.params, self.density_net.params, self.color_net.params, pts, dirs,
                                self, data.get('n_valid_dev'), planes)

    def run_density_planes(self, planes):
        """run_density for positions stored as three planes [3, n] (the sampler's grid refresh): -> [n,1] view, row stride 4, of this
        module's persistent buffer: valid until this module's next call of this method"""
        n = planes.shape[1]
        with torch.no_grad():
            enc_t = ops._buf(planes.device, (self.embedder_pos.meta.n_output_dims, (n + 63) // 64 * 64), 'density_enc')
            raw = ops._buf(planes.device, (n, 4), 'density_raw_%x' % id(self))       # (consumed by the caller's splat launch before the next query)
            ops.hashgrid_fwd(self.embedder_pos.params.detach(), planes, self.embedder_pos.meta, enc_t=enc_t, ld=enc_t.shape[1])
            ops.nerf_mlp_fwd(enc_t, None, n, self.density_net.params.detach(), None, self.density_net.n_hidden, self.color_net.n_hidden,
                             self.pad_value, raw=raw)
        return raw[:, 3:4]

    def density_splat_planes(self, planes, indices, grid_tmp):
        """the grid refresh's density query with K8 inside (ops.nerf_density_splat): positions as planes [3, n], each point's optical
        thickness merged into grid_tmp[indices[i]]; False when this topology has no fused forward (the caller then queries and splats)"""
        nhd, nhc = self.density_net.n_hidden, self.color_net.n_hidden
        if not ops.density_splat_supported(nhd, nhc):
            return False
        n = planes.shape[1]
        with torch.no_grad():
            enc_t = ops._buf(planes.device, (self.embedder_pos.meta.n_output_dims, (n + 63) // 64 * 64), 'density_enc')
            ops.hashgrid_fwd(self.embedder_pos.params.detach(), planes, self.embedder_pos.meta, enc_t=enc_t, ld=enc_t.shape[1])
            ops.nerf_density_splat(enc_t, n, self.density_net.params.detach(), nhd, nhc, indices, grid_tmp)
        return True

    def run_density(self, pts_flat):
        """hashnerf_mlp.py:107-111: encode + density_net, channel 0 -> [N,1] fp32 (no grad)."""
        pts = self._rows(pts_flat)
        n = pts.shape[0]
        with torch.no_grad():
            enc_t = ops.hashgrid_fwd(self.embedder_pos.params, pts, self.embedder_pos.meta)
            raw = ops.nerf_mlp_fwd(enc_t, None, n, self.density_net.params, None, self.density_net.n_hidden,
                                   self.color_net.n_hidden, self.pad_value)
        return raw[:, 3:4]
