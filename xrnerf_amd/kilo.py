"""Registry entries of the KiloNeRF rendering path (BASELINE config #5, configs/kilonerf/kilonerf_finetune_*.py):
`KiloNerfNetwork`, `KiloNerfMLP`, `KiloNerfFourierEmbedder` and the `MultiNetwork` parameter container, with the
constructor signatures, `data` dict keys and parameter names / layouts of
  /root/reference/xrnerf/models/networks/kilonerf.py:17-154, mlps/kilonerf_mlp.py:27-190,
  mlps/multi_modules.py:238-392,405-668 (MultiNetworkLinear / MultiNetwork, `multimatmul` weight layout [N, in, out]),
  embedders/kilonerf_fourier_embedder.py:60-102.

`KiloNerfMLP.forward` is ONE call into the C-ABI (xr_kilo_mlp_forward, xrnerf_amd/csrc/xr_kilo.hip): the reference's
reorder_points_and_dirs + kilonerf_cuda.global_to_local + compute_fourier_features + six MAGMA grouped GEMMs + two
scatters; fine-tuning gradients (the reference's AddMultiMatMul.backward chain) are one more call
(xr_kilo_mlp_backward).  No CPU path: host tensors raise.
"""
import math

import torch
from torch import nn

from . import builder, ops
from .builder import EMBEDDERS, MLPS, NETWORKS
from .vanilla import NerfNetwork


@EMBEDDERS.register_module()
class KiloNerfFourierEmbedder(nn.Module):
    """holds the frequency counts; the features themselves ([x | cos(x 2^k) | sin(x 2^k)] per channel,
    kilonerf_fourier_embedder.py:33-52) are produced in registers inside the MLP kernel"""

    def __init__(self, num_networks, multires=10, multires_dirs=4, input_ch=3, **kwargs):
        super().__init__()
        if input_ch != 3:
            raise NotImplementedError('input_ch must be 3')
        self.num_networks, self.multires, self.multires_dirs, self.input_ch = num_networks, multires, multires_dirs, input_ch
        self.embed_ch = (2 * multires + 1) * input_ch
        self.embed_ch_dirs = (2 * multires_dirs + 1) * input_ch

    def get_embed_ch(self):
        return self.embed_ch, self.embed_ch_dirs

    def forward(self, data, fourier_embedding_implementation='pytorch'):
        raise NotImplementedError('the Fourier features are fused into KiloNerfMLP.forward (xr_kilo_mlp_forward)')


class MultiNetworkLinear(nn.Module):
    """parameters of one layer of all networks in the reference's `multimatmul` layout (multi_modules.py:238-340):
    weight [N, in, out], bias [N, out]; kaiming-uniform(a=sqrt(5)) / fan-in-uniform initialisation"""

    def __init__(self, num_networks, in_features, out_features):
        super().__init__()
        self.num_networks, self.in_features, self.out_features = num_networks, in_features, out_features
        bound = 1.0 / math.sqrt(in_features)                 # gain*sqrt(3/fan_in) with gain = sqrt(2/(1+5)) -> 1/sqrt(fan_in)
        self.weight = nn.Parameter(torch.empty(num_networks, in_features, out_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(num_networks, out_features).uniform_(-bound, bound))


class MultiNetwork(nn.Module):
    """the distilled networks' parameters under the reference's names (multi_modules.py:405-565, late_feed_direction,
    relu): pts_linears.{l}, alpha_linear, feature_linear, direction_layer, rgb_linear"""

    def __init__(self, num_networks, num_position_channels, num_direction_channels, num_output_channels=4,
                 hidden_layer_size=32, num_hidden_layers=2, refeed_position_index=None, late_feed_direction=True,
                 direction_layer_size=32, nonlinearity='relu', **kwargs):
        super().__init__()
        if not late_feed_direction or refeed_position_index is not None or nonlinearity != 'relu':
            raise NotImplementedError('only the late_feed_direction / relu / no-refeed architecture of the reference configs')
        if hidden_layer_size != 32 or direction_layer_size != 32:
            raise NotImplementedError('hidden_layer_size and direction_layer_size must be 32 (every reference config)')
        self.num_networks = num_networks
        self.num_position_channels, self.num_direction_channels = num_position_channels, num_direction_channels
        self.hidden_layer_size, self.num_hidden_layers = hidden_layer_size, num_hidden_layers
        self.direction_layer_size = direction_layer_size
        H = hidden_layer_size
        self.pts_linears = nn.ModuleList([MultiNetworkLinear(num_networks, num_position_channels if l == 0 else H, H)
                                          for l in range(num_hidden_layers)])
        self.alpha_linear = MultiNetworkLinear(num_networks, H, 1)
        self.feature_linear = MultiNetworkLinear(num_networks, H, H)
        self.direction_layer = MultiNetworkLinear(num_networks, num_direction_channels + H, direction_layer_size)
        self.rgb_linear = MultiNetworkLinear(num_networks, direction_layer_size, 3)
        self.view_dependent_parameters = list(self.direction_layer.parameters()) + list(self.rgb_linear.parameters())
        self._packed, self._packed_key = None, None

    def ordered_parameters(self):
        """the parameter tensors in block order: (weight, bias) of pts_linears.*, alpha, feature, direction, rgb"""
        layers = list(self.pts_linears) + [self.alpha_linear, self.feature_linear, self.direction_layer, self.rgb_linear]
        return [t for l in layers for t in (l.weight, l.bias)]

    @staticmethod
    def pack(params):
        """[N, stride] blocks in the kernel's order (xr_kilo.hip: kilo_param_floats) from ordered_parameters()"""
        N = params[0].shape[0]
        z = params[1].new_zeros
        parts = []
        for w, b in zip(params[0:-8:2], params[1:-8:2]):                  # hidden layers
            parts += [w.reshape(N, -1), b]
        aw, ab, fw, fb, dw, db, rw, rb = params[-8:]
        parts += [aw.reshape(N, -1), ab, z((N, 3)), fw.reshape(N, -1), fb, dw.reshape(N, -1), db,
                  torch.cat([rw, z((N, rw.shape[1], 1))], -1).reshape(N, -1), rb, z((N, 1))]
        return torch.cat(parts, 1).to(torch.float32).contiguous()

    @staticmethod
    def unpack_like(blocks, params):
        """inverse of pack for a [N, stride] tensor of the same layout (the gradient blocks): one tensor per parameter"""
        N = params[0].shape[0]
        out, off = [], 0

        def take(n):
            nonlocal off
            t = blocks[:, off:off + n]
            off += n
            return t
        for w, b in zip(params[0:-8:2], params[1:-8:2]):
            out += [take(w[0].numel()).reshape(w.shape), take(b.shape[1]).reshape(b.shape)]
        aw, ab, fw, fb, dw, db, rw, rb = params[-8:]
        out += [take(aw[0].numel()).reshape(aw.shape), take(1).reshape(ab.shape)]
        take(3)
        out += [take(fw[0].numel()).reshape(fw.shape), take(fb.shape[1]).reshape(fb.shape)]
        out += [take(dw[0].numel()).reshape(dw.shape), take(db.shape[1]).reshape(db.shape)]
        out += [take(rw.shape[1] * 4).reshape(N, rw.shape[1], 4)[..., :3].contiguous(), take(3).reshape(rb.shape)]
        return out

    def packed(self):
        """cached pack(ordered_parameters()), rebuilt when a parameter changed"""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and self._packed_key == key:
            return self._packed
        with torch.no_grad():
            self._packed = self.pack(self.ordered_parameters())
        self._packed_key = key
        return self._packed


def load_reference_distilled_checkpoint(path):
    """The reference's distillation checkpoint ({'root_nodes': [Node, ...]}, written by SaveDistillResultsHook,
    core/hooks/save_distill_results_hook.py:380-420) pickles its own classes (xrnerf.utils.data_helper.Node, each leaf
    holding a single-network xrnerf.models.mlps.multi_modules.MultiNetwork).  This reads it WITHOUT that package: every
    `xrnerf.*` class is unpickled as an attribute bag, the tree is walked exactly like KiloNerfMLP.init_mlp
    (kilonerf_mlp.py:46-66: breadth-first, leq_child before gt_child) and the leaves' weights are merged into the
    `multimatmul` layout [N, in, out] (:104-121).  -> dict(domain_mins, domain_maxs, state_dict, num_hidden_layers).
    Like torch.load without weights_only this executes a pickle: only for checkpoints you trust."""
    import pickle
    import types

    class _Bag:
        pass

    bags = {}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == 'xrnerf' or module.startswith('xrnerf.'):
                return bags.setdefault((module, name), type(name, (_Bag,), {'__module__': module}))
            return super().find_class(module, name)

    shim = types.ModuleType('xrnerf_amd_pickle_shim')
    shim.Unpickler, shim.load, shim.loads = _Unpickler, pickle.load, pickle.loads
    shim.__dict__.update({k: getattr(pickle, k) for k in ('HIGHEST_PROTOCOL', 'DEFAULT_PROTOCOL', 'PickleError', 'UnpicklingError')})
    cp = torch.load(path, map_location='cpu', pickle_module=shim, weights_only=False)
    queue = list(cp['root_nodes'])
    leaves = []
    for node in queue:                                   # the list grows while it is walked, like the reference's
        if hasattr(node, 'network'):
            leaves.append(node)
        else:
            queue.append(node.leq_child)
            queue.append(node.gt_child)
    if not leaves:
        raise ValueError('no leaf network in the checkpoint')

    def layer(net, *path):
        m = net
        for key in path:
            m = m._modules[key]
        w, b = m._parameters['weight'].detach(), m._parameters['bias'].detach()
        return w[0].t().contiguous(), b[0].contiguous()          # 'bmm' layout [1, out, in] -> [in, out]

    p0 = leaves[0].network
    n_hidden = int(p0.num_hidden_layers)
    names = [('pts_linears.%d' % l, ('pts_linears', str(l))) for l in range(n_hidden)]
    names += [(n, (n,)) for n in ('alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear')]
    sd = {}
    for name, path in names:
        ws, bs = zip(*[layer(leaf.network, *path) for leaf in leaves])
        sd[name + '.weight'], sd[name + '.bias'] = torch.stack(ws, 0), torch.stack(bs, 0)
    return {'domain_mins': torch.tensor([list(l.domain_min) for l in leaves], dtype=torch.float32),
            'domain_maxs': torch.tensor([list(l.domain_max) for l in leaves], dtype=torch.float32),
            'state_dict': sd, 'num_hidden_layers': n_hidden}


@MLPS.register_module()
class KiloNerfMLP(nn.Module):
    """kilonerf_mlp.py:27-190.  Checkpoints: `occupancy_checkpoint` is the reference's (a tensor saved with torch.save);
    `distilled_checkpoint` is either a plain dict {'domain_mins', 'domain_maxs', 'state_dict'[, 'num_hidden_layers']} with
    the multi network's state under the reference's parameter names, or the reference's own distillation checkpoint
    (a pickle of its Node / MultiNetwork objects), which is read without the reference package by
    `load_reference_distilled_checkpoint` -- pass `trust_pickle=True` through the mlp config to allow that (it executes
    a pickle, as the reference's torch.load does)."""

    def __init__(self, resolution=None, distilled_config=None, occupancy_checkpoint=None, distilled_checkpoint=None,
                 embedder=None, trust_pickle=False):
        super().__init__()
        self.resolution = list(resolution)
        self.distilled_config = distilled_config
        self.embedder = builder.build_embedder(embedder)
        occ = (torch.load(occupancy_checkpoint, map_location='cpu', weights_only=True) if isinstance(occupancy_checkpoint, str)
               else occupancy_checkpoint)
        cp = distilled_checkpoint
        if isinstance(cp, str):
            try:
                # weights_only spelled out: torch < 2.6 defaults to a full unpickle, which `trust_pickle=False` must not allow
                cp = torch.load(cp, map_location='cpu', weights_only=True)
            except Exception:                                         # noqa: BLE001  the reference's pickled Node tree
                if not trust_pickle:
                    raise NotImplementedError(
                        'distilled_checkpoint is not a plain dict of tensors: the reference pickle of Node objects is read '
                        'by kilo.load_reference_distilled_checkpoint -- set trust_pickle=True in the mlp config to allow it')
                cp = load_reference_distilled_checkpoint(cp)
        if not isinstance(cp, dict) or 'state_dict' not in cp:
            raise NotImplementedError('distilled_checkpoint must hold domain_mins / domain_maxs / state_dict')
        self._init_from(occ, cp['domain_mins'], cp['domain_maxs'], cp['state_dict'], cp.get('num_hidden_layers'))

    @classmethod
    def from_arrays(cls, resolution, occupancy, domain_mins, domain_maxs, state_dict, embedder, num_hidden_layers=None):
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.resolution, self.distilled_config = list(resolution), None
        self.embedder = builder.build_embedder(embedder)
        self._init_from(occupancy, domain_mins, domain_maxs, state_dict, num_hidden_layers)
        return self

    def _init_from(self, occupancy, domain_mins, domain_maxs, state_dict, num_hidden_layers):
        dm = torch.as_tensor(domain_mins, dtype=torch.float32)
        self.register_buffer('domain_mins', dm.clone())
        self.register_buffer('domain_maxs', torch.as_tensor(domain_maxs, dtype=torch.float32).clone())
        if occupancy is not None:
            self.register_buffer('occupancy_grid', torch.as_tensor(occupancy).reshape(-1).to(torch.bool))
        else:
            self.occupancy_grid = None
        if num_hidden_layers is None:
            num_hidden_layers = 1 + max(int(k.split('.')[1]) for k in state_dict if k.startswith('pts_linears.'))
        pos_ch, dir_ch = self.embedder.get_embed_ch()
        self.multi_network = MultiNetwork(dm.shape[0], pos_ch, dir_ch, 4, 32, num_hidden_layers, None, True, 32, 'relu')
        self.multi_network.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)

    def get_view_dependent_parameters(self):
        return self.multi_network.view_dependent_parameters

    def _host3(self, t):
        """python floats of a 3-vector.  A device tensor costs one read-back per call: the values are NOT cached by tensor
        identity (the reference pipeline builds these tensors anew every batch, and a freed address is reused)."""
        if not torch.is_tensor(t):
            return [float(v) for v in t]
        return [float(v) for v in t.detach().reshape(-1).tolist()]

    def forward(self, data):
        fixed_res = [x // 16 for x in self.resolution]                      # kilonerf_mlp.py:146
        gmin, gmax = self._host3(data['global_domain_min']), self._host3(data['global_domain_max'])
        kw = dict(pts=data['pts']) if 'pts' in data else dict(rays_o=data['rays_o'], rays_d=data['rays_d'], z_vals=data['z_vals'])
        mn = self.multi_network
        args = (data['viewdirs'], gmin, gmax, fixed_res, self.resolution, self.occupancy_grid, self.domain_mins,
                self.domain_maxs)
        tail = (self.embedder.multires, self.embedder.multires_dirs, mn.num_hidden_layers)
        if torch.is_grad_enabled() and any(p.requires_grad for p in mn.parameters()):
            data['raw'] = _KiloMlpFn.apply(args, tail, kw, *mn.ordered_parameters())      # fine-tuning: gradients flow
        else:
            data['raw'] = ops.kilo_mlp_forward(*args, mn.packed(), *tail, **kw)
        return data


class _KiloMlpFn(torch.autograd.Function):
    """KiloNerfMLP.forward with parameter gradients (the reference: AddMultiMatMul, multi_modules.py:198-236, six times):
    one forward call, one backward call (xr_kilo_mlp_backward) that re-runs the tiny MLPs and accumulates the packed
    gradient blocks; sample positions / directions carry no gradient (they are data)"""

    _grad_blocks = {}          # (device, N, stride) -> a zero-filled gradient block buffer (the unpack launch leaves it zero again)

    @staticmethod
    def forward(ctx, args, tail, kw, *params):
        on_dev = ops._on_device(params[0]) and hasattr(_lib_handle(), 'xr_kilo_pack_params')
        packed = ops.kilo_pack_params(params, *tail) if on_dev else MultiNetwork.pack([p.detach() for p in params])
        ctx.args, ctx.tail, ctx.kw, ctx.packed, ctx.params, ctx.native = args, tail, kw, packed, params, on_dev
        raw = ops.kilo_mlp_forward(*args, packed, *tail, **kw)
        ctx.ws_gen = ops.kilo_ws_generation()      # the workspace holds these samples' assignment until the next kilo call
        return raw

    @staticmethod
    def backward(ctx, draw):
        if not ctx.native:
            g = ops.kilo_mlp_backward(draw.contiguous(), *ctx.args, ctx.packed, *ctx.tail, **ctx.kw)
            return (None, None, None) + tuple(MultiNetwork.unpack_like(g, ctx.params))
        key = (str(ctx.packed.device), ) + tuple(ctx.packed.shape)
        buf = _KiloMlpFn._grad_blocks.get(key)
        if buf is None:
            buf = _KiloMlpFn._grad_blocks[key] = torch.zeros_like(ctx.packed)
        ops.kilo_mlp_backward(draw.contiguous(), *ctx.args, ctx.packed, *ctx.tail, reuse_generation=ctx.ws_gen, grad=buf, **ctx.kw)
        grads = ops.kilo_unpack_grads(buf, ctx.params, *ctx.tail, clear=True)
        return (None, None, None) + tuple(grads)


def _lib_handle():
    from . import _lib
    return _lib.load()


@NETWORKS.register_module()
class KiloNerfNetwork(NerfNetwork):
    """networks/kilonerf.py:17-154 (rendering: forward / batchify_forward of NerfNetwork with N_importance = 0)"""

    def __init__(self, cfg, mlp=None, mlp_fine=None, render=None):
        super().__init__(cfg, mlp=mlp, mlp_fine=mlp_fine, render=render)
        self.l2_regularization_lambda = dict(cfg).get('l2_regularization_lambda')

    def train_step(self, data, optimizer, **kwargs):
        """networks/kilonerf.py:27-58: MSE on the rendered colours + lambda * sum of the L2 norms of the view-dependent
        parameters (direction layer and rgb head of all networks)"""
        from .networks import img2mse, mse2psnr, unfold_batching
        for k in data:
            data[k] = unfold_batching(data[k])
        ret = self.forward(data, is_test=False)
        img_loss = img2mse(ret['rgb'], data['target_s'])
        psnr = mse2psnr(img_loss)
        loss = img_loss
        l2_loss = img_loss.new_zeros(())
        if self.l2_regularization_lambda is not None:
            vd = self.mlp.get_view_dependent_parameters()
            reg = vd[0].norm(2)
            for p in vd[1:]:
                reg = reg + p.norm(2)
            l2_loss = self.l2_regularization_lambda * reg
            loss = loss + l2_loss
        if 'coarse_rgb' in ret:
            loss = loss + img2mse(ret['coarse_rgb'], data['target_s'])
        # the reference reads three scalars back per iteration (.item()); here they stay on the device until looked at
        log_vars = {'loss': loss.detach(), 'psnr': psnr.detach(), 'L2 reg': l2_loss.detach()}
        return {'loss': loss, 'log_vars': log_vars, 'num_samples': ret['rgb'].shape[0]}


# ------------------------------------------------------------------ synthetic Lego-shaped scene (bench / smoke / tests)
LEGO_RESOLUTION = [144, 256, 160]             # configs/kilonerf/kilonerf_finetune_Synthetic_NeRF_base01.py:18
LEGO_GMIN = [-0.67, -1.2, -0.37]              # bounding box of the NSVF Synthetic_NeRF Lego scene
LEGO_GMAX = [0.67, 1.2, 1.03]


def synthetic_scene(device, resolution=None, gmin=None, gmax=None, seed=0, fill=0.07, weight_scale=2.5):
    """occupancy (union of boxes filling ~`fill` of the cells), node domains exactly as get_nodes_fixed_resolution
    builds them (datasets/kilonerf_node_dataset.py:108-135) and random weights for resolution//16 networks"""
    import itertools
    import numpy as np
    resolution = list(resolution or LEGO_RESOLUTION)
    gmin, gmax = np.float32(gmin or LEGO_GMIN), np.float32(gmax or LEGO_GMAX)
    fixed = [r // 16 for r in resolution]
    rng = np.random.default_rng(seed)
    occ = np.zeros(resolution, bool)
    while occ.mean() < fill:
        c = rng.uniform(0.25, 0.75, 3) * resolution
        h = rng.uniform(0.03, 0.12, 3) * resolution
        lo, hi = np.maximum((c - h).astype(int), 0), np.minimum((c + h).astype(int) + 1, resolution)
        occ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
    voxel = (gmax.astype(np.float64) - gmin.astype(np.float64)) / np.array(fixed)
    dmins, dmaxs = [], []
    for vi in itertools.product(*[range(r) for r in fixed]):
        dmins.append((gmin.astype(np.float64) + np.array(vi) * voxel).tolist())
        dmaxs.append((gmin.astype(np.float64) + (np.array(vi) + 1) * voxel).tolist())
    N = len(dmins)
    g = torch.Generator().manual_seed(seed)
    mn = MultiNetwork(N, 63, 27)
    with torch.no_grad():
        for p in mn.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * weight_scale / math.sqrt(p.shape[1] if p.dim() == 3 else 32))
    emb = dict(type='KiloNerfFourierEmbedder', num_networks=1, input_ch=3, multires=10, multires_dirs=4)
    mlp = KiloNerfMLP.from_arrays(resolution, occ.reshape(-1), torch.tensor(dmins), torch.tensor(dmaxs), mn.state_dict(), emb)
    return mlp.to(device), torch.tensor(gmin), torch.tensor(gmax)


def orbit_poses(n, radius=3.3, elevation=0.5):
    """camera-to-world matrices [n,4,4] on a circle around the scene, looking at its centre (OpenGL axes like the
    Blender / NSVF loaders: -z forward, y up)"""
    out = []
    for k in range(n):
        th = 2 * math.pi * k / n
        cam = torch.tensor([radius * math.cos(elevation) * math.cos(th), radius * math.cos(elevation) * math.sin(th),
                            radius * math.sin(elevation) + 0.3])
        fwd = torch.tensor([0., 0., 0.3]) - cam
        fwd = fwd / fwd.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0., 0., 1.]))
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        m = torch.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, cam
        out.append(m)
    return torch.stack(out)


def camera_rays(pose, H, W, focal, device):
    """KilonerfGetRays (datasets/pipelines/create.py:253-300: kilonerf_cuda.get_rays_d == the commented torch code):
    dirs = ((i - cx)/fx, -(j - cy)/fy, -1) rotated by c2w[:3,:3], origins = c2w[:3,3]; viewdirs = normalised dirs"""
    pose = torch.as_tensor(pose, dtype=torch.float32, device=device)
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing='ij')
    dirs = torch.stack([(i - 0.5 * W) / focal, -(j - 0.5 * H) / focal, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * pose[:3, :3], -1).reshape(-1, 3)
    rays_o = pose[:3, 3].expand(rays_d.shape).contiguous()
    return rays_o, rays_d.contiguous(), (rays_d / rays_d.norm(dim=-1, keepdim=True)).contiguous()


@torch.no_grad()
def render_frame(mlp, gmin, gmax, pose, H, W, focal, near=2.0, far=6.0, n_samples=384, white_bkgd=True, rays=None,
                 fused=True):
    """one frame through the reference's test path: rays -> 384 uniform samples (GetZvals, not randomized) ->
    KiloNerfMLP -> NerfRender.  fused: one C-ABI call that never writes a per-sample tensor except the network id
    (xr_kilo_render_rays); otherwise the module-level path (z_vals, raw [R,S,4], NerfRender) -- same pixels."""
    dev = mlp.domain_mins.device
    rays_o, rays_d, viewdirs = rays if rays is not None else camera_rays(pose, H, W, focal, dev)
    R = rays_o.shape[0]
    if fused:
        nf = torch.empty((2, R), device=dev)
        nf[0].fill_(near); nf[1].fill_(far)
        return ops.kilo_render_rays(rays_o, rays_d, viewdirs, nf[0], nf[1], n_samples, mlp._host3(gmin), mlp._host3(gmax),
                                    [x // 16 for x in mlp.resolution], mlp.resolution, mlp.occupancy_grid, mlp.domain_mins,
                                    mlp.domain_maxs, mlp.multi_network.packed(), mlp.embedder.multires,
                                    mlp.embedder.multires_dirs, mlp.multi_network.num_hidden_layers, white_bkgd)
    z = ops.mip_zvals(torch.full((R,), near, device=dev), torch.full((R,), far, device=dev), n_samples)
    data = {'rays_o': rays_o, 'rays_d': rays_d, 'viewdirs': viewdirs, 'z_vals': z, 'global_domain_min': gmin,
            'global_domain_max': gmax}
    raw = mlp(data)['raw']
    rgb, disp, acc, _ = ops.nerf_render_forward(raw, z, rays_d, white_bkgd)
    return rgb, disp, acc
