"""`HashNerfRender`: the density -> transmittance compositor behind the registry name of
/root/reference/xrnerf/models/renders/hashnerf_render.py:16-177, on the MI355X kernels.

What the reference spreads over one module method and two wrapper functions is organised here around ONE value: the marched
state a compositor launch needs (`Marched`), read off the sampler once.  The render module builds it, picks the training or the
test compositor, and returns `(data, {'rgb'[, 'alpha']})` -- the registry contract (`forward(data, sampler, is_test)`).
"""
from collections import namedtuple

import torch
from torch import nn

from . import ops
from .builder import RENDERS

# what K3 / K4 / K5 read besides the network output: K1's coordinate rows, the per-ray (count, base) pairs before and after K2's clip,
# the density grid's mean (K4's regulariser), the two activation codes (raymarch_shared.h:619-625)
Marched = namedtuple('Marched', 'coords numsteps numsteps_clipped grid_mean rgb_act density_act')


def marched_state(sampler):
    """the sampler's public per-batch attributes (the ones hashnerf_render.py:35-41,50 reads) as one value"""
    return Marched(sampler.coords, sampler.rays_numsteps, getattr(sampler, 'rays_numsteps_compacted', None),
                   sampler.density_grid_mean, int(sampler.rgb_activation), int(sampler.density_activation))


def _check_f32(**tensors):
    for name, t in tensors.items():
        if t.dtype != torch.float32:
            raise TypeError('%s must be float32 (got %s)' % (name, t.dtype))


class _Composite(torch.autograd.Function):
    """K3 forward / K4 backward (calc_rgb.cu:6-140) as one autograd node over the network output"""

    @staticmethod
    def forward(ctx, raw, background, m):
        _check_f32(raw=raw, coords=m.coords, background=background)
        if m.numsteps.dtype != torch.int32 or m.numsteps_clipped.dtype != torch.int32:
            raise TypeError('per-ray (count, base) pairs must be int32')
        raw = raw.contiguous()
        rgb = ops.calc_rgb_forward(raw, m.coords, m.numsteps, m.numsteps_clipped, background, m.rgb_act, m.density_act)
        ctx.save_for_backward(raw, rgb)
        ctx.marched = m
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        raw, rgb = ctx.saved_tensors
        m = ctx.marched
        d_raw = torch.zeros_like(raw)                  # rows no ray covers keep a zero gradient (hashnerf_render.py:121-123)
        ops.calc_rgb_backward(raw, m.numsteps_clipped, m.coords, d_rgb.contiguous(), rgb, m.grid_mean, m.rgb_act, m.density_act,
                              out=d_raw)
        return d_raw, None, None


def composite_test(raw, m, bg_color):
    """K5 (calc_rgb.cu:144-206): colour + alpha per ray against a constant background read on the host (like the reference's
    `bg_color_cpu`, calc_rgb.cu:366,378)"""
    bg = [float(v) for v in bg_color.detach().cpu().reshape(-1)[:3]]
    with torch.no_grad():
        return ops.calc_rgb_inference(raw.contiguous(), m.coords, m.numsteps, bg, m.rgb_act, m.density_act)


@RENDERS.register_module()
class HashNerfRender(nn.Module):
    def __init__(self, bg_color=None, **kwarg):
        super().__init__()
        self.bg_color = torch.tensor(bg_color).to(dtype=torch.float32)

    def forward(self, data, sampler, is_test=False):
        m = marched_state(sampler)
        if is_test:
            rgb, alpha = composite_test(data['raw'], m, self.bg_color)
            return data, {'rgb': rgb, 'alpha': alpha}
        return data, {'rgb': _Composite.apply(data['raw'], data['bg_color'].detach(), m)}
