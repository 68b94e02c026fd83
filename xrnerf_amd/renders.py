"""`HashNerfRender`: density -> transmittance compositor, the registered type of
/root/reference/xrnerf/models/renders/hashnerf_render.py:16-177, on the MI355X kernels."""
import torch
from torch import nn
from torch.autograd import Function

from . import ops
from .builder import RENDERS


class _calc_rgb_bp(Function):
    """hashnerf_render.py:60-147 (K3 forward, K4 backward)."""

    @staticmethod
    def forward(ctx, network_output, coords_in, rays_numsteps, rays_numsteps_compacted, training_background_color,
                density_grid_mean, rgb_activation, density_activation, aabb_range):
        assert network_output.dtype == torch.float32 and coords_in.dtype == torch.float32, 'data type error!!!'
        assert rays_numsteps.dtype == torch.int32 and rays_numsteps_compacted.dtype == torch.int32, 'data type error!!!'
        assert training_background_color.dtype == torch.float32, 'data type error!!!'
        network_output = network_output.contiguous()
        rgb_output = ops.calc_rgb_forward(network_output, coords_in, rays_numsteps, rays_numsteps_compacted,
                                          training_background_color, rgb_activation, density_activation)
        ctx.save_for_backward(network_output, rays_numsteps_compacted, coords_in, rgb_output, density_grid_mean)
        ctx.extro = [rgb_activation, density_activation, aabb_range]
        return rgb_output

    @staticmethod
    def backward(ctx, grad_rgb_output):
        network_output, rays_numsteps_compacted, coords_in, rgb_output, density_grid_mean = ctx.saved_tensors
        rgb_activation, density_activation, aabb_range = ctx.extro
        grad_network_output = torch.zeros_like(network_output)     # rows no ray covers stay 0 (:121-123)
        ops.calc_rgb_backward(network_output, rays_numsteps_compacted, coords_in, grad_rgb_output.contiguous(),
                              rgb_output, density_grid_mean, rgb_activation, density_activation,
                              out=grad_network_output)
        return grad_network_output, None, None, None, None, None, None, None, None


calc_rgb_bp = _calc_rgb_bp.apply


def calc_rgb_nobp(network_output, coords_in, rays_numsteps, bg_color_cpu, rgb_activation, density_activation,
                  aabb_range):
    """hashnerf_render.py:150-177 (K5); bg_color_cpu: shape (3,), read on the host like there."""
    bg = [float(v) for v in bg_color_cpu.detach().cpu().reshape(-1)[:3]]
    with torch.no_grad():
        return ops.calc_rgb_inference(network_output.contiguous(), coords_in, rays_numsteps, bg, rgb_activation,
                                      density_activation)


@RENDERS.register_module()
class HashNerfRender(nn.Module):
    def __init__(self, bg_color=None, **kwarg):
        super().__init__()
        self.bg_color = torch.tensor(bg_color).to(dtype=torch.float32)

    def forward(self, data, sampler, is_test=False):
        network_output = data['raw']
        coords = sampler.coords
        aabb_range = sampler.aabb_range
        rays_numsteps = sampler.rays_numsteps
        density_grid_mean = sampler.density_grid_mean
        rgb_activation = int(sampler.rgb_activation)
        density_activation = int(sampler.density_activation)
        if is_test:
            rgb_output, alpha_output = calc_rgb_nobp(network_output, coords, rays_numsteps, self.bg_color,
                                                     rgb_activation, density_activation, aabb_range)
            ret = {'rgb': rgb_output, 'alpha': alpha_output}
        else:
            bg_color = data['bg_color'].detach()
            rgb_output = calc_rgb_bp(network_output, coords, rays_numsteps, sampler.rays_numsteps_compacted, bg_color,
                                     density_grid_mean, rgb_activation, density_activation, aabb_range)
            ret = {'rgb': rgb_output}
        return data, ret
