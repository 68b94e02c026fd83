"""`linear_act`: y = act(x W^T + b) of the 8x256 NeRF MLP (nerf_mlp.py:62-94: nn.Linear + F.relu) as one autograd node on
the fp32-MFMA kernels of xrnerf_amd/csrc/xr_gemm.hip: forward with bias + relu in the epilogue; backward = input
gradient and weight gradient with the relu mask applied while the incoming gradient is loaded (no masked copy of it,
no separate relu / bias passes); the bias gradient is a masked column sum (xr_linear_backward_bias).
Layers the kernel does not take (K or N not a multiple of 4: the 283-wide view layer, the 1- and 3-wide heads) and
host tensors use torch's own linear."""
import torch
import torch.nn.functional as F

from . import ops


class _LinearAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        y = ops.linear_forward(x, w, b, relu)
        ctx.relu, ctx.has_bias = relu, b is not None
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        mask = y if ctx.relu else None
        dx = ops.linear_backward_input(dy, mask, w) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2] and ctx.needs_input_grad[1]:
            dw, db = ops.linear_backward_weight_bias(dy, mask, x)      # (the bias gradient rides on the weight-gradient product)
        else:
            dw = ops.linear_backward_weight(dy, mask, x) if ctx.needs_input_grad[1] else None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = ops.linear_backward_bias(dy, mask)
        return dx, dw, db, None


def linear_act(x, weight, bias=None, relu=False):
    if ops.linear_ok(x, weight):
        return _LinearAct.apply(x, weight, bias, relu)
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


def linear_act_padded(x, weight, bias=None, relu=False):
    """linear_act for any K / N on the device: zero columns / rows bring both to multiples of 4 (exact: the padded
    products are 0 * 0, the padded outputs are dropped), so narrow heads (1, 3 outputs) and the 283-wide view layer stay
    on the MFMA kernel instead of hipBLASLt's K = 1 / K = 3 gradient GEMMs (0.4-0.7 ms each at 131072 rows)"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0):
        return linear_act(x, weight, bias, relu)
    N, K = weight.shape
    pk, pn = (-K) % 4, (-N) % 4
    if pk:
        x = torch.cat([x, x.new_zeros((x.shape[0], pk))], -1)
        weight = F.pad(weight, (0, pk))
    if pn:
        weight = F.pad(weight, (0, 0, 0, pn))
        bias = F.pad(bias, (0, pn)) if bias is not None else None
    y = linear_act(x, weight, bias, relu)
    return y[:, :N] if pn else y
