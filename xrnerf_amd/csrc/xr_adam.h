// Adam update of one parameter, shared by the optimiser kernels (xr_misc.hip) and the table scatter that applies it to the
// entries it has just accumulated (xr_scatter.hip, XrAdamArgs): one definition, so the fused and the separate path agree bit for bit.
#pragma once
#include "xr_common.h"

// torch.optim.Adam semantics (L2 weight decay folded into the gradient).
// gs: a factor on the incoming gradient (1/world_size of data-parallel averaging, applied here instead of in a pass of its
// own over the 48.8-MB gradient); rounded on its own first, so the update is bit for bit the one of `g *= gs` + this kernel
__device__ inline void adam1(float& p, float g, float& m, float& v, float b1, float b2, float step_size, float bc2s,
                             float eps, float wd, float gs = 1.f) {
    g = __fmul_rn(g, gs);
    g = g + wd * p;
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p = p - step_size * (m / (sqrtf(v) / bc2s + eps));
}
// EMA copy kept beside the parameters (mmcv's EMAHook): e <- (1 - mom) e + mom p
__device__ inline float ema1(float e, float p, float mom) { return (1.f - mom) * e + mom * p; }

// what a kernel needs to apply the update itself: the whole tensor's p / m / v / ema (ema nullable) + the step's constants
// (step_size = lr / (1 - beta1^t), bc2s = sqrt(1 - beta2^t), as xr_adam_step_multi passes them to its kernel)
struct XrAdamArgs { float* p; float* m; float* v; float* ema; float b1, b2, step_size, bc2s, eps, wd, mom, gs; };
