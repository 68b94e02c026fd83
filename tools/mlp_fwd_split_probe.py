"""forward of the fused MLP (1, 2) by operand arithmetic, all in ONE process on the binary as built: fp32 MFMA, 3-way bf16 split, 2-way
fp16 split (ops.set_f32_forward) -- error against a float64 statement from features of 1e-4 up to the documented boundary of the fp16
split (features +-4000; hidden activations up to 65504) and beyond it (saturation: bounded error + a count in the range word, never
inf / NaN); time at 2^18 rows.  Record: profiles/r06_mlp_fwd_f16x2_split_probe.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops, synthetic as S, build
dev = torch.device('cuda:0'); n = 1 << 18; m = 20000
print('library:', build.info())
wd, wc = S.mlp_weights(32, 64, 1, 16, 4), S.mlp_weights(32, 64, 2, 16, 5)
g = torch.Generator().manual_seed(0)
dirs = torch.rand((n, 3), generator=g).to(dev)


def f64(enc, dirs, twd, twc, sat=False):
    """float64 statement; sat: the split's saturation of its operands (features x 16, hidden activations, colour inputs at +-65504)"""
    e = enc.double().t(); W = [w.double() for w in ops._net_layers(twd, 1)]
    clamp = (lambda t: t.clamp(-65504.0, 65504.0)) if sat else (lambda t: t)
    h = clamp(torch.relu(clamp(e * 16.0) / 16.0 @ W[0].t())); dout = h @ W[1].t()
    cin = clamp(torch.cat([dout[:, 1:16], ops.sh4(dirs).double(), torch.ones((e.shape[0], 1), dtype=torch.float64, device=dev)], 1))
    C = [w.double() for w in ops._net_layers(twc, 2)]
    h1 = clamp(torch.relu(cin @ C[0].t())); h2 = clamp(torch.relu(h1 @ C[1].t())); cout = h2 @ C[2].t()
    return torch.cat([cout[:, :3], dout[:, :1]], 1), float(max(h.max(), h1.max(), h2.max()))


def case(label, enc, wd_, wc_=wc, sat=False):
    twd, twc = torch.from_numpy(wd_).to(dev), torch.from_numpy(wc_).to(dev)
    ref, hmax = f64(enc[:, :m], dirs[:m], twd, twc, sat)
    out, ev = {}, {}
    for mode in ('mfma', 'bf16x3', 'f16x2'):
        ops.set_f32_forward(mode)
        ops.mlp_range_events(dev, reset=True)
        raw = ops.nerf_mlp_fwd(enc, dirs, n, twd, twc, 1, 2)
        ev[mode] = ops.mlp_range_events(dev, reset=True)
        d = (raw[:m].double() - ref).abs()
        out[mode] = float(d.max()) if bool(torch.isfinite(raw).all()) else float('nan')
    scale = float(ref.abs().max())
    print('%-44s max|feature| %-8.3g max hidden %-8.3g max|raw| %-8.3g  max abs error vs float64%s: fp32 MFMA %.3g (%.1e rel)   bf16 x 3 %.3g (%.1e)   '
          'fp16 x 2 %.3g (%.1e)   range events (f16x2) %d' % (label, float(enc[:, :m].abs().max()), hmax, scale, ' [saturating statement]' if sat else '',
                                                             out['mfma'], out['mfma'] / scale, out['bf16x3'], out['bf16x3'] / scale, out['f16x2'], out['f16x2'] / scale, ev['f16x2']), flush=True)
    ops.set_f32_forward('f16x2')


for scale in (1e-4, 1e-2, 0.5, 20.0):
    case('features ~ N(0, %g)' % scale, (torch.randn((32, n), generator=g) * scale).to(dev), wd)
# the documented boundary: features up to +-4000 (x 2^4 = 64000 < 65504), hidden activations up to ~3.6e4 (first layer x 4, its output layer x 0.25, the colour net's first layer x 0.125)
enc_b = ((torch.rand((32, n), generator=g) * 2 - 1) * 4000.0).to(dev); enc_b[:, 0] = 4000.0; enc_b[:, 1] = -4000.0
wd_b = wd.copy(); wd_b[:2048] *= 4.0; wd_b[2048:] *= 0.25
wc_b = wc.copy(); wc_b[:2048] *= 0.125
case('boundary: features +-4000, hidden ~3.6e4', enc_b, wd_b, wc_b)
# the floor of the split, for the record: an operand below 2^-3 has an fp16-subnormal low part, i.e. it is carried to 2^-25 ABSOLUTE;
# output-layer weights of ~2e-5 against activations of 3.6e4 show it (every realistic regime above does not)
wd_s = wd.copy(); wd_s[:2048] *= 4.0; wd_s[2048:] *= 1e-4
case('floor: weights ~2e-5 x activations 3.6e4', enc_b, wd_s)
# beyond it: features of 1e4 (saturate at 4094), first layer x 8 (hidden activations above 65504 saturate)
enc_o = enc_b.clone(); enc_o[:, 5] = 1e4; enc_o[3, 6] = -3e5
case('beyond: features 1e4 / -3e5', enc_o, wd_b, wc_b, sat=True)
wd_o = wd.copy(); wd_o[:2048] *= 8.0; wd_o[2048:] *= 0.25
case('beyond: hidden activations > 65504', enc_b, wd_o, wc_b, sat=True)

enc = (torch.randn((32, n), generator=g) * 0.1).to(dev); raw = torch.empty((n, 4), device=dev)
twd, twc = torch.from_numpy(wd).to(dev), torch.from_numpy(wc).to(dev)


def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps


for mode in ('mfma', 'bf16x3', 'f16x2'):
    ops.set_f32_forward(mode)
    print('%-7s forward at 2^18 rows: %.1f us   density only: %.1f us' % (mode, 1e3 * timeit(lambda: ops.nerf_mlp_fwd(enc, dirs, n, twd, twc, 1, 2, raw=raw)),
          1e3 * timeit(lambda: ops.nerf_mlp_fwd(enc, None, n, twd, None, 1, 2, raw=raw))))
ops.set_f32_forward('f16x2')
ops.mlp_range_tracking(dev, False)        # the training loop's normal iterations: saturation without the count
print('f16x2 without the range count (xr_set_mlp_range_word(NULL)): forward at 2^18 rows: %.1f us   density only: %.1f us' % (
    1e3 * timeit(lambda: ops.nerf_mlp_fwd(enc, dirs, n, twd, twc, 1, 2, raw=raw)), 1e3 * timeit(lambda: ops.nerf_mlp_fwd(enc, None, n, twd, None, 1, 2, raw=raw))))
ops.mlp_range_tracking(dev, True)
