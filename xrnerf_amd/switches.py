"""The runtime switches of the host side, in one place (the C library reads XR_MLP_BWD_DW, XR_MLP_LIVE, XR_GEMM_F32 and XR_SC_TEST
itself).  tools/README.md lists every switch with the test that covers it; everything else that earlier rounds A/B-tested is either
a compile-time constant of one source (tools/build_variant.sh + XRNERF_LIB) or gone, with its record under profiles/.

  XRNERF_LIB                   path of another build of the library (_lib.py)
  XRNERF_MLP_PRECISION         f32 (default: fp32 storage, gradients and accumulation) | f16 (the reference's tcnn arithmetic) -- ops.set_precision
  XRNERF_F32_FORWARD           the fused MLP's product arithmetic in the f32 mode (ops.set_f32_forward): f16x2 (default -- every fp32
                               operand as two fp16 parts, three fp16 MFMAs per product block: ~4e-7 relative on the raw outputs, NOT
                               fp32-exact; operands above 65504 are saturated and counted, the trainer warns) | bf16x3 (three bf16
                               parts, fp32-rounding accuracy, no range limit below fp32's) | mfma (v_mfma_f32_32x32x2_f32: plain fp32).
                               The backward follows through XR_MLP_BWD_DW (h2f default | b2x | f32): `mfma` + XR_MLP_BWD_DW=f32 is
                               the fp32-exact pair the strict gradient bars of the tests are held against
  XRNERF_DP                    allreduce (default) | allreduce_bf16 (the table gradient crosses the links as bf16) | zero1 -- the
                               data-parallel gradient exchange (train.Trainer, dist.py)
  XRNERF_TRAINER               "k=v,..." overrides of Trainer's keyword switches: native_loop, fuse_adam, direct_step, march_window
                               (side | main | off), prefetch_k6 (A/B runs of bench.py / tools without editing code)
  XRNERF_STEP                  fused (default: one native call per training step) | py (the same entry points issued one by one from
                               Python: per-entry-point timers, the kernels' host build) | modular (sampler -> mlp -> render -> autograd)
  XRNERF_FRAME                 one_launch (default: a chunked test frame as one launch per kernel, same pixels) | async (the chunk loop
                               without a read-back per chunk) | sync (the reference's loop) | ert (early ray termination: samples behind
                               T < 1e-4 are not evaluated; pixels within 1e-4 of the default -- the reference has no such path)
  XRNERF_TCNN_STRICT_DEFAULTS  1: tcnn's default 5 hidden layers where the config's `num_layers` is not a tcnn key (mlps.py)
  XRNERF_VAL_RANK0_ONLY        1: validation frames on rank 0 only, like the reference (networks.py)
"""
import os


def step_mode():
    m = os.environ.get('XRNERF_STEP', 'fused')
    if m not in ('fused', 'py', 'modular'):
        raise ValueError("XRNERF_STEP must be 'fused', 'py' or 'modular' (got %r)" % m)
    return m


def frame_mode():
    m = os.environ.get('XRNERF_FRAME', 'one_launch')
    if m not in ('one_launch', 'async', 'sync', 'ert'):
        raise ValueError("XRNERF_FRAME must be 'one_launch', 'async', 'sync' or 'ert' (got %r)" % m)
    return m


TRAINER_KEYS = {'native_loop': bool, 'fuse_adam': bool, 'direct_step': bool, 'march_window': str, 'prefetch_k6': bool}


def trainer_overrides():
    """XRNERF_TRAINER="fuse_adam=0,march_window=off" -> {'fuse_adam': False, 'march_window': 'off'}"""
    out = {}
    for item in filter(None, os.environ.get('XRNERF_TRAINER', '').split(',')):
        k, _, v = item.partition('=')
        k = k.strip()
        if k not in TRAINER_KEYS:
            raise ValueError('XRNERF_TRAINER: unknown key %r (known: %s)' % (k, ', '.join(sorted(TRAINER_KEYS))))
        out[k] = bool(int(v)) if TRAINER_KEYS[k] is bool else TRAINER_KEYS[k](v)
    return out
