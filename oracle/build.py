"""TEST INFRASTRUCTURE ONLY. Builds the two halves of the CPU oracle.

  1. oracle/libngp_oracle.so   <- oracle/ngp_oracle.c, our plain-C restatement of the whole
     Instant-NGP hot path (always built; travels to the GPU box as a prebuilt .so).
  2. oracle/_ref/libref_raymarch.so <- the REFERENCE'S OWN kernels
     (/root/reference/extensions/ngp_raymarch/src/*.cu) compiled for the CPU through
     oracle/shim, straight from where they lie (only when /root/reference exists; the
     recipe is SURVEY.md Appendix D).  Two shadow files are generated on the fly into
     oracle/_ref/shadow/ (git-ignored): raymarch_shared.h with its single <<<>>> launch
     replaced by a serial (block, thread) loop, and update_bitfield.cu with the
     warp-shuffle block_reduce launch replaced by a serial sum.

Usage: python oracle/build.py [--force]
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/extensions/ngp_raymarch'
OUT = os.path.join(HERE, '_ref')
CFLAGS = ['-O2', '-fPIC', '-ffp-contract=off', '-fno-fast-math']


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(' '.join(cmd) + '\n' + r.stdout + r.stderr)
        raise RuntimeError('oracle build failed')


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs if os.path.exists(s))


def build_port(force=False):
    src = os.path.join(HERE, 'ngp_oracle.c')
    dst = os.path.join(HERE, 'libngp_oracle.so')
    if not force and _newer(dst, [src]):
        return dst
    if not os.path.exists(src):
        if os.path.exists(dst):
            return dst
        raise RuntimeError('oracle/ngp_oracle.c missing')
    _run(['gcc', '-std=c11', '-shared', '-fopenmp'] + CFLAGS + ['-o', dst, src, '-lm'])
    return dst


def _make_shadows():
    sh = os.path.join(OUT, 'shadow')
    os.makedirs(sh, exist_ok=True)
    s = open(os.path.join(REF, 'include/raymarch_shared.h')).read()
    pat = ('kernel<<<n_blocks_linear(n_elements), n_threads_linear, shmem_size, stream>>>'
           '((uint32_t)n_elements, args...);')
    assert s.count(pat) == 1, 'raymarch_shared.h launch statement not found exactly once'
    s = s.replace(pat, (
        '{ blockDim.x = n_threads_linear; const uint32_t nb_ = n_blocks_linear(n_elements);\n'
        '\t  for (uint32_t b_ = 0; b_ < nb_; ++b_) for (uint32_t t_ = 0; t_ < n_threads_linear; ++t_) {\n'
        '\t    blockIdx.x = b_; threadIdx.x = t_; kernel((uint32_t)n_elements, args...); } }'))
    open(os.path.join(sh, 'raymarch_shared.h'), 'w').write(s)
    u = open(os.path.join(REF, 'src/update_bitfield.cu')).read()
    pat2 = ('block_reduce<T, T_OUT, F><<<blocks * n_sums, threads, 0, stream>>>'
            '(n_elements, fun, device_pointer, workspace, blocks);')
    assert u.count(pat2) == 1, 'update_bitfield.cu reduce launch not found exactly once'
    u = u.replace(pat2, (
        '{ (void)blocks; for (uint32_t e_ = 0; e_ < n_elements * N_ELEMS_PER_LOAD; ++e_) '
        'workspace[0] += fun(device_pointer[e_]); }'))
    open(os.path.join(sh, 'update_bitfield.cu'), 'w').write(u)
    return sh


def build_ref(force=False):
    """Returns the path of libref_raymarch.so, or None when the reference tree is absent
    (the GPU box) and no prebuilt copy travelled with the snapshot."""
    dst = os.path.join(OUT, 'libref_raymarch.so')
    if not os.path.isdir(REF):
        return dst if os.path.exists(dst) else None
    cap = os.path.join(HERE, 'ref_capi.cpp')
    shim = [os.path.join(HERE, 'shim', f) for f in ('cuda_runtime.h', 'torch/extension.h')]
    if not force and _newer(dst, [cap] + shim):
        return dst
    os.makedirs(OUT, exist_ok=True)
    sh = _make_shadows()
    inc = ['-I', sh, '-I', os.path.join(HERE, 'shim'), '-I', os.path.join(REF, 'src'),
           '-I', os.path.join(REF, 'include'), '-I', os.path.join(REF, 'include/op_include/eigen'),
           '-I', os.path.join(REF, 'include/op_include/pcg32')]
    tus = ['GLOBALS', 'K1', 'K2', 'K345', 'K6', 'K7', 'K8', 'K9', 'K1011']

    def cc(tu):
        obj = os.path.join(OUT, 'ref_%s.o' % tu)
        _run(['g++', '-std=c++17', '-x', 'c++', '-w', '-c'] + CFLAGS + ['-DTU_' + tu] + inc +
             [cap, '-o', obj])
        return obj

    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, tus))
    _run(['g++', '-shared', '-o', dst] + objs)
    return dst


if __name__ == '__main__':
    force = '--force' in sys.argv
    print(build_port(force))
    print(build_ref(force))
