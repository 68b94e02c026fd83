"""Builds xrnerf_amd/libxrnerf_mi355.so (gfx950 only) with hipcc, in-tree.

`python -m xrnerf_amd.build [--force]`.  hipcc cross-compiles without a GPU; the built .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libxrnerf_mi355.so')
OBJ = os.path.join(HERE, 'build')
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wall',
          '-Wno-unused-variable', '-Wno-unused-but-set-variable']
# K1/K6/K11 make index decisions on `o + t*d`-style expressions: keep mul and add un-fused so they
# agree bit for bit with the CPU-compiled reference.
SOURCES = {
    'xr_raymarch.hip': ['-ffp-contract=off'],
    'xr_grid.hip': ['-ffp-contract=off'],
    # the hash-grid cell index is floor(x*scale+0.5): an index decision at scale up to 2047
    'xr_encode.hip': ['-ffp-contract=off'],
    # same index decisions and the same weight products as the gather
    # (-simplifycfg-sink-common=false: LLVM otherwise merges the `rank slot k` stores of different unrolled items into one
    # block with a dynamic slot index, which turns the per-thread rank registers of k_scatter_bin3 into scratch memory)
    'xr_scatter.hip': ['-ffp-contract=off', '-mllvm', '-simplifycfg-sink-common=false'],
    'xr_mlp.hip': [],
    'xr_misc.hip': ['-ffp-contract=off'],
    # Mip-NeRF stages: fp32 in the reference's operation order (lower + (upper-lower)*rand etc.)
    'xr_mip.hip': ['-ffp-contract=off'],
    # KiloNeRF: sample positions o + d*z and the cell index arithmetic must round like the reference's tensor ops
    # (the MLP's FMAs are explicit fmaf calls)
    'xr_kilo.hip': ['-ffp-contract=off'],
    'xr_gemm.hip': [],
    # host-side step executor (calls the entry points above in sequence)
    'xr_step.hip': [],
    # RCCL driven from native code (dlopen'ed on first use): the data-parallel loop's gradient exchange
    'xr_dist.hip': [],
}


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', 'hipcc'):
        if os.path.exists(c) or c == 'hipcc':
            return c


def sources_hash():
    """sha256 over the library's sources (csrc/*, the public header, this recipe) in a fixed order"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    for f in files + [os.path.join(HERE, '..', 'include', 'xrnerf_mi355.h'), os.path.abspath(__file__)]:
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


STAMP = OUT + '.stamp'          # "<sources hash> <library hash>" written by the build that produced OUT


def _file_hash(path):
    import hashlib
    with open(path, 'rb') as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def _stale(dst, srcs):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, 'xr_common.h'), os.path.join(CSRC, 'xr_mip_math.h'), os.path.join(CSRC, 'xr_hashgrid.h'),
               os.path.join(CSRC, 'xr_scatter.h'), os.path.join(CSRC, 'xr_adam.h'), os.path.join(HERE, '..', 'include', 'xrnerf_mi355.h'),
               os.path.abspath(__file__)]
    have_src = all(os.path.exists(os.path.join(CSRC, s)) for s in SOURCES)
    if not have_src:
        if os.path.exists(OUT):
            return OUT
        raise RuntimeError('xrnerf_amd/csrc sources missing and no prebuilt library')
    jobs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o, extra))

    def cc(job):
        s, o, extra = job
        cmd = [_hipcc()] + COMMON + extra + os.environ.get('XR_EXTRA_HIPCC_FLAGS', '').split() + ['-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stdout + r.stderr))
        if verbose and (r.stdout or r.stderr):
            sys.stderr.write(r.stdout + r.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(len(jobs)) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + ['-ldl']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed: %s\n%s' % (' '.join(cmd), r.stdout + r.stderr))
        with open(STAMP, 'w') as fh:
            fh.write('%s %s\n' % (sources_hash(), _file_hash(OUT)))
        global BUILT_HERE
        BUILT_HERE = True
    return OUT


BUILT_HERE = False


def info():
    """where the loaded binary comes from: was it compiled in this process, and is it the build of exactly the sources beside it?"""
    out = {'library': os.path.basename(OUT), 'compiled_in_this_process': BUILT_HERE}
    try:
        src, lib = open(STAMP).read().split()
        out['library_sha16'] = _file_hash(OUT)[:16]
        out['binary_is_the_stamped_build'] = lib == _file_hash(OUT)
        out['sources_match_the_stamped_build'] = src == sources_hash() if os.path.isdir(CSRC) else None
        out['sources_sha16'] = src[:16]
    except (OSError, ValueError):
        out['stamp'] = 'missing'
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
