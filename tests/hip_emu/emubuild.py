"""TEST INFRASTRUCTURE ONLY: builds tests/hip_emu/_build/libemu_<name>.so = one kernel source of xrnerf_amd/csrc compiled
for the HOST through the HIP-on-CPU shim (hip/hip_runtime.h + emu.cpp).  The source is used as it is, except that its
`extern __shared__ T name[];` declarations (no host spelling exists) become `T* name = (T*)emu::dyn_lds();`."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'xrnerf_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
# translation units that call into another one (library-internal C++ interfaces): built into the same shared object
COMPANIONS = {'xr_encode': ('xr_scatter',)}


def build(name, extra=()):
    """name: 'xr_mip' | 'xr_kilo' | 'xr_gemm' -> path of the shared object (rebuilt when a source is newer).
    Several processes may ask at once (the 2-rank gloo tests): one builds under a file lock, into temporary names that are renamed
    into place, the others wait and find it up to date."""
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(CSRC, name + '.hip')
    so = os.path.join(OUT, 'libemu_%s.so' % name)
    deps = [src] + [os.path.join(CSRC, u + '.hip') for u in COMPANIONS.get(name, ())] + [os.path.join(HERE, 'emu.cpp'), os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]

    def fresh():
        return os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps)
    if fresh():
        return so
    with open(os.path.join(OUT, '.lock_%s' % name), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if fresh():                                   # another process built it while this one waited
                return so
            tag = '.%d' % os.getpid()
            cpps = []
            for unit in (name,) + COMPANIONS.get(name, ()):
                usrc = os.path.join(CSRC, unit + '.hip')
                text = open(usrc).read()
                text, n = re.subn(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\s*\[\s*\]\s*;',
                                  r'\1* \2 = (\1*)emu::dyn_lds();', text)
                cpp = os.path.join(OUT, unit + '_host.cpp')
                with open(cpp + tag, 'w') as f:
                    f.write('// generated from %s by tests/hip_emu/emubuild.py (%d dynamic-LDS declarations rewritten)\n' % (usrc, n) + text)
                os.replace(cpp + tag, cpp)
                cpps.append(cpp)
            cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O1', '-g0', '-fPIC', '-shared', '-ffp-contract=off', '-Wno-everything',
                   '-I', HERE, '-I', CSRC] + cpps + [os.path.join(HERE, 'emu.cpp'), '-o', so + tag] + list(extra)
            subprocess.check_call(cmd)
            os.replace(so + tag, so)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return so


if __name__ == '__main__':
    import sys
    for n in sys.argv[1:] or ['xr_mip', 'xr_kilo', 'xr_gemm']:
        print(build(n))
