"""Golden fixture for K1 on an UNBOUNDED-style scene (aabb_scale = 16, five active cascades: SURVEY.md section 8d
config #4), from the reference's own ray_sampler.cu compiled for the CPU (oracle/build.py).  Build container only.

    python tests/golden/make_golden_cascades.py  ->  tests/golden/ref_raymarch_cascades.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def cascade_inputs():
    """shared by the generator and the tests: nested occupancy (a ball of radius 0.3 around the centre, seen at
    every cascade resolution, plus a far shell only the coarse cascades can see) and rays from far outside"""
    from xrnerf_amd import synthetic as S
    grid = S.sphere_density_grid(0.3)
    idx = np.arange(S.G3, dtype=np.uint32)
    xyz = np.stack([S.morton3d_invert(idx >> np.uint32(k)) for k in range(3)], -1).astype(np.float32)
    for level in (3, 4):                                   # a shell at radius ~3: only cascades 3 and 4 hold it
        pos = ((xyz + 0.5) / S.GRID - 0.5) * (2.0 ** level) + 0.5
        r = np.linalg.norm(pos - 0.5, axis=1)
        grid[level * S.G3:(level + 1) * S.G3] = np.maximum(grid[level * S.G3:(level + 1) * S.G3], ((r > 2.8) & (r < 3.2)).astype(np.float32))
    rng = np.random.default_rng(77)
    n = 192
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = (0.5 + 6.0 * u).astype(np.float32)                # outside the 16-wide box? no: inside it (|o - 0.5| = 6 < 8)
    tgt = 0.5 + rng.uniform(-0.4, 0.4, (n, 3))
    d = tgt - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o[:16] = (0.5 + 12.0 * u[:16]).astype(np.float32)     # 16 rays start outside the box and enter through a face
    d[:16] = ((0.5 - o[:16]) / np.linalg.norm(0.5 - o[:16], axis=1, keepdims=True)).astype(np.float32)
    return grid, o, d, (-7.5, 8.5)


def main():
    import oracle as O
    assert O.have_ref(), 'needs /root/reference (run in the build container)'
    grid, o, d, aabb = cascade_inputs()
    bf = O.bitfield_given_mean(grid, np.float32(0.5), backend='ref')
    c, ri, ns, cnt = O.rays_sampler(o, d, bf, aabb=aabb, rng_calls=0, backend='ref')
    s = int(cnt[1])
    out = {'numsteps': ns, 'index': ri, 'counter': cnt, 'coords': c[:s]}
    np.savez_compressed(os.path.join(HERE, 'ref_raymarch_cascades.npz'), **out)
    dt = c[:s, 3]
    print('rays', o.shape[0], 'samples', s, 'max per ray', int(ns[:, 0].max()), 'distinct dt', len(np.unique(dt)))


if __name__ == '__main__':
    main()
