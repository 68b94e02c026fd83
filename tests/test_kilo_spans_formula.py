"""The per-ray sample spans of the fused KiloNeRF frame path (k_kilo_spans, xrnerf_amd/csrc/xr_kilo.hip) must be
CONSERVATIVE: no lattice sample that passes the reference's domain test (transforms.py:107-110) may fall outside its
ray's [lo, hi).  This is a fp32 numpy replica of the kernel's formula held against the oracle's exact per-sample test on
a few hundred thousand rays, including the adversarial families (direction components of 0, 1e-6 .. 1e-2, origins inside
the box or on its faces, grazing rays, far < near).  The kernel itself is compared bit for bit with the dense path on the
GPU (tests/test_gpu_kilo.py); this test covers far more rays than fit there."""
import numpy as np

F = np.float32


def spans_replica(o, d, near, far, lo_eps, hi_eps, n_s):
    """fp32 mirror of k_kilo_spans (lindisp == 0)"""
    o, d, near, far = (np.asarray(a, np.float32) for a in (o, d, near, far))
    z0, z1 = np.minimum(near, far).copy(), np.maximum(near, far).copy()
    empty = np.zeros(o.shape[0], bool)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        for a in range(3):
            da, oa = d[:, a], o[:, a]
            big = np.abs(da) >= F(1e-3)
            za = ((lo_eps[a] - oa) / da).astype(np.float32)
            zb = ((hi_eps[a] - oa) / da).astype(np.float32)
            z0 = np.where(big, np.maximum(z0, np.minimum(za, zb)), z0)
            z1 = np.where(big, np.minimum(z1, np.maximum(za, zb)), z1)
            empty |= (da == 0) & ((oa <= lo_eps[a] - F(1e-3)) | (oa >= hi_eps[a] + F(1e-3)))
        scale = (F(n_s - 1) / (far - near)).astype(np.float32)
        a_, b_ = ((z0 - near) * scale).astype(np.float32), ((z1 - near) * scale).astype(np.float32)
    a2, b2 = np.minimum(a_, b_), np.maximum(a_, b_)
    fl, fh = np.floor(a2) - F(2), np.ceil(b2) + F(3)
    lo = np.where(fl <= 0, 0, np.where(fl >= n_s, n_s, fl)).astype(np.int64)
    hi = np.where(fh <= 0, 0, np.where(fh >= n_s, n_s, fh)).astype(np.int64)
    hi = np.maximum(hi, lo)
    dead = empty | ~(z0 <= z1)
    lo[dead] = 0
    hi[dead] = 0
    same = far == near
    lo[same & ~empty] = 0
    hi[same & ~empty] = n_s
    return lo, hi


def rays(rng, n, lo, hi):
    o = np.empty((n, 3)); d = np.empty((n, 3))
    kind = rng.integers(0, 8, n)
    tgt = rng.uniform(lo, hi, (n, 3))
    org = rng.normal(0, 1, (n, 3)); org = 3.5 * org / np.linalg.norm(org, axis=-1, keepdims=True)
    o[:], d[:] = org, tgt - org
    k = kind == 1                                   # origin inside the domain
    o[k] = rng.uniform(lo, hi, (k.sum(), 3)); d[k] = rng.normal(0, 1, (k.sum(), 3))
    k = kind == 2                                   # one exactly-zero direction component
    ax = rng.integers(0, 3, n)
    d[k, ax[k]] = 0.0
    k = kind == 3                                   # tiny direction components around the 1e-3 switch
    d[k, ax[k]] = rng.choice([1e-6, 1e-4, 9.9e-4, 1.01e-3, 5e-3, -1e-5, -1.1e-3], k.sum())
    k = kind == 4                                   # origin on / next to a face, grazing along it
    o[k, ax[k]] = np.where(rng.uniform(0, 1, k.sum()) < 0.5, lo[ax[k]], hi[ax[k]]) + rng.normal(0, 2e-3, k.sum())
    d[k, ax[k]] = rng.normal(0, 3e-4, k.sum())
    k = kind == 5                                   # misses
    d[k] = org[k] + rng.normal(0, 0.3, (k.sum(), 3))
    d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.6, 1.4, (n, 1))
    d[kind == 2, ax[kind == 2]] = 0.0
    return o.astype(np.float32), d.astype(np.float32)


def test_spans_never_drop_a_sample_the_reference_evaluates():
    import kilo_oracle as K
    from xrnerf_amd import kilo
    rng = np.random.default_rng(99)
    gmin, gmax = np.float32(kilo.LEGO_GMIN), np.float32(kilo.LEGO_GMAX)
    lo_eps, hi_eps = gmin + F(0.001), gmax - F(0.001)
    n_s = 384
    total_in = total_span = 0
    for near, far in ((2.0, 6.0), (0.5, 8.0), (6.0, 2.0), (0.0, 4.0)):
        for _ in range(6):
            n = 20000
            o, d = rays(rng, n, gmin, gmax)
            nr, fr = np.full(n, near, np.float32), np.full(n, far, np.float32)
            lo, hi = spans_replica(o, d, nr, fr, lo_eps, hi_eps, n_s)
            z = K.get_pts(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32),            # the un-jittered lattice,
                          np.linspace(0., 1., n_s).astype(np.float32)[None])[0, :, 0]          # torch.linspace to 1 ulp
            zz = (nr[:, None] * (F(1) - z) + fr[:, None] * z).astype(np.float32)
            inside = np.ones((n, n_s), bool)
            for a in range(3):
                p = (o[:, a:a + 1] + d[:, a:a + 1] * zz).astype(np.float32)
                inside &= (p > lo_eps[a]) & (p < hi_eps[a])
            s = np.arange(n_s)[None, :]
            covered = (s >= lo[:, None]) & (s < hi[:, None])
            assert not (inside & ~covered).any(), 'a span dropped a sample inside the domain'
            total_in += int(inside.sum()); total_span += int(covered.sum())
    assert total_in > 1e6 and total_span < 0.5 * 24 * 20000 * n_s          # and the spans do prune (most samples are outside)
