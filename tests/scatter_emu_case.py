"""TEST INFRASTRUCTURE: one process = one setting of the scatter's test switch (XR_SC_TEST = "min_n=..,block=..,rl=..,rl_chunks=.."
is read once per process).  Runs xr_hashgrid_bwd of the kernels' host build (tests/hip_emu) against
oracle/ngp_oracle.c on `n` positions drawn as `mode` and prints one line per check; exit code 0 = all within tolerance.
usage: python tests/scatter_emu_case.py <n> <rand|rays|cluster|faces>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'hip_emu'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np   # noqa: E402
import torch         # noqa: E402


def positions(n, mode, rng):
    x = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    if mode == 'cluster':            # every sample in one cell: sub-bins overflow into the overflow lists
        x[:] = np.array([0.4371, 0.5113, 0.6207], np.float32)
        x[: n // 3] += rng.uniform(0, 1e-3, (n // 3, 3)).astype(np.float32)
    elif mode == 'rays':             # 20 consecutive samples per ray, sqrt(3)/1024 apart: the run-length paths
        nr = n // 20
        o3 = rng.uniform(0.2, 0.8, (nr, 3)).astype(np.float32)
        d3 = rng.normal(0, 1, (nr, 3)).astype(np.float32)
        d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
        k = np.arange(nr * 20)
        x[:nr * 20] = np.clip(o3[k // 20] + (np.float32(0.0017) * (k % 20).astype(np.float32))[:, None] * d3[k // 20], 0, 1)
    elif mode == 'faces':            # piled on the domain's upper faces: tcnn's linear index leaves the lattice row / wraps around
        x[:, 0] = 1.0
        x[: n // 2, 1] = 1.0
        x[: n // 4, 2] = 1.0
    x[0] = [0.0, 1.0, 0.5]
    x[1] = [1.0, 1.0, 1.0]           # index wrap-around at the upper corner of the domain
    return x


def main(n, mode):
    import emulib
    import oracle as O
    ok = True
    with emulib.emulated_ops():
        from xrnerf_amd import ops
        meta, om = ops.GridMeta(), O.GridMeta()
        rng = np.random.default_rng(n)
        x = positions(n, mode, rng)
        dy = rng.normal(0, 1, (n, 32)).astype(np.float32)
        dy[5:9] = 0
        ref = O.hashgrid_bwd(x, dy, om)
        ld = (n + 63) // 64 * 64
        dt = torch.zeros((32, ld))
        dt[:, :n] = torch.from_numpy(dy).t()
        dt = dt.contiguous()
        tx = torch.from_numpy(x)

        def report(tag, g, refv):
            nonlocal ok
            err = np.abs(g.numpy() - refv)
            tol = 2e-5 * max(1.0, float(np.abs(refv).max()))
            bad = [lv for lv in range(16) if err[2 * int(meta.offset[lv]):2 * int(meta.offset[lv + 1])].max() > tol]
            ok = ok and not bad
            print('%-26s max err %.3e (|ref| max %.3e) levels out of tolerance: %s' % (tag, err.max(), np.abs(refv).max(), bad))

        for ow in (False, True):
            g = torch.full((meta.n_params,), 7.0) if ow else torch.zeros(meta.n_params)
            ops.hashgrid_bwd(tx, dt, meta, g, overwrite=ow)
            report('overwrite=%s' % ow, g, ref)
        g = torch.zeros(meta.n_params)
        ops.hashgrid_bwd(tx, dt, meta, g, levels=(8, 16))
        ops.hashgrid_bwd(tx, dt, meta, g, levels=(0, 8))
        report('levels 8-16, then 0-8', g, ref)
        g = torch.full((meta.n_params,), 3.0)
        ops.hashgrid_bwd(tx, dt, meta, g, levels=(3, 16), overwrite=True)
        ops.hashgrid_bwd(tx, dt, meta, g, levels=(0, 3), overwrite=True)
        report('overwrite 3-16, then 0-3', g, ref)
        # one binned level beside three run-length levels (fewer bin workgroups per sample block than run-length levels whose maxima
        # they record: the other mapping of xr_scatter.hip's k_scatter_bin3)
        g = torch.full((meta.n_params,), 3.0)
        ops.hashgrid_bwd(tx, dt, meta, g, levels=(0, 4), overwrite=True)
        ops.hashgrid_bwd(tx, dt, meta, g, levels=(4, 16), overwrite=True)
        report('overwrite 0-4, then 4-16', g, ref)
        # the fixed-point scale of the LDS sums follows the gradients' magnitude (xr_scatter.hip, S3_FIX): the same gradients times 2^-70
        g = torch.full((meta.n_params,), 3.0)
        ops.hashgrid_bwd(tx, (dt * 2.0 ** -70).contiguous(), meta, g, overwrite=True)
        report('gradients x 2^-70', g * 2.0 ** 70, ref)
        live = np.flatnonzero(rng.uniform(size=n) < 0.45).astype(np.int32)
        dy2 = np.zeros_like(dy)
        dy2[live] = dy[live]
        rows = torch.zeros(n, dtype=torch.int32)
        rows[:len(live)] = torch.from_numpy(live)
        dtp = dt.clone()
        dtp[:, np.setdiff1d(np.arange(n), live)] = float('nan')       # rows outside the list must never be read
        g = torch.full((meta.n_params,), 5.0)
        ops.hashgrid_bwd(tx, dtp.contiguous(), meta, g, live=(rows, torch.tensor([len(live), 0, 0, 0], dtype=torch.int32)), overwrite=True)
        report('live-row list', g, O.hashgrid_bwd(x, dy2, om))
        dy3 = dy.copy()
        dy3[n // 3:] = 0
        g = torch.zeros(meta.n_params)
        ops.hashgrid_bwd(tx, dt, meta, g, n_dev=torch.tensor([n // 3], dtype=torch.int32))
        report('device-side count n/3', g, O.hashgrid_bwd(x, dy3, om))
        g = torch.full((meta.n_params,), 9.0)
        ops.hashgrid_bwd(tx, dt, meta, g, n_dev=torch.tensor([0], dtype=torch.int32), overwrite=True)
        report('device-side count 0', g, np.zeros_like(ref))
        # the optimiser's update applied by the scatter itself (xr_hashgrid_bwd_adam) = scatter (overwrite) + xr_adam_step_multi,
        # bit for bit, on a live-row list; two steps so that m / v / ema carry over
        if ops.hashgrid_bwd_adam_supported(n, meta):
            P = [torch.from_numpy(rng.uniform(-1e-4, 1e-4, meta.n_params).astype(np.float32)) for _ in range(2)]
            P[1] = P[0].clone()
            M, V, E = ([torch.zeros(meta.n_params) for _ in range(2)] for _ in range(3))
            for k in range(2):
                E[k].copy_(P[k])
            live_t = (rows, torch.tensor([len(live), 0, 0, 0], dtype=torch.int32))
            same = True
            for step in (1, 2):
                mom = min(0.05, step / (100.0 + step - 1))
                g = torch.full((meta.n_params,), 5.0)
                ops.hashgrid_bwd(tx, dtp.contiguous(), meta, g, live=live_t, overwrite=True)
                ops.adam_step_multi([P[0]], [g], [M[0]], [V[0]], step, 1e-2, 0.9, 0.99, 1e-15, 1e-6, [E[0]], mom)
                ops.hashgrid_bwd_adam(tx, dtp.contiguous(), meta, ops.adam_fuse(P[1], M[1], V[1], E[1], step, 1e-2, 0.9, 0.99, 1e-15, 1e-6, mom),
                                      live=live_t)
                same = same and all(torch.equal(a[0], a[1]) for a in (P, M, V, E))
            moved = float((P[0] - E[0]).abs().max()) > 0
            ok = ok and same and moved
            print('%-26s parameters / m / v / ema identical to scatter + optimiser launch: %s' % ('fused optimiser update', same and moved))
        else:
            print('%-26s not available with these switches' % 'fused optimiser update')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main(int(sys.argv[1]), sys.argv[2]))
