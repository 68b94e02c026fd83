"""HBM traffic per launch from rocprofv3 PMC passes of `bench.py` (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
WRITE_SIZE cannot share a pass; both are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced
reads at 64 B -> the true read volume lies between the raw figure and twice it).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [hit/miss csv] > profiles/rNN_pmc_traffic.json

Per kernel: mean over the steady-state launches (the 16 launches before the last one of each kernel that runs
once per training step), plus the sums over the kernels that make up one C-ABI entry point."""
import collections, csv, json, sys

ENTRY = {   # entry point -> kernels launched by it (include/xrnerf_mi355.h)
    'xr_hashgrid_bwd': ['k_scatter_bin2', 'k_scatter_accum2', 'k_scatter_bin', 'k_scatter_accum', 'k_hashgrid_bwd', 'k_reduce_replicas',
                        'void k_scatter_bin3<4096>', 'void k_scatter_bin3<2048>', 'void k_scatter_bin3<1024>', 'void k_scatter_accum3<13>', 'void k_scatter_accum3<13, 512>', 'void k_scatter_accum3<13, 1024>', 'k_scatter_dense_rl', 'void k_scatter_acc<13, 512>', 'void k_scatter_acc<13, 1024>',
                        'k_scatter_fold'],
    'xr_hashgrid_fwd': ['k_hashgrid_fwd'],
    'xr_nerf_mlp_bwd': ['k_nerf_mlp_bwd_1_2', 'void k_nerf_mlp_bwd_1_2<true>', 'void k_nerf_mlp_bwd_1_2<false>', 'k_reduce_partials'] +
                       ['void k_nerf_mlp_bwd_1_2<%s, %d>' % (l, m) for l in ('true', 'false') for m in range(5)] +
                       ['void k_nerf_mlp_bwd_deep<true>', 'void k_nerf_mlp_bwd_deep<false>'],
    'xr_live_rows': ['k_live_count', 'k_live_fill'],
    'xr_composite_train': ['k_composite_train', 'k_composite_train_w'],
    'xr_nerf_mlp_fwd': ['void k_nerf_mlp_fwd<1, 2, true>', 'void k_nerf_mlp_fwd_b3<true>', 'void k_nerf_mlp_fwd_h2<true>', 'void k_nerf_mlp_fwd_h2<true, false>', 'void k_nerf_mlp_fwd_h2<true, true>', 'void k_nerf_mlp_fwd_deep<true>'],
    'xr_calc_rgb_backward': ['k_composite_bwd'],
    'xr_adam_step': ['k_adam_multi', 'void k_adam_multi<true>', 'void k_adam_multi<false>'],
}


def per_kernel(path):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].split('(')[0]
        rows[name][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    out = {}
    for name, ctrs in rows.items():
        out[name] = {}
        for c, vals in ctrs.items():
            vals.sort()
            tail = [v for _, v in vals[-17:-1]] or [v for _, v in vals]
            out[name][c] = (sum(tail) / len(tail), len(tail))
    return out


def show(path):
    d = json.load(open(path))
    for k in ('xr_hashgrid_bwd', 'xr_hashgrid_fwd', 'xr_nerf_mlp_bwd', 'xr_nerf_mlp_fwd', 'xr_adam_step', 'xr_live_rows', 'xr_composite_train'):
        if k in d:
            print(k, {a: (round(b / 1e6, 1) if isinstance(b, float) and b > 1e4 else b) for a, b in d[k].items()})


def main():
    if sys.argv[1] == '--print':
        return show(sys.argv[2])
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    hm = per_kernel(sys.argv[3]) if len(sys.argv) > 3 else {}
    res = {}
    for name in sorted(set(fetch) | set(write)):
        f = fetch.get(name, {}).get('FETCH_SIZE', (0.0, 0))
        w = write.get(name, {}).get('WRITE_SIZE', (0.0, 0))
        e = {'fetch_kib_raw': f[0], 'write_kib_raw': w[0], 'bytes_raw': 1024.0 * (f[0] + w[0]),
             'bytes_fetch_x2': 1024.0 * (2 * f[0] + w[0]), 'launches_averaged': max(f[1], w[1])}
        h = hm.get(name, {})
        if 'TCC_HIT_sum' in h and 'TCC_MISS_sum' in h:
            e['l2_hit_rate'] = h['TCC_HIT_sum'][0] / max(h['TCC_HIT_sum'][0] + h['TCC_MISS_sum'][0], 1.0)
        res[name] = e
    for entry, ks in ENTRY.items():
        have = [res[k] for k in ks if k in res]
        if have:
            res[entry] = {'kernels': [k for k in ks if k in res], 'bytes_raw': sum(h['bytes_raw'] for h in have),
                          'bytes_fetch_x2': sum(h['bytes_fetch_x2'] for h in have)}
    json.dump(res, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
