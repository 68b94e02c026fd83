"""Is the table scatter bit-for-bit repeatable when another process shares the GPU?  The bin kernel ranks a sub-bin's items with LDS
atomics (their order follows the wave schedule) and the accumulate kernel sums them in fp64 -- exact, hence order-free, only while the
addends of an entry span < 2^(29 - log2 count).  Takes the scatter's inputs from the second step of a small training job, repeats
the launch K times with and without a second process training on the same GPU, and counts (launch, level) pairs whose slice differs
from the first launch's.   usage: python tools/scatter_determinism_probe.py [K]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd import ops
from xrnerf_amd.train import Trainer

dev = torch.device('cuda', 0)
if len(sys.argv) > 1 and sys.argv[1] == 'noise':
    tr = Trainer(dev, n_img=3, H=128, W=128, ema=False)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        tr.run(16); torch.cuda.synchronize()
    sys.exit(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tr = Trainer(dev, n_img=3, H=128, W=128, ema=False, native_loop=False, fuse_adam=False)
for _ in range(2):
    tr.step()
torch.cuda.synchronize()
net = tr.net
b = net._step_bufs[net._step_turn]
meta = net.mlp.embedder_pos.meta
coords = net.sampler.coords[:b.n_rows]
off = [2 * int(o) for o in meta.offset]
# words of the workspace head that hold sub-bin fill counts only: 13 binned levels x >= 32 partitions x (rows / 2048) sample blocks
COUNT_WORDS = 13 * 32 * max(1, (b.n_rows + 2047) // 2048)
print('rows', b.n_rows, 'live', int(b.live[1][0]) if b.live is not None else None)


def sweep(tag, alternate=False):
    """alternate: launches k and k + 1 scatter different gradients (the second one negated), each compared with its own first result --
    a launch that picked up anything left behind by the launch before it shows, which identical launches would hide"""
    refs, bad, worst = [None, None], [0] * meta.n_levels, 0.0
    dencs = [b.denc_t, -b.denc_t] if alternate else [b.denc_t, b.denc_t]
    cref, cbad, bref = None, 0, None
    for k in range(K):
        ref = refs[k & 1]
        g = torch.full((meta.n_params,), float('nan'), device=dev)
        ops.hashgrid_bwd(coords, dencs[k & 1], meta, g, live=b.live, overwrite=True)
        # the BIN kernel's outputs that do not depend on the wave schedule: the fill counts of every (level, partition, sample block)
        # sub-bin, at the head of the scatter's workspace (the item -> partition map is a function of the inputs: +-denc give the same counts)
        ws = ops._workspaces.get((str(dev), 'hgb'))
        if ws is not None:
            cnt = ws[:min(ws.numel(), 4 * COUNT_WORDS)].view(torch.int32).clone()
            if cref is None:
                cref = cnt
            elif not torch.equal(cnt, cref):
                cbad += 1
                d = torch.nonzero(cnt != cref).reshape(-1)
                print('   launch %d: %d sub-bin fill counts differ from the first launch (words %d..%d; first: %d instead of %d)'
                      % (k, d.numel(), int(d[0]), int(d[-1]), int(cnt[d[0]]), int(cref[d[0]])), flush=True)
        # the bin kernel's OTHER output, the items themselves: an order-free checksum per float4 component over the whole bin area (the order
        # of a sub-bin's items follows the wave schedule; slots behind the fill counts are never written).  Default layout only:
        # 13 binned levels, 2048-sample blocks, 262144 rows -> counts 399872 B, bins 13 x 1572864 items
        if ws is not None and os.environ.get('CHECK_BINS') and b.n_rows == 262144 and not alternate:
            # (the hashed levels' part: the two dense binned levels come first in the bin area and do overflow their sub-bins on coherent rays --
            # WHICH of a sub-bin's items go to the overflow list follows the schedule)
            items = ws[399872 + 2 * 1572864 * 16:399872 + 13 * 1572864 * 16].view(torch.int32).view(-1, 4)
            cs = items.to(torch.int64).sum(0)
            if k == 0:
                bref = cs
            elif not torch.equal(cs, bref):
                print('   launch %d: the bin area differs from the first launch in components %s (x = entry pair, y = feature-0 value, z = feature-1 '
                      'value, w = x weight): checksum deltas %s' % (k, [c for c in range(4) if int(cs[c]) != int(bref[c])], (cs - bref).tolist()), flush=True)
        if ref is None:
            refs[k & 1] = g
            continue
        ne = g.view(torch.int32) != ref.view(torch.int32)
        if bool(ne.any()):
            worst = max(worst, float(((g - ref).abs() / ref.abs().clamp_min(1e-30))[ne].max()))
            for l in range(meta.n_levels):
                m = ne[off[l]:off[l + 1]]
                if bool(m.any()):
                    bad[l] += 1
                    a, r = g[off[l]:off[l + 1]][m], ref[off[l]:off[l + 1]][m]
                    idx = torch.nonzero(m).reshape(-1)
                    # where inside the level: accumulate workgroups (2^13 entries = 16384 floats each) hit, floats per workgroup, features
                    wg = idx // 16384
                    uw, cw = torch.unique(wg, return_counts=True)
                    print('      %d accumulate workgroups hit (floats per workgroup: max %d, min %d); feature-0 / feature-1 floats %d / %d; '
                          'diff / ref of the first five: %s' % (uw.numel(), int(cw.max()), int(cw.min()), int((idx % 2 == 0).sum()), int((idx % 2 == 1).sum()),
                                                                 ' '.join('%.4g' % float(v) for v in ((a - r) / r.abs().clamp_min(1e-30))[:5])), flush=True)
                    print('   launch %d level %d: %d entries differ (first at %d, last at %d of %d), max |diff| %.3g where |ref| max %.3g (level max %.3g), nan %d'
                          % (k, l, int(m.sum()), int(idx[0]), int(idx[-1]), m.numel(), float((a - r).abs().nan_to_num(0).max()), float(r.abs().max()),
                             float(ref[off[l]:off[l + 1]].abs().max()), int(torch.isnan(a).sum())), flush=True)
    print(tag, 'launches', K, 'levels that differed from the first launch (count per level):', bad, 'worst relative difference %.3g' % worst,
          '| launches whose sub-bin fill counts differed:', cbad, flush=True)


def sweep_others(tag):
    """the same question for the other kernels of the step: gather, fused MLP forward and backward (live rows)"""
    mlp = net.mlp
    table, wd, wc = mlp.embedder_pos.params.detach(), mlp.density_net.params.detach(), mlp.color_net.params.detach()
    n = b.n_rows
    dirs = net.sampler.coords[:n, 4:7]
    ref, bad = None, {}
    for k in range(K):
        enc = ops.hashgrid_fwd(table, coords, meta)
        raw = ops.nerf_mlp_fwd(enc, dirs, n, wd, wc, 1, 2)
        gd, gc = torch.zeros_like(wd), torch.zeros_like(wc)
        de = torch.zeros_like(enc)
        ops.nerf_mlp_bwd(enc, dirs, n, wd, wc, 1, 2, b.draw, gd, gc, denc_t=de, live=b.live)
        cur = dict(gather=enc, mlp_forward=raw, mlp_backward_dw=torch.cat([gd, gc]), mlp_backward_dx=de)
        if ref is None:
            ref = cur
            continue
        for name in cur:
            ne = cur[name].view(torch.int32) != ref[name].view(torch.int32)
            if bool(ne.any()):
                bad[name] = bad.get(name, 0) + 1
                print('   launch %d %s: %d values differ, max |diff| %.3g (max |ref| %.3g)' % (k, name, int(ne.sum()), float((cur[name] - ref[name]).abs().max()), float(ref[name].abs().max())), flush=True)
    print(tag, 'launches', K, 'kernels that differed from the first launch:', bad or 'none', flush=True)


NOISE = int(os.environ.get('NOISE_PROCS', '1'))
if not os.environ.get('SHARED_ONLY'):
    sweep('alone, same input every launch        ')
    sweep('alone, alternating inputs             ', True)
# NOISE_CU_MASK (e.g. "0:128-255", with HSA_CU_MASK=0:0-127 for this process): the competitors on their own compute units
nenv = dict(os.environ)
if os.environ.get('NOISE_CU_MASK'):
    nenv['HSA_CU_MASK'] = os.environ['NOISE_CU_MASK']
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'noise', os.environ.get('NOISE_SECONDS', '110')], env=nenv) for _ in range(NOISE)]
time.sleep(15)
sweep('GPU shared (%d other), same input every launch   ' % NOISE)
if not os.environ.get('SHARED_ONLY'):
    sweep('GPU shared, alternating inputs        ', True)
for p in ps:
    p.wait()
