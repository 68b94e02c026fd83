"""Round-5 review item 5, the cheap probe: can the request-bound hash lookup run beside the issue-bound fused-MLP forward?
On marched Lego samples (ray-ordered, 2^18 rows, trained-size weights):
  A  lookup(all) -> MLP(all), one stream                      (what the step does)
  B  lookup(h1) -> MLP(h1) -> lookup(h2) -> MLP(h2), one stream  (what halving alone costs)
  C  main: lookup(h1) -> MLP(h1) -> [wait] -> MLP(h2);  side: [wait lookup(h1)] -> lookup(h2)   (lookup(h2) beside MLP(h1))
  D  the same with four quarters
  E  lookup(all) on one stream beside MLP(all) of the previous buffer on the other: the co-run time against the sum and the max
Kill criterion: C or D must beat A by more than 8 us to be worth two more event operations in the step.
usage: python tools/overlap_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from microbench_fwd3 import setup

np, torch, O, ops, S, dev, c, n, meta, table = setup()
n = n // 256 * 256
ld = (n + 63) // 64 * 64
enc = torch.empty((32, ld), device=dev)
enc2 = torch.empty((32, ld), device=dev)
raw = torch.empty((n, 4), device=dev)
raw2 = torch.empty((n, 4), device=dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
wd, wc = t(S.mlp_weights(32, 64, 1, 16, 4)), t(S.mlp_weights(32, 64, 2, 16, 5))
pos, dirs = c[:, :3], c[:, 4:7]
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream()


def lookup(a, b, e=enc):
    ops.hashgrid_fwd(table, pos, meta, enc_t=e, ld=ld, row0=a, count=b - a)


def mlp(a, b, e=enc, r=raw):
    ops.nerf_mlp_fwd(e, dirs, n, wd, wc, 1, 2, raw=r, row0=a, count=b - a)


def seq(parts):
    def f():
        for a, b in parts:
            lookup(a, b); mlp(a, b)
    return f


def piped(parts):
    evs = [(torch.cuda.Event(), torch.cuda.Event()) for _ in parts]

    def f():
        lookup(*parts[0])
        for i, (a, b) in enumerate(parts):
            if i + 1 < len(parts):
                e0, e1 = evs[i]
                e0.record(main)                        # lookup(i) is enqueued: the side stream may start lookup(i + 1)
                with torch.cuda.stream(side):
                    side.wait_event(e0)
                    lookup(*parts[i + 1])
                    e1.record(side)
            mlp(a, b)
            if i + 1 < len(parts):
                main.wait_event(evs[i][1])
    return f


def corun():
    e0, e1 = torch.cuda.Event(), torch.cuda.Event()

    def f():
        e0.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e0)
            lookup(0, n, enc2)
            e1.record(side)
        mlp(0, n, enc, raw2)
        main.wait_event(e1)
    return f


def timeit(f, reps=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


cut = lambda k: [(i * n // k, (i + 1) * n // k) for i in range(k)]
lookup(0, n); mlp(0, n)
ref = raw.clone()
res = {}
res['lookup alone'] = timeit(lambda: lookup(0, n))
res['MLP forward alone'] = timeit(lambda: mlp(0, n))
res['A one stream, whole batch'] = timeit(seq(cut(1)))
res['B one stream, two halves'] = timeit(seq(cut(2)))
res['C two streams, two halves'] = timeit(piped(cut(2)))
same_c = bool(torch.equal(raw, ref))
res['D two streams, four quarters'] = timeit(piped(cut(4)))
same_d = bool(torch.equal(raw, ref))
res['E lookup(all) beside MLP(all)'] = timeit(corun())
print('rows %d, forward arithmetic %s' % (n, ops.f32_forward()))
for k, v in res.items():
    print('  %-34s %7.1f us' % (k, v))
print('  outputs of C / D equal A bit for bit: %s / %s' % (same_c, same_d))
print('  E against lookup + MLP = %.1f us (no overlap) and max = %.1f us (perfect overlap)' % (res['lookup alone'] + res['MLP forward alone'],
                                                                                            max(res['lookup alone'], res['MLP forward alone'])))
