"""per training iteration (k_adam_multi to k_adam_multi): span, idle time of the main queue between its kernels,
and the gap between Adam and the first main-queue kernel of the next iteration"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adam_multi')]
mainq = rows[idx[0]]['Queue_Id']
out = []
for a, b in zip(idx[:-1], idx[1:]):
    ks = [r for r in rows[a:b + 1] if r['Queue_Id'] == mainq]
    span = (int(ks[-1]['End_Timestamp']) - int(ks[0]['End_Timestamp'])) / 1e3
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks[1:]) / 1e3
    front = (int(ks[1]['Start_Timestamp']) - int(ks[0]['End_Timestamp'])) / 1e3
    upd = any(r['Kernel_Name'].startswith('k6_') for r in rows[a:b])
    out.append((span, busy, front, upd))
for i, (s, bz, f, u) in enumerate(out[-40:]):
    print('it %3d  span %7.1f  busy %7.1f  idle %6.1f  front gap %6.1f %s' % (i, s, bz, s - bz, f, 'UPDATE' if u else ''))
n = [o for o in out[-40:] if not o[3]]
print('non-update mean: span %.1f busy %.1f idle %.1f front %.1f' % tuple(sum(o[k] for o in n) / len(n) for k in (0, 1, 2, 2)))
