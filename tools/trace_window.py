"""print the kernel sequence of one full training iteration (between two end-of-iteration kernels) from a
rocprofv3 kernel_trace.csv: start offset, duration, queue, name"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# an iteration ends with the table scatter's fold kernel (round 6: one in-order stream, the MLP tensors' update rides inside the scatter's
# binning launch); traces of earlier rounds end it with the k_adam_multi launch of the helper stream
_last = 'k_scatter_fold' if any('k_scatter_fold' in r['Kernel_Name'] for r in rows) else 'k_adam_multi'
idx = [i for i, r in enumerate(rows) if _last in r['Kernel_Name']]
arg = sys.argv[2] if len(sys.argv) > 2 else '-2'
if arg == 'spans':            # one line per window: its span and the longest kernel in it (to find the steady state)
    for w in range(1, len(idx)):
        a, b = idx[w - 1], idx[w]
        span = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3
        print('%4d %8.1f us  %d kernels' % (w, span, b - a))
    sys.exit(0)
if arg == 'refresh':          # the last window that holds a grid refresh (k9_ema)
    arg = str(max(w for w in range(1, len(idx)) if any('k9_ema' in r['Kernel_Name'] for r in rows[idx[w - 1]:idx[w]])))
which = int(float(arg) * len(idx)) if '.' in arg else int(arg)
which = max(1, min(len(idx) - 1, which)) if which >= 0 else which
print('# window %d of %d' % (which, len(idx)))
a, b = idx[which - 1], idx[which]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  +%8.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
print('iteration span %.1f us' % ((int(rows[b]['End_Timestamp']) - t0) / 1e3))
