"""print the kernel sequence of one grid-update training iteration (the window between two k_adam_multi launches
that contains k6_generate) from a rocprofv3 kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adam_multi')]
wins = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any(rows[i]['Kernel_Name'].startswith('k6_') for i in range(a, b))]
a, b = wins[-1]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  +%8.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
print('iteration span %.1f us' % ((int(rows[b]['End_Timestamp']) - t0) / 1e3))
