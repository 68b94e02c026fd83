"""mean per-dispatch value of every counter in a rocprofv3 counter_collection.csv for kernels whose name starts with
argv[2] (the last 20 dispatches: steady state)"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith(sys.argv[2])]
by = collections.defaultdict(list)
for r in rows:
    by[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in by.items():
    v = v[-20:]
    print('%-44s %16.0f   (n=%d)' % (k, sum(v) / len(v), len(v)))
