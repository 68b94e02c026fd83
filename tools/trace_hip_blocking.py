"""from a rocprofv3 hip_api_trace.csv: which HIP runtime calls take long on the host (blocking calls) in steady state"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]                       # steady state
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg[r['Function']]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for k, (n, t, m) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print('%-44s calls %6d  total %10.1f us  mean %8.2f  max %9.1f' % (k, n, t, t / n, m))
