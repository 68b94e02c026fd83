"""HIP runtime calls the host makes between the launch of k_adam_multi and the launch of the next k2_clip"""
import csv, sys
api = list(csv.DictReader(open(sys.argv[1])))
api.sort(key=lambda r: int(r['Start_Timestamp']))
ker = {r['Correlation_Id']: r['Kernel_Name'] for r in csv.DictReader(open(sys.argv[2]))}
names = [(r, ker.get(r['Correlation_Id'], '')) for r in api]
idx = [i for i, (r, k) in enumerate(names) if k.startswith('k_adam_multi')]
a = idx[-3]
t0 = int(names[a][0]['Start_Timestamp'])
for r, k in names[a:a + 400]:
    if r['Function'] in ('hipGetDevice', 'hipSetDevice', 'hipGetLastError', 'hipGetDeviceCount', 'hipDevicePrimaryCtxGetState', 'hipEventQuery'):
        continue
    print('%8.1f us  %-26s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, r['Function'], k[:40]))
    if k.startswith('k_hashgrid_fwd'):
        break
