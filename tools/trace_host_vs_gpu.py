"""is the compute stream waiting for the host?  For the last iterations: host time of the hipLaunchKernel call of
k2_clip / k_adam_multi versus the GPU start of those kernels (rocprofv3 --kernel-trace --hip-runtime-trace)."""
import csv, sys
api = {r['Correlation_Id']: r for r in csv.DictReader(open(sys.argv[1])) if r['Function'] == 'hipLaunchKernel'}
ker = list(csv.DictReader(open(sys.argv[2])))
ker.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [r for r in ker if r['Kernel_Name'].startswith('k_adam_multi')]
cl = [r for r in ker if r['Kernel_Name'].startswith('k2_clip')]
for a in ad[-8:-1]:
    t0 = int(a['End_Timestamp'])
    nxt = [c for c in cl if int(c['Start_Timestamp']) > t0]
    if not nxt: continue
    c = nxt[0]
    ha, hc = api.get(a['Correlation_Id']), api.get(c['Correlation_Id'])
    print('adam: host launch %8.1f us before its GPU start | k2_clip: host launch at %+7.1f us, GPU start at %+7.1f us (0 = adam end)' % (
        (int(a['Start_Timestamp']) - int(ha['Start_Timestamp'])) / 1e3, (int(hc['Start_Timestamp']) - t0) / 1e3,
        (int(c['Start_Timestamp']) - t0) / 1e3))
