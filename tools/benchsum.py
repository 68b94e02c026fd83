import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('f32 %.4g rays/s  %.4f ms/step  normal %s refresh %s' % (d["value"], d["ms_per_step"], d.get("device_ms_normal_iteration"), d.get("device_ms_refresh_iteration")))
for k,v in (d.get("roofline_kernels") or {}).items(): print("   %-22s %s" % (k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("avg_launch_us","avg_us","frac","achieved")}))
f=d.get("ngp_f16_mlp_mode")
if f: print('f16 %.4g rays/s  %.4f ms/step' % (f["value"], f["ms_per_step"]), {k:round(v["avg_launch_us"],1) for k,v in f["kernels"].items()})
