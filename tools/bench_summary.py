import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"failed",e); continue
    r=d['roofline']; t=d.get('tensor_mesh')
    print(f.split('/')[-1], "u: %.3f ms/step FJ %.1f us (%.3f) F %.1f us | spmv %.1f | path %.3f"%(d['ms_per_step'], 1e3*r['ms_per_launch'], r['frac'], 1e3*r['ms_residual_only'], 1e3*d['roofline_other']['ms_per_launch'], d['assembly_plus_spmv']['frac']), end='')
    if t: print(" || t: %.3f ms/step FJ %.1f us (%.3f) F %.1f us path %.3f"%(t['ms_per_step'], 1e3*t['roofline']['ms_per_launch'], t['roofline']['frac'], 1e3*t['roofline']['ms_residual_only'], t['assembly_plus_spmv']['frac']))
    else: print()
