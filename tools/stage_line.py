import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in r['per_stage'].items()}, 'build', d.get('lattice_build_warm_ms'), d.get('lattice_build_warm_ms_clean_table'))
