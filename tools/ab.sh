#!/bin/bash
# A/B of environment settings with repetitions: tools/ab.sh REPS "ENV=a" "ENV=b" ...   (prints min/median of each stage)
reps=$1; shift
for cfg in "$@"; do
  for i in $(seq $reps); do env $cfg python bench.py --no-cpu-baseline 2>/dev/null; done | python -c "
import json,sys,statistics as st
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
f=lambda k:[r['roofline']['per_stage'][k]['ms'] for r in rows]
print('$cfg', 'step min %.3f med %.3f |' % (min(r['ms_per_step'] for r in rows), st.median(r['ms_per_step'] for r in rows)), ' '.join('%s min %.3f med %.3f' % (k, min(f(k)), st.median(f(k))) for k in ('splat','blur','slice')), '| build %.2f' % min(r['lattice_build_warm_ms'] for r in rows))"
done
