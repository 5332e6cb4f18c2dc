for cfg in "PHL_TILE_P=256" "PHL_TILE_P=128" "PHL_TILE_P=64" "PHL_TILE_P=256"; do
  for i in 1 2; do env $cfg python bench.py --workload band8 --no-cpu-baseline --no-regimes --no-mean-field --steps 60 --warmup 20 2>/dev/null; done | python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
f=lambda k:min(r['roofline']['per_stage'][k]['ms'] for r in rows)
print('$cfg', 'step %.4f' % min(r['ms_per_step'] for r in rows), 'splat %.4f blur %.4f slice %.4f' % (f('splat'),f('blur'),f('slice')), rows[0]['tiles'])"
done
