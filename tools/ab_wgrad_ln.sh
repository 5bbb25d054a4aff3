# A/B on one box: LayerNorm-1's backward as a memory-bound role of the weight-gradient launch (EDITOR_WGRAD_LN=1, on 32 / 64 / 96 CUs)
# against the default (weight gradients on the side stream, LayerNorm-1 backward its own launch)
for r in 1 2; do
for v in "0 64" "1 32" "1 64" "1 96"; do
set -- $v
EDITOR_WGRAD_LN=$1 EDITOR_WGRAD_LN_CUS=$2 python bench.py --no-cpu-baseline --no-modes --no-eval 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('WGRAD_LN=$1 CUS=$2', 'ms', d['ms_per_step'], 'replay', (d.get('replay_only') or {}).get('ms_per_step'), 'frac', r['frac'], 'loss', d['config']['loss'])"
done; done
