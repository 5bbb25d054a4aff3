for r in 1 2; do
for v in 0 1; do
EDITOR_REVERSE_ROWS=$v python bench.py --no-cpu-baseline --no-modes --no-eval 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('REV=$v', 'ms', d['ms_per_step'], 'replay', d.get('replay_only',{}).get('ms_per_step') if isinstance(d.get('replay_only'),dict) else d.get('replay_only'), 'frac', r['frac'], r.get('by_kind'))"
done; done
