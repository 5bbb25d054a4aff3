# A/B on one box: the HMA head's per-modality blocks as one grouped node (EDITOR_GROUP_BLOCKS=1, default) or three nodes (=0)
for r in 1 2; do
for v in 0 1; do
EDITOR_GROUP_BLOCKS=$v python bench.py --no-cpu-baseline --no-modes --no-eval 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('GROUP_BLOCKS=$v', 'ms', d['ms_per_step'], 'replay', d.get('replay_only'), 'frac', r['frac'], 'loss', d['config']['loss'])"
done; done
