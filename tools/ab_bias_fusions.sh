# A/B on one box: the qkv bias gradient from the attention backward's accumulators (EDITOR_ATTN_COLSUM) and the fc2-side 16-bit
# gradient copy + its column sums from the next block's LayerNorm-1 backward (EDITOR_HANDOFF_CAST); default both on
for r in 1 2; do
for v in "0 0" "1 0" "0 1" "1 1"; do
set -- $v
EDITOR_ATTN_COLSUM=$1 EDITOR_HANDOFF_CAST=$2 python bench.py --no-cpu-baseline --no-modes --no-eval 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ATTN_COLSUM=$1 HANDOFF_CAST=$2', 'ms', d['ms_per_step'], 'replay', d.get('replay_only'), 'frac', r['frac'], 'loss', d['config']['loss'])"
done; done
