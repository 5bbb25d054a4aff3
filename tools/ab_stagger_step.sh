#!/bin/bash
# usage (GPU box): tools/ab_stagger_step.sh > gpurun_out/<tag>_ab_stagger_step.txt
# Same-box A/B of the round-6 stagger switches inside the whole step (bare step: inputs resident, hipGraph replay): ms per step, two
# passes in alternating order.
run() { env "$@" python bench.py --no-cpu-baseline --no-modes --no-eval --no-replay --no-h2d --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"; }
for pass in 1 2; do
  echo "pass $pass  base $(run X=1)  fc2d32 $(run EDITOR_STAGGER_FC2D=32)  fc2d24 $(run EDITOR_STAGGER_FC2D=24)  proj8 $(run EDITOR_STAGGER_PROJ=8)  both $(run EDITOR_STAGGER_FC2D=32 EDITOR_STAGGER_PROJ=8)  base $(run X=1)"
done
