#!/bin/bash
# same-box A/B of the ping-pong kernel's tile-group height (EDITOR_GEMM_GM): variant libraries editor_amd/exp_gm<N>.so built with
#   hipcc ... -DEDITOR_GEMM_GM=<N> -c csrc/gemm_bf16.hip   (linked with the other objects); prints replay-only ms per step
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-modes --steps 20 2>/dev/null | tail -1 | python -c "import sys, json; j=json.loads(sys.stdin.readline()); print('gm=4 (built-in)', j['replay_only']['ms_per_step'], j['roofline']['frac'])"
  for gm in 2 8 16; do
    EDITOR_LIB_VARIANT=editor_amd/exp_gm$gm.so python tools/bench_variant.py --no-cpu-baseline --no-modes --steps 20 2>/dev/null | tail -1 | python -c "import sys, json; j=json.loads(sys.stdin.readline()); print('gm=$gm', j['replay_only']['ms_per_step'], j['roofline']['frac'])"
  done
done
