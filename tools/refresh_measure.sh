#!/bin/bash
# usage (GPU box, via gpurun): tools/refresh_measure.sh <round-tag>
# The part of tools/final_measure.sh that a change to the step's non-GEMM kernels touches: every bench line, the reproducibility check,
# the two kernel profiles, the launch count, the memory-traffic counters, the attention timings.  (The GEMM probes are unaffected.)
tag=${1:-r04}
mkdir -p gpurun_out
python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_default.json
for dt in f16x2 f16 f16x2s; do
  python bench.py --dtype $dt --no-cpu-baseline --no-modes 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_$dt.json
done
for b in 16 32 64; do
  python bench.py --batch $b --no-cpu-baseline --no-modes --steps 12 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_b$b.json
done
python bench.py --gpus 1 --spawn 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_spawn1.json
for ps in RGBNT100 MSVR310 SYNTH4L; do
  python bench.py --preset $ps --no-cpu-baseline --no-modes 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_$(echo $ps | tr A-Z a-z).json
done
DBG_B=128 python tools/repro_check.py > gpurun_out/${tag}_repro_check.txt 2>&1
bash tools/prof.sh ${tag} --no-replay --no-h2d --no-modes --no-eval > gpurun_out/${tag}_prof.txt 2>&1
cp gpurun_out/prof_${tag}/kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv
EDITOR_WGRAD_STREAM=0 bash tools/prof.sh ${tag}s --no-replay --no-h2d --no-modes --no-eval > gpurun_out/${tag}_prof_serial.txt 2>&1
cp gpurun_out/prof_${tag}s/kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats_serial.csv
bash tools/launch_count.sh ${tag} > /dev/null 2>&1
TAG=${tag} bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc_traffic.log 2>&1
python tools/attn_bench.py > gpurun_out/${tag}_attn_bench.txt 2>&1
head -c 700 gpurun_out/${tag}_bench_line_default.json; echo
for f in f16x2 f16x2s f16 rgbnt100 msvr310 synth4l b16 b32 b64 spawn1; do head -c 330 gpurun_out/${tag}_bench_line_$f.json; echo; done
tail -4 gpurun_out/${tag}_repro_check.txt
tail -3 gpurun_out/${tag}_pmc_traffic.log
