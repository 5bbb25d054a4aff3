#!/bin/bash
# usage (GPU box, via gpurun): tools/final_measure.sh <round-tag>   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
tag=${1:-r02}
mkdir -p gpurun_out
python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_default.json
python bench.py --dtype f16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_f16.json
python bench.py --dtype f32 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_f32.json
for ps in RGBNT100 MSVR310 SYNTH4L; do
  python bench.py --preset $ps --no-cpu-baseline --no-h2d 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_$(echo $ps | tr A-Z a-z).json
done
DBG_B=128 python tools/repro_check.py > gpurun_out/${tag}_repro_check.txt 2>&1
bash tools/prof.sh ${tag} --no-replay --no-h2d > gpurun_out/${tag}_prof.txt 2>&1
cp gpurun_out/prof_${tag}/kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv
EDITOR_WGRAD_STREAM=0 bash tools/prof.sh ${tag}s --no-replay --no-h2d > gpurun_out/${tag}_prof_serial.txt 2>&1
cp gpurun_out/prof_${tag}s/kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats_serial.csv
bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc_traffic.log 2>&1
GEMM_EPI=1 python tools/gemm_bench.py > gpurun_out/${tag}_gemm_bench.txt 2>&1
head -c 600 gpurun_out/${tag}_bench_line_default.json; echo
head -c 300 gpurun_out/${tag}_bench_line_f16.json; echo
head -c 300 gpurun_out/${tag}_bench_line_f32.json; echo
for ps in rgbnt100 msvr310 synth4l; do head -c 260 gpurun_out/${tag}_bench_line_$ps.json; echo; done
tail -4 gpurun_out/${tag}_repro_check.txt
tail -3 gpurun_out/${tag}_pmc_traffic.log
