#!/bin/bash
# usage (GPU box, via gpurun): tools/final_measure.sh <round-tag>   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
tag=${1:-r06}
mkdir -p gpurun_out
python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_default.json
EDITOR_DROP_SKIP=0 python bench.py --no-cpu-baseline --no-modes --no-eval 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_noskip.json
python bench.py --dtype f16x2 --no-cpu-baseline --no-modes 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_f16x2.json
python bench.py --dtype f16 --no-cpu-baseline --no-modes 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_f16.json
python bench.py --dtype f16x2s --no-cpu-baseline --no-modes 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_f16x2s.json
# the per-rank compute of the strong-scaling series (global batch 128 on 8 / 4 / 2 GPUs): one GPU at B = 16 / 32 / 64
for b in 16 32 64; do
  python bench.py --batch $b --no-cpu-baseline --no-modes --steps 12 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_b$b.json
done
# the launcher with one RCCL rank (what `python bench.py --gpus N` does for N > 1), fp32 and bf16 gradient wire
python bench.py --gpus 1 --spawn 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_spawn1.json
python bench.py --gpus 1 --spawn --grad-wire bf16 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_spawn1_wire16.json
for ps in RGBNT100 MSVR310 SYNTH4L; do
  python bench.py --preset $ps --no-cpu-baseline --no-modes 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_$(echo $ps | tr A-Z a-z).json
done
python bench.py --preset SYNTH4L --batch 64 --no-cpu-baseline --no-modes --no-replay 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_synth4l_b64.json
python bench.py --preset SYNTH4L --batch 64 --act-light --no-cpu-baseline --no-modes --no-replay 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_line_synth4l_b64_light.json
DBG_B=128 python tools/repro_check.py > gpurun_out/${tag}_repro_check.txt 2>&1
bash tools/prof.sh ${tag} --no-replay --no-h2d --no-modes --no-eval > gpurun_out/${tag}_prof.txt 2>&1
cp gpurun_out/prof_${tag}/kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv
EDITOR_WGRAD_STREAM=0 bash tools/prof.sh ${tag}s --no-replay --no-h2d --no-modes --no-eval > gpurun_out/${tag}_prof_serial.txt 2>&1
cp gpurun_out/prof_${tag}s/kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats_serial.csv
cp gpurun_out/prof_${tag}s/kernel_stats.hash gpurun_out/${tag}_bench_kernel_stats_serial.hash
EDITOR_WGRAD_STREAM=0 bash tools/prof.sh ${tag}x2 --dtype f16x2s --no-replay --no-h2d --no-modes --no-eval > gpurun_out/${tag}_prof_f16x2s.txt 2>&1
cp gpurun_out/prof_${tag}x2/kernel_stats.csv gpurun_out/${tag}_f16x2s_kernel_stats_serial.csv
EDITOR_WGRAD_STREAM=0 bash tools/prof.sh ${tag}h --dtype f16 --no-replay --no-h2d --no-modes --no-eval > gpurun_out/${tag}_prof_f16.txt 2>&1
cp gpurun_out/prof_${tag}h/kernel_stats.csv gpurun_out/${tag}_f16_kernel_stats_serial.csv
bash tools/pmc_mfma.sh > gpurun_out/${tag}_pmc_mfma.txt 2>&1
python tools/stagger_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_stagger_sweep.txt
bash tools/launch_count.sh ${tag} > /dev/null 2>&1
python tools/select_time.py > gpurun_out/${tag}_select_time.txt 2>&1
TAG=${tag} bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc_traffic.log 2>&1
GEMM_EPI=1 GEMM_SPLIT=1 python tools/gemm_bench.py > gpurun_out/${tag}_gemm_bench.txt 2>&1
python tools/attn_bench.py > gpurun_out/${tag}_attn_bench.txt 2>&1
head -c 700 gpurun_out/${tag}_bench_line_default.json; echo
for f in noskip f16x2 f16x2s f16 rgbnt100 msvr310 synth4l b16 b32 b64 spawn1 spawn1_wire16; do head -c 330 gpurun_out/${tag}_bench_line_$f.json; echo; done
tail -4 gpurun_out/${tag}_repro_check.txt
tail -3 gpurun_out/${tag}_pmc_traffic.log
