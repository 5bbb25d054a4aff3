#!/bin/bash
# usage (GPU box): bash tools/gemm_bound_probe.sh  -> gpurun_out/gemm_bound_probe.txt : the probe at the default clock and, if the
# box lets us, under a 1 600 MHz cap (rocm-smi --setperfdeterminism)
mkdir -p gpurun_out
out=gpurun_out/gemm_bound_probe.txt
echo "== default clocks" > $out
python tools/gemm_bound_probe.py >> $out 2>&1
echo "== rocm-smi --setperfdeterminism 1600" >> $out
if /opt/rocm/bin/rocm-smi --setperfdeterminism 1600 >> $out 2>&1; then
  /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i sclk >> $out
  python tools/gemm_bound_probe.py >> $out 2>&1
  /opt/rocm/bin/rocm-smi --resetperfdeterminism >> $out 2>&1
fi
cat $out
