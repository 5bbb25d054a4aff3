#!/bin/bash
# usage: tools/res_usage.sh <file.hip> [extra hipcc flags]   -> VGPRs / SGPRs / scratch / spills per kernel (device-only compile, no GPU needed)
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics --cuda-device-only -c "$src" -o /tmp/res_usage.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/res_usage.txt
python3 - <<'PY'
import re, subprocess
name = None; rows = {}
for ln in open("/tmp/res_usage.txt"):
    m = re.search(r"Function Name: (\S+)", ln)
    if m: name = m.group(1); rows[name] = {}
    for k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize \\[bytes/lane\\]", "VGPRs Spill", "SGPRs Spill", "LDS Size \\[bytes/block\\]"):
        m = re.search(r"\s%s: (\d+)" % k, ln)
        if m and name: rows[name][k.split(" ")[0] + ("S" if "Spill" in k else "")] = int(m.group(1))
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
for n, (k, d) in zip(names, rows.items()):
    print("%4d v %3d s  scratch %4d  spill v%d s%d  %s" % (d.get("VGPRs", -1), d.get("TotalSGPRs", -1), d.get("ScratchSize", 0),
          d.get("VGPRsS", 0), d.get("SGPRsS", 0), n.replace("(anonymous namespace)::", "")[:120]))
PY
