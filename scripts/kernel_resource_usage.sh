#!/bin/bash
# Register / scratch / LDS budget of every kernel of one source file (gfx950), from hipcc's own remarks.
# usage: scripts/kernel_resource_usage.sh gpu-physics-engine_amd/csrc/k_native.hip [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
name = None; rec = {}
def flush():
    if name: print("%-62s TotalSGPRs: %s VGPRs: %s AGPRs: %s ScratchSize [bytes/lane]: %s Occupancy [waves/SIMD]: %s LDS Size [bytes/block]: %s" % (name, rec.get("TotalSGPRs"), rec.get("VGPRs"), rec.get("AGPRs"), rec.get("ScratchSize [bytes/lane]"), rec.get("Occupancy [waves/SIMD]"), rec.get("LDS Size [bytes/block]")))
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        flush(); rec = {}
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z][^:]*): (\d+)", line)
    if m: rec[m.group(1).strip()] = m.group(2)
flush()
'
