#!/bin/bash
# kernel_regs.sh <file.hip> [name filter] [extra hipcc flags...]: VGPRs / spills / occupancy per kernel (compiler remarks)
f=$1; filt=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /VGPRs Spill/ {print name, "vgpr", v, "scratch", sc, "spill", $(NF-1), "occ", o}' | \
  c++filt | sed 's/([^)]*)//; s/void dmf:://' | grep -E "$filt"
