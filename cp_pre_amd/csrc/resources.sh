#!/bin/bash
# Compact per-kernel resource table (VGPR / spill / scratch / LDS / occupancy) from hipcc remarks.
cd "$(dirname "$0")"
for f in "${@:-star_march.hip calib.hip stencil_generic.hip}"; do
  for src in $f; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $src -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /VGPR Spill/ {sp=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
       /Occupancy/ {oc=$(NF-1)} /LDS Size/ {lds=$(NF-1); printf "%-4s v=%-4s spill=%-4s scratch=%-6s occ=%-3s lds=%-7s ", "", v, sp, sc, oc, lds; system("echo " name " | c++filt | cut -c1-110")}'
  done
done
