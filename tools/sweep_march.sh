#!/bin/bash
# experiment: sweep the march-kernel tuning knob (PRE_MARCH_VAR) on the eval microbench
for v in 0 1 2 3 4 5 6; do
  echo "== PRE_MARCH_VAR=$v"
  PRE_MARCH_VAR=$v timeout -k 10 300 python tools/microbench.py eval 2>&1 | grep -E "ns_momentum|wave|mhd_momentum|mhd_cont" | head -8
done
