#!/bin/bash
# Marginal C3 job with and without a (world-size-1) RCCL process group (time-major residual, planewise selects)
R=${GRAFT_REPO_ROOT:-$PWD}
P=$R/gpurun_out/marg_ab
rm -rf "$P"; mkdir -p "$P"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for tag in nogroup group nogroup2 group2; do
  if [[ $tag == group* ]]; then export PRE_BENCH_FORCE_GROUP=1; else unset PRE_BENCH_FORCE_GROUP; fi
  timeout -k 10 400 python3 $R/bench.py --mode marginal --steps 4 --warmup 1 --no-cpu-baseline > "$P/$tag.log" 2>&1
  tail -1 "$P/$tag.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'ms_per_step', round(d['ms_per_step'],2), 'eval launch ms', round(d['roofline']['avg_launch_ms'],2), d['config']['workload'][-90:])" || tail -5 "$P/$tag.log"
done
