#!/bin/bash
# SQ / HBM counters of one kernel (name substring) of any program, one counter group per rocprofv3 pass:
#   bash tools/exp/pmc_kernel.sh <tag> <kernel substring> <program args after python3 ...>
#   e.g. bash tools/exp/pmc_kernel.sh ind_nt march_kernel tools/exp/eval_job.py induction nt 1024x64x256x256
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; kern=$2; shift 2
P=$R/gpurun_out/pmc_$tag
rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-40)
  ( cd "$R" && timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$P/$g" -o x -- python3 "$@" > "$P/$g.log" 2>&1 )
done
python3 - "$P" "$kern" <<'PY'
import csv, glob, sys, collections
P, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(P + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kern not in k: continue
        acc[k.replace("(anonymous namespace)::", "")[:90]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        vals = list(v.values())
        print(f"   {c:30s} dispatches={len(vals)} mean={sum(vals)/len(vals):.5g}")
PY
grep -h "GB/s" "$P"/*.log | head -2
find "$P" -name "*.csv" -size +2M -delete
