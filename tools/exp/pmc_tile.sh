#!/bin/bash
# SQ / HBM counters of the per-cell select at (n, M): bash tools/exp/pmc_tile.sh 512 2097152 tag
R=${GRAFT_REPO_ROOT:-$PWD}
n=$1; M=$2; tag=${3:-tile}
P=$R/gpurun_out/pmc_$tag
rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$P/$g" -o x -- python3 $R/tools/exp/tile_probe.py $n $M 1 > "$P/$g.log" 2>&1
done
python3 - "$P" <<'PY'
import csv, glob, sys, collections
P = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(P + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "kth" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
find "$P" -name "*.csv" -size +2M -delete
