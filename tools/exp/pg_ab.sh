#!/bin/bash
# Joint C3 step with and without a (world-size-1) RCCL process group, interleaved in ONE job per variant and traced:
# do RCCL's kernels overlap march_kernel?  bash tools/exp/pg_ab.sh   (through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
P=$R/gpurun_out/pg_ab
rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for tag in nogroup group nogroup2 group2; do
  if [[ $tag == group* ]]; then export PRE_BENCH_FORCE_GROUP=1; else unset PRE_BENCH_FORCE_GROUP; fi
  timeout -k 10 300 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$P/$tag.log" 2>&1
  tail -1 "$P/$tag.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'ms_per_step', round(d['ms_per_step'],2), 'eval launch ms', round(d['roofline']['avg_launch_ms'],2), 'full-pass step', round(d['ms_per_step_full_score_pass'],1))"
done
export PRE_BENCH_FORCE_GROUP=1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$P/trace" -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$P/trace.log" 2>&1
python3 - "$P" <<'PY'
import csv, glob, sys
P = sys.argv[1]
f = glob.glob(P + "/trace/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
rows.sort()
nc = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
mk = [r for r in rows if "march_kernel" in r[2]]
print("nccl/rccl kernels:", len(nc), "march launches:", len(mk))
ov = 0
for a in nc:
    for b in mk:
        o = min(a[1], b[1]) - max(a[0], b[0])
        if o > 0:
            ov += 1
            print(f"  OVERLAP {a[2]} [{(a[1]-a[0])/1e3:.1f} us] with march_kernel [{(b[1]-b[0])/1e6:.2f} ms]: {o/1e3:.1f} us")
print("overlapping pairs:", ov)
for a in nc[:12]:
    print(f"  {a[2]}  {(a[1]-a[0])/1e3:.1f} us")
# gaps between consecutive kernels of a step (idle device time)
names = {}
for r in rows:
    names.setdefault(r[2], []).append((r[1]-r[0])/1e6)
for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  {k:60s} n={len(v):4d} avg {sum(v)/len(v):8.3f} ms total {sum(v):9.2f} ms")
PY
find "$P" -name "*.csv" -size +1M -delete
