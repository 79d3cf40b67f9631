#!/bin/bash
# Collect the round's rocprofv3 evidence on the MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh'
# then   python tools/distill_profiles.py r06   turns gpurun_out/prof/* into profiles/r06/*.
# Trace and counter passes are separate runs (PMC is never combined with other trace domains).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
P=$R/gpurun_out/prof
PART=${PART:-all}       # a: the C3 passes, b: the other configs (a gpurun call is at most 20 minutes: run the two apart), c1: C1 alone
[ "$PART" != "b" ] && [ "$PART" != "c1" ] && rm -rf "$P"
mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, rocprof args..., -- bench args
    tag=$1; shift
    echo "== $tag" >&2
    timeout -k 10 500 rocprofv3 "$@" > "$P/bench_$tag.log" 2>&1
}
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --no-parity"
if [ "$PART" != "b" ]; then
run trace    --kernel-trace --stats --output-format csv -d "$P/trace"   -- $B --steps 3 --warmup 1
run trace_m  --kernel-trace --stats --output-format csv -d "$P/trace_m" -- $B --steps 2 --warmup 1 --mode marginal
run fetch    --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch" -- $B --steps 1 --warmup 0
run write    --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write" -- $B --steps 1 --warmup 0
# SQ counters on a quarter batch (same per-wave behaviour, shorter run); one small group per pass
run sq1      --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$P/sq1" -- $B --steps 1 --warmup 0 --batch 1024
run sq2      --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$P/sq2" -- $B --steps 1 --warmup 0 --batch 1024
# the per-cell select (marginal mode): instruction mix and HBM fetch, quarter batch
run sq1m     --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$P/sq1m" -- $B --steps 1 --warmup 0 --batch 1024 --mode marginal
run sq2m     --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$P/sq2m" -- $B --steps 1 --warmup 0 --batch 1024 --mode marginal
run fetchm   --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetchm" -- $B --steps 1 --warmup 0 --batch 1024 --mode marginal
# C3 marginal at the full batch: HBM traffic of the residual kernel writing |res| into the row-padded score buffer
run fetch_m  --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch_m" -- $B --steps 1 --warmup 0 --mode marginal
run write_m  --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write_m" -- $B --steps 1 --warmup 0 --mode marginal
fi
if [ "$PART" = "c1" ]; then
    c=c1
    run trace_$c --kernel-trace --stats --output-format csv -d "$P/trace_$c" -- $B --config $c --steps 3 --warmup 1
    run fetch_$c --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch_$c" -- $B --config $c --steps 1 --warmup 0
    run write_$c --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write_$c" -- $B --config $c --steps 1 --warmup 0
    echo done >&2; exit 0
fi
if [ "$PART" != "a" ]; then
for c in c1 c2 c5; do
    run trace_$c --kernel-trace --stats --output-format csv -d "$P/trace_$c" -- $B --config $c --steps 3 --warmup 1
    run fetch_$c --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch_$c" -- $B --config $c --steps 1 --warmup 0
    run write_$c --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write_$c" -- $B --config $c --steps 1 --warmup 0
done
# C4: each of its five equations (Marginal/MHD_Residuals_CP.py:225-278)
for e in induction continuity momentum energy gauss; do
    run trace_c4_$e --kernel-trace --stats --output-format csv -d "$P/trace_c4_$e" -- $B --config c4 --equation $e --steps 3 --warmup 1
    run fetch_c4_$e --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch_c4_$e" -- $B --config c4 --equation $e --steps 1 --warmup 0
    run write_c4_$e --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write_c4_$e" -- $B --config c4 --equation $e --steps 1 --warmup 0
done
# C4 fed the reference callers' Nt-fastest views (Marginal/MHD_Residuals_CP.py:326-346): flat_march_kernel<...<3>>
for e in induction momentum; do
    run trace_c4_${e}_nt --kernel-trace --stats --output-format csv -d "$P/trace_c4_${e}_nt" -- $B --config c4 --equation $e --layout nt --steps 3 --warmup 1
    run fetch_c4_${e}_nt --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch_c4_${e}_nt" -- $B --config c4 --equation $e --layout nt --steps 1 --warmup 0
    run write_c4_${e}_nt --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write_c4_${e}_nt" -- $B --config c4 --equation $e --layout nt --steps 1 --warmup 0
done
# C5 at its single-GPU size [65536,200,512]
run trace_c5w --kernel-trace --stats --output-format csv -d "$P/trace_c5w" -- $B --config c5 --batch 65536 --steps 3 --warmup 1
run fetch_c5w --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch_c5w" -- $B --config c5 --batch 65536 --steps 1 --warmup 0
run write_c5w --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write_c5w" -- $B --config c5 --batch 65536 --steps 1 --warmup 0
# the six-field functors: instruction mix and waits
for e in momentum energy; do
    run sq1_c4_$e --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$P/sq1_c4_$e" -- $B --config c4 --equation $e --steps 1 --warmup 0
    run sq2_c4_$e --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$P/sq2_c4_$e" -- $B --config c4 --equation $e --steps 1 --warmup 0
done
fi
# gpurun copies back at most 64 MiB: the per-dispatch traces are not needed once rocprofv3 has written the statistics
find "$P" -name "*_kernel_trace.csv" -delete
echo done >&2
