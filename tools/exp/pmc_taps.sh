# FETCH_SIZE / WRITE_SIZE of the tap-list kernels on the microbench shapes (separate passes, PMC only)
R=${GRAFT_REPO_ROOT:-$PWD}
P=$R/gpurun_out/pmc_taps
rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$P/fetch" -- python3 $R/tools/microbench.py ${MB_CASE:-generic} > "$P/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$P/write" -- python3 $R/tools/microbench.py ${MB_CASE:-generic} > "$P/write.log" 2>&1
find "$P" -name "*_kernel_trace.csv" -delete
