#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs that a GPU call left under gpurun_out/prof/ into the small,
committed summaries under profiles/<round>/ (usage: python tools/distill_profiles.py r01)."""
import csv, glob, collections, json, shutil, os, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = f"profiles/{rnd}"
os.makedirs(out, exist_ok=True)
NAMES = (('march_kernel', 'march_kernel<NSMomentum<0>,8,64>'), ('joint_score', 'joint_score_kernel'), ('std_from_moments', 'std_from_moments_kernel'),
         ('moments_kernel', 'moments_kernel'), ('kth_axis0_pass<9', 'kth_axis0_pass<9,1>'), ('kth_axis0_pass<6', 'kth_axis0_pass<6,10>'),
         ('kth_axis0_pass<5', 'kth_axis0_pass<5,10>'), ('kth_kernel', 'kth_kernel'))
def short(k):
    for key, name in NAMES:
        if key in k:
            return name
def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None
for tag, src, log, cmd in (('joint', 'trace', 'bench_trace.log', '--steps 3'), ('marginal', 'trace_m', 'bench_trace_m.log', '--mode marginal --steps 2')):
    f = newest(f'gpurun_out/prof/{src}/runc/*_kernel_stats.csv')
    if not f:
        continue
    shutil.copy(f, f'{out}/bench_c3_{tag}_kernel_stats.csv')
    line = [l for l in open(f'gpurun_out/prof/{log}') if l.startswith('{"metric"')][-1]
    with open(f'{out}/bench_c3_{tag}_summary.txt', 'w') as o:
        o.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py {cmd} --warmup 1 --no-cpu-baseline   (MI355X)\n{line}\n")
        o.write(f"{'kernel':40s} {'calls':>6s} {'avg_ms':>10s} {'total_ms':>10s} {'%':>7s}\n")
        for r in csv.DictReader(open(f)):
            n = short(r['Name'])
            if n:
                o.write(f"{n:40s} {r['Calls']:>6s} {float(r['AverageNs'])/1e6:10.3f} {float(r['TotalDurationNs'])/1e6:10.1f} {float(r['Percentage']):7.2f}\n")
pm = {}
for name in ('fetch', 'write'):
    f = newest(f'gpurun_out/prof/{name}/runc/*_counter_collection.csv')
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k:
            agg[k][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in agg.items():
        per = list(d.values())
        pm[(name, k)] = (len(per), sum(per) / len(per))
if pm:
    B = 4096; cells = B * 10 * 512 * 512
    mk = 'march_kernel<NSMomentum<0>,8,64>'
    with open(f'{out}/pmc_hbm_c3.txt', 'w') as o:
        o.write("rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes)\n   -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline   (C3, joint, batch 4096; MI355X)\n")
        o.write("counter unit KiB.  gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane\ncoalesced streaming read -> x2; WRITE_SIZE is exact for 16 B/lane streaming stores.\n\n")
        for (name, k), (n, avg) in sorted(pm.items(), key=lambda x: (x[0][1], x[0][0])):
            gb = avg * 1024 / 1e9 * (2 if name == 'fetch' else 1)
            o.write(f"{name.upper()+'_SIZE':11s} {k:36s} dispatches={n:3d} avg_KiB={avg:.6g} corrected_GB_per_dispatch={gb:.3f}\n")
        f = pm[('fetch', mk)][1] * 1024 * 2; w = pm[('write', mk)][1] * 1024
        alg_r, alg_w = 12 * cells, 4 * cells * 8 // 10      # 10 planes read, 8 interior planes written
        o.write(f"\n{mk} per launch [4096,10,512,512]: algorithmic read {alg_r/1e9:.3f} GB + write {alg_w/1e9:.3f} GB = {(alg_r+alg_w)/1e9:.3f} GB\n")
        o.write(f"   measured HBM traffic  read {f/1e9:.3f} GB + write {w/1e9:.3f} GB = {(f+w)/1e9:.3f} GB  ({(f+w)/(alg_r+alg_w):.4f} x algorithmic)\n")
    json.dump({"workload": {"batch": 4096, "slab": 8, "nx": 512, "ny": 512}, "kernel": mk,
               "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "traffic_bytes_per_launch": f + w,
               "source": f"{out}/pmc_hbm_c3.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction)"},
              open(f'{out}/pmc_hbm_c3.json', 'w'), indent=1)
for f in sorted(os.listdir(out)):
    print(f)
