#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs that `tools/profile_round.sh` left under gpurun_out/prof/ into the small,
committed summaries under profiles/<round>/ (usage: python tools/distill_profiles.py r01)."""
import collections, csv, glob, json, os, re, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = f"profiles/{rnd}"
os.makedirs(out, exist_ok=True)
MK = "march_kernel<NSMomentum<0>,8,64>"


def short(k):
    """'void (anonymous namespace)::march_kernel<(anonymous namespace)::NSMomentum<0>, 8, 64, false>(...)' ->
    'march_kernel<NSMomentum<0>,8,64>'; torch/rccl kernels keep a trimmed name."""
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*$", "", k).strip().replace("> >", ">>")
    k = re.sub(r",\s*false,\s*true>$", ",SEG>", k)             # march_kernel<Fn,NR,TYQ,BC=false,SEG=true>
    k = re.sub(r",\s*false,\s*false>$", ">", k)
    k = re.sub(r",\s*(false|true)>$", lambda m: ">" if m.group(1) == "false" else ",BC>", k)
    return k.replace(", ", ",")[:70]


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def bench_line(log):
    lines = [l for l in open(log) if l.startswith('{"metric"')] if os.path.exists(log) else []
    return lines[-1].strip() if lines else "(bench line not captured)"


CMDS = {"trace": ("c3_joint", "--no-secondary --no-parity --steps 3 --warmup 1"),
        "trace_m": ("c3_marginal", "--no-secondary --no-parity --steps 2 --warmup 1 --mode marginal"),
        "trace_c1": ("c1", "--config c1 --steps 3 --warmup 1"), "trace_c2": ("c2", "--config c2 --steps 3 --warmup 1"), "trace_c5": ("c5", "--config c5 --steps 3 --warmup 1")}
EQS = ("induction", "continuity", "momentum", "energy", "gauss")
for e in EQS:
    CMDS[f"trace_c4_{e}"] = (f"c4_{e}", f"--config c4 --equation {e} --steps 3 --warmup 1")
for e in ("induction", "momentum"):
    CMDS[f"trace_c4_{e}_nt"] = (f"c4_{e}_ntfast", f"--config c4 --equation {e} --layout nt --steps 3 --warmup 1")
CMDS["trace_c5w"] = ("c5_whole", "--config c5 --batch 65536 --steps 3 --warmup 1")
for src, (tag, cmd) in CMDS.items():
    f = newest(f"gpurun_out/prof/{src}/**/*_kernel_stats.csv")
    if not f:
        continue
    shutil.copy(f, f"{out}/bench_{tag}_kernel_stats.csv")
    with open(f"{out}/bench_{tag}_summary.txt", "w") as o:
        o.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline {cmd}   (MI355X)\n")
        o.write(bench_line(f"gpurun_out/prof/bench_{src}.log") + "\n\n")
        o.write(f"{'kernel':70s} {'calls':>6s} {'avg_ms':>10s} {'total_ms':>10s} {'%':>7s}\n")
        for r in csv.DictReader(open(f)):
            if float(r["Percentage"]) >= 0.01:
                o.write(f"{short(r['Name']):70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e6:10.3f} "
                        f"{float(r['TotalDurationNs'])/1e6:10.1f} {float(r['Percentage']):7.2f}\n")


def counters(src):
    """{(counter, kernel): (dispatches, average over dispatches of the per-dispatch sum)}"""
    f = newest(f"gpurun_out/prof/{src}/**/*_counter_collection.csv")
    res = {}
    if not f:
        return res
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[(r["Counter_Name"], short(r["Kernel_Name"]))][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for key, d in agg.items():
        res[key] = (len(d), sum(d.values()) / len(d))
    return res


OURS = ("march_kernel", "flat_march_kernel", "moments_kernel", "moments_segmax_kernel", "joint_score_kernel", "joint_score_pruned_kernel", "segmin_kernel",
        "std_from_moments_kernel", "kth_")
sys.path.insert(0, os.getcwd())
import bench                                             # split_slabs / CONFIGS: the workloads the passes ran

import subprocess
try:        # the commit the counters were collected at (bench.py ignores the file once the kernel source differs from it)
    HEAD = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception:
    HEAD = None
SRC_NOTE = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction"


def hbm_report(tag, fetch_src, write_src, kernel, cmd, workload, alg_bytes, halo_bytes=None, note=""):
    pm = {**counters(fetch_src), **counters(write_src)}
    if ("FETCH_SIZE", kernel) not in pm or ("WRITE_SIZE", kernel) not in pm:
        return
    with open(f"{out}/pmc_hbm_{tag}.txt", "w") as o:
        o.write("rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes)\n"
                f"   -- python3 bench.py {cmd}   (MI355X)\n")
        o.write("counter unit KiB.  gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane\n"
                "coalesced streaming read -> x2; WRITE_SIZE is exact for 16 B/lane streaming stores.\n\n")
        for (name, k), (n, avg) in sorted(pm.items(), key=lambda x: (x[0][1], x[0][0])):
            if k.startswith(OURS):
                gb = avg * 1024 / 1e9 * (2 if name == "FETCH_SIZE" else 1)
                o.write(f"{name:11s} {k:40s} dispatches={n:3d} avg_KiB={avg:.6g} corrected_GB_per_dispatch={gb:.3f}\n")
        f = pm[("FETCH_SIZE", kernel)][1] * 1024 * 2
        w = pm[("WRITE_SIZE", kernel)][1] * 1024
        o.write(f"\n{kernel} per launch (average over the profiled launches){note}:\n")
        o.write(f"   algorithmic bytes (SURVEY 8d: 4*(F+1) B x cells computed)       {alg_bytes/1e9:.3f} GB\n")
        if halo_bytes:
            o.write(f"   + the two halo rows / planes each slab re-reads (overhead)       {halo_bytes/1e9:.3f} GB\n")
        o.write(f"   measured HBM traffic  read {f/1e9:.3f} GB + write {w/1e9:.3f} GB = {(f+w)/1e9:.3f} GB  = {(f+w)/alg_bytes:.4f} x algorithmic")
        if halo_bytes:
            o.write(f", {(f+w)/halo_bytes:.4f} x (algorithmic + slab halo)")
        o.write("\n")
    json.dump({"workload": workload, "kernel": kernel, "head": HEAD, "kernel_src_sha16": bench.kernel_src_sha16(),
               "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
               "traffic_bytes_per_launch": f + w, "algorithmic_bytes_per_launch": alg_bytes,
               "source": f"{out}/pmc_hbm_{tag}.txt ({SRC_NOTE})"}, open(f"{out}/pmc_hbm_{tag}.json", "w"), indent=1)


c3 = bench.CONFIGS["c3"]["shape"]
slab = 13
try:                                                     # the slab size the profiled run chose (bench line of the trace pass)
    slab = int(json.loads(bench_line("gpurun_out/prof/bench_fetch.log"))["config"]["slab"])
except Exception:
    pass
axis = "t"
try:
    axis = json.loads(bench_line("gpurun_out/prof/bench_fetch.log"))["config"].get("slab_axis", "t")
except Exception:
    pass
slabs = bench.split_slabs(c3[2] - 2 if axis == "x" else c3[1], slab)      # x-slabs: the reference's interior rows 1 .. Nx-2
cells_xy = c3[0] * (c3[1] * c3[3] if axis == "x" else c3[2] * c3[3])        # batch x cells per row (x-slabs) / plane (t-slabs)
hbm_report("c3", "fetch", "write", MK, "--steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity",
           {"batch": c3[0], "nt": c3[1], "nx": c3[2], "ny": c3[3], "slab": slab, "slab_axis": axis, "rows": "interior"},
           sum(16 * sl * cells_xy for sl in slabs) / len(slabs),
           sum((12 * (sl + 2) + 4 * sl) * cells_xy for sl in slabs) / len(slabs),
           note=(f" [4096,64,S+2,512] x3 -> [4096,64,S,512], S in {slabs}" if axis == "x" else
                 f" [4096,S+2,512,512] x3 -> [4096,S,512,512], S in {slabs}"))
for e in EQS:
    cfg = bench.mhd_config(e)
    shp = cfg["shape"]
    cells = shp[0] * shp[1] * shp[2] * shp[3]
    hbm_report(f"c4_{e}", f"fetch_c4_{e}", f"write_c4_{e}", cfg["kernel"], f"--config c4 --equation {e} --steps 1 --warmup 0 --no-cpu-baseline",
               {"batch": shp[0], "nt": shp[1], "nx": shp[2], "ny": shp[3]}, cfg["bpc"] * cells, note=f" {list(shp)} ({e})")
# C3 marginal (same x-slabs, |res| into the row-padded score buffer)
hbm_report("c3_marginal", "fetch_m", "write_m", MK, "--steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity --mode marginal",
           {"batch": c3[0], "nt": c3[1], "nx": c3[2], "ny": c3[3], "slab": slab, "slab_axis": axis, "rows": "interior"},
           sum(16 * sl * cells_xy for sl in slabs) / len(slabs),
           sum((12 * (sl + 2) + 4 * sl) * cells_xy for sl in slabs) / len(slabs),
           note=f" marginal mode: [4096,64,S+2,512] x3 -> |res| [4096,64,S,512] in rows 64 floats further apart, S in {slabs}")
# C4 in the surrogate's Nt-fastest layout
for e in ("induction", "momentum"):
    cfg = bench.mhd_config(e)
    shp = cfg["shape"]
    cells = shp[0] * shp[1] * shp[2] * shp[3]
    kern = cfg["kernel"].replace("march_kernel<", "flat_march_kernel<").replace("<0>,8,64>", "<3>>")
    hbm_report(f"c4_{e}_ntfast", f"fetch_c4_{e}_nt", f"write_c4_{e}_nt", kern,
               f"--config c4 --equation {e} --layout nt --steps 1 --warmup 0 --no-cpu-baseline",
               {"batch": shp[0], "nt": shp[1], "nx": shp[2], "ny": shp[3]}, cfg["bpc"] * cells,
               note=f" {list(shp)} ({e}), fields [BS,6,Nx,Ny,Nt].permute(0,1,4,2,3)")
# C5 at its single-GPU size
cfg5 = bench.CONFIGS["c5"]
hbm_report("c5_whole", "fetch_c5w", "write_c5w", cfg5["kernel"], "--config c5 --batch 65536 --steps 1 --warmup 0 --no-cpu-baseline",
           {"batch": 65536, "nt": cfg5["shape"][1], "nx": cfg5["shape"][2], "ny": 0}, cfg5["bpc"] * 65536 * cfg5["shape"][1] * cfg5["shape"][2],
           note=" [65536, 200, 512]")
for c in ("c1", "c2", "c5"):
    cfg = bench.CONFIGS[c]
    shp = cfg["shape"]
    cells = 1
    for d in shp:
        cells *= d
    kern = cfg["kernel"]
    hbm_report(c, f"fetch_{c}", f"write_{c}", kern, f"--config {c} --steps 1 --warmup 0 --no-cpu-baseline",
               {"batch": shp[0], "nt": shp[1], "nx": shp[2], "ny": shp[3] if len(shp) == 4 else 0}, cfg["bpc"] * cells,
               note=f" {list(shp)}")

for tag, srcs, mode in (("pmc_sq_c3.txt", ("sq1", "sq2"), "joint"), ("pmc_sq_c3_marginal.txt", ("sq1m", "sq2m", "fetchm"), "marginal")):
    sq = {}
    for src in srcs:
        sq.update(counters(src))
    if not sq:
        continue
    with open(f"{out}/{tag}", "w") as o:
        o.write(f"rocprofv3 --pmc <counters, one group per pass> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch 1024"
                f"{' --mode marginal' if mode == 'marginal' else ''}\n"
                f"(C3 {mode}, quarter batch: 1024 samples per launch, the slab plan of the run; per-dispatch sums over all SEs/XCDs, averaged over dispatches;\n"
                f" FETCH_SIZE in KiB, x2 for bytes on gfx950)\n\n")
        kernels = sorted({k for (_, k) in sq if k.startswith(OURS)})
        for k in kernels:
            o.write(k + "\n")
            vals = {c: v for (c, kk), (n, v) in sq.items() if kk == k}
            for c, v in sorted(vals.items()):
                o.write(f"    {c:24s} {v:16.6g}\n")
            wv = vals.get("SQ_WAVES")
            if wv:
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                    if c in vals:
                        o.write(f"    {c + ' / wave':24s} {vals[c] / wv:16.1f}\n")
            if vals.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in vals:
                o.write(f"    {'ACTIVE_INST_VALU / BUSY_CYCLES':32s} {vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_BUSY_CYCLES']:8.3f}\n")
            o.write("\n")
for e in ("momentum", "energy"):
    sq = {}
    for src in (f"sq1_c4_{e}", f"sq2_c4_{e}"):
        sq.update(counters(src))
    if not sq:
        continue
    with open(f"{out}/pmc_sq_c4_{e}.txt", "w") as o:
        o.write(f"rocprofv3 --pmc <counters, one group per pass> --kernel-trace -- python3 bench.py --config c4 --equation {e} --steps 1 --warmup 0 --no-cpu-baseline\n"
                f"(per-dispatch sums over all SEs/XCDs, averaged over dispatches)\n\n")
        for k in sorted({k for (_, k) in sq if k.startswith(OURS)}):
            o.write(k + "\n")
            vals = {c: v for (c, kk), (n, v) in sq.items() if kk == k}
            for c, v in sorted(vals.items()):
                o.write(f"    {c:24s} {v:16.6g}\n")
            wv = vals.get("SQ_WAVES")
            if wv:
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                    if c in vals:
                        o.write(f"    {c + ' / wave':24s} {vals[c] / wv:16.1f}\n")
            if vals.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in vals:
                o.write(f"    {'ACTIVE_INST_VALU / BUSY_CYCLES':32s} {vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_BUSY_CYCLES']:8.3f}\n")
            o.write("\n")
print("wrote", sorted(os.listdir(out)))
