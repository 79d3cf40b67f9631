"""Does the distance between the rows of a tile (= pages touched per byte) bound the per-cell select?  The same
kernels on the same amount of data, as P planes of [n, M] with the rows `pitch` floats apart (pre_kth_axis0_planes_f32):
rows 2 MiB apart put every row of a tile in its own page region, 2.25 KiB apart (a 512-cell row + 64 floats of pad: what
pipeline.sample_inner gives a C3 slab) ~900 rows in one.  With --alt <lib.so> the launches alternate between the two
builds (e.g. one compiled with -DKA_NO_C32: the streaming form for 1024 < n <= 2048)."""
import argparse, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, inductive_cp as icp

ap = argparse.ArgumentParser()
ap.add_argument("--alt", default=None)
ap.add_argument("--ns", default="512,1024,1500,2048,4096")
args = ap.parse_args()
dev = torch.device("cuda:0")
main_path = _lib.SO_PATH
libs = {"new": _lib.load()}
if args.alt:
    _lib._lib, _lib.SO_PATH = None, args.alt
    libs["alt"] = _lib.load()
    _lib.SO_PATH = main_path
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for n in [int(x) for x in args.ns.split(",")]:
    for (M, pitch) in ((524288, 524288), (524288, 524288 + 64), (65536, 65536), (8192, 8192), (512, 512), (512, 576), (256, 320)):
        P = max(1, (1 << 30) // (n * pitch))              # ~4.3 GB of scores
        torch.manual_seed(n)
        buf = torch.randn(P * n * pitch, device=dev).abs_()
        s = buf.as_strided((P, n, M), (n * pitch, pitch, 1))
        ks = [icp.kth_index(n, n, a) for a in alphas]
        out = torch.empty(P, len(ks), M, device=dev)
        kk = _lib.iarr32(ks)
        res = {}
        for tag, lib in libs.items():
            def run():
                _lib.check(lib.pre_kth_axis0_planes_f32(_lib.ptr(buf), n * pitch, pitch, P, n, M, kk, len(ks), _lib.ptr(out), M,
                                                        len(ks) * M, _lib.stream()), "planes")
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            ref = torch.sort(s[P - 1, :, :256], dim=0).values[ks]
            res[tag] = (ms, torch.equal(out[P - 1, :, :256], ref))
        line = "  ".join(f"{t}: {ms:.3f} ms {4*P*n*M/ms/1e6:5.0f} GB/s exact={ok}" for t, (ms, ok) in res.items())
        print(f"n={n} M={M} pitch={pitch} planes={P} rows {4*pitch/1024:.2f} KiB apart: {line}", flush=True)
        del buf, s, out
