#!/usr/bin/env python3
"""Per-kernel timings on the MI355X (HIP events on torch's current stream).
    python tools/microbench.py [eval] [calib] [select]
Prints algorithmic GB/s per kernel; used to steer optimisation, not a contract output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_pre_amd import _lib, inductive_cp as icp, pipeline
from cp_pre_amd import residuals as R
from cp_pre_amd.convops_2d import ConvOperator

dev = torch.device("cuda:0")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, nbytes):
    print(f"{name:58s} {ms:9.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)


def bench_eval():
    for (B, T, X, Y) in [(512, 10, 512, 512), (512, 32, 256, 256), (256, 64, 256, 256)]:
        cells = B * T * X * Y
        v = torch.empty(B, 6, T, X, Y, device=dev).uniform_(0.5, 1.5)
        out = torch.empty(B, T, X, Y, device=dev)
        ns = R.NavierStokes(0.01, 1 / X, 1 / Y)
        report(f"ns_momentum [{B},{T},{X},{Y}] 16B/cell", timeit(lambda: ns.residual_momentum(v[:, :3], True, out=out)), 16 * cells)
        report(f"ns_continuity (linear2) 12B/cell", timeit(lambda: ns.residual_continuity(v[:, :2], True)), 12 * cells)
        w = R.PRE_Wave(0.01, 0.02)
        report(f"wave additive kernel (linear1) 8B/cell", timeit(lambda: w.residual(v[:, 0], True)), 8 * cells)
        mhd = R.MHD()
        report(f"mhd_continuity 16B/cell", timeit(lambda: mhd.residual_continuity(v, True)), 16 * cells)
        report(f"mhd_induction 20B/cell", timeit(lambda: mhd.residual_induction(v, True)), 20 * cells)
        report(f"mhd_momentum 28B/cell", timeit(lambda: mhd.residual_momentum(v, True)), 28 * cells)
        report(f"mhd_energy 28B/cell", timeit(lambda: mhd.residual_energy(v, True)), 28 * cells)
        del v, out
    # reduced MHD (JOREK): the script's [BS,F,Nx,Ny,Nt] layout (Nt fastest through unstack_fields) and an Ny-fastest copy
    for (B, N, Nt) in [(512, 256, 20), (128, 512, 40)]:
        v3 = torch.empty(B, 3, N, N, Nt, device=dev).uniform_(0.5, 1.5)
        jo = R.JOREK(torch.linspace(1.0, 2.0, N))
        cells = B * N * N * Nt
        report(f"jorek continuity [BS={B},Nx=Ny={N},Nt={Nt}] Nt-fastest 12B/cell", timeit(lambda: jo.residual_continuity(v3, True)), 12 * cells)
        report(f"jorek temperature, Nt-fastest 16B/cell", timeit(lambda: jo.residual_temperature(v3, True)), 16 * cells)
        vy = v3.permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
        report(f"jorek temperature, Ny-fastest 16B/cell", timeit(lambda: jo.residual_temperature(vy, True)), 16 * cells)
        del v3, vy
    B, T, X = 8192, 200, 512
    u = torch.empty(B, T, X, device=dev).uniform_(0.5, 1.5)
    bur = R.Burgers(2 / 512, 1.25 / 200, 0.002)
    report(f"burgers [{B},{T},{X}] 8B/cell", timeit(lambda: bur.residual(u, True)), 8 * B * T * X)
    adv = R.Advection(1.0, 0.005, 0.01)
    report(f"advection additive kernel [{B},{T},{X}] 8B/cell", timeit(lambda: adv.residual(u, True)), 8 * B * T * X)
    L = ConvOperator(("x", "y"), 2)
    for (B, T, X, Y) in [(256, 10, 510, 510), (256, 10, 201, 201)]:
        xo = torch.randn(B, T, X, Y, device=dev)
        report(f"laplacian, odd width [{B},{T},{X},{Y}] 8B/cell (unaligned rows + tail cols)", timeit(lambda: L(xo)), 8 * xo.numel())
        crop = torch.randn(B, T, X + 2, Y + 2, device=dev)[..., 1:-1, 1:-1]
        report(f"laplacian on a cropped view [{B},{T},{X},{Y}] 8B/cell", timeit(lambda: L(crop)), 8 * crop.numel())
        del xo, crop


def bench_ceilings():
    """What plain streaming kernels reach on this box (torch elementwise / reduction kernels, 10.7 GB per tensor): the yardsticks
    the fused kernels' traffic is held against.  1R+1W: copy; 3R+1W (the NS momentum mix): addcmul; read-only: sum."""
    n = 1 << 31                                            # 2^31 floats = 8.6 GB per tensor
    a, b, c, d = (torch.empty(n, device=dev).uniform_(0.5, 1.5) for _ in range(4))
    report("ceiling: copy d <- a (1R + 1W) 8B/elem", timeit(lambda: d.copy_(a)), 8 * n)
    report("ceiling: add d <- a + b (2R + 1W) 12B/elem", timeit(lambda: torch.add(a, b, out=d)), 12 * n)
    report("ceiling: addcmul d <- a + b*c (3R + 1W) 16B/elem", timeit(lambda: torch.addcmul(a, b, c, out=d)), 16 * n)
    report("ceiling: sum(a) (1R) 4B/elem", timeit(lambda: a.sum()), 4 * n)
    del a, b, c, d


def bench_generic():
    """Tap sets off the 7-point star: the L1/L2-served tap-list kernel."""
    x = torch.randn(256, 10, 512, 512, device=dev)
    D4, D6 = ConvOperator(("x", "y"), 2, taylor_order=4), ConvOperator(("x", "y"), 2, taylor_order=6)
    report("taylor-4 laplacian (9 taps, 5^3) [256,10,512,512] 8B/cell", timeit(lambda: D4(x)), 8 * x.numel())
    report("taylor-6 laplacian (13 taps, 7^3) 8B/cell", timeit(lambda: D6(x)), 8 * x.numel())
    W = ConvOperator()
    D_tt = ConvOperator("t", 2)
    k5 = torch.zeros(5, 5, 5)
    k5[1:4, 1:4, 1:4] = D_tt.kernel
    W.kernel = k5 - 0.25 * D4.kernel
    report("wave, taylor-4 laplacian, additive 5^3 kernel (12 taps) 8B/cell", timeit(lambda: W(x)), 8 * x.numel())
    dense = ConvOperator()
    dense.kernel = torch.randn(3, 3, 3)
    report("dense 3^3 kernel (27 taps) 8B/cell", timeit(lambda: dense(x)), 8 * x.numel())
    xt = torch.randn(256, 512, 512, 10, device=dev).permute(0, 3, 1, 2)
    report("taylor-4 laplacian on an Nt-fastest view [256,10,512,512] 8B/cell", timeit(lambda: D4(xt)), 8 * xt.numel())
    for nt in (10, 20, 40):
        xs = torch.randn(256 * 40 // nt, 256, 256, nt, device=dev).permute(0, 3, 1, 2)
        report(f"taylor-4 laplacian, Nt-fastest [{xs.shape[0]},{nt},256,256] 8B/cell", timeit(lambda: D4(xs)), 8 * xs.numel())
        report(f"taylor-6 laplacian, Nt-fastest [{xs.shape[0]},{nt},256,256] 8B/cell", timeit(lambda: D6(xs)), 8 * xs.numel())
        report(f"wave + taylor-4 (12 taps), Nt-fastest 8B/cell", timeit(lambda: W(xs)), 8 * xs.numel())
        del xs
    xo = torch.randn(256, 10, 201, 201, device=dev)
    report("taylor-4 laplacian, odd width [256,10,201,201] 8B/cell", timeit(lambda: D4(xo)), 8 * xo.numel())


def bench_calib():
    for (n, T, X, Y) in [(1024, 10, 512, 512), (8192, 1, 200, 512)]:
        res = torch.randn(n, T, X, Y, device=dev)
        M = T * X * Y
        mom = pipeline.HipOps.zeros_moments(M, dev)
        report(f"moments_axis0_f64 [{n},{M}] 4B", timeit(lambda: pipeline.HipOps.add_moments(res, mom)), 4 * n * M)
        mod = pipeline.HipOps.std_from_moments(mom, n * 7, (T, X, Y), 0.0)
        sc = pipeline.HipOps.zeros_scores(n, dev)
        report(f"joint_score crop1 [{n},{T},{X},{Y}] 4B", timeit(lambda: pipeline.HipOps.max_scores(res, mod, (1, 1, 1) if T > 2 else (0, 1, 1), sc)), 4 * n * M)
        report(f"std_axis0 numpy-order [{n},{M}] 8B", timeit(lambda: icp.modulation_func(res, None)), 8 * n * M)
        # (three DISTINCT buffers: rounds 1-4 passed the same tensor three times - the second read hit the first one's lines in
        # L2 and the "12 B per element" line read 92 % of peak on 8 B of real traffic)
        other, dst = torch.empty_like(res), torch.empty_like(res)
        other.copy_(res).mul_(0.5)
        report(f"absdiff [{n * M}] 12B (a, b, out distinct)",
               timeit(lambda: _lib.load().pre_absdiff_f32(_lib.ptr(res), _lib.ptr(other), _lib.ptr(dst), n * M, _lib.stream())), 12 * n * M)
        del res, other, dst


def bench_select():
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    for (n, M) in [(256, 10 * 512 * 512), (512, 32 * 256 * 256), (1000, 4 * 512 * 512), (1024, 10 * 512 * 512), (2048, 2 * 512 * 512), (4096, 2 * 512 * 512), (8192, 254 * 254)]:
        s = torch.randn(n, M, device=dev).abs_()
        ks = [icp.kth_index(n, n, a) for a in alphas]
        ms = timeit(lambda: icp.kth_axis0(s, ks), reps=3, warm=1)
        report(f"kth_axis0 10 ranks [{n},{M}] one read of the scores = 4B", ms, 4 * n * M)
        del s
    s = torch.randn(65536 * 8, device=dev)
    report("kth scalar 10 ranks N=524288", timeit(lambda: icp.kth_axis0(s, [icp.kth_index(s.numel(), s.numel(), a) for a in alphas])), 4 * s.numel())


if __name__ == "__main__":
    what = sys.argv[1:] or ["eval", "calib", "select"]
    if "eval" in what: bench_eval()
    if "calib" in what: bench_calib()
    if "select" in what: bench_select()


def bench_copy():
    """Ceilings on this box: float4 device copy (torch) and the cost of re-laying out a permuted view."""
    n = 1 << 30
    a = torch.empty(n, device=dev).uniform_()
    b = torch.empty_like(a)
    report("torch copy_ 4 GiB (8 B/elem)", timeit(lambda: b.copy_(a)), 8 * n)
    report("torch a*2 -> b (8 B/elem)", timeit(lambda: torch.mul(a, 2.0, out=b)), 8 * n)
    del a, b
    sur = torch.empty(256, 3, 256, 256, 32, device=dev).uniform_()          # [BS,F,Nx,Ny,Nt]
    view = sur.permute(0, 1, 4, 2, 3)
    report("torch .contiguous() of permuted [256,3,32,256,256] view (8 B/elem)", timeit(lambda: view.contiguous()), 8 * sur.numel())
    one = view[:, 0]
    report("torch .contiguous() of one permuted field (8 B/elem)", timeit(lambda: one.contiguous()), 8 * one.numel())


def bench_permuted():
    """Residuals on the surrogate's native layout [BS,F,Nx,Ny,Nt] (zero-copy axis relabelling)."""
    for (B, T, X, Y) in [(256, 64, 256, 256), (64, 64, 512, 512), (800, 20, 256, 256), (1600, 10, 256, 256), (128, 128, 256, 256)]:
        cells = B * T * X * Y
        sur = torch.empty(B, 6, X, Y, T, device=dev).uniform_(0.5, 1.5)
        v = sur.permute(0, 1, 4, 2, 3)
        ns = R.NavierStokes(0.01, 1 / X, 1 / Y)
        report(f"ns_momentum Nt-fastest [{B},{T},{X},{Y}] 16B/cell", timeit(lambda: ns.residual_momentum(v[:, :3], True)), 16 * cells)
        w = R.PRE_Wave(0.01, 0.02)
        report("wave additive kernel Nt-fastest 8B/cell", timeit(lambda: w.residual(v[:, 0], True)), 8 * cells)
        mhd = R.MHD()
        report("mhd_induction Nt-fastest 20B/cell", timeit(lambda: mhd.residual_induction(v, True)), 20 * cells)
        del sur, v


def bench_spatial():
    """2-D spatial operators with fused boundary conditions (Utils/VectorConvOps_Spatial.py)."""
    from cp_pre_amd import vector_convops_spatial as V
    for (B, X, Y) in [(4096, 512, 512), (16384, 256, 256)]:
        a = torch.empty(B, 1, X, Y, device=dev).uniform_()
        b = torch.empty(B, 1, X, Y, device=dev).uniform_()
        cells = B * X * Y
        L = V.Laplace(scale=1.0, boundary_cond='periodic', device=dev)
        D = V.Divergence(scale=1.0, boundary_cond='periodic', device=dev)
        with torch.no_grad():
            report(f"spatial Laplace periodic fused [{B},1,{X},{Y}] 8B/cell", timeit(lambda: L(a)), 8 * cells)
            report(f"spatial Divergence periodic fused 12B/cell", timeit(lambda: D(a, b)), 12 * cells)
            report(f"spatial Laplace: pad_signal + valid conv (reference recipe) 8B/cell", timeit(lambda: L.laplace(L.bc.pad_signal(a))), 8 * cells)
        del a, b


def bench_spectral():
    """Spectral family: libcp_pre_fft.so (fused embed / multiply / crop around hipFFT) vs the torch.fft composition."""
    from cp_pre_amd import _spectral as S
    D = ConvOperator()
    D.kernel = ConvOperator("t", 2).kernel - 0.25 * ConvOperator(("x", "y"), 2).kernel
    for shape in [(64, 32, 256, 256), (16, 64, 512, 512)]:
        x = torch.randn(*shape, device=dev)
        with torch.no_grad():
            report(f"spectral_convolution native {list(shape)} 8B/cell", timeit(lambda: S.fft_xcorr(x, D.kernel)), 8 * x.numel())
            report("spectral_convolution torch.fft composition", timeit(lambda: S._torch_fft_xcorr(x, D.kernel)), 8 * x.numel())
            report("differentiate native", timeit(lambda: S.differentiate(x, D.kernel, True, True)), 8 * x.numel())
            report("differentiate torch.fft composition", timeit(lambda: S._torch_differentiate(x, D.kernel, True, True)), 8 * x.numel())
        del x


if __name__ == "__main__" and "spectral" in sys.argv[1:]:
    bench_spectral()
if __name__ == "__main__" and "generic" in sys.argv[1:]:
    bench_generic()
if __name__ == "__main__" and "spatial" in sys.argv[1:]:
    bench_spatial()
if __name__ == "__main__" and "copy" in sys.argv[1:]:
    bench_copy()
if __name__ == "__main__" and "ceilings" in sys.argv[1:]:
    bench_ceilings()
if __name__ == "__main__" and "permuted" in sys.argv[1:]:
    bench_permuted()
