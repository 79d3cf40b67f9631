// star_march.hip - streaming evaluation of 7-point-star stencils and fused PDE residuals
// over [B,T,X,Y] fp32 fields on gfx950 (MI355X).
//
// Bound: HBM.  Algorithmic traffic is 4*(F+1) bytes per cell (F input fields read once,
// one residual written once).  Structure of one workgroup (NR x TYQ threads):
//   * owns an (x,y) tile of NR rows x 4*TYQ columns of ONE sample and marches over t;
//   * every thread keeps a 4-plane sliding window (t-1, t, t+1 and the in-flight t+2) of
//     its own float4 per field in REGISTERS, so each input cell is fetched once per
//     workgroup and the t-taps cost nothing;
//   * the current plane is staged through LDS (double-buffered, one barrier per plane) for
//     the x-neighbours, including one halo row above and below the tile;
//   * y-neighbours come from the adjacent lane by a wavefront shuffle; only the two edge
//     lanes of each 64-wide wave fetch a halo scalar;
//   * loads for plane t+2 are issued before plane t is computed (software prefetch);
//   * blockIdx is remapped so each XCD's L2 sees a contiguous run of tiles (shared halos).
// The residual algebra is a compile-time functor; operator weights are run-time scalars
// taken from the caller's dense 3x3x3 kernels, so the reference's kernel-construction
// quirks are inherited (see include/cp_pre_hip.h).
#include "common.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int MAXF = 6;

struct Geom {
    const float *f[MAXF];
    long long sB[MAXF], sT[MAXF], sX[MAXF];
    float *out;
    long long oB, oT, oX;        // output strides (elements); the marched-to axis is contiguous
    int B, T, X, Y;
    int Yc;                      // columns computed by the streaming kernel: Y rounded down to a multiple of 4
                                 // (the <= 3 remaining columns of an odd-width grid go to the generic kernel)
    int tSeg, nTSeg, nXT, nYT;
    int flags;
    int flat;                    // 1: short contiguous axis merged with the next one (flat_march_kernel)
    int tfree;                   // 1: no operator of the functor has a tap along the marched axis (set by the entry points)
};

// c, t-, t+, x-, x+, y-, y+
struct Star { float c, tm, tp, xm, xp, ym, yp; };

struct Nbr { float4 c, tm, tp, xm, xp, ym, yp; };

__device__ __forceinline__ float4 f4(float s) { return make_float4(s, s, s, s); }
__device__ __forceinline__ float4 fabs4(const float4 &a) { return make_float4(fabsf(a.x), fabsf(a.y), fabsf(a.z), fabsf(a.w)); }

enum Kind { K_T3, K_X3, K_Y3, K_XY5, K_TX5, K_STAR7 };

template <int KIND>
__device__ __forceinline__ float4 apply(const Star &w, const Nbr &n)
{
    if (KIND == K_T3) return w.tm * n.tm + w.c * n.c + w.tp * n.tp;
    if (KIND == K_X3) return w.xm * n.xm + w.c * n.c + w.xp * n.xp;
    if (KIND == K_Y3) return w.ym * n.ym + w.c * n.c + w.yp * n.yp;
    if (KIND == K_XY5) return w.xm * n.xm + w.ym * n.ym + w.c * n.c + w.yp * n.yp + w.xp * n.xp;
    if (KIND == K_TX5) return w.tm * n.tm + w.xm * n.xm + w.c * n.c + w.xp * n.xp + w.tp * n.tp;
    return w.tm * n.tm + w.xm * n.xm + w.ym * n.ym + w.c * n.c + w.yp * n.yp + w.xp * n.xp + w.tp * n.tp;
}

// ------------------------------------------------------------------ residual functors
// MODE 0: the tap structure the reference constructs (D_t,D_y along Nt; D_x along Nx;
//         Laplacian on the (Nx,Ny) cross).  MODE 1: D_y along Ny (the physically intended
//         stencil).  MODE 2: every operator a general 7-point star.
// MODE 3 / 4: MODE 0 / 1 after the axis relabelling for Nt-fastest views (kernel axes =
//         logical (Nx, Ny, Nt)): logical t-taps sit on the kernel's y axis, x-taps on its t axis,
//         y-taps on its x axis.
template <int MODE> struct OpKinds {
    static constexpr int DT = MODE == 2 ? K_STAR7 : (MODE >= 3 ? K_Y3 : K_T3);
    static constexpr int DX = MODE == 2 ? K_STAR7 : (MODE >= 3 ? K_T3 : K_X3);
    static constexpr int DY = MODE == 2 ? K_STAR7 : (MODE == 0 ? K_T3 : MODE == 1 ? K_Y3 : MODE == 3 ? K_Y3 : K_X3);
    static constexpr int LAP = MODE == 2 ? K_STAR7 : (MODE >= 3 ? K_TX5 : K_XY5);
};

// Which fields does a functor need x-NEIGHBOURS of?  Only those are staged through LDS (and only their halo rows are
// fetched): a field that a functor reads at the centre alone, or only along t / y in the tap structure of its MODE, skips
// the LDS store, the two LDS reads, the halo-row loads and its share of the tile (MHD momentum never differentiates rho;
// in the reference's construction D_y has its taps on Nt, so what is only ever under D_y / D_t needs no x-neighbour:
// MHD energy stages 4 of its 6 fields, continuity 2 of 3).  O_*: the operators applied to a field; XMASK bit i = field i is
// staged.  A functor without XMASK stages everything.  (A wrong mask cannot pass silently: the x-neighbours of an unstaged
// field are NaN.)
enum { O_DT = 1, O_DX = 2, O_DY = 4, O_LAP = 8 };
constexpr bool kind_has_x(int k) { return k == K_X3 || k == K_XY5 || k == K_TX5 || k == K_STAR7; }
template <int MODE> constexpr bool ops_have_x(int ops)
{
    using K = OpKinds<MODE>;
    return ((ops & O_DT) && kind_has_x(K::DT)) || ((ops & O_DX) && kind_has_x(K::DX)) || ((ops & O_DY) && kind_has_x(K::DY)) ||
           ((ops & O_LAP) && kind_has_x(K::LAP));
}
template <int MODE> constexpr unsigned xmask_of(int o0, int o1 = 0, int o2 = 0, int o3 = 0, int o4 = 0, int o5 = 0)
{
    return (ops_have_x<MODE>(o0) ? 1u : 0u) | (ops_have_x<MODE>(o1) ? 2u : 0u) | (ops_have_x<MODE>(o2) ? 4u : 0u) |
           (ops_have_x<MODE>(o3) ? 8u : 0u) | (ops_have_x<MODE>(o4) ? 16u : 0u) | (ops_have_x<MODE>(o5) ? 32u : 0u);
}

struct Linear1 {       // out = S(f0): any single ConvOperator / additive kernel (README.md:47-54)
    static constexpr int F = 1;
    struct Params { Star s; };
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[1], const Params &p) { return apply<K_STAR7>(p.s, n[0]); }
};

struct Linear2 {       // out = Sa(f0) + ratio*Sb(f1)
    static constexpr int F = 2;
    struct Params { Star a, b; float ratio; };
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[2], const Params &p)
    {
        return apply<K_STAR7>(p.a, n[0]) + p.ratio * apply<K_STAR7>(p.b, n[1]);
    }
};

// dxdy = dx*dy, dtdy = dt*dy, dtdx = dt*dx, nudt = nu*dt, each product rounded once in fp32 on the host:
// the reference multiplies the two scalars one after the other onto the tensor (two roundings);
// the folded form differs by <= 1 ulp per term (1e-7 relative, tolerance 1e-5) and removes a
// third of the kernel's packed multiplies.
struct NSParams { Star Dt, Dx, Dy, L; float dxdy, dtdy, dtdx, nudt; };
struct BurgersParams { Star Dt, Dx, Dxx; float dx, dt, nu, c3; };
struct MHDParams { Star Dt, Dx, Dy; float gamma, gm2; };

template <int MODE>
struct NSMomentum {    // Marginal/NS_Residuals_CP.py:231-240
    static constexpr int F = 3;
    // Nt-fastest relabelling (MODE 3/4) needs 130-138 VGPRs unconstrained = 3 waves/SIMD; capped at 128 it
    // spills 0-6 dwords and runs 4 waves/SIMD: +11 % (4.7-5.0 TB/s).  The same cap on the MHD induction kernel
    // (146 VGPRs, 17 dwords spilled) was -30 %: scratch traffic inside the plane loop.
    static constexpr int MIN_WAVES = MODE >= 3 ? 4 : 1;
    static constexpr unsigned XMASK = xmask_of<MODE>(O_DT | O_DX | O_DY | O_LAP, O_DT | O_DX | O_DY | O_LAP, O_DX | O_DY);
    using Params = NSParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[3], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &u = n[0], &v = n[1], &pr = n[2];
        float4 rx = apply<K::DT>(p.Dt, u) * p.dxdy;
        rx = rx + u.c * apply<K::DX>(p.Dx, u) * p.dtdy;
        rx = rx + v.c * apply<K::DY>(p.Dy, u) * p.dtdx;
        rx = rx - apply<K::LAP>(p.L, u) * p.nudt;
        rx = rx + apply<K::DX>(p.Dx, pr) * p.dtdy;
        float4 ry = apply<K::DT>(p.Dt, v) * p.dxdy;
        ry = ry + u.c * apply<K::DX>(p.Dx, v) * p.dtdx;
        ry = ry + v.c * apply<K::DY>(p.Dy, v) * p.dtdy;
        ry = ry - apply<K::LAP>(p.L, v) * p.nudt;
        ry = ry + apply<K::DY>(p.Dy, pr) * p.dtdx;
        return rx + ry;
    }
};

// 1-D Burgers on the [1,B,T,X] view: the script's D_t runs along our x axis, D_x / D_xx along y.
template <int MODE>
struct Burgers {       // Joint/Burgers_Residuals_CP.py:182-187
    static constexpr int F = 1;
    using Params = BurgersParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[1], const Params &p)
    {
        // MODE 0: Nx fastest (D_t on the kernel's x axis, D_x on y); MODE 3: Nt fastest (swapped)
        constexpr int KT = MODE == 0 ? K_X3 : MODE == 3 ? K_Y3 : K_STAR7, KX = MODE == 0 ? K_Y3 : MODE == 3 ? K_X3 : K_STAR7;
        const Nbr &u = n[0];
        float4 r = p.dx * apply<KT>(p.Dt, u);
        r = r + (p.dt * u.c) * apply<KX>(p.Dx, u);
        r = r - (p.nu * apply<KX>(p.Dxx, u)) * p.c3;
        return r;
    }
};

template <int MODE>
struct MHDContinuity { // Marginal/MHD_Residuals_CP.py:225-231   fields rho,u,v
    static constexpr int F = 3;
    static constexpr unsigned XMASK = xmask_of<MODE>(O_DT | O_DX | O_DY, O_DX, O_DY);
    using Params = MHDParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[3], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &rho = n[0], &u = n[1], &v = n[2];
        float4 r = apply<K::DT>(p.Dt, rho) + u.c * apply<K::DX>(p.Dx, rho);
        r = r + rho.c * apply<K::DX>(p.Dx, u);
        r = r + v.c * apply<K::DY>(p.Dy, rho);
        r = r + rho.c * apply<K::DY>(p.Dy, v);
        return r;
    }
};

template <int MODE>
struct MHDMomentum {   // Marginal/MHD_Residuals_CP.py:234-243   fields rho,u,v,p,Bx,By
    static constexpr int F = 6;
    static constexpr unsigned XMASK = xmask_of<MODE>(0, O_DT | O_DX | O_DY, O_DT | O_DX | O_DY, O_DX | O_DY, O_DX | O_DY, O_DX | O_DY);
    using Params = MHDParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[6], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &rho = n[0], &u = n[1], &v = n[2], &pr = n[3], &bx = n[4], &by = n[5];
        const float4 irho = f4(1.0f) / rho.c, bxr = bx.c / rho.c, byr = by.c / rho.c;
        float4 rx = apply<K::DT>(p.Dt, u) + u.c * apply<K::DX>(p.Dx, u);
        rx = rx + irho * apply<K::DX>(p.Dx, pr);
        rx = rx - (2.0f * bxr) * apply<K::DX>(p.Dx, bx);
        rx = rx + v.c * apply<K::DY>(p.Dy, u);
        rx = rx - byr * apply<K::DY>(p.Dy, bx);
        rx = rx - bxr * apply<K::DY>(p.Dy, by);
        float4 ry = apply<K::DT>(p.Dt, v) + u.c * apply<K::DX>(p.Dx, v);
        ry = ry + irho * apply<K::DY>(p.Dy, pr);
        ry = ry - (2.0f * byr) * apply<K::DY>(p.Dy, by);
        ry = ry + v.c * apply<K::DY>(p.Dy, v);
        ry = ry - byr * apply<K::DX>(p.Dx, bx);
        ry = ry - bxr * apply<K::DX>(p.Dx, by);
        return rx + ry;
    }
};

template <int MODE>
struct MHDEnergy {     // Marginal/MHD_Residuals_CP.py:247-256; PRE_estimations.py:70-80
    static constexpr int F = 6;
    static constexpr unsigned XMASK = xmask_of<MODE>(O_DT, O_DX | O_DY, O_DX | O_DY, O_DX | O_DY, O_DX, O_DY);
    using Params = MHDParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[6], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &rho = n[0], &u = n[1], &v = n[2], &pr = n[3], &bx = n[4], &by = n[5];
        const float4 bx2 = bx.c * bx.c, by2 = by.c * by.c;
        const float4 pgas = pr.c - 0.5f * (bx2 + by2);
        float4 r = apply<K::DT>(p.Dt, rho) + u.c * apply<K::DX>(p.Dx, pr);
        r = r + v.c * apply<K::DY>(p.Dy, pr);
        r = r + (p.gm2 * (u.c * bx.c + v.c * by.c)) * (apply<K::DX>(p.Dx, bx) + apply<K::DY>(p.Dy, by));
        r = r + (p.gamma * pgas + by2) * apply<K::DX>(p.Dx, u);
        r = r + (p.gamma * pgas + bx2) * apply<K::DY>(p.Dy, v);
        r = r - (bx.c * by.c) * (apply<K::DY>(p.Dy, u) + apply<K::DX>(p.Dx, v));
        return r;
    }
};

template <int MODE>
struct MHDInduction {  // Marginal/MHD_Residuals_CP.py:259-268   fields u,v,Bx,By
    static constexpr int F = 4;
    static constexpr unsigned XMASK = xmask_of<MODE>(O_DX | O_DY, O_DX | O_DY, O_DT | O_DX | O_DY, O_DT | O_DX | O_DY);
    using Params = MHDParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[4], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &u = n[0], &v = n[1], &bx = n[2], &by = n[3];
        float4 rx = apply<K::DT>(p.Dt, bx) - by.c * apply<K::DY>(p.Dy, u);
        rx = rx + bx.c * apply<K::DY>(p.Dy, v);
        rx = rx - v.c * apply<K::DY>(p.Dy, bx);
        rx = rx + u.c * apply<K::DY>(p.Dy, by);
        float4 ry = apply<K::DT>(p.Dt, by) + by.c * apply<K::DX>(p.Dx, u);
        ry = ry - bx.c * apply<K::DX>(p.Dx, v);
        ry = ry - v.c * apply<K::DX>(p.Dx, bx);
        ry = ry + u.c * apply<K::DX>(p.Dx, by);
        return rx + ry;
    }
};

// Reduced MHD (JOREK), Marginal/JOREK_residuals_CP.py:207-243 (twin: Joint/JOREK_residuals_CP.py).  Fields rho, phi, T
// and the radius R - in the script a 1-D grid tensor broadcast along the LAST axis of the [BS,Nt,Nx,Ny] fields, here one
// more "field" whose view repeats that row (zero strides on the other axes): only its centre value is used.  D_R / D_RR
// share the tap structure of D_x, D_Z / D_ZZ that of D_y (the reference's 'y' operators have their taps along Nt,
// SURVEY 0.5 - inherited through the dense kernels like everywhere else).  Evaluation order = the script's.
//   continuity:  res = a0*D_t(rho) - (a1*R)*X(rho) - (a2*rho)*D_Z(phi) - a3*Y(rho)
//                X(f) = D_R(f)*D_Z(phi) - D_R(phi)*D_Z(f),  Y(f) = (D_RR(f) + (1/R)*D_R(f)) + D_ZZ(f)
//                norms=False: a = (1, 1, 2, D);  norms=True: the script's folded scalars (host, fp32, same order)
//   temperature: res = T*D_t(rho) + rho*D_t(T) - (rho*R)*X(T) + (T*R)*X(rho) + ((a0*rho)*T)*D_Z(phi) + a3*Y(T),
//                a0 = 2*gamma, a3 = K
struct JorekParams { Star Dt, DR, DZ, DRR, DZZ; float a0, a1, a2, a3; };

template <int MODE>
struct JorekContinuity {
    static constexpr int F = 3;        // rho, phi, R
    static constexpr unsigned XMASK = xmask_of<MODE>(O_DT | O_DX | O_DY, O_DX | O_DY, 0);      // (R: its centre value only)
    using Params = JorekParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[3], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &rho = n[0], &phi = n[1];
        const float4 R = n[2].c;
        const float4 dRrho = apply<K::DX>(p.DR, rho), dZphi = apply<K::DY>(p.DZ, phi);
        const float4 X = dRrho * dZphi - apply<K::DX>(p.DR, phi) * apply<K::DY>(p.DZ, rho);
        const float4 Y = (apply<K::DX>(p.DRR, rho) + (f4(1.0f) / R) * dRrho) + apply<K::DY>(p.DZZ, rho);
        float4 r = p.a0 * apply<K::DT>(p.Dt, rho);
        r = r - (p.a1 * R) * X;
        r = r - (p.a2 * rho.c) * dZphi;
        r = r - p.a3 * Y;
        return r;
    }
};

template <int MODE>
struct JorekTemperature {
    static constexpr int F = 4;        // rho, phi, T, R
    static constexpr unsigned XMASK = xmask_of<MODE>(O_DT | O_DX | O_DY, O_DX | O_DY, O_DT | O_DX | O_DY, 0);
    using Params = JorekParams;
    static __device__ __forceinline__ float4 eval(const Nbr (&n)[4], const Params &p)
    {
        using K = OpKinds<MODE>;
        const Nbr &rho = n[0], &phi = n[1], &T = n[2];
        const float4 R = n[3].c;
        const float4 dZphi = apply<K::DY>(p.DZ, phi), dRphi = apply<K::DX>(p.DR, phi);
        const float4 dRT = apply<K::DX>(p.DR, T);
        const float4 XT = dRT * dZphi - dRphi * apply<K::DY>(p.DZ, T);
        const float4 Xr = apply<K::DX>(p.DR, rho) * dZphi - dRphi * apply<K::DY>(p.DZ, rho);
        const float4 YT = (apply<K::DX>(p.DRR, T) + (f4(1.0f) / R) * dRT) + apply<K::DY>(p.DZZ, T);
        float4 r = T.c * apply<K::DT>(p.Dt, rho) + rho.c * apply<K::DT>(p.Dt, T);
        r = r - (rho.c * R) * XT;
        r = r + (T.c * R) * Xr;
        r = r + ((p.a0 * rho.c) * T.c) * dZphi;
        r = r + p.a3 * YT;
        return r;
    }
};

// ------------------------------------------------------------------ the marching kernel
// Global float4 accesses are declared 4-byte aligned: gfx950 runs with unaligned access enabled and
// the compiler still emits one global_load/store_dwordx4, so views whose base or row pitch is not a
// multiple of 16 bytes (odd grid widths, offset slices) stream through the same kernel.
struct __attribute__((aligned(4))) F4u { float x, y, z, w; };
__device__ __forceinline__ float4 ldg4(const float *p)
{
    const F4u v = *reinterpret_cast<const F4u *>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4(float *p, const float4 &r) { *reinterpret_cast<F4u *>(p) = F4u{r.x, r.y, r.z, r.w}; }

// y-neighbours from the adjacent lane: the value of lane - 1 / lane + 1 of the 64-wide wave (a wave's first / last lane
// gets something unspecified: the callers give those lanes their edge scalar).  MARCH_DPP: one `v_mov_b32_dpp wave_shr:1 /
// wave_shl:1` each (gfx9 DPP wave shifts: tools/exp/dpp_probe.hip) instead of `__shfl_up / __shfl_down`, which compile to
// ds_bpermute_b32 - a trip through the LDS crossbar and an lgkmcnt wait per neighbour.
#ifndef MARCH_DPP
#define MARCH_DPP 0
#endif
__device__ __forceinline__ float lane_below(float x)
{
#if MARCH_DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, false));
#else
    return __shfl_up(x, 1);
#endif
}
__device__ __forceinline__ float lane_above(float x)
{
#if MARCH_DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, false));
#else
    return __shfl_down(x, 1);
#endif
}

// Barrier that orders LDS traffic only: __syncthreads() would also drain vmcnt and with it
// the global prefetches that are meant to stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// the halo of one plane as this thread holds it: ONE float of the row above / below the tile per field (the two rows are
// 8*TYQ floats, fetched by the first 8*TYQ threads of the workgroup, a float each - as float4s held by the threads of the
// tile's first and last row they cost every thread 8 registers per field, and these registers are live across the
// functor: round 3), and ONE y-neighbour scalar (`ye`): the y- cell of a wave's first lane, the y+ cell of a wave's / tile's
// last lane - no lane is both.  (A row's last computed quad elsewhere takes its y+ cell from the next lane like every
// interior quad: the lanes beyond the computed width hold the grid's next column, if there is one, in C.x.)  The
// boundary-condition instantiations, where the last quad of a row maps its y+ cell wherever it lies, keep a second one.
// (Functors of five or more fields run one workgroup per CU whatever they save - their LDS tile is 120 KB - and keep the
// float4 form, a thread of the tile's first / last row fetching its own quad of the row beyond: fewer load instructions,
// measured 2.5-3.4 % faster on MHD momentum / energy.)
template <int F, bool BC, bool COOP> struct Halo {
    typename std::conditional<COOP, float, float4>::type row[F];
    float ye[F];
    float yr[BC ? F : 1];
};

// Boundary conditions on the (x, y) rim for the BC=true instantiations (Utils/boundary_conditions.py:
// BoundaryManager.pad_signal followed by a 'valid' conv == a 'same' conv whose out-of-domain neighbour
// is a mapped in-domain cell or a constant).  For each side: idx >= 0 = row / column to read instead
// of the cell just outside (periodic: the opposite edge; neumann/outflow: the edge itself; symmetric:
// one inside the edge), idx < 0 = the constant val (dirichlet).  Radius-1 stars never see corners.
struct BCInfo { int xlo, xhi, ylo, yhi; float vxlo, vxhi, vylo, vyhi; };
struct NoBC {};

// experiment knobs for the functors of five or more fields (tools/exp/build_variants.sh; the defaults are the product)
#ifndef MARCH6_NR
#define MARCH6_NR 8
#endif
#ifndef MARCH6_TYQ
#define MARCH6_TYQ 64
#endif
#ifndef MARCH6_MINW
#define MARCH6_MINW 1
#endif
#ifndef MARCH1_TSEG
#define MARCH1_TSEG 0          // 8: marches of 8 planes for a one-field functor WITH t-taps on batched 3-D fields (T >= 32) - measured
                               // -3 ... -4 % of the wave kernel's time for +10 % of fabric traffic (FETCH 1.007x -> 1.11x): not taken
#endif
#ifndef MARCHN_TSEG
#define MARCHN_TSEG 0
#endif
#ifndef TFREE_TSEG
#define TFREE_TSEG 8            // planes per march when no operator has a tap along the marched axis (2: +5 ... +22 % slower, 4: mixed,
                                // 8: -2 ... -10 % against pick_tseg's 32+: profiles/r06/march_ab_tfree_tseg.txt)
#endif
#ifndef MARCH1_ANYB
#define MARCH1_ANYB 0
#endif
#ifndef MARCH6_AHEAD
#define MARCH6_AHEAD 2          // 3: the functors of five or more fields request their own cells three planes ahead
#endif
#ifndef MARCH_COOP_MAXF
#define MARCH_COOP_MAXF 4
#endif

// Fn::MIN_WAVES (optional): waves per SIMD the register allocator must leave room for
template <class Fn, class = void> struct MinWaves { static constexpr int value = Fn::F >= 5 ? MARCH6_MINW : 1; };
template <class Fn> struct MinWaves<Fn, std::void_t<decltype(Fn::MIN_WAVES)>> { static constexpr int value = Fn::MIN_WAVES; };

// Fn::XMASK (optional): the fields staged through LDS for their x-neighbours (default: all)
template <class Fn, class = void> struct XMask { static constexpr unsigned value = (1u << Fn::F) - 1u; };
template <class Fn> struct XMask<Fn, std::void_t<decltype(Fn::XMASK)>> { static constexpr unsigned value = Fn::XMASK & ((1u << Fn::F) - 1u); };
template <class Fn> struct Staged {
    static constexpr unsigned M = XMask<Fn>::value;
    static constexpr int count = __builtin_popcount(M);
    static constexpr int FX = count > 0 ? count : 1;                      // (array extent; nothing is stored when count == 0)
    static __device__ __forceinline__ constexpr bool has(int i) { return (M >> i) & 1u; }
    static __device__ __forceinline__ constexpr int slot(int i) { return __builtin_popcount(M & ((1u << i) - 1u)); }
};

template <class Fn, int NR, int TYQ, bool BC = false>
__global__ void __launch_bounds__(NR *TYQ, MinWaves<Fn>::value)
march_kernel(const Geom g, const typename Fn::Params prm, const typename std::conditional<BC, BCInfo, NoBC>::type bc)
{
    constexpr int F = Fn::F;
    using SX = Staged<Fn>;
    constexpr int AHEAD = (F >= 5 && MARCH6_AHEAD == 3) ? 3 : 2;       // planes between a plane's request and its use as t + 1
    static_assert(NR >= 2, "tile needs at least two rows (top and bottom halo owners differ)");
    __shared__ float4 lds[2][SX::FX][NR + 2][TYQ];

    const int q = threadIdx.x, ty = threadIdx.y;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int yt = L % g.nYT; L /= g.nYT;
    const int xt = L % g.nXT; L /= g.nXT;
    const int ts = L % g.nTSeg;
    const int b = L / g.nTSeg;

    const int x = xt * NR + ty, y = (yt * TYQ + q) * 4;
    const bool inb = (x < g.X) && (y < g.Yc);
    // BC: the row just below the domain (x == X, only in a partial last tile) is a ghost row that
    // feeds the x+ neighbour of row X-1; it loads its mapped row and never stores
    int xl = x;                 // row this thread loads as its "own"
    float ghost = 0.f;
    bool ldown = inb;
    if constexpr (BC) {
        if (x == g.X && y < g.Yc) { xl = bc.xhi; ghost = bc.vxhi; ldown = bc.xhi >= 0; }
    } else {
        // PRE_FLAG_HALO_X, partial last tile: row X is real data and the x+ neighbour of row X-1 (loaded, never stored)
        if ((g.flags & PRE_FLAG_HALO_X) && x == g.X && y < g.Yc) ldown = true;
    }
    int t0 = ts * g.tSeg;
    int t1 = min(t0 + g.tSeg, g.T);
    if (g.flags & PRE_FLAG_INTERIOR_T) {       // the caller crops the t rim: neither compute nor store it
        t0 = max(t0, 1);
        t1 = min(t1, g.T - 1);
    }

    // halo-row duty: the workgroup's first 4*TYQ threads fetch the row above the tile, the next 4*TYQ the row below, one
    // float each (a wave = 64 consecutive floats of one row)
    static_assert(NR >= 8 && (4 * TYQ) % 64 == 0, "the two halo rows are fetched by the first 8*TYQ threads, a wave per 64 floats");
    constexpr bool COOP = F <= MARCH_COOP_MAXF;
    const int hl = ty * TYQ + q;                     // linear thread index
    const bool hduty = COOP ? hl < 8 * TYQ : (ty == 0 || ty == NR - 1), hbot = COOP ? hl >= 4 * TYQ : ty == NR - 1;
    const int hcol = COOP ? hl & (4 * TYQ - 1) : 4 * q;          // column within the tile (!COOP: of the quad's first cell)
    const int hy = yt * (4 * TYQ) + hcol;            // column of the grid
    int hx = hbot ? xt * NR + NR : xt * NR - 1;
    // PRE_FLAG_HALO_X: rows -1 and X of the views exist (an x-slab of a larger grid): read, not zero padding
    const bool halox = (g.flags & PRE_FLAG_HALO_X) != 0;
    bool hrow = hduty && (halox ? (hx >= -1 && hx <= g.X) : (hx >= 0 && hx < g.X)) && (hy < g.Yc);
    float hfill = 0.f;          // value of an out-of-domain halo row
    const int hslot = hbot ? NR + 1 : 0;
    // y-halo duty: the edge lanes of each wave (and of the tile) fetch one scalar
    const bool ledge = ((q & 63) == 0);
    bool redge = ((q & 63) == 63) || (q == TYQ - 1);
    bool lload = ledge && inb && (y > 0);
    bool rload = redge && inb && (y + 4 < g.Y);
    int yloff = -1, yroff = 4;  // element offsets of the y- / y+ scalar relative to the own float4
    float ylfill = 0.f, yrfill = 0.f;
    // a lane just beyond the computed width (the <= 3 last columns of an odd-width grid are left to the generic kernel)
    // holds the grid's column y in C.x: the y+ cell of the row's last computed quad, taken by the shuffle like any other
    const bool tailq = !BC && (x < g.X) && (y >= g.Yc) && (y < g.Y);
    if constexpr (BC) {
        if (hduty && hy < g.Yc && (hx == -1 || hx == g.X)) {
            const int m = hx < 0 ? bc.xlo : bc.xhi;
            hfill = hx < 0 ? bc.vxlo : bc.vxhi;
            hrow = m >= 0;
            hx = m >= 0 ? m : 0;
        }
        // (the fused BC entries take whole quads only: Yc == Y)
        if (inb && y == 0) { lload = bc.ylo >= 0; yloff = bc.ylo; ylfill = bc.vylo; }
        if (inb && y + 4 >= g.Y) { redge = true; rload = bc.yhi >= 0; yroff = bc.yhi - y; yrfill = bc.vyhi; }
    }
    // non-BC: ONE edge scalar per lane (y- for a wave's first lane, y+ for a wave's / tile's last lane)
    const bool eload = BC ? lload : (ledge ? lload : rload);
    const int eoff4 = 4 * (BC ? yloff : (ledge ? yloff : yroff));
    const float efill = BC ? ylfill : 0.f;

    // Addresses: a plane of a field of this sample is a wave-uniform BUFFER DESCRIPTOR (scalar registers, advanced by scalar
    // arithmetic), a thread's place in it a 32-bit byte offset - its own quad (voff), the halo-row float it fetches
    // (hoff), its edge scalar (voff + eoff4).  As 64-bit pointers these cost 4 registers per field, live across the
    // functor, plus a 64-bit vector add per load (round 3).  The descriptors are based one row BEFORE row 0, so that
    // row -1 (PRE_FLAG_HALO_X) has a non-negative offset; the host has checked that every offset fits 32 bits.
    unsigned int voff[F], hoff[F];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        voff[i] = (unsigned int)(((long long)(xl + 1) * g.sX[i] + y) * 4);
        hoff[i] = (unsigned int)(((long long)(hx + 1) * g.sX[i] + hy) * 4);
    }
    float *outp = g.out + (long long)b * g.oB + (long long)x * g.oX + y;
    const long long oT = g.oT;
    // the planes this workgroup may touch: all of them - or, when no operator has a tap along the marched axis (1-D residuals
    // on [1,B,T,X], spatial operators, D_x / Laplacians on 3-D fields), its own segment only: a segment then costs no window
    // prologue, and the host cuts the axis into marches of a few planes (TFREE_TSEG)
    const int tlo = g.tfree ? t0 : 0, thi = g.tfree ? t1 : g.T;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto plane = [&](int i, int t) __attribute__((always_inline)) {
        const float *p = g.f[i] + ((long long)b * g.sB[i] - g.sX[i] + (long long)t * g.sT[i]);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, -1, 0x00020000);
    };

    auto load_own = [&](int t, float4(&dst)[F]) __attribute__((always_inline)) {
        const bool ok = ldown && (t >= tlo) && (t < thi);
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if (ok) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(plane(i, t), (int)voff[i], 0, 0);
                dst[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            } else {
                dst[i] = f4(BC ? ghost : 0.f);
                if (tailq && (t >= tlo) && (t < thi))
                    dst[i].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane(i, t), (int)voff[i], 0, 0));
            }
        }
    };
    auto load_halo = [&](int t, Halo<F, BC, COOP> &h) __attribute__((always_inline)) {
        const bool okt = (t >= tlo) && (t < thi);
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if (!SX::has(i)) {                                   // no x-neighbours of this field are read: no halo row
                if constexpr (COOP) h.row[i] = 0.f; else h.row[i] = f4(0.f);
            } else if constexpr (COOP) {
                h.row[i] = (hrow && okt) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane(i, t), (int)hoff[i], 0, 0))
                                         : (BC ? hfill : 0.f);
            } else if (hrow && okt) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(plane(i, t), (int)hoff[i], 0, 0);
                h.row[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            } else {
                h.row[i] = f4(BC ? hfill : 0.f);
            }
            h.ye[i] = (eload && okt) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane(i, t), (int)voff[i] + eoff4, 0, 0))
                                     : (BC ? efill : 0.f);
            if constexpr (BC)
                h.yr[i] = (rload && okt) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane(i, t), (int)voff[i] + 4 * yroff, 0, 0))
                                         : yrfill;
        }
    };

    // One plane.  P,C,N hold planes t-1,t,t+1 of the own cells; D receives plane t+2;
    // hc is the halo of plane t, hn receives the halo of plane t+1.  The caller rotates the
    // roles instead of moving registers, so D/hn stay in flight until they are first read.
    auto step = [&](int t, float4(&P)[F], float4(&C)[F], float4(&N)[F], float4(&D)[F],
                    Halo<F, BC, COOP> &hc, Halo<F, BC, COOP> &hn) __attribute__((always_inline)) {
        const int bi = (t - t0) & 1;
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if (!SX::has(i)) continue;
            const int k = SX::slot(i);
            lds[bi][k][ty + 1][q] = C[i];
            if constexpr (COOP) {
                if (hduty) reinterpret_cast<float *>(&lds[bi][k][hslot][0])[hcol] = hc.row[i];
            } else {
                if (hduty) lds[bi][k][hslot][q] = hc.row[i];
            }
        }
        // halo first: it is consumed first (next plane's LDS staging), and vmcnt retires in
        // issue order, so the own-cell loads of plane t+2 stay in flight behind it
        // (measured +7 % on NS momentum vs the other order; non-temporal stores: -25 %; round 3: the own-cell loads with
        // the slc / nt bit -5 ... -12 % on every functor, with glc +-0)
        load_halo(t + 1, hn);
        load_own(t + AHEAD, D);
        if constexpr (SX::count > 0) lds_barrier();

        Nbr n[F];
#pragma unroll
        for (int i = 0; i < F; ++i) {
            n[i].c = C[i];
            n[i].tm = P[i];
            n[i].tp = N[i];
            if (SX::has(i)) {
                n[i].xm = lds[bi][SX::slot(i)][ty][q];
                n[i].xp = lds[bi][SX::slot(i)][ty + 2][q];
            } else {
                n[i].xm = n[i].xp = f4(__builtin_nanf(""));      // never read by the functor (or the result says so)
            }
            float lft = lane_below(C[i].w);
            float rgt = lane_above(C[i].x);
            lft = ledge ? hc.ye[i] : lft;
            rgt = redge ? (BC ? hc.yr[i] : hc.ye[i]) : rgt;
            n[i].ym = make_float4(lft, C[i].x, C[i].y, C[i].z);
            n[i].yp = make_float4(C[i].y, C[i].z, C[i].w, rgt);
        }
        float4 r = Fn::eval(n, prm);
        if (g.flags & PRE_FLAG_ABS) r = fabs4(r);
        if (inb) {
            stg4(outp + (long long)t * oT, r);
        }
    };

    Halo<F, BC, COOP> h0, h1;
    if constexpr (AHEAD == 3) {
        // experiment (MARCH6_AHEAD=3): the own cells of plane t + 3 are requested before plane t is computed - a five-plane
        // ring, five steps per trip; the halo pair has then flipped an odd number of times and is copied back (F + F/4
        // registers per five planes)
        float4 w0[F], w1[F], w2[F], w3[F], w4[F];
        load_own(t0 - 1, w0);
        load_own(t0, w1);
        load_own(t0 + 1, w2);
        load_own(t0 + 2, w3);
        load_halo(t0, h0);
        for (int t = t0; t < t1; t += 5) {
            step(t, w0, w1, w2, w4, h0, h1);
            if (t + 1 >= t1) break;
            step(t + 1, w1, w2, w3, w0, h1, h0);
            if (t + 2 >= t1) break;
            step(t + 2, w2, w3, w4, w1, h0, h1);
            if (t + 3 >= t1) break;
            step(t + 3, w3, w4, w0, w2, h1, h0);
            if (t + 4 >= t1) break;
            step(t + 4, w4, w0, w1, w3, h0, h1);
            h0 = h1;
        }
    } else {
        float4 w0[F], w1[F], w2[F], w3[F];
        load_own(t0 - 1, w0);
        load_own(t0, w1);
        load_own(t0 + 1, w2);
        load_halo(t0, h0);
        for (int t = t0; t < t1; t += 4) {
            step(t, w0, w1, w2, w3, h0, h1);
            if (t + 1 >= t1) break;
            step(t + 1, w1, w2, w3, w0, h1, h0);
            if (t + 2 >= t1) break;
            step(t + 2, w2, w3, w0, w1, h0, h1);
            if (t + 3 >= t1) break;
            step(t + 3, w3, w0, w1, w2, h1, h0);
        }
    }
}

// ------------------------------------------------------------------ host side
bool star_from_dense27(const float *K, Star *s)
{
    // axes (Nt,Nx,Ny); index (a,b,c) -> offset (a-1,b-1,c-1).  True iff all weight is on the star.
    auto at = [&](int a, int b, int c) { return K[(a * 3 + b) * 3 + c]; };
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            for (int c = 0; c < 3; ++c) {
                const int off = (a != 1) + (b != 1) + (c != 1);
                if (off > 1 && at(a, b, c) != 0.0f) return false;
            }
    s->c = at(1, 1, 1);
    s->tm = at(0, 1, 1); s->tp = at(2, 1, 1);
    s->xm = at(1, 0, 1); s->xp = at(1, 2, 1);
    s->ym = at(1, 1, 0); s->yp = at(1, 1, 2);
    return true;
}

struct Shape { bool t, x, y; };
Shape shape_of(const Star &s) { return {s.tm != 0.f || s.tp != 0.f, s.xm != 0.f || s.xp != 0.f, s.ym != 0.f || s.yp != 0.f}; }

// which compiled tap structure do (D_t, D_x, D_y[, Lap]) fit?  0 reference, 1 y-fixed, 2 general
int pick_mode(const Star &Dt, const Star &Dx, const Star &Dy, const Star *L)
{
    const Shape st = shape_of(Dt), sx = shape_of(Dx), sy = shape_of(Dy);
    const bool dt_ok = !st.x && !st.y, dx_ok = !sx.t && !sx.y;
    const bool lap_ok = !L || !shape_of(*L).t;
    if (dt_ok && dx_ok && lap_ok && !sy.x && !sy.y) return 0;
    if (dt_ok && dx_ok && lap_ok && !sy.x && !sy.t) return 1;
    return 2;
}

// ---- how finely to cut the marched axis (round 6) ------------------------------------------------------------------
// A workgroup marches a whole t segment; the chip holds `slots` workgroups at once, so a launch runs in rounds and the
// last round is as full as it happens to be: 3200 workgroups on 768 slots are 4.2 rounds - the fifth runs a sixth full
// and the launch takes 5 rounds' time (NS momentum on [800,20,256,256] Nt-fastest: 4.1 instead of 3.4 ms when a
// different chunk width changed nothing but that).  Segments cost their window prologue (two planes loaded without an
// output).  Chosen: the segment length that minimises (1 + 3 / tSeg) x (ceil(rounds) + 1/2) / rounds; rounds below 1 = the
// share of the chip that is busy at all (small problems: as many segments as the 16-plane floor allows, as before).
int chip_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

template <class K> int resident_per_cu(K kernel, int threads)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 1;
    return n;
}

int pick_tseg(long long tiles, int T, long long slots)
{
    int best = T;
    double bestc = 1e300;
    for (int tSeg = T;; tSeg = (tSeg + 1) / 2) {
        const long long wgs = tiles * ((T + tSeg - 1) / tSeg);
        const double rounds = (double)wgs / (double)slots;
        // (+ half a round: workgroups do not finish in lockstep, and the fewer the rounds the more of the launch is its ragged
        // end - [200,20,512,512] Nt-fastest: 1000 workgroups of 512 planes on 512 slots, "two full rounds", ran 3.6 % slower
        // than 4000 of 128)
        const double full = (rounds <= 1.0 ? 1.0 : (double)((wgs + slots - 1) / slots)) + 0.5;
        const double cost = (1.0 + 3.0 / tSeg) * full / rounds;     // (planes t0 - 1, t1 and the prefetched t1 + 1 are read for nothing)
        if (cost < bestc * 0.99) { best = tSeg; bestc = cost; }       // (near-ties go to the longer segments)
        if (tSeg <= 16) break;
    }
    return best;
}

template <class Fn, int NR, int TYQ, bool BC = false>
int launch_tiled(Geom &g, const typename Fn::Params &prm, hipStream_t st, const BCInfo *bc = nullptr)
{
    static_assert(2 * Staged<Fn>::FX * (NR + 2) * TYQ * 16 <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    {
    g.nXT = (g.X + NR - 1) / NR;
    g.nYT = (g.Yc + 4 * TYQ - 1) / (4 * TYQ);
    for (int i = 0; i < Fn::F; ++i)            // a thread's place in a plane is a 32-bit byte offset (from one row before row 0)
        if (g.sX[i] < 0 || ((long long)(g.X + 2 + NR) * g.sX[i] + g.Y + 8) * 4 >= (1LL << 32)) return PRE_E_UNSUPPORTED;
    // split long T axes so that the grid fills the chip and its last round of workgroups is nearly full, without paying
    // the 2-plane window prologue too often (pick_tseg)
    long long tiles = (long long)g.B * g.nXT * g.nYT;
    static const int per_cu = resident_per_cu(march_kernel<Fn, NR, TYQ, BC>, NR * TYQ);
    int tSeg = pick_tseg(tiles, g.T, (long long)per_cu * chip_cus());
    // (experiment, off: one-field functors WITH t-taps on batched 3-D fields - the wave kernel - run 3-4 % faster as marches of
    // 8 planes than of the whole T axis, profiles/r06/march_ab_one_field_tseg.txt, but the three planes a segment reads for
    // nothing show up at the fabric: FETCH 1.007x -> 1.11x algorithmic.  The tap-free rule below has no such cost.)
    if (MARCH1_TSEG > 0 && Fn::F == 1 && (g.B > 1 || MARCH1_ANYB) && g.T >= 4 * MARCH1_TSEG && tSeg > MARCH1_TSEG) tSeg = MARCH1_TSEG;
    // no tap along the marched axis: segments are free (the kernel loads its own planes only) - marches of TFREE_TSEG planes
    if (TFREE_TSEG > 0 && g.tfree && tSeg > TFREE_TSEG) tSeg = TFREE_TSEG;
    if (MARCHN_TSEG > 0 && tSeg > MARCHN_TSEG) tSeg = MARCHN_TSEG;             // (experiment: a cap for every functor)
    g.tSeg = tSeg;
    g.nTSeg = (g.T + tSeg - 1) / tSeg;
    tiles *= g.nTSeg;
    if (tiles <= 0 || tiles * TYQ > 0xffffffffLL) return PRE_E_SHAPE;      // the dispatch packet counts work-items in 32 bits
    if constexpr (BC) {
        hipLaunchKernelGGL((march_kernel<Fn, NR, TYQ, true>), dim3((unsigned)tiles), dim3(TYQ, NR), 0, st, g, prm, *bc);
    } else {
        hipLaunchKernelGGL((march_kernel<Fn, NR, TYQ, false>), dim3((unsigned)tiles), dim3(TYQ, NR), 0, st, g, prm, NoBC{});
    }
    PRE_LAUNCH_CHECK();
    return PRE_OK;
    }
}

// ------------------------------------------------------------------ flat form: short contiguous axis
// Kernel axes (t, x, y) with a SHORT y extent Ty whose rows follow each other in memory (x stride == Ty): the
// (x, y) plane is one contiguous row of L = X*Ty cells.  A workgroup owns a chunk of 512 quads of that merged
// row and marches over t exactly as above (register window, prefetch), but the neighbours within the plane are
// all taken from the flat LDS copy of the chunk (+ a halo of 32 quads per side):
//   x -/+  =  Ty cells back / ahead in the merged row (beyond the row = the zero padding of x = -1 / X),
//   y -/+  =  the previous / next cell, masked where that cell belongs to the neighbouring x row (y = -1 / Ty).
// Every lane works whatever Ty is, and Ty itself need not be a multiple of 4 (only L).  Used for the surrogate's
// native Nt-fastest layout with Nt < 96 (after the axis relabelling: t = Nx, x = Ny, y = Nt) and for narrow grids in
// the reference layout (below 96 columns the regular kernel is down to its 64-column tile: measured at 80 columns
// wave 3.7 -> 4.9, MHD induction 3.4 -> 4.4 TB/s, NS momentum 3.7 -> 3.5; from 100 columns up the regular tiles win).
constexpr int FLAT_NT = 512, FLAT_H = 32;
#ifndef FLAT_MAX_Y
#define FLAT_MAX_Y 96           // contiguous extents below this take the flat (merged-axis) form when the layout allows
#endif
#ifndef FLAT_NT_GAIN
#define FLAT_NT_GAIN 8          // a narrower chunk must save this many per cent of a row's lanes to be taken (measured:
                                // profiles/r06/flat_ab_chunk_width.txt - 4 chunks of 320 lost 7 % to 3 of 448 at Nt = 20, 256 wide)
#endif
#ifndef FLAT_Q4
#define FLAT_Q4 0              // 1: a second instantiation for Ty % 4 == 0 without six of the eight row-end masks per field -
                               // measured -0.2 ... -0.4 % (profiles/r06/flat_ab_q4.txt): not worth doubling the instantiations
#endif
#ifndef FLAT_SYNC
#define FLAT_SYNC 0            // experiment: a bare s_barrier per plane in the kernels that stage nothing
#endif
#ifndef FLAT_NOLDS_NT
#define FLAT_NOLDS_NT 512       // widest chunk of a functor that stages nothing (no LDS, no barrier: the workgroup size is free)
#endif

// Round 6: only the fields a functor reads x-NEIGHBOURS of (Staged<Fn>, as in march_kernel) go through LDS, and their halo is
// as wide as an x-neighbour is far - ceil(Ty / 4) quads per side instead of a fixed 32 (Ty = 10 on a 256-wide grid: 6 halo
// quads per 320-quad chunk instead of 64 - a fifth of the chunk's loads).  A field that is read at the centre and along
// t / y only takes its y-neighbours from the adjacent lane (wavefront shuffle), the first / last lane of a wave fetching one
// scalar - no LDS store, no halo, and when no field is staged (every MHD functor in the surrogate's Nt-fastest layout: its
// taps lie on the kernel's t and y axes) no LDS and no barrier at all.
#ifndef FLAT_STAGE_ALL
#define FLAT_STAGE_ALL 0        // experiment: 1 = every field through LDS (rounds 2-5), whatever the functor reads
#endif
#ifndef FLAT_HALO_FULL
#define FLAT_HALO_FULL 0        // experiment: 1 = FLAT_H halo quads per side whatever Ty (rounds 2-5)
#endif
template <class Fn> struct AllStaged {
    static constexpr int count = Fn::F, FX = Fn::F;
    static __device__ __forceinline__ constexpr bool has(int) { return true; }
    static __device__ __forceinline__ constexpr int slot(int i) { return i; }
};
template <class Fn> using FlatStaged = typename std::conditional<FLAT_STAGE_ALL != 0, AllStaged<Fn>, Staged<Fn>>::type;

template <int F> struct FlatHalo {
    float4 q[F];     // staged fields: the halo quad this thread fetches
    float e[F];      // unstaged fields: the y- cell of a wave's first lane / the y+ cell of its last lane
};

// Q4 (experiment, -DFLAT_Q4=1): Ty is a multiple of 4 (Nt = 64, 40, 20, ...): a quad never straddles a row end, so only its
// first cell can lack a y- neighbour and only its last a y+ one - six of the eight masks per field are gone.
template <class Fn, bool Q4>
__global__ void __launch_bounds__(FLAT_NT, MinWaves<Fn>::value)
flat_march_kernel(const Geom g, const typename Fn::Params prm)
{
    constexpr int F = Fn::F;
    using SX = FlatStaged<Fn>;
    constexpr bool ANY = SX::count > 0;
    __shared__ float4 lds[2][SX::FX][ANY ? FLAT_NT + 2 * FLAT_H : 1];
    const int q = threadIdx.x;
    unsigned Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ch = Lb % g.nYT; Lb /= g.nYT;
    const int ts = Lb % g.nTSeg;
    const int b = Lb / g.nTSeg;
    const int Ty = g.Y, L = g.X * g.Y;
    // threads per chunk: 512, or fewer when that wastes fewer lanes on the last chunk of a row (chosen by the host;
    // the LDS image is sized for 512 either way)
    const int NT = blockDim.x;
    const int m0 = ch * NT * 4, m = m0 + 4 * q;
    const bool inb = m < L;
    int t0 = ts * g.tSeg, t1 = min(t0 + g.tSeg, g.T);
    if (g.flags & PRE_FLAG_INTERIOR_T) {       // the caller crops the t rim: neither compute nor store it
        t0 = max(t0, 1);
        t1 = min(t1, g.T - 1);
    }

    // halo duty (staged fields): the first / last HQ threads fetch one quad left / right of the chunk, HQ = the quads an
    // x-neighbour (Ty cells away) can reach into (L % 4 == 0: a quad is entirely inside the row or entirely padding).
    // The LDS image keeps room for FLAT_H quads per side: the left halo ends at slot FLAT_H, the right one starts at FLAT_H + NT.
    const int HQ = FLAT_HALO_FULL ? FLAT_H : min(FLAT_H, (Ty + 3) >> 2);
    const bool hl = q < HQ, hr = q >= NT - HQ;
    const int hm = hl ? m0 - 4 * (HQ - q) : m0 + 4 * NT + 4 * (q - (NT - HQ));
    const bool hok = (hl || hr) && hm >= 0 && hm < L;
    const int hslot = hl ? FLAT_H - HQ + q : FLAT_H + NT + (q - (NT - HQ));

    // which of my four cells have a y- / y+ neighbour inside their own x row
    bool lok[4], rok[4];
    {
        int ph = m % Ty;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lok[j] = (Q4 && j > 0) || ph != 0;
            rok[j] = (Q4 && j < 3) || ph != Ty - 1;
            ph = ph + 1 == Ty ? 0 : ph + 1;
        }
    }
    // edge duty (unstaged fields): a wave's first lane fetches the cell before its quad, its last lane the cell after -
    // if that cell is a y-neighbour at all (same x row; the row's last cell has rok == false, so nothing beyond L is read)
    const bool ledge = (q & 63) == 0, redge = (q & 63) == 63;
    const bool eload = inb && (ledge ? lok[0] : (redge && rok[3]));

    // a plane of a field of this sample = a wave-uniform buffer descriptor; the thread's own quad and its halo quad are
    // two 32-bit byte offsets shared by every field (the flat form takes fields of one in-plane layout): as 64-bit
    // pointers they cost 4 registers per field (JOREK temperature in its native layout: 139 registers, one workgroup per CU)
    const unsigned int voff = (unsigned int)m * 4u, hoff = (unsigned int)hm * 4u;       // (hm < 0: never loaded)
    const unsigned int eoff = ledge ? voff - 4u : voff + 16u;                           // (never loaded where it would be outside)
    float *outp = g.out + (long long)b * g.oB + m;
    const long long oT = g.oT;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto rsrc = [&](int i, int t) __attribute__((always_inline)) {
        const float *p = g.f[i] + ((long long)b * g.sB[i] + (long long)t * g.sT[i]);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, -1, 0x00020000);
    };
    auto quad = [&](int i, int t, unsigned int off) __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc(i, t), (int)off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };

    auto load_own = [&](int t, float4(&dst)[F]) __attribute__((always_inline)) {
        const bool ok = inb && (t >= 0) && (t < g.T);
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if (ok) dst[i] = quad(i, t, voff);
            else dst[i] = f4(0.f);
        }
    };
    auto load_halo = [&](int t, FlatHalo<F> &h) __attribute__((always_inline)) {
        const bool okt = (t >= 0) && (t < g.T);
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if (SX::has(i)) {
                h.e[i] = 0.f;
                if (hok && okt) h.q[i] = quad(i, t, hoff);
                else h.q[i] = f4(0.f);
            } else {
                h.q[i] = f4(0.f);
                h.e[i] = (eload && okt) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc(i, t), (int)eoff, 0, 0)) : 0.f;
            }
        }
    };

    auto step = [&](int t, float4(&P)[F], float4(&C)[F], float4(&N)[F], float4(&D)[F], FlatHalo<F> &hc,
                    FlatHalo<F> &hn) __attribute__((always_inline)) {
        const int bi = (t - t0) & 1;
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if (!SX::has(i)) continue;
            const int k = SX::slot(i);
            lds[bi][k][FLAT_H + q] = C[i];
            if (hl || hr) lds[bi][k][hslot] = hc.q[i];
        }
        load_halo(t + 1, hn);
        load_own(t + 2, D);
        if constexpr (ANY) lds_barrier();
        else if (FLAT_SYNC) __builtin_amdgcn_s_barrier();       // (no data to order: keeps the workgroup's waves on one plane)

        Nbr n[F];
#pragma unroll
        for (int i = 0; i < F; ++i) {
            n[i].c = C[i];
            n[i].tm = P[i];
            n[i].tp = N[i];
            float lft, rgt;
            if (SX::has(i)) {
                const int k = SX::slot(i);
                const float *row = reinterpret_cast<const float *>(&lds[bi][k][0]) + 4 * (FLAT_H + q);      // my first cell
                if (Q4) {                                // the x neighbours are whole quads
                    n[i].xm = lds[bi][k][FLAT_H + q - (Ty >> 2)];
                    n[i].xp = lds[bi][k][FLAT_H + q + (Ty >> 2)];
                } else if ((Ty & 1) == 0) {              // Ty = 10, 30, 50 (T_out of the reference scripts): 8-byte aligned pairs
                    const float2 a = *reinterpret_cast<const float2 *>(row - Ty), b = *reinterpret_cast<const float2 *>(row + 2 - Ty);
                    const float2 c = *reinterpret_cast<const float2 *>(row + Ty), d = *reinterpret_cast<const float2 *>(row + 2 + Ty);
                    n[i].xm = make_float4(a.x, a.y, b.x, b.y);
                    n[i].xp = make_float4(c.x, c.y, d.x, d.y);
                } else {
                    n[i].xm = make_float4(row[-Ty], row[1 - Ty], row[2 - Ty], row[3 - Ty]);
                    n[i].xp = make_float4(row[Ty], row[Ty + 1], row[Ty + 2], row[Ty + 3]);
                }
                lft = row[-1];
                rgt = row[4];
            } else {
                n[i].xm = n[i].xp = f4(__builtin_nanf(""));      // never read by the functor (or the result says so)
                lft = lane_below(C[i].w);
                rgt = lane_above(C[i].x);
                lft = ledge ? hc.e[i] : lft;
                rgt = redge ? hc.e[i] : rgt;
            }
            if constexpr (Q4) {
                n[i].ym = make_float4(lok[0] ? lft : 0.f, C[i].x, C[i].y, C[i].z);
                n[i].yp = make_float4(C[i].y, C[i].z, C[i].w, rok[3] ? rgt : 0.f);
            } else {
                n[i].ym = make_float4(lok[0] ? lft : 0.f, lok[1] ? C[i].x : 0.f, lok[2] ? C[i].y : 0.f, lok[3] ? C[i].z : 0.f);
                n[i].yp = make_float4(rok[0] ? C[i].y : 0.f, rok[1] ? C[i].z : 0.f, rok[2] ? C[i].w : 0.f, rok[3] ? rgt : 0.f);
            }
        }
        float4 r = Fn::eval(n, prm);
        if (g.flags & PRE_FLAG_ABS) r = fabs4(r);
        if (inb) stg4(outp + (long long)t * oT, r);
    };

    float4 w0[F], w1[F], w2[F], w3[F];
    FlatHalo<F> h0, h1;
    load_own(t0 - 1, w0);
    load_own(t0, w1);
    load_own(t0 + 1, w2);
    load_halo(t0, h0);
    for (int t = t0; t < t1; t += 4) {
        step(t, w0, w1, w2, w3, h0, h1);
        if (t + 1 >= t1) break;
        step(t + 1, w1, w2, w3, w0, h1, h0);
        if (t + 2 >= t1) break;
        step(t + 2, w2, w3, w0, w1, h0, h1);
        if (t + 3 >= t1) break;
        step(t + 3, w3, w0, w1, w2, h1, h0);
    }
}

template <class Fn>
int launch_flat(Geom &g, const typename Fn::Params &prm, hipStream_t st)
{
    static_assert(2 * FlatStaged<Fn>::FX * (FLAT_NT + 2 * FLAT_H) * 16 <= 160 * 1024, "chunk does not fit the 160 KiB LDS");
    g.nXT = 1;
    if ((long long)g.X * g.Y >= (1LL << 30)) return PRE_E_UNSUPPORTED;      // a thread's place in a plane is a 32-bit byte offset
    // chunk = 512 quads, or 448 / 384 / 320 / 256 when that leaves fewer idle lanes in the row's last chunk (the
    // surrogate's Nt = 10 on a 256-wide grid is a row of 640 quads: two chunks of 320 instead of 512 + 128)
    const long long quads = (long long)g.X * g.Y / 4;
    // cost of a row = chunks x (quads + the halo quads staged per chunk: ceil(Ty / 4) per side for a functor that stages
    // any field, none otherwise); ties go to the wider chunk
    const int halo = FLAT_HALO_FULL ? 2 * FLAT_H : FlatStaged<Fn>::count > 0 ? 2 * ((g.Y + 3) / 4 < FLAT_H ? (g.Y + 3) / 4 : FLAT_H) : 0;
    int nt = FlatStaged<Fn>::count > 0 ? FLAT_NT : FLAT_NOLDS_NT;
    for (int c = nt - 64; c >= 256; c -= 64)
        if ((quads + c - 1) / c * (c + halo) * 100 < (quads + nt - 1) / nt * (nt + halo) * (100 - FLAT_NT_GAIN)) nt = c;
    g.nYT = (int)((quads + nt - 1) / nt);
    long long tiles = (long long)g.B * g.nYT;
    const bool q4 = FLAT_Q4 && (g.Y & 3) == 0;
    // (by form and chunk width; 0 = not asked yet.  Host threads may race to fill a slot: they write the same value)
    static std::atomic<int> per_cu[2][FLAT_NT / 64 + 1] = {};
    int occ = per_cu[q4][nt / 64].load(std::memory_order_relaxed);
    if (!occ) {
        occ = q4 ? resident_per_cu(flat_march_kernel<Fn, FLAT_Q4 != 0>, nt) : resident_per_cu(flat_march_kernel<Fn, false>, nt);
        per_cu[q4][nt / 64].store(occ, std::memory_order_relaxed);
    }
    const int tSeg = pick_tseg(tiles, g.T, (long long)occ * chip_cus());
    g.tSeg = tSeg;
    g.nTSeg = (g.T + tSeg - 1) / tSeg;
    tiles *= g.nTSeg;
    if (tiles <= 0 || tiles * nt > 0xffffffffLL) return PRE_E_SHAPE;
    if (q4) hipLaunchKernelGGL((flat_march_kernel<Fn, FLAT_Q4 != 0>), dim3((unsigned)tiles), dim3(nt), 0, st, g, prm);
    else hipLaunchKernelGGL((flat_march_kernel<Fn, false>), dim3((unsigned)tiles), dim3(nt), 0, st, g, prm);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

template <class Fn, bool BC = false>
int launch(Geom &g, const typename Fn::Params &prm, hipStream_t st, const BCInfo *bc = nullptr)
{
    if constexpr (!BC)
        if (g.flat) return launch_flat<Fn>(g, prm, st);
    // 512 threads per workgroup; rows of the tile trade halo re-reads (2/NR) against columns covered
    // 8 rows x 256 columns: measured best of {4,8,16} rows (16 rows = 1024 threads, one workgroup per CU: -5 %)
    // (also measured: 16x128 and 32x64 tiles at 512 threads, -2..-7 % on every functor)
    // smaller workgroups for the register-heavy MHD functors (to fit 3 instead of 1 per CU) measured in round 2
    // (profiles/r02/tile_ab_mhd.txt): 4x64 -2..-12 %, 8x32 -1..-14 %, 16x16 0..-7 % at [1024,64,256,256]; only at
    // T = 10 do 8x32 / 16x16 gain (+6 % induction, +2 % momentum / energy): not worth 45 more instantiations;
    // 12x64 (768 threads: 3 waves per SIMD where the 82 KB tile of 8x64 leaves induction 2) -5..-8 %;
    // own cells prefetched three planes ahead instead of two on induction (200 VGPRs): within the +-5 % run-to-run noise;
    // induction capped at 128 VGPRs (17 dwords spilled) so that two of its 80 KB workgroups share a CU: -30 %
    // round 3, x-slabs (64-plane marches: x-neighbour tiles drift apart, 8 % of the input is fetched twice): NS momentum
    // 16x64 (1024 threads) 50.5 ms, 16x32 48.5, 32x16 50.7 against 47.4 for 8x64 (gpurun_out/r3b/nr_*.log)
    if constexpr (Fn::F >= 5)
        if (g.Y >= 192) return launch_tiled<Fn, MARCH6_NR, MARCH6_TYQ, BC>(g, prm, st, bc);
    if (g.Y >= 192) return launch_tiled<Fn, 8, 64, BC>(g, prm, st, bc);
    if (g.Y >= 96) return launch_tiled<Fn, 16, 32, BC>(g, prm, st, bc);
    return launch_tiled<Fn, 32, 16, BC>(g, prm, st, bc);
}

// Fill the kernel geometry from the caller's views and RELABEL the axes so that the kernel's
// contiguous "y" axis is whichever of (T, X, Y) has unit stride in every view:
//   Y contiguous (reference layout [BS,Nt,Nx,Ny])            -> identity
//   T contiguous (surrogate output [BS,F,Nx,Ny,Nt] seen through permute(0,1,4,2,3),
//                 Marginal/Wave_Residuals_CP.py:216)          -> kernel axes (X, Y, T)
//   X contiguous                                              -> kernel axes (T, Y, X)
// The star weights are permuted with the axes, the output view must share the layout, and the
// arithmetic is unchanged - zero-copy for the layouts real callers hand in.  Returns
// PRE_E_UNSUPPORTED when no common unit-stride axis / alignment exists.
int prepare(Geom &g, int &relabeled, const pre_field_t *const *fs, int nf, const pre_out_t *out,
            int64_t B, int64_t T, int64_t X, int64_t Y, int flags, Star *const *stars, int nstars, bool relaxed = false,
            bool allow_flat = true)
{
    if (!out || !out->ptr || B <= 0 || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (B > 0x7fffffff || T > 0x7fffffff || X > 0x7fffffff || Y > 0x7fffffff) return PRE_E_SHAPE;
    for (int i = 0; i < nf; ++i)
        if (!fs[i] || !fs[i]->ptr) return PRE_E_NULL;
    const int64_t D[3] = {T, X, Y};
    auto stride = [](const pre_field_t *f, int ax) { return ax == 0 ? f->sT : ax == 1 ? f->sX : f->sY; };
    auto ostride = [&](int ax) { return ax == 0 ? out->sT : ax == 1 ? out->sX : out->sY; };
    auto all_unit = [&](int ax) {
        if (ostride(ax) != 1) return false;
        for (int i = 0; i < nf; ++i)
            if (stride(fs[i], ax) != 1) return false;
        return true;
    };
    int p[3];
    if (all_unit(2)) { p[0] = 0; p[1] = 1; p[2] = 2; }
    else if (all_unit(0)) { p[0] = 1; p[1] = 2; p[2] = 0; }
    else if (all_unit(1)) { p[0] = 0; p[1] = 2; p[2] = 1; }
    else return PRE_E_UNSUPPORTED;
    relabeled = p[2] == 2 ? 0 : (p[2] == 0 ? 1 : 2);      // 0 identity, 1 kernel axes (X,Y,T), 2 kernel axes (T,Y,X)
    // relaxed (single-field linear operators): any contiguous extent >= 4 (the last extent % 4 columns are
    // left to the caller); fused multi-field kernels need extent % 4 == 0.  No alignment requirement
    // beyond the 4 bytes of a float: the float4 accesses are unaligned-capable (F4u).
    // A SHORT contiguous axis (the surrogate's Nt = 10..40 in its native [BS,F,Nx,Ny,Nt] layout) would leave most
    // lanes of a row idle.  When the next axis is contiguous with it (stride == extent) the two are merged into
    // one long axis for the flat form of the kernel: only their product has to be a multiple of 4.
    if ((flags & PRE_FLAG_HALO_X) && relabeled) return PRE_E_UNSUPPORTED;      // the halo rows are on the caller's x axis
    if (flags & PRE_FLAG_HALO_X) allow_flat = false;                           // (the flat form pads x with zeros)
    bool flat = allow_flat && D[p[2]] < FLAT_MAX_Y && ostride(p[1]) == D[p[2]] && (D[p[1]] * D[p[2]]) % 4 == 0 && D[p[1]] > 1;
    for (int i = 0; i < nf; ++i) flat = flat && stride(fs[i], p[1]) == D[p[2]];
    g.flat = flat;
    g.tfree = 0;
    if (!flat && (relaxed ? D[p[2]] < 4 : D[p[2]] % 4 != 0)) return PRE_E_UNSUPPORTED;
    for (int i = 0; i < nf; ++i) {
        g.f[i] = fs[i]->ptr; g.sB[i] = fs[i]->sB; g.sT[i] = stride(fs[i], p[0]); g.sX[i] = stride(fs[i], p[1]);
    }
    for (int i = nf; i < MAXF; ++i) { g.f[i] = nullptr; g.sB[i] = g.sT[i] = g.sX[i] = 0; }
    g.out = out->ptr; g.oB = out->sB; g.oT = ostride(p[0]); g.oX = ostride(p[1]);
    g.B = (int)B; g.T = (int)D[p[0]]; g.X = (int)D[p[1]]; g.Y = (int)D[p[2]];
    g.Yc = flat ? g.Y : (g.Y & ~3);
    g.flags = relabeled ? (flags & ~PRE_FLAG_INTERIOR_T) : flags;     // the skipped rim is on the LOGICAL t axis
    if (flags & PRE_FLAG_OUT_INTERIOR_T) {
        // `out` holds planes 1..T-2 only: address it as if plane 0 existed one plane stride before its base; the
        // kernels never touch planes 0 and T-1 under PRE_FLAG_INTERIOR_T
        if (relabeled || T < 3) return PRE_E_UNSUPPORTED;
        g.out -= g.oT;
        g.flags |= PRE_FLAG_INTERIOR_T;
    }
    if (relabeled)
        for (int k = 0; k < nstars; ++k) {
            const Star o = *stars[k];
            const float m[3] = {o.tm, o.xm, o.ym}, q[3] = {o.tp, o.xp, o.yp};
            stars[k]->tm = m[p[0]]; stars[k]->tp = q[p[0]];
            stars[k]->xm = m[p[1]]; stars[k]->xp = q[p[1]];
            stars[k]->ym = m[p[2]]; stars[k]->yp = q[p[2]];
        }
    return PRE_OK;
}

// tap structure after the relabelling: the Nt-fastest permutation maps modes 0/1 to 3/4,
// every other permuted layout runs the general-star instantiation
inline int relabeled_mode(int mode, int rel) { return rel == 0 ? mode : (rel == 1 && mode < 2 ? mode + 3 : 2); }

// no operator has a tap along the kernel's marched axis (after prepare() has relabelled the stars): the marching kernel then
// loads a segment's own planes only (Geom::tfree)
inline int no_t_taps(Star *const *stars, int n)
{
    for (int k = 0; k < n; ++k)
        if (stars[k]->tm != 0.f || stars[k]->tp != 0.f) return 0;
    return 1;
}

template <template <int> class FnT, class P>
int launch_mode(int mode, Geom &g, const P &prm, hipStream_t st)
{
    if (mode == 0) return launch<FnT<0>>(g, prm, st);
    if (mode == 1) return launch<FnT<1>>(g, prm, st);
    if (mode == 3) return launch<FnT<3>>(g, prm, st);
    if (mode == 4) return launch<FnT<4>>(g, prm, st);
    return launch<FnT<2>>(g, prm, st);
}

}  // namespace

// Internal: called by stencil_generic.hip when a tap list is star-shaped and the layout allows it.
// On success *tail_axis / *tail_from describe the columns the streaming kernel did NOT compute (the last
// extent % 4 cells of the contiguous axis, given as the caller's axis 0=T,1=X,2=Y and first index), or
// *tail_axis = -1 if everything was computed.
int pre_star_try_linear1(const pre_field_t *in, const pre_out_t *out, const float star7[7],
                         int64_t B, int64_t T, int64_t X, int64_t Y, int flags, hipStream_t st,
                         int *tail_axis, int64_t *tail_from)
{
    const pre_field_t *fs[1] = {in};
    Linear1::Params p;
    p.s = Star{star7[0], star7[1], star7[2], star7[3], star7[4], star7[5], star7[6]};
    Star *stars[1] = {&p.s};
    Geom g;
    int rel;
    int rc = prepare(g, rel, fs, 1, out, B, T, X, Y, flags, stars, 1, true);
    if (rc) return rc;
    *tail_axis = -1;
    g.tfree = no_t_taps(stars, 1);
    if (g.Yc < g.Y && (flags & PRE_FLAG_HALO_X)) return PRE_E_UNSUPPORTED;      // (the tail pass pads x with zeros)
    if (g.Yc < g.Y) {
        *tail_axis = rel == 0 ? 2 : (rel == 1 ? 0 : 1);
        *tail_from = g.Yc;
        g.flags &= ~PRE_FLAG_INTERIOR_T;           // keep it simple: the tail pass computes every plane
    }
    return launch<Linear1>(g, p, st);
}

extern "C" {

int pre_residual_ns_momentum_f32(const pre_field_t *u, const pre_field_t *v, const pre_field_t *p, const pre_out_t *out,
                                 const float *K_t, const float *K_x, const float *K_y, const float *K_xx_yy,
                                 float dt, float dx, float dy, float nu,
                                 int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!K_t || !K_x || !K_y || !K_xx_yy) return PRE_E_NULL;
    const pre_field_t *fs[3] = {u, v, p};
    NSParams prm;
    if (!star_from_dense27(K_t, &prm.Dt) || !star_from_dense27(K_x, &prm.Dx) ||
        !star_from_dense27(K_y, &prm.Dy) || !star_from_dense27(K_xx_yy, &prm.L))
        return PRE_E_UNSUPPORTED;
    const int mode = pick_mode(prm.Dt, prm.Dx, prm.Dy, &prm.L);      // on the caller's axes
    Star *stars[4] = {&prm.Dt, &prm.Dx, &prm.Dy, &prm.L};
    Geom g;
    int rel;
    int rc = prepare(g, rel, fs, 3, out, B, T, X, Y, flags, stars, 4);
    if (rc) return rc;
    prm.dxdy = dx * dy; prm.dtdy = dt * dy; prm.dtdx = dt * dx; prm.nudt = nu * dt;
    g.tfree = no_t_taps(stars, 4);
    return launch_mode<NSMomentum>(relabeled_mode(mode, rel), g, prm, as_stream(stream));
}

int pre_residual_linear2_f32(const pre_field_t *f0, const pre_field_t *f1, const pre_out_t *out,
                             const float *K_a, const float *K_b, float ratio,
                             int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!K_a || !K_b) return PRE_E_NULL;
    const pre_field_t *fs[2] = {f0, f1};
    Linear2::Params prm;
    if (!star_from_dense27(K_a, &prm.a) || !star_from_dense27(K_b, &prm.b)) return PRE_E_UNSUPPORTED;
    Star *stars[2] = {&prm.a, &prm.b};
    Geom g;
    int rel;
    int rc = prepare(g, rel, fs, 2, out, B, T, X, Y, flags, stars, 2);
    if (rc) return rc;
    prm.ratio = ratio;
    g.tfree = no_t_taps(stars, 2);
    return launch<Linear2>(g, prm, as_stream(stream));
}

int pre_residual_burgers_f32(const float *u, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                             const float *K_t, const float *K_x, const float *K_xx,
                             float dx, float dt, float nu, float c3,
                             int64_t B, int64_t T, int64_t X, int flags, void *stream)
{
    if (!u || !in_strides || !out || !out_strides || !K_t || !K_x || !K_xx) return PRE_E_NULL;
    if (flags & (PRE_FLAG_OUT_INTERIOR_T | PRE_FLAG_HALO_X)) return PRE_E_UNSUPPORTED;     // [B,T,X]: the marched axis is the batch
    // [B,T,X] -> [1, B, T, X]; 3x3 kernel (a over Nt, b over Nx) -> dense27 index (1, a, b)
    pre_field_t f{u, 0, in_strides[0], in_strides[1], in_strides[2]};
    pre_out_t o{out, 0, out_strides[0], out_strides[1], out_strides[2]};
    const pre_field_t *fs[1] = {&f};
    float d27[3][27] = {};
    const float *k9[3] = {K_t, K_x, K_xx};
    for (int op = 0; op < 3; ++op)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) d27[op][(1 * 3 + a) * 3 + c] = k9[op][a * 3 + c];
    BurgersParams prm;
    if (!star_from_dense27(d27[0], &prm.Dt) || !star_from_dense27(d27[1], &prm.Dx) || !star_from_dense27(d27[2], &prm.Dxx))
        return PRE_E_UNSUPPORTED;
    // mode 0 needs D_t purely along Nt (our x) and D_x, D_xx purely along Nx (our y)
    const Shape a = shape_of(prm.Dt), b2 = shape_of(prm.Dx), c2 = shape_of(prm.Dxx);
    const int mode = (!a.y && !b2.x && !c2.x) ? 0 : 2;
    Star *stars[3] = {&prm.Dt, &prm.Dx, &prm.Dxx};
    Geom g;
    int rel;
    int rc = prepare(g, rel, fs, 1, &o, 1, B, T, X, flags, stars, 3);
    if (rc) return rc;
    prm.dx = dx; prm.dt = dt; prm.nu = nu; prm.c3 = c3;
    g.tfree = no_t_taps(stars, 3);             // (always: the marched axis of [1,B,T,X] is the batch)
    return launch_mode<Burgers>(rel == 0 ? mode : (rel == 2 && mode == 0 ? 3 : 2), g, prm, as_stream(stream));
}

int pre_residual_mhd_f32(int eq, const pre_field_t fields[6], const pre_out_t *out,
                         const float *K_t, const float *K_x, const float *K_y, double gamma,
                         int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!fields || !K_t || !K_x || !K_y) return PRE_E_NULL;
    if (eq < 0 || eq > 3) return PRE_E_RANGE;
    MHDParams prm;
    if (!star_from_dense27(K_t, &prm.Dt) || !star_from_dense27(K_x, &prm.Dx) || !star_from_dense27(K_y, &prm.Dy))
        return PRE_E_UNSUPPORTED;
    prm.gamma = (float)gamma;
    prm.gm2 = (float)(gamma - 2.0);   // "(gamma-2)" is a float64 Python scalar in the reference
    int mode = pick_mode(prm.Dt, prm.Dx, prm.Dy, nullptr);
    const pre_field_t *all[6] = {&fields[0], &fields[1], &fields[2], &fields[3], &fields[4], &fields[5]};
    Star *stars[3] = {&prm.Dt, &prm.Dx, &prm.Dy};
    Geom g;
    int rel;
    hipStream_t st = as_stream(stream);
    if (eq == 0) {
        const pre_field_t *fs[3] = {all[0], all[1], all[2]};
        int rc = prepare(g, rel, fs, 3, out, B, T, X, Y, flags, stars, 3);
        if (rc) return rc;
        g.tfree = no_t_taps(stars, 3);
        return launch_mode<MHDContinuity>(relabeled_mode(mode, rel), g, prm, st);
    }
    if (eq == 3) {
        const pre_field_t *fs[4] = {all[1], all[2], all[4], all[5]};
        int rc = prepare(g, rel, fs, 4, out, B, T, X, Y, flags, stars, 3);
        if (rc) return rc;
        g.tfree = no_t_taps(stars, 3);
        return launch_mode<MHDInduction>(relabeled_mode(mode, rel), g, prm, st);
    }
    int rc = prepare(g, rel, all, 6, out, B, T, X, Y, flags, stars, 3);
    if (rc) return rc;
    g.tfree = no_t_taps(stars, 3);
    if (eq == 1) return launch_mode<MHDMomentum>(relabeled_mode(mode, rel), g, prm, st);
    return launch_mode<MHDEnergy>(relabeled_mode(mode, rel), g, prm, st);
}

int pre_residual_jorek_f32(int eq, const pre_field_t fields[3], const pre_field_t *Rb, const pre_out_t *out,
                           const float *K_t, const float *K_R, const float *K_Z, const float *K_RR, const float *K_ZZ,
                           const float coef[4], int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!fields || !Rb || !K_t || !K_R || !K_Z || !K_RR || !K_ZZ || !coef) return PRE_E_NULL;
    if (eq < 0 || eq > 1) return PRE_E_RANGE;
    JorekParams prm;
    if (!star_from_dense27(K_t, &prm.Dt) || !star_from_dense27(K_R, &prm.DR) || !star_from_dense27(K_Z, &prm.DZ) ||
        !star_from_dense27(K_RR, &prm.DRR) || !star_from_dense27(K_ZZ, &prm.DZZ))
        return PRE_E_UNSUPPORTED;
    prm.a0 = coef[0]; prm.a1 = coef[1]; prm.a2 = coef[2]; prm.a3 = coef[3];
    // D_RR must fit D_R's compiled tap structure and D_ZZ that of D_Z, else every operator is a general star
    int mode = pick_mode(prm.Dt, prm.DR, prm.DZ, nullptr);
    {
        const Shape rr = shape_of(prm.DRR), zz = shape_of(prm.DZZ);
        const bool rr_ok = !rr.t && !rr.y, zz_ok = mode == 0 ? (!zz.x && !zz.y) : (!zz.x && !zz.t);
        if (mode != 2 && !(rr_ok && zz_ok)) mode = 2;
    }
    Star *stars[5] = {&prm.Dt, &prm.DR, &prm.DZ, &prm.DRR, &prm.DZZ};
    Geom g;
    int rel;
    hipStream_t st = as_stream(stream);
    if (eq == 0) {
        const pre_field_t *fs[3] = {&fields[0], &fields[1], Rb};
        int rc = prepare(g, rel, fs, 3, out, B, T, X, Y, flags, stars, 5);
        if (rc) return rc;
        g.tfree = no_t_taps(stars, 5);
        return launch_mode<JorekContinuity>(relabeled_mode(mode, rel), g, prm, st);
    }
    const pre_field_t *fs[4] = {&fields[0], &fields[1], &fields[2], Rb};
    int rc = prepare(g, rel, fs, 4, out, B, T, X, Y, flags, stars, 5);
    if (rc) return rc;
    g.tfree = no_t_taps(stars, 5);
    return launch_mode<JorekTemperature>(relabeled_mode(mode, rel), g, prm, st);
}

// ---- 2-D spatial operators with boundary conditions ----------------------------
namespace {
// pre_bc_t side -> (index to read, constant); n = extent of the axis
bool bc_side(int mode, float value, int64_t n, bool hi, int *idx, float *val)
{
    *val = 0.f;
    switch (mode) {
    case PRE_BC_CONSTANT: *idx = -1; *val = value; return true;
    case PRE_BC_REPLICATE: *idx = hi ? (int)n - 1 : 0; return true;
    case PRE_BC_PERIODIC: *idx = hi ? 0 : (int)n - 1; return true;
    case PRE_BC_REFLECT: if (n < 2) return false; *idx = hi ? (int)n - 2 : 1; return true;
    default: return false;
    }
}

int fill_bc(const pre_bc_t *bc, int64_t X, int64_t Y, BCInfo *o)
{
    if (!bc) return PRE_E_NULL;
    // top/bottom act on the first spatial axis (X, rows), left/right on the second (Y, columns)
    if (!bc_side(bc->mode[2], bc->value[2], X, false, &o->xlo, &o->vxlo) || !bc_side(bc->mode[3], bc->value[3], X, true, &o->xhi, &o->vxhi) ||
        !bc_side(bc->mode[0], bc->value[0], Y, false, &o->ylo, &o->vylo) || !bc_side(bc->mode[1], bc->value[1], Y, true, &o->yhi, &o->vyhi))
        return PRE_E_RANGE;
    return PRE_OK;
}

bool star_from_dense9(const float *K, Star *s)     // 3x3 kernel, axes (X, Y)
{
    if (K[0] != 0.f || K[2] != 0.f || K[6] != 0.f || K[8] != 0.f) return false;
    *s = Star{K[4], 0.f, 0.f, K[1], K[7], K[3], K[5]};
    return true;
}
}  // namespace

int pre_spatial2d_bc_f32(const float *in, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                         const float *K, const pre_bc_t *bc, int64_t B, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!in || !out || !in_strides || !out_strides || !K) return PRE_E_NULL;
    if (flags & (PRE_FLAG_OUT_INTERIOR_T | PRE_FLAG_HALO_X)) return PRE_E_UNSUPPORTED;
    // planes [B,X,Y] -> [1,B,X,Y]: the plane axis is the tap-free marching axis
    pre_field_t f{in, 0, in_strides[0], in_strides[1], in_strides[2]};
    pre_out_t o{out, 0, out_strides[0], out_strides[1], out_strides[2]};
    if (f.sY != 1 || o.sY != 1) return PRE_E_UNSUPPORTED;            // no relabelling here: the BCs are tied to the axes
    const pre_field_t *fs[1] = {&f};
    Linear1::Params prm;
    if (!star_from_dense9(K, &prm.s)) return PRE_E_UNSUPPORTED;
    BCInfo info;
    int rc = fill_bc(bc, X, Y, &info);
    if (rc) return rc;
    Geom g;
    int rel;
    rc = prepare(g, rel, fs, 1, &o, 1, B, X, Y, flags & ~PRE_FLAG_INTERIOR_T, nullptr, 0, false, false);
    if (rc) return rc;
    g.tfree = 1;                               // (a 2-D operator: the marched axis is the batch of planes)
    return launch<Linear1, true>(g, prm, as_stream(stream), &info);
}

int pre_spatial2d_linear2_bc_f32(const float *in0, const int64_t s0[3], const float *in1, const int64_t s1[3], float *out,
                                 const int64_t out_strides[3], const float *K0, const float *K1, float ratio,
                                 const pre_bc_t *bc, int64_t B, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!in0 || !in1 || !out || !s0 || !s1 || !out_strides || !K0 || !K1) return PRE_E_NULL;
    if (flags & (PRE_FLAG_OUT_INTERIOR_T | PRE_FLAG_HALO_X)) return PRE_E_UNSUPPORTED;
    pre_field_t f0{in0, 0, s0[0], s0[1], s0[2]}, f1{in1, 0, s1[0], s1[1], s1[2]};
    pre_out_t o{out, 0, out_strides[0], out_strides[1], out_strides[2]};
    if (f0.sY != 1 || f1.sY != 1 || o.sY != 1) return PRE_E_UNSUPPORTED;
    const pre_field_t *fs[2] = {&f0, &f1};
    Linear2::Params prm;
    if (!star_from_dense9(K0, &prm.a) || !star_from_dense9(K1, &prm.b)) return PRE_E_UNSUPPORTED;
    prm.ratio = ratio;
    BCInfo info;
    int rc = fill_bc(bc, X, Y, &info);
    if (rc) return rc;
    Geom g;
    int rel;
    rc = prepare(g, rel, fs, 2, &o, 1, B, X, Y, flags & ~PRE_FLAG_INTERIOR_T, nullptr, 0, false, false);
    if (rc) return rc;
    g.tfree = 1;
    return launch<Linear2, true>(g, prm, as_stream(stream), &info);
}

}  // extern "C"
