// acc_march.hip - streaming evaluation of DENSE tap sets over three time planes (|dt| <= 1, |dx|, |dy| <= R, R = 1 or 2)
// on [B,T,X,Y] fp32 fields: a user-assigned 3x3x3 kernel, or the additive wave kernel with a Taylor-4 Laplacian
// (Utils/ConvOps_2d.py:36-62 - a 5x5 stencil that kernel_3d puts on slab 1 of a 5^3 kernel - plus D_tt).  No reference
// script builds such a set (every operator the scripts add up is a 7-point star: star_march.hip); they are reachable
// through the additive-kernel idiom (README.md:47-54) and through `D.kernel = anything`.
//
// Bound: HBM, 8 B per cell (read once, written once).  One workgroup (8 rows x 64 quads = 512 threads) owns an (x, y)
// tile of one sample and marches over the INPUT planes p:
//   * plane p is staged in LDS once (the tile's rows, R halo rows above and below, R halo columns left and right: the
//     rows by the threads that own them, the halo a float per thread), double-buffered, one LDS-only barrier per plane;
//   * a thread reads, for each row offset dx, its own quad and the R floats either side of it from LDS (the left / right
//     neighbours of a quad ARE the adjacent floats of the staged row: no shuffles, no edge cases) and adds the plane's
//     contribution to THREE accumulators: out[p+1] += w[dt=-1] * in[p], out[p] += w[0] * in[p], out[p-1] += w[+1] * in[p];
//   * after plane p the accumulator of out[p-1] is complete: stored, zeroed, and re-used for out[p+2] (the roles rotate
//     by a 3x unrolled loop, as do the register sets of the planes in flight: p+1 and p+2 are prefetched);
//   * (dt, dx) rows without a tap are skipped by wave-uniform branches (a row mask built on the host).
// Compared with the tiled kernel of stencil_generic.hip this reads every input plane once per workgroup instead of once
// per output plane (3x less LDS staging and L2 traffic) and keeps no register window (three float4 accumulators).
#include "common.h"

namespace {

constexpr int AM_NR = 8, AM_TYQ = 64, AM_NT = AM_NR * AM_TYQ;

template <int R> struct AMWeights {
    float w[3][2 * R + 1][2 * R + 1];     // [dt + 1][dx + R][dy + R]
    unsigned int rows;                    // bit (dt + 1) * (2R + 1) + dx + R: the row holds a non-zero tap
    unsigned int wide;                    // same bits: the row holds a tap beside its centre (dy != 0)
};

struct AMGeom {
    const float *in;
    long long sB, sT, sX;
    float *out;
    long long oB, oT, oX;
    int B, T, X, Y;
    int tSeg, nTSeg, nXT, nYT;
    int flags;
};

__device__ __forceinline__ void am_barrier()      // orders LDS traffic only: the global prefetches stay in flight
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int R>
__global__ void __launch_bounds__(AM_NT) acc_march_kernel(const AMGeom g, const AMWeights<R> W)
{
    constexpr int D = 2 * R + 1, ROWS = AM_NR + 2 * R, RW = 4 * AM_TYQ + 8;      // LDS row: 4 pad | 256 cells | 4 pad floats
    constexpr int NH = 2 * R * 4 * AM_TYQ, NE = ROWS * 2 * R, NX = NH + NE, KX = (NX + AM_NT - 1) / AM_NT;
    __shared__ __attribute__((aligned(16))) float lds[2][ROWS * RW];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

    const int q = threadIdx.x, ty = threadIdx.y, tid = ty * AM_TYQ + q;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int yt = L % g.nYT; L /= g.nYT;
    const int xt = L % g.nXT; L /= g.nXT;
    const int ts = L % g.nTSeg;
    const int b = L / g.nTSeg;
    const int x0 = xt * AM_NR, y0 = yt * 4 * AM_TYQ, x = x0 + ty, y = y0 + 4 * q;
    const bool inb = x < g.X && y < g.Y;                     // (Y % 4 == 0: a quad is inside or outside as a whole)
    int t0 = ts * g.tSeg, t1 = min(t0 + g.tSeg, g.T);
    if (g.flags & PRE_FLAG_INTERIOR_T) { t0 = max(t0, 1); t1 = min(t1, g.T - 1); }

    // a plane of this sample is a wave-uniform buffer descriptor based R rows before row 0; a thread's places in it are
    // 32-bit byte offsets: its own quad, and the KX halo floats it fetches (halo rows first, then the edge columns)
    const unsigned int voff = (unsigned int)(((long long)(x + R) * g.sX + y) * 4);
    const int own_lds = (ty + R) * RW + 4 + 4 * q;
    unsigned int xoff[KX];
    int xlds[KX];
    bool xok[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int idx = k * AM_NT + tid;
        int row, col, gy;                                    // LDS row, LDS column (float), grid column
        if (idx < NH) {
            const int hr = idx / (4 * AM_TYQ), c = idx % (4 * AM_TYQ);
            row = hr < R ? hr : AM_NR + hr;
            col = 4 + c;
            gy = y0 + c;
        } else {
            const int e = idx - NH, c = e % (2 * R);
            row = e / (2 * R);
            col = c < R ? 4 - R + c : 4 + 4 * AM_TYQ + (c - R);
            gy = c < R ? y0 - R + c : y0 + 4 * AM_TYQ + (c - R);
        }
        const int gx = x0 - R + row;
        xok[k] = idx < NX && gx >= 0 && gx < g.X && gy >= 0 && gy < g.Y;
        xoff[k] = (unsigned int)(((long long)(gx + R) * g.sX + gy) * 4);
        xlds[k] = idx < NX ? row * RW + col : -1;
    }
    float *outp = g.out + (long long)b * g.oB + (long long)x * g.oX + y;

    struct Plane { float4 own; float ex[KX]; };
    auto load = [&](int p, Plane &P) __attribute__((always_inline)) {
        const bool okp = p >= 0 && p < g.T;                  // (wave-uniform)
        const float *base = g.in + ((long long)b * g.sB - (long long)R * g.sX + (long long)p * g.sT);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, -1, 0x00020000);
        if (okp && inb) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
            P.own = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else {
            P.own = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < KX; ++k)
            P.ex[k] = (okp && xok[k]) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)xoff[k], 0, 0)) : 0.f;
    };

    const int p0 = t0 - 1, p1 = t1;                          // input planes t0-1 .. t1 feed outputs t0 .. t1-1
    // One input plane: `cur` holds plane p, `nxt2` receives plane p+2; aprev / acur / anext accumulate out[p-1], out[p],
    // out[p+1].  The caller rotates the roles.
    auto step = [&](int p, Plane &cur, Plane &nxt2, float4 &aprev, float4 &acur, float4 &anext) __attribute__((always_inline)) {
        float *buf = lds[(p - p0) & 1];
        *reinterpret_cast<float4 *>(buf + own_lds) = cur.own;
#pragma unroll
        for (int k = 0; k < KX; ++k)
            if (xlds[k] >= 0) buf[xlds[k]] = cur.ex[k];
        load(p + 2, nxt2);
        am_barrier();
        if (p >= 0 && p < g.T) {                             // (a plane outside the grid is zero padding: contributes nothing)
#pragma unroll
            for (int dxi = 0; dxi < D; ++dxi) {
                const unsigned int rowbits = (W.rows >> dxi) & (1u | (1u << D) | (1u << (2 * D)));
                if (!rowbits) continue;                      // (wave-uniform)
                const float *row = buf + (ty + dxi) * RW + 4 + 4 * q;
                float v[4 + 2 * R];
                const float4 o = *reinterpret_cast<const float4 *>(row);
                v[R] = o.x; v[R + 1] = o.y; v[R + 2] = o.z; v[R + 3] = o.w;
                const unsigned int widebits = (W.wide >> dxi) & (1u | (1u << D) | (1u << (2 * D)));
#pragma unroll
                for (int j = 0; j < R; ++j) v[j] = v[R + 4 + j] = 0.f;
                if (widebits) {                              // (the floats either side of the quad: only where a row needs them)
                    if constexpr (R == 1) {
                        v[0] = row[-1];
                        v[5] = row[4];
                    } else {
                        const float2 l = *reinterpret_cast<const float2 *>(row - 2), r2 = *reinterpret_cast<const float2 *>(row + 4);
                        v[0] = l.x; v[1] = l.y; v[6] = r2.x; v[7] = r2.y;
                    }
                }
                auto add = [&](float4 &a, int dti) __attribute__((always_inline)) {
#pragma unroll
                    for (int dyi = 0; dyi < D; ++dyi) {
                        const float w = W.w[dti][dxi][dyi];
                        a.x = __builtin_fmaf(w, v[dyi], a.x);
                        a.y = __builtin_fmaf(w, v[dyi + 1], a.y);
                        a.z = __builtin_fmaf(w, v[dyi + 2], a.z);
                        a.w = __builtin_fmaf(w, v[dyi + 3], a.w);
                    }
                };
                // (a row with its centre tap only - the arms of a cross-shaped 5x5 stencil, D_tt - costs one FMA per cell, not
                // 2R+1; testing every tap instead was slower: 75 scalar branches per plane break up the FMA stream)
                auto centre = [&](float4 &a, int dti) __attribute__((always_inline)) {
                    const float w = W.w[dti][dxi][R];
                    a.x = __builtin_fmaf(w, o.x, a.x); a.y = __builtin_fmaf(w, o.y, a.y);
                    a.z = __builtin_fmaf(w, o.z, a.z); a.w = __builtin_fmaf(w, o.w, a.w);
                };
                if (rowbits & 1u) { if (widebits & 1u) add(anext, 0); else centre(anext, 0); }     // w[dt = -1]: plane p is the t-1 neighbour of out[p+1]
                if (rowbits & (1u << D)) { if (widebits & (1u << D)) add(acur, 1); else centre(acur, 1); }
                if (rowbits & (1u << (2 * D))) { if (widebits & (1u << (2 * D))) add(aprev, 2); else centre(aprev, 2); }   // w[dt = +1]
            }
        }
        const int t = p - 1;                                 // out[p-1] has all three planes now
        if (t >= t0 && t < t1 && inb) {
            float4 r = aprev;
            if (g.flags & PRE_FLAG_ABS) r = make_float4(fabsf(r.x), fabsf(r.y), fabsf(r.z), fabsf(r.w));
            struct __attribute__((aligned(4))) F4u { float x, y, z, w; };
            *reinterpret_cast<F4u *>(outp + (long long)t * g.oT) = F4u{r.x, r.y, r.z, r.w};
        }
        aprev = make_float4(0.f, 0.f, 0.f, 0.f);             // becomes the accumulator of out[p+2]
    };

    Plane P0, P1, P2;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    load(p0, P0);
    load(p0 + 1, P1);
    for (int p = p0; p <= p1; p += 3) {
        step(p, P0, P2, a0, a1, a2);
        if (p + 1 > p1) break;
        step(p + 1, P1, P0, a1, a2, a0);
        if (p + 2 > p1) break;
        step(p + 2, P2, P1, a2, a0, a1);
    }
}

template <int R>
int am_launch(AMGeom &g, const AMWeights<R> &W, hipStream_t st)
{
    g.nXT = (g.X + AM_NR - 1) / AM_NR;
    g.nYT = (g.Y + 4 * AM_TYQ - 1) / (4 * AM_TYQ);
    long long tiles = (long long)g.B * g.nXT * g.nYT;
    int tSeg = g.T;
    while (tiles * ((g.T + tSeg - 1) / tSeg) < 2048 && tSeg > 16) tSeg = (tSeg + 1) / 2;
    g.tSeg = tSeg;
    g.nTSeg = (g.T + tSeg - 1) / tSeg;
    tiles *= g.nTSeg;
    if (tiles <= 0 || tiles * AM_TYQ > 0xffffffffLL) return PRE_E_SHAPE;
    hipLaunchKernelGGL((acc_march_kernel<R>), dim3((unsigned)tiles), dim3(AM_TYQ, AM_NR), 0, st, g, W);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

}  // namespace

// Internal (stencil_generic.hip): a tap list with |dt| <= 1 and |dx|, |dy| <= 2 on Y-contiguous views with Y % 4 == 0.
// w125[(dt + 1) * 25 + (dx + 2) * 5 + (dy + 2)].  PRE_E_UNSUPPORTED: the caller runs the tiled kernel.
int pre_acc_march_try(const pre_field_t *in, const pre_out_t *out, const float *w125, int radius,
                      int64_t B, int64_t T, int64_t X, int64_t Y, int flags, hipStream_t st)
{
    if (in->sY != 1 || out->sY != 1 || Y % 4 != 0 || Y < 64 || radius < 1 || radius > 2) return PRE_E_UNSUPPORTED;
    if (flags & (PRE_FLAG_OUT_INTERIOR_T | PRE_FLAG_HALO_X)) return PRE_E_UNSUPPORTED;
    if (in->sX < 0 || ((X + 12) * in->sX + Y + 16) * 4 >= (1LL << 32)) return PRE_E_UNSUPPORTED;      // 32-bit offsets in a plane
    AMGeom g;
    g.in = in->ptr; g.sB = in->sB; g.sT = in->sT; g.sX = in->sX;
    g.out = out->ptr; g.oB = out->sB; g.oT = out->sT; g.oX = out->sX;
    g.B = (int)B; g.T = (int)T; g.X = (int)X; g.Y = (int)Y;
    g.flags = flags;
    if (radius == 1) {
        AMWeights<1> W;
        W.rows = W.wide = 0u;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c)
                for (int d = 0; d < 3; ++d) {
                    const float v = w125[a * 25 + (c + 1) * 5 + (d + 1)];
                    W.w[a][c][d] = v;
                    if (v != 0.f) W.rows |= 1u << (a * 3 + c);
                    if (v != 0.f && d != 1) W.wide |= 1u << (a * 3 + c);
                }
        return am_launch<1>(g, W, st);
    }
    AMWeights<2> W;
    W.rows = W.wide = 0u;
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 5; ++c)
            for (int d = 0; d < 5; ++d) {
                const float v = w125[a * 25 + c * 5 + d];
                W.w[a][c][d] = v;
                if (v != 0.f) W.rows |= 1u << (a * 5 + c);
                if (v != 0.f && d != 2) W.wide |= 1u << (a * 5 + c);
            }
    return am_launch<2>(g, W, st);
}
