// stencil_generic.hip - C-ABI entry points for ConvOperator.convolution and the generic
// (any stride, any odd kernel extent up to 7) tap-list kernel.
//
// pre_stencil3d_f32 first offers the tap list to the streaming star kernel
// (star_march.hip); tap sets that are not on the 7-point star, views without a unit-stride
// axis, and the last (extent % 4) columns of an odd-width grid run here:
// Four forms:
//   plane_taps_kernel    tap sets on ONE input plane (Taylor-4/6 Laplacians, purely spatial kernels) on views with a
//                        unit-stride axis of >= 64 cells: a wave marches down the rows of a 256-column strip with the
//                        rows in a register window;
//   generic_tile_kernel  other tap sets on such views (axis relabelled to be the last one): 16 x 256
//                        output tiles, each needed input plane staged once in LDS, taps grouped by row;
//   flat_taps_kernel     views whose unit-stride axis is short but contiguous with the next one (the surrogate's
//                        Nt-fastest layout): the two axes merged into one row, chunks of 1024 cells staged in LDS;
//   generic_kernel       anything else (and sub-box tails): one thread per cell, neighbours via L2.
// It is the correctness floor of the library (5^3 / 7^3 Taylor kernels, padded additive kernels,
// fully strided views, odd sizes); it is not on the benchmarked path.
#include "common.h"

int pre_star_try_linear1(const pre_field_t *in, const pre_out_t *out, const float star7[7],
                         int64_t B, int64_t T, int64_t X, int64_t Y, int flags, hipStream_t st,
                         int *tail_axis, int64_t *tail_from);
int pre_acc_march_try(const pre_field_t *in, const pre_out_t *out, const float *w125, int radius,
                      int64_t B, int64_t T, int64_t X, int64_t Y, int flags, hipStream_t st);

namespace {

constexpr int MAX_TAPS = 343;   // 7*7*7

struct TapList {
    int n;
    float w[MAX_TAPS];
    int off[MAX_TAPS];          // (dt+8) | (dx+8)<<4 | (dy+8)<<8
};

// Thread map of both kernels (no 64-bit divisions): a block is 2^wsh columns wide and 256 >> wsh rows tall;
// blockIdx.x walks the (column-strip, row-group) tiles of a plane, blockIdx.y the planes, blockIdx.z the batch.
__global__ void __launch_bounds__(256) generic_kernel(const float *__restrict__ in, long long sB, long long sT,
                                                      long long sX, long long sY, float *__restrict__ out,
                                                      long long oB, long long oT, long long oX, long long oY,
                                                      int B, int T, int X, int Y, int t0, int x0, int y0,
                                                      int wsh, int nstrips, int flags, const TapList taps)
{
    // outputs: the sub-box [t0,T) x [x0,X) x [y0,Y) (all of it for t0=x0=y0=0); taps see the whole domain
    const int strip = blockIdx.x % nstrips, rowgrp = blockIdx.x / nstrips;
    const int y = y0 + (strip << wsh) + (threadIdx.x & ((1 << wsh) - 1));
    const int x = x0 + rowgrp * (256 >> wsh) + (threadIdx.x >> wsh);
    if (y >= Y || x >= X) return;
    for (int b = blockIdx.z; b < B; b += gridDim.z)
    for (int t = t0 + blockIdx.y; t < T; t += gridDim.y) {
        const float *base = in + b * sB;
        float acc = 0.f;
        for (int i = 0; i < taps.n; ++i) {
            const int o = taps.off[i];
            const int tt = t + ((o & 15) - 8), xx = x + (((o >> 4) & 15) - 8), yy = y + (((o >> 8) & 15) - 8);
            if (tt >= 0 && tt < T && xx >= 0 && xx < X && yy >= 0 && yy < Y)
                acc += taps.w[i] * base[tt * sT + xx * sX + yy * sY];
        }
        out[b * oB + t * oT + x * oX + y * oY] = (flags & PRE_FLAG_ABS) ? fabsf(acc) : acc;
    }
}

// ---- LDS-tiled form: views with a unit-stride axis (relabelled to be the last one) ------------------------
// The vector L1 does not merge requests to lines that are still in flight and is far smaller than the
// footprint of the resident waves, so every per-tap global load is an L2 request: a one-load-per-tap kernel
// runs at (L2 rate) / ntaps however the loads are shaped (measured: 0.45 ms per tap on 671 Mcells, aligned or
// not).  Here a block stages each input plane it needs ONCE in LDS (16 x 256 outputs + the reach of the tap
// set, zero-filled outside the domain = the zero padding of F.conv3d) and the taps read LDS: grouped by
// (dt, dx) row, one aligned ds_read_b128 under the thread's 4 cells plus the quads left / right of it when
// the row has dy < 0 / dy > 0 taps, shifted in registers.
constexpr int MAX_ROWS = 49;             // 7 * 7
constexpr int TILE_R = 16, TILE_C = 256;
constexpr int LDS_R = TILE_R + 6, LDS_Q = TILE_C / 4 + 2;     // +-3 rows, one quad left and right

struct __attribute__((aligned(32))) RowEnt {
    int off;                    // (dt+8) | (dx+8)<<4 | mask<<8, mask bit k: dy = k-3 present
    float w[7];                 // dy = -3..3
};
struct RowList {
    int n;
    int lo, hi;                 // reach of the tap set along the row (dx) axis
    RowEnt e[MAX_ROWS];         // sorted by (dt, dx); one 32-byte entry = one scalar load per tap row
};

struct __attribute__((aligned(4))) G4u { float x, y, z, w; };

template <bool A16>                                              // A16: every quad of the views is 16-byte aligned
__global__ void __launch_bounds__(256) generic_tile_kernel(const float *__restrict__ in, long long sB, long long sT,
                                                           long long sX, float *__restrict__ out, long long oB,
                                                           long long oT, long long oX, int B, int T, int X, int Y,
                                                           int nstrips, int flags, const RowList rows)
{
    __shared__ float4 tile[LDS_R][LDS_Q];
    const int strip = blockIdx.x % nstrips, rowgrp = blockIdx.x / nstrips;
    const int x0 = rowgrp * TILE_R, y0 = strip * TILE_C;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nload = (TILE_R + rows.hi - rows.lo) * LDS_Q;      // quads of the tile the tap set can touch
    for (int b = blockIdx.z; b < B; b += gridDim.z)
    for (int t = blockIdx.y; t < T; t += gridDim.y) {
        const float *base = in + b * sB;
        float a[4][4] = {};
        int staged = 99;                                          // dt of the plane now in LDS
        for (int i = 0; i < rows.n; ++i) {
            const RowEnt ent = rows.e[i];
            const int of = ent.off;
            const int dt = (of & 15) - 8, dx = ((of >> 4) & 15) - 8, mask = of >> 8;
            if (t + dt < 0 || t + dt >= T) continue;              // whole plane is padding
            if (dt != staged) {
                staged = dt;
                __syncthreads();                                  // readers of the previous plane are done
                const float *pl = base + (t + dt) * sT;
                for (int q = threadIdx.x; q < nload; q += 256) {
                    const int rr = q / LDS_Q + 3 + rows.lo, qc = q % LDS_Q;
                    const int gx = x0 + rr - 3, gy = y0 + 4 * qc - 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gx >= 0 && gx < X) {
                        const float *src = pl + gx * sX + gy;
                        if (gy >= 0 && gy + 3 < Y) {
                            if (A16) v = *reinterpret_cast<const float4 *>(src);
                            else {
                                const G4u u = *reinterpret_cast<const G4u *>(src);      // (four dword loads)
                                v = make_float4(u.x, u.y, u.z, u.w);
                            }
                        } else {
                            if (gy >= 0 && gy < Y) v.x = src[0];
                            if (gy + 1 >= 0 && gy + 1 < Y) v.y = src[1];
                            if (gy + 2 >= 0 && gy + 2 < Y) v.z = src[2];
                            if (gy + 3 >= 0 && gy + 3 < Y) v.w = src[3];
                        }
                    }
                    tile[rr][qc] = v;
                }
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                         // my rows: wv, wv+4, wv+8, wv+12
                const float4 *row = tile[wv + 4 * k + dx + 3];
                const float4 C = row[lane + 1];
                float4 L = make_float4(0.f, 0.f, 0.f, 0.f), R = L;
                if (mask & 0x07) L = row[lane];
                if (mask & 0x70) R = row[lane + 2];
                const float e[12] = {L.x, L.y, L.z, L.w, C.x, C.y, C.z, C.w, R.x, R.y, R.z, R.w};
#pragma unroll
                for (int d = 0; d < 7; ++d)                       // dy = d - 3, ascending: the dense kernel's tap order
                    if (mask & (1 << d)) {                        // wave-uniform
                        const float wd = ent.w[d];
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[k][j] += wd * e[1 + d + j];
                    }
            }
        }
        const int y = y0 + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + wv + 4 * k;
            if (x >= X || y >= Y) continue;
            if (flags & PRE_FLAG_ABS)
                for (int j = 0; j < 4; ++j) a[k][j] = fabsf(a[k][j]);
            float *o = out + b * oB + t * oT + x * oX + y;
            if (y + 3 < Y) {
                if (A16) *reinterpret_cast<float4 *>(o) = make_float4(a[k][0], a[k][1], a[k][2], a[k][3]);
                else *reinterpret_cast<G4u *>(o) = G4u{a[k][0], a[k][1], a[k][2], a[k][3]};
            } else for (int j = 0; j < Y - y; ++j) o[j] = a[k][j];
        }
    }
}

// ---- single-plane tap sets on a long unit-stride axis: register-window row march ------------------------------
// The Taylor-4 / Taylor-6 Laplacians (Utils/ConvOps_2d.py:36-62 through kernel_3d: every tap on kernel slab 1) and any
// purely spatial kernel read ONE input plane per output plane, so the only reuse is along the row axis.  A wave owns a
// 256-column strip of one plane and marches down its rows, a lane = 4 adjacent columns.  Each input row is loaded once
// (one 16-byte buffer load per lane; lanes 0 and 63 also fetch the three cells left / right of the strip), lives in a register
// window of 2R+1 rows plus 5-6 rows of lookahead (their loads are in flight while older rows are used) and is never
// re-read: no LDS, no barrier, no halo rows except at segment ends.  Columns left / right of a lane's quad come from
// the neighbouring lanes by DPP wavefront shifts.  Zero padding = buffer descriptors whose extent is the row (or 0 for
// rows / planes outside the domain): out-of-range lanes read 0 and their stores are dropped by the hardware.
struct PlaneTaps {
    int d0;                     // offset of the input plane
    int mask[7];                // per row offset d1 = -3..3: bit k set = tap at d2 = k-3
    float w[7][7];
};

typedef float pt_v4 __attribute__((ext_vector_type(4)));
typedef float pt_v3 __attribute__((ext_vector_type(3)));
typedef unsigned int pt_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pt_from_left(float halo, float c)     // lane i <- lane i-1; lane 0 keeps its halo value
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(halo), __float_as_int(c), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float pt_from_right(float halo, float c)    // lane i <- lane i+1; lane 63 keeps its halo value
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(halo), __float_as_int(c), 0x130, 0xf, 0xf, false));
}

// CROSS: rows d1 != 0 hold at most their centre tap (the Taylor Laplacians, any "plus"-shaped set): straight-line code,
// every weight of the footprint multiplied (zeros included, as the dense F.conv3d does).  Otherwise per-row masks.
template <int R, bool CROSS>
__global__ void __launch_bounds__(256, (R == 1 ? 5 : 4)) plane_taps_kernel(const float *__restrict__ in, long long sB, long long s0,
                                                                           long long s1, float *__restrict__ out, long long oB,
                                                                           long long o0, long long o1, int E0, int E1, int E2,
                                                                           int nstrips, int nseg, int seg, long long units,
                                                                           int flags, const PlaneTaps taps)
{
    constexpr int W = 2 * R + 1, LA = R == 3 ? 5 : 6, NS = W + LA;      // LA rows of lookahead: their loads are in flight
    const int lane = threadIdx.x & 63;
    // (readfirstlane: the wave index is uniform, and saying so keeps descriptors, row indices and branches scalar)
    long long u = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (u >= units) return;
    const int strip = (int)(u % nstrips);
    u /= nstrips;
    const int sg = (int)(u % nseg);
    u /= nseg;
    const int e0 = (int)(u % E0), b = (int)(u / E0);
    const int xs = sg * seg, xe = min(E1, xs + seg);
    const int pin = e0 + taps.d0;
    const bool plane_ok = pin >= 0 && pin < E0;
    const float *ip = in + b * sB + (plane_ok ? pin : 0) * s0;
    float *op = out + b * oB + e0 * o0;
    const unsigned int rowbytes = (unsigned)E2 * 4u;
    const unsigned int voff = (unsigned)(strip * 256 + 4 * lane) * 4u;
    // the three cells left of the strip for lane 0, right of it for lane 63 (strip 0: wraps to an out-of-range offset -> 0)
    const unsigned int hoff = lane == 0 ? voff - 12u : lane == 63 ? voff + 16u : 0xfffffff0u;

    pt_v4 C[NS];
    pt_v3 H[NS];
    auto load_row = [&](pt_v4 &c, pt_v3 &h, int xr) __attribute__((always_inline)) {
        const bool ok = plane_ok && xr >= 0 && xr < E1 && xr < xe + R;
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ip + (ok ? xr : 0) * s1), 0, ok ? rowbytes : 0u, 0x00020000);
        c = __builtin_bit_cast(pt_v4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
        h = __builtin_bit_cast(pt_v3, __builtin_amdgcn_raw_buffer_load_b96(r, hoff, 0, 0));
    };
    // rows xs-R .. xs+R+LA-1 into slots 0 .. NS-2; slot(row) = (row - (xs - R)) mod NS
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) load_row(C[k], H[k], xs - R + k);

    for (int xb = xs; xb < xe; xb += NS) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int x = xb + i;
            if (x < xe) {                                         // (wave-uniform)
                load_row(C[(i + NS - 1) % NS], H[(i + NS - 1) % NS], x + R + LA);        // into the slot of row x-R-1
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = -R; d <= R; ++d) {
                    const pt_v4 c = C[(i + d + R) % NS];
                    const int m = CROSS ? (d ? 8 : 0x7f) : taps.mask[d + 3];
                    if (m == 0) continue;
                    if (m == 8) {                                 // the centre column only
                        const float wd = taps.w[d + 3][3];
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[j] += wd * c[j];
                        continue;
                    }
                    const pt_v3 h = H[(i + d + R) % NS];
                    float e[12] = {0.f, 0.f, 0.f, 0.f, c[0], c[1], c[2], c[3], 0.f, 0.f, 0.f, 0.f};
                    if (m & 0x07) { e[1] = pt_from_left(h[0], c[1]); e[2] = pt_from_left(h[1], c[2]); e[3] = pt_from_left(h[2], c[3]); }
                    if (m & 0x70) { e[8] = pt_from_right(h[0], c[0]); e[9] = pt_from_right(h[1], c[1]); e[10] = pt_from_right(h[2], c[2]); }
#pragma unroll
                    for (int dd = 0; dd < 7; ++dd)                // d2 = dd - 3, ascending: the dense kernel's tap order
                        if (m & (1 << dd)) {
                            const float wd = taps.w[d + 3][dd];
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] += wd * e[1 + dd + j];
                        }
                }
                if (flags & PRE_FLAG_ABS)
                    for (int j = 0; j < 4; ++j) a[j] = fabsf(a[j]);
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(op + x * o1, 0, rowbytes, 0x00020000);
                const pt_v4 av = {a[0], a[1], a[2], a[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pt_u4, av), ro, voff, 0, 0);
            }
        }
    }
}

// ---- flat form: SHORT unit-stride axis whose rows follow each other in memory ---------------------------
// The surrogate's native layout [BS,F,Nx,Ny,Nt] seen through permute(0,1,4,2,3) has Nt = 10..40 as its unit-stride
// axis (Marginal/NS_Residuals_CP.py:282): a 256-column tile would keep a handful of lanes busy, and such views used
// to fall to the one-thread-per-cell kernel (~0.7 TB/s).  Memory axes (A0, A1, A2) = (slowest, stride E2, unit):
// A1 and A2 are merged into one row of L = E1*E2 cells.  A workgroup owns a chunk of 2048 (1024) merged cells of one A0
// row (two quads per thread); for every A0 offset of the tap set it stages chunk + 128 cells of halo per side in LDS once, and a tap
// (d0, d1, d2) reads the cell d1*E2 + d2 further along the merged row, masked where a1 + d1 or a2 + d2 leaves the
// domain (= the zero padding of F.conv3d).  Needs L % 4 == 0 and max|d1|*E2 + max|d2| <= 128.
constexpr int FT_H = 32;                       // halo quads per side
constexpr int FT_P = 2;                        // quads per thread: the (scalar) tap decoding is paid once for both

// FT_Q threads; a chunk is FT_P * FT_Q quads, thread t owns quads t, t + FT_Q, ...
template <int FT_Q>
__global__ void __launch_bounds__(FT_Q) flat_taps_kernel(const float *__restrict__ in, long long sB, long long s0,
                                                        float *__restrict__ out, long long oB, long long o0, int B, int E0,
                                                        int E1, int E2, int nchunks, int flags, const TapList taps)
{
    constexpr int CQ = FT_P * FT_Q;                               // quads per chunk
    __shared__ float4 lds4[CQ + 2 * FT_H];
    const float *lds = reinterpret_cast<const float *>(lds4);
    // neighbouring A0 rows are nchunks blocks apart: keep them in one XCD's L2
    const unsigned int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = blk % nchunks, r = blk / nchunks;
    const int L = E1 * E2, m0 = chunk * (4 * CQ);
    int m[FT_P];
    // validity of my cells' neighbours, per offset: bit 4*(d+3) + j of v1 (v2) says that cell j of the quad stays
    // inside the A1 (A2) extent when moved by d = -3..3
    unsigned int v1[FT_P], v2[FT_P];
#pragma unroll
    for (int p = 0; p < FT_P; ++p) {
        m[p] = m0 + 4 * ((int)threadIdx.x + p * FT_Q);
        v1[p] = v2[p] = 0u;
        int c1 = m[p] / E2, c2 = m[p] % E2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int d = -3; d <= 3; ++d) {
                v1[p] |= ((unsigned)(c1 + d) < (unsigned)E1 ? 1u : 0u) << (4 * (d + 3) + j);
                v2[p] |= ((unsigned)(c2 + d) < (unsigned)E2 ? 1u : 0u) << (4 * (d + 3) + j);
            }
            const bool wrap = c2 + 1 == E2;
            c2 = wrap ? 0 : c2 + 1;
            c1 += wrap ? 1 : 0;
        }
    }
    for (int b = blockIdx.z; b < B; b += gridDim.z) {
        float acc[FT_P][4] = {};
        int staged = 99;                                          // d0 of the row now in LDS
        for (int i = 0; i < taps.n; ++i) {                        // sorted by d0
            const int of = taps.off[i];
            const int d0 = (of & 15) - 8, d1 = ((of >> 4) & 15) - 8, d2 = ((of >> 8) & 15) - 8;
            const int rr = r + d0;
            if (rr < 0 || rr >= E0) continue;                     // the whole row is padding (block-uniform)
            const float *row = in + b * sB + rr * s0;
            const int dc = d1 * E2 + d2;
            float e[FT_P][4];
            if (of & (1 << 12)) {
                // a row that carries one or two taps only (the x arms of a Taylor Laplacian): read the shifted quads
                // straight from global memory (L2: it is the centre row of a neighbouring workgroup) instead of staging
#pragma unroll
                for (int p = 0; p < FT_P; ++p) {
                    const int g = m[p] + dc;
                    if (g >= 0 && g + 3 < L) {
                        const G4u u = *reinterpret_cast<const G4u *>(row + g);
                        e[p][0] = u.x; e[p][1] = u.y; e[p][2] = u.z; e[p][3] = u.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) e[p][j] = (g + j >= 0 && g + j < L) ? row[g + j] : 0.f;
                    }
                }
            } else {
                if (d0 != staged) {
                    staged = d0;
                    __syncthreads();                              // readers of the previous row are done
                    for (int q = threadIdx.x; q < CQ + 2 * FT_H; q += FT_Q) {
                        const int g = m0 - 4 * FT_H + 4 * q;      // L % 4 == 0: a quad is all inside or all outside
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (g >= 0 && g < L) {
                            const G4u u = *reinterpret_cast<const G4u *>(row + g);
                            v = make_float4(u.x, u.y, u.z, u.w);
                        }
                        lds4[q] = v;
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int p = 0; p < FT_P; ++p) {
                    const float *src = lds + 4 * FT_H + (m[p] - m0) + dc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[p][j] = src[j];
                }
            }
            const float w = taps.w[i];
#pragma unroll
            for (int p = 0; p < FT_P; ++p) {
                const unsigned int ok = (v1[p] >> (4 * (d1 + 3))) & (v2[p] >> (4 * (d2 + 3)));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] += (ok >> j) & 1u ? w * e[p][j] : 0.f;
            }
        }
#pragma unroll
        for (int p = 0; p < FT_P; ++p)
            if (m[p] < L) {
                if (flags & PRE_FLAG_ABS)
                    for (int j = 0; j < 4; ++j) acc[p][j] = fabsf(acc[p][j]);
                *reinterpret_cast<G4u *>(out + b * oB + r * o0 + m[p]) = G4u{acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
            }
    }
}

// ---- kernel gradient of the zero-padded cross-correlation (autograd: d loss / d kernel) -------------------
// gk[dt][dx][dy] += sum over cells of g[c] * x[c + (dt,dx,dy)] for a (kt,kx,ky) kernel with extents in {1,3}:
// one pass over g, the three x planes staged in LDS as above, 27 accumulators per thread, block reduction,
// one double atomic per tap and block.  (torch composes this as kt*kx*ky sliced multiply-and-sum passes.)
__global__ void __launch_bounds__(256) wgrad27_kernel(const float *__restrict__ xin, long long sB, long long sT, long long sX,
                                                      const float *__restrict__ gin, long long gB, long long gT, long long gX,
                                                      int B, int T, int X, int Y, int rt, int rx, int ry, int nstrips,
                                                      double *__restrict__ gk)
{
    __shared__ float4 tile[TILE_R + 2][LDS_Q];
    __shared__ float red[4][27];
    const int strip = blockIdx.x % nstrips, rowgrp = blockIdx.x / nstrips;
    const int x0 = rowgrp * TILE_R, y0 = strip * TILE_C;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    for (int b = blockIdx.z; b < B; b += gridDim.z)
    for (int t = blockIdx.y; t < T; t += gridDim.y) {
        // my 4 rows x 4 cells of the upstream gradient (zero outside the domain)
        float gq[4][4];
        const int y = y0 + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + wv + 4 * k;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gq[k][j] = (x < X && y + j < Y) ? gin[b * gB + t * gT + x * gX + y + j] : 0.f;
        }
        for (int dt = -rt; dt <= rt; ++dt) {
            __syncthreads();
            const int tt = t + dt;
            for (int q = threadIdx.x; q < (TILE_R + 2) * LDS_Q; q += 256) {
                const int rr = q / LDS_Q, qc = q % LDS_Q;
                const int gx = x0 + rr - 1, gy = y0 + 4 * qc - 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tt >= 0 && tt < T && gx >= 0 && gx < X) {
                    const float *src = xin + b * sB + tt * sT + gx * sX + gy;
                    if (gy >= 0 && gy < Y) v.x = src[0];
                    if (gy + 1 >= 0 && gy + 1 < Y) v.y = src[1];
                    if (gy + 2 >= 0 && gy + 2 < Y) v.z = src[2];
                    if (gy + 3 >= 0 && gy + 3 < Y) v.w = src[3];
                }
                tile[rr][qc] = v;
            }
            __syncthreads();
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int dx = dxi - 1;
                if (dx < -rx || dx > rx) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 *row = tile[wv + 4 * k + dx + 1];
                    const float4 L = row[lane], C = row[lane + 1], R = row[lane + 2];
                    const float e[12] = {L.x, L.y, L.z, L.w, C.x, C.y, C.z, C.w, R.x, R.y, R.z, R.w};
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi) {
                        float sum = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) sum += gq[k][j] * e[4 + j + dyi - 1];
                        // the t index of the accumulator is not a compile-time constant: add to all three, masked
#pragma unroll
                        for (int dti = 0; dti < 3; ++dti)
                            acc[(dti * 3 + dxi) * 3 + dyi] += (dti - 1 == dt) ? sum : 0.f;
                    }
                }
            }
        }
    }
    // block reduction, then one double atomic per tap
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        const int k = threadIdx.x, dt = k / 9 - 1, dx = (k / 3) % 3 - 1, dy = k % 3 - 1;
        if (dt >= -rt && dt <= rt && dx >= -rx && dx <= rx && dy >= -ry && dy <= ry) {
            const double v = (double)red[0][k] + red[1][k] + red[2][k] + red[3][k];
            atomicAdd(&gk[((dt + rt) * (2 * rx + 1) + (dx + rx)) * (2 * ry + 1) + (dy + ry)], v);
        }
    }
}

// periodic-wall mismatch (Marginal/NS_Residuals_CP.py:468-478): out[b,t,i] = (u[b,t,P(i)] - u[b,t,Q(i)]) * dx with P, Q
// two opposite edges of the [X,Y] plane.  O(B T L) work: one thread per output, edge offsets precomputed by the host.
__global__ void __launch_bounds__(256) edge_residual_kernel(const float *__restrict__ u, long long sB, long long sT, long long sI,
                                                            long long offP, long long offQ, long long T, long long L,
                                                            long long total, float dx, float *__restrict__ out)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const long long i = g % L, bt = g / L, t = bt % T, b = bt / T;
    const float *row = u + b * sB + t * sT + i * sI;
    out[g] = (row[offP] - row[offQ]) * dx;
}

}  // namespace

extern "C" {

int pre_abi_version(void) { return 8; }

int pre_edge_residual_f32(const pre_field_t *u, int wall, float dx, int64_t B, int64_t T, int64_t X, int64_t Y, float *out,
                          void *stream)
{
    if (!u || !u->ptr || !out) return PRE_E_NULL;
    if (B < 0 || T < 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (wall < 0 || wall > 3) return PRE_E_RANGE;
    // top: u[0,:] - u[X-1,:]   bottom: u[X-1,:] - u[0,:]   (rows; L = Y)     left: u[:,0] - u[:,Y-1]   right: the opposite (L = X)
    const bool rows = wall < 2;
    const long long L = rows ? Y : X, sI = rows ? u->sY : u->sX, far = rows ? (X - 1) * u->sX : (Y - 1) * u->sY;
    const long long offP = (wall == 0 || wall == 2) ? 0 : far, offQ = far - offP;
    const long long total = (long long)B * T * L;
    if (total == 0) return PRE_OK;
    if ((total + 255) / 256 > 0x7fffffffLL) return PRE_E_SHAPE;
    hipLaunchKernelGGL(edge_residual_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), u->ptr,
                       (long long)u->sB, (long long)u->sT, sI, offP, offQ, (long long)T, L, total, dx, out);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_stencil3d_f32(const pre_field_t *in, const pre_out_t *out, const float *tap_w, const int32_t *tap_off, int ntaps,
                      int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!in || !in->ptr || !out || !out->ptr || (ntaps > 0 && (!tap_w || !tap_off))) return PRE_E_NULL;
    if (B <= 0 || T <= 0 || X <= 0 || Y <= 0 || ntaps < 0) return PRE_E_NULL;
    if (ntaps > MAX_TAPS) return PRE_E_SHAPE;
    if (flags & PRE_FLAG_OUT_INTERIOR_T) return PRE_E_UNSUPPORTED;      // fused residual entries only
    if (B > 0x7fffffff || T > 0x7fffffff || X > 0x7fffffff || Y > 0x7fffffff) return PRE_E_SHAPE;
    hipStream_t st = as_stream(stream);

    // star-shaped?  fold duplicates, then try the streaming kernel
    bool star = true;
    float s7[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ntaps; ++i) {
        const int dt = tap_off[3 * i], dx = tap_off[3 * i + 1], dy = tap_off[3 * i + 2];
        if (dt < -3 || dt > 3 || dx < -3 || dx > 3 || dy < -3 || dy > 3) return PRE_E_SHAPE;
        const int nz = (dt != 0) + (dx != 0) + (dy != 0);
        if (nz > 1 || dt < -1 || dt > 1 || dx < -1 || dx > 1 || dy < -1 || dy > 1) { star = false; continue; }
        const int slot = dt ? (dt < 0 ? 1 : 2) : dx ? (dx < 0 ? 3 : 4) : dy ? (dy < 0 ? 5 : 6) : 0;
        s7[slot] += tap_w[i];
    }
    int box[3] = {0, 0, 0};                       // first (t, x, y) the generic kernel has to compute
    if (star) {
        int tail_axis = -1;
        int64_t tail_from = 0;
        int rc = pre_star_try_linear1(in, out, s7, B, T, X, Y, flags, st, &tail_axis, &tail_from);
        if (rc != PRE_E_UNSUPPORTED) {
            if (rc != PRE_OK || tail_axis < 0) return rc;
            box[tail_axis] = (int)tail_from;       // streaming kernel done; <= 3 leftover columns follow
        }
    }
    if (flags & PRE_FLAG_HALO_X) return PRE_E_UNSUPPORTED;      // only the streaming star kernel reads the halo rows

    // whole domain, input and output share a unit-stride axis long enough to fill a tile row: relabel it to be
    // the last one (the tap list is symmetric in the axes) and group the taps by (dt, dx) row
    const int64_t D[3] = {T, X, Y}, si[3] = {in->sT, in->sX, in->sY}, so[3] = {out->sT, out->sX, out->sY};
    int unit = -1;
    for (int a = 2; a >= 0 && unit < 0; --a)
        if (si[a] == 1 && so[a] == 1 && D[a] >= 64) unit = a;
    if (unit >= 0 && !box[0] && !box[1] && !box[2] && ntaps > 0) {
        const int p[3] = {unit == 0 ? 2 : 0, unit == 1 ? 2 : 1, unit};     // swap `unit` with the last axis
        const long long nstrips = (D[p[2]] + TILE_C - 1) / TILE_C;
        const long long tiles = nstrips * ((D[p[1]] + TILE_R - 1) / TILE_R);
        if (tiles < (1 << 23)) {
            float w[7][7][7] = {};
            int mask[7][7] = {};
            for (int i = 0; i < ntaps; ++i) {
                const int o0 = tap_off[3 * i + p[0]] + 3, o1 = tap_off[3 * i + p[1]] + 3, o2 = tap_off[3 * i + p[2]] + 3;
                w[o0][o1][o2] += tap_w[i];
                mask[o0][o1] |= 1 << o2;
            }
            RowList rows;
            rows.n = rows.lo = rows.hi = 0;
            for (int o0 = 0; o0 < 7; ++o0)
                for (int o1 = 0; o1 < 7; ++o1) {
                    if (!mask[o0][o1]) continue;
                    rows.e[rows.n].off = (o0 - 3 + 8) | ((o1 - 3 + 8) << 4) | (mask[o0][o1] << 8);
                    for (int d = 0; d < 7; ++d) rows.e[rows.n].w[d] = w[o0][o1][d];
                    rows.lo = o1 - 3 < rows.lo ? o1 - 3 : rows.lo;
                    rows.hi = o1 - 3 > rows.hi ? o1 - 3 : rows.hi;
                    ++rows.n;
                }
            // every tap on one input plane, quads aligned: the register-window row march
            int d0only = 99, r1 = 0;
            for (int i = 0; i < ntaps; ++i) {
                const int o0 = tap_off[3 * i + p[0]], o1 = tap_off[3 * i + p[1]];
                d0only = d0only == 99 || d0only == o0 ? o0 : 100;
                r1 = (o1 < 0 ? -o1 : o1) > r1 ? (o1 < 0 ? -o1 : o1) : r1;
            }
            const long long E0 = D[p[0]], E1 = D[p[1]], E2 = D[p[2]];
            const bool aligned = E2 % 4 == 0 && !((uintptr_t)in->ptr & 15) && !((uintptr_t)out->ptr & 15) &&
                                 !((in->sB | si[p[0]] | si[p[1]] | out->sB | so[p[0]] | so[p[1]]) & 3);
            if (d0only < 99 && aligned && E2 <= (1 << 28)) {
                PlaneTaps pt;
                pt.d0 = d0only;
                for (int o1 = 0; o1 < 7; ++o1) {
                    pt.mask[o1] = mask[d0only + 3][o1];
                    for (int d = 0; d < 7; ++d) pt.w[o1][d] = w[d0only + 3][o1][d];
                }
                const long long pstrips = (E2 + 255) / 256, units0 = B * E0 * pstrips;
                long long nseg = (8192 + units0 - 1) / units0;                 // enough waves to fill the chip ...
                if (nseg > E1 / 32) nseg = E1 / 32;                            // ... from segments of at least 32 rows
                if (nseg < 1) nseg = 1;
                const long long seg = (E1 + nseg - 1) / nseg;
                nseg = (E1 + seg - 1) / seg;
                const long long units = units0 * nseg, blocks = (units + 3) / 4;
                if (blocks <= 0x7fffffffLL) {
#define PRE_PT_LAUNCH(RR, CR)                                                                                                   \
    hipLaunchKernelGGL((plane_taps_kernel<RR, CR>), dim3((unsigned)blocks), dim3(256), 0, st, in->ptr, (long long)in->sB,          \
                       (long long)si[p[0]], (long long)si[p[1]], out->ptr, (long long)out->sB, (long long)so[p[0]],               \
                       (long long)so[p[1]], (int)E0, (int)E1, (int)E2, (int)pstrips, (int)nseg, (int)seg, units, flags, pt)
                    bool cross = true;
                    for (int o1 = 0; o1 < 7; ++o1) cross = cross && (o1 == 3 || !(pt.mask[o1] & ~8));
                    if (r1 <= 1) { if (cross) PRE_PT_LAUNCH(1, true); else PRE_PT_LAUNCH(1, false); }
                    else if (r1 == 2) { if (cross) PRE_PT_LAUNCH(2, true); else PRE_PT_LAUNCH(2, false); }
                    else { if (cross) PRE_PT_LAUNCH(3, true); else PRE_PT_LAUNCH(3, false); }
#undef PRE_PT_LAUNCH
                    PRE_LAUNCH_CHECK();
                    return PRE_OK;
                }
            }
            // taps on three adjacent time planes within two cells of the centre, reference layout: the accumulator march
            // (acc_march.hip; a dense 3^3 kernel, the wave kernel with a Taylor-4 Laplacian)
            if (unit == 2) {
                bool fits = true;
                int rad = 1;
                float w125[125] = {};
                for (int i = 0; i < ntaps; ++i) {
                    const int dt = tap_off[3 * i], dx = tap_off[3 * i + 1], dy = tap_off[3 * i + 2];
                    if (dt < -1 || dt > 1 || dx < -2 || dx > 2 || dy < -2 || dy > 2) { fits = false; break; }
                    if (dx < -1 || dx > 1 || dy < -1 || dy > 1) rad = 2;
                    w125[(dt + 1) * 25 + (dx + 2) * 5 + (dy + 2)] += tap_w[i];
                }
                if (fits) {
                    const int rc = pre_acc_march_try(in, out, w125, rad, B, T, X, Y, flags, st);
                    if (rc != PRE_E_UNSUPPORTED) return rc;
                }
            }
            const dim3 grid((unsigned)tiles, (unsigned)(D[p[0]] < 65535 ? D[p[0]] : 65535), (unsigned)(B < 65535 ? B : 65535));
            // (`aligned` without the width condition: a ragged last quad is handled element-wise either way)
            const bool a16 = !((uintptr_t)in->ptr & 15) && !((uintptr_t)out->ptr & 15) &&
                             !((in->sB | si[p[0]] | si[p[1]] | out->sB | so[p[0]] | so[p[1]]) & 3);
            if (a16)
                hipLaunchKernelGGL(generic_tile_kernel<true>, grid, dim3(256), 0, st, in->ptr, (long long)in->sB, (long long)si[p[0]],
                                   (long long)si[p[1]], out->ptr, (long long)out->sB, (long long)so[p[0]], (long long)so[p[1]],
                                   (int)B, (int)D[p[0]], (int)D[p[1]], (int)D[p[2]], (int)nstrips, flags, rows);
            else
                hipLaunchKernelGGL(generic_tile_kernel<false>, grid, dim3(256), 0, st, in->ptr, (long long)in->sB, (long long)si[p[0]],
                                   (long long)si[p[1]], out->ptr, (long long)out->sB, (long long)so[p[0]], (long long)so[p[1]],
                                   (int)B, (int)D[p[0]], (int)D[p[1]], (int)D[p[2]], (int)nstrips, flags, rows);
            PRE_LAUNCH_CHECK();
            return PRE_OK;
        }
    }
    // short unit-stride axis A2 (< 64 cells: the tile kernel above would idle) whose rows follow each other along
    // another axis A1 (stride == extent of A2) in the input and in the output: flat form
    if (!box[0] && !box[1] && !box[2] && ntaps > 0) {
        for (int a2 = 2; a2 >= 0; --a2) {
            if (si[a2] != 1 || so[a2] != 1 || D[a2] >= 64 || D[a2] < 2) continue;
            for (int a1 = 2; a1 >= 0; --a1) {
                if (a1 == a2 || si[a1] != D[a2] || so[a1] != D[a2] || D[a1] < 2) continue;
                const int a0 = 3 - a1 - a2;
                const long long L = D[a1] * D[a2];
                int reach = 0, r1 = 0, r2 = 0;
                for (int i = 0; i < ntaps; ++i) {
                    const int d1 = tap_off[3 * i + a1], d2 = tap_off[3 * i + a2];
                    r1 = (d1 < 0 ? -d1 : d1) > r1 ? (d1 < 0 ? -d1 : d1) : r1;
                    r2 = (d2 < 0 ? -d2 : d2) > r2 ? (d2 < 0 ? -d2 : d2) : r2;
                }
                reach = r1 * (int)D[a2] + r2;
                // chunks of 2048 cells (256 threads) unless 1024-cell ones (128 threads) waste fewer lanes on the last
                // chunk of a row (Nt = 10, Ny = 256: 2560 cells = 1.25 big chunks or 2.5 small ones)
                const long long q = L / 4;
                const bool small = ((q + 128 * FT_P - 1) / (128 * FT_P)) * 128 < ((q + 256 * FT_P - 1) / (256 * FT_P)) * 256;
                const int ftq = (small ? 128 : 256) * FT_P;
                const long long nchunks = (q + ftq - 1) / ftq;
                if (L % 4 != 0 || L > 0x7fffffffLL || reach > 4 * FT_H || nchunks * D[a0] > 0x7fffffffLL) continue;
                TapList ft;                                        // taps on the memory axes, sorted by d0 (stable)
                ft.n = 0;
                for (int d0 = -3; d0 <= 3; ++d0) {
                    int cnt = 0;
                    for (int i = 0; i < ntaps; ++i) cnt += tap_off[3 * i + a0] == d0;
                    const int direct = cnt <= 2 ? 1 << 12 : 0;    // rows with one or two taps are read without staging
                    for (int i = 0; i < ntaps; ++i)
                        if (tap_off[3 * i + a0] == d0) {
                            ft.w[ft.n] = tap_w[i];
                            ft.off[ft.n++] = (d0 + 8) | ((tap_off[3 * i + a1] + 8) << 4) | ((tap_off[3 * i + a2] + 8) << 8) | direct;
                        }
                }
                // a block walks the batch axis with stride gridDim.z: enough blocks to fill the chip (~16 k), few enough
                // that the per-thread set-up (validity masks) is paid once for many samples
                long long gz = (16384 + nchunks * D[a0] - 1) / (nchunks * D[a0]);
                gz = gz < 1 ? 1 : (gz > B ? B : gz);
                const dim3 grid((unsigned)(nchunks * D[a0]), 1u, (unsigned)(gz < 65535 ? gz : 65535));
                if (small)
                    hipLaunchKernelGGL(flat_taps_kernel<128>, grid, dim3(128), 0, st, in->ptr, (long long)in->sB, (long long)si[a0],
                                       out->ptr, (long long)out->sB, (long long)so[a0], (int)B, (int)D[a0], (int)D[a1],
                                       (int)D[a2], (int)nchunks, flags, ft);
                else
                    hipLaunchKernelGGL(flat_taps_kernel<256>, grid, dim3(256), 0, st, in->ptr, (long long)in->sB, (long long)si[a0],
                                       out->ptr, (long long)out->sB, (long long)so[a0], (int)B, (int)D[a0], (int)D[a1],
                                       (int)D[a2], (int)nchunks, flags, ft);
                PRE_LAUNCH_CHECK();
                return PRE_OK;
            }
        }
    }
    TapList taps;
    taps.n = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        taps.w[i] = tap_w[i];
        taps.off[i] = (tap_off[3 * i] + 8) | ((tap_off[3 * i + 1] + 8) << 4) | ((tap_off[3 * i + 2] + 8) << 8);
    }
    const long long nT = T - box[0], nX = X - box[1], nY = Y - box[2];
    int wsh = 0;
    while (wsh < 6 && (1 << wsh) < nY) ++wsh;
    const long long nstrips = (nY + (1 << wsh) - 1) >> wsh, rows = 256 >> wsh;
    const long long tiles = nstrips * ((nX + rows - 1) / rows);
    if (tiles >= (1 << 23)) return PRE_E_SHAPE;              // a plane of > 2^31 cells
    const dim3 grid((unsigned)tiles, (unsigned)(nT < 65535 ? nT : 65535), (unsigned)(B < 65535 ? B : 65535));
    hipLaunchKernelGGL(generic_kernel, grid, dim3(256), 0, st, in->ptr, (long long)in->sB,
                       (long long)in->sT, (long long)in->sX, (long long)in->sY, out->ptr, (long long)out->sB,
                       (long long)out->sT, (long long)out->sX, (long long)out->sY, (int)B, (int)T, (int)X, (int)Y, box[0], box[1],
                       box[2], wsh, (int)nstrips, flags, taps);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_stencil3d_wgrad_f32(const pre_field_t *x, const pre_field_t *g, int kt, int kx, int ky, int64_t B, int64_t T, int64_t X,
                            int64_t Y, double *gk, void *stream)
{
    if (!x || !x->ptr || !g || !g->ptr || !gk) return PRE_E_NULL;
    if (B <= 0 || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if ((kt != 1 && kt != 3) || (kx != 1 && kx != 3) || (ky != 1 && ky != 3)) return PRE_E_UNSUPPORTED;
    if (x->sY != 1 || g->sY != 1) return PRE_E_UNSUPPORTED;
    if (B > 0x7fffffff || T > 0x7fffffff || X > 0x7fffffff || Y > 0x7fffffff) return PRE_E_SHAPE;
    const long long nstrips = (Y + TILE_C - 1) / TILE_C, tiles = nstrips * ((X + TILE_R - 1) / TILE_R);
    if (tiles >= (1 << 23)) return PRE_E_SHAPE;
    // few, fat blocks along (t, batch): every block ends with 27 atomics
    const dim3 grid((unsigned)tiles, (unsigned)(T < 64 ? T : 64), (unsigned)(B < 64 ? B : 64));
    hipLaunchKernelGGL(wgrad27_kernel, grid, dim3(256), 0, as_stream(stream), x->ptr, (long long)x->sB, (long long)x->sT,
                       (long long)x->sX, g->ptr, (long long)g->sB, (long long)g->sT, (long long)g->sX, (int)B, (int)T, (int)X,
                       (int)Y, kt / 2, kx / 2, ky / 2, (int)nstrips, gk);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_stencil2d_f32(const float *in, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                      const float *tap_w, const int32_t *tap_off, int ntaps, int64_t B, int64_t T, int64_t X, int flags,
                      void *stream)
{
    if (!in || !in_strides || !out || !out_strides || (ntaps > 0 && !tap_off)) return PRE_E_NULL;
    if (ntaps < 0 || ntaps > MAX_TAPS) return PRE_E_SHAPE;
    if (flags & PRE_FLAG_HALO_X) return PRE_E_UNSUPPORTED;      // (the kernel's row axis is the caller's Nt here)
    // [B,T,X] with taps (dt,dx)  ==  [1,B,T,X] with taps (0,dt,dx): the batch axis becomes the
    // (tap-free) marching axis, Nt the row axis and Nx the contiguous axis.
    int32_t off3[3 * MAX_TAPS];
    for (int i = 0; i < ntaps; ++i) {
        off3[3 * i] = 0;
        off3[3 * i + 1] = tap_off[2 * i];
        off3[3 * i + 2] = tap_off[2 * i + 1];
    }
    pre_field_t f{in, 0, in_strides[0], in_strides[1], in_strides[2]};
    pre_out_t o{out, 0, out_strides[0], out_strides[1], out_strides[2]};
    return pre_stencil3d_f32(&f, &o, tap_w, off3, ntaps, 1, B, T, X, flags, stream);
}

}  // extern "C"
