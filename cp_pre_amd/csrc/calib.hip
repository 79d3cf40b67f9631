// calib.hip - nonconformity scores, modulation, joint score, the scalar radix-select quantile and
// coverage for gfx950 (the per-cell select over the batch axis lives in kth_axis0.hip).  All kernels are HBM-bound streaming passes (4 B per element per
// pass); there is no contraction here and nothing is reshaped into one.
//
// Layout convention: a calibration tensor is contiguous [n, M] - n samples (batch axis,
// slowest) by M cells.  "lane = cell": consecutive lanes read consecutive cells of one
// sample row (coalesced), and a lane walks the batch axis for its own cell, so per-cell
// state (running sums, radix prefixes) lives in registers and never crosses lanes.
#include "common.h"

// The reductions below reproduce numpy's operation order: a*b+c must stay two roundings.
// This file is compiled with -ffp-contract=off (csrc/Makefile); the pragma covers other builds.
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------ |a-b|
// |a - b| (or |a|): one SHORT block per 1024 quads - 16 KB of each array, contiguous, no loop - dispatched in order, instead
// of a capped grid whose threads stride through the arrays: round 6, interleaved on one box (profiles/r06/absdiff_ab.txt):
// grid-stride 4.92 TB/s (unrolling it four-fold: +0.5 %; a non-power-of-two stride: +4 %), one long contiguous run per block
// 5.22, short blocks 5.62 - the dispatcher keeps the chip's accesses in one compact window of each array.
__global__ void __launch_bounds__(256) absdiff_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      float *__restrict__ out, long long n)
{
    const bool vec = !(((uintptr_t)a | (uintptr_t)out | (uintptr_t)(b ? b : a)) & 15);
    long long done = 0;
    if (vec) {
        const long long n4 = n / 4;
        const float4 *a4 = reinterpret_cast<const float4 *>(a);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        float4 *o4 = reinterpret_cast<float4 *>(out);
        for (long long blk = blockIdx.x; blk * 1024 < n4; blk += gridDim.x) {      // (one trip unless the grid was capped)
            const long long i = blk * 1024 + threadIdx.x;
            float4 v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 256 < n4) v[u] = a4[i + u * 256];
            if (b) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u * 256 < n4) {
                        w[u] = b4[i + u * 256];
                        v[u] = make_float4(v[u].x - w[u].x, v[u].y - w[u].y, v[u].z - w[u].z, v[u].w - w[u].w);
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 256 < n4) o4[i + u * 256] = make_float4(fabsf(v[u].x), fabsf(v[u].y), fabsf(v[u].z), fabsf(v[u].w));
        }
        done = n4 * 4;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = done + tid; i < n; i += stride) out[i] = fabsf(b ? a[i] - b[i] : a[i]);
}

// ------------------------------------------------------------------ std over axis 0, numpy order
// numpy's np.std(x, axis=0) on float32 [n, M]: s = x[0]+x[1]+... (sequential fp32), mean = s/n,
// d = x-mean, d = d*d, v = d[0]+d[1]+... (sequential fp32), sqrt(v/n).  Reproduced op for op
// (no fma contraction), so the result is bit-identical to the numpy oracle.
__global__ void __launch_bounds__(256) std_axis0_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        int n, long long M, float eps, float *__restrict__ mod)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    const float *pa = a + c, *pb = b ? b + c : nullptr;
    float s = 0.f;
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
        const float v = pb ? pa[(long long)i * M] - pb[(long long)i * M] : pa[(long long)i * M];
        s = s + v;
    }
    const float mean = s / (float)n;
    float acc = 0.f;
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
        const float v = pb ? pa[(long long)i * M] - pb[(long long)i * M] : pa[(long long)i * M];
        const float d = v - mean;
        const float dd = d * d;
        acc = acc + dd;
    }
    mod[c] = sqrtf(acc / (float)n) + eps;
}

// ------------------------------------------------------------------ streaming moments (fp64)
__global__ void __launch_bounds__(256) moments_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      int n, long long M, long long ld, int rows_per_split,
                                                      double *__restrict__ sum, double *__restrict__ sumsq)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    const int i0 = blockIdx.y * rows_per_split;
    const int i1 = min(n, i0 + rows_per_split);
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) {
        float v = a[(long long)i * ld + c];
        if (b) v = v - b[(long long)i * ld + c];
        const double d = (double)v;
        s += d;
        q += d * d;
    }
    if (gridDim.y == 1) { sum[c] += s; sumsq[c] += q; }
    else { atomicAdd(&sum[c], s); atomicAdd(&sumsq[c], q); }
}

// Moments of a residual tensor [n, T, X, Y] (Y % 64 == 0) AND, from the same read, the bounds of the pruned joint
// score: segmax[i][tc][x][y / 64] = bit pattern of max |a| of sample i over the (up to) 16 planes of chunk tc and the
// 64 columns of the segment (cells within cx / cy of the x / y rim excluded, 0 for rim rows).  A thread owns one
// (x, y) column of a chunk and keeps its 2 x 16 fp64 sums in registers; a wave is one segment, so the per-sample
// maximum is a wave reduction and one store.  Per cell and split the samples are added in ascending order, as
// moments_kernel does (the split counts differ, so the fp64 sums agree up to the order of the additions).
constexpr int MS_TMAX = 16;
#ifndef MS_WANT1
#define MS_WANT1 1024           // blocks the many-plane moments pass wants at least (more = the batch axis split finer)
#endif
#ifndef MS_STAGE
#define MS_STAGE 1
#endif
// TCH planes x US samples in flight per thread (16 loads either way): <16,1> for slabs of many planes, <4,4> and
// <1,16> when the tensor has only a few (C5 arrives as [n,1,Nt,Nx]).  A segment always spans MS_TMAX planes:
// chunk tc = blockIdx.z covers planes [tc*MS_TMAX, ...) and for TCH < MS_TMAX there is one chunk.
// max over the wave of non-negative patterns, valid in lane 63: row_shr 1,2,4,8 (invalid source lanes read 0), then
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
__device__ __forceinline__ unsigned int wave_max_u32_to_lane63(unsigned int v)
{
#define PRE_DPP_MAX(ctrl, rows)                                                                                                 \
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, true))
    PRE_DPP_MAX(0x111, 0xf);
    PRE_DPP_MAX(0x112, 0xf);
    PRE_DPP_MAX(0x114, 0xf);
    PRE_DPP_MAX(0x118, 0xf);
    PRE_DPP_MAX(0x142, 0xa);
    PRE_DPP_MAX(0x143, 0xc);
#undef PRE_DPP_MAX
    return v;
}

// the value lane (l ^ O) holds, O in {1, 2, 4, 8}, by DPP (VALU) instead of ds_bpermute: the wave reductions of the
// few-plane moments pass were bound by the LDS pipe's crossbar (one bpermute per element and wave: moments + maxima 6.1 ms
// on C5's 26.8 GB against 4.8 for the plain moments)
template <int O>
__device__ __forceinline__ unsigned int dpp_xor(unsigned int x)
{
    if constexpr (O == 1) return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (O == 8) return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);  // row_ror:8
    else {
        // O == 4: banks 0, 2 of a row (lanes with bit 2 clear) read 4 lanes up, banks 1, 3 read 4 lanes down
        const int t = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);                                 // row_shl:4
        return (unsigned int)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);                         // row_shr:4
    }
}

// Wave maxima of US samples at once.  Step `width`: lanes with that bit clear keep samples [0, H), the others [H, 2H), and
// each takes its partner's maxima of the samples it keeps.  Widths 32 and 16 are gfx950's v_permlane32_swap /
// v_permlane16_swap: ONE swap of (m[k], m[H+k]) puts, in the lanes that keep sample k, the partner's m[k] next to their
// own, and likewise for H+k in the other lanes - one swap and one max per pair of samples; widths 8 and 4 are DPP row moves.
typedef unsigned int ms_u32x2 __attribute__((ext_vector_type(2)));
template <int US, int H>
__device__ __forceinline__ void ms_butterfly(unsigned int (&m)[US], int lane)
{
    if constexpr (H >= 1) {
        constexpr int width = 64 * H / US;                        // 32 for the first step
#pragma unroll
        for (int k = 0; k < H; ++k) {
            if constexpr (width == 32) {
                const ms_u32x2 r = __builtin_amdgcn_permlane32_swap(m[k], m[H + k], false, false);
                m[k] = max(r.x, r.y);
            } else if constexpr (width == 16) {
                const ms_u32x2 r = __builtin_amdgcn_permlane16_swap(m[k], m[H + k], false, false);
                m[k] = max(r.x, r.y);
            } else {
                const bool up = lane & width;
                const unsigned int mine = up ? m[H + k] : m[k], send = up ? m[k] : m[H + k];
                m[k] = max(mine, dpp_xor<width>(send));
            }
        }
        ms_butterfly<US, H / 2>(m, lane);
    }
}

// US samples starting at sample i of one thread's column: ms_load issues the loads, ms_use adds them to the sums and
// writes the per-sample wave maxima.
// one plane of one sample at the wave-uniform address `p` (a buffer descriptor assembled by scalar instructions), this lane's
// cell at byte offset `cb`: no vector address arithmetic (the compiler otherwise spends a 64-bit vector add on every load)
__device__ __forceinline__ float ms_cell(const float *p, unsigned int cb, unsigned int plane_bytes)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, plane_bytes, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)cb, 0, 0));
}

// (`au`: wave-uniform pointer to plane t0 of sample 0, `c`: BYTE offset of this lane's cell - the loads then take a scalar base
// and a 32-bit lane offset, with no 64-bit vector address arithmetic per load.  FULL: the chunk has all TCH planes - no selects.)
template <int TCH, int US, bool FULL>
__device__ __forceinline__ void ms_load(const float *__restrict__ au, unsigned int c, long long row_stride, long long plane, int i, int nt,
                                        float (&v)[US][TCH])
{
#pragma unroll
    for (int u = 0; u < US; ++u)
#pragma unroll
        for (int t = 0; t < TCH; ++t)       // (no branch around a load: planes past the chunk's last re-read it and are not used)
            v[u][t] = ms_cell(au + ((i + u) * row_stride + (FULL || TCH == 1 || t < nt ? t : nt - 1) * plane), c, (unsigned int)plane * 4u);
}

// `stage` (US == 16, a 1024-thread block = 16 adjacent segments): the maxima of a trip's 16 samples x 16 segments go
// through LDS and leave as 64 contiguous bytes per sample.  A wave storing ONE word per sample writes 4 bytes for every
// 256 it reads, each into a different 64-byte sector of segmax [n][segments]: on single-plane data (C5) that was a quarter
// more HBM traffic than the scores themselves (moments + maxima 6.1 ms against 4.8 for the plain moments).
struct MsStage { unsigned int (*m)[17]; unsigned int *row0; long long nseg_left; };    // LDS [segment][sample], this block's first segment of sample 0
template <int TCH, int US, bool FULL>
__device__ __forceinline__ void ms_use(const float (&v)[US][TCH], int i, int nt, bool scored, double (&s)[TCH], double (&q)[TCH],
                                       unsigned int *__restrict__ seg, long long seg_stride, const MsStage *stage = nullptr)
{
    unsigned int m[US];
#pragma unroll
    for (int u = 0; u < US; ++u) {
        // max |x| without an AND per element: the signed maximum of the raw patterns is the largest non-negative value (a
        // +NaN above everything), the unsigned maximum the largest-magnitude negative one if there is any (a -NaN on top)
        int mp = 0;
        unsigned int mn = 0u;
#pragma unroll
        for (int t = 0; t < TCH; ++t) {     // (branch-free: a plane past the chunk's last adds +0 to sums that are never stored)
            const float x = (FULL || TCH == 1 || t < nt) ? v[u][t] : 0.f;
            const double d = (double)x;
            s[t] += d;
            q[t] = __builtin_fma(d, d, q[t]);           // (d * d is exact in fp64 for an fp32 d: the same sum as multiply-then-add)
            mp = max(mp, (int)__float_as_uint(x));
            mn = max(mn, __float_as_uint(x));
        }
        m[u] = max((unsigned int)mp, mn & 0x7fffffffu);                     // non-negative floats order like their patterns, NaN on top
        m[u] = scored ? m[u] : 0u;
    }
    // wave maxima of US samples at once: each butterfly step halves the samples a lane is responsible for (US-1
    // exchanges instead of 6 US), after which lane L holds sample L / (64/US) and the remaining lane bits are
    // reduced as usual
    const int lane = threadIdx.x & 63;
    if constexpr (US == 1) {
        // one sample: the wave maximum by DPP (row shifts, then the two row broadcasts) lands in lane 63 after six
        // VALU steps, without the LDS-pipe round trips of six ds_bpermute shuffles.  (C3 slab, in units of the residual
        // launch of the same run: 0.254 with shuffles and a branch around every load; 0.233 with DPP, branch-free
        // loads and sums, and the next sample's loads issued before this one is summed; the plain moments pass: 0.218)
        m[0] = wave_max_u32_to_lane63(m[0]);
        if (lane == 63 && seg) seg[i * seg_stride] = m[0];          // (seg == nullptr: a wave wholly past the plane)
    } else {
        ms_butterfly<US, US / 2>(m, lane);
        if constexpr (64 / US > 8) m[0] = max(m[0], dpp_xor<8>(m[0]));
        if constexpr (64 / US > 4) m[0] = max(m[0], dpp_xor<4>(m[0]));
        m[0] = max(m[0], dpp_xor<2>(m[0]));
        m[0] = max(m[0], dpp_xor<1>(m[0]));
        if (US == 16 && stage) {
            if ((lane & 3) == 0) stage->m[threadIdx.x >> 6][lane >> 2] = m[0];
            __syncthreads();
            if (threadIdx.x < 256) {
                const int smp = (int)(threadIdx.x >> 4), sg = (int)(threadIdx.x & 15);
                if (sg < stage->nseg_left) stage->row0[(i + smp) * seg_stride + sg] = stage->m[sg][smp];
            }
            __syncthreads();
        } else if ((lane & (64 / US - 1)) == 0) seg[(i + lane / (64 / US)) * seg_stride] = m[0];
    }
}

// the sample loop of moments_segmax_kernel (two load buffers: the next group's loads are in flight while this one is summed)
template <int TCH, int US, bool FULL>
__device__ __forceinline__ void ms_samples(const float *__restrict__ au, unsigned int c, long long row_stride, long long plane, int i0, int i1,
                                           int nt, bool scored, double (&s)[TCH], double (&q)[TCH], unsigned int *__restrict__ seg,
                                           long long seg_stride, const MsStage *stage = nullptr)
{
    int i = i0;
    if constexpr (US == 1) {
        // (the last trip re-reads its own sample)
        float va[1][TCH], vb[1][TCH];
        if (i < i1) ms_load<TCH, 1, FULL>(au, c, row_stride, plane, i, nt, va);
        for (; i < i1; i += 2) {
            ms_load<TCH, 1, FULL>(au, c, row_stride, plane, min(i + 1, i1 - 1), nt, vb);
            ms_use<TCH, 1, FULL>(va, i, nt, scored, s, q, seg, seg_stride);
            if (i + 1 >= i1) break;
            ms_load<TCH, 1, FULL>(au, c, row_stride, plane, min(i + 2, i1 - 1), nt, va);
            ms_use<TCH, 1, FULL>(vb, i + 1, nt, scored, s, q, seg, seg_stride);
        }
    } else {
        float v[US][TCH], w[US][TCH], v1[1][TCH];
        const int ilast = i0 + (i1 - i0) / US * US - US;                                   // start of the last full group
        if (i + US <= i1) ms_load<TCH, US, FULL>(au, c, row_stride, plane, i, nt, v);
        for (; i + US <= i1; i += 2 * US) {
            ms_load<TCH, US, FULL>(au, c, row_stride, plane, min(i + US, ilast), nt, w);
            ms_use<TCH, US, FULL>(v, i, nt, scored, s, q, seg, seg_stride, stage);
            if (i + 2 * US > i1) { i += US; break; }
            ms_load<TCH, US, FULL>(au, c, row_stride, plane, min(i + 2 * US, ilast), nt, v);
            ms_use<TCH, US, FULL>(w, i + US, nt, scored, s, q, seg, seg_stride, stage);
        }
        for (; i < i1; ++i) {
            ms_load<TCH, 1, FULL>(au, c, row_stride, plane, i, nt, v1);
            ms_use<TCH, 1, FULL>(v1, i, nt, scored, s, q, seg, seg_stride);
        }
    }
}

// (US == 16, single-plane data: 1024-thread blocks = 16 adjacent segments, whose maxima leave through LDS - see MsStage;
// every wave of such a block stays for the barriers, a wave wholly past the plane re-reads the last cell and counts nowhere)
template <int TCH, int US>
__global__ void __launch_bounds__(US == 16 && MS_STAGE ? 1024 : 256)
moments_segmax_kernel(const float *__restrict__ a, long long row_stride, int n, int T, int X,
                                                             int Y, int cx, int cy, int rows_per_split, double *__restrict__ sum,
                                                             double *__restrict__ sumsq, unsigned int *__restrict__ segmax)
{
    // a segment = 64 consecutive cells of the flattened (x, y) plane x 16 planes (it may straddle rows: a bound needs no
    // geometry); lanes past the plane's last cell re-read it and count nowhere
    const long long plane = (long long)X * Y, cl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!(US == 16 && MS_STAGE) && (cl & ~63LL) >= plane) return;                                         // a wave wholly past the plane
    const bool live = cl < plane;
    const long long c = live ? cl : plane - 1;
    const int x = (int)(c / Y), y = (int)(c - (long long)x * Y);
    const long long nseg = (plane + 63) / 64;
    const bool scored = live && x >= cx && x < X - cx && y >= cy && y < Y - cy;
    const int i0 = blockIdx.y * rows_per_split, i1 = min(n, i0 + rows_per_split);
    const int tc = blockIdx.z, TC = gridDim.z, t0 = tc * MS_TMAX, nt = min(TCH, T - t0);
    const float *au = a + (long long)t0 * plane;                                           // wave-uniform
    const long long seg_stride = (long long)TC * nseg;                                     // segmax [n][TC][nseg]
    unsigned int *seg = segmax + (long long)tc * nseg + (cl >> 6);
    double s[TCH], q[TCH];
#pragma unroll
    for (int t = 0; t < TCH; ++t) s[t] = q[t] = 0.0;
    const unsigned int cb = (unsigned int)c * 4u;                                           // (X * Y <= 2^30: checked by the host)
    __shared__ unsigned int stage_m[US == 16 ? 16 : 1][17];
    MsStage st{stage_m, segmax + (long long)tc * nseg + (long long)blockIdx.x * (blockDim.x >> 6),
               nseg - (long long)blockIdx.x * (blockDim.x >> 6)};
    const MsStage *stage = (US == 16 && MS_STAGE) ? &st : nullptr;
    if (US == 16 && MS_STAGE && (cl & ~63LL) >= plane) seg = nullptr;     // (a wave wholly past the plane stays for the barriers but stores nothing)
    if (nt == TCH) ms_samples<TCH, US, true>(au, cb, row_stride, plane, i0, i1, nt, scored, s, q, seg, seg_stride, stage);
    else ms_samples<TCH, US, false>(au, cb, row_stride, plane, i0, i1, nt, scored, s, q, seg, seg_stride, stage);
    sum += (long long)t0 * plane + c;
    sumsq += (long long)t0 * plane + c;
#pragma unroll
    for (int t = 0; t < TCH; ++t)
        if (t < nt && live) {
            if (gridDim.y == 1) { sum[t * plane] += s[t]; sumsq[t * plane] += q[t]; }
            else { atomicAdd(&sum[t * plane], s[t]); atomicAdd(&sumsq[t * plane], q[t]); }
        }
}

__global__ void __launch_bounds__(256) std_from_moments_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq,
                                                               double n_total, long long M, float eps, float *__restrict__ mod)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    const double mean = sum[c] / n_total;
    double var = sumsq[c] / n_total - mean * mean;
    var = var < 0.0 ? 0.0 : var;              // round-off below zero -> 0; NaN stays NaN (np.std propagates it)
    mod[c] = (float)sqrt(var) + eps;
}

// ------------------------------------------------------------------ joint score
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// grid (row-chunks, n); one block reduces JS_RB rows (t,x) of sample blockIdx.y.
// max_i |v_i|/s_i with IEEE division on every element would make the pass VALU-bound (a
// correctly rounded fp32 divide is ~10 instructions).  Each lane keeps its exact running
// maximum m and first tests |v| > m(1-2^-20)*s (one multiply, one compare): a quotient that
// rounds above m always passes (|v|/s >= q(1-2^-24) > m(1-2^-24)), so only the few candidates
// that can raise the maximum pay for the exact divide, and the result is bit-identical to
// dividing everything.
constexpr int JS_RB = 32;
#ifndef JS_TARGET_CELLS
#define JS_TARGET_CELLS 65536    // cells a block of the full score pass takes at least (spans of JS_RB rows; round 6: 8192 ->
                                 // 65536, -5 ... -11 % on every shape of profiles/r06/score_pass_block_ab.txt: 32 KB blocks were too short-lived)
#endif
// NaN: np.max propagates it.  A NaN residual or modulation fails `av <= thr*sv` and reaches the divide; 0/0 (a
// constant cell: modulation 0 and residual 0) reaches it through `sv == 0`; a NaN quotient sets the sticky flag and
// the sample's score becomes NaN.
__device__ __forceinline__ void js_update(float av, float sv, float &m, float &thr, bool &nan)
{
    // (sv below the smallest normal number: thr * sv is then rounded with an ABSOLUTE error, and "a quotient that rounds
    // above m always passes" no longer follows - such cells, and sv == 0, always take the divide)
    if (!(av <= thr * sv) || sv < 1.17549435e-38f) {
        const float q = av / sv;
        if (q != q) nan = true;
        else if (q > m) { m = q; thr = m * 0.99999905f; }
    }
}

__global__ void __launch_bounds__(256) joint_score_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ mod, int T, int X, int Y,
                                                          int ct, int cx, int cy, int groups, int smp_fastest,
                                                          const unsigned int *__restrict__ only, float *__restrict__ scores)
{
    const long long plane = (long long)X * Y, vol = plane * T;
    // smp_fastest: consecutive workgroups take the SAME rows of consecutive samples, so the modulation rows they
    // share stay in L2 (chunk-fastest order swept the whole modulation, 14 MB for a C3 slab, once per sample).
    // (Tried on top and dropped: four load pairs in flight per thread - 74 VGPRs, 15 % slower.)
    const int smp = smp_fastest ? blockIdx.x : blockIdx.y;
    const unsigned int cblk = smp_fastest ? blockIdx.y : blockIdx.x;
    if (only && !only[smp]) return;                           // (block-uniform) the pass over the flagged samples only
    const float *pa = a + smp * vol, *pb = b ? b + smp * vol : nullptr;
    const long long nrows = (long long)T * X;
    const bool vec = (Y % 4 == 0) && !(((uintptr_t)a | (uintptr_t)mod | (uintptr_t)(b ? b : a)) & 15);
    __shared__ unsigned int rowmask;
    __shared__ float red[4];
    float m = 0.f, thr = 0.f;
    bool any = false, nan = false;
    // `groups` spans of JS_RB rows per block (short rows - e.g. the surrogate's Nt = 10 cells - would otherwise
    // leave a block with a few hundred cells and one atomic each)
    for (int gi = 0; gi < groups; ++gi) {
    const long long r0 = ((long long)cblk * groups + gi) * JS_RB;
    if (r0 >= nrows) break;
    __syncthreads();             // the previous span's rowmask has been read by everyone
    if (threadIdx.x < 64) {      // which of this span's rows survive the (t, x) crop
        const long long r = r0 + threadIdx.x;
        bool ok = threadIdx.x < JS_RB && r < nrows;
        if (ok) {
            const int t = (int)(r / X), x = (int)(r % X);
            ok = !(t < ct || t >= T - ct || x < cx || x >= X - cx);
        }
        const unsigned long long bm = __ballot(ok);
        if (threadIdx.x == 0) rowmask = (unsigned int)bm;
    }
    __syncthreads();
    const unsigned int rows = rowmask;
    any = any || rows != 0u;
    if (rows && vec) {
        // the block's rows are one contiguous span: walk it as float4 items, all 256 lanes busy
        const int Y4 = Y / 4, dr = 256 / Y4, dy = 256 % Y4;
        int row = threadIdx.x / Y4, y4 = threadIdx.x % Y4;
        while (row < JS_RB) {
            if ((rows >> row) & 1u) {
                const long long o = (r0 + row) * Y + 4 * y4;
                float4 v = *reinterpret_cast<const float4 *>(pa + o);
                if (pb) { const float4 w = *reinterpret_cast<const float4 *>(pb + o); v = make_float4(v.x - w.x, v.y - w.y, v.z - w.z, v.w - w.w); }
                const float4 sg = *reinterpret_cast<const float4 *>(mod + o);
                const float av[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
                const float sv[4] = {sg.x, sg.y, sg.z, sg.w};
                const int y = 4 * y4;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (y + k >= cy && y + k < Y - cy) js_update(av[k], sv[k], m, thr, nan);
            }
            y4 += dy;
            row += dr;
            if (y4 >= Y4) { y4 -= Y4; ++row; }
        }
    } else if (rows) {
        // same walk cell by cell (Y % 4 != 0 or unaligned: e.g. the surrogate's Nt-fastest order, rows of Nt = 10 cells)
        const int dr = 256 / Y, dy = 256 % Y;
        int row = threadIdx.x / Y, y = threadIdx.x % Y;
        while (row < JS_RB) {
            if (((rows >> row) & 1u) && y >= cy && y < Y - cy) {
                const long long o = (r0 + row) * Y + y;
                float v = pa[o];
                if (pb) v -= pb[o];
                js_update(fabsf(v), mod[o], m, thr, nan);
            }
            y += dy;
            row += dr;
            if (y >= Y) { y -= Y; ++row; }
        }
    }
    }   // spans
    m = wave_max(m);
    if (__ballot(nan)) m = __uint_as_float(0x7fc00000u);         // canonical quiet NaN
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0 && any) {
        // non-negative floats order like their bit patterns, and the NaN pattern lies above +inf: the unsigned
        // maximum is the float maximum with NaN sticky (across blocks and across per-slab calls)
        const unsigned int u = max(max(__float_as_uint(red[0]), __float_as_uint(red[1])),
                                   max(__float_as_uint(red[2]), __float_as_uint(red[3])));
        atomicMax(reinterpret_cast<unsigned int *>(scores) + smp, u);
    }
}

// Flat form for a SHORT innermost axis (the surrogate's memory order [n, Nx, Ny, Nt], Nt = 10..40: rows of 10 cells
// kept 10 of a wave's 64 lanes busy on the row walk above, 1.5 TB/s).  A block takes JSF_CELLS consecutive cells of one
// sample as quads, whatever the row length; the (t, x, y) position of a quad's first cell is one division per quad
// and the crop test walks the four cells from there.  vol % 4 == 0 and 16-byte aligned pointers.
constexpr int JSF_CELLS = 8192;
__global__ void __launch_bounds__(256) joint_score_flat_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                               const float *__restrict__ mod, int T, int X, int Y, int ct,
                                                               int cx, int cy, const unsigned int *__restrict__ only,
                                                               float *__restrict__ scores)
{
    const long long vol = (long long)T * X * Y;
    const int smp = blockIdx.x;
    if (only && !only[smp]) return;
    const float4 *pa = reinterpret_cast<const float4 *>(a + smp * vol);
    const float4 *pb = b ? reinterpret_cast<const float4 *>(b + smp * vol) : nullptr;
    const float4 *pm = reinterpret_cast<const float4 *>(mod);
    const unsigned int c0 = blockIdx.y * (unsigned)JSF_CELLS;     // vol < 2^31 (host check): 32-bit cell indices
    __shared__ float red[4];
    float m = 0.f, thr = 0.f;
    bool nan = false;
    // position of my first quad (one 32-bit division pair per thread), then 1024 cells further per trip
    unsigned int f = c0 + 4u * threadIdx.x;
    const unsigned int row0 = f / (unsigned)Y;
    int y0 = (int)(f - row0 * (unsigned)Y), x0 = (int)(row0 % (unsigned)X), t0 = (int)(row0 / (unsigned)X);
    const int dy = 1024 % Y, drow = 1024 / Y, dx = drow % X, dt = drow / X;
    for (int k = 0; k < JSF_CELLS / 1024 && f < (unsigned)vol; ++k, f += 1024u) {
        float4 v = pa[f >> 2];
        if (pb) { const float4 w = pb[f >> 2]; v = make_float4(v.x - w.x, v.y - w.y, v.z - w.z, v.w - w.w); }
        const float4 sg = pm[f >> 2];
        const float av[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
        const float sv[4] = {sg.x, sg.y, sg.z, sg.w};
        int y = y0, x = x0, t = t0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = t >= ct && t < T - ct && x >= cx && x < X - cx && y >= cy && y < Y - cy;
            if (in) js_update(av[j], sv[j], m, thr, nan);
            if (++y == Y) { y = 0; if (++x == X) { x = 0; ++t; } }
        }
        y0 += dy;
        const int carry = y0 >= Y;
        y0 -= carry ? Y : 0;
        x0 += dx + carry;                                         // < 2X
        const int cx2 = x0 >= X;
        x0 -= cx2 ? X : 0;
        t0 += dt + cx2;
    }
    m = wave_max(m);
    if (__ballot(nan)) m = __uint_as_float(0x7fc00000u);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int u = max(max(__float_as_uint(red[0]), __float_as_uint(red[1])),
                                   max(__float_as_uint(red[2]), __float_as_uint(red[3])));
        if (u) atomicMax(reinterpret_cast<unsigned int *>(scores) + smp, u);
    }
}

// ------------------------------------------------------------------ pruned joint score (branch and bound)
// The score of a sample is a MAXIMUM over cells, and for a segment S (64 consecutive cells of the flattened (x, y)
// plane x up to 16 planes)
//     max_{c in S} fl(|r_c| / mod_c)  <=  fl( max_S |r_c| / min_S mod_c )
// because correctly rounded division is monotone in both arguments.  moments_segmax_kernel delivers max_S |r| per
// sample (segmax) from the read the moments need anyway, segmin_kernel min_S mod once per slab; a block per sample then (1) evaluates the segment with the
// largest bound exactly, (2) lists the segments whose bound still exceeds the best so far (earlier slabs' score
// included) and evaluates only those.  On noise-like residuals that is a handful of the 4096 segments of a C3 slab:
// the pass reads the 16 KB of bounds per sample instead of 13.6 MB of residual.
constexpr int JP_SEG = 64;

__global__ void __launch_bounds__(64) segmin_kernel(const float *__restrict__ mod, int T, int X, int Y, int cx, int cy,
                                                    long long nseg, float *__restrict__ segmin)
{
    const long long plane = (long long)X * Y, seg = blockIdx.x, c = seg * JP_SEG + (long long)threadIdx.x;
    const int tc = blockIdx.y, x = (int)(c / Y), y = (int)(c - (long long)x * Y);
    float m = __builtin_inff();
    bool bad = false;
    if (c < plane && x >= cx && x < X - cx && y >= cy && y < Y - cy)
        for (int t = tc * MS_TMAX; t < min(T, (tc + 1) * MS_TMAX); ++t) {
            const float v = mod[t * plane + c];
            bad |= !(v > 0.f);                                    // NaN or <= 0: no usable bound
            m = fminf(m, v);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    bad = __ballot(bad) != 0;
    if (threadIdx.x == 0) segmin[(long long)tc * nseg + seg] = bad ? 0.f : m;
}

// WAVE: many samples with short lists (C5: 65 536 samples of one 200 x 512 plane = 1600 segments each) - one WAVE per
// sample, four samples per workgroup: no wave waits at a barrier for the others' global loads, and the workgroup adds its
// four samples' statistics with ONE global atomic per counter (a block per sample put 2 x 65 536 atomics on one cache line:
// 1.05 ms of the kernel's 1.8, tools/exp/c5_stats_probe.py).  Otherwise a block of 4-16 waves per sample shares the list.
template <bool WAVE>
__global__ void __launch_bounds__(1024) joint_score_pruned_kernel(const float *__restrict__ res, long long row_stride,
                                                                  const float *__restrict__ mod,
                                                                  const unsigned int *__restrict__ segmax,
                                                                  const float *__restrict__ segmin, int n, int T, int X, int Y, int cx,
                                                                  int cy, int per_chunk, int total, float *__restrict__ scores,
                                                                  unsigned int *__restrict__ flags,
                                                                  unsigned long long *__restrict__ stats)
{
    extern __shared__ unsigned int work_all[];                    // `total` = TC * X * nseg segment ids (WAVE: per wave)
    __shared__ unsigned int red[16], redi[16], nwork_[4], sbest_[4], blk_read, blk_flagged;
    const int lane = threadIdx.x & 63;
    const int wave = WAVE ? 0 : (int)(threadIdx.x >> 6), nwaves = WAVE ? 1 : (int)(blockDim.x >> 6);
    const int mywave = (int)(threadIdx.x >> 6);                   // (WAVE: which of the block's samples)
    const int tid = WAVE ? lane : (int)threadIdx.x, nthreads = WAVE ? 64 : (int)blockDim.x;
    const int smp = WAVE ? (int)blockIdx.x * 4 + mywave : (int)blockIdx.x;
    unsigned int *work = work_all + (WAVE ? mywave * total : 0);
    unsigned int &nwork = nwork_[WAVE ? mywave : 0], &sbest = sbest_[WAVE ? mywave : 0];
    // a workgroup-wide rendezvous where the waves of a block share a sample; within ONE wave the LDS is in order already
    auto sync = [&]() __attribute__((always_inline)) {
        if constexpr (WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    if (WAVE && threadIdx.x == 0) { blk_read = 0u; blk_flagged = 0u; }
    if (WAVE) __syncthreads();
    unsigned int my_read = 0u, my_flagged = 0u;                   // this sample's statistics (WAVE: added up per block)
    do {                                                          // (one trip; `break` = this sample is done)
    if (WAVE && smp >= n) break;
    const float *pr = res + smp * row_stride;
    const unsigned int *pm = segmax + (long long)smp * total;
    const long long plane = (long long)X * Y;
    const unsigned int NANBITS = 0x7fc00000u;

    // upper bound of a segment's scores as a bit pattern (NaN above everything: "must be read")
    auto bound = [&](int j) __attribute__((always_inline)) {
        const float sm = segmin[j];
        if (sm == 0.f) return NANBITS;                            // a NaN or zero modulation inside: 0/0 and x/NaN are NaN
        const unsigned int mx = pm[j];
        if (mx == 0u) return 0u;                                  // rim row / all residuals zero over positive modulations
        const float q = __uint_as_float(mx) / sm;                 // mx NaN -> NaN
        return q != q ? NANBITS : __float_as_uint(q);
    };
    // exact maximum of |r| / mod over one segment, by one wave (a lane per column, up to 16 planes in flight)
    auto evaluate = [&](int j) __attribute__((always_inline)) {
        const int tc = j / per_chunk;
        const long long c = (long long)(j - tc * per_chunk) * JP_SEG + lane;               // cell of the flattened plane
        const int x = (int)(c / Y), y = (int)(c - (long long)x * Y);
        const int t0 = tc * MS_TMAX, planes = min(MS_TMAX, T - t0);
        unsigned int m = 0u;
        if (c < plane && x >= cx && x < X - cx && y >= cy && y < Y - cy) {
            const long long o = (long long)t0 * plane + c;
            float rv[MS_TMAX], mv[MS_TMAX];
#pragma unroll
            for (int t = 0; t < MS_TMAX; ++t) {                   // (no branch around a load: a short chunk re-reads its last plane)
                const int tt = t < planes ? t : planes - 1;
                rv[t] = pr[o + tt * plane];
                mv[t] = mod[o + tt * plane];
            }
#pragma unroll
            for (int t = 0; t < MS_TMAX; ++t) {
                // (a quotient <= 0 - a negative modulation - never raises the full pass's maximum either: js_update)
                const float q = fabsf(rv[t]) / mv[t];
                m = max(m, q != q ? NANBITS : (q > 0.f ? __float_as_uint(q) : 0u));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o));
        return m;
    };

    // (1) the segment with the largest bound: evaluated first, so that most of the others never are
    unsigned int bb = 0u, bj = 0u;
    for (int j = tid; j < total; j += nthreads) {
        const unsigned int b = bound(j);
        if (b > bb) { bb = b; bj = (unsigned)j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int ob = (unsigned int)__shfl_xor((int)bb, o), oj = (unsigned int)__shfl_xor((int)bj, o);
        if (ob > bb || (ob == bb && oj < bj)) { bb = ob; bj = oj; }
    }
    if (!WAVE && lane == 0) { red[wave] = bb; redi[wave] = bj; }
    if (tid == 0) nwork = 0u;
    sync();
    if (!WAVE) {
        bb = red[0], bj = redi[0];                                // (every thread the same)
        for (int w = 1; w < nwaves; ++w)
            if (red[w] > bb || (red[w] == bb && redi[w] < bj)) { bb = red[w]; bj = redi[w]; }
    }
    unsigned int best = __float_as_uint(scores[smp]);            // earlier slabs (non-negative or NaN: orders as uint)
    if (bb == 0u) {                                               // nothing scored in this slab (block- / wave-uniform)
        if (flags && tid == 0) flags[smp] = 0u;
        break;
    }
    if (bb > best) best = max(best, evaluate((int)bj));          // (every wave evaluates it: no exchange needed)
    // (2) every other segment that can still beat it
    for (int j = tid; j < total; j += nthreads)
        if ((unsigned)j != bj && bound(j) > best) work[atomicAdd(&nwork, 1u)] = (unsigned)j;
    if (tid == 0) sbest = best;
    sync();
    // the waves share the list; the best score so far is shared through LDS (it only grows, and skipping a segment
    // whose bound does not exceed ANY score already seen is always safe)
    const unsigned int nw = nwork;
    {                                                             // [0] += segments read, [2] += samples swept whole (at the end)
        const bool giveup = flags && 4u * nw > (unsigned)total;
        my_read = giveup ? (unsigned)total : nw + 1u;
        my_flagged = giveup ? 1u : 0u;
    }
    if (flags) {
        // The bounds do not prune this sample when more than a quarter of its segments would be read, 8 KB at a time (a
        // modulation that jumps between neighbouring cells: max |r| / min mod says little): it is FLAGGED for the full
        // pass over the flagged samples that the driver launches behind this kernel (pre_joint_score_flagged_f32).
        const bool giveup = 4u * nw > (unsigned)total;
        if (tid == 0) flags[smp] = giveup ? 1u : 0u;
        if (giveup) {
            if (tid == 0) atomicMax(reinterpret_cast<unsigned int *>(scores) + smp, best);
            break;
        }
    }
    for (unsigned int i = wave; i < nw; i += nwaves) {
        const int j = (int)work[i];
        best = max(best, *(volatile unsigned int *)&sbest);
        if (bound(j) > best) {                                    // (wave-uniform)
            const unsigned int v = evaluate(j);
            if (v > best) { best = v; if (lane == 0) atomicMax(&sbest, v); }
        }
    }
    sync();
    if (tid == 0) atomicMax(reinterpret_cast<unsigned int *>(scores) + smp, max(best, sbest));
    } while (false);
    // statistics: [1] (segments: total per sample, whatever happened to it) once per launch, [0] / [2] once per block
    if (!stats) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[1], (unsigned long long)n * (unsigned long long)total);
    if constexpr (WAVE) {
        if (lane == 0 && my_read) atomicAdd(&blk_read, my_read);
        if (lane == 0 && my_flagged) atomicAdd(&blk_flagged, my_flagged);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (blk_read) atomicAdd(&stats[0], (unsigned long long)blk_read);
            if (blk_flagged) atomicAdd(&stats[2], (unsigned long long)blk_flagged);
        }
    } else if (threadIdx.x == 0) {
        if (my_read) atomicAdd(&stats[0], (unsigned long long)my_read);
        if (my_flagged) atomicAdd(&stats[2], 1ull);
    }
}

// ------------------------------------------------------------------ scalar k-th (radix select)
struct KList { int nk; long long k[16]; };

__global__ void __launch_bounds__(1024) kth_kernel(const float *__restrict__ s, long long N, const KList kl, float *__restrict__ out)
{
    // one workgroup per requested rank (N per-sample scores fit L2: the re-reads are cache hits)
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sh_prefix;
    __shared__ long long sh_rank;
    __shared__ int sh_nan;
    const int j = blockIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned int prefix = 0, mask = 0;
    long long rank = kl.k[j];
    if (threadIdx.x == 0) sh_nan = 0;
    const bool vec = !((uintptr_t)s & 15);
    const long long N8 = vec ? N / 8 : 0;                       // items of two float4 per thread and trip
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        auto count = [&](float v) __attribute__((always_inline)) {
            if (shift == 24 && v != v) sh_nan = 1;              // np.quantile: any NaN score makes every quantile NaN
            const unsigned int key = f2key(v);
            const bool hit = (key & mask) == prefix;
            const unsigned int d = (key >> shift) & 255u;
            // scores of one calibration set share their top byte (sign + exponent): 1024 threads adding to ONE LDS
            // counter serialise lane by lane.  When every matching lane of the wave has the same digit, one lane adds
            // the count (the later digits are spread and take the plain path).
            const unsigned long long hm = __ballot(hit);
            if (hm) {
                const int first = __builtin_ctzll(hm);
                const unsigned int d0 = __builtin_amdgcn_readlane((int)d, first);
                if (__ballot(hit && d == d0) == hm) {
                    if (lane == first) atomicAdd(&hist[d0], (unsigned)__popcll(hm));
                } else if (hit) {
                    atomicAdd(&hist[d], 1u);
                }
            }
        };
        // eight scores per thread and trip, both loads issued before the first is used (one dependent scalar load per
        // trip made the single workgroup latency-bound: 0.43 ms for N = 524288)
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        for (long long i = threadIdx.x; i < N8; i += blockDim.x) {
            const float4 a = s4[2 * i], b = s4[2 * i + 1];
            count(a.x); count(a.y); count(a.z); count(a.w);
            count(b.x); count(b.y); count(b.z); count(b.w);
        }
        for (long long i = 8 * N8 + threadIdx.x; i < N; i += blockDim.x) count(s[i]);
        __syncthreads();
        if (threadIdx.x < 64) {
            // digit of the rank: wave-parallel scan of the 256 bins (4 per lane)
            const unsigned int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            long long incl = (long long)c0 + c1 + c2 + c3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const long long up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            long long excl = incl - ((long long)c0 + c1 + c2 + c3);
            const bool mine = (excl <= rank && rank < incl) || (lane == 63 && rank >= incl);      // (rank < N: the last lane is a guard)
            if (mine) {
                unsigned int d = 4 * lane;
                if (rank >= excl + c0) { excl += c0; ++d; if (rank >= excl + c1) { excl += c1; ++d; if (rank >= excl + c2) { excl += c2; ++d; } } }
                sh_prefix = prefix | (d << shift);
                sh_rank = rank - excl;
            }
        }
        __syncthreads();
        prefix = sh_prefix;
        rank = sh_rank;
        mask |= 255u << shift;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = sh_nan ? __uint_as_float(0x7fc00000u) : key2f(prefix);
}

// ------------------------------------------------------------------ coverage
__global__ void __launch_bounds__(256) cov_count_kernel(const float *__restrict__ y, const float *__restrict__ lo,
                                                        const float *__restrict__ hi, long long total, long long M,
                                                        int per_sample, unsigned long long *count)
{
    unsigned int local = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long c = per_sample ? i : i % M;
        const float v = y[i];
        local += (v >= lo[c]) && (v <= hi[c]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    __shared__ unsigned int red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}

// Vector form for M % 4 == 0 and 16-byte aligned rows: grid (quad-column chunks, sample groups).  A thread owns ONE
// column of 4 cells: its bounds are loaded once (per-cell bounds) and its samples stream by as float4, eight loads in
// flight - against one scalar load and a 64-bit `i % M` per element above (2.0-2.2 TB/s at the reference's sizes).
__global__ void __launch_bounds__(256) cov_count4_kernel(const float4 *__restrict__ y, const float4 *__restrict__ lo,
                                                         const float4 *__restrict__ hi, int n, long long M4, int per_sample,
                                                         int rows_per, unsigned long long *count)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int local = 0;
    if (q < M4) {
        const int s0 = blockIdx.y * rows_per, s1 = min(n, s0 + rows_per);
        float4 l = lo[q], h = hi[q];
        auto inside = [&](const float4 &v, const float4 &a, const float4 &b) __attribute__((always_inline)) {
            return (unsigned)((v.x >= a.x) && (v.x <= b.x)) + (unsigned)((v.y >= a.y) && (v.y <= b.y)) +
                   (unsigned)((v.z >= a.z) && (v.z <= b.z)) + (unsigned)((v.w >= a.w) && (v.w <= b.w));
        };
        int s = s0;
        if (!per_sample) {
            for (; s + 7 < s1; s += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = y[(long long)(s + u) * M4 + q];
#pragma unroll
                for (int u = 0; u < 8; ++u) local += inside(v[u], l, h);
            }
        }
        for (; s < s1; ++s) {
            const long long o = (long long)s * M4 + q;
            if (per_sample) { l = lo[o]; h = hi[o]; }
            local += inside(y[o], l, h);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    __shared__ unsigned int red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0 && (red[0] | red[1] | red[2] | red[3])) atomicAdd(count, (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}

// grid (chunks, n): clear inside[i] if any cell of sample i leaves [lo, hi]
__global__ void __launch_bounds__(256) cov_joint_kernel(const float *__restrict__ y, const float *__restrict__ lo,
                                                        const float *__restrict__ hi, long long M, int per_sample,
                                                        uint8_t *inside)
{
    const int smp = blockIdx.y;
    const float *py = y + smp * M;
    const float *plo = lo + (per_sample ? smp * M : 0), *phi = hi + (per_sample ? smp * M : 0);
    bool bad = false;
    if (M % 4 == 0 && !(((uintptr_t)y | (uintptr_t)lo | (uintptr_t)hi) & 15)) {      // rows of whole, aligned quads
        const float4 *y4 = reinterpret_cast<const float4 *>(py), *l4 = reinterpret_cast<const float4 *>(plo),
                     *h4 = reinterpret_cast<const float4 *>(phi);
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < M / 4; q += (long long)gridDim.x * blockDim.x) {
            const float4 v = y4[q], a = l4[q], b = h4[q];
            bad |= !((v.x >= a.x) && (v.x <= b.x) && (v.y >= a.y) && (v.y <= b.y) && (v.z >= a.z) && (v.z <= b.z) &&
                     (v.w >= a.w) && (v.w <= b.w));
        }
    } else {
        for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < M; c += (long long)gridDim.x * blockDim.x) {
            const float v = py[c];
            bad |= !((v >= plo[c]) && (v <= phi[c]));
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) inside[smp] = 0;
}

// grid (chunks, n): counts[i] += #cells of sample i inside [lo,hi] (outside == 0) or with
// y <= lo || y >= hi (outside != 0)   (Active_Learning/Advection_AL_Marginal.py:190-193)
__global__ void __launch_bounds__(256) cov_rowcount_kernel(const float *__restrict__ y, const float *__restrict__ lo,
                                                           const float *__restrict__ hi, long long M, int per_sample,
                                                           int outside, unsigned int *__restrict__ counts)
{
    const int smp = blockIdx.y;
    const float *py = y + smp * M;
    const float *plo = lo + (per_sample ? smp * M : 0), *phi = hi + (per_sample ? smp * M : 0);
    unsigned int local = 0;
    auto hit = [&](float v, float a, float b) __attribute__((always_inline)) {
        return (unsigned)(outside ? ((v <= a) || (v >= b)) : ((v >= a) && (v <= b)));
    };
    if (M % 4 == 0 && !(((uintptr_t)y | (uintptr_t)lo | (uintptr_t)hi) & 15)) {      // rows of whole, aligned quads
        const float4 *y4 = reinterpret_cast<const float4 *>(py), *l4 = reinterpret_cast<const float4 *>(plo),
                     *h4 = reinterpret_cast<const float4 *>(phi);
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < M / 4; q += (long long)gridDim.x * blockDim.x) {
            const float4 v = y4[q], a = l4[q], b = h4[q];
            local += hit(v.x, a.x, b.x) + hit(v.y, a.y, b.y) + hit(v.z, a.z, b.z) + hit(v.w, a.w, b.w);
        }
    } else {
        for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < M; c += (long long)gridDim.x * blockDim.x)
            local += hit(py[c], plo[c], phi[c]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(counts + smp, local);
}

inline unsigned grid_for(long long items, int block, long long cap = 256LL * 32)
{
    long long g = (items + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

}  // namespace

extern "C" {

int pre_absdiff_f32(const float *a, const float *b, float *out, int64_t n, void *stream)
{
    if (!a || !out || n < 0) return PRE_E_NULL;
    if (n == 0) return PRE_OK;
    hipLaunchKernelGGL(absdiff_kernel, dim3(grid_for(n / 16 + 1, 256, 0x7fffffffLL)), dim3(256), 0, as_stream(stream), a, b, out, (long long)n);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_std_axis0_f32(const float *a, const float *b, int64_t n, int64_t M, float eps, float *mod, void *stream)
{
    if (!a || !mod || n <= 0 || M <= 0) return PRE_E_NULL;
    if (n > 0x7fffffff || M > 0xffffff00LL) return PRE_E_SHAPE;        // one thread per cell, 32-bit work-item count
    // one thread per cell (the sums are sequential, numpy's order): few cells -> one wave per workgroup, so that the
    // waves spread over all CUs (C5's 102 400 cells are 400 workgroups of 256 threads: 1.6 per CU)
    const int bs = M >= (1 << 19) ? 256 : 64;
    hipLaunchKernelGGL(std_axis0_kernel, dim3((unsigned)((M + bs - 1) / bs)), dim3(bs), 0, as_stream(stream), a, b, (int)n,
                       (long long)M, eps, mod);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_moments_axis0_f64(const float *a, const float *b, int64_t n, int64_t M, int64_t row_stride, double *sum,
                          double *sumsq, void *stream)
{
    if (!a || !sum || !sumsq || n <= 0 || M <= 0 || row_stride < M) return PRE_E_NULL;
    if (n > 0x7fffffff || M > 0xffffff00LL) return PRE_E_SHAPE;        // one thread per cell, 32-bit work-item count
    const long long bx = (M + 255) / 256;
    // enough workgroups to fill 256 CUs even when M is small: split the batch axis
    long long splits = 1;
    while (bx * splits < 1024 && splits * 32 < n) splits *= 2;
    const int rows = (int)((n + splits - 1) / splits);
    splits = (n + rows - 1) / rows;
    hipLaunchKernelGGL(moments_kernel, dim3((unsigned)bx, (unsigned)splits), dim3(256), 0, as_stream(stream), a, b, (int)n,
                       (long long)M, (long long)row_stride, rows, sum, sumsq);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_moments_segmax_f64(const float *a, int64_t row_stride, int64_t n, int64_t T, int64_t X, int64_t Y, int crop_x, int crop_y,
                           double *sum, double *sumsq, uint32_t *segmax, void *stream)
{
    if (!a || !sum || !sumsq || !segmax || n <= 0 || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (crop_x < 0 || crop_y < 0 || row_stride < T * X * Y) return PRE_E_RANGE;
    const int us = T == 1 ? 16 : T <= 4 ? 4 : 1;
    const int bt = (us == 16 && MS_STAGE) ? 1024 : 256;                                // (single-plane data: 16 segments per block, see MsStage)
    const long long bx = (X * Y + bt - 1) / bt, TC = (T + MS_TMAX - 1) / MS_TMAX;
    if (n > 0x7fffffff || bx > 0x7fffffffLL || TC > 65535 || T > 0x7fffffff) return PRE_E_SHAPE;
    if (X * Y >= (1LL << 30)) return PRE_E_SHAPE;                        // (a lane's cell is a 32-bit BYTE offset from the plane's base)
    long long splits = 1;
    // (the few-plane forms run 8 waves per SIMD; the 1024-thread form one block per CU at a time: enough splits of the batch axis
    // that the last, partial round of blocks is a small share - 800 blocks on 256 CUs were 3.1 rounds, the fourth a quarter full)
    const long long want = us == 16 ? 16384 : us > 1 ? 2048 : MS_WANT1;
    while (bx * (bt / 256) * TC * splits < want && splits * 32 * us < n) splits *= 2;
    const int rows = (int)((n + splits - 1) / splits);
    splits = (n + rows - 1) / rows;
    const dim3 grid((unsigned)bx, (unsigned)splits, (unsigned)TC);
#define PRE_MS_LAUNCH(TCH, US)                                                                                                \
    hipLaunchKernelGGL((moments_segmax_kernel<TCH, US>), grid, dim3(bt), 0, as_stream(stream), a, (long long)row_stride, (int)n,    \
                       (int)T, (int)X, (int)Y, crop_x, crop_y, rows, sum, sumsq, segmax)
    if (us == 16) PRE_MS_LAUNCH(1, 16);
    else if (us == 4) PRE_MS_LAUNCH(4, 4);
    else PRE_MS_LAUNCH(MS_TMAX, 1);
#undef PRE_MS_LAUNCH
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_std_from_moments_f32(const double *sum, const double *sumsq, int64_t n_total, int64_t M, float eps, float *mod,
                             void *stream)
{
    if (!sum || !sumsq || !mod || n_total <= 0 || M <= 0) return PRE_E_NULL;
    if (M > 0xffffff00LL) return PRE_E_SHAPE;                           // one thread per cell, 32-bit work-item count
    hipLaunchKernelGGL(std_from_moments_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), sum, sumsq,
                       (double)n_total, (long long)M, eps, mod);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

static int joint_score_launch(const float *a, const float *b, const float *mod, int64_t n, int64_t T, int64_t X, int64_t Y,
                              int crop_t, int crop_x, int crop_y, const uint32_t *only, float *scores, void *stream)
{
    if (!a || !mod || !scores || n <= 0 || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (crop_t < 0 || crop_x < 0 || crop_y < 0) return PRE_E_RANGE;
    if (n > 65535 * 1024LL || T * X > 0x7fffffffLL * JS_RB || Y > 0x7fffffff) return PRE_E_SHAPE;
    {
        // short rows: flat form (sample-fastest block order, like the row form)
        const long long vol = (long long)T * X * Y, nblk = (vol + JSF_CELLS - 1) / JSF_CELLS;
        if (Y < 64 && vol % 4 == 0 && vol < 0x7fffffffLL && nblk <= 65535 && n <= 0x7fffffff &&
            !(((uintptr_t)a | (uintptr_t)mod | (uintptr_t)(b ? b : a)) & 15)) {
            hipLaunchKernelGGL(joint_score_flat_kernel, dim3((unsigned)n, (unsigned)nblk), dim3(256), 0, as_stream(stream), a, b, mod,
                               (int)T, (int)X, (int)Y, crop_t, crop_x, crop_y, only, scores);
            PRE_LAUNCH_CHECK();
            return PRE_OK;
        }
    }
    // spans of JS_RB rows; a block takes as many as give it >= JS_TARGET_CELLS cells (but leaves >= ~2 k blocks when possible)
    const long long spans = (T * X + JS_RB - 1) / JS_RB;
    long long groups = JS_TARGET_CELLS / (JS_RB * Y);
    if (groups < 1) groups = 1;
    while (groups > 1 && (spans / groups) * n < 2048) groups /= 2;
    const long long chunks = (spans + groups - 1) / groups;
    // gridDim.y is limited to 65535: walk the batch axis in slices
    for (int64_t s0 = 0; s0 < n; s0 += 65535) {
        const int64_t ns = (n - s0) < 65535 ? (n - s0) : 65535;
        const long long vol = (long long)T * X * Y;
        const int swap = chunks <= 65535;
        hipLaunchKernelGGL(joint_score_kernel, swap ? dim3((unsigned)ns, (unsigned)chunks) : dim3((unsigned)chunks, (unsigned)ns),
                           dim3(256), 0, as_stream(stream), a + s0 * vol, b ? b + s0 * vol : nullptr, mod, (int)T, (int)X, (int)Y,
                           crop_t, crop_x, crop_y, (int)groups, swap, only ? only + s0 : nullptr, scores + s0);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

int pre_joint_score_f32(const float *a, const float *b, const float *mod, int64_t n, int64_t T, int64_t X, int64_t Y,
                        int crop_t, int crop_x, int crop_y, float *scores, void *stream)
{
    return joint_score_launch(a, b, mod, n, T, X, Y, crop_t, crop_x, crop_y, nullptr, scores, stream);
}

int pre_joint_score_flagged_f32(const float *a, const float *b, const float *mod, int64_t n, int64_t T, int64_t X, int64_t Y,
                                int crop_t, int crop_x, int crop_y, const uint32_t *flags, float *scores, void *stream)
{
    if (!flags) return PRE_E_NULL;
    return joint_score_launch(a, b, mod, n, T, X, Y, crop_t, crop_x, crop_y, flags, scores, stream);
}

int pre_segmin_mod_f32(const float *mod, int64_t T, int64_t X, int64_t Y, int crop_x, int crop_y, float *segmin, void *stream)
{
    if (!mod || !segmin || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (crop_x < 0 || crop_y < 0) return PRE_E_RANGE;
    const long long nseg = (X * Y + JP_SEG - 1) / JP_SEG, TC = (T + MS_TMAX - 1) / MS_TMAX;
    if (X > 0x7fffffff || Y > 0x7fffffff || nseg > 0x7fffffffLL || TC > 65535) return PRE_E_SHAPE;
    hipLaunchKernelGGL(segmin_kernel, dim3((unsigned)nseg, (unsigned)TC), dim3(64), 0, as_stream(stream), mod, (int)T, (int)X, (int)Y,
                       crop_x, crop_y, nseg, segmin);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

// bytes of LDS a workgroup of the current device may hold (64 KiB if the runtime cannot say)
static int pruned_lds_max()
{
    int dev = 0, lds_max = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds_max <= 0) {
        (void)hipGetLastError();
        lds_max = 64 * 1024;
    }
    return lds_max;
}

int64_t pre_joint_score_pruned_max_segments(void) { return (pruned_lds_max() - 256) / 4; }

int pre_joint_score_pruned_f32(const float *res, int64_t row_stride, const float *mod, const uint32_t *segmax, const float *segmin,
                               int64_t n, int64_t T, int64_t X, int64_t Y, int crop_x, int crop_y, float *scores,
                               uint32_t *flags, unsigned long long *stats, void *stream)
{
    if (!res || !mod || !segmax || !segmin || !scores || n <= 0 || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (crop_x < 0 || crop_y < 0 || row_stride < T * X * Y) return PRE_E_RANGE;
    const long long nseg = (X * Y + JP_SEG - 1) / JP_SEG, TC = (T + MS_TMAX - 1) / MS_TMAX, total = TC * nseg;
    if (n > 0x7fffffff || T > 0x7fffffff || X > 0x7fffffff || Y > 0x7fffffff) return PRE_E_SHAPE;
    // the work list lives in LDS, next to 136 bytes of static state: up to what a workgroup of this device may hold
    // (gfx950: 160 KiB - a whole-T slab of 62 planes x 512 x 512, the strong-scaling C3 shard, has 16384 segments);
    // above 64 KiB the kernel has to be told
    const int lds_max = pruned_lds_max();
    if (total * 4 + 256 > lds_max) return PRE_E_UNSUPPORTED;
    if (total * 4 + 256 > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(joint_score_pruned_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_max - 256) != hipSuccess) {
            (void)hipGetLastError();
            return PRE_E_UNSUPPORTED;
        }
    }
    // a block per sample: few samples get more waves each to work through their lists
    // (many samples with short lists: a wave per sample, four samples per block - see the kernel)
    if (n >= 4096 && total <= 2048) {
        hipLaunchKernelGGL(joint_score_pruned_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)total * 16,
                           as_stream(stream), res, (long long)row_stride, mod, segmax, segmin, (int)n, (int)T, (int)X, (int)Y, crop_x,
                           crop_y, (int)nseg, (int)total, scores, flags, stats);
        PRE_LAUNCH_CHECK();
        return PRE_OK;
    }
    const int threads = n >= 2048 ? 256 : n >= 512 ? 512 : 1024;
    hipLaunchKernelGGL(joint_score_pruned_kernel<false>, dim3((unsigned)n), dim3(threads), (size_t)total * 4, as_stream(stream), res,
                       (long long)row_stride, mod, segmax, segmin, (int)n, (int)T, (int)X, (int)Y, crop_x, crop_y, (int)nseg, (int)total,
                       scores, flags, stats);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_kth_f32(const float *scores, int64_t N, const int64_t *ks, int nk, float *out, void *stream)
{
    if (!scores || !ks || !out || N <= 0 || nk <= 0) return PRE_E_NULL;
    for (int j = 0; j < nk; ++j)
        if (ks[j] < 0 || ks[j] >= N) return PRE_E_RANGE;
    for (int j0 = 0; j0 < nk; j0 += 16) {
        KList kl;
        kl.nk = (nk - j0) < 16 ? (nk - j0) : 16;
        for (int j = 0; j < kl.nk; ++j) kl.k[j] = ks[j0 + j];
        hipLaunchKernelGGL(kth_kernel, dim3(kl.nk), dim3(1024), 0, as_stream(stream), scores, (long long)N, kl, out + j0);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

int pre_cov_count_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M, int per_sample_bounds,
                      unsigned long long *count, void *stream)
{
    if (!y || !lo || !hi || !count || n <= 0 || M <= 0) return PRE_E_NULL;
    if (M % 4 == 0 && !(((uintptr_t)y | (uintptr_t)lo | (uintptr_t)hi) & 15) && n <= 0x7fffffff) {
        const long long M4 = M / 4, bx = (M4 + 255) / 256;
        // sample groups: enough workgroups to fill the chip (~4 k), rows in multiples of 8 (the unrolled trip)
        long long groups = (4096 + bx - 1) / bx;
        long long rows_per = (n + groups - 1) / groups;
        rows_per = (rows_per + 7) / 8 * 8;
        groups = (n + rows_per - 1) / rows_per;
        if (bx <= 0x7fffffffLL && groups <= 65535) {
            hipLaunchKernelGGL(cov_count4_kernel, dim3((unsigned)bx, (unsigned)groups), dim3(256), 0, as_stream(stream),
                               reinterpret_cast<const float4 *>(y), reinterpret_cast<const float4 *>(lo),
                               reinterpret_cast<const float4 *>(hi), (int)n, M4, per_sample_bounds, (int)rows_per, count);
            PRE_LAUNCH_CHECK();
            return PRE_OK;
        }
    }
    hipLaunchKernelGGL(cov_count_kernel, dim3(grid_for(n * M, 256)), dim3(256), 0, as_stream(stream), y, lo, hi,
                       (long long)(n * M), (long long)M, per_sample_bounds, count);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_cov_rowcount_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M, int per_sample_bounds,
                         int outside, uint32_t *counts, void *stream)
{
    if (!y || !lo || !hi || !counts || n <= 0 || M <= 0) return PRE_E_NULL;
    const unsigned gx = grid_for(M, 256, 64);
    for (int64_t s0 = 0; s0 < n; s0 += 65535) {
        const int64_t ns = (n - s0) < 65535 ? (n - s0) : 65535;
        hipLaunchKernelGGL(cov_rowcount_kernel, dim3(gx, (unsigned)ns), dim3(256), 0, as_stream(stream), y + s0 * M,
                           per_sample_bounds ? lo + s0 * M : lo, per_sample_bounds ? hi + s0 * M : hi, (long long)M,
                           per_sample_bounds, outside, counts + s0);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

int pre_cov_joint_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M, int per_sample_bounds,
                      uint8_t *inside, void *stream)
{
    if (!y || !lo || !hi || !inside || n <= 0 || M <= 0) return PRE_E_NULL;
    const unsigned gx = grid_for(M, 256, 64);
    for (int64_t s0 = 0; s0 < n; s0 += 65535) {
        const int64_t ns = (n - s0) < 65535 ? (n - s0) : 65535;
        hipLaunchKernelGGL(cov_joint_kernel, dim3(gx, (unsigned)ns), dim3(256), 0, as_stream(stream), y + s0 * M,
                           per_sample_bounds ? lo + s0 * M : lo, per_sample_bounds ? hi + s0 * M : hi, (long long)M,
                           per_sample_bounds, inside + s0);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

}  // extern "C"
