// calib.hip - nonconformity scores, modulation, joint score, radix-select quantiles and
// coverage for gfx950.  All kernels are HBM-bound streaming passes (4 B per element per
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
__global__ void __launch_bounds__(256) absdiff_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      float *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = !(((uintptr_t)a | (uintptr_t)out | (uintptr_t)(b ? b : a)) & 15);
    long long done = 0;
    if (vec) {
        const long long n4 = n / 4;
        const float4 *a4 = reinterpret_cast<const float4 *>(a);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        float4 *o4 = reinterpret_cast<float4 *>(out);
        for (long long i = tid; i < n4; i += stride) {
            float4 v = a4[i];
            if (b) { const float4 w = b4[i]; v = make_float4(v.x - w.x, v.y - w.y, v.z - w.z, v.w - w.w); }
            o4[i] = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
        }
        done = n4 * 4;
    }
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
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
        const float v = pb ? pa[(long long)i * M] - pb[(long long)i * M] : pa[(long long)i * M];
        s = s + v;
    }
    const float mean = s / (float)n;
    float acc = 0.f;
#pragma unroll 8
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
// NaN: np.max propagates it.  A NaN residual or modulation fails `av <= thr*sv` and reaches the divide; 0/0 (a
// constant cell: modulation 0 and residual 0) reaches it through `sv == 0`; a NaN quotient sets the sticky flag and
// the sample's score becomes NaN.
__device__ __forceinline__ void js_update(float av, float sv, float &m, float &thr, bool &nan)
{
    if (!(av <= thr * sv) || sv == 0.f) {
        const float q = av / sv;
        if (q != q) nan = true;
        else if (q > m) { m = q; thr = m * 0.99999905f; }
    }
}

__global__ void __launch_bounds__(256) joint_score_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ mod, int T, int X, int Y,
                                                          int ct, int cx, int cy, int groups, float *__restrict__ scores)
{
    const long long plane = (long long)X * Y, vol = plane * T;
    const int smp = blockIdx.y;
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
    const long long r0 = ((long long)blockIdx.x * groups + gi) * JS_RB;
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
    unsigned int prefix = 0, mask = 0;
    long long rank = kl.k[j];
    if (threadIdx.x == 0) sh_nan = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        for (long long i = threadIdx.x; i < N; i += blockDim.x) {
            const float v = s[i];
            if (shift == 24 && v != v) sh_nan = 1;          // np.quantile: any NaN score makes every quantile NaN
            const unsigned int key = f2key(v);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long cum = 0;
            unsigned int d = 0;
            for (; d < 255; ++d) {
                if (cum + hist[d] > rank) break;
                cum += hist[d];
            }
            sh_prefix = prefix | (d << shift);
            sh_rank = rank - cum;
        }
        __syncthreads();
        prefix = sh_prefix;
        rank = sh_rank;
        mask |= 255u << shift;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = sh_nan ? __uint_as_float(0x7fc00000u) : key2f(prefix);
}

// ------------------------------------------------------------------ per-cell k-th over axis 0
// MSD radix select over [n, M] for ALL requested ranks at once, ONE launch: histogram sweeps of 8 + 6 + 6 + 6 + 6
// bits until every (cell, rank) of the tile has <= KA_CAP elements left under its prefix, then one collecting
// sweep and a rank count over the survivors (ka_collect) - 3 sweeps on typical data up to n ~ 2000, 4 at
// n = 4096, all 5 histogram sweeps only under heavy ties.  A 1024-thread workgroup owns 64 adjacent cells
// (256 B of every sample row - narrower column tiles lose DRAM efficiency fast: 128 B -> 0.7x, 64 B -> 0.3x,
// tools/exp/colread.hip): lane = cell everywhere.
//   sweep:  one wave = one row, so the 64 LDS atomics of a wave-instruction never hit the same
//           counter.  Counters are 16 bit (n < 65536), cells c and c+32 share a word; histogram row r
//           (= slot*bins + bin) is 32 words.  Per cell there is one histogram per *distinct* prefix among
//           its ranks ("slot"; ranks ascend, so equal prefixes are adjacent); the first sweep has one slot.
//   narrow: thread (cell c = tid & 63, rank j = tid >> 6) owns the state (prefix, residual rank) of
//           its pair in registers and walks the bins of its slot serially - 640 independent walks,
//           no cross-lane traffic (the first sweep's 256 bins are pre-summed in 16 groups by all
//           1024 threads).  New prefixes reach the sweeping waves through the idle histogram memory.
// 80 KiB of LDS: two workgroups per CU (tools/exp/ldsocc.hip), one sweeps while the other narrows.
// The kernel is VALU-bound (a wave64 instruction holds its SIMD16 for 4 cycles), so the work per
// element is what is optimised: 3-instruction key, 3-instruction counter address, match loops sized to the
// slots actually in use.  For small n the tile (n * 256 B) stays L2-resident between sweeps.
constexpr int KA_W = 64, KA_MAXK = 10, KA_WAVES = 16;
constexpr int KA_HIST_WORDS = KA_MAXK * 64 * 32;          // 80 KiB; the first sweep uses 256*32 of them
constexpr int KA_GROUPS_AT = 256 * 32;                    // 16 x 64 group sums of the first sweep live here
constexpr int KA_FLAGS_AT = 1008, KA_CAP = 31;            // per-wave "many survivors" flags (640..1023 is never used otherwise)
struct KAList { int nk; int k[KA_MAXK]; };

// counter of (histogram row, cell): cells c and c+32 share a word (16-bit halves), so the 32 lanes the LDS serves
// per cycle always hit 32 different banks whatever rows they address - no rotation needed, 3 address ops
__device__ __forceinline__ int ka_word(int row, int lane31) { return row * 32 + lane31; }

// My cell's DISTINCT prefixes (published in hist[j*64 + cell] by the previous narrowing), compacted to the
// front: slot i = i-th distinct prefix (ranks ascend, so equal prefixes are adjacent); unused entries hold the
// sentinel 1 (low bit set: never equals a masked key).  lmax = most slots any cell of the tile has (wave-uniform
// loop bound for the match); myslot = slot of this thread's own rank.  Ends with a barrier: hist is free again.
__device__ __forceinline__ void ka_prefixes(unsigned int *hist, int nk, int lane, int wave, unsigned int (&pf)[KA_MAXK],
                                            int &lmax, int &myslot)
{
    unsigned int *scr = hist + 1024 + wave * (KA_MAXK * 64);
    unsigned int prev = 0;
    int L = 0;
#pragma unroll
    for (int j = 0; j < KA_MAXK; ++j) {
        const unsigned int p = j < nk ? hist[j * 64 + lane] : 0u;
        if (j < nk && (j == 0 || p != prev)) { scr[L * 64 + lane] = p; ++L; }
        if (j == wave) myslot = L - 1;
        prev = p;
    }
#pragma unroll
    for (int j = 0; j < KA_MAXK; ++j) pf[j] = j < L ? scr[j * 64 + lane] : 1u;
    int m = L;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    lmax = __builtin_amdgcn_readfirstlane(m);
    __syncthreads();
}

template <int N> struct ka_ic { static constexpr int value = N; };

// slot (1-based) whose prefix equals hi, 0 if none; LM = static bound on the number of slots in use
template <int LM>
__device__ __forceinline__ int ka_match(unsigned int hi, const unsigned int (&pf)[KA_MAXK])
{
    int m = 0;
#pragma unroll
    for (int j = 0; j < LM; ++j) m = (hi == pf[j]) ? j + 1 : m;
    return m;
}

// rows wave, wave+16, ... of my cell, eight loads in flight
template <class F>
__device__ __forceinline__ void ka_sweep(const float *__restrict__ col, bool cok, int n, long long M, int wave, F &&f)
{
    if (!cok) return;
    int i = wave;
    for (; i + 7 * KA_WAVES < n; i += 8 * KA_WAVES) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(long long)(i + u * KA_WAVES) * M];
#pragma unroll
        for (int u = 0; u < 8; ++u) f(v[u]);
    }
    for (; i < n; i += KA_WAVES) f(col[(long long)i * M]);
}

// the match against the cell's prefixes costs 2 VALU instructions per slot and element: instantiate the sweep for
// a few static slot counts and branch (wave-uniformly) on the tile's actual maximum
template <class G>
__device__ __forceinline__ void ka_by_slots(int lmax, G &&g)
{
    if (lmax <= 2) g(ka_ic<2>{});
    else if (lmax <= 4) g(ka_ic<4>{});
    else if (lmax <= 7) g(ka_ic<7>{});
    else g(ka_ic<KA_MAXK>{});
}

// Returns (block-uniform) whether some (cell, rank) of the tile still has more than KA_CAP elements under its
// prefix after this sweep.
template <int BITS, int SLOTS>
__device__ __forceinline__ bool ka_pass(const float *__restrict__ col, bool cok, int n, long long M, int nk, int shift,
                                        unsigned int *hist, unsigned int &myp, unsigned int &myr, int lane, int wave, int tid)
{
    constexpr int NB = 1 << BITS;
    const unsigned int mask = (shift + BITS == 32) ? 0u : ~0u << (shift + BITS);
    const bool state = wave < nk;                 // this thread owns (cell = lane, rank = wave)

    // my cell's DISTINCT prefixes, compacted to the front (slot i = i-th distinct prefix); the
    // sentinel 1 (low bit set) never equals a masked key.  lmax = most slots any cell has: the
    // match loop runs to that wave-uniform bound instead of KA_MAXK.  myslot = slot of my own rank.
    unsigned int pf[KA_MAXK];
    int lmax = 1, myslot = 0;
    if (SLOTS > 1) ka_prefixes(hist, nk, lane, wave, pf, lmax, myslot);
    for (int i = tid; i < SLOTS * NB * 32; i += 1024) hist[i] = 0u;
    __syncthreads();

    const unsigned int inc = 1u << (16 * (lane >> 5));
    const int half = lane & 31, sh16 = 16 * (lane >> 5);
    if (SLOTS == 1) {
        ka_sweep(col, cok, n, M, wave, [&](float v) __attribute__((always_inline)) {
            atomicAdd(&hist[ka_word((int)((f2key(v) >> shift) & (NB - 1)), half)], inc);
        });
    } else {
        ka_by_slots(lmax, [&](auto lm) __attribute__((always_inline)) {
            ka_sweep(col, cok, n, M, wave, [&](float v) __attribute__((always_inline)) {
                const unsigned int key = f2key(v);
                const int m = ka_match<decltype(lm)::value>(key & mask, pf);
                if (m) atomicAdd(&hist[ka_word((m - 1) * NB + (int)((key >> shift) & (NB - 1)), half)], inc);
            });
        });
    }
    __syncthreads();

    // narrow: walk the bins of my slot for my cell until the running count passes my rank
    auto cnt = [&](int row) __attribute__((always_inline)) { return (hist[ka_word(row, half)] >> sh16) & 0xffffu; };
    int bin0 = 0, bin1 = NB;
    unsigned int cum = 0;
    bool many = false;                     // more than KA_CAP elements share my (now longer) prefix
    if (SLOTS == 1) {
        // first sweep (one slot, NB bins): 16 groups of NB/16 bins are summed by all 1024 threads first
        constexpr int GB = NB / KA_WAVES;
        unsigned int gs = 0;
#pragma unroll 8
        for (int u = 0; u < GB; ++u) gs += cnt(wave * GB + u);
        hist[KA_GROUPS_AT + wave * 64 + lane] = gs;
        __syncthreads();
        if (state) {
            // group of my rank = number of groups whose inclusive running count is <= myr (branch-free: the 16
            // LDS reads are independent and pipeline; a `break` loop serialised them behind their latency)
            unsigned int run = 0;
            int g = 0;
#pragma unroll
            for (int u = 0; u < KA_WAVES; ++u) {
                run += hist[KA_GROUPS_AT + u * 64 + lane];
                const bool le = run <= myr;
                g += le;
                cum = le ? run : cum;
            }
            g = min(g, KA_WAVES - 1);
            bin0 = g * GB;
            bin1 = bin0 + GB;
        }
    }
    if (state) {
        // digit = number of bins (from bin0) whose inclusive running count is <= myr; `cum` ends as the count
        // before the chosen bin.  Branch-free for the same reason as above.
        const int base = (SLOTS == 1 ? 0 : myslot) * NB;
        unsigned int run = cum;
        int d = 0;
#pragma unroll 8
        for (int bin = bin0; bin < bin1; ++bin) {
            run += cnt(base + bin);
            const bool le = run <= myr;
            d += le;
            cum = le ? run : cum;
        }
        const int digit = bin0 + min(d, bin1 - bin0 - 1);
        many = cnt(base + digit) > (unsigned)KA_CAP;
        myp |= (unsigned)digit << shift;
        myr -= cum;
    }
    const bool wmany = __ballot(many) != 0;
    __syncthreads();                       // everyone is done reading the histograms
    if (shift > 0 && state) hist[wave * 64 + lane] = myp;     // publish for the next sweep's matching
    if (lane == 0) hist[KA_FLAGS_AT + wave] = wmany;
    __syncthreads();
    return __ballot(hist[KA_FLAGS_AT + (lane & (KA_WAVES - 1))] != 0u) != 0;
}

// Finish without further histogram sweeps once every (cell, rank) of the tile has at most KA_CAP elements left
// under its prefix: ONE more sweep appends each surviving key to the list of its slot - the idle histogram
// memory, list[slot][i][cell] with the fill counter in row KA_CAP - and the owner thread picks its rank from
// the <= KA_CAP candidates by counting (k-th smallest = the smallest key with more than k keys <= it).
// Typical data needs 9 + 6 known bits for that (two histogram sweeps + this one instead of five); heavy
// ties never get there and take all five histogram sweeps.
__device__ __forceinline__ int ka_list(int slot, int i, int cell) { return (slot * 32 + i) * 64 + cell; }

__device__ __forceinline__ void ka_collect(const float *__restrict__ col, bool cok, int n, long long M, int nk, int known,
                                           unsigned int *hist, unsigned int &myp, unsigned int myr, int lane, int wave, int tid)
{
    const unsigned int mask = ~0u << known;
    unsigned int pf[KA_MAXK];
    int lmax = 1, myslot = 0;
    ka_prefixes(hist, nk, lane, wave, pf, lmax, myslot);
    for (int i = tid; i < KA_HIST_WORDS; i += 1024) hist[i] = ((i >> 6) & 31) == KA_CAP ? 0u : 0xffffffffu;   // counters / sentinels
    __syncthreads();

    ka_by_slots(lmax, [&](auto lm) __attribute__((always_inline)) {
        ka_sweep(col, cok, n, M, wave, [&](float v) __attribute__((always_inline)) {
            const unsigned int key = f2key(v);
            const int m = ka_match<decltype(lm)::value>(key & mask, pf);
            if (m) {
                const unsigned int pos = atomicAdd(&hist[ka_list(m - 1, KA_CAP, lane)], 1u);
                hist[ka_list(m - 1, (int)pos, lane)] = key;      // pos < KA_CAP: the histogram counted these elements
            }
        });
    });
    __syncthreads();

    if (wave < nk) {
        const int c = (int)hist[ka_list(myslot, KA_CAP, lane)];
        int cmax = c;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
        cmax = __builtin_amdgcn_readfirstlane(cmax);
        unsigned int ans = 0xffffffffu;
        for (int i = 0; i < cmax; ++i) {
            const unsigned int ki = hist[ka_list(myslot, i, lane)];          // sentinel beyond my own count
            unsigned int le = 0;
            for (int j = 0; j < cmax; ++j) le += hist[ka_list(myslot, j, lane)] <= ki;
            if (i < c && le > myr) ans = min(ans, ki);
        }
        myp = ans;
    }
}

__global__ void __launch_bounds__(1024, 8) kth_axis0_kernel(const float *__restrict__ s, int n, long long M, long long tile0,
                                                            const KAList kl, float *__restrict__ out)
{
    __shared__ unsigned int hist[KA_HIST_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = kl.nk;
    const long long c0 = (tile0 + blockIdx.x) * KA_W, c = c0 + lane;
    const bool cok = c < M;
    const float *col = s + c;
    const bool state = wave < nk;
    unsigned int myp = 0u, myr = state ? (unsigned)kl.k[wave] : 0u;
    // 8 + 6 + 6 + 6 + 6 bits; stop counting as soon as the survivors fit the lists
    int known = 24;                        // lowest known bit so far
    bool many = ka_pass<8, 1>(col, cok, n, M, nk, 24, hist, myp, myr, lane, wave, tid);
#pragma unroll 1
    for (int shift = 18; many && shift >= 0; shift -= 6) {
        many = ka_pass<6, KA_MAXK>(col, cok, n, M, nk, shift, hist, myp, myr, lane, wave, tid);
        known = shift;
    }
    if (known > 0) ka_collect(col, cok, n, M, nk, known, hist, myp, myr, lane, wave, tid);   // else all 32 bits are counted
    if (state && cok) out[(long long)wave * M + c] = key2f(myp);
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

// grid (chunks, n): clear inside[i] if any cell of sample i leaves [lo, hi]
__global__ void __launch_bounds__(256) cov_joint_kernel(const float *__restrict__ y, const float *__restrict__ lo,
                                                        const float *__restrict__ hi, long long M, int per_sample,
                                                        uint8_t *inside)
{
    const int smp = blockIdx.y;
    const float *py = y + smp * M;
    const float *plo = lo + (per_sample ? smp * M : 0), *phi = hi + (per_sample ? smp * M : 0);
    bool bad = false;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < M; c += (long long)gridDim.x * blockDim.x) {
        const float v = py[c];
        bad |= !((v >= plo[c]) && (v <= phi[c]));
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
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < M; c += (long long)gridDim.x * blockDim.x) {
        const float v = py[c];
        local += outside ? ((v <= plo[c]) || (v >= phi[c])) : ((v >= plo[c]) && (v <= phi[c]));
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
    hipLaunchKernelGGL(absdiff_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), a, b, out, (long long)n);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_std_axis0_f32(const float *a, const float *b, int64_t n, int64_t M, float eps, float *mod, void *stream)
{
    if (!a || !mod || n <= 0 || M <= 0) return PRE_E_NULL;
    if (n > 0x7fffffff || M > 0xffffff00LL) return PRE_E_SHAPE;        // one thread per cell, 32-bit work-item count
    hipLaunchKernelGGL(std_axis0_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, (int)n,
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

int pre_joint_score_f32(const float *a, const float *b, const float *mod, int64_t n, int64_t T, int64_t X, int64_t Y,
                        int crop_t, int crop_x, int crop_y, float *scores, void *stream)
{
    if (!a || !mod || !scores || n <= 0 || T <= 0 || X <= 0 || Y <= 0) return PRE_E_NULL;
    if (crop_t < 0 || crop_x < 0 || crop_y < 0) return PRE_E_RANGE;
    if (n > 65535 * 1024LL || T * X > 0x7fffffffLL * JS_RB || Y > 0x7fffffff) return PRE_E_SHAPE;
    // spans of JS_RB rows; a block takes as many as give it >= ~8 k cells (but leaves >= ~2 k blocks when possible)
    const long long spans = (T * X + JS_RB - 1) / JS_RB;
    long long groups = 8192 / (JS_RB * Y);
    if (groups < 1) groups = 1;
    while (groups > 1 && (spans / groups) * n < 2048) groups /= 2;
    const long long chunks = (spans + groups - 1) / groups;
    // gridDim.y is limited to 65535: walk the batch axis in slices
    for (int64_t s0 = 0; s0 < n; s0 += 65535) {
        const int64_t ns = (n - s0) < 65535 ? (n - s0) : 65535;
        const long long vol = (long long)T * X * Y;
        hipLaunchKernelGGL(joint_score_kernel, dim3((unsigned)chunks, (unsigned)ns), dim3(256), 0, as_stream(stream),
                           a + s0 * vol, b ? b + s0 * vol : nullptr, mod, (int)T, (int)X, (int)Y, crop_t, crop_x, crop_y,
                           (int)groups, scores + s0);
        PRE_LAUNCH_CHECK();
    }
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

int pre_kth_axis0_f32(const float *scores, int64_t n, int64_t M, const int32_t *ks, int nk, float *out, void *stream)
{
    if (!scores || !ks || !out || n <= 0 || M <= 0 || nk <= 0) return PRE_E_NULL;
    if (n >= 65536 || nk > 64) return PRE_E_SHAPE;
    for (int j = 0; j < nk; ++j) {
        if (ks[j] < 0 || ks[j] >= n) return PRE_E_RANGE;
        if (j > 0 && ks[j] < ks[j - 1]) return PRE_E_RANGE;      // ascending (slots rely on it)
    }
    const long long tiles = (M + KA_W - 1) / KA_W;
    const long long per_launch = 1LL << 21;                     // x 1024 threads: the dispatch packet counts work-items in 32 bits
    for (int j0 = 0; j0 < nk; j0 += KA_MAXK) {
        KAList kl;
        kl.nk = (nk - j0) < KA_MAXK ? (nk - j0) : KA_MAXK;
        for (int j = 0; j < kl.nk; ++j) kl.k[j] = ks[j0 + j];
        for (long long t0 = 0; t0 < tiles; t0 += per_launch) {
            const long long nt = tiles - t0 < per_launch ? tiles - t0 : per_launch;
            hipLaunchKernelGGL(kth_axis0_kernel, dim3((unsigned)nt), dim3(1024), 0, as_stream(stream), scores, (int)n,
                               (long long)M, t0, kl, out + (long long)j0 * M);
            PRE_LAUNCH_CHECK();
        }
    }
    return PRE_OK;
}

int pre_cov_count_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M, int per_sample_bounds,
                      unsigned long long *count, void *stream)
{
    if (!y || !lo || !hi || !count || n <= 0 || M <= 0) return PRE_E_NULL;
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
