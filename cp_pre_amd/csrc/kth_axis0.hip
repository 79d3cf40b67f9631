// kth_axis0.hip - per-cell order statistics over the batch axis of [n, M] scores (marginal q-hat:
// calibrate(scores, n, alpha) with 2-D+ scores, call sites Marginal/Wave_Residuals_CP.py:288,
// Marginal/NS_Residuals_CP.py:310), all requested ranks of a cell at once, ONE launch, no workspace.
//
// Bound: HBM.  Algorithmic traffic 4 B per element per sweep; the point of the design is FEW sweeps.
//
// Regimes by n (pre_kth_axis0_planes_f32 at the end of the file; every one exact, one launch, no workspace):
//   n <= 168          kth_small_kernel: a cell's column in its lane's registers, Batcher network pruned to its rows, one read;
//   168 < n <= 240    kth_pair_kernel: two lanes per cell, half the column each, one cross-lane merge step, one read;
//   .. <= 1024        kth_tile_kernel: a persistent workgroup keeps the 64-cell tile in registers (16 .. 64 rows per
//                     thread), exact window, every sweep out of registers, next tile prefetched in place: one read;
//   1024 < n <= 2048  the same on 32-cell tiles, two rows per load: one read;
//   n > 2048          kth_axis0_kernel, the STREAMING form described next: sample + 2 sweeps on typical data
//                     (to n = 4096 with the register tiles' bookkeeping - tags in the histogram words, pooled lists:
//                     ka_fast_tags; 47-entry lists and a bitmap to 9216; the general radix form beyond).
//
// A 1024-thread workgroup owns 64 adjacent cells (256 B of every sample row - narrower column tiles lose DRAM
// efficiency fast: 128 B -> 0.7x, 64 B -> 0.3x, tools/exp/colread.hip).  lane = cell everywhere, one wave = one
// row, so the 64 LDS atomics of a wave-instruction never hit the same counter.
//
// MSD radix select on kk = key - klo[cell]  (key = order-preserving uint32 image of the fp32 score):
//   0. SAMPLE (<= 1/16 sweep): 256 evenly spaced rows (64, with the range widened by half on each side, for
//      n <= 1024: a tile the register form left to this one) give each cell the window [klo, khi] of its sample and
//      its own shift s = the smallest with (khi - klo) >> s <= NB1 - 2 (per cell: one tile-wide shift would be set
//      by the cell whose sample happens to hold the smallest value, 15 binades instead of 9 on |N(0,1)| scores).
//   1. FIRST DIGIT (1 sweep, one slot): NB1 = 512 or 1024 buckets over the cell's own window, + an underflow
//      counter (key < klo) and an overflow bucket.  A fixed top-byte digit wastes its bins on exponents the data
//      never takes; the window puts all of them where the cell's scores are (|N(0,1)|-like scores: ~16 known
//      bits instead of 8).
//      FAST form (n up to ~6 NB1): bucket = floor(fma(v - vlo, sf, 1)) in fp32, sf = (NB1-1)/(vhi - vlo): buckets of
//      equal width in the VALUE, all NB1 - 1 of them inside the window.  The key is log-like in the value: equal
//      key-widths spend most buckets on the sparse small values and give a bell-shaped score distribution twice
//      the peak occupancy (measured: 1023 key-linear buckets left ~half the tiles of n = 4096 |N(0,1)| scores with
//      a pair above CAP; value-linear ones have a peak mean occupancy of 9).  Any map that is monotone in the
//      key order keeps the select exact (an fp32 fma with a positive factor, a clamp and floor are; a NaN, whose key
//      is the largest, would not be, so a NaN in the tile sends it to the general form); if every pair then has
//      <= CAP elements in its bucket, step 3 follows directly: elements are matched to pairs by bucket number
//      - a per-cell bitmap of target rows in LDS answers "is this element wanted?" with one conflict-free read,
//      and only the ~1 % that are pay for the search of their slot: 9 VALU instructions per element instead of
//      28 - and the rank is picked among their keys.  Otherwise (ties, heavy
//      tails, non-finite window):
//      GENERAL form: buckets of width 2^s (digit = kk >> s), so that later digits are bit fields of kk.
//      If some rank falls outside its window (extreme ranks, unrepresentative sample) the tile repeats this step
//      with klo = 0 and the plain top-9/10-bit digit: exactness never depends on the sample.
//   2. while some (cell, rank) still has more than CAP = 31 (29) elements under its prefix: one more
//      histogram sweep of 6 (5) bits per DISTINCT prefix of the cell ("slot"; ranks ascend, so equal
//      prefixes are adjacent), counters for <= 10 slots.
//   3. COLLECT (1 sweep): each surviving element is appended to the list of its slot (the idle histogram
//      memory) and the owner thread picks its rank among the <= 31 candidates by counting.
// Typical data: sample + 2 sweeps (8.25 B per element) up to n ~ 2000 with 512 buckets (80 KiB LDS, two
// workgroups per CU) and up to n ~ 6000 with 1024 buckets (133 KiB, one workgroup per CU); beyond that the
// general form, sample + 3 sweeps; round 1's fixed 8+6+6+6+6-bit digits needed 3 and 4 sweeps there.  Heavy ties never get under CAP and take every
// histogram sweep (exact either way).
//
// Counters: 16 bit, cells c and c+32 sharing a word (the 32 lanes the LDS serves per cycle hit 32 different
// banks whatever rows they address), valid for n < 65536; n >= 65536 (BASELINE C5: 65536 samples) runs the
// WIDE instantiation with one 32-bit counter per cell (512 first-digit buckets, 5-bit later digits).
//
// narrow: thread (cell = tid & 63, rank j = tid >> 6) owns the state (prefix, residual rank) of its pair in
// registers and walks the bins of its slot - 640 independent walks, no cross-lane traffic, written branch-free
// (digit = number of bins whose running count is <= the rank) so the LDS reads pipeline.  New prefixes reach the
// sweeping waves through the idle histogram memory.
#include "common.h"

namespace {

#ifndef KA_U_A
#define KA_U_A 8
#endif
#ifndef KA_U_B
#define KA_U_B 16
#endif
#ifndef KA_REV
#define KA_REV 1                // the collect sweep of the tag forms runs from the tile's last rows to its first (ka_sweep_h_rev)
#endif
#ifndef KA_KEEP
#define KA_KEEP 80
#endif
#ifndef KA_KEEP_BIG
#define KA_KEEP_BIG 80        // rows kept by the 47-entry-list instantiation
#endif
constexpr int KA_W = 64, KA_MAXK = 10, KA_WAVES = 16, KA_MAPROW = 68;
constexpr int KA_TAGS_MAX_N = 4096;                       // rows up to which the list tags fit above a 12-bit count (ka_fast_tags<.., false>)
constexpr int KA_LATE_WORDS = KA_MAXK * 64 * 32;          // later sweeps (and up to 31-entry collect lists): 80 KiB
constexpr int KA_FLAGS_AT = 1008;                         // per-wave flags (words 640..1023 are never used otherwise)
struct KAList { int nk; int k[KA_MAXK]; int o[KA_MAXK]; };    // ranks (ascending) and the output row of each
// A launch selects `planes` independent [n, M] score matrices (plane p at s + p * PS, its result rows at out + p * OPS, the
// rows of a result OS apart): the tiles of all planes form ONE grid, tile t = (plane t / tpp, cells 64 (t % tpp) ...), so a
// driver that holds a time-major residual slab (pipeline.time_major: [T][n][plane]) selects all its planes in one launch
// instead of T launches that each end on a ragged round of workgroups.
// (sp, sc: a persistent grid's step from one of its tiles to the next, grid = sp * tpp + sc, so that the loop advances
// (plane, tile in plane) by additions: a 64-bit division per tile costs the register-tile kernels ~100 VALU instructions
// of the ~1100 a tile of n = 130..512 rows takes - measured 8-11 %)
struct KAPlanes { long long tpp, PS, OS, OPS; long long sp, sc; };
__device__ __forceinline__ void ka_locate(const KAPlanes &pl, long long tile, long long &plane, long long &c0, int W = KA_W)
{
    long long t;                           // tile within its plane (wave-uniform)
    if (tile <= 0xffffffffLL) {            // tpp < 2^31 (host check): 32-bit division whenever the tile index allows
        const unsigned int q = (unsigned int)tile / (unsigned int)pl.tpp;
        plane = q;
        t = (long long)((unsigned int)tile - q * (unsigned int)pl.tpp);
    } else {
        plane = tile / pl.tpp;
        t = tile - plane * pl.tpp;
    }
    c0 = t * W;
}
// the tile one grid step behind (plane, c0)
__device__ __forceinline__ void ka_advance(const KAPlanes &pl, long long &plane, long long &c0, int W = KA_W)
{
    plane += pl.sp;
    c0 += pl.sc * W;
    if (c0 >= pl.tpp * W) { c0 -= pl.tpp * W; ++plane; }
}

template <bool WIDE> struct Ctr {
    static constexpr int CW = WIDE ? 64 : 32;             // words per histogram row
    static __device__ __forceinline__ int word(int row, int lane) { return WIDE ? row * 64 + lane : row * 32 + (lane & 31); }
    static __device__ __forceinline__ unsigned int inc(int lane) { return WIDE ? 1u : 1u << (16 * (lane >> 5)); }
    static __device__ __forceinline__ unsigned int get(const unsigned int *hist, int row, int lane)
    {
        return WIDE ? hist[row * 64 + lane] : (hist[row * 32 + (lane & 31)] >> (16 * (lane >> 5))) & 0xffffu;
    }
};

// LSX > 0: lists of LSX - 1 entries instead of 31 - room for them by the bitmap form of the table even with one workgroup
// per CU.  <10, false, 48> serves 4096 < n <= 9216: at n = 8192 (a C4 / C5 calibration set per cell once the shards of
// 8 ranks meet) the 1024 value-linear buckets hold 8 elements on average and ~22 where a bell-shaped score distribution
// is densest - over the 31-entry cap in some (cell, rank) of most tiles, which used to send the whole range to the
// general form's sample + 3-4 sweeps; 47 entries hold them (P(Poisson(22) > 47) ~ 1e-6) and the tile is done in sample +
// 2 sweeps, minus the rows kept in registers.
template <int LOG_NB1, bool WIDE, int LSX = 0>
struct KACfg {
    static constexpr int NB1 = 1 << LOG_NB1, CW = WIDE ? 64 : 32;
    static constexpr int FIRST_WORDS = (NB1 + 1) * CW + KA_WAVES * 64;       // first-digit histogram + group sums
    static constexpr int WG_PER_CU = 2 * (FIRST_WORDS > KA_LATE_WORDS ? FIRST_WORDS : KA_LATE_WORDS) * 4 <= 160 * 1024 ? 2 : 1;
    // collect lists list[slot][i][cell], i < CAP, fill counter in row CAP; behind them the fast form's table "which
    // slot, if any, wants this row of this cell":
    //   two workgroups per CU (both must fit 80 KiB; 27-entry lists): a bitmap, word [row >> 4][cell], bit (row & 15)
    //     = "a target row", bits 16.. = the slot of the word's first target row;
    //   one workgroup per CU (31-entry lists): room for a byte per (row, cell) holding slot + 1 - five instructions
    //     less per element (measured at n = 4096: 3.60 vs 3.77 ms) at the price of 2-way bank conflicts inside a quad
    //     of lanes, which nothing waits for.
    static constexpr bool BYTEMAP = WG_PER_CU == 1 && LSX == 0;
    // (byte map rows are KA_MAPROW = 68 bytes apart, not 64: with 16-word rows the 32 lanes an LDS cycle serves fall on
    // 16 of the 32 banks whatever rows they look up - measured 34 % of the LDS cycles lost to conflicts; 17 words spread them)
    static constexpr int BM_WORDS = BYTEMAP ? (NB1 + 1) * (KA_MAPROW / 4) : ((NB1 + 1 + 15) / 16) * 64;
    static constexpr int LS = LSX > 0 ? LSX : WG_PER_CU == 2 ? 28 : 32, CAP = LS - 1;
    static constexpr int BM_AT = KA_MAXK * LS * 64;
    static constexpr int COLLECT_WORDS = BM_AT + BM_WORDS;
    static constexpr int W1 = FIRST_WORDS > COLLECT_WORDS ? FIRST_WORDS : COLLECT_WORDS;
    // TAGS (round 5, n <= 4096 with 1024 buckets): the fast form keeps "which list wants this row" in the histogram words
    // and its lists in a pool behind them (ka_fast_tags below) - it takes all 160 KiB the CU has
    static constexpr bool TAGS = LOG_NB1 == 10 && !WIDE && LSX == 0;
    // TAGS_ARR (4096 < n <= 12288): the same with the tags in an array of their own (ka_fast_tags<.., true>): 151 KB
    static constexpr bool TAGS_ARR = LOG_NB1 == 10 && !WIDE && LSX == 48;
    static constexpr int WORDS0 = W1 > KA_LATE_WORDS ? W1 : KA_LATE_WORDS;
    static constexpr int WORDS = TAGS ? 40960 : TAGS_ARR ? 38400 : WORDS0;
    static constexpr int U = WG_PER_CU == 2 ? KA_U_A : KA_U_B;             // rows per batch of loads
    static constexpr int BITS = WIDE ? 5 : 6;                               // 10 slots x 2^BITS bins x CW words = 80 KiB
    static_assert(WORDS * 4 * WG_PER_CU <= 160 * 1024, "does not fit the 160 KiB LDS");
};

// elements below the cell's window get all ones: above every valid prefix, so they never match one
__device__ __forceinline__ unsigned int ka_kk(float v, unsigned int klo)
{
    const unsigned int key = f2key(v);
    return key >= klo ? key - klo : 0xffffffffu;
}

// My cell's DISTINCT prefixes (published in hist[j*64 + cell] by the previous narrowing), compacted to the
// front: slot i = i-th distinct prefix; unused entries hold the sentinel 1.  lmax = most slots any cell of the tile
// has (wave-uniform loop bound for the match); myslot = slot of this thread's own rank.  Ends with a barrier: hist
// is free again.
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

// slot (1-based) whose prefix equals hi, 0 if none; LM = static bound on the number of slots in use.  The lowest
// slot wins: the sentinel of the unused slots (1) can equal a legitimate value (a fully known kk, bucket row 1),
// and the real slots come first; a match with an unused slot alone only touches that slot's idle counters.
template <int LM>
__device__ __forceinline__ int ka_match(unsigned int hi, const unsigned int (&pf)[KA_MAXK])
{
    int m = 0;
#pragma unroll
    for (int j = LM - 1; j >= 0; --j) m = (hi == pf[j]) ? j + 1 : m;
    return m;
}

// one row of the tile: 64 consecutive floats at the wave-uniform address `p`, lane l reading p[l] (byte offset loff)
__device__ __forceinline__ float ka_row(const float *p, int loff)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 256, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, 0, 0));
}

// rows wave, wave+16, ... of my cell, U loads in flight per lane.  `col` = the tile's first cell and `wave` are
// wave-uniform (readfirstlane'd by the kernel): each row is read through a buffer descriptor whose base is advanced
// by scalar adds, and every load uses the SAME per-lane offset register (per-lane 64-bit addresses cost 2 registers
// per load in flight and spilled).  (Issuing the next batch before processing the current one was tried: no gain,
// the sweeps are bound by VALU issue, not by loads in flight.)
// (`first`: start at this thread's row number `first`, i.e. skip rows wave, wave+16, ... wave+16(first-1))
template <int U, class F>
__device__ __forceinline__ void ka_sweep(const float *__restrict__ col, bool cok, int n, long long M, int wave, F &&f, int first = 0)
{
    if (!cok) return;
    constexpr int STEP = U * KA_WAVES, SPAN = (U - 1) * KA_WAVES;
    const int loff = (int)(threadIdx.x & 63u) * 4;
    const long long stride = (long long)KA_WAVES * M;
    int i = wave + first * KA_WAVES;
    const float *p = col + (long long)i * M;
    for (; i + SPAN < n; i += STEP) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = ka_row(p, loff); p += stride; }
#pragma unroll
        for (int u = 0; u < U; ++u) f(v[u]);
    }
    // the last, partial batch: its loads are issued together too (rows beyond n through an empty descriptor) and the rows
    // below n processed under their wave-uniform tests.  (Row by row - load, wait, process - the up to U - 1 rows of the
    // tail were U - 1 serialised HBM latencies per sweep: a quarter of a tile's time at n = 3000, where 11 rows per thread
    // are left over; n = 3000 2.24 -> 2.9 TB/s.)
    if (i < n) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const __amdgpu_buffer_rsrc_t r =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, i + u * KA_WAVES < n ? 256 : 0, 0x00020000);
            v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, 0, 0));
            p += stride;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * KA_WAVES < n) f(v[u]);
    }
}

// the same sweep handing the functor H rows at a time: g(v, nv) with v[0 .. nv) inside the column (nv wave-uniform)
template <int U, int H, class G>
__device__ __forceinline__ void ka_sweep_h(const float *__restrict__ col, bool cok, int n, long long M, int wave, G &&g, int first = 0)
{
    static_assert(U % H == 0, "half batches");
    if (!cok) return;
    constexpr int STEP = U * KA_WAVES, SPAN = (U - 1) * KA_WAVES;
    const int loff = (int)(threadIdx.x & 63u) * 4;
    const long long stride = (long long)KA_WAVES * M;
    int i = wave + first * KA_WAVES;
    const float *p = col + (long long)i * M;
    for (; i + SPAN < n; i += STEP) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = ka_row(p, loff); p += stride; }
#pragma unroll
        for (int h = 0; h < U; h += H) g(*reinterpret_cast<float(*)[H]>(&v[h]), H);
    }
    if (i < n) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const __amdgpu_buffer_rsrc_t r =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, i + u * KA_WAVES < n ? 256 : 0, 0x00020000);
            v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, 0, 0));
            p += stride;
        }
        const int left = (n - i + KA_WAVES - 1) / KA_WAVES;       // rows of mine in this batch (wave-uniform)
#pragma unroll
        for (int h = 0; h < U; h += H)
            if (h < left) g(*reinterpret_cast<float(*)[H]>(&v[h]), left - h < H ? left - h : H);
    }
}

// ... and the same BACKWARDS: the partial last batch first, then the full batches from the highest rows down to `first`.
// The collect sweep of the two-sweep forms runs this way (round 6): the histogram sweep has just read the tile's rows in
// ascending order, so its LAST rows are the ones most recently brought on die - with one 1-2 MB tile per CU in flight,
// half a gigabyte chip-wide, the 256 MiB Infinity Cache still holds the rows at most ~2000 loads back when the second
// sweep starts, and none of the rows the forward order asked for first (they are the oldest).  The rows a thread keeps in
// registers between the sweeps are its FIRST rows, the coldest ones.  Which elements join which list does not depend on
// the order (the lists are sorted afterwards).
template <int U, int H, class G>
__device__ __forceinline__ void ka_sweep_h_rev(const float *__restrict__ col, bool cok, int n, long long M, int wave, G &&g, int first = 0)
{
    static_assert(U % H == 0, "half batches");
    if (!cok) return;
    constexpr int STEP = U * KA_WAVES, SPAN = (U - 1) * KA_WAVES;
    const int loff = (int)(threadIdx.x & 63u) * 4;
    const long long stride = (long long)KA_WAVES * M;
    const int i0 = wave + first * KA_WAVES;
    const int nb = n - SPAN - i0 > 0 ? (n - SPAN - i0 + STEP - 1) / STEP : 0;       // full batches (wave-uniform)
    {
        const int i = i0 + nb * STEP;
        if (i < n) {
            const float *p = col + (long long)i * M;
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const __amdgpu_buffer_rsrc_t r =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, i + u * KA_WAVES < n ? 256 : 0, 0x00020000);
                v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, 0, 0));
                p += stride;
            }
            const int left = (n - i + KA_WAVES - 1) / KA_WAVES;       // rows of mine in this batch (wave-uniform)
#pragma unroll
            for (int h = 0; h < U; h += H)
                if (h < left) g(*reinterpret_cast<float(*)[H]>(&v[h]), left - h < H ? left - h : H);
        }
    }
    for (int j = nb - 1; j >= 0; --j) {
        const float *p = col + (long long)(i0 + j * STEP) * M;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ka_row(p + (long long)(U - 1 - u) * stride, loff);      // (highest row first)
#pragma unroll
        for (int h = 0; h < U; h += H) g(*reinterpret_cast<float(*)[H]>(&v[h]), H);
    }
}

// where a sweep takes the tile's elements from: memory (every sweep re-reads the column tile)
struct HbmSrc {
    const float *col;
    bool cok;
    int n;
    long long M;
    int wave;
    template <int U, class F> __device__ __forceinline__ void sweep(F &&f) const { ka_sweep<U>(col, cok, n, M, wave, f); }
};
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

// block-uniform OR of per-wave flag words (through 16 spare LDS words); ends with a barrier passed by everyone
__device__ __forceinline__ unsigned int ka_or(unsigned int *hist, unsigned int w, int lane, int wave)
{
    if (lane == 0) hist[KA_FLAGS_AT + wave] = w;
    __syncthreads();
    unsigned int x = hist[KA_FLAGS_AT + (lane & (KA_WAVES - 1))];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) x |= __shfl_xor(x, o);
    return __builtin_amdgcn_readfirstlane(x);
}

// ---- step 0: the cell's window from ~256 evenly spaced rows: klo and the shift of the general form (per lane), and
// the map t = fma(v - vlo, sf, 1) of the fast form (sf < 0: the window is not finite, the fast form must not be used)
template <int LOG_NB1>
__device__ __forceinline__ int ka_window(const float *__restrict__ col, bool cok, int n, long long M, unsigned int *hist,
                                         unsigned int &klo, float &sf, float &vlo, int lane, int wave)
{
    // sample rows floor(j*n/m), j = wave, wave+16, ...: 256 of them, or 64 for n <= 1024, where 256 rows would be a
    // quarter to all of a sweep; the small sample's range is then widened by half on each side (below)
    const int m = n <= 1024 ? (n < 64 ? n : 64) : 256;
    unsigned int kmin = 0xffffffffu, kmax = 0u;
    if (cok) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = wave + u * KA_WAVES;
            v[u] = j < m ? ka_row(col + (long long)(((long long)j * n) / m) * M, lane * 4) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (wave + u * KA_WAVES < m) {
                const unsigned int k = f2key(v[u]);
                kmin = min(kmin, k);
                kmax = max(kmax, k);
            }
    }
    hist[wave * 64 + lane] = kmin;
    hist[1024 + wave * 64 + lane] = kmax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < KA_WAVES; ++w) {
        kmin = min(kmin, hist[w * 64 + lane]);
        kmax = max(kmax, hist[1024 + w * 64 + lane]);
    }
    __syncthreads();
    klo = cok ? kmin : 0u;
    const unsigned int range = cok ? kmax - kmin : 0u;
    // general form: smallest shift with (range >> shift) <= NB1 - 2 (bucket NB1 - 1 is the overflow bucket)
    int s = range == 0u ? 0 : max(0, (32 - __clz((int)range)) - LOG_NB1);
    if ((range >> s) > (unsigned)((1 << LOG_NB1) - 2)) ++s;
    // fast form: rows 1 .. NB1-1 hold the window (t in [1, NB1)), row 0 = below it, row NB1 = above it:
    // sf a hair under (NB1-1)/(vhi - vlo), so that the window's top value stays below NB1
    vlo = key2f(kmin);
    float r = key2f(kmax) - vlo;
    if (m < n && m < 256) {                                // 64 of n rows: the 5 % / 95 % ranks may lie outside their range
        vlo -= 0.5f * r;
        r *= 2.0f;
    }
    sf = ((float)((1 << LOG_NB1) - 1) / r) * 0.999999f;
    // no fast form for this cell when the window is empty (one value sampled: sf = inf, and with sf = 0 an infinite
    // score would turn into 0 * inf = NaN and be filed BELOW the window), not finite, or so narrow that sf overflows
    if (!(r > 0.f) || !(r < __builtin_inff()) || !(fabsf(vlo) < __builtin_inff()) || !(sf < __builtin_inff())) sf = -1.f;
    if (!cok) { sf = 0.f; vlo = 0.f; }
    return s;
}

// row of the fast form: 0 for t < 1 (below the window), NB1 at or beyond the window's end.  (v - vlo first: folding
// vlo into the fma's addend would be one instruction less but rounds the window position by ulp(vlo * sf) rows.)
template <int NB1>
__device__ __forceinline__ int ka_frow(float v, float sf, float vlo)
{
    return (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(v - vlo, sf, 1.0f), 0.f, (float)NB1);
}

// ---- narrowing after a first-digit sweep.  Buckets d = 0..NB1-1 live in histogram rows row0 + d, the elements
// below the window in row `urow`.  For the thread's (cell, rank): bucket, its count, the rank left inside it;
// `outside`: the rank lies below the window or in the overflow bucket NB1-1 (`window`).
template <int LOG_NB1, bool WIDE>
__device__ __forceinline__ void ka_narrow_first(unsigned int *hist, int row0, int urow, bool window, bool state,
                                                unsigned int &myr, int &digit, unsigned int &count, bool &outside,
                                                int lane, int wave)
{
    using C = Ctr<WIDE>;
    constexpr int NB1 = 1 << LOG_NB1, GB = NB1 / KA_WAVES, GROUPS_AT = (NB1 + 1) * C::CW;
    // 16 groups of GB bins are summed by all 1024 threads first
    unsigned int gs = 0;
#pragma unroll 8
    for (int u = 0; u < GB; ++u) gs += C::get(hist, row0 + wave * GB + u, lane);
    hist[GROUPS_AT + wave * 64 + lane] = gs;
    __syncthreads();
    outside = false;
    digit = 0;
    count = 0;
    if (state) {
        const unsigned int under = C::get(hist, urow, lane);
        outside = myr < under;
        const unsigned int r = outside ? 0u : myr - under;
        // group of my rank = number of groups whose inclusive running count is <= r (branch-free: the LDS reads
        // are independent and pipeline)
        unsigned int run = 0, cum = 0;
        int g = 0;
#pragma unroll
        for (int u = 0; u < KA_WAVES; ++u) {
            run += hist[GROUPS_AT + u * 64 + lane];
            const bool le = run <= r;
            g += le;
            cum = le ? run : cum;
        }
        g = min(g, KA_WAVES - 1);
        const int bin0 = g * GB;
        run = cum;
        int d = 0;
#pragma unroll 8
        for (int bin = bin0; bin < bin0 + GB; ++bin) {
            run += C::get(hist, row0 + bin, lane);
            const bool le = run <= r;
            d += le;
            cum = le ? run : cum;
        }
        digit = bin0 + min(d, GB - 1);
        outside = outside || (window && digit == NB1 - 1);
        count = C::get(hist, row0 + digit, lane);
        myr = r - cum;
    }
}

template <int LS> __device__ __forceinline__ int ka_list(int slot, int i, int cell) { return (slot * LS + i) * 64 + cell; }

// counters 0, entries all ones (the sentinel the pick relies on)
template <int LS> __device__ __forceinline__ void ka_list_init(unsigned int *hist, int tid)
{
    int rm = (tid >> 6) % LS;              // list row of word i: (i >> 6) % LS, advanced without a division per store
    for (int i = tid; i < KA_MAXK * LS * 64; i += 1024) {
        hist[i] = rm == LS - 1 ? 0u : 0xffffffffu;
        rm += 16;
        rm = rm >= LS ? rm - LS : rm;
    }
}

// the owner thread picks its rank among the <= CAP candidates of its slot by counting (k-th smallest = the
// smallest candidate with more than k candidates <= it)
template <int LS>
__device__ __forceinline__ unsigned int ka_pick(const unsigned int *hist, int myslot, bool open, unsigned int myr, int lane)
{
    const int c = open ? (int)hist[ka_list<LS>(myslot, LS - 1, lane)] : 0;
    int cmax = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    unsigned int ans = 0xffffffffu;
    for (int i = 0; i < cmax; ++i) {
        const unsigned int ki = hist[ka_list<LS>(myslot, i, lane)];      // sentinel beyond my own count
        unsigned int le = 0;
        for (int j = 0; j < cmax; ++j) le += hist[ka_list<LS>(myslot, j, lane)] <= ki;
        if (i < c && le > myr) ans = min(ans, ki);
    }
    return ans;
}

// ---- the fast form (steps 1 + 3).  Returns true (block-uniform) when the tile is finished: `ans` then holds the
// KEY of the thread's (cell, rank).  Returns false when some pair has more than CAP elements in its bucket, a
// rank lies outside its window, the window is not finite, or the tile holds a NaN: nothing is kept, the caller
// runs the general form; *below reports whether a rank lay outside its window.
template <int LOG_NB1, bool WIDE, int LSX = 0>
__device__ __forceinline__ bool ka_fast(const float *__restrict__ col, bool cok, int n, long long M, int nk, float sf, float vlo,
                                        unsigned int *hist, unsigned int k0, unsigned int &ans, bool &below, int lane,
                                        int wave, int tid)
{
    using C = Ctr<WIDE>;
    using Cfg = KACfg<LOG_NB1, WIDE, LSX>;
    constexpr int NB1 = 1 << LOG_NB1, U = Cfg::U, LS = Cfg::LS, CAP = Cfg::CAP;
    const bool state = wave < nk;
    for (int i = tid; i < (NB1 + 1) * C::CW; i += 1024) hist[i] = 0u;
    __syncthreads();
    const unsigned int inc = C::inc(lane);
    unsigned long long nan = 0ull;
    // KEEP: where one workgroup has the CU to itself (1024 buckets: n > 2048, 256 rows per thread at n = 4096) there are
    // ~60 registers to spare: the first KEEP rows of every thread stay in them after this sweep, and the collect sweep
    // re-reads only the others - 2.06 reads of the scores become 1.88 at n = 4096 (the kernel is HBM-bound there)
    constexpr int KEEP = (Cfg::WG_PER_CU == 1 && !WIDE) ? (LSX ? KA_KEEP_BIG : KA_KEEP) : 0;
    static_assert(KEEP * KA_WAVES <= 2048 || KEEP == 0, "the kept rows must exist for every n the instantiation serves");
    static_assert(KEEP % U == 0, "the kept rows are loaded in whole batches of U");
    // (80 of the 256 rows a thread has at n = 4096: 119 of the 128 registers a thread of a 16-wave workgroup may hold; 88
    // would still fit - 125 - but is not a whole number of load batches, and 96 spills)
    float kept[KEEP ? KEEP : 1];
    auto count1 = [&](float v) __attribute__((always_inline)) {
        nan |= __ballot(v != v);
        atomicAdd(&hist[C::word(ka_frow<NB1>(v, sf, vlo), lane)], inc);
    };
    if constexpr (KEEP > 0) {
        if (cok) {
            const int loff = lane * 4;
            const float *p = col + (long long)wave * M;
#pragma unroll
            for (int u0 = 0; u0 < KEEP; u0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) { kept[u0 + u] = ka_row(p, loff); p += (long long)KA_WAVES * M; }
#pragma unroll
                for (int u = 0; u < U; ++u) count1(kept[u0 + u]);
            }
        }
    }
    ka_sweep<U>(col, cok, n, M, wave, count1, KEEP);
    __syncthreads();
    unsigned int myr = k0, count;
    int digit;
    bool outside;
    ka_narrow_first<LOG_NB1, WIDE>(hist, 1, 0, true, state, myr, digit, count, outside, lane, wave);
    const bool bad = state && (outside || sf < 0.f), many = state && count > (unsigned)CAP;
    const unsigned int w = (__ballot(many) != 0 ? 1u : 0u) | (__ballot(bad) != 0 ? 2u : 0u) | (nan != 0ull ? 4u : 0u);
    __syncthreads();                       // everyone is done reading the histograms
    if (state) hist[wave * 64 + lane] = (unsigned)(digit + 1);         // publish my ROW for the collect's matching
    const unsigned int fl = ka_or(hist, w, lane, wave);
    below = (fl & 2u) != 0u;
    if (fl) return false;

    // collect: every pair has <= CAP elements in its row.  An element is tested against the bitmap of its cell's
    // target rows (one conflict-free LDS read).  Slots are the cell's distinct target rows in ascending order, so
    // the slot of a hit is the number of target rows below it: the word's base count + a popcount - four
    // instructions, which matters because a wave takes the branch when ANY of its 64 cells hits (~60 % of the rows).
    unsigned int pf[KA_MAXK];
    int lmax = 1, myslot = 0;
    ka_prefixes(hist, nk, lane, wave, pf, lmax, myslot);
    ka_list_init<LS>(hist, tid);
    for (int i = tid; i < Cfg::BM_WORDS; i += 1024) hist[Cfg::BM_AT + i] = 0u;
    __syncthreads();
    const int myrow = digit + 1;
    if constexpr (Cfg::BYTEMAP) {
        unsigned char *map = reinterpret_cast<unsigned char *>(hist + Cfg::BM_AT);
        if (state) map[myrow * KA_MAPROW + lane] = (unsigned char)(myslot + 1);
        __syncthreads();
        auto collect = [&](float v) __attribute__((always_inline)) {
            const int m = map[ka_frow<NB1>(v, sf, vlo) * KA_MAPROW + lane];
            if (m) {
                const unsigned int pos = atomicAdd(&hist[ka_list<LS>(m - 1, CAP, lane)], 1u);
                if (pos < (unsigned)CAP) hist[ka_list<LS>(m - 1, (int)pos, lane)] = f2key(v);  // (always: the histogram counted them)
            }
        };
        if constexpr (KEEP > 0) {
            if (cok) {
#pragma unroll
                for (int u = 0; u < KEEP; ++u) collect(kept[u]);
            }
        }
        ka_sweep<U>(col, cok, n, M, wave, collect, KEEP);
    } else {
        unsigned int *myword = &hist[Cfg::BM_AT + (myrow >> 4) * 64 + lane];
        if (state) atomicOr(myword, 1u << (myrow & 15));
        __syncthreads();
        if (state) atomicOr(myword, (unsigned)(myslot - __popc(*myword & ((1u << (myrow & 15)) - 1u))) << 16);
        __syncthreads();
        auto collect = [&](float v) __attribute__((always_inline)) {
            const int row = ka_frow<NB1>(v, sf, vlo);
            const unsigned int w = hist[Cfg::BM_AT + (row >> 4) * 64 + lane], bit = (unsigned)row & 15u;
            if ((w >> bit) & 1u) {
                const int slot = (int)(w >> 16) + __popc(w & ((1u << bit) - 1u));
                const unsigned int pos = atomicAdd(&hist[ka_list<LS>(slot, CAP, lane)], 1u);
                if (pos < (unsigned)CAP) hist[ka_list<LS>(slot, (int)pos, lane)] = f2key(v);   // (always: the histogram counted them)
            }
        };
        if constexpr (KEEP > 0) {
            if (cok) {
#pragma unroll
                for (int u = 0; u < KEEP; ++u) collect(kept[u]);
            }
        }
        ka_sweep<U>(col, cok, n, M, wave, collect, KEEP);
    }
    __syncthreads();
    if (state) ans = ka_pick<LS>(hist, myslot, true, myr, lane);
    return true;
}

// ---- the fast form with the register tiles' bookkeeping (round 5; 2048 < n <= 4096, 1024 buckets, one workgroup per CU).
// Same two sweeps as ka_fast, same window, same rows; what changed is everything between and around them:
//   * "which list, if any, wants this row of this cell" is a 4-bit tag above the count in the histogram word of the row (a
//     count is at most n - 1 <= 4095 here: the sample that defines the window is part of the column, so two rows are
//     occupied) - written by the rank's owner with a compare-and-swap, read back by the collect sweep from the word the
//     first sweep incremented: conflict-free (the byte map's reads fell on random banks: a third of the LDS cycles), and no
//     70 KB map to clear, no published rows, no prefix compaction (three barriers less);
//   * the lists are exact-size segments of one pool per cell (the histogram knows the sizes): no 80 KB of lists to
//     initialise; the pool lies behind the histogram (which stays live through the collect sweep) and shares its first
//     words with the group sums of the narrowing (dead by then);
//   * the owner sorts its <= 31 candidates in registers (a network) instead of counting them against each other in LDS.
// Returns as ka_fast does; nothing it leaves behind matters to the general form, which clears what it uses.
template <int N, int P, int NW> __device__ __forceinline__ void ks_sort(unsigned int (&v)[NW]);
template <int N>
__device__ __forceinline__ unsigned int ka_pick_net(const unsigned int *pool, unsigned int base, unsigned int count, unsigned int myr)
{
    unsigned int c[N];
    const char *p = reinterpret_cast<const char *>(pool) + base;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned int j = min((unsigned)i, count - 1u);                  // (never beyond my own segment: the pool has no slack)
        const unsigned int x = f2key(__uint_as_float(*reinterpret_cast<const unsigned int *>(p + j * 256u)));
        c[i] = (unsigned)i < count ? x : 0xffffffffu;
    }
    ks_sort<N, 1, N>(c);
    unsigned int ans = c[0];
#pragma unroll
    for (int i = 1; i < N; ++i) ans = myr == (unsigned)i ? c[i] : ans;
    return ans;
}

#ifdef KA_DEBUG_FAIL
__device__ unsigned long long ka_debug_fail[8];
#endif
// ARR = false (2048 < n <= 4096): the tags live in the histogram words as described above; lists of up to 31, pool behind the
// histogram.  ARR = true (4096 < n <= 12288: a count no longer leaves a 16-bit counter four spare bits): the tags have
// their own array - a nibble per (row, cell), word [row >> 3][cell]: as conflict-free as the histogram - and the pool
// takes the histogram's place once the narrowing has read it (448 entries per cell); lists of up to 63 (the bucket at the
// mode of n = 9216 |N(0,1)| scores holds 25, at n = 12288 it holds 33), sorted by a 64-input network.  This form replaces
// the 47-entry lists + bitmap of rounds 3-4 between 4097 and 9216 rows (published rows, prefix compaction, 123 KB of lists
// to initialise, a pick that counts through LDS) and the general radix form's third sweep between 9217 and 12288.
template <int LOG_NB1, bool ARR>
__device__ __forceinline__ bool ka_fast_tags(const float *__restrict__ col, bool cok, int n, long long M, int nk, float sf, float vlo,
                                             unsigned int *hist, unsigned int k0, unsigned int &ans, bool &below, int lane,
                                             int wave, int tid)
{
    using Cfg = KACfg<LOG_NB1, false, ARR ? 48 : 0>;
    // 896 buckets, not 1024: the 16 KB that frees take the pool (ARR = false) from 116 to 179 entries per cell.  With 116,
    // 20 % of the tiles of n = 4096 |N(0,1)| scores exhausted some cell's pool (ten lists of ~9 around the mode: 75 on
    // average, 125 in the fullest of a tile's 64 cells; tools/exp/stream_fail_probe.py) and went to the general form -
    // slower than the byte-map form this replaces; a bucket is 14 % fuller for it, far below the 31 a list may hold.
    // (ARR = true: the same 896 leave room for the tag array.)
    constexpr int NB1 = 896, U = Cfg::U, CAP = ARR ? 63 : 31, GB = NB1 / KA_WAVES;
    constexpr unsigned int CMASK = ARR ? 0xffffu : 0xfffu;     // a count (ARR = false: at most n - 1 <= 4095, the tag above it)
    // (ARR = false: the caller sends only n <= KA_TAGS_MAX_N here; a count of 4096 would read as 0 under a set tag bit.
    // n - 1 because a window that is not flat has at least two occupied rows; a flat one is `bad` and redone.)
    static_assert(ARR || KA_TAGS_MAX_N - 1 <= (int)CMASK, "a 12-bit count under the tag");
    static_assert(NB1 % KA_WAVES == 0 && NB1 <= (1 << LOG_NB1), "buckets");
    sf = sf > 0.f ? sf * ((float)(NB1 - 1) / (float)((1 << LOG_NB1) - 1)) : sf;      // (ka_window scaled the window to 2^LOG_NB1 - 1 rows)
    constexpr int HIST_WORDS = (NB1 + 1) * 32;                 // rows 0 (below the window) .. NB1 (at or beyond its end)
    constexpr int TAG_WORDS = ARR ? ((NB1 + 1 + 7) / 8) * 64 : 0;
    // side: list fill pointers, pool pointers, per-wave flags, a scratch word per lane, (ARR) the narrowing's group sums
    constexpr int SIDE_WORDS = KA_MAXK * 64 + 64 + KA_WAVES + 64 + (ARR ? KA_WAVES * 64 : 0);
    constexpr int POOL = ARR ? HIST_WORDS / 64 : (Cfg::WORDS - HIST_WORDS - SIDE_WORDS) / 64;
    constexpr int POOL_AT = ARR ? 0 : HIST_WORDS;              // (ARR = false: its first 1024 words are the group sums)
    constexpr int SIDE_AT = ARR ? HIST_WORDS + TAG_WORDS : POOL_AT + POOL * 64;
    static_assert(POOL >= 96 && SIDE_AT + SIDE_WORDS <= Cfg::WORDS, "pool");
    unsigned int *pool = hist + POOL_AT, *tags = hist + HIST_WORDS, *cnt = hist + SIDE_AT, *ptr = cnt + KA_MAXK * 64, *flg = ptr + 64;
    unsigned int *grp = ARR ? flg + KA_WAVES + 64 : pool;
    const bool state = wave < nk;
    const int l31 = lane & 31, sh = 16 * (lane >> 5), tsh = 12 + sh;
    for (int i = tid; i < HIST_WORDS + TAG_WORDS; i += 1024) hist[i] = 0u;
    if (tid < 64) ptr[tid] = 0u;
    __syncthreads();
    const unsigned int inc = 1u << sh;
    unsigned long long nan = 0ull;
    constexpr int KEEP = KA_KEEP;
    static_assert(KEEP * KA_WAVES <= 2048 && KEEP % U == 0, "kept rows");
    float kept[KEEP];
    char *hb = reinterpret_cast<char *>(hist + l31);           // word (row, my cell): hb + row * 128
    auto count1 = [&](float v) __attribute__((always_inline)) {
        nan |= __ballot(v != v);
        atomicAdd(reinterpret_cast<unsigned int *>(hb + (ka_frow<NB1>(v, sf, vlo) << 7)), inc);
    };
    if (cok) {
        const int loff = lane * 4;
        const float *p = col + (long long)wave * M;
#pragma unroll
        for (int u0 = 0; u0 < KEEP; u0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) { kept[u0 + u] = ka_row(p, loff); p += (long long)KA_WAVES * M; }
#pragma unroll
            for (int u = 0; u < U; ++u) count1(kept[u0 + u]);
        }
    }
    ka_sweep<U>(col, cok, n, M, wave, count1, KEEP);
    __syncthreads();

    // ---- narrowing (ka_narrow_first's walks; ARR = false: 12-bit counts)
    unsigned int myr = k0, count = 0;
    int digit = 0;
    bool outside = false;
    {
        unsigned int gs = 0;
        const unsigned int *h = hist + (1 + wave * GB) * 32 + l31;
#pragma unroll 8
        for (int u = 0; u < GB; ++u) gs += h[u * 32];          // (both cells of the word at once: a sum is at most n)
        grp[wave * 64 + lane] = (gs >> sh) & 0xffffu;
    }
    __syncthreads();
    if (state) {
        const unsigned int under = (hist[l31] >> sh) & CMASK;
        outside = myr < under;
        const unsigned int r = outside ? 0u : myr - under;
        unsigned int run = 0, cum = 0;
        int g = 0;
#pragma unroll
        for (int u = 0; u < KA_WAVES; ++u) {
            run += grp[u * 64 + lane];
            const bool le = run <= r;
            g += le;
            cum = le ? run : cum;
        }
        g = min(g, KA_WAVES - 1);
        const unsigned int *h = hist + (1 + g * GB) * 32 + l31;
        run = cum;
        int d = 0;
#pragma unroll 8
        for (int u = 0; u < GB; ++u) {
            run += (h[u * 32] >> sh) & CMASK;
            const bool le = run <= r;
            d += le;
            cum = le ? run : cum;
        }
        digit = g * GB + min(d, GB - 1);                       // bucket digit lives in row 1 + digit
        outside = outside || digit == NB1 - 1;                 // (row NB1: at or beyond the window's end)
        count = (h[min(d, GB - 1) * 32] >> sh) & CMASK;
        myr = r - cum;
    }
    // lanes beyond the last cell of a partial tile (!cok) hold an all-zero histogram: their narrowing ends in the overflow
    // row, which must not send the whole tile to the general form (round-5 advice: every plane whose M is not a multiple of
    // 64 paid that on its last tile) - they neither fail, nor tag, nor allocate, and their pick is never written
    const bool owner = state && cok;
    if (!cok) count = 0u;
    bool many = owner && count > (unsigned)CAP;
    const bool bad = owner && (outside || sf < 0.f);
    __syncthreads();                                           // every owner has read what it needs: the pool is free
    // the word that carries my row's tag, and where in it
    const int myrow = digit + 1;
    unsigned int *a = ARR ? tags + (myrow >> 3) * 64 + lane : hist + myrow * 32 + l31;
    const int ash = ARR ? (myrow & 7) * 4 : tsh;
    if (owner && !many && !bad) {
        // win the tag FIRST, then take the entries: two ranks of a cell that share a row may both see it untagged, and an
        // allocation before the compare-and-swap would charge the pool once per rank instead of once per distinct row
        // (round-5 advice: on quantised data that exhausts pools spuriously, and whether it does depends on wave timing).
        // The loser adopts the winner's list; `cnt` and the tag are read only after the next barrier.
        unsigned int old = *a;
        bool won = false;
        while (((old >> ash) & 15u) == 0u) {
            const unsigned int prev = atomicCAS(a, old, old | ((unsigned)(wave + 1) << ash));
            if (prev == old) { won = true; break; }
            old = prev;                                        // (another tag of the word was set meanwhile)
        }
        if (won) {
            const unsigned int start = atomicAdd(&ptr[lane], count);
            if (start + count > (unsigned)POOL) many = true;   // (pool exhausted: ties - the general form)
            else cnt[wave * 64 + lane] = (start * 64u + (unsigned)lane) * 4u;
        }
    }
    const unsigned int w = (__ballot(many) != 0 ? 1u : 0u) | (__ballot(bad) != 0 ? 2u : 0u) | (nan != 0ull ? 4u : 0u);
    if (lane == 0) flg[wave] = w;
    int cmax = owner ? (int)count : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    __syncthreads();
    unsigned int fl = flg[lane & (KA_WAVES - 1)];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) fl |= __shfl_xor(fl, o);
    fl = __builtin_amdgcn_readfirstlane(fl);
    below = (fl & 2u) != 0u;
#ifdef KA_DEBUG_FAIL
    if (tid == 0) {
        atomicAdd(&ka_debug_fail[0], 1ull);
        if (fl & 1u) atomicAdd(&ka_debug_fail[1], 1ull);
        if (fl & 2u) atomicAdd(&ka_debug_fail[2], 1ull);
        if (fl & 4u) atomicAdd(&ka_debug_fail[3], 1ull);
        atomicAdd(&ka_debug_fail[4], (unsigned long long)cmax);
        atomicMax(&ka_debug_fail[5], (unsigned long long)cmax);
        atomicMax(&ka_debug_fail[6], (unsigned long long)ptr[0]);
    }
#endif
    if (fl) return false;
    const int myslot = owner ? (int)((*a >> ash) & 15u) - 1 : 0;

    // ---- collect: an element whose row is tagged joins the tagged list (inside its segment always: the histogram counted).
    // (ARR: the appends overwrite the histogram - nothing reads it any more.)
    char *cb = reinterpret_cast<char *>(cnt + lane) - 256;     // fill pointer of list m - 1 of my cell: cb + m * 256
    char *lb = reinterpret_cast<char *>(pool);
    const char *tb = reinterpret_cast<const char *>(tags + lane);      // (ARR) word ((row >> 3), my cell): tb + (row >> 3) * 256
    // H rows at a time, branch-free (as in the register tiles): the H tag words are read together, every element issues
    // its returning add (of 0, on a word that ignores it, when its row is untagged) and its store (to a scratch word of its
    // lane): one LDS round trip per phase and half batch instead of two per element
    constexpr int H = 8;
    char *dummy = reinterpret_cast<char *>(flg + KA_WAVES + lane);
    unsigned int *dummy_cnt = reinterpret_cast<unsigned int *>(dummy);          // (an add of 0 changes nothing wherever it lands)
    auto collect = [&](float (&v)[H], int nv) __attribute__((always_inline)) {
        unsigned int m[H];
        unsigned int rsh[H];
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const int row = ka_frow<NB1>(v[u], sf, vlo);
            if constexpr (ARR) {
                m[u] = *reinterpret_cast<const unsigned int *>(tb + ((row >> 3) << 8));
                rsh[u] = (unsigned)(row & 7) << 2;
            } else {
                m[u] = *reinterpret_cast<const unsigned int *>(hb + (row << 7));
                rsh[u] = (unsigned)tsh;
            }
        }
        bool hit[H];
#pragma unroll
        for (int u = 0; u < H; ++u) {
            m[u] = __builtin_amdgcn_ubfe(m[u], rsh[u], 4u);
            hit[u] = m[u] != 0u && u < nv;                         // (rows beyond n came back 0.0 through an empty descriptor)
            unsigned int *fp = hit[u] ? reinterpret_cast<unsigned int *>(cb + (m[u] << 8)) : dummy_cnt;
            m[u] = atomicAdd(fp, hit[u] ? 256u : 0u);
        }
#pragma unroll
        for (int u = 0; u < H; ++u) *reinterpret_cast<float *>(hit[u] ? lb + m[u] : dummy) = v[u];
    };
    if (cok) {
#pragma unroll
        for (int u = 0; u < KEEP; u += H) collect(*reinterpret_cast<float(*)[H]>(&kept[u]), H);
    }
#if KA_REV
    ka_sweep_h_rev<U, H>(col, cok, n, M, wave, collect, KEEP);
#else
    ka_sweep_h<U, H>(col, cok, n, M, wave, collect, KEEP);
#endif
    __syncthreads();
    if (state) {                                               // (whole waves: cmax is theirs)
        const unsigned int base = cnt[myslot * 64 + lane] - count * 256u;
        if (count == 0u) ans = 0u;                             // (cannot happen for a rank inside its window; defined anyway)
        else if (cmax <= 8) ans = ka_pick_net<8>(pool, base, count, myr);
        else if (cmax <= 16) ans = ka_pick_net<16>(pool, base, count, myr);
        else if (!ARR || cmax <= 32) ans = ka_pick_net<32>(pool, base, count, myr);
        else ans = ka_pick_net<64>(pool, base, count, myr);
    }
    return true;
}

// ---- step 1, general form: buckets (key - klo) >> shift (clamped to the overflow bucket NB1-1) in rows
// 0..NB1-1, row NB1 = key < klo.  `window`: a rank that lands in the overflow bucket or below the window is
// reported through `outside`.  Returns (block-uniform) whether some pair has more than CAP elements left.
template <int LOG_NB1, bool WIDE, int U, int CAP, class Src>
__device__ __forceinline__ bool ka_first(const Src &src, int nk, int shift,
                                         unsigned int klo, bool window, unsigned int *hist, unsigned int &myp,
                                         unsigned int &myr, bool &outside, bool &nanl, int lane, int wave, int tid)
{
    using C = Ctr<WIDE>;
    constexpr int NB1 = 1 << LOG_NB1;
    const bool state = wave < nk;
    for (int i = tid; i < (NB1 + 1) * C::CW; i += 1024) hist[i] = 0u;
    __syncthreads();
    const unsigned int inc = C::inc(lane);
    src.template sweep<U>([&](float v) __attribute__((always_inline)) {
        nanl |= v != v;                    // (this sweep sees every element of my cell in my wave's rows)
        const unsigned int key = f2key(v);
        const unsigned int d = min((key - klo) >> shift, (unsigned)(NB1 - 1));
        atomicAdd(&hist[C::word(key >= klo ? (int)d : NB1, lane)], inc);
    });
    __syncthreads();
    int digit;
    unsigned int count;
    ka_narrow_first<LOG_NB1, WIDE>(hist, 0, NB1, window, state, myr, digit, count, outside, lane, wave);
    const bool many = state && shift > 0 && count > (unsigned)CAP;          // shift == 0: every bit is known
    if (state) myp = (unsigned)digit << shift;
    const unsigned int w = (__ballot(many) != 0 ? 1u : 0u) | (__ballot(outside) != 0 ? 2u : 0u);
    __syncthreads();                       // everyone is done reading the histograms
    if (state) hist[wave * 64 + lane] = myp;                  // publish for the next sweep's matching
    const unsigned int fl = ka_or(hist, w, lane, wave);
    outside = (fl & 2u) != 0u;                                // block-uniform from here on
    return (fl & 1u) != 0u;
}

// ---- step 2: one more digit of BITS bits at `shift` under the prefixes known down to bit `pshift` (both per
// cell; shift + BITS may exceed pshift for the last digit: the overlapping bits are fixed by the prefix match, so
// only consistent bins fill; a cell that is already fully known, pshift == 0, just recounts its ties).  Returns
// whether some pair still has more than CAP elements under its prefix.
template <int BITS, bool WIDE, int U, int CAP, class Src>
__device__ __forceinline__ bool ka_pass(const Src &src, int nk, int pshift,
                                        int shift, unsigned int klo, unsigned int *hist, unsigned int &myp,
                                        unsigned int &myr, int lane, int wave, int tid)
{
    using C = Ctr<WIDE>;
    constexpr int NB = 1 << BITS;
    const unsigned int mask = ~0u << pshift;
    const bool state = wave < nk;
    unsigned int pf[KA_MAXK];
    int lmax = 1, myslot = 0;
    ka_prefixes(hist, nk, lane, wave, pf, lmax, myslot);
    for (int i = tid; i < KA_MAXK * NB * C::CW; i += 1024) hist[i] = 0u;
    __syncthreads();

    const unsigned int inc = C::inc(lane);
    ka_by_slots(lmax, [&](auto lm) __attribute__((always_inline)) {
        src.template sweep<U>([&](float v) __attribute__((always_inline)) {
            const unsigned int kk = ka_kk(v, klo);
            const int m = ka_match<decltype(lm)::value>(kk & mask, pf);
            if (m) atomicAdd(&hist[C::word((m - 1) * NB + (int)((kk >> shift) & (NB - 1)), lane)], inc);
        });
    });
    __syncthreads();

    bool many = false;
    if (state) {
        // digit = number of bins whose inclusive running count is <= myr; `cum` ends as the count before it
        const int base = myslot * NB;
        unsigned int run = 0, cum = 0;
        int d = 0;
#pragma unroll 8
        for (int bin = 0; bin < NB; ++bin) {
            run += C::get(hist, base + bin, lane);
            const bool le = run <= myr;
            d += le;
            cum = le ? run : cum;
        }
        const int digit = min(d, NB - 1);
        many = shift > 0 && C::get(hist, base + digit, lane) > (unsigned)CAP;
        if (pshift > 0) {
            myp |= (unsigned)digit << shift;
            myr -= cum;
        }
    }
    const unsigned int w = __ballot(many) != 0 ? 1u : 0u;
    __syncthreads();                       // everyone is done reading the histograms
    if (state) hist[wave * 64 + lane] = myp;
    return ka_or(hist, w, lane, wave) != 0u;
}

// ---- step 3, general form: every (cell, rank) of the tile has at most CAP elements left under its prefix
// (known down to bit `known`, per cell): ONE sweep appends each surviving kk to the list of its slot -
// list[slot][i][cell], fill counter in row CAP - and the owner thread picks its rank.
template <int U, int LS, class Src>
__device__ __forceinline__ void ka_collect(const Src &src, int nk, int known,
                                           unsigned int klo, unsigned int *hist, unsigned int &myp, unsigned int myr,
                                           int lane, int wave, int tid)
{
    const unsigned int mask = ~0u << known;
    unsigned int pf[KA_MAXK];
    int lmax = 1, myslot = 0;
    ka_prefixes(hist, nk, lane, wave, pf, lmax, myslot);
    ka_list_init<LS>(hist, tid);
    __syncthreads();

    ka_by_slots(lmax, [&](auto lm) __attribute__((always_inline)) {
        src.template sweep<U>([&](float v) __attribute__((always_inline)) {
            const unsigned int kk = ka_kk(v, klo);
            const int m = ka_match<decltype(lm)::value>(kk & mask, pf);
            if (m) {
                const unsigned int pos = atomicAdd(&hist[ka_list<LS>(m - 1, LS - 1, lane)], 1u);
                // pos < CAP for every pair still open (the histogram counted its elements); a fully known cell
                // (known == 0) matches all its ties, which are not needed
                if (pos < (unsigned)(LS - 1)) hist[ka_list<LS>(m - 1, (int)pos, lane)] = kk;
            }
        });
    });
    __syncthreads();
    if (wave < nk) {                       // (whole waves: the pick shuffles across the lanes)
        const unsigned int ans = ka_pick<LS>(hist, myslot, known > 0, myr, lane);
        if (known > 0) myp = ans;
    }
}

// one tile (64 cells from c0) of the streaming form, all phases, by the whole workgroup
template <int LOG_NB1, bool WIDE, int LSX = 0>
__device__ __forceinline__ void ka_tile(const float *__restrict__ s, int n, long long M, long long S, long long c0, const KAList &kl,
                                        int fast, float *__restrict__ out, long long OS, unsigned int *hist)
{
    using Cfg = KACfg<LOG_NB1, WIDE, LSX>;
    constexpr int U = Cfg::U, BITS = Cfg::BITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = kl.nk;
    const long long c = c0 + lane;
    const bool cok = c < M;
    const float *col = s + c0;             // the tile's first cell: wave-uniform (lane offsets are added at the loads)
    const bool state = wave < nk;
    const unsigned int k0 = state ? (unsigned)kl.k[wave] : 0u;

    unsigned int klo = 0u;
    float sf = 0.f, vlo = 0.f;
    // (the helpers below use their `M` argument only as the distance between rows: they get the row stride S)
    int shift = ka_window<LOG_NB1>(col, cok, n, S, hist, klo, sf, vlo, lane, wave);
    bool outside = false;
    if (fast) {
        unsigned int key = 0u;
        bool done;
        if constexpr (Cfg::TAGS) {
            if (n <= KA_TAGS_MAX_N) done = ka_fast_tags<LOG_NB1, false>(col, cok, n, S, nk, sf, vlo, hist, k0, key, outside, lane, wave, tid);
            else done = ka_fast<LOG_NB1, WIDE, LSX>(col, cok, n, S, nk, sf, vlo, hist, k0, key, outside, lane, wave, tid);
        } else if constexpr (Cfg::TAGS_ARR) {
            done = ka_fast_tags<LOG_NB1, true>(col, cok, n, S, nk, sf, vlo, hist, k0, key, outside, lane, wave, tid);
        } else {
            done = ka_fast<LOG_NB1, WIDE, LSX>(col, cok, n, S, nk, sf, vlo, hist, k0, key, outside, lane, wave, tid);
        }
        if (done) {
            if (state && cok) out[(long long)kl.o[wave] * OS + c] = key2f(key);
            return;
        }
        __syncthreads();                   // (every wave has read the flags before the histogram memory is cleared again)
    }
    unsigned int myp = 0u, myr = k0;
    bool many = true, nanl = false;
    const HbmSrc src{col, cok, n, S, wave};
    if (!outside) {
        many = ka_first<LOG_NB1, WIDE, U, Cfg::CAP>(src, nk, shift, klo, true, hist, myp, myr, outside, nanl, lane, wave, tid);
        if (outside) __syncthreads();
    }
    if (outside) {                         // some rank lies outside its cell's sampled window: plain top digit
        klo = 0u;
        shift = 32 - LOG_NB1;
        myp = 0u;
        myr = k0;
        many = ka_first<LOG_NB1, WIDE, U, Cfg::CAP>(src, nk, shift, klo, false, hist, myp, myr, outside, nanl, lane, wave, tid);
    }
    int known = shift;                     // lowest known bit of my cell's kk so far (per lane)
#pragma unroll 1
    while (many) {
        const int sh = known > BITS ? known - BITS : 0;
        many = ka_pass<BITS, WIDE, U, Cfg::CAP>(src, nk, known, sh, klo, hist, myp, myr, lane, wave, tid);
        known = sh;
    }
    __syncthreads();
    if (ka_or(hist, __ballot(state && known > 0) != 0 ? 1u : 0u, lane, wave))      // else every bit of every cell is counted
        ka_collect<U, Cfg::LS>(src, nk, known, klo, hist, myp, myr, lane, wave, tid);
    // np.quantile: a NaN anywhere in a cell's column makes every quantile of that cell NaN (the fast form never
    // finishes a tile that holds one).  Cell flags are OR-ed across the waves through 64 LDS words.
    __syncthreads();
    if (tid < 64) hist[tid] = 0u;
    __syncthreads();
    if (nanl) hist[lane] = 1u;
    __syncthreads();
    if (state && cok) out[(long long)kl.o[wave] * OS + c] = hist[lane] ? __uint_as_float(0x7fc00000u) : key2f(myp + klo);
}

template <int LOG_NB1, bool WIDE, int LSX = 0>
__global__ void __launch_bounds__(1024, (4 * KACfg<LOG_NB1, WIDE, LSX>::WG_PER_CU))
kth_axis0_kernel(const float *__restrict__ s, int n, long long M, long long S, long long tile0, const KAList kl, int fast,
                 float *__restrict__ out, const KAPlanes pl)
{
    __shared__ unsigned int hist[KACfg<LOG_NB1, WIDE, LSX>::WORDS];
    long long plane, c0;
    ka_locate(pl, tile0 + blockIdx.x, plane, c0);
    ka_tile<LOG_NB1, WIDE, LSX>(s + plane * pl.PS, n, M, S, c0, kl, fast, out + plane * pl.OPS, pl.OS, hist);
}

// ---- 128 < n <= 1024: the tile lives in REGISTERS ---------------------------------------------------------------
// (the reference's own calibration sets: n_cal = 1000, Marginal/Wave_Residuals_CP.py:284-290; BASELINE C2: n = 512.)
// At these n the sweeps above are short - 16 to 64 rows per thread - and what a tile costs is the chain
// sample -> count -> narrow -> collect -> pick with an HBM round trip per batch of loads, twice over, plus two reads of
// the scores.  Here ONE persistent 1024-thread workgroup per CU keeps the whole 64-cell tile in its registers
// (thread (wave, lane) holds rows wave, wave + 16, ... of cell lane: R = 32 or 64 registers), so the scores are read
// from HBM exactly ONCE, every sweep runs out of registers, and the window of a cell is its exact [min, max] instead
// of a sample's.  The next tile is loaded INTO THE SAME REGISTERS while the last sweep of the current one consumes
// them (row u is re-loaded right after its element has been collected), so the loads are in flight during the rest
// of that sweep, the pick and the next tile's set-up; the barriers in between fence the LDS only and leave vmcnt
// alone.  The arithmetic is the fast form's: NB1 value-linear buckets over the window (16-bit counters, two cells
// per word), narrow, then every element reads "which list wants this row" from a tag above the count in the histogram word
// of its row (KTCfg below) and the owner picks its rank among the <= CAP candidates; cells whose column is constant, or holds a NaN, are settled by the window
// alone.  Tiles the fast form cannot finish (a bucket above CAP: ties, one huge outlier stretching the window; an
// infinite window) are marked and redone by the streaming form above once the workgroup has finished its loop.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr unsigned int KT_MARK = 0xffc0dead;      // "left to the streaming kernel" (results are input values or 0x7fc00000)

// experiment build only (tools/exp/build_variants.sh ... -DKT_CLOCK): thread 0 of every workgroup stamps the shader clock at
// the phase boundaries of its first 64 tiles (tools/exp/tile_phases.py reads them back)
#ifdef KT_CLOCK
__device__ unsigned long long kt_clock_buf[1024 * 64 * 8];
#define KT_STAMP(i)                                                                                                         \
    do {                                                                                                                    \
        if (threadIdx.x == 0 && kt_it < 64 && blockIdx.x < 1024)                                                            \
            kt_clock_buf[((size_t)blockIdx.x * 64 + kt_it) * 8 + (i)] = __builtin_readcyclecounter();                       \
    } while (0)
#else
#define KT_STAMP(i)
#endif

// WGS = workgroups per CU.  Two (R = 32: 64 registers per thread, 80 KiB each) hide each other's barriers and LDS round
// trips - at these tile sizes the phases are latency chains, not throughput; that takes 256 first-digit buckets.
// One (R = 64: the data alone are half the register file) has 512 buckets.
//
// LDS (round 5): [histogram + group sums | lists | side].  "Which list, if any, wants this row of this cell" used to be a
// byte map (a byte per (row, cell), rows 68 bytes apart): the 64 byte reads of a wave-instruction of the collect sweep
// fell on ~random banks, 3.5 lanes deep on the fullest one - the sweep cost 300 clocks per element and wave, four times the
// histogram sweep next to it, whose word (row, lane & 31) is conflict-free by construction (tools/exp/tile_phases.py,
// profiles/r05/tile_phases_*.txt: the same sweep without any memory load still took 85 % of its time - it was never
// waiting for HBM).  Now the answer lives IN the histogram word the element's row was counted in: a count is at most
// n <= 2048, 12 bits, so bits 12-15 of each 16-bit half (bits 16-19 of a 32-cell tile's full-word counters) are free for
// "list + 1" - written by the rank's owner with a compare-and-swap once the narrowing has found its row, read back by the
// collect sweep from the very address the histogram sweep incremented: conflict-free, no map memory, nothing to clean up
// (the next tile's histogram clear wipes the tags).  The lists therefore no longer alias the histogram - which also
// lets the owners pick their ranks while the other waves are already taking the next tile's window and clearing its
// histogram: no barrier at the end of a tile.  Five barriers per tile instead of seven.
template <int LOG_NB1, int WGS>
struct KTCfg {
    static constexpr int NB1 = 1 << LOG_NB1;
    static constexpr int HIST_WORDS = (NB1 + 1) * 32 + KA_WAVES * 64;        // first-digit histogram (+ tags) + group sums
    // The candidate lists of a cell share ONE pool of POOL entries (entry i of cell c: word i * 64 + c): a list is a
    // segment of exactly `count` entries - the histogram knows the size - that its owner allocates with one atomic add
    // on the cell's pool pointer.  Ten fixed lists of 15 (what the same memory holds with two workgroups per CU) sent
    // 2 % of the tiles of n = 512 |N(0,1)| scores to the streaming form - a bucket at the mode holds 5 on average - and
    // their late, lonely repeats cost 19 %; from the pool a list can have all 31 entries the pick can sort while the ~40
    // a cell's ten ranks need in total always fit.
    // (the pick reads a fixed 8 / 16 / 32 entries from a list's start whatever its length: SLACK entries behind the last
    // allocatable one keep those reads inside the pool)
    static constexpr int SLACK = WGS == 2 ? 16 : 32, POOL = (WGS == 2 ? 160 : 320) - SLACK, CAP = 31;
    static constexpr int LIST_AT = HIST_WORDS, LIST_WORDS = (POOL + SLACK) * 64;
    // side arrays: window (min key, max key per cell), list fill pointers, pool pointers, per-wave flags
    static constexpr int SIDE_AT = LIST_AT + LIST_WORDS;
    static constexpr int WIN_AT = 0, CNT_AT = 128, PTR_AT = CNT_AT + KA_MAXK * 64, FLG_AT = PTR_AT + 64, SIDE_WORDS = FLG_AT + KA_WAVES;
    static constexpr int HOT_WORDS = SIDE_AT + SIDE_WORDS + 64;            // (+ a scratch word per lane)
    // (the marked tiles are redone by the streaming form with 512 buckets: 80 KiB from the start of the same block)
    static constexpr int TOTAL = HOT_WORDS > KACfg<9, false>::WORDS ? HOT_WORDS : KACfg<9, false>::WORDS;
    static_assert(TOTAL * 4 * WGS <= 160 * 1024, "does not fit the 160 KiB LDS");
};

// row of an element under the EXACT window [vlo, vlo + r], sf = (NB1-1)/r * 0.999999: t = (v - vlo) sf + 1 lies in
// [1, NB1) for every finite element without any clamp (0 <= v - vlo <= r, and the 1e-6 margin is 8 ulps of NB1); a NaN
// gives row 0 (v_cvt_u32_f32 of a NaN; the cell is settled by its flag), and a window that is not finite has sf = 0:
// row 1, or 0 for inf * 0.  So rows never leave [0, NB1) - the streaming form's v_med3 is not needed here.
__device__ __forceinline__ unsigned int kt_frow(float v, float sf, float vlo)
{
    return (unsigned int)__builtin_fmaf(v - vlo, sf, 1.0f);       // (v_cvt_u32_f32 saturates: negative -> 0, NaN -> 0)
}

// one row of a tile: 64 consecutive floats at the wave-uniform address p, `valid` bytes of them inside the tensor
// (0: a row beyond n, or no next tile - the load then returns 0 and moves nothing)
__device__ __forceinline__ float kt_row(const float *p, int valid, int loff)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, valid, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, 0, 0));
}
// ... at `soff` bytes behind p: the scalar offset operand of the load.  A group of rows then shares ONE descriptor base: per
// row the 64-bit base advance and re-masking of the descriptor (three scalar instructions of the eleven a row of the
// collect sweep cost, in a phase that issues two scalar instructions per vector one) become one 32-bit add.  On gfx9 /
// CDNA the range check is (lane offset >= num_records - scalar offset) - the scalar offset DOES count (measured: with
// num_records = `valid` every row but a group's first came back 0) - so the descriptor's extent is valid + soff: the lanes
// beyond `valid` bytes of the row still read 0, and a row beyond n (valid == 0) reads nothing at all.
__device__ __forceinline__ float kt_row_s(const float *p, int valid, int loff, unsigned int soff)
{
    // (unsigned 32-bit arithmetic: the host keeps the largest offset + extent below 2^32, not below 2^31)
    const unsigned int extent = valid ? (unsigned int)valid + soff : 0u;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)extent, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, (int)soff, 0));
}

// (loop-invariant values derived from the lane / wave index - LDS addresses for every phase, 64 row offsets, 64
// comparisons u < nu - are otherwise all hoisted out of the tile loop and kept live next to the 64 data registers:
// the phases take opaque copies, so that each recomputes the few it needs)
__device__ __forceinline__ int kt_opq_v(int x) { asm volatile("" : "+v"(x)); return x; }
// the lane index, re-made where it is needed (two instructions, nothing kept live or spilled for it)
__device__ __forceinline__ int kt_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ int kt_opq_s(int x) { asm volatile("" : "+s"(x)); return x; }
// (as asm: fminf / fmaxf come with a canonicalising v_max x, x per operand; the hardware's min / max already return the
// other operand when one is a NaN, which is what the window wants)
__device__ __forceinline__ float kt_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float kt_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float kt_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float kt_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// the narrowing after the first-digit sweep of the register-resident form (exact window: nothing below it, nothing in
// the overflow row).  Same walks as ka_narrow_first, kept short in registers: the tile's 64 data registers are live.
// C32 (32-cell tiles): one 32-bit counter per cell and word instead of two 16-bit ones.  row0 = the count of histogram
// row 0: the tile's padding rows and the cell's NaNs (nothing else maps there).
template <int LOG_NB1, bool C32>
__device__ __forceinline__ void kt_narrow(unsigned int *hist, bool state, unsigned int &myr, int &digit, unsigned int &count,
                                          unsigned int &row0, int lane, int wave)
{
    constexpr int NB1 = 1 << LOG_NB1, GB = NB1 / KA_WAVES, GROUPS_AT = (NB1 + 1) * 32;
    // (a count is at most n <= 2048; the bits above it may carry the list tag of an owner that is already done: masked off)
    constexpr unsigned int MSK = C32 ? 0xffffu : 0xfffu;
    const int l31 = lane & 31, sh = C32 ? 0 : 16 * (lane >> 5);
    // 16 groups of GB bins (rows 1 + d) are summed by all 1024 threads first; the two cells of a word are added
    // together (no carry: a sum is at most n < 65536) and taken apart at the end
    // (16 reads in flight per wait in all three walks: they are chains of LDS round trips - round 5: 8 / 16 / 8 gave 1-3 % of
    // a tile over round 4's 4 / 4 / 4, 16 / 16 / 16 another 0.7-2.4 %)
#ifndef KT_UNROLL_A
#define KT_UNROLL_A 16
#endif
#ifndef KT_UNROLL_G
#define KT_UNROLL_G 16
#endif
#ifndef KT_UNROLL_B
#define KT_UNROLL_B 16
#endif
    {
        unsigned int gs = 0;
        const unsigned int *h = hist + (1 + wave * GB) * 32 + l31;
#pragma unroll KT_UNROLL_A
        for (int u = 0; u < GB; ++u) gs += h[u * 32];
        hist[GROUPS_AT + wave * 64 + lane] = (gs >> sh) & MSK;
    }
    __syncthreads();
    digit = 0;
    count = 0;
    row0 = 0;
    if (state) {
        row0 = (hist[l31] >> sh) & MSK;
        unsigned int run = 0, cum = 0;
        int g = 0;
#pragma unroll KT_UNROLL_G
        for (int u = 0; u < KA_WAVES; ++u) {
            run += hist[GROUPS_AT + u * 64 + lane];
            const bool le = run <= myr;
            g += le;
            cum = le ? run : cum;
        }
        g = min(g, KA_WAVES - 1);
        const unsigned int *h = hist + (1 + g * GB) * 32 + l31;
        run = cum;
        int d = 0;
#pragma unroll KT_UNROLL_B
        for (int u = 0; u < GB; ++u) {
            run += (h[u * 32] >> sh) & MSK;
            const bool le = run <= myr;
            d += le;
            cum = le ? run : cum;
        }
        d = min(d, GB - 1);
        digit = g * GB + d;
        count = (h[d * 32] >> sh) & MSK;
        myr -= cum;
    }
}

// element `myr` of the `count` (<= N) entries of list `slot` of cell `cell` in ascending order (`keep` for count == 0).  The
// lists hold the scores' raw bit patterns (the sweeping threads - all of them - do not pay for the key transform; the
// owners - ten waves, a dozen entries - do); the sort is ks_sort's in-register network, declared below
template <int N, int P, int NW> __device__ __forceinline__ void ks_sort(unsigned int (&v)[NW]);
template <int N>
__device__ __forceinline__ unsigned int kt_pick(const unsigned int *pool, unsigned int base, unsigned int count, unsigned int myr,
                                                unsigned int keep)
{
    unsigned int c[N];
    const char *p = reinterpret_cast<const char *>(pool) + base;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // (entries beyond my own count may lie in another list's segment, or - harmlessly - just beyond the pool: inside the
        // workgroup's LDS block either way; they are replaced by the padding)
        const unsigned int x = f2key(__uint_as_float(*reinterpret_cast<const unsigned int *>(p + i * 256)));
        c[i] = (unsigned)i < count ? x : 0xffffffffu;
    }
    ks_sort<N, 1, N>(c);
    // (an empty list - a constant column, settled by its window - keeps `keep` whatever its rank: without the guard a rank
    // below N picked the padding, and the cell came out NaN whenever no other cell of its tile sent the tile to the
    // streaming form: latent since round 3, found by tools/exp/smalln_probe.py)
    const unsigned int sel = count ? myr : 0u;
    unsigned int ans = count ? c[0] : keep;
#pragma unroll
    for (int i = 1; i < N; ++i) ans = sel == (unsigned)i ? c[i] : ans;
    return ans;
}

// C32: a tile is 32 cells wide and a wave reads TWO rows of it per load (lanes 0-31 row 2j, lanes 32-63 row 2j + 1, 128
// bytes each), so R = 64 registers per thread hold n <= 2048 rows: the calibration sets between 1024 and 2048 samples are
// read once too (the streaming form read them 2.13 times).  Half-width row segments cost DRAM efficiency (128 B: 0.7x,
// tools/exp/colread.hip), but one read at 0.7 beats two at 1.  Everything per cell - window, histogram (a full 32-bit
// counter per cell: the two half-waves of an instruction meet the LDS in different cycles), tags, lists - is addressed by
// cell = lane & 31; what a rank's owner computes it computes in both half-waves alike, and the lower one writes it.
//
// Rows beyond n: a thread's registers u >= its row count are set to NaN when the window is taken (their loads came back 0
// through an empty descriptor), and from there on NOTHING tests a row again: the hardware min / max skip a NaN, a NaN's
// histogram row is 0 (v_cvt_u32_f32), which no rank reads and no list wants.  Row 0 thereby counts padding + NaNs, so
// "does the column hold a NaN" is one comparison per cell (row 0 != the tile's padding) instead of a test per element.
// Registers beyond the LIVE ones (a thread's rows rounded up to whole batches) are skipped batch by batch.
template <int LOG_NB1, int R, int WGS, bool C32>
__global__ void __launch_bounds__(1024, 4 * WGS)
kth_tile_kernel(const float *__restrict__ s, int n, long long M, long long S, long long ntiles, const KAList kl,
                float *__restrict__ out, const KAPlanes pl)
{
    using Cfg = KTCfg<LOG_NB1, WGS>;
    constexpr int NB1 = Cfg::NB1, CAP = Cfg::CAP, POOL = Cfg::POOL;
    constexpr int W = C32 ? 32 : 64;                  // cells per tile
    constexpr int RPT = C32 ? 32 : KA_WAVES;          // rows between a thread's registers u and u + 1
    constexpr int Q = R / 4;
    constexpr int BATCH = Q % 8 == 0 ? 8 : Q % 6 == 0 ? 6 : 4;
    constexpr int GRP = BATCH == 6 ? 3 : 4;           // rows that share a descriptor base (host: (GRP - 1) rows' stride < 2^32 bytes)
    static_assert(Q % 2 == 0 && (R / 2) % BATCH == 0 && Q % BATCH == 0 && BATCH % GRP == 0 && BATCH % 2 == 0, "register blocks");
    // the list tag sits above the count in the histogram word (kt_narrow's MSK): a tile's rows must fit under it
    static_assert(C32 ? R * 32 <= 0xffff : R * KA_WAVES <= 0xfff + 1, "a row count must stay below the tag bits");
    __shared__ unsigned int lds[Cfg::TOTAL];
    unsigned int *hist = lds, *lists = lds + Cfg::LIST_AT, *side = lds + Cfg::SIDE_AT;
    unsigned int *win = side + Cfg::WIN_AT, *cnt = side + Cfg::CNT_AT, *ptr = side + Cfg::PTR_AT, *flg = side + Cfg::FLG_AT;
    const int tid0 = threadIdx.x, wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int nk = kl.nk;
    const int S4 = (int)(S * 4);                       // (C32: the second row of a load, in bytes; host: S < 2^29)
    const unsigned int step4 = (unsigned int)(S * 4 * RPT);      // bytes from a thread's register u to u + 1 (host: 3 of them + a row pair < 2^32)
    // LIVE registers (block-uniform): a thread's rows, rounded up to whole batches - the sweeps run over these and skip the
    // rest batch by batch (rounds 3-4 skipped the last quarter of the registers or nothing: just above a register size a
    // quarter of every sweep was padding - n = 600: 42 of 48 registers live, n = 1200: 40 of 64)
    // FIRST: the registers every n this instantiation serves fills (no test); STEP: the granularity of the skipping beyond
    // (two workgroups per CU: the last quarter or nothing - a batch there IS a quarter, and the extra tests cost 2-3 %)
    constexpr int STEP = WGS == 2 ? Q : BATCH;
    constexpr int FIRST = C32 ? R / 2 + BATCH : R / 2 + Q;
    static_assert(FIRST % BATCH == 0 && (R - FIRST) % STEP == 0 && STEP % BATCH == 0, "live-register blocks");
    const int live = max(FIRST, min(R, ((n + RPT - 1) / RPT + STEP - 1) / STEP * STEP));
    const unsigned int pads = (unsigned)(live * RPT - n);                  // padding rows per cell

    // one-time LDS state: empty window (the list fill pointers and the pool pointers are set tile by tile)
    for (int i = tid0; i < Cfg::SIDE_WORDS; i += 1024) side[i] = i < 64 ? 0xffffffffu : 0u;

    // records of the descriptor of register u (bytes readable from its base): `vb` of a row inside n, both rows of a C32 pair
    // (a thread's first R/2 registers are always rows of the tile: the host sends only R/2 < rows per thread <= R here)
    auto records = [&](int wave, int u, int vb) __attribute__((always_inline)) -> int {
        if constexpr (C32) {
            if (u < R / 2) return vb != 0 ? S4 + vb : 0;
            const int k = n - 2 * wave - 32 * u;              // rows of the pair inside n (selects, not branches)
            return (k > 0 ? vb : 0) + (k > 1 && vb != 0 ? S4 : 0);
        } else {
            if (u < R / 2) return vb;
            return n - wave > KA_WAVES * u ? vb : 0;
        }
    };
    // my byte offset inside a register's descriptor (C32: lanes beyond the tile's cells point nowhere)
    auto lane_off = [&](int lane, int vb) __attribute__((always_inline)) -> int {
        if constexpr (C32) {
            const int cb4 = (lane & 31) * 4;
            return cb4 < vb ? (lane >> 5) * S4 + cb4 : (int)0xfffffff0u;
        } else {
            return lane * 4;
        }
    };

    long long tile = blockIdx.x;
    long long plane, c0;                   // of `tile`, advanced with it
    ka_locate(pl, tile, plane, c0, W);
    float v[R];
    {
        const int vb = (int)((M - c0) * 4 < W * 4 ? (M - c0) * 4 : W * 4);
        const float *p = s + plane * pl.PS + c0 + (long long)(C32 ? 2 * wave0 : wave0) * S;
        const int loff = lane_off(tid0 & 63, vb);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            v[u] = (u < FIRST || u < live) ? kt_row(p, records(wave0, u, vb), loff) : 0.f;
            p += (long long)RPT * S;
        }
    }
    __syncthreads();

#ifdef KT_CLOCK
    int kt_it = -1;
#endif
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
#ifdef KT_CLOCK
        ++kt_it;
#endif
        KT_STAMP(0);
        const int wave = kt_opq_s(wave0);
        const bool state = wave < nk;
        const unsigned int k0 = state ? (unsigned)kl.k[wave] : 0u;

        // ---- the cell's exact window; clear the histogram meanwhile
        {
            const int lane = kt_lane(), cell = C32 ? (lane & 31) : lane;
            // registers beyond the thread's rows -> NaN (C32: the two half-waves differ by one row, a per-lane test)
            const float qnan = __uint_as_float(0x7fc00000u);
            auto pad = [&](int u) __attribute__((always_inline)) {
                if constexpr (C32) v[u] = (2 * wave + (lane >> 5) + 32 * u < n) ? v[u] : qnan;
                else v[u] = (kt_opq_s(wave) + KA_WAVES * u < n) ? v[u] : qnan;
            };
            // two rows per instruction (min3 / max3; the hardware's min / max return the other operand when one is a NaN)
            float mn = kt_min(v[0], v[1]), mx = kt_max(v[0], v[1]);
#pragma unroll
            for (int u = 2; u < R / 2; u += 2) {
                mn = kt_min3(mn, v[u], v[u + 1]);
                mx = kt_max3(mx, v[u], v[u + 1]);
            }
            // (registers below n / RPT hold a row in every thread: only the block that straddles it is padded)
            const int ufull = n / RPT;
            auto block = [&](const int u0, const int u1) __attribute__((always_inline)) {
                if (!C32 && u1 <= ufull) {                          // (32-cell tiles: measured 0.5-1.4 % slower with the test)
#pragma unroll
                    for (int u = u0; u < u1; u += 2) {
                        mn = kt_min3(mn, v[u], v[u + 1]);
                        mx = kt_max3(mx, v[u], v[u + 1]);
                    }
                } else {
#pragma unroll
                    for (int u = u0; u < u1; u += 2) {
                        pad(u);
                        pad(u + 1);
                        mn = kt_min3(mn, v[u], v[u + 1]);
                        mx = kt_max3(mx, v[u], v[u + 1]);
                    }
                }
            };
            block(R / 2, FIRST);
#pragma unroll
            for (int u0 = FIRST; u0 < R; u0 += STEP)
                if (u0 < live) block(u0, u0 + STEP);
            // (a column of NaNs leaves a NaN, whose key is above every number's: "not a finite window" below)
            atomicMin(&win[cell], f2key(mn));
            atomicMax(&win[64 + cell], f2key(mx));
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
            for (int i = wave * 64 + lane; i < ((NB1 + 1) * 32) / 4; i += 1024) h4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        KT_STAMP(1);
        unsigned int kmin, kmax;
        float vlo, sf;
        bool flat, badwin;
        {
            const int lane = kt_lane(), cell = C32 ? (lane & 31) : lane;
            kmin = win[cell];
            kmax = win[64 + cell];
            vlo = key2f(kmin);
            const float r = key2f(kmax) - vlo;
            const bool finite_lo = fabsf(vlo) < __builtin_inff();
            // a constant column (its numbers; NaNs are counted apart) is settled by the window alone
            flat = r == 0.f && finite_lo;
            sf = ((float)(NB1 - 1) / r) * 0.999999f;
            // (an infinite or all-NaN column has r = NaN: not flat, not finite)
            badwin = !flat && (!(r < __builtin_inff()) || !finite_lo || !(sf < __builtin_inff()));
            if (flat || badwin) sf = 0.f;
        }

        // ---- first digit: NB1 - 1 value-linear buckets over the window (rows 1 .. NB1-1; row 0: padding and NaNs)
        // (keeping the row numbers for the second sweep, two per register, was measured at R = 32 with one workgroup per
        // CU: no gain - the sweeps were latency-, not instruction-bound - and with two per CU there are no registers for it)
        {
            const int lane = kt_lane();
            const unsigned int inc = C32 ? 1u : Ctr<false>::inc(lane);
            char *hb = reinterpret_cast<char *>(hist + (lane & 31));
#pragma unroll
            for (int u = 0; u < FIRST; ++u) {
                const unsigned int row = kt_frow(v[u], sf, vlo);
                atomicAdd(reinterpret_cast<unsigned int *>(hb + (row << 7)), inc);          // word row * 32 + (lane & 31)
            }
#pragma unroll
            for (int u0 = FIRST; u0 < R; u0 += STEP) {
                if (u0 < live) {
#pragma unroll
                    for (int u = u0; u < u0 + STEP; ++u) {
                        const unsigned int row = kt_frow(v[u], sf, vlo);
                        atomicAdd(reinterpret_cast<unsigned int *>(hb + (row << 7)), inc);
                    }
                }
            }
        }
        __syncthreads();
        KT_STAMP(2);
        unsigned int myr = k0, count, fl;
        int digit, myslot = wave, cmax = 0;
        bool open, nancell;
        {
            const int lane = kt_lane(), cell = C32 ? (lane & 31) : lane;
            if (wave == KA_WAVES - 1) {                              // (everyone has read the window)
                win[cell] = 0xffffffffu;
                win[64 + cell] = 0u;
                ptr[cell] = 0u;                                      // this tile's pool is empty (allocations follow the next barrier)
            }
            unsigned int row0;
            kt_narrow<LOG_NB1, C32>(hist, state, myr, digit, count, row0, lane, wave);
            // np.quantile: a NaN anywhere in the column makes every quantile of the cell NaN.  (Exact whenever the window is
            // finite: then every number maps to a row >= 1; a tile with a cell whose window is not is redone anyway.)
            nancell = state && row0 != pads;
            open = state && !flat && !nancell;
            bool many = open && count > (unsigned)CAP;
            const bool bad = state && badwin;
            // The list of a rank = the list of the FIRST owner to tag the rank's row in its cell.  An owner whose row carries
            // no tag yet takes `count` entries of the cell's pool (the histogram knows the size: nothing can overflow a
            // list, whatever the other cells of the tile decide), points the list's fill pointer at them and writes
            // "list + 1" above the row's count with a compare-and-swap (bits 12-15 of my cell's half of the word, bits 16-19
            // of a 32-cell tile's) - unless another rank of the cell got there first: then that rank's list is mine too (and
            // my entries stay unused).  (C32: the two half-waves hold the same cells; the lower one acts.)
            const int tsh = C32 ? 16 : 12 + 16 * (lane >> 5);
            unsigned int *a = hist + (digit + 1) * 32 + (lane & 31);
            if (open && !many && (!C32 || lane < 32)) {
                // the tag is won FIRST and only the winner takes entries (round-5 advice: allocating before the
                // compare-and-swap charged the pool once per rank instead of once per distinct row - spurious exhaustion
                // on quantised data, dependent on wave timing); `cnt` and the tags are read after the next barrier
                unsigned int old = *a;
                bool won = false;
                while (((old >> tsh) & 15u) == 0u) {
                    const unsigned int prev = atomicCAS(a, old, old | ((unsigned)(wave + 1) << tsh));
                    if (prev == old) { won = true; break; }          // (else: another rank's of my cell, or ...
                    old = prev;                                      //  ... the other cell of the word was tagged meanwhile)
                }
                if (won) {
                    const unsigned int start = atomicAdd(&ptr[cell], count);
                    if (start + count > (unsigned)POOL) many = true; // (the pool is exhausted: ties - left to the streaming form)
                    else cnt[wave * 64 + cell] = (start * 64u + (unsigned)cell) * 4u;
                }
            }
            const unsigned int w = (__ballot(many) != 0 ? 1u : 0u) | (__ballot(bad) != 0 ? 2u : 0u);
            if (lane == 0) flg[wave] = w;
            cmax = open ? (int)count : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
            cmax = __builtin_amdgcn_readfirstlane(cmax);
            __syncthreads();                                        // the histograms are read, rows tagged, flags published
            fl = flg[lane & (KA_WAVES - 1)];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) fl |= __shfl_xor(fl, o);
            fl = __builtin_amdgcn_readfirstlane(fl);
            // whose list holds my rank's candidates (0 when the tile is not being finished and my row went untagged)
            myslot = open ? (int)((*a >> tsh) & 15u) - 1 : wave;
        }
        KT_STAMP(3);

        const bool more = tile + gridDim.x < ntiles;               // (block-uniform)
        long long nplane = plane, nc0 = c0;
        ka_advance(pl, nplane, nc0, W);
        const int nvb = more ? (int)((M - nc0) * 4 < W * 4 ? (M - nc0) * 4 : W * 4) : 0;
        const float *nbase = s + nplane * pl.PS + nc0 + (long long)(C32 ? 2 * wave : wave) * S;   // my first row of the next tile
        float *outp = out + plane * pl.OPS;

        // ---- collect + pick (when every pair has its <= CAP candidates: ok, block-uniform)
        const bool ok = fl == 0u;
        KT_STAMP(4);
        {
            // (the sweep is opaque to the optimiser: left alone it keeps the 64 row numbers of the first sweep for this
            // one - in scratch memory, whose loads then queue behind the prefetch below and make every element wait)
            float sf2 = sf, vlo2 = vlo;
            asm volatile("" : "+v"(sf2), "+v"(vlo2));
            const int lane = kt_lane(), cell = C32 ? (lane & 31) : lane;
            const int nloff = lane_off(lane, nvb);
            const float *np = nbase;
            const char *hb = reinterpret_cast<const char *>(hist + (lane & 31));     // the word of (row, my cell): hb + row * 128
            const unsigned int tsh = C32 ? 16u : 12u + 16u * (unsigned)(lane >> 5);
            char *cb = reinterpret_cast<char *>(cnt + cell) - 256;             // fill pointer of list m - 1 of my cell: cb + m * 256
            char *lb = reinterpret_cast<char *>(lists);
            char *dummy = reinterpret_cast<char *>(side + Cfg::SIDE_WORDS + lane);      // (64 scratch words behind the side arrays)
            // BATCH rows at a time: their histogram words are read together (one LDS latency per batch, not per element); an
            // element whose row is tagged joins the list - a position below CAP always: the histogram counted the list's
            // elements.  No row is tested: padding and NaNs look up row 0, which is never tagged, and a tile that is not
            // being finished has its tagged lists filled for nothing.
            auto batch = [&](const int u0) __attribute__((always_inline)) {
                unsigned int m[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const unsigned int row = kt_frow(v[u0 + i], sf2, vlo2);
                    m[i] = *reinterpret_cast<const unsigned int *>(hb + (row << 7));
                }
                // (opaque AFTER the batch's loads are issued - the same barrier inside the loop above made every read wait
                // for the one before it)
                if constexpr (BATCH == 8)
                    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]));
                else if constexpr (BATCH == 6)
                    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]));
                else
                    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]));
                // (round 4 measured the batch's returning adds issued together, each under its own test: eight more live
                // registers, spills under the 64-register cap of the two-workgroup form, no gain with one)
                // branch-free (round 5: -1..-4 % of a tile over a test, a returning add, a wait and a store per element): every
                // element issues its returning add - of 0 for an unwanted one, on a word of the window array, which ignores it -
                // and its store - to a scratch word of its lane: the batch's adds are in flight together, ONE wait per batch.
                // A wanted element joins its list at a position inside the list's segment always: the histogram counted them.
                bool hit[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    m[i] = __builtin_amdgcn_ubfe(m[i], tsh, 4u);
                    hit[i] = m[i] != 0u;
                    // (the list's fill pointer IS the byte offset of its next free entry, 256 bytes = one entry across the
                    // cells further each time; raw bits: the owners transform the few they pick from)
                    m[i] = atomicAdd(reinterpret_cast<unsigned int *>(cb + (m[i] << 8)), hit[i] ? 256u : 0u);
                }
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int u = u0 + i;
                    *reinterpret_cast<float *>(hit[i] ? lb + m[i] : dummy) = v[u];
                    // the next tile, row by row into the register just consumed.  An unconditional load - rows beyond n, or
                    // beyond the last tile, through an empty descriptor - and the ONLY one in the loop, whether the tile is
                    // being finished or not: no copy has to wait for it here, no second definition to reconcile
                    v[u] = kt_row_s(np, records(wave, u, nvb), nloff, (unsigned int)(i % GRP) * step4);
                    if (i % GRP == GRP - 1) np += (long long)GRP * RPT * S;
                }
            };
#pragma unroll
            for (int u0 = 0; u0 < FIRST; u0 += BATCH) batch(u0);
#pragma unroll
            for (int u0 = FIRST; u0 < R; u0 += STEP) {
                if (u0 < live) {
#pragma unroll
                    for (int u1 = u0; u1 < u0 + STEP; u1 += BATCH) batch(u1);
                }
            }
        }
        lds_barrier();
        KT_STAMP(5);
        if (ok) {
            const int lane = kt_lane(), cell = C32 ? (lane & 31) : lane;
            unsigned int ans = kmin;                                // flat: the column's one value
            if (wave < nk) {                                        // (whole waves: cmax is theirs)
                // my rank among the <= CAP candidates of my list: all of them into registers at once (entries beyond my
                // own count are replaced by all ones), a sorting network, element myr.  The list's fill pointer has reached
                // its end: the first entry lies `count` entries before it.
                const unsigned int cn = open ? count : 0u;
                const unsigned int base = open ? cnt[myslot * 64 + cell] - cn * 256u : 0u;
                if (cmax <= 8) ans = kt_pick<8>(lists, base, cn, myr, ans);
                else if (cmax <= 16) ans = kt_pick<16>(lists, base, cn, myr, ans);
                else if constexpr (WGS == 1) ans = kt_pick<32>(lists, base, cn, myr, ans);
                else if (open) {                                    // (no 32 registers to spare: count in place)
                    const char *lp = reinterpret_cast<const char *>(lists) + base;
                    ans = 0xffffffffu;
                    for (int i = 0; i < (int)count; ++i) {
                        const unsigned int ki = f2key(__uint_as_float(*reinterpret_cast<const unsigned int *>(lp + i * 256)));
                        unsigned int le = 0;
                        for (int j = 0; j < (int)count; ++j)
                            le += f2key(__uint_as_float(*reinterpret_cast<const unsigned int *>(lp + j * 256))) <= ki;
                        if (le > myr) ans = min(ans, ki);
                    }
                }
            }
            const long long c = c0 + cell;
            if (state && c < M && (!C32 || lane < 32))
                outp[(long long)kl.o[wave] * pl.OS + c] = nancell ? __uint_as_float(0x7fc00000u) : key2f(ans);
        } else if (wave == 0 && kt_lane() == 0) {
            // not finished by the fast form (a bucket above CAP: ties, an outlier stretching the window; an infinite
            // window): the tile is MARKED - a NaN pattern no result can have, in the first rank's output of its first
            // cell - and redone by the streaming form after this loop (in the loop its code would compete with the 64
            // data registers)
            outp[(long long)kl.o[0] * pl.OS + c0] = __uint_as_float(KT_MARK);
        }
        // NO barrier here: the pool and its pointers are not touched again before the next tile's narrowing, three barriers
        // from now (the tags go with that tile's histogram clear), and the waves that own no rank are already taking its window
        KT_STAMP(6);
        plane = nplane;
        c0 = nc0;
    }
    // ---- my own marked tiles, by the streaming form above: v[] is dead, its registers are free for it (64 marks per
    // load, lane = one of my tiles; the result of a tile overwrites its mark).  (C32: the streaming form's 64 lanes see
    // the row as ending with the tile's 32 cells.)
    __syncthreads();
    {
        const int lane = threadIdx.x & 63;
        const float *marks = out + (long long)kl.o[0] * pl.OS;
        const long long mine = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;      // tiles blockIdx.x + i gridDim.x
#pragma unroll 1
        for (long long ib = 0; ib < mine; ib += 64) {
            const long long t = blockIdx.x + (ib + lane) * gridDim.x;
            bool mk = false;
            if (ib + lane < mine) {
                long long tp, tc0;
                ka_locate(pl, t, tp, tc0, W);
                mk = __float_as_uint(marks[tp * pl.OPS + tc0]) == KT_MARK;
            }
            unsigned long long todo = __ballot(mk);
#pragma unroll 1
            while (todo) {
                const int b = __builtin_ctzll(todo);
                todo &= todo - 1;
                long long mp, mc0;
                ka_locate(pl, blockIdx.x + (ib + b) * gridDim.x, mp, mc0, W);
                const long long mend = mc0 + W < M ? mc0 + W : M;
                ka_tile<9, false>(s + mp * pl.PS, n, mend, S, mc0, kl, 1, out + mp * pl.OPS, pl.OS, lds);
                __syncthreads();
            }
        }
    }
}

// ---- small calibration sets (n <= 128; the reference scripts use n_cal = 100 and 1000): the whole column of a
// cell fits in ITS LANE's registers.  One wave = 64 adjacent cells, lane = cell: n coalesced row loads (256 B
// each, all issued before the first is used), keys padded with all ones to N = 64 / 128, a fully unrolled sorting
// network of v_min_u32 / v_max_u32 on the register array (Batcher's odd-even merge sort: 543 / 1471 compare-exchanges),
// and the requested ranks - wave-uniform - are read with register-relative addressing.  (N = 256 was measured too:
// 256 registers per lane leave one wave per SIMD: 1.73 ms against 1.93 ms for the radix form on [256, 2.6M], for
// 40 s more compile time - not instantiated.)  No LDS, no barrier, no atomics, one read
// of the scores; the radix machinery above spends ~1500 instructions per thread on per-tile set-up alone, which at
// n = 100 is 6 elements per thread.
// The network is Batcher's odd-even merge sort: 19 / 63 / 191 / 543 / 1471 compare-exchanges for N = 8 / 16 / 32 / 64 / 128
// (the bitonic network of rounds 1-3: 24 / 80 / 240 / 672 / 1792), every one ascending.  One stage = the exchanges with
// partner distance K while runs of length 2P are being merged, fully unrolled; the stages are chained by template
// recursion because the optimizer refuses to unroll the loop nest as a whole.
// NW <= N wires: the network for N = 2^k inputs WITHOUT the exchanges that touch a wire >= NW.  Those wires would hold the
// padding (all ones, the largest key), every exchange is ascending, so none of them would ever move anything: what is left
// sorts NW inputs (n = 100 in 104 wires: 1157 exchanges of the 1471).
template <int N, int P, int K, int NW>
__device__ __forceinline__ void ks_stage(unsigned int (&v)[NW])
{
#pragma unroll
    for (int x = 0; x < NW; ++x) {
        const int y = x - K % P;
        if (y >= 0 && y % (2 * K) < K && x + K < NW && x / (2 * P) == (x + K) / (2 * P)) {
            // (as asm: left as umin / umax, LLVM's n-ary reassociation pass spends minutes on a 1471-exchange network)
            unsigned int lo, hi;
            asm("v_min_u32 %0, %1, %2" : "=v"(lo) : "v"(v[x]), "v"(v[x + K]));
            asm("v_max_u32 %0, %1, %2" : "=v"(hi) : "v"(v[x]), "v"(v[x + K]));
            v[x] = lo;
            v[x + K] = hi;
        }
    }
}
template <int N, int P, int K, int NW>
__device__ __forceinline__ void ks_merge(unsigned int (&v)[NW])
{
    ks_stage<N, P, K, NW>(v);
    if constexpr (K > 1) ks_merge<N, P, K / 2, NW>(v);
}
// (second parameter: the run length merged first; callers start at 1)
template <int N, int P, int NW = N>
__device__ __forceinline__ void ks_sort(unsigned int (&v)[NW])
{
    ks_merge<N, P, P, NW>(v);
    if constexpr (2 * P < N) ks_sort<N, 2 * P, NW>(v);
}
// v[k] for a wave-uniform k: the registers are viewed as 32-wide vectors, whose dynamic extract with a uniform index
// lowers to register-relative addressing (s_set_gpr_idx / v_movrels) - a handful of instructions per rank instead
// of a compare-and-select per register
typedef unsigned int ks_u32x32 __attribute__((ext_vector_type(32)));
typedef unsigned int ms_u32x2_k __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ unsigned int ks_take(const unsigned int (&v)[N], int k)
{
    unsigned int r = 0u;
#pragma unroll
    for (int g = 0; g < (N + 31) / 32; ++g)
        if ((k >> 5) == g) {                                      // wave-uniform
            ks_u32x32 x;
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = 32 * g + i < N ? v[32 * g + i] : 0u;
            r = x[k & 31];
        }
    return r;
}

// N = the registers (wires) of a lane: n rounded up to a multiple of 8
template <int N>
__global__ void __launch_bounds__(256) kth_small_kernel(const float *__restrict__ s, int n, long long M, long long S, const KAList kl,
                                                       float *__restrict__ out, const KAPlanes pl, long long ntiles)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;            // whole wave beyond the last tile
    long long plane, c0;
    ka_locate(pl, tile, plane, c0);
    const long long c = c0 + lane;
    s += plane * pl.PS;
    out += plane * pl.OPS;
    const bool cok = c < M;
    // all row loads first, unconditionally (rows beyond n re-read row n-1, lanes beyond M are dropped by the buffer's
    // range check): a load inside its own `if` is waited for on the spot, and 128 serialised HBM latencies are what
    // the kernel then costs
    const int valid = (int)((M - c0) * 4 < 256 ? (M - c0) * 4 : 256);
    const float *p = s + c0;
    float raw[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, valid, 0x00020000);
        raw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, 0, 0));
        p += (i + 1 < n) ? S : 0;
    }
    unsigned int v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = i < n ? f2key(raw[i]) : 0xffffffffu;
    ks_sort<(N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : N <= 64 ? 64 : N <= 128 ? 128 : 256), 1, N>(v);
    // np.quantile: a NaN in the column makes every quantile of the cell NaN.  Sorted by key, positive NaNs sit above
    // +inf at the top of the n real entries and negative ones below -inf at the bottom
    const bool nan = ks_take<N>(v, n - 1) > 0xff800000u || v[0] < 0x007fffffu;
#pragma unroll
    for (int j = 0; j < KA_MAXK; ++j) {
        if (j >= kl.nk) break;                                    // wave-uniform
        const unsigned int r = ks_take<N>(v, kl.k[j]);
        if (cok) out[(long long)kl.o[j] * pl.OS + c] = nan ? __uint_as_float(0x7fc00000u) : key2f(r);
    }
}

// ---- 168 < n <= 240: TWO lanes per cell, each sorting half the column in its registers (round 2 had this shape with
// bitonic networks; round 3 replaced it by the 16-row register tiles, 2.3-3.0 TB/s; with the pruned Batcher networks it is
// back).  A wave = 32 cells, lane l and l + 32 the two halves of cell l & 31 (a load reads two 128-byte runs, n0 rows
// apart).  The lower lane sorts rows [0, n0) ascending; the upper lane rows [n0, n) ascending in the COMPLEMENTED key domain,
// i.e. descending - so that register i of the pair holds A_i and B_(last - i), the partners of the first step of a
// bitonic merge, and one exchange (v_permlane32_swap) per register gives both lanes "own" and "partner".  After
// r_i = min(own_i, ~partner_i) - the same expression in both lanes' own domains - the lower lane holds the smaller half of
// the union and the upper lane the larger half (complemented), each a bitonic sequence that a pruned bitonic merge sorts
// descending.  Padding: wires >= NW are virtual; lower-lane padding is +inf, upper-lane padding -inf (complemented: the
// maximum of its domain, as the ascending network wants), which the rank arithmetic at the end counts off.
template <int NW, int J>
__device__ __forceinline__ void kp_merge_desc(unsigned int (&v)[NW])
{
#pragma unroll
    for (int x = 0; x < NW; ++x)
        if ((x & J) == 0 && x + J < NW) {                        // (a partner >= NW is a virtual minimum: the exchange would move nothing)
            unsigned int lo, hi;
            asm("v_min_u32 %0, %1, %2" : "=v"(lo) : "v"(v[x]), "v"(v[x + J]));
            asm("v_max_u32 %0, %1, %2" : "=v"(hi) : "v"(v[x]), "v"(v[x + J]));
            v[x] = hi;
            v[x + J] = lo;
        }
    if constexpr (J > 1) kp_merge_desc<NW, J / 2>(v);
}

template <int NW>
__global__ void __launch_bounds__(256) kth_pair_kernel(const float *__restrict__ s, int n, long long M, long long S, const KAList kl,
                                                      float *__restrict__ out, const KAPlanes pl, long long ntiles)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;            // whole wave beyond the last tile
    long long plane, c0;
    ka_locate(pl, tile, plane, c0, 32);
    const int cell = lane & 31, h = lane >> 5;
    const long long c = c0 + cell;
    s += plane * pl.PS;
    out += plane * pl.OPS;
    const bool cok = c < M;
    const int n0 = (n + 1) / 2;                                   // rows of the lower lane; the upper lane has n - n0 (n0 or n0 - 1)
    const int nmine = h ? n - n0 : n0;
    const unsigned int hm = h ? 0xffffffffu : 0u;                 // the upper lane works on complemented keys
    // all row loads first (cf. kth_small_kernel): register i = row i of the lower half and row n0 + i of the upper half
    const int vb = (int)((M - c0) * 4 < 128 ? (M - c0) * 4 : 128);
    const int hoff = (int)((long long)n0 * S * 4);                // (host: < 2^31)
    const int loff = cell * 4 < vb ? h * hoff + cell * 4 : (int)0xfffffff0u;
    const float *p = s + c0;
    float raw[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int rec = i < n - n0 ? hoff + vb : i < n0 ? vb : 0;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, rec, 0x00020000);
        raw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, loff, 0, 0));
        p += (i + 1 < n0) ? S : 0;
    }
    unsigned int v[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) v[i] = i < nmine ? f2key(raw[i]) ^ hm : 0xffffffffu;
    ks_sort<128, 1, NW>(v);
    // first step of the bitonic merge across the lane pair, then each lane's half sorted descending in its own domain
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const ms_u32x2_k r = __builtin_amdgcn_permlane32_swap(v[i], v[i], false, false);     // .x: the lower lane's, .y: the upper lane's
        unsigned int a = r.x ^ hm, b = r.y ^ ~hm;                 // lower: (own, ~partner); upper: (~partner, own)
        asm("v_min_u32 %0, %1, %2" : "=v"(v[i]) : "v"(a), "v"(b));
    }
    kp_merge_desc<NW, 64>(v);
    // Ascending order of the union of 256 logical values: the lower lane's virtual -inf wires, its registers from NW - 1
    // down to 0, the upper lane's registers 0 .. NW - 1 (complemented), its virtual +inf wires.  The upper lane's pU padding
    // rows (-inf) lie at the bottom of the lower lane's registers: rank k of the column is position k + pU of the registers.
    const int pU = NW - (n - n0);
    // np.quantile: a NaN in the column makes every quantile of the cell NaN: the smallest score (lower lane) a negative NaN
    // or the largest (upper lane, position n0 - 1 + NW - NW) a positive one
    const unsigned int klow = ks_take<NW>(v, NW - 1 - pU), khigh = ~ks_take<NW>(v, n0 - 1);
    const unsigned long long nb = __ballot(h ? khigh > 0xff800000u : klow < 0x007fffffu);
    const bool nan = ((nb >> cell) | (nb >> (cell + 32))) & 1ull;
#pragma unroll
    for (int j = 0; j < KA_MAXK; ++j) {
        if (j >= kl.nk) break;                                    // wave-uniform
        const int u = kl.k[j] + pU;                               // (wave-uniform)
        const bool upper = u >= NW;
        const unsigned int r = upper ? ~ks_take<NW>(v, u - NW) : ks_take<NW>(v, NW - 1 - u);
        if (cok && (h != 0) == upper) out[(long long)kl.o[j] * pl.OS + c] = nan ? __uint_as_float(0x7fc00000u) : key2f(r);
    }
}

template <int NW>
int launch_kth_pair(const float *scores, int n, long long M, long long S, const int32_t *ks, const int32_t *rows, int nk, float *out,
                    const KAPlanes &pl0, long long planes, hipStream_t st)
{
    KAPlanes pl = pl0;
    pl.tpp = (M + 31) / 32;
    const long long tiles = pl.tpp * planes, blocks = (tiles + 3) / 4;
    if (blocks > 0x7fffffffLL) return PRE_E_SHAPE;
    for (int j0 = 0; j0 < nk; j0 += KA_MAXK) {
        KAList kl;
        kl.nk = (nk - j0) < KA_MAXK ? (nk - j0) : KA_MAXK;
        for (int j = 0; j < KA_MAXK; ++j) { kl.k[j] = j < kl.nk ? ks[j0 + j] : 0; kl.o[j] = j < kl.nk ? rows[j0 + j] : 0; }
        hipLaunchKernelGGL((kth_pair_kernel<NW>), dim3((unsigned)blocks), dim3(256), 0, st, scores, n, M, S, kl, out, pl, tiles);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

template <int N>
int launch_kth_small(const float *scores, int n, long long M, long long S, const int32_t *ks, const int32_t *rows, int nk, float *out,
                     const KAPlanes &pl, long long planes, hipStream_t st)
{
    const long long tiles = pl.tpp * planes, blocks = (tiles + 3) / 4;
    if (blocks > 0x7fffffffLL) return PRE_E_SHAPE;
    for (int j0 = 0; j0 < nk; j0 += KA_MAXK) {
        KAList kl;
        kl.nk = (nk - j0) < KA_MAXK ? (nk - j0) : KA_MAXK;
        for (int j = 0; j < KA_MAXK; ++j) { kl.k[j] = j < kl.nk ? ks[j0 + j] : -1; kl.o[j] = j < kl.nk ? rows[j0 + j] : 0; }
        hipLaunchKernelGGL((kth_small_kernel<N>), dim3((unsigned)blocks), dim3(256), 0, st, scores, n, M, S, kl, out, pl, tiles);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

template <int LOG_NB1, bool WIDE, int LSX = 0>
int launch_kth(const float *scores, int n, long long M, long long S, const int32_t *ks, const int32_t *rows, int nk, float *out,
               const KAPlanes &pl, long long planes, hipStream_t st)
{
    const long long tiles = pl.tpp * planes;
    const long long per_launch = 1LL << 21;                     // x 1024 threads: the dispatch packet counts work-items in 32 bits
    for (int j0 = 0; j0 < nk; j0 += KA_MAXK) {
        KAList kl;
        kl.nk = (nk - j0) < KA_MAXK ? (nk - j0) : KA_MAXK;
        for (int j = 0; j < KA_MAXK; ++j) { kl.k[j] = j < kl.nk ? ks[j0 + j] : 0; kl.o[j] = j < kl.nk ? rows[j0 + j] : 0; }
        for (long long t0 = 0; t0 < tiles; t0 += per_launch) {
            const long long nt = tiles - t0 < per_launch ? tiles - t0 : per_launch;
            // the fast first digit pays while a full bucket holds well under CAP elements (n <= ~6 NB1 on
            // bell-shaped scores with 31-entry lists, 12 NB1 with the tag-array form's 63); beyond that it would be a wasted sweep
            hipLaunchKernelGGL((kth_axis0_kernel<LOG_NB1, WIDE, LSX>), dim3((unsigned)nt), dim3(1024), 0, st, scores, n, M, S, t0, kl,
                               n <= (LSX ? 12 : 6) * (1 << LOG_NB1) ? 1 : 0, out, pl);
            PRE_LAUNCH_CHECK();
        }
    }
    return PRE_OK;
}

template <int LOG_NB1, int R, int WGS, bool C32 = false>
int launch_kth_tile(const float *scores, int n, long long M, long long S, const int32_t *ks, const int32_t *rows, int nk, float *out,
                    const KAPlanes &pl0, long long planes, hipStream_t st)
{
    KAPlanes pl = pl0;
    if (C32) pl.tpp = (M + 31) / 32;
    const long long tiles = pl.tpp * planes;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long long grid = tiles < WGS * cus ? tiles : WGS * cus;   // persistent workgroups: WGS per CU
    for (int j0 = 0; j0 < nk; j0 += KA_MAXK) {
        KAList kl;
        kl.nk = (nk - j0) < KA_MAXK ? (nk - j0) : KA_MAXK;
        for (int j = 0; j < KA_MAXK; ++j) { kl.k[j] = j < kl.nk ? ks[j0 + j] : 0; kl.o[j] = j < kl.nk ? rows[j0 + j] : 0; }
        KAPlanes pg = pl;
        pg.sp = grid / pl.tpp;
        pg.sc = grid % pl.tpp;
        hipLaunchKernelGGL((kth_tile_kernel<LOG_NB1, R, WGS, C32>), dim3((unsigned)grid), dim3(1024), 0, st, scores, n, M, S, tiles, kl, out, pg);
        PRE_LAUNCH_CHECK();
    }
    return PRE_OK;
}

}  // namespace

extern "C" int pre_kth_axis0_planes_f32(const float *scores, int64_t plane_stride, int64_t row_stride, int64_t planes, int64_t n,
                                        int64_t M, const int32_t *ks, int nk, float *out, int64_t out_rank_stride,
                                        int64_t out_plane_stride, void *stream)
{
    if (!scores || !ks || !out || n <= 0 || M <= 0 || nk <= 0 || planes <= 0) return PRE_E_NULL;
    if (row_stride < M || out_rank_stride < M || plane_stride < 0 || out_plane_stride < 0) return PRE_E_RANGE;
    if (planes > 1 && (plane_stride < M || out_plane_stride < M)) return PRE_E_RANGE;
    const long long S = (long long)row_stride;
    if (n > 0x7fffffff || nk > 64) return PRE_E_SHAPE;
    KAPlanes pl;
    pl.tpp = ((long long)M + KA_W - 1) / KA_W;
    pl.PS = (long long)plane_stride;
    pl.OS = (long long)out_rank_stride;
    pl.OPS = (long long)out_plane_stride;
    pl.sp = pl.sc = 0;                     // (set by the persistent launcher)
    if (pl.tpp > 0x7fffffffLL || planes > 0x7fffffffLL || pl.tpp * planes > (1LL << 40)) return PRE_E_SHAPE;
    // the kernels want ascending ranks (their slots rely on it): sort here, each result goes to its caller's row
    int32_t sk[64], rows[64];
    for (int j = 0; j < nk; ++j) {
        if (ks[j] < 0 || ks[j] >= n) return PRE_E_RANGE;
        int i = j;
        for (; i > 0 && sk[i - 1] > ks[j]; --i) { sk[i] = sk[i - 1]; rows[i] = rows[i - 1]; }
        sk[i] = ks[j];
        rows[i] = j;
    }
    ks = sk;
    hipStream_t st = as_stream(stream);
    const long long P = (long long)planes;
#define KA_ARGS scores, (int)n, (long long)M, S, ks, rows, nk, out, pl, P, st
    // n <= 128: the column sorted in its lane's registers, one instantiation per 8 rows (its network has the exchanges of
    // its own wires only)
    switch ((n + 7) / 8) {
    case 1: return launch_kth_small<8>(KA_ARGS);
    case 2: return launch_kth_small<16>(KA_ARGS);
    case 3: return launch_kth_small<24>(KA_ARGS);
    case 4: return launch_kth_small<32>(KA_ARGS);
    case 5: return launch_kth_small<40>(KA_ARGS);
    case 6: return launch_kth_small<48>(KA_ARGS);
    case 7: return launch_kth_small<56>(KA_ARGS);
    case 8: return launch_kth_small<64>(KA_ARGS);
    case 9: return launch_kth_small<72>(KA_ARGS);
    case 10: return launch_kth_small<80>(KA_ARGS);
    case 11: return launch_kth_small<88>(KA_ARGS);
    case 12: return launch_kth_small<96>(KA_ARGS);
    case 13: return launch_kth_small<104>(KA_ARGS);
    case 14: return launch_kth_small<112>(KA_ARGS);
    case 15: return launch_kth_small<120>(KA_ARGS);
    case 16: return launch_kth_small<128>(KA_ARGS);
    case 17: return launch_kth_small<136>(KA_ARGS);
    case 18: return launch_kth_small<144>(KA_ARGS);
    case 19: return launch_kth_small<152>(KA_ARGS);
    case 20: return launch_kth_small<160>(KA_ARGS);
    case 21: return launch_kth_small<168>(KA_ARGS);
    default: break;
    }
    // 168 < n <= 240: two lanes per cell, half the column each (the second half of a load is addressed by a 32-bit byte
    // offset: n0 rows < 2^31 bytes).  (Measured against the 16-row register tiles: 3.6 vs 2.3 TB/s at n = 170, 3.2 vs 2.4 at
    // 200, 3.05 vs ~2.7 at 230; at 256 - 128 wires, 198 registers, two waves per SIMD - 2.86 vs 2.98: the tiles keep 241-256.)
    if (n <= 240 && (long long)((n + 1) / 2) * S * 4 < (1LL << 31)) {
        switch (((n + 1) / 2 + 7) / 8) {
        case 11: return launch_kth_pair<88>(KA_ARGS);
        case 12: return launch_kth_pair<96>(KA_ARGS);
        case 13: return launch_kth_pair<104>(KA_ARGS);
        case 14: return launch_kth_pair<112>(KA_ARGS);
        case 15: return launch_kth_pair<120>(KA_ARGS);
        default: break;
        }
    }
    // 128 < n <= 1024: the tile in registers, read once (16, 24, 32 rows per thread: two workgroups per CU; 48, 64: one).
    // Every instantiation serves R/2 < rows per thread <= R (its first R/2 rows need no "is this row below n" test)
    // (rows up to three register steps apart share a descriptor base and are told apart by a 32-bit scalar byte offset: 48 S x 4
    // bytes < 2^32, i.e. rows of up to 22 M cells; beyond, the streaming form)
    const bool tile_ok = S * 192 + 256 < (1LL << 32);
#ifndef KT_NB_SMALL
#define KT_NB_SMALL 8
#endif
    if (n <= 256 && tile_ok) return launch_kth_tile<KT_NB_SMALL, 16, 2>(KA_ARGS);
    if (n <= 384 && tile_ok) return launch_kth_tile<KT_NB_SMALL, 24, 2>(KA_ARGS);
    if (n <= 512 && tile_ok) return launch_kth_tile<8, 32, 2>(KA_ARGS);
    if (n <= 768 && tile_ok) return launch_kth_tile<9, 48, 1>(KA_ARGS);
    if (n <= 1024 && tile_ok) return launch_kth_tile<9, 64, 1>(KA_ARGS);
    // 16-bit counters hold n < 65536; 1024 first-digit buckets (one workgroup per CU) pay off once 512 buckets
    // would leave more than CAP elements per bucket (n above ~2000)
    // 1024 < n <= 2048: 32-cell tiles, two rows per load, still in registers and read once (S < 2^29: the second row of a
    // load is addressed by a 32-bit byte offset)
#ifndef KA_NO_C32
    if (n <= 2048 && S * 388 + 128 < (1LL << 32)) return launch_kth_tile<9, 64, 1, true>(KA_ARGS);     // (... 96 S x 4 bytes + a row pair: 11 M cells)
#endif
    if (n >= 65536) return launch_kth<9, true>(KA_ARGS);
    // 4096 < n <= 12288: the tag-array form (lists of up to 63 from a pool of 448 per cell; round 5 - rounds 3-4: 47-entry
    // lists and a bitmap to 9216, the general form beyond; measured: +20-25 % to 9216, +33-63 % to 12000, -6 % at 14336,
    // where buckets above 63 send too many tiles to the general form)
    if (n > 4096 && n <= 12288) return launch_kth<10, false, 48>(KA_ARGS);
    if (n > 2048) return launch_kth<10, false>(KA_ARGS);
    return launch_kth<9, false>(KA_ARGS);
#undef KA_ARGS
}

#ifdef KA_DEBUG_FAIL
extern "C" int pre_debug_ka_fail(void *dst)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ka_debug_fail), sizeof(ka_debug_fail));
}
#endif
#ifdef KT_CLOCK
extern "C" int pre_debug_kt_clock(void *dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(kt_clock_buf), bytes < sizeof(kt_clock_buf) ? bytes : sizeof(kt_clock_buf));
}
#endif

extern "C" int pre_kth_axis0_strided_f32(const float *scores, int64_t row_stride, int64_t n, int64_t M, const int32_t *ks, int nk,
                                         float *out, void *stream)
{
    return pre_kth_axis0_planes_f32(scores, 0, row_stride, 1, n, M, ks, nk, out, M, 0, stream);
}

extern "C" int pre_kth_axis0_f32(const float *scores, int64_t n, int64_t M, const int32_t *ks, int nk, float *out, void *stream)
{
    return pre_kth_axis0_planes_f32(scores, 0, M, 1, n, M, ks, nk, out, M, 0, stream);
}
