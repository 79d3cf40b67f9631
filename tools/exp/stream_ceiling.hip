// stream_ceiling.hip - what can F concurrent input streams + 1 output stream reach on the MI355X with the SHAPE of
// march_kernel and none of its work?  (the six-field MHD functors run at 0.61-0.68 of peak, 13 % under the
// float4-copy ceiling that was measured for 1-3 streams; is there a lower ceiling for seven streams at one 512-thread
// workgroup per CU?)
//
// One workgroup = 8 rows x 64 float4 lanes = an (x, y) tile of 8 x 256 cells of one sample of [B, F, T, X, Y] fields; it
// marches over t, each thread loading its own float4 of every field of plane t + D (D = prefetch depth in planes) through a
// wave-uniform buffer descriptor before it sums and stores plane t.  LDS bytes are declared (and touched) only to pin the
// number of workgroups a CU holds: 100 KB -> one, 50 KB -> two, 25 KB -> four.
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/stream_ceiling tools/exp/stream_ceiling.hip
//   ./tools/exp/stream_ceiling            (prints one line per variant)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (cp_pre_amd/csrc/common.h: each XCD walks a contiguous run of tiles)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks)
{
    const unsigned nx = 8u;
    const unsigned per = nblocks / nx, rem = nblocks % nx;
    const unsigned xcd = bid % nx, idx = bid / nx;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

__global__ void fill_kernel(float *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = 0.5f + (float)(h >> 8) * (1.0f / 16777216.0f);
    }
}

// TS: planes a workgroup marches (0 = the whole T axis); the t segments of a sample are dealt like the product's: for each sample,
// for each segment, all its tiles
template <int F, int D, int LDSB, int TS = 0>
__global__ void __launch_bounds__(512) stream_kernel(const float *in, float *out, int T, int X, int Y, long long sF, long long sB,
                                                     long long plane, long long oB)
{
    __shared__ float pad[LDSB / 4];
    const int q = threadIdx.x, ty = threadIdx.y;
    const int nXT = X / 8, nYT = Y / 256;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int yt = L % nYT; L /= nYT;
    const int xt = L % nXT; L /= nXT;
    const int nseg = TS ? (T + TS - 1) / TS : 1;
    const int ts = L % nseg;
    const int b = L / nseg;
    const int tb = TS ? ts * TS : 0, te = TS ? (tb + TS < T ? tb + TS : T) : T;
    const int x = xt * 8 + ty, y = (yt * 64 + q) * 4;
    pad[ty * 64 + q] = (float)x;                                  // (keeps the allocation alive)
    const unsigned voff = (unsigned)((x * Y + y) * 4);
    auto rsrc = [&](int i, int t) __attribute__((always_inline)) {
        const float *p = in + (long long)b * sB + (long long)i * sF + (long long)t * plane;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, -1, 0x00020000);
    };
    float4 w[D + 1][F];
    auto load = [&](int t, float4(&dst)[F]) __attribute__((always_inline)) {
        if (t < te) {
#pragma unroll
            for (int i = 0; i < F; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc(i, t), (int)voff, 0, 0);
                dst[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) load(tb + s, w[s]);
    float *op = out + (long long)b * oB + (long long)x * Y + y;
    for (int t0 = tb; t0 < te; t0 += D + 1) {
#pragma unroll
        for (int s = 0; s <= D; ++s) {
            const int t = t0 + s;
            if (t >= te) break;
            load(t + D, w[(s + D) % (D + 1)]);
            float4 r = w[s][0];
#pragma unroll
            for (int i = 1; i < F; ++i) { r.x += w[s][i].x; r.y += w[s][i].y; r.z += w[s][i].z; r.w += w[s][i].w; }
            *reinterpret_cast<float4 *>(op + (long long)t * plane) = r;
        }
    }
    if (pad[(ty * 64 + q + 1) & 511] == -12345.f) out[0] = 0.f;
}

// padP / padF / padB: floats added to the plane / field / sample stride of the INPUT (0 = the dense [B,F,T,X,Y] tensor the
// reference's `vars` is); padO: floats added to the sample stride of the output
template <int F, int D, int LDSB, int TS = 0>
void run(const float *in, float *out, int B, int T, int X, int Y, int Ftot, long long padP = 0, long long padF = 0, long long padB = 0,
         long long padO = 0)
{
    const long long plane = (long long)X * Y + padP, sF = (long long)T * plane + padF, sB = (long long)Ftot * sF + padB;
    const long long oB = (long long)T * plane + padO;
    // the padded views must lie inside the two allocations (checked on the host: an out-of-bounds stream faults the GPU)
    extern size_t g_nin, g_nout;
    const long long need_in = (long long)(B - 1) * sB + (long long)(Ftot - 1) * sF + (long long)(T - 1) * plane + (long long)X * Y;
    const long long need_out = (long long)(B - 1) * oB + (long long)(T - 1) * plane + (long long)X * Y;
    if (need_in > (long long)g_nin || need_out > (long long)g_nout) {
        printf("F = %d, pads plane %lld field %lld sample %lld out %lld: does not fit the buffers, skipped\n", F, padP, padF, padB, padO);
        return;
    }
    const unsigned grid = (unsigned)((long long)B * (X / 8) * (Y / 256) * (TS ? (T + TS - 1) / TS : 1));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stream_kernel<F, D, LDSB, TS>, 512, 0));
    float best = 1e30f, sum = 0.f;
    const int reps = 5;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<F, D, LDSB, TS>), dim3(grid), dim3(64, 8), 0, 0, in, out, T, X, Y, sF, sB, plane, oB);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = 4.0 * (F + 1) * B * T * (double)X * Y;
    printf("F = %d + 1 streams, prefetch %d, %3d KB LDS -> %d wg/CU, %d planes per workgroup, pads plane %lld field %lld sample %lld out %lld floats: "
           "%7.3f ms mean (%7.3f best)  %6.0f GB/s = %.3f of 8 TB/s\n",
           F, D, LDSB / 1024, per_cu, TS ? TS : T, padP, padF, padB, padO, sum / reps, best, bytes / (sum / reps) / 1e6, bytes / (sum / reps) / 1e6 / 8000.0);
    fflush(stdout);
}

size_t g_nin = 0, g_nout = 0;

int main(int argc, char **argv)
{
    const int B = 1000, Ftot = 6, T = 64, X = 256, Y = 256;             // (1000 samples: room for the padded strides in the same buffers)
    const size_t nin = (size_t)1024 * Ftot * T * X * Y + (64u << 20), nout = (size_t)1024 * T * X * Y + (16u << 20);
    float *in, *out;
    g_nin = nin; g_nout = nout;
    CK(hipMalloc(&in, nin * 4));
    CK(hipMalloc(&out, nout * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, in, nin);          // (not zeros: U(0.5, 1.5) like the benchmarks' fields)
    CK(hipMemset(out, 0, nout * 4));
    CK(hipDeviceSynchronize());
    printf("fields [%d,%d,%d,%d,%d] fp32, output [%d,%d,%d,%d]; tile 8 x 256 cells per 512-thread workgroup, T marched\n",
           B, Ftot, T, X, Y, B, T, X, Y);
    if (argc > 1 && argv[1][0] == 't') {      // the segment experiment: shorter marches = more, shorter workgroups in dispatch order
        for (int rep = 0; rep < 2; ++rep) {
            run<6, 2, 100 * 1024, 0>(in, out, B, T, X, Y, Ftot);
            run<6, 2, 100 * 1024, 32>(in, out, B, T, X, Y, Ftot);
            run<6, 2, 100 * 1024, 16>(in, out, B, T, X, Y, Ftot);
            run<6, 2, 100 * 1024, 8>(in, out, B, T, X, Y, Ftot);
            run<6, 2, 100 * 1024, 4>(in, out, B, T, X, Y, Ftot);
            run<3, 2, 50 * 1024, 0>(in, out, B, T, X, Y, Ftot);
            run<3, 2, 50 * 1024, 16>(in, out, B, T, X, Y, Ftot);
            run<3, 2, 50 * 1024, 8>(in, out, B, T, X, Y, Ftot);
            run<3, 2, 50 * 1024, 4>(in, out, B, T, X, Y, Ftot);
            run<1, 2, 50 * 1024, 0>(in, out, B, T, X, Y, Ftot);
            run<1, 2, 50 * 1024, 8>(in, out, B, T, X, Y, Ftot);
            run<1, 2, 50 * 1024, 2>(in, out, B, T, X, Y, Ftot);
        }
        return 0;
    }
    if (argc > 1) {            // the stride experiment: do power-of-two plane / field / sample strides cost these streams anything?
        for (int rep = 0; rep < 2; ++rep) {
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 0, 0, 0, 1088);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 0, 1088);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 0, 1088, 0, 4160);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 0, 16448, 0, 4160);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 64);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 1024);
            run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot, 1024, 1088, 0, 4160);
            run<3, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot);
            run<3, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot, 0, 1088, 0, 4160);
            run<3, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot, 1024, 1088, 0, 4160);
            run<1, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot);
            run<1, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot, 0, 0, 0, 4160);
            run<1, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot, 1024, 0, 0, 4160);
        }
        return 0;
    }
    run<6, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot);
    run<6, 3, 100 * 1024>(in, out, B, T, X, Y, Ftot);
    run<6, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot);
    run<6, 3, 50 * 1024>(in, out, B, T, X, Y, Ftot);
    run<6, 2, 25 * 1024>(in, out, B, T, X, Y, Ftot);
    run<4, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot);
    run<4, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot);
    run<4, 3, 50 * 1024>(in, out, B, T, X, Y, Ftot);
    run<4, 2, 25 * 1024>(in, out, B, T, X, Y, Ftot);
    run<3, 2, 100 * 1024>(in, out, B, T, X, Y, Ftot);
    run<3, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot);
    run<3, 2, 25 * 1024>(in, out, B, T, X, Y, Ftot);
    run<1, 2, 50 * 1024>(in, out, B, T, X, Y, Ftot);
    run<1, 2, 25 * 1024>(in, out, B, T, X, Y, Ftot);
    return 0;
}
