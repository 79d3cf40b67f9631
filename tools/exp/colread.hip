// Experiment: achievable HBM read bandwidth for column-tile walks over a row-major [n, M] fp32
// tensor, as a function of the tile width W (bytes per row segment = 4*W).  Each workgroup
// (1024 threads) owns W adjacent columns and walks all n rows; lanes are arranged W-wide, so one
// wave-instruction touches 64/W rows.  Result: how narrow can a per-cell selection tile be before
// DRAM/TLB efficiency collapses?   hipcc -O3 --offload-arch=gfx950 colread.hip -o colread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int W>
__global__ void __launch_bounds__(1024) colsum(const float *__restrict__ s, int n, long long M, float *__restrict__ out)
{
    const int tid = threadIdx.x, cell = tid % W, rsub = tid / W;
    constexpr int RPI = 1024 / W;
    const long long c = (long long)blockIdx.x * W + cell;
    if (c >= M) return;
    const float *col = s + c;
    float acc = 0.f;
    int i = rsub;
    for (; i + 7 * RPI < n; i += 8 * RPI) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(long long)(i + u * RPI) * M];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n; i += RPI) acc += col[(long long)i * M];
    atomicAdd(out + c, acc);
}

template <int W>
float run(const float *d, int n, long long M, float *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)((M + W - 1) / W);
    colsum<W><<<grid, 1024>>>(d, n, M, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) colsum<W><<<grid, 1024>>>(d, n, M, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main()
{
    const int ns[3] = {4096, 1024, 256};
    const long long Ms[3] = {524288, 2097152, 2621440};
    for (int k = 0; k < 3; ++k) {
        const int n = ns[k]; const long long M = Ms[k];
        float *d, *out;
        hipMalloc(&d, sizeof(float) * n * M); hipMalloc(&out, sizeof(float) * M);
        hipMemset(d, 0, sizeof(float) * n * M); hipMemset(out, 0, sizeof(float) * M);
        const double gb = 4.0 * n * M / 1e9;
        printf("n=%d M=%lld (%.1f GB): ", n, M, gb);
        printf("W=8 %.0f GB/s | ", gb / run<8>(d, n, M, out) * 1e3);
        printf("W=16 %.0f GB/s | ", gb / run<16>(d, n, M, out) * 1e3);
        printf("W=32 %.0f GB/s | ", gb / run<32>(d, n, M, out) * 1e3);
        printf("W=64 %.0f GB/s | ", gb / run<64>(d, n, M, out) * 1e3);
        printf("W=128 %.0f GB/s | ", gb / run<128>(d, n, M, out) * 1e3);
        printf("W=256 %.0f GB/s\n", gb / run<256>(d, n, M, out) * 1e3);
        hipFree(d); hipFree(out);
    }
    return 0;
}
