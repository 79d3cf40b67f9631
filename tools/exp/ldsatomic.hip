// Experiment: cost of LDS atomics (no return) per wave-instruction on gfx950 under different address
// patterns.  One 1024-thread workgroup per CU-slot, each thread issues ITERS atomics; reports clocks
// per wave-instruction per CU (2 workgroups/CU resident when LDS allows).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int WORDS = 20480, ITERS = 4096;
// mode 0: lane-linear conflict-free (word = lane + 64*k)        mode 1: pairs share a word (16-bit halves)
// mode 2: random row per lane, bank = (half+row)&31 (the select kernel's map)   mode 3: all lanes same word
// mode 4: like 2 but 32-bit counters, one word per lane: word = row*64 + lane
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned seed)
{
    __shared__ unsigned h[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 1024) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned r = seed + threadIdx.x * 2654435761u;
    for (int i = 0; i < ITERS; ++i) {
        r = r * 1664525u + 1013904223u;
        const int row = (r >> 20) % 256;
        int w; unsigned inc = 1;
        if (MODE == 0) w = lane + 64 * (i & 255);
        if (MODE == 1) { w = (lane >> 1) + 32 * (i & 511); inc = 1u << (16 * (lane & 1)); }
        if (MODE == 2) { w = row * 32 + (((lane >> 1) + row) & 31); inc = 1u << (16 * (lane & 1)); }
        if (MODE == 3) w = i & 1023;
        if (MODE == 4) w = row * 64 + lane;
        atomicAdd(&h[w], inc);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = h[blockIdx.x % WORDS];
}
template <int MODE> void run(unsigned *out, const char *what)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<512, 1024>>>(out, 1); hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<512, 1024>>>(out, 2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 2 blocks x 16 waves x ITERS wave-instructions
    const double clk = ms * 1e-3 * 2.4e9 / (2.0 * 16 * ITERS);
    printf("%-58s %.3f ms  %.1f clk per wave-atomic per CU\n", what, ms, clk);
}
int main()
{
    unsigned *out; hipMalloc(&out, 4096);
    run<0>(out, "conflict-free, one word per lane");
    run<1>(out, "two lanes share a word (16-bit halves)");
    run<2>(out, "select map: random row per lane, shared words");
    run<3>(out, "all lanes one word");
    run<4>(out, "random row per lane, one word per lane");
    return 0;
}
