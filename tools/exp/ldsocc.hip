// Experiment: how many 1024-thread workgroups with S bytes of static LDS does a gfx950 CU hold?
// Each block spins ~50 us; grid = 512 blocks on 256 CUs: one round (~50 us) means 2 blocks/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS>
__global__ void __launch_bounds__(1024, 8) spin(unsigned *out, long long ticks)
{
    __shared__ unsigned h[WORDS];
    h[threadIdx.x % WORDS] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) out[blockIdx.x] = h[(blockIdx.x * 7) % WORDS];
}
template <int WORDS> void run(unsigned *out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    spin<WORDS><<<512, 1024>>>(out, 5000);   // 100 MHz clock: 5000 ticks = 50 us
    hipDeviceSynchronize();
    hipEventRecord(a);
    spin<WORDS><<<512, 1024>>>(out, 5000);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("LDS %6d B: 512 blocks of 1024 threads took %.1f us  (%s)\n", WORDS * 4, ms * 1e3, ms < 0.08 ? "2 blocks/CU" : "1 block/CU");
}
int main()
{
    unsigned *out; hipMalloc(&out, 4096);
    run<16384>(out); run<18432>(out); run<19456>(out); run<20224>(out); run<20352>(out); run<20480>(out); run<20992>(out);
    return 0;
}
