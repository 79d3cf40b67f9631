// which way do the DPP wave shifts move data on gfx950?  hipcc --offload-arch=gfx950 -o tools/exp/dpp_probe tools/exp/dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
    const int x = 100 + (int)threadIdx.x;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);            // wave_shr:1
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);       // wave_shl:1
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1  lane0=%d lane1=%d lane15=%d lane16=%d lane31=%d lane32=%d lane63=%d\n", h[0], h[1], h[15], h[16], h[31], h[32], h[63]);
    printf("wave_shl:1  lane0=%d lane1=%d lane15=%d lane16=%d lane31=%d lane32=%d lane62=%d lane63=%d\n", h[64], h[65], h[79], h[80], h[95], h[96], h[126], h[127]);
    int ok = 1;
    for (int i = 1; i < 64; ++i) ok &= h[i] == 100 + i - 1;
    for (int i = 0; i < 63; ++i) ok &= h[64 + i] == 100 + i + 1;
    printf("wave_shr:1 == value of lane-1, wave_shl:1 == value of lane+1 on all 64 lanes: %s\n", ok ? "YES" : "NO");
    return 0;
}
