// does buffer_load_dwordx4 honour a 4-byte aligned (not 16-byte aligned) address on gfx950?  (tools/exp probe)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *p, float *o, int shift_base, int shift_off)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + shift_base), 0, -1, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (threadIdx.x * 4 + shift_off) * 4, 0, 0);
    o[threadIdx.x * 4 + 0] = __uint_as_float(v.x); o[threadIdx.x * 4 + 1] = __uint_as_float(v.y);
    o[threadIdx.x * 4 + 2] = __uint_as_float(v.z); o[threadIdx.x * 4 + 3] = __uint_as_float(v.w);
}
int main()
{
    float h[1024], *d, *o, g[256];
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof g);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int sb = 0; sb < 4; ++sb)
        for (int so = 0; so < 4; ++so) {
            k<<<1, 64>>>(d, o, sb, so);
            hipMemcpy(g, o, sizeof g, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < 256; ++i) bad += g[i] != (float)(i + sb + so);
            printf("base+%d off+%d: %s (first %g %g %g %g)\n", sb, so, bad ? "WRONG" : "ok", g[0], g[1], g[2], g[3]);
        }
    return 0;
}
