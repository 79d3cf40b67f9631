// stencil_generic.hip - C-ABI entry points for ConvOperator.convolution and the generic
// (any stride, any odd kernel extent up to 7) tap-list kernel.
//
// pre_stencil3d_f32 first offers the tap list to the streaming star kernel
// (star_march.hip); tap sets that are not on the 7-point star, views without a unit-stride
// axis, and the last (extent % 4) columns of an odd-width grid run here:
// one thread per output cell, coalesced along y, neighbours served by L1/L2.  It is the
// correctness floor of the library (5^3 / 7^3 Taylor kernels, permuted views, odd sizes);
// it is not on the benchmarked path.
#include "common.h"

int pre_star_try_linear1(const pre_field_t *in, const pre_out_t *out, const float star7[7],
                         int64_t B, int64_t T, int64_t X, int64_t Y, int flags, hipStream_t st,
                         int *tail_axis, int64_t *tail_from);

namespace {

constexpr int MAX_TAPS = 343;   // 7*7*7

struct TapList {
    int n;
    float w[MAX_TAPS];
    int off[MAX_TAPS];          // (dt+8) | (dx+8)<<4 | (dy+8)<<8
};

__global__ void __launch_bounds__(256) generic_kernel(const float *__restrict__ in, long long sB, long long sT,
                                                      long long sX, long long sY, float *__restrict__ out,
                                                      long long oB, long long oT, long long oX, long long oY,
                                                      int B, int T, int X, int Y, int t0, int x0, int y0,
                                                      int flags, const TapList taps)
{
    // outputs: the sub-box [t0,T) x [x0,X) x [y0,Y) (all of it for t0=x0=y0=0); taps see the whole domain
    const int nT = T - t0, nX = X - x0, nY = Y - y0;
    const long long plane = (long long)nX * nY;
    const long long total = (long long)B * nT * plane;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int y = y0 + (int)(idx % nY);
        const int x = x0 + (int)((idx / nY) % nX);
        const int t = t0 + (int)((idx / plane) % nT);
        const int b = (int)(idx / (plane * nT));
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

}  // namespace

extern "C" {

int pre_abi_version(void) { return 2; }

int pre_stencil3d_f32(const pre_field_t *in, const pre_out_t *out, const float *tap_w, const int32_t *tap_off, int ntaps,
                      int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream)
{
    if (!in || !in->ptr || !out || !out->ptr || (ntaps > 0 && (!tap_w || !tap_off))) return PRE_E_NULL;
    if (B <= 0 || T <= 0 || X <= 0 || Y <= 0 || ntaps < 0) return PRE_E_NULL;
    if (ntaps > MAX_TAPS) return PRE_E_SHAPE;
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

    TapList taps;
    taps.n = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        taps.w[i] = tap_w[i];
        taps.off[i] = (tap_off[3 * i] + 8) | ((tap_off[3 * i + 1] + 8) << 4) | ((tap_off[3 * i + 2] + 8) << 8);
    }
    const long long total = (long long)B * (T - box[0]) * (X - box[1]) * (Y - box[2]);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in->ptr, (long long)in->sB,
                       (long long)in->sT, (long long)in->sX, (long long)in->sY, out->ptr, (long long)out->sB,
                       (long long)out->sT, (long long)out->sX, (long long)out->sY, (int)B, (int)T, (int)X, (int)Y, box[0], box[1],
                       box[2], flags, taps);
    PRE_LAUNCH_CHECK();
    return PRE_OK;
}

int pre_stencil2d_f32(const float *in, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                      const float *tap_w, const int32_t *tap_off, int ntaps, int64_t B, int64_t T, int64_t X, int flags,
                      void *stream)
{
    if (!in || !in_strides || !out || !out_strides || (ntaps > 0 && !tap_off)) return PRE_E_NULL;
    if (ntaps < 0 || ntaps > MAX_TAPS) return PRE_E_SHAPE;
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
