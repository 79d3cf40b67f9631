// spectral.hip - libcp_pre_fft.so: ConvOperator's spectral family (spectral_convolution / differentiate /
// integrate) as   embed -> hipFFT R2C -> spectrum multiply -> hipFFT C2R -> crop   on one HIP stream.
// Contract and reference citations: include/cp_pre_fft.h.  gfx950 only.
//
// HBM passes per call (padded array of N reals, half-spectrum of ~N/2 complex): embed 4+4 B, R2C, multiply
// 8+8 B per bin, C2R, crop 4+4 B - the three hand-written kernels are single streaming passes; the kernel
// spectrum is never materialised (evaluated per frequency bin from the <= 7^3 dense weights, once per bin and
// reused across the batch).
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <new>
#include "../../include/cp_pre_fft.h"
#include "../../include/cp_pre_hip.h"

struct pre_fft {
    hipfftHandle r2c, c2r;
    int nd;
    int64_t n[3], inv_last, batch;
};

namespace {

#define FFT_CHECK(call)                                  \
    do {                                                 \
        hipfftResult r__ = (call);                       \
        if (r__ != HIPFFT_SUCCESS) return 1000 + (int)r__; \
    } while (0)
#define LAUNCH_CHECK()                                   \
    do {                                                 \
        hipError_t e__ = hipGetLastError();              \
        if (e__ != hipSuccess) return (int)e__;          \
    } while (0)

// R[b][i0][i1][i2] = in[b, i0-p0, i1-p1, i2-p2] or 0.   grid: (ceil(n2/256), n1, min(n0*batch, 65535))
__global__ void __launch_bounds__(256) embed_kernel(const float *__restrict__ in, long long sB, long long s0, long long s1,
                                                    long long s2, int d0, int d1, int d2, int p0, int p1, int p2,
                                                    float *__restrict__ R, int n0, int n1, int n2, long long planes)
{
    const int i2 = blockIdx.x * 256 + threadIdx.x, i1 = blockIdx.y;
    if (i2 >= n2) return;
    const int j1 = i1 - p1, j2 = i2 - p2;
    const bool in12 = j1 >= 0 && j1 < d1 && j2 >= 0 && j2 < d2;
    for (long long z = blockIdx.z; z < planes; z += gridDim.z) {
        const long long b = z / n0;
        const int i0 = (int)(z - b * n0), j0 = i0 - p0;
        float v = 0.f;
        if (in12 && j0 >= 0 && j0 < d0) v = in[b * sB + j0 * s0 + j1 * s1 + j2 * s2];
        R[(z * n1 + i1) * n2 + i2] = v;
    }
}

template <int K>
struct Dense { float w[K][K][K]; };

struct cplx { float re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// exp(-2 pi i * (i * p mod n) / n): the argument is reduced exactly in integers first
__device__ __forceinline__ cplx twiddle(int i, int p, int n)
{
    const int r = (int)(((long long)i * p) % n);
    float s, c;
    sincospif(-2.0f * ((float)r / (float)n), &s, &c);
    return {c, s};
}

// C[b][i0][i1][i2] *= g(K^(i0,i1,i2)).   grid: (ceil(m/256), n1, n0); the batch loop reuses g.
template <int K>
__global__ void __launch_bounds__(256) multiply_kernel(float2 *__restrict__ C, int n0, int n1, int n2, int m, long long batch,
                                                       int mode, float eps, const Dense<K> ker)
{
    const int i2 = blockIdx.x * 256 + threadIdx.x, i1 = blockIdx.y, i0 = blockIdx.z;
    if (i2 >= m) return;
    cplx e0[K], e1[K], e2[K];
#pragma unroll
    for (int p = 0; p < K; ++p) {
        e0[p] = twiddle(i0, p, n0);
        e1[p] = twiddle(i1, p, n1);
        e2[p] = twiddle(i2, p, n2);
    }
    cplx kh = {0.f, 0.f};
#pragma unroll
    for (int p0 = 0; p0 < K; ++p0) {
        cplx t1 = {0.f, 0.f};
#pragma unroll
        for (int p1 = 0; p1 < K; ++p1) {
            cplx t2 = {0.f, 0.f};
#pragma unroll
            for (int p2 = 0; p2 < K; ++p2) {
                const float w = ker.w[p0][p1][p2];
                t2.re += w * e2[p2].re;
                t2.im += w * e2[p2].im;
            }
            const cplx u = cmul(t2, e1[p1]);
            t1.re += u.re;
            t1.im += u.im;
        }
        const cplx u = cmul(t1, e0[p0]);
        kh.re += u.re;
        kh.im += u.im;
    }
    if (mode & PRE_FFT_CONJ) kh.im = -kh.im;
    if (mode & PRE_FFT_INVERT) {
        const float a = kh.re + eps, b = kh.im, d = a * a + b * b;
        kh = {a / d, -b / d};
    }
    const long long per = (long long)n0 * n1 * m;
    float2 *c = C + ((long long)i0 * n1 + i1) * m + i2;
    for (long long b = 0; b < batch; ++b, c += per) {
        const float2 x = *c;
        *c = make_float2(x.x * kh.re - x.y * kh.im, x.x * kh.im + x.y * kh.re);
    }
}

// out[b, j0, j1, j2] = scale * R[b][j0][j1][j2],  R rows of length r2.   grid: (ceil(o2/256), o1, min(o0*batch, 65535))
__global__ void __launch_bounds__(256) crop_kernel(const float *__restrict__ R, int n0, int n1, int r2, float scale,
                                                   float *__restrict__ out, long long sB, long long s0, long long s1,
                                                   long long s2, int o0, int o1, int o2, long long planes)
{
    const int j2 = blockIdx.x * 256 + threadIdx.x, j1 = blockIdx.y;
    if (j2 >= o2) return;
    for (long long z = blockIdx.z; z < planes; z += gridDim.z) {
        const long long b = z / o0;
        const int j0 = (int)(z - b * o0);
        out[b * sB + j0 * s0 + j1 * s1 + j2 * s2] = scale * R[((b * n0 + j0) * n1 + j1) * r2 + j2];
    }
}

template <int K>
int launch_multiply(float2 *C, const pre_fft *h, int m, int mode, float eps, const float *kernel, const int64_t kd[3],
                    hipStream_t st)
{
    Dense<K> ker;
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b)
            for (int c = 0; c < K; ++c)
                ker.w[a][b][c] = (a < kd[0] && b < kd[1] && c < kd[2]) ? kernel[(a * kd[1] + b) * kd[2] + c] : 0.f;
    const dim3 grid((unsigned)((m + 255) / 256), (unsigned)h->n[1], (unsigned)h->n[0]);
    hipLaunchKernelGGL(multiply_kernel<K>, grid, dim3(256), 0, st, C, (int)h->n[0], (int)h->n[1], (int)h->n[2], m,
                       (long long)h->batch, mode, eps, ker);
    LAUNCH_CHECK();
    return PRE_OK;
}

}  // namespace

extern "C" {

int pre_fft_abi_version(void) { return 1; }

int pre_fft_create(pre_fft_t **handle, int nd, const int64_t n[3], int64_t inv_last, int64_t batch)
{
    if (!handle || !n) return PRE_E_NULL;
    if ((nd != 2 && nd != 3) || batch <= 0 || batch > 0x7fffffff) return PRE_E_SHAPE;
    if (n[0] <= 0 || n[1] <= 0 || n[2] < 2 || (nd == 2 && n[0] != 1)) return PRE_E_SHAPE;
    if (n[0] > 65535 || n[1] > 65535 || n[2] > 0x7fffff) return PRE_E_SHAPE;          // grid limits / exact float reduction
    if (inv_last != n[2] && !(n[2] % 2 == 1 && inv_last == n[2] - 1)) return PRE_E_SHAPE;
    pre_fft *h = new (std::nothrow) pre_fft;
    if (!h) return PRE_E_NULL;
    h->nd = nd;
    for (int a = 0; a < 3; ++a) h->n[a] = n[a];
    h->inv_last = inv_last;
    h->batch = batch;
    int fwd[3] = {(int)n[0], (int)n[1], (int)n[2]}, inv[3] = {(int)n[0], (int)n[1], (int)inv_last};
    int *f = nd == 3 ? fwd : fwd + 1, *i = nd == 3 ? inv : inv + 1;
    hipfftResult r = hipfftPlanMany(&h->r2c, nd, f, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, (int)batch);
    if (r != HIPFFT_SUCCESS) { delete h; return 1000 + (int)r; }
    r = hipfftPlanMany(&h->c2r, nd, i, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2R, (int)batch);
    if (r != HIPFFT_SUCCESS) { hipfftDestroy(h->r2c); delete h; return 1000 + (int)r; }
    *handle = h;
    return PRE_OK;
}

int pre_fft_destroy(pre_fft_t *h)
{
    if (!h) return PRE_E_NULL;
    hipfftDestroy(h->r2c);
    hipfftDestroy(h->c2r);
    delete h;
    return PRE_OK;
}

int pre_fft_work_bytes(const pre_fft_t *h, size_t *bytes)
{
    if (!h || !bytes) return PRE_E_NULL;
    const size_t real = (size_t)h->batch * h->n[0] * h->n[1] * h->n[2] * sizeof(float);
    const size_t spec = (size_t)h->batch * h->n[0] * h->n[1] * (h->n[2] / 2 + 1) * 2 * sizeof(float);
    *bytes = ((real + 255) & ~(size_t)255) + spec;
    return PRE_OK;
}

int pre_spectral_apply_f32(pre_fft_t *h, const float *in, const int64_t is[4], const int64_t dims[3], const int64_t pad_lo[3],
                           const float *kernel, const int64_t kd[3], int mode, float eps, float *out, const int64_t os[4],
                           const int64_t od[3], void *work, void *stream)
{
    if (!h || !in || !is || !dims || !pad_lo || !kernel || !kd || !out || !os || !od || !work) return PRE_E_NULL;
    if (mode & ~(PRE_FFT_CONJ | PRE_FFT_INVERT)) return PRE_E_UNSUPPORTED;
    for (int a = 0; a < 3; ++a) {
        if (dims[a] <= 0 || pad_lo[a] < 0 || dims[a] + pad_lo[a] > h->n[a]) return PRE_E_SHAPE;
        if (kd[a] <= 0 || kd[a] > 7 || kd[a] > h->n[a]) return PRE_E_SHAPE;
        if (od[a] <= 0 || od[a] > (a == 2 ? h->inv_last : h->n[a])) return PRE_E_SHAPE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n0 = (int)h->n[0], n1 = (int)h->n[1], n2 = (int)h->n[2], m = n2 / 2 + 1;
    float *R = (float *)work;
    const size_t real = (size_t)h->batch * n0 * n1 * n2 * sizeof(float);
    float2 *C = (float2 *)((char *)work + ((real + 255) & ~(size_t)255));

    long long planes = (long long)h->batch * n0;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)n1, (unsigned)(planes < 65535 ? planes : 65535)),
                       dim3(256), 0, st, in, (long long)is[0], (long long)is[1], (long long)is[2], (long long)is[3],
                       (int)dims[0], (int)dims[1], (int)dims[2], (int)pad_lo[0], (int)pad_lo[1], (int)pad_lo[2], R, n0, n1, n2,
                       planes);
    LAUNCH_CHECK();
    FFT_CHECK(hipfftSetStream(h->r2c, st));
    FFT_CHECK(hipfftExecR2C(h->r2c, R, (hipfftComplex *)C));
    const int kmax = (int)(kd[0] > kd[1] ? (kd[0] > kd[2] ? kd[0] : kd[2]) : (kd[1] > kd[2] ? kd[1] : kd[2]));
    int rc = kmax <= 3 ? launch_multiply<3>(C, h, m, mode, eps, kernel, kd, st)
           : kmax <= 5 ? launch_multiply<5>(C, h, m, mode, eps, kernel, kd, st)
                       : launch_multiply<7>(C, h, m, mode, eps, kernel, kd, st);
    if (rc != PRE_OK) return rc;
    FFT_CHECK(hipfftSetStream(h->c2r, st));
    FFT_CHECK(hipfftExecC2R(h->c2r, (hipfftComplex *)C, R));
    const int r2 = (int)h->inv_last;
    const float scale = (float)(1.0 / ((double)n0 * n1 * r2));
    planes = (long long)h->batch * od[0];
    hipLaunchKernelGGL(crop_kernel, dim3((unsigned)((od[2] + 255) / 256), (unsigned)od[1], (unsigned)(planes < 65535 ? planes : 65535)),
                       dim3(256), 0, st, R, n0, n1, r2, scale, out, (long long)os[0], (long long)os[1], (long long)os[2],
                       (long long)os[3], (int)od[0], (int)od[1], (int)od[2], planes);
    LAUNCH_CHECK();
    return PRE_OK;
}

}  // extern "C"
