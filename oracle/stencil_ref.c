/* Oracle (TEST INFRASTRUCTURE): plain-C restatement of the reference's arithmetic
 * call, the single-channel zero-padded cross-correlation
 *   F.conv3d(field[:,None], K[None,None], padding=(k0//2,k1//2,k2//2))   Utils/ConvOps_2d.py:149
 *   F.conv2d(field[:,None], K[None,None], padding=(k0//2,k1//2))          Utils/ConvOps_1d.py:150
 * out[b,i0,i1,i2] = sum_{a0,a1,a2} K[a0,a1,a2] * in[b, i0+a0-k0/2, i1+a1-k1/2, i2+a2-k2/2]
 * with out-of-range input = 0.  float32 accumulate, taps in row-major order, zero
 * weights skipped.  Independent of torch; checked against the golden vectors in
 * tests/test_oracle_golden.py.  Never linked into the product library.
 */
#include <stdint.h>
#include <stddef.h>

int oracle_xcorr3d_f32(const float *in, float *out, const float *K,
                       int k0, int k1, int k2,
                       int64_t B, int64_t n0, int64_t n1, int64_t n2)
{
    if (!in || !out || !K || k0 < 1 || k1 < 1 || k2 < 1 || !(k0 & k1 & k2 & 1)) return -1;
    const int p0 = k0 / 2, p1 = k1 / 2, p2 = k2 / 2;
    const int64_t vol = n0 * n1 * n2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t i0 = 0; i0 < n0; ++i0)
            for (int64_t i1 = 0; i1 < n1; ++i1) {
                float *o = out + b * vol + (i0 * n1 + i1) * n2;
                for (int64_t i2 = 0; i2 < n2; ++i2) o[i2] = 0.0f;
                for (int a0 = 0; a0 < k0; ++a0) {
                    const int64_t j0 = i0 + a0 - p0;
                    if (j0 < 0 || j0 >= n0) continue;
                    for (int a1 = 0; a1 < k1; ++a1) {
                        const int64_t j1 = i1 + a1 - p1;
                        if (j1 < 0 || j1 >= n1) continue;
                        const float *row = in + b * vol + (j0 * n1 + j1) * n2;
                        for (int a2 = 0; a2 < k2; ++a2) {
                            const float w = K[(a0 * k1 + a1) * k2 + a2];
                            if (w == 0.0f) continue;
                            const int64_t s = a2 - p2;
                            const int64_t lo = s < 0 ? -s : 0;
                            const int64_t hi = s > 0 ? n2 - s : n2;
                            for (int64_t i2 = lo; i2 < hi; ++i2) o[i2] += w * row[i2 + s];
                        }
                    }
                }
            }
    return 0;
}

int oracle_xcorr2d_f32(const float *in, float *out, const float *K,
                       int k0, int k1, int64_t B, int64_t n0, int64_t n1)
{
    /* [B,n0,n1] with a k0*k1 kernel == [B,1,n0,n1] with a 1*k0*k1 kernel */
    return oracle_xcorr3d_f32(in, out, K, 1, k0, k1, B, 1, n0, n1);
}
