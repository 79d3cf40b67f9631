/* cp_pre_fft.h - C ABI of libcp_pre_fft.so: the spectral (FFT) family of ConvOperator on MI355X.
 *
 * Replaces, for device-resident fp32 fields, the arithmetic of
 *   ConvOperator.spectral_convolution   Utils/ConvOps_2d.py:153-176, Utils/ConvOps_1d.py:153-175,
 *                                       Utils/ConvOps_Spatial.py:139-158  (-> fft_conv,
 *                                       Utils/fft_conv_pytorch/fft_conv.py:35-131)
 *   ConvOperator.differentiate          Utils/ConvOps_2d.py:179-228, Utils/ConvOps_1d.py:178-225
 *   ConvOperator.integrate              Utils/ConvOps_2d.py:231-284, Utils/ConvOps_1d.py:228-283
 * which are all "zero-pad, rfftn, multiply by a function of the kernel spectrum, irfftn, crop".
 * One call = embed (pad) kernel -> hipFFT R2C -> spectrum multiply kernel -> hipFFT C2R -> crop/scale
 * kernel.  The kernel spectrum is evaluated analytically from the taps inside the multiply kernel
 * (K^(w) = sum_taps w_p exp(-2 pi i w.p / n): what rfftn of the zero-padded kernel computes).
 *
 * A separate library from libcp_pre_hip.so because it links hipFFT.  Plans live in an opaque handle
 * the CALLER owns (create / destroy); hipFFT allocates its own work area inside the handle at creation.
 * Every call is asynchronous on the given HIP stream; no global state.
 * Return codes: 0 ok; < 0 as in cp_pre_hip.h (PRE_E_*); > 0: 1000 + hipfftResult, or a hipError_t.
 */
#ifndef CP_PRE_FFT_H
#define CP_PRE_FFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pre_fft pre_fft_t;

#define PRE_FFT_CONJ 1   /* multiply by conj(K^): cross-correlation (fft_conv.py:77; `correlation=True`) */
#define PRE_FFT_INVERT 2 /* multiply by 1 / (K^ + eps) (after the optional conj): `inverse=True`, integrate() */

int pre_fft_abi_version(void);

/* Plans for real transforms of `batch` arrays of n[0] x n[1] x n[2] (nd = 3) or n[1] x n[2] (nd = 2, n[0] = 1),
 * last axis contiguous.  inv_last = length of the last axis the inverse transform produces: n[2], or
 * n[2] - 1 when n[2] is odd and the reference calls irfftn without a size (differentiate / integrate:
 * Utils/ConvOps_2d.py:219,274 - torch then assumes an even length 2*(n[2]/2+1 - 1)). */
int pre_fft_create(pre_fft_t **handle, int nd, const int64_t n[3], int64_t inv_last, int64_t batch);
int pre_fft_destroy(pre_fft_t *handle);

/* Scratch (device bytes) the caller passes to pre_spectral_apply_f32 as `work`: the padded real array and its
 * half-spectrum. */
int pre_fft_work_bytes(const pre_fft_t *handle, size_t *bytes);

/* out[b, j0, j1, j2] = irfftn( rfftn(pad(in))[b] * g(K^) )[j0, j1, j2]   for j < out_dims,
 *   pad(in)[b, i0, i1, i2] = in[b, i0 - pad_lo[0], i1 - pad_lo[1], i2 - pad_lo[2]], zero outside dims;
 *   K^ = rfftn of the dense kernel (kdims[0] x kdims[1] x kdims[2], host floats, row-major) placed at the origin
 *        of a zero array of the transform size;
 *   g(K^) = K^ | conj(K^) | 1/(K^ + eps) | 1/(conj(K^) + eps)     per `mode`;
 *   irfftn normalised by 1 / (n[0] * n[1] * inv_last), as torch.fft.irfftn does.
 * in / out: device fp32 with element strides {batch, axis0, axis1, axis2}; dims / out_dims: extents of the
 * three axes (axis0 = 1 for nd = 2); out_dims[i] <= n[i] (inv_last for the last axis). */
int pre_spectral_apply_f32(pre_fft_t *handle, const float *in, const int64_t in_strides[4], const int64_t dims[3],
                           const int64_t pad_lo[3], const float *kernel, const int64_t kdims[3], int mode, float eps,
                           float *out, const int64_t out_strides[4], const int64_t out_dims[3], void *work,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
