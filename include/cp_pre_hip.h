/* cp_pre_hip.h - C ABI of libcp_pre_hip.so, the MI355X (gfx950) implementation of the
 * CP-PRE physics-residual hot path.
 *
 * The reference (gitvicky/CP-PRE) is 100 % Python and has no FFI; the entry points
 * below are what a binding for this path replaces.  Each one cites the reference
 * interface it stands in for (paths relative to the reference repository).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - all tensors are IEEE fp32; sizes are in elements, strides in elements;
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises, nothing allocates;
 *   - return value: 0 = ok, <0 = argument error (PRE_E_*), >0 = hipError_t;
 *   - no mutable global state (two device properties are cached on first use, read-only afterwards: the CU count and each
 *     marching kernel's resident workgroups per CU, which size its t segments), re-entrant, host-thread-safe.
 */
#ifndef CP_PRE_HIP_H
#define CP_PRE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRE_OK             0
#define PRE_E_NULL        -1   /* null pointer / bad size                                  */
#define PRE_E_SHAPE       -2   /* unsupported shape (even kernel extent, > 343 taps, ...)    */
#define PRE_E_UNSUPPORTED -3   /* operator kernels not star-shaped: caller composes unfused */
#define PRE_E_RANGE       -4   /* k / crop out of range                                     */

#define PRE_FLAG_ABS       1   /* store |residual| (marginal score, Marginal/Wave_Residuals_CP.py:280) */
#define PRE_FLAG_INTERIOR_T 2  /* fused residuals / star stencils: planes t=0 and t=T-1 of `out` MAY be left
                                  unwritten - for callers that crop the t rim anyway
                                  (res[...,1:-1,1:-1,1:-1], Marginal/NS_Residuals_CP.py:240) */
#define PRE_FLAG_OUT_INTERIOR_T 4 /* fused residuals (ns_momentum, linear2, mhd): `out` is a [B,T-2,X,Y] view that holds
                                  ONLY the interior planes - plane t of the residual (1 <= t <= T-2) is stored at
                                  out[b, t-1]; the rim planes are neither computed nor stored (implies
                                  PRE_FLAG_INTERIOR_T).  For streaming drivers that feed t-slabs with their two halo
                                  planes and keep one residual buffer of the slab's interior.  PRE_E_UNSUPPORTED when
                                  the views are not Y-contiguous (relabelled layouts) or T < 3. */

#define PRE_FLAG_HALO_X 8      /* fused residuals / star stencils: rows x = -1 and x = X of every field view EXIST in memory
                                  (ptr - sX and ptr + X*sX are readable) and are the x-neighbours of rows 0 and X-1
                                  instead of the zero padding; `out` holds the X rows.  For streaming drivers that cut
                                  the grid into x-slabs (T whole, one halo ROW per side: 2/X re-read instead of the 2/T
                                  halo planes of a t-slab; the reference's conv3d sees the whole grid,
                                  Utils/ConvOps_2d.py:149).  PRE_E_UNSUPPORTED when the views are not Y-contiguous, the
                                  taps are not star-shaped, Y % 4 != 0, or on the 1-D / boundary-condition entries. */

/* A strided view of one field [B,T,X,Y] (what `vars[:, i]` or a permuted surrogate
 * output is, Marginal/NS_Residuals_CP.py:282; Other_UQ/Evaluation/PRE_estimations.py:41). */
typedef struct {
    const float *ptr;
    int64_t sB, sT, sX, sY;
} pre_field_t;

/* A strided OUTPUT view [B,T,X,Y].  The library writes exactly the B*T*X*Y addressed elements.
 * The streaming kernels need the input views and the output view to share one unit-stride axis
 * (any of T, X, Y - the axes are relabelled internally, so the surrogate's native
 * [BS,F,Nx,Ny,Nt] layout seen through permute(0,1,4,2,3) is consumed without a copy);
 * otherwise the entry returns PRE_E_UNSUPPORTED (fused residuals) or runs the generic kernel
 * (pre_stencil*). */
typedef struct {
    float *ptr;
    int64_t sB, sT, sX, sY;
} pre_out_t;

#define PRE_ABI_VERSION 8
int pre_abi_version(void);     /* == PRE_ABI_VERSION (v8: pre_kth_axis0_planes_f32, pre_joint_score_pruned_max_segments);
                                  a binding checks it at load time: signatures may change between versions */

/* ---- a4/a5/a6: ConvOperator.convolution ---------------------------------------------
 * Utils/ConvOps_2d.py:135-150  F.conv3d(field[:,None], K[None,None], padding=k//2)
 * Utils/ConvOps_1d.py:130-150  F.conv2d(...)
 * Zero-padded single-channel cross-correlation given as a tap list (host arrays):
 *   out[b,t,x,y] = sum_i w[i] * in[b, t+off[3i], x+off[3i+1], y+off[3i+2]]   (0 outside)
 * `in` and `out` may be any strided views.  |off| <= 3, ntaps <= 343.
 * Star-shaped 3x3x3 tap sets on a y-contiguous 16-byte aligned view take the streaming
 * kernel; everything else takes the generic kernel.  Same result either way. */
int pre_stencil3d_f32(const pre_field_t *in, const pre_out_t *out,
                      const float *tap_w /*host*/, const int32_t *tap_off /*host, 3*ntaps*/, int ntaps,
                      int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream);

/* [B,T,X] fields with a 2-D kernel; tap_off is 2*ntaps (dt,dx); strides = {sB,sT,sX}. */
int pre_stencil2d_f32(const float *in, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                      const float *tap_w /*host*/, const int32_t *tap_off /*host*/, int ntaps,
                      int64_t B, int64_t T, int64_t X, int flags, void *stream);

/* Gradient of pre_stencil3d_f32 with respect to a dense (kt,kx,ky) kernel, extents in {1,3} (autograd: the
 * reference puts ConvOperator in a physics-informed loss and sets D.kernel.requires_grad = True,
 * Physics_Informed/Wave_FNO_PI.py:202-228):  gk[dt][dx][dy] += sum_cells g[c] * x[c + (dt,dx,dy)], zero padding.
 * x, g: device views with unit stride along Y; gk: device doubles [kt*kx*ky], zeroed by the caller.
 * PRE_E_UNSUPPORTED for other extents / layouts (the caller composes the products). */
int pre_stencil3d_wgrad_f32(const pre_field_t *x, const pre_field_t *g, int kt, int kx, int ky,
                            int64_t B, int64_t T, int64_t X, int64_t Y, double *gk /*device*/, void *stream);

/* ---- a9: fused PDE residuals --------------------------------------------------------
 * Operators are passed as the DENSE 3x3x3 kernels the caller's ConvOperator objects hold
 * (27 floats each, host, axes (Nt,Nx,Ny)), so every construction quirk of the reference
 * (D_y == D_t, Utils/ConvOps_2d.py:72-73) is inherited.  Returns PRE_E_UNSUPPORTED if a
 * kernel has a tap off the 7-point star; the caller then composes pre_stencil3d_f32 calls.
 * `out` is the UNCROPPED residual view (the caller crops [...,1:-1,1:-1,1:-1]). */

/* Marginal/NS_Residuals_CP.py:231-240; Other_UQ/Evaluation/PRE_estimations.py:40-50 */
int pre_residual_ns_momentum_f32(const pre_field_t *u, const pre_field_t *v, const pre_field_t *p, const pre_out_t *out,
                                 const float *K_t, const float *K_x, const float *K_y, const float *K_xx_yy,
                                 float dt, float dx, float dy, float nu,
                                 int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream);
/* Marginal/NS_Residuals_CP.py:222-228   res = D_x(u) + ratio*D_y(v)   (also Divergence/Curl/gauss:
 * out = Ka(f0) + ratio*Kb(f1), Utils/VectorConvOps.py:38,65; Marginal/MHD_Residuals_CP.py:272-278) */
int pre_residual_linear2_f32(const pre_field_t *f0, const pre_field_t *f1, const pre_out_t *out,
                             const float *K_a, const float *K_b, float ratio,
                             int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream);
/* Joint/Burgers_Residuals_CP.py:171-187   [B,T,X]; 3x3 kernels (9 floats, host), axes (Nt,Nx);
 * res = dx*D_t(u) + dt*u*D_x(u) - nu*D_xx(u)*c3   with c3 = 2*dt/dx evaluated by the caller in fp32 */
int pre_residual_burgers_f32(const float *u, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                             const float *K_t, const float *K_x, const float *K_xx,
                             float dx, float dt, float nu, float c3,
                             int64_t B, int64_t T, int64_t X, int flags, void *stream);
/* Marginal/MHD_Residuals_CP.py:225-278; eq: 0 continuity, 1 momentum, 2 energy, 3 induction.
 * fields = {rho,u,v,p,Bx,By} (all six views must be valid even if the equation skips some). */
int pre_residual_mhd_f32(int eq, const pre_field_t fields[6], const pre_out_t *out,
                         const float *K_t, const float *K_x, const float *K_y, double gamma,
                         int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream);

/* Reduced MHD (JOREK), Marginal/JOREK_residuals_CP.py:207-243 (twin Joint/JOREK_residuals_CP.py:210-243); ABI v6.
 * fields = {rho, phi, T}: the views the script's unstack_fields makes ([BS,Nt,Nx,Ny], Nt fastest in the surrogate's
 * memory).  Rb: the radius grid as a view of the same logical shape that repeats R along the LAST axis (the script
 * broadcasts its 1-D R tensor that way) - zero strides on the other axes, unit stride on the axis the fields are
 * contiguous on.  K_*: the dense kernels of D_t, D_R, D_Z, D_RR, D_ZZ (:201-205).
 *   eq 0 continuity:  a0*D_t(rho) - (a1*R)*X(rho) - (a2*rho)*D_Z(phi) - a3*((D_RR(rho) + (1/R)*D_R(rho)) + D_ZZ(rho)),
 *                     X(f) = D_R(f)*D_Z(phi) - D_R(phi)*D_Z(f); coef = {a0,a1,a2,a3} = {1,1,2,D}, or the script's
 *                     norms=True scalars {2*dx*dy, dt, (2*dt*dy)*2, (4*dt)*D} folded in fp32 in its order
 *   eq 1 temperature: T*D_t(rho) + rho*D_t(T) - (rho*R)*X(T) + (T*R)*X(rho) + ((a0*rho)*T)*D_Z(phi)
 *                     + a3*((D_RR(T) + (1/R)*D_R(T)) + D_ZZ(T));  coef = {2*gamma, -, -, K} */
int pre_residual_jorek_f32(int eq, const pre_field_t fields[3], const pre_field_t *Rb, const pre_out_t *out,
                           const float *K_t, const float *K_R, const float *K_Z, const float *K_RR, const float *K_ZZ,
                           const float coef[4], int64_t B, int64_t T, int64_t X, int64_t Y, int flags, void *stream);

/* ---- 8f rank 4: 2-D spatial operators with boundary conditions -------------------------------
 * Utils/ConvOps_Spatial.py:83-136 (F.conv2d, 'valid') applied to a field padded by
 * Utils/boundary_conditions.py:81-185 (BoundaryManager.pad_signal), as the Gradient / Laplace /
 * Divergence / Curl of Utils/VectorConvOps_Spatial.py:33-165 do - fused: a 'same'-sized output whose
 * out-of-domain neighbours are mapped cells (periodic / neumann-outflow / symmetric) or a constant
 * (dirichlet).  Planes [B,X,Y] (the [BS,1,Nx,Ny] field with the channel squeezed), strides {sB,sX,sY}
 * with sY == 1; K is the 3x3 kernel (9 host floats, axes (Nx,Ny)), cross-shaped; otherwise
 * PRE_E_UNSUPPORTED and the caller pads with device ops and runs pre_stencil3d_f32. */
#define PRE_BC_CONSTANT  0     /* 'dirichlet' (value)            */
#define PRE_BC_REPLICATE 1     /* 'neumann', 'outflow'           */
#define PRE_BC_PERIODIC  2     /* 'periodic'                     */
#define PRE_BC_REFLECT   3     /* 'symmetric' (F.pad 'reflect')  */
typedef struct {
    int mode[4];               /* left, right (columns / Ny), top, bottom (rows / Nx) */
    float value[4];
} pre_bc_t;
int pre_spatial2d_bc_f32(const float *in, const int64_t in_strides[3], float *out, const int64_t out_strides[3],
                         const float *K /*host, 9*/, const pre_bc_t *bc /*host*/,
                         int64_t B, int64_t X, int64_t Y, int flags, void *stream);
/* out = K0(pad(in0)) + ratio*K1(pad(in1))   (Divergence: ratio 1; Curl: K0=grad_x on input_y, ratio -1) */
int pre_spatial2d_linear2_bc_f32(const float *in0, const int64_t s0[3], const float *in1, const int64_t s1[3],
                                 float *out, const int64_t out_strides[3], const float *K0, const float *K1,
                                 float ratio, const pre_bc_t *bc /*host*/,
                                 int64_t B, int64_t X, int64_t Y, int flags, void *stream);

/* ---- a9: periodic_bc_residual(u, wall) ---------------------------------------------------
 * Marginal/NS_Residuals_CP.py:468-478: the mismatch between two opposite edges of every [X,Y] plane, times dx (ABI v8):
 *   wall 0 'top':  u[...,0,:] - u[...,-1,:]     1 'bottom': u[...,-1,:] - u[...,0,:]      -> out [B,T,Y] contiguous
 *   wall 2 'left': u[...,:,0] - u[...,:,-1]     3 'right':  u[...,:,-1] - u[...,:,0]      -> out [B,T,X] contiguous
 * u: any strided [B,T,X,Y] view; fp32 (a - b) * dx, the reference's own two roundings. */
int pre_edge_residual_f32(const pre_field_t *u, int wall, float dx, int64_t B, int64_t T, int64_t X, int64_t Y, float *out,
                          void *stream);

/* ---- a10: marginal nonconformity score ------------------------------------------------
 * out = |a - b| (b may be NULL: |a|).  Marginal/Wave_Residuals_CP.py:219,280 */
int pre_absdiff_f32(const float *a, const float *b, float *out, int64_t n, void *stream);

/* ---- a12: modulation_func(a, b) = std(a-b, axis 0), ddof 0 ----------------------------
 * (Neural_PDE.UQ.inductive_cp, absent; Tests/test_advection_inv_sampling_marginal.py:428)
 * a, b: contiguous [n, M] (b may be NULL).
 * pre_std_axis0_f32: the whole calibration set resident; float32 sequential two-pass,
 *   the exact operation order numpy uses for an axis-0 reduction, + eps (Joint/MHD_Residuals_CP.py:350).
 * pre_moments_axis0_f64: streaming / sharded form; ACCUMULATES sum and sum of squares of
 *   (a-b) over the n rows into sum[M], sumsq[M] (fp64, caller zeroes them before the first
 *   chunk, all-reduces them across ranks), then pre_std_from_moments_f32 finishes.  Rows are
 *   row_stride elements apart (>= M; M for a dense [n,M] tensor), so a contiguous sub-range of
 *   cells - e.g. the interior t planes of an uncropped residual - can be reduced in place. */
int pre_std_axis0_f32(const float *a, const float *b, int64_t n, int64_t M, float eps, float *mod, void *stream);
int pre_moments_axis0_f64(const float *a, const float *b, int64_t n, int64_t M, int64_t row_stride,
                          double *sum, double *sumsq, void *stream);
int pre_std_from_moments_f32(const double *sum, const double *sumsq, int64_t n_total, int64_t M,
                             float eps, float *mod, void *stream);

/* ---- a13: ncf_metric_joint(a, b, modulation) = max_cells |a-b|/mod per sample -----------
 * (Tests/test_advection_inv_sampling_marginal.py:430-431).  a, b: contiguous [n,T,X,Y]
 * UNCROPPED, mod: [T,X,Y]; only cells with crop <= index < extent-crop on each of the last
 * three axes take part (crop_t/x/y = 1 reproduces the reference's [...,1:-1,1:-1,1:-1]).
 * scores[n] must be zero-filled by the caller; the kernel max-accumulates into it, so
 * per-slab calls over a split T axis compose.  NaN propagates like np.max: a NaN residual or modulation,
 * or 0/0, makes the sample's score NaN (sticky across calls); x/0 with x != 0 gives inf. */
int pre_joint_score_f32(const float *a, const float *b, const float *mod,
                        int64_t n, int64_t T, int64_t X, int64_t Y,
                        int crop_t, int crop_x, int crop_y, float *scores, void *stream);

/* Branch-and-bound form of the streaming joint score.  The score of a sample is a maximum, and for a segment (one row
 * x, 64 columns, the slab's planes)  max |r_c| / min mod_c  bounds every |r_c| / mod_c in it (correctly rounded division
 * is monotone), so a segment whose bound does not exceed the sample's best score so far - scores[i] from earlier slabs,
 * then the segment with the largest bound - is never read.  For the same modulation the scores are those of pre_joint_score_f32, bit for bit; on noise-like
 * residuals it reads well under 1 % of them.
 * A segment is 64 consecutive cells of the flattened (x, y) plane x 16 planes: NS = ceil(X*Y/64) segments per
 * chunk of planes, TC = ceil(T/16) chunks, index [tc][s] (a segment may straddle rows: a bound needs no geometry).
 *   pre_moments_segmax_f64: pre_moments_axis0_f64 for n samples of T contiguous planes [T,X,Y] each, sample i at
 *     a + i*row_stride, that ALSO writes, from the same read, segmax (uint32 [n][TC][NS]) = the bit pattern of
 *     max |a| of sample i over the segment, cells within crop_x / crop_y of the x / y rim excluded (0 if none is
 *     left).  sum, sumsq: [T*X*Y] as pre_moments_axis0_f64 (fp64 sums of the same terms; only the order of the
 *     additions may differ).
 *   pre_segmin_mod_f32: segmin[tc][s] = min of mod[T,X,Y] over the segment's uncropped cells; +inf if there is none,
 *     0 if the segment holds a NaN or a non-positive modulation (-> always read).
 *   pre_joint_score_pruned_f32: scores as pre_joint_score_f32 with crop_t = 0 over the same planes; TC*NS*4 + 256 bytes
 *     must fit the LDS of a workgroup (the work list lives there; gfx950: TC*NS <= 40896; else PRE_E_UNSUPPORTED).  flags (device uint32 [n], may be NULL; ABI v6): a sample
 *     whose bounds leave more than a quarter of its segments to read (a modulation that jumps between neighbouring
 *     cells) is not read segment by segment but flagged, flags[i] = 1 (else 0), for pre_joint_score_flagged_f32 - the
 *     full pass at its full speed over those samples.  stats (device, 3 x uint64, may be NULL): [0] += segments read
 *     (all of a flagged sample's), [1] += segments, [2] += samples flagged - what a streaming driver needs to drop the
 *     bounds for the slabs that follow, and a benchmark to say what fraction of the pass its data let it skip.
 * A t crop is applied by handing in the interior planes: a + crop_t*X*Y with T - 2*crop_t planes, row_stride unchanged. */
int pre_moments_segmax_f64(const float *a, int64_t row_stride, int64_t n, int64_t T, int64_t X, int64_t Y, int crop_x,
                           int crop_y, double *sum, double *sumsq, uint32_t *segmax, void *stream);
int pre_segmin_mod_f32(const float *mod, int64_t T, int64_t X, int64_t Y, int crop_x, int crop_y, float *segmin,
                       void *stream);
int pre_joint_score_pruned_f32(const float *res, int64_t row_stride, const float *mod, const uint32_t *segmax,
                               const float *segmin, int64_t n, int64_t T, int64_t X, int64_t Y, int crop_x, int crop_y,
                               float *scores, uint32_t *flags, unsigned long long *stats, void *stream);
/* The longest work list (TC*NS) pre_joint_score_pruned_f32 accepts on the CURRENT device - (LDS bytes a workgroup may
 * hold - 256) / 4; gfx950: 40896 - so that a driver can choose its route before it pays for the segment maxima (ABI v8). */
int64_t pre_joint_score_pruned_max_segments(void);
/* pre_joint_score_f32 over the samples i with flags[i] != 0 only (the others' workgroups leave at once): the pass a
 * driver launches behind pre_joint_score_pruned_f32 for the samples it flagged (ABI v6). */
int pre_joint_score_flagged_f32(const float *a, const float *b, const float *mod,
                                int64_t n, int64_t T, int64_t X, int64_t Y,
                                int crop_t, int crop_x, int crop_y, const uint32_t *flags, float *scores, void *stream);

/* ---- a11: calibrate(scores, n, alpha) ---------------------------------------------------
 * (Neural_PDE.UQ.inductive_cp, absent; call sites Marginal/Wave_Residuals_CP.py:288,
 * Joint/Burgers_Residuals_CP.py:283).  Exact order statistics on the order-preserving uint32 image of fp32 (n <= 240:
 * the cell's column sorted in the registers of one or two lanes; <= 2048: the tile held in a workgroup's registers, bucket
 * select over its exact value window; above: sample-guided bucket / MSD radix select streaming the rows twice or more -
 * the scores are read once for every n <= 2048); result is bit-for-bit an input value.
 * ks: host array of 0-based sorted ranks (the caller derives them from alpha).
 * pre_kth_f32:       scores[N]            -> out[nk]; any NaN score makes every result NaN (np.quantile)
 * pre_kth_axis0_f32: scores[n, M] contiguous -> out[nk, M]   (per-cell over the batch axis),
 *                    n < 2^31 (32-bit counters from n = 65536 on), ks in any order (out[j] belongs to ks[j]),
 *                    nk <= 64; a NaN anywhere in a cell's column makes every result of that cell NaN (np.quantile). */
int pre_kth_f32(const float *scores, int64_t N, const int64_t *ks /*host*/, int nk, float *out, void *stream);
int pre_kth_axis0_f32(const float *scores, int64_t n, int64_t M, const int32_t *ks /*host*/, int nk,
                      float *out, void *stream);
/* ... with the rows row_stride >= M elements apart (ABI v6).  A power-of-two distance between the rows of a cell's column
 * - M = 2^k cells - lines every row of a tile up on the same low address bits and costs the column sweeps 4-11 % on
 * the MI355X (profiles/r03/row_pitch.txt); a driver that owns the buffer pads its rows by 64 floats. */
int pre_kth_axis0_strided_f32(const float *scores, int64_t row_stride, int64_t n, int64_t M, const int32_t *ks /*host*/, int nk,
                              float *out, void *stream);
/* ... for `planes` independent score matrices in ONE launch (ABI v8): plane p = [n, M] rows at scores + p*plane_stride
 * (rows row_stride apart), its results at out + p*out_plane_stride + j*out_rank_stride + cell.  What a driver that keeps a
 * TIME-MAJOR residual slab ([T][n][plane (+pad)], the zero-copy send layout of the sharded marginal calibration) calls
 * once per slab, or once per received run of planes, instead of once per plane: the tiles of all planes share one grid,
 * so the ragged last round of workgroups happens once, not `planes` times (Marginal/NS_Residuals_CP.py:310-327 is one
 * np.quantile over the whole [n, Nt, Nx, Ny] array).  planes = 1: pre_kth_axis0_strided_f32. */
int pre_kth_axis0_planes_f32(const float *scores, int64_t plane_stride, int64_t row_stride, int64_t planes, int64_t n, int64_t M,
                             const int32_t *ks /*host*/, int nk, float *out, int64_t out_rank_stride, int64_t out_plane_stride,
                             void *stream);

/* ---- a14: emp_cov / emp_cov_joint / filter_sims_joint -----------------------------------
 * (Joint/Burgers_Residuals_CP.py:298-300; Tests/test_advection_inv_sampling_marginal.py:465)
 * y: [n, M] contiguous; lo, hi: [M] (broadcast over n) or [n, M] when per_sample_bounds != 0.
 * pre_cov_count_f32:  count[0] += #{ lo <= y <= hi }           (uint64, caller zeroes)
 * pre_cov_joint_f32:  inside[i] = all_cells(lo <= y[i] <= hi)  (uint8 per sample; caller fills with 1) */
int pre_cov_count_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M,
                      int per_sample_bounds, unsigned long long *count, void *stream);
/* filter_sims_within_bounds (Active_Learning/Advection_AL_Marginal.py:169-198): counts[i] +=
 * #cells of sample i with lo <= y <= hi (outside == 0) or y <= lo || y >= hi (outside != 0);
 * uint32 per sample, caller zeroes; the caller divides by M and compares with the threshold. */
int pre_cov_rowcount_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M,
                         int per_sample_bounds, int outside, uint32_t *counts, void *stream);
int pre_cov_joint_f32(const float *y, const float *lo, const float *hi, int64_t n, int64_t M,
                      int per_sample_bounds, uint8_t *inside, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CP_PRE_HIP_H */
