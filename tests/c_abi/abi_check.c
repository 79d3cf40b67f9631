/* A C99 client of libcp_pre_hip.so: no Python, no torch - only the HIP runtime for device memory.
 * Shows what a maintainer binding the library from another language sees, and checks each call
 * against plain C loops written here (the reference's arithmetic: zero-padded cross-correlation,
 * Utils/ConvOps_2d.py:149; the NS momentum expression, Marginal/NS_Residuals_CP.py:231-240; an order
 * statistic per cell; the streaming joint chain; MHD induction, JOREK continuity, Burgers and the 1-D stencil; the
 * two-operator linear residual, the kernel gradient, the boundary-condition operators, the resident std, the coverage
 * counts, the multi-plane select and the periodic-wall residual): every symbol of cp_pre_hip.h is called.
 * Built and run by tests/test_gpu_parity.py::test_c_abi_client.  Exit code 0 = all ok.
 *
 *   hipcc -x c tests/c_abi/abi_check.c -Iinclude -Lcp_pre_amd -lcp_pre_hip -Wl,-rpath,$PWD/cp_pre_amd -o abi_check
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cp_pre_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define EXPECT(cond, what) do { if (!(cond)) { printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } else { printf("ok:   %s\n", what); } } while (0)

enum { B = 3, T = 6, X = 10, Y = 64, N = B * T * X * Y };

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f + 0.5f; }

/* zero-padded cross-correlation with a dense 3x3x3 kernel, axes (Nt,Nx,Ny) */
static void xcorr27(const float *f, const float *K, float *out)
{
    for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) for (int x = 0; x < X; ++x) for (int y = 0; y < Y; ++y) {
        double acc = 0.0;
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) for (int d = 0; d < 3; ++d) {
            const int tt = t + a - 1, xx = x + c - 1, yy = y + d - 1;
            if (tt >= 0 && tt < T && xx >= 0 && xx < X && yy >= 0 && yy < Y)
                acc += (double)K[(a * 3 + c) * 3 + d] * f[((b * T + tt) * X + xx) * Y + yy];
        }
        out[((b * T + t) * X + x) * Y + y] = (float)acc;
    }
}

static double rel_err(const float *a, const float *b, int n)
{
    double num = 0.0, den = 0.0;
    for (int i = 0; i < n; ++i) { const double d = fabs((double)a[i] - b[i]); if (d > num) num = d; if (fabs(b[i]) > den) den = fabs(b[i]); }
    return den > 0 ? num / den : num;
}

static int cmp_float(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }

int main(void)
{
    int failures = 0;
    setvbuf(stdout, NULL, _IONBF, 0);              /* a crash must not swallow the lines before it */
    EXPECT(pre_abi_version() == PRE_ABI_VERSION && PRE_ABI_VERSION == 8, "pre_abi_version() == PRE_ABI_VERSION == 8");

    /* ---- the reference's kernels: kernel_3d(stencil, axis) with the stencil on slab 1 (Utils/ConvOps_2d.py:67-79) */
    float Kt[27] = {0}, Kx[27] = {0}, Ky[27] = {0}, Kl[27] = {0};
    Kt[(0 * 3 + 1) * 3 + 1] = -1.f; Kt[(2 * 3 + 1) * 3 + 1] = 1.f;               /* 't', 1                        */
    Kx[(1 * 3 + 0) * 3 + 1] = -1.f; Kx[(1 * 3 + 2) * 3 + 1] = 1.f;               /* 'x', 1                        */
    memcpy(Ky, Kt, sizeof Kt);                                                     /* 'y', 1 == 't', 1 (:72-73)     */
    Kl[(1 * 3 + 0) * 3 + 1] = Kl[(1 * 3 + 2) * 3 + 1] = Kl[(1 * 3 + 1) * 3 + 0] = Kl[(1 * 3 + 1) * 3 + 2] = 1.f;
    Kl[(1 * 3 + 1) * 3 + 1] = -4.f;                                                /* ('x','y'), 2                  */

    static float hu[N], hv[N], hp[N], want[N], got[N], tmp[10][N];
    unsigned seed = 12345u;
    for (int i = 0; i < N; ++i) { hu[i] = frand(&seed); hv[i] = frand(&seed); hp[i] = frand(&seed); }
    float *du, *dv, *dp, *dout;
    CHECK_HIP(hipMalloc((void **)&du, sizeof hu)); CHECK_HIP(hipMalloc((void **)&dv, sizeof hv));
    CHECK_HIP(hipMalloc((void **)&dp, sizeof hp)); CHECK_HIP(hipMalloc((void **)&dout, sizeof got));
    CHECK_HIP(hipMemcpy(du, hu, sizeof hu, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    const pre_field_t fu = {du, (int64_t)T * X * Y, (int64_t)X * Y, Y, 1}, fv = {dv, (int64_t)T * X * Y, (int64_t)X * Y, Y, 1},
                      fp = {dp, (int64_t)T * X * Y, (int64_t)X * Y, Y, 1};
    const pre_out_t fo = {dout, (int64_t)T * X * Y, (int64_t)X * Y, Y, 1};

    /* ---- a4: ConvOperator.convolution with the Laplacian as a tap list, then with PRE_FLAG_ABS */
    {
        float w[5] = {1.f, 1.f, -4.f, 1.f, 1.f};
        int32_t off[15] = {0, -1, 0, 0, 0, -1, 0, 0, 0, 0, 0, 1, 0, 1, 0};
        int rc = pre_stencil3d_f32(&fu, &fo, w, off, 5, B, T, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        xcorr27(hu, Kl, want);
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_stencil3d_f32: 5-point Laplacian vs C loops (<= 1e-5)");
        rc = pre_stencil3d_f32(&fu, &fo, w, off, 5, B, T, X, Y, PRE_FLAG_ABS, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) want[i] = fabsf(want[i]);
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_stencil3d_f32: PRE_FLAG_ABS epilogue");
    }

    /* ---- a9: fused NS momentum residual vs the operator-by-operator expression */
    {
        const float dt = 0.01f, dx = 0.1f, dy = 0.2f, nu = 0.001f;
        int rc = pre_residual_ns_momentum_f32(&fu, &fv, &fp, &fo, Kt, Kx, Ky, Kl, dt, dx, dy, nu, B, T, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        xcorr27(hu, Kt, tmp[0]); xcorr27(hu, Kx, tmp[1]); xcorr27(hu, Ky, tmp[2]); xcorr27(hu, Kl, tmp[3]); xcorr27(hp, Kx, tmp[4]);
        xcorr27(hv, Kt, tmp[5]); xcorr27(hv, Kx, tmp[6]); xcorr27(hv, Ky, tmp[7]); xcorr27(hv, Kl, tmp[8]); xcorr27(hp, Ky, tmp[9]);
        for (int i = 0; i < N; ++i) {
            const float rx = tmp[0][i] * dx * dy + hu[i] * tmp[1][i] * dt * dy + hv[i] * tmp[2][i] * dt * dx - nu * tmp[3][i] * dt + tmp[4][i] * dt * dy;
            const float ry = tmp[5][i] * dx * dy + hu[i] * tmp[6][i] * dt * dx + hv[i] * tmp[7][i] * dt * dy - nu * tmp[8][i] * dt + tmp[9][i] * dt * dx;
            want[i] = rx + ry;
        }
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_residual_ns_momentum_f32 vs operator-by-operator C loops (<= 1e-5)");
        /* PRE_FLAG_HALO_X (ABI v7): rows [3, 8) of the same fields as an x-slab whose rows 2 and 8 exist in memory -
         * the slab's residual rows are the whole grid's (an x-slab driver: T whole, one halo row per side) */
        {
            enum { X0 = 3, XS = 5 };
            const pre_field_t su = {du + X0 * Y, fu.sB, fu.sT, fu.sX, 1}, sv = {dv + X0 * Y, fv.sB, fv.sT, fv.sX, 1},
                              sp = {dp + X0 * Y, fp.sB, fp.sT, fp.sX, 1};
            float *dslab;
            static float hslab[B * T * XS * Y];
            CHECK_HIP(hipMalloc((void **)&dslab, sizeof hslab));
            const pre_out_t so = {dslab, (int64_t)T * XS * Y, (int64_t)XS * Y, Y, 1};
            rc = pre_residual_ns_momentum_f32(&su, &sv, &sp, &so, Kt, Kx, Ky, Kl, dt, dx, dy, nu, B, T, XS, Y, PRE_FLAG_HALO_X, st);
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(hslab, dslab, sizeof hslab, hipMemcpyDeviceToHost));
            int same = rc == PRE_OK;
            for (int b = 0; b < B && same; ++b) for (int t = 0; t < T && same; ++t) for (int x = 0; x < XS && same; ++x)
                same = memcmp(&hslab[((b * T + t) * XS + x) * Y], &got[((b * T + t) * X + X0 + x) * Y], Y * sizeof(float)) == 0;
            EXPECT(same, "PRE_FLAG_HALO_X: an x-slab with its halo rows == the whole grid's rows, bit for bit");
            CHECK_HIP(hipFree(dslab));
        }
        float Kbad[27];
        memcpy(Kbad, Kl, sizeof Kl);
        Kbad[0] = 1.f;                                                             /* a corner tap: off the star */
        rc = pre_residual_ns_momentum_f32(&fu, &fv, &fp, &fo, Kt, Kx, Ky, Kbad, dt, dx, dy, nu, B, T, X, Y, 0, st);
        EXPECT(rc == PRE_E_UNSUPPORTED, "fused residual declines a non-star kernel with PRE_E_UNSUPPORTED");
    }

    /* ---- a10/a11: |u - v| scores and the per-cell order statistics, exact */
    {
        enum { n = B * T, M = X * Y };                                             /* view the field as [n, M] scores */
        CHECK_HIP(hipStreamSynchronize(st));
        int rc = pre_absdiff_f32(du, dv, dout, N, st);
        int32_t ks[3] = {0, n / 2, n - 1};
        float *dq;
        static float hq[3 * M], col[n];
        CHECK_HIP(hipMalloc((void **)&dq, sizeof hq));
        int rc2 = pre_kth_axis0_f32(dout, n, M, ks, 3, dq, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost));
        int exact = 1;
        for (int c = 0; c < M; ++c) {
            for (int i = 0; i < n; ++i) col[i] = fabsf(hu[i * M + c] - hv[i * M + c]);
            qsort(col, n, sizeof(float), cmp_float);
            for (int j = 0; j < 3; ++j) exact &= (hq[j * M + c] == col[ks[j]]);
        }
        EXPECT(rc == PRE_OK && rc2 == PRE_OK && exact, "pre_absdiff_f32 + pre_kth_axis0_f32: bit-exact vs qsort per cell");
        /* ranks in any order: out[j] belongs to ks[j] */
        int32_t mixed[3] = {ks[2], ks[0], ks[1]};
        int rc3 = pre_kth_axis0_f32(dout, n, M, mixed, 3, dq, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost));
        int exact2 = 1;
        for (int c = 0; c < M; ++c) {
            for (int i = 0; i < n; ++i) col[i] = fabsf(hu[i * M + c] - hv[i * M + c]);
            qsort(col, n, sizeof(float), cmp_float);
            for (int j = 0; j < 3; ++j) exact2 &= (hq[j * M + c] == col[mixed[j]]);
        }
        EXPECT(rc3 == PRE_OK && exact2, "pre_kth_axis0_f32: ranks in any order, out[j] <-> ks[j]");
        /* rows with a pitch: the first M - 2 cells of every row, rows still M apart */
        int rc4 = pre_kth_axis0_strided_f32(dout, M, n, M - 2, ks, 3, dq, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost));
        int exact3 = 1;
        for (int c = 0; c < M - 2; ++c) {
            for (int i = 0; i < n; ++i) col[i] = fabsf(hu[i * M + c] - hv[i * M + c]);
            qsort(col, n, sizeof(float), cmp_float);
            for (int j = 0; j < 3; ++j) exact3 &= (hq[j * (M - 2) + c] == col[ks[j]]);
        }
        EXPECT(rc4 == PRE_OK && exact3, "pre_kth_axis0_strided_f32: rows with a pitch");
        EXPECT(pre_kth_axis0_strided_f32(dout, M - 3, n, M - 2, ks, 3, dq, st) == PRE_E_RANGE, "row_stride < M -> PRE_E_RANGE");
        int32_t bad[2] = {2, (int32_t)n};
        EXPECT(pre_kth_axis0_f32(dout, n, M, bad, 2, dq, st) == PRE_E_RANGE, "rank >= n -> PRE_E_RANGE");
        EXPECT(pre_kth_axis0_f32(NULL, n, M, ks, 3, dq, st) == PRE_E_NULL, "null pointer -> PRE_E_NULL");
        CHECK_HIP(hipFree(dq));
    }

    /* ---- a12/a13/a11, the streaming joint chain (Joint/Burgers_Residuals_CP.py:272-285): segment maxima + moments in
     * one read, modulation from the moments, segment minima, branch-and-bound score with its flagged full pass, scalar
     * q-hat - against the plain recipe in C (std over samples, max |r|/mod over the cropped cells, qsort) */
    {
        enum { n = B, M = T * X * Y, NS = (X * Y + 63) / 64, TC = (T + 15) / 16 };
        double *dsum, *dsq;
        uint32_t *dsegmax, *dflags;
        unsigned long long *dstats;
        float *dmod, *dsegmin, *dsc, *dq;
        static float hmod[M], hsc[n], hq[2], wmod[M], wsc[n];
        unsigned long long hstats[3];
        CHECK_HIP(hipMalloc((void **)&dsum, M * sizeof(double))); CHECK_HIP(hipMalloc((void **)&dsq, M * sizeof(double)));
        CHECK_HIP(hipMalloc((void **)&dsegmax, n * TC * NS * 4)); CHECK_HIP(hipMalloc((void **)&dflags, n * 4));
        CHECK_HIP(hipMalloc((void **)&dstats, 24)); CHECK_HIP(hipMalloc((void **)&dmod, M * 4));
        CHECK_HIP(hipMalloc((void **)&dsegmin, TC * NS * 4)); CHECK_HIP(hipMalloc((void **)&dsc, n * 4));
        CHECK_HIP(hipMalloc((void **)&dq, 8));
        CHECK_HIP(hipMemsetAsync(dsum, 0, M * sizeof(double), st)); CHECK_HIP(hipMemsetAsync(dsq, 0, M * sizeof(double), st));
        CHECK_HIP(hipMemsetAsync(dstats, 0, 24, st)); CHECK_HIP(hipMemsetAsync(dsc, 0, n * 4, st));
        int rc = pre_moments_segmax_f64(du, M, n, T, X, Y, 1, 1, dsum, dsq, dsegmax, st);
        int rc2 = pre_std_from_moments_f32(dsum, dsq, n, M, 0.f, dmod, st);
        int rc3 = pre_segmin_mod_f32(dmod, T, X, Y, 1, 1, dsegmin, st);
        int rc4 = pre_joint_score_pruned_f32(du, M, dmod, dsegmax, dsegmin, n, T, X, Y, 1, 1, dsc, dflags, dstats, st);
        int rc5 = pre_joint_score_flagged_f32(du, NULL, dmod, n, T, X, Y, 0, 1, 1, dflags, dsc, st);
        int64_t kq[2] = {0, n - 1};
        int rc6 = pre_kth_f32(dsc, n, kq, 2, dq, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hmod, dmod, sizeof hmod, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hsc, dsc, sizeof hsc, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hstats, dstats, 24, hipMemcpyDeviceToHost));
        for (int c = 0; c < M; ++c) {
            double s1 = 0, s2 = 0;
            for (int i = 0; i < n; ++i) { s1 += hu[i * M + c]; s2 += (double)hu[i * M + c] * hu[i * M + c]; }
            const double mean = s1 / n, var = s2 / n - mean * mean;
            wmod[c] = (float)sqrt(var > 0 ? var : 0);
        }
        for (int i = 0; i < n; ++i) {
            float m = 0.f;
            for (int t = 0; t < T; ++t) for (int x = 1; x < X - 1; ++x) for (int y = 1; y < Y - 1; ++y) {
                const int c = (t * X + x) * Y + y;
                const float q = fabsf(hu[i * M + c]) / hmod[c];               /* the device's own modulation: same quotients */
                if (q > m) m = q;
            }
            wsc[i] = m;
        }
        int same = 1;
        for (int i = 0; i < n; ++i) same &= (hsc[i] == wsc[i]);
        float srt[n];
        memcpy(srt, wsc, sizeof srt);
        qsort(srt, n, sizeof(float), cmp_float);
        EXPECT(rc == PRE_OK && rc2 == PRE_OK && rel_err(hmod, wmod, M) <= 1e-6, "pre_moments_segmax_f64 + pre_std_from_moments_f32 vs C std (<= 1e-6)");
        EXPECT(rc3 == PRE_OK && rc4 == PRE_OK && rc5 == PRE_OK && same, "pre_segmin_mod_f32 + pre_joint_score_pruned_f32 + pre_joint_score_flagged_f32: scores bit-exact vs C max |r|/mod");
        EXPECT(hstats[1] == (unsigned long long)n * TC * NS && hstats[0] <= hstats[1], "pruned score statistics: segments read <= segments");
        EXPECT(rc6 == PRE_OK && hq[0] == srt[0] && hq[1] == srt[n - 1], "pre_kth_f32: scalar order statistics of the scores");
        /* the un-pruned route through the plain entry points gives the same scores */
        CHECK_HIP(hipMemsetAsync(dsc, 0, n * 4, st));
        CHECK_HIP(hipMemsetAsync(dsum, 0, M * sizeof(double), st)); CHECK_HIP(hipMemsetAsync(dsq, 0, M * sizeof(double), st));
        int rc7 = pre_moments_axis0_f64(du, NULL, n, M, M, dsum, dsq, st);
        int rc8 = pre_joint_score_f32(du, NULL, dmod, n, T, X, Y, 0, 1, 1, dsc, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hsc, dsc, sizeof hsc, hipMemcpyDeviceToHost));
        same = 1;
        for (int i = 0; i < n; ++i) same &= (hsc[i] == wsc[i]);
        EXPECT(rc7 == PRE_OK && rc8 == PRE_OK && same, "pre_moments_axis0_f64 + pre_joint_score_f32: the same scores");
        hipFree(dsum); hipFree(dsq); hipFree(dsegmax); hipFree(dflags); hipFree(dstats); hipFree(dmod); hipFree(dsegmin); hipFree(dsc); hipFree(dq);
    }

    /* ---- a9: MHD induction (Marginal/MHD_Residuals_CP.py:259-268), fields {rho,u,v,p,Bx,By} = {u,u,v,p,p,v} here */
    {
        const pre_field_t f6[6] = {fu, fu, fv, fp, fp, fv};
        int rc = pre_residual_mhd_f32(3, f6, &fo, Kt, Kx, Ky, 5.0 / 3.0, B, T, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        const float *u = hu, *v = hv, *bx = hp, *by = hv;
        xcorr27(bx, Kt, tmp[0]); xcorr27(u, Ky, tmp[1]); xcorr27(v, Ky, tmp[2]); xcorr27(bx, Ky, tmp[3]); xcorr27(by, Ky, tmp[4]);
        xcorr27(by, Kt, tmp[5]); xcorr27(u, Kx, tmp[6]); xcorr27(v, Kx, tmp[7]); xcorr27(bx, Kx, tmp[8]); xcorr27(by, Kx, tmp[9]);
        for (int i = 0; i < N; ++i) {
            const float rx = tmp[0][i] - by[i] * tmp[1][i] + bx[i] * tmp[2][i] - v[i] * tmp[3][i] + u[i] * tmp[4][i];
            const float ry = tmp[5][i] + by[i] * tmp[6][i] - bx[i] * tmp[7][i] - v[i] * tmp[8][i] + u[i] * tmp[9][i];
            want[i] = rx + ry;
        }
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_residual_mhd_f32 (induction) vs operator-by-operator C loops (<= 1e-5)");
    }

    /* ---- a9: reduced MHD / JOREK continuity (Marginal/JOREK_residuals_CP.py:207-221), R broadcast along the last axis */
    {
        static float hR[Y];
        float *dR;
        float Krr[27] = {0}, Kzz[27] = {0};
        const float D = 3.4f, g = 5.0f / 3.0f;
        for (int y = 0; y < Y; ++y) hR[y] = 1.2f + 0.01f * y;
        Krr[(1 * 3 + 0) * 3 + 1] = Krr[(1 * 3 + 2) * 3 + 1] = g; Krr[(1 * 3 + 1) * 3 + 1] = -2.f * g;       /* 'x', 2, scale gamma */
        Kzz[(0 * 3 + 1) * 3 + 1] = Kzz[(2 * 3 + 1) * 3 + 1] = g; Kzz[(1 * 3 + 1) * 3 + 1] = -2.f * g;       /* 'y', 2: along Nt   */
        CHECK_HIP(hipMalloc((void **)&dR, sizeof hR));
        CHECK_HIP(hipMemcpy(dR, hR, sizeof hR, hipMemcpyHostToDevice));
        const pre_field_t f3[3] = {fu, fv, fp}, fR = {dR, 0, 0, 0, 1};
        const float coef[4] = {1.f, 1.f, 2.f, D};
        int rc = pre_residual_jorek_f32(0, f3, &fR, &fo, Kt, Kx, Ky, Krr, Kzz, coef, B, T, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        const float *rho = hu, *phi = hv;
        xcorr27(rho, Kt, tmp[0]); xcorr27(rho, Kx, tmp[1]); xcorr27(phi, Ky, tmp[2]); xcorr27(phi, Kx, tmp[3]); xcorr27(rho, Ky, tmp[4]);
        xcorr27(rho, Krr, tmp[5]); xcorr27(rho, Kzz, tmp[6]);
        for (int i = 0; i < N; ++i) {
            const float R = hR[i % Y];
            want[i] = tmp[0][i] - R * (tmp[1][i] * tmp[2][i] - tmp[3][i] * tmp[4][i]) - 2.f * rho[i] * tmp[2][i]
                      - D * (tmp[5][i] + (1.f / R) * tmp[1][i] + tmp[6][i]);
        }
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_residual_jorek_f32 (continuity) vs operator-by-operator C loops (<= 1e-5)");
        hipFree(dR);
    }

    /* ---- a5/a9 1-D: pre_stencil2d_f32 and the Burgers residual (Joint/Burgers_Residuals_CP.py:182-187) on [B*T, X, Y]
     * read as a 1-D problem [BS, Nt, Nx] = [B*T, X, Y]; 3x3 kernels, axes (Nt, Nx) */
    {
        enum { BS = B * T };
        const int64_t st3[3] = {(int64_t)X * Y, Y, 1};
        float k_t[9] = {0, -1, 0, 0, 0, 0, 0, 1, 0}, k_x[9] = {0, 0, 0, -1, 0, 1, 0, 0, 0}, k_xx[9] = {0, 0, 0, 1, -2, 1, 0, 0, 0};
        float w[3] = {1.f, -2.f, 1.f};
        int32_t off[6] = {0, -1, 0, 0, 0, 1};
        int rc = pre_stencil2d_f32(du, st3, dout, st3, w, off, 3, BS, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        /* D_xx(u)[b,t,x] = u[x-1] - 2u[x] + u[x+1] along the last axis, zero padded */
        for (int r = 0; r < BS * X; ++r) for (int y = 0; y < Y; ++y) {
            const float *row = hu + r * Y;
            want[r * Y + y] = (y > 0 ? row[y - 1] : 0.f) - 2.f * row[y] + (y < Y - 1 ? row[y + 1] : 0.f);
        }
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_stencil2d_f32: D_xx tap list vs C loops (<= 1e-5)");
        const float dx = 2.f / 64, dt = 1.25f / 10, nu = 0.002f, c3 = 2.f * dt / dx;
        rc = pre_residual_burgers_f32(du, st3, dout, st3, k_t, k_x, k_xx, dx, dt, nu, c3, BS, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        for (int b = 0; b < BS; ++b) for (int t = 0; t < X; ++t) for (int x = 0; x < Y; ++x) {
            const float *u = hu + (size_t)b * X * Y;
            const float c = u[t * Y + x];
            const float Dt = (t < X - 1 ? u[(t + 1) * Y + x] : 0.f) - (t > 0 ? u[(t - 1) * Y + x] : 0.f);
            const float Dx = (x < Y - 1 ? u[t * Y + x + 1] : 0.f) - (x > 0 ? u[t * Y + x - 1] : 0.f);
            const float Dxx = (x > 0 ? u[t * Y + x - 1] : 0.f) - 2.f * c + (x < Y - 1 ? u[t * Y + x + 1] : 0.f);
            want[((size_t)b * X + t) * Y + x] = dx * Dt + dt * c * Dx - nu * Dxx * c3;
        }
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_residual_burgers_f32 vs C loops (<= 1e-5)");
    }

    /* ---- a9: res = D_x(u) + ratio * D_y(v) (Marginal/NS_Residuals_CP.py:222-228) */
    {
        const float ratio = 0.75f;
        int rc = pre_residual_linear2_f32(&fu, &fv, &fo, Kx, Ky, ratio, B, T, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        xcorr27(hu, Kx, tmp[0]); xcorr27(hv, Ky, tmp[1]);
        for (int i = 0; i < N; ++i) want[i] = tmp[0][i] + ratio * tmp[1][i];
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-5, "pre_residual_linear2_f32 vs C loops (<= 1e-5)");
    }

    /* ---- 8f rank 3: gradient of the stencil with respect to its dense 3x3x3 kernel (Physics_Informed/Wave_FNO_PI.py:202-228) */
    {
        double *dgk, hgk[27], wgk[27] = {0};
        CHECK_HIP(hipMalloc((void **)&dgk, sizeof hgk));
        CHECK_HIP(hipMemsetAsync(dgk, 0, sizeof hgk, st));
        int rc = pre_stencil3d_wgrad_f32(&fu, &fv, 3, 3, 3, B, T, X, Y, dgk, st);       /* x = u, upstream gradient g = v */
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hgk, dgk, sizeof hgk, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) for (int x = 0; x < X; ++x) for (int y = 0; y < Y; ++y)
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) for (int d = 0; d < 3; ++d) {
                const int tt = t + a - 1, xx = x + c - 1, yy = y + d - 1;
                if (tt >= 0 && tt < T && xx >= 0 && xx < X && yy >= 0 && yy < Y)
                    wgk[(a * 3 + c) * 3 + d] += (double)hv[((b * T + t) * X + x) * Y + y] * hu[((b * T + tt) * X + xx) * Y + yy];
            }
        double worst = 0;
        for (int i = 0; i < 27; ++i) { const double e = fabs(hgk[i] - wgk[i]) / fabs(wgk[i]); if (e > worst) worst = e; }
        EXPECT(rc == PRE_OK && worst <= 1e-5, "pre_stencil3d_wgrad_f32: 27 kernel gradients vs C loops (fp32 partial sums per thread, fp64 across blocks: <= 1e-5)");
        EXPECT(pre_stencil3d_wgrad_f32(&fu, &fv, 5, 3, 3, B, T, X, Y, dgk, st) == PRE_E_UNSUPPORTED, "kernel extent 5 -> PRE_E_UNSUPPORTED");
        hipFree(dgk);
    }

    /* ---- 8f rank 4: spatial operators on [BS, X, Y] planes with boundary conditions (Utils/ConvOps_Spatial.py:83-136 on a
     * field padded by Utils/boundary_conditions.py:81-185): a 'same'-sized cross-correlation whose out-of-domain
     * neighbours are wrapped (periodic), clamped (neumann), mirrored (symmetric) or a constant (dirichlet) */
    {
        enum { BS = B * T };
        const int64_t st3[3] = {(int64_t)X * Y, Y, 1};
        const float Kgx[9] = {0, -0.5f, 0, 0, 0, 0, 0, 0.5f, 0}, Kgy[9] = {0, 0, 0, -0.5f, 0, 0.5f, 0, 0, 0};   /* axes (Nx, Ny) */
        const pre_bc_t bc = {{PRE_BC_PERIODIC, PRE_BC_PERIODIC, PRE_BC_REPLICATE, PRE_BC_CONSTANT}, {0.f, 0.f, 0.f, 0.25f}};
        const pre_bc_t bc2 = {{PRE_BC_REFLECT, PRE_BC_CONSTANT, PRE_BC_PERIODIC, PRE_BC_PERIODIC}, {0.f, -1.5f, 0.f, 0.f}};
#define AT(f, b, x, y, BC) ((y) < 0 ? ((BC).mode[0] == PRE_BC_CONSTANT ? (BC).value[0] : (f)[((b) * X + (x)) * Y + ((BC).mode[0] == PRE_BC_PERIODIC ? Y - 1 : (BC).mode[0] == PRE_BC_REFLECT ? 1 : 0)]) \
                         : (y) >= Y ? ((BC).mode[1] == PRE_BC_CONSTANT ? (BC).value[1] : (f)[((b) * X + (x)) * Y + ((BC).mode[1] == PRE_BC_PERIODIC ? 0 : (BC).mode[1] == PRE_BC_REFLECT ? Y - 2 : Y - 1)]) \
                         : (x) < 0 ? ((BC).mode[2] == PRE_BC_CONSTANT ? (BC).value[2] : (f)[((b) * X + ((BC).mode[2] == PRE_BC_PERIODIC ? X - 1 : (BC).mode[2] == PRE_BC_REFLECT ? 1 : 0)) * Y + (y)]) \
                         : (x) >= X ? ((BC).mode[3] == PRE_BC_CONSTANT ? (BC).value[3] : (f)[((b) * X + ((BC).mode[3] == PRE_BC_PERIODIC ? 0 : (BC).mode[3] == PRE_BC_REFLECT ? X - 2 : X - 1)) * Y + (y)]) \
                         : (f)[((b) * X + (x)) * Y + (y)])
        int rc = pre_spatial2d_bc_f32(du, st3, dout, st3, Kgx, &bc, BS, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        for (int b = 0; b < BS; ++b) for (int x = 0; x < X; ++x) for (int y = 0; y < Y; ++y)
            want[(b * X + x) * Y + y] = 0.5f * AT(hu, b, x + 1, y, bc) - 0.5f * AT(hu, b, x - 1, y, bc);
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-6, "pre_spatial2d_bc_f32: d/dx with neumann top, dirichlet bottom vs C loops");
        rc = pre_spatial2d_bc_f32(du, st3, dout, st3, Kgy, &bc2, BS, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        for (int b = 0; b < BS; ++b) for (int x = 0; x < X; ++x) for (int y = 0; y < Y; ++y)
            want[(b * X + x) * Y + y] = 0.5f * AT(hu, b, x, y + 1, bc2) - 0.5f * AT(hu, b, x, y - 1, bc2);
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-6, "pre_spatial2d_bc_f32: d/dy with symmetric left, dirichlet right vs C loops");
        /* Divergence: D_x(u) + D_y(v); Curl: ratio -1 (Utils/VectorConvOps_Spatial.py:97-165) */
        rc = pre_spatial2d_linear2_bc_f32(du, st3, dv, st3, dout, st3, Kgx, Kgy, -1.f, &bc, BS, X, Y, 0, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        for (int b = 0; b < BS; ++b) for (int x = 0; x < X; ++x) for (int y = 0; y < Y; ++y)
            want[(b * X + x) * Y + y] = (0.5f * AT(hu, b, x + 1, y, bc) - 0.5f * AT(hu, b, x - 1, y, bc))
                                        - (0.5f * AT(hv, b, x, y + 1, bc) - 0.5f * AT(hv, b, x, y - 1, bc));
        EXPECT(rc == PRE_OK && rel_err(got, want, N) <= 1e-6, "pre_spatial2d_linear2_bc_f32: K0(u) - K1(v) with mixed boundaries vs C loops");
#undef AT
        const float Kdense[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        EXPECT(pre_spatial2d_bc_f32(du, st3, dout, st3, Kdense, &bc, BS, X, Y, 0, st) == PRE_E_UNSUPPORTED, "a corner tap -> PRE_E_UNSUPPORTED");
    }

    /* ---- a12 resident form: modulation_func = std over the batch axis, numpy's float32 order */
    {
        enum { n = B * T, M = X * Y };
        float *dmod;
        static float hmod[M], wmod[M];
        CHECK_HIP(hipMalloc((void **)&dmod, sizeof hmod));
        int rc = pre_std_axis0_f32(du, dv, n, M, 0.f, dmod, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hmod, dmod, sizeof hmod, hipMemcpyDeviceToHost));
        for (int c = 0; c < M; ++c) {
            double s1 = 0, s2 = 0;
            for (int i = 0; i < n; ++i) s1 += (double)hu[i * M + c] - hv[i * M + c];
            for (int i = 0; i < n; ++i) { const double d = ((double)hu[i * M + c] - hv[i * M + c]) - s1 / n; s2 += d * d; }
            wmod[c] = (float)sqrt(s2 / n);
        }
        EXPECT(rc == PRE_OK && rel_err(hmod, wmod, M) <= 1e-5, "pre_std_axis0_f32 vs C two-pass std (<= 1e-5)");
        hipFree(dmod);
    }

    /* ---- a14: coverage counts against per-cell bounds (Joint/Burgers_Residuals_CP.py:298-300;
     * Active_Learning/Advection_AL_Marginal.py:169-198) */
    {
        enum { n = B * T, M = X * Y };
        static float hlo[M], hhi[M];
        float *dlo, *dhi;
        unsigned long long *dcount, hcount = 0, wcount = 0;
        uint32_t *drow, hrow[n], hrow2[n], wrow[n], wrow2[n];
        uint8_t *dins, hins[n], wins[n];
        for (int c = 0; c < M; ++c) { hlo[c] = 0.5f + 0.2f * (float)(c % 3) * 0.1f; hhi[c] = 1.45f - 0.01f * (float)(c % 7); }
        for (int c = 0; c < M; ++c) { hlo[c] = (c % 5 == 0) ? hu[2 * M + c] : hlo[c]; }         /* a bound that IS a sample: <= counts */
        CHECK_HIP(hipMalloc((void **)&dlo, sizeof hlo)); CHECK_HIP(hipMalloc((void **)&dhi, sizeof hhi));
        CHECK_HIP(hipMalloc((void **)&dcount, 8)); CHECK_HIP(hipMalloc((void **)&drow, sizeof hrow)); CHECK_HIP(hipMalloc((void **)&dins, sizeof hins));
        CHECK_HIP(hipMemcpy(dlo, hlo, sizeof hlo, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dhi, hhi, sizeof hhi, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemsetAsync(dcount, 0, 8, st)); CHECK_HIP(hipMemsetAsync(drow, 0, sizeof hrow, st)); CHECK_HIP(hipMemsetAsync(dins, 1, sizeof hins, st));
        int rc = pre_cov_count_f32(du, dlo, dhi, n, M, 0, dcount, st);
        int rc2 = pre_cov_rowcount_f32(du, dlo, dhi, n, M, 0, 0, drow, st);
        int rc3 = pre_cov_joint_f32(du, dlo, dhi, n, M, 0, dins, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(&hcount, dcount, 8, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(hrow, drow, sizeof hrow, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hins, dins, sizeof hins, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemsetAsync(drow, 0, sizeof hrow, st));
        int rc4 = pre_cov_rowcount_f32(du, dlo, dhi, n, M, 0, 1, drow, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hrow2, drow, sizeof hrow2, hipMemcpyDeviceToHost));
        int same = 1;
        for (int i = 0; i < n; ++i) {
            wrow[i] = wrow2[i] = 0;
            for (int c = 0; c < M; ++c) {
                const float y = hu[i * M + c];
                wrow[i] += (y >= hlo[c] && y <= hhi[c]);
                wrow2[i] += (y <= hlo[c] || y >= hhi[c]);
            }
            wins[i] = wrow[i] == M;
            wcount += wrow[i];
            same &= hrow[i] == wrow[i] && hrow2[i] == wrow2[i] && hins[i] == wins[i];
        }
        EXPECT(rc == PRE_OK && hcount == wcount && wcount > 0 && wcount < (unsigned long long)n * M, "pre_cov_count_f32: cells inside [lo, hi], exact");
        EXPECT(rc2 == PRE_OK && rc3 == PRE_OK && rc4 == PRE_OK && same, "pre_cov_rowcount_f32 (inside / outside) + pre_cov_joint_f32: per-sample counts and flags, exact");
        hipFree(dlo); hipFree(dhi); hipFree(dcount); hipFree(drow); hipFree(dins);
    }

    /* ---- a11, several score matrices in one launch (ABI v8): the field as 3 planes of [6, M] scores */
    {
        enum { P = B, n = T, M = X * Y, NK = 2 };
        int32_t ks[NK] = {n - 1, 1};
        float *dq;
        static float hq[P * NK * M], col[n];
        CHECK_HIP(hipMalloc((void **)&dq, sizeof hq));
        /* results as [plane][rank][cell] */
        int rc = pre_kth_axis0_planes_f32(du, (int64_t)n * M, M, P, n, M, ks, NK, dq, M, (int64_t)NK * M, st);
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost));
        int exact = 1;
        for (int p = 0; p < P; ++p) for (int c = 0; c < M; ++c) {
            for (int i = 0; i < n; ++i) col[i] = hu[(p * n + i) * M + c];
            qsort(col, n, sizeof(float), cmp_float);
            for (int j = 0; j < NK; ++j) exact &= (hq[(p * NK + j) * M + c] == col[ks[j]]);
        }
        EXPECT(rc == PRE_OK && exact, "pre_kth_axis0_planes_f32: 3 planes in one launch, bit-exact vs qsort per cell");
        EXPECT(pre_kth_axis0_planes_f32(du, M - 1, M, P, n, M, ks, NK, dq, M, (int64_t)NK * M, st) == PRE_E_RANGE, "planes closer than a row is long -> PRE_E_RANGE");
        EXPECT(pre_joint_score_pruned_max_segments() >= (64 * 1024 - 256) / 4, "pre_joint_score_pruned_max_segments() >= 16320 (64 KiB of LDS)");
        hipFree(dq);
    }

    /* ---- a9: periodic_bc_residual (Marginal/NS_Residuals_CP.py:468-478): opposite edges of every plane, times dx */
    {
        const float dx = 1.f / 64;
        float *de;
        static float he[B * T * Y], we[B * T * Y];
        CHECK_HIP(hipMalloc((void **)&de, sizeof he));
        int ok = 1;
        for (int wall = 0; wall < 4; ++wall) {
            const int L = wall < 2 ? Y : X;
            int rc = pre_edge_residual_f32(&fu, wall, dx, B, T, X, Y, de, st);
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(he, de, sizeof he, hipMemcpyDeviceToHost));
            for (int bt = 0; bt < B * T; ++bt) for (int i = 0; i < L; ++i) {
                const float *pl = hu + (size_t)bt * X * Y;
                const float first = wall < 2 ? pl[i] : pl[i * Y], last = wall < 2 ? pl[(X - 1) * Y + i] : pl[i * Y + Y - 1];
                we[bt * L + i] = ((wall == 0 || wall == 2) ? first - last : last - first) * dx;
            }
            ok &= rc == PRE_OK && memcmp(he, we, (size_t)B * T * L * sizeof(float)) == 0;
        }
        EXPECT(ok, "pre_edge_residual_f32: all four walls, bit-exact vs C");
        EXPECT(pre_edge_residual_f32(&fu, 4, dx, B, T, X, Y, de, st) == PRE_E_RANGE, "wall 4 -> PRE_E_RANGE");
        hipFree(de);
    }

    CHECK_HIP(hipStreamDestroy(st));
    CHECK_HIP(hipFree(du)); CHECK_HIP(hipFree(dv)); CHECK_HIP(hipFree(dp)); CHECK_HIP(hipFree(dout));
    printf(failures ? "%d check(s) FAILED\n" : "all checks passed\n", failures);
    return failures ? 1 : 0;
}
