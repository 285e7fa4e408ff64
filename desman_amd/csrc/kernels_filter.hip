// kernels_filter.hip -- row f3: the likelihood-ratio variant filter's per-variant step
// (desman/Variant_Filter.py:348-356).  One lane per variant position:
//   BLL[v] = -sum_b f_vb ln eta[A_v, b]
//   p_v    = argmin_{p in (0, upperP)} mixNLL(p) = -sum_b f_vb ln(p eta[A_v,b] + (1-p) eta[B_v,b])   (:38-41)
//   MLL[v] = mixNLL(p_v)
// The minimiser is the bounded Brent method the reference calls
// (scipy.optimize.minimize_scalar(method='bounded'), xatol = 1e-5, maxiter = 500; SciPy's
// _minimize_scalar_bounded, restated step for step -- same golden-section / parabolic
// decisions, same tolerances -- so p agrees with SciPy to rounding).
#include "dsm_device.h"
#include "dsm_host.h"

__device__ __forceinline__ double mix_nll(double p, const double *ea, const double *eb, const double *f)
{
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) acc += f[b] * (-log(p * ea[b] + (1.0 - p) * eb[b]));     // np.dot(f, -np.log(mix))
    return acc;
}

__device__ double fminbound_mix(double x1, double x2, const double *ea, const double *eb, const double *f,
                                double xatol, int maxfun)
{
    const double sqrt_eps = sqrt(2.2e-16);
    const double golden_mean = 0.5 * (3.0 - sqrt(5.0));
    double a = x1, b = x2;
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf;
    double fx = mix_nll(x, ea, eb, f);
    int num = 1;
    double fu, ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    double tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        bool golden = true;
        if (fabs(e) > tol1) {                         // try a parabolic step
            golden = false;
            double r = (xf - nfc) * (fx - ffulc);
            double q = (xf - fulc) * (fx - fnfc);
            double p = (xf - fulc) * q - (xf - nfc) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = rat;
            if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
                rat = (p + 0.0) / q;
                x = xf + rat;
                if (((x - a) < tol2) || ((b - x) < tol2)) {
                    const double d = xm - xf;
                    const double si = (d > 0.0 ? 1.0 : d < 0.0 ? -1.0 : 0.0) + (d == 0.0 ? 1.0 : 0.0);
                    rat = tol1 * si;
                }
            } else {
                golden = true;
            }
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        const double si = (rat > 0.0 ? 1.0 : rat < 0.0 ? -1.0 : 0.0) + (rat == 0.0 ? 1.0 : 0.0);
        x = xf + si * fmax(fabs(rat), tol1);
        fu = mix_nll(x, ea, eb, f);
        num += 1;
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
        } else {
            if (x < xf) a = x; else b = x;
            if ((fu <= fnfc) || (nfc == xf)) {
                fulc = nfc; ffulc = fnfc;
                nfc = x; fnfc = fu;
            } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
                fulc = x; ffulc = fu;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
        tol2 = 2.0 * tol1;
        if (num >= maxfun) break;
    }
    return xf;
}

__global__ __launch_bounds__(256) void lrt_kernel(const double *__restrict__ ffreq, const int32_t *__restrict__ maxA,
                                                  const int32_t *__restrict__ maxB, const double *__restrict__ eta,
                                                  double upperP, int optimise, int V, double *__restrict__ p,
                                                  double *__restrict__ MLL, double *__restrict__ BLL)
{
    __shared__ double es[16];
    if (threadIdx.x < 16) es[threadIdx.x] = eta[threadIdx.x];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    double f[4], ea[4], eb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { f[b] = ffreq[(size_t)v * 4 + b]; ea[b] = es[maxA[v] * 4 + b]; eb[b] = es[maxB[v] * 4 + b]; }
    double bll = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) bll += log(ea[b]) * f[b];                 // (log(eta[maxA,:]) * freq).sum(axis=1)
    BLL[v] = -bll;
    double pv = p[v];
    if (optimise) pv = fminbound_mix(0.0, upperP, ea, eb, f, 1.0e-5, 500);
    p[v] = pv;
    MLL[v] = mix_nll(pv, ea, eb, f);
}

extern "C" int dsm_lrt_step(int device, const double *ffreq, const int32_t *maxA, const int32_t *maxB,
                            const double *eta, double upperP, int optimise, int V, double *p_inout, double *MLL,
                            double *BLL)
{
    if (!ffreq || !maxA || !maxB || !eta || !p_inout || !MLL || !BLL || V < 0) { dsm_set_error("lrt_step: bad arguments"); return DSM_ERR_ARG; }
    if (V == 0) return DSM_OK;
    if (dsm_device_count() <= 0) { dsm_set_error("no HIP device visible"); return DSM_ERR_NODEVICE; }
    HIP_TRY(hipSetDevice(device));
    double *d_f = nullptr, *d_eta = nullptr, *d_p = nullptr, *d_m = nullptr, *d_b = nullptr;
    int32_t *d_a = nullptr, *d_bb = nullptr;
    const size_t n = (size_t)V;
    hipError_t e = hipSuccess;
    auto M = [&](void **ptr, size_t bytes) { if (e == hipSuccess) e = hipMalloc(ptr, bytes); };
    M((void **)&d_f, n * 4 * sizeof(double)); M((void **)&d_eta, 16 * sizeof(double)); M((void **)&d_p, n * sizeof(double));
    M((void **)&d_m, n * sizeof(double)); M((void **)&d_b, n * sizeof(double)); M((void **)&d_a, n * sizeof(int32_t));
    M((void **)&d_bb, n * sizeof(int32_t));
    int rc = DSM_OK;
    if (e != hipSuccess) { dsm_set_error("lrt_step: hipMalloc failed: %s", hipGetErrorString(e)); rc = DSM_ERR_NOMEM; }
    if (rc == DSM_OK) {
        e = hipMemcpy(d_f, ffreq, n * 4 * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_eta, eta, 16 * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_p, p_inout, n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_a, maxA, n * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_bb, maxB, n * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(lrt_kernel, dim3((V + 255) / 256), dim3(256), 0, 0, d_f, d_a, d_bb, d_eta, upperP, optimise, V,
                               d_p, d_m, d_b);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(p_inout, d_p, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(MLL, d_m, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(BLL, d_b, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { dsm_set_error("lrt_step: %s", hipGetErrorString(e)); rc = DSM_ERR_HIP; }
    }
    (void)hipFree(d_f); (void)hipFree(d_eta); (void)hipFree(d_p); (void)hipFree(d_m); (void)hipFree(d_b);
    (void)hipFree(d_a); (void)hipFree(d_bb);
    return rc;
}
