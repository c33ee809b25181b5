// DESC refinement tail: reweighted Lie-algebraic averaging.
//
// Reference text reproduced:
//   Algorithms/DESC.m:265-313      RR = R_ij', Q = R2Q(R_init), QQ = R2Q(RR); weights 1/S^0.75 clipped to
//                                  [1e-4, 1e4]; loop: Weighted_LAA, residuals, RSVec = (1-lam) Res + lam S,
//                                  quantile truncation, stop at score <= 1e-3 or 100 iterations; q2R
//   Utils/Weighted_LAA.m:4-51      per-edge residual quaternion and log map, weighted least squares
//                                  (W*A) \ (W*B), exp map, quaternion update
//   Utils/Build_Amatrix.m:6-13     incidence matrix with node 1 grounded
//   Utils/R2Q.m:7-14, Utils/q2R.m:1-23
// MATLAB solves the m x (n-1) weighted least-squares problem by sparse QR.  Here the normal
// equations (a grounded graph Laplacian with edge weights w^2, three right-hand sides) are solved
// on the device by Jacobi-preconditioned conjugate gradients in f64 with device-resident scalars
// (no host round trip inside the CG loop); per-edge and per-node maps are plain HIP kernels.
// `quantile` (Hazen positions, DESC.m:276,301) is evaluated on the host from a copy of RSVec
// (one O(m) selection per refinement step).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "device_utils.h"

namespace desc {
namespace {

struct Quat { double a, x, y, z; };

// ---- per-edge / per-node maps ------------------------------------------------------------
// R2Q.m:9-12 for a column-major 3x3 block (optionally transposed)
__device__ __forceinline__ Quat r2q(const double* R, bool transpose) {
    const double r11 = R[0], r22 = R[4], r33 = R[8];
    double r32 = R[5], r23 = R[7], r13 = R[6], r31 = R[2], r21 = R[1], r12 = R[3];      // (r,c) at r + 3c
    if (transpose) { double t; t = r32; r32 = r23; r23 = t; t = r13; r13 = r31; r31 = t; t = r21; r21 = r12; r12 = t; }
    Quat q;
    q.a = (r11 + r22 + r33 - 1.0) / 2.0; q.x = (r32 - r23) / 2.0; q.y = (r13 - r31) / 2.0; q.z = (r21 - r12) / 2.0;
    q.a = sqrt((q.a + 1.0) / 2.0);
    q.x = (q.x / q.a) / 2.0; q.y = (q.y / q.a) / 2.0; q.z = (q.z / q.a) / 2.0;
    return q;
}
__global__ void k_r2q(const double* R, Quat* Q, int64_t count, int transpose) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (int64_t)gridDim.x * blockDim.x) Q[t] = r2q(R + 9 * t, transpose);
}
// q2R.m
__global__ void k_q2r(const Quat* Q, double* R, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const Quat q = Q[t];
        double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double c2 = q.a;
        if (fabs(fabs(c2) - 1.0) > 1e-12) {
            const double s2 = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
            const double s = 2.0 * s2 * c2, c = 2.0 * c2 * c2 - 1.0, cc = 1.0 - c;
            const double n1 = q.x / s2, n2 = q.y / s2, n3 = q.z / s2;
            const double n12 = n1 * n2 * cc, n23 = n2 * n3 * cc, n31 = n3 * n1 * cc, n1s = n1 * s, n2s = n2 * s, n3s = n3 * s;
            M[0] = c + n1 * n1 * cc; M[3] = n12 - n3s;        M[6] = n31 + n2s;        // column-major: (r,c) at r + 3c
            M[1] = n12 + n3s;        M[4] = c + n2 * n2 * cc; M[7] = n23 - n1s;
            M[2] = n31 - n2s;        M[5] = n23 + n1s;        M[8] = c + n3 * n3 * cc;
        }
        for (int k = 0; k < 9; ++k) R[9 * t + k] = M[k];
    }
}

// Weighted_LAA.m:9-37: residual quaternion w = -(conj(Qj) (QQ Qi)), B = log map (3 per edge)
__global__ void k_edge_log(const Quat* Q, const Quat* QQ, const int32_t* ii, const int32_t* jj, double* B, int64_t m) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
        const Quat qq = QQ[e], qi = Q[ii[e]], qj = Q[jj[e]];
        // w = QQ * Qi
        const double w1 = qq.a * qi.a - (qq.x * qi.x + qq.y * qi.y + qq.z * qi.z);
        const double w2 = qq.a * qi.x + qi.a * qq.x + (qq.y * qi.z - qq.z * qi.y);
        const double w3 = qq.a * qi.y + qi.a * qq.y + (qq.z * qi.x - qq.x * qi.z);
        const double w4 = qq.a * qi.z + qi.a * qq.z + (qq.x * qi.y - qq.y * qi.x);
        // w = inv(Qj) * w  (as written in the reference: the negated product, the same rotation)
        double v1 = -qj.a * w1 - (qj.x * w2 + qj.y * w3 + qj.z * w4);
        const double v2 = -qj.a * w2 + w1 * qj.x + (qj.y * w4 - qj.z * w3);
        const double v3 = -qj.a * w3 + w1 * qj.y + (qj.z * w2 - qj.x * w4);
        const double v4 = -qj.a * w4 + w1 * qj.z + (qj.x * w3 - qj.y * w2);
        const double s2 = sqrt(v2 * v2 + v3 * v3 + v4 * v4);
        v1 = 2.0 * atan2(s2, v1);
        if (v1 < -M_PI) v1 += 2.0 * M_PI;
        if (v1 >= M_PI) v1 -= 2.0 * M_PI;
        const double f = v1 / s2;
        double b1 = v2 * f, b2 = v3 * f, b3 = v4 * f;
        if (isnan(b1)) b1 = 0.0;                                                     // :35
        if (isnan(b2)) b2 = 0.0;
        if (isnan(b3)) b3 = 0.0;
        B[3 * e] = b1; B[3 * e + 1] = b2; B[3 * e + 2] = b3;
    }
}

// per CSR slot t of node v: neighbour adj[t], edge eid[t], sgn[t] = +1 if v is the edge's j (A has
// +1 in column j, -1 in column i).  16 lanes per node row.
// rhs_v = sum_t sgn * w_e^2 * B_e ;  diag_v = sum_t w_e^2          (normal equations A'W^2A x = A'W^2 B)
__global__ __launch_bounds__(256) void k_rhs(const int32_t* rowptr, const int32_t* eid, const int8_t* sgn, const double* wts,
                                             const double* B, double* rhs, double* diag, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int vb = row0 - (row0 % 4); vb < n; vb += nrows) {
        const int v = vb + (row0 % 4);
        double a0 = 0, a1 = 0, a2 = 0, dg = 0;
        if (v < n)
            for (int t = rowptr[v] + l16; t < rowptr[v + 1]; t += 16) {
                const int e = eid[t];
                const double w2 = wts[e] * wts[e], sg = (double)sgn[t];
                a0 += sg * w2 * B[3 * (int64_t)e]; a1 += sg * w2 * B[3 * (int64_t)e + 1]; a2 += sg * w2 * B[3 * (int64_t)e + 2];
                dg += w2;
            }
        a0 = group16_sum(a0); a1 = group16_sum(a1); a2 = group16_sum(a2); dg = group16_sum(dg);
        if (v < n && l16 == 0) { rhs[3 * v] = a0; rhs[3 * v + 1] = a1; rhs[3 * v + 2] = a2; diag[v] = dg; }
    }
}
// q_v = sum_t w_e^2 (p_v - p_u) for v != 0 (node 0 = MATLAB node 1 is grounded: its unknown is fixed at 0)
__global__ __launch_bounds__(256) void k_lap(const int32_t* rowptr, const int32_t* adj, const int32_t* eid, const double* wts,
                                             const double* p, double* q, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int vb = row0 - (row0 % 4); vb < n; vb += nrows) {
        const int v = vb + (row0 % 4);
        double a0 = 0, a1 = 0, a2 = 0;
        if (v < n && v > 0) {
            const double p0 = p[3 * v], p1 = p[3 * v + 1], p2 = p[3 * v + 2];
            for (int t = rowptr[v] + l16; t < rowptr[v + 1]; t += 16) {
                const int u = adj[t];
                const double w2 = wts[eid[t]] * wts[eid[t]];
                a0 += w2 * (p0 - p[3 * u]); a1 += w2 * (p1 - p[3 * u + 1]); a2 += w2 * (p2 - p[3 * u + 2]);
            }
        }
        a0 = group16_sum(a0); a1 = group16_sum(a1); a2 = group16_sum(a2);
        if (v < n && l16 == 0) { q[3 * v] = a0; q[3 * v + 1] = a1; q[3 * v + 2] = a2; }
    }
}

// ---- CG with device-resident scalars (3 right-hand sides share the operator) --------------
struct CgScal { double rz[3], rz_new[3], pq[3], bnorm[3], rnorm[3]; };

// one workgroup: column-wise dot products of two n x 3 arrays (fixed order -> reproducible)
__global__ __launch_bounds__(256) void k_dot3(const double* a, const double* b, int n, double* out3) {
    __shared__ double sh[3][256];
    double s[3] = {0, 0, 0};
    for (int v = threadIdx.x; v < n; v += 256) for (int c = 0; c < 3; ++c) s[c] += a[3 * v + c] * b[3 * v + c];
    for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x < 3) out3[threadIdx.x] = sh[threadIdx.x][0];
}
// x = 0, r = rhs (node 0 zeroed), z = r/diag, p = z
__global__ void k_cg_init(const double* rhs, const double* diag, double* x, double* r, double* z, double* p, int n) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) {
            const double rv = v > 0 ? rhs[3 * v + c] : 0.0;
            const double zv = (v > 0 && diag[v] > 0) ? rv / diag[v] : 0.0;
            x[3 * v + c] = 0.0; r[3 * v + c] = rv; z[3 * v + c] = zv; p[3 * v + c] = zv;
        }
}
// alpha = rz/pq ; x += alpha p ; r -= alpha q ; z = r/diag
__global__ void k_cg_update(const CgScal* sc, const double* diag, const double* p, const double* q, double* x, double* r, double* z, int n) {
    double al[3];
    for (int c = 0; c < 3; ++c) al[c] = sc->pq[c] > 0 ? sc->rz[c] / sc->pq[c] : 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) {
            const double xv = x[3 * v + c] + al[c] * p[3 * v + c];
            const double rv = r[3 * v + c] - al[c] * q[3 * v + c];
            x[3 * v + c] = xv; r[3 * v + c] = rv;
            z[3 * v + c] = (v > 0 && diag[v] > 0) ? rv / diag[v] : 0.0;
        }
}
// beta = rz_new/rz ; p = z + beta p ; rz = rz_new
__global__ void k_cg_dir(CgScal* sc, const double* z, double* p, int n) {
    double be[3];
    for (int c = 0; c < 3; ++c) be[c] = sc->rz[c] > 0 ? sc->rz_new[c] / sc->rz[c] : 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) p[3 * v + c] = z[3 * v + c] + be[c] * p[3 * v + c];
}
__global__ void k_cg_roll(CgScal* sc) { if (threadIdx.x < 3) sc->rz[threadIdx.x] = sc->rz_new[threadIdx.x]; }

// Weighted_LAA.m:40-50: score, exp map, Q <- Q * W ; x holds the tangent solution (row 0 = 0)
__global__ __launch_bounds__(256) void k_node_update(const double* x, Quat* Q, double* Wv /* n x 3: vector part of the quaternion W */,
                                                     int n, double* score_partial) {
    double sc = 0.0;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256) {
        const double t1 = x[3 * v], t2 = x[3 * v + 1], t3 = x[3 * v + 2];
        const double theta = sqrt(t1 * t1 + t2 * t2 + t3 * t3);
        if (v > 0) sc += theta;                                                       // :40 (rows 2:end)
        double wa = cos(theta / 2.0);
        const double f = sin(theta / 2.0) / theta;
        double wx = t1 * f, wy = t2 * f, wz = t3 * f;
        if (isnan(wa)) wa = 0.0;                                                      // :46
        if (isnan(wx)) wx = 0.0;
        if (isnan(wy)) wy = 0.0;
        if (isnan(wz)) wz = 0.0;
        Wv[3 * v] = wx; Wv[3 * v + 1] = wy; Wv[3 * v + 2] = wz;
        const Quat q = Q[v];
        Quat o;
        o.a = q.a * wa - (q.x * wx + q.y * wy + q.z * wz);
        o.x = q.a * wx + wa * q.x + (q.y * wz - q.z * wy);
        o.y = q.a * wy + wa * q.y + (q.z * wx - q.x * wz);
        o.z = q.a * wz + wa * q.z + (q.x * wy - q.y * wx);
        Q[v] = o;
    }
    sc = group_sum<64>(sc);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) score_partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// DESC.m:289-291: E = A*W(2:end,2:4) - B, ResVec = |E|/pi, RSVec = (1-lam) ResVec + lam S
__global__ void k_rsvec(const double* Wv, const double* B, const int32_t* ii, const int32_t* jj, const double* S, double* RS,
                        int64_t m, double lam) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = ii[e], j = jj[e];
        double s = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double ax = (j > 0 ? Wv[3 * j + c] : 0.0) - (i > 0 ? Wv[3 * i + c] : 0.0);
            const double d = ax - B[3 * e + c];
            s += d * d;
        }
        RS[e] = (1.0 - lam) * (sqrt(s) / M_PI) + lam * S[e];
    }
}
// DESC.m:298-303
__global__ void k_weights(const double* RS, double* wts, int64_t m, double thresh, double wmax, double wmin) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
        double w = 1.0 / pow(RS[e], 0.75);
        if (w > wmax) w = wmax;
        if (RS[e] > thresh) w = wmin;
        wts[e] = w;
    }
}

// ---- MATLAB's quantile on the device: a 4096-bin histogram over [lo, hi] locates the two order statistics the Hazen
// interpolation needs, the values of their bins (a few hundred of m) are collected and ordered on the host.
constexpr int QBINS = 4096;
__device__ __forceinline__ int qbin(double x, double lo, double scale) {
    const int b = (int)((x - lo) * scale);
    return b < 0 ? 0 : (b >= QBINS ? QBINS - 1 : b);
}
__global__ __launch_bounds__(256) void k_minmax(const double* x, int64_t m, double* out /* [grid][2] */) {
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256) { lo = fmin(lo, x[e]); hi = fmax(hi, x[e]); }
    __shared__ double sl[256], sh[256];
    sl[threadIdx.x] = lo; sh[threadIdx.x] = hi;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) { sl[threadIdx.x] = fmin(sl[threadIdx.x], sl[threadIdx.x + s2]); sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s2]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = sl[0]; out[2 * blockIdx.x + 1] = sh[0]; }
}
__global__ __launch_bounds__(256) void k_qhist(const double* x, int64_t m, double lo, double scale, unsigned* hist) {
    __shared__ unsigned h[QBINS];
    for (int t = threadIdx.x; t < QBINS; t += 256) h[t] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256) atomicAdd(&h[qbin(x[e], lo, scale)], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < QBINS; t += 256) if (h[t]) atomicAdd(&hist[t], h[t]);
}
__global__ __launch_bounds__(256) void k_qcollect(const double* x, int64_t m, double lo, double scale, int b0, int b1, double* out, unsigned* count, unsigned cap) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256) {
        const int b = qbin(x[e], lo, scale);
        if (b == b0 || b == b1) { const unsigned p = atomicAdd(count, 1u); if (p < cap) out[p] = x[e]; }
    }
}

// MATLAB quantile(x, p): Hazen plotting positions (k-0.5)/n, linear interpolation, clamped
double matlab_quantile(hvec<double>& x, double p) {
    const size_t n = x.size();
    if (n == 0) return NAN;
    double pos = p * (double)n + 0.5;                    // 1-based fractional index
    if (pos <= 1.0) return *std::min_element(x.begin(), x.end());
    if (pos >= (double)n) return *std::max_element(x.begin(), x.end());
    const size_t lo = (size_t)std::floor(pos) - 1;       // 0-based
    const double fr = pos - std::floor(pos);
    std::nth_element(x.begin(), x.begin() + lo, x.end());
    const double a = x[lo];
    const double b = *std::min_element(x.begin() + lo + 1, x.end());
    return a + fr * (b - a);
}

// quantile(x, p) of a device vector, MATLAB's definition (the same order statistics as matlab_quantile above);
// scratch: d_mm [64][2] doubles, d_hist QBINS unsigned + 1 counter, d_cand cap doubles
int device_quantile(const double* d_x, int64_t m, double p, double* d_mm, unsigned* d_hist, double* d_cand, unsigned cap, double* result) {
    if (m == 0) { *result = NAN; return DESC_OK; }
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (m + 255) / 256));
    hipLaunchKernelGGL(k_minmax, dim3(64), dim3(256), 0, 0, d_x, m, d_mm);
    double mm[128];
    DESC_HIP(hipMemcpy(mm, d_mm, sizeof mm, hipMemcpyDeviceToHost));
    double lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < 64; ++b) { lo = std::min(lo, mm[2 * b]); hi = std::max(hi, mm[2 * b + 1]); }
    const double pos = p * (double)m + 0.5;                 // 1-based fractional index
    if (pos <= 1.0) { *result = lo; return DESC_OK; }
    if (pos >= (double)m) { *result = hi; return DESC_OK; }
    if (!(hi > lo)) { *result = lo; return DESC_OK; }
    const int64_t k0 = (int64_t)std::floor(pos) - 1;        // 0-based rank of the lower order statistic; the upper one is k0 + 1
    const double fr = pos - std::floor(pos);
    const double scale = (double)QBINS / (hi - lo) * (1.0 - 1e-12);
    DESC_HIP(hipMemset(d_hist, 0, sizeof(unsigned) * (QBINS + 1)));
    hipLaunchKernelGGL(k_qhist, dim3(g), dim3(256), 0, 0, d_x, m, lo, scale, d_hist);
    hvec<unsigned> hist(QBINS);
    DESC_HIP(hipMemcpy(hist.data(), d_hist, sizeof(unsigned) * QBINS, hipMemcpyDeviceToHost));
    int64_t acc = 0; int b0 = -1, b1 = -1; int64_t base0 = 0;
    for (int b = 0; b < QBINS; ++b) {
        if (b0 < 0 && acc + hist[b] > (uint64_t)k0) { b0 = b; base0 = acc; }
        if (b0 >= 0 && acc + hist[b] > (uint64_t)(k0 + 1)) { b1 = b; break; }
        acc += hist[b];
    }
    if (b0 < 0 || b1 < 0) return fail(DESC_ERR_STATE, "quantile histogram inconsistent");
    const uint64_t need = (uint64_t)hist[b0] + (b1 != b0 ? hist[b1] : 0);
    if (need > cap) {                                       // a bin too full to collect (heavily tied data): exact host path
        hvec<double> all((size_t)m);
        DESC_HIP(hipMemcpy(all.data(), d_x, sizeof(double) * m, hipMemcpyDeviceToHost));
        *result = matlab_quantile(all, p);
        return DESC_OK;
    }
    hipLaunchKernelGGL(k_qcollect, dim3(g), dim3(256), 0, 0, d_x, m, lo, scale, b0, b1, d_cand, d_hist + QBINS, cap);
    hvec<double> cand((size_t)need);
    DESC_HIP(hipMemcpy(cand.data(), d_cand, sizeof(double) * need, hipMemcpyDeviceToHost));
    std::sort(cand.begin(), cand.end());
    // cand = bin b0 (ranks base0 ...) followed, if different, by bin b1 (which starts at rank >= k0 + 1)
    const double a = cand[(size_t)(k0 - base0)];
    const double bnext = (b1 == b0) ? cand[(size_t)(k0 + 1 - base0)] : cand[(size_t)hist[b0] + 0 + (size_t)0];
    *result = a + fr * (bnext - a);
    return DESC_OK;
}

struct DevR {
    hvec<void*> p;
    ~DevR() { for (void* q : p) dev_free(q); }
    template <class T> int alloc(T** out, size_t count) {
        void* q = nullptr;
        DESC_HIP(dev_alloc(&q, sizeof(T) * (count ? count : 1)));
        p.push_back(q); *out = (T*)q;
        return DESC_OK;
    }
};

}  // namespace
}  // namespace desc

using namespace desc;

// Build_Amatrix.m:10: -1 at the smaller endpoint i, +1 at j -- per CSR slot
__global__ void k_incidence_sign(const int32_t* rowptr, const int32_t* adj, int8_t* sgn, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int v = row0; v < n; v += nrows)
        for (int t = rowptr[v] + l16; t < rowptr[v + 1]; t += 16) sgn[t] = v < adj[t] ? -1 : +1;
}

extern "C" int desc_refine_run(const desc_problem* prob, const double* s_vec, const double* R_init, double stop_threshold,
                               int32_t max_iters, int32_t device, double* R_out, desc_refine_info* info) {
    if (!prob || !s_vec || !R_init || !R_out) return fail(DESC_ERR_INVALID, "NULL argument");
    if (prob->n == 0) return validate_problem(prob, true);
    auto t0 = std::chrono::steady_clock::now();
    desc_device_problem* dp = nullptr;
    int rc = desc_problem_upload(prob, device, &dp);
    if (rc) return rc;
    rc = desc_refine_run_dev(dp, s_vec, R_init, stop_threshold, max_iters, R_out, info);
    desc_problem_free(dp);
    if (!rc && info) info->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int desc_refine_run_dev(const desc_device_problem* dp, const double* s_vec, const double* R_init, double stop_threshold,
                                   int32_t max_iters, double* R_out, desc_refine_info* info) {
    return no_throw("desc_refine_run_dev", [&]() -> int {
    if (!dp || !s_vec || !R_init || !R_out) return fail(DESC_ERR_INVALID, "NULL argument");
    int rc = DESC_OK;
    const int64_t n = dp->n, m = dp->m;
    if (n == 0) return DESC_OK;
    DESC_HIP(hipSetDevice(dp->device));
    auto t0 = std::chrono::steady_clock::now();
    if (stop_threshold <= 0) stop_threshold = 1e-3;      // DESC.m:272
    if (max_iters <= 0) max_iters = 100;
    const double weight_max = 1e4, weight_min = 1e-4;    // DESC.m:280-281

    // initial weights (DESC.m:274-282): quantile(S_vec, 1) = max -> nothing is truncated yet; evaluated on
    // the device by the same kernel as the re-weighting steps
    double thresh0 = -INFINITY;
    for (int64_t e = 0; e < m; ++e) thresh0 = std::max(thresh0, s_vec[e]);
    DevR D;
    const int32_t *d_rowptr = dp->d_rowptr, *d_adj = dp->d_adj, *d_eid = dp->d_adj_eid, *d_ii = dp->d_ii, *d_jj = dp->d_jj;
    const double* d_rij = dp->d_rij;
    int8_t* d_sgn;
    double *d_Rinit, *d_w, *d_S, *d_B, *d_RS, *d_rhs, *d_diag, *d_x, *d_r, *d_z, *d_p, *d_q, *d_Wv, *d_score, *d_Rout;
    Quat *d_Q, *d_QQ; CgScal* d_sc;
    const int sgrid = 64;
    if ((rc = D.alloc(&d_sgn, 2 * m)) || (rc = D.alloc(&d_Rinit, 9 * n)) ||
        (rc = D.alloc(&d_w, m)) || (rc = D.alloc(&d_S, m)) || (rc = D.alloc(&d_B, 3 * m)) || (rc = D.alloc(&d_RS, m)) ||
        (rc = D.alloc(&d_rhs, 3 * n)) || (rc = D.alloc(&d_diag, n)) || (rc = D.alloc(&d_x, 3 * n)) || (rc = D.alloc(&d_r, 3 * n)) ||
        (rc = D.alloc(&d_z, 3 * n)) || (rc = D.alloc(&d_p, 3 * n)) || (rc = D.alloc(&d_q, 3 * n)) || (rc = D.alloc(&d_Wv, 3 * n)) ||
        (rc = D.alloc(&d_score, sgrid)) || (rc = D.alloc(&d_Rout, 9 * n)) || (rc = D.alloc(&d_Q, n)) || (rc = D.alloc(&d_QQ, m)) ||
        (rc = D.alloc(&d_sc, 1))) return rc;
    DESC_HIP(hipMemcpy(d_Rinit, R_init, sizeof(double) * 9 * n, hipMemcpyHostToDevice));
    if (m) {
        DESC_HIP(hipMemcpy(d_S, s_vec, sizeof(double) * m, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_incidence_sign, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (n * 16 + 255) / 256))), dim3(256), 0, 0,
                           d_rowptr, d_adj, d_sgn, (int)n);
    }
    const int egrid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (m + 255) / 256));
    if (m) hipLaunchKernelGGL(k_weights, dim3(egrid), dim3(256), 0, 0, d_S, d_w, m, thresh0, weight_max, weight_min);
    const int ngrid = (int)std::max<int64_t>(1, std::min<int64_t>(512, (n + 255) / 256));
    const int rgrid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n * 16 + 255) / 256));
    hipLaunchKernelGGL(k_r2q, dim3(ngrid), dim3(256), 0, 0, d_Rinit, d_Q, n, 0);            // Q = R2Q(R_init)        (DESC.m:270)
    if (m) hipLaunchKernelGGL(k_r2q, dim3(egrid), dim3(256), 0, 0, d_rij, d_QQ, m, 1);      // QQ = R2Q(permute(RijMat)) (:265,271)

    double score = INFINITY, quant_ratio = 1.0;
    const double quant_ratio_min = 0.8;
    int Iteration = 1, cg_total = 0, cg_unconverged = 0;
    double cg_worst = 0.0;
    hvec<double> part(sgrid);
    constexpr unsigned QCAP = 1u << 20;
    double *d_mm, *d_cand; unsigned* d_qh;
    if ((rc = D.alloc(&d_mm, 128)) || (rc = D.alloc(&d_cand, QCAP)) || (rc = D.alloc(&d_qh, QBINS + 1))) return rc;
    CgScal hs;
    while (score > stop_threshold && Iteration < max_iters) {                               // DESC.m:287
        const double lam = 1.0 / (Iteration + 1);
        // ---- Weighted_LAA
        if (m) hipLaunchKernelGGL(k_edge_log, dim3(egrid), dim3(256), 0, 0, d_Q, d_QQ, d_ii, d_jj, d_B, m);
        hipLaunchKernelGGL(k_rhs, dim3(rgrid), dim3(256), 0, 0, d_rowptr, d_eid, d_sgn, d_w, d_B, d_rhs, d_diag, (int)n);
        hipLaunchKernelGGL(k_cg_init, dim3(ngrid), dim3(256), 0, 0, d_rhs, d_diag, d_x, d_r, d_z, d_p, (int)n);
        hipLaunchKernelGGL(k_dot3, dim3(1), dim3(256), 0, 0, d_r, d_z, (int)n, &d_sc->rz[0]);
        hipLaunchKernelGGL(k_dot3, dim3(1), dim3(256), 0, 0, d_r, d_r, (int)n, &d_sc->bnorm[0]);
        const int cg_max = (int)std::min<int64_t>(20000, 20 * n + 200);
        int k = 0;
        for (k = 1; k <= cg_max; ++k) {
            hipLaunchKernelGGL(k_lap, dim3(rgrid), dim3(256), 0, 0, d_rowptr, d_adj, d_eid, d_w, d_p, d_q, (int)n);
            hipLaunchKernelGGL(k_dot3, dim3(1), dim3(256), 0, 0, d_p, d_q, (int)n, &d_sc->pq[0]);
            hipLaunchKernelGGL(k_cg_update, dim3(ngrid), dim3(256), 0, 0, d_sc, d_diag, d_p, d_q, d_x, d_r, d_z, (int)n);
            hipLaunchKernelGGL(k_dot3, dim3(1), dim3(256), 0, 0, d_r, d_z, (int)n, &d_sc->rz_new[0]);
            hipLaunchKernelGGL(k_cg_dir, dim3(ngrid), dim3(256), 0, 0, d_sc, d_z, d_p, (int)n);
            hipLaunchKernelGGL(k_cg_roll, dim3(1), dim3(64), 0, 0, d_sc);
            if (k % 25 == 0 || k == cg_max) {                                               // convergence probe
                hipLaunchKernelGGL(k_dot3, dim3(1), dim3(256), 0, 0, d_r, d_r, (int)n, &d_sc->rnorm[0]);
                DESC_HIP(hipMemcpy(&hs, d_sc, sizeof hs, hipMemcpyDeviceToHost));
                bool done = true;
                for (int c = 0; c < 3; ++c) if (hs.rnorm[c] > 1e-26 * hs.bnorm[c] && hs.rnorm[c] > 1e-300) done = false;   // |r| <= 1e-13 |b|
                if (done || k == cg_max) {
                    for (int c = 0; c < 3; ++c) if (hs.bnorm[c] > 0) cg_worst = std::max(cg_worst, std::sqrt(hs.rnorm[c] / hs.bnorm[c]));
                    if (!done) ++cg_unconverged;
                    break;
                }
            }
        }
        cg_total += std::min(k, cg_max);
        hipLaunchKernelGGL(k_node_update, dim3(sgrid), dim3(256), 0, 0, d_x, d_Q, d_Wv, (int)n, d_score);
        DESC_HIP(hipMemcpy(part.data(), d_score, sizeof(double) * sgrid, hipMemcpyDeviceToHost));
        score = 0.0; for (double v : part) score += v;
        score /= (double)n;                                                                 // Weighted_LAA.m:40
        // ---- residuals and new weights (DESC.m:289-303)
        if (m) {
            hipLaunchKernelGGL(k_rsvec, dim3(egrid), dim3(256), 0, 0, d_Wv, d_B, d_ii, d_jj, d_S, d_RS, m, lam);
            quant_ratio = std::max(quant_ratio_min, quant_ratio - 0.05);
            double thresh = 0.0;                                                            // quantile(RSVec, quant_ratio)  (DESC.m:299)
            if ((rc = device_quantile(d_RS, m, quant_ratio, d_mm, d_qh, d_cand, QCAP, &thresh))) return rc;
            hipLaunchKernelGGL(k_weights, dim3(egrid), dim3(256), 0, 0, d_RS, d_w, m, thresh, weight_max, weight_min);
        }
        DESC_HIP(hipGetLastError());
        if (info && info->verbose) printf("Iter %d: ||\xce\x94R||= %f\n", Iteration, score);                 // DESC.m:305
        ++Iteration;
    }
    if (cg_unconverged)
        fprintf(stderr, "[desc_amd] warning: %d of %d Weighted_LAA solves stopped at the PCG iteration cap (relative residual up to %.3e)\n",
                cg_unconverged, Iteration - 1, cg_worst);
    hipLaunchKernelGGL(k_q2r, dim3(ngrid), dim3(256), 0, 0, d_Q, d_Rout, n);                // DESC.m:309-312
    DESC_HIP(hipDeviceSynchronize());
    DESC_HIP(hipMemcpy(R_out, d_Rout, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
    if (info) {
        info->iters = Iteration - 1; info->score = score; info->cg_iters = cg_total;
        info->cg_unconverged = cg_unconverged; info->cg_residual = cg_worst;
        info->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return DESC_OK;
    });
}
