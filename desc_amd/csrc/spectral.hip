// Spectral / GCW synchronisation: top-3 eigenvectors of the 3n x 3n block
// connection matrix by block subspace iteration on the device, then per-node projection
// onto SO(3).
//
// Reference text reproduced:
//   Algorithms/Spectral.m:18-46   Rij_blk (blocks R_ij / R_ij'), eigs(.,3,'la'), sign fix, SVD projection
//   Utils/GCW.m:9-36              the same with weights 1/(s^1.5+1e-8), row-normalised (:20-21)
// The reference builds a dense 3n x 3n matrix (1.8 GB at n = 5000) and calls eigs (Krylov-
// Schur).  Here the matrix is block-CSR with 2m blocks of 72 B (each edge in both endpoint
// rows, transposed in the larger endpoint's row); GCW's row-normalised D^-1 A is iterated in
// its symmetric similar form D^-1/2 A D^-1/2 (eigenvectors mapped back by D^-1/2 and
// re-normalised to unit 2-norm, as eigs returns them).  Block size 6 (3 wanted + 3 guard
// vectors), Chebyshev-filtered subspace iteration (the filter damps [-sigma, smallest Ritz value]),
// Rayleigh-Ritz in every outer step (b x b work on the host), explicit-residual stop.
// HBM-bound: 144 B and 54*b/3 flops per block and step; MFMA is not used (3x3 blocks, f64).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "device_utils.h"

namespace desc {
namespace {

constexpr int BW = 6;    // block width of the subspace iteration

// y[v] = alpha * sum_t blocks[t] * x[adj[t]] + s1 * x[v] + s2 * z[v];  x, y, z: (3n x BW) row-major
// (z may alias y: every element is read before it is written by the same lane).
// The 2m blocks are stored component-major (blocks[q * nslots + t], q = r + 3k): the lanes of a row read consecutive
// doubles per component (coalesced) instead of nine 8-byte picks at a 72-byte stride; the operand rows of x (144 B,
// L2-resident) come in as nine 16-byte loads.
// TPR threads share one node row: the whole 256-thread workgroup for rows of hundreds of slots (C2, C4, C5: the round-2 kernel gave
// a row to 16 lanes, i.e. 62 dependent load->FMA rounds per lane and 1.2 waves per SIMD on the whole chip at n = 5000 -- 0.35 ms per
// product at C4 for 0.36 GB of matrix), one wave for short rows.  Two slots are in flight per lane; partial sums are combined in a
// fixed order (DPP butterfly, then the four waves in index order): bitwise reproducible.
template <int TPR>
__global__ __launch_bounds__(256) void k_bsr_spmm(const int32_t* rowptr, const int32_t* adj, const double* blocks, int64_t nslots, const double* x,
                                                  const double* z, double* y, int n, double alpha, double s1, double s2) {
    static_assert(TPR == 64 || TPR == 256, "one wave or one workgroup per row");
    constexpr int RPB = 256 / TPR;
    __shared__ double sh[4][3 * BW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tr = threadIdx.x % TPR;
    for (int v0 = blockIdx.x * RPB; v0 < n; v0 += gridDim.x * RPB) {
        const int v = v0 + threadIdx.x / TPR;
        double acc[3][BW];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < BW; ++c) acc[r][c] = 0.0;
        if (v < n) {
            const int r1 = rowptr[v + 1];
            for (int t = rowptr[v] + tr; t < r1; t += 2 * TPR) {
                const bool two = t + TPR < r1;
                const int t2 = two ? t + TPR : t;
                double B[2][9];                                      // column-major 3x3: B(r,k) = B[r + 3k]
#pragma unroll
                for (int q = 0; q < 9; ++q) { B[0][q] = blocks[(int64_t)q * nslots + t]; B[1][q] = blocks[(int64_t)q * nslots + t2]; }
                const double2* xa = reinterpret_cast<const double2*>(x + (int64_t)3 * BW * adj[t]);
                const double2* xb = reinterpret_cast<const double2*>(x + (int64_t)3 * BW * adj[t2]);
                double xr[2][3 * BW];
#pragma unroll
                for (int q = 0; q < 3 * BW / 2; ++q) {
                    const double2 u = xa[q], w = xb[q];
                    xr[0][2 * q] = u.x; xr[0][2 * q + 1] = u.y; xr[1][2 * q] = w.x; xr[1][2 * q + 1] = w.y;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !two) break;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int c = 0; c < BW; ++c) {
                            const double xv = xr[h][k * BW + c];
#pragma unroll
                            for (int r = 0; r < 3; ++r) acc[r][c] += B[h][r + 3 * k] * xv;
                        }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < BW; ++c) acc[r][c] = group_sum<64>(acc[r][c]);
        if (TPR == 256) {
            __syncthreads();                                         // the previous row's sums have been read
            if (lane == 0)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < BW; ++c) sh[wv][r * BW + c] = acc[r][c];
            __syncthreads();
            if (v < n && threadIdx.x < 3 * BW) {
                const int64_t o = (int64_t)3 * v * BW + threadIdx.x;
                const double tot = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
                y[o] = alpha * tot + s1 * x[o] + (s2 != 0.0 ? s2 * z[o] : 0.0);
            }
        } else if (v < n && lane == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < BW; ++c) {
                    const int64_t o = ((int64_t)3 * v + r) * BW + c;
                    y[o] = alpha * acc[r][c] + s1 * x[o] + (s2 != 0.0 ? s2 * z[o] : 0.0);
                }
        }
    }
}

// The same product on the f64 matrix cores, for the record ("MFMA: measure, expect HBM/L2-bound"):
// v_mfma_f64_4x4x4 multiplies four independent 4x4x4 tiles per instruction, one per group of 16 lanes (which lanes: see k_bsr_spmm_mfma).  A tile = one CSR slot:
// A = its 3x3 block padded to 4x4, B = rows 0..2 of the operand x[adj] x four of its BW = 6 columns (two instructions per slot for
// the six columns), D accumulates that 16-lane group's share of y[v].  One lane holds ONE element of A and of B, so a slot costs 16 lanes
// x 3 loads (9 of the 16 A lanes and 9-12 of the B lanes carry data) against 1 lane x 27 loads in k_bsr_spmm: 1.8x the load lane-operations
// for 54 useful multiply-adds out of the 128 the two instructions perform.
// exact probes: unit vectors through the instruction.  rowmask[la] = lanes of D that see A's lane la (b = ones), colmask[lb] = lanes
// of D that see B's lane lb (a = ones), compat[la][lb] = 1 iff A's lane la and B's lane lb meet in some product (same block, same k)
__global__ __launch_bounds__(64) void k_mfma_layout_probe(unsigned long long* rowmask, unsigned long long* colmask, unsigned char* compat) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, 1.0, 0.0, 0, 0, 0);
        const unsigned long long mk = __ballot(d != 0.0);
        if (lane == 0) rowmask[la] = mk;
    }
    for (int lb = 0; lb < 64; ++lb) {
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, lane == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
        const unsigned long long mk = __ballot(d != 0.0);
        if (lane == 0) colmask[lb] = mk;
    }
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
            const unsigned long long mk = __ballot(d != 0.0);
            if (lane == 0) compat[la * 64 + lb] = mk != 0ull;
        }
}
// Operand layout of v_mfma_f64_4x4x4 on gfx950 (found with the probe above, checked against it at every call of
// desc_debug_spmm_variants): the four tiles are NOT 16-lane groups -- tile y = (lane >> 2) & 3; A: row = lane & 3, k = lane >> 4;
// B: column = lane & 3, k = lane >> 4; D: row = lane >> 4, column = lane & 3.
__global__ __launch_bounds__(256) void k_bsr_spmm_mfma(const int32_t* rowptr, const int32_t* adj, const double* blocks, int64_t nslots, const double* x,
                                                       const double* z, double* y, int n, double alpha, double s1, double s2) {
    __shared__ double sh[16][3 * BW];                                // one partial y per tile (4 waves x 4 tiles)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lx = lane & 3, ty = (lane >> 2) & 3, lr = lane >> 4, tile = 4 * wv + ty;
    const bool a_on = lx < 3 && lr < 3, b_on = lr < 3;               // A element (row lx, component lr); B element (component lr, column lx)
    for (int v = blockIdx.x; v < n; v += gridDim.x) {
        double d0 = 0.0, d1 = 0.0;                                   // columns 0..3 and 4..7 (6, 7 unused) of row lr of this tile's partial y
        const int r1 = rowptr[v + 1];
        for (int t0 = rowptr[v]; t0 < r1; t0 += 16) {                // 16 slots per workgroup step, one per tile
            const int t = t0 + tile;
            const bool on = t < r1;
            const int tt = on ? t : t0;
            const int u = adj[tt];
            const double av = (on && a_on) ? blocks[(int64_t)(lx + 3 * lr) * nslots + tt] : 0.0;
            const double b0 = (on && b_on) ? x[(int64_t)3 * BW * u + lr * BW + lx] : 0.0;
            const double b1 = (on && b_on && lx < BW - 4) ? x[(int64_t)3 * BW * u + lr * BW + 4 + lx] : 0.0;
            d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b1, d1, 0, 0, 0);
        }
        __syncthreads();
        if (lr < 3) { sh[tile][lr * BW + lx] = d0; if (lx < BW - 4) sh[tile][lr * BW + 4 + lx] = d1; }
        __syncthreads();
        if (threadIdx.x < 3 * BW) {
            double tot = 0.0;
            for (int g = 0; g < 16; ++g) tot += sh[g][threadIdx.x];
            const int64_t o = (int64_t)3 * v * BW + threadIdx.x;
            y[o] = alpha * tot + s1 * x[o] + (s2 != 0.0 ? s2 * z[o] : 0.0);
        }
    }
}

// partial Gram matrices per workgroup: G1 = X'Y, G2 = Y'Y  (rows = 3n)
__global__ __launch_bounds__(256) void k_gram(const double* X, const double* Y, int64_t rows, double* partials) {
    double g1[BW][BW], g2[BW][BW];
#pragma unroll
    for (int a = 0; a < BW; ++a)
#pragma unroll
        for (int b = 0; b < BW; ++b) { g1[a][b] = 0.0; g2[a][b] = 0.0; }
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        double xr[BW], yr[BW];
#pragma unroll
        for (int c = 0; c < BW; ++c) { xr[c] = X[r * BW + c]; yr[c] = Y[r * BW + c]; }
#pragma unroll
        for (int a = 0; a < BW; ++a)
#pragma unroll
            for (int b = 0; b < BW; ++b) { g1[a][b] += xr[a] * yr[b]; g2[a][b] += yr[a] * yr[b]; }
    }
    __shared__ double sh[4][2 * BW * BW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < BW; ++a)
#pragma unroll
        for (int b = 0; b < BW; ++b) {
            const double s1 = group_sum<64>(g1[a][b]), s2 = group_sum<64>(g2[a][b]);
            if (lane == 0) { sh[wv][a * BW + b] = s1; sh[wv][BW * BW + a * BW + b] = s2; }
        }
    __syncthreads();
    if (threadIdx.x < 2 * BW * BW)
        partials[(int64_t)blockIdx.x * 2 * BW * BW + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// Xnew = Y * C  (C: BW x BW row-major, passed by value)
struct SmallMat { double c[BW * BW]; };
// blocks[t] (column-major 3x3) of CSR slot t in row v: w_e * dinv[v] * dinv[u] * (v < u ? R_e : R_e')
// -- Spectral.m:24-33 / GCW.m:14-21 without the dense matrix; one wave per row, lanes over its slots
__global__ __launch_bounds__(256) void k_assemble_blocks(const int32_t* rowptr, const int32_t* adj, const int32_t* adj_eid, const double* rij,
                                                         const double* wts, const double* dinv, double* blocks, int64_t nslots, int n) {
    const int lane = threadIdx.x & 63;
    const int w0 = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    for (int v = w0; v < n; v += nw) {
        const double dv = dinv[v];
        for (int t = rowptr[v] + lane; t < rowptr[v + 1]; t += 64) {
            const int u = adj[t], e = adj_eid[t];
            const double du = dinv[u];
            const double w = (wts ? wts[e] : 1.0) * (v < u ? dv : du) * (v < u ? du : dv);     // same rounding in (v,u) and (u,v): exactly symmetric
            const double* R = rij + 9 * (int64_t)e;
            double* b = blocks + t;                                  // component q at b[q * nslots]
            if (v < u) { for (int q = 0; q < 9; ++q) b[(int64_t)q * nslots] = w * R[q]; }
            else { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) b[(int64_t)(r + 3 * c) * nslots] = w * R[c + 3 * r]; }
        }
    }
}

// GCW.m:20 on the device: Weights = 1 ./ (SVec.^(1.5) + 1e-8) per edge
__global__ void k_gcw_weights(const double* S, double* w, int64_t m) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) w[e] = 1.0 / (pow(S[e], 1.5) + 1e-8);
}
// weighted degree of every node (GCW.m:21 sum(Weights, 2)): 16 lanes per CSR row, fixed order
__global__ __launch_bounds__(256) void k_row_wsum(const int32_t* rowptr, const int32_t* adj_eid, const double* w, double* deg, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int vb = row0 - (row0 % 4); vb < n; vb += nrows) {
        const int v = vb + (row0 % 4);
        double acc = 0.0;
        if (v < n) for (int t = rowptr[v] + l16; t < rowptr[v + 1]; t += 16) acc += w ? w[adj_eid[t]] : 1.0;
        acc = group16_sum(acc);
        if (v < n && l16 == 0) deg[v] = acc;
    }
}

struct ResArgs { double z[BW * 3]; double theta[3]; };
// explicit residuals of the three wanted Ritz pairs: partial sums of |Y z_c - theta_c X z_c|^2
// (the Gram-matrix form z'G2z - theta^2 cancels catastrophically below ~1e-8 relative)
__global__ __launch_bounds__(256) void k_residual(const double* X, const double* Y, int64_t rows, ResArgs a, double* partials) {
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double yz = 0.0, xz = 0.0;
#pragma unroll
            for (int k = 0; k < BW; ++k) { yz += Y[r * BW + k] * a.z[k * 3 + c]; xz += X[r * BW + k] * a.z[k * 3 + c]; }
            const double d = yz - a.theta[c] * xz;
            acc[c] += d * d;
        }
    }
    __shared__ double sh[4][3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const double s = group_sum<64>(acc[c]); if (lane == 0) sh[wv][c] = s; }
    __syncthreads();
    if (threadIdx.x < 3) partials[(int64_t)blockIdx.x * 3 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ void k_combine(const double* Y, double* Xn, int64_t rows, SmallMat C) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        double yr[BW];
#pragma unroll
        for (int c = 0; c < BW; ++c) yr[c] = Y[r * BW + c];
#pragma unroll
        for (int b = 0; b < BW; ++b) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < BW; ++a) s += yr[a] * C.c[a * BW + b];
            Xn[r * BW + b] = s;
        }
    }
}

// ---- small dense helpers (host) ---------------------------------------------------------
// cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (row-major); eigenvalues
// descending in w, eigenvectors in the columns of V
template <int N>
void jacobi_eig(const double* Ain, double* w, double* V) {
    double A[N][N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { A[i][j] = 0.5 * (Ain[i * N + j] + Ain[j * N + i]); V[i * N + j] = i == j; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < N; ++p) for (int q = p + 1; q < N; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                if (std::fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < N; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < N; ++k) { const double vkp = V[k * N + p], vkq = V[k * N + q]; V[k * N + p] = c * vkp - s * vkq; V[k * N + q] = s * vkp + c * vkq; }
            }
    }
    int idx[N];
    for (int i = 0; i < N; ++i) idx[i] = i;
    std::sort(idx, idx + N, [&](int a, int b) { return A[a][a] > A[b][b]; });
    double Vs[N * N];
    for (int c = 0; c < N; ++c) { w[c] = A[idx[c]][idx[c]]; for (int k = 0; k < N; ++k) Vs[k * N + c] = V[k * N + idx[c]]; }
    std::memcpy(V, Vs, sizeof Vs);
}

// R = U*diag(1,1,det(U*V'))*V' for the 3x3 block M (row-major)  (Spectral.m:43-45)
void project_so3(const double* M, double* R) {
    // eigen-decomposition of M'M gives V and the singular values; U = M V / sigma
    double MtM[9], w[3], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s = 0; for (int k = 0; k < 3; ++k) s += M[k * 3 + a] * M[k * 3 + b]; MtM[a * 3 + b] = s; }
    jacobi_eig<3>(MtM, w, V);
    double U[9];
    const double s0 = std::sqrt(std::max(w[0], 0.0));
    int good = 0;
    for (int c = 0; c < 3; ++c) {
        const double sc = std::sqrt(std::max(w[c], 0.0));
        if (sc > 1e-12 * std::max(s0, 1e-300) && sc > 1e-300) {
            for (int r = 0; r < 3; ++r) { double s = 0; for (int k = 0; k < 3; ++k) s += M[r * 3 + k] * V[k * 3 + c]; U[r * 3 + c] = s / sc; }
            good = c + 1;
        } else break;
    }
    if (good == 0) { for (int i = 0; i < 9; ++i) { U[i] = (i % 4 == 0); V[i] = (i % 4 == 0); } good = 3; }   // svd(0): U = V = I
    if (good == 1) {      // complete an orthonormal basis
        double a[3] = {U[0], U[3], U[6]}; int k = std::fabs(a[0]) < std::fabs(a[1]) ? (std::fabs(a[0]) < std::fabs(a[2]) ? 0 : 2) : (std::fabs(a[1]) < std::fabs(a[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[k] = 1;
        double b[3] = {a[1] * e[2] - a[2] * e[1], a[2] * e[0] - a[0] * e[2], a[0] * e[1] - a[1] * e[0]};
        const double nb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        for (int r = 0; r < 3; ++r) U[r * 3 + 1] = b[r] / nb;
        good = 2;
    }
    if (good == 2) {
        const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
        U[2] = a[1] * b[2] - a[2] * b[1]; U[5] = a[2] * b[0] - a[0] * b[2]; U[8] = a[0] * b[1] - a[1] * b[0];
    }
    auto det3 = [](const double* A) { return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]); };
    double UVt[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += U[r * 3 + k] * V[c * 3 + k]; UVt[r * 3 + c] = s; }
    const double d = det3(UVt) < 0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1] + d * U[r * 3 + 2] * V[c * 3 + 2];
}

struct Dev {
    hvec<void*> p;
    ~Dev() { for (void* q : p) dev_free(q); }
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

extern "C" int desc_spectral_run(const desc_problem* prob, const double* weights, int32_t normalize_rows, double tol,
                                 int32_t max_iters, int32_t device, double* R_out, desc_spectral_info* info) {
    if (!prob || !R_out) return fail(DESC_ERR_INVALID, "NULL argument");
    if (prob->n == 0) return validate_problem(prob, true);
    auto t0 = std::chrono::steady_clock::now();
    desc_device_problem* dp = nullptr;
    int rc = desc_problem_upload(prob, device, &dp);
    if (rc) return rc;
    rc = desc_spectral_run_dev(dp, weights, normalize_rows, tol, max_iters, R_out, info);
    desc_problem_free(dp);
    if (!rc && info) info->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

static int spectral_impl(const desc_device_problem* dp, const double* weights, const double* gcw_svec, int32_t normalize_rows, double tol,
                         int32_t max_iters, double* R_out, desc_spectral_info* info);

extern "C" int desc_spectral_run_dev(const desc_device_problem* dp, const double* weights, int32_t normalize_rows, double tol,
                                     int32_t max_iters, double* R_out, desc_spectral_info* info) {
    return no_throw("desc_spectral_run_dev", [&]() -> int {
    return spectral_impl(dp, weights, nullptr, normalize_rows, tol, max_iters, R_out, info);
    });
}
// R_est = GCW(Ind, AdjMat, RijMat, SVec) -- Utils/GCW.m:9-36 with the weights formed on the device from SVec (m doubles)
extern "C" int desc_gcw_run_dev(const desc_device_problem* dp, const double* s_vec, double tol, int32_t max_iters, double* R_out,
                                desc_spectral_info* info) {
    return no_throw("desc_gcw_run_dev", [&]() -> int {
    if (!s_vec) return fail(DESC_ERR_INVALID, "NULL argument");
    return spectral_impl(dp, nullptr, s_vec, 1, tol, max_iters, R_out, info);
    });
}

static int spectral_impl(const desc_device_problem* dp, const double* weights, const double* gcw_svec, int32_t normalize_rows, double tol,
                         int32_t max_iters, double* R_out, desc_spectral_info* info) {
    if (!dp || !R_out) return fail(DESC_ERR_INVALID, "NULL argument");
    int rc = DESC_OK;
    const int64_t n = dp->n, m = dp->m;
    if (n == 0) return DESC_OK;
    DESC_HIP(hipSetDevice(dp->device));
    auto t0 = std::chrono::steady_clock::now();
    auto t_lap = t0;
    const char* tenv = std::getenv("DESC_DEBUG_TIMING");
    const bool timing = tenv && std::atoi(tenv) != 0;
    auto lap = [&](const char* what) {                      // diagnostics: DESC_DEBUG_TIMING=1 prints where the call spends its time
        if (!timing) return;
        (void)hipDeviceSynchronize();
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[desc_amd] spectral %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    if (tol <= 0) tol = 1e-13;
    if (max_iters <= 0) max_iters = 500;

    // block CSR: every edge in both endpoint rows; the index part lives with the device problem, the 2m blocks are
    // assembled on the device ((i,j) slot = R, (j,i) slot = R')
    hvec<double> deg((size_t)n, 0.0);
    Dev W;                                                  // edge weights on the device (NULL: all ones)
    double* d_w = nullptr;
    if (gcw_svec) {                                         // weights and weighted degrees entirely on the device
        double *d_s, *d_deg;
        Dev Tmp;
        if ((rc = W.alloc(&d_w, m)) || (rc = Tmp.alloc(&d_s, m)) || (rc = Tmp.alloc(&d_deg, n))) return rc;
        if (m) DESC_HIP(hipMemcpy(d_s, gcw_svec, sizeof(double) * m, hipMemcpyHostToDevice));
        if (m) hipLaunchKernelGGL(k_gcw_weights, dim3((unsigned)std::min<int64_t>(2048, (m + 255) / 256)), dim3(256), 0, 0, d_s, d_w, m);
        hipLaunchKernelGGL(k_row_wsum, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (n * 16 + 255) / 256))), dim3(256), 0, 0,
                           dp->d_rowptr, dp->d_adj_eid, d_w, d_deg, (int)n);
        DESC_HIP(hipMemcpy(deg.data(), d_deg, sizeof(double) * n, hipMemcpyDeviceToHost));
        for (int64_t v = 0; v < n; ++v) if (!std::isfinite(deg[v])) return fail(DESC_ERR_INVALID, "S_vec holds a negative or non-finite entry (node %lld)", (long long)v);
    } else {
        if (!weights) {                                     // unit weights: the degree is the CSR row length (no pass over the edges)
            for (int64_t v = 0; v < n; ++v) deg[v] = (double)(dp->rowptr[v + 1] - dp->rowptr[v]);
        } else
        for (int64_t e = 0; e < m; ++e) {
            const double w = weights[e];
            if (!(w >= 0) || !std::isfinite(w)) return fail(DESC_ERR_INVALID, "weight %lld is not a finite non-negative number", (long long)e);
            deg[dp->ii[e]] += w; deg[dp->jj[e]] += w;
        }
        if (weights && m) {
            if ((rc = W.alloc(&d_w, m))) return rc;
            DESC_HIP(hipMemcpy(d_w, weights, sizeof(double) * m, hipMemcpyHostToDevice));
        }
    }
    lap("weights + degrees");
    double sigma = 0.0;
    hvec<double> dinv((size_t)n, 1.0);               // D^-1/2
    if (normalize_rows) {
        for (int64_t v = 0; v < n; ++v) dinv[v] = deg[v] > 0 ? 1.0 / std::sqrt(deg[v]) : 0.0;
        sigma = 1.0;                                        // spectrum of D^-1/2 A D^-1/2 lies in [-1,1]
    } else {
        for (int64_t v = 0; v < n; ++v) sigma = std::max(sigma, deg[v]);   // ||A||_2 <= max weighted degree (orthogonal blocks)
    }
    const int64_t rows = 3 * n;
    // deterministic start: hashed pseudo-random entries in [-1,1)
    hvec<double> X0((size_t)rows * BW);
    for (size_t t = 0; t < X0.size(); ++t) X0[t] = (double)(int64_t)(mix64(0xC0FFEEull + t) >> 11) / 4503599627370496.0 - 1.0;

    Dev D;
    const int32_t *d_rowptr = dp->d_rowptr, *d_adj = dp->d_adj; double *d_blocks, *d_X, *d_Y, *d_part;
    const int ggrid = 256;
    if ((rc = D.alloc(&d_blocks, 18 * m)) ||
        (rc = D.alloc(&d_X, rows * BW)) || (rc = D.alloc(&d_Y, rows * BW)) || (rc = D.alloc(&d_part, (size_t)ggrid * 2 * BW * BW))) return rc;
    if (m) {
        Dev T;                                               // assembly inputs, released before the iteration starts
        const int32_t* d_eid = dp->d_adj_eid; const double* d_rij = dp->d_rij; double* d_dinv;
        if ((rc = T.alloc(&d_dinv, n))) return rc;
        DESC_HIP(hipMemcpy(d_dinv, dinv.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_assemble_blocks, dim3((unsigned)std::min<int64_t>(4096, (n + 3) / 4)), dim3(256), 0, 0, d_rowptr, d_adj, d_eid, d_rij, d_w,
                           d_dinv, d_blocks, (int64_t)2 * m, (int)n);
        DESC_HIP(hipGetLastError());
        DESC_HIP(hipDeviceSynchronize());
    }
    DESC_HIP(hipMemcpy(d_Y, X0.data(), sizeof(double) * rows * BW, hipMemcpyHostToDevice));
    lap("assemble + start block");

    hvec<double> part((size_t)ggrid * 2 * BW * BW);
    double G1[BW * BW], G2[BW * BW];
    auto grams = [&](const double* A, const double* B) -> int {
        hipLaunchKernelGGL(k_gram, dim3(ggrid), dim3(256), 0, 0, A, B, rows, d_part);
        DESC_HIP(hipMemcpy(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost));
        for (int t = 0; t < BW * BW; ++t) { double s1 = 0, s2 = 0; for (int b = 0; b < ggrid; ++b) { s1 += part[(size_t)b * 2 * BW * BW + t]; s2 += part[(size_t)b * 2 * BW * BW + BW * BW + t]; } G1[t] = s1; G2[t] = s2; }
        return DESC_OK;
    };
    // C = Z * L^-T where K = Z' G Z = L L'  (columns of Y*C are orthonormal)
    auto ortho_coeffs = [&](const double* G, const double* Z, SmallMat& C) -> bool {
        double K[BW][BW], L[BW][BW] = {};
        for (int a = 0; a < BW; ++a) for (int b = 0; b < BW; ++b) { double s = 0; for (int p = 0; p < BW; ++p) for (int q = 0; q < BW; ++q) s += Z[p * BW + a] * G[p * BW + q] * Z[q * BW + b]; K[a][b] = s; }
        for (int a = 0; a < BW; ++a) {
            for (int b = 0; b <= a; ++b) {
                double s = 0.5 * (K[a][b] + K[b][a]);
                for (int k = 0; k < b; ++k) s -= L[a][k] * L[b][k];
                if (a == b) { if (!(s > 0)) return false; L[a][a] = std::sqrt(s); } else L[a][b] = s / L[b][b];
            }
        }
        // Linv' : solve L' X = I  -> X = L^-T ; then C = Z X
        double Li[BW][BW] = {};
        for (int c = 0; c < BW; ++c)
            for (int r = BW - 1; r >= 0; --r) { double s = (r == c); for (int k = r + 1; k < BW; ++k) s -= L[k][r] * Li[k][c]; Li[r][c] = s / L[r][r]; }
        for (int a = 0; a < BW; ++a) for (int b = 0; b < BW; ++b) { double s = 0; for (int k = 0; k < BW; ++k) s += Z[a * BW + k] * Li[k][b]; C.c[a * BW + b] = s; }
        return true;
    };
    double Ident[BW * BW];
    for (int t = 0; t < BW * BW; ++t) Ident[t] = (t % (BW + 1) == 0);
    // X = orth(X0)
    SmallMat C;
    if ((rc = grams(d_Y, d_Y))) return rc;
    if (!ortho_coeffs(G2, Ident, C)) return fail(DESC_ERR_INVALID, "degenerate start basis");
    hipLaunchKernelGGL(k_combine, dim3(512), dim3(256), 0, 0, d_Y, d_X, rows, C);

    // Chebyshev-filtered subspace iteration (Zhou & Saad): each outer step damps the unwanted
    // part of the spectrum [lo, cut] with a degree-CHEB_DEG Chebyshev polynomial of A, then
    // Rayleigh-Ritz.  lo = -sigma is a lower bound of the spectrum; cut = smallest Ritz value of
    // the block (the largest unwanted eigenvalue's estimate).  A plain power step (degree 1)
    // needs ~1/gap products when the top of the spectrum is clustered (GCW weights span many
    // orders of magnitude); the filter needs ~1/sqrt(gap).
    // Round 4 (profiles/r04_spectral_experiments.txt): the damped interval follows the spectrum instead of the norm bound.  lo = -sigma (max
    // weighted degree) is safe, but for the graphs DESC meets it lies ~20x below the bulk of the spectrum (C4: -1100 against ~-57), and a
    // Chebyshev polynomial scaled to [-1100, 57] grows only 4x per product at the wanted end (700) -- slower than the plain power method.
    // After a first short pass with the safe bound the block's own Ritz values estimate the bulk: lo = -2 max(|theta_4|, |theta_6|), never above
    // -2 % of sigma.  An eigenvalue below that is amplified too, but far less than the wanted ones, and Rayleigh-Ritz keeps it out of the top
    // three; should the block drift to the negative end all the same (theta_6 <= lo, or the residual stops falling) the safe bound comes back
    // for the rest of the call.  The degree of a pass is what takes the residual to 0.2 tol -- T_d((theta_3 - c) / e) >= res / (0.2 tol) -- but
    // never more than what amplifies the wanted directions 1e6 times over the guard vectors: beyond ~1e8 the Gram matrix of the filtered
    // 3n x 6 block is singular in double precision and the Cholesky-based orthogonalisation breaks down (the first form of this change did).
    // Also tried: the degree from the digits per product of the previous pass -- 77 products instead of 52, a pass's gain is not proportional
    // to its degree.  DESC_DEBUG_SPECTRAL_TIGHT=0: round 3's fixed scheme (three passes of degree 16 at C4: 52 products).
    constexpr int CHEB_DEG = 16, CHEB_MIN = 2;
    const int CHEB_FIRST = [] { const char* v = std::getenv("DESC_DEBUG_SPECTRAL_FIRST"); return v ? std::max(2, std::atoi(v)) : 8; }();      // degree of the first pass (safe bound)
    const bool tight = [] { const char* v = std::getenv("DESC_DEBUG_SPECTRAL_TIGHT"); return v ? std::atoi(v) != 0 : true; }();
    bool tight_ok = tight;
    int cheb_deg = tight ? CHEB_FIRST : CHEB_DEG, deg_prev = 0;
    double res_prev = -1.0;
    double *d_P, *d_Q;
    if ((rc = D.alloc(&d_P, rows * BW)) || (rc = D.alloc(&d_Q, rows * BW))) return rc;
    double theta[BW] = {}, Z[BW * BW], res = 1e300;
    int it = 0, products = 0;
    bool converged = false;
    const bool wide_rows = 2 * m >= 192 * n;                   // average row of >= 192 slots: a workgroup per row, else a wave
    const int sgrid = (int)std::min<int64_t>(8192, wide_rows ? n : (n + 3) / 4);
    auto spmm = [&](const double* x, const double* z, double* y, double alpha, double s1, double s2) {
        if (wide_rows) hipLaunchKernelGGL(k_bsr_spmm<256>, dim3(sgrid), dim3(256), 0, 0, d_rowptr, d_adj, d_blocks, (int64_t)2 * m, x, z, y, (int)n, alpha, s1, s2);
        else hipLaunchKernelGGL(k_bsr_spmm<64>, dim3(sgrid), dim3(256), 0, 0, d_rowptr, d_adj, d_blocks, (int64_t)2 * m, x, z, y, (int)n, alpha, s1, s2);
        ++products;
    };
    double lo = -sigma;
    double cut = 0.0;                                         // set after the first Rayleigh-Ritz
    for (it = 1; it <= max_iters; ++it) {
        // ---- Rayleigh-Ritz on the current orthonormal basis X
        spmm(d_X, d_X, d_Y, 1.0, 0.0, 0.0);                  // Y = A X
        if ((rc = grams(d_X, d_Y))) return rc;               // G1 = X'AX, G2 = Y'Y
        jacobi_eig<BW>(G1, theta, Z);
        {
            ResArgs ra;
            for (int k = 0; k < BW; ++k) for (int c = 0; c < 3; ++c) ra.z[k * 3 + c] = Z[k * BW + c];
            for (int c = 0; c < 3; ++c) ra.theta[c] = theta[c];
            hipLaunchKernelGGL(k_residual, dim3(ggrid), dim3(256), 0, 0, d_X, d_Y, rows, ra, d_part);
            DESC_HIP(hipMemcpy(part.data(), d_part, sizeof(double) * 3 * ggrid, hipMemcpyDeviceToHost));
            res = 0.0;
            const double scale = std::max(std::fabs(theta[0]), sigma);
            for (int c = 0; c < 3; ++c) { double r2 = 0; for (int bb = 0; bb < ggrid; ++bb) r2 += part[(size_t)3 * bb + c]; res = std::max(res, std::sqrt(r2) / std::max(scale, 1e-300)); }
        }
        if (timing) fprintf(stderr, "[desc_amd] spectral  outer %d: residual %.3e after %d products (last pass: degree %d)\n", it, res, products, deg_prev);
        if (res <= tol) { converged = true; break; }

        // ---- filter: P <- p(A) X with p small on [lo, cut], large above
        cut = theta[BW - 1];
        const double top = theta[0];
        if (tight_ok && it >= 2) {
            if (theta[BW - 1] <= lo || (res_prev > 0.0 && res > res_prev)) { lo = -sigma; tight_ok = false; }          // drifting to the negative end: the safe bound
            else lo = std::max(-sigma, -std::max(2.0 * std::max(std::fabs(theta[3]), std::fabs(theta[BW - 1])), 0.02 * sigma));
        }
        if (!(cut > lo) || !(top > cut)) cut = lo + 0.5 * (top - lo);   // degenerate block: fall back to a mild filter
        const double e = 0.5 * (cut - lo), c = 0.5 * (cut + lo);
        if (tight && it >= 2) {
            const double xi3 = (theta[2] - c) / e;
            cheb_deg = CHEB_DEG;
            if (xi3 > 1.0 + 1e-6 && res > 0.0) {
                const double want = std::acosh(std::max(2.0, res / (0.2 * tol))), cap = std::acosh(1e6);
                cheb_deg = (int)std::max<double>(CHEB_MIN, std::min<double>(CHEB_DEG, std::ceil(std::min(want, cap) / std::acosh(xi3))));
            }
        }
        res_prev = res; deg_prev = cheb_deg;
        double s_prev = e / (top - c);
        const double s1c = s_prev;
        // P = (A X - c X) * s_prev / e
        spmm(d_X, d_X, d_P, s_prev / e, -c * s_prev / e, 0.0);
        double* Xp = d_X; double* Pp = d_P; double* Qp = d_Q;  // X_{k-1}, X_k, scratch
        for (int k = 2; k <= cheb_deg; ++k) {
            const double s_new = 1.0 / (2.0 / s1c - s_prev);
            // Q = (2 s_new / e) (A P - c P) - (s_prev s_new) X_{k-1}
            spmm(Pp, Xp, Qp, 2.0 * s_new / e, -2.0 * s_new * c / e, -s_prev * s_new);
            double* t3 = Xp; Xp = Pp; Pp = Qp; Qp = t3;
            s_prev = s_new;
        }
        // ---- X <- orth(filtered block)
        if ((rc = grams(Pp, Pp))) return rc;
        SmallMat Co;
        if (!ortho_coeffs(G2, Ident, Co)) return fail(DESC_ERR_INVALID, "subspace iteration broke down (rank-deficient block)");
        double* Xn = (Pp == d_X) ? (Xp == d_P ? d_Q : d_P) : d_X;       // a buffer that is not Pp
        hipLaunchKernelGGL(k_combine, dim3(512), dim3(256), 0, 0, Pp, Xn, rows, Co);
        if (Xn != d_X) DESC_HIP(hipMemcpyAsync(d_X, Xn, sizeof(double) * rows * BW, hipMemcpyDeviceToDevice, 0));
    }
    // Ritz vectors of the converged subspace: V = X Z (X still holds the basis G1 was formed with)
    // Z belongs to the basis in d_X in both exits (the loop leaves right after a Rayleigh-Ritz,
    // or after max_iters with the last Ritz rotation still unapplied)
    if (!converged && it > max_iters) {                       // last pass replaced X after its RR: redo RR once
        spmm(d_X, d_X, d_Y, 1.0, 0.0, 0.0);
        if ((rc = grams(d_X, d_Y))) return rc;
        jacobi_eig<BW>(G1, theta, Z);
    }
    lap("subspace iteration");
    SmallMat CZ; std::memcpy(CZ.c, Z, sizeof Z);
    hipLaunchKernelGGL(k_combine, dim3(512), dim3(256), 0, 0, d_X, d_Y, rows, CZ);
    DESC_HIP(hipGetLastError());
    hvec<double> Vh((size_t)rows * BW);
    DESC_HIP(hipMemcpy(Vh.data(), d_Y, sizeof(double) * rows * BW, hipMemcpyDeviceToHost));

    // back to the eigenvectors of D^-1 A, unit 2-norm columns (what eigs returns)   (GCW.m:21,27)
    hvec<double> V((size_t)rows * 3);
    double nrm[3] = {0, 0, 0};
    for (int64_t v = 0; v < n; ++v) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        const double x = Vh[((size_t)3 * v + r) * BW + c] * (normalize_rows ? dinv[v] : 1.0);
        V[((size_t)3 * v + r) * 3 + c] = x; nrm[c] += x * x;
    }
    for (size_t t = 0; t < V.size(); ++t) V[t] /= std::sqrt(nrm[t % 3]);
    // V(:,1) = V(:,1)*sign(det(V(1:3,:)))   (Spectral.m:39 / GCW.m:28)
    {
        const double* A = V.data();
        const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
        const double sg = det > 0 ? 1.0 : (det < 0 ? -1.0 : 0.0);
        for (int64_t r = 0; r < rows; ++r) V[(size_t)r * 3] *= sg;
    }
    {   // per-node SVD projection (Spectral.m:41-46): independent 3x3 problems, host threads for large n
        unsigned hw = std::thread::hardware_concurrency();
        const int T = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / 512));
        run_threads(T, [&](int t) {
            for (int64_t v = n * t / T; v < n * (t + 1) / T; ++v) {
                double R[9];
                project_so3(&V[(size_t)9 * v], R);           // rows 3v..3v+2 of V = the node's 3x3 block, row-major
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_out[9 * v + r + 3 * c] = R[r * 3 + c];   // MATLAB column-major 3x3xn
            }
        });
    }
    lap("normalise + project");
    if (info) {
        info->iters = std::min(it, max_iters);
        info->converged = converged ? 1 : 0;
        info->residual = res;
        for (int c = 0; c < 3; ++c) info->eigenvalues[c] = theta[c];
        info->products = products;
        info->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return DESC_OK;
}


// Diagnostics (tools/next_rows_bench.py): times the block SpMM of the Spectral connection matrix (unit weights) in its two forms
// -- k_bsr_spmm (vector FMA) and k_bsr_spmm_mfma (v_mfma_f64_4x4x4) -- on the same operand, `reps` products each, HIP events.
// out[0], out[1] = ms per product (VALU, MFMA; -1 if the MFMA operand layout could not be identified), out[2] = max |difference| of
// the two results, out[3] = the layout code found.
extern "C" int desc_debug_spmm_variants(const desc_device_problem* dp, int32_t reps, double* out) {
    return no_throw("desc_debug_spmm_variants", [&]() -> int {
    if (!dp || !out || reps < 1) return fail(DESC_ERR_INVALID, "bad argument");
    DESC_HIP(hipSetDevice(dp->device));
    const int64_t n = dp->n, m = dp->m, rows = 3 * n;
    if (n == 0 || m == 0) return fail(DESC_ERR_INVALID, "empty problem");
    int rc;
    Dev D;
    double *d_blocks, *d_X, *d_Y1, *d_Y2, *d_dinv;
    if ((rc = D.alloc(&d_blocks, 18 * m)) || (rc = D.alloc(&d_X, rows * BW)) || (rc = D.alloc(&d_Y1, rows * BW)) || (rc = D.alloc(&d_Y2, rows * BW)) ||
        (rc = D.alloc(&d_dinv, n))) return rc;
    hvec<double> ones((size_t)n, 1.0), X0((size_t)rows * BW);
    for (size_t t = 0; t < X0.size(); ++t) X0[t] = (double)(int64_t)(mix64(0xC0FFEEull + t) >> 11) / 4503599627370496.0 - 1.0;
    DESC_HIP(hipMemcpy(d_dinv, ones.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    DESC_HIP(hipMemcpy(d_X, X0.data(), sizeof(double) * rows * BW, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_assemble_blocks, dim3((unsigned)std::min<int64_t>(4096, (n + 3) / 4)), dim3(256), 0, 0, dp->d_rowptr, dp->d_adj, dp->d_adj_eid, dp->d_rij,
                       (const double*)nullptr, d_dinv, d_blocks, (int64_t)2 * m, (int)n);
    // operand layout of v_mfma_f64_4x4x4: the unit-vector probes must show exactly the layout k_bsr_spmm_mfma is written for --
    // A's lane (x, y, r) reaches D's lanes (*, y, x), B's lane (x, y, r) reaches D's lanes (x, y, *), with lane = 16 r + 4 y + x,
    // and an A lane meets a B lane iff they share the tile y and the summation index r
    int lay_ok = 1;
    {
        unsigned long long* d_rm; unsigned long long* d_cm; unsigned char* d_cp;
        if ((rc = D.alloc(&d_rm, 64)) || (rc = D.alloc(&d_cm, 64)) || (rc = D.alloc(&d_cp, 4096))) return rc;
        hipLaunchKernelGGL(k_mfma_layout_probe, dim3(1), dim3(64), 0, 0, d_rm, d_cm, d_cp);
        unsigned long long rm[64], cm[64]; unsigned char cp[4096];
        DESC_HIP(hipMemcpy(rm, d_rm, sizeof rm, hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(cm, d_cm, sizeof cm, hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(cp, d_cp, sizeof cp, hipMemcpyDeviceToHost));
        if (std::getenv("DESC_DEBUG_TIMING"))                  // diagnostics: the raw footprints
            for (int q = 0; q < 64; ++q) fprintf(stderr, "[desc_amd] mfma probe lane %2d: A -> D %016llx   B -> D %016llx\n", q, rm[q], cm[q]);
        for (int l = 0; l < 64; ++l) {
            const int lx = l & 3, ly = (l >> 2) & 3;
            if (rm[l] != 0xFull << (16 * lx + 4 * ly)) lay_ok = 0;
            if (cm[l] != 0x0001000100010001ull << (4 * ly + lx)) lay_ok = 0;
            for (int l2 = 0; l2 < 64; ++l2)
                if ((cp[l * 64 + l2] != 0) != (((l >> 2) & 3) == ((l2 >> 2) & 3) && (l >> 4) == (l2 >> 4))) lay_ok = 0;
        }
    }
    const int lay = lay_ok ? 1 : -1;
    const bool wide = 2 * m >= 192 * n;
    const int sgrid = (int)std::min<int64_t>(8192, wide ? n : (n + 3) / 4);
    hipEvent_t e0, e1;
    DESC_HIP(hipEventCreate(&e0)); DESC_HIP(hipEventCreate(&e1));
    auto timed = [&](int which) -> double {
        for (int r = -2; r < reps; ++r) {                              // two untimed warm-up products
            if (r == 0) (void)hipEventRecord(e0, 0);
            if (which == 0) {
                if (wide) hipLaunchKernelGGL(k_bsr_spmm<256>, dim3(sgrid), dim3(256), 0, 0, dp->d_rowptr, dp->d_adj, d_blocks, (int64_t)2 * m, d_X, d_X, d_Y1, (int)n, 1.0, 0.0, 0.0);
                else hipLaunchKernelGGL(k_bsr_spmm<64>, dim3(sgrid), dim3(256), 0, 0, dp->d_rowptr, dp->d_adj, d_blocks, (int64_t)2 * m, d_X, d_X, d_Y1, (int)n, 1.0, 0.0, 0.0);
            } else
                hipLaunchKernelGGL(k_bsr_spmm_mfma, dim3((unsigned)std::min<int64_t>(8192, n)), dim3(256), 0, 0, dp->d_rowptr, dp->d_adj, d_blocks, (int64_t)2 * m, d_X, d_X, d_Y2, (int)n, 1.0, 0.0, 0.0);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / reps;
    };
    out[0] = timed(0);
    out[1] = lay >= 0 ? timed(1) : -1.0;
    out[2] = 0.0; out[3] = lay;
    if (lay >= 0) {
        hvec<double> y1((size_t)rows * BW), y2((size_t)rows * BW);
        DESC_HIP(hipMemcpy(y1.data(), d_Y1, sizeof(double) * rows * BW, hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(y2.data(), d_Y2, sizeof(double) * rows * BW, hipMemcpyDeviceToHost));
        for (size_t t = 0; t < y1.size(); ++t) out[2] = std::max(out[2], std::fabs(y1[t] - y2[t]));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DESC_HIP(hipGetLastError());
    return DESC_OK;
    });
}
