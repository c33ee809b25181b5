// Host-side marshalling of the reference's argument formats (the caller's side of DESC_PGD.m:14, `Ind` and `RijMat` as MATLAB / NumPy hold
// them) into the C ABI's arrays: one threaded pass each, no device code.  A binding that already holds int32 endpoints and MATLAB's own
// 3 x 3 x m memory (the MEX shim) needs neither.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "node_plan.h"

using namespace desc;

namespace {

// status of a range of rows: first offending row and what is wrong with it (smallest row wins, like a sequential scan)
struct RowStatus { int64_t row = -1; int kind = 0; int32_t max_node = 0; bool sorted = true; };
enum { BAD_NONE = 0, BAD_NOT_INTEGER = 1, BAD_RANGE = 2 };

template <class T> inline bool to_node(T v, int64_t& out) { out = (int64_t)v; return true; }
template <> inline bool to_node<double>(double v, int64_t& out) {
    if (!(v >= -9.0e15 && v <= 9.0e15)) return false;          // NaN, Inf and values beyond the exactly-representable integers
    out = (int64_t)v;
    return (double)out == v;
}

template <class T>
int marshal_edges_t(const T* ind, int64_t m, int64_t row_stride, int64_t col_stride, int32_t* ind_i, int32_t* ind_j, int64_t* n_out, int32_t* sorted_out) {
    const int64_t grain = 1 << 18;
    unsigned hw = std::thread::hardware_concurrency();
    const int T_ = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, m / grain));
    std::vector<RowStatus> st((size_t)T_);
    run_threads(T_, [&](int t) {
        RowStatus s;
        const int64_t a = m * t / T_, b = m * (t + 1) / T_;
        int64_t pi = -1, pj = -1;
        if (a > 0) {                                            // the row before the range, for the order test (unchecked values compare as they are)
            int64_t x, y;
            if (to_node<T>(ind[(a - 1) * row_stride], x) && to_node<T>(ind[(a - 1) * row_stride + col_stride], y)) { pi = x; pj = y; }
        }
        for (int64_t e = a; e < b; ++e) {
            int64_t i, j;
            if (!to_node<T>(ind[e * row_stride], i) || !to_node<T>(ind[e * row_stride + col_stride], j)) { s.row = e; s.kind = BAD_NOT_INTEGER; break; }
            if (i < 1 || i >= j || j > 0x7FFFFFFFll) { s.row = e; s.kind = BAD_RANGE; break; }
            if (i < pi || (i == pi && j <= pj)) s.sorted = false;
            pi = i; pj = j;
            ind_i[e] = (int32_t)(i - 1); ind_j[e] = (int32_t)(j - 1);
            if ((int32_t)j > s.max_node) s.max_node = (int32_t)j;
        }
        st[(size_t)t] = s;
    });
    bool sorted = true; int32_t n = 0;
    for (const RowStatus& s : st) {
        if (s.row >= 0) {
            if (s.kind == BAD_NOT_INTEGER) return fail(DESC_ERR_INVALID, "Ind must hold integer node ids (row %lld)", (long long)s.row);
            return fail(DESC_ERR_INVALID, "Ind rows must be 1-based with Ind(:,1) < Ind(:,2) (row %lld)", (long long)s.row);
        }
        sorted = sorted && s.sorted;
        n = std::max(n, s.max_node);
    }
    *n_out = n;                                                 // n = max(Ind(:)), DESC_PGD.m:21
    *sorted_out = sorted ? 1 : 0;
    return DESC_OK;
}

}  // namespace

extern "C" {

int desc_marshal_edges(const void* ind, int32_t dtype, int64_t m, int64_t row_stride, int64_t col_stride,
                       int32_t* ind_i, int32_t* ind_j, int64_t* n_out, int32_t* sorted_out) {
    return no_throw("desc_marshal_edges", [&]() -> int {
        if (m < 0 || (m > 0 && (!ind || !ind_i || !ind_j)) || !n_out || !sorted_out) return fail(DESC_ERR_INVALID, "NULL argument or negative m");
        if (m >= (1ll << 30)) return fail(DESC_ERR_TOO_LARGE, "m = %lld exceeds 2^30-1", (long long)m);
        switch (dtype) {
            case DESC_DTYPE_F64: return marshal_edges_t((const double*)ind, m, row_stride, col_stride, ind_i, ind_j, n_out, sorted_out);
            case DESC_DTYPE_I64: return marshal_edges_t((const int64_t*)ind, m, row_stride, col_stride, ind_i, ind_j, n_out, sorted_out);
            case DESC_DTYPE_I32: return marshal_edges_t((const int32_t*)ind, m, row_stride, col_stride, ind_i, ind_j, n_out, sorted_out);
        }
        return fail(DESC_ERR_INVALID, "dtype must be DESC_DTYPE_F64, _I64 or _I32");
    });
}

int desc_marshal_rij(const double* R, int64_t m, int64_t stride_r, int64_t stride_c, int64_t stride_l, const int64_t* perm, double* out) {
    return no_throw("desc_marshal_rij", [&]() -> int {
        if (m < 0 || (m > 0 && (!R || !out))) return fail(DESC_ERR_INVALID, "NULL argument or negative m");
        host_parallel(m, [&](int64_t a, int64_t b) {
            for (int64_t l = a; l < b; ++l) {
                const double* src = R + (perm ? perm[l] : l) * stride_l;
                double* dst = out + 9 * l;
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < 3; ++r) dst[r + 3 * c] = src[r * stride_r + c * stride_c];          // MATLAB memory order of a 3 x 3 x m array
            }
        }, 1 << 15);
        return DESC_OK;
    });
}

}  // extern "C"
