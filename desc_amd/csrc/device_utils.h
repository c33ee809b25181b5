// Device-side helpers for gfx950 (wave64): group reductions by DPP / permlane-swap,
// XCD-aware block remap, HIP error plumbing.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

#define DESC_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess)                                                                 \
            return desc::fail(DESC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                            \
    } while (0)

namespace desc {

// device memory through the library's block cache (devmem.hip): same contract as hipMalloc / hipFree
hipError_t dev_alloc(void** out, size_t bytes);
hipError_t dev_alloc_uncached(void** out, size_t bytes);      // MTYPE UC block (same cache, never handed out as an ordinary one)
void dev_free(void* p);
void dev_free_idle(void* p);                                   // the same without the device synchronisation: the caller has just synchronised
// Streams are pooled like blocks: creating and destroying a HIP stream costs 1-2 ms each (a hardware queue), more than the whole PGD loop of
// the reference's demo-size graphs (C1: 1.7 ms of 6.3 ms per solve).  stream_release: the stream must be idle.
hipError_t stream_acquire(hipStream_t* out);
void stream_release(hipStream_t s);

// One DPP-moved copy of a double (two 32-bit v_mov_b32_dpp).  All four controls used
// here are permutations of the full wave, so every lane is written and no `old` value
// has to be preserved (mov_dpp leaves it undefined: no extra register copy).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }

// sums over aligned groups of 8 lanes (half a DPP row); every lane gets its group's total
__device__ __forceinline__ double group8_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);    // quad_perm:[1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);    // quad_perm:[2,3,0,1]
    v += dpp_mov_f64<0x141>(v);   // row_half_mirror
    return v;
}
__device__ __forceinline__ int group8_sum(int v) {
    v += dpp_mov_i32<0xB1>(v);
    v += dpp_mov_i32<0x4E>(v);
    v += dpp_mov_i32<0x141>(v);
    return v;
}
// sums over one DPP row (16 lanes); every lane gets its row's total
__device__ __forceinline__ double group16_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ int group16_sum(int v) {
    v += dpp_mov_i32<0xB1>(v);
    v += dpp_mov_i32<0x4E>(v);
    v += dpp_mov_i32<0x141>(v);
    v += dpp_mov_i32<0x140>(v);
    return v;
}
// v_permlane16_swap: exchanges the odd 16-lane rows of the first operand with the
// even rows of the second; with both operands = v the two results are
// [r0,r0,r2,r2] and [r1,r1,r3,r3], whose sum is the xor-16 butterfly.
__device__ __forceinline__ double swap16_sum_f64(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// v_permlane32_swap: exchanges the upper half of the first operand with the lower
// half of the second -> [lo,lo] and [hi,hi]; sum = xor-32 butterfly.
__device__ __forceinline__ double swap32_sum_f64(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Sum over aligned groups of G lanes (G = 16, 32 or 64); every lane of a group
// receives its group's total.  Fixed butterfly order -> bitwise reproducible.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "group width");
    v += dpp_mov_f64<0xB1>(v);    // quad_perm:[1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);    // quad_perm:[2,3,0,1]
    v += dpp_mov_f64<0x141>(v);   // row_half_mirror
    if (G >= 16) v += dpp_mov_f64<0x140>(v);   // row_mirror
    if (G >= 32) v = swap16_sum_f64(v);
    if (G >= 64) v = swap32_sum_f64(v);
    return v;
}

// number of set predicate lanes in this lane's group of G
template <int G>
__device__ __forceinline__ int group_count(bool pred, int lane) {
    unsigned long long mk = __ballot(pred);
    if (G == 64) return __popcll(mk);
    constexpr unsigned long long field = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
    return __popcll((mk >> ((lane / G) * G)) & field);
}

// Wave-uniform load through the scalar cache (s_load_dword, counted by lgkmcnt, not by
// the in-order vmcnt the software pipelines rely on).  Only for tables that no kernel
// writes while it runs; the index must be wave-uniform.
__device__ __forceinline__ int uniform_load(const int32_t* p, int i) {
    typedef const int32_t __attribute__((address_space(4)))* cptr_t;
    return ((cptr_t)(unsigned long long)p)[i];
}

// device copy of desc::sample_key (common.h): the cycle-sampling key
__device__ __forceinline__ uint64_t d_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint64_t d_sample_key(uint64_t seed, uint64_t edge, uint64_t k) {
    const uint64_t a = d_mix64(seed ^ ((edge + 1) * 0x9E3779B97F4A7C15ull));
    return d_mix64(a ^ ((k + 1) * 0xD1B54A32D192ED03ull));
}

// one 72-byte rotation block into registers: four 16-byte loads + one 8-byte load (blocks are 8-byte aligned: 72 e bytes; gfx950
// serves unaligned 16-byte global loads) instead of nine 8-byte ones -- a gathered block costs the address pipeline one pass per load
// instruction and lane, so the setup kernels that gather two blocks per cycle (S0_long, DESC_PGD.m:129-147) issue 10 instead of 18
typedef double dbl2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void load_block9(const double* p, double* o) {
    const dbl2_a8* q = reinterpret_cast<const dbl2_a8*>(p);
    const dbl2_a8 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
    o[0] = v0.x; o[1] = v0.y; o[2] = v1.x; o[3] = v1.y; o[4] = v2.x; o[5] = v2.y; o[6] = v3.x; o[7] = v3.y; o[8] = p[8];
}

// Blocks b and b+8 share an XCD (round-robin dispatch, observed; speed only).
// Give each XCD a contiguous range of logical blocks so neighbouring edge
// segments -- which gather each other's cycle weights -- share one L2.
__device__ __forceinline__ int xcd_logical_block(int b, int nb) {
    return (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
}

}  // namespace desc
