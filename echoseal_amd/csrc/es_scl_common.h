// es_scl_common.h -- pieces shared by the list-decoder kernels (es_scl.hip: one frame per wave;
// es_scl_multi.hip: several frames per wave).
#ifndef ES_SCL_COMMON_H
#define ES_SCL_COMMON_H
#include "es_internal.h"
#include "es_math.h"

namespace {

constexpr int N = ES_POLAR_N;
constexpr int NLEV = 10;
constexpr int KINFO = ES_POLAR_K;         // 448 data positions (440 info + 8 CRC)

__device__ __forceinline__ void wave_fence_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_fence_global()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}


struct SclArgs {
    unsigned long long* dbg;
    const void* llr; int is_f64; long long B;
    es_frozen_mask frozen;
    const uint16_t* data_pos;
    const uint64_t* exp_tab;
    double* scratch;
    unsigned* slot_bits; int n_slots, slot_words;   // multi-frame kernel: bitmap of slab slots (one per resident block)
    uint8_t* hard_info; uint8_t* hard_ok;
    uint8_t* cand_info; double* cand_metric; uint8_t* cand_ok; int32_t* ncand;
    int skip_if_hard_ok;
    int lsz;                                  // the caller's list size (<= the kernel's template capacity L): paths kept per sort, row stride of the outputs
};

__device__ __forceinline__ uint64_t ptr_set(uint64_t p, int depth, int slot)
{
    const int sh = 6 * (depth - 1);
    return (p & ~(63ULL << sh)) | ((uint64_t)slot << sh);
}
__device__ __forceinline__ int ptr_get(uint64_t p, int depth) { return (int)((p >> (6 * (depth - 1))) & 63ULL); }

// lane l <-> lane l ^ S for a compile-time S, through DPP where the data-parallel primitives reach
// (S = 1, 2: quad_perm; S = 4, 8: a row shift each way and a select); otherwise ds_bpermute.
template <int S>
__device__ __forceinline__ int xor_lanes_b32(int v, int lane)
{
    if constexpr (S == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (S == 4 || S == 8) {
        const int up = __builtin_amdgcn_mov_dpp(v, 0x100 + S, 0xf, 0xf, true);             // row_shl:S  (lane l gets l+S)
        const int dn = __builtin_amdgcn_mov_dpp(v, 0x110 + S, 0xf, 0xf, true);             // row_shr:S  (lane l gets l-S)
        return (lane & S) ? dn : up;
    } else return __shfl_xor(v, S);
}
template <int S>
__device__ __forceinline__ double xor_lanes_f64(double x, int lane)
{
    uint64_t u; __builtin_memcpy(&u, &x, 8);
    const uint32_t lo = (uint32_t)xor_lanes_b32<S>((int)(uint32_t)u, lane);
    const uint32_t hi = (uint32_t)xor_lanes_b32<S>((int)(uint32_t)(u >> 32), lane);
    u = ((uint64_t)hi << 32) | lo;
    double r; __builtin_memcpy(&r, &u, 8); return r;
}

// S is a loop-unrolled constant at every call site, so the switch folds away.
__device__ __forceinline__ double xor_lanes_f64_sw(double x, int S, int lane)
{
    switch (S) {
        case 1: return xor_lanes_f64<1>(x, lane);
        case 2: return xor_lanes_f64<2>(x, lane);
        case 4: return xor_lanes_f64<4>(x, lane);
        case 8: return xor_lanes_f64<8>(x, lane);
        case 16: return xor_lanes_f64<16>(x, lane);
        default: return xor_lanes_f64<32>(x, lane);
    }
}

__device__ __forceinline__ uint8_t crc8_bytes(const uint8_t* b, int n)
{
    uint32_t reg = 0;
    for (int i = 0; i < n; ++i) {
        reg ^= b[i];
        #pragma unroll
        for (int k = 0; k < 8; ++k) reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
    }
    return (uint8_t)reg;
}


}  // namespace
#endif
