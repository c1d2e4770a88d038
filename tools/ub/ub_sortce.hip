// Checks the hand-written compare-exchange steps of es_scl_wide.hip against plain C++ on random (key, index) pairs with ties.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#define ES_CE_DPP(CTRL, lo, hi, ix, tmask)                                                                                   \
    do { uint32_t t_;                                                                                                        \
        asm volatile("s_nop 1\n\t"                                                                                           \
                     "v_sub_co_u32_dpp %3, vcc, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                              \
                     "v_subb_co_u32_dpp %3, vcc, %0, %0, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_subb_co_u32_dpp %3, vcc, %1, %1, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
                     "s_xor_b64 vcc, vcc, %4\n\t"                                                                            \
                     "v_cndmask_b32_dpp %2, %2, %2, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_cndmask_b32_dpp %0, %0, %0, vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_cndmask_b32_dpp %1, %1, %1, vcc " CTRL " row_mask:0xf bank_mask:0xf"                                 \
                     : "+v"(lo), "+v"(hi), "+v"(ix), "=&v"(t_) : "s"(tmask) : "vcc", "scc");                                 \
    } while (0)
__device__ __forceinline__ void ce_pair(uint32_t& lo, uint32_t& hi, uint32_t& ix, uint32_t olo, uint32_t ohi, uint32_t oix, unsigned long long tmask)
{
    uint32_t t_;
    asm volatile("v_sub_co_u32 %3, vcc, %2, %6\n\t"
                 "v_subb_co_u32 %3, vcc, %0, %4, vcc\n\t"
                 "v_subb_co_u32 %3, vcc, %1, %5, vcc\n\t"
                 "s_xnor_b64 vcc, vcc, %7\n\t"
                 "v_cndmask_b32 %2, %6, %2, vcc\n\t"
                 "v_cndmask_b32 %0, %4, %0, vcc\n\t"
                 "v_cndmask_b32 %1, %5, %1, vcc"
                 : "+v"(lo), "+v"(hi), "+v"(ix), "=&v"(t_) : "v"(olo), "v"(ohi), "v"(oix), "s"(tmask) : "vcc", "scc");
}
__device__ __forceinline__ void ce_inlane(uint32_t& lo0, uint32_t& hi0, uint32_t& ix0, uint32_t& lo1, uint32_t& hi1, uint32_t& ix1, unsigned long long amask)
{
    uint32_t t_, nlo0, nhi0, nix0;
    asm volatile("v_sub_co_u32 %3, vcc, %9, %6\n\t"
                 "v_subb_co_u32 %3, vcc, %7, %4, vcc\n\t"
                 "v_subb_co_u32 %3, vcc, %8, %5, vcc\n\t"
                 "s_xnor_b64 vcc, vcc, %10\n\t"
                 "v_cndmask_b32 %0, %4, %7, vcc\n\t"
                 "v_cndmask_b32 %1, %5, %8, vcc\n\t"
                 "v_cndmask_b32 %2, %6, %9, vcc\n\t"
                 "v_cndmask_b32 %7, %7, %4, vcc\n\t"
                 "v_cndmask_b32 %8, %8, %5, vcc\n\t"
                 "v_cndmask_b32 %9, %9, %6, vcc"
                 : "=&v"(nlo0), "=&v"(nhi0), "=&v"(nix0), "=&v"(t_), "+v"(lo0), "+v"(hi0), "+v"(ix0), "+v"(lo1), "+v"(hi1), "+v"(ix1)
                 : "s"(amask) : "vcc", "scc");
    lo0 = nlo0; hi0 = nhi0; ix0 = nix0;
}
__device__ bool before(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) { return ka < kb || (ka == kb && ia < ib); }
// mode 0: dpp dl=1, 1: dpp dl=2, 2: dpp dl=8 (row_ror), 3: ce_pair with shfl dl=4, 4: in-lane
__global__ void k(const uint64_t* key, const uint32_t* idx, int mode, uint64_t* okey, uint32_t* oidx, uint64_t* rkey, uint32_t* ridx)
{
    const int p = threadIdx.x;
    uint64_t k0 = key[2 * p], k1 = key[2 * p + 1]; uint32_t i0 = idx[2 * p], i1 = idx[2 * p + 1];
    const int dl = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 8 : 4;
    const bool tmin = ((p & dl) == 0) == (((p >> 4) & 1) == 0);
    // reference
    uint64_t r0 = k0, r1 = k1; uint32_t s0 = i0, s1 = i1;
    if (mode == 4) { const bool sw = before(k1, i1, k0, i0) == tmin; if (sw) { r0 = k1; s0 = i1; r1 = k0; s1 = i0; } }
    else {
        const uint64_t o0 = __shfl_xor(k0, dl), o1 = __shfl_xor(k1, dl); const uint32_t q0 = __shfl_xor(i0, dl), q1 = __shfl_xor(i1, dl);
        if (!(before(k0, i0, o0, q0) == tmin)) { r0 = o0; s0 = q0; }
        if (!(before(k1, i1, o1, q1) == tmin)) { r1 = o1; s1 = q1; }
    }
    rkey[2 * p] = r0; rkey[2 * p + 1] = r1; ridx[2 * p] = s0; ridx[2 * p + 1] = s1;
    uint32_t lo0 = (uint32_t)k0, hi0 = (uint32_t)(k0 >> 32), lo1 = (uint32_t)k1, hi1 = (uint32_t)(k1 >> 32);
    const unsigned long long tm = __builtin_amdgcn_ballot_w64(tmin);
    if (mode == 0) { ES_CE_DPP("quad_perm:[1,0,3,2]", lo0, hi0, i0, tm); ES_CE_DPP("quad_perm:[1,0,3,2]", lo1, hi1, i1, tm); }
    else if (mode == 1) { ES_CE_DPP("quad_perm:[2,3,0,1]", lo0, hi0, i0, tm); ES_CE_DPP("quad_perm:[2,3,0,1]", lo1, hi1, i1, tm); }
    else if (mode == 2) { ES_CE_DPP("row_ror:8", lo0, hi0, i0, tm); ES_CE_DPP("row_ror:8", lo1, hi1, i1, tm); }
    else if (mode == 3) {
        const uint32_t a0 = __shfl_xor(lo0, 4), b0 = __shfl_xor(hi0, 4), c0 = __shfl_xor(i0, 4), a1 = __shfl_xor(lo1, 4), b1 = __shfl_xor(hi1, 4), c1 = __shfl_xor(i1, 4);
        ce_pair(lo0, hi0, i0, a0, b0, c0, tm); ce_pair(lo1, hi1, i1, a1, b1, c1, tm);
    } else ce_inlane(lo0, hi0, i0, lo1, hi1, i1, tm);
    okey[2 * p] = ((uint64_t)hi0 << 32) | lo0; okey[2 * p + 1] = ((uint64_t)hi1 << 32) | lo1; oidx[2 * p] = i0; oidx[2 * p + 1] = i1;
}
int main()
{
    std::mt19937_64 rng(1);
    std::vector<uint64_t> key(128), ok(128), rk(128); std::vector<uint32_t> idx(128), oi(128), ri(128);
    uint64_t *dk, *dok, *drk; uint32_t *di, *doi, *dri;
    hipMalloc(&dk, 1024); hipMalloc(&dok, 1024); hipMalloc(&drk, 1024); hipMalloc(&di, 512); hipMalloc(&doi, 512); hipMalloc(&dri, 512);
    int bad = 0; int badm[5] = {0, 0, 0, 0, 0};
    for (int trial = 0; trial < 2000; ++trial) {
        for (int e = 0; e < 128; ++e) {
            const int kind = rng() % 4;
            key[e] = kind == 0 ? 0x7ff0000000000000ULL : kind == 1 ? (0x4000000000000000ULL + (rng() % 3)) : kind == 2 ? (rng() >> 2) : ((rng() >> 2) & 0xffffffff00000000ULL) | (rng() % 2 ? 0xffffffffu : 0u);
            idx[e] = kind == 0 ? 0xFFFFu : (uint32_t)e;
        }
        hipMemcpy(dk, key.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), 512, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 5; ++mode) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dk, di, mode, dok, doi, drk, dri);
            hipMemcpy(ok.data(), dok, 1024, hipMemcpyDeviceToHost); hipMemcpy(oi.data(), doi, 512, hipMemcpyDeviceToHost);
            hipMemcpy(rk.data(), drk, 1024, hipMemcpyDeviceToHost); hipMemcpy(ri.data(), dri, 512, hipMemcpyDeviceToHost);
            for (int e = 0; e < 128; ++e) if (ok[e] != rk[e] || oi[e] != ri[e]) { if (bad < 10) printf("mode %d trial %d elem %d: got %llx/%u want %llx/%u\n", mode, trial, e, (unsigned long long)ok[e], oi[e], (unsigned long long)rk[e], ri[e]); ++bad; ++badm[mode]; }
        }
    }
    printf("mismatches: %d  by mode: %d %d %d %d %d\n", bad, badm[0], badm[1], badm[2], badm[3], badm[4]);
    return bad != 0;
}
