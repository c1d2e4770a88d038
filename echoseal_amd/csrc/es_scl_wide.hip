// es_scl_wide.hip -- SCL decoder for LARGE lists (L = 64, 128, 256), one workgroup per frame,
// one LANE per path.  Same arithmetic and the same bookkeeping idea as es_scl.hip (per-depth slot
// pointers instead of path copies, trace-back instead of per-path bit arrays, stable rank sort),
// but with L > 32 a frame no longer fits one wavefront:
//   * lane p owns path p and walks the elements of a tree node serially; the L paths advance in
//     lock step with a workgroup barrier after every depth (a path may read its parent's slot,
//     which another wave has just written);
//   * the LLR tree of all paths lives in an L2/Infinity-Cache resident scratch slab laid out
//     [element][slot] so that the lanes of a wave touch one contiguous row;
//   * slot-pointer tables (one byte per depth and path), partial-sum bit blocks and the candidate
//     metrics live in LDS; the tables are double-buffered and re-indexed by parent at each sort.
// The detector's default list size is 256 (rtwm/detector.py:27); this kernel is what lets
// WatermarkDetector(key).verify(...) run unchanged.  Values are bit-identical to the reference
// list decoder for the same reason as in es_scl.hip (es_math.h).
#include "es_internal.h"
#include "es_math.h"

namespace {

constexpr int N = ES_POLAR_N;
constexpr int NLEV = 10;
constexpr int KINFO = ES_POLAR_K;

struct WideArgs {
    const void* llr; int is_f64; long long B;
    es_frozen_mask frozen;
    const uint16_t* data_pos;
    const uint64_t* exp_tab;
    double* alpha;        // [slots][1024][L]
    uint16_t* tb;         // [slots][448][L]
    uint8_t* hard_info; uint8_t* hard_ok;
    uint8_t* cand_info; double* cand_metric; uint8_t* cand_ok; int32_t* ncand;
    int skip_if_hard_ok;
    int lsz;                                  // the caller's list size (<= L)
};

__device__ __forceinline__ uint8_t crc8_bytes_w(const uint8_t* b, int n)
{
    uint32_t reg = 0;
    for (int i = 0; i < n; ++i) {
        reg ^= b[i];
        #pragma unroll
        for (int k = 0; k < 8; ++k) reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
    }
    return (uint8_t)reg;
}

template <int L>
struct WideLds {
    uint64_t exp_tab[ES_EXP_TAB_WORDS];
    double   candm[2 * L];
    double   sp[2][L];                  // softplus pair of the even sibling, by slot
    uint32_t betaL[L][33];              // +1 pad: lanes hit different banks
    uint32_t curb[L][17];
    uint16_t sidx[2 * L];               // candidate index travelling with its metric through the sort
    uint16_t dpos[KINFO];
    uint8_t  ptrA[2][NLEV + 1][L];
    uint8_t  ptrB[2][NLEV + 1][L];
    uint32_t hardw[32];
    uint8_t  hbytes[56];
    int      flag;
};

template <int L>
__global__ __launch_bounds__(L) void es_scl_wide_kernel(WideArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    WideLds<L>& W = *reinterpret_cast<WideLds<L>*>(smem_raw);
    const int p = threadIdx.x;                       // path == lane
    for (int i = p; i < ES_EXP_TAB_WORDS; i += L) W.exp_tab[i] = a.exp_tab[i];
    for (int i = p; i < KINFO; i += L) W.dpos[i] = a.data_pos[i];
    __syncthreads();
    const uint64_t* tab = W.exp_tab;
    double* const A = a.alpha + (long long)blockIdx.x * N * L;       // element e of slot s at A[e*L + s]
    uint16_t* const TB = a.tb + (long long)blockIdx.x * KINFO * L;

    for (long long f = blockIdx.x; f < a.B; f += gridDim.x) {
        const float* llr32 = (const float*)a.llr + f * N;
        const double* llr64 = (const double*)a.llr + f * N;

        // ---------------- hard decision (fastpolar.py:260-268), first wave
        if (p < 64) {
            const int lane = p;
            uint32_t word = 0;
            for (int c = 0; c < 16; ++c) {
                const double v = a.is_f64 ? llr64[64 * c + lane] : (double)llr32[64 * c + lane];
                const unsigned long long m = __ballot(v > 0.0);
                if (((lane & 31) >> 1) == c) word = (lane & 1) ? (uint32_t)(m >> 32) : (uint32_t)m;
            }
            word ^= (word >> 1) & 0x55555555u;
            word ^= (word >> 2) & 0x33333333u;
            word ^= (word >> 4) & 0x0f0f0f0fu;
            word ^= (word >> 8) & 0x00ff00ffu;
            word ^= (word >> 16) & 0x0000ffffu;
            #pragma unroll
            for (int hw = 1; hw < 32; hw <<= 1) {
                const uint32_t o = __shfl_xor(word, hw);
                if (!((lane & 31) & hw)) word ^= o;
            }
            if (lane < 32) W.hardw[lane] = word;
        }
        __syncthreads();
        if (p < 56) {
            uint32_t byte = 0;
            #pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int pos = W.dpos[8 * p + b];
                byte |= ((W.hardw[pos >> 5] >> (pos & 31)) & 1u) << (7 - b);
            }
            W.hbytes[p] = (uint8_t)byte;
        }
        __syncthreads();
        if (p == 0) {
            const int ok = crc8_bytes_w(W.hbytes, ES_INFO_BYTES) == W.hbytes[ES_INFO_BYTES];
            W.flag = ok;
            a.hard_ok[f] = (uint8_t)ok;
        }
        if (p < ES_INFO_BYTES) a.hard_info[f * ES_INFO_BYTES + p] = W.hbytes[p];
        __syncthreads();
        if (W.flag && a.skip_if_hard_ok) {                 // no list for this record: its candidate rows read as zeros
            if (p == 0) a.ncand[f] = 0;
            for (int k = p; k < a.lsz * ES_INFO_BYTES; k += L) a.cand_info[f * a.lsz * ES_INFO_BYTES + k] = 0;
            if (p < a.lsz) { a.cand_metric[f * a.lsz + p] = 0.0; a.cand_ok[f * a.lsz + p] = 0; }
            __syncthreads();
            continue;
        }

        // ---------------- list decoding (fastpolar.py:278-330)
        int cur = 0;                                   // which pointer table is live
        for (int d = 0; d <= NLEV; ++d) { W.ptrA[0][d][p] = 0; W.ptrB[0][d][p] = 0; }
        double metric = 0.0;
        int cnt = 1, info_idx = 0;
        __syncthreads();

        for (int i = 0; i < N; ++i) {
            const int top = (i == 0) ? 1 : NLEV - __builtin_ctz((unsigned)i);
            const int sp_slot = W.ptrA[cur][NLEV][p];
            for (int d = top; d <= NLEV; ++d) {
                const int S = N >> d;
                const bool is_g = (i >> (NLEV - d)) & 1;
                const int ps = W.ptrA[cur][d - 1][p];
                const int bs = W.ptrB[cur][d][p];
                const double* par = A + (long long)(2 * S) * L + ps;          // depth d-1 block at elements [2S, 4S)
                double* dst = A + (long long)S * L + p;                        // depth d block at elements [S, 2S)
                auto ld_pair = [&](int j, double& pa, double& pb) {
                    if (d == 1) { pa = a.is_f64 ? llr64[j] : (double)llr32[j]; pb = a.is_f64 ? llr64[j + S] : (double)llr32[j + S]; }
                    else { pa = par[(long long)j * L]; pb = par[(long long)(j + S) * L]; }
                };
                if (is_g) {
                    int j = 0;
                    for (; j + 4 <= S; j += 4) {                             // four independent load pairs in flight
                        double pa[4], pb[4];
                        #pragma unroll
                        for (int u = 0; u < 4; ++u) ld_pair(j + u, pa[u], pb[u]);
                        #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t wbits = W.betaL[bs][(S + j + u) >> 5];
                            dst[(long long)(j + u) * L] = es_polar_g(pa[u], pb[u], (wbits >> ((S + j + u) & 31)) & 1u);
                        }
                    }
                    for (; j < S; ++j) {
                        double pa, pb; ld_pair(j, pa, pb);
                        const uint32_t wbits = W.betaL[bs][(S + j) >> 5];
                        dst[(long long)j * L] = es_polar_g(pa, pb, (wbits >> ((S + j) & 31)) & 1u);
                    }
                } else if (d == NLEV) {
                    const double pa = par[0], pb = par[(long long)L];
                    double sd, ss;
                    dst[0] = es_polar_f_sp(pa, pb, tab, &sd, &ss);
                    W.sp[0][p] = sd; W.sp[1][p] = ss;
                } else if (i == 0) {
                    // first chain: every path is still a copy of path 0, so the node is computed once, the lanes
                    // sharing its S elements, into slot 0, and every path points at it
                    for (int j = p; j < S; j += L) {
                        double pa, pb;
                        if (d == 1) { pa = a.is_f64 ? llr64[j] : (double)llr32[j]; pb = a.is_f64 ? llr64[j + S] : (double)llr32[j + S]; }
                        else { pa = A[(long long)(2 * S + j) * L]; pb = A[(long long)(2 * S + j + S) * L]; }
                        A[(long long)(S + j) * L] = es_polar_f(pa, pb, tab);
                    }
                } else {
                    double pa, pb;                                           // operands of element j+1 are loaded while f(j) runs
                    ld_pair(0, pa, pb);
                    for (int j = 0; j < S; ++j) {
                        double na = 0.0, nb = 0.0;
                        if (j + 1 < S) ld_pair(j + 1, na, nb);
                        dst[(long long)j * L] = es_polar_f(pa, pb, tab);
                        pa = na; pb = nb;
                    }
                }
                __threadfence_block();
                __syncthreads();
                W.ptrA[cur][d][p] = (uint8_t)((i == 0 && d != NLEV) ? 0 : p);
            }
            const double lam = A[(long long)1 * L + p];

            const bool frozen = (a.frozen.w[i >> 5] >> (i & 31)) & 1u;
            const double al = __builtin_fabs(lam);
            double lp;
            if (i & 1) {
                const uint32_t ub = (W.betaL[W.ptrB[cur][NLEV][p]][0] >> 1) & 1u;
                lp = W.sp[ub ? 0 : 1][sp_slot];
            } else {
                lp = es_softplus_neg(-al, tab);
            }
            const uint32_t pref = (lam >= 0.0) ? 1u : 0u;
            uint32_t bit = 0;
            if (frozen) {
                double pen = lp;
                if (pref != 0u) pen = lp + al;
                metric = metric + pen;
            } else {
                const double m0 = metric + ((pref != 0u) ? lp + al : lp);
                const double m1 = metric + ((pref != 1u) ? lp + al : lp);
                // the 2*cnt candidates (path order, bit 0 then bit 1) sorted by (metric, candidate index) = Python's
                // stable list.sort: bitonic network on (key, index) pairs in LDS, one compare-exchange per lane and
                // stage.  Stages whose partner distance stays inside a wave's 128 elements need only a wave fence.
                const bool live = p < cnt;
                const int nc = 2 * cnt;
                int nsort = 2; while (nsort < nc) nsort <<= 1;
                W.candm[2 * p] = live ? m0 : __builtin_inf(); W.candm[2 * p + 1] = live ? m1 : __builtin_inf();
                W.sidx[2 * p] = live ? (uint16_t)(2 * p) : (uint16_t)0xFFFF; W.sidx[2 * p + 1] = live ? (uint16_t)(2 * p + 1) : (uint16_t)0xFFFF;
                __syncthreads();
                for (int k = 2; k <= nsort; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        if (j >= 128) __syncthreads();
                        if (2 * p < nsort) {
                            const int lo = 2 * j * (p / j) + (p % j), hi = lo + j;
                            const double ka = W.candm[lo], kb = W.candm[hi];
                            const uint16_t ia = W.sidx[lo], ib = W.sidx[hi];
                            const bool a_first = (ka < kb) || (ka == kb && ia < ib);
                            const bool asc = (lo & k) == 0;
                            if (a_first != asc) { W.candm[lo] = kb; W.candm[hi] = ka; W.sidx[lo] = ib; W.sidx[hi] = ia; }
                        }
                        if (j >= 128) __syncthreads();
                        else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
                    }
                }
                __syncthreads();
                const int keep = nc < a.lsz ? nc : a.lsz;          // a.lsz <= L: lists of any size run on the next power of two's kernel
                const int myr = p < keep ? p : 0;
                const int myc = W.sidx[myr];
                const double mym = W.candm[myr];
                const int parent = myc >> 1;
                bit = (uint32_t)(myc & 1);
                metric = mym;
                const int nxt = cur ^ 1;
                #pragma unroll
                for (int d = 0; d <= NLEV; ++d) {
                    W.ptrA[nxt][d][p] = W.ptrA[cur][d][parent];
                    W.ptrB[nxt][d][p] = W.ptrB[cur][d][parent];
                }
                if (p < keep) TB[(long long)info_idx * L + p] = (uint16_t)myc;
                cur = nxt;
                cnt = keep;
                ++info_idx;
                __syncthreads();
            }

            // partial sums (fastpolar.py:156-183); each lane folds its own path
            const int t = __builtin_ctz(~(unsigned)i);
            if (t < NLEV) {
                uint32_t cw = bit;
                const int t5 = t < 5 ? t : 5;
                for (int s = 0; s < t5; ++s) {
                    const int S = 1 << s;
                    const int bs = W.ptrB[cur][NLEV - s][p];
                    const uint32_t left = (W.betaL[bs][0] >> S) & ((1u << S) - 1u);
                    cw = (left ^ cw) | (cw << S);
                }
                __syncthreads();                       // all reads of dword 0 done before anyone rewrites it
                if (t <= 5) {
                    if (t < 5) {
                        const int Sp = 1 << t;
                        const uint32_t mask = ((1u << Sp) - 1u) << Sp;
                        W.betaL[p][0] = (W.betaL[p][0] & ~mask) | (cw << Sp);
                    } else {
                        W.betaL[p][1] = cw;
                    }
                } else {
                    W.curb[p][0] = cw;
                    for (int s = 5; s < t; ++s) {
                        const int Wd = 1 << (s - 5);
                        const int bs = W.ptrB[cur][NLEV - s][p];
                        for (int w = 0; w < Wd; ++w) {
                            const uint32_t c0 = W.curb[p][w];
                            const uint32_t lf = W.betaL[bs][Wd + w];
                            W.curb[p][Wd + w] = c0;
                            W.curb[p][w] = c0 ^ lf;
                        }
                    }
                    __syncthreads();                   // every path has read the left blocks it needs
                    const int Wp = 1 << (t - 5);
                    for (int w = 0; w < Wp; ++w) W.betaL[p][Wp + w] = W.curb[p][w];
                }
                W.ptrB[cur][NLEV - t][p] = (uint8_t)p;
                __syncthreads();
            }
        }

        // ---------------- final ordering, trace-back, CRC
        W.candm[p] = metric;
        __syncthreads();
        int rank = 0;
        for (int k = 0; k < cnt; ++k) {
            const double mk = W.candm[k];
            rank += ((mk < metric) || (mk == metric && k < p)) ? 1 : 0;
        }
        if (p < cnt) {
            uint8_t* out = a.cand_info + (f * a.lsz + rank) * ES_INFO_BYTES;
            int curp = p;
            uint32_t acc = 0, reg = 0, last = 0;
            // trace back from the last information step; bytes come out last-to-first, so the CRC
            // (which runs first-to-last) is computed in a second pass over the stored bytes
            for (int tt = KINFO - 1; tt >= 0; --tt) {
                const uint32_t c = TB[(long long)tt * L + curp];
                acc |= (c & 1u) << (7 - (tt & 7));
                curp = (int)(c >> 1);
                if ((tt & 7) == 0) {
                    if ((tt >> 3) < ES_INFO_BYTES) out[tt >> 3] = (uint8_t)acc; else last = acc;
                    acc = 0;
                }
            }
            __threadfence_block();
            for (int k = 0; k < ES_INFO_BYTES; ++k) {
                reg ^= out[k];
                #pragma unroll
                for (int b = 0; b < 8; ++b) reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
            }
            a.cand_metric[f * a.lsz + rank] = metric;
            a.cand_ok[f * a.lsz + rank] = (uint8_t)(reg == last);
        }
        if (p == 0) a.ncand[f] = cnt;
        __syncthreads();
    }
}

template <int L>
int launch_wide(es_ctx* ctx, WideArgs a, int64_t B, hipStream_t st)
{
    const size_t lds = sizeof(WideLds<L>);
    if (!(ctx->wide_attr_mask & (unsigned)L)) {      // per context (= per device): the attribute belongs to the device's copy of the kernel
        ES_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&es_scl_wide_kernel<L>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->wide_attr_mask |= (unsigned)L;
    }
    long long blocks = B;
    const long long cap = ctx->wide_slots;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_scl_wide_kernel<L>, dim3((unsigned)blocks), dim3(L), lds, st, a);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

}  // namespace

// Scratch for the wide kernel: per resident workgroup 1024*L doubles + 448*L uint16.
size_t es_scl_wide_scratch_bytes(const es_ctx* ctx, int* slots_out)
{
    if (ctx->list_size_max <= 32) { *slots_out = 0; return 0; }
    const int slots = ctx->num_cu * 2;
    *slots_out = slots;
    const size_t Lm = (size_t)(es_list_cap(ctx->list_size_max) < 64 ? 64 : es_list_cap(ctx->list_size_max));
    return (size_t)slots * (N * Lm * sizeof(double) + KINFO * Lm * sizeof(uint16_t));
}

int es_launch_scl_wide(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
                       uint8_t* hard_info, uint8_t* hard_ok, uint8_t* cand_info, double* cand_metric,
                       uint8_t* cand_ok, int32_t* ncand, hipStream_t st)
{
    if (!ctx->d_wide_scratch) { ctx->err = "es_scl_batch: context was created with list_size_max <= 32"; return ES_EINVAL; }
    WideArgs a{};
    a.llr = llr; a.is_f64 = (dtype == ES_DTYPE_F64); a.B = B;
    a.frozen = ctx->frozen; a.data_pos = ctx->d_data_pos; a.exp_tab = ctx->d_exp_tab;
    a.alpha = reinterpret_cast<double*>(ctx->d_wide_scratch);
    a.tb = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(ctx->d_wide_scratch) +
                                       (size_t)ctx->wide_slots * N * (size_t)(es_list_cap(ctx->list_size_max) < 64 ? 64 : es_list_cap(ctx->list_size_max)) * sizeof(double));
    a.hard_info = hard_info; a.hard_ok = hard_ok; a.cand_info = cand_info;
    a.cand_metric = cand_metric; a.cand_ok = cand_ok; a.ncand = ncand;
    a.skip_if_hard_ok = skip_if_hard_ok;
    a.lsz = L;
    switch (L <= 64 ? 64 : (L <= 128 ? 128 : 256)) {
        case 64:  return launch_wide<64>(ctx, a, B, st);
        case 128: return launch_wide<128>(ctx, a, B, st);
        case 256: return launch_wide<256>(ctx, a, B, st);
        default: ctx->err = "wide list_size must be 64, 128 or 256"; return ES_EINVAL;
    }
}
