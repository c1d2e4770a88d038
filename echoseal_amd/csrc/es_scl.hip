// es_scl.hip -- Polar(1024,448)+CRC-8 hard-decision shortcut and successive-cancellation LIST
// decoder for gfx950, one 64-lane wavefront per frame.
//
// What it computes is PolarCode.decode of the reference (rtwm/fastpolar.py:254-359) up to, but
// not including, the validator callback: the hard-decision candidate with its CRC flag
// (:260-268) and the L list survivors in ascending path-metric order with metrics and CRC flags
// (:278-341).  The host applies validator / selection rules (:268-276, :335-359) to that list.
//
// Mapping to the hardware (no dense contraction here, so no MFMA; the work is float64 VALU and
// cross-lane bookkeeping):
//   * 64 lanes = L paths x P lanes (P = 64/L).  All paths of a frame advance in lock step, so the
//     whole decoder is wave-synchronous: no __syncthreads in the bit loop, only wave fences.
//   * LLR tree: depth d holds 1024>>d values per path.  Depths 1..3 (512+256+128 values, touched
//     2/4/8 times per decode) live in an HBM/L2 scratch slab owned by the wave; depths 4..10
//     (127 values) live in LDS.  Depth 0 (channel LLRs) is read straight from the input.
//   * No path copies.  Every path owns one storage slot per depth; a path remembers, per depth,
//     WHICH slot holds its data (6 bits per depth packed in a 64-bit register, one word for the
//     LLR tree and one for the partial-sum tree).  A list sort only permutes those two words and
//     the metric with ds_bpermute (__shfl).  A depth that is recomputed is always written to the
//     path's own slot; all paths recompute the same depths at the same time, so a slot is never
//     overwritten while another path still needs it.
//   * Decided bits are not stored per path; each information step records (parent, bit) per
//     survivor and the 448 data bits are recovered by a trace-back at the end.
//   * Path-metric sort: candidates 2p+b sit in lanes (p, q=b); rank = number of candidates that
//     sort strictly before (metric, then candidate index) -- exactly Python's stable list.sort.
//   * float64 f / penalty use es_math.h (glibc-exact exp/log1p) with the 2 KB exp table in LDS.
//
// Build with -ffp-contract=off: every rounding step in es_math.h is explicit.
#include "es_scl_common.h"
#ifdef ES_SCL_STAMPS
#include <cstdio>
#endif

namespace {

#ifndef ES_SCL_FCHAINS
#define ES_SCL_FCHAINS 4                            /* independent f evaluations in flight in the slot-storage levels (2: 1.5 % slower at 1 024 frames) */
#endif
constexpr int GDEPTH = 3;                 // depths 1..GDEPTH live in global scratch
constexpr int GSLOT = 512 + 256 + 128;    // doubles per path slot in global scratch
constexpr int LDS_ROW = 136;              // 128 doubles + 8 pad (bank spread between slots)

template <int L>
struct SclWave {
    double   alphaS[L][LDS_ROW];          // depth d (4..10): values at [1024>>d, 2*(1024>>d))
    double   candm[2 * L];
    uint32_t betaL[L][32];                // left-sibling partial sums, block of S bits at bit S
    uint32_t curb[L][16];                 // transient right block while folding upward
    uint32_t hardw[32];
    uint8_t  tb[KINFO][L];                // trace-back: (parent << 1) | bit
    uint8_t  sel[L < 4 ? 4 : L];
    uint8_t  outb[L][56];
};

template <int L> struct SclCfg {
    static constexpr int WPB = (L <= 8) ? 4 : (L == 16 ? 2 : 1);
    // LDS admits two blocks per CU; with four waves per block that is two waves per SIMD, which the register
    // allocation must allow (256 VGPRs).  Longer lists run one wave per SIMD at most and may use more.
    static constexpr int MIN_WAVES = (L <= 8) ? 2 : 1;
};

#ifdef ES_SCL_STAMPS
#define ES_STAMP(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); (acc) += _t - t_last; t_last = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ES_STAMP(acc) do { } while (0)
#endif

template <int L>
__global__ __launch_bounds__(64 * SclCfg<L>::WPB, SclCfg<L>::MIN_WAVES) void es_scl_kernel(SclArgs a)
{
    constexpr int WPB = SclCfg<L>::WPB;
    constexpr int P = 64 / L;
    constexpr int LGP = (P == 64) ? 6 : (P == 32) ? 5 : (P == 16) ? 4 : (P == 8) ? 3 : (P == 4) ? 2 : 1;
    constexpr int RD = NLEV - LGP;      // depths RD..10 (block sizes P..1) live in registers
    __shared__ __attribute__((aligned(16))) uint64_t s_exp[ES_EXP_TAB_WORDS];
    __shared__ uint16_t s_dpos[KINFO];
    __shared__ SclWave<L> s_wave[WPB];

    for (int i = threadIdx.x; i < ES_EXP_TAB_WORDS; i += blockDim.x) s_exp[i] = a.exp_tab[i];
    for (int i = threadIdx.x; i < KINFO; i += blockDim.x) s_dpos[i] = a.data_pos[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int path = lane / P;
    const int q = lane % P;
    SclWave<L>& W = s_wave[wv];
    const long long wave_id = (long long)blockIdx.x * WPB + wv;
    const long long n_waves = (long long)gridDim.x * WPB;
    double* const scr = a.scratch + wave_id * (long long)(L * GSLOT);
    const uint64_t* const tab = s_exp;

    for (long long f = wave_id; f < a.B; f += n_waves) {
        const float* llr32 = (const float*)a.llr + f * N;
        const double* llr64 = (const double*)a.llr + f * N;

        // ---------------- hard decision -> butterfly -> data bits -> CRC (fastpolar.py:260-268)
        {
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
            wave_fence_lds();
            if (lane < 56) {
                uint32_t byte = 0;
                #pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int pos = s_dpos[8 * lane + b];
                    byte |= ((W.hardw[pos >> 5] >> (pos & 31)) & 1u) << (7 - b);
                }
                W.outb[0][lane] = (uint8_t)byte;
            }
            wave_fence_lds();
            int ok = 0;
            if (lane == 0) ok = (crc8_bytes(W.outb[0], ES_INFO_BYTES) == W.outb[0][ES_INFO_BYTES]);
            ok = __shfl(ok, 0);
            if (lane < ES_INFO_BYTES) a.hard_info[f * ES_INFO_BYTES + lane] = W.outb[0][lane];
            if (lane == 0) a.hard_ok[f] = (uint8_t)ok;
            wave_fence_lds();
            if (ok && a.skip_if_hard_ok) {                 // no list for this record: its candidate rows read as zeros
                if (lane == 0) a.ncand[f] = 0;
                for (int k = lane; k < a.lsz * ES_INFO_BYTES; k += 64) a.cand_info[f * a.lsz * ES_INFO_BYTES + k] = 0;
                if (lane < a.lsz) { a.cand_metric[f * a.lsz + lane] = 0.0; a.cand_ok[f * a.lsz + lane] = 0; }
                continue;
            }
        }

        // ---------------- list decoding (fastpolar.py:278-330)
        uint64_t ptrA = 0, ptrB = 0;      // every path starts as a mirror of path 0
        double metric = 0.0;
        // Bottom of the LLR tree in registers: ar[k] is element (q mod (P>>k)) of this path's open
        // node at depth RD+k, replicated over the path's P lanes.  sp_* = softplus pair of the even leaf.
        double ar[LGP + 1];
        #pragma unroll
        for (int k = 0; k <= LGP; ++k) ar[k] = 0.0;
        double sp_diff = 0.0, sp_sum = 0.0, lp_odd = 0.0;
        // partial-sum blocks of depths 6..10 (sizes 16..1, block of S bits at bit S): one dword per path,
        // carried by value and moved together with ptrB at a sort; deeper words stay in LDS (betaL).
        uint32_t b0 = 0;
        uint32_t frozen_word = 0;
        int cnt = 1;                      // live paths
        int info_idx = 0;
        if (lane < 32) { for (int s = 0; s < L; ++s) W.betaL[s][lane] = 0; }
        wave_fence_lds();

#ifdef ES_SCL_STAMPS
        unsigned long long t_big = 0, t_small = 0, t_dec_f = 0, t_dec_i = 0, t_beta = 0, t_misc = 0;
        unsigned long long t_s1 = 0, t_s2 = 0, t_s3 = 0, t_s4 = 0;
        unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
        for (int i = 0; i < N; ++i) {
            if ((i & 31) == 0) frozen_word = a.frozen.w[i >> 5];      // one scalar load per 32 bits
            // --- all-frozen aligned block starting here?  (rate-0 node: no decisions inside it, so its
            // leaf LLRs can be produced level by level with every lane busy instead of leaf by leaf)
            int blk = 0;                                              // log2 of the block size, 0 = none
            if ((i & 1) == 0 && P >= 2 && i != 0) {       // (the first chain lives in the shared slot 0: leaf-serial there)
                const uint32_t fw = frozen_word >> (i & 31);
                #pragma unroll
                for (int t = 1; t <= 5; ++t) {
                    const int S = 1 << t;
                    const uint32_t m = (S == 32) ? 0xffffffffu : ((1u << S) - 1u);
                    if (blk == t - 1 && (i & (S - 1)) == 0 && (fw & m) == m) blk = t;
                }
                if (blk == 5 && (i & 63) == 0 && a.frozen.w[(i >> 5) + 1] == 0xffffffffu) blk = 6;
            }
            const int d_stop = blk ? NLEV - blk : NLEV;               // deepest depth the leaf-serial chain computes
            // --- LLR chain: recompute the depths that changed since leaf i-1 (fastpolar.py:127-154)
            const int top = (i == 0) ? 1 : NLEV - __builtin_ctz((unsigned)i);
            // (a) depths above the register-resident part: slot storage in scratch / LDS
            for (int d = top; d < RD && d <= d_stop; ++d) {
                const int S = N >> d;
                const bool is_g = (i >> (NLEV - d)) & 1;
                const int ps = (d > 1) ? ptr_get(ptrA, d - 1) : 0;
                const int bs = ptr_get(ptrB, d);
                // parent block (depth d-1, 2S values) and destination block (depth d, S values)
                const double* par_g = scr + ps * GSLOT + (N - 4 * S);          // depth d-1 in scratch: 0,512,768
                const double* par_l = &W.alphaS[ps][2 * S];
                // First chain (i == 0): every path is still a mirror of path 0, so the node is computed ONCE,
                // by all 64 lanes, into slot 0, and every path points at it; a path's own slot takes over at
                // the next recomputation of this depth, which all paths do together.
                const bool first = (i == 0);
                const int own = first ? 0 : path;
                const int j0 = first ? lane : q;
                const int jst = first ? 64 : P;
                double* dst_g = scr + own * GSLOT + (N - 2 * S);
                double* dst_l = &W.alphaS[own][S];
                // Loaders and stores are chosen OUTSIDE the element loops (channel LLRs / scratch / LDS): with the choice inside, control flow
                // sits between a load and its use and the compiler waits for everything outstanding after every load -- the g loops of the
                // top depths (pure load latency at one wave per SIMD) ran one memory round trip per element pair.
                auto run_level = [&](auto load_pair, auto store_out) {
                    if (is_g) {
                        int j = q;
                        for (; j + 3 * P < S; j += 4 * P) {                       // four independent pairs in flight
                            double xa[4], xb[4];
                            #pragma unroll
                            for (int v = 0; v < 4; ++v) load_pair(j + v * P, xa[v], xb[v]);
                            #pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int jj = j + v * P;
                                const uint32_t wbits = (S <= 16) ? b0 : W.betaL[bs][(S + jj) >> 5];
                                store_out(jj, es_polar_g(xa[v], xb[v], (wbits >> ((S + jj) & 31)) & 1u));
                            }
                        }
                        for (; j < S; j += P) {
                            double pa, pb; load_pair(j, pa, pb);
                            const uint32_t wbits = (S <= 16) ? b0 : W.betaL[bs][(S + j) >> 5];
                            store_out(j, es_polar_g(pa, pb, (wbits >> ((S + j) & 31)) & 1u));
                        }
                    } else {
                        int j = j0;
                        if constexpr (ES_SCL_FCHAINS == 4 && L >= 8)   // (L = 1: four chains are 3.6 % slower; L = 4: no difference; L = 8: 1.3 % faster)
                        for (; j + 3 * jst < S; j += 4 * jst) {    // four independent f chains in flight (one memory round trip per four elements)
                            double a0, c0, a1, c1, a2, c2, a3, c3;
                            load_pair(j, a0, c0); load_pair(j + jst, a1, c1); load_pair(j + 2 * jst, a2, c2); load_pair(j + 3 * jst, a3, c3);
                            const double o0 = es_polar_f(a0, c0, tab);
                            const double o1 = es_polar_f(a1, c1, tab);
                            const double o2 = es_polar_f(a2, c2, tab);
                            const double o3 = es_polar_f(a3, c3, tab);
                            store_out(j, o0); store_out(j + jst, o1); store_out(j + 2 * jst, o2); store_out(j + 3 * jst, o3);
                        }
                        for (; j + jst < S; j += 2 * jst) {        // two independent f chains in flight
                            double a0, c0, a1, c1; load_pair(j, a0, c0); load_pair(j + jst, a1, c1);
                            const double o0 = es_polar_f(a0, c0, tab);
                            const double o1 = es_polar_f(a1, c1, tab);
                            store_out(j, o0); store_out(j + jst, o1);
                        }
                        if (j < S) { double pa, pb; load_pair(j, pa, pb); store_out(j, es_polar_f(pa, pb, tab)); }
                    }
                };
                auto st_g = [&](int j, double v) { dst_g[j] = v; };
                auto st_l = [&](int j, double v) { dst_l[j] = v; };
                auto ld_g = [&](int j, double& pa, double& pb) { pa = par_g[j]; pb = par_g[j + S]; };
                auto ld_l = [&](int j, double& pa, double& pb) { pa = par_l[j]; pb = par_l[j + S]; };
                if (d == 1) {                                                  // the channel LLRs (depth 1 lives in the scratch slab)
                    if (a.is_f64) run_level([&](int j, double& pa, double& pb) { pa = llr64[j]; pb = llr64[j + S]; }, st_g);
                    else run_level([&](int j, double& pa, double& pb) { pa = (double)llr32[j]; pb = (double)llr32[j + S]; }, st_g);
                } else if (d <= GDEPTH) run_level(ld_g, st_g);
                else if (d - 1 <= GDEPTH) run_level(ld_g, st_l);
                else run_level(ld_l, st_l);
                if (d <= GDEPTH) wave_fence_global(); else wave_fence_lds();
                ptrA = ptr_set(ptrA, d, own);
#ifdef ES_SCL_STAMPS
                ES_STAMP(t_big);
#endif
            }
            // (b) depth RD: one element per lane, straight from the slot of depth RD-1 into a register
            if (RD >= top && RD <= d_stop) {
                const int d = RD;
                const bool is_g = (i >> (NLEV - d)) & 1;
                double pa, pb;
                if (d == 1) {
                    if (a.is_f64) { pa = llr64[q]; pb = llr64[q + P]; }
                    else { pa = (double)llr32[q]; pb = (double)llr32[q + P]; }
                } else if (d - 1 <= GDEPTH) {
                    const double* par = scr + ptr_get(ptrA, d - 1) * GSLOT + (N - 4 * P);
                    pa = par[q]; pb = par[q + P];
                } else {
                    const double* par = &W.alphaS[ptr_get(ptrA, d - 1)][2 * P];
                    pa = par[q]; pb = par[q + P];
                }
                if (is_g) {
                    const uint32_t wbits = (P <= 16) ? b0 : W.betaL[ptr_get(ptrB, d)][(P + q) >> 5];
                    ar[0] = es_polar_g(pa, pb, (wbits >> ((P + q) & 31)) & 1u);
                } else {
                    ar[0] = es_polar_f(pa, pb, tab);
                }
            }
            // (c) depths RD+1..10: register to register.  The node has S = P>>k elements, so the two
            // softplus terms of f(a,b) go to two different lanes (lane bit S selects |a-b| or |a+b|)
            // and are exchanged with one xor-shuffle: one softplus stream instead of two.
            #pragma unroll
            for (int k = 1; k <= LGP; ++k) {
                const int d = RD + k;
                if (d >= top && d <= d_stop) {
                    const int S = P >> k;
                    const bool is_g = (i >> (NLEV - d)) & 1;
                    const double own = ar[k - 1];
                    const double oth = xor_lanes_f64_sw(own, S, lane);
                    const bool hi = (q & S) != 0;                    // this lane holds parent[j+S]
                    const double pa = hi ? oth : own;                // parent[j]
                    const double pb = hi ? own : oth;                // parent[j+S]
                    if (is_g) {
                        const int j = q & (S - 1);
                        const uint32_t wbits = (S <= 16) ? b0 : W.betaL[ptr_get(ptrB, d)][(S + j) >> 5];
                        ar[k] = es_polar_g(pa, pb, (wbits >> ((S + j) & 31)) & 1u);
                    } else {
                        const double sum = pa + pb;
                        const double d1 = pa - pb, d2 = 0.0 - sum;
                        const bool pos1 = d1 > 0, pos2 = d2 > 0;
                        const double t1 = pos1 ? -d1 : d1, t2 = pos2 ? -d2 : d2;
                        const double mine = es_softplus_neg(hi ? t2 : t1, tab);
                        const double theirs = xor_lanes_f64_sw(mine, S, lane);
                        const double L1 = hi ? theirs : mine;         // log1p(exp(-|a-b|))
                        const double L2 = hi ? mine : theirs;         // log1p(exp(-|a+b|))
                        double r1 = (pos1 ? pa : pb) + L1;
                        if (pa == pb) r1 = pa + ES_LOGE2;
                        double r2 = (pos2 ? 0.0 : sum) + L2;
                        if (0.0 == sum) r2 = 0.0 + ES_LOGE2;
                        ar[k] = r1 - r2;
                        if (k == LGP) { sp_diff = L1; sp_sum = L2; }  // penalties of the odd sibling
                    }
                }
            }
#ifdef ES_SCL_STAMPS
            ES_STAMP(t_small);
#endif
            uint32_t bit = 0;
            if (blk) {
                // ---------------- rate-0 block of S = 2^blk leaves (all bits 0: every g is b + a)
                const int S = 1 << blk;
                double* const X = &W.alphaS[path][S];                 // the node's S values (own slot), reused in place
                if (S > P) {
                    for (int h = S >> 1; h >= P; h >>= 1) {           // nodes of 2h values -> children of h values
                        const int lh = 31 - __builtin_clz((unsigned)h);
                        int idx = q;
                        for (; idx + P < (S >> 1); idx += 2 * P) {    // two independent f chains in flight
                            const int e0 = ((idx >> lh) << (lh + 1)) + (idx & (h - 1));
                            const int e1 = (((idx + P) >> lh) << (lh + 1)) + ((idx + P) & (h - 1));
                            const double a0 = X[e0], c0 = X[e0 + h], a1 = X[e1], c1 = X[e1 + h];
                            const double o0 = es_polar_f(a0, c0, tab);
                            const double o1 = es_polar_f(a1, c1, tab);
                            X[e0] = o0; X[e0 + h] = es_polar_g(a0, c0, 0u);
                            X[e1] = o1; X[e1 + h] = es_polar_g(a1, c1, 0u);
                        }
                        if (idx < (S >> 1)) {
                            const int e0 = ((idx >> lh) << (lh + 1)) + (idx & (h - 1));
                            const double a0 = X[e0], c0 = X[e0 + h];
                            X[e0] = es_polar_f(a0, c0, tab); X[e0 + h] = es_polar_g(a0, c0, 0u);
                        }
                        wave_fence_lds();
                    }
                }
                // sub-blocks of P values, one per lane: LGP more levels in registers (lane-split softplus as
                // in (c)), then the P leaf penalties added to the metric in leaf order (fastpolar.py:281-286)
                // (a block smaller than P sits in ar[LGP - blk], replicated, and starts further down)
                const int k0 = (blk >= LGP) ? 0 : LGP - blk;
                const int nsub = (S > P) ? S / P : 1;
                const int nleaf = (S < P) ? S : P;
                for (int sb = 0; sb < nsub; ++sb) {
                    double x = ar[0];
                    if (S > P) x = X[sb * P + q];
                    #pragma unroll
                    for (int k = 1; k <= LGP; ++k) if (k == k0) x = ar[k];
                    double L2last = 0.0;
                    #pragma unroll
                    for (int k = 1; k <= LGP; ++k) {
                        if (k <= k0) continue;
                        const int h = P >> k;
                        const double oth = xor_lanes_f64_sw(x, h, lane);
                        const bool hi = (q & h) != 0;
                        const double pa = hi ? oth : x, pb = hi ? x : oth;
                        const double sum = pa + pb;
                        const double d1 = pa - pb, d2 = 0.0 - sum;
                        const bool pos1 = d1 > 0, pos2 = d2 > 0;
                        const double t1 = pos1 ? -d1 : d1, t2 = pos2 ? -d2 : d2;
                        const double mine = es_softplus_neg(hi ? t2 : t1, tab);
                        const double theirs = xor_lanes_f64_sw(mine, h, lane);
                        const double L1 = hi ? theirs : mine;
                        const double L2 = hi ? mine : theirs;
                        double r1 = (pos1 ? pa : pb) + L1;
                        if (pa == pb) r1 = pa + ES_LOGE2;
                        double r2 = (pos2 ? 0.0 : sum) + L2;
                        if (0.0 == sum) r2 = 0.0 + ES_LOGE2;
                        x = hi ? es_polar_g(pa, pb, 0u) : (r1 - r2);
                        L2last = L2;
                    }
                    const double al = __builtin_fabs(x);
                    // an odd leaf's penalty term is the log1p(exp(-|a+b|)) its even sibling's f just used
                    const double lp = es_softplus_neg(-al, tab);
                    double pen = (P >= 2 && (q & 1)) ? L2last : lp;
                    if (x >= 0.0) pen = pen + al;
                    #pragma unroll 8
                    for (int k = 0; k < P; ++k) if (k < nleaf) metric = metric + __shfl(pen, path * P + k);
                }
                // partial sums of the block are all zero: clear the interior levels, then let the ordinary
                // upward fold run from the block's last leaf
                if (S >= 32) b0 = 0; else b0 &= ~((1u << S) - 1u);
                for (int sl = 5; sl < blk; ++sl) {
                    const int Wd = 1 << (sl - 5);
                    for (int w = q; w < Wd; w += P) W.betaL[path][Wd + w] = 0;
                    ptrB = ptr_set(ptrB, NLEV - sl, path);
                }
                wave_fence_lds();
                i += S - 1;
                ES_STAMP(t_misc);
            } else {
            const double lam = ar[LGP];

            // --- decision
            const bool frozen = (frozen_word >> (i & 31)) & 1u;
            const double al = __builtin_fabs(lam);
            double lp;
            if (i & 1) {
                lp = lp_odd;                                         // set when the even sibling was decided
            } else {
                lp = es_softplus_neg(-al, tab);
            }
            const uint32_t pref = (lam >= 0.0) ? 1u : 0u;
            if (frozen) {                                             // fastpolar.py:281-286
                double pen = lp;
                if (pref != 0u) pen = lp + al;
                metric = metric + pen;
                lp_odd = sp_sum;                                      // sibling g = b + a when this bit is 0
            } else {                                                  // fastpolar.py:288-330
                double pen = lp;
                if ((uint32_t)q != pref) pen = lp + al;
                const double m = metric + pen;
                // stable rank of candidate c = 2*path + q among the 2*cnt live candidates
                // (metrics staged in LDS; broadcast reads).  A v_readlane variant was measured slower.
                const bool is_cand = (q < 2) && (path < cnt);
                const int c = 2 * path + q;
                ES_STAMP(t_s1);
                if (is_cand) W.candm[c] = m;
                wave_fence_lds();
                const int nc = 2 * cnt;
                // The 2L x 2L comparison matrix is split over the P lanes of a path: candidate b = q & 1
                // is held by the P/2 lanes with that parity, each of which ranks it against a slice of
                // the candidates; the partial ranks are then summed across those lanes.
                constexpr int G = (P >= 2) ? P / 2 : 1;              // lanes sharing one candidate
                constexpr int SPAN = (2 * L + G - 1) / G;            // candidates each of them compares against
                const int cb = q & 1;
                double mc;                                           // metric of candidate 2*path + cb
                if constexpr (P == 8) {
                    // lane (path, q) wants lane (path, q & 1): inside a quad quad_perm [0,1,0,1]; the upper quad of the
                    // path then takes the lower quad's copy (row_shr:4) -- DPP moves instead of an LDS-pipe permute
                    uint64_t u; __builtin_memcpy(&u, &m, 8);
                    int lo = (int)(uint32_t)u, hi = (int)(uint32_t)(u >> 32);
                    lo = __builtin_amdgcn_mov_dpp(lo, 0x44, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x44, 0xf, 0xf, true);
                    const int lo2 = __builtin_amdgcn_mov_dpp(lo, 0x114, 0xf, 0xf, true), hi2 = __builtin_amdgcn_mov_dpp(hi, 0x114, 0xf, 0xf, true);
                    if (q & 4) { lo = lo2; hi = hi2; }
                    u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
                    __builtin_memcpy(&mc, &u, 8);
                } else mc = __shfl(m, path * P + cb);
                const int cc_ = 2 * path + cb;
                const int k0 = (q >> 1) * SPAN;
                int rank = 0;
                #pragma unroll
                for (int u = 0; u < SPAN; ++u) {
                    const int k = k0 + u;
                    const double mk = W.candm[k < 2 * L ? k : 0];
                    rank += ((k < nc) && ((mk < mc) || (mk == mc && k < cc_))) ? 1 : 0;
                }
                if constexpr (P >= 4) rank += xor_lanes_b32<2>(rank, lane);
                if constexpr (P >= 8) rank += xor_lanes_b32<4>(rank, lane);
                if constexpr (P >= 16) rank += xor_lanes_b32<8>(rank, lane);
                if constexpr (P >= 32) rank += xor_lanes_b32<16>(rank, lane);
                if constexpr (P >= 64) rank += xor_lanes_b32<32>(rank, lane);
                ES_STAMP(t_s2);
                const int keep = nc < a.lsz ? nc : a.lsz;          // a.lsz <= L: lists of any size run on the next power of two's kernel
                // new path r continues the candidate of rank r; `src` = lane holding that candidate
                // (parent * P + bit).  Dead paths mirror rank 0.
                // (an inverse permutation through per-rank ballots and a v_readlane rank loop were both
                // measured slower than this LDS scatter / gather)
                if (is_cand && rank < keep) W.sel[rank] = (uint8_t)c;
                wave_fence_lds();
                const int cc = W.sel[path < keep ? path : 0];
                const int src = (cc >> 1) * P + (cc & 1);
                ES_STAMP(t_s3);
                const int parent = src / P;
                bit = (uint32_t)(src % P);
                const int myc = 2 * parent + (int)bit;
                metric = __shfl(m, src);
                ptrA = __shfl(ptrA, parent * P);
                ptrB = __shfl((ptrB & 0xffffffffULL) | ((uint64_t)b0 << 32), parent * P);
                b0 = (uint32_t)(ptrB >> 32);
                ptrB &= 0xffffffffULL;
                // register-resident nodes follow their path; a depth whose subtree is complete after
                // this leaf is dead and need not move
                #pragma unroll
                for (int k = 0; k <= LGP; ++k)
                    if (((i + 1) & ((1 << (LGP - k)) - 1)) != 0) ar[k] = __shfl(ar[k], parent * P + q);
                // penalty of the odd sibling: parent lane 0 offers log1p(exp(-|b+a|)) (bit 0), lane 1
                // log1p(exp(-|b-a|)) (bit 1); one shuffle fetches the right one
                if (!(i & 1)) lp_odd = __shfl((q & 1) ? sp_diff : sp_sum, src);
                if (q == 0 && path < keep) W.tb[info_idx][path] = (uint8_t)myc;
                cnt = keep;
                ++info_idx;
                wave_fence_lds();
                ES_STAMP(t_s4);
            }

#ifdef ES_SCL_STAMPS
            if (frozen) ES_STAMP(t_dec_f); else ES_STAMP(t_dec_i);
#endif
            }
            // --- partial sums: fold upward while the node is a right child (fastpolar.py:156-183)
            const int t = __builtin_ctz(~(unsigned)i);                // trailing ones of i
            if (t < NLEV) {
                uint32_t cur = bit;
                const int t5 = t < 5 ? t : 5;
                for (int s = 0; s < t5; ++s) {
                    const int S = 1 << s;
                    const uint32_t left = (b0 >> S) & ((1u << S) - 1u);
                    cur = (left ^ cur) | (cur << S);
                }
                if (t <= 5) {
                    if (t < 5) {
                        const int Sp = 1 << t;
                        const uint32_t mask = ((1u << Sp) - 1u) << Sp;
                        b0 = (b0 & ~mask) | (cur << Sp);
                    } else if (q == 0) {
                        W.betaL[path][1] = cur;
                    }
                } else {
                    if (q == 0) W.curb[path][0] = cur;
                    wave_fence_lds();
                    for (int s = 5; s < t; ++s) {
                        const int Wd = 1 << (s - 5);                  // words in the current block
                        const int bs = ptr_get(ptrB, NLEV - s);
                        for (int w = q; w < Wd; w += P) {
                            const uint32_t c0 = W.curb[path][w];
                            const uint32_t lf = W.betaL[bs][Wd + w];
                            W.curb[path][Wd + w] = c0;
                            W.curb[path][w] = c0 ^ lf;
                        }
                        wave_fence_lds();
                    }
                    const int Wp = 1 << (t - 5);
                    for (int w = q; w < Wp; w += P) W.betaL[path][Wp + w] = W.curb[path][w];
                }
                if (t >= 5) {                                         // blocks of 32+ bits live in LDS slots
                    ptrB = ptr_set(ptrB, NLEV - t, path);
                    wave_fence_lds();
                }
            }
            ES_STAMP(t_beta);
        }

#ifdef ES_SCL_STAMPS
        ES_STAMP(t_misc);
#endif
        // ---------------- final ordering (fastpolar.py:335), trace-back, CRC
        if (q == 0) W.candm[path] = metric;
        wave_fence_lds();
        int rank = 0;
        for (int k = 0; k < cnt; ++k) {
            const double mk = W.candm[k];
            rank += ((mk < metric) || (mk == metric && k < path)) ? 1 : 0;
        }
        if (q == 0 && path < cnt) {
            int cur = path;
            uint32_t acc = 0;
            for (int tt = KINFO - 1; tt >= 0; --tt) {
                const uint32_t c = W.tb[tt][cur];
                acc |= (c & 1u) << (7 - (tt & 7));
                cur = (int)(c >> 1);
                if ((tt & 7) == 0) { W.outb[path][tt >> 3] = (uint8_t)acc; acc = 0; }
            }
            const int ok = crc8_bytes(W.outb[path], ES_INFO_BYTES) == W.outb[path][ES_INFO_BYTES];
            a.cand_metric[f * a.lsz + rank] = metric;
            a.cand_ok[f * a.lsz + rank] = (uint8_t)ok;
        }
        wave_fence_lds();
        if (path < cnt) {
            for (int k = q; k < ES_INFO_BYTES; k += P)
                a.cand_info[(f * a.lsz + rank) * ES_INFO_BYTES + k] = W.outb[path][k];
        }
        if (lane == 0) a.ncand[f] = cnt;
        wave_fence_lds();
#ifdef ES_SCL_STAMPS
        ES_STAMP(t_misc);
        if (lane == 0 && a.dbg && f < 64) {
            unsigned long long* o = a.dbg + f * 8;
            o[0] = t_big; o[1] = t_small; o[2] = t_dec_f; o[3] = t_dec_i; o[4] = t_beta; o[5] = t_misc;
            o[6] = t_s1; o[7] = t_s2; a.dbg[64 * 8 + 0] = t_s3; a.dbg[64 * 8 + 1] = t_s4;
        }
#endif
    }
}

// ---- polar encode (fastpolar.py:237-252): one wave per frame --------------------------------
__global__ __launch_bounds__(256) void es_polar_encode_kernel(const uint8_t* info, long long B,
                                                              const uint16_t* data_pos, uint8_t* code)
{
    __shared__ uint16_t s_dpos[KINFO];
    __shared__ uint32_t s_words[4][32];
    __shared__ uint8_t s_bytes[4][56];
    for (int i = threadIdx.x; i < KINFO; i += blockDim.x) s_dpos[i] = data_pos[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long f = (long long)blockIdx.x * 4 + wv; f < B; f += n_waves) {
        if (lane < ES_INFO_BYTES) s_bytes[wv][lane] = info[f * ES_INFO_BYTES + lane];
        if (lane < 32) s_words[wv][lane] = 0;
        wave_fence_lds();
        if (lane == 0) s_bytes[wv][ES_INFO_BYTES] = crc8_bytes(s_bytes[wv], ES_INFO_BYTES);
        wave_fence_lds();
        for (int tpos = lane; tpos < KINFO; tpos += 64) {
            const uint32_t bit = (s_bytes[wv][tpos >> 3] >> (7 - (tpos & 7))) & 1u;
            const int pos = s_dpos[tpos];
            if (bit) atomicOr(&s_words[wv][pos >> 5], 1u << (pos & 31));
        }
        wave_fence_lds();
        uint32_t word = s_words[wv][lane & 31];
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
        if (lane < 32) s_words[wv][lane] = word;
        wave_fence_lds();
        for (int k = lane; k < N; k += 64) code[f * N + k] = (uint8_t)((s_words[wv][k >> 5] >> (k & 31)) & 1u);
        wave_fence_lds();
    }
}

// ---- diagnostic: the device's log1p(exp(t)) (es_softplus_neg) on a vector, for the bit-for-bit parity test ----------
__global__ __launch_bounds__(256) void es_softplus_kernel(const double* t, long long n, const uint64_t* exp_tab, double* out)
{
    __shared__ __attribute__((aligned(16))) uint64_t s_exp[ES_EXP_TAB_WORDS];
    for (int i = threadIdx.x; i < ES_EXP_TAB_WORDS; i += blockDim.x) s_exp[i] = exp_tab[i];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = es_softplus_neg(t[i], s_exp);
}

template <int L>
long long scl_blocks(const es_ctx* ctx, long long B)
{
    constexpr int WPB = SclCfg<L>::WPB;
    long long blocks = (B + WPB - 1) / WPB;
    const long long max_blocks = (long long)ctx->num_cu * 2;    // LDS admits two blocks per CU
    return blocks < max_blocks ? blocks : max_blocks;
}

template <int L>
size_t scl_scratch_need(const es_ctx* ctx)
{
    return (size_t)ctx->num_cu * 2 * SclCfg<L>::WPB * L * GSLOT * sizeof(double);
}

template <int L>
int launch_scl(es_ctx* ctx, const SclArgs& a0, int64_t B, hipStream_t st)
{
    constexpr int WPB = SclCfg<L>::WPB;
    const long long blocks = scl_blocks<L>(ctx, B);
    if ((size_t)blocks * WPB * L * GSLOT * sizeof(double) > ctx->scl_scratch_bytes) {
        ctx->err = "es_scl_batch: scratch slab too small for this list size"; return ES_ENOMEM;
    }
    SclArgs a = a0;
    a.scratch = ctx->d_scl_scratch;
#ifdef ES_SCL_STAMPS
    static unsigned long long* dbg = nullptr;
    if (!dbg) { ES_HIP_CHECK(ctx, hipMalloc(&dbg, 65 * 8 * 8)); }
    ES_HIP_CHECK(ctx, hipMemset(dbg, 0, 65 * 8 * 8));
    a.dbg = dbg;
#endif
    { const int rc = es_slab_enter(ctx, 0, 0x100 | L, false, st); if (rc) return rc; }     // slab indexed by block: never shared between streams
    hipLaunchKernelGGL(es_scl_kernel<L>, dim3((unsigned)blocks), dim3(64 * WPB), 0, st, a);
    ES_HIP_CHECK(ctx, hipGetLastError());
    { const int rc = es_slab_leave(ctx, 0, 0x100 | L, false, st); if (rc) return rc; }
#ifdef ES_SCL_STAMPS
    {   // diagnostic build only: print the per-segment cycle shares of the first frames
        unsigned long long h[65 * 8];
        ES_HIP_CHECK(ctx, hipDeviceSynchronize());
        ES_HIP_CHECK(ctx, hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
        const char* nm[6] = {"chain S>=16", "chain S<=8", "decide frozen", "decide info", "beta fold", "rate-0 blocks + misc"};
        unsigned long long tot = 0; for (int k = 0; k < 6; ++k) tot += h[k];
        fprintf(stderr, "[scl stamps L=%d] frame0 total %llu cycles:", L, tot);
        for (int k = 0; k < 6; ++k) fprintf(stderr, " %s=%llu (%.1f%%)", nm[k], h[k], 100.0 * h[k] / (tot ? tot : 1));
        fprintf(stderr, "\n   info-bit split: pre(lp,pen)=%llu  candm+rank=%llu  select=%llu  shuffles+tb=%llu\n", h[6], h[7], h[64 * 8], h[64 * 8 + 1]);
    }
#endif
    return ES_OK;
}

}  // namespace

size_t es_scl_scratch_bytes(const es_ctx* ctx)
{
    size_t need = 0, n;
    const int lmax = es_list_cap(ctx->list_size_max);   /* lists above 32 use es_scl_wide.hip */
    if (lmax >= 1  && (n = scl_scratch_need<1>(ctx))  > need) need = n;
    if (lmax >= 2  && (n = scl_scratch_need<2>(ctx))  > need) need = n;
    if (lmax >= 4  && (n = scl_scratch_need<4>(ctx))  > need) need = n;
    if (lmax >= 8  && (n = scl_scratch_need<8>(ctx))  > need) need = n;
    if (lmax >= 16 && (n = scl_scratch_need<16>(ctx)) > need) need = n;
    if (lmax >= 32 && (n = scl_scratch_need<32>(ctx)) > need) need = n;
    return need;
}

int es_launch_scl(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
                  uint8_t* hard_info, uint8_t* hard_ok, uint8_t* cand_info, double* cand_metric,
                  uint8_t* cand_ok, int32_t* ncand, hipStream_t st)
{
    SclArgs a{};
    a.dbg = nullptr;
    a.llr = llr; a.is_f64 = (dtype == ES_DTYPE_F64); a.B = B;
    a.frozen = ctx->frozen; a.data_pos = ctx->d_data_pos; a.exp_tab = ctx->d_exp_tab;
    a.hard_info = hard_info; a.hard_ok = hard_ok; a.cand_info = cand_info;
    a.cand_metric = cand_metric; a.cand_ok = cand_ok; a.ncand = ncand;
    a.skip_if_hard_ok = skip_if_hard_ok;
    a.lsz = L;
    int LP = 1; while (LP < L) LP <<= 1;                  // kernel capacity: the next power of two
    switch (LP) {
        case 1:  return launch_scl<1>(ctx, a, B, st);
        case 2:  return launch_scl<2>(ctx, a, B, st);
        case 4:  return launch_scl<4>(ctx, a, B, st);
        case 8:  return launch_scl<8>(ctx, a, B, st);
        case 16: return launch_scl<16>(ctx, a, B, st);
        case 32: return launch_scl<32>(ctx, a, B, st);
        default: ctx->err = "list_size must be one of 1,2,4,8,16,32"; return ES_EINVAL;
    }
}

int es_launch_softplus(es_ctx* ctx, const double* t, int64_t n, double* out, hipStream_t st)
{
    long long blocks = (n + 255) / 256;
    if (blocks > (long long)ctx->num_cu * 16) blocks = (long long)ctx->num_cu * 16;
    hipLaunchKernelGGL(es_softplus_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t, (long long)n, ctx->d_exp_tab, out);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_polar_encode(es_ctx* ctx, const uint8_t* info, int64_t B, uint8_t* code, hipStream_t st)
{
    long long blocks = (B + 3) / 4;
    if (blocks > (long long)ctx->num_cu * 8) blocks = (long long)ctx->num_cu * 8;
    hipLaunchKernelGGL(es_polar_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, st, info,
                       (long long)B, ctx->d_data_pos, code);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
