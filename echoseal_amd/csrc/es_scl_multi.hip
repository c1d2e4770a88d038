// es_scl_multi.hip -- successive-cancellation LIST decoder for SHORT lists (L = 1, 2, 4, 8; also L = 16 as one frame
// per wave on this kernel's smaller LDS footprint) with SEVERAL frames per wavefront.  Same arithmetic, same bookkeeping and the same results as es_scl.hip; what changes is the
// mapping to the hardware.
//
// Why: with one frame per wave (es_scl.hip) a path owns P = 64/L lanes, and the bottom of the LLR tree cannot
// keep them busy: a leaf-level f() has two softplus terms per path, a leaf penalty one.  At L = 8 (P = 8) more
// than half of all softplus evaluations run on 1 or 2 useful lanes out of 8, and L = 1 is no faster than L = 8.
// Here a wave always carries 16 paths x 4 lanes: FR = 16/L frames of L paths each.  The top of the tree is as
// efficient as before (every lane busy), the bottom wastes half as many lanes, and the wave's bookkeeping
// (loop control, slot pointers, partial sums, sort latency) is shared by FR frames.  The frames of a wave
// advance in lock step (the schedule depends on the leaf index only); a sort ranks a path against the
// candidates of its own frame.
//
// LDS per wave is 9.9 KB (tree depths 6..7, partial sums, windowed trace-back), three 4-wave blocks per CU (a block always
// covers the four SIMDs of its CU): three waves = 3*FR frames per SIMD, <= 168 VGPRs.  Tree depths 1..5 live in the scratch
// slab, depths 8..10 in registers.  Blocks are NOT persistent: a wave decodes one group of FR frames and leaves, so that
// kernels of other streams (the front end of the next batch) get wave slots as this launch proceeds; the block's part of
// the slab is a slot claimed from a bitmap at block start and released at its end, which also lets concurrent launches
// of one context share the slab -- launches of the SAME slot geometry (same lanes per path): a bit maps to a different slab region per
// geometry, so es_slab_enter (es_api.hip) orders a launch of another geometry, on another stream, behind the outstanding ones.  Reference lines as in es_scl.hip (rtwm/fastpolar.py:254-359).
//
// Build with -ffp-contract=off: every rounding step in es_math.h is explicit.
#include "es_scl_common.h"

namespace {

// Lanes per path P: 4 (16 paths = 16/L frames per wave) or 2 (32 paths = 32/L frames per wave).  With two lanes per path
// the leaf-level f (two softplus terms) fills both lanes and only the leaf penalties run half empty: 11.3 k softplus
// lane-slots per path and decode instead of 13.3 k; the price is twice the per-wave state (LDS, slab).
template <int PP>
struct MCfg {
    static constexpr int P = PP;                                   // lanes per path
    static constexpr int NP = 64 / PP;                             // paths per wave
    static constexpr int LGP = (PP == 4) ? 2 : 1;                  // log2(P)
    static constexpr int RD = NLEV - LGP;                          // depths RD..10 (sizes P .. 1) live in registers
    static constexpr int GDEPTH = (PP == 4) ? 5 : 6;               // depths 1..GDEPTH live in the global scratch slab
    static constexpr int GSLOT = N - (N >> GDEPTH);                // doubles per path slot in the slab (992 / 1008)
    static constexpr int ROW = 2 * (N >> (GDEPTH + 1)) + 4;        // LDS row: depths GDEPTH+1 .. RD-1 at [S, 2S); + 4 pad (bank spread between paths)
    static constexpr bool TBW_GLOBAL = (PP == 2);                  // trace-back windows in the slab instead of LDS (LDS budget: three blocks per CU)
    // slot-storage loops: load pairs in flight in the g loops / next f operand pair requested ahead.  Measured (B = 65 536, L = 8): two lanes per
    // path 2.32 -> 2.43 M frames/s with (2, prefetch); four lanes per path is fastest with neither (2.19 M; (2, -) 1.93 M: spills)
    static constexpr int GBATCH = (PP == 2) ? 2 : 1;
    static constexpr bool PREFETCH = (PP == 2);
};
constexpr int MWIN = KINFO / 32;                   // trace-back windows of 32 information bits
constexpr int MWPB = 4;                            // waves per block
#ifndef ES_MULTI_MINW
#define ES_MULTI_MINW 3                            // waves per SIMD the register allocation must allow (LDS admits that many blocks per CU)
#endif
#ifndef ES_MULTI_ILP
#define ES_MULTI_ILP 1                             // independent f evaluations in flight per lane in the slot-storage loops
#endif
constexpr int MMINW = ES_MULTI_MINW;
template <int PP> constexpr int mslab_doubles() { return MCfg<PP>::NP * MCfg<PP>::GSLOT + (MCfg<PP>::TBW_GLOBAL ? MWIN * MCfg<PP>::NP / 2 : 0); }   // per wave

template <int L, int PP>
struct MWave {
    static constexpr int MNP = MCfg<PP>::NP;
    double   alphaS[MNP][MCfg<PP>::ROW];
    double   candm[2 * MNP];                       // frame fr: [2*L*fr, 2*L*(fr+1))
    uint32_t betaL[MNP][32];                       // left-sibling partial sums, block of S bits at bit S
    union {                                        // curb lives inside the bit loop, outb before and after it
        uint32_t curb[MNP][16];
        uint8_t  outb[MNP][56];
    };
    uint32_t hardw[32];
    uint32_t tbw[MCfg<PP>::TBW_GLOBAL ? 1 : MWIN][MNP];   // trace-back by windows of 32 information bits: the window's bits (first = MSB) ...
    uint8_t  tba[MWIN][MNP];                       // ... and the path (within the frame) this path descended from at the window's start
    uint8_t  sel[MNP];
};

template <int L, int PP>
__global__ __launch_bounds__(64 * MWPB, ((L <= 8 || PP == 2) ? MMINW : 1)) void es_scl_multi_kernel(SclArgs a)
{
    using C = MCfg<PP>;
    constexpr int P = C::P, LGP = C::LGP, RD = C::RD, MNP = C::NP, MGDEPTH = C::GDEPTH, MGSLOT = C::GSLOT;
    constexpr int FR = MNP / L;                    // frames per wave
    constexpr int FL = 64 / FR;                    // lanes per frame
    static_assert(L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || (L == 32 && PP == 2), "lists of at most 16 paths (32 with two lanes per path)");
    static_assert(PP == 2 || PP == 4, "two or four lanes per path");
    __shared__ __attribute__((aligned(16))) uint64_t s_exp[ES_EXP_TAB_WORDS];
    __shared__ uint16_t s_dpos[KINFO];
    __shared__ MWave<L, PP> s_wave[MWPB];

    for (int i = threadIdx.x; i < ES_EXP_TAB_WORDS; i += blockDim.x) s_exp[i] = a.exp_tab[i];
    for (int i = threadIdx.x; i < KINFO; i += blockDim.x) s_dpos[i] = a.data_pos[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int path = lane / P;                     // 0..MNP-1 within the wave
    const int q = lane % P;
    const int fr = path / L;                       // frame within the wave
    const int pl = path % L;                       // path within the frame
    const int fp0 = fr * L;                        // first path of the frame
    MWave<L, PP>& W = s_wave[wv];
    // ---- claim a slab slot for this block (bit per slot; the launch never has more resident blocks than slots)
    __shared__ int s_slot;
    if (threadIdx.x == 0) {
        int slot = -1;
        unsigned w = blockIdx.x % (unsigned)a.slot_words;
        for (int tries = 0; slot < 0 && tries < (1 << 22); ++tries) {
            const unsigned v = __hip_atomic_load(&a.slot_bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned valid = (w == (unsigned)a.slot_words - 1 && (a.n_slots & 31)) ? ((1u << (a.n_slots & 31)) - 1u) : 0xffffffffu;
            const unsigned freeb = ~v & valid;
            if (freeb) {
                const int bit = __ffs(freeb) - 1;
                const unsigned old = atomicOr(&a.slot_bits[w], 1u << bit);
                if (!(old & (1u << bit))) slot = (int)(w * 32u) + bit;
            } else {
                w = (w + 1u) % (unsigned)a.slot_words;
                if (tries > 64) __builtin_amdgcn_s_sleep(8);
            }
        }
        s_slot = slot;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0) {                                                    // cannot happen (residency <= slots, see the launcher); never spin
        constexpr int FRq = MCfg<PP>::NP / L;                           // forever, and never fail silently: the frames read ncand = -1
        for (long long ff = ((long long)blockIdx.x * MWPB + (threadIdx.x >> 6)) * FRq + (threadIdx.x & 63); (threadIdx.x & 63) < FRq && ff < a.B; ff += a.B) a.ncand[ff] = -1;
        return;
    }
    const long long wave_id = (long long)blockIdx.x * MWPB + wv;
    double* const scr = a.scratch + ((long long)slot * MWPB + wv) * (long long)mslab_doubles<PP>();
    uint32_t* const tbw_g = reinterpret_cast<uint32_t*>(scr + MNP * MGSLOT);          // P = 2: trace-back windows [MWIN][MNP]
    // Slab layout of a wave: one row of MGSLOT doubles per path slot, depth d at offset N - 2 (N >> d).  (An interleaved layout
    // -- element j of slot s at ((j / P) * MNP + s) * P + j % P, so that a wave instruction touches 512 contiguous bytes --
    // was measured SLOWER, 1.59 M against 2.02 M frames/s with four lanes per path: with a row per path every lane gets a
    // second hit on each cache line, and the kernel is bound by vector issue, not by the memory pipeline.)
    auto gaddr = [&](int off, int slot, int j) -> double* { return scr + slot * MGSLOT + off + j; };
    const uint64_t* const tab = s_exp;
    const long long n_groups = (a.B + FR - 1) / FR;

    for (long long g = wave_id; g < n_groups; g += (long long)gridDim.x * MWPB) {          // one group per wave (grid = groups / 4)
        const long long f_raw = g * FR + fr;
        const bool f_valid = f_raw < a.B;
        const long long f = f_valid ? f_raw : a.B - 1;          // a missing frame mirrors the last one (never stored)
        const float* llr32 = (const float*)a.llr + f * N;
        const double* llr64 = (const double*)a.llr + f * N;

        // ---------------- hard decision -> butterfly -> data bits -> CRC (fastpolar.py:260-268), frame by frame
        uint32_t active_mask = 0;                              // wave-uniform: frames that go through the list loop
        for (int fi = 0; fi < FR; ++fi) {
            const long long ff = g * FR + fi;
            if (ff >= a.B) break;
            const float* l32 = (const float*)a.llr + ff * N;
            const double* l64 = (const double*)a.llr + ff * N;
            uint32_t word = 0;
            for (int c = 0; c < 16; ++c) {
                const double v = a.is_f64 ? l64[64 * c + lane] : (double)l32[64 * c + lane];
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
            if (lane < ES_INFO_BYTES) a.hard_info[ff * ES_INFO_BYTES + lane] = W.outb[0][lane];
            if (lane == 0) a.hard_ok[ff] = (uint8_t)ok;
            wave_fence_lds();
            if (ok && a.skip_if_hard_ok) {                 // no list for this record: its candidate rows read as zeros
                if (lane == 0) a.ncand[ff] = 0;
                #pragma unroll 1
                for (int k = lane; k < a.lsz * ES_INFO_BYTES; k += 64) a.cand_info[ff * a.lsz * ES_INFO_BYTES + k] = 0;
                if (lane < a.lsz) { a.cand_metric[ff * a.lsz + lane] = 0.0; a.cand_ok[ff * a.lsz + lane] = 0; }
            } else active_mask |= 1u << fi;
        }
        if (active_mask == 0) continue;
        const bool f_store = f_valid && ((active_mask >> fr) & 1u);

        // ---------------- list decoding (fastpolar.py:278-330)
        uint64_t ptrA = 0, ptrB = 0;
        #pragma unroll
        for (int d = 1; d <= NLEV; ++d) { ptrA = ptr_set(ptrA, d, fp0); ptrB = ptr_set(ptrB, d, fp0); }   // mirrors of the frame's path 0
        double metric = 0.0;
        double ar[LGP + 1];
        #pragma unroll
        for (int k = 0; k <= LGP; ++k) ar[k] = 0.0;
        double sp_diff = 0.0, sp_sum = 0.0, lp_odd = 0.0;
        uint32_t b0 = 0;
        uint32_t hist = 0, anc = 0;       // trace-back window (see the sort)
        uint32_t frozen_word = 0;
        int cnt = 1;                      // live paths per frame
        int info_idx = 0;
        if (lane < 32) { for (int s = 0; s < MNP; ++s) W.betaL[s][lane] = 0; }
        wave_fence_lds();

        for (int i = 0; i < N; ++i) {
            if ((i & 31) == 0) frozen_word = a.frozen.w[i >> 5];
            // --- all-frozen aligned block starting here (rate-0 node)?  log2 of its size, 0 = none
            int blk = 0;
            if ((i & 1) == 0 && i != 0) {
                const uint32_t fw = frozen_word >> (i & 31);
                #pragma unroll
                for (int t = 1; t <= 5; ++t) {
                    const int S = 1 << t;
                    const uint32_t m = (S == 32) ? 0xffffffffu : ((1u << S) - 1u);
                    if (blk == t - 1 && (i & (S - 1)) == 0 && (fw & m) == m) blk = t;
                }
                if (blk == 5 && (i & 63) == 0 && a.frozen.w[(i >> 5) + 1] == 0xffffffffu) blk = 6;
            }
            const int d_stop = blk ? NLEV - blk : NLEV;
            // --- LLR chain: recompute the depths that changed since leaf i-1 (fastpolar.py:127-154)
            const int top = (i == 0) ? 1 : NLEV - __builtin_ctz((unsigned)i);
            // (a) depths above the register-resident part: slot storage in scratch / LDS
            for (int d = top; d < RD && d <= d_stop; ++d) {
                const int S = N >> d;
                const bool is_g = (i >> (NLEV - d)) & 1;
                const int ps = (d > 1) ? ptr_get(ptrA, d - 1) : 0;
                const int bs = ptr_get(ptrB, d);
                // First chain (i == 0): the paths of a frame are still copies of its path 0, so the node is computed
                // once per frame, by the frame's FL lanes, into that path's slot.
                const bool first = (i == 0);
                const int own = first ? fp0 : path;
                const int j0 = first ? (lane % FL) : q;
                const int jst = first ? FL : P;
                const double* par_l = &W.alphaS[ps][2 * S];                    // (depth d-1 in the slab: region offset N - 4S; depth d: N - 2S)
                double* dst_l = &W.alphaS[own][S];
                // Loaders and stores are chosen OUTSIDE the element loops (channel LLRs / slab / LDS): with the choice inside, control flow sits
                // between a load and its use and the compiler waits for everything outstanding after every load (as in es_scl_wide.hip).
                auto run_level = [&](auto load_pair, auto store_out) {
                    if (is_g) {
                        int j = q;
                        if constexpr (C::GBATCH > 1) {
                            for (; j + (C::GBATCH - 1) * P < S; j += C::GBATCH * P) {      // (g is an add: the loop is load latency) independent pairs in flight
                                double xa[C::GBATCH], xb[C::GBATCH];
                                #pragma unroll
                                for (int v = 0; v < C::GBATCH; ++v) load_pair(j + v * P, xa[v], xb[v]);
                                #pragma unroll
                                for (int v = 0; v < C::GBATCH; ++v) {
                                    const int jj = j + v * P;
                                    const uint32_t wbits = (S <= 16) ? b0 : W.betaL[bs][(S + jj) >> 5];
                                    store_out(jj, es_polar_g(xa[v], xb[v], (wbits >> ((S + jj) & 31)) & 1u));
                                }
                            }
                        }
                        for (; j < S; j += P) {
                            double pa, pb; load_pair(j, pa, pb);
                            const uint32_t wbits = (S <= 16) ? b0 : W.betaL[bs][(S + j) >> 5];
                            store_out(j, es_polar_g(pa, pb, (wbits >> ((S + j) & 31)) & 1u));
                        }
                    } else {
                        int j = j0;
#if ES_MULTI_ILP == 2
                        for (; j + jst < S; j += 2 * jst) {        // two independent f chains in flight
                            double a0, c0, a1, c1; load_pair(j, a0, c0); load_pair(j + jst, a1, c1);
                            const double o0 = es_polar_f(a0, c0, tab);
                            const double o1 = es_polar_f(a1, c1, tab);
                            store_out(j, o0); store_out(j + jst, o1);
                        }
#endif
                        // one f (= two interleaved softplus chains) in flight per lane: the other waves of the SIMD hide the rest
                        if constexpr (C::PREFETCH) {
                            // ... and the NEXT pair of parents already on its way (past the end: this lane's last element again)
                            if (j < S) {
                                double pa, pb; load_pair(j, pa, pb);
                                for (; j < S; j += jst) {
                                    double na, nb;
                                    load_pair(j + jst < S ? j + jst : j, na, nb);
                                    store_out(j, es_polar_f(pa, pb, tab));
                                    pa = na; pb = nb;
                                }
                            }
                        } else {
                            for (; j < S; j += jst) { double pa, pb; load_pair(j, pa, pb); store_out(j, es_polar_f(pa, pb, tab)); }
                        }
                    }
                };
                auto st_slab = [&](int j, double v) { *gaddr(N - 2 * S, own, j) = v; };
                auto st_lds = [&](int j, double v) { dst_l[j] = v; };
                auto ld_slab = [&](int j, double& pa, double& pb) { pa = *gaddr(N - 4 * S, ps, j); pb = *gaddr(N - 4 * S, ps, j + S); };
                auto ld_lds = [&](int j, double& pa, double& pb) { pa = par_l[j]; pb = par_l[j + S]; };
                if (d == 1) {                                                  // the channel LLRs (depth 1 always lives in the slab)
                    if (a.is_f64) run_level([&](int j, double& pa, double& pb) { pa = llr64[j]; pb = llr64[j + S]; }, st_slab);
                    else run_level([&](int j, double& pa, double& pb) { pa = (double)llr32[j]; pb = (double)llr32[j + S]; }, st_slab);
                } else if (d <= MGDEPTH) run_level(ld_slab, st_slab);
                else if (d - 1 <= MGDEPTH) run_level(ld_slab, st_lds);
                else run_level(ld_lds, st_lds);
                if (d <= MGDEPTH) wave_fence_global(); else wave_fence_lds();
                ptrA = ptr_set(ptrA, d, own);
            }
            // (b) depth RD: one element per lane, straight from the slot of depth RD-1 into a register
            if (RD >= top && RD <= d_stop) {
                const int d = RD;
                const bool is_g = (i >> (NLEV - d)) & 1;
                const double* par = &W.alphaS[ptr_get(ptrA, d - 1)][2 * P];
                const double pa = par[q], pb = par[q + P];
                if (is_g) ar[0] = es_polar_g(pa, pb, (b0 >> ((P + q) & 31)) & 1u);
                else ar[0] = es_polar_f(pa, pb, tab);
            }
            // (c) depths RD+1..10: register to register, lane-split softplus (see es_scl.hip)
            #pragma unroll
            for (int k = 1; k <= LGP; ++k) {
                const int d = RD + k;
                if (d >= top && d <= d_stop) {
                    const int S = P >> k;
                    const bool is_g = (i >> (NLEV - d)) & 1;
                    const double own = ar[k - 1];
                    const double oth = xor_lanes_f64_sw(own, S, lane);
                    const bool hi = (q & S) != 0;
                    const double pa = hi ? oth : own;
                    const double pb = hi ? own : oth;
                    if (is_g) {
                        const int j = q & (S - 1);
                        ar[k] = es_polar_g(pa, pb, (b0 >> ((S + j) & 31)) & 1u);
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

            uint32_t bit = 0;
            if (blk) {
                // ---------------- rate-0 block of S = 2^blk leaves (all bits 0: every g is b + a); see es_scl.hip
                const int S = 1 << blk;
                if (S > P) {
                    const int dn = NLEV - blk;                        // the node's depth: 4 (scratch) .. 7 (LDS)
                    double* const Xl = &W.alphaS[path][S];
                    const bool in_g = dn <= MGDEPTH;
                    auto ld = [&](int e) { return in_g ? *gaddr(N - 2 * S, path, e) : Xl[e]; };
                    auto st = [&](int e, double v) { if (in_g) *gaddr(N - 2 * S, path, e) = v; else Xl[e] = v; };
                    for (int h = S >> 1; h >= P; h >>= 1) {           // nodes of 2h values -> children of h values
                        const int lh = 31 - __builtin_clz((unsigned)h);
                        int idx = q;
#if ES_MULTI_ILP == 2
                        for (; idx + P < (S >> 1); idx += 2 * P) {    // two independent f chains in flight
                            const int e0 = ((idx >> lh) << (lh + 1)) + (idx & (h - 1));
                            const int e1 = (((idx + P) >> lh) << (lh + 1)) + ((idx + P) & (h - 1));
                            const double a0 = ld(e0), c0 = ld(e0 + h), a1 = ld(e1), c1 = ld(e1 + h);
                            const double o0 = es_polar_f(a0, c0, tab);
                            const double o1 = es_polar_f(a1, c1, tab);
                            st(e0, o0); st(e0 + h, es_polar_g(a0, c0, 0u));
                            st(e1, o1); st(e1 + h, es_polar_g(a1, c1, 0u));
                        }
#endif
                        for (; idx < (S >> 1); idx += P) {
                            const int e0 = ((idx >> lh) << (lh + 1)) + (idx & (h - 1));
                            const double a0 = ld(e0), c0 = ld(e0 + h);
                            st(e0, es_polar_f(a0, c0, tab)); st(e0 + h, es_polar_g(a0, c0, 0u));
                        }
                        if (in_g) wave_fence_global(); else wave_fence_lds();
                    }
                }
                const int k0 = (blk >= LGP) ? 0 : LGP - blk;
                const int nsub = (S > P) ? S / P : 1;
                const int nleaf = (S < P) ? S : P;
                for (int sb = 0; sb < nsub; ++sb) {
                    double x = ar[0];
                    if (S > P) x = (NLEV - blk <= MGDEPTH) ? *gaddr(N - 2 * S, path, sb * P + q) : W.alphaS[path][S + sb * P + q];
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
                    const double lp = es_softplus_neg(-al, tab);
                    double pen = (q & 1) ? L2last : lp;
                    if (x >= 0.0) pen = pen + al;
                    #pragma unroll
                    for (int k = 0; k < P; ++k) if (k < nleaf) metric = metric + __shfl(pen, path * P + k);
                }
                if (S >= 32) b0 = 0; else b0 &= ~((1u << S) - 1u);
                for (int sl = 5; sl < blk; ++sl) {
                    const int Wd = 1 << (sl - 5);
                    for (int w = q; w < Wd; w += P) W.betaL[path][Wd + w] = 0;
                    ptrB = ptr_set(ptrB, NLEV - sl, path);
                }
                wave_fence_lds();
                i += S - 1;
            } else {
            const double lam = ar[LGP];

            // --- decision
            const bool frozen = (frozen_word >> (i & 31)) & 1u;
            const double al = __builtin_fabs(lam);
            double lp;
            if (i & 1) lp = lp_odd;                                  // set when the even sibling was decided
            else lp = es_softplus_neg(-al, tab);
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
                // stable rank of candidate 2*pl + q among the 2*cnt live candidates of the frame
                const bool is_cand = (q < 2) && (pl < cnt);
                const int cl = 2 * pl + q;
                if (is_cand) W.candm[2 * fp0 + cl] = m;
                wave_fence_lds();
                const int nc = 2 * cnt;
                constexpr int G = (P >= 2) ? P / 2 : 1;             // lanes sharing one candidate
                constexpr int SPAN = (2 * L + G - 1) / G;            // candidates each of them compares against
                const int cb = q & 1;
                double mc = m;                                       // metric of candidate 2*pl + cb: lane (path, q & 1) -- the lane itself when P = 2
                if constexpr (P == 4) {
                    uint64_t u; __builtin_memcpy(&u, &m, 8);
                    const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)u, 0x44, 0xf, 0xf, true);          // quad_perm [0,1,0,1]
                    const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), 0x44, 0xf, 0xf, true);
                    u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
                    __builtin_memcpy(&mc, &u, 8);
                }
                const int cc_ = 2 * pl + cb;
                const int kk0 = (q >> 1) * SPAN;
                int rank = 0;
                #pragma unroll 8
                for (int u = 0; u < SPAN; ++u) {                     // (at most eight metrics in flight: registers)
                    const int k = kk0 + u;
                    const double mk = W.candm[2 * fp0 + (k < 2 * L ? k : 0)];
                    rank += ((k < nc) && ((mk < mc) || (mk == mc && k < cc_))) ? 1 : 0;
                }
                if constexpr (P == 4) rank += xor_lanes_b32<2>(rank, lane);
                const int keep = nc < a.lsz ? nc : a.lsz;          // a.lsz <= L: lists of any size run on the next power of two's kernel
                if (is_cand && rank < keep) W.sel[fp0 + rank] = (uint8_t)cl;
                wave_fence_lds();
                const int cc = W.sel[fp0 + (pl < keep ? pl : 0)];
                const int src = (fp0 + (cc >> 1)) * P + (cc & 1);
                const int parent = src / P;                          // path index within the wave
                bit = (uint32_t)(cc & 1);
                metric = __shfl(m, src);
                ptrA = __shfl(ptrA, parent * P);
                ptrB = __shfl((ptrB & 0xffffffffULL) | ((uint64_t)b0 << 32), parent * P);
                b0 = (uint32_t)(ptrB >> 32);
                ptrB &= 0xffffffffULL;
                #pragma unroll
                for (int k = 0; k <= LGP; ++k)
                    if (((i + 1) & ((1 << (LGP - k)) - 1)) != 0) ar[k] = __shfl(ar[k], parent * P + q);
                if (!(i & 1)) lp_odd = __shfl((q & 1) ? sp_diff : sp_sum, src);
                // trace-back by windows of 32 information bits: a path carries the bits it decided inside the current
                // window (hist) and the path it descended from at the window's start (anc); both follow the path at a
                // sort, and a full window is written to LDS once: the final trace-back then takes 14 steps, not 448.
                {
                    const uint32_t anc_own = (info_idx & 31) == 0 ? (uint32_t)pl : anc;
                    const uint32_t packed = __shfl((int)((info_idx & 31) == 0 ? 0u : hist), parent * P);
                    anc = (uint32_t)__shfl((int)anc_own, parent * P);
                    hist = (packed << 1) | bit;
                    if ((info_idx & 31) == 31 && q == 0) {
                        if constexpr (C::TBW_GLOBAL) tbw_g[(info_idx >> 5) * MNP + path] = hist; else W.tbw[info_idx >> 5][path] = hist;
                        W.tba[info_idx >> 5][path] = (uint8_t)anc;
                    }
                }
                cnt = keep;
                ++info_idx;
                wave_fence_lds();
            }
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
        }

        // ---------------- final ordering (fastpolar.py:335), trace-back, CRC -- per frame
        if constexpr (C::TBW_GLOBAL) wave_fence_global();               // the windows written to the slab are read back by other lanes
        if (q == 0) W.candm[path] = metric;
        wave_fence_lds();
        int rank = 0;
        for (int k = 0; k < cnt; ++k) {
            const double mk = W.candm[fp0 + k];
            rank += ((mk < metric) || (mk == metric && k < pl)) ? 1 : 0;
        }
        if (q == 0 && pl < cnt) {
            int cur = pl;
            for (int w = MWIN - 1; w >= 0; --w) {
                const uint32_t word = C::TBW_GLOBAL ? tbw_g[w * MNP + fp0 + cur] : W.tbw[w][fp0 + cur];   // information bits 32w .. 32w+31, first = MSB
                cur = (int)W.tba[w][fp0 + cur];
                W.outb[path][4 * w + 0] = (uint8_t)(word >> 24); W.outb[path][4 * w + 1] = (uint8_t)(word >> 16);
                W.outb[path][4 * w + 2] = (uint8_t)(word >> 8);  W.outb[path][4 * w + 3] = (uint8_t)word;
            }
            if (f_store) {
                const int ok = crc8_bytes(W.outb[path], ES_INFO_BYTES) == W.outb[path][ES_INFO_BYTES];
                a.cand_metric[f * a.lsz + rank] = metric;
                a.cand_ok[f * a.lsz + rank] = (uint8_t)ok;
            }
        }
        wave_fence_lds();
        if (pl < cnt && f_store) {
            #pragma unroll 1
            for (int k = q; k < ES_INFO_BYTES; k += P)                       // (unrolled, its 28 index registers stay live through the whole bit loop)
                a.cand_info[(f * a.lsz + rank) * ES_INFO_BYTES + k] = W.outb[path][k];
        }
        if (q == 0 && pl == 0 && f_store) a.ncand[f] = cnt;
        wave_fence_lds();
    }
    // ---- release the slot: every wave's slab stores are done (and performed) before the bit clears.  The fence is agent-scope: on a
    // chip whose XCDs have their own L2 it also writes this XCD's dirty slab lines back, so a late eviction here cannot overwrite
    // what the slot's next owner -- possibly on another XCD -- has written since (an owner only ever reads what it wrote itself).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // (release only: an acquire would also invalidate this XCD's L2 under the blocks still running)
    __syncthreads();
    if (threadIdx.x == 0) atomicAnd(&a.slot_bits[slot >> 5], ~(1u << (slot & 31)));
}

template <int L, int PP>
int launch_multi(es_ctx* ctx, const SclArgs& a0, int64_t B, hipStream_t st)
{
    constexpr int FR = MCfg<PP>::NP / L;
    const long long groups = (B + FR - 1) / FR;
    const long long blocks = (groups + MWPB - 1) / MWPB;          // one group per wave; the hardware keeps <= MMINW blocks per CU resident
    const int n_slots = ctx->num_cu * MMINW;
    if ((size_t)n_slots * MWPB * mslab_doubles<PP>() * sizeof(double) > ctx->scl_scratch_bytes || !ctx->d_slot_bits) {
        ctx->err = "es_scl_batch: scratch slab too small for the multi-frame kernel"; return ES_ENOMEM;
    }
    if (blocks >= (1LL << 31)) { ctx->err = "es_scl_batch: batch too large for one launch"; return ES_EINVAL; }
    SclArgs a = a0;
    a.scratch = ctx->d_scl_scratch;
    a.slot_bits = ctx->d_slot_bits; a.n_slots = n_slots; a.slot_words = (n_slots + 31) / 32;
    { const int rc = es_slab_enter(ctx, 0, 0x200 | PP, true, st); if (rc) return rc; }      // slot stride depends on PP only
    hipLaunchKernelGGL((es_scl_multi_kernel<L, PP>), dim3((unsigned)blocks), dim3(64 * MWPB), 0, st, a);
    ES_HIP_CHECK(ctx, hipGetLastError());
    { const int rc = es_slab_leave(ctx, 0, 0x200 | PP, true, st); if (rc) return rc; }
    return ES_OK;
}

}  // namespace

size_t es_scl_multi_scratch_bytes(const es_ctx* ctx)
{
    const size_t per_wave = mslab_doubles<2>() > mslab_doubles<4>() ? mslab_doubles<2>() : mslab_doubles<4>();
    return (size_t)ctx->num_cu * MMINW * MWPB * per_wave * sizeof(double);
}

int es_launch_scl_multi(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
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
    // lanes per path (es_set_option "scl_lanes": 2, 4, or 0 = choose): two lanes per path need fewer instructions per frame
    // (measured 2.15 M against 2.03 M frames/s at 65 536 frames) but a wave then carries 32/L frames and takes 1.6x as long, so
    // the choice falls on it once the batch gives every wave slot of the chip at least two such waves
    const bool two = ctx->scl_lanes == 2 || (ctx->scl_lanes == 0 && (long long)B * LP / 32 >= 2LL * ctx->num_cu * MMINW * MWPB);
    switch (LP) {
        case 1: return two ? launch_multi<1, 2>(ctx, a, B, st) : launch_multi<1, 4>(ctx, a, B, st);
        case 2: return two ? launch_multi<2, 2>(ctx, a, B, st) : launch_multi<2, 4>(ctx, a, B, st);
        case 4: return two ? launch_multi<4, 2>(ctx, a, B, st) : launch_multi<4, 4>(ctx, a, B, st);
        case 8: return two ? launch_multi<8, 2>(ctx, a, B, st) : launch_multi<8, 4>(ctx, a, B, st);
        case 16: return two ? launch_multi<16, 2>(ctx, a, B, st) : launch_multi<16, 4>(ctx, a, B, st);
        case 32: return launch_multi<32, 2>(ctx, a, B, st);          // one frame per wave, 32 paths x 2 lanes
        default: ctx->err = "the multi-frame list decoder serves list sizes up to 32"; return ES_EINVAL;
    }
}
