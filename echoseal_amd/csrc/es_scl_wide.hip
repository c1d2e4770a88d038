// es_scl_wide.hip -- successive-cancellation LIST decoder with ONE LANE PER PATH: lists of 64, 128 and 256 paths (one frame
// per block of L lanes; the detector's default list size is 256, rtwm/detector.py:27 -- this kernel is what lets
// WatermarkDetector(key).verify(...) run unchanged), and shorter lists as 64/L whole frames per one-wave block -- the
// throughput mapping of large launches.  Same arithmetic and the same bookkeeping idea as es_scl.hip (per-depth slot
// pointers instead of path copies, trace-back instead of per-path bit arrays, Python's stable sort order).
//
// Where things live (lane p owns path p and slot p of every slot-indexed store):
//   * LLR tree depths 1..DL-1 (512..16 values per path): a scratch slab laid out [element][slot] (no cache holds it), so the lanes of a
//     wave touch one contiguous row; a lane walks a node's elements serially.  DL = 7 in one-wave blocks, 8 in blocks of several waves.
//     The g half of depth 1 (bits 512..) is never stored: llr[e + 512] +- llr[e], formed on the fly where it is consumed (ES_WIDE_L1R).
//   * depths DL..9 (8 + 4 + 2 values at DL = 7): LDS, [element][slot]; depth 10 (the leaf LLR): a register.
//   * slot pointers (which slot holds my data at depth d): one byte per depth packed in two 64-bit registers,
//     partial-sum blocks of 1..16 bits in one 32-bit register (as in es_scl.hip), of 32 and 64 bits in LDS by
//     slot, of 128 / 256 / 512 bits (read a few times per frame) in the slab.
//   * a sort moves registers only.  One-wave blocks: the survivor of rank r -- lane r -- fetches sorted element r and its parent's
//     (pointers, small partial sums, trace-back window, softplus pair of the even sibling) straight from the registers of the lanes that
//     hold them (ds_bpermute).  Blocks of several waves publish them to LDS and read the parent's after the sort's barriers.
//   * trace-back by windows of 32 information bits (as in es_scl_multi.hip): a path carries the bits of the
//     current window and the path it descended from at the window's start; a full window goes to the slab once.
//
// The sort is a bitonic network over the 2L candidates on (metric, candidate index) -- a strict total order, so
// the result is Python's stable list.sort -- with two candidates per lane IN REGISTERS: partner lanes exchange
// through DPP / ds_swizzle / ds_bpermute, only the stages whose partner sits in another wave (3 of 45 at L = 256)
// go through LDS and a workgroup barrier.
//
// Barriers.  A path may read a slot that another wave wrote, but only one it inherited at a sort (whose barriers
// order the write before the read); and a lane only ever writes its OWN slot.  What remains is write-after-read:
// before a step that writes shared slots, every wave must have finished the reads of earlier steps -- one barrier,
// and only when no sort has happened since those reads (`dirty`).  A depth-by-depth barrier is not needed: within a
// step a path reads what it has itself just written.
//
// SHORT lists (LF = 1..64 paths per frame): the block is one wave that carries 64/LF whole frames (8 at LF = 8); the sort network
// stops at the frame's 2 LF candidates and every "barrier" is a wave fence.  All 64 lanes are busy at every tree depth, which the
// lanes-share-a-path kernels (es_scl.hip, es_scl_multi.hip) cannot offer at the bottom of the tree: 126 k instead of 184 k / 224 k
// vector instructions per frame at L = 8.  The price is latency: a wave carries its frames through the whole decode (~7 ms), so
// es_scl_batch picks this mapping for large launches only (or es_set_option "scl_lanes" = 1: the grouped pipeline).
//
// Blocks are not persistent: a block decodes its frames and leaves; its slab slot comes from a bitmap (as in es_scl_multi.hip).
//
// Values are bit-identical to the reference list decoder for the same reason as in es_scl.hip (es_math.h).
// This file's kernels use dynamic LDS only, with the exp table first in it: the table sits at LDS address 0 (checked at kernel start),
// which lets es_math.h address its entries without adding a base.
#define ES_EXP_TAB_LDS_ADDR 0u
#include "es_scl_common.h"
#include <type_traits>
#ifndef ES_WIDE_GBATCH
#define ES_WIDE_GBATCH 8                          /* load pairs in flight in the lane-serial g loops (divides 8; 8: +1 % over 4, measured) */
#endif

#ifndef ES_WIDE_WPS
#define ES_WIDE_WPS 3                             /* waves per SIMD the kernel is compiled for (168 VGPRs at 3) */
#endif
#ifndef ES_WIDE_FUSE_GF
#define ES_WIDE_FUSE_GF 1                         /* 1: the g at the top of a step and the f level below it in one pass (slab levels 2..6); 3: also pairs of f levels */
#endif
#ifndef ES_WIDE_DEFER
#define ES_WIDE_DEFER 12                          /* where the generic softplus is a deferred cold path instead of a branch after each evaluation: 1 slab f loops, 2 depth 8, 4 depth 9, 8 depth 10 */
#endif
#ifndef ES_WIDE_COMPACT
#define ES_WIDE_COMPACT 1                         /* skip_if_hard_ok: blocks draw frames from a counter until they hold a full group that failed the hard decision (0: fixed groups, settled frames ride along as idle lanes) */
#endif
#ifndef ES_WIDE_FDIST
#define ES_WIDE_FDIST 2                           /* f loops: operand pairs requested this many f evaluations ahead (1: rotation by copy; 2: unrolled by three) */
#endif
#ifndef ES_WIDE_L1R
#define ES_WIDE_L1R 1                             /* the g half of tree depth 1 is never stored: its elements are llr[e + 512] +- llr[e] by one partial-sum bit, formed on the fly by the two passes that consume them (bits 512 and 768) */
#endif
#ifndef ES_WIDE_LDS_DEPTH
#define ES_WIDE_LDS_DEPTH 7                       /* one-wave blocks: first LLR-tree depth kept in LDS (7: 8 + 4 + 2 rows of doubles by slot; 8: round 3's layout; 6 needs two waves per SIMD) */
#endif

namespace {

struct WideArgs {
    const void* llr; int is_f64; long long B;
    es_frozen_mask frozen;
    const uint16_t* data_pos;
    const uint64_t* exp_tab;
    double* alpha;            // [slots][1024][L]   (elements 8..1023 used: depth d at [1024 >> d, 2 * (1024 >> d)))
    unsigned char* aux;       // [slots][ES_WIDE_AUX_PER_PATH * L]: trace-back windows, wide partial-sum blocks, fold scratch
    uint8_t* hard_info; uint8_t* hard_ok;
    uint8_t* cand_info; double* cand_metric; uint8_t* cand_ok; int32_t* ncand;
    int skip_if_hard_ok;
    int lsz;                                  // the caller's list size (<= L)
    unsigned* slot_bits; int n_slots, slot_words;   // bitmap of slab slots (one per resident block), as in es_scl_multi.hip
    int* cursor;                              // null: block g decodes frames [g * FRG, (g + 1) * FRG).  Else (skip_if_hard_ok) blocks draw frames from this counter
                                              //   (zeroed before the launch) until they hold FRG that failed the hard decision: settled frames never ride along
    int n_info, info_bytes;                   // GK instantiations: data bits K (information + CRC) and bytes of a packed information row, ceil((K - 8) / 8)
    int prio;                                 // wave priority 0..3 (es_set_option "scl_prio"): a later launch of a burst may overtake an earlier one
};

constexpr int MWIN_W = KINFO / 32;            // trace-back windows of the default code (448 data bits)
constexpr int MWIN_MAX = N / 32;              // ... of any code (GK instantiations: K data bits, 16 <= K <= 1024, K % 8 == 0)
// aux slab per path: windows 32 x (4 + 2) B (14 used by the default code), partial-sum blocks of 128 / 256 / 512 bits 28 x 4 B, fold scratch 16 x 4 B
constexpr int WIDE_AUX_PER_PATH = MWIN_MAX * 6 + 28 * 4 + 16 * 4;

// LLR-tree depths DL..9 live in LDS, [element][slot]: depth d at rows [wide_low_row(DL, d), +1024 >> d).  A block that is one wave keeps depth 7
// there too (DL = 7: a step whose top is depth 7 or below then touches the slab only to READ depth 6, and the values that are written and
// read back within a few hundred cycles never leave the CU); blocks of several waves hold a frame of 128 / 256 paths and have no LDS to spare.
constexpr int wide_dl(bool one_wave) { return one_wave ? ES_WIDE_LDS_DEPTH : 8; }
constexpr int wide_low_row(int DL, int d) { int r = 0; for (int k = DL; k < d; ++k) r += N >> k; return r; }

// What follows a path through a sort.  Several waves per frame (NB = 2: double-buffered by information index, a wave may still be reading
// while another is a sort further): every path publishes it to LDS and the survivor of rank r reads its parent's.  ONE wave per block
// (NB = 1): nothing is stored -- the survivor fetches its parent's registers with ds_bpermute (the lanes of a wave run in lock step).
template <int L, int NB>
struct WidePub {
    double   skey[NB][2 * L];           // cross-wave sort stages and the read-out
    double   xsp[NB][2][L];             // softplus pair of the even sibling,
    uint64_t xpa[NB][L];                // ... LLR-tree slot pointers,
    uint64_t xpb[NB][L];                // ... partial-sum slot pointers | depth-9 pointer << 40 | window ancestor << 48,
    uint32_t xb0[NB][L];                // ... partial sums of 1..16 bits,
    uint32_t xhist[NB][L];              // ... bits of the current trace-back window
    uint16_t sidx[NB][2 * L];
};
#ifndef ES_WIDE_GATHER_BPERM
#define ES_WIDE_GATHER_BPERM 1                    /* one-wave blocks: survivors take their parent's state with ds_bpermute (0: through LDS, as blocks of several waves do) */
#endif
#if ES_WIDE_GATHER_BPERM
template <int L> struct WidePub<L, 1> { };
#endif

template <int L, int NB>
struct WideLds {
    static constexpr int DL = wide_dl(NB == 1);
    static constexpr int NLOW = wide_low_row(DL, 10);
    uint64_t exp_tab[ES_EXP_TAB_WORDS];
    double   low[NLOW][L];              // depths DL..9, by slot (also: the hard decision's 184 bytes and the final ordering's metrics, outside the list loop)
    uint32_t betaM[3][L];               // partial-sum blocks of 32 (row 0) and 64 bits (1, 2), by slot (wider ones, touched a few times per frame: slab)
    int      flag;
    WidePub<L, NB> pub;                 // (the frames drawn from the cursor, up to 64 ints per one-wave block, borrow row 1 of `low` before the list loop starts)
};

__device__ __forceinline__ uint64_t p8_set(uint64_t w, int k, int v) { const int sh = 8 * k; return (w & ~(255ULL << sh)) | ((uint64_t)(uint32_t)v << sh); }
__device__ __forceinline__ int p8_get(uint64_t w, int k) { return (int)((w >> (8 * k)) & 255ULL); }

// lane l <-> lane l ^ D inside a wave, D a constant after unrolling
__device__ __forceinline__ int wxor_b32(int v, int D)
{
    switch (D) {
        case 1:  return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
        case 2:  return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
        case 4:  return __builtin_amdgcn_ds_swizzle(v, 0x101F);                // bit mode: xor 4
        case 8:  return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);    // row_ror:8
        case 16: return __builtin_amdgcn_ds_swizzle(v, 0x401F);                // bit mode: xor 16
        default: return __shfl_xor(v, 32);
    }
}
__device__ __forceinline__ double wxor_f64(double x, int D)
{
    uint64_t u; __builtin_memcpy(&u, &x, 8);
    const uint32_t lo = (uint32_t)wxor_b32((int)(uint32_t)u, D);
    const uint32_t hi = (uint32_t)wxor_b32((int)(uint32_t)(u >> 32), D);
    u = ((uint64_t)hi << 32) | lo;
    double r; __builtin_memcpy(&r, &u, 8); return r;
}

__device__ __forceinline__ bool cand_before(double ka, uint32_t ia, double kb, uint32_t ib) { return (ka < kb) || (ka == kb && ia < ib); }

#ifndef ES_WIDE_SORT_ASM
#define ES_WIDE_SORT_ASM 1
#endif
#if ES_WIDE_SORT_ASM
// Compare-exchange steps of the sort network, written out: path metrics are sums of non-negative penalties (or +inf for a dead
// lane), so their bit patterns order like their values and (metric, index) compares as the 96-bit unsigned number hi:lo:index --
// ONE borrow chain of three subtractions leaves "mine sorts before the partner's" in VCC, and with the partner in a lane that DPP
// reaches the partner's words are an operand modifier of the subtractions and of the three selects: 6 vector instructions per
// element and stage where the compiler's rendering of cand_before() + selects takes 10.  With the DPP modifier the subtraction is
// partner - mine (the modified operand is the minuend; tools/ub/ub_dppvcc.hip), so VCC = "the partner sorts before mine" and keeping
// mine is VCC xor take_min (equal elements are two dead lanes' identical fillers: either may be kept).  (s_nop 1: a DPP operand must
// not have been written by the two preceding vector instructions; the compiler cannot see into the block.)  tools/ub/ub_sortce.hip
// checks every step form against plain C++.  The scalar xor / xnor of the mask also writes SCC: it is in the clobber list ("scc") -- without
// it the compiler kept a scalar compare's result live across a step (round 4: `info_idx % 32 == 0` evaluated before the sort and selected
// on after it came out wrong in the 64-path instantiation, whose scheduling happened to put the compare there).
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
// the partner's words already fetched (ds_swizzle / ds_bpermute / LDS): keep mine where (mine < other) == take_min
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
// a lane's own two elements: swap where (element 1 < element 0) == ascending
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
#endif

// Bitonic network over 2L (key, index) pairs, element e = 2 * lane + b held as (k0, i0) / (k1, i1); ascending on exit.
// (LF < L: independent networks over aligned groups of LF lanes -- one per frame; pl = lane index within the frame)
template <int L, int LF>
__device__ __forceinline__ void wide_sort(double& k0, uint32_t& i0, double& k1, uint32_t& i1, const int p, const int pl, WideLds<L, (LF > 64 ? 2 : 1)>& W, int& buf)
{
#if ES_WIDE_SORT_ASM
    uint64_t u0, u1; __builtin_memcpy(&u0, &k0, 8); __builtin_memcpy(&u1, &k1, 8);
    uint32_t lo0 = (uint32_t)u0, hi0 = (uint32_t)(u0 >> 32), lo1 = (uint32_t)u1, hi1 = (uint32_t)(u1 >> 32);
    #pragma unroll
    for (int k = 2; k <= 2 * LF; k <<= 1) {
        const bool asc = ((2 * pl) & k) == 0;                    // k == 2 LF: always ascending
        #pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j == 1) {                                        // partner = the lane's other element
                ce_inlane(lo0, hi0, i0, lo1, hi1, i1, __builtin_amdgcn_ballot_w64(asc));
            } else {
                const int dl = j >> 1;                           // partner lane p ^ dl, same b
                const unsigned long long tm = __builtin_amdgcn_ballot_w64(((pl & dl) == 0) == asc);     // take the smaller of the pair
                if (dl == 1) { ES_CE_DPP("quad_perm:[1,0,3,2]", lo0, hi0, i0, tm); ES_CE_DPP("quad_perm:[1,0,3,2]", lo1, hi1, i1, tm); }
                else if (dl == 2) { ES_CE_DPP("quad_perm:[2,3,0,1]", lo0, hi0, i0, tm); ES_CE_DPP("quad_perm:[2,3,0,1]", lo1, hi1, i1, tm); }
                else if (dl == 8) { ES_CE_DPP("row_ror:8", lo0, hi0, i0, tm); ES_CE_DPP("row_ror:8", lo1, hi1, i1, tm); }
                else {
                    uint32_t a0 = 0, b0 = 0, c0 = 0, a1 = 0, b1 = 0, c1 = 0;
                    bool cross = false;
                    if constexpr (LF > 64) {
                        if (dl >= 64) {
                            cross = true;
                            W.pub.skey[buf][2 * p] = __builtin_bit_cast(double, ((uint64_t)hi0 << 32) | lo0);
                            W.pub.skey[buf][2 * p + 1] = __builtin_bit_cast(double, ((uint64_t)hi1 << 32) | lo1);
                            W.pub.sidx[buf][2 * p] = (uint16_t)i0; W.pub.sidx[buf][2 * p + 1] = (uint16_t)i1;
                            __syncthreads();
                            const int o = p ^ dl;
                            const uint64_t x0 = __builtin_bit_cast(uint64_t, W.pub.skey[buf][2 * o]), x1 = __builtin_bit_cast(uint64_t, W.pub.skey[buf][2 * o + 1]);
                            a0 = (uint32_t)x0; b0 = (uint32_t)(x0 >> 32); a1 = (uint32_t)x1; b1 = (uint32_t)(x1 >> 32);
                            c0 = W.pub.sidx[buf][2 * o]; c1 = W.pub.sidx[buf][2 * o + 1];
                            buf ^= 1;
                        }
                    }
                    if (!cross) {
                        a0 = (uint32_t)wxor_b32((int)lo0, dl); b0 = (uint32_t)wxor_b32((int)hi0, dl); c0 = (uint32_t)wxor_b32((int)i0, dl);
                        a1 = (uint32_t)wxor_b32((int)lo1, dl); b1 = (uint32_t)wxor_b32((int)hi1, dl); c1 = (uint32_t)wxor_b32((int)i1, dl);
                    }
                    ce_pair(lo0, hi0, i0, a0, b0, c0, tm);
                    ce_pair(lo1, hi1, i1, a1, b1, c1, tm);
                }
            }
        }
    }
    u0 = ((uint64_t)hi0 << 32) | lo0; u1 = ((uint64_t)hi1 << 32) | lo1;
    __builtin_memcpy(&k0, &u0, 8); __builtin_memcpy(&k1, &u1, 8);
#else
    #pragma unroll
    for (int k = 2; k <= 2 * LF; k <<= 1) {
        const bool asc = ((2 * pl) & k) == 0;                    // k == 2 LF: always ascending
        #pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j == 1) {                                        // partner = the lane's other element
                const bool sw = cand_before(k1, i1, k0, i0) == asc;
                const double tk = sw ? k1 : k0; const uint32_t ti = sw ? i1 : i0;
                k1 = sw ? k0 : k1; i1 = sw ? i0 : i1; k0 = tk; i0 = ti;
            } else {
                const int dl = j >> 1;                           // partner lane p ^ dl, same b
                const bool take_min = (((pl & dl) == 0) == asc);
                double ok0 = 0, ok1 = 0; uint32_t oi0 = 0, oi1 = 0;
                bool cross = false;
                if constexpr (LF > 64) {
                    if (dl >= 64) {
                        cross = true;
                        W.pub.skey[buf][2 * p] = k0; W.pub.skey[buf][2 * p + 1] = k1;
                        W.pub.sidx[buf][2 * p] = (uint16_t)i0; W.pub.sidx[buf][2 * p + 1] = (uint16_t)i1;
                        __syncthreads();
                        const int o = p ^ dl;
                        ok0 = W.pub.skey[buf][2 * o]; ok1 = W.pub.skey[buf][2 * o + 1];
                        oi0 = W.pub.sidx[buf][2 * o]; oi1 = W.pub.sidx[buf][2 * o + 1];
                        buf ^= 1;
                    }
                }
                if (!cross) {
                    ok0 = wxor_f64(k0, dl); ok1 = wxor_f64(k1, dl);
                    oi0 = (uint32_t)wxor_b32((int)i0, dl); oi1 = (uint32_t)wxor_b32((int)i1, dl);
                }
                const bool m0 = cand_before(k0, i0, ok0, oi0) == take_min;   // keep mine?
                const bool m1 = cand_before(k1, i1, ok1, oi1) == take_min;
                k0 = m0 ? k0 : ok0; i0 = m0 ? i0 : oi0;
                k1 = m1 ? k1 : ok1; i1 = m1 ? i1 : oi1;
            }
        }
    }
#endif
}

// cold path of the slab f loops: the whole level again with the generic softplus where a lane needs it (values identical for the others)
__device__ __attribute__((noinline)) void f_level_exact(const double* par, double* dst, int S, int L, const uint64_t* tab)
{
    for (int j = 0; j < S; ++j) dst[(long long)j * L] = es_polar_f(par[(long long)j * L], par[(long long)(j + S) * L], tab);
}

// hard decision -> butterfly -> data bits -> CRC of one frame by one wave (fastpolar.py:260-268); returns the CRC verdict to every lane
template <bool GK>
__device__ __forceinline__ int hard_decision_wave(const WideArgs& a, long long ff, int lane, uint32_t* hardw, uint8_t* hbytes, const uint16_t* dpos)
{
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
    if (lane < 32) hardw[lane] = word;
    wave_fence_lds();
    int ok = 0;
    if constexpr (GK) {
        // K data bits (K - 8 information bits, then the CRC's 8), MSB first in ceil(K / 8) bytes; neither count need be a whole number of bytes
        const int K = a.n_info, nbits = K - 8, nib = a.info_bytes, rem = nbits & 7;
        for (int bi = lane; bi < ((K + 7) >> 3); bi += 64) {
            uint32_t byte = 0;
            #pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int j = 8 * bi + b;
                const int pos = dpos[j < K ? j : 0];
                if (j < K) byte |= ((hardw[pos >> 5] >> (pos & 31)) & 1u) << (7 - b);
            }
            hbytes[bi] = (uint8_t)byte;
        }
        wave_fence_lds();
        if (lane == 0) {
            uint32_t reg = crc8_bytes(hbytes, nbits >> 3);
            const uint32_t lo = hbytes[nbits >> 3], hi = (((nbits >> 3) + 1) < ((K + 7) >> 3)) ? hbytes[(nbits >> 3) + 1] : 0u;
            for (int b = 0; b < rem; ++b) {
                reg ^= ((lo >> (7 - b)) & 1u) << 7;
                reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
            }
            ok = (reg == ((((lo << 8) | hi) >> (8 - rem)) & 0xffu));
        }
        ok = __shfl(ok, 0);
        for (int k = lane; k < nib; k += 64) {
            uint32_t v = hbytes[k];
            if (rem && k == nib - 1) v &= 0xffu << (8 - rem);          // the row is np.packbits(information bits): zero padding, not CRC bits
            a.hard_info[ff * nib + k] = (uint8_t)v;
        }
    } else {
        if (lane < 56) {
            uint32_t byte = 0;
            #pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int pos = dpos[8 * lane + b];
                byte |= ((hardw[pos >> 5] >> (pos & 31)) & 1u) << (7 - b);
            }
            hbytes[lane] = (uint8_t)byte;
        }
        wave_fence_lds();
        if (lane == 0) ok = (crc8_bytes(hbytes, ES_INFO_BYTES) == hbytes[ES_INFO_BYTES]);
        ok = __shfl(ok, 0);
        if (lane < ES_INFO_BYTES) a.hard_info[ff * ES_INFO_BYTES + lane] = hbytes[lane];
    }
    if (lane == 0) a.hard_ok[ff] = (uint8_t)ok;
    wave_fence_lds();
    return ok;
}

// L lanes per block, LF paths per frame.  LF > 64: one frame per block, the block is the group (barrier = __syncthreads).
// LF <= 64: every wave is a group of 64 / LF whole frames (barrier = wave fence), L / LF frames per block.
// GK: any code Polar(1024, K) + CRC-8, 9 <= K <= 1024 (the reference's PolarCode takes any K, rtwm/fastpolar.py:209-234): K, the row width
// and the number of trace-back windows are run-time values (a.n_info, a.info_bytes).  The default instantiation (K = 448, everything the
// reference itself instantiates) keeps them as compile-time constants: its code is unchanged by the existence of the other.
template <int L, int LF, bool GK = false>
__global__ __launch_bounds__(L, ES_WIDE_WPS) void es_scl_wide_kernel(WideArgs a)
{
    constexpr int MW = GK ? MWIN_MAX : MWIN_W;           // rows of the trace-back arrays in the aux slab
    const int NIB = GK ? a.info_bytes : ES_INFO_BYTES;   // bytes of a packed information row
    constexpr bool WAVE = (LF <= 64);                // the block is one wave
    constexpr int FRG = L / LF;                      // frames per block
    constexpr int NB = WAVE ? 1 : 2;
    static_assert(LF >= 1 && LF <= L && (L % 64) == 0 && L <= 256 && (WAVE ? L == 64 : L == LF), "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    WideLds<L, NB>& W = *reinterpret_cast<WideLds<L, NB>*>(smem_raw);
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio >= 3) __builtin_amdgcn_s_setprio(3);
    const int p = threadIdx.x;                       // slot == lane of the block
    const int lane = p & 63, wv = p >> 6;
    const int pl = p % LF;                           // path within its frame
    const int fp0 = p - pl;                          // slot of the frame's path 0
    using WLds = WideLds<L, NB>;
    static_assert(__builtin_offsetof(WLds, exp_tab) == 0, "the exp table must be the first member (ES_EXP_TAB_LDS_ADDR)");
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw != ES_EXP_TAB_LDS_ADDR) __builtin_trap();   // never: no static LDS in this file
    for (int i = p; i < ES_EXP_TAB_WORDS; i += L) W.exp_tab[i] = a.exp_tab[i];
    __syncthreads();
    auto group_sync = [&]() { if constexpr (WAVE) wave_fence_global(); else __syncthreads(); };
    const uint64_t* tab = W.exp_tab;
    // ---- claim a slab slot for this block (bit per slot; LDS and registers keep resident blocks <= slots).  Blocks are not
    // persistent: a block decodes its frames and leaves, so a launch has no tail of half-empty iterations and the wave slots it
    // frees go to whatever is queued (the front end of the next batches, the next group's list decoder).
    if (p == 0) {
        int slot = -1;
        unsigned w = blockIdx.x % (unsigned)a.slot_words;
        const bool drained = a.cursor && __hip_atomic_load(a.cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.B;   // (a launch is sized for B frames: blocks that come after the last frame was drawn leave at once)
        if (drained) slot = -2;
        for (int tries = 0; slot == -1 && tries < (1 << 22); ++tries) {
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
        W.flag = slot;
    }
    __syncthreads();
    const int slot = W.flag;
    __syncthreads();
    if (slot == -2) return;
    if (slot < 0) {                                                    // cannot happen (see the launcher); never spin forever, never fail silently:
        constexpr int FRB = FRG;                                        // the block's frames read ncand = -1
        if (a.cursor) { if (p < FRB) { const long long ff = atomicAdd(a.cursor, 1); if (ff < a.B) a.ncand[ff] = -1; } }
        else for (long long ff = (long long)blockIdx.x * FRB + p; p < FRB && ff < a.B; ff += a.B) a.ncand[ff] = -1;
        return;
    }
    double* const A = a.alpha + (long long)slot * N * L;             // element e of slot s at A[e*L + s]
    unsigned char* const aux = a.aux + (long long)slot * WIDE_AUX_PER_PATH * L;
    uint32_t* const TBW = reinterpret_cast<uint32_t*>(aux);                          // [14][L] window bits (first = MSB)
    uint32_t* const BG = TBW + MW * L;                                           // [28][L] partial-sum blocks of 128 (rows 0..3), 256 (4..11) and 512 bits (12..27), by slot
    uint32_t* const CB = BG + 28 * L;                                                // [16][L] fold scratch, own column only
    uint16_t* const TBA = reinterpret_cast<uint16_t*>(CB + 16 * L);                  // [14][L] path (within the frame) at the window's start

    // partial-sum word `wi` (block of 32*Wd bits at words [Wd, 2Wd)) of slot s
    auto beta_ld = [&](int wi, int s) -> uint32_t { return wi < 4 ? W.betaM[wi - 1][s] : BG[(wi - 4) * L + s]; };
    auto beta_st = [&](int wi, int s, uint32_t v) { if (wi < 4) W.betaM[wi - 1][s] = v; else BG[(wi - 4) * L + s] = v; };

    constexpr int DL = WLds::DL;                                                             // first LLR-tree depth held in LDS
    auto lrow = [](int d) constexpr { return wide_low_row(WLds::DL, d); };                   // its first row in W.low
    uint32_t* const hd_words = reinterpret_cast<uint32_t*>(&W.low[0][0]);                   // 32 words ...
    uint8_t* const hd_bytes = reinterpret_cast<uint8_t*>(&W.low[0][16]);                    // ... and up to 128 bytes of the hard decision (first wave; the rows are idle outside the list loop)
    int* const drawn = reinterpret_cast<int*>(&W.low[1][0]);                                 // frames drawn from the cursor (<= 64 ints), read back before the list loop writes the rows
    const long long n_groups = a.cursor ? (long long)gridDim.x : (a.B + FRG - 1) / FRG;      // (cursor: one draw per block)
    for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {      // one group of FRG frames per block (grid = groups)
        long long f; bool f_valid;
        uint64_t active_mask = 0;                              // block-uniform: frames that go through the list loop (up to 64 of them)
        if (a.cursor) {
            // ---------------- draw frames until FRG of them fail the hard decision (fastpolar.py:260-268) or none are left
            if constexpr (WAVE) {
                int n = 0;
                while (n < FRG) {
                    int ff = 0;
                    if (lane == 0) ff = atomicAdd(a.cursor, 1);
                    ff = __shfl(ff, 0);
                    if (ff >= a.B) break;
                    const int ok = hard_decision_wave<GK>(a, ff, lane, hd_words, hd_bytes, a.data_pos);
                    if (ok) {                                  // settled: its candidate rows read as zeros
                        if (lane == 0) a.ncand[ff] = 0;
                        #pragma unroll 1
                        for (int k = lane; k < a.lsz * NIB; k += 64) a.cand_info[(long long)ff * a.lsz * NIB + k] = 0;
                        #pragma unroll 1
                        for (int k = lane; k < a.lsz; k += 64) { a.cand_metric[(long long)ff * a.lsz + k] = 0.0; a.cand_ok[(long long)ff * a.lsz + k] = 0; }
                    } else { if (lane == 0) drawn[n] = ff; ++n; }
                }
                wave_fence_lds();
                if (n == 0) break;
                f_valid = (lane / LF) < n;
                f = drawn[f_valid ? lane / LF : 0];             // a missing frame mirrors the first one (never stored)
                active_mask = (n >= 64) ? ~0ULL : ((1ULL << n) - 1ULL);
            } else {
                int ff;
                for (;;) {
                    if (p == 0) drawn[0] = atomicAdd(a.cursor, 1);
                    __syncthreads();
                    ff = drawn[0];
                    if (ff >= a.B) break;
                    if (wv == 0) {
                        const int ok = hard_decision_wave<GK>(a, ff, lane, hd_words, hd_bytes, a.data_pos);
                        if (lane == 0) W.flag = ok;
                    }
                    __syncthreads();
                    const int ok = W.flag;
                    __syncthreads();
                    if (!ok) break;
                    if (p == 0) a.ncand[ff] = 0;
                    for (int k = p; k < a.lsz * NIB; k += L) a.cand_info[(long long)ff * a.lsz * NIB + k] = 0;
                    if (p < a.lsz) { a.cand_metric[(long long)ff * a.lsz + p] = 0.0; a.cand_ok[(long long)ff * a.lsz + p] = 0; }
                }
                if (ff >= a.B) break;
                f = ff; f_valid = true; active_mask = 1ULL;
            }
        } else {
        const long long f_raw = g * FRG + (WAVE ? lane / LF : 0);
        f_valid = f_raw < a.B;
        f = f_valid ? f_raw : a.B - 1;          // a missing frame mirrors the last one (never stored)

        // ---------------- hard decision (fastpolar.py:260-268): frame by frame, one wave each
        if constexpr (WAVE) {
            for (int fi = 0; fi < FRG; ++fi) {
                const long long ff = g * FRG + fi;
                if (ff >= a.B) break;
                const int ok = hard_decision_wave<GK>(a, ff, lane, hd_words, hd_bytes, a.data_pos);
                if (ok && a.skip_if_hard_ok) {                 // no list for this record: its candidate rows read as zeros
                    if (lane == 0) a.ncand[ff] = 0;
                    #pragma unroll 1
                    for (int k = lane; k < a.lsz * NIB; k += 64) a.cand_info[ff * a.lsz * NIB + k] = 0;
                    #pragma unroll 1
                    for (int k = lane; k < a.lsz; k += 64) { a.cand_metric[ff * a.lsz + k] = 0.0; a.cand_ok[ff * a.lsz + k] = 0; }
                } else active_mask |= 1ULL << fi;
            }
        } else {
            if (wv == 0) {
                const int ok = hard_decision_wave<GK>(a, f, lane, hd_words, hd_bytes, a.data_pos);
                if (lane == 0) W.flag = ok;
            }
            __syncthreads();
            if (W.flag && a.skip_if_hard_ok) {
                if (p == 0) a.ncand[f] = 0;
                for (int k = p; k < a.lsz * NIB; k += L) a.cand_info[f * a.lsz * NIB + k] = 0;
                if (p < a.lsz) { a.cand_metric[f * a.lsz + p] = 0.0; a.cand_ok[f * a.lsz + p] = 0; }
            } else active_mask = 1ULL;
            __syncthreads();
        }
        }
        const float* llr32 = (const float*)a.llr + f * N;
        const double* llr64 = (const double*)a.llr + f * N;
        if (active_mask == 0) continue;
        const bool f_store = f_valid && ((active_mask >> (WAVE ? lane / LF : 0)) & 1ULL);

        // ---------------- list decoding (fastpolar.py:278-330)
        // slot of my LLR block at depth d = 1..8: byte d-1 (every path starts as a copy of its frame's path 0)
        uint64_t pa_ = (uint64_t)(uint32_t)fp0 * 0x0101010101010101ULL;
        // slot of my partial-sum block at depth d = 1..5: byte d-1; byte 5: LLR slot at depth 9
        uint64_t pb_ = (uint64_t)(uint32_t)fp0 * 0x0000010101010101ULL;
        uint32_t b0 = 0;                  // partial-sum blocks of S = 1..16 bits, block of S bits at bit S
        uint32_t hist = 0, anc = 0;       // trace-back window
        double metric = 0.0, lam = 0.0, sp_diff = 0.0, sp_sum = 0.0, lp_odd = 0.0;
        int cnt = 1, info_idx = 0;
        bool dirty = false;               // shared slots may still be being read by a wave that is behind (group-uniform)
        group_sync();

        for (int i = 0; i < N; ++i) {
            const bool frozen = (a.frozen.w[i >> 5] >> (i & 31)) & 1u;
            if (i == 0) {
                // First chain: the paths of a frame are still copies of its path 0, so each node is computed once, the frame's
                // lanes sharing its elements, into that path's slot (a barrier per depth here: lanes read what other lanes wrote).
                #pragma unroll 1
                for (int d = 1; d <= 9; ++d) {
                    const int S = N >> d;
                    const int rs = wide_low_row(DL, d - 1), rd = wide_low_row(DL, d);      // rows of depth d-1 / d when they are in LDS
                    #pragma unroll 1
                    for (int j = pl; j < S; j += LF) {
                        double pa, pb;
                        if (d == 1) { pa = a.is_f64 ? llr64[j] : (double)llr32[j]; pb = a.is_f64 ? llr64[j + S] : (double)llr32[j + S]; }
                        else if (d <= DL) { pa = A[(long long)(2 * S + j) * L + fp0]; pb = A[(long long)(2 * S + j + S) * L + fp0]; }
                        else { pa = W.low[rs + j][fp0]; pb = W.low[rs + j + S][fp0]; }
                        const double v = es_polar_f(pa, pb, tab);
                        if (d < DL) A[(long long)(S + j) * L + fp0] = v; else W.low[rd + j][fp0] = v;
                    }
                    group_sync();
                }
                lam = es_polar_f_sp(W.low[lrow(9)][fp0], W.low[lrow(9) + 1][fp0], tab, &sp_diff, &sp_sum);
                dirty = true;
            } else {
                const int top = NLEV - __builtin_ctz((unsigned)i);
                if (top <= 9 && dirty) { group_sync(); dirty = false; }         // this step writes slots
                // --- depths top..DL-1: slab to slab, the lane walks the node
                bool have_dl = false;                                               // depth DL already formed (in LDS) by a fused pass of this step
                int d_first = top;
#if ES_WIDE_L1R
                // Depth 1 of the SECOND half of the code word (bits 512..) is g(llr[e], llr[e + 512], u1[e]) = llr[e + 512] +- llr[e]: one bit of the
                // 512-bit partial-sum block picks one of two values that the frame's channel LLRs already hold.  It is never stored (512 doubles written
                // and twice read per path otherwise: 8.5 % of the slab traffic, and a 512-element load-add-store loop): the two passes that consume it --
                // the f level of depth 2 at bit 512, the g level of depth 2 (fused with the f level of depth 3) at bit 768 -- form its elements on the
                // fly from the channel LLRs (shared by the frame's paths: the lanes of a frame read the same addresses).
                // (float32 channel LLRs -- what es_llr_batch produces; a float64 launch keeps depth 1 in the slab.  The operands come from the L1 / L2,
                // not from the slab: one operand set ahead, rotated by copy, is enough.)
                if ((i == 512 || i == 768) && !a.is_f64) {
                    const int bs1 = p8_get(pb_, 0);                                 // slot of my 512-bit partial-sum block
                    double* const dst2 = A + (long long)256 * L + p;                // depth 2 at elements [256, 512)
                    auto g1 = [&](float lo, float hi, uint32_t u) { return es_polar_g((double)lo, (double)hi, u); };      // depth-1 element from llr[e], llr[e + 512]
                    if (i == 512) {
                        auto ld = [&](int j, float (&q)[4]) { q[0] = llr32[j]; q[1] = llr32[j + 512]; q[2] = llr32[j + 256]; q[3] = llr32[j + 768]; };
                        float q[4], qn[4];
                        ld(0, q);
                        #pragma unroll 1
                        for (int j0 = 0; j0 < 256; j0 += 32) {
                            const uint32_t wa = beta_ld((512 + j0) >> 5, bs1), wb = beta_ld((512 + j0 + 256) >> 5, bs1);
                            #pragma unroll 1
                            for (int u = 0; u < 32; ++u) {
                                const int j = j0 + u;
                                ld(j < 255 ? j + 1 : 255, qn);
                                dst2[(long long)j * L] = es_polar_f(g1(q[0], q[1], (wa >> u) & 1u), g1(q[2], q[3], (wb >> u) & 1u), tab);
                                #pragma unroll
                                for (int k = 0; k < 4; ++k) q[k] = qn[k];
                            }
                        }
                    } else {
                        const int bs2 = p8_get(pb_, 1);                             // slot of my 256-bit block
                        double* const dst3 = A + (long long)128 * L + p;            // depth 3 at elements [128, 256)
                        auto ld = [&](int j, float (&q)[8]) {
                            q[0] = llr32[j]; q[1] = llr32[j + 512]; q[2] = llr32[j + 256]; q[3] = llr32[j + 768];
                            q[4] = llr32[j + 128]; q[5] = llr32[j + 640]; q[6] = llr32[j + 384]; q[7] = llr32[j + 896];
                        };
                        float q[8], qn[8];
                        ld(0, q);
                        #pragma unroll 1
                        for (int j0 = 0; j0 < 128; j0 += 32) {
                            const uint32_t w1a = beta_ld((512 + j0) >> 5, bs1), w1b = beta_ld((512 + j0 + 256) >> 5, bs1);
                            const uint32_t w1c = beta_ld((512 + j0 + 128) >> 5, bs1), w1d = beta_ld((512 + j0 + 384) >> 5, bs1);
                            const uint32_t w2a = beta_ld((256 + j0) >> 5, bs2), w2b = beta_ld((256 + j0 + 128) >> 5, bs2);
                            #pragma unroll 1
                            for (int u = 0; u < 32; ++u) {
                                const int j = j0 + u;
                                ld(j < 127 ? j + 1 : 127, qn);
                                const double x = es_polar_g(g1(q[0], q[1], (w1a >> u) & 1u), g1(q[2], q[3], (w1b >> u) & 1u), (w2a >> u) & 1u);   // depth-2 element j
                                const double y = es_polar_g(g1(q[4], q[5], (w1c >> u) & 1u), g1(q[6], q[7], (w1d >> u) & 1u), (w2b >> u) & 1u);   // ... and j + 128
                                dst2[(long long)j * L] = x;
                                dst2[(long long)(j + 128) * L] = y;
                                dst3[(long long)j * L] = es_polar_f(x, y, tab);
                                #pragma unroll
                                for (int k = 0; k < 8; ++k) q[k] = qn[k];
                            }
                        }
                    }
                    pa_ = p8_set(pa_, 1, p);
                    d_first = 3;
                    if (i == 768) { pa_ = p8_set(pa_, 2, p); d_first = 4; }
                }
#endif
                for (int d = d_first; d < DL; ++d) {
                    const int S = N >> d;
                    const bool is_g = (i >> (NLEV - d)) & 1;
                    const int ps = (d > 1) ? p8_get(pa_, d - 2) : 0;
                    const double* par = A + (long long)(2 * S) * L + ps;          // depth d-1 block at elements [2S, 4S)
                    double* dst = A + (long long)S * L + p;                        // depth d block at elements [S, 2S)
                    // The loaders are chosen OUTSIDE the element loops (slab / float32 LLRs / float64 LLRs): a branch inside a loader puts
                    // control flow between a load and its use, and the compiler then waits for ALL outstanding loads (s_waitcnt vmcnt(0)) right
                    // after issuing each pair -- the batches below would run one memory round trip per pair instead of GBATCH pairs in flight.
                    auto ld_slab = [&](int j, double& pa, double& pb) { pa = par[(long long)j * L]; pb = par[(long long)(j + S) * L]; };
#if ES_WIDE_FUSE_GF
                    if (d > 1 && d < (DL < 7 ? DL : 7) && (is_g || (ES_WIDE_FUSE_GF & 2) || ((ES_WIDE_FUSE_GF & 4) && d + 1 == DL))) {
                        // Two levels in ONE pass: level d (the g of the bit just decided when it is the top of the step, else an f) and the f
                        // level below it.  The two level-d results that make an f operand pair -- elements i and i + S/2 -- are formed
                        // together, stored (their g child reads them later) and consumed from registers, so that level d+1 does not read
                        // back what was written S elements earlier (at these sizes from the Infinity Cache or HBM).
                        const int bs = (is_g && d <= 5) ? p8_get(pb_, d - 1) : 0;
                        const int H = S >> 1;
                        double* dstB = A + (long long)H * L + p;                   // depth d+1 block at elements [H, 2H)
                        auto ld4 = [&](int j, double (&q)[4]) {
                            q[0] = par[(long long)j * L]; q[1] = par[(long long)(j + S) * L];
                            q[2] = par[(long long)(j + H) * L]; q[3] = par[(long long)(j + H + S) * L];
                        };
                        // (the lower level's destination -- slab, or LDS when it is depth DL -- is chosen outside the element loops, like the loaders)
                        auto fused_pass = [&](auto stB) {
                        auto pair_gf = [&](int j, const double (&q)[4], uint32_t u1, uint32_t u2) {
                            const double g1 = es_polar_g(q[0], q[1], u1), g2 = es_polar_g(q[2], q[3], u2);
                            dst[(long long)j * L] = g1;
                            dst[(long long)(j + H) * L] = g2;
                            stB(j, es_polar_f(g1, g2, tab));
                        };
                        auto pair_ff = [&](int j, const double (&q)[4]) {
                            const double f1 = es_polar_f(q[0], q[1], tab);
                            dst[(long long)j * L] = f1;
                            const double f2 = es_polar_f(q[2], q[3], tab);
                            dst[(long long)(j + H) * L] = f2;
                            stB(j, es_polar_f(f1, f2, tab));
                        };
                        const int blk = H < 32 ? H : 32;
                        for (int i0 = 0; i0 < H; i0 += blk) {
                            uint32_t w1 = 0, w2 = 0;                                // partial-sum bits of elements i0 .. and i0 + H ..
                            if (is_g) {
                                if (S >= 64) { w1 = beta_ld((S + i0) >> 5, bs); w2 = beta_ld((S + i0 + H) >> 5, bs); }
                                else if (S == 32) { w1 = beta_ld(1, bs); w2 = w1 >> 16; }
                                else { w1 = b0 >> S; w2 = w1 >> H; }
                            }
                            double q0[4], q1[4];
                            ld4(i0, q0); ld4(i0 + 1, q1);
                            if (is_g) {
                                for (int u = 0; u < blk; u += 2) {                  // two operand sets in rotation (renamed, not copied)
                                    pair_gf(i0 + u, q0, (w1 >> u) & 1u, (w2 >> u) & 1u);
                                    ld4(u + 2 < blk ? i0 + u + 2 : i0 + blk - 1, q0);
                                    pair_gf(i0 + u + 1, q1, (w1 >> (u + 1)) & 1u, (w2 >> (u + 1)) & 1u);
                                    ld4(u + 3 < blk ? i0 + u + 3 : i0 + blk - 1, q1);
                                }
                            } else {
                                for (int u = 0; u < blk; u += 2) {
                                    pair_ff(i0 + u, q0);
                                    ld4(u + 2 < blk ? i0 + u + 2 : i0 + blk - 1, q0);
                                    pair_ff(i0 + u + 1, q1);
                                    ld4(u + 3 < blk ? i0 + u + 3 : i0 + blk - 1, q1);
                                }
                            }
                        }
                        };
                        bool to_lds = false;
                        if constexpr (DL <= 7) to_lds = (d + 1 == DL);
                        if (to_lds) { fused_pass([&](int j, double v) { W.low[j][p] = v; }); have_dl = true; }       // depth DL: rows 0 ..
                        else fused_pass([&](int j, double v) { dstB[(long long)j * L] = v; });
                        pa_ = p8_set(pa_, d - 1, p);
                        ++d;                                                        // level d+1 is done too
                        pa_ = p8_set(pa_, d - 1, p);
                        continue;
                    }
#endif
                    if (is_g) {
                        const int bs = (d <= 5) ? p8_get(pb_, d - 1) : 0;
                        auto g_level = [&](auto ld) {
                            for (int j0 = 0; j0 < S; j0 += 32) {
                                uint32_t wbits; int nb;
                                if (S >= 32) { wbits = beta_ld((S + j0) >> 5, bs); nb = 32; }
                                else { wbits = b0 >> S; nb = S; }
                                for (int u = 0; u < nb; u += ES_WIDE_GBATCH) {     // (g is an add: the loop is memory latency) GBATCH independent load pairs in flight
                                    double xa[ES_WIDE_GBATCH], xb[ES_WIDE_GBATCH];
                                    #pragma unroll
                                    for (int v = 0; v < ES_WIDE_GBATCH; ++v) ld(j0 + u + v, xa[v], xb[v]);
                                    #pragma unroll
                                    for (int v = 0; v < ES_WIDE_GBATCH; ++v) dst[(long long)(j0 + u + v) * L] = es_polar_g(xa[v], xb[v], (wbits >> (u + v)) & 1u);
                                }
                            }
                        };
                        if (d > 1) g_level(ld_slab);
                        else if (a.is_f64) g_level([&](int j, double& pa, double& pb) { pa = llr64[j]; pb = llr64[j + S]; });      // d == 1 (i == 512): the channel LLRs
                        else g_level([&](int j, double& pa, double& pb) { pa = (double)llr32[j]; pb = (double)llr32[j + S]; });
                    } else {
                        // an f level is never the top of a step (the top is the g whose bit was just decided): d >= 2, operands from the slab
#if ES_WIDE_FDIST == 2
                        // three operand pairs in rotation, the loop unrolled by three so that the rotation is a renaming, not a copy (a copy of a
                        // register that a load is still filling waits for the load): the operands of element j + 3 are requested right after
                        // f(j) and used two f evaluations later.  Loads past the end re-read the last element (S >= 8).
                        // The loop body is straight-line: operands outside the range of the straight-line softplus (|t| >= 512) only raise `bad`, and the level is then redone with the generic form in a cold loop after it.
#if ES_WIDE_DEFER & 1
                        double a0, b0_, a1, b1_, a2, b2_;
                        int bad = 0;
                        ld_slab(0, a0, b0_); ld_slab(1, a1, b1_); ld_slab(2, a2, b2_);
                        for (int j = 0; ; j += 3) {
                            dst[(long long)j * L] = es_polar_f_fast(a0, b0_, tab, &bad);
                            if (j + 1 >= S) break;
                            ld_slab(j + 3 < S ? j + 3 : S - 1, a0, b0_);
                            dst[(long long)(j + 1) * L] = es_polar_f_fast(a1, b1_, tab, &bad);
                            if (j + 2 >= S) break;
                            ld_slab(j + 4 < S ? j + 4 : S - 1, a1, b1_);
                            dst[(long long)(j + 2) * L] = es_polar_f_fast(a2, b2_, tab, &bad);
                            if (j + 3 >= S) break;
                            ld_slab(j + 5 < S ? j + 5 : S - 1, a2, b2_);
                        }
                        if (__builtin_amdgcn_ballot_w64(bad != 0) != 0ULL) f_level_exact(par, dst, S, L, tab);     // rare
#else
                        double a0, b0_, a1, b1_, a2, b2_;
                        ld_slab(0, a0, b0_); ld_slab(1, a1, b1_); ld_slab(2, a2, b2_);
                        for (int j = 0; ; j += 3) {
                            dst[(long long)j * L] = es_polar_f(a0, b0_, tab);
                            if (j + 1 >= S) break;
                            ld_slab(j + 3 < S ? j + 3 : S - 1, a0, b0_);
                            dst[(long long)(j + 1) * L] = es_polar_f(a1, b1_, tab);
                            if (j + 2 >= S) break;
                            ld_slab(j + 4 < S ? j + 4 : S - 1, a1, b1_);
                            dst[(long long)(j + 2) * L] = es_polar_f(a2, b2_, tab);
                            if (j + 3 >= S) break;
                            ld_slab(j + 5 < S ? j + 5 : S - 1, a2, b2_);
                        }
#endif
#else
                        double pa, pb, qa, qb;                                   // operands of elements j+1 and j+2 are on their way while f(j) runs (S >= 8)
                        ld_slab(0, pa, pb); ld_slab(1, qa, qb);
                        for (int j = 0; j < S - 2; ++j) {
                            double na, nb;
                            ld_slab(j + 2, na, nb);
                            dst[(long long)j * L] = es_polar_f(pa, pb, tab);
                            pa = qa; pb = qb; qa = na; qb = nb;
                        }
                        dst[(long long)(S - 2) * L] = es_polar_f(pa, pb, tab);
                        dst[(long long)(S - 1) * L] = es_polar_f(qa, qb, tab);
#endif
                    }
                    pa_ = p8_set(pa_, d - 1, p);
                }
                // --- depths DL..9 in LDS: level DL from the slab (unless a fused pass has just formed it), the others LDS -> LDS; S = 1024 >> d
                // elements by slot.  (ES_WIDE_DEFER: the generic softplus as a cold path after the level instead of a branch per evaluation --
                // bit 1 depth 8, bit 2 depth 9; levels of 8 or 16 elements always.)
                auto low_level = [&](auto dc, auto from_slab) {
                    constexpr int d = decltype(dc)::value;
                    constexpr int S = N >> d;
                    constexpr bool SLAB = decltype(from_slab)::value;
                    constexpr bool DEFER = S > 4 || (d == 8 ? (ES_WIDE_DEFER & 2) : (ES_WIDE_DEFER & 4)) != 0;
                    constexpr int rs = wide_low_row(DL, d - 1), rd = wide_low_row(DL, d);
                    constexpr int C = S < 4 ? S : 4;                                  // f evaluations per chunk (registers are indexed statically: no private arrays)
                    const int ps = p8_get(pa_, d - 2);                                // slot of my depth d-1 block
                    const double* par = A + (long long)(2 * S) * L + ps;              // (SLAB: depth d-1 at elements [2S, 4S))
                    auto src = [&](int e) -> double { if constexpr (SLAB) return par[(long long)e * L]; else return W.low[rs + e][ps]; };
                    if ((i >> (NLEV - d)) & 1) {
                        double x[2 * S];
                        #pragma unroll
                        for (int u = 0; u < 2 * S; ++u) x[u] = src(u);
                        #pragma unroll
                        for (int j = 0; j < S; ++j) W.low[rd + j][p] = es_polar_g(x[j], x[j + S], (b0 >> (S + j)) & 1u);
                    } else if constexpr (DEFER) {
                        int bad = 0;
                        #pragma unroll 1
                        for (int c = 0; c < S; c += C) {
                            double x[2 * C];
                            #pragma unroll
                            for (int u = 0; u < C; ++u) { x[u] = src(c + u); x[C + u] = src(c + u + S); }
                            #pragma unroll
                            for (int u = 0; u < C; ++u) W.low[rd + c + u][p] = es_polar_f_fast(x[u], x[C + u], tab, &bad);
                        }
                        if (__builtin_amdgcn_ballot_w64(bad != 0) != 0ULL) {          // rare: the level again with the generic form
                            #pragma unroll 1
                            for (int j = 0; j < S; ++j) W.low[rd + j][p] = es_polar_f(src(j), src(j + S), tab);
                        }
                    } else {
                        double x[2 * S];
                        #pragma unroll
                        for (int u = 0; u < 2 * S; ++u) x[u] = src(u);
                        #pragma unroll 2
                        for (int j = 0; j < S; ++j) W.low[rd + j][p] = es_polar_f(x[j], x[j + S], tab);
                    }
                    if constexpr (d <= 8) pa_ = p8_set(pa_, d - 1, p); else pb_ = p8_set(pb_, 5, p);
                };
                using std::integral_constant;
                if (top <= DL && !have_dl) low_level(integral_constant<int, DL>{}, std::true_type{});
                if constexpr (DL < 7) { if (top <= 7) low_level(integral_constant<int, 7>{}, std::false_type{}); }
                if constexpr (DL < 8) { if (top <= 8) low_level(integral_constant<int, 8>{}, std::false_type{}); }
                if (top <= 9) low_level(integral_constant<int, 9>{}, std::false_type{});
                // --- depth 10: LDS -> register
                {
                    const int ps = p8_get(pb_, 5);
                    const double xa = W.low[lrow(9)][ps], xb = W.low[lrow(9) + 1][ps];
                    if (i & 1) lam = es_polar_g(xa, xb, (b0 >> 1) & 1u);
#if ES_WIDE_DEFER & 8
                    else {
                        int bad = 0;
                        lam = es_polar_f_fast_sp(xa, xb, tab, &sp_diff, &sp_sum, &bad);
                        if (__builtin_amdgcn_ballot_w64(bad != 0) != 0ULL) lam = es_polar_f_sp(xa, xb, tab, &sp_diff, &sp_sum);
                    }
#else
                    else lam = es_polar_f_sp(xa, xb, tab, &sp_diff, &sp_sum);
#endif
                }
                dirty = true;
            }

            // --- decision
            const double al = __builtin_fabs(lam);
            double lp;
            if (i & 1) lp = lp_odd;                                  // set when the even sibling was decided
            else lp = es_softplus_neg(-al, tab);
            const uint32_t pref = (lam >= 0.0) ? 1u : 0u;
            uint32_t bit = 0;
            if (frozen) {                                             // fastpolar.py:281-286
                double pen = lp;
                if (pref != 0u) pen = lp + al;
                metric = metric + pen;
                lp_odd = sp_sum;                                      // sibling g = b + a when this bit is 0
            } else {                                                  // fastpolar.py:288-330
                const bool live = pl < cnt;
                const int nc = 2 * cnt;
                double k0 = live ? metric + ((pref != 0u) ? lp + al : lp) : __builtin_inf();
                double k1 = live ? metric + ((pref != 1u) ? lp + al : lp) : __builtin_inf();
                uint32_t i0 = live ? (uint32_t)(2 * pl) : 0xFFFFu, i1 = live ? (uint32_t)(2 * pl + 1) : 0xFFFFu;   // candidate index within the frame
                const bool wstart = (info_idx & 31) == 0;
                const int keep = nc < a.lsz ? nc : a.lsz;          // a.lsz <= LF: lists of any size run on the next power of two's kernel
                const int myr = pl < keep ? pl : 0;                // dead paths mirror rank 0
                if constexpr (WAVE && ES_WIDE_GATHER_BPERM) {
                    // One wave: the survivor of rank r takes sorted element r and then its parent's state straight from the registers of the
                    // lanes that hold them (ds_bpermute: no LDS storage, no fence -- the lanes of a wave run in lock step).
                    int buf = 0;
                    wide_sort<L, LF>(k0, i0, k1, i1, p, pl, W, buf);
                    auto fetch32 = [](int src, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); };
                    auto fetch64 = [&](int src, uint64_t v) { return ((uint64_t)fetch32(src, (uint32_t)(v >> 32)) << 32) | fetch32(src, (uint32_t)v); };
                    const int holder = fp0 + (myr >> 1);           // sorted element e sits in lane e / 2 of the frame, half e % 2
                    const uint64_t e0 = fetch64(holder, __builtin_bit_cast(uint64_t, k0)), e1 = fetch64(holder, __builtin_bit_cast(uint64_t, k1));
                    const uint32_t ei = fetch32(holder, (i0 & 0xFFFFu) | (i1 << 16));
                    metric = __builtin_bit_cast(double, (myr & 1) ? e1 : e0);
                    const uint32_t myc = (myr & 1) ? (ei >> 16) : (ei & 0xFFFFu);
                    const int parent = fp0 + ((int)(myc >> 1) & (LF - 1));
                    bit = myc & 1u;
                    const uint64_t t = fetch64(parent, pb_ | ((uint64_t)(wstart ? (uint32_t)pl : anc) << 48));
                    pa_ = fetch64(parent, pa_);
                    pb_ = t & 0xFFFFFFFFFFFFULL;
                    anc = (uint32_t)(t >> 48);
                    b0 = fetch32(parent, b0);
                    hist = (fetch32(parent, wstart ? 0u : hist) << 1) | bit;
                    if (!(i & 1)) {
                        const uint64_t sd = fetch64(parent, __builtin_bit_cast(uint64_t, sp_diff)), ss = fetch64(parent, __builtin_bit_cast(uint64_t, sp_sum));
                        lp_odd = __builtin_bit_cast(double, bit ? sd : ss);
                    }
                } else {
                // publish what follows a path through the sort
                const int xb = (NB == 2) ? (info_idx & 1) : 0;
                W.pub.xpa[xb][p] = pa_;
                W.pub.xpb[xb][p] = pb_ | ((uint64_t)(wstart ? (uint32_t)pl : anc) << 48);
                W.pub.xb0[xb][p] = b0;
                W.pub.xhist[xb][p] = wstart ? 0u : hist;
                if (!(i & 1)) { W.pub.xsp[xb][0][p] = sp_diff; W.pub.xsp[xb][1][p] = sp_sum; }
                int buf = 0;
                wide_sort<L, LF>(k0, i0, k1, i1, p, pl, W, buf);
                W.pub.skey[buf][2 * p] = k0; W.pub.skey[buf][2 * p + 1] = k1;
                W.pub.sidx[buf][2 * p] = (uint16_t)i0; W.pub.sidx[buf][2 * p + 1] = (uint16_t)i1;
                if constexpr (WAVE) wave_fence_lds(); else __syncthreads();
                const uint32_t myc = W.pub.sidx[buf][2 * fp0 + myr];
                metric = W.pub.skey[buf][2 * fp0 + myr];
                const int parent = fp0 + ((int)(myc >> 1) & (LF - 1));
                bit = myc & 1u;
                pa_ = W.pub.xpa[xb][parent];
                const uint64_t t = W.pub.xpb[xb][parent];
                pb_ = t & 0xFFFFFFFFFFFFULL;
                anc = (uint32_t)(t >> 48);
                b0 = W.pub.xb0[xb][parent];
                hist = (W.pub.xhist[xb][parent] << 1) | bit;
                if (!(i & 1)) lp_odd = W.pub.xsp[xb][bit ? 0 : 1][parent];
                }
                if ((info_idx & 31) == 31) { TBW[(info_idx >> 5) * L + p] = hist; TBA[(info_idx >> 5) * L + p] = (uint16_t)anc; }
                cnt = keep;
                ++info_idx;
                if constexpr (!WAVE) dirty = false;                // every wave passed the sort's barriers after its reads
                else if constexpr (!ES_WIDE_GATHER_BPERM) wave_fence_lds();   // (one wave: the published rows are rewritten two sorts later, in order)
            }

            // --- partial sums: fold upward while the node is a right child (fastpolar.py:156-183); each lane its own path
            const int t = __builtin_ctz(~(unsigned)i);
            if (t < NLEV) {
                uint32_t cw = bit;
                const int t5 = t < 5 ? t : 5;
                for (int s = 0; s < t5; ++s) {
                    const int S = 1 << s;
                    const uint32_t left = (b0 >> S) & ((1u << S) - 1u);
                    cw = (left ^ cw) | (cw << S);
                }
                if (t < 5) {
                    const int Sp = 1 << t;
                    const uint32_t mask = ((1u << Sp) - 1u) << Sp;
                    b0 = (b0 & ~mask) | (cw << Sp);
                } else {
                    if (t > 5) {
                        CB[p] = cw;
                        for (int s = 5; s < t; ++s) {
                            const int Wd = 1 << (s - 5);
                            const int bs = p8_get(pb_, NLEV - s - 1);
                            for (int w = 0; w < Wd; ++w) {
                                const uint32_t c0 = CB[w * L + p];
                                const uint32_t lf = beta_ld(Wd + w, bs);
                                CB[(Wd + w) * L + p] = c0;
                                CB[w * L + p] = c0 ^ lf;
                            }
                        }
                    }
                    if (dirty) { group_sync(); dirty = false; }     // the block about to be rewritten may still be being read
                    const int Wp = 1 << (t - 5);
                    if (t == 5) beta_st(1, p, cw);
                    else for (int w = 0; w < Wp; ++w) beta_st(Wp + w, p, CB[w * L + p]);
                    pb_ = p8_set(pb_, NLEV - t - 1, p);
                }
            }
        }

        // ---------------- final ordering (fastpolar.py:335), trace-back, CRC -- per frame
        group_sync();
        W.low[0][p] = metric;                                       // (the list loop is over: the LLR rows are free; a lane's own cell -- waves that are groups of their own run out of step)
        group_sync();
        int rank = 0;
        for (int k = 0; k < cnt; ++k) {
            const double mk = W.low[0][fp0 + k];
            rank += ((mk < metric) || (mk == metric && k < pl)) ? 1 : 0;
        }
        if constexpr (GK) {
            const int r = a.n_info & 31;                             // a last, partial window: its bits move up to the MSB end like a full one's
            if (r) { TBW[(a.n_info >> 5) * L + p] = hist << (32 - r); TBA[(a.n_info >> 5) * L + p] = (uint16_t)anc; }
            group_sync();
        }
        if (pl < cnt && f_store) {
            uint8_t* out = a.cand_info + (f * a.lsz + rank) * NIB;
            int curp = pl;
            if constexpr (GK) {
                const int nw = (a.n_info + 31) >> 5, nbits = a.n_info - 8, rem = nbits & 7, kc = nbits >> 3;
                uint32_t reg = 0, lo = 0, hi = 0;                    // lo, hi: the two bytes the 8 CRC bits lie in
                #pragma unroll 1
                for (int w = nw - 1; w >= 0; --w) {                  // windows are read last to first (each names its ancestor in the one before) ...
                    const uint32_t wd = TBW[w * L + fp0 + curp];
                    curp = (int)TBA[w * L + fp0 + curp] & (LF - 1);
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = 4 * w + q;
                        const uint32_t byte = (wd >> (24 - 8 * q)) & 0xffu;
                        if (k == kc) lo = byte; else if (k == kc + 1) hi = byte;
                        if (k < NIB) out[k] = (uint8_t)((rem && k == NIB - 1) ? (byte & (0xffu << (8 - rem))) : byte);
                    }
                }
                #pragma unroll 1
                for (int k = 0; k < kc; ++k) {                       // ... and the CRC runs first to last over the row just written (this lane's own bytes)
                    reg ^= out[k];
                    #pragma unroll
                    for (int b = 0; b < 8; ++b) reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
                }
                for (int b = 0; b < rem; ++b) {
                    reg ^= ((lo >> (7 - b)) & 1u) << 7;
                    reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
                }
                a.cand_metric[f * a.lsz + rank] = metric;
                a.cand_ok[f * a.lsz + rank] = (uint8_t)(reg == ((((lo << 8) | hi) >> (8 - rem)) & 0xffu));
            } else {
            uint32_t wd[MWIN_W];
            #pragma unroll
            for (int w = MWIN_W - 1; w >= 0; --w) {
                wd[w] = TBW[w * L + fp0 + curp];                  // information bits 32w .. 32w+31, first = MSB
                curp = (int)TBA[w * L + fp0 + curp] & (LF - 1);
            }
            uint32_t reg = 0;
            #pragma unroll
            for (int k = 0; k < ES_INFO_BYTES; ++k) {
                const uint32_t byte = (wd[k >> 2] >> (24 - 8 * (k & 3))) & 0xffu;
                out[k] = (uint8_t)byte;
                reg ^= byte;
                #pragma unroll
                for (int b = 0; b < 8; ++b) reg = (reg & 0x80u) ? ((reg << 1) ^ 0x07u) & 0xffu : (reg << 1) & 0xffu;
            }
            a.cand_metric[f * a.lsz + rank] = metric;
            a.cand_ok[f * a.lsz + rank] = (uint8_t)(reg == (wd[MWIN_W - 1] & 0xffu));
            }
        }
        if (pl == 0 && f_store) a.ncand[f] = cnt;
        group_sync();
    }
    // ---- release the slot: every wave's slab stores are done (and performed) before the bit clears.  The fence is agent-scope: on a
    // chip whose XCDs have their own L2 it also writes this XCD's dirty slab lines back, so a late eviction here cannot overwrite
    // what the slot's next owner -- possibly on another XCD -- has written since (an owner only ever reads what it wrote itself).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // (release only: an acquire would also invalidate this XCD's L2 under the blocks still running)
    __syncthreads();
    if (p == 0) atomicAnd(&a.slot_bits[slot >> 5], ~(1u << (slot & 31)));
}


template <int L, int LF, bool GK = false>
int launch_wide(es_ctx* ctx, WideArgs a, int64_t B, hipStream_t st)
{
    const size_t lds = sizeof(WideLds<L, (LF > 64 ? 2 : 1)>);
    static_assert((sizeof(WideLds<256, 2>) + 1279) / 1280 * 1280 * 3 <= 160 * 1024, "three workgroups per CU at L = 256 (LDS is handed out in 1 280-byte granules)");
    static_assert((sizeof(WideLds<128, 2>) + 1279) / 1280 * 1280 * 6 <= 160 * 1024, "six two-wave workgroups per CU at L = 128");
#if ES_WIDE_GATHER_BPERM && ES_WIDE_LDS_DEPTH >= 7
    static_assert((sizeof(WideLds<64, 1>) + 1279) / 1280 * 1280 * 4 * ES_WIDE_WPS <= 160 * 1024, "4 x ES_WIDE_WPS one-wave workgroups per CU");
#endif
    static_assert((LF & (LF - 1)) == 0, "power of two");
    constexpr unsigned attr_bit = (unsigned)LF << (GK ? 9 : 0);        // one instantiation per list capacity (and per kind of code)
    if (!(ctx->wide_attr_mask & attr_bit)) {                           // per context (= per device): the attribute belongs to the device's copy of the kernel
        ES_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&es_scl_wide_kernel<L, LF, GK>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->wide_attr_mask |= attr_bit;
    }
    // One slab slot per resident workgroup.  The slab is sized in units of the context's largest block (Lm lanes); a smaller block
    // gets Lm / L times as many slots out of the same bytes.
    const int Lm = es_wide_lanes_max(ctx);
    constexpr int FRB = L / LF;                                     // frames per block
    const long long blocks = (B + FRB - 1) / FRB;                    // one frame group per wave / block; the hardware keeps <= 3 blocks (L = 128: 5) per CU resident
    if (blocks >= (1LL << 31)) { ctx->err = "es_scl_batch: batch too large for one launch"; return ES_EINVAL; }
    a.slot_bits = ctx->d_wide_slot_bits;
    a.n_slots = ctx->wide_slots * (Lm / L);
    a.slot_words = (a.n_slots + 31) / 32;
    { const int rc = es_slab_enter(ctx, 1, 0x300 | L, true, st); if (rc) return rc; }       // slot stride depends on the block's lanes only
    hipLaunchKernelGGL((es_scl_wide_kernel<L, LF, GK>), dim3((unsigned)blocks), dim3(L), lds, st, a);
    ES_HIP_CHECK(ctx, hipGetLastError());
    { const int rc = es_slab_leave(ctx, 1, 0x300 | L, true, st); if (rc) return rc; }
    return ES_OK;
}

}  // namespace

// Scratch for this file's kernels: per resident workgroup of Lm lanes 1024*Lm doubles + WIDE_AUX_PER_PATH*Lm bytes.
// Lists of up to 64 paths run in blocks of 256 lanes, so every context that has the scratch has it for 256-lane blocks.
size_t es_scl_wide_scratch_bytes(const es_ctx* ctx, int* slots_out)
{
    if (!ctx->wide_enabled) { *slots_out = 0; return 0; }
    const size_t Lm = (size_t)es_wide_lanes_max(ctx);
    const int slots = ctx->num_cu * ES_WIDE_WPS * (int)(256 / Lm);          // 4 x ES_WIDE_WPS waves per CU
    *slots_out = slots;
    return (size_t)slots * (N * Lm * sizeof(double) + (size_t)WIDE_AUX_PER_PATH * Lm);
}

int es_launch_scl_wide(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
                       uint8_t* hard_info, uint8_t* hard_ok, uint8_t* cand_info, double* cand_metric,
                       uint8_t* cand_ok, int32_t* ncand, hipStream_t st)
{
    if (!ctx->d_wide_scratch) { ctx->err = "es_scl_batch: this context has no scratch for the lane-per-path list decoder (list_size_max <= 32 and scl_lanes 1 never requested before es_reserve)"; return ES_EINVAL; }
    WideArgs a{};
    a.llr = llr; a.is_f64 = (dtype == ES_DTYPE_F64); a.B = B;
    a.frozen = ctx->frozen; a.data_pos = ctx->d_data_pos; a.exp_tab = ctx->d_exp_tab;
    const size_t Lm = (size_t)es_wide_lanes_max(ctx);
    a.alpha = reinterpret_cast<double*>(ctx->d_wide_scratch);
    a.aux = reinterpret_cast<unsigned char*>(ctx->d_wide_scratch) + (size_t)ctx->wide_slots * N * Lm * sizeof(double);
    a.hard_info = hard_info; a.hard_ok = hard_ok; a.cand_info = cand_info;
    a.cand_metric = cand_metric; a.cand_ok = cand_ok; a.ncand = ncand;
    a.skip_if_hard_ok = skip_if_hard_ok;
    a.lsz = L;
    a.prio = ctx->scl_prio;
    a.n_info = ctx->n_info; a.info_bytes = (ctx->n_info - 8 + 7) / 8;
    if (skip_if_hard_ok && ES_WIDE_COMPACT) {             // frames drawn from a counter: see the kernel (settled frames never ride along as idle lanes)
        if (B >= (1LL << 31) - (1LL << 24)) { ctx->err = "es_scl_batch: batch too large for one launch"; return ES_EINVAL; }   // (the counter runs past B by one draw per block)
        const int rc = es_cursor_next(ctx, st, &a.cursor);
        if (rc) return rc;
        ES_HIP_CHECK(ctx, hipMemsetAsync(a.cursor, 0, sizeof(int), st));
    }
    int LP = 1; while (LP < L) LP <<= 1;                  // kernel capacity: the next power of two
    if (ctx->n_info != KINFO) switch (LP) {               // a code other than the reference's own K = 448: the run-time-K instantiations
        case 1:   return launch_wide<64, 1, true>(ctx, a, B, st);
        case 2:   return launch_wide<64, 2, true>(ctx, a, B, st);
        case 4:   return launch_wide<64, 4, true>(ctx, a, B, st);
        case 8:   return launch_wide<64, 8, true>(ctx, a, B, st);
        case 16:  return launch_wide<64, 16, true>(ctx, a, B, st);
        case 32:  return launch_wide<64, 32, true>(ctx, a, B, st);
        case 64:  return launch_wide<64, 64, true>(ctx, a, B, st);
        case 128: return launch_wide<128, 128, true>(ctx, a, B, st);
        case 256: return launch_wide<256, 256, true>(ctx, a, B, st);
        default: ctx->err = "list_size must be in 1..256"; return ES_EINVAL;
    }
    switch (LP) {
        case 1:   return launch_wide<64, 1>(ctx, a, B, st);
        case 2:   return launch_wide<64, 2>(ctx, a, B, st);
        case 4:   return launch_wide<64, 4>(ctx, a, B, st);
        case 8:   return launch_wide<64, 8>(ctx, a, B, st);
        case 16:  return launch_wide<64, 16>(ctx, a, B, st);
        case 32:  return launch_wide<64, 32>(ctx, a, B, st);
        case 64:  return launch_wide<64, 64>(ctx, a, B, st);
        case 128: return launch_wide<128, 128>(ctx, a, B, st);
        case 256: return launch_wide<256, 256>(ctx, a, B, st);
        default: ctx->err = "list_size must be in 1..256"; return ES_EINVAL;
    }
}
