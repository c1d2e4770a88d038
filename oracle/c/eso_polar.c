/* eso_polar.c -- CPU oracle for the Polar(1024,448)+CRC-8 encode / SCL decode of the reference.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in echoseal_amd/ (the product) may link, import or call
 * this file; it is the checker that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg compare the HIP kernels against.
 *
 * It is a plain-C restatement of rtwm/fastpolar.py (PolarCode, _ListPath) written with full
 * per-path copies instead of the reference's copy-on-write arrays, and with the standard
 * O(N log N) successive-cancellation schedule instead of the reference's "invalidate every
 * ancestor, recompute root->leaf" walk (rtwm/fastpolar.py:127-154,185-190).  Both produce the
 * same float64 values because each recomputation applies the same f/g to the same operands;
 * tests/test_oracle_polar.py pins that claim against vectors produced by the reference itself
 * (tests/golden/polar_*.npz, generator: oracle/refshim/gen_golden.py).
 *
 * Float64 transcendentals come from echoseal_amd/csrc/es_math.h (bit-exact with glibc).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "es_math.h"
#include "es_polar_q.h"

#define NN 1024
#define NLEV 10
#define KMAX 1024
#define CRC 8
/* K (information + CRC bits) is 448 in everything the reference instantiates (rtwm/polar_fast.py:18-24); PolarCode itself takes any
 * 0 < K <= N (rtwm/fastpolar.py:209-234) and eso_polar_set_k follows it there (the tests of other K). */
static int g_K = 448;
#define KK g_K
#define INFO (g_K - CRC)

static const uint64_t EXPTAB[ES_EXP_TAB_WORDS] = ES_EXP_TAB_INIT;

static int g_tables_ready = 0;
static uint8_t g_frozen[NN];
static int32_t g_data_pos[KMAX];

/* rtwm/fastpolar.py:220-230 -- frozen = all True; frozen[rel[:K]] = False (the K LEAST reliable
 * indices become information positions); _data_pos = flatnonzero(~frozen) (ascending index). */
static void build_tables(void)
{
    if (g_tables_ready) return;
    memset(g_frozen, 1, NN);
    for (int i = 0; i < KK; i++) g_frozen[ES_POLAR_Q1024[i]] = 0;
    int n = 0;
    for (int i = 0; i < NN; i++) if (!g_frozen[i]) g_data_pos[n++] = i;
    g_tables_ready = 1;
}

void eso_polar_tables(uint8_t* frozen, int32_t* data_pos)
{
    build_tables();
    memcpy(frozen, g_frozen, NN);
    memcpy(data_pos, g_data_pos, sizeof(int32_t) * (size_t)g_K);
}

/* K of the code the following calls use (default 448); returns the previous value */
int eso_polar_set_k(int K)
{
    const int old = g_K;
    if (K > CRC && K <= KMAX && K != g_K) { g_K = K; g_tables_ready = 0; build_tables(); }
    return old;
}

/* rtwm/fastpolar.py:362-371 -- CRC-8, poly 0x07, init 0, MSB first, no reflection. */
uint8_t eso_crc8(const uint8_t* bits, int n)
{
    uint8_t reg = 0;
    for (int i = 0; i < n; i++) {
        reg ^= (uint8_t)((bits[i] & 1) << 7);
        reg = (reg & 0x80) ? (uint8_t)((reg << 1) ^ 0x07) : (uint8_t)(reg << 1);
    }
    return reg;
}

/* rtwm/fastpolar.py:376-389 -- in-place butterfly, stage s has blocks of 2^(s+1): the first
 * half of each block is XORed with the second half. */
void eso_polar_transform(uint8_t* x)
{
    for (int half = 1; half < NN; half <<= 1)
        for (int i = 0; i < NN; i += 2 * half)
            for (int j = 0; j < half; j++) x[i + j] ^= x[i + half + j];
}

/* rtwm/fastpolar.py:237-252 */
void eso_polar_encode(const uint8_t* info440, uint8_t* code1024)
{
    build_tables();
    uint8_t crc = eso_crc8(info440, INFO);
    memset(code1024, 0, NN);
    for (int i = 0; i < INFO; i++) code1024[g_data_pos[i]] = info440[i] & 1;
    for (int i = 0; i < CRC; i++) code1024[g_data_pos[INFO + i]] = (crc >> (7 - i)) & 1;
    eso_polar_transform(code1024);
}

static int crc_ok(const uint8_t* data448)
{
    uint8_t crc = eso_crc8(data448, INFO);
    for (int i = 0; i < CRC; i++)
        if (((crc >> (7 - i)) & 1) != data448[INFO + i]) return 0;
    return 1;
}

/* rtwm/fastpolar.py:260-268 -- hard decision -> transform -> zero frozen -> CRC.
 * Returns crc flag, writes 440 info bits. */
int eso_polar_hard(const double* llr, uint8_t* info440)
{
    build_tables();
    uint8_t x[NN], data[KMAX];
    for (int i = 0; i < NN; i++) x[i] = llr[i] > 0.0;
    eso_polar_transform(x);
    for (int i = 0; i < KK; i++) data[i] = x[g_data_pos[i]];
    memcpy(info440, data, INFO);
    return crc_ok(data);
}

/* ---- one list path (rtwm/fastpolar.py:59-190, flattened) ---------------------------------- */
typedef struct {
    double  metric;
    uint8_t u[NN];
    /* LLRs of the node currently open at depth l (1..10) live at alpha[(1024>>l) ..]; depth 0
       (the channel LLRs) is shared by all paths and kept outside the struct. */
    double  alpha[NN];
    /* beta[l]: partial sums at depth l, absolute positions as in the reference. */
    uint8_t beta[NLEV + 1][NN];
} path_t;

/* LLR of leaf `i`: recompute the depths that changed since leaf i-1 (fastpolar.py:127-154). */
static double path_calc_llr(path_t* p, const double* chan, int i)
{
    int top;                                   /* shallowest depth to recompute */
    if (i == 0) top = 1;
    else top = NLEV - __builtin_ctz((unsigned)i);
    for (int lev = top; lev <= NLEV; lev++) {
        const int size = NN >> lev;            /* node width at this depth */
        const int node = i >> (NLEV - lev);
        const double* par = (lev == 1) ? chan : &p->alpha[2 * size]; /* parent block, width 2*size */
        double* dst = &p->alpha[size];
        if ((node & 1) == 0) {
            for (int j = 0; j < size; j++) dst[j] = es_polar_f(par[j], par[j + size], EXPTAB);
        } else {
            const uint8_t* bl = &p->beta[lev][(node - 1) * size];
            for (int j = 0; j < size; j++) dst[j] = es_polar_g(par[j], par[j + size], bl[j]);
        }
    }
    return p->alpha[1];
}

/* rtwm/fastpolar.py:156-183 -- write the decision, fold partial sums upward on odd nodes. */
static void path_extend(path_t* p, int i, int bit)
{
    p->u[i] = (uint8_t)bit;
    int lev = NLEV, node = i;
    p->beta[lev][i] = (uint8_t)bit;
    while ((node & 1) && lev > 0) {
        const int size = NN >> lev;
        const uint8_t* left = &p->beta[lev][(node - 1) * size];
        const uint8_t* right = &p->beta[lev][node * size];
        uint8_t* par = &p->beta[lev - 1][(node >> 1) * 2 * size];
        for (int j = 0; j < size; j++) { par[j] = left[j] ^ right[j]; par[size + j] = right[j]; }
        node >>= 1;
        lev--;
    }
}

typedef struct { double metric; int idx; int bit; } cand_t;

/* Python's list.sort(key=metric) is stable: equal metrics keep their original order. */
static void stable_sort_cands(cand_t* c, int n)
{
    for (int i = 1; i < n; i++) {
        cand_t t = c[i];
        int j = i - 1;
        while (j >= 0 && t.metric < c[j].metric) { c[j + 1] = c[j]; j--; }
        c[j + 1] = t;
    }
}

/* Full list decode (rtwm/fastpolar.py:278-330) plus the ordering of the final scan (:335).
 * Outputs, for each surviving path in ascending-metric (stable) order:
 *   cand_info[r][440], cand_metric[r], cand_crc[r].  Returns the number of paths. */
int eso_scl_list(const double* llr, int L, uint8_t* cand_info, double* cand_metric, uint8_t* cand_crc)
{
    build_tables();
    path_t** paths = (path_t**)calloc((size_t)L, sizeof(path_t*));
    path_t** next = (path_t**)calloc((size_t)L, sizeof(path_t*));
    cand_t* cands = (cand_t*)malloc(sizeof(cand_t) * 2 * (size_t)L);
    int* uses = (int*)malloc(sizeof(int) * (size_t)L);
    int npaths = 1;
    paths[0] = (path_t*)calloc(1, sizeof(path_t));

    for (int i = 0; i < NN; i++) {
        if (g_frozen[i]) {                                     /* :281-286 */
            for (int k = 0; k < npaths; k++) {
                double lam = path_calc_llr(paths[k], llr, i);
                paths[k]->metric += es_metric_penalty(lam, 0, EXPTAB);
                path_extend(paths[k], i, 0);
            }
            continue;
        }
        int nc = 0;                                            /* :288-293 */
        for (int k = 0; k < npaths; k++) {
            double lam = path_calc_llr(paths[k], llr, i);
            double base = paths[k]->metric;
            cands[nc].metric = base + es_metric_penalty(lam, 0, EXPTAB); cands[nc].idx = k; cands[nc].bit = 0; nc++;
            cands[nc].metric = base + es_metric_penalty(lam, 1, EXPTAB); cands[nc].idx = k; cands[nc].bit = 1; nc++;
        }
        stable_sort_cands(cands, nc);                          /* :298-299 */
        int keep = nc < L ? nc : L;
        memset(uses, 0, sizeof(int) * (size_t)L);
        for (int r = 0; r < keep; r++) {                       /* :301-324 */
            path_t* src = paths[cands[r].idx];
            path_t* dst;
            if (uses[cands[r].idx]++ == 0) dst = src;          /* first survivor keeps the object */
            else { dst = (path_t*)malloc(sizeof(path_t)); memcpy(dst, src, sizeof(path_t)); }
            next[r] = dst;
        }
        /* clones were taken before any extension in the reference (:307-312); copying the
           source before it is extended is equivalent, so extend only after all copies exist. */
        for (int r = 0; r < keep; r++) {
            next[r]->metric = cands[r].metric;
            path_extend(next[r], i, cands[r].bit);
        }
        for (int k = 0; k < npaths; k++) if (uses[k] == 0) free(paths[k]);   /* :326-328 */
        path_t** t = paths; paths = next; next = t;
        npaths = keep;
    }

    /* :335  sorted(paths, key=metric), stable */
    cand_t* order = cands;
    for (int k = 0; k < npaths; k++) { order[k].metric = paths[k]->metric; order[k].idx = k; order[k].bit = 0; }
    stable_sort_cands(order, npaths);
    for (int r = 0; r < npaths; r++) {
        const path_t* p = paths[order[r].idx];
        uint8_t data[KMAX];
        for (int j = 0; j < KK; j++) data[j] = p->u[g_data_pos[j]];
        memcpy(cand_info + (size_t)r * INFO, data, INFO);
        cand_metric[r] = p->metric;
        cand_crc[r] = (uint8_t)crc_ok(data);
    }
    for (int k = 0; k < npaths; k++) free(paths[k]);
    free(paths); free(next); free(cands); free(uses);
    return npaths;
}

/* PolarCode.decode(llr, validator=None) -> (info bits, ok)   (rtwm/fastpolar.py:254-359).
 * `took_list` (nullable) reports whether the list loop ran. */
int eso_polar_decode(const double* llr, int L, uint8_t* info440, int* took_list)
{
    uint8_t hard[KMAX];
    if (took_list) *took_list = 0;
    if (eso_polar_hard(llr, hard)) { memcpy(info440, hard, INFO); return 1; }   /* :268-276 */
    if (took_list) *took_list = 1;
    uint8_t* ci = (uint8_t*)malloc((size_t)L * INFO);
    double* cm = (double*)malloc(sizeof(double) * (size_t)L);
    uint8_t* cc = (uint8_t*)malloc((size_t)L);
    int n = eso_scl_list(llr, L, ci, cm, cc);
    int ok = 0;
    /* :332-359 with validator None: the first CRC-passing path in metric order wins; otherwise the
       lowest-metric path (strict <, so the first of equals), falling back to the hard decision
       only if every metric is +inf/nan. */
    int pick = -1;
    for (int r = 0; r < n && pick < 0; r++) if (cc[r]) { pick = r; ok = 1; }
    if (pick < 0) {
        double best = es_u2d(0x7ff0000000000000ULL);
        for (int r = 0; r < n; r++) if (cm[r] < best) { best = cm[r]; pick = r; }
    }
    if (pick >= 0) memcpy(info440, ci + (size_t)pick * INFO, INFO);
    else memcpy(info440, hard, INFO);
    free(ci); free(cm); free(cc);
    return ok;
}

/* vector helpers used by tests/test_oracle_math.py */
void eso_exp_vec(const double* x, double* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = es_exp(x[i], EXPTAB); }
void eso_log1p_vec(const double* x, double* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = es_log1p(x[i]); }
void eso_logaddexp_vec(const double* a, const double* b, double* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = es_logaddexp(a[i], b[i], EXPTAB); }
void eso_polar_f_vec(const double* a, const double* b, double* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = es_polar_f(a[i], b[i], EXPTAB); }
void eso_penalty_vec(const double* l, int bit, double* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = es_metric_penalty(l[i], (uint32_t)bit, EXPTAB); }
