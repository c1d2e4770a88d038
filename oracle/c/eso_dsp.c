/* eso_dsp.c -- CPU oracle for the sync and soft-demodulation stages of the reference detector.
 *
 * TEST INFRASTRUCTURE ONLY (see eso_polar.c).  Plain-C restatement of
 *   rtwm/detector.py:59-60     band-pass (scipy.signal.lfilter, direct form II transposed)
 *   rtwm/detector.py:76-79     normalised cross-correlation with the 63-chip template
 *   rtwm/detector.py:83-86     median / MAD threshold
 *   rtwm/detector.py:87-99     non-maximum suppression + top-5 fallback
 *   rtwm/detector.py:296-416   _llr: matched filter, chip-shift search, despread, robust scaling
 *
 * Arithmetic notes (what is bit-defined and what is not):
 *  - lfilter: same operation order as SciPy's C loop, no FMA -> bit-identical to scipy (tested).
 *  - NCC numerator / window energy: the reference calls BLAS ddot through np.correlate /
 *    np.convolve, whose summation order depends on the BLAS kernel of the machine.  Here the
 *    order is FIXED: ascending tap index, fused multiply-add for the numerator, plain adds for the
 *    energy.  Results agree with the reference to ~1e-15 relative, decisions (peaks) identically.
 *  - matched filter: the reference's np.convolve on float32 is BLAS sdot (machine dependent).
 *    Here each output is accumulated in float64 (products of float32 are exact in float64) in
 *    ascending sample order and rounded once to float32.
 *  - everything after the matched filter follows NumPy's float32 semantics exactly: pairwise
 *    summation (blocks of 128, 8 accumulators), float32 mean/std, medians by selection.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRE_L 63
#define HDR_L 128
#define NPAY 1024
#define PAYLOAD_START (PRE_L + HDR_L)

/* ---- a2: scipy.signal.lfilter(b, a, x) for float32 x, float64 coefficients ---------------- */
/* zi: nullable in/out state of nb-1 doubles (zero state when NULL). */
void eso_lfilter(const double* b_in, const double* a_in, int nb, const float* x, double* y,
                 int64_t n, double* zi)
{
    double b[16], a[16], z[16];
    const double a0 = a_in[0];
    for (int k = 0; k < nb; k++) { b[k] = b_in[k] / a0; a[k] = a_in[k] / a0; }
    for (int k = 0; k < nb - 1; k++) z[k] = zi ? zi[k] : 0.0;
    for (int64_t t = 0; t < n; t++) {
        const double xn = (double)x[t];
        double yn;
        if (nb > 1) {
            yn = z[0] + b[0] * xn;
            for (int k = 0; k < nb - 2; k++) z[k] = (z[k + 1] + xn * b[k + 1]) - yn * a[k + 1];
            z[nb - 2] = xn * b[nb - 1] - yn * a[nb - 1];
        } else {
            yn = xn * b[0];
        }
        y[t] = yn;
    }
    if (zi) for (int k = 0; k < nb - 1; k++) zi[k] = z[k];
}

/* same filter on float64 input (used for the template and the matched-filter design) */
void eso_lfilter_f64(const double* b_in, const double* a_in, int nb, const double* x, double* y,
                     int64_t n)
{
    double b[16], a[16], z[16];
    const double a0 = a_in[0];
    for (int k = 0; k < nb; k++) { b[k] = b_in[k] / a0; a[k] = a_in[k] / a0; }
    for (int k = 0; k < nb - 1; k++) z[k] = 0.0;
    for (int64_t t = 0; t < n; t++) {
        const double xn = x[t];
        double yn = z[0] + b[0] * xn;
        for (int k = 0; k < nb - 2; k++) z[k] = (z[k + 1] + xn * b[k + 1]) - yn * a[k + 1];
        z[nb - 2] = xn * b[nb - 1] - yn * a[nb - 1];
        y[t] = yn;
    }
}

/* ---- a4: corr[i] = sum_k y[i+k] tpl[k] / (sqrt(sum_k y[i+k]^2) + 1e-12) -------------------
 * Numerator: FMA chain over ascending tap index.
 * Window energy: all-positive partial sums shared by chunks of XC_CHUNK consecutive lags (the
 * reference's own order is whatever BLAS ddot does; any fixed order of 63 non-negative terms is
 * accurate to a few ulp).  For lag i in the chunk starting at c = i - i % XC_CHUNK:
 *     core = y2[c+18] + ... + y2[c+62]            (ascending)
 *     head = y2[c+17] + y2[c+16] + ... + y2[i]    (descending accumulation; empty when i = c+18)
 *     tail = y2[c+63] + ... + y2[i+62]            (ascending; empty when i = c)
 *     E[i] = (head + core) + tail
 * which is exactly what one GPU lane (one chunk) accumulates in registers. */
#define XC_CHUNK 19
void eso_ncc(const double* y, int64_t n, const double* tpl, int L, double* corr)
{
    const int64_t n_lags = n - L + 1;
    for (int64_t c = 0; c < n_lags; c += XC_CHUNK) {
        double core = 0.0;
        for (int j = XC_CHUNK - 1; j < L; j++) core = core + y[c + j] * y[c + j];
        double head[XC_CHUNK], tail[XC_CHUNK];
        head[XC_CHUNK - 1] = 0.0;
        for (int r = XC_CHUNK - 2; r >= 0; r--) head[r] = head[r + 1] + y[c + r] * y[c + r];
        tail[0] = 0.0;
        for (int r = 1; r < XC_CHUNK; r++) {
            const int64_t j = c + L - 1 + r;
            tail[r] = (j < n) ? tail[r - 1] + y[j] * y[j] : tail[r - 1];
        }
        for (int r = 0; r < XC_CHUNK && c + r < n_lags; r++) {
            const int64_t i = c + r;
            double num = 0.0;
            for (int k = 0; k < L; k++) num = __builtin_fma(y[i + k], tpl[k], num);
            const double en = (head[r] + core) + tail[r];
            corr[i] = num / (sqrt(en) + 1e-12);
        }
    }
}

static int cmp_f64(const void* p, const void* q)
{
    const double a = *(const double*)p, b = *(const double*)q;
    return (a > b) - (a < b);
}

/* np.median on float64: middle element, or the mean of the two middle elements. */
static double median_f64(const double* v, int64_t n, double* scratch)
{
    memcpy(scratch, v, sizeof(double) * (size_t)n);
    qsort(scratch, (size_t)n, sizeof(double), cmp_f64);
    if (n & 1) return scratch[n / 2];
    return (scratch[n / 2 - 1] + scratch[n / 2]) / 2.0;
}

/* ---- a5: thr = min(med + 4.5*1.4826*mad, 0.95) -------------------------------------------- */
double eso_cfar_threshold(const double* corr, int64_t n, double* med_out, double* mad_out)
{
    double* s = (double*)malloc(sizeof(double) * (size_t)n * 2);
    double* dev = s + n;
    const double med = median_f64(corr, n, s);
    for (int64_t i = 0; i < n; i++) dev[i] = fabs(corr[i] - med);
    double* s2 = (double*)malloc(sizeof(double) * (size_t)n);
    const double mad = median_f64(dev, n, s2) + 1e-12;
    free(s2); free(s);
    double thr = med + 4.5 * 1.4826 * mad;
    if (0.95 < thr) thr = 0.95;
    if (med_out) *med_out = med;
    if (mad_out) *mad_out = mad;
    return thr;
}

/* ---- a6: peaks.  i is a peak iff corr[i] >= thr and corr[i] >= max(corr[i-607 : i+608]).
 * Returns the number of peaks written (ascending index, at most max_peaks; the total count is
 * stored in *total).  When there is none, the reference falls back to the 5 largest
 * correlations in descending order; ties there are resolved by NumPy's unstable argsort and are
 * declared ambiguous -- this restatement prefers the higher index (what a stable ascending sort
 * followed by [-5:][::-1] gives).  *fallback tells which branch ran. */
int eso_pick_peaks(const double* corr, int64_t n, double thr, int min_distance, int32_t* peaks,
                   int max_peaks, int* total, int* fallback)
{
    int cnt = 0, tot = 0;
    for (int64_t i = 0; i < n; i++) {
        if (corr[i] < thr) continue;
        int64_t lo = i - min_distance; if (lo < 0) lo = 0;
        int64_t hi = i + min_distance + 1; if (hi > n) hi = n;
        double m = corr[lo];
        for (int64_t j = lo + 1; j < hi; j++) if (corr[j] > m) m = corr[j];
        if (corr[i] >= m) { if (cnt < max_peaks) peaks[cnt++] = (int32_t)i; tot++; }
    }
    *fallback = 0;
    if (tot == 0) {
        *fallback = 1;
        int k = n < 5 ? (int)n : 5;
        int64_t taken[5];
        for (int r = 0; r < k; r++) {
            int64_t best = -1;
            for (int64_t i = 0; i < n; i++) {
                int used = 0;
                for (int q = 0; q < r; q++) used |= (taken[q] == i);
                if (used) continue;
                if (best < 0 || corr[i] >= corr[best]) best = i;
            }
            taken[r] = best;
            if (r < max_peaks) peaks[r] = (int32_t)best;
        }
        cnt = k < max_peaks ? k : max_peaks; tot = k;
    }
    if (total) *total = tot;
    return cnt;
}

/* ---- NumPy float32 reductions --------------------------------------------------------------- */
/* np.add.reduce on a contiguous float32 vector (pairwise: <8 plain, <=128 eight lanes, else split). */
static float pairwise_f32(const float* a, int64_t n)
{
    if (n < 8) {
        float r = 0.0f;
        for (int64_t i = 0; i < n; i++) r = r + a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_f32(a, n2) + pairwise_f32(a + n2, n - n2);
}

float eso_sum_f32(const float* a, int64_t n) { return pairwise_f32(a, n); }

static float mean_f32(const float* a, int64_t n) { return pairwise_f32(a, n) / (float)n; }

static int cmp_f32(const void* p, const void* q)
{
    const float a = *(const float*)p, b = *(const float*)q;
    return (a > b) - (a < b);
}

static float median_f32(const float* v, int64_t n, float* scratch)
{
    memcpy(scratch, v, sizeof(float) * (size_t)n);
    qsort(scratch, (size_t)n, sizeof(float), cmp_f32);
    if (n & 1) return scratch[n / 2];
    return (scratch[n / 2 - 1] + scratch[n / 2]) / 2.0f;
}

/* ---- a8: _llr --------------------------------------------------------------------------------
 * frame      : band-passed samples (float64, as sliced from y), flen of them (1215 normally)
 * pn_bits    : 1024 PN bits {0,1} for the payload (variant already applied by the caller)
 * h, ntaps   : matched-filter taps (float32)
 * llr        : out, 1024 float32
 * diag       : nullable out [4]: best_s, best_score, runner-up score, n                         */
void eso_llr(const double* frame, int flen, const uint8_t* pn_bits, const float* h, int ntaps,
             float* llr, double* diag)
{
    memset(llr, 0, sizeof(float) * NPAY);
    if (diag) { diag[0] = 0; diag[1] = -1; diag[2] = -1; diag[3] = 0; }
    const int mem = ntaps - 1;
    if (PAYLOAD_START >= flen) return;                                  /* :320-321 */
    const int npl = flen - PAYLOAD_START;                               /* payload samples */
    const int prefix = mem < PAYLOAD_START ? mem : PAYLOAD_START;       /* :327 */
    const int nfull = prefix + npl;
    float* rx = (float*)malloc(sizeof(float) * (size_t)nfull);
    for (int i = 0; i < nfull; i++) rx[i] = (float)frame[PAYLOAD_START - prefix + i];

    const int nmf = nfull + ntaps - 1;                                  /* np.convolve 'full' */
    float* mf = (float*)malloc(sizeof(float) * (size_t)nmf);
    for (int j = 0; j < nmf; j++) {
        int i0 = j - (ntaps - 1); if (i0 < 0) i0 = 0;
        int i1 = j < nfull - 1 ? j : nfull - 1;
        double acc = 0.0;
        for (int i = i0; i <= i1; i++) acc += (double)rx[i] * (double)h[j - i];
        mf[j] = (float)acc;
    }
    const int offset = prefix + mem;                                    /* :335 */
    const int n = NPAY < npl ? NPAY : npl;                              /* :337 */
    if (n <= 0) { free(rx); free(mf); return; }

    int raw_shift = n / 2;                                              /* :351-353 */
    if (4 * ntaps < raw_shift) raw_shift = 4 * ntaps;
    if (HDR_L < raw_shift) raw_shift = HDR_L;
    const int max_shift = mem > raw_shift ? mem : raw_shift;
    const int start = offset - max_shift > 0 ? offset - max_shift : 0;
    const int stop = nmf < offset + n + max_shift ? nmf : offset + n + max_shift;
    const float* win = mf + start;
    const int nwin = stop - start;
    const int base = offset - start;

    int guard = ntaps / 2 > 24 ? ntaps / 2 : 24;                        /* :361-363 */
    if (n / 4 < guard) guard = n / 4;
    if (guard >= n) guard = n / 4 > 0 ? n / 4 : 0;

    float* pn = (float*)malloc(sizeof(float) * (size_t)n);
    float* d = (float*)malloc(sizeof(float) * (size_t)n * 3);
    float* scratch = d + n;
    float* dev = d + 2 * n;
    for (int i = 0; i < n; i++) pn[i] = 2.0f * (float)pn_bits[i] - 1.0f;

    int best_s = 0;                                                     /* :366-379 */
    double best = -1.0, second = -1.0;
    for (int s = -max_shift; s <= max_shift; s++) {
        const int i0 = base + s, i1 = i0 + n;
        if (i0 < 0 || i1 > nwin) continue;
        for (int i = guard; i < n; i++) d[i] = fabsf(win[i0 + i] * pn[i]);
        const double score = (double)mean_f32(d + guard, n - guard);
        if (score > best) { second = best; best = score; best_s = s; }
        else if (score > second) second = score;
    }
    const int i0 = base + best_s;                                       /* :382-385 */
    for (int i = 0; i < n; i++) d[i] = win[i0 + i] * pn[i];

    const float* tail = (n > guard + 8) ? d + guard : d;                /* :395 */
    const int nt = (n > guard + 8) ? n - guard : n;
    const float mu32 = mean_f32(tail, nt);                              /* :396 */
    const float medv = median_f32(tail, nt, scratch);                   /* :399 */
    for (int i = 0; i < nt; i++) dev[i] = fabsf(tail[i] - medv);
    const double mad = (double)median_f32(dev, nt, scratch) + 1e-12;
    const double sigma_mad = 1.4826 * mad;                              /* :400 */
    for (int i = 0; i < nt; i++) { const float c = tail[i] - mu32; dev[i] = c * c; }   /* np.std */
    const float var32 = pairwise_f32(dev, nt) / (float)nt;
    const double sigma_std = (double)sqrtf(var32) + 1e-12;              /* :401 */
    double sigma = sigma_mad > sigma_std ? sigma_mad : sigma_std;       /* :402 */
    if (0.1 > sigma) sigma = 0.1;
    double scale = 2.0 / (sigma * sigma);                               /* :404 */
    if (scale < 0.5) scale = 0.5;
    if (scale > 30.0) scale = 30.0;
    const float scale32 = (float)scale;      /* NumPy weak-scalar rule: f32 array * python float */
    for (int i = 0; i < n; i++) {                                       /* :397,405 */
        float v = (d[i] - mu32) * scale32;
        if (v < -12.0f) v = -12.0f;
        if (v > 12.0f) v = 12.0f;
        llr[i] = v;
    }
    if (diag) { diag[0] = best_s; diag[1] = best; diag[2] = second; diag[3] = n; }
    free(pn); free(d); free(mf); free(rx);
}

/* ---- header decode: WatermarkDetector._decode_header (rtwm/detector.py:452-515) ---------------
 * frame   : band-passed samples (float64), flen >= 191 of them used
 * hdr_pn  : 128 PN bits {0,1} of counter 0 (static header PN)
 * h,ntaps : matched-filter taps
 * out[3]  : ok (0/1), 16-bit value, score.   diag (nullable): best_s, |corr| of the best shift.
 * NumPy float32 semantics throughout (pairwise sums; the (16,8) row sums are 8-element pairwise
 * leaves; python-float constants are weak and cast to float32). */
void eso_decode_header(const double* frame, int flen, const uint8_t* hdr_pn, const float* h, int ntaps,
                       double* out, double* diag)
{
    out[0] = 0; out[1] = 0; out[2] = 0.0;
    if (diag) { diag[0] = 0; diag[1] = -1; }
    if (flen < PRE_L + HDR_L) return;                                   /* :461-462 */
    const int mem = ntaps - 1;
    const int prefix = mem < PRE_L ? mem : PRE_L;                       /* :466 */
    const int nfull = prefix + HDR_L;
    float rx[PRE_L + HDR_L];
    for (int i = 0; i < nfull; i++) rx[i] = (float)frame[PRE_L - prefix + i];
    const int nmf = nfull + ntaps - 1;
    float* mf = (float*)malloc(sizeof(float) * (size_t)nmf);
    for (int j = 0; j < nmf; j++) {                                     /* :473, same MF definition as _llr */
        int i0 = j - (ntaps - 1); if (i0 < 0) i0 = 0;
        int i1 = j < nfull - 1 ? j : nfull - 1;
        double acc = 0.0;
        for (int i = i0; i <= i1; i++) acc += (double)rx[i] * (double)h[j - i];
        mf[j] = (float)acc;
    }
    const int offset = mem + prefix;                                    /* :474 */
    int max_shift = HDR_L / 2 + prefix;                                 /* :475-478 */
    if (4 * ntaps < max_shift) max_shift = 4 * ntaps;
    if (max_shift < mem) max_shift = mem;
    const int start = offset - max_shift > 0 ? offset - max_shift : 0;
    const int stop = nmf < offset + HDR_L + max_shift ? nmf : offset + HDR_L + max_shift;
    const float* win = mf + start;
    const int nwin = stop - start;
    const int base = offset - start;
    int guard = ntaps / 8 < 32 ? ntaps / 8 : 32;                        /* :484 */
    if (guard < 8) guard = 8;
    float pn[HDR_L], d[HDR_L];
    for (int i = 0; i < HDR_L; i++) pn[i] = 2.0f * (float)hdr_pn[i] - 1.0f;
    int best_s = 0; double best = -1.0;
    for (int s = -max_shift; s <= max_shift; s++) {                     /* :487-497 */
        const int i0 = base + s;
        if (i0 < 0 || i0 + HDR_L > nwin) continue;
        for (int i = guard; i < HDR_L; i++) d[i] = win[i0 + i] * pn[i];
        const double score = fabs((double)pairwise_f32(d + guard, HDR_L - guard));
        if (score > best) { best = score; best_s = s; }
    }
    const int i0 = base + best_s;
    for (int i = 0; i < HDR_L; i++) d[i] = win[i0 + i] * pn[i];          /* :498-500 */
    float sums[16], asum[16], dd[HDR_L];
    int npos = 0; unsigned val = 0;
    for (int b = 0; b < 16; b++) {                                      /* :503-509 */
        sums[b] = pairwise_f32(d + 8 * b, 8);
        asum[b] = fabsf(sums[b]);
        val = (val << 1) | (sums[b] < 0.0f ? 1u : 0u);
        npos += sums[b] > 0.0f;
    }
    for (int i = 0; i < HDR_L; i++) dd[i] = d[i] * d[i];
    const float mean_abs = pairwise_f32(asum, 16) / 16.0f;
    const float rms = sqrtf(pairwise_f32(dd, HDR_L) / (float)HDR_L) + (float)1e-12;   /* :505 */
    const float margin = mean_abs / rms;
    const float mu = pairwise_f32(d, HDR_L) / (float)HDR_L;              /* np.std(d), :512 */
    for (int i = 0; i < HDR_L; i++) { const float c = d[i] - mu; dd[i] = c * c; }
    const float sd = sqrtf(pairwise_f32(dd, HDR_L) / (float)HDR_L) + (float)1e-12;
    out[0] = (npos >= 10) && (margin > 0.5f);                           /* :513 */
    out[1] = (double)val;
    out[2] = (double)(mean_abs / sd);
    if (diag) { diag[0] = best_s; diag[1] = best; }
    free(mf);
}
