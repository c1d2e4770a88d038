/* es_math.h -- float64 exp / log1p / logaddexp with a FIXED operation order.
 *
 * Why this exists.  The reference's SCL decoder (rtwm/fastpolar.py:18-23, 32-40) computes
 *     f(a,b)      = np.logaddexp(a,b) - np.logaddexp(0.0, a+b)
 *     penalty(l)  = np.log1p(np.exp(-abs(l)))
 * in float64.  NumPy's logaddexp inner loop is  max + log1p(exp(-|x-y|))  evaluated with the
 * C library's scalar exp()/log1p(), i.e. on Linux/x86-64 with glibc (>= 2.28):
 *     exp   : the table-driven algorithm of Arm's optimized-routines (N = 128, degree-5
 *             polynomial), built with FMA contraction on FMA-capable CPUs,
 *     log1p : the classic fdlibm algorithm with glibc's split Horner scheme, no FMA.
 * Path metrics are compared with exact float64 ordering, so a last-bit difference in exp/log1p can
 * flip a sort.  The device math library does not promise those exact roundings, so the kernel and
 * the CPU oracle both use THIS restatement, which fixes every rounding step (explicit fma where
 * glibc's FMA build fuses, separate mul/add elsewhere; compile with -ffp-contract=off).
 * tests/test_oracle_math.py checks it bit-for-bit against the host libm on millions of inputs.
 *
 * The same file is compiled by gcc (oracle, host tests) and by hipcc for gfx950 (kernels).
 */
#ifndef ES_MATH_H
#define ES_MATH_H

#include <stdint.h>
#include "es_exp_tab.h"

#if defined(__HIPCC__)
#define ES_HD __host__ __device__ __forceinline__
#else
#define ES_HD static inline
#endif

#define ES_FMA(a, b, c) __builtin_fma((a), (b), (c))

ES_HD uint64_t es_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
ES_HD double   es_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
ES_HD int32_t  es_hi32(double x) { return (int32_t)(es_d2u(x) >> 32); }

/* ---- exp ------------------------------------------------------------------------------- */
#define ES_EXP_INVLN2N   0x1.71547652b82fep+7     /* 128/ln2            */
#define ES_EXP_SHIFT     0x1.8p+52
#define ES_EXP_NLN2HI    (-0x1.62e42fefa0000p-8)  /* -ln2/128, high part */
#define ES_EXP_NLN2LO    (-0x1.cf79abc9e3b3ap-47) /* -ln2/128, low part  */
#define ES_EXP_C2        0x1.ffffffffffdbdp-2
#define ES_EXP_C3        0x1.555555555543cp-3
#define ES_EXP_C4        0x1.55555cf172b91p-5
#define ES_EXP_C5        0x1.1111167a4d017p-7

/* exp(x) for any finite or infinite x <= 0 and for 0 <= x < 512 (the decoder only ever passes
 * x = -|t|).  `tab` is the 256-word table of es_exp_tab.h (LDS copy on the device).           */
ES_HD double es_exp(double x, const uint64_t* tab)
{
    const uint64_t xb = es_d2u(x);
    uint32_t abstop = (uint32_t)(xb >> 52) & 0x7ffu;
    int special = 0;
    if (abstop - 0x3c9u >= 0x3fu) {
        if ((int32_t)(abstop - 0x3c9u) < 0)
            return 1.0 + x;                          /* |x| < 2^-54 (also +-0) */
        if (abstop >= 0x409u) {                      /* |x| >= 1024, inf, nan  */
            if (xb == 0xfff0000000000000ULL) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;    /* nan, +inf */
            return (xb >> 63) ? 0.0 : es_u2d(0x7ff0000000000000ULL);
        }
        special = 1;                                 /* 512 <= |x| < 1024 */
    }
    /* x = k*ln2/128 + r */
    double kd = ES_FMA(x, ES_EXP_INVLN2N, ES_EXP_SHIFT);
    const uint64_t ki = es_d2u(kd);
    kd = kd - ES_EXP_SHIFT;
    double r = ES_FMA(kd, ES_EXP_NLN2HI, x);
    r = ES_FMA(kd, ES_EXP_NLN2LO, r);
    const uint32_t idx = 2u * (uint32_t)(ki & 127u);
    const uint64_t top = ki << 45;
    const double tail = es_u2d(tab[idx]);
    uint64_t sbits = tab[idx + 1] + top;
    const double p23 = ES_FMA(r, ES_EXP_C3, ES_EXP_C2);
    const double tr = r + tail;
    const double r2 = r * r;
    const double p45 = ES_FMA(r, ES_EXP_C5, ES_EXP_C4);
    const double t = ES_FMA(p23, r2, tr);
    const double r4 = r2 * r2;
    const double tmp = ES_FMA(r4, p45, t);
    if (!special) {
        const double scale = es_u2d(sbits);
        return ES_FMA(scale, tmp, scale);
    }
    if ((ki & 0x80000000ULL) == 0) {                 /* k > 0: not reached for x <= 0 */
        sbits -= 1009ULL << 52;
        const double scale = es_u2d(sbits);
        return 0x1p1009 * ES_FMA(scale, tmp, scale);
    }
    /* k < 0: result may be subnormal; round once */
    sbits += 1022ULL << 52;
    const double scale = es_u2d(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
        double lo = (scale - y) + st;
        const double hi = 1.0 + y;
        lo = ((1.0 - hi) + y) + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
}

/* ---- log1p ------------------------------------------------------------------------------ */
#define ES_LN2_HI 6.93147180369123816490e-01
#define ES_LN2_LO 1.90821492927058770002e-10
#define ES_LP1 6.666666666666735130e-01
#define ES_LP2 3.999999999940941908e-01
#define ES_LP3 2.857142874366239149e-01
#define ES_LP4 2.222219843214978396e-01
#define ES_LP5 1.818357216161805012e-01
#define ES_LP6 1.531383769920937332e-01
#define ES_LP7 1.479819860511658591e-01

/* log1p(x) for finite x > -1 (the decoder passes x = exp(-|t|) in [0,1]). */
ES_HD double es_log1p(double x)
{
    const int32_t hx = es_hi32(x);
    const int32_t ax = hx & 0x7fffffff;
    int32_t k = 1, hu = 0;
    double f = 0.0;
    double cn = 0.0, cd = 1.0;                        /* correction term c = cn / cd */

    if (hx < 0x3FDA827A) {                            /* x < sqrt(2)-1 */
        if (ax >= 0x3ff00000) {                       /* x <= -1: not used by the decoder */
            if (x == -1.0) return -es_u2d(0x7ff0000000000000ULL);
            return (x - x) / (x - x);
        }
        if (ax < 0x3e200000) {                        /* |x| < 2^-29 */
            if (ax < 0x3c900000) return x;            /* |x| < 2^-54 */
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= (int32_t)0xbfd2bec4) {    /* -0.2929 < x < 0.41422 */
            k = 0; f = x; hu = 1;
        }
    } else if (hx >= 0x7ff00000) {
        return x + x;
    }
    if (k != 0) {
        double u;
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = es_hi32(u);
            k = (hu >> 20) - 1023;
            cn = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
            cd = u;
        } else {
            u = x;
            hu = es_hi32(u);
            k = (hu >> 20) - 1023;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = es_u2d((es_d2u(u) & 0xffffffffULL) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32));
        } else {
            k += 1;
            u = es_u2d((es_d2u(u) & 0xffffffffULL) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32));
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    /* fdlibm divides right where c is formed; dividing here instead (0/1 when k == 0) yields the
       same correctly rounded quotient and lets this division overlap the one for s below. */
    double c = cn / cd;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    if (hu == 0) {                                    /* |f| < 2^-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c = c + dk * ES_LN2_LO;
            return dk * ES_LN2_HI + c;
        }
        const double R0 = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R0;
        return dk * ES_LN2_HI - ((R0 - (dk * ES_LN2_LO + c)) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R1 = z * ES_LP1;
    const double z2 = z * z;
    const double R2 = ES_LP2 + z * ES_LP3;
    const double z4 = z2 * z2;
    const double R3 = ES_LP4 + z * ES_LP5;
    const double z6 = z4 * z2;
    const double R4 = ES_LP6 + z * ES_LP7;
    const double R = ((R1 + z2 * R2) + z4 * R3) + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ES_LN2_HI - ((hfsq - (s * (hfsq + R) + (dk * ES_LN2_LO + c))) - f);
}

/* max of two numbers (never NaN here): one of the operands, bit for bit.  On the device a bare v_max_f64 -- the builtin would first
 * "canonicalise" operands that come straight from memory (two more instructions) to quieten signalling NaNs that cannot occur. */
ES_HD double es_max_num(double x, double y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#else
    return x > y ? x : y;
#endif
}

/* IEEE-754 quotient n / d for operands in the benign range the straight-line softplus below passes (d in [1, 3],
 * n zero or of magnitude in [2^-120, 2]): on the device, the division sequence hipcc emits for `/` (v_rcp_f64, two Newton
 * steps, quotient, residual, one correction) WITHOUT its range guards -- v_div_scale (which scales only operands whose
 * exponents are extreme: none here), v_div_fmas (a plain fma when nothing was scaled) and v_div_fixup (zero / inf / nan /
 * denormal patch-ups: none here).  Same arithmetic, hence the same correctly rounded quotient, 8 instructions instead of 11.
 * tests/test_gpu_parity.py::test_device_softplus_bits checks the device function against the C library bit for bit. */
ES_HD double es_div_normal(double n, double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    double e = ES_FMA(-d, r, 1.0);
    r = ES_FMA(r, e, r);
    e = ES_FMA(-d, r, 1.0);
    r = ES_FMA(r, e, r);
    const double q = n * r;
    e = ES_FMA(-d, q, n);
    return ES_FMA(e, r, q);
#else
    return n / d;
#endif
}

/* ---- numpy's logaddexp inner loop (npymath npy_logaddexp) --------------------------------- */
#define ES_LOGE2 0.693147180559945309417232121458176568

/* softplus of a non-positive argument: log1p(exp(t)), t = -|d|  -- generic (branchy) form */
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline)) static double es_softplus_neg_generic(double t, const uint64_t* tab) { return es_log1p(es_exp(t, tab)); }
#else
ES_HD double es_softplus_neg_generic(double t, const uint64_t* tab) { return es_log1p(es_exp(t, tab)); }
#endif

/* Straight-line (branch-free) evaluation of the same value for |t| < 512:
 * es_exp's main path followed by es_log1p's tiny / k == 0 / k != 0 paths AND its |f| < 2^-20 corner, merged with selects.
 * Every arithmetic step is the one the generic functions perform for that input, so the result is
 * bit-identical; *ok = 0 flags inputs outside that range (caller falls back to the generic form).
 * Having no control flow lets the compiler interleave several independent evaluations, which is
 * what hides the ~1.3 k-cycle dependent latency of one evaluation on a single wavefront.
 * (ES_SOFTPLUS_CORNER 0 leaves the corner to the fall-back, as rounds 1-2 did: 2-17 % of the f evaluations from depth 5 down
 * land in it -- more the weaker the signal --, so nearly every wave ran the generic form beside the straight-line one.) */
#ifndef ES_SOFTPLUS_CORNER
#define ES_SOFTPLUS_CORNER 1
#endif
ES_HD double es_softplus_neg_fast(double t, const uint64_t* tab, int* ok)
{
    /* ---- exp(t), main path of es_exp ---- */
    double kd = ES_FMA(t, ES_EXP_INVLN2N, ES_EXP_SHIFT);
    const uint64_t ki = es_d2u(kd);
    kd = kd - ES_EXP_SHIFT;
    double r = ES_FMA(kd, ES_EXP_NLN2HI, t);
    r = ES_FMA(kd, ES_EXP_NLN2LO, r);
    const uint32_t idx = 2u * (uint32_t)(ki & 127u);
#if defined(__HIP_DEVICE_COMPILE__) && defined(ES_EXP_TAB_LDS_ADDR)
    /* the including kernel keeps its copy of the table at a KNOWN LDS address (it checks that at run time): the entry's address is then
       the scaled index itself, without the add of a link-time base that the compiler does not fold (one instruction per evaluation) */
    typedef __attribute__((address_space(3))) const uint64_t es_lds_u64;
    es_lds_u64* const te = (es_lds_u64*)(uint32_t)((ES_EXP_TAB_LDS_ADDR) + idx * 8u);
    const double tail = es_u2d(te[0]);
    const uint64_t sbits = te[1] + (ki << 45);
    (void)tab;
#else
    const double tail = es_u2d(tab[idx]);
    const uint64_t sbits = tab[idx + 1] + (ki << 45);
#endif
    const double p23 = ES_FMA(r, ES_EXP_C3, ES_EXP_C2);
    const double tr = r + tail;
    const double r2 = r * r;
    const double p45 = ES_FMA(r, ES_EXP_C5, ES_EXP_C4);
    const double tq = ES_FMA(p23, r2, tr);
    const double r4 = r2 * r2;
    const double tmp = ES_FMA(r4, p45, tq);
    const double scale = es_u2d(sbits);
    const double y = ES_FMA(scale, tmp, scale);           /* in (0, 1] */

    /* ---- log1p(y) ---- */
    const int32_t hy = es_hi32(y);
    const int tiny29 = hy < 0x3e200000;                   /* y < 2^-29  -> y - y*y/2 (which is y itself below 2^-54) */
    const int k0 = hy < 0x3FDA827A;                       /* y < sqrt(2)-1: k = 0, f = y */
    const double u = 1.0 + y;
    const uint32_t hu0 = (uint32_t)es_hi32(u);
    /* y <= 1 here, so u <= 2 and es_log1p's two forms of the rounding error of 1 + y coincide: for u < 2 (exponent 0) it
       takes y - (u - 1); at u == 2 (y == 1, exponent 1) its 1 - (u - y) and this expression are both exactly 0 */
    const double cn1 = y - (u - 1.0);
    /* k != 0.  Then k == 1: y >= sqrt(2)-1 puts 1 + y in [1.41421353.., 2], whose high word is >= 0x3ff6a09e (fdlibm: k = 0 + 1) or, at
       u == 2, has a zero fraction (k = 1 + 0).  fdlibm rewrites u's exponent so that u/2 remains (1.0 when u == 2) and takes
       f = that - 1: halving is exact, so ONE fma(u, 0.5, -1) rounds exactly as that subtraction does.  With k == 1 the products
       k*ln2_hi and k*ln2_lo are the constants themselves. */
    const double f1 = ES_FMA(u, 0.5, -1.0);
    const double f = k0 ? y : f1;
    const double c = es_div_normal(cn1, u);               /* only read when k != 0, where es_log1p divides exactly these */
    const double hfsq = 0.5 * f * f;
    const double s = es_div_normal(f, 2.0 + f);
    const double z = s * s;
    const double R1 = z * ES_LP1;
    const double z2 = z * z;
    const double R2 = ES_LP2 + z * ES_LP3;
    const double z4 = z2 * z2;
    const double R3 = ES_LP4 + z * ES_LP5;
    const double z6 = z4 * z2;
    const double R4 = ES_LP6 + z * ES_LP7;
    const double R = ((R1 + z2 * R2) + z4 * R3) + z6 * R4;
    const double sR = s * (hfsq + R);
    const double res0 = f - (hfsq - sR);
    const double resk = ES_LN2_HI - ((hfsq - (sR + (ES_LN2_LO + c))) - f);
    double res = k0 ? res0 : resk;
#if ES_SOFTPLUS_CORNER
    /* fdlibm's |f| < 2^-20 corner, which with k == 1 is u within 3 * 2^-20 below 2 (the fraction field of its high word >= 0xffffd), i.e.
       1.1e-16 < |t| < 2.86e-6: differences and sums of LLRs a few f levels down the tree are that small all the time (f(x, y) ~ x*y/2 for
       small operands), so the corner is part of the straight-line form, not of the fall-back.  f != 0 there (u < 2). */
    const double Rc = hfsq * (1.0 - 0.66666666666666666 * f);
    const double resc = ES_LN2_HI - ((Rc - (ES_LN2_LO + c)) - f);
    res = ((uint32_t)(hu0 - 0x3FFFFFFDu) < 3u) ? resc : res;
#endif
    const double rt = y - y * y * 0.5;
    res = tiny29 ? rt : res;
    /* Range of the straight-line form.  exp: every |t| < 512 -- below 2^-54 (in practice t == 0, which two LLRs clipped
     * to the same +-12 produce all the time) the main path yields exactly the 1.0 that es_exp's early return (1 + t)
     * rounds to.  log1p: its |f| < 2^-20 corner (fdlibm's hu == 0: for k == 1 the fraction field of u's high word is >= 0xffffd,
     * i.e. the high word is 0x3ffffffd..0x3fffffff) is the select above; at u == 2 (y == 1, that t == 0 case; high word
     * 0x40000000) f == 0 and fdlibm's shortcut  k*ln2_hi + (c + k*ln2_lo)  is what resk evaluates to term by term. */
#if ES_SOFTPLUS_CORNER
    *ok = (__builtin_fabs(t) < 512.0);
#else
    *ok = (__builtin_fabs(t) < 512.0) && ((uint32_t)(hu0 - 0x3FFFFFFDu) >= 3u);
#endif
    return res;
}

ES_HD double es_softplus_neg(double t, const uint64_t* tab)
{
    int ok;
    double r = es_softplus_neg_fast(t, tab, &ok);
    if (!ok) r = es_softplus_neg_generic(t, tab);
    return r;
}

/* Branch-free form of npy_logaddexp (identical values: for d > 0 the reference evaluates
 * x + log1p(exp(-d)), otherwise y + log1p(exp(d)); x == y short-circuits to x + ln 2).  Lanes of
 * a wavefront disagree on the sign of d all the time, so the two-branch form would run exp/log1p
 * twice.  *sp receives log1p(exp(-|x-y|)), which the decoder reuses as a path-metric penalty. */
ES_HD double es_logaddexp_sp(double x, double y, const uint64_t* tab, double* sp)
{
    const double d = x - y;
    const int pos = d > 0;
    const double t = pos ? -d : d;
    const double L = es_softplus_neg(t, tab);
    *sp = L;
    double r = (pos ? x : y) + L;
    if (x == y) r = x + ES_LOGE2;
    return r;
}

ES_HD double es_logaddexp(double x, double y, const uint64_t* tab)
{
    double sp;
    return es_logaddexp_sp(x, y, tab, &sp);
}

/* rtwm/fastpolar.py:18-23  _f_function; also returns the two softplus terms
 *   *sp_diff = log1p(exp(-|a-b|)),  *sp_sum = log1p(exp(-|a+b|))
 * which are exactly the penalties log1p(exp(-|g|)) of the sibling leaf g = b -/+ a. */
ES_HD double es_polar_f_sp(double a, double b, const uint64_t* tab, double* sp_diff, double* sp_sum)
{
    /* npy_logaddexp(x, y) is  max(x, y) + log1p(exp(-|x - y|))  -- its branch on the sign of x - y picks exactly that -- and
       x + ln 2 when x == y, which the formula gives too: log1p(exp(-0)) evaluates to ln2_hi + ln2_lo = the double nearest ln 2 =
       npymath's LOGE2 (tests/test_oracle_math.py pins f against NumPy on equal and opposite operands).  Two independent softplus
       evaluations, written out so that they can be interleaved. */
    const double d1 = a - b;
    const double sum = a + b;
    const double t1 = -__builtin_fabs(d1);
    const double t2 = -__builtin_fabs(sum);
    int ok1, ok2;
    double L1 = es_softplus_neg_fast(t1, tab, &ok1);
    double L2 = es_softplus_neg_fast(t2, tab, &ok2);
    if (!ok1) L1 = es_softplus_neg_generic(t1, tab);
    if (!ok2) L2 = es_softplus_neg_generic(t2, tab);
    *sp_diff = L1;
    *sp_sum = L2;
    const double r1 = es_max_num(a, b) + L1;             /* logaddexp(a, b)     */
    const double r2 = es_max_num(sum, 0.0) + L2;         /* logaddexp(0, a + b) */
    return r1 - r2;
}


/* The same value WITHOUT the generic fall-back: straight-line code only.  Lanes whose operands leave the range of the straight-line
 * softplus set *bad (or-ed in) and get an unspecified result; the caller recomputes those with es_polar_f_sp (a cold path OUTSIDE its hot
 * loop: a call inside a loop makes the compiler wait for every outstanding load at the loop head and keeps the two chains apart). */
ES_HD double es_polar_f_fast_sp(double a, double b, const uint64_t* tab, double* sp_diff, double* sp_sum, int* bad)
{
    const double d1 = a - b;
    const double sum = a + b;
    int ok1, ok2;
    const double L1 = es_softplus_neg_fast(-__builtin_fabs(d1), tab, &ok1);
    const double L2 = es_softplus_neg_fast(-__builtin_fabs(sum), tab, &ok2);
    *bad |= !(ok1 & ok2);
    *sp_diff = L1;
    *sp_sum = L2;
    const double r1 = es_max_num(a, b) + L1;
    const double r2 = es_max_num(sum, 0.0) + L2;
    return r1 - r2;
}

ES_HD double es_polar_f_fast(double a, double b, const uint64_t* tab, int* bad)
{
    double s0, s1;
    return es_polar_f_fast_sp(a, b, tab, &s0, &s1, bad);
}

ES_HD double es_polar_f(double a, double b, const uint64_t* tab)
{
    double s0, s1;
    return es_polar_f_sp(a, b, tab, &s0, &s1);
}

/* rtwm/fastpolar.py:26-29  _g_function:  b + (1 - 2u) * a  */
ES_HD double es_polar_g(double a, double b, uint32_t u)
{
    /* (1 - 2u) is exactly +1 or -1 and (+-1) * a is exactly +-a: flipping the sign bit gives the same addend */
    return b + es_u2d(es_d2u(a) ^ ((uint64_t)(u & 1u) << 63));
}

/* rtwm/fastpolar.py:32-40  _metric_penalty (glibc-exact flavour; see DESIGN.md "penalty") */
ES_HD double es_metric_penalty(double llr, uint32_t bit, const uint64_t* tab)
{
    const double al = __builtin_fabs(llr);
    double p = es_log1p(es_exp(-al, tab));
    const uint32_t preferred = (llr >= 0.0) ? 1u : 0u;
    if (bit != preferred) p = p + al;
    return p;
}

#endif /* ES_MATH_H */
