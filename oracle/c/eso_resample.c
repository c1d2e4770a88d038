/* eso_resample.c -- CPU ORACLE (test infrastructure only) for input conditioning:
 * resample_to (rtwm/utils.py:58-66) = scipy.signal.resample_poly(audio, up, down) with the default Kaiser(5.0) design.
 * The filter design (firwin) and the padding arithmetic are Python in SciPy and are reused by the host code as they
 * are; what is restated here is the compiled inner loop of scipy.signal.upfirdn (SciPy 1.15.3, `_upfirdn_apply`,
 * mode 'constant', cval 0), whose source is not in the image: for every output sample the products x[i] * h[...] are
 * added to an accumulator that starts at 0, in ascending input index, in the arithmetic of the output type (float32
 * when signal and filter are float32, else float64), multiply and add rounded separately.  Pinned bit for bit against
 * scipy.signal.resample_poly itself (tests/test_oracle_resample.py). */
#include <stdint.h>

/* h_tf = SciPy's transposed, flipped, zero-padded polyphase layout (_pad_h): phase t occupies h_tf[t*hpp .. (t+1)*hpp).
 * Output y of the full upfirdn result: t = (y*down) % up, x_idx = (y*down) / up,
 *   out[y] = sum_{j=0}^{hpp-1} x[x_idx - hpp + 1 + j] * h_tf[t*hpp + j]   (x outside [0, n_x) is 0).
 * Computes outputs y0 .. y0+n_out-1. */
#define ESO_UPFIRDN(NAME, T)                                                                                   \
void NAME(const T* x, long n_x, const T* h_tf, long hpp, long up, long down, long y0, long n_out, T* out)      \
{                                                                                                              \
    for (long k = 0; k < n_out; ++k) {                                                                         \
        const long yy = y0 + k;                                                                                \
        const long t = (yy * down) % up, x_idx = (yy * down) / up;                                             \
        long lo = x_idx - hpp + 1, hi = x_idx;                                                                 \
        long hidx = t * hpp;                                                                                   \
        if (lo < 0) { hidx -= lo; lo = 0; }                                                                    \
        if (hi > n_x - 1) hi = n_x - 1;                                                                        \
        volatile T acc = 0;                                                                                    \
        for (long i = lo; i <= hi; ++i) { const volatile T p = x[i] * h_tf[hidx++]; acc = acc + p; }           \
        out[k] = acc;                                                                                          \
    }                                                                                                          \
}
ESO_UPFIRDN(eso_upfirdn_f32, float)
ESO_UPFIRDN(eso_upfirdn_f64, double)
