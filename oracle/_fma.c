/* Test infrastructure for oracle/em_numpy.py: element-wise fused multiply-add over double arrays.
 * NumPy has no fma ufunc, but the reference's np.dot / np.linalg.norm on 2- and 3-vectors round as a
 * fused chain (BLAS), and the oracle has to round the same way to stay bit-faithful. */
#include <math.h>
#include <stddef.h>

void vpk_vfma(const double* a, const double* b, const double* c, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = fma(a[i], b[i], c[i]);
}

/* the reference squares a NumPy float64 *scalar* with `** 2` (probability_functions.py:174,222), which
 * calls libm pow(x, 2.0) -- not always the correctly rounded x*x */
void vpk_vpow2(const double* a, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = pow(a[i], 2.0);
}
