/* oracle/kdtree_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's KNN / one-sided-Hausdorff hot path:
 *   - nanoflann 1.4.2 kd-tree build + exact search (external/nanoflann/nanoflann.hpp, vendored
 *     in the reference) and
 *   - the pcu driver around it (src/point_cloud_distance.cpp:21-99, :211-225).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (point_cloud_utils_amd) never does.
 *
 * Parity pinning: this restatement is checked bit-for-bit (indices, distance bits, and the
 * kd-tree permutation itself) against oracle/_ref/libpcu_ref.so, which is the reference's own
 * nanoflann.hpp compiled in place with the reference's flags, and against the committed
 * fixtures in tests/golden/ that were generated from that library (tests/test_oracle.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile). No -ffast-math.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define T float
#define SUF f32
#define TMAX FLT_MAX
#define SQRT(x) sqrtf(x)
#include "kdtree_body.inc"
#undef T
#undef SUF
#undef TMAX
#undef SQRT

#define T double
#define SUF f64
#define TMAX DBL_MAX
#define SQRT(x) sqrt(x)
#include "kdtree_body.inc"
#undef T
#undef SUF
#undef TMAX
#undef SQRT

/* Secondary oracle: exact-arithmetic brute force, result ordered by (d2, index).
 * Equals the kd-tree result whenever no exact distance tie touches the top-k.
 * `out_tie` (nullable, nq bytes) is set when the k-th and (k+1)-th, or two kept, d2 are equal. */
#define BRUTE(T, SUF, TMAXV, SQRTF)                                                                      \
int pcu_oracle_brute_knn_##SUF(const T* q, int64_t nq, const T* r, int64_t nr, int k, int squared,       \
                               T* out_d, int64_t* out_i, uint8_t* out_tie) {                             \
    if (k <= 0 || nq <= 0 || nr <= 0) return -1;                                                         \
    T* bd = (T*)malloc((size_t)(k + 1) * sizeof(T));                                                     \
    int64_t* bi = (int64_t*)malloc((size_t)(k + 1) * sizeof(int64_t));                                   \
    for (int64_t i = 0; i < nq; ++i) {                                                                   \
        int64_t cnt = 0;                                                                                 \
        const T qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];                                     \
        for (int64_t j = 0; j < nr; ++j) {                                                               \
            T dx = qx - r[3 * j], dy = qy - r[3 * j + 1], dz = qz - r[3 * j + 2];                        \
            T d = 0; d += dx * dx; d += dy * dy; d += dz * dz;                                           \
            /* keep k+1 smallest by (d, j); j ascending so equal d never displaces an earlier one */     \
            if (cnt == k + 1 && !(d < bd[k])) continue;                                                  \
            int64_t p = cnt < k + 1 ? cnt : k;                                                           \
            while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }                \
            bd[p] = d; bi[p] = j;                                                                        \
            if (cnt < k + 1) ++cnt;                                                                      \
        }                                                                                                \
        int64_t found = cnt < k ? cnt : k;                                                               \
        uint8_t tie = 0;                                                                                 \
        for (int64_t j = 1; j < cnt; ++j) if (bd[j] == bd[j - 1]) tie = 1;                               \
        if (out_tie) out_tie[i] = tie;                                                                   \
        for (int64_t j = 0; j < found; ++j) { out_i[i * k + j] = bi[j]; out_d[i * k + j] = squared ? bd[j] : SQRTF(bd[j]); } \
        for (int64_t j = found; j < k; ++j) { out_i[i * k + j] = -1; out_d[i * k + j] = (T)-1.0; }       \
    }                                                                                                    \
    free(bd); free(bi);                                                                                  \
    return 0;                                                                                            \
}
BRUTE(float, f32, FLT_MAX, sqrtf)
BRUTE(double, f64, DBL_MAX, sqrt)
