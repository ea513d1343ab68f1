// oracle/ref_morton_shim.cpp -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
//
// C-ABI driver around the reference's own MortonCode64 (/root/reference/src/common/morton_code.cpp + .h, compiled where they
// lie by oracle/Makefile into oracle/_ref/libpcu_ref_morton.so). Restates the element loops of src/morton.cpp: morton_encode
// (:185-243), morton_decode (:258-307), morton_add / morton_subtract (:27-91, :109-171) and the window selection of morton_knn
// (:339-383; the sort_dist comparator of :384-403 reads uninitialised query coordinates and is not restated).
#include <cstdint>
#include <cstddef>
#include <algorithm>
#include "morton_code.h"

extern "C" {
void pcu_ref_morton_encode(const int32_t* pts, int64_t n, uint64_t* codes) {
    for (int64_t i = 0; i < n; ++i) { int32_t px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2]; MortonCode64 code(px, py, pz); codes[i] = code.get_data(); }
}
void pcu_ref_morton_decode(const uint64_t* codes, int64_t n, int32_t* pts) {
    for (int64_t i = 0; i < n; ++i) { int32_t px, py, pz; MortonCode64(codes[i]).decode(px, py, pz); pts[3 * i] = px; pts[3 * i + 1] = py; pts[3 * i + 2] = pz; }
}
void pcu_ref_morton_addsub(const uint64_t* c1, const uint64_t* c2, int64_t n, int sub, uint64_t* out) {
    for (int64_t i = 0; i < n; ++i) { MortonCode64 a(c1[i]), b(c2[i]); out[i] = (sub ? (a - b) : (a + b)).get_data(); }
}
int pcu_ref_morton_knn_window(const uint64_t* codes, int64_t n, const uint64_t* qcodes, int64_t m, int k, int64_t* nn) {
    k = std::min<int64_t>(k, n);
    for (int64_t i = 0; i < m; ++i) {
        const uint64_t* code_ptr = std::lower_bound(codes, codes + n, qcodes[i]);
        std::ptrdiff_t idx = code_ptr - codes;
        const int half_k_up = k / 2, half_k_down = k - half_k_up;
        std::ptrdiff_t upper_bound = idx + half_k_up, lower_bound = idx - half_k_down;
        if (upper_bound >= n) { lower_bound -= (upper_bound - n); upper_bound = n; }
        if (lower_bound < 0) { upper_bound += -lower_bound; lower_bound = 0; }
        for (int j = 0; j < (upper_bound - lower_bound); ++j) nn[i * k + j] = lower_bound + j;
    }
    return k;
}
}
