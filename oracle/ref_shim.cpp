// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
//
// Thin C-ABI driver around the reference's *vendored, unmodified* nanoflann
// (/root/reference/external/nanoflann/nanoflann.hpp, v1.4.2), #included where it lies.
// It restates the driver loop of /root/reference/src/point_cloud_distance.cpp:21-99
// (shortest_distances_nanoflann) and the arg-max of :221-225 (one_sided_hausdorff_distance),
// with the same kd-tree instantiation (:37), the same triple tree build (:41-42 +
// nanoflann.hpp:1357,:2288) and the same OpenMP policy (:29-30, common.h:182-212).
//
// The reference module itself cannot be built offline (numpyeigen / Eigen are fetched at
// configure time), so two stand-ins are needed:
//   * PyErr_CheckSignals()/pybind11::error_already_set -- referenced by the one patched
//     line nanoflann.hpp:1004; stubbed to "no signal pending";
//   * a minimal Eigen-like dense matrix (Scalar, Index, ColsAtCompileTime, rows/cols/coeff),
//     which is all KDTreeEigenMatrixAdaptor (nanoflann.hpp:2244-2352) touches.
//
// Built by oracle/Makefile into oracle/_ref/libpcu_ref.so with the reference's flags
// (CMakeLists.txt:3-4,151,223): -O3 -DNDEBUG -std=c++17 -msse3 -fopenmp.
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <array>
#include <vector>
#include <thread>
#include <functional>
#include <stdexcept>
#include <chrono>
#if defined(_OPENMP)
#include <omp.h>
#endif

#define PyErr_CheckSignals() 0
namespace pybind11 { struct error_already_set {}; }

#include "nanoflann.hpp"

namespace {

template <typename T>
struct DenseRowMajor {          // Eigen::Matrix<T, Dynamic, Dynamic, RowMajor> stand-in
    using Scalar = T;
    using Index  = std::ptrdiff_t;   // Eigen::Index
    enum { ColsAtCompileTime = -1, RowsAtCompileTime = -1 };
    std::vector<T> buf;
    Index nrows, ncols;
    DenseRowMajor(const T* p, Index r, Index c) : buf(p, p + r * c), nrows(r), ncols(c) {}  // deep copy (:152-153)
    Index rows() const { return nrows; }
    Index cols() const { return ncols; }
    T coeff(Index i, Index j) const { return buf[i * ncols + j]; }
    T operator()(Index i, Index j) const { return buf[i * ncols + j]; }
};

// common.h:182-212
struct OmpSetParallelism {
    int old_num_threads = 1;
    bool is_nop;
    OmpSetParallelism(int num_threads, bool use_parallel) {
        is_nop = !use_parallel;
#if defined(_OPENMP)
        if (is_nop) return;
        old_num_threads = omp_get_num_threads();
        if (num_threads < 0) omp_set_num_threads((int)std::thread::hardware_concurrency());
        else omp_set_num_threads(num_threads);
#endif
    }
    ~OmpSetParallelism() {
#if defined(_OPENMP)
        if (!is_nop) omp_set_num_threads(old_num_threads);
#endif
    }
};

// point_cloud_distance.cpp:21-99. `timing` (nullable): [0]=copy+build seconds, [1]=search seconds.
template <typename T>
void shortest_distances(const T* q, int64_t nq, const T* r, int64_t nr, int num_nbrs,
                        bool squared_dist, int max_points_per_leaf, int num_threads,
                        T* distances, int64_t* corrs, double* timing) {
    using Mat = DenseRowMajor<T>;
    auto t0 = std::chrono::steady_clock::now();
    Mat query_mat(q, nq, 3);
    Mat dataset_mat(r, nr, 3);

    const int MIN_PARALLEL_INPUT_SIZE = 100000;
    const bool run_parallel = query_mat.rows() >= MIN_PARALLEL_INPUT_SIZE && num_threads != 0;
    OmpSetParallelism set_parallel(num_threads, run_parallel);

    using KdTreeType = nanoflann::KDTreeEigenMatrixAdaptor<Mat, 3, nanoflann::metric_L2_Simple>;
    using IndexType  = typename KdTreeType::IndexType;
    using ScalarType = typename KdTreeType::num_t;

    KdTreeType mat_index(3, std::cref(dataset_mat), max_points_per_leaf);
    mat_index.index->buildIndex();
    auto t1 = std::chrono::steady_clock::now();

#if defined(_OPENMP)
#pragma omp parallel if (run_parallel)
#endif
    {
        std::array<ScalarType, 3> query_point;
        std::vector<IndexType> out_indices(num_nbrs);
        std::vector<ScalarType> out_dists_sqr(num_nbrs);
#if defined(_OPENMP)
#pragma omp for
#endif
        for (int i = 0; i < (int)query_mat.rows(); ++i) {
            for (int j = 0; j < 3; ++j) query_point[j] = query_mat(i, j);
            const size_t founds = mat_index.index->knnSearch(query_point.data(), num_nbrs,
                                                             out_indices.data(), out_dists_sqr.data());
            for (size_t k = 0; k < founds; k++) {
                corrs[(int64_t)i * num_nbrs + k] = out_indices[k];
                if (squared_dist) distances[(int64_t)i * num_nbrs + k] = out_dists_sqr[k];
                else              distances[(int64_t)i * num_nbrs + k] = sqrt(out_dists_sqr[k]);
            }
            for (int k = (int)founds; k < num_nbrs; k++) {
                corrs[(int64_t)i * num_nbrs + k] = -1;
                distances[(int64_t)i * num_nbrs + k] = -1.0;
            }
        }
    }
    auto t2 = std::chrono::steady_clock::now();
    if (timing) {
        timing[0] = std::chrono::duration<double>(t1 - t0).count();
        timing[1] = std::chrono::duration<double>(t2 - t1).count();
    }
}

// point_cloud_distance.cpp:211-225: k=1, num_threads defaulted to 0 (serial), first max wins.
template <typename T>
void one_sided_hausdorff(const T* s, int64_t ns, const T* t, int64_t nt, bool squared,
                         int max_leaf, T* out_d, int64_t* out_i, int64_t* out_j, double* timing) {
    std::vector<T> dists(ns);
    std::vector<int64_t> corrs(ns);
    shortest_distances<T>(s, ns, t, nt, 1, squared, max_leaf, /*num_threads=*/0, dists.data(), corrs.data(), timing);
    int64_t best = 0;                       // Eigen maxCoeff(&row,&col): strict '>' visitor, first max kept
    for (int64_t i = 1; i < ns; ++i) if (dists[i] > dists[best]) best = i;
    *out_d = dists[best];
    *out_i = best;
    *out_j = corrs[best];
}

}  // namespace

extern "C" {

int pcu_ref_knn_f32(const float* q, int64_t nq, const float* r, int64_t nr, int k, int squared,
                    int max_leaf, int num_threads, float* out_d, int64_t* out_i, double* timing) {
    try { shortest_distances<float>(q, nq, r, nr, k, squared != 0, max_leaf, num_threads, out_d, out_i, timing); }
    catch (...) { return -1; }
    return 0;
}
int pcu_ref_knn_f64(const double* q, int64_t nq, const double* r, int64_t nr, int k, int squared,
                    int max_leaf, int num_threads, double* out_d, int64_t* out_i, double* timing) {
    try { shortest_distances<double>(q, nq, r, nr, k, squared != 0, max_leaf, num_threads, out_d, out_i, timing); }
    catch (...) { return -1; }
    return 0;
}
int pcu_ref_one_sided_hausdorff_f32(const float* s, int64_t ns, const float* t, int64_t nt, int squared,
                                    int max_leaf, float* out_d, int64_t* out_i, int64_t* out_j, double* timing) {
    try { one_sided_hausdorff<float>(s, ns, t, nt, squared != 0, max_leaf, out_d, out_i, out_j, timing); }
    catch (...) { return -1; }
    return 0;
}
int pcu_ref_one_sided_hausdorff_f64(const double* s, int64_t ns, const double* t, int64_t nt, int squared,
                                    int max_leaf, double* out_d, int64_t* out_i, int64_t* out_j, double* timing) {
    try { one_sided_hausdorff<double>(s, ns, t, nt, squared != 0, max_leaf, out_d, out_i, out_j, timing); }
    catch (...) { return -1; }
    return 0;
}
int pcu_ref_hardware_concurrency(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
