// csrc/normals.h -- point-cloud normals by local plane fits.
//
// Replaces estimate_local_normal_knn / estimate_local_normal_rbf (src/point_cloud_normals.cpp:115-173, :48-113) and the
// drivers estimate_normals_* (:175-302). The reference fits a plane to a point's neighbourhood -- its k nearest neighbours
// (self-KNN on nanoflann, :139) or all points inside a ball (nanoflann radiusSearch, :75) -- by taking the right singular
// vector of the smallest singular value of the (m, 3) matrix A of (weighted) neighbour offsets (Eigen::JacobiSVD, :155-160).
// That vector is the eigenvector of the smallest eigenvalue of the 3x3 matrix A^T A, which is what is computed here:
//   * the offsets are formed exactly as the reference forms them: (p_j - q) in the INPUT type, then widened to double and
//     scaled by the weight (double);
//   * A^T A is accumulated in double, one neighbour after the other in the reference's order (ascending distance);
//   * a cyclic Jacobi eigen-solver (double, in registers) diagonalises it.
// The sign of a singular vector is a convention of Eigen's implementation (not part of the reference checkout): without view
// directions the normal is defined up to sign; with view directions it is flipped towards the sensor and filtered exactly as
// :161-169 does (sign(), acos(), strict '>').
#pragma once
#include "pcu_types.h"
#include "grid.h"

namespace pcu {

struct Sym3 { double xx, xy, xz, yy, yz, zz; };

// Eigenvector of the smallest eigenvalue of the symmetric positive semi-definite matrix S (unit length).
__device__ inline void smallest_eigenvector(const Sym3& S, double& nx, double& ny, double& nz) {
    double a[3][3] = {{S.xx, S.xy, S.xz}, {S.xy, S.yy, S.yz}, {S.xz, S.yz, S.zz}};
    const double tr = S.xx + S.yy + S.zz;
    if (!(tr > 0) || !isfinite(tr)) { nx = 0; ny = 0; nz = 1; return; }       // no spread at all (all offsets zero) or non-finite input
    const double sc = 1.0 / tr;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[i][j] *= sc;
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    bool last = false;
    for (int sweep = 0; sweep < 12; ++sweep) {
        // (the matrix is scaled to trace 1 and cyclic Jacobi converges quadratically. An eigenvector is off by ~ off / gap, and thin
        // neighbourhoods -- near-collinear points: two eigenvalues of 1e-8 and 1e-16 -- have gaps far below 1: so the sweeps stop ONE SWEEP AFTER
        // the off-diagonal mass has fallen below 1e-20 (that sweep squares it: beyond double precision for any gap the data can resolve), not at
        // the threshold itself; round 4 ran all twelve sweeps for every point, round 5 stopped at the threshold)
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (last || off == 0.0) break;
        last = off < 1e-20;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = r == 2 ? 1 : 0, q = r == 0 ? 1 : 2;                 // (0,1), (0,2), (1,2)
            const double apq = a[p][q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;          // (correctly rounded: rsqrt() is not, and the rotations accumulate in v)
            a[p][p] -= t * apq; a[q][q] += t * apq; a[p][q] = 0; a[q][p] = 0;
            const int o = 3 - p - q;                                           // the third index
            const double aop = a[o][p], aoq = a[o][q];
            a[o][p] = a[p][o] = c * aop - s * aoq;
            a[o][q] = a[q][o] = s * aop + c * aoq;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double vip = v[i][p], viq = v[i][q];
                v[i][p] = c * vip - s * viq; v[i][q] = s * vip + c * viq;
            }
        }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    const double x = m == 0 ? v[0][0] : (m == 1 ? v[0][1] : v[0][2]);
    const double y = m == 0 ? v[1][0] : (m == 1 ? v[1][1] : v[1][2]);
    const double z = m == 0 ? v[2][0] : (m == 1 ? v[2][1] : v[2][2]);
    const double inv = 1.0 / sqrt((x * x + y * y) + z * z);
    nx = x * inv; ny = y * inv; nz = z * inv;
}

// Orientation and filtering by the view direction (src/point_cloud_normals.cpp:161-169): returns false if the point is dropped.
__device__ __forceinline__ bool orient_and_filter(double& nx, double& ny, double& nz, double dx, double dy, double dz, double drop_angle_threshold) {
    const double d = (nx * dx + ny * dy) + nz * dz;
    const double sg = (double)((0.0 < d) - (d < 0.0));
    nx *= sg; ny *= sg; nz *= sg;
    const double ang = acos((nx * dx + ny * dy) + nz * dz);
    return !(ang > drop_angle_threshold);
}

template <typename T>
struct NormalsKnnArgs {
    const T* pts; const T* dirs;       // (n,3); dirs nullable
    const long long* nbr;              // (n,k) neighbour rows, ascending distance, -1 = not found
    int n, k;
    double drop;
    T* out_n; unsigned char* keep;     // (n,3), (n)
};

// One thread per point: its k neighbours (from the KNN pass) -> A^T A -> normal.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_normals_knn(const NormalsKnnArgs<T> a) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const long long* row = a.nbr + (size_t)i * a.k;
    const T qx = a.pts[3 * (size_t)i], qy = a.pts[3 * (size_t)i + 1], qz = a.pts[3 * (size_t)i + 2];
    Sym3 S = {0, 0, 0, 0, 0, 0};
    bool ok = row[a.k - 1] >= 0;       // founds < num_neighbors: the point is discarded (:143-146)
    if (ok) {
        struct __attribute__((packed, aligned(4))) P3 { T v[3]; };      // a neighbour = ONE 12 / 24-byte gather (three scalar loads cost the vector-memory path three instructions)
        // (a thread's row of k int64 ids lies k * 8 bytes from its neighbour's: every load of it is a per-lane gather, so the ids come two per 16-byte load)
        struct __attribute__((packed, aligned(8))) Id2 { long long v[2]; };
        auto add = [&](long long r) {
            const P3 nb = *reinterpret_cast<const P3*>(a.pts + 3 * r);
            const double ox = (double)(T)(nb.v[0] - qx), oy = (double)(T)(nb.v[1] - qy), oz = (double)(T)(nb.v[2] - qz);
            S.xx += ox * ox; S.xy += ox * oy; S.xz += ox * oz; S.yy += oy * oy; S.yz += oy * oz; S.zz += oz * oz;
        };
        int j = 0;
        for (; j + 1 < a.k; j += 2) { const Id2 p = *reinterpret_cast<const Id2*>(row + j); add(p.v[0]); add(p.v[1]); }       // (the reference's order: ascending distance)
        if (j < a.k) add(row[j]);
    }
    double nx = 0, ny = 0, nz = 0;
    if (ok) {
        smallest_eigenvector(S, nx, ny, nz);
        if (a.dirs) ok = orient_and_filter(nx, ny, nz, (double)a.dirs[3 * (size_t)i], (double)a.dirs[3 * (size_t)i + 1], (double)a.dirs[3 * (size_t)i + 2], a.drop);
    }
    a.out_n[3 * (size_t)i] = (T)nx; a.out_n[3 * (size_t)i + 1] = (T)ny; a.out_n[3 * (size_t)i + 2] = (T)nz;
    a.keep[i] = ok ? 1 : 0;
}

template <typename T>
struct NormalsBallArgs {
    const GridParams<T>* gp; const Pt4<T>* sorted; const unsigned* cell_start;   // grid index of the cloud (cells of about the search radius)
    const T* dirs; int n;
    T radius;                          // nanoflann's RadiusResultSet keeps points with d2 < radius, d2 the SQUARED distance: the
                                       // reference hands it ball_radius itself (:75), i.e. the ball really has radius sqrt(ball_radius)
    double ball_radius; int min_pts; int max_pts; int weight_rbf;
    double drop;
    T* out_n; unsigned char* keep;
};

// One lane per point (in cell order): every point of the cells within reach is tested with the reference's arithmetic
// (d2 = ((dx*dx)+(dy*dy))+(dz*dz) in T, `d2 < radius`), members are accumulated straight into A^T A. The reference sorts a
// neighbourhood by distance before the SVD; the order only changes the rounding of the sums (documented tolerance).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_normals_ball(const NormalsBallArgs<T> a) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= a.n) return;
    const GridParams<T>& g = *a.gp;
    const Pt4<T> q = a.sorted[t];
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const int ccx = grid_cell(g, 0, q.x), ccy = grid_cell(g, 1, q.y), ccz = grid_cell(g, 2, q.z);
    // cells that can hold a member: |coordinate difference| < sqrt(radius) (+ the face slack of the grid)
    const T reach = (T)sqrt((double)a.radius) * ((T)1 + (T)8 * Limits<T>::eps);
    const int R = (int)fmin(4096.0, ceil((double)(reach + g.slack[0] + g.slack[1] + g.slack[2]) * (double)g.inv_h));
    const int x0 = max(ccx - R, 0), x1 = min(ccx + R, Gx - 1);
    const int y0 = max(ccy - R, 0), y1 = min(ccy + R, Gy - 1);
    const int z0 = max(ccz - R, 0), z1 = min(ccz + R, Gz - 1);
    // max_pts_per_ball > 0: the reference keeps a RANDOM subset of that size of a larger neighbourhood (std::shuffle on rand(),
    // :82-85 -- not reproducible even between two runs of the reference). Here the subset is drawn systematically: every
    // (count / max_pts)-th member in scan order, i.e. evenly through the ball's cells; a second scan, only for such points.
    Sym3 S = {0, 0, 0, 0, 0, 0};
    int count = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const long long total = count;
        const bool subset = pass == 1;
        if (subset) { if (!(a.max_pts > 0 && total > a.max_pts)) break; S = Sym3{0, 0, 0, 0, 0, 0}; count = 0; }
        long long m = 0;
        for (int cz = z0; cz <= z1; ++cz)
            for (int cy = y0; cy <= y1; ++cy) {
                const int lo = row_run_lo(Gx, grid_row(Gy, cy, cz), x0, x1);
                const unsigned s = a.cell_start[lo], e = a.cell_start[lo + (x1 - x0 + 1)];
                for (unsigned p = s; p < e; ++p) {
                    const Pt4<T> c = a.sorted[p];
                    const T dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
                    const T d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
                    if (d2 < a.radius) {
                        if (subset) { const bool take = ((m + 1) * a.max_pts) / total > (m * a.max_pts) / total; ++m; if (!take) continue; }
                        double w = 1.0;
                        if (a.weight_rbf) {                 // Wendland weight of :330-335 on d = sqrt(d2) and ball_radius
                            const double r = sqrt((double)d2) / a.ball_radius, v1 = 1.0 - r, v2 = 4 * r + 1.0;
                            w = v1 * v1 * v1 * v1 * v2;
                        }
                        const double ox = (double)(T)(c.x - q.x) * w, oy = (double)(T)(c.y - q.y) * w, oz = (double)(T)(c.z - q.z) * w;
                        S.xx += ox * ox; S.xy += ox * oy; S.xz += ox * oz; S.yy += oy * oy; S.yz += oy * oz; S.zz += oz * oz;
                        ++count;
                    }
                }
            }
    }
    bool ok = count >= a.min_pts;
    double nx = 0, ny = 0, nz = 0;
    const size_t i = (size_t)q.idx;
    if (ok) {
        smallest_eigenvector(S, nx, ny, nz);
        if (a.dirs) ok = orient_and_filter(nx, ny, nz, (double)a.dirs[3 * i], (double)a.dirs[3 * i + 1], (double)a.dirs[3 * i + 2], a.drop);
    }
    a.out_n[3 * i] = (T)nx; a.out_n[3 * i + 1] = (T)ny; a.out_n[3 * i + 2] = (T)nz;
    a.keep[i] = ok ? 1 : 0;
}

}  // namespace pcu
