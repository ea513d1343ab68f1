#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../point_cloud_utils_amd/csrc/kd_order.h"
using namespace pcu;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float T;
__global__ void k_dump(const KdNode<T>* nodes, int n) {
    for (int i = 0; i < n; ++i) printf("node %d [%d,%d) c1 %d c2 %d feat %d act %d\n", i, nodes[i].left, nodes[i].right, nodes[i].child1, nodes[i].child2, nodes[i].divfeat, nodes[i].active);
}
int main() {
    const int M = 200;
    std::vector<T> pts(M * 3);
    unsigned s = 12345; for (auto& v : pts) { s = s * 1664525u + 1013904223u; v = (s >> 8) / 16777216.0f; }
    T* dpts; CK(hipMalloc(&dpts, M * 3 * sizeof(T))); CK(hipMemcpy(dpts, pts.data(), M * 3 * sizeof(T), hipMemcpyHostToDevice));
    GridParams<T> hg{}; for (int j = 0; j < 3; ++j) { hg.gmin[j] = 1e30f; hg.gmax[j] = -1e30f; }
    for (int i = 0; i < M; ++i) for (int j = 0; j < 3; ++j) { hg.gmin[j] = std::min(hg.gmin[j], pts[3*i+j]); hg.gmax[j] = std::max(hg.gmax[j], pts[3*i+j]); }
    GridParams<T>* dgp; CK(hipMalloc(&dgp, sizeof hg)); CK(hipMemcpy(dgp, &hg, sizeof hg, hipMemcpyHostToDevice));
    KdBuild<T> b;
    size_t max_nodes = 2 * M + 2, max_level = M + 2, max_items = M / kKdChunk + max_level + 2;
    int* counters;
    CK(hipMalloc(&b.E, M * sizeof(Pt4<T>))); CK(hipMalloc(&b.nodes, max_nodes * sizeof(KdNode<T>))); CK(hipMalloc(&counters, 64));
    CK(hipMalloc(&b.level_nodes, max_level * 4)); CK(hipMalloc(&b.next_nodes, max_level * 4));
    CK(hipMalloc(&b.item_node, max_items * 4)); CK(hipMalloc(&b.item_chunk, max_items * 4));
    CK(hipMalloc(&b.chunk_bl, max_items * 4)); CK(hipMalloc(&b.chunk_br, max_items * 4));
    CK(hipMalloc(&b.BLpos, M * 4)); CK(hipMalloc(&b.BRpos, M * 4));
    b.n_nodes = counters; b.n_next = counters + 1; b.n_items = counters + 2; b.leaf_max = 10;
    CK(hipMemset(counters, 0, 64));
    hipLaunchKernelGGL(k_kd_init_elems<T>, dim3(1), dim3(kBlock), 0, 0, dpts, M, b.E);
    hipLaunchKernelGGL(k_kd_root<T>, dim3(1), dim3(64), 0, 0, b, dgp, M);
    int n_level = 1;
    for (int level = 0; n_level > 0; ++level) {
        int ub = M / kKdChunk + n_level + 1, nlb = (n_level + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_kd_plan<T>, dim3(1), dim3(kBlock), 0, 0, b, n_level);
        hipLaunchKernelGGL(k_kd_minmax<T>, dim3(ub), dim3(kBlock), 0, 0, b);
        hipLaunchKernelGGL(k_kd_choose<T>, dim3(nlb), dim3(kBlock), 0, 0, b, n_level);
        hipLaunchKernelGGL(k_kd_count<T>, dim3(ub), dim3(kBlock), 0, 0, b);
        for (int ph = 0; ph < 2; ++ph) {
            hipLaunchKernelGGL(k_kd_bad_count<T>, dim3(ub), dim3(kBlock), 0, 0, b, ph);
            hipLaunchKernelGGL(k_kd_chunk_scan<T>, dim3(nlb), dim3(kBlock), 0, 0, b, n_level, ph);
            hipLaunchKernelGGL(k_kd_lists<T>, dim3(ub), dim3(kBlock), 0, 0, b, ph);
            hipLaunchKernelGGL(k_kd_swap<T>, dim3(ub), dim3(kBlock), 0, 0, b, ph);
        }
        hipLaunchKernelGGL(k_kd_split<T>, dim3(nlb), dim3(kBlock), 0, 0, b, n_level);
        int n_next; CK(hipMemcpy(&n_next, b.n_next, 4, hipMemcpyDeviceToHost));
        std::swap(b.level_nodes, b.next_nodes); n_level = n_next;
        printf("level %d -> next %d\n", level, n_next);
    }
    int nn; CK(hipMemcpy(&nn, b.n_nodes, 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_dump, dim3(1), dim3(1), 0, 0, b.nodes, nn); CK(hipDeviceSynchronize());
    // search 4 queries = first 4 elements as Pt4
    int* qlist; int* qcount; CK(hipMalloc(&qlist, 16)); CK(hipMalloc(&qcount, 4));
    int hq[4] = {0, 1, 2, 3}, four = 4; CK(hipMemcpy(qlist, hq, 16, hipMemcpyHostToDevice)); CK(hipMemcpy(qcount, &four, 4, hipMemcpyHostToDevice));
    KdSearchArgs<T> a; a.E = b.E; a.nodes = b.nodes; a.qsorted = b.E; a.qlist = qlist; a.qcount_dev = qcount; a.k = 2; a.squared = 1;
    T* od; long long* oi; CK(hipMalloc(&od, M * 2 * sizeof(T))); CK(hipMalloc(&oi, M * 2 * 8));
    a.out_d = od; a.out_i = oi; CK(hipMalloc(&a.scratch_d, 4 * 2 * sizeof(T))); CK(hipMalloc(&a.scratch_i, 4 * 2 * 4)); a.error_flag = counters + 3;
    hipLaunchKernelGGL(k_kd_search<T>, dim3(1), dim3(64), 0, 0, a); CK(hipDeviceSynchronize());
    int err; CK(hipMemcpy(&err, counters + 3, 4, hipMemcpyDeviceToHost)); printf("search err %d\n", err);
    return 0;
}
