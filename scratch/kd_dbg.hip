#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../point_cloud_utils_amd/csrc/kd_order.h"
using namespace pcu;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float T;
int main() {
    const int M = 50;
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
    printf("sizeof KdNode %zu KdBuild %zu\n", sizeof(KdNode<T>), sizeof(KdBuild<T>));
    hipLaunchKernelGGL(k_kd_init_elems<T>, dim3(1), dim3(kBlock), 0, 0, dpts, M, b.E); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_kd_root<T>, dim3(1), dim3(64), 0, 0, b, dgp, M); CK(hipDeviceSynchronize());
    int n_level = 1;
    hipLaunchKernelGGL(k_kd_plan<T>, dim3(1), dim3(kBlock), 0, 0, b, n_level); CK(hipDeviceSynchronize()); printf("plan ok\n");
    int hc[4]; CK(hipMemcpy(hc, counters, 16, hipMemcpyDeviceToHost)); printf("n_nodes %d n_next %d n_items %d\n", hc[0], hc[1], hc[2]);
    int in0, ic0; CK(hipMemcpy(&in0, b.item_node, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ic0, b.item_chunk, 4, hipMemcpyDeviceToHost)); printf("item0 node %d chunk %d\n", in0, ic0);
    hipLaunchKernelGGL(k_kd_minmax<T>, dim3(2), dim3(kBlock), 0, 0, b); CK(hipDeviceSynchronize()); printf("minmax ok\n");
    hipLaunchKernelGGL(k_kd_choose<T>, dim3(1), dim3(kBlock), 0, 0, b, n_level); CK(hipDeviceSynchronize()); printf("choose ok\n");
    KdNode<T> hn; CK(hipMemcpy(&hn, b.nodes, sizeof hn, hipMemcpyDeviceToHost));
    printf("root: [%d,%d) active %d divfeat %d cut %g nchunks %d base %d\n", hn.left, hn.right, hn.active, hn.divfeat, hn.cutval, hn.nchunks, hn.chunk_base);
    hipLaunchKernelGGL(k_kd_count<T>, dim3(2), dim3(kBlock), 0, 0, b); CK(hipDeviceSynchronize()); printf("count ok\n");
    CK(hipMemcpy(&hn, b.nodes, sizeof hn, hipMemcpyDeviceToHost)); printf("lt %d le %d\n", hn.lt, hn.le);
    return 0;
}
