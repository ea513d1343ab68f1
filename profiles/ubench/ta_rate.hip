// TA / vector-L1 throughput of per-lane gathers on gfx950 (measurement helper, not part of the product; results: profiles/r02_ubench.txt).
// Every wave issues ITER dependent-free loads from a 32 KB window (L1-resident after the first touch); patterns differ in bytes
// per lane and in how lanes share addresses. Reports cycles per wave-instruction per CU (blocks = 256 CUs x 8 waves x ...).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
template <int W, int MODE, int ACT = 64>
__global__ __launch_bounds__(256) void k(const char* __restrict__ base, int iters, float* out, long long* cyc) {
    const int lane = threadIdx.x & 63;
    if (ACT < 64 && (lane * 7 % 64) >= ACT) return;     // ACT scattered lanes stay
    unsigned off;
    // MODE 0: lanes 64 B apart (distinct lines); 1: consecutive records of W bytes; 2: groups of 4 lanes share an address, groups 64 B apart
    // 3: groups of 4 lanes share, groups consecutive records; 4: all lanes same address
    if (MODE == 0) off = lane * 64;
    else if (MODE == 1) off = lane * W;
    else if (MODE == 2) off = (lane >> 2) * 64;
    else if (MODE == 3) off = (lane >> 2) * W;
    else off = 0;
    off += (threadIdx.x >> 6) * 4096;
    float s = 0;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        const char* p = base + ((off + (unsigned)i * 192u) & 32767u);
        if (W == 16) { f4 v = *reinterpret_cast<const f4*>(p); s += v.x + v.y + v.z + v.w; }
        else if (W == 12) { F3 v = *reinterpret_cast<const F3*>(p); s += v.x + v.y + v.z; }
        else if (W == 8) { f2 v = *reinterpret_cast<const f2*>(p); s += v.x + v.y; }
        else { s += *reinterpret_cast<const float*>(p); }
    }
    const long long t1 = clock64();
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}
template <int W, int MODE, int ACT = 64> void run(const char* base, float* out, long long* cyc, const char* name) {
    const int iters = 4096, blocks = 256 * 8;
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, MODE, ACT>), dim3(blocks), dim3(256), 0, 0, base, 64, out, cyc);
    hipMemset(cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, MODE, ACT>), dim3(blocks), dim3(256), 0, 0, base, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per CU = blocks*4 waves*iters / 256 CUs
    const double wi_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%-28s W=%2d  %.3f ms  -> %.2f ns per wave-instr per CU (= %.1f cycles at 2.4 GHz), %.1f B/clk/CU useful\n", name, W, ms,
           ms * 1e6 / wi_per_cu, ms * 1e6 / wi_per_cu * 2.4, 64.0 * W / (ms * 1e6 / wi_per_cu * 2.4));
}
int main() {
    char* base; float* out; long long* cyc;
    hipMalloc((void**)&base, 65536); hipMemset(base, 0, 65536); hipMalloc((void**)&out, 64); hipMalloc((void**)&cyc, 8);
#define R(W, M, NAME) run<W, M>(base, out, cyc, NAME)
    R(16, 0, "distinct lines"); R(12, 0, "distinct lines"); R(8, 0, "distinct lines"); R(4, 0, "distinct lines");
    R(16, 1, "consecutive"); R(12, 1, "consecutive"); R(8, 1, "consecutive"); R(4, 1, "consecutive");
    R(16, 2, "4-lane shared, 64B apart"); R(12, 2, "4-lane shared, 64B apart");
    R(16, 3, "4-lane shared, consecutive"); R(12, 3, "4-lane shared, consecutive");
    R(16, 4, "all lanes same"); R(12, 4, "all lanes same"); R(4, 4, "all lanes same");
    run<12, 0, 32>(base, out, cyc, "distinct, 32 lanes"); run<12, 0, 16>(base, out, cyc, "distinct, 16 lanes"); run<12, 0, 8>(base, out, cyc, "distinct, 8 lanes"); run<12, 0, 2>(base, out, cyc, "distinct, 2 lanes");
    run<12, 2, 32>(base, out, cyc, "4-shared, 32 lanes"); run<12, 2, 16>(base, out, cyc, "4-shared, 16 lanes"); run<12, 2, 8>(base, out, cyc, "4-shared, 8 lanes");
    run<16, 0, 16>(base, out, cyc, "distinct, 16 lanes"); run<16, 0, 8>(base, out, cyc, "distinct, 8 lanes");
    return 0;
}
