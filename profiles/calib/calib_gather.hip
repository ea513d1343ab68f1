// profiles/calib/calib_gather.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of
// the search and index-build kernels (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your
// own access pattern"). Every kernel touches each 16-byte record of a 1 GiB array (4x the Infinity Cache) exactly once, so
// the HBM bytes are known: 1 GiB.
//   k_stream16   coalesced: lane i reads record i                                   (global_load_dwordx4, the calibrated case)
//   k_gather16   per-lane gather as k_search1_flat does it: within a 4 KB window a wave's lanes read records 64 B apart
//                (64 different 64-byte segments per load instruction, 16 B used of each); the block's four waves together
//                use every byte of the window, i.e. the other three quarters are L1/L2 hits shortly after
//   k_gather12   the same addresses, 12 of the 16 bytes (global_load_dwordx3: the candidate loads of the k = 1 kernel)
//   k_wstream16 / k_wscatter16   the same two patterns as 16-byte stores
// Build: hipcc --offload-arch=gfx950 -O3 -o profiles/calib/calib_gather profiles/calib/calib_gather.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_fetch -- profiles/calib/calib_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct alignas(16) Rec { float x, y, z; int idx; };
struct __attribute__((packed, aligned(4))) Rec3 { float x, y, z; };

__device__ __forceinline__ size_t gather_index(size_t t) {
    const size_t win = t >> 8;                       // 256 records = 4 KB per block
    const unsigned l = (unsigned)t & 63u, v = ((unsigned)t >> 6) & 3u;
    return (win << 8) + (size_t)(l * 4u + v);
}
__global__ __launch_bounds__(256) void k_stream16(const Rec* __restrict__ a, size_t n, float* out) {
    float s = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) { const Rec r = a[t]; s += r.x + r.y + r.z + (float)r.idx; }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_gather16(const Rec* __restrict__ a, size_t n, float* out) {
    float s = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) { const Rec r = a[gather_index(t)]; s += r.x + r.y + r.z + (float)r.idx; }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_gather12(const Rec* __restrict__ a, size_t n, float* out) {
    float s = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const Rec3 r = *reinterpret_cast<const Rec3*>(a + gather_index(t)); s += r.x + r.y + r.z;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_wstream16(Rec* __restrict__ a, size_t n) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) { Rec r; r.x = (float)t; r.y = 1.f; r.z = 2.f; r.idx = (int)t; a[t] = r; }
}
__global__ __launch_bounds__(256) void k_wscatter16(Rec* __restrict__ a, size_t n) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) { Rec r; r.x = (float)t; r.y = 1.f; r.z = 2.f; r.idx = (int)t; a[gather_index(t)] = r; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(Rec);
    Rec* a = nullptr; float* out = nullptr;
    CK(hipMalloc((void**)&a, bytes)); CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(a, 0, bytes));
    const int blocks = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(k_gather16, dim3(blocks), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(k_gather12, dim3(blocks), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(k_wstream16, dim3(blocks), dim3(256), 0, 0, a, n);
        hipLaunchKernelGGL(k_wscatter16, dim3(blocks), dim3(256), 0, 0, a, n);
    }
    CK(hipDeviceSynchronize());
    printf("calib_gather: %zu bytes per kernel, 5 kernels x 3 repetitions\n", bytes);
    return 0;
}
