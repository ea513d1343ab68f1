// VALU issue rate on gfx950 (measurement helper; results: profiles/r02_ubench.txt): shader-clock cycles per wave64 instruction per SIMD, by instruction kind.
// Inline asm (the compiler would SLP-pack or fold C code); 8 waves per SIMD; 16 independent destination registers per kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w[16], w2 = {1.0001f, 0.9999f};
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = f2{v[i], v[i] + 1.f};
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc");
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define I_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
#define I_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
#define I_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
#define I_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(b) : "vcc");
#define I_IADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[i]) : "v"(b));
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(b));
#define I_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(b));
#define I_CNDPAIR(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[i]) : "v"(b), "v"(c) : "vcc");
#define I_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[i]) : "v"(w2));
#define I_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(w2));
#define I_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_MINI(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define I_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
#define I_CMPE64(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(v[i]), "v"(b) : "s20", "s21");
        if (KIND == 0) { REP16(I_FMA) REP16(I_FMA) } else if (KIND == 1) { REP16(I_ADD) REP16(I_ADD) } else if (KIND == 2) { REP16(I_MUL) REP16(I_MUL) }
        else if (KIND == 3) { REP16(I_MIN) REP16(I_MIN) } else if (KIND == 4) { REP16(I_MIN3) REP16(I_MIN3) } else if (KIND == 5) { REP16(I_CND) REP16(I_CND) }
        else if (KIND == 6) { REP16(I_CMP) REP16(I_CMP) } else if (KIND == 7) { REP16(I_IADD) REP16(I_IADD) } else if (KIND == 8) { REP16(I_AND) REP16(I_AND) }
        else if (KIND == 9) { REP16(I_LSHLADD) REP16(I_LSHLADD) } else if (KIND == 10) { REP16(I_MOV) REP16(I_MOV) } else if (KIND == 11) { REP16(I_SUB) REP16(I_SUB) }
        else if (KIND == 12) { REP16(I_CMPE64) REP16(I_CMPE64) }
        else if (KIND == 13) { REP16(I_CND64) REP16(I_CND64) } else if (KIND == 14) { REP16(I_CNDPAIR) } else if (KIND == 15) { REP16(I_PKMUL) REP16(I_PKMUL) }
        else if (KIND == 16) { REP16(I_PKADD) REP16(I_PKADD) } else if (KIND == 17) { REP16(I_MAX) REP16(I_MAX) } else if (KIND == 18) { REP16(I_MINI) REP16(I_MINI) } else if (KIND == 19) { REP16(I_OR3) REP16(I_OR3) }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + w[i].x + w[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}
template <int KIND> void run(float* out, unsigned long long* cyc, const char* name) {
    const int iters = 4000, blocks = 2048;           // 8 blocks per CU = 8 waves per SIMD, all resident at once
    (void)hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_block_cycles = (double)h / blocks;           // shader-clock cycles one block (= 1 wave per SIMD) was in the loop
    const double instr_per_simd_meanwhile = 8.0 * iters * 32;     // 8 waves per SIMD issue concurrently
    printf("%-16s %.3f ms  clock64: %.2f ticks per wave-instruction per SIMD;  wall: %.3f ns\n", name, ms, per_block_cycles / instr_per_simd_meanwhile,
           ms * 1e6 / instr_per_simd_meanwhile);
}
int main() {
    float* out; unsigned long long* cyc; (void)hipMalloc((void**)&out, 2048 * 256 * 4); (void)hipMalloc((void**)&cyc, 8);
    run<0>(out, cyc, "warmup");
    run<0>(out, cyc, "v_fma_f32"); run<1>(out, cyc, "v_add_f32"); run<11>(out, cyc, "v_sub_f32"); run<2>(out, cyc, "v_mul_f32"); run<3>(out, cyc, "v_min_f32"); run<4>(out, cyc, "v_min3_f32");
    run<5>(out, cyc, "v_cndmask_b32"); run<6>(out, cyc, "v_cmp_lt_f32 vcc"); run<12>(out, cyc, "v_cmp_lt_f32 sgpr"); run<7>(out, cyc, "v_add_u32"); run<8>(out, cyc, "v_and_b32");
    run<9>(out, cyc, "v_lshl_add_u32"); run<10>(out, cyc, "v_mov_b32");
    run<13>(out, cyc, "v_cndmask e64 sgpr"); run<14>(out, cyc, "cmp+cndmask pair"); run<15>(out, cyc, "v_pk_mul_f32"); run<16>(out, cyc, "v_pk_add_f32"); run<17>(out, cyc, "v_max_f32"); run<18>(out, cyc, "v_min_u32"); run<19>(out, cyc, "v_or3_b32");
    return 0;
}
