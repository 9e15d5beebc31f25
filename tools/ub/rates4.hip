#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k0(float *out, int iters) {
    unsigned u = threadIdx.x, v = 0, w = 5, d0, d1, d2, d3, d4, d5, d6, d7; unsigned long long m2 = 0;
    d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = 0;
    for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\t" "v_cndmask_b32_e64 %0, %1, %2, vcc\n\tv_cndmask_b32_e64 %0, %2, %1, vcc\n\tv_cndmask_b32_e64 %0, %1, %2, vcc\n\tv_cndmask_b32_e64 %0, %2, %1, vcc\n\tv_cndmask_b32_e64 %0, %1, %2, vcc\n\tv_cndmask_b32_e64 %0, %2, %1, vcc\n\tv_cndmask_b32_e64 %0, %1, %2, vcc\n\tv_cndmask_b32_e64 %0, %2, %1, vcc" : "=v"(v) : "v"(u), "v"(w) : "vcc");)) }
    asm volatile("" :: "v"(v), "s"(m2), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
}
__global__ __launch_bounds__(512) void k1(float *out, int iters) {
    unsigned u = threadIdx.x, v = 0, w = 5, d0, d1, d2, d3, d4, d5, d6, d7; unsigned long long m2 = 0;
    d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = 0;
    for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile("v_cmp_lt_u32 vcc, %8, %9\n\t" "v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %9, %8, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %9, %8, vcc\n\tv_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %9, %8, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %9, %8, vcc" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(u), "v"(w) : "vcc");)) }
    asm volatile("" :: "v"(v), "s"(m2), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
}
__global__ __launch_bounds__(512) void k2(float *out, int iters) {
    unsigned u = threadIdx.x, v = 0, w = 5, d0, d1, d2, d3, d4, d5, d6, d7; unsigned long long m2 = 0;
    d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = 0;
    for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\t" "v_cndmask_b32 %0, %1, %2, vcc\n\tv_and_b32 %0, %2, %1\n\tv_and_b32 %0, %1, %2\n\tv_and_b32 %0, %2, %1\n\tv_and_b32 %0, %1, %2\n\tv_and_b32 %0, %2, %1\n\tv_and_b32 %0, %1, %2\n\tv_and_b32 %0, %2, %1" : "=v"(v) : "v"(u), "v"(w) : "vcc");)) }
    asm volatile("" :: "v"(v), "s"(m2), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
}
__global__ __launch_bounds__(512) void k3(float *out, int iters) {
    unsigned u = threadIdx.x, v = 0, w = 5, d0, d1, d2, d3, d4, d5, d6, d7; unsigned long long m2 = 0;
    d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = 0;
    for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %2, %1\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %2, %1\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %2, %1\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cmp_lt_u32 vcc, %2, %1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(v) : "v"(u), "v"(w) : "vcc");)) }
    asm volatile("" :: "v"(v), "s"(m2), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
}
__global__ __launch_bounds__(512) void k4(float *out, int iters) {
    unsigned u = threadIdx.x, v = 0, w = 5, d0, d1, d2, d3, d4, d5, d6, d7; unsigned long long m2 = 0;
    d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = 0;
    for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc\n\tv_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(v) : "v"(u), "v"(w) : "vcc");)) }
    asm volatile("" :: "v"(v), "s"(m2), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
}
int main() {
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 2000;
    const char *names[] = {"cmp + 8 cndmask_e64 vcc", "cmp + 8 cndmask_e32 vcc, 8 dsts", "cmp + 1 cndmask_e32 + 7 v_and", "8 x (cmp + cndmask_e32)", "v_addc_co_u32 vcc x8"};
    void (*ks[])(float *, int) = {k0, k1, k2, k3, k4};
    for (int i = 0; i < 5; i++) {
        hipLaunchKernelGGL(ks[i], dim3(cus * 4), dim3(512), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[i], dim3(cus * 4), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = 8.0 * iters * 64;
        printf("%-22s %7.3f ms  %5.2f cycles per asm statement at 2.4 GHz\n", names[i], ms, ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
