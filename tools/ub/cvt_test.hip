// What v_cvt_pk_u8_f32 does with fractions, and an add under round-toward-minus-infinity.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, float *outf, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
    float a = in[i], r;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32 %0, 0x43000000, %1\n\t"
                 "s_nop 1\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(r) : "v"(a));
    outf[i] = r;
    unsigned c;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
                 "s_nop 1\n\t"
                 "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
                 "s_nop 1\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(c) : "v"(r));
    out[64 + i] = c;
}
int main() {
    float h[16] = {0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 254.5f, 254.9f, 255.5f, 256.0f, -0.5f, -1e-8f, 127.99999f, -3.7f, 1e-8f, 99.999f, -127.99999f};
    float *d; unsigned *o; float *of;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 128 * 4); hipMalloc(&of, 64 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, of, 16);
    unsigned ho[128]; float hf[64];
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hf, of, sizeof hf, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf("%14.9g -> cvt %3u | +128 rounded down = %.9g -> cvt %3u\n", h[i], ho[i], hf[i], ho[64 + i]);
    return 0;
}
