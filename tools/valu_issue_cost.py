"""Issue cost of vector instructions on gfx950, measured: what a SIMD pays per wavefront instruction when eight
wavefronts keep it busy.  Writes valu_ops.hip (one kernel per instruction form, each an unrolled stream over eight
independent registers), to be built and run on the GPU box:
    python3 tools/valu_issue_cost.py && hipcc -O2 --offload-arch=gfx950 -w -o valu_ops valu_ops.hip && ./valu_ops
The kernels of this library are bound by instruction issue; the table says what the floor per instruction is
(profiles/r3/06_valu_issue_cost.txt: 2.4 cycles for v_mov_b32 / v_mul_f32 / v_add_f32, 3.3 for 32-bit adds, right
shifts and two-operand logic, 4.3 for nearly everything else, packed and double precision included)."""
ops = [
 ("v_add_u32", 'v_add_u32 %0, %0, %1', 'i'),
 ("v_add_u32 (again)", 'v_add_u32 %0, %0, %1', 'i'),
 ("v_add_u32 imm", 'v_add_u32 %0, 3, %0', 'i'),
 ("v_sub_u32", 'v_sub_u32 %0, %0, %1', 'i'),
 ("v_mul_f32", 'v_mul_f32 %0, %0, %1', 'i'),
 ("v_add_f32", 'v_add_f32 %0, %0, %1', 'i'),
 ("v_lshlrev_b32(v)", 'v_lshlrev_b32 %0, %1, %0', 'i'),
 ("v_lshrrev_b32 imm", 'v_lshrrev_b32 %0, 3, %0', 'i'),
 ("v_ashrrev_i32(v)", 'v_ashrrev_i32 %0, %1, %0', 'i'),
 ("v_or_b32", 'v_or_b32 %0, %0, %1', 'i'),
 ("v_and_b32 imm", 'v_and_b32 %0, 63, %0', 'i'),
 ("v_max_u32", 'v_max_u32 %0, %0, %1', 'i'),
 ("v_cndmask_e32 (vcc set once)", 'v_cndmask_b32 %0, %0, %1, vcc', 'v'),
 ("v_fma_f32", 'v_fma_f32 %0, %0, %1, %0', 'i'),
 ("v_lshlrev_b32", 'v_lshlrev_b32 %0, 3, %0', 'i'),
 ("v_lshrrev_b32(v)", 'v_lshrrev_b32 %0, %1, %0', 'i'),
 ("v_and_b32", 'v_and_b32 %0, %0, %1', 'i'),
 ("v_or3_b32", 'v_or3_b32 %0, %0, %1, %1', 'i'),
 ("v_add3_u32", 'v_add3_u32 %0, %0, %1, %1', 'i'),
 ("v_and_or_b32", 'v_and_or_b32 %0, %0, %1, %1', 'i'),
 ("v_lshl_add_u32", 'v_lshl_add_u32 %0, %0, 2, %1', 'i'),
 ("v_lshl_or_b32", 'v_lshl_or_b32 %0, %0, 2, %1', 'i'),
 ("v_add_lshl_u32", 'v_add_lshl_u32 %0, %0, %1, 2', 'i'),
 ("v_bfe_u32(imm)", 'v_bfe_u32 %0, %0, 3, 5', 'i'),
 ("v_bfe_u32(v)", 'v_bfe_u32 %0, %0, %1, %1', 'i'),
 ("v_alignbit_b32", 'v_alignbit_b32 %0, %0, %1, %1', 'i'),
 ("v_alignbyte_b32", 'v_alignbyte_b32 %0, %0, %1, %1', 'i'),
 ("v_perm_b32", 'v_perm_b32 %0, %0, %1, %1', 'i'),
 ("v_med3_i32", 'v_med3_i32 %0, %0, %1, %1', 'i'),
 ("v_min_u32", 'v_min_u32 %0, %0, %1', 'i'),
 ("v_max3_u32", 'v_max3_u32 %0, %0, %1, %1', 'i'),
 ("v_cndmask_e32(vcc)", 'v_cndmask_b32 %0, %0, %1, vcc', 'i'),
 ("v_cndmask_e64(s)", 'v_cndmask_b32 %0, %0, %1, s[20:21]', 'i'),
 ("v_cmp_lt_u32 vcc", 'v_cmp_lt_u32 vcc, %0, %1', 'c'),
 ("v_cmp_lt_u32 sgpr", 'v_cmp_lt_u32 s[22:23], %0, %1', 'c'),
 ("v_cmp+v_cndmask", 'v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc', 'i2'),
 ("v_mul_lo_u32", 'v_mul_lo_u32 %0, %0, %1', 'i'),
 ("v_mul_hi_u32", 'v_mul_hi_u32 %0, %0, %1', 'i'),
 ("v_mul_u32_u24", 'v_mul_u32_u24 %0, %0, %1', 'i'),
 ("v_mad_u32_u24", 'v_mad_u32_u24 %0, %0, %1, %1', 'i'),
 ("v_mad_u64_u32", 'v_mad_u64_u32 %0, vcc, %2, %2, %0', 'l'),
 ("v_lshl_add_u64", 'v_lshl_add_u64 %0, %0, 1, %0', 'l'),
 ("v_add_u32_sdwa", 'v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD', 'i'),
 ("v_lshlrev_b32_sdwa", 'v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0', 'i'),
 ("v_pk_add_u16", 'v_pk_add_u16 %0, %0, %1', 'i'),
 ("v_sat_pk_u8_i16", 'v_sat_pk_u8_i16 %0, %0', 'i'),
 ("v_pk_add_f32", 'v_pk_add_f32 %0, %0, %0', 'l'),
 ("v_pk_fma_f32", 'v_pk_fma_f32 %0, %0, %0, %0', 'l'),
 ("v_mul_f64", 'v_mul_f64 %0, %0, %0', 'l'),
 ("v_cvt_f32_i32", 'v_cvt_f32_i32 %0, %0', 'i'),
 ("v_cvt_u32_f32", 'v_cvt_u32_f32 %0, %0', 'i'),
 ("v_cvt_pk_u8_f32", 'v_cvt_pk_u8_f32 %0, %0, 1, %1', 'i'),
 ("v_floor_f32", 'v_floor_f32 %0, %0', 'i'),
 ("v_rcp_f32", 'v_rcp_f32 %0, %0', 'i'),
 ("v_mov_b32", 'v_mov_b32 %0, %1', 'i'),
 ("v_mov_b64", 'v_mov_b64 %0, %0', 'l'),
 ("v_readfirstlane", 'v_readfirstlane_b32 s24, %0', 'c'),
 ("v_readlane", 'v_readlane_b32 s24, %0, 5', 'c'),
 ("v_bitop3_b32", 'v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96', 'i'),
 ("v_xor_b32", 'v_xor_b32 %0, %0, %1', 'i'),
 ("v_ffbh_u32", 'v_ffbh_u32 %0, %0', 'i'),
 ("v_bcnt_u32_b32", 'v_bcnt_u32_b32 %0, %0, %1', 'i'),
 ("v_mbcnt_lo", 'v_mbcnt_lo_u32_b32 %0, %0, %1', 'i'),
 ("v_pk_lshlrev_b16", 'v_pk_lshlrev_b16 %0, %1, %0', 'i'),
 ("v_cmp_eq_u32_sdwa", 'v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:DWORD', 'c'),
 ("dpp row_shr", 'v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf', 'i'),
 ("v_add_u32_dpp", 'v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf', 'i'),
 ("v_permlane32_swap", 'v_permlane32_swap_b32 %0, %0', 'i'),
]
src = ['''#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 3000
''']
for n,(name,asm,kind) in enumerate(ops):
    def line(i):
        a = asm.replace('\n', '\\n\\t')
        if kind in ('l','dl'):
            a = a.replace('%0','%[D]').replace('%2','%[S]').replace('%1','%[S]').replace('%3','%[L]')
        else:
            a = a.replace('%0','%[F]').replace('%1','%[S]').replace('%3','%[L]')
        return 'asm volatile("%s" : [F] "+v"(f[%d]), [D] "+v"(d[%d]) : [S] "v"(seed), [L] "v"(lds) : %s"s20","s21","s22","s23","s24","s25", "memory");' % (a, i, i, '' if kind == 'v' else '"vcc", ')
    src.append('__global__ __launch_bounds__(512) void k%d(unsigned *out, unsigned seed) {\n  __shared__ unsigned sh[2048]; sh[threadIdx.x] = seed; __syncthreads();\n  unsigned f[8]; unsigned long long d[8]; const unsigned lds = (threadIdx.x * 4u) & 8191u;\n  for (int i = 0; i < 8; i++) { f[i] = seed + i + threadIdx.x; d[i] = f[i] * 0x100000001ull; }\n  asm volatile("s_mov_b64 s[20:21], 0x5555\\n s_mov_b32 s25, 0" ::: "s20","s21","s25");\n  for (int it = 0; it < N_ITER; it++) {\n' % n)
    for i in range(8): src.append('    ' + line(i) + '\n')
    src.append('  }\n  asm volatile("s_waitcnt lgkmcnt(0)");\n  unsigned s = 0; for (int i = 0; i < 8; i++) s += f[i] + (unsigned) d[i];\n  if (s == 0x12345678u) out[0] = s;\n}\n')
src.append('typedef void (*K)(unsigned *, unsigned);\nstruct E { const char *name; K k; int per; };\nstatic E es[] = {\n')
for n,(name,asm,kind) in enumerate(ops):
    src.append('  {"%s", k%d, %d},\n' % (name, n, 2 if kind == 'i2' else 1))
src.append('''};
int main() {
  unsigned *out; (void) hipMalloc(&out, 4);
  hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
  const int blocks = 256 * 4;
  double base = 0;
  for (int w = 0; w < 60; w++) es[0].k<<<blocks, 512>>>(out, 3);
  (void) hipDeviceSynchronize();
  for (auto &e : es) {
    e.k<<<blocks, 512>>>(out, 3);
    (void) hipEventRecord(a);
    for (int r = 0; r < 3; r++) e.k<<<blocks, 512>>>(out, 3);
    (void) hipEventRecord(b); (void) hipEventSynchronize(b);
    float ms; (void) hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double inst = 8.0 * N_ITER * 8 * e.per;
    const double ns = ms * 1e6 / inst;
    if (base == 0) base = ns;
    printf("%-22s %7.3f ns per wavefront instruction per SIMD   x%.2f of v_add_u32\\n", e.name, ns, ns / base);
  }
  return 0;
}
''')
open('valu_ops.hip','w').write(''.join(src))
