// microbenchmark: issue cost (cycles per wave64 instruction per SIMD at 4 waves/SIMD) of the VALU instructions the
// traversal loop is made of. build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(float *out, int iters) \
{ \
  float a0 = threadIdx.x*1e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f; \
  float b0 = 0.999f, b1 = 1e-3f; \
  for(int i=0;i<iters;i++) \
  { \
    REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) \
  } \
  out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
}
// each ASM string holds 8 instructions (one per accumulator)
BODY(k_fma,   "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
BODY(k_mul,   "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
BODY(k_max3,  "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
BODY(k_cndm,  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
BODY(k_cmpsel, "v_cmp_lt_f32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n s_nop 1\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n s_nop 1\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n s_nop 1\n v_cndmask_b32 %3, %3, %8, vcc\n")
__global__ void __launch_bounds__(256) k_pkmul(float *out, int iters)
{
  double a0 = threadIdx.x*1e-3, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, a4 = a0 + 4.0, a5 = a0 + 5.0, a6 = a0 + 6.0, a7 = a0 + 7.0, b0 = 0.999;
  for(int i=0;i<iters;i++)
  {
    REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
  }
  out[blockIdx.x*blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
BODY(k_rcp,   "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
BODY(k_mov,   "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
BODY(k_addu,  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
BODY(k_salu,  "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
BODY(k_mix,   "v_mul_f32 %0, %0, %8\n s_nop 0\n v_mul_f32 %1, %1, %8\n s_nop 0\n v_mul_f32 %2, %2, %8\n s_nop 0\n v_mul_f32 %3, %3, %8\n s_nop 0\n")
BODY(k_cnd_e64, "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n")
BODY(k_cnd_ab, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
BODY(k_cmp,   "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
BODY(k_cmp64, "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n")
BODY(k_cmp4sel4, "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[22:23]\n v_cndmask_b32 %2, %2, %8, s[24:25]\n v_cndmask_b32 %3, %3, %8, s[26:27]\n")
BODY(k_min,   "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n")
BODY(k_med3,  "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n")
BODY(k_fma3,  "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n")
BODY(k_sqrt,  "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
BODY(k_bfi, "v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n")
BODY(k_bfeu, "v_bfe_u32 %0, %0, %8, 1\n v_bfe_u32 %1, %1, %8, 1\n v_bfe_u32 %2, %2, %8, 1\n v_bfe_u32 %3, %3, %8, 1\n v_bfe_u32 %4, %4, %8, 1\n v_bfe_u32 %5, %5, %8, 1\n v_bfe_u32 %6, %6, %8, 1\n v_bfe_u32 %7, %7, %8, 1\n")
BODY(k_bfei, "v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1\n v_bfe_i32 %4, %4, 3, 1\n v_bfe_i32 %5, %5, 3, 1\n v_bfe_i32 %6, %6, 3, 1\n v_bfe_i32 %7, %7, 3, 1\n")
BODY(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
BODY(k_lshr, "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7\n")
BODY(k_or3, "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %8, %9\n v_or3_b32 %5, %5, %8, %9\n v_or3_b32 %6, %6, %8, %9\n v_or3_b32 %7, %7, %8, %9\n")
BODY(k_andor, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n")
BODY(k_lshlor, "v_lshl_or_b32 %0, %0, 2, %9\n v_lshl_or_b32 %1, %1, 2, %9\n v_lshl_or_b32 %2, %2, 2, %9\n v_lshl_or_b32 %3, %3, 2, %9\n v_lshl_or_b32 %4, %4, 2, %9\n v_lshl_or_b32 %5, %5, 2, %9\n v_lshl_or_b32 %6, %6, 2, %9\n v_lshl_or_b32 %7, %7, 2, %9\n")
BODY(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
BODY(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n")
BODY(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
BODY(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
BODY(k_mad24, "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n")
BODY(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %9\n v_lshl_add_u32 %1, %1, 2, %9\n v_lshl_add_u32 %2, %2, 2, %9\n v_lshl_add_u32 %3, %3, 2, %9\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n")
BODY(k_bcnt, "v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n")
BODY(k_cvt, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n")
BODY(k_maxu, "v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8\n")
BODY(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
BODY(k_mule64, "v_mul_f32_e64 %0, %0, %8\n v_mul_f32_e64 %1, %1, %8\n v_mul_f32_e64 %2, %2, %8\n v_mul_f32_e64 %3, %3, %8\n v_mul_f32_e64 %4, %4, %8\n v_mul_f32_e64 %5, %5, %8\n v_mul_f32_e64 %6, %6, %8\n v_mul_f32_e64 %7, %7, %8\n")
BODY(k_cmpclass, "v_cmp_class_f32 vcc, %0, %8\n v_cmp_class_f32 vcc, %1, %8\n v_cmp_class_f32 vcc, %2, %8\n v_cmp_class_f32 vcc, %3, %8\n v_cmp_class_f32 vcc, %4, %8\n v_cmp_class_f32 vcc, %5, %8\n v_cmp_class_f32 vcc, %6, %8\n v_cmp_class_f32 vcc, %7, %8\n")
BODY(k_cmpu, "v_cmp_eq_u32 vcc, %0, %8\n v_cmp_eq_u32 vcc, %1, %8\n v_cmp_eq_u32 vcc, %2, %8\n v_cmp_eq_u32 vcc, %3, %8\n v_cmp_eq_u32 vcc, %4, %8\n v_cmp_eq_u32 vcc, %5, %8\n v_cmp_eq_u32 vcc, %6, %8\n v_cmp_eq_u32 vcc, %7, %8\n")
BODY(k_movdpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")


// double precision (the glibc-exact sincos of mi_kernels.h is made of these)
#define BODY64(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(float *out, int iters) \
{ \
  double a0 = threadIdx.x*1e-3, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, a4 = a0 + 4.0, a5 = a0 + 5.0, a6 = a0 + 6.0, a7 = a0 + 7.0, b0 = 0.999, b1 = 1e-3; \
  for(int i=0;i<iters;i++) \
  { \
    REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) \
  } \
  out[blockIdx.x*blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
}
BODY64(k_mul64, "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n")
BODY64(k_add64, "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
BODY64(k_fma64, "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")


// round 6: the integer instructions of address arithmetic, of the generator (64-bit shifts) and of IEEE division
#define BODYI64(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(float *out, int iters) \
{ \
  unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  unsigned int b0 = 176u + (threadIdx.x & 1u), b1 = 12345u + threadIdx.x; \
  for(int i=0;i<iters;i++) \
  { \
    REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc");) \
  } \
  out[blockIdx.x*blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
}
BODYI64(k_mad64, "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n ")
BODYI64(k_lshl64, "v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3\n v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7\n ")
BODYI64(k_lshr64, "v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3\n v_lshrrev_b64 %4, 3, %4\n v_lshrrev_b64 %5, 3, %5\n v_lshrrev_b64 %6, 3, %6\n v_lshrrev_b64 %7, 3, %7\n ")
BODYI64(k_lshladd64, "v_lshl_add_u64 %0, %0, 2, %0\n v_lshl_add_u64 %1, %1, 2, %1\n v_lshl_add_u64 %2, %2, 2, %2\n v_lshl_add_u64 %3, %3, 2, %3\n v_lshl_add_u64 %4, %4, 2, %4\n v_lshl_add_u64 %5, %5, 2, %5\n v_lshl_add_u64 %6, %6, 2, %6\n v_lshl_add_u64 %7, %7, 2, %7\n ")
BODY(k_mullo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n ")
BODY(k_mulhi, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n ")
BODY(k_mul24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n ")
BODY(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n ")
BODY(k_divscale, "v_div_scale_f32 %0, vcc, %0, %8, %9\n v_div_scale_f32 %1, vcc, %1, %8, %9\n v_div_scale_f32 %2, vcc, %2, %8, %9\n v_div_scale_f32 %3, vcc, %3, %8, %9\n v_div_scale_f32 %4, vcc, %4, %8, %9\n v_div_scale_f32 %5, vcc, %5, %8, %9\n v_div_scale_f32 %6, vcc, %6, %8, %9\n v_div_scale_f32 %7, vcc, %7, %8, %9\n ")
BODY(k_divfmas, "v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9\n ")
BODY(k_divfixup, "v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n ")

// round 6: does it matter whether the destination is one of the sources? No: add 2.4 either way, min 4.4.
// An oddity on the way: RUNS of v_cndmask_b32 that read VCC back to back measure 15-23 cycles each -- behind a never-written VCC (k_cndm, k_cnd_ab, k_salu_vcc_sel) and
// behind one v_cmp + s_nop too (k_cmp1sel8) --, with an SGPR pair as mask 4.4 (k_cnd_e64), and with other vector instructions between compare and selects (k_cmp1x8sel8)
// 2.7 on average over the nine. The path kernels issue a vector instruction every 3.65 cycles, which is what the class costs of their mix add up to
// (0.42 x 2.5 + 0.58 x 4.4): no such runs in them.
BODY(k_add_ab, "v_add_f32 %0, %8, %9\n v_add_f32 %1, %8, %9\n v_add_f32 %2, %8, %9\n v_add_f32 %3, %8, %9\n v_add_f32 %4, %8, %9\n v_add_f32 %5, %8, %9\n v_add_f32 %6, %8, %9\n v_add_f32 %7, %8, %9\n ")
BODY(k_add_rot, "v_add_f32 %0, %1, %2\n v_add_f32 %1, %2, %3\n v_add_f32 %2, %3, %4\n v_add_f32 %3, %4, %5\n v_add_f32 %4, %5, %6\n v_add_f32 %5, %6, %7\n v_add_f32 %6, %7, %0\n v_add_f32 %7, %0, %1\n ")
BODY(k_cnd_rot, "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n v_cndmask_b32 %6, %7, %0, vcc\n v_cndmask_b32 %7, %0, %1, vcc\n ")
BODY(k_cnd_b0, "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n ")
BODY(k_cnd_ab2, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n ")
BODY(k_min_ab, "v_min_f32 %0, %8, %9\n v_min_f32 %1, %8, %9\n v_min_f32 %2, %8, %9\n v_min_f32 %3, %8, %9\n v_min_f32 %4, %8, %9\n v_min_f32 %5, %8, %9\n v_min_f32 %6, %8, %9\n v_min_f32 %7, %8, %9\n ")
BODY(k_mov_ab, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n ")

BODY(k_cmp1sel8, "v_cmp_lt_f32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
BODY(k_cmp1x8sel8, "v_cmp_lt_f32 vcc, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n")
BODY(k_salu_vcc_sel, "s_mov_b64 vcc, s[20:21]\n s_nop 4\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")

template<class K> static void run(const char *name, K kern, float *d, int per_asm)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for(int wps = 1; wps <= 4; wps *= 4)
  {
    const int blocks = 256*wps;
    kern<<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters*8*per_asm;
    printf("%-10s waves/SIMD %d: %.3f ms  %.2f cycles@2.4GHz per instruction per SIMD\n", name, wps, ms, ms*1e-3*2.4e9/(n*wps));
  }
}
int main()
{
  float *d; if(hipMalloc(&d, 256*4*256*sizeof(float)) != hipSuccess) return 1;
  run("fma", k_fma, d, 8); run("mul", k_mul, d, 8); run("max3", k_max3, d, 8); run("cndmask", k_cndm, d, 8);
  run("cmp+nop+sel", k_cmpsel, d, 4); run("pk_mul", k_pkmul, d, 8); run("rcp", k_rcp, d, 8); run("mov", k_mov, d, 8);
  run("add_u32", k_addu, d, 8); run("s_nop", k_salu, d, 8); run("mul+s_nop", k_mix, d, 4);
  run("cnd_e64", k_cnd_e64, d, 8); run("cnd_ab", k_cnd_ab, d, 8); run("cmp vcc", k_cmp, d, 8); run("cmp sgpr", k_cmp64, d, 8);
  run("4cmp+4sel", k_cmp4sel4, d, 8); run("min", k_min, d, 8); run("med3", k_med3, d, 8); run("fma 3 vgpr", k_fma3, d, 8); run("sqrt", k_sqrt, d, 8);
  run("bfi", k_bfi, d, 8); run("bfeu", k_bfeu, d, 8); run("bfei", k_bfei, d, 8); run("and", k_and, d, 8); run("lshr", k_lshr, d, 8); run("or3", k_or3, d, 8); run("andor", k_andor, d, 8); run("lshlor", k_lshlor, d, 8); run("perm", k_perm, d, 8); run("add3", k_add3, d, 8); run("sub", k_sub, d, 8); run("xor", k_xor, d, 8); run("mad24", k_mad24, d, 8); run("lshladd", k_lshladd, d, 8); run("bcnt", k_bcnt, d, 8); run("cvt", k_cvt, d, 8); run("maxu", k_maxu, d, 8); run("fmac", k_fmac, d, 8); run("mule64", k_mule64, d, 8); run("cmpclass", k_cmpclass, d, 8); run("cmpu", k_cmpu, d, 8); run("movdpp", k_movdpp, d, 8);
  run("mul_f64", k_mul64, d, 8); run("add_f64", k_add64, d, 8); run("fma_f64", k_fma64, d, 8);
  run("mad_u64_u32", k_mad64, d, 8); run("lshl_b64", k_lshl64, d, 8); run("lshr_b64", k_lshr64, d, 8); run("lshl_add_u64", k_lshladd64, d, 8);
  run("mul_lo_u32", k_mullo, d, 8); run("mul_hi_u32", k_mulhi, d, 8); run("mul_u32_u24", k_mul24, d, 8); run("mul_hi_u24", k_mulhi24, d, 8);
  run("div_scale", k_divscale, d, 8); run("div_fmas", k_divfmas, d, 8); run("div_fixup", k_divfixup, d, 8);
  run("add a=b+c", k_add_ab, d, 8); run("add rot", k_add_rot, d, 8); run("cnd rot", k_cnd_rot, d, 8); run("cnd a=b0:a", k_cnd_b0, d, 8); run("cnd a=b0:b1", k_cnd_ab2, d, 8); run("min a=b0,b1", k_min_ab, d, 8);
  run("cmp+8sel", k_cmp1sel8, d, 9); run("cmp4add4sel", k_cmp1x8sel8, d, 9); run("smov+8sel", k_salu_vcc_sel, d, 8);
  return 0;
}
