// development microbenchmark / proof (GPU box): is  rcp + two Newton steps + v_div_fixup  the correctly rounded 1/x for EVERY float?
// Compares against the compiler's IEEE division 1.0f/x over all 2^32 bit patterns; likewise a/b for a set of numerators.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/rcp_exact.hip -o tools/micro/rcp_exact && tools/micro/rcp_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
__device__ __forceinline__ float rcp_fast(float b)
{
  float y = __builtin_amdgcn_rcpf(b);
  float e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(y, e, y);
  e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(y, e, y);
  return __builtin_amdgcn_div_fixupf(y, b, 1.0f);
}
__device__ __forceinline__ float div_fast(float a, float b)
{
  float y = __builtin_amdgcn_rcpf(b);
  float e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(y, e, y);
  float q = a*y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  return __builtin_amdgcn_div_fixupf(q, b, a);
}
__device__ __forceinline__ float sqrt_fast(float x)
{ /* v_sqrt_f32 is within 1 ulp: pick among s-1ulp, s, s+1ulp by the sign of the exact residuals (the middle of the compiler's
     correctly rounded expansion, without its scaling of denormal inputs) */
  const float s = __builtin_amdgcn_sqrtf(x);
  const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
  const float rlo = __builtin_fmaf(-lo, s, x), rhi = __builtin_fmaf(-hi, s, x);
  float r = rlo <= 0.0f ? lo : s;
  r = rhi > 0.0f ? hi : r;
  return r;
}
__global__ void check_sqrt(unsigned long long *bad, uint32_t *first)
{
  for(unsigned long long i = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x*blockDim.x)
  {
    const float x = __uint_as_float((uint32_t)i);
    const float a = sqrt_fast(x), b = sqrtf(x);
    const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
    const uint32_t ex = ((uint32_t)i >> 23) & 255u;
    if(!same) { if(atomicAdd(bad, 1ull) == 0) *first = (uint32_t)i; if(ex >= 1 || (uint32_t)i == 0 || (uint32_t)i == 0x80000000u) atomicAdd(bad + 1, 1ull);   /* bad[1]: not a denormal */
                if(!((uint32_t)i >> 31)) { atomicMin(first + 2, ex); atomicMax(first + 3, ex); } }
  }
}
__global__ void check_rcp(unsigned long long *bad, uint32_t *first)
{
  for(unsigned long long i = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x*blockDim.x)
  {
    const float x = __uint_as_float((uint32_t)i);
    const float a = rcp_fast(x), b = 1.0f/x;
    const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
    const uint32_t ex = ((uint32_t)i >> 23) & 255u;
    if(!same) { if(atomicAdd(bad, 1ull) == 0) *first = (uint32_t)i; if(ex >= 1 && ex <= 252) atomicAdd(bad + 1, 1ull); }   /* bad[1]: normal x below 2^126 */
  }
}
__global__ void check_div(unsigned long long *bad, uint32_t *first, uint32_t numer_bits)
{
  const float n = __uint_as_float(numer_bits);
  for(unsigned long long i = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x*blockDim.x)
  {
    const float x = __uint_as_float((uint32_t)i);
    const float a = div_fast(n, x), b = n/x;
    const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
    /* safe domain: x normal, the exact quotient's exponent well inside the normal range */
    const int ex = (int)(((uint32_t)i >> 23) & 255u), en = (int)((numer_bits >> 23) & 255u);
    const bool safe = ex >= 1 && ex <= 254 && en >= 1 && en <= 254 && en - ex >= -125 && en - ex <= 126;
    if(!same) { atomicAdd(bad, 1ull); if(safe && atomicAdd(bad + 1, 1ull) == 0) first[1] = (uint32_t)i; }
  }
}
int main()
{
  unsigned long long *bad, h[2] = {0, 0}; uint32_t *first, hf[2] = {0, 0};
  hipMalloc(&bad, 16); hipMalloc(&first, 16); hipMemset(bad, 0, 16); hipMemset(first, 0, 16);
  hipLaunchKernelGGL(check_rcp, dim3(4096), dim3(256), 0, 0, bad, first);
  hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
  printf("rcp: %llu of 2^32 inputs differ from 1.0f/x (first bit pattern %08x); %llu of them for normal |x| < 2^126\n", h[0], hf[0], h[1]);
  hipMemset(bad, 0, 16);
  { const uint32_t init[4] = {0, 0, 255, 0}; hipMemcpy(first, init, 16, hipMemcpyHostToDevice); }
  hipLaunchKernelGGL(check_sqrt, dim3(4096), dim3(256), 0, 0, bad, first);
  hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
  { uint32_t ex[4]; const uint32_t init[4] = {0, 0, 255, 0}; hipMemcpy(ex, first, 16, hipMemcpyDeviceToHost); printf("sqrt: failing positive inputs have exponent fields %u .. %u\n", ex[2], ex[3]); (void)init; }
  printf("sqrt: %llu of 2^32 inputs differ from sqrtf(x) (first bit pattern %08x); %llu of them for inputs that are not denormal\n", h[0], hf[0], h[1]);
  const float numer[] = { 1.0f, 3.0f, 0.1f, 1e-20f, 1e20f, 7.123456e-3f, -2.5f, 3.4e38f, 1.2e-38f, 1e-40f, 0.0f, 1.9999999f };
  for(float n : numer)
  {
    hipMemset(bad, 0, 16);
    uint32_t bits; memcpy(&bits, &n, 4);
    hipLaunchKernelGGL(check_div, dim3(4096), dim3(256), 0, 0, bad, first, bits);
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
    printf("div: numerator %g: %llu of 2^32 denominators differ from a/b, %llu inside the safe domain (first %08x)\n", n, h[0], h[1], hf[1]);
  }
  return 0;
}
