// microbenchmark: does a wave64 VALU instruction cost less when whole 16-lane quarters of EXEC are off?
// build: hipcc --offload-arch=gfx950 -O3 exec_skip.hip -o exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float *out, int active, int stride, int iters)
{
  const int lane = threadIdx.x & 63;
  float a0 = threadIdx.x*1e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
  const float m = 0.999f, c = 1e-3f;
  if((lane % stride) == 0 && (lane / stride) < active)
  {
    for(int i=0;i<iters;i++)
    {
#pragma unroll 16
      for(int j=0;j<16;j++)
      {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
}
int main()
{
  float *d; hipMalloc(&d, 256*4*256*sizeof(float)*4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  struct { int active, stride; const char *what; } cfg[] = {
    {64, 1, "64 lanes"}, {32, 1, "lanes 0-31"}, {16, 1, "lanes 0-15"}, {1, 1, "lane 0"},
    {16, 4, "16 lanes, every 4th"}, {4, 16, "4 lanes, one per quarter"}, {32, 2, "32 lanes, every 2nd"} };
  for(auto &c : cfg)
    for(int wps = 1; wps <= 4; wps *= 2)
    {
      const int blocks = 256*wps;       // 4 waves per block -> wps waves per SIMD
      k<<<blocks, 256>>>(d, c.active, c.stride, 10);
      hipEventRecord(e0);
      k<<<blocks, 256>>>(d, c.active, c.stride, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_wave = (double)iters*16*8;
      const double cyc = ms*1e-3*2.4e9;
      printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f cycles per VALU instr per SIMD\n", c.what, wps, ms, cyc/(instr_per_wave*wps));
    }
  return 0;
}
