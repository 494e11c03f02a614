// development proof (GPU box): mi_sincosf (mi_kernels.h: glibc's sincosf restated in device doubles) against the HOST's sincosf for every
// float in [0, 2 pi] (all angles of the path tracer are 2 pi u) and, sampled, up to 120 -- bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fopenmp -DMI_HOST_SINCOS -Icorona-13_amd/csrc -Iinclude -Icorona-13_amd/host tools/micro/sincos_exact.hip -o tools/micro/sincos_exact && tools/micro/sincos_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "corona_mi.h"
#include "mi_device.h"
#include "mi_kernels.h"
__global__ void eval(uint32_t first, uint32_t n, uint32_t stride, float *s, float *c)
{
  const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
  if(i >= n) return;
  const float y = __uint_as_float(first + i*stride);
  mi_sincosf(y, s + i, c + i);
}
static long check(uint32_t first, uint32_t last, uint32_t stride, const char *what)
{
  const uint32_t chunk = 1u << 26;
  float *ds, *dc;
  hipMalloc(&ds, chunk*4); hipMalloc(&dc, chunk*4);
  std::vector<float> hs(chunk), hc(chunk);
  long bad = 0, tot = 0;
  for(uint64_t f = first; f <= last; f += (uint64_t)chunk*stride)
  {
    const uint32_t n = (uint32_t)((last - f)/stride + 1 < chunk ? (last - f)/stride + 1 : chunk);
    hipLaunchKernelGGL(eval, dim3((n + 255)/256), dim3(256), 0, 0, (uint32_t)f, n, stride, ds, dc);
    hipMemcpy(hs.data(), ds, n*4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), dc, n*4, hipMemcpyDeviceToHost);
#pragma omp parallel for reduction(+:bad)
    for(uint32_t i=0;i<n;i++)
    {
      uint32_t u = (uint32_t)f + i*stride; float y, s, c; memcpy(&y, &u, 4);
      sincosf(y, &s, &c);
      if(memcmp(&s, &hs[i], 4) || memcmp(&c, &hc[i], 4)) { if(bad < 5) printf("  y %a: host %a %a device %a %a\n", y, s, c, hs[i], hc[i]); bad++; }
    }
    tot += n;
  }
  printf("%s: %ld arguments, %ld differ from the host's sincosf\n", what, tot, bad);
  hipFree(ds); hipFree(dc);
  return bad;
}
int main()
{
  float twopi = 6.2831855f, lim = 119.99f; uint32_t a, b;
  memcpy(&a, &twopi, 4); memcpy(&b, &lim, 4);
  long bad = check(0, a, 1, "every float in [0, 2 pi]");
  check(0x80000000u, 0x80000000u + a, 1, "every float in [-2 pi, -0]");
  check(a, b, 7, "every 7th float in (2 pi, 120)");
  return bad != 0;
}
