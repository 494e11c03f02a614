/* ch_rgb2spec_lut.c -- reader of the reference's RGB -> spectrum coefficient table ("SPEC", u32 res, float scale[res],
 * float data[3*res^3*3], include/rgb2spec.h:28-64) and its trilinear fetch (rgb2spec_fetch, include/rgb2spec.h:87-128,
 * rgb2spec_find_interval 66-84).
 *
 * Compiled under the reference build's floating-point contract (Makefile: LUT_CFLAGS = arch.example's -O3 -ffast-math with
 * FMA), like the QBVH builder: the interpolation is a chain of a*x0 + b*x1 terms that gcc contracts into FMAs, and only
 * the same contraction gives the reference's coefficients to the last bit (tests/test_host.py pins all colours of the
 * test scenes against coefficients dumped from the real reference).
 */
#include "ch_internal.h"
#include <stdlib.h>
#include <string.h>

typedef struct lut_t { uint32_t res; float *scale; float *data; } lut_t;

static int lut_load(const char *fn, lut_t *l)
{
  FILE *f = fopen(fn, "rb");
  if(!f) return 1;
  char magic[4];
  if(fread(magic, 4, 1, f) != 1 || memcmp(magic, "SPEC", 4) || fread(&l->res, 4, 1, f) != 1 || l->res < 2 || l->res > 1024)
  { fclose(f); return 1; }
  const size_t ns = l->res, nd = (size_t)l->res*l->res*l->res*9;
  l->scale = (float *)malloc(ns*sizeof(float));
  l->data  = (float *)malloc(nd*sizeof(float));
  if(!l->scale || !l->data || fread(l->scale, sizeof(float), ns, f) != ns || fread(l->data, sizeof(float), nd, f) != nd)
  { fclose(f); free(l->scale); free(l->data); return 1; }
  fclose(f);
  return 0;
}

static void lut_fetch(const lut_t *l, const float rgb[3], float out[3])
{
  const int res = (int)l->res;
  int i = 0;
  for(int j=1;j<3;j++) if(rgb[j] >= rgb[i]) i = j;          /* largest component, ties to the last */
  const float z = rgb[i], sc = (res-1)/z;
  const float x = rgb[(i+1)%3]*sc, y = rgb[(i+2)%3]*sc;
  uint32_t xi = (uint32_t)x, yi = (uint32_t)y;
  if(xi > (uint32_t)res-2) xi = res-2;
  if(yi > (uint32_t)res-2) yi = res-2;
  /* largest zi with scale[zi] < z (binary search over res-1 intervals) */
  int left = 0, last = res-2, size = last;
  while(size > 0)
  {
    const int half = size >> 1, mid = left + half + 1;
    if(l->scale[mid] < z) { left = mid; size -= half+1; } else size = half;
  }
  const uint32_t zi = left < last ? left : last;
  size_t off = ((((size_t)i*res + zi)*res + yi)*res + xi)*3;
  const size_t dx = 3, dy = 3*(size_t)res, dz = 3*(size_t)res*res;
  const float x1 = x - xi, x0 = 1.f - x1, y1 = y - yi, y0 = 1.f - y1;
  const float z1 = (z - l->scale[zi])/(l->scale[zi+1] - l->scale[zi]), z0 = 1.f - z1;
  const float *d = l->data;
  for(int j=0;j<3;j++, off++)
    out[j] = ((d[off]*x0 + d[off+dx]*x1)*y0 + (d[off+dy]*x0 + d[off+dy+dx]*x1)*y1)*z0
           + ((d[off+dz]*x0 + d[off+dz+dx]*x1)*y0 + (d[off+dz+dy]*x0 + d[off+dz+dy+dx]*x1)*y1)*z1;
}


int ch_lut_rgb_to_coeff(const char *lut_path, const float rgb[3], float coeff[3], float *mul_out)
{ /* spectrum_rgb_to_coeff (include/spectrum.h:29-38) on a non-black colour: scale by the largest component when that exceeds 1,
     fetch. The division belongs to the contract too (-ffast-math turns it into a multiplication by the reciprocal).
     0 = ok, 1 = the table could not be read */
  lut_t l;
  if(lut_load(lut_path, &l)) return 1;
  float col[3];
  float mul = rgb[0] > rgb[1] ? rgb[0] : rgb[1];
  mul = mul > rgb[2] ? mul : rgb[2];
  if(mul == 0.0f || mul < 1.0f) mul = 1.0f;
  for(int k=0;k<3;k++) col[k] = rgb[k] / mul;
  lut_fetch(&l, col, coeff);
  free(l.scale); free(l.data);
  *mul_out = mul;
  return 0;
}
