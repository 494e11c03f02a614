/* ch_pfm.c -- PFM output/input in the reference's layout and the regression metric.
 *   writer: fb_export, include/framebuffer.h:142-175 ("PF\nW H\n-1.0", '0'-padded so that the pixel
 *           data starts 16-byte aligned, rows j = 0 first, value = fb * gain)
 *   metric: tools/img/pfmdiff.c:75-86, sqrt( sum_px (dR^2+dG^2+dB^2) / (W*H) )
 */
#include "ch_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int ch_pfm_write(const char *filename, const float *fb, uint32_t width, uint32_t height, float gain)
{
  FILE *f = fopen(filename, "wb");
  if(!f) return 1;
  char header[128];
  snprintf(header, sizeof(header), "PF\n%u %u\n-1.0", width, height);
  size_t len = strlen(header);
  fputs(header, f);
  while((len + 1) & 0xf) { fputc('0', f); len++; }
  fputc('\n', f);
  float *row = (float *)malloc(sizeof(float)*3*width);
  for(uint32_t j=0;j<height;j++)
  {
    for(uint32_t i=0;i<3*width;i++) row[i] = fb[3*(size_t)width*j + i]*gain;
    fwrite(row, sizeof(float), 3*width, f);
  }
  free(row);
  fclose(f);
  return 0;
}

int ch_pfm_read(const char *filename, float **fb, uint32_t *width, uint32_t *height)
{
  FILE *f = fopen(filename, "rb");
  if(!f) return 1;
  char magic[3] = {0}, scale[64];
  unsigned w = 0, h = 0;
  if(fscanf(f, "%2s %u %u %63s", magic, &w, &h, scale) != 4 || strcmp(magic, "PF") || !w || !h) { fclose(f); return 1; }
  fgetc(f);   /* the single whitespace after the scale token */
  float *d = (float *)malloc(sizeof(float)*3*(size_t)w*h);
  if(!d || fread(d, sizeof(float), 3*(size_t)w*h, f) != 3*(size_t)w*h) { fclose(f); free(d); return 1; }
  fclose(f);
  *fb = d; *width = w; *height = h;
  return 0;
}

double ch_pfm_rmse(const float *a, const float *b, uint32_t width, uint32_t height)
{
  double sum = 0.0;
  const size_t n = (size_t)width*height;
  for(size_t k=0;k<3*n;k++) { const double d = (double)a[k] - (double)b[k]; sum += d*d; }
  return sqrt(sum/(double)n);
}
