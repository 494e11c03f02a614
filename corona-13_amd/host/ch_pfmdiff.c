/* ch_pfmdiff.c -- compare two PFM images the way the reference's regression scripts do
 * (tools/img/pfmdiff.c:75-86: rmse = sqrt(sum over pixels and channels of (a-b)^2 / (width*height)),
 * printed as "[pfmdiff] rmse: %g"). Usage: pfmdiff-mi a.pfm b.pfm; exit status 0 = compared, 2 = unreadable / size mismatch. */
#include "ch_host.h"
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char *argv[])
{
  if(argc < 3) { fprintf(stderr, "usage: %s a.pfm b.pfm\n", argv[0]); return 1; }
  float *a = 0, *b = 0;
  uint32_t wa, ha, wb, hb;
  if(ch_pfm_read(argv[1], &a, &wa, &ha) || ch_pfm_read(argv[2], &b, &wb, &hb)) { fprintf(stderr, "[pfmdiff] could not read the images\n"); return 2; }
  if(wa != wb || ha != hb) { fprintf(stderr, "[pfmdiff] image dimensions don't match! (%ux%u vs %ux%u)\n", wa, ha, wb, hb); return 2; }
  fprintf(stdout, "[pfmdiff] rmse: %g\n", ch_pfm_rmse(a, b, wa, ha));
  free(a); free(b);
  return 0;
}
