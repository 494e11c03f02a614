/* oracle/refharness/unit_harness.c -- TEST INFRASTRUCTURE, build-container only (our code).
 *
 * Calls individual functions of the *real* reference (headers included from /root/reference by
 * oracle/Makefile) and prints known-answer vectors as JSON lines. tests/golden/make_golden.py runs
 * it and commits the output as fixtures; nothing here ships with the product.
 *
 *   unit_harness coeff <lut> r g b [r g b ...]     rgb2spec_fetch (include/rgb2spec.h:87-128) after
 *                                                  the scaling of spectrum_rgb_to_coeff (include/spectrum.h:29-38)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include <stdint.h>
#include <xmmintrin.h>
#ifndef MIN
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#endif
#include "rgb2spec.h"

static int cmd_coeff(int argc, char **argv)
{
  rgb2spec_t *m = rgb2spec_init(argv[0]);
  if(!m) { fprintf(stderr, "cannot load %s\n", argv[0]); return 1; }
  for(int a=1;a+2<argc;a+=3)
  {
    float rgb[3] = { atof(argv[a]), atof(argv[a+1]), atof(argv[a+2]) };
    float mul = MAX(MAX(rgb[0], rgb[1]), rgb[2]);
    if(mul == 0.0f || mul < 1.0f) mul = 1.0f;
    float col[3], out[3];
    for(int k=0;k<3;k++) col[k] = rgb[k]/mul;
    rgb2spec_fetch(m, col, out);
    printf("{\"rgb\": [%.9g, %.9g, %.9g], \"mul\": %.9g, \"coeff\": [%.9g, %.9g, %.9g], \"eval_fast\": [",
        rgb[0], rgb[1], rgb[2], mul, out[0], out[1], out[2]);
    for(int l=0;l<10;l++) printf("%s%.9g", l ? ", " : "", rgb2spec_eval_fast(out, 360.0f + 50.0f*l));
    printf("], \"eval_precise\": [");
    for(int l=0;l<10;l++) printf("%s%.9g", l ? ", " : "", rgb2spec_eval_precise(out, 360.0f + 50.0f*l));
    printf("]}\n");
  }
  return 0;
}

int main(int argc, char **argv)
{
  if(argc >= 3 && !strcmp(argv[1], "coeff")) return cmd_coeff(argc-2, argv+2);
  fprintf(stderr, "usage: unit_harness coeff <lut> r g b ...\n");
  return 1;
}
