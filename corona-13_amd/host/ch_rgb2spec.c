/* ch_rgb2spec.c -- RGB -> smooth reflectance spectrum (sigmoid of a quadratic polynomial,
 * Jakob & Hanika 2019), host side, init time only.
 *
 * Reference behaviour: spectrum_rgb_to_coeff (include/spectrum.h:29-38) divides by max(rgb) when
 * that is > 1 and fetches three coefficients trilinearly from a 64^3 LUT (rgb2spec_fetch,
 * include/rgb2spec.h:87-128; file format "SPEC", u32 res, float scale[res], float data[3*res^3*3],
 * include/rgb2spec.h:28-64). Run-time evaluation is rgb2spec_eval_fast (include/rgb2spec.h:145-149).
 *
 * Three sources, in this order:
 *   1. a reference-format LUT file if the caller names one (exactly the reference's numbers),
 *   2. a closed form for achromatic colours (constant spectrum: c0 = c1 = 0),
 *   3. a small Gauss-Newton fit in XYZ under illuminant E for chromatic colours.
 * Black is special-cased to (0,0,0) with mul = 0: the reference divides by zero there and relies on
 * NaN-swallowing clamps -- the value it ends up with is 0 for every slot.
 */
#include "ch_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

float ch_coeff_eval(const float coeff[3], float lambda)
{
  const float x = (coeff[0]*lambda + coeff[1])*lambda + coeff[2];
  const float y = 1.0f/sqrtf(x*x + 1.0f);
  return .5f*x*y + .5f;
}

/* --- direct fit ----------------------------------------------------------------------- */
/* eRGB ("linear rec709 adapted to illuminant E") -> XYZ, rows sum to the E white point (1,1,1). */
static const double ergb_to_xyz[3][3] = {
  {0.496859, 0.339094, 0.164047},
  {0.256193, 0.678188, 0.065619},
  {0.023290, 0.113031, 0.863978},
};

extern const float *ch_cie_table(void);   /* 96x3, set by the scene loader */

static void spec_to_xyz(const double c[3], const float *cie, double xyz[3], double jac[3][3])
{ /* integrate sigmoid(poly(lambda)) against the CMFs under illuminant E, normalised so that a
     constant 1 spectrum maps to (1,1,1) per channel */
  double sum[3] = {0}, norm[3] = {0}, j[3][3] = {{0}};
  for(int i=0;i<95;i++)
  {
    const double l = 360.0 + 5.0*i;
    const double x = (c[0]*l + c[1])*l + c[2];
    const double r = 1.0/sqrt(1.0 + x*x);
    const double s = .5*x*r + .5, ds = .5*r*r*r;
    for(int k=0;k<3;k++)
    {
      const double w = cie[3*i+k];
      sum[k] += w*s; norm[k] += w;
      j[k][0] += w*ds*l*l; j[k][1] += w*ds*l; j[k][2] += w*ds;
    }
  }
  for(int k=0;k<3;k++)
  {
    xyz[k] = sum[k]/norm[k];
    for(int m=0;m<3;m++) jac[k][m] = j[k][m]/norm[k];
  }
}

static int solve3(double a[3][3], double b[3], double x[3])
{
  int p[3] = {0,1,2};
  for(int c=0;c<3;c++)
  {
    int best = c;
    for(int r=c+1;r<3;r++) if(fabs(a[p[r]][c]) > fabs(a[p[best]][c])) best = r;
    int t = p[c]; p[c] = p[best]; p[best] = t;
    if(fabs(a[p[c]][c]) < 1e-300) return 1;
    for(int r=c+1;r<3;r++)
    {
      const double f = a[p[r]][c]/a[p[c]][c];
      for(int k=c;k<3;k++) a[p[r]][k] -= f*a[p[c]][k];
      b[p[r]] -= f*b[p[c]];
    }
  }
  for(int c=2;c>=0;c--)
  {
    double s = b[p[c]];
    for(int k=c+1;k<3;k++) s -= a[p[c]][k]*x[k];
    x[c] = s/a[p[c]][c];
  }
  return 0;
}

static int fit_chromatic(const float rgb[3], float coeff[3])
{
  const float *cie = ch_cie_table();
  if(!cie) return 1;
  double target[3];
  for(int k=0;k<3;k++) target[k] = ergb_to_xyz[k][0]*rgb[0] + ergb_to_xyz[k][1]*rgb[1] + ergb_to_xyz[k][2]*rgb[2];
  /* polynomial in normalised wavelength t = (l-360)/470 for conditioning, converted at the end */
  double c[3] = {0, 0, 0};
  const double mean = (rgb[0]+rgb[1]+rgb[2])/3.0;
  const double m = fmin(fmax(mean, 1e-3), 1.0-1e-3);
  c[2] = (m - .5)/sqrt(m*(1.0-m));
  for(int it=0;it<50;it++)
  {
    /* c (normalised) -> nm coefficients */
    const double s = 1.0/470.0, o = -360.0/470.0;
    double cn[3] = { c[0]*s*s, 2*c[0]*s*o + c[1]*s, c[0]*o*o + c[1]*o + c[2] };
    double xyz[3], jn[3][3], jac[3][3];
    spec_to_xyz(cn, cie, xyz, jn);
    /* chain rule d cn / d c */
    for(int k=0;k<3;k++)
    {
      jac[k][0] = jn[k][0]*s*s + jn[k][1]*2*s*o + jn[k][2]*o*o;
      jac[k][1] = jn[k][1]*s + jn[k][2]*o;
      jac[k][2] = jn[k][2];
    }
    double r[3], d[3];
    double err = 0;
    for(int k=0;k<3;k++) { r[k] = xyz[k] - target[k]; err += r[k]*r[k]; }
    if(err < 1e-14) break;
    if(solve3(jac, r, d)) break;
    for(int k=0;k<3;k++) c[k] -= d[k];
    const double mx = fmax(fmax(fabs(c[0]), fabs(c[1])), fabs(c[2]));
    if(mx > 200.0) for(int k=0;k<3;k++) c[k] *= 200.0/mx;
  }
  const double s = 1.0/470.0, o = -360.0/470.0;
  coeff[0] = (float)(c[0]*s*s);
  coeff[1] = (float)(2*c[0]*s*o + c[1]*s);
  coeff[2] = (float)(c[0]*o*o + c[1]*o + c[2]);
  return 0;
}

float ch_rgb_to_coeff(const float rgb[3], float coeff[3], const char *lut_path)
{
  float mul = fmaxf(fmaxf(rgb[0], rgb[1]), rgb[2]);
  if(mul == 0.0f)
  { /* black */
    coeff[0] = coeff[1] = coeff[2] = 0.0f;
    return 0.0f;
  }
  if(lut_path)
  {
    float lmul;
    if(!ch_lut_rgb_to_coeff(lut_path, rgb, coeff, &lmul)) return lmul;
    fprintf(stderr, "[ch] could not load rgb2spec lut `%s', fitting coefficients directly\n", lut_path);
  }
  if(mul < 1.0f) mul = 1.0f;
  float col[3];
  for(int k=0;k<3;k++) col[k] = rgb[k]/mul;
  if(col[0] == col[1] && col[1] == col[2])
  { /* constant spectrum s: invert the sigmoid. s = 1 saturates; cap like the published optimiser (|c| <= 200) */
    const float s = col[0];
    coeff[0] = coeff[1] = 0.0f;
    coeff[2] = s >= 1.0f ? 200.0f : (s - .5f)/sqrtf(s*(1.0f-s));
    if(coeff[2] > 200.0f) coeff[2] = 200.0f;
    return mul;
  }
  if(fit_chromatic(col, coeff))
  {
    coeff[0] = coeff[1] = 0.0f; coeff[2] = 0.0f;
    fprintf(stderr, "[ch] rgb2spec fit failed for (%g %g %g)\n", rgb[0], rgb[1], rgb[2]);
  }
  return mul;
}
