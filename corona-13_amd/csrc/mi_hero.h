/* mi_hero.h -- hero wavelengths: the shading of the HERO instantiations of the megakernel (mi_megakernel.h), i.e. the reference built with
 * -DMF_COUNT=4 (include/mf.h:280-423; mi_scene_set_wavelengths, include/corona_mi.h).
 *
 * A path carries FOUR wavelengths. The first (component 0, the hero) decides everything geometric -- which microfacet, reflect or transmit,
 * Russian roulette --, all four weigh the path, and one splat adds the four colours (src/pathspace.c:189,218-221,253; src/sampler.d/pt.c:30-54,
 * ptdl.c:78-148; include/pathspace/nee.h:188-191; src/shaders/dielectric.c:240-415; src/view.c:455-463 with include/spectrum.h:185-195).
 *
 * One LANE per path, as in every other kernel: the traversal is the scalar kernels' (a ray is a ray), and the spectral part of a vertex --
 * colours, indices of refraction, Fresnel terms, pdfs, weights -- is evaluated for l = 0..3 by the SAME device functions the scalar kernels
 * use (mi_kernels.h) where a quantity has one wavelength in it (colours, indices of refraction), and by four-component versions of the bsdf
 * functions below: the geometry -- surface, sampled microfacet, emitter sample -- once, the Fresnel terms, pdfs and weights in loops over
 * l = 0..3, expression for expression what the scalar function computes for one component (the first version called the scalar functions
 * four times: same paths, 1.4 times the kernel time, profiles/r05_hero.txt). The places where the reference looks across the components
 * (mf_any, mf_all, mf(x, 0), mf_hsum) are written out.
 * Component 0 lives in the PathState the megakernel knows (ray, pdf product, throughput ...: what the traversal slices and the parking
 * of path state touch), components 1..3 in the arrays of PathStateHero behind it: 18 registers more.
 * Every kind of scene the scalar kernels take (MEDIA / MB switches as there), `rand` or `halton` point sampler: what the reference's MF_COUNT = 4
 * build was pinned on (tests/test_oracle_hero.py). The exchange between waves (mi_regroup.h) carries the fifteen words of components 1..3 that are live
 * between two rays in eight more slots of a pool entry. */
#ifndef MI_HERO_H
#define MI_HERO_H

#include "mi_path.h"

struct PathStateHero : PathState
{
  float lambda_x[3], throughput_x[3], pdf_x[3], sh_value_x[3];
  double pdfprod_x[3];
  /* (no cur_ior_x: the index of refraction of the volume a ray travels in is that of the innermost shape of the path's nesting stack, at the
     component's wavelength -- shape_interior_ior(media_top_shape(ps.media), lambda_l), one 16-byte load and four divisions per vertex instead
     of three registers through every traversal slice; component 0 keeps the PathState's cur_ior, which the scalar code maintains) */
};

__device__ __forceinline__ float hero_hsum(const float *a) { return (a[0] + a[1]) + (a[2] + a[3]); }   /* mf_hsum: two _mm_hadd_ps, include/mf.h:301-306 */

/* view_splat for four components: mf_any(value > 0) && mf_all(value < FLT_MAX) && mf_all(value == value) (src/view.c:455-463), then
   spectrum_p_to_xyz: xyz[k] += cmf_k(lambda_l) p[l] for l = 0..3 in that order (include/spectrum.h:185-195) */
__device__ __forceinline__ bool hero_splat_colour(const DScene &sc, const float *lam, const float *value, float *col)
{
  const bool ok = (value[0] > 0.0f || value[1] > 0.0f || value[2] > 0.0f || value[3] > 0.0f) &&
                  (value[0] < FLT_MAX && value[1] < FLT_MAX && value[2] < FLT_MAX && value[3] < FLT_MAX) &&
                  (value[0] == value[0] && value[1] == value[1] && value[2] == value[2] && value[3] == value[3]);
  col[0] = col[1] = col[2] = 0.0f;
  if(ok)
  {
    float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for(int l=0;l<MI_MF;l++)
    {
      float c[3];
      spectrum_to_xyz(sc, lam[l], value[l], c);
      sum[0] += c[0]; sum[1] += c[1]; sum[2] += c[2];
    }
    col[0] = sum[0]; col[1] = sum[1]; col[2] = sum[2];
  }
  return ok;
}

template<bool RECORD>
__device__ __forceinline__ void hero_rec_vertex(const DScene &sc, unsigned long long slot, int v, const float *thr, const float *pdf, const Shading *sh, const float *eta)
{
  if(!RECORD || v >= MI_REC_MAX_VERTS || !sc.hero_ext) return;
  mi_hero_ext &x = sc.hero_ext[slot];
#pragma unroll
  for(int l=0;l<MI_MF;l++)
  {
    x.throughput[v][l] = thr[l]; x.pdf[v][l] = pdf[l];
    x.rd[v][l] = sh ? sh[l].rd : 0.0f; x.rg[v][l] = sh ? sh[l].rg : 0.0f; x.em[v][l] = sh ? sh[l].em : 0.0f; x.eta[v][l] = eta ? eta[l] : 0.0f;
  }
}

template<bool RECORD, bool HALTON, bool MEDIA = false, class CNT>
__device__ __forceinline__ void path_generate_hero(const DScene &sc, PathStateHero &ps, unsigned long long index, mi_path_record *rec, unsigned long long slot, CNT &cnt,
                                                   float px = -1.0f, float py = -1.0f)
{ /* (HALTON: the four draws of path_init ask for the same dimension and get the same number -- the wavelengths are a quarter of the range apart) */
  path_generate<RECORD, HALTON, MEDIA, CNT, true>(sc, ps, index, rec, cnt, px, py, ps.lambda_x);
#pragma unroll
  for(int l=0;l<3;l++) { ps.throughput_x[l] = ps.throughput; ps.pdf_x[l] = ps.pdf; ps.pdfprod_x[l] = 1.0; ps.sh_value_x[l] = 0.0f; }
  if(RECORD && sc.hero_ext)
  {
    mi_hero_ext &x = sc.hero_ext[slot];
    x.lambda[0] = ps.lambda; x.lambda[1] = ps.lambda_x[0]; x.lambda[2] = ps.lambda_x[1]; x.lambda[3] = ps.lambda_x[2];
    const float thr[4] = {ps.throughput, ps.throughput, ps.throughput, ps.throughput}, one[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    hero_rec_vertex<RECORD>(sc, slot, 0, thr, one, nullptr, nullptr);
  }
}

/* the splat of a next-event connection that path_visible has passed */
template<bool RECORD, class CNT>
__device__ __forceinline__ void shadow_splat_hero(const DScene &sc, PathStateHero &ps, mi_path_record *rec, unsigned long long slot, CNT &cnt, SplatReq &splat)
{
  ps.sh_pending = 0;
  const float lam[4] = {ps.lambda, ps.lambda_x[0], ps.lambda_x[1], ps.lambda_x[2]};
  const float value[4] = {ps.sh_value, ps.sh_value_x[0], ps.sh_value_x[1], ps.sh_value_x[2]};
  float col[3];
  const bool ok = hero_splat_colour(sc, lam, value, col);
  if(RECORD)
  {
    if(rec->num_splats < MI_REC_MAX_SPLATS)
    {
      if(sc.hero_ext) for(int l=0;l<MI_MF;l++) sc.hero_ext[slot].splat_value[rec->num_splats][l] = value[l];
      mi_path_splat &sp = rec->splat[rec->num_splats++];
      sp.length = ps.sh_length; sp.tech = s_tech_nee; sp.value = value[0];
      sp.col[0] = col[0]; sp.col[1] = col[1]; sp.col[2] = col[2];
    }
  }
  if(ok)
  {
    MI_COUNT(cnt, 5, 1);
    if(!RECORD) { splat.pending = true; splat.c0 = col[0]; splat.c1 = col[1]; splat.c2 = col[2]; }
  }
}

template<bool RECORD, class CNT>
__device__ __forceinline__ void shadow_resolve_hero(const DScene &sc, PathStateHero &ps, const Hit &hit, mi_path_record *rec, unsigned long long slot, CNT &cnt, SplatReq &splat)
{ /* path_visible, src/pathspace.c:311-344 */
  ps.sh_pending = 0;
  const bool visible = (hit.dist >= ps.sh_dist) || (hit.prim == MI_NOPRIM) || (hit.prim == (ps.sh_light & ~MI_LIGHT_ANYHIT));
  if(visible) shadow_splat_hero<RECORD>(sc, ps, rec, slot, cnt, splat);
}

/* ------------------------------------------------------------------------------------------ the bsdfs for four components
 * The scalar kernels' functions (mi_kernels.h: sample_* / brdf_* / pdf_*) with the geometry -- the microfacet, the outgoing direction, the
 * shadowing terms -- computed once and the wavelength-dependent part (index of refraction, Fresnel term, colour) in a loop over the
 * components, expression for expression what the scalar function computes for that component. */
struct HeroSample
{
  V3 omega;
  uint32_t mode;
  float pdf[MI_MF], weight[MI_MF];
};
struct HeroEval { float value[MI_MF]; uint32_t mode; };

template<class PS>
__device__ __forceinline__ void sample_diffuse_hero(PS &pts, const Surf &sf, const Shading *sh, bool any_rd, uint32_t mode_in, HeroSample &bs)
{ /* sample_d, src/shader.c:165-205 */
  const float x1 = pts(MI_DIM_OMEGA_X);
  const float x2 = pts(MI_DIM_OMEGA_Y);
  const float sq = mi_sqrt(x1);
  const float c0 = mi_sqrt((float)(1.0 - (double)x1));
  const float ang = (float)(2*MI_PI_D*(double)x2);
  float sn, cs;
  mi_sincosf(ang, &sn, &cs);
  const float c1 = sq*cs, c2 = sq*sn;
  bs.omega = mk3(c0*sf.n.x + c1*sf.a.x + c2*sf.b.x, c0*sf.n.y + c1*sf.a.y + c2*sf.b.y, c0*sf.n.z + c1*sf.a.z + c2*sf.b.z);
  bs.mode = mode_in;
#pragma unroll
  for(int l=0;l<MI_MF;l++) { bs.pdf[l] = (float)(1.0f/MI_PI_D); bs.weight[l] = 0.0f; }
  const float cos_out_ng = dot3(sf.gn, bs.omega);
  if(sf.flags & s_inside) { if(cos_out_ng >= 0.0f) return; }
  else if(cos_out_ng <= 0.0f) return;
#pragma unroll
  for(int l=0;l<MI_MF;l++) bs.weight[l] = sh[l].rd;
  if(any_rd) bs.mode = s_diffuse | s_reflect;                      /* mf_any(mf_gt(throughput, 0)), src/shader.c:202 */
}

template<class PS>
__device__ __forceinline__ void sample_dielectric_hero(PS &pts, const Surf &sf, const Shading *sh, const V3 wi, const float *eta, bool any_im,
                                                       uint32_t mode_in, HeroSample &bs)
{ /* sample, dielectric.c:240-415 with MF_COUNT = 4: the microfacet, the reflect / transmit choice and the outgoing direction are the hero's
     (mf(eta_ratio, 0), mf(R, 0), mf(cost2, 0), mf(cost, 0)); every component weighs that direction with its own index of refraction */
  bs.mode = mode_in; bs.omega = mk3(0, 0, 0);
#pragma unroll
  for(int l=0;l<MI_MF;l++) { bs.weight[l] = 0.0f; bs.pdf[l] = 1.0f; }
  if(eta[0] < 0.0f) return;
  if(any_im)
  {
    bs.omega = wi;
    bs.mode = s_specular | s_transmit;
#pragma unroll
    for(int l=0;l<MI_MF;l++) bs.weight[l] = sh[l].rg;
    return;
  }
  const V3 n = sf.n;
  float pdf_h = 1.0f;
  V3 h = n;
  const float r = sh[0].roughness;
  const float cos_in = -dot3(sf.n, wi);
  if(r > GLOSSY_THR)
  {
    const V3 wit = mk3(-dot3(sf.a, wi), -dot3(sf.b, wi), cos_in);
    /* the MF_COUNT = 4 reference: this plugin only builds with clang, which evaluates call arguments left to right (tests/test_oracle_hero.py);
       the scalar build (gcc) draws them the other way round */
    const float U1 = pts(MI_DIM_OMEGA_X);
    const float U2 = pts(MI_DIM_OMEGA_Y);
    const V3 ht = ggx_sample_h(wit, r, r, U1, U2);
    h = mk3(ht.x*sf.a.x + ht.y*sf.b.x + ht.z*n.x, ht.x*sf.a.y + ht.y*sf.b.y + ht.z*n.y, ht.x*sf.a.z + ht.y*sf.b.z + ht.z*n.z);
    pdf_h = ggx_pdf_h(wi, h, n, r);
  }
  float pdf = pdf_h;
  const float cosr = -dot3(wi, h);
  if(cosr <= 0.0f) return;
  const float n2 = 1.0f;
  float cost2[MI_MF], cost[MI_MF], R[MI_MF];
#pragma unroll
  for(int l=0;l<MI_MF;l++)
  {
    const float nr = eta[l]/n2;
    cost2[l] = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
    cost[l] = cost2[l] <= 0.0f ? 0.0f : mi_sqrt(cost2[l]);
    R[l] = fresnel_dielectric(eta[l], n2, cosr, cost[l]);
  }
  if(pts(MI_DIM_SCATTER_MODE) <= R[0])
  {
    bs.mode = s_reflect;
    bs.omega = mk3(wi.x + 2.0f*cosr*h.x, wi.y + 2.0f*cosr*h.y, wi.z + 2.0f*cosr*h.z);
    if(dot3(bs.omega, n) <= 0.0f) return;
    pdf *= mi_rcp(4.0f*cosr);
    if(r > GLOSSY_THR)
    {
      const float pdf_c = pdf/fabsf(dot3(bs.omega, n));
#pragma unroll
      for(int l=0;l<MI_MF;l++) bs.pdf[l] = R[l]*pdf_c;
      bs.mode |= s_glossy;
      if(dot3(bs.omega, n)*dot3(bs.omega, h) < 0.0f) return;
      const float G1 = ggx_G1(bs.omega, n, r);
#pragma unroll
      for(int l=0;l<MI_MF;l++) bs.weight[l] = sh[l].rg*G1;
      return;
    }
    bs.mode = s_reflect | s_specular;
#pragma unroll
    for(int l=0;l<MI_MF;l++) { bs.pdf[l] = R[l]; bs.weight[l] = sh[l].rg; }
  }
  else
  {
    if(cost2[0] <= 0.0f) return;                                     /* "can't sample hero, we're all dead" */
    const float f = eta[0]*cosr - cost[0];
    bs.omega = normalise3(mk3(wi.x*eta[0] + f*h.x, wi.y*eta[0] + f*h.y, wi.z*eta[0] + f*h.z));
    if(dot3(bs.omega, n) >= 0.0f) return;
    if(r <= GLOSSY_THR)
    { /* "specular transmit always selects single wavelength": mask = mf_hero = _mm_set_epi32(0, ~0, ~0, ~0) (include/mf.h:300) zeroes the
         components whose mask is set, and _mm_set_epi32 lists the highest element first -- components 0, 1, 2 die, component 3 carries on */
      bs.mode = s_specular | s_transmit;
#pragma unroll
      for(int l=0;l<MI_MF;l++) { bs.pdf[l] = l == MI_MF - 1 ? 1.0f - R[l] : 0.0f; bs.weight[l] = l == MI_MF - 1 ? sh[l].rg : 0.0f; }
      return;
    }
    /* dielectric.c:353-411: the sampled half vector connects wi and wo for the hero's index of refraction only; every component reconstructs
       the one ITS index needs, with its own Fresnel term */
    const V3 wo = bs.omega;
    bs.mode = s_transmit | s_glossy;
    const float G1 = ggx_G1(wo, n, r);
    const float cos_on = fabsf(dot3(wo, n));
#pragma unroll
    for(int l=0;l<MI_MF;l++)
    {
      const float n1 = eta[l], nr = n1/n2;
      bool mask = false;
      float h0 = n1*wi.x - n2*wo.x, h1 = n1*wi.y - n2*wo.y, h2 = n1*wi.z - n2*wo.z;
      const float hilen = 1.0f/mi_sqrt(h0*h0 + (h1*h1 + h2*h2));
      h0 *= hilen; h1 *= hilen; h2 *= hilen;
      if(n2 < n1) { h0 = -h0; h1 = -h1; h2 = -h2; }
      const float cosh2 = h0*n.x + (h1*n.y + h2*n.z);
      mask |= cosh2 < 0.0f;
      const float cosr2 = h0*-wi.x + (h1*-wi.y + h2*-wi.z);
      mask |= cosr2 <= 0.0f;
      const float cost2b = 1.0f - (nr*nr)*(1.0f - cosr2*cosr2);
      const float costb = cost2b <= 0.0f ? 0.0f : mi_sqrt(cost2b);
      const float R2 = fresnel_dielectric(n1, n2, cosr2, costb);
      const float denom = n1*cosr2 - n2*costb;
      float pdf2 = ggx_pdf_h_cos(cosh2, cos_in, cosr2, r);
      pdf2 = pdf2*(((n2*n2)*costb)/(denom*denom));
      bs.pdf[l] = mask ? 0.0f : (pdf2*(1.0f - R2))/cos_on;
      bs.weight[l] = mask ? 0.0f : sh[l].rg*G1;
    }
  }
}

template<class PS>
__device__ __forceinline__ void sample_metal_hero(const DScene &sc, PS &pts, const Surf &sf, const Shading *sh, const V3 wi, const float *n1, int mat,
                                                  const float *lam, uint32_t mode_in, HeroSample &bs)
{ /* sample, metal.c:219-265: one microfacet, the conductor's n and k at each wavelength */
  bs.mode = mode_in; bs.omega = mk3(0, 0, 0);
#pragma unroll
  for(int l=0;l<MI_MF;l++) { bs.weight[l] = 0.0f; bs.pdf[l] = 1.0f; }
  const V3 n = sf.n;
  V3 h = n;
  float pdf_h = 1.0f;
  const float r = sh[0].roughness;
  if(r > 1e-4f)
  {
    const V3 wit = mk3(-dot3(sf.a, wi), -dot3(sf.b, wi), -dot3(n, wi));
    const float U1 = pts(MI_DIM_OMEGA_X);           /* clang's order, see sample_dielectric_hero */
    const float U2 = pts(MI_DIM_OMEGA_Y);
    const V3 ht = ggx_sample_h(wit, r, r, U1, U2);
    h = mk3(ht.x*sf.a.x + ht.y*sf.b.x + ht.z*n.x, ht.x*sf.a.y + ht.y*sf.b.y + ht.z*n.y, ht.x*sf.a.z + ht.y*sf.b.z + ht.z*n.z);
    pdf_h = ggx_pdf_h(wi, h, n, r);
  }
  float pdf = pdf_h;
  const float cosr = -dot3(wi, h);
  if(!(cosr > 0.0f)) return;
  float R[MI_MF];
#pragma unroll
  for(int l=0;l<MI_MF;l++)
  {
    const int i = (int)DCLAMP((lam[l] - 360.0f)/5.0f, 0, 94);
    const float n2 = sc.metal_ior[(mat*95 + i)*2 + 0], k2 = -sc.metal_ior[(mat*95 + i)*2 + 1];
    R[l] = (sc.metal_reference && metal_reference_kills(n1[l], n2, k2, cosr)) ? 0.0f : fresnel_metal(n1[l], n2, k2, cosr);
  }
  bs.mode = s_reflect;
  bs.omega = mk3(wi.x + 2.0f*cosr*h.x, wi.y + 2.0f*cosr*h.y, wi.z + 2.0f*cosr*h.z);
  if(dot3(bs.omega, n) <= 0.0f) return;
  pdf *= mi_rcp(4.0f*cosr);
  if(r > 1e-4f)
  {
    const float p = pdf/fabsf(dot3(bs.omega, n));
#pragma unroll
    for(int l=0;l<MI_MF;l++) bs.pdf[l] = p;
    bs.mode |= s_glossy;
    if(dot3(bs.omega, n)*dot3(bs.omega, h) < 0.0f) return;
    const float G1 = ggx_G1(bs.omega, n, r);
#pragma unroll
    for(int l=0;l<MI_MF;l++) bs.weight[l] = R[l]*(sh[l].rg*G1);
    return;
  }
  bs.mode |= s_specular;
#pragma unroll
  for(int l=0;l<MI_MF;l++) bs.weight[l] = R[l]*sh[l].rg;
}

__device__ __forceinline__ HeroEval brdf_diffuse_hero(const Surf &sf, const Shading *sh, const V3 wo)
{ /* brdf_d, src/shader.c:207-252 */
  HeroEval r; r.mode = s_diffuse | s_reflect;
#pragma unroll
  for(int l=0;l<MI_MF;l++) r.value[l] = 0.0f;
  const float cos_out_ns = dot3(sf.n, wo);
  if(cos_out_ns <= 0) return r;
  const float cos_out_ng = dot3(sf.gn, wo);
  if(sf.flags & s_inside) { if(cos_out_ng >= 0.0f) return r; }
  else if(cos_out_ng <= 0.0f) return r;
#pragma unroll
  for(int l=0;l<MI_MF;l++) r.value[l] = (float)((double)sh[l].rd*((double)1.0f/MI_PI_D));
  return r;
}

/* the transmission branches of brdf / pdf: the half vector depends on the component's index of refraction, so everything does */
__device__ __forceinline__ float brdf_dielectric_transmit1(const V3 n, const V3 wi, const V3 wo, float cos_in, float cos_out, float n1, float rg, float r, bool glossy)
{ /* brdf, dielectric.c:478-541, one component */
  const float n2 = 1.0f;
  bool mask = false;
  float h0 = n1*wi.x - n2*wo.x, h1 = n1*wi.y - n2*wo.y, h2 = n1*wi.z - n2*wo.z;
  const float hilen = mi_rcp(mi_sqrt(h0*h0 + (h1*h1 + h2*h2)));
  h0 *= hilen; h1 *= hilen; h2 *= hilen;
  float cosh2 = h0*n.x + (h1*n.y + h2*n.z);
  const bool cosh_lt0 = cosh2 < 0.0f;
  mask |= cosh_lt0 && (n1 < n2);
  mask |= !cosh_lt0 && (n2 < n1);
  if(cosh_lt0) { cosh2 = -cosh2; h0 = -h0; h1 = -h1; h2 = -h2; }
  const float cosr2 = h0*-wi.x + (h1*-wi.y + h2*-wi.z);
  mask |= cosr2 <= 0.0f;
  const float nr = n1/n2;
  const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr2*cosr2);
  const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
  const float R2 = fresnel_dielectric(n1, n2, cosr2, cost);
  const float DG1 = ggx_pdf_h_cos(cosh2, cos_in, cosr2, r);
  const float G1 = ggx_G1_cos(cos_in, r);
  const float cos_hwo = h0*wo.x + (h1*wo.y + h2*wo.z);
  mask |= cos_hwo >= 0.0f;
  float denom = n1*cosr2 - n2*cost;
  denom = denom*denom;
  if(glossy) return mask ? 0.0f : ((rg*(1.0f - R2))*((n2*n2)*(cost*(DG1*(G1*(1.0f/fabsf(cos_out)))))))/denom;
  mask |= cosh2 < HALFVEC_COS_THR;
  return mask ? 0.0f : rg*DCLAMP(1.0f - R2, 0.0f, 1.0f);
}

__device__ __forceinline__ HeroEval brdf_dielectric_hero(const Surf &sf, const Shading *sh, const V3 wi, const V3 wo, const float *eta, bool any_im)
{ /* brdf, dielectric.c:418-541: mf(eta_ratio, 0) < 0 and mf_any(indexmatched) decide for all components */
  HeroEval res; res.mode = s_absorb;
#pragma unroll
  for(int l=0;l<MI_MF;l++) res.value[l] = 0.0f;
  const V3 n = sf.n;
  const float cos_in  = -dot3(n, wi);
  const float cos_out =  dot3(n, wo);
  if(eta[0] < 0.0f) return res;
  const float n2 = 1.0f;
  const bool index_matched = any_im;
  if(cos_out == 0.0f || cos_in == 0.0f) return res;
  if(!index_matched && (cos_in*cos_out > 0)) res.mode = s_reflect;
  else res.mode = s_transmit;
  const float r = sh[0].roughness;
  if((r > GLOSSY_THR) && !index_matched) res.mode |= s_glossy;
  else res.mode |= s_specular;
  if(index_matched)
  {
    const float dot_wo_n = dot3(wo, n);
    const V3 h = normalise3(mk3(-wi.x + wo.x - 2.0f*dot_wo_n*n.x, -wi.y + wo.y - 2.0f*dot_wo_n*n.y, -wi.z + wo.z - 2.0f*dot_wo_n*n.z));
    const float cosh = dot3(h, n);
    if(cosh < 0.0f) return res;
    if(cosh < HALFVEC_COS_THR) return res;
#pragma unroll
    for(int l=0;l<MI_MF;l++) res.value[l] = sh[l].rg;
    return res;
  }
  else if(res.mode & s_reflect)
  {
    const V3 h = normalise3(mk3(-wi.x + wo.x, -wi.y + wo.y, -wi.z + wo.z));
    const float cosh = dot3(h, n);
    if(cosh < 0.0f) return res;
    const float DG1 = (res.mode & s_specular) ? 1.0f : ggx_pdf_h(wi, h, n, r);
    if(DG1 == 0) return res;
    const float cosr = -dot3(h, wi);
    if(cosr < 0.0f) return res;
    const float G1 = ggx_G1(wo, n, r);
    const bool glossy = (res.mode & s_glossy) != 0;
    if(!glossy && cosh < HALFVEC_COS_THR) return res;
    const float geo = DG1*G1/(4.0f*fabsf(cosr*cos_out));
#pragma unroll
    for(int l=0;l<MI_MF;l++)
    {
      const float nr = eta[l]/n2;
      const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
      const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
      const float R = fresnel_dielectric(eta[l], n2, cosr, cost);
      res.value[l] = glossy ? (sh[l].rg*R)*geo : sh[l].rg*R;
    }
    return res;
  }
  else
  {
    if(cos_in == 0.0f) return res;
    const bool glossy = (res.mode & s_glossy) != 0;
#pragma unroll
    for(int l=0;l<MI_MF;l++) res.value[l] = brdf_dielectric_transmit1(n, wi, wo, cos_in, cos_out, eta[l], sh[l].rg, r, glossy);
    return res;
  }
}

__device__ __forceinline__ void pdf_dielectric_hero(const Surf &sf, const Shading *sh, const V3 wi, const V3 wo, const float *eta, bool any_im, uint32_t mode, float *out)
{ /* pdf, dielectric.c:96-237 */
#pragma unroll
  for(int l=0;l<MI_MF;l++) out[l] = 0.0f;
  const V3 n = sf.n;
  const float cos_in  = -dot3(n, wi);
  const float cos_out =  dot3(n, wo);
  if(cos_in*cos_out == 0.0f) return;
  if(cos_out > 0.0f && !(mode & s_reflect))  return;
  if(cos_out < 0.0f && !(mode & s_transmit)) return;
  if(eta[0] < 0.0f) return;                              /* mf_all(eta < 0): the nesting breaks for all components or for none */
  const float n2 = 1.0f;
  const float r = sh[0].roughness;
  if(any_im)
  {
    const float dot_wo_n = dot3(wo, n);
    const V3 h = normalise3(mk3(-wi.x + wo.x - 2.0f*dot_wo_n*n.x, -wi.y + wo.y - 2.0f*dot_wo_n*n.y, -wi.z + wo.z - 2.0f*dot_wo_n*n.z));
    const float cosh = dot3(h, n);
    if(mode != (s_transmit | s_specular)) return;
    if(cosh < HALFVEC_COS_THR) return;
#pragma unroll
    for(int l=0;l<MI_MF;l++) out[l] = 1.0f;
    return;
  }
  if(mode & s_reflect)
  {
    const V3 h = normalise3(sub3(wi, wo));
    const float cosh = fabsf(dot3(h, n));
    const float cosr = fabsf(dot3(h, wi));
    const bool spec = (mode & s_specular) != 0;
    const float geo = spec ? 0.0f : ggx_pdf_h_cos(cosh, cos_in, cosr, r);
    const float i4 = mi_rcp(4.0f*fabsf(dot3(wo, h)));
#pragma unroll
    for(int l=0;l<MI_MF;l++)
    {
      const float nr = eta[l]/n2;
      const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
      const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
      const float R = fresnel_dielectric(eta[l], n2, cosr, cost);
      if(spec) { out[l] = cosh < HALFVEC_COS_THR ? 0.0f : R; continue; }
      float pdf = 1.0f;
      pdf = pdf*i4;
      pdf = pdf*R;
      pdf = pdf*geo;
      pdf = pdf/fabsf(cos_out);
      out[l] = !(pdf > 0.0f) ? 0.0f : pdf;
    }
    return;
  }
#pragma unroll
  for(int l=0;l<MI_MF;l++)
  {
    const float n1 = eta[l];
    bool mask = false;
    float h0 = n1*wi.x - n2*wo.x, h1 = n1*wi.y - n2*wo.y, h2 = n1*wi.z - n2*wo.z;
    const float hilen = mi_rcp(mi_sqrt(h0*h0 + (h1*h1 + h2*h2)));
    h0 *= hilen; h1 *= hilen; h2 *= hilen;
    if(n2 < n1) { h0 = -h0; h1 = -h1; h2 = -h2; }
    const float cosh = h0*n.x + (h1*n.y + h2*n.z);
    mask |= cosh < 0.0f;
    const float cosr = h0*-wi.x + (h1*-wi.y + h2*-wi.z);
    mask |= cosr <= 0.0f;
    const float nr = n1/n2;
    const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
    const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
    const float R = fresnel_dielectric(n1, n2, cosr, cost);
    if(mode & s_specular) { mask |= cosh < HALFVEC_COS_THR; out[l] = mask ? 0.0f : DCLAMP(1.0f - R, 0.0f, 1.0f); continue; }
    float pdf = 1.0f;
    const float denom = n1*cosr - n2*cost;
    pdf = pdf*(((n2*n2)*cost)/(denom*denom));
    pdf = pdf*DCLAMP(1.0f - R, 0.0f, 1.0f);
    pdf = pdf*ggx_pdf_h_cos(cosh, cos_in, cosr, r);
    pdf = pdf/fabsf(cos_out);
    mask |= !(pdf > 0.0f);
    out[l] = mask ? 0.0f : pdf;
  }
}

__device__ __forceinline__ HeroEval brdf_metal_hero(const DScene &sc, const Surf &sf, const Shading *sh, const V3 wi, const V3 wo, const float *n1, int mat, const float *lam)
{ /* brdf, metal.c:268-310 */
  HeroEval res; res.mode = s_absorb;
#pragma unroll
  for(int l=0;l<MI_MF;l++) res.value[l] = 0.0f;
  const V3 n = sf.n;
  const float cos_in = -dot3(n, wi), cos_out = dot3(n, wo);
  if(cos_out <= 0.0f || cos_in <= 0.0f) return res;
  res.mode = s_reflect;
  const float r = sh[0].roughness;
  if(r > 1e-4f) res.mode |= s_glossy; else res.mode |= s_specular;
  const V3 h = normalise3(mk3(-wi.x + wo.x, -wi.y + wo.y, -wi.z + wo.z));
  const float cosh = dot3(h, n);
  if(cosh < 0.0f) return res;
  const float DG1 = ggx_pdf_h(wi, h, n, r);
  if(DG1 == 0) return res;
  const float cosr = -dot3(h, wi);
  if(cosr < 0.0f) return res;
  const float G1 = ggx_G1(wo, n, r);
  const bool glossy = (res.mode & s_glossy) != 0;
  if(!glossy && cosh < HALFVEC_COS_THR) return res;
  const float geo = DG1*G1/(4.0f*fabsf(cosr*cos_out));
#pragma unroll
  for(int l=0;l<MI_MF;l++)
  {
    const int i = (int)DCLAMP((lam[l] - 360.0f)/5.0f, 0, 94);
    const float n2 = sc.metal_ior[(mat*95 + i)*2 + 0], k2 = -sc.metal_ior[(mat*95 + i)*2 + 1];
    const float R = fresnel_metal(n1[l], n2, k2, cosr);
    res.value[l] = glossy ? (sh[l].rg*R)*geo : sh[l].rg*R;
  }
  return res;
}

/* run_prepare_ops (mi_kernels.h) for four wavelengths: every op is loaded once */
__device__ __forceinline__ void run_prepare_ops_hero(const DScene &sc, const DMaterial &m, uint32_t num_ops, const Surf &sf, const float *lam, Shading *sh)
{
#pragma unroll
  for(int l=0;l<MI_MF;l++) { sh[l].roughness = 1.0f; sh[l].rs = sh[l].rd = sh[l].rg = sh[l].em = 0.0f; }
  for(uint32_t k=0;k<num_ops;k++)
  {
    const uint4 oa = *(const uint4 *)&m.op[k];
    const float4 ob = *(const float4 *)((const char *)&m.op[k] + 16);
    mi_shade_op op;
    op.kind = oa.x; op.slot = oa.y; op.coeff[0] = __uint_as_float(oa.z); op.coeff[1] = __uint_as_float(oa.w);
    op.coeff[2] = ob.x; op.mul = ob.y; op.roughness = ob.z;
    if(op.kind == MI_OP_COLOR)
    {
#pragma unroll
      for(int l=0;l<MI_MF;l++)
      {
        sh[l].roughness = op.roughness;
        const float val = op.mul*spectrum_eval(op.coeff, lam[l]);
        if(op.slot == MI_SLOT_EMISSION) set_slot(sh[l], op.slot, val);
        else if(op.slot != MI_SLOT_UNUSED) set_slot(sh[l], op.slot, DCLAMP(val, 0.0f, 1.0f));
      }
    }
    else
    {
      const float u = sf.s, t = sf.t;
      const int i = (int)(14.0f*u) % 14, j = (int)(10.0f*t) % 10;
      const float xu = 14.0f*u, xt = 10.0f*t;
      const float fu = xu - truncf(xu), ft = xt - truncf(xt);
      const bool border = fu < 0.1f || fu > 0.9f || ft < 0.1f || ft > 0.9f;
#pragma unroll
      for(int l=0;l<MI_MF;l++)
      {
        float val;
        if(border) val = 0.3f;
        else
        {
          const int b = (int)((lam[l] - 380.0f)/10.0f);
          if(b < 0 || b >= 36) val = 0.0f;
          else val = sc.checker[36*(14*j + i) + b];
        }
        set_slot(sh[l], op.slot, val);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ media and emitters for four components */
/* the medium of the edge under way per component: component 0 is the PathState's (the scalar code maintains it), 1..3 are looked up where the
   path is -- the innermost shape of its nesting stack, or the exterior -- at their wavelengths (like the index of refraction: PathStateHero) */
__device__ __forceinline__ void hero_media(const DScene &sc, const PathStateHero &ps, const float *lam, Medium *med)
{
  med[0] = ps.cur;
  const int top = media_top_shape(ps.media);
#pragma unroll
  for(int l=1;l<MI_MF;l++) med[l] = shape_interior_medium(sc, top, lam[l]);
}
/* shader_vol_pdf towards a surface, src/shader.c:108-131: whether the edge has a pdf other than 1 is the HERO's question (mf(mu_s, 0) > 0), the
   value is the component's */
__device__ __forceinline__ float hero_pdf_to_surface(const Medium &hero, const Medium &m, float dist) { return (hero.med >= 0 && hero.mu_s > 0.0f) ? expf(-dist*m.mu_t) : 1.0f; }

/* lights_sample_next_event + the emitter's shader_prepare for four wavelengths: the point on the emitter once, its emission per component.
   Plain kernels: one-burst records from LDS or L2; extended kernels: records where the scene has them (light_record_sample, mi_path.h),
   else the generic chain emitter list -> primitive -> shading record -> material (moving or analytic emitters) */
template<bool HALTON, bool MEDIA, bool MB>
__device__ __forceinline__ void hero_sample_emitter(const DScene &sc, float r1, float r2, float r3, const float *lam, const V3 from, float scramble, float time,
                                                    uint32_t &lpe, Surf &ls, float *lem, float &lrough, float &lpdf, float &ldist, V3 &ol)
{
  if(!MB && (!MEDIA || sc.lights != nullptr))
  {
    const uint32_t t = (!MEDIA && sc.num_lights <= 4) ? sample_cdf4(sc.light_cdf4, (int)sc.num_lights, r1) : sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
    float4 q0, q1, q2, q3, q4, q5, q6, q7, q8, q9;
    if(!MEDIA && MI_LIGHTS_LDS && sc.num_lights <= MI_LIGHTS_LDS)
    {
      const float4 *lq = lights_lds<HALTON>() + t*(uint32_t)(sizeof(DLight)/16);
      q0 = lq[0]; q1 = lq[1]; q2 = lq[2]; q3 = lq[3]; q4 = lq[4]; q5 = lq[5]; q6 = lq[6]; q7 = lq[7]; q8 = lq[8]; q9 = lq[9];
    }
    else
    {
      const float4 *lq = (const float4 *)(sc.lights + t);
      q0 = lq[0]; q1 = lq[1]; q2 = lq[2]; q3 = lq[3]; q4 = lq[4]; q5 = lq[5]; q6 = lq[6]; q7 = lq[7]; q8 = lq[8]; q9 = lq[9];
    }
    lpe = __float_as_uint(q9.x);
    const bool quad = __float_as_uint(q9.y) == MI_PRIM_QUAD;
    const V3 v0 = mk3(q0.x, q0.y, q0.z), v1 = mk3(q0.w, q1.x, q1.y), v2 = mk3(q1.z, q1.w, q2.x), v3 = mk3(q2.y, q2.z, q2.w);
    float hu, hv;
    if(quad) { hu = r2; hv = r3; }
    else { const float a = mi_sqrt(r2); hu = r3*a; hv = (1.0f-r3)*a; }
    const bool second = quad && !(hv >= hu);
    const float u = second ? hu - hv : hu;
    const float vv = !quad ? hv : second ? hv : hv - hu;
    ls.x = second ? tri_retime(v0, v2, v3, u, vv) : tri_retime(v0, v1, v2, u, vv);
    ls.u = hu; ls.v = hv;
    ol = sub3(ls.x, from);
    ldist = mi_sqrt(dot3(ol, ol));
    const double il = 1./(double)ldist;
    ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
    const V3 n0 = mk3(q3.x, q3.y, q3.z);
    const V3 na = second ? mk3(q4.z, q4.w, q5.x) : mk3(q3.w, q4.x, q4.y);
    const V3 nb = second ? mk3(q5.y, q5.z, q5.w) : mk3(q4.z, q4.w, q5.x);
    ls.gn = second ? mk3(q6.w, q7.x, q7.y) : mk3(q6.x, q6.y, q6.z);
    const float w = 1.0f - u - vv;
    ls.n = normalise3(mk3(u*nb.x + vv*na.x + w*n0.x, u*nb.y + vv*na.y + w*n0.y, u*nb.z + vv*na.z + w*n0.z));
    ls.flags = 0;
    const float ec[3] = { q7.z, q7.w, q8.x };
#pragma unroll
    for(int l=0;l<MI_MF;l++) lem[l] = q8.y*spectrum_eval(ec, lam[l]);
    lrough = q8.z;
    lpdf = q8.w;
  }
  else
  {
    const uint32_t t = sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
    lpe = sc.light_prim[t];
    const uint32_t lp = lpe & ~MI_LIGHT_ANYHIT;
    ls.x = prim_sample<MB>(sc.prims[lp], sc.primgeo[lp], r2, r3, ls.u, ls.v, MB ? sc.prims_t1 + lp : nullptr, time);
    ol = sub3(ls.x, from);
    ldist = mi_sqrt(dot3(ol, ol));
    const double il = 1./(double)ldist;
    ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
    const uint4 lhead = *(const uint4 *)&sc.primgeo[lp];
    surface_setup<MB>(sc, lp, lhead, ol, scramble, ls, time);
    Shading lsh[4];
    run_prepare_ops_hero(sc, sc.materials[lhead.y], sc.materials[lhead.y].num_ops, ls, lam, lsh);
#pragma unroll
    for(int l=0;l<MI_MF;l++) lem[l] = lsh[l].em;
    lrough = lsh[0].roughness;
    lpdf = sc.light_L[t];
  }
}
/* the emitter's edf towards -ol over the sampling pdf, per component (nee.h:92-130 with lights_sample_next_event's tail); lpdf becomes the vertex pdf */
__device__ __forceinline__ void hero_emitter_edf(const DScene &sc, const Surf &ls, const V3 ol, const float *lem, float lrough, float &lpdf, float *edf)
{
  double dir_term;
  if(lrough > 1.0f-1e-4f) dir_term = (double)1.0f/MI_PI_D;
  else
  {
    const float phongexp = 2.0f/(lrough*lrough) - 2.0f;
    dir_term = (double)(powf(-dot3(ls.gn, ol), phongexp)*(phongexp + 2.0f))/(2.0f*MI_PI_D);
  }
#pragma unroll
  for(int l=0;l<MI_MF;l++) edf[l] = (float)((double)(lem[l]/lpdf)*dir_term);
  lpdf = lpdf*sc.p_geo;
#pragma unroll
  for(int l=0;l<MI_MF;l++) edf[l] = edf[l]/sc.p_geo;
}

/* path_shade_volume (mi_path.h) for four components: the extension ray ended at the free-flight distance the HERO's medium sampled */
template<bool RECORD, bool PTDL, bool HALTON, bool MB, class CNT>
__device__ __forceinline__ void path_shade_volume_hero(const DScene &sc, PathStateHero &ps, mi_path_record *rec, unsigned long long slot, CNT &cnt)
{
  const int v = ps.length;
  bool alive = true;
  const V3 omega = ps.dir;
  const float dist = ps.clip;
  const float lam[4] = {ps.lambda, ps.lambda_x[0], ps.lambda_x[1], ps.lambda_x[2]};
  const float thr[4] = {ps.throughput, ps.throughput_x[0], ps.throughput_x[1], ps.throughput_x[2]};
  const float pdf_in[4] = {ps.pdf, ps.pdf_x[0], ps.pdf_x[1], ps.pdf_x[2]};
  double pp[4] = {ps.pdfprod, ps.pdfprod_x[0], ps.pdfprod_x[1], ps.pdfprod_x[2]};
  Medium med[4];
  hero_media(sc, ps, lam, med);
  Surf sf;
  const V3 rorg = ray_origin<PTDL>(ps, false);
  sf.x = mk3(rorg.x + dist*ps.dir.x, rorg.y + dist*ps.dir.y, rorg.z + dist*ps.dir.z);
  sf.n = omega; sf.gn = omega;
  get_scrambled_onb(ps.scramble, sf.n, sf.a, sf.b);
  sf.u = sf.v = sf.s = sf.t = 0.0f; sf.flags = 0;
  const uint32_t material_modes = s_volume | s_glossy;
  const float G = ps.prev_cos*1.0f/(dist*dist);
  float vpdf[4], vthr[4], epdf[4];
#pragma unroll
  for(int l=0;l<MI_MF;l++)
  { /* per component: transmittance exp(-d mu_t), pdf = transmittance * mu_t at the distance the hero sampled (src/shader.c:95-97) */
    const float eT = expf(-dist*med[l].mu_t);
    epdf[l] = eT*med[l].mu_t;
    vpdf[l] = (pdf_in[l]*epdf[l])*G;
    pp[l] *= (double)vpdf[l];
    vthr[l] = thr[l]*(eT/epdf[l]);
  }
  ps.length++;
  MI_COUNT(cnt, 6, 1);
  if(RECORD)
  {
    Shading z; z.roughness = z.rs = z.rd = z.rg = z.em = 0.0f;
    rec_vertex<RECORD>(rec, v, MI_PRIMID_INVALID, dist, sf.x, sf.n, sf.gn, omega, s_absorb, 0, vthr[0], vpdf[0], 0.0f, 0.0f, z, 0.0f, med[0].med);
    const Shading zz[4] = {z, z, z, z};
    hero_rec_vertex<RECORD>(sc, slot, v, vthr, vpdf, zz, nullptr);
    rec->length = ps.length; rec->throughput = (thr[0]*0.0f)/epdf[0];
  }
  {
    ps.prev_x = sf.x; ps.org_eps = 0.0f; ps.ignore = MI_NOPRIM;
    if(!PTDL) ps.org = sf.x;
    ps.prev_cos = 0.0f; ps.throughput = 0.0f; ps.pdf = 0.0f;
    ps.prev_material_modes = material_modes;
    if(PTDL)
    {
      ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0;
      ps.sh_value_x[0] = ps.sh_value_x[1] = ps.sh_value_x[2] = 0.0f;
    }
  }
  if(PTDL && ps.length >= (int)sc.max_verts) alive = false;
  if(PTDL && alive)
  { /* next event estimation from the volume vertex */
    (void)rng_next(ps.rng);
    PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_nee(v + 1));
    const float rnd = pts(MI_DIM_NEE_LIGHT1);
    if(!(rnd < sc.p_sky) && rnd < sc.p_sky + sc.p_geo)
    {
      const float r3 = pts(MI_DIM_NEE_Y);
      const float r2 = pts(MI_DIM_NEE_X);
      const float r1 = pts(MI_DIM_NEE_LIGHT2);
      uint32_t lpe;
      Surf ls;
      V3 ol;
      float ldist, lpdf, lem[4], lrough, edf[4];
      hero_sample_emitter<HALTON, true, MB>(sc, r1, r2, r3, lam, sf.x, ps.scramble, ps.time, lpe, ls, lem, lrough, lpdf, ldist, ol);
      hero_emitter_edf(sc, ls, ol, lem, lrough, lpdf, edf);
      if(edf[0] > 0.0f || edf[1] > 0.0f || edf[2] > 0.0f || edf[3] > 0.0f)
      {
        const float hg = eval_hg(med[0].g, omega, ol);
        float bsdf[4];
#pragma unroll
        for(int l=0;l<MI_MF;l++) bsdf[l] = med[l].mu_s*hg;               /* medium_rgb.c:98-102 */
        if(bsdf[0] > 0.0f || bsdf[1] > 0.0f || bsdf[2] > 0.0f || bsdf[3] > 0.0f)
        {
          const float eps = 1e-4f*DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)));
          V3 rd = sub3(ls.x, sf.x);
          rd = scale3(rd, mi_rcp(mi_sqrt(dot3(rd, rd))));
          const V3 ro = sf.x;
          const V3 dv = mk3(ls.x.x - eps*rd.x - ro.x, ls.x.y - eps*rd.y - ro.y, ls.x.z - eps*rd.z - ro.z);
          const float total_dist = mi_sqrt(dot3(dv, dv));
          if(!(dot3(ls.gn, rd) >= 0) && total_dist > 0.0f)
          {
            const float Gn = 1.0f*fabsf(dot3(ls.n, ol))/(ldist*ldist);
            float tn[4], ours[4], sums[4];
#pragma unroll
            for(int l=0;l<MI_MF;l++)
            {
              const float T = media_transmittance(med[l], ldist);
              float t = ((vthr[l]*bsdf[l])*(T*edf[l]))*Gn;
              t = t + (vthr[l]*bsdf[l])*((0.0f*Gn)/lpdf);
              const float wn = lpdf/(lpdf + 0.0f/T);
              tn[l] = t*wn;
              const float pe = (hero_pdf_to_surface(med[0], med[l], ldist)*hg)*Gn;
              const double our = (double)(1.0f*lpdf)*pp[l], other = (double)pe*pp[l];
              ours[l] = (float)our; sums[l] = (float)(other + our);
            }
            const float hs = hero_hsum(sums);
            if(tn[0]/1.0f > 0.0f || tn[1]/1.0f > 0.0f || tn[2]/1.0f > 0.0f || tn[3]/1.0f > 0.0f)
            {
              ps.sh_pending = 1;
              ps.prev_x = sf.x; ps.org_eps = 0.0f;
              ps.sh_dir = rd; ps.sh_dist = total_dist;
              ps.sh_light = lpe; ps.ignore = MI_NOPRIM;
              ps.sh_value = (tn[0]/1.0f)*(ours[0]/hs);
#pragma unroll
              for(int l=1;l<MI_MF;l++) ps.sh_value_x[l-1] = (tn[l]/1.0f)*(ours[l]/hs);
              ps.sh_length = ps.length + 1;
            }
          }
        }
      }
    }
  }
  if(alive && ps.length >= (int)sc.max_verts) alive = false;
  if(alive && !(vthr[0] > 0.0f || vthr[1] > 0.0f || vthr[2] > 0.0f || vthr[3] > 0.0f))
  {
    alive = false;
    if(RECORD && v < MI_REC_MAX_VERTS)
    {
      rec->v[v].throughput = 0.0f; rec->v[v].mode = s_absorb;
      if(sc.hero_ext) for(int l=0;l<MI_MF;l++) sc.hero_ext[slot].throughput[v][l] = 0.0f;
    }
  }
  if(alive)
  { /* the phase function has no wavelength in it: one direction, one pdf; the weight mu_s per component */
    PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v + 1));
    const float r2 = pts(MI_DIM_OMEGA_Y);          /* (medium_rgb.c compiles with gcc in the MF_COUNT = 4 build too: its order) */
    const float r1 = pts(MI_DIM_OMEGA_X);
    const float g = med[0].g;
    float o0, o1, o2, pdf;
    if(g == 0.0f)
    {
      const float z = 1.f - 2.f*r1;
      const float r = mi_sqrt(1.f - z*z);
      const float phi = (float)(2.f*MI_PI_D*(double)r2);
      float sn, cs;
      mi_sincosf(phi, &sn, &cs);
      o0 = r*cs; o1 = r*sn; o2 = z;
      pdf = (float)(1.0/(4.0*MI_PI_D));
    }
    else
    {
      const float sqr = (1.0f-g*g)/(1.0f+g*(2.0f*r1-1.0f));
      const float cos_theta = 1.0f/(2.0f*g)*(1.0f + g*g - sqr*sqr);
      const float phi = (float)(2.0f*MI_PI_D*(double)r2);
      const float l = mi_sqrt(fmaxf(0.0f, 1.0f-cos_theta*cos_theta));
      float sn, cs;
      mi_sincosf(phi, &sn, &cs);
      o0 = cos_theta; o1 = cs*l; o2 = sn*l;
      pdf = (float)(1.0/(4.0*MI_PI_D)*(double)(1.0f-g*g)/(double)powf(1.0f + g*g - 2.0f*g*cos_theta, 3.0f/2.0f));
    }
    V3 wo = mk3(sf.n.x*o0 + sf.a.x*o1 + sf.b.x*o2, sf.n.y*o0 + sf.a.y*o1 + sf.b.y*o2, sf.n.z*o0 + sf.a.z*o1 + sf.b.z*o2);
    wo = normalise3(wo);
    float nthr[4];
#pragma unroll
    for(int l=0;l<MI_MF;l++) nthr[l] = vthr[l]*med[l].mu_s;
    const uint32_t vmode = s_glossy | s_volume;
    if(nthr[0] <= 0.0f && nthr[1] <= 0.0f && nthr[2] <= 0.0f && nthr[3] <= 0.0f) alive = false;       /* mf_all(throughput <= 0), src/pathspace.c:253 */
    if(RECORD && v < MI_REC_MAX_VERTS) rec->v[v].mode = alive ? vmode : (uint32_t)s_absorb;
    if(alive)
    {
      ps.org = sf.x;
      ps.dir = wo;
      ps.ignore = MI_NOPRIM;
      ps.prev_x = sf.x; ps.org_eps = 0.0f;
      ps.prev_cos = 1.0f;
      ps.prev_throughput = vthr[0];
      ps.prev_mode = vmode;
      ps.prev_material_modes = material_modes;
      ps.throughput = nthr[0]; ps.pdf = pdf;
#pragma unroll
      for(int l=1;l<MI_MF;l++) { ps.throughput_x[l-1] = nthr[l]; ps.pdf_x[l-1] = pdf; }
    }
  }
  ps.pdfprod = pp[0]; ps.pdfprod_x[0] = pp[1]; ps.pdfprod_x[1] = pp[2]; ps.pdfprod_x[2] = pp[3];
  if(!alive) { ps.active = 0; cnt.c[4]++; }
}

/* path_shade (mi_path.h) for four components; the comments there name the reference lines of every step, here only what differs */
template<bool RECORD, bool PTDL, bool HALTON, bool MEDIA = false, bool MB = false, class CNT>
__device__ __forceinline__ void path_shade_hero(const DScene &sc, PathStateHero &ps, const Hit &hit, const uint32_t *shape_material, const float *shape_L,
                                                mi_path_record *rec, unsigned long long slot, CNT &cnt, SplatReq &splat)
{
  if(MEDIA && hit.prim == MI_NOPRIM && ps.clip < FLT_MAX)
  {
    path_shade_volume_hero<RECORD, PTDL, HALTON, MB>(sc, ps, rec, slot, cnt);
    return;
  }
  const int v = ps.length;
  bool alive = true;
  const V3 omega = ps.dir;
  const float lam[4] = {ps.lambda, ps.lambda_x[0], ps.lambda_x[1], ps.lambda_x[2]};
  float thr[4] = {ps.throughput, ps.throughput_x[0], ps.throughput_x[1], ps.throughput_x[2]};
  const float pdf_in[4] = {ps.pdf, ps.pdf_x[0], ps.pdf_x[1], ps.pdf_x[2]};
  float ior[4] = {ps.cur_ior, 1.0f, 1.0f, 1.0f};                /* e[v].vol.ior per component: see PathStateHero (filled in where a material asks for it) */
  double pp[4] = {ps.pdfprod, ps.pdfprod_x[0], ps.pdfprod_x[1], ps.pdfprod_x[2]};
  Medium med[4];                                                 /* the medium of the edge that ends here (extended kernels) */
  if(MEDIA) hero_media(sc, ps, lam, med);
  if(PTDL)
  {
    ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0;
    ps.sh_value_x[0] = ps.sh_value_x[1] = ps.sh_value_x[2] = 0.0f;
  }
  if(hit.prim == MI_NOPRIM)
  { /* environment vertex; black sky: the path ends */
    const float G = ps.prev_cos;
    float vpdf[4];
#pragma unroll
    for(int l=0;l<MI_MF;l++) { vpdf[l] = MEDIA ? (pdf_in[l]*1.0f)*G : pdf_in[l]*G; pp[l] *= (double)vpdf[l]; }
    ps.length++;
    MI_COUNT(cnt, 6, 1);
    if(RECORD)
    {
      const V3 x = mk3(ps.prev_x.x + sc.far_dist*omega.x, ps.prev_x.y + sc.far_dist*omega.y, ps.prev_x.z + sc.far_dist*omega.z);
      Shading z; z.roughness = 1.0f; z.rs = z.rd = z.rg = z.em = 0.0f;
      float ethr[4];
#pragma unroll
      for(int l=0;l<MI_MF;l++) ethr[l] = MEDIA ? thr[l]*(media_transmittance(med[l], FLT_MAX)/1.0f) : thr[l];
      rec_vertex<RECORD>(rec, v, MI_PRIMID_INVALID, FLT_MAX, x, mk3(0, 0, 0), mk3(0, 0, 0), omega, s_absorb, s_environment, ethr[0], vpdf[0], 0.0f, 0.0f, z, 0.0f, -1);
      const Shading zz[4] = {z, z, z, z};
      hero_rec_vertex<RECORD>(sc, slot, v, ethr, vpdf, zz, nullptr);
      rec->length = ps.length; rec->throughput = 0.0f;
    }
    alive = false;
  }
  else
  {
    Surf sf;
    const V3 rorg = ray_origin<PTDL>(ps, false);
    sf.x = mk3(rorg.x + hit.dist*ps.dir.x, rorg.y + hit.dist*ps.dir.y, rorg.z + hit.dist*ps.dir.z);
    sf.u = hit.u; sf.v = hit.v;
    const DPrimGeo &pshade = sc.primgeo[hit.prim];
    const uint4 head = *(const uint4 *)&pshade;
    const DMaterial &mat = sc.materials[head.y];
    const uint4 mhead = *(const uint4 *)&mat;
    const uint32_t mat_bsdf = mhead.x;
    const float mat_p0 = __uint_as_float(mhead.z), mat_p1 = __uint_as_float(mhead.w);
    surface_setup<MB>(sc, hit.prim, head, omega, ps.scramble, sf, ps.time);
    MI_PHASE(cnt, 2)
    const uint32_t shape = (head.w >> 3) & 0x1fffffffu;
    Shading sh[4];
    run_prepare_ops_hero(sc, mat, mhead.y, sf, lam, sh);
    uint32_t material_modes = 0;
    float eta[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    bool any_im = false;         /* mf_any(indexmatched(eta_ratio, 1)), dielectric.c:61-65 */
    if(RECORD || mat_bsdf != MI_BSDF_DIFFUSE)
    {
      {
        const int top_in = media_top_shape(ps.media);
#pragma unroll
        for(int l=1;l<MI_MF;l++) ior[l] = shape_interior_ior(sc, shape_material, top_in, lam[l]);
      }
      Media hyp = ps.media;
      media_apply(hyp, shape, (sf.flags & s_inside) != 0);
      if(hyp.broken) { eta[0] = eta[1] = eta[2] = eta[3] = -1.0f; }
      else
      {
        const int top = media_top_shape(hyp);
#pragma unroll
        for(int l=0;l<MI_MF;l++)
        {
          float interior_self = 1.0f;
          if(mat_bsdf == MI_BSDF_DIELECTRIC) interior_self = eta_from_abbe(mat_p0, mat_p1, lam[l]);
          const float ior2 = (top == (int)shape) ? interior_self : shape_interior_ior(sc, shape_material, top, lam[l]);
          eta[l] = ior[l]/ior2;
        }
      }
    }
    const bool any_rd = sh[0].rd > 0.0f || sh[1].rd > 0.0f || sh[2].rd > 0.0f || sh[3].rd > 0.0f;
    if(mat_bsdf == MI_BSDF_DIFFUSE) { if(any_rd) material_modes = s_reflect | s_diffuse; }           /* mf_any(rd > 0), src/shader.c:161 */
    else if(mat_bsdf == MI_BSDF_DIELECTRIC)
    {
      material_modes = s_reflect | s_transmit;
      any_im = fabsf(1.0f - eta[0]/1.0f) < 1e-3f || fabsf(1.0f - eta[1]/1.0f) < 1e-3f || fabsf(1.0f - eta[2]/1.0f) < 1e-3f || fabsf(1.0f - eta[3]/1.0f) < 1e-3f;
      if(any_im) sh[0].roughness = sh[1].roughness = sh[2].roughness = sh[3].roughness = 0.0f;       /* indexmatched() is an mf_any, dielectric.c:61-65 */
      if(sh[0].roughness > GLOSSY_THR) material_modes |= s_glossy; else material_modes |= s_specular;
    }
    else if(mat_bsdf == MI_BSDF_METAL)
    {
      material_modes = s_reflect;
      if(sh[0].roughness > 1e-4f) material_modes |= s_glossy; else material_modes |= s_specular;
    }

    MI_PHASE(cnt, 3)
    const uint32_t type = MB ? head.x & 7u : head.x;                /* the motion-blur kernels flag moving primitives in bit 3 (MI_GEO_MB) */
    if((type > 2 || hit.dist < 1e-4f) && hit.prim == ps.ignore)
    { /* self-intersection */
      alive = false;
      if(RECORD)
      {
        rec->length = ps.length; rec->throughput = 0.0f;
        if(v >= 1 && v-1 < MI_REC_MAX_VERTS && !(ps.prev_mode & s_emit)) rec->v[v-1].mode = s_absorb;
      }
    }
    else
    {
      uint32_t mode = s_absorb;
      const bool any_em = sh[0].em > 0.0f || sh[1].em > 0.0f || sh[2].em > 0.0f || sh[3].em > 0.0f;     /* mf_any(em > 0), src/pathspace.c:876-877 */
      if(any_em && !(sf.flags & s_inside)) { mode = s_emit; material_modes = s_emit; }
      const float G = ps.prev_cos*fabsf(dot3(sf.n, omega))/(hit.dist*hit.dist);
      float vpdf[4], eT[4], epdf[4];
      double pp_before[4];
#pragma unroll
      for(int l=0;l<MI_MF;l++)
      { /* edge in a medium: transmittance per component; whether the edge has a pdf is the hero's question (hero_pdf_to_surface) */
        eT[l] = MEDIA ? media_transmittance(med[l], hit.dist) : 1.0f;
        epdf[l] = MEDIA ? hero_pdf_to_surface(med[0], med[l], hit.dist) : 1.0f;
        vpdf[l] = MEDIA ? (pdf_in[l]*epdf[l])*G : pdf_in[l]*G;
        pp_before[l] = pp[l]; pp[l] *= (double)vpdf[l];
      }
      ps.length++;
      MI_COUNT(cnt, 6, 1);
      float path_throughput[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if(mode & s_emit)
      { /* lights_eval_vertex: mf_all(em <= 0) cannot hold here; one edf, the emission per component */
        float edf = 0.0f;
        const bool facing = !(dot3(sf.gn, omega) >= 0.0);
        if(facing)
        {
          if(sh[0].roughness > 1.0f-1e-4f) edf = (float)(1.0f/MI_PI_D);
          else
          {
            const float phongexp = 2.0f/(sh[0].roughness*sh[0].roughness) - 2.0f;
            edf = (float)((double)(powf(fabsf(dot3(sf.gn, omega)), phongexp)*(phongexp+2.0f))/(2.0f*MI_PI_D));
          }
        }
#pragma unroll
        for(int l=0;l<MI_MF;l++)
        {
          const float Le = facing ? edf*sh[l].em : 0.0f;
          path_throughput[l] = MEDIA ? (thr[l]*0.0f)/epdf[l] + (thr[l]*(eT[l]/epdf[l]))*Le : 0.0f + thr[l]*Le;
        }
      }
      float vthr[4];
#pragma unroll
      for(int l=0;l<MI_MF;l++) vthr[l] = MEDIA ? thr[l]*(eT[l]/epdf[l]) : thr[l];
      if(RECORD)
      {
        rec_vertex<RECORD>(rec, v, MI_GEO_PRIMID(pshade), hit.dist, sf.x, sf.n, sf.gn, omega, mode, sf.flags, vthr[0], vpdf[0], sf.u, sf.v, sh[0], eta[0], (int)head.y);
        hero_rec_vertex<RECORD>(sc, slot, v, vthr, vpdf, sh, eta);
        rec->length = ps.length; rec->throughput = path_throughput[0];
      }
      if(mode & s_emit)
      {
        float w[4], value[4];
        if(PTDL)
        { /* sampler_mis, ptdl.c:78-88: the balance heuristic over techniques AND wavelengths -- md_2f(our) / mf_hsum(md_2f(other + our)) */
          float nee = 0.0f;
          if(ps.length >= 3 && (ps.prev_material_modes & (s_diffuse | s_glossy)) && sc.p_geo > 0) nee = sc.p_geo*shape_L[shape];
          float ours[4], sums[4];
#pragma unroll
          for(int l=0;l<MI_MF;l++)
          {
            const double our = (double)vpdf[l]*pp_before[l], other = (double)(1.0f*nee)*pp_before[l];
            ours[l] = (float)our; sums[l] = (float)(other + our);
          }
          const float hs = hero_hsum(sums);
#pragma unroll
          for(int l=0;l<MI_MF;l++) w[l] = ours[l]/hs;
        }
        else
        { /* sampler_mis, pt.c:30-38 */
          float fp[4];
#pragma unroll
          for(int l=0;l<MI_MF;l++) fp[l] = (float)pp[l];
          const float hs = hero_hsum(fp);
#pragma unroll
          for(int l=0;l<MI_MF;l++) w[l] = fp[l]/hs;
        }
#pragma unroll
        for(int l=0;l<MI_MF;l++) value[l] = PTDL ? path_throughput[l]*w[l] : w[l]*path_throughput[l];
        float col[3];
        const bool ok = hero_splat_colour(sc, lam, value, col);
        if(RECORD && rec->num_splats < MI_REC_MAX_SPLATS)
        {
          if(sc.hero_ext) for(int l=0;l<MI_MF;l++) sc.hero_ext[slot].splat_value[rec->num_splats][l] = value[l];
          mi_path_splat &sp = rec->splat[rec->num_splats++];
          sp.length = ps.length; sp.tech = s_tech_extend; sp.value = value[0];
          sp.col[0] = col[0]; sp.col[1] = col[1]; sp.col[2] = col[2];
        }
        if(ok)
        {
          MI_COUNT(cnt, 5, 1);
          if(!RECORD) { splat.pending = true; splat.c0 += col[0]; splat.c1 += col[1]; splat.c2 += col[2]; }
        }
        if(!PTDL && ps.length > 3)
        { /* path_russian_roulette on the hero's throughputs (pt.c:50: mf(throughput, 0)); all four are scaled */
          const float p_survival = DMIN(1.0f, vthr[0]/ps.prev_throughput);
          PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v));
          const float rr = pts(MI_DIM_RUSSIAN_R);
          const float scale = rr >= p_survival ? mi_rcp(1.0f-p_survival) : mi_rcp(p_survival);
          if(rr >= p_survival) alive = false;
#pragma unroll
          for(int l=0;l<MI_MF;l++) vthr[l] = vthr[l]*scale;
          if(RECORD && v < MI_REC_MAX_VERTS)
          {
            float rp[4];
#pragma unroll
            for(int l=0;l<MI_MF;l++) rp[l] = alive ? vpdf[l]*p_survival : vpdf[l]*(1.0f-p_survival);
            rec->v[v].throughput = vthr[0]; rec->v[v].pdf = rp[0];
            hero_rec_vertex<RECORD>(sc, slot, v, vthr, rp, sh, eta);
          }
        }
      }
      {
        const float eps0 = DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)))*1e-4f;
        ps.prev_x = sf.x; ps.org_eps = eps0; ps.ignore = hit.prim;
        if(!PTDL) ps.org = sf.x;
        ps.prev_cos = 0.0f; ps.throughput = 0.0f; ps.pdf = 0.0f;
        ps.prev_material_modes = material_modes;
      }
      if(PTDL && ps.length >= (int)sc.max_verts) alive = false;
      if(PTDL && alive)
      { /* next event estimation */
        (void)rng_next(ps.rng);
        if(material_modes & (s_diffuse | s_glossy))
        {
          PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_nee(v + 1));
          const float rnd = pts(MI_DIM_NEE_LIGHT1);
          if(!(rnd < sc.p_sky) && rnd < sc.p_sky + sc.p_geo)
          {
            const float r3 = pts(MI_DIM_NEE_Y);
            const float r2 = pts(MI_DIM_NEE_X);
            const float r1 = pts(MI_DIM_NEE_LIGHT2);
            uint32_t lpe;
            Surf ls;
            float lem[4], lrough;
            float lpdf, ldist;
            V3 ol;
            hero_sample_emitter<HALTON, MEDIA, MB>(sc, r1, r2, r3, lam, sf.x, ps.scramble, ps.time, lpe, ls, lem, lrough, lpdf, ldist, ol);
            float edf[4];
            hero_emitter_edf(sc, ls, ol, lem, lrough, lpdf, edf);
            if(edf[0] > 0.0f || edf[1] > 0.0f || edf[2] > 0.0f || edf[3] > 0.0f)                       /* mf_any(edf > 0), nee.h:188 */
            {
              HeroEval he;
              if(mat_bsdf == MI_BSDF_DIFFUSE) he = brdf_diffuse_hero(sf, sh, ol);
              else if(mat_bsdf == MI_BSDF_DIELECTRIC) he = brdf_dielectric_hero(sf, sh, omega, ol, eta, any_im);
              else he = brdf_metal_hero(sc, sf, sh, omega, ol, ior, (int)mat_p0, lam);
              bool okn = he.value[0] > 0.0f || he.value[1] > 0.0f || he.value[2] > 0.0f || he.value[3] > 0.0f;   /* mf_any(bsdf > 0), nee.h:191 */
              Medium nmed[4];                                              /* volume of the connection edge, per component */
              if(MEDIA) { nmed[0] = med[0]; nmed[1] = med[1]; nmed[2] = med[2]; nmed[3] = med[3]; }
              if(okn && (he.mode & s_transmit))
              {
                Media hyp = ps.media;
                media_apply(hyp, shape, (sf.flags & s_inside) != 0);
                if(hyp.broken) okn = false;
                else if(MEDIA)
                {
                  const int ctop = media_top_shape(hyp);
#pragma unroll
                  for(int l=0;l<MI_MF;l++) nmed[l] = shape_interior_medium(sc, ctop, lam[l]);
                }
              }
              if(okn)
              {
                const float eps = 1e-4f*DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)));
                V3 rd = sub3(ls.x, sf.x);
                rd = scale3(rd, mi_rcp(mi_sqrt(dot3(rd, rd))));
                const V3 ro = mk3(sf.x.x + eps*rd.x, sf.x.y + eps*rd.y, sf.x.z + eps*rd.z);
                const V3 dv = mk3(ls.x.x - eps*rd.x - ro.x, ls.x.y - eps*rd.y - ro.y, ls.x.z - eps*rd.z - ro.z);
                const float total_dist = mi_sqrt(dot3(dv, dv));
                if(!(dot3(ls.gn, rd) >= 0) && total_dist > 0.0f)
                {
                  const float Gn = fabsf(dot3(sf.n, ol))*fabsf(dot3(ls.n, ol))/(ldist*ldist);
                  float tn[4], ours[4], sums[4];
                  float pbs[4];
                  if(mat_bsdf == MI_BSDF_DIFFUSE) pbs[0] = pbs[1] = pbs[2] = pbs[3] = (float)(1.0f/MI_PI_D);
                  else if(mat_bsdf == MI_BSDF_DIELECTRIC) pdf_dielectric_hero(sf, sh, omega, ol, eta, any_im, he.mode, pbs);
                  else pbs[0] = pbs[1] = pbs[2] = pbs[3] = pdf_metal(sf, sh[0], omega, ol, he.mode);     /* no wavelength in it */
#pragma unroll
                  for(int l=0;l<MI_MF;l++)
                  {
                    const float nT = MEDIA ? media_transmittance(nmed[l], ldist) : 1.0f;
                    float t = ((vthr[l]*he.value[l])*(nT*edf[l]))*Gn;
                    t = t + (vthr[l]*he.value[l])*((0.0f*Gn)/lpdf);
                    const float wn = lpdf/(lpdf + 0.0f/nT);
                    tn[l] = t*wn;
                    const float pb = pbs[l];
                    const float pe = ((MEDIA ? hero_pdf_to_surface(nmed[0], nmed[l], ldist) : 1.0f)*pb)*Gn;
                    const double our = (double)(1.0f*lpdf)*pp[l], other = (double)pe*pp[l];
                    ours[l] = (float)our; sums[l] = (float)(other + our);
                  }
                  const float hs = hero_hsum(sums);
                  if(tn[0]/1.0f > 0.0f || tn[1]/1.0f > 0.0f || tn[2]/1.0f > 0.0f || tn[3]/1.0f > 0.0f)   /* mf_any(throughput > 0), ptdl.c:142 */
                  {
                    ps.sh_pending = 1;
                    ps.prev_x = sf.x; ps.org_eps = eps;
                    ps.sh_dir = rd; ps.sh_dist = total_dist;
                    ps.sh_light = lpe; ps.ignore = hit.prim;
                    ps.sh_value = (tn[0]/1.0f)*(ours[0]/hs);
#pragma unroll
                    for(int l=1;l<MI_MF;l++) ps.sh_value_x[l-1] = (tn[l]/1.0f)*(ours[l]/hs);
                    ps.sh_length = ps.length + 1;
                  }
                }
              }
            }
          }
        }
      }
      MI_PHASE(cnt, 4)
      if(alive && ps.length >= (int)sc.max_verts) alive = false;
      if(alive && !(vthr[0] > 0.0f || vthr[1] > 0.0f || vthr[2] > 0.0f || vthr[3] > 0.0f))             /* !mf_any(throughput > 0), src/pathspace.c:189 */
      {
        alive = false;
        if(RECORD && v < MI_REC_MAX_VERTS)
        {
          rec->v[v].throughput = 0.0f; rec->v[v].mode = s_absorb;
          if(sc.hero_ext) for(int l=0;l<MI_MF;l++) sc.hero_ext[slot].throughput[v][l] = 0.0f;
        }
      }
      if(alive)
      {
        get_scrambled_onb(ps.scramble, sf.n, sf.a, sf.b);
        PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v + 1));
        HeroSample hs;
        if(mat_bsdf == MI_BSDF_DIFFUSE) sample_diffuse_hero(pts, sf, sh, any_rd, mode, hs);
        else if(mat_bsdf == MI_BSDF_DIELECTRIC) sample_dielectric_hero(pts, sf, sh, omega, eta, any_im, mode, hs);
        else sample_metal_hero(sc, pts, sf, sh, omega, ior, (int)mat_p0, lam, mode, hs);
        MI_PHASE(cnt, 7)
        const V3 out = normalise3(hs.omega);
        const float dts = ((sf.flags & s_inside) ? -1 : 1)*dot3(sf.gn, out);
        const bool wrong_side = ((hs.mode & s_reflect) && (dts < 0.f)) || ((hs.mode & s_transmit) && (dts > 0.f));
        float nthr[4];
#pragma unroll
        for(int l=0;l<MI_MF;l++) nthr[l] = vthr[l]*(wrong_side ? 0.0f : hs.weight[l]);
        uint32_t vmode = hs.mode;
        bool ok = !(nthr[0] <= 0.0f && nthr[1] <= 0.0f && nthr[2] <= 0.0f && nthr[3] <= 0.0f);         /* mf_all(throughput <= 0), src/pathspace.c:253 */
        if(ok && (vmode & s_transmit))
        {
          media_apply(ps.media, shape, (sf.flags & s_inside) != 0);
          if(ps.media.broken) ok = false;
          else
          {
            const int top = media_top_shape(ps.media);
            ior[0] = shape_interior_ior(sc, shape_material, top, lam[0]);
            if(MEDIA) ps.cur = shape_interior_medium(sc, top, lam[0]);
          }
        }
        if(!ok)
        {
          alive = false;
          if(!(vmode & s_emit)) vmode = s_absorb;
        }
        if(RECORD && v < MI_REC_MAX_VERTS) rec->v[v].mode = vmode;
        if(alive)
        {
          const float eps = DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)))*1e-4f;
          ps.org = mk3(sf.x.x + eps*out.x, sf.x.y + eps*out.y, sf.x.z + eps*out.z);
          ps.dir = out;
          ps.ignore = hit.prim;
          ps.prev_x = sf.x; ps.org_eps = eps;
          ps.prev_cos = fabsf(dot3(sf.n, out));
          ps.prev_throughput = vthr[0];
          ps.prev_mode = vmode;
          ps.prev_material_modes = material_modes;
          ps.throughput = nthr[0]; ps.pdf = hs.pdf[0]; ps.cur_ior = ior[0];
#pragma unroll
          for(int l=1;l<MI_MF;l++) { ps.throughput_x[l-1] = nthr[l]; ps.pdf_x[l-1] = hs.pdf[l]; }
        }
      }
    }
  }
  ps.pdfprod = pp[0]; ps.pdfprod_x[0] = pp[1]; ps.pdfprod_x[1] = pp[2]; ps.pdfprod_x[2] = pp[3];
  if(!alive) { ps.active = 0; cnt.c[4]++; }
}

#endif
