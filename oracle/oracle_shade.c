/* oracle_shade.c -- TEST INFRASTRUCTURE: CPU restatement of the shading side of the reference hot
 * path (see oracle.h): shader dispatch, hit-point set-up, builtin diffuse, colour sources, rough
 * dielectric (GGX VNDF), nested-dielectric medium stack.
 */
#include "o_core.h"

/* ---------------------------------------------------------------- spectra */
/* test switch: evaluate the sigmoid the way the reference BUILD does on this host -- fused multiply-adds (its arch flags define
 * __FMA__) and the hardware's 12-bit rsqrtss (include/rgb2spec.h:130-149). Off by default: the HIP path is compared against the
 * exact form below. Golden tests turn it on where the approximation would otherwise dominate the comparison with the
 * reference dumps (thin media: mu_t is a small sigmoid value, the approximation is worth 0.1-0.4 % there). */
static int o_reference_rsqrt = 0;
#if defined(__x86_64__)
#include <xmmintrin.h>
float oracle_rsqrtss(float x) { return _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(x))); }
int oracle_set_reference_rsqrt(int on) { o_reference_rsqrt = on; return 1; }
#else
float oracle_rsqrtss(float x) { return 1.0f/sqrtf(x); }
int oracle_set_reference_rsqrt(int on) { (void)on; return 0; }
#endif

float o_spectrum_eval(const float coeff[3], float lambda)
{ /* rgb2spec_eval_fast, include/rgb2spec.h:145-149 -- the reference uses the 12-bit rsqrtss
     approximation here; we use the exact reciprocal square root (tolerance stated in the tests) */
  if(o_reference_rsqrt)
  {
    const float xr = fmaf(fmaf(coeff[0], lambda, coeff[1]), lambda, coeff[2]);
    const float yr = oracle_rsqrtss(fmaf(xr, xr, 1.f));
    return fmaf(.5f*xr, yr, .5f);
  }
  const float x = (coeff[0]*lambda + coeff[1])*lambda + coeff[2];
  const float y = 1.0f/sqrtf(x*x + 1.0f);
  return .5f*x*y + .5f;
}

static float o_eta_from_abbe(float n_d, float V_d, float lambda)
{ /* spectrum_cauchy_from_abbe + spectrum_eta_from_abbe, include/spectrum.h:40-63 */
  float A, B;
  if(V_d == 0.0f) { A = n_d; B = 0.0f; }
  else
  {
    const float l_C = .6563f, l_F = .4861f, l_D = .587561f;
    const float c = (l_C*l_C * l_F*l_F)/(l_C*l_C - l_F*l_F);
    B = (n_d - 1.0f)/V_d * c;
    A = n_d - B/(l_D*l_D);
  }
  return A + (B*1e6f)/(lambda*lambda);
}

/* ---------------------------------------------------------------- nested dielectrics */
static int o_edge_medium(const o_path *path, int e, int eta_ratio)
{ /* _path_edge_medium, src/pathspace.c:80-115 */
  int stack_v[O_MAX_VERTS] = {0};
  int sp = 1;
  for(int k=1;k<e;k++)
  {
    if(((k == e-1) && eta_ratio) || (path->v[k].mode & s_transmit))
    {
      if(!(path->v[k].flags & s_inside)) stack_v[sp++] = k;
      else
      {
        if(sp == 0) return -1;
        for(int m=sp-1;m>=0;m--)
        {
          if(MI_PRIMID_SHAPE(path->v[stack_v[m]].hit.prim) == MI_PRIMID_SHAPE(path->v[k].hit.prim))
          {
            stack_v[m] = stack_v[--sp];
            break;
          }
          if(m == 0) return -1;
        }
      }
    }
  }
  int result = 0;
  for(int i=1;i<sp;i++)
    if(MI_PRIMID_SHAPE(path->v[stack_v[i]].hit.prim) < MI_PRIMID_SHAPE(path->v[result].hit.prim))
      result = stack_v[i];
  return result;
}

float o_path_eta_ratio(const o_path *path, int v)
{ /* path_eta_ratio, src/pathspace.c:117-124 */
  if(path->v[v].hit.prim == MI_PRIMID_INVALID) return 1.0f;
  const int mv = o_edge_medium(path, v+1, 1);
  if(mv < 0) return -1.0f;
  return path->e[v].vol.ior / path->v[mv].interior.ior;
}

int o_path_edge_init_volume(o_path *path, int v)
{ /* path_edge_init_volume, src/pathspace.c:127-146 */
  if(!(path->v[v-1].mode & s_transmit)) path->e[v].vol = path->e[v-1].vol;
  else
  {
    const int mv = o_edge_medium(path, v, 0);
    if(mv < 0) return 1;
    path->e[v].vol = path->v[mv].interior;
  }
  return 0;
}

/* ---------------------------------------------------------------- prepare */
static void o_set_slot(o_path *p, int v, uint32_t slot, float val)
{ /* tex_set_slot, src/shaders/texture.h:34-66 */
  switch(slot)
  {
    case MI_SLOT_DIFFUSE:   p->v[v].shading.rd = val; return;
    case MI_SLOT_SPECULAR:  p->v[v].shading.rs = val; return;
    case MI_SLOT_GLOSSY:    p->v[v].shading.rg = val; return;
    case MI_SLOT_ROUGHNESS: p->v[v].shading.roughness = val; return;
    case MI_SLOT_EMISSION:  p->v[v].shading.em = val; return;
    case MI_SLOT_VOLUME:    /* homogeneous interior media get their lobe modes here, texture.h:48-53 */
      p->v[v].material_modes = s_volume | s_glossy;
      p->v[v].interior.mu_s = val;
      p->v[v].interior.mu_t = 1.0f;
      return;
    default: return;
  }
}

static void o_prepare_op(const mi_scene_desc *s, const mi_shade_op *op, o_path *p, int v)
{
  if(op->kind == MI_OP_COLOR)
  { /* color.c:75-82 + tex_set_slot_coeff, texture.h:69-84 */
    p->v[v].shading.roughness = op->roughness;
    const float val = op->mul * o_spectrum_eval(op->coeff, p->lambda);
    if(op->slot == MI_SLOT_EMISSION) o_set_slot(p, v, op->slot, val);
    else if(op->slot != MI_SLOT_UNUSED) o_set_slot(p, v, op->slot, OCLAMP(val, 0.0f, 1.0f));
  }
  else if(op->kind == MI_OP_CHECKER)
  { /* colorcheckersg.c:195-206,244-262 */
    const float u = p->v[v].hit.s, t = p->v[v].hit.t;
    const int i = (int)(14.0f*u) % 14, j = (int)(10.0f*t) % 10;
    float val;
    if(fmodf(14.0f*u, 1.0f) < 0.1f || fmodf(14.0f*u, 1.0f) > 0.9f || fmodf(10.0f*t, 1.0f) < 0.1f || fmodf(10.0f*t, 1.0f) > 0.9f)
      val = 0.3f;
    else
    {
      const int l = (p->lambda - 380.0f)/10.0f;
      if(l < 0 || l >= 36) val = 0.0f;
      else val = s->checker[36*(14*j + i) + l];
    }
    o_set_slot(p, v, op->slot, val);
  }
}

void o_prepare_medium(const mi_scene_desc *s, o_path *p, int v, int medium)
{ /* the prepare chain of a medium shader on vertex v: `interior <surface> <medium>` runs it before the surface's
     (src/shaders/interior.c:101-118), shader_exterior_medium on the sensor (src/shader.c:552-564). mult.c:154-167: its colour
     op in the volume slot first (texture.h:48-53), then medium_rgb.c:45-59; the volume keeps the medium's shader id */
  const mi_material *med = s->materials + medium;
  for(uint32_t k=0;k<med->num_ops;k++) o_prepare_op(s, med->op + k, p, v);
  p->v[v].interior.mean_cos = med->mean_cos;
  const float old_mu_t = p->v[v].interior.mu_t;
  p->v[v].interior.mu_t = med->param[3]*o_spectrum_eval(med->param, p->lambda);
  p->v[v].interior.mu_s = p->v[v].interior.mu_s*(p->v[v].interior.mu_t/old_mu_t);
  p->v[v].interior.shader = medium;
}

float o_shader_prepare(o_ctx *c, o_path *p, int v)
{ /* shader_prepare, src/shader.c:462-542 */
  const mi_scene_desc *s = c->s;
  if(p->v[v].flags & s_environment)
  {
    memset(&p->v[v].shading, 0, sizeof(o_shading));
    p->v[v].shading.roughness = 1.0f;
    p->v[v].shading.em = 0.0f;                       /* sky_black, src/shader.c:267-270 */
    return 1.0f;
  }
  /* manifold_init, include/pathspace/manifold.h:215-232 (branch compiled without COMPUTE_MANIFOLD_STUFF) */
  o_hit *hit = &p->v[v].hit;
  if(hit->prim == MI_PRIMID_INVALID)
  {
    if(p->v[v].mode & s_sensor) return 1.0f;         /* manifold_init leaves the sensor alone, manifold.h:112-114 */
    /* volume scattering, manifold.h:236-246: the frame looks along the incoming direction */
    for(int k=0;k<3;k++) hit->n[k] = hit->gn[k] = p->e[v].omega[k];
    o_get_scrambled_onb(p->scramble, hit->n, hit->a, hit->b);
    if(p->e[v].vol.shader >= 0 && v > 0)
    { /* homogeneous medium, src/shader.c:478-501: the vertex takes the edge's volume, assume a glossy lobe */
      p->v[v].interior = p->e[v].vol;
      p->v[v].material_modes = s_volume | s_glossy;
    }
    return 1.0f;
  }
  o_prims_get_normal(s, hit->prim, hit, p->time);
  if(dot3(p->e[v].omega, hit->gn) > 0.0f)
  {
    for(int i=0;i<3;i++) hit->n[i] = -hit->n[i];
    p->v[v].flags |= s_inside;
  }
  else p->v[v].flags &= ~s_inside;
  o_get_scrambled_onb(p->scramble, hit->n, hit->a, hit->b);

  hit->shader = s->shapes[MI_PRIMID_SHAPE(hit->prim)].material;
  memset(&p->v[v].shading, 0, sizeof(o_shading));
  p->v[v].shading.roughness = 1.0f;
  memset(&p->v[v].interior, 0, sizeof(o_volume));    /* path_volume_vacuum */
  p->v[v].interior.ior = 1.0f; p->v[v].interior.shader = -1;

  const mi_material *m = s->materials + hit->shader;
  if(m->interior >= 0) o_prepare_medium(s, p, v, m->interior);
  for(uint32_t k=0;k<m->num_ops;k++) o_prepare_op(s, m->op + k, p, v);      /* mult.c:154-167 */
  if(m->bsdf == MI_BSDF_DIFFUSE)
  { /* prepare_d, src/shader.c:157-162 */
    if(o_g_any(O_CTX(p), p->v[v].shading.rd > 0.0f)) p->v[v].material_modes = s_reflect | s_diffuse;      /* mf_any */
  }
  else if(m->bsdf == MI_BSDF_DIELECTRIC)
  { /* dielectric.c:67-81 */
    p->v[v].interior.ior = o_eta_from_abbe(m->param[0], m->param[1], p->lambda);
    p->v[v].material_modes = s_reflect | s_transmit;
    const float eta = o_path_eta_ratio(p, v);
    if(o_g_any(O_CTX(p), fabsf(1.0f - eta) < 1e-3f)) p->v[v].shading.roughness = 0.0f;        /* indexmatched: mf_any over the wavelengths, dielectric.c:61-65 */
    if(p->v[v].shading.roughness > 1e-3f) p->v[v].material_modes |= s_glossy;
    else p->v[v].material_modes |= s_specular;
  }
  else if(m->bsdf == MI_BSDF_METAL)
  { /* metal.c:59-67 */
    p->v[v].material_modes = s_reflect;
    if(p->v[v].shading.roughness > 1e-4f) p->v[v].material_modes |= s_glossy;
    else p->v[v].material_modes |= s_specular;
  }
  p->v[v].eta = o_path_eta_ratio(p, v);
  return 1.0f;
}

/* ---------------------------------------------------------------- builtin diffuse */
static float o_sample_diffuse(o_ctx *c, o_path *p)
{ /* sample_d, src/shader.c:165-205 (path tracing direction) */
  const int v = p->length;
  const float x1 = o_point(c, p, v, o_dim_omega_x);
  const float x2 = o_point(c, p, v, o_dim_omega_y);
  const float sq = sqrtf(x1);
  const float *n = p->v[v-1].hit.n;
  for(int k=0;k<3;k++)
    p->e[v].omega[k] =
      sqrtf(1.0 - x1)      * n[k] +
      sq*cosf(2*M_PI*x2)   * p->v[v-1].hit.a[k] +
      sq*sinf(2*M_PI*x2)   * p->v[v-1].hit.b[k];
  p->v[v].pdf = 1.0f/M_PI;
  const float cos_out_ng = dot3(p->v[v-1].hit.gn, p->e[v].omega);
  if(p->v[v-1].flags & s_inside) { if(cos_out_ng >= 0.0f) return 0.0f; }
  else if(cos_out_ng <= 0.0f) return 0.0f;
  const float throughput = p->v[v-1].shading.rd;
  if(o_g_any(c, throughput > 0.0f)) p->v[v-1].mode = s_diffuse | s_reflect;       /* mf_any, src/shader.c:202 */
  return throughput;
}

static float o_brdf_diffuse(o_path *p, int v)
{ /* brdf_d, src/shader.c:207-252 (path tracing direction) */
  p->v[v].mode = s_diffuse | s_reflect;
  const float cos_out_ns = dot3(p->v[v].hit.n, p->e[v+1].omega);
  if(cos_out_ns <= 0) return 0.0f;
  const float cos_out_ng = dot3(p->v[v].hit.gn, p->e[v+1].omega);
  if(p->v[v].flags & s_inside) { if(cos_out_ng >= 0.0f) return 0.0f; }
  else if(cos_out_ng <= 0.0f) return 0.0f;
  return p->v[v].shading.rd * (1.0f/M_PI);
}

/* ---------------------------------------------------------------- GGX, src/shaders/ggx.h */
static float o_ggx_G1(const float *w, const float *n, float roughness)
{ /* ggx_shadowing_smith_G1, ggx.h:29-36 */
  const float r2 = roughness*roughness;
  const float cos_th = fabsf(dot3(w, n));
  const float sin_th = sqrtf(fmaxf(0.0f, 1.0f - cos_th*cos_th));
  const float tan_th = sin_th/cos_th;
  return 2.0f/(1.0f + sqrtf(1.0f + r2*tan_th*tan_th));
}

static float o_ggx_G1_cos(float cos_wn, float roughness)
{ /* ggx_shadowing_smith_G1_mf, ggx.h:38-46 */
  const float r2 = roughness*roughness;
  const float sin_wn = sqrtf(OCLAMP(1.0f - cos_wn*cos_wn, 0.0f, 1.0f));
  const float tan_th = sin_wn/cos_wn;
  return 2.0f/(1.0f + sqrtf(1.0f + r2*tan_th*tan_th));
}

static void o_ggx_sample11(float tan_theta_i, float U1, float U2, float *slope_x, float *slope_y)
{ /* _ggx_sample11, ggx.h:59-110 */
  if(tan_theta_i < 0.0001f)
  {
    const float r = sqrtf(U1/fmaxf(1e-8f, 1-U1));
    const float phi = 2.0f*M_PI*U2;
    *slope_x = r*cosf(phi);
    *slope_y = r*sinf(phi);
    return;
  }
  const float a = 1.0f/tan_theta_i;
  const float G1 = 2.0f/(1.0f + sqrtf(1.0f + 1.0f/(a*a)));
  const float A = 2.0f*U1/G1 - 1.0f;
  const float tmp = 1.0f/(A*A - 1.0f);
  const float B = tan_theta_i;
  const float D = sqrtf(fmaxf(0.0f, B*B*tmp*tmp - (A*A - B*B)*tmp));
  float sx1 = B*tmp - D, sx2 = B*tmp + D;
  if(!(fabsf(sx1) < FLT_MAX)) sx1 = 0.0f;
  if(!(fabsf(sx2) < FLT_MAX)) sx2 = 0.0f;
  *slope_x = (A < 0.0f || sx2*tan_theta_i > 1.0f) ? sx1 : sx2;
  float S;
  if(U2 > 0.5f) { S = 1.0f;  U2 = 2.0f*(U2 - 0.5f); }
  else          { S = -1.0f; U2 = 2.0f*(0.5f - U2); }
  const float z = (U2*(U2*(U2*(-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (U2*(U2*(U2*(U2*0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
  *slope_y = S*z*sqrtf(1.0 + *slope_x * *slope_x);
}

static void o_ggx_sample_h(const float *wi, float rx, float ry, float U1, float U2, float *h)
{ /* ggx_sample_h, ggx.h:115-162 */
  float wi_[3] = { rx*wi[0], ry*wi[1], fabsf(wi[2]) };
  o_normalise(wi_);
  float tan_theta = 0.0f, sin_phi = 0.0f, cos_phi = 1.0f;
  if(wi_[2] < 0.99999)
  {
    const float len = sqrtf(wi_[0]*wi_[0] + wi_[1]*wi_[1]);
    tan_theta = len/wi_[2];
    sin_phi = wi_[1]/len;
    cos_phi = wi_[0]/len;
  }
  float slope_x, slope_y;
  o_ggx_sample11(tan_theta, U1, U2, &slope_x, &slope_y);
  const float tmp = cos_phi*slope_x - sin_phi*slope_y;
  slope_y = sin_phi*slope_x + cos_phi*slope_y;
  slope_x = tmp;
  slope_x = rx*slope_x;
  slope_y = ry*slope_y;
  const float inv_h = sqrtf(slope_x*slope_x + slope_y*slope_y + 1.0);
  h[0] = -slope_x/inv_h;
  h[1] = -slope_y/inv_h;
  h[2] = 1.0/inv_h;
  if(!(inv_h > 0.0)) { h[0] = h[2] = 0.0f; h[1] = 1.0f; }
}

static float o_ggx_pdf_h(const float *wi, const float *h, const float *n, float roughness)
{ /* ggx_pdf_h, ggx.h:167-182 */
  const float r2 = roughness*roughness;
  const float cos_th = fabsf(dot3(h, n));
  const float sin_th = sqrtf(fmaxf(0.0f, 1.0f - cos_th*cos_th));
  const float tan_th = sin_th/cos_th;
  const float D_h = r2/(M_PI * cos_th*cos_th*cos_th*cos_th * (r2 + tan_th*tan_th)*(r2 + tan_th*tan_th));
  const float G1 = o_ggx_G1(wi, n, roughness);
  return fabsf(G1*dot3(wi, h)*D_h/dot3(wi, n));
}

static float o_ggx_pdf_h_cos(float cosh, float cos_in, float cosr, float roughness)
{ /* ggx_pdf_h_mf, ggx.h:184-201 */
  const float r2 = roughness*roughness;
  const float cosh2 = cosh*cosh;
  const float sin_th = sqrtf(OCLAMP(1.0f - cosh2, 0.0f, 1.0f));
  const float tan_th = sin_th/fabsf(cosh);
  const float den = tan_th*tan_th + r2;
  const float ct4 = cosh2*cosh2;
  const float D_h = r2/((M_PI*ct4)*(den*den));   /* mf_set1(M_PI) is a double in the scalar build */
  const float G1 = o_ggx_G1_cos(cos_in, roughness);
  return fabsf((G1*cosr)*(D_h/cos_in));
}

/* ---------------------------------------------------------------- dielectric, src/shaders/dielectric.c */
#define HALFVEC_COS_THR .999
#define GLOSSY_THR 1e-3f

static int o_indexmatched(float n1, float n2)
{ /* dielectric.c:61-65 */
  return fabsf(1.0f - n1/n2) < 1e-3f;
}

static float o_fresnel(float n1, float n2, float cosr, float cost)
{ /* dielectric.c:83-94 */
  if(cost <= 0.0f) return 1.0f;
  const float r1 = n1*cosr, r2 = n2*cosr, t1 = n1*cost, t2 = n2*cost;
  const float Rs = (r1 - t2)/(r1 + t2);
  const float Rp = (t1 - r2)/(t1 + r2);
  return OCLAMP((Rs*Rs + Rp*Rp)*.5f, 0.0f, 1.0f);
}

static float o_sample_dielectric(o_ctx *c, o_path *p)
{ /* sample, dielectric.c:240-415 (MF_COUNT == 1 branches, culled_modes == 0) */
  const int v = p->length-1;
  const float eta_ratio = o_path_eta_ratio(p, v);
  /* hero wavelengths (c->grp, MF_COUNT = 4): the microfacet, the reflect / transmit choice and the outgoing direction are the HERO's (component 0:
     mf(eta_ratio, 0), mf(R, 0), mf(cost2, 0), mf(cost, 0)); every lane then evaluates value and pdf of that direction for its own index of refraction */
  const float eta_hero = o_g_hero(c, eta_ratio);
  if(eta_hero < 0.0f) return 0.0f;
  if(o_g_any(c, o_indexmatched(eta_ratio, 1.0f)))
  {
    for(int k=0;k<3;k++) p->e[v+1].omega[k] = p->e[v].omega[k];
    p->v[v].mode = s_specular | s_transmit;
    p->v[v+1].pdf = 1.0f;
    return p->v[v].shading.rg;
  }
  const float *n = p->v[v].hit.n;
  float ht[3] = {0.0, 0.0, 1.0};
  float pdf_h = 1.0f;
  float h[3] = {n[0], n[1], n[2]};
  const float r = p->v[v].shading.roughness;
  const float cos_in = -dot3(p->v[v].hit.n, p->e[v].omega);
  if(r > GLOSSY_THR)
  {
    const float wit[3] = { -dot3(p->v[v].hit.a, p->e[v].omega), -dot3(p->v[v].hit.b, p->e[v].omega), cos_in };
    /* the reference draws both numbers as call arguments (dielectric.c:266); gcc evaluates them
       right to left: first draw -> U2, second -> U1 */
    float U1, U2;
    if(c->grp) { U1 = o_point(c, p, v+1, o_dim_omega_x); U2 = o_point(c, p, v+1, o_dim_omega_y); }   /* the MF_COUNT=4 reference: this plugin only builds with clang, which evaluates left to right (oracle/Makefile mf4) */
    else       { U2 = o_point(c, p, v+1, o_dim_omega_y); U1 = o_point(c, p, v+1, o_dim_omega_x); }
    o_ggx_sample_h(wit, r, r, U1, U2, ht);
    for(int k=0;k<3;k++) h[k] = ht[0]*p->v[v].hit.a[k] + ht[1]*p->v[v].hit.b[k] + ht[2]*n[k];
    pdf_h = o_ggx_pdf_h(p->e[v].omega, h, n, r);
  }
  float pdf = pdf_h;
  const float cosr = -dot3(p->e[v].omega, h);
  if(cosr <= 0.0f) return 0.0f;

  const float n1 = eta_ratio, n2 = 1.0f;
  const float nr = n1/n2;
  const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
  const float cost = cost2 <= 0.0f ? 0.0f : sqrtf(cost2);
  const float R = o_fresnel(n1, n2, cosr, cost);
  const float R_hero = o_g_hero(c, R), cost2_hero = o_g_hero(c, cost2), cost_hero = o_g_hero(c, cost);

  if(o_point(c, p, v+1, o_dim_scatter_mode) <= R_hero)
  {
    p->v[v].mode = s_reflect;
    for(int k=0;k<3;k++) p->e[v+1].omega[k] = p->e[v].omega[k] + 2.0f*cosr*h[k];
    if(dot3(p->e[v+1].omega, n) <= 0.0f) return 0.0f;
    pdf *= 1.0f/(4.0f*cosr);
    if(r > GLOSSY_THR)
    {
      p->v[v+1].pdf = R*(pdf/fabsf(dot3(p->e[v+1].omega, n)));
      p->v[v].mode |= s_glossy;
      if(dot3(p->e[v+1].omega, n)*dot3(p->e[v+1].omega, h) < 0.0f) return 0.0f;
      return p->v[v].shading.rg * o_ggx_G1(p->e[v+1].omega, n, p->v[v].shading.roughness);
    }
    p->v[v+1].pdf = R;
    p->v[v].mode = s_reflect | s_specular;
    return p->v[v].shading.rg;
  }
  else
  {
    if(cost2_hero <= 0.0f) return 0.0f;                                     /* "can't sample hero, we're all dead" */
    const float f = eta_hero*cosr - cost_hero;
    for(int k=0;k<3;k++) p->e[v+1].omega[k] = p->e[v].omega[k]*eta_hero + f*h[k];
    o_normalise(p->e[v+1].omega);
    if(dot3(p->e[v+1].omega, n) >= 0.0f) return 0.0f;
    if(r <= GLOSSY_THR)
    {
      /* "specular transmit always selects single wavelength": mask = mf_hero = _mm_set_epi32(0u, ~0u, ~0u, ~0u) (include/mf.h:300) zeroes the components
         whose mask bits are set -- _mm_set_epi32 lists the HIGHEST element first, so that is components 0, 1, 2: the one that survives is component 3 */
      const int masked = c->grp && c->lane != O_MF(c) - 1;       /* (eight components: _mm256_set_epi32(0u, ~0u x 7), include/mf.h:42 -- the eighth survives) */
      p->v[v+1].pdf = masked ? 0.0f : 1.0f - R;
      p->v[v].mode = s_specular | s_transmit;
      return masked ? 0.0f : p->v[v].shading.rg;
    }
    if(c->grp)
    { /* dielectric.c:353-411: "we sampled a half vector for the configuration of wi and wo. unfortunately it's only valid for the hero wavelength":
         every component reconstructs the half vector ITS index of refraction needs to connect wi and wo, with its own Fresnel term */
      const float *wi = p->e[v].omega, *wo = p->e[v+1].omega;
      int mask = 0;
      float h0 = n1*wi[0] - n2*wo[0], h1 = n1*wi[1] - n2*wo[1], h2 = n1*wi[2] - n2*wo[2];
      const float hilen = 1.0f/sqrtf(h0*h0 + (h1*h1 + h2*h2));
      h0 *= hilen; h1 *= hilen; h2 *= hilen;
      if(n2 < n1) { h0 = -h0; h1 = -h1; h2 = -h2; }
      const float cosh2 = h0*n[0] + (h1*n[1] + h2*n[2]);
      mask |= cosh2 < 0.0f;
      const float cosr2 = h0*-wi[0] + (h1*-wi[1] + h2*-wi[2]);
      mask |= cosr2 <= 0.0f;
      const float cost2b = 1.0f - (nr*nr)*(1.0f - cosr2*cosr2);
      const float costb = cost2b <= 0.0f ? 0.0f : sqrtf(cost2b);
      const float R2 = o_fresnel(n1, n2, cosr2, costb);
      const float denom = n1*cosr2 - n2*costb;
      float pdf2 = o_ggx_pdf_h_cos(cosh2, cos_in, cosr2, p->v[v].shading.roughness);
      pdf2 = pdf2*(((n2*n2)*costb)/(denom*denom));
      p->v[v+1].pdf = mask ? 0.0f : (pdf2*(1.0f - R2))/fabsf(dot3(p->e[v+1].omega, n));
      p->v[v].mode = s_transmit | s_glossy;
      const float G1 = o_ggx_G1(wo, n, p->v[v].shading.roughness);
      return mask ? 0.0f : p->v[v].shading.rg*G1;
    }
    const float denom = n1*cosr - n2*cost;
    pdf *= n2*n2*cost/(denom*denom);
    p->v[v+1].pdf = (pdf*(1.0f - R))/fabsf(dot3(p->e[v+1].omega, n));
    p->v[v].mode = s_transmit | s_glossy;
    const float G1 = o_ggx_G1(p->e[v+1].omega, n, p->v[v].shading.roughness);
    return p->v[v].shading.rg*G1;
  }
}

static float o_brdf_dielectric(o_path *p, int v)
{ /* brdf, dielectric.c:418-541 (scalar) */
  const float cos_in  = -dot3(p->v[v].hit.n, p->e[v].omega);
  const float cos_out =  dot3(p->v[v].hit.n, p->e[v+1].omega);
  const float eta_ratio = o_path_eta_ratio(p, v);
  if(o_g_hero(O_CTX(p), eta_ratio) < 0.0f) return 0.0f;                       /* mf(eta_ratio, 0) < 0, dielectric.c:423 */
  const float n1 = eta_ratio, n2 = 1.0f;
  const int index_matched = o_g_any(O_CTX(p), o_indexmatched(n1, n2));       /* mf_any over the wavelengths */
  if(cos_out == 0.0f || cos_in == 0.0f) return 0.0f;
  if(!index_matched && (cos_in*cos_out > 0)) p->v[v].mode = s_reflect;
  else p->v[v].mode = s_transmit;
  const float r = p->v[v].shading.roughness;
  if((r > GLOSSY_THR) && !index_matched) p->v[v].mode |= s_glossy;
  else p->v[v].mode |= s_specular;
  const float *wi = p->e[v].omega, *wo = p->e[v+1].omega, *n = p->v[v].hit.n;

  if(index_matched)
  {
    float h[3];
    const float dot_wo_n = dot3(wo, n);
    for(int k=0;k<3;k++) h[k] = -wi[k] + wo[k] - 2.0f*dot_wo_n*n[k];
    o_normalise(h);
    const float cosh = dot3(h, n);
    if(cosh < 0.0f) return 0.0f;
    if(cosh < HALFVEC_COS_THR) return 0.0f;
    return p->v[v].shading.rg;
  }
  else if(p->v[v].mode & s_reflect)
  {
    float h[3];
    for(int k=0;k<3;k++) h[k] = -wi[k] + wo[k];
    o_normalise(h);
    const float cosh = dot3(h, n);
    if(cosh < 0.0f) return 0.0f;
    const float DG1 = (p->v[v].mode & s_specular) ? 1.0f : o_ggx_pdf_h(wi, h, n, r);
    if(DG1 == 0) return 0.0f;
    const float cosr = -dot3(h, wi);
    if(cosr < 0.0f) return 0.0f;
    const float nr = n1/n2;
    const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
    const float cost = cost2 <= 0.0f ? 0.0f : sqrtf(cost2);
    const float R = o_fresnel(n1, n2, cosr, cost);
    const float G1 = o_ggx_G1(wo, n, r);
    if(p->v[v].mode & s_glossy)
      return (p->v[v].shading.rg*R)*(DG1*G1/(4.0f*fabsf(cosr*cos_out)));
    if(cosh < HALFVEC_COS_THR) return 0.0f;
    return p->v[v].shading.rg*R;
  }
  else
  {
    int mask = 0;
    float h0 = n1*wi[0] - n2*wo[0], h1 = n1*wi[1] - n2*wo[1], h2 = n1*wi[2] - n2*wo[2];
    const float hilen = 1.0f/sqrtf(h0*h0 + (h1*h1 + h2*h2));
    h0 *= hilen; h1 *= hilen; h2 *= hilen;
    float cosh2 = h0*n[0] + (h1*n[1] + h2*n[2]);
    const int cosh_lt0 = cosh2 < 0.0f;
    mask |= cosh_lt0 && (n1 < n2);
    mask |= !cosh_lt0 && (n2 < n1);
    if(cosh_lt0) { cosh2 = -cosh2; h0 = -h0; h1 = -h1; h2 = -h2; }
    const float cosr2 = h0*-wi[0] + (h1*-wi[1] + h2*-wi[2]);
    mask |= cosr2 <= 0.0f;
    const float nr = n1/n2;
    const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr2*cosr2);
    const float cost = cost2 <= 0.0f ? 0.0f : sqrtf(cost2);
    const float R2 = o_fresnel(n1, n2, cosr2, cost);
    const float DG1 = o_ggx_pdf_h_cos(cosh2, cos_in, cosr2, r);
    const float G1 = o_ggx_G1_cos(cos_in, r);
    const float cos_hwo = h0*wo[0] + (h1*wo[1] + h2*wo[2]);
    mask |= cos_hwo >= 0.0f;
    float denom = n1*cosr2 - n2*cost;
    denom = denom*denom;
    if(cos_in == 0.0f) return 0.0f;
    if(p->v[v].mode & s_glossy)
      return mask ? 0.0f : ((p->v[v].shading.rg*(1.0f - R2))*((n2*n2)*(cost*(DG1*(G1*(1.0f/fabsf(cos_out)))))))/denom;
    mask |= cosh2 < HALFVEC_COS_THR;
    return mask ? 0.0f : p->v[v].shading.rg*OCLAMP(1.0f - R2, 0.0f, 1.0f);
  }
}

static float o_pdf_dielectric(o_path *p, int e1, int v, int e2)
{ /* pdf, dielectric.c:96-237 (forward direction e1 < e2 is the only one pt/ptdl asks for) */
  float wi[3], wo[3], n[3];
  for(int k=0;k<3;k++) { wi[k] = p->e[e1].omega[k]; wo[k] = p->e[e2].omega[k]; n[k] = p->v[v].hit.n[k]; }
  const float cos_in  = -dot3(n, wi);
  const float cos_out =  dot3(n, wo);
  if(cos_in*cos_out == 0.0f) return 0.0f;
  if(cos_out > 0.0f && !(p->v[v].mode & s_reflect))  return 0.0f;
  if(cos_out < 0.0f && !(p->v[v].mode & s_transmit)) return 0.0f;
  float h[3];
  const float eta = o_path_eta_ratio(p, v);
  if(o_g_all(O_CTX(p), eta < 0.0f)) return 0.0f;                              /* mf_all(mf_lt(eta, 0)), dielectric.c:133 */
  const float n1 = eta, n2 = 1.0f;
  int mask = 0;
  float cosr = 0.0f, cosh = 0.0f;
  if(o_g_any(O_CTX(p), o_indexmatched(n1, n2)))
  {
    const float dot_wo_n = dot3(wo, n);
    for(int k=0;k<3;k++) h[k] = -wi[k] + wo[k] - 2.0f*dot_wo_n*n[k];
    o_normalise(h);
    cosh = dot3(h, n);
    if(p->v[v].mode != (s_transmit | s_specular)) return 0.0f;
    if(cosh < HALFVEC_COS_THR) return 0.0f;
    return 1.0f;
  }
  else if(p->v[v].mode & s_reflect)
  {
    for(int k=0;k<3;k++) h[k] = wi[k] - wo[k];
    o_normalise(h);
    cosh = fabsf(dot3(h, n));
    cosr = fabsf(dot3(h, wi));
  }
  else
  {
    float h0 = n1*wi[0] - n2*wo[0], h1 = n1*wi[1] - n2*wo[1], h2 = n1*wi[2] - n2*wo[2];
    const float hilen = 1.0f/sqrtf(h0*h0 + (h1*h1 + h2*h2));
    h0 *= hilen; h1 *= hilen; h2 *= hilen;
    if(n2 < n1) { h0 = -h0; h1 = -h1; h2 = -h2; }
    h[0] = h0; h[1] = h1; h[2] = h2;
    cosh = h0*n[0] + (h1*n[1] + h2*n[2]);
    mask |= cosh < 0.0f;
    cosr = h0*-wi[0] + (h1*-wi[1] + h2*-wi[2]);
    mask |= cosr <= 0.0f;
  }
  const float nr = n1/n2;
  const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
  const float cost = cost2 <= 0.0f ? 0.0f : sqrtf(cost2);
  const float R = o_fresnel(n1, n2, cosr, cost);
  float pdf = 1.0f;
  if(p->v[v].mode & s_reflect)
  {
    if(p->v[v].mode & s_specular)
    {
      mask |= cosh < HALFVEC_COS_THR;
      return mask ? 0.0f : R;
    }
    pdf = pdf*(1.0f/(4.0f*fabsf(dot3(wo, h))));
    pdf = pdf*R;
  }
  else
  {
    if(p->v[v].mode & s_specular)
    {
      mask |= cosh < HALFVEC_COS_THR;
      return mask ? 0.0f : OCLAMP(1.0f - R, 0.0f, 1.0f);
    }
    const float denom = n1*cosr - n2*cost;
    pdf = pdf*(((n2*n2)*cost)/(denom*denom));
    pdf = pdf*OCLAMP(1.0f - R, 0.0f, 1.0f);
  }
  pdf = pdf*o_ggx_pdf_h_cos(cosh, cos_in, cosr, p->v[v].shading.roughness);
  pdf = pdf/fabsf(cos_out);
  mask |= !(pdf > 0.0f);
  return mask ? 0.0f : pdf;
}

/* ---------------------------------------------------------------- metal, src/shaders/metal.c */
static void o_metal_ior(const mi_scene_desc *s, int mat, float lambda, float *n, float *k)
{ /* fresnel_get_ior_mf, src/shaders/fresnel.h:519-531 */
  const int i = OCLAMP((lambda - 360.0f)/5.0f, 0, 94);
  *n =  s->metal_ior[(mat*95 + i)*2 + 0];
  *k = -s->metal_ior[(mat*95 + i)*2 + 1];
}

static float o_fresnel_metal(float n1, float n2, float k2, float cosr)
{ /* fresnel, metal.c:79-157: unpolarised reflectance at a conductor, complex arithmetic expanded */
  const float etar =   (n1*n2)/(n2*n2 + k2*k2);
  const float etai = -((n1*k2)/(n2*n2 + k2*k2));
  const float eta2r = etar*etar - etai*etai;
  const float eta2i = (2.0f*etar)*etai;
  const float sinr = 1.0f - cosr*cosr;
  const float cost2r = 1.0f - eta2r*sinr;
  const float cost2i = eta2i*(-sinr);
  const float len = sqrtf(cost2r*cost2r + cost2i*cost2i);
  const float costr = sqrtf(0.5f*(cost2r + len));
  float costi = sqrtf(0.5f*(len - cost2r));
  if(cost2i < 0.0f) costi = -costi;
  const float n1cosr = n1*cosr, n2cosrr = n2*cosr, n2cosri = k2*cosr;
  const float n1costr = n1*costr, n1costi = n1*costi;
  const float n2costr = n2*costr - k2*costi;
  const float n2costi = k2*costr + n2*costi;
  const float Rs2 = ((n1cosr - n2costr)*(n1cosr - n2costr) + n2costi*n2costi) /
                    ((n1cosr + n2costr)*(n1cosr + n2costr) + n2costi*n2costi);
  const float Rp2 = ((n1costr - n2cosrr)*(n1costr - n2cosrr) + (n1costi - n2cosri)*(n1costi - n2cosri)) /
                    ((n1costr + n2cosrr)*(n1costr + n2cosrr) + (n1costi + n2cosri)*(n1costi + n2cosri));
  return OCLAMP((Rs2 + Rp2)*.5f, 0.0f, 1.0f);
}

/* What the REFERENCE BUILD's sample() does on top of the formula above (test switch, off by default; oracle_set_reference_metal).
 * The plugin as gcc 11 -O3 -ffast-math -march=x86-64-v3 compiles it (disassembly of libmetal.so in oracle/_ref, sample+0x1bb..0x259)
 * forms costi = sqrt(0.5 (len - cost2r)) as sqrt(0.5 fma(eta2r, sinr, len - 1)) with cost2r = fma(-eta2r, sinr, 1) and
 * len = sqrt(fma(cost2r, cost2r, cost2i^2)): near normal incidence on the microfacet cost2i^2 vanishes beside cost2r^2, len == cost2r,
 * and what is left under the root is the ROUNDING ERROR of cost2r -- negative for every other sample. The NaN runs through Rs, Rp
 * into the final clamp, which maps it to R = 0: the sample gets weight 0 and the path ends. Measured on the plugin: exactly half of
 * the samples with sin^2 below 2.4e-4 cost2r / |eta2i| (3e-3 for gold at 525 nm, 5e-2 at 720 nm), none above; 2.2 % of all samples
 * at normal incidence and roughness 0.3 (the reference fails its own battle test there: ebsdf .6137 against bsdf .6299), 4.3 % at
 * 720 nm. brdf() and pdf() of the same plugin are compiled differently and lose nothing. This predicate restates the compiled
 * sequence operation by operation (fmaf where the plugin has a fused instruction), so it decides like the plugin on the same
 * cosr -- the cosr of a path agrees with the reference's to the last bit only where its own history does, so path by path the
 * killed samples are not the reference's, in number and in place on the hemisphere they are. */
static int o_reference_metal = 0;
int oracle_set_reference_metal(int on) { o_reference_metal = on; return 1; }
static int o_metal_reference_kills(float n1, float n2, float k2, float cosr)
{
  const float sinr = fmaf(-cosr, cosr, 1.0f);
  const float den = fmaf(n2, n2, k2*k2);
  const float etar = (n1*n2)/den;
  const float etai = -((k2*n1)/den);
  const float cost2i = (fmaf(cosr, cosr, -1.0f)*-2.0f)*(etar*etai);
  const float eta2r = fmaf(etar, etar, -(etai*etai));
  const float cost2r = fmaf(-eta2r, sinr, 1.0f);
  const float len = sqrtf(fmaf(cost2r, cost2r, cost2i*cost2i));
  return 0.5f*fmaf(eta2r, sinr, len - 1.0f) < 0.0f;
}

static float o_sample_metal(o_ctx *c, o_path *p)
{ /* sample, metal.c:219-265 */
  const mi_scene_desc *s = c->s;
  const int v = p->length-1;
  const mi_material *m = s->materials + p->v[v].hit.shader;
  const float *n = p->v[v].hit.n;
  float h[3] = {n[0], n[1], n[2]};
  float pdf_h = 1.0f;
  const float r = p->v[v].shading.roughness;
  if(r > 1e-4f)
  {
    const float wit[3] = { -dot3(p->v[v].hit.a, p->e[v].omega), -dot3(p->v[v].hit.b, p->e[v].omega), -dot3(n, p->e[v].omega) };
    float ht[3];
    float U1, U2;
    if(c->grp) { U1 = o_point(c, p, v+1, o_dim_omega_x); U2 = o_point(c, p, v+1, o_dim_omega_y); }   /* the MF_COUNT=4 reference: this plugin only builds with clang, which evaluates left to right (oracle/Makefile mf4) */
    else       { U2 = o_point(c, p, v+1, o_dim_omega_y); U1 = o_point(c, p, v+1, o_dim_omega_x); }
    o_ggx_sample_h(wit, r, r, U1, U2, ht);
    for(int k=0;k<3;k++) h[k] = ht[0]*p->v[v].hit.a[k] + ht[1]*p->v[v].hit.b[k] + ht[2]*n[k];
    pdf_h = o_ggx_pdf_h(p->e[v].omega, h, n, r);
  }
  float pdf = pdf_h;
  const float cosr = -dot3(p->e[v].omega, h);
  if(!(cosr > 0.0f)) return 0.0f;
  float n2, k2;
  o_metal_ior(s, (int)m->param[0], p->lambda, &n2, &k2);
  const float R = (o_reference_metal && o_metal_reference_kills(p->e[v].vol.ior, n2, k2, cosr)) ? 0.0f : o_fresnel_metal(p->e[v].vol.ior, n2, k2, cosr);
  p->v[v].mode = s_reflect;
  for(int k=0;k<3;k++) p->e[v+1].omega[k] = p->e[v].omega[k] + 2.0f*cosr*h[k];
  if(dot3(p->e[v+1].omega, n) <= 0.0f) return 0.0f;
  pdf *= 1.0f/(4.0f*cosr);
  if(r > 1e-4f)
  {
    p->v[v+1].pdf = pdf/fabsf(dot3(p->e[v+1].omega, n));
    p->v[v].mode |= s_glossy;
    if(dot3(p->e[v+1].omega, n)*dot3(p->e[v+1].omega, h) < 0.0f) return 0.0f;
    return R*(p->v[v].shading.rg*o_ggx_G1(p->e[v+1].omega, n, p->v[v].shading.roughness));
  }
  p->v[v].mode |= s_specular;
  return R*p->v[v].shading.rg;
}

static float o_brdf_metal(o_ctx *c, o_path *p, int v)
{ /* brdf, metal.c:268-310 */
  const mi_scene_desc *s = c->s;
  const mi_material *m = s->materials + p->v[v].hit.shader;
  const float *n = p->v[v].hit.n;
  const float cos_in  = -dot3(n, p->e[v].omega);
  const float cos_out =  dot3(n, p->e[v+1].omega);
  float n2, k2;
  o_metal_ior(s, (int)m->param[0], p->lambda, &n2, &k2);
  if(cos_out <= 0.0f || cos_in <= 0.0f) return 0.0f;
  p->v[v].mode = s_reflect;
  const float r = p->v[v].shading.roughness;
  if(r > 1e-4f) p->v[v].mode |= s_glossy; else p->v[v].mode |= s_specular;
  float h[3];
  for(int k=0;k<3;k++) h[k] = -p->e[v].omega[k] + p->e[v+1].omega[k];
  o_normalise(h);
  const float cosh = dot3(h, n);
  if(cosh < 0.0f) return 0.0f;
  const float DG1 = o_ggx_pdf_h(p->e[v].omega, h, n, r);
  if(DG1 == 0) return 0.0f;
  const float cosr = -dot3(h, p->e[v].omega);
  if(cosr < 0.0f) return 0.0f;
  const float R = o_fresnel_metal(p->e[v].vol.ior, n2, k2, cosr);
  const float G1 = o_ggx_G1(p->e[v+1].omega, n, r);
  if(p->v[v].mode & s_glossy) return (p->v[v].shading.rg*R)*(DG1*G1/(4.0f*fabsf(cosr*cos_out)));
  if(cosh < HALFVEC_COS_THR) return 0.0f;
  return p->v[v].shading.rg*R;
}

static float o_pdf_metal(o_path *p, int v)
{ /* pdf, metal.c:170-216 (forward direction) */
  if(!(p->v[v].mode & s_reflect)) return 0.0f;
  const float *n = p->v[v].hit.n, *wi = p->e[v].omega, *wo = p->e[v+1].omega;
  const float cos_in = -dot3(n, wi), cos_out = dot3(n, wo);
  if(cos_in < 0.0f) return 0.0f;
  if(cos_out < 0.0f) return 0.0f;
  float h[3];
  for(int k=0;k<3;k++) h[k] = wi[k] - wo[k];
  o_normalise(h);
  if(p->v[v].mode & s_specular)
  {
    const float cosh = fabsf(dot3(h, n));
    if(cosh < HALFVEC_COS_THR) return 0.0f;
    return 1.0f;
  }
  float pdf = 1.0f/(4.0f*fabsf(dot3(wo, h)));
  pdf *= o_ggx_pdf_h(wi, h, n, p->v[v].shading.roughness);
  pdf /= fabsf(cos_out);
  if(!(pdf > 0.0f)) return 0.0f;
  return pdf;
}

/* ---------------------------------------------------------------- dispatch */
static float o_sample_medium(o_ctx *c, o_path *p);
float o_shader_sample(o_ctx *c, o_path *p)
{ /* shader_sample, src/shader.c:577-590 */
  const int v = p->length-1;
  const mi_material *m = c->s->materials + p->v[v].hit.shader;
  float throughput = 0.0f;
  if(m->bsdf == MI_BSDF_DIFFUSE)         throughput = o_sample_diffuse(c, p);
  else if(m->bsdf == MI_BSDF_DIELECTRIC) throughput = o_sample_dielectric(c, p);
  else if(m->bsdf == MI_BSDF_METAL)      throughput = o_sample_metal(c, p);
  else if(m->bsdf == MI_BSDF_MEDIUM)     throughput = o_sample_medium(c, p);
  o_normalise(p->e[v+1].omega);
  const float dt = ((p->v[v].flags & s_inside) ? -1 : 1)*dot3(p->v[v].hit.gn, p->e[v+1].omega);
  if(((p->v[v].mode & s_reflect) && (dt < 0.f)) || ((p->v[v].mode & s_transmit) && (dt > 0.f)))
    return 0.0f;
  return throughput;
}

/* ---------------------------------------------------------------- homogeneous medium: Henyey-Greenstein */
static float o_eval_hg(float g, const float *wi, const float *wo)
{ /* sample_eval_hg, include/sampler_common.h:338-355 */
  if(g == 0.0f) return 1.0f/(4.0f*M_PI);
  const float cos_theta = dot3(wi, wo);
  return 1.0f/(4.0f*M_PI)*(1.0f-g*g)/powf(1.0f + g*g - 2.0f*g*cos_theta, 3.0f/2.0f);
}

static float o_sample_medium(o_ctx *c, o_path *p)
{ /* medium_rgb.c:61-72 + sample_hg, include/sampler_common.h:286-316. The two numbers are call arguments: drawn right to left
     (omega_y first) by the reference build, like dielectric.c:266 */
  const int v = p->length-1;
  p->v[v].mode |= s_glossy | s_volume;
  const o_hit *hit = &p->v[v].hit;
  const float g = p->v[v].interior.mean_cos;
  const float r2 = o_point(c, p, v+1, o_dim_omega_y);
  const float r1 = o_point(c, p, v+1, o_dim_omega_x);
  float out[3], pdf;
  if(g == 0.0f)
  { /* sample_sphere, include/sampler_common.h */
    const float z = 1.0f - 2.0f*r1;
    const float r = sqrtf(1.0f - z*z);
    const float phi = 2.0f*M_PI*r2;
    out[0] = r*cosf(phi); out[1] = r*sinf(phi); out[2] = z;
    pdf = 1.0f/(4.0f*M_PI);
  }
  else
  {
    const float sqr = (1.0f-g*g)/(1.0f+g*(2.0f*r1-1.0f));
    const float cos_theta = 1.0f/(2.0f*g)*(1.0f + g*g - sqr*sqr);
    const float phi = 2.0f*M_PI*r2;
    const float l = sqrtf(fmaxf(0.0f, 1.0f-cos_theta*cos_theta));
    out[0] = cos_theta;
    out[1] = cosf(phi)*l;
    out[2] = sinf(phi)*l;
    pdf = 1.0f/(4.0f*M_PI)*(1.0f-g*g)/powf(1.0f + g*g - 2.0f*g*cos_theta, 3.0f/2.0f);
  }
  p->v[v+1].pdf = pdf;
  for(int k=0;k<3;k++) p->e[v+1].omega[k] = hit->n[k]*out[0] + hit->a[k]*out[1] + hit->b[k]*out[2];
  return p->v[v].interior.mu_s;
}

static int o_vertex_shader(const o_path *p, int v)
{ /* src/shader.c:448-455,568-575: a volume vertex without its own shader falls back on the shader of its volume */
  int shader = p->v[v].hit.shader;
  if(shader < 0 && (p->v[v].mode & s_volume)) shader = p->v[v].interior.shader;
  return shader;
}

float o_shader_brdf(o_ctx *c, o_path *p, int v)
{ /* shader_brdf, src/shader.c:568-575 */
  const mi_material *m = c->s->materials + o_vertex_shader(p, v);
  if(m->bsdf == MI_BSDF_MEDIUM)
  { /* medium_rgb.c:98-102 */
    p->v[v].mode |= s_glossy | s_volume;
    return p->v[v].interior.mu_s*o_eval_hg(p->v[v].interior.mean_cos, p->e[v].omega, p->e[v+1].omega);
  }
  if(m->bsdf == MI_BSDF_DIFFUSE)    return o_brdf_diffuse(p, v);
  if(m->bsdf == MI_BSDF_DIELECTRIC) return o_brdf_dielectric(p, v);
  if(m->bsdf == MI_BSDF_METAL)      return o_brdf_metal(c, p, v);
  return 0.0f;
}

float o_shader_pdf(o_ctx *c, o_path *p, int v)
{ /* shader_pdf, src/shader.c:448-455 */
  const mi_material *m = c->s->materials + o_vertex_shader(p, v);
  if(m->bsdf == MI_BSDF_MEDIUM)
  { /* medium_rgb.c:92-96 */
    if(!(p->v[v].mode & s_volume)) return 0.0f;
    return o_eval_hg(p->v[v].interior.mean_cos, p->e[v].omega, p->e[v+1].omega);
  }
  if(m->bsdf == MI_BSDF_DIFFUSE)    return 1.0f/M_PI;                  /* pdf_d, src/shader.c:254-257 */
  if(m->bsdf == MI_BSDF_DIELECTRIC) return o_pdf_dielectric(p, v, v, v+1);
  if(m->bsdf == MI_BSDF_METAL)      return o_pdf_metal(p, v);
  return 0.0f;
}
