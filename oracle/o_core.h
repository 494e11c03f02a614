/* o_core.h -- TEST INFRASTRUCTURE: shared types of the CPU restatement (see oracle.h).
 * Path/vertex/edge records mirror the fields of the reference's path_t that the pt/ptdl
 * hot path touches (include/pathspace.h:91-225, include/corona_common.h:119-141). */
#ifndef O_CORE_H
#define O_CORE_H

#include "oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

#define O_MAX_VERTS 33

/* vertex_scattermode_t / vertex_flags_t, include/pathspace.h:57-82 */
enum { s_absorb = 0, s_reflect = 1, s_transmit = 2, s_volume = 4, s_fiber = 8, s_emit = 16, s_sensor = 32,
       s_diffuse = 64, s_glossy = 128, s_specular = 256 };
enum { s_none = 0, s_inside = 1, s_environment = 2 };
enum { s_tech_extend = 1, s_tech_nee = 2 };   /* include/pathspace/tech.h (values only used as tags) */

#define dot3(a, b) ((a)[0]*(b)[0] + (a)[1]*(b)[1] + (a)[2]*(b)[2])
#define cross3(a, b, r) do { (r)[0] = (a)[1]*(b)[2] - (b)[1]*(a)[2]; (r)[1] = (a)[2]*(b)[0] - (b)[2]*(a)[0]; (r)[2] = (a)[0]*(b)[1] - (b)[0]*(a)[1]; } while(0)
/* reference macros: NaN falls through to the second operand (include/corona_common.h:168-170) */
#define OMAX(a, b) ((a) > (b) ? (a) : (b))
#define OMIN(a, b) ((a) < (b) ? (a) : (b))
#define OCLAMP(a, m, M) OMIN(OMAX(a, m), M)

static inline void o_normalise(float *f)
{ /* include/corona_common.h:172-176 */
  const float len = 1.0f/sqrtf(dot3(f, f));
  for(int k=0;k<3;k++) f[k] *= len;
}

static inline void o_get_onb(const float *n, float *u, float *v)
{ /* include/corona_common.h:178-198 */
  if(fabsf(n[1]) < 0.5) { const float up[3] = {0, 1, 0}; cross3(n, up, u); }
  else                  { const float rg[3] = {1, 0, 0}; cross3(n, rg, u); }
  o_normalise(u);
  cross3(n, u, v);
}

static inline void o_get_scrambled_onb(const float scramble, const float *n, float *u, float *v)
{ /* include/corona_common.h:200-215 */
  if(fabsf(n[1]) < scramble) { const float up[3] = {0, 1, 0}; cross3(n, up, u); }
  else                       { const float rg[3] = {1, 0, 0}; cross3(n, rg, u); }
  o_normalise(u);
  cross3(n, u, v);
}

typedef struct o_ray { float pos[3], dir[3], time, min_dist; mi_primid ignore; } o_ray;      /* ray_t */

typedef struct o_hit
{ /* hit_t */
  mi_primid prim;
  float u, v;
  float r, s, t;
  float a[3], b[3], n[3], x[3], gn[3];
  int shader;
  float dist;
} o_hit;

typedef struct o_shading { float roughness, rs, rd, rg, em; } o_shading;            /* vertex_shading_t */
typedef struct o_volume { float ior; int shader; float mu_s, mu_t, mean_cos; } o_volume;   /* vertex_volume_t, include/pathspace.h:105-118 (homogeneous, no extra lobes) */

typedef struct o_vertex
{ /* vertex_t */
  o_hit hit;
  float pdf, throughput, total_throughput;
  int tech;
  o_shading shading;
  o_volume interior;
  float eta;                     /* diffgeo.eta */
  uint32_t flags, mode, material_modes;
  int rand_beg, rand_cnt;        /* first random dimension of this vertex and how many it owns (include/pathspace.h:157-158) */
} o_vertex;

typedef struct o_edge
{ /* edge_t */
  float contribution, transmittance, pdf;
  float omega[3];
  float dist;
  o_volume vol;
} o_edge;

struct o_ctx;
typedef struct o_path
{ /* path_t */
  struct o_ctx *ctx;             /* (hero wavelengths: the lane's context, for the group hooks below; NULL = scalar) */
  float lambda, throughput;
  int length;
  float time;
  uint64_t index;
  float scramble;                /* tangent_frame_scrambling */
  float pixel_i, pixel_j;
  o_vertex v[O_MAX_VERTS];
  o_edge   e[O_MAX_VERTS+1];
} o_path;

/* Hero wavelengths (the reference built with -DMF_COUNT=4, include/mf.h:280-423): every spectral quantity of a path is a vector of four, one per
 * wavelength; geometry and every decision follow component 0, the hero. The restatement runs FOUR scalar lanes in lock step -- one thread per
 * wavelength, each executing the scalar code above and below with its own lambda -- and the few places where the reference looks across the
 * components (mf_any / mf_all / mf(x, 0) / mf_hsum) are group hooks: every lane contributes its value, a barrier, every lane reads the result.
 * All four lanes draw the same random numbers and take the same branches (every wavelength-dependent branch goes through a hook), so they meet
 * at the same hooks in the same order. Scalar mode (grp == NULL, everything before round 5): a hook is the identity on the lane's own value. */
#include <pthread.h>
#define O_MF_MAX 8                /* MF_COUNT = 4 (SSE, include/mf.h:280-423) or 8 (AVX, include/mf.h:22-279): o_group.n lanes */
#define O_MF(c) ((c)->grp->n)
typedef struct o_group
{
  volatile int arrived, sense;   /* sense-reversing spin barrier: the lanes meet twice per hook, a futex sleep each time cost 2 ms per path */
  int n;                         /* lanes = wavelengths per path */
  volatile float f[O_MF_MAX];
  volatile double d[O_MF_MAX];
  volatile int b[O_MF_MAX];
  volatile float col[O_MF_MAX][3];
} o_group;

typedef struct o_ctx
{
  const mi_scene_desc *s;
  o_group *grp;                  /* hero wavelengths: the four lanes' meeting point, NULL in scalar mode */
  int lane;                      /* 0 = hero */
  int sense;                     /* this lane's side of the barrier */
  uint64_t rng0, rng1;           /* xorshift128+ state */
  float *fb;
  int atomic_fb;
  mi_path_record *rec;
  float *hero_ext;               /* hero wavelengths: all components of the finished path (every lane writes its own column): oracle_hero_ext with n columns */
  int hero_splats;
  uint64_t cnt[8];
} o_ctx;

#include <sched.h>
static inline void o_g_meet(o_ctx *c)
{
  o_group *g = c->grp;
  const int s = c->sense ^= 1;
  /* what a lane wrote into the group before it arrives is visible to every lane that has seen the new sense (release / acquire on `sense`,
     acquire-release on the arrival count: the last lane has seen the others' writes and publishes them with its own) */
  if(__atomic_add_fetch(&g->arrived, 1, __ATOMIC_ACQ_REL) == g->n) { __atomic_store_n(&g->arrived, 0, __ATOMIC_RELAXED); __atomic_store_n(&g->sense, s, __ATOMIC_RELEASE); }
  else { int spins = 0; while(__atomic_load_n(&g->sense, __ATOMIC_ACQUIRE) != s) { if(++spins > 4000) sched_yield(); else __builtin_ia32_pause(); } }
}
static inline int o_g_any(o_ctx *c, int cond)       /* mf_any */
{
  if(!c || !c->grp) return cond;
  c->grp->b[c->lane] = cond; o_g_meet(c);
  int r = 0;
  for(int l=0;l<c->grp->n;l++) r |= c->grp->b[l];
  o_g_meet(c);
  return r;
}
static inline int o_g_all(o_ctx *c, int cond)       /* mf_all */
{
  if(!c || !c->grp) return cond;
  c->grp->b[c->lane] = cond; o_g_meet(c);
  int r = 1;
  for(int l=0;l<c->grp->n;l++) r &= c->grp->b[l];
  o_g_meet(c);
  return r;
}
static inline float o_g_hero(o_ctx *c, float x)     /* mf(x, 0) */
{
  if(!c || !c->grp) return x;
  c->grp->f[c->lane] = x; o_g_meet(c);
  const float r = c->grp->f[0];
  o_g_meet(c);
  return r;
}
static inline float o_g_hsum(o_ctx *c, float x)     /* mf_hsum: _mm_hadd_ps twice = (a0 + a1) + (a2 + a3), include/mf.h:301-306; eight components: _mm256_hadd_ps twice,
                                                       then the upper half is added to the lower, include/mf.h:44-51 = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) */
{
  if(!c || !c->grp) return x;
  c->grp->f[c->lane] = x; o_g_meet(c);
  float r = (c->grp->f[0] + c->grp->f[1]) + (c->grp->f[2] + c->grp->f[3]);
  if(c->grp->n == 8) r = r + ((c->grp->f[4] + c->grp->f[5]) + (c->grp->f[6] + c->grp->f[7]));
  o_g_meet(c);
  return r;
}
#define O_CTX(p) ((p)->ctx)
/* the n-column layout of oracle_hero_ext (oracle.h) = dump_ext_t of the dump harness built with -DMF_COUNT=n: lambda[n], then six per-vertex fields [V][n], then splat_value[S][n] */
#define O_EXT_LAMBDA(c)        ((c)->hero_ext + (c)->lane)
#define O_EXT_VERTEX(c, F, v)  ((c)->hero_ext + (c)->grp->n*(1 + (F)*MI_REC_MAX_VERTS + (v)) + (c)->lane)      /* F: 0 throughput, 1 pdf, 2 rd, 3 rg, 4 em, 5 eta */
#define O_EXT_SPLAT(c, k)      ((c)->hero_ext + (c)->grp->n*(1 + 6*MI_REC_MAX_VERTS + (k)) + (c)->lane)
#define O_EXT_FLOATS(n)        ((n)*(1 + 6*MI_REC_MAX_VERTS + MI_REC_MAX_SPLATS))

/* oracle_rng: src/points.d/xorshift128p.c */
float o_rand(o_ctx *c);
void  o_rand_seed(o_ctx *c, uint64_t index, uint64_t frame);

/* pointsampler(path, dim): MOD_pointsampler = rand (src/pointsampler.d/rand.c:48-55, ignores the dimension) or halton
 * (src/pointsampler.d/halton.c:69-84: dimension = rand_beg of vertex v + dim, index = low 32 bits of the path index).
 * v = the vertex under construction = path->length at the call (length-1 inside path_russian_roulette, src/pathspace.c:278-281).
 * rand_beg / rand_cnt are kept as the reference keeps them: thinlens.c:100-103, src/pathspace.c:199,208,298, nee.h:108,225,231. */
enum { o_dim_image_x = 0, o_dim_image_y = 1, o_dim_lambda = 2, o_dim_time = 3, o_dim_aperture_x = 4, o_dim_aperture_y = 5, o_dim_camid = 6,
       o_dim_omega_x = 1, o_dim_omega_y = 2, o_dim_scatter_mode = 3, o_dim_russian_r = 4,
       o_dim_nee_light1 = 0, o_dim_nee_light2 = 1, o_dim_nee_x = 2, o_dim_nee_y = 3 };   /* include/pathspace.h:16-53 */
float o_point(o_ctx *c, const o_path *p, int v, int dim);

/* oracle_halton.c */
void  o_halton_prepare(uint64_t seed);
float o_halton_sample(uint32_t dim, uint32_t index);

/* oracle_geo.c */
void  o_accel_intersect(o_ctx *c, const o_ray *ray, o_hit *hit);
void  o_prims_get_normal(const mi_scene_desc *s, mi_primid pi, o_hit *hit, float time);   /* prims_get_normal_time, src/prims.c:254-366 */
void  o_prims_sample(const mi_scene_desc *s, mi_primid pi, float r0, float r1, o_hit *hit, float time);
void  o_prims_offset_ray(const o_hit *hit, o_ray *ray);
float o_prims_get_ray(const o_hit *h1, const o_hit *h2, o_ray *ray);

/* oracle_shade.c */
float o_shader_prepare(o_ctx *c, o_path *p, int v);
void  o_prepare_medium(const mi_scene_desc *s, o_path *p, int v, int medium);
float o_shader_sample(o_ctx *c, o_path *p);
float o_shader_brdf(o_ctx *c, o_path *p, int v);
float o_shader_pdf(o_ctx *c, o_path *p, int v);
float o_path_eta_ratio(const o_path *p, int v);
int   o_path_edge_init_volume(o_path *p, int v);
float o_spectrum_eval(const float coeff[3], float lambda);

#endif
