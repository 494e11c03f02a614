/* oracle/refharness/render_dump.c -- TEST INFRASTRUCTURE, build-container only.
 *
 * Our own MOD_render module for the *real* reference (hanatos/corona-13): it
 * implements the reference's render interface (include/render.h:13-28) the way
 * src/render.d/gi.c:81-110 does (reseed RNG per index, pointsampler_mutate, plain MC,
 * splat through view_splat) and additionally appends a fixed-size binary record of
 * every finished path with index < CORONA_DUMP_N to the file CORONA_DUMP_FILE.
 * Linked by oracle/Makefile against the reference's sources where they lie; used by
 * tests/golden/make_golden.py to produce the per-path golden vectors that pin
 * oracle/ (our CPU restatement) and, through it, the HIP path.
 *
 * Run with MOD_points=xorshift128p and `-t 1` (bit-reproducible).
 */
#include "corona_common.h"
#include "render.h"
#include "points.h"
#include "pointsampler.h"
#include "threads.h"
#include "pathspace.h"
#include "view.h"
#include "spectrum.h"
#include <float.h>

#include "prims.h"
#include "accel.h"
#include <pthread.h>

/* Layout of the reference's private accel_t / qbvh_node_t (src/accel.d/qbvhmp.c:62-81,175-193, motion-blur
 * build): declared here only to *read* the finished tree for the golden topology fixture. */
typedef struct { float4_t aabb0[6]; float4_t aabb1[6]; uint64_t child[4]; uint64_t parent; int64_t axis0, axis00, axis01; } ref_node_t;
typedef struct { void *queue; uint64_t built; pthread_mutex_t mutex; float *prim_aabb; uint64_t num_nodes; uint64_t node_bufsize;
                 float aabb[6]; ref_node_t *tree; struct prims_t *prims; } ref_accel_t;

static void dump_tree(const char *fn)
{
  const ref_accel_t *a = (const ref_accel_t *)rt.accel;
  FILE *f = fopen(fn, "wb");
  if(!f) return;
  uint64_t hdr[3] = { 0x65657274ull /* 'tree' */, a->num_nodes, rt.prims->num_prims };
  fwrite(hdr, sizeof(hdr), 1, f);
  fwrite(a->aabb, sizeof(float), 6, f);
  for(uint64_t n=0;n<a->num_nodes;n++)
  {
    const ref_node_t *nd = a->tree + n;
    float box[6][4];
    for(int k=0;k<6;k++) for(int c=0;c<4;c++) box[k][c] = nd->aabb0[k].f[c];
    fwrite(box, sizeof(box), 1, f);
    fwrite(nd->child, sizeof(uint64_t), 4, f);
    int32_t ax[4] = { (int32_t)nd->axis0, (int32_t)nd->axis00, (int32_t)nd->axis01, (int32_t)nd->parent };
    fwrite(ax, sizeof(ax), 1, f);
  }
  fwrite(rt.prims->primid, sizeof(primid_t), rt.prims->num_prims, f);
  /* appended (readers of the older layout stop above): the shutter-close boxes of every node, qbvh_node_t.aabb1 */
  for(uint64_t n=0;n<a->num_nodes;n++)
  {
    float box[6][4];
    for(int k=0;k<6;k++) for(int c=0;c<4;c++) box[k][c] = a->tree[n].aabb1[k].f[c];
    fwrite(box, sizeof(box), 1, f);
  }
  fclose(f);
}

#define DUMP_MAX_VERTS 8      /* vertices recorded per path (first 8) */
#define DUMP_MAX_SPLATS 8

typedef struct dump_vertex_t
{
  uint64_t prim;              /* raw primid_t bits */
  float dist;                 /* e[v].dist */
  float x[3], n[3], gn[3];
  float omega[3];             /* e[v].omega */
  uint32_t mode, flags;
  float throughput, pdf;
  float u, v;
  float rd, rg, em, roughness;
  float eta;                  /* diffgeo.eta */
  int32_t shader;
}
dump_vertex_t;                /* 112 bytes */

typedef struct dump_splat_t
{
  int32_t length;             /* path->length when splatted */
  int32_t tech;               /* v[length-1].tech */
  float value;                /* spectral value handed to render_splat */
  float col[3];               /* camera (XYZ) colour that view_splat accumulates */
}
dump_splat_t;                 /* 24 bytes */

typedef struct dump_rec_t
{
  uint64_t index;
  float pixel_i, pixel_j, lambda, time, scramble, throughput;
  int32_t length, num_splats;
  dump_splat_t splat[DUMP_MAX_SPLATS];
  dump_vertex_t v[DUMP_MAX_VERTS];
}
dump_rec_t;

/* hero wavelengths (-DMF_COUNT=4, `make -C oracle mf4`): the record above takes the hero component (index 0) of every spectral quantity; the
   extension behind it takes all MF_COUNT components. MF_COUNT = 1: no extension, the file format of rounds 1-4. */
#if MF_COUNT > 1
typedef struct dump_ext_t
{
  float lambda[MF_COUNT];
  float throughput[DUMP_MAX_VERTS][MF_COUNT], pdf[DUMP_MAX_VERTS][MF_COUNT];
  float rd[DUMP_MAX_VERTS][MF_COUNT], rg[DUMP_MAX_VERTS][MF_COUNT], em[DUMP_MAX_VERTS][MF_COUNT], eta[DUMP_MAX_VERTS][MF_COUNT];
  float splat_value[DUMP_MAX_SPLATS][MF_COUNT];
}
dump_ext_t;
#endif

typedef struct render_t
{
  atomic_int_fast32_t clear_tls;
  uint64_t dump_n;
  FILE *dump_file;
}
render_t;

typedef struct render_tls_t
{
  path_t path0, path1;
  path_t *curr_path, *tent_path;
  dump_rec_t rec;
#if MF_COUNT > 1
  dump_ext_t ext;
#endif
}
render_tls_t;

static void *clear_tls(void *arg)
{
  path_init(rt_tls.render->curr_path, 0, 0);
  path_init(rt_tls.render->tent_path, 0, 0);
  rt.render->clear_tls++;
  while(rt.render->clear_tls < rt.num_threads) sched_yield();
  return 0;
}

render_t *render_init()
{
  render_t *r = (render_t *)common_alloc(256, sizeof(render_t));
  memset(r, 0, sizeof(*r));
  const char *n = getenv("CORONA_DUMP_N"), *fn = getenv("CORONA_DUMP_FILE");
  r->dump_n = n ? strtoull(n, 0, 10) : 0;
  r->dump_file = (fn && r->dump_n) ? fopen(fn, "wb") : 0;
  if(r->dump_file)
  {
#if MF_COUNT > 1
    uint32_t hdr[4] = { 0x70647263u /* 'crdp' */, (uint32_t)(sizeof(dump_rec_t) + sizeof(dump_ext_t)), DUMP_MAX_VERTS | (MF_COUNT << 16), PATHSPACE_MAX_VERTS };
#else
    uint32_t hdr[4] = { 0x70647263u /* 'crdp' */, (uint32_t)sizeof(dump_rec_t), DUMP_MAX_VERTS, PATHSPACE_MAX_VERTS };
#endif
    fwrite(hdr, sizeof(hdr), 1, r->dump_file);
  }
  return r;
}

void render_cleanup(render_t *r)
{
  if(r->dump_file) fclose(r->dump_file);
  free(r);
}

render_tls_t *render_tls_init()
{
  render_tls_t *r = (render_tls_t *)common_alloc(256, sizeof(render_tls_t));
  path_init(&r->path0, 0, 0);
  path_init(&r->path1, 0, 0);
  r->curr_path = &r->path0;
  r->tent_path = &r->path1;
  return r;
}

void render_tls_cleanup(render_tls_t *r) { free(r); }

void render_clear()
{
  threads_t *t = rt.threads;
  rt.render->clear_tls = 0;
  for(int k=0;k<rt.num_threads;k++)
    pthread_pool_task_init(t->task + k, &t->pool, clear_tls, t);
  pthread_pool_wait(&t->pool);
}

void render_print_info(FILE *fd)
{
  fprintf(fd, "render   : global illumination (path dump harness)\n");
}

void render_sample_path(uint64_t index)
{
  render_tls_t *tls = rt_tls.render;
  path_t *tent = tls->tent_path;
  const int dump = rt.render->dump_file && index < rt.render->dump_n;
  tent->index = index;
  if(index == 0 && getenv("CORONA_DUMP_TREE")) dump_tree(getenv("CORONA_DUMP_TREE"));
  if(dump) { memset(&tls->rec, 0, sizeof(tls->rec)); tls->rec.index = index; }
#if MF_COUNT > 1
  if(dump) memset(&tls->ext, 0, sizeof(tls->ext));
#endif
  points_set_state(rt.points, common_get_threadid(), index, rt.anim_frame);
  pointsampler_mutate(tls->curr_path, tent);
  if(dump)
  {
    dump_rec_t *r = &tls->rec;
    r->pixel_i = tent->sensor.pixel_i;
    r->pixel_j = tent->sensor.pixel_j;
    r->lambda = mf(tent->lambda, 0);
    r->time = tent->time;
    r->scramble = tent->tangent_frame_scrambling;
    r->throughput = mf(tent->throughput, 0);
    r->length = tent->length;
    for(int v=0;v<tent->length && v<DUMP_MAX_VERTS;v++)
    {
      dump_vertex_t *d = r->v + v;
      const vertex_t *s = tent->v + v;
      memcpy(&d->prim, &s->hit.prim, 8);
      d->dist = tent->e[v].dist;
      for(int k=0;k<3;k++) { d->x[k] = s->hit.x[k]; d->n[k] = s->hit.n[k]; d->gn[k] = s->hit.gn[k]; d->omega[k] = tent->e[v].omega[k]; }
      d->mode = s->mode; d->flags = s->flags;
      d->throughput = mf(s->throughput, 0); d->pdf = mf(s->pdf, 0);
      d->u = s->hit.u; d->v = s->hit.v;
      d->rd = mf(s->shading.rd, 0); d->rg = mf(s->shading.rg, 0); d->em = mf(s->shading.em, 0); d->roughness = s->shading.roughness;
      d->eta = mf(s->diffgeo.eta, 0);
#if MF_COUNT > 1
      for(int l=0;l<MF_COUNT;l++)
      {
        tls->ext.throughput[v][l] = mf(s->throughput, l); tls->ext.pdf[v][l] = mf(s->pdf, l);
        tls->ext.rd[v][l] = mf(s->shading.rd, l); tls->ext.rg[v][l] = mf(s->shading.rg, l); tls->ext.em[v][l] = mf(s->shading.em, l);
        tls->ext.eta[v][l] = mf(s->diffgeo.eta, l);
      }
#endif
      d->shader = s->hit.shader;
    }
    fwrite(r, sizeof(*r), 1, rt.render->dump_file);
#if MF_COUNT > 1
    for(int l=0;l<MF_COUNT;l++) tls->ext.lambda[l] = mf(tent->lambda, l);
    fwrite(&tls->ext, sizeof(tls->ext), 1, rt.render->dump_file);
#endif
  }
  /* plain MC: pointsampler_accept() == 0 for MOD_pointsampler=rand, nothing to swap */
}

void render_splat(const path_t *p, const mf_t value)
{
  render_tls_t *tls = rt_tls.render;
  if(rt.render->dump_file && p->index < rt.render->dump_n && tls->rec.num_splats < DUMP_MAX_SPLATS)
  {
    dump_splat_t *s = tls->rec.splat + tls->rec.num_splats++;
    s->length = p->length;
    s->tech = p->length ? p->v[p->length-1].tech : -1;
    s->value = mf(value, 0);
    s->col[0] = s->col[1] = s->col[2] = 0.0f;
#if MF_COUNT > 1
    for(int l=0;l<MF_COUNT;l++) tls->ext.splat_value[tls->rec.num_splats-1][l] = mf(value, l);
    /* same filter as view_splat (src/view.c:455-463) */
    if(mf_any(mf_gt(value, mf_set1(0.0f))) && mf_all(mf_lt(value, mf_set1(FLT_MAX))) && mf_all(mf_eq(value, value)))
      spectrum_p_to_camera(p->lambda, value, s->col);
#else
    /* same filter as view_splat (src/view.c:455-463) */
    if(value > 0.0f && value < FLT_MAX && value == value)
      spectrum_p_to_camera(p->lambda, value, s->col);
#endif
  }
  view_splat(p, value);
}
