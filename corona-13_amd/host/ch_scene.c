/* ch_scene.c -- scene loading on the host: .nra2 text, .geo binaries, .cam binary, film set-up,
 * material compilation, emitter CDF. Produces the mi_scene_desc the HIP backend consumes.
 *
 * Behaviour follows the reference (all re-stated, nothing copied):
 *   .nra2 grammar            src/shader.c:623-760 (sky line, N shader lines), src/corona_common.c:30-68 (M shape lines)
 *   .geo container           include/prims.h:26-35, src/prims.c:751-831
 *   .cam (legacy 152 B / v1) include/camera.h:13-35,77-99,153-196
 *   film + camera defaults   src/view.c:143-175 (cam_init), 247-296 (32-padding), 921-948 (view_cam_read)
 *   camera frame             src/view.c:903-919, include/quaternion.h
 *   shader chain semantics   src/shaders/mult.c:75-167, color.c:36-82, colorcheckersg.c:207-262,
 *                            dielectric.c:38-58, metal.c:44-67
 *   emitters                 src/shaders/color.c:65-73, src/lights.d/list.c:56-104
 */
#include "ch_internal.h"
#include <ctype.h>
#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { SH_DIFFUSE, SH_COLOR, SH_CHECKER, SH_MULT, SH_DIELECTRIC, SH_METAL, SH_MEDIUM, SH_INTERIOR, SH_EXTERIOR, SH_OTHER };

typedef struct ch_shader
{
  char name[32];
  int  kind;
  /* color / checker */
  uint32_t slot;
  float rgb[3], coeff[3], mul, roughness;
  /* mult */
  int num_pre, pre[MI_MAX_OPS], host;
  /* dielectric: n_d, abbe; metal: table id; medium_rgb: mean cosine (mu_t colour in rgb/coeff/mul); interior: surface, medium in pre[0], host */
  float param[4];
}
ch_shader;

struct ch_scene
{
  mi_scene_desc desc;
  ch_options opt;
  char basename[1024], searchpath[1024];
  int num_shaders;
  ch_shader *shader;
  ch_geo geo;
  mi_primid *primid;
  mi_node *nodes;
  mi_node_aabb *nodes_t1;        /* shutter-close boxes of the nodes, or NULL (no moving primitive) */
  mi_material *materials;
  mi_primid *light_primid;
  float *light_cdf, *light_L;
  float *cie, *checker, *metal;
  float view_gain;
};

static const float *g_cie_table = 0;
const float *ch_cie_table(void) { return g_cie_table; }

/* ---------------------------------------------------------------- data tables */
static float *load_f32(const char *dir, const char *name, size_t count)
{
  char fn[2048];
  snprintf(fn, sizeof(fn), "%s/%s", dir, name);
  FILE *f = fopen(fn, "rb");
  if(!f) return 0;
  float *d = (float *)malloc(count*sizeof(float));
  if(d && fread(d, sizeof(float), count, f) != count) { free(d); d = 0; }
  fclose(f);
  return d;
}

static void default_data_dir(char *out, size_t len)
{
  const char *env = getenv("CORONA_MI_DATA");
  if(env) { snprintf(out, len, "%s", env); return; }
  Dl_info info;
  if(dladdr((void *)&default_data_dir, &info) && info.dli_fname)
  { /* <pkg>/host/libcorona_host.so -> <pkg>/data */
    snprintf(out, len, "%s", info.dli_fname);
    char *c = strrchr(out, '/');
    if(c) *c = 0; else snprintf(out, len, ".");
    strncat(out, "/../data", len - strlen(out) - 1);
    return;
  }
  snprintf(out, len, "data");
}

/* ---------------------------------------------------------------- .nra2: shaders */
static uint32_t parse_slot(char c)
{ /* src/shaders/texture.h tex_parse_slot */
  switch(c)
  {
    case 'd': return MI_SLOT_DIFFUSE;
    case 's': return MI_SLOT_SPECULAR;
    case 'e': return MI_SLOT_EMISSION;
    case 'v': return MI_SLOT_VOLUME;
    case 'g': return MI_SLOT_GLOSSY;
    case 'r': return MI_SLOT_ROUGHNESS;
    default:  return MI_SLOT_UNUSED;
  }
}

static int metal_id(const char *name)
{ /* src/shaders/fresnel.h:21-24,551-558 */
  static const char *mats[] = {"Ti", "Cu", "Fe", "Au", "Ag"};
  for(int i=0;i<5;i++) if(!strcasecmp(name, mats[i])) return i;
  return -1;
}

static int parse_shader_line(ch_scene *s, int id, char *line)
{
  ch_shader *sh = s->shader + id;
  memset(sh, 0, sizeof(*sh));
  char *hash = strchr(line, '#');
  if(hash) *hash = 0;                       /* the plugins' init() stop reading at what they need; rest is comment */
  char name[64] = {0};
  int off = 0;
  if(sscanf(line, "%63s%n", name, &off) < 1) return 1;
  snprintf(sh->name, sizeof(sh->name), "%s", name);
  const char *args = line + off;
  sh->roughness = 1.0f;
  if(!strcmp(name, "diffuse")) sh->kind = SH_DIFFUSE;
  else if(!strcmp(name, "color"))
  {
    sh->kind = SH_COLOR;
    char c = 0;
    if(sscanf(args, " %c %f %f %f %f", &c, sh->rgb, sh->rgb+1, sh->rgb+2, &sh->roughness) < 4)
    {
      fprintf(stderr, "[ch] color: expecting [dgsevr] r g b [roughness] in shader %d\n", id);
      return 1;
    }
    sh->slot = parse_slot(c);
    sh->mul = ch_rgb_to_coeff(sh->rgb, sh->coeff, s->opt.rgb2spec_lut);
  }
  else if(!strcmp(name, "colorcheckersg"))
  {
    sh->kind = SH_CHECKER;
    char c = 'd';
    sscanf(args, " %c %f", &c, &sh->roughness);
    sh->slot = parse_slot(c);
  }
  else if(!strcmp(name, "mult"))
  {
    sh->kind = SH_MULT;
    int n = 0, o = 0, num = 0;
    if(sscanf(args, "%d%n", &num, &o) < 1 || num < 0 || num > MI_MAX_OPS)
    {
      fprintf(stderr, "[ch] mult: bad pre-shader count in shader %d (max %d)\n", id, MI_MAX_OPS);
      return 1;
    }
    args += o;
    sh->num_pre = num;
    for(int k=0;k<num;k++)
    {
      if(sscanf(args, "%d%n", &n, &o) < 1) return 1;
      args += o;
      sh->pre[k] = n < 0 ? id + n : n;      /* negative = relative to ourselves, mult.c:104-106 */
    }
    if(sscanf(args, "%d", &n) < 1) return 1;
    sh->host = n < 0 ? id + n : n;
  }
  else if(!strcmp(name, "dielectric"))
  {
    sh->kind = SH_DIELECTRIC;
    const int i = sscanf(args, "%f %f", sh->param, sh->param+1);
    if(i < 1) { sh->param[0] = 1.5f; sh->param[1] = 50.0f; fprintf(stderr, "[ch] dielectric: expecting n_d [abbe]\n"); return 1; }
    if(i != 2) sh->param[1] = 50.0f;
  }
  else if(!strcmp(name, "metal"))
  {
    sh->kind = SH_METAL;
    char mat[64] = {0};
    if(sscanf(args, "%63s", mat) < 1) return 1;
    int m = metal_id(mat);
    if(m < 0) { fprintf(stderr, "[ch] metal: unknown material `%s', using Ti\n", mat); m = 0; }
    sh->param[0] = (float)m;
  }
  else if(!strcmp(name, "medium_rgb"))
  { /* src/shaders/medium_rgb.c:104-129: mean free paths r g b [dm] -> collision coefficients -> spectrum, mean cosine */
    sh->kind = SH_MEDIUM;
    float mfp[3];
    if(sscanf(args, "%f %f %f %f", mfp, mfp+1, mfp+2, sh->param) != 4)
    { fprintf(stderr, "[ch] medium_rgb: expecting <mean free path r g b> <mean cosine> in shader %d\n", id); return 1; }
    for(int k=0;k<3;k++) sh->rgb[k] = 1.0f/mfp[k];
    sh->mul = ch_rgb_to_coeff(sh->rgb, sh->coeff, s->opt.rgb2spec_lut);
  }
  else if(!strcmp(name, "interior"))
  { /* src/shaders/interior.c:52-72: <surface id> <interior id>, negative = relative */
    sh->kind = SH_INTERIOR;
    int a = 0, b = 0;
    if(sscanf(args, "%d %d", &a, &b) != 2) { fprintf(stderr, "[ch] interior: expecting <surface id> <interior id> in shader %d\n", id); return 1; }
    sh->pre[0] = a < 0 ? id + a : a;
    sh->host = b < 0 ? id + b : b;
  }
  else if(!strcmp(name, "exterior"))
  { /* src/shader.c:699-716: <medium shader id> <light>: the global exterior medium */
    sh->kind = SH_EXTERIOR;
    int light = 0;
    sh->host = -1;
    sscanf(args, "%d %d", &sh->host, &light);
    if(light && sh->host >= 0) { fprintf(stderr, "[ch] exterior: volume lights are outside the scope of this backend (shader %d)\n", id); return 2; }
  }
  else sh->kind = SH_OTHER;                 /* outside the hot-path scope; fine unless a shape uses it */
  return 0;
}

/* compile shader id into a closed material. returns 0 on success */
static int compile_material(const ch_scene *s, int id, mi_material *m)
{
  memset(m, 0, sizeof(*m));
  m->bsdf = MI_BSDF_NONE;
  m->interior = -1;
  const ch_shader *sh = s->shader + id;
  if(sh->kind == SH_INTERIOR)
  { /* the surface's material plus a link to the medium that fills the shape (interior.c:101-118) */
    if(sh->pre[0] < 0 || sh->pre[0] >= s->num_shaders || sh->host < 0 || sh->host >= s->num_shaders) return 1;
    if(s->shader[sh->pre[0]].kind == SH_INTERIOR) return 1;
    mi_material medium;
    if(compile_material(s, sh->host, &medium) || medium.bsdf != MI_BSDF_MEDIUM) return 1;
    if(compile_material(s, sh->pre[0], m) || m->bsdf == MI_BSDF_MEDIUM) { m->bsdf = MI_BSDF_NONE; return 1; }
    m->interior = sh->host;
    return 0;
  }
  const ch_shader *host = sh;
  if(sh->kind == SH_MULT)
  {
    if(sh->host < 0 || sh->host >= s->num_shaders) return 1;
    host = s->shader + sh->host;
    for(int k=0;k<sh->num_pre;k++)
    {
      if(sh->pre[k] < 0 || sh->pre[k] >= s->num_shaders) return 1;
      const ch_shader *p = s->shader + sh->pre[k];
      mi_shade_op *op = m->op + m->num_ops;
      if(p->kind == SH_COLOR)
      {
        op->kind = MI_OP_COLOR; op->slot = p->slot;
        memcpy(op->coeff, p->coeff, sizeof(op->coeff));
        op->mul = p->mul; op->roughness = p->roughness;
      }
      else if(p->kind == SH_CHECKER)
      {
        op->kind = MI_OP_CHECKER; op->slot = p->slot; op->mul = 1.0f; op->roughness = p->roughness;
      }
      else return 1;
      m->num_ops++;
    }
  }
  switch(host->kind)
  {
    case SH_DIFFUSE:    m->bsdf = MI_BSDF_DIFFUSE; break;
    case SH_DIELECTRIC: m->bsdf = MI_BSDF_DIELECTRIC; m->param[0] = host->param[0]; m->param[1] = host->param[1]; break;
    case SH_METAL:      m->bsdf = MI_BSDF_METAL; m->param[0] = host->param[0]; break;
    case SH_MEDIUM:     m->bsdf = MI_BSDF_MEDIUM; memcpy(m->param, host->coeff, 3*sizeof(float)); m->param[3] = host->mul; m->mean_cos = host->param[0]; break;
    default: return 1;
  }
  return 0;
}

/* ---------------------------------------------------------------- .geo */
static int load_geo(ch_scene *s, uint32_t shapeid, const char *name)
{
  char fn[2048];
  snprintf(fn, sizeof(fn), "%s.geo", name);
  FILE *f = fopen(fn, "rb");
  if(!f) { snprintf(fn, sizeof(fn), "%s/%s.geo", s->searchpath, name); f = fopen(fn, "rb"); }
  if(!f) { fprintf(stderr, "[ch] could not load geo `%s'\n", name); return 1; }
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *d = (uint8_t *)malloc(size);
  if(!d || fread(d, 1, size, f) != (size_t)size) { fclose(f); free(d); return 1; }
  fclose(f);
  struct { int32_t magic, version; uint64_t num_prims, vtxidx_offset, vertex_offset; } h;
  if(size < (long)sizeof(h)) { free(d); return 1; }
  memcpy(&h, d, sizeof(h));
  /* the header is untrusted: bound num_prims by the file size (and the backend's 2^26 limit) before any arithmetic on it */
  if(h.magic != 0xc01337 || h.version != 2 || h.num_prims > ((uint64_t)size - sizeof(h))/8 || h.num_prims >= (1ull << 26) ||
     h.vtxidx_offset < sizeof(h) + 8*h.num_prims || h.vertex_offset < h.vtxidx_offset || h.vertex_offset > (uint64_t)size)
  {
    fprintf(stderr, "[ch] geo `%s': bad magic/version/offsets\n", name);
    free(d); return 1;
  }
  const uint64_t nvi = (h.vertex_offset - h.vtxidx_offset)/sizeof(mi_vtxidx);
  const uint64_t nv  = ((uint64_t)size - h.vertex_offset)/sizeof(mi_vtx);
  ch_geo *g = &s->geo;
  mi_shape *sh = g->shapes + shapeid;
  const uint64_t old = s->desc.num_prims;
  if(g->num_vtxidx + nvi >= (1ull << 32) || g->num_vtx + nv >= (1ull << 32) || old + h.num_prims >= (1ull << 26))
  { fprintf(stderr, "[ch] geo `%s': scene too large for this backend\n", name); free(d); return 1; }
  mi_vtxidx *nvidx = (mi_vtxidx *)realloc(g->vtxidx, (g->num_vtxidx + nvi + 1)*sizeof(mi_vtxidx));
  if(nvidx) g->vtxidx = nvidx;
  mi_vtx *nvtx = (mi_vtx *)realloc(g->vtx, (g->num_vtx + nv + 1)*sizeof(mi_vtx));
  if(nvtx) g->vtx = nvtx;
  /* append primids, stamping the shape id (src/prims.c:751-752) */
  mi_primid *nprim = (mi_primid *)realloc(s->primid, (old + h.num_prims + 1)*sizeof(mi_primid));
  if(nprim) s->primid = nprim;
  if(!nvidx || !nvtx || !nprim) { fprintf(stderr, "[ch] geo `%s': out of memory\n", name); free(d); return 1; }
  sh->num_prims = (uint32_t)h.num_prims;
  sh->vtxidx_base = (uint32_t)g->num_vtxidx;
  sh->vtx_base = (uint32_t)g->num_vtx;
  memcpy(g->vtxidx + g->num_vtxidx, d + h.vtxidx_offset, nvi*sizeof(mi_vtxidx));
  memcpy(g->vtx + g->num_vtx, d + h.vertex_offset, nv*sizeof(mi_vtx));
  for(uint64_t k=0;k<h.num_prims;k++)
  {
    mi_primid p; memcpy(&p, d + sizeof(h) + 8*k, 8);
    p = (p & ~(0x1fffffffull << 3)) | ((uint64_t)shapeid << 3);
    const uint32_t vc = MI_PRIMID_VCNT(p), mb = MI_PRIMID_MB(p);
    if(vc < 1 || vc > 4) { fprintf(stderr, "[ch] geo `%s': primitive kind %u unsupported\n", name, vc); free(d); return 1; }
    if(MI_PRIMID_VI(p) + vc > nvi) { fprintf(stderr, "[ch] geo `%s': vertex index out of range\n", name); free(d); return 1; }
    /* a motion-blurred primitive addresses vertex pairs: 2 v = shutter open, 2 v + 1 = shutter close (include/geo.h:108-138) */
    for(uint32_t j=0;j<vc;j++) if((uint64_t)(mb+1)*g->vtxidx[sh->vtxidx_base + MI_PRIMID_VI(p) + j].v + mb >= nv)
    { fprintf(stderr, "[ch] geo `%s': vertex out of range\n", name); free(d); return 1; }
    s->primid[old + k] = p;
  }
  g->num_vtxidx += nvi; g->num_vtx += nv;
  s->desc.num_prims = old + h.num_prims;
  free(d);
  return 0;
}

float ch_prim_area(const ch_geo *g, mi_primid pi)
{ /* src/prims.c:133-154, include/geo/{sphere,line}.h */
  const mi_shape *sh = g->shapes + MI_PRIMID_SHAPE(pi);
  const mi_vtxidx *vi = g->vtxidx + sh->vtxidx_base + MI_PRIMID_VI(pi);
  const mi_vtx *vtx = g->vtx + sh->vtx_base;
  const uint32_t vcnt = MI_PRIMID_VCNT(pi);
  const uint32_t so = MI_PRIMID_MB(pi) + 1;           /* shutter-open state of a moving primitive */
  if(vcnt == MI_PRIM_SPHERE)
  {
    float r; memcpy(&r, &vtx[so*vi[0].v].n, 4);
    return 4.0f*(float)M_PI*r*r;
  }
  if(vcnt == MI_PRIM_LINE)
  {
    const float *v0 = vtx[so*vi[0].v].v, *v1 = vtx[so*vi[1].v].v;
    float r0, r1; memcpy(&r0, &vtx[so*vi[0].v].n, 4); memcpy(&r1, &vtx[so*vi[1].v].n, 4);
    const float d[3] = {v1[0]-v0[0], v1[1]-v0[1], v1[2]-v0[2]};
    const float h = sqrtf(d[0]*d[0]+d[1]*d[1]+d[2]*d[2]);
    const float l = sqrtf(r0*r0 + h*h);
    return (float)M_PI*r1*l - (float)M_PI*r0*l;
  }
  float area = 0.0f;
  const uint32_t st = MI_PRIMID_MB(pi) + 1;           /* shutter-open area of a moving primitive, src/prims.c:140-152 */
  const float *v0 = vtx[st*vi[0].v].v;
  for(uint32_t t=0;t+2<vcnt;t++)
  {
    const float *v1 = vtx[st*vi[t+1].v].v, *v2 = vtx[st*vi[t+2].v].v;
    float e1[3], e2[3], n[3];
    for(int k=0;k<3;k++) { e1[k] = v1[k]-v0[k]; e2[k] = v2[k]-v0[k]; }
    n[0] = e1[1]*e2[2] - e2[1]*e1[2];
    n[1] = e1[2]*e2[0] - e2[2]*e1[0];
    n[2] = e1[0]*e2[1] - e2[0]*e1[1];
    area += sqrtf(n[0]*n[0]+n[1]*n[1]+n[2]*n[2])*.5f;
  }
  return area;
}

/* ---------------------------------------------------------------- camera */
typedef struct quat_t { float w, x[3]; } quat_t;

static void quat_mul(quat_t *a, const quat_t *p)
{ /* Hamilton product a*p */
  const quat_t r = *a;
  a->x[0] = r.w*p->x[0] + r.x[0]*p->w    + r.x[1]*p->x[2] - r.x[2]*p->x[1];
  a->x[1] = r.w*p->x[1] - r.x[0]*p->x[2] + r.x[1]*p->w    + r.x[2]*p->x[0];
  a->x[2] = r.w*p->x[2] + r.x[0]*p->x[1] - r.x[1]*p->x[0] + r.x[2]*p->w;
  a->w    = r.w*p->w    - r.x[0]*p->x[0] - r.x[1]*p->x[1] - r.x[2]*p->x[2];
}

static void quat_rotate(const quat_t *q, float *v)
{ /* q v q' */
  quat_t vq = {0.0f, {v[0], v[1], v[2]}}, inv = *q, res = *q;
  for(int k=0;k<3;k++) inv.x[k] = -inv.x[k];
  quat_mul(&res, &vq);
  quat_mul(&res, &inv);
  for(int k=0;k<3;k++) v[k] = res.x[k];
}

static void normalise3(float *v)
{
  const float il = 1.0f/sqrtf(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]);
  for(int k=0;k<3;k++) v[k] *= il;
}

static const float f_stop_tab[] = {0.5f, 0.7f, 1.0f, 1.4f, 2, 2.8f, 4, 5.6f, 8, 11, 16, 22, 32, 45, 64, 90, 128};
static const float exposure_tab[] = {60.0f, 30.0f, 15.0f, 8.0f, 4.0f, 2.0f, 1.0f, 0.5f, 1.0f/4.0f, 1.0f/8.0f,
  1.0f/15.0f, 1.0f/30.0f, 1.0f/60.0f, 1.0f/125.0f, 1.0f/250.0f, 1.0f/500.0f, 1.0f/1000.0f, 1.0f/2000.0f, 1.0f/4000.0f, 1.0f/8000.0f};

static int load_camera(ch_scene *s)
{
  /* defaults, src/view.c:143-175 */
  float pos[3] = {0.0f, -10.0f, 1.0f}, pos1[3] = {0.0f, -10.0f, 1.0f};
  const float ax[3] = {0, 1.0f/sqrtf(2.0f), 1.0f/sqrtf(2.0f)};
  quat_t q = { cosf((float)M_PI/2.0f), {sinf((float)M_PI/2.0f)*ax[0], sinf((float)M_PI/2.0f)*ax[1], sinf((float)M_PI/2.0f)*ax[2]} }, q1 = q;
  float focus = 10.0f, crop = 1.0f, focal = 0.5f, iso = 100.0f;
  int av = 9, tv = 13;

  char fn[2048];
  if(s->opt.cam_file) snprintf(fn, sizeof(fn), "%s", s->opt.cam_file);
  else snprintf(fn, sizeof(fn), "%s01.cam", s->basename);
  FILE *f = fopen(fn, "rb");
  if(f)
  {
    uint8_t buf[160];
    const size_t n = fread(buf, 1, sizeof(buf), f);
    fclose(f);
    if(n == 152)
    { /* legacy layout, include/camera.h:77-99: note crop_factor is NOT taken over (camera.h:170-182) */
      const float *fl = (const float *)buf; const int32_t *il = (const int32_t *)buf;
      memcpy(pos, fl+1, 12); memcpy(&q, fl+4, 16);
      iso = fl[16]; memcpy(&q1, fl+17, 16); memcpy(pos1, fl+21, 12);
      focus = fl[29]; av = il[34]; focal = fl[35]; tv = il[37];
    }
    else if(n == 104 && !memcmp(buf, "CCAM", 4) && ((const int32_t *)buf)[1] == 1)
    { /* camera_t, include/camera.h:13-35 */
      const float *fl = (const float *)buf; const int32_t *il = (const int32_t *)buf;
      memcpy(pos, fl+2, 12); memcpy(pos1, fl+5, 12); memcpy(&q, fl+8, 16); memcpy(&q1, fl+12, 16);
      focus = fl[18]; crop = fl[21]; av = il[22]; tv = il[23]; focal = fl[24]; iso = fl[25];
    }
    else fprintf(stderr, "[ch] camera file `%s' has unknown format, using default camera\n", fn);
  }
  else if(s->opt.verbose) fprintf(stderr, "[ch] no camera file `%s', using default camera\n", fn);

  if(tv < 0 || tv > 20) tv = 13;                      /* src/view.c:926 */
  if(tv >= 20) tv = 19;
  if(av < 0 || av >= 17) av = 9;
  if(iso < 1 || iso > 409600) iso = 100;
  if(s->opt.iso > 0) iso = s->opt.iso;
  mi_camera *c = &s->desc.cam;
  /* shutter-open and shutter-close state differ: camera motion blur, the frame is interpolated per path (src/view.c:903-919) */
  c->moving = (memcmp(pos, pos1, 12) || memcmp(&q, &q1, 16)) ? 1 : 0;
  memcpy(c->pos_t1, pos1, 12);
  c->orient[0] = q.w; memcpy(c->orient + 1, q.x, 12);
  c->orient_t1[0] = q1.w; memcpy(c->orient_t1 + 1, q1.x, 12);
  const float w = (float)s->desc.width, h = (float)s->desc.height;
  if(s->desc.width > s->desc.height) { c->film_width = 0.35f/crop; c->film_height = h/w*c->film_width; }
  else                               { c->film_height = 0.35f/crop; c->film_width = w/h*c->film_height; }
  memcpy(c->pos, pos, 12);
  float a[3] = {1,0,0}, b[3] = {0,1,0}, nn[3] = {0,0,1};
  quat_rotate(&q, a); quat_rotate(&q, b); quat_rotate(&q, nn);
  normalise3(a); normalise3(b); normalise3(nn);
  memcpy(c->a, a, 12); memcpy(c->b, b, 12); memcpy(c->n, nn, 12);
  c->focus = focus; c->focal_length = focal;
  c->f_stop = f_stop_tab[av];
  c->exposure_time = exposure_tab[tv];
  c->iso = iso;
  c->time_scale = fminf(1.0f, c->exposure_time/(1.0f/30.0f));   /* src/view.c:881-891 */
  return 0;
}

/* ---------------------------------------------------------------- lights */
/* must run before the BVH build permutes s->primid (emitter prims are listed in shape/file order) */
static int init_lights(ch_scene *s)
{
  uint32_t total = 0;
  for(uint32_t sid=0;sid<s->geo.num_shapes;sid++)
  {
    const mi_material *m = s->materials + s->geo.shapes[sid].material;
    for(uint32_t k=0;k<m->num_ops;k++) if(m->op[k].kind == MI_OP_COLOR && m->op[k].slot == MI_SLOT_EMISSION)
      total += s->geo.shapes[sid].num_prims;
  }
  s->light_primid = (mi_primid *)calloc(total + 1, sizeof(mi_primid));
  s->light_cdf = (float *)calloc(total + 1, sizeof(float));
  s->light_L = (float *)calloc(total + 1, sizeof(float));
  if(!s->light_primid || !s->light_cdf || !s->light_L) return MI_ERR_NOMEM;
  uint32_t off = 0;
  uint64_t prim_base = 0;
  for(uint32_t sid=0;sid<s->geo.num_shapes;sid++)
  {
    const mi_shape *sh = s->geo.shapes + sid;
    const mi_material *m = s->materials + sh->material;
    for(uint32_t k=0;k<m->num_ops;k++) if(m->op[k].kind == MI_OP_COLOR && m->op[k].slot == MI_SLOT_EMISSION)
    {
      const float *c = m->op[k].coeff;
      /* color.c:65-73: mean of the spectrum at four probe wavelengths times the scale */
      const float L = m->op[k].mul*(ch_coeff_eval(c, 400.0f) + ch_coeff_eval(c, 480.0f) + ch_coeff_eval(c, 560.0f) + ch_coeff_eval(c, 660.0f))/4.0f;
      for(uint32_t i=0;i<sh->num_prims;i++)
      { /* list.c:56-74 */
        s->light_primid[off+i] = s->primid[prim_base + i];
        s->light_cdf[off+i] = ch_prim_area(&s->geo, s->primid[prim_base + i])*L;
        s->light_L[off+i] = L;
      }
      off += sh->num_prims;
    }
    prim_base += sh->num_prims;
  }
  mi_lights *l = &s->desc.lights;
  l->num_prims = total;
  l->primid = s->light_primid; l->cdf = s->light_cdf; l->L = s->light_L;
  /* list.c:76-104 (sky is black in scope, no volume lights) */
  l->p_sky = 0.0f; l->p_vol = 0.0f; l->p_geo = total ? 1.0f : 0.0f;
  if(!total) return 0;
  float sum = 0.0f;
  for(uint32_t k=0;k<total;k++) sum += s->light_cdf[k];
  for(uint32_t k=0;k<total;k++) s->light_L[k] /= sum;
  for(uint32_t k=1;k<total;k++) s->light_cdf[k] += s->light_cdf[k-1];
  for(uint32_t k=0;k+1<total;k++) s->light_cdf[k] /= s->light_cdf[total-1];
  s->light_cdf[total-1] = 1.0f;
  return 0;
}

/* ---------------------------------------------------------------- per-scene coefficient cache */
/* optional "<basename>.rgb2spec" next to the scene: lines "r g b  c0 c1 c2 mul" that pin the
 * coefficients of `color` shaders with exactly that rgb (our extension; lets a scene carry the
 * numbers of a particular LUT without shipping the 9.4 MB table). */
static void apply_coeff_cache(ch_scene *s)
{
  const char *mode = getenv("CORONA_MI_RGB2SPEC");      /* "fit": ignore the file, use the host's own fit (tests of that path) */
  if(mode && !strcmp(mode, "fit")) return;
  char fn[2048];
  snprintf(fn, sizeof(fn), "%s.rgb2spec", s->basename);
  FILE *f = fopen(fn, "rb");
  if(!f) return;
  char line[512];
  while(fgets(line, sizeof(line), f))
  {
    float rgb[3], c[3], mul;
    if(line[0] == '#' || sscanf(line, "%f %f %f %f %f %f %f", rgb, rgb+1, rgb+2, c, c+1, c+2, &mul) != 7) continue;
    for(int i=0;i<s->num_shaders;i++) if((s->shader[i].kind == SH_COLOR || s->shader[i].kind == SH_MEDIUM) &&
        s->shader[i].rgb[0] == rgb[0] && s->shader[i].rgb[1] == rgb[1] && s->shader[i].rgb[2] == rgb[2])
    {
      memcpy(s->shader[i].coeff, c, sizeof(c));
      s->shader[i].mul = mul;
    }
  }
  fclose(f);
}

static int compile_all_materials(ch_scene *s)
{
  for(int i=0;i<s->num_shaders;i++) compile_material(s, i, s->materials + i);
  /* the last `exterior` line wins (src/shader.c:699-716) */
  s->desc.exterior = 0;
  for(int i=0;i<s->num_shaders;i++) if(s->shader[i].kind == SH_EXTERIOR)
  {
    const int m = s->shader[i].host;
    if(m < 0) { s->desc.exterior = 0; continue; }
    if(m >= i || s->materials[m].bsdf != MI_BSDF_MEDIUM)
    { fprintf(stderr, "[ch] exterior: shader %d is not a medium defined before line %d\n", m, i); return MI_ERR_UNSUPPORTED; }
    s->desc.exterior = (uint32_t)m + 1;
  }
  for(uint32_t sid=0;sid<s->geo.num_shapes;sid++)
  {
    const int m = s->geo.shapes[sid].material;
    if(s->materials[m].bsdf == MI_BSDF_NONE || s->materials[m].bsdf == MI_BSDF_MEDIUM)
    {
      fprintf(stderr, "[ch] shape %u uses shader %d (`%s'), which is outside the scope of this backend\n", sid, m, s->shader[m].name);
      return MI_ERR_UNSUPPORTED;
    }
    for(uint32_t k=0;k<s->materials[m].num_ops;k++)
      if(s->materials[m].op[k].kind == MI_OP_CHECKER && !s->checker)
      { fprintf(stderr, "[ch] colorcheckersg needs colorchecker_sg.f32 in the data directory\n"); return MI_ERR_ARG; }
    if(s->materials[m].bsdf == MI_BSDF_METAL && !s->metal)
    { fprintf(stderr, "[ch] metal needs metal_ior.f32 in the data directory\n"); return MI_ERR_ARG; }
  }
  return 0;
}

/* ---------------------------------------------------------------- entry points */
static int read_line(FILE *f, char *line, size_t len)
{
  if(!fgets(line, (int)len, f)) return 1;
  size_t n = strlen(line);
  while(n && (line[n-1] == '\n' || line[n-1] == '\r')) line[--n] = 0;
  return 0;
}

int ch_scene_load(const char *nra2_path, const ch_options *opt_in, ch_scene **out)
{
  ch_scene *s = (ch_scene *)calloc(1, sizeof(ch_scene));
  if(!s) return MI_ERR_NOMEM;
  if(opt_in) s->opt = *opt_in;
  int err = MI_ERR_ARG;
  FILE *f = fopen(nra2_path, "rb");
  if(!f) { fprintf(stderr, "[ch] can't open %s for reading\n", nra2_path); free(s); return MI_ERR_ARG; }

  /* basename / searchpath, src/main.c:279-286 */
  snprintf(s->searchpath, sizeof(s->searchpath), "%s", nra2_path);
  char *c = strrchr(s->searchpath, '/');
  if(c) *c = 0; else snprintf(s->searchpath, sizeof(s->searchpath), ".");
  snprintf(s->basename, sizeof(s->basename), "%s", nra2_path);
  c = strrchr(s->basename, '.');
  if(c && c != s->basename && !strchr(c, '/')) *c = 0;

  mi_scene_desc *d = &s->desc;
  d->struct_size = sizeof(mi_scene_desc);
  d->abi_version = MI_ABI_VERSION;
  d->width  = s->opt.width  ? s->opt.width  : 1024;
  d->height = s->opt.height ? s->opt.height : 576;
  while(d->width  & 0x1f) d->width++;                 /* src/view.c:294-296 */
  while(d->height & 0x1f) d->height++;
  d->max_verts = s->opt.max_verts ? s->opt.max_verts : 32;
  d->sampler = s->opt.sampler;
  d->pointsampler = s->opt.pointsampler;
  d->frame = s->opt.frame ? s->opt.frame : 1;
  s->view_gain = 1.0f;

  char ddir[2048];
  if(s->opt.data_dir) snprintf(ddir, sizeof(ddir), "%s", s->opt.data_dir);
  else default_data_dir(ddir, sizeof(ddir));
  s->cie = load_f32(ddir, "cie1931_xyz.f32", 96*3);
  s->checker = load_f32(ddir, "colorchecker_sg.f32", 140*36);
  s->metal = load_f32(ddir, "metal_ior.f32", 5*95*2);
  if(!s->cie) { fprintf(stderr, "[ch] missing %s/cie1931_xyz.f32 (set CORONA_MI_DATA)\n", ddir); goto fail; }
  g_cie_table = s->cie;
  d->cie_xyz = s->cie; d->checker = s->checker; d->metal_ior = s->metal;

  char line[4096];
  /* sky */
  if(read_line(f, line, sizeof(line))) goto fail;
  if(strncmp(line, "black", 5))
  {
    fprintf(stderr, "[ch] sky `%s': only `black' is inside the scope of this backend\n", line);
    err = MI_ERR_UNSUPPORTED; goto fail;
  }
  /* shaders */
  if(read_line(f, line, sizeof(line)) || sscanf(line, "%d", &s->num_shaders) != 1 || s->num_shaders < 0 || s->num_shaders > 4096) goto fail;
  s->shader = (ch_shader *)calloc(s->num_shaders + 1, sizeof(ch_shader));
  s->materials = (mi_material *)calloc(s->num_shaders + 1, sizeof(mi_material));
  for(int i=0;i<s->num_shaders;i++)
  {
    if(read_line(f, line, sizeof(line))) goto fail;
    if(parse_shader_line(s, i, line)) { fprintf(stderr, "[ch] could not parse shader line %d\n", i); goto fail; }
  }
  apply_coeff_cache(s);
  /* shapes */
  int num_shapes = 0;
  if(read_line(f, line, sizeof(line)) || sscanf(line, "%d", &num_shapes) != 1 || num_shapes < 0) goto fail;
  s->geo.shapes = (mi_shape *)calloc(num_shapes + 1, sizeof(mi_shape));
  for(int i=0;i<num_shapes;i++)
  {
    int shader = 0; char geo[1024], tex[512];
    if(read_line(f, line, sizeof(line))) break;
    if(sscanf(line, "%d %1023s %511s", &shader, geo, tex) < 2)
    { fprintf(stderr, "[ch] WARN: malformed shape line: %s\n", line); continue; }
    if(shader < 0 || shader >= s->num_shaders)
    { fprintf(stderr, "[ch] WARN: shader %d of shape %d out of bounds, using 0\n", shader, i); shader = 0; }
    const uint32_t sid = s->geo.num_shapes;
    if(load_geo(s, sid, geo)) continue;               /* failed shapes are skipped explicitly (src/prims.c:783-788) */
    s->geo.shapes[sid].material = shader;
    s->geo.num_shapes++;
  }
  fclose(f); f = 0;
  if(s->geo.num_shapes > 255) { fprintf(stderr, "[ch] more than 255 shapes are outside the scope of this backend\n"); err = MI_ERR_UNSUPPORTED; goto fail; }

  if((err = compile_all_materials(s))) goto fail;
  if((err = init_lights(s))) goto fail;
  if((err = load_camera(s))) goto fail;
  if((err = ch_qbvh_build(&s->geo, s->primid, d->num_prims, &s->nodes, &d->num_nodes, d->aabb, &s->nodes_t1))) goto fail;

  d->nodes = s->nodes; d->nodes_t1 = s->nodes_t1; d->primid = s->primid;
  d->num_shapes = s->geo.num_shapes; d->shapes = s->geo.shapes;
  d->num_vtxidx = s->geo.num_vtxidx; d->vtxidx = s->geo.vtxidx;
  d->num_vtx = s->geo.num_vtx; d->vtx = s->geo.vtx;
  d->num_materials = s->num_shaders; d->materials = s->materials;
  if(s->opt.verbose)
    fprintf(stderr, "[ch] %s: %lu prims, %u shapes, %u nodes, %u emitter prims, film %ux%u, max verts %u\n",
        s->basename, (unsigned long)d->num_prims, d->num_shapes, d->num_nodes, d->lights.num_prims, d->width, d->height, d->max_verts);
  *out = s;
  return MI_OK;
fail:
  if(f) fclose(f);
  ch_scene_free(s);
  return err ? err : MI_ERR_ARG;
}

const mi_scene_desc *ch_scene_desc(const ch_scene *s) { return &s->desc; }
int ch_scene_num_shaders(const ch_scene *s) { return s->num_shaders; }
const char *ch_scene_shader_name(const ch_scene *s, int i) { return (i >= 0 && i < s->num_shaders) ? s->shader[i].name : ""; }

int ch_scene_set_color_coeff(ch_scene *s, int id, const float coeff[3], float mul)
{
  if(id < 0 || id >= s->num_shaders || (s->shader[id].kind != SH_COLOR && s->shader[id].kind != SH_MEDIUM)) return MI_ERR_ARG;
  memcpy(s->shader[id].coeff, coeff, 3*sizeof(float));
  s->shader[id].mul = mul;
  for(int i=0;i<s->num_shaders;i++) compile_material(s, i, s->materials + i);
  /* emitter weights depend on the coefficients; rebuild the list from the (permuted) primid array is not
   * possible, but L only scales all prims of a shape uniformly: recompute L and the cdf in place. */
  mi_lights *l = &s->desc.lights;
  if(l->num_prims)
  {
    float sum = 0.0f;
    for(uint32_t k=0;k<l->num_prims;k++)
    {
      const mi_material *m = s->materials + s->geo.shapes[MI_PRIMID_SHAPE(s->light_primid[k])].material;
      float L = 0.0f;
      for(uint32_t o=0;o<m->num_ops;o++) if(m->op[o].kind == MI_OP_COLOR && m->op[o].slot == MI_SLOT_EMISSION)
      {
        const float *c = m->op[o].coeff;
        L = m->op[o].mul*(ch_coeff_eval(c, 400.0f) + ch_coeff_eval(c, 480.0f) + ch_coeff_eval(c, 560.0f) + ch_coeff_eval(c, 660.0f))/4.0f;
        break;
      }
      s->light_L[k] = L;
      s->light_cdf[k] = ch_prim_area(&s->geo, s->light_primid[k])*L;
      sum += s->light_cdf[k];
    }
    for(uint32_t k=0;k<l->num_prims;k++) s->light_L[k] /= sum;
    for(uint32_t k=1;k<l->num_prims;k++) s->light_cdf[k] += s->light_cdf[k-1];
    for(uint32_t k=0;k+1<l->num_prims;k++) s->light_cdf[k] /= s->light_cdf[l->num_prims-1];
    s->light_cdf[l->num_prims-1] = 1.0f;
  }
  return MI_OK;
}

float ch_scene_gain(const ch_scene *s, uint64_t spp)
{ /* src/view.c:651-657 */
  return s->view_gain * s->desc.cam.iso / (100.0f * (float)(spp ? spp : 1));
}

void ch_scene_free(ch_scene *s)
{
  if(!s) return;
  if(g_cie_table == s->cie) g_cie_table = 0;
  free(s->shader); free(s->geo.shapes); free(s->geo.vtxidx); free(s->geo.vtx);
  free(s->primid); free(s->nodes); free(s->nodes_t1); free(s->materials);
  free(s->light_primid); free(s->light_cdf); free(s->light_L);
  free(s->cie); free(s->checker); free(s->metal);
  free(s);
}
