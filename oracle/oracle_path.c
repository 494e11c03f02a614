/* oracle_path.c -- TEST INFRASTRUCTURE: CPU restatement of the path-level part of the reference
 * hot path (see oracle.h): per-path RNG, thin-lens camera, path_extend / path_propagate, the pt and
 * ptdl samplers, next-event estimation, Blackman-Harris splat, and the public oracle_* entry points.
 */
#include "o_core.h"
#include <pthread.h>
#include <stdlib.h>
#include <sys/time.h>

/* ---------------------------------------------------------------- RNG */
float o_rand(o_ctx *c)
{ /* points_rand, src/points.d/xorshift128p.c:61-74 */
  uint64_t s1 = c->rng0;
  const uint64_t s0 = c->rng1;
  c->rng0 = s0;
  s1 ^= s1 << 23;
  s1 ^= s1 >> 17;
  s1 ^= s0;
  s1 ^= s0 >> 26;
  c->rng1 = s1;
  const uint32_t v = 0x3f800000u | (uint32_t)((c->rng0 + c->rng1) >> 41);
  float f; memcpy(&f, &v, 4);
  return f - 1.0f;
}

static int o_pixels_from_index = 0;       /* oracle_set_pixels_from_index: 0 the pixel is sampled; 1 it is given by the path's index exactly as the reference's
                                             tiled branch does it (gi.c:88-95: integer pixel, the reference's seeding); 2 MI_PIXELS_FROM_INDEX as the product
                                             defines it (corona_mi.h): the same pixel, a position inside it from the path's own two image-plane numbers,
                                             hashed seeding */
void oracle_set_pixels_from_index(int mode) { o_pixels_from_index = mode; }

void o_rand_seed(o_ctx *c, uint64_t index, uint64_t frame)
{ /* points_set_state, src/points.d/xorshift128p.c:53-59, called from render_sample_path with
     (tid, index, rt.anim_frame), src/render.d/gi.c:88; thread id 0 */
  c->rng0 = 1 + index;
  c->rng1 = 2 + frame;
  if(o_pixels_from_index == 2)
  { /* MI_PIXELS_FROM_INDEX (corona_mi.h), NOT the reference: the two state words go through the splitmix64 finaliser first. Seeded as above, paths
       i and i + 1 draw correlated first numbers (their wavelengths: r = 0.96) -- harmless while the pixel is one of those numbers, a colour cast
       that takes thousands of samples to average out once neighbouring indices are neighbouring pixels (oracle_set_pixels_from_index(1) shows it) */
    uint64_t z = c->rng0 + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30))*0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27))*0x94d049bb133111ebull; c->rng0 = z ^ (z >> 31);
    z = c->rng1 + 2*0x9e3779b97f4a7c15ull + c->rng0;
    z = (z ^ (z >> 30))*0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27))*0x94d049bb133111ebull; c->rng1 = z ^ (z >> 31);
    if(!(c->rng0 | c->rng1)) c->rng0 = 1;
  }
  for(int k=0;k<10;k++) (void)o_rand(c);
}

float o_point(o_ctx *c, const o_path *p, int v, int dim)
{
  if(c->s->pointsampler != MI_POINTS_HALTON) return o_rand(c);
  const int d = p->v[v].rand_beg + dim;
  if(d >= 256) return o_rand(c);                                      /* "degenerate to pure random", halton.c:78-80 */
  return o_halton_sample((uint32_t)d, (uint32_t)p->index);            /* "this clips the bits in p->index to 32", halton.c:83 */
}

float oracle_rand_sequence(uint64_t index, uint64_t frame, int n, float *out)
{
  o_ctx c; memset(&c, 0, sizeof(c));
  o_rand_seed(&c, index, frame);
  float last = 0.0f;
  for(int k=0;k<n;k++) { last = o_rand(&c); if(out) out[k] = last; }
  return last;
}

/* ---------------------------------------------------------------- geometric terms */
static float o_path_lambert(const o_path *p, int v, const float *omega)
{ /* path_lambert, src/pathspace.c:45-56 */
  if(!(p->v[v].mode & s_sensor) && p->v[v].hit.prim == MI_PRIMID_INVALID) return 1.0f;
  return fabsf(dot3(p->v[v].hit.n, omega));
}

static float o_path_G(const o_path *p, int e)
{ /* path_G, src/pathspace.c:59-69 */
  if(p->v[e].flags & s_environment)   return o_path_lambert(p, e-1, p->e[e].omega);
  if(p->v[e-1].flags & s_environment) return o_path_lambert(p, e, p->e[e].omega);
  return o_path_lambert(p, e-1, p->e[e].omega)*o_path_lambert(p, e, p->e[e].omega)/(p->e[e].dist*p->e[e].dist);
}

/* ---------------------------------------------------------------- emitters */
static float o_lights_eval_vertex(const o_path *path, int v)
{ /* lights_eval_vertex, src/lights.d/list.c:242-275 (path tracing direction) */
  if(o_g_all(O_CTX(path), path->v[v].shading.em <= 0.0f)) return 0.0f;       /* mf_all(mf_lte(em, 0)), list.c:246 */
  float edf = 1.0f;
  const float *omega = path->e[v].omega;
  if(path->v[v].hit.prim != MI_PRIMID_INVALID)
  {
    if(dot3(path->v[v].hit.gn, omega) >= 0.0) return 0.0f;
    if(path->v[v].shading.roughness > 1.0f-1e-4f) edf = 1.0f/M_PI;
    else
    {
      const float phongexp = 2.0f/(path->v[v].shading.roughness*path->v[v].shading.roughness) - 2.0f;
      edf = powf(fabsf(dot3(path->v[v].hit.gn, omega)), phongexp)*(phongexp+2.0f)/(2.0f*M_PI);
    }
  }
  return edf*path->v[v].shading.em;
}

static void o_path_update_throughput(o_path *path, int v)
{ /* path_update_throughput, src/pathspace.c:148-164 */
  path->throughput = (path->v[v].throughput*path->e[v].contribution)/path->e[v].pdf;
  path->v[v].throughput = path->v[v].throughput*(path->e[v].transmittance/path->e[v].pdf);
  if((path->v[0].mode & s_sensor) && (path->v[v].mode & s_emit))
    path->throughput = path->throughput + path->v[v].throughput*o_lights_eval_vertex(path, v);
}

/* ---------------------------------------------------------------- camera */
static void o_quaternion_mult(float *in, const float *p)
{ /* quaternion_mult, include/quaternion.h:41-48; quaternions as w, x, y, z */
  const float r[4] = { in[0], in[1], in[2], in[3] };
  in[1] = r[0]*p[1] + r[1]*p[0] + r[2]*p[3] - r[3]*p[2];
  in[2] = r[0]*p[2] - r[1]*p[3] + r[2]*p[0] + r[3]*p[1];
  in[3] = r[0]*p[3] + r[1]*p[2] - r[2]*p[1] + r[3]*p[0];
  in[0] = r[0]*p[0] - r[1]*p[1] - r[2]*p[2] - r[3]*p[3];
}

static void o_quaternion_transform(const float *q, float *v)
{ /* quaternion_transform, include/quaternion.h:60-75: q v q' */
  const float vq[4] = { 0.0f, v[0], v[1], v[2] }, inv[4] = { q[0], -q[1], -q[2], -q[3] };
  float res[4] = { q[0], q[1], q[2], q[3] };
  o_quaternion_mult(res, vq);
  o_quaternion_mult(res, inv);
  for(int k=0;k<3;k++) v[k] = res[k+1];
}

static void o_view_cam_init_frame(const mi_camera *cam, float time, o_hit *hit)
{ /* view_cam_init_frame, src/view.c:903-919 + quaternion_slerp, include/quaternion.h:86-110 */
  const float *q = cam->orient, *p = cam->orient_t1;
  float r[4];
  const float cos_theta_2 = q[0]*p[0] + (q[1]*p[1] + q[2]*p[2] + q[3]*p[3]);
  if(fabsf(cos_theta_2) >= 1.0f) memcpy(r, q, sizeof(r));
  else
  {
    const float theta_2 = acosf(cos_theta_2);
    const float sin_theta_2 = sqrtf(1.0f - cos_theta_2*cos_theta_2);
    if(fabsf(sin_theta_2) < 1e-10f) for(int k=0;k<4;k++) r[k] = (q[k] + p[k])*.5f;
    else
    {
      const float a = sinf((1.0f - time)*theta_2)/sin_theta_2;
      const float b = sinf(time*theta_2)/sin_theta_2;
      for(int k=0;k<4;k++) r[k] = q[k]*a + p[k]*b;
    }
  }
  hit->a[0] = hit->b[1] = hit->n[2] = 1.0f;
  hit->a[1] = hit->a[2] = hit->b[0] = hit->b[2] = hit->n[0] = hit->n[1] = 0.0f;
  o_quaternion_transform(r, hit->a);
  o_quaternion_transform(r, hit->b);
  o_quaternion_transform(r, hit->n);
  for(int k=0;k<3;k++) hit->gn[k] = hit->n[k];
  for(int k=0;k<3;k++) hit->x[k] = cam->pos[k]*(1.0f-time) + cam->pos_t1[k]*time;
  o_normalise(hit->a);
  o_normalise(hit->b);
  o_normalise(hit->n);
}

static float o_camera_sample(o_ctx *c, o_path *p)
{ /* camera_sample + _camera_sample_internal, src/camera.d/thinlens.c:68-128; view_cam_init_frame,
     src/view.c:903-919 (static camera: frame precomputed by the host) */
  const mi_scene_desc *s = c->s;
  const mi_camera *cam = &s->cam;
  const float W = (float)s->width, H = (float)s->height;
  /* the pixel: two numbers of the point sampler -- or, when the path's pixel has been set (path_set_pixel, include/pathspace.h:355-360), that
     pixel and NO numbers for the two image dimensions (thinlens.c:117-118: `p->sensor.pixel_set ? p->sensor.pixel_i : pointsampler(...)`).
     The pixel of path `index` is what render_sample_path's tiled branch computes, src/render.d/gi.c:88-95 (oracle_set_pixels_from_index). */
  float i, j;
  if(o_pixels_from_index == 1 || o_pixels_from_index == 2)
  {
    const uint64_t w = s->width, h = s->height;                                     /* gi.c:89-92, literally */
    const uint64_t frame = p->index / (w*h);
    const uint64_t y = (p->index - frame * (w*h))/w;
    const uint64_t x = (p->index - frame * (w*h) - y * w);
    i = (float)x; j = (float)y;
    if(o_pixels_from_index == 2)
    { /* the product's mode: the position inside the pixel from the two numbers camera_sample would have turned into the pixel itself */
      i += o_point(c, p, 0, o_dim_image_x);
      j += o_point(c, p, 0, o_dim_image_y);
    }
  }
  else
  {
    i = o_point(c, p, 0, o_dim_image_x)*W;
    j = o_point(c, p, 0, o_dim_image_y)*H;
  }
  const float r1 = o_point(c, p, 0, o_dim_aperture_x);
  const float r2 = o_point(c, p, 0, o_dim_aperture_y);
  const float lens_radius = (.5f/cam->f_stop)*cam->focal_length;
  const float u = cosf(2*M_PI*r1)*sqrtf(r2)*lens_radius;
  const float v = sinf(2*M_PI*r1)*sqrtf(r2)*lens_radius;

  p->v[0].rand_cnt = 7;                                  /* s_dim_num_pt_beg, thinlens.c:100-103 */
  p->v[1].rand_beg = p->v[0].rand_beg + p->v[0].rand_cnt;
  p->v[1].rand_cnt = 1;
  o_hit *h = &p->v[0].hit;
  for(int k=0;k<3;k++) { h->a[k] = cam->a[k]; h->b[k] = cam->b[k]; h->n[k] = cam->n[k]; h->gn[k] = cam->n[k]; h->x[k] = cam->pos[k]; }
  if(cam->moving) o_view_cam_init_frame(cam, p->time, h);
  const float f = cam->focus/cam->focal_length;
  const float f_dir = cam->focus;
  const float f_rg = -cam->film_width*f/W;
  const float f_up = -cam->film_height*f/H;
  float aoff[3];
  for(int k=0;k<3;k++) aoff[k] = u*h->a[k] + v*h->b[k];
  for(int k=0;k<3;k++) p->e[1].omega[k] = f_dir*h->n[k] + ((i-.5f*W)*f_rg*h->a[k] + (j-.5f*H)*f_up*h->b[k]) - aoff[k];
  o_normalise(p->e[1].omega);

  const float A = M_PI*cam->focal_length*cam->focal_length/(4.0f*cam->f_stop*cam->f_stop);
  const float pdf_a = 1./A;
  const float sensor = 106.86535f*100.0f*cam->exposure_time;
  const float dot = dot3(p->e[1].omega, h->n);
  const float dot4 = dot*dot*dot*dot;
  h->prim = MI_PRIMID_INVALID;
  h->shader = -1;
  p->v[0].pdf = 1.0f;
  p->v[0].material_modes = p->v[0].mode = s_sensor;
  p->pixel_i = OCLAMP(i, 0.0, W-1e-4f);
  p->pixel_j = OCLAMP(j, 0.0, H-1e-4f);
  const float G = dot4/(cam->focal_length*cam->focal_length);
  const float pdf_v = 1.0f/(cam->film_width*cam->film_height);
  p->v[1].pdf = pdf_v*pdf_a/G;
  for(int k=0;k<3;k++) h->x[k] += aoff[k];
  return sensor*G/(pdf_a*pdf_v);
}

static float o_view_cam_pdf(const o_ctx *c, const o_path *p)
{ /* camera_pdf(c, p, 0) for the path tracing direction, src/camera.d/thinlens.c (pdf of v[1] in projected
     solid angle as set by _camera_sample_internal) -- only reachable from ptdl for 2-vertex emitter hits */
  const mi_camera *cam = &c->s->cam;
  const float A = M_PI*cam->focal_length*cam->focal_length/(4.0f*cam->f_stop*cam->f_stop);
  const float dot = dot3(p->e[1].omega, p->v[0].hit.n);
  const float G = dot*dot*dot*dot/(cam->focal_length*cam->focal_length);
  return (1.0f/(cam->film_width*cam->film_height))*(1.0f/A)/G;
}

/* ---------------------------------------------------------------- homogeneous media on an edge */
static float o_vol_transmittance(o_path *p, int e)
{ /* shader_vol_transmittance, src/shader.c:48-74 (homogeneous default case) */
  p->e[e].contribution = 0.0f;
  p->e[e].pdf = 1.0f;
  if(p->e[e].vol.shader >= 0)
  {
    if((p->v[e].flags & s_environment) || (p->v[e-1].flags & s_environment)) p->e[e].transmittance = expf(-10000.0f*p->e[e].vol.mu_t);
    else p->e[e].transmittance = expf(-p->e[e].dist*p->e[e].vol.mu_t);
    p->e[e].pdf = 1.0f;
    p->e[e].contribution = 0.0f;
    return p->e[e].transmittance;
  }
  p->e[e].transmittance = 1.0f;
  return p->e[e].transmittance;
}

static float o_vol_sample(o_ctx *c, o_path *p, int e)
{ /* shader_vol_sample, src/shader.c:76-106: free flight distance; pointsampler(p, s_dim_free_path) with p->length == e */
  float dist = FLT_MAX;
  p->e[e].contribution = 0.0f;
  p->e[e].pdf = 1.0f;
  p->e[e].transmittance = 1.0f;
  if(p->e[e].vol.shader >= 0)
  {
    if(o_g_hero(c, p->e[e].vol.mu_s) > 0.0f)                 /* mf(mu_s, 0) > 0: the hero's medium decides whether a distance is sampled ... */
    {
      const float rf = o_point(c, p, e, 0);                /* s_dim_free_path */
      dist = -logf(1.0f - rf)/o_g_hero(c, p->e[e].vol.mu_t);   /* ... and samples it (mf(mu_t, 0), src/shader.c:95); transmittance and pdf per component */
      if(!(dist > 0.0)) dist = 1e-15;
      p->e[e].pdf = p->e[e].transmittance = expf(-dist*p->e[e].vol.mu_t);
      if(dist < p->e[e].dist) p->e[e].pdf = p->e[e].pdf*p->e[e].vol.mu_t;
    }
    else p->e[e].transmittance = expf(-p->e[e].dist*p->e[e].vol.mu_t);
  }
  return dist;
}

static float o_vol_pdf(const o_path *p, int e)
{ /* shader_vol_pdf, src/shader.c:108-131 */
  float pdf = 1.0f;
  if(p->e[e].vol.shader >= 0 && o_g_hero(O_CTX(p), p->e[e].vol.mu_s) > 0.0f)      /* mf(mu_s, 0) > 0, src/shader.c:122 */
  {
    pdf = expf(-p->e[e].dist*p->e[e].vol.mu_t);
    if(!(p->v[e].flags & s_environment) && p->v[e].hit.prim == MI_PRIMID_INVALID) pdf = pdf*p->e[e].vol.mu_t;
  }
  return pdf;
}

/* ---------------------------------------------------------------- propagate / extend */
static int o_path_propagate(o_ctx *c, o_path *path, int v)
{ /* path_propagate(path, v, s_propagate_sample), src/pathspace.c:697-895; homogeneous media only */
  const mi_scene_desc *s = c->s;
  if(o_path_edge_init_volume(path, v)) return 1;
  path->v[v].mode = s_absorb;
  path->v[v].flags = s_none;
  /* homogeneous + scattering: sample the clip distance first, then trace up to it (src/pathspace.c:717-751) */
  const int sample_vol_first = o_g_hero(c, path->e[v].vol.mu_s) > 0.0;          /* mf(mu_s, 0) > 0, src/pathspace.c:720 */
  int vshader = -1;
  float clipdist = FLT_MAX;
  if(sample_vol_first)
  {
    path->e[v].dist = FLT_MAX;
    clipdist = o_vol_sample(c, path, v);
  }
  if(clipdist < FLT_MAX) vshader = path->e[v].vol.shader;
  o_hit *hit = &path->v[v].hit;
  o_ray ray;
  hit->prim = MI_PRIMID_INVALID;
  hit->dist = clipdist;
  hit->shader = vshader;
  for(int k=0;k<3;k++) ray.pos[k] = path->v[v-1].hit.x[k];
  for(int k=0;k<3;k++) ray.dir[k] = path->e[v].omega[k];
  ray.time = path->time;
  ray.ignore = MI_PRIMID_INVALID;
  ray.min_dist = 0.0;
  if(path->v[v-1].hit.prim != MI_PRIMID_INVALID) o_prims_offset_ray(&path->v[v-1].hit, &ray);
  o_accel_intersect(c, &ray, hit);
  path->e[v].dist = hit->dist;
  if(hit->dist < FLT_MAX)
  {
    for(int k=0;k<3;k++) hit->x[k] = ray.pos[k] + hit->dist*ray.dir[k];
    o_shader_prepare(c, path, v);
  }
  /* self-intersection, src/pathspace.c:807-820 */
  if((MI_PRIMID_VCNT(hit->prim) > 2 || path->e[v].dist < 1e-4f) &&
      hit->prim != MI_PRIMID_INVALID && hit->prim == path->v[v-1].hit.prim)
    return 5;
  if(!sample_vol_first)
  { /* src/pathspace.c:822-838; a homogeneous medium without scattering only attenuates, the sampled distance is FLT_MAX */
    (void)o_vol_sample(c, path, v);
  }
  else if(path->e[v].vol.shader >= 0 && !(path->v[v].material_modes & s_volume))
    path->e[v].pdf = path->e[v].transmittance = o_vol_transmittance(path, v);   /* geometry before the sampled distance */
  if(path->e[v].dist >= FLT_MAX)
  { /* environment, src/pathspace.c:856-873 */
    path->v[v].flags |= s_environment;
    o_shader_prepare(c, path, v);
    const float *aabb = s->aabb;
    const float far = 2.0f*OMAX(aabb[5] - aabb[2], OMAX(aabb[4] - aabb[1], aabb[3] - aabb[0]));
    for(int k=0;k<3;k++) hit->x[k] = path->v[v-1].hit.x[k] + far*path->e[v].omega[k];
  }
  /* (mf_any on either, src/pathspace.c:876-877; both hooks are always called: the four lanes of a hero path meet at the same hooks) */
  const int any_contribution = o_g_any(c, path->e[v].contribution > 0.0f), any_em = o_g_any(c, path->v[v].shading.em > 0.0f);
  if(any_contribution || (any_em && !(path->v[v].flags & s_inside)))
    path->v[v].material_modes = path->v[v].mode = s_emit;
  path->v[v].pdf = path->v[v].pdf*path->e[v].pdf;
  return 0;
}

static int o_path_extend(o_ctx *c, o_path *path)
{ /* path_extend, src/pathspace.c:167-271 */
  const mi_scene_desc *s = c->s;
  if(path->length >= (int)s->max_verts) return 1;
  int v = path->length;
  if(v)
  {
    memset(path->v + v, 0, sizeof(o_vertex));
    memset(path->e + v, 0, sizeof(o_edge));
  }
  if(path->length)
  {
    if(path->v[v-1].flags & s_environment) return 1;
    if(!o_g_any(c, path->v[v-1].throughput > 0.0f))          /* !mf_any(mf_gt(throughput, 0)), src/pathspace.c:189 */
    {
      path->v[v-1].throughput = 0.0f;
      path->v[v-1].mode = s_absorb;
      return 1;
    }
    path->v[v].rand_beg = path->v[v-1].rand_beg + path->v[v-1].rand_cnt;
    path->v[v].pdf = 1.0f;
    path->v[v].throughput = path->v[v-1].throughput;
    path->v[v].throughput = path->v[v].throughput*o_shader_sample(c, path);
    path->v[v].rand_cnt = 5;                               /* s_dim_num_extend */
  }
  else
  {
    /* draw order: scramble, lambda, time, camid (path_extend), camid again (view_cam_sample,
       src/view.c:846-847), then image x/y, aperture x/y (thinlens.c:117-121) */
    path->scramble = 0.1f + o_rand(c)*(0.9f-0.1f);
    float lf = fmodf(o_point(c, path, 0, o_dim_lambda) + 0/(float)1, 1.0f);
    if(c->grp)
    { /* MF_COUNT = 4, src/pathspace.c:218-221: the point sampler is asked once PER COMPONENT -- with the `rand` sampler four numbers --, component l
         takes fmodf(number l + l / 4, 1). Every lane draws all four (the lanes' generators stay in step) and keeps its own. */
      float lfs[O_MF_MAX];
      lfs[0] = lf;
      for(int l=1;l<O_MF(c);l++) lfs[l] = fmodf(o_point(c, path, 0, o_dim_lambda) + l/(float)O_MF(c), 1.0f);
      lf = lfs[c->lane];
    }
    path->lambda = 360 + (830 - 360)*lf;                          /* spectrum_sample_lambda, include/spectrum.h:206-210 */
    path->time = o_point(c, path, 0, o_dim_time)*s->cam.time_scale; /* view_sample_time, src/view.c:881-891 */
    (void)o_point(c, path, 0, o_dim_camid);                        /* view_sample_camid: one camera */
    (void)o_point(c, path, 0, o_dim_camid);
    path->v[0].throughput = 1.0f*o_camera_sample(c, path);         /* num_cams * camera_sample */
    path->v[0].throughput = path->v[0].throughput/1.0f;            /* view_pdf_camid */
    { /* shader_exterior_medium, src/shader.c:544-565: vacuum, or the global medium's prepare chain with mode, material modes
         and shading of the sensor vertex put back afterwards */
      memset(&path->v[0].interior, 0, sizeof(o_volume));
      path->v[0].interior.ior = 1.0f; path->v[0].interior.shader = -1;
      if(s->exterior)
      {
        const uint32_t mode = path->v[0].mode, material_modes = path->v[0].material_modes;
        const o_shading shading = path->v[0].shading;
        o_prepare_medium(s, path, 0, (int)s->exterior - 1);
        path->v[0].mode = mode; path->v[0].material_modes = material_modes; path->v[0].shading = shading;
      }
      path->e[0].vol = path->v[0].interior;
    }
    path->length++;
    path->v[1].throughput = path->v[0].throughput;
    path->v[0].tech = s_tech_extend;
    v++;
  }
  if(o_g_all(c, path->v[v].throughput <= 0.0f) || o_path_propagate(c, path, v))       /* mf_all(mf_lte(throughput, 0)), src/pathspace.c:253 */
  {
    if(!(path->v[v-1].mode & s_emit)) path->v[v-1].mode = s_absorb;
    path->v[v].throughput = -0.0f;
    return 1;
  }
  path->v[v].pdf = path->v[v].pdf*o_path_G(path, v);
  path->length++;
  o_path_update_throughput(path, v);
  path->v[v].total_throughput = path->throughput;
  path->v[v].tech = s_tech_extend;
  return 0;
}

static int o_path_russian_roulette(o_ctx *c, o_path *path, float p_survival)
{ /* path_russian_roulette, src/pathspace.c:273-292 */
  const int v = path->length-1;
  const float rr = o_point(c, path, v, o_dim_russian_r);
  if(rr >= p_survival)
  {
    path->v[v].throughput = path->v[v].throughput*(1.0f/(1.0f-p_survival));
    path->v[v].pdf = path->v[v].pdf*(1.0f-p_survival);
    return 1;
  }
  path->v[v].throughput = path->v[v].throughput*(1.0f/p_survival);
  path->v[v].pdf = path->v[v].pdf*p_survival;
  return 0;
}

static float o_path_throughput(const o_path *path)
{ /* path_throughput, src/pathspace.c:374-381 */
  if(path->length < 2) return 0.0f;
  return path->throughput;
}

/* ---------------------------------------------------------------- splat */
static float o_bh_w(float n)
{ /* filter_bh_w, include/filter/blackmanharris.h:28-41 */
  const float NN = 4.0f;
  if(n > NN-1.0f || n < 0.0f) return 0.0f;
  const float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
  const float N_1 = 1.0f/(NN-1.0f);
  const float cos1 = cosf(2.0f*M_PI*n*N_1);
  const float cos2 = cosf(4.0f*M_PI*n*N_1);
  const float cos3 = cosf(6.0f*M_PI*n*N_1);
  return a0 - a1*cos1 + a2*cos2 - a3*cos3;
}

static void o_fb_add(o_ctx *c, float *p, float inc)
{ /* common_atomic_add, include/corona_common.h:316-329 */
  if(!c->atomic_fb) { *p += inc; return; }
  uint32_t *ip = (uint32_t *)p;
  uint32_t oldi, newi;
  do
  {
    float f = *(volatile float *)p;
    memcpy(&oldi, &f, 4);
    f += inc;
    memcpy(&newi, &f, 4);
  }
  while(!__sync_bool_compare_and_swap(ip, oldi, newi));
}

static void o_splat(o_ctx *c, const o_path *path, float value)
{ /* view_splat, src/view.c:455-463 -> spectrum_p_to_camera, include/spectrum.h:172-203 (camera space = XYZ)
     -> filter_blackmanharris_splat, include/filter/blackmanharris.h:43-77 */
  const mi_scene_desc *s = c->s;
  float col[3] = {0.0f, 0.0f, 0.0f};
  /* hero wavelengths: mf_any(value > 0), mf_all(value < FLT_MAX), mf_all(value == value) -- all three hooks always (the lanes meet) */
  const int any_pos = o_g_any(c, value > 0.0f), all_fin = o_g_all(c, value < FLT_MAX), all_num = o_g_all(c, value == value);
  const int ok = any_pos && all_fin && all_num;
  if(ok)
  {
    float f = (path->lambda - 360)/5;
    const int i = (int)f;
    f -= i;
    for(int k=0;k<3;k++) col[k] = ((1-f)*s->cie_xyz[3*i+k] + f*s->cie_xyz[3*(i+1)+k])*value;
  }
  if(c->grp)
  { /* spectrum_p_to_xyz, include/spectrum.h:185-195: xyz[k] += b[k] * p[l] for l = 0 .. 3 in that order; the hero lane splats the sum */
    for(int k=0;k<3;k++) c->grp->col[c->lane][k] = col[k];
    o_g_meet(c);
    for(int k=0;k<3;k++) { col[k] = 0.0f; for(int l=0;l<O_MF(c);l++) col[k] += c->grp->col[l][k]; }
    o_g_meet(c);
    if(c->hero_ext && c->hero_splats < MI_REC_MAX_SPLATS) *O_EXT_SPLAT(c, c->hero_splats) = value;
    c->hero_splats++;
    if(c->lane) return;
  }
  if(c->rec && c->rec->num_splats < MI_REC_MAX_SPLATS)
  {
    mi_path_splat *sp = c->rec->splat + c->rec->num_splats++;
    sp->length = path->length;
    sp->tech = path->length ? path->v[path->length-1].tech : -1;
    sp->value = value;
    memcpy(sp->col, col, sizeof(col));
  }
  if(!ok) return;
  c->cnt[5]++;
  if(!c->fb) return;
  const int wd = (int)s->width, ht = (int)s->height;
  const float pi = path->pixel_i, pj = path->pixel_j;
  const int x0 = (int)(pi - 1.5f), y0 = (int)(pj - 1.5f);
  const int u0 = -x0 < 0 ? 0 : -x0, v0 = -y0 < 0 ? 0 : -y0;
  const int u4 = x0 + 4 > wd ? wd - x0 : 4, v4 = y0 + 4 > ht ? ht - y0 : 4;
  float weight = 0.0f;
  for(int v=v0;v<v4;v++) for(int u=u0;u<u4;u++)
  {
    const float uu = (x0 + u + .5f) - pi, vv = (y0 + v + .5f) - pj;
    weight += o_bh_w(sqrtf(uu*uu + vv*vv) + 1.5f);
  }
  if(weight <= 0) return;
  weight = 1.0f/weight;
  for(int v=v0;v<v4;v++) for(int u=u0;u<u4;u++)
  {
    const float uu = (x0 + u + .5f) - pi, vv = (y0 + v + .5f) - pj;
    const float f = weight*o_bh_w(sqrtf(uu*uu + vv*vv) + 1.5f);
    float *px = c->fb + 3*((size_t)(x0+u) + (size_t)wd*(y0+v));
    for(int k=0;k<3;k++) o_fb_add(c, px+k, col[k]*f);
  }
}

/* ---------------------------------------------------------------- pt */
static void o_sampler_pt(o_ctx *c, o_path *path)
{ /* sampler_create_path, src/sampler.d/pt.c:40-54 */
  while(1)
  {
    if(o_path_extend(c, path)) return;
    if(path->v[path->length-1].mode & s_emit)
    {
      /* sampler_mis, pt.c:30-38: hero-wavelength weight, == 1 unless the pdf product leaves float range */
      double pdf = 1.0;
      for(int v=1;v<path->length;v++) pdf = pdf*(double)path->v[v].pdf;
      const float w = (float)pdf/o_g_hsum(c, (float)pdf);            /* mf_div(md_2f(pdf), mf_set1(mf_hsum(md_2f(pdf)))) */
      o_splat(c, path, w*o_path_throughput(path));
      if(path->length > 3)
      { /* pt.c:50: the hero's throughputs decide for all four (mf(throughput, 0)) */
        const float t1 = o_g_hero(c, path->v[path->length-1].throughput), t2 = o_g_hero(c, path->v[path->length-2].throughput);
        if(o_path_russian_roulette(c, path, OMIN(1.0f, t1/t2)))
          return;
      }
    }
  }
}

/* ---------------------------------------------------------------- next event estimation (ptdl) */
static float o_lights_pdf_next_event(const o_ctx *c, const o_path *p, int v)
{ /* lights_pdf_next_event, src/lights.d/list.c:106-128 */
  const mi_lights *l = &c->s->lights;
  if(p->v[v].hit.prim == MI_PRIMID_INVALID) return 0.0f;
  const uint32_t sid = MI_PRIMID_SHAPE(p->v[v].hit.prim);
  unsigned int min = 0, max = l->num_prims;
  unsigned int t = max/2;
  while(t != min)
  {
    if(MI_PRIMID_SHAPE(l->primid[t-1]) < sid) min = t;
    else max = t;
    t = (min + max)/2;
  }
  if(MI_PRIMID_SHAPE(l->primid[t]) != sid) return 0.0f;
  return l->L[t];
}

static int o_nee_possible(const o_path *p, int v)
{ /* nee_possible, include/pathspace/nee.h:8-19 */
  return (p->v[v].material_modes & (s_diffuse | s_glossy)) ? 1 : 0;
}

static float o_nee_pdf(const o_ctx *c, const o_path *p, int v)
{ /* nee_pdf = nee_pdf_nee (+ 0 without FNEE), include/pathspace/nee.h:21-47,76-79, for v == length-1 */
  if(p->length < 3) return 0.0f;
  const int v1 = v ? v-1 : v+1;
  if(!o_nee_possible(p, v1)) return 0.0f;
  const mi_lights *l = &c->s->lights;
  if(p->v[v].flags & s_environment) return 0.0f;        /* p_sky == 0 for the black sky */
  if(p->v[v].hit.prim != MI_PRIMID_INVALID && (p->v[v].mode & s_emit) && l->p_geo > 0)
    return l->p_geo*o_lights_pdf_next_event(c, p, v);
  return 0.0f;
}

static float o_path_pdf_extend(o_ctx *c, o_path *path, int v)
{ /* path_pdf_extend, src/pathspace.c:384-400 */
  float pdf;
  if(v == 0) return 1.0f;
  else if(v == 1) pdf = o_view_cam_pdf(c, path);
  else pdf = o_shader_pdf(c, path, v-1);
  return (o_vol_pdf(path, v)*pdf)*o_path_G(path, v);
}

static int o_path_visible(o_ctx *c, o_path *p, int v)
{ /* path_visible, src/pathspace.c:311-344: closest-hit loop up to the light's primitive */
  o_ray ray;
  ray.time = p->time;
  float total_dist = o_prims_get_ray(&p->v[v-1].hit, &p->v[v].hit, &ray);
  if(dot3(p->v[v].hit.gn, ray.dir) >= 0) return 0;
  o_hit hit = p->v[v-1].hit;
  const o_vertex lightv = p->v[v];
  while(total_dist > 0.0f)
  {
    hit.dist = total_dist;
    hit.prim = MI_PRIMID_INVALID;
    o_accel_intersect(c, &ray, &hit);
    if(hit.dist >= total_dist) break;
    if(hit.prim == MI_PRIMID_INVALID) break;
    if(hit.prim == lightv.hit.prim) break;
    total_dist -= hit.dist;
    for(int k=0;k<3;k++) ray.pos[k] = hit.x[k] = ray.pos[k] + hit.dist*ray.dir[k];
    p->v[v].hit = hit;
    const float prep = o_shader_prepare(c, p, v);
    if(prep >= 0.0) return 0;
    o_prims_offset_ray(&hit, &ray);
  }
  p->v[v] = lightv;
  return 1;
}

static uint32_t o_sample_cdf(const float *cdf, int num, float rand)
{ /* sample_cdf, include/sampler_common.h:206-226 */
  unsigned int min = 0, max = num;
  unsigned int t = max/2;
  while(t != min)
  {
    if(cdf[t] <= rand) min = t;
    else max = t;
    t = (min + max)/2;
  }
  if(max < (unsigned)num && cdf[t] <= rand) t = max;
  return t;
}

static float o_lights_sample_next_event(o_ctx *c, o_path *p)
{ /* lights_sample_next_event + _lights_sample_next_event, src/lights.d/list.c:130-174.
     arguments are drawn right to left by the reference build: nee_y, nee_x, then the light selector */
  const mi_lights *l = &c->s->lights;
  const int v = p->length;
  const float r3 = o_point(c, p, v, o_dim_nee_y);
  const float r2 = o_point(c, p, v, o_dim_nee_x);
  const float r1 = o_point(c, p, v, o_dim_nee_light2);
  const unsigned int t = o_sample_cdf(l->cdf, l->num_prims, r1);
  p->v[v].hit.prim = l->primid[t];
  o_prims_sample(c->s, l->primid[t], r2, r3, &p->v[v].hit, p->time);
  for(int k=0;k<3;k++) p->e[v].omega[k] = p->v[v].hit.x[k] - p->v[v-1].hit.x[k];
  p->e[v].dist = sqrtf(dot3(p->e[v].omega, p->e[v].omega));
  for(int k=0;k<3;k++) p->e[v].omega[k] *= 1./p->e[v].dist;
  o_shader_prepare(c, p, v);
  p->v[v].pdf = l->L[t];
  if(p->v[v].shading.roughness > 1.0f-1e-4f) p->v[v].material_modes = p->v[v].mode = s_emit | s_diffuse;
  else p->v[v].material_modes = p->v[v].mode = s_emit | s_glossy;
  float edf = p->v[v].shading.em/p->v[v].pdf;
  if(p->v[v].shading.roughness > 1.0f-1e-4f) edf = edf*(1.0f/M_PI);
  else
  {
    const float phongexp = 2.0f/(p->v[v].shading.roughness*p->v[v].shading.roughness) - 2.0f;
    edf = edf*(powf(-dot3(p->v[v].hit.gn, p->e[v].omega), phongexp)*(phongexp + 2.0f)/(2.0f*M_PI));
  }
  return edf;
}

static int o_nee_sample(o_ctx *c, o_path *p)
{ /* nee_sample, include/pathspace/nee.h:87-243 (no FNEE, black sky, no volume lights) */
  const mi_scene_desc *s = c->s;
  if(p->v[p->length-1].flags & s_environment) return 1;
  if(p->length >= (int)s->max_verts) return 1;
  const int v = p->length;
  float edf = 0.0f, bsdf = 0.0f, transmittance = 1.0f;
  int failed = 1;
  if(o_nee_possible(p, v-1))
  {
    const float p_sky = s->lights.p_sky, p_geo = s->lights.p_geo;
    memset(p->v + v, 0, sizeof(o_vertex));
    memset(p->e + v, 0, sizeof(o_edge));
    p->v[v].rand_beg = p->v[v-1].rand_beg + p->v[v-1].rand_cnt;
    p->v[v].tech = s_tech_nee;
    const float rand = o_point(c, p, v, o_dim_nee_light1);
    if(rand < p_sky) { /* envmap sampling: out of scope (black sky has p_sky == 0) */ }
    else if(rand < p_sky + p_geo)
    {
      edf = o_lights_sample_next_event(c, p);
      p->v[v].pdf = p->v[v].pdf*p_geo;
      edf = edf/p_geo;
    }
    if(o_g_any(c, edf > 0.0f))                                          /* !mf_any(mf_gt(edf, 0)) -> fail, nee.h:188 */
    {
      bsdf = o_shader_brdf(c, p, v-1);
      if(o_g_any(c, bsdf > 0.0f) && !o_path_edge_init_volume(p, v) && o_path_visible(c, p, v))       /* nee.h:191 */
      {
        o_shader_prepare(c, p, v);
        const float G = o_path_G(p, v);
        transmittance = o_vol_transmittance(p, v);
        p->v[v].throughput = ((p->v[v-1].throughput*bsdf)*(transmittance*edf))*G;
        p->v[v].throughput = p->v[v].throughput + (p->v[v-1].throughput*bsdf)*((p->e[v].contribution*G)/p->v[v].pdf);
        p->throughput = p->v[v].throughput;
        failed = 0;
      }
    }
  }
  if(failed)
  {
    p->v[v].pdf = 0.0f;
    p->throughput = 0.0f;
    p->v[v].throughput = 0.0f;
    p->v[v].flags = s_none;
    p->v[v].mode = s_absorb;
    p->v[v].rand_cnt = 4;                                  /* s_dim_num_nee */
    p->length++;
    return 0;
  }
  p->v[v].rand_cnt = 4;
  p->length++;
  const float pdf_nee = p->v[v].pdf, pdf_fnee = 0.0f;
  const float weight = pdf_nee/(pdf_nee + pdf_fnee/transmittance);
  p->throughput = p->throughput*weight;
  p->v[v].throughput = p->v[v].throughput*weight;
  p->v[v].total_throughput = p->v[v].throughput;
  p->v[v].pdf = pdf_nee + pdf_fnee;
  return 0;
}

static void o_path_pop(o_path *path)
{ /* path_pop, src/pathspace.c:294-308 */
  const int v = path->length-1;
  path->v[v-1].rand_cnt += path->v[v].rand_cnt;
  path->v[v-1].mode &= s_emit;
  path->length--;
  path->throughput = path->v[v-1].total_throughput;
}

static float o_ptdl_mis(const o_path *p, float pdf, float pdf2)
{ /* sampler_mis, src/sampler.d/ptdl.c:78-88: the combined balance heuristic over wavelengths and techniques (hero: the sum runs over the four lanes) */
  double pdf_path = 1.0;
  for(int v=1;v<p->length-1;v++) pdf_path = pdf_path*(double)p->v[v].pdf;
  const double our = (double)pdf*pdf_path;
  const double other = (double)pdf2*pdf_path;
  return (float)our/o_g_hsum(O_CTX(p), (float)(other + our));
}

static void o_sampler_ptdl(o_ctx *c, o_path *path)
{ /* sampler_create_path, src/sampler.d/ptdl.c:112-150 (nee_probability == 1) */
  const mi_scene_desc *s = c->s;
  while(1)
  {
    if(o_path_extend(c, path)) return;
    const int v = path->length-1;
    if(path->v[v].mode & s_emit)
    {
      const float weight = o_ptdl_mis(path, path->v[v].pdf, 1.0f*o_nee_pdf(c, path, v));
      o_splat(c, path, o_path_throughput(path)*weight);
    }
    if(path->length >= (int)s->max_verts) return;
    const float rr = 1.0f;
    if(o_rand(c) < rr)
    {
      if(o_nee_sample(c, path)) return;
      const int v2 = path->length-1;
      const float throughput = o_path_throughput(path)/rr;
      if(o_g_any(c, throughput > 0.0f) && (path->v[v2].mode & s_emit))            /* mf_any, ptdl.c:142 */
      {
        const float weight = o_ptdl_mis(path, rr*path->v[v2].pdf, o_path_pdf_extend(c, path, v2));
        o_splat(c, path, throughput*weight);
      }
      o_path_pop(path);
    }
  }
}

/* ---------------------------------------------------------------- entry points */
static void o_fill_record(const o_path *p, mi_path_record *r)
{
  r->pixel_i = p->pixel_i; r->pixel_j = p->pixel_j; r->lambda = p->lambda; r->time = p->time;
  r->scramble = p->scramble; r->throughput = p->throughput; r->length = p->length;
  for(int v=0;v<p->length && v<MI_REC_MAX_VERTS;v++)
  {
    mi_path_vertex *d = r->v + v;
    const o_vertex *sv = p->v + v;
    d->prim = sv->hit.prim;
    d->dist = p->e[v].dist;
    for(int k=0;k<3;k++) { d->x[k] = sv->hit.x[k]; d->n[k] = sv->hit.n[k]; d->gn[k] = sv->hit.gn[k]; d->omega[k] = p->e[v].omega[k]; }
    d->mode = sv->mode; d->flags = sv->flags;
    d->throughput = sv->throughput; d->pdf = sv->pdf;
    d->u = sv->hit.u; d->v = sv->hit.v;
    d->rd = sv->shading.rd; d->rg = sv->shading.rg; d->em = sv->shading.em; d->roughness = sv->shading.roughness;
    d->eta = sv->eta;
    d->shader = sv->hit.shader;
  }
}

static void o_hero_fill_ext(const o_path *p, o_ctx *c);
static void o_trace(o_ctx *c, uint64_t index)
{ /* render_sample_path, src/render.d/gi.c:81-105 -> pointsampler_mutate -> path_init + sampler_create_path */
  o_path path;
  path.ctx = c;
  path.lambda = 0.0f; path.throughput = 0.0f; path.length = 0; path.time = 0; path.index = index;
  path.scramble = 0.0f; path.pixel_i = path.pixel_j = 0.0f;
  memset(path.v, 0, 2*sizeof(o_vertex));
  memset(path.e, 0, 2*sizeof(o_edge));
  o_rand_seed(c, index, c->s->frame);
  if(c->s->sampler == MI_SAMPLER_PTDL) o_sampler_ptdl(c, &path);
  else o_sampler_pt(c, &path);
  if(c->hero_ext) o_hero_fill_ext(&path, c);
  c->cnt[4]++;
  c->cnt[6] += path.length;
  if(c->rec) o_fill_record(&path, c->rec);
}

static void o_prepare_points(const mi_scene_desc *s, uint64_t end)
{ /* pointsampler_init(frame) + pointsampler_prepare_frame, src/pointsampler.d/halton.c:46-52,122-129: the permutations are
     drawn again with the next seed once the end of the progression passes a multiple of 2^32 path indices */
  if(s->pointsampler == MI_POINTS_HALTON) o_halton_prepare(s->frame + (end >> 32));
}

static void o_trace_path(const mi_scene_desc *s, uint64_t index, float *fb, mi_path_record *rec, uint64_t *counters);
void oracle_trace_path(const mi_scene_desc *s, uint64_t index, float *fb, mi_path_record *rec, uint64_t *counters)
{
  o_prepare_points(s, index + 1);
  o_trace_path(s, index, fb, rec, counters);
}

static void o_trace_path(const mi_scene_desc *s, uint64_t index, float *fb, mi_path_record *rec, uint64_t *counters)
{
  o_ctx c;
  memset(&c, 0, sizeof(c));
  c.s = s; c.fb = fb; c.rec = rec;
  if(rec) { memset(rec, 0, sizeof(*rec)); rec->index = index; }
  o_trace(&c, index);
  if(counters) for(int k=0;k<8;k++) counters[k] += c.cnt[k];
}

void oracle_trace_records(const mi_scene_desc *s, uint64_t first, uint64_t count, mi_path_record *out)
{
  o_prepare_points(s, first + count);
  for(uint64_t i=0;i<count;i++) o_trace_path(s, first + i, 0, out + i, 0);
}

/* ---------------------------------------------------------------- hero wavelengths: four lanes in lock step (o_core.h) */
typedef struct o_hero_lane { o_ctx c; uint64_t first, count; mi_path_record *out; float *ext; } o_hero_lane;

static void o_hero_fill_ext(const o_path *p, o_ctx *c)
{
  *O_EXT_LAMBDA(c) = p->lambda;
  for(int v=0;v<p->length && v<MI_REC_MAX_VERTS;v++)
  {
    *O_EXT_VERTEX(c, 0, v) = p->v[v].throughput; *O_EXT_VERTEX(c, 1, v) = p->v[v].pdf;
    *O_EXT_VERTEX(c, 2, v) = p->v[v].shading.rd; *O_EXT_VERTEX(c, 3, v) = p->v[v].shading.rg; *O_EXT_VERTEX(c, 4, v) = p->v[v].shading.em; *O_EXT_VERTEX(c, 5, v) = p->v[v].eta;
  }
}

/* test switch: the hero lanes flush denormal floats to zero like the reference BUILD does (-ffast-math links crtfastmath: FTZ | DAZ in MXCSR). It
 * matters where a component's pdf or throughput underflows -- deep paths through a medium: the reference's MIS weight of such a connection is
 * 0 / 0 = NaN in all four components (view_splat drops it), with denormals kept it is a finite weight on a contribution of 1e-14. Off by default
 * (the device keeps denormals); tests/test_oracle_hero.py turns it on for the comparison with the reference's dumps. */
static int o_reference_ftz = 0;
#if defined(__x86_64__)
#include <xmmintrin.h>
int oracle_set_reference_ftz(int on) { o_reference_ftz = on; return 1; }
#else
int oracle_set_reference_ftz(int on) { (void)on; return 0; }
#endif

static void *o_hero_worker(void *arg)
{
  o_hero_lane *L = (o_hero_lane *)arg;
#if defined(__x86_64__)
  if(o_reference_ftz) _mm_setcsr(_mm_getcsr() | 0x8040u);        /* this thread only: it ends with the call */
#endif
  for(uint64_t i=0;i<L->count;i++)
  {
    L->c.rec = L->c.lane == 0 && L->out ? L->out + i : 0;
    if(L->c.rec) { memset(L->c.rec, 0, sizeof(*L->c.rec)); L->c.rec->index = L->first + i; }
    L->c.hero_ext = L->ext ? L->ext + i*O_EXT_FLOATS(L->c.grp->n) : 0;
    L->c.hero_splats = 0;
    o_trace(&L->c, L->first + i);
  }
  return 0;
}

/* paths [first, first + count) with MF_COUNT = 4 wavelengths each. out: the usual records with the HERO component of every spectral quantity
   (what refharness/render_dump.c writes from a -DMF_COUNT=4 build); ext: all four components (may be NULL); fb: framebuffer or NULL */
void oracle_hero_trace(const mi_scene_desc *s, uint64_t first, uint64_t count, mi_path_record *out, oracle_hero_ext *ext, float *fb, uint64_t *counters)
{
  oracle_hero_trace_n(s, 4, first, count, out, (float *)ext, fb, counters);
}
/* ... with n = 4 (MF_COUNT = 4, SSE) or 8 (MF_COUNT = 8, AVX: include/mf.h:22-279) wavelengths; ext: O_EXT_FLOATS(n) floats per path, the n-column layout of oracle_hero_ext */
void oracle_hero_trace_n(const mi_scene_desc *s, int n, uint64_t first, uint64_t count, mi_path_record *out, float *ext, float *fb, uint64_t *counters)
{
  if(n != 4 && n != 8) return;
  o_prepare_points(s, first + count);
  o_group grp;
  memset(&grp, 0, sizeof(grp));
  grp.n = n;
  o_hero_lane lane[O_MF_MAX];
  pthread_t th[O_MF_MAX];
  if(ext) memset(ext, 0, sizeof(float)*O_EXT_FLOATS(n)*count);
  for(int l=0;l<n;l++)
  {
    memset(lane + l, 0, sizeof(o_hero_lane));
    lane[l].c.s = s; lane[l].c.grp = &grp; lane[l].c.lane = l; lane[l].c.fb = l == 0 ? fb : 0;
    lane[l].first = first; lane[l].count = count; lane[l].out = out; lane[l].ext = ext;
    pthread_create(th + l, 0, o_hero_worker, lane + l);
  }
  for(int l=0;l<n;l++) pthread_join(th[l], 0);
  if(counters) for(int i=0;i<8;i++) counters[i] += lane[0].c.cnt[i];
}

typedef struct o_job
{
  const mi_scene_desc *s;
  float *fb;
  uint64_t *counter, end;
  uint64_t cnt[8];
  int atomic_fb;
} o_job;

static void *o_worker(void *arg)
{ /* work_sample, src/view.c:618-628 */
  o_job *j = (o_job *)arg;
  o_ctx c;
  memset(&c, 0, sizeof(c));
  c.s = j->s; c.fb = j->fb; c.atomic_fb = j->atomic_fb;
  while(1)
  {
    const uint64_t i = __sync_fetch_and_add(j->counter, 1);
    if(i >= j->end) break;
    o_trace(&c, i);
  }
  memcpy(j->cnt, c.cnt, sizeof(c.cnt));
  return 0;
}

/* The paths of frames [first_frame, first_frame + frames) whose pixel lies in a 32 x 32 tile t = member (mod members), tiles counted row by
 * row (tile size: include/render_tiles.h:156; the walk over a tile's pixels: src/render_tiles.c:71-81, here row by row -- the order does not
 * matter to the sum). Sets the pixels-from-index mode. Single-threaded per call unless threads > 1 (then one tile row of work at a time). */
typedef struct o_tilejob { const mi_scene_desc *s; float *fb; uint64_t *counter, items, first_frame; uint32_t member, members, local, tiles_x; uint64_t cnt[8]; int atomic_fb; } o_tilejob;
static void *o_tile_worker(void *arg)
{
  o_tilejob *j = (o_tilejob *)arg;
  o_ctx c;
  memset(&c, 0, sizeof(c));
  c.s = j->s; c.fb = j->fb; c.atomic_fb = j->atomic_fb;
  const uint64_t W = j->s->width, H = j->s->height;
  while(1)
  {
    const uint64_t it = __sync_fetch_and_add(j->counter, 1);
    if(it >= j->items) break;
    const uint64_t p = it & 1023u, r = it >> 10, f = r / j->local, lt = r - f*j->local;
    const uint64_t t = j->member + lt*j->members, ty = t / j->tiles_x, tx = t - ty*j->tiles_x;
    const uint64_t x = tx*32 + (p & 31), y = ty*32 + (p >> 5);
    o_trace(&c, ((j->first_frame + f)*H + y)*W + x);           /* the index gi.c:88-93 turns back into (frame, y, x) */
  }
  memcpy(j->cnt, c.cnt, sizeof(c.cnt));
  return 0;
}
double oracle_render_tiles(const mi_scene_desc *s, uint64_t first_frame, uint64_t frames, uint32_t member, uint32_t members, float *fb, int threads, uint64_t *counters)
{
  if(threads < 1) threads = 1;
  if(threads > 256) threads = 256;
  if(!o_pixels_from_index) oracle_set_pixels_from_index(2);
  const uint32_t tiles_x = s->width/32, tiles = tiles_x*(s->height/32);
  const uint32_t local = member < tiles ? (tiles - member + members - 1)/members : 0;
  struct timeval t0, t1;
  o_prepare_points(s, (first_frame + frames)*(uint64_t)s->width*s->height);
  gettimeofday(&t0, 0);
  uint64_t counter = 0;
  o_tilejob job[256];
  pthread_t th[256];
  for(int k=0;k<threads;k++)
  {
    memset(job + k, 0, sizeof(o_tilejob));
    job[k].s = s; job[k].fb = fb; job[k].counter = &counter; job[k].items = frames*local*1024u; job[k].first_frame = first_frame;
    job[k].member = member; job[k].members = members; job[k].local = local ? local : 1; job[k].tiles_x = tiles_x; job[k].atomic_fb = threads > 1;
  }
  if(threads == 1) o_tile_worker(job);
  else
  {
    for(int k=0;k<threads;k++) pthread_create(th + k, 0, o_tile_worker, job + k);
    for(int k=0;k<threads;k++) pthread_join(th[k], 0);
  }
  if(counters) for(int k=0;k<threads;k++) for(int i=0;i<8;i++) counters[i] += job[k].cnt[i];
  gettimeofday(&t1, 0);
  return (t1.tv_sec - t0.tv_sec) + 1e-6*(t1.tv_usec - t0.tv_usec);
}

double oracle_render(const mi_scene_desc *s, uint64_t first, uint64_t count, float *fb, int threads, uint64_t *counters)
{
  if(threads < 1) threads = 1;
  if(threads > 256) threads = 256;
  struct timeval t0, t1;
  o_prepare_points(s, first + count);
  gettimeofday(&t0, 0);
  uint64_t counter = first;
  o_job job[256];
  pthread_t th[256];
  for(int k=0;k<threads;k++)
  {
    memset(job + k, 0, sizeof(o_job));
    job[k].s = s; job[k].fb = fb; job[k].counter = &counter; job[k].end = first + count; job[k].atomic_fb = threads > 1;
  }
  if(threads == 1) o_worker(job);
  else
  {
    for(int k=0;k<threads;k++) pthread_create(th + k, 0, o_worker, job + k);
    for(int k=0;k<threads;k++) pthread_join(th[k], 0);
  }
  if(counters) for(int k=0;k<threads;k++) for(int i=0;i<8;i++) counters[i] += job[k].cnt[i];
  gettimeofday(&t1, 0);
  return (t1.tv_sec - t0.tv_sec) + 1e-6*(t1.tv_usec - t0.tv_usec);
}
