/* mi_path.h -- per-path logic of the pt/ptdl hot path (the persistent megakernel of mi_abi.hip keeps the path state in registers).
 *
 *   path_generate   path_init + the length==0 half of path_extend: wavelength, time, thin-lens camera ray
 *                   (src/pathspace.c:13-28,210-249, src/camera.d/thinlens.c:68-128)
 *   path_shade      everything between two rays of one path: finish vertex v from the hit (shader_prepare, emission,
 *                   pdf*G, splat, Russian roulette / next event estimation), then sample the bsdf for the next ray
 *                   (src/pathspace.c:167-292,697-895, src/shader.c:462-590, src/sampler.d/{pt,ptdl}.c)
 *   shadow_resolve  path_visible's verdict for a pending next-event connection (src/pathspace.c:311-344)
 */
#ifndef MI_PATH_H
#define MI_PATH_H

#include "mi_kernels.h"

struct PathState
{
  /* ray to trace next */
  V3 org, dir;              /* ptdl does not keep `org` (nor an origin of the shadow ray): both rays of a vertex start at
                               prev_x + org_eps * direction, formed when the ray is started (ray_origin) -- two registers less */
  float org_eps;            /* prims_offset_ray's epsilon at the vertex the rays leave (0: camera, volume vertex) */
  uint32_t ignore;          /* primitive the ray starts on */
  /* vertex v-1 (the one the ray leaves) */
  V3 prev_x;
  float prev_cos;           /* path_lambert(v-1, omega): |n.omega| or 1 */
  float prev_throughput;    /* v[v-1].throughput */
  /* vertex v being created */
  float throughput;         /* v[v].throughput (after the bsdf sample at v-1) */
  float pdf;                /* v[v].pdf as left by the bsdf sample (projected solid angle) */
  double pdfprod;           /* prod_{k>=1} v[k].pdf, pt.c:30-38 */
  float cur_ior;            /* e[v].vol.ior */
  Media media;
  /* per path */
  float lambda, scramble, pixel_i, pixel_j;   /* (pairs that travel together through LDS lie next to each other: mi_regroup.h) */
  Rng rng;
  unsigned long long index;
  int length;               /* number of complete vertices */
  uint32_t active;          /* path alive: an extension ray is waiting to be traced */
  uint32_t prev_mode;
  /* ptdl: pending shadow ray of the next-event estimate made at the last vertex (bit fields for these small words were tried:
     pt -0.7 %, ptdl +0.3 %) */
  uint32_t sh_pending;
  uint32_t prev_material_modes;
  /* ptdl: pending shadow ray of the next-event estimate made at the last vertex */
  V3 sh_dir;
  float sh_dist, sh_value;
  uint32_t sh_light;         /* the shadow ray starts on ps.ignore, like the extension ray of the same vertex */
  int sh_length;
  /* homogeneous media (MEDIA instantiations only): the volume of the edge under way (e[v].vol) and the free-flight distance
     sampled for it (FLT_MAX: none) */
  Medium cur;
  float clip;
  float time;               /* the path's time in the shutter interval (motion-blurred primitives) */
};

/* origin of the extension ray (shadow = false) or of the pending shadow ray of the vertex at ps.prev_x */
template<bool PTDL>
__device__ __forceinline__ V3 ray_origin(const PathState &ps, bool shadow)
{
  if(!PTDL) return ps.org;
  const V3 d = shadow ? ps.sh_dir : ps.dir;
  return mk3(ps.prev_x.x + ps.org_eps*d.x, ps.prev_x.y + ps.org_eps*d.y, ps.prev_x.z + ps.org_eps*d.z);
}

#ifndef MI_EARLY_KILL
#define MI_EARLY_KILL 1
#endif
/* a splat to be carried out by the wave (splat_wave) after the divergent part of the iteration */
struct SplatReq { bool pending; float c0, c1, c2; };

template<bool RECORD>
__device__ __forceinline__ void rec_vertex(mi_path_record *rec, int v, uint64_t prim, float dist, const V3 x, const V3 n, const V3 gn,
                                           const V3 omega, uint32_t mode, uint32_t flags, float throughput, float pdf, float u, float vv,
                                           const Shading &sh, float eta, int shader)
{
  if(!RECORD || v >= MI_REC_MAX_VERTS) return;
  mi_path_vertex &d = rec->v[v];
  d.prim = prim; d.dist = dist;
  d.x[0] = x.x; d.x[1] = x.y; d.x[2] = x.z;
  d.n[0] = n.x; d.n[1] = n.y; d.n[2] = n.z;
  d.gn[0] = gn.x; d.gn[1] = gn.y; d.gn[2] = gn.z;
  d.omega[0] = omega.x; d.omega[1] = omega.y; d.omega[2] = omega.z;
  d.mode = mode; d.flags = flags; d.throughput = throughput; d.pdf = pdf; d.u = u; d.v = vv;
  d.rd = sh.rd; d.rg = sh.rg; d.em = sh.em; d.roughness = sh.roughness; d.eta = eta; d.shader = shader;
}

/* view_cam_init_frame, src/view.c:903-919: the camera frame at `time` in the shutter interval -- quaternion_slerp of the
 * shutter-open and shutter-close orientations (include/quaternion.h:86-110), the three axes through quaternion_transform
 * (60-75), normalised; position interpolated linearly. Quaternions are w, x, y, z. */
__device__ __forceinline__ void quat_mult(float *in, const float *p)
{ /* quaternion_mult, include/quaternion.h:41-48 */
  const float r0 = in[0], r1 = in[1], r2 = in[2], r3 = in[3];
  in[1] = r0*p[1] + r1*p[0] + r2*p[3] - r3*p[2];
  in[2] = r0*p[2] - r1*p[3] + r2*p[0] + r3*p[1];
  in[3] = r0*p[3] + r1*p[2] - r2*p[1] + r3*p[0];
  in[0] = r0*p[0] - r1*p[1] - r2*p[2] - r3*p[3];
}
__device__ __forceinline__ V3 quat_transform(const float *q, const V3 v)
{
  const float vq[4] = { 0.0f, v.x, v.y, v.z }, inv[4] = { q[0], -q[1], -q[2], -q[3] };
  float res[4] = { q[0], q[1], q[2], q[3] };
  quat_mult(res, vq);
  quat_mult(res, inv);
  return mk3(res[1], res[2], res[3]);
}
__device__ __forceinline__ void camera_frame_at(const mi_camera &cam, float time, V3 &a, V3 &b, V3 &n, V3 &pos)
{
  const float *q = cam.orient, *p = cam.orient_t1;
  float r[4];
  const float cos_theta_2 = q[0]*p[0] + (q[1]*p[1] + q[2]*p[2] + q[3]*p[3]);
  if(fabsf(cos_theta_2) >= 1.0f) { r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3]; }
  else
  {
    const float theta_2 = acosf(cos_theta_2);
    const float sin_theta_2 = mi_sqrt(1.0f - cos_theta_2*cos_theta_2);
    if(fabsf(sin_theta_2) < 1e-10f) for(int k=0;k<4;k++) r[k] = (q[k] + p[k])*.5f;
    else
    {
      const float wa = mi_sinf((1.0f - time)*theta_2)/sin_theta_2;
      const float wb = mi_sinf(time*theta_2)/sin_theta_2;
      for(int k=0;k<4;k++) r[k] = q[k]*wa + p[k]*wb;
    }
  }
  a = normalise3(quat_transform(r, mk3(1.0f, 0.0f, 0.0f)));
  b = normalise3(quat_transform(r, mk3(0.0f, 1.0f, 0.0f)));
  n = normalise3(quat_transform(r, mk3(0.0f, 0.0f, 1.0f)));
  pos = mk3(cam.pos[0]*(1.0f-time) + cam.pos_t1[0]*time, cam.pos[1]*(1.0f-time) + cam.pos_t1[1]*time, cam.pos[2]*(1.0f-time) + cam.pos_t1[2]*time);
}

/* start path `index`: afterwards ps holds the camera ray as the pending extension ray */
/* Tile-owned sharding (mi_render_tiles): work item j of a launch -> the path it stands for. Member g of G owns the 32 x 32 tiles
 * t = g (mod G), tiles counted row by row (the reference's tile size, include/render_tiles.h:156). Items walk the member's tiles round robin:
 * item j is pixel (j / T) mod 1024 of local tile j mod T in frame j / (1024 T) (T = the member's tile count) -- neighbouring items, i.e. the
 * lanes of a wave and the waves of a workgroup, work in DIFFERENT tiles. Dealt out tile by tile instead (a wave inside one tile, a workgroup
 * inside one stretch of the film) the same launch took 19.0 instead of 15.6 ms on cfg 2 and 49.9 instead of 28.4 ms on cfg 3: the workgroups'
 * contiguous item ranges then cover regions of different cost (sky against glass), and the sixteen taps of a tile's splats queue up at the
 * same framebuffer addresses (profiles/r05_tiles.txt).
 * The path's INDEX is what render_sample_path's pixel branch (src/render.d/gi.c:88-93) inverts: frame * W H + y W + x -- so a path has the
 * same index, generator state and pixel whichever member renders it, and G members together render exactly the paths
 * [first_frame W H, (first_frame + frames) W H). */
__device__ __forceinline__ unsigned long long tile_path(const DScene &sc, unsigned long long j, float &px, float &py)
{
  const unsigned long long r = j/sc.tiles_local;
  const uint32_t lt = (uint32_t)(j - r*sc.tiles_local);
  const uint32_t p = (uint32_t)r & 1023u;
  const unsigned long long f = r >> 10;
  const uint32_t t = sc.tile_member + lt*sc.tile_members;
  const uint32_t ty = t/sc.tiles_x, tx = t - ty*sc.tiles_x;
  const uint32_t x = tx*32u + (p & 31u), y = ty*32u + (p >> 5);
  px = (float)x; py = (float)y;
  return (f*sc.height + y)*sc.width + x;
}

template<bool RECORD, bool HALTON, bool MEDIA, class CNT, bool HERO = false>
__device__ __forceinline__ void path_generate(const DScene &sc, PathState &ps, unsigned long long index, mi_path_record *rec, CNT &cnt, float px = -1.0f, float py = -1.0f,
                                              float *lambda_x = nullptr)
{
  /* path_init + first half of path_extend (length == 0), src/pathspace.c:13-28,210-249 */
  MI_BLK(cnt, 0)
  ps.index = index;
  if(sc.pixels_from_index) rng_seed_hashed(ps.rng, ps.index, sc.frame);     /* (why: mi_kernels.h) */
  else rng_seed_jump(ps.rng, sc, ps.index);
  PointSampler<HALTON> pts(sc, ps.rng, index, 0);
  ps.scramble = 0.1f + rng_next(ps.rng)*(0.9f-0.1f);          /* points_rand, not the point sampler: src/pathspace.c:213 */
  const float lf0 = pts.template camera<MI_DIM_LAMBDA>() + 0/(float)1;
  const float lf = lf0 < 1.0f ? lf0 : fmodf(lf0, 1.0f);     /* fmodf(x, 1) == x for 0 <= x < 1; the libm loop only runs otherwise */
  ps.lambda = 360 + (830 - 360)*lf;
  if(HERO)
  { /* MF_COUNT = 4, src/pathspace.c:218-221: the point sampler is asked once PER COMPONENT, component l takes fmodf(number l + l/4, 1) */
#pragma unroll
    for(int l=1;l<4;l++) lambda_x[l-1] = 360 + (830 - 360)*fmodf(pts.template camera<MI_DIM_LAMBDA>() + l/(float)4, 1.0f);
  }
  const float time = pts.template camera<MI_DIM_TIME>()*sc.cam.time_scale;
  if(!HALTON) { (void)rng_next(ps.rng); (void)rng_next(ps.rng); }   /* view_sample_camid twice (one camera): src/pathspace.c:226, src/view.c:846-847 */
  /* camera_sample, src/camera.d/thinlens.c:68-128; everything that does not depend on the random numbers is in sc.cc */
  const mi_camera &cam = sc.cam;
  const DCamConst &cc = sc.cc;
  const float W = cc.W, H = cc.H;
  /* the film position: sampled -- or, DScene.pixels_from_index, inside the pixel the path's index names (path_set_pixel, include/pathspace.h:355-360:
     camera_sample takes a caller's position instead of asking the point sampler, src/camera.d/thinlens.c:117-118). The position inside the
     pixel comes from the two numbers that would have been the sampled position, so the rest of the path's numbers stay where they are. */
  float ci = pts.template camera<MI_DIM_IMAGE_X>(), cj = pts.template camera<MI_DIM_IMAGE_Y>();
  if(sc.pixels_from_index)
  {
    if(px < 0.0f)
    { /* gi.c:88-93: frame = index / (W H), y = rest / W, x = rest - y W (contiguous index ranges: mi_render, mi_trace_paths) */
      const unsigned long long wh = (unsigned long long)sc.width*sc.height, rest = index - (index/wh)*wh;
      const uint32_t y = (uint32_t)(rest/sc.width);
      py = (float)y; px = (float)(uint32_t)(rest - (unsigned long long)y*sc.width);
    }
    ci += px; cj += py;
  }
  else { ci *= W; cj *= H; }
  const float r1 = pts.template camera<MI_DIM_APERTURE_X>();
  const float r2 = pts.template camera<MI_DIM_APERTURE_Y>();
  const float ang = (float)(2*MI_PI_D*(double)r1);
  float sn, cs;
  mi_sincosf(ang, &sn, &cs);                 /* one range reduction for both (same values as sinf/cosf) */
  const float lu = cs*mi_sqrt(r2)*cc.lens_radius;
  const float lv = sn*mi_sqrt(r2)*cc.lens_radius;
  V3 ca = ld3(cam.a), cb = ld3(cam.b), cn = ld3(cam.n), cpos = ld3(cam.pos);
  /* camera motion blur, src/view.c:903-919. Only in the MEDIA ("extended") instantiations: even as a never-taken uniform branch
     it cost the plain kernels 0.9 % (A/B on one box: 2722 vs 2746 Msamples/s), so scenes with a moving camera run those */
  if(MEDIA && cam.moving) camera_frame_at(cam, time, ca, cb, cn, cpos);
  ps.time = time;
  const V3 aoff = mk3(lu*ca.x + lv*cb.x, lu*ca.y + lv*cb.y, lu*ca.z + lv*cb.z);
  const float ki = (ci-.5f*W)*cc.f_rg, kj = (cj-.5f*H)*cc.f_up;
  V3 om = mk3(cc.f_dir*cn.x + (ki*ca.x + kj*cb.x) - aoff.x,
              cc.f_dir*cn.y + (ki*ca.y + kj*cb.y) - aoff.y,
              cc.f_dir*cn.z + (ki*ca.z + kj*cb.z) - aoff.z);
  om = normalise3(om);
  const float pdf_a = cc.pdf_a, sensor = cc.sensor;
  const float dt = dot3(om, cn);
  const float dot4 = dt*dt*dt*dt;
  ps.pixel_i = (float)DCLAMP((double)ci, 0.0, (double)cc.Wc);
  ps.pixel_j = (float)DCLAMP((double)cj, 0.0, (double)cc.Hc);
  const float G = dot4/cc.fl2;
  const float pdf_v = cc.pdf_v;
  ps.pdf = pdf_v*pdf_a/G;
  const V3 x0 = mk3(cpos.x + aoff.x, cpos.y + aoff.y, cpos.z + aoff.z);
  const float thr0 = sensor*G/cc.pdf_av;
  ps.org = x0; ps.dir = om; ps.ignore = MI_NOPRIM;
  ps.prev_x = x0; ps.org_eps = 0.0f;
  ps.prev_cos = fabsf(dot3(cn, om));        /* path_lambert on the sensor vertex */
  ps.prev_throughput = thr0;
  ps.prev_mode = s_sensor;
  ps.throughput = thr0;
  ps.pdfprod = 1.0;
  ps.cur_ior = 1.0f;
  ps.media.ids = 0; ps.media.count = 0; ps.media.broken = 0;
  ps.cur = medium_vacuum(); ps.clip = FLT_MAX;                /* shader_exterior_medium, src/shader.c:544-565: vacuum or the global medium */
  if(MEDIA) ps.cur = shape_interior_medium(sc, -1, ps.lambda);
  ps.length = 1;
  ps.active = 1;
  ps.prev_material_modes = s_sensor;
  MI_COUNT(cnt, 6, 1);                                  /* the sensor vertex */
  if(RECORD)
  {
    rec->index = ps.index; rec->pixel_i = ps.pixel_i; rec->pixel_j = ps.pixel_j; rec->lambda = ps.lambda;
    rec->time = time; rec->scramble = ps.scramble; rec->throughput = 0.0f; rec->length = 1; rec->num_splats = 0;
    Shading z; z.roughness = z.rs = z.rd = z.rg = z.em = 0.0f;
    rec_vertex<RECORD>(rec, 0, MI_PRIMID_INVALID, 0.0f, x0, cn, cn, mk3(0, 0, 0), s_sensor, 0, thr0, 1.0f, 0.0f, 0.0f, z, 0.0f, -1);
  }
}

/* the shadow ray of the pending next-event connection has been traced into `hit` */
/* the splat of a next-event connection that path_visible has passed (ps.sh_value, ps.sh_length) */
template<bool RECORD, class CNT>
__device__ __forceinline__ void shadow_splat(const DScene &sc, PathState &ps, mi_path_record *rec, CNT &cnt, SplatReq &splat)
{
  ps.sh_pending = 0;
  {
    const float value = ps.sh_value;
    const bool ok = splat_value_ok(value);
    float col[3] = {0.0f, 0.0f, 0.0f};
    if(ok) spectrum_to_xyz(sc, ps.lambda, value, col);
    if(RECORD)
    {
          if(rec->num_splats < MI_REC_MAX_SPLATS)
      {
        mi_path_splat &sp = rec->splat[rec->num_splats++];
        sp.length = ps.sh_length; sp.tech = s_tech_nee; sp.value = value;
        sp.col[0] = col[0]; sp.col[1] = col[1]; sp.col[2] = col[2];
      }
    }
    if(ok)
    {
      MI_COUNT(cnt, 5, 1);
      if(!RECORD) { splat.pending = true; splat.c0 = col[0]; splat.c1 = col[1]; splat.c2 = col[2]; }
    }
  }
}

template<bool RECORD, class CNT>
__device__ __forceinline__ void shadow_resolve(const DScene &sc, PathState &ps, const Hit &hit, mi_path_record *rec, CNT &cnt, SplatReq &splat)
{
   /* path_visible, src/pathspace.c:311-344: closest hit up to the emitter's primitive (all surfaces in scope are opaque) */
  ps.sh_pending = 0;
  const bool visible = (hit.dist >= ps.sh_dist) || (hit.prim == MI_NOPRIM) || (hit.prim == (ps.sh_light & ~MI_LIGHT_ANYHIT));
  if(visible) shadow_splat<RECORD>(sc, ps, rec, cnt, splat);
}

/* ------------------------------------------------------------------------------------------ homogeneous media
 * free-flight distance of the extension ray about to be traced: shader_vol_sample, src/shader.c:76-106, called from
 * path_propagate before the ray is cast when the edge's medium scatters (src/pathspace.c:717-751). Returns the distance the
 * traversal is clipped to (FLT_MAX: vacuum or a purely absorbing medium). */
template<bool PTDL, bool HALTON>
__device__ __forceinline__ float media_free_flight(const DScene &sc, PathState &ps)
{
  ps.clip = FLT_MAX;
  if(ps.cur.med >= 0 && ps.cur.mu_s > 0.0f)
  {
    const int v = ps.length;
    PointSampler<HALTON> pts(sc, ps.rng, ps.index, v == 1 ? 7 : rand_beg_extend<PTDL>(v));
    const float rf = pts(0);                                   /* s_dim_free_path */
    float dist = -logf(1.0f - rf)/ps.cur.mu_t;
    if(!(dist > 0.0)) dist = 1e-15;
    ps.clip = dist;
  }
  return ps.clip;
}

/* transmittance of the medium `m` over `dist`, and the pdf factor of the edge when its end point is on geometry
 * (shader_vol_transmittance src/shader.c:48-74, shader_vol_pdf 108-131) */
__device__ __forceinline__ float media_transmittance(const Medium &m, float dist) { return m.med >= 0 ? expf(-dist*m.mu_t) : 1.0f; }
__device__ __forceinline__ float media_pdf_to_surface(const Medium &m, float dist) { return (m.med >= 0 && m.mu_s > 0.0f) ? expf(-dist*m.mu_t) : 1.0f; }

/* next event estimation from a one-burst emitter record (DLight, mi_device.h: static triangle / quad emitters with colour-only
 * materials) in the EXTENDED kernels, when the scene's emitters allow it (sc.lights): lights_sample_next_event + prims_sample +
 * prims_retime + the emitter's shader_prepare (src/lights.d/list.c:130-174, src/prims.c:178-252) from ten 16-B loads instead of the
 * chain emitter list -> primitive -> shading record -> material -> ops. The arithmetic of the plain kernels' branch in path_shade
 * (kept inline there: as a function it cost that kernel 3.6 %). */
__device__ __forceinline__ void light_record_sample(const DScene &sc, float r1, float r2, float r3, float lambda, const V3 from,
                                                    uint32_t &lpe, Surf &ls, Shading &lsh, float &lpdf, float &ldist, V3 &ol)
{
  const uint32_t t = sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
  const float4 *lq = (const float4 *)(sc.lights + t);
  const float4 q0 = lq[0], q1 = lq[1], q2 = lq[2], q3 = lq[3], q4 = lq[4], q5 = lq[5], q6 = lq[6], q7 = lq[7], q8 = lq[8], q9 = lq[9];
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
  const V3 na = second ? mk3(q4.z, q4.w, q5.x) : mk3(q3.w, q4.x, q4.y);      /* n2 : n1 */
  const V3 nb = second ? mk3(q5.y, q5.z, q5.w) : mk3(q4.z, q4.w, q5.x);      /* n3 : n2 */
  ls.gn = second ? mk3(q6.w, q7.x, q7.y) : mk3(q6.x, q6.y, q6.z);
  const float w = 1.0f - u - vv;
  ls.n = normalise3(mk3(u*nb.x + vv*na.x + w*n0.x, u*nb.y + vv*na.y + w*n0.y, u*nb.z + vv*na.z + w*n0.z));
  ls.flags = 0;
  const float ec[3] = { q7.z, q7.w, q8.x };
  lsh.em = q8.y*spectrum_eval(ec, lambda);
  lsh.roughness = q8.z;
  lpdf = q8.w;
}

/* the extension ray ended at the sampled free-flight distance ps.clip before any geometry: a volume vertex
 * (path_propagate src/pathspace.c:745-751,771-776; shader_prepare src/shader.c:476-501; manifold_init manifold.h:236-246;
 * phase function src/shaders/medium_rgb.c:61-102; next event estimation as for surfaces) */
template<bool RECORD, bool PTDL, bool HALTON, bool MB, class CNT>
__device__ __forceinline__ void path_shade_volume(const DScene &sc, PathState &ps, mi_path_record *rec, CNT &cnt)
{
  const int v = ps.length;
  bool alive = true;
  const V3 omega = ps.dir;
  const float dist = ps.clip;
  const Medium med = ps.cur;
  Surf sf;
  const V3 rorg = ray_origin<PTDL>(ps, false);
  sf.x = mk3(rorg.x + dist*ps.dir.x, rorg.y + dist*ps.dir.y, rorg.z + dist*ps.dir.z);
  sf.n = omega; sf.gn = omega;                               /* the frame looks along the incoming direction */
  get_scrambled_onb(ps.scramble, sf.n, sf.a, sf.b);
  sf.u = sf.v = sf.s = sf.t = 0.0f; sf.flags = 0;
  const uint32_t material_modes = s_volume | s_glossy;
  /* edge: pdf = transmittance * mu_t at the sampled distance (src/shader.c:95-97), on-"surface" pdf with G = cos / d^2, cos = 1 here */
  const float eT = expf(-dist*med.mu_t);
  const float epdf = eT*med.mu_t;
  const float G = ps.prev_cos*1.0f/(dist*dist);
  const float vpdf = (ps.pdf*epdf)*G;
  ps.pdfprod *= (double)vpdf;
  ps.length++;
  MI_COUNT(cnt, 6, 1);
  const float vthr = ps.throughput*(eT/epdf);
  if(RECORD)
  {
    Shading z; z.roughness = z.rs = z.rd = z.rg = z.em = 0.0f;
    rec_vertex<RECORD>(rec, v, MI_PRIMID_INVALID, dist, sf.x, sf.n, sf.gn, omega, s_absorb, 0, vthr, vpdf, 0.0f, 0.0f, z, 0.0f, med.med);
    rec->length = ps.length; rec->throughput = (ps.throughput*0.0f)/epdf;
  }
#if MI_EARLY_KILL
  { /* as in path_shade: what the state held about the previous vertex and the arrived ray is read, what replaces it is known */
    ps.prev_x = sf.x; ps.org_eps = 0.0f; ps.ignore = MI_NOPRIM;
    if(!PTDL) ps.org = sf.x;
    ps.prev_cos = 0.0f; ps.throughput = 0.0f; ps.pdf = 0.0f;
    ps.prev_material_modes = material_modes;
    if(PTDL) { ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0; }
  }
#endif
  if(PTDL && ps.length >= (int)sc.max_verts) alive = false;
  if(PTDL && alive)
  { /* next event estimation from the volume vertex, ptdl.c:136-148 */
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
      Shading lsh;
      V3 ol;
      float ldist, lpdf;
      if(!MB && sc.lights != nullptr) light_record_sample(sc, r1, r2, r3, ps.lambda, sf.x, lpe, ls, lsh, lpdf, ldist, ol);
      else
      {
        const uint32_t t = sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
        lpe = sc.light_prim[t];                                 /* bit 31: any-hit shadow ray allowed (mi_device.h) */
        const uint32_t lp = lpe & ~MI_LIGHT_ANYHIT;
        ls.x = prim_sample<MB>(sc.prims[lp], sc.primgeo[lp], r2, r3, ls.u, ls.v, MB ? sc.prims_t1 + lp : nullptr, ps.time);
        ol = sub3(ls.x, sf.x);
        ldist = mi_sqrt(dot3(ol, ol));
        const double il = 1./(double)ldist;
        ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
        const uint4 lhead = *(const uint4 *)&sc.primgeo[lp];
        surface_setup<MB>(sc, lp, lhead, ol, ps.scramble, ls, ps.time);
        run_prepare_ops(sc, sc.materials[lhead.y], sc.materials[lhead.y].num_ops, ls, ps.lambda, lsh);
        lpdf = sc.light_L[t];
      }
      float edf = lsh.em/lpdf;
      if(lsh.roughness > 1.0f-1e-4f) edf = (float)((double)edf*((double)1.0f/MI_PI_D));
      else
      {
        const float phongexp = 2.0f/(lsh.roughness*lsh.roughness) - 2.0f;
        edf = (float)((double)edf*((double)(powf(-dot3(ls.gn, ol), phongexp)*(phongexp + 2.0f))/(2.0f*MI_PI_D)));
      }
      lpdf = lpdf*sc.p_geo;
      edf = edf/sc.p_geo;
      if(edf > 0.0f)
      {
        const float hg = eval_hg(med.g, omega, ol);
        const float bsdf = med.mu_s*hg;                        /* medium_rgb.c:98-102 */
        if(bsdf > 0.0f)
        { /* prims_get_ray, src/prims.c:390-492: no offset at a volume vertex, the usual one at the emitter */
          const float eps = 1e-4f*DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)));
          V3 rd = sub3(ls.x, sf.x);
          rd = scale3(rd, mi_rcp(mi_sqrt(dot3(rd, rd))));
          const V3 ro = sf.x;
          const V3 dv = mk3(ls.x.x - eps*rd.x - ro.x, ls.x.y - eps*rd.y - ro.y, ls.x.z - eps*rd.z - ro.z);
          const float total_dist = mi_sqrt(dot3(dv, dv));
          if(!(dot3(ls.gn, rd) >= 0) && total_dist > 0.0f)
          {
            const float Gn = 1.0f*fabsf(dot3(ls.n, ol))/(ldist*ldist);
            const float T = media_transmittance(med, ldist);
            float tn = ((vthr*bsdf)*(T*edf))*Gn;
            tn = tn + (vthr*bsdf)*((0.0f*Gn)/lpdf);
            const float wn = lpdf/(lpdf + 0.0f/T);
            tn = tn*wn;
            const float pe = (media_pdf_to_surface(med, ldist)*hg)*Gn;
            const double pp = ps.pdfprod;
            const double our = (double)(1.0f*lpdf)*pp, other = (double)pe*pp;
            const float wm = (float)our/(float)(other + our);
            if(tn/1.0f > 0.0f)
            {
              ps.sh_pending = 1;
              ps.prev_x = sf.x; ps.org_eps = 0.0f;                  /* = ro: no offset at a volume vertex */
              ps.sh_dir = rd; ps.sh_dist = total_dist;
              ps.sh_light = lpe; ps.ignore = MI_NOPRIM;
              ps.sh_value = (tn/1.0f)*wm;
              ps.sh_length = ps.length + 1;
            }
          }
        }
      }
    }
  }
  if(alive && ps.length >= (int)sc.max_verts) alive = false;
  if(alive && !(vthr > 0.0f))
  {
    alive = false;
    if(RECORD && v < MI_REC_MAX_VERTS) { rec->v[v].throughput = 0.0f; rec->v[v].mode = s_absorb; }
  }
  if(alive)
  { /* medium_rgb.c:61-72 + sample_hg, include/sampler_common.h:286-316 (the two numbers are call arguments: omega_y is drawn first) */
    PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v + 1));
    const float r2 = pts(MI_DIM_OMEGA_Y);
    const float r1 = pts(MI_DIM_OMEGA_X);
    const float g = med.g;
    float o0, o1, o2, pdf;
    if(g == 0.0f)
    { /* sample_sphere */
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
    const float nthr = vthr*med.mu_s;
    const uint32_t vmode = s_glossy | s_volume;
    if(nthr <= 0.0f) alive = false;
    if(RECORD && v < MI_REC_MAX_VERTS) rec->v[v].mode = alive ? vmode : (uint32_t)s_absorb;
    if(alive)
    { /* the next ray starts at the vertex itself: prims_offset_ray only applies on geometry, src/pathspace.c:759-761 */
      ps.org = sf.x;
      ps.dir = wo;
      ps.ignore = MI_NOPRIM;
      ps.prev_x = sf.x; ps.org_eps = 0.0f;
      ps.prev_cos = 1.0f;                                      /* path_lambert at a volume vertex */
      ps.prev_throughput = vthr;
      ps.prev_mode = vmode;
      ps.prev_material_modes = material_modes;
      ps.throughput = nthr;
      ps.pdf = pdf;
    }
  }
  if(!alive) { ps.active = 0; cnt.c[4]++; }
}

/* the extension ray left the scene: environment vertex, src/pathspace.c:856-873; black sky => nothing to add, the path ends.
 * (A function of its own since round 4: the kernels that trade vertices between waves, mi_regroup.h, end such paths before the exchange.) */
template<bool RECORD, bool MEDIA, class CNT>
__device__ __forceinline__ void path_escape(const DScene &sc, PathState &ps, mi_path_record *rec, CNT &cnt)
{
  const int v = ps.length;
  const V3 omega = ps.dir;
  {
    const float G = ps.prev_cos;                   /* path_G with an environment end point */
    /* a purely absorbing medium reaches here with clip == FLT_MAX: transmittance exp(-FLT_MAX mu_t), pdf 1 (src/shader.c:99-100) */
    const float env_T = MEDIA ? media_transmittance(ps.cur, FLT_MAX) : 1.0f;
    const float vpdf = MEDIA ? (ps.pdf*1.0f)*G : ps.pdf*G;
    ps.pdfprod *= (double)vpdf;
    ps.length++;
    MI_COUNT(cnt, 6, 1);
    if(RECORD)
    {
      const V3 x = mk3(ps.prev_x.x + sc.far_dist*omega.x, ps.prev_x.y + sc.far_dist*omega.y, ps.prev_x.z + sc.far_dist*omega.z);
      Shading z; z.roughness = 1.0f; z.rs = z.rd = z.rg = z.em = 0.0f;
      rec_vertex<RECORD>(rec, v, MI_PRIMID_INVALID, FLT_MAX, x, mk3(0, 0, 0), mk3(0, 0, 0), omega, s_absorb, s_environment,
                         MEDIA ? ps.throughput*(env_T/1.0f) : ps.throughput, vpdf, 0.0f, 0.0f, z, 0.0f, -1);
      rec->length = ps.length; rec->throughput = 0.0f;
    }
  }
  ps.active = 0; cnt.c[4]++;
}

/* the extension ray ps.org/ps.dir has been traced into `hit`: create vertex v = ps.length, then either end the path
 * (ps.active = 0) or leave the next extension ray (and, for ptdl, possibly a shadow ray) in ps */
template<bool RECORD, bool PTDL, bool HALTON, bool MEDIA = false, bool MB = false, class CNT>
__device__ __forceinline__ void path_shade(const DScene &sc, PathState &ps, const Hit &hit, const uint32_t *shape_material, const float *shape_L,
                                           mi_path_record *rec, CNT &cnt, SplatReq &splat)
{
  if(MEDIA && hit.prim == MI_NOPRIM && ps.clip < FLT_MAX)
  {
    path_shade_volume<RECORD, PTDL, HALTON, MB>(sc, ps, rec, cnt);
    return;
  }

  MI_BLK(cnt, 1)
  const int v = ps.length;                         /* index of the vertex being created */
  bool alive = true;
  const V3 omega = ps.dir;
#if MI_EARLY_KILL
  if(PTDL)
  { /* the connection of the previous vertex is resolved before its extension ray is shaded: what it left in the path state is dead,
       but only overwritten if this vertex makes a connection of its own -- seven registers through the whole function otherwise */
    ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0;
  }
#endif
  if(hit.prim == MI_NOPRIM)
  { /* left the scene: environment vertex, src/pathspace.c:856-873; black sky => nothing to add, path ends */
    const float G = ps.prev_cos;                   /* path_G with an environment end point */
    /* a purely absorbing medium reaches here with clip == FLT_MAX: transmittance exp(-FLT_MAX mu_t), pdf 1 (src/shader.c:99-100) */
    const float env_T = MEDIA ? media_transmittance(ps.cur, FLT_MAX) : 1.0f;
    const float vpdf = MEDIA ? (ps.pdf*1.0f)*G : ps.pdf*G;
    ps.pdfprod *= (double)vpdf;
    ps.length++;
    MI_COUNT(cnt, 6, 1);
    if(RECORD)
    {
      const V3 x = mk3(ps.prev_x.x + sc.far_dist*omega.x, ps.prev_x.y + sc.far_dist*omega.y, ps.prev_x.z + sc.far_dist*omega.z);
      Shading z; z.roughness = 1.0f; z.rs = z.rd = z.rg = z.em = 0.0f;
      rec_vertex<RECORD>(rec, v, MI_PRIMID_INVALID, FLT_MAX, x, mk3(0, 0, 0), mk3(0, 0, 0), omega, s_absorb, s_environment,
                         MEDIA ? ps.throughput*(env_T/1.0f) : ps.throughput, vpdf, 0.0f, 0.0f, z, 0.0f, -1);
      rec->length = ps.length; rec->throughput = 0.0f;
    }
    alive = false;
  }
  else
  {
    /* shader_prepare, src/shader.c:462-542 */
    MI_BLK(cnt, 2)
    Surf sf;
    const V3 rorg = ray_origin<PTDL>(ps, false);
    sf.x = mk3(rorg.x + hit.dist*ps.dir.x, rorg.y + hit.dist*ps.dir.y, rorg.z + hit.dist*ps.dir.z);
    sf.u = hit.u; sf.v = hit.v;
    /* the record's header first (one 16-B load): it names the material, whose fetch is then under way while the
       surface is set up */
    const DPrimGeo &pshade = sc.primgeo[hit.prim];
    const uint4 head = *(const uint4 *)&pshade;                     /* type, material, uv0, primid_lo */
    const DMaterial &mat = sc.materials[head.y];
    const uint4 mhead = *(const uint4 *)&mat;                       /* bsdf, num_ops, param[0..1] */
    const uint32_t mat_bsdf = mhead.x;
    const float mat_p0 = __uint_as_float(mhead.z), mat_p1 = __uint_as_float(mhead.w);
    surface_setup<MB>(sc, hit.prim, head, omega, ps.scramble, sf, ps.time);
    MI_PHASE(cnt, 2)
    const uint32_t shape = (head.w >> 3) & 0x1fffffffu;             /* MI_PRIMID_SHAPE */
    Shading sh;
    run_prepare_ops(sc, mat, mhead.y, sf, ps.lambda, sh);
    uint32_t material_modes = 0;
    float eta_ratio = 1.0f;      /* path_eta_ratio(v): e[v].vol.ior / ior behind the interface, src/pathspace.c:117-124 */
    /* (only the dielectric and the metal read it -- and the path records: a wave of diffuse vertices, which the exchange between waves
       makes the common case, skips the medium stack's walk and the look-up of the ior behind the interface) */
    if(RECORD || mat_bsdf != MI_BSDF_DIFFUSE)
    {
      Media hyp = ps.media;
      media_apply(hyp, shape, (sf.flags & s_inside) != 0);
      float interior_self = 1.0f;
      if(mat_bsdf == MI_BSDF_DIELECTRIC) interior_self = eta_from_abbe(mat_p0, mat_p1, ps.lambda);
      if(hyp.broken) eta_ratio = -1.0f;
      else
      {
        const int top = media_top_shape(hyp);
        const float ior2 = (top == (int)shape) ? interior_self : shape_interior_ior(sc, shape_material, top, ps.lambda);
        eta_ratio = ps.cur_ior/ior2;
      }
    }
    if(mat_bsdf == MI_BSDF_DIFFUSE) { if(sh.rd > 0.0f) material_modes = s_reflect | s_diffuse; }
    else if(mat_bsdf == MI_BSDF_DIELECTRIC)
    {
      material_modes = s_reflect | s_transmit;
      if(fabsf(1.0f - eta_ratio/1.0f) < 1e-3f) sh.roughness = 0.0f;
      if(sh.roughness > GLOSSY_THR) material_modes |= s_glossy; else material_modes |= s_specular;
    }
    else if(mat_bsdf == MI_BSDF_METAL)
    {
      material_modes = s_reflect;
      if(sh.roughness > 1e-4f) material_modes |= s_glossy; else material_modes |= s_specular;
    }

    MI_PHASE(cnt, 3)
    /* self-intersection, src/pathspace.c:807-820 */
    const uint32_t type = MB ? head.x & 7u : head.x;                /* the motion-blur kernels flag moving primitives in bit 3 (MI_GEO_MB) */
    if((type > 2 || hit.dist < 1e-4f) && hit.prim == ps.ignore)
    {
      alive = false;
      if(RECORD)
      {
        rec->length = ps.length; rec->throughput = 0.0f;
        /* path_extend after a failed propagate, src/pathspace.c:252-257: the vertex the ray left absorbs unless it emits */
        if(v >= 1 && v-1 < MI_REC_MAX_VERTS && !(ps.prev_mode & s_emit)) rec->v[v-1].mode = s_absorb;
      }
    }
    else
    {
      uint32_t mode = s_absorb;
      if(sh.em > 0.0f && !(sf.flags & s_inside)) { mode = s_emit; material_modes = s_emit; }
      /* path_extend tail, src/pathspace.c:261-270 */
      const float G = ps.prev_cos*fabsf(dot3(sf.n, omega))/(hit.dist*hit.dist);
      /* edge in a medium, geometry before the sampled distance: pdf = transmittance (scattering medium, src/pathspace.c:834-838)
         or transmittance with pdf 1 (absorbing only, src/shader.c:99-100) */
      const float eT = MEDIA ? media_transmittance(ps.cur, hit.dist) : 1.0f;
      const float epdf = MEDIA ? media_pdf_to_surface(ps.cur, hit.dist) : 1.0f;
      const float vpdf = MEDIA ? (ps.pdf*epdf)*G : ps.pdf*G;
      const double pp_before = ps.pdfprod;
      ps.pdfprod *= (double)vpdf;
      ps.length++;
      MI_COUNT(cnt, 6, 1);
      float path_throughput = 0.0f;
      if(mode & s_emit)
      { /* lights_eval_vertex, src/lights.d/list.c:242-275 */
        float Le = 0.0f;
        if(sh.em > 0.0f && !(dot3(sf.gn, omega) >= 0.0))
        {
          float edf;
          if(sh.roughness > 1.0f-1e-4f) edf = (float)(1.0f/MI_PI_D);
          else
          {
            const float phongexp = 2.0f/(sh.roughness*sh.roughness) - 2.0f;
            edf = (float)((double)(powf(fabsf(dot3(sf.gn, omega)), phongexp)*(phongexp+2.0f))/(2.0f*MI_PI_D));
          }
          Le = edf*sh.em;
        }
        path_throughput = MEDIA ? (ps.throughput*0.0f)/epdf + (ps.throughput*(eT/epdf))*Le : 0.0f + ps.throughput*Le;
      }
      float vthr = MEDIA ? ps.throughput*(eT/epdf) : ps.throughput;
      if(RECORD)
      {
        rec_vertex<RECORD>(rec, v, MI_GEO_PRIMID(pshade), hit.dist, sf.x, sf.n, sf.gn, omega, mode, sf.flags, vthr, vpdf, sf.u, sf.v, sh,
                           eta_ratio, (int)head.y);
        rec->length = ps.length; rec->throughput = path_throughput;
      }
      if(mode & s_emit)
      { /* sampler_create_path: src/sampler.d/pt.c:45-52 / src/sampler.d/ptdl.c:116-121 */
        MI_BLK(cnt, 3)
        float w;
        if(PTDL)
        { /* balance heuristic against next event estimation, ptdl.c:78-88 + nee_pdf, include/pathspace/nee.h:21-47 */
          float nee = 0.0f;
          if(ps.length >= 3 && (ps.prev_material_modes & (s_diffuse | s_glossy)) && sc.p_geo > 0) nee = sc.p_geo*shape_L[shape];
          const double pp = pp_before;
          const double our = (double)vpdf*pp, other = (double)(1.0f*nee)*pp;
          w = (float)our/(float)(other + our);
        }
        else
        {
          const float fp = (float)ps.pdfprod;
          w = fp/fp;
        }
        const float value = PTDL ? path_throughput*w : w*path_throughput;
        const bool ok = splat_value_ok(value);
        float col[3] = {0.0f, 0.0f, 0.0f};
        if(ok) spectrum_to_xyz(sc, ps.lambda, value, col);
        if(RECORD && rec->num_splats < MI_REC_MAX_SPLATS)
        {
          mi_path_splat &sp = rec->splat[rec->num_splats++];
          sp.length = ps.length; sp.tech = s_tech_extend; sp.value = value;
          sp.col[0] = col[0]; sp.col[1] = col[1]; sp.col[2] = col[2];
        }
        if(ok)
        {
          MI_COUNT(cnt, 5, 1);
          if(!RECORD) { splat.pending = true; splat.c0 += col[0]; splat.c1 += col[1]; splat.c2 += col[2]; }   /* += : a chained ptdl lane may have splatted its connection this iteration (same pixel) */
        }
        if(!PTDL && ps.length > 3)
        { /* path_russian_roulette, src/pathspace.c:273-292 */
          const float p_survival = DMIN(1.0f, vthr/ps.prev_throughput);
          PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v));   /* the last vertex's dimensions, src/pathspace.c:278-281 */
          const float rr = pts(MI_DIM_RUSSIAN_R);
          if(rr >= p_survival) { vthr = vthr*mi_rcp(1.0f-p_survival); alive = false; }
          else vthr = vthr*mi_rcp(p_survival);
          if(RECORD && v < MI_REC_MAX_VERTS)
          {
            rec->v[v].throughput = vthr;
            rec->v[v].pdf = alive ? vpdf*p_survival : vpdf*(1.0f-p_survival);
          }
        }
      }
#if MI_EARLY_KILL
      /* Everything the path state holds about vertex v-1 and the ray that arrived has been read by now, and what replaces it is known
         -- the compiler cannot see that, because the assignments at the end of the function are conditional (the path may end; nobody
         reads its state then). Made here, the old values do not occupy registers through next event estimation and the bsdf sample,
         where the kernel's register pressure peaks: the vertex itself (sf.x) is what the next rays leave from. */
      {
        const float eps0 = DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)))*1e-4f;
        ps.prev_x = sf.x; ps.org_eps = eps0; ps.ignore = hit.prim;
        if(!PTDL) ps.org = sf.x;
        ps.prev_cos = 0.0f; ps.throughput = 0.0f; ps.pdf = 0.0f;
        ps.prev_material_modes = material_modes;
      }
#endif
      if(PTDL && ps.length >= (int)sc.max_verts) alive = false;           /* ptdl.c:122 */
      if(PTDL && alive)
      { /* next event estimation at vertex v: ptdl.c:136-148, nee_sample include/pathspace/nee.h:87-243 */
        (void)rng_next(ps.rng);                                            /* points_rand < nee_probability == 1 */
        if(material_modes & (s_diffuse | s_glossy))
        {
          PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_nee(v + 1));          /* the next-event vertex is v+1, nee.h:92,108 */
          const float rnd = pts(MI_DIM_NEE_LIGHT1);
          if(!(rnd < sc.p_sky) && rnd < sc.p_sky + sc.p_geo)
          { /* lights_sample_next_event, src/lights.d/list.c:130-174 (arguments drawn right to left) */
            MI_BLK(cnt, 4)
            const float r3 = pts(MI_DIM_NEE_Y);
            const float r2 = pts(MI_DIM_NEE_X);
            const float r1 = pts(MI_DIM_NEE_LIGHT2);
            uint32_t lpe, lp;
            Surf ls;
            Shading lsh;
            float lpdf, ldist;
            V3 ol;
            if(!MEDIA && !MB)
            { /* plain kernels: every emitter primitive is a static triangle / quad with a colour-only material and has a DLight record
                 (mi_scene_create sends every other scene to the extended kernels): one burst of ten 16-B loads instead of the chain
                 emitter list -> primitive -> shading record -> material -> ops. Same arithmetic as the generic branch below. */
              const uint32_t t = sc.num_lights <= 4 ? sample_cdf4(sc.light_cdf4, (int)sc.num_lights, r1) : sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
              float4 q0, q1, q2, q3, q4, q5, q6, q7, q8, q9;
              if(MI_LIGHTS_LDS && sc.num_lights <= MI_LIGHTS_LDS)
              { /* from LDS (staged by lds_setup): short latency, the compiler places each read next to its use */
                const float4 *lq = lights_lds<HALTON>() + t*(uint32_t)(sizeof(DLight)/16);
                q0 = lq[0]; q1 = lq[1]; q2 = lq[2]; q3 = lq[3]; q4 = lq[4]; q5 = lq[5]; q6 = lq[6]; q7 = lq[7]; q8 = lq[8]; q9 = lq[9];
              }
              else
              {
                const float4 *lq = (const float4 *)(sc.lights + t);
                q0 = lq[0]; q1 = lq[1]; q2 = lq[2]; q3 = lq[3]; q4 = lq[4]; q5 = lq[5]; q6 = lq[6]; q7 = lq[7]; q8 = lq[8]; q9 = lq[9];
              }
              lpe = __float_as_uint(q9.x); lp = lpe & ~MI_LIGHT_ANYHIT;
              const bool quad = __float_as_uint(q9.y) == MI_PRIM_QUAD;
              const V3 v0 = mk3(q0.x, q0.y, q0.z), v1 = mk3(q0.w, q1.x, q1.y), v2 = mk3(q1.z, q1.w, q2.x), v3 = mk3(q2.y, q2.z, q2.w);
              /* prims_sample + prims_retime, src/prims.c:178-252 */
              float hu, hv;
              if(quad) { hu = r2; hv = r3; }
              else { const float a = mi_sqrt(r2); hu = r3*a; hv = (1.0f-r3)*a; }
              const bool second = quad && !(hv >= hu);
              const float u = second ? hu - hv : hu;
              const float vv = !quad ? hv : second ? hv : hv - hu;
              ls.x = second ? tri_retime(v0, v2, v3, u, vv) : tri_retime(v0, v1, v2, u, vv);
              ls.u = hu; ls.v = hv;
              ol = sub3(ls.x, sf.x);
              ldist = mi_sqrt(dot3(ol, ol));
              const double il = 1./(double)ldist;
              ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
              /* prims_get_normal_time for triangles / quads (surface_setup); the flip towards the ray changes neither |n.omega| nor gn */
              const V3 n0 = mk3(q3.x, q3.y, q3.z);
              const V3 na = second ? mk3(q4.z, q4.w, q5.x) : mk3(q3.w, q4.x, q4.y);      /* n2 : n1 */
              const V3 nb = second ? mk3(q5.y, q5.z, q5.w) : mk3(q4.z, q4.w, q5.x);      /* n3 : n2 */
              ls.gn = second ? mk3(q6.w, q7.x, q7.y) : mk3(q6.x, q6.y, q6.z);
              const float w = 1.0f - u - vv;
              ls.n = normalise3(mk3(u*nb.x + vv*na.x + w*n0.x, u*nb.y + vv*na.y + w*n0.y, u*nb.z + vv*na.z + w*n0.z));
              ls.flags = 0;
              /* the material's prepare chain, all plain colours (run_prepare_ops): emission and the last line's roughness */
              const float ec[3] = { q7.z, q7.w, q8.x };
              lsh.em = q8.y*spectrum_eval(ec, ps.lambda);
              lsh.roughness = q8.z;
              lpdf = q8.w;
            }
            else if(MEDIA && !MB && sc.lights != nullptr)
            { /* extended kernels, emitters with records */
              light_record_sample(sc, r1, r2, r3, ps.lambda, sf.x, lpe, ls, lsh, lpdf, ldist, ol);
              lp = lpe & ~MI_LIGHT_ANYHIT;
            }
            else
            {
              const uint32_t t = sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
              lpe = sc.light_prim[t]; lp = lpe & ~MI_LIGHT_ANYHIT;   /* bit 31: any-hit shadow ray allowed (mi_device.h) */
              ls.x = prim_sample<MB>(sc.prims[lp], sc.primgeo[lp], r2, r3, ls.u, ls.v, MB ? sc.prims_t1 + lp : nullptr, ps.time);
              ol = sub3(ls.x, sf.x);
              ldist = mi_sqrt(dot3(ol, ol));
              const double il = 1./(double)ldist;
              ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
              const uint4 lhead = *(const uint4 *)&sc.primgeo[lp];
              surface_setup<MB>(sc, lp, lhead, ol, ps.scramble, ls, ps.time);
              run_prepare_ops(sc, sc.materials[lhead.y], sc.materials[lhead.y].num_ops, ls, ps.lambda, lsh);
              lpdf = sc.light_L[t];
            }
            float edf = lsh.em/lpdf;
            if(lsh.roughness > 1.0f-1e-4f) edf = (float)((double)edf*((double)1.0f/MI_PI_D));
            else
            {
              const float phongexp = 2.0f/(lsh.roughness*lsh.roughness) - 2.0f;
              const V3 lgn = (ls.flags & s_inside) ? ls.gn : ls.gn;
              edf = (float)((double)edf*((double)(powf(-dot3(lgn, ol), phongexp)*(phongexp + 2.0f))/(2.0f*MI_PI_D)));
            }
            lpdf = lpdf*sc.p_geo;
            edf = edf/sc.p_geo;
            if(edf > 0.0f)
            {
              BsdfEval be;
              if(mat_bsdf == MI_BSDF_DIFFUSE) be = brdf_diffuse(sf, sh, ol);
              else if(mat_bsdf == MI_BSDF_DIELECTRIC) be = brdf_dielectric(sf, sh, omega, ol, eta_ratio);
              else be = brdf_metal(sc, sf, sh, omega, ol, ps.cur_ior, (int)mat_p0, ps.lambda);
              bool okn = be.value > 0.0f;
              Medium nmed = ps.cur;                                        /* volume of the connection edge */
              if(okn && (be.mode & s_transmit))
              { /* path_edge_init_volume on the connection edge */
                Media hyp = ps.media;
                media_apply(hyp, shape, (sf.flags & s_inside) != 0);
                if(hyp.broken) okn = false;
                else if(MEDIA) nmed = shape_interior_medium(sc, media_top_shape(hyp), ps.lambda);
              }
              if(okn)
              { /* prims_get_ray, src/prims.c:390-492 */
                const float eps = 1e-4f*DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)));
                V3 rd = sub3(ls.x, sf.x);
                rd = scale3(rd, mi_rcp(mi_sqrt(dot3(rd, rd))));
                const V3 ro = mk3(sf.x.x + eps*rd.x, sf.x.y + eps*rd.y, sf.x.z + eps*rd.z);
                const V3 dv = mk3(ls.x.x - eps*rd.x - ro.x, ls.x.y - eps*rd.y - ro.y, ls.x.z - eps*rd.z - ro.z);
                const float total_dist = mi_sqrt(dot3(dv, dv));
                if(!(dot3(ls.gn, rd) >= 0) && total_dist > 0.0f)
                {
                  const float Gn = fabsf(dot3(sf.n, ol))*fabsf(dot3(ls.n, ol))/(ldist*ldist);
                  const float nT = MEDIA ? media_transmittance(nmed, ldist) : 1.0f;
                  float tn = ((vthr*be.value)*(nT*edf))*Gn;
                  tn = tn + (vthr*be.value)*((0.0f*Gn)/lpdf);
                  const float wn = lpdf/(lpdf + 0.0f/nT);
                  tn = tn*wn;
                  /* sampler_mis(path, rr*pdf_nee, path_pdf_extend(path, v+1)), ptdl.c:143-146 */
                  float pb;
                  if(mat_bsdf == MI_BSDF_DIFFUSE) pb = (float)(1.0f/MI_PI_D);
                  else if(mat_bsdf == MI_BSDF_DIELECTRIC) pb = pdf_dielectric(sf, sh, omega, ol, eta_ratio, be.mode);
                  else pb = pdf_metal(sf, sh, omega, ol, be.mode);
                  const float pe = ((MEDIA ? media_pdf_to_surface(nmed, ldist) : 1.0f)*pb)*Gn;
                  const double pp = ps.pdfprod;
                  const double our = (double)(1.0f*lpdf)*pp, other = (double)pe*pp;
                  const float wm = (float)our/(float)(other + our);
                  if(tn/1.0f > 0.0f)
                  {
                    ps.sh_pending = 1;
                    ps.prev_x = sf.x; ps.org_eps = eps;                 /* ro = sf.x + eps*rd is formed again when the ray starts */
                    ps.sh_dir = rd; ps.sh_dist = total_dist;
                    ps.sh_light = lpe; ps.ignore = hit.prim;
                    ps.sh_value = (tn/1.0f)*wm;
                    ps.sh_length = ps.length + 1;
                  }
                }
              }
            }
          }
        }
      }
      MI_PHASE(cnt, 4)
      /* next path_extend, src/pathspace.c:167-259 */
      if(alive && ps.length >= (int)sc.max_verts) alive = false;
      if(alive && !(vthr > 0.0f))
      {
        alive = false;
        if(RECORD && v < MI_REC_MAX_VERTS) { rec->v[v].throughput = 0.0f; rec->v[v].mode = s_absorb; }
      }
      if(alive)
      {
        BsdfSample bs;
        get_scrambled_onb(ps.scramble, sf.n, sf.a, sf.b);              /* the vertex's tangent frame, see surface_setup */
        PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v + 1));   /* the vertex the sample leads to */
        if(mat_bsdf == MI_BSDF_DIFFUSE) { MI_BLK(cnt, 5) sample_diffuse(pts, sf, sh, mode, bs); }
        else if(mat_bsdf == MI_BSDF_DIELECTRIC) { MI_BLK(cnt, 6) sample_dielectric(pts, sf, sh, omega, eta_ratio, mode, bs); }
        else { MI_BLK(cnt, 7) sample_metal(sc, pts, sf, sh, omega, ps.cur_ior, (int)mat_p0, ps.lambda, mode, bs); }
        MI_PHASE(cnt, 7)
        /* shader_sample tail, src/shader.c:582-589 */
        bs.omega = normalise3(bs.omega);
        const float dts = ((sf.flags & s_inside) ? -1 : 1)*dot3(sf.gn, bs.omega);
        float weight = bs.weight;
        if(((bs.mode & s_reflect) && (dts < 0.f)) || ((bs.mode & s_transmit) && (dts > 0.f))) weight = 0.0f;
        const float nthr = vthr*weight;
        uint32_t vmode = bs.mode;
        bool ok = !(nthr <= 0.0f);
        if(ok && (vmode & s_transmit))
        { /* path_edge_init_volume for the next edge, src/pathspace.c:127-146 */
          media_apply(ps.media, shape, (sf.flags & s_inside) != 0);
          if(ps.media.broken) ok = false;
          else
          {
            const int top = media_top_shape(ps.media);
            ps.cur_ior = shape_interior_ior(sc, shape_material, top, ps.lambda);
            if(MEDIA) ps.cur = shape_interior_medium(sc, top, ps.lambda);
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
          /* set up the next ray, src/pathspace.c:754-761, src/prims.c:374-388 */
          const float eps = DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)))*1e-4f;
          ps.org = mk3(sf.x.x + eps*bs.omega.x, sf.x.y + eps*bs.omega.y, sf.x.z + eps*bs.omega.z);
          ps.dir = bs.omega;
          ps.ignore = hit.prim;
          ps.prev_x = sf.x; ps.org_eps = eps;
          ps.prev_cos = fabsf(dot3(sf.n, bs.omega));
          ps.prev_throughput = vthr;
          ps.prev_mode = vmode;
          ps.prev_material_modes = material_modes;
          ps.throughput = nthr;
          ps.pdf = bs.pdf;
        }
      }
    }
  }
  if(!alive) { ps.active = 0; cnt.c[4]++; }
}

#endif
