/* mi_path.h -- per-path logic of the pt/ptdl hot path, shared by the two kernel organisations of mi_abi.hip
 * (persistent megakernel with path state in registers; wavefront pipeline with path state in HBM).
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
  V3 org, dir;
  uint32_t ignore;          /* primitive the ray starts on */
  /* vertex v-1 (the one the ray leaves) */
  V3 prev_x;
  float prev_cos;           /* path_lambert(v-1, omega): |n.omega| or 1 */
  float prev_throughput;    /* v[v-1].throughput */
  uint32_t prev_mode;
  /* vertex v being created */
  float throughput;         /* v[v].throughput (after the bsdf sample at v-1) */
  float pdf;                /* v[v].pdf as left by the bsdf sample (projected solid angle) */
  double pdfprod;           /* prod_{k>=1} v[k].pdf, pt.c:30-38 */
  float cur_ior;            /* e[v].vol.ior */
  Media media;
  /* per path */
  float lambda, pixel_i, pixel_j, scramble;
  int length;               /* number of complete vertices */
  Rng rng;
  unsigned long long index;
  uint32_t active;          /* path alive: an extension ray is waiting to be traced */
  /* ptdl: pending shadow ray of the next-event estimate made at the last vertex */
  uint32_t sh_pending;
  uint32_t prev_material_modes;
  V3 sh_org, sh_dir;
  float sh_dist, sh_value;
  uint32_t sh_light, sh_ignore;
  int sh_length;
};

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

/* start path `index`: afterwards ps holds the camera ray as the pending extension ray */
template<bool RECORD, bool HALTON>
__device__ __forceinline__ void path_generate(const DScene &sc, PathState &ps, unsigned long long index, mi_path_record *rec, uint32_t *cnt)
{
  /* path_init + first half of path_extend (length == 0), src/pathspace.c:13-28,210-249 */
  ps.index = index;
  rng_seed(ps.rng, ps.index, sc.frame);
  PointSampler<HALTON> pts(sc, ps.rng, index, 0);
  ps.scramble = 0.1f + rng_next(ps.rng)*(0.9f-0.1f);          /* points_rand, not the point sampler: src/pathspace.c:213 */
  const float lf0 = pts.template camera<MI_DIM_LAMBDA>() + 0/(float)1;
  const float lf = lf0 < 1.0f ? lf0 : fmodf(lf0, 1.0f);     /* fmodf(x, 1) == x for 0 <= x < 1; the libm loop only runs otherwise */
  ps.lambda = 360 + (830 - 360)*lf;
  const float time = pts.template camera<MI_DIM_TIME>()*sc.cam.time_scale;
  if(!HALTON) { (void)rng_next(ps.rng); (void)rng_next(ps.rng); }   /* view_sample_camid twice (one camera): src/pathspace.c:226, src/view.c:846-847 */
  /* camera_sample, src/camera.d/thinlens.c:68-128; everything that does not depend on the random numbers is in sc.cc */
  const mi_camera &cam = sc.cam;
  const DCamConst &cc = sc.cc;
  const float W = cc.W, H = cc.H;
  const float ci = pts.template camera<MI_DIM_IMAGE_X>()*W;
  const float cj = pts.template camera<MI_DIM_IMAGE_Y>()*H;
  const float r1 = pts.template camera<MI_DIM_APERTURE_X>();
  const float r2 = pts.template camera<MI_DIM_APERTURE_Y>();
  const float ang = (float)(2*MI_PI_D*(double)r1);
  float sn, cs;
  sincosf(ang, &sn, &cs);                 /* one range reduction for both (same values as sinf/cosf) */
  const float lu = cs*sqrtf(r2)*cc.lens_radius;
  const float lv = sn*sqrtf(r2)*cc.lens_radius;
  const V3 ca = ld3(cam.a), cb = ld3(cam.b), cn = ld3(cam.n);
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
  const V3 x0 = mk3(cam.pos[0] + aoff.x, cam.pos[1] + aoff.y, cam.pos[2] + aoff.z);
  const float thr0 = sensor*G/cc.pdf_av;
  ps.org = x0; ps.dir = om; ps.ignore = MI_NOPRIM;
  ps.prev_x = x0;
  ps.prev_cos = fabsf(dot3(cn, om));        /* path_lambert on the sensor vertex */
  ps.prev_throughput = thr0;
  ps.prev_mode = s_sensor;
  ps.throughput = thr0;
  ps.pdfprod = 1.0;
  ps.cur_ior = 1.0f;
  ps.media.ids = 0; ps.media.count = 0; ps.media.broken = 0;
  ps.length = 1;
  ps.active = 1;
  ps.prev_material_modes = s_sensor;
  cnt[6]++;                                  /* the sensor vertex */
  if(RECORD)
  {
    rec->index = ps.index; rec->pixel_i = ps.pixel_i; rec->pixel_j = ps.pixel_j; rec->lambda = ps.lambda;
    rec->time = time; rec->scramble = ps.scramble; rec->throughput = 0.0f; rec->length = 1; rec->num_splats = 0;
    Shading z; z.roughness = z.rs = z.rd = z.rg = z.em = 0.0f;
    rec_vertex<RECORD>(rec, 0, MI_PRIMID_INVALID, 0.0f, x0, cn, cn, mk3(0, 0, 0), s_sensor, 0, thr0, 1.0f, 0.0f, 0.0f, z, 0.0f, -1);
  }
}

/* the shadow ray of the pending next-event connection has been traced into `hit` */
template<bool RECORD>
__device__ __forceinline__ void shadow_resolve(const DScene &sc, PathState &ps, const Hit &hit, mi_path_record *rec, uint32_t *cnt, SplatReq &splat)
{
   /* path_visible, src/pathspace.c:311-344: closest hit up to the emitter's primitive (all surfaces in scope are opaque) */
  ps.sh_pending = 0;
  const bool visible = (hit.dist >= ps.sh_dist) || (hit.prim == MI_NOPRIM) || (hit.prim == ps.sh_light);
  if(visible)
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
      cnt[5]++;
      if(!RECORD) { splat.pending = true; splat.c0 = col[0]; splat.c1 = col[1]; splat.c2 = col[2]; }
    }
  }
}

/* the extension ray ps.org/ps.dir has been traced into `hit`: create vertex v = ps.length, then either end the path
 * (ps.active = 0) or leave the next extension ray (and, for ptdl, possibly a shadow ray) in ps */
template<bool RECORD, bool PTDL, bool HALTON>
__device__ __forceinline__ void path_shade(const DScene &sc, PathState &ps, const Hit &hit, const uint32_t *shape_material, const float *shape_L,
                                           mi_path_record *rec, uint32_t *cnt, SplatReq &splat)
{

  const int v = ps.length;                         /* index of the vertex being created */
  bool alive = true;
  const V3 omega = ps.dir;
  if(hit.prim == MI_NOPRIM)
  { /* left the scene: environment vertex, src/pathspace.c:856-873; black sky => nothing to add, path ends */
    const float G = ps.prev_cos;                   /* path_G with an environment end point */
    const float vpdf = ps.pdf*G;
    ps.pdfprod *= (double)vpdf;
    ps.length++;
    cnt[6]++;
    if(RECORD)
    {
      const V3 x = mk3(ps.prev_x.x + sc.far_dist*omega.x, ps.prev_x.y + sc.far_dist*omega.y, ps.prev_x.z + sc.far_dist*omega.z);
      Shading z; z.roughness = 1.0f; z.rs = z.rd = z.rg = z.em = 0.0f;
      rec_vertex<RECORD>(rec, v, MI_PRIMID_INVALID, FLT_MAX, x, mk3(0, 0, 0), mk3(0, 0, 0), omega, s_absorb, s_environment,
                         ps.throughput, vpdf, 0.0f, 0.0f, z, 0.0f, -1);
      rec->length = ps.length; rec->throughput = 0.0f;
    }
    alive = false;
  }
  else
  {
    /* shader_prepare, src/shader.c:462-542 */
    Surf sf;
    sf.x = mk3(ps.org.x + hit.dist*ps.dir.x, ps.org.y + hit.dist*ps.dir.y, ps.org.z + hit.dist*ps.dir.z);
    sf.u = hit.u; sf.v = hit.v;
    /* the record's header first (one 16-B load): it names the material, whose fetch is then under way while the
       surface is set up */
    const DPrimGeo &pshade = sc.primgeo[hit.prim];
    const uint4 head = *(const uint4 *)&pshade;                     /* type, material, uv0, primid_lo */
    const DMaterial &mat = sc.materials[head.y];
    const uint4 mhead = *(const uint4 *)&mat;                       /* bsdf, num_ops, param[0..1] */
    const uint32_t mat_bsdf = mhead.x;
    const float mat_p0 = __uint_as_float(mhead.z), mat_p1 = __uint_as_float(mhead.w);
    surface_setup(sc, hit.prim, head, omega, ps.scramble, sf);
    MI_PHASE(cnt, 2)
    const uint32_t shape = (head.w >> 3) & 0x1fffffffu;             /* MI_PRIMID_SHAPE */
    Shading sh;
    run_prepare_ops(sc, mat, mhead.y, sf, ps.lambda, sh);
    uint32_t material_modes = 0;
    float eta_ratio = 1.0f;      /* path_eta_ratio(v): e[v].vol.ior / ior behind the interface, src/pathspace.c:117-124 */
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
    const uint32_t type = head.x;
    if((type > 2 || hit.dist < 1e-4f) && hit.prim == ps.ignore)
    {
      alive = false;
      if(RECORD) { rec->length = ps.length; rec->throughput = 0.0f; }
    }
    else
    {
      uint32_t mode = s_absorb;
      if(sh.em > 0.0f && !(sf.flags & s_inside)) { mode = s_emit; material_modes = s_emit; }
      /* path_extend tail, src/pathspace.c:261-270 */
      const float G = ps.prev_cos*fabsf(dot3(sf.n, omega))/(hit.dist*hit.dist);
      const float vpdf = ps.pdf*G;
      const double pp_before = ps.pdfprod;
      ps.pdfprod *= (double)vpdf;
      ps.length++;
      cnt[6]++;
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
        path_throughput = 0.0f + ps.throughput*Le;
      }
      float vthr = ps.throughput;
      if(RECORD)
      {
        rec_vertex<RECORD>(rec, v, MI_GEO_PRIMID(pshade), hit.dist, sf.x, sf.n, sf.gn, omega, mode, sf.flags, vthr, vpdf, sf.u, sf.v, sh,
                           eta_ratio, (int)head.y);
        rec->length = ps.length; rec->throughput = path_throughput;
      }
      if(mode & s_emit)
      { /* sampler_create_path: src/sampler.d/pt.c:45-52 / src/sampler.d/ptdl.c:116-121 */
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
          cnt[5]++;
          if(!RECORD) { splat.pending = true; splat.c0 = col[0]; splat.c1 = col[1]; splat.c2 = col[2]; }
        }
        if(!PTDL && ps.length > 3)
        { /* path_russian_roulette, src/pathspace.c:273-292 */
          const float p_survival = DMIN(1.0f, vthr/ps.prev_throughput);
          PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v));   /* the last vertex's dimensions, src/pathspace.c:278-281 */
          const float rr = pts(MI_DIM_RUSSIAN_R);
          if(rr >= p_survival) { vthr = vthr*(1.0f/(1.0f-p_survival)); alive = false; }
          else vthr = vthr*(1.0f/p_survival);
          if(RECORD && v < MI_REC_MAX_VERTS)
          {
            rec->v[v].throughput = vthr;
            rec->v[v].pdf = alive ? vpdf*p_survival : vpdf*(1.0f-p_survival);
          }
        }
      }
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
            const float r3 = pts(MI_DIM_NEE_Y);
            const float r2 = pts(MI_DIM_NEE_X);
            const float r1 = pts(MI_DIM_NEE_LIGHT2);
            const uint32_t t = sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
            const uint32_t lp = sc.light_prim[t];
            Surf ls;
            ls.x = prim_sample(sc.prims[lp], sc.primgeo[lp], r2, r3, ls.u, ls.v);
            V3 ol = sub3(ls.x, sf.x);
            const float ldist = sqrtf(dot3(ol, ol));
            const double il = 1./(double)ldist;
            ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
            const uint4 lhead = *(const uint4 *)&sc.primgeo[lp];
            surface_setup(sc, lp, lhead, ol, ps.scramble, ls);
            Shading lsh;
            run_prepare_ops(sc, sc.materials[lhead.y], sc.materials[lhead.y].num_ops, ls, ps.lambda, lsh);
            float lpdf = sc.light_L[t];
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
              if(okn && (be.mode & s_transmit))
              { /* path_edge_init_volume on the connection edge */
                Media hyp = ps.media;
                media_apply(hyp, shape, (sf.flags & s_inside) != 0);
                if(hyp.broken) okn = false;
              }
              if(okn)
              { /* prims_get_ray, src/prims.c:390-492 */
                const float eps = 1e-4f*DMAX(DMAX(.5f, fabsf(sf.x.x)), DMAX(fabsf(sf.x.y), fabsf(sf.x.z)));
                V3 rd = sub3(ls.x, sf.x);
                rd = scale3(rd, 1.0f/sqrtf(dot3(rd, rd)));
                const V3 ro = mk3(sf.x.x + eps*rd.x, sf.x.y + eps*rd.y, sf.x.z + eps*rd.z);
                const V3 dv = mk3(ls.x.x - eps*rd.x - ro.x, ls.x.y - eps*rd.y - ro.y, ls.x.z - eps*rd.z - ro.z);
                const float total_dist = sqrtf(dot3(dv, dv));
                if(!(dot3(ls.gn, rd) >= 0) && total_dist > 0.0f)
                {
                  const float Gn = fabsf(dot3(sf.n, ol))*fabsf(dot3(ls.n, ol))/(ldist*ldist);
                  float tn = ((vthr*be.value)*(1.0f*edf))*Gn;
                  tn = tn + (vthr*be.value)*((0.0f*Gn)/lpdf);
                  const float wn = lpdf/(lpdf + 0.0f/1.0f);
                  tn = tn*wn;
                  /* sampler_mis(path, rr*pdf_nee, path_pdf_extend(path, v+1)), ptdl.c:143-146 */
                  float pb;
                  if(mat_bsdf == MI_BSDF_DIFFUSE) pb = (float)(1.0f/MI_PI_D);
                  else if(mat_bsdf == MI_BSDF_DIELECTRIC) pb = pdf_dielectric(sf, sh, omega, ol, eta_ratio, be.mode);
                  else pb = pdf_metal(sf, sh, omega, ol, be.mode);
                  const float pe = (1.0f*pb)*Gn;
                  const double pp = ps.pdfprod;
                  const double our = (double)(1.0f*lpdf)*pp, other = (double)pe*pp;
                  const float wm = (float)our/(float)(other + our);
                  if(tn/1.0f > 0.0f)
                  {
                    ps.sh_pending = 1;
                    ps.sh_org = ro; ps.sh_dir = rd; ps.sh_dist = total_dist;
                    ps.sh_light = lp; ps.sh_ignore = hit.prim;
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
        PointSampler<HALTON> pts(sc, ps.rng, ps.index, rand_beg_extend<PTDL>(v + 1));   /* the vertex the sample leads to */
        if(mat_bsdf == MI_BSDF_DIFFUSE) sample_diffuse(pts, sf, sh, mode, bs);
        else if(mat_bsdf == MI_BSDF_DIELECTRIC) sample_dielectric(pts, sf, sh, omega, eta_ratio, mode, bs);
        else sample_metal(sc, pts, sf, sh, omega, ps.cur_ior, (int)mat_p0, ps.lambda, mode, bs);
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
          ps.prev_x = sf.x;
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
  if(!alive) { ps.active = 0; cnt[4]++; }
}

#endif
