/* mi_abi.hip -- the persistent path tracing kernel and the C ABI (include/corona_mi.h) of
 * libcorona_mi.so for gfx950. Device helpers live in mi_kernels.h, the device layout in mi_device.h.
 *
 * Replaces everything the reference reaches from work_sample() (src/view.c:618-628): one launch
 * traces path indices [first, first+count) and splats them into the device framebuffer.
 */
#include "mi_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#ifndef MI_BLOCK
#define MI_BLOCK 1024    /* threads per workgroup: 16 waves/CU = 4 per SIMD (128 VGPRs each) */
#endif
#ifndef MI_STACK
#define MI_STACK 12      /* LDS traversal stack entries per lane; deeper entries overflow to HBM (mi_device.h) */
#endif

/* ======================================================================================= kernel */
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

template<bool RECORD, bool PTDL>
__global__ __launch_bounds__(MI_BLOCK) void mi_path_kernel(DScene sc, unsigned long long first, unsigned long long count,
                                                           const uint32_t *shape_material, const float *shape_L, mi_path_record *records,
                                                           uint2 *stack_overflow)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t N = sc.num_nodes;
  float4 *lds_nodes = (float4 *)smem;
  uint32_t *lds_axes = (uint32_t *)(smem + (size_t)MI_NODE_FIELDS*N*16);
  const size_t stack_off = (((size_t)MI_NODE_FIELDS*N*16 + (size_t)N*4) + 15) & ~(size_t)15;
  uint2 *lds_stack = (uint2 *)(smem + stack_off);

  /* stage the BVH into LDS once per workgroup (coalesced 16-B loads) */
  for(uint32_t i=threadIdx.x;i<MI_NODE_FIELDS*N;i+=MI_BLOCK) lds_nodes[i] = sc.nodes[i];
  for(uint32_t i=threadIdx.x;i<N;i+=MI_BLOCK) lds_axes[i] = sc.node_axes[i];
  __syncthreads();

  Lds lds;
  lds.nodes = lds_nodes; lds.axes = lds_axes; lds.stack = lds_stack + threadIdx.x; lds.num_nodes = N;
  lds.overflow_stride = gridDim.x*MI_BLOCK;
  lds.overflow = stack_overflow + (size_t)blockIdx.x*MI_BLOCK + threadIdx.x;

  uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PathState ps;
  ps.active = 0;
  ps.sh_pending = 0;
  bool exhausted = false;
  const unsigned lane = __lane_id();

  while(true)
  {
    /* ------------------------------------------------------------ refill idle lanes (wave-level compaction of the work queue) */
    if(!exhausted)
    {
      const bool want = !ps.active && !ps.sh_pending;
      const unsigned long long m = __ballot(want);
      if(m)
      {
        const unsigned n = __popcll(m);
        unsigned long long base = 0;
        if(lane == (unsigned)(__ffsll((long long)m) - 1)) base = atomicAdd(sc.work, (unsigned long long)n);
        base = __shfl(base, __ffsll((long long)m) - 1);
        if(want)
        {
          const unsigned rank = __popcll(m & ((1ull << lane) - 1ull));
          const unsigned long long i = base + rank;
          if(i < count)
          {
            /* path_init + first half of path_extend (length == 0), src/pathspace.c:13-28,210-249 */
            ps.index = first + i;
            rng_seed(ps.rng, ps.index, sc.frame);
            ps.scramble = 0.1f + rng_next(ps.rng)*(0.9f-0.1f);
            const float lf = fmodf(rng_next(ps.rng) + 0/(float)1, 1.0f);
            ps.lambda = 360 + (830 - 360)*lf;
            const float time = rng_next(ps.rng)*sc.cam.time_scale;
            (void)rng_next(ps.rng);
            (void)rng_next(ps.rng);
            /* camera_sample, src/camera.d/thinlens.c:68-128 */
            const mi_camera &cam = sc.cam;
            const float W = (float)sc.width, H = (float)sc.height;
            const float ci = rng_next(ps.rng)*W;
            const float cj = rng_next(ps.rng)*H;
            const float r1 = rng_next(ps.rng);
            const float r2 = rng_next(ps.rng);
            const float lens_radius = (.5f/cam.f_stop)*cam.focal_length;
            const float ang = (float)(2*MI_PI_D*(double)r1);
            const float lu = cosf(ang)*sqrtf(r2)*lens_radius;
            const float lv = sinf(ang)*sqrtf(r2)*lens_radius;
            const V3 ca = ld3(cam.a), cb = ld3(cam.b), cn = ld3(cam.n);
            const float f = cam.focus/cam.focal_length;
            const float f_dir = cam.focus;
            const float f_rg = -cam.film_width*f/W;
            const float f_up = -cam.film_height*f/H;
            const V3 aoff = mk3(lu*ca.x + lv*cb.x, lu*ca.y + lv*cb.y, lu*ca.z + lv*cb.z);
            const float ki = (ci-.5f*W)*f_rg, kj = (cj-.5f*H)*f_up;
            V3 om = mk3(f_dir*cn.x + (ki*ca.x + kj*cb.x) - aoff.x,
                        f_dir*cn.y + (ki*ca.y + kj*cb.y) - aoff.y,
                        f_dir*cn.z + (ki*ca.z + kj*cb.z) - aoff.z);
            om = normalise3(om);
            const float A = (float)(MI_PI_D*(double)cam.focal_length*(double)cam.focal_length/(double)(4.0f*cam.f_stop*cam.f_stop));
            const float pdf_a = (float)(1./(double)A);
            const float sensor = 106.86535f*100.0f*cam.exposure_time;
            const float dt = dot3(om, cn);
            const float dot4 = dt*dt*dt*dt;
            ps.pixel_i = (float)DCLAMP((double)ci, 0.0, (double)(W-1e-4f));
            ps.pixel_j = (float)DCLAMP((double)cj, 0.0, (double)(H-1e-4f));
            const float G = dot4/(cam.focal_length*cam.focal_length);
            const float pdf_v = 1.0f/(cam.film_width*cam.film_height);
            ps.pdf = pdf_v*pdf_a/G;
            const V3 x0 = mk3(cam.pos[0] + aoff.x, cam.pos[1] + aoff.y, cam.pos[2] + aoff.z);
            const float thr0 = sensor*G/(pdf_a*pdf_v);
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
              mi_path_record *rec = records + i;
              rec->index = ps.index; rec->pixel_i = ps.pixel_i; rec->pixel_j = ps.pixel_j; rec->lambda = ps.lambda;
              rec->time = time; rec->scramble = ps.scramble; rec->throughput = 0.0f; rec->length = 1; rec->num_splats = 0;
              Shading z; z.roughness = z.rs = z.rd = z.rg = z.em = 0.0f;
              rec_vertex<RECORD>(rec, 0, MI_PRIMID_INVALID, 0.0f, x0, cn, cn, mk3(0, 0, 0), s_sensor, 0, thr0, 1.0f, 0.0f, 0.0f, z, 0.0f, -1);
            }
          }
          else exhausted = true;
        }
      }
    }
    if(!__any(ps.active || ps.sh_pending)) break;

    /* ------------------------------------------------------------ one ray per busy lane: a pending shadow ray first, else the extension ray */
    bool splat_req = false;
    float splat_c0 = 0.0f, splat_c1 = 0.0f, splat_c2 = 0.0f;
    const bool do_shadow = PTDL && ps.sh_pending;
    Hit hit;
    hit.prim = MI_NOPRIM; hit.dist = do_shadow ? ps.sh_dist : FLT_MAX; hit.u = hit.v = 0.0f;
    if(do_shadow || ps.active)
      accel_intersect<MI_BLOCK, MI_STACK>(lds, sc.prims, do_shadow ? ps.sh_org : ps.org, do_shadow ? ps.sh_dir : ps.dir,
                                          do_shadow ? ps.sh_ignore : ps.ignore, hit, cnt);

    if(do_shadow)
    { /* path_visible, src/pathspace.c:311-344: closest hit up to the emitter's primitive (all surfaces in scope are opaque) */
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
          mi_path_record *rec = records + (ps.index - first);
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
          if(!RECORD) { splat_req = true; splat_c0 = col[0]; splat_c1 = col[1]; splat_c2 = col[2]; }
        }
      }
    }
    /* ------------------------------------------------------------ finish vertex v, then sample the next direction */
    else if(ps.active)
    {
      mi_path_record *rec = RECORD ? records + (ps.index - first) : nullptr;
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
        surface_setup(sc, hit.prim, omega, ps.scramble, sf);
        const DPrimShade &pshade = sc.primshade[hit.prim];
        const DMaterial &mat = sc.materials[pshade.material];
        const uint32_t shape = MI_PRIMID_SHAPE(pshade.primid);
        Shading sh;
        run_prepare_ops(sc, mat, sf, ps.lambda, sh);
        uint32_t material_modes = 0;
        float eta_ratio = 1.0f;      /* path_eta_ratio(v): e[v].vol.ior / ior behind the interface, src/pathspace.c:117-124 */
        {
          Media hyp = ps.media;
          media_apply(hyp, shape, (sf.flags & s_inside) != 0);
          float interior_self = 1.0f;
          if(mat.bsdf == MI_BSDF_DIELECTRIC) interior_self = eta_from_abbe(mat.param[0], mat.param[1], ps.lambda);
          if(hyp.broken) eta_ratio = -1.0f;
          else
          {
            const int top = media_top_shape(hyp);
            const float ior2 = (top == (int)shape) ? interior_self : shape_interior_ior(sc, shape_material, top, ps.lambda);
            eta_ratio = ps.cur_ior/ior2;
          }
        }
        if(mat.bsdf == MI_BSDF_DIFFUSE) { if(sh.rd > 0.0f) material_modes = s_reflect | s_diffuse; }
        else if(mat.bsdf == MI_BSDF_DIELECTRIC)
        {
          material_modes = s_reflect | s_transmit;
          if(fabsf(1.0f - eta_ratio/1.0f) < 1e-3f) sh.roughness = 0.0f;
          if(sh.roughness > GLOSSY_THR) material_modes |= s_glossy; else material_modes |= s_specular;
        }
        else if(mat.bsdf == MI_BSDF_METAL)
        {
          material_modes = s_reflect;
          if(sh.roughness > 1e-4f) material_modes |= s_glossy; else material_modes |= s_specular;
        }

        /* self-intersection, src/pathspace.c:807-820 */
        const uint32_t type = sc.prims[hit.prim].type;
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
            rec_vertex<RECORD>(rec, v, pshade.primid, hit.dist, sf.x, sf.n, sf.gn, omega, mode, sf.flags, vthr, vpdf, sf.u, sf.v, sh,
                               eta_ratio, (int)pshade.material);
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
              if(!RECORD) { splat_req = true; splat_c0 = col[0]; splat_c1 = col[1]; splat_c2 = col[2]; }
            }
            if(!PTDL && ps.length > 3)
            { /* path_russian_roulette, src/pathspace.c:273-292 */
              const float p_survival = DMIN(1.0f, vthr/ps.prev_throughput);
              const float rr = rng_next(ps.rng);
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
              const float rnd = rng_next(ps.rng);
              if(!(rnd < sc.p_sky) && rnd < sc.p_sky + sc.p_geo)
              { /* lights_sample_next_event, src/lights.d/list.c:130-174 (arguments drawn right to left) */
                const float r3 = rng_next(ps.rng);
                const float r2 = rng_next(ps.rng);
                const float r1 = rng_next(ps.rng);
                const uint32_t t = sample_cdf(sc.light_cdf, (int)sc.num_lights, r1);
                const uint32_t lp = sc.light_prim[t];
                Surf ls;
                ls.x = prim_sample(sc.prims[lp], r2, r3, ls.u, ls.v);
                V3 ol = sub3(ls.x, sf.x);
                const float ldist = sqrtf(dot3(ol, ol));
                const double il = 1./(double)ldist;
                ol = mk3((float)((double)ol.x*il), (float)((double)ol.y*il), (float)((double)ol.z*il));
                surface_setup(sc, lp, ol, ps.scramble, ls);
                const DPrimShade &lshade = sc.primshade[lp];
                Shading lsh;
                run_prepare_ops(sc, sc.materials[lshade.material], ls, ps.lambda, lsh);
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
                  if(mat.bsdf == MI_BSDF_DIFFUSE) be = brdf_diffuse(sf, sh, ol);
                  else if(mat.bsdf == MI_BSDF_DIELECTRIC) be = brdf_dielectric(sf, sh, omega, ol, eta_ratio);
                  else be = brdf_metal(sc, sf, sh, omega, ol, ps.cur_ior, (int)mat.param[0], ps.lambda);
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
                      if(mat.bsdf == MI_BSDF_DIFFUSE) pb = (float)(1.0f/MI_PI_D);
                      else if(mat.bsdf == MI_BSDF_DIELECTRIC) pb = pdf_dielectric(sf, sh, omega, ol, eta_ratio, be.mode);
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
            if(mat.bsdf == MI_BSDF_DIFFUSE) sample_diffuse(ps.rng, sf, sh, mode, bs);
            else if(mat.bsdf == MI_BSDF_DIELECTRIC) sample_dielectric(ps.rng, sf, sh, omega, eta_ratio, mode, bs);
            else sample_metal(sc, ps.rng, sf, sh, omega, ps.cur_ior, (int)mat.param[0], ps.lambda, mode, bs);
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
    /* ------------------------------------------------------------ splats of this iteration, cooperatively */
    if(!RECORD) splat_wave(sc, splat_req, ps.pixel_i, ps.pixel_j, splat_c0, splat_c1, splat_c2);
  }

  atomicMax(sc.counters + 7, (unsigned long long)cnt[7]);     /* deepest traversal stack use */
  /* ------------------------------------------------------------ flush work counters: wave reduction, one atomic per wave */
#pragma unroll
  for(int k=0;k<8;k++)
  {
    unsigned long long c = cnt[k];
    for(int off=32;off>0;off>>=1) c += __shfl_down(c, off);
    if(k == 7) continue;                    /* slot 7 is a maximum, flushed above */
    if(lane == 0 && c) atomicAdd(sc.counters + k, c);
  }
}

/* ======================================================================================= host side */
static thread_local char g_err[512] = "";
static int g_device = -1;

#define HIPCHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { \
  snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; } } while(0)

static int fail(int code, const char *msg)
{
  snprintf(g_err, sizeof(g_err), "%s", msg);
  fprintf(stderr, "[mi] %s\n", msg);
  return code;
}

struct mi_scene
{
  DScene d;
  uint32_t width, height;
  void *d_nodes, *d_axes, *d_prims, *d_primshade, *d_materials, *d_light_prim, *d_light_cdf, *d_light_L;
  void *d_cie, *d_checker, *d_metal, *d_counters, *d_work, *d_shape_material, *d_shape_L, *d_overflow;
  float *d_fb_own, *d_fb;
  hipStream_t stream_own, stream;
  hipEvent_t ev0, ev1;
  int have_timing;
  size_t lds_bytes;
  int grid;
  uint64_t launches;
};

extern "C" const char *mi_last_error(void) { return g_err; }

extern "C" int mi_init(int device)
{
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MI_ERR_DEVICE, "no HIP device visible");
  if(device < 0)
  {
    const char *lr = getenv("LOCAL_RANK");
    device = lr ? atoi(lr) : 0;
  }
  if(device >= n) device = device % n;
  HIPCHK(hipSetDevice(device));
  g_device = device;
  return MI_OK;
}

template<typename T> static int upload(void **dst, const T *src, size_t count)
{
  *dst = nullptr;
  if(!count) count = 1;
  HIPCHK(hipMalloc(dst, count*sizeof(T)));
  if(src) HIPCHK(hipMemcpy(*dst, src, count*sizeof(T), hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(*dst, 0, count*sizeof(T)));
  return MI_OK;
}

static int tree_depth(const mi_scene_desc *h, uint32_t node, int depth)
{
  int best = depth;
  for(int c=0;c<4;c++)
  {
    const uint64_t ch = h->nodes[node].child[c];
    if(!(ch & MI_NODE_LEAF))
    {
      if(ch >= h->num_nodes || depth > 200) return 1000;
      const int d = tree_depth(h, (uint32_t)ch, depth+1);
      if(d > best) best = d;
    }
  }
  return best;
}

extern "C" int mi_scene_create(const mi_scene_desc *h, mi_scene **out)
{
  if(!h || !out) return fail(MI_ERR_ARG, "mi_scene_create: null argument");
  if(h->struct_size != sizeof(mi_scene_desc) || h->abi_version != MI_ABI_VERSION)
    return fail(MI_ERR_ARG, "mi_scene_create: mi_scene_desc size/version mismatch");
  if(g_device < 0) { const int e = mi_init(-1); if(e) return e; }
  if(!h->width || !h->height || (h->width & 31) || (h->height & 31)) return fail(MI_ERR_ARG, "film size must be a non-zero multiple of 32");
  if(h->max_verts < 2 || h->max_verts > 32) return fail(MI_ERR_ARG, "max_verts must be in [2,32]");
  if(h->sampler != MI_SAMPLER_PT && h->sampler != MI_SAMPLER_PTDL) return fail(MI_ERR_ARG, "unknown sampler");
  if(!h->num_nodes || !h->nodes || !h->cie_xyz) return fail(MI_ERR_ARG, "scene has no nodes / tables");
  if(h->num_shapes > 255) return fail(MI_ERR_UNSUPPORTED, "more than 255 shapes");
  if(h->num_prims >= (1u << 26)) return fail(MI_ERR_UNSUPPORTED, "more than 2^26 primitives");

  const int depth = tree_depth(h, 0, 0);
  if(depth > 200) return fail(MI_ERR_ARG, "BVH deeper than 200 levels or cyclic");
  const int stack_need = 3*(depth+1);                       /* at most 3 pushes per inner node on the way down */

  mi_scene *s = (mi_scene *)calloc(1, sizeof(mi_scene));
  if(!s) return fail(MI_ERR_NOMEM, "out of host memory");
  s->width = h->width; s->height = h->height;
  DScene &d = s->d;
  d.width = h->width; d.height = h->height; d.max_verts = h->max_verts; d.sampler = h->sampler; d.frame = h->frame;
  d.num_nodes = h->num_nodes; d.num_prims = (uint32_t)h->num_prims;
  memcpy(d.aabb, h->aabb, sizeof(d.aabb));
  {
    const float ex = h->aabb[3]-h->aabb[0], ey = h->aabb[4]-h->aabb[1], ez = h->aabb[5]-h->aabb[2];
    const float m1 = ey > ex ? ey : ex;
    d.far_dist = 2.0f*(ez > m1 ? ez : m1);
  }

  /* nodes: SoA of 16-byte lanes, children as 32-bit links */
  const uint32_t N = h->num_nodes;
  std::vector<float> nodes((size_t)MI_NODE_FIELDS*N*4);
  std::vector<uint32_t> axes(N);
  for(uint32_t n=0;n<N;n++)
  {
    const mi_node &nd = h->nodes[n];
    for(int k=0;k<6;k++) for(int c=0;c<4;c++) nodes[((size_t)k*N + n)*4 + c] = nd.aabb[k][c];
    for(int c=0;c<4;c++)
    {
      uint32_t link;
      if(nd.child[c] & MI_NODE_LEAF)
      {
        const uint64_t first = (nd.child[c] ^ MI_NODE_LEAF) >> 5, cntp = nd.child[c] & 31;
        link = MI_LEAF32 | (uint32_t)(first << 5) | (uint32_t)cntp;
      }
      else link = (uint32_t)nd.child[c];
      memcpy(&nodes[((size_t)6*N + n)*4 + c], &link, 4);
    }
    axes[n] = (uint32_t)(nd.axis0 & 3) | ((uint32_t)(nd.axis00 & 3) << 2) | ((uint32_t)(nd.axis01 & 3) << 4);
  }
  /* primitives: resolve primid -> vtxidx -> vtx once */
  std::vector<DPrim> prims(h->num_prims ? h->num_prims : 1);
  std::vector<DPrimShade> pshade(h->num_prims ? h->num_prims : 1);
  memset(prims.data(), 0, prims.size()*sizeof(DPrim));
  memset(pshade.data(), 0, pshade.size()*sizeof(DPrimShade));
  for(uint64_t i=0;i<h->num_prims;i++)
  {
    const mi_primid pi = h->primid[i];
    const uint32_t shape = MI_PRIMID_SHAPE(pi), vc = MI_PRIMID_VCNT(pi);
    if(shape >= h->num_shapes || vc < 1 || vc > 4 || MI_PRIMID_MB(pi)) { free(s); return fail(MI_ERR_UNSUPPORTED, "primitive kind outside the scope"); }
    const mi_shape &sh = h->shapes[shape];
    const mi_vtxidx *vi = h->vtxidx + sh.vtxidx_base + MI_PRIMID_VI(pi);
    const mi_vtx *vtx = h->vtx + sh.vtx_base;
    DPrim &p = prims[i]; DPrimShade &q = pshade[i];
    p.type = vc;
    q.primid = pi; q.material = (uint32_t)sh.material;
    if((uint32_t)sh.material >= h->num_materials || h->materials[sh.material].bsdf > MI_BSDF_METAL)
    { free(s); return fail(MI_ERR_UNSUPPORTED, "shape uses a material outside the scope"); }
    for(uint32_t k=0;k<vc;k++) { q.n[k] = vtx[vi[k].v].n; q.uv[k] = vi[k].uv; }
    if(vc == MI_PRIM_SPHERE)
    {
      memcpy(p.v[0], vtx[vi[0].v].v, 12);
      memcpy(&p.v[1][0], &vtx[vi[0].v].n, 4);
    }
    else if(vc == MI_PRIM_LINE)
    {
      memcpy(p.v[0], vtx[vi[0].v].v, 12);
      memcpy(p.v[1], vtx[vi[1].v].v, 12);
      memcpy(&p.v[2][0], &vtx[vi[0].v].n, 4);
      memcpy(&p.v[2][1], &vtx[vi[1].v].n, 4);
    }
    else for(uint32_t k=0;k<vc;k++) memcpy(p.v[k], vtx[vi[k].v].v, 12);
  }
  std::vector<DMaterial> mats(h->num_materials ? h->num_materials : 1);
  for(uint32_t i=0;i<h->num_materials;i++)
  {
    mats[i].bsdf = h->materials[i].bsdf; mats[i].num_ops = h->materials[i].num_ops;
    memcpy(mats[i].op, h->materials[i].op, sizeof(mats[i].op));
    memcpy(mats[i].param, h->materials[i].param, sizeof(mats[i].param));
  }
  std::vector<uint32_t> shape_mat(h->num_shapes ? h->num_shapes : 1);
  for(uint32_t i=0;i<h->num_shapes;i++) shape_mat[i] = (uint32_t)h->shapes[i].material;
  std::vector<float> shape_L(h->num_shapes ? h->num_shapes : 1, 0.0f);
  for(uint32_t k=0;k<h->lights.num_prims;k++)
  {
    const uint32_t sid = MI_PRIMID_SHAPE(h->lights.primid[k]);
    if(sid < h->num_shapes && shape_L[sid] == 0.0f) shape_L[sid] = h->lights.L[k];   /* lights_pdf_next_event: L of the shape */
  }
  /* emitters: original primid -> builder-order index */
  std::vector<uint32_t> lprim(h->lights.num_prims ? h->lights.num_prims : 1);
  for(uint32_t k=0;k<h->lights.num_prims;k++)
  {
    uint32_t found = MI_NOPRIM;
    for(uint64_t i=0;i<h->num_prims;i++) if(h->primid[i] == h->lights.primid[k]) { found = (uint32_t)i; break; }
    if(found == MI_NOPRIM) { free(s); return fail(MI_ERR_ARG, "emitter primitive not in the primitive list"); }
    lprim[k] = found;
  }

  int e = MI_OK;
#define UP(dst, src, cnt) if(!e) e = upload(&s->dst, src, cnt)
  UP(d_nodes, nodes.data(), nodes.size());
  UP(d_axes, axes.data(), axes.size());
  UP(d_prims, prims.data(), prims.size());
  UP(d_primshade, pshade.data(), pshade.size());
  UP(d_materials, mats.data(), mats.size());
  UP(d_shape_material, shape_mat.data(), shape_mat.size());
  UP(d_shape_L, shape_L.data(), shape_L.size());
  UP(d_light_prim, lprim.data(), lprim.size());
  UP(d_light_cdf, h->lights.cdf, (size_t)h->lights.num_prims);
  UP(d_light_L, h->lights.L, (size_t)h->lights.num_prims);
  UP(d_cie, h->cie_xyz, (size_t)96*3);
  UP(d_checker, h->checker, h->checker ? (size_t)140*36 : 0);
  UP(d_metal, h->metal_ior, h->metal_ior ? (size_t)5*95*2 : 0);
  UP(d_counters, (const unsigned long long *)nullptr, (size_t)8);
  UP(d_work, (const unsigned long long *)nullptr, (size_t)1);
#undef UP
  if(!e)
  {
    void *fb = nullptr;
    if(hipMalloc(&fb, sizeof(float)*3*(size_t)h->width*h->height) != hipSuccess) e = fail(MI_ERR_NOMEM, "cannot allocate the device framebuffer");
    else { s->d_fb_own = (float *)fb; hipMemset(fb, 0, sizeof(float)*3*(size_t)h->width*h->height); }
  }
  if(e) { mi_scene_destroy(s); return e; }
  s->d_fb = s->d_fb_own;
  if(hipStreamCreateWithFlags(&s->stream_own, hipStreamNonBlocking) != hipSuccess ||
     hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess)
  { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "cannot create stream/events"); }
  s->stream = s->stream_own;

  d.nodes = (const float4 *)s->d_nodes; d.node_axes = (const uint32_t *)s->d_axes;
  d.prims = (const DPrim *)s->d_prims; d.primshade = (const DPrimShade *)s->d_primshade;
  d.materials = (const DMaterial *)s->d_materials;
  d.num_lights = h->lights.num_prims;
  d.light_prim = (const uint32_t *)s->d_light_prim; d.light_cdf = (const float *)s->d_light_cdf; d.light_L = (const float *)s->d_light_L;
  d.p_sky = h->lights.p_sky; d.p_geo = h->lights.p_geo; d.p_vol = h->lights.p_vol;
  d.cam = h->cam;
  d.cie_xyz = (const float *)s->d_cie; d.checker = (const float *)s->d_checker; d.metal_ior = (const float *)s->d_metal;
  d.fb = s->d_fb;
  d.counters = (unsigned long long *)s->d_counters;
  d.work = (unsigned long long *)s->d_work;

  const size_t node_bytes = (((size_t)MI_NODE_FIELDS*N*16 + (size_t)N*4) + 15) & ~(size_t)15;
  s->lds_bytes = node_bytes + (size_t)MI_STACK*MI_BLOCK*sizeof(uint2);
  if(s->lds_bytes > 160*1024) { mi_scene_destroy(s); return fail(MI_ERR_UNSUPPORTED, "BVH does not fit the LDS-resident traversal of this build"); }
  if(hipFuncSetAttribute((const void *)mi_path_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes) != hipSuccess ||
     hipFuncSetAttribute((const void *)mi_path_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes) != hipSuccess ||
     hipFuncSetAttribute((const void *)mi_path_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes) != hipSuccess ||
     hipFuncSetAttribute((const void *)mi_path_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes) != hipSuccess)
  { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "cannot raise the dynamic LDS limit"); }
  hipDeviceProp_t prop;
  if(hipGetDeviceProperties(&prop, g_device) != hipSuccess) { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "hipGetDeviceProperties failed"); }
  int per_cu = (int)((160*1024)/s->lds_bytes);
  if(per_cu < 1) per_cu = 1;
  if(per_cu*MI_BLOCK > 2048) per_cu = 2048/MI_BLOCK;
  s->grid = prop.multiProcessorCount*per_cu;
  {
    const size_t extra = stack_need > MI_STACK ? (size_t)(stack_need - MI_STACK) : 1;
    if(hipMalloc(&s->d_overflow, extra*(size_t)s->grid*MI_BLOCK*sizeof(uint2)) != hipSuccess)
    { mi_scene_destroy(s); return fail(MI_ERR_NOMEM, "cannot allocate the traversal stack overflow area"); }
  }
  *out = s;
  return MI_OK;
}

extern "C" int mi_scene_set_framebuffer(mi_scene *s, float *device_fb)
{
  if(!s) return fail(MI_ERR_ARG, "null scene");
  s->d_fb = device_fb ? device_fb : s->d_fb_own;
  s->d.fb = s->d_fb;
  return MI_OK;
}

extern "C" int mi_scene_set_stream(mi_scene *s, void *hip_stream)
{
  if(!s) return fail(MI_ERR_ARG, "null scene");
  s->stream = hip_stream ? (hipStream_t)hip_stream : s->stream_own;
  return MI_OK;
}

extern "C" int mi_render(mi_scene *s, uint64_t first_index, uint64_t count)
{
  if(!s) return fail(MI_ERR_ARG, "null scene");
  if(!count) return MI_OK;
  HIPCHK(hipMemsetAsync(s->d_work, 0, sizeof(unsigned long long), s->stream));
  int grid = s->grid;
  const uint64_t need = (count + MI_BLOCK - 1)/MI_BLOCK;
  if((uint64_t)grid > need) grid = (int)need;
  HIPCHK(hipEventRecord(s->ev0, s->stream));
  if(s->d.sampler == MI_SAMPLER_PTDL)
    hipLaunchKernelGGL((mi_path_kernel<false, true>), dim3(grid), dim3(MI_BLOCK), s->lds_bytes, s->stream,
                       s->d, (unsigned long long)first_index, (unsigned long long)count, (const uint32_t *)s->d_shape_material,
                       (const float *)s->d_shape_L, (mi_path_record *)nullptr, (uint2 *)s->d_overflow);
  else
    hipLaunchKernelGGL((mi_path_kernel<false, false>), dim3(grid), dim3(MI_BLOCK), s->lds_bytes, s->stream,
                       s->d, (unsigned long long)first_index, (unsigned long long)count, (const uint32_t *)s->d_shape_material,
                       (const float *)s->d_shape_L, (mi_path_record *)nullptr, (uint2 *)s->d_overflow);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(s->ev1, s->stream));
  s->have_timing = 1;
  s->launches++;
  return MI_OK;
}

extern "C" int mi_sync(mi_scene *s)
{
  if(!s) return fail(MI_ERR_ARG, "null scene");
  HIPCHK(hipStreamSynchronize(s->stream));
  return MI_OK;
}

extern "C" int mi_fb_read(mi_scene *s, float *host_fb, int accumulate)
{
  if(!s || !host_fb) return fail(MI_ERR_ARG, "null argument");
  const size_t n = 3*(size_t)s->width*s->height;
  HIPCHK(hipStreamSynchronize(s->stream));
  if(!accumulate) { HIPCHK(hipMemcpy(host_fb, s->d_fb, n*sizeof(float), hipMemcpyDeviceToHost)); return MI_OK; }
  std::vector<float> tmp(n);
  HIPCHK(hipMemcpy(tmp.data(), s->d_fb, n*sizeof(float), hipMemcpyDeviceToHost));
  for(size_t i=0;i<n;i++) host_fb[i] += tmp[i];
  return MI_OK;
}

extern "C" int mi_fb_clear(mi_scene *s)
{
  if(!s) return fail(MI_ERR_ARG, "null scene");
  HIPCHK(hipMemsetAsync(s->d_fb, 0, sizeof(float)*3*(size_t)s->width*s->height, s->stream));
  return MI_OK;
}

extern "C" float *mi_fb_device_ptr(mi_scene *s) { return s ? s->d_fb : nullptr; }

extern "C" int mi_counters(mi_scene *s, uint64_t out[8])
{
  if(!s || !out) return fail(MI_ERR_ARG, "null argument");
  HIPCHK(hipStreamSynchronize(s->stream));
  unsigned long long tmp[8];
  HIPCHK(hipMemcpy(tmp, s->d_counters, sizeof(tmp), hipMemcpyDeviceToHost));
  for(int k=0;k<8;k++) out[k] = tmp[k];
  return MI_OK;
}

extern "C" int mi_trace_paths(mi_scene *s, uint64_t first_index, uint64_t count, mi_path_record *host_out)
{
  if(!s || !host_out) return fail(MI_ERR_ARG, "null argument");
  if(!count) return MI_OK;
  void *d_rec = nullptr;
  HIPCHK(hipMalloc(&d_rec, count*sizeof(mi_path_record)));
  hipError_t e = hipMemsetAsync(d_rec, 0, count*sizeof(mi_path_record), s->stream);
  if(e == hipSuccess) e = hipMemsetAsync(s->d_work, 0, sizeof(unsigned long long), s->stream);
  if(e == hipSuccess)
  {
    int grid = s->grid;
    const uint64_t need = (count + MI_BLOCK - 1)/MI_BLOCK;
    if((uint64_t)grid > need) grid = (int)need;
    if(s->d.sampler == MI_SAMPLER_PTDL)
      hipLaunchKernelGGL((mi_path_kernel<true, true>), dim3(grid), dim3(MI_BLOCK), s->lds_bytes, s->stream,
                         s->d, (unsigned long long)first_index, (unsigned long long)count, (const uint32_t *)s->d_shape_material,
                         (const float *)s->d_shape_L, (mi_path_record *)d_rec, (uint2 *)s->d_overflow);
    else
      hipLaunchKernelGGL((mi_path_kernel<true, false>), dim3(grid), dim3(MI_BLOCK), s->lds_bytes, s->stream,
                         s->d, (unsigned long long)first_index, (unsigned long long)count, (const uint32_t *)s->d_shape_material,
                         (const float *)s->d_shape_L, (mi_path_record *)d_rec, (uint2 *)s->d_overflow);
    e = hipGetLastError();
  }
  if(e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if(e == hipSuccess) e = hipMemcpy(host_out, d_rec, count*sizeof(mi_path_record), hipMemcpyDeviceToHost);
  hipFree(d_rec);
  if(e != hipSuccess) { snprintf(g_err, sizeof(g_err), "mi_trace_paths: %s", hipGetErrorString(e)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  return MI_OK;
}

extern "C" int mi_last_kernel_ms(mi_scene *s, float *ms)
{
  if(!s || !ms) return fail(MI_ERR_ARG, "null argument");
  if(!s->have_timing) { *ms = 0.0f; return MI_OK; }
  HIPCHK(hipEventSynchronize(s->ev1));
  HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
  return MI_OK;
}

extern "C" void mi_scene_destroy(mi_scene *s)
{
  if(!s) return;
  void *bufs[] = { s->d_nodes, s->d_axes, s->d_prims, s->d_primshade, s->d_materials, s->d_light_prim, s->d_light_cdf, s->d_light_L,
                   s->d_cie, s->d_checker, s->d_metal, s->d_counters, s->d_work, s->d_shape_material, s->d_shape_L, s->d_overflow, s->d_fb_own };
  for(void *b : bufs) if(b) hipFree(b);
  if(s->stream_own) hipStreamDestroy(s->stream_own);
  if(s->ev0) hipEventDestroy(s->ev0);
  if(s->ev1) hipEventDestroy(s->ev1);
  free(s);
}

extern "C" void mi_shutdown(void) { g_device = -1; }
