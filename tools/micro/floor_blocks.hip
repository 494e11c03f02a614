/* tools/micro/floor_blocks.hip -- development helper (build container, nothing runs): the blocks of the hot path compiled in isolation,
 * ONE and TWO dependent executions of each, so that tools/valu_floor.py can read the vector instructions a single execution needs from
 * the difference of the two kernels' code (no prologue, no epilogue, no loop control, every lane doing work). These are the very functions
 * of mi_kernels.h / mi_path.h, inlined as in the megakernel; inputs come from memory and results go back so that nothing folds away.
 * What the numbers are for: bench.py's `roofline.valu_floor_per_path` = (node visits x N_node + primitive tests x N_prim + vertices x
 * N_vertex + N_generate) / 64 per path -- the wave instructions the path's work needs when every instruction serves 64 useful lanes. */
#include "mi_path.h"
#define FB_BLOCK 1024

struct FbIO { float f[64]; uint32_t u[32]; };

/* ---- one node visit of the exact rounds (fast slab test: the literal SSE-semantics path of waves with an infinite 1/dir is pruned) */
template<int N> __global__ __launch_bounds__(FB_BLOCK) void fb_node_visit(DScene sc, FbIO *io, uint2 *overflow)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds lds = lds_setup<FB_BLOCK, true>(sc, smem, overflow);
  FbIO &q = io[threadIdx.x];
  TraceState ts; ts.idx = q.f[0]; ts.idy = q.f[1]; ts.idz = q.f[2]; ts.time = 0.0f; ts.sp = (int)q.u[1]; ts.done = false; ts.anyhit = false;
  const V3 o = mk3(q.f[3], q.f[4], q.f[5]), d = mk3(q.f[6], q.f[7], q.f[8]);
  RayBox rb = raybox_setup<false>(o, d, ts, lds.num_nodes);
  rb.slow = false;
  Counters<false> cnt;
  uint32_t current = q.u[0];
  int sp = ts.sp;
  float dist = q.f[9];
  lds_uint2 *lstack = (lds_uint2 *)lds.stack;
#pragma unroll
  for(int k=0;k<N;k++)
  {
    node_visit<FB_BLOCK, 7, false, false>(lds, lstack, rb, o, dist, current, sp, cnt, [&]()
    { /* the pop of the exact rounds (trace_round) */
      current = MI_LEAF32;
      while(sp > 0) { sp--; const uint2 e = stack_top<FB_BLOCK, 7>(lds, lstack, sp); if(!(__uint_as_float(e.y) > dist)) { current = e.x; break; } }
    });
    current &= 0x7fffffffu;            /* stays an inner node for the next execution */
  }
  q.u[0] = current; q.u[1] = (uint32_t)sp;
}

/* ---- one primitive test of the distributed leaf phase: record fetch + the branch-free triangle / quad test */
template<int N> __global__ __launch_bounds__(FB_BLOCK) void fb_prim_test(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  const V3 o = mk3(q.f[3], q.f[4], q.f[5]), d = mk3(q.f[6], q.f[7], q.f[8]);
  Hit h; h.prim = MI_NOPRIM; h.dist = q.f[9]; h.u = h.v = 0.0f;
  uint32_t prim = q.u[0];
#pragma unroll
  for(int k=0;k<N;k++)
  {
    const PrimRegs rec = prim_load(sc.prims, prim);
    const uint32_t type = __float_as_uint(rec.q3.x);
    const bool both = triquad_intersect<true>(rec, type, o, d, h, prim);
    prim = prim + 1u + (both ? 1u : 0u);
  }
  q.u[0] = h.prim; q.f[9] = h.dist; q.f[10] = h.u; q.f[11] = h.v;
}

/* ---- one new path: path_generate */
template<int N> __global__ __launch_bounds__(FB_BLOCK) void fb_generate(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  Counters<false> cnt;
  PathState ps;
  unsigned long long index = q.u[0];
  float acc = 0.0f;
  if(sc.pixels_from_index) return;      /* the floor is that of the sampled film position (the bench configuration): the other mode's branch folds away */
#pragma unroll
  for(int k=0;k<N;k++)
  {
    path_generate<false, false, false>(sc, ps, index, nullptr, cnt);
    index += (unsigned long long)(ps.pixel_i + ps.dir.x);
    acc += ps.lambda + ps.pdf + ps.throughput + ps.org.x + ps.org.y + ps.org.z + ps.dir.y + ps.dir.z + ps.pixel_j + ps.scramble + ps.prev_cos;
  }
  q.f[0] = acc; q.u[0] = (uint32_t)index; q.u[1] = (uint32_t)ps.rng.s0; q.u[2] = (uint32_t)ps.rng.s1;
}

/* ---- one surface vertex of the pt sampler: everything path_shade does between two rays (all material branches present in the code;
 *      valu_floor.py subtracts the bsdf blocks a vertex of a given class does not run) */
template<int N, bool PTDL> __global__ __launch_bounds__(FB_BLOCK) void fb_vertex(DScene sc, FbIO *io, const uint32_t *shape_material, const float *shape_L)
{
  FbIO &q = io[threadIdx.x];
  Counters<false> cnt;
  PathState ps;
  path_generate<false, false, false>(sc, ps, q.u[0], nullptr, cnt);
  Hit hit; hit.prim = q.u[1]; hit.dist = q.f[0]; hit.u = q.f[1]; hit.v = q.f[2];
  SplatReq splat; splat.pending = false; splat.c0 = splat.c1 = splat.c2 = 0.0f;
#pragma unroll
  for(int k=0;k<N;k++)
  {
    __builtin_assume(hit.prim != MI_NOPRIM);
    path_shade<false, PTDL, false, false, false>(sc, ps, hit, shape_material, shape_L, nullptr, cnt, splat);
    hit.prim = ps.ignore + 1u; hit.dist = ps.pdf + 1.0f + (PTDL ? ps.sh_dist + ps.sh_value + ps.sh_dir.x + ps.sh_dir.y + ps.sh_dir.z + (float)ps.sh_light + (float)ps.sh_pending : 0.0f); hit.u = ps.dir.x; hit.v = ps.dir.y;
  }
  q.f[0] = ps.org.x + ps.org.y + ps.org.z + ps.dir.z + ps.throughput + ps.prev_cos + splat.c0 + splat.c1 + splat.c2 + (float)ps.pdfprod + ps.cur_ior;
  q.u[0] = ps.active | (splat.pending ? 2u : 0u) | ((uint32_t)ps.media.ids << 2) | ((uint32_t)ps.rng.s0 << 8);
}

/* ---- ptdl: the verdict of a finished shadow ray and the splat of its connection (shadow_resolve) */
template<int N> __global__ __launch_bounds__(FB_BLOCK) void fb_shadow(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  Counters<false> cnt;
  PathState ps;
  path_generate<false, false, false>(sc, ps, q.u[0], nullptr, cnt);
  ps.sh_dist = q.f[0]; ps.sh_value = q.f[1]; ps.sh_light = q.u[1]; ps.sh_length = 3;
  Hit hit; hit.prim = q.u[2]; hit.dist = q.f[2]; hit.u = hit.v = 0.0f;
  SplatReq splat; splat.pending = false; splat.c0 = splat.c1 = splat.c2 = 0.0f;
  float acc = 0.0f;
#pragma unroll
  for(int k=0;k<N;k++)
  {
    shadow_resolve<false>(sc, ps, hit, nullptr, cnt, splat);
    acc += splat.c0 + splat.c1 + splat.c2; ps.sh_value = acc; hit.dist = acc; ps.lambda += splat.c0;
  }
  q.f[0] = acc; q.u[0] = splat.pending ? 1u : 0u;
}

/* ---- one pass of the cooperative splat: four splats, sixteen lanes each (splat_wave) */
template<int N> __global__ __launch_bounds__(FB_BLOCK) void fb_splat(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  float pi = q.f[0], pj = q.f[1], c0 = q.f[2], c1 = q.f[3], c2 = q.f[4];
#pragma unroll
  for(int k=0;k<N;k++)
  { /* exactly four lanes of the wave hold a splat: one pass */
    splat_wave(sc, (__lane_id() & 15u) == (unsigned)k, pi, pj, c0, c1, c2);
    pi += 1.0f;
  }
}

/* ---- ptdl: bsdf and pdf of a connection (the next-event block evaluates the vertex's own bsdf) */
template<int N, int WHICH> __global__ __launch_bounds__(FB_BLOCK) void fb_eval(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  Surf sf;
  sf.n = mk3(q.f[0], q.f[1], q.f[2]); sf.gn = sf.n; sf.x = mk3(0, 0, 0); sf.u = sf.v = sf.s = sf.t = 0.0f; sf.flags = q.u[2];
  get_scrambled_onb(q.f[3], sf.n, sf.a, sf.b);
  Shading sh; sh.roughness = q.f[4]; sh.rs = q.f[5]; sh.rd = q.f[6]; sh.rg = q.f[7]; sh.em = 0.0f;
  V3 wi = mk3(q.f[8], q.f[9], q.f[10]), wo = mk3(q.f[13], q.f[14], q.f[15]);
  float acc = 0.0f;
#pragma unroll
  for(int k=0;k<N;k++)
  {
    BsdfEval be;
    float pb;
    if(WHICH == 0) { be = brdf_diffuse(sf, sh, wo); pb = (float)(1.0f/MI_PI_D); }
    else if(WHICH == 1) { be = brdf_dielectric(sf, sh, wi, wo, q.f[11]); pb = pdf_dielectric(sf, sh, wi, wo, q.f[11], be.mode); }
    else { be = brdf_metal(sc, sf, sh, wi, wo, 1.0f, (int)q.u[3], q.f[12]); pb = pdf_metal(sf, sh, wi, wo, be.mode); }
    acc += be.value + pb + (float)be.mode;
    wo = mk3(wo.y + acc, wo.z, wo.x);
  }
  q.f[0] = acc;
}

/* ---- surface set-up alone, for one kind of primitive (TYPE = vertex count: 1 sphere, 2 line, 4 quad; 0 = whatever the record says: every kind's code) */
template<int N, int TYPE> __global__ __launch_bounds__(FB_BLOCK) void fb_setup(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  Surf sf;
  sf.x = mk3(q.f[0], q.f[1], q.f[2]); sf.u = q.f[3]; sf.v = q.f[4];
  V3 omega = mk3(q.f[5], q.f[6], q.f[7]);
  uint32_t prim = q.u[0];
  float acc = 0.0f;
#pragma unroll
  for(int k=0;k<N;k++)
  {
    uint4 head = *(const uint4 *)&sc.primgeo[prim];
    if(TYPE) head.x = (uint32_t)TYPE;
    surface_setup<false>(sc, prim, head, omega, q.f[8], sf, 0.0f);
    acc += sf.n.x + sf.n.y + sf.n.z + sf.gn.x + sf.gn.y + sf.gn.z + sf.u + sf.v + sf.s + sf.t + (float)sf.flags;
    omega = sf.n; prim += (uint32_t)sf.flags + 1u; sf.x = sf.gn;
  }
  q.f[0] = acc;
}

/* ---- the bsdf sample blocks alone */
template<int N, int WHICH> __global__ __launch_bounds__(FB_BLOCK) void fb_sample(DScene sc, FbIO *io)
{
  FbIO &q = io[threadIdx.x];
  Rng rng; rng.s0 = q.u[0]; rng.s1 = q.u[1];
  PointSampler<false> pts(sc, rng, 0ull, 8);
  Surf sf;
  sf.n = mk3(q.f[0], q.f[1], q.f[2]); sf.gn = sf.n; sf.x = mk3(0, 0, 0); sf.u = sf.v = sf.s = sf.t = 0.0f; sf.flags = q.u[2];
  get_scrambled_onb(q.f[3], sf.n, sf.a, sf.b);
  Shading sh; sh.roughness = q.f[4]; sh.rs = q.f[5]; sh.rd = q.f[6]; sh.rg = q.f[7]; sh.em = 0.0f;
  V3 wi = mk3(q.f[8], q.f[9], q.f[10]);
  BsdfSample bs;
  float acc = 0.0f;
#pragma unroll
  for(int k=0;k<N;k++)
  {
    if(WHICH == 0) sample_diffuse(pts, sf, sh, 0u, bs);
    else if(WHICH == 1) sample_dielectric(pts, sf, sh, wi, q.f[11], 0u, bs);
    else sample_metal(sc, pts, sf, sh, wi, 1.0f, (int)q.u[3], q.f[12], 0u, bs);
    wi = bs.omega; acc += bs.weight + bs.pdf + (float)bs.mode;
  }
  q.f[0] = acc + wi.x + wi.y + wi.z; q.u[0] = (uint32_t)rng.s0;
}

template __global__ void fb_node_visit<1>(DScene, FbIO *, uint2 *);
template __global__ void fb_node_visit<2>(DScene, FbIO *, uint2 *);
template __global__ void fb_prim_test<1>(DScene, FbIO *);
template __global__ void fb_prim_test<2>(DScene, FbIO *);
template __global__ void fb_generate<1>(DScene, FbIO *);
template __global__ void fb_generate<2>(DScene, FbIO *);
template __global__ void fb_vertex<1, false>(DScene, FbIO *, const uint32_t *, const float *);
template __global__ void fb_vertex<2, false>(DScene, FbIO *, const uint32_t *, const float *);
template __global__ void fb_vertex<1, true>(DScene, FbIO *, const uint32_t *, const float *);
template __global__ void fb_vertex<2, true>(DScene, FbIO *, const uint32_t *, const float *);
template __global__ void fb_shadow<1>(DScene, FbIO *);
template __global__ void fb_shadow<2>(DScene, FbIO *);
template __global__ void fb_splat<1>(DScene, FbIO *);
template __global__ void fb_splat<2>(DScene, FbIO *);
template __global__ void fb_sample<1, 0>(DScene, FbIO *);
template __global__ void fb_sample<2, 0>(DScene, FbIO *);
template __global__ void fb_sample<1, 1>(DScene, FbIO *);
template __global__ void fb_sample<2, 1>(DScene, FbIO *);
template __global__ void fb_sample<1, 2>(DScene, FbIO *);
template __global__ void fb_sample<2, 2>(DScene, FbIO *);
template __global__ void fb_setup<1, 0>(DScene, FbIO *);
template __global__ void fb_setup<2, 0>(DScene, FbIO *);
template __global__ void fb_setup<1, 1>(DScene, FbIO *);
template __global__ void fb_setup<2, 1>(DScene, FbIO *);
template __global__ void fb_setup<1, 2>(DScene, FbIO *);
template __global__ void fb_setup<2, 2>(DScene, FbIO *);
template __global__ void fb_setup<1, 4>(DScene, FbIO *);
template __global__ void fb_setup<2, 4>(DScene, FbIO *);
template __global__ void fb_eval<1, 0>(DScene, FbIO *);
template __global__ void fb_eval<2, 0>(DScene, FbIO *);
template __global__ void fb_eval<1, 1>(DScene, FbIO *);
template __global__ void fb_eval<2, 1>(DScene, FbIO *);
template __global__ void fb_eval<1, 2>(DScene, FbIO *);
template __global__ void fb_eval<2, 2>(DScene, FbIO *);
