/* mi_regroup.h -- material queues inside the persistent megakernel (north_star: "sorted material queues").
 *
 * The reference shades a vertex through its material's function pointers (shader_prepare / shader_sample / shader_brdf,
 * /root/reference/src/shader.c:462-590; the largest divergent block is src/shaders/dielectric.c:240-415). A wave of the megakernel that
 * shades whatever its 64 lanes happened to hit runs EVERY material's code in almost every iteration, each at a fraction of its lanes
 * (tools/block_probe.py: 25 of 64 lanes in the diffuse block, 11 in the dielectric one, both executed in 99 % of the passes).
 *
 * Here the sixteen waves of a workgroup trade path vertices through per-class POOLS in LDS, without ever waiting for each other (a
 * workgroup barrier per iteration was measured first: +30 % kernel time on cfg 2, +21 % on cfg 3 -- the waves' traversal slices are too
 * uneven for lock step, profiles/r04_regroup_ab.txt):
 *
 *   after its traversal slice a wave looks at the surface vertices its lanes have arrived at (class = the material's bsdf, one
 *   word per primitive, DPrimGeo.cls) and at the pools; it picks ONE class to shade in this iteration -- a class whose pool plus
 *   its own lanes make a full batch if there is one, else the class most of its own lanes are in --, POSTS the vertices of the other
 *   classes into their pools (the whole path state that is live between two rays: MI_POOL_SLOTS 8-byte words) and PULLS vertices of
 *   the chosen class into the lanes that have become free (those whose path has just ended + those that posted).
 *   path_shade then runs as before; its other classes' blocks find no lane and are skipped. A full pool means "shade in place",
 *   which is what every kernel did before -- the exchange is an optimisation on top of the same code, not a second pipeline.
 *
 * Which lane finishes a path does not matter to the path: its generator, pixel, wavelength and throughput travel with it, counters are
 * summed over all lanes. The pools are guarded by ONE spin lock in LDS taken by a wave for the ~30 LDS instructions of an exchange.
 * End of the launch: a wave whose index range has run dry posts nothing and pulls from any pool; it only leaves once the pools
 * are empty, and a wave that posts is alive and will check again -- so the last wave out sees them empty.
 */
#ifndef MI_REGROUP_H
#define MI_REGROUP_H

#include "mi_path.h"

#ifndef MI_REGROUP
#define MI_REGROUP 1
#endif
#define MI_POOL_CLASSES 3          /* diffuse, dielectric, metal (DPrimGeo.cls: compact index among the bsdfs the scene uses) */
#ifndef MI_POOL_HIGH
#define MI_POOL_HIGH 48            /* own lanes + pool of a class from which on a wave turns to that class */
#endif
#ifndef MI_POOL_AGE
#define MI_POOL_AGE 16             /* ... or when fewer than this many entries of its pool are still free */
#endif
#ifndef MI_POOL_BYTES_MAX
#define MI_POOL_BYTES_MAX (48*1024)   /* trees that are traversed from HBM leave more LDS than the pools can use */
#endif
#ifndef MI_POOL_MIN_POST
#define MI_POOL_MIN_POST 1
#endif

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
#define MI_SEL3(A, I) ((I) == 0 ? (A)[0] : (I) == 1 ? (A)[1] : (A)[2])

struct PoolCtl { uint32_t lock, cnt[MI_POOL_CLASSES]; };

/* 8-byte words of a path vertex on its way through a pool */
template<bool RECORD, bool HALTON> struct PoolLayout { static constexpr int SLOTS = RECORD ? 16 : HALTON ? 15 : 14; };

struct Pool
{
  lds_uint2 *data;          /* [class][slot][cap] */
  lds_u32_t *ctl;           /* PoolCtl */
  uint32_t cap;             /* entries per class; 0 = no exchange */
  uint32_t classes;
};

template<bool RECORD, bool HALTON>
__device__ __forceinline__ Pool pool_setup(const DScene &sc, unsigned char *base, PoolCtl *ctl)
{
  Pool p;
  p.data = (lds_uint2 *)base;
  p.ctl = (lds_u32_t *)ctl;
  p.classes = sc.pool_classes;
  p.cap = p.classes > 1u ? sc.pool_bytes/(uint32_t)(PoolLayout<RECORD, HALTON>::SLOTS*8)/p.classes : 0u;
  if(p.cap > 255u) p.cap = 255u;
  if(p.cap < 16u) p.cap = 0u;
  return p;
}

__device__ __forceinline__ uint32_t pool_total(const Pool &p)
{
  uint32_t t = 0;
  for(int c=0;c<MI_POOL_CLASSES;c++) t += __hip_atomic_load(p.ctl + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return t;
}

/* The exchange of one wave iteration. Call from ALL lanes of the wave.
 *   surf      this lane's extension ray has ended on a primitive and the vertex is not shaded yet (hit, ps are that vertex's)
 *   cls       its class
 *   freelane  this lane holds no path and no pending work
 *   drain     the workgroup's index range has run dry
 * Afterwards: lanes that posted are free (ps.active = 0, tracing = false); lanes that pulled hold a vertex to shade
 * (tracing = true, ts.done = true, tr_shadow = false). */
template<bool RECORD, bool PTDL, bool HALTON>
__device__ __forceinline__ void regroup_exchange(const Pool &pool, PathState &ps, Hit &hit, TraceState &ts, bool &tracing, bool &tr_shadow,
                                                 bool surf, uint32_t cls, bool freelane, bool drain)
{
  constexpr int NS = PoolLayout<RECORD, HALTON>::SLOTS;
  const uint32_t cap = pool.cap;
  if(!cap) return;
  const unsigned lane = __lane_id();
  mi_u64 mc[MI_POOL_CLASSES];
  uint32_t n[MI_POOL_CLASSES], p[MI_POOL_CLASSES];
#pragma unroll
  for(int c=0;c<MI_POOL_CLASSES;c++) { mc[c] = __ballot(surf && cls == (uint32_t)c); n[c] = __popcll(mc[c]); }
  const mi_u64 mfree = __ballot(freelane);
  const uint32_t F = __popcll(mfree);
  /* what this wave will do, from the pools' fill (decide): first on an unlocked look -- most iterations of a scene with one dominant
     class have nothing to trade --, then again under the lock */
  int chosen = -1;
  uint32_t k[MI_POOL_CLASSES], m = 0;
  auto decide = [&]()
  {
    chosen = -1; m = 0;
#pragma unroll
    for(int c=0;c<MI_POOL_CLASSES;c++) k[c] = 0;
    if(drain)
    { /* nothing is posted any more; the fullest pool is emptied into the free lanes */
      uint32_t best = 0;
#pragma unroll
      for(int c=0;c<MI_POOL_CLASSES;c++) if(p[c] > 0u && n[c] + p[c] > best) { best = n[c] + p[c]; chosen = c; }
      if(chosen >= 0) { const uint32_t pc = MI_SEL3(p, chosen); m = pc < F ? pc : F; }
      return;
    }
    uint32_t best = 0;
#pragma unroll
    for(int c=0;c<MI_POOL_CLASSES;c++)
    { /* a full batch: own lanes + pool; a pool about to overflow counts as one (its entries must not wait for ever) */
      const uint32_t t = n[c] + p[c];
      if(p[c] > 0u && (t >= MI_POOL_HIGH || p[c] + MI_POOL_AGE > cap) && t > best) { best = t; chosen = c; }
    }
    if(chosen < 0)
    { /* the class most of the wave's own vertices are in */
      best = 0;
#pragma unroll
      for(int c=0;c<MI_POOL_CLASSES;c++) if(n[c] > best) { best = n[c]; chosen = c; }
      if(chosen < 0)
      { /* no vertex of its own: the fullest pool */
#pragma unroll
        for(int c=0;c<MI_POOL_CLASSES;c++) if(p[c] > best) { best = p[c]; chosen = c; }
        if(chosen < 0) return;
      }
    }
    uint32_t freed = F;
#pragma unroll
    for(int c=0;c<MI_POOL_CLASSES;c++) if(c != chosen)
    {
      const uint32_t room = cap - p[c];
      k[c] = n[c] < room ? n[c] : room;
      if(k[c] < MI_POOL_MIN_POST) k[c] = 0;
      freed += k[c];
    }
    { const uint32_t pc = MI_SEL3(p, chosen); m = pc < freed ? pc : freed; }
  };
  /* (the counts are the same in every lane: as scalars the whole decision runs on the scalar unit; no array is indexed at run time) */
#pragma unroll
  for(int c=0;c<MI_POOL_CLASSES;c++) p[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(pool.ctl + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  decide();
  {
    uint32_t any = m;
#pragma unroll
    for(int c=0;c<MI_POOL_CLASSES;c++) any += k[c];
    if(!any) return;
  }
  /* ---- lock (one lane spins; LDS operations of a wave are carried out in order, so the data written under the lock is in place
     before the store that releases it) */
  if(lane == 0)
    while(__hip_atomic_exchange(pool.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for(int c=0;c<MI_POOL_CLASSES;c++) p[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(pool.ctl + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  decide();
  /* ---- post */
  bool post = false;
  uint32_t entry = 0, pcls = 0;
#pragma unroll
  for(int c=0;c<MI_POOL_CLASSES;c++)
  {
    if(k[c] == 0u) continue;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mc[c] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mc[c], 0u));
    if(surf && cls == (uint32_t)c && rank < k[c]) { post = true; entry = p[c] + rank; pcls = (uint32_t)c; }
  }
  const mi_u64 mpost = __ballot(post);
  if(post)
  {
    lds_uint2 *e = pool.data + (size_t)pcls*NS*cap + entry;
    const V3 o = ray_origin<PTDL>(ps, false);
    const unsigned long long pp = (unsigned long long)__double_as_longlong(ps.pdfprod);
    const uint32_t packed = ((uint32_t)ps.length & 0x3ffu) | ((ps.media.count & 0xfu) << 10) | ((ps.media.broken & 1u) << 14) |
                            ((hit.prim == ps.ignore ? 1u : 0u) << 15) | ((ps.prev_material_modes & 0xffffu) << 16);
    e[0*cap] = mi_u32x2{__float_as_uint(o.x), __float_as_uint(o.y)};
    e[1*cap] = mi_u32x2{__float_as_uint(o.z), __float_as_uint(ps.dir.x)};
    e[2*cap] = mi_u32x2{__float_as_uint(ps.dir.y), __float_as_uint(ps.dir.z)};
    e[3*cap] = mi_u32x2{hit.prim, __float_as_uint(hit.dist)};
    e[4*cap] = mi_u32x2{__float_as_uint(hit.u), __float_as_uint(hit.v)};
    /* (pairs are fields that lie next to each other in PathState: the compiler widens the load of a pair's first word to both and
       keeps a struct whose widened loads overlap in private memory) */
    e[5*cap] = mi_u32x2{__float_as_uint(ps.prev_cos), __float_as_uint(ps.prev_throughput)};
    e[6*cap] = mi_u32x2{__float_as_uint(ps.throughput), __float_as_uint(ps.pdf)};
    e[7*cap] = mi_u32x2{(uint32_t)pp, (uint32_t)(pp >> 32)};
    e[8*cap] = mi_u32x2{packed, __float_as_uint(ps.cur_ior)};
    e[9*cap] = mi_u32x2{(uint32_t)ps.media.ids, (uint32_t)(ps.media.ids >> 32)};
    e[10*cap] = mi_u32x2{__float_as_uint(ps.pixel_i), __float_as_uint(ps.pixel_j)};
    e[11*cap] = mi_u32x2{__float_as_uint(ps.lambda), __float_as_uint(ps.scramble)};
    e[12*cap] = mi_u32x2{(uint32_t)ps.rng.s0, (uint32_t)(ps.rng.s0 >> 32)};
    e[13*cap] = mi_u32x2{(uint32_t)ps.rng.s1, (uint32_t)(ps.rng.s1 >> 32)};
    if constexpr(NS > 14) e[14*cap] = mi_u32x2{(uint32_t)ps.index, (uint32_t)(ps.index >> 32)};
    if constexpr(NS > 15) e[15*cap] = mi_u32x2{ps.prev_mode, 0u};
    tracing = false; ps.active = 0; ps.sh_pending = 0;
  }
  /* ---- pull: the free lanes (those that just posted included) take the top m entries of the chosen pool */
  const mi_u64 mtake = mfree | mpost;
  const uint32_t trank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mtake >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mtake, 0u));
  const bool pull = chosen >= 0 && (freelane || post) && trank < m;
  if(pull)
  {
    const lds_uint2 *e = pool.data + (size_t)chosen*NS*cap + (MI_SEL3(p, chosen) - 1u - trank);
    const mi_u32x2 w0 = e[0*cap], w1 = e[1*cap], w2 = e[2*cap], w3 = e[3*cap], w4 = e[4*cap], w5 = e[5*cap], w6 = e[6*cap], w7 = e[7*cap],
                   w8 = e[8*cap], w9 = e[9*cap], w10 = e[10*cap], w11 = e[11*cap], w12 = e[12*cap], w13 = e[13*cap];
    const V3 o = mk3(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w1.x));
    if(PTDL) { ps.prev_x = o; ps.org_eps = 0.0f; }      /* ray_origin: prev_x + 0 * dir */
    else ps.org = o;
    ps.dir = mk3(__uint_as_float(w1.y), __uint_as_float(w2.x), __uint_as_float(w2.y));
    hit.prim = w3.x; hit.dist = __uint_as_float(w3.y);
    hit.u = __uint_as_float(w4.x); hit.v = __uint_as_float(w4.y);
    ps.prev_cos = __uint_as_float(w5.x); ps.prev_throughput = __uint_as_float(w5.y);
    ps.throughput = __uint_as_float(w6.x); ps.pdf = __uint_as_float(w6.y);
    ps.pdfprod = __longlong_as_double((long long)((unsigned long long)w7.x | ((unsigned long long)w7.y << 32)));
    ps.cur_ior = __uint_as_float(w8.y);
    const uint32_t packed = w8.x;
    ps.length = (int)(packed & 0x3ffu);
    ps.media.count = (packed >> 10) & 0xfu; ps.media.broken = (packed >> 14) & 1u;
    ps.ignore = ((packed >> 15) & 1u) ? hit.prim : MI_NOPRIM;     /* path_shade only asks whether the ray came back to the primitive it left */
    ps.prev_material_modes = packed >> 16;
    ps.media.ids = (unsigned long long)w9.x | ((unsigned long long)w9.y << 32);
    ps.pixel_i = __uint_as_float(w10.x); ps.pixel_j = __uint_as_float(w10.y);
    ps.lambda = __uint_as_float(w11.x); ps.scramble = __uint_as_float(w11.y);
    ps.rng.s0 = (unsigned long long)w12.x | ((unsigned long long)w12.y << 32);
    ps.rng.s1 = (unsigned long long)w13.x | ((unsigned long long)w13.y << 32);
    if constexpr(NS > 14) { const mi_u32x2 w14 = e[14*cap]; ps.index = (unsigned long long)w14.x | ((unsigned long long)w14.y << 32); }
    if constexpr(NS > 15) { const mi_u32x2 w15 = e[15*cap]; ps.prev_mode = w15.x; }
    ps.active = 1; ps.sh_pending = 0;
    if(PTDL) { ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0; }
    tracing = true; tr_shadow = false;
    ts.done = true; ts.sp = 0; ts.current = MI_LEAF32; ts.anyhit = false;
  }
  /* ---- new counts, unlock */
  if(lane == 0)
  {
#pragma unroll
    for(int c=0;c<MI_POOL_CLASSES;c++)
    {
      const uint32_t v = p[c] + k[c] - (c == chosen ? m : 0u);
      if(v != p[c]) __hip_atomic_store(pool.ctl + 1 + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    /* the entries are written, the pulled ones read (s_waitcnt lgkmcnt(0)) */
  if(lane == 0) __hip_atomic_store(pool.ctl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#endif
