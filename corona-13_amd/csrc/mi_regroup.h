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
 *   after its traversal slice a wave looks at the vertices its lanes have arrived at (class = the material's bsdf, two bits per
 *   primitive in a table staged into LDS; in the extended kernels volume vertices are a class of their own) and at the pools. If its own
 *   vertices of a class plus what that class's pool holds FILL the wave's lanes -- as far as lanes become free: ended paths + the other
 *   classes' vertices, which are posted --, it SHADES THAT CLASS: posts the others, pulls from the pool. Otherwise it POSTS ALL its
 *   vertices while entries are left, shades nothing and goes on tracing: its lanes start new paths together, and the vertices are shaded
 *   later by a wave they fill (MI_POOL_POLICY 2; policy 1 -- always shade the class most own lanes are in -- is the first version).
 *   What travels is the whole path state that is live between two rays (PoolLayout::SLOTS 8-byte words). path_shade then runs as before;
 *   the blocks of the classes the wave does not hold find no lane and are skipped. A full pool means "shade in place", which is what
 *   every kernel did before -- the exchange is an optimisation on top of the same code, not a second pipeline.
 *
 * Which lane finishes a path does not matter to the path: its generator, pixel, wavelength and throughput travel with it, counters are
 * summed over all lanes (tests/test_gpu_parity.py::test_exchange_between_waves_changes_no_path: byte-identical path records with and
 * without the exchange). The lists of entries are guarded by ONE lock word in LDS that carries the pools' counts (below); a wave holds
 * it twice per exchange for a handful of instructions, the vertices are copied outside.
 * End of the launch: a wave whose index range has run dry posts nothing and pulls from any pool; it only leaves once the pools
 * are empty, and a wave that posts is alive and will check again -- so the last wave out sees them empty.
 */
#ifndef MI_REGROUP_H
#define MI_REGROUP_H

#include "mi_path.h"
#include "mi_hero.h"

#ifndef MI_REGROUP
#define MI_REGROUP 1
#endif
#define MI_POOL_CLASSES 4          /* diffuse, dielectric, metal (DPrimGeo.cls: compact index among the bsdfs the scene uses) and, in the extended kernels,
                                      volume vertices (DScene.pool_volume_class) */
#ifndef MI_POOL_HIGH
#define MI_POOL_HIGH 48            /* own lanes + pool of a class from which on a wave turns to that class */
#endif
#ifndef MI_POOL_POLICY
#define MI_POOL_POLICY 2
#endif
#ifndef MI_POOL_AGE
#define MI_POOL_AGE 24             /* ... or when fewer than this many entries are still free */
#endif
#ifndef MI_NODES_TOP_POOL
#define MI_NODES_TOP_POOL (24*1024)   /* a tree that does not fit LDS: bytes the pools keep when the TOP of the tree is staged (mi_abi.hip); the rest of the
                                         LDS behind stacks and job lists goes to node records. Sweep: profiles/r05_large_tree.txt */
#endif
#ifndef MI_POOL_BYTES_MAX
#define MI_POOL_BYTES_MAX (48*1024)   /* trees that are traversed from HBM leave more LDS than the pools can use */
#endif

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef __attribute__((address_space(3))) unsigned short lds_u16_t;
typedef unsigned int mi_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) mi_u32x4 lds_uint4;
#define MI_SEL3(A, I) ((I) == 0 ? (A)[0] : (I) == 1 ? (A)[1] : (I) == 2 ? (A)[2] : (A)[3])
#define MI_SUM4(A) ((A)[0] + (A)[1] + (A)[2] + (A)[3])

/* control words. `state` IS the lock: five 12-bit counts {entries listed per class 0..3, free entries} packed into 64 bits while nobody
 * is inside a critical section, all ones while somebody is. A wave enters with ONE atomic exchange (all ones in, the counts out -- or
 * all ones out: spin) and leaves by storing the new counts: lock, look and unlock cost one LDS round trip together (as three
 * separate operations each was a round trip through an LDS that fifteen other waves traverse a tree in: 6 000 ticks held per
 * section). `hint` is a copy of the counts for looks from outside (written when leaving; may lag). */
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
#define MI_POOL_LOCKED (~0ull)
struct __attribute__((aligned(16))) PoolCtl { unsigned long long state, hint; };

/* 8-byte words of a path vertex on its way through a pool */
/* (HERO kernels, mi_hero.h: eight more for components 1..3 of wavelength, throughput, pdf and pdf product) */
#define MI_POOL_HERO_SLOTS 8
template<bool RECORD, bool HALTON, bool MEDIA = false, bool HERO = false> struct PoolLayout { static constexpr int BASE = RECORD ? 16 : HALTON ? 15 : 14, SLOTS = BASE + (MEDIA ? 3 : 0) + (HERO ? MI_POOL_HERO_SLOTS : 0); };

/* The pools share their storage: E entries of SLOTS words ([slot][E], so that the lanes of a wave write neighbouring addresses), a list of
 * entry numbers per class, and the list of free entries. A vertex is written / read OUTSIDE the lock (its entry is then on no list); the
 * lock only covers taking entries off lists and putting them on: two short critical sections per exchange, at the highest issue priority.
 * Tried and measured against it (profiles/r04_regroup_ab.txt): per-class stacks copied under the lock (9 700 ticks held and 78 000 waited
 * per acquisition; this version 6 100 / 12 500 in the same instrumented build) and a lock-free version that claims entries by clearing
 * bits of per-class bitmaps with LDS atomics -- no waiting at all, but 70 more vector instructions and 15-25 more spilled registers per
 * exchange: cfg 2 17.5 against 17.4 ms, cfg 3 34.3 against 32.3. */
struct Pool
{
  lds_uint2 *data;          /* [slot][E] */
  lds_u16_t *list;          /* [MI_POOL_CLASSES + 1][E]: entry numbers per class, then the free entries */
  lds_u32_t *ctl;           /* PoolCtl */
  lds_u32_t *cls;           /* the primitives' classes, two bits each, staged behind the pools -- or NULL: DScene.prim_cls through L2 */
  uint32_t E;               /* entries; 0 = no exchange */
  uint32_t score;           /* DScene.pool_score */
};

template<bool RECORD, bool HALTON, bool MEDIA, bool HERO = false>
__device__ __forceinline__ Pool pool_setup(const DScene &sc, unsigned char *base, PoolCtl *ctl)
{
  constexpr uint32_t NS = PoolLayout<RECORD, HALTON, MEDIA, HERO>::SLOTS;
  Pool p;
  p.ctl = (lds_u32_t *)ctl;
  uint32_t E = sc.pool_classes > 1u ? sc.pool_bytes/(NS*8u + 2u*(MI_POOL_CLASSES + 1u)) : 0u;
  if(E > 1024u) E = 1024u;
  E &= ~7u;
  if(E < 32u) E = 0u;
  p.E = E;
  p.score = MEDIA ? sc.pool_score : 0u;
  p.data = (lds_uint2 *)base;
  p.list = (lds_u16_t *)(base + (size_t)NS*8u*E);
  p.cls = (E && sc.pool_cls_bytes) ? (lds_u32_t *)(base + sc.pool_bytes) : nullptr;
  return p;
}
/* call from all threads of the workgroup before the barrier that starts the kernel. (One thread per entry: E <= 1024 = the scalar kernels' workgroup; the HERO
   ptdl kernels run 768 threads and their 22- to 27-word entries cap E at MI_POOL_BYTES_MAX / 186 = 264.) */
__device__ __forceinline__ void pool_init(const Pool &p, PoolCtl *ctl)
{
  if(threadIdx.x == 0) { ctl->state = (unsigned long long)p.E << 48; ctl->hint = ctl->state; }
  if(threadIdx.x < p.E) p.list[MI_POOL_CLASSES*p.E + threadIdx.x] = (unsigned short)threadIdx.x;
}
/* ... with the scene: stages the class table */
__device__ __forceinline__ void pool_stage_classes(const Pool &p, const DScene &sc)
{
  if(p.cls) for(uint32_t i=threadIdx.x;i<(sc.num_prims + 15u)/16u;i+=blockDim.x) p.cls[i] = sc.prim_cls[i];
}
/* the class of a primitive (DPrimGeo.cls): one LDS read where the table is staged -- the wave asks between its traversal slice and the
   exchange, with nothing else to do while the answer is under way */
__device__ __forceinline__ uint32_t pool_class_of(const Pool &p, const DScene &sc, uint32_t prim)
{
  const uint32_t w = p.cls ? p.cls[prim >> 4] : sc.prim_cls[prim >> 4];
  return (w >> ((prim & 15u)*2u)) & 3u;
}

__device__ __forceinline__ bool pool_empty(const Pool &p)
{ /* no complete vertex waits in any class (entries a wave is still writing are that wave's business: it is alive and looks again) */
  const unsigned long long h = __hip_atomic_load((lds_u64_t *)p.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return (h & 0xffffffffffffull) == 0ull;      /* the four classes' counts */
}

/* enter a critical section: the counts, the same in every lane */
__device__ __forceinline__ unsigned long long pool_enter(const Pool &pool)
{
  unsigned long long st = MI_POOL_LOCKED;
  if(__lane_id() == 0)
    while((st = __hip_atomic_exchange((lds_u64_t *)pool.ctl, MI_POOL_LOCKED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == MI_POOL_LOCKED) __builtin_amdgcn_s_sleep(1);
  st = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)st) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(st >> 32)) << 32);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  return st;
}
/* leave it with new counts. The fence waits for the section's READS (another wave may overwrite those list entries afterwards) and
   makes its WRITES (list entries) visible before the store that opens the lock: a workgroup-scope release in both cases -- one
   s_waitcnt lgkmcnt(0); the round-4 version relied on a wave's DS operations being carried out in order for the writes, which is what
   the hardware does but not what the memory model promises (A/B of the two: profiles/r05_levers_ab.txt).
   The hint is stored FIRST, i.e. while the lock is still held: written after the store that opens the lock, a wave that stalled between
   the two stores could overwrite a newer hint of the next holder with its stale counts -- and a stale hint that shows empty classes
   lets draining waves leave the kernel (pool_empty) while vertices are still listed. */
template<bool READS>
__device__ __forceinline__ void pool_leave(const Pool &pool, unsigned long long st)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if(__lane_id() == 0)
  {
    __hip_atomic_store((lds_u64_t *)pool.ctl + 1, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store((lds_u64_t *)pool.ctl, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
/* five 12-bit counts (at most 1024 entries) */
#define MI_POOL_UNPACK(ST, P, NFREE) { (P)[0] = (uint32_t)(ST) & 0xfffu; (P)[1] = (uint32_t)((ST) >> 12) & 0xfffu; (P)[2] = (uint32_t)((ST) >> 24) & 0xfffu; \
                                       (P)[3] = (uint32_t)((ST) >> 36) & 0xfffu; (NFREE) = (uint32_t)((ST) >> 48) & 0xfffu; }
#define MI_POOL_PACK(P, NFREE) ((unsigned long long)(P)[0] | ((unsigned long long)(P)[1] << 12) | ((unsigned long long)(P)[2] << 24) | ((unsigned long long)(P)[3] << 36) | ((unsigned long long)(NFREE) << 48))


/* a vertex into / out of an entry: the whole path state that is live between two rays (NS 8-byte words, [word][E]) */
template<bool RECORD, bool PTDL, bool HALTON, bool MEDIA, bool HERO = false>
__device__ __forceinline__ void pool_write_vertex(lds_uint2 *e, uint32_t E, PathState &ps, const Hit &hit, bool &tracing)
{
  constexpr int NB = PoolLayout<RECORD, HALTON, MEDIA>::BASE;
    const V3 o = ray_origin<PTDL>(ps, false);
    const unsigned long long pp = (unsigned long long)__double_as_longlong(ps.pdfprod);
    const uint32_t packed = ((uint32_t)ps.length & 0x3ffu) | ((ps.media.count & 0xfu) << 10) | ((ps.media.broken & 1u) << 14) |
                            ((hit.prim == ps.ignore ? 1u : 0u) << 15) | ((ps.prev_material_modes & 0xffffu) << 16);
    /* (pairs are fields that lie next to each other in PathState: the compiler widens the load of a pair's first word to both and
       keeps a struct whose widened loads overlap in private memory) */
    e[0*E] = mi_u32x2{__float_as_uint(o.x), __float_as_uint(o.y)};
    e[1*E] = mi_u32x2{__float_as_uint(o.z), __float_as_uint(ps.dir.x)};
    e[2*E] = mi_u32x2{__float_as_uint(ps.dir.y), __float_as_uint(ps.dir.z)};
    e[3*E] = mi_u32x2{hit.prim, __float_as_uint(hit.dist)};
    e[4*E] = mi_u32x2{__float_as_uint(hit.u), __float_as_uint(hit.v)};
    e[5*E] = mi_u32x2{__float_as_uint(ps.prev_cos), __float_as_uint(ps.prev_throughput)};
    e[6*E] = mi_u32x2{__float_as_uint(ps.throughput), __float_as_uint(ps.pdf)};
    e[7*E] = mi_u32x2{(uint32_t)pp, (uint32_t)(pp >> 32)};
    e[8*E] = mi_u32x2{packed, __float_as_uint(ps.cur_ior)};
    e[9*E] = mi_u32x2{(uint32_t)ps.media.ids, (uint32_t)(ps.media.ids >> 32)};
    e[10*E] = mi_u32x2{__float_as_uint(ps.pixel_i), __float_as_uint(ps.pixel_j)};
    e[11*E] = mi_u32x2{__float_as_uint(ps.lambda), __float_as_uint(ps.scramble)};
    e[12*E] = mi_u32x2{(uint32_t)ps.rng.s0, (uint32_t)(ps.rng.s0 >> 32)};
    e[13*E] = mi_u32x2{(uint32_t)ps.rng.s1, (uint32_t)(ps.rng.s1 >> 32)};
    if constexpr(NB > 14) e[14*E] = mi_u32x2{(uint32_t)ps.index, (uint32_t)(ps.index >> 32)};
    if constexpr(NB > 15) e[15*E] = mi_u32x2{ps.prev_mode, 0u};
    if constexpr(MEDIA)
    { /* extended kernels: the medium of the edge that ended here, the sampled free-flight distance (a volume vertex lies there), the path's time */
      e[(NB + 0)*E] = mi_u32x2{__float_as_uint(ps.cur.mu_s), __float_as_uint(ps.cur.mu_t)};
      e[(NB + 1)*E] = mi_u32x2{__float_as_uint(ps.cur.g), (uint32_t)ps.cur.med};
      e[(NB + 2)*E] = mi_u32x2{__float_as_uint(ps.clip), __float_as_uint(ps.time)};
    }
    if constexpr(HERO)
    {
      const PathStateHero &h = static_cast<const PathStateHero &>(ps);
      constexpr int NH = PoolLayout<RECORD, HALTON, MEDIA>::SLOTS;
      e[(NH + 0)*E] = mi_u32x2{__float_as_uint(h.lambda_x[0]), __float_as_uint(h.lambda_x[1])};
      e[(NH + 1)*E] = mi_u32x2{__float_as_uint(h.lambda_x[2]), __float_as_uint(h.throughput_x[0])};
      e[(NH + 2)*E] = mi_u32x2{__float_as_uint(h.throughput_x[1]), __float_as_uint(h.throughput_x[2])};
      e[(NH + 3)*E] = mi_u32x2{__float_as_uint(h.pdf_x[0]), __float_as_uint(h.pdf_x[1])};
      e[(NH + 4)*E] = mi_u32x2{__float_as_uint(h.pdf_x[2]), 0u};
#pragma unroll
      for(int l=0;l<3;l++) { const unsigned long long q = (unsigned long long)__double_as_longlong(h.pdfprod_x[l]); e[(NH + 5 + l)*E] = mi_u32x2{(uint32_t)q, (uint32_t)(q >> 32)}; }
    }
    tracing = false; ps.active = 0; ps.sh_pending = 0;
}
template<bool RECORD, bool PTDL, bool HALTON, bool MEDIA, bool HERO = false>
__device__ __forceinline__ void pool_read_vertex(const lds_uint2 *e, uint32_t E, PathState &ps, Hit &hit, TraceState &ts, bool &tracing, bool &tr_shadow)
{
  constexpr int NB = PoolLayout<RECORD, HALTON, MEDIA>::BASE;
    const mi_u32x2 w0 = e[0*E], w1 = e[1*E], w2 = e[2*E], w3 = e[3*E], w4 = e[4*E], w5 = e[5*E], w6 = e[6*E], w7 = e[7*E],
                   w8 = e[8*E], w9 = e[9*E], w10 = e[10*E], w11 = e[11*E], w12 = e[12*E], w13 = e[13*E];
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
    if constexpr(NB > 14) { const mi_u32x2 w14 = e[14*E]; ps.index = (unsigned long long)w14.x | ((unsigned long long)w14.y << 32); }
    if constexpr(NB > 15) { const mi_u32x2 w15 = e[15*E]; ps.prev_mode = w15.x; }
    if constexpr(MEDIA)
    {
      const mi_u32x2 a = e[(NB + 0)*E], b = e[(NB + 1)*E], c = e[(NB + 2)*E];
      ps.cur.mu_s = __uint_as_float(a.x); ps.cur.mu_t = __uint_as_float(a.y); ps.cur.g = __uint_as_float(b.x); ps.cur.med = (int)b.y;
      ps.clip = __uint_as_float(c.x); ps.time = __uint_as_float(c.y);
    }
    if constexpr(HERO)
    {
      PathStateHero &h = static_cast<PathStateHero &>(ps);
      constexpr int NH = PoolLayout<RECORD, HALTON, MEDIA>::SLOTS;
      const mi_u32x2 a = e[(NH + 0)*E], b = e[(NH + 1)*E], c = e[(NH + 2)*E], d = e[(NH + 3)*E], f = e[(NH + 4)*E];
      h.lambda_x[0] = __uint_as_float(a.x); h.lambda_x[1] = __uint_as_float(a.y); h.lambda_x[2] = __uint_as_float(b.x);
      h.throughput_x[0] = __uint_as_float(b.y); h.throughput_x[1] = __uint_as_float(c.x); h.throughput_x[2] = __uint_as_float(c.y);
      h.pdf_x[0] = __uint_as_float(d.x); h.pdf_x[1] = __uint_as_float(d.y); h.pdf_x[2] = __uint_as_float(f.x);
#pragma unroll
      for(int l=0;l<3;l++) { const mi_u32x2 q = e[(NH + 5 + l)*E]; h.pdfprod_x[l] = __longlong_as_double((long long)((unsigned long long)q.x | ((unsigned long long)q.y << 32))); }
      h.sh_value_x[0] = h.sh_value_x[1] = h.sh_value_x[2] = 0.0f;
    }
    ps.active = 1; ps.sh_pending = 0;
    if(PTDL) { ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0; }
    tracing = true; tr_shadow = false;
    ts.done = true; ts.sp = 0; ts.current = MI_LEAF32; ts.anyhit = false;
}

/* The exchange of one wave iteration. Call from ALL lanes of the wave.
 *   surf      this lane's extension ray has ended on a primitive and the vertex is not shaded yet (hit, ps are that vertex's)
 *   cls       its class
 *   freelane  this lane holds no path and no pending work
 *   drain     the workgroup's index range has run dry
 *   PRIO      the wave's issue priority outside the critical sections (inside: the highest -- a wave that holds the lock while its
 *             SIMD's other waves run their traversal slices at a higher priority keeps fifteen waves waiting)
 * Afterwards: lanes that posted are free (ps.active = 0, tracing = false); lanes that pulled hold a vertex to shade
 * (tracing = true, ts.done = true, tr_shadow = false). */
template<bool RECORD, bool PTDL, bool HALTON, bool MEDIA, int PRIO, bool HERO = false, class CNT>
__device__ __forceinline__ void regroup_exchange(const Pool &pool, PathState &ps, Hit &hit, TraceState &ts, bool &tracing, bool &tr_shadow,
                                                 bool surf, uint32_t cls, bool freelane, bool drain, CNT &cnt)
{
  constexpr int NC = MEDIA ? MI_POOL_CLASSES : MI_POOL_CLASSES - 1;      /* the plain kernels have no volume vertices: what belongs to the fourth class folds away */
  const uint32_t E = pool.E;
  if(!E) return;
  mi_u64 mc[MI_POOL_CLASSES];
  uint32_t n[MI_POOL_CLASSES] = { 0u, 0u, 0u, 0u }, p[MI_POOL_CLASSES], nfree;
  /* (the look at the pools' fill is issued before the classes of the lanes' hits are needed: the two LDS reads travel together) */
  unsigned long long hint = __hip_atomic_load((lds_u64_t *)pool.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
  for(int c=0;c<NC;c++) { mc[c] = __ballot(surf && cls == (uint32_t)c); n[c] = __popcll(mc[c]); }
  const mi_u64 mfree = __ballot(freelane);
  const uint32_t F = __popcll(mfree);
  /* what this wave will do, from the pools' fill: first on an unlocked look -- iterations of a scene with one dominant class often
     have nothing to trade --, then again under the lock. The counts are the same in every lane: as scalars the whole decision runs
     on the scalar unit; no array is indexed at run time. */
  int chosen = -1;
  uint32_t k[MI_POOL_CLASSES] = { 0u, 0u, 0u, 0u }, m = 0;
  auto look = [&]()
  { /* from outside: the hint */
    unsigned long long h = hint;
    h = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)h) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(h >> 32)) << 32);
    MI_POOL_UNPACK(h, p, nfree)
  };
  auto decide = [&]()
  {
    chosen = -1; m = 0;
#pragma unroll
    for(int c=0;c<NC;c++) k[c] = 0;
    if(drain)
    { /* nothing is posted any more; the fullest pool is emptied into the free lanes */
      uint32_t best = 0;
#pragma unroll
      for(int c=0;c<NC;c++) if(p[c] > 0u && n[c] + p[c] > best) { best = n[c] + p[c]; chosen = c; }
      if(chosen >= 0) { const uint32_t pc = MI_SEL3(p, chosen); m = pc < F ? pc : F; }
      return;
    }
    uint32_t best = 0;
#if MI_POOL_POLICY == 2
    /* policy 2: a wave shades a class only when that fills its lanes -- its own vertices of the class plus what the pool holds, as far
       as lanes become free (ended paths + the other classes' vertices, which are posted) --, and otherwise posts ALL its vertices
       while entries are left and goes on tracing: the lanes start new paths together, the vertices are shaded later by full waves */
    const uint32_t N = MI_SUM4(n);
#pragma unroll
    for(int c=0;c<NC;c++)
    {
      const uint32_t others = N - n[c] < nfree ? N - n[c] : nfree;
      const uint32_t can = F + others;
      const uint32_t t = n[c] + (p[c] < can ? p[c] : can);
      /* which of the classes that fill the wave: the largest batch -- or (Pool.score, round 6: scenes in a global fog) the one whose POOL is fullest, so that a minority
         class gets its full batches too instead of waiting until the pools are saturated with it (profiles/r06_levers.txt block 6: fog 37.1 -> 31.1 ms) */
      const uint32_t s_c = (MEDIA && pool.score) ? p[c] + 1u : t;       /* (extended kernels only: the plain kernels compile to what they were) */
      if((p[c] > 0u || n[c] > 0u) && (t >= MI_POOL_HIGH || nfree < MI_POOL_AGE) && s_c > best) { best = s_c; chosen = c; }
    }
    if(chosen < 0 && N > 0u && nfree >= N + MI_POOL_AGE)
    {
#pragma unroll
      for(int c=0;c<NC;c++) k[c] = n[c];
      return;
    }
#else
#pragma unroll
    for(int c=0;c<NC;c++)
    { /* a full batch: own lanes + pool; when the entries run out the fullest pool counts as one (its vertices must not wait for ever) */
      const uint32_t t = n[c] + p[c];
      if(p[c] > 0u && (t >= MI_POOL_HIGH || nfree < MI_POOL_AGE) && t > best) { best = t; chosen = c; }
    }
#endif
    if(chosen < 0)
    { /* the class most of the wave's own vertices are in */
      best = 0;
#pragma unroll
      for(int c=0;c<NC;c++) if(n[c] > best) { best = n[c]; chosen = c; }
      if(chosen < 0)
      { /* no vertex of its own: the fullest pool */
#pragma unroll
        for(int c=0;c<NC;c++) if(p[c] > best) { best = p[c]; chosen = c; }
        if(chosen < 0) return;
      }
    }
    uint32_t freed = F, room = nfree;
#pragma unroll
    for(int c=0;c<NC;c++) if(c != chosen)
    {
      k[c] = n[c] < room ? n[c] : room;
      room -= k[c];
      freed += k[c];
    }
    { const uint32_t pc = MI_SEL3(p, chosen); m = pc < freed ? pc : freed; }
  };
  look();
  decide();
  if(m + MI_SUM4(k) == 0u) return;
  /* (what does not depend on the counts is formed before the lock is taken) */
  uint32_t crank = 0;         /* rank of the lane's vertex among the wave's vertices of its class */
#pragma unroll
  for(int c=0;c<NC;c++)
  {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mc[c] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mc[c], 0u));
    if(cls == (uint32_t)c) crank = rank;
  }
  /* ---- first critical section: entries off the free list for the vertices to post, entries off the chosen class's list to pull */
#ifdef MI_PROFILE_POOL
  const unsigned long long t_wait = clock64();
#endif
  __builtin_amdgcn_s_setprio(3);
  {
    const unsigned long long st = pool_enter(pool);
    MI_POOL_UNPACK(st, p, nfree)
  }
#ifdef MI_PROFILE_POOL
  const unsigned long long t_got = clock64();
  MI_POOLSTAT(cnt, 3, (uint32_t)((t_got - t_wait) >> 4), 1)      /* ticks / 16 waiting for the lock */
#endif
#ifdef MI_POOL_CS_SLEEP      /* experiment: how much the kernel time depends on the length of the critical section */
  __builtin_amdgcn_s_sleep(MI_POOL_CS_SLEEP);
#endif
  /* the plan was made on the hint; inside, the true counts only cut it down */
  if(chosen >= 0 || MI_SUM4(k) != 0u)
  {
    uint32_t room = nfree, freed = F;
#pragma unroll
    for(int c=0;c<NC;c++) { if(k[c] > room) k[c] = room; room -= k[c]; freed += k[c]; }
    if(chosen >= 0) { const uint32_t pc = MI_SEL3(p, chosen); if(m > pc) m = pc; if(m > freed) m = freed; }
  }
#ifdef MI_PROFILE_POOL
  {
    uint32_t posted = 0, left = 0, nmax = 0;
    for(int c=0;c<NC;c++) { posted += k[c]; if(c != chosen) left += n[c] - k[c]; if(n[c] > nmax) nmax = n[c]; }
    MI_POOLSTAT(cnt, 0, posted, 1)
    MI_POOLSTAT(cnt, 1, m, m ? 1 : 0)
    if(chosen >= 0 && MI_SEL3(n, chosen) < nmax) MI_POOLSTAT(cnt, 2, MI_SEL3(n, chosen) + m, 1)
    MI_POOLSTAT(cnt, 4, left, left ? 1 : 0)
    if(chosen >= 0) MI_POOLSTAT(cnt, 5, MI_SEL3(n, chosen) + m, 1)       /* lanes of the chosen class shaded in an iteration with an exchange */
  }
#endif
  const bool post = surf && (int)cls != chosen && crank < MI_SEL3(k, cls);
  const mi_u64 mpost = __ballot(post);
  const uint32_t K = MI_SUM4(k);
  const uint32_t prank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mpost >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mpost, 0u));
  const mi_u64 mtake = mfree | mpost;
  const uint32_t trank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mtake >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mtake, 0u));
  const bool pull = chosen >= 0 && (freelane || post) && trank < m;
  uint32_t id_post = 0, id_pull = 0;
  if(post) id_post = pool.list[MI_POOL_CLASSES*E + (nfree - 1u - prank)];
  if(pull) id_pull = pool.list[(uint32_t)chosen*E + (MI_SEL3(p, chosen) - 1u - trank)];
  {
    uint32_t q[MI_POOL_CLASSES] = { p[0], p[1], p[2], p[3] };
#pragma unroll
    for(int c=0;c<NC;c++) if(c == chosen) q[c] -= m;
    pool_leave<true>(pool, MI_POOL_PACK(q, nfree - K));
  }
#ifdef MI_PROFILE_POOL
  MI_POOLSTAT(cnt, 6, (uint32_t)((clock64() - t_got) >> 4), 1)   /* ticks / 16 holding it */
#endif
  __builtin_amdgcn_s_setprio(PRIO);
  /* ---- the vertices themselves, outside the lock: the entries are on no list */
  if(post) pool_write_vertex<RECORD, PTDL, HALTON, MEDIA, HERO>(pool.data + id_post, E, ps, hit, tracing);
  if(pull) pool_read_vertex<RECORD, PTDL, HALTON, MEDIA, HERO>(pool.data + id_pull, E, ps, hit, ts, tracing, tr_shadow);
  /* ---- second critical section: the written entries onto their classes' lists, the read ones back onto the free list */
  __builtin_amdgcn_s_setprio(3);
  {
    const unsigned long long st = pool_enter(pool);
    MI_POOL_UNPACK(st, p, nfree)
  }
  if(post) pool.list[cls*E + MI_SEL3(p, cls) + crank] = (unsigned short)id_post;
  if(pull) pool.list[MI_POOL_CLASSES*E + nfree + trank] = (unsigned short)id_pull;
  {
    const uint32_t q[MI_POOL_CLASSES] = { p[0] + k[0], p[1] + k[1], p[2] + k[2], p[3] + k[3] };
    pool_leave<false>(pool, MI_POOL_PACK(q, nfree + m));
  }
  __builtin_amdgcn_s_setprio(PRIO);
}


#endif
