/* mi_wavefront.h -- the second persistent kernel of libcorona_mi.so (round 6): rays and vertices as work items of a workgroup.
 *
 * mi_path_kernel (mi_megakernel.h) ties a path to a LANE: the lane traces the path's ray, shades the vertex, traces the next ray. Its traversal
 * slice therefore runs at 46 % of its lane slots (profiles/r06_lanes.txt: of 12.2 slots per ray in the node loop 5.65 visit a node, 3.6 wait with a
 * leaf for the round's leaf phase, 2.9 belong to lanes whose ray has ended or that hold none), its shading passes at 48 of 64 lanes, path_generate
 * at 41 -- a lane whose ray ends early has nothing to do until the wave's slowest rays let the slice end, and the exchange between waves
 * (mi_regroup.h) can only sort what the pools' 212 entries hold.
 *
 * Here a path is an ENTRY of a per-workgroup table in global memory (L2-resident: 14 eight-byte words per path, [word][entry] so that a wave's
 * accesses are contiguous) and the sixteen waves of the workgroup are workers that take whatever job is there in full batches:
 *
 *   generate   64 free entries + 64 path indices  -> path_generate -> 64 rays                     (all 64 lanes start a path)
 *   shade      64 vertices of ONE material class  -> path_shade    -> rays / ended paths          (one bsdf's code, all lanes in it)
 *   trace      a wave that traces holds only RAYS (origin, direction, closest hit, traversal state: 17 registers a lane); whenever
 *              MI_WF_TOPUP of its lanes have finished theirs it writes the hits back to the entries, hands the entries to their
 *              classes' queues and takes new rays from the ray queue -- the traversal rounds (trace_round + leaf_jobs, unchanged: same
 *              visits, same tests, same hits, the reference's counters) go on with full lanes instead of decaying to a tail.
 *
 * The queues (entry numbers, 16 bits) live in LDS under the one lock word that carries all counts (mi_regroup.h: pool_enter / pool_leave);
 * an entry number is on exactly one list or held by exactly one lane. What a wave does next is decided from the counts alone:
 * shade a class that fills 64 lanes, else start 64 paths, else trace; partial batches only when nothing else can make progress.
 * The hot functions are the megakernel's own (path_generate, path_shade, trace_round, leaf_jobs, splat_wave): a path takes the same
 * branches with the same numbers, the records of the RECORD instantiations are byte for byte those of mi_path_kernel
 * (tests/test_gpu_wavefront.py).
 *
 * Replaces, like mi_path_kernel: the pool dispatch of src/view.c:643-645 over work_sample (src/view.c:618-628) -> render_sample_path
 * (src/render.d/gi.c:81-105) -> sampler_create_path (src/sampler.d/pt.c:30-54).
 */
#ifndef MI_WAVEFRONT_H
#define MI_WAVEFRONT_H

#include "mi_megakernel.h"

#ifndef MI_WAVEFRONT_DEFAULT
#define MI_WAVEFRONT_DEFAULT 0    /* 1: plain pt scenes render with this kernel unless CORONA_MI_WAVEFRONT=0 (mi_abi.hip) */
#endif
#ifndef MI_WF_TOPUP
#define MI_WF_TOPUP 16        /* a tracing wave turns to the queues when this many of its lanes are without a ray under way and rays wait */
#endif
#ifndef MI_WF_RETIRE
#define MI_WF_RETIRE 32       /* ... or when this many finished rays wait to be handed on (no rays in the queue) */
#endif
#ifndef MI_WF_ENTRIES
#define MI_WF_ENTRIES 3968    /* paths in flight per workgroup at most (12-bit counts in the lock word: < 4096) */
#endif
#ifndef MI_WF_COLUMN
#define MI_WF_COLUMN 8        /* stack entries per lane in LDS (three of them the leaf phase's result slots): shorter columns than the megakernel's ten leave
                                 the lists room for twice the entries */
#endif
#ifndef MI_WF_DEFER
#define MI_WF_DEFER 0         /* 1: entries are handed on one turn late (no wait for the stores into them) */
#endif
#ifndef MI_WF_PARTIAL
#define MI_WF_PARTIAL 1       /* vertices of a class from which on an idle wave shades a partial batch while other waves still trace */
#endif

/* -DMI_PROFILE_WF (development build, tools/wf_probe.py): the seven work counters of the production kernel become
 *   1: 0 shading passes, 1 lanes in them, 2 generating passes, 3 lanes in them, 5 queue turns of tracing waves, 6 rays taken in them, 7 (max) unused
 *   2: lane 0's clock ticks / 16 in 0 shading, 1 generating, 2 queue turns of tracing waves, 3 traversal rounds, 5 waiting, 6 deciding
 *   3: 0 traversal rounds, 1 busy lanes in them, 2 waits, 3 partial shading passes, 5 episodes of tracing, 6 lanes valid at a queue turn */
#ifdef MI_PROFILE_WF
#define MI_WFP(MODE, K, V) { if(MI_PROFILE_WF == (MODE) && __lane_id() == 0) pw[K] += (V); }
#define MI_WFT(K) { if(MI_PROFILE_WF == 2) { const unsigned long long t_ = clock64(); if(__lane_id() == 0) pw[K] += (uint32_t)((t_ - t_last) >> 4); t_last = t_; } }
#else
#define MI_WFP(MODE, K, V) {}
#define MI_WFT(K) {}
#endif

/* An entry = one path, 128 bytes = ONE cache line (the lanes of a wave hold entries from anywhere in the table: word-major arrays cost a line per
   word and lane -- 7.6 ms per 16 spp at 1024 entries against the megakernel's 4.5, profiles/r06_wavefront.txt), eight 16-byte quads:
     0  origin x y z, direction x          1  direction y z, primitive the ray starts on, packed {length, medium stack count / broken, previous modes}
     2  the hit: primitive, distance | came back to the primitive it left << 31, u, v          (written by the tracing wave)
     3  prev_cos, prev_throughput, throughput, pdf        4  pdf product (double), cur_ior, lambda
     5  medium stack ids (64 bits), pixel i j             6  generator state                        7  scramble, path index (64 bits), previous vertex' mode
   the records' instantiations append quad 8: the previous vertex' position (environment vertex of a path that leaves the scene) */
template<bool RECORD> struct WfLayout { static constexpr uint32_t QUADS = RECORD ? 10u : 8u; };
#define MI_WF_QUADS_MAX 10u

typedef unsigned int mi_g32x4 __attribute__((ext_vector_type(4)));

/* the path state after path_generate / path_shade -> entry (FRESH: a new path, the quad no vertex changes is written too) */
template<bool RECORD, bool HALTON, bool FRESH>
__device__ __forceinline__ void wf_write_path(mi_g32x4 *e, const PathState &ps)
{
  const unsigned long long pp = (unsigned long long)__double_as_longlong(ps.pdfprod);
  const uint32_t packed = ((uint32_t)ps.length & 0x3ffu) | ((ps.media.count & 0xfu) << 10) | ((ps.media.broken & 1u) << 14) | ((ps.prev_material_modes & 0xffffu) << 16);
  e[0] = mi_g32x4{__float_as_uint(ps.org.x), __float_as_uint(ps.org.y), __float_as_uint(ps.org.z), __float_as_uint(ps.dir.x)};
  e[1] = mi_g32x4{__float_as_uint(ps.dir.y), __float_as_uint(ps.dir.z), ps.ignore, packed};
  e[3] = mi_g32x4{__float_as_uint(ps.prev_cos), __float_as_uint(ps.prev_throughput), __float_as_uint(ps.throughput), __float_as_uint(ps.pdf)};
  e[4] = mi_g32x4{(uint32_t)pp, (uint32_t)(pp >> 32), __float_as_uint(ps.cur_ior), __float_as_uint(ps.lambda)};
  e[5] = mi_g32x4{(uint32_t)ps.media.ids, (uint32_t)(ps.media.ids >> 32), __float_as_uint(ps.pixel_i), __float_as_uint(ps.pixel_j)};
  e[6] = mi_g32x4{(uint32_t)ps.rng.s0, (uint32_t)(ps.rng.s0 >> 32), (uint32_t)ps.rng.s1, (uint32_t)(ps.rng.s1 >> 32)};
  if(FRESH || RECORD) e[7] = mi_g32x4{__float_as_uint(ps.scramble), (uint32_t)ps.index, (uint32_t)(ps.index >> 32), ps.prev_mode};
  if(RECORD) e[8] = mi_g32x4{__float_as_uint(ps.prev_x.x), __float_as_uint(ps.prev_x.y), __float_as_uint(ps.prev_x.z), 0u};
}
/* entry -> the vertex a shading wave works on: the path state and the hit its ray ended in */
template<bool RECORD, bool HALTON>
__device__ __forceinline__ void wf_read_vertex(const mi_g32x4 *e, PathState &ps, Hit &hit)
{
  const mi_g32x4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3], q4 = e[4], q5 = e[5], q6 = e[6], q7 = e[7];
  ps.org = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
  ps.dir = mk3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
  hit.prim = q2.x; hit.dist = __uint_as_float(q2.y & 0x7fffffffu);
  hit.u = __uint_as_float(q2.z); hit.v = __uint_as_float(q2.w);
  ps.ignore = (q2.y >> 31) ? hit.prim : MI_NOPRIM;       /* path_shade only asks whether the ray came back to the primitive it left */
  const uint32_t packed = q1.w;
  ps.length = (int)(packed & 0x3ffu);
  ps.media.count = (packed >> 10) & 0xfu; ps.media.broken = (packed >> 14) & 1u;
  ps.prev_material_modes = packed >> 16;
  ps.prev_cos = __uint_as_float(q3.x); ps.prev_throughput = __uint_as_float(q3.y);
  ps.throughput = __uint_as_float(q3.z); ps.pdf = __uint_as_float(q3.w);
  ps.pdfprod = __longlong_as_double((long long)((unsigned long long)q4.x | ((unsigned long long)q4.y << 32)));
  ps.cur_ior = __uint_as_float(q4.z); ps.lambda = __uint_as_float(q4.w);
  ps.media.ids = (unsigned long long)q5.x | ((unsigned long long)q5.y << 32);
  ps.pixel_i = __uint_as_float(q5.z); ps.pixel_j = __uint_as_float(q5.w);
  ps.rng.s0 = (unsigned long long)q6.x | ((unsigned long long)q6.y << 32);
  ps.rng.s1 = (unsigned long long)q6.z | ((unsigned long long)q6.w << 32);
  ps.scramble = __uint_as_float(q7.x);
  ps.index = (unsigned long long)q7.y | ((unsigned long long)q7.z << 32);
  ps.prev_mode = q7.w; ps.prev_x = ps.org; ps.org_eps = 0.0f;
  if(RECORD) { const mi_g32x4 q8 = e[8]; ps.prev_x = mk3(__uint_as_float(q8.x), __uint_as_float(q8.y), __uint_as_float(q8.z)); }
  ps.active = 1; ps.sh_pending = 0;
  ps.sh_dir = mk3(0.0f, 0.0f, 0.0f); ps.sh_dist = 0.0f; ps.sh_value = 0.0f; ps.sh_light = 0u; ps.sh_length = 0;
  ps.cur = medium_vacuum(); ps.clip = FLT_MAX; ps.time = 0.0f;
}

/* ======================================================================================= the kernel */
template<bool RECORD, bool PTDL, bool NODES_LDS, bool HALTON, bool COUNT>
__global__ __launch_bounds__(MI_BLOCK) void mi_wave_kernel(DScene sc_arg, unsigned long long first, unsigned long long count,
                                                           const uint32_t *shape_material, const float *shape_L, mi_path_record *records,
                                                           uint2 *stack_overflow, uint2 *table_all, uint32_t table_entries)
{
  static_assert(!PTDL, "the wavefront kernel renders the pt sampler");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BLK = MI_BLOCK;
  constexpr int COLUMN = MI_WF_COLUMN;                  /* the host lays the scene's LDS out for both kernels (DScene.wf_list_bytes) */
  constexpr int STACK = COLUMN - MI_JOB_SLOTS;
  constexpr uint32_t QUADS = WfLayout<RECORD>::QUADS;
  static_assert(STACK >= MI_STACK_MIN, "the overflow area is sized for MI_STACK_MIN entries in LDS (mi_abi.hip)");
  const Lds lds = lds_setup<BLK, NODES_LDS, HALTON, PTDL, COLUMN, false>(sc_arg, smem, stack_overflow);
  __shared__ unsigned int blk_next, n_tracing;
  __shared__ PoolCtl wf_ctl;
  /* queues behind the job lists: lists 0..2 = vertices by class, 3 = rays, 4 = free entries */
  Pool q;
  {
    unsigned char *base = lds.jobs - (threadIdx.x >> 6)*MI_JOBS_LDS + (BLK/64)*MI_JOBS_LDS;
    uint32_t E = sc_arg.wf_list_bytes/(2u*(MI_POOL_CLASSES + 1u));
    if(E > table_entries) E = table_entries;
    E &= ~63u;
    q.E = E; q.score = 0u; q.data = nullptr; q.list = (lds_u16_t *)base; q.ctl = (lds_u32_t *)&wf_ctl;
    q.cls = sc_arg.pool_cls_bytes ? (lds_u32_t *)(base + sc_arg.wf_list_bytes) : nullptr;
    if(threadIdx.x == 0) { wf_ctl.state = (unsigned long long)E << 48; wf_ctl.hint = wf_ctl.state; blk_next = 0u; n_tracing = 0u; }
    for(uint32_t i=threadIdx.x;i<E;i+=BLK) q.list[MI_POOL_CLASSES*E + i] = (unsigned short)i;
    pool_stage_classes(q, sc_arg);
  }
  __syncthreads();
  const uint32_t E = q.E;
  mi_g32x4 *table = (mi_g32x4 *)table_all + (size_t)blockIdx.x*MI_WF_QUADS_MAX*table_entries;
  const unsigned long long nb = gridDim.x;
  const unsigned long long blk_lo = count/nb*blockIdx.x + (blockIdx.x < count%nb ? blockIdx.x : count%nb);
  const uint32_t blk_n = (uint32_t)(count/nb + (blockIdx.x < count%nb ? 1 : 0));        /* < 2^31 (launch_chunk, mi_abi.hip) */
  const unsigned lane = __lane_id();
  Counters<COUNT || RECORD> cnt;
#define MI_MBCNT64(M) __builtin_amdgcn_mbcnt_hi((uint32_t)((M) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(M), 0u))
#define MI_SCALAR(X) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(X)))

  /* One turn at the queues -- the only place the lists are touched. Every lane may hand ONE entry on (`pend`: bit 31 set, the list in bits 16..18, the entry
     in the low 16 bits -- words it wrote into the entry since are made visible first) and the wave may take up to `want_n` entries off list `from`, the
     lanes with `want` set in lane order. Entries are handed on one turn late on purpose: the stores into an entry have long arrived when the wave comes
     back to the queues, so the release costs no wait (as a turn right behind the stores it cost ~1500 cycles per shading pass and per tracing round). */
  auto turn = [&](uint32_t &pend, int from, uint32_t want_n, bool want, uint32_t &got) -> uint32_t
  {
    const bool has = (pend >> 31) != 0u;
    const uint32_t pl = (pend >> 16) & 7u;
    const mi_u64 mh = __ballot(has);
    mi_u64 ml[MI_POOL_CLASSES + 1];
    uint32_t rank = 0;
#pragma unroll
    for(uint32_t l=0;l<=MI_POOL_CLASSES;l++)
    {
      ml[l] = __ballot(has && pl == l);
      const uint32_t r = MI_MBCNT64(ml[l]);
      if(pl == l) rank = r;
    }
    const mi_u64 mw = __ballot(want);
    const uint32_t rw = MI_MBCNT64(mw);
    if(mh) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        /* the entries' words are written before their numbers are listed */
    uint32_t c[MI_POOL_CLASSES + 1], m = 0;
    const unsigned long long st = pool_enter(q);
    MI_POOL_UNPACK(st, c, c[MI_POOL_CLASSES])
    if(has) q.list[pl*E + (pl == 0u ? c[0] : pl == 1u ? c[1] : pl == 2u ? c[2] : pl == 3u ? c[3] : c[4]) + rank] = (unsigned short)(pend & 0xffffu);
#pragma unroll
    for(uint32_t l=0;l<=MI_POOL_CLASSES;l++) c[l] += (uint32_t)__popcll(ml[l]);
    if(from >= 0)
    {
      const uint32_t top = from == 0 ? c[0] : from == 1 ? c[1] : from == 2 ? c[2] : from == 3 ? c[3] : c[4];
      const uint32_t nw = (uint32_t)__popcll(mw);
      m = top < want_n ? top : want_n;
      if(m > nw) m = nw;
      if(want && rw < m) got = q.list[(uint32_t)from*E + top - 1u - rw];
#pragma unroll
      for(uint32_t l=0;l<=MI_POOL_CLASSES;l++) if((int)l == from) c[l] -= m;
    }
    pool_leave<true>(q, MI_POOL_PACK(c, c[MI_POOL_CLASSES]));
    pend = 0u;
    if(m) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return m;
  };
#define MI_WF_PEND(LIST, ID) (0x80000000u | ((uint32_t)(LIST) << 16) | (ID))

  uint32_t idle_spins = 0;
  uint32_t pend = 0u;            /* the entry this lane hands on at the wave's next turn at the queues */
#ifdef MI_PROFILE_WF
  uint32_t pw[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
  unsigned long long t_last = clock64();
#endif
  while(true)
  {
    /* the scene descriptor through the kernarg segment (MI_SCENE_LAZY, mi_megakernel.h) */
    const DScene *sc_lazy;
    {
      typedef __attribute__((address_space(4))) const unsigned char *mi_kernarg_ptr;
      mi_kernarg_ptr k = (mi_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(k));
      sc_lazy = (const DScene *)k;
    }
    const DScene &sc = *sc_lazy;

    /* ------------------------------------------------------------ what is there to do? (a wave arrives here holding no entry but what it hands on) */
    uint32_t p[MI_POOL_CLASSES], nfree;
    {
      unsigned long long h = __hip_atomic_load((lds_u64_t *)q.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      h = (unsigned long long)MI_SCALAR((uint32_t)h) | ((unsigned long long)MI_SCALAR((uint32_t)(h >> 32)) << 32);
      MI_POOL_UNPACK(h, p, nfree)
    }
    /* (what this wave is about to hand on counts: a wave that waits for its own entries waits for ever) */
    const mi_u64 mpend = __ballot((pend >> 31) != 0u);
    const uint32_t own3 = (uint32_t)__popcll(__ballot((pend >> 31) && ((pend >> 16) & 7u) == 3u)), own4 = (uint32_t)__popcll(__ballot((pend >> 31) && ((pend >> 16) & 7u) == 4u));
    const uint32_t taken = MI_SCALAR(__hip_atomic_load(&blk_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    const uint32_t remain = taken < blk_n ? blk_n - taken : 0u;
    const uint32_t ntr = MI_SCALAR(__hip_atomic_load(&n_tracing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    int cbest = 0;
    uint32_t pbest = p[0];
    if(p[1] > pbest) { pbest = p[1]; cbest = 1; }
    if(p[2] > pbest) { pbest = p[2]; cbest = 2; }
    const uint32_t rays = p[3] + own3, nfree_own = nfree + own4;
    enum { DO_SHADE, DO_GENERATE, DO_TRACE, DO_WAIT, DO_EXIT };
    int job;
    if(pbest >= 64u) job = DO_SHADE;
    else if(nfree_own >= 64u && remain > 0u) job = DO_GENERATE;
    else if(rays > 0u) job = DO_TRACE;
    else if(pbest > 0u && (ntr == 0u || pbest >= (uint32_t)MI_WF_PARTIAL*64u)) job = DO_SHADE;       /* nobody will bring more: a partial batch */
    else if(nfree_own > 0u && remain > 0u && ntr == 0u && pbest == 0u) job = DO_GENERATE;
    else if(nfree == E && remain == 0u && !mpend) job = DO_EXIT;
    else job = DO_WAIT;

    MI_WFT(6)
    if(job == DO_EXIT) break;
    if(job == DO_WAIT)
    {
      if(mpend) { uint32_t none = 0; (void)turn(pend, -1, 0u, false, none); continue; }      /* what this wave holds back may be what the others wait for */
      __builtin_amdgcn_s_sleep(16);
      MI_WFP(3, 2, 1)
      MI_WFT(5)
      if(++idle_spins > (1u << 24)) { if(lane == 0) atomicMax(sc_arg.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8 + 7, 0xdead0000ull); break; }    /* a bounded wait: a stuck workgroup leaves its mark in the stack-depth counter and the launch ends (the path count is then short) */
      continue;
    }
    idle_spins = 0;

    if(job == DO_SHADE || job == DO_GENERATE)
    {
      /* ---------------------------------------------------------- take up to 64 entries off the class's list / the free list */
      uint32_t id = 0;
      const uint32_t m = turn(pend, job == DO_SHADE ? cbest : (int)MI_POOL_CLASSES, job == DO_SHADE ? 64u : (remain < 64u ? remain : 64u), true, id);
      if(m == 0u) continue;
      if(job == DO_SHADE) { MI_WFP(1, 0, 1) MI_WFP(1, 1, m) if(m < 64u) MI_WFP(3, 3, 1) } else { MI_WFP(1, 2, 1) MI_WFP(1, 3, m) }
      const bool have = lane < m;
      PathState ps;
      ps.active = 0; ps.sh_pending = 0;
      SplatReq splat;
      splat.pending = false; splat.c0 = splat.c1 = splat.c2 = 0.0f;
      bool alive = false;
      if(job == DO_SHADE)
      {
        if(have)
        {
          Hit hit;
          wf_read_vertex<RECORD, HALTON>(table + id*QUADS, ps, hit);
          if(!RECORD) __builtin_assume(hit.prim != MI_NOPRIM);       /* rays that left the scene end where they were traced */
          path_shade<RECORD, PTDL, HALTON, false, false>(sc, ps, hit, shape_material, shape_L, RECORD ? records + (ps.index - first) : nullptr, cnt, splat);
          alive = ps.active != 0;
          if(alive) wf_write_path<RECORD, HALTON, false>(table + id*QUADS, ps);
        }
      }
      else
      {
        uint32_t base = 0;
        if(lane == 0) base = atomicAdd(&blk_next, m);
        base = MI_SCALAR(base);
        const unsigned long long i = blk_lo + base + lane;
        if(have && base + lane < blk_n)
        {
          if(!RECORD && sc.tile_members)
          {
            float px, py;
            const unsigned long long index = tile_path(sc, first + i, px, py);
            path_generate<RECORD, HALTON, false>(sc, ps, index, nullptr, cnt, px, py);
          }
          else path_generate<RECORD, HALTON, false>(sc, ps, first + i, RECORD ? records + i : nullptr, cnt);
          alive = true;
          wf_write_path<RECORD, HALTON, true>(table + id*QUADS, ps);
        }
      }
      /* the paths that go on are rays now, the others' entries are free -- handed on at the next turn */
      if(have) pend = MI_WF_PEND(alive ? 3u : (uint32_t)MI_POOL_CLASSES, id);
      if(!MI_WF_DEFER) { uint32_t none = 0; (void)turn(pend, -1, 0u, false, none); }
      if(!RECORD && job == DO_SHADE) splat_wave(sc, splat.pending, ps.pixel_i, ps.pixel_j, splat.c0, splat.c1, splat.c2);
      if(job == DO_SHADE) { MI_WFT(0) } else { MI_WFT(1) }
      continue;
    }

    /* ------------------------------------------------------------ trace: this wave's lanes hold rays until the ray queue is empty and theirs have ended */
    {
      bool valid = false;
      uint32_t id = 0, ignore = MI_NOPRIM;
      V3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(1.0f, 0.0f, 0.0f);
      Hit hit;
      hit.prim = MI_NOPRIM; hit.dist = FLT_MAX; hit.u = hit.v = 0.0f;
      TraceState ts;
      ts.done = true; ts.sp = 0; ts.current = MI_LEAF32; ts.anyhit = false; ts.idx = ts.idy = ts.idz = 0.0f; ts.time = 0.0f; ts.prims_t1 = nullptr;
      MI_WFP(3, 5, 1)
      if(MI_PRIO) __builtin_amdgcn_s_setprio(MI_PRIO_PT_TRACE);      /* as in the megakernel: the traversal's chains of dependent LDS reads first */
      while(true)
      {
        const bool fin = valid && ts.done;
        const mi_u64 mfin = __ballot(fin);
        const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid)), nfin = (uint32_t)__popcll(mfin), nbusy = nvalid - nfin;
        uint32_t qrays;
        {
          unsigned long long h = __hip_atomic_load((lds_u64_t *)q.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          qrays = MI_SCALAR((uint32_t)(h >> 36)) & 0xfffu;
        }
        const bool own_rays = __any((pend >> 31) && ((pend >> 16) & 7u) == 3u);      /* (a wave that turns from shading to tracing brings its rays along) */
        if(nbusy == 0u || nfin >= (uint32_t)MI_WF_RETIRE || ((qrays > 0u || own_rays) && 64u - nbusy >= (uint32_t)MI_WF_TOPUP))
        { /* a turn at the queues: what was finished by the last turn is handed on, new rays are taken; the rays finished since have their hits
             written into their entries and are handed on at the next turn */
          MI_WFP(1, 5, 1) MI_WFP(3, 6, nvalid)
          uint32_t nid = 0;
          const bool want = fin || !valid;
          auto retire = [&]()
          {
            if(fin)
            {
              const bool esc = hit.prim == MI_NOPRIM;
              if(RECORD || !esc)
              { /* (the records' environment vertex is written by path_shade) */
                const uint32_t cls = esc ? 0u : pool_class_of(q, sc, hit.prim);
                table[id*QUADS + 2u] = mi_g32x4{hit.prim, __float_as_uint(hit.dist) | ((hit.prim == ignore && hit.prim != MI_NOPRIM) ? 0x80000000u : 0u),
                                                __float_as_uint(hit.u), __float_as_uint(hit.v)};
                pend = MI_WF_PEND(cls, id);
              }
              else
              { /* black sky: nothing to add, the path ends (path_escape, mi_path.h) */
                MI_COUNT(cnt, 6, 1); cnt.c[4]++;
                pend = MI_WF_PEND(MI_POOL_CLASSES, id);
              }
              valid = false;
            }
          };
          if(!MI_WF_DEFER) retire();
          const uint32_t m = turn(pend, 3, 64u, want, nid);
          MI_WFP(1, 6, m)
          if(MI_WF_DEFER) retire();
          if(lane == 0 && m != nfin) atomicAdd(&n_tracing, m - nfin);
          if(want && MI_MBCNT64(__ballot(want)) < m)
          {
            id = nid;
            const mi_g32x4 q0 = table[id*QUADS], q1 = table[id*QUADS + 1u];
            o = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
            d = mk3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
            ignore = q1.z;
            hit.prim = MI_NOPRIM; hit.dist = FLT_MAX; hit.u = hit.v = 0.0f;
            trace_begin(lds, ts, d, cnt);
            valid = true;
          }
          MI_WFT(2)
          if(!__any(valid)) { if(MI_PRIO) __builtin_amdgcn_s_setprio(MI_PRIO_PT_SHADE); break; }            /* nothing left and nothing to take: this wave is free for other work (what it hands on goes with it) */
        }
        const bool busy = valid && !ts.done;
        if(__any(busy))
        {
#ifdef MI_PROFILE_TRAV
          cnt.c[31] = (uint32_t)clock64(); cnt.c[15]++;
#endif
          if(busy) trace_round<BLK, STACK, false, false, true>(lds, sc.prims, o, d, ignore, hit, ts, cnt);
          leaf_jobs<BLK, STACK, false, false>(lds, sc.prims, o, d, ignore, hit, ts, busy, cnt);
          MI_WFP(3, 0, 1) MI_WFP(3, 1, (uint32_t)__popcll(__ballot(busy)))
          MI_WFT(3)
        }
      }
    }
  }
#undef MI_WF_PEND
#undef MI_MBCNT64
#undef MI_SCALAR

#ifdef MI_PROFILE_TRAV     /* development build: counters 0..3 become lane 0's ticks in the node loop, the job set-up, the job passes and the owners' epilogue of the
                              traversal rounds (mi_kernels.h: MI_TT), 7 the number of rounds */
  {
    unsigned long long *sh = sc_arg.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8;
    if(lane == 0) for(int k=0;k<8;k++) if(k != 4) atomicAdd(sh + k, (unsigned long long)cnt.c[8 + k]);
  }
#endif
#ifdef MI_PROFILE_LOOPS    /* development build: box hits / splats / vertices become wave-level inner iterations / leaf slots / analytic passes (mi_megakernel.h) */
  cnt.c[2] = cnt.c[8]; cnt.c[5] = cnt.c[9]; cnt.c[6] = cnt.c[10];
#endif
  unsigned long long *shard = sc_arg.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8;
  if(cnt.on) atomicMax(shard + 7, (unsigned long long)cnt.c[7]);     /* deepest traversal stack use */
#pragma unroll
  for(int k=0;k<7;k++)
  {
#ifdef MI_PROFILE_WF
    if(!cnt.on && k != 4) { if(lane == 0 && pw[k]) atomicAdd(shard + k, (unsigned long long)pw[k]); continue; }
#endif
    if(!cnt.on && k != 4) continue;         /* the plain kernels only count paths */
    unsigned long long c = cnt.c[k];
    for(int off=32;off>0;off>>=1) c += __shfl_down(c, off);
    if(lane == 0 && c) atomicAdd(shard + k, c);
  }
}

/* ---------------------------------------------------------------------------------------- kernel table (as mi_path_part, mi_megakernel.h) */
template<bool PTDL> const void *mi_wave_part(unsigned which, const PathLaunch *L);

#ifdef MI_PART_DEFINE
template<bool PTDL, bool R, bool N, bool H, bool C> static const void *mi_wave_go(const PathLaunch *L)
{
  constexpr bool valid = !(R && !C) && !PTDL
#ifdef MI_DEV_FAST
                         && (!H || MI_DEV_FAST == 2) && N
#endif
                         ;
  if constexpr(valid)
  {
    if(L) hipLaunchKernelGGL((mi_wave_kernel<R, PTDL, N, H, C>), dim3(L->grid), dim3(MI_BLOCK), L->lds_bytes, L->stream, L->d, L->first, L->n,
                             L->shape_material, L->shape_L, L->rec, L->overflow, L->table, L->table_entries);
    return (const void *)mi_wave_kernel<R, PTDL, N, H, C>;
  }
  else { fprintf(stderr, "[mi] internal: kernel variant not built\n"); abort(); }
}
template<bool PTDL, int LEFT, bool... B> static const void *mi_wave_pick(unsigned which, const PathLaunch *L)
{
  if constexpr(LEFT == 0) return mi_wave_go<PTDL, B...>(L);
  else return (which & 1u) ? mi_wave_pick<PTDL, LEFT - 1, B..., true>(which >> 1, L) : mi_wave_pick<PTDL, LEFT - 1, B..., false>(which >> 1, L);
}
template<bool PTDL> const void *mi_wave_part(unsigned which, const PathLaunch *L)
{
  return mi_wave_pick<PTDL, 4>(which, L);
}
#endif

#endif
