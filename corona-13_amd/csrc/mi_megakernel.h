/* mi_megakernel.h -- the persistent path tracing kernel (mi_path_kernel) and the ray-level test kernel (mi_intersect_kernel) of
 * libcorona_mi.so for gfx950, and the table of their instantiations.
 *
 * The megakernel has eight template switches (RECORD, PTDL, NODES_LDS, HALTON, MEDIA, MB, COUNT, FAST); the valid combinations are
 * compiled in PARTS -- one translation unit per (PTDL, MEDIA, MB, FAST), csrc/mi_part.hip with -DMI_PART=k -- so that the library
 * builds in parallel (make -j), and each part dispatches over its remaining four switches (mi_path_part). mi_abi.hip holds the
 * host side and picks the part.
 */
#ifndef MI_MEGAKERNEL_H
#define MI_MEGAKERNEL_H

#include "mi_path.h"
#include "mi_hero.h"
#include "mi_regroup.h"
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

#ifndef MI_BLOCK
#define MI_BLOCK 1024    /* threads per workgroup: 16 waves/CU = 4 per SIMD (128 VGPRs each) */
#endif
/* ... per kind of kernel (the LDS layout scales with it: lds_setup<BLOCK>; the host lays the scene out for MI_BLOCK, a smaller workgroup uses less).
   HERO kernels (mi_hero.h; 18 registers more path state), same box, 59 M paths: pt 1024 / 768 / 512 threads: 20.5 / 22.5 / - ms (without the exchange
   26.4 / 29.6 / 40.1); ptdl 49.6 / 42.3 / - (64.2 / 60.9 / 79.8): the ptdl kernel spills 135 registers at 128, the pt kernel 55 */
#ifndef MI_BLOCK_PTDL
#define MI_BLOCK_PTDL MI_BLOCK
#endif
#ifndef MI_BLOCK_HERO
#define MI_BLOCK_HERO 1024
#endif
#ifndef MI_BLOCK_HERO_PTDL
#define MI_BLOCK_HERO_PTDL 768
#endif
#ifndef MI_BLOCK_PTDL_MEDIA
#define MI_BLOCK_PTDL_MEDIA MI_BLOCK_PTDL     /* the extended ptdl kernels (media, moving camera, motion blur). 768 threads, same box (profiles/r05_hero.txt): media 17.2 -> 20.0 ms,
                                                 media ptdl 32.2 -> 35.3, fog 37.5 -> 43.5, fog ptdl 115.1 -> 122.4, moving camera 16.6 -> 19.5, moving geometry 26.9 -> 32.1:
                                                 every scalar kernel wants its sixteen waves, however much it spills (107 registers in the media ptdl kernel) */
#endif
#ifndef MI_BLOCK_MEDIA
#define MI_BLOCK_MEDIA MI_BLOCK               /* the extended pt kernels */
#endif
#define MI_BLOCK_OF(HERO, PTDL, MEDIA) ((HERO) ? ((PTDL) ? MI_BLOCK_HERO_PTDL : MI_BLOCK_HERO) : (MEDIA) ? ((PTDL) ? MI_BLOCK_PTDL_MEDIA : MI_BLOCK_MEDIA) : ((PTDL) ? MI_BLOCK_PTDL : MI_BLOCK))
#ifndef MI_TAIL_LANES
#define MI_TAIL_LANES 16       /* a traversal slice ends when fewer rays than this are still under way */
#endif
#ifndef MI_TAIL_LANES_PTDL
#define MI_TAIL_LANES_PTDL 12  /* ptdl shades more per vertex (next event estimation); A/B 6 / 8 / 10 / 12 / 16: 38.67 / 38.44 / 38.22 / 38.14 / 38.19 ms */
#endif
#ifndef MI_ANYHIT
#define MI_ANYHIT 1      /* shadow rays towards flagged emitters stop at the first occluder (MI_LIGHT_ANYHIT, mi_device.h) */
#endif
#ifndef MI_SHADOW_CACHE
#define MI_SHADOW_CACHE 0   /* 1: the production ptdl kernels test an any-hit shadow ray against the lane's LAST OCCLUDER before they traverse (the reference's cache in
                               front of accel_visible, src/accel.d/qbvhmp.c:1392-1490: one primitive per thread, :186-187, :1405). The verdict cannot change -- any
                               occluder inside the connection is an occluder. Measured (profiles/r06_levers.txt) and switched off: a lane's consecutive shadow rays
                               belong to unrelated vertices (the exchange moves paths between lanes), the test costs every wave that starts one shadow ray a
                               primitive test, and the cache hits too rarely to pay for it */
#endif
#ifndef MI_LEAF_JOBS
#define MI_LEAF_JOBS 1
#endif
#ifndef MI_LEAF_JOBS_MEDIA_PTDL
#define MI_LEAF_JOBS_MEDIA_PTDL 1   /* ... and in the extended ptdl kernels: a loss before path_shade_volume retired its dead values early too (scenes/0055_media 51.1
                                       against 48.9 ms), a gain since: 48.1 against 48.9, global fog 119.1 / 120.7, moving camera 39.8 / 41.2 */
#endif
#ifndef MI_LEAF_JOBS_PTDL
#define MI_LEAF_JOBS_PTDL 1   /* the distributed leaf phase in the ptdl kernels too: a loss as long as the kernel spilled 30-47 registers (rounds 1-2:
                                 +3 %), a gain since the path state's dead values are retired early (MI_EARLY_KILL, mi_path.h): cfg 3 36.9 -> 35.9 ms */
#endif
#ifndef MI_LEAF_JOBS_MEDIA
#define MI_LEAF_JOBS_MEDIA 1  /* ... in the extended (media / moving camera) pt kernels: +7 % (0055_media, 0056_fog, 0058_cam_mb) */
#endif
#ifndef MI_LEAF_JOBS_MB
#define MI_LEAF_JOBS_MB 1     /* ... in the motion-blur kernels. Rounds 2-5: off (1748 against 1923 Msamples/s on 0059_mb: every moving primitive was a put-off test, the
                                 job passes found nothing to do and cost three stack entries). Round 6: a moving triangle / quad IS a job -- the worker interpolates its
                                 vertices at the owner's time (leaf_jobs, mi_kernels.h): 0059_mb 26.7 -> 22.7 ms (profiles/r06_levers.txt) */
#endif
#ifndef MI_PARK_TRACE
#define MI_PARK_TRACE 1   /* the tail lanes' traversal state waits in LDS while the others shade: 1 = in the ptdl FAST kernel (38.6 against 40.4 ms),
                             2 = in every kernel with a distributed leaf phase (costs the pt kernels 0.1-0.2 ms of 19: same-box A/B) */
#endif
#ifndef MI_CHAIN
#define MI_CHAIN 0        /* 1: ptdl kernels (exact rounds): a lane traces the shadow ray and the extension ray of a vertex back to back in one slice.
                             Same results; cfg 3 40.0 ms against 38.2 at the slice tail of 12 lanes, 37.96 against 38.0 at 24 (same-box A/B): the
                             lanes without a connection wait for the chained ones as long as those used to wait for the slice to end. Off.
                             2 (round 6): the same in the plain ptdl kernels NEXT TO the exchange between waves -- the chained connection is splatted in front of the
                             exchange, which may hand the vertex and with it the lane's pixel to another wave. Same paths; cfg 3 31.9 against 26.9 ms (98 against 12
                             spilled registers: the path state a slice may now read stays live through it; profiles/r06_levers.txt block 9). Off. */
#endif
#ifndef MI_PRIO
#define MI_PRIO 1         /* issue priority of a wave (s_setprio) by part of its iteration: the pt kernels put the traversal slice first (its chains
                             of dependent LDS reads are what a wave waits for), the ptdl kernels the shading (next event estimation, where their
                             spilled registers come back). Same-box A/B, cfg 2 / cfg 3: slice first 18.73 / 38.57 ms, shading first 19.09 / 38.02,
                             no priorities 18.89 / 38.32 */
#endif
#ifndef MI_PRIO_PT_TRACE
#define MI_PRIO_PT_TRACE 3
#define MI_PRIO_PT_SHADE 0
#endif
#ifndef MI_PRIO_PTDL_TRACE
#define MI_PRIO_PTDL_TRACE 0
#define MI_PRIO_PTDL_SHADE 3
#endif
#ifndef MI_SCENE_LAZY
#define MI_SCENE_LAZY 1     /* A/B (profiles/r05_levers_ab.txt): cfg 2 15.82 -> 15.48 ms, cfg 3 28.46 -> 28.21; spilled SGPRs 89 -> 24 (pt), 151 -> 82 (ptdl) */
#endif
#ifndef MI_REGROUP_MB
#define MI_REGROUP_MB 0      /* ... in the motion-blur kernels (round 5 experiment): the pools take the place of the lowest levels of the tree, which are then read from L2.
                                0059_mb, same box: 30.31 ms with (33-59 spilled registers more, hybrid node fetch), 26.73 ms without, the whole tree in LDS: off */
#endif
#ifndef MI_REGROUP_MEDIA
#define MI_REGROUP_MEDIA 1   /* the exchange in the extended kernels too (media, moving camera): volume vertices are a class of their own */
#endif
#ifndef MI_PARK_HERO_PT
#define MI_PARK_HERO_PT 0    /* entries of path state the HERO pt kernel parks in the lane's stack column during a slice (its 55 spilled registers against a shallower LDS
                                stack). Same box, cfg 2 / cfg 4, 0 / 3 / 4 entries: 20.43 / 20.36 / 20.64 ms and 22.18 / 22.07 / 22.41: nothing to gain, off */
#endif
#ifndef MI_REGROUP_HERO
#define MI_REGROUP_HERO 1    /* the exchange in the HERO kernels (A/B: profiles/r05_hero.txt) */
#endif
#ifndef MI_REFILL_MIN
#define MI_REFILL_MIN 12   /* with the exchange between waves (mi_regroup.h) the lanes of a wave become free in bursts (a wave that posts all its vertices) and in
                                  dribbles (one that pulled what the pool had): the dribbles wait. cfg 2 / cfg 3 with 1 / 8 / 16 / 24: 15.94 / 15.74 / 15.77 / 15.89 ms and
                                  30.3 / 29.4 / 29.3 / 29.3; without the exchange a loss (18.7 against 18.4 ms) */
#endif
#ifndef MI_REGROUP_EARLY_SHADOW
#define MI_REGROUP_EARLY_SHADOW 0
#endif
#ifndef MI_PARK_PATH
#define MI_PARK_PATH 2    /* FAST kernels (1: ptdl only): part of the path state waits in LDS for the length of a traversal slice (PARK_PS) */
#endif
#ifndef MI_PARK_ENTRIES
#define MI_PARK_ENTRIES 4      /* ptdl: 8 dwords parked (generator, pdf product, pixel); A/B 4 / 5 / 6 / 7 entries: 38.79 / 40.35 / 38.62 / 39.73 ms */
#endif
#ifndef MI_PARK_ENTRIES_EXACT
#define MI_PARK_ENTRIES_EXACT 3   /* ptdl kernels with the exact rounds and the distributed leaf phase: generator and pdf product (6 dwords) wait in the lane's
                                     stack column during a slice: 16 against 21 spilled registers, cfg 3 35.9 against 36.3 ms (2 / 4 / 5 entries: 18 / 18 / 18) */
#endif
#ifndef MI_PARK_ENTRIES_PT
#define MI_PARK_ENTRIES_PT 4   /* pt: generator, pdf product, pixel (one spilled register left; 2 / 3 / 4 entries: 18.24 / 18.23 / 18.22 ms at 6 / 3 / 1 spilled
                                  registers, 5 entries: none, but the shallower LDS stack costs 1 %) */
#endif
/* the shallowest LDS stack among the instantiations (FAST kernels with parked path state; motion-blur kernels): the overflow area
   in HBM is sized for it */
#define MI_PARK_ENTRIES_MAX (MI_PARK_ENTRIES > MI_PARK_ENTRIES_PT ? MI_PARK_ENTRIES : MI_PARK_ENTRIES_PT)
#ifndef MI_STACK_LDS_RG
#define MI_STACK_LDS_RG 10     /* entries per lane of the plain kernels' stack columns since they trade vertices through pools in LDS (mi_regroup.h): 16 KB more
                                  for the pools are worth more than two stack entries -- cfg 2 / cfg 3, columns of 12 / 11 / 10 / 9 / 8 entries:
                                  (17.3) / 16.28 / 16.01 / 15.93 / 16.09 ms and (31.9) / 30.8 / 30.4 / 30.4 / 30.6 ms; the same columns without the exchange: 18.5 / 18.4 */
#endif
#define MI_STACK_LDS_PLAIN (MI_REGROUP ? MI_STACK_LDS_RG : MI_STACK_LDS)
#define MI_STACK_MIN ((MI_STACK_LDS_PLAIN - 3 - MI_PARK_ENTRIES_MAX) < MI_STACK_LDS_MB ? (MI_STACK_LDS_PLAIN - 3 - MI_PARK_ENTRIES_MAX) : MI_STACK_LDS_MB)
#ifndef MI_STACK
#if MI_LEAF_JOBS
#define MI_STACK (MI_STACK_LDS - 3)   /* the top three entries of a lane's column hold the results of the distributed leaf phase (leaf_jobs) */
#else
#define MI_STACK MI_STACK_LDS         /* LDS traversal stack entries per lane; deeper entries overflow to HBM (mi_device.h) */
#endif
#endif

/* ======================================================================================= persistent megakernel */
template<bool RECORD, bool PTDL, bool NODES_LDS, bool HALTON = false, bool MEDIA = false, bool MB = false, bool COUNT = true, bool FAST = false, bool NORG = false, bool HERO = false>
__global__ __launch_bounds__(MI_BLOCK_OF(HERO, PTDL, MEDIA)) void mi_path_kernel(DScene sc, unsigned long long first, unsigned long long count,
                                                           const uint32_t *shape_material, const float *shape_L, mi_path_record *records,
                                                           uint2 *stack_overflow)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BLK = MI_BLOCK_OF(HERO, PTDL, MEDIA);     /* threads of the workgroup */
  constexpr int COLUMN = MB ? MI_STACK_LDS_MB : (NORG || (MEDIA && !MI_REGROUP_MEDIA)) ? MI_STACK_LDS : MI_STACK_LDS_PLAIN;      /* stack entries per lane in LDS */
  const Lds lds = lds_setup<BLK, NODES_LDS, HALTON, PTDL && !MEDIA, COLUMN, MB>(sc, smem, stack_overflow);

  /* every workgroup owns a contiguous part of the path index range and hands it out through an LDS counter: the
     wave-level refill below then needs no global atomic at all (one shared counter costs ~11 ns per wave refill) */
  __shared__ unsigned int blk_next;
  /* material queues (mi_regroup.h): the plain kernels trade surface vertices between the waves of the workgroup, by class of the material */
  /* NORG: the extended kernels once more WITHOUT the exchange. Rounds 4-5 ran scenes in a scattering exterior medium (a global fog) on them: three of four
     vertices are volume vertices there and the rule "shade the largest batch" never turned to the surface ones (fog ptdl 124 against 115 ms). Round 6: with the
     rule "shade the class whose pool is fullest" (DScene.pool_score, mi_regroup.h) the exchange pays in a fog too (fog 37.3 -> 31.3 ms) and such scenes run the
     kernels WITH it; these instantiations remain for CORONA_MI_NORG=1 and for builds without MI_REGROUP_MEDIA */
  /* HERO (mi_hero.h): four wavelengths per path, plain scenes. The LDS layout is the plain kernels' (the scene was laid out for them); a pool
     entry is eight words longer, so the pools hold fewer */
  static_assert(!HERO || !FAST, "hero wavelengths: exact rounds");
  constexpr bool REGROUP = MI_REGROUP && (!MB || MI_REGROUP_MB) && !NORG && (!MEDIA || MI_REGROUP_MEDIA) && (!HERO || MI_REGROUP_HERO);
  __shared__ PoolCtl pool_ctl;
  if(threadIdx.x == 0) blk_next = 0;
  Pool pool;
  pool.E = 0u;
  if(REGROUP)
  {
    pool = pool_setup<RECORD, HALTON, MEDIA, HERO>(sc, lds.jobs - (threadIdx.x >> 6)*MI_JOBS_LDS + (BLK/64)*MI_JOBS_LDS, &pool_ctl);
    pool_init(pool, &pool_ctl);
    pool_stage_classes(pool, sc);
  }
  __syncthreads();
  const unsigned long long nb = gridDim.x;
  const unsigned long long blk_lo = count/nb*blockIdx.x + (blockIdx.x < count%nb ? blockIdx.x : count%nb);
  const unsigned long long blk_hi = blk_lo + count/nb + (blockIdx.x < count%nb ? 1 : 0);

  /* the primitive tests of a traversal round are dealt out over all lanes of the wave (leaf_jobs) -- in the pt kernels and, since the
     path state's dead values are retired early (MI_EARLY_KILL), in the ptdl kernels; the motion-blur kernels keep the per-lane leaf
     loop: every moving primitive is a put-off test there (A/B in DESIGN.md) */
  constexpr bool JOBS = !FAST && MI_LEAF_JOBS && (!PTDL || MI_LEAF_JOBS_PTDL) && (!MEDIA || MI_LEAF_JOBS_MEDIA) && (!(PTDL && MEDIA) || MI_LEAF_JOBS_MEDIA_PTDL) && (!MB || MI_LEAF_JOBS_MB);
  /* PARK_PS: the part of the path state no traversal round looks at (generator, pdf product, pixel: 8 dwords) waits in the lane's LDS
     column for the length of a slice, so that the rounds' registers (a job pass holds a whole primitive record and a second ray) do
     not push path state into scratch, from where the shading blocks would fetch it back word by word. Costs four stack entries. */
  constexpr bool CHAIN = MI_CHAIN && PTDL && !FAST && (MI_CHAIN == 1 || (!MEDIA && !MB && !HERO));    /* shadow ray and extension ray of a vertex in one slice (below; 2: in the plain kernels, next to the exchange) */
  constexpr bool PARK_PS = MI_PARK_PATH && !MB && (FAST ? (PTDL || MI_PARK_PATH == 2) : (JOBS && (PTDL || (HERO && MI_PARK_HERO_PT > 0)) && MI_PARK_ENTRIES_EXACT > 0));
  constexpr int PARK_N = !PARK_PS ? 0 : !FAST ? ((HERO && !PTDL) ? MI_PARK_HERO_PT : MI_PARK_ENTRIES_EXACT) : PTDL ? MI_PARK_ENTRIES : MI_PARK_ENTRIES_PT;   /* 8-byte entries of the column that hold parked path state */
  constexpr int RESULT_SLOTS = (JOBS || FAST) ? 3 : 0;                       /* FAST: the rounds of trace_round_spec (mi_kernels.h), same result slots */
  constexpr int STACK = COLUMN - RESULT_SLOTS - PARK_N;
  static_assert(STACK >= MI_STACK_MIN, "the overflow area is sized for MI_STACK_MIN entries in LDS (mi_abi.hip)");
  static_assert(!(CHAIN && PARK_PS && MEDIA), "a chained lane draws the free-flight distance of its extension ray inside the slice: the generator must not be parked");
  Counters<COUNT || RECORD> cnt;        /* COUNT = false: only the path count (see Counters, mi_kernels.h) */
  typename std::conditional<HERO, PathStateHero, PathState>::type ps;
  ps.active = 0;
  ps.sh_pending = 0;
  bool exhausted = false;
  const unsigned lane = __lane_id();
  TraceState ts;
  ts.done = true;
  ts.time = 0.0f; ts.prims_t1 = MB ? sc.prims_t1 : nullptr;      /* (every lane of a wave works on the leaf jobs of a motion-blur round, also one that has not started a ray yet) */
  Hit hit;
  hit.prim = MI_NOPRIM; hit.dist = FLT_MAX; hit.u = hit.v = 0.0f;
  bool tracing = false, tr_shadow = false;
  constexpr bool SCACHE = MI_SHADOW_CACHE && PTDL && MI_ANYHIT && (!COUNT || MI_SHADOW_CACHE == 2) && !RECORD && !MB && !HERO;
  uint32_t occluder = MI_NOPRIM;         /* SCACHE: the primitive that ended this lane's last occluded shadow ray */

  MI_PHASE_INIT(cnt)
#ifdef MI_PROFILE_TRAV
  cnt.c[31] = (uint32_t)clock64();
#endif
  while(true)
  {
#if MI_SCENE_LAZY
    /* the scene descriptor is the kernel's first by-value argument, i.e. it lives in the kernarg segment. Read through the plain argument,
       every field is loaded at the kernel's entry and stays in a scalar register (or a spill lane) for the whole launch; read through a
       pointer the compiler cannot see through, a field is s_load'ed in the iteration that uses it (lever 1a; A/B in
       profiles/r05_levers_ab.txt) */
    const DScene *sc_lazy;
    {
      typedef __attribute__((address_space(4))) const unsigned char *mi_kernarg_ptr;
      mi_kernarg_ptr k = (mi_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(k));
      sc_lazy = (const DScene *)k;
    }
    const DScene &sc = *sc_lazy;
#endif
    /* ------------------------------------------------------------ refill idle lanes (wave-level compaction of the work queue) */
    if(!exhausted)
    {
      const bool want = !ps.active && !ps.sh_pending;
      const unsigned long long m = __ballot(want);
      /* MI_REFILL_MIN: a few idle lanes wait for company -- path_generate costs the wave the same for 3 lanes as for 60 -- unless the wave
         has little else under way */
      const int rmin = (REGROUP && pool.E) ? MI_REFILL_MIN : 1;         /* (only where the exchange runs: without it waiting is a loss) */
      if(m && (rmin <= 1 || __popcll(m) >= rmin || __popcll(__ballot(tracing || ps.active || ps.sh_pending)) < 64 - 2*rmin))
      {
        const unsigned n = __popcll(m);
        unsigned int base = 0;
        if(lane == (unsigned)(__ffsll((long long)m) - 1)) base = atomicAdd(&blk_next, n);     /* LDS atomic: this block's own range */
        base = __shfl(base, __ffsll((long long)m) - 1);
        if(want)
        {
          const unsigned rank = __popcll(m & ((1ull << lane) - 1ull));
          const unsigned long long i = blk_lo + base + rank;
          if(i < blk_hi)
          {
            if(!RECORD && sc.tile_members)
            { /* tile-owned sharding (mi_render_tiles): the launch's items are pixels of this member's tiles, tile_path() names their paths */
              float px, py;
              const unsigned long long index = tile_path(sc, first + i, px, py);
              if constexpr(HERO) path_generate_hero<RECORD, HALTON, MEDIA>(sc, ps, index, nullptr, 0ull, cnt, px, py);
              else path_generate<RECORD, HALTON, MEDIA>(sc, ps, index, nullptr, cnt, px, py);
            }
            else if constexpr(HERO) path_generate_hero<RECORD, HALTON, MEDIA>(sc, ps, first + i, RECORD ? records + i : nullptr, i, cnt);
            else path_generate<RECORD, HALTON, MEDIA>(sc, ps, first + i, RECORD ? records + i : nullptr, cnt);
          }
          else exhausted = true;
        }
      }
    }
    const bool exhausted_wave = __any(exhausted);   /* this block's index range has run dry */
    if(!__any(ps.active || ps.sh_pending))
    { /* nothing left in this wave -- but vertices other waves posted may still wait in the pools: a wave only leaves when they are empty */
      if(!REGROUP || !pool.E || pool_empty(pool)) break;
    }
    MI_PHASE(cnt, 0)

    /* ------------------------------------------------------------ one ray per busy lane: a pending shadow ray first, else the extension ray */
    if(!tracing && (ps.active || ps.sh_pending))
    {
      tr_shadow = PTDL && ps.sh_pending;
      hit.prim = MI_NOPRIM; hit.dist = tr_shadow ? ps.sh_dist : (MEDIA ? media_free_flight<PTDL, HALTON>(sc, ps) : FLT_MAX); hit.u = hit.v = 0.0f;
      trace_begin(lds, ts, tr_shadow ? ps.sh_dir : ps.dir, cnt);
      if(PTDL && MI_ANYHIT) ts.anyhit = tr_shadow && (ps.sh_light & MI_LIGHT_ANYHIT);
      if(MB) { ts.time = ps.time; ts.prims_t1 = sc.prims_t1; }      /* motion-blurred primitives are tested at the path's time */
      tracing = true;
      if(SCACHE && ts.anyhit && occluder != MI_NOPRIM && occluder != ps.ignore)
      { /* the last occluder first: a triangle or quad that cuts this connection too ends the ray before it starts */
        const PrimRegs rec = prim_load(sc.prims, occluder);
        Hit h;
        h.prim = MI_NOPRIM; h.dist = ps.sh_dist; h.u = h.v = 0.0f;
        if(__float_as_uint(rec.q3.x) >= MI_PRIM_TRI) triquad_intersect(rec, __float_as_uint(rec.q3.x), ray_origin<PTDL>(ps, true), ps.sh_dir, h, occluder);
        if(h.prim != MI_NOPRIM && h.dist < ps.sh_dist) { hit = h; ts.done = true; ts.sp = 0; ts.current = MI_LEAF32; }
      }
    }
    MI_TT(cnt, 5)
    /* ------------------------------------------------------------ a slice of traversal: while-while rounds until only a tail of
       MI_TAIL_LANES rays is still under way. Those lanes keep their traversal state (registers + LDS stack) and go on in the
       next iteration next to the fresh rays of the lanes that shade now, so one long ray does not hold 63 lanes idle. */
    {
      V3 o = ray_origin<PTDL>(ps, tr_shadow), d = tr_shadow ? ps.sh_dir : ps.dir;
      const uint32_t ignore = ps.ignore;     /* the shadow ray of a vertex starts on the same primitive as its extension ray */
      const unsigned tail = exhausted_wave ? 1u : (unsigned)(PTDL ? MI_TAIL_LANES_PTDL : MI_TAIL_LANES);
      lds_uint2 *parked = (lds_uint2 *)lds.stack + (STACK + RESULT_SLOTS)*BLK;
      if(MI_PRIO) __builtin_amdgcn_s_setprio(PTDL ? MI_PRIO_PTDL_TRACE : MI_PRIO_PT_TRACE);
      if(PARK_PS)
      {
        parked[0] = mi_u32x2{(uint32_t)ps.rng.s0, (uint32_t)(ps.rng.s0 >> 32)};
        parked[BLK] = mi_u32x2{(uint32_t)ps.rng.s1, (uint32_t)(ps.rng.s1 >> 32)};
        const unsigned long long pp = (unsigned long long)__double_as_longlong(ps.pdfprod);
        if(PARK_N >= 3) parked[2*BLK] = mi_u32x2{(uint32_t)pp, (uint32_t)(pp >> 32)};
        if(PARK_N >= 4) parked[3*BLK] = mi_u32x2{__float_as_uint(ps.pixel_i), __float_as_uint(ps.pixel_j)};
        if(PARK_N >= 5) parked[4*BLK] = mi_u32x2{__float_as_uint(ps.lambda), __float_as_uint(ps.scramble)};
        if(PARK_N >= 6) parked[5*BLK] = mi_u32x2{__float_as_uint(ps.throughput), __float_as_uint(ps.pdf)};
        if(PARK_N >= 7) parked[6*BLK] = mi_u32x2{__float_as_uint(ps.prev_cos), __float_as_uint(ps.cur_ior)};
        if(PARK_N >= 8) parked[7*BLK] = mi_u32x2{(uint32_t)ps.media.ids, (uint32_t)(ps.media.ids >> 32)};
      }
      while(true)
      {
        if(CHAIN)
        { /* a lane whose shadow ray is through goes straight on with the extension ray of the same vertex (both are known since the
             vertex was shaded, the path state holds both): path_visible's verdict (src/pathspace.c:311-344) is taken now, the splat
             it may lead to is made in front of the next vertex's shading (shadow_splat) -- the order the reference has them in. The
             lane neither idles to the end of the slice nor sits through a shading phase in which it has nothing but that verdict to do. */
          const bool sw = tracing && ts.done && tr_shadow && ps.active;
          if(__any(sw))
          {
            if(sw)
            {
              const bool visible = (hit.dist >= ps.sh_dist) || (hit.prim == MI_NOPRIM) || (hit.prim == (ps.sh_light & ~MI_LIGHT_ANYHIT));
              ps.sh_pending = visible ? 2 : 0;
              tr_shadow = false;
              hit.prim = MI_NOPRIM; hit.dist = MEDIA ? media_free_flight<PTDL, HALTON>(sc, ps) : FLT_MAX; hit.u = hit.v = 0.0f;
              trace_begin(lds, ts, ps.dir, cnt);
              if(MB) { ts.time = ps.time; ts.prims_t1 = sc.prims_t1; }
              d = ps.dir; o = ray_origin<PTDL>(ps, false);
            }
          }
        }
        const bool busy = tracing && !ts.done;
        const unsigned nbusy = __popcll(__ballot(busy));
        if(!nbusy) break;
        if(nbusy < tail && __any(tracing && ts.done)) break;
        if(FAST) { trace_round_spec<BLK, STACK, MB, PTDL && MI_ANYHIT, MI_SPEC_FMA == 2 || (MI_SPEC_FMA == 1 && PTDL)>(lds, sc.prims, o, d, ignore, hit, ts, busy, cnt); continue; }
        if(busy) trace_round<BLK, STACK, MB, PTDL && MI_ANYHIT, JOBS>(lds, sc.prims, o, d, ignore, hit, ts, cnt);
        if(JOBS) leaf_jobs<BLK, STACK, MB, PTDL && MI_ANYHIT>(lds, sc.prims, o, d, ignore, hit, ts, busy, cnt);   /* every lane of the wave takes part */
      }
      if(PARK_PS)
      {
        const mi_u32x2 a = parked[0], b = parked[BLK];
        ps.rng.s0 = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
        ps.rng.s1 = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
        if(PARK_N >= 3) { const mi_u32x2 c = parked[2*BLK]; ps.pdfprod = __longlong_as_double((long long)((unsigned long long)c.x | ((unsigned long long)c.y << 32))); }
        if(PARK_N >= 4) { const mi_u32x2 e = parked[3*BLK]; ps.pixel_i = __uint_as_float(e.x); ps.pixel_j = __uint_as_float(e.y); }
        if(PARK_N >= 5) { const mi_u32x2 f = parked[4*BLK]; ps.lambda = __uint_as_float(f.x); ps.scramble = __uint_as_float(f.y); }
        if(PARK_N >= 6) { const mi_u32x2 f = parked[5*BLK]; ps.throughput = __uint_as_float(f.x); ps.pdf = __uint_as_float(f.y); }
        if(PARK_N >= 7) { const mi_u32x2 f = parked[6*BLK]; ps.prev_cos = __uint_as_float(f.x); ps.cur_ior = __uint_as_float(f.y); }
        if(PARK_N >= 8) { const mi_u32x2 f = parked[7*BLK]; ps.media.ids = (unsigned long long)f.x | ((unsigned long long)f.y << 32); }
      }
    }
    if(MI_PRIO) __builtin_amdgcn_s_setprio(PTDL ? MI_PRIO_PTDL_SHADE : MI_PRIO_PT_SHADE);
#ifdef MI_EXP_BARRIER      /* experiment 0 of the regrouping work: what lock step of the sixteen waves costs by itself (cfg 2 +30 %, cfg 3 +21 %) */
    __builtin_amdgcn_s_barrier();
#endif
    MI_PHASE(cnt, 1)
    MI_TT(cnt, 4)
    /* The lanes of the tail keep their ray. Its traversal state (closest hit so far, node, stack pointer: 6 dwords; the three 1/dir
       are formed again) would sit in eleven registers through the shading of the other lanes, where the kernel's register pressure
       peaks -- between two rounds the three result slots of the lane's LDS column (leaf phase) are free and take it instead. */
    constexpr bool PARK = MI_PARK_TRACE && (MI_PARK_TRACE == 2 ? (FAST || JOBS) : MI_PARK_TRACE == 3 ? ((FAST || JOBS) && PTDL) : (FAST && PTDL)) && !MB;
    const bool keep = tracing && !ts.done;
    if(PARK && keep)
    {
      lds_uint2 *col = (lds_uint2 *)lds.stack + STACK*BLK;
      col[0] = mi_u32x2{hit.prim, __float_as_uint(hit.dist)};
      col[BLK] = mi_u32x2{__float_as_uint(hit.u), __float_as_uint(hit.v)};
      col[2*BLK] = mi_u32x2{ts.current, (uint32_t)ts.sp | (ts.anyhit ? 0x10000u : 0u)};
    }
    SplatReq splat;
    splat.pending = false; splat.c0 = splat.c1 = splat.c2 = 0.0f;
    if(REGROUP)
    { /* material queues (mi_regroup.h). What needs no material is done where the ray ended: the verdict of a shadow ray, the end of a
         path that left the scene. Then the wave trades surface vertices with the pools, and shades what it holds afterwards. */
      if(CHAIN)
      { /* a chained lane's connection is splatted in front of the next vertex's shading -- here that means in front of the exchange, which may hand the vertex
           (and with it the lane's pixel) to another wave */
        const bool owes = tracing && ts.done && !tr_shadow && ps.sh_pending == 2;
        if(__any(owes))
        {
          SplatReq cs;
          cs.pending = false; cs.c0 = cs.c1 = cs.c2 = 0.0f;
          if(owes) shadow_splat<RECORD>(sc, ps, RECORD ? records + (ps.index - first) : nullptr, cnt, cs);
          if(!RECORD) splat_wave(sc, cs.pending, ps.pixel_i, ps.pixel_j, cs.c0, cs.c1, cs.c2);
        }
      }
      const bool fin = tracing && ts.done;
      uint32_t cls = 0u;
      /* (extended kernels: an extension ray that ended at its sampled free-flight distance has a volume vertex there) */
      const bool volume = MEDIA && hit.prim == MI_NOPRIM && ps.clip < FLT_MAX;
      const bool surf0 = fin && !tr_shadow && (hit.prim != MI_NOPRIM || volume);
      if(surf0 && pool.E) cls = volume ? sc.pool_volume_class : pool_class_of(pool, sc, hit.prim);
#if MI_REGROUP_EARLY_SHADOW
      if(fin && (tr_shadow || (hit.prim == MI_NOPRIM && !volume)))
      {
        tracing = false;
        mi_path_record *rec = RECORD ? records + (ps.index - first) : nullptr;
        if(tr_shadow) { if(SCACHE && hit.prim != MI_NOPRIM && hit.dist < ps.sh_dist && hit.prim != (ps.sh_light & ~MI_LIGHT_ANYHIT)) occluder = hit.prim; shadow_resolve<RECORD>(sc, ps, hit, rec, cnt, splat); }
        else path_escape<RECORD, MEDIA>(sc, ps, rec, cnt);
      }
#else
      /* (the verdict of a finished shadow ray stays where it was, next to path_shade: resolving it first made lanes free that
         almost always owe a splat, at the price of a third divergent region -- cfg 3 +4 % without the exchange) */
      if(fin && !tr_shadow && hit.prim == MI_NOPRIM && !volume)
      {
        tracing = false;
        if constexpr(HERO) path_shade_hero<RECORD, PTDL, HALTON, MEDIA, MB>(sc, ps, hit, shape_material, shape_L, RECORD ? records + (ps.index - first) : nullptr, RECORD ? ps.index - first : 0ull, cnt, splat);
        else path_escape<RECORD, MEDIA>(sc, ps, RECORD ? records + (ps.index - first) : nullptr, cnt);
      }
#endif
      /* a lane that still owes this iteration a splat keeps its pixel: it is free from the next iteration on */
      const bool freelane = !tracing && !ps.active && !ps.sh_pending && !splat.pending;
      regroup_exchange<RECORD, PTDL, HALTON, MEDIA, MI_PRIO ? (PTDL ? MI_PRIO_PTDL_SHADE : MI_PRIO_PT_SHADE) : 0, HERO>(pool, ps, hit, ts, tracing, tr_shadow, surf0, cls, freelane, exhausted_wave, cnt);
      MI_TT(cnt, 4)       /* (trav probe: part 4 = the end of the slice + the exchange between waves) */
      if(tracing && ts.done)
      {
        tracing = false;
        mi_path_record *rec = RECORD ? records + (ps.index - first) : nullptr;
        if constexpr(HERO)
        {
          const unsigned long long slot = RECORD ? ps.index - first : 0ull;
          if(PTDL && tr_shadow) shadow_resolve_hero<RECORD>(sc, ps, hit, rec, slot, cnt, splat);
          else
          {
            if(!MEDIA) __builtin_assume(hit.prim != MI_NOPRIM);
            path_shade_hero<RECORD, PTDL, HALTON, MEDIA, MB>(sc, ps, hit, shape_material, shape_L, rec, slot, cnt, splat);
          }
        }
        else if(PTDL && !MI_REGROUP_EARLY_SHADOW && tr_shadow) { if(SCACHE && hit.prim != MI_NOPRIM && hit.dist < ps.sh_dist && hit.prim != (ps.sh_light & ~MI_LIGHT_ANYHIT)) occluder = hit.prim; shadow_resolve<RECORD>(sc, ps, hit, rec, cnt, splat); }
        else
        {
          if(!MEDIA) __builtin_assume(hit.prim != MI_NOPRIM);     /* paths that left the scene have ended above */
          path_shade<RECORD, PTDL, HALTON, MEDIA, MB>(sc, ps, hit, shape_material, shape_L, rec, cnt, splat);
        }
      }
    }
    else if(tracing && ts.done)
    {
      tracing = false;
      mi_path_record *rec = RECORD ? records + (ps.index - first) : nullptr;
      if constexpr(HERO)
      {
        const unsigned long long slot = RECORD ? ps.index - first : 0ull;
        if(tr_shadow) shadow_resolve_hero<RECORD>(sc, ps, hit, rec, slot, cnt, splat);
        else path_shade_hero<RECORD, PTDL, HALTON, MEDIA, MB>(sc, ps, hit, shape_material, shape_L, rec, slot, cnt, splat);
      }
      else if(tr_shadow) { if(SCACHE && hit.prim != MI_NOPRIM && hit.dist < ps.sh_dist && hit.prim != (ps.sh_light & ~MI_LIGHT_ANYHIT)) occluder = hit.prim; shadow_resolve<RECORD>(sc, ps, hit, rec, cnt, splat); }
      else
      {
        if(CHAIN && ps.sh_pending == 2) shadow_splat<RECORD>(sc, ps, rec, cnt, splat);      /* the connection made at the vertex this ray left */
        path_shade<RECORD, PTDL, HALTON, MEDIA, MB>(sc, ps, hit, shape_material, shape_L, rec, cnt, splat);
      }
    }

    if(PARK)
    {
      if(keep)
      {
        const lds_uint2 *col = (const lds_uint2 *)lds.stack + STACK*BLK;
        const mi_u32x2 a = col[0], b = col[BLK], c = col[2*BLK];
        hit.prim = a.x; hit.dist = __uint_as_float(a.y); hit.u = __uint_as_float(b.x); hit.v = __uint_as_float(b.y);
        ts.current = c.x; ts.sp = (int)(c.y & 0xffffu); ts.anyhit = (c.y & 0x10000u) != 0; ts.done = false;
        const V3 d = tr_shadow ? ps.sh_dir : ps.dir;
        ts.idx = mi_rcp(d.x); ts.idy = mi_rcp(d.y); ts.idz = mi_rcp(d.z);
      }
      else
      {
        hit.prim = MI_NOPRIM; hit.dist = FLT_MAX; hit.u = hit.v = 0.0f;
        ts.current = MI_LEAF32; ts.sp = 0; ts.anyhit = false; ts.done = true; ts.idx = ts.idy = ts.idz = 0.0f;
      }
    }
    /* ------------------------------------------------------------ splats of this iteration, cooperatively */
    MI_PHASE(cnt, 5)
    MI_TT(cnt, 5)
    if(!RECORD) splat_wave(sc, splat.pending, ps.pixel_i, ps.pixel_j, splat.c0, splat.c1, splat.c2);
    MI_PHASE(cnt, 6)
    MI_TT(cnt, 6)
#ifdef MI_PROFILE_TRAV
    cnt.c[15]++;            /* wave iterations */
#endif
#ifdef MI_PROFILE_POOL
    { const unsigned long long t_now = clock64(); MI_POOLSTAT(cnt, 7, 0, 1) cnt.c[30] = (uint32_t)(t_now >> 4); }   /* wave iterations (a maximum over the workgroups) */
#endif
  }

#ifdef MI_PROFILE_LOOPS    /* development build: box hits / splats / vertices become wave-level inner iterations / leaf slots / analytic passes */
  cnt.c[2] = cnt.c[8]; cnt.c[5] = cnt.c[9]; cnt.c[6] = cnt.c[10];
#endif
#ifdef MI_PROFILE_TRAV     /* development build: counters 0..6 become lane 0's ticks per part of the wave iteration (tools/trav_probe.py); 7 stays */
  for(int k=0;k<8;k++) cnt.c[k] = lane ? 0u : cnt.c[8 + k];
#endif
#if defined(MI_PROFILE_BLOCKS) || defined(MI_PROFILE_POOL)
  unsigned long long block_out[8];
  for(int k=0;k<8;k++) block_out[k] = (unsigned long long)cnt.c[8 + k] | ((unsigned long long)cnt.c[16 + k] << 36);
#endif
#ifdef MI_PROFILE_PHASES   /* development build: the 8 counters become lane 0's phase ticks | occurrences << 36 (tools/phase_probe.py) */
  unsigned long long phase_out[8];
  for(int k=0;k<8;k++) phase_out[k] = lane ? 0ull : ((unsigned long long)cnt.c[8 + k] | ((unsigned long long)cnt.c[16 + k] << 36));
#endif
  unsigned long long *shard = sc.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8;
#if !defined(MI_PROFILE_PHASES) && !defined(MI_PROFILE_TRAV) && !defined(MI_PROFILE_BLOCKS) && !defined(MI_PROFILE_POOL)
  if(cnt.on) atomicMax(shard + 7, (unsigned long long)cnt.c[7]);     /* deepest traversal stack use */
#endif
  /* ------------------------------------------------------------ flush work counters: wave reduction, one atomic per wave */
#pragma unroll
  for(int k=0;k<8;k++)
  {
#if !defined(MI_PROFILE_TRAV) && !defined(MI_PROFILE_BLOCKS) && !defined(MI_PROFILE_POOL)
    if(!cnt.on && k != 4) continue;         /* the plain kernels only count paths */
#endif
    unsigned long long c = cnt.c[k];
#if defined(MI_PROFILE_BLOCKS) || defined(MI_PROFILE_POOL)
    c = block_out[k];
#endif
#ifdef MI_PROFILE_PHASES
    c = phase_out[k];
    if(k == 7) { if(lane == 0 && c) atomicAdd(shard + 7, c); continue; }
#endif
    for(int off=32;off>0;off>>=1) c += __shfl_down(c, off);
#if !defined(MI_PROFILE_TRAV) && !defined(MI_PROFILE_BLOCKS) && !defined(MI_PROFILE_POOL)
    if(k == 7) continue;                    /* slot 7 is a maximum, flushed above */
#endif
    if(lane == 0 && c) atomicAdd(shard + k, c);
  }
}

/* ======================================================================================= unit hook: rays in, hits out */
template<bool NODES_LDS, bool FAST>
__global__ __launch_bounds__(MI_BLOCK) void mi_intersect_kernel(DScene sc, const mi_ray *rays, unsigned long long n, mi_hit *out,
                                                                uint2 *stack_overflow)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds lds = lds_setup<MI_BLOCK, NODES_LDS>(sc, smem, stack_overflow);
  Counters<true> cnt;
  for(unsigned long long base=(unsigned long long)blockIdx.x*MI_BLOCK; base<n; base+=(unsigned long long)gridDim.x*MI_BLOCK)
  {
    const unsigned long long i = base + threadIdx.x;
    const bool live = i < n;                     /* the lanes behind the last ray still take part in the traversal's wave-level steps */
    const mi_ray r = rays[live ? i : n - 1];
    Hit hit;
    hit.prim = MI_NOPRIM; hit.dist = r.max_dist; hit.u = hit.v = 0.0f;
    accel_intersect<MI_BLOCK, MI_STACK, FAST>(lds, sc.prims, mk3(r.pos[0], r.pos[1], r.pos[2]), mk3(r.dir[0], r.dir[1], r.dir[2]), r.ignore, hit, cnt, live);
    if(live)
    {
      mi_hit h;
      h.prim = hit.prim; h.primid = hit.prim == MI_NOPRIM ? MI_PRIMID_INVALID : MI_GEO_PRIMID(sc.primgeo[hit.prim]);
      h.dist = hit.dist; h.u = hit.u; h.v = hit.v; h.pad[0] = h.pad[1] = 0;
      out[i] = h;
    }
  }
  unsigned long long *shard = sc.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8;
  atomicMax(shard + 7, (unsigned long long)cnt.c[7]);
  for(int k=0;k<4;k++) if(cnt.c[k]) atomicAdd(shard + k, (unsigned long long)cnt.c[k]);
}


/* ---------------------------------------------------------------------------------------- kernel table
 * A part = one (PTDL, MEDIA, MB, FAST, NORG, HERO); inside it `which` selects bit 0 RECORD, 1 NODES_LDS, 2 HALTON, 3 COUNT. MB implies MEDIA and
 * has no FAST rounds (its leaf phase stays per lane); the RECORD kernels always count.
 * MI_DEV_FAST (development builds, tools/variants.sh): only the plain tree-in-LDS kernels (2: with the Halton ones, 3: with the extended kernels' exact rounds) -- the other
 * parts compile to stubs. L = NULL: return the kernel's address without launching (hipFuncSetAttribute). */
struct PathLaunch
{
  DScene d;
  int grid;
  size_t lds_bytes;
  hipStream_t stream;
  unsigned long long first, n;
  const uint32_t *shape_material;
  const float *shape_L;
  mi_path_record *rec;
  uint2 *overflow;
  uint2 *table;               /* the wavefront kernel's path tables (mi_wavefront.h), one per workgroup */
  uint32_t table_entries;
};
#define MI_WHICH_RECORD 1u
#define MI_WHICH_NODES_LDS 2u
#define MI_WHICH_HALTON 4u
#define MI_WHICH_COUNT 8u

template<bool PTDL, bool MEDIA, bool MB, bool FAST, bool NORG = false, bool HERO = false> const void *mi_path_part(unsigned which, const PathLaunch *L);

static inline bool mi_path_which_valid(unsigned which)
{
  if((which & MI_WHICH_RECORD) && !(which & MI_WHICH_COUNT)) return false;
#ifdef MI_DEV_FAST
  if(!(which & MI_WHICH_NODES_LDS) || ((which & MI_WHICH_HALTON) && MI_DEV_FAST != 2)) return false;
#endif
  return true;
}

#ifdef MI_PART_DEFINE
template<bool PTDL, bool MEDIA, bool MB, bool FAST, bool NORG, bool HERO, bool R, bool N, bool H, bool C> static const void *mi_path_go(const PathLaunch *L)
{
  constexpr bool valid = !(MB && !MEDIA) && !(MB && FAST) && !(R && !C) && !(NORG && (!MEDIA || MB)) && !(HERO && FAST)
#ifdef MI_DEV_FAST
                         && (!H || MI_DEV_FAST == 2) && (!MEDIA || (MI_DEV_FAST == 3 && !FAST && !NORG && !HERO)) && !MB && N
#endif
                         ;
  if constexpr(valid)
  {
    if(L) hipLaunchKernelGGL((mi_path_kernel<R, PTDL, N, H, MEDIA, MB, C, FAST, NORG, HERO>), dim3(L->grid), dim3(MI_BLOCK_OF(HERO, PTDL, MEDIA)), L->lds_bytes, L->stream, L->d, L->first, L->n,
                             L->shape_material, L->shape_L, L->rec, L->overflow);
    return (const void *)mi_path_kernel<R, PTDL, N, H, MEDIA, MB, C, FAST, NORG, HERO>;
  }
  else { fprintf(stderr, "[mi] internal: kernel variant not built\n"); abort(); }
}
template<bool PTDL, bool MEDIA, bool MB, bool FAST, bool NORG, bool HERO, int LEFT, bool... B> static const void *mi_path_pick(unsigned which, const PathLaunch *L)
{
  if constexpr(LEFT == 0) return mi_path_go<PTDL, MEDIA, MB, FAST, NORG, HERO, B...>(L);
  else return (which & 1u) ? mi_path_pick<PTDL, MEDIA, MB, FAST, NORG, HERO, LEFT - 1, B..., true>(which >> 1, L)
                           : mi_path_pick<PTDL, MEDIA, MB, FAST, NORG, HERO, LEFT - 1, B..., false>(which >> 1, L);
}
template<bool PTDL, bool MEDIA, bool MB, bool FAST, bool NORG, bool HERO> const void *mi_path_part(unsigned which, const PathLaunch *L)
{
  return mi_path_pick<PTDL, MEDIA, MB, FAST, NORG, HERO, 4>(which, L);
}
#endif

#endif
