/* mi_wavefront.h -- the same pt/ptdl path logic as the persistent megakernel (mi_path.h), organised as a
 * two-kernel wavefront pipeline over a pool of P path slots that live in HBM:
 *
 *   wf_logic   one thread per slot, slot == thread (structure-of-arrays state => every load/store is coalesced):
 *              resolve the pending shadow ray, shade the hit of the pending extension ray (path_shade), re-fill dead
 *              slots with new camera paths (path_generate), leave the next rays in the slot.
 *   wf_trace   persistent workgroups with the BVH in LDS; lanes pull ray jobs from a queue counter and re-fill
 *              themselves as soon as their ray is finished (ballot + prefix rank, one atomic per wave), so a wave does
 *              not wait for its longest ray. Needs ~64 VGPRs instead of the megakernel's 128+spills.
 *
 * Why: in the megakernel a wave executes 13 k VALU instructions per ray-iteration for ~2.5 k useful ones per lane
 * (profiles/r01_pmc_summary.json: 21 % lane utilisation) because every lane carries exactly one ray through
 * traversal AND shading. Splitting the two removes the coupling; the price is path state traffic
 * (~45 dwords read + written per ray, coalesced).
 *
 * Status (round 1, measured on MI355X, cfg 2): 1029 Msamples/s against 1213 for the megakernel -- wf_trace alone costs
 * what the whole megakernel iteration costs (traversal is ~56 % VALU-issue bound in both), so the split only adds the
 * state traffic. It is kept as an alternative organisation (CORONA_MI_MODE=wave) with identical results (same parity
 * tests), and as the vehicle for per-material shading queues / ray sorting in later rounds.
 */
#ifndef MI_WAVEFRONT_H
#define MI_WAVEFRONT_H

#include "mi_path.h"

enum
{
  F_ORGX, F_ORGY, F_ORGZ, F_DIRX, F_DIRY, F_DIRZ, F_IGNORE,
  F_PREVCOS, F_PREVTHR, F_THR, F_PDF, F_PDFPROD_LO, F_PDFPROD_HI, F_IOR, F_MEDIA_LO, F_MEDIA_HI, F_MEDIA_CNT,
  F_LAMBDA, F_PIXI, F_PIXJ, F_SCRAMBLE, F_LENGTH, F_RNG0, F_RNG1, F_RNG2, F_RNG3, F_FLAGS, F_PMM,
  F_SHOX, F_SHOY, F_SHOZ, F_SHDX, F_SHDY, F_SHDZ, F_SHDIST, F_SHVALUE, F_SHLIGHT, F_SHIGNORE,
  F_HITPRIM, F_HITDIST, F_HITU, F_HITV, F_SHHITPRIM, F_SHHITDIST,
  F_COUNT
};
#define WF_LIVE_SHARDS 256
#define WF_ACTIVE 1u
#define WF_SHADOW 2u

struct WFPool
{
  uint32_t *s;                      /* [P/64][F_COUNT][64] */
  uint32_t P;
  unsigned long long *trace_head;   /* next ray job */
  unsigned long long *live;         /* [WF_LIVE_SHARDS] set to 1 by workgroups that leave rays pending after wf_logic */
};

/* AoSoA: tiles of 64 slots, [tile][field][lane]. A wave's 64 slots are one contiguous 11 KB block (one or two pages)
 * while every field access is still a coalesced 256-B row. A plain [field][P] layout touches F_COUNT pages 8 MB
 * apart per wave and is TLB-bound (measured: wf_logic 2.4x slower, and slower the larger the pool). */
#define WF(f) pool.s[((size_t)(slot >> 6)*F_COUNT + (f))*64 + (slot & 63u)]
#define WFF(f) __uint_as_float(WF(f))

#define WF_LOGIC_BLOCK 256
#ifndef WF_REFILL_AT
#define WF_REFILL_AT 48            /* re-fill a wave's idle lanes once fewer than this many rays are still in flight */
#endif
#define WF_CHUNK 512              /* ray jobs a wave claims with one global atomic */

template<bool PTDL>
__global__ __launch_bounds__(WF_LOGIC_BLOCK) void wf_logic(DScene sc, WFPool pool, unsigned long long first, unsigned long long count,
                                                const uint32_t *shape_material, const float *shape_L)
{
  __shared__ unsigned int blk_want;
  __shared__ unsigned long long blk_base;
  const uint32_t slot = blockIdx.x*WF_LOGIC_BLOCK + threadIdx.x;
  const bool inrange = slot < pool.P;
  const unsigned lane = __lane_id();
  uint32_t cnt[MI_CNT] = {0};
  PathState ps;
  ps.active = 0; ps.sh_pending = 0;
  ps.pixel_i = ps.pixel_j = 0.0f; ps.lambda = 400.0f;
  uint32_t flags = 0;
  if(inrange) flags = WF(F_FLAGS);
  if(flags)
  {
    ps.lambda = WFF(F_LAMBDA); ps.pixel_i = WFF(F_PIXI); ps.pixel_j = WFF(F_PIXJ); ps.scramble = WFF(F_SCRAMBLE);
    ps.length = (int)WF(F_LENGTH);
    ps.rng.s0 = (unsigned long long)WF(F_RNG0) | ((unsigned long long)WF(F_RNG1) << 32);
    ps.rng.s1 = (unsigned long long)WF(F_RNG2) | ((unsigned long long)WF(F_RNG3) << 32);
    ps.active = (flags & WF_ACTIVE) ? 1 : 0;
    ps.sh_pending = (flags & WF_SHADOW) ? 1 : 0;
  }
  ps.index = 0;

  /* ---- pending next-event connection: path_visible's verdict */
  SplatReq splat;
  splat.pending = false; splat.c0 = splat.c1 = splat.c2 = 0.0f;
  if(PTDL && ps.sh_pending)
  {
    ps.sh_dist = WFF(F_SHDIST); ps.sh_value = WFF(F_SHVALUE); ps.sh_light = WF(F_SHLIGHT); ps.sh_length = ps.length + 1;
    Hit sh;
    sh.prim = WF(F_SHHITPRIM); sh.dist = WFF(F_SHHITDIST); sh.u = sh.v = 0.0f;
    shadow_resolve<false>(sc, ps, sh, nullptr, cnt, splat);
  }
  if(PTDL) splat_wave(sc, splat.pending, ps.pixel_i, ps.pixel_j, splat.c0, splat.c1, splat.c2);

  /* ---- the extension ray's hit: finish vertex v, sample the next ray */
  splat.pending = false;
  if(ps.active)
  {
    ps.org = mk3(WFF(F_ORGX), WFF(F_ORGY), WFF(F_ORGZ));
    ps.dir = mk3(WFF(F_DIRX), WFF(F_DIRY), WFF(F_DIRZ));
    ps.ignore = WF(F_IGNORE);
    ps.prev_cos = WFF(F_PREVCOS); ps.prev_throughput = WFF(F_PREVTHR); ps.throughput = WFF(F_THR); ps.pdf = WFF(F_PDF);
    ps.pdfprod = __hiloint2double((int)WF(F_PDFPROD_HI), (int)WF(F_PDFPROD_LO));
    ps.cur_ior = WFF(F_IOR);
    ps.media.ids = (unsigned long long)WF(F_MEDIA_LO) | ((unsigned long long)WF(F_MEDIA_HI) << 32);
    { const uint32_t mc = WF(F_MEDIA_CNT); ps.media.count = mc & 0xffu; ps.media.broken = mc >> 8; }
    ps.prev_material_modes = WF(F_PMM);
    ps.prev_mode = 0; ps.prev_x = mk3(0, 0, 0);
    Hit hit;
    hit.prim = WF(F_HITPRIM); hit.dist = WFF(F_HITDIST); hit.u = WFF(F_HITU); hit.v = WFF(F_HITV);
    path_shade<false, PTDL, false>(sc, ps, hit, shape_material, shape_L, nullptr, cnt, splat);
  }
  splat_wave(sc, splat.pending, ps.pixel_i, ps.pixel_j, splat.c0, splat.c1, splat.c2);

  /* ---- re-fill dead slots: ONE global atomic per workgroup (same-address atomics serialise), ranks via LDS + ballot */
  {
    if(threadIdx.x == 0) blk_want = 0;
    __syncthreads();
    const bool want = inrange && !ps.active && !ps.sh_pending;
    const unsigned long long m = __ballot(want);
    const int leader = m ? __ffsll((long long)m) - 1 : 0;
    unsigned int wave_off = 0;
    if(m && (int)lane == leader) wave_off = atomicAdd(&blk_want, (unsigned int)__popcll(m));
    wave_off = __shfl(wave_off, leader);
    __syncthreads();
    /* this workgroup's own part of the index range and its own progress counter: an uncontended atomic */
    const unsigned long long nb = gridDim.x;
    const unsigned long long lo = count/nb*blockIdx.x + (blockIdx.x < count%nb ? blockIdx.x : count%nb);
    const unsigned long long hi = lo + count/nb + (blockIdx.x < count%nb ? 1 : 0);
    if(threadIdx.x == 0)
    {
      unsigned long long base = hi;
      if(blk_want)
      {
        base = lo + *(volatile unsigned long long *)(sc.work + blockIdx.x);
        if(base < hi) base = lo + atomicAdd(sc.work + blockIdx.x, (unsigned long long)blk_want);
      }
      blk_base = base;
    }
    __syncthreads();
    if(want)
    {
      const unsigned long long i = blk_base + wave_off + __popcll(m & ((1ull << lane) - 1ull));
      if(i < hi)
      {
        path_generate<false, false>(sc, ps, first + i, nullptr, cnt);
      }
    }
  }

  /* ---- store the slot */
  const uint32_t nflags = (ps.active ? WF_ACTIVE : 0u) | (ps.sh_pending ? WF_SHADOW : 0u);
  if(inrange && (flags | nflags))
  {
    WF(F_FLAGS) = nflags;
    if(nflags)
    {
      WF(F_LAMBDA) = __float_as_uint(ps.lambda); WF(F_PIXI) = __float_as_uint(ps.pixel_i); WF(F_PIXJ) = __float_as_uint(ps.pixel_j);
      WF(F_SCRAMBLE) = __float_as_uint(ps.scramble); WF(F_LENGTH) = (uint32_t)ps.length;
      WF(F_RNG0) = (uint32_t)ps.rng.s0; WF(F_RNG1) = (uint32_t)(ps.rng.s0 >> 32);
      WF(F_RNG2) = (uint32_t)ps.rng.s1; WF(F_RNG3) = (uint32_t)(ps.rng.s1 >> 32);
    }
    if(ps.active)
    {
      WF(F_ORGX) = __float_as_uint(ps.org.x); WF(F_ORGY) = __float_as_uint(ps.org.y); WF(F_ORGZ) = __float_as_uint(ps.org.z);
      WF(F_DIRX) = __float_as_uint(ps.dir.x); WF(F_DIRY) = __float_as_uint(ps.dir.y); WF(F_DIRZ) = __float_as_uint(ps.dir.z);
      WF(F_IGNORE) = ps.ignore;
      WF(F_PREVCOS) = __float_as_uint(ps.prev_cos); WF(F_PREVTHR) = __float_as_uint(ps.prev_throughput);
      WF(F_THR) = __float_as_uint(ps.throughput); WF(F_PDF) = __float_as_uint(ps.pdf);
      WF(F_PDFPROD_LO) = (uint32_t)__double2loint(ps.pdfprod); WF(F_PDFPROD_HI) = (uint32_t)__double2hiint(ps.pdfprod);
      WF(F_IOR) = __float_as_uint(ps.cur_ior);
      WF(F_MEDIA_LO) = (uint32_t)ps.media.ids; WF(F_MEDIA_HI) = (uint32_t)(ps.media.ids >> 32);
      WF(F_MEDIA_CNT) = ps.media.count | (ps.media.broken << 8);
      WF(F_PMM) = ps.prev_material_modes;
    }
    if(PTDL && ps.sh_pending)
    {
      WF(F_SHOX) = __float_as_uint(ps.sh_org.x); WF(F_SHOY) = __float_as_uint(ps.sh_org.y); WF(F_SHOZ) = __float_as_uint(ps.sh_org.z);
      WF(F_SHDX) = __float_as_uint(ps.sh_dir.x); WF(F_SHDY) = __float_as_uint(ps.sh_dir.y); WF(F_SHDZ) = __float_as_uint(ps.sh_dir.z);
      WF(F_SHDIST) = __float_as_uint(ps.sh_dist); WF(F_SHVALUE) = __float_as_uint(ps.sh_value);
      WF(F_SHLIGHT) = ps.sh_light; WF(F_SHIGNORE) = ps.sh_ignore;
    }
  }
  /* ---- rays pending for wf_trace: a flag per workgroup shard (same-address stores serialise like atomics) + work counters */
  if(__syncthreads_or(nflags != 0) && threadIdx.x == 0) *(volatile unsigned long long *)(pool.live + (blockIdx.x % WF_LIVE_SHARDS)) = 1ull;
  unsigned long long *shard = sc.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8;
#pragma unroll
  for(int k=4;k<7;k++)
  {
    unsigned long long c = cnt[k];
    for(int off=32;off>0;off>>=1) c += __shfl_down(c, off);
    if(lane == 0 && c) atomicAdd(shard + k, c);
  }
}

/* ---------------------------------------------------------------------------------------------------- trace */
template<int BLOCK, int STACK, bool PTDL, bool NODES_LDS>
__global__ __launch_bounds__(BLOCK) void wf_trace(DScene sc, WFPool pool, uint2 *stack_overflow)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds lds = lds_setup<BLOCK, NODES_LDS>(sc, smem, stack_overflow);

  const unsigned lane = __lane_id();
  const unsigned long long njobs = PTDL ? 2ull*pool.P : pool.P;
  uint32_t cnt[MI_CNT] = {0};
  bool busy = false, exhausted = false;
  unsigned long long next = 0, chunk_end = 0;        /* wave-uniform: this wave's claimed part of the job range */
  uint32_t slot = 0, is_shadow = 0;
  TraceState ts;
  ts.done = true; ts.sp = 0; ts.current = MI_LEAF32; ts.idx = ts.idy = ts.idz = 1.0f;
  V3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
  uint32_t ignore = MI_NOPRIM;
  Hit hit;
  hit.prim = MI_NOPRIM; hit.dist = 0.0f; hit.u = hit.v = 0.0f;

  while(true)
  {
    /* retire finished rays */
    if(busy && ts.done)
    {
      if(is_shadow) { WF(F_SHHITPRIM) = hit.prim; WF(F_SHHITDIST) = __float_as_uint(hit.dist); }
      else { WF(F_HITPRIM) = hit.prim; WF(F_HITDIST) = __float_as_uint(hit.dist); WF(F_HITU) = __float_as_uint(hit.u); WF(F_HITV) = __float_as_uint(hit.v); }
      busy = false;
    }
    /* re-fill idle lanes from this wave's chunk of the job range; a new chunk costs one global atomic per WF_CHUNK jobs.
       Up to four pulls so that empty slots (dead paths) do not starve the wave. */
    for(int pull=0;pull<4 && !exhausted;pull++)
    {
      const unsigned long long m = __ballot(!busy);
      if(!m) break;
      if(next >= chunk_end)
      {
        unsigned long long base = 0;
        if(lane == 0) base = atomicAdd(pool.trace_head, (unsigned long long)WF_CHUNK);
        base = __shfl(base, 0);
        if(base >= njobs) { exhausted = true; break; }
        next = base; chunk_end = base + WF_CHUNK < njobs ? base + WF_CHUNK : njobs;
      }
      const unsigned long long avail = chunk_end - next;
      const unsigned rank = __popcll(m & ((1ull << lane) - 1ull));
      const unsigned take = (unsigned)(__popcll(m) < avail ? __popcll(m) : avail);
      if(!busy && rank < take)
      {
        const unsigned long long j = next + rank;
        is_shadow = (PTDL && j >= pool.P) ? 1u : 0u;
        slot = (uint32_t)(is_shadow ? j - pool.P : j);
        const uint32_t flags = WF(F_FLAGS);
        if(flags & (is_shadow ? WF_SHADOW : WF_ACTIVE))
        {
          if(is_shadow)
          {
            o = mk3(WFF(F_SHOX), WFF(F_SHOY), WFF(F_SHOZ)); d = mk3(WFF(F_SHDX), WFF(F_SHDY), WFF(F_SHDZ));
            ignore = WF(F_SHIGNORE); hit.dist = WFF(F_SHDIST);
          }
          else
          {
            o = mk3(WFF(F_ORGX), WFF(F_ORGY), WFF(F_ORGZ)); d = mk3(WFF(F_DIRX), WFF(F_DIRY), WFF(F_DIRZ));
            ignore = WF(F_IGNORE); hit.dist = FLT_MAX;
          }
          hit.prim = MI_NOPRIM; hit.u = hit.v = 0.0f;
          trace_begin(ts, d, cnt);
          busy = true;
        }
      }
      next += take;
    }
    const unsigned long long bm = __ballot(busy);
    if(!bm) { if(exhausted) break; else continue; }
    /* traverse until a quarter of the wave is idle again (or, once the queue is dry, until everybody is done) */
    const int refill_at = exhausted ? 0 : WF_REFILL_AT;
    do
    {
      trace_round<BLOCK, STACK>(lds, sc.prims, o, d, ignore, hit, ts, cnt);
    }
    while(__popcll(__ballot(busy && !ts.done)) > refill_at);
  }
  unsigned long long *shard = sc.counters + (size_t)(blockIdx.x % MI_COUNTER_SHARDS)*8;
  atomicMax(shard + 7, (unsigned long long)cnt[7]);
#ifdef MI_PROFILE_LOOPS
  const int nflush = 7;
#else
  const int nflush = 4;
#endif
#pragma unroll
  for(int k=0;k<nflush;k++)
  {
    unsigned long long c = cnt[k];
    for(int off=32;off>0;off>>=1) c += __shfl_down(c, off);
    if(lane == 0 && c) atomicAdd(shard + k, c);
  }
}

#undef WF
#undef WFF
#endif
