/* oracle_halton.c -- TEST INFRASTRUCTURE: CPU restatement of the reference's Halton point sampler
 * (SURVEY 8(f) row 2): src/pointsampler.d/halton.c:69-84 on top of the vendored sampler ext/halton/halton.h
 * (Leonhard Gruenschloss' generated Halton sampler, 256 dimensions = the first 256 primes, digit permutations drawn by
 * halton_init_random(frame), ext/halton/halton.h:3244-3274; the generator that fixes the per-base constants is
 * ext/halton/halton_gen.py:101-141).
 *
 * Per dimension d > 0 with base b: P = the largest power of b that is <= 500 (`digits` digits looked up at once),
 * M = the largest power of P below 2^32 (G = log_P M table look-ups), value = (sum_k perm[(index / P^k) % P] * P^(G-1-k))
 * * (float)(0x1.fffffcp-1 / M), all in 32-bit unsigned arithmetic; dimension 0 is the bit reversal.
 * The permutations come from srand48(seed) / lrand48(): X' = (0x5DEECE66D X + 0xB) mod 2^48, X0 = seed << 16 | 0x330E,
 * lrand48 = X >> 17 (POSIX; restated here so that the tables do not depend on the C library, tests compare with libc).
 */
#include "o_core.h"
#include <pthread.h>
#include <stdlib.h>

#define O_HALTON_DIMS 256
#define O_HALTON_MAX_BASE 1619

typedef struct o_halton_dim { uint32_t base, P, groups, offset; float scale; } o_halton_dim;

static o_halton_dim o_hdim[O_HALTON_DIMS];
static uint16_t *o_htable = 0;          /* concatenated per-dimension tables */
static uint32_t o_htable_len = 0;
static uint64_t o_hseed = ~0ull;
static int o_hlayout_done = 0;
static pthread_mutex_t o_hmutex = PTHREAD_MUTEX_INITIALIZER;

static void o_halton_layout(void)
{
  if(o_hlayout_done) return;
  uint32_t off = 0;
  int n = 0;
  for(uint32_t cand=2;n<O_HALTON_DIMS;cand++)
  {
    int prime = 1;
    for(uint32_t k=2;k*k<=cand;k++) if(cand % k == 0) { prime = 0; break; }
    if(!prime) continue;
    o_halton_dim *d = o_hdim + n++;
    d->base = cand;
    uint32_t P = cand;
    while(P*cand <= 500) P *= cand;                      /* halton_gen.py:105-109 */
    uint64_t M = P;
    uint32_t G = 1;
    while(M*P < (1ull<<32)) { M *= P; G++; }             /* halton_gen.py:113-116 */
    d->P = P; d->groups = G; d->offset = off;
    d->scale = (float)(0x1.fffffcp-1 / (double)M);
    if(cand != 2) off += P;                              /* base 2 needs no table */
  }
  o_htable_len = off;
  o_hlayout_done = 1;
}

static uint64_t o_lcg;
static void o_srand48(uint64_t seed) { o_lcg = ((seed & 0xffffffffull) << 16) | 0x330Eull; }
static long o_lrand48(void)
{
  o_lcg = (0x5DEECE66Dull*o_lcg + 0xBull) & 0xffffffffffffull;
  return (long)(o_lcg >> 17);
}

static uint16_t o_halton_invert(uint32_t base, uint32_t digits, uint32_t index, const uint16_t *perm)
{ /* _halton_invert, ext/halton/halton.h:2683-2693 */
  uint32_t result = 0;
  for(uint32_t i=0;i<digits;i++) { result = result*base + perm[index % base]; index /= base; }
  return (uint16_t)result;
}

/* tables for `seed` (= rt.anim_frame + number of re-initialisations, src/pointsampler.d/halton.c:46-52,122-129) */
static void o_halton_prepare_locked(uint64_t seed)
{
  o_halton_layout();
  if(o_htable && o_hseed == seed) return;
  if(!o_htable) o_htable = (uint16_t *)malloc(sizeof(uint16_t)*o_htable_len);
  /* one random permutation per base 4..1619 (all of them, prime or not, consume generator output);
     bases 1..3 keep the identity, ext/halton/halton.h:3251-3270 */
  uint16_t *perm = (uint16_t *)malloc(sizeof(uint16_t)*(O_HALTON_MAX_BASE+1));
  o_srand48(seed);
  int dim = 1;                                            /* next prime dimension to fill (dimension 0 = base 2) */
  for(uint32_t base=3;base<=O_HALTON_MAX_BASE;base++)
  {
    for(uint32_t i=0;i<base;i++) perm[i] = (uint16_t)i;
    if(base >= 4)
      for(uint32_t i=0;i<base-1;i++)
      {
        const size_t j = i + (size_t)(o_lrand48() / (long)(2147483648u/(base - i) + 1u));
        const uint16_t t = perm[j]; perm[j] = perm[i]; perm[i] = t;
      }
    if(dim < O_HALTON_DIMS && o_hdim[dim].base == base)
    {
      const o_halton_dim *d = o_hdim + dim++;
      uint32_t digits = 0;
      for(uint32_t p=1;p<d->P;p*=base) digits++;
      for(uint32_t i=0;i<d->P;i++) o_htable[d->offset + i] = o_halton_invert(base, digits, i, perm);
    }
  }
  free(perm);
  o_hseed = seed;
}

void o_halton_prepare(uint64_t seed)
{
  pthread_mutex_lock(&o_hmutex);
  o_halton_prepare_locked(seed);
  pthread_mutex_unlock(&o_hmutex);
}

float o_halton_sample(uint32_t dim, uint32_t index)
{ /* halton_sample, ext/halton/halton.h:2418-2680 */
  if(dim == 0)
  { /* halton2: bit reversal written into the mantissa, ext/halton/halton.h:291-306 */
    index = (index << 16) | (index >> 16);
    index = ((index & 0x00ff00ffu) << 8) | ((index & 0xff00ff00u) >> 8);
    index = ((index & 0x0f0f0f0fu) << 4) | ((index & 0xf0f0f0f0u) >> 4);
    index = ((index & 0x33333333u) << 2) | ((index & 0xccccccccu) >> 2);
    index = ((index & 0x55555555u) << 1) | ((index & 0xaaaaaaaau) >> 1);
    const uint32_t u = 0x3f800000u | (index >> 9);
    float f; memcpy(&f, &u, 4);
    return f - 1.0f;
  }
  const o_halton_dim *d = o_hdim + dim;
  const uint16_t *perm = o_htable + d->offset;
  uint32_t sum = 0;
  for(uint32_t g=0;g<d->groups;g++) { sum = sum*d->P + perm[index % d->P]; index /= d->P; }
  return (float)sum*d->scale;
}

/* test hooks: the concatenated tables (returns the number of entries; out may be NULL) and single samples */
uint32_t oracle_halton_tables(uint64_t seed, uint16_t *out, uint32_t *dim_base, uint32_t *dim_offset)
{
  pthread_mutex_lock(&o_hmutex);
  o_halton_prepare_locked(seed);
  if(out) memcpy(out, o_htable, sizeof(uint16_t)*o_htable_len);
  for(int d=0;d<O_HALTON_DIMS;d++) { if(dim_base) dim_base[d] = o_hdim[d].base; if(dim_offset) dim_offset[d] = o_hdim[d].offset; }
  pthread_mutex_unlock(&o_hmutex);
  return o_htable_len;
}

float oracle_halton_sample(uint64_t seed, uint32_t dim, uint32_t index)
{
  o_halton_prepare(seed);
  return o_halton_sample(dim, index);
}
