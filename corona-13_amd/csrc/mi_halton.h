/* mi_halton.h -- host side of the Halton point sampler (MI_POINTS_HALTON): the per-dimension constants and the digit
 * permutation tables the kernels read (halton_sample, mi_kernels.h).
 *
 * Replaces pointsampler_init / pointsampler_prepare_frame of src/pointsampler.d/halton.c:46-52,122-129, i.e.
 * halton_init_random(frame) of the vendored sampler ext/halton/halton.h:3244-3274 (L. Gruenschloss' generated Halton
 * sampler; the constants follow its generator, ext/halton/halton_gen.py:101-141):
 *   dimension d uses the d-th prime b; P = b^k is the largest power <= 500, so k digits are looked up at once;
 *   M = P^G is the largest power of P below 2^32; value = (sum_g table[(index / P^g) % P] * P^(G-1-g)) * (float)(0x1.fffffcp-1 / M)
 *   table[i] = the k base-b digits of i, each sent through the base's permutation, in reversed order
 *   permutations: identity for b <= 3, else a Fisher-Yates shuffle driven by lrand48() after srand48(seed), bases 4..1619
 *   in order (composite bases consume numbers too).
 * srand48/lrand48 are restated (POSIX: X' = (0x5DEECE66D X + 0xB) mod 2^48, X0 = (seed mod 2^32) << 16 | 0x330E, result X >> 17)
 * so the tables do not depend on the C library or on other users of its generator state.
 */
#ifndef MI_HALTON_H
#define MI_HALTON_H

#include <cstdint>
#include <cstring>
#include <vector>

#define MI_HALTON_DIMS 256

struct HaltonTables
{
  uint32_t dim[MI_HALTON_DIMS][4];   /* P, floor(2^32/P), offset | groups << 24, float bits of the scale */
  uint32_t base[MI_HALTON_DIMS], digits[MI_HALTON_DIMS];
  std::vector<uint16_t> perm;        /* concatenated tables, dimension 0 (base 2, bit reversal) has none */
};

static inline void halton_layout(HaltonTables &t)
{
  uint32_t offset = 0, d = 0;
  for(uint32_t b=2;d<MI_HALTON_DIMS;b++)
  {
    bool prime = true;
    for(uint32_t k=2;k*k<=b && prime;k++) prime = b % k != 0;
    if(!prime) continue;
    uint32_t P = b, digits = 1;
    while(P*b <= 500u) { P *= b; digits++; }
    uint64_t M = P;
    uint32_t groups = 1;
    while(M*P < (1ull << 32)) { M *= P; groups++; }
    const float scale = (float)(0x1.fffffcp-1/(double)M);
    uint32_t bits;
    memcpy(&bits, &scale, 4);
    t.base[d] = b; t.digits[d] = digits;
    t.dim[d][0] = P;
    t.dim[d][1] = (uint32_t)((1ull << 32)/P);
    t.dim[d][2] = offset | (groups << 24);
    t.dim[d][3] = bits;
    if(d) offset += P;
    d++;
  }
  t.perm.assign(offset, 0);
}

/* the kernels hard-wire the camera's dimensions 1..5 (halton_camera, mi_kernels.h): P, groups, table offset */
static inline bool halton_camera_constants_ok(const HaltonTables &t)
{
  static const uint32_t want[5][3] = { {243, 4, 0}, {125, 4, 243}, {343, 3, 368}, {121, 4, 711}, {169, 4, 832} };
  for(int d=1;d<=5;d++)
    if(t.dim[d][0] != want[d-1][0] || (t.dim[d][2] >> 24) != want[d-1][1] || (t.dim[d][2] & 0xffffffu) != want[d-1][2]) return false;
  return true;
}

static inline void halton_fill(HaltonTables &t, uint64_t seed)
{
  uint64_t x = ((seed & 0xffffffffull) << 16) | 0x330Eull;
  std::vector<uint16_t> sigma(t.base[MI_HALTON_DIMS-1]);
  uint32_t d = 1;
  for(uint32_t b=3;b<=t.base[MI_HALTON_DIMS-1];b++)
  {
    for(uint32_t i=0;i<b;i++) sigma[i] = (uint16_t)i;
    if(b > 3)
      for(uint32_t i=0;i+1<b;i++)
      {
        x = (0x5DEECE66Dull*x + 0xBull) & ((1ull << 48) - 1);
        const uint64_t r = x >> 17;                                 /* lrand48() */
        const uint64_t j = i + r/((1ull << 31)/(b - i) + 1);
        const uint16_t tmp = sigma[j]; sigma[j] = sigma[i]; sigma[i] = tmp;
      }
    if(d < MI_HALTON_DIMS && t.base[d] == b)
    {
      uint16_t *table = t.perm.data() + (t.dim[d][2] & 0xffffffu);
      for(uint32_t i=0;i<t.dim[d][0];i++)
      {
        uint32_t v = 0, rest = i;
        for(uint32_t k=0;k<t.digits[d];k++) { v = v*b + sigma[rest % b]; rest /= b; }
        table[i] = (uint16_t)v;
      }
      d++;
    }
  }
}

#endif
