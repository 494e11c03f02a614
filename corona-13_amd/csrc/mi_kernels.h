/* mi_kernels.h -- HIP device code of the pt/ptdl path tracing hot path for gfx950 (MI355X).
 *
 * This header holds the device functions of the persistent megakernel (mi_abi.hip): random numbers, primitive tests, resumable QBVH traversal (trace_round),
 * surface set-up, materials and BSDFs, emitter sampling, splat. Every lane owns one path at a time,
 *      NEW -> [ EXTEND-ray -> traverse -> shade ( -> SHADOW-ray -> traverse -> connect ) ]* -> NEW
 * and a lane whose path ends re-fills itself in place (wave64 ballot + prefix rank on a workgroup-local counter).
 * There is exactly one traversal site; extension and shadow rays of different lanes share it.
 *
 * Memory placement (see mi_device.h): BVH nodes staged once per workgroup into LDS when they fit (SoA of 16-byte
 * lanes so that divergent lanes spread over all LDS slots; otherwise read from HBM/L2, lds_setup), per-lane traversal
 * stack in LDS ([entry][thread] so a wave's accesses are conflict free), primitives as single 64-B records from
 * L2/HBM, per-primitive shading constants precomputed at upload (DPrimGeo), framebuffer splats as hardware float
 * atomics. MFMA is not used: there is no dense contraction anywhere in this path.
 *
 * Arithmetic follows the reference operation by operation (fp contraction off) so that a path with
 * the same random numbers takes the same branches:
 *   camera      src/camera.d/thinlens.c:68-128          rng    src/points.d/xorshift128p.c:53-74
 *   traversal   src/accel.d/qbvhmp.c:1188-1390          prims  src/prims.c:638-672, include/geo/{triangle,sphere,line}.h
 *   path        src/pathspace.c:80-292,697-895          pt     src/sampler.d/pt.c:30-54
 *   shading     src/shader.c:157-257,462-590, src/shaders/{color,colorcheckersg,dielectric,metal}.c, ggx.h
 *   emitters    src/lights.d/list.c:106-275             nee    include/pathspace/nee.h:87-243, src/sampler.d/ptdl.c:78-150
 *   splat       src/view.c:455-463, include/spectrum.h:172-203, include/filter/blackmanharris.h:28-77
 */
#ifndef MI_KERNELS_H
#define MI_KERNELS_H

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include "mi_device.h"

#define MI_PI_D 3.14159265358979323846
#define MI_PI_F 3.14159274101257324219f   /* (float)M_PI */

/* vertex_scattermode_t / vertex_flags_t, include/pathspace.h:57-82 */
enum { s_absorb = 0, s_reflect = 1, s_transmit = 2, s_volume = 4, s_fiber = 8, s_emit = 16, s_sensor = 32,
       s_diffuse = 64, s_glossy = 128, s_specular = 256 };
enum { s_none = 0, s_inside = 1, s_environment = 2 };
enum { s_tech_extend = 1, s_tech_nee = 2 };

/* reference MIN/MAX macros: NaN falls through to the second operand */
#define DMAX(a, b) ((a) > (b) ? (a) : (b))
#define DMIN(a, b) ((a) < (b) ? (a) : (b))
#define DCLAMP(a, m, M) DMIN(DMAX(a, m), M)

struct V3 { float x, y, z; };
#define MI_HD __host__ __device__ __forceinline__   /* per-primitive constants are precomputed on the host with the very same code */
/* 1.0f/x. On the device: v_rcp_f32 + two Newton steps + v_div_fixup_f32 (6 instructions) instead of the compiler's IEEE division
 * (11: two v_div_scale, v_rcp, five fma, v_div_fmas, v_div_fixup). Bit-identical to the correctly rounded 1.0f/x for every
 * x in {+-0, +-inf, NaN} and every normal x with |x| < 2^126 -- checked over all 2^32 bit patterns by tools/micro/rcp_exact.hip;
 * what the scaling steps of the long form are for, a denormal x or |x| >= 2^126 (no length, determinant or cosine of this
 * geometry), can come out one ulp off. The host (per-primitive constants at upload) divides. */
MI_HD float mi_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_IEEE_DIV)
  float y = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, y, 1.0f);
  y = __builtin_fmaf(y, e, y);
  e = __builtin_fmaf(-x, y, 1.0f);
  y = __builtin_fmaf(y, e, y);
  return __builtin_amdgcn_div_fixupf(y, x, 1.0f);
#else
  return 1.0f/x;
#endif
}
/* sincosf(y). Default: the device libm's. With -DMI_HOST_SINCOS: as the HOST computes it -- the reference (and the oracle) call glibc's
 * sincosf / sinf / cosf, the device libm's results differ from those in the last ulp now and then, and that decides about half of the
 * paths that part ways with the oracle (grazing hits): 4 M-path soaks, paths with a different primitive sequence, device libm / this:
 * cfg 2 12 / 6, cfg 3 13 / 6, cfg 4 64 / 33, metal ptdl 35 / 1. glibc's float routines (sysdeps/ieee754/flt-32/s_sincosf.c, 2.35: quadrant
 * by a scaled conversion, a degree-4 and a degree-3 polynomial in x^2, all in double, one rounding to float at the end) are restated here
 * operation by operation in device doubles; the coefficients are the table in libm.so's data (`__sincosf_table`). Bit-identical to the
 * host's sincosf for every float in [-2 pi, 2 pi] -- 2 x 1 086 918 620 arguments, checked on the host (the restatement against glibc) and on
 * the device (tools/micro/sincos_exact.hip); between 17 and 120, where the host's FMA build of glibc rounds the reduction once less, 1 in
 * 10^6 differs by an ulp. Arguments of 120 and beyond (none here: every angle is 2 pi u or a multiple of pi below 6 pi) go to the device
 * libm. Not the default because a double operation costs 4.7 cycles against 2.8 (tools/micro/valu_cost.hip): cfg 2 +1.8 %, cfg 3 +3.9 %
 * kernel time. */
MI_HD void mi_sincosf(float y, float *sinp, float *cosp)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(MI_HOST_SINCOS)
  const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;            /* abstop12 */
  if(top >= ((0x42f00000u >> 20) & 0x7ffu)) { sincosf(y, sinp, cosp); return; }     /* |y| >= 120, inf, NaN */
  double x = (double)y;
  int n = 0;
  if(top < ((0x3f490fdbu >> 20) & 0x7ffu))
  { /* |y| < pi/4 */
    if(top < ((0x39800000u >> 20) & 0x7ffu)) { *sinp = y; *cosp = 1.0f; return; }   /* |y| < 2^-12 */
  }
  else
  { /* reduce_fast: hpi_inv is 2/pi * 2^24, the quadrant ends up in bits 24..31 */
    const double r = x*0x1.45f306dc9c883p+23;
    n = ((int)r + 0x800000) >> 24;
    x = x - (double)n*0x1.921fb54442d18p+0;
    if(((n ^ (n >> 1)) & 1) != 0) x = -x;                             /* sign[n & 3] = {1, -1, -1, 1} */
  }
  const double x2 = x*x;
  /* sincosf_poly; for n & 2 glibc switches to a table with the cosine coefficients negated: the same sums with the sign flipped */
  const double x4 = x2*x2, x3 = x2*x;
  const double c2 = -0x1.6c087e89a359dp-10 + x2*0x1.99343027bf8c3p-16;
  const double s1 = 0x1.1107605230bc4p-7 + x2*-0x1.994eb3774cf24p-13;
  const double c1 = 0x1p0 + x2*-0x1.ffffffd0c621cp-2;
  const double x5 = x3*x2, x6 = x4*x2;
  const double s = x + x3*-0x1.555545995a603p-3;
  const double c = c1 + x4*0x1.55553e1068f19p-5;
  const float fs = (float)(s + x5*s1);
  float fc = (float)(c + x6*c2);
  if(n & 2) fc = -fc;
  *sinp = (n & 1) ? fc : fs;
  *cosp = (n & 1) ? fs : fc;
#else
  sincosf(y, sinp, cosp);
#endif
}
MI_HD float mi_sinf(float y) { float s, c; mi_sincosf(y, &s, &c); return s; }
MI_HD float mi_cosf(float y) { float s, c; mi_sincosf(y, &s, &c); return c; }

/* sqrtf(x). On the device: v_sqrt_f32 (within one ulp) and the choice among its two neighbours by the sign of the exact
 * residuals x - s*(s -+ 1ulp) -- the core of the compiler's correctly rounded expansion without the scaling it wraps around
 * it for denormal arguments (9 instead of 16 instructions). Bit-identical to sqrtf for +-0, inf, NaN, negative x and every
 * x >= 2^-104 (4.9e-32), over all 2^32 bit patterns (tools/micro/rcp_exact.hip); smaller positive x (no squared length or
 * discriminant here) can come out one ulp off. */
MI_HD float mi_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_IEEE_DIV)
  const float s = __builtin_amdgcn_sqrtf(x);
  const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
  const float rlo = __builtin_fmaf(-lo, s, x), rhi = __builtin_fmaf(-hi, s, x);
  float r = rlo <= 0.0f ? lo : s;
  r = rhi > 0.0f ? hi : r;
  return r;
#else
  return sqrtf(x);
#endif
}
MI_HD V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MI_HD float dot3(const V3 a, const V3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
MI_HD V3 cross3(const V3 a, const V3 b)
{ /* crossproduct macro, include/corona_common.h:161-164 */
  return mk3(a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y);
}
MI_HD V3 sub3(const V3 a, const V3 b) { return mk3(a.x-b.x, a.y-b.y, a.z-b.z); }
MI_HD V3 scale3(const V3 a, float s) { return mk3(a.x*s, a.y*s, a.z*s); }
MI_HD V3 neg3(const V3 a) { return mk3(-a.x, -a.y, -a.z); }
MI_HD V3 normalise3(const V3 a)
{ /* normalise, include/corona_common.h:172-176 */
  const float len = mi_rcp(mi_sqrt(dot3(a, a)));
  return scale3(a, len);
}
MI_HD V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }

MI_HD void get_onb(const V3 n, V3 &u, V3 &v)
{ /* get_onb, include/corona_common.h:178-198 */
  if(fabsf(n.y) < 0.5) u = cross3(n, mk3(0, 1, 0));
  else                 u = cross3(n, mk3(1, 0, 0));
  u = normalise3(u);
  v = cross3(n, u);
}
__device__ __forceinline__ void get_scrambled_onb(float scramble, const V3 n, V3 &u, V3 &v)
{ /* get_scrambled_onb, include/corona_common.h:200-215 */
  if(fabsf(n.y) < scramble) u = cross3(n, mk3(0, 1, 0));
  else                      u = cross3(n, mk3(1, 0, 0));
  u = normalise3(u);
  v = cross3(n, u);
}

/* ------------------------------------------------------------------------------------------ rng */
struct Rng { unsigned long long s0, s1; };
__device__ __forceinline__ float rng_next(Rng &r)
{ /* points_rand, src/points.d/xorshift128p.c:61-74 */
  unsigned long long s1 = r.s0;
  const unsigned long long s0 = r.s1;
  r.s0 = s0;
  s1 ^= s1 << 23;
  s1 ^= s1 >> 17;
  s1 ^= s0;
  s1 ^= s0 >> 26;
  r.s1 = s1;
  const uint32_t v = 0x3f800000u | (uint32_t)((r.s0 + r.s1) >> 41);
  return __uint_as_float(v) - 1.0f;
}
__device__ __forceinline__ void rng_seed(Rng &r, unsigned long long index, unsigned long long frame)
{ /* points_set_state, src/points.d/xorshift128p.c:53-59 with thread id 0 (src/render.d/gi.c:88) */
  r.s0 = 1 + index;
  r.s1 = 2 + frame;
  for(int k=0;k<10;k++) (void)rng_next(r);
}
#ifndef MI_RNG_JUMP
#define MI_RNG_JUMP 1     /* points_set_state's ten warm-up rounds by table look-up: 140 of path_generate's 527 vector instructions (profiles/r06_levers.txt block 5) */
#endif
/* points_set_state(index, frame) = ten rounds of the generator on (1 + index, 2 + frame), src/points.d/xorshift128p.c:53-59. One round maps (s0, s1) to
 * (s1, f(s0) ^ g(s1)) with f, g built from shifts and XORs: linear over GF(2), and so are ten rounds. The seed splits into bit-disjoint parts --
 * the four bytes of the low word of 1 + index, and the rest (its high word, 2 + frame: the same for every path of a launch) -- so the warmed-up state is
 * the XOR of the ten rounds of each part: four 16-byte table entries and a launch constant, the very same 128 bits the rounds compute. */
__device__ __forceinline__ void rng_seed_jump(Rng &r, const DScene &sc, unsigned long long index)
{
  const unsigned long long x = 1ull + index;
  if(MI_RNG_JUMP && sc.rng_jump && (uint32_t)(x >> 32) == sc.rng_jump_hi)
  {
    const uint32_t lo = (uint32_t)x;
    const uint4 a = sc.rng_jump[lo & 255u], b = sc.rng_jump[256u + ((lo >> 8) & 255u)], c = sc.rng_jump[512u + ((lo >> 16) & 255u)], d = sc.rng_jump[768u + (lo >> 24)];
    const uint32_t w0 = sc.rng_jump_c[0] ^ a.x ^ b.x ^ c.x ^ d.x, w1 = sc.rng_jump_c[1] ^ a.y ^ b.y ^ c.y ^ d.y;
    const uint32_t w2 = sc.rng_jump_c[2] ^ a.z ^ b.z ^ c.z ^ d.z, w3 = sc.rng_jump_c[3] ^ a.w ^ b.w ^ c.w ^ d.w;
    r.s0 = (unsigned long long)w0 | ((unsigned long long)w1 << 32);
    r.s1 = (unsigned long long)w2 | ((unsigned long long)w3 << 32);
  }
  else rng_seed(r, index, sc.frame);
}
__device__ __forceinline__ unsigned long long mi_splitmix64(unsigned long long z)
{
  z = (z ^ (z >> 30))*0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27))*0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ void rng_seed_hashed(Rng &r, unsigned long long index, unsigned long long frame)
{ /* MI_PIXELS_FROM_INDEX only (corona_mi.h), NOT the reference's seeding: both state words through the splitmix64 finaliser, then the same ten
     rounds. Seeded as above, the first numbers of paths i and i + 1 are correlated (their wavelengths: r = 0.96) -- harmless while a path's
     pixel is one of those numbers, a colour cast that takes thousands of samples to average out once neighbouring indices are neighbouring pixels */
  r.s0 = mi_splitmix64(1 + index + 0x9e3779b97f4a7c15ull);
  r.s1 = mi_splitmix64(2 + frame + 2*0x9e3779b97f4a7c15ull + r.s0);
  if(!(r.s0 | r.s1)) r.s0 = 1;
  for(int k=0;k<10;k++) (void)rng_next(r);
}

/* ---------------------------------------------------------------------------------- point sampler
 * pointsampler(path, dim), MOD_pointsampler = rand (src/pointsampler.d/rand.c:48-55: the next number of the per-path
 * generator whatever the dimension) or halton (src/pointsampler.d/halton.c:69-84 over ext/halton/halton.h: permuted radical
 * inverse of the low 32 bits of the path index in the (rand_beg + dim)-th prime base; dimensions >= 256 fall back to the
 * generator). All integer arithmetic is exact, the float conversion and the one multiplication round as on the host. */
enum { MI_DIM_IMAGE_X = 0, MI_DIM_IMAGE_Y = 1, MI_DIM_LAMBDA = 2, MI_DIM_TIME = 3, MI_DIM_APERTURE_X = 4, MI_DIM_APERTURE_Y = 5, MI_DIM_CAMID = 6,
       MI_DIM_OMEGA_X = 1, MI_DIM_OMEGA_Y = 2, MI_DIM_SCATTER_MODE = 3, MI_DIM_RUSSIAN_R = 4,
       MI_DIM_NEE_LIGHT1 = 0, MI_DIM_NEE_LIGHT2 = 1, MI_DIM_NEE_X = 2, MI_DIM_NEE_Y = 3 };      /* include/pathspace.h:16-53 */

/* The first MI_HALTON_LDS entries of the concatenated permutation tables (dimensions 1 .. ~17: the camera's five and the first
 * two or three path vertices' -- the dimensions every path draws) are staged into LDS at the start of the workgroup's dynamic
 * LDS by the HALTON instantiations (lds_setup); the rest of the 387 KB stays in L2. 0 = everything from L2 (A/B switch). */
#ifndef MI_HALTON_LDS
#define MI_HALTON_LDS 0      /* entries of the Halton permutation tables' head staged into LDS. Rounds 2-3: 4096 (+0.2 % pt, +0.5 % ptdl against L2: the look-ups are in
                                flight together anyway). Round 4: the 8 KB are worth more to the pools of the exchange between waves (mi_regroup.h): 0010 Halton pt
                                17.92 -> 17.26 ms, ptdl 33.2 -> 31.5 with the tables left in L2 */
#endif
__device__ __forceinline__ const unsigned short *halton_lds()
{
  extern __shared__ __attribute__((aligned(16))) unsigned char mi_dynamic_lds[];
  return (const unsigned short *)mi_dynamic_lds;
}

__device__ __forceinline__ float halton_sample(const DScene &sc, uint32_t dim, uint32_t index)
{
  if(dim == 0) return __uint_as_float(0x3f800000u | (__brev(index) >> 9)) - 1.0f;            /* halton2, ext/halton/halton.h:291-306 */
  const uint4 d = sc.halton_dim[dim];
  const uint32_t P = d.x, recip = d.y, groups = d.z >> 24;
  const uint32_t off = d.z & 0xffffffu;
  const unsigned short *perm = sc.halton_perm + off;
  const bool staged = MI_HALTON_LDS && off + P <= MI_HALTON_LDS;                              /* this dimension's table is in LDS */
  const unsigned short *lperm = halton_lds() + off;
  /* at most 7 groups (P = 23: 23^7 < 2^32). First all remainders, then all look-ups (in flight together instead of one
     load latency per group), then the sum; groups beyond the dimension's own read entry 0 and are left out of the sum */
  uint32_t rem[7], digit[7];
#pragma unroll
  for(int g=0;g<7;g++)
  { /* index / P and index % P: floor(2^32/P) as reciprocal is at most one too small */
    uint32_t q = __umulhi(index, recip);
    uint32_t r = index - q*P;
    if(r >= P) { q++; r -= P; }
    rem[g] = (uint32_t)g < groups ? r : 0u;
    index = q;
  }
  if(staged)
  {
#pragma unroll
    for(int g=0;g<7;g++) digit[g] = lperm[rem[g]];
  }
  else
  {
#pragma unroll
    for(int g=0;g<7;g++) digit[g] = perm[rem[g]];
  }
  uint32_t sum = 0;
#pragma unroll
  for(int g=0;g<7;g++) if((uint32_t)g < groups) sum = sum*P + digit[g];
  return (float)sum*__uint_as_float(d.w);
}

/* the camera's dimensions (1..5: bases 3, 5, 7, 11, 13) are the same for every path: constants instead of the descriptor load,
 * divisions by constants, all table look-ups in flight together. P, G, OFF as halton_layout (mi_halton.h) computes them. */
template<uint32_t P, uint32_t G, uint32_t OFF>
__device__ __forceinline__ float halton_const(const DScene &sc, uint32_t index)
{
  const unsigned short *perm = (MI_HALTON_LDS && OFF + P <= MI_HALTON_LDS) ? halton_lds() + OFF : sc.halton_perm + OFF;
  uint32_t digit[G];
#pragma unroll
  for(uint32_t g=0;g<G;g++) { digit[g] = perm[index % P]; index /= P; }
  uint32_t sum = 0;
  double M = 1.0;
#pragma unroll
  for(uint32_t g=0;g<G;g++) { sum = sum*P + digit[g]; M *= (double)P; }
  return (float)sum*(float)(0x1.fffffcp-1/M);
}
template<int DIM>
__device__ __forceinline__ float halton_camera(const DScene &sc, uint32_t index)
{
  static_assert(DIM >= 0 && DIM <= 5, "camera dimensions");
  if(DIM == 0) return __uint_as_float(0x3f800000u | (__brev(index) >> 9)) - 1.0f;
  if(DIM == 1) return halton_const<243, 4, 0>(sc, index);
  if(DIM == 2) return halton_const<125, 4, 243>(sc, index);
  if(DIM == 3) return halton_const<343, 3, 368>(sc, index);
  if(DIM == 4) return halton_const<121, 4, 711>(sc, index);
  return halton_const<169, 4, 832>(sc, index);
}

/* vertex v's first dimension: the camera owns 7, v[1] the free-path dimension, every extension 5; ptdl's next-event vertex
 * owns 4 more, which path_pop hands to the vertex before it (thinlens.c:100-103, src/pathspace.c:199,208,298, nee.h:108,231) */
template<bool PTDL> __device__ __forceinline__ int rand_beg_extend(int v) { return PTDL ? 12 + 9*(v-2) : 8 + 5*(v-2); }
__device__ __forceinline__ int rand_beg_nee(int v) { return 8 + 9*(v-2); }

template<bool HALTON>
struct PointSampler
{
  const DScene &sc;
  Rng &rng;
  uint32_t index;
  int beg;                       /* rand_beg of the vertex under construction */
  __device__ __forceinline__ PointSampler(const DScene &sc_, Rng &rng_, unsigned long long index_, int beg_) : sc(sc_), rng(rng_), index((uint32_t)index_), beg(beg_) {}
  __device__ __forceinline__ float operator()(int dim)
  {
    if(HALTON && beg + dim < 256) return halton_sample(sc, (uint32_t)(beg + dim), index);
    return rng_next(rng);
  }
  template<int DIM> __device__ __forceinline__ float camera()
  {
    if(HALTON) return halton_camera<DIM>(sc, index);
    return rng_next(rng);
  }
};

/* ------------------------------------------------------------------------------------------ work counters
 * c[0..7]: rays, node visits, box hits, primitive tests, paths, splats, vertices, deepest stack -- the reference's -DACCEL_DEBUG
 * counters (src/accel.d/qbvhmp.c:83-90,1168-1173) and a few more. Like there they are a debug facility: the COUNT = false
 * instantiations of the kernels keep only the path count c[4] (the seven others are live in every loop and cost the ptdl kernel
 * 10 % through register pressure, the pt kernel 1 %); mi_scene_set_counters() selects the counting kernels.
 * Development builds append more: -DMI_PROFILE_LOOPS wave-level loop iterations (c[8..10]), -DMI_PROFILE_PHASES lane-0 clock
 * ticks per phase (c[8+k]), their occurrences (c[16+k]), the last marker (c[30]) and time stamp (c[31]). */
#if defined(MI_PROFILE_PHASES) || defined(MI_PROFILE_LOOPS) || defined(MI_PROFILE_TRAV) || defined(MI_PROFILE_BLOCKS) || defined(MI_PROFILE_POOL)
#define MI_CNT 32
#else
#define MI_CNT 8
#endif
template<bool ON> struct Counters
{
  static constexpr bool on = ON;
  uint32_t c[MI_CNT];
  __device__ __forceinline__ Counters() { for(int k=0;k<MI_CNT;k++) c[k] = 0; }
};
#define MI_COUNT(cnt, k, v) do { if((cnt).on) (cnt).c[k] += (v); } while(0)
#define MI_COUNT_MAX(cnt, k, v) do { if((cnt).on) (cnt).c[k] = (cnt).c[k] > (v) ? (cnt).c[k] : (v); } while(0)
#ifdef MI_PROFILE_PHASES
#define MI_PHASE_INIT(cnt) { (cnt).c[31] = (uint32_t)clock64(); (cnt).c[30] = 6; }
/* markers run 0 1 [2 3 4 7] 5 6 per iteration; an interval counts only if this lane also passed the marker before it */
#define MI_PHASE(cnt, k) { const uint32_t t_ = (uint32_t)clock64(); \
  const uint32_t pred_ = (k) == 0 ? 6 : (k) == 5 ? 7 : (k) == 7 ? 4 : (k) - 1; \
  if((cnt).c[30] == pred_) { (cnt).c[8 + (k)] += t_ - (cnt).c[31]; (cnt).c[16 + (k)]++; } \
  (cnt).c[31] = t_; (cnt).c[30] = (k); }
#else
#define MI_PHASE_INIT(cnt)
#define MI_PHASE(cnt, k)
#endif
/* -DMI_PROFILE_BLOCKS (development build, tools/block_probe.py): how full the wave is where it executes a block of the shading code:
 * every lane that runs block k counts itself, the first active lane counts the execution; counter k leaves the kernel as lanes | executions << 36 */
/* -DMI_PROFILE_POOL (development build, tools/pool_probe.py): the exchange between waves (mi_regroup.h) in the same format: per wave
 * iteration 0 = exchanges made | lanes posted, 1 = lanes pulled, 2 = turns to a class the wave's own lanes are not mostly in | lanes
 * shaded in such a turn, 3 = spins on the lock, 4 = vertices that could not be posted (pool full) */
#ifdef MI_PROFILE_POOL
#define MI_POOLSTAT(cnt, k, lanes, execs) { if(__lane_id() == 0) { (cnt).c[8 + (k)] += (lanes); (cnt).c[16 + (k)] += (execs); } }
#else
#define MI_POOLSTAT(cnt, k, lanes, execs)
#endif
#ifdef MI_PROFILE_BLOCKS
#define MI_BLK(cnt, k) { (cnt).c[8 + (k)]++; if(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) (cnt).c[16 + (k)]++; }
#else
#define MI_BLK(cnt, k)
#endif
/* -DMI_PROFILE_TRAV (development build, tools/trav_probe.py): lane 0's clock ticks per part of a wave iteration, summed over the
 * launch into counters 0..6: 0 node loop, 1 job list set-up, 2 job passes, 3 owners' epilogue (results, sphere / line tests, pop),
 * 4 rest of the slice loop, 5 refill + ray start + shading, 6 splat. MI_TT adds the time since the previous mark to part k. */
#ifdef MI_PROFILE_TRAV
#define MI_TT(cnt, k) { const uint32_t t_ = (uint32_t)clock64(); (cnt).c[8 + (k)] += t_ - (cnt).c[31]; (cnt).c[31] = t_; }
#else
#define MI_TT(cnt, k)
#endif

/* ------------------------------------------------------------------------------------------ hit */
struct Hit
{
  uint32_t prim;       /* builder-order primitive index, 0xffffffff = none */
  float dist, u, v;
};
#define MI_NOPRIM 0xffffffffu

/* ------------------------------------------------------------------------------------------ primitives */
__device__ __forceinline__ float sphere_t(const V3 c, float radius, const V3 ro, const V3 rd)
{ /* _geo_sphere_intersect, include/geo/sphere.h:112-144, with the roundings of the reference BUILD (its FMA contraction, from the
     disassembly of prims_intersect; the oracle restates the same sequence): dot products y*y, then fused x, then fused z; the
     discriminant fma(b, b, -(4 a) c). The quadratic cancels badly for rays from afar; the fused forms put the hit point 1.4e-5
     (rms) off the sphere instead of 1.8e-5, and fewer grazing rays start inside it. */
  const V3 o = sub3(ro, c);
  const float a = __builtin_fmaf(rd.z, rd.z, __builtin_fmaf(rd.x, rd.x, rd.y*rd.y));
  const float od = __builtin_fmaf(rd.z, o.z, __builtin_fmaf(rd.x, o.x, rd.y*o.y));
  const float b = od + od;
  const float cc = __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.x, o.x, o.y*o.y)) - radius*radius;
  if(a == 0)
  {
    if(b != 0) return -cc/b;
    return -FLT_MAX;
  }
  const float discrim = __builtin_fmaf(b, b, -((a*4.0f)*cc));
  if(discrim < 0) return -FLT_MAX;
  const float sq = mi_sqrt(discrim);
  const float temp = b < 0 ? -0.5f*(b - sq) : -0.5f*(b + sq);
  const float x0 = temp/a, x1 = cc/temp;
  if(x0 <= 0.0f) return x1;
  else if(x1 <= 0.0f) return x0;
  else return fminf(x0, x1);
}

/* Line primitives (truncated cones with radii r0, r1) carry their per-primitive constants precomputed at upload, with
 * the very float operations the reference performs per test (mi_abi.hip: pack_line):
 *   dword 0-2 v0 | 3 r0 | 4 r1 | 5 |v1-v0| | 6-8 unit axis d | 9-11 cylinder: onb a, cone: tip | 12 type | 13-15 cylinder: onb b, cone: cos_a2,-,-
 * For a cylinder hit.u / hit.v transport the raw cross-section coordinates (out[1], out[2]); the angle
 * hit->v = atan2f(out[1], out[2])/2pi (include/geo/line.h:486) is evaluated once at shading time, not per candidate. */
__device__ __forceinline__ void line_intersect(const DPrim &p, const V3 ro, const V3 rd, Hit &hit, uint32_t prim, uint32_t ignore)
{ /* geo_line_intersect + _geo_line_intersect_{cylinder,cone}, include/geo/line.h:313-505 (hair strips out of scope) */
  const float *f = &p.v[0][0];
  const V3 v0 = mk3(f[0], f[1], f[2]);
  const float r0 = f[3], r1 = f[4];
  const bool linestrip = DMAX(r0, r1) <= 1e-2f;
  if(linestrip && ignore == prim) return;
  const V3 d = mk3(f[6], f[7], f[8]);
  if(fabsf(r1-r0) < 1e-3)
  {
    const float dlen = f[5];
    const V3 a = mk3(f[9], f[10], f[11]), b = mk3(f[13], f[14], f[15]);
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, w0 = 0.0f, w1 = 0.0f, w2 = 0.0f;
    const float px[3] = {ro.x - v0.x, ro.y - v0.y, ro.z - v0.z}, dx[3] = {rd.x, rd.y, rd.z};
    const float dd[3] = {d.x, d.y, d.z}, aa[3] = {a.x, a.y, a.z}, bb[3] = {b.x, b.y, b.z};
#pragma unroll
    for(int k=0;k<3;k++)
    {
      o0 += px[k]*dd[k]; o1 += px[k]*aa[k]; o2 += px[k]*bb[k];
      w0 += dx[k]*dd[k]; w1 += dx[k]*aa[k]; w2 += dx[k]*bb[k];
    }
    const float A = w1*w1 + w2*w2;
    const float B = 2.0f*(o1*w1 + o2*w2);
    const float C = o1*o1 + o2*o2 - r0*r0;
    const float discr = (float)((double)(B*B) - 4.0*(double)A*(double)C);
    if(discr < 0.0) return;
    const float sq = mi_sqrt(discr);
    const float temp = B < 0 ? -0.5f*(B - sq) : -0.5f*(B + sq);
    const float t0 = temp/A, t1 = C/temp;
    float t, out0, out1, out2;
    bool ok = false;
    if(t0 <= 0.0f || t1 <= 0.0f)
    {
      t = t0 <= 0.0f ? t1 : t0;
      out0 = o0 + t*w0; out1 = o1 + t*w1; out2 = o2 + t*w2;
      ok = out0 >= 0.0 && out0 <= dlen;
    }
    else
    {
      t = fminf(t0, t1);
      out0 = o0 + t*w0; out1 = o1 + t*w1; out2 = o2 + t*w2;
      ok = out0 >= 0.0 && out0 <= dlen;
      if(!ok)
      {
        t = fmaxf(t0, t1);
        out0 = o0 + t*w0; out1 = o1 + t*w1; out2 = o2 + t*w2;
        ok = out0 >= 0.0 && out0 <= dlen;
      }
    }
    if(!ok) return;
    if(t > 0.0f && t < hit.dist)
    {
      hit.dist = t; hit.prim = prim;
      hit.u = out1; hit.v = out2;                 /* raw; angle taken at shading time */
    }
  }
  else
  {
    const float d_len = f[5], cos_a2 = f[13];
    const float cos_dr = dot3(d, rd);
    const V3 tip = mk3(f[9], f[10], f[11]);      /* v0 + (-r0*d_len/(r1-r0))*d, formed at upload */
    const V3 o = sub3(ro, tip);
    const float cos_do = dot3(d, o);
    const float cos_ro = dot3(rd, o);
    const float cos_oo = dot3(o, o);
    const float c2 = cos_dr*cos_dr - cos_a2;
    const float c1 = cos_dr*cos_do - cos_a2*cos_ro;
    const float c0 = cos_do*cos_do - cos_a2*cos_oo;
    float tmin = -1.0f, dist = hit.dist, hu = hit.u;
    if(fabsf(c2) > 0.0)
    {
      const float discr = c1*c1 - c0*c2;
      if(discr < 0.0f) return;
      const float root = mi_sqrt(discr);
#pragma unroll
      for(int i=-1;i<2;i+=2)
      {
        const float t = (-c1 + i*root)/c2;
        if(t > 0.0 && t < dist)
        {
          const V3 x = mk3(ro.x + t*rd.x - v0.x, ro.y + t*rd.y - v0.y, ro.z + t*rd.z - v0.z);
          const float dt = dot3(x, d);
          if(dt >= 0.0f && dt <= d_len)
          {
            hu = dt/d_len;                          /* hit->v (angle around the axis) is recomputed from the hit point at shading time */
            tmin = dist = t;
          }
        }
      }
    }
    if((linestrip && tmin > 1e-3f) || (!linestrip && tmin > 0.0f))
    {
      hit.dist = tmin; hit.prim = prim; hit.u = hu; hit.v = 0.0f;
    }
  }
}

struct PrimRegs { float4 q0, q1, q2, q3; };   /* one DPrim as four 16-byte loads: v0.xyz e1.x | e1.yz e2.xy | e2.z e3.xyz | type pad */
__device__ __forceinline__ PrimRegs prim_load(const DPrim *prims, uint32_t prim)
{
  const float4 *q = (const float4 *)(prims + prim);
  PrimRegs r; r.q0 = q[0]; r.q1 = q[1]; r.q2 = q[2]; r.q3 = q[3];
  return r;
}

template<bool BOTH = false>
__device__ __forceinline__ bool triquad_intersect(const PrimRegs &r, uint32_t type, const V3 o, const V3 d, Hit &hit, uint32_t prim)
{ /* prims_intersect for tris and quads, src/prims.c:645-663: quad = tri(v0,v1,v2), and only if that misses tri(v0,v2,v3).
     Both triangles are evaluated without branches (same arithmetic as geo_tri_intersect, include/geo/triangle.h:263-305)
     and the reference's priority is applied afterwards, so a wave does not diverge on which half was hit. */
  /* the record holds v0 and the three edges v1-v0, v2-v0, v3-v0 (formed at upload with the same float subtraction) */
  const V3 v0 = mk3(r.q0.x, r.q0.y, r.q0.z), eA1 = mk3(r.q0.w, r.q1.x, r.q1.y), e02 = mk3(r.q1.z, r.q1.w, r.q2.x), eB2 = mk3(r.q2.y, r.q2.z, r.q2.w);
  const V3 tv = sub3(o, v0);
  /* triangle A: edge1 = v1-v0, edge2 = v2-v0 */
  const V3 pA = cross3(d, e02);
  const float invA = mi_rcp(dot3(eA1, pA));
  const float vA = dot3(tv, pA)*invA;
  const V3 qA = cross3(tv, eA1);
  const float uA = dot3(d, qA)*invA;
  const float tA = dot3(e02, qA)*invA;
  const bool preA = !(vA < 0.0f || vA > 1.0f) && !(uA < 0.0f || uA + vA > 1.0f) && tA > 0.0f;     /* crossed in front of the ray */
  /* triangle B: edge1 = v2-v0, edge2 = v3-v0 */
  const V3 pB = cross3(d, eB2);
  const float invB = mi_rcp(dot3(e02, pB));
  const float vB = dot3(tv, pB)*invB;
  const V3 qB = cross3(tv, e02);
  const float uB = dot3(d, qB)*invB;
  const float tB = dot3(eB2, qB)*invB;
  const bool preB = (type == MI_PRIM_QUAD) && !(vB < 0.0f || vB > 1.0f) && !(uB < 0.0f || uB + vB > 1.0f) && tB > 0.0f;
  /* BOTH: both halves of a quad are crossed in front of the ray -- only a folded (non-planar) quad. Which half the reference then
     reports depends on the running closest hit (src/prims.c:654-663: the second half is tested iff the first one is not accepted),
     i.e. on the primitives of the leaf tested BEFORE it: leaf_jobs, which tests all of a leaf's primitives against the distance the
     leaf started with, hands such a leaf to the per-lane loop. (The per-lane loop works in order except that it puts off spheres,
     lines and moving primitives; a quad that this could affect is marked at upload, see mi_mark_ordered_kernel.) */
  const bool both = BOTH && preA && preB;
  const bool hitA = preA && tA <= hit.dist, hitB = preB && tB <= hit.dist;
  if(hitA)
  {
    hit.dist = tA; hit.prim = prim; hit.u = uA;
    hit.v = (type == MI_PRIM_QUAD) ? vA + uA : vA;              /* tri (v0 v1 v2) uv => quad uv = (u, v+u) */
  }
  else if(hitB)
  {
    hit.dist = tB; hit.prim = prim; hit.u = uB + vB; hit.v = vB;  /* tri (v0 v2 v3) uv => quad uv = (u+v, v) */
  }
  return both;
}

/* the 64-byte record of a line segment (truncated cone / cylinder) as line_intersect reads it, from its two end points and
 * radii: formed once at upload for a static line, per ray for a moving one (include/geo/line.h:313-462) */
MI_HD void pack_line(DPrim &p, const V3 v0, const V3 v1, float r0, float r1)
{
  float *f = &p.v[0][0];
  float d[3] = {v1.x-v0.x, v1.y-v0.y, v1.z-v0.z};
  const float dlen = mi_sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
  for(int k=0;k<16;k++) f[k] = 0.0f;
  f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = r0; f[4] = r1; f[5] = dlen;
  if(fabsf(r1-r0) < 1e-3)
  { /* cylinder: d *= 1.0f/dlen; get_onb(d, a, b) */
    const float inv = 1.0f/dlen;
    for(int k=0;k<3;k++) d[k] *= inv;
    float a[3], b[3];
    if(fabsf(d[1]) < 0.5) { a[0] = d[1]*0.0f - 1.0f*d[2]; a[1] = d[2]*0.0f - 0.0f*d[0]; a[2] = d[0]*1.0f - 0.0f*d[1]; }   /* d x (0,1,0) */
    else                  { a[0] = d[1]*0.0f - 0.0f*d[2]; a[1] = d[2]*1.0f - 0.0f*d[0]; a[2] = d[0]*0.0f - 1.0f*d[1]; }   /* d x (1,0,0) */
    const float il = 1.0f/mi_sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]);
    for(int k=0;k<3;k++) a[k] *= il;
    b[0] = d[1]*a[2] - a[1]*d[2]; b[1] = d[2]*a[0] - a[2]*d[0]; b[2] = d[0]*a[1] - a[0]*d[1];
    f[6] = d[0]; f[7] = d[1]; f[8] = d[2];
    f[9] = a[0]; f[10] = a[1]; f[11] = a[2];
    f[13] = b[0]; f[14] = b[1]; f[15] = b[2];
  }
  else
  { /* cone: d *= 1.0/d_len (double), cos_a2 */
    for(int k=0;k<3;k++) d[k] = (float)(d[k]*(1.0/dlen));
    f[6] = d[0]; f[7] = d[1]; f[8] = d[2];
    const float tt = -r0*dlen/(r1-r0);            /* apex of the cone, line.h:395-397 */
    f[9] = v0.x + tt*d[0]; f[10] = v0.y + tt*d[1]; f[11] = v0.z + tt*d[2];
    f[13] = dlen*dlen/((r1-r0)*(r1-r0) + dlen*dlen);
  }
  p.type = MI_PRIM_LINE;
}

/* shading-side constants of a line (DPrimGeo.f[0..15], [26..33]; line.h:123-161) from its end points and the packed record */
MI_HD void line_shading_consts(float *g, const DPrim &p, const V3 v0, const V3 v1)
{
  V3 d = sub3(v1, v0);
  const float ilen_d = 1.0f/mi_sqrt(dot3(d, d));
  d = scale3(d, ilen_d);
  V3 a, b; get_onb(d, a, b);
  V3 ac, bc; get_onb(mk3(p.v[2][0], p.v[2][1], p.v[2][2]), ac, bc);    /* dwords 6..8: the intersection's unit axis */
  g[0] = d.x; g[1] = d.y; g[2] = d.z; g[3] = ilen_d;
  g[4] = a.x; g[5] = a.y; g[6] = a.z; g[7] = b.x; g[8] = b.y; g[9] = b.z;
  g[10] = ac.x; g[11] = ac.y; g[12] = ac.z; g[13] = bc.x; g[14] = bc.y; g[15] = bc.z;
  g[26] = v1.x; g[27] = v1.y; g[28] = v1.z; g[29] = v0.x; g[30] = v0.y; g[31] = v0.z;
  g[32] = p.v[1][0]; g[33] = p.v[1][1];           /* r0, r1 (dwords 3, 4 of the line record) */
}

/* a moving sphere / line (DPrim type 0, pad[0] = 1 / 2: shutter-open centre / end points in v[0], v[1], radii in v[2][0..1];
 * shutter-close positions in t1.v[0], v[1]) as the static record it is at `time` (geo_get_vertex_time; radii stay those of
 * the shutter-open vertices, sphere.h:7-11, line.h:10-16) */
__device__ __forceinline__ DPrim moving_analytic_at(const DPrim &p, const DPrimT1 &t1, float time, V3 &v0, V3 &v1)
{
  const float w0 = 1.0f - time, w1 = time;
  v0 = mk3(w0*p.v[0][0] + w1*t1.v[0][0], w0*p.v[0][1] + w1*t1.v[0][1], w0*p.v[0][2] + w1*t1.v[0][2]);
  v1 = mk3(w0*p.v[1][0] + w1*t1.v[1][0], w0*p.v[1][1] + w1*t1.v[1][1], w0*p.v[1][2] + w1*t1.v[1][2]);
  DPrim r;
  if(p.pad[0] == MI_PRIM_SPHERE)
  {
    for(int k=0;k<16;k++) (&r.v[0][0])[k] = 0.0f;
    r.v[0][0] = v0.x; r.v[0][1] = v0.y; r.v[0][2] = v0.z; r.v[1][0] = p.v[2][0];
    r.type = MI_PRIM_SPHERE;
  }
  else pack_line(r, v0, v1, p.v[2][0], p.v[2][1]);
  return r;
}

struct TraceState;
template<bool MB>
__device__ __forceinline__ void analytic_intersect(const DPrim *prims, uint32_t prim, const V3 o, const V3 d, uint32_t ignore, Hit &hit,
                                                   float time, const DPrimT1 *prims_t1)
{ /* prims_intersect for spheres and lines, src/prims.c:665-668; and for every primitive the leaf loops put off (DPrim.type 0) */
  /* the whole 64-B record in four 16-B loads up front (one memory round trip), then registers only */
  const float4 *q = (const float4 *)(prims + prim);
  const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  DPrim p;
  p.v[0][0] = q0.x; p.v[0][1] = q0.y; p.v[0][2] = q0.z; p.v[1][0] = q0.w;
  p.v[1][1] = q1.x; p.v[1][2] = q1.y; p.v[2][0] = q1.z; p.v[2][1] = q1.w;
  p.v[2][2] = q2.x; p.v[3][0] = q2.y; p.v[3][1] = q2.z; p.v[3][2] = q2.w;
  p.type = __float_as_uint(q3.x); p.pad[0] = __float_as_uint(q3.y); p.pad[1] = __float_as_uint(q3.z); p.pad[2] = __float_as_uint(q3.w);
  if(p.type == 0 && p.pad[1] == MI_PRIM_ORDERED)
  { /* a static triangle / quad that has to wait for the primitives put off before it in its leaf (mi_mark_ordered_kernel) */
    if(prim == ignore) return;
    PrimRegs r;
    r.q0 = q0; r.q1 = q1; r.q2 = q2; r.q3 = q3;
    triquad_intersect(r, p.pad[0], o, d, hit, prim);
    return;
  }
  if(MB && p.type == 0 && p.pad[0] < MI_PRIM_TRI)
  { /* moving sphere / cone / cylinder: the static record at the ray's time, then the usual tests below */
    V3 a0, a1;
    p = moving_analytic_at(p, prims_t1[prim], time, a0, a1);
  }
  if(MB && p.type == 0)
  { /* motion-blurred triangle / quad: the record holds the shutter-open vertices, prims_t1 the shutter-close ones.
       geo_get_vertex_time (include/geo.h:120-138): (1-t) v(open) + t v(close) per component, then the usual test on the
       edges of the interpolated vertices (prims_intersect, src/prims.c:645-663) */
    if(prim == ignore) return;
    const float4 *q1p = (const float4 *)(prims_t1 + prim);
    const float4 c0 = q1p[0], c1 = q1p[1], c2 = q1p[2];
    const float w0 = 1.0f - time, w1 = time;
    const V3 v0 = mk3(w0*q0.x + w1*c0.x, w0*q0.y + w1*c0.y, w0*q0.z + w1*c0.z);
    const V3 v1 = mk3(w0*q0.w + w1*c0.w, w0*q1.x + w1*c1.x, w0*q1.y + w1*c1.y);
    const V3 v2 = mk3(w0*q1.z + w1*c1.z, w0*q1.w + w1*c1.w, w0*q2.x + w1*c2.x);
    const V3 v3 = mk3(w0*q2.y + w1*c2.y, w0*q2.z + w1*c2.z, w0*q2.w + w1*c2.w);
    const V3 e1 = sub3(v1, v0), e2 = sub3(v2, v0), e3 = sub3(v3, v0);
    PrimRegs r;
    r.q0 = make_float4(v0.x, v0.y, v0.z, e1.x); r.q1 = make_float4(e1.y, e1.z, e2.x, e2.y); r.q2 = make_float4(e2.z, e3.x, e3.y, e3.z); r.q3 = q3;
    triquad_intersect(r, p.pad[0], o, d, hit, prim);
    return;
  }
  if(p.type == MI_PRIM_SPHERE)
  { /* geo_sphere_intersect, include/geo/sphere.h:146-166; u,v are recomputed at shading time */
    const float t = sphere_t(ld3(p.v[0]), p.v[1][0], o, d);
    if(t > 0.0f && t < hit.dist) { hit.dist = t; hit.prim = prim; }
  }
  else if(p.type == MI_PRIM_LINE) line_intersect(p, o, d, hit, prim, ignore);
}

/* ------------------------------------------------------------------------------------------ traversal */
#ifndef MI_STACK_LDS
#define MI_STACK_LDS 12  /* entries per lane of the LDS stack area (mi_abi.hip: MI_STACK = what traversal may use of them) */
#endif
struct Lds
{
  /* node records (mi_device.h): a link carries the record's OFFSET in 16-byte lanes, valid for both homes */
  const float4 *nodes_adj;  /* HBM / L2: the record at offset L starts at nodes_adj + L */
  uint32_t nodes_lds_addr;  /* LDS byte address of the staged records: the one at offset L < top_off starts at nodes_lds_addr + 16 L */
  bool nodes_in_lds;        /* the WHOLE tree is staged: compile-time constant after inlining (lds_setup<NODES_LDS>) */
  uint32_t top_off;         /* offsets below this are staged in LDS (may be 0) */
  uint32_t root;            /* link of node 0 (offset 0 | its split axes << MI_AXES_SHIFT) */
  bool has_t1;              /* the records carry the child boxes at shutter close (motion-blur kernels read them) */
  uint2 *stack;             /* [STACK][BLOCK] in LDS, this thread's column */
  uint2 *overflow;          /* [extra][total threads] in HBM: entries beyond STACK (rare). The workgroup's row; the thread's column is
                               added where it is used, so that no per-thread 64-bit pointer lives in registers through the kernel */
  uint32_t overflow_stride;
  uint32_t num_nodes;
  unsigned char *jobs;      /* leaf_jobs: this wave's list of MI_JOBS_MAX source lanes, in LDS behind the stacks */
};
#define MI_JOBS_MAX 256     /* primitive tests one wave deals out per round at most (64 lanes x 4); more: sequential leaf loop */
#define MI_JOBS_LDS 512     /* bytes of a wave's job list in LDS (the FAST rounds deal out up to MI_SPEC_JOBS_MAX = 512 tests) */
#define MI_JOB_SLOTS 3      /* stack entries of a lane's LDS column that leaf_jobs uses for results: best (8 B), uv (8 B), analytic mask */

template<int BLOCK, int STACK>
__device__ __forceinline__ void stack_push(const Lds &lds, int sp, uint2 e)
{
  if(sp < STACK) lds.stack[sp*BLOCK] = e;
  else lds.overflow[(size_t)(sp - STACK)*lds.overflow_stride + threadIdx.x] = e;
}
typedef unsigned int mi_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) mi_u32x2 lds_uint2;   /* typed LDS pointer: ds_read/ds_write instead of flat */
typedef float mi_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const mi_f32x4 lds_f32x4;
template<int BLOCK, int STACK>
__device__ __forceinline__ uint2 stack_top(const Lds &lds, const lds_uint2 *lstack, int sp)
{
  if(sp < STACK) { const mi_u32x2 v = lstack[sp*BLOCK]; return make_uint2(v.x, v.y); }
  return lds.overflow[(size_t)(sp - STACK)*lds.overflow_stride + threadIdx.x];
}

/* Workgroup prologue shared by all traversal kernels. NODES_LDS: the BVH is staged into LDS once per workgroup
 * (coalesced 16-B loads) in front of the traversal stacks; otherwise it does not fit next to the stacks and is read from
 * HBM / L2 through the same SoA layout (mi_device.h), LDS holds the stacks only. Call from all threads (barrier inside). */
/* ptdl: the one-burst emitter records (DLight) of up to MI_LIGHTS_LDS emitter primitives sit in LDS, behind the Halton head */
#ifndef MI_LIGHTS_LDS
#define MI_LIGHTS_LDS 16
#endif
template<bool HALTON>
__device__ __forceinline__ const float4 *lights_lds()
{
  extern __shared__ __attribute__((aligned(16))) unsigned char mi_dynamic_lds[];
  return (const float4 *)(mi_dynamic_lds + (HALTON ? 2*MI_HALTON_LDS : 0));
}

#ifndef MI_STACK_LDS_MB
#define MI_STACK_LDS_MB 7   /* entries per lane of the motion-blur kernels' stack columns: the five entries less make room for the nodes' second box set */
#endif
template<int BLOCK, bool NODES_LDS, bool HALTON = false, bool LIGHTS = false, int COLUMN = MI_STACK_LDS, bool T1 = false>
__device__ __forceinline__ Lds lds_setup(const DScene &sc, unsigned char *smem, uint2 *stack_overflow)
{
  const uint32_t N = sc.num_nodes;
  Lds lds;
  uint2 *lds_stack;
  if(HALTON && MI_HALTON_LDS)
  { /* the hot head of the Halton permutation tables first (halton_lds()), tree and stacks behind it */
    uint32_t *dst = (uint32_t *)smem;
    const uint32_t *src = (const uint32_t *)sc.halton_perm;
    for(uint32_t i=threadIdx.x;i<MI_HALTON_LDS/2;i+=BLOCK) dst[i] = src[i];
    smem += 2*MI_HALTON_LDS;
    if(!NODES_LDS) __syncthreads();
  }
  if(LIGHTS && MI_LIGHTS_LDS)
  {
    if(sc.lights && sc.num_lights <= MI_LIGHTS_LDS)
      for(uint32_t i=threadIdx.x;i<sc.num_lights*(uint32_t)(sizeof(DLight)/16);i+=BLOCK) ((float4 *)smem)[i] = ((const float4 *)sc.lights)[i];
    smem += MI_LIGHTS_LDS*sizeof(DLight);
    if(!NODES_LDS) __syncthreads();
  }
  { /* the top of the tree (all of it in the NODES_LDS instantiations): the first K records, pad lanes dropped -- 7 of 8 lanes, with the
       shutter-close boxes 13 of 16 (which only the motion-blur kernels stage: T1) */
    const uint32_t K = NODES_LDS ? N : sc.nodes_lds;
    const bool t1 = sc.nodes_t1 != 0u;
    const uint32_t SH = t1 ? 2u*MI_NODE_STRIDE : MI_NODE_STRIDE;
    const uint32_t SL = t1 ? MI_NODE_FIELDS + MI_NODE_T1_FIELDS : MI_NODE_FIELDS;      /* the links are baked for these strides (mi_bake_links_kernel) */
    mi_f32x4 *lds_nodes = (mi_f32x4 *)smem;
    lds_stack = (uint2 *)(smem + (size_t)SL*K*16);
    const mi_f32x4 *src = (const mi_f32x4 *)sc.nodes;
    if(!t1) for(uint32_t i=threadIdx.x;i<MI_NODE_FIELDS*K;i+=BLOCK) { const uint32_t n = i/MI_NODE_FIELDS, f = i - n*MI_NODE_FIELDS; lds_nodes[i] = src[(size_t)n*MI_NODE_STRIDE + f]; }
    else for(uint32_t i=threadIdx.x;i<(MI_NODE_FIELDS + MI_NODE_T1_FIELDS)*K;i+=BLOCK)
    { /* (kernels that never read the second box set still keep the records' layout: the links are baked for it) */
      const uint32_t n = i/(MI_NODE_FIELDS + MI_NODE_T1_FIELDS), f = i - n*(MI_NODE_FIELDS + MI_NODE_T1_FIELDS);
      lds_nodes[i] = src[(size_t)n*2u*MI_NODE_STRIDE + (f < MI_NODE_FIELDS ? f : f + (MI_NODE_T1_HBM - MI_NODE_T1_LDS))];
    }
    __syncthreads();
    lds.nodes_in_lds = NODES_LDS; lds.top_off = K*SL;
    lds.nodes_adj = sc.nodes + (size_t)K*(SH - SL);
    lds.nodes_lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    lds.has_t1 = T1 && t1;
  }
  lds.stack = lds_stack + threadIdx.x; lds.num_nodes = N; lds.root = sc.root_link;
  lds.jobs = (unsigned char *)(lds_stack + (size_t)COLUMN*BLOCK) + (threadIdx.x >> 6)*MI_JOBS_LDS;
  lds.overflow_stride = gridDim.x*BLOCK;
  lds.overflow = stack_overflow + (size_t)blockIdx.x*BLOCK;
  return lds;
}

#ifndef MI_TAIL_INNER
#define MI_TAIL_INNER 6      /* end of round 3, with MI_LEAF_EXIT 64: 3 / 4 / 6 / 8 = 35.85 / 35.55 / 35.39 / 35.45 ms on cfg 3, 18.80 / 18.64 / 18.55 / 18.55 on cfg 2 (exact rounds) */
#endif
#ifndef MI_LEAF_EXIT
#define MI_LEAF_EXIT 64   /* leaf_jobs kernels: the node loop of a round ends once this many lanes wait with a leaf; 64 = never. Round 2: 48 (a full pass of jobs;
                             A/B 24..56); re-swept at the end of round 3 (32 / 40 / 48 / 56 / 64: 36.10 / 36.09 / 35.77 / 35.74 / 35.30-35.55 ms on cfg 3): the rule is off */
#endif
#ifndef MI_LEAF_JOBS
#define MI_LEAF_JOBS 1   /* 1: the primitive tests of a round are dealt out over all 64 lanes of the wave (leaf_jobs) */
#endif


struct TraceState
{ /* resumable traversal of one ray: survives between rounds so that a wave can re-fill idle lanes in between */
  int sp;
  uint32_t current;
  bool done;
  bool anyhit;           /* shadow ray that may end at the first occluder (MI_LIGHT_ANYHIT); only read by the ANYHIT instantiations */
  float idx, idy, idz;   /* 1/dir, computed once per ray (qbvhmp.c:1291-1295) */
  /* motion blur (MB instantiations only): the ray's time and the shutter-close records */
  float time;
  const DPrimT1 *prims_t1;
};

template<class CNT>
__device__ __forceinline__ void trace_begin(const Lds &lds, TraceState &ts, const V3 d, CNT &cnt)
{
  MI_COUNT(cnt, 0, 1);
  ts.idx = mi_rcp(d.x); ts.idy = mi_rcp(d.y); ts.idz = mi_rcp(d.z);
  ts.sp = 0;
  ts.current = lds.root;      /* node 0 = root, with its split axes */
  ts.done = false;
  ts.anyhit = false;
}

/* what follows a leaf in accel_intersect: pop the next subtree that starts in front of the closest hit (qbvhmp.c:1357-1364,1380-1386) */
#ifndef MI_POP_PREFETCH
#define MI_POP_PREFETCH 1     /* cfg 2 16.17 -> 16.07 ms, cfg 3 28.82 -> 28.67 (same-box A/B, gpurun_out/ab_pre.txt) */
#endif
/* (top / has_top: the stack's top entry, read by the caller earlier -- together with the leaf's results, so that the pop does not start a
   second round trip through LDS when the results have arrived; only entries that live in LDS are read ahead) */
template<int BLOCK, int STACK, bool ANYHIT>
__device__ __forceinline__ void leaf_finish(const Lds &lds, const Hit &hit, TraceState &ts, const mi_u32x2 top = mi_u32x2{0u, 0u}, bool has_top = false)
{
  lds_uint2 *lstack = (lds_uint2 *)lds.stack;
  int sp = ts.sp;
  uint32_t current = MI_LEAF32;
  bool done = true;
  if(ANYHIT && ts.anyhit && hit.prim != MI_NOPRIM) sp = 0;   /* an occluder is all a shadow ray needs to know (MI_LIGHT_ANYHIT) */
  bool first = has_top;
  while(sp > 0)
  {
    sp--;
    const uint2 e = first ? make_uint2(top.x, top.y) : stack_top<BLOCK, STACK>(lds, lstack, sp);
    first = false;
    if(!(__uint_as_float(e.y) > hit.dist)) { current = e.x; done = false; break; }
  }
  ts.sp = sp; ts.current = current; ts.done = done;
}

/* the leaf `current`, primitive by primitive on this lane (qbvhmp.c:1366-1379) */
template<bool MB, class CNT>
__device__ __forceinline__ void leaf_tests(const DPrim *prims, uint32_t current, const V3 o, const V3 d, uint32_t ignore,
                                           Hit &hit, const TraceState &ts, CNT &cnt)
{
  const uint32_t idxp = (current ^ MI_LEAF32) >> 5;
  const uint32_t num = current & 31u;
  /* triangles and quads first (software pipelined: the next primitive's 64 B are in flight while this one is
     intersected); spheres / cones / cylinders of this leaf are remembered and intersected afterwards, so that the
     wave runs that rare, long code once per leaf round instead of once per primitive slot. Every primitive of the
     leaf is still tested exactly once against the running closest hit (prims_intersect, src/prims.c:638-672), and the
     closest hit does not depend on that order (at equal distance a triangle's `<=` beats a sphere's / line's `<` from
     either side) -- with one exception: a folded quad crossed in both halves reports the half the RUNNING distance lets
     through (src/prims.c:654-663). Where that can matter -- a non-planar quad behind a primitive that is put off -- the
     quad and everything behind it in the leaf is put off as well (type 0 + MI_PRIM_ORDERED, set at upload by
     mi_mark_ordered_kernel), so that leaf is worked through in the reference's order. */
  uint32_t analytic = 0;
  /* two record buffers in ping-pong: the load of primitive i+1 is in flight while i is intersected, and no
     16-register copy is needed per iteration */
#define MI_LEAF_STEP(R, I) { MI_COUNT(cnt, 3, 1); \
    const uint32_t type = __float_as_uint((R).q3.x); \
    if(type >= MI_PRIM_TRI) { if(idxp + (I) != ignore) triquad_intersect((R), type, o, d, hit, idxp + (I)); }   /* triangle.h:271 */ \
    else analytic |= 1u << (I); }
  PrimRegs ra = prim_load(prims, num ? idxp : 0), rb;      /* rb is loaded before each use (same condition) */
  for(uint32_t i=0;i<num;i+=2)
  {
#ifdef MI_PROFILE_LOOPS
    { const unsigned nl = __popcll(__ballot(1)); if(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) cnt.c[9] += MI_PROFILE_LOOPS == 2 ? 2*nl : 2; }   /* wave-level leaf slots (2: lane slots of lanes still in their leaf) */
#endif
    if(i + 1 < num) rb = prim_load(prims, idxp + i + 1);
    MI_LEAF_STEP(ra, i)
    if(i + 1 < num)
    {
      if(i + 2 < num) ra = prim_load(prims, idxp + i + 2);
      MI_LEAF_STEP(rb, i + 1)
    }
  }
#undef MI_LEAF_STEP
  while(analytic)
  {
#ifdef MI_PROFILE_LOOPS
    { const unsigned nl = __popcll(__ballot(1)); if(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) cnt.c[10] += MI_PROFILE_LOOPS == 2 ? nl : 1; }   /* wave-level analytic passes (2: lanes in them) */
#endif
    const uint32_t i = __ffs(analytic) - 1;
    analytic &= analytic - 1;
    analytic_intersect<MB>(prims, idxp + i, o, d, ignore, hit, MB ? ts.time : 0.0f, MB ? ts.prims_t1 : nullptr);
  }
}

/* the leaf a lane holds (ts.current) on that lane, then the pop */
template<int BLOCK, int STACK, bool MB, bool ANYHIT, class CNT>
__device__ __forceinline__ void leaf_sequential(const Lds &lds, const DPrim *prims, const V3 o, const V3 d, uint32_t ignore,
                                                Hit &hit, TraceState &ts, CNT &cnt)
{
  leaf_tests<MB>(prims, ts.current, o, d, ignore, hit, ts, cnt);
  leaf_finish<BLOCK, STACK, ANYHIT>(lds, hit, ts);
}

/* Distributed leaf phase (MI_LEAF_JOBS). In the sequential leaf loop a round costs the wave as many test slots as its LONGEST leaf
 * has primitives (6 in the reference's trees), while the average lane holds 1.8 -- 31 % of the slots test something. Here every
 * (lane, primitive) pair of the round becomes a job and the jobs are dealt out over ALL 64 lanes, also those whose own ray is
 * finished or still at an inner node:
 *   owners        lanes that hold a leaf: num = primitives in it, exclusive prefix sum over the wave (ballots of the bits of num)
 *   job list      jobs[prefix + k] = owner lane, one byte per job, in the wave's LDS list
 *   worker lane j takes job j: fetches the owner's ray (ds_bpermute: origin, direction, closest distance at the start of the leaf,
 *                 ignored primitive, leaf link), the primitive's record, runs the same triquad_intersect
 *   results       a hit goes to the owner's slot `best` with ds_min_u64 on (distance bits << 32 | 31 - k): the closest wins and,
 *                 at equal distance, the later primitive -- what the sequential `t <= hit.dist` does; the winner leaves u, v
 *   owners        read best / uv, run the sphere / cone / cylinder tests of the leaf (mask gathered by the workers), pop.
 * Per ray the same primitives are tested against the same ray with the same arithmetic; the closest hit a leaf yields is that of
 * the sequential loop except where its result depends on the ORDER of the running distance: a quad both of whose halves are
 * crossed (triquad_intersect<true>) -- such a test poisons the owner's slot and the owner runs the sequential loop.
 * Counters are unchanged (every primitive of the leaf counts once). */
typedef unsigned long long mi_u64;
typedef __attribute__((address_space(3))) mi_u64 lds_u64;
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template<int BLOCK, int STACK, bool MB, bool ANYHIT, class CNT>
__device__ __forceinline__ void leaf_jobs(const Lds &lds, const DPrim *prims, const V3 o, const V3 d, uint32_t ignore,
                                          Hit &hit, TraceState &ts, bool busy, CNT &cnt)
{
  const unsigned lane = __lane_id();
  const bool own = busy && !ts.done && (ts.current & MI_LEAF32);
  if(!__any(own)) return;
  const uint32_t cur = ts.current;
  const uint32_t num = own ? cur & 31u : 0u;
  /* exclusive prefix sum of num over the wave from the ballots of its bits (leaves of the reference's trees hold <= 6) */
  const mi_u64 b0 = __ballot(num & 1u), b1 = __ballot(num & 2u), b2 = __ballot(num & 4u);
  const uint32_t J = __popcll(b0) + 2u*__popcll(b1) + 4u*__popcll(b2);
  /* leaves longer than the prefix sum covers, or more jobs than the list holds: every owner runs the per-lane loop (below) */
  const bool fits = !__any(num > 7u) && J <= MI_JOBS_MAX;
#define MI_MBCNT(M) __builtin_amdgcn_mbcnt_hi((uint32_t)((M) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(M), 0u))
  const uint32_t prefix = MI_MBCNT(b0) + 2u*MI_MBCNT(b1) + 4u*MI_MBCNT(b2);
#undef MI_MBCNT
  lds_u8 *jobs = (lds_u8 *)lds.jobs;
  lds_uint2 *col = (lds_uint2 *)lds.stack;                    /* this lane's LDS column; lane s of the wave: col + (s - lane) */
  lds_u64 *best = (lds_u64 *)(col + (STACK + 0)*BLOCK);
  lds_uint2 *uvs = col + (STACK + 1)*BLOCK;
  lds_u32 *anl = (lds_u32 *)(col + (STACK + 2)*BLOCK);
  if(own && fits)
  {
    for(uint32_t k=0;k<num;k++) jobs[prefix + k] = (unsigned char)lane;
    *best = ((mi_u64)__float_as_uint(hit.dist) << 32) | 0xffffffffull;
    *anl = 0u;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  MI_TT(cnt, 1)
  for(uint32_t base=0;fits && base<J;base+=64u)
  {
#ifdef MI_PROFILE_LOOPS
    if(lane == 0) cnt.c[9] += MI_PROFILE_LOOPS == 2 ? (J - base < 64u ? J - base : 64u) : 1;   /* wave-level test slots (2: jobs in them) */
#endif
    const uint32_t j = base + lane;
    const bool valid = j < J;
    const int src = valid ? (int)jobs[j] : (int)lane;
    const uint32_t scur = (uint32_t)__shfl((int)cur, src), spre = (uint32_t)__shfl((int)prefix, src);
    const uint32_t k = j - spre, prim = valid ? ((scur ^ MI_LEAF32) >> 5) + k : 0u;
    /* the record is needed whatever its type turns out to be: all four 16-byte loads are issued at once, and the owner's ray
       arrives (ds_bpermute) while they are in flight. The test itself is branch-free; what it found counts if the job is real. */
    const PrimRegs rec = prim_load(prims, prim);
    const V3 so = mk3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src)), sd = mk3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
    Hit h;
    h.prim = MI_NOPRIM; h.u = h.v = 0.0f;
    h.dist = __shfl(hit.dist, src);
    const uint32_t sign = (uint32_t)__shfl((int)ignore, src);
    const int off = src - (int)lane;
    uint32_t type = __float_as_uint(rec.q3.x);
    PrimRegs use = rec;
#ifndef MI_MB_MOVING_JOBS
#define MI_MB_MOVING_JOBS 1
#endif
    if(MB && MI_MB_MOVING_JOBS)
    { /* motion-blur kernels: a moving triangle / quad (type 0, pad[0] = vertex count; the record holds the shutter-open VERTICES, prims_t1 the shutter-close
         ones) is a job like a static one -- its vertices at the OWNER's time (geo_get_vertex_time, include/geo.h:120-138: (1-t) v(open) + t v(close) per
         component), then the edges and the usual test, the arithmetic of analytic_intersect. So is a static triangle / quad that mi_mark_ordered_kernel put
         off behind it (pad[1] = MI_PRIM_ORDERED, plain record): what the order of the tests can change -- a quad crossed in both halves -- poisons the
         owner's slot below and the owner works through its leaf in the reference's order; every other result is the minimum over the leaf, later wins a tie. */
      const uint32_t pad0 = __float_as_uint(rec.q3.y), pad1 = __float_as_uint(rec.q3.z);
      const bool put_off = type == 0u && pad0 >= MI_PRIM_TRI;
      const float stime = __shfl(ts.time, src);          /* (outside the branch: ds_bpermute only reads lanes that execute it) */
      if(put_off && pad1 != MI_PRIM_ORDERED)
      {
        const float4 *q1p = (const float4 *)(ts.prims_t1 + prim);
        const float4 c0 = q1p[0], c1 = q1p[1], c2 = q1p[2];
        const float w0 = 1.0f - stime, w1 = stime;
        const V3 v0 = mk3(w0*rec.q0.x + w1*c0.x, w0*rec.q0.y + w1*c0.y, w0*rec.q0.z + w1*c0.z);
        const V3 v1 = mk3(w0*rec.q0.w + w1*c0.w, w0*rec.q1.x + w1*c1.x, w0*rec.q1.y + w1*c1.y);
        const V3 v2 = mk3(w0*rec.q1.z + w1*c1.z, w0*rec.q1.w + w1*c1.w, w0*rec.q2.x + w1*c2.x);
        const V3 v3 = mk3(w0*rec.q2.y + w1*c2.y, w0*rec.q2.z + w1*c2.z, w0*rec.q2.w + w1*c2.w);
        const V3 e1 = sub3(v1, v0), e2 = sub3(v2, v0), e3 = sub3(v3, v0);
        use.q0 = make_float4(v0.x, v0.y, v0.z, e1.x); use.q1 = make_float4(e1.y, e1.z, e2.x, e2.y); use.q2 = make_float4(e2.z, e3.x, e3.y, e3.z);
      }
      if(put_off) type = pad0;
    }
    const bool both = triquad_intersect<true>(use, type, so, sd, h, prim);
    const bool tq = valid && type >= MI_PRIM_TRI && prim != sign;       /* triangle.h:271 */
    const bool cand = tq && !both && h.prim != MI_NOPRIM;
    const mi_u64 key = ((mi_u64)__float_as_uint(h.dist) << 32) | (mi_u64)(31u - k);
    if(tq && (both || cand)) __hip_atomic_fetch_min(best + off, both ? (mi_u64)0 : key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   /* 0: poison, the owner goes sequential */
    if(valid && type < MI_PRIM_TRI) __hip_atomic_fetch_or(anl + 2*off, 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if(__any(cand))
    { /* the job that holds the owner's minimum so far leaves its u, v (a later, closer one overwrites them) */
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if(cand && best[off] == key) uvs[off] = mi_u32x2{__float_as_uint(h.u), __float_as_uint(h.v)};
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  MI_TT(cnt, 2)
  if(own)
  {
    const bool has_top = MI_POP_PREFETCH && ts.sp > 0 && ts.sp <= STACK;
    mi_u32x2 top = mi_u32x2{0u, 0u};
    if(has_top) top = ((lds_uint2 *)lds.stack)[(ts.sp - 1)*BLOCK];       /* the pop's first read travels with the results */
    const mi_u64 res = fits ? *best : 0;
    if(res == 0)
    { /* a folded quad crossed twice: what it yields depends on the running distance -- this lane's leaf in the reference's order */
      leaf_sequential<BLOCK, STACK, MB, ANYHIT>(lds, prims, o, d, ignore, hit, ts, cnt);
    }
    else
    {
      MI_COUNT(cnt, 3, num);
      const uint32_t idxp = (cur ^ MI_LEAF32) >> 5;
      if((uint32_t)res != 0xffffffffu)
      {
        const mi_u32x2 uv = *uvs;
        hit.dist = __uint_as_float((uint32_t)(res >> 32)); hit.prim = idxp + (31u - (uint32_t)res);
        hit.u = __uint_as_float(uv.x); hit.v = __uint_as_float(uv.y);
      }
      uint32_t analytic = *anl;
      while(analytic)
      {
#ifdef MI_PROFILE_LOOPS
        { const unsigned nl = __popcll(__ballot(1)); if(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) cnt.c[10] += MI_PROFILE_LOOPS == 2 ? nl : 1; }
#endif
        const uint32_t i = __ffs(analytic) - 1;
        analytic &= analytic - 1;
        analytic_intersect<MB>(prims, idxp + i, o, d, ignore, hit, MB ? ts.time : 0.0f, MB ? ts.prims_t1 : nullptr);
      }
      leaf_finish<BLOCK, STACK, ANYHIT>(lds, hit, ts, top, has_top);
    }
  }
  MI_TT(cnt, 3)
}

/* the ray geometry a node visit needs, fixed for the length of a round */
#ifndef MI_HYBRID
#define MI_HYBRID 1       /* 0 (experiments): the NODES_LDS = false instantiations read every node from HBM / L2, as in rounds 1-4 */
#endif
struct RayBox
{
  uint32_t nearbits, offx, offy, offz;   /* sign bits of the direction; byte offset of each axis' entry plane inside a node record: 48 (the upper planes, lanes 3..5) for negative directions */
  float idx, idy, idz;
  bool slow;                             /* a lane of the wave has an infinite 1/dir: literal SSE-semantics slab test for the whole wave */
  /* FMA slabs (FAST rounds only, MI_SPEC_FMA): -o/d per axis, and the absolute part of the slack of the box test */
  float nox, noy, noz, slack;
  float w0, w1;                          /* motion-blur kernels: 1 - time, time of the ray (weights of the shutter-open / -close boxes) */
};
#ifndef MI_PUSH_BLIND
#define MI_PUSH_BLIND 1     /* cfg 2 16.12 -> 16.00 ms (FAST 16.91 -> 16.67), cfg 3 29.35 -> 29.27: 127 -> 102 scalar instructions per node visit (tools/valu_floor.py) */
#endif
#ifndef MI_SPEC_FMA
#define MI_SPEC_FMA 0      /* 0: off. 1: in the ptdl FAST kernel and the ray-level test kernel, 2: in every FAST kernel -- cfg 2 17.71-17.79 against 18.01 ms,
                              cfg 3 (FAST rounds) 37.1 against 38.0; but the relaxed test lets through boxes the reference's test rejects, and a
                              primitive in such a box can be hit: 0.25 paths per million then carry a hit the reference does not have (same-library
                              comparison of the FAST against the exact rounds, tests/dev/fast_vs_exact.py). Results first: off */
#endif
#define MI_FMA_REL 1.0000019073486328125f     /* 1 + 2^-19: relative part of the slack (16 x the 2^-24 unit roundoff both sides can be off by) */
template<bool FMA = false>
__device__ __forceinline__ RayBox raybox_setup(const V3 o, const V3 d, const TraceState &ts, uint32_t N)
{
  RayBox rb;
  const uint32_t near_x = __float_as_uint(d.x) >> 31, near_y = __float_as_uint(d.y) >> 31, near_z = __float_as_uint(d.z) >> 31;
  rb.nearbits = near_x | (near_y << 1) | (near_z << 2);
  rb.idx = ts.idx; rb.idy = ts.idy; rb.idz = ts.idz;
  rb.w0 = 1.0f - ts.time; rb.w1 = ts.time;
  rb.offx = near_x ? 48u : 0u; rb.offy = near_y ? 48u : 0u; rb.offz = near_z ? 48u : 0u;       /* bytes: three lanes from the lower to the upper plane of an axis */
  if(FMA)
  { /* huge 1/dir (a component of the direction below 1e-30) go the literal way too: plane/d - o/d may be inf - inf there */
    rb.slow = __any(!(fabsf(rb.idx) < 1e30f) || !(fabsf(rb.idy) < 1e30f) || !(fabsf(rb.idz) < 1e30f));
    const float ox = o.x*rb.idx, oy = o.y*rb.idy, oz = o.z*rb.idz;
    rb.nox = -ox; rb.noy = -oy; rb.noz = -oz;
    rb.slack = 4.76837158203125e-07f*fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));     /* 2^-21 max |o/d| */
  }
  else
  {
    rb.slow = __any(isinf(rb.idx) || isinf(rb.idy) || isinf(rb.idz));
    rb.nox = rb.noy = rb.noz = rb.slack = 0.0f;
  }
  return rb;
}

/* one inner node of accel_intersect (src/accel.d/qbvhmp.c:1188-1246,1313-1354): the four child boxes against the ray clipped to
 * `dist`, front-to-back order from split axes and ray signs; the nearest hit child becomes `current`, the others are pushed
 * far-first. Returns false if no child is hit (the caller pops). */
template<int BLOCK, int STACK, bool FMA = false, bool MB = false, class CNT, class MISS>
__device__ __forceinline__ bool node_visit(const Lds &lds, lds_uint2 *lstack, const RayBox &rb, const V3 o, float dist, uint32_t &current, int &sp, CNT &cnt, MISS &&miss,
                                           float *entry = nullptr)      /* entry: receives the entry distance of the child that becomes `current` */
{ /* miss(): what the caller does when no child is hit, as the else branch of the test (the exact rounds pop right there -- as a second
     conditional after the call the same code compiled to a node loop with four more divergent regions, +6 % branches) */ /* FMA (FAST rounds): a plane's distance as fma(plane, 1/d, -o/d) instead of (plane - o)*(1/d) -- one instruction instead of two,
     24 fewer per visit. The two differ by at most 2^-24 (3 |t| + |o/d|) (one rounding of o/d up front instead of one of the
     difference), so the box counts as hit when lo <= hi (1 + 2^-19) + 2^-21 max|o/d|: every box the reference's test passes passes
     here (and a few more: counted work, not results), and the entry distances pushed with the subtrees are compared with the same
     slack when they are popped (cull_distance). */
  const uint32_t nearbits = rb.nearbits;
  const float idx = rb.idx, idy = rb.idy, idz = rb.idz;
  const bool slow = rb.slow;
  const uint32_t ax = (current >> MI_AXES_SHIFT) & 63u;        /* the node's split axes travel in the link (mi_bake_links_kernel) */
  /* The node's record: seven 16-byte lanes at compile-time offsets from ONE address. bx / by / bz = byte offset of the lane that holds
     the ray's ENTRY plane of each axis (0: the lower planes, 48: the upper ones); the exit plane sits 48 bytes to the other side.
     MB, if the tree carries the shutter-close boxes (lds.has_t1): the planes at the ray's time, aabb0 (1 - t) + aabb1 t (qbvhmp.c:1208-1224:
     two products and a sum as there). Rounding is monotonic, so the interpolated lower plane of a box never lies above its upper one and
     the sign-selected slabs below stay what the reference's min / max of the two plane distances evaluate to. */
  float4 nx, ny, nz, fx, fy, fz, child4;
  const float w0 = rb.w0, w1 = rb.w1;
  auto mix = [&](const float4 a_, const float4 b_) { return make_float4(a_.x*w0 + b_.x*w1, a_.y*w0 + b_.y*w1, a_.z*w0 + b_.z*w1, a_.w*w0 + b_.w*w1); };
  auto fetch = [&](uint32_t bx, uint32_t by, uint32_t bz)
  {
    auto from_lds = [&]()
    { /* byte address of the record: a 24-bit multiply reads only the link's offset bits, so the axes in front of them need no mask in the
         chain pop -> address -> LDS read (written as the instruction: the compiler turns a 24-bit multiply by 16 back into shift + mask) */
      uint32_t a;
      asm("v_mad_u32_u24 %0, %1, 16, %2" : "=v"(a) : "v"(current), "s"(lds.nodes_lds_addr));
#define MI_LDS_LANE(ADDR, OFF) ({ const mi_f32x4 q_ = *(lds_f32x4 *)(uintptr_t)((ADDR) + (uint32_t)(OFF)); make_float4(q_.x, q_.y, q_.z, q_.w); })
      const uint32_t ax_ = a + bx, ay_ = a + by, az_ = a + bz, cx_ = a - bx, cy_ = a - by, cz_ = a - bz;
      child4 = MI_LDS_LANE(a, 96);
      nx = MI_LDS_LANE(ax_, 0);  ny = MI_LDS_LANE(ay_, 16); nz = MI_LDS_LANE(az_, 32);
      fx = MI_LDS_LANE(cx_, 48); fy = MI_LDS_LANE(cy_, 64); fz = MI_LDS_LANE(cz_, 80);
      if(MB && lds.has_t1)
      {
        constexpr uint32_t T = 16u*MI_NODE_T1_LDS;
        nx = mix(nx, MI_LDS_LANE(ax_, T + 0));  ny = mix(ny, MI_LDS_LANE(ay_, T + 16)); nz = mix(nz, MI_LDS_LANE(az_, T + 32));
        fx = mix(fx, MI_LDS_LANE(cx_, T + 48)); fy = mix(fy, MI_LDS_LANE(cy_, T + 64)); fz = mix(fz, MI_LDS_LANE(cz_, T + 80));
      }
#undef MI_LDS_LANE
    };
    auto from_hbm = [&]()
    { /* one 128-byte line (two with the shutter-close boxes) */
      const unsigned char *r = (const unsigned char *)(lds.nodes_adj + (current & MI_NODE_MASK));
#define MI_HBM_LANE(P, OFF) (*(const float4 *)((P) + (OFF)))
      const unsigned char *ax_ = r + bx, *ay_ = r + by, *az_ = r + bz, *cx_ = r - bx, *cy_ = r - by, *cz_ = r - bz;
      child4 = MI_HBM_LANE(r, 96);
      nx = MI_HBM_LANE(ax_, 0);  ny = MI_HBM_LANE(ay_, 16); nz = MI_HBM_LANE(az_, 32);
      fx = MI_HBM_LANE(cx_, 48); fy = MI_HBM_LANE(cy_, 64); fz = MI_HBM_LANE(cz_, 80);
      if(MB && lds.has_t1)
      {
        constexpr uint32_t T = 16u*MI_NODE_T1_HBM;
        nx = mix(nx, MI_HBM_LANE(ax_, T + 0));  ny = mix(ny, MI_HBM_LANE(ay_, T + 16)); nz = mix(nz, MI_HBM_LANE(az_, T + 32));
        fx = mix(fx, MI_HBM_LANE(cx_, T + 48)); fy = mix(fy, MI_HBM_LANE(cy_, T + 64)); fz = mix(fz, MI_HBM_LANE(cz_, T + 80));
      }
#undef MI_HBM_LANE
    };
    if(lds.nodes_in_lds) from_lds();
    else if(!MI_HYBRID || !lds.top_off) from_hbm();
    else if((current & MI_NODE_MASK) < lds.top_off) from_lds();      /* the top of the tree is staged, the rest is not (per lane) */
    else from_hbm();
  };
  float tm0, tm1, tm2, tm3;
  mi_u64 M0, M1, M2, M3;    /* child c is hit: lane masks (the compares' own results) in scalar registers, combined by scalar instructions below */
  if(FMA && !slow)
  {
    fetch(rb.offx, rb.offy, rb.offz);
#define SLAB(J, C, TM) { \
    const float lo = fmaxf(fmaxf(fmaxf(__builtin_fmaf(nx.C, idx, rb.nox), __builtin_fmaf(ny.C, idy, rb.noy)), __builtin_fmaf(nz.C, idz, rb.noz)), 0.0f); \
    const float hi = fminf(fminf(fminf(__builtin_fmaf(fx.C, idx, rb.nox), __builtin_fmaf(fy.C, idy, rb.noy)), __builtin_fmaf(fz.C, idz, rb.noz)), dist); \
    TM = lo; J = __ballot(lo <= __builtin_fmaf(hi, MI_FMA_REL, rb.slack)); }
    SLAB(M0, x, tm0) SLAB(M1, y, tm1) SLAB(M2, z, tm2) SLAB(M3, w, tm3)
#undef SLAB
  }
  else if(!slow)
  { /* 4 child slabs, qbvhmp.c:1188-1246. The ray's sign bits pick the entry / exit plane of every slab, which is what
       the reference's min(t0,t1) / max(t0,t1) evaluate to for finite 1/dir and b_min <= b_max (rounding is monotonic;
       empty children are uploaded as [-FLT_MAX, FLT_MAX], see upload_nodes). That leaves v_max3/v_min3 chains
       instead of 48 compare+select pairs (each pair costs a VCC hazard nop on gfx950). */
    fetch(rb.offx, rb.offy, rb.offz);
#define SLAB(J, C, TM) { \
    const float lo = fmaxf(fmaxf(fmaxf((nx.C - o.x)*idx, (ny.C - o.y)*idy), (nz.C - o.z)*idz), 0.0f); \
    const float hi = fminf(fminf(fminf((fx.C - o.x)*idx, (fy.C - o.y)*idy), (fz.C - o.z)*idz), dist); \
    TM = lo; J = __ballot(lo <= hi); }
    SLAB(M0, x, tm0) SLAB(M1, y, tm1) SLAB(M2, z, tm2) SLAB(M3, w, tm3)
#undef SLAB
  }
  else
  { /* a lane of this wave has a zero direction component (1/dir infinite): 0*inf NaNs are possible and the reference's
       SSE min/max semantics (second operand on NaN) decide; evaluate them literally with ordered compares */
    fetch(0u, 0u, 0u);       /* n* = the lower planes, f* = the upper ones */
#define SLAB(J, X0, X1, Y0, Y1, Z0, Z1, TM) { \
    float lo = 0.0f, hi = dist; \
    float t0 = ((X0) - o.x)*idx, t1 = ((X1) - o.x)*idx; \
    float mn = t0 < t1 ? t0 : t1, mx = t0 > t1 ? t0 : t1; \
    lo = lo > mn ? lo : mn; hi = hi < mx ? hi : mx; \
    t0 = ((Y0) - o.y)*idy; t1 = ((Y1) - o.y)*idy; \
    mn = t0 < t1 ? t0 : t1; mx = t0 > t1 ? t0 : t1; \
    lo = lo > mn ? lo : mn; hi = hi < mx ? hi : mx; \
    t0 = ((Z0) - o.z)*idz; t1 = ((Z1) - o.z)*idz; \
    mn = t0 < t1 ? t0 : t1; mx = t0 > t1 ? t0 : t1; \
    lo = lo > mn ? lo : mn; hi = hi < mx ? hi : mx; \
    TM = lo; J = __ballot(lo <= hi); }
    SLAB(M0, nx.x, fx.x, ny.x, fy.x, nz.x, fz.x, tm0)
    SLAB(M1, nx.y, fx.y, ny.y, fy.y, nz.y, fz.y, tm1)
    SLAB(M2, nx.z, fx.z, ny.z, fy.z, nz.z, fz.z, tm2)
    SLAB(M3, nx.w, fx.w, ny.w, fy.w, nz.w, fz.w, tm3)
#undef SLAB
  }
  const uint4 child = make_uint4(__float_as_uint(child4.x), __float_as_uint(child4.y), __float_as_uint(child4.z), __float_as_uint(child4.w));
  const bool any_child = __builtin_amdgcn_inverse_ballot_w64(M0 | M1 | M2 | M3);
  if(any_child)
  {
    MI_COUNT(cnt, 1, 1);
    MI_COUNT(cnt, 2, (uint32_t)__builtin_amdgcn_inverse_ballot_w64(M0) + (uint32_t)__builtin_amdgcn_inverse_ballot_w64(M1) + (uint32_t)__builtin_amdgcn_inverse_ballot_w64(M2) + (uint32_t)__builtin_amdgcn_inverse_ballot_w64(M3));
    /* front-to-back order from split axes and ray signs, qbvhmp.c:1313-1320: the near half is children
       {2*near0, 2*near0+1}, ordered by the sign along its own split axis; likewise the far half */
    const uint32_t axis0 = ax & 3u;
    const mi_u64 N0 = __ballot((nearbits >> axis0) & 1u);
    const bool near0 = __builtin_amdgcn_inverse_ballot_w64(N0);
    const uint32_t axis1n = near0 ? ((ax >> 4) & 3u) : ((ax >> 2) & 3u);
    const uint32_t axis1f = near0 ? ((ax >> 2) & 3u) : ((ax >> 4) & 3u);
    const mi_u64 N1N = __ballot((nearbits >> axis1n) & 1u), N1F = __ballot((nearbits >> axis1f) & 1u);
    const bool near1n = __builtin_amdgcn_inverse_ballot_w64(N1N), near1f = __builtin_amdgcn_inverse_ballot_w64(N1F);
    const uint32_t ca0 = near0 ? child.z : child.x, ca1 = near0 ? child.w : child.y;
    const uint32_t cb0 = near0 ? child.x : child.z, cb1 = near0 ? child.y : child.w;
    const float ta0 = near0 ? tm2 : tm0, ta1 = near0 ? tm3 : tm1;
    const float tb0 = near0 ? tm0 : tm2, tb1 = near0 ? tm1 : tm3;
    const uint32_t c00 = near1n ? ca1 : ca0, c01 = near1n ? ca0 : ca1;
    const uint32_t c10 = near1f ? cb1 : cb0, c11 = near1f ? cb0 : cb1;
    const float t01 = near1n ? ta0 : ta1;
    const float t10 = near1f ? tb1 : tb0, t11 = near1f ? tb0 : tb1;
    /* the hit flags go through the same two selections as lane masks in scalar registers (s_and / s_andn2 / s_or on the
       ballots: the scalar unit, not the vector pipes this kernel is bound by) and come back as predicates */
    /* conditional swaps: (X, Y) = N ? (B, A) : (A, B), lane by lane, four scalar instructions each */
    const mi_u64 D0 = (M0 ^ M2) & N0, D1 = (M1 ^ M3) & N0;
    const mi_u64 HA0 = M0 ^ D0, HB0 = M2 ^ D0, HA1 = M1 ^ D1, HB1 = M3 ^ D1;          /* hit flags of the near half (A), of the far half (B) */
    const mi_u64 DN = (HA0 ^ HA1) & N1N, DF = (HB0 ^ HB1) & N1F;
    const mi_u64 H00 = HA0 ^ DN, H01 = HA1 ^ DN, H10 = HB0 ^ DF, H11 = HB1 ^ DF;
    /* the first hit child in order n00,n01,n10,n11 becomes current; later ones are pushed far-first (qbvhmp.c:1336-1354) */
    const bool h00 = __builtin_amdgcn_inverse_ballot_w64(H00), h01 = __builtin_amdgcn_inverse_ballot_w64(H01), h10 = __builtin_amdgcn_inverse_ballot_w64(H10);
    const bool p11 = __builtin_amdgcn_inverse_ballot_w64(H11 & (H00 | H01 | H10));
    const bool p10 = __builtin_amdgcn_inverse_ballot_w64(H10 & (H00 | H01));
    const bool p01 = __builtin_amdgcn_inverse_ballot_w64(H01 & H00);
    if(sp + 3 <= STACK)
    { /* all three slots are in LDS */
#if MI_PUSH_BLIND
      /* written whether or not the child is pushed -- an entry above the top of the stack is never read --, the stack pointer moves by the
         flag: three stores without a divergent region each (each region is two scalar instructions, a branch and an EXEC restore) */
      lstack[sp*BLOCK] = mi_u32x2{c11, __float_as_uint(t11)}; sp += p11 ? 1 : 0;
      lstack[sp*BLOCK] = mi_u32x2{c10, __float_as_uint(t10)}; sp += p10 ? 1 : 0;
      lstack[sp*BLOCK] = mi_u32x2{c01, __float_as_uint(t01)}; sp += p01 ? 1 : 0;
#else
      if(p11) { lstack[sp*BLOCK] = mi_u32x2{c11, __float_as_uint(t11)}; sp++; }
      if(p10) { lstack[sp*BLOCK] = mi_u32x2{c10, __float_as_uint(t10)}; sp++; }
      if(p01) { lstack[sp*BLOCK] = mi_u32x2{c01, __float_as_uint(t01)}; sp++; }
#endif
    }
    else
    {
      if(p11) { stack_push<BLOCK, STACK>(lds, sp, make_uint2(c11, __float_as_uint(t11))); sp++; }
      if(p10) { stack_push<BLOCK, STACK>(lds, sp, make_uint2(c10, __float_as_uint(t10))); sp++; }
      if(p01) { stack_push<BLOCK, STACK>(lds, sp, make_uint2(c01, __float_as_uint(t01))); sp++; }
    }
    current = h00 ? c00 : h01 ? c01 : h10 ? c10 : c11;
    if(entry) { const float t00 = near1n ? ta1 : ta0; *entry = h00 ? t00 : h01 ? t01 : h10 ? t10 : t11; }
    MI_COUNT_MAX(cnt, 7, (uint32_t)sp);
  }
  else miss();
  return any_child;
}

/* one "while-while" round of accel_intersect (src/accel.d/qbvhmp.c:1262-1390, static boxes): descend inner nodes until this
 * lane holds a leaf (or runs out of work), then intersect that leaf and pop the next subtree */
template<int BLOCK, int STACK, bool MB = false, bool ANYHIT = false, bool JOBS = false, class CNT>
__device__ __forceinline__ void trace_round(const Lds &lds, const DPrim *prims, const V3 o, const V3 d, uint32_t ignore,
                                            Hit &hit, TraceState &ts, CNT &cnt)
{
  const RayBox rb = raybox_setup<false>(o, d, ts, lds.num_nodes);
  int sp = ts.sp;
  uint32_t current = ts.done ? MI_LEAF32 : ts.current;
  bool done = ts.done;
  lds_uint2 *lstack = (lds_uint2 *)lds.stack;
  {
    while(true)
    { /* descend until every lane holds a leaf -- or only a tail of MI_TAIL_INNER lanes is still descending while others
         wait with a leaf: those go on descending in the next round (their `current` stays an inner node) */
      const bool inner = !(current & MI_LEAF32);
      const unsigned ninner = __popcll(__ballot(inner));
      if(!ninner) break;
      if(ninner < MI_TAIL_INNER && __any((current & MI_LEAF32) && !done)) break;
      if(JOBS && MI_LEAF_EXIT < 64 && __popcll(__ballot((current & MI_LEAF32) && !done)) >= MI_LEAF_EXIT) break;   /* enough leaves wait to fill the lanes of a job pass */
#ifdef MI_PROFILE_LOOPS
      const unsigned nround = __popcll(__ballot(!done));    /* lanes of this round whose ray is still under way */
#endif
      if(!inner) continue;
#ifdef MI_PROFILE_LOOPS
      if(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) cnt.c[8] += MI_PROFILE_LOOPS == 2 ? nround : 1;   /* wave-level inner iterations (2: lanes still under way) */
#endif
      /* MI_POP_PREFETCH: the stack's top entry is read together with the node's planes -- if no child is hit the pop has it already
         instead of starting a round trip of its own (a visit that hits a child pushes above it and leaves it alone) */
      const bool has_top = MI_POP_PREFETCH && sp > 0 && sp <= STACK;
      mi_u32x2 top = mi_u32x2{0u, 0u};
      if(has_top) top = lstack[(sp - 1)*BLOCK];
      node_visit<BLOCK, STACK, false, MB>(lds, lstack, rb, o, hit.dist, current, sp, cnt, [&]()
      { /* pop, skipping entries that start behind the current hit (qbvhmp.c:1357-1364) */
        current = MI_LEAF32;          /* empty leaf: falls out of this loop; `done` if the stack runs dry */
        done = true;
        bool first = has_top;
        while(sp > 0)
        {
          sp--;
          const uint2 e = first ? make_uint2(top.x, top.y) : stack_top<BLOCK, STACK>(lds, lstack, sp);
          first = false;
          if(!(__uint_as_float(e.y) > hit.dist)) { current = e.x; done = false; break; }
        }
      });
    }
  }
  ts.sp = sp; ts.current = current; ts.done = done;
  MI_TT(cnt, 0)
  if(!JOBS && !done && (current & MI_LEAF32)) leaf_sequential<BLOCK, STACK, MB, ANYHIT>(lds, prims, o, d, ignore, hit, ts, cnt);   /* JOBS: the caller runs leaf_jobs with all lanes */
}

/* ------------------------------------------------------------------------------------------ FAST traversal rounds
 * The rounds above keep the reference's order of operations ray by ray (that is what makes the node / box / primitive counters
 * equal its -DACCEL_DEBUG totals): a lane that reaches a leaf waits until the wave's leaf phase has tested it, because the next
 * node it may enter depends on the distance that leaf leaves. 53 % of the node loop's lane-slots were such waits.
 * trace_round_spec lets the lane go on instead: the leaf is PUT ASIDE (up to MI_SPEC_K of them per round, plus the one it holds
 * when the node loop ends), the next subtree is popped against the unchanged distance, and the leaf phase tests the put-aside
 * leaves of all lanes together -- more jobs per pass, fewer rounds, fewer idle slots. What it yields:
 *   - the same closest hit, bit for bit: the nodes visited are a SUPERSET of the reference's (a subtree is entered iff its box
 *     is hit in front of the distance known at that time, which is never shorter than the reference's at the same point of
 *     its traversal), the order in which a ray reaches its leaves is the reference's (same stack discipline), every primitive
 *     is tested against the distance the round started with and the results are merged by (distance, position in that order):
 *     closest wins, later wins a tie -- what the running `t <= dist` of the sequential loop does. A subtree pushed during the
 *     round that the shorter distance would have culled at its box test is culled when it is popped (same comparison).
 *   - different work counters: speculative node visits and primitive tests are real work and are counted as such.
 * Leaves in which the order of the tests matters (a folded quad crossed in both halves) poison the lane's result slot as in
 * leaf_jobs and the lane works through its leaves sequentially, in order. */
#ifndef MI_SPEC_K
#define MI_SPEC_K 1           /* leaves a lane may put aside per round; one more can wait in `current`. A/B on cfg 2, 1 / 2 / 3: 18.68 / 18.92 / 20.25 ms */
#endif
#ifndef MI_SPEC_BLOCKED
#define MI_SPEC_BLOCKED 64    /* the node loop of a round ends once this many lanes hold a leaf they can no longer put aside; 64 = never. Re-swept
                                 after the exact node loop was repaired (cfg 2, same box): 16 / 24 / 32 / 40 / 48 / 64 = 18.87 / 18.50 / 18.31 / 18.18 /
                                 18.20 / 18.19 ms -- a blocked lane loses less by waiting than the wave loses by turning to its leaves early */
#endif
#ifndef MI_SPEC_TAIL_INNER
#define MI_SPEC_TAIL_INNER 6  /* ... and once fewer than this many lanes still move while others wait (the exact rounds' MI_TAIL_INNER is 4):
                                 2 / 4 / 6 / 8 / 12 = 19.24 / 18.19 / 18.13 / 18.13-18.16 / 18.31 ms */
#endif
#ifndef MI_SPEC_ANYHIT_WAITS
#define MI_SPEC_ANYHIT_WAITS 0   /* A/B: a shadow ray that may stop at its first occluder does not run ahead of its leaves */
#endif
#ifndef MI_SPEC_EXACT
#define MI_SPEC_EXACT 1       /* the entry-distance bookkeeping that keeps the FAST rounds' hits the reference's where rounding is not monotone (trace_round_spec):
                                 cfg 2 18.22 against 18.05 ms without; FAST against exact rounds of one library on 16 M + 8 M paths: 0 against 4 + 1 differing */
#endif
#define MI_SPEC_JOBS_MAX 512  /* job list entries per wave (one byte each: owner lane | slot << 6) */

template<int BLOCK, int STACK>
__device__ __forceinline__ void stack_pop(const Lds &lds, lds_uint2 *lstack, float dist, int &sp, uint32_t &current, bool &done, float &entry)
{ /* pop, skipping entries that start behind the current hit (qbvhmp.c:1357-1364); entry = the popped entry's distance */
  current = MI_LEAF32;
  done = true;
  while(sp > 0)
  {
    sp--;
    const uint2 e = stack_top<BLOCK, STACK>(lds, lstack, sp);
    if(!(__uint_as_float(e.y) > dist)) { current = e.x; done = false; entry = __uint_as_float(e.y); break; }
  }
}

template<int BLOCK, int STACK, bool MB, bool ANYHIT, bool FMA, class CNT>
__device__ __forceinline__ void trace_round_spec(const Lds &lds, const DPrim *prims, const V3 o, const V3 d, uint32_t ignore,
                                                 Hit &hit, TraceState &ts, bool busy, CNT &cnt)
{ /* call from ALL lanes of the wave; busy = this lane has a ray under way */
  constexpr int K = MI_SPEC_K;
  static_assert(K >= 1 && K <= 3, "slots are two bits of a job byte");
  const unsigned lane = __lane_id();
  const RayBox rb = raybox_setup<FMA>(o, d, ts, lds.num_nodes);
  /* what a popped subtree's entry distance is compared with: the closest hit so far, with the slack of the FMA box test */
  /* (a lane whose own ray is degenerate only ever takes part in literal rounds: its entries carry the literal test's distances and
     are compared as the reference compares them; any other lane's entries may come from FMA rounds and always get the slack) */
  const bool own_literal = !FMA || !(fabsf(rb.idx) < 1e30f) || !(fabsf(rb.idy) < 1e30f) || !(fabsf(rb.idz) < 1e30f);
#define MI_CULL_DIST (own_literal ? hit.dist : __builtin_fmaf(hit.dist, MI_FMA_REL, rb.slack))
  lds_uint2 *lstack = (lds_uint2 *)lds.stack;
  int sp = ts.sp;
  bool done = ts.done || !busy;
  uint32_t current = done ? MI_LEAF32 : ts.current;
  uint32_t lf[K + 1];                                   /* the leaves of this round in the order the ray reached them; 0 = none */
#pragma unroll
  for(int k=0;k<=K;k++) lf[k] = 0u;
  /* Keeping the result the reference's when rounding is not monotone. A leaf reached AFTER a leaf was put aside has been reached
     through box tests against the distance of the round's start; the reference makes those tests against the distance the put-aside
     leaf leaves, and does not get there if one of them fails. In exact arithmetic a hit in such a leaf lies behind its boxes' entry
     and loses anyway; in floats it can be closer by an ulp or -- on the shared edge of two quads in different leaves -- equally far
     and win the tie as the later one (0.25 such paths per million on regression/0010_pt). So the lane keeps `lo`, the largest entry
     distance among the tests since its last pop, and (a) an inner node it still holds when the round ends goes back on the stack
     with lo as its entry distance -- the next round pops it against the distance this round leaves, as the reference would have
     tested it --, (b) the held leaf's results only count if lo is not behind what the put-aside leaf leaves (epilogue). */
  float lo = 0.0f;          /* only read after a leaf was put aside, and that is followed by a pop, which sets it */
  /* -------- node loop */
  while(true)
  {
    const bool inner = !(current & MI_LEAF32);          /* finished lanes carry MI_LEAF32 */
    const bool leaf = (current & MI_LEAF32) && !done;
    /* room to put it aside (an empty leaf needs none). Not in a wave with a degenerate ray (1/dir infinite): there the box tests
       can yield NaN, the reference's verdict then depends on the distance it happened to know (SSE operand order), and "tested
       against a longer distance = superset" no longer holds -- such waves wait with their leaves like the exact rounds do */
    const bool advance = leaf && (!(current & 31u) || (!rb.slow && lf[K-1] == 0u && !(ANYHIT && MI_SPEC_ANYHIT_WAITS && ts.anyhit)));
    const mi_u64 mmove = __ballot(inner || advance);
    if(!mmove) break;
    const mi_u64 mwait = __ballot(leaf && !advance);
    if(__popcll(mmove) < MI_SPEC_TAIL_INNER && (mwait || __any(lf[0] != 0u))) break;
    if(__popcll(mwait) >= MI_SPEC_BLOCKED) break;
#ifdef MI_PROFILE_LOOPS
    { const unsigned nround = __popcll(__ballot(!done)); if(lane == 0) cnt.c[8] += MI_PROFILE_LOOPS == 2 ? nround : 1; }
#endif
    bool pop = false;
    if(inner)
    {
      float entry = 0.0f;
      pop = !node_visit<BLOCK, STACK, FMA>(lds, lstack, rb, o, hit.dist, current, sp, cnt, [](){}, MI_SPEC_EXACT ? &entry : nullptr);
      if(MI_SPEC_EXACT && !pop) lo = fmaxf(lo, entry);
    }
    else if(advance)
    {
      if(current & 31u)
      {
#pragma unroll
        for(int k=0;k<K;k++) if(lf[k] == 0u) { lf[k] = current; break; }
      }
      pop = true;
    }
    if(pop) stack_pop<BLOCK, STACK>(lds, lstack, MI_CULL_DIST, sp, current, done, lo);
  }
  MI_TT(cnt, 0)
  if(MI_SPEC_EXACT && !done && !(current & MI_LEAF32) && lf[0] != 0u)
  { /* (a) */
    stack_push<BLOCK, STACK>(lds, sp, make_uint2(current, __float_as_uint(lo))); sp++;
    current = MI_LEAF32;                                /* an empty leaf: the next round's node loop pops */
  }
  /* -------- leaf phase: the (lane, slot, primitive) tests of all put-aside leaves dealt out over the 64 lanes */
  const bool holds = (current & MI_LEAF32) && !done;    /* the leaf the lane still holds is tested too, then popped (so is the entry of (a), against the round's result) */
  if(holds) lf[K] = current;
  uint32_t n[K + 1], num = 0;
#pragma unroll
  for(int k=0;k<=K;k++) { n[k] = lf[k] & 31u; num += n[k]; }
  if(__any(num != 0u || holds))
  {
    bool big = false;
#pragma unroll
    for(int k=0;k<=K;k++) big = big || n[k] > 7u;
    /* exclusive prefix sum of num (<= 7 (K + 1) <= 28) over the wave from the ballots of its bits */
    const mi_u64 b0 = __ballot(num & 1u), b1 = __ballot(num & 2u), b2 = __ballot(num & 4u), b3 = __ballot(num & 8u), b4 = __ballot(num & 16u);
    const uint32_t J = __popcll(b0) + 2u*__popcll(b1) + 4u*__popcll(b2) + 8u*__popcll(b3) + 16u*__popcll(b4);
    const bool fits = !__any(big) && J <= MI_SPEC_JOBS_MAX;
#define MI_MBCNT(M) __builtin_amdgcn_mbcnt_hi((uint32_t)((M) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(M), 0u))
    const uint32_t prefix = MI_MBCNT(b0) + 2u*MI_MBCNT(b1) + 4u*MI_MBCNT(b2) + 8u*MI_MBCNT(b3) + 16u*MI_MBCNT(b4);
#undef MI_MBCNT
    lds_u8 *jobs = (lds_u8 *)lds.jobs;
    lds_uint2 *col = (lds_uint2 *)lds.stack;
    lds_u64 *best = (lds_u64 *)(col + (STACK + 0)*BLOCK);
    lds_uint2 *uvs = col + (STACK + 1)*BLOCK;
    lds_u32 *anl = (lds_u32 *)(col + (STACK + 2)*BLOCK);                /* + 1: closest triangle / quad distance of the put-aside leaves (slots < K) */
    if(num && fits)
    {
      uint32_t at = prefix;
#pragma unroll
      for(int k=0;k<=K;k++) { for(uint32_t i=0;i<n[k];i++) jobs[at + i] = (unsigned char)(lane | ((unsigned)k << 6)); at += n[k]; }
      *best = ((mi_u64)__float_as_uint(hit.dist) << 32) | 0xffffffffull;
      anl[0] = 0u; anl[1] = __float_as_uint(hit.dist);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    MI_TT(cnt, 1)
    for(uint32_t base=0;fits && base<J;base+=64u)
    {
#ifdef MI_PROFILE_LOOPS
      if(lane == 0) cnt.c[9] += MI_PROFILE_LOOPS == 2 ? (J - base < 64u ? J - base : 64u) : 1;
#endif
      const uint32_t j = base + lane;
      const bool valid = j < J;
      const uint32_t e = valid ? (uint32_t)jobs[j] : lane;
      const int src = (int)(e & 63u);
      const uint32_t slot = e >> 6;
      uint32_t sl[K + 1];
#pragma unroll
      for(int k=0;k<=K;k++) sl[k] = (uint32_t)__shfl((int)lf[k], src);
      const uint32_t spre = (uint32_t)__shfl((int)prefix, src);
      uint32_t link = sl[0], off = 0u, acc = 0u;
#pragma unroll
      for(int k=1;k<=K;k++) { acc += sl[k-1] & 31u; if(slot == (uint32_t)k) { link = sl[k]; off = acc; } }
      const uint32_t kk = j - spre - off;
      const uint32_t prim = valid ? ((link ^ MI_LEAF32) >> 5) + kk : 0u;
      const uint32_t pos = (slot << 3) | kk;             /* order of the tests on the owner's ray: slot, then position in the leaf */
      const PrimRegs rec = prim_load(prims, prim);
      const V3 so = mk3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src)), sd = mk3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
      Hit h;
      h.prim = MI_NOPRIM; h.u = h.v = 0.0f;
      h.dist = __shfl(hit.dist, src);
      const uint32_t sign = (uint32_t)__shfl((int)ignore, src);
      const int offl = src - (int)lane;
      const uint32_t type = __float_as_uint(rec.q3.x);
      const bool both = triquad_intersect<true>(rec, type, so, sd, h, prim);
      const bool tq = valid && type >= MI_PRIM_TRI && prim != sign;       /* triangle.h:271 */
      const bool cand = tq && !both && h.prim != MI_NOPRIM;
      const mi_u64 key = ((mi_u64)__float_as_uint(h.dist) << 32) | (mi_u64)(31u - pos);
      if(tq && (both || cand)) __hip_atomic_fetch_min(best + offl, both ? (mi_u64)0 : key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   /* 0: poison, the owner goes sequential */
      if(valid && type < MI_PRIM_TRI) __hip_atomic_fetch_or(anl + 2*offl, 1u << pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if(MI_SPEC_EXACT && cand && slot < (uint32_t)K) __hip_atomic_fetch_min(anl + 2*offl + 1, __float_as_uint(h.dist), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   /* distances are positive: ordered as integers */
      if(__any(cand))
      { /* the job that holds the owner's minimum so far leaves its u, v (a later, closer one overwrites them) */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if(cand && best[offl] == key) uvs[offl] = mi_u32x2{__float_as_uint(h.u), __float_as_uint(h.v)};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    MI_TT(cnt, 2)
    if(num)
    {
      mi_u64 res = fits ? *best : 0;
      /* (b) above: does the reference reach the held leaf after the put-aside one? Only in doubt when the held leaf's winner lies in
         front of the entry distance `lo` of its own boxes -- then the distance the put-aside leaf leaves decides (its triangles /
         quads: anl[1]; with sphere / line tests among its primitives the lane takes the in-order path below, which applies the
         reference's test as it goes) */
      static_assert(!MI_SPEC_EXACT || K == 1, "the entry-distance bookkeeping covers one put-aside leaf per round");
      const bool second = MI_SPEC_EXACT && holds && lf[0] != 0u;          /* the held leaf was reached past a put-aside one */
      if(second && res != 0 && (uint32_t)res != 0xffffffffu && ((31u - (uint32_t)res) >> 3) == (uint32_t)K && lo > __uint_as_float((uint32_t)(res >> 32)))
      {
        const float d0 = fminf(hit.dist, __uint_as_float(anl[1]));
        if((anl[0] & 0xffu) != 0u || lo > d0) res = 0;
      }
      if(res == 0)
      { /* order matters in one of the leaves (or the list does not fit): this lane's leaves one after the other, each in order */
#pragma unroll
        for(int k=0;k<=K;k++) if(n[k] && !(second && k == K && lo > hit.dist)) leaf_tests<MB>(prims, lf[k], o, d, ignore, hit, ts, cnt);
      }
      else
      {
        MI_COUNT(cnt, 3, num);
        if((uint32_t)res != 0xffffffffu)
        {
          const uint32_t pos = 31u - (uint32_t)res;
          uint32_t link = lf[0];
#pragma unroll
          for(int k=1;k<=K;k++) if((pos >> 3) == (uint32_t)k) link = lf[k];
          const mi_u32x2 uv = *uvs;
          hit.dist = __uint_as_float((uint32_t)(res >> 32)); hit.prim = ((link ^ MI_LEAF32) >> 5) + (pos & 7u);
          hit.u = __uint_as_float(uv.x); hit.v = __uint_as_float(uv.y);
        }
        uint32_t analytic = *anl;
        while(analytic)
        {
#ifdef MI_PROFILE_LOOPS
          { const unsigned nl = __popcll(__ballot(1)); if(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) cnt.c[10] += MI_PROFILE_LOOPS == 2 ? nl : 1; }
#endif
          if(second && lo > hit.dist) { analytic &= 0xffu; if(!analytic) break; }   /* the reference does not get to the held leaf: only the put-aside leaf's tests count */
          const uint32_t pos = __ffs(analytic) - 1;
          analytic &= analytic - 1;
          uint32_t link = lf[0];
#pragma unroll
          for(int k=1;k<=K;k++) if((pos >> 3) == (uint32_t)k) link = lf[k];
          analytic_intersect<MB>(prims, ((link ^ MI_LEAF32) >> 5) + (pos & 7u), o, d, ignore, hit, MB ? ts.time : 0.0f, MB ? ts.prims_t1 : nullptr);
        }
      }
    }
    if(ANYHIT && ts.anyhit && hit.prim != MI_NOPRIM && !done) { sp = 0; current = MI_LEAF32; done = true; }   /* an occluder is all such a shadow ray needs */
    else if(holds) stack_pop<BLOCK, STACK>(lds, lstack, MI_CULL_DIST, sp, current, done, lo);
#undef MI_CULL_DIST
    MI_TT(cnt, 3)
  }
  if(busy) { ts.sp = sp; ts.current = current; ts.done = done; }
}

template<int BLOCK, int STACK, bool FAST = false, class CNT>
__device__ __forceinline__ void accel_intersect(const Lds &lds, const DPrim *prims, const V3 o, const V3 d, uint32_t ignore,
                                                Hit &hit, CNT &cnt, bool live = true)
{ /* closest hit for one ray per lane; the wave iterates until every lane is done. Call from ALL lanes of the wave (live = false:
     this lane has no ray): the distributed leaf phase deals work out to every lane */
  TraceState ts;
  trace_begin(lds, ts, d, cnt);
  if(!live) { ts.done = true; MI_COUNT(cnt, 0, (uint32_t)-1); }
  if(FAST)
  {
    while(__any(!ts.done)) trace_round_spec<BLOCK, STACK, false, false, MI_SPEC_FMA != 0>(lds, prims, o, d, ignore, hit, ts, !ts.done, cnt);
    return;
  }
#if MI_LEAF_JOBS
  while(__any(!ts.done))
  {
    const bool busy = !ts.done;
    if(busy) trace_round<BLOCK, STACK, false, false, true>(lds, prims, o, d, ignore, hit, ts, cnt);
    leaf_jobs<BLOCK, STACK, false, false>(lds, prims, o, d, ignore, hit, ts, busy, cnt);
  }
#else
  while(!ts.done) trace_round<BLOCK, STACK>(lds, prims, o, d, ignore, hit, ts, cnt);
#endif
}

/* ------------------------------------------------------------------------------------------ geometry at the hit */
MI_HD V3 decode_normal(uint32_t enc)
{ /* geo_decode_normal, include/geo.h:24-44 */
  const uint32_t p0 = enc & 0xffffu, p1 = enc >> 16;
  const uint32_t v0 = 0x3f800000u | ((p0 & 0x7fffu) << 8);
  const uint32_t v1 = 0x3f800000u | ((p1 & 0x7fffu) << 8);
  float x = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, 2.0f*__builtin_bit_cast(float, v0) - 2.0f) | ((p0 & 0x8000u) << 16));
  float y = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, 2.0f*__builtin_bit_cast(float, v1) - 2.0f) | ((p1 & 0x8000u) << 16));
  const float z = 1.0f - (fabsf(x) + fabsf(y));
  if(z < 0.0f)
  {
    const float oldx = x;
    x = (1.0f - fabsf(y)) * ((oldx < 0.0f) ? -1.0f : 1.0f);
    y = (1.0f - fabsf(oldx)) * ((y < 0.0f) ? -1.0f : 1.0f);
  }
  return normalise3(mk3(x, y, z));
}

MI_HD float half2float(uint32_t h)
{ /* half_to_float, include/half.h:57-80 */
  const uint32_t sign = (h & 0x8000u) << 16;
  uint32_t o = (h & 0x7fffu) << 13;
  const uint32_t ex = 0x0f800000u & o;
  o += (127 - 15) << 23;
  if(ex == 0x0f800000u) o += (128 - 16) << 23;
  else if(ex == 0)
  {
    o += 1 << 23;
    o = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, o) - __builtin_bit_cast(float, 113u << 23));
  }
  return __builtin_bit_cast(float, o | sign);
}

struct Surf
{
  V3 x, n, gn, a, b;
  float u, v, s, t;
  uint32_t flags;
};

MI_HD V3 tri_geo_normal(const V3 v0, const V3 v1, const V3 v2)
{ /* geo_tri_get_normal, include/geo/triangle.h:63-70 */
  return normalise3(mk3((v1.y-v0.y)*(v2.z-v0.z) - (v1.z-v0.z)*(v2.y-v0.y),
                        (v1.z-v0.z)*(v2.x-v0.x) - (v1.x-v0.x)*(v2.z-v0.z),
                        (v1.x-v0.x)*(v2.y-v0.y) - (v1.y-v0.y)*(v2.x-v0.x)));
}

template<bool MB = false>
__device__ __forceinline__ void surface_setup(const DScene &sc, uint32_t prim, const uint4 head, const V3 omega, float scramble, Surf &sf, float time = 0.0f)
{ /* prims_get_normal_time (src/prims.c:254-366) + manifold_init (include/pathspace/manifold.h:215-232) */
  const DPrimGeo &geo = sc.primgeo[prim];
  const float *g = geo.f;                   /* per-primitive constants, see DPrimGeo */
  const uint32_t type = MB ? head.x & 7u : head.x;   /* head = the record's first 16 bytes: type (| MI_GEO_MB), material, uv0, primid_lo */
  float gm[35];
  if(MB && type < MI_PRIM_TRI && (head.x & MI_GEO_MB))
  { /* moving sphere / line: its constants at the path's time (those of a static one are precomputed in DPrimGeo) */
    V3 a0, a1;
    const DPrim at = moving_analytic_at(sc.prims[prim], sc.prims_t1[prim], time, a0, a1);
    for(int k=0;k<35;k++) gm[k] = geo.f[k];                 /* texture coordinates stay (f[18..25]) */
    if(type == MI_PRIM_SPHERE) { gm[29] = a0.x; gm[30] = a0.y; gm[31] = a0.z; gm[32] = at.v[1][0]; }
    else line_shading_consts(gm, at, a0, a1);
    g = gm;
  }
  if(type < MI_PRIM_TRI)
  { /* sphere (sphere.h:51-62,160-161) and line (line.h:123-161). The two share one atan2f site: a wave that holds hits
       of both kinds runs the long libm sequence once. */
    const V3 v0 = mk3(g[29], g[30], g[31]);        /* sphere: centre */
    const float r0 = g[32], r1 = g[33];            /* sphere: g[32] = radius */
    const bool sphere = type == MI_PRIM_SPHERE;
    const bool cylinder = !sphere && fabsf(r1-r0) < 1e-3;
    const V3 x = sub3(sf.x, v0);
    float ay, ax;
    if(sphere) { ay = x.y/r0; ax = x.x/r0; }
    else if(cylinder) { ay = sf.u; ax = sf.v; }    /* hit.u/hit.v carry out[1], out[2] of the intersection (line.h:484-486) */
    else { ay = dot3(mk3(g[10], g[11], g[12]), x); ax = dot3(mk3(g[13], g[14], g[15]), x); }   /* cone: line.h:445-449 */
    const float ang = (float)((double)atan2f(ay, ax)/(2.0f*MI_PI_D));
    if(sphere)
    {
      sf.u = ang;
      sf.v = (float)((double)acosf(DCLAMP(x.z/r0, -1.0f, 1.0f))/MI_PI_D);
      sf.gn = normalise3(x);
      sf.n = sf.gn;
    }
    else
    {
      const V3 d = mk3(g[0], g[1], g[2]), a = mk3(g[4], g[5], g[6]), b = mk3(g[7], g[8], g[9]);
      const float ilen_d = g[3];
      sf.v = ang;
      if(cylinder) sf.u = dot3(x, d)*ilen_d;
      if(fabsf(r0-r1) < 1e-3f && r0 < 0.01f) { sf.n = sf.gn = mk3(0, 0, 0); }
      else
      {
        const float phi = (float)(2.0*MI_PI_D*(double)sf.v);
        float sinphi, cosphi;
        mi_sincosf(phi, &sinphi, &cosphi);
        const V3 n = mk3(a.x*sinphi + b.x*cosphi, a.y*sinphi + b.y*cosphi, a.z*sinphi + b.z*cosphi);
        const float rr = r1 - r0;
        if(fabsf(rr) < 1e-3) sf.n = n;
        else sf.n = normalise3(mk3(n.x - d.x*(r1-r0)*ilen_d, n.y - d.y*(r1-r0)*ilen_d, n.z - d.z*(r1-r0)*ilen_d));
        sf.gn = sf.n;
      }
    }
  }
  else if(MB && (head.x & MI_GEO_MB))
  { /* motion blurred: vertices and decoded normals of the hit half at the path's time (geo_get_vertex_time / geo_get_normal_time,
       include/geo.h:120-162), then geo_tri_get_normal (include/geo/triangle.h:63-82) on them */
    const bool second = type == MI_PRIM_QUAD && !(sf.v >= sf.u);
    const float u = second ? sf.u - sf.v : sf.u;
    const float v = type == MI_PRIM_TRI ? sf.v : second ? sf.v : sf.v - sf.u;
    const int i1 = second ? 2 : 1, i2 = second ? 3 : 2;
    const DPrim &p0 = sc.prims[prim];
    const DPrimT1 &p1 = sc.prims_t1[prim];
    const float w0 = 1.0f - time, w1 = time;
    const V3 v0 = mk3(w0*p0.v[0][0] + w1*p1.v[0][0], w0*p0.v[0][1] + w1*p1.v[0][1], w0*p0.v[0][2] + w1*p1.v[0][2]);
    const V3 va = mk3(w0*p0.v[i1][0] + w1*p1.v[i1][0], w0*p0.v[i1][1] + w1*p1.v[i1][1], w0*p0.v[i1][2] + w1*p1.v[i1][2]);
    const V3 vb = mk3(w0*p0.v[i2][0] + w1*p1.v[i2][0], w0*p0.v[i2][1] + w1*p1.v[i2][1], w0*p0.v[i2][2] + w1*p1.v[i2][2]);
    const V3 n0 = mk3(w0*g[0] + w1*p1.n[0][0], w0*g[1] + w1*p1.n[0][1], w0*g[2] + w1*p1.n[0][2]);
    const V3 na = mk3(w0*g[3*i1] + w1*p1.n[i1][0], w0*g[3*i1+1] + w1*p1.n[i1][1], w0*g[3*i1+2] + w1*p1.n[i1][2]);
    const V3 nb = mk3(w0*g[3*i2] + w1*p1.n[i2][0], w0*g[3*i2+1] + w1*p1.n[i2][1], w0*g[3*i2+2] + w1*p1.n[i2][2]);
    sf.gn = tri_geo_normal(v0, va, vb);
    const float w = 1.0f - u - v;
    sf.n = normalise3(mk3(u*nb.x + v*na.x + w*n0.x, u*nb.y + v*na.y + w*n0.y, u*nb.z + v*na.z + w*n0.z));
  }
  else
  { /* triangle / quad halves (v0 v1 v2) and (v0 v2 v3): decoded vertex normals and both geometric normals come from DPrimGeo */
    const bool second = type == MI_PRIM_QUAD && !(sf.v >= sf.u);
    const float u = second ? sf.u - sf.v : sf.u;
    const float v = type == MI_PRIM_TRI ? sf.v : second ? sf.v : sf.v - sf.u;
    const float *n1 = g + (second ? 6 : 3), *n2 = g + (second ? 9 : 6);
    sf.gn = ld3(g + (second ? 15 : 12));
    const float w = 1.0f - u - v;
    sf.n = normalise3(mk3(u*n2[0] + v*n1[0] + w*g[0], u*n2[1] + v*n1[1] + w*g[1], u*n2[2] + v*n1[2] + w*g[2]));
  }
  /* texture coordinates, src/prims.c:300-365; the half / fixed-point uv of the record are decoded at upload (g[18..25]) */
  if(head.z == 0) { sf.s = sf.u; sf.t = sf.v; }
  else if(type == MI_PRIM_SPHERE) { sf.s = sf.u + g[18]; sf.t = sf.v + g[19]; }
  else if(type == MI_PRIM_LINE) { sf.s = g[18]; sf.t = g[19]; }
  else
  {
    const bool second = type == MI_PRIM_QUAD && !(sf.v >= sf.u);
    const float u = second ? sf.u - sf.v : sf.u;
    const float v = type == MI_PRIM_TRI ? sf.v : second ? sf.v : sf.v - sf.u;
    const float *t1 = g + (second ? 22 : 20), *t2 = g + (second ? 24 : 22);
    sf.s = (1.0f-u-v)*g[18] + v*t1[0] + u*t2[0];
    sf.t = (1.0f-u-v)*g[19] + v*t1[1] + u*t2[1];
  }
  /* flip towards the ray, tangent frame */
  sf.flags = s_none;
  if(dot3(omega, sf.gn) > 0.0f) { sf.n = neg3(sf.n); sf.flags |= s_inside; }
  /* the tangent frame get_scrambled_onb(scramble, n) (manifold.h:226-232) is formed by the caller where it is first needed -- the
     bsdf sample at the end of path_shade -- instead of living through next event estimation (six registers of the ptdl kernel) */
}

/* ------------------------------------------------------------------------------------------ spectra, shading inputs */
struct Shading { float roughness, rs, rd, rg, em; };

__device__ __forceinline__ float spectrum_eval(const float *coeff, float lambda)
{ /* rgb2spec_eval_fast, include/rgb2spec.h:145-149 (exact rsqrt instead of rsqrtss) */
  const float x = (coeff[0]*lambda + coeff[1])*lambda + coeff[2];
  const float y = mi_rcp(mi_sqrt(x*x + 1.0f));
  return .5f*x*y + .5f;
}

__device__ __forceinline__ float eta_from_abbe(float n_d, float V_d, float lambda)
{ /* include/spectrum.h:40-63 */
  float A, B;
  if(V_d == 0.0f) { A = n_d; B = 0.0f; }
  else
  {
    const float l_C = .6563f, l_F = .4861f, l_D = .587561f;
    const float c = (l_C*l_C * l_F*l_F)/(l_C*l_C - l_F*l_F);
    B = (n_d - 1.0f)/V_d * c;
    A = n_d - B/(l_D*l_D);
  }
  return A + (B*1e6f)/(lambda*lambda);
}

__device__ __forceinline__ void set_slot(Shading &sh, uint32_t slot, float val)
{ /* tex_set_slot, src/shaders/texture.h:34-66 */
  if(slot == MI_SLOT_DIFFUSE) sh.rd = val;
  else if(slot == MI_SLOT_SPECULAR) sh.rs = val;
  else if(slot == MI_SLOT_GLOSSY) sh.rg = val;
  else if(slot == MI_SLOT_ROUGHNESS) sh.roughness = val;
  else if(slot == MI_SLOT_EMISSION) sh.em = val;
}

__device__ __forceinline__ void run_prepare_ops(const DScene &sc, const DMaterial &m, uint32_t num_ops, const Surf &sf, float lambda, Shading &sh)
{ /* mult.c:154-167 -> color.c:75-82 / colorcheckersg.c:244-262 */
  sh.roughness = 1.0f; sh.rs = sh.rd = sh.rg = sh.em = 0.0f;
  for(uint32_t k=0;k<num_ops;k++)
  {
    /* one op = two 16-B loads issued together: kind, slot, coeff[0..1] | coeff[2], mul, roughness, pad */
    const uint4 oa = *(const uint4 *)&m.op[k];
    const float4 ob = *(const float4 *)((const char *)&m.op[k] + 16);
    mi_shade_op op;
    op.kind = oa.x; op.slot = oa.y; op.coeff[0] = __uint_as_float(oa.z); op.coeff[1] = __uint_as_float(oa.w);
    op.coeff[2] = ob.x; op.mul = ob.y; op.roughness = ob.z;
    if(op.kind == MI_OP_COLOR)
    {
      sh.roughness = op.roughness;
      const float val = op.mul*spectrum_eval(op.coeff, lambda);
      if(op.slot == MI_SLOT_EMISSION) set_slot(sh, op.slot, val);
      else if(op.slot != MI_SLOT_UNUSED) set_slot(sh, op.slot, DCLAMP(val, 0.0f, 1.0f));
    }
    else
    {
      const float u = sf.s, t = sf.t;
      const int i = (int)(14.0f*u) % 14, j = (int)(10.0f*t) % 10;
      float val;
      /* fmodf(x, 1.0f) == x - truncf(x) exactly (the fraction of a float is a float); the libm loop is not needed */
      const float xu = 14.0f*u, xt = 10.0f*t;
      const float fu = xu - truncf(xu), ft = xt - truncf(xt);
      if(fu < 0.1f || fu > 0.9f || ft < 0.1f || ft > 0.9f) val = 0.3f;
      else
      {
        const int l = (int)((lambda - 380.0f)/10.0f);
        if(l < 0 || l >= 36) val = 0.0f;
        else val = sc.checker[36*(14*j + i) + l];
      }
      set_slot(sh, op.slot, val);
    }
  }
}

/* ------------------------------------------------------------------------------------------ nested dielectrics */
/* _path_edge_medium (src/pathspace.c:80-115) keeps, per query, the multiset of shapes the path is inside of and
 * returns the one with the smallest shape id. The reference recomputes it from the vertex history every time;
 * we carry the multiset along the path (<= MI_MEDIA entries of 8-bit shape ids, packed) -- same result. */
#define MI_MEDIA 8
struct Media
{
  unsigned long long ids;   /* MI_MEDIA x 8 bit shape ids */
  uint32_t count;
  uint32_t broken;          /* nesting broke (kill the path when the medium is needed) */
};

__device__ __forceinline__ void media_apply(Media &m, uint32_t shape, bool inside)
{ /* one transmission event at a vertex on `shape` */
  if(!inside)
  {
    if(m.count < MI_MEDIA) { m.ids |= (unsigned long long)shape << (8*m.count); m.count++; }
    else m.broken = 1;      /* deeper nesting than we carry: treated as broken (counted, never seen in scope) */
  }
  else
  {
    /* search from the top for the shape, replace it by the last entry */
    int found = -1;
    for(int k=(int)m.count-1;k>=0;k--) if(((m.ids >> (8*k)) & 0xffu) == shape) { found = k; break; }
    if(found < 0) { m.broken = 1; return; }
    m.count--;
    const unsigned long long last = (m.ids >> (8*m.count)) & 0xffu;
    m.ids &= ~(0xffull << (8*found));
    m.ids |= last << (8*found);
    m.ids &= ~(0xffull << (8*m.count));
  }
}

__device__ __forceinline__ int media_top_shape(const Media &m)
{ /* smallest shape id in the set, -1 = exterior (vacuum) */
  int best = -1;
  for(uint32_t k=0;k<m.count;k++)
  {
    const int s = (int)((m.ids >> (8*k)) & 0xffu);
    if(best < 0 || s < best) best = s;
  }
  return best;
}

/* ior of the interior of `shape` as its prepare() would set it (dielectric.c:72, vacuum otherwise) */
__device__ __forceinline__ float shape_interior_ior(const DScene &sc, const uint32_t *shape_material, int shape, float lambda)
{
  if(shape < 0) return 1.0f;
  /* per shape: bsdf, material id, n_d, abbe of its material in one 16-B entry (instead of shape -> material -> bsdf -> params) */
  const uint4 e = ((const uint4 *)shape_material)[shape];
  if(e.x == MI_BSDF_DIELECTRIC) return eta_from_abbe(__uint_as_float(e.z), __uint_as_float(e.w), lambda);
  return 1.0f;
}

/* ------------------------------------------------------------------------------------------ homogeneous media
 * the volume a shape's interior prepare chain leaves on a surface vertex (interior.c:101-118, mult.c:154-167: the colour op in
 * the volume slot sets mu_s = clamp(albedo), mu_t = 1 (texture.h:48-53), then medium_rgb.c:45-59 scales both by its mu_t) */
struct Medium { float mu_s, mu_t, g; int med; };
__device__ __forceinline__ Medium medium_vacuum() { Medium m; m.mu_s = 0.0f; m.mu_t = 0.0f; m.g = 0.0f; m.med = -1; return m; }
__device__ __forceinline__ Medium shape_interior_medium(const DScene &sc, int shape, float lambda)
{
  Medium m = medium_vacuum();
  const DShapeMedium &e = sc.shape_medium[shape < 0 ? sc.exterior_index : (uint32_t)shape];   /* outside every shape: the exterior medium */
  if(e.med < 0) return m;
  const float val = e.albedo[3]*spectrum_eval(e.albedo, lambda);
  float mu_s = DCLAMP(val, 0.0f, 1.0f);
  const float old_mu_t = 1.0f;
  m.mu_t = e.mu_t[3]*spectrum_eval(e.mu_t, lambda);
  mu_s = mu_s*(m.mu_t/old_mu_t);
  m.mu_s = mu_s; m.g = e.g; m.med = e.med;
  return m;
}
__device__ __forceinline__ float eval_hg(float g, const V3 wi, const V3 wo)
{ /* sample_eval_hg, include/sampler_common.h:338-355 */
  if(g == 0.0f) return (float)(1.0/(4.0*MI_PI_D));
  const float cos_theta = dot3(wi, wo);
  return (float)(1.0/(4.0*MI_PI_D)*(double)(1.0f-g*g)/(double)powf(1.0f + g*g - 2.0f*g*cos_theta, 3.0f/2.0f));   /* the reference's double promotion */
}

/* ------------------------------------------------------------------------------------------ GGX, src/shaders/ggx.h */
__device__ __forceinline__ float ggx_G1(const V3 w, const V3 n, float roughness)
{ /* ggx.h:29-36 */
  const float r2 = roughness*roughness;
  const float cos_th = fabsf(dot3(w, n));
  const float sin_th = mi_sqrt(fmaxf(0.0f, 1.0f - cos_th*cos_th));
  const float tan_th = sin_th/cos_th;
  return 2.0f*mi_rcp(1.0f + mi_sqrt(1.0f + r2*tan_th*tan_th));     /* 2/x == 2*RN(1/x): scaling by two commutes with rounding */
}
__device__ __forceinline__ float ggx_G1_cos(float cos_wn, float roughness)
{ /* ggx.h:38-46 */
  const float r2 = roughness*roughness;
  const float sin_wn = mi_sqrt(DCLAMP(1.0f - cos_wn*cos_wn, 0.0f, 1.0f));
  const float tan_th = sin_wn/cos_wn;
  return 2.0f*mi_rcp(1.0f + mi_sqrt(1.0f + r2*tan_th*tan_th));     /* 2/x == 2*RN(1/x): scaling by two commutes with rounding */
}
__device__ __forceinline__ void ggx_sample11(float tan_theta_i, float U1, float U2, float &slope_x, float &slope_y)
{ /* ggx.h:59-110 */
  if(tan_theta_i < 0.0001f)
  {
    const float r = mi_sqrt(U1/fmaxf(1e-8f, 1-U1));
    const float phi = (float)(2.0f*MI_PI_D*(double)U2);
    float sn, cs;
    mi_sincosf(phi, &sn, &cs);
    slope_x = r*cs;
    slope_y = r*sn;
    return;
  }
  const float a = mi_rcp(tan_theta_i);
  const float G1 = 2.0f*mi_rcp(1.0f + mi_sqrt(1.0f + mi_rcp(a*a)));
  const float A = 2.0f*U1/G1 - 1.0f;
  const float tmp = mi_rcp(A*A - 1.0f);
  const float B = tan_theta_i;
  const float D = mi_sqrt(fmaxf(0.0f, B*B*tmp*tmp - (A*A - B*B)*tmp));
  float sx1 = B*tmp - D, sx2 = B*tmp + D;
  if(!(fabsf(sx1) < FLT_MAX)) sx1 = 0.0f;
  if(!(fabsf(sx2) < FLT_MAX)) sx2 = 0.0f;
  slope_x = (A < 0.0f || sx2*tan_theta_i > 1.0f) ? sx1 : sx2;
  float S;
  if(U2 > 0.5f) { S = 1.0f;  U2 = 2.0f*(U2 - 0.5f); }
  else          { S = -1.0f; U2 = 2.0f*(0.5f - U2); }
  const float z = (U2*(U2*(U2*(-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (U2*(U2*(U2*(U2*0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
  slope_y = S*z*mi_sqrt((float)(1.0 + (double)(slope_x*slope_x)));
}
__device__ __forceinline__ V3 ggx_sample_h(const V3 wi, float rx, float ry, float U1, float U2)
{ /* ggx.h:115-162 */
  const V3 wi_ = normalise3(mk3(rx*wi.x, ry*wi.y, fabsf(wi.z)));
  float tan_theta = 0.0f, sin_phi = 0.0f, cos_phi = 1.0f;
  if(wi_.z < 0.99999)
  {
    const float len = mi_sqrt(wi_.x*wi_.x + wi_.y*wi_.y);
    tan_theta = len/wi_.z;
    sin_phi = wi_.y/len;
    cos_phi = wi_.x/len;
  }
  float slope_x, slope_y;
  ggx_sample11(tan_theta, U1, U2, slope_x, slope_y);
  const float tmp = cos_phi*slope_x - sin_phi*slope_y;
  slope_y = sin_phi*slope_x + cos_phi*slope_y;
  slope_x = tmp;
  slope_x = rx*slope_x;
  slope_y = ry*slope_y;
  const float inv_h = mi_sqrt((float)((double)(slope_x*slope_x + slope_y*slope_y) + 1.0));
  V3 h = mk3(-slope_x/inv_h, -slope_y/inv_h, (float)(1.0/(double)inv_h));
  if(!(inv_h > 0.0)) h = mk3(0.0f, 1.0f, 0.0f);
  return h;
}
__device__ __forceinline__ float ggx_pdf_h(const V3 wi, const V3 h, const V3 n, float roughness)
{ /* ggx.h:167-182 */
  const float r2 = roughness*roughness;
  const float cos_th = fabsf(dot3(h, n));
  const float sin_th = mi_sqrt(fmaxf(0.0f, 1.0f - cos_th*cos_th));
  const float tan_th = sin_th/cos_th;
  const double A2 = (double)(r2 + tan_th*tan_th);
  const float D_h = (float)((double)r2/(MI_PI_D*(double)cos_th*(double)cos_th*(double)cos_th*(double)cos_th*A2*A2));
  const float G1 = ggx_G1(wi, n, roughness);
  return fabsf(G1*dot3(wi, h)*D_h/dot3(wi, n));
}
__device__ __forceinline__ float ggx_pdf_h_cos(float cosh, float cos_in, float cosr, float roughness)
{ /* ggx.h:184-201 */
  const float r2 = roughness*roughness;
  const float cosh2 = cosh*cosh;
  const float sin_th = mi_sqrt(DCLAMP(1.0f - cosh2, 0.0f, 1.0f));
  const float tan_th = sin_th/fabsf(cosh);
  const float den = tan_th*tan_th + r2;
  const float ct4 = cosh2*cosh2;
  const float D_h = (float)((double)r2/((MI_PI_D*(double)ct4)*(double)(den*den)));
  const float G1 = ggx_G1_cos(cos_in, roughness);
  return fabsf((G1*cosr)*(D_h/cos_in));
}

#define HALFVEC_COS_THR .999
#define GLOSSY_THR 1e-3f

__device__ __forceinline__ float fresnel_dielectric(float n1, float n2, float cosr, float cost)
{ /* dielectric.c:83-94 */
  if(cost <= 0.0f) return 1.0f;
  const float r1 = n1*cosr, r2 = n2*cosr, t1 = n1*cost, t2 = n2*cost;
  const float Rs = (r1 - t2)/(r1 + t2);
  const float Rp = (t1 - r2)/(t1 + r2);
  return DCLAMP((Rs*Rs + Rp*Rp)*.5f, 0.0f, 1.0f);
}

__device__ __forceinline__ float fresnel_metal(float n1, float n2, float k2, float cosr)
{ /* metal.c:79-157 */
  const float etar =   (n1*n2)/(n2*n2 + k2*k2);
  const float etai = -((n1*k2)/(n2*n2 + k2*k2));
  const float eta2r = etar*etar - etai*etai;
  const float eta2i = (2.0f*etar)*etai;
  const float sinr = 1.0f - cosr*cosr;
  const float cost2r = 1.0f - eta2r*sinr;
  const float cost2i = eta2i*(-sinr);
  const float len = mi_sqrt(cost2r*cost2r + cost2i*cost2i);
  const float costr = mi_sqrt(0.5f*(cost2r + len));
  float costi = mi_sqrt(0.5f*(len - cost2r));
  if(cost2i < 0.0f) costi = -costi;
  const float n1cosr = n1*cosr, n2cosrr = n2*cosr, n2cosri = k2*cosr;
  const float n1costr = n1*costr, n1costi = n1*costi;
  const float n2costr = n2*costr - k2*costi;
  const float n2costi = k2*costr + n2*costi;
  const float Rs2 = ((n1cosr - n2costr)*(n1cosr - n2costr) + n2costi*n2costi) /
                    ((n1cosr + n2costr)*(n1cosr + n2costr) + n2costi*n2costi);
  const float Rp2 = ((n1costr - n2cosrr)*(n1costr - n2cosrr) + (n1costi - n2cosri)*(n1costi - n2cosri)) /
                    ((n1costr + n2cosrr)*(n1costr + n2cosrr) + (n1costi + n2cosri)*(n1costi + n2cosri));
  return DCLAMP((Rs2 + Rp2)*.5f, 0.0f, 1.0f);
}

/* MI_METAL_REFERENCE (mi_scene_set_metal_reference): what the reference BUILD's metal sample() does on top of fresnel_metal. Its
 * plugin (gcc 11 -O3 -ffast-math -march=x86-64-v3; disassembly of libmetal.so, sample+0x1bb..0x259) forms costi as
 * sqrt(0.5 fma(eta2r, sinr, len - 1)) with cost2r = fma(-eta2r, sinr, 1), len = sqrt(fma(cost2r, cost2r, cost2i^2)): near normal
 * incidence on the microfacet len == cost2r and what is left under the root is the rounding error of cost2r, negative for every
 * other sample -- NaN, clamped to R = 0, the path ends. Half of the samples with sin^2 < 2.4e-4 cost2r / |eta2i| (3e-3 for gold at
 * 525 nm, 5e-2 at 720 nm), 2-4 % of all samples of a rough gold surface; its brdf() / pdf() are compiled differently and lose
 * nothing. Restated operation by operation (fused where the plugin fuses), same predicate in the CPU restatement used by the tests. */
__device__ __forceinline__ bool metal_reference_kills(float n1, float n2, float k2, float cosr)
{
  const float sinr = __builtin_fmaf(-cosr, cosr, 1.0f);
  const float den = __builtin_fmaf(n2, n2, k2*k2);
  const float etar = (n1*n2)/den;
  const float etai = -((k2*n1)/den);
  const float cost2i = (__builtin_fmaf(cosr, cosr, -1.0f)*-2.0f)*(etar*etai);
  const float eta2r = __builtin_fmaf(etar, etar, -(etai*etai));
  const float cost2r = __builtin_fmaf(-eta2r, sinr, 1.0f);
  const float len = sqrtf(__builtin_fmaf(cost2r, cost2r, cost2i*cost2i));
  return 0.5f*__builtin_fmaf(eta2r, sinr, len - 1.0f) < 0.0f;
}

#define MI_MF 4      /* wavelengths per path in the HERO kernels (mi_hero.h) */

/* ------------------------------------------------------------------------------------------ bsdf sampling */
struct BsdfSample
{
  V3 omega;          /* e[v+1].omega (not yet normalised by shader_sample) */
  float pdf;         /* v[v+1].pdf as set by the plugin (projected solid angle) */
  float weight;      /* returned throughput factor */
  uint32_t mode;     /* v[v].mode after sampling */
};

template<class PS>
__device__ __forceinline__ void sample_diffuse(PS &pts, const Surf &sf, const Shading &sh, uint32_t mode_in, BsdfSample &bs)
{ /* sample_d, src/shader.c:165-205 */
  const float x1 = pts(MI_DIM_OMEGA_X);
  const float x2 = pts(MI_DIM_OMEGA_Y);
  const float sq = mi_sqrt(x1);
  const float c0 = mi_sqrt((float)(1.0 - (double)x1));
  const float ang = (float)(2*MI_PI_D*(double)x2);
  float sn, cs;
  mi_sincosf(ang, &sn, &cs);                 /* one range reduction for both (same values as sinf/cosf) */
  const float c1 = sq*cs, c2 = sq*sn;
  bs.omega = mk3(c0*sf.n.x + c1*sf.a.x + c2*sf.b.x, c0*sf.n.y + c1*sf.a.y + c2*sf.b.y, c0*sf.n.z + c1*sf.a.z + c2*sf.b.z);
  bs.pdf = (float)(1.0f/MI_PI_D);
  bs.mode = mode_in;
  bs.weight = 0.0f;
  const float cos_out_ng = dot3(sf.gn, bs.omega);
  if(sf.flags & s_inside) { if(cos_out_ng >= 0.0f) return; }
  else if(cos_out_ng <= 0.0f) return;
  bs.weight = sh.rd;
  if(bs.weight > 0.0f) bs.mode = s_diffuse | s_reflect;
}

template<class PS>
__device__ __forceinline__ void sample_dielectric(PS &pts, const Surf &sf, const Shading &sh, const V3 wi, float eta_ratio,
                                                  uint32_t mode_in, BsdfSample &bs)
{ /* sample, dielectric.c:240-415 (MF_COUNT == 1) */
  bs.mode = mode_in; bs.weight = 0.0f; bs.pdf = 1.0f; bs.omega = mk3(0, 0, 0);
  if(eta_ratio < 0.0f) return;
  if(fabsf(1.0f - eta_ratio/1.0f) < 1e-3f)
  {
    bs.omega = wi;
    bs.mode = s_specular | s_transmit;
    bs.pdf = 1.0f;
    bs.weight = sh.rg;
    return;
  }
  const V3 n = sf.n;
  float pdf_h = 1.0f;
  V3 h = n;
  const float r = sh.roughness;
  const float cos_in = -dot3(sf.n, wi);
  if(r > GLOSSY_THR)
  {
    const V3 wit = mk3(-dot3(sf.a, wi), -dot3(sf.b, wi), cos_in);
    const float U2 = pts(MI_DIM_OMEGA_Y);          /* argument evaluation order of the reference build, SURVEY app. B */
    const float U1 = pts(MI_DIM_OMEGA_X);
    const V3 ht = ggx_sample_h(wit, r, r, U1, U2);
    h = mk3(ht.x*sf.a.x + ht.y*sf.b.x + ht.z*n.x, ht.x*sf.a.y + ht.y*sf.b.y + ht.z*n.y, ht.x*sf.a.z + ht.y*sf.b.z + ht.z*n.z);
    pdf_h = ggx_pdf_h(wi, h, n, r);
  }
  float pdf = pdf_h;
  const float cosr = -dot3(wi, h);
  if(cosr <= 0.0f) return;
  const float n1 = eta_ratio, n2 = 1.0f;
  const float nr = n1/n2;
  const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
  const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
  const float R = fresnel_dielectric(n1, n2, cosr, cost);
  if(pts(MI_DIM_SCATTER_MODE) <= R)
  {
    bs.mode = s_reflect;
    bs.omega = mk3(wi.x + 2.0f*cosr*h.x, wi.y + 2.0f*cosr*h.y, wi.z + 2.0f*cosr*h.z);
    if(dot3(bs.omega, n) <= 0.0f) return;
    pdf *= mi_rcp(4.0f*cosr);
    if(r > GLOSSY_THR)
    {
      bs.pdf = R*(pdf/fabsf(dot3(bs.omega, n)));
      bs.mode |= s_glossy;
      if(dot3(bs.omega, n)*dot3(bs.omega, h) < 0.0f) return;
      bs.weight = sh.rg*ggx_G1(bs.omega, n, sh.roughness);
      return;
    }
    bs.pdf = R;
    bs.mode = s_reflect | s_specular;
    bs.weight = sh.rg;
  }
  else
  {
    if(cost2 <= 0.0f) return;
    const float f = eta_ratio*cosr - cost;
    bs.omega = normalise3(mk3(wi.x*eta_ratio + f*h.x, wi.y*eta_ratio + f*h.y, wi.z*eta_ratio + f*h.z));
    if(dot3(bs.omega, n) >= 0.0f) return;
    if(r <= GLOSSY_THR)
    {
      bs.pdf = 1.0f - R;
      bs.mode = s_specular | s_transmit;
      bs.weight = sh.rg;
      return;
    }
    const float denom = n1*cosr - n2*cost;
    pdf *= n2*n2*cost/(denom*denom);
    bs.pdf = (pdf*(1.0f - R))/fabsf(dot3(bs.omega, n));
    bs.mode = s_transmit | s_glossy;
    bs.weight = sh.rg*ggx_G1(bs.omega, n, sh.roughness);
  }
}

template<class PS>
__device__ __forceinline__ void sample_metal(const DScene &sc, PS &pts, const Surf &sf, const Shading &sh, const V3 wi, float n1,
                                             int mat, float lambda, uint32_t mode_in, BsdfSample &bs)
{ /* sample, metal.c:219-265 */
  bs.mode = mode_in; bs.weight = 0.0f; bs.pdf = 1.0f; bs.omega = mk3(0, 0, 0);
  const V3 n = sf.n;
  V3 h = n;
  float pdf_h = 1.0f;
  const float r = sh.roughness;
  if(r > 1e-4f)
  {
    const V3 wit = mk3(-dot3(sf.a, wi), -dot3(sf.b, wi), -dot3(n, wi));
    const float U2 = pts(MI_DIM_OMEGA_Y);
    const float U1 = pts(MI_DIM_OMEGA_X);
    const V3 ht = ggx_sample_h(wit, r, r, U1, U2);
    h = mk3(ht.x*sf.a.x + ht.y*sf.b.x + ht.z*n.x, ht.x*sf.a.y + ht.y*sf.b.y + ht.z*n.y, ht.x*sf.a.z + ht.y*sf.b.z + ht.z*n.z);
    pdf_h = ggx_pdf_h(wi, h, n, r);
  }
  float pdf = pdf_h;
  const float cosr = -dot3(wi, h);
  if(!(cosr > 0.0f)) return;
  const int i = (int)DCLAMP((lambda - 360.0f)/5.0f, 0, 94);              /* fresnel.h:519-531 */
  const float n2 = sc.metal_ior[(mat*95 + i)*2 + 0], k2 = -sc.metal_ior[(mat*95 + i)*2 + 1];
  const float R = (sc.metal_reference && metal_reference_kills(n1, n2, k2, cosr)) ? 0.0f : fresnel_metal(n1, n2, k2, cosr);
  bs.mode = s_reflect;
  bs.omega = mk3(wi.x + 2.0f*cosr*h.x, wi.y + 2.0f*cosr*h.y, wi.z + 2.0f*cosr*h.z);
  if(dot3(bs.omega, n) <= 0.0f) return;
  pdf *= mi_rcp(4.0f*cosr);
  if(r > 1e-4f)
  {
    bs.pdf = pdf/fabsf(dot3(bs.omega, n));
    bs.mode |= s_glossy;
    if(dot3(bs.omega, n)*dot3(bs.omega, h) < 0.0f) return;
    bs.weight = R*(sh.rg*ggx_G1(bs.omega, n, sh.roughness));
    return;
  }
  bs.mode |= s_specular;
  bs.weight = R*sh.rg;
}

/* ------------------------------------------------------------------------------------------ bsdf evaluation / pdf (ptdl) */
struct BsdfEval { float value; uint32_t mode; };

__device__ __forceinline__ BsdfEval brdf_diffuse(const Surf &sf, const Shading &sh, const V3 wo)
{ /* brdf_d, src/shader.c:207-252 (path tracing direction) */
  BsdfEval r; r.mode = s_diffuse | s_reflect; r.value = 0.0f;
  const float cos_out_ns = dot3(sf.n, wo);
  if(cos_out_ns <= 0) return r;
  const float cos_out_ng = dot3(sf.gn, wo);
  if(sf.flags & s_inside) { if(cos_out_ng >= 0.0f) return r; }
  else if(cos_out_ng <= 0.0f) return r;
  r.value = (float)((double)sh.rd*((double)1.0f/MI_PI_D));
  return r;
}

__device__ __forceinline__ BsdfEval brdf_dielectric(const Surf &sf, const Shading &sh, const V3 wi, const V3 wo, float eta_ratio)
{ /* brdf, dielectric.c:418-541 (scalar) */
  BsdfEval res; res.value = 0.0f; res.mode = s_absorb;
  const V3 n = sf.n;
  const float cos_in  = -dot3(n, wi);
  const float cos_out =  dot3(n, wo);
  if(eta_ratio < 0.0f) return res;
  const float n1 = eta_ratio, n2 = 1.0f;
  const bool index_matched = fabsf(1.0f - n1/n2) < 1e-3f;
  if(cos_out == 0.0f || cos_in == 0.0f) return res;
  if(!index_matched && (cos_in*cos_out > 0)) res.mode = s_reflect;
  else res.mode = s_transmit;
  const float r = sh.roughness;
  if((r > GLOSSY_THR) && !index_matched) res.mode |= s_glossy;
  else res.mode |= s_specular;
  if(index_matched)
  {
    const float dot_wo_n = dot3(wo, n);
    const V3 h = normalise3(mk3(-wi.x + wo.x - 2.0f*dot_wo_n*n.x, -wi.y + wo.y - 2.0f*dot_wo_n*n.y, -wi.z + wo.z - 2.0f*dot_wo_n*n.z));
    const float cosh = dot3(h, n);
    if(cosh < 0.0f) return res;
    if(cosh < HALFVEC_COS_THR) return res;
    res.value = sh.rg;
    return res;
  }
  else if(res.mode & s_reflect)
  {
    const V3 h = normalise3(mk3(-wi.x + wo.x, -wi.y + wo.y, -wi.z + wo.z));
    const float cosh = dot3(h, n);
    if(cosh < 0.0f) return res;
    const float DG1 = (res.mode & s_specular) ? 1.0f : ggx_pdf_h(wi, h, n, r);
    if(DG1 == 0) return res;
    const float cosr = -dot3(h, wi);
    if(cosr < 0.0f) return res;
    const float nr = n1/n2;
    const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
    const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
    const float R = fresnel_dielectric(n1, n2, cosr, cost);
    const float G1 = ggx_G1(wo, n, r);
    if(res.mode & s_glossy) { res.value = (sh.rg*R)*(DG1*G1/(4.0f*fabsf(cosr*cos_out))); return res; }
    if(cosh < HALFVEC_COS_THR) return res;
    res.value = sh.rg*R;
    return res;
  }
  else
  {
    bool mask = false;
    float h0 = n1*wi.x - n2*wo.x, h1 = n1*wi.y - n2*wo.y, h2 = n1*wi.z - n2*wo.z;
    const float hilen = mi_rcp(mi_sqrt(h0*h0 + (h1*h1 + h2*h2)));
    h0 *= hilen; h1 *= hilen; h2 *= hilen;
    float cosh2 = h0*n.x + (h1*n.y + h2*n.z);
    const bool cosh_lt0 = cosh2 < 0.0f;
    mask |= cosh_lt0 && (n1 < n2);
    mask |= !cosh_lt0 && (n2 < n1);
    if(cosh_lt0) { cosh2 = -cosh2; h0 = -h0; h1 = -h1; h2 = -h2; }
    const float cosr2 = h0*-wi.x + (h1*-wi.y + h2*-wi.z);
    mask |= cosr2 <= 0.0f;
    const float nr = n1/n2;
    const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr2*cosr2);
    const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
    const float R2 = fresnel_dielectric(n1, n2, cosr2, cost);
    const float DG1 = ggx_pdf_h_cos(cosh2, cos_in, cosr2, r);
    const float G1 = ggx_G1_cos(cos_in, r);
    const float cos_hwo = h0*wo.x + (h1*wo.y + h2*wo.z);
    mask |= cos_hwo >= 0.0f;
    float denom = n1*cosr2 - n2*cost;
    denom = denom*denom;
    if(cos_in == 0.0f) return res;
    if(res.mode & s_glossy)
    {
      res.value = mask ? 0.0f : ((sh.rg*(1.0f - R2))*((n2*n2)*(cost*(DG1*(G1*(1.0f/fabsf(cos_out)))))))/denom;
      return res;
    }
    mask |= cosh2 < HALFVEC_COS_THR;
    res.value = mask ? 0.0f : sh.rg*DCLAMP(1.0f - R2, 0.0f, 1.0f);
    return res;
  }
}

__device__ __forceinline__ float pdf_dielectric(const Surf &sf, const Shading &sh, const V3 wi, const V3 wo, float eta, uint32_t mode)
{ /* pdf, dielectric.c:96-237 (forward direction) */
  const V3 n = sf.n;
  const float cos_in  = -dot3(n, wi);
  const float cos_out =  dot3(n, wo);
  if(cos_in*cos_out == 0.0f) return 0.0f;
  if(cos_out > 0.0f && !(mode & s_reflect))  return 0.0f;
  if(cos_out < 0.0f && !(mode & s_transmit)) return 0.0f;
  if(eta < 0.0f) return 0.0f;
  const float n1 = eta, n2 = 1.0f;
  bool mask = false;
  float cosr = 0.0f, cosh = 0.0f;
  V3 h;
  if(fabsf(1.0f - n1/n2) < 1e-3f)
  {
    const float dot_wo_n = dot3(wo, n);
    h = normalise3(mk3(-wi.x + wo.x - 2.0f*dot_wo_n*n.x, -wi.y + wo.y - 2.0f*dot_wo_n*n.y, -wi.z + wo.z - 2.0f*dot_wo_n*n.z));
    cosh = dot3(h, n);
    if(mode != (s_transmit | s_specular)) return 0.0f;
    if(cosh < HALFVEC_COS_THR) return 0.0f;
    return 1.0f;
  }
  else if(mode & s_reflect)
  {
    h = normalise3(sub3(wi, wo));
    cosh = fabsf(dot3(h, n));
    cosr = fabsf(dot3(h, wi));
  }
  else
  {
    float h0 = n1*wi.x - n2*wo.x, h1 = n1*wi.y - n2*wo.y, h2 = n1*wi.z - n2*wo.z;
    const float hilen = mi_rcp(mi_sqrt(h0*h0 + (h1*h1 + h2*h2)));
    h0 *= hilen; h1 *= hilen; h2 *= hilen;
    if(n2 < n1) { h0 = -h0; h1 = -h1; h2 = -h2; }
    h = mk3(h0, h1, h2);
    cosh = h0*n.x + (h1*n.y + h2*n.z);
    mask |= cosh < 0.0f;
    cosr = h0*-wi.x + (h1*-wi.y + h2*-wi.z);
    mask |= cosr <= 0.0f;
  }
  const float nr = n1/n2;
  const float cost2 = 1.0f - (nr*nr)*(1.0f - cosr*cosr);
  const float cost = cost2 <= 0.0f ? 0.0f : mi_sqrt(cost2);
  const float R = fresnel_dielectric(n1, n2, cosr, cost);
  float pdf = 1.0f;
  if(mode & s_reflect)
  {
    if(mode & s_specular) { mask |= cosh < HALFVEC_COS_THR; return mask ? 0.0f : R; }
    pdf = pdf*mi_rcp(4.0f*fabsf(dot3(wo, h)));
    pdf = pdf*R;
  }
  else
  {
    if(mode & s_specular) { mask |= cosh < HALFVEC_COS_THR; return mask ? 0.0f : DCLAMP(1.0f - R, 0.0f, 1.0f); }
    const float denom = n1*cosr - n2*cost;
    pdf = pdf*(((n2*n2)*cost)/(denom*denom));
    pdf = pdf*DCLAMP(1.0f - R, 0.0f, 1.0f);
  }
  pdf = pdf*ggx_pdf_h_cos(cosh, cos_in, cosr, sh.roughness);
  pdf = pdf/fabsf(cos_out);
  mask |= !(pdf > 0.0f);
  return mask ? 0.0f : pdf;
}

__device__ __forceinline__ BsdfEval brdf_metal(const DScene &sc, const Surf &sf, const Shading &sh, const V3 wi, const V3 wo, float n1, int mat, float lambda)
{ /* brdf, metal.c:268-310 */
  BsdfEval res; res.value = 0.0f; res.mode = s_absorb;
  const V3 n = sf.n;
  const float cos_in = -dot3(n, wi), cos_out = dot3(n, wo);
  if(cos_out <= 0.0f || cos_in <= 0.0f) return res;
  res.mode = s_reflect;
  const float r = sh.roughness;
  if(r > 1e-4f) res.mode |= s_glossy; else res.mode |= s_specular;
  const V3 h = normalise3(mk3(-wi.x + wo.x, -wi.y + wo.y, -wi.z + wo.z));
  const float cosh = dot3(h, n);
  if(cosh < 0.0f) return res;
  const float DG1 = ggx_pdf_h(wi, h, n, r);
  if(DG1 == 0) return res;
  const float cosr = -dot3(h, wi);
  if(cosr < 0.0f) return res;
  const int i = (int)DCLAMP((lambda - 360.0f)/5.0f, 0, 94);
  const float n2 = sc.metal_ior[(mat*95 + i)*2 + 0], k2 = -sc.metal_ior[(mat*95 + i)*2 + 1];
  const float R = fresnel_metal(n1, n2, k2, cosr);
  const float G1 = ggx_G1(wo, n, r);
  if(res.mode & s_glossy) { res.value = (sh.rg*R)*(DG1*G1/(4.0f*fabsf(cosr*cos_out))); return res; }
  if(cosh < HALFVEC_COS_THR) return res;
  res.value = sh.rg*R;
  return res;
}

__device__ __forceinline__ float pdf_metal(const Surf &sf, const Shading &sh, const V3 wi, const V3 wo, uint32_t mode)
{ /* pdf, metal.c:170-216 */
  if(!(mode & s_reflect)) return 0.0f;
  const V3 n = sf.n;
  const float cos_in = -dot3(n, wi), cos_out = dot3(n, wo);
  if(cos_in < 0.0f) return 0.0f;
  if(cos_out < 0.0f) return 0.0f;
  const V3 h = normalise3(sub3(wi, wo));
  if(mode & s_specular)
  {
    const float cosh = fabsf(dot3(h, n));
    if(cosh < HALFVEC_COS_THR) return 0.0f;
    return 1.0f;
  }
  float pdf = mi_rcp(4.0f*fabsf(dot3(wo, h)));
  pdf *= ggx_pdf_h(wi, h, n, sh.roughness);
  pdf /= fabsf(cos_out);
  if(!(pdf > 0.0f)) return 0.0f;
  return pdf;
}

/* ------------------------------------------------------------------------------------------ emitter sampling (ptdl) */
__device__ __forceinline__ uint32_t sample_cdf(const float *cdf, int num, float rand)
{ /* sample_cdf, include/sampler_common.h:206-226 */
  unsigned int mn = 0, mx = num;
  unsigned int t = mx/2;
  while(t != mn)
  {
    if(cdf[t] <= rand) mn = t;
    else mx = t;
    t = (mn + mx)/2;
  }
  if(mx < (unsigned)num && cdf[t] <= rand) t = mx;
  return t;
}

__device__ __forceinline__ uint32_t sample_cdf4(const float *c4, int num, float rand)
{ /* sample_cdf over at most four entries held in scalar registers: the same search, no memory round trips */
  unsigned int mn = 0, mx = num;
  unsigned int t = mx/2;
#define MI_CDF4(T) ((T) == 0 ? c4[0] : (T) == 1 ? c4[1] : (T) == 2 ? c4[2] : c4[3])
  for(int it=0;it<3 && t != mn;it++)          /* mx - mn halves from at most 4: at most two steps */
  {
    if(MI_CDF4(t) <= rand) mn = t;
    else mx = t;
    t = (mn + mx)/2;
  }
  if(mx < (unsigned)num && MI_CDF4(t) <= rand) t = mx;
#undef MI_CDF4
  return t;
}

__device__ __forceinline__ V3 tri_retime(const V3 v0, const V3 v1, const V3 v2, float u, float v)
{ /* geo_tri_retime, include/geo/triangle.h:51-61 */
  const float w = 1.0f - u - v;
  return mk3(w*v0.x + v*v1.x + u*v2.x, w*v0.y + v*v1.y + u*v2.y, w*v0.z + v*v1.z + u*v2.z);
}

/* a point on a sphere / on the mantle of a cone or cylinder (geo_sphere_retime, include/geo/sphere.h:38-49; geo_line_retime, include/geo/line.h:88-121) */
__device__ __forceinline__ V3 sphere_sample_at(const V3 c, float radius, float r0, float r1, float &hu, float &hv)
{
  hu = r0; hv = (float)((double)acosf(r1)/MI_PI_D);
  const float x1 = (float)((double)(-(mi_cosf((float)((double)hv*MI_PI_D))-1.f))/2.0), x2 = hu;
  const float z = 1.f - 2.f*x1, rr = mi_sqrt(1.f - z*z);
  const float phi = (float)(2.f*MI_PI_D*(double)x2);
  return mk3(c.x + radius*(rr*mi_cosf(phi)), c.y + radius*(rr*mi_sinf(phi)), c.z + radius*z);
}
__device__ __forceinline__ V3 line_sample_at(const V3 v0, const V3 v1, float lr0, float lr1, float r0, float r1, float &hu, float &hv)
{
  hu = r0; hv = r1;
  float y;
  if(fabsf(lr1-lr0) < 1e-3f) y = hu;
  else y = (mi_sqrt((lr1*lr1 - lr0*lr0)*hu + lr0*lr0) - lr0)/(lr1-lr0);
  const float phi = (float)(2.0*MI_PI_D*(double)hv);
  float sinphi, cosphi; mi_sincosf(phi, &sinphi, &cosphi);
  V3 d = sub3(v1, v0);
  d = scale3(d, mi_rcp(mi_sqrt(dot3(d, d))));
  V3 a, b; get_onb(d, a, b);
  return mk3(v0.x + (v1.x - v0.x)*y + a.x*sinphi + b.x*cosphi, v0.y + (v1.y - v0.y)*y + a.y*sinphi + b.y*cosphi,
             v0.z + (v1.z - v0.z)*y + a.z*sinphi + b.z*cosphi);
}

template<bool MB = false>
__device__ __forceinline__ V3 prim_sample(const DPrim &p, const DPrimGeo &geo, float r0, float r1, float &hu, float &hv,
                                          const DPrimT1 *t1 = nullptr, float time = 0.0f)
{ /* prims_sample + prims_retime, src/prims.c:178-252 */
  const bool ordered = p.type == 0 && p.pad[1] == MI_PRIM_ORDERED;     /* a static triangle / quad under its leaf-order mark (mi_mark_ordered_kernel) */
  if(MB && p.type == 0 && !ordered && p.pad[0] < MI_PRIM_TRI)
  { /* moving emitter (sphere / cone / cylinder): centre / end points at the path's time (geo_get_vertex_time), the radii of the shutter-open
       vertices -- the record moving_analytic_at makes for the intersection */
    V3 a0, a1;
    (void)moving_analytic_at(p, *t1, time, a0, a1);
    if(p.pad[0] == MI_PRIM_SPHERE) return sphere_sample_at(a0, p.v[2][0], r0, r1, hu, hv);
    return line_sample_at(a0, a1, p.v[2][0], p.v[2][1], r0, r1, hu, hv);
  }
  if(MB && p.type == 0 && !ordered)
  { /* moving emitter (triangle / quad): the record holds the shutter-open vertices, *t1 the shutter-close ones; sample the
       primitive as it is at the path's time */
    const float w0 = 1.0f - time, w1 = time;
    V3 vt[4];
    for(int k=0;k<4;k++) vt[k] = mk3(w0*p.v[k][0] + w1*t1->v[k][0], w0*p.v[k][1] + w1*t1->v[k][1], w0*p.v[k][2] + w1*t1->v[k][2]);
    if(p.pad[0] == MI_PRIM_QUAD)
    {
      hu = r0; hv = r1;
      if(hv >= hu) return tri_retime(vt[0], vt[1], vt[2], hu, hv - hu);
      return tri_retime(vt[0], vt[2], vt[3], hu - hv, hv);
    }
    const float a = mi_sqrt(r0);
    hu = r1*a; hv = (1.0f-r1)*a;
    return tri_retime(vt[0], vt[1], vt[2], hu, hv);
  }
  const uint32_t type = ordered ? p.pad[0] : p.type;
  const float *gv = geo.f + 26;             /* v1, v2, v3 of a triangle / quad (DPrim keeps v0 and the edges) */
  if(type == MI_PRIM_QUAD)
  {
    hu = r0; hv = r1;
    if(hv >= hu) return tri_retime(ld3(p.v[0]), ld3(gv), ld3(gv + 3), hu, hv - hu);
    return tri_retime(ld3(p.v[0]), ld3(gv + 3), ld3(gv + 6), hu - hv, hv);
  }
  if(type == MI_PRIM_TRI)
  {
    const float a = mi_sqrt(r0);
    hu = r1*a; hv = (1.0f-r1)*a;
    return tri_retime(ld3(p.v[0]), ld3(gv), ld3(gv + 3), hu, hv);
  }
  if(type == MI_PRIM_SPHERE) return sphere_sample_at(ld3(p.v[0]), p.v[1][0], r0, r1, hu, hv);
  const float *f = &p.v[0][0];
  return line_sample_at(mk3(f[0], f[1], f[2]), mk3(geo.f[26], geo.f[27], geo.f[28]), f[3], f[4], r0, r1, hu, hv);
}

/* ------------------------------------------------------------------------------------------ splat */
#ifndef MI_SPLAT_SKIP_ZERO
#define MI_SPLAT_SKIP_ZERO 1   /* the window is 0 beyond 1.5 pixels from the splat (filter_bh_w: n > N - 1): of the 4 x 4 taps the reference adds, 7 on average carry a weight,
                                  the others add 0.0 to their pixels -- an addition that changes no float of a film that starts at +0. Those taps issue no atomics
                                  (profiles/r06_levers.txt block 8) */
#endif
__device__ __forceinline__ float bh_w(float n)
{ /* filter_bh_w, include/filter/blackmanharris.h:28-41 */
  const float NN = 4.0f;
  if(n > NN-1.0f || n < 0.0f) return 0.0f;
  const float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
  const float N_1 = 1.0f/(NN-1.0f);
  const float cos1 = mi_cosf((float)(2.0f*MI_PI_D*(double)n*(double)N_1));
  const float cos2 = mi_cosf((float)(4.0f*MI_PI_D*(double)n*(double)N_1));
  const float cos3 = mi_cosf((float)(6.0f*MI_PI_D*(double)n*(double)N_1));
  return a0 - a1*cos1 + a2*cos2 - a3*cos3;
}

__device__ __forceinline__ bool splat_value_ok(float value)
{ /* view_splat, src/view.c:455-463 */
  return (value > 0.0f) && (value < FLT_MAX) && (value == value);
}

__device__ __forceinline__ void spectrum_to_xyz(const DScene &sc, float lambda, float value, float *col)
{ /* spectrum_p_to_camera, include/spectrum.h:172-203 */
  float f = (lambda - 360)/5;
  const int i = (int)f;
  f -= i;
  for(int k=0;k<3;k++) col[k] = ((1-f)*sc.cie_xyz[3*i+k] + f*sc.cie_xyz[3*(i+1)+k])*value;
}

/* filter_blackmanharris_splat (include/blackmanharris.h:43-77) + filter_box_splat (box.h:23-36), executed by the whole
 * wave for the lanes that have a splat pending, four splats per pass: each group of sixteen lanes takes one splat and
 * each of its lanes one of the 4x4 taps (3 cosf per tap instead of 96 per lane); the normalisation sum is formed in the
 * reference's tap order, then each tap lane issues its three hardware float atomics. */
__device__ __forceinline__ void splat_wave(const DScene &sc, bool pending, float pi, float pj, float c0, float c1, float c2)
{
  unsigned long long m = __ballot(pending);
  const unsigned lane = __lane_id();
  const unsigned grp = lane >> 4, base = lane & 48u;
  const int wd = (int)sc.width, ht = (int)sc.height;
  while(m)
  {
    /* group g serves the g-th pending lane of this pass */
    const unsigned long long m1 = m & (m - 1), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);
    const unsigned long long mine = grp == 0 ? m : grp == 1 ? m1 : grp == 2 ? m2 : m3;
    m = m3 & (m3 - 1);
    const bool have = mine != 0;
    const int src = have ? __ffsll((long long)mine) - 1 : 0;
    const float spi = __shfl(pi, src), spj = __shfl(pj, src);
    const float s0 = __shfl(c0, src), s1 = __shfl(c1, src), s2 = __shfl(c2, src);
    const int x0 = (int)(spi - 1.5f), y0 = (int)(spj - 1.5f);
    const int u0 = -x0 < 0 ? 0 : -x0, v0 = -y0 < 0 ? 0 : -y0;
    const int u4 = x0 + 4 > wd ? wd - x0 : 4, v4 = y0 + 4 > ht ? ht - y0 : 4;
    const int u = lane & 3, v = (lane >> 2) & 3;
    const bool inside = have && v >= v0 && v < v4 && u >= u0 && u < u4;
    float f = 0.0f;
    if(inside)
    {
      const float uu = (x0 + u + .5f) - spi, vv = (y0 + v + .5f) - spj;
      f = bh_w(mi_sqrt(uu*uu + vv*vv) + 1.5f);
    }
    float weight = 0.0f;
#pragma unroll
    for(int k=0;k<16;k++) weight += __shfl(f, base + k);     /* taps outside the image contribute exactly 0, as if skipped */
    if(inside && weight > 0)
    {
      const float g = mi_rcp(weight)*f;
#if defined(MI_EXP_SPLAT) && MI_EXP_SPLAT == 1     /* experiment (profiles/r06_levers.txt block 4): the filter is evaluated, nothing is written */
      if(g != g) sc.fb[0] = g;
      continue;
#endif
#if defined(MI_EXP_SPLAT) && MI_EXP_SPLAT == 2     /* experiment: every tap goes to one 32 x 32 tile of the film (what a tile buffer on chip would absorb: no traffic behind L2) */
      float *px = sc.fb + 3*((size_t)((x0+u) & 31) + (size_t)wd*((y0+v) & 31));
#else
      float *px = sc.fb + 3*((size_t)(x0+u) + (size_t)wd*(y0+v));
#endif
      const float a0 = s0*g, a1 = s1*g, a2 = s2*g;
#if MI_SPLAT_SKIP_ZERO
      if(a0 != 0.0f || a1 != 0.0f || a2 != 0.0f)      /* (a NaN is not 0: it is written as the reference writes it) */
#endif
      {
        atomicAdd(px+0, a0);
        atomicAdd(px+1, a1);
        atomicAdd(px+2, a2);
      }
    }
  }
}

#endif
