/* mi_abi.hip -- the persistent path tracing kernel and the C ABI (include/corona_mi.h) of
 * libcorona_mi.so for gfx950. Device helpers live in mi_kernels.h, the device layout in mi_device.h.
 *
 * Replaces everything the reference reaches from work_sample() (src/view.c:618-628): one launch
 * traces path indices [first, first+count) and splats them into the device framebuffer.
 */
#include "mi_megakernel.h"
#include "mi_wavefront.h"
#include "mi_build.h"
#include "mi_halton.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>

/* ======================================================================================= upload-time kernels */
/* Upload-time pass over the child links (one thread per link), once the scene knows how many nodes it stages in LDS: an inner link
 * (node index until now) becomes the child's RECORD OFFSET in 16-byte lanes (mi_device.h: n < K ? n SL : K SL + (n - K) SH) with the
 * child's split axes in bits 24..29 -- a node visit then needs neither a multiplication by the record size nor a separate look-up of
 * its axes. `axes` = axis0 | axis00 << 2 | axis01 << 4 per node, as the host builder / the device build leave them. */
__global__ void mi_bake_links_kernel(float4 *nodes, const uint32_t *axes, uint32_t N, uint32_t K, uint32_t SL, uint32_t SH)
{
  const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
  if(i >= 4u*N) return;
  uint32_t *link = (uint32_t *)(nodes + (size_t)(i >> 2)*SH + 6) + (i & 3u);
  const uint32_t l = *link;
  if(l & MI_LEAF32) return;
  const uint32_t n = l & MI_NODE_MASK;
  *link = (n < K ? n*SL : K*SL + (n - K)*SH) | (axes[n] << MI_AXES_SHIFT);
}

/* Upload-time pass over the leaves (one thread per child link of the tree, after the primitive records are in place -- for a
 * device-built tree after the build): the leaf loops test a leaf's triangles and quads first and put off its spheres, lines and
 * moving primitives. That changes nothing (see leaf_sequential) unless a quad can be crossed in both halves, which takes a folded
 * quad: a static quad whose fourth vertex leaves the plane of the first three by more than 1e-5 of its size and which follows a
 * primitive that is put off, or anything that follows a moving quad. From there to the end of its leaf every triangle / quad
 * is put off too (type 0, pad[0] = vertex count, pad[1] = MI_PRIM_ORDERED): the put-off tests run in the leaf's order. */
__global__ void mi_mark_ordered_kernel(const float4 *nodes, uint32_t N, uint32_t SH, DPrim *prims)
{
  const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
  if(i >= 4u*N) return;
  const uint32_t link = ((const uint32_t *)(nodes + (size_t)(i >> 2)*SH + 6))[i & 3u];
  if(!(link & MI_LEAF32)) return;
  const uint32_t first = (link ^ MI_LEAF32) >> 5, num = link & 31u;
  bool deferred = false, ordered = false;
  for(uint32_t k=0;k<num;k++)
  {
    DPrim &p = prims[first + k];
    if(p.type >= MI_PRIM_TRI)
    {
      if(!ordered && deferred && p.type == MI_PRIM_QUAD)
      {
        const V3 e1 = ld3(p.v[1]), e2 = ld3(p.v[2]), e3 = ld3(p.v[3]);
        const V3 n = cross3(e1, e2);
        const float vol = fabsf(dot3(n, e3)), ref = sqrtf(dot3(n, n))*sqrtf(dot3(e3, e3));
        if(!(vol <= 1e-5f*ref)) ordered = true;
      }
      if(ordered) { p.pad[0] = p.type; p.pad[1] = MI_PRIM_ORDERED; p.type = 0; }
    }
    else
    {
      deferred = true;
      if(p.type == 0 && p.pad[0] == MI_PRIM_QUAD) ordered = true;       /* a moving quad: its shape changes with time */
    }
  }
}

/* the classes of the primitives (DPrimGeo.cls, mi_regroup.h) two bits each, in the records' final order: a wave asks for the class of its
   lanes' hits once per iteration, between its traversal slice and the exchange -- from LDS where the table fits (16 primitives per word) */
__global__ void mi_pack_cls_kernel(const DPrimGeo *geo, uint32_t n, uint32_t *out)
{
  const uint32_t w = blockIdx.x*blockDim.x + threadIdx.x;
  if(16u*w >= n) return;
  uint32_t v = 0;
  for(uint32_t k=0;k<16u && 16u*w + k < n;k++) v |= (geo[16u*w + k].cls & 3u) << (2u*k);
  out[w] = v;
}

/* ======================================================================================= host side */
static thread_local char g_err[512] = "";
static int g_device = -1;

/* first line of every entry point that takes a scene: the device is a property of the scene, not of the calling thread */
#define MI_ENTER(s, what) do { if(!(s)) return fail(MI_ERR_ARG, what); \
  if(hipSetDevice((s)->device) != hipSuccess) return fail(MI_ERR_DEVICE, "cannot select the scene's device"); } while(0)
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
  int device;                       /* the GPU this scene lives on: every entry point selects it (hipSetDevice is per host thread) */
  void *d_nodes, *d_axes, *d_prims, *d_primgeo, *d_materials, *d_light_prim, *d_light_cdf, *d_light_L;
  void *d_cie, *d_checker, *d_metal, *d_counters, *d_shape_material, *d_shape_L, *d_overflow;
  float *d_fb_own, *d_fb;
  float *h_stage;                   /* pinned staging buffer of mi_fb_read(accumulate), with the events of its two chunks in flight */
  hipEvent_t ev_stage[2];
  hipStream_t stream_own, stream;
  hipEvent_t ev0, ev1;
  int have_timing;
  size_t lds_bytes;
  bool nodes_lds;                   /* the whole BVH is staged in LDS (fits next to the stacks); else its first d.nodes_lds nodes are, the rest is read from HBM */
  bool nodes_t1;                    /* the node records carry the child boxes at shutter close */
  bool device_built;                /* tree made by mi_build.h */
  int stack_need;                   /* stack entries a ray may need */
  int grid;
  uint64_t launches;
  uint64_t kernel_launches_last;
  int counting;                     /* launch the COUNT instantiations (mi_scene_set_counters / CORONA_MI_COUNTERS) */
  int fast;                         /* launch the FAST instantiations (mi_scene_set_traversal / CORONA_MI_TRAVERSAL): same hits, other work counters */
  bool media;                       /* some shape is filled with a homogeneous medium: MEDIA instantiations */
  bool norg;                        /* ... of those, the ones without the exchange between waves (scattering exterior medium) */
  bool hero;                        /* launch the HERO instantiations: four wavelengths per path (mi_scene_set_wavelengths) */
  void *d_shape_medium, *d_prims_t1, *d_lights, *d_prim_cls;
  void *d_rng_jump;                 /* the generator's jump tables (rng_seed_jump, mi_kernels.h) */
  /* wavefront kernel (mi_wavefront.h): plain pt scenes; the workgroups' path tables */
  bool wavefront;
  void *d_wf_table;
  uint32_t wf_entries;
  /* Halton point sampler */
  bool halton;
  HaltonTables *halton_tables;
  uint64_t halton_epoch;            /* end index >> 32 the device tables were drawn for */
  void *d_halton_dim, *d_halton_perm;
};

static int upload_rng_jump(mi_scene *s);      /* (below, next to the launch that forms the constant) */

/* ---------------------------------------------------------------------------------------- kernel table
 * The megakernel's instantiations live in twenty-two parts (mi_megakernel.h, mi_part.hip), one translation unit each. */
extern template const void *mi_path_part<false, false, false, false>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  false, false, false>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  false, false>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  false, false>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  true,  false>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  true,  false>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  false, true,  true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  false, true,  true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, false, false, false, false, true>(unsigned, const PathLaunch *);     /* HERO (mi_hero.h): plain, media, motion blur, no exchange */
extern template const void *mi_path_part<true,  false, false, false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  false, false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  false, false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  true,  false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  true,  false, false, true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<false, true,  false, false, true,  true>(unsigned, const PathLaunch *);
extern template const void *mi_path_part<true,  true,  false, false, true,  true>(unsigned, const PathLaunch *);

extern template const void *mi_wave_part<false>(unsigned, const PathLaunch *);       /* the wavefront kernel (mi_wavefront.h, part 24) */

static const void *path_kernel(bool ptdl, bool media, bool mb, bool fast, bool norg, unsigned which, const PathLaunch *L, bool hero = false)
{
#ifdef MI_DEV_FAST
  if((media && MI_DEV_FAST != 3) || mb || hero) { fprintf(stderr, "[mi] internal: development build without the extended kernels\n"); abort(); }
#endif
  if(hero)
  { /* four wavelengths per path: the exact rounds of the scene's variant */
    if(mb) return ptdl ? mi_path_part<true, true, true, false, false, true>(which, L) : mi_path_part<false, true, true, false, false, true>(which, L);
    if(media && norg) return ptdl ? mi_path_part<true, true, false, false, true, true>(which, L) : mi_path_part<false, true, false, false, true, true>(which, L);
    if(media) return ptdl ? mi_path_part<true, true, false, false, false, true>(which, L) : mi_path_part<false, true, false, false, false, true>(which, L);
    return ptdl ? mi_path_part<true, false, false, false, false, true>(which, L) : mi_path_part<false, false, false, false, false, true>(which, L);
  }
  if(mb) return ptdl ? mi_path_part<true, true, true, false>(which, L) : mi_path_part<false, true, true, false>(which, L);   /* no FAST rounds with moving primitives */
  if(media)
  {
    if(norg)
    { /* a scene in a scattering exterior medium: the extended kernels without the exchange between waves */
      if(fast) return ptdl ? mi_path_part<true, true, false, true, true>(which, L) : mi_path_part<false, true, false, true, true>(which, L);
      return ptdl ? mi_path_part<true, true, false, false, true>(which, L) : mi_path_part<false, true, false, false, true>(which, L);
    }
    if(fast) return ptdl ? mi_path_part<true, true, false, true>(which, L) : mi_path_part<false, true, false, true>(which, L);
    return ptdl ? mi_path_part<true, true, false, false>(which, L) : mi_path_part<false, true, false, false>(which, L);
  }
  if(fast) return ptdl ? mi_path_part<true, false, false, true>(which, L) : mi_path_part<false, false, false, true>(which, L);
  return ptdl ? mi_path_part<true, false, false, false>(which, L) : mi_path_part<false, false, false, false>(which, L);
}

extern "C" const char *mi_last_error(void) { return g_err; }
extern "C" int mi_current_device(void) { return g_device; }

extern "C" int mi_init(int device)
{
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MI_ERR_DEVICE, "no HIP device visible");
  if(device < 0)
  { /* implicit choice: LOCAL_RANK of a one-process-per-GPU launcher -- unless the launcher already narrowed this process down to one
       visible GPU (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank), where LOCAL_RANK > 0 does not name a device */
    const char *lr = getenv("LOCAL_RANK");
    device = lr ? atoi(lr) : 0;
    if(device >= n && n == 1) device = 0;
  }
  if(device >= n) return fail(MI_ERR_ARG, "mi_init: no such device (one process per GPU: pass LOCAL_RANK, not a global rank)");
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

/* per-primitive constants of a line (truncated cone) primitive, computed once with the same float operations the
 * reference performs inside every intersection test (include/geo/line.h:313-335,401-416, include/corona_common.h:178-198);
 * layout documented at line_intersect (mi_kernels.h) */
static int tree_depth(const mi_scene_desc *h)
{ /* depth of the handed-over tree, or -1 if it is not one: a child link out of range, a node reached twice (cycle or shared
     subtree) or deeper than 200 levels. Every node is visited at most once, so this is O(num_nodes) whatever the links say. */
  std::vector<bool> seen(h->num_nodes, false);
  std::vector<std::pair<uint32_t, int>> todo;
  todo.push_back({0u, 0});
  seen[0] = true;
  int best = 0;
  while(!todo.empty())
  {
    const uint32_t node = todo.back().first;
    const int depth = todo.back().second;
    todo.pop_back();
    if(depth > best) best = depth;
    if(depth > 200) return -1;
    for(int c=0;c<4;c++)
    {
      const uint64_t ch = h->nodes[node].child[c];
      if(ch & MI_NODE_LEAF) continue;
      if(ch >= h->num_nodes || seen[ch]) return -1;
      seen[ch] = true;
      todo.push_back({(uint32_t)ch, depth + 1});
    }
  }
  return best;
}

/* ---------------------------------------------------------------------------------------- device build (mi_build.h) */
static int build_on_device(mi_scene *s, const mi_scene_desc *h, uint32_t *num_nodes, int *stack_need)
{
  const uint32_t n = (uint32_t)h->num_prims;
  BuildBufs b;
  memset(&b, 0, sizeof(b));
  b.n = n;
  b.leaf_max = s->d_prims_t1 ? MI_BUILD_LEAF : MI_BUILD_LEAF_JOBS;          /* by the kind of leaf phase the scene's kernels run (mi_build.h) */
  { const char *le = getenv("CORONA_MI_BUILD_LEAF"); if(le && atoi(le) >= 1 && atoi(le) <= 7) b.leaf_max = atoi(le); }
  hipEvent_t t0 = nullptr, t1 = nullptr;
  const bool verbose = getenv("CORONA_MI_VERBOSE") != nullptr;
  if(verbose) { (void)hipEventCreate(&t0); (void)hipEventCreate(&t1); }
  std::vector<void *> tmp;
  auto dev = [&](size_t bytes) -> void * { void *p = nullptr; if(hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr; tmp.push_back(p); return p; };
  auto release = [&]() { for(void *p : tmp) (void)hipFree(p); };
#define BALLOC(field, type, count) if(!(b.field = (type *)dev(sizeof(type)*(size_t)(count)))) { release(); return fail(MI_ERR_NOMEM, "device build: out of memory"); }
  BALLOC(box, float, 8*(size_t)n) BALLOC(key_in, uint32_t, n) BALLOC(key, uint32_t, n) BALLOC(val_in, uint32_t, n) BALLOC(perm, uint32_t, n)
  BALLOC(left, int, n) BALLOC(right, int, n) BALLOC(parent, int, n) BALLOC(leaf_parent, int, n) BALLOC(first, int, n) BALLOC(last, int, n)
  BALLOC(ibox, float, 8*(size_t)n) BALLOC(visits, unsigned int, n) BALLOC(count, int, n) BALLOC(perm2, uint32_t, n) BALLOC(cost, float, n)
  /* moving primitives: a second box set (shutter close), like the reference's aabb1; CORONA_MI_BUILD_T1=0 keeps one box around both states (A/B) */
  const bool two_states = s->d_prims_t1 != nullptr && !(getenv("CORONA_MI_BUILD_T1") && !atoi(getenv("CORONA_MI_BUILD_T1")));
  if(two_states) { BALLOC(box1, float, 8*(size_t)n) BALLOC(ibox1, float, 8*(size_t)n) }
#undef BALLOC
  if(verbose) (void)hipEventRecord(t0, 0);
  hipError_t e = hipMemset(b.visits, 0, sizeof(unsigned int)*n);
  const int grid = (int)((n + BL_BLOCK - 1)/BL_BLOCK);
  const float ext[3] = { h->aabb[3]-h->aabb[0], h->aabb[4]-h->aabb[1], h->aabb[5]-h->aabb[2] };
  const float3 slo = make_float3(h->aabb[0], h->aabb[1], h->aabb[2]);
  const float3 sinv = make_float3(ext[0] > 0 ? 1.0f/ext[0] : 0.0f, ext[1] > 0 ? 1.0f/ext[1] : 0.0f, ext[2] > 0 ? 1.0f/ext[2] : 0.0f);
  if(e == hipSuccess)
  {
    hipLaunchKernelGGL(bl_boxes, dim3(grid), dim3(BL_BLOCK), 0, 0, b, (const DPrim *)s->d_prims, (const DPrimGeo *)s->d_primgeo, (const DPrimT1 *)s->d_prims_t1, slo, sinv);
    e = hipGetLastError();
  }
  if(e == hipSuccess)
  { /* sort (Morton code, primitive) pairs */
    size_t bytes = 0;
    e = rocprim::radix_sort_pairs(nullptr, bytes, b.key_in, b.key, b.val_in, b.perm, (size_t)n, 0, 30, 0);
    void *scratch = e == hipSuccess ? dev(bytes) : nullptr;
    if(e == hipSuccess && !scratch) e = hipErrorOutOfMemory;
    if(e == hipSuccess) e = rocprim::radix_sort_pairs(scratch, bytes, b.key_in, b.key, b.val_in, b.perm, (size_t)n, 0, 30, 0);
  }
  if(e == hipSuccess)
  {
    hipLaunchKernelGGL(bl_hierarchy, dim3(grid), dim3(BL_BLOCK), 0, 0, b);
    hipLaunchKernelGGL(bl_refit, dim3(grid), dim3(BL_BLOCK), 0, 0, b);
    e = hipGetLastError();
  }
  unsigned int stats[4] = {0, 0, 0, 0};
  { /* SAH refinement: passes of tree rotations over the binary tree (mi_build.h); CORONA_MI_BUILD_SAH=0 keeps the plain LBVH (A/B) */
    int passes = 2;                 /* measured on regression/0010_pt and scenes/0059_mb (profiles/r04_devtree.txt): 1 / 2 / 3 (= converged) passes -> node visits 0.999 /
                                       0.957 / 0.951 and 1.050 / 1.016 / 1.025 x the reference's on its own tree, 64 spp in 15.48 / 15.53 / 15.91 and 25.65 / 25.57 / 26.33 ms */
    float ct = 1.0f;                /* cost of a primitive test in units of an inner (binary) node's box test: 2 and 4 buy fewer primitive tests with more node visits (no gain) */
    { const char *pe = getenv("CORONA_MI_BUILD_SAH"); if(pe && atoi(pe) >= 0 && atoi(pe) <= 16) passes = atoi(pe); }
    { const char *ce = getenv("CORONA_MI_BUILD_CT"); if(ce && atof(ce) > 0.0 && atof(ce) <= 64.0) ct = (float)atof(ce); }
    unsigned int *rot = passes ? (unsigned int *)dev(4) : nullptr;
    if(passes && !rot) e = hipErrorOutOfMemory;
    if(e == hipSuccess && rot) e = hipMemset(rot, 0, 4);
    for(int k=0;k<passes && e == hipSuccess && n > 2;k++)
    {
      e = hipMemset(b.visits, 0, sizeof(unsigned int)*n);
      if(e != hipSuccess) break;
      hipLaunchKernelGGL(bl_rotate, dim3(grid), dim3(BL_BLOCK), 0, 0, b, rot, ct);
      e = hipGetLastError();
    }
    if(e == hipSuccess && rot && verbose) e = hipMemcpy(&stats[1], rot, 4, hipMemcpyDeviceToHost);
    stats[2] = (unsigned int)passes;
  }
  uint32_t N = 0;
  if(e == hipSuccess)
  { /* top-down collapse, one launch per level of the 4-wide tree */
    const uint32_t cap = n - 1;
    float4 *tnodes = (float4 *)dev((size_t)MI_NODE_FIELDS*cap*16);
    float4 *tnodes_t1 = two_states ? (float4 *)dev((size_t)6*cap*16) : nullptr;
    if(two_states && !tnodes_t1) e = hipErrorOutOfMemory;
    uint32_t *taxes = (uint32_t *)dev((size_t)cap*4);
    int *la = (int *)dev(sizeof(int)*(size_t)n), *lb = (int *)dev(sizeof(int)*(size_t)n);
    unsigned int *qa = (unsigned int *)dev(sizeof(int)*(size_t)n), *qb = (unsigned int *)dev(sizeof(int)*(size_t)n);
    unsigned int *fa = (unsigned int *)dev(sizeof(int)*(size_t)n), *fb = (unsigned int *)dev(sizeof(int)*(size_t)n);
    unsigned int *cnt = (unsigned int *)dev(8);
    if(!tnodes || !taxes || !la || !lb || !qa || !qb || !fa || !fb || !cnt) e = hipErrorOutOfMemory;
    unsigned int hc[2] = {1, 0};
    const int zero = 0;
    if(e == hipSuccess) e = hipMemcpy(la, &zero, 4, hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(qa, &zero, 4, hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(fa, &zero, 4, hipMemcpyHostToDevice);
    unsigned int n_in = 1, levels = 0;
    while(e == hipSuccess && n_in)
    {
      hc[1] = 0;
      e = hipMemcpy(cnt, hc, 8, hipMemcpyHostToDevice);
      if(e != hipSuccess) break;
      CollapseLists L = { la, qa, fa, lb, qb, fb, cnt, n_in };
      hipLaunchKernelGGL(bl_collapse, dim3((n_in + BL_BLOCK - 1)/BL_BLOCK), dim3(BL_BLOCK), 0, 0, b, L, tnodes, taxes, cap, tnodes_t1);
      e = hipGetLastError();
      if(e == hipSuccess) e = hipMemcpy(hc, cnt, 8, hipMemcpyDeviceToHost);
      n_in = hc[1];
      std::swap(la, lb); std::swap(qa, qb); std::swap(fa, fb);
      levels++;
    }
    N = hc[0]; stats[0] = levels;
    if(e == hipSuccess && (!N || N >= MI_LEAF32 || N > cap)) e = hipErrorInvalidValue;
    const uint32_t SH = two_states ? 2u*MI_NODE_STRIDE : MI_NODE_STRIDE;       /* lanes of a node record (mi_device.h) */
    if(e == hipSuccess && (hipMalloc(&s->d_nodes, (size_t)SH*N*16) != hipSuccess || hipMalloc(&s->d_axes, (size_t)N*4) != hipSuccess)) e = hipErrorOutOfMemory;
    if(e == hipSuccess) e = hipMemset(s->d_nodes, 0, (size_t)SH*N*16);
    if(e == hipSuccess)
    {
      hipLaunchKernelGGL(bl_repack, dim3((MI_NODE_FIELDS*N + BL_BLOCK - 1)/BL_BLOCK), dim3(BL_BLOCK), 0, 0, (float4 *)s->d_nodes, (const float4 *)tnodes, N, cap, (uint32_t)MI_NODE_FIELDS, SH);
      e = hipGetLastError();
      if(e == hipSuccess && two_states)
      {
        hipLaunchKernelGGL(bl_repack, dim3((MI_NODE_T1_FIELDS*N + BL_BLOCK - 1)/BL_BLOCK), dim3(BL_BLOCK), 0, 0, (float4 *)s->d_nodes + MI_NODE_T1_HBM, (const float4 *)tnodes_t1, N, cap,
                           (uint32_t)MI_NODE_T1_FIELDS, SH);
        e = hipGetLastError();
        s->nodes_t1 = true;
      }
      if(e == hipSuccess) e = hipMemcpy(s->d_axes, taxes, (size_t)N*4, hipMemcpyDeviceToDevice);
    }
  }
  if(e == hipSuccess)
  { /* primitive records into sorted order; emitter indices follow */
    void *np = nullptr, *ng = nullptr, *nt = nullptr;
    uint32_t *inv = (uint32_t *)dev(sizeof(uint32_t)*(size_t)n);
    if(!inv || hipMalloc(&np, sizeof(DPrim)*(size_t)n) != hipSuccess || hipMalloc(&ng, sizeof(DPrimGeo)*(size_t)n) != hipSuccess) e = hipErrorOutOfMemory;
    if(e == hipSuccess && s->d_prims_t1 && hipMalloc(&nt, sizeof(DPrimT1)*(size_t)n) != hipSuccess) e = hipErrorOutOfMemory;
    if(e == hipSuccess)
    {
      if(nt) hipLaunchKernelGGL(bl_gather<DPrimT1>, dim3(grid), dim3(BL_BLOCK), 0, 0, (DPrimT1 *)nt, (const DPrimT1 *)s->d_prims_t1, (const uint32_t *)b.perm2, n);
      hipLaunchKernelGGL(bl_gather<DPrim>, dim3(grid), dim3(BL_BLOCK), 0, 0, (DPrim *)np, (const DPrim *)s->d_prims, (const uint32_t *)b.perm2, n);
      hipLaunchKernelGGL(bl_gather<DPrimGeo>, dim3(grid), dim3(BL_BLOCK), 0, 0, (DPrimGeo *)ng, (const DPrimGeo *)s->d_primgeo, (const uint32_t *)b.perm2, n);
      hipLaunchKernelGGL(bl_invert, dim3(grid), dim3(BL_BLOCK), 0, 0, inv, (const uint32_t *)b.perm2, n);
      if(h->lights.num_prims)
        hipLaunchKernelGGL(bl_remap, dim3((h->lights.num_prims + BL_BLOCK - 1)/BL_BLOCK), dim3(BL_BLOCK), 0, 0, (uint32_t *)s->d_light_prim, (const uint32_t *)inv,
                           h->lights.num_prims);
      e = hipGetLastError();
      if(e == hipSuccess) e = hipDeviceSynchronize();
      if(e == hipSuccess)
      {
        (void)hipFree(s->d_prims); (void)hipFree(s->d_primgeo); s->d_prims = np; s->d_primgeo = ng; np = ng = nullptr;
        if(nt) { (void)hipFree(s->d_prims_t1); s->d_prims_t1 = nt; nt = nullptr; }
      }
    }
    if(np) (void)hipFree(np);
    if(ng) (void)hipFree(ng);
    if(nt) (void)hipFree(nt);
  }
  if(verbose) (void)hipEventRecord(t1, 0);
  if(e == hipSuccess) e = hipDeviceSynchronize();
  if(verbose)
  {
    float ms = 0.0f;
    if(e == hipSuccess && hipEventElapsedTime(&ms, t0, t1) == hipSuccess)
      fprintf(stderr, "[mi] device build: %u primitives -> %u 4-wide nodes, %u levels, %u rotations in %u passes, %.3f ms on the device\n", n, N, stats[0], stats[1], stats[2], ms);
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
  }
  release();
  if(e != hipSuccess) { snprintf(g_err, sizeof(g_err), "device build: %s", hipGetErrorString(e)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  *num_nodes = N;
  *stack_need = 3*((int)stats[0] + 1);
  return MI_OK;
}

static int scene_create_on(const mi_scene_desc *h, int device, mi_scene **out);

extern "C" int mi_scene_create(const mi_scene_desc *h, mi_scene **out)
{ /* on the device mi_init chose for this process (one process per GPU); mi_group_create places scenes on devices by index */
  if(g_device < 0) { const int e = mi_init(-1); if(e) return e; }
  return scene_create_on(h, g_device, out);
}

static int scene_create_on(const mi_scene_desc *h, int device, mi_scene **out)
{
  if(!h || !out) return fail(MI_ERR_ARG, "mi_scene_create: null argument");
  if(h->struct_size != sizeof(mi_scene_desc) || h->abi_version != MI_ABI_VERSION)
    return fail(MI_ERR_ARG, "mi_scene_create: mi_scene_desc size/version mismatch");
  if(!h->width || !h->height || (h->width & 31) || (h->height & 31)) return fail(MI_ERR_ARG, "film size must be a non-zero multiple of 32");
  if(h->max_verts < 2 || h->max_verts > 32) return fail(MI_ERR_ARG, "max_verts must be in [2,32]");
  if(h->sampler != MI_SAMPLER_PT && h->sampler != MI_SAMPLER_PTDL) return fail(MI_ERR_ARG, "unknown sampler");
  if(h->pointsampler != MI_POINTS_RAND && h->pointsampler != MI_POINTS_HALTON) return fail(MI_ERR_ARG, "unknown point sampler");
  if(!h->cie_xyz) return fail(MI_ERR_ARG, "scene has no tables");
  const bool device_build = !h->nodes;                      /* no tree handed over: build it on the device (mi_build.h) */
  if(!device_build && !h->num_nodes) return fail(MI_ERR_ARG, "scene has no nodes");
  if(device_build && h->num_prims < 2) return fail(MI_ERR_UNSUPPORTED, "the device build needs at least two primitives");
  if(h->num_shapes > 255) return fail(MI_ERR_UNSUPPORTED, "more than 255 shapes");
  if(h->num_prims >= (1u << 26)) return fail(MI_ERR_UNSUPPORTED, "more than 2^26 primitives");

  const int depth = device_build ? 0 : tree_depth(h);
  if(depth < 0) return fail(MI_ERR_ARG, "the BVH is not a tree of at most 200 levels (child link out of range, node reached twice, or too deep)");
  /* every shape's material is used below (per-primitive records, per-shape tables), also for shapes without primitives */
  for(uint32_t i=0;i<h->num_shapes;i++)
    if(h->shapes[i].material < 0 || (uint32_t)h->shapes[i].material >= h->num_materials)
      return fail(MI_ERR_ARG, "a shape refers to a material outside the descriptor's material list");
  int stack_need = 3*(depth+1);                             /* at most 3 pushes per inner node on the way down */

  mi_scene *s = (mi_scene *)calloc(1, sizeof(mi_scene));
  if(!s) return fail(MI_ERR_NOMEM, "out of host memory");
  s->width = h->width; s->height = h->height;
  s->device = device;
  if(hipSetDevice(s->device) != hipSuccess) { free(s); return fail(MI_ERR_DEVICE, "cannot select the device"); }   /* mi_init may have run on another thread */
  DScene &d = s->d;
  d.width = h->width; d.height = h->height; d.max_verts = h->max_verts; d.sampler = h->sampler; d.frame = h->frame;
  d.num_nodes = h->num_nodes; d.num_prims = (uint32_t)h->num_prims;
  memcpy(d.aabb, h->aabb, sizeof(d.aabb));
  {
    const float ex = h->aabb[3]-h->aabb[0], ey = h->aabb[4]-h->aabb[1], ez = h->aabb[5]-h->aabb[2];
    const float m1 = ey > ex ? ey : ex;
    d.far_dist = 2.0f*(ez > m1 ? ez : m1);
  }

  /* nodes: one record of SH 16-byte lanes per node (mi_device.h), children as 32-bit links -- node indices here, record offsets once the
     scene knows how many nodes it stages in LDS (mi_bake_links_kernel). Nodes are renumbered breadth first from the root, so that the
     first K nodes are the top of the tree whatever order the caller's builder numbered them in (the reference's: depth first). */
  uint32_t N = device_build ? 0 : h->num_nodes;
  const bool host_t1 = !device_build && h->nodes_t1;
  const uint32_t SH_host = host_t1 ? 2u*MI_NODE_STRIDE : MI_NODE_STRIDE;
  std::vector<float> nodes((size_t)SH_host*N*4, 0.0f);
  std::vector<uint32_t> axes(N);
  std::vector<bool> in_leaf(device_build ? 0 : (size_t)h->num_prims, false);
  std::vector<uint32_t> bfs, newid(N, 0xffffffffu);        /* bfs[new] = old, newid[old] = new */
  if(N)
  { /* (tree_depth above has checked the links: every node is reachable at most once) */
    bfs.reserve(N);
    bfs.push_back(0); newid[0] = 0;
    for(size_t q=0;q<bfs.size();q++)
      for(int c=0;c<4;c++)
      {
        const uint64_t ch = h->nodes[bfs[q]].child[c];
        if(!(ch & MI_NODE_LEAF) && ch < N && newid[ch] == 0xffffffffu) { newid[ch] = (uint32_t)bfs.size(); bfs.push_back((uint32_t)ch); }
      }
    for(uint32_t n=0;n<N;n++) if(newid[n] == 0xffffffffu) { newid[n] = (uint32_t)bfs.size(); bfs.push_back(n); }     /* unreachable nodes keep a place behind the tree */
  }
  for(uint32_t nn=0;nn<N;nn++)
  {
    const uint32_t n = bfs[nn];
    const mi_node &nd = h->nodes[n];
    float *rec = &nodes[(size_t)nn*SH_host*4];
    /* lanes 0..2 hold the lower, 3..5 the upper plane per axis. The reference marks empty children by an inverted box
       (min = FLT_MAX, max = -FLT_MAX, qbvhmp.c:1095-1112) and evaluates min(t0,t1)/max(t0,t1), which is symmetric in the two
       planes; the kernel picks entry/exit planes by ray sign instead, so inverted slabs are stored in ascending order */
    for(int k=0;k<3;k++) for(int c=0;c<4;c++)
    {
      const float b0 = nd.aabb[k][c], b1 = nd.aabb[k+3][c];
      rec[k*4 + c]     = b0 > b1 ? b1 : b0;
      rec[(k+3)*4 + c] = b0 > b1 ? b0 : b1;
    }
    for(int c=0;c<4;c++)
    {
      uint32_t link;
      if(nd.child[c] & MI_NODE_LEAF)
      {
        const uint64_t first = (nd.child[c] ^ MI_NODE_LEAF) >> 5, cntp = nd.child[c] & 31;
        if(first + cntp > h->num_prims) { free(s); return fail(MI_ERR_ARG, "a leaf of the tree points outside the primitive list"); }
        for(uint64_t q=first;q<first+cntp;q++)
        { /* leaves are disjoint ranges of the primitive list: the leaf-order pass (mi_mark_ordered_kernel) rewrites records leaf by leaf */
          if(in_leaf[q]) { free(s); return fail(MI_ERR_ARG, "two leaves of the tree share a primitive"); }
          in_leaf[q] = true;
        }
        link = MI_LEAF32 | (uint32_t)(first << 5) | (uint32_t)cntp;
      }
      else if(nd.child[c] < (uint64_t)N) link = newid[(uint32_t)nd.child[c]];
      else link = MI_LEAF32;          /* a node no ray reaches (tree_depth has checked every reachable link) may hold anything: an empty leaf instead of an index beyond the table */
      memcpy(&rec[6*4 + c], &link, 4);
    }
    axes[nn] = (uint32_t)(nd.axis0 & 3) | ((uint32_t)(nd.axis00 & 3) << 2) | ((uint32_t)(nd.axis01 & 3) << 4);
    /* the shutter-close boxes, if the caller's tree carries them: lanes 8..13, same treatment of empty children */
    if(host_t1) for(int k=0;k<3;k++) for(int c=0;c<4;c++)
    {
      const float b0 = h->nodes_t1[n].aabb[k][c], b1 = h->nodes_t1[n].aabb[k+3][c];
      rec[(MI_NODE_T1_HBM + k)*4 + c]     = b0 > b1 ? b1 : b0;
      rec[(MI_NODE_T1_HBM + k+3)*4 + c] = b0 > b1 ? b0 : b1;
    }
  }
  s->nodes_t1 = host_t1;
  /* primitives: resolve primid -> vtxidx -> vtx once */
  std::vector<DPrim> prims(h->num_prims ? h->num_prims : 1);
  std::vector<DPrimGeo> pgeo(h->num_prims ? h->num_prims : 1);
  std::vector<DPrimT1> prims_t1;                            /* allocated when the first motion-blurred primitive shows up */
  memset(pgeo.data(), 0, pgeo.size()*sizeof(DPrimGeo));
  memset(prims.data(), 0, prims.size()*sizeof(DPrim));
  /* material queues (mi_regroup.h): the class of a primitive is the compact index of its material's bsdf among the surface bsdfs the scene uses */
  uint32_t bsdf_class[MI_BSDF_METAL + 1] = { 0u, 0u, 0u }, num_classes = 0;
  {
    bool used[MI_BSDF_METAL + 1] = { false, false, false };
    for(uint32_t i=0;i<h->num_shapes;i++)
      if((uint32_t)h->shapes[i].material < h->num_materials && h->materials[h->shapes[i].material].bsdf <= MI_BSDF_METAL) used[h->materials[h->shapes[i].material].bsdf] = true;
    for(uint32_t b=0;b<=MI_BSDF_METAL;b++) if(used[b]) bsdf_class[b] = num_classes++;
  }
  for(uint64_t i=0;i<h->num_prims;i++)
  {
    const mi_primid pi = h->primid[i];
    const uint32_t shape = MI_PRIMID_SHAPE(pi), vc = MI_PRIMID_VCNT(pi);
    const uint32_t mb = MI_PRIMID_MB(pi);
    if(shape >= h->num_shapes || vc < 1 || vc > 4) { free(s); return fail(MI_ERR_UNSUPPORTED, "primitive kind outside the scope"); }
    const mi_shape &sh = h->shapes[shape];
    const mi_vtxidx *vi = h->vtxidx + sh.vtxidx_base + MI_PRIMID_VI(pi);
    const mi_vtx *vtx = h->vtx + sh.vtx_base;
    /* the descriptor is the caller's: every index this primitive uses must lie inside the arrays handed over */
    bool in_range = (uint64_t)sh.vtxidx_base + MI_PRIMID_VI(pi) + vc <= h->num_vtxidx;
    for(uint32_t k=0;in_range && k<vc;k++) in_range = (uint64_t)sh.vtx_base + (uint64_t)(mb + 1)*vi[k].v + mb < h->num_vtx;
    if(!in_range) { free(s); return fail(MI_ERR_ARG, "primitive refers to vertices outside the arrays of the descriptor"); }
    DPrim &p = prims[i]; DPrimGeo &q = pgeo[i];
    p.type = vc;
    q.type = vc; q.material = (uint32_t)sh.material; q.uv0 = vi[0].uv;
    q.primid_lo = (uint32_t)pi; q.primid_hi = (uint32_t)(pi >> 32);
    if((uint32_t)sh.material >= h->num_materials || h->materials[sh.material].bsdf > MI_BSDF_METAL)
    { free(s); return fail(MI_ERR_UNSUPPORTED, "shape uses a material outside the scope"); }
    q.cls = bsdf_class[h->materials[sh.material].bsdf];
    if(vc == MI_PRIM_LINE) { pgeo[i].f[18] = (vi[0].uv >> 21)/2048.0f; pgeo[i].f[19] = ((vi[0].uv & 0x1ffc00u) >> 10)/2048.0f; }
    else for(uint32_t k=0;k<vc;k++) { pgeo[i].f[18+2*k] = half2float(vi[k].uv & 0xffffu); pgeo[i].f[19+2*k] = half2float(vi[k].uv >> 16); }
    if(mb && vc < MI_PRIM_TRI)
    { /* moving sphere / line: DPrim type 0 keeps the shutter-open positions and radii, DPrimT1 the shutter-close positions; the
         packed record and the shading constants are formed per ray / per hit at the path's time (moving_analytic_at) */
      if(prims_t1.empty()) { prims_t1.resize(h->num_prims); memset(prims_t1.data(), 0, prims_t1.size()*sizeof(DPrimT1)); }
      DPrimT1 &t1 = prims_t1[i];
      p.type = 0; p.pad[0] = vc;
      q.type = vc | MI_GEO_MB;
      for(uint32_t k=0;k<vc;k++)
      {
        memcpy(p.v[k], vtx[2*vi[k].v].v, 12);
        memcpy(t1.v[k], vtx[2*vi[k].v + 1].v, 12);
        memcpy(&p.v[2][k], &vtx[2*vi[k].v].n, 4);              /* radii: those of the shutter-open vertices */
      }
    }
    else if(vc == MI_PRIM_SPHERE)
    {
      memcpy(p.v[0], vtx[vi[0].v].v, 12);
      memcpy(&p.v[1][0], &vtx[vi[0].v].n, 4);
      memcpy(&q.f[29], vtx[vi[0].v].v, 12);           /* centre and radius for the shading side */
      memcpy(&q.f[32], &vtx[vi[0].v].n, 4);
    }
    else if(vc == MI_PRIM_LINE)
    {
      const V3 v0 = ld3(vtx[vi[0].v].v), v1 = ld3(vtx[vi[1].v].v);
      float r0, r1;
      memcpy(&r0, &vtx[vi[0].v].n, 4); memcpy(&r1, &vtx[vi[1].v].n, 4);
      pack_line(p, v0, v1, r0, r1);
      /* shading-side frame of the line (line.h:123-161), with the functions the kernel would run per vertex */
      line_shading_consts(pgeo[i].f, p, v0, v1);
    }
    else if(mb)
    { /* motion-blurred triangle / quad (vertices interleaved: 2 v = shutter open, 2 v + 1 = shutter close, include/geo.h:108-138):
         DPrim type 0 holds the shutter-open vertices, DPrimT1 the shutter-close vertices and normals; everything else is
         formed per ray / per hit at the path's time (analytic_intersect<true>, surface_setup<true>) */
      float *g = pgeo[i].f;
      if(prims_t1.empty()) { prims_t1.resize(h->num_prims); memset(prims_t1.data(), 0, prims_t1.size()*sizeof(DPrimT1)); }
      DPrimT1 &t1 = prims_t1[i];
      p.type = 0; p.pad[0] = vc;
      q.type = vc | MI_GEO_MB;
      for(uint32_t k=0;k<vc;k++)
      {
        const mi_vtx &a = vtx[2*vi[k].v], &b = vtx[2*vi[k].v + 1];
        memcpy(p.v[k], a.v, 12);
        memcpy(t1.v[k], b.v, 12);
        const V3 n0 = decode_normal(a.n), n1 = decode_normal(b.n);
        g[3*k] = n0.x; g[3*k+1] = n0.y; g[3*k+2] = n0.z;
        t1.n[k][0] = n1.x; t1.n[k][1] = n1.y; t1.n[k][2] = n1.z;
      }
    }
    else
    {
      float *g = pgeo[i].f;
      V3 vv[4] = {};
      for(uint32_t k=0;k<vc;k++) vv[k] = ld3(vtx[vi[k].v].v);
      memcpy(p.v[0], vtx[vi[0].v].v, 12);
      for(uint32_t k=1;k<vc;k++)
      { /* the intersection test works on edges (triangle.h:271-283): the same float subtraction, once */
        const V3 e = sub3(vv[k], vv[0]);
        p.v[k][0] = e.x; p.v[k][1] = e.y; p.v[k][2] = e.z;
        g[26 + 3*(k-1)] = vv[k].x; g[27 + 3*(k-1)] = vv[k].y; g[28 + 3*(k-1)] = vv[k].z;
      }
      for(uint32_t k=0;k<vc;k++) { const V3 n = decode_normal(vtx[vi[k].v].n); g[3*k] = n.x; g[3*k+1] = n.y; g[3*k+2] = n.z; }
      const V3 ga = tri_geo_normal(vv[0], vv[1], vv[2]);
      g[12] = ga.x; g[13] = ga.y; g[14] = ga.z;
      if(vc == MI_PRIM_QUAD)
      {
        const V3 gb = tri_geo_normal(vv[0], vv[2], vv[3]);
        g[15] = gb.x; g[16] = gb.y; g[17] = gb.z;
      }
    }
  }
  std::vector<DMaterial> mats(h->num_materials ? h->num_materials : 1);
  for(uint32_t i=0;i<h->num_materials;i++)
  {
    mats[i].bsdf = h->materials[i].bsdf; mats[i].num_ops = h->materials[i].num_ops;
    memcpy(mats[i].op, h->materials[i].op, sizeof(mats[i].op));
    memcpy(mats[i].param, h->materials[i].param, sizeof(mats[i].param));
  }
  std::vector<uint32_t> shape_mat(4*(size_t)(h->num_shapes ? h->num_shapes : 1), 0u);   /* per shape: bsdf, material, param[0], param[1] */
  for(uint32_t i=0;i<h->num_shapes;i++)
  {
    const mi_material &m = h->materials[h->shapes[i].material];
    shape_mat[4*i+0] = m.bsdf; shape_mat[4*i+1] = (uint32_t)h->shapes[i].material;
    memcpy(&shape_mat[4*i+2], &m.param[0], 4); memcpy(&shape_mat[4*i+3], &m.param[1], 4);
  }
  /* homogeneous media: per shape the interior medium of its material (`interior <surface> <medium>`) */
  std::vector<DShapeMedium> shape_med((size_t)h->num_shapes + 1);
  bool any_media = false;
  auto medium_entry = [&](DShapeMedium &e, uint32_t id) -> const char *
  { /* the medium's prepare chain: exactly one colour op in the volume slot (the albedo), then medium_rgb */
    if(id >= h->num_materials || h->materials[id].bsdf != MI_BSDF_MEDIUM) return "not a medium material";
    const mi_material &med = h->materials[id];
    if(med.num_ops != 1 || med.op[0].kind != MI_OP_COLOR || med.op[0].slot != MI_SLOT_VOLUME)
      return "a medium needs exactly one `color v` (albedo) in front of medium_rgb";
    memcpy(e.albedo, med.op[0].coeff, 12); e.albedo[3] = med.op[0].mul;
    memcpy(e.mu_t, med.param, 16);
    e.g = med.mean_cos; e.med = (int32_t)id;
    return nullptr;
  };
  {
    DShapeMedium &e = shape_med[h->num_shapes];
    memset(&e, 0, sizeof(e));
    e.med = -1;
    if(h->exterior)
    {
      const char *why = medium_entry(e, h->exterior - 1);
      if(why) { free(s); return fail(MI_ERR_ARG, why); }
      any_media = true;
    }
  }
  for(uint32_t i=0;i<h->num_shapes;i++)
  {
    DShapeMedium &e = shape_med[i];
    memset(&e, 0, sizeof(e));
    e.med = -1;
    const mi_material &m = h->materials[h->shapes[i].material];
    if(m.bsdf == MI_BSDF_MEDIUM) { free(s); return fail(MI_ERR_UNSUPPORTED, "a medium can only be the interior of a surface material"); }
    if(m.interior < 0) continue;
    const char *why = medium_entry(e, (uint32_t)m.interior);
    if(why) { free(s); return fail(MI_ERR_ARG, why); }
    any_media = true;
  }
  std::vector<float> shape_L(h->num_shapes ? h->num_shapes : 1, 0.0f);
  for(uint32_t k=0;k<h->lights.num_prims;k++)
  {
    const uint32_t sid = MI_PRIMID_SHAPE(h->lights.primid[k]);
    if(sid < h->num_shapes && shape_L[sid] == 0.0f) shape_L[sid] = h->lights.L[k];   /* lights_pdf_next_event: L of the shape */
  }
  /* emitters: original primid -> builder-order index (one hash map over the primitive list: O(P + L), not O(P L)) */
  std::vector<uint32_t> lprim(h->lights.num_prims ? h->lights.num_prims : 1);
  if(h->lights.num_prims)
  {
    std::unordered_map<uint64_t, uint32_t> where;
    where.reserve((size_t)h->num_prims*2);
    for(uint64_t i=0;i<h->num_prims;i++) where.emplace((uint64_t)h->primid[i], (uint32_t)i);      /* the first occurrence wins, like the scan it replaces */
    for(uint32_t k=0;k<h->lights.num_prims;k++)
    {
      const auto it = where.find((uint64_t)h->lights.primid[k]);
      if(it == where.end()) { free(s); return fail(MI_ERR_ARG, "emitter primitive not in the primitive list"); }
      lprim[k] = it->second;
    }
  }

  /* any-hit shadow rays (MI_LIGHT_ANYHIT, mi_device.h): emitter primitives a connection ray cannot cross inside the connection */
  std::vector<uint32_t> lflag(lprim.size(), 0u);
  {
    const char *sm = getenv("CORONA_MI_SHADOW");          /* "closest": the reference's traversal for every shadow ray (counter parity) */
    const bool closest = sm && !strcmp(sm, "closest");
    for(uint32_t k=0;k<h->lights.num_prims && !closest;k++)
    {
      const DPrim &p = prims[lprim[k]];
      if(p.type == MI_PRIM_TRI) lflag[k] = MI_LIGHT_ANYHIT;
      else if(p.type == MI_PRIM_QUAD)
      { /* planar: v3 - v0 has no component along the normal of (v0 v1 v2), to float accuracy */
        const V3 e1 = ld3(p.v[1]), e2 = ld3(p.v[2]), e3 = ld3(p.v[3]);
        const V3 n = cross3(e1, e2);
        const float off = fabsf(dot3(n, e3)), scale = sqrtf(dot3(n, n))*sqrtf(dot3(e3, e3));
        if(off <= 1e-5f*scale) lflag[k] = MI_LIGHT_ANYHIT;
      }
    }
  }
  int e = MI_OK;
#define UP(dst, src, cnt) if(!e) e = upload(&s->dst, src, cnt)
  if(!device_build)
  {
    UP(d_nodes, nodes.data(), nodes.size());
    UP(d_axes, axes.data(), axes.size());
  }
  UP(d_prims, prims.data(), prims.size());
  UP(d_primgeo, pgeo.data(), pgeo.size());
  UP(d_materials, mats.data(), mats.size());
  UP(d_shape_material, shape_mat.data(), shape_mat.size());
  UP(d_shape_L, shape_L.data(), shape_L.size());
  /* the MEDIA instantiations are the "extended" kernels: participating media and/or a moving camera (mi_path.h, path_generate) */
  if(!prims_t1.empty())
  {
    UP(d_prims_t1, prims_t1.data(), prims_t1.size());
  }
  if(any_media || h->cam.moving || !prims_t1.empty()) s->media = true;
  UP(d_shape_medium, shape_med.data(), shape_med.size());        /* small; the extended kernels read it (all-vacuum entries for scenes without media) */
  UP(d_light_prim, lprim.data(), lprim.size());
  UP(d_light_cdf, h->lights.cdf, (size_t)h->lights.num_prims);
  UP(d_light_L, h->lights.L, (size_t)h->lights.num_prims);
  UP(d_cie, h->cie_xyz, (size_t)96*3);
  UP(d_checker, h->checker, h->checker ? (size_t)140*36 : 0);
  UP(d_metal, h->metal_ior, h->metal_ior ? (size_t)5*95*2 : 0);
  UP(d_counters, (const unsigned long long *)nullptr, (size_t)8*MI_COUNTER_SHARDS);
#undef UP
  if(!e && device_build)
  {
    e = build_on_device(s, h, &N, &stack_need);
    d.num_nodes = N;
  }
  /* (a link holds a record offset in 24 bits, mi_device.h: 2^24 lanes of 16 bytes) */
  if(!e && (uint64_t)N*(s->nodes_t1 ? 2u*MI_NODE_STRIDE : MI_NODE_STRIDE) >= (1ull << MI_AXES_SHIFT)) e = fail(MI_ERR_UNSUPPORTED, "more nodes than a link can address (2^24 lanes of 16 bytes)");
  if(!e && N)
  { /* tree and primitive records are in their final order: leaves in which the order of the tests matters (a folded quad behind a
       primitive the leaf loops put off) are marked, see mi_mark_ordered_kernel */
    hipLaunchKernelGGL(mi_mark_ordered_kernel, dim3((4*N + 255)/256), dim3(256), 0, 0, (const float4 *)s->d_nodes, N, (uint32_t)(s->nodes_t1 ? 2u*MI_NODE_STRIDE : MI_NODE_STRIDE), (DPrim *)s->d_prims);
    if(hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) e = fail(MI_ERR_DEVICE, "cannot mark the leaves");
  }
  if(!e && h->num_prims)
  { /* ... and the primitives' classes for the exchange between waves, packed (mi_pack_cls_kernel) */
    const uint32_t words = (uint32_t)((h->num_prims + 15)/16);
    if(hipMalloc(&s->d_prim_cls, (size_t)words*4) != hipSuccess) e = fail(MI_ERR_NOMEM, "cannot allocate the class table");
    else
    {
      hipLaunchKernelGGL(mi_pack_cls_kernel, dim3((words + 255)/256), dim3(256), 0, 0, (const DPrimGeo *)s->d_primgeo, (uint32_t)h->num_prims, (uint32_t *)s->d_prim_cls);
      if(hipGetLastError() != hipSuccess) e = fail(MI_ERR_DEVICE, "cannot pack the class table");
    }
  }
  if(!e && h->lights.num_prims)
  { /* the list is in its final (builder or device-sorted) order now: set the any-hit flags */
    std::vector<uint32_t> cur(h->lights.num_prims);
    if(hipMemcpy(cur.data(), s->d_light_prim, cur.size()*4, hipMemcpyDeviceToHost) != hipSuccess) e = fail(MI_ERR_DEVICE, "cannot read back the emitter list");
    for(size_t k=0;k<cur.size();k++) cur[k] |= lflag[k];
    if(!e && hipMemcpy(s->d_light_prim, cur.data(), cur.size()*4, hipMemcpyHostToDevice) != hipSuccess) e = fail(MI_ERR_DEVICE, "cannot update the emitter list");
    /* one-burst emitter records for next event estimation (DLight, mi_device.h): possible when every emitter primitive is a static
       triangle / quad and its material's prepare chain consists of plain colours; any other scene runs the generic branch, which
       only the extended kernels carry */
    std::vector<DLight> lights(h->lights.num_prims);
    bool fast = true;
    for(uint32_t k=0;k<h->lights.num_prims && fast;k++)
    {
      const DPrim &p = prims[lprim[k]];
      const DPrimGeo &g = pgeo[lprim[k]];
      const mi_material &m = h->materials[g.material];
      if(p.type != MI_PRIM_TRI && p.type != MI_PRIM_QUAD) { fast = false; break; }
      DLight &L = lights[k];
      memset(&L, 0, sizeof(L));
      memcpy(L.v[0], p.v[0], 12);
      memcpy(L.v[1], &g.f[26], 36);                    /* v1, v2, v3 */
      memcpy(L.n[0], &g.f[0], 48);
      memcpy(L.gn[0], &g.f[12], 24);
      L.roughness = 1.0f;                               /* run_prepare_ops' initial state */
      for(uint32_t o=0;o<m.num_ops;o++)
      {
        if(m.op[o].kind != MI_OP_COLOR) { fast = false; break; }
        L.roughness = m.op[o].roughness;
        if(m.op[o].slot == MI_SLOT_EMISSION) { memcpy(L.em_coeff, m.op[o].coeff, 12); L.em_mul = m.op[o].mul; }
      }
      L.L = h->lights.L[k];
      L.prim = cur[k];
      L.type = p.type;
    }
    if(h->sampler == MI_SAMPLER_PTDL)
    { /* only next event estimation reads emitter records: a pt scene stays with the plain kernels whatever its emitters are */
      if(fast) { if(!e) e = upload(&s->d_lights, lights.data(), lights.size()); }
      else s->media = true;                             /* the extended kernels keep the generic emitter code */
    }
  }
  if(!e)
  {
    void *fb = nullptr;
    if(hipMalloc(&fb, sizeof(float)*3*(size_t)h->width*h->height) != hipSuccess) e = fail(MI_ERR_NOMEM, "cannot allocate the device framebuffer");
    else if(hipMemset(fb, 0, sizeof(float)*3*(size_t)h->width*h->height) != hipSuccess) { (void)hipFree(fb); e = fail(MI_ERR_DEVICE, "cannot clear the device framebuffer"); }
    else s->d_fb_own = (float *)fb;
  }
  if(e) { mi_scene_destroy(s); return e; }
  s->d_fb = s->d_fb_own;
  if(hipStreamCreateWithFlags(&s->stream_own, hipStreamNonBlocking) != hipSuccess ||
     hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess)
  { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "cannot create stream/events"); }
  s->stream = s->stream_own;

  d.nodes = (const float4 *)s->d_nodes; d.nodes_t1 = s->nodes_t1 ? 1u : 0u;
  d.prims = (const DPrim *)s->d_prims; d.primgeo = (const DPrimGeo *)s->d_primgeo; d.prim_cls = (const uint32_t *)s->d_prim_cls;
  d.materials = (const DMaterial *)s->d_materials;
  d.shape_medium = (const DShapeMedium *)s->d_shape_medium;
  d.prims_t1 = (const DPrimT1 *)s->d_prims_t1;
  d.exterior_index = h->num_shapes;
  d.num_lights = h->lights.num_prims;
  d.light_prim = (const uint32_t *)s->d_light_prim; d.light_cdf = (const float *)s->d_light_cdf; d.light_L = (const float *)s->d_light_L;
  d.lights = (const DLight *)s->d_lights;
  for(uint32_t k=0;k<4;k++) d.light_cdf4[k] = k < h->lights.num_prims ? h->lights.cdf[k] : 1.0f;
  d.p_sky = h->lights.p_sky; d.p_geo = h->lights.p_geo; d.p_vol = h->lights.p_vol;
  d.cam = h->cam;
  { /* same expressions, same order as path_generate evaluated them per path before (thinlens.c:68-128) */
    const mi_camera &cam = h->cam;
    DCamConst &c = d.cc;
    c.W = (float)h->width; c.H = (float)h->height;
    c.lens_radius = (.5f/cam.f_stop)*cam.focal_length;
    const float f = cam.focus/cam.focal_length;
    c.f_dir = cam.focus;
    c.f_rg = -cam.film_width*f/c.W;
    c.f_up = -cam.film_height*f/c.H;
    const float A = (float)(MI_PI_D*(double)cam.focal_length*(double)cam.focal_length/(double)(4.0f*cam.f_stop*cam.f_stop));
    c.pdf_a = (float)(1./(double)A);
    c.sensor = 106.86535f*100.0f*cam.exposure_time;
    c.fl2 = cam.focal_length*cam.focal_length;
    c.pdf_v = 1.0f/(cam.film_width*cam.film_height);
    c.pdf_av = c.pdf_a*c.pdf_v;
    c.Wc = c.W-1e-4f; c.Hc = c.H-1e-4f;
  }
  d.cie_xyz = (const float *)s->d_cie; d.checker = (const float *)s->d_checker; d.metal_ior = (const float *)s->d_metal;
  d.fb = s->d_fb;
  d.counters = (unsigned long long *)s->d_counters;

  /* a scene with moving primitives runs the motion-blur kernels: shallower stack columns, and both box sets of the nodes in LDS */
  const bool mb_kernels = s->d_prims_t1 != nullptr;
  /* lanes of a node record in LDS / in HBM: the links are baked for them (mi_device.h) */
  const uint32_t SL = s->nodes_t1 ? MI_NODE_FIELDS + MI_NODE_T1_FIELDS : MI_NODE_FIELDS, SH = s->nodes_t1 ? 2u*MI_NODE_STRIDE : MI_NODE_STRIDE;
  const size_t node_bytes = (size_t)SL*N*16;
  /* (the column of the plain kernels is shorter since round 4: they use the LDS for the pools of mi_regroup.h; s->media is final here) */
  /* a scene in a scattering exterior medium (global fog) runs the extended kernels WITHOUT the exchange: nearly all its vertices are volume
     vertices, one class (measured: scenes/0056_fog ptdl 124 ms with, 115 without; scenes/0055_media ptdl 35 with, 46 without) */
  /* Round 6: with the rule "shade the class whose POOL is fullest" (DScene.pool_score) the exchange pays in a fog too -- fog 37.1 -> 31.1 ms, fog ptdl 116.3 -> 105.7 -- so
     such a scene runs the extended kernels WITH the exchange and that rule; CORONA_MI_NORG=1 brings the kernels without it back (profiles/r06_levers.txt block 6) */
  { const DShapeMedium &ext = shape_med[h->num_shapes];
    const bool fog = s->media && !mb_kernels && ext.med >= 0 && ext.mu_t[3] > 0.0f && ext.albedo[3] > 0.0f;
    s->norg = s->media && !mb_kernels && (!MI_REGROUP || !MI_REGROUP_MEDIA);
    d.pool_score = fog ? 1u : 0u;
    const char *ne = getenv("CORONA_MI_NORG");        /* (experiments: 1 = never the exchange in the extended kernels, 0 = always) */
    if(ne && ne[0] && s->media && !mb_kernels) s->norg = atoi(ne) != 0;
    const char *pe = getenv("CORONA_MI_POOL_SCORE");  /* (experiments: the rule whatever the scene) */
    if(pe && pe[0]) d.pool_score = atoi(pe) ? 1u : 0u; }
  const int column = mb_kernels ? MI_STACK_LDS_MB : s->norg ? MI_STACK_LDS : MI_STACK_LDS_PLAIN;
  const size_t stack_bytes = (size_t)column*MI_BLOCK*sizeof(uint2) + (size_t)(MI_BLOCK/64)*MI_JOBS_LDS;   /* + the waves' job lists */
  const size_t isect_stack_bytes = (size_t)MI_STACK_LDS*MI_BLOCK*sizeof(uint2) + (size_t)(MI_BLOCK/64)*MI_JOBS_LDS;   /* mi_intersect_kernel: full columns, no pools */
  /* The tree lives in LDS next to the traversal stacks when it fits (0010_pt: 48 KB + 80 KB of 160 KB). Of a larger tree the TOP is
     staged -- as many of its breadth-first numbered nodes as the LDS takes next to stacks, job lists and the pools' share
     (MI_NODES_TOP_POOL) -- and the NODES_LDS = false instantiations read the rest from HBM / L2, one 128-byte record per visit.
     CORONA_MI_NODES=global forces that for a tree that would fit (tests), CORONA_MI_NODES_TOP=<nodes> limits the staged top (0: none). */
  const char *nodes_env = getenv("CORONA_MI_NODES");
  const size_t halton_bytes = h->pointsampler == MI_POINTS_HALTON ? (size_t)2*MI_HALTON_LDS : 0;     /* staged head of the permutation tables */
  /* plain ptdl kernels: emitter records in LDS. The same predicate as the kernel's (lds_setup<..., LIGHTS = PTDL && !MEDIA> advances by these
     bytes whether or not the scene has emitter records: a ptdl scene without emitters still needs them allocated) */
  const size_t lights_bytes = (h->sampler == MI_SAMPLER_PTDL && !s->media) ? (size_t)MI_LIGHTS_LDS*sizeof(DLight) : 0;
  /* (both kinds of kernel must find room: the path kernels with their columns, the ray-level test kernel with full columns -- one K for
     both, the links are baked for it) */
  const size_t lds_total = 160*1024, static_bytes = 256;    /* blk_next, the pools' control words, alignment */
  /* (mi_intersect refuses scenes with moving primitives -- its rays carry no time --, so the ray-level kernel's full columns do not count there:
     the motion-blur kernels keep both box sets of cfg-sized trees in LDS next to their 7-entry columns) */
  const size_t fixed_path = halton_bytes + lights_bytes + stack_bytes + static_bytes, fixed_isect = mb_kernels ? 0 : isect_stack_bytes + static_bytes;
  s->nodes_lds = fixed_path + node_bytes <= lds_total && fixed_isect + node_bytes <= lds_total && !(nodes_env && !strcmp(nodes_env, "global"));
  bool scatters_ = false;
  for(size_t i=0;i<shape_med.size();i++) if(shape_med[i].med >= 0 && shape_med[i].mu_t[3] > 0.0f && shape_med[i].albedo[3] > 0.0f) scatters_ = true;
  const bool pools_wanted = MI_REGROUP && !s->norg && (!mb_kernels || MI_REGROUP_MB) && num_classes + ((s->media && scatters_) ? 1u : 0u) > 1u;   /* (the predicate of the pools below) */
  /* motion-blur kernels (round 5): both box sets of a cfg-sized tree fill the LDS next to the stacks -- no room for the pools of the exchange
     between waves. With the top of the tree staged instead of all of it (the lowest levels from L2) the pools fit: MI_REGROUP_MB */
  if(mb_kernels && pools_wanted && s->nodes_lds && fixed_path + node_bytes + (size_t)MI_NODES_TOP_POOL > lds_total && !getenv("CORONA_MI_MB_ALL_LDS")) s->nodes_lds = false;
  uint32_t K = N;
  if(!s->nodes_lds)
  {
    const char *pe = getenv("CORONA_MI_NODES_POOL");              /* (experiments: the pools' share in bytes) */
    const size_t pool_share = !pools_wanted ? 0 : (pe && pe[0] && atol(pe) >= 0) ? (size_t)atol(pe) : (size_t)MI_NODES_TOP_POOL;
    const size_t room_path = fixed_path + pool_share < lds_total ? lds_total - fixed_path - pool_share : 0;
    const size_t room_isect = fixed_isect < lds_total ? lds_total - fixed_isect : 0;
    K = (uint32_t)((room_path < room_isect ? room_path : room_isect)/((size_t)SL*16));
    if(K > N) K = N;
    const char *te = getenv("CORONA_MI_NODES_TOP");
    if(te && te[0] && atol(te) >= 0 && (uint32_t)atol(te) < K) K = (uint32_t)atol(te);
    if(nodes_env && !strcmp(nodes_env, "global") && !te) K = 0;       /* the all-or-nothing switch of rounds 1-4, for tests */
  }
  d.nodes_lds = K;
  s->lds_bytes = halton_bytes + lights_bytes + (size_t)SL*K*16 + stack_bytes;
  if(N)
  { /* now that K is known: node indices in the links become record offsets, with the child's split axes (mi_bake_links_kernel) */
    uint32_t root_axes = 0;
    hipLaunchKernelGGL(mi_bake_links_kernel, dim3((4*N + 255)/256), dim3(256), 0, 0, (float4 *)s->d_nodes, (const uint32_t *)s->d_axes, N, K, SL, SH);
    if(hipGetLastError() != hipSuccess || hipMemcpy(&root_axes, s->d_axes, 4, hipMemcpyDeviceToHost) != hipSuccess)
    { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "cannot bake the child links"); }
    d.root_link = root_axes << MI_AXES_SHIFT;       /* node 0: record offset 0 */
  }
  { /* material queues (mi_regroup.h): the pools take what is left of the CU's LDS behind the job lists (plain kernels only: the extended
       ones carry more path state than an entry holds). CORONA_MI_REGROUP=0 switches the exchange off, =<bytes> limits the pools. */
    const char *re = getenv("CORONA_MI_REGROUP");
    size_t room = s->lds_bytes + static_bytes < lds_total ? lds_total - s->lds_bytes - static_bytes : 0;
    if(re && atol(re) >= 0 && (size_t)atol(re) < room) room = (size_t)atol(re);
    if(room > MI_POOL_BYTES_MAX) room = MI_POOL_BYTES_MAX;
    room &= ~(size_t)15;
    /* extended kernels: volume vertices are one more class when a medium of the scene scatters */
    bool scatters = false;
    for(size_t i=0;i<shape_med.size();i++) if(shape_med[i].med >= 0 && shape_med[i].mu_t[3] > 0.0f && shape_med[i].albedo[3] > 0.0f) scatters = true;
    d.pool_volume_class = num_classes;
    const uint32_t classes = num_classes + ((s->media && scatters) ? 1u : 0u);
    /* the class table goes into LDS behind the pools when that costs them at most a fifth of their room (16 primitives per word) */
    const size_t cls_bytes = (((size_t)h->num_prims + 15)/16)*4;
    const size_t cls_lds = cls_bytes*5 <= room ? ((cls_bytes + 15) & ~(size_t)15) : 0;
    /* the exchange runs only if EVERY kernel the scene can launch gets its 32 entries: the same formula as pool_setup (mi_regroup.h) with the
       widest entry among them (the RECORD kernels', three more words in the extended kernels). A scene in the gap -- room for the pools of
       some kernels but not of others -- used to pay for pools (shorter stack columns, no FAST rounds) that traded nothing. */
    const uint32_t ns_max = (uint32_t)PoolLayout<true, true, false>::SLOTS + (s->media ? 3u : 0u) + (uint32_t)MI_POOL_HERO_SLOTS;     /* (the HERO kernels' entries are the widest: mi_scene_set_wavelengths may come later) */
    const uint32_t e_min = (uint32_t)((room - cls_lds)/(ns_max*8u + 2u*(MI_POOL_CLASSES + 1u))) & ~7u;
    const bool on = MI_REGROUP && classes > 1 && !s->norg && (!mb_kernels || MI_REGROUP_MB) && e_min >= 32u;
    d.pool_classes = on ? classes : 0u;
    d.pool_cls_bytes = on ? (uint32_t)cls_lds : 0u;
    if(on) room -= d.pool_cls_bytes;
    d.pool_bytes = on ? (uint32_t)room : 0u;
    { /* the wavefront kernel (mi_wavefront.h): plain pt scenes, no media, no moving primitives. It lays the LDS out for itself -- shorter stack
         columns (MI_WF_COLUMN), and everything behind the job lists but the class table belongs to its five lists of entry numbers -- next to the
         megakernel's layout of the same scene (records of hero paths, mi_scene_set_wavelengths: those kernels keep their own). CORONA_MI_WAVEFRONT=0 / 1
         switches it off / on (default: MI_WAVEFRONT_DEFAULT); it needs room for 256 entries. */
      const char *we = getenv("CORONA_MI_WAVEFRONT");
      const bool wanted = (we && we[0]) ? atoi(we) != 0 : MI_WAVEFRONT_DEFAULT != 0;
      const size_t wf_fixed = halton_bytes + lights_bytes + (size_t)SL*K*16 + (size_t)MI_WF_COLUMN*MI_BLOCK*sizeof(uint2) + (size_t)(MI_BLOCK/64)*MI_JOBS_LDS + static_bytes;
      size_t wroom = wf_fixed < lds_total ? lds_total - wf_fixed : 0;
      wroom &= ~(size_t)15;
      const size_t wcls = cls_bytes*5 <= wroom ? ((cls_bytes + 15) & ~(size_t)15) : 0;
      wroom -= wcls;
      s->wavefront = wanted && h->sampler == MI_SAMPLER_PT && !s->media && !mb_kernels && wroom/(2u*(MI_POOL_CLASSES + 1u)) >= 256u &&
                     (on ? wcls == cls_lds : true);            /* (one DScene.pool_cls_bytes for both kernels) */
      if(s->wavefront)
      {
        if(!on) { d.pool_classes = classes; d.pool_cls_bytes = (uint32_t)wcls; }
        d.wf_list_bytes = (uint32_t)wroom;
        if(s->lds_bytes + d.pool_bytes + d.pool_cls_bytes < wf_fixed - static_bytes + wroom + wcls) s->lds_bytes = wf_fixed - static_bytes + wroom + wcls - d.pool_bytes - d.pool_cls_bytes;
      }
      const char *ee = getenv("CORONA_MI_WAVEFRONT_ENTRIES");
      s->wf_entries = (ee && atoi(ee) >= 256 && atoi(ee) <= 4032) ? (uint32_t)atoi(ee) & ~63u : (uint32_t)MI_WF_ENTRIES;
    }
    s->lds_bytes += d.pool_bytes + d.pool_cls_bytes;
    /* one launch size for every kernel of the scene: mi_intersect_kernel keeps full stack columns */
    const size_t isect_bytes = mb_kernels ? 0 : (size_t)SL*K*16 + isect_stack_bytes;
    if(s->lds_bytes < isect_bytes) s->lds_bytes = isect_bytes;
  }
  s->device_built = device_build; s->stack_need = stack_need;
  { const char *ce = getenv("CORONA_MI_COUNTERS"); s->counting = ce && atoi(ce) ? 1 : 0; }
  { /* default: the FAST rounds where they win (same-box A/B) -- the plain pt kernels: cfg 2 18.2 against 18.7 ms.
       The extended pt kernels break even (scenes/0055_media 21.6 / 21.4, moving camera 20.3 / 20.4), a global fog (41.3 / 39.8) and the
       ptdl kernels (40.7 / 35.2; media 56.6 / 48.8) are quicker with the exact rounds */
    /* Round 4: with the exchange between waves (mi_regroup.h) the pools want the LDS the FAST rounds park path state in, and the exact
       rounds are the quicker ones (cfg 2 15.7 against 16.7 ms): 'auto' = FAST only for plain pt scenes that run without the exchange */
    const char *te = getenv("CORONA_MI_TRAVERSAL");
    const bool fast_default = h->sampler == MI_SAMPLER_PT && !s->media && d.pool_bytes == 0u;
    if(te && strcmp(te, "exact") && strcmp(te, "fast") && strcmp(te, "auto") && te[0])
      fprintf(stderr, "[mi] CORONA_MI_TRAVERSAL=%s is not one of exact / fast / auto: using auto\n", te);
    s->fast = (te && !strcmp(te, "exact")) ? 0 : (te && !strcmp(te, "fast")) ? 1 : fast_default;
  }
  { const char *me = getenv("CORONA_MI_METAL"); d.metal_reference = (me && !strcmp(me, "reference")) ? 1u : 0u; }
  { /* the kernels this scene can launch (record / counting / traversal variants of its configuration) may use the whole LDS */
    std::vector<const void *> kernels = { (const void *)mi_intersect_kernel<true, false>, (const void *)mi_intersect_kernel<false, false>,
                                          (const void *)mi_intersect_kernel<true, true>, (const void *)mi_intersect_kernel<false, true> };
    for(unsigned k=0;k<8;k++)
    {
      const unsigned which = ((k & 1u) ? MI_WHICH_RECORD | MI_WHICH_COUNT : 0u) | ((k & 2u) ? MI_WHICH_COUNT : 0u) | (s->nodes_lds ? MI_WHICH_NODES_LDS : 0u) |
                             (h->pointsampler == MI_POINTS_HALTON ? MI_WHICH_HALTON : 0u);
#ifdef MI_DEV_FAST
      if(s->media && (k & 4u)) continue;       /* (development builds hold the extended kernels' exact rounds only) */
#endif
      if(mi_path_which_valid(which))
        kernels.push_back(path_kernel(h->sampler == MI_SAMPLER_PTDL, s->media, s->d_prims_t1 != nullptr, (k & 4u) != 0, s->norg, which, nullptr));
    }
    if(s->wavefront)
      for(unsigned k=0;k<4;k++)
      {
        const unsigned which = ((k & 1u) ? MI_WHICH_RECORD | MI_WHICH_COUNT : 0u) | ((k & 2u) ? MI_WHICH_COUNT : 0u) | (s->nodes_lds ? MI_WHICH_NODES_LDS : 0u) |
                               (h->pointsampler == MI_POINTS_HALTON ? MI_WHICH_HALTON : 0u);
        if(mi_path_which_valid(which)) kernels.push_back(mi_wave_part<false>(which, nullptr));
      }
    for(const void *k : kernels)
      if(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes) != hipSuccess)
      { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "cannot raise the dynamic LDS limit"); }
  }
  if(h->pointsampler == MI_POINTS_HALTON)
  { /* constants now, permutation tables at the first mi_render / mi_trace_paths (they depend on the end index, ensure_halton) */
    s->halton = true;
    s->halton_epoch = ~0ull;
    s->halton_tables = new HaltonTables;
    halton_layout(*s->halton_tables);
    if(!halton_camera_constants_ok(*s->halton_tables)) { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "internal: Halton camera constants out of date"); }
    if(hipMalloc(&s->d_halton_dim, sizeof(s->halton_tables->dim)) != hipSuccess ||
       hipMalloc(&s->d_halton_perm, s->halton_tables->perm.size()*sizeof(uint16_t)) != hipSuccess ||
       hipMemcpy(s->d_halton_dim, s->halton_tables->dim, sizeof(s->halton_tables->dim), hipMemcpyHostToDevice) != hipSuccess)
    { mi_scene_destroy(s); return fail(MI_ERR_NOMEM, "cannot allocate the Halton tables"); }
    d.halton_dim = (const uint4 *)s->d_halton_dim;
    d.halton_perm = (const unsigned short *)s->d_halton_perm;
  }
  hipDeviceProp_t prop;
  if(hipGetDeviceProperties(&prop, s->device) != hipSuccess) { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "hipGetDeviceProperties failed"); }
  int per_cu = (int)((160*1024)/s->lds_bytes);
  if(per_cu < 1) per_cu = 1;
  if(per_cu*MI_BLOCK > 2048) per_cu = 2048/MI_BLOCK;
  s->grid = prop.multiProcessorCount*per_cu;
  {
    const size_t extra = stack_need > MI_STACK_MIN ? (size_t)(stack_need - MI_STACK_MIN) : 1;
    if(hipMalloc(&s->d_overflow, extra*(size_t)s->grid*MI_BLOCK*sizeof(uint2)) != hipSuccess)
    { mi_scene_destroy(s); return fail(MI_ERR_NOMEM, "cannot allocate the traversal stack overflow area"); }
  }
  if(upload_rng_jump(s) != MI_OK) { mi_scene_destroy(s); return fail(MI_ERR_NOMEM, "cannot allocate the generator's jump tables"); }
  if(s->wavefront && hipMalloc(&s->d_wf_table, (size_t)s->grid*MI_WF_QUADS_MAX*s->wf_entries*sizeof(uint4)) != hipSuccess)
  { mi_scene_destroy(s); return fail(MI_ERR_NOMEM, "cannot allocate the wavefront kernel's path tables"); }
  /* uploads and clears above ran on the null stream, rendering runs on a non-blocking one: everything is in place before the first launch */
  if(hipDeviceSynchronize() != hipSuccess) { mi_scene_destroy(s); return fail(MI_ERR_DEVICE, "device synchronisation failed after the scene upload"); }
  *out = s;
  return MI_OK;
}

extern "C" int mi_scene_set_framebuffer(mi_scene *s, float *device_fb)
{
  MI_ENTER(s, "null scene");
  s->d_fb = device_fb ? device_fb : s->d_fb_own;
  s->d.fb = s->d_fb;
  return MI_OK;
}

extern "C" int mi_scene_set_counters(mi_scene *s, int enable)
{
  MI_ENTER(s, "null scene");
  s->counting = enable ? 1 : 0;
  return MI_OK;
}

extern "C" int mi_scene_set_metal_reference(mi_scene *s, int enable)
{
  MI_ENTER(s, "null scene");
  s->d.metal_reference = enable ? 1u : 0u;
  return MI_OK;
}

extern "C" int mi_scene_set_traversal(mi_scene *s, int mode)
{
  MI_ENTER(s, "null scene");
  if(mode != MI_TRAVERSAL_EXACT && mode != MI_TRAVERSAL_FAST) return fail(MI_ERR_ARG, "mi_scene_set_traversal: unknown mode");
  s->fast = mode == MI_TRAVERSAL_FAST;
  return MI_OK;
}

extern "C" int mi_scene_get_traversal(mi_scene *s)
{ /* the mode the next mi_render uses: scenes with moving primitives always run the exact rounds */
  if(!s) return -1;
  return (s->fast && !s->d_prims_t1) ? MI_TRAVERSAL_FAST : MI_TRAVERSAL_EXACT;
}

extern "C" int mi_scene_set_stream(mi_scene *s, void *hip_stream)
{
  MI_ENTER(s, "null scene");
  s->stream = hip_stream == MI_STREAM_DEFAULT ? (hipStream_t)0 : hip_stream ? (hipStream_t)hip_stream : s->stream_own;
  return MI_OK;
}

/* pointsampler_prepare_frame, src/pointsampler.d/halton.c:122-129: the permutations are drawn with seed frame + (end >> 32),
   i.e. anew whenever the end of the rendered range passes a multiple of 2^32 path indices (the index is cut to 32 bits) */
static int ensure_halton(mi_scene *s, uint64_t end_index)
{
  if(!s->halton) return MI_OK;
  const uint64_t epoch = end_index >> 32;
  if(epoch == s->halton_epoch) return MI_OK;
  halton_fill(*s->halton_tables, s->d.frame + epoch);
  HIPCHK(hipStreamSynchronize(s->stream));                 /* launches in flight still read the old tables */
  HIPCHK(hipMemcpy(s->d_halton_perm, s->halton_tables->perm.data(), s->halton_tables->perm.size()*sizeof(uint16_t), hipMemcpyHostToDevice));
  s->halton_epoch = epoch;
  return MI_OK;
}

/* ten rounds of xorshift128+ on the host (points_set_state's warm-up, src/points.d/xorshift128p.c:53-74): rng_seed_jump's tables and launch constants */
static void rng_ten_rounds(uint64_t &a0, uint64_t &a1)
{
  for(int k=0;k<10;k++)
  {
    uint64_t s1 = a0;
    const uint64_t s0 = a1;
    a0 = s0;
    s1 ^= s1 << 23; s1 ^= s1 >> 17; s1 ^= s0; s1 ^= s0 >> 26;
    a1 = s1;
  }
}
static int upload_rng_jump(mi_scene *s)
{ /* [4][256]: the ten rounds of the seed whose only set bits are byte b of s0 = value v */
  std::vector<uint32_t> t(4*256*4);
  for(int b=0;b<4;b++) for(int v=0;v<256;v++)
  {
    uint64_t a0 = (uint64_t)v << (8*b), a1 = 0;
    rng_ten_rounds(a0, a1);
    uint32_t *e = &t[(size_t)(b*256 + v)*4];
    e[0] = (uint32_t)a0; e[1] = (uint32_t)(a0 >> 32); e[2] = (uint32_t)a1; e[3] = (uint32_t)(a1 >> 32);
  }
  if(hipMalloc(&s->d_rng_jump, t.size()*4) != hipSuccess || hipMemcpy(s->d_rng_jump, t.data(), t.size()*4, hipMemcpyHostToDevice) != hipSuccess) return MI_ERR_NOMEM;
  s->d.rng_jump = (const uint4 *)s->d_rng_jump;
  return MI_OK;
}

static void launch_path_kernel(mi_scene *s, bool record, int grid, uint64_t first, uint64_t n, mi_path_record *rec)
{
  { /* the launch constant of rng_seed_jump: the rounds of (high word of 1 + index, 2 + frame) for the launch's first index; lanes of a launch that crosses a
       multiple of 2^32 see another high word and run the rounds themselves */
    uint64_t a0 = (1ull + first) & 0xffffffff00000000ull, a1 = 2ull + s->d.frame;
    s->d.rng_jump_hi = (uint32_t)(a0 >> 32);
    rng_ten_rounds(a0, a1);
    s->d.rng_jump_c[0] = (uint32_t)a0; s->d.rng_jump_c[1] = (uint32_t)(a0 >> 32); s->d.rng_jump_c[2] = (uint32_t)a1; s->d.rng_jump_c[3] = (uint32_t)(a1 >> 32);
  } /* pick the instantiation: the part by PTDL (sampler) x MEDIA ("extended": media, moving camera, emitters without a one-burst
     record) x MB (moving primitives) x FAST (traversal rounds), inside it RECORD (test hook) x NODES_LDS (tree fits LDS) x HALTON
     (point sampler) x COUNT (debug counters) */
  const unsigned which = (record ? MI_WHICH_RECORD : 0u) | (s->nodes_lds ? MI_WHICH_NODES_LDS : 0u) | (s->halton ? MI_WHICH_HALTON : 0u) |
                         ((s->counting || record) ? MI_WHICH_COUNT : 0u);
  PathLaunch L = { s->d, grid, s->lds_bytes, s->stream, (unsigned long long)first, (unsigned long long)n, (const uint32_t *)s->d_shape_material,
                   (const float *)s->d_shape_L, rec, (uint2 *)s->d_overflow, (uint2 *)s->d_wf_table, s->wf_entries };
  if(s->wavefront && !s->hero) { (void)mi_wave_part<false>(which, &L); return; }
  (void)path_kernel(s->d.sampler == MI_SAMPLER_PTDL, s->media, s->d_prims_t1 != nullptr, s->fast != 0, s->norg, which, &L, s->hero);
}

/* the share of [first, first + count) that member k of n takes: contiguous, remainder indices to the lowest members */
static inline void group_share(uint64_t first, uint64_t count, int n, int k, uint64_t *start, uint64_t *my)
{
  const uint64_t base = count/(uint64_t)n, rem = count%(uint64_t)n;
  *my = base + ((uint64_t)k < rem ? 1 : 0);
  *start = first + (uint64_t)k*base + ((uint64_t)k < rem ? (uint64_t)k : rem);
}
/* the next launch of a range of `left` path indices on a device with `grid` resident workgroups: every workgroup hands out its part of the
   launch's range through a 32-bit LDS counter, so a part stays below 2^31 paths; a small range starts fewer workgroups */
static inline uint64_t launch_chunk(uint64_t left, int grid, int *launch_grid)
{
  const uint64_t per_launch = (uint64_t)grid << 31;
  const uint64_t n = left < per_launch ? left : per_launch;
  const uint64_t need = (n + MI_BLOCK - 1)/MI_BLOCK;
  *launch_grid = (uint64_t)grid > need ? (int)need : grid;
  return n;
}

extern "C" int mi_plan_launches(uint64_t first_index, uint64_t count, int members, int grid, mi_launch *out, int max_out)
{
  if(members < 1 || grid < 1) return fail(MI_ERR_ARG, "mi_plan_launches: bad argument");
  int num = 0;
  for(int k=0;k<members;k++)
  {
    uint64_t start, my;
    group_share(first_index, count, members, k, &start, &my);
    for(uint64_t done = 0; done < my; )
    {
      int g;
      const uint64_t n = launch_chunk(my - done, grid, &g);
      if(out && num < max_out) { out[num].member = k; out[num].grid = g; out[num].first = start + done; out[num].count = n; }
      num++;
      done += n;
    }
  }
  return num;
}

extern "C" int mi_render(mi_scene *s, uint64_t first_index, uint64_t count)
{
  MI_ENTER(s, "null scene");
  if(!count) return MI_OK;
  { const int e = ensure_halton(s, first_index + count); if(e) return e; }
  HIPCHK(hipEventRecord(s->ev0, s->stream));
  s->kernel_launches_last = 0;
  for(uint64_t done = 0; done < count; )
  {
    int grid;
    const uint64_t n = launch_chunk(count - done, s->grid, &grid);        /* (mi_plan_launches: the same arithmetic, for hosts without a GPU) */
    launch_path_kernel(s, false, grid, first_index + done, n, nullptr);
    HIPCHK(hipGetLastError());
    s->kernel_launches_last++;
    done += n;
  }
  HIPCHK(hipEventRecord(s->ev1, s->stream));
  s->have_timing = 1;
  s->launches += s->kernel_launches_last;
  return MI_OK;
}

extern "C" int mi_scene_set_pixels(mi_scene *s, int mode)
{
  MI_ENTER(s, "null scene");
  if(mode != MI_PIXELS_SAMPLED && mode != MI_PIXELS_FROM_INDEX) return fail(MI_ERR_ARG, "mi_scene_set_pixels: unknown mode");
  s->d.pixels_from_index = mode == MI_PIXELS_FROM_INDEX ? 1u : 0u;
  return MI_OK;
}

extern "C" int mi_scene_set_wavelengths(mi_scene *s, int count)
{
  MI_ENTER(s, "null scene");
  if(count != 1 && count != MI_WAVELENGTHS_HERO) return fail(MI_ERR_ARG, "mi_scene_set_wavelengths: 1 or 4 wavelengths per path");
  if(count == 1) { s->hero = false; return MI_OK; }
  for(unsigned k=0;k<3;k++)
  { /* the scene's three HERO kernels (production, counting, record) may use the LDS the scene was laid out for */
    const unsigned which = (k == 2 ? MI_WHICH_RECORD | MI_WHICH_COUNT : k == 1 ? MI_WHICH_COUNT : 0u) | (s->nodes_lds ? MI_WHICH_NODES_LDS : 0u) | (s->halton ? MI_WHICH_HALTON : 0u);
    const void *kernel = path_kernel(s->d.sampler == MI_SAMPLER_PTDL, s->media, s->d_prims_t1 != nullptr, false, s->norg, which, nullptr, true);
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
  }
  s->hero = true;
  return MI_OK;
}

extern "C" int mi_render_tiles(mi_scene *s, uint64_t first_frame, uint64_t frames, uint32_t member, uint32_t members)
{ /* the paths of frames [first_frame, first_frame + frames) whose pixel lies in a tile t = member (mod members): the launch's work items are
     (frame, local tile, pixel of the tile) triples in that order, 1024 per tile; tile_path() (mi_path.h) names the path of an item */
  MI_ENTER(s, "null scene");
  if(!members || member >= members) return fail(MI_ERR_ARG, "mi_render_tiles: member must be below members");
  if(!s->d.pixels_from_index) return fail(MI_ERR_ARG, "mi_render_tiles: the scene samples its pixels (mi_scene_set_pixels(s, MI_PIXELS_FROM_INDEX) first)");
  const uint32_t tiles_x = s->width/32u, tiles = tiles_x*(s->height/32u);
  const uint32_t local = member < tiles ? (tiles - member + members - 1u)/members : 0u;
  if(!frames || !local) return MI_OK;
  if(frames > (~0ull >> 11)/local) return fail(MI_ERR_ARG, "mi_render_tiles: too many frames");
  { const int e = ensure_halton(s, (first_frame + frames)*(uint64_t)s->width*s->height); if(e) return e; }
  DScene &d = s->d;
  d.tile_members = members; d.tile_member = member; d.tiles_local = local; d.tiles_x = tiles_x;
  const uint64_t first_item = first_frame*local*1024u, count = frames*local*1024u;
  hipError_t err = hipEventRecord(s->ev0, s->stream);
  s->kernel_launches_last = 0;
  for(uint64_t done = 0; done < count && err == hipSuccess; )
  {
    int grid;
    const uint64_t n = launch_chunk(count - done, s->grid, &grid);
    launch_path_kernel(s, false, grid, first_item + done, n, nullptr);
    err = hipGetLastError();
    s->kernel_launches_last++;
    done += n;
  }
  d.tile_members = 0;          /* (the descriptor travels by value with every launch: mi_render and mi_trace_paths see a plain scene again) */
  if(err == hipSuccess) err = hipEventRecord(s->ev1, s->stream);
  if(err != hipSuccess) { snprintf(g_err, sizeof(g_err), "mi_render_tiles: %s", hipGetErrorString(err)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  s->have_timing = 1;
  s->launches += s->kernel_launches_last;
  return MI_OK;
}

extern "C" int mi_sync(mi_scene *s)
{
  MI_ENTER(s, "null scene");
  HIPCHK(hipStreamSynchronize(s->stream));
  return MI_OK;
}

extern "C" int mi_fb_read(mi_scene *s, float *host_fb, int accumulate)
{
  if(!s || !host_fb) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  const size_t n = 3*(size_t)s->width*s->height;
  HIPCHK(hipStreamSynchronize(s->stream));
  if(!accumulate) { HIPCHK(hipMemcpy(host_fb, s->d_fb, n*sizeof(float), hipMemcpyDeviceToHost)); return MI_OK; }
  /* accumulate: through a pinned staging buffer of the scene (kept for the next call) in 16 MB chunks -- the add of chunk k (a plain
     loop over contiguous floats, which the compiler vectorises) runs while chunk k+1 is in flight */
  if(!s->h_stage)
  {
    HIPCHK(hipHostMalloc((void **)&s->h_stage, n*sizeof(float), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&s->ev_stage[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&s->ev_stage[1], hipEventDisableTiming));
  }
  const size_t chunk = (size_t)1 << 22, nchunks = (n + chunk - 1)/chunk;
  auto issue = [&](size_t k) -> hipError_t
  {
    const size_t off = k*chunk, c = n - off < chunk ? n - off : chunk;
    hipError_t e = hipMemcpyAsync(s->h_stage + off, s->d_fb + off, c*sizeof(float), hipMemcpyDeviceToHost, s->stream);
    return e != hipSuccess ? e : hipEventRecord(s->ev_stage[k & 1], s->stream);
  };
  HIPCHK(issue(0));
  for(size_t k=0;k<nchunks;k++)
  {
    if(k + 1 < nchunks) HIPCHK(issue(k + 1));
    HIPCHK(hipEventSynchronize(s->ev_stage[k & 1]));
    const size_t off = k*chunk, c = n - off < chunk ? n - off : chunk;
    float *__restrict dst = host_fb + off;
    const float *__restrict src = s->h_stage + off;
    for(size_t i=0;i<c;i++) dst[i] += src[i];
  }
  return MI_OK;
}

extern "C" int mi_fb_clear(mi_scene *s)
{
  MI_ENTER(s, "null scene");
  HIPCHK(hipMemsetAsync(s->d_fb, 0, sizeof(float)*3*(size_t)s->width*s->height, s->stream));
  return MI_OK;
}

extern "C" float *mi_fb_device_ptr(mi_scene *s) { return s ? s->d_fb : nullptr; }

extern "C" int mi_counters(mi_scene *s, uint64_t out[8])
{
  if(!s || !out) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  HIPCHK(hipStreamSynchronize(s->stream));
  std::vector<unsigned long long> tmp((size_t)8*MI_COUNTER_SHARDS);
  HIPCHK(hipMemcpy(tmp.data(), s->d_counters, tmp.size()*sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for(int k=0;k<8;k++) out[k] = 0;
  for(int sh=0;sh<MI_COUNTER_SHARDS;sh++)
  {
    for(int k=0;k<7;k++) out[k] += tmp[(size_t)sh*8 + k];
#if defined(MI_PROFILE_PHASES) || defined(MI_PROFILE_TRAV)
    out[7] += tmp[(size_t)sh*8 + 7];
#else
    if(tmp[(size_t)sh*8 + 7] > out[7]) out[7] = tmp[(size_t)sh*8 + 7];
#endif
  }
  return MI_OK;
}

extern "C" int mi_trace_paths(mi_scene *s, uint64_t first_index, uint64_t count, mi_path_record *host_out)
{
  if(!s || !host_out) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  if(!count) return MI_OK;
  { const int e = ensure_halton(s, first_index + count); if(e) return e; }
  void *d_rec = nullptr;
  HIPCHK(hipMalloc(&d_rec, count*sizeof(mi_path_record)));
  hipError_t e = hipMemsetAsync(d_rec, 0, count*sizeof(mi_path_record), s->stream);
  if(e == hipSuccess)
  {
    int grid = s->grid;
    const uint64_t need = (count + MI_BLOCK - 1)/MI_BLOCK;
    if((uint64_t)grid > need) grid = (int)need;
    launch_path_kernel(s, true, grid, first_index, count, (mi_path_record *)d_rec);
    e = hipGetLastError();
  }
  if(e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if(e == hipSuccess) e = hipMemcpy(host_out, d_rec, count*sizeof(mi_path_record), hipMemcpyDeviceToHost);
  (void)hipFree(d_rec);
  if(e != hipSuccess) { snprintf(g_err, sizeof(g_err), "mi_trace_paths: %s", hipGetErrorString(e)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  return MI_OK;
}

extern "C" int mi_trace_paths_hero(mi_scene *s, uint64_t first_index, uint64_t count, mi_path_record *host_out, mi_hero_ext *host_ext)
{
  if(!s || !host_out) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  if(!s->hero) return fail(MI_ERR_ARG, "mi_trace_paths_hero: the scene renders one wavelength per path (mi_scene_set_wavelengths(s, MI_WAVELENGTHS_HERO) first)");
  if(!count) return MI_OK;
  { const int eh = ensure_halton(s, first_index + count); if(eh) return eh; }
  void *d_rec = nullptr, *d_ext = nullptr;
  HIPCHK(hipMalloc(&d_rec, count*sizeof(mi_path_record)));
  hipError_t e = hipMemsetAsync(d_rec, 0, count*sizeof(mi_path_record), s->stream);
  if(e == hipSuccess && host_ext) e = hipMalloc(&d_ext, count*sizeof(mi_hero_ext));
  if(e == hipSuccess && host_ext) e = hipMemsetAsync(d_ext, 0, count*sizeof(mi_hero_ext), s->stream);
  if(e == hipSuccess)
  {
    int grid = s->grid;
    const uint64_t need = (count + MI_BLOCK - 1)/MI_BLOCK;
    if((uint64_t)grid > need) grid = (int)need;
    s->d.hero_ext = (mi_hero_ext *)d_ext;
    launch_path_kernel(s, true, grid, first_index, count, (mi_path_record *)d_rec);
    s->d.hero_ext = nullptr;
    e = hipGetLastError();
  }
  if(e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if(e == hipSuccess) e = hipMemcpy(host_out, d_rec, count*sizeof(mi_path_record), hipMemcpyDeviceToHost);
  if(e == hipSuccess && host_ext) e = hipMemcpy(host_ext, d_ext, count*sizeof(mi_hero_ext), hipMemcpyDeviceToHost);
  (void)hipFree(d_rec);
  if(d_ext) (void)hipFree(d_ext);
  if(e != hipSuccess) { snprintf(g_err, sizeof(g_err), "mi_trace_paths_hero: %s", hipGetErrorString(e)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  return MI_OK;
}

extern "C" int mi_intersect(mi_scene *s, const mi_ray *rays, uint64_t n, mi_hit *host_out)
{
  if(!s || !rays || !host_out) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  if(s->d_prims_t1) return fail(MI_ERR_UNSUPPORTED, "mi_intersect: rays carry no time, the scene has motion-blurred primitives");
  if(!n) return MI_OK;
  void *d_rays = nullptr, *d_hits = nullptr;
  hipError_t e = hipMalloc(&d_rays, n*sizeof(mi_ray));
  if(e == hipSuccess) e = hipMalloc(&d_hits, n*sizeof(mi_hit));
  if(e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n*sizeof(mi_ray), hipMemcpyHostToDevice, s->stream);
  if(e == hipSuccess)
  {
    int grid = s->grid;
    const uint64_t need = (n + MI_BLOCK - 1)/MI_BLOCK;
    if((uint64_t)grid > need) grid = (int)need;
#define MI_ISECT(N, F) hipLaunchKernelGGL((mi_intersect_kernel<N, F>), dim3(grid), dim3(MI_BLOCK), s->lds_bytes, s->stream, s->d, (const mi_ray *)d_rays, \
                                        (unsigned long long)n, (mi_hit *)d_hits, (uint2 *)s->d_overflow)
    if(s->nodes_lds) { if(s->fast) MI_ISECT(true, true); else MI_ISECT(true, false); }
    else             { if(s->fast) MI_ISECT(false, true); else MI_ISECT(false, false); }
#undef MI_ISECT
    e = hipGetLastError();
  }
  if(e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if(e == hipSuccess) e = hipMemcpy(host_out, d_hits, n*sizeof(mi_hit), hipMemcpyDeviceToHost);
  if(d_rays) (void)hipFree(d_rays);
  if(d_hits) (void)hipFree(d_hits);
  if(e != hipSuccess) { snprintf(g_err, sizeof(g_err), "mi_intersect: %s", hipGetErrorString(e)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  return MI_OK;
}

extern "C" int mi_last_kernel_ms(mi_scene *s, float *ms)
{
  if(!s || !ms) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  if(!s->have_timing) { *ms = 0.0f; return MI_OK; }
  HIPCHK(hipEventSynchronize(s->ev1));
  HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
  return MI_OK;
}

extern "C" int mi_last_kernel_launches(mi_scene *s, uint64_t *launches)
{
  if(!s || !launches) return fail(MI_ERR_ARG, "null argument");
  *launches = s->kernel_launches_last;
  return MI_OK;
}

extern "C" int mi_scene_stats(mi_scene *s, uint32_t out[4])
{
  if(!s || !out) return fail(MI_ERR_ARG, "null argument");
  out[0] = s->d.num_nodes; out[1] = s->nodes_lds ? 1u : 0u; out[2] = (uint32_t)s->stack_need; out[3] = s->device_built ? 1u : 0u;
  return MI_OK;
}

extern "C" int mi_scene_lds_nodes(mi_scene *s, uint32_t *staged)
{
  if(!s || !staged) return fail(MI_ERR_ARG, "null argument");
  *staged = s->nodes_lds ? s->d.num_nodes : s->d.nodes_lds;
  return MI_OK;
}

extern "C" int mi_scene_kernel_name(mi_scene *s, char *buf, size_t len)
{ /* the instantiation launch_path_kernel picks for a plain mi_render (RECORD = false), spelled as rocprofv3 spells it */
  if(!s || !buf || len == 0) return fail(MI_ERR_ARG, "null argument");
  const bool mb = s->d_prims_t1 != nullptr;
  auto tf = [](bool b) { return b ? "true" : "false"; };
  if(s->wavefront && !s->hero)
  {
    const int n = snprintf(buf, len, "mi_wave_kernel<false, false, %s, %s, %s>", tf(s->nodes_lds), tf(s->halton), tf(s->counting != 0));
    return (n < 0 || (size_t)n >= len) ? fail(MI_ERR_ARG, "mi_scene_kernel_name: buffer too small") : MI_OK;
  }
  const int n = snprintf(buf, len, "mi_path_kernel<false, %s, %s, %s, %s, %s, %s, %s, %s, %s>", tf(s->d.sampler == MI_SAMPLER_PTDL), tf(s->nodes_lds), tf(s->halton),
                         tf(s->media), tf(mb), tf(s->counting != 0), tf(s->fast != 0 && !mb && !s->hero), tf(s->norg), tf(s->hero));
  return (n < 0 || (size_t)n >= len) ? fail(MI_ERR_ARG, "mi_scene_kernel_name: buffer too small") : MI_OK;
}

extern "C" void mi_scene_destroy(mi_scene *s)
{
  if(!s) return;
  (void)hipSetDevice(s->device);
  void *bufs[] = { s->d_nodes, s->d_axes, s->d_prims, s->d_primgeo, s->d_materials, s->d_light_prim, s->d_light_cdf, s->d_light_L,
                   s->d_cie, s->d_checker, s->d_metal, s->d_counters, s->d_shape_material, s->d_shape_L, s->d_overflow, s->d_fb_own,
                   s->d_halton_dim, s->d_halton_perm, s->d_shape_medium, s->d_prims_t1, s->d_lights, s->d_prim_cls, s->d_wf_table, s->d_rng_jump };
  delete s->halton_tables;
  if(s->h_stage) { (void)hipHostFree(s->h_stage); (void)hipEventDestroy(s->ev_stage[0]); (void)hipEventDestroy(s->ev_stage[1]); }
  for(void *b : bufs) if(b) (void)hipFree(b);
  if(s->stream_own) (void)hipStreamDestroy(s->stream_own);
  if(s->ev0) (void)hipEventDestroy(s->ev0);
  if(s->ev1) (void)hipEventDestroy(s->ev1);
  free(s);
}


/* ======================================================================================= test hook: the BSDF battle test on the device
 * The reference's tools/battle-test.c:57-266 (what regression/0052_dielectric and 0053_dielectric run) on the kernels' own
 * sample_* / brdf_* / pdf_* functions: a synthetic vertex with n = gn = (0, 0, +-1), shading rs .06 rd .8 rg 1 em 0 and the given
 * roughness, in vacuum; per incidence angle k (omega_in = (0, sqrt(u), -+sqrt(1 - u)), u = k / (count - 0.5))
 *   ebsdf, epdf   spp * size^2 calls of sample(): sum of the returned weights / number of samples that land in the tested hemisphere
 *   bsdf, pdf     brdf() and pdf() summed over a size^2 grid on the projected hemisphere (times 4 / size^2)
 * which regression/makebattletest.sh:13-14 compares ((bsdf - ebsdf)^2 < 1e-5, (pdf - epdf)^2 < 1e-5). */
struct BsdfTestSetup { Surf sf; Shading sh; V3 wi; float eta_ratio; uint32_t bsdf; int metal; float lambda; };
__device__ __forceinline__ BsdfTestSetup bsdf_test_setup(const mi_bsdf_test &t, uint32_t k)
{
  BsdfTestSetup b;
  const float nz = t.reflect ? 1.0f : -1.0f;
  b.sf.x = mk3(0, 0, 0); b.sf.n = b.sf.gn = mk3(0.0f, 0.0f, nz);
  get_onb(b.sf.n, b.sf.a, b.sf.b);                          /* battle-test.c:99: the plain frame, not the scrambled one */
  b.sf.u = b.sf.v = b.sf.s = b.sf.t = 0.0f; b.sf.flags = 0;
  b.sh.rs = 0.06f; b.sh.rd = 0.8f; b.sh.em = 0.0f; b.sh.rg = 1.0f; b.sh.roughness = t.roughness;
  const float u = k/((float)(t.count - 1) + .5f), v = 0.0f;
  b.wi = mk3(sqrtf(u)*sinf(2.0f*MI_PI_F*v), sqrtf(u)*cosf(2.0f*MI_PI_F*v), t.reflect ? -sqrtf(1 - u) : sqrtf(1 - u));
  b.bsdf = t.bsdf; b.metal = (int)t.param[0]; b.lambda = t.lambda;
  b.eta_ratio = 1.0f;
  if(t.bsdf == MI_BSDF_DIELECTRIC)
  { /* prepare, dielectric.c:64-81: e[1] is vacuum, the interior the glass: path_eta_ratio = 1 / eta(lambda) */
    b.eta_ratio = 1.0f/eta_from_abbe(t.param[0], t.param[1], t.lambda);
    if(fabsf(1.0f - b.eta_ratio/1.0f) < 1e-3f) b.sh.roughness = 0.0f;
  }
  return b;
}

__global__ void mi_bsdf_test_sample_kernel(DScene sc, mi_bsdf_test t, uint32_t k, unsigned long long samples, double *out)
{ /* out[0] += sum of weights, out[1] += samples in the tested hemisphere */
  const BsdfTestSetup b = bsdf_test_setup(t, k);
  double sw = 0.0, sn = 0.0;
  for(unsigned long long i=(unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; i<samples; i+=(unsigned long long)gridDim.x*blockDim.x)
  {
    Rng rng;
    rng_seed(rng, i, 666ull + k);
    PointSampler<false> pts(sc, rng, i, 0);
    BsdfSample bs;
    if(b.bsdf == MI_BSDF_DIFFUSE) sample_diffuse(pts, b.sf, b.sh, s_absorb, bs);
    else if(b.bsdf == MI_BSDF_DIELECTRIC) sample_dielectric(pts, b.sf, b.sh, b.wi, b.eta_ratio, s_absorb, bs);
    else sample_metal(sc, pts, b.sf, b.sh, b.wi, 1.0f, b.metal, b.lambda, s_absorb, bs);
    float w = bs.weight;
    if(!(bs.omega.x == bs.omega.x)) w = 1e20f;             /* battle-test.c:138-145: a NaN direction shows up as a huge weight */
    if(bs.omega.z <= 0.0f || !(bs.omega.z == bs.omega.z)) continue;
    if(w <= 0.0f) continue;
    sw += (double)w; sn += 1.0;
  }
  for(int off=32;off>0;off>>=1) { sw += __shfl_down(sw, off); sn += __shfl_down(sn, off); }
  if(__lane_id() == 0) { atomicAdd(out + 0, sw); atomicAdd(out + 1, sn); }
}

__global__ void mi_bsdf_test_eval_kernel(DScene sc, mi_bsdf_test t, uint32_t k, double *out)
{ /* out[2] += brdf over the grid, out[3] += pdf over the grid (battle-test.c:177-213; the spp repetitions of a cell are identical) */
  const BsdfTestSetup b = bsdf_test_setup(t, k);
  double sb = 0.0, sp = 0.0;
  const unsigned long long cells = (unsigned long long)t.size*t.size;
  for(unsigned long long c=(unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; c<cells; c+=(unsigned long long)gridDim.x*blockDim.x)
  {
    const int i = (int)(c % t.size), j = (int)(c / t.size);
    const float ox = (float)(2.0*i/(float)t.size - 1.), oy = (float)(2.0*j/(float)t.size - 1.);
    const float len2 = ox*ox + oy*oy;
    if(!(len2 < 1.)) continue;
    const V3 wo = mk3(ox, oy, sqrtf(1 - len2));
    BsdfEval be;
    float pdf;
    if(b.bsdf == MI_BSDF_DIFFUSE) { be = brdf_diffuse(b.sf, b.sh, wo); pdf = (float)(1.0f/MI_PI_D); }
    else if(b.bsdf == MI_BSDF_DIELECTRIC) { be = brdf_dielectric(b.sf, b.sh, b.wi, wo, b.eta_ratio); pdf = pdf_dielectric(b.sf, b.sh, b.wi, wo, b.eta_ratio, be.mode); }
    else { be = brdf_metal(sc, b.sf, b.sh, b.wi, wo, 1.0f, b.metal, b.lambda); pdf = pdf_metal(b.sf, b.sh, b.wi, wo, be.mode); }
    const float scale = 4.0f/(float)(t.size*t.size);
    sb += (double)(be.value*scale); sp += (double)(pdf*scale);
  }
  for(int off=32;off>0;off>>=1) { sb += __shfl_down(sb, off); sp += __shfl_down(sp, off); }
  if(__lane_id() == 0) { atomicAdd(out + 2, sb); atomicAdd(out + 3, sp); }
}

extern "C" int mi_bsdf_test_run(mi_scene *s, const mi_bsdf_test *t, double *out)
{
  if(!s || !t || !out) return fail(MI_ERR_ARG, "null argument");
  MI_ENTER(s, "null scene");
  if(t->bsdf > MI_BSDF_METAL || !t->count || t->count > 64 || !t->size || !t->spp) return fail(MI_ERR_ARG, "mi_bsdf_test_run: bad description");
  if(t->bsdf == MI_BSDF_METAL && (!s->d_metal || t->param[0] < 0.0f || t->param[0] > 4.0f)) return fail(MI_ERR_ARG, "mi_bsdf_test_run: no such metal");
  double *d_out = nullptr;
  HIPCHK(hipMalloc((void **)&d_out, sizeof(double)*4*t->count));
  hipError_t e = hipMemsetAsync(d_out, 0, sizeof(double)*4*t->count, s->stream);
  const unsigned long long samples = (unsigned long long)t->spp*t->size*t->size;
  for(uint32_t k=0;k<t->count && e == hipSuccess;k++)
  {
    hipLaunchKernelGGL(mi_bsdf_test_sample_kernel, dim3(1024), dim3(256), 0, s->stream, s->d, *t, k, samples, d_out + 4*k);
    hipLaunchKernelGGL(mi_bsdf_test_eval_kernel, dim3(256), dim3(256), 0, s->stream, s->d, *t, k, d_out + 4*k);
    e = hipGetLastError();
  }
  if(e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if(e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(double)*4*t->count, hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  if(e != hipSuccess) { snprintf(g_err, sizeof(g_err), "mi_bsdf_test_run: %s", hipGetErrorString(e)); fprintf(stderr, "[mi] %s\n", g_err); return MI_ERR_DEVICE; }
  for(uint32_t k=0;k<t->count;k++) { out[4*k+1] /= (double)samples; const double w = out[4*k+0]/(double)samples; out[4*k+0] = w; }
  /* order like the reference prints them: ebsdf bsdf epdf pdf */
  for(uint32_t k=0;k<t->count;k++) { const double ebsdf = out[4*k+0], epdf = out[4*k+1], bsdf = out[4*k+2], pdf = out[4*k+3]; out[4*k+0] = ebsdf; out[4*k+1] = bsdf; out[4*k+2] = epdf; out[4*k+3] = pdf; }
  return MI_OK;
}

/* ======================================================================================= several GPUs behind the C ABI
 * mi_group: one scene per device, driven by ONE host thread -- what the reference's view_render does with its pthread pool
 * (src/view.c:630-695: hand the progression's path indices to the workers, wait, the framebuffer is shared) for the GPUs of a node.
 * Paths are independent, so member k renders its contiguous share of [first, first + count) (the same split as
 * shard_range of the Python view / bench.py) into its own device framebuffer; mi_group_fb_reduce then adds the members'
 * framebuffers into member 0's: ncclReduce(ncclFloat, ncclSum, root 0) over xGMI when the devices are distinct and RCCL can be
 * loaded (dlopen: the library does not link against it), else peer copies into a buffer on the root's device and an add kernel
 * there (also the path of a group with a device named twice: tests on a one-GPU box). */
#include <dlfcn.h>

__global__ void mi_fb_add_kernel(float *__restrict dst, const float *__restrict src, size_t n)
{
  for(size_t i=(size_t)blockIdx.x*blockDim.x + threadIdx.x; i<n; i+=(size_t)gridDim.x*blockDim.x) dst[i] += src[i];
}

struct mi_rccl
{ /* the five entry points of librccl used here (rccl.h: ncclCommInitAll :236, ncclCommDestroy :260, ncclReduce :550, ncclGroupStart/End :923) */
  void *lib;
  int (*CommInitAll)(void **comm, int ndev, const int *devlist);
  int (*CommDestroy)(void *comm);
  int (*Reduce)(const void *send, void *recv, size_t count, int datatype, int op, int root, void *comm, hipStream_t stream);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
  const char *(*GetErrorString)(int);
};

struct mi_group
{
  int n;
  mi_scene **member;
  void **comm;                /* RCCL communicators, one per member, or NULL: peer-copy reduce */
  mi_rccl rccl;
  float *d_stage;             /* peer-copy reduce: one framebuffer on the root's device */
  hipEvent_t *rendered;       /* per member: everything queued on its stream for its framebuffer (render, clear) is complete */
  hipEvent_t reduced;         /* peer-copy reduce: the root has read the members' framebuffers */
  size_t fb_floats;
  bool broken;                /* a member's launch failed in mi_group_render: the framebuffers hold a partial frame until mi_group_fb_clear */
};

/* the group entry points select their members' devices; the calling thread gets its own device back (callers mix in torch / HIP code) */
struct DeviceRestore
{
  int dev; bool ok;
  DeviceRestore() : dev(0), ok(hipGetDevice(&dev) == hipSuccess) {}
  ~DeviceRestore() { if(ok) (void)hipSetDevice(dev); }
};

static bool rccl_load(mi_rccl *r)
{
  memset(r, 0, sizeof(*r));
  const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  for(const char *nm : names) if((r->lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
  if(!r->lib) return false;
  r->CommInitAll = (int (*)(void **, int, const int *))dlsym(r->lib, "ncclCommInitAll");
  r->CommDestroy = (int (*)(void *))dlsym(r->lib, "ncclCommDestroy");
  r->Reduce = (int (*)(const void *, void *, size_t, int, int, int, void *, hipStream_t))dlsym(r->lib, "ncclReduce");
  r->GroupStart = (int (*)(void))dlsym(r->lib, "ncclGroupStart");
  r->GroupEnd = (int (*)(void))dlsym(r->lib, "ncclGroupEnd");
  r->GetErrorString = (const char *(*)(int))dlsym(r->lib, "ncclGetErrorString");
  if(r->CommInitAll && r->CommDestroy && r->Reduce && r->GroupStart && r->GroupEnd) return true;
  dlclose(r->lib); r->lib = nullptr;
  return false;
}

extern "C" void mi_group_destroy(mi_group *g)
{
  if(!g) return;
  DeviceRestore restore;
  for(int k=0;k<g->n;k++)
  {
    if(g->comm && g->comm[k]) { (void)hipSetDevice(g->member[k]->device); (void)g->rccl.CommDestroy(g->comm[k]); }
    if(g->rendered && g->rendered[k]) { (void)hipSetDevice(g->member[k]->device); (void)hipEventDestroy(g->rendered[k]); }
  }
  if(g->d_stage) { (void)hipSetDevice(g->member[0]->device); (void)hipFree(g->d_stage); }
  if(g->reduced) { (void)hipSetDevice(g->member[0]->device); (void)hipEventDestroy(g->reduced); }
  for(int k=0;k<g->n;k++) if(g->member && g->member[k]) mi_scene_destroy(g->member[k]);
  if(g->rccl.lib) dlclose(g->rccl.lib);
  free(g->member); free(g->comm); free(g->rendered);
  free(g);
}

extern "C" int mi_group_create(const mi_scene_desc *desc, const int *devices, int n, mi_group **out)
{
  if(!desc || !out || n < 1 || n > 64) return fail(MI_ERR_ARG, "mi_group_create: bad argument");
  DeviceRestore restore;
  int visible = 0;
  if(hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return fail(MI_ERR_DEVICE, "no HIP device visible");
  mi_group *g = (mi_group *)calloc(1, sizeof(mi_group));
  if(!g) return fail(MI_ERR_NOMEM, "out of host memory");
  g->n = n;
  g->member = (mi_scene **)calloc(n, sizeof(mi_scene *));
  g->rendered = (hipEvent_t *)calloc(n, sizeof(hipEvent_t));
  if(!g->member || !g->rendered) { mi_group_destroy(g); return fail(MI_ERR_NOMEM, "out of host memory"); }
  bool distinct = true;
  for(int k=0;k<n;k++)
  {
    const int dev = devices ? devices[k] : k;                         /* NULL: devices 0 .. n-1 */
    if(dev < 0 || dev >= visible) { mi_group_destroy(g); return fail(MI_ERR_ARG, "mi_group_create: no such device"); }
    for(int j=0;j<k;j++) if(g->member[j]->device == dev) distinct = false;
    const int e = scene_create_on(desc, dev, &g->member[k]);
    if(e) { mi_group_destroy(g); return e; }
    if(hipSetDevice(dev) != hipSuccess || hipEventCreateWithFlags(&g->rendered[k], hipEventDisableTiming) != hipSuccess)
    { mi_group_destroy(g); return fail(MI_ERR_DEVICE, "mi_group_create: cannot create an event"); }
  }
  g->fb_floats = 3*(size_t)desc->width*desc->height;
  /* the reduce: RCCL when the members sit on distinct devices (CORONA_MI_GROUP_REDUCE=peer forces the copies; =rccl insists) */
  const char *mode = getenv("CORONA_MI_GROUP_REDUCE");
  const bool want_rccl = mode ? !strcmp(mode, "rccl") : n > 1;
  if(want_rccl && distinct && rccl_load(&g->rccl))
  {
    g->comm = (void **)calloc(n, sizeof(void *));
    std::vector<int> devs(n);
    for(int k=0;k<n;k++) devs[k] = g->member[k]->device;
    const int r = g->comm ? g->rccl.CommInitAll(g->comm, n, devs.data()) : 1;
    if(r != 0)
    {
      fprintf(stderr, "[mi] mi_group_create: ncclCommInitAll failed (%s), using peer copies\n", g->rccl.GetErrorString ? g->rccl.GetErrorString(r) : "?");
      free(g->comm); g->comm = nullptr;
    }
  }
  else if(mode && !strcmp(mode, "rccl")) { mi_group_destroy(g); return fail(MI_ERR_UNSUPPORTED, "mi_group_create: RCCL reduce asked for, but the devices are not distinct or librccl cannot be loaded"); }
  if(!g->comm && n > 1)
  {
    if(hipSetDevice(g->member[0]->device) != hipSuccess || hipMalloc((void **)&g->d_stage, g->fb_floats*sizeof(float)) != hipSuccess ||
       hipEventCreateWithFlags(&g->reduced, hipEventDisableTiming) != hipSuccess)
    { mi_group_destroy(g); return fail(MI_ERR_NOMEM, "mi_group_create: cannot allocate the staging framebuffer"); }
    for(int k=1;k<n;k++) if(g->member[k]->device != g->member[0]->device)
    { /* peer access where the hardware offers it; hipMemcpyPeerAsync works without it (through the host) */
      int can = 0;
      if(hipDeviceCanAccessPeer(&can, g->member[0]->device, g->member[k]->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(g->member[k]->device, 0);
      (void)hipGetLastError();
    }
  }
  *out = g;
  return MI_OK;
}

extern "C" int mi_group_size(mi_group *g) { return g ? g->n : 0; }
extern "C" mi_scene *mi_group_scene(mi_group *g, int k) { return (g && k >= 0 && k < g->n) ? g->member[k] : nullptr; }
extern "C" int mi_group_uses_rccl(mi_group *g) { return g && g->comm ? 1 : 0; }

extern "C" int mi_group_render(mi_group *g, uint64_t first_index, uint64_t count)
{ /* member k takes the k-th contiguous share of the range (remainder indices to the lowest members); returns once everything is queued */
  if(!g) return fail(MI_ERR_ARG, "null group");
  DeviceRestore restore;
  for(int k=0;k<g->n;k++)
  {
    uint64_t start, my;
    group_share(first_index, count, g->n, k, &start, &my);
    int e = mi_render(g->member[k], start, my);
    if(!e && hipEventRecord(g->rendered[k], g->member[k]->stream) != hipSuccess) e = fail(MI_ERR_DEVICE, "mi_group_render: cannot record an event");
    if(e)
    { /* the members before k have their shares queued: let them finish, and refuse to reduce or read the partial frame (mi_group_fb_clear
         makes the group usable again) */
      for(int j=0;j<k;j++) (void)mi_sync(g->member[j]);
      g->broken = true;
      return e;
    }
  }
  return MI_OK;
}

extern "C" int mi_group_render_tiles(mi_group *g, uint64_t first_frame, uint64_t frames)
{ /* member k renders the tiles t = k (mod n) of every frame: no member touches another's pixels except through the filter's 4 x 4 footprint
     at tile borders, which the reduce adds up like everything else */
  if(!g) return fail(MI_ERR_ARG, "null group");
  DeviceRestore restore;
  for(int k=0;k<g->n;k++)
  {
    int e = g->member[k]->d.pixels_from_index ? MI_OK : mi_scene_set_pixels(g->member[k], MI_PIXELS_FROM_INDEX);
    if(!e) e = mi_render_tiles(g->member[k], first_frame, frames, (uint32_t)k, (uint32_t)g->n);
    if(!e && hipEventRecord(g->rendered[k], g->member[k]->stream) != hipSuccess) e = fail(MI_ERR_DEVICE, "mi_group_render_tiles: cannot record an event");
    if(e)
    {
      for(int j=0;j<k;j++) (void)mi_sync(g->member[j]);
      g->broken = true;
      return e;
    }
  }
  return MI_OK;
}

extern "C" int mi_group_fb_reduce(mi_group *g)
{ /* member 0's framebuffer += the others', which are cleared: afterwards the root holds everything rendered so far */
  if(!g) return fail(MI_ERR_ARG, "null group");
  if(g->broken) return fail(MI_ERR_DEVICE, "mi_group_fb_reduce: a launch of the last mi_group_render failed, the framebuffers hold a partial frame (mi_group_fb_clear first)");
  if(g->n == 1) return MI_OK;
  DeviceRestore restore;
  mi_scene *root = g->member[0];
  if(g->comm)
  {
    int r = g->rccl.GroupStart();
    for(int k=0;k<g->n && r == 0;k++)
    {
      HIPCHK(hipSetDevice(g->member[k]->device));
      r = g->rccl.Reduce(g->member[k]->d_fb, g->member[k]->d_fb, g->fb_floats, 7 /* ncclFloat32 */, 0 /* ncclSum */, 0, g->comm[k], g->member[k]->stream);
    }
    const int r2 = g->rccl.GroupEnd();
    if(r != 0 || r2 != 0) return fail(MI_ERR_DEVICE, "mi_group_fb_reduce: ncclReduce failed");
  }
  else
  {
    HIPCHK(hipSetDevice(root->device));
    for(int k=1;k<g->n;k++)
    {
      HIPCHK(hipStreamWaitEvent(root->stream, g->rendered[k], 0));
      HIPCHK(hipMemcpyPeerAsync(g->d_stage, root->device, g->member[k]->d_fb, g->member[k]->device, g->fb_floats*sizeof(float), root->stream));
      hipLaunchKernelGGL(mi_fb_add_kernel, dim3(2048), dim3(256), 0, root->stream, root->d_fb, (const float *)g->d_stage, g->fb_floats);
      HIPCHK(hipGetLastError());
    }
    /* the members' buffers may be cleared once the root has read them */
    HIPCHK(hipEventRecord(g->reduced, root->stream));
    for(int k=1;k<g->n;k++) { HIPCHK(hipSetDevice(g->member[k]->device)); HIPCHK(hipStreamWaitEvent(g->member[k]->stream, g->reduced, 0)); }
  }
  for(int k=1;k<g->n;k++)
  { /* ... and the root's NEXT copy of a member's buffer has to wait for this clear: `rendered` covers it (two reduces in a row, or a
       reduce and a read, used to copy buffers whose clear was still in flight) */
    const int e = mi_fb_clear(g->member[k]); if(e) return e;
    HIPCHK(hipEventRecord(g->rendered[k], g->member[k]->stream));
  }
  return MI_OK;
}

extern "C" int mi_group_sync(mi_group *g)
{
  if(!g) return fail(MI_ERR_ARG, "null group");
  DeviceRestore restore;
  for(int k=0;k<g->n;k++) { const int e = mi_sync(g->member[k]); if(e) return e; }
  return MI_OK;
}

extern "C" int mi_group_fb_clear(mi_group *g)
{
  if(!g) return fail(MI_ERR_ARG, "null group");
  DeviceRestore restore;
  for(int k=0;k<g->n;k++)
  {
    const int e = mi_fb_clear(g->member[k]); if(e) return e;
    HIPCHK(hipEventRecord(g->rendered[k], g->member[k]->stream));
  }
  g->broken = false;
  return MI_OK;
}

extern "C" int mi_group_fb_read(mi_group *g, float *host_fb, int accumulate)
{ /* reduce, then the root's framebuffer to the host (copy or add, like mi_fb_read) */
  if(!g) return fail(MI_ERR_ARG, "null group");
  DeviceRestore restore;
  int e = mi_group_fb_reduce(g);
  if(!e) e = mi_group_sync(g);
  if(!e) e = mi_fb_read(g->member[0], host_fb, accumulate);
  return e;
}

extern "C" int mi_group_counters(mi_group *g, uint64_t out[8])
{ /* sums over the members (slot 7, the deepest stack, is their maximum) */
  if(!g || !out) return fail(MI_ERR_ARG, "null argument");
  DeviceRestore restore;
  for(int i=0;i<8;i++) out[i] = 0;
  for(int k=0;k<g->n;k++)
  {
    uint64_t c[8];
    const int e = mi_counters(g->member[k], c);
    if(e) return e;
    for(int i=0;i<7;i++) out[i] += c[i];
    if(c[7] > out[7]) out[7] = c[7];
  }
  return MI_OK;
}

extern "C" void mi_shutdown(void) { g_device = -1; }
