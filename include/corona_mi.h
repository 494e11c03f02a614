/* corona_mi.h -- C ABI of the MI355X path-tracing backend (libcorona_mi.so).
 *
 * This is the drop-in boundary for ONE hot path of hanatos/corona-13: everything the
 * reference reaches from work_sample() (src/view.c:618-628), i.e. the loop that
 * view_render() (src/view.c:630-695, lines 643-645) hands to its pthread pool:
 *
 *     for every path index i in [counter, end):  render_sample_path(i)
 *                                                (src/render.d/gi.c:81-105)
 *
 * with render_sample_path -> pointsampler_mutate -> sampler_create_path (pt: src/sampler.d/pt.c:40-54,
 * ptdl: src/sampler.d/ptdl.c:112-150) -> path_extend/path_propagate (src/pathspace.c:167-271,697-895)
 * -> accel_intersect (src/accel.d/qbvhmp.c:1262-1390) / prims_intersect (src/prims.c:638-672)
 * -> shader_prepare/shader_sample (src/shader.c:462-542,577-590) -> view_splat (src/view.c:455-463).
 *
 * The host side (scene files, QBVH build, light CDF, progression loop, PFM output)
 * stays plain C and fills one mi_scene_desc; the backend owns a device-resident copy
 * and a device framebuffer with the layout of the reference's fb_init(..,3,..)
 * (include/framebuffer.h:76-113): float[3*(x + width*y)], un-normalised sums.
 *
 * Conventions mirror the reference: plain pointers and sizes, int return
 * 0 = ok / negative = failure, diagnostics on stderr with an "[mi]" prefix, single
 * calling thread per scene. No torch / C++ types cross this boundary.
 */
#ifndef CORONA_MI_H
#define CORONA_MI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 3

/* ---- error codes (negative ints, reference style: non-zero == failure) ---------- */
#define MI_OK              0
#define MI_ERR_ARG        -1   /* bad argument / inconsistent scene description      */
#define MI_ERR_DEVICE     -2   /* no usable HIP device, or a HIP call failed           */
#define MI_ERR_NOMEM      -3
#define MI_ERR_UNSUPPORTED -4  /* scene uses a feature outside the hot-path scope      */

/* ---- samplers (compile-time MOD_sampler in the reference) ----------------------- */
#define MI_SAMPLER_PT   0      /* src/sampler.d/pt.c   */
#define MI_SAMPLER_PTDL 1      /* src/sampler.d/ptdl.c */

/* ---- primitive id: the reference's packed 64-bit primid_t
 *      (include/corona_common.h:45-53): extra:3 | shapeid:29 | vi:28 | mb:1 | vcnt:3 */
typedef uint64_t mi_primid;
#define MI_PRIMID_INVALID 0xffffffffffffffffull
#define MI_PRIMID_EXTRA(p)   ((uint32_t)((p) & 7u))
#define MI_PRIMID_SHAPE(p)   ((uint32_t)(((p) >> 3) & 0x1fffffffu))
#define MI_PRIMID_VI(p)      ((uint32_t)(((p) >> 32) & 0x0fffffffu))
#define MI_PRIMID_MB(p)      ((uint32_t)(((p) >> 60) & 1u))
#define MI_PRIMID_VCNT(p)    ((uint32_t)(((p) >> 61) & 7u))
#define MI_PRIM_SPHERE 1       /* include/prims.h:9-18 */
#define MI_PRIM_LINE   2
#define MI_PRIM_TRI    3
#define MI_PRIM_QUAD   4

/* ---- geometry store (include/prims.h:20-47), flattened over shapes ----------------- */
typedef struct mi_vtxidx { uint32_t v, uv; } mi_vtxidx;            /* prims_vtxidx_t */
typedef struct mi_vtx    { float v[3]; uint32_t n; } mi_vtx;       /* prims_vtx_t: n = oct normal or radius bits */
typedef struct mi_shape
{
  int32_t  material;          /* shader index of the shape line in the .nra2           */
  uint32_t num_prims;
  uint32_t vtxidx_base;       /* primid.vi is relative to this offset into vtxidx[]    */
  uint32_t vtx_base;          /* vtxidx.v  is relative to this offset into vtx[]       */
} mi_shape;

/* ---- 4-wide BVH node as the host builder emits it (cf. qbvh_node_t, src/accel.d/qbvhmp.c:62-81,
 *      static scene: aabb1 == aabb0). child: bit63 = leaf, then (first_prim<<5)|count, else node index. */
#define MI_NODE_LEAF (1ull<<63)
typedef struct mi_node
{
  float    aabb[6][4];        /* [0..2][c] = min xyz of child c, [3..5][c] = max xyz   */
  uint64_t child[4];
  int32_t  axis0, axis00, axis01;
  int32_t  parent;
} mi_node;
typedef struct mi_node_aabb { float aabb[6][4]; } mi_node_aabb;   /* one more set of child boxes, same layout as mi_node.aabb */

/* ---- materials: the dlopen'ed shader chain of one .nra2 material, compiled to a closed set
 *      (src/shader.c:693-757, src/shaders/{mult,color,colorcheckersg,dielectric,metal}.c) ---------- */
#define MI_BSDF_DIFFUSE    0   /* builtin diffuse, src/shader.c:157-257          */
#define MI_BSDF_DIELECTRIC 1   /* src/shaders/dielectric.c                       */
#define MI_BSDF_METAL      2   /* src/shaders/metal.c                            */
#define MI_BSDF_MEDIUM     3   /* homogeneous medium, src/shaders/medium_rgb.c: only as the interior of a surface material */
#define MI_BSDF_NONE       255 /* shader kind outside the scope; error if a shape uses it */

#define MI_OP_COLOR   0        /* src/shaders/color.c:75-82                      */
#define MI_OP_CHECKER 1        /* src/shaders/colorcheckersg.c:244-262           */

#define MI_SLOT_DIFFUSE  0     /* tex_slot_t, src/shaders/texture.h              */
#define MI_SLOT_SPECULAR 1
#define MI_SLOT_GLOSSY   2
#define MI_SLOT_EMISSION 3
#define MI_SLOT_VOLUME   4     /* albedo of a medium: mu_s = albedo * mu_t (texture.h:48-53, medium_rgb.c:45-59) */
#define MI_SLOT_ROUGHNESS 5
#define MI_SLOT_UNUSED   6

#define MI_MAX_OPS 4
typedef struct mi_shade_op
{
  uint32_t kind;              /* MI_OP_*                                          */
  uint32_t slot;              /* MI_SLOT_*                                        */
  float    coeff[3];          /* rgb2spec sigmoid-polynomial coefficients         */
  float    mul;               /* scale (> 1 for emission)                         */
  float    roughness;         /* written to shading.roughness by MI_OP_COLOR      */
  uint32_t pad;
} mi_shade_op;

typedef struct mi_material
{
  uint32_t    bsdf;           /* MI_BSDF_*                                        */
  uint32_t    num_ops;        /* prepare chain, executed in order before the bsdf's own prepare */
  mi_shade_op op[MI_MAX_OPS];
  float       param[4];       /* dielectric: n_d, abbe ; metal: table id, -, - ; medium: rgb2spec coefficients of mu_t, scale */
  float       mean_cos;       /* medium: Henyey-Greenstein mean cosine g                                  */
  int32_t     interior;       /* `interior <surface> <medium>` (src/shaders/interior.c): shader id of the MI_BSDF_MEDIUM material
                                 that fills the shape (its colour op in the volume slot = single-scattering albedo), or -1 */
} mi_material;

/* ---- thin-lens camera, resolved for a static camera (src/camera.d/thinlens.c:68-128,
 *      src/view.c:903-919,938-948) -------------------------------------------------------------- */
typedef struct mi_camera
{
  float pos[3];
  float a[3], b[3], n[3];     /* normalised camera frame (right, up, forward)     */
  float focus, focal_length;
  float film_width, film_height;
  float f_stop;               /* view_av2fstop(aperture_value)                    */
  float exposure_time;        /* view_tv2time(exposure_value)                     */
  float iso;
  float time_scale;           /* view_sample_time: min(1, exposure/ (1/30))       */
  /* camera motion blur (src/view.c:903-919): the frame of a path is the slerp of the two orientations at the path's time in
     [0, time_scale], the position the lerp of pos and pos_t1. moving == 0: pos/a/b/n above are the (static) frame. */
  uint32_t moving;
  float pos_t1[3];
  float orient[4], orient_t1[4];   /* quaternions w, x, y, z (include/quaternion.h:23-27) */
} mi_camera;

/* ---- emitter list (src/lights.d/list.c:56-104) ---------------------------------- */
typedef struct mi_lights
{
  uint32_t         num_prims;
  const mi_primid *primid;    /* with shapeid filled in                            */
  const float     *cdf;       /* normalised, cdf[num-1] == 1                       */
  const float     *L;         /* per prim pdf = L/sum(L*area)                      */
  float            p_sky, p_geo, p_vol;
} mi_lights;

/* ---- point sampler (MOD_pointsampler) ---------------------------------------------
 * MI_POINTS_RAND   src/pointsampler.d/rand.c:48-55: every dimension is the next number of the per-path generator.
 * MI_POINTS_HALTON src/pointsampler.d/halton.c:69-84: dimension d of path `index` is the Faure-style permuted radical
 *                  inverse of the low 32 bits of the index in the d-th prime base (ext/halton/halton.h, 256 dimensions),
 *                  d = rand_beg of the vertex under construction + the path_sample_dim_t offset (include/pathspace.h:16-53);
 *                  the permutations are drawn from srand48(frame + (end_index >> 32)) by the backend itself
 *                  (pointsampler_init / pointsampler_prepare_frame, halton.c:46-52,122-129), where end_index is the end of
 *                  the mi_render range. The tangent-frame scrambling and ptdl's nee_probability draw stay on the per-path
 *                  generator as in the reference (src/pathspace.c:213, src/sampler.d/ptdl.c:137). */
#define MI_POINTS_RAND   0
#define MI_POINTS_HALTON 1

/* ---- everything the backend needs ------------------------------------------------- */
typedef struct mi_scene_desc
{
  uint32_t struct_size;       /* = sizeof(mi_scene_desc), checked                 */
  uint32_t abi_version;       /* = MI_ABI_VERSION                                 */

  uint32_t width, height;     /* film, already padded to multiples of 32 (src/view.c:294-296) */
  uint32_t max_verts;         /* PATHSPACE_MAX_VERTS (include/pathspace.h:10-13)  */
  uint32_t sampler;           /* MI_SAMPLER_*                                     */
  uint64_t frame;             /* rt.anim_frame, seeds the per-path generator      */

  uint32_t         num_nodes;
  const mi_node   *nodes;     /* node 0 = root. NULL (with num_nodes 0): no tree is handed over and the backend builds its own
                                 4-wide BVH on the device (replaces accel_build, src/accel.d/qbvhmp.c:425-1144); primid may
                                 then be in any order, hits report the same primids, traversal counters differ */
  float            aabb[6];   /* scene box (accel_aabb)                           */
  uint64_t         num_prims;
  const mi_primid *primid;    /* in builder order (leaves index into this)        */

  uint32_t         num_shapes;
  const mi_shape  *shapes;
  uint64_t         num_vtxidx;
  const mi_vtxidx *vtxidx;
  uint64_t         num_vtx;
  const mi_vtx    *vtx;

  uint32_t           num_materials;
  const mi_material *materials;   /* indexed by shader id                         */

  mi_lights  lights;
  mi_camera  cam;

  const float *cie_xyz;       /* 96 x 3: CIE 1931 2-deg CMF, 360..830 nm step 5 + one zero row (include/spectrum.h:66-170) */
  const float *checker;       /* 140 x 36 colour-checker reflectances, 380 nm step 10 (src/shaders/colorcheckersg.c:51) or NULL */
  const float *metal_ior;     /* 5 x 95 x 2 (n,k) conductor tables, 360 nm step 5 (src/shaders/fresnel.h:21-27) or NULL */

  uint32_t pointsampler;      /* MI_POINTS_*: which MOD_pointsampler maps (path, dimension) to a number in [0,1) */
  uint32_t exterior;          /* 0: the scene sits in vacuum; k+1: material k (MI_BSDF_MEDIUM) is the global exterior medium,
                                 `exterior <k> 0` in the .nra2 (src/shader.c:544-565,699-716; volume lights are out of scope) */
  const mi_node_aabb *nodes_t1; /* [num_nodes] or NULL. The child boxes at shutter CLOSE (qbvh_node_t.aabb1, src/accel.d/qbvhmp.c:62-81,
                                 259-283,854-873: leaves refitted to the shutter-close state of their primitives, inner nodes bottom-up).
                                 With it mi_node.aabb holds the shutter-OPEN boxes and every ray tests the boxes interpolated at its
                                 path's time, aabb (1 - t) + aabb1 t (qbvhmp.c:1188-1224) -- the reference's traversal work for scenes
                                 with motion-blurred primitives. NULL: mi_node.aabb must enclose the whole motion (static test) */
} mi_scene_desc;

typedef struct mi_scene mi_scene;   /* opaque, device resident */

/* Select and initialise the HIP device this process renders on (one process per GPU).
 * device < 0: use LOCAL_RANK from the environment, else 0.   replaces: threads_init (include/threads.h:68-130) */
int  mi_init(int device);
/* the device mi_init selected for this process, or -1 before it ran */
int  mi_current_device(void);

/* Upload the scene, build the device layout, allocate + clear the device framebuffer.
 * replaces: the per-module *_init state reachable from work_sample (accel/prims/shader/lights/view). */
int  mi_scene_create(const mi_scene_desc *host, mi_scene **out);

/* Use caller-owned device memory (3*width*height floats) as the framebuffer, e.g. a
 * torch tensor, so that torch.distributed (RCCL) can reduce it in place. NULL = internal. */
int  mi_scene_set_framebuffer(mi_scene *s, float *device_fb);

/* Launch on this HIP stream (a hipStream_t passed as void*); NULL = the backend's own (non-blocking) stream, which is NOT ordered
 * with the device's default stream; MI_STREAM_DEFAULT = the device's default (null) stream itself -- what a caller whose other
 * work (clears, RCCL collectives of a framework running on the null stream) must be ordered with the renders has to pass. */
#define MI_STREAM_DEFAULT ((void *)(intptr_t)-1)
int  mi_scene_set_stream(mi_scene *s, void *hip_stream);

/* Trace path indices [first, first+count) and splat them into the device framebuffer.
 * Stream ordered: returns once all work is queued on the scene's stream (the wavefront pipeline polls its own
 * queue counter while doing so); use mi_sync() or stream order to wait.  replaces: src/view.c:643-645. */
int  mi_render(mi_scene *s, uint64_t first_index, uint64_t count);

/* Block until all queued work of this scene has finished. */
int  mi_sync(mi_scene *s);

/* Copy (accumulate==0) or add (accumulate!=0) the un-normalised device framebuffer into
 * host_fb[3*width*height]. Synchronous.  replaces: the shared mmap'ed fb the workers CAS-add into. */
int  mi_fb_read(mi_scene *s, float *host_fb, int accumulate);

/* Zero the device framebuffer (view_clear_frame, src/view.c:108-126). */
int  mi_fb_clear(mi_scene *s);

/* Raw device pointer of the framebuffer in use. */
float *mi_fb_device_ptr(mi_scene *s);

/* Work counters since creation, same four quantities as the reference's -DACCEL_DEBUG
 * (src/accel.d/qbvhmp.c:83-90): [0] rays (accel_intersect calls), [1] node visits with >=1 box hit,
 * [2] box hits, [3] primitive tests; plus [4] paths, [5] splats, [6] path vertices, [7] deepest traversal stack.
 * Like the reference's, they are a debug facility: only [4] (paths) is counted by default; mi_scene_set_counters(s, 1)
 * (or CORONA_MI_COUNTERS=1 in the environment when the scene is created) selects the counting kernels for the renders that
 * follow -- same results, the ptdl kernel is about 10 % slower with them. mi_intersect and mi_trace_paths always count. */
int  mi_counters(mi_scene *s, uint64_t out[8]);
int  mi_scene_set_counters(mi_scene *s, int enable);

/* The reference BUILD's metal sampler ends 2-4 % of the paths at a rough conductor that its formula does not: in its compiled
 * sample() (gcc -O3 -ffast-math, FMA) the imaginary part of the transmitted cosine comes out as the square root of a rounding
 * error near normal incidence on the microfacet, negative every other time; the NaN is clamped to R = 0 (src/shaders/metal.c:79-157,
 * 219-265; csrc/mi_kernels.h: metal_reference_kills). Off (default): the formula as written, sampling and evaluation consistent
 * (the BSDF battle test passes). On (or CORONA_MI_METAL=reference when the scene is created): the same samples are ended as the
 * reference build ends them -- images of metal scenes then carry the reference's energy, the battle test fails where the
 * reference's own does. (With four wavelengths per path, mi_scene_set_wavelengths, the predicate is applied per component; the MF_COUNT = 4
 * reference's metal plugin is a clang build whose compiled Fresnel term was not examined for the artefact: leave the switch off there.) */
int  mi_scene_set_metal_reference(mi_scene *s, int enable);

/* How a ray walks the tree. Both modes return the same closest hit, bit for bit (distance, primitive, u, v), hence the same paths
 * and images; they differ in the WORK they do for it:
 *   MI_TRAVERSAL_EXACT  the reference's order of operations ray by ray (accel_intersect, src/accel.d/qbvhmp.c:1262-1390): a leaf is
 *                       tested before the next subtree is chosen, so node visits / box hits / primitive tests equal the
 *                       reference's -DACCEL_DEBUG totals. For counter parity and as the yardstick of the fast mode.
 *   MI_TRAVERSAL_FAST   a lane that reaches a leaf puts it aside and goes on descending against the distance known so far; the
 *                       put-aside leaves of the whole wave are tested together. Same hits, path records and images (the rounds keep
 *                       track of what the reference would not have reached, csrc/mi_kernels.h: trace_round_spec); a few per cent
 *                       more node visits and primitive tests (they are counted), fewer idle lane-slots.
 * A scene is created in the mode that is faster for its kernels: FAST for plain pt scenes (2.5 % on regression/0010_pt), EXACT for
 * ptdl and for scenes with media or a moving camera (break even or slower there); CORONA_MI_TRAVERSAL=exact|fast in the environment
 * overrides that, mi_scene_set_traversal changes it later. Scenes with motion-blurred primitives always run the exact
 * rounds. mi_scene_get_traversal returns the mode the next mi_render uses. No reference counterpart. */
#define MI_TRAVERSAL_EXACT 0
#define MI_TRAVERSAL_FAST  1
int  mi_scene_set_traversal(mi_scene *s, int mode);
int  mi_scene_get_traversal(mi_scene *s);

/* Debug/test entry: trace `count` paths starting at `first` and write one mi_path_record per
 * path (no splatting into the framebuffer). Used by the parity tests to compare path by path. */
#define MI_REC_MAX_VERTS 8
#define MI_REC_MAX_SPLATS 8
typedef struct mi_path_vertex
{
  uint64_t prim;
  float    dist;
  float    x[3], n[3], gn[3], omega[3];
  uint32_t mode, flags;
  float    throughput, pdf;
  float    u, v;
  float    rd, rg, em, roughness;
  float    eta;
  int32_t  shader;
} mi_path_vertex;
typedef struct mi_path_splat { int32_t length, tech; float value; float col[3]; } mi_path_splat;
typedef struct mi_path_record
{
  uint64_t index;
  float    pixel_i, pixel_j, lambda, time, scramble, throughput;
  int32_t  length, num_splats;
  mi_path_splat  splat[MI_REC_MAX_SPLATS];
  mi_path_vertex v[MI_REC_MAX_VERTS];
} mi_path_record;
int  mi_trace_paths(mi_scene *s, uint64_t first_index, uint64_t count, mi_path_record *host_out);

/* Hero wavelengths: four wavelengths per path.
 * replaces: the compile-time switch MF_COUNT of the reference (include/mf.h:280-423; `-DMF_COUNT=4` in a build's config): every spectral
 * quantity of a path becomes a vector of four. path_init draws one wavelength per component (src/pathspace.c:218-221); component 0, the hero,
 * decides sampling, Russian roulette and geometry (dielectric.c:293,326-328; pt.c:50); throughputs and pdfs are kept per component (a rough
 * transmission re-derives its half vector per component, dielectric.c:353-411; a specular one keeps one component, :331-343); the MIS weight
 * of a technique is its pdf over the SUM of all techniques' pdfs at all four wavelengths (pt.c:30-38, ptdl.c:78-88); view_splat adds the four
 * colours (src/view.c:455-463, include/spectrum.h:185-195). The same estimator in expectation, less colour noise per path; measured on the reference: the
 * MF_COUNT = 4 build's 512-spp mean of the bench film lies within 0.06 % (pt) / 0.25 % (ptdl) of the scalar build's converged mean
 * (tests/golden/mf4_film_means.json) -- renders with four wavelengths are held against THAT build's mean (bench.py).
 *   mi_scene_set_wavelengths(s, MI_WAVELENGTHS_HERO): the renders and traces that follow run the HERO kernels; (s, 1) goes back.
 * Every scene the backend takes, with either point sampler: the extended kernels (media: the free-flight distance is the hero medium's, transmittance
 * and pdf per component, src/shader.c:76-131; moving camera, moving geometry and emitters) have HERO instantiations too; all of it pinned to per-path dumps
 * of the reference built that way (tests/test_oracle_hero.py). Path i of a hero render is NOT path i of a
 * scalar render: three more numbers are drawn before the camera sample.
 *   mi_trace_paths_hero: mi_trace_paths, plus (ext != NULL) all four components of what the record holds for component 0 -- the layout of
 * the reference-side dump harness' extension block (oracle/refharness/render_dump.c built with -DMF_COUNT=4). */
#define MI_WAVELENGTHS_HERO 4
typedef struct mi_hero_ext
{
  float lambda[MI_WAVELENGTHS_HERO];
  float throughput[MI_REC_MAX_VERTS][MI_WAVELENGTHS_HERO], pdf[MI_REC_MAX_VERTS][MI_WAVELENGTHS_HERO];
  float rd[MI_REC_MAX_VERTS][MI_WAVELENGTHS_HERO], rg[MI_REC_MAX_VERTS][MI_WAVELENGTHS_HERO], em[MI_REC_MAX_VERTS][MI_WAVELENGTHS_HERO], eta[MI_REC_MAX_VERTS][MI_WAVELENGTHS_HERO];
  float splat_value[MI_REC_MAX_SPLATS][MI_WAVELENGTHS_HERO];
} mi_hero_ext;
int  mi_scene_set_wavelengths(mi_scene *s, int count);
int  mi_trace_paths_hero(mi_scene *s, uint64_t first_index, uint64_t count, mi_path_record *host_out, mi_hero_ext *host_ext);

/* Test hook: closest hit of n caller-supplied rays, i.e. accel_intersect (src/accel.d/qbvhmp.c:1262-1390)
 * + prims_intersect (src/prims.c:638-672) on their own. `ignore` is the builder-order index of the primitive
 * the ray starts on (ray_t.ignore, include/corona_common.h) or MI_RAY_NO_IGNORE; `max_dist` initialises hit.dist.
 * u,v are the reference's hit.u / hit.v for triangles and quads; for spheres and lines they are evaluated at
 * shading time in this backend and are not comparable here. Adds to mi_counters like a render. */
#define MI_RAY_NO_IGNORE 0xffffffffu
typedef struct mi_ray { float pos[3], dir[3]; uint32_t ignore; float max_dist; } mi_ray;
typedef struct mi_hit { mi_primid primid; uint32_t prim; float dist, u, v; uint32_t pad[2]; } mi_hit;   /* 32 B */
int  mi_intersect(mi_scene *s, const mi_ray *rays, uint64_t n, mi_hit *host_out);

/* Test hook: the reference's BSDF battle test (tools/battle-test.c:57-266; regression/0052_dielectric, 0053_dielectric) on the
 * kernels' own sample / eval / pdf functions. A synthetic vertex in vacuum with normal (0, 0, +1) (reflect = 1: the reflected
 * hemisphere is tested) or (0, 0, -1) (reflect = 0: the transmitted one), shading rs .06, rd .8, rg 1 and the given roughness; for
 * each of `count` incidence angles out[4k..4k+3] = ebsdf, bsdf, epdf, pdf like the reference prints them: the integral of bsdf cos
 * estimated from spp * size^2 calls of sample() and summed from the evaluation over a size^2 grid, and the same for the pdf.
 * regression/makebattletest.sh:13-14 passes iff (bsdf - ebsdf)^2 < 1e-5 and (pdf - epdf)^2 < 1e-5. The scene only lends its tables. */
typedef struct mi_bsdf_test
{
  uint32_t bsdf;          /* MI_BSDF_DIFFUSE / MI_BSDF_DIELECTRIC / MI_BSDF_METAL */
  float    param[2];      /* dielectric: n_d, Abbe number; metal: table (0 Ti, 1 Cu, 2 Fe, 3 Au, 4 Ag; src/shaders/fresnel.h:21-27) */
  float    roughness;
  uint32_t reflect;
  uint32_t count;         /* incidence angles (the regression tests: 4) */
  float    lambda;        /* nm (the tool: 525) */
  uint32_t size, spp;     /* grid (512) and samples per cell (8) */
} mi_bsdf_test;
int  mi_bsdf_test_run(mi_scene *s, const mi_bsdf_test *t, double *out);

/* Time of the last mi_render launch on the device in milliseconds (HIP events on the scene's
 * stream), and kernel launches since creation. For bench.py's roofline figure. */
int  mi_last_kernel_ms(mi_scene *s, float *ms);
/* number of kernel launches the last mi_render needed: 1 unless the index range exceeds what one launch takes (2^31 paths per
 * workgroup); mi_last_kernel_ms then spans the whole sequence */
int  mi_last_kernel_launches(mi_scene *s, uint64_t *launches);

/* Pixels from path indices, and tile-owned sharding.
 * replaces: the branch of render_sample_path that the reference keeps for tiled rendering (src/render.d/gi.c:88-95: frame = index / (W H),
 * y, x from the rest, pointsampler_mutate_with_pixel -> path_set_pixel, include/pathspace.h:355-360; camera_sample then takes the caller's
 * position instead of two numbers of the point sampler, src/camera.d/thinlens.c:117-118) and the tile scheme of include/render_tiles.h:148-170
 * (32 x 32 pixel tiles handed to workers, src/render_tiles.c:29-88).
 *   mi_scene_set_pixels(s, MI_PIXELS_FROM_INDEX): from now on path i of mi_render / mi_trace_paths starts inside pixel (x, y) = (q mod W, q / W),
 *   q = i mod W H (gi.c:89-92) -- one sample per pixel per W H consecutive indices. MI_PIXELS_SAMPLED (default): the film position is
 *   sampled, as regression/0010_pt renders. Two things differ from the reference's (dead, `#if 0`) branch, both measured with the oracle's
 *   literal restatement of it (tests/test_oracle_golden.py::test_pixels_from_indices_*):
 *     - the position is (x + u, y + v) with u, v the two numbers the sampled mode turns into the film position, not the pixel's corner
 *       (x, y): the film is sampled as densely as in the default mode and the path's other numbers stay what they are;
 *     - the path's generator is seeded through a 64-bit hash (splitmix64) of the index. The reference seeds with 1 + index and ten warm-up
 *       rounds (points_set_state, src/points.d/xorshift128p.c:53-59), which leaves the first numbers of CONSECUTIVE indices correlated
 *       (the wavelengths of paths i and i + 1: r = 0.96). Sampled film positions hide that; with neighbouring indices on neighbouring
 *       pixels whole rows share their wavelengths, and a 256-spp mean of cfg 3 is still 1.5 % off in X and Z.
 *   This mode has no path-for-path reference (the branch never runs in the reference's builds): it is pinned statistically, against the
 *   reference's converged render of the same film (tests/test_gpu_parity.py::test_tile_sharding_statistics_against_the_reference_render).
 *   mi_render_tiles(s, first_frame, frames, member, members): of the paths [first_frame W H, (first_frame + frames) W H) those whose
 *   pixel lies in a tile t = member (mod members), tiles counted row by row. A path has the same index, random numbers and pixel
 *   whichever member renders it: `members` scenes (GPUs) together splat exactly what one mi_render of the range splats, each into
 *   its own pixels (+ the two-pixel rim of the 4 x 4 filter footprint) -- the framebuffers are then summed as with index sharding.
 * Asynchronous like mi_render. */
#define MI_PIXELS_SAMPLED    0
#define MI_PIXELS_FROM_INDEX 1
int  mi_scene_set_pixels(mi_scene *s, int mode);
int  mi_render_tiles(mi_scene *s, uint64_t first_frame, uint64_t frames, uint32_t member, uint32_t members);

/* what the backend made of the scene: out[0] 4-wide nodes, out[1] 1 if the tree is staged in LDS (0: traversed from HBM),
 * out[2] traversal stack entries a ray may need, out[3] 1 if the tree was built on the device (mi_scene_desc.nodes == NULL).
 * No reference counterpart (the reference prints accel statistics to its log, src/accel.d/qbvhmp.c:1121-1144). */
int  mi_scene_stats(mi_scene *s, uint32_t out[4]);
/* how many of the tree's nodes (numbered breadth first: the top of the tree) the kernels stage in LDS: all of them if out[1] of
 * mi_scene_stats is 1, else as many as fit next to the traversal stacks and the material queues' pools -- the rest is read from
 * HBM / L2, one 128-byte record per node visit (qbvh_node_t, src/accel.d/qbvhmp.c:62-81, is what the reference reads per visit).
 * CORONA_MI_NODES_TOP=<n> limits it, CORONA_MI_NODES=global forces 0 (tests). */
int  mi_scene_lds_nodes(mi_scene *s, uint32_t *staged);
/* the name of the kernel instantiation the next mi_render of this scene launches, as a profiler prints it
 * ("mi_path_kernel<false, PTDL, NODES_LDS, HALTON, MEDIA, MB, COUNT, FAST, NORG>"): the library picks it from the scene (media, moving
 * primitives, exterior fog, LDS residency of the tree, traversal mode, debug counters) -- bench.py and tools/profile.sh attach counter
 * profiles to the kernel by this name instead of re-deriving the choice. No reference counterpart. */
/* (round 6) a plain pt scene created under CORONA_MI_WAVEFRONT=1 launches "mi_wave_kernel<false, false, NODES_LDS, HALTON, COUNT>" instead: the same paths through
 * another schedule -- every path an entry of a per-workgroup table, the workgroup's waves take rays and vertices from queues in full batches (csrc/mi_wavefront.h).
 * It replaces the same reference code (the pool dispatch over work_sample, src/view.c:618-645); results do not depend on the choice (tests/test_gpu_wavefront.py). */
int  mi_scene_kernel_name(mi_scene *s, char *buf, size_t len);

void mi_scene_destroy(mi_scene *s);
void mi_shutdown(void);

/* ---------------------------------------------------------------------------------------- several GPUs of one node, one host thread
 * replaces: the worker pool behind view_render (src/view.c:630-695: the progression's path indices are handed out to all workers,
 * which splat into one shared framebuffer) for the GPUs of a node. A group holds one copy of the scene per device; path indices
 * are independent, so mi_group_render gives member k the k-th contiguous share of [first, first + count) with no exchange during
 * rendering; the only exchange is the framebuffer: mi_group_fb_reduce adds the members' framebuffers into member 0's (and clears
 * the others) -- ncclReduce(ncclFloat, ncclSum, 3*W*H, root 0) over xGMI when the devices are distinct (librccl is loaded with
 * dlopen, the library does not link against it), else peer copies to the root's device and an add kernel there.
 * devices == NULL means devices 0 .. n-1; a device may be named twice (two members on one GPU: tests, or two streams of one device).
 * mi_group_scene(g, k) is member k's scene for the per-scene calls (mi_scene_set_traversal, mi_scene_set_counters, mi_last_kernel_ms ...).
 * One process per GPU with an external reduce (bench.py: torch.distributed all_reduce) remains the other way to use several GPUs. */
typedef struct mi_group mi_group;
int  mi_group_create(const mi_scene_desc *desc, const int *devices, int n, mi_group **out);
int  mi_group_size(mi_group *g);
mi_scene *mi_group_scene(mi_group *g, int k);
int  mi_group_uses_rccl(mi_group *g);                                    /* 1: ncclReduce, 0: peer copies + add kernel */
int  mi_group_render(mi_group *g, uint64_t first_index, uint64_t count); /* queues the shares on the members' streams, returns */
/* ... by tiles: member k renders the 32 x 32 tiles t = k (mod n) of frames [first_frame, first_frame + frames) (mi_render_tiles; switches
 * the members to MI_PIXELS_FROM_INDEX) */
int  mi_group_render_tiles(mi_group *g, uint64_t first_frame, uint64_t frames);
int  mi_group_fb_reduce(mi_group *g);                                    /* member 0 += members 1..n-1, which are cleared; stream ordered */
int  mi_group_fb_read(mi_group *g, float *host_fb, int accumulate);      /* reduce, wait, mi_fb_read of member 0 */
int  mi_group_fb_clear(mi_group *g);
int  mi_group_sync(mi_group *g);
int  mi_group_counters(mi_group *g, uint64_t out[8]);
void mi_group_destroy(mi_group *g);

/* planning hook (no device is touched): the kernel launches mi_group_render(first_index, count) queues on a group of `members` devices whose
 * scenes run `grid` resident workgroups each -- the arithmetic mi_group_render and mi_render use themselves: member k's contiguous share
 * (remainder indices to the lowest members, the job counter of src/view.c:618-645 split over devices), cut into launches that keep every
 * workgroup's part below 2^31 path indices (a workgroup hands its part out through a 32-bit counter). Writes at most max_out entries,
 * returns the number of launches. Lets a host without GPUs check a configuration (8 x MI355X, 3840x2160, 1024 spp) before it runs. */
typedef struct mi_launch { int32_t member, grid; uint64_t first, count; } mi_launch;
int  mi_plan_launches(uint64_t first_index, uint64_t count, int members, int grid, mi_launch *out, int max_out);

/* human readable description of the last error on this thread ("" if none) */
const char *mi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
