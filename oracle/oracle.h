/* oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the reference pt/ptdl hot path.
 *
 * Plain scalar C that follows the reference's algorithm function by function (each function
 * cites the file:line it restates) on the same mi_scene_desc the HIP backend consumes.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (corona-13_amd/) never does.
 *
 * Pinned against the REAL reference: per-path golden records dumped from the reference
 * built in the build container (oracle/_ref, tests/golden/make_golden.py) -- see
 * tests/test_oracle_golden.py.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "corona_mi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* trace path `index` exactly like render_sample_path(index) (src/render.d/gi.c:81-105) with the
 * per-path xorshift128+ generator of src/points.d/xorshift128p.c (tid = 0).
 *   fb  : float[3*W*H] un-normalised framebuffer to splat into, or NULL
 *   rec : record to fill, or NULL
 *   counters[8]: accumulated (same meaning as mi_counters), or NULL */
void oracle_trace_path(const mi_scene_desc *s, uint64_t index, float *fb, mi_path_record *rec, uint64_t *counters);

/* trace [first, first+count) on `threads` pthreads (atomic float adds into fb like the reference's
 * common_atomic_add, include/corona_common.h:316-329). returns seconds spent. */
double oracle_render(const mi_scene_desc *s, uint64_t first, uint64_t count, float *fb, int threads, uint64_t *counters);

/* Pixels from path indices -- the tiled branch of render_sample_path, src/render.d/gi.c:88-95 (`#if 0` in the reference's default build):
 * the pixel of path `index` is q = index mod W H, (x, y) = (q mod W, q / W), handed to camera_sample through path_set_pixel, which then asks the
 * point sampler for no film position (src/camera.d/thinlens.c:117-118). A process-wide switch; affects every entry point.
 *   mode 0: sampled (default).
 *   mode 1: THE REFERENCE'S BRANCH, literally: position = the pixel's corner (x, y), generator seeded as always (points_set_state).
 *   mode 2: MI_PIXELS_FROM_INDEX as the product defines it (corona_mi.h): position (x + u, y + v) with u, v the path's two film numbers, generator
 *           seeded through splitmix64. This is a restatement of the PRODUCT's mode, kept here as the checker of its kernels.
 * PARITY OF BOTH MODES IS UNPINNED path for path: the reference's builds never run the branch, no golden dumps exist. Mode 2 is pinned statistically
 * (tile means against the reference's converged render of the same film); mode 1 serves to show why the product departs from it
 * (tests/test_oracle_golden.py::test_pixels_from_indices_*). */
void oracle_set_pixels_from_index(int mode);
/* ... and of the frames [first_frame, first_frame + frames) the paths whose pixel lies in a 32 x 32 tile t = member (mod members) (tile scheme
 * of include/render_tiles.h:148-170): what mi_render_tiles renders. Switches mode 2 on if the pixels are sampled. Returns seconds. */
double oracle_render_tiles(const mi_scene_desc *s, uint64_t first_frame, uint64_t frames, uint32_t member, uint32_t members, float *fb, int threads, uint64_t *counters);

/* Hero wavelengths: the path tracers as the reference built with -DMF_COUNT=4 runs them (include/mf.h:280-423: four wavelengths per path, geometry and
 * decisions by component 0; src/pathspace.c:215-221, src/sampler.d/pt.c:30-38, ptdl.c:78-88, src/shaders/dielectric.c:240-415). Pinned against per-path
 * dumps of that build (`make -C oracle mf4`, tests/golden/make_golden_mf4.py -> tests/golden/paths_mf4_*.npz, tests/test_oracle_golden.py).
 * out[i]: the record of path first + i with the HERO component of every spectral quantity; ext[i] (or NULL): all four components, the layout of the
 * dump harness' extension block; fb (or NULL): framebuffer to splat into; counters (or NULL): += the hero lane's. */
#define ORACLE_MF 4
typedef struct oracle_hero_ext
{
  float lambda[ORACLE_MF];
  float throughput[MI_REC_MAX_VERTS][ORACLE_MF], pdf[MI_REC_MAX_VERTS][ORACLE_MF];
  float rd[MI_REC_MAX_VERTS][ORACLE_MF], rg[MI_REC_MAX_VERTS][ORACLE_MF], em[MI_REC_MAX_VERTS][ORACLE_MF], eta[MI_REC_MAX_VERTS][ORACLE_MF];
  float splat_value[MI_REC_MAX_SPLATS][ORACLE_MF];
} oracle_hero_ext;
int  oracle_set_reference_ftz(int on);     /* hero lanes: flush denormals like the reference build (see oracle_path.c) */
void oracle_hero_trace(const mi_scene_desc *s, uint64_t first, uint64_t count, mi_path_record *out, oracle_hero_ext *ext, float *fb, uint64_t *counters);
/* the same with n = 4 or 8 wavelengths per path (MF_COUNT = 8: the AVX branch, include/mf.h:22-279; pinned to dumps of that build, tests/test_oracle_hero.py);
 * ext (or NULL): per path the layout of oracle_hero_ext with n columns instead of four = n (1 + 6 MI_REC_MAX_VERTS + MI_REC_MAX_SPLATS) floats */
void oracle_hero_trace_n(const mi_scene_desc *s, int n, uint64_t first, uint64_t count, mi_path_record *out, float *ext, float *fb, uint64_t *counters);

/* fill records for [first, first+count) */
void oracle_trace_records(const mi_scene_desc *s, uint64_t first, uint64_t count, mi_path_record *out);

/* single-function entry points for unit parity tests --------------------------------- */
typedef struct oracle_ray { float pos[3], dir[3]; uint64_t ignore; float max_dist; } oracle_ray;
typedef struct oracle_hitrec { uint64_t prim; float dist, u, v; } oracle_hitrec;
void oracle_intersect(const mi_scene_desc *s, const oracle_ray *rays, uint64_t n, oracle_hitrec *out, uint64_t *counters);
float oracle_rand_sequence(uint64_t index, uint64_t frame, int n, float *out);

#ifdef __cplusplus
}
#endif
#endif
