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
