/* ch_internal.h -- private declarations shared by the host-side C files */
#ifndef CH_INTERNAL_H
#define CH_INTERNAL_H

#include "ch_host.h"

/* flattened geometry of all shapes (include/prims.h:49-83 re-stated without mmap) */
typedef struct ch_geo
{
  uint32_t   num_shapes;
  mi_shape  *shapes;
  uint64_t   num_vtxidx;
  mi_vtxidx *vtxidx;
  uint64_t   num_vtx;
  mi_vtx    *vtx;
} ch_geo;

void ch_prim_bounds(const ch_geo *g, mi_primid pi, float *box6);                        /* enclosing the whole motion */
void ch_prim_bounds_at(const ch_geo *g, mi_primid pi, float *box6, int state);          /* 0 shutter open, 1 shutter close, 2 both */
int  ch_qbvh_build(const ch_geo *g, mi_primid *primid, uint64_t num_prims,
                   mi_node **nodes_out, uint32_t *num_nodes_out, float *aabb6, mi_node_aabb **nodes_t1_out);
float ch_prim_area(const ch_geo *g, mi_primid pi);

/* sigmoid-polynomial spectrum at wavelength lambda [nm] (include/rgb2spec.h:139-149, exact rsqrt) */
float ch_coeff_eval(const float coeff[3], float lambda);
/* spectrum_rgb_to_coeff through a reference-format table (ch_rgb2spec_lut.c): coefficients and scale of a non-black rgb; 0 = ok */
int ch_lut_rgb_to_coeff(const char *lut_path, const float rgb[3], float coeff[3], float *mul_out);

#endif
