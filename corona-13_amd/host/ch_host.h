/* ch_host.h -- plain-C host side of the MI355X path-tracing backend.
 *
 * Re-states (from scratch) the pieces of corona-13 the hot path needs on the host and that
 * the north-star keeps in C: the .nra2 scene parser (src/shader.c:623-760,
 * src/corona_common.c:30-68), the .geo loader (src/prims.c:751-831), the .cam reader and film
 * logic (include/camera.h:153-196, src/view.c:247-394,921-948), the 4-wide QBVH builder
 * (src/accel.d/qbvhmp.c:425-1144), the emitter CDF (src/lights.d/list.c:56-104), RGB->spectrum
 * coefficients (include/spectrum.h:29-38, include/rgb2spec.h:87-128) and PFM output
 * (include/framebuffer.h:142-175).  The result is one mi_scene_desc (include/corona_mi.h) that
 * is handed to the HIP backend through the C ABI.
 */
#ifndef CH_HOST_H
#define CH_HOST_H

#include "corona_mi.h"
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ch_options
{
  uint32_t width, height;       /* requested film size (-w/-h), padded to 32 internally; 0 = 1024x576 */
  uint32_t max_verts;           /* PATHSPACE_MAX_VERTS, 0 = 32                                         */
  uint32_t sampler;             /* MI_SAMPLER_*                                                        */
  uint64_t frame;               /* --frame, 0 = 1                                                      */
  const char *cam_file;         /* -c, NULL = <basename>01.cam                                         */
  const char *rgb2spec_lut;     /* optional reference-format "SPEC" coefficient LUT (data/ergb2spec.coeff);
                                   NULL = solve coefficients directly                                  */
  const char *data_dir;         /* directory with cie1931_xyz.f32 etc; NULL = $CORONA_MI_DATA or next to the library */
  float iso;                    /* --iso override, <= 0 = camera file                                  */
  int   build_threads;          /* reserved                                                            */
  int   verbose;
  uint32_t pointsampler;        /* MI_POINTS_* (MOD_pointsampler: rand or halton)                      */
} ch_options;

typedef struct ch_scene ch_scene;

/* load scene.nra2 (+ .geo, .cam), build the QBVH and all tables. returns 0 on success. */
int  ch_scene_load(const char *nra2_path, const ch_options *opt, ch_scene **out);
const mi_scene_desc *ch_scene_desc(const ch_scene *s);
/* override the rgb2spec coefficients of shader `shader_id` (a `color` line): used by the
 * parity tests to inject the reference's own init-time constants. */
int  ch_scene_set_color_coeff(ch_scene *s, int shader_id, const float coeff[3], float mul);
int  ch_scene_num_shaders(const ch_scene *s);
/* kind string of shader i as written in the scene file ("color", "mult", ...) */
const char *ch_scene_shader_name(const ch_scene *s, int i);
void ch_scene_free(ch_scene *s);

/* gain the reference applies when exporting: view.gain * iso / (100 * spp) (src/view.c:651-657) */
float ch_scene_gain(const ch_scene *s, uint64_t spp);

/* PFM i/o in the reference's layout (rows j=0 first, scale -1.0, header padded to 16 bytes) */
int  ch_pfm_write(const char *filename, const float *fb, uint32_t width, uint32_t height, float gain);
int  ch_pfm_read(const char *filename, float **fb, uint32_t *width, uint32_t *height);
/* tools/img/pfmdiff.c:75-86 metric: sqrt( sum_px (dR^2+dG^2+dB^2) / (W*H) ) */
double ch_pfm_rmse(const float *a, const float *b, uint32_t width, uint32_t height);

/* rgb -> sigmoid polynomial coefficients, returns the scale `mul` (spectrum_rgb_to_coeff) */
float ch_rgb_to_coeff(const float rgb[3], float coeff[3], const char *lut_path);

#ifdef __cplusplus
}
#endif
#endif
