/* mi_part.hip -- one part of the megakernel's instantiation table (mi_megakernel.h): compiled once per MI_PART = 0..9,
 * MI_PART = PTDL | variant << 1 | FAST << 3 with variant 0 plain, 1 MEDIA ("extended"), 2 MEDIA + MB (motion blur; no FAST rounds),
 * 3 MEDIA without the exchange between waves (NORG: scenes in a scattering exterior medium).
 * MI_PART = 16 | variant << 1 | PTDL: the HERO kernels (four wavelengths per path, mi_hero.h) of the four variants, exact rounds. */
#define MI_PART_DEFINE
#include "mi_megakernel.h"
#include "mi_wavefront.h"

#ifndef MI_PART
#error "compile with -DMI_PART=k"
#endif
#if MI_PART >= 24       /* 24: the wavefront kernel (mi_wavefront.h) of the pt sampler */
template const void *mi_wave_part<(MI_PART & 1) != 0>(unsigned, const PathLaunch *);
#else
#define P_PTDL  ((MI_PART & 1) != 0)
#define P_VAR   ((MI_PART >> 1) & 3)
#define P_FAST  (((MI_PART >> 3) & 1) != 0)
#define P_HERO  ((MI_PART >> 4) != 0)
static_assert(!(P_VAR == 2 && P_FAST), "no such part");
static_assert(!P_HERO || !P_FAST, "no such part");
template const void *mi_path_part<P_PTDL, P_VAR >= 1, P_VAR == 2, P_FAST, P_VAR == 3, P_HERO>(unsigned, const PathLaunch *);
#endif
