/* mi_device.h -- device-resident scene layout of the MI355X backend (HBM / LDS).
 *
 *  nodes      one 128-byte record per node = ONE cache line of HBM / L2 (round 5; rounds 1-4: field-major SoA, seven lines per visit):
 *               lane 0..2  min x,y,z of the 4 children (16 B each)       lane 3..5  max x,y,z
 *               lane 6     4 child links: bit31 leaf | first_prim<<5 | count, else the child's RECORD OFFSET (below) | the CHILD's split
 *                          axes << 24 (axis0 | axis00<<2 | axis01<<4, MI_AXES_SHIFT: a visit knows its node's axes from the link it
 *                          arrived by -- no second read per visit; baked at upload by mi_bake_links_kernel)
 *               lane 7     pad
 *             (reference qbvh_node_t: 256 B, src/accel.d/qbvhmp.c:62-81). A tree with moving primitives carries the child boxes at shutter
 *             close in the same record: 256 B, lanes 8..13 (same order as 0..5), 14..15 pad.
 *             Nodes are numbered breadth first (renumbered at upload; the device build numbers by level), so "the first K nodes" are the
 *             top of the tree: those are staged into LDS without the pad lanes (112 / 208 B) -- all of them when the tree fits, else as
 *             many as the LDS takes next to stacks and pools; the rest is read from HBM / L2, one (two) line(s) per visit.
 *             RECORD OFFSET of node n, in 16-byte lanes: n < K ? n * SL : K * SL + (n - K) * SH  with SL = 7 (13) lanes in LDS, SH = 8 (16)
 *             in HBM: the LDS address of a staged record is base + 16 * offset (one v_mad_u32_u24, the axes bits fall off), the HBM address
 *             of any record is (nodes + 16 K (SH - SL)) + 16 * offset, "staged" is offset < K * SL -- a visit never multiplies by a record
 *             size. Its seven 16-byte reads then have compile-time offsets from that ONE address; the entry / exit plane of an axis
 *             is picked by adding +-48 bytes per lane. 24 offset bits: 2 M nodes of a tree that is traversed from HBM.
 *  prims      one 64-B record per primitive in builder (leaf) order, pre-resolved at upload so an
 *             intersection test is ONE aligned fetch instead of primid -> vtxidx -> vtx
 *             (src/prims.c:638-672, include/geo.h:120-138)
 *  primgeo    one 176-B record per primitive, touched once per path vertex (decoded normals, line frames, uv, material, primid)
 */
#ifndef MI_DEVICE_H
#define MI_DEVICE_H

#include "corona_mi.h"

#define MI_LEAF32 0x80000000u
#define MI_AXES_SHIFT 24           /* inner link: node index in bits 0..23, the node's split axes in bits 24..29 */
#define MI_NODE_MASK 0x00ffffffu
#define MI_NODE_FIELDS 7           /* 16-byte lanes of a node record in LDS (112 B) */
#define MI_NODE_STRIDE 8           /* ... in HBM (128 B: one cache line per node) */
#define MI_NODE_T1_FIELDS 6        /* lanes of a node's shutter-close boxes: LDS lanes 7..12, HBM lanes 8..13 of a record of 13 / 16 lanes */
#define MI_NODE_T1_LDS 7           /* first shutter-close lane of a record in LDS */
#define MI_NODE_T1_HBM 8           /* ... in HBM */
#define MI_COUNTER_SHARDS 256
/* light_prim[] bit 31: a shadow ray towards this emitter primitive may stop at the FIRST occluder it finds (any-hit) instead of
 * running the closest-hit traversal to its end like the reference's live path_visible (src/pathspace.c:311-344 -> accel_intersect;
 * its accel_visible, src/accel.d/qbvhmp.c:1392-1490, is the any-hit the reference keeps for this). The verdict
 *   visible = closest hit beyond the connection || no hit || closest hit IS the emitter primitive
 * is the same boolean as "some other primitive is hit inside the connection" exactly when the emitter primitive itself cannot be
 * hit inside the (shortened) connection: a triangle or a planar quad is crossed once, at the sampled point, which lies eps
 * beyond the ray's end. Spheres, cones, cylinders, non-planar or moving quads keep the closest-hit traversal (flag clear). */
#define MI_LIGHT_ANYHIT 0x80000000u

struct DPrim                       /* 64 B */
{
  float v[4][3];                   /* tri/quad: v[0] = v0, v[1..3] = edges v1-v0, v2-v0, v3-v0. sphere: v[0] centre, v[1][0] radius.
                                      line: see line_intersect (mi_kernels.h) */
  uint32_t type;                   /* vcnt: 1 sphere, 2 line, 3 tri, 4 quad; 0: tested after the leaf's plain triangles / quads, in leaf order, kind in pad[] */
  uint32_t pad[3];                 /* type 0: pad[0] = vcnt; pad[1] = MI_PRIM_ORDERED for a STATIC triangle / quad (same record as type 3 / 4), else moving */
};
#define MI_PRIM_ORDERED 1u         /* set by mi_mark_ordered_kernel (mi_kernels.h) */

struct DPrimT1                     /* 96 B, motion-blurred triangles / quads only: the shutter-close state of a primitive whose DPrim (type 0,
                                      pad[0] = vertex count) holds the four shutter-open VERTICES instead of v0 + edges. Vertices and normals
                                      are interpolated per ray / hit at the path's time (include/geo.h:120-162) */
{
  float v[4][3];
  float n[4][3];                   /* decoded vertex normals at shutter close (those at shutter open are in DPrimGeo) */
};
#define MI_GEO_MB 8u               /* DPrimGeo.type bit: motion blurred (type & 7 = vertex count) */

struct DPrimGeo                    /* 176 B: everything the shading side needs about one primitive, in one record; the float
                                      constants are precomputed at upload with the kernel's own (host+device) functions */
{
  /* header, fetched first as one 16-B load: it decides the branches and starts the material fetch */
  uint32_t type;                   /* vcnt: 1 sphere, 2 line, 3 tri, 4 quad */
  uint32_t material;
  uint32_t uv0;                    /* raw uv word of vertex 0: 0 = the primitive has no texture coordinates (src/prims.c:300) */
  uint32_t primid_lo;              /* the reference's packed primid (records, medium stack shape id) */
  uint32_t primid_hi;
  uint32_t cls;                    /* class of the material for the exchange between waves (mi_regroup.h): compact index of its bsdf among those the scene uses */
  uint32_t pad[2];
  float f[35];                     /* tri/quad: decoded vertex normals n0..n3 [0..11], geometric normal of (v0 v1 v2) [12..14] and of
                                      (v0 v2 v3) [15..17]. line: unit axis d [0..2], 1/|v1-v0| [3], onb a [4..6], b [7..9] of d;
                                      cone: onb of the intersection-side axis [10..12], [13..15].
                                      [18..25] texture coordinates: tri/quad (s,t) of v0..v3, sphere offset, line (s,t).
                                      [26..34] tri/quad: vertices v1, v2, v3 (DPrim holds v0 and the edges);
                                               line: v1 [26..28], v0 [29..31], r0 [32], r1 [33]; sphere: centre [29..31], radius [32] */
  float pad2;
};
#define MI_GEO_PRIMID(g) ((uint64_t)(g).primid_lo | ((uint64_t)(g).primid_hi << 32))

struct DMaterial                   /* 144 B; the header (one 16-B load) and each 32-B op (two 16-B loads) are fetched whole */
{
  uint32_t bsdf, num_ops;
  float param[2];                  /* dielectric: n_d, abbe; metal: table id */
  mi_shade_op op[MI_MAX_OPS];
};

struct DShapeMedium                /* 48 B per shape: the homogeneous medium filling it (`interior <surface> <medium>`), or med < 0 */
{
  float albedo[4];                 /* rgb2spec coefficients + scale of the medium's colour op in the volume slot: single-scattering albedo */
  float mu_t[4];                   /* coefficients + scale of the extinction coefficient, medium_rgb.c:45-59 */
  float g;                         /* Henyey-Greenstein mean cosine */
  int32_t med;                     /* shader id of the medium (vertex records, interior.shader) or -1 */
  uint32_t pad[2];
};

struct DLight                      /* 160 B: everything next event estimation needs about ONE emitter primitive, when that is a static triangle / quad whose
                                      material is a chain of plain `color` lines (no texture): fetched whole in one burst instead of the chain
                                      emitter list -> primitive record -> shading record -> material -> ops (five dependent round trips).
                                      Scenes with any other emitter run the generic path of the extended kernels. */
{
  float v[4][3];                   /* vertices (prims_sample / prims_retime, src/prims.c:178-252) */
  float n[4][3];                   /* decoded vertex normals (prims_get_normal_time, src/prims.c:254-366) */
  float gn[2][3];                  /* geometric normals of (v0 v1 v2) and (v0 v2 v3) */
  float em_coeff[3], em_mul;       /* the emission the material's prepare chain leaves: em_mul * S(em_coeff, lambda) (color.c:75-82) */
  float roughness;                 /* ... and the roughness its last colour line leaves */
  float L;                         /* light_L[] of this entry (lights_pdf_next_event, src/lights.d/list.c:106-128) */
  uint32_t prim;                   /* builder-order primitive index | MI_LIGHT_ANYHIT (= light_prim[]) */
  uint32_t type;                   /* MI_PRIM_TRI / MI_PRIM_QUAD */
  uint32_t pad[2];
};

struct DCamConst                   /* per-launch camera constants of camera_sample (src/camera.d/thinlens.c:68-128), formed once at
                                      upload with the float / double expressions the kernel would evaluate per path */
{
  float lens_radius, f_dir, f_rg, f_up, pdf_a, pdf_v, sensor, fl2, pdf_av, W, H, Wc, Hc;
};

struct DScene
{
  /* film */
  uint32_t width, height, max_verts, sampler;
  uint64_t frame;
  /* accel */
  uint32_t num_nodes, num_prims;
  const float4  *nodes;            /* [num_nodes][SH] records, SH = MI_NODE_STRIDE lanes (2 MI_NODE_STRIDE with shutter-close boxes) */
  uint32_t nodes_t1;               /* 1: the records carry the child boxes at shutter close (mi_scene_desc.nodes_t1) */
  uint32_t nodes_lds;              /* K: the first K nodes (breadth-first numbering: the top of the tree) are staged into LDS; = num_nodes in the
                                      NODES_LDS instantiations, fewer (or 0) in the others, which read the rest from HBM / L2 */
  uint32_t root_link;              /* link of node 0: its split axes << MI_AXES_SHIFT */
  uint32_t metal_reference;        /* metal sample() ends the paths the reference BUILD's NaN ends (mi_scene_set_metal_reference) */
  const DPrim  *prims;
  const DPrimGeo *primgeo;
  float aabb[6];
  float far_dist;                  /* 2 * largest box extent, src/pathspace.c:867-870 */
  /* materials / lights / camera / tables */
  const DMaterial *materials;
  uint32_t num_lights;
  const uint32_t *light_prim;      /* builder-order primitive index of each emitter prim | MI_LIGHT_ANYHIT */
  const float *light_cdf, *light_L;
  const DLight *lights;            /* [num_lights] or NULL: the one-burst records of next event estimation (plain kernels) */
  float light_cdf4[4];             /* light_cdf[0..3] by value (kernel argument = scalar registers) when there are at most four emitter primitives */
  float p_sky, p_geo, p_vol;
  mi_camera cam;
  DCamConst cc;
  const float *cie_xyz, *checker, *metal_ior;
  /* output */
  float *fb;
  unsigned long long *counters;    /* [MI_COUNTER_SHARDS][8]: same-address atomics serialise at ~11 ns each, so every
                                      workgroup adds into its own shard; mi_counters() sums them */
  /* Halton point sampler (MI_POINTS_HALTON): per dimension {P = digits looked up at once, floor(2^32/P), table offset | groups << 24,
     float bits of the scale}; the digit-permutation tables, concatenated (387 KB, L2 resident) */
  const DPrimT1 *prims_t1;          /* [num_prims] or NULL: shutter-close state of motion-blurred primitives (extended kernels only) */
  const DShapeMedium *shape_medium; /* [num_shapes + 1] (MEDIA kernels only): the medium filling each shape; the last entry is the global
                                       exterior medium (`exterior <medium> 0`, src/shader.c:544-565), med < 0 = vacuum */
  uint32_t exterior_index;          /* = number of shapes */
  const uint4 *halton_dim;
  const unsigned short *halton_perm;
  /* material queues (mi_regroup.h): bytes of LDS behind the job lists that the pools may use, classes in use (< 2: no exchange) */
  uint32_t pool_bytes, pool_classes;
  uint32_t pool_volume_class;       /* extended kernels: the class of volume vertices (= number of surface classes) */
  uint32_t pool_cls_bytes;          /* bytes of the packed class table staged into LDS behind the pools (0: looked up in prim_cls through L2) */
  uint32_t pool_score;              /* 1: among the classes that fill a wave the one whose POOL is fullest is shaded (a scene in a scattering exterior medium: volume vertices
                                       outnumber the surface ones 3 : 1 and a "largest batch" rule never turns to the minority; mi_regroup.h), 0: the largest batch */
  uint32_t wf_list_bytes;           /* wavefront kernel (mi_wavefront.h): bytes of LDS behind ITS job lists for the five lists of entry numbers; its copy of the class table follows */
  const uint32_t *prim_cls;         /* [ceil(num_prims / 16)]: DPrimGeo.cls of every primitive, two bits each (mi_pack_cls_kernel) */
  /* pixels from path indices (mi_scene_set_pixels) and tile-owned sharding (mi_render_tiles): see mi_path.h, tile_path() */
  uint32_t pixels_from_index;       /* 1: path i starts inside pixel (i mod W H) -- the hook of render_sample_path's tiled branch, src/render.d/gi.c:88-95 (mi_scene_set_pixels) */
  uint32_t tile_members;            /* > 0 during mi_render_tiles: the launch enumerates the pixels of the 32 x 32 tiles t = tile_member (mod tile_members) */
  uint32_t tile_member, tiles_local, tiles_x;
  mi_hero_ext *hero_ext;            /* RECORD launches of the HERO kernels (mi_trace_paths_hero): all four components per path, or NULL */
  /* the generator's state after seeding and its ten warm-up rounds, without running them (rng_seed_jump, mi_kernels.h): the state update of xorshift128+ is
     linear over GF(2), so ten rounds of (s0, s1) = (1 + index, 2 + frame) are the XOR of a launch constant (the rounds of (high word of 1 + index, 2 + frame):
     rng_jump_c, set per launch) and four table entries, one per byte of the low word of 1 + index (rng_jump: [4][256] states, 16 KB, the same for every scene) */
  const uint4 *rng_jump;            /* or NULL: run the rounds */
  uint32_t rng_jump_hi;             /* high word of 1 + index the constant was formed for: a lane with another one (a launch across a multiple of 2^32) runs the rounds */
  uint32_t rng_jump_c[4];           /* s0 low, s0 high, s1 low, s1 high */
};

#endif
