/* oracle_geo.c -- TEST INFRASTRUCTURE: CPU restatement of QBVH traversal and primitive
 * intersection / normals / sampling of the reference (see oracle.h). Scalar C, one child lane
 * at a time where the reference uses SSE; same operation order.
 */
#include "o_core.h"

/* ---------------------------------------------------------------- geometry fetch */
static void o_decode_normal(uint32_t enc, float *vec);
static inline const mi_vtx *o_vtx(const mi_scene_desc *s, mi_primid pi, int k)
{ /* geo_get_vertex, include/geo.h:108-112: shutter-open vertex; a motion-blurred primitive keeps its two states interleaved */
  const mi_shape *sh = s->shapes + MI_PRIMID_SHAPE(pi);
  return s->vtx + sh->vtx_base + (MI_PRIMID_MB(pi)+1)*s->vtxidx[sh->vtxidx_base + MI_PRIMID_VI(pi) + k].v;
}

static inline void o_vertex_time(const mi_scene_desc *s, mi_primid pi, int k, float time, float *out)
{ /* geo_get_vertex_time, include/geo.h:120-138 (mul, mul, add per component as the SSE code does) */
  const mi_vtx *v0 = o_vtx(s, pi, k);
  if(MI_PRIMID_MB(pi)) for(int i=0;i<3;i++) out[i] = (1.0f-time)*v0->v[i] + time*v0[1].v[i];
  else for(int i=0;i<3;i++) out[i] = v0->v[i];
}

static inline mi_vtx o_vtx_at(const mi_scene_desc *s, mi_primid pi, int k, float time)
{ /* position at `time` (geo_get_vertex_time), radius / normal word of the shutter-open vertex (sphere.h:7-11, line.h:10-16) */
  mi_vtx r = *o_vtx(s, pi, k);
  o_vertex_time(s, pi, k, time, r.v);
  return r;
}

static inline void o_normal_time(const mi_scene_desc *s, mi_primid pi, int k, float time, float *n)
{ /* geo_get_normal_time, include/geo.h:152-162 */
  const mi_vtx *v0 = o_vtx(s, pi, k);
  if(MI_PRIMID_MB(pi))
  {
    float n0[3], n1[3];
    o_decode_normal(v0->n, n0);
    o_decode_normal(v0[1].n, n1);
    for(int i=0;i<3;i++) n[i] = (1.0f-time)*n0[i] + time*n1[i];
  }
  else o_decode_normal(v0->n, n);
}

static inline uint32_t o_uvbits(const mi_scene_desc *s, mi_primid pi, int k)
{
  const mi_shape *sh = s->shapes + MI_PRIMID_SHAPE(pi);
  return s->vtxidx[sh->vtxidx_base + MI_PRIMID_VI(pi) + k].uv;
}

static inline float o_bits2float(uint32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline uint32_t o_float2bits(float f) { uint32_t i; memcpy(&i, &f, 4); return i; }

static void o_decode_normal(uint32_t enc, float *vec)
{ /* geo_decode_normal, include/geo.h:24-44: two 16-bit sign+15-bit fixed point octahedral coords */
  const uint16_t p0 = enc & 0xffff, p1 = enc >> 16;
  const uint32_t v0 = 0x3f800000u | ((uint32_t)(p0 & 0x7fff) << 8);
  const uint32_t v1 = 0x3f800000u | ((uint32_t)(p1 & 0x7fff) << 8);
  vec[0] = o_bits2float(o_float2bits(2.0f*o_bits2float(v0) - 2.0f) | ((uint32_t)(p0 & 0x8000) << 16));
  vec[1] = o_bits2float(o_float2bits(2.0f*o_bits2float(v1) - 2.0f) | ((uint32_t)(p1 & 0x8000) << 16));
  vec[2] = 1.0f - (fabsf(vec[0]) + fabsf(vec[1]));
  if(vec[2] < 0.0f)
  {
    const float oldx = vec[0];
    vec[0] = (1.0f - fabsf(vec[1])) * ((oldx < 0.0f) ? -1.0f : 1.0f);
    vec[1] = (1.0f - fabsf(oldx))   * ((vec[1] < 0.0f) ? -1.0f : 1.0f);
  }
  o_normalise(vec);
}

static float o_half2float(uint16_t h)
{ /* half_to_float, include/half.h:57-80 (IEEE binary16 -> binary32) */
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t o = (uint32_t)(h & 0x7fff) << 13;
  const uint32_t exp = 0x0f800000u & o;
  o += (127 - 15) << 23;
  if(exp == 0x0f800000u) o += (128 - 16) << 23;
  else if(exp == 0)
  {
    o += 1 << 23;
    const float f = o_bits2float(o) - o_bits2float(113u << 23);
    o = o_float2bits(f);
  }
  return o_bits2float(o | sign);
}

/* ---------------------------------------------------------------- primitive intersection */
static int o_tri_intersect(const float *v0, const float *v1, const float *v2, mi_primid pi, const o_ray *ray, o_hit *hit)
{ /* geo_tri_intersect, include/geo/triangle.h:263-305: Moeller-Trumbore, no epsilons, no culling */
  if(pi == ray->ignore) return 0;
  float e1[3], e2[3], tv[3], pv[3], qv[3];
  for(int k=0;k<3;k++) { e1[k] = v1[k] - v0[k]; e2[k] = v2[k] - v0[k]; }
  cross3(ray->dir, e2, pv);
  const float det = dot3(e1, pv);
  const float inv_det = 1.0f/det;
  for(int k=0;k<3;k++) tv[k] = ray->pos[k] - v0[k];
  const float v = dot3(tv, pv)*inv_det;
  if(v < 0.0f || v > 1.0f) return 0;
  cross3(tv, e1, qv);
  const float u = dot3(ray->dir, qv)*inv_det;
  if(u < 0.0f || u + v > 1.0f) return 0;
  const float dist = dot3(e2, qv)*inv_det;
  if(dist > ray->min_dist && dist <= hit->dist)
  {
    hit->dist = dist; hit->prim = pi; hit->u = u; hit->v = v;
    return 1;
  }
  return 0;
}

static float o_sphere_t(const float *center, float radius, const o_ray *ray)
{ /* _geo_sphere_intersect, include/geo/sphere.h:112-144 -- with the roundings of the reference BUILD (gcc -O3 -ffast-math with FMA;
     disassembly of prims_intersect in oracle/_ref, +0xb4c..0xd8e): the three dot products are y*y first, then fused x, then
     fused z, and the discriminant is fma(b, b, -(4 a) c). The quadratic cancels badly for a ray from afar (b^2 and 4ac agree in
     their leading digits), the fused forms carry one rounding less where it matters: hit points of the plain-C evaluation lie
     1.8e-5 off the sphere (rms, rays from the camera), the reference's 1.4e-5 -- and a hit point inside the sphere sends its
     grazing rays back into it. With these roundings the distance is the reference's bit for bit on the same ray. */
  const float dx = ray->dir[0], dy = ray->dir[1], dz = ray->dir[2];
  const float ox = ray->pos[0]-center[0], oy = ray->pos[1]-center[1], oz = ray->pos[2]-center[2];
  const float a = fmaf(dz, dz, fmaf(dx, dx, dy*dy));
  const float od = fmaf(dz, oz, fmaf(dx, ox, dy*oy));
  const float b = od + od;
  const float c = fmaf(oz, oz, fmaf(ox, ox, oy*oy)) - radius*radius;
  if(a == 0)
  {
    if(b != 0) return -c/b;
    return -FLT_MAX;
  }
  const float discrim = fmaf(b, b, -((a*4.0f)*c));
  if(discrim < 0) return -FLT_MAX;
  const float sq = sqrtf(discrim);
  const float temp = b < 0 ? -0.5f*(b - sq) : -0.5f*(b + sq);
  const float x0 = temp/a, x1 = c/temp;
  if(x0 <= 0.0f) return x1;
  else if(x1 <= 0.0f) return x0;
  else return fminf(x0, x1);
}

static int o_sphere_intersect(const mi_scene_desc *s, mi_primid pi, const o_ray *ray, o_hit *hit)
{ /* geo_sphere_intersect, include/geo/sphere.h:146-166 (no `ignore` test: relies on the ray offset) */
  const mi_vtx c_ = o_vtx_at(s, pi, 0, ray->time), *c = &c_;
  const float radius = o_bits2float(c->n);
  const float t = o_sphere_t(c->v, radius, ray);
  if(t > ray->min_dist && t < hit->dist)
  {
    hit->dist = t; hit->prim = pi;
    for(int k=0;k<3;k++) hit->x[k] = ray->pos[k] + t*ray->dir[k];
    hit->u = atan2f((hit->x[1]-c->v[1])/radius, (hit->x[0]-c->v[0])/radius)/(2.0f*M_PI);
    hit->v = acosf(OCLAMP((hit->x[2]-c->v[2])/radius, -1.0f, 1.0f))/M_PI;
    return 1;
  }
  return 0;
}

static float o_cylinder_t(const float *v0, const float *v1, float r, const o_ray *ray, float *out, float *len)
{ /* _geo_line_intersect_cylinder, include/geo/line.h:313-398 (r >= 0.01 branch; hair strips are out of scope) */
  float d[3], a[3], b[3], o[3] = {0.0f, 0.0f, 0.0f}, w[3] = {0.0f, 0.0f, 0.0f};
  for(int k=0;k<3;k++) d[k] = v1[k] - v0[k];
  const float dlen = sqrtf(dot3(d, d));
  if(len) *len = dlen;
  for(int k=0;k<3;k++) d[k] *= 1.0f/dlen;
  o_get_onb(d, a, b);
  for(int k=0;k<3;k++)
  {
    o[0] += (ray->pos[k] - v0[k])*d[k];
    o[1] += (ray->pos[k] - v0[k])*a[k];
    o[2] += (ray->pos[k] - v0[k])*b[k];
    w[0] += ray->dir[k]*d[k];
    w[1] += ray->dir[k]*a[k];
    w[2] += ray->dir[k]*b[k];
  }
  const float A = w[1]*w[1] + w[2]*w[2];
  const float B = 2.0f*(o[1]*w[1] + o[2]*w[2]);
  const float C = o[1]*o[1] + o[2]*o[2] - r*r;
  const float discr = (float)((double)(B*B) - 4.0*(double)A*(double)C);   /* `4.0` promotes to double in the reference */
  if(discr < 0.0) return -1.0f;
  const float sq = sqrtf(discr);
  const float temp = B < 0 ? -0.5f*(B - sq) : -0.5f*(B + sq);
  const float t0 = temp/A, t1 = C/temp;
  float t;
  if(t0 <= 0.0f) t = t1;
  else if(t1 <= 0.0f) t = t0;
  else
  {
    t = fminf(t0, t1);
    for(int i=0;i<2;i++)
    {
      for(int k=0;k<3;k++) out[k] = o[k] + t*w[k];
      if(out[0] >= 0.0 && out[0] <= dlen) return t;
      t = fmaxf(t0, t1);
    }
    return -1.0f;
  }
  for(int k=0;k<3;k++) out[k] = o[k] + t*w[k];
  if(out[0] >= 0.0 && out[0] <= dlen) return t;
  return -1.0f;
}

static float o_cone_t(const float *v0, const float *v1, float r0, float r1, const o_ray *ray, float dist, o_hit *hit)
{ /* _geo_line_intersect_cone, include/geo/line.h:401-462 (hit != NULL: ray direction is normalised) */
  float d[3];
  for(int k=0;k<3;k++) d[k] = v1[k] - v0[k];
  const float d_len = sqrtf(dot3(d, d));
  for(int k=0;k<3;k++) d[k] *= 1.0/d_len;
  const float cos_dr = dot3(d, ray->dir);
  const float cos_a2 = d_len*d_len/((r1-r0)*(r1-r0) + d_len*d_len);
  float tip[3], o[3];
  const float tt = -r0*d_len/(r1-r0);
  for(int k=0;k<3;k++) tip[k] = v0[k] + tt*d[k];
  for(int k=0;k<3;k++) o[k] = ray->pos[k] - tip[k];
  const float cos_do = dot3(d, o);
  const float cos_ro = dot3(ray->dir, o);
  const float cos_oo = dot3(o, o);
  const float c2 = cos_dr*cos_dr - cos_a2;
  const float c1 = cos_dr*cos_do - cos_a2*cos_ro;
  const float c0 = cos_do*cos_do - cos_a2*cos_oo;
  float tmin = -1.0f;
  if(fabsf(c2) > 0.0)
  {
    const float discr = c1*c1 - c0*c2;
    if(discr < 0.0f) return -1.0f;
    const float root = sqrtf(discr);
    float x[3];
    for(int i=-1;i<2;i+=2)
    {
      const float t = (-c1 + i*root)/c2;
      if(t > 0.0 && t < dist)
      {
        for(int k=0;k<3;k++) x[k] = ray->pos[k] + t*ray->dir[k] - v0[k];
        const float dt = dot3(x, d);
        if(dt >= 0.0f && dt <= d_len)
        {
          if(hit)
          {
            hit->u = dt/d_len;
            float a[3], b[3];
            o_get_onb(d, a, b);
            hit->v = atan2f(dot3(a, x), dot3(b, x))/(2.0f*M_PI);
          }
          tmin = dist = t;
        }
      }
    }
  }
  return tmin;
}

static int o_line_intersect(const mi_scene_desc *s, mi_primid pi, const o_ray *ray, o_hit *hit)
{ /* geo_line_intersect, include/geo/line.h:464-505 */
  const mi_vtx v0_ = o_vtx_at(s, pi, 0, ray->time), v1_ = o_vtx_at(s, pi, 1, ray->time), *v0 = &v0_, *v1 = &v1_;
  const float r0 = o_bits2float(v0->n), r1 = o_bits2float(v1->n);
  const int linestrip = OMAX(r0, r1) <= 1e-2f;
  if(linestrip && ray->ignore == pi) return 0;
  float out[3], len;
  if(fabsf(r1-r0) < 1e-3)
  {
    const float t = o_cylinder_t(v0->v, v1->v, r0, ray, out, &len);
    if(t > ray->min_dist && t < hit->dist)
    {
      hit->dist = t; hit->prim = pi;
      hit->u = out[0]/len;
      hit->v = atan2f(out[1], out[2])/(2.0f*M_PI);
      return 1;
    }
  }
  else
  {
    const float t = o_cone_t(v0->v, v1->v, r0, r1, ray, hit->dist, hit);
    if((linestrip && t > OMAX(ray->min_dist, 1e-3f)) || (!linestrip && t > ray->min_dist))
    {
      hit->dist = t; hit->prim = pi;
      return 1;
    }
  }
  return 0;
}

static void o_prims_intersect(const mi_scene_desc *s, mi_primid pi, const o_ray *ray, o_hit *hit)
{ /* prims_intersect, src/prims.c:638-672 */
  const uint32_t vcnt = MI_PRIMID_VCNT(pi);
  if(vcnt == MI_PRIM_TRI || vcnt == MI_PRIM_QUAD)
  {
    float v0[3], v1[3], v2[3], v3[3];
    o_vertex_time(s, pi, 0, ray->time, v0); o_vertex_time(s, pi, 1, ray->time, v1); o_vertex_time(s, pi, 2, ray->time, v2);
    if(vcnt == 3) o_tri_intersect(v0, v1, v2, pi, ray, hit);
    else
    {
      if(o_tri_intersect(v0, v1, v2, pi, ray, hit)) { hit->v += hit->u; return; }
      o_vertex_time(s, pi, 3, ray->time, v3);
      if(o_tri_intersect(v0, v2, v3, pi, ray, hit)) hit->u += hit->v;
    }
  }
  else if(vcnt == MI_PRIM_SPHERE) o_sphere_intersect(s, pi, ray, hit);
  else if(vcnt == MI_PRIM_LINE)   o_line_intersect(s, pi, ray, hit);
}

/* ---------------------------------------------------------------- traversal */
void o_accel_intersect(o_ctx *c, const o_ray *ray, o_hit *hit)
{ /* accel_intersect, src/accel.d/qbvhmp.c:1262-1390 (static boxes: the time lerp 1208-1224 is the identity) */
  const mi_scene_desc *s = c->s;
  c->cnt[0]++;
  int near[3], far[3];
  float invdir[3];
  for(int k=0;k<3;k++)
  {
    near[k] = (int)(o_float2bits(ray->dir[k]) >> 31);
    far[k] = 1 ^ near[k];
    invdir[k] = 1.0f/ray->dir[k];
  }
  uint64_t stack[3*100];
  float stack_dist[3*100];
  int sp = 0;
  uint64_t current;
  const mi_node *node = s->nodes;
  /* scenes with motion-blurred primitives carry a second set of boxes (shutter close): every ray tests the boxes interpolated at
     its time, aabb0 (1 - t) + aabb1 t (qbvhmp.c:1208-1224); static scenes test aabb0 as it is (there the lerp is the identity) */
  const float w0 = 1.0f - ray->time, w1 = ray->time;
  while(1)
  {
    float tmin[4];
    int hitm[4], any = 0;
    const mi_node_aabb *nt1 = s->nodes_t1 ? s->nodes_t1 + (node - s->nodes) : 0;
    for(int j=0;j<4;j++)
    { /* aabb_intersect, qbvhmp.c:1188-1246, with SSE min/max semantics (second operand on NaN) */
      float lo = 0.0f, hi = hit->dist;
      for(int k=0;k<3;k++)
      {
        const float b0 = nt1 ? node->aabb[k][j]*w0   + nt1->aabb[k][j]*w1   : node->aabb[k][j];
        const float b1 = nt1 ? node->aabb[k+3][j]*w0 + nt1->aabb[k+3][j]*w1 : node->aabb[k+3][j];
        const float t0 = (b0 - ray->pos[k])*invdir[k];
        const float t1 = (b1 - ray->pos[k])*invdir[k];
        const float mn = t0 < t1 ? t0 : t1;
        const float mx = t0 > t1 ? t0 : t1;
        lo = lo > mn ? lo : mn;
        hi = hi < mx ? hi : mx;
      }
      tmin[j] = lo;
      hitm[j] = lo <= hi;
      any |= hitm[j];
    }
    int popped = 0;
    if(!any) goto pop;
    c->cnt[1]++;
    for(int j=0;j<4;j++) c->cnt[2] += hitm[j];
    {
      /* front-to-back order from the split axes and the ray signs, qbvhmp.c:1313-1320 */
      const int axis0 = node->axis0;
      const int axis1n = near[axis0] ? node->axis01 : node->axis00;
      const int axis1f = near[axis0] ? node->axis00 : node->axis01;
      const int n11 = (far [axis0]<<1) | far [axis1f];
      const int n10 = (far [axis0]<<1) | near[axis1f];
      const int n01 = (near[axis0]<<1) | far [axis1n];
      const int n00 = (near[axis0]<<1) | near[axis1n];
#define PUSH(n) do { stack_dist[sp] = tmin[n]; stack[sp++] = node->child[n]; } while(0)
      if(hitm[n00])
      {
        current = node->child[n00];
        if(hitm[n11]) PUSH(n11);
        if(hitm[n10]) PUSH(n10);
        if(hitm[n01]) PUSH(n01);
      }
      else if(hitm[n01])
      {
        current = node->child[n01];
        if(hitm[n11]) PUSH(n11);
        if(hitm[n10]) PUSH(n10);
      }
      else if(hitm[n10])
      {
        current = node->child[n10];
        if(hitm[n11]) PUSH(n11);
      }
      else current = node->child[n11];
#undef PUSH
      popped = 1;
    }
pop:
    if(!popped)
    {
      do
      {
        if(sp == 0) return;
        sp--;
        current = stack[sp];
      }
      while(stack_dist[sp] > hit->dist);
    }
    while(current & MI_NODE_LEAF)
    {
      uint64_t idx = (current ^ MI_NODE_LEAF) >> 5;
      const uint64_t num = current & 31;
      for(uint64_t i=0;i<num;i++)
      {
        c->cnt[3]++;
        o_prims_intersect(s, s->primid[idx], ray, hit);
        idx++;
      }
      do
      {
        if(sp == 0) return;
        --sp;
        current = stack[sp];
      }
      while(stack_dist[sp] > hit->dist);
    }
    node = s->nodes + current;
  }
}

/* ---------------------------------------------------------------- normals, uv */
static void o_tri_normal(const float *v0, const float *v1, const float *v2, const float *n0, const float *n1, const float *n2,
                         float u, float v, o_hit *hit)
{ /* geo_tri_get_normal, include/geo/triangle.h:63-82 */
  hit->gn[0] = (v1[1]-v0[1])*(v2[2]-v0[2]) - (v1[2]-v0[2])*(v2[1]-v0[1]);
  hit->gn[1] = (v1[2]-v0[2])*(v2[0]-v0[0]) - (v1[0]-v0[0])*(v2[2]-v0[2]);
  hit->gn[2] = (v1[0]-v0[0])*(v2[1]-v0[1]) - (v1[1]-v0[1])*(v2[0]-v0[0]);
  o_normalise(hit->gn);
  const float w = 1.0f - u - v;
  for(int k=0;k<3;k++) hit->n[k] = u*n2[k] + v*n1[k] + w*n0[k];
  o_normalise(hit->n);
}

static void o_decode_uv(uint32_t enc, float *uv)
{ /* geo_decode_uv, include/geo.h:84-89 */
  uv[0] = o_half2float(enc & 0xffff);
  uv[1] = o_half2float(enc >> 16);
}

void o_prims_get_normal(const mi_scene_desc *s, mi_primid pi, o_hit *hit, float time)
{ /* prims_get_normal_time, src/prims.c:254-366 */
  const uint32_t vcnt = MI_PRIMID_VCNT(pi);
  if(vcnt == MI_PRIM_SPHERE)
  { /* geo_sphere_get_normal_time, include/geo/sphere.h:51-62 */
    const mi_vtx c_ = o_vtx_at(s, pi, 0, time), *c = &c_;
    for(int k=0;k<3;k++) hit->gn[k] = hit->x[k] - c->v[k];
    o_normalise(hit->gn);
    memcpy(hit->n, hit->gn, sizeof(float)*3);
  }
  else if(vcnt == MI_PRIM_LINE)
  { /* geo_line_get_normal_time, include/geo/line.h:123-161 */
    const mi_vtx v0_ = o_vtx_at(s, pi, 0, time), v1_ = o_vtx_at(s, pi, 1, time), *v0 = &v0_, *v1 = &v1_;
    const float r0 = o_bits2float(v0->n), r1 = o_bits2float(v1->n);
    if(fabsf(r0-r1) < 1e-3f && r0 < 0.01f)
    {
      for(int k=0;k<3;k++) hit->n[k] = hit->gn[k] = 0.0f;
    }
    else
    {
      float d[3], a[3], b[3];
      for(int k=0;k<3;k++) d[k] = v1->v[k] - v0->v[k];
      const float ilen_d = 1.0f/sqrtf(dot3(d, d));
      for(int k=0;k<3;k++) d[k] *= ilen_d;
      o_get_onb(d, a, b);
      const float phi = 2.0*M_PI*hit->v;
      float sinphi, cosphi;
      sincosf(phi, &sinphi, &cosphi);
      float n[3];
      for(int k=0;k<3;k++) n[k] = a[k]*sinphi + b[k]*cosphi;
      const float rr = r1 - r0;
      if(fabsf(rr) < 1e-3) { for(int k=0;k<3;k++) hit->n[k] = n[k]; }
      else
      {
        for(int k=0;k<3;k++) hit->n[k] = n[k] - d[k]*(r1-r0)*ilen_d;
        o_normalise(hit->n);
      }
      memcpy(hit->gn, hit->n, sizeof(float)*3);
    }
  }
  else
  {
    float n0[3], n1[3], n2[3], n3[3], v0[3], v1[3], v2[3], v3[3];
    o_vertex_time(s, pi, 0, time, v0); o_normal_time(s, pi, 0, time, n0);
    o_vertex_time(s, pi, 2, time, v2); o_normal_time(s, pi, 2, time, n2);
    if(vcnt == 3)
    {
      o_vertex_time(s, pi, 1, time, v1); o_normal_time(s, pi, 1, time, n1);
      o_tri_normal(v0, v1, v2, n0, n1, n2, hit->u, hit->v, hit);
    }
    else if(vcnt == 4)
    {
      if(hit->v >= hit->u)
      {
        o_vertex_time(s, pi, 1, time, v1); o_normal_time(s, pi, 1, time, n1);
        o_tri_normal(v0, v1, v2, n0, n1, n2, hit->u, hit->v - hit->u, hit);
      }
      else
      {
        o_vertex_time(s, pi, 3, time, v3); o_normal_time(s, pi, 3, time, n3);
        o_tri_normal(v0, v2, v3, n0, n2, n3, hit->u - hit->v, hit->v, hit);
      }
    }
  }
  /* texture coordinates, src/prims.c:300-365 */
  if(o_uvbits(s, pi, 0) == 0)
  {
    hit->s = hit->u;
    hit->t = hit->v;
  }
  else
  {
    hit->r = 0.0f;
    float uv0[3], uv1[3], uv2[3], uv3[3];
    if(vcnt == MI_PRIM_SPHERE)
    {
      o_decode_uv(o_uvbits(s, pi, 0), uv0);
      hit->s = hit->u + uv0[0];
      hit->t = hit->v + uv0[1];
    }
    else if(vcnt == MI_PRIM_LINE)
    { /* geo_decode_uvw 11/11/10 fixed point, include/geo.h:91-102 */
      const uint32_t e0 = o_uvbits(s, pi, 0), e1 = o_uvbits(s, pi, 1);
      uv0[0] = (e0 >> 21)/2048.0f; uv0[1] = ((e0 & 0x1ffc00) >> 10)/2048.0f; uv0[2] = (e0 & 0x3ff)/1024.0f;
      uv1[2] = (e1 & 0x3ff)/1024.0f;
      hit->s = uv0[0];
      hit->t = uv0[1];
      hit->r = (1.0f - hit->u)*uv0[2] + hit->u*uv1[2];
    }
    else
    {
      o_decode_uv(o_uvbits(s, pi, 0), uv0);
      o_decode_uv(o_uvbits(s, pi, 2), uv2);
      if(vcnt == 3)
      {
        o_decode_uv(o_uvbits(s, pi, 1), uv1);
        hit->s = (1.0f-hit->u-hit->v)*uv0[0] + hit->v*uv1[0] + hit->u*uv2[0];
        hit->t = (1.0f-hit->u-hit->v)*uv0[1] + hit->v*uv1[1] + hit->u*uv2[1];
      }
      if(vcnt == 4)
      {
        if(hit->v >= hit->u)
        {
          o_decode_uv(o_uvbits(s, pi, 1), uv1);
          const float u = hit->u, v = hit->v - hit->u;
          hit->s = (1.0f-u-v)*uv0[0] + v*uv1[0] + u*uv2[0];
          hit->t = (1.0f-u-v)*uv0[1] + v*uv1[1] + u*uv2[1];
        }
        else
        {
          o_decode_uv(o_uvbits(s, pi, 3), uv3);
          const float u = hit->u - hit->v, v = hit->v;
          hit->s = (1.0f-u-v)*uv0[0] + v*uv2[0] + u*uv3[0];
          hit->t = (1.0f-u-v)*uv0[1] + v*uv2[1] + u*uv3[1];
        }
      }
    }
  }
}

/* ---------------------------------------------------------------- area sampling (ptdl) */
static void o_tri_retime(const float *v0, const float *v1, const float *v2, float u, float v, o_hit *hit)
{ /* geo_tri_retime, include/geo/triangle.h:51-61 */
  const float w = 1.0f - u - v;
  for(int k=0;k<3;k++) hit->x[k] = w*v0[k] + v*v1[k] + u*v2[k];
}

static void o_prims_retime(const mi_scene_desc *s, mi_primid pi, o_hit *hit, float time)
{ /* prims_retime, src/prims.c:178-214 */
  hit->prim = pi;
  const uint32_t vcnt = MI_PRIMID_VCNT(pi);
  if(vcnt == MI_PRIM_SPHERE)
  { /* geo_sphere_retime, include/geo/sphere.h:38-49 + sample_sphere (include/sampler_common.h) */
    const mi_vtx c_ = o_vtx_at(s, pi, 0, time), *c = &c_;
    const float r = o_bits2float(c->n);
    const float x1 = -(cosf(hit->v*M_PI)-1.f)/2.f, x2 = hit->u;
    const float z = 1.f - 2.f*x1, rr = sqrtf(1.f - z*z);
    const float phis = 2.f*M_PI*x2;
    const float d[3] = { rr*cosf(phis), rr*sinf(phis), z };
    for(int k=0;k<3;k++) hit->x[k] = c->v[k] + r*d[k];
  }
  else if(vcnt == MI_PRIM_LINE)
  { /* geo_line_retime, include/geo/line.h:88-121 */
    const mi_vtx v0_ = o_vtx_at(s, pi, 0, time), v1_ = o_vtx_at(s, pi, 1, time), *v0 = &v0_, *v1 = &v1_;
    const float r0 = o_bits2float(v0->n), r1 = o_bits2float(v1->n);
    float y;
    if(fabsf(r1-r0) < 1e-3f) y = hit->u;
    else y = (sqrtf((r1*r1 - r0*r0)*hit->u + r0*r0) - r0)/(r1-r0);
    const float phi = 2.0*M_PI*hit->v;
    float sinphi, cosphi; sincosf(phi, &sinphi, &cosphi);
    float d[3], a[3], b[3];
    for(int k=0;k<3;k++) d[k] = v1->v[k] - v0->v[k];
    const float il = 1.0f/sqrtf(dot3(d, d));
    for(int k=0;k<3;k++) d[k] *= il;
    o_get_onb(d, a, b);
    for(int k=0;k<3;k++) hit->x[k] = v0->v[k] + (v1->v[k] - v0->v[k])*y + a[k]*sinphi + b[k]*cosphi;
  }
  else if(vcnt == MI_PRIM_QUAD)
  {
    float v0[3], v1[3], v2[3];
    o_vertex_time(s, pi, 0, time, v0); o_vertex_time(s, pi, 2, time, v2);
    if(hit->v >= hit->u) { o_vertex_time(s, pi, 1, time, v1); o_tri_retime(v0, v1, v2, hit->u, hit->v - hit->u, hit); }
    else                 { o_vertex_time(s, pi, 3, time, v1); o_tri_retime(v0, v2, v1, hit->u - hit->v, hit->v, hit); }
  }
  else if(vcnt == MI_PRIM_TRI)
  {
    float v0[3], v1[3], v2[3];
    o_vertex_time(s, pi, 0, time, v0); o_vertex_time(s, pi, 1, time, v1); o_vertex_time(s, pi, 2, time, v2);
    o_tri_retime(v0, v1, v2, hit->u, hit->v, hit);
  }
}

void o_prims_sample(const mi_scene_desc *s, mi_primid pi, float r0, float r1, o_hit *hit, float time)
{ /* prims_sample, src/prims.c:216-252 */
  const uint32_t vcnt = MI_PRIMID_VCNT(pi);
  if(vcnt == MI_PRIM_SPHERE) { hit->u = r0; hit->v = acosf(r1)/M_PI; }
  else if(vcnt == MI_PRIM_LINE || vcnt == MI_PRIM_QUAD) { hit->u = r0; hit->v = r1; }
  else if(vcnt == MI_PRIM_TRI)
  {
    const float a = sqrtf(r0);
    hit->u = r1*a;
    hit->v = (1.0f-r1)*a;
  }
  o_prims_retime(s, pi, hit, time);
}

/* ---------------------------------------------------------------- ray bias */
void o_prims_offset_ray(const o_hit *hit, o_ray *ray)
{ /* prims_offset_ray, src/prims.c:374-388 */
  const float eps = OMAX(OMAX(.5f, fabsf(hit->x[0])), OMAX(fabsf(hit->x[1]), fabsf(hit->x[2])))*1e-4f;
  ray->ignore = hit->prim;
  ray->min_dist = 0.0f;
  for(int k=0;k<3;k++) ray->pos[k] = hit->x[k] + eps*ray->dir[k];
}

float o_prims_get_ray(const o_hit *h1, const o_hit *h2, o_ray *ray)
{ /* prims_get_ray, src/prims.c:390-492 (compiled #if 1 branch) */
  const float eps = 1e-4f*OMAX(OMAX(.5f, fabsf(h1->x[0])), OMAX(fabsf(h1->x[1]), fabsf(h1->x[2])));
  ray->ignore = h1->prim;
  ray->min_dist = 0;
  float dir[3];
  for(int k=0;k<3;k++) ray->dir[k] = h2->x[k] - h1->x[k];
  const float ilen = 1.0f/sqrtf(dot3(ray->dir, ray->dir));
  for(int k=0;k<3;k++) ray->dir[k] *= ilen;
  for(int k=0;k<3;k++)
  {
    if(h1->prim == MI_PRIMID_INVALID) ray->pos[k] = h1->x[k];
    else ray->pos[k] = h1->x[k] + eps*ray->dir[k];
    if(h2->prim == MI_PRIMID_INVALID) dir[k] = h2->x[k] - ray->pos[k];
    else dir[k] = h2->x[k] - eps*ray->dir[k] - ray->pos[k];
  }
  return sqrtf(dot3(dir, dir));
}

/* ---------------------------------------------------------------- unit entry point */
void oracle_intersect(const mi_scene_desc *s, const oracle_ray *rays, uint64_t n, oracle_hitrec *out, uint64_t *counters)
{
  o_ctx c;
  memset(&c, 0, sizeof(c));
  c.s = s;
  for(uint64_t i=0;i<n;i++)
  {
    o_ray r;
    memcpy(r.pos, rays[i].pos, 12); memcpy(r.dir, rays[i].dir, 12);
    r.time = 0.0f; r.min_dist = 0.0f; r.ignore = rays[i].ignore;
    o_hit h;
    memset(&h, 0, sizeof(h));
    h.prim = MI_PRIMID_INVALID; h.dist = rays[i].max_dist;
    o_accel_intersect(&c, &r, &h);
    out[i].prim = h.prim; out[i].dist = h.dist; out[i].u = h.u; out[i].v = h.v;
  }
  if(counters) for(int k=0;k<8;k++) counters[k] += c.cnt[k];
}
