/* ch_qbvh.c -- host-side 4-wide BVH builder (binned SAH, kd-style scoring, two-level split per node).
 *
 * Follows the *behaviour* of the reference builder so that the tree -- and with it the
 * traversal work counters the roofline model is defined on -- is the same:
 *   src/accel.d/qbvhmp.c:38-54    constants (7 SAH planes, <= 6 prims per leaf, depth 100)
 *   src/accel.d/qbvhmp.c:343-345  per-primitive cost (quad 16, line 1, else 8)
 *   src/accel.d/qbvhmp.c:425-525  accel_get_split_with_dim: 8 bins over the parent box, every step-th prim
 *   src/accel.d/qbvhmp.c:528-570  partition by box centroid >= split -> right
 *   src/accel.d/qbvhmp.c:875-1022 node_job_work: axis0 then axis00/axis01, degenerate splits, leaves
 *   include/geo/{triangle,sphere,line}.h bounds of the primitive kinds
 * Single threaded and deterministic (the reference's parallel build produces the same tree for
 * scenes below its parallel-sort threshold). Written from the algorithm description, not copied.
 */
#include "ch_internal.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define QB_PLANES      7
#define QB_LEAF_PRIMS  6
#define QB_MAX_DEPTH   100
#define QB_LOG_STEP    3

/* reference macro semantics (NaN falls through to the second operand) */
#define QMAX(a, b) ((a) > (b) ? (a) : (b))
#define QMIN(a, b) ((a) < (b) ? (a) : (b))
#define QCLAMP(a, m, M) QMIN(QMAX(a, m), M)

typedef struct qb_t
{
  const ch_geo *geo;
  mi_primid *primid;     /* permuted in place */
  float     *box;        /* 6 floats per prim, permuted alongside */
  uint64_t   num_prims;
  mi_node   *nodes;
  uint32_t   num_nodes, cap_nodes;
}
qb_t;

static void qb_onb(const float *n, float *u, float *v)
{ /* include/corona_common.h:178-198 */
  if(fabsf(n[1]) < 0.5f) { u[0] = -n[2]; u[1] = 0.0f; u[2] = n[0]; }     /* n x (0,1,0) */
  else                   { u[0] = 0.0f; u[1] = n[2]; u[2] = -n[1]; }     /* n x (1,0,0) */
  const float il = 1.0f/sqrtf(u[0]*u[0]+u[1]*u[1]+u[2]*u[2]);
  for(int k=0;k<3;k++) u[k] *= il;
  v[0] = n[1]*u[2] - u[1]*n[2];
  v[1] = n[2]*u[0] - u[2]*n[0];
  v[2] = n[0]*u[1] - u[0]*n[1];
}

/* bounds of one primitive: `state` 0 = at shutter open (what the reference builds its tree on, prims_get_bounds_shutter_open),
 * 1 = at shutter close (what it refits the second box set to, prims_get_bounds_shutter_close, src/prims.c:41-60),
 * 2 = the box enclosing both (the whole linear motion). A static primitive has one box. */
void ch_prim_bounds_at(const ch_geo *g, mi_primid pi, float *box, int state)
{
  const mi_shape *sh = g->shapes + MI_PRIMID_SHAPE(pi);
  const mi_vtxidx *vi = g->vtxidx + sh->vtxidx_base + MI_PRIMID_VI(pi);
  const mi_vtx *vtx = g->vtx + sh->vtx_base;
  const uint32_t vcnt = MI_PRIMID_VCNT(pi);
  if(MI_PRIMID_MB(pi) && state == 2)
  {
    float b1[6];
    ch_prim_bounds_at(g, pi, box, 0);
    ch_prim_bounds_at(g, pi, b1, 1);
    for(int d=0;d<3;d++) { box[d] = fminf(box[d], b1[d]); box[3+d] = fmaxf(box[3+d], b1[3+d]); }
    return;
  }
  if(vcnt < MI_PRIM_TRI && MI_PRIMID_MB(pi))
  { /* moving sphere / line (vertices interleaved shutter open / close): the static primitive of that state; radii are those of
       the shutter-open vertices (sphere.h:7-11, line.h:10-16) */
    ch_geo one = *g;
    mi_vtxidx idx[2];
    mi_vtx v[2];
    mi_shape sh1 = *sh;
    sh1.vtxidx_base = 0; sh1.vtx_base = 0;
    mi_shape shapes[256];                                   /* the loader admits at most 255 shapes */
    if(MI_PRIMID_SHAPE(pi) > 255) { for(int d=0;d<6;d++) box[d] = 0.0f; return; }
    shapes[MI_PRIMID_SHAPE(pi)] = sh1;
    one.shapes = shapes; one.vtxidx = idx; one.vtx = v;
    const mi_primid still = (pi & ~(1ull << 60)) & ~(0x0fffffffull << 32);     /* same primitive, not moving, vertex index 0 */
    for(uint32_t k=0;k<vcnt;k++)
    {
      idx[k].v = k; idx[k].uv = vi[k].uv;
      v[k] = vtx[2*vi[k].v + (state ? 1 : 0)];
      v[k].n = vtx[2*vi[k].v].n;
    }
    ch_prim_bounds_at(&one, still, box, 0);
    return;
  }
  if(vcnt == MI_PRIM_SPHERE)
  { /* include/geo/sphere.h:16-22 */
    const mi_vtx *c = vtx + vi[0].v;
    float r; memcpy(&r, &c->n, 4);
    for(int d=0;d<3;d++) { box[d] = c->v[d] - r; box[3+d] = c->v[d] + r; }
  }
  else if(vcnt == MI_PRIM_LINE)
  { /* include/geo/line.h:24-38 */
    const mi_vtx *v0 = vtx + vi[0].v, *v1 = vtx + vi[1].v;
    float r0, r1; memcpy(&r0, &v0->n, 4); memcpy(&r1, &v1->n, 4);
    float d[3], a[3], b[3];
    for(int k=0;k<3;k++) d[k] = v1->v[k] - v0->v[k];
    const float il = 1.0f/sqrtf(d[0]*d[0]+d[1]*d[1]+d[2]*d[2]);
    for(int k=0;k<3;k++) d[k] *= il;
    qb_onb(d, a, b);
    for(int dim=0;dim<3;dim++)
    {
      const float theta = atan2f(a[dim], b[dim]);
      const float m = fabsf(sinf(theta)*a[dim]) + fabsf(cosf(theta)*b[dim]);
      box[dim]   = fminf(v0->v[dim] - r0*m, v1->v[dim] - r1*m);
      box[3+dim] = fmaxf(v0->v[dim] + r0*m, v1->v[dim] + r1*m);
    }
  }
  else
  { /* include/geo/triangle.h:7-33; motion-blurred triangles / quads keep their vertices interleaved shutter open / close
       (include/geo.h:108-138) */
    const uint32_t mb = MI_PRIMID_MB(pi), t = (mb && state) ? 1u : 0u;
    for(int d=0;d<3;d++)
    {
      float m = vtx[(mb+1)*vi[0].v + t].v[d], M = m;
      for(uint32_t k=1;k<vcnt;k++)
      {
        const float x = vtx[(mb+1)*vi[k].v + t].v[d];
        m = fminf(x, m); M = fmaxf(x, M);
      }
      box[d] = m; box[3+d] = M;
    }
  }
}

/* the box enclosing the whole motion of one primitive (scene box, device build) */
void ch_prim_bounds(const ch_geo *g, mi_primid pi, float *box) { ch_prim_bounds_at(g, pi, box, 2); }

/* best of 7 equidistant planes along dimension d of box `aabb` for prims [left,right) */
static float qb_split(const qb_t *q, int64_t left, int64_t right, const float *aabb, int d, float *split)
{
  float best = (aabb[d] + aabb[3+d])*0.5f;
  float best_score = FLT_MAX;
  if(right == left) { *split = best; return best_score; }

  int binmin[QB_PLANES+1] = {0}, binmax[QB_PLANES+1] = {0};
  const int p = d == 2 ? 0 : d+1, qd = d == 0 ? 2 : d-1;
  const int64_t step = (int)(log10f(QB_LOG_STEP*(right - left) + 1.0f) + 1.0f);
  const float lo = aabb[d], hi = aabb[3+d];
  const float width = hi - lo;
  for(int64_t k=left;k<right;k+=step)
  {
    const float m = q->box[6*k + d], M = q->box[6*k + 3 + d];
    const int imin = (int)QCLAMP((QB_PLANES+1)*(m - lo)/width, 0, QB_PLANES);
    const int imax = (int)QCLAMP((QB_PLANES+1)*(M - lo)/width, 0, QB_PLANES);
    const uint32_t vcnt = MI_PRIMID_VCNT(q->primid[k]);
    const int cost = vcnt == 4 ? 16 : (vcnt == 2 ? 1 : 8);
    binmin[imin] += cost;
    binmax[imax] += cost;
  }
  int nl = binmin[0], nr = 0;
  const float ep = aabb[3+p] - aabb[p], eq = aabb[3+qd] - aabb[qd];
  const float com = ep*eq;
  for(int i=1;i<QB_PLANES+1;i++) nr += binmax[i];
  for(int k=0;k<QB_PLANES;k++)
  {
    const float splitc = lo + (k+1)*width/(QB_PLANES + 1.0f);
    const float wl = splitc - lo, wr = hi - splitc;
    const float stepl = com + ep*wl + eq*wl;
    const float stepr = com + ep*wr + eq*wr;
    const float score = stepl*nl + stepr*nr;
    if(score < best_score) { best_score = score; best = splitc; }
    nl += binmin[k+1];
    nr -= binmax[k+1];
  }
  *split = best;
  return best_score;
}

/* in-place partition: centroid >= split goes to the right block. returns start of right block,
 * fills tight boxes of both halves. */
static int64_t qb_partition(qb_t *q, int axis, float split, int64_t begin, int64_t back, float *bl, float *br)
{
  int64_t right = back;
  for(int k=0;k<3;k++) { bl[k] = br[k] = FLT_MAX; bl[3+k] = br[3+k] = -FLT_MAX; }
  for(int64_t i=begin;i<right;)
  {
    float *b = q->box + 6*i;
    if(.5f*(b[axis] + b[3+axis]) >= split)
    {
      for(int k=0;k<3;k++) { br[k] = QMIN(br[k], b[k]); br[3+k] = QMAX(br[3+k], b[3+k]); }
      --right;
      const mi_primid t = q->primid[i]; q->primid[i] = q->primid[right]; q->primid[right] = t;
      float tmp[6];
      memcpy(tmp, b, sizeof(tmp)); memcpy(b, q->box + 6*right, sizeof(tmp)); memcpy(q->box + 6*right, tmp, sizeof(tmp));
    }
    else
    {
      for(int k=0;k<3;k++) { bl[k] = QMIN(bl[k], b[k]); bl[3+k] = QMAX(bl[3+k], b[3+k]); }
      i++;
    }
  }
  return right;
}

static uint64_t qb_leaf_link(int64_t first, int64_t count)
{
  return MI_NODE_LEAF | ((uint64_t)first << 5) | ((uint64_t)count & 31u);
}

/* build node `ni` over [left,right) inside box paabb. parent < 0 for the root. */
static void qb_node(qb_t *q, uint32_t ni, int64_t left, int64_t right, const float *paabb, int depth, int32_t parent, int slot)
{
  mi_node *node = q->nodes + ni;
  node->parent = parent;
  const int64_t n = right - left;
  int64_t part[5];
  float box[4][6];
  part[0] = left; part[4] = right;

  int axis0 = 0, axis00 = 0, axis01 = 0;
  float split0, split1l, split1r, split, score;
  float best = qb_split(q, left, right, paabb, 0, &split0);
  if((score = qb_split(q, left, right, paabb, 1, &split)) < best) { best = score; split0 = split; axis0 = 1; }
  if((score = qb_split(q, left, right, paabb, 2, &split)) < best) { best = score; split0 = split; axis0 = 2; }

  if(best < 0.0f && n < 32 && parent >= 0)
  { /* qbvhmp.c:899-912 */
    q->nodes[parent].child[slot] = qb_leaf_link(left, n);
    return;
  }

  part[2] = qb_partition(q, axis0, split0, left, right, box[0], box[1]);

  best = qb_split(q, left, part[2], box[0], 0, &split1l);
  if((score = qb_split(q, left, part[2], box[0], 1, &split)) < best) { best = score; split1l = split; axis00 = 1; }
  if((score = qb_split(q, left, part[2], box[0], 2, &split)) < best) { split1l = split; axis00 = 2; }
  best = qb_split(q, part[2], right, box[1], 0, &split1r);
  if((score = qb_split(q, part[2], right, box[1], 1, &split)) < best) { best = score; split1r = split; axis01 = 1; }
  if((score = qb_split(q, part[2], right, box[1], 2, &split)) < best) { split1r = split; axis01 = 2; }

  float bl[6], br[6];
  part[1] = qb_partition(q, axis00, split1l, left, part[2], bl, br);
  memcpy(box[0], bl, sizeof(bl)); memcpy(box[1], br, sizeof(br));
  part[3] = qb_partition(q, axis01, split1r, part[2], right, box[2], box[3]);

  if(right - part[3] == n || part[3] - part[2] == n || part[2] - part[1] == n || part[1] - left == n)
  { /* one child owns everything, qbvhmp.c:930-969 */
    if(parent < 0 || n > QB_LEAF_PRIMS)
    {
      part[2] = (right + left)/2;
      part[3] = (right + part[2])/2;
      part[1] = (part[2] + left)/2;
      for(int c=0;c<4;c++)
      {
        for(int k=0;k<3;k++) { box[c][k] = FLT_MAX; box[c][3+k] = -FLT_MAX; }
        for(int64_t k=part[c];k<part[c+1];k++) for(int i=0;i<3;i++)
        {
          if(q->box[6*k+i]   < box[c][i])   box[c][i]   = q->box[6*k+i];
          if(q->box[6*k+3+i] > box[c][3+i]) box[c][3+i] = q->box[6*k+3+i];
        }
      }
    }
    else
    {
      q->nodes[parent].child[slot] = qb_leaf_link(left, n);
      return;
    }
  }

  node->axis0 = axis0; node->axis00 = axis00; node->axis01 = axis01;
  for(int k=0;k<6;k++) for(int c=0;c<4;c++) node->aabb[k][c] = box[c][k];

  /* children are numbered consecutively in slot order (qbvhmp.c:975-995) */
  int inner[4], childcnt = 0;
  for(int c=0;c<4;c++)
  {
    inner[c] = !(depth == QB_MAX_DEPTH || part[c+1] - part[c] <= QB_LEAF_PRIMS || q->num_nodes >= q->cap_nodes - 1);
    childcnt += inner[c];
  }
  uint32_t next = q->num_nodes;
  q->num_nodes += childcnt;
  for(int c=0;c<4;c++)
  {
    if(!inner[c]) q->nodes[ni].child[c] = qb_leaf_link(part[c], part[c+1] - part[c]);
    else          q->nodes[ni].child[c] = next++;
  }
  for(int c=0;c<4;c++) if(inner[c])
  {
    float cb[6]; memcpy(cb, box[c], sizeof(cb));
    qb_node(q, (uint32_t)q->nodes[ni].child[c], part[c], part[c+1], cb, depth+1, (int32_t)ni, c);
  }
}

/* the second box set of the reference's nodes (qbvh_node_t.aabb1): a leaf's box bounds the shutter-close state of its primitives
 * (bound_leaf_t1, qbvhmp.c:854-873), an inner child's the four boxes of that child, bottom-up (accel_refit, qbvhmp.c:259-283) */
static void qb_refit_t1(const qb_t *q, mi_node_aabb *t1, uint32_t ni)
{
  const mi_node *node = q->nodes + ni;
  for(int c=0;c<4;c++)
  {
    float *lo[3], *hi[3];
    for(int d=0;d<3;d++) { lo[d] = &t1[ni].aabb[d][c]; hi[d] = &t1[ni].aabb[3+d][c]; *lo[d] = FLT_MAX; *hi[d] = -FLT_MAX; }
    if(node->child[c] & MI_NODE_LEAF)
    {
      const uint64_t first = (node->child[c] ^ MI_NODE_LEAF) >> 5, cnt = node->child[c] & 31u;
      for(uint64_t k=first;k<first+cnt;k++)
      {
        float b[6];
        ch_prim_bounds_at(q->geo, q->primid[k], b, 1);
        for(int d=0;d<3;d++) { *lo[d] = QMIN(*lo[d], b[d]); *hi[d] = QMAX(*hi[d], b[3+d]); }
      }
    }
    else
    {
      const uint32_t ch = (uint32_t)node->child[c];
      qb_refit_t1(q, t1, ch);
      for(int d=0;d<3;d++) { *lo[d] = t1[ch].aabb[d][0]; *hi[d] = t1[ch].aabb[3+d][0]; }
      for(int k=1;k<4;k++) for(int d=0;d<3;d++) { *lo[d] = QMIN(*lo[d], t1[ch].aabb[d][k]); *hi[d] = QMAX(*hi[d], t1[ch].aabb[3+d][k]); }
    }
  }
}

/* builds the tree on the shutter-open boxes like the reference (compute_aabb, qbvhmp.c:1034-1065). aabb = the scene box over the
 * whole motion. nodes_t1_out (may be NULL): the shutter-close boxes, allocated only if a primitive moves (else *nodes_t1_out = NULL) */
int ch_qbvh_build(const ch_geo *g, mi_primid *primid, uint64_t num_prims, mi_node **nodes_out, uint32_t *num_nodes_out, float *aabb,
                  mi_node_aabb **nodes_t1_out)
{
  qb_t q;
  memset(&q, 0, sizeof(q));
  q.geo = g; q.primid = primid; q.num_prims = num_prims;
  q.cap_nodes = (uint32_t)(num_prims > 100 ? num_prims : 100);   /* qbvhmp.c:293 */
  q.nodes = (mi_node *)calloc(q.cap_nodes, sizeof(mi_node));
  q.box = (float *)malloc(sizeof(float)*6*(num_prims ? num_prims : 1));
  if(!q.nodes || !q.box) { free(q.nodes); free(q.box); return MI_ERR_NOMEM; }
  float open[6];
  int moving = 0;
  for(int k=0;k<3;k++) { aabb[k] = open[k] = FLT_MAX; aabb[3+k] = open[3+k] = -FLT_MAX; }
  for(uint64_t i=0;i<num_prims;i++)
  {
    float whole[6];
    ch_prim_bounds_at(g, primid[i], q.box + 6*i, 0);
    ch_prim_bounds_at(g, primid[i], whole, 2);
    if(MI_PRIMID_MB(primid[i])) moving = 1;
    for(int k=0;k<3;k++)
    {
      if(open[k]   > q.box[6*i+k])   open[k]   = q.box[6*i+k];
      if(open[3+k] < q.box[6*i+3+k]) open[3+k] = q.box[6*i+3+k];
      if(aabb[k]   > whole[k])   aabb[k]   = whole[k];
      if(aabb[3+k] < whole[3+k]) aabb[3+k] = whole[3+k];
    }
  }
  q.num_nodes = 1;
  if(num_prims == 0)
  { /* qbvhmp.c:1095-1112: root with four empty leaves */
    mi_node *n = q.nodes;
    for(int c=0;c<4;c++)
    {
      for(int k=0;k<3;k++) { n->aabb[k][c] = FLT_MAX; n->aabb[3+k][c] = -FLT_MAX; }
      n->child[c] = MI_NODE_LEAF;
    }
    n->axis0 = 0; n->axis00 = 1; n->axis01 = 1; n->parent = -1;
  }
  else qb_node(&q, 0, 0, (int64_t)num_prims, open, 0, -1, 0);
  if(nodes_t1_out)
  {
    *nodes_t1_out = NULL;
    if(moving && num_prims)
    {
      mi_node_aabb *t1 = (mi_node_aabb *)calloc(q.num_nodes, sizeof(mi_node_aabb));
      if(!t1) { free(q.nodes); free(q.box); return MI_ERR_NOMEM; }
      qb_refit_t1(&q, t1, 0);
      *nodes_t1_out = t1;
    }
  }
  free(q.box);
  *nodes_out = q.nodes;
  *num_nodes_out = q.num_nodes;
  return MI_OK;
}
