/* mi_build.h -- 4-wide BVH construction on the device (replaces the host build
 * src/accel.d/qbvhmp.c:425-1144 when the caller hands over no tree, mi_scene_desc.nodes == NULL).
 *
 * LBVH (Karras 2012) + collapse: primitive boxes -> 30-bit Morton codes of the box centres -> radix sort (rocPRIM) ->
 * binary radix tree, one thread per internal node -> bottom-up box refit -> top-down collapse into 4-wide nodes
 * (bl_collapse, one launch per level); binary subtrees of at most MI_BUILD_LEAF primitives become leaves (their
 * primitives are contiguous in sorted order, so a leaf is "first << 5 | count" like the reference's). The output is in
 * the traversal's SoA node layout (mi_device.h), with the split axes and the lower/upper child order the ordered
 * traversal expects (children {0,1} = lower half along axis0, {2,3} = upper half). Primitive records are then gathered
 * into sorted order.
 *
 * Scenes with moving primitives get the nodes' shutter-close boxes too (round 4: box1 / ibox1, DScene.nodes_t1), refitted on the same
 * topology as the reference does (src/accel.d/qbvhmp.c:259-283); the traversal interpolates the two box sets at the ray's time.
 *
 * SAH refinement (round 4, bl_rotate): between refit and collapse the binary tree is improved by tree rotations (Kensler 2008) -- bottom
 * up, every node tries to trade one of its children for a grandchild on the other side and keeps the trade that lowers the SAH cost
 * of what the collapse will make of the subtree the most; a few passes (CORONA_MI_BUILD_SAH, default 2; 0 = the plain LBVH). The reference gets its quality from a binned SAH sweep at build time
 * (src/accel.d/qbvhmp.c:425-525, 854-873); here the Morton order gives the topology in one sort and the rotations repair its worst
 * splits. Rotated subtrees no longer cover contiguous ranges of the Morton order: the collapse hands out the final positions top
 * down (a node's range is split among its children by their primitive counts) and writes the final permutation (perm2).
 *
 * The tree differs from the reference builder's (binned SAH sweep), so node-visit counters differ; closest hits do not
 * depend on the tree (tests/test_gpu_parity.py: same primitive and distance as the oracle on the host-built tree, bit
 * for bit, apart from exact ties).
 */
#ifndef MI_BUILD_H
#define MI_BUILD_H

#include "mi_kernels.h"

#ifndef MI_BUILD_LEAF
#define MI_BUILD_LEAF 2           /* most primitives in a leaf where every lane works through its own leaf (motion-blur kernels): 2 measured best in rounds 1-3
                                     (2/3/4/6/8 tried) and again in round 4: scenes/0059_mb 26.5 ms with 2, 28.8 with 4 */
#endif
#ifndef MI_BUILD_LEAF_JOBS
#define MI_BUILD_LEAF_JOBS 4      /* ... where the primitive tests of a round are dealt out over all lanes (leaf_jobs: every other kernel): regression/0010_pt
                                     15.60 ms with 4 (538 nodes: the tree fits LDS), 15.67 with 2 (985 nodes, read from HBM), 16.0 with the reference's tree;
                                     node visits 1.018 x the reference's count on its own tree (2: 1.089 x), primitive tests 0.73 x (0.44 x) */
#endif
#define BL_BLOCK 256

struct BuildBufs
{
  uint32_t n;                     /* primitives */
  int leaf_max;                   /* most primitives in a leaf */
  float *box;                     /* [n][8]: lo xyz, pad, hi xyz, pad -- in ORIGINAL order */
  uint32_t *key_in, *key, *val_in, *perm;        /* Morton codes and primitive ids, unsorted / sorted */
  int *left, *right, *parent;     /* binary radix tree: children of internal node i (>= 0: internal, < 0: ~leaf position), parents */
  int *leaf_parent;               /* parent of sorted leaf i */
  int *first, *last;              /* sorted range of internal node i (as built: not kept up by the rotations) */
  int *count;                     /* primitives below internal node i */
  float *cost;                    /* SAH cost of the subtree below internal node i (bl_rotate) */
  uint32_t *perm2;                /* final order of the primitives: position -> original primitive (written by the collapse) */
  float *ibox;                    /* [n-1][8] boxes of the internal nodes */
  float *box1, *ibox1;            /* scenes with moving primitives: the same at shutter CLOSE (box / ibox then hold the shutter-open state), else NULL */
  unsigned int *visits;           /* refit arrival counters */
};

__device__ __forceinline__ uint32_t bl_expand(uint32_t v)
{ /* 10 bits -> every third bit */
  v = (v*0x00010001u) & 0xFF0000FFu;
  v = (v*0x00000101u) & 0x0F00F00Fu;
  v = (v*0x00000011u) & 0xC30C30C3u;
  v = (v*0x00000005u) & 0x49249249u;
  return v;
}

__global__ __launch_bounds__(BL_BLOCK) void bl_boxes(BuildBufs b, const DPrim *prims, const DPrimGeo *geo, const DPrimT1 *t1, float3 slo, float3 sinv)
{
  const uint32_t i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i >= b.n) return;
  const DPrim &p = prims[i];
  const float *g = geo[i].f;
  float lo[3], hi[3], lo1[3], hi1[3];
  bool moving = false;
  if(p.type == 0)
  { /* motion-blurred primitive: the record holds the shutter-open vertices, t1 the shutter-close ones. Round 4: one box per state, as the
       reference keeps them in its nodes (aabb0 / aabb1, src/accel.d/qbvhmp.c:62-81,259-283) -- the traversal interpolates them at the ray's
       time (the vertices move linearly, so the interpolated box holds the primitive at every time in between). One box around both
       states, as before, cost +7 / +14 / +33 % node visits / box hits / primitive tests on scenes/0059_mb. */
    moving = true;
    for(int k=0;k<3;k++) { lo[k] = hi[k] = p.v[0][k]; lo1[k] = hi1[k] = t1[i].v[0][k]; }
    for(uint32_t v=0;v<p.pad[0];v++) for(int k=0;k<3;k++)
    {
      const float x0 = p.v[v][k], x1 = t1[i].v[v][k];
      lo[k] = fminf(lo[k], x0); hi[k] = fmaxf(hi[k], x0);
      lo1[k] = fminf(lo1[k], x1); hi1[k] = fmaxf(hi1[k], x1);
    }
    if(p.pad[0] < MI_PRIM_TRI)
    { /* moving sphere / line: pad by the (larger) radius */
      const float r = fmaxf(p.v[2][0], p.v[2][1])*1.0001f + 1e-6f;
      for(int k=0;k<3;k++) { lo[k] -= r; hi[k] += r; lo1[k] -= r; hi1[k] += r; }
    }
  }
  else if(p.type >= MI_PRIM_TRI)
  {
    for(int k=0;k<3;k++) lo[k] = hi[k] = p.v[0][k];
    for(uint32_t v=1;v<p.type;v++) for(int k=0;k<3;k++)
    { const float x = g[26 + 3*(v-1) + k]; lo[k] = fminf(lo[k], x); hi[k] = fmaxf(hi[k], x); }
  }
  else if(p.type == MI_PRIM_SPHERE)
  {
    const float r = p.v[1][0]*1.0001f + 1e-6f;
    for(int k=0;k<3;k++) { lo[k] = p.v[0][k] - r; hi[k] = p.v[0][k] + r; }
  }
  else
  { /* line: both end points, padded by the larger radius */
    const float *f = &p.v[0][0];
    const float r = fmaxf(f[3], f[4])*1.0001f + 1e-6f;
    for(int k=0;k<3;k++) { lo[k] = fminf(f[k], g[26+k]) - r; hi[k] = fmaxf(f[k], g[26+k]) + r; }
  }
  if(!moving) for(int k=0;k<3;k++) { lo1[k] = lo[k]; hi1[k] = hi[k]; }
  if(!b.box1) for(int k=0;k<3;k++) { lo[k] = fminf(lo[k], lo1[k]); hi[k] = fmaxf(hi[k], hi1[k]); }      /* no second box set: one box around both states */
  float *o = b.box + 8*(size_t)i;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = 0.0f;
  if(b.box1)
  {
    float *o1 = b.box1 + 8*(size_t)i;
    o1[0] = lo1[0]; o1[1] = lo1[1]; o1[2] = lo1[2]; o1[3] = 0.0f; o1[4] = hi1[0]; o1[5] = hi1[1]; o1[6] = hi1[2]; o1[7] = 0.0f;
  }
  /* the Morton code from the centre of the primitive's whole motion */
  const float cx = ((fminf(lo[0], lo1[0]) + fmaxf(hi[0], hi1[0]))*0.5f - slo.x)*sinv.x, cy = ((fminf(lo[1], lo1[1]) + fmaxf(hi[1], hi1[1]))*0.5f - slo.y)*sinv.y,
              cz = ((fminf(lo[2], lo1[2]) + fmaxf(hi[2], hi1[2]))*0.5f - slo.z)*sinv.z;
  const uint32_t qx = (uint32_t)fminf(fmaxf(cx*1024.0f, 0.0f), 1023.0f), qy = (uint32_t)fminf(fmaxf(cy*1024.0f, 0.0f), 1023.0f),
                 qz = (uint32_t)fminf(fmaxf(cz*1024.0f, 0.0f), 1023.0f);
  b.key_in[i] = (bl_expand(qx) << 2) | (bl_expand(qy) << 1) | bl_expand(qz);
  b.val_in[i] = i;
}

__device__ __forceinline__ int bl_delta(const uint32_t *key, int n, int i, int j)
{ /* length of the common prefix of the (key, index) pairs i and j; -1 outside the array */
  if(j < 0 || j >= n) return -1;
  const uint32_t a = key[i], c = key[j];
  if(a != c) return __clz(a ^ c);
  return 32 + __clz((uint32_t)i ^ (uint32_t)j);
}

__global__ __launch_bounds__(BL_BLOCK) void bl_hierarchy(BuildBufs b)
{ /* Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", one thread per internal node */
  const int n = (int)b.n;
  const int i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i >= n - 1) return;
  const uint32_t *key = b.key;
  const int d = bl_delta(key, n, i, i+1) - bl_delta(key, n, i, i-1) >= 0 ? 1 : -1;
  const int dmin = bl_delta(key, n, i, i-d);
  int lmax = 2;
  while(bl_delta(key, n, i, i + lmax*d) > dmin) lmax *= 2;
  int l = 0;
  for(int t=lmax/2;t>=1;t/=2) if(bl_delta(key, n, i, i + (l+t)*d) > dmin) l += t;
  const int j = i + l*d;
  const int dnode = bl_delta(key, n, i, j);
  int s = 0;
  for(int t=(l+1)/2;;t=(t+1)/2)
  {
    if(bl_delta(key, n, i, i + (s+t)*d) > dnode) s += t;
    if(t == 1) break;
  }
  const int gamma = i + s*d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int lc = lo == gamma ? ~gamma : gamma;
  const int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  b.left[i] = lc; b.right[i] = rc; b.first[i] = lo; b.last[i] = hi; b.count[i] = hi - lo + 1;
  if(lc >= 0) b.parent[lc] = i; else b.leaf_parent[~lc] = i;
  if(rc >= 0) b.parent[rc] = i; else b.leaf_parent[~rc] = i;
  if(i == 0) b.parent[0] = -1;
}

__device__ __forceinline__ void bl_load_box(const BuildBufs &b, int child, float *lo, float *hi)
{
  /* volatile: internal boxes may have been written by another CU moments ago (bl_refit); bypass the incoherent L1 */
  const volatile float *p = child >= 0 ? b.ibox + 8*(size_t)child : b.box + 8*(size_t)b.perm[~child];
  lo[0] = p[0]; lo[1] = p[1]; lo[2] = p[2]; hi[0] = p[4]; hi[1] = p[5]; hi[2] = p[6];
}

__device__ __forceinline__ void bl_load_box1(const BuildBufs &b, int child, float *lo, float *hi)
{ /* the shutter-close box of a child */
  const volatile float *p = child >= 0 ? b.ibox1 + 8*(size_t)child : b.box1 + 8*(size_t)b.perm[~child];
  lo[0] = p[0]; lo[1] = p[1]; lo[2] = p[2]; hi[0] = p[4]; hi[1] = p[5]; hi[2] = p[6];
}

__global__ __launch_bounds__(BL_BLOCK) void bl_refit(BuildBufs b)
{ /* one thread per leaf walks up; the second thread to arrive at a node forms its box */
  const int i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i >= (int)b.n) return;
  int node = b.leaf_parent[i];
  while(node >= 0)
  {
    __threadfence();
    if(atomicAdd(&b.visits[node], 1u) == 0) return;
    __threadfence();
    float l0[3], h0[3], l1[3], h1[3];
    bl_load_box(b, b.left[node], l0, h0);
    bl_load_box(b, b.right[node], l1, h1);
    volatile float *o = b.ibox + 8*(size_t)node;
    for(int k=0;k<3;k++) { o[k] = fminf(l0[k], l1[k]); o[4+k] = fmaxf(h0[k], h1[k]); }
    if(b.box1)
    { /* the shutter-close boxes are refitted on the same topology (qbvhmp.c:259-283) */
      bl_load_box1(b, b.left[node], l0, h0);
      bl_load_box1(b, b.right[node], l1, h1);
      volatile float *o1 = b.ibox1 + 8*(size_t)node;
      for(int k=0;k<3;k++) { o1[k] = fminf(l0[k], l1[k]); o1[4+k] = fmaxf(h0[k], h1[k]); }
    }
    node = b.parent[node];
  }
}

__device__ __forceinline__ int bl_count(const BuildBufs &b, int child) { return child >= 0 ? b.count[child] : 1; }
__device__ __forceinline__ bool bl_large(const BuildBufs &b, int child)
{ /* internal node with more primitives than a leaf may hold */
  return child >= 0 && b.count[child] > b.leaf_max;
}

/* ---- SAH refinement: tree rotations, bottom up (one thread per leaf walks up; the second thread to arrive at a node works on it, so
 * both subtrees below are final for this pass and nobody else touches them). The cost of a subtree is what the collapse will make of
 * it: a leaf of n primitives (n <= leaf_max) costs ct * n * area, an inner node area + the cost of its two children (areas are not
 * divided by the root's: only differences are compared). At node N with children X and S (S an inner node with children G and K) the
 * trade "X <-> G" leaves N's box as it is and makes S = {X, K}; all (up to four) trades are priced, the best one that lowers the cost
 * is carried out. Scenes with two box sets (moving primitives) price the sum of the areas at shutter open and close. */
struct BlBox { float lo[3], hi[3], lo1[3], hi1[3]; };
__device__ __forceinline__ int bl_ldi(const int *p) { return *(const volatile int *)p; }
__device__ __forceinline__ BlBox bl_box_of(const BuildBufs &b, int child)
{
  BlBox x;
  bl_load_box(b, child, x.lo, x.hi);
  if(b.box1) bl_load_box1(b, child, x.lo1, x.hi1);
  else for(int k=0;k<3;k++) { x.lo1[k] = x.lo[k]; x.hi1[k] = x.hi[k]; }
  return x;
}
__device__ __forceinline__ BlBox bl_union(const BlBox &a, const BlBox &c)
{
  BlBox x;
  for(int k=0;k<3;k++) { x.lo[k] = fminf(a.lo[k], c.lo[k]); x.hi[k] = fmaxf(a.hi[k], c.hi[k]); x.lo1[k] = fminf(a.lo1[k], c.lo1[k]); x.hi1[k] = fmaxf(a.hi1[k], c.hi1[k]); }
  return x;
}
__device__ __forceinline__ float bl_area2(const BlBox &x)
{
  const float dx = x.hi[0]-x.lo[0], dy = x.hi[1]-x.lo[1], dz = x.hi[2]-x.lo[2];
  const float ex = x.hi1[0]-x.lo1[0], ey = x.hi1[1]-x.lo1[1], ez = x.hi1[2]-x.lo1[2];
  return (dx*dy + dy*dz + dz*dx) + (ex*ey + ey*ez + ez*ex);
}
__device__ __forceinline__ void bl_set_parent(const BuildBufs &b, int child, int parent)
{
  if(child >= 0) b.parent[child] = parent; else b.leaf_parent[~child] = parent;
}

__device__ __forceinline__ float bl_cost_of(const BuildBufs &b, int child, const BlBox &box, float ct)
{ /* cost of the subtree below a child of the node under work (its own pass through bl_rotate is over) */
  return child >= 0 ? *(const volatile float *)(b.cost + child) : ct*bl_area2(box);
}
__device__ __forceinline__ float bl_cost_node(const BuildBufs &b, int n, float area, float cl, float cr, float ct)
{
  return n <= b.leaf_max ? ct*(float)n*area : area + cl + cr;
}

__global__ __launch_bounds__(BL_BLOCK) void bl_rotate(BuildBufs b, unsigned int *rotations, float ct)
{
  const int i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i >= (int)b.n) return;
  int node = bl_ldi(b.leaf_parent + i);
  while(node >= 0)
  {
    __threadfence();
    if(atomicAdd(&b.visits[node], 1u) == 0) return;
    __threadfence();
    const int c[2] = { bl_ldi(b.left + node), bl_ldi(b.right + node) };
    const BlBox cb[2] = { bl_box_of(b, c[0]), bl_box_of(b, c[1]) };
    float cc[2] = { bl_cost_of(b, c[0], cb[0], ct), bl_cost_of(b, c[1], cb[1], ct) };
    const int ncount = bl_ldi(b.count + node);
    const float narea = bl_area2(bl_union(cb[0], cb[1]));
    if(ncount > b.leaf_max)
    { /* (a node that ends up inside a leaf has nothing to gain) */
      float best = cc[0] + cc[1];
      const float margin = 1e-6f*best;
      int bx = -1, bg = -1;           /* trade child c[bx] of `node` for grandchild number bg of its sibling c[1-bx] */
      BlBox bbox;
      float bcost = 0.0f;
      int bcount = 0;
      for(int x=0;x<2;x++)
      {
        const int sgl = c[1-x];
        if(sgl < 0) continue;
        const int g[2] = { bl_ldi(b.left + sgl), bl_ldi(b.right + sgl) };
        for(int k=0;k<2;k++)
        { /* X = c[x] takes the place of G = g[k]; K = g[1-k] stays */
          const BlBox gb = bl_box_of(b, g[k]), kb = bl_box_of(b, g[1-k]);
          const BlBox u = bl_union(cb[x], kb);
          const int n2 = (c[x] >= 0 ? bl_ldi(b.count + c[x]) : 1) + (g[1-k] >= 0 ? bl_ldi(b.count + g[1-k]) : 1);
          const float c2 = bl_cost_node(b, n2, bl_area2(u), cc[x], bl_cost_of(b, g[1-k], kb, ct), ct);
          const float total = bl_cost_of(b, g[k], gb, ct) + c2;
          if(total < best - margin) { best = total; bx = x; bg = k; bbox = u; bcost = c2; bcount = n2; }
        }
      }
      if(bx >= 0)
      {
        const int X = c[bx], S = c[1-bx];
        const int g[2] = { bl_ldi(b.left + S), bl_ldi(b.right + S) };
        const int G = g[bg];
        if(bx == 0) b.left[node] = G; else b.right[node] = G;
        if(bg == 0) b.left[S] = X; else b.right[S] = X;
        bl_set_parent(b, G, node);
        bl_set_parent(b, X, S);
        volatile float *o = b.ibox + 8*(size_t)S;
        for(int k=0;k<3;k++) { o[k] = bbox.lo[k]; o[4+k] = bbox.hi[k]; }
        if(b.box1) { volatile float *o1 = b.ibox1 + 8*(size_t)S; for(int k=0;k<3;k++) { o1[k] = bbox.lo1[k]; o1[4+k] = bbox.hi1[k]; } }
        *(volatile int *)(b.count + S) = bcount;
        *(volatile float *)(b.cost + S) = bcost;
        atomicAdd(rotations, 1u);
        /* the node's children are now G and S */
        const BlBox gb = bl_box_of(b, G);
        cc[bx] = bl_cost_of(b, G, gb, ct); cc[1-bx] = bcost;
      }
    }
    *(volatile float *)(b.cost + node) = bl_cost_node(b, ncount, narea, cc[0], cc[1], ct);
    node = bl_ldi(b.parent + node);
  }
}

__device__ __forceinline__ int bl_split_axis(const float *l0, const float *h0, const float *l1, const float *h1, bool &swap)
{ /* axis along which the two boxes' centres are farthest apart; swap = the first box is the upper one */
  float best = -1.0f; int axis = 0; swap = false;
  for(int k=0;k<3;k++)
  {
    const float dlt = (l1[k] + h1[k]) - (l0[k] + h0[k]);
    if(fabsf(dlt) > best) { best = fabsf(dlt); axis = k; swap = dlt < 0.0f; }
  }
  return axis;
}

/* ---- collapse, top down: a 4-wide node takes the two children of its binary node and keeps opening the candidate with
 * the largest surface area until it has four: large subtrees first, then leaves of two primitives (one primitive per slot
 * gets its own box test for free). Folding every second binary level instead left the low levels half empty: 6.6 instead
 * of 6.2 node visits and 2.2 instead of 1.8 primitive tests per ray on regression/0010_pt. Level by level: every node of the current list allocates the indices of its large children and
 * appends them to the next list. Children are ordered by their box centres: the two lower ones along the axis of largest
 * spread go to slots {0,1}, the upper ones to {2,3}, each pair ordered along its own axis of largest separation. */
struct CollapseLists
{
  const int *in;                  /* binary node of every 4-wide node of this level */
  const unsigned int *in_q;       /* its 4-wide index */
  const unsigned int *in_first;   /* first position of its primitives in the final order */
  int *out;
  unsigned int *out_q;
  unsigned int *out_first;
  unsigned int *counters;         /* [0] nodes allocated so far, [1] entries in `out` */
  unsigned int n_in;
};

__device__ __forceinline__ float bl_area(const float *lo, const float *hi)
{
  const float dx = hi[0]-lo[0], dy = hi[1]-lo[1], dz = hi[2]-lo[2];
  return dx*dy + dy*dz + dz*dx;
}

__global__ __launch_bounds__(BL_BLOCK) void bl_collapse(BuildBufs b, CollapseLists L, float4 *nodes, uint32_t *axes, uint32_t stride, float4 *nodes_t1)
{
  const unsigned int t = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(t >= L.n_in) return;
  const int root = L.in[t];
  const uint32_t q = L.in_q[t];
  int cand[4] = { b.left[root], b.right[root], 0, 0 };
  float lo[4][3], hi[4][3];
  bl_load_box(b, cand[0], lo[0], hi[0]);
  bl_load_box(b, cand[1], lo[1], hi[1]);
  int m = 2;
  while(m < 4)
  {
    int best = -1; float barea = -1.0f;
    for(int j=0;j<m;j++) if(bl_large(b, cand[j])) { const float a = bl_area(lo[j], hi[j]); if(a > barea) { barea = a; best = j; } }
    if(best < 0) /* nothing large left: open a small leaf instead, one primitive per slot gets its own box test for free */
      for(int j=0;j<m;j++) if(cand[j] >= 0) { const float a = bl_area(lo[j], hi[j]); if(a > barea) { barea = a; best = j; } }
    if(best < 0) break;
    const int c = cand[best];
    cand[best] = b.left[c]; cand[m] = b.right[c];
    bl_load_box(b, cand[best], lo[best], hi[best]);
    bl_load_box(b, cand[m], lo[m], hi[m]);
    m++;
  }
  /* axis of largest spread of the centres; order the candidates along it */
  float cmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for(int j=0;j<m;j++) for(int k=0;k<3;k++) { const float c = lo[j][k] + hi[j][k]; cmin[k] = fminf(cmin[k], c); cmax[k] = fmaxf(cmax[k], c); }
  int axis0 = 0;
  for(int k=1;k<3;k++) if(cmax[k]-cmin[k] > cmax[axis0]-cmin[axis0]) axis0 = k;
  int order[4] = {0, 1, 2, 3};
  for(int i=1;i<m;i++) for(int j=i;j>0;j--)
  {
    const float ca = lo[order[j-1]][axis0] + hi[order[j-1]][axis0], cb = lo[order[j]][axis0] + hi[order[j]][axis0];
    if(ca > cb) { const int tmp = order[j-1]; order[j-1] = order[j]; order[j] = tmp; }
  }
  /* slots: m = 4: {o0,o1 | o2,o3}; m = 3: {o0,o1 | o2,-}; m = 2: {o0,- | o1,-} */
  int slot[4] = {-1, -1, -1, -1};
  if(m == 4) { slot[0] = order[0]; slot[1] = order[1]; slot[2] = order[2]; slot[3] = order[3]; }
  else if(m == 3) { slot[0] = order[0]; slot[1] = order[1]; slot[2] = order[2]; }
  else { slot[0] = order[0]; slot[2] = order[1]; }
  int axis1[2] = {0, 0};
  for(int h=0;h<2;h++) if(slot[2*h] >= 0 && slot[2*h+1] >= 0)
  {
    bool sw;
    axis1[h] = bl_split_axis(lo[slot[2*h]], hi[slot[2*h]], lo[slot[2*h+1]], hi[slot[2*h+1]], sw);
    if(sw) { const int tmp = slot[2*h]; slot[2*h] = slot[2*h+1]; slot[2*h+1] = tmp; }
  }
  /* final positions: this node's range is dealt out to the candidates by their primitive counts */
  unsigned int cfirst[4] = {0, 0, 0, 0};
  { unsigned int pos = L.in_first[t]; for(int j=0;j<m;j++) { cfirst[j] = pos; pos += (unsigned int)bl_count(b, cand[j]); } }
  uint32_t link[4];
  float olo[4][3], ohi[4][3], olo1[4][3], ohi1[4][3];
  for(int c=0;c<4;c++)
  {
    if(slot[c] < 0)
    { /* empty: inverted box (never entered), empty leaf link */
      for(int k=0;k<3;k++) { olo[c][k] = olo1[c][k] = FLT_MAX; ohi[c][k] = ohi1[c][k] = -FLT_MAX; }
      link[c] = MI_LEAF32;
      continue;
    }
    const int j = slot[c], child = cand[j];
    for(int k=0;k<3;k++) { olo[c][k] = lo[j][k]; ohi[c][k] = hi[j][k]; }
    if(nodes_t1) bl_load_box1(b, child, olo1[c], ohi1[c]);
    if(bl_large(b, child))
    {
      const unsigned int nq = atomicAdd(&L.counters[0], 1u);
      const unsigned int pos = atomicAdd(&L.counters[1], 1u);
      L.out[pos] = child; L.out_q[pos] = nq; L.out_first[pos] = cfirst[j];
      link[c] = nq;
    }
    else
    { /* a leaf: the primitives below `child` take the positions cfirst[j] ... in the final order */
      int stk[8], sp = 0;
      unsigned int k = 0;
      stk[sp++] = child;
      while(sp)
      {
        const int e = stk[--sp];
        if(e < 0) b.perm2[cfirst[j] + k++] = b.perm[~e];
        else { stk[sp++] = b.right[e]; stk[sp++] = b.left[e]; }
      }
      link[c] = MI_LEAF32 | ((uint32_t)cfirst[j] << 5) | k;
    }
  }
  for(int k=0;k<3;k++)
  {
    nodes[(size_t)k*stride + q] = make_float4(olo[0][k], olo[1][k], olo[2][k], olo[3][k]);
    nodes[(size_t)(k+3)*stride + q] = make_float4(ohi[0][k], ohi[1][k], ohi[2][k], ohi[3][k]);
  }
  if(nodes_t1) for(int k=0;k<3;k++)
  {
    nodes_t1[(size_t)k*stride + q] = make_float4(olo1[0][k], olo1[1][k], olo1[2][k], olo1[3][k]);
    nodes_t1[(size_t)(k+3)*stride + q] = make_float4(ohi1[0][k], ohi1[1][k], ohi1[2][k], ohi1[3][k]);
  }
  uint4 lk = make_uint4(link[0], link[1], link[2], link[3]);
  nodes[(size_t)6*stride + q] = *(float4 *)&lk;
  axes[q] = (uint32_t)axis0 | ((uint32_t)axis1[0] << 2) | ((uint32_t)axis1[1] << 4);
}

__global__ __launch_bounds__(BL_BLOCK) void bl_repack(float4 *dst, const float4 *src, uint32_t N, uint32_t stride, uint32_t fields, uint32_t dst_stride)
{ /* [fields][stride] -> the first `fields` lanes of N node records of dst_stride lanes (mi_device.h) */
  const uint32_t i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i < fields*N) dst[(size_t)(i%N)*dst_stride + i/N] = src[(size_t)(i/N)*stride + i%N];
}

template<class T>
__global__ __launch_bounds__(BL_BLOCK) void bl_gather(T *out, const T *in, const uint32_t *perm, uint32_t n)
{
  const uint32_t i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i < n) out[i] = in[perm[i]];
}

__global__ __launch_bounds__(BL_BLOCK) void bl_invert(uint32_t *inv, const uint32_t *perm, uint32_t n)
{
  const uint32_t i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i < n) inv[perm[i]] = i;
}

__global__ __launch_bounds__(BL_BLOCK) void bl_remap(uint32_t *idx, const uint32_t *inv, uint32_t n)
{
  const uint32_t i = blockIdx.x*BL_BLOCK + threadIdx.x;
  if(i < n) idx[i] = inv[idx[i]];
}

#endif
