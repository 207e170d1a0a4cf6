// lt_prep.hip -- scene preparation on the device (gfx950): see lt_prep.hpp.
//
// Everything here is a restatement, for 64-wide wavefronts, of lt_retree.hpp's host algorithms, and produces the same bytes:
//   * the checks of lt_retree::collect_leaves and of lt_capi.hip's validate_scene, one thread per node;
//   * lt_retree::reference_order in closed form: in a pre-order tree the subtree of node i is the index range [i, end(i)), so the
//     leaves under a node are (end - i + 1) / 2 and a leaf's position in the walk of a direction-sign octant is a sum over its
//     ancestors -- one thread per node walks UP (parents have smaller indices: the walk ends);
//   * lt_retree::build: binned SAH, 32 bins on three axes, top down and level by level.  A range of leaves is split by
//       - many workgroups (more than kChunk leaves: bounds, bins, split and a stable scatter as four launches per level,
//         atomic min / max on order-preserving integers: lt_retree::ordered),
//       - one wavefront looping over its leaves (more than 64), or
//       - one wavefront that finishes the whole subtree with its leaves in registers (at most 64);
//     box unions, bin indices, costs and the first-minimum rule are the host's, so are the stable partitions: same tree;
//   * lt_retree::collapse_wide: the group roots level by level, their numbers by a prefix sum over the nodes.
// No kernel waits for another workgroup; every loop is bounded by a count known when it starts.
#include "lt_prep.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lt_retree.hpp"

namespace lt_prep {

using lt_retree::ordered;
using lt_retree::unordered;

constexpr uint32_t kNone = 0xffffffffu, kRight = 0x80000000u;
constexpr int kBins = lt_retree::kBins;
constexpr uint32_t kChunk = 2048;     // leaves per workgroup of the many-workgroup path; ranges above this take it
constexpr uint32_t kTiny = 64;        // ranges up to this are finished by one wavefront
constexpr int kBinWords = 7;          // ordered lo[3], ordered hi[3], count
constexpr int kRangeBins = 3 * 32 * kBinWords;
constexpr uint32_t kMinIdentity = 0xffffffffu, kMaxIdentity = 0u;   // in the ordered domain

struct Range { uint32_t start, end, node, depth; };
struct SplitRec { int32_t dim, bin; float cmin, scale; uint32_t left, pad0, pad1, pad2; };

// control words on the device
enum { kCtlFlags = 0, kCtlBvhHeight = 1, kCtlOwnHeight = 2, kCtlGroups = 3, kCtlChunks = 4, kCtlWords = 16 };
// per level: ranges for the many-workgroup path, for looping wavefronts, for subtree wavefronts; group roots of the wide collapse
enum { kCntBig = 0, kCntMid = 1, kCntTiny = 2, kCntWide = 3, kCntWords = 4 };

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) {
  for (int m = 32; m; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
  for (int m = 32; m; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) {
  for (int m = 32; m; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
  return v;
}
// a maximum many wavefronts report: one address serves an atomic every ~12 ns, so only a value above what stands there is sent
__device__ __forceinline__ void raise_max(uint32_t* p, uint32_t v) {
  if (__atomic_load_n(p, __ATOMIC_RELAXED) < v) atomicMax(p, v);
}
__device__ __forceinline__ float half_area(const float* lo, const float* hi) {
  const float dx = __fsub_rn(hi[0], lo[0]), dy = __fsub_rn(hi[1], lo[1]), dz = __fsub_rn(hi[2], lo[2]);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dy), __fmul_rn(dy, dz)), __fmul_rn(dz, dx));
}
__device__ __forceinline__ int bin_of(float c, float cmin, float scale) {
  return min(kBins - 1, max(0, (int)__fmul_rn(__fsub_rn(c, cmin), scale)));
}
__device__ __forceinline__ float bin_scale(float d) { return d > 0.0f ? __fdiv_rn((float)kBins, d) : 0.0f; }
__device__ __forceinline__ uint32_t max_child(uint32_t count, int heightLimit, uint32_t depth) {
  const int room = heightLimit - (int)depth - 1;
  if (room >= 31) return count;
  const uint64_t cap = 1ull << (room > 0 ? room : 0);
  return cap < count ? (uint32_t)cap : count;
}

__device__ __forceinline__ float pick(const float* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }
__device__ __forceinline__ int picki(const int* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

struct Leaf { float4 a, b; float lo[3], hi[3], c[3]; };
__device__ __forceinline__ void load_leaf(const float4* __restrict__ nd, uint32_t i, Leaf& l) {
  l.a = nd[2 * (size_t)i];
  l.b = nd[2 * (size_t)i + 1];
  l.lo[0] = l.a.x; l.lo[1] = l.a.y; l.lo[2] = l.a.z;
  l.hi[0] = l.a.w; l.hi[1] = l.b.x; l.hi[2] = l.b.y;
  for (int k = 0; k < 3; k++) l.c[k] = __fadd_rn(__fmul_rn(0.5f, l.lo[k]), __fmul_rn(0.5f, l.hi[k]));
}
__device__ __forceinline__ void store_interior(float4* __restrict__ out, uint32_t node, const float* lo, const float* hi, uint32_t off, int dim) {
  out[2 * (size_t)node] = make_float4(lo[0], lo[1], lo[2], hi[0]);
  out[2 * (size_t)node + 1] = make_float4(hi[1], hi[2], __uint_as_float(off), __uint_as_float((uint32_t)dim << 16));
}

// ------------------------------------------------------------------------------------------ checks, leaf order table
__global__ void k_check_nodes(const float4* __restrict__ nd, uint32_t n, uint32_t n_prims, uint32_t* __restrict__ parent,
                              uint32_t* __restrict__ seen, uint32_t* __restrict__ ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 a = nd[2 * (size_t)i], b = nd[2 * (size_t)i + 1];
  const uint32_t w = __float_as_uint(b.w), cnt = w & 0xffffu, axis = (w >> 16) & 0xffu;
  const int32_t off = (int32_t)__float_as_uint(b.z);
  const float lo[3] = {a.x, a.y, a.z}, hi[3] = {a.w, b.x, b.y};
  uint32_t f = 0;
  if (i == 0)
    for (int k = 0; k < 3; k++)
      if (!(lo[k] > -0x1p+40f && hi[k] < 0x1p+40f && lo[k] <= hi[k])) f |= kFlagNotNested;
  if (cnt != 0) {
    if (off < 0 || (uint32_t)off >= n_prims) f |= kFlagStructure;
    else if (atomicOr(&seen[(uint32_t)off >> 5], 1u << (off & 31)) & (1u << (off & 31))) f |= kFlagDuplicate;
  } else if (i + 1 >= n || off <= (int32_t)i + 1 || (uint32_t)off >= n || axis > 2) {
    f |= kFlagStructure;
  } else {
    const uint32_t kids[2] = {i + 1, (uint32_t)off};
    for (int q = 0; q < 2; q++) {
      const float4 ca = nd[2 * (size_t)kids[q]], cb = nd[2 * (size_t)kids[q] + 1];
      const float clo[3] = {ca.x, ca.y, ca.z}, chi[3] = {ca.w, cb.x, cb.y};
      for (int k = 0; k < 3; k++)
        if (!(clo[k] >= lo[k] && chi[k] <= hi[k] && clo[k] <= chi[k])) f |= kFlagNotNested;
    }
    if (atomicExch(&parent[i + 1], i) != kNone) f |= kFlagNotProper;
    if (atomicExch(&parent[(uint32_t)off], i | kRight) != kNone) f |= kFlagNotProper;
  }
  if (f) atomicOr(&ctl[kCtlFlags], f);
}

__global__ void k_check_prims(const int32_t* __restrict__ prims, uint32_t n_prims, uint32_t n_mats, uint32_t* __restrict__ ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_prims) return;
  const int32_t m = prims[19 * (size_t)i + 18];
  if (m < 0 || (uint32_t)m >= n_mats) atomicOr(&ctl[kCtlFlags], (uint32_t)kFlagPrimitives);
}

// up[i] = (parent word, end, off, axis) of node i, where end is where the subtree of node i ends: a left child ends where its
// sibling starts, a right child where its parent does.  (One 16-byte record per step of the walk up in k_leaf_ranks.)
__global__ void k_subtree_ends(const float4* __restrict__ nd, uint32_t n, const uint32_t* __restrict__ parent, uint4* __restrict__ up,
                               uint32_t* __restrict__ ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t f = 0, x = i, e = i + 1;
  bool ok = true;
  for (int steps = 0; x != 0; steps++) {
    const uint32_t p = parent[x];
    if (p == kNone) { f |= kFlagNotProper; ok = false; break; }
    if (!(p & kRight)) break;
    if (steps >= 64) { f |= kFlagTooDeep; ok = false; break; }
    x = p & ~kRight;
  }
  const float4 b = nd[2 * (size_t)i + 1];
  if (ok) {
    e = x == 0 ? n : __float_as_uint(nd[2 * (size_t)parent[x] + 1].z);
    const bool leaf = (__float_as_uint(b.w) & 0xffffu) != 0u;
    if (leaf ? e != i + 1 : !(__float_as_uint(b.z) < e)) f |= kFlagNotProper;
  }
  up[i] = make_uint4(parent[i], e, __float_as_uint(b.z), (__float_as_uint(b.w) >> 16) & 3u);
  if (f) atomicOr(&ctl[kCtlFlags], f);
}

// For leaves: the depth, the eight positions of lt_retree::reference_order and the leaf's place in the caller's pre-order (= the
// position for the all-positive octant), which is the order the host build starts from.
__global__ void k_leaf_ranks(const float4* __restrict__ nd, uint32_t n, uint32_t n_prims, const uint4* __restrict__ up,
                             uint32_t* __restrict__ rank8, uint32_t* __restrict__ order, uint32_t* __restrict__ ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t depth = 0, f = 0;
  if (i < n) {
    const float4 b = nd[2 * (size_t)i + 1];
    const bool leaf = (__float_as_uint(b.w) & 0xffffu) != 0u;   // (the deepest node is a leaf; k_subtree_ends has seen to it that every node has a parent)
    if (leaf) {
      uint32_t base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t c = i, pw = up[i].x;
      while (c != 0) {
        if (pw == kNone) { f |= kFlagNotProper; break; }
        if (++depth > 64) { f |= kFlagTooDeep; break; }
        const uint32_t p = pw & ~kRight, right = pw >> 31;
        const uint4 rec = up[p];
        // under the right child: the left subtree's leaves come first where the direction is positive on the parent's axis;
        // under the left child: the right subtree's where it is negative
        const uint32_t amount = right ? (rec.z - p) / 2u : (rec.y - rec.z + 1u) / 2u;
        for (uint32_t o = 0; o < 8; o++)
          if (((o >> rec.w) & 1u) != right) base[o] += amount;
        c = p;
        pw = rec.x;
      }
      if (f == 0) {
        const uint32_t prim = __float_as_uint(b.z);
        if (prim < n_prims) {
          uint4* r = (uint4*)(rank8 + 8 * (size_t)prim);
          r[0] = make_uint4(base[0], base[1], base[2], base[3]);
          r[1] = make_uint4(base[4], base[5], base[6], base[7]);
        }
        if (base[0] < (n + 1) / 2) order[base[0]] = i;
      }
    }
  }
  const uint32_t deepest = wave_max_u(f ? 0u : depth);
  if (lane_id() == 0 && deepest) raise_max(&ctl[kCtlBvhHeight], deepest);
  if (f) atomicOr(&ctl[kCtlFlags], f);
}

// ------------------------------------------------------------------------------------------ the split of one range
struct Split { int dim, bin; uint32_t left; };

// One wavefront, all 64 lanes: the host's sweep over 3 x 31 planes as two scans over the bins (a union does not depend on the
// order it is taken in), the cost in the host's arithmetic, the first minimum in (axis, bin) order.
__device__ Split sah_eval(const uint32_t* bins, const float* d, uint32_t maxChild) {
  const uint32_t lane = lane_id();
  float bestCost = 3.402823466e+38f;
  uint32_t bestIdx = kNone;
  for (int a = 0; a < 3; a++) {
    if (!(d[a] > 0.0f)) continue;
    uint32_t lo[3] = {kMinIdentity, kMinIdentity, kMinIdentity}, hi[3] = {kMaxIdentity, kMaxIdentity, kMaxIdentity}, cnt = 0;
    if (lane < 32) {
      const uint32_t* b = bins + (a * 32 + lane) * kBinWords;
      for (int k = 0; k < 3; k++) { lo[k] = b[k]; hi[k] = b[3 + k]; }
      cnt = b[6];
    }
    uint32_t pl[3], ph[3], pc = cnt, sl[3], sh[3], sc = cnt;
    for (int k = 0; k < 3; k++) { pl[k] = sl[k] = lo[k]; ph[k] = sh[k] = hi[k]; }
    for (int s = 1; s < 32; s <<= 1) {
      const bool up = lane >= (uint32_t)s, down = lane + s < 64;
      for (int k = 0; k < 3; k++) {
        const uint32_t ul = (uint32_t)__shfl_up((int)pl[k], s), uh = (uint32_t)__shfl_up((int)ph[k], s);
        const uint32_t dl = (uint32_t)__shfl_down((int)sl[k], s), dh = (uint32_t)__shfl_down((int)sh[k], s);
        if (up) { pl[k] = min(pl[k], ul); ph[k] = max(ph[k], uh); }
        if (down) { sl[k] = min(sl[k], dl); sh[k] = max(sh[k], dh); }
      }
      const uint32_t uc = (uint32_t)__shfl_up((int)pc, s), dc = (uint32_t)__shfl_down((int)sc, s);
      if (up) pc += uc;
      if (down) sc += dc;
    }
    // the plane behind bin `lane`: the bins up to it against the bins behind it (lanes 32.. hold no bins: identities)
    uint32_t rl[3], rh[3];
    for (int k = 0; k < 3; k++) { rl[k] = (uint32_t)__shfl_down((int)sl[k], 1); rh[k] = (uint32_t)__shfl_down((int)sh[k], 1); }
    const uint32_t rc = (uint32_t)__shfl_down((int)sc, 1);
    bool valid = lane < (uint32_t)(kBins - 1) && pc != 0 && rc != 0 && pc <= maxChild && rc <= maxChild;
    float cost = __builtin_inff();
    if (valid) {
      float flo[3], fhi[3], glo[3], ghi[3];
      for (int k = 0; k < 3; k++) { flo[k] = unordered(pl[k]); fhi[k] = unordered(ph[k]); glo[k] = unordered(rl[k]); ghi[k] = unordered(rh[k]); }
      cost = __fadd_rn(__fmul_rn(half_area(flo, fhi), (float)pc), __fmul_rn(half_area(glo, ghi), (float)rc));
    }
    uint32_t idx = (uint32_t)a * 32u + lane;
    for (int m = 32; m; m >>= 1) {
      const float oc = __shfl_xor(cost, m);
      const uint32_t oi = (uint32_t)__shfl_xor((int)idx, m);
      if (oc < cost || (oc == cost && oi < idx)) { cost = oc; idx = oi; }
    }
    if (cost < bestCost) { bestCost = cost; bestIdx = idx; }
  }
  Split s{-1, -1, 0u};
  if (bestIdx != kNone) {
    s.dim = (int)(bestIdx >> 5);
    s.bin = (int)(bestIdx & 31u);
    const uint32_t c = lane <= (uint32_t)s.bin && lane < 32 ? bins[(s.dim * 32 + lane) * kBinWords + 6] : 0u;
    s.left = wave_sum_u(c);
  }
  return s;
}

__device__ __forceinline__ void clear_bins(uint32_t* bins) {
  for (uint32_t i = lane_id(); i < (uint32_t)kRangeBins; i += 64) {
    const uint32_t k = i % kBinWords;
    bins[i] = k < 3 ? kMinIdentity : kMaxIdentity;   // (k == 6, the count: 0 as well)
  }
}
__device__ __forceinline__ void bin_leaf(uint32_t* bins, const Leaf& l, const float* cmin, const float* scale, const float* d) {
  for (int a = 0; a < 3; a++) {
    if (!(d[a] > 0.0f)) continue;
    uint32_t* b = bins + (a * 32 + bin_of(l.c[a], cmin[a], scale[a])) * kBinWords;
    for (int k = 0; k < 3; k++) { atomicMin(&b[k], ordered(l.lo[k])); atomicMax(&b[3 + k], ordered(l.hi[k])); }
    atomicAdd(&b[6], 1u);
  }
}
__device__ __forceinline__ int largest_extent(const float* d) { return (d[0] > d[1] && d[0] > d[2]) ? 0 : (d[1] > d[2] ? 1 : 2); }

// Where a child range goes: its list for the next level (called by one lane).  A single leaf is written by whoever placed it.
__device__ void emit_child(Range c, Range* __restrict__ nextBig, Range* __restrict__ nextMid, Range* __restrict__ nextTiny,
                           uint32_t* __restrict__ nextCounts) {
  const uint32_t count = c.end - c.start;
  if (count < 2) return;
  if (count <= kTiny) nextTiny[atomicAdd(&nextCounts[kCntTiny], 1u)] = c;
  else if (count <= kChunk) nextMid[atomicAdd(&nextCounts[kCntMid], 1u)] = c;
  else nextBig[atomicAdd(&nextCounts[kCntBig], 1u)] = c;
}

// ------------------------------------------------------------------------------------------ at most 64 leaves: the whole subtree
// One wavefront, its leaves in registers, lane = position: a range of the subtree is a run of lanes [s, e), and what a lane keeps
// beside its leaf is the run it stands in (s, e, the node the run will become, its depth).  A stable partition moves the leaves
// between lanes (ds_permute); the runs stay where they are and are cut in two.  Two ways to split:
//   * a run of more than kSmall leaves: one at a time, with wavefront-wide reductions and the 32 bins in LDS (sah_eval);
//   * ALL runs of at most kSmall leaves at once -- seven in eight of a subtree's nodes --, every lane pricing the planes behind its
//     own leaf's bins against the leaves of its own run, fetched lane by lane (ds_bpermute from s + j).  A plane that matters is
//     the plane behind some leaf's bin: any other plane has the same two sides as the nearest such plane below it, hence the
//     same cost, and the host's sweep keeps the first of equals.
constexpr uint32_t kSmall = 16;

__device__ __forceinline__ uint32_t from_lane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane << 2), (int)v); }
__device__ __forceinline__ uint32_t to_lane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_ds_permute((int)(lane << 2), (int)v); }
__device__ __forceinline__ float to_lane_f(float v, uint32_t lane) { return __uint_as_float(to_lane(__float_as_uint(v), lane)); }

__global__ __launch_bounds__(256) void k_tiny(const float4* __restrict__ nd, const uint32_t* __restrict__ order, const Range* __restrict__ list,
                                               const uint32_t* __restrict__ counts, float4* __restrict__ out, int heightLimit,
                                               uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s_bins[4][kRangeBins];
  const uint32_t wave = threadIdx.x >> 6, lane = lane_id(), w = blockIdx.x * 4 + wave;
  if (w >= counts[kCntTiny]) return;
  const Range rg = list[w];
  const uint32_t n = rg.end - rg.start;
  Leaf l;
  uint32_t leaf = lane < n ? order[rg.start + lane] : 0u;
  load_leaf(nd, leaf, l);
  uint32_t olo[3], ohi[3], oc[3];
  for (int k = 0; k < 3; k++) { olo[k] = ordered(l.lo[k]); ohi[k] = ordered(l.hi[k]); oc[k] = ordered(l.c[k]); }
  // the run this lane stands in; a lane beyond the range, or whose run is down to one leaf, stands in a run of its own
  uint32_t rs = lane < n ? 0u : lane, re = lane < n ? n : lane + 1u, rnode = rg.node, rdepth = rg.depth;
  uint32_t* bins = s_bins[wave];
  uint32_t deepest = 0, bad = 0;
  const uint64_t laneBit = 1ull << lane, below = laneBit - 1ull;
  for (int guard = 0; guard < 64; guard++) {
    const uint32_t cnt = re - rs;
    const uint64_t open = __ballot(cnt >= 2u), large = __ballot(cnt > kSmall);
    if (open == 0ull) break;
    float lo[3], hi[3], cmin[3], cmax[3], d[3], scale[3];
    int myBin[3], dim = 0;
    uint32_t left = cnt / 2u;
    bool act, pred = false, split = false;
    if (large != 0ull) {
      // ---- one large run: the run of the first lane that stands in one
      const int head = __builtin_ctzll(large);
      const uint32_t s = read_lane(rs, head), e = read_lane(re, head), depth = read_lane(rdepth, head), c = e - s;
      act = lane >= s && lane < e;
      for (int k = 0; k < 3; k++) {
        lo[k] = unordered(wave_min_u(act ? olo[k] : kMinIdentity));
        hi[k] = unordered(wave_max_u(act ? ohi[k] : kMaxIdentity));
        cmin[k] = unordered(wave_min_u(act ? oc[k] : kMinIdentity));
        cmax[k] = unordered(wave_max_u(act ? oc[k] : kMaxIdentity));
        d[k] = __fsub_rn(cmax[k], cmin[k]);
        scale[k] = bin_scale(d[k]);
        myBin[k] = bin_of(l.c[k], cmin[k], scale[k]);
      }
      dim = largest_extent(d);
      left = c / 2u;
      const uint32_t maxChild = max_child(c, heightLimit, depth);
      if (maxChild >= (c + 1u) / 2u) {
        clear_bins(bins);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (act) bin_leaf(bins, l, cmin, scale, d);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const Split sp = sah_eval(bins, d, maxChild);
        if (sp.dim >= 0) {
          split = true;
          dim = sp.dim;
          pred = picki(myBin, dim) <= sp.bin;
        }
      }
      if (!split) {
        if (pick(d, dim) > 0.0f) {   // the c / 2 smallest (centroid, leaf index) go left
          const uint32_t kh = dim == 0 ? oc[0] : (dim == 1 ? oc[1] : oc[2]);
          uint32_t rank = 0;
          for (uint64_t m = __ballot(act); m; m &= m - 1) {
            const int src = __builtin_ctzll(m);
            const uint32_t oh = read_lane(kh, src), ol = read_lane(leaf, src);
            rank += (oh < kh || (oh == kh && ol < leaf)) ? 1u : 0u;
          }
          pred = rank < left;
        } else {
          pred = lane - s < left;
        }
      }
    } else {
      // ---- every run of 2 .. kSmall leaves at once
      act = cnt >= 2u;
      const uint32_t maxCnt = wave_max_u(act ? cnt : 0u);
      uint32_t a[12];
      for (int k = 0; k < 12; k++) a[k] = (k % 6) < 3 ? kMinIdentity : kMaxIdentity;
      for (uint32_t j = 0; j < maxCnt; j++) {
        const bool has = j < cnt;
        const uint32_t src = has ? rs + j : lane;
        for (int k = 0; k < 3; k++) {
          const uint32_t vl = from_lane(olo[k], src), vh = from_lane(ohi[k], src), vc = from_lane(oc[k], src);
          if (has) { a[k] = min(a[k], vl); a[3 + k] = max(a[3 + k], vh); a[6 + k] = min(a[6 + k], vc); a[9 + k] = max(a[9 + k], vc); }
        }
      }
      for (int k = 0; k < 3; k++) {
        lo[k] = unordered(a[k]); hi[k] = unordered(a[3 + k]); cmin[k] = unordered(a[6 + k]); cmax[k] = unordered(a[9 + k]);
        d[k] = __fsub_rn(cmax[k], cmin[k]);
        scale[k] = bin_scale(d[k]);
        myBin[k] = bin_of(l.c[k], cmin[k], scale[k]);
      }
      dim = largest_extent(d);
      const uint32_t maxChild = max_child(cnt, heightLimit, rdepth);
      const bool wantsPlane = act && cnt > 2u && maxChild >= (cnt + 1u) / 2u;
      float cost = __builtin_inff();
      uint32_t idx = kNone;
      if (__ballot(wantsPlane) != 0ull) {
        for (int ax = 0; ax < 3; ax++) {
          uint32_t L[6] = {kMinIdentity, kMinIdentity, kMinIdentity, kMaxIdentity, kMaxIdentity, kMaxIdentity};
          uint32_t R[6] = {kMinIdentity, kMinIdentity, kMinIdentity, kMaxIdentity, kMaxIdentity, kMaxIdentity};
          uint32_t nL = 0;
          for (uint32_t j = 0; j < maxCnt; j++) {
            const bool has = j < cnt;
            const uint32_t src = has ? rs + j : lane;
            const int ob = (int)from_lane((uint32_t)myBin[ax], src);
            const bool isL = has && ob <= myBin[ax], isR = has && !(ob <= myBin[ax]);
            nL += isL ? 1u : 0u;
            for (int k = 0; k < 3; k++) {
              const uint32_t vl = from_lane(olo[k], src), vh = from_lane(ohi[k], src);
              L[k] = isL ? min(L[k], vl) : L[k];
              L[3 + k] = isL ? max(L[3 + k], vh) : L[3 + k];
              R[k] = isR ? min(R[k], vl) : R[k];
              R[3 + k] = isR ? max(R[3 + k], vh) : R[3 + k];
            }
          }
          const uint32_t nR = cnt - nL;
          if (wantsPlane && d[ax] > 0.0f && myBin[ax] < kBins - 1 && nR != 0u && nL <= maxChild && nR <= maxChild) {
            float flo[3], fhi[3], glo[3], ghi[3];
            for (int k = 0; k < 3; k++) { flo[k] = unordered(L[k]); fhi[k] = unordered(L[3 + k]); glo[k] = unordered(R[k]); ghi[k] = unordered(R[3 + k]); }
            const float c = __fadd_rn(__fmul_rn(half_area(flo, fhi), (float)nL), __fmul_rn(half_area(glo, ghi), (float)nR));
            const uint32_t i = (uint32_t)ax * 32u + (uint32_t)myBin[ax];
            if (c < cost || (c == cost && i < idx)) { cost = c; idx = i; }
          }
        }
        // the run's best: the minimum over its lanes of (cost, plane)
        float bc = cost;
        uint32_t bi = idx;
        for (uint32_t j = 0; j < maxCnt; j++) {
          const bool has = j < cnt;
          const uint32_t src = has ? rs + j : lane;
          const float oc2 = __uint_as_float(from_lane(__float_as_uint(cost), src));
          const uint32_t oi = from_lane(idx, src);
          if (has && (oc2 < bc || (oc2 == bc && oi < bi))) { bc = oc2; bi = oi; }
        }
        if (wantsPlane && bc < 3.402823466e+38f && bi != kNone) {
          split = true;
          dim = (int)(bi >> 5);
          pred = picki(myBin, dim) <= (int)(bi & 31u);
        }
      }
      // the runs without a plane: the cnt / 2 smallest (centroid, leaf index) go left, or the first half as it stands
      if (__ballot(act && !split) != 0ull) {
        const uint32_t kh = dim == 0 ? oc[0] : (dim == 1 ? oc[1] : oc[2]);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < maxCnt; j++) {
          const bool has = j < cnt;
          const uint32_t src = has ? rs + j : lane;
          const uint32_t oh = from_lane(kh, src), ol = from_lane(leaf, src);
          rank += (has && (oh < kh || (oh == kh && ol < leaf))) ? 1u : 0u;
        }
        if (!split) pred = pick(d, dim) > 0.0f ? rank < left : lane - rs < left;
      }
    }
    // ---- the stable partition of the runs that took part: a leaf's new lane from the lanes of its run below it, the runs cut in two
    const uint64_t mL = __ballot(act && pred), mR = __ballot(act && !pred);
    const uint64_t runMask = (re >= 64u ? ~0ull : ((1ull << re) - 1ull)) & ~((1ull << rs) - 1ull);
    if (act) left = (uint32_t)__popcll(mL & runMask);
    if (act && (left == 0u || left >= re - rs)) bad = 1;
    if (__ballot(bad != 0u) != 0ull) { bad = 1; break; }
    const uint32_t dest = !act ? lane : (pred ? rs + (uint32_t)__popcll(mL & runMask & below) : rs + left + (uint32_t)__popcll(mR & runMask & below));
    if (act && lane == rs) store_interior(out, rnode, lo, hi, rnode + 2u * left, dim);
    leaf = to_lane(leaf, dest);
    l.a = make_float4(to_lane_f(l.a.x, dest), to_lane_f(l.a.y, dest), to_lane_f(l.a.z, dest), to_lane_f(l.a.w, dest));
    l.b = make_float4(to_lane_f(l.b.x, dest), to_lane_f(l.b.y, dest), to_lane_f(l.b.z, dest), to_lane_f(l.b.w, dest));
    l.lo[0] = l.a.x; l.lo[1] = l.a.y; l.lo[2] = l.a.z;
    l.hi[0] = l.a.w; l.hi[1] = l.b.x; l.hi[2] = l.b.y;
    for (int k = 0; k < 3; k++) {
      l.c[k] = __fadd_rn(__fmul_rn(0.5f, l.lo[k]), __fmul_rn(0.5f, l.hi[k]));
      olo[k] = ordered(l.lo[k]); ohi[k] = ordered(l.hi[k]); oc[k] = ordered(l.c[k]);
    }
    if (act) {
      deepest = max(deepest, rdepth + 1u);
      if (lane < rs + left) { re = rs + left; rnode = rnode + 1u; }
      else { rnode = rnode + 2u * left; rs = rs + left; }
      rdepth++;
      if (re - rs == 1u) {   // a run of one: the leaf's node
        out[2 * (size_t)rnode] = l.a;
        out[2 * (size_t)rnode + 1] = l.b;
      }
    }
  }
  deepest = wave_max_u(deepest);
  bad |= __ballot(re - rs >= 2u) != 0ull ? 1u : 0u;   // (64 rounds cut any 64 leaves into single ones)
  if (lane == 0) {
    raise_max(&ctl[kCtlOwnHeight], deepest);
    if (bad) atomicOr(&ctl[kCtlFlags], (uint32_t)kFlagInternal);
  }
}

// four leaves per lane and round of a looping wavefront: the indices first, then the nodes
struct WaveLeaves { uint32_t leaf[4]; float4 a[4], b[4]; };
__device__ __forceinline__ void load_wave(const float4* __restrict__ nd, const uint32_t* __restrict__ order, uint32_t base, uint32_t end, WaveLeaves& wl) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = base + (uint32_t)j * 64u + lane_id();
    wl.leaf[j] = i < end ? order[i] : kNone;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t at = wl.leaf[j] != kNone ? wl.leaf[j] : 0u;
    wl.a[j] = nd[2 * (size_t)at];
    wl.b[j] = nd[2 * (size_t)at + 1];
  }
}
template <class LEAVES>
__device__ __forceinline__ void leaf_of(const LEAVES& cl, int j, Leaf& l) {
  l.a = cl.a[j];
  l.b = cl.b[j];
  l.lo[0] = l.a.x; l.lo[1] = l.a.y; l.lo[2] = l.a.z;
  l.hi[0] = l.a.w; l.hi[1] = l.b.x; l.hi[2] = l.b.y;
  for (int k = 0; k < 3; k++) l.c[k] = __fadd_rn(__fmul_rn(0.5f, l.lo[k]), __fmul_rn(0.5f, l.hi[k]));
}

// ------------------------------------------------------------------------------------------ one wavefront per range, looping
constexpr int kMidBlock = 512;
__device__ void k_mid_range(const float4* __restrict__ nd, const uint32_t* __restrict__ orderIn, uint32_t* __restrict__ orderOut, const Range rg,
                            uint32_t* bins, float4* __restrict__ out, int heightLimit, uint32_t* __restrict__ ctl, Range* child);
__global__ __launch_bounds__(kMidBlock) void k_mid(const float4* __restrict__ nd, const uint32_t* __restrict__ orderIn, uint32_t* __restrict__ orderOut,
                                              const Range* __restrict__ list, const uint32_t* __restrict__ counts, float4* __restrict__ out,
                                              int heightLimit, Range* __restrict__ nextBig, Range* __restrict__ nextMid, Range* __restrict__ nextTiny,
                                              uint32_t* __restrict__ nextCounts, uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s_bins[kMidBlock / 64][kRangeBins];
  __shared__ uint32_t s_req[3], s_base[3];
  const uint32_t wave = threadIdx.x >> 6, lane = lane_id(), w = blockIdx.x * (kMidBlock / 64) + wave;
  if (threadIdx.x < 3) s_req[threadIdx.x] = 0u;
  __syncthreads();
  // the children of this wavefront's range: their lists' slots are claimed per WORKGROUP (one atomic per list instead of two per range)
  Range child[2] = {Range{0u, 0u, 0u, 0u}, Range{0u, 0u, 0u, 0u}};
  int kind[2] = {-1, -1};
  uint32_t slot[2] = {0u, 0u};
  if (w < counts[kCntMid]) k_mid_range(nd, orderIn, orderOut, list[w], s_bins[wave], out, heightLimit, ctl, child);
  if (lane == 0)
    for (int q = 0; q < 2; q++) {
      const uint32_t count = child[q].end - child[q].start;
      if (count < 2) continue;
      kind[q] = count <= kTiny ? kCntTiny : (count <= kChunk ? kCntMid : kCntBig);
      slot[q] = atomicAdd(&s_req[kind[q]], 1u);
    }
  __syncthreads();
  if (threadIdx.x < 3 && s_req[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&nextCounts[threadIdx.x], s_req[threadIdx.x]);
  __syncthreads();
  if (lane == 0)
    for (int q = 0; q < 2; q++) {
      if (kind[q] < 0) continue;
      Range* to = kind[q] == kCntTiny ? nextTiny : (kind[q] == kCntMid ? nextMid : nextBig);
      to[s_base[kind[q]] + slot[q]] = child[q];
    }
}

// one range of k_mid (all 64 lanes); child[]: the two ranges it leaves behind (lane 0's copy counts)
__device__ void k_mid_range(const float4* __restrict__ nd, const uint32_t* __restrict__ orderIn, uint32_t* __restrict__ orderOut, const Range rg,
                            uint32_t* bins, float4* __restrict__ out, int heightLimit, uint32_t* __restrict__ ctl, Range* child) {
  const uint32_t lane = lane_id();
  const uint32_t cnt = rg.end - rg.start;
  // bounds
  uint32_t olo[3] = {kMinIdentity, kMinIdentity, kMinIdentity}, ohi[3] = {kMaxIdentity, kMaxIdentity, kMaxIdentity};
  uint32_t ocmin[3] = {kMinIdentity, kMinIdentity, kMinIdentity}, ocmax[3] = {kMaxIdentity, kMaxIdentity, kMaxIdentity};
  for (uint32_t base = rg.start; base < rg.end; base += 256) {
    WaveLeaves wl;
    load_wave(nd, orderIn, base, rg.end, wl);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (wl.leaf[j] == kNone) continue;
      Leaf l;
      leaf_of(wl, j, l);
      for (int k = 0; k < 3; k++) {
        olo[k] = min(olo[k], ordered(l.lo[k]));
        ohi[k] = max(ohi[k], ordered(l.hi[k]));
        ocmin[k] = min(ocmin[k], ordered(l.c[k]));
        ocmax[k] = max(ocmax[k], ordered(l.c[k]));
      }
    }
  }
  float lo[3], hi[3], cmin[3], cmax[3], d[3], scale[3];
  for (int k = 0; k < 3; k++) {
    lo[k] = unordered(wave_min_u(olo[k]));
    hi[k] = unordered(wave_max_u(ohi[k]));
    cmin[k] = unordered(wave_min_u(ocmin[k]));
    cmax[k] = unordered(wave_max_u(ocmax[k]));
    d[k] = __fsub_rn(cmax[k], cmin[k]);
    scale[k] = bin_scale(d[k]);
  }
  int dim = largest_extent(d);
  uint32_t left = cnt / 2;
  int mode = 2, bestBin = 0;   // 0: bins up to bestBin go left; 1: keys below the pivot; 2: the first half as it stands
  uint64_t pivot = 0;
  const uint32_t maxChild = max_child(cnt, heightLimit, rg.depth);
  if (cnt > 2 && maxChild >= (cnt + 1) / 2) {
    clear_bins(bins);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t base = rg.start; base < rg.end; base += 256) {
      // (four CONSECUTIVE leaves per lane, united in registers while the bin stays: see k_big_bins)
      WaveLeaves wl;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t i = base + lane * 4u + (uint32_t)j;
        wl.leaf[j] = i < rg.end ? orderIn[i] : kNone;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t at = wl.leaf[j] != kNone ? wl.leaf[j] : 0u;
        wl.a[j] = nd[2 * (size_t)at];
        wl.b[j] = nd[2 * (size_t)at + 1];
      }
      for (int a = 0; a < 3; a++) {
        if (!(d[a] > 0.0f)) continue;
        int cur = -1;
        uint32_t acc[6] = {kMinIdentity, kMinIdentity, kMinIdentity, kMaxIdentity, kMaxIdentity, kMaxIdentity}, count = 0;
        auto flush = [&]() {
          if (cur < 0) return;
          uint32_t* w = bins + (a * 32 + cur) * kBinWords;
          for (int k = 0; k < 3; k++) { atomicMin(&w[k], acc[k]); atomicMax(&w[3 + k], acc[3 + k]); }
          atomicAdd(&w[6], count);
        };
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (wl.leaf[j] == kNone) continue;
          Leaf l;
          leaf_of(wl, j, l);
          const int bin = bin_of(l.c[a], cmin[a], scale[a]);
          if (bin != cur) {
            flush();
            cur = bin;
            for (int k = 0; k < 3; k++) { acc[k] = kMinIdentity; acc[3 + k] = kMaxIdentity; }
            count = 0;
          }
          for (int k = 0; k < 3; k++) { acc[k] = min(acc[k], ordered(l.lo[k])); acc[3 + k] = max(acc[3 + k], ordered(l.hi[k])); }
          count++;
        }
        flush();
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const Split sp = sah_eval(bins, d, maxChild);
    if (sp.dim >= 0) { mode = 0; dim = sp.dim; bestBin = sp.bin; left = sp.left; }
  }
  if (mode != 0 && pick(d, dim) > 0.0f) {
    // the key of rank cnt / 2 among (ordered centroid, leaf index), eight bits at a time (the histogram lives where the bins did)
    mode = 1;
    uint32_t k = cnt / 2;
    for (int shift = 56; shift >= 0; shift -= 8) {
      for (uint32_t i = lane; i < 256; i += 64) bins[i] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint64_t mask = shift == 56 ? 0ull : ~0ull << (shift + 8);
      for (uint32_t i = rg.start + lane; i < rg.end; i += 64) {
        Leaf l;
        const uint32_t leaf = orderIn[i];
        load_leaf(nd, leaf, l);
        const uint64_t key = ((uint64_t)ordered(pick(l.c, dim)) << 32) | leaf;
        if ((key & mask) == (pivot & mask)) atomicAdd(&bins[(uint32_t)(key >> shift) & 255u], 1u);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t h0 = bins[4 * lane], h1 = bins[4 * lane + 1], h2 = bins[4 * lane + 2], h3 = bins[4 * lane + 3];
      const uint32_t mine = h0 + h1 + h2 + h3;
      uint32_t incl = mine;
      for (int s = 1; s < 64; s <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)incl, s);
        if (lane >= (uint32_t)s) incl += u;
      }
      const uint32_t excl = incl - mine;
      const bool owner = k >= excl && k < incl;
      uint32_t digit = 0, rest = 0;
      if (owner) {
        uint32_t r = k - excl;
        if (r < h0) { digit = 4 * lane; }
        else if ((r -= h0) < h1) { digit = 4 * lane + 1; }
        else if ((r -= h1) < h2) { digit = 4 * lane + 2; }
        else { r -= h2; digit = 4 * lane + 3; }
        rest = r;
      }
      const uint64_t who = __ballot(owner);
      if (who == 0) { mode = 3; break; }   // (cannot happen: k < the number of keys that share the prefix)
      const int src = __builtin_ctzll(who);
      digit = (uint32_t)__shfl((int)digit, src);
      k = (uint32_t)__shfl((int)rest, src);
      pivot |= (uint64_t)digit << shift;
    }
  }
  if (mode == 3 || left == 0 || left >= cnt) {
    if (lane == 0) atomicOr(&ctl[kCtlFlags], (uint32_t)kFlagInternal);
    return;
  }
  // stable partition into the next level's order
  const uint32_t rightNode = rg.node + 2u * left, right = cnt - left;
  uint32_t doneL = 0, doneR = 0;
  const float cminD = pick(cmin, dim), scaleD = pick(scale, dim);
  for (uint32_t base = rg.start; base < rg.end; base += 256) {
    WaveLeaves wl;
    load_wave(nd, orderIn, base, rg.end, wl);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t i = base + (uint32_t)j * 64u + lane, leaf = wl.leaf[j];
      const bool in = leaf != kNone;
      Leaf l;
      leaf_of(wl, j, l);
      const float cD = pick(l.c, dim);
      bool pred;
      if (mode == 0) pred = bin_of(cD, cminD, scaleD) <= bestBin;
      else if (mode == 1) pred = (((uint64_t)ordered(cD) << 32) | leaf) < pivot;
      else pred = i - rg.start < left;
      const uint64_t mL = __ballot(in && pred), mR = __ballot(in && !pred), below = (1ull << lane) - 1ull;
      if (in) {
        const uint32_t to = pred ? rg.start + doneL + (uint32_t)__popcll(mL & below) : rg.start + left + doneR + (uint32_t)__popcll(mR & below);
        orderOut[to] = leaf;
        if (left == 1 && to == rg.start) { out[2 * (size_t)(rg.node + 1)] = l.a; out[2 * (size_t)(rg.node + 1) + 1] = l.b; }
        if (right == 1 && to == rg.start + left) { out[2 * (size_t)rightNode] = l.a; out[2 * (size_t)rightNode + 1] = l.b; }
      }
      doneL += (uint32_t)__popcll(mL);
      doneR += (uint32_t)__popcll(mR);
    }
  }
  if (lane == 0) {
    if (doneL != left) atomicOr(&ctl[kCtlFlags], (uint32_t)kFlagInternal);
    store_interior(out, rg.node, lo, hi, rightNode, dim);
    child[0] = Range{rg.start, rg.start + left, rg.node + 1, rg.depth + 1};
    child[1] = Range{rg.start + left, rg.end, rightNode, rg.depth + 1};
    raise_max(&ctl[kCtlOwnHeight], rg.depth + 1);   // (both children exist one level down)
  }
}

// ------------------------------------------------------------------------------------------ many workgroups per range
// workgroup r readies the accumulators of range r; workgroup 0 also lays the ranges out in chunks of kChunk leaves
__global__ __launch_bounds__(256) void k_big_setup(const Range* __restrict__ list, uint32_t nBig, uint32_t* __restrict__ chunkBase,
                                                    uint32_t* __restrict__ chunkRange, uint32_t* __restrict__ acc, uint32_t* __restrict__ bins,
                                                    uint32_t* __restrict__ ctl) {
  const uint32_t r = blockIdx.x;
  if (threadIdx.x < 12) acc[12 * (size_t)r + threadIdx.x] = (threadIdx.x % 6) < 3 ? kMinIdentity : kMaxIdentity;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kRangeBins; i += 256) bins[(size_t)r * kRangeBins + i] = (i % kBinWords) < 3 ? kMinIdentity : kMaxIdentity;
  if (r != 0) return;
  if (threadIdx.x == 0) {
    uint32_t base = 0;
    for (uint32_t q = 0; q < nBig; q++) {
      chunkBase[q] = base;
      base += (list[q].end - list[q].start + kChunk - 1) / kChunk;
    }
    chunkBase[nBig] = base;
    ctl[kCtlChunks] = base;
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < nBig; q += 256)
    for (uint32_t c = chunkBase[q]; c < chunkBase[q + 1]; c++) chunkRange[c] = q;
}

struct ChunkOf { uint32_t r, s, e; Range rg; };
__device__ __forceinline__ bool chunk_of(const Range* __restrict__ list, const uint32_t* __restrict__ chunkBase, const uint32_t* __restrict__ chunkRange,
                                         const uint32_t* __restrict__ ctl, ChunkOf& c) {
  if (blockIdx.x >= ctl[kCtlChunks]) return false;
  c.r = chunkRange[blockIdx.x];
  c.rg = list[c.r];
  c.s = c.rg.start + (blockIdx.x - chunkBase[c.r]) * kChunk;
  c.e = min(c.s + kChunk, c.rg.end);
  return true;
}

// the leaves of a chunk, eight per thread: all indices first, then all nodes (two rounds of memory latency, not sixteen)
struct ChunkLeaves { uint32_t leaf[8]; float4 a[8], b[8]; };
__device__ __forceinline__ void load_chunk(const float4* __restrict__ nd, const uint32_t* __restrict__ order, uint32_t s, uint32_t e, ChunkLeaves& cl) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t i = s + (uint32_t)j * 256u + threadIdx.x;
    cl.leaf[j] = i < e ? order[i] : kNone;
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t at = cl.leaf[j] != kNone ? cl.leaf[j] : 0u;
    cl.a[j] = nd[2 * (size_t)at];
    cl.b[j] = nd[2 * (size_t)at + 1];
  }
}
__global__ __launch_bounds__(256) void k_big_bounds(const float4* __restrict__ nd, const uint32_t* __restrict__ order, const Range* __restrict__ list,
                                                     const uint32_t* __restrict__ chunkBase, const uint32_t* __restrict__ chunkRange,
                                                     uint32_t* __restrict__ acc, const uint32_t* __restrict__ ctl) {
  ChunkOf c;
  if (!chunk_of(list, chunkBase, chunkRange, ctl, c)) return;
  __shared__ uint32_t s_v[4][12];
  uint32_t v[12];
  for (int k = 0; k < 12; k++) v[k] = (k % 6) < 3 ? kMinIdentity : kMaxIdentity;
  ChunkLeaves cl;
  load_chunk(nd, order, c.s, c.e, cl);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if (cl.leaf[j] == kNone) continue;
    Leaf l;
    leaf_of(cl, j, l);
    for (int k = 0; k < 3; k++) {
      v[k] = min(v[k], ordered(l.lo[k]));
      v[3 + k] = max(v[3 + k], ordered(l.hi[k]));
      v[6 + k] = min(v[6 + k], ordered(l.c[k]));
      v[9 + k] = max(v[9 + k], ordered(l.c[k]));
    }
  }
  for (int k = 0; k < 12; k++) {
    const uint32_t w = (k % 6) < 3 ? wave_min_u(v[k]) : wave_max_u(v[k]);
    if (lane_id() == 0) s_v[threadIdx.x >> 6][k] = w;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const uint32_t k = threadIdx.x;
    if ((k % 6) < 3) atomicMin(&acc[12 * (size_t)c.r + k], min(min(s_v[0][k], s_v[1][k]), min(s_v[2][k], s_v[3][k])));
    else atomicMax(&acc[12 * (size_t)c.r + k], max(max(s_v[0][k], s_v[1][k]), max(s_v[2][k], s_v[3][k])));
  }
}

struct Bounds { float lo[3], hi[3], cmin[3], d[3], scale[3]; };
__device__ __forceinline__ void read_bounds(const uint32_t* __restrict__ acc, uint32_t r, Bounds& b) {
  for (int k = 0; k < 3; k++) {
    b.lo[k] = unordered(acc[12 * (size_t)r + k]);
    b.hi[k] = unordered(acc[12 * (size_t)r + 3 + k]);
    b.cmin[k] = unordered(acc[12 * (size_t)r + 6 + k]);
    b.d[k] = __fsub_rn(unordered(acc[12 * (size_t)r + 9 + k]), b.cmin[k]);
    b.scale[k] = bin_scale(b.d[k]);
  }
}

__global__ __launch_bounds__(256) void k_big_bins(const float4* __restrict__ nd, const uint32_t* __restrict__ order, const Range* __restrict__ list,
                                                   const uint32_t* __restrict__ chunkBase, const uint32_t* __restrict__ chunkRange,
                                                   const uint32_t* __restrict__ acc, uint32_t* __restrict__ bins, const uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s_bins[kRangeBins];
  ChunkOf c;
  if (!chunk_of(list, chunkBase, chunkRange, ctl, c)) return;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kRangeBins; i += 256) s_bins[i] = (i % kBinWords) < 3 ? kMinIdentity : kMaxIdentity;
  Bounds b;
  read_bounds(acc, c.r, b);
  __syncthreads();
  {
    // a thread takes eight CONSECUTIVE leaves of the chunk: neighbours in the order are neighbours in space (the caller's
    // pre-order, kept by the stable partitions), so they mostly fall into one bin -- the thread unites them in registers and
    // sends a bin's seven atomics when the bin changes (a wavefront's 64 leaves one by one would queue up on one LDS word each)
    ChunkLeaves cl;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = c.s + threadIdx.x * 8u + (uint32_t)j;
      cl.leaf[j] = i < c.e ? order[i] : kNone;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t at = cl.leaf[j] != kNone ? cl.leaf[j] : 0u;
      cl.a[j] = nd[2 * (size_t)at];
      cl.b[j] = nd[2 * (size_t)at + 1];
    }
    for (int a = 0; a < 3; a++) {
      if (!(b.d[a] > 0.0f)) continue;
      int cur = -1;
      uint32_t acc[6] = {kMinIdentity, kMinIdentity, kMinIdentity, kMaxIdentity, kMaxIdentity, kMaxIdentity}, count = 0;
      auto flush = [&]() {
        if (cur < 0) return;
        uint32_t* w = s_bins + (a * 32 + cur) * kBinWords;
        for (int k = 0; k < 3; k++) { atomicMin(&w[k], acc[k]); atomicMax(&w[3 + k], acc[3 + k]); }
        atomicAdd(&w[6], count);
      };
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (cl.leaf[j] == kNone) continue;
        Leaf l;
        leaf_of(cl, j, l);
        const int bin = bin_of(l.c[a], b.cmin[a], b.scale[a]);
        if (bin != cur) {
          flush();
          cur = bin;
          for (int k = 0; k < 3; k++) { acc[k] = kMinIdentity; acc[3 + k] = kMaxIdentity; }
          count = 0;
        }
        for (int k = 0; k < 3; k++) { acc[k] = min(acc[k], ordered(l.lo[k])); acc[3 + k] = max(acc[3 + k], ordered(l.hi[k])); }
        count++;
      }
      flush();
    }
  }
  __syncthreads();
  uint32_t* g = bins + (size_t)c.r * kRangeBins;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kRangeBins; i += 256) {
    const uint32_t k = i % kBinWords;
    if (s_bins[i - k + 6] == 0) continue;   // an empty bin
    if (k < 3) atomicMin(&g[i], s_bins[i]);
    else if (k < 6) atomicMax(&g[i], s_bins[i]);
    else atomicAdd(&g[i], s_bins[i]);
  }
}

// every workgroup of a range finds the range's split for itself (the same function of the same bins) and counts the leaves of
// its chunk that go left; the range's first workgroup writes the node and hands on the children -- or, when no plane is
// feasible, hands the range to the looping wavefronts of this level (which know the median rule)
__global__ __launch_bounds__(256) void k_big_split(const float4* __restrict__ nd, const uint32_t* __restrict__ order, const Range* __restrict__ list,
                                                    const uint32_t* __restrict__ chunkBase, const uint32_t* __restrict__ chunkRange,
                                                    const uint32_t* __restrict__ acc, const uint32_t* __restrict__ bins, SplitRec* __restrict__ splits,
                                                    uint32_t* __restrict__ chunkLeft, float4* __restrict__ out, int heightLimit,
                                                    Range* __restrict__ curMid, uint32_t* __restrict__ curCounts, Range* __restrict__ nextBig,
                                                    Range* __restrict__ nextMid, Range* __restrict__ nextTiny, uint32_t* __restrict__ nextCounts,
                                                    uint32_t* __restrict__ ctl) {
  __shared__ Split s_split;
  __shared__ uint32_t s_count[4];
  ChunkOf c;
  if (!chunk_of(list, chunkBase, chunkRange, ctl, c)) return;
  Bounds b;
  read_bounds(acc, c.r, b);
  const uint32_t cnt = c.rg.end - c.rg.start;
  if (threadIdx.x < 64) {
    const Split sp = sah_eval(bins + (size_t)c.r * kRangeBins, b.d, max_child(cnt, heightLimit, c.rg.depth));
    if (threadIdx.x == 0) s_split = sp;
  }
  __syncthreads();
  const Split sp = s_split;
  const bool first = c.s == c.rg.start;
  if (sp.dim < 0 || sp.left == 0 || sp.left >= cnt) {
    if (first && threadIdx.x == 0) {
      splits[c.r].dim = -1;
      curMid[atomicAdd(&curCounts[kCntMid], 1u)] = c.rg;
    }
    return;
  }
  uint32_t mine = 0;
  {
    const float cminD = pick(b.cmin, sp.dim), scaleD = pick(b.scale, sp.dim);
    ChunkLeaves cl;
    load_chunk(nd, order, c.s, c.e, cl);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (cl.leaf[j] == kNone) continue;
      Leaf l;
      leaf_of(cl, j, l);
      mine += bin_of(pick(l.c, sp.dim), cminD, scaleD) <= sp.bin ? 1u : 0u;
    }
  }
  mine = wave_sum_u(mine);
  if (lane_id() == 0) s_count[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    chunkLeft[blockIdx.x] = s_count[0] + s_count[1] + s_count[2] + s_count[3];
    if (first) {
      SplitRec rec{sp.dim, sp.bin, pick(b.cmin, sp.dim), pick(b.scale, sp.dim), sp.left, 0u, 0u, 0u};
      splits[c.r] = rec;
      const uint32_t rightNode = c.rg.node + 2u * sp.left;
      store_interior(out, c.rg.node, b.lo, b.hi, rightNode, sp.dim);
      emit_child(Range{c.rg.start, c.rg.start + sp.left, c.rg.node + 1, c.rg.depth + 1}, nextBig, nextMid, nextTiny, nextCounts);
      emit_child(Range{c.rg.start + sp.left, c.rg.end, rightNode, c.rg.depth + 1}, nextBig, nextMid, nextTiny, nextCounts);
      raise_max(&ctl[kCtlOwnHeight], c.rg.depth + 1);
    }
  }
}

__global__ __launch_bounds__(256) void k_big_scatter(const float4* __restrict__ nd, const uint32_t* __restrict__ orderIn, uint32_t* __restrict__ orderOut,
                                                      const Range* __restrict__ list, const uint32_t* __restrict__ chunkBase,
                                                      const uint32_t* __restrict__ chunkRange, const SplitRec* __restrict__ splits,
                                                      const uint32_t* __restrict__ chunkLeft, float4* __restrict__ out, uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s_sum[4], s_cl[8][4], s_cr[8][4];
  ChunkOf c;
  if (!chunk_of(list, chunkBase, chunkRange, ctl, c)) return;
  const SplitRec sp = splits[c.r];
  if (sp.dim < 0) return;
  // leaves of this range that go left in the chunks before this one
  uint32_t before = 0;
  for (uint32_t q = chunkBase[c.r] + threadIdx.x; q < blockIdx.x; q += 256) before += chunkLeft[q];
  before = wave_sum_u(before);
  if (lane_id() == 0) s_sum[threadIdx.x >> 6] = before;
  __syncthreads();
  before = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
  const uint32_t cnt = c.rg.end - c.rg.start, right = cnt - sp.left, rightNode = c.rg.node + 2u * sp.left;
  const uint32_t toL = c.rg.start + before, toR = c.rg.start + sp.left + (c.s - c.rg.start - before);
  const uint32_t wave = threadIdx.x >> 6;
  // the chunk in rounds of 256 leaves: where a leaf goes = the leaves of its side in earlier rounds, in earlier wavefronts of
  // its round, in lower lanes of its wavefront
  ChunkLeaves cl;
  load_chunk(nd, orderIn, c.s, c.e, cl);
  bool pred[8];
  uint64_t mL[8], mR[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const bool in = cl.leaf[j] != kNone;
    Leaf l;
    leaf_of(cl, j, l);
    pred[j] = bin_of(pick(l.c, sp.dim), sp.cmin, sp.scale) <= sp.bin;
    mL[j] = __ballot(in && pred[j]);
    mR[j] = __ballot(in && !pred[j]);
    if (lane_id() == 0) { s_cl[j][wave] = (uint32_t)__popcll(mL[j]); s_cr[j][wave] = (uint32_t)__popcll(mR[j]); }
  }
  __syncthreads();
  uint32_t runL = 0, runR = 0;
  const uint64_t below = (1ull << lane_id()) - 1ull;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint32_t wl = 0, wr = 0, allL = 0, allR = 0;
    for (uint32_t q = 0; q < 4; q++) {
      if (q < wave) { wl += s_cl[j][q]; wr += s_cr[j][q]; }
      allL += s_cl[j][q];
      allR += s_cr[j][q];
    }
    if (cl.leaf[j] != kNone) {
      const uint32_t to = pred[j] ? toL + runL + wl + (uint32_t)__popcll(mL[j] & below) : toR + runR + wr + (uint32_t)__popcll(mR[j] & below);
      orderOut[to] = cl.leaf[j];
      if (sp.left == 1 && to == c.rg.start) { out[2 * (size_t)(c.rg.node + 1)] = cl.a[j]; out[2 * (size_t)(c.rg.node + 1) + 1] = cl.b[j]; }
      if (right == 1 && to == c.rg.start + sp.left) { out[2 * (size_t)rightNode] = cl.a[j]; out[2 * (size_t)rightNode + 1] = cl.b[j]; }
    }
    runL += allL;
    runR += allR;
  }
  (void)ctl;
}

// ------------------------------------------------------------------------------------------ 4-wide groups
__global__ void k_wide_mark(const lt_retree::Node* __restrict__ own, const uint32_t* __restrict__ frontier, const uint32_t* __restrict__ counts,
                            uint32_t* __restrict__ next, uint32_t* __restrict__ nextCounts, uint32_t* __restrict__ groupOf) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t kids[4], interior = 0;
  if (t < counts[kCntWide]) {
    const uint32_t b = frontier[t];
    groupOf[b] = 0u;
    const int count = lt_retree::wide_kids(own, b, kids);
    for (int k = 0; k < count; k++)
      if (own[kids[k]].cnt == 0) interior++;   // (they stand first: wide_kids' order)
  }
  // one append per wavefront
  uint32_t incl = interior;
  for (int s = 1; s < 64; s <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)incl, s);
    if (lane_id() >= (uint32_t)s) incl += u;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63);
  uint32_t base = 0;
  if (lane_id() == 0 && total) base = atomicAdd(&nextCounts[kCntWide], total);
  base = (uint32_t)__shfl((int)base, 0);
  for (uint32_t k = 0; k < interior; k++) next[base + incl - interior + k] = kids[k];
}

// exclusive prefix sum over "is a group's node" (groupOf == 0) in three launches of 2048 nodes per workgroup
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* __restrict__ groupOf, uint32_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t s[4];
  uint32_t mine = 0;
  const uint32_t base = blockIdx.x * 2048u;
  for (uint32_t i = base + threadIdx.x; i < min(base + 2048u, n); i += 256) mine += groupOf[i] == 0u ? 1u : 0u;
  mine = wave_sum_u(mine);
  if (lane_id() == 0) s[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void k_scan_top(uint32_t* __restrict__ sums, uint32_t blocks, uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s[4];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < blocks; base += 256) {
    const uint32_t i = base + threadIdx.x, v = i < blocks ? sums[i] : 0u;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)incl, d);
      if (lane_id() >= (uint32_t)d) incl += u;
    }
    if (lane_id() == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) before += s[q];
    if (i < blocks) sums[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) ctl[kCtlGroups] = carry;
}
__global__ __launch_bounds__(256) void k_scan_apply(uint32_t* __restrict__ groupOf, uint32_t n, const uint32_t* __restrict__ sums) {
  __shared__ uint32_t s[4];
  uint32_t running = sums[blockIdx.x];
  const uint32_t base = blockIdx.x * 2048u;
  for (uint32_t round = 0; round < 8; round++) {
    const uint32_t i = base + round * 256u + threadIdx.x;
    const bool marked = i < n && groupOf[i] == 0u;
    const uint64_t m = __ballot(marked);
    __syncthreads();
    if (lane_id() == 0) s[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t q = 0; q < 4; q++) {
      if (q < (threadIdx.x >> 6)) before += s[q];
      all += s[q];
    }
    if (marked) groupOf[i] = running + before + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
    running += all;
  }
}
__global__ void k_wide_children(const lt_retree::Node* __restrict__ own, uint32_t n, const uint32_t* __restrict__ groupOf, uint32_t* __restrict__ children) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const uint32_t g = groupOf[b];
  if (g == kNone) return;
  uint32_t kids[4];
  const int count = lt_retree::wide_kids(own, b, kids);
  for (int k = 0; k < 4; k++) children[4 * (size_t)g + k] = k < count ? kids[k] : kNone;
}

__global__ void k_seed(Range* __restrict__ big, Range* __restrict__ mid, Range* __restrict__ tiny, uint32_t* __restrict__ counts, uint32_t n,
                       uint32_t* __restrict__ frontier, uint32_t* __restrict__ wideCounts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (n) emit_child(Range{0u, n, 0u, 0u}, big, mid, tiny, counts);
  if (frontier) { frontier[0] = 0u; wideCounts[kCntWide] = 1u; }
}

// ------------------------------------------------------------------------------------------ host side
namespace {
struct Carver {
  size_t size = 0;
  size_t take(size_t bytes) {
    const size_t at = size;
    size += (bytes + 255) & ~(size_t)255;
    return at;
  }
};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

hipError_t check_primitives(const void* d_prims, uint32_t n_prims, uint32_t n_mats, hipStream_t stream, uint32_t* d_word, bool& ok) {
  hipError_t e = hipMemsetAsync(d_word, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_check_prims, dim3((n_prims + 255) / 256), dim3(256), 0, stream, (const int32_t*)d_prims, n_prims, n_mats, d_word);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  uint32_t flags = 0;
  if ((e = hipMemcpyAsync(&flags, d_word, sizeof(flags), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  ok = flags == 0;
  return hipSuccess;
}

void release(Out& out, const Allocator& al) {
  for (void** p : {&out.d_nodes2, &out.d_rank8, &out.d_children, &out.d_groupOf}) {
    if (*p) al.put(al.self, *p);
    *p = nullptr;
  }
}

#define LT_PREP_CHECK(expr)                      \
  do {                                           \
    const hipError_t e_ = (expr);                \
    if (e_ != hipSuccess) {                      \
      if (scratch) al.put(al.self, scratch);     \
      release(out, al);                          \
      return e_;                                 \
    }                                            \
  } while (0)

hipError_t run(const void* d_nodes, uint32_t n_nodes, const void* d_prims, uint32_t n_prims, uint32_t n_mats, int maxHeight, int slack,
               bool ownSplits, hipStream_t stream, const Allocator& al, Out& out) {
  static_assert(kBins == 32, "the device build scans 32 bins with 32 lanes");
  out = Out{};
  uint8_t* scratch = nullptr;
  if (n_nodes < 3 || (n_nodes & 1u) == 0u || n_nodes > 0x7fffffffu) {   // (a tree of two leaves at least has an odd number of nodes)
    out.flags = kFlagNotProper;
    return hipSuccess;
  }
  const uint32_t N = n_nodes, n = (N + 1) / 2;
  if ((uint64_t)n * 2 - 1 > 0x03ffffffull) { out.flags = kFlagNoRoom; return hipSuccess; }
  const int log2n = lt_retree::ceil_log2(n);
  const int heightLimit = std::min(maxHeight, log2n + slack);
  if (ownSplits && heightLimit < log2n) { out.flags = kFlagNoRoom; return hipSuccess; }
  const int maxLevels = 66;
  const uint32_t maxBig = n / kChunk + 2, maxMid = n / 32 + 2, maxTiny = n / 2 + 2, maxChunks = n / kChunk + maxBig + 2;
  Carver cv;
  const size_t o_parent = cv.take((size_t)N * 4), o_up = cv.take((size_t)N * 16), o_seen = cv.take(((size_t)n_prims + 31) / 32 * 4);
  const size_t o_order0 = cv.take((size_t)n * 4), o_order1 = cv.take((size_t)n * 4);
  const size_t o_big0 = cv.take((size_t)maxBig * 16), o_big1 = cv.take((size_t)maxBig * 16);
  const size_t o_mid0 = cv.take((size_t)maxMid * 16), o_mid1 = cv.take((size_t)maxMid * 16);
  const size_t o_tiny0 = cv.take((size_t)maxTiny * 16), o_tiny1 = cv.take((size_t)maxTiny * 16);
  const size_t o_counts = cv.take((size_t)(maxLevels + 2) * kCntWords * 4), o_ctl = cv.take(kCtlWords * 4);
  const size_t o_chunkBase = cv.take((size_t)(maxBig + 1) * 4), o_chunkRange = cv.take((size_t)maxChunks * 4), o_chunkLeft = cv.take((size_t)maxChunks * 4);
  const size_t o_acc = cv.take((size_t)maxBig * 12 * 4), o_bins = cv.take((size_t)maxBig * kRangeBins * 4), o_splits = cv.take((size_t)maxBig * sizeof(SplitRec));
  const size_t o_front0 = cv.take((size_t)n * 4), o_front1 = cv.take((size_t)n * 4), o_sums = cv.take(((size_t)(2 * n) / 2048 + 2) * 4);
  LT_PREP_CHECK(al.get(al.self, (void**)&scratch, cv.size));
  auto at = [&](size_t off) { return scratch + off; };
  uint32_t* ctl = (uint32_t*)at(o_ctl);
  uint32_t* counts = (uint32_t*)at(o_counts);
  const float4* nd = (const float4*)d_nodes;
  const double t0 = now_ms();

  // ---- checks, leaf order table, the leaves in the caller's pre-order
  LT_PREP_CHECK(al.get(al.self, &out.d_rank8, (size_t)n_prims * 32));
  LT_PREP_CHECK(hipMemsetAsync(at(o_parent), 0xff, (size_t)N * 4, stream));
  LT_PREP_CHECK(hipMemsetAsync(at(o_seen), 0, ((size_t)n_prims + 31) / 32 * 4, stream));
  LT_PREP_CHECK(hipMemsetAsync(counts, 0, (size_t)(maxLevels + 2) * kCntWords * 4, stream));
  LT_PREP_CHECK(hipMemsetAsync(ctl, 0, kCtlWords * 4, stream));
  LT_PREP_CHECK(hipMemsetAsync(out.d_rank8, 0xff, (size_t)n_prims * 32, stream));
  LT_PREP_CHECK(hipMemsetAsync(at(o_order0), 0xff, (size_t)n * 4, stream));
  const dim3 perNode((N + 255) / 256), block(256);
  hipLaunchKernelGGL(k_check_nodes, perNode, block, 0, stream, nd, N, n_prims, (uint32_t*)at(o_parent), (uint32_t*)at(o_seen), ctl);
  if (d_prims) hipLaunchKernelGGL(k_check_prims, dim3((n_prims + 255) / 256), block, 0, stream, (const int32_t*)d_prims, n_prims, n_mats, ctl);
  hipLaunchKernelGGL(k_subtree_ends, perNode, block, 0, stream, nd, N, (const uint32_t*)at(o_parent), (uint4*)at(o_up), ctl);
  hipLaunchKernelGGL(k_leaf_ranks, perNode, block, 0, stream, nd, N, n_prims, (const uint4*)at(o_up), (uint32_t*)out.d_rank8,
                     (uint32_t*)at(o_order0), ctl);
  LT_PREP_CHECK(hipGetLastError());
  uint32_t h_ctl[kCtlWords];
  LT_PREP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, stream));
  LT_PREP_CHECK(hipStreamSynchronize(stream));
  out.flags = h_ctl[kCtlFlags];
  out.bvh_height = (int)h_ctl[kCtlBvhHeight];
  out.ms_check = (float)(now_ms() - t0);
  if (out.flags) {
    al.put(al.self, scratch);
    release(out, al);
    return hipSuccess;
  }

  // ---- the own tree
  const double t1 = now_ms();
  out.n_own = 2 * n - 1;
  LT_PREP_CHECK(al.get(al.self, &out.d_nodes2, (size_t)out.n_own * 32));
  float4* own = (float4*)out.d_nodes2;
  if (!ownSplits) {
    // the caller's splits: a proper pre-order tree without unreachable nodes IS lt_retree::copy's output
    if (out.bvh_height > maxHeight) { out.flags = kFlagNoRoom; al.put(al.self, scratch); release(out, al); return hipSuccess; }
    LT_PREP_CHECK(hipMemcpyAsync(own, d_nodes, (size_t)N * 32, hipMemcpyDeviceToDevice, stream));
    out.own_height = out.bvh_height;
  } else {
    Range* big[2] = {(Range*)at(o_big0), (Range*)at(o_big1)};
    Range* mid[2] = {(Range*)at(o_mid0), (Range*)at(o_mid1)};
    Range* tiny[2] = {(Range*)at(o_tiny0), (Range*)at(o_tiny1)};
    uint32_t* order[2] = {(uint32_t*)at(o_order0), (uint32_t*)at(o_order1)};
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, stream, big[0], mid[0], tiny[0], counts, n, (uint32_t*)nullptr, (uint32_t*)nullptr);
    uint32_t h_counts[kCntWords] = {n > kChunk ? 1u : 0u, (n > kTiny && n <= kChunk) ? 1u : 0u, n <= kTiny ? 1u : 0u, 0u};
    int level = 0;
    for (; level < maxLevels; level++) {
      const uint32_t nb = h_counts[kCntBig], nm = h_counts[kCntMid], nt = h_counts[kCntTiny];
      if (nb == 0 && nm == 0 && nt == 0) break;
      if (nb > maxBig || nm + nb > maxMid || nt > maxTiny) { out.flags = kFlagInternal; break; }
      const int cur = level & 1, nxt = cur ^ 1;
      uint32_t* curCounts = counts + (size_t)level * kCntWords;
      uint32_t* nextCounts = counts + (size_t)(level + 1) * kCntWords;
      if (nb) {
        const dim3 chunks(n / kChunk + nb + 1);
        hipLaunchKernelGGL(k_big_setup, dim3(nb), block, 0, stream, big[cur], nb, (uint32_t*)at(o_chunkBase), (uint32_t*)at(o_chunkRange),
                           (uint32_t*)at(o_acc), (uint32_t*)at(o_bins), ctl);
        hipLaunchKernelGGL(k_big_bounds, chunks, block, 0, stream, nd, order[cur], big[cur], (const uint32_t*)at(o_chunkBase),
                           (const uint32_t*)at(o_chunkRange), (uint32_t*)at(o_acc), ctl);
        hipLaunchKernelGGL(k_big_bins, chunks, block, 0, stream, nd, order[cur], big[cur], (const uint32_t*)at(o_chunkBase),
                           (const uint32_t*)at(o_chunkRange), (const uint32_t*)at(o_acc), (uint32_t*)at(o_bins), ctl);
        hipLaunchKernelGGL(k_big_split, chunks, block, 0, stream, nd, order[cur], big[cur], (const uint32_t*)at(o_chunkBase),
                           (const uint32_t*)at(o_chunkRange), (const uint32_t*)at(o_acc), (const uint32_t*)at(o_bins), (SplitRec*)at(o_splits),
                           (uint32_t*)at(o_chunkLeft), own, heightLimit, mid[cur], curCounts, big[nxt], mid[nxt], tiny[nxt], nextCounts, ctl);
        hipLaunchKernelGGL(k_big_scatter, chunks, block, 0, stream, nd, order[cur], order[nxt], big[cur], (const uint32_t*)at(o_chunkBase),
                           (const uint32_t*)at(o_chunkRange), (const SplitRec*)at(o_splits), (const uint32_t*)at(o_chunkLeft), own, ctl);
      }
      if (nm + nb)
        hipLaunchKernelGGL(k_mid, dim3((nm + nb + kMidBlock / 64 - 1) / (kMidBlock / 64)), dim3(kMidBlock), 0, stream, nd, order[cur], order[nxt], mid[cur], curCounts, own, heightLimit,
                           big[nxt], mid[nxt], tiny[nxt], nextCounts, ctl);
      if (nt)
        hipLaunchKernelGGL(k_tiny, dim3((nt + 3) / 4), block, 0, stream, nd, order[cur], tiny[cur], curCounts, own, heightLimit, ctl);
      LT_PREP_CHECK(hipGetLastError());
      LT_PREP_CHECK(hipMemcpyAsync(h_counts, nextCounts, sizeof(h_counts), hipMemcpyDeviceToHost, stream));
      LT_PREP_CHECK(hipStreamSynchronize(stream));
    }
    out.levels = level;
    if (level == maxLevels) out.flags |= kFlagInternal;
    LT_PREP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, stream));
    LT_PREP_CHECK(hipStreamSynchronize(stream));
    out.flags |= h_ctl[kCtlFlags];
    out.own_height = (int)h_ctl[kCtlOwnHeight];
    if (out.own_height > heightLimit) out.flags |= kFlagInternal;
    if (out.flags) { al.put(al.self, scratch); release(out, al); return hipSuccess; }
  }
  out.ms_build = (float)(now_ms() - t1);

  // ---- the 4-wide groups
  const double t2 = now_ms();
  LT_PREP_CHECK(al.get(al.self, &out.d_groupOf, (size_t)out.n_own * 4));
  LT_PREP_CHECK(al.get(al.self, &out.d_children, (size_t)(n - 1) * 16));
  LT_PREP_CHECK(hipMemsetAsync(out.d_groupOf, 0xff, (size_t)out.n_own * 4, stream));
  LT_PREP_CHECK(hipMemsetAsync(counts, 0, (size_t)(maxLevels + 2) * kCntWords * 4, stream));
  uint32_t* front[2] = {(uint32_t*)at(o_front0), (uint32_t*)at(o_front1)};
  hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, stream, (Range*)nullptr, (Range*)nullptr, (Range*)nullptr, counts, 0u, front[0], counts);
  uint32_t width = 1;
  int wlevel = 0;
  for (; wlevel < maxLevels; wlevel++) {
    if (width == 0) break;
    if (width > n) { out.flags = kFlagInternal; break; }
    uint32_t* curCounts = counts + (size_t)wlevel * kCntWords;
    uint32_t* nextCounts = counts + (size_t)(wlevel + 1) * kCntWords;
    hipLaunchKernelGGL(k_wide_mark, dim3((width + 255) / 256), block, 0, stream, (const lt_retree::Node*)own, front[wlevel & 1], curCounts,
                       front[(wlevel & 1) ^ 1], nextCounts, (uint32_t*)out.d_groupOf);
    LT_PREP_CHECK(hipGetLastError());
    uint32_t h_next[kCntWords];
    LT_PREP_CHECK(hipMemcpyAsync(h_next, nextCounts, sizeof(h_next), hipMemcpyDeviceToHost, stream));
    LT_PREP_CHECK(hipStreamSynchronize(stream));
    width = h_next[kCntWide];
  }
  if (wlevel == maxLevels) out.flags |= kFlagInternal;
  out.wide_height = wlevel - 1;
  const uint32_t scanBlocks = (out.n_own + 2047) / 2048;
  hipLaunchKernelGGL(k_scan_sums, dim3(scanBlocks), block, 0, stream, (const uint32_t*)out.d_groupOf, out.n_own, (uint32_t*)at(o_sums));
  hipLaunchKernelGGL(k_scan_top, dim3(1), block, 0, stream, (uint32_t*)at(o_sums), scanBlocks, ctl);
  hipLaunchKernelGGL(k_scan_apply, dim3(scanBlocks), block, 0, stream, (uint32_t*)out.d_groupOf, out.n_own, (const uint32_t*)at(o_sums));
  hipLaunchKernelGGL(k_wide_children, dim3((out.n_own + 255) / 256), block, 0, stream, (const lt_retree::Node*)own, out.n_own,
                     (const uint32_t*)out.d_groupOf, (uint32_t*)out.d_children);
  LT_PREP_CHECK(hipGetLastError());
  float rootBox[8];
  LT_PREP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, stream));
  LT_PREP_CHECK(hipMemcpyAsync(rootBox, own, sizeof(rootBox), hipMemcpyDeviceToHost, stream));
  LT_PREP_CHECK(hipStreamSynchronize(stream));
  out.groups = h_ctl[kCtlGroups];
  for (int k = 0; k < 3; k++) { out.root_lo[k] = rootBox[k]; out.root_hi[k] = rootBox[3 + k]; }
  out.ms_wide = (float)(now_ms() - t2);
  al.put(al.self, scratch);
  scratch = nullptr;
  if (out.flags) release(out, al);
  return hipSuccess;
}

}  // namespace lt_prep
