// The backend's own traversal tree over the reference's leaves: the host-side build (lt_prep.hip makes the same tree with kernels
// at lt_hip_set_scene; this one serves the scenes that path declines, small scenes, the diagnostic entry points and the tests).
//
// Why a different tree gives the same pixels.  The reference's traversal (acc.cl:132-217) reaches a leaf iff the ray passes the
// slab test (acc.cl:113-130) of every ancestor's box and of the leaf's own box, and it never clips against the closest hit.
// For a ray whose origin and inverse direction are finite the slab test is monotonic in the box: (bound - o) * inv is a
// monotonic function of `bound` in float arithmetic (rounding is monotonic), so enlarging a box can only lower the largest
// entry distance and raise the smallest exit distance -- a ray that passes the test of a box passes the test of every box that
// encloses it.  When every node's box encloses its children's boxes (checked here, node by node, on the caller's buffer), "passes
// every ancestor and the leaf" is therefore the same set of leaves as "passes the leaf's own box", whatever hierarchy is used
// to enumerate candidates -- provided that hierarchy's boxes enclose their leaves too (they are unions of them) and the leaf's
// own box, bit for bit the reference's, is still tested.  The triangle tests that follow are the reference's; which of them run
// is the same set; what differs is the ORDER, which matters in one place only: two triangles whose accepted hits have exactly
// equal t (intersectTriangle accepts t < payload.t, so the first one tested stays).  Shadow rays do not care (their callers
// read hitType only); closest-hit walks resolve such ties with the reference's order (lt_device.hpp, `rank8`).
// Rays with a non-finite component, the counting kernels (whose per-ray node counts are the reference's) and scenes whose
// boxes do not nest keep walking the caller's tree.
//
// The tree: binned surface-area heuristic (32 bins per axis over the centroid bounds, all three axes), one reference leaf per
// leaf, the reference's 32-byte node layout and pre-order numbering (left child = i + 1), so that every walk and the
// child-pair kernel read it as they read the caller's buffer.  Without clipping, the expected number of nodes a ray visits is
// exactly the surface-area sum the heuristic minimises (32 bins are where that sum stops falling on the 1 M-triangle wall: 17.2
// box areas of the root per ray against the caller's 40.5).  Height is bounded (the packet walks' stack is one register's 64
// lanes, the per-lane walks' stack three entries per level of 4-wide groups).
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#if defined(__HIPCC__) || defined(__HIP__)
#define LT_RETREE_HD __host__ __device__
#else
#define LT_RETREE_HD
#endif

namespace lt_retree {

// Floats as unsigned integers in the same order (-0 below +0; NaN does not occur: the boxes are checked first).  Box unions are
// taken in this order on the host and on the device (lt_prep.hip: atomic min / max on these integers), so that the two builds
// agree on the sign of a zero bound too.
LT_RETREE_HD inline uint32_t ordered(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
LT_RETREE_HD inline float unordered(uint32_t o) {
  return __builtin_bit_cast(float, o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu));
}
LT_RETREE_HD inline float tmin(float a, float b) { return ordered(b) < ordered(a) ? b : a; }
LT_RETREE_HD inline float tmax(float a, float b) { return ordered(b) > ordered(a) ? b : a; }

struct Node {   // LinearBVHNode (include/lens_trace/acceleration_structure_explicit.h:20-32)
  float lo[3], hi[3];
  int32_t off;
  uint16_t cnt;
  uint8_t axis, pad;
};
static_assert(sizeof(Node) == 32, "LinearBVHNode is 32 bytes");

#ifndef LT_RETREE_BINS
#define LT_RETREE_BINS 32
#endif
constexpr int kBins = LT_RETREE_BINS;

// CPUs this process may run on (its affinity mask: a container's share, not the machine's core count)
inline int available_cpus() {
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) return std::max(1, CPU_COUNT(&set));
  return (int)std::max(1u, std::thread::hardware_concurrency());
}

inline int ceil_log2(uint32_t x) {
  int l = 0;
  while ((1ull << l) < x) l++;
  return l;
}

// Reachable leaves of the caller's tree in its own pre-order, or false when the boxes do not nest (or hold a NaN).
inline bool collect_leaves(const Node* nd, uint32_t n_nodes, std::vector<uint32_t>& leaves) {
  leaves.clear();
  for (int a = 0; a < 3; a++)   // every box lies inside the root's: all bounds below 2^40 (the packet walks' conservative test, lt_walk_asm.hpp)
    if (!(nd[0].lo[a] > -0x1p+40f && nd[0].hi[a] < 0x1p+40f && nd[0].lo[a] <= nd[0].hi[a])) return false;
  std::vector<uint32_t> stack{0u};
  while (!stack.empty()) {
    const uint32_t i = stack.back();
    stack.pop_back();
    const Node& p = nd[i];
    if (p.cnt != 0) { leaves.push_back(i); continue; }
    const uint32_t kids[2] = {i + 1, (uint32_t)p.off};
    for (uint32_t k : kids) {
      const Node& c = nd[k];
      for (int a = 0; a < 3; a++)
        if (!(c.lo[a] >= p.lo[a] && c.hi[a] <= p.hi[a] && c.lo[a] <= c.hi[a])) return false;
    }
    stack.push_back(kids[1]);
    stack.push_back(kids[0]);
    if (leaves.size() + stack.size() > n_nodes) return false;   // (a shared child: not a tree)
  }
  return true;
}

struct Range { uint32_t start, end, node; int depth; };

// Builds the subtree of the leaves order[start, end) at out[node ...] (a pre-order subtree of k leaves owns out[node, node + 2k - 1)
// and order[start, end): nothing is shared between subtrees); returns the deepest level reached.  With `deferred`, subtrees of at
// most `deferCount` leaves below the root are not built but listed there, for other threads.
inline int build_range(const Node* nd, const float* centroid, std::vector<uint32_t>& order, Node* out, Range root, int heightLimit,
                       std::vector<Range>* deferred = nullptr, uint32_t deferCount = 0) {
  int height = root.depth;
  std::vector<Range> work{root};
  struct Bin { float lo[3], hi[3]; uint32_t count; };
  while (!work.empty()) {
    const Range r = work.back();
    work.pop_back();
    if (deferred && r.node != root.node && r.end - r.start <= deferCount) { deferred->push_back(r); continue; }
    Node& node = out[r.node];
    height = std::max(height, r.depth);
    const uint32_t count = r.end - r.start;
    if (count == 1) {
      node = nd[order[r.start]];   // the reference's leaf, box and offset bit for bit
      continue;
    }
    float cmin[3], cmax[3];
    for (int a = 0; a < 3; a++) {
      node.lo[a] = cmin[a] = std::numeric_limits<float>::max();
      node.hi[a] = cmax[a] = -std::numeric_limits<float>::max();
    }
    for (uint32_t i = r.start; i < r.end; i++) {
      const Node& p = nd[order[i]];
      const float* c = centroid + 3 * (size_t)order[i];
      for (int a = 0; a < 3; a++) {
        node.lo[a] = tmin(node.lo[a], p.lo[a]);
        node.hi[a] = tmax(node.hi[a], p.hi[a]);
        cmin[a] = tmin(cmin[a], c[a]);
        cmax[a] = tmax(cmax[a], c[a]);
      }
    }
    const float d[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
    int dim = (d[0] > d[1] && d[0] > d[2]) ? 0 : (d[1] > d[2] ? 1 : 2);
    uint32_t mid = r.start + count / 2;
    bool split = false;
    // a child of k leaves needs ceil(log2 k) more levels at least: both children must fit under the height limit
    const int room = heightLimit - r.depth - 1;
    const uint32_t maxChild = room >= 31 ? count : (uint32_t)std::min<uint64_t>(count, 1ull << std::max(0, room));
    if (count > 2 && maxChild >= (count + 1) / 2) {
      Bin bins[3][kBins];
      for (int a = 0; a < 3; a++)
        for (Bin& b : bins[a]) {
          for (int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<float>::max(); b.hi[k] = -std::numeric_limits<float>::max(); }
          b.count = 0;
        }
      float scale[3];
      for (int a = 0; a < 3; a++) scale[a] = d[a] > 0.0f ? (float)kBins / d[a] : 0.0f;
      for (uint32_t i = r.start; i < r.end; i++) {
        const Node& p = nd[order[i]];
        const float* c = centroid + 3 * (size_t)order[i];
        for (int a = 0; a < 3; a++) {
          if (!(d[a] > 0.0f)) continue;
          Bin& b = bins[a][std::min(kBins - 1, std::max(0, (int)((c[a] - cmin[a]) * scale[a])))];
          b.count++;
          for (int k = 0; k < 3; k++) { b.lo[k] = tmin(b.lo[k], p.lo[k]); b.hi[k] = tmax(b.hi[k], p.hi[k]); }
        }
      }
      auto half_area = [](const float* lo, const float* hi) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
      };
      float bestCost = std::numeric_limits<float>::max();
      int bestDim = -1, bestBin = -1;
      for (int a = 0; a < 3; a++) {
        if (!(d[a] > 0.0f)) continue;
        float rightArea[kBins];
        uint32_t rightCount[kBins];
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); }
        uint32_t c = 0;
        for (int b = kBins - 1; b >= 0; b--) {
          const Bin& bn = bins[a][b];
          if (bn.count) for (int k = 0; k < 3; k++) { lo[k] = tmin(lo[k], bn.lo[k]); hi[k] = tmax(hi[k], bn.hi[k]); }
          c += bn.count;
          rightCount[b] = c;
          rightArea[b] = c ? half_area(lo, hi) : 0.0f;
        }
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); }
        c = 0;
        for (int b = 0; b + 1 < kBins; b++) {   // split after bin b
          const Bin& bn = bins[a][b];
          if (bn.count) for (int k = 0; k < 3; k++) { lo[k] = tmin(lo[k], bn.lo[k]); hi[k] = tmax(hi[k], bn.hi[k]); }
          c += bn.count;
          const uint32_t rc = rightCount[b + 1];
          if (c == 0 || rc == 0 || c > maxChild || rc > maxChild) continue;
          const float cost = half_area(lo, hi) * (float)c + rightArea[b + 1] * (float)rc;
          if (cost < bestCost) { bestCost = cost; bestDim = a; bestBin = b; }
        }
      }
      if (bestDim >= 0) {
        const float sc = scale[bestDim], c0 = cmin[bestDim];
        auto it = std::stable_partition(order.begin() + r.start, order.begin() + r.end, [&](uint32_t i) {
          return std::min(kBins - 1, std::max(0, (int)((centroid[3 * (size_t)i + bestDim] - c0) * sc))) <= bestBin;
        });
        mid = (uint32_t)(it - order.begin());
        dim = bestDim;
        split = true;
      }
    }
    if (!split && d[dim] > 0.0f) {
      // median of the largest centroid extent (no plane separates the centroids under the height limit, or two leaves): the
      // count / 2 leaves with the smallest (centroid, leaf index) go left, both halves in the order they stand in -- a function
      // of the set alone and of that order, which the device build reproduces with a selection and a stable partition
      auto key = [&](uint32_t i) { return ((uint64_t)ordered(centroid[3 * (size_t)i + dim]) << 32) | i; };
      std::vector<uint64_t> keys(count);
      for (uint32_t i = 0; i < count; i++) keys[i] = key(order[r.start + i]);
      std::nth_element(keys.begin(), keys.begin() + count / 2, keys.end());
      const uint64_t pivot = keys[count / 2];
      std::stable_partition(order.begin() + r.start, order.begin() + r.end, [&](uint32_t i) { return key(i) < pivot; });
    }
    node.axis = (uint8_t)dim;
    node.cnt = 0;
    node.pad = 0;
    const uint32_t leftCount = mid - r.start;
    node.off = (int32_t)(r.node + 2 * leftCount);   // r.node + 1 + (2 * leftCount - 1)
    work.push_back({mid, r.end, (uint32_t)node.off, r.depth + 1});
    work.push_back({r.start, mid, r.node + 1, r.depth + 1});
  }
  return height;
}

// The caller's own hierarchy, re-emitted in pre-order without its unreachable nodes (LT_RETREE=0: same walks, the caller's
// splits).  Returns the height or -1 as build() does.
inline int copy(const void* nodes, uint32_t n_nodes, int maxHeight, std::vector<Node>& out) {
  const Node* nd = reinterpret_cast<const Node*>(nodes);
  std::vector<uint32_t> leaves;
  if (!collect_leaves(nd, n_nodes, leaves) || leaves.size() < 2) return -1;
  if ((uint64_t)leaves.size() * 2 - 1 > 0x03ffffffull) return -1;
  out.assign(2 * leaves.size() - 1, Node{});
  struct Item { uint32_t src, parent; int depth; bool right; };
  std::vector<Item> stack{{0u, 0xffffffffu, 0, false}};
  uint32_t next = 0;
  int height = 0;
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const uint32_t at = next++;
    out[at] = nd[it.src];
    if (it.right) out[it.parent].off = (int32_t)at;
    height = std::max(height, it.depth);
    if (nd[it.src].cnt == 0) {
      stack.push_back({(uint32_t)nd[it.src].off, at, it.depth + 1, true});
      stack.push_back({it.src + 1, at, it.depth + 1, false});
    }
  }
  return height <= maxHeight ? height : -1;
}

// out: 2 * leaves - 1 nodes.  Returns the height (interior ancestors of the deepest leaf), or -1 when no tree is built (boxes do
// not nest, fewer than two leaves, or the tree cannot fit under maxHeight).
inline int build(const void* nodes, uint32_t n_nodes, int maxHeight, int slack, std::vector<Node>& out, int threads = 0) {
  const Node* nd = reinterpret_cast<const Node*>(nodes);
  std::vector<uint32_t> order;
  if (!collect_leaves(nd, n_nodes, order) || order.size() < 2) return -1;
  const uint32_t n = (uint32_t)order.size();
  if ((uint64_t)n * 2 - 1 > 0x03ffffffull) return -1;   // (64-byte pair records behind 32-bit byte offsets)
  const int heightLimit = std::min(maxHeight, ceil_log2(n) + slack);
  if (heightLimit < ceil_log2(n)) return -1;
  std::vector<float> centroid(3 * (size_t)n_nodes);
  for (uint32_t i : order)
    for (int a = 0; a < 3; a++) centroid[3 * (size_t)i + a] = 0.5f * nd[i].lo[a] + 0.5f * nd[i].hi[a];
  out.assign(2 * (size_t)n - 1, Node{});
  // the top of the tree on this thread until there are enough independent subtrees for the others (a pre-order subtree of k
  // leaves owns out[node, node + 2k - 1) and order[start, end): nothing is shared)
  if (threads <= 0) {
    const char* e = getenv("LT_RETREE_THREADS");
    threads = e ? std::max(1, std::min(64, atoi(e))) : std::min(16, available_cpus());
  }
  if (n < 50000u) threads = 1;
  if (threads == 1) return build_range(nd, centroid.data(), order, out.data(), Range{0, n, 0, 0}, heightLimit);
  // A queue of ranges: a large range gets its own node split by whoever takes it, and its two halves go back on the queue (one
  // thread at the root, two below it, four ...); a range of at most n / (8 threads) leaves is built to the bottom by one thread.
  // Every split is the one the single-threaded build makes, so the tree does not depend on the thread count.
  const uint32_t whole = n / (8u * (uint32_t)threads) + 1u;
  std::mutex mx;
  std::condition_variable cv;
  std::vector<Range> queue{Range{0, n, 0, 0}};
  int busy = 0, height = 0;
  auto worker = [&]() {
    std::unique_lock<std::mutex> lk(mx);
    for (;;) {
      cv.wait(lk, [&]() { return !queue.empty() || busy == 0; });
      if (queue.empty()) return;   // (nothing queued and nobody who could queue anything: done)
      const Range r = queue.back();
      queue.pop_back();
      busy++;
      lk.unlock();
      std::vector<Range> kids;
      int h;
      if (r.end - r.start <= whole) h = build_range(nd, centroid.data(), order, out.data(), r, heightLimit);
      else h = build_range(nd, centroid.data(), order, out.data(), r, heightLimit, &kids, r.end - r.start - 1u);   // this node only
      lk.lock();
      height = std::max(height, h);
      for (const Range& k : kids) queue.push_back(k);
      busy--;
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
  } catch (...) {   // (no more threads to be had -- a pids limit, eight ranks on one host: whoever exists builds; the tree is the same)
  }
  worker();
  for (std::thread& th : pool) th.join();
  return height;
}

// The own tree collapsed into 4-wide groups for the per-lane walks (lt_device.hpp, traverse_own_lane): a group holds up to four
// nodes of the binary tree -- the two children of an interior node, the one with the largest box replaced by ITS two children,
// and once more -- and stands where that interior node stood.  A per-lane walk is a chain of dependent memory round trips, one
// per visited record; four boxes per record instead of one make the chain four times shorter (1 M-triangle wall: 23 group
// visits per ray against 94 node visits; soup 73 against 294).  Groups are numbered in depth-first order (a group's first
// interior child follows it), leaves sit in the last slots of their group (pushed last, popped first).
//   children[4 g + k] : binary node in slot k of group g, or 0xffffffff
//   groupOf[b]        : the group that replaced interior binary node b, or 0xffffffff (b is a leaf, or was dissolved into its
//                       parent's group)
// Returns the height of the group tree (groups above the deepest group), or -1 when two leaves refer to the same primitive
// (the leaf records of the walk are indexed by primitive offset).
// (the up to four nodes of the group that stands for interior node b; returns how many)
LT_RETREE_HD inline int wide_kids(const Node* own, uint32_t b, uint32_t kids[4]) {
  auto half_area = [&](uint32_t i) {
    const Node& p = own[i];
    const float dx = p.hi[0] - p.lo[0], dy = p.hi[1] - p.lo[1], dz = p.hi[2] - p.lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  kids[0] = b + 1u; kids[1] = (uint32_t)own[b].off; kids[2] = kids[3] = 0xffffffffu;
  int count = 2;
  while (count < 4) {
    int best = -1;
    float bestArea = -1.0f;
    for (int k = 0; k < count; k++)
      if (own[kids[k]].cnt == 0) {
        const float a = half_area(kids[k]);
        if (a > bestArea) { bestArea = a; best = k; }   // (the first of equals: deterministic)
      }
    if (best < 0) break;
    const uint32_t d = kids[best];
    kids[best] = d + 1u;
    kids[count++] = (uint32_t)own[d].off;
  }
  // interior children first, leaves last, each kind in the order it stands in (a stable partition of at most four)
  uint32_t sorted[4];
  int m = 0;
  for (int k = 0; k < count; k++) if (own[kids[k]].cnt == 0) sorted[m++] = kids[k];
  for (int k = 0; k < count; k++) if (own[kids[k]].cnt != 0) sorted[m++] = kids[k];
  for (int k = 0; k < count; k++) kids[k] = sorted[k];
  return count;
}
inline int wide_kids(const std::vector<Node>& own, uint32_t b, uint32_t kids[4]) { return wide_kids(own.data(), b, kids); }

inline int collapse_wide(const std::vector<Node>& own, uint32_t n_prims, std::vector<uint32_t>& children, std::vector<uint32_t>& groupOf, int threads = 0) {
  const uint32_t n = (uint32_t)own.size();
  children.clear();
  groupOf.assign(n, 0xffffffffu);
  {
    std::vector<bool> seen(n_prims, false);
    for (const Node& nd : own)
      if (nd.cnt != 0) {
        if ((uint32_t)nd.off >= n_prims || seen[(uint32_t)nd.off]) return -1;
        seen[(uint32_t)nd.off] = true;
      }
  }
  if (threads <= 0) {
    const char* e = getenv("LT_RETREE_THREADS");
    threads = e ? std::max(1, std::min(64, atoi(e))) : std::min(16, available_cpus());
  }
  if (n < 100000u) threads = 1;
  // 1. which interior nodes get a group of their own (the others are dissolved into their parent's), top down: a queue of
  //    subtrees [node, end) -- in pre-order a subtree is a range of indices -- large ones handed on, small ones walked to the
  //    bottom by whoever takes them.  groupOf[b] = 0 marks a group's node for now.
  struct Task { uint32_t node, end; int depth; };
  std::mutex mx;
  std::condition_variable cv;
  std::vector<Task> queue{Task{0u, n, 0}};
  int busy = 0, height = 0;
  const uint32_t grain = std::max(4096u, n / (8u * (uint32_t)threads));
  auto worker = [&]() {
    std::unique_lock<std::mutex> lk(mx);
    for (;;) {
      cv.wait(lk, [&]() { return !queue.empty() || busy == 0; });
      if (queue.empty()) return;
      const Task first = queue.back();
      queue.pop_back();
      busy++;
      lk.unlock();
      std::vector<Task> local{first}, handOn;
      int h = 0;
      while (!local.empty()) {
        const Task t = local.back();
        local.pop_back();
        h = std::max(h, t.depth);
        groupOf[t.node] = 0u;
        uint32_t kids[4];
        const int count = wide_kids(own, t.node, kids);
        for (int k = 0; k < count; k++) {
          if (own[kids[k]].cnt != 0) continue;
          // the child's subtree ends where the next node of the group (in index order) starts, or where the group's does
          uint32_t end = t.end;
          for (int j = 0; j < count; j++)
            if (kids[j] > kids[k] && kids[j] < end) end = kids[j];
          const Task c{kids[k], end, t.depth + 1};
          if (threads > 1 && c.end - c.node > grain && t.end - t.node > 2u * grain) handOn.push_back(c); else local.push_back(c);
        }
      }
      lk.lock();
      height = std::max(height, h);
      for (const Task& t : handOn) queue.push_back(t);
      busy--;
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
  } catch (...) {
  }
  worker();
  for (std::thread& th : pool) th.join();
  pool.clear();
  // 2. groups numbered in the order of their nodes (pre-order: a group's first interior child follows it closely)
  uint32_t groups = 0;
  for (uint32_t b = 0; b < n; b++)
    if (groupOf[b] == 0u) groupOf[b] = groups++;
  // 3. the groups' slots
  children.assign(4 * (size_t)groups, 0xffffffffu);
  auto fill = [&](uint32_t lo, uint32_t hi) {
    for (uint32_t b = lo; b < hi; b++) {
      if (groupOf[b] == 0xffffffffu) continue;
      uint32_t kids[4];
      const int count = wide_kids(own, b, kids);
      for (int k = 0; k < count; k++) children[4 * (size_t)groupOf[b] + k] = kids[k];
    }
  };
  try {
    for (int t = 1; t < threads; t++) pool.emplace_back(fill, (uint32_t)((uint64_t)n * t / threads), (uint32_t)((uint64_t)n * (t + 1) / threads));
  } catch (...) {
  }
  fill(0u, (uint32_t)((uint64_t)n / threads));
  const size_t started = pool.size();
  for (std::thread& th : pool) th.join();
  for (int t = 1 + (int)started; t < threads; t++) fill((uint32_t)((uint64_t)n * t / threads), (uint32_t)((uint64_t)n * (t + 1) / threads));
  return height;
}

// rank8[8 * primitive + octant]: position of the primitive's (first) leaf in the caller's tree walked depth-first, near child
// first, by a ray whose direction signs are `octant` (bit a = component a negative: the reference's dirIsNeg[node->axis],
// acc.cl:150-160).  0xffffffff for primitives no leaf refers to.
// In closed form: a leaf's position is the number of leaves the walk meets before it -- for every ancestor whose FAR child (for
// that octant) the leaf lies under, the leaves under the near child.  The tree is in pre-order (children behind their parent), so
// one backward pass counts the leaves under every node and one forward pass hands every node the eight positions of its first
// leaf: no walks, no threads.  (Nodes the root does not reach keep no position, as in a walk.)
inline void reference_order(const void* nodes, uint32_t n_nodes, uint32_t n_prims, std::vector<uint32_t>& rank8) {
  const Node* nd = reinterpret_cast<const Node*>(nodes);
  rank8.assign(8 * (size_t)n_prims, 0xffffffffu);
  if (n_nodes == 0) return;
  std::vector<uint32_t> leaves(n_nodes);
  for (uint32_t i = n_nodes; i-- > 0;) leaves[i] = nd[i].cnt != 0 ? 1u : leaves[i + 1] + leaves[(uint32_t)nd[i].off];
  struct Base { uint32_t r[8]; };
  std::vector<Base> base(n_nodes);
  std::vector<uint8_t> reached(n_nodes, 0);
  for (int o = 0; o < 8; o++) base[0].r[o] = 0u;
  reached[0] = 1;
  for (uint32_t i = 0; i < n_nodes; i++) {
    if (!reached[i]) continue;
    const Node& p = nd[i];
    if (p.cnt != 0) {
      uint32_t* r = &rank8[8 * (size_t)(uint32_t)p.off];
      for (int o = 0; o < 8; o++)
        if (base[i].r[o] < r[o]) r[o] = base[i].r[o];   // (two leaves on one primitive: the first one the walk meets)
      continue;
    }
    const uint32_t left = i + 1, right = (uint32_t)p.off;
    for (int o = 0; o < 8; o++) {
      const bool neg = ((o >> p.axis) & 1) != 0;   // the right child is the near one
      base[left].r[o] = base[i].r[o] + (neg ? leaves[right] : 0u);
      base[right].r[o] = base[i].r[o] + (neg ? 0u : leaves[left]);
    }
    reached[left] = reached[right] = 1;
  }
}

}  // namespace lt_retree
