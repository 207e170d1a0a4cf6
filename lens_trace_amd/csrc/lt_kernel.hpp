// lt_kernel.hpp -- the render kernel of liblenstrace-hip.so, as a device-side header: included by lt_capi.hip for the six
// built-in programs and by the run-time compiled translation unit of a user program (lt_hip_resolve_program ->
// hipRTC), whose `user_shade` it calls (the reference JIT-compiles a user's kernel *file*,
// src/opencl/renderer_opencl.cpp:35-54; here the user supplies the shade step in HIP and gets the built-in traversal).
#pragma once
#include "lt_device.hpp"

using namespace lt;

// One lane per pixel; a workgroup is ONE wavefront covering an 8x8 pixel square (no intra-workgroup tail: the
// LDS stack and the wave slot are released as soon as that wave's slowest ray ends).  Workgroup ids are
// remapped so that the blocks one XCD receives (ids congruent mod 8) cover one contiguous part of the
// image: each XCD's private L2 then holds the BVH subtrees of its own image region.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
  const uint32_t q = n / 8u, r = n % 8u, xcd = b % 8u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + b / 8u;
}

// One 8x8 pixel square (logical index b, already XCD-ordered) by one wavefront.
// (pos: the square's position in the hand-out order -- what a queued shadow ray's slot is made of, SceneDev::shadowPackets == 3)
template <int PROGRAM, class CFG>
__device__ __forceinline__ void render_square(const SceneDev& sc, const FrameParams& fp, float* __restrict__ out, uint32_t b,
                                              uint32_t frame, Stack<CFG::kDeep>& st, Counters& c, uint32_t pos) {
  constexpr bool STATS = CFG::kStats;
  const uint32_t k = b / fp.blocksPerTile, sb = b % fp.blocksPerTile;
  const uint32_t sbx = sb % fp.blocksPerTileX, sby = sb / fp.blocksPerTileX;
  const uint32_t tile = fp.tileFirst + k * fp.tileStride;
  const uint32_t tx = tile % fp.tilesX, ty = tile / fp.tilesX;
  const uint32_t lane = threadIdx.x;
  const uint32_t lx = sbx * 8u + (lane & 7u);
  const uint32_t ly = sby * 8u + (lane >> 3);
  const uint32_t x = tx * fp.tileW + lx, y = ty * fp.tileH + ly;
  const bool valid = k < fp.tilesInCall && lx < fp.tileW && ly < fp.tileH && x < fp.width && y < fp.height;
  constexpr bool queueing = PROGRAM == kAccumulatorQueue;
  const uint32_t qslot = (pos * fp.fusedFrames + frame) * (uint32_t)kBlock + lane;
  bool queued = false;
  if (valid) {
    Counters pc{};   // this pixel's own counters (diagnostic output), folded into the lane's totals below
    const uint32_t pixel = (k * fp.tileH + ly) * fp.tileW + lx;
    const V3 color = shade_pixel<PROGRAM, CFG>(sc, fp, fp.frameCount + frame, (int)x, (int)y, st, STATS ? pc : c, qslot, pixel, frame, queued);
    float* o = out + (size_t)frame * fp.frameStride + (((size_t)k * fp.tileH + ly) * fp.tileW + lx) * fp.depth;
    if (STATS && fp.pixelCounters) {
      o[0] = (float)pc.rays; o[1] = (float)pc.shadow; o[2] = (float)pc.nodes; o[3] = (float)pc.tris;
    } else if (fp.accumulateN <= 0) {   // overwrite, or first frame of a running mean (`if (frameCount > 0)` guard)
      o[0] = color.x; o[1] = color.y; o[2] = color.z;
    } else {                     // accumulator.frag:12-18: (c + acc*n) / (n+1)
      const float n = (float)fp.accumulateN, n1 = (float)(fp.accumulateN + 1);
      o[0] = (color.x + (o[0] * n)) / n1;
      o[1] = (color.y + (o[1] * n)) / n1;
      o[2] = (color.z + (o[2] * n)) / n1;
    }
    if (STATS) {
      c.rays += pc.rays; c.shadow += pc.shadow; c.nodes += pc.nodes; c.tris += pc.tris;
#ifdef LT_DEBUG_WAVE_COUNTERS
      c.wInner += pc.wInner; c.wTri += pc.wTri; c.wOuter += pc.wOuter;
#endif
    }
  }
  // (every slot of the square says what it holds -- all 16 bytes of it: a wave's dwords 16 bytes apart would be sixteen
  // partly written lines instead of four whole ones)
  if (queueing && !queued) ((uint4*)sc.shadowQueue)[2 * (size_t)sc.shadowCap + qslot] = make_uint4(kDeadSlot, 0u, 0u, 0u);
}

// Registers: the traversal wants every wave slot (8 per SIMD = 64 VGPRs); the single-bounce programs fit that with a few
// spilled values in their shading code; the 16-bounce / 25-sample programs as ONE kernel would spill 85-140 values at 8 and
// run best at 5 waves per SIMD (Cornell GI 1080p, 16 bounces, 16 frames per launch: 31.5 / 30.2 / 32.8 / 45.7 ms at
// 4 / 5 / 6 / 8); the stage kernels of the wavefront GI pipeline hold one ray's state and take 8 (1 M-triangle wall, 4K,
// 16 frames: 498 / 432 / 395 / 363 ms at 4 / 5 / 6 / 8; Cornell 24.0 -> 24.9 ms).  basic_lighting (one shadow ray, 25 samples
// per pixel) sides with the single-bounce programs: 8 waves are +9 % (blob) / +13 % (wall) over 5.
#ifndef LT_GI_WAVES
#define LT_GI_WAVES 5
#endif
#ifndef LT_GI_STAGE_WAVES
#define LT_GI_STAGE_WAVES 8
#endif
#ifndef LT_ACC_WAVES
#define LT_ACC_WAVES 8
#endif
constexpr int waves_per_simd(int program) { return (program == kGI || program == kGI25) ? LT_GI_WAVES : LT_ACC_WAVES; }

template <int PROGRAM, class CFG>
__device__ __forceinline__ void render_kernel_body(const SceneDev& sc, const FrameParams& fp, float* __restrict__ out,
                                                   unsigned long long* __restrict__ stats, uint32_t* __restrict__ queues) {
  extern __shared__ int lds_stack[];   // [BVH height (<= kLdsStack)][kBlock], sized by the launch
  constexpr bool STATS = CFG::kStats;
  Stack<CFG::kDeep> st;
  st.lds = lds_stack + threadIdx.x;
  st.rows = (int)fp.ldsRows;
  Counters c{};

  // Two ways to hand out the 8x8 squares, one loop (a single inlined copy of the renderer):
  //  * one square per workgroup, the hardware dispatcher doing the scheduling (workgroup ids remapped per XCD);
  //  * persistent wavefronts: a grid just large enough to fill the chip, every wave pulling squares from the queue of the
  //    XCD it runs on (its contiguous share of the logical square list, so each XCD's L2 keeps serving one image region)
  //    and, when that is drained, from the other XCDs' queues.  Every wave reaches the exit: each queue hands out at most
  //    its share, and the loop ends after one empty sweep over all eight.
  const uint32_t n = fp.totalSquares, q = n / 8u, r = n % 8u;
  const uint32_t home = fp.persistent ? (__builtin_amdgcn_s_getreg((3u << 11) | 20u) & 7u) : 0u;   // HW_REG_XCC_ID
  uint32_t sweep = 0;
  bool done = false;
  while (!done) {
    uint32_t b, frame = 0, pos = 0;
    if (!fp.persistent) {
      b = xcd_remap(blockIdx.x, gridDim.x);
      pos = b;
      done = true;
    } else {
      const uint32_t xcd = (home + sweep) & 7u;
      const uint32_t share = q + (xcd < r ? 1u : 0u), start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
      uint32_t t = 0;
      if (threadIdx.x == 0) t = atomicAdd(&queues[xcd * kQueueStride], 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      if (t >= share * fp.fusedFrames) {   // (share * fusedFrames < 2^32: checked by the host)
        done = ++sweep >= 8u;
        continue;
      }
      // frame-major inside the XCD's share (every frame of a square stays on its XCD), the head squares of all frames first
      const uint32_t head = fp.order ? fp.orderHead[xcd] : 0u;
      if (t < head * fp.fusedFrames) {
        frame = t / head;
        t -= frame * head;
      } else if (fp.fusedFrames > 1u) {
        t -= head * fp.fusedFrames;
        if (fp.squareMajor) {   // the frames of a square side by side: their rays meet the same nodes while those are in the L2
          frame = t % fp.fusedFrames;
          t = head + t / fp.fusedFrames;
        } else {
          frame = t / (share - head);
          t = head + (t - frame * (share - head));
        }
      }
      b = start + t;
      pos = b;
      if (fp.order) b = (uint32_t)__builtin_amdgcn_readfirstlane((int)fp.order[b]);
    }
    render_square<PROGRAM, CFG>(sc, fp, out, b, frame, st, c, pos);
  }
  if (STATS) {
    atomicAdd(&stats[0], (unsigned long long)c.rays);
    atomicAdd(&stats[1], (unsigned long long)c.shadow);
    atomicAdd(&stats[2], (unsigned long long)c.nodes);
    atomicAdd(&stats[3], (unsigned long long)c.tris);
#ifdef LT_DEBUG_WAVE_COUNTERS
    atomicAdd(&stats[4], (unsigned long long)c.wInner);
    atomicAdd(&stats[5], (unsigned long long)c.wTri);
    atomicAdd(&stats[6], (unsigned long long)c.wOuter);
#endif
  }
}

template <int PROGRAM, class CFG>
__global__ __launch_bounds__(kBlock, waves_per_simd(PROGRAM)) void lt_render_kernel(SceneDev sc, FrameParams fp, float* __restrict__ out,
                                                          unsigned long long* __restrict__ stats, uint32_t* __restrict__ queues) {
  render_kernel_body<PROGRAM, CFG>(sc, fp, out, stats, queues);
}

// =====================================================================================================================
// Wavefront pipeline for the global-illumination programs (gi.cl:241-375): the 16-bounce loop of `shade` as one kernel per
// stage with ACTIVE-PATH COMPACTION between stages, instead of one lane carrying a pixel through up to 34 rays while its
// neighbours idle (a path ends at the first occluded light sample or miss; on the Cornell box the mean is 3.9 rays per pixel,
// the maximum 34).
//   lt_gi_primary_kernel : camera ray, direct term, first extension ray      -> appends the surviving paths to queue 0
//   lt_gi_bounce_kernel  : extension ray d of every path in queue d, light sample + shadow ray, indirect term d
//                                                                             -> appends the surviving paths to queue d+1
//   lt_gi_resolve_kernel : direct + indirect, the 25-sample blend of the resources/ variant, clamp, running mean, store
// Queues are structure-of-float4 arrays in HBM (64 bytes per path: origin+film.x, direction, previous normal, {pixel,
// previous primitive, film.y}); a wave appends with one atomic: ballot of the surviving lanes, popcount, prefix count
// (mbcnt) for each lane's slot.  All kernels are persistent: a chip-filling grid whose waves pull work (8x8 squares from
// per-XCD queues / 64-path chunks) with an atomic, the queue length being read from device memory, so the host never waits
// between stages.  Per pixel the arithmetic and its order are those of the reference loop: the indirect sum receives its
// terms in depth order, and a path that hits a light adds the terms of all remaining depths at once (the reference
// re-traces the identical ray each time: same hit, same term).  Used when work is not being counted; the counting variants
// run the one-lane-per-pixel kernel, which re-traces like the reference does.
struct GiQueue { float4* o; float4* d; float4* n; uint4* m; };
constexpr uint32_t kDeadPath = kDeadSlot;    // m.x of a slot of a direct-mapped queue that holds no path

struct GiParams {
  GiQueue q[2];            // ping-pong: stage d reads q[d & 1], writes q[(d + 1) & 1]
  float4* direct;          // per pixel (compact output index): direct colour
  float4* indirect;        // per pixel: indirect sum so far
  float4* blend;           // per pixel: running blend of the 25-sample variant (between chunks of its samples)
  uint32_t* counts;        // [maxDepth + 1] queue lengths, kQueueStride dwords apart (one cache line per counter)
  uint32_t* work;          // [maxDepth + 1] chunk counters of the bounce launches, likewise
  uint32_t sample;         // sampleIndex passed to shade (frameCount, or frameCount*32 + k) of the launch's first frame
  uint32_t raw;            // != 0: the resolve stage stores direct + indirect as is (25-sample variant: lt_gi_blend25_kernel
                           // blends, clamps and accumulates afterwards); 0: it clamps (the frame's own colour is final)
  // Several frames per set of launches (FrameParams::fusedFrames of them; for the 25-sample variant the "frames" are the
  // samples k of one frame): frame f uses sample + f, its pixels live at [f * pixels, (f + 1) * pixels) of direct / indirect,
  // and every path carries its frame (m.w).
  uint32_t pixels;         // compact output pixels of one frame
  uint32_t ldsRows;        // rows of each wave's LDS stack (the stage kernels with several waves per workgroup need it)
  // The extension rays of a stage traced ahead of it by lt_trace_kernel (below): hits[e] = (primitive, hitType, u, v) of path e
  // of the stage's queue, or null when the stage traces them itself (small LDS-resident scenes; scenes without an own tree).
  const uint4* hits;
  // ... and then the stage runs as three kernels with a compacted list between them instead of one (below): the paths whose
  // extension ray hit a surface (hitList, hitCount), their light samples and shadow rays (so = position + tmax, sd = direction,
  // sm = (path, primitive, n.l), sn = normal), which lt_trace_kernel walks as any-hit rays (their fate: `occluded`, by list entry).
  // Queue 0 is then not appended to but DIRECT-MAPPED: the path of lane l of square p (position in the XCD-contiguous square
  // list) and frame f sits at slot (p * frames + f) * 64 + l, dead slots (no surface hit, pixel outside the image) marked by
  // m.x = kDeadPath.  No atomic in the camera stage (its 2 M appends to one counter were a 12 ns queue of their own), and the
  // queue is in the squares' order: an eighth of it is an eighth of the image, which lt_trace_kernel hands to one XCD.
  uint32_t directQueue;
  uint32_t* hitList;
  uint32_t* hitCount;      // [maxDepth + 1], kQueueStride dwords apart
  float4* so; float4* sd; uint4* sm; float4* sn;
  const uint32_t* occluded;   // [i of the list] != 0: the shadow ray met something
};

__device__ __forceinline__ bool square_pixel(const FrameParams& fp, uint32_t b, uint32_t& x, uint32_t& y, uint32_t& pix) {
  const uint32_t k = b / fp.blocksPerTile, sb = b % fp.blocksPerTile;
  const uint32_t sbx = sb % fp.blocksPerTileX, sby = sb / fp.blocksPerTileX;
  const uint32_t tile = fp.tileFirst + k * fp.tileStride;
  const uint32_t tx = tile % fp.tilesX, ty = tile / fp.tilesX;
  const uint32_t lane = threadIdx.x;
  const uint32_t lx = sbx * 8u + (lane & 7u), ly = sby * 8u + (lane >> 3);
  x = tx * fp.tileW + lx;
  y = ty * fp.tileH + ly;
  pix = (k * fp.tileH + ly) * fp.tileW + lx;
  return k < fp.tilesInCall && lx < fp.tileW && ly < fp.tileH && x < fp.width && y < fp.height;
}

// one atomic per wave: slot of this lane among the lanes with `keep`
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool keep) {
  const unsigned long long m = __ballot(keep);
  if (m == 0ull) return 0u;
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if ((int)__lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  return base + (uint32_t)__popcll(m & ((1ull << __lane_id()) - 1ull));
}

template <class CFG>
__global__ __launch_bounds__(kBlock, LT_GI_STAGE_WAVES) void lt_gi_primary_kernel(SceneDev sc, FrameParams fp, GiParams gp, uint32_t* __restrict__ queues) {
  extern __shared__ int lds_stack[];
  Stack<CFG::kDeep> st;
  st.lds = lds_stack + threadIdx.x;
  st.rows = (int)fp.ldsRows;
  Counters c{};
  const uint32_t n = fp.totalSquares, q = n / 8u, r = n % 8u;
  const uint32_t home = __builtin_amdgcn_s_getreg((3u << 11) | 20u) & 7u;
  if (gp.directQueue && blockIdx.x == 0 && threadIdx.x == 0) gp.counts[0] = n * fp.fusedFrames * (uint32_t)kBlock;   // (read by the launches behind this one)
  for (uint32_t sweep = 0; sweep < 8u;) {
    const uint32_t xcd = (home + sweep) & 7u;
    const uint32_t share = q + (xcd < r ? 1u : 0u), start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    uint32_t t = 0;
    if (threadIdx.x == 0) t = atomicAdd(&queues[xcd * kQueueStride], 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= share * fp.fusedFrames) { sweep++; continue; }
    const uint32_t frame = t / share;   // frame-major inside the XCD's share
    t -= frame * share;
    uint32_t x, y, pix;
    const bool valid = square_pixel(fp, start + t, x, y, pix);
    pix += frame * gp.pixels;
    bool alive = false;
    V4 position{}, normal{}, dir{};
    float fx = 0.0f, fy = 0.0f;
    int prim = 0;
    if (valid) {
      const Ray ray = camera_ray<CFG::kDevLibm>(fp, (int)x, (int)y, fx, fy);
      const uint32_t s = gp.sample + frame;
      V3 direct{0.0f, 0.0f, 0.0f};
      Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
      traverse_camera<kGI, CFG::kDeep, false>(sc, ray, pl, st, c);
      if (is_light(sc.lights, pl.prim)) {
        direct = V3{1.0f, 1.0f, 1.0f};
      } else if (pl.hitType == 1) {
        const float* pr = prim_ptr(sc, pl.prim);
        const Material* m = sc.mats + prim_material(pr);
        float ndotl;
        if (direct_light<kGIPrimary, CFG>(sc, pr, pl.prim, pl.u, pl.v, fx, fy, (float)s, (float)(s + 1u), (float)(s + 2u), 1.0f, position,
                                          normal, ndotl, st, c)) {
          direct = V3{m->diffuse[0] * ndotl, m->diffuse[1] * ndotl, m->diffuse[2] * ndotl};
        }
        const V4 hemi = uniform_sample_hemisphere<CFG::kDevLibm>(random_<CFG::kDevLibm>(fx, fy, (float)(s + 3u)), random_<CFG::kDevLibm>(fx, fy, (float)(s + 4u)));
        dir = align_hemisphere<CFG::kDevLibm>(hemi, normal);
        prim = pl.prim;
        alive = fp.giMaxDepth > 0;
      }
      gp.direct[pix] = make_float4(direct.x, direct.y, direct.z, 0.0f);
      gp.indirect[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    uint32_t slot;
    if (gp.directQueue) {
      slot = ((start + t) * fp.fusedFrames + frame) * (uint32_t)kBlock + threadIdx.x;
      if (!alive) gp.q[0].m[slot] = make_uint4(kDeadPath, 0u, 0u, 0u);
    } else {
      slot = wave_append(&gp.counts[0], alive);
    }
    if (alive) {
      gp.q[0].o[slot] = make_float4(position.x, position.y, position.z, fx);
      gp.q[0].d[slot] = make_float4(dir.x, dir.y, dir.z, dir.w);
      gp.q[0].n[slot] = make_float4(normal.x, normal.y, normal.z, normal.w);
      gp.q[0].m[slot] = make_uint4(pix, (uint32_t)prim, __float_as_uint(fy), frame);
    }
  }
}

// Workgroup of the kLdsScene instantiations: eight wavefronts share one LDS copy of the scene (a 4.7 KB Cornell box + eight
// 2 KB stacks = 21 KB per workgroup: the CU still holds its 32 waves).
constexpr int kLdsSceneWaves = 8;

// The workgroup's LDS: [scene copy (kLdsScene only): nodes, then traversal triangles][one stack of gp.ldsRows rows per wave].
// Copies the scene (all threads, then a barrier) and returns this thread's stack column.
template <class CFG>
__device__ __forceinline__ int* stage_lds_setup(SceneDev& sc, uint32_t ldsRows, int* lds) {
  uint32_t stackBase = 0;   // dwords
  if constexpr (CFG::kLdsScene) {
    const uint32_t nodeF4 = sc.n_nodes * 2u, triF4 = sc.n_prims * 3u;
    float4* dst = (float4*)lds;
    for (uint32_t i = threadIdx.x; i < nodeF4; i += blockDim.x) dst[i] = sc.nodes[i];
    for (uint32_t i = threadIdx.x; i < triF4; i += blockDim.x) dst[nodeF4 + i] = sc.tris[i];
    __syncthreads();
    sc.ldsNodes = (uint32_t)(size_t)(__attribute__((address_space(3))) int*)lds;
    sc.ldsTris = sc.ldsNodes + nodeF4 * 16u;
    stackBase = (nodeF4 + triF4) * 4u;
  }
  return lds + stackBase + (threadIdx.x / kBlock) * ldsRows * kBlock + (threadIdx.x % kBlock);
}

template <class CFG>
__global__ __launch_bounds__(CFG::kLdsScene ? kBlock * kLdsSceneWaves : kBlock, LT_GI_STAGE_WAVES)
void lt_gi_bounce_kernel(SceneDev sc, FrameParams fp, GiParams gp, uint32_t depth) {
  extern __shared__ int lds_stack[];
  Stack<CFG::kDeep> st;
  st.lds = stage_lds_setup<CFG>(sc, gp.ldsRows, lds_stack);
  st.rows = (int)gp.ldsRows;
  Counters c{};
  const GiQueue in = gp.q[depth & 1u], out = gp.q[(depth + 1u) & 1u];
  const uint32_t total = gp.counts[depth * kQueueStride];
  const int d = (int)depth;
  const uint32_t lane = threadIdx.x % kBlock;
  for (;;) {
    uint32_t chunk = 0;
    if (lane == 0) chunk = atomicAdd(&gp.work[depth * kQueueStride], 1u);
    chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
    if ((uint64_t)chunk * kBlock >= total) break;
    const uint32_t e = chunk * kBlock + lane;
    bool alive = false;
    V4 epos{}, enorm{}, ndir{};
    uint4 misc = make_uint4(0u, 0u, 0u, 0u);
    float fx = 0.0f;
    int hitPrim = 0;
    if (e < total) {
      const float4 o = in.o[e], dd = in.d[e], nn = in.n[e];
      misc = in.m[e];
      fx = o.w;
      const float fy = __uint_as_float(misc.z);
      const uint32_t pix = misc.x;
      const Ray ext{mk4(o.x, o.y, o.z, 1.0f), mk4(dd.x, dd.y, dd.z, dd.w)};
      const V4 previousNormal = mk4(nn.x, nn.y, nn.z, nn.w);
      Hit epl{0, 0, kFltMax, 0.0f, 0.0f};
      if (gp.hits != nullptr) {   // traced ahead of this launch by lt_trace_kernel (t is not read below)
        const uint4 h = gp.hits[e];
        epl.prim = (int)h.x; epl.hitType = (int)h.y; epl.u = __uint_as_float(h.z); epl.v = __uint_as_float(h.w);
      } else {
        traverse<kGI, CFG::kDeep, false, false, CFG::kLdsScene>(sc, ext, true, (int)misc.y, epl, st, c);
      }
      const uint32_t s = gp.sample + misc.w, sd = s + depth;
      float4 ind = gp.indirect[pix];
      if (is_light(sc.lights, epl.prim)) {
        // the reference keeps looping with the SAME ray (gi.cl:319-321): same hit, one more term per remaining depth
        const float k = dot4(previousNormal, ext.d);
        for (int dd2 = d; dd2 < fp.giMaxDepth; dd2++) {
          const float w = (float)(1.0 / (double)(dd2 + 1));
          ind.x = Math<CFG::kDevLibm>::mad(w * 1.0f, k, ind.x);
          ind.y = Math<CFG::kDevLibm>::mad(w * 1.0f, k, ind.y);
          ind.z = Math<CFG::kDevLibm>::mad(w * 1.0f, k, ind.z);
        }
        gp.indirect[pix] = ind;
      } else if (epl.hitType == 1) {
        const float w = (float)(1.0 / (double)(d + 1));
        const float* epr = prim_ptr(sc, epl.prim);
        const Material* em = sc.mats + prim_material(epr);
        float endotl;
        if (direct_light<kGI, CFG>(sc, epr, epl.prim, epl.u, epl.v, fx, fy, (float)(sd + 5u), (float)(sd + 6u), (float)(sd + 7u), 1.0f,
                                   epos, enorm, endotl, st, c)) {
          ind.x = Math<CFG::kDevLibm>::mad(w * em->diffuse[0], endotl, ind.x);
          ind.y = Math<CFG::kDevLibm>::mad(w * em->diffuse[1], endotl, ind.y);
          ind.z = Math<CFG::kDevLibm>::mad(w * em->diffuse[2], endotl, ind.z);
          gp.indirect[pix] = ind;
          const V4 hemi = uniform_sample_hemisphere<CFG::kDevLibm>(random_<CFG::kDevLibm>(fx, fy, (float)(sd + 8u)), random_<CFG::kDevLibm>(fx, fy, (float)(sd + 9u)));
          ndir = align_hemisphere<CFG::kDevLibm>(hemi, enorm);
          hitPrim = epl.prim;
          alive = d + 1 < fp.giMaxDepth;
        }
      }
    }
    const uint32_t slot = wave_append(&gp.counts[(depth + 1u) * kQueueStride], alive);
    if (alive) {
      out.o[slot] = make_float4(epos.x, epos.y, epos.z, fx);
      out.d[slot] = make_float4(ndir.x, ndir.y, ndir.z, ndir.w);
      out.n[slot] = make_float4(enorm.x, enorm.y, enorm.z, enorm.w);
      out.m[slot] = make_uint4(misc.x, (uint32_t)hitPrim, misc.z, misc.w);
    }
  }
}

// ---- a bounce stage in pieces, around lt_trace_kernel (further down), when its extension rays were traced ahead of it:
//   lt_gi_classify_kernel : every path of the stage's queue: a light hit adds its terms (gi.cl:319-321); a surface hit joins the list
//   lt_gi_shadow_kernel   : every path of the list: the light sample (gi.cl:323-349) and its shadow ray, into the shadow queue
//   (lt_trace_kernel, any-hit, over the shadow queue)
//   lt_gi_finish_kernel   : every path of the list whose shadow ray found nothing: the indirect term, the next extension ray
//                           (gi.cl:350-366), appended to the next stage's queue
// Same arithmetic per path as lt_gi_bounce_kernel, same order of a pixel's terms (one path per pixel and frame).  What it buys:
// on the 1 M-triangle wall nine extension rays in ten hit nothing, and a wavefront of the one-kernel stage ran the light
// sampling (three double-precision random() calls) and a per-lane shadow walk for the few lanes that did; here those lanes of
// all wavefronts are packed into full wavefronts first.
// (kClassifyBatch paths per lane and list append: an atomic on ONE address completes every ~12 ns chip-wide, and one per 64 paths
// made this kernel -- 16 bytes in, at most 4 out per path -- take 33 ms for the 133 M paths of the 1 M-triangle wall's first stage)
constexpr int kClassifyBatch = 8;
template <class CFG>
__global__ __launch_bounds__(256) void lt_gi_classify_kernel(SceneDev sc, FrameParams fp, GiParams gp, uint32_t depth) {
  const GiQueue in = gp.q[depth & 1u];
  const uint32_t total = gp.counts[depth * kQueueStride];
  const uint32_t lane = threadIdx.x % kBlock, wave = (blockIdx.x * blockDim.x + threadIdx.x) / kBlock, waves = gridDim.x * blockDim.x / kBlock;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (uint64_t base = (uint64_t)wave * (kBlock * kClassifyBatch); base < total; base += (uint64_t)waves * (kBlock * kClassifyBatch)) {
    bool surface[kClassifyBatch];
    uint32_t before[kClassifyBatch];   // surface hits of this wave's batch in front of this lane's k-th path
    uint32_t count = 0u;
#pragma unroll
    for (int k = 0; k < kClassifyBatch; k++) {
      const uint64_t e = base + (uint64_t)k * kBlock + lane;
      surface[k] = false;
      if (e < total && !(gp.directQueue && depth == 0u && in.m[e].x == kDeadPath)) {
        const uint4 h = gp.hits[e];
        if (is_light(sc.lights, (int)h.x)) {
          // the reference keeps looping with the SAME ray (gi.cl:319-321): same hit, one more term per remaining depth
          const float4 dd = in.d[e], nn = in.n[e];
          const uint32_t pix = in.m[e].x;
          const float kk = dot4(mk4(nn.x, nn.y, nn.z, nn.w), mk4(dd.x, dd.y, dd.z, dd.w));
          float4 ind = gp.indirect[pix];
          for (int dd2 = (int)depth; dd2 < fp.giMaxDepth; dd2++) {
            const float w = (float)(1.0 / (double)(dd2 + 1));
            ind.x = Math<CFG::kDevLibm>::mad(w * 1.0f, kk, ind.x);
            ind.y = Math<CFG::kDevLibm>::mad(w * 1.0f, kk, ind.y);
            ind.z = Math<CFG::kDevLibm>::mad(w * 1.0f, kk, ind.z);
          }
          gp.indirect[pix] = ind;
        } else {
          surface[k] = h.y == 1u;
        }
      }
      const unsigned long long m = __ballot(surface[k]);
      before[k] = count + (uint32_t)__popcll(m & below);
      count += (uint32_t)__popcll(m);
    }
    if (count == 0u) continue;
    uint32_t slot0 = 0u;
    if (lane == 0u) slot0 = atomicAdd(&gp.hitCount[depth * kQueueStride], count);
    slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
#pragma unroll
    for (int k = 0; k < kClassifyBatch; k++)
      if (surface[k]) gp.hitList[slot0 + before[k]] = (uint32_t)(base + (uint64_t)k * kBlock + lane);
  }
}

template <class CFG>
__global__ __launch_bounds__(256) void lt_gi_shadow_kernel(SceneDev sc, FrameParams fp, GiParams gp, uint32_t depth) {
  const GiQueue in = gp.q[depth & 1u];
  const uint32_t total = gp.hitCount[depth * kQueueStride];
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint32_t e = gp.hitList[i];
    const uint4 h = gp.hits[e], misc = in.m[e];
    const float fx = in.o[e].w, fy = __uint_as_float(misc.z);
    const uint32_t sd = gp.sample + misc.w + depth;
    V4 epos, enorm, toLight;
    float tmax, endotl;
    light_sample<CFG>(sc, prim_ptr(sc, (int)h.x), __uint_as_float(h.z), __uint_as_float(h.w), fx, fy, (float)(sd + 5u), (float)(sd + 6u), (float)(sd + 7u),
                      1.0f, epos, enorm, toLight, tmax, endotl);
    gp.so[i] = make_float4(epos.x, epos.y, epos.z, tmax);
    gp.sd[i] = make_float4(toLight.x, toLight.y, toLight.z, toLight.w);
    gp.sm[i] = make_uint4(e, h.x, __float_as_uint(endotl), 0u);
    gp.sn[i] = make_float4(enorm.x, enorm.y, enorm.z, enorm.w);
  }
}

template <class CFG>
__global__ __launch_bounds__(256) void lt_gi_finish_kernel(SceneDev sc, FrameParams fp, GiParams gp, uint32_t depth) {
  const GiQueue in = gp.q[depth & 1u], out = gp.q[(depth + 1u) & 1u];
  const uint32_t total = gp.hitCount[depth * kQueueStride];
  const uint32_t stride = gridDim.x * blockDim.x;
  const int d = (int)depth;
  for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += stride) {   // (whole wavefronts take part in wave_append)
    const uint32_t i = base + threadIdx.x;
    bool alive = false;
    V4 ndir{};
    float4 so = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sn = so;
    uint4 sm = make_uint4(0u, 0u, 0u, 0u), misc = sm;
    float fx = 0.0f;
    if (i < total && gp.occluded[i] == 0u) {   // the shadow ray found nothing (gi.cl:351)
      so = gp.so[i]; sn = gp.sn[i]; sm = gp.sm[i];
      const uint32_t e = sm.x;
      misc = in.m[e];
      fx = in.o[e].w;
      const float fy = __uint_as_float(misc.z);
      const uint32_t pix = misc.x, sd = gp.sample + misc.w + depth;
      const float w = (float)(1.0 / (double)(d + 1));
      const Material* em = sc.mats + prim_material(prim_ptr(sc, (int)sm.y));
      const float endotl = __uint_as_float(sm.z);
      float4 ind = gp.indirect[pix];
      ind.x = Math<CFG::kDevLibm>::mad(w * em->diffuse[0], endotl, ind.x);
      ind.y = Math<CFG::kDevLibm>::mad(w * em->diffuse[1], endotl, ind.y);
      ind.z = Math<CFG::kDevLibm>::mad(w * em->diffuse[2], endotl, ind.z);
      gp.indirect[pix] = ind;
      const V4 hemi = uniform_sample_hemisphere<CFG::kDevLibm>(random_<CFG::kDevLibm>(fx, fy, (float)(sd + 8u)), random_<CFG::kDevLibm>(fx, fy, (float)(sd + 9u)));
      ndir = align_hemisphere<CFG::kDevLibm>(hemi, mk4(sn.x, sn.y, sn.z, sn.w));
      alive = d + 1 < fp.giMaxDepth;
    }
    const uint32_t slot = wave_append(&gp.counts[(depth + 1u) * kQueueStride], alive);
    if (alive) {
      out.o[slot] = make_float4(so.x, so.y, so.z, fx);
      out.d[slot] = make_float4(ndir.x, ndir.y, ndir.z, ndir.w);
      out.n[slot] = sn;
      out.m[slot] = make_uint4(misc.x, sm.y, misc.z, misc.w);
    }
  }
}

// =====================================================================================================================
// lt_trace_kernel: a queue of rays through the per-lane walk over the own tree (lt_device.hpp: own_walk_step), with LANE REFILL.
// Incoherent rays differ in cost by two orders of magnitude -- a bounce ray that leaves the 1 M-triangle wall visits a handful
// of groups, one that grazes it several hundred leaves -- and a wavefront that walks 64 of them side by side runs until its
// slowest lane is done (measured on that scene's bounce stage: 14-17 % of the lanes active in the average vector instruction,
// profiles/r2/gi_wall, gpurun_out r3).  Here a lane whose ray is done takes the next ray of the queue: whenever at least
// `refill` lanes of the wave are idle (or all of them), the idle lanes claim that many rays with ONE atomic (ballot, popcount,
// prefix count) and set them up; then every lane with a ray makes one step.  A lane's state is its ray and its stack, nothing
// of the shading that produced the ray or will consume the hit: that is why this is a kernel of its own.
//   rays:   o[i] = (origin.xyz, tmax of an any-hit ray), d[i] = direction.xyzw, m[i].y = the primitive the ray starts on (ignored,
//           acc.cl:188)
//   result: hit[i] = (primitive, hitType, u, v) of a closest-hit ray (t is not kept: no caller reads it); occluded[i] = hitType of an
//           any-hit ray (its caller reads nothing else)
// Rays the walks over the own tree do not take (a non-finite component, magnitudes beyond packet_ray_ok) are walked at once, in
// the reference's order over the caller's tree, by the lane that drew them.  Every wave reaches the exit: the queue hands out
// each index once, a lane's walk ends (the stack only holds entries of a finite tree), and the loop ends when the queue is drained
// and no lane holds a ray.
constexpr int kTraceClaim = 512;   // rays a wave claims from the queue per atomic
struct TraceParams {
  const float4* o;
  const float4* d;
  const uint4* m;
  uint4* hit;              // closest-hit walks
  uint32_t* occluded;      // any-hit walks: one word per ray, != 0 when something was hit (their callers read nothing else)
  const uint32_t* count;   // rays in the queue (device memory: written by the stage that filled it)
  uint32_t* next;          // work counters (zeroed by the host): one per eighth of the queue, kQueueStride dwords apart
  uint32_t refill;         // idle lanes that trigger a refill
  uint32_t dead;           // != 0: the queue is direct-mapped, slots with m.x == kDeadPath hold no ray
};

template <int PROGRAM, bool ANYHIT>
__global__ __launch_bounds__(kBlock, 8) void lt_trace_kernel(SceneDev sc, TraceParams tp) {
  using u64 = unsigned long long;
  extern __shared__ int lds_stack[];   // [kTraceRows stack rows][kTraceStage rows of staged rays], 64 lanes each
  int* const col = lds_stack + threadIdx.x;
  int* const stage = lds_stack + kTraceRows * kBlock;
  const uint32_t total = tp.count[0];
  const uint32_t lane = threadIdx.x;
  const u64 below = (1ull << lane) - 1ull;
  // (a short queue is shared out in small claims, so that it is not a few waves' work; a long one in claims of kTraceClaim)
  const uint32_t share = total / (gridDim.x * 4u) / (uint32_t)kBlock * (uint32_t)kBlock;
  const uint32_t claim = share < (uint32_t)kBlock ? (uint32_t)kBlock : (share > (uint32_t)kTraceClaim ? (uint32_t)kTraceClaim : share);
  bool active = false;
  // the wave's staged batch of rays: stageCount of them (each with its index in the queue), the first stageTaken handed out; they
  // come out of the wave's claim [claimNext, claimEnd) of the queue (kTraceClaim rays per atomic: one address serves an atomic
  // every ~12 ns, 80 M per second chip-wide)
  uint32_t stageCount = 0u, stageTaken = 0u, claimNext = 0u, claimEnd = 0u, sweep = 0u;
  const uint32_t home = __builtin_amdgcn_s_getreg((3u << 11) | 20u) & 7u;   // HW_REG_XCC_ID
  bool drained = false;   // the queue has handed out its last ray, and the wave's claim is staged
  Ray ray{};
  float ix = 0.0f, iy = 0.0f, iz = 0.0f;
  OwnRay w{};
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  uint32_t idx = 0u, e = 0u;
  int sp = 0;
  int deep[kOwnRows + kOwnDeep - kTraceRows];
  for (;;) {
    const u64 idle = __builtin_amdgcn_ballot_w64(!active);
    const uint32_t nIdle = (uint32_t)__popcll(idle);
    if ((nIdle >= tp.refill || nIdle == (uint32_t)kBlock) && (stageTaken < stageCount || !drained)) {
      if (stageTaken == stageCount) {
        // the next 64 rays of the queue, one per lane (coalesced), parked in LDS: the lanes' own registers hold their walks
        // The queue in eighths, one per XCD (an eighth of a direct-mapped queue is an eighth of the image: the XCD's L2 then
        // serves the groups and leaves of one part of the scene): a wave claims from the eighth of the XCD it runs on, then from
        // the others'.
        while (claimNext == claimEnd && sweep < 8u) {
          const uint32_t part = (home + sweep) & 7u;
          const uint32_t lo = (uint32_t)((uint64_t)total * part / 8u / kBlock * kBlock), hi = part == 7u ? total : (uint32_t)((uint64_t)total * (part + 1u) / 8u / kBlock * kBlock);
          uint32_t got = 0u;
          if (lane == 0u) got = atomicAdd(&tp.next[part * kQueueStride], claim);
          got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
          if (got >= hi - lo) { sweep++; continue; }
          claimNext = lo + got;
          claimEnd = hi - claimNext < claim ? hi : claimNext + claim;
        }
        const uint32_t base = claimNext;
        const uint32_t batch = claimEnd - base < (uint32_t)kBlock ? claimEnd - base : (uint32_t)kBlock;
        claimNext = base + batch;
        drained = claimNext == claimEnd && sweep >= 8u;
        stageTaken = 0u;
        float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f), dd = o;
        uint32_t ign = 0u;
        bool live = lane < batch;
        if (live) {
          const uint2 mm = *(const uint2*)&tp.m[base + lane];
          ign = mm.y;
          live = !(tp.dead && mm.x == kDeadPath);
        }
        if (live) { o = tp.o[base + lane]; dd = tp.d[base + lane]; }
        const u64 lm = __builtin_amdgcn_ballot_w64(live);
        stageCount = (uint32_t)__popcll(lm);
        if (live) {
          const uint32_t at = (uint32_t)__popcll(lm & below);
          stage[0 * kBlock + at] = __float_as_int(o.x); stage[1 * kBlock + at] = __float_as_int(o.y); stage[2 * kBlock + at] = __float_as_int(o.z);
          stage[3 * kBlock + at] = __float_as_int(dd.x); stage[4 * kBlock + at] = __float_as_int(dd.y); stage[5 * kBlock + at] = __float_as_int(dd.z);
          stage[6 * kBlock + at] = __float_as_int(dd.w); stage[7 * kBlock + at] = (int)ign;
          stage[8 * kBlock + at] = __float_as_int(o.w);   // tmax of an any-hit ray
          stage[9 * kBlock + at] = (int)(base + lane);
        }
        // (one wavefront per workgroup: its own LDS writes are visible to it once they have completed -- the reads below wait for them)
      }
      const uint32_t take = nIdle < stageCount - stageTaken ? nIdle : stageCount - stageTaken;
      const uint32_t mine = (uint32_t)__popcll(idle & below);
      if (!active && mine < take) {
        const uint32_t s = stageTaken + mine;
        idx = (uint32_t)stage[9 * kBlock + s];
        ray = Ray{mk4(__int_as_float(stage[0 * kBlock + s]), __int_as_float(stage[1 * kBlock + s]), __int_as_float(stage[2 * kBlock + s]), 1.0f),
                  mk4(__int_as_float(stage[3 * kBlock + s]), __int_as_float(stage[4 * kBlock + s]), __int_as_float(stage[5 * kBlock + s]),
                      __int_as_float(stage[6 * kBlock + s]))};
        const int ign = stage[7 * kBlock + s];
        ix = 1.0f / ray.d.x; iy = 1.0f / ray.d.y; iz = 1.0f / ray.d.z;
        pl = Hit{0, 0, ANYHIT ? __int_as_float(stage[8 * kBlock + s]) : kFltMax, 0.0f, 0.0f};
        const bool finite = __builtin_fabsf(ix) < __builtin_inff() && __builtin_fabsf(iy) < __builtin_inff() && __builtin_fabsf(iz) < __builtin_inff() &&
                            __builtin_fabsf(ray.o.x) < __builtin_inff() && __builtin_fabsf(ray.o.y) < __builtin_inff() &&
                            __builtin_fabsf(ray.o.z) < __builtin_inff();
        if (finite && packet_ray_ok(ray, ix, iy, iz)) {
          w = own_ray(sc, ray, ix, iy, iz, ign);
          e = 0u;
          sp = 0;
          active = true;
        } else {   // (rare) the reference's order over the caller's tree, here and now
          ScratchStack ss;
          Counters c{};
          traverse_nodes_impl<PROGRAM, ScratchStack, false, false, ANYHIT, false>(sc, ray, ix, iy, iz, true, ign, pl, ss, c);
          if (ANYHIT) tp.occluded[idx] = (uint32_t)pl.hitType;
          else tp.hit[idx] = make_uint4((uint32_t)pl.prim, (uint32_t)pl.hitType, __float_as_uint(pl.u), __float_as_uint(pl.v));
        }
      }
      stageTaken += take;
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
      if (drained && stageTaken == stageCount) break;
      continue;
    }
    if (active) {
      if (own_walk_step<PROGRAM, ANYHIT, kTraceRows>(sc, ray, ix, iy, iz, w, pl, col, deep, e, sp)) {
        if (ANYHIT) tp.occluded[idx] = (uint32_t)pl.hitType;
        else tp.hit[idx] = make_uint4((uint32_t)pl.prim, (uint32_t)pl.hitType, __float_as_uint(pl.u), __float_as_uint(pl.v));
        active = false;
      }
    }
  }
}

// accumulator's shadow rays walked by lt_trace_kernel (SceneDev::shadowPackets == 3): a sample whose ray met an occluder is
// black (acc.cl:276-279: the colour is only assigned when the shadow payload's hitType is 0)
__global__ __launch_bounds__(256) void lt_shadow_resolve_kernel(const uint4* __restrict__ m, const uint32_t* __restrict__ occluded, uint32_t slots,
                                                                float* __restrict__ out, unsigned long long frameStride, uint32_t depth) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
    const uint4 mm = m[i];
    if (mm.x == kDeadSlot || occluded[i] == 0u) continue;
    float* o = out + (size_t)mm.z * frameStride + (size_t)mm.x * depth;
    o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
  }
}

// direct + indirect (gi.cl:374), clamp (:409-411), running mean, store
template <class CFG>
__global__ void lt_gi_resolve_kernel(FrameParams fp, GiParams gp, float* __restrict__ out, uint32_t pixels) {
  const uint32_t vpix = blockIdx.x * blockDim.x + threadIdx.x;   // (frame, pixel) of a fused launch
  if (vpix >= pixels) return;
  const uint32_t frame = vpix / gp.pixels, pix = vpix - frame * gp.pixels;
  // only pixels inside the image were written by the primary stage
  const uint32_t perTile = fp.tileW * fp.tileH, k = pix / perTile, rem = pix % perTile, ly = rem / fp.tileW, lx = rem % fp.tileW;
  const uint32_t tile = fp.tileFirst + k * fp.tileStride, tx = tile % fp.tilesX, ty = tile / fp.tilesX;
  if (k >= fp.tilesInCall || tx * fp.tileW + lx >= fp.width || ty * fp.tileH + ly >= fp.height) return;
  const float4 di = gp.direct[vpix], in = gp.indirect[vpix];
  V3 color{di.x + in.x, di.y + in.y, di.z + in.z};
  if (fp.clampOutput && !gp.raw) color = V3{Math<CFG::kDevLibm>::clamp01(color.x), Math<CFG::kDevLibm>::clamp01(color.y), Math<CFG::kDevLibm>::clamp01(color.z)};
  float* o = out + (size_t)frame * fp.frameStride + (size_t)pix * fp.depth;   // fused: this frame's un-accumulated slice
  if (fp.accumulateN <= 0) {
    o[0] = color.x; o[1] = color.y; o[2] = color.z;
  } else {
    const float n = (float)fp.accumulateN, n1 = (float)(fp.accumulateN + 1);
    o[0] = (color.x + (o[0] * n)) / n1;
    o[1] = (color.y + (o[1] * n)) / n1;
    o[2] = (color.z + (o[2] * n)) / n1;
  }
}

// The 25-sample variant (resources/kernels/opencl/global_illumination.cl:408-420): samples k0 .. k0+n-1 of one frame, stored
// un-clamped by the resolve stage at samples + j * stride, are blended in order, `c = (1-a)*c + a*c_k`, a = (25-k)/25; the
// partial blend waits in gp.blend between chunks; after sample 24 the colour is clamped (linearKernel) and stored or folded
// into the running mean like any frame.  Pixels of edge tiles outside the image are skipped.
template <class CFG>
__global__ void lt_gi_blend25_kernel(FrameParams fp, GiParams gp, const float* __restrict__ samples, uint32_t k0, uint32_t n,
                                     float* __restrict__ out, uint32_t pixels) {
  const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= pixels) return;
  const uint32_t perTile = fp.tileW * fp.tileH, k = pix / perTile, rem = pix % perTile, ly = rem / fp.tileW, lx = rem % fp.tileW;
  const uint32_t tile = fp.tileFirst + k * fp.tileStride, tx = tile % fp.tilesX, ty = tile / fp.tilesX;
  if (k >= fp.tilesInCall || tx * fp.tileW + lx >= fp.width || ty * fp.tileH + ly >= fp.height) return;
  V3 color{0.0f, 0.0f, 0.0f};
  if (k0 > 0u) {
    const float4 b = gp.blend[pix];
    color = V3{b.x, b.y, b.z};
  }
  for (uint32_t j = 0; j < n; j++) {
    const float* c = samples + (size_t)j * fp.frameStride + (size_t)pix * fp.depth;
    const V3 cn{c[0], c[1], c[2]};
    const uint32_t kk = k0 + j;
    if (kk == 0u) {
      color = cn;
    } else {
      const float a = Math<CFG::kDevLibm>::div25((float)(25 - (int)kk));
      color = V3{Math<CFG::kDevLibm>::mad(1.0f - a, color.x, a * cn.x), Math<CFG::kDevLibm>::mad(1.0f - a, color.y, a * cn.y),
                 Math<CFG::kDevLibm>::mad(1.0f - a, color.z, a * cn.z)};
    }
  }
  if (k0 + n < 25u) {
    gp.blend[pix] = make_float4(color.x, color.y, color.z, 0.0f);
    return;
  }
  if (fp.clampOutput) color = V3{Math<CFG::kDevLibm>::clamp01(color.x), Math<CFG::kDevLibm>::clamp01(color.y), Math<CFG::kDevLibm>::clamp01(color.z)};
  float* o = out + (size_t)pix * fp.depth;
  if (fp.accumulateN <= 0) {
    o[0] = color.x; o[1] = color.y; o[2] = color.z;
  } else {
    const float nf = (float)fp.accumulateN, n1 = (float)(fp.accumulateN + 1);
    o[0] = (color.x + (o[0] * nf)) / n1;
    o[1] = (color.y + (o[1] * nf)) / n1;
    o[2] = (color.z + (o[2] * nf)) / n1;
  }
}
