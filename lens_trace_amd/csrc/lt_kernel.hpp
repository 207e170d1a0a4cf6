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
template <int PROGRAM, class CFG>
__device__ __forceinline__ void render_square(const SceneDev& sc, const FrameParams& fp, float* __restrict__ out, uint32_t b,
                                              Stack<CFG::kDeep>& st, Counters& c) {
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
  if (valid) {
    Counters pc{};   // this pixel's own counters (diagnostic output), folded into the lane's totals below
    const V3 color = shade_pixel<PROGRAM, CFG>(sc, fp, (int)x, (int)y, st, STATS ? pc : c);
    float* o = out + (((size_t)k * fp.tileH + ly) * fp.tileW + lx) * fp.depth;
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
}

// Registers: the traversal is latency-bound and wants every wave slot (8 per SIMD = 64 VGPRs); the single-bounce programs fit
// that with a few spilled values in their shading code; the 16-bounce / 25-sample programs would spill 85-140 values at 8
// and run best at 5 waves per SIMD (Cornell GI 1080p, 16 bounces: 3.7 / 3.4 / 3.2 / 3.4 / 4.4 ms at 3 / 4 / 5 / 6 / 8).
#ifndef LT_GI_WAVES
#define LT_GI_WAVES 5
#endif
#ifndef LT_ACC_WAVES
#define LT_ACC_WAVES 8
#endif
constexpr int waves_per_simd(int program) { return (program == kBasic || program == kAccumulator || program == kCustom) ? LT_ACC_WAVES : LT_GI_WAVES; }

template <int PROGRAM, class CFG>
__device__ __forceinline__ void render_kernel_body(const SceneDev& sc, const FrameParams& fp, float* __restrict__ out,
                                                   unsigned long long* __restrict__ stats, uint32_t* __restrict__ queues) {
  extern __shared__ int lds_stack[];   // [BVH height (<= kLdsStack)][kBlock], sized by the launch
  constexpr bool STATS = CFG::kStats;
  Stack<CFG::kDeep> st;
  st.lds = lds_stack + threadIdx.x;
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
    uint32_t b;
    if (!fp.persistent) {
      b = xcd_remap(blockIdx.x, gridDim.x);
      done = true;
    } else {
      const uint32_t xcd = (home + sweep) & 7u;
      const uint32_t share = q + (xcd < r ? 1u : 0u), start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
      uint32_t t = 0;
      if (threadIdx.x == 0) t = atomicAdd(&queues[xcd], 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      if (t >= share) {
        done = ++sweep >= 8u;
        continue;
      }
      b = start + t;
    }
    render_square<PROGRAM, CFG>(sc, fp, out, b, st, c);
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
