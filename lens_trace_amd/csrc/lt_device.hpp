// lt_device.hpp -- device-side arithmetic of the lens_trace hot path for gfx950 (CDNA4).
//
// Every function states the reference lines it computes (paths relative to the reference tree;
// "acc.cl" = examples/accumulator/resources/kernels/accumulator.cl, "gi.cl" =
// examples/global_illumination/resources/kernels/global_illumination.cl, "basic.cl" =
// resources/kernels/opencl/basic.cl).  This translation unit MUST be compiled with
// -ffp-contract=off: user-level expressions of the reference are evaluated operation by operation,
// and the OpenCL builtins are spelled with explicit fmaf() exactly as ROCm's OpenCL device library
// defines them on gfx950 (dot = fma chain, cross = fma(a,b,-(c*d))); see DESIGN.md "Floating-point
// model".  hipcc's defaults give IEEE f32 divide / sqrt and keep f32 denormals.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else   // hipRTC has no system headers; its own fixed-width types live in __hip_internal
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "lt_walk_asm.hpp"

namespace lt {

constexpr int kBlock = 64;         // one wavefront per workgroup: a finished wave frees its LDS and wave slot at once
constexpr int kLdsStack = 32;      // most traversal-stack entries per lane held in LDS by the walks that keep an LDS stack (counting kernels, LDS-resident scenes)
constexpr int kQueueStride = 64;   // dwords between work counters: one 256-byte line each.  (Eight per-XCD counters in one
                                   // cache line serialise every wave of the chip on one memory channel, 12 ns per work
                                   // item: 25 ms of a 2-million-item launch, whatever the items cost.  Taking several
                                   // items per atomic on top of the padding does not pay: -1 % at 4, -2.4 % at 8.)
constexpr int kPacketEntry = 4;    // dwords per entry of a packet walk's wave-uniform stack (node reference + 64-bit lane mask, one spare)
constexpr int kPacketRows = 2;     // ... which therefore fits 32 entries -- kLdsStack levels -- in two 256-byte rows of the wave's LDS
constexpr int kMaxStack = 64;      // the reference's nodesToVisit[64] (acc.cl:137)
constexpr int kOwnRows = 16;       // LDS rows (one dword per lane each) of the stack of a per-lane walk over the own tree's 4-wide groups ...
constexpr int kOwnDeep = 48;       // ... and the entries beyond them, in private memory (the build caps the groups' height at (64 - 4) / 3)
constexpr int kTraceRows = 10;     // lt_trace_kernel: LDS rows of its lanes' stacks (kOwnRows + kOwnDeep - kTraceRows entries in private memory) ...
constexpr int kTraceStage = 10;    // ... and of its staged rays (origin 3, direction 4, ignored primitive, tmax, index): 20 rows = 5 KB, 32 waves per CU
constexpr float kFltMax = 3.402823466e+38f;
constexpr uint32_t kDeadSlot = 0xffffffffu;   // first word of the third array of a direct-mapped ray queue's slot that holds no ray

enum Program { kBasic = 0, kBasicLighting = 1, kAccumulator = 2, kGI = 3, kGI25 = 4, kCustom = 5,
               kAccumulatorQueue = 7,   // accumulator whose shadow rays are not walked but queued for lt_trace_kernel (SceneDev::shadowPackets
                                        // == 3): an instantiation of its own, so that the others carry none of the queue's code or registers
               kGIPrimary = 6,   // the global-illumination programs' camera-ray stage (lt_gi_primary_kernel): kGI's arithmetic; its shadow
                                 // rays start on camera hits, as coherent as accumulator's, and may walk as any-hit packets
               kUser = 1000 };

struct V4 { float x, y, z, w; };
struct V3 { float x, y, z; };
struct Ray { V4 o, d; };
struct Hit { int prim; int hitType; float t, u, v; };   // RayPayload, acc.cl:55-61

struct Material { float diffuse[3]; float ior; float dissolve; float emission[3]; };  // model.h:26-31
struct Lights { uint32_t count; uint32_t primitives[64]; };                          // acc.cl:45-48

// Scene as it sits in HBM.
//   nodes : the reference's LinearBVHNode array, verbatim, read as two 16-byte halves per node.
//   tris  : traversal-side triangles re-tiled at upload, 48-byte stride = 3 x float4:
//           (A.x A.y A.z e1.x) (e1.y e1.z e2.x e2.y) (e2.z 0 0 0), e1 = B-A, e2 = C-A computed
//           with the same float subtractions intersectTriangle performs per call (acc.cl:77-78).
//   prims : the reference's 76-byte Primitive array, verbatim (shading reads positions+normals+material).
struct SceneDev {
  const float4* nodes;
  // The backend's own tree over the caller's leaves (lt_retree.hpp: same leaves, same boxes, binned-SAH hierarchy), walked by
  // every finite ray of the non-counting kernels; null (and rank8 null) when the scene has none: its boxes do not nest.
  const float4* ownPairs;   // 64-byte records of the packet walks (lt_own_pair_kernel)
  // With it, for the closest-hit walks: rank8[8 * primitive + octant] = position of the primitive's leaf in the REFERENCE's
  // depth-first order for rays of that direction-sign octant.  intersectTriangle keeps the first of two hits with equal t
  // (`t < payload.t`, acc.cl:104); a walk that meets the leaves in another order keeps the one with the lower rank.  Null
  // when the walks follow the caller's tree in the reference's order themselves.
  const uint32_t* rank8;
  // The own tree as the per-lane walks read it (traverse_own_lane): 64-byte records, nWide 4-wide groups first -- four child
  // slots of 16 bytes: the child's box quantised to 16 bits per bound on a grid over the scene's bounds, rounded outwards, and a
  // link (the child's own group; or 0x80000000 | record index of a leaf) -- then one leaf record per primitive offset (record
  // nWide + offset: the leaf's own box bit for bit, its re-tiled triangle, the offset), lt_wide_kernel / lt_wide_leaf_kernel.
  // A per-lane walk is a chain of dependent memory round trips, 64 different addresses each, and spends two thirds of its
  // wave-cycles waiting for them (profiles/r2/gi_wall): four boxes per round trip make the chain four times shorter than the
  // binary tree's, and 16 bytes per box cost the vector-memory address path (one lane-address per clock and CU:
  // tools/probes/gather_calib.hip) half of what a 32-byte node does.  Legal for the reason the packet walks' pushed-out boxes
  // are (lt_walk_asm.hpp): only a LEAF's own box needs the reference's exact test, and that comes from the leaf's record.
  // The grid sits in front of the records: wide[-2] = (origin.xyz, -), wide[-1] = (step.xyz, -), fetched by scalar loads at the
  // start of a walk (kernel arguments occupy SGPRs for the whole kernel, and the render kernels have none to spare).
  const uint4* wide;
  uint32_t nWide;
  const float4* tris;
  const float* prims;       // 19 floats per primitive
  const Material* mats;
  const Lights* lights;
  uint32_t n_nodes, n_prims, n_mats;
  uint32_t shadowPackets;   // shadow rays of the non-counting kernels: 0 per lane, 1 as any-hit packets, 2 chosen per wavefront (shadowSpread),
                            // 3 (accumulator) not walked by the render kernel at all but QUEUED for lt_trace_kernel (lt_kernel.hpp),
                            // whose lanes take a new ray when theirs is done: the kernel stores the colour an unoccluded sample
                            // has, lt_shadow_resolve_kernel blacks out the occluded ones;
                            // the host times the four on a scene's first frame (lt_capi.hip)
  // mode 3: the direct-mapped shadow queue, capacity shadowCap slots: origin + tmax, direction, (pixel, ignored primitive, frame, -)
  // at shadowQueue, + shadowCap, + 2 shadowCap; slot = (position of the square in the hand-out order * frames + frame) * 64 + lane;
  // a slot without a ray carries kDeadSlot as its pixel
  float4* shadowQueue;
  uint32_t shadowCap;
  // A scene of a few hundred triangles (the Cornell box: 83 nodes + 42 triangles = 4.7 KB) lives in LDS for the per-lane walks of
  // the kernels instantiated with Config::kLdsScene: byte offsets of the workgroup's copies of `nodes` and `tris` in its LDS
  // (filled by the kernel itself, lt_kernel.hpp).  An incoherent per-lane walk is 64 distinct 32-byte fetches per visited node,
  // and what saturates is the vector-memory address path (texture addresser 82 % busy on the Cornell box's bounce stage): LDS
  // reads take that path out of the walk.
  uint32_t ldsNodes, ldsTris;
  uint32_t fastRcp;         // != 0: intersectTriangle's 1 / det as the reference's as-shipped build computes it (Math<2>::rcp), in every walk
  float shadowSpread;       // shadowPackets == 2: a wave's shadow rays walk as a packet iff every origin lies within sqrt(shadowSpread) x its
                            // own ray's length of the first lane's origin (one surface patch looking at one light), else per lane
};

typedef float F4v __attribute__((ext_vector_type(4)));
typedef float F8v __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) F4v* ConstF4;   // constant address space: uniform loads become s_load
typedef const __attribute__((address_space(4))) F8v* ConstF8;
typedef float F16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) F16v* ConstF16;
__device__ __forceinline__ float4 ld_const(ConstF4 p) { const F4v v = *p; return make_float4(v.x, v.y, v.z, v.w); }
typedef float LdsVec4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) LdsVec4* LdsF4;   // (a plain vector type: HIP's float4 class cannot be read through an address-space pointer)
__device__ __forceinline__ float4 ld_lds(LdsF4 p) { const LdsVec4 v = *p; return make_float4(v.x, v.y, v.z, v.w); }

struct Counters {
  uint32_t rays, shadow, nodes, tris;
#ifdef LT_DEBUG_WAVE_COUNTERS
  uint32_t wInner, wTri, wOuter;   // wave-level executions of the node step / triangle block / outer iteration (leader lane counts)
#endif
};
#ifdef LT_DEBUG_WAVE_COUNTERS
#define LT_WAVE_COUNT(field) do { if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) c.field++; } while (0)
#else
#define LT_WAVE_COUNT(field) do { } while (0)
#endif

// ---------------------------------------------------------------- builtins
__device__ __forceinline__ V4 mk4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
__device__ __forceinline__ V4 add4(V4 a, V4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ V4 sub4(V4 a, V4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ V4 scale4(float s, V4 a) { return mk4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ V4 neg4(V4 a) { return mk4(-a.x, -a.y, -a.z, -a.w); }

__device__ __forceinline__ float dot4(V4 a, V4 b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return __builtin_fmaf(ay, by, ax * bx); }
__device__ __forceinline__ V4 cross4(V4 a, V4 b) {
  return mk4(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
             __builtin_fmaf(a.x, b.y, -(a.y * b.x)), 0.0f);
}
// The four leaf functions where ROCm's OpenCL device library uses a hardware approximation or its own float trig.
//   Math<0> "portable": correctly rounded forms every IEEE machine reproduces -- the CPU oracle's definition.
//   Math<1> "device libm": what the reference's OpenCL kernels get on this GPU (read from their gfx950 ISA):
//       rsqrt  = v_rsq_f32 with the library's denormal pre-scale, sqrt (inside distance) = v_sqrt_f32 with the
//       backend's denormal scaling, sin/cos(float) = ocml's, clamp = v_med3_f32.  With it the HIP path is
//       bit-identical to the reference's own kernels compiled for gfx950 with -ffp-contract=off
//       -cl-fp32-correctly-rounded-divide-sqrt (tests/test_gpu_reference_kernels.py).
//   Math<2> "as shipped": Math<1> plus what clBuildProgram's NULL options (src/opencl/renderer_opencl.cpp:50) change in the
//       reference's USER-level expressions on this GPU, read from the LLVM IR and ISA of that build of the six kernel files:
//       * -ffp-contract=on: `a*b + c` inside ONE source expression is one fused multiply-add -- the left product when both
//         addends are products (clang's rule).  Sites: the camera's yaw rotation, every barycentric interpolation
//         A*b.x + B*b.y + C*b.z, random()'s dot + 1113.1*seed (in double), refract's two, 1 - z*z and the change of basis of
//         the hemisphere sample, the `indirect +=` terms, the 25-sample blend.  mad() below.
//       * float division and sqrt at the default 2.5 / 3 ulp: x / y = ldexp(frexp_mant(x) * v_rcp_f32(frexp_mant(y)),
//         frexp_exp(x) - frexp_exp(y)) -- the film position x / width, 1 / det in intersectTriangle, (25 - k) / 25 --
//         and v_sqrt_f32 with the backend's denormal scaling.  fdiv() / rcp() / sqrt_user() below.
//       With it the HIP path is bit-identical to the reference's kernels as RendererOpenCL builds them.
template <int DEVLIBM>
struct Math {
  static constexpr bool kShipped = DEVLIBM == 2;
  static __device__ __forceinline__ float mad(float a, float b, float c) { return kShipped ? __builtin_fmaf(a, b, c) : a * b + c; }
  static __device__ __forceinline__ double mad(double a, double b, double c) { return kShipped ? __builtin_fma(a, b, c) : a * b + c; }
  static __device__ __forceinline__ float rcp(float x) {   // 1.0f / x
    if (kShipped) return __builtin_amdgcn_ldexpf(__builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(x)), -__builtin_amdgcn_frexp_expf(x));
    return 1.0f / x;
  }
  static __device__ __forceinline__ float fdiv(float a, float b) {
    if (kShipped)
      return __builtin_amdgcn_ldexpf(__builtin_amdgcn_frexp_mantf(a) * __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(b)),
                                     __builtin_amdgcn_frexp_expf(a) - __builtin_amdgcn_frexp_expf(b));
    return a / b;
  }
  static __device__ __forceinline__ float sqrt_user(float x) { return kShipped ? sqrt_in_distance(x) : __builtin_sqrtf(x); }
  // x / 25.0f (the 25-sample blend's attenuation): the constant's half of the fast division is folded at compile time in the
  // reference build -- frexp_mant(25) = 0.78125, its reciprocal 1.28f = 0x3fa3d70a, frexp_exp(25) = 5
  static __device__ __forceinline__ float div25(float x) {
    if (kShipped) return __builtin_amdgcn_ldexpf(__builtin_amdgcn_frexp_mantf(x) * __uint_as_float(0x3fa3d70au), __builtin_amdgcn_frexp_expf(x) - 5);
    return x / (float)25;
  }
  static __device__ __forceinline__ float rsqrt(float x) {
    if (DEVLIBM) {
      const bool small = x < 1.17549435e-38f;
      const float r = __builtin_amdgcn_rsqf(small ? x * 0x1p+24f : x);
      return small ? r * 4096.0f : r;
    }
    return (float)(1.0 / sqrt((double)x));
  }
  static __device__ __forceinline__ float sqrt_in_distance(float x) {
    if (DEVLIBM) {
      const bool small = x < 1.17549435e-38f;
      return __builtin_ldexpf(__builtin_amdgcn_sqrtf(__builtin_ldexpf(x, small ? 32 : 0)), small ? -16 : 0);
    }
    return __builtin_sqrtf(x);
  }
  static __device__ __forceinline__ float cos(float x) { return DEVLIBM ? ::cosf(x) : (float)::cos((double)x); }
  static __device__ __forceinline__ float sin(float x) { return DEVLIBM ? ::sinf(x) : (float)::sin((double)x); }
  static __device__ __forceinline__ float clamp01(float x) {
    return DEVLIBM ? __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f) : __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f);
  }
};

template <int DEVLIBM>
__device__ inline V4 normalize4(V4 p) {
  if (p.x == 0.0f && p.y == 0.0f && p.z == 0.0f && p.w == 0.0f) return p;
  float l2 = dot4(p, p);
  if (l2 < 1.17549435e-38f) {
    p = scale4(0x1p+86f, p);
    l2 = dot4(p, p);
  } else if (l2 == __builtin_inff()) {
    p = scale4(0x1p-66f, p);
    l2 = dot4(p, p);
    if (l2 == __builtin_inff()) {
      p = mk4(__builtin_copysignf(__builtin_isinf(p.x) ? 1.0f : 0.0f, p.x), __builtin_copysignf(__builtin_isinf(p.y) ? 1.0f : 0.0f, p.y),
              __builtin_copysignf(__builtin_isinf(p.z) ? 1.0f : 0.0f, p.z), __builtin_copysignf(__builtin_isinf(p.w) ? 1.0f : 0.0f, p.w));
      l2 = dot4(p, p);
    }
  }
  return scale4(Math<DEVLIBM>::rsqrt(l2), p);
}

template <int DEVLIBM>
__device__ inline float distance4(V4 a, V4 b) {
  V4 d = sub4(a, b);
  float l2 = dot4(d, d);
  if (l2 < 1.17549435e-38f) {
    d = scale4(0x1p+86f, d);
    return Math<DEVLIBM>::sqrt_in_distance(dot4(d, d)) * 0x1p-86f;
  } else if (l2 == __builtin_inff()) {
    d = scale4(0x1p-66f, d);
    return Math<DEVLIBM>::sqrt_in_distance(dot4(d, d)) * 0x1p+66f;
  }
  return Math<DEVLIBM>::sqrt_in_distance(l2);
}

// fmod(x, M_PI), bit for bit.  fmod is exact -- the result IS x - n y for n = trunc(x / y), a double -- so any way of finding n
// gives the library's bits: n from one multiplication by 1 / pi (within one of the truth while n < 2^40), the remainder from ONE
// fused multiply-add (no rounding: the exact value is a double), a second one when n was off by one (the sign, resp. r >= y,
// of the rounded remainder tells: rounding never crosses 0 or the double y).  The library's fmod is a long-division loop.
// (tests/test_gpu_parity.py holds it against fmod() over the arguments random() makes and over edge cases.)
__device__ __forceinline__ double fmod_pi(double x) {
  const double ax = __builtin_fabs(x);
  if (!(ax < 0x1p+40)) return fmod(x, M_PI);   // (huge, infinite or NaN: the library's own)
  double n = __builtin_floor(ax * 0.31830988618379067);
  double r = __builtin_fma(-n, M_PI, ax);
  if (r < 0.0) r = __builtin_fma(-(n - 1.0), M_PI, ax);
  else if (r >= M_PI) r = __builtin_fma(-(n + 1.0), M_PI, ax);
  return __builtin_copysign(r, x);
}

// acc.cl:63-66: float dot, then double add / fmod / sin / mul, then float fract
template <int M = 0>
__device__ inline float random_(float uvx, float uvy, float seed) {
  float d = dot2(uvx, uvy, 12.9898f, 78.233f);
  double x = Math<M>::kShipped ? __builtin_fma(1113.1, (double)seed, (double)d) : (double)d + 1113.1 * (double)seed;
  float a = (float)(sin(fmod_pi(x)) * 43758.5453);
  return a - __builtin_floorf(a);
}

// ---------------------------------------------------------------- traversal
// acc.cl:72-111 on the re-tiled triangle.  PROGRAM picks the epsilon flavour: basic.cl:78 compares in
// float against 1e-7f, basic_lighting.cl:4 in double against 1e-7, the others in double against 1e-4.
// (rank8 / octant / prim: the order table of SceneDev::rank8 for walks over the backend's own tree; null = reference order)
template <int PROGRAM>
__device__ __forceinline__ bool intersect_triangle_data(const float4 t0, const float4 t1, const float4 t2, const Ray& ray, Hit& pl, bool fastRcp = false,
                                                        const uint32_t* rank8 = nullptr, uint32_t octant = 0u, int prim = 0) {
  V4 A = mk4(t0.x, t0.y, t0.z, 1.0f);
  V4 v0v1 = mk4(t0.w, t1.x, t1.y, 0.0f);
  V4 v0v2 = mk4(t1.z, t1.w, t2.x, 0.0f);
  V4 pvec = cross4(ray.d, v0v2);
  float det = dot4(v0v1, pvec);
  // The double compares of the reference, `(double)fabs(det) < 1e-7` and `< 1e-4`, as float compares against the smallest
  // float that is >= the double constant (0x33d6bf95, 0x38d1b718): for a float x, (double)x < c  <=>  x < that float.  Exact
  // (tests/test_capi_cpu.py checks the two constants), and one v_cmp_lt_f32 instead of v_cvt_f64_f32 + v_cmp_lt_f64.
  if (PROGRAM == kBasic || PROGRAM == kCustom) {
    if (__builtin_fabsf(det) < 0.0000001f) return false;
  } else if (PROGRAM == kBasicLighting) {
    if (__builtin_fabsf(det) < __uint_as_float(0x33d6bf95u)) return false;
  } else {
    if (__builtin_fabsf(det) < __uint_as_float(0x38d1b718u)) return false;
  }
  // (fastRcp: SceneDev::fastRcp, the as-shipped flavour's 1 / det -- Math<2>::rcp; wave-uniform)
  float invDet = fastRcp ? Math<2>::rcp(det) : 1.0f / det;
  V4 tvec = sub4(ray.o, A);
  float u = dot4(tvec, pvec) * invDet;
  if (u < 0.0f || u > 1.0f) return false;
  V4 qvec = cross4(tvec, v0v1);
  float v = dot4(ray.d, qvec) * invDet;
  if (v < 0.0f || u + v > 1.0f) return false;
  float tt = dot4(v0v2, qvec) * invDet;
  bool take = tt < pl.t;   // no t > 0 test in the reference
  if (tt == pl.t && rank8 != nullptr && pl.hitType == 1)   // (rare: the same t, bit for bit, from two triangles)
    take = rank8[8 * (size_t)prim + octant] < rank8[8 * (size_t)pl.prim + octant];
  if (take) {
    pl.t = tt; pl.u = u; pl.v = v;
    return true;
  }
  return false;
}

template <int PROGRAM>
__device__ __forceinline__ bool intersect_triangle(const float4* __restrict__ tris, int prim, const Ray& ray, Hit& pl, bool fastRcp = false,
                                                   const uint32_t* rank8 = nullptr, uint32_t octant = 0u) {
  const float4* t = tris + 3 * (size_t)prim;
  return intersect_triangle_data<PROGRAM>(t[0], t[1], t[2], ray, pl, fastRcp, rank8, octant, prim);
}
template <int PROGRAM>
__device__ __forceinline__ bool intersect_triangle_lds(uint32_t ldsTris, int prim, const Ray& ray, Hit& pl, bool fastRcp = false) {
  const LdsF4 t = (LdsF4)(size_t)(ldsTris + 48u * (uint32_t)prim);
  return intersect_triangle_data<PROGRAM>(ld_lds(t), ld_lds(t + 1), ld_lds(t + 2), ray, pl, fastRcp);
}

// The same test for the packet walks, where the triangle sits in SGPRs and the whole wave runs it anyway: no per-lane early
// outs (each one costs an exec-mask save / restore and a copy of the payload through every exit), one wave-uniform exit after
// the `u` test, selects at the end.  `active` = this lane visits the leaf.  Same predicates in the same negated forms as the
// reference's `return false` tests, so a NaN falls through exactly where it does there; lanes that are not active compute on
// whatever their registers hold and are masked out of every decision (no float exception traps on this path).
template <int PROGRAM>
__device__ __forceinline__ bool intersect_triangle_packet(const float4 t0, const float4 t1, const float4 t2, const Ray& ray, Hit& pl,
                                                          bool active, int prim, bool fastRcp = false, const uint32_t* rank8 = nullptr,
                                                          uint32_t octant = 0u) {
  const V4 A = mk4(t0.x, t0.y, t0.z, 1.0f);
  const V4 v0v1 = mk4(t0.w, t1.x, t1.y, 0.0f);
  const V4 v0v2 = mk4(t1.z, t1.w, t2.x, 0.0f);
  const V4 pvec = cross4(ray.d, v0v2);
  const float det = dot4(v0v1, pvec);
  const float eps = (PROGRAM == kBasic || PROGRAM == kCustom) ? 0.0000001f
                    : (PROGRAM == kBasicLighting) ? __uint_as_float(0x33d6bf95u) : __uint_as_float(0x38d1b718u);
  bool ok = active && !(__builtin_fabsf(det) < eps);
  const float invDet = fastRcp ? Math<2>::rcp(det) : 1.0f / det;
  const V4 tvec = sub4(ray.o, A);
  const float u = dot4(tvec, pvec) * invDet;
  ok = ok && !(u < 0.0f || u > 1.0f);
  if (__builtin_amdgcn_ballot_w64(ok) == 0ull) return false;
  const V4 qvec = cross4(tvec, v0v1);
  const float v = dot4(ray.d, qvec) * invDet;
  ok = ok && !(v < 0.0f || u + v > 1.0f);
  const float tt = dot4(v0v2, qvec) * invDet;
  bool take = tt < pl.t;
  if (rank8 != nullptr && __builtin_amdgcn_ballot_w64(ok && tt == pl.t && pl.hitType == 1) != 0ull) {   // (rare)
    if (ok && tt == pl.t && pl.hitType == 1) take = rank8[8 * (size_t)prim + octant] < rank8[8 * (size_t)pl.prim + octant];
  }
  ok = ok && take;
  pl.t = ok ? tt : pl.t;
  pl.u = ok ? u : pl.u;
  pl.v = ok ? v : pl.v;
  pl.prim = ok ? prim : pl.prim;
  pl.hitType = ok ? 1 : pl.hitType;
  return ok;
}

// ... and for the any-hit packet walk: the caller of a shadow ray reads nothing but "was anything accepted" (acc.cl:276), the
// first accepted triangle ends the lane's walk, so `t < payload.t` is always a test against the ray's initial tmax and no
// payload needs to be carried or updated at all.
template <int PROGRAM>
__device__ __forceinline__ bool intersect_triangle_anyhit(const float4 t0, const float4 t1, const float4 t2, const Ray& ray, float tmax,
                                                          bool active, bool fastRcp = false) {
  const V4 A = mk4(t0.x, t0.y, t0.z, 1.0f);
  const V4 v0v1 = mk4(t0.w, t1.x, t1.y, 0.0f);
  const V4 v0v2 = mk4(t1.z, t1.w, t2.x, 0.0f);
  const V4 pvec = cross4(ray.d, v0v2);
  const float det = dot4(v0v1, pvec);
  const float eps = (PROGRAM == kBasic || PROGRAM == kCustom) ? 0.0000001f
                    : (PROGRAM == kBasicLighting) ? __uint_as_float(0x33d6bf95u) : __uint_as_float(0x38d1b718u);
  bool ok = active && !(__builtin_fabsf(det) < eps);
  const float invDet = fastRcp ? Math<2>::rcp(det) : 1.0f / det;
  const V4 tvec = sub4(ray.o, A);
  const float u = dot4(tvec, pvec) * invDet;
  ok = ok && !(u < 0.0f || u > 1.0f);
  if (__builtin_amdgcn_ballot_w64(ok) == 0ull) return false;
  const V4 qvec = cross4(tvec, v0v1);
  const float v = dot4(ray.d, qvec) * invDet;
  ok = ok && !(v < 0.0f || u + v > 1.0f);
  const float tt = dot4(v0v2, qvec) * invDet;
  return ok && (tt < tmax);
}

// The per-lane traversal stack: entries [0, rows) live in LDS (column `lane`, row stride kBlock, so every wave access is one
// conflict-free row; rows = what the launch allocated, at most kLdsStack), deeper entries (only in the DEEP form, which the host
// launches whenever the scene's measured height exceeds the rows) in a per-lane scratch array.
template <bool B, class T, class F> struct SelectType { using type = T; };
template <class T, class F> struct SelectType<false, T, F> { using type = F; };

template <bool DEEP>
struct Stack {
  int* lds;            // &lds_stack[threadIdx.x]
  int rows;            // LDS rows the launch gave each lane (FrameParams::ldsRows): the DEEP form keeps entries [0, rows) there and the
                       // rest in private memory, whatever the host allocated; the other form is launched only when rows >= the tree's height
  int deep[DEEP ? kMaxStack : 1];
  __device__ __forceinline__ void push(int sp, int v) {
    if (!DEEP || sp < rows) lds[sp * kBlock] = v; else deep[sp - rows] = v;
  }
  __device__ __forceinline__ int pop(int sp) {
    if (!DEEP || sp < rows) return lds[sp * kBlock];
    return deep[sp - rows];
  }
  // A stack position as the loop variable of the per-lane walk: the LDS row pointer itself when the whole stack is in LDS
  // (stepping it by a row is one add; an index would cost a shift-add per access), the entry index otherwise.
  using Pos = typename SelectType<DEEP, int, int*>::type;
  __device__ __forceinline__ Pos bottom() { if constexpr (DEEP) return 0; else return lds; }
  __device__ __forceinline__ bool above_bottom(Pos p) { if constexpr (DEEP) return p > 0; else return p > lds; }
  __device__ __forceinline__ Pos below(Pos p) { if constexpr (DEEP) return p > 0 ? p - 1 : 0; else return p > lds ? p - kBlock : lds; }
  __device__ __forceinline__ Pos step(Pos p, bool up) { if constexpr (DEEP) return p + (up ? 1 : -1); else return p + (up ? kBlock : -kBlock); }
  __device__ __forceinline__ int load(Pos p) { if constexpr (DEEP) return pop(p); else return *p; }
  __device__ __forceinline__ void store(Pos p, int v) { if constexpr (DEEP) push(p, v); else *p = v; }
};

// The reference's own `int nodesToVisit[64]` (acc.cl:137) in private memory, for the walks of the non-counting kernels that must
// follow the caller's tree in the reference's order: rays with a non-finite component (the image's centre row / column), and
// every ray of a scene whose boxes do not nest.  Rare, so their stack costs no LDS (the launches of those kernels reserve two
// rows per wave, for the packet walks).
struct ScratchStack {
  int e[kMaxStack];
  using Pos = int;
  __device__ __forceinline__ Pos bottom() { return 0; }
  __device__ __forceinline__ bool above_bottom(Pos p) { return p > 0; }
  __device__ __forceinline__ Pos below(Pos p) { return p > 0 ? p - 1 : 0; }
  __device__ __forceinline__ Pos step(Pos p, bool up) { return p + (up ? 1 : -1); }
  __device__ __forceinline__ int load(Pos p) { return e[p]; }
  __device__ __forceinline__ void store(Pos p, int v) { e[p] = v; }
};

// acc.cl:132-171 (intersect) and :173-217 (intersectIgnorePrimitiveIndex): same node order (near child
// first by dirIsNeg[axis]), same box test (acc.cl:113-130, no clipping against the closest hit), leaf =
// primitives[primitivesOffset] only (the reference's leaf loop never adds i; re-testing the same triangle
// primitiveCount times leaves the payload unchanged after the first test, so it is tested once here).
// The reference's slab test (acc.cl:113-130), compare for compare.  Its NaN behaviour matters: an axis-parallel ray
// has invDir = +-inf, and (bound - origin) * inf is NaN whenever the origin lies on the bound's plane; the
// reference's `if (tyMin > tMin) tMin = tyMin` forms then keep or drop the NaN in a definite way.
__device__ __forceinline__ bool box_test_reference(float lox, float loy, float loz, float hix, float hiy, float hiz, const Ray& ray,
                                                   float ix, float iy, float iz, bool nx, bool ny, bool nz) {
  float tMin = ((nx ? hix : lox) - ray.o.x) * ix;
  float tMax = ((nx ? lox : hix) - ray.o.x) * ix;
  const float tyMin = ((ny ? hiy : loy) - ray.o.y) * iy;
  const float tyMax = ((ny ? loy : hiy) - ray.o.y) * iy;
  bool hit = !(tMin > tyMax || tyMin > tMax);
  if (tyMin > tMin) tMin = tyMin;
  if (tyMax < tMax) tMax = tyMax;
  const float tzMin = ((nz ? hiz : loz) - ray.o.z) * iz;
  const float tzMax = ((nz ? loz : hiz) - ray.o.z) * iz;
  hit = hit && !(tMin > tzMax || tzMin > tMax);
  if (tzMin > tMin) tMin = tzMin;
  if (tzMax < tMax) tMax = tzMax;
  return hit && (tMax > 0.0f);
}

// The same predicate for rays whose origin and inverse direction are all finite: then no product below can be NaN
// (finite - finite is finite, finite * finite is finite or +-inf), per axis the entry distance the reference picks by
// dirIsNeg is min(t0, t1) and the exit distance max(t0, t1) (rounding is monotonic), and its chain of compares and
// updates is exactly "largest entry <= smallest exit, and smallest exit > 0".  Same result, a third fewer instructions
// (no sign selects: v_min/v_max3 instead).
__device__ __forceinline__ bool box_test_finite(float lox, float loy, float loz, float hix, float hiy, float hiz, const Ray& ray,
                                                float ix, float iy, float iz) {
  const float tx0 = (lox - ray.o.x) * ix, tx1 = (hix - ray.o.x) * ix;
  const float ty0 = (loy - ray.o.y) * iy, ty1 = (hiy - ray.o.y) * iy;
  const float tz0 = (loz - ray.o.z) * iz, tz1 = (hiz - ray.o.z) * iz;
  const float tEnter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fminf(tz0, tz1));
  const float tExit = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fmaxf(tz0, tz1));
  return tEnter <= tExit && tExit > 0.0f;
}

// Slab test as a wave-wide lane mask, for the packet walks.  NEG >= 0: the direction signs of every ray of the wave are known at
// compile time (bit a = component a negative), so the entry plane of an axis is simply the box's max (negative direction) or
// min and no v_min / v_max is needed -- same predicate as box_test_finite: (bound - o) * inv is monotonic in bound, so the
// selected product IS the min (or max) of the two, up to the sign of a zero, which neither max3 / min3 nor the compares can
// tell apart.  NEG < 0: signs unknown, box_test_finite's min / max form.
// The wave's lane mask of a slab test, as the AND of the two compares' masks: `ballot(a && b)` makes the compiler
// materialise the bool in a VGPR and compare it again (two vector instructions per test); two ballots and a scalar AND do not.
template <int NEG>   // NEG < 0: signs unknown (box_test_finite's min / max form)
__device__ __forceinline__ unsigned long long box_mask(float lox, float loy, float loz, float hix, float hiy, float hiz, const Ray& ray,
                                                       float ix, float iy, float iz) {
  float tEnter, tExit;
  if (NEG >= 0) {
    const float nearX = (NEG & 1) ? hix : lox, farX = (NEG & 1) ? lox : hix;
    const float nearY = (NEG & 2) ? hiy : loy, farY = (NEG & 2) ? loy : hiy;
    const float nearZ = (NEG & 4) ? hiz : loz, farZ = (NEG & 4) ? loz : hiz;
    tEnter = __builtin_fmaxf(__builtin_fmaxf((nearX - ray.o.x) * ix, (nearY - ray.o.y) * iy), (nearZ - ray.o.z) * iz);
    tExit = __builtin_fminf(__builtin_fminf((farX - ray.o.x) * ix, (farY - ray.o.y) * iy), (farZ - ray.o.z) * iz);
  } else {
    const float tx0 = (lox - ray.o.x) * ix, tx1 = (hix - ray.o.x) * ix;
    const float ty0 = (loy - ray.o.y) * iy, ty1 = (hiy - ray.o.y) * iy;
    const float tz0 = (loz - ray.o.z) * iz, tz1 = (hiz - ray.o.z) * iz;
    tEnter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fminf(tz0, tz1));
    tExit = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fmaxf(tz0, tz1));
  }
  return __builtin_amdgcn_ballot_w64(tEnter <= tExit) & __builtin_amdgcn_ballot_w64(tExit > 0.0f);
}

template <bool FINITE>
__device__ __forceinline__ bool box_test(float lox, float loy, float loz, float hix, float hiy, float hiz, const Ray& ray, float ix,
                                         float iy, float iz, bool nx, bool ny, bool nz) {
  return FINITE ? box_test_finite(lox, loy, loz, hix, hiy, hiz, ray, ix, iy, iz)
                : box_test_reference(lox, loy, loz, hix, hiy, hiz, ray, ix, iy, iz, nx, ny, nz);
}

// The per-lane walk (bounce rays; shadow rays where a scene's are incoherent; camera rays of the few waves that cannot form a
// packet).  One node per iteration, as few instructions as possible: since launches stopped being dominated by their slowest
// wavefront (several samples per launch, lt_capi.hip) it is bound by occupancy times instructions on the node-to-node chain:
//  * a leaf found in iteration i is only *noted* (`pend`); its triangle is fetched and tested at the top of iteration i+1,
//    right behind the issue of the next node's loads, so both latencies overlap (testing it inside iteration i costs 25 %).
//    Exact: the reference's traversal never reads the payload (no clipping against payload.t, acc.cl:113-130), and a lane
//    still tests its leaves in reference order;
//  * the stack is plain -- push and pop go to the lane's LDS column -- and the step has no branch: every lane reads the entry
//    below its top and stores once (interior lanes the far child above the top, the others the value they just read, back
//    where it was), selects decide what counts: +2 % over an if / else.  (Storing above the top unconditionally is another
//    +0.6 %, but a lane at a deepest leaf would write one row past the launch's LDS stack.)
//    (Under one launch per sample, where the chain per node mattered more than the instruction count, keeping the top entry
//    in a register and batching the triangle tests until 16 lanes held one were each worth a few per cent; with fused
//    launches they cost 5 % and 2-6 %.)
// ANYHIT (shadow rays, only when not counting work): the callers of a shadow ray read nothing but `hitType == 0`
// (acc.cl:276, gi.cl:295,:351), so the walk may stop at the first accepted triangle -- same pixels, fewer node visits
// than the reference algorithm performs.  The counting (STATS) instantiations never use it.
template <int PROGRAM, class STACK, bool STATS, bool FINITE, bool ANYHIT, bool LDSSCENE = false>
__device__ inline void traverse_nodes_impl(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, bool useIgnore, int ignore,
                                     Hit& pl, STACK& st, Counters& c) {
  const bool nx = ix < 0.0f, ny = iy < 0.0f, nz = iz < 0.0f;
  const uint32_t negBits = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);   // dirIsNeg[axis] = bit `axis` (axis <= 2: set_scene)
  const int ign = useIgnore ? ignore : -1;   // leaf offsets are >= 0
  int cur = 0;
  typename STACK::Pos sp = st.bottom();   // where the next entry goes
  int pend = -1;
  uint32_t pendCount = 0;
  bool alive = true;
  while (alive) {
    // 32-bit byte offset from the (wave-uniform) array base: lets the load use the SGPR-base + VGPR-offset form instead of
    // 64-bit address arithmetic on the node-to-node dependency chain (n_nodes * 32 < 2^32 is checked at set_scene)
    float4 a, b;   // a = min.x min.y min.z max.x ; b = max.y max.z offset count|axis<<16
    if constexpr (LDSSCENE) {
      const LdsF4 n = (LdsF4)(size_t)(sc.ldsNodes + ((uint32_t)cur << 5));
      a = ld_lds(n); b = ld_lds(n + 1);
    } else {
      const float4* n = (const float4*)((const char*)sc.nodes + ((uint32_t)cur << 5));
      a = n[0]; b = n[1];
    }
    // the entry below the top is read now, beside the node fetch, whether or not this node turns out to need it
    const typename STACK::Pos below = st.below(sp);   // (the bottom row, unused, for a lane that is about to end)
    const int popped = st.load(below);
    if (STATS) c.nodes++;
    LT_WAVE_COUNT(wInner);
    const bool hit = box_test<FINITE>(a.x, a.y, a.z, a.w, b.x, b.y, ray, ix, iy, iz, nx, ny, nz);
    const uint32_t meta = __float_as_uint(b.w);
    const int off = __float_as_int(b.z);
    const uint32_t count = meta & 0xffffu;
    const bool newLeaf = hit && count != 0 && off != ign;
    if (pend >= 0) {   // the leaf noted in the previous iteration
      LT_WAVE_COUNT(wTri);
      if (STATS) c.tris += pendCount;    // the reference *calls* intersectTriangle primitiveCount times
      if (LDSSCENE ? intersect_triangle_lds<PROGRAM>(sc.ldsTris, pend, ray, pl, sc.fastRcp != 0u)
                   : intersect_triangle<PROGRAM>(sc.tris, pend, ray, pl, sc.fastRcp != 0u)) {
        pl.prim = pend;
        pl.hitType = 1;
        if (ANYHIT) return;
      }
      pend = -1;
    }
    {   // branch-free step: selects instead of an interior / leaf-or-miss branch (fewer scalar instructions, no exec juggling)
      const bool inner = hit && count == 0;
      const bool neg = (negBits >> ((meta >> 16) & 0xffu)) & 1u;
      const int left = cur + 1;
      // one unconditional store: interior lanes push the far child; the others rewrite the entry they just read (a lane at
      // a deepest leaf has no row `sp` to scribble on)
      st.store(inner ? sp : below, inner ? (neg ? left : off) : popped);
      pend = newLeaf ? off : pend;
      if (STATS) pendCount = newLeaf ? count : pendCount;
      alive = inner || st.above_bottom(sp);
      cur = inner ? (neg ? off : left) : popped;
      sp = st.step(sp, inner);
    }
  }
  if (pend >= 0) {
    LT_WAVE_COUNT(wTri);
    if (STATS) c.tris += pendCount;
    if (LDSSCENE ? intersect_triangle_lds<PROGRAM>(sc.ldsTris, pend, ray, pl, sc.fastRcp != 0u)
                 : intersect_triangle<PROGRAM>(sc.tris, pend, ray, pl, sc.fastRcp != 0u)) {
      pl.prim = pend;
      pl.hitType = 1;
    }
  }
}

// The per-lane walk over the backend's own tree (finite rays of the non-counting kernels; lt_retree.hpp says why any order over
// any enclosing hierarchy finds the reference's set of leaves), collapsed into 4-wide groups (lt_retree::collapse_wide).  One
// loop, one 64-byte record per step, whatever the record is: a group -- its four child slots are tested conservatively on their
// 16-bit boxes and the links of those entered go on the lane's stack -- or a leaf, which gets the reference's own slab test
// (acc.cl:113-130) of its own box and the reference's triangle test (acc.cl:72-111); then the next entry is popped.  Closest-hit
// walks settle equal-t ties with the reference's leaf order (SceneDev::rank8); any-hit walks (shadow rays: their callers read
// hitType only) stop at the first accepted hit.  The stack: kOwnRows dwords per lane in LDS (column `lane` of the wave's rows),
// the rest in private memory.
//
// The conservative test on a quantised box [O + ql S, O + qh S] (per axis; O, S floats taken as exact reals; the build checks in
// double that O + ql S <= lo - 8u|lo| and O + qh S >= hi + 8u|hi| for the node's true box, u = 2^-24).  Per lane, once per ray:
//     sI = fl(S inv),  p = fl(o inv),  c = fma(O, inv, -p),  m = 2^-21 (E |inv| + |O inv| + |p|) + 2^-140,  E = 65535 S   (per axis)
//     cN = fl(c - m),  cF = fl(c + m),  (qn, qf) = (ql, qh) where inv >= 0, (qh, ql) where inv < 0
// and per node     tN = max_a fma(qn_a, sI_a, cN_a),  tF = min_a fma(qf_a, sI_a, cF_a),  accept iff tF >= max(tN, 0+).
// It accepts whenever the reference's test (acc.cl:113-130) accepts the node's true box (hence any box inside it).  Per axis, inv > 0
// (the other sign mirrors): t(q) := fma(q, sI, c) = (O + q S - o) inv + D with |D| <= u (q S |inv| + |o inv| + |O inv - p| + |t|)
// (1 + 2u) <= 2.02 u K, K = E |inv| + |O inv| + |p| -- the roundings of sI, p, c and of the fma itself; the reference computes
// tN_ref = fl(fl(lo - o) inv) >= (lo - o) inv - 2.01 u (|lo| + |o|) |inv|.  With O + ql S <= lo - 8u |lo|:
//     t(ql) - tN_ref <= -8u |lo inv| + 2.01 u |lo inv| + 2.01 u |o inv| + |D| <= 4.1 u K,
// and folding the margin into the constant costs one more rounding, u (|c| + m) <= 1.1 u K: fma(ql, sI, cN) <= tN_ref as soon as
// m >= 5.2 u K; m = 2^-21 K = 8 u K.  Symmetrically fma(qh, sI, cF) >= tF_ref.  So tN <= tN_ref and tF >= tF_ref axis by axis, and
// tF_ref >= max(tN_ref, 0+) implies tF >= max(tN, 0+).  The margins are PER AXIS on purpose: an axis the ray is almost parallel
// to has a huge |inv| and hence huge absolute errors, but they concern that axis' own (equally huge) entry and exit distances
// only -- one margin for all three axes (the form the packet walks use, where origins are small multiples of the directions)
// makes such a ray accept every box within 2^-19 E |inv| of its path: a handful of rays per million that walk half the tree and
// hold their wavefronts' launch back (measured: the 1 M-triangle wall's bounce stages twice as long).  All magnitudes stay below
// 2^102 (|o|, |O|, E < 2^41, |inv| < 2^60: packet_ray_ok, checked per wave; the scene's bounds below 2^40:
// lt_retree::collect_leaves), gradual underflow adds at most 2^-148 per operation: the 2^-140.
// tests/test_own_hierarchy_cpu.py tries the inequality on random and grazing rays against quantised boxes made by the build's
// own arithmetic.
struct Own16Ray { float sx, sy, sz, nx, ny, nz, fx, fy, fz; };   // sI; cN; cF
__device__ __forceinline__ Own16Ray own16_ray(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz) {
  Own16Ray r;
  const F8v fr = *(ConstF8)((unsigned long long)sc.wide - 32ull);   // origin.xyz - step.xyz -
  const float px = ray.o.x * ix, py = ray.o.y * iy, pz = ray.o.z * iz;
  r.sx = fr.s4 * ix; r.sy = fr.s5 * iy; r.sz = fr.s6 * iz;
  const float cx = __builtin_fmaf(fr.s0, ix, -px), cy = __builtin_fmaf(fr.s1, iy, -py), cz = __builtin_fmaf(fr.s2, iz, -pz);
  const float mx = (__builtin_fabsf(65535.0f * fr.s4 * ix) + __builtin_fabsf(fr.s0 * ix) + __builtin_fabsf(px)) * 0x1p-21f + 0x1p-140f;
  const float my = (__builtin_fabsf(65535.0f * fr.s5 * iy) + __builtin_fabsf(fr.s1 * iy) + __builtin_fabsf(py)) * 0x1p-21f + 0x1p-140f;
  const float mz = (__builtin_fabsf(65535.0f * fr.s6 * iz) + __builtin_fabsf(fr.s2 * iz) + __builtin_fabsf(pz)) * 0x1p-21f + 0x1p-140f;
  r.nx = cx - mx; r.ny = cy - my; r.nz = cz - mz;
  r.fx = cx + mx; r.fy = cy + my; r.fz = cz + mz;
  return r;
}
__device__ __forceinline__ bool own16_box_test(const uint4 q, const Own16Ray& r, bool negx, bool negy, bool negz) {
  const float lx = (float)(q.x & 0xffffu), hx = (float)(q.y >> 16), ly = (float)(q.x >> 16), hy = (float)(q.z & 0xffffu), lz = (float)(q.y & 0xffffu),
              hz = (float)(q.z >> 16);
  const float tN = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(negx ? hx : lx, r.sx, r.nx), __builtin_fmaf(negy ? hy : ly, r.sy, r.ny)),
                                   __builtin_fmaf(negz ? hz : lz, r.sz, r.nz));
  const float tF = __builtin_fminf(__builtin_fminf(__builtin_fmaf(negx ? lx : hx, r.sx, r.fx), __builtin_fmaf(negy ? ly : hy, r.sy, r.fy)),
                                   __builtin_fmaf(negz ? lz : hz, r.sz, r.fz));
  return tF >= __builtin_fmaxf(tN, __uint_as_float(1u));
}

// One step of that walk for one lane: the record `e` (a group, or a leaf: bit 31) is fetched and dealt with, the next entry
// popped into `e`.  Returns true when the walk is over -- the stack is empty, or an any-hit walk has accepted a hit.
struct OwnRay {   // what a lane keeps about its ray while it walks
  Own16Ray qr;
  uint32_t ignLink;   // link of the leaf the ray starts on (never entered: acc.cl:188)
  uint32_t octant;
  bool negx, negy, negz;
};
__device__ __forceinline__ OwnRay own_ray(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, int ign) {
  OwnRay w;
  w.qr = own16_ray(sc, ray, ix, iy, iz);
  w.negx = ix < 0.0f; w.negy = iy < 0.0f; w.negz = iz < 0.0f;
  w.octant = (w.negx ? 1u : 0u) | (w.negy ? 2u : 0u) | (w.negz ? 4u : 0u);
  // (ign = -1 gives the link of a leaf record that does not exist: it equals no slot's)
  w.ignLink = 0x80000000u | (sc.nWide + (uint32_t)ign);
  return w;
}
template <int PROGRAM, bool ANYHIT, int ROWS = kOwnRows>
__device__ __forceinline__ bool own_walk_step(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, const OwnRay& w, Hit& pl, int* col,
                                              int* deep, uint32_t& e, int& sp) {
  const uint4* rec = (const uint4*)((const char*)sc.wide + ((size_t)(e & 0x7fffffffu) << 6));
  const uint4 s0 = rec[0], s1 = rec[1], s2 = rec[2], s3 = rec[3];
  if ((int)e < 0) {   // a leaf: A e1.x | e1.yz e2.xy | e2.z lo | hi prim
    // The triangle first, the reference's test of the leaf's own box for the lanes that hit it: the box is what the group's
    // conservative test has just let this lane through, so it hardly ever fails, while most triangles are missed -- and a test
    // that some lane of the wavefront needs is paid for by all of them.  (Accepted = both, in either order.)
    const int prim = (int)s3.w;
    const float4 t0 = make_float4(__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z), __uint_as_float(s0.w));
    const float4 t1 = make_float4(__uint_as_float(s1.x), __uint_as_float(s1.y), __uint_as_float(s1.z), __uint_as_float(s1.w));
    Hit trial = pl;
    if (intersect_triangle_data<PROGRAM>(t0, t1, make_float4(__uint_as_float(s2.x), 0.0f, 0.0f, 0.0f), ray, trial, sc.fastRcp != 0u,
                                         ANYHIT ? nullptr : sc.rank8, w.octant, prim) &&
        box_test_finite(__uint_as_float(s2.y), __uint_as_float(s2.z), __uint_as_float(s2.w), __uint_as_float(s3.x), __uint_as_float(s3.y),
                        __uint_as_float(s3.z), ray, ix, iy, iz)) {
      pl = trial;
      pl.prim = prim;
      pl.hitType = 1;
      if (ANYHIT) return true;
    }
  } else {
    // the links of the slots entered go on the stack (leaves sit in a group's last slots: they come off first).  (Keeping the last
    // one in a register instead of pushing and popping it was measured: 5 % slower in lt_trace_kernel, 1 % faster in the render
    // kernels' walks.)
    const uint4 slot[4] = {s0, s1, s2, s3};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (own16_box_test(slot[k], w.qr, w.negx, w.negy, w.negz) && slot[k].w != w.ignLink) {
        if (sp < ROWS) col[sp * kBlock] = (int)slot[k].w; else deep[sp - ROWS] = (int)slot[k].w;
        sp++;
      }
    }
  }
  if (sp == 0) return true;
  sp--;
  e = (uint32_t)(sp < ROWS ? col[sp * kBlock] : deep[sp - ROWS]);
  return false;
}

template <int PROGRAM, bool ANYHIT>
__device__ inline void traverse_own_lane(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, int ign, Hit& pl, int* col) {
  const OwnRay w = own_ray(sc, ray, ix, iy, iz, ign);
  int deep[kOwnDeep];
  int sp = 0;
  uint32_t e = 0u;   // the root's group
  while (!own_walk_step<PROGRAM, ANYHIT>(sc, ray, ix, iy, iz, w, pl, col, deep, e, sp)) {}
}

// Compile-time configuration of one kernel instantiation.
template <bool DEEP_, bool STATS_, int DEVLIBM_, bool LDSSCENE_ = false>
struct Config {
  static constexpr bool kDeep = DEEP_;       // BVH deeper than the LDS stack: spill entries >= kLdsStack to scratch
  static constexpr bool kStats = STATS_;     // count rays / node visits / triangle tests
  static constexpr int kDevLibm = DEVLIBM_;  // math flavour: 0 portable, 1 device-library leaf math, 2 as shipped (Math<>)
  static constexpr bool kLdsScene = LDSSCENE_; // the per-lane walks read nodes and triangles from the workgroup's LDS copy (SceneDev::ldsNodes)
};

template <int PROGRAM, bool DEEP, bool STATS, bool SHADOW, bool LDSSCENE = false>
__device__ inline void traverse(const SceneDev& sc, const Ray& ray, bool useIgnore, int ignore, Hit& pl, Stack<DEEP>& st, Counters& c);

// ---------------------------------------------------------------- packet traversal
// The 64 camera rays of a wavefront (an 8x8 pixel square) visit almost the same nodes, so a wave whose rays are all finite walks
// the tree ONCE: node data arrives through scalar loads, each lane runs the same slab test against its own ray, control flow is
// scalar.  Two forms: traverse_packet (next) for the counting kernels -- over the caller's tree, one node per step, a 64-bit
// lane mask ("this lane hit every ancestor") beside each node index on a wave-uniform stack, so that every lane's tested nodes,
// the order of its leaves and its work counters are exactly those of its own reference traversal -- and packet_walk (further
// down; lt_walk_asm.hpp) for everything else: over the backend's own tree, order-free, mask-free.

template <int PROGRAM, bool STATS>
__device__ inline void traverse_packet(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, bool nxU, bool nyU, bool nzU,
                                       Hit& pl, int* ldsWave, Counters& c) {
  using u64 = unsigned long long;
  const ConstF4 nodes = (ConstF4)(unsigned long long)sc.nodes;
  const ConstF4 tris = (ConstF4)(unsigned long long)sc.tris;
  const int lane = (int)__lane_id();
  u64 mask = __ballot(1);
  const int leader = __ffsll((long long)mask) - 1;
  const uint32_t negBitsU = (nxU ? 1u : 0u) | (nyU ? 2u : 0u) | (nzU ? 4u : 0u);
  int cur = 0, sp = 0;
  for (;;) {
    const uint32_t ci = (uint32_t)__builtin_amdgcn_readfirstlane(cur);
    // the whole 32-byte record in ONE s_load_dwordx8 (as two dwordx4 the compiler sinks the first half into the `in`
    // branch below and the two scalar-load latencies add up)
    const F8v nd = *(ConstF8)(nodes + 2 * (size_t)ci);
    const float4 a = make_float4(nd.s0, nd.s1, nd.s2, nd.s3), b = make_float4(nd.s4, nd.s5, nd.s6, nd.s7);
    if (STATS && ((mask >> lane) & 1ull)) c.nodes++;
#ifdef LT_DEBUG_WAVE_COUNTERS
    if (lane == 0) c.wInner++;
    if ((mask >> lane) & 1ull) c.wOuter++;
#endif
    // every active lane runs the slab test (the wave pays for it anyway); masking the ballot with the wave-uniform `mask`
    // instead of branching on the lane's bit keeps the control flow scalar
    const u64 hmask = __builtin_amdgcn_ballot_w64(box_test_finite(a.x, a.y, a.z, a.w, b.x, b.y, ray, ix, iy, iz)) & mask;
    const uint32_t meta = __float_as_uint(b.w);
    const int off = __float_as_int(b.z);
    const uint32_t count = meta & 0xffffu;
    if (hmask != 0ull && count == 0u) {
      const bool neg = (negBitsU >> ((meta >> 16) & 0xffu)) & 1u;   // dirIsNeg[axis], shared by the wave
      const int farChild = neg ? (int)ci + 1 : off;
      if (lane == leader) {   // any lane may be switched off (image edge): the first active one writes the entry
        ldsWave[sp * kPacketEntry + 0] = farChild;
        ldsWave[sp * kPacketEntry + 1] = (int)(uint32_t)hmask;
        ldsWave[sp * kPacketEntry + 2] = (int)(uint32_t)(hmask >> 32);
      }
      sp++;
      cur = neg ? off : (int)ci + 1;
      mask = hmask;
      continue;
    }
    if (hmask != 0ull) {   // leaf whose box some lanes hit
#ifdef LT_DEBUG_WAVE_COUNTERS
      if (lane == 0) c.wTri++;
#endif
      if ((hmask >> lane) & 1ull) {
        if (STATS) c.tris += count;    // the reference *calls* intersectTriangle primitiveCount times
        const ConstF4 t = tris + 3 * (size_t)(uint32_t)off;
        const float4 t0 = ld_const(t), t1 = ld_const(t + 1), t2 = ld_const(t + 2);
        if (intersect_triangle_data<PROGRAM>(t0, t1, t2, ray, pl, sc.fastRcp != 0u)) {
          pl.prim = off;
          pl.hitType = 1;
        }
      }
    }
    if (sp == 0) break;
    sp--;
    cur = __builtin_amdgcn_readfirstlane(ldsWave[sp * kPacketEntry + 0]);      // same address in every lane: broadcast reads
    mask = (u64)(uint32_t)__builtin_amdgcn_readfirstlane(ldsWave[sp * kPacketEntry + 1]) |
           ((u64)(uint32_t)__builtin_amdgcn_readfirstlane(ldsWave[sp * kPacketEntry + 2]) << 32);
  }
}

// ---- the packet walks of the non-counting kernels, over the backend's own tree (lt_walk_asm.hpp describes the walk and proves
// its conservative interior test; this is its plain-C++ form: the any-hit walk of waves whose rays do not share a direction-sign
// octant (NEG < 0), and every packet walk of a build with -DLT_NO_ASM_WALKS).  sc.ownPairs[i]: interior node i -> its two
// children's boxes (pushed outwards) and references, leaf i -> the leaf's own box, its re-tiled triangle, its primitive offset
// (lt_own_pair_kernel).  The wave-uniform stack -- one dword per entry -- sits in the wave's two LDS rows.
struct PacketRayC { float px, py, pz, mg; };
template <int NEG>
__device__ __forceinline__ unsigned long long box_mask_cheap(float lox, float loy, float loz, float hix, float hiy, float hiz, float ix, float iy,
                                                            float iz, const PacketRayC& pr) {
  float tEnter, tExit;
  if (NEG >= 0) {
    const float nearX = (NEG & 1) ? hix : lox, farX = (NEG & 1) ? lox : hix;
    const float nearY = (NEG & 2) ? hiy : loy, farY = (NEG & 2) ? loy : hiy;
    const float nearZ = (NEG & 4) ? hiz : loz, farZ = (NEG & 4) ? loz : hiz;
    tEnter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(nearX, ix, -pr.px), __builtin_fmaf(nearY, iy, -pr.py)), __builtin_fmaf(nearZ, iz, -pr.pz));
    tExit = __builtin_fminf(__builtin_fminf(__builtin_fmaf(farX, ix, -pr.px), __builtin_fmaf(farY, iy, -pr.py)), __builtin_fmaf(farZ, iz, -pr.pz));
  } else {   // fma is monotonic in the bound and lo' <= hi': the smaller product is the near one
    const float tx0 = __builtin_fmaf(lox, ix, -pr.px), tx1 = __builtin_fmaf(hix, ix, -pr.px);
    const float ty0 = __builtin_fmaf(loy, iy, -pr.py), ty1 = __builtin_fmaf(hiy, iy, -pr.py);
    const float tz0 = __builtin_fmaf(loz, iz, -pr.pz), tz1 = __builtin_fmaf(hiz, iz, -pr.pz);
    tEnter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fminf(tz0, tz1));
    tExit = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fmaxf(tz0, tz1));
  }
  return __builtin_amdgcn_ballot_w64(tExit + pr.mg >= __builtin_fmaxf(tEnter, __uint_as_float(1u)));
}

template <int PROGRAM, int NEG, bool ANYHIT>
__device__ inline void packet_walk_cpp(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, int ign, Hit& pl, int* ldsWave) {
  using u64 = unsigned long long;
  const __attribute__((address_space(4))) char* const pairs = (const __attribute__((address_space(4))) char*)(unsigned long long)sc.ownPairs;
  const int lane = (int)__lane_id();
  PacketRayC pr;
  pr.px = ray.o.x * ix; pr.py = ray.o.y * iy; pr.pz = ray.o.z * iz;
  pr.mg = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(pr.px), __builtin_fabsf(pr.py)), __builtin_fabsf(pr.pz)) * 0x1p-19f + 0x1p-140f;
  const float tmax = pl.t;
  const bool fast = sc.fastRcp != 0u;
  u64 live = __builtin_amdgcn_ballot_w64(true);   // any-hit: the lanes still looking for an occluder
  uint32_t cur = 0u;
  int sp = 0;
  for (;;) {
    const F16v r = *(ConstF16)(pairs + (cur << 6));   // (bit 31 of a leaf reference falls off the 32-bit byte offset)
    if ((int)cur >= 0) {
      const u64 hmL = box_mask_cheap<NEG>(r.s0, r.s1, r.s2, r.s3, r.s4, r.s5, ix, iy, iz, pr) & live;
      const u64 hmR = box_mask_cheap<NEG>(r.s8, r.s9, r.sa, r.sb, r.sc, r.sd, ix, iy, iz, pr) & live;
      // (every active lane stores the same dword at the same address: no exec juggling for a "leader")
      if (hmL != 0ull) ldsWave[sp++] = __float_as_int(r.s6);
      if (hmR != 0ull) ldsWave[sp++] = __float_as_int(r.se);
    } else {
      // the leaf's own box, the reference's own test
      const u64 m = box_mask<NEG>(r.s9, r.sa, r.sb, r.sc, r.sd, r.se, ray, ix, iy, iz) & live;
      const int prim = __float_as_int(r.sf);
      if (m != 0ull) {
        const float4 t0 = make_float4(r.s0, r.s1, r.s2, r.s3), t1 = make_float4(r.s4, r.s5, r.s6, r.s7), t2 = make_float4(r.s8, 0.0f, 0.0f, 0.0f);
        const bool active = ((m >> lane) & 1ull) != 0ull && (!ANYHIT || prim != ign);
        if constexpr (ANYHIT) {
          live &= ~__builtin_amdgcn_ballot_w64(intersect_triangle_anyhit<PROGRAM>(t0, t1, t2, ray, tmax, active, fast));
          if (live == 0ull) break;
        } else {
          intersect_triangle_packet<PROGRAM>(t0, t1, t2, ray, pl, active, prim, fast, sc.rank8, (uint32_t)(NEG < 0 ? 0 : NEG));
        }
      }
    }
    if (sp == 0) break;
    cur = (uint32_t)__builtin_amdgcn_readfirstlane(ldsWave[--sp]);
  }
  if constexpr (ANYHIT) pl.hitType = ((live >> lane) & 1ull) != 0ull ? pl.hitType : 1;
}

// One packet walk: the hand-written form when the wave's rays share the direction-sign octant NEG, the C++ form otherwise.
template <int PROGRAM, int NEG, bool ANYHIT>
__device__ __forceinline__ void packet_walk(const SceneDev& sc, const Ray& ray, float ix, float iy, float iz, int ign, Hit& pl, int* ldsWave) {
#ifndef LT_NO_ASM_WALKS
  if constexpr (NEG >= 0 || ANYHIT) {
    const float eps = (PROGRAM == kBasic || PROGRAM == kCustom) ? 0.0000001f
                      : (PROGRAM == kBasicLighting) ? __uint_as_float(0x33d6bf95u) : __uint_as_float(0x38d1b718u);
    // the wave's first LDS row (byte address), where the walk parks its stack register's lanes (LT_ASM_WALK)
    const uint32_t ldsrow = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) int*)ldsWave);
    if constexpr (ANYHIT) {
      const unsigned long long open = packet_anyhit_walk<NEG>((const void*)sc.ownPairs, ray.o.x, ray.o.y, ray.o.z, ix, iy, iz, ray.d.x, ray.d.y,
                                                              ray.d.z, ray.d.w, pl.t, ign, eps, sc.fastRcp, __builtin_amdgcn_ballot_w64(true), ldsrow);
      pl.hitType = ((open >> __lane_id()) & 1ull) != 0ull ? pl.hitType : 1;
    } else {
      packet_closest_walk<NEG>((const void*)sc.ownPairs, (const void*)sc.rank8, ray.o.x, ray.o.y, ray.o.z, ix, iy, iz, ray.d.x, ray.d.y,
                               ray.d.z, ray.d.w, eps, sc.fastRcp, pl.t, pl.u, pl.v, pl.prim, pl.hitType, ldsrow);
    }
    return;
  }
#endif
  packet_walk_cpp<PROGRAM, NEG, ANYHIT>(sc, ray, ix, iy, iz, ign, pl, ldsWave);
}

// Which waves walk as a packet: every ray finite (lt_retree.hpp) and small enough that no product of the conservative test can
// overflow (lt_walk_asm.hpp: |origin| < 2^40, |1 / direction| < 2^60; the scene's bounds are below 2^40 or it has no own tree).
__device__ __forceinline__ bool packet_ray_ok(const Ray& ray, float ix, float iy, float iz) {
  return __builtin_fabsf(ix) < 0x1p+60f && __builtin_fabsf(iy) < 0x1p+60f && __builtin_fabsf(iz) < 0x1p+60f &&
         __builtin_fabsf(ray.o.x) < 0x1p+40f && __builtin_fabsf(ray.o.y) < 0x1p+40f && __builtin_fabsf(ray.o.z) < 0x1p+40f;
}

// Camera rays: packet traversal when the wave qualifies, the per-lane traversal otherwise.
template <int PROGRAM, bool DEEP, bool STATS>
__device__ inline void traverse_camera(const SceneDev& sc, const Ray& ray, Hit& pl, Stack<DEEP>& st, Counters& c) {
  const float ix = 1.0f / ray.d.x, iy = 1.0f / ray.d.y, iz = 1.0f / ray.d.z;
  const bool finite = __builtin_fabsf(ix) < __builtin_inff() && __builtin_fabsf(iy) < __builtin_inff() &&
                      __builtin_fabsf(iz) < __builtin_inff() && __builtin_fabsf(ray.o.x) < __builtin_inff() &&
                      __builtin_fabsf(ray.o.y) < __builtin_inff() && __builtin_fabsf(ray.o.z) < __builtin_inff();
  const bool nx = ix < 0.0f, ny = iy < 0.0f, nz = iz < 0.0f;
  const unsigned long long all = __builtin_amdgcn_ballot_w64(true), bx = __builtin_amdgcn_ballot_w64(nx), by = __builtin_amdgcn_ballot_w64(ny),
                           bz = __builtin_amdgcn_ballot_w64(nz);   // (the builtin takes the bool as is; __ballot(int) re-materialises it)
  const bool uniformSigns = (bx == 0ull || bx == all) && (by == 0ull || by == all) && (bz == 0ull || bz == all);
#ifndef LT_NO_PACKETS
  constexpr bool kPackets = !DEEP;   // the wave-uniform stack shares the LDS rows, which cover BVH heights <= kLdsStack
#else
  constexpr bool kPackets = false;
#endif
  // (the counting kernels walk the caller's tree as a packet, one node per step, lane masks on the stack: every lane's node and
  // triangle counts are those of its own reference traversal; the others need the backend's own tree)
  if (kPackets && __all(finite) && uniformSigns && (STATS || (sc.rank8 != nullptr && __all(packet_ray_ok(ray, ix, iy, iz))))) {
    if (STATS) c.rays++;
    if (STATS) traverse_packet<PROGRAM, STATS>(sc, ray, ix, iy, iz, bx != 0ull, by != 0ull, bz != 0ull, pl, st.lds - __lane_id(), c);
    else {
      int* const row = st.lds - __lane_id();
      switch ((bx != 0ull ? 1 : 0) | (by != 0ull ? 2 : 0) | (bz != 0ull ? 4 : 0)) {   // one specialisation per sign octant
        case 0: packet_walk<PROGRAM, 0, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        case 1: packet_walk<PROGRAM, 1, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        case 2: packet_walk<PROGRAM, 2, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        case 3: packet_walk<PROGRAM, 3, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        case 4: packet_walk<PROGRAM, 4, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        case 5: packet_walk<PROGRAM, 5, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        case 6: packet_walk<PROGRAM, 6, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
        default: packet_walk<PROGRAM, 7, false>(sc, ray, ix, iy, iz, -1, pl, row); break;
      }
    }
  } else {
    traverse<PROGRAM, DEEP, STATS, false>(sc, ray, false, 0, pl, st, c);
  }
}

template <int PROGRAM, bool DEEP, bool STATS, bool SHADOW, bool LDSSCENE>
__device__ inline void traverse(const SceneDev& sc, const Ray& ray, bool useIgnore, int ignore, Hit& pl,
                                Stack<DEEP>& st, Counters& c) {
  constexpr bool ANYHIT = SHADOW && !STATS;
  if (STATS) c.rays++;
  const float ix = 1.0f / ray.d.x, iy = 1.0f / ray.d.y, iz = 1.0f / ray.d.z;   // (float)(1.0/(double)x) == 1.0f/x
  // |x| < inf is false for NaN and for +-inf
  const bool finite = __builtin_fabsf(ix) < __builtin_inff() && __builtin_fabsf(iy) < __builtin_inff() &&
                      __builtin_fabsf(iz) < __builtin_inff() && __builtin_fabsf(ray.o.x) < __builtin_inff() &&
                      __builtin_fabsf(ray.o.y) < __builtin_inff() && __builtin_fabsf(ray.o.z) < __builtin_inff();
  if (__all(finite)) {
    // (not in the global-illumination programs: most of their shadow rays start at bounce hits and are incoherent, and the
    // extra walks cost their register-heavy kernels a third of their speed on small scenes)
    // (the walks over the own tree test interior boxes conservatively, which wants rays of ordinary magnitudes: packet_ray_ok)
    const bool ownWalks = sc.rank8 != nullptr && __all(packet_ray_ok(ray, ix, iy, iz));
    bool asPacket = ANYHIT && !DEEP && PROGRAM != kGI && PROGRAM != kGI25 && sc.shadowPackets != 0u && ownWalks;
    if (asPacket && sc.shadowPackets == 2u) {   // per wavefront: are these 64 rays one bundle?
      auto first = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
      const float rx = first(ray.o.x), ry = first(ray.o.y), rz = first(ray.o.z);
      const float ex = ray.o.x - rx, ey = ray.o.y - ry, ez = ray.o.z - rz;
      asPacket = __builtin_amdgcn_ballot_w64(ex * ex + ey * ey + ez * ez > sc.shadowSpread * (pl.t * pl.t)) == 0ull;
    }
    if (asPacket) {
      int* const row = st.lds - __lane_id();
      const int ign = useIgnore ? ignore : -1;
      const unsigned long long all = __builtin_amdgcn_ballot_w64(true), bx = __builtin_amdgcn_ballot_w64(ix < 0.0f),
                               by = __builtin_amdgcn_ballot_w64(iy < 0.0f), bz = __builtin_amdgcn_ballot_w64(iz < 0.0f);
      if ((bx == 0ull || bx == all) && (by == 0ull || by == all) && (bz == 0ull || bz == all)) {
        switch ((bx != 0ull ? 1 : 0) | (by != 0ull ? 2 : 0) | (bz != 0ull ? 4 : 0)) {
          case 0: packet_walk<PROGRAM, 0, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          case 1: packet_walk<PROGRAM, 1, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          case 2: packet_walk<PROGRAM, 2, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          case 3: packet_walk<PROGRAM, 3, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          case 4: packet_walk<PROGRAM, 4, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          case 5: packet_walk<PROGRAM, 5, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          case 6: packet_walk<PROGRAM, 6, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
          default: packet_walk<PROGRAM, 7, true>(sc, ray, ix, iy, iz, ign, pl, row); return;
        }
      }
      packet_walk<PROGRAM, -1, true>(sc, ray, ix, iy, iz, ign, pl, row);   // (the sign-generic form of the walk)
      return;
    }
    if constexpr (!STATS && !LDSSCENE) {
      if (ownWalks) {   // (the scene has a tree of the backend's own: lt_hip_set_scene)
        traverse_own_lane<PROGRAM, ANYHIT>(sc, ray, ix, iy, iz, useIgnore ? ignore : -1, pl, st.lds);
        return;
      }
    }
  }
  // In the reference's order over the caller's tree: the counting kernels and the LDS-resident small scenes with the LDS stack,
  // the others -- rays with a non-finite component, e.g. the image-centre column / row where a direction component is exactly
  // 0, and scenes without a tree of the backend's own -- with the stack in private memory.
  if constexpr (STATS || LDSSCENE) {
    if (__all(finite)) traverse_nodes_impl<PROGRAM, Stack<DEEP>, STATS, true, ANYHIT, LDSSCENE>(sc, ray, ix, iy, iz, useIgnore, ignore, pl, st, c);
    else traverse_nodes_impl<PROGRAM, Stack<DEEP>, STATS, false, ANYHIT, LDSSCENE>(sc, ray, ix, iy, iz, useIgnore, ignore, pl, st, c);
  } else {
    ScratchStack ss;
    traverse_nodes_impl<PROGRAM, ScratchStack, false, false, ANYHIT, false>(sc, ray, ix, iy, iz, useIgnore, ignore, pl, ss, c);
  }
}

// ---------------------------------------------------------------- shading helpers
__device__ __forceinline__ V3 barycentrics(float u, float v) {   // (float3)(1.0 - u - v, u, v), first in double
  return V3{(float)((1.0 - (double)u) - (double)v), u, v};
}
// float3 A*w.x + B*w.y + C*w.z evaluated (A*wx + B*wy) + C*wz (acc.cl:247)
template <int M = 0>
__device__ __forceinline__ V3 bary3(const float* a, const float* b, const float* cc, V3 w) {
  if (Math<M>::kShipped)   // fma(C, w.z, fma(A, w.x, B * w.y))
    return V3{__builtin_fmaf(cc[0], w.z, __builtin_fmaf(a[0], w.x, b[0] * w.y)), __builtin_fmaf(cc[1], w.z, __builtin_fmaf(a[1], w.x, b[1] * w.y)),
              __builtin_fmaf(cc[2], w.z, __builtin_fmaf(a[2], w.x, b[2] * w.y))};
  return V3{(a[0] * w.x + b[0] * w.y) + cc[0] * w.z, (a[1] * w.x + b[1] * w.y) + cc[1] * w.z,
            (a[2] * w.x + b[2] * w.y) + cc[2] * w.z};
}
__device__ __forceinline__ const float* prim_ptr(const SceneDev& sc, int p) { return sc.prims + 19 * (size_t)p; }
__device__ __forceinline__ int prim_material(const float* pr) { return __float_as_int(pr[18]); }

// `primitiveIndex == lights.primitives[x]` for x < count, no hitType check (acc.cl:233-237)
__device__ __forceinline__ bool is_light(const Lights* L, int prim) {
  bool hit = false;
  const uint32_t n = L->count;
  for (uint32_t x = 0; x < n; x++) hit = hit || ((uint32_t)prim == L->primitives[x & 63u]);
  return hit;
}
// int(random * count) indexes the light list; index == count reads its zero-initialised tail (Q7)
__device__ __forceinline__ const float* light_prim(const SceneDev& sc, float rnd) {
  const int idx = (int)(rnd * (float)sc.lights->count);
  const uint32_t p = (idx >= 0 && idx < 64) ? sc.lights->primitives[idx] : 0u;
  return prim_ptr(sc, (int)p);
}

// Light sample + shadow ray: acc.cl:239-279, basic_lighting.cl:234-274, gi.cl:267-297 and :323-349.
// normal_w is 0 in accumulator/basic_lighting, 1 in GI (extractDataFromBarycentrics returns w = 1, gi.cl:238).
// light_sample: everything up to the shadow ray -- the point on the light, the hit's position and normal, the ray towards the
// light, its length minus the shadow epsilon (tmax) and n.l; direct_light: that, and the ray's walk.
template <class CFG>
__device__ __forceinline__ void light_sample(const SceneDev& sc, const float* pr, float u, float v, float fx, float fy, float seedIndex, float seedU,
                                             float seedV, float normal_w, V4& position, V4& normal, V4& toLight, float& tmax, float& ndotl) {
  // Order of evaluation (not of arithmetic: every value is the reference's): the light sample first -- three random() calls,
  // i.e. three double-precision sin / fmod evaluations that want every register -- and only then the hit primitive's own
  // positions and normals, fenced so that the compiler does not issue those loads ahead of the randoms and carry 19 floats
  // through them in scratch memory.
  const float* lp = light_prim(sc, random_<CFG::kDevLibm>(fx, fy, seedIndex));
  float uvx = random_<CFG::kDevLibm>(fx, fy, seedU);
  float uvy = random_<CFG::kDevLibm>(fx, fy, seedV);
  if (uvx + uvy > 1.0f) {
    uvx = 1.0f - uvx;
    uvy = 1.0f - uvy;
  }
  const V3 lb = barycentrics(uvx, uvy);
  const V3 l3 = bary3<CFG::kDevLibm>(lp + 0, lp + 3, lp + 6, lb);
  const V4 lightPosition = mk4(l3.x, l3.y, l3.z, 1.0f);
  asm volatile("" ::: "memory");

  const V3 b = barycentrics(u, v);
  const V3 p3 = bary3<CFG::kDevLibm>(pr + 0, pr + 3, pr + 6, b);
  position = mk4(p3.x, p3.y, p3.z, 1.0f);
  const V3 n3 = bary3<CFG::kDevLibm>(pr + 9, pr + 12, pr + 15, b);
  normal = mk4(n3.x, n3.y, n3.z, normal_w);

  toLight = normalize4<CFG::kDevLibm>(sub4(lightPosition, position));
  tmax = (float)((double)distance4<CFG::kDevLibm>(position, lightPosition) - 0.01);
  // (before the walk, not after it as acc.cl:277 has it: the value does not depend on the walk, and the nine normal floats need
  // not stay in registers -- i.e. in scratch memory, 48 bytes per pixel and sample -- while it runs)
  ndotl = dot4(toLight, normal);
}

template <int PROGRAM, class CFG>
__device__ inline bool direct_light(const SceneDev& sc, const float* pr, int primIndex, float u, float v, float fx,
                                    float fy, float seedIndex, float seedU, float seedV, float normal_w, V4& position,
                                    V4& normal, float& ndotl, Stack<CFG::kDeep>& st, Counters& c) {
  V4 toLight;
  float tmax;
  light_sample<CFG>(sc, pr, u, v, fx, fy, seedIndex, seedU, seedV, normal_w, position, normal, toLight, tmax, ndotl);
  Hit spl{0, 0, tmax, 0.0f, 0.0f};
  const Ray shadowRay{position, toLight};
  asm volatile("" : "+v"(ndotl));   // (pins it here: left alone, the compiler sinks the interpolation of the normal behind the walk)
  if (CFG::kStats) c.shadow++;
  traverse<PROGRAM, CFG::kDeep, CFG::kStats, true, CFG::kLdsScene>(sc, shadowRay, true, primIndex, spl, st, c);
  asm volatile("" ::: "memory");    // (what the caller reads of the primitive afterwards -- its material -- is fetched afterwards)
  return spl.hitType == 0;
}

// basic.cl:65-71
template <int M = 0>
__device__ inline V4 refract_(V4 I, V4 N, float firstIOR, float secondIOR) {
  const float n = secondIOR == 1.0f ? firstIOR : Math<M>::fdiv(firstIOR, secondIOR);   // (x / 1.0f folds to x at compile time in the reference)
  const float cosI = -dot4(N, I);
  const float sinT2 = (float)((double)(n * n) * (1.0 - (double)(cosI * cosI)));
  const float cosT = (float)sqrt(1.0 - (double)sinT2);
  if (Math<M>::kShipped) {   // n * I + (n * cosI - cosT) * N, contracted: fma(n, I, fma(n, cosI, -cosT) * N)
    const float k = __builtin_fmaf(n, cosI, -cosT);
    return mk4(__builtin_fmaf(n, I.x, k * N.x), __builtin_fmaf(n, I.y, k * N.y), __builtin_fmaf(n, I.z, k * N.z), __builtin_fmaf(n, I.w, k * N.w));
  }
  return add4(scale4(n, I), scale4(n * cosI - cosT, N));
}

// basic.cl:225-277
template <class CFG>
__device__ inline void trace_ray_through_lens(const SceneDev& sc, Hit& pl, Ray& ray, Stack<CFG::kDeep>& st, Counters& c) {
  const float* pr = prim_ptr(sc, pl.prim);
  const Material* m = sc.mats + prim_material(pr);
  V3 b = barycentrics(pl.u, pl.v);
  V3 p3 = bary3<CFG::kDevLibm>(pr + 0, pr + 3, pr + 6, b);
  V4 position = mk4(p3.x, p3.y, p3.z, 1.0f);
  V3 n3 = bary3<CFG::kDevLibm>(pr + 9, pr + 12, pr + 15, b);
  V4 normal = mk4(n3.x, n3.y, n3.z, 0.0f);
  V4 tdir = refract_<CFG::kDevLibm>(ray.d, normal, 1.0f, m->ior);

  Hit pl2{0, 0, kFltMax, 0.0f, 0.0f};
  const Ray ray2{position, tdir};
  traverse<kBasic, CFG::kDeep, CFG::kStats, false>(sc, ray2, true, pl.prim, pl2, st, c);

  pr = prim_ptr(sc, pl2.prim);
  m = sc.mats + prim_material(pr);
  b = barycentrics(pl2.u, pl2.v);
  p3 = bary3<CFG::kDevLibm>(pr + 0, pr + 3, pr + 6, b);
  position = mk4(p3.x, p3.y, p3.z, 1.0f);
  n3 = bary3<CFG::kDevLibm>(pr + 9, pr + 12, pr + 15, b);
  normal = mk4(n3.x, n3.y, n3.z, 0.0f);
  tdir = refract_<CFG::kDevLibm>(tdir, neg4(normal), m->ior, 1.0f);

  pl = Hit{0, 0, kFltMax, 0.0f, 0.0f};
  ray.o = position;
  ray.d = tdir;
  traverse<kBasic, CFG::kDeep, CFG::kStats, false>(sc, ray, true, pl2.prim, pl, st, c);
}

// basic.cl:279-307
template <class CFG>
__device__ inline V3 shade_basic(const SceneDev& sc, Ray ray, Stack<CFG::kDeep>& st, Counters& c) {
  V3 out{0.0f, 0.0f, 0.0f};
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  traverse_camera<kBasic, CFG::kDeep, CFG::kStats>(sc, ray, pl, st, c);
  if (pl.hitType == 1) {
    const float* pr = prim_ptr(sc, pl.prim);
    const Material* m = sc.mats + prim_material(pr);
    if ((double)m->dissolve < 1.0) {
      trace_ray_through_lens<CFG>(sc, pl, ray, st, c);
      if (pl.hitType == 1) {
        pr = prim_ptr(sc, pl.prim);
        m = sc.mats + prim_material(pr);
      }
    }
    out = V3{m->diffuse[0], m->diffuse[1], m->diffuse[2]};
  }
  return out;
}

// examples/custom_kernel/resources/kernels/custom_opencl.cl:226-246 -- basic.cl without the lens code; the colour is the
// hit's barycentrics (u, v, 1.0 - u - v), the last one computed in double
template <class CFG>
__device__ inline V3 shade_custom(const SceneDev& sc, const Ray& ray, Stack<CFG::kDeep>& st, Counters& c) {
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  traverse_camera<kCustom, CFG::kDeep, CFG::kStats>(sc, ray, pl, st, c);
  if (pl.hitType == 1) return V3{pl.u, pl.v, (float)((1.0 - (double)pl.u) - (double)pl.v)};
  return V3{0.0f, 0.0f, 0.0f};
}

// acc.cl:219-282 / basic_lighting.cl:220-277
// (qslot / qpixel / qframe: where a queued shadow ray goes and which stored colour it decides -- SceneDev::shadowPackets == 3;
// queued: it went)
template <int PROGRAM, class CFG>
__device__ inline V3 shade_lighting(const SceneDev& sc, const Ray& cameraRay, float fx, float fy, uint32_t s,
                                    Stack<CFG::kDeep>& st, Counters& c, uint32_t qslot, uint32_t qpixel, uint32_t qframe, bool& queued) {
  V3 out{0.0f, 0.0f, 0.0f};
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  traverse_camera<PROGRAM, CFG::kDeep, CFG::kStats>(sc, cameraRay, pl, st, c);
  if (PROGRAM == kAccumulator || PROGRAM == kAccumulatorQueue) {
    if (is_light(sc.lights, pl.prim)) return V3{1.0f, 1.0f, 1.0f};
  }
  if (pl.hitType == 1) {
    const float* pr = prim_ptr(sc, pl.prim);
    V4 position, normal;
    float ndotl;
    if constexpr (PROGRAM == kAccumulatorQueue) {
      // the shadow ray goes into the queue (lt_trace_kernel walks it); the colour is the unoccluded sample's until
      // lt_shadow_resolve_kernel has looked at the ray's fate
      V4 toLight;
      float tmax;
      light_sample<CFG>(sc, pr, pl.u, pl.v, fx, fy, (float)s, (float)(s + 1u), (float)(s + 2u), 0.0f, position, normal, toLight, tmax, ndotl);
      sc.shadowQueue[qslot] = make_float4(position.x, position.y, position.z, tmax);
      sc.shadowQueue[(size_t)sc.shadowCap + qslot] = make_float4(toLight.x, toLight.y, toLight.z, toLight.w);
      ((uint4*)sc.shadowQueue)[2 * (size_t)sc.shadowCap + qslot] = make_uint4(qpixel, (uint32_t)pl.prim, qframe, 0u);
      queued = true;
      const Material* m = sc.mats + prim_material(pr);
      out = V3{m->diffuse[0] * ndotl, m->diffuse[1] * ndotl, m->diffuse[2] * ndotl};
    } else {
      if (direct_light<PROGRAM, CFG>(sc, pr, pl.prim, pl.u, pl.v, fx, fy, (float)s, (float)(s + 1u), (float)(s + 2u), 0.0f, position, normal, ndotl, st, c)) {
        const Material* m = sc.mats + prim_material(pr);
        out = V3{m->diffuse[0] * ndotl, m->diffuse[1] * ndotl, m->diffuse[2] * ndotl};
      }
    }
  }
  return out;
}

// gi.cl:68-74
template <int DEVLIBM>
__device__ inline V4 uniform_sample_hemisphere(float uvx, float uvy) {
  const float z = uvx;
  const float r = Math<DEVLIBM>::sqrt_user(__builtin_fmaxf(0.0f, Math<DEVLIBM>::kShipped ? __builtin_fmaf(-z, z, 1.0f) : 1.0f - z * z));
  const float phi = (float)(2.0 * M_PI * (double)uvy);
  return mk4(r * Math<DEVLIBM>::cos(phi), z, r * Math<DEVLIBM>::sin(phi), 0.0f);
}
// gi.cl:76-81
template <int DEVLIBM>
__device__ inline V4 align_hemisphere(V4 h, V4 up) {
  const V4 right = normalize4<DEVLIBM>(cross4(up, mk4(0.0072f, 1.0f, 0.0034f, 0.0f)));
  const V4 forward = cross4(right, up);
  if (Math<DEVLIBM>::kShipped)   // fma(h.z, forward, fma(h.x, right, h.y * up))
    return mk4(__builtin_fmaf(h.z, forward.x, __builtin_fmaf(h.x, right.x, h.y * up.x)), __builtin_fmaf(h.z, forward.y, __builtin_fmaf(h.x, right.y, h.y * up.y)),
               __builtin_fmaf(h.z, forward.z, __builtin_fmaf(h.x, right.z, h.y * up.z)), __builtin_fmaf(h.z, forward.w, __builtin_fmaf(h.x, right.w, h.y * up.w)));
  return add4(add4(scale4(h.x, right), scale4(h.y, up)), scale4(h.z, forward));
}

// gi.cl:241-375
template <class CFG>
__device__ inline V3 shade_gi(const SceneDev& sc, const Ray& cameraRay, float fx, float fy, uint32_t s, int maxDepth,
                              Stack<CFG::kDeep>& st, Counters& c) {
  V3 direct{0.0f, 0.0f, 0.0f}, indirect{0.0f, 0.0f, 0.0f};
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  traverse_camera<kGI, CFG::kDeep, CFG::kStats>(sc, cameraRay, pl, st, c);
  if (is_light(sc.lights, pl.prim)) {
    direct = V3{1.0f, 1.0f, 1.0f};
  } else if (pl.hitType == 1) {
    const float* pr = prim_ptr(sc, pl.prim);
    const Material* m = sc.mats + prim_material(pr);
    V4 position, normal;
    float ndotl;
    if (direct_light<kGI, CFG>(sc, pr, pl.prim, pl.u, pl.v, fx, fy, (float)s, (float)(s + 1u), (float)(s + 2u), 1.0f,
                                       position, normal, ndotl, st, c)) {
      direct = V3{m->diffuse[0] * ndotl, m->diffuse[1] * ndotl, m->diffuse[2] * ndotl};
    }
    V4 hemi = uniform_sample_hemisphere<CFG::kDevLibm>(random_<CFG::kDevLibm>(fx, fy, (float)(s + 3u)), random_<CFG::kDevLibm>(fx, fy, (float)(s + 4u)));
    Ray ext{position, align_hemisphere<CFG::kDevLibm>(hemi, normal)};
    V4 previousNormal = normal;
    int previousPrimitive = pl.prim;
    bool rayActive = true;
    for (int d = 0; d < maxDepth && rayActive; d++) {
      Hit epl{0, 0, kFltMax, 0.0f, 0.0f};
      traverse<kGI, CFG::kDeep, CFG::kStats, false>(sc, ext, true, previousPrimitive, epl, st, c);
      const float w = (float)(1.0 / (double)(d + 1));
      const uint32_t sd = s + (uint32_t)d;
      if (is_light(sc.lights, epl.prim)) {
        // hit a light: add and keep looping with the SAME ray (gi.cl:319-321)
        const float k = dot4(previousNormal, ext.d);
        indirect.x = Math<CFG::kDevLibm>::mad(w * 1.0f, k, indirect.x);
        indirect.y = Math<CFG::kDevLibm>::mad(w * 1.0f, k, indirect.y);
        indirect.z = Math<CFG::kDevLibm>::mad(w * 1.0f, k, indirect.z);
      } else if (epl.hitType == 1) {
        const float* epr = prim_ptr(sc, epl.prim);
        const Material* em = sc.mats + prim_material(epr);
        V4 epos, enorm;
        float endotl;
        if (direct_light<kGI, CFG>(sc, epr, epl.prim, epl.u, epl.v, fx, fy, (float)(sd + 5u), (float)(sd + 6u),
                                           (float)(sd + 7u), 1.0f, epos, enorm, endotl, st, c)) {
          indirect.x = Math<CFG::kDevLibm>::mad(w * em->diffuse[0], endotl, indirect.x);
          indirect.y = Math<CFG::kDevLibm>::mad(w * em->diffuse[1], endotl, indirect.y);
          indirect.z = Math<CFG::kDevLibm>::mad(w * em->diffuse[2], endotl, indirect.z);
          hemi = uniform_sample_hemisphere<CFG::kDevLibm>(random_<CFG::kDevLibm>(fx, fy, (float)(sd + 8u)), random_<CFG::kDevLibm>(fx, fy, (float)(sd + 9u)));
          ext.o = epos;
          ext.d = align_hemisphere<CFG::kDevLibm>(hemi, enorm);
          previousNormal = enorm;
          previousPrimitive = epl.prim;
        } else {
          rayActive = false;
        }
      } else {
        rayActive = false;
      }
    }
  }
  return V3{direct.x + indirect.x, direct.y + indirect.y, direct.z + indirect.z};
}

// Per-frame uniforms.  cos/sin(yaw) are evaluated once on the host as (float)cos((double)yaw) (portable math); the
// reference evaluates cos(camera->yaw) per work-item (acc.cl:309-310) -- the device-libm flavour does the same.
struct FrameParams {
  float camx, camy, camz;
  // aperaturePosition (0, 0, 5) of acc.cl:306, passed as run-time values on purpose: with a literal 0 the
  // compiler folds `0.0f - film` into a negate source modifier on the consumers (1/d), which turns the +0
  // direction component of the image-centre column/row into -0 and flips dirIsNeg there.
  float apx, apy, apz;
  float cosYaw, sinYaw, yaw;
  uint32_t frameCount;
  uint32_t width, height, depth;
  int32_t clampOutput;      // linearKernel of the lighting programs clamps, tileKernel does not (acc.cl:316-318/:356-358)
  int32_t giMaxDepth;
  int32_t pixelCounters;    // diagnostic: write the pixel's work counters instead of its colour
  int32_t accumulateN;      // < 0: overwrite; >= 0: running mean with n = accumulateN (accumulator.frag:10-20)
  // tile sharding (lenstrace_hip.h): tiles tile_first + k*tile_stride, k < tilesInCall
  uint32_t tileW, tileH, tilesX, tileFirst, tileStride, tilesInCall;
  uint32_t blocksPerTileX, blocksPerTile;   // 8x8-pixel wavefront squares per image tile
  uint32_t totalSquares;                    // tilesInCall * blocksPerTile
  uint32_t ldsRows;                         // 256-byte LDS rows of each wavefront in this launch (Stack::rows)
  uint32_t persistent;                      // != 0: waves pull squares from per-XCD queues instead of one square per workgroup
  // several samples in one launch (persistent mode): work item = (frame f, square), frame f uses frameCount + f and stores
  // its colours, un-accumulated, at out + f * frameStride; lt_running_mean_kernel folds them in frame order afterwards
  uint32_t fusedFrames;                     // >= 1
  uint32_t squareMajor;                     // != 0: a queue hands out the frames of a square one after the other (else a frame's squares)
  unsigned long long frameStride;           // floats between the sample images of a fused launch
  // persistent mode: the order in which an XCD's share of squares is handed out (position -> square index), or null for
  // the natural order.  The host puts the squares that cannot take the fast path (a pixel with an exactly-zero direction
  // component: image-centre row / column) at the head of each share -- orderHead[xcd] of them -- and a launch hands out the
  // head squares of ALL its frames before anything else, so that its longest wavefronts start first.
  const uint32_t* order;
  uint32_t orderHead[8];
};

// Camera ray of pixel (x,y): acc.cl:304-312.
template <int DEVLIBM>
__device__ __forceinline__ Ray camera_ray(const FrameParams& fp, int x, int y, float& fx, float& fy) {
  const V4 cameraPosition = mk4(fp.camx, fp.camy, fp.camz, 1.0f);
  const V4 film = mk4(Math<DEVLIBM>::fdiv((float)x, (float)fp.width) - 0.5f, Math<DEVLIBM>::fdiv((float)y, (float)fp.height) - 0.5f, 0.0f, 1.0f);
  const V4 aperture = mk4(fp.apx, fp.apy, fp.apz, 1.0f);
  Ray ray{add4(cameraPosition, film), sub4(aperture, film)};
  const float cy = DEVLIBM ? ::cosf(fp.yaw) : fp.cosYaw, sy = DEVLIBM ? ::sinf(fp.yaw) : fp.sinYaw;
  const float newX = Math<DEVLIBM>::mad(cy, ray.d.x, sy * ray.d.z);
  const float newZ = Math<DEVLIBM>::mad(-sy, ray.d.x, cy * ray.d.z);
  ray.d.x = newX;
  ray.d.z = newZ;
  fx = film.x;
  fy = film.y;
  return ray;
}

#ifdef LT_USER_PROGRAM
// The shade step of a run-time compiled user program (the counterpart of `shade` in the reference's kernel files, e.g.
// custom_opencl.cl:226-246): defined by the user's source file, called once per pixel with the camera ray of
// linearKernel (acc.cl:304-312), the film position and the camera's frameCount.  Everything in this header is at its
// disposal: traverse_camera<>, traverse<>, intersect helpers, random_, Math<>, the scene buffers.
template <class CFG>
__device__ V3 user_shade(const SceneDev& sc, const Ray& cameraRay, float filmX, float filmY, uint32_t frameCount,
                         Stack<CFG::kDeep>& st, Counters& c);
#endif

// The body of linearKernel / tileKernel for one pixel, all five programs
// (acc.cl:314-318, basic.cl:338-342, basic_lighting.cl:309-321, resources gi :408-420).
template <int PROGRAM, class CFG>
__device__ inline V3 shade_pixel(const SceneDev& sc, const FrameParams& fp, uint32_t frameCount, int x, int y, Stack<CFG::kDeep>& st,
                                 Counters& c, uint32_t qslot, uint32_t qpixel, uint32_t qframe, bool& queued) {
  float fx, fy;
  const Ray ray = camera_ray<CFG::kDevLibm>(fp, x, y, fx, fy);
  V3 color;
  if (PROGRAM == kBasic) {
    color = shade_basic<CFG>(sc, ray, st, c);
  } else if (PROGRAM == kCustom) {
    color = shade_custom<CFG>(sc, ray, st, c);
#ifdef LT_USER_PROGRAM
  } else if (PROGRAM == kUser) {
    color = user_shade<CFG>(sc, ray, fx, fy, frameCount, st, c);
#endif
  } else if (PROGRAM == kAccumulator || PROGRAM == kAccumulatorQueue) {
    color = shade_lighting<PROGRAM, CFG>(sc, ray, fx, fy, frameCount, st, c, qslot, qpixel, qframe, queued);
  } else if (PROGRAM == kGI) {
    color = shade_gi<CFG>(sc, ray, fx, fy, frameCount, fp.giMaxDepth, st, c);
  } else {
    const uint32_t base = frameCount * 32u;
    for (int k = 0; k < 25; k++) {
      const V3 cn = (PROGRAM == kBasicLighting)
                        ? shade_lighting<kBasicLighting, CFG>(sc, ray, fx, fy, base + (uint32_t)k, st, c, 0u, 0u, 0u, queued)
                        : shade_gi<CFG>(sc, ray, fx, fy, base + (uint32_t)k, fp.giMaxDepth, st, c);
      if (k == 0) {
        color = cn;
      } else {
        const float a = Math<CFG::kDevLibm>::div25((float)(25 - k));
        color = V3{Math<CFG::kDevLibm>::mad(1.0f - a, color.x, a * cn.x), Math<CFG::kDevLibm>::mad(1.0f - a, color.y, a * cn.y),
                   Math<CFG::kDevLibm>::mad(1.0f - a, color.z, a * cn.z)};
      }
    }
  }
  if (PROGRAM != kBasic && PROGRAM != kCustom && PROGRAM != kUser && fp.clampOutput) color = V3{Math<CFG::kDevLibm>::clamp01(color.x), Math<CFG::kDevLibm>::clamp01(color.y), Math<CFG::kDevLibm>::clamp01(color.z)};
  return color;
}

}  // namespace lt
