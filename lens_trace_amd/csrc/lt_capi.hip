// lt_capi.hip -- liblenstrace-hip.so: the C ABI of include/lenstrace_hip.h over hand-written gfx950 kernels.
// Host side of what the reference does in RendererOpenCL::render() (src/opencl/renderer_opencl.cpp:56-153).
#include "lt_kernel.hpp"
#include "lt_retree.hpp"
#include "lt_own16.hpp"
#include "lt_prep.hpp"

#include "../../include/lenstrace_hip.h"

#include <dlfcn.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

// Triangle re-tiling at upload: 76-byte Primitive -> 48-byte (A, B-A, C-A, 0 0 0).
__global__ void lt_retile_kernel(const float* __restrict__ prims, float4* __restrict__ tris, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = prims + 19 * (size_t)i;
  const float ax = p[0], ay = p[1], az = p[2];
  tris[3 * (size_t)i + 0] = make_float4(ax, ay, az, p[3] - ax);
  tris[3 * (size_t)i + 1] = make_float4(p[4] - ay, p[5] - az, p[6] - ax, p[7] - ay);
  tris[3 * (size_t)i + 2] = make_float4(p[8] - az, 0.0f, 0.0f, 0.0f);
}

// The records of the packet walks (lt_walk_asm.hpp, packet_walk_cpp), one 64-byte slot per node of the backend's own tree:
//   interior node i: [left child's box, its reference, -][right child's box, its reference, -], the boxes pushed outwards by
//                    2^-21 of each bound and one float more (the walks' conservative test needs lo' <= lo - 6 * 2^-24 |lo|),
//                    a reference = the child's index, with bit 31 set when the child is a leaf;
//   leaf i:          [A, e1 = B - A, e2 = C - A of its triangle (lt_retile_kernel's arithmetic)][the leaf's own box, bit for
//                    bit][the primitive offset]: what the reference's leaf test needs, in one scalar load.
__device__ __forceinline__ float lt_outwards(float b, bool up) { return lt_own16::outwards(b, up); }
__global__ void lt_own_pair_kernel(const float4* __restrict__ nodes, const float* __restrict__ prims, float4* __restrict__ pairs, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 a = nodes[2 * (size_t)i], b = nodes[2 * (size_t)i + 1];
  if ((__float_as_uint(b.w) & 0xffffu) != 0u) {
    const float* p = prims + 19 * (size_t)__float_as_uint(b.z);
    const float ax = p[0], ay = p[1], az = p[2];
    pairs[4 * (size_t)i + 0] = make_float4(ax, ay, az, p[3] - ax);
    pairs[4 * (size_t)i + 1] = make_float4(p[4] - ay, p[5] - az, p[6] - ax, p[7] - ay);
    pairs[4 * (size_t)i + 2] = make_float4(p[8] - az, a.x, a.y, a.z);
    pairs[4 * (size_t)i + 3] = make_float4(a.w, b.x, b.y, b.z);
    return;
  }
  const uint32_t child[2] = {i + 1u, __float_as_uint(b.z)};
  for (int k = 0; k < 2; k++) {
    const float4 ca = nodes[2 * (size_t)child[k]], cb = nodes[2 * (size_t)child[k] + 1];
    const bool leaf = (__float_as_uint(cb.w) & 0xffffu) != 0u;
    pairs[4 * (size_t)i + 2 * k] = make_float4(lt_outwards(ca.x, false), lt_outwards(ca.y, false), lt_outwards(ca.z, false), lt_outwards(ca.w, true));
    pairs[4 * (size_t)i + 2 * k + 1] = make_float4(lt_outwards(cb.x, true), lt_outwards(cb.y, true), __uint_as_float(child[k] | (leaf ? 0x80000000u : 0u)), 0.0f);
  }
}

// The per-lane walks' records (SceneDev::wide, traverse_own_lane), 64 bytes each, one array:
//   [0, groups)                          the 4-wide groups of lt_retree::collapse_wide: four 16-byte child slots -- the child's box
//                                        on a 16-bit grid over the scene's bounds, rounded outwards (lt_own16.hpp: in real
//                                        arithmetic O + ql S <= lo - 8u|lo| and O + qh S >= hi + 8u|hi|), and its link;
//   [groups, groups + n_prims]           leaf records by primitive offset: the triangle re-tiled (A, B - A, C - A: lt_retile_kernel's
//                                        arithmetic), the leaf's own box bit for bit, the offset -- what the reference's leaf test
//                                        needs -- and one more behind them whose box is NaN (the target of empty slots).
// *bad is set when a bound falls off the grid (the host sizes the grid from the root's box with room to spare; the scene then
// simply gets no hierarchy of the backend's own).
struct Own16Frame { float O[3], S[3]; };
__global__ void lt_wide_kernel(const float4* __restrict__ nodes, const uint32_t* __restrict__ children, const uint32_t* __restrict__ groupOf,
                               uint4* __restrict__ wide, uint32_t groups, uint32_t n_prims, Own16Frame fr, uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // slot
  if (i >= 4u * groups) return;
  const uint32_t child = children[i];
  lt_own16::Rec r = lt_own16::empty_slot(groups, n_prims);
  if (child != 0xffffffffu) {
    const float4 a = nodes[2 * (size_t)child], b = nodes[2 * (size_t)child + 1];
    const float lo[3] = {a.x, a.y, a.z}, hi[3] = {a.w, b.x, b.y};
    const bool leaf = (__float_as_uint(b.w) & 0xffffu) != 0u;
    const uint32_t link = leaf ? (0x80000000u | (groups + __float_as_uint(b.z))) : groupOf[child];
    if (!lt_own16::slot_record(lo, hi, link, fr.O, fr.S, r)) atomicOr(bad, 1u);
  }
  wide[i] = make_uint4(r.x, r.y, r.z, r.w);
}
__global__ void lt_wide_leaf_kernel(const float4* __restrict__ nodes, const float* __restrict__ prims, float4* __restrict__ wide, uint32_t n,
                                    uint32_t groups, uint32_t n_prims) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // node of the own tree; thread n writes the record behind the last primitive's
  if (i > n) return;
  if (i == n) {
    const float q = __uint_as_float(0x7fc00000u);
    float4* r = wide + 4 * ((size_t)groups + n_prims);
    r[0] = r[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    r[2] = make_float4(0.0f, q, q, q);
    r[3] = make_float4(q, q, q, __uint_as_float(n_prims));
    return;
  }
  const float4 a = nodes[2 * (size_t)i], b = nodes[2 * (size_t)i + 1];
  if ((__float_as_uint(b.w) & 0xffffu) == 0u) return;
  const uint32_t prim = __float_as_uint(b.z);
  const float* p = prims + 19 * (size_t)prim;
  const float ax = p[0], ay = p[1], az = p[2];
  float4* r = wide + 4 * ((size_t)groups + prim);
  r[0] = make_float4(ax, ay, az, p[3] - ax);
  r[1] = make_float4(p[4] - ay, p[5] - az, p[6] - ax, p[7] - ay);
  r[2] = make_float4(p[8] - az, a.x, a.y, a.z);
  r[3] = make_float4(a.w, b.x, b.y, b.z);
}

// Gathered per-rank tile stacks -> row-major image (root side of the one gather per frame).
__global__ void lt_untile_kernel(const float* __restrict__ gathered, uint64_t floatsPerRank, uint32_t nRanks, uint32_t W,
                                 uint32_t H, uint32_t depth, uint32_t tileW, uint32_t tileH, uint32_t tilesX,
                                 float* __restrict__ image) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)W * H) return;
  const uint32_t x = (uint32_t)(i % W), y = (uint32_t)(i / W);
  const uint32_t tx = x / tileW, ty = y / tileH, tile = ty * tilesX + tx;
  const uint32_t rank = tile % nRanks, k = tile / nRanks;
  const float* src = gathered + (uint64_t)rank * floatsPerRank +
                     (((uint64_t)k * tileH + (y - ty * tileH)) * tileW + (x - tx * tileW)) * depth;
  float* dst = image + i * depth;
  for (uint32_t ch = 0; ch < depth; ch++) dst[ch] = src[ch];
}

// Folds the n sample images of a fused launch (samples + f * stride, f < n, in frame order) into the running mean held in
// `out`: accumulator.frag:10-20, `(c + acc*n) / (n+1)` with n = base + f, the same expression in the same order as the
// read-modify-write of render_square, so the result is bit for bit what n single-sample launches leave behind.  Pixels of
// edge tiles that lie outside the image are not touched (render_square never writes them).
__global__ void lt_running_mean_kernel(const float* __restrict__ samples, uint32_t n, uint64_t stride, float* __restrict__ out,
                                       uint64_t first, uint64_t floats, int32_t base, FrameParams fp, int checkPadding) {
  const uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (this launch folds the floats [first, floats))
  if (i >= floats) return;
  if (i % fp.depth >= 3u) return;   // render_square / the GI resolve stage write channels 0..2 only; the others are the caller's
  if (checkPadding) {
    const uint64_t pix = i / fp.depth, perTile = (uint64_t)fp.tileW * fp.tileH;
    const uint32_t k = (uint32_t)(pix / perTile), rem = (uint32_t)(pix % perTile);
    const uint32_t ly = rem / fp.tileW, lx = rem % fp.tileW, tile = fp.tileFirst + k * fp.tileStride;
    if ((tile % fp.tilesX) * fp.tileW + lx >= fp.width || (tile / fp.tilesX) * fp.tileH + ly >= fp.height) return;
  }
  float acc = base > 0 ? out[i] : 0.0f;
  for (uint32_t f = 0; f < n; f++) {
    const float c = samples[(uint64_t)f * stride + i];
    const int32_t N = base + (int32_t)f;
    if (N <= 0) {
      acc = c;
    } else {
      const float nf = (float)N, n1 = (float)(N + 1);
      acc = (c + (acc * nf)) / n1;
    }
  }
  out[i] = acc;
}

// The walks' slab tests compare against the smallest positive float (`tExit >= max(tEnter, 0x00000001)`: "tExit > 0" in one
// instruction) and their conservative tests carry a 2^-140 margin: both need float32 denormals to be KEPT, which is hipcc's
// default for gfx950 and what this library and its run-time compiled user programs are built with.  A build with
// -fgpu-flush-denormals-to-zero would flush the constant to 0 and accept leaves the reference rejects; lt_hip_create runs this
// once and refuses such a build instead.
__global__ void lt_denormal_probe_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
  const float tiny = __uint_as_float(in[0]);                       // 0x00000001
  out[0] = __float_as_uint(__builtin_fmaxf(tiny, in[1] ? 1.0f : 0.0f));   // max(denormal, 0): the denormal, unless it was flushed
  out[1] = __float_as_uint(tiny + tiny);                           // 0x00000002
}

// ---------------------------------------------------------------------------------- context
struct SceneHash {
  uint64_t buf[4] = {0, 0, 0, 0};   // nodes, primitives, materials, lights
  bool operator==(const SceneHash& o) const { return memcmp(buf, o.buf, sizeof(buf)) == 0; }
};

// Device buffers of scenes gone by, kept for the next scene of the same shape: an animation hands over buffers of the same sizes
// frame after frame, and a hipFree / hipMalloc pair per buffer (a dozen of them, each a device-wide wait) was a tenth of
// lt_hip_set_scene.  Exact sizes only; at most kSpareCap bytes lie idle (LT_SCENE_POOL_BYTES, 0: every buffer goes straight
// back to the runtime).
struct ScenePool {
  std::multimap<size_t, void*> spare;
  std::map<void*, size_t> sizes;   // of every buffer this pool handed out
  size_t spareBytes = 0, cap = (size_t)8 << 30;
  hipError_t get(void** p, size_t bytes) {
    if (bytes == 0) bytes = 4;
    auto it = spare.find(bytes);
    if (it != spare.end()) {
      *p = it->second;
      spareBytes -= bytes;
      spare.erase(it);
      return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && !spare.empty()) {   // (memory is short: the idle buffers go first)
      clear();
      e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) sizes[*p] = bytes;
    return e;
  }
  void put(void* p) {
    if (!p) return;
    auto it = sizes.find(p);
    if (it == sizes.end()) { (void)hipFree(p); return; }
    const size_t bytes = it->second;
    if (bytes > cap) { sizes.erase(it); (void)hipFree(p); return; }
    if (spareBytes + bytes > cap) clear();
    spare.emplace(bytes, p);
    spareBytes += bytes;
  }
  void clear() {
    for (auto& kv : spare) { sizes.erase(kv.second); (void)hipFree(kv.second); }
    spare.clear();
    spareBytes = 0;
  }
  static hipError_t get_cb(void* self, void** p, size_t bytes) { return ((ScenePool*)self)->get(p, bytes); }
  static void put_cb(void* self, void* p) { ((ScenePool*)self)->put(p); }
};

struct lt_hip_context {
  int device = -1;
  ScenePool pool;
  std::string err;
  hipStream_t stream = nullptr;      // own stream for lt_hip_render
  void *d_nodes = nullptr, *d_tris = nullptr, *d_prims = nullptr, *d_mats = nullptr, *d_lights = nullptr;
  void *d_nodes2 = nullptr, *d_pairs2 = nullptr;   // the backend's own tree over the scene's leaves (lt_retree.hpp), or null
  void* d_rank8 = nullptr;                         // with it: the reference's leaf order per direction-sign octant (SceneDev::rank8)
  void* d_wide = nullptr;                          // ... and the per-lane walks' 4-wide groups and leaf records (SceneDev::wide; 64 bytes in front: the grid)
  uint32_t n_wide = 0;                             // groups
  int wide_height = 0;
  int height2 = 0;                                 // its height
  uint32_t n_nodes2 = 0;
  float retree_ms = 0.0f;                          // host time of its build
  uint32_t lds_ref_bytes = 0;                      // LDS of a wave whose per-lane stack follows the caller's tree (set per render call)
  uint32_t n_nodes = 0, n_prims = 0, n_mats = 0;
  int bvh_height = 0;
  bool has_scene = false;
  bool device_prepared = false;      // the resident scene's derived structures were made by lt_prep.hip (not by lt_retree.hpp on the host)
  uint64_t verdict_sizes[4] = {0, 0, 0, 0};   // sizes of the scene the shadow-walk verdicts below were timed on
  bool verdict_sizes_valid = false;
  bool speculate_next = true;        // lt_hip_render_scene: the last scene handed over with a frame was the resident one (render while hashing)
  SceneHash scene_hash{};                                    // content hashes (one per buffer) ...
  uint64_t scene_sizes[4] = {0, 0, 0, 0};                   // ... and sizes of the resident scene (lt_hip_set_scene)
  uint32_t scene_uploads = 0, scene_reused = 0;
  float* d_out = nullptr;            // staging output for lt_hip_render
  uint64_t d_out_bytes = 0;
  void* h_out = nullptr;             // ... and its pinned host twin: the read-back lands here at the link's rate, piece by piece
  uint64_t h_out_bytes = 0;          //     (a caller's pageable buffer would be read back through the runtime's small bounce buffers)
  hipEvent_t out_ev[8] = {};         // one event per piece
  hipEvent_t fold_ev[8] = {};        // ... and one behind the fold of each piece (launch_running_mean), when the call's last fold is cut into them
  uint64_t fold_piece_bytes = 0;     //     (set by lt_hip_render around its render: bytes per piece, 0 = one fold launch)
  bool fold_pieced = false;
  hipStream_t copy_stream = nullptr; //     the pieces travel on a stream of their own, each behind its fold
  unsigned long long* d_stats = nullptr;
  uint32_t* d_queues = nullptr;      // persistent mode: 8 per-XCD work counters per launch of a call
  uint32_t queue_frames = 0;
  int shadow_mode[6] = {-1, -1, -1, -1, -1, -1};   // per built-in program: shadow rays as any-hit packets (1) or per lane (0); -1 = not timed yet
  hipEvent_t cal_ev[12] = {};
  std::map<std::vector<uint32_t>, int> shadow_modes;   // (program, W, H, tile geometry) -> the walk timed faster for it on the resident scene
  void* d_shadowq = nullptr;         // accumulator's queued shadow rays (shadow mode 3): origin+tmax, direction, (pixel, primitive, frame), occluded: 52 bytes per slot
  uint64_t shadowq_slots = 0;
  uint32_t* d_shadowCtl = nullptr;   // ... the trace launch's eight work counters (kQueueStride apart) and, behind them, the queue's length
  float* d_samples = nullptr;        // un-accumulated sample images of a fused multi-sample launch
  uint64_t d_samples_bytes = 0;
  uint32_t* d_order = nullptr;       // persistent mode: hand-out order of the squares (slow-path squares first), cached
  uint64_t order_capacity = 0;
  std::vector<uint32_t> order_key;   // what d_order was built for
  uint32_t order_head[8] = {0};      // slow-path squares at the head of each XCD's share
  int cu_count = 256;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> mean_events;   // pairs around the running-mean kernels of the last call
  uint32_t mean_pairs = 0;
  hipStream_t last_stream = nullptr;
  bool pending = false, pending_stats = false;
  lt_hip_stats last{};
  // wavefront GI pipeline: path queues, per-pixel direct / indirect / blend, control block (queue lengths, work counters)
  void* d_gi[17] = {nullptr};
  uint64_t gi_pixels = 0;
  uint32_t* d_giCtl = nullptr;
  // user programs (hipRTC), cached by path like the reference's programMap
  struct UserProgram { hipModule_t module; hipFunction_t lds, ldsStrict, ldsPortable; };   // one kernel per math flavour
  std::vector<UserProgram> user_programs;
  std::map<std::string, int> user_program_ids;
};

static thread_local std::string g_create_error;

#define LT_HIP_CHECK(ctx, call)                                                                     \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
      return LT_ERR_HIP;                                                                            \
    }                                                                                               \
  } while (0)

static int fail(lt_hip_context* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

extern "C" int lt_hip_abi_version(void) { return LT_HIP_ABI_VERSION; }

extern "C" const char* lt_hip_last_error(const lt_hip_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int lt_hip_create(int device_index, lt_hip_context** out_ctx) {
  if (!out_ctx) return fail(nullptr, LT_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(nullptr, LT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device_index < 0 || device_index >= n) return fail(nullptr, LT_ERR_INVALID_ARGUMENT, "device_index out of range");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device_index)) != hipSuccess)
    return fail(nullptr, LT_ERR_NO_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, LT_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
  lt_hip_context* ctx = new lt_hip_context();
  ctx->device = device_index;
  ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("LT_SCENE_POOL_BYTES")) ctx->pool.cap = (size_t)strtoull(e, nullptr, 10);
  auto bail = [&](const char* what, hipError_t er) {
    std::string m = std::string(what) + ": " + hipGetErrorString(er);
    delete ctx;
    return fail(nullptr, LT_ERR_HIP, m);
  };
  if ((e = hipSetDevice(device_index)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc((void**)&ctx->d_stats, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  {   // float32 denormals must be kept (lt_denormal_probe_kernel)
    const uint32_t in[2] = {1u, 0u};
    uint32_t out[2] = {0u, 0u};
    uint32_t* d = (uint32_t*)ctx->d_stats;
    if ((e = hipMemcpy(d, in, sizeof(in), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    hipLaunchKernelGGL(lt_denormal_probe_kernel, dim3(1), dim3(1), 0, ctx->stream, (const uint32_t*)d, d + 2);
    if ((e = hipGetLastError()) != hipSuccess) return bail("kernel launch (is this library built for gfx950?)", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
    if ((e = hipMemcpy(out, d + 2, sizeof(out), hipMemcpyDeviceToHost)) != hipSuccess) return bail("hipMemcpy", e);
    if (out[0] != 1u || out[1] != 2u) {
      (void)hipFree(ctx->d_stats);
      (void)hipEventDestroy(ctx->ev0);
      (void)hipEventDestroy(ctx->ev1);
      (void)hipStreamDestroy(ctx->stream);
      delete ctx;
      return fail(nullptr, LT_ERR_NO_DEVICE, "this build flushes float32 denormals to zero (-fgpu-flush-denormals-to-zero?): the walks' slab tests need them kept");
    }
  }
  *out_ctx = ctx;
  return LT_OK;
}

static void free_scene(lt_hip_context* ctx) {
  for (void** p : {&ctx->d_nodes, &ctx->d_tris, &ctx->d_prims, &ctx->d_mats, &ctx->d_lights, &ctx->d_nodes2, &ctx->d_pairs2, &ctx->d_rank8, &ctx->d_wide}) {
    if (*p) ctx->pool.put(*p);
    *p = nullptr;
  }
  ctx->has_scene = false;
  ctx->device_prepared = false;
}

extern "C" int lt_hip_destroy(lt_hip_context* ctx) {
  if (!ctx) return LT_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  free_scene(ctx);
  ctx->pool.clear();
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->h_out) (void)hipHostFree(ctx->h_out);
  for (hipEvent_t e : ctx->out_ev) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->fold_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->d_stats) (void)hipFree(ctx->d_stats);
  if (ctx->d_queues) (void)hipFree(ctx->d_queues);
  if (ctx->d_samples) (void)hipFree(ctx->d_samples);
  if (ctx->d_shadowq) (void)hipFree(ctx->d_shadowq);
  if (ctx->d_shadowCtl) (void)hipFree(ctx->d_shadowCtl);
  if (ctx->d_order) (void)hipFree(ctx->d_order);
  for (auto& up : ctx->user_programs) (void)hipModuleUnload(up.module);
  for (void*& b : ctx->d_gi) if (b) (void)hipFree(b);
  if (ctx->d_giCtl) (void)hipFree(ctx->d_giCtl);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (hipEvent_t e : ctx->mean_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->cal_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return LT_OK;
}

extern "C" int lt_hip_program_from_path(const char* path, int* out_program) {
  if (!path || !out_program) return LT_ERR_INVALID_ARGUMENT;
  std::string p(path);
  size_t slash = p.find_last_of('/');
  std::string base = slash == std::string::npos ? p : p.substr(slash + 1);
  size_t dot = base.find_last_of('.');
  if (dot != std::string::npos) base = base.substr(0, dot);
  if (base == "basic") *out_program = LT_PROGRAM_BASIC;
  else if (base == "basic_lighting") *out_program = LT_PROGRAM_BASIC_LIGHTING;
  else if (base == "accumulator") *out_program = LT_PROGRAM_ACCUMULATOR;
  else if (base == "custom_opencl") *out_program = LT_PROGRAM_CUSTOM_OPENCL;
  else if (base == "global_illumination25") *out_program = LT_PROGRAM_GLOBAL_ILLUMINATION_25;
  else if (base == "global_illumination") {
    const bool shipped25 = p.find("resources/kernels/opencl/") != std::string::npos && p.find("examples/") == std::string::npos;
    *out_program = shipped25 ? LT_PROGRAM_GLOBAL_ILLUMINATION_25 : LT_PROGRAM_GLOBAL_ILLUMINATION;
  } else return LT_ERR_UNKNOWN_PROGRAM;
  return LT_OK;
}

// ---------------------------------------------------------------------------------- user programs (hipRTC)
// hipRTC is loaded on first use (dlopen), so the library has no link-time dependency on it.
namespace {
struct Hiprtc {
  void* lib = nullptr;
  int (*createProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compileProgram)(void*, int, const char**) = nullptr;
  int (*getProgramLogSize)(void*, size_t*) = nullptr;
  int (*getProgramLog)(void*, char*) = nullptr;
  int (*getCodeSize)(void*, size_t*) = nullptr;
  int (*getCode)(void*, char*) = nullptr;
  int (*destroyProgram)(void**) = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
    for (const char* name : {"libhiprtc.so.7", "libhiprtc.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { err = "hipRTC library not found (libhiprtc.so)"; return false; }
#define LT_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)); if (!field) { err = std::string("hipRTC symbol missing: ") + sym; return false; }
    LT_SYM(createProgram, "hiprtcCreateProgram") LT_SYM(compileProgram, "hiprtcCompileProgram")
    LT_SYM(getProgramLogSize, "hiprtcGetProgramLogSize") LT_SYM(getProgramLog, "hiprtcGetProgramLog")
    LT_SYM(getCodeSize, "hiprtcGetCodeSize") LT_SYM(getCode, "hiprtcGetCode") LT_SYM(destroyProgram, "hiprtcDestroyProgram")
#undef LT_SYM
    return true;
  }
};
Hiprtc g_hiprtc;

// directory of the device headers: <this library>/../csrc, or $LT_CSRC_DIR
std::string csrc_dir() {
  if (const char* e = getenv("LT_CSRC_DIR")) return e;
  Dl_info info;
  if (dladdr((const void*)&lt_hip_abi_version, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t slash = p.find_last_of('/');
    return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/../csrc";
  }
  return "lens_trace_amd/csrc";
}
}  // namespace

static int compile_user_program(lt_hip_context* ctx, const std::string& path, int* out_program) {
  std::ifstream in(path);
  if (!in) return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, "cannot read user program " + path);
  std::stringstream user;
  user << in.rdbuf();
  std::string err;
  if (!g_hiprtc.load(err)) return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, err);
  const std::string src = "#define LT_USER_PROGRAM 1\n#include \"lt_kernel.hpp\"\n#line 1 \"" + path + "\"\n" + user.str() +
      "\n#define LT_USER_KERNEL(name, DEEP, DEVLIBM) extern \"C\" __global__ __launch_bounds__(64) void name(SceneDev sc, FrameParams fp, float* out, unsigned long long* stats, uint32_t* queues) { render_kernel_body<kUser, Config<DEEP, false, DEVLIBM>>(sc, fp, out, stats, queues); }\n"
      "LT_USER_KERNEL(lt_user_kernel_lds, false, 2)\n"
      "LT_USER_KERNEL(lt_user_kernel_lds_strict, false, 1)\n"
      "LT_USER_KERNEL(lt_user_kernel_lds_portable, false, 0)\n";
  void* prog = nullptr;
  if (g_hiprtc.createProgram(&prog, src.c_str(), "lt_user_program.hip", 0, nullptr, nullptr) != 0)
    return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, "hiprtcCreateProgram failed");
  const std::string inc = "-I" + csrc_dir();
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", inc.c_str()};
  const int rc = g_hiprtc.compileProgram(prog, 5, opts);
  size_t n = 0;
  std::string log;
  if (g_hiprtc.getProgramLogSize(prog, &n) == 0 && n > 1) {
    log.resize(n);
    g_hiprtc.getProgramLog(prog, &log[0]);
  }
  if (rc != 0) {
    g_hiprtc.destroyProgram(&prog);
    return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, "user program " + path + " failed to compile:\n" + log);
  }
  std::vector<char> code;
  if (g_hiprtc.getCodeSize(prog, &n) != 0 || n == 0) { g_hiprtc.destroyProgram(&prog); return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, "hiprtcGetCodeSize failed"); }
  code.resize(n);
  g_hiprtc.getCode(prog, code.data());
  g_hiprtc.destroyProgram(&prog);
  lt_hip_context::UserProgram up{};
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  LT_HIP_CHECK(ctx, hipModuleLoadData(&up.module, code.data()));
  LT_HIP_CHECK(ctx, hipModuleGetFunction(&up.lds, up.module, "lt_user_kernel_lds"));
  LT_HIP_CHECK(ctx, hipModuleGetFunction(&up.ldsStrict, up.module, "lt_user_kernel_lds_strict"));
  LT_HIP_CHECK(ctx, hipModuleGetFunction(&up.ldsPortable, up.module, "lt_user_kernel_lds_portable"));
  ctx->user_programs.push_back(up);
  *out_program = LT_PROGRAM_USER_BASE + (int)ctx->user_programs.size() - 1;
  ctx->user_program_ids[path] = *out_program;
  return LT_OK;
}

extern "C" int lt_hip_resolve_program(lt_hip_context* ctx, const char* path, int* out_program) {
  if (!ctx || !path || !out_program) return LT_ERR_INVALID_ARGUMENT;
  if (lt_hip_program_from_path(path, out_program) == LT_OK) return LT_OK;
  const std::string p(path);
  if (p.size() < 5 || p.compare(p.size() - 4, 4, ".hip") != 0)
    return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, "no built-in program for " + p + " (user programs are .hip files)");
  auto it = ctx->user_program_ids.find(p);
  if (it != ctx->user_program_ids.end()) { *out_program = it->second; return LT_OK; }
  return compile_user_program(ctx, p, out_program);
}

// Host-side validation: nothing with an out-of-range index or a cycle may reach a kernel.
// Returns the BVH height (max number of interior ancestors of a node) or -1 with msg set.
static int validate_scene(const uint8_t* nodes, uint32_t n_nodes, const uint8_t* prims, uint32_t n_prims, uint32_t n_mats,
                          const uint8_t* lights, std::string& msg) {
  struct N { float lo[3], hi[3]; int32_t off; uint16_t cnt; uint8_t axis, pad; };
  static_assert(sizeof(N) == 32, "LinearBVHNode is 32 bytes");
  const N* nd = reinterpret_cast<const N*>(nodes);
  for (uint32_t i = 0; i < n_nodes; i++) {
    if (nd[i].cnt > 0) {
      if (nd[i].off < 0 || (uint32_t)nd[i].off >= n_prims) { msg = "leaf primitivesOffset out of range"; return -1; }
    } else {
      // pre-order layout: left child = i+1, right child = secondChildOffset > i+1; forward-only => no cycles
      if (i + 1 >= n_nodes || nd[i].off <= (int32_t)i + 1 || (uint32_t)nd[i].off >= n_nodes) { msg = "interior node children out of range"; return -1; }
      if (nd[i].axis > 2) { msg = "split axis out of range"; return -1; }
    }
  }
  for (uint32_t i = 0; i < n_prims; i++) {
    int32_t m;
    memcpy(&m, prims + 76 * (size_t)i + 72, 4);
    if (m < 0 || (uint32_t)m >= n_mats) { msg = "materialIndex out of range"; return -1; }
  }
  uint32_t lc;
  memcpy(&lc, lights, 4);
  if (lc > 64) { msg = "more than 64 emissive triangles"; return -1; }
  for (uint32_t i = 0; i < lc; i++) {
    uint32_t p;
    memcpy(&p, lights + 4 + 4 * i, 4);
    if (p >= n_prims) { msg = "light primitive out of range"; return -1; }
  }
  // height by forward propagation (children always have larger indices than their parent)
  std::vector<int> depth(n_nodes, -1);
  depth[0] = 0;
  int height = 0;
  for (uint32_t i = 0; i < n_nodes; i++) {
    if (depth[i] < 0) continue;   // unreachable node: harmless
    if (depth[i] > height) height = depth[i];
    if (nd[i].cnt == 0) {   // (max: a malformed buffer may share a child between parents; the LDS stack is sized by this height)
      depth[i + 1] = std::max(depth[i + 1], depth[i] + 1);
      depth[nd[i].off] = std::max(depth[nd[i].off], depth[i] + 1);
    }
  }
  return height;
}

// Host threads for the memory-bound host passes of lt_hip_set_scene (content hash): what the process may run on
// (sched_getaffinity: a container's CPU share, not the machine's core count), at most 16 unless LT_HOST_THREADS says otherwise
// (140 MB on the GPU box's host: 2.5 ms on 2 threads, 0.85 on 8, 0.70 on 12-16, 1.1 on 32 -- starting a thread costs what it
// hashes in 30 microseconds; tests/tools/hash_threads.py).
static int host_threads() {
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(16, CPU_COUNT(&set));
  if (const char* e = getenv("LT_HOST_THREADS")) n = atoi(e);
  return std::max(1, std::min(64, n));
}

// 64-bit content hash of a host buffer: four independent multiply-rotate lanes over 32-byte blocks, per 4 MiB piece; the pieces'
// hashes are folded in order, so the value does not depend on how many threads computed them (one pass over host memory at the
// memory system's rate: the four scene buffers of the 1 M-triangle scene, 140 MB, take a few milliseconds on the GPU box's
// host; bench.py's e2e figures measure it).  Used to tell "the same scene again" from "a buffer was edited in place" without an upload.
static uint64_t hash_piece(const uint8_t* b, uint64_t n, uint64_t seed) {
  constexpr uint64_t K = 0x9E3779B97F4A7C15ull;
  uint64_t h[4] = {seed ^ K, seed + 0xC2B2AE3D27D4EB4Full, seed ^ 0x165667B19E3779F9ull, seed + 0x27D4EB2F165667C5ull};
  auto block = [&](const uint8_t* q) {
    uint64_t w[4];
    memcpy(w, q, 32);
    for (int k = 0; k < 4; k++) {
      h[k] = (h[k] ^ w[k]) * K;
      h[k] = (h[k] << 29) | (h[k] >> 35);
    }
  };
  uint64_t i = 0;
  for (; i + 32 <= n; i += 32) block(b + i);
  if (i < n) {
    uint8_t tail[32] = {0};
    memcpy(tail, b + i, (size_t)(n - i));
    block(tail);
  }
  uint64_t r = n * K;
  for (int k = 0; k < 4; k++) {
    r = (r ^ h[k]) * K;
    r ^= r >> 32;
  }
  return r;
}

// The four scene buffers as one list of pieces, hashed by up to host_threads() threads (fewer when threads cannot be had: the
// pieces left are hashed by this one).
static SceneHash hash_scene(const void* const bufs[4], const uint64_t sizes[4]) {
  constexpr uint64_t kPiece = 4ull << 20;
  struct Piece { const uint8_t* p; uint64_t n; int of; };
  std::vector<Piece> pieces;
  for (int k = 0; k < 4; k++)
    for (uint64_t off = 0; off < sizes[k]; off += kPiece) pieces.push_back({(const uint8_t*)bufs[k] + off, std::min(kPiece, sizes[k] - off), k});
  std::vector<uint64_t> hashes(pieces.size());
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t i; (i = next.fetch_add(1)) < pieces.size();) hashes[i] = hash_piece(pieces[i].p, pieces[i].n, (uint64_t)i);
  };
  std::vector<std::thread> pool;
  const int threads = (int)std::min<size_t>((size_t)host_threads(), pieces.size());
  try {
    for (int t = 1; t < threads; t++) pool.emplace_back(work);
  } catch (...) {   // (no more threads to be had: whoever exists does the work)
  }
  work();
  for (std::thread& th : pool) th.join();
  SceneHash r;
  for (int k = 0; k < 4; k++) r.buf[k] = 0x243F6A8885A308D3ull + (uint64_t)k;
  for (size_t i = 0; i < pieces.size(); i++) {
    uint64_t& b = r.buf[pieces[i].of];
    b = (b ^ hashes[i]) * 0x9E3779B97F4A7C15ull;
    b ^= b >> 29;
  }
  return r;
}

extern "C" int lt_hip_own_hierarchy(const void* nodes, uint64_t node_bytes, int height_slack, void* out_nodes, uint64_t out_bytes,
                                    uint32_t* rank8, uint32_t n_prims) {
  if (!nodes || node_bytes == 0 || node_bytes % 32 || node_bytes > 0xffffffffull) return -1;
  const uint32_t n_nodes = (uint32_t)(node_bytes / 32);
  {   // the same structural checks lt_hip_set_scene makes before it looks at a buffer (indices in range, forward-only children)
    const lt_retree::Node* nd = (const lt_retree::Node*)nodes;
    for (uint32_t i = 0; i < n_nodes; i++) {
      if (nd[i].cnt != 0) { if (nd[i].off < 0 || (rank8 && (uint32_t)nd[i].off >= n_prims)) return -1; }
      else if (i + 1 >= n_nodes || nd[i].off <= (int32_t)i + 1 || (uint32_t)nd[i].off >= n_nodes || nd[i].axis > 2) return -1;
    }
  }
  std::vector<lt_retree::Node> own;
  const int h = height_slack < 0 ? lt_retree::copy(nodes, n_nodes, 30, own) : lt_retree::build(nodes, n_nodes, 30, height_slack, own);
  if (h < 0) return -1;
  if (out_nodes) {
    if (out_bytes < own.size() * sizeof(lt_retree::Node)) return -1;
    memcpy(out_nodes, own.data(), own.size() * sizeof(lt_retree::Node));
  }
  if (rank8) {
    std::vector<uint32_t> r;
    lt_retree::reference_order(nodes, n_nodes, n_prims, r);
    memcpy(rank8, r.data(), r.size() * sizeof(uint32_t));
  }
  return h;
}

extern "C" int lt_hip_own_wide(const void* own_nodes, uint64_t node_bytes, uint32_t n_prims, float* origin_step, void* out_slots, uint64_t out_bytes,
                               uint32_t* out_groups) {
  if (!own_nodes || !origin_step || !out_slots || !out_groups || node_bytes == 0 || node_bytes % 32 || node_bytes > 0xffffffffull) return -1;
  const uint32_t n = (uint32_t)(node_bytes / 32);
  const lt_retree::Node* src = (const lt_retree::Node*)own_nodes;
  if (n < 3 || src[0].cnt != 0) return -1;
  for (uint32_t i = 0; i < n; i++)   // (a pre-order tree: children behind their parent, in range)
    if (src[i].cnt == 0 && (i + 1 >= n || src[i].off <= (int32_t)i + 1 || (uint32_t)src[i].off >= n)) return -1;
  const std::vector<lt_retree::Node> own(src, src + n);
  std::vector<uint32_t> children, groupOf;
  const int height = lt_retree::collapse_wide(own, n_prims, children, groupOf);
  if (height < 0) return -1;
  const uint32_t groups = (uint32_t)(children.size() / 4);
  if (out_bytes < (uint64_t)groups * 64) return -1;
  for (int a = 0; a < 3; a++) lt_own16::frame(own[0].lo[a], own[0].hi[a], origin_step[a], origin_step[3 + a]);
  lt_own16::Rec* out = (lt_own16::Rec*)out_slots;
  bool ok = true;
  for (size_t i = 0; i < children.size(); i++) {
    const uint32_t c = children[i];
    out[i] = lt_own16::empty_slot(groups, n_prims);
    if (c == 0xffffffffu) continue;
    const uint32_t link = own[c].cnt != 0 ? (0x80000000u | (groups + (uint32_t)own[c].off)) : groupOf[c];
    ok = lt_own16::slot_record(own[c].lo, own[c].hi, link, origin_step, origin_step + 3, out[i]) && ok;
  }
  *out_groups = groups;
  return ok ? height : -1;
}

static int set_scene_impl(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims, uint64_t prim_bytes, const void* materials,
                          uint64_t material_bytes, const void* lights, uint64_t light_bytes, const SceneHash* known_hash);

extern "C" int lt_hip_set_scene(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims,
                                uint64_t prim_bytes, const void* materials, uint64_t material_bytes, const void* lights,
                                uint64_t light_bytes) {
  try {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = set_scene_impl(ctx, nodes, node_bytes, prims, prim_bytes, materials, material_bytes, lights, light_bytes, nullptr);
    if (getenv("LT_DEBUG_SCENE_TIMING"))
      fprintf(stderr, "[lt set_scene] %-28s %7.2f ms\n", "lt_hip_set_scene, all of it", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return rc;
  } catch (const std::exception& e) {   // (std::bad_alloc, std::system_error of a thread: nothing may cross the C ABI)
    return fail(ctx, LT_ERR_HIP, std::string("lt_hip_set_scene: ") + e.what());
  }
}

// The shadow-ray walk timed fastest for a scene (render_on_stream) is kept for a scene of the same shape -- the next pose of an
// animation, an edited material: it is a matter of speed, never of pixels, and timing it again costs five frames -- and forgotten
// when the sizes change (another scene).
static void new_scene_walk_verdicts(lt_hip_context* ctx, const uint64_t sizes[4]) {
  if (ctx->verdict_sizes_valid && memcmp(sizes, ctx->verdict_sizes, sizeof(ctx->verdict_sizes)) == 0 && !getenv("LT_RETIME_EVERY_SCENE")) return;
  for (int& m : ctx->shadow_mode) m = -1;
  ctx->shadow_modes.clear();
  memcpy(ctx->verdict_sizes, sizes, sizeof(ctx->verdict_sizes));
  ctx->verdict_sizes_valid = true;
}

// The device path of lt_hip_set_scene.  kDeviceDeclined: nothing of ctx was touched, the host path decides.
constexpr int kDeviceDeclined = -1000;
static int set_scene_on_device(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims, uint64_t prim_bytes,
                               const void* materials, uint64_t material_bytes, const void* lights, uint64_t light_bytes, const SceneHash& hash,
                               bool timing) {
  const uint32_t n_nodes = (uint32_t)(node_bytes / 32), n_prims = (uint32_t)(prim_bytes / 76), n_mats = (uint32_t)(material_bytes / 32);
  auto tmark = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[lt set_scene] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tmark).count());
    tmark = now;
  };
  {   // the light list is 260 bytes: checked here
    uint32_t lc;
    memcpy(&lc, lights, 4);
    if (lc > 64) return kDeviceDeclined;
    for (uint32_t i = 0; i < lc; i++) {
      uint32_t pi;
      memcpy(&pi, (const uint8_t*)lights + 4 + 4 * i, 4);
      if (pi >= n_prims) return kDeviceDeclined;
    }
  }
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  void *d_nodes = nullptr, *d_prims = nullptr;
  const lt_prep::Allocator al{&ctx->pool, &ScenePool::get_cb, &ScenePool::put_cb};
  struct Guard {   // (whatever is still set when this returns goes back)
    ScenePool& pool; const lt_prep::Allocator& al; void** a; void** b; lt_prep::Out* o;
    ~Guard() { pool.put(*a); pool.put(*b); lt_prep::release(*o, al); }
  };
  lt_prep::Out prep;
  Guard guard{ctx->pool, al, &d_nodes, &d_prims, &prep};
  LT_HIP_CHECK(ctx, ctx->pool.get(&d_nodes, node_bytes));
  LT_HIP_CHECK(ctx, ctx->pool.get(&d_prims, prim_bytes));
  LT_HIP_CHECK(ctx, hipMemcpy(d_nodes, nodes, node_bytes, hipMemcpyHostToDevice));
  lap("upload nodes");
  // The hierarchy is built from the nodes alone: the primitives (the larger buffer) travel meanwhile, sent by a thread of their
  // own (a copy from pageable memory keeps its caller until it is done) -- or, if that thread cannot be had, right here.
  std::thread primThread;
  hipError_t primError = hipSuccess;
  bool primSent = false;
  try {
    primThread = std::thread([&]() {
      primError = hipSetDevice(ctx->device);
      if (primError == hipSuccess) primError = hipMemcpy(d_prims, prims, prim_bytes, hipMemcpyHostToDevice);
    });
    primSent = true;
  } catch (...) {
  }
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{primThread};   // (every exit below waits for it)
  if (!primSent) LT_HIP_CHECK(ctx, hipMemcpy(d_prims, prims, prim_bytes, hipMemcpyHostToDevice));
  const char* re = getenv("LT_RETREE");
  const char* sl = getenv("LT_RETREE_SLACK");
  const auto t0 = std::chrono::steady_clock::now();
  // (height <= 30: the packet walks' stack, one VGPR, holds 2 * height + 2 entries at most; LT_RETREE=0: the caller's splits)
  LT_HIP_CHECK(ctx, lt_prep::run(d_nodes, n_nodes, nullptr, n_prims, n_mats, 30, sl ? atoi(sl) : 2, !(re && atoi(re) == 0), ctx->stream, al, prep));
  if (timing) fprintf(stderr, "[lt set_scene] device: checks + leaf order %.2f ms, own hierarchy %.2f ms (%d levels), 4-wide groups %.2f ms, flags %u\n",
                      prep.ms_check, prep.ms_build, prep.levels, prep.ms_wide, prep.flags);
  if (prep.flags != 0 || prep.bvh_height > kMaxStack) return kDeviceDeclined;
  lap("device preparation");
  if (primThread.joinable()) primThread.join();
  LT_HIP_CHECK(ctx, primError);
  bool primsOk = false;
  LT_HIP_CHECK(ctx, lt_prep::check_primitives(d_prims, n_prims, n_mats, ctx->stream, (uint32_t*)ctx->d_stats, primsOk));
  if (!primsOk) return kDeviceDeclined;
  lap("primitives arrived, checked");
  // from here on the scene is good: it replaces the resident one
  LT_HIP_CHECK(ctx, hipDeviceSynchronize());
  free_scene(ctx);
  lap("free the resident scene");
  ctx->d_nodes = d_nodes; d_nodes = nullptr;
  ctx->d_prims = d_prims; d_prims = nullptr;
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_tris, (size_t)n_prims * 48));
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_mats, material_bytes));
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_lights, light_bytes));
  LT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_mats, materials, material_bytes, hipMemcpyHostToDevice, ctx->stream));
  LT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_lights, lights, light_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(lt_retile_kernel, dim3((n_prims + 255) / 256), dim3(256), 0, ctx->stream, (const float*)ctx->d_prims, (float4*)ctx->d_tris, n_prims);
  LT_HIP_CHECK(ctx, hipGetLastError());
  ctx->height2 = 0;
  ctx->retree_ms = 0.0f;
  const uint32_t n2 = prep.n_own, groups = prep.groups;
  const int hw = prep.wide_height;
  bool ownOk = hw >= 0 && groups > 0 && 3 * hw + 4 <= kOwnRows + kOwnDeep && (uint64_t)groups + n_prims + 1 < 0x7fffffffull;
  if (ownOk) {
    ctx->d_nodes2 = prep.d_nodes2; prep.d_nodes2 = nullptr;
    ctx->d_rank8 = prep.d_rank8; prep.d_rank8 = nullptr;
    LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_pairs2, (size_t)n2 * 64));
    hipLaunchKernelGGL(lt_own_pair_kernel, dim3((n2 + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2, (const float*)ctx->d_prims,
                       (float4*)ctx->d_pairs2, n2);
    LT_HIP_CHECK(ctx, hipGetLastError());
    Own16Frame fr;
    for (int a = 0; a < 3; a++) lt_own16::frame(prep.root_lo[a], prep.root_hi[a], fr.O[a], fr.S[a]);
    const size_t records = (size_t)groups + n_prims + 1;
    LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_wide, records * 64 + 64));
    const float head[16] = {0, 0, 0, 0, 0, 0, 0, 0, fr.O[0], fr.O[1], fr.O[2], 0.0f, fr.S[0], fr.S[1], fr.S[2], 0.0f};
    LT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_wide, head, sizeof(head), hipMemcpyHostToDevice, ctx->stream));
    LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long), ctx->stream));
    uint4* wide = (uint4*)ctx->d_wide + 4;
    hipLaunchKernelGGL(lt_wide_kernel, dim3((4 * groups + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2,
                       (const uint32_t*)prep.d_children, (const uint32_t*)prep.d_groupOf, wide, groups, n_prims, fr, (uint32_t*)ctx->d_stats);
    LT_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(lt_wide_leaf_kernel, dim3((n2 + 1 + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2,
                       (const float*)ctx->d_prims, (float4*)wide, n2, groups, n_prims);
    LT_HIP_CHECK(ctx, hipGetLastError());
    uint32_t bad = 0;
    LT_HIP_CHECK(ctx, hipMemcpyAsync(&bad, ctx->d_stats, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
    lap("mallocs, launches");
    LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ownOk = bad == 0;
    lap("retile, pair and wide records");
  }
  if (!ownOk) {   // the scene then walks the caller's tree
    for (void** p : {&ctx->d_nodes2, &ctx->d_pairs2, &ctx->d_wide, &ctx->d_rank8}) { ctx->pool.put(*p); *p = nullptr; }
  } else {
    ctx->n_nodes2 = n2;
    ctx->height2 = prep.own_height;
    ctx->n_wide = groups;
    ctx->wide_height = hw;
  }
  lt_prep::release(prep, al);
  LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->retree_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  lap("release");
  ctx->n_nodes = n_nodes;
  ctx->n_prims = n_prims;
  ctx->n_mats = n_mats;
  ctx->bvh_height = prep.bvh_height;
  ctx->has_scene = true;
  ctx->scene_hash = hash;
  ctx->scene_sizes[0] = node_bytes; ctx->scene_sizes[1] = prim_bytes; ctx->scene_sizes[2] = material_bytes; ctx->scene_sizes[3] = light_bytes;
  ctx->scene_uploads++;
  ctx->device_prepared = true;
  new_scene_walk_verdicts(ctx, ctx->scene_sizes);
  return LT_OK;
}

static int set_scene_impl(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims, uint64_t prim_bytes, const void* materials,
                          uint64_t material_bytes, const void* lights, uint64_t light_bytes, const SceneHash* known_hash) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  if (!nodes || !prims || !materials || !lights) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "NULL scene buffer");
  if (node_bytes == 0 || node_bytes % 32 || prim_bytes == 0 || prim_bytes % 76 || material_bytes == 0 || material_bytes % 32 ||
      light_bytes != 260)
    return fail(ctx, LT_ERR_BAD_SCENE, "scene buffer sizes are not whole multiples of LinearBVHNode(32) / Primitive(76) / Material(32) / LightContainer(260)");
  if (node_bytes > 0xffffffffull || prim_bytes / 76 > 0x7fffffffull / 48) return fail(ctx, LT_ERR_BAD_SCENE, "scene too large for 32-bit byte offsets (4 GiB of nodes / 2 GiB of traversal triangles)");
  const uint32_t n_nodes = (uint32_t)(node_bytes / 32), n_prims = (uint32_t)(prim_bytes / 76), n_mats = (uint32_t)(material_bytes / 32);
  // The reference uploads all buffers on every render() (renderer_opencl.cpp:107-120); here the resident copy is kept when the
  // caller hands over the same content again: sizes and a hash of EVERY byte (so an in-place edit of any vertex, node or
  // material is honoured).  LT_SCENE_ALWAYS_UPLOAD=1 turns the shortcut off.
  const uint64_t sizes[4] = {node_bytes, prim_bytes, material_bytes, light_bytes};
  const void* const bufs[4] = {nodes, prims, materials, lights};
  const auto tHash = std::chrono::steady_clock::now();
  const SceneHash hash = known_hash ? *known_hash : hash_scene(bufs, sizes);
  if (getenv("LT_DEBUG_SCENE_TIMING"))
    fprintf(stderr, "[lt set_scene] %-28s %7.2f ms\n", known_hash ? "hash (known)" : "hash", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tHash).count());
  if (ctx->has_scene && memcmp(sizes, ctx->scene_sizes, sizeof(sizes)) == 0 && !getenv("LT_SCENE_ALWAYS_UPLOAD")) {
    if (hash == ctx->scene_hash) {
      ctx->scene_reused++;
      return LT_OK;
    }
    // An edit that left the nodes alone -- a material, a light, a primitive's normals or material index -- leaves the hierarchies
    // alone too: the buffers that changed are uploaded, and what is derived from the primitives (traversal triangles, the leaf
    // records of both walks) is made again by the kernels that made it: no host-side build, no second copy of the tree.
    if (hash.buf[0] == ctx->scene_hash.buf[0]) {
      const bool primsChanged = hash.buf[1] != ctx->scene_hash.buf[1];
      std::string why;
      if (primsChanged) {
        for (uint32_t i = 0; i < n_prims && why.empty(); i++) {
          int32_t m;
          memcpy(&m, (const uint8_t*)prims + 76 * (size_t)i + 72, 4);
          if (m < 0 || (uint32_t)m >= n_mats) why = "materialIndex out of range";
        }
      }
      uint32_t lc;
      memcpy(&lc, lights, 4);
      if (lc > 64) why = "more than 64 emissive triangles";
      for (uint32_t i = 0; i < lc && i < 64 && why.empty(); i++) {
        uint32_t pi;
        memcpy(&pi, (const uint8_t*)lights + 4 + 4 * i, 4);
        if (pi >= n_prims) why = "light primitive out of range";
      }
      if (!why.empty()) return fail(ctx, LT_ERR_BAD_SCENE, why);
      LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
      LT_HIP_CHECK(ctx, hipDeviceSynchronize());   // (nothing of an earlier call may still be reading what is about to change)
      if (primsChanged) {
        LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_prims, prims, prim_bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(lt_retile_kernel, dim3((n_prims + 255) / 256), dim3(256), 0, ctx->stream, (const float*)ctx->d_prims, (float4*)ctx->d_tris, n_prims);
        if (ctx->d_nodes2 && ctx->d_pairs2 && ctx->d_wide) {
          const uint32_t n2 = ctx->n_nodes2;
          hipLaunchKernelGGL(lt_own_pair_kernel, dim3((n2 + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2, (const float*)ctx->d_prims,
                             (float4*)ctx->d_pairs2, n2);
          hipLaunchKernelGGL(lt_wide_leaf_kernel, dim3((n2 + 1 + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2, (const float*)ctx->d_prims,
                             (float4*)((uint4*)ctx->d_wide + 4), n2, ctx->n_wide, n_prims);
        }
        LT_HIP_CHECK(ctx, hipGetLastError());
      }
      LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_mats, materials, material_bytes, hipMemcpyHostToDevice));
      LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_lights, lights, light_bytes, hipMemcpyHostToDevice));
      LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      ctx->scene_hash = hash;
      ctx->scene_uploads++;
      new_scene_walk_verdicts(ctx, sizes);
      return LT_OK;
    }
  }
  const bool timing = getenv("LT_DEBUG_SCENE_TIMING") != nullptr;
  auto tmark = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[lt set_scene] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tmark).count());
    tmark = now;
  };
  std::string msg;
  // Scene preparation on the device (lt_prep.hip): nodes and primitives go up first, kernels check them, make the leaf order
  // table, build the own hierarchy and collapse it; the host passes below are what remains for scenes that path declines
  // (a malformed buffer -- the host words the error --, a tree that is not in the scene builder's pre-order, boxes that do not
  // nest) and for small ones, where a host build costs less than the launches.  LT_DEVICE_BUILD=0 / 1: never / whenever possible.
  {
    const char* db = getenv("LT_DEVICE_BUILD");
    const bool device = db ? atoi(db) != 0 : n_nodes >= 8192u;
    if (device) {
      const int rc = set_scene_on_device(ctx, nodes, node_bytes, prims, prim_bytes, materials, material_bytes, lights, light_bytes, hash, timing);
      if (rc != kDeviceDeclined) return rc;
      lap("device preparation declined");
    }
  }
  const int height = validate_scene((const uint8_t*)nodes, n_nodes, (const uint8_t*)prims, n_prims, n_mats, (const uint8_t*)lights, msg);
  lap("validate");
  if (height < 0) return fail(ctx, LT_ERR_BAD_SCENE, msg);
  if (height > kMaxStack) return fail(ctx, LT_ERR_BAD_SCENE, "BVH deeper than the reference's 64-entry traversal stack");

  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  LT_HIP_CHECK(ctx, hipDeviceSynchronize());
  free_scene(ctx);
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_nodes, node_bytes));
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_prims, prim_bytes));
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_tris, (size_t)n_prims * 48));
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_mats, material_bytes));
  LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_lights, light_bytes));
  LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_nodes, nodes, node_bytes, hipMemcpyHostToDevice));
  LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_prims, prims, prim_bytes, hipMemcpyHostToDevice));
  LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_mats, materials, material_bytes, hipMemcpyHostToDevice));
  LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_lights, lights, light_bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(lt_retile_kernel, dim3((n_prims + 255) / 256), dim3(256), 0, ctx->stream, (const float*)ctx->d_prims,
                     (float4*)ctx->d_tris, n_prims);
  LT_HIP_CHECK(ctx, hipGetLastError());
  lap("free, malloc, upload scene");
  // The backend's own hierarchy over the same leaves (lt_retree.hpp says why the pixels cannot change), for every finite ray of
  // the non-counting kernels.  LT_RETREE=0 keeps the caller's splits (same structures, same walks).  A scene whose boxes do not
  // nest gets none: its rays walk the caller's tree one by one, in the reference's order.
  ctx->height2 = 0;
  ctx->retree_ms = 0.0f;
  {
    const char* re = getenv("LT_RETREE");
    std::vector<lt_retree::Node> own;
    const auto t0 = std::chrono::steady_clock::now();
    const char* sl = getenv("LT_RETREE_SLACK");
    // (height <= 30: the packet walks' stack, one VGPR, holds 2 * height + 2 entries at most; LT_RETREE=0: the caller's splits)
    // (the leaf order table depends on the caller's tree alone: it is made by a thread of its own beside the build -- or, if that
    // thread cannot be had, after it)
    std::vector<uint32_t> rank8;
    std::thread rankThread;
    bool rankStarted = false;
    std::atomic<bool> rankFailed{false};
    try {
      rankThread = std::thread([&]() {
        try { lt_retree::reference_order(nodes, n_nodes, n_prims, rank8); } catch (...) { rankFailed = true; }   // (nothing escapes a thread)
      });
      rankStarted = true;
    } catch (...) {
    }
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{rankThread};   // (every exit below waits for it)
    const int h2 = (re && atoi(re) == 0) ? lt_retree::copy(nodes, n_nodes, 30, own) : lt_retree::build(nodes, n_nodes, 30, sl ? atoi(sl) : 2, own);
    lap("own hierarchy (host build)");
    if (h2 >= 0) {
      ctx->retree_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
      const uint32_t n2 = (uint32_t)own.size();
      LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_nodes2, (size_t)n2 * 32));
      LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_pairs2, (size_t)n2 * 64));
      LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_nodes2, own.data(), (size_t)n2 * 32, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(lt_own_pair_kernel, dim3((n2 + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2,
                         (const float*)ctx->d_prims, (float4*)ctx->d_pairs2, n2);
      LT_HIP_CHECK(ctx, hipGetLastError());
      // ... and the per-lane walks' 4-wide groups and leaf records, made from the same upload of the tree
      lap("upload own tree, pair kernel");
      std::vector<uint32_t> children, groupOf;
      const int hw = lt_retree::collapse_wide(own, n_prims, children, groupOf);
      const uint32_t groups = (uint32_t)(children.size() / 4);
      lap("collapse into 4-wide groups");
      // (the walk's stack: at most three waiting entries per level of groups and the four of the last one)
      bool ownOk = hw >= 0 && 3 * hw + 4 <= kOwnRows + kOwnDeep && (uint64_t)groups + n_prims + 1 < 0x7fffffffull;
      if (ownOk) {
        Own16Frame fr;
        for (int a = 0; a < 3; a++) lt_own16::frame(own[0].lo[a], own[0].hi[a], fr.O[a], fr.S[a]);
        const size_t records = (size_t)groups + n_prims + 1;
        void *d_children = nullptr, *d_groupOf = nullptr;
        LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_wide, records * 64 + 64));   // (64 bytes in front: the grid, read by the walks themselves)
        LT_HIP_CHECK(ctx, ctx->pool.get(&d_children, children.size() * 4));
        LT_HIP_CHECK(ctx, ctx->pool.get(&d_groupOf, groupOf.size() * 4));
        const float head[16] = {0, 0, 0, 0, 0, 0, 0, 0, fr.O[0], fr.O[1], fr.O[2], 0.0f, fr.S[0], fr.S[1], fr.S[2], 0.0f};
        LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_wide, head, sizeof(head), hipMemcpyHostToDevice));
        LT_HIP_CHECK(ctx, hipMemcpy(d_children, children.data(), children.size() * 4, hipMemcpyHostToDevice));
        LT_HIP_CHECK(ctx, hipMemcpy(d_groupOf, groupOf.data(), groupOf.size() * 4, hipMemcpyHostToDevice));
        LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long), ctx->stream));
        uint4* wide = (uint4*)ctx->d_wide + 4;
        hipLaunchKernelGGL(lt_wide_kernel, dim3((4 * groups + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2,
                           (const uint32_t*)d_children, (const uint32_t*)d_groupOf, wide, groups, n_prims, fr, (uint32_t*)ctx->d_stats);
        LT_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(lt_wide_leaf_kernel, dim3((n2 + 1 + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->d_nodes2,
                           (const float*)ctx->d_prims, (float4*)wide, n2, groups, n_prims);
        LT_HIP_CHECK(ctx, hipGetLastError());
        uint32_t bad = 0;
        LT_HIP_CHECK(ctx, hipMemcpyAsync(&bad, ctx->d_stats, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
        LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->pool.put(d_children);
        ctx->pool.put(d_groupOf);
        ownOk = bad == 0;   // (a bound off the grid cannot happen for a grid sized from the root's box)
        lap("wide records (upload, kernels)");
      }
      if (!ownOk) {   // the scene then walks the caller's tree
        for (void** p : {&ctx->d_nodes2, &ctx->d_pairs2, &ctx->d_wide}) { ctx->pool.put(*p); *p = nullptr; }
      } else {   // (the own tree's 32-byte form stays resident: an edit of the primitives alone re-makes the leaf records from it)
        ctx->n_nodes2 = n2;
        ctx->height2 = h2;
        ctx->n_wide = groups;
        ctx->wide_height = hw;
        if (rankThread.joinable()) rankThread.join();
        if (!rankStarted || rankFailed) lt_retree::reference_order(nodes, n_nodes, n_prims, rank8);
        LT_HIP_CHECK(ctx, ctx->pool.get(&ctx->d_rank8, rank8.size() * sizeof(uint32_t)));
        LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_rank8, rank8.data(), rank8.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        lap("leaf order table (host, upload)");
      }
      ctx->retree_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->n_nodes = n_nodes;
  ctx->n_prims = n_prims;
  ctx->n_mats = n_mats;
  ctx->bvh_height = height;
  ctx->has_scene = true;
  ctx->scene_hash = hash;
  memcpy(ctx->scene_sizes, sizes, sizeof(sizes));
  ctx->scene_uploads++;
  new_scene_walk_verdicts(ctx, sizes);
  return LT_OK;
}

// ---------------------------------------------------------------------------------- render
struct TilePlan {
  uint32_t tileW, tileH, tilesX, tilesY, tileFirst, tileStride, tilesInCall, bptx, bpty;
  uint64_t floats;
};

static int plan_tiles(const lt_hip_render_desc* d, TilePlan& p, std::string& msg) {
  if (!d || d->struct_size != sizeof(lt_hip_render_desc)) { msg = "bad lt_hip_render_desc (struct_size)"; return LT_ERR_INVALID_ARGUMENT; }
  if (d->width == 0 || d->height == 0 || d->depth < 3) { msg = "image dimensions must be W>0, H>0, depth>=3"; return LT_ERR_INVALID_ARGUMENT; }
  if ((uint64_t)d->width * d->height > 0x7fffffffull) { msg = "image too large"; return LT_ERR_INVALID_ARGUMENT; }
  if (d->tile_w == 0) {
    p.tileW = d->width; p.tileH = d->height; p.tileFirst = 0; p.tileStride = 1;
  } else {
    if (d->tile_h == 0 || d->tile_stride == 0) { msg = "tile_h and tile_stride must be > 0 when tile_w > 0"; return LT_ERR_INVALID_ARGUMENT; }
    p.tileW = d->tile_w; p.tileH = d->tile_h; p.tileFirst = d->tile_first; p.tileStride = d->tile_stride;
  }
  p.tilesX = (d->width + p.tileW - 1) / p.tileW;
  p.tilesY = (d->height + p.tileH - 1) / p.tileH;
  const uint32_t total = p.tilesX * p.tilesY;
  p.tilesInCall = p.tileFirst < total ? (total - p.tileFirst + p.tileStride - 1) / p.tileStride : 0;
  p.bptx = (p.tileW + 7) / 8;
  p.bpty = (p.tileH + 7) / 8;
  p.floats = (uint64_t)p.tilesInCall * p.tileW * p.tileH * d->depth;
  return LT_OK;
}

extern "C" int lt_hip_output_floats(const lt_hip_render_desc* desc, uint64_t* out_floats) {
  TilePlan p;
  std::string msg;
  if (!out_floats) return LT_ERR_INVALID_ARGUMENT;
  int rc = plan_tiles(desc, p, msg);
  if (rc) return rc;
  *out_floats = p.floats;
  return LT_OK;
}

struct LaunchConfig { bool deep, stats; int devlibm; };   // devlibm: the math flavour, Math<0 / 1 / 2> (lt_device.hpp)

template <int PROGRAM>
static void launch_program(const LaunchConfig& k, dim3 grid, uint32_t lds, hipStream_t s, const SceneDev& sc, const FrameParams& fp,
                           float* out, unsigned long long* st, uint32_t* queues) {
#define LT_LAUNCH(D, S, M) hipLaunchKernelGGL((lt_render_kernel<PROGRAM, Config<D, S, M>>), grid, dim3(kBlock), lds, s, sc, fp, out, st, queues)
  // (DEEP -- LDS stack rows beyond kLdsStack spilled to scratch -- concerns the counting kernels only: the others keep no per-lane
  // stack in LDS, whatever the height of the caller's tree)
  if constexpr (PROGRAM == kAccumulatorQueue) {   // (never a counting launch)
    if (k.devlibm == 2) LT_LAUNCH(false, false, 2); else if (k.devlibm == 1) LT_LAUNCH(false, false, 1); else LT_LAUNCH(false, false, 0);
  } else if (k.devlibm == 2) {     // the default flavour: the reference kernels as RendererOpenCL builds them
    if (k.stats) { if (k.deep) LT_LAUNCH(true, true, 2); else LT_LAUNCH(false, true, 2); } else LT_LAUNCH(false, false, 2);
  } else if (k.devlibm == 1) {   // strict build of the reference kernels
    if (k.stats) { if (k.deep) LT_LAUNCH(true, true, 1); else LT_LAUNCH(false, true, 1); } else LT_LAUNCH(false, false, 1);
  } else {
    if (k.stats) { if (k.deep) LT_LAUNCH(true, true, 0); else LT_LAUNCH(false, true, 0); } else LT_LAUNCH(false, false, 0);
  }
#undef LT_LAUNCH
}

// 8 per-XCD square queues, queue lengths, bounce work counters, trace work counters (extension rays, shadow rays), hit-list lengths:
// every counter in a cache line of its own
constexpr uint32_t kGiCtlWords = (8 + 19 * (kMaxStack + 2)) * kQueueStride;   // (the trace launches' counters come in eights: one per eighth of their queue)

static int ensure_gi_buffers(lt_hip_context* ctx, uint64_t pixels) {
  if (!ctx->d_giCtl) LT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_giCtl, kGiCtlWords * sizeof(uint32_t)));
  if (ctx->gi_pixels >= pixels) return LT_OK;
  for (void*& b : ctx->d_gi) { if (b) LT_HIP_CHECK(ctx, hipFree(b)); b = nullptr; }
  ctx->gi_pixels = 0;
  for (int i = 0; i < 17; i++) LT_HIP_CHECK(ctx, hipMalloc(&ctx->d_gi[i], pixels * 16));
  ctx->gi_pixels = pixels;
  return LT_OK;
}

// fp.fusedFrames samples of a global-illumination program through one set of stage launches of the wavefront pipeline
// (lt_kernel.hpp): frames sample, sample + 1, ... of the single-sample program (blend25Out == nullptr: the resolve stage
// writes each frame's clamped colour to out + frame * fp.frameStride, or straight to the image when there is one frame), or
// samples k0 .. k0 + fusedFrames - 1 of ONE frame of the 25-sample variant (blend25Out = the image: the resolve stage
// writes raw colours to `out`, lt_gi_blend25_kernel blends them in order and finishes the pixel with accumulateN).
template <class CFG>
static int launch_gi_sample(lt_hip_context* ctx, hipStream_t s, const SceneDev& sc, const FrameParams& fp, float* out, uint32_t lds,
                            uint64_t pixels, uint32_t sample, uint32_t k0, float* blend25Out, int32_t accumulateN, uint32_t& launches) {
  // `pixels` = compact output pixels of ONE frame
  GiParams gp{};
  for (int k = 0; k < 2; k++) {
    gp.q[k].o = (float4*)ctx->d_gi[4 * k + 0]; gp.q[k].d = (float4*)ctx->d_gi[4 * k + 1];
    gp.q[k].n = (float4*)ctx->d_gi[4 * k + 2]; gp.q[k].m = (uint4*)ctx->d_gi[4 * k + 3];
  }
  gp.direct = (float4*)ctx->d_gi[8]; gp.indirect = (float4*)ctx->d_gi[9]; gp.blend = (float4*)ctx->d_gi[10];
  uint32_t* queues = ctx->d_giCtl;
  gp.counts = ctx->d_giCtl + 8 * kQueueStride;
  gp.work = ctx->d_giCtl + (8 + (kMaxStack + 2)) * kQueueStride;
  gp.sample = sample;
  gp.raw = blend25Out ? 1u : 0u;
  gp.pixels = (uint32_t)pixels;
  const uint64_t vpixels = pixels * fp.fusedFrames;
  LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_giCtl, 0, kGiCtlWords * sizeof(uint32_t), s));
  const uint32_t resident = (uint32_t)ctx->cu_count * 4u * LT_GI_STAGE_WAVES;
  const uint32_t gridA = (uint32_t)std::min<uint64_t>((uint64_t)fp.totalSquares * fp.fusedFrames, resident);
  // the primary stage's shadow rays are accumulator's (same light samples from the same camera hits): any-hit packets unless this
  // scene's accumulator frames were timed faster per lane (render_on_stream) or LT_SHADOW_PACKETS says otherwise; the bounce
  // stages' shadow rays start on scattered bounce hits and stay per lane
  SceneDev scPrimary = sc;
  {
    const char* spe = getenv("LT_SHADOW_PACKETS");
    const int timed = ctx->shadow_mode[LT_PROGRAM_ACCUMULATOR];
    scPrimary.shadowPackets = spe ? (uint32_t)std::max(0, std::min(3, atoi(spe))) : (timed < 0 ? 1u : (uint32_t)timed);
    if (scPrimary.shadowPackets == 3u) scPrimary.shadowPackets = 0u;   // (queued is accumulator's own; where it won, the rays are not packets)
  }
  // A scene of a few hundred triangles rides in LDS for the bounce stages' per-lane walks (Config::kLdsScene): workgroups of
  // eight waves share one copy.  LT_GI_LDS_SCENE=0 turns it off (A/B measurements).
  const uint64_t sceneLdsBytes = (uint64_t)ctx->n_nodes * 32 + (uint64_t)ctx->n_prims * 48;
  const char* le = getenv("LT_GI_LDS_SCENE");
  const bool ldsScene = ctx->bvh_height <= kLdsStack && sceneLdsBytes <= 16384 && !(le && atoi(le) == 0);   // (its walks keep the LDS stack)
  // With a tree of the backend's own to walk, a bounce stage is five launches instead of one (lt_kernel.hpp): its extension rays
  // through lt_trace_kernel, whose lanes take a new ray when theirs is done; the paths sorted into light hits, misses and
  // surface hits; the surface hits' light samples; their shadow rays through lt_trace_kernel; the survivors' next rays.
  // LT_GI_TRACE=0: the one-kernel stage (A/B measurements).
  const char* te = getenv("LT_GI_TRACE");
  const bool pretrace = !ldsScene && ctx->d_rank8 != nullptr && !(te && atoi(te) == 0);
  gp.hitCount = ctx->d_giCtl + (8 + 2 * (kMaxStack + 2)) * kQueueStride;
  gp.directQueue = pretrace ? 1u : 0u;
  hipLaunchKernelGGL((lt_gi_primary_kernel<CFG>), dim3(gridA), dim3(kBlock), lds, s, scPrimary, fp, gp, queues);
  LT_HIP_CHECK(ctx, hipGetLastError());
  launches++;
  gp.ldsRows = ctx->lds_ref_bytes / (kBlock * sizeof(int));   // (read by the multi-wave workgroups of the LDS-scene launches only)
  uint32_t* traceWork = ctx->d_giCtl + (8 + 3 * (kMaxStack + 2)) * kQueueStride;    // 8 per stage
  uint32_t* shadowWork = ctx->d_giCtl + (8 + 11 * (kMaxStack + 2)) * kQueueStride;  // 8 per stage
  gp.hitList = (uint32_t*)ctx->d_gi[12];
  gp.so = (float4*)ctx->d_gi[13]; gp.sd = (float4*)ctx->d_gi[14]; gp.sm = (uint4*)ctx->d_gi[15]; gp.sn = (float4*)ctx->d_gi[16];
  const char* re = getenv("LT_TRACE_REFILL");
  const uint32_t refill = re ? (uint32_t)std::max(1, std::min(64, atoi(re))) : 24u;
  const uint32_t traceLds = (uint32_t)((kTraceRows + kTraceStage) * kBlock * sizeof(int));
  const dim3 streamGrid((uint32_t)ctx->cu_count * 8u), streamBlock(256);
  for (int d = 0; d < fp.giMaxDepth; d++) {
    gp.hits = nullptr;
    if (pretrace) {
      TraceParams tp{};
      const GiQueue& q = gp.q[d & 1];
      tp.o = q.o; tp.d = q.d; tp.m = q.m;
      tp.hit = (uint4*)ctx->d_gi[11];
      tp.count = gp.counts + (size_t)d * kQueueStride;
      tp.next = traceWork + (size_t)d * 8 * kQueueStride;
      tp.refill = refill;
      tp.dead = d == 0 ? 1u : 0u;
      gp.hits = tp.hit;
      hipLaunchKernelGGL((lt_trace_kernel<kGI, false>), dim3(resident), dim3(kBlock), traceLds, s, sc, tp);
      hipLaunchKernelGGL((lt_gi_classify_kernel<CFG>), streamGrid, streamBlock, 0, s, sc, fp, gp, (uint32_t)d);
      hipLaunchKernelGGL((lt_gi_shadow_kernel<CFG>), streamGrid, streamBlock, 0, s, sc, fp, gp, (uint32_t)d);
      tp.o = gp.so; tp.d = gp.sd; tp.m = gp.sm;
      tp.occluded = (uint32_t*)ctx->d_gi[11];   // (the extension rays' hits have been read by then: lt_gi_shadow_kernel is behind us in the stream)
      gp.occluded = tp.occluded;
      tp.count = gp.hitCount + (size_t)d * kQueueStride;
      tp.next = shadowWork + (size_t)d * 8 * kQueueStride;
      tp.dead = 0u;
      hipLaunchKernelGGL((lt_trace_kernel<kGI, true>), dim3(resident), dim3(kBlock), traceLds, s, sc, tp);
      hipLaunchKernelGGL((lt_gi_finish_kernel<CFG>), streamGrid, streamBlock, 0, s, sc, fp, gp, (uint32_t)d);
      LT_HIP_CHECK(ctx, hipGetLastError());
      launches += 5;
      continue;
    }
    if (ldsScene) {
      using CFGL = Config<false, false, CFG::kDevLibm, true>;
      hipLaunchKernelGGL((lt_gi_bounce_kernel<CFGL>), dim3((resident + kLdsSceneWaves - 1) / kLdsSceneWaves), dim3(kBlock * kLdsSceneWaves),
                         (uint32_t)sceneLdsBytes + kLdsSceneWaves * ctx->lds_ref_bytes, s, sc, fp, gp, (uint32_t)d);
    } else {
      hipLaunchKernelGGL((lt_gi_bounce_kernel<CFG>), dim3(resident), dim3(kBlock), lds, s, sc, fp, gp, (uint32_t)d);
    }
    LT_HIP_CHECK(ctx, hipGetLastError());
    launches++;
  }
  hipLaunchKernelGGL((lt_gi_resolve_kernel<CFG>), dim3((uint32_t)((vpixels + 255) / 256)), dim3(256), 0, s, fp, gp, out, (uint32_t)vpixels);
  LT_HIP_CHECK(ctx, hipGetLastError());
  launches++;
  if (blend25Out) {
    FrameParams fb = fp;
    fb.accumulateN = accumulateN;
    hipLaunchKernelGGL((lt_gi_blend25_kernel<CFG>), dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, s, fb, gp, out, k0, fp.fusedFrames,
                       blend25Out, (uint32_t)pixels);
    LT_HIP_CHECK(ctx, hipGetLastError());
  }
  return LT_OK;
}

// Hand-out order of the 8x8 squares in persistent mode.  Each XCD keeps its contiguous share of the logical square list
// (render_kernel_body); inside a share, the squares holding a pixel of the image's centre row (direction.y == 0 exactly) or,
// with an unrotated camera, centre column (direction.x == 0) come first, the others follow in Z order over 64x64-pixel blocks.  Those wavefronts cannot use the packet walk or the
// NaN-free box test, and where scene geometry lies in the camera's axis planes (x = camera.x on the 1 M-triangle wall) the
// reference's NaN semantics make their rays visit every box touching the plane: 1.9 ms for such a square against 0.3 ms
// for its neighbours.  Started last they are a launch's tail; started first they overlap with everything else.
static int ensure_square_order(lt_hip_context* ctx, const lt_hip_render_desc* d, const TilePlan& p, bool unrotated, hipStream_t s,
                               const uint32_t** order, uint32_t head[8]) {
  *order = nullptr;
  for (int i = 0; i < 8; i++) head[i] = 0;
  const int64_t cx = (unrotated && d->width % 2 == 0) ? d->width / 2 : -1, cy = d->height % 2 == 0 ? d->height / 2 : -1;
  const uint32_t bpt = p.bptx * p.bpty;
  const uint64_t n = (uint64_t)p.tilesInCall * bpt;
  if (n == 0) return LT_OK;
  const std::vector<uint32_t> key = {d->width, d->height, p.tileW, p.tileH, p.tileFirst, p.tileStride, (uint32_t)cx, (uint32_t)cy};
  if (key == ctx->order_key) {
    *order = ctx->d_order;
    for (int i = 0; i < 8; i++) head[i] = ctx->order_head[i];
    return LT_OK;
  }
  std::vector<uint32_t> ord((size_t)n);
  const uint64_t q = n / 8, r = n % 8;
  for (uint32_t xcd = 0; xcd < 8; xcd++) {
    const uint64_t share = q + (xcd < r ? 1 : 0), start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    size_t pos = (size_t)start;
    std::vector<uint32_t> rest;
    rest.reserve((size_t)share);
    for (uint64_t b = start; b < start + share; b++) {
      const uint32_t k = (uint32_t)(b / bpt), sb = (uint32_t)(b % bpt);
      const uint32_t tile = p.tileFirst + k * p.tileStride;
      const int64_t x0 = (int64_t)(tile % p.tilesX) * p.tileW + (sb % p.bptx) * 8, y0 = (int64_t)(tile / p.tilesX) * p.tileH + (sb / p.bptx) * 8;
      // (a square is clipped to its tile: an 8-pixel span that crosses the tile edge does not reach the next tile's pixels)
      const int64_t x1 = std::min<int64_t>(x0 + 8, (int64_t)(tile % p.tilesX) * p.tileW + p.tileW);
      const int64_t y1 = std::min<int64_t>(y0 + 8, (int64_t)(tile / p.tilesX) * p.tileH + p.tileH);
      if ((cx >= x0 && cx < x1) || (cy >= y0 && cy < y1)) ord[pos++] = (uint32_t)b; else rest.push_back((uint32_t)b);
    }
    ctx->order_head[xcd] = (uint32_t)(pos - (size_t)start);
    // the rest in Z order over blocks of 8 x 8 squares (64 x 64 pixels): the ~1000 squares an XCD works on at any moment
    // then cover a compact image region instead of two full-width rows of squares, and its L2 a smaller part of the tree
    // (+3.6 % on the 1 M-triangle soup, neutral on the wall)
    {
      constexpr uint32_t B = 8;
      auto keyOf = [&](uint32_t b) {
        const uint32_t k = b / bpt, sb = b % bpt, tile = p.tileFirst + k * p.tileStride;
        const uint32_t sx = ((tile % p.tilesX) * p.tileW) / 8 + sb % p.bptx, sy = ((tile / p.tilesX) * p.tileH) / 8 + sb / p.bptx;
        const uint32_t bx = sx / B, by = sy / B;
        uint64_t z = 0;
        for (int i = 0; i < 16; i++) z |= ((uint64_t)((bx >> i) & 1u) << (2 * i)) | ((uint64_t)((by >> i) & 1u) << (2 * i + 1));
        return (z << 32) | ((uint64_t)(sy % B) << 16) | (sx % B);
      };
      std::vector<std::pair<uint64_t, uint32_t>> keyed;
      keyed.reserve(rest.size());
      for (uint32_t b : rest) keyed.emplace_back(keyOf(b), b);
      std::sort(keyed.begin(), keyed.end());
      for (size_t i = 0; i < keyed.size(); i++) rest[i] = keyed[i].second;
    }
    std::copy(rest.begin(), rest.end(), ord.begin() + pos);
  }
  ctx->order_key.clear();
  if (ctx->order_capacity < n) {
    if (ctx->d_order) LT_HIP_CHECK(ctx, hipFree(ctx->d_order));
    ctx->d_order = nullptr;
    ctx->order_capacity = 0;
    LT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_order, n * sizeof(uint32_t)));
    ctx->order_capacity = n;
  }
  // (rare path: image size, tiling or camera rotation changed)  No launch of an earlier call, on whatever stream, may still
  // be reading the old order; and `ord` must outlive the copy.
  LT_HIP_CHECK(ctx, hipDeviceSynchronize());
  LT_HIP_CHECK(ctx, hipMemcpy(ctx->d_order, ord.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
  ctx->order_key = key;
  *order = ctx->d_order;
  for (int i = 0; i < 8; i++) head[i] = ctx->order_head[i];
  return LT_OK;
}

// The wavefront GI pipeline for one iteration of render_on_stream's frame loop: one set of stage launches for the
// fp.fusedFrames frames of the single-sample program (samplesPerSet == 0; colours go to stageOut), or ceil(25 / samplesPerSet)
// sets for the 25 samples of the one frame of the 25-sample variant (raw colours to ctx->d_samples, blended into `image`).
static int launch_gi_sets(lt_hip_context* ctx, hipStream_t s, const SceneDev& sc, const FrameParams& fp, const LaunchConfig& lc, uint32_t lds,
                          uint64_t giPixels, uint32_t samplesPerSet, uint64_t floats, float* stageOut, float* image, uint32_t& launches) {
  const bool gi25 = samplesPerSet != 0u;
  const uint32_t sets = gi25 ? (25u + samplesPerSet - 1u) / samplesPerSet : 1u;
  for (uint32_t set = 0; set < sets; set++) {
    const uint32_t k0 = set * samplesPerSet;
    FrameParams fs = fp;
    float* blendOut = nullptr;
    float* out = stageOut;
    uint32_t sample = fp.frameCount;
    if (gi25) {
      fs.fusedFrames = std::min(samplesPerSet, 25u - k0);
      fs.frameStride = floats;
      fs.accumulateN = -1;
      blendOut = image;
      out = ctx->d_samples;
      sample = fp.frameCount * 32u + k0;
    }
    int rc;
#define LT_GI(D, M) launch_gi_sample<Config<D, false, M>>(ctx, s, sc, fs, out, lds, giPixels, sample, k0, blendOut, fp.accumulateN, launches)
    rc = lc.devlibm == 2 ? LT_GI(false, 2) : lc.devlibm == 1 ? LT_GI(false, 1) : LT_GI(false, 0);   // (never a counting launch: no deep-tree form)
#undef LT_GI
    if (rc) return rc;
  }
  return LT_OK;
}

// Folds the nf sample images a fused launch left in ctx->d_samples into `out` (timed by its own event pair, so that
// lt_hip_stats::render_ms can leave it out).
static int launch_running_mean(lt_hip_context* ctx, hipStream_t s, const FrameParams& fp, uint64_t floats, uint32_t nf, int32_t base,
                               bool paddedTiles, float* out, bool lastOfCall) {
  const uint32_t threads = 256;
  while (ctx->mean_events.size() < 2 * (size_t)(ctx->mean_pairs + 1)) {
    hipEvent_t e;
    LT_HIP_CHECK(ctx, hipEventCreate(&e));
    ctx->mean_events.push_back(e);
  }
  LT_HIP_CHECK(ctx, hipEventRecord(ctx->mean_events[2 * ctx->mean_pairs], s));
  ctx->fold_pieced = false;
  if (lastOfCall && ctx->fold_piece_bytes != 0 && out == ctx->d_out && s == ctx->stream) {
    // lt_hip_render: the call's last fold in the eight pieces of the read-back, an event behind each, so that piece k travels
    // (enqueue_readback, on the copy stream) while piece k + 1 is folded
    for (hipEvent_t& e : ctx->fold_ev) if (!e) LT_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint64_t per = ctx->fold_piece_bytes / sizeof(float);
    for (int k = 0; k < 8; k++) {
      const uint64_t lo = std::min(floats, (uint64_t)k * per), hi = std::min(floats, lo + per);
      if (hi > lo)
        lt_running_mean_kernel<<<dim3((uint32_t)((hi - lo + threads - 1) / threads)), dim3(threads), 0, s>>>(ctx->d_samples, nf, floats, out, lo, hi, base, fp,
                                                                                                         paddedTiles ? 1 : 0);
      LT_HIP_CHECK(ctx, hipEventRecord(ctx->fold_ev[k], s));
    }
    ctx->fold_pieced = true;
  } else {
    lt_running_mean_kernel<<<dim3((uint32_t)((floats + threads - 1) / threads)), dim3(threads), 0, s>>>(ctx->d_samples, nf, floats, out, 0ull, floats,
                                                                                                    base, fp, paddedTiles ? 1 : 0);
  }
  LT_HIP_CHECK(ctx, hipGetLastError());
  LT_HIP_CHECK(ctx, hipEventRecord(ctx->mean_events[2 * ctx->mean_pairs + 1], s));
  ctx->mean_pairs++;
  return LT_OK;
}

// How a call is cut into launches (render_on_stream): which execution path the global-illumination programs take, how many
// frames (or, for the 25-sample variant, samples of one frame) travel through one launch, and the scratch memory for them.
struct FusionPlan {
  bool giWavefront = false;     // wavefront pipeline instead of the one-lane-per-pixel kernel
  bool gi25Sets = false;        // 25-sample variant through the pipeline: samplesPerSet samples per set of stage launches
  uint32_t chunk = 1;           // frames per launch (> 1: fused; lt_running_mean_kernel folds them)
  uint32_t samplesPerSet = 0;
  uint64_t giPixels = 0;        // compact output pixels of one frame
};

static int plan_fusion(lt_hip_context* ctx, const lt_hip_render_desc* d, const TilePlan& p, uint32_t frames, uint64_t nblocks, bool stats,
                       bool persistent, int giMaxDepth, FusionPlan& out) {
  // The global-illumination programs run as a wavefront pipeline with path compaction when the scene is big enough for the
  // traversal to dominate the ~18 launches and the queue traffic per sample (1 M-triangle wall at 4K, 16 bounces: 31 ms
  // against 52 ms for the one-lane-per-pixel kernel; 42-triangle Cornell box at 1080p: 3.5 ms against 2.7 ms), or when the
  // launches serve many frames at once and paths are long (Cornell 1080p, 16 frames per call: 1.55 ms against 1.89 ms per
  // sample at 16 bounces, but 1.12 against 0.78 ms at 4; the 25 samples of one frame of the 25-sample variant count as many:
  // 37 against 55 ms at 16 bounces, 27.5 against 21 ms at 4), and never when work is being counted (the counting kernels
  // re-trace like the reference does).  LT_GI_MEGAKERNEL=1 / =0 force one or the other (A/B measurements, tests of both
  // paths on small scenes).
  const char* ge = getenv("LT_GI_MEGAKERNEL");
  const bool giProgram = d->program == LT_PROGRAM_GLOBAL_ILLUMINATION || d->program == LT_PROGRAM_GLOBAL_ILLUMINATION_25;
  // (a scene small enough to ride in LDS through the bounce stages, launch_gi_sample, takes the pipeline from 2 bounces on:
  // Cornell 1080p, 16 frames per call, 4 bounces: 8.6 against 10.9 ms; 25-sample variant 12.9 against 19.9 ms)
  const bool ldsScene = ctx->bvh_height <= kLdsStack && (uint64_t)ctx->n_nodes * 32 + (uint64_t)ctx->n_prims * 48 <= 16384 &&
                        !(getenv("LT_GI_LDS_SCENE") && atoi(getenv("LT_GI_LDS_SCENE")) == 0);
  const bool giManyLongPaths = giMaxDepth > (ldsScene ? 1 : 8) && (d->program == LT_PROGRAM_GLOBAL_ILLUMINATION_25 ||
                                                                   (d->program == LT_PROGRAM_GLOBAL_ILLUMINATION && frames > 1 && d->accumulate));
  const bool giWavefront = giProgram && !stats && nblocks > 0 && (ge ? atoi(ge) == 0 : (ctx->n_prims >= 1024u || giManyLongPaths));
  const uint64_t giPixels = (uint64_t)p.tilesInCall * p.tileW * p.tileH;
  const uint64_t giSlots = std::max<uint64_t>(giPixels, nblocks * kBlock);   // (a direct-mapped path queue has a slot per lane of every square)
  if (giWavefront && giSlots > 0xffffffffull) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "too many pixels for the GI path queues");
  // Several samples of a running mean in ONE launch.  A launch cannot end before its slowest wavefront does -- one 8x8 square
  // is a dependent chain of several hundred node fetches, ~0.3-0.6 ms on the 1 M-triangle scene, 1.9 ms for the squares on
  // the image's centre column -- so a launch per sample pays that drain once per sample: 0.65 ms of a 4.5 ms launch for the
  // whole 4K frame, and of a 1.2 ms launch for one GPU's eighth of it.  Fused, the work items are (frame, square) pairs, all
  // independent: each stores its un-accumulated colour in its frame's slice of a scratch buffer and lt_running_mean_kernel
  // folds the slices in frame order afterwards (same arithmetic, same order: bit-identical).  LT_FUSED_FRAMES=0 turns it
  // off (A/B measurements), LT_FUSED_BYTES caps the scratch memory (default 16 GiB of the 288; tests use it to force chunks).
  // The wavefront GI pipeline fuses the same way (single-sample program only: the 25-sample blend is sequential per pixel):
  // its ~18 stage launches then serve all frames of a chunk, each path carrying its frame; its per-frame scratch is the path
  // queues and the direct / indirect images (11 arrays of 16 bytes per pixel) besides the sample image.
  // The 25-sample variant fuses the samples of ONE frame instead (its frames stay sequential): `chunk` is then the number of
  // samples k per set of launches, and lt_gi_blend25_kernel replaces the running mean.
  uint32_t chunk = 1;
  const bool giFusable = giWavefront && d->program == LT_PROGRAM_GLOBAL_ILLUMINATION;
  const bool gi25Sets = giWavefront && d->program == LT_PROGRAM_GLOBAL_ILLUMINATION_25;
  if (gi25Sets || ((giFusable || (persistent && !giWavefront)) && !stats && frames > 1 && d->accumulate && nblocks > 0)) {
    const char* fe = getenv("LT_FUSED_FRAMES");
    const char* fb = getenv("LT_FUSED_BYTES");
    const uint64_t cap = fb ? strtoull(fb, nullptr, 10) : (16ull << 30);
    const uint64_t frameBytes = p.floats * sizeof(float);
    const uint64_t scratchPerFrame = frameBytes + (giWavefront ? giSlots * 16 * 17 : 0);
    if (!(fe && atoi(fe) == 0) && frameBytes > 0)
      chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(gi25Sets ? 25u : frames), cap / scratchPerFrame, 0xffffffffull / nblocks,
                                                                  giWavefront ? 0xffffffffull / std::max<uint64_t>(giSlots, 1) : ~0ull}));
    if ((chunk > 1 || gi25Sets) && ctx->d_samples_bytes < chunk * frameBytes) {
      if (ctx->d_samples) LT_HIP_CHECK(ctx, hipFree(ctx->d_samples));
      ctx->d_samples = nullptr;
      ctx->d_samples_bytes = 0;
      // a device that cannot spare the scratch memory gets shorter launches, down to one sample per launch
      while ((chunk > 1 || gi25Sets) && hipMalloc((void**)&ctx->d_samples, chunk * frameBytes) != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_samples = nullptr;
        if (chunk == 1) return fail(ctx, LT_ERR_HIP, "out of device memory for one sample image");
        chunk /= 2;
      }
      ctx->d_samples_bytes = ctx->d_samples ? chunk * frameBytes : 0;
    }
  }
  if (giWavefront) {
    int erc;
    while ((erc = ensure_gi_buffers(ctx, giSlots * chunk)) != LT_OK && chunk > 1) chunk /= 2;   // (frees what it got, retries smaller)
    if (erc) return erc;
  }
  out.giWavefront = giWavefront;
  out.gi25Sets = gi25Sets;
  out.giPixels = giPixels;
  out.samplesPerSet = gi25Sets ? chunk : 0u;   // 25-sample variant: samples k per set of stage launches
  out.chunk = gi25Sets ? 1u : chunk;           // ... and its frames stay one per iteration of the caller's loop
  return LT_OK;
}

static int render_on_stream(lt_hip_context* ctx, const lt_hip_render_desc* d, float* out_device, uint64_t out_bytes, hipStream_t s) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  if (!ctx->has_scene) return fail(ctx, LT_ERR_NO_SCENE, "lt_hip_render before lt_hip_set_scene");
  TilePlan p;
  std::string msg;
  int rc = plan_tiles(d, p, msg);
  if (rc) return fail(ctx, rc, msg);
  const bool userProgram = d->program >= LT_PROGRAM_USER_BASE;
  if (userProgram ? (size_t)(d->program - LT_PROGRAM_USER_BASE) >= ctx->user_programs.size()
                  : (d->program < LT_PROGRAM_BASIC || d->program > LT_PROGRAM_CUSTOM_OPENCL))
    return fail(ctx, LT_ERR_UNKNOWN_PROGRAM, "unknown program");
  if (userProgram && (d->flags & (LT_RENDER_FLAG_STATS | LT_RENDER_FLAG_PIXEL_COUNTERS)))
    return fail(ctx, LT_ERR_INVALID_ARGUMENT, "user programs are compiled without the counting variants");
  if (d->kernel_mode != LT_KERNEL_MODE_LINEAR && d->kernel_mode != LT_KERNEL_MODE_TILE) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "unknown kernel mode");
  if (!out_device) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "output pointer is NULL");
  if (out_bytes < p.floats * sizeof(float)) return fail(ctx, LT_ERR_BUFFER_TOO_SMALL, "output buffer smaller than the image/tile stack");
  if (d->gi_max_depth < 0 || d->gi_max_depth > 64) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "gi_max_depth out of range");
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));

  float cam[7];
  uint32_t camFrame;
  memcpy(cam, d->camera, 28);
  memcpy(&camFrame, d->camera + 24, 4);

  SceneDev sc;
  sc.nodes = (const float4*)ctx->d_nodes;
  sc.wide = ctx->d_wide ? (const uint4*)ctx->d_wide + 4 : nullptr;   // (behind the 64 bytes that hold the grid)
  sc.ownPairs = (const float4*)ctx->d_pairs2;
  sc.rank8 = (const uint32_t*)ctx->d_rank8;
  sc.nWide = ctx->d_rank8 ? ctx->n_wide : 0u;
  sc.tris = (const float4*)ctx->d_tris;
  sc.prims = (const float*)ctx->d_prims;
  sc.mats = (const Material*)ctx->d_mats;
  sc.lights = (const Lights*)ctx->d_lights;
  sc.n_nodes = ctx->n_nodes; sc.n_prims = ctx->n_prims; sc.n_mats = ctx->n_mats;
  // Shadow rays as any-hit packets, per lane, chosen per wavefront (traverse(), lt_device.hpp) or queued for lt_trace_kernel
  // (accumulator): which is fastest depends on the scene (wall: packets; soup: the queue), so each (scene, program, image geometry,
  // frames per launch) is timed once, on the first launch that can be repeated without changing the result, and the fastest
  // walk kept.  LT_SHADOW_PACKETS=0/1/2/3 forces one (tests, A/B measurements).
  const char* spe = getenv("LT_SHADOW_PACKETS");
  const bool hasShadowRays = d->program == LT_PROGRAM_ACCUMULATOR || d->program == LT_PROGRAM_BASIC_LIGHTING;   // (the GI programs' kernels hold the per-lane walk only)
  // (keyed on the image geometry too: how coherent a wavefront's 64 shadow rays are depends on how large its 8x8 pixels are in
  // the scene; shadow_mode[program] keeps the most recent verdict for callers without a geometry of their own: the GI pipeline)
  std::vector<uint32_t> shadowKey = {(uint32_t)d->program, d->width, d->height, p.tileW, p.tileH, p.tileFirst, p.tileStride, 0u};   // (+ frames per launch, below)
  int shadowMode = 0;   // (looked up once the frames per launch are known, below)
  sc.shadowPackets = 0u;
  sc.shadowQueue = nullptr;
  sc.shadowCap = 0u;
  {
    const char* se = getenv("LT_SHADOW_SPREAD");
    const float thr = se ? (float)atof(se) : 0.02f;
    sc.shadowSpread = thr * thr;
  }
  sc.ldsNodes = sc.ldsTris = 0u;
  sc.fastRcp = 0u;

  FrameParams fp{};
  fp.camx = cam[0]; fp.camy = cam[1]; fp.camz = cam[2];
  fp.apx = 0.0f; fp.apy = 0.0f; fp.apz = 5.0f;
  fp.cosYaw = (float)std::cos((double)cam[3]);
  fp.sinYaw = (float)std::sin((double)cam[3]);
  fp.yaw = cam[3];
  fp.width = d->width; fp.height = d->height; fp.depth = d->depth;
  fp.clampOutput = d->kernel_mode == LT_KERNEL_MODE_LINEAR;
  {
    const char* sm = getenv("LT_SQUARE_MAJOR");
    fp.squareMajor = (sm && atoi(sm) == 0) ? 0u : 1u;
  }
  fp.giMaxDepth = d->gi_max_depth ? d->gi_max_depth : 16;
  fp.tileW = p.tileW; fp.tileH = p.tileH; fp.tilesX = p.tilesX; fp.tileFirst = p.tileFirst; fp.tileStride = p.tileStride;
  fp.tilesInCall = p.tilesInCall;
  fp.blocksPerTileX = p.bptx; fp.blocksPerTile = p.bptx * p.bpty;

  const bool pixelCounters = (d->flags & LT_RENDER_FLAG_PIXEL_COUNTERS) != 0;
  if (pixelCounters && d->depth < 4) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "LT_RENDER_FLAG_PIXEL_COUNTERS needs depth >= 4");
  fp.pixelCounters = pixelCounters;
  const bool stats = pixelCounters || (d->flags & LT_RENDER_FLAG_STATS) != 0;
  const bool deep = ctx->bvh_height > kLdsStack;
  // The default flavour is bit-identical to the reference's OpenCL kernels as RendererOpenCL builds them on this GPU (Math<2>);
  // LT_RENDER_FLAG_STRICT_MATH / _PORTABLE_MATH select the other two (include/lenstrace_hip.h).
  if ((d->flags & LT_RENDER_FLAG_PORTABLE_MATH) && (d->flags & LT_RENDER_FLAG_STRICT_MATH))
    return fail(ctx, LT_ERR_INVALID_ARGUMENT, "LT_RENDER_FLAG_PORTABLE_MATH and LT_RENDER_FLAG_STRICT_MATH exclude each other");
  const int devlibm = (d->flags & LT_RENDER_FLAG_PORTABLE_MATH) ? 0 : (d->flags & LT_RENDER_FLAG_STRICT_MATH) ? 1 : 2;
  sc.fastRcp = devlibm == 2 ? 1u : 0u;
  const LaunchConfig lc{deep, stats, devlibm};
  const uint32_t frames = d->frame_count ? d->frame_count : 1;
  const uint64_t nblocks = (uint64_t)p.tilesInCall * fp.blocksPerTile;
  if (nblocks > 0x7fffffffull) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "too many workgroups");

  if (stats) LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_stats, 0, 8 * sizeof(unsigned long long), s));
  // persistent wavefronts by default; LT_PERSISTENT=0 selects one-square-per-workgroup dispatch (A/B measurements)
  const char* pe = getenv("LT_PERSISTENT");
  const bool persistent = !pe || atoi(pe) != 0;
  if (persistent) {
    if (ctx->queue_frames < frames) {
      if (ctx->d_queues) LT_HIP_CHECK(ctx, hipFree(ctx->d_queues));
      ctx->d_queues = nullptr;
      ctx->queue_frames = 0;
      LT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_queues, (size_t)frames * 8 * kQueueStride * sizeof(uint32_t)));
      ctx->queue_frames = frames;
    }
    LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queues, 0, (size_t)frames * 8 * kQueueStride * sizeof(uint32_t), s));
  }
  fp.totalSquares = (uint32_t)nblocks;
  fp.persistent = persistent;
  fp.order = nullptr;
  if (persistent && !getenv("LT_NATURAL_ORDER")) {
    const int orc = ensure_square_order(ctx, d, p, fp.sinYaw == 0.0f, s, &fp.order, fp.orderHead);
    if (orc) return orc;
  }
  FusionPlan fu;
  {
    const int frc = plan_fusion(ctx, d, p, frames, nblocks, stats, persistent, fp.giMaxDepth, fu);
    if (frc) return frc;
  }
  const bool giWavefront = fu.giWavefront, gi25Sets = fu.gi25Sets, fused = fu.chunk > 1;
  const uint32_t chunk = fu.chunk, samplesPerSet = fu.samplesPerSet;
  const uint64_t giPixels = fu.giPixels;
  const bool paddedTiles = d->width % p.tileW != 0 || d->height % p.tileH != 0;
  {   // (frames per launch in three classes: a launch pays a fixed price for its slowest squares, which the walks share out differently)
    const uint32_t lf = fused ? std::min(chunk, frames) : 1u;
    shadowKey[7] = lf == 1u ? 1u : lf < 8u ? 2u : 8u;
  }
  if (spe) shadowMode = std::max(0, std::min(3, atoi(spe)));
  else if (hasShadowRays) {
    auto it = ctx->shadow_modes.find(shadowKey);
    shadowMode = it == ctx->shadow_modes.end() ? -1 : it->second;
  }
  if (shadowMode < 0 && (d->flags & LT_RENDER_FLAG_NO_WALK_TIMING))   // the caller wants no timing launches in this call
    shadowMode = ctx->shadow_mode[d->program] >= 0 ? ctx->shadow_mode[d->program] : 1;
  if (shadowMode == 3 && d->program != LT_PROGRAM_ACCUMULATOR) shadowMode = 0;   // (queued shadow rays are accumulator's)
  sc.shadowPackets = shadowMode > 0 ? (uint32_t)shadowMode : 0u;
  ctx->mean_pairs = 0;
  LT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, s));
  uint32_t launches = 0;
  if (nblocks > 0) {
    uint32_t launchIndex = 0;
    for (uint32_t f = 0; f < frames; launchIndex++) {
      const uint32_t nf = fused ? std::min(chunk, frames - f) : 1u;   // frames of this launch
      fp.frameCount = d->frame_count ? d->frame_first + f : camFrame;
      fp.accumulateN = (d->frame_count && d->accumulate && !fused) ? (int32_t)(d->accumulate_base + f) : -1;
      fp.fusedFrames = nf;
      fp.frameStride = fused ? p.floats : 0;
      float* const out_launch = fused ? ctx->d_samples : out_device;
      const uint32_t firstFrame = f;
      f += nf;
      const uint32_t resident = (uint32_t)ctx->cu_count * 32u;   // every wave slot of the chip, once
      const dim3 grid(persistent ? (uint32_t)std::min<uint64_t>(nblocks * nf, resident) : (uint32_t)nblocks);
      uint32_t* queues = persistent ? ctx->d_queues + (size_t)launchIndex * 8 * kQueueStride : nullptr;
      // LDS stack rows: a lane never holds more entries than a node has interior ancestors (= bvh_height, validate_scene)
      // The counting kernels (and the LDS-resident small scenes of the GI bounce stage: launch_gi_sample) keep one stack entry
      // per lane and level of the caller's tree in LDS; the others the kOwnRows rows of the per-lane walks over the own tree (the
      // packet walks park their stack register in the first of them).
      ctx->lds_ref_bytes = (uint32_t)std::max(kPacketRows, std::min(ctx->bvh_height, kLdsStack)) * kBlock * sizeof(int);
      uint32_t lds = stats ? ctx->lds_ref_bytes : (uint32_t)std::max(kPacketRows, kOwnRows) * kBlock * sizeof(int);
      // LT_DEBUG_LDS_ROWS (occupancy experiments, tests): more rows than the launch needs, for every kernel; FEWER only for the
      // counting kernels, whose deep-tree form keeps what does not fit in private memory (the others index their rows with
      // compile-time bounds).  The kernel is told what it got (FrameParams::ldsRows) and the form is chosen from that.
      if (const char* e = getenv("LT_DEBUG_LDS_ROWS")) {
        const uint32_t want = (uint32_t)std::max(1, std::min(160, atoi(e))) * (uint32_t)(kBlock * sizeof(int));
        lds = stats ? want : std::max(lds, want);
      }
      fp.ldsRows = lds / (uint32_t)(kBlock * sizeof(int));
      LaunchConfig lcl = lc;
      lcl.deep = stats && (uint32_t)ctx->bvh_height > fp.ldsRows;   // (the non-counting kernels have no deep-tree form: they keep no per-lane stack of the caller's tree in LDS)
      if (giWavefront) {
        SceneDev scGi = sc;
        scGi.shadowPackets = spe ? sc.shadowPackets : 0u;   // the pipeline's bounce stages cast incoherent shadow rays: per lane
        const int grc = launch_gi_sets(ctx, s, scGi, fp, lcl, lds, giPixels, gi25Sets ? samplesPerSet : 0u, p.floats, out_launch, out_device, launches);
        if (grc) return grc;
        launches--;   // (counted again below)
      } else if (userProgram) {
        const lt_hip_context::UserProgram& up = ctx->user_programs[d->program - LT_PROGRAM_USER_BASE];
        unsigned long long* statsPtr = ctx->d_stats;
        float* outPtr = out_launch;
        void* args[] = {(void*)&sc, (void*)&fp, (void*)&outPtr, (void*)&statsPtr, (void*)&queues};
        const hipFunction_t fn = devlibm == 2 ? up.lds : devlibm == 1 ? up.ldsStrict : up.ldsPortable;
        LT_HIP_CHECK(ctx, hipModuleLaunchKernel(fn, grid.x, 1, 1, kBlock, 1, 1, lds, s, args, nullptr));
      } else {
        // One frame-set of the call with a given shadow-ray walk: the render launch and, when the shadow rays are queued (mode 3:
        // accumulator, a tree of the backend's own, a launch that overwrites what it writes), lt_trace_kernel over the queue and
        // the kernel that blacks out the occluded samples.
        auto launch_render = [&](const FrameParams& fpl, dim3 g) {
          switch (d->program) {
            case LT_PROGRAM_BASIC: launch_program<kBasic>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues); break;
            case LT_PROGRAM_BASIC_LIGHTING: launch_program<kBasicLighting>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues); break;
            case LT_PROGRAM_ACCUMULATOR:
              if (sc.shadowPackets == 3u) launch_program<kAccumulatorQueue>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues);
              else launch_program<kAccumulator>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues);
              break;
            case LT_PROGRAM_GLOBAL_ILLUMINATION: launch_program<kGI>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues); break;
            case LT_PROGRAM_GLOBAL_ILLUMINATION_25: launch_program<kGI25>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues); break;
            default: launch_program<kCustom>(lcl, g, lds, s, sc, fpl, out_launch, ctx->d_stats, queues); break;
          }
        };
        const bool queueOk = d->program == LT_PROGRAM_ACCUMULATOR && persistent && !stats && ctx->d_rank8 != nullptr && fp.accumulateN < 0 &&
                             nblocks * nf * kBlock < 0xffffffffull;
        auto launch_walk = [&](uint32_t mode, const FrameParams& fpl, dim3 g) -> int {
          sc.shadowPackets = mode;
          if (mode != 3u) { launch_render(fpl, g); return LT_OK; }
          const uint64_t slots = nblocks * fpl.fusedFrames * kBlock;
          if (ctx->shadowq_slots < slots) {
            if (ctx->d_shadowq) LT_HIP_CHECK(ctx, hipFree(ctx->d_shadowq));
            ctx->d_shadowq = nullptr;
            ctx->shadowq_slots = 0;
            LT_HIP_CHECK(ctx, hipMalloc(&ctx->d_shadowq, slots * 52));   // (origin + tmax, direction, pixel / primitive / frame: 48 bytes; its fate: 4)
            ctx->shadowq_slots = slots;
          }
          if (!ctx->d_shadowCtl) LT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_shadowCtl, 9 * kQueueStride * sizeof(uint32_t)));
          LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_shadowCtl, 0, 8 * kQueueStride * sizeof(uint32_t), s));
          LT_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->d_shadowCtl + 8 * kQueueStride), (int)(uint32_t)slots, 1, s));
          sc.shadowQueue = (float4*)ctx->d_shadowq;
          sc.shadowCap = (uint32_t)ctx->shadowq_slots;
          launch_render(fpl, g);
          TraceParams tp{};
          tp.o = sc.shadowQueue; tp.d = sc.shadowQueue + sc.shadowCap; tp.m = (const uint4*)(sc.shadowQueue + 2 * (size_t)sc.shadowCap);
          tp.occluded = (uint32_t*)(sc.shadowQueue + 3 * (size_t)sc.shadowCap);
          tp.count = ctx->d_shadowCtl + 8 * kQueueStride;
          tp.next = ctx->d_shadowCtl;
          const char* re = getenv("LT_TRACE_REFILL");
          tp.refill = re ? (uint32_t)std::max(1, std::min(64, atoi(re))) : 24u;
          tp.dead = 1u;
          hipLaunchKernelGGL((lt_trace_kernel<kGI, true>), dim3(resident), dim3(kBlock), (uint32_t)((kTraceRows + kTraceStage) * kBlock * sizeof(int)), s, sc, tp);
          hipLaunchKernelGGL(lt_shadow_resolve_kernel, dim3((uint32_t)ctx->cu_count * 8u), dim3(256), 0, s, tp.m, (const uint32_t*)tp.occluded, (uint32_t)slots, out_launch,
                             fpl.frameStride, fpl.depth);
          LT_HIP_CHECK(ctx, hipGetLastError());
          launches += 2;
          return LT_OK;
        };
        if (shadowMode == 3 && !queueOk) shadowMode = spe ? 0 : -1;   // (a forced or remembered mode 3 where it cannot run)
        // Time the shadow-ray walks (any-hit packets, per lane, chosen per wavefront, queued for lt_trace_kernel) once per (scene,
        // program, image geometry, frames per launch), ahead of a launch whose output they may scribble on (it overwrites what it
        // writes, as every fused launch does): the launch as it is -- a verdict on fewer frames is another launch's verdict: a
        // launch pays a fixed price for its slowest squares, which the walks share out differently -- once per walk after one
        // untimed run.  The fastest wins, unless the walk an earlier verdict on this scene picked is within 3 % of it: two walks
        // that close must not take turns from call to call.
        const bool calibrate = shadowMode < 0 && persistent && !stats && ctx->bvh_height <= kLdsStack && fp.accumulateN < 0;
        if (calibrate) {
          for (hipEvent_t& e : ctx->cal_ev) if (!e) LT_HIP_CHECK(ctx, hipEventCreate(&e));
          const uint32_t kOrder[4] = {1u, 0u, 2u, 3u};
          const int candidates = queueOk ? 4 : 3;
          for (int pass = -1; pass < candidates; pass++) {
            if (pass >= 0) LT_HIP_CHECK(ctx, hipEventRecord(ctx->cal_ev[2 * pass], s));
            const int wrc = launch_walk(kOrder[pass < 0 ? 0 : pass], fp, grid);
            if (wrc) return wrc;
            if (pass >= 0) LT_HIP_CHECK(ctx, hipEventRecord(ctx->cal_ev[2 * pass + 1], s));
            LT_HIP_CHECK(ctx, hipMemsetAsync(queues, 0, 8 * kQueueStride * sizeof(uint32_t), s));
          }
          float t[4] = {0, 0, 0, 0};
          LT_HIP_CHECK(ctx, hipEventSynchronize(ctx->cal_ev[2 * candidates - 1]));
          for (int k = 0; k < candidates; k++) LT_HIP_CHECK(ctx, hipEventElapsedTime(&t[k], ctx->cal_ev[2 * k], ctx->cal_ev[2 * k + 1]));
          if (getenv("LT_DEBUG_CALIBRATION"))
            fprintf(stderr, "shadow-walk timing (ms, %u frames): packets %.3f, per lane %.3f, per wavefront %.3f, queued %.3f\n", fp.fusedFrames, t[0], t[1], t[2], t[3]);
          int best = 0;
          for (int k = 1; k < candidates; k++) if (t[k] < t[best]) best = k;
          shadowMode = (int)kOrder[best];
          const int earlier = ctx->shadow_mode[d->program];
          for (int k = 0; k < candidates; k++)
            if ((int)kOrder[k] == earlier && t[k] <= 1.03f * t[best]) shadowMode = earlier;
          ctx->shadow_modes[shadowKey] = shadowMode;
          ctx->shadow_mode[d->program] = shadowMode;
          launches += (uint32_t)candidates + 1u;
        }
        const int wrc = launch_walk((uint32_t)std::max(0, shadowMode), fp, grid);
        if (wrc) return wrc;
      }
      LT_HIP_CHECK(ctx, hipGetLastError());
      launches++;
      if (fused) {
        const int mrc = launch_running_mean(ctx, s, fp, p.floats, nf, (int32_t)(d->accumulate_base + firstFrame), paddedTiles, out_device, f >= frames);
        if (mrc) return mrc;
      }
    }
  }
  LT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, s));
  ctx->last = lt_hip_stats{};
  ctx->last.frames = frames;
  ctx->last.kernel_launches = launches;
  ctx->last.shadow_packets = shadowMode;
  // pixels actually inside the image for this call's tiles
  uint64_t px = 0;
  for (uint32_t k = 0; k < p.tilesInCall; k++) {
    const uint32_t tile = p.tileFirst + k * p.tileStride, tx = tile % p.tilesX, ty = tile / p.tilesX;
    const uint32_t w = std::min(p.tileW, d->width - tx * p.tileW), h = std::min(p.tileH, d->height - ty * p.tileH);
    px += (uint64_t)w * h;
  }
  ctx->last.pixels = px;
  ctx->last_stream = s;
  ctx->pending = true;
  ctx->pending_stats = stats;
  return LT_OK;
}

static int finish_pending(lt_hip_context* ctx) {
  if (!ctx->pending) return LT_OK;
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  LT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.0f;
  LT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->last.kernel_ms = ms;
  float meanMs = 0.0f;
  for (uint32_t i = 0; i < ctx->mean_pairs; i++) {
    float m = 0.0f;
    LT_HIP_CHECK(ctx, hipEventElapsedTime(&m, ctx->mean_events[2 * i], ctx->mean_events[2 * i + 1]));
    meanMs += m;
  }
  ctx->last.render_ms = ms - meanMs;
  if (ctx->pending_stats) {
    unsigned long long h[8];
    LT_HIP_CHECK(ctx, hipMemcpy(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost));
#ifdef LT_DEBUG_WAVE_COUNTERS
    fprintf(stderr, "[lt debug] wave-level: node steps %llu (lane-level %llu, utilisation %.3f), triangle blocks %llu (lane-level %llu, utilisation %.3f), outer iterations %llu\n",
            h[4], h[2], (double)h[2] / (64.0 * (double)h[4]), h[5], h[3], (double)h[3] / (64.0 * (double)h[5]), h[6]);
#endif
    ctx->last.rays = h[0]; ctx->last.shadow_rays = h[1]; ctx->last.node_visits = h[2]; ctx->last.tri_tests = h[3];
  }
  ctx->pending = false;
  return LT_OK;
}

extern "C" int lt_hip_render_device(lt_hip_context* ctx, const lt_hip_render_desc* desc, float* out_device, uint64_t out_bytes,
                                    void* hip_stream) {
  return render_on_stream(ctx, desc, out_device, out_bytes, (hipStream_t)hip_stream);
}

// The read-back of lt_hip_render: device -> the context's pinned buffer in eight pieces (enqueued behind the kernels), each piece
// copied on to the caller's buffer by a few host threads as soon as it has arrived, so that the link and the host's copy overlap.
// LT_PINNED_READBACK=0: one hipMemcpyAsync into the caller's (pageable) buffer, as round 2 did it.
static int enqueue_readback(lt_hip_context* ctx, uint64_t need, float* out_host, bool& staged) {
  const char* pe = getenv("LT_PINNED_READBACK");
  staged = !(pe && atoi(pe) == 0) && need >= (1u << 20);
  if (staged && ctx->h_out_bytes < need) {
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    ctx->h_out = nullptr;
    ctx->h_out_bytes = 0;
    if (hipHostMalloc(&ctx->h_out, need, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      ctx->h_out = nullptr;
      staged = false;
    } else {
      ctx->h_out_bytes = need;
    }
  }
  if (!staged) {
    LT_HIP_CHECK(ctx, hipMemcpyAsync(out_host, ctx->d_out, need, hipMemcpyDeviceToHost, ctx->stream));
    return LT_OK;
  }
  for (hipEvent_t& e : ctx->out_ev) if (!e) LT_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const uint64_t piece = ((need + 7) / 8 + 4095) / 4096 * 4096;
  hipStream_t cs = ctx->stream;
  if (ctx->fold_pieced && ctx->fold_piece_bytes == piece) {   // (the frame's last fold came in these pieces: each travels behind its own)
    if (!ctx->copy_stream) LT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    cs = ctx->copy_stream;
  }
  for (int k = 0; k < 8; k++) {
    const uint64_t off = std::min(need, (uint64_t)k * piece), n = std::min(piece, need - off);
    if (cs != ctx->stream) LT_HIP_CHECK(ctx, hipStreamWaitEvent(cs, ctx->fold_ev[k], 0));
    if (n) LT_HIP_CHECK(ctx, hipMemcpyAsync((char*)ctx->h_out + off, (const char*)ctx->d_out + off, n, hipMemcpyDeviceToHost, cs));
    LT_HIP_CHECK(ctx, hipEventRecord(ctx->out_ev[k], cs));
  }
  return LT_OK;
}
static int finish_readback(lt_hip_context* ctx, uint64_t need, float* out_host, bool staged) {
  if (!staged) {
    LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LT_OK;
  }
  const uint64_t piece = ((need + 7) / 8 + 4095) / 4096 * 4096;
  const int threads = std::max(1, std::min(8, host_threads()));
  std::atomic<int> failed{0};
  auto work = [&](int t, int of) {
    (void)hipSetDevice(ctx->device);
    for (int k = 0; k < 8; k++) {
      if (hipEventSynchronize(ctx->out_ev[k]) != hipSuccess) { failed = 1; return; }
      const uint64_t off = std::min(need, (uint64_t)k * piece), n = std::min(piece, need - off);
      const uint64_t lo = off + n * (uint64_t)t / (uint64_t)of, hi = off + n * (uint64_t)(t + 1) / (uint64_t)of;
      memcpy((char*)out_host + lo, (const char*)ctx->h_out + lo, (size_t)(hi - lo));
    }
  };
  std::vector<std::thread> pool;
  int started = 1;
  try {
    for (; started < threads; started++) pool.emplace_back(work, started, threads);
  } catch (...) {
  }
  work(0, threads);
  for (std::thread& th : pool) th.join();
  for (int t = started; t < threads; t++) work(t, threads);   // (the slices of the threads that could not be had)
  LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->copy_stream) LT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
  if (failed) return fail(ctx, LT_ERR_HIP, "read-back failed");
  return LT_OK;
}

static int render_to_host(lt_hip_context* ctx, const lt_hip_render_desc* desc, float* out_host, uint64_t out_bytes, uint64_t& need, bool& staged) {
  if (!out_host) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "output pointer is NULL");
  TilePlan p;
  std::string msg;
  int rc = plan_tiles(desc, p, msg);
  if (rc) return fail(ctx, rc, msg);
  need = p.floats * sizeof(float);
  if (out_bytes < need) return fail(ctx, LT_ERR_BUFFER_TOO_SMALL, "outputBufferSize smaller than W*H*depth floats");
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (ctx->d_out_bytes < need) {
    if (ctx->d_out) LT_HIP_CHECK(ctx, hipFree(ctx->d_out));
    ctx->d_out = nullptr;
    ctx->d_out_bytes = 0;
    LT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_out, need ? need : 4));
    ctx->d_out_bytes = need;
  }
  // a running mean continues from the caller's buffer when accumulate_base > 0
  if (desc->frame_count && desc->accumulate && desc->accumulate_base > 0)
    LT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_out, out_host, need, hipMemcpyHostToDevice, ctx->stream));
  else
    LT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_out, 0, need, ctx->stream));
  {   // (the pieces of the read-back, for the call's last fold: enqueue_readback's arithmetic; LT_PINNED_READBACK=0: one piece, one fold)
    const char* pe = getenv("LT_PINNED_READBACK");
    ctx->fold_piece_bytes = (!(pe && atoi(pe) == 0) && need >= (1u << 20)) ? ((need + 7) / 8 + 4095) / 4096 * 4096 : 0;
  }
  ctx->fold_pieced = false;
  rc = render_on_stream(ctx, desc, ctx->d_out, need, ctx->stream);
  if (rc == LT_OK) rc = enqueue_readback(ctx, need, out_host, staged);
  ctx->fold_piece_bytes = 0;
  return rc;
}

extern "C" int lt_hip_render(lt_hip_context* ctx, const lt_hip_render_desc* desc, float* out_host, uint64_t out_bytes) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t need = 0;
  bool staged = false;
  int rc = render_to_host(ctx, desc, out_host, out_bytes, need, staged);
  if (rc) return rc;
  rc = finish_readback(ctx, need, out_host, staged);
  if (rc) return rc;
  rc = finish_pending(ctx);
  if (rc) return rc;
  ctx->last.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return LT_OK;
}

// lt_hip_set_scene + lt_hip_render in one call, as the plugin's render() needs them (the reference hands over its scene on every
// call, renderer_opencl.cpp:107-120): when the buffers have the resident scene's sizes, the frame is rendered and read back ON
// THE ASSUMPTION that nothing changed while this thread hashes the buffers; the hash decides whether the frame stands.  It
// nearly always does -- and then the hash of 140 MB of scene has cost nothing, hidden behind the frame's 17 ms -- or the scene
// is uploaded and the frame rendered again.
extern "C" int lt_hip_render_scene(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims, uint64_t prim_bytes,
                                   const void* materials, uint64_t material_bytes, const void* lights, uint64_t light_bytes,
                                   const lt_hip_render_desc* desc, float* out_host, uint64_t out_bytes) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  if (!nodes || !prims || !materials || !lights) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "NULL scene buffer");
  try {
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t sizes[4] = {node_bytes, prim_bytes, material_bytes, light_bytes};
    const void* const bufs[4] = {nodes, prims, materials, lights};
    const bool continues = desc && desc->frame_count && desc->accumulate && desc->accumulate_base > 0;   // (reads the caller's buffer: no second try)
    // Rendering from the resident copy while the host hashes what came with the frame pays when the scene is the resident one --
    // a still scene, every call but the first.  A caller whose scene changed last time (an animation) gets the hash first: a
    // millisecond in front of the frame instead of a whole frame rendered for nothing.
    const bool speculate = ctx->has_scene && ctx->speculate_next && memcmp(sizes, ctx->scene_sizes, sizeof(sizes)) == 0 &&
                           !getenv("LT_SCENE_ALWAYS_UPLOAD") && !continues;
    uint64_t need = 0;
    bool staged = false;
    int rc;
    if (speculate) {
      rc = render_to_host(ctx, desc, out_host, out_bytes, need, staged);
      if (rc) return rc;
      const SceneHash hash = hash_scene(bufs, sizes);   // (while the GPU renders)
      rc = finish_readback(ctx, need, out_host, staged);
      if (rc) return rc;
      if (hash == ctx->scene_hash) {
        ctx->scene_reused++;
        rc = finish_pending(ctx);
        if (rc) return rc;
        ctx->last.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return LT_OK;
      }
      rc = finish_pending(ctx);
      if (rc) return rc;
      ctx->speculate_next = false;
      rc = set_scene_impl(ctx, nodes, node_bytes, prims, prim_bytes, materials, material_bytes, lights, light_bytes, &hash);
    } else {
      const uint32_t reused = ctx->scene_reused;
      rc = set_scene_impl(ctx, nodes, node_bytes, prims, prim_bytes, materials, material_bytes, lights, light_bytes, nullptr);
      ctx->speculate_next = rc == LT_OK && ctx->scene_reused != reused;   // (the resident scene again: the next frame may start at once)
    }
    if (rc) return rc;
    rc = lt_hip_render(ctx, desc, out_host, out_bytes);
    if (rc == LT_OK) ctx->last.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
  } catch (const std::exception& e) {
    return fail(ctx, LT_ERR_HIP, std::string("lt_hip_render_scene: ") + e.what());
  }
}

extern "C" int lt_hip_untile(lt_hip_context* ctx, const float* gathered, uint64_t floats_per_rank, uint32_t n_ranks, uint32_t width,
                             uint32_t height, uint32_t depth, uint32_t tile_w, uint32_t tile_h, float* image_out, void* hip_stream) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  if (!gathered || !image_out || !n_ranks || !width || !height || depth < 1 || !tile_w || !tile_h)
    return fail(ctx, LT_ERR_INVALID_ARGUMENT, "lt_hip_untile: bad argument");
  const uint32_t tilesX = (width + tile_w - 1) / tile_w, tilesY = (height + tile_h - 1) / tile_h;
  const uint64_t maxTilesPerRank = ((uint64_t)tilesX * tilesY + n_ranks - 1) / n_ranks;
  if (floats_per_rank < maxTilesPerRank * tile_w * tile_h * depth) return fail(ctx, LT_ERR_BUFFER_TOO_SMALL, "lt_hip_untile: floats_per_rank too small");
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const uint64_t n = (uint64_t)width * height;
  hipLaunchKernelGGL(lt_untile_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, gathered,
                     floats_per_rank, n_ranks, width, height, depth, tile_w, tile_h, tilesX, image_out);
  LT_HIP_CHECK(ctx, hipGetLastError());
  return LT_OK;
}

extern "C" int lt_hip_synchronize(lt_hip_context* ctx, void* hip_stream) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  LT_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)hip_stream));
  return LT_OK;
}

extern "C" int lt_hip_read_scene_structure(lt_hip_context* ctx, int what, void* out, uint64_t capacity, uint64_t* out_bytes) {
  if (!ctx) return LT_ERR_INVALID_ARGUMENT;
  if (!out_bytes || what < 0 || what > 3) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "lt_hip_read_scene_structure: bad arguments");
  if (!ctx->has_scene) return fail(ctx, LT_ERR_NO_SCENE, "no scene uploaded");
  LT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool own = ctx->d_nodes2 && ctx->d_wide && ctx->d_rank8;
  uint32_t info[4] = {(uint32_t)ctx->height2, (uint32_t)ctx->wide_height, ctx->n_wide, ctx->device_prepared ? 1u : 0u};
  const void* src = nullptr;
  uint64_t bytes = 0;
  if (what == 3) { bytes = sizeof(info); }
  else if (own && what == 0) { src = ctx->d_nodes2; bytes = (uint64_t)ctx->n_nodes2 * 32; }
  else if (own && what == 1) { src = ctx->d_rank8; bytes = (uint64_t)ctx->n_prims * 32; }
  else if (own && what == 2) { src = ctx->d_wide; bytes = ((uint64_t)ctx->n_wide + ctx->n_prims + 1) * 64 + 64; }
  *out_bytes = bytes;
  if (!out || bytes == 0) return LT_OK;
  if (capacity < bytes) return fail(ctx, LT_ERR_INVALID_ARGUMENT, "lt_hip_read_scene_structure: buffer too small");
  if (what == 3) { memcpy(out, info, sizeof(info)); return LT_OK; }
  LT_HIP_CHECK(ctx, hipDeviceSynchronize());
  LT_HIP_CHECK(ctx, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return LT_OK;
}

extern "C" int lt_hip_get_stats(lt_hip_context* ctx, lt_hip_stats* out) {
  if (!ctx || !out) return LT_ERR_INVALID_ARGUMENT;
  int rc = finish_pending(ctx);
  if (rc) return rc;
  *out = ctx->last;
  out->scene_uploads = ctx->scene_uploads;
  out->scene_reused = ctx->scene_reused;
  out->own_tree_height = ctx->d_rank8 ? ctx->height2 : -1;
  out->own_tree_ms = ctx->retree_ms;
  return LT_OK;
}
