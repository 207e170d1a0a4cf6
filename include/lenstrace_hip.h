/*
 * lenstrace_hip.h -- C ABI of the MI355X-native lens_trace ray-trace backend (liblenstrace-hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, POD structs, int status codes, no C++
 * types and no exceptions.  It carries exactly what the reference's renderer plugin does inside
 * Renderer::render() (all citations relative to the reference tree):
 *
 *   reference step                                                         replaced by
 *   ---------------------------------------------------------------------  -------------------------
 *   RendererOpenCL::RendererOpenCL(), device/context/queue                  lt_hip_create
 *     src/opencl/renderer_opencl.cpp:11-24
 *   ~RendererOpenCL()                          :26-33                       lt_hip_destroy
 *   compileKernel(path) + programMap lookup    :35-54, :67-70               lt_hip_program_from_path
 *     (a kernel *source path* selects the program; here its basename
 *      selects one of the five pre-compiled HIP programs)
 *   5 x clCreateBuffer + clEnqueueWriteBuffer  :107-120                     lt_hip_set_scene
 *     (node / ordered-primitive / material / light-container buffers;
 *      uploaded once and cached instead of on every render() call)
 *   the two together, per render() call                                    lt_hip_render_scene
 *   camera buffer upload + kernel args + NDRange launches + wait +          lt_hip_render
 *     blocking read-back into pOutputBuffer    :119-149                     (lt_hip_render_device keeps
 *                                                                            the pixels in HBM)
 *   examples/accumulator frame loop + accumulator.frag running mean         frame_count / accumulate
 *     examples/accumulator/src/main.cpp:296-325, shaders/accumulator.frag   fields of lt_hip_render_desc
 *   printf("Kernel Error: %d") / build log     :3-9, :142-144               lt_hip_last_error
 *
 * Launch semantics are the CUDA backend's (src/cuda/renderer_cuda.cpp:74-88): every pixel of the
 * W x H image is rendered exactly once, whatever W and H are (the OpenCL backend's truncating
 * work-block maths, renderer_opencl.cpp:90, is a launch-decomposition artefact the reference's own
 * CustomBlockSize test declares invisible).
 *
 * Buffer layouts are the reference's, verbatim:
 *   nodes     LinearBVHNode[M] 32 B  include/lens_trace/acceleration_structure_explicit.h:20-32
 *   prims     Primitive[N]     76 B  :34-42 (BVH order)
 *   materials Material[K]      32 B  include/lens_trace/model.h:26-31
 *   lights    LightContainer  260 B  acceleration_structure_explicit.h:44-47
 *   camera    7 x 4 B                src/camera.cpp:14-19 (frameCount uint in the 7th slot)
 *   output    float[H][W][depth], 3 floats written per pixel at (y*W+x)*depth
 */
#ifndef LENSTRACE_HIP_H
#define LENSTRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_HIP_ABI_VERSION 4

typedef struct lt_hip_context lt_hip_context;

/* status codes (0 = success) */
enum {
  LT_OK = 0,
  LT_ERR_INVALID_ARGUMENT = 1,
  LT_ERR_NO_DEVICE = 2,      /* no usable gfx950 device / HIP runtime failure at create */
  LT_ERR_HIP = 3,            /* a HIP call failed; text in lt_hip_last_error */
  LT_ERR_NO_SCENE = 4,       /* render before set_scene */
  LT_ERR_BAD_SCENE = 5,      /* index out of range, BVH deeper than the reference's 64-entry stack, ... */
  LT_ERR_BUFFER_TOO_SMALL = 6,
  LT_ERR_UNKNOWN_PROGRAM = 7
};

/* the kernel files the reference ships (five renderer programs + the custom_kernel example's), selected by kernelFilePath in RenderProperties* */
enum {
  LT_PROGRAM_BASIC = 0,                  /* resources/kernels/opencl/basic.cl */
  LT_PROGRAM_BASIC_LIGHTING = 1,         /* resources/kernels/opencl/basic_lighting.cl (25 blended samples) */
  LT_PROGRAM_ACCUMULATOR = 2,            /* examples/accumulator/resources/kernels/accumulator.cl */
  LT_PROGRAM_GLOBAL_ILLUMINATION = 3,    /* examples/global_illumination/resources/kernels/global_illumination.cl */
  LT_PROGRAM_GLOBAL_ILLUMINATION_25 = 4, /* resources/kernels/opencl/global_illumination.cl (25 blended samples) */
  LT_PROGRAM_CUSTOM_OPENCL = 5           /* examples/custom_kernel/resources/kernels/custom_opencl.cl (barycentrics as colour) */
};

/* KernelMode of include/lens_trace/structures.h:20-23.  Pixels do not depend on it, except that the
 * lighting programs' linearKernel clamps to [0,1] and their tileKernel does not (accumulator.cl:316-318
 * vs :356-358). */
enum { LT_KERNEL_MODE_LINEAR = 0, LT_KERNEL_MODE_TILE = 1 };

enum {
  LT_RENDER_FLAG_STATS = 1u,    /* count rays / node visits / triangle tests with device atomics (slower) */
  LT_RENDER_FLAG_PIXEL_COUNTERS = 2u,/* diagnostic (implies STATS, needs depth >= 4): instead of the colour, write each
                                        pixel's own {rays, shadow rays, node visits, triangle tests} as 4 floats */
  LT_RENDER_FLAG_DEVICE_LIBM = 4u,   /* accepted and ignored (ABI 2 name of what became LT_RENDER_FLAG_STRICT_MATH) */
  /* Floating-point flavour ("Floating-point model"): what the reference leaves to its OpenCL implementation.
   * DEFAULT (no flag): the reference's kernel files exactly as RendererOpenCL builds them on this GPU -- clBuildProgram with
   *   NULL options (src/opencl/renderer_opencl.cpp:50): `a*b + c` inside one source expression is one fused multiply-add,
   *   float divide / sqrt have the OpenCL default accuracy (v_rcp_f32 / v_sqrt_f32 based), the builtins (dot, cross,
   *   normalize, distance, sin, cos, clamp) are ROCm's OpenCL device library's.  Output is bit-identical to those kernels
   *   (tests/test_gpu_reference_kernels.py, test_gpu_full_size.py, test_gpu_configs.py).
   * LT_RENDER_FLAG_STRICT_MATH: the same kernels built with -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt: one IEEE
   *   operation per source operation outside the builtins.  Bit-identical to that build.
   * LT_RENDER_FLAG_PORTABLE_MATH: strict, and the four builtin leaf functions where the device library uses hardware
   *   approximations (rsqrt, sqrt, sinf / cosf, clamp) in correctly rounded forms every IEEE machine reproduces: what the CPU
   *   oracle computes.  Differs from STRICT by <= 1-2 ulp in those functions, which the stochastic programs' random() can
   *   amplify into a flipped ray decision in ~0.03 % of pixels.
   * PORTABLE and STRICT exclude each other. */
  LT_RENDER_FLAG_PORTABLE_MATH = 8u,
  LT_RENDER_FLAG_STRICT_MATH = 16u,
  /* The first accumulator / basic_lighting call of a (scene, image geometry, frames per launch) runs its launch once per
   * shadow-ray walk (4-5 extra launches inside that call, a host-side wait for their times, all of it counted in that call's
   * kernel_ms / kernel_launches) and keeps the fastest.  With this flag a call that has no verdict yet times nothing: it uses
   * the scene's most recent verdict for the program, or any-hit packets.  Pixels do not depend on the walk. */
  LT_RENDER_FLAG_NO_WALK_TIMING = 32u
};

typedef struct lt_hip_render_desc {
  uint32_t struct_size;         /* sizeof(lt_hip_render_desc), for ABI growth */
  int32_t program;              /* LT_PROGRAM_* */
  int32_t kernel_mode;          /* LT_KERNEL_MODE_* */
  uint32_t width, height, depth;/* imageDimensions[3]; depth >= 3 */
  uint8_t camera[28];           /* Camera::getCameraBuffer() */
  /* Progressive rendering (examples/accumulator/src/main.cpp:296-325).  frame_count == 0: one frame with
   * the camera buffer's own frameCount, plain overwrite -- exactly Renderer::render().  frame_count > 0:
   * frames with frameCount = frame_first .. frame_first+frame_count-1; with accumulate != 0 they are folded
   * into the output by accumulator.frag's running mean acc = (c + acc*n)/(n+1), n = accumulate_base,
   * accumulate_base+1, ... (n == 0 replaces); with accumulate == 0 the last frame wins.
   * An accumulating call renders all its frames in one kernel launch (up to LT_FUSED_BYTES of scratch memory,
   * default 16 GiB, per launch) and folds them in frame order: same floats as frame_count calls of one frame,
   * without their per-launch cost -- pass as many frames per call as the application allows. */
  uint32_t frame_first;
  uint32_t frame_count;
  uint32_t accumulate;
  uint32_t accumulate_base;
  /* Image-tile sharding (one process per GPU).  The image is cut into tile_w x tile_h tiles, row-major,
   * edge tiles clipped; this call renders tiles tile_first, tile_first+tile_stride, ... and stores tile k of
   * the call contiguously at float offset k*tile_w*tile_h*depth, row pitch tile_w (pixels of clipped tiles
   * outside the image are not written).  tile_w == 0 means "whole image as one tile", which makes the
   * output layout the reference's (y*W+x)*depth. */
  uint32_t tile_w, tile_h;
  uint32_t tile_first, tile_stride;
  int32_t gi_max_depth;         /* 0 = the reference constant 16 (global_illumination.cl:307) */
  uint32_t flags;               /* LT_RENDER_FLAG_* */
} lt_hip_render_desc;

typedef struct lt_hip_stats {
  uint64_t rays;                /* calls of intersect + intersectIgnorePrimitiveIndex (valid with FLAG_STATS) */
  uint64_t shadow_rays;
  uint64_t node_visits;         /* intersectBounds calls */
  uint64_t tri_tests;           /* intersectTriangle calls */
  uint64_t pixels;              /* pixels rendered by the last call (per frame) */
  uint32_t frames;              /* frames rendered by the last call */
  uint32_t kernel_launches;
  float kernel_ms;              /* HIP-event time over the kernels of the last call, on the call's stream */
  float total_ms;               /* lt_hip_render only: upload + kernels + read-back wall time */
  float render_ms;              /* kernel_ms without the running-mean kernels that follow fused multi-sample launches */
  int32_t shadow_packets;       /* how the last call walked its shadow rays: 1 = any-hit packets, 0 = per lane, 2 = chosen per wavefront,
                                 * 3 = queued and walked by a kernel of their own whose lanes take a new ray when theirs is done
                                 * (accumulator; picked per scene, program, image geometry and frames per launch by timing the
                                 * candidates once; LT_SHADOW_PACKETS=0/1/2/3 forces), -1 = not timed yet */
  uint32_t scene_uploads;       /* lt_hip_set_scene calls of this context that uploaded ... */
  uint32_t scene_reused;        /* ... and those that found the resident scene's content unchanged (full hash) and kept it */
  int32_t own_tree_height;      /* height of the backend's own hierarchy over the scene's leaves (built at lt_hip_set_scene, walked by
                                 * every finite ray of the non-counting kernels; LT_RETREE=0 keeps the caller's splits), -1 = none: every walk uses the caller's tree */
  float own_tree_ms;            /* time of its preparation inside lt_hip_set_scene (device or host) */
} lt_hip_stats;

int lt_hip_abi_version(void);

/* device_index: HIP ordinal (one context per GPU, one process per GPU under torch.distributed). */
int lt_hip_create(int device_index, lt_hip_context** out_ctx);
int lt_hip_destroy(lt_hip_context* ctx);

/* Text of the last error on ctx (or of the last failed lt_hip_create when ctx == NULL). Never NULL. */
const char* lt_hip_last_error(const lt_hip_context* ctx);

/* Maps RenderProperties*::kernelFilePath to LT_PROGRAM_* by basename ("…/accumulator.cl" -> ACCUMULATOR).
 * "global_illumination" resolves to the 25-sample program when the path contains "resources/kernels/opencl/"
 * and does not contain "examples/", as in the reference tree. */
int lt_hip_program_from_path(const char* kernel_file_path, int* out_program);

/* program ids >= LT_PROGRAM_USER_BASE are user programs of one context, handed out by lt_hip_resolve_program */
#define LT_PROGRAM_USER_BASE 1000

/* The reference's "kernelFilePath names a source file that is compiled at first use and cached by path"
 * (renderer_opencl.cpp:35-54, :67-70): a path whose basename is a built-in program resolves like
 * lt_hip_program_from_path; any other path ending in ".hip" is read as a USER PROGRAM -- a HIP source file defining
 *     template <class CFG> __device__ lt::V3 lt::user_shade(const SceneDev&, const Ray& cameraRay, float filmX, float filmY,
 *                                                            uint32_t frameCount, Stack<CFG::kDeep>&, Counters&);
 * (the shade step; traversal, camera rays, framebuffer and scheduling are the built-in ones, lens_trace_amd/csrc/lt_kernel.hpp)
 * -- compiled for gfx950 with hipRTC (-O3 -ffp-contract=off), cached by path for the life of the context.  Compile errors
 * are reported through lt_hip_last_error (LT_ERR_UNKNOWN_PROGRAM). */
int lt_hip_resolve_program(lt_hip_context* ctx, const char* kernel_file_path, int* out_program);

/* Uploads (host pointers) and validates the four scene buffers; keeps them resident until the next
 * set_scene / destroy.  A call whose four buffers have the sizes and the content (a hash of every byte) of the resident
 * scene returns at once and keeps it: callers may pass their scene on every render, as the reference does
 * (renderer_opencl.cpp:107-120), and in-place edits are honoured.
 * Derived at upload: the traversal-side triangle array (48-byte stride: A, B-A, C-A) and -- when every node's box encloses
 * its children's, which the reference's own builder guarantees -- the backend's OWN hierarchy over the caller's leaves
 * (binned surface-area heuristic, built by kernels -- lens_trace_amd/csrc/lt_prep.hip: ~6.5 ms for the whole call on a million
 * triangles -- or, for small or unusual buffers, by host threads: lt_retree.hpp), the 64-byte records of its packet
 * walks, the 4-wide groups of quantised boxes and the leaf records of its per-lane walks and the reference's leaf order per
 * direction-sign octant.  The caller's LinearBVHNode array stays resident and is
 * what the counting kernels (LT_RENDER_FLAG_STATS / _PIXEL_COUNTERS), rays with a non-finite component and scenes whose
 * boxes do not nest walk, in the reference's order.  Pixels do not depend on which hierarchy a ray walked
 * (lens_trace_amd/csrc/lt_retree.hpp).  LT_RETREE=0 keeps the caller's splits. */
int lt_hip_set_scene(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims,
                     uint64_t prim_bytes, const void* materials, uint64_t material_bytes, const void* lights,
                     uint64_t light_bytes);

/* Number of floats the output of `desc` needs (whole image: W*H*depth; tiled: tiles_of_call*tile_w*tile_h*depth). */
int lt_hip_output_floats(const lt_hip_render_desc* desc, uint64_t* out_floats);

/* Reference semantics: synchronous, fills caller-owned HOST memory (pOutputBuffer). */
int lt_hip_render(lt_hip_context* ctx, const lt_hip_render_desc* desc, float* out_host, uint64_t out_bytes);

/* lt_hip_set_scene followed by lt_hip_render, as one call -- what a plugin's render() does with a caller that hands over its
 * scene every time (renderer_opencl.cpp:107-120).  Same results and same statuses as the two calls; when the four buffers have
 * the resident scene's sizes the frame is rendered while the host hashes them, and rendered again only if they turn out to have
 * changed -- so that an unchanged scene costs no hashing time on top of the frame.  After a call that found the scene changed
 * (an animation) the next call hashes first: a millisecond in front of the frame instead of a frame rendered for nothing. */
int lt_hip_render_scene(lt_hip_context* ctx, const void* nodes, uint64_t node_bytes, const void* prims, uint64_t prim_bytes,
                        const void* materials, uint64_t material_bytes, const void* lights, uint64_t light_bytes,
                        const lt_hip_render_desc* desc, float* out_host, uint64_t out_bytes);

/* Same, but the pixels stay in HBM: out_device is device memory of the context's GPU, the work is enqueued
 * on hip_stream (a hipStream_t, NULL = default stream) and NOT waited for. */
int lt_hip_render_device(lt_hip_context* ctx, const lt_hip_render_desc* desc, float* out_device,
                         uint64_t out_bytes, void* hip_stream);

/* Scatters a gathered stack of per-rank tile buffers (rank r holds tiles r, r+n_ranks, ...) into the
 * reference's row-major image.  All pointers are device memory; runs on hip_stream. */
int lt_hip_untile(lt_hip_context* ctx, const float* gathered, uint64_t floats_per_rank, uint32_t n_ranks,
                  uint32_t width, uint32_t height, uint32_t depth, uint32_t tile_w, uint32_t tile_h,
                  float* image_out, void* hip_stream);

int lt_hip_synchronize(lt_hip_context* ctx, void* hip_stream);

/* Host only, no context, no GPU: the hierarchy lt_hip_set_scene would build over the leaves of `nodes` (LinearBVHNode array,
 * node_bytes = 32 * count), written to out_nodes (capacity out_bytes; 32 bytes x (2 x leaves - 1)) in the caller's own layout
 * and pre-order numbering, and, when rank8 != NULL, the reference's leaf order (8 uint32 per primitive, n_prims primitives:
 * position of the primitive's leaf in the reference's depth-first order for each direction-sign octant).  Returns the
 * hierarchy's height, or -1 when the scene gets none (boxes that do not nest, fewer than two leaves, bounds >= 2^40).
 * height_slack: levels allowed above ceil(log2 leaves) (lt_hip_set_scene uses 2); < 0 copies the caller's splits. */
int lt_hip_own_hierarchy(const void* nodes, uint64_t node_bytes, int height_slack, void* out_nodes, uint64_t out_bytes,
                         uint32_t* rank8, uint32_t n_prims);

/* Host only, no context, no GPU: what lt_hip_set_scene makes of that hierarchy (own_nodes: the array lt_hip_own_hierarchy wrote)
 * for its per-lane walks -- the tree collapsed into 4-wide groups, numbered depth first.  out_slots receives four 16-byte child
 * slots per group: the child's box on a 16-bit grid over the scene's bounds, rounded outwards, as six uint16 (lo.x lo.y lo.z hi.x
 * hi.y hi.z), then one uint32 link: the child's own group, or 0x80000000 | (groups + primitive offset) for a leaf (the index of its
 * 64-byte record behind the groups); an empty slot holds lo = 65535, hi = 0 and the link 0x80000000 | (groups + n_prims).
 * origin_step receives the grid: origin xyz, step xyz (bound = origin + q * step); *out_groups the number of groups (at most
 * leaves - 1: size out_slots for that).  Returns the height of the group tree, or -1 (bad arguments, two leaves on one
 * primitive, a bound off the grid).  The same arithmetic runs on the GPU at upload (lens_trace_amd/csrc/lt_own16.hpp). */
int lt_hip_own_wide(const void* own_nodes, uint64_t node_bytes, uint32_t n_prims, float* origin_step, void* out_slots, uint64_t out_bytes,
                    uint32_t* out_groups);

/* Diagnostics: copies one of the structures lt_hip_set_scene derived for the resident scene back to the host (tests hold the
 * device-side preparation, lens_trace_amd/csrc/lt_prep.hip, against the host-side one, lt_retree.hpp, with it).
 * what: 0 = the own tree (32-byte nodes, pre-order), 1 = the leaf order table (8 x uint32 per primitive), 2 = the per-lane
 * walks' array (64-byte grid header, then 64-byte records: groups, leaf records by primitive offset, the sentinel),
 * 3 = four uint32: own-tree height, group-tree height, groups, 1 if the device prepared the scene (0: the host).
 * out == NULL: only *out_bytes (the size) is set.  LT_ERR_NO_SCENE without a scene; *out_bytes = 0 when the scene has no such
 * structure (it then walks the caller's tree). */
int lt_hip_read_scene_structure(lt_hip_context* ctx, int what, void* out, uint64_t capacity, uint64_t* out_bytes);

/* Statistics of the most recent render call on ctx (waits for it to finish). */
int lt_hip_get_stats(lt_hip_context* ctx, lt_hip_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* LENSTRACE_HIP_H */
