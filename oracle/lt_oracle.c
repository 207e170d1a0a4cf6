/*
 * lt_oracle.c -- CPU restatement of the lens_trace ray-trace hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lens_trace_amd/ (the product) may
 * include, link, import or execute this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * What it restates (all paths relative to /root/reference):
 *   resources/kernels/opencl/basic.cl                               (program BASIC)
 *   resources/kernels/opencl/basic_lighting.cl                      (program BASIC_LIGHTING)
 *   examples/accumulator/resources/kernels/accumulator.cl           (program ACCUMULATOR)
 *   examples/global_illumination/resources/kernels/global_illumination.cl   (program GI)
 *   resources/kernels/opencl/global_illumination.cl                 (program GI25)
 *   examples/custom_kernel/resources/kernels/custom_opencl.cl       (program CUSTOM: basic.cl without the lens code, shade
 *                                                                    returns the barycentrics as colour, :232-243)
 *   examples/accumulator/resources/shaders/accumulator.frag:10-20   (running mean)
 *
 * Floating-point model ("strict OpenCL C on ROCm"):  user-level expressions are
 * evaluated operation by operation with no FMA contraction, IEEE division and
 * square root, and the float/double islands exactly where OpenCL C's literal
 * typing puts them (1.0, 0.0001, M_PI are double: cl_khr_fp64 is enabled at the
 * top of every kernel file).  The OpenCL *builtins* (dot, cross, normalize,
 * distance, clamp) follow the definitions ROCm 7.2's OpenCL device library
 * gives them on gfx950 -- read from the LLVM IR of the reference kernels
 * compiled with `clang -x cl -target amdgcn-amd-amdhsa -mcpu=gfx950
 * -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt`:
 *     dot(a,b)   = fma(a.w,b.w, fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)))
 *     cross(a,b) = (fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)), fma(a.x,b.y,-(a.y*b.x)), 0)
 *     normalize  = p * rsqrt(dot(p,p))  (with the library's zero/denormal/inf guards)
 *     distance   = sqrt(dot(d,d))       (same guards)
 * The three places where that library uses a hardware approximation or its own
 * libm are made portable so that CPU and GPU can agree bit for bit:
 *     rsqrt(x)      := (float)(1.0 / sqrt((double)x))     [device: v_rsq_f32, 1 ulp]
 *     sqrt in distance := correctly rounded sqrtf         [device: v_sqrt_f32, 1 ulp]
 *     cos/sin(float):= (float)cos/sin((double)x)          [device: ocml sinf/cosf]
 * double sin() and fmod() come from the platform libm on both sides (fmod is
 * exact; two <=1-ulp double sines disagree after the float rounding in
 * random() with probability ~1e-10 per call).
 *
 * Parity pinning: the reference's own tests hold known answers only for
 * `basic` on green_wall.obj (tests/opencl_renderer_test.cc:51-228:
 * CorrectColor, KernelMode, CustomBlockSize); tests/test_oracle.py checks all
 * three against this file on buffers dumped from the reference's own host
 * classes (oracle/_ref/ref_host_dump).  For accumulator / basic_lighting /
 * global_illumination / lens the reference holds no fixture: on CPU those are
 * "parity unpinned"; on the GPU box they are cross-checked against the
 * reference's own .cl compiled for gfx950 (oracle/_ref/ *.co, see
 * oracle/build_ref.sh and DESIGN.md section 3).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- buffer layouts (acceleration_structure_explicit.h:20-47, model.h:26-31,
 *      camera.cpp:14-19; device copies basic.cl:3-41) ---- */
typedef struct {
  float boundsMin[3];
  float boundsMax[3];
  int32_t offset;          /* union { primitivesOffset; secondChildOffset } */
  uint16_t primitiveCount; /* 0 = interior */
  uint8_t axis;
  uint8_t pad;
} Node;                    /* 32 B */

typedef struct {
  float positionA[3], positionB[3], positionC[3];
  float normalA[3], normalB[3], normalC[3];
  int32_t materialIndex;
} Prim;                    /* 76 B */

typedef struct {
  float diffuse[3];
  float ior;
  float dissolve;
  float emission[3];
} Mat;                     /* 32 B */

typedef struct {
  uint32_t count;
  uint32_t primitives[64];
} Lights;                  /* 260 B */

typedef struct {
  float position[3];
  float yaw, pitch, roll;
  uint32_t frameCount;
} Cam;                     /* 28 B */

typedef char check_node[(sizeof(Node) == 32) ? 1 : -1];
typedef char check_prim[(sizeof(Prim) == 76) ? 1 : -1];
typedef char check_light[(sizeof(Lights) == 260) ? 1 : -1];
typedef char check_cam[(sizeof(Cam) == 28) ? 1 : -1];

enum { LT_BASIC = 0, LT_BASIC_LIGHTING = 1, LT_ACCUMULATOR = 2, LT_GI = 3, LT_GI25 = 4, LT_CUSTOM = 5 };
enum { LT_MODE_LINEAR = 0, LT_MODE_TILE = 1 };

typedef struct {
  uint64_t rays;        /* calls of intersect + intersectIgnorePrimitiveIndex */
  uint64_t shadow_rays; /* subset of rays that are shadow rays */
  uint64_t node_visits; /* calls of intersectBounds */
  uint64_t tri_tests;   /* calls of intersectTriangle */
  uint64_t max_stack;   /* deepest nodesToVisit use */
} lt_oracle_stats;

typedef struct { float x, y, z, w; } f4;
typedef struct { float x, y, z; } f3;
typedef struct { f4 origin, direction; } Ray;
typedef struct { int primitiveIndex; int hitType; float t, u, v; } Payload;

typedef struct {
  const Node* nodes;
  const Prim* prims;
  const Mat* mats;
  const Lights* lights;
  int gi_max_depth;     /* reference constant 16 (global_illumination.cl(ex):307) */
  int error;            /* set to 1 on traversal stack overflow (>64, UB in the reference) */
  lt_oracle_stats st;
} Ctx;

/* ---- builtins, ROCm OpenCL device-library definitions (see header) ---- */
static inline f4 mk4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
static inline f4 add4(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 sub4(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline f4 scale4(float s, f4 a) { return mk4(s * a.x, s * a.y, s * a.z, s * a.w); }
static inline f4 neg4(f4 a) { return mk4(-a.x, -a.y, -a.z, -a.w); }

static inline float dot4(f4 a, f4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
static inline float dot2(float ax, float ay, float bx, float by) { return fmaf(ay, by, ax * bx); }
static inline f4 cross4(f4 a, f4 b) {
  return mk4(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)), 0.0f);
}
static inline float rsqrt_portable(float x) { return (float)(1.0 / sqrt((double)x)); }
static inline float cosf_portable(float x) { return (float)cos((double)x); }
static inline float sinf_portable(float x) { return (float)sin((double)x); }

static f4 normalize4(f4 p) {
  if (p.x == 0.0f && p.y == 0.0f && p.z == 0.0f && p.w == 0.0f) return p;
  float l2 = dot4(p, p);
  if (l2 < FLT_MIN) {
    p = scale4(0x1p+86f, p);
    l2 = dot4(p, p);
  } else if (l2 == INFINITY) {
    p = scale4(0x1p-66f, p);
    l2 = dot4(p, p);
    if (l2 == INFINITY) {
      p = mk4(copysignf(isinf(p.x) ? 1.0f : 0.0f, p.x), copysignf(isinf(p.y) ? 1.0f : 0.0f, p.y),
              copysignf(isinf(p.z) ? 1.0f : 0.0f, p.z), copysignf(isinf(p.w) ? 1.0f : 0.0f, p.w));
      l2 = dot4(p, p);
    }
  }
  return scale4(rsqrt_portable(l2), p);
}

static float distance4(f4 a, f4 b) {
  f4 d = sub4(a, b);
  float l2 = dot4(d, d);
  if (l2 < FLT_MIN) {
    d = scale4(0x1p+86f, d);
    return sqrtf(dot4(d, d)) * 0x1p-86f;
  } else if (l2 == INFINITY) {
    d = scale4(0x1p-66f, d);
    return sqrtf(dot4(d, d)) * 0x1p+66f;
  }
  return sqrtf(l2);
}

static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* accumulator.cl:63-66 -- float dot, then double add/fmod/sin/mul, float fract */
static float random_(float uvx, float uvy, float seed) {
  float d = dot2(uvx, uvy, 12.9898f, 78.233f);
  double x = (double)d + 1113.1 * (double)seed;
  float a = (float)(sin(fmod(x, M_PI)) * 43758.5453);
  return a - floorf(a);
}

/* accumulator.cl:72-111 (eps 0.0001 double), basic_lighting.cl:4 (1e-7 double),
 * basic.cl:77-117 (const float EPSILON = 0.0000001 -> float compare) */
static int intersectTriangle(Ctx* c, int program, Payload* pl, Ray ray, const Prim* pr) {
  c->st.tri_tests++;
  f4 A = mk4(pr->positionA[0], pr->positionA[1], pr->positionA[2], 1.0f);
  f4 B = mk4(pr->positionB[0], pr->positionB[1], pr->positionB[2], 1.0f);
  f4 C = mk4(pr->positionC[0], pr->positionC[1], pr->positionC[2], 1.0f);
  f4 v0v1 = sub4(B, A);
  f4 v0v2 = sub4(C, A);
  f4 pvec = cross4(ray.direction, v0v2);
  float det = dot4(v0v1, pvec);
  if (program == LT_BASIC || program == LT_CUSTOM) {
    const float EPSILON = 0.0000001;
    if (fabsf(det) < EPSILON) return 0;
  } else if (program == LT_BASIC_LIGHTING) {
    if ((double)fabsf(det) < 0.0000001) return 0;
  } else {
    if ((double)fabsf(det) < 0.0001) return 0;
  }
  float invDet = 1 / det;
  f4 tvec = sub4(ray.origin, A);
  float u = dot4(tvec, pvec) * invDet;
  if (u < 0 || u > 1) return 0;
  f4 qvec = cross4(tvec, v0v1);
  float v = dot4(ray.direction, qvec) * invDet;
  if (v < 0 || u + v > 1) return 0;
  float t = dot4(v0v2, qvec) * invDet;
  if (t < pl->t) { /* no t > 0 test */
    pl->t = t;
    pl->u = u;
    pl->v = v;
    return 1;
  }
  return 0;
}

/* accumulator.cl:113-130 */
static int intersectBounds(Ctx* c, Ray ray, f4 invDir, const int dirIsNeg[3], const Node* n) {
  c->st.node_visits++;
  const float* lo = n->boundsMin;
  const float* hi = n->boundsMax;
#define BND(neg, i) (((neg) == 0) ? lo[i] : hi[i])
  float tMin = (BND(dirIsNeg[0], 0) - ray.origin.x) * invDir.x;
  float tMax = (BND(1 - dirIsNeg[0], 0) - ray.origin.x) * invDir.x;
  float tyMin = (BND(dirIsNeg[1], 1) - ray.origin.y) * invDir.y;
  float tyMax = (BND(1 - dirIsNeg[1], 1) - ray.origin.y) * invDir.y;
  if (tMin > tyMax || tyMin > tMax) return 0;
  if (tyMin > tMin) tMin = tyMin;
  if (tyMax < tMax) tMax = tyMax;
  float tzMin = (BND(dirIsNeg[2], 2) - ray.origin.z) * invDir.z;
  float tzMax = (BND(1 - dirIsNeg[2], 2) - ray.origin.z) * invDir.z;
#undef BND
  if (tMin > tzMax || tzMin > tMax) return 0;
  if (tzMin > tMin) tMin = tzMin;
  if (tzMax < tMax) tMax = tzMax;
  return tMax > 0;
}

/* accumulator.cl:132-171 (ignore < 0: intersect) and :173-217
 * (intersectIgnorePrimitiveIndex).  The leaf loop tests primitives[offset]
 * primitiveCount times, never offset+i: restated as written. */
static void traverse(Ctx* c, int program, Payload* pl, Ray ray, int useIgnore, int ignore) {
  c->st.rays++;
  /* 1.0 / x is a double divide narrowed to float: equal to the float divide */
  f4 invDir = mk4((float)(1.0 / (double)ray.direction.x), (float)(1.0 / (double)ray.direction.y),
                  (float)(1.0 / (double)ray.direction.z), 0.0f);
  int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
  int toVisitOffset = 0, cur = 0;
  int nodesToVisit[64];
  for (;;) {
    const Node* node = &c->nodes[cur];
    if (intersectBounds(c, ray, invDir, dirIsNeg, node)) {
      if (node->primitiveCount > 0) {
        for (int i = 0; i < node->primitiveCount; i++) {
          if ((!useIgnore || node->offset != ignore) &&
              intersectTriangle(c, program, pl, ray, &c->prims[node->offset])) {
            pl->primitiveIndex = node->offset;
            pl->hitType = 1;
          }
        }
        if (toVisitOffset == 0) break;
        cur = nodesToVisit[--toVisitOffset];
      } else {
        if (toVisitOffset >= 64) { c->error = 1; return; }
        if (dirIsNeg[node->axis]) {
          nodesToVisit[toVisitOffset++] = cur + 1;
          cur = node->offset;
        } else {
          nodesToVisit[toVisitOffset++] = node->offset;
          cur = cur + 1;
        }
        if ((uint64_t)toVisitOffset > c->st.max_stack) c->st.max_stack = toVisitOffset;
      }
    } else {
      if (toVisitOffset == 0) break;
      cur = nodesToVisit[--toVisitOffset];
    }
  }
}

static inline Payload payload_init(float t) { Payload p = {0, 0, t, 0, 0}; return p; }

/* float3 a*bx + b*by + c*bz, evaluated (a*bx + b*by) + c*bz (accumulator.cl:247) */
static inline f3 bary3(const float* a, const float* b, const float* cc, f3 w) {
  f3 r;
  r.x = (a[0] * w.x + b[0] * w.y) + cc[0] * w.z;
  r.y = (a[1] * w.x + b[1] * w.y) + cc[1] * w.z;
  r.z = (a[2] * w.x + b[2] * w.y) + cc[2] * w.z;
  return r;
}
/* (float3)(1.0 - u - v, u, v): the first component is computed in double */
static inline f3 barycentrics(float u, float v) {
  f3 r = {(float)((1.0 - (double)u) - (double)v), u, v};
  return r;
}

static int is_light(const Lights* L, int primitiveIndex) {
  int hit = 0;
  for (uint32_t x = 0; x < L->count; x++)
    if ((uint32_t)primitiveIndex == L->primitives[x % 64]) hit = 1;
  return hit;
}

/* light triangle for index int(random * count); index == count reads
 * the zero-initialised tail of LightContainer.primitives */
static const Prim* light_prim(const Ctx* c, float rnd) {
  int idx = (int)(rnd * (float)c->lights->count);
  uint32_t p = (idx >= 0 && idx < 64) ? c->lights->primitives[idx] : 0;
  return &c->prims[p];
}

/* ---- basic.cl:65-71 ---- */
static f4 refract_(f4 I, f4 N, float firstIOR, float secondIOR) {
  float n = firstIOR / secondIOR;
  float cosI = -dot4(N, I);
  float sinT2 = (float)((double)(n * n) * (1.0 - (double)(cosI * cosI)));
  float cosT = (float)sqrt(1.0 - (double)sinT2);
  return add4(scale4(n, I), scale4(n * cosI - cosT, N));
}

/* basic.cl:225-277 */
static void traceRayThroughLens(Ctx* c, Payload* pl, Ray* ray) {
  const Prim* pr = &c->prims[pl->primitiveIndex];
  const Mat* m = &c->mats[pr->materialIndex];
  f3 b = barycentrics(pl->u, pl->v);
  f3 p3 = bary3(pr->positionA, pr->positionB, pr->positionC, b);
  f4 position = mk4(p3.x, p3.y, p3.z, 1.0f);
  f3 n3 = bary3(pr->normalA, pr->normalB, pr->normalC, b);
  f4 normal = mk4(n3.x, n3.y, n3.z, 0.0f);
  f4 tdir = refract_(ray->direction, normal, 1.0f, m->ior);

  Payload pl2 = payload_init(FLT_MAX);
  Ray ray2 = {position, tdir};
  traverse(c, LT_BASIC, &pl2, ray2, 1, pl->primitiveIndex);

  pr = &c->prims[pl2.primitiveIndex];
  m = &c->mats[pr->materialIndex];
  b = barycentrics(pl2.u, pl2.v);
  p3 = bary3(pr->positionA, pr->positionB, pr->positionC, b);
  position = mk4(p3.x, p3.y, p3.z, 1.0f);
  n3 = bary3(pr->normalA, pr->normalB, pr->normalC, b);
  normal = mk4(n3.x, n3.y, n3.z, 0.0f);
  tdir = refract_(tdir, neg4(normal), m->ior, 1.0f);

  *pl = payload_init(FLT_MAX);
  ray->origin = position;
  ray->direction = tdir;
  traverse(c, LT_BASIC, pl, *ray, 1, pl2.primitiveIndex);
}

/* basic.cl:279-307 */
static f3 shade_basic(Ctx* c, Ray cameraRay) {
  f3 out = {0, 0, 0};
  Payload pl = payload_init(FLT_MAX);
  Ray ray = cameraRay;
  traverse(c, LT_BASIC, &pl, ray, 0, 0);
  if (pl.hitType == 1) {
    const Prim* pr = &c->prims[pl.primitiveIndex];
    const Mat* m = &c->mats[pr->materialIndex];
    if ((double)m->dissolve < 1.0) {
      traceRayThroughLens(c, &pl, &ray);
      if (pl.hitType == 1) {
        pr = &c->prims[pl.primitiveIndex];
        m = &c->mats[pr->materialIndex];
      }
    }
    out.x = m->diffuse[0]; out.y = m->diffuse[1]; out.z = m->diffuse[2];
  }
  return out;
}

/* custom_opencl.cl:226-246: colour = (u, v, 1.0 - u - v), the last component computed in double */
static f3 shade_custom(Ctx* c, Ray cameraRay) {
  f3 out = {0, 0, 0};
  Payload pl = payload_init(FLT_MAX);
  traverse(c, LT_CUSTOM, &pl, cameraRay, 0, 0);
  if (pl.hitType == 1) {
    out.x = pl.u;
    out.y = pl.v;
    out.z = (float)((1.0 - (double)pl.u) - (double)pl.v);
  }
  return out;
}

/* light sample + shadow ray shared by accumulator.cl:239-279, basic_lighting.cl:234-274
 * and the direct/extension terms of global_illumination.cl(ex):267-297,:323-349.
 * normal_w is 0 in accumulator/basic_lighting and 1 in GI.
 * Returns 1 when the light sample is unoccluded; *ndotl = dot(toLight, normal). */
static int direct_light(Ctx* c, int program, const Prim* pr, int primIndex, float u, float v,
                        float fx, float fy, float seedIndex, float seedU, float seedV, float normal_w,
                        f4* position_out, f4* normal_out, float* ndotl) {
  f3 b = barycentrics(u, v);
  f3 p3 = bary3(pr->positionA, pr->positionB, pr->positionC, b);
  f4 position = mk4(p3.x, p3.y, p3.z, 1.0f);
  f3 n3 = bary3(pr->normalA, pr->normalB, pr->normalC, b);
  f4 normal = mk4(n3.x, n3.y, n3.z, normal_w);

  const Prim* lp = light_prim(c, random_(fx, fy, seedIndex));
  float uvx = random_(fx, fy, seedU);
  float uvy = random_(fx, fy, seedV);
  if (uvx + uvy > 1.0f) {
    uvx = 1.0f - uvx;
    uvy = 1.0f - uvy;
  }
  f3 lb = barycentrics(uvx, uvy);
  f3 l3 = bary3(lp->positionA, lp->positionB, lp->positionC, lb);
  f4 lightPosition = mk4(l3.x, l3.y, l3.z, 1.0f);

  f4 toLight = normalize4(sub4(lightPosition, position));
  Payload spl = payload_init((float)((double)distance4(position, lightPosition) - 0.01));
  Ray shadowRay = {position, toLight};
  c->st.shadow_rays++;
  traverse(c, program, &spl, shadowRay, 1, primIndex);

  *position_out = position;
  *normal_out = normal;
  *ndotl = dot4(toLight, normal);
  return spl.hitType == 0;
}

/* accumulator.cl:219-282 / basic_lighting.cl:220-277 */
static f3 shade_lighting(Ctx* c, int program, Ray cameraRay, float fx, float fy, uint32_t sampleIndex) {
  f3 out = {0, 0, 0};
  Payload pl = payload_init(FLT_MAX);
  traverse(c, program, &pl, cameraRay, 0, 0);
  if (program == LT_ACCUMULATOR) { /* accumulator.cl:233-237; no hitType check */
    if (is_light(c->lights, pl.primitiveIndex)) { out.x = out.y = out.z = 1.0f; return out; }
  }
  if (pl.hitType == 1) {
    const Prim* pr = &c->prims[pl.primitiveIndex];
    const Mat* m = &c->mats[pr->materialIndex];
    f4 position, normal;
    float ndotl;
    /* uint arithmetic on the seed, then uint -> float at the call */
    if (direct_light(c, program, pr, pl.primitiveIndex, pl.u, pl.v, fx, fy, (float)sampleIndex,
                     (float)(sampleIndex + 1u), (float)(sampleIndex + 2u), 0.0f, &position, &normal, &ndotl)) {
      out.x = m->diffuse[0] * ndotl;
      out.y = m->diffuse[1] * ndotl;
      out.z = m->diffuse[2] * ndotl;
    }
  }
  return out;
}

/* global_illumination.cl(ex):68-74 */
static f4 uniformSampleHemisphere(float uvx, float uvy) {
  float z = uvx;
  float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
  float phi = (float)(2.0 * M_PI * (double)uvy);
  return mk4(r * cosf_portable(phi), z, r * sinf_portable(phi), 0.0f);
}
/* global_illumination.cl(ex):76-81 */
static f4 alignHemisphere(f4 h, f4 up) {
  f4 right = normalize4(cross4(up, mk4(0.0072f, 1.0f, 0.0034f, 0.0f)));
  f4 forward = cross4(right, up);
  return add4(add4(scale4(h.x, right), scale4(h.y, up)), scale4(h.z, forward));
}

/* global_illumination.cl(ex):241-375 */
static f3 shade_gi(Ctx* c, Ray cameraRay, float fx, float fy, uint32_t s) {
  f3 direct = {0, 0, 0}, indirect = {0, 0, 0};
  Payload pl = payload_init(FLT_MAX);
  traverse(c, LT_GI, &pl, cameraRay, 0, 0);
  if (is_light(c->lights, pl.primitiveIndex)) {
    direct.x = direct.y = direct.z = 1.0f;
  } else if (pl.hitType == 1) {
    const Prim* pr = &c->prims[pl.primitiveIndex];
    const Mat* m = &c->mats[pr->materialIndex];
    f4 position, normal;
    float ndotl;
    if (direct_light(c, LT_GI, pr, pl.primitiveIndex, pl.u, pl.v, fx, fy, (float)s, (float)(s + 1u),
                     (float)(s + 2u), 1.0f, &position, &normal, &ndotl)) {
      direct.x = m->diffuse[0] * ndotl;
      direct.y = m->diffuse[1] * ndotl;
      direct.z = m->diffuse[2] * ndotl;
    }
    f4 hemi = uniformSampleHemisphere(random_(fx, fy, (float)(s + 3u)), random_(fx, fy, (float)(s + 4u)));
    Ray ext = {position, alignHemisphere(hemi, normal)};
    f4 previousNormal = normal;
    int previousPrimitive = pl.primitiveIndex;
    int rayActive = 1;
    for (int d = 0; d < c->gi_max_depth && rayActive; d++) {
      Payload epl = payload_init(FLT_MAX);
      traverse(c, LT_GI, &epl, ext, 1, previousPrimitive);
      float w = (float)(1.0 / (double)(d + 1));
      uint32_t sd = s + (uint32_t)d;
      if (is_light(c->lights, epl.primitiveIndex)) {
        /* (float3)(w) * (1,1,1) * dot(previousNormal, dir); the loop goes on with the same ray */
        float k = dot4(previousNormal, ext.direction);
        indirect.x += (w * 1.0f) * k;
        indirect.y += (w * 1.0f) * k;
        indirect.z += (w * 1.0f) * k;
      } else if (epl.hitType == 1) {
        const Prim* epr = &c->prims[epl.primitiveIndex];
        const Mat* em = &c->mats[epr->materialIndex];
        f4 epos, enorm;
        float endotl;
        if (direct_light(c, LT_GI, epr, epl.primitiveIndex, epl.u, epl.v, fx, fy, (float)(sd + 5u),
                         (float)(sd + 6u), (float)(sd + 7u), 1.0f, &epos, &enorm, &endotl)) {
          indirect.x += (w * em->diffuse[0]) * endotl;
          indirect.y += (w * em->diffuse[1]) * endotl;
          indirect.z += (w * em->diffuse[2]) * endotl;
          hemi = uniformSampleHemisphere(random_(fx, fy, (float)(sd + 8u)), random_(fx, fy, (float)(sd + 9u)));
          ext.origin = epos;
          ext.direction = alignHemisphere(hemi, enorm);
          previousNormal = enorm;
          previousPrimitive = epl.primitiveIndex;
        } else {
          rayActive = 0;
        }
      } else {
        rayActive = 0;
      }
    }
  }
  f3 out = {direct.x + indirect.x, direct.y + indirect.y, direct.z + indirect.z};
  return out;
}

/* one pixel: the body of linearKernel / tileKernel after blockIDX/blockIDY are known
 * (accumulator.cl:304-318 / :344-358; basic.cl:329-342; basic_lighting.cl:309-321) */
static void pixel(Ctx* c, int program, int mode, const Cam* cam, int blockIDX, int blockIDY, uint32_t W,
                  uint32_t H, float* rgb) {
  f4 cameraPosition = mk4(cam->position[0], cam->position[1], cam->position[2], 1.0f);
  f4 film = mk4(((float)blockIDX / (float)W) - 0.5f, ((float)blockIDY / (float)H) - 0.5f, 0.0f, 1.0f);
  f4 aperture = mk4(0.0f, 0.0f, 5.0f, 1.0f);
  Ray ray = {add4(cameraPosition, film), sub4(aperture, film)};
  float cy = cosf_portable(cam->yaw), sy = sinf_portable(cam->yaw);
  float newX = (cy * ray.direction.x) + (sy * ray.direction.z);
  float newZ = (-sy * ray.direction.x) + (cy * ray.direction.z);
  ray.direction.x = newX;
  ray.direction.z = newZ;

  f3 color;
  if (program == LT_BASIC) {
    color = shade_basic(c, ray);
  } else if (program == LT_CUSTOM) {
    color = shade_custom(c, ray);
  } else if (program == LT_ACCUMULATOR) {
    color = shade_lighting(c, program, ray, film.x, film.y, cam->frameCount);
  } else if (program == LT_GI) {
    color = shade_gi(c, ray, film.x, film.y, cam->frameCount);
  } else { /* 25 blended samples: basic_lighting.cl:309-316, resources GI :408-415 */
    uint32_t base = cam->frameCount * 32u;
    color = (program == LT_BASIC_LIGHTING) ? shade_lighting(c, program, ray, film.x, film.y, base + 0u)
                                           : shade_gi(c, ray, film.x, film.y, base + 0u);
    for (int x = 1; x < 25; x++) {
      float a = ((float)(25 - x)) / (float)25;
      f3 cn = (program == LT_BASIC_LIGHTING) ? shade_lighting(c, program, ray, film.x, film.y, base + (uint32_t)x)
                                             : shade_gi(c, ray, film.x, film.y, base + (uint32_t)x);
      f3 at = {((1.0f - a) * color.x) + (a * cn.x), ((1.0f - a) * color.y) + (a * cn.y),
               ((1.0f - a) * color.z) + (a * cn.z)};
      color = at;
    }
  }
  /* linearKernel of the lighting flavours clamps, tileKernel and basic do not */
  if (program != LT_BASIC && program != LT_CUSTOM && mode == LT_MODE_LINEAR) {
    color.x = clamp01(color.x); color.y = clamp01(color.y); color.z = clamp01(color.z);
  }
  rgb[0] = color.x; rgb[1] = color.y; rgb[2] = color.z;
}

static void ctx_init(Ctx* c, const void* nodes, const void* prims, const void* mats, const void* lights,
                     int gi_max_depth) {
  memset(c, 0, sizeof(*c));
  c->nodes = (const Node*)nodes;
  c->prims = (const Prim*)prims;
  c->mats = (const Mat*)mats;
  c->lights = (const Lights*)lights;
  c->gi_max_depth = gi_max_depth > 0 ? gi_max_depth : 16;
}

static void stats_add(lt_oracle_stats* dst, const lt_oracle_stats* s) {
  if (!dst) return;
  dst->rays += s->rays; dst->shadow_rays += s->shadow_rays;
  dst->node_visits += s->node_visits; dst->tri_tests += s->tri_tests;
  if (s->max_stack > dst->max_stack) dst->max_stack = s->max_stack;
}

/* Whole-image render, rows [y0,y1): every pixel exactly once (the CUDA backend's
 * ceil-div launch semantics, renderer_cuda.cpp:74-88).
 * out is the full W*H*depth image; only rows [y0,y1) are written.
 * stats (may be NULL) is ADDED to, so row bands can be run from several threads
 * with one stats struct each.  Returns 0, or 1 on traversal stack overflow. */
int lt_oracle_render(int program, int mode, const void* nodes, const void* prims, const void* mats,
                     const void* lights, const void* camera28, float* out, uint32_t W, uint32_t H,
                     uint32_t depth, uint32_t y0, uint32_t y1, int gi_max_depth, lt_oracle_stats* stats) {
  Ctx c;
  ctx_init(&c, nodes, prims, mats, lights, gi_max_depth);
  Cam cam;
  memcpy(&cam, camera28, sizeof(cam));
  if (y1 > H) y1 = H;
  for (uint32_t y = y0; y < y1; y++)
    for (uint32_t x = 0; x < W; x++) {
      float rgb[3];
      pixel(&c, program, mode, &cam, (int)x, (int)y, W, H, rgb);
      size_t id = ((size_t)y * W + x) * depth;
      out[id + 0] = rgb[0]; out[id + 1] = rgb[1]; out[id + 2] = rgb[2];
    }
  stats_add(stats, &c.st);
  return c.error;
}

/* Per-pixel work counters for debugging / heat maps: counts[(y*W+x)*4 + {0,1,2,3}] =
 * {rays, shadow rays, node visits, triangle tests} of that pixel. */
int lt_oracle_pixel_counters(int program, int mode, const void* nodes, const void* prims, const void* mats,
                             const void* lights, const void* camera28, uint32_t W, uint32_t H, int gi_max_depth,
                             uint32_t* counts) {
  Ctx c;
  ctx_init(&c, nodes, prims, mats, lights, gi_max_depth);
  Cam cam;
  memcpy(&cam, camera28, sizeof(cam));
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) {
      float rgb[3];
      lt_oracle_stats before = c.st;
      pixel(&c, program, mode, &cam, (int)x, (int)y, W, H, rgb);
      uint32_t* o = counts + ((size_t)y * W + x) * 4;
      o[0] = (uint32_t)(c.st.rays - before.rays);
      o[1] = (uint32_t)(c.st.shadow_rays - before.shadow_rays);
      o[2] = (uint32_t)(c.st.node_visits - before.node_visits);
      o[3] = (uint32_t)(c.st.tri_tests - before.tri_tests);
    }
  return c.error;
}

/* The OpenCL backend's launch decomposition, restated (renderer_opencl.cpp:80-146 and
 * the index arithmetic of linearKernel / tileKernel, accumulator.cl:296-302 / :333-342):
 * workBlockCount = (W / gsx) * (H / gsy) launches of global size (gsx,gsy); tileKernel
 * additionally re-tiles ids through the work-group size (lsx,lsy).  Pixels no launch
 * reaches keep whatever `out` held (the truncation of SURVEY Q10). */
int lt_oracle_render_opencl_launch(int program, int mode, const void* nodes, const void* prims,
                                   const void* mats, const void* lights, const void* camera28, float* out,
                                   uint32_t W, uint32_t H, uint32_t depth, uint32_t gsx, uint32_t gsy,
                                   uint32_t lsx, uint32_t lsy, int gi_max_depth, lt_oracle_stats* stats) {
  Ctx c;
  ctx_init(&c, nodes, prims, mats, lights, gi_max_depth);
  Cam cam;
  memcpy(&cam, camera28, sizeof(cam));
  if (gsx == 0 || gsy == 0 || lsx == 0 || lsy == 0 || gsx % lsx || gsy % lsy) return 2;
  uint32_t workBlockCount = (W / gsx) * (H / gsy);
  uint32_t ngx = gsx / lsx;
  for (uint32_t currentBlock = 0; currentBlock < workBlockCount; currentBlock++)
    for (uint32_t gy = 0; gy < gsy; gy++)
      for (uint32_t gx = 0; gx < gsx; gx++) {
        int blockIDX, blockIDY;
        if (mode == LT_MODE_LINEAR) {
          blockIDY = (int)(((currentBlock / (W / gsx)) * gsy) + gy);
          blockIDX = (int)(((currentBlock % (W / gsx)) * gsx) + gx);
        } else {
          uint32_t groupx = gx / lsx, groupy = gy / lsy, lx = gx % lsx, ly = gy % lsy;
          int localBlockID = (int)(groupy * ngx + groupx);
          int localIDY = (int)(((localBlockID / (gsx / lsx)) * lsy) + ly);
          int localIDX = (int)(((localBlockID % (gsx / lsx)) * lsx) + lx);
          blockIDY = (int)(((currentBlock / (W / gsx)) * gsy) + localIDY);
          blockIDX = (int)(((currentBlock % (W / gsx)) * gsx) + localIDX);
        }
        if ((uint32_t)blockIDX >= W || (uint32_t)blockIDY >= H) continue;
        float rgb[3];
        pixel(&c, program, mode, &cam, blockIDX, blockIDY, W, H, rgb);
        size_t id = ((size_t)blockIDY * W + blockIDX) * depth;
        out[id + 0] = rgb[0]; out[id + 1] = rgb[1]; out[id + 2] = rgb[2];
      }
  stats_add(stats, &c.st);
  return c.error;
}

/* examples/accumulator/resources/shaders/accumulator.frag:10-20:
 * acc <- (c + acc*n) / (n+1), n = frameCount of the frame being added. */
void lt_oracle_accumulate(float* acc, const float* frame, uint64_t count, uint32_t n) {
  if (n == 0) { /* `if (frameCount > 0)` guard of the shader: frame 0 replaces */
    memcpy(acc, frame, count * sizeof(float));
    return;
  }
  float fn = (float)n, fn1 = (float)(n + 1u);
  for (uint64_t i = 0; i < count; i++) acc[i] = (frame[i] + (acc[i] * fn)) / fn1;
}

float lt_oracle_random(float u, float v, float seed) { return random_(u, v, seed); }

/* single-ray probe for unit tests: returns hitType, fills primitiveIndex,t,u,v */
int lt_oracle_trace(int program, const void* nodes, const void* prims, const float origin[4],
                    const float direction[4], float tmax, int useIgnore, int ignore, int* primitiveIndex,
                    float* tuv) {
  Ctx c;
  ctx_init(&c, nodes, prims, NULL, NULL, 16);
  Payload pl = payload_init(tmax);
  Ray r = {mk4(origin[0], origin[1], origin[2], origin[3]), mk4(direction[0], direction[1], direction[2], direction[3])};
  traverse(&c, program, &pl, r, useIgnore, ignore);
  *primitiveIndex = pl.primitiveIndex;
  tuv[0] = pl.t; tuv[1] = pl.u; tuv[2] = pl.v;
  return pl.hitType;
}
