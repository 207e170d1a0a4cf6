// hip_renderer_test.cpp -- the reference's renderer tests (tests/opencl_renderer_test.cc and
// tests/cuda_renderer_test.cc: ValidEngine, ValidBuffer, CustomBlockSize, KernelMode, CorrectColor) re-stated
// for RendererHIP through the C++ plugin surface, with a 30-line test runner (gtest is not in the image).
// The wall model is written by the test itself (same geometry as the reference's green_wall.obj: a 50x50 quad
// at z ~ 0, Kd 0 1 0); the camera, image size, kernel path and sampled indices are the reference's.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "lens_trace/hip/renderer_hip.h"

static int g_failed = 0, g_checks = 0;
#define EXPECT_TRUE(c) do { g_checks++; if (!(c)) { g_failed++; printf("  EXPECT_TRUE failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); } } while (0)
#define EXPECT_FLOAT_EQ(a, b) do { g_checks++; float a_ = (a), b_ = (b); if (!(fabsf(a_ - b_) <= 4 * 1.1920929e-7f * fmaxf(fabsf(a_), fabsf(b_)))) { g_failed++; printf("  EXPECT_FLOAT_EQ failed: %s=%g %s=%g (%s:%d)\n", #a, a_, #b, b_, __FILE__, __LINE__); } } while (0)

static std::string g_dir;

static std::string writeWall() {
  const std::string obj = g_dir + "/green_wall.obj", mtl = g_dir + "/green_wall.mtl";
  FILE* f = fopen(mtl.c_str(), "w");
  fprintf(f, "newmtl Material\nKd 0.0 1.0 0.0\nKe 0.0 0.0 0.0\nNi 1.45\nd 1.0\n");
  fclose(f);
  f = fopen(obj.c_str(), "w");
  fprintf(f, "mtllib green_wall.mtl\nv -25 -25 -0.000001\nv 25 -25 -0.000001\nv -25 25 0.000001\nv 25 25 0.000001\n"
             "vn 0 0 1\nusemtl Material\nf 1//1 2//1 4//1 3//1\n");
  fclose(f);
  return obj;
}

struct Fixture {
  Camera* pCamera;
  Model* pModel;
  AccelerationStructureExplicit* pAS;
  Fixture() {
    pCamera = new Camera(0, 2.5, -50, 0);
    pModel = new Model(writeWall());
    AccelerationStructureExplicitProperties p = {};
    p.sType = STRUCTURE_TYPE_ACCELERATION_STRUCTURE_PROPERTIES;
    p.accelerationStructureExplicitType = ACCELERATION_STRUCTURE_TYPE_BVH;
    p.pModel = pModel;
    pAS = new AccelerationStructureExplicit(p);
  }
  ~Fixture() { delete pAS; delete pModel; delete pCamera; }
  RenderPropertiesHIP props(void* out, uint64_t bytes) {
    RenderPropertiesHIP r = {};
    r.sType = STRUCTURE_TYPE_RENDER_PROPERTIES_HIP;
    r.pNext = NULL;
    r.kernelFilePath = "resources/kernels/opencl/basic.cl";
    r.kernelMode = KERNEL_MODE_LINEAR;
    r.threadOrganizationMode = THREAD_ORGANIZATION_MODE_MAX_FIT;
    r.imageDimensions[0] = 100; r.imageDimensions[1] = 100; r.imageDimensions[2] = 3;
    r.pOutputBuffer = out;
    r.outputBufferSize = bytes;
    r.pAccelerationStructureExplicit = pAS;
    r.pModel = pModel;
    r.pCamera = pCamera;
    return r;
  }
};

static void CreateEngineTEST_ValidEngine() {
  RendererHIP* renderer = new RendererHIP();
  EXPECT_TRUE(renderer != NULL);
  EXPECT_TRUE(renderer->isValid());
  delete renderer;
}

static void RenderBufferTEST_ValidBuffer() {
  RendererHIP* renderer = new RendererHIP();
  EXPECT_TRUE(renderer != NULL);
  uint64_t size = sizeof(float) * 100 * 100 * 3;
  void* out = malloc(size);
  Fixture fx;
  RenderPropertiesHIP rp = fx.props(out, size);
  renderer->render(&rp);
  delete renderer;
  free(out);
}

static void RenderBufferTEST_CustomBlockSize() {
  RendererHIP* renderer = new RendererHIP();
  uint64_t size = sizeof(float) * 100 * 100 * 3;
  float* a = (float*)malloc(size); float* b = (float*)malloc(size); float* c = (float*)malloc(size);
  Fixture fx;
  RenderPropertiesHIP rp = fx.props(a, size);
  renderer->render(&rp);
  rp.pOutputBuffer = b;
  rp.threadOrganizationMode = THREAD_ORGANIZATION_MODE_CUSTOM;
  ThreadOrganizationHIP to = {};
  to.sType = STRUCTURE_TYPE_THREAD_ORGANIZATION_HIP;
  to.blockSize[0] = 8; to.blockSize[1] = 8;
  rp.threadOrganization = to;
  renderer->render(&rp);
  rp.pOutputBuffer = c;
  to.blockSize[0] = 4; to.blockSize[1] = 4;
  rp.threadOrganization = to;
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100; x += 32) {
    EXPECT_FLOAT_EQ(a[x], b[x]);
    EXPECT_FLOAT_EQ(b[x], c[x]);
  }
  delete renderer;
  free(c); free(b); free(a);
}

static void RenderBufferTEST_KernelMode() {
  RendererHIP* renderer = new RendererHIP();
  uint64_t size = sizeof(float) * 100 * 100 * 3;
  float* a = (float*)malloc(size); float* b = (float*)malloc(size);
  Fixture fx;
  RenderPropertiesHIP rp = fx.props(a, size);
  renderer->render(&rp);
  rp.pOutputBuffer = b;
  rp.kernelMode = KERNEL_MODE_TILE;
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100 * 3; x += 32) EXPECT_FLOAT_EQ(a[x], b[x]);
  delete renderer;
  free(b); free(a);
}

static void RenderBufferTEST_CorrectColor() {
  RendererHIP* renderer = new RendererHIP();
  uint64_t size = sizeof(float) * 100 * 100 * 3;
  float* out = (float*)malloc(size);
  memset(out, 0xff, size);
  Fixture fx;
  RenderPropertiesHIP rp = fx.props(out, size);
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100; x += 8 * 3) {
    EXPECT_FLOAT_EQ(out[x + 0], 0.0);
    EXPECT_FLOAT_EQ(out[x + 1], 1.0);
    EXPECT_FLOAT_EQ(out[x + 2], 0.0);
  }
  delete renderer;
  free(out);
}

// beyond the reference's tests: the progressive extension struct chained through pNext
static void RenderBufferTEST_ProgressiveExtension() {
  RendererHIP* renderer = new RendererHIP();
  uint64_t size = sizeof(float) * 100 * 100 * 3;
  float* a = (float*)malloc(size); float* b = (float*)malloc(size);
  Fixture fx;
  RenderPropertiesHIP rp = fx.props(a, size);
  renderer->render(&rp);
  ProgressivePropertiesHIP pp = {};
  pp.sType = STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP;
  pp.frameFirst = 1; pp.frameCount = 4; pp.accumulate = 1;
  rp.pNext = &pp;
  rp.pOutputBuffer = b;
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100 * 3; x += 32) EXPECT_FLOAT_EQ(a[x], b[x]);   // basic is frame-independent
  delete renderer;
  free(b); free(a);
}

// beyond the reference's tests: the backend extension struct -- math flavour, and the scene-change contract.  The reference
// uploads every buffer on every render() (src/opencl/renderer_opencl.cpp:107-120), so an in-place edit of a scene buffer
// between two calls must show, without any call to invalidateScene().
static void RenderBufferTEST_BackendExtension() {
  RendererHIP* renderer = new RendererHIP();
  uint64_t size = sizeof(float) * 100 * 100 * 3;
  float* a = (float*)malloc(size); float* b = (float*)malloc(size);
  Fixture fx;
  RenderPropertiesHIP rp = fx.props(a, size);
  renderer->render(&rp);
  BackendPropertiesHIP bp = {};
  bp.sType = STRUCTURE_TYPE_BACKEND_PROPERTIES_HIP;
  bp.portableMath = 1;
  ProgressivePropertiesHIP pp = {};
  pp.sType = STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP;
  pp.frameFirst = 0; pp.frameCount = 1;
  bp.pNext = &pp;                      // both extensions in one chain
  rp.pNext = &bp;
  rp.pOutputBuffer = b;
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100 * 3; x += 32) EXPECT_FLOAT_EQ(a[x], b[x]);   // basic on an unrotated camera: both flavours agree
  // in-place edit, default contract (sceneVersion 0: content hash): green -> red
  float* diffuse = (float*)fx.pModel->getMaterialBuffer();
  diffuse[0] = 1.0f; diffuse[1] = 0.0f;
  rp.pNext = NULL;
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100; x += 8 * 3) { EXPECT_FLOAT_EQ(b[x + 0], 1.0); EXPECT_FLOAT_EQ(b[x + 1], 0.0); }
  // versioned contract: same version -> the resident scene is used as is; new version -> looked at again
  bp.pNext = NULL; bp.portableMath = 0; bp.sceneVersion = 7;
  rp.pNext = &bp;
  renderer->render(&rp);
  diffuse[0] = 0.0f; diffuse[2] = 1.0f;   // red -> blue, version unchanged: the caller said nothing changed
  renderer->render(&rp);
  EXPECT_FLOAT_EQ(b[0], 1.0); EXPECT_FLOAT_EQ(b[2], 0.0);
  bp.sceneVersion = 8;
  renderer->render(&rp);
  for (int x = 0; x < 100 * 100; x += 8 * 3) { EXPECT_FLOAT_EQ(b[x + 0], 0.0); EXPECT_FLOAT_EQ(b[x + 2], 1.0); }
  delete renderer;
  free(b); free(a);
}

int main(int argc, char** argv) {
  g_dir = argc > 1 ? argv[1] : "/tmp";
  struct { const char* name; void (*fn)(); } tests[] = {
      {"CreateEngineTEST.ValidEngine", CreateEngineTEST_ValidEngine},
      {"RenderBufferTEST.ValidBuffer", RenderBufferTEST_ValidBuffer},
      {"RenderBufferTEST.CustomBlockSize", RenderBufferTEST_CustomBlockSize},
      {"RenderBufferTEST.KernelMode", RenderBufferTEST_KernelMode},
      {"RenderBufferTEST.CorrectColor", RenderBufferTEST_CorrectColor},
      {"RenderBufferTEST.ProgressiveExtension", RenderBufferTEST_ProgressiveExtension},
      {"RenderBufferTEST.BackendExtension", RenderBufferTEST_BackendExtension},
  };
  int bad = 0;
  for (auto& t : tests) {
    const int before = g_failed;
    printf("[ RUN      ] %s\n", t.name);
    t.fn();
    printf(g_failed == before ? "[       OK ] %s\n" : "[  FAILED  ] %s\n", t.name);
    bad += g_failed != before;
  }
  printf("%d checks, %d failed, %d of %d tests failed\n", g_checks, g_failed, bad, (int)(sizeof(tests) / sizeof(tests[0])));
  return bad ? 1 : 0;
}
