// renderer_hip.cpp -- RendererHIP: the thin C++ plugin over the C ABI (include/lenstrace_hip.h).
// Mirrors RendererOpenCL::render (reference src/opencl/renderer_opencl.cpp:56-153): tag check that prints and
// carries on, buffers pulled from the opaque AccelerationStructureExplicit / Model / Camera pointers, synchronous
// fill of the caller's pOutputBuffer, void return with errors reported on stdout.
#include "lens_trace/hip/renderer_hip.h"

#include <stdio.h>
#include <string.h>

#include "lenstrace_hip.h"

// 64-bit FNV-1a over the sizes and a strided sample of the four scene buffers: cheap (a few KB per call), and enough to
// tell a different scene that reuses the same addresses from the cached one.
static uint64_t sceneFingerprint(const void* const* buf, const uint64_t* size) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](const unsigned char* p, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
  };
  for (int b = 0; b < 4; b++) {
    const unsigned char* p = (const unsigned char*)buf[b];
    const uint64_t n = size[b];
    mix((const unsigned char*)&n, sizeof(n));
    const uint64_t step = n / 4096 + 1;
    for (uint64_t i = 0; i + 8 <= n; i += step * 8) mix(p + i, 8);
    mix(p + (n > 256 ? n - 256 : 0), n > 256 ? 256 : n);
  }
  return h;
}

RendererHIP::RendererHIP() : RendererHIP(0) {}

RendererHIP::RendererHIP(int deviceIndex) : context(nullptr) {
  memset(cachedKey, 0, sizeof(cachedKey));
  memset(cachedSize, 0, sizeof(cachedSize));
  cachedFingerprint = 0;
  if (lt_hip_create(deviceIndex, &context) != LT_OK) {
    printf("ERROR: RendererHIP: %s\n", lt_hip_last_error(nullptr));
    context = nullptr;
  }
}

RendererHIP::~RendererHIP() { lt_hip_destroy(context); }

const char* RendererHIP::getLastError() const { return lt_hip_last_error(context); }

void RendererHIP::invalidateScene() { memset(cachedKey, 0, sizeof(cachedKey)); }

void RendererHIP::render(void* pRenderProperties) {
  RenderPropertiesHIP* props = (RenderPropertiesHIP*)pRenderProperties;
  if (props->sType != STRUCTURE_TYPE_RENDER_PROPERTIES_HIP) {
    printf("ERROR: RenderPropertiesHIP sType\n");
  }
  if (props->threadOrganizationMode == THREAD_ORGANIZATION_MODE_CUSTOM &&
      props->threadOrganization.sType != STRUCTURE_TYPE_THREAD_ORGANIZATION_HIP) {
    printf("ERROR: ThreadOrganizationHIP sType\n");
  }
  if (!context) {
    printf("Kernel Error: no HIP context\n");
    return;
  }
  AccelerationStructureExplicit* pAS = (AccelerationStructureExplicit*)props->pAccelerationStructureExplicit;
  Model* pModel = (Model*)props->pModel;
  Camera* pCamera = (Camera*)props->pCamera;

  int program = 0;
  if (lt_hip_resolve_program(context, props->kernelFilePath.c_str(), &program) != LT_OK) {
    printf("%s\n", lt_hip_last_error(context));   // like the reference's build log (renderer_opencl.cpp:50-53)
    return;
  }

  const void* key[4] = {pAS->getNodeBuffer(), pAS->getOrderedPrimitiveBuffer(), pModel->getMaterialBuffer(),
                        pAS->getLightContainerBuffer()};
  const uint64_t size[4] = {pAS->getNodeBufferSize(), pAS->getOrderedPrimitiveBufferSize(), pModel->getMaterialBufferSize(),
                            pAS->getLightContainerBufferSize()};
  const uint64_t fingerprint = sceneFingerprint(key, size);
  if (memcmp(key, cachedKey, sizeof(key)) != 0 || memcmp(size, cachedSize, sizeof(size)) != 0 || fingerprint != cachedFingerprint) {
    if (lt_hip_set_scene(context, key[0], size[0], key[1], size[1], key[2], size[2], key[3], size[3]) != LT_OK) {
      printf("Kernel Error: %s\n", lt_hip_last_error(context));
      memset(cachedKey, 0, sizeof(cachedKey));
      return;
    }
    memcpy(cachedKey, key, sizeof(key));
    memcpy(cachedSize, size, sizeof(size));
    cachedFingerprint = fingerprint;
  }

  lt_hip_render_desc desc;
  memset(&desc, 0, sizeof(desc));
  desc.struct_size = sizeof(desc);
  desc.program = program;
  desc.kernel_mode = props->kernelMode == KERNEL_MODE_TILE ? LT_KERNEL_MODE_TILE : LT_KERNEL_MODE_LINEAR;
  desc.width = (uint32_t)props->imageDimensions[0];
  desc.height = (uint32_t)props->imageDimensions[1];
  desc.depth = (uint32_t)props->imageDimensions[2];
  if (pCamera->getCameraBufferSize() != sizeof(desc.camera)) {
    printf("Kernel Error: camera buffer is not 28 bytes\n");
    return;
  }
  memcpy(desc.camera, pCamera->getCameraBuffer(), sizeof(desc.camera));
  for (void* p = props->pNext; p != nullptr;) {
    StructureType t = *(StructureType*)p;
    if (t == STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP) {
      ProgressivePropertiesHIP* pp = (ProgressivePropertiesHIP*)p;
      desc.frame_first = pp->frameFirst;
      desc.frame_count = pp->frameCount;
      desc.accumulate = pp->accumulate;
      desc.accumulate_base = pp->accumulateBase;
      desc.gi_max_depth = pp->giMaxDepth;
      p = pp->pNext;
    } else {
      break;
    }
  }
  if (lt_hip_render(context, &desc, (float*)props->pOutputBuffer, props->outputBufferSize) != LT_OK) {
    printf("Kernel Error: %s\n", lt_hip_last_error(context));
  }
}
