// renderer_hip.cpp -- RendererHIP: the thin C++ plugin over the C ABI (include/lenstrace_hip.h).
// Mirrors RendererOpenCL::render (reference src/opencl/renderer_opencl.cpp:56-153): tag check that prints and
// carries on, buffers pulled from the opaque AccelerationStructureExplicit / Model / Camera pointers, synchronous
// fill of the caller's pOutputBuffer, void return with errors reported on stdout.
#include "lens_trace/hip/renderer_hip.h"

#include <stdio.h>
#include <string.h>

#include "lenstrace_hip.h"

RendererHIP::RendererHIP() : RendererHIP(0) {}

RendererHIP::RendererHIP(int deviceIndex) : context(nullptr) {
  memset(cachedKey, 0, sizeof(cachedKey));
  memset(cachedSize, 0, sizeof(cachedSize));
  cachedVersion = 0;
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
  // extension structs (the reference leaves pNext NULL)
  const ProgressivePropertiesHIP* progressive = nullptr;
  const BackendPropertiesHIP* backend = nullptr;
  for (void* p = props->pNext; p != nullptr;) {
    const StructureType t = *(StructureType*)p;
    if (t == STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP) {
      progressive = (const ProgressivePropertiesHIP*)p;
      p = progressive->pNext;
    } else if (t == STRUCTURE_TYPE_BACKEND_PROPERTIES_HIP) {
      backend = (const BackendPropertiesHIP*)p;
      p = backend->pNext;
    } else {
      printf("ERROR: unknown sType in the pNext chain of RenderPropertiesHIP\n");
      break;
    }
  }
  // A caller that versions its scene has it looked at only when the objects or the number change; anyone else hands it over with
  // every frame (lt_hip_render_scene below: hashed in full while the frame renders, uploaded only when it changed).
  const uint64_t version = backend ? backend->sceneVersion : 0;
  const bool sameObjects = memcmp(key, cachedKey, sizeof(key)) == 0 && memcmp(size, cachedSize, sizeof(size)) == 0;
  const bool known = sameObjects && version != 0 && version == cachedVersion;

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
  if (progressive) {
    desc.frame_first = progressive->frameFirst;
    desc.frame_count = progressive->frameCount;
    desc.accumulate = progressive->accumulate;
    desc.accumulate_base = progressive->accumulateBase;
    desc.gi_max_depth = progressive->giMaxDepth;
  }
  if (backend && backend->portableMath) desc.flags |= LT_RENDER_FLAG_PORTABLE_MATH;
  if (backend && backend->strictMath) desc.flags |= LT_RENDER_FLAG_STRICT_MATH;
  const int rc = known ? lt_hip_render(context, &desc, (float*)props->pOutputBuffer, props->outputBufferSize)
                       : lt_hip_render_scene(context, key[0], size[0], key[1], size[1], key[2], size[2], key[3], size[3], &desc,
                                             (float*)props->pOutputBuffer, props->outputBufferSize);
  if (rc != LT_OK) {
    printf("Kernel Error: %s\n", lt_hip_last_error(context));
    if (!known) memset(cachedKey, 0, sizeof(cachedKey));
    return;
  }
  memcpy(cachedKey, key, sizeof(key));
  memcpy(cachedSize, size, sizeof(size));
  cachedVersion = version;
}
