// renderer_hip.h -- the MI355X renderer plugin: `class RendererHIP : public Renderer` with the same
// construct / render(void*) / destroy contract as RendererOpenCL (reference
// include/lens_trace/opencl/renderer_opencl.h:16-43) and RendererCUDA (cuda/renderer_cuda.h:16-29).
// RenderPropertiesHIP has the fields of RenderPropertiesCUDA (structures.h:66-79) under new sType tags.
#pragma once
#include "lens_trace/hip/lens_trace_api.h"

struct lt_hip_context;

struct ThreadOrganizationHIP {
  StructureType sType;          // STRUCTURE_TYPE_THREAD_ORGANIZATION_HIP
  void* pNext;
  uint64_t blockSize[2];        // accepted and ignored: pixels never depend on the launch decomposition
};

// Optional extension, chained through RenderPropertiesHIP::pNext (the reference leaves pNext unused): run the
// examples' progressive loop on the device -- frames frameFirst..frameFirst+frameCount-1, folded with
// accumulator.frag's running mean -- instead of one read-back per sample.
struct ProgressivePropertiesHIP {
  StructureType sType;          // STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP
  void* pNext;
  uint32_t frameFirst;
  uint32_t frameCount;
  uint32_t accumulate;
  uint32_t accumulateBase;
  int32_t giMaxDepth;           // 0 = the reference's 16
};

struct RenderPropertiesHIP {
  StructureType sType;          // STRUCTURE_TYPE_RENDER_PROPERTIES_HIP
  void* pNext;
  std::string kernelFilePath;   // basename selects the built-in program (basic, basic_lighting, accumulator, global_illumination)
  KernelMode kernelMode;
  ThreadOrganizationMode threadOrganizationMode;
  ThreadOrganizationHIP threadOrganization;
  uint64_t imageDimensions[3];
  void* pOutputBuffer;
  uint64_t outputBufferSize;
  void* pAccelerationStructureExplicit;
  void* pModel;
  void* pCamera;
};

class RendererHIP final : public Renderer {
 public:
  RendererHIP();                // first GPU; RendererHIP(int) picks a HIP ordinal
  explicit RendererHIP(int deviceIndex);
  ~RendererHIP();
  void render(void* pRenderProperties);
  bool isValid() const { return context != nullptr; }
  // Scene buffers are uploaded once and reused while render() keeps receiving the same buffers (same addresses, sizes
  // and content fingerprint).  Call this after modifying a scene buffer in place.
  void invalidateScene();
  const char* getLastError() const;

 private:
  lt_hip_context* context;
  // scene cache key: the reference re-uploads all buffers on every render(); here they stay in HBM until the
  // caller hands over different objects or buffers
  const void* cachedKey[4];
  uint64_t cachedSize[4];
  uint64_t cachedFingerprint;
};
