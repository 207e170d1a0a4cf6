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

// Optional extension, chained through pNext like the one above (in any order).
struct BackendPropertiesHIP {
  StructureType sType;          // STRUCTURE_TYPE_BACKEND_PROPERTIES_HIP
  void* pNext;
  // Floating-point flavour (include/lenstrace_hip.h, LT_RENDER_FLAG_*_MATH).  Both 0 (default): render() is bit-identical to
  // RendererOpenCL running the same kernel file on the MI355X (clBuildProgram with NULL options, as the reference builds it).
  // strictMath: the same kernels built with -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt.  portableMath: strict,
  // with the device library's approximate leaf functions in correctly rounded forms (what the CPU oracle computes).
  uint32_t portableMath;
  uint32_t strictMath;
  // Scene-change contract.  0 (default): every render() hands the four scene buffers over with the frame, as the reference's
  // callers do (renderer_opencl.cpp:107-120): they are hashed in full WHILE the frame renders (lt_hip_render_scene) and uploaded
  // -- and the frame rendered again -- only when a byte changed; bench.py reports what that costs end to end
  // (config.e2e_frame_ms_plugin, scene_hash_ms).  != 0: the caller versions its scene -- the buffers are looked at only when the
  // pointers, sizes or this number differ from the previous call's.
  uint64_t sceneVersion;
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
  // Scene buffers stay resident while render() keeps receiving the same content (BackendPropertiesHIP::sceneVersion).
  // Forces the next render() to upload again.
  void invalidateScene();
  const char* getLastError() const;

 private:
  lt_hip_context* context;
  // scene cache key: the reference re-uploads all buffers on every render(); here they stay in HBM until the
  // caller hands over different objects or buffers
  const void* cachedKey[4];
  uint64_t cachedSize[4];
  uint64_t cachedVersion;
};
