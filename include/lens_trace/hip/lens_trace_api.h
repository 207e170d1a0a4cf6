// lens_trace_api.h -- host-side API surface of lens_trace that the renderer plugin is written against,
// declared from scratch for the MI355X backend (same public names, argument meaning and ownership as the
// reference, so its tests and examples compile against it with only the backend type renamed):
//
//   Renderer                          reference include/lens_trace/renderer.h:5-9
//   StructureType ... KernelMode ...  reference include/lens_trace/structures.h:5-36
//   AccelerationStructureExplicitProperties                       structures.h:91-96
//   Camera                            reference include/lens_trace/camera.h:10-42, src/camera.cpp
//   Model / PrimitiveInfo / Material  reference include/lens_trace/model.h:10-67, src/model.cpp:7-82
//   AccelerationStructureExplicit / LinearBVHNode / Primitive / LightContainer
//                                     reference include/lens_trace/acceleration_structure_explicit.h:9-75
//
// In the reference tree these live in five headers of liblenstrace; a maintainer adding this backend keeps
// theirs and only adds lens_trace/hip/renderer_hip.h.  The scene classes here are this
// repository's own implementations (own .obj/.mtl subset reader, deterministic BVH builder) that emit the
// reference's buffer layouts byte for byte.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

// ---- tags -------------------------------------------------------------------------------------------
// Existing enumerators keep the reference's values; the HIP ones are appended.
enum StructureType {
  STRUCTURE_TYPE_RENDER_PROPERTIES_OPENCL,
  STRUCTURE_TYPE_THREAD_ORGANIZATION_OPENCL,
  STRUCTURE_TYPE_RENDER_PROPERTIES_CUDA,
  STRUCTURE_TYPE_THREAD_ORGANIZATION_CUDA,
  STRUCTURE_TYPE_BUFFER_TO_IMAGE_PROPERTIES,
  STRUCTURE_TYPE_ACCELERATION_STRUCTURE_PROPERTIES,
  STRUCTURE_TYPE_RENDER_PROPERTIES_HIP,
  STRUCTURE_TYPE_THREAD_ORGANIZATION_HIP,
  STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP,
  STRUCTURE_TYPE_BACKEND_PROPERTIES_HIP
};
enum RenderPlatform { RENDER_PLATFORM_OPENCL, RENDER_PLATFORM_CUDA, RENDER_PLATFORM_OPTIX, RENDER_PLATFORM_HIP };
enum KernelMode { KERNEL_MODE_LINEAR, KERNEL_MODE_TILE };
enum ThreadOrganizationMode { THREAD_ORGANIZATION_MODE_MAX_FIT, THREAD_ORGANIZATION_MODE_CUSTOM };
// ..._BVH is the reference's builder (median split of the largest centroid extent, acceleration_structure_explicit.cpp:47-137);
// ..._BVH_SAH is this backend's addition: binned surface-area-heuristic splits, SAME node / primitive / light layouts, one
// triangle per leaf, height bounded (see scene_host.cpp) -- any renderer that reads the reference's buffers reads these.
enum AccelerationStructureExplicitType { ACCELERATION_STRUCTURE_TYPE_BVH, ACCELERATION_STRUCTURE_TYPE_BVH_SAH };
enum ImageType { IMAGE_TYPE_JPEG };

struct AccelerationStructureExplicitProperties {
  StructureType sType;
  void* pNext;
  AccelerationStructureExplicitType accelerationStructureExplicitType;
  void* pModel;
};

// ---- image output (include/lens_trace/image_writer.h, structures.h:81-89) --------------------------------
struct BufferToImageProperties {
  StructureType sType;
  void* pNext;
  void* pBuffer;               // float[W*H*D], values in [0,1]
  uint64_t bufferSize;
  uint64_t imageDimensions[3];
  ImageType imageType;
  const char* filename;
};

class ImageWriter {
 public:
  static void writeBufferToImage(BufferToImageProperties bufferToImageProperties);   // value*255 -> 8 bits -> quality-100 JPEG
};

// ---- the plugin interface ------------------------------------------------------------------------------
class Renderer {
 public:
  virtual void render(void* pRenderProperties) = 0;
};

// ---- camera: 28-byte buffer {position[3], yaw, pitch, roll, uint frameCount} ----------------------------
class Camera {
 public:
  Camera(float positionX, float positionY, float positionZ, float yaw = 0, float pitch = 0, float roll = 0);
  ~Camera();
  float getPositionX();
  float getPositionY();
  float getPositionZ();
  float getYaw();
  float getPitch();
  float getRoll();
  uint32_t getFrameCount();
  void setPosition(float x, float y, float z);
  void updatePosition(float x, float y, float z);       // adds to the position
  void setRotation(float yaw, float pitch, float roll);
  void updateRotation(float yaw, float pitch, float roll);
  void incrementFrameCount();
  void resetFrameCount();
  void* getCameraBuffer();
  uint64_t getCameraBufferSize();

 private:
  void sync();
  float position[3];
  float yaw, pitch, roll;
  uint32_t frameCount;
  unsigned char buffer[28];
};

// ---- model ----------------------------------------------------------------------------------------------
struct PrimitiveInfo {
  float positionA[3], positionB[3], positionC[3];
  float normalA[3], normalB[3], normalC[3];
  int materialIndex;
  float boundsMin[3], boundsMax[3];
  float centroid[3];            // centre of the AABB (not the vertex mean), as the reference computes it
};

struct Material {               // 32 bytes
  float diffuse[3];
  float ior;
  float dissolve;
  float emission[3];
};

class Model {
 public:
  // Wavefront .obj + .mtl (subset: v, vn, vt, f with v / v/vt / v//vn / v/vt/vn and negative indices,
  // mtllib, usemtl; newmtl, Kd, Ke, Ni, d, Tr).  Polygons are triangulated: quads on their shorter diagonal
  // (ties on 1-3, as tinyobjloader does), larger polygons by ear clipping.
  explicit Model(std::string fileName);
  // From raw arrays (synthetic scenes): 9 floats of positions and of normals per triangle.
  Model(const float* positions, const float* normals, const int* materialIndices, uint64_t triangleCount,
        const Material* materials, uint64_t materialCount);
  ~Model();
  std::string getFileName();
  bool checkError();            // prints warning / error text, returns success
  std::vector<PrimitiveInfo>* getPrimitiveInfoListP();
  uint64_t getMaterialBufferSize();
  void* getMaterialBuffer();

 private:
  void addTriangle(const float* p, const float* n, int materialIndex);
  std::vector<PrimitiveInfo> primitiveInfoList;
  std::vector<Material> materialList;
  std::string fileName, warning, error;
  bool success;
};

// ---- acceleration structure ---------------------------------------------------------------------------
struct LinearBVHNode {          // 32 bytes; pre-order: the left child of node i is node i+1
  float boundsMin[3];
  float boundsMax[3];
  union {
    int primitivesOffset;       // leaf
    int secondChildOffset;      // interior
  };
  uint16_t primitiveCount;      // 0 = interior
  uint8_t axis;
  uint8_t pad[1];
};

struct Primitive {              // 76 bytes
  float positionA[3], positionB[3], positionC[3];
  float normalA[3], normalB[3], normalC[3];
  int materialIndex;
};

struct LightContainer {         // 260 bytes
  uint32_t count;
  uint32_t primitives[64];
};

class AccelerationStructureExplicit {
 public:
  explicit AccelerationStructureExplicit(AccelerationStructureExplicitProperties properties);
  ~AccelerationStructureExplicit();
  uint64_t getNodeBufferSize();
  void* getNodeBuffer();
  uint64_t getOrderedPrimitiveBufferSize();
  void* getOrderedPrimitiveBuffer();
  uint64_t getLightContainerBufferSize();
  void* getLightContainerBuffer();
  int getHeight();              // interior ancestors of the deepest node (the traversal stack it needs)

 private:
  std::vector<LinearBVHNode> nodes;
  std::vector<Primitive> orderedPrimitives;
  LightContainer lightContainer;
  int height;
};
