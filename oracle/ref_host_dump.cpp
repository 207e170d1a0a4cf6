// ref_host_dump.cpp -- TEST INFRASTRUCTURE (oracle/_ref).  Driver around the REFERENCE's own
// host classes (compiled from /root/reference/src/{model,acceleration_structure_explicit,
// camera,resource}.cpp where they lie; see oracle/build_ref.sh).  It loads an .obj through the
// reference's Model + AccelerationStructureExplicit + Camera and dumps the five raw buffers the
// renderer plugin uploads (src/opencl/renderer_opencl.cpp:107-120) into one "LTSB" file, so the
// BVH the reference built (including its uninitialised-centroid-bounds behaviour) is
// frozen as a fixture.  Only this container has /root/reference; fixtures are committed under
// tests/golden/ by oracle/make_golden.py.
#include "lens_trace/acceleration_structure_explicit.h"
#include "lens_trace/camera.h"
#include "lens_trace/model.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: ref_host_dump <model.obj (relative to cwd, must contain '/')> <out.ltsb> [camx camy camz yaw]\n");
    return 2;
  }
  float cx = argc > 3 ? atof(argv[3]) : 0.f, cy = argc > 4 ? atof(argv[4]) : 2.5f, cz = argc > 5 ? atof(argv[5]) : -50.f;
  float yaw = argc > 6 ? atof(argv[6]) : 0.f;
  Camera* pCamera = new Camera(cx, cy, cz, yaw);
  Model* pModel = new Model(argv[1]);
  AccelerationStructureExplicitProperties props = {};
  props.sType = STRUCTURE_TYPE_ACCELERATION_STRUCTURE_PROPERTIES;
  props.pNext = NULL;
  props.accelerationStructureExplicitType = ACCELERATION_STRUCTURE_TYPE_BVH;
  props.pModel = pModel;
  AccelerationStructureExplicit* pAS = new AccelerationStructureExplicit(props);

  FILE* f = fopen(argv[2], "wb");
  if (!f) { perror("fopen"); return 1; }
  uint32_t magic = 0x4253544c /* "LTSB" */, version = 1;
  uint64_t sizes[5] = {pAS->getNodeBufferSize(), pAS->getOrderedPrimitiveBufferSize(), pModel->getMaterialBufferSize(),
                       pAS->getLightContainerBufferSize(), pCamera->getCameraBufferSize()};
  fwrite(&magic, 4, 1, f);
  fwrite(&version, 4, 1, f);
  fwrite(sizes, 8, 5, f);
  fwrite(pAS->getNodeBuffer(), 1, sizes[0], f);
  fwrite(pAS->getOrderedPrimitiveBuffer(), 1, sizes[1], f);
  fwrite(pModel->getMaterialBuffer(), 1, sizes[2], f);
  fwrite(pAS->getLightContainerBuffer(), 1, sizes[3], f);
  fwrite(pCamera->getCameraBuffer(), 1, sizes[4], f);
  fclose(f);
  printf("%s: nodes=%llu prims=%llu materials=%llu\n", argv[1], (unsigned long long)(sizes[0] / 32),
         (unsigned long long)(sizes[1] / 76), (unsigned long long)(sizes[2] / 32));
  return 0;
}
