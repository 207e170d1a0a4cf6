// A user program: the shade step of the reference's custom_kernel example
// (examples/custom_kernel/resources/kernels/custom_opencl.cl:226-246), written against lt_kernel.hpp.
namespace lt {
template <class CFG>
__device__ V3 user_shade(const SceneDev& sc, const Ray& cameraRay, float filmX, float filmY, uint32_t frameCount,
                         Stack<CFG::kDeep>& st, Counters& c) {
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  traverse_camera<kCustom, CFG::kDeep, CFG::kStats>(sc, cameraRay, pl, st, c);
  if (pl.hitType == 1) return V3{pl.u, pl.v, (float)((1.0 - (double)pl.u) - (double)pl.v)};
  return V3{0.0f, 0.0f, 0.0f};
}
}  // namespace lt
