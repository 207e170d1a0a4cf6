// A user program that exposes the raw hit record: colour = (t, primitiveIndex, hitType), then a second ray straight back
// along the camera ray from the hit point (ignoring the hit primitive) whose hit flag is added to the third channel.
namespace lt {
template <class CFG>
__device__ V3 user_shade(const SceneDev& sc, const Ray& cameraRay, float filmX, float filmY, uint32_t frameCount,
                         Stack<CFG::kDeep>& st, Counters& c) {
  Hit pl{0, 0, kFltMax, 0.0f, 0.0f};
  traverse_camera<kAccumulator, CFG::kDeep, CFG::kStats>(sc, cameraRay, pl, st, c);
  if (pl.hitType != 1) return V3{-1.0f, -1.0f, 0.0f};
  const float* pr = prim_ptr(sc, pl.prim);
  const V3 b = barycentrics(pl.u, pl.v);
  const V3 p3 = bary3(pr + 0, pr + 3, pr + 6, b);
  const Ray back{mk4(p3.x, p3.y, p3.z, 1.0f), neg4(cameraRay.d)};
  Hit pl2{0, 0, kFltMax, 0.0f, 0.0f};
  traverse<kAccumulator, CFG::kDeep, CFG::kStats, false>(sc, back, true, pl.prim, pl2, st, c);
  return V3{pl.t, (float)pl.prim, 1.0f + 2.0f * (float)pl2.hitType};
}
}  // namespace lt
