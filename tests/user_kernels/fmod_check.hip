// A user program used as a probe: does lt::fmod_pi (lt_device.hpp: one multiplication, one or two fused multiply-adds) return
// the library's fmod(x, M_PI) bit for bit?  Every pixel checks the arguments random() makes for it -- fma(1113.1, seed, dot(uv,
// (12.9898, 78.233))) for 96 seeds from frameCount * 96 on -- and a handful of scaled / negated / edge values; the colour is
// (mismatches, values checked, 0).
namespace lt {
__device__ inline unsigned long long fmod_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
template <class CFG>
__device__ V3 user_shade(const SceneDev& sc, const Ray& cameraRay, float filmX, float filmY, uint32_t frameCount,
                         Stack<CFG::kDeep>& st, Counters& c) {
  const float d = dot2(filmX, filmY, 12.9898f, 78.233f);
  uint32_t bad = 0, n = 0;
  auto check = [&](double x) {
    const double a = fmod_pi(x), b = fmod(x, M_PI);
    const bool same = fmod_bits(a) == fmod_bits(b) || (a != a && b != b);
    bad += same ? 0u : 1u;
    n++;
  };
  for (uint32_t k = 0; k < 96; k++) {
    const double x = __builtin_fma(1113.1, (double)(float)(frameCount * 96u + k), (double)d);
    check(x);
    check(-x);
    check(x * 0x1p+20);
    check(x * 0x1p-30);
    check((double)d + 1113.1 * (double)(float)(frameCount * 96u + k));
  }
  const double edge[] = {0.0, -0.0, M_PI, -M_PI, 2.0 * M_PI, 3.141592653589793, 3.1415926535897927, 3.1415926535897936, 6.283185307179586,
                         6.283185307179585, 1e-310, -1e-310, 0x1p+39, 0x1p+40, 0x1.fffffffffffffp+39, 1e300, -1e300, __builtin_inf(), -__builtin_inf(),
                         __builtin_nan(""), M_PI * 1048576.0, M_PI * 1048577.0, 355.0, 103993.0, 245850922.0};
  for (double e : edge) { check(e); check(e + (double)d * 0x1p-40); }
  return V3{(float)bad, (float)n, 0.0f};
}
}  // namespace lt
