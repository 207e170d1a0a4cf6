// lt_own16.hpp -- the 16-byte child slots of the per-lane walks' 4-wide groups (SceneDev::wide, traverse_own_lane in
// lt_device.hpp): how a node of the backend's own tree is put on a 16-bit grid over the scene's bounds.  Host and device share
// this arithmetic (lt_wide_kernel at upload; lt_hip_own_wide on the host for the CPU tests).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define LT_HD __host__ __device__
#else
#define LT_HD
#endif

namespace lt_own16 {

LT_HD inline uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
LT_HD inline float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// A bound pushed away from its box by 2^-21 of its magnitude and one float more: lo' <= lo - 9 * 2^-24 |lo| (what the
// conservative slab tests of the packet walks and of the per-lane walks need: >= 6 resp. 8 units of 2^-24 |bound|).
LT_HD inline float outwards(float b, bool up) {
  const float t = up ? b + __builtin_fabsf(b) * 0x1p-21f : b - __builtin_fabsf(b) * 0x1p-21f;
  uint32_t u = f2u(t);
  if ((u & 0x7fffffffu) == 0u) return u2f((up ? 0u : 0x80000000u) | 1u);    // +-0 -> the smallest denormal of that side
  const bool away = (t > 0.0f) == up;   // moving away from zero = the next larger magnitude
  u = away ? u + 1u : u - 1u;
  return u2f(u);
}

// The grid of one axis for a scene whose root box spans [lo, hi]: origin O and step S (floats) with O <= outwards(lo) and
// O + 65535 S >= outwards(hi), both with a few steps to spare.
LT_HD inline void frame(float lo, float hi, float& O, float& S) {
  const float l = outwards(outwards(lo, false), false), h = outwards(outwards(hi, true), true);
  O = l;
  double step = ((double)h - (double)l) / 65528.0;
  const double floorStep = __builtin_fmax(__builtin_fabs((double)l), __builtin_fabs((double)h)) * 0x1p-28 + 0x1p-120;   // (a flat scene: any positive step)
  if (!(step > floorStep)) step = floorStep;
  float s = (float)step;
  if ((double)s < step) s = u2f(f2u(s) + 1u);   // round the step up
  S = s;
}

// The quantised bounds of one axis of one node: the largest ql with O + ql S <= outwards(lo) and the smallest qh with
// O + qh S >= outwards(hi), the sums formed in double.  false when either falls off the grid.
LT_HD inline bool quantise(float lo, float hi, float Of, float Sf, uint32_t& ql, uint32_t& qh) {
  const double O = (double)Of, S = (double)Sf;
  const double tl = (double)outwards(lo, false), th = (double)outwards(hi, true);
  double q = __builtin_floor((tl - O) / S);
  q = __builtin_fmin(__builtin_fmax(q, -2.0), 65537.0);
  while (O + q * S > tl && q >= 0.0) q -= 1.0;
  bool ok = q >= 0.0 && q <= 65535.0;
  ql = (uint32_t)__builtin_fmin(__builtin_fmax(q, 0.0), 65535.0);
  q = __builtin_ceil((th - O) / S);
  q = __builtin_fmin(__builtin_fmax(q, -2.0), 65537.0);
  while (O + q * S < th && q <= 65535.0) q += 1.0;
  ok = ok && q >= 0.0 && q <= 65535.0;
  qh = (uint32_t)__builtin_fmin(__builtin_fmax(q, 0.0), 65535.0);
  return ok;
}

// One child slot of a 4-wide group (lt_retree::collapse_wide): the child's box on the grid + its link -- the child's own group
// for an interior node; 0x80000000 | (groups + primitive offset), the index of the leaf's 64-byte record behind the groups, for a
// leaf; an empty slot gets a box no ray can enter (lo = 65535, hi = 0) and the link of the record behind the last primitive's,
// whose box is NaN: should arithmetic ever let a ray into the slot, the exact test of that record turns it away.
struct Rec { uint32_t x, y, z, w; };
LT_HD inline bool slot_record(const float* lo, const float* hi, uint32_t link, const float* O, const float* S, Rec& r) {
  uint32_t ql[3], qh[3];
  bool ok = true;
  for (int k = 0; k < 3; k++) ok = quantise(lo[k], hi[k], O[k], S[k], ql[k], qh[k]) && ok;
  r.x = ql[0] | (ql[1] << 16);
  r.y = ql[2] | (qh[0] << 16);
  r.z = qh[1] | (qh[2] << 16);
  r.w = link;
  return ok;
}
LT_HD inline Rec empty_slot(uint32_t groups, uint32_t n_prims) { return Rec{0xffffffffu, 0x0000ffffu, 0u, 0x80000000u | (groups + n_prims)}; }

}  // namespace lt_own16
