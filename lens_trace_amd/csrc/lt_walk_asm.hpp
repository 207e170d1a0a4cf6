// lt_walk_asm.hpp -- the packet walks over the backend's own tree, hand-written for gfx950 (CDNA4).
//
// A packet = the 64 rays of a wavefront (an 8x8 pixel square's camera rays, or the shadow rays that leave its hit points), all
// finite, all in one direction-sign octant.  The wave walks the tree ONCE, node data arriving through scalar loads, each lane
// running the same slab test against its own ray.  What makes that exact is in lt_retree.hpp: over a tree whose boxes nest, a
// finite ray reaches a leaf of the reference's traversal iff it passes the slab test of the leaf's OWN box, in whatever order
// and through whatever enclosing boxes the candidates are enumerated.  So here
//
//   * interior nodes are tested CONSERVATIVELY and cheaply: a child-pair record (lt_own_pair_kernel, 64 bytes, one
//     s_load_dwordx16) holds the two children's boxes pushed outwards by 2^-21 of each bound, and the test is
//         tNear = max_a fma(near'_a, inv_a, -p_a),  tFar = min_a fma(far'_a, inv_a, -p_a),  tFar + M >= max(tNear, 0+)
//     with p_a = fl(o_a * inv_a) and M = 2^-19 * max|p_a| + 2^-140 per lane: 11 vector instructions per box instead of the 16 of
//     the reference's (bound - o) * inv form, and it accepts whenever that form does (proof below) -- a box entered needlessly
//     costs a few instructions, a box skipped wrongly would cost a pixel;
//   * a leaf gets the reference's own test: its record (the leaf's slot of the same array) carries the leaf's box bit for bit,
//     the re-tiled triangle and the primitive offset; the slab test (acc.cl:113-130 in octant form: tExit >= max(tEnter, 0+),
//     0+ = the smallest positive float, exact for every non-NaN input) and the triangle test (acc.cl:72-111: cross =
//     fma(a, b, -(c * d)), dot = fma chain + the `w` terms, IEEE 1 / det by the div_scale / rcp / fma / div_fmas / div_fixup
//     sequence hipcc emits for `1.0f / x` -- or, for the as-shipped math flavour (operand `fast`), the 2.5-ulp form the
//     reference's NULL build options give it) run with EXEC narrowed test by test (v_cmpx): what is left of EXEC at the end is
//     the mask of lanes that accept the hit;
//   * no order, no lane masks: every popped node is tested by every lane still in the walk (a lane outside the node's parent
//     cannot be inside the node), so a stack entry is ONE dword -- a node index, bit 31 set for a leaf -- and the stack is one
//     VGPR whose 64 lanes are its 64 slots (v_writelane / v_readlane with the stack pointer in M0): no LDS, no latency (the
//     register's 64 lanes are parked in the wave's LDS row around a walk that runs with lanes switched off: LT_ASM_WALK).  Pushes
//     are branch-free: write the child at the top, then add "some lane hit it" (SCC of s_cmp_lg_u64) to the pointer.  Depth:
//     at most two waiting entries per level plus four at the frontier, 2 * height + 2 <= 62 (the build caps the height at 30);
//   * two interior nodes per iteration when the stack holds two: their records are fetched together, halving the number of
//     dependent memory round trips, which is what this walk waits for (on the 1 M-triangle scene one record fetch in three
//     hits the scalar cache and two in three of the rest the XCD's L2);
//   * closest-hit walks (camera rays) keep the payload (t, u, v, primitive, hitType: RayPayload, acc.cl:55-61) in five VGPRs;
//     two accepted hits with bit-equal t are settled by the reference's leaf order (SceneDev::rank8); any-hit walks (shadow
//     rays: their callers read hitType only) drop a lane from `open` at its first accepted hit and end when `open` is empty.
//
// The conservative test accepts whenever the reference's does.  Per axis, with i = inv_a > 0 (the other sign mirrors), near
// bound lo, lo' <= lo - 6u|lo| (u = 2^-24; lt_own_pair_kernel subtracts 2^-21 |lo| and steps one float further down), all
// magnitudes below 2^101 so nothing overflows (|o|, |bound| < 2^40, |inv| < 2^60: checked per wave / per scene):
//     reference: tN = fl(fl(lo - o) i)            >= (lo - o) i - 2u (|lo| + |o|) i          (two roundings)
//     here:      tN' = fl(lo' i - p), p = fl(o i)  <= (lo - o) i - 6u |lo| i + u |o i| + u (|lo'| i + |p|)   (p's rounding, fma's)
//                                                  <= (lo - o) i - 4u |lo| i + 2.01 u |o i|
//  so tN' <= tN + 4.01 u |o i| <= tN + M/2 (M >= 32 u max|p|), and symmetrically tF' >= tF - M/2 for the far bound; hence
//  tF >= max(tN, 0+) implies tF' + M >= tF + M/2 >= max(tN', 0+).  (Gradual underflow adds at most 2^-148 per operation: the
//  2^-140 in M.)
//
// Hazards (none of the instructions below needs a manually inserted wait state on gfx950 beyond these): v_div_fmas reads the
// VCC written by the second v_div_scale four VALU instructions earlier; v_rcp's result is first read three instructions
// later; SGPRs written by VALU (v_cmp, v_readlane) are read by SALU / as SMEM offsets only; M0 (the lane select of
// v_readlane / v_writelane) is written by SALU only; an SMEM instruction reads its address operands at issue, so the offset
// register is reused right after.  EXEC and M0 are saved on entry and restored on exit.
#pragma once

namespace lt {

typedef unsigned long long lt_u64;

// ---- fixed scalar registers (all clobbered; operands live elsewhere; s32 / s33, the ABI's stack and frame pointers, are never
// touched) -------------------------------------------------------------------------------------------------------------------
// record 0 s[36:51], record 1 s[52:67]: interior node = [left box lo hi, reference, -][right box lo hi, reference, -];
// a reference with bit 31 set is a leaf (0x80000000 | node index), else an interior node's index
#define LT_R_REC0 "s[36:51]"
#define LT_R_REC1 "s[52:67]"
#define LT_R_REF0L "s42"
#define LT_R_REF0R "s50"
#define LT_R_REF1L "s58"
#define LT_R_REF1R "s66"
// leaf record, in record 0's registers: triangle A, e1 = B - A, e2 = C - A (lt_retile_kernel's arithmetic), the leaf's box, the
// primitive offset
#define LT_R_AX "s36"
#define LT_R_AY "s37"
#define LT_R_AZ "s38"
#define LT_R_E1X "s39"
#define LT_R_E1Y "s40"
#define LT_R_E1Z "s41"
#define LT_R_E2X "s42"
#define LT_R_E2Y "s43"
#define LT_R_E2Z "s44"
#define LT_R_PRIM "s51"
// (a kernel that keeps eight waves per SIMD owns s0 .. s71: everything below fits under that)
#define LT_R_HML "s[68:69]"    /* lanes that hit the child just tested; in the leaf test: v_cmpx destination, tie masks */
#define LT_R_HMR "s[70:71]"
#define LT_R_EXEC "s[16:17]"   /* EXEC on entry */
#define LT_R_TMPM "s[18:19]"
#define LT_R_TMPLO "s18"
#define LT_R_TMPHI "s19"
#define LT_R_CUR "s20"
#define LT_R_CUR2 "s21"
#define LT_R_M0 "s22"          /* M0 on entry */
#define LT_R_LEAF "s23"        /* scratch */
#define LT_ASM_CLOBBERS                                                                                                         \
  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", \
  "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23",                               \
  "vcc", "scc", "memory"

// conservative slab test of one child of an interior record (see the head of the file): 11 instructions
#define LT_ASM_BOXC(NX, NY, NZ, FX, FY, FZ, OUT)       \
  "v_fma_f32 %[t0], " NX ", %[ix], -%[px]\n"          \
  "v_fma_f32 %[t1], " NY ", %[iy], -%[py]\n"          \
  "v_fma_f32 %[t2], " NZ ", %[iz], -%[pz]\n"          \
  "v_max3_f32 %[t0], %[t0], %[t1], %[t2]\n"           \
  "v_fma_f32 %[t1], " FX ", %[ix], -%[px]\n"          \
  "v_fma_f32 %[t2], " FY ", %[iy], -%[py]\n"          \
  "v_fma_f32 %[t3], " FZ ", %[iz], -%[pz]\n"          \
  "v_min3_f32 %[t1], %[t1], %[t2], %[t3]\n"           \
  "v_add_f32_e32 %[t1], %[t1], %[mg]\n"               \
  "v_max_f32_e32 %[t0], 1, %[t0]\n"                   \
  "v_cmp_ge_f32_e64 " OUT ", %[t1], %[t0]\n"

// the reference's slab test of a leaf's own box (acc.cl:113-130, octant form), narrowing EXEC to the lanes that pass
#define LT_ASM_BOXX(NX, NY, NZ, FX, FY, FZ)            \
  "v_sub_f32_e32 %[t0], " NX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t1], " NY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t2], " NZ ", %[oz]\n"              \
  "v_mul_f32_e32 %[t0], %[t0], %[ix]\n"               \
  "v_mul_f32_e32 %[t1], %[t1], %[iy]\n"               \
  "v_mul_f32_e32 %[t2], %[t2], %[iz]\n"               \
  "v_max3_f32 %[t0], %[t0], %[t1], %[t2]\n"           \
  "v_sub_f32_e32 %[t1], " FX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t2], " FY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t3], " FZ ", %[oz]\n"              \
  "v_mul_f32_e32 %[t1], %[t1], %[ix]\n"               \
  "v_mul_f32_e32 %[t2], %[t2], %[iy]\n"               \
  "v_mul_f32_e32 %[t3], %[t3], %[iz]\n"               \
  "v_min3_f32 %[t1], %[t1], %[t2], %[t3]\n"           \
  "v_max_f32_e32 %[t0], 1, %[t0]\n"                   \
  "v_cmpx_ge_f32_e64 " LT_R_HML ", %[t1], %[t0]\n"

// ... the same test AFTER the triangle test (LT_ASM_WALK: a leaf's box is what the parent's conservative test has just let some
// lane through, so it hardly ever stops the whole wavefront, while most triangles are missed by every lane: the box test then
// only has to run for the lanes that hit the triangle, i.e. seldom).  t0 .. t3 hold invDet, u, v, t by then.
#define LT_ASM_BOXX_LATE(NX, NY, NZ, FX, FY, FZ)       \
  "v_sub_f32_e32 %[t4], " NX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t5], " NY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t6], " NZ ", %[oz]\n"              \
  "v_mul_f32_e32 %[t4], %[t4], %[ix]\n"               \
  "v_mul_f32_e32 %[t5], %[t5], %[iy]\n"               \
  "v_mul_f32_e32 %[t6], %[t6], %[iz]\n"               \
  "v_max3_f32 %[t4], %[t4], %[t5], %[t6]\n"           \
  "v_sub_f32_e32 %[t5], " FX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t6], " FY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t7], " FZ ", %[oz]\n"              \
  "v_mul_f32_e32 %[t5], %[t5], %[ix]\n"               \
  "v_mul_f32_e32 %[t6], %[t6], %[iy]\n"               \
  "v_mul_f32_e32 %[t7], %[t7], %[iz]\n"               \
  "v_min3_f32 %[t5], %[t5], %[t6], %[t7]\n"           \
  "v_max_f32_e32 %[t4], 1, %[t4]\n"                   \
  "v_cmpx_ge_f32_e64 " LT_R_HML ", %[t5], %[t4]\n"
#define LT_ASM_BOXX_G_LATE(LX, LY, LZ, HX, HY, HZ)     \
  "v_sub_f32_e32 %[t4], " LX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t5], " HX ", %[ox]\n"              \
  "v_mul_f32_e32 %[t4], %[t4], %[ix]\n"               \
  "v_mul_f32_e32 %[t5], %[t5], %[ix]\n"               \
  "v_min_f32_e32 %[t6], %[t4], %[t5]\n"               \
  "v_max_f32_e32 %[t7], %[t4], %[t5]\n"               \
  "v_sub_f32_e32 %[t4], " LY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t5], " HY ", %[oy]\n"              \
  "v_mul_f32_e32 %[t4], %[t4], %[iy]\n"               \
  "v_mul_f32_e32 %[t5], %[t5], %[iy]\n"               \
  "v_min_f32_e32 %[t8], %[t4], %[t5]\n"               \
  "v_max_f32_e32 %[t9], %[t4], %[t5]\n"               \
  "v_sub_f32_e32 %[t4], " LZ ", %[oz]\n"              \
  "v_sub_f32_e32 %[t5], " HZ ", %[oz]\n"              \
  "v_mul_f32_e32 %[t4], %[t4], %[iz]\n"               \
  "v_mul_f32_e32 %[t5], %[t5], %[iz]\n"               \
  "v_min_f32_e32 %[t10], %[t4], %[t5]\n"              \
  "v_max_f32_e32 %[t0], %[t4], %[t5]\n"               \
  "v_max3_f32 %[t6], %[t6], %[t8], %[t10]\n"          \
  "v_min3_f32 %[t7], %[t7], %[t9], %[t0]\n"           \
  "v_max_f32_e32 %[t6], 1, %[t6]\n"                   \
  "v_cmpx_ge_f32_e64 " LT_R_HML ", %[t7], %[t6]\n"

// Box register triples: record 0 left s36-38 / s39-41, right s44-46 / s47-49; record 1 left s52-54 / s55-57, right s60-62 /
// s63-65; leaf box s45-47 / s48-50.
#define LT_LOHI_0 "s36", "s37", "s38", "s39", "s40", "s41"
#define LT_LOHI_1 "s44", "s45", "s46", "s47", "s48", "s49"
#define LT_LOHI_2 "s52", "s53", "s54", "s55", "s56", "s57"
#define LT_LOHI_3 "s60", "s61", "s62", "s63", "s64", "s65"
#define LT_LOHI_LEAF "s45", "s46", "s47", "s48", "s49", "s50"
// ... and for a wave whose rays do not share an octant (shadow rays around a light overhead): the same two tests with each
// axis' near / far plane picked per lane, min / max of the two products (fma and (bound - o) * inv are monotonic in the bound and
// lo <= hi, so the smaller product is the near plane's): 17 and 22 instructions
#define LT_ASM_BOXC_G(LX, LY, LZ, HX, HY, HZ, OUT)     \
  "v_fma_f32 %[t0], " LX ", %[ix], -%[px]\n"          \
  "v_fma_f32 %[t1], " HX ", %[ix], -%[px]\n"          \
  "v_fma_f32 %[t2], " LY ", %[iy], -%[py]\n"          \
  "v_fma_f32 %[t3], " HY ", %[iy], -%[py]\n"          \
  "v_fma_f32 %[t4], " LZ ", %[iz], -%[pz]\n"          \
  "v_fma_f32 %[t5], " HZ ", %[iz], -%[pz]\n"          \
  "v_min_f32_e32 %[t6], %[t0], %[t1]\n"               \
  "v_min_f32_e32 %[t7], %[t2], %[t3]\n"               \
  "v_min_f32_e32 %[t8], %[t4], %[t5]\n"               \
  "v_max_f32_e32 %[t0], %[t0], %[t1]\n"               \
  "v_max_f32_e32 %[t2], %[t2], %[t3]\n"               \
  "v_max_f32_e32 %[t4], %[t4], %[t5]\n"               \
  "v_max3_f32 %[t6], %[t6], %[t7], %[t8]\n"           \
  "v_min3_f32 %[t0], %[t0], %[t2], %[t4]\n"           \
  "v_add_f32_e32 %[t0], %[t0], %[mg]\n"               \
  "v_max_f32_e32 %[t6], 1, %[t6]\n"                   \
  "v_cmp_ge_f32_e64 " OUT ", %[t0], %[t6]\n"
#define LT_ASM_BOXX_G(LX, LY, LZ, HX, HY, HZ)          \
  "v_sub_f32_e32 %[t0], " LX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t1], " HX ", %[ox]\n"              \
  "v_sub_f32_e32 %[t2], " LY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t3], " HY ", %[oy]\n"              \
  "v_sub_f32_e32 %[t4], " LZ ", %[oz]\n"              \
  "v_sub_f32_e32 %[t5], " HZ ", %[oz]\n"              \
  "v_mul_f32_e32 %[t0], %[t0], %[ix]\n"               \
  "v_mul_f32_e32 %[t1], %[t1], %[ix]\n"               \
  "v_mul_f32_e32 %[t2], %[t2], %[iy]\n"               \
  "v_mul_f32_e32 %[t3], %[t3], %[iy]\n"               \
  "v_mul_f32_e32 %[t4], %[t4], %[iz]\n"               \
  "v_mul_f32_e32 %[t5], %[t5], %[iz]\n"               \
  "v_min_f32_e32 %[t6], %[t0], %[t1]\n"               \
  "v_min_f32_e32 %[t7], %[t2], %[t3]\n"               \
  "v_min_f32_e32 %[t8], %[t4], %[t5]\n"               \
  "v_max_f32_e32 %[t0], %[t0], %[t1]\n"               \
  "v_max_f32_e32 %[t2], %[t2], %[t3]\n"               \
  "v_max_f32_e32 %[t4], %[t4], %[t5]\n"               \
  "v_max3_f32 %[t6], %[t6], %[t7], %[t8]\n"           \
  "v_min3_f32 %[t0], %[t0], %[t2], %[t4]\n"           \
  "v_max_f32_e32 %[t6], 1, %[t6]\n"                   \
  "v_cmpx_ge_f32_e64 " LT_R_HML ", %[t0], %[t6]\n"
// What LT_ASM_WALK takes as its first argument, LT_NF_<octant> or LT_NF_G, names a pair of tests: for the octant NEG the near
// plane of axis a is the box's max when direction component a is negative (bit a of NEG).
#define LT_BC_LT_NF_0(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(LX, LY, LZ, HX, HY, HZ, OUT)
#define LT_BX_LT_NF_0(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(LX, LY, LZ, HX, HY, HZ)
#define LT_BXL_LT_NF_0(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(LX, LY, LZ, HX, HY, HZ)
#define LT_BC_LT_NF_1(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(HX, LY, LZ, LX, HY, HZ, OUT)
#define LT_BX_LT_NF_1(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(HX, LY, LZ, LX, HY, HZ)
#define LT_BXL_LT_NF_1(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(HX, LY, LZ, LX, HY, HZ)
#define LT_BC_LT_NF_2(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(LX, HY, LZ, HX, LY, HZ, OUT)
#define LT_BX_LT_NF_2(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(LX, HY, LZ, HX, LY, HZ)
#define LT_BXL_LT_NF_2(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(LX, HY, LZ, HX, LY, HZ)
#define LT_BC_LT_NF_3(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(HX, HY, LZ, LX, LY, HZ, OUT)
#define LT_BX_LT_NF_3(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(HX, HY, LZ, LX, LY, HZ)
#define LT_BXL_LT_NF_3(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(HX, HY, LZ, LX, LY, HZ)
#define LT_BC_LT_NF_4(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(LX, LY, HZ, HX, HY, LZ, OUT)
#define LT_BX_LT_NF_4(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(LX, LY, HZ, HX, HY, LZ)
#define LT_BXL_LT_NF_4(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(LX, LY, HZ, HX, HY, LZ)
#define LT_BC_LT_NF_5(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(HX, LY, HZ, LX, HY, LZ, OUT)
#define LT_BX_LT_NF_5(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(HX, LY, HZ, LX, HY, LZ)
#define LT_BXL_LT_NF_5(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(HX, LY, HZ, LX, HY, LZ)
#define LT_BC_LT_NF_6(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(LX, HY, HZ, HX, LY, LZ, OUT)
#define LT_BX_LT_NF_6(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(LX, HY, HZ, HX, LY, LZ)
#define LT_BXL_LT_NF_6(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(LX, HY, HZ, HX, LY, LZ)
#define LT_BC_LT_NF_7(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC(HX, HY, HZ, LX, LY, LZ, OUT)
#define LT_BX_LT_NF_7(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX(HX, HY, HZ, LX, LY, LZ)
#define LT_BXL_LT_NF_7(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_LATE(HX, HY, HZ, LX, LY, LZ)
#define LT_BC_LT_NF_G(LX, LY, LZ, HX, HY, HZ, OUT) LT_ASM_BOXC_G(LX, LY, LZ, HX, HY, HZ, OUT)
#define LT_BX_LT_NF_G(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_G(LX, LY, LZ, HX, HY, HZ)
#define LT_BXL_LT_NF_G(LX, LY, LZ, HX, HY, HZ) LT_ASM_BOXX_G_LATE(LX, LY, LZ, HX, HY, HZ)
#define LT_BC(NF, LOHI, OUT) LT_BC_##NF(LOHI, OUT)
#define LT_BX(NF, LOHI) LT_BX_##NF(LOHI)
#define LT_BXL(NF, LOHI) LT_BXL_##NF(LOHI)

// branch-free push of a child reference: written at the top, kept iff some lane hit the child
#define LT_ASM_PUSH(REF, HM)                        \
  "v_writelane_b32 %[stk], " REF ", m0\n"           \
  "s_cmp_lg_u64 " HM ", 0\n"                        \
  "s_addc_u32 m0, m0, 0\n"

// det, 1 / det, u (EXEC narrowed by the det and u tests); leaves: t0 = invDet, t1 = u, t7 t8 t9 = tvec
#define LT_ASM_TRI_PART1                                                                                                        \
  "v_mul_f32_e64 %[t4], %[dz], -" LT_R_E2Y "\n"     /* pvec = cross(d, e2) */                                                   \
  "v_mul_f32_e64 %[t5], %[dx], -" LT_R_E2Z "\n"                                                                                 \
  "v_fmac_f32_e32 %[t4], " LT_R_E2Z ", %[dy]\n"                                                                                 \
  "v_fmac_f32_e32 %[t5], " LT_R_E2X ", %[dz]\n"                                                                                 \
  "v_mul_f32_e64 %[t6], %[dy], -" LT_R_E2X "\n"                                                                                 \
  "v_mul_f32_e32 %[t0], " LT_R_E1X ", %[t4]\n"                                                                                  \
  "v_fmac_f32_e32 %[t6], " LT_R_E2Y ", %[dx]\n"                                                                                 \
  "v_fmac_f32_e32 %[t0], " LT_R_E1Y ", %[t5]\n"                                                                                 \
  "v_fmac_f32_e32 %[t0], " LT_R_E1Z ", %[t6]\n"                                                                                 \
  "v_add_f32_e32 %[t0], 0, %[t0]\n"                 /* det */                                                                   \
  "s_cmp_lg_u32 %[fast], 0\n"                                                                                                   \
  "s_cbranch_scc1 .LfastRcp%=\n"                                                                                                \
  "v_div_scale_f32 %[t1], " LT_R_HML ", %[t0], %[t0], 1.0\n"                                                                    \
  "v_rcp_f32_e32 %[t2], %[t1]\n"                                                                                                \
  "v_cmpx_nlt_f32_e64 " LT_R_HML ", |%[t0]|, %[eps]\n" /* !(fabs(det) < epsilon) */                                             \
  "v_subrev_f32_e32 %[t7], " LT_R_AX ", %[ox]\n"    /* tvec = o - A */                                                          \
  "v_subrev_f32_e32 %[t8], " LT_R_AY ", %[oy]\n"                                                                                \
  "v_fma_f32 %[t3], -%[t1], %[t2], 1.0\n"                                                                                       \
  "v_fmac_f32_e32 %[t2], %[t3], %[t2]\n"                                                                                        \
  "v_div_scale_f32 %[t3], vcc, 1.0, %[t0], 1.0\n"                                                                               \
  "v_mul_f32_e32 %[t9], %[t3], %[t2]\n"                                                                                         \
  "v_fma_f32 %[t10], -%[t1], %[t9], %[t3]\n"                                                                                    \
  "v_fmac_f32_e32 %[t9], %[t10], %[t2]\n"                                                                                       \
  "v_fma_f32 %[t1], -%[t1], %[t9], %[t3]\n"                                                                                     \
  "v_div_fmas_f32 %[t1], %[t1], %[t2], %[t9]\n"                                                                                 \
  "v_div_fixup_f32 %[t0], %[t1], %[t0], 1.0\n"      /* invDet = 1 / det, correctly rounded */                                   \
  "s_branch .LrcpDone%=\n"                                                                                                      \
  ".LfastRcp%=:\n"                                  /* the as-shipped build's 1 / det: ldexp(rcp(frexp_mant), -frexp_exp) */    \
  "v_frexp_mant_f32_e32 %[t1], %[t0]\n"                                                                                         \
  "v_rcp_f32_e32 %[t1], %[t1]\n"                                                                                                \
  "v_cmpx_nlt_f32_e64 " LT_R_HML ", |%[t0]|, %[eps]\n"                                                                          \
  "v_subrev_f32_e32 %[t7], " LT_R_AX ", %[ox]\n"                                                                                \
  "v_subrev_f32_e32 %[t8], " LT_R_AY ", %[oy]\n"                                                                                \
  "v_frexp_exp_i32_f32_e32 %[t2], %[t0]\n"                                                                                      \
  "v_sub_u32_e32 %[t2], 0, %[t2]\n"                                                                                             \
  "v_ldexp_f32 %[t0], %[t1], %[t2]\n"                                                                                           \
  ".LrcpDone%=:\n"                                                                                                              \
  "v_subrev_f32_e32 %[t9], " LT_R_AZ ", %[oz]\n"                                                                                \
  "v_mul_f32_e32 %[t1], %[t7], %[t4]\n"                                                                                         \
  "v_fmac_f32_e32 %[t1], %[t8], %[t5]\n"                                                                                        \
  "v_fmac_f32_e32 %[t1], %[t9], %[t6]\n"                                                                                        \
  "v_add_f32_e32 %[t1], 0, %[t1]\n"                                                                                             \
  "v_mul_f32_e32 %[t1], %[t1], %[t0]\n"             /* u */                                                                     \
  "v_cmpx_ngt_f32_e64 " LT_R_HML ", 0, %[t1]\n"     /* !(u < 0) */                                                              \
  "v_cmpx_nlt_f32_e64 " LT_R_HML ", 1.0, %[t1]\n"   /* !(u > 1) */                                                              \
  "s_cbranch_execz .LleafEnd%=\n"

// v, u + v, t (EXEC narrowed by the v and u + v tests); leaves: t1 = u, t2 = v, t3 = t
#define LT_ASM_TRI_PART2                                                                                                        \
  "v_mul_f32_e64 %[t4], %[t9], -" LT_R_E1Y "\n"     /* qvec = cross(tvec, e1) */                                                \
  "v_fmac_f32_e32 %[t4], " LT_R_E1Z ", %[t8]\n"                                                                                 \
  "v_mul_f32_e64 %[t5], %[t7], -" LT_R_E1Z "\n"                                                                                 \
  "v_mul_f32_e64 %[t6], %[t8], -" LT_R_E1X "\n"                                                                                 \
  "v_fmac_f32_e32 %[t5], " LT_R_E1X ", %[t9]\n"                                                                                 \
  "v_fmac_f32_e32 %[t6], " LT_R_E1Y ", %[t7]\n"                                                                                 \
  "v_mul_f32_e32 %[t2], %[dx], %[t4]\n"                                                                                         \
  "v_fmac_f32_e32 %[t2], %[dy], %[t5]\n"                                                                                        \
  "v_fmac_f32_e32 %[t2], %[dz], %[t6]\n"                                                                                        \
  "v_fmac_f32_e32 %[t2], 0, %[dw]\n"                                                                                            \
  "v_mul_f32_e32 %[t2], %[t2], %[t0]\n"             /* v */                                                                     \
  "v_add_f32_e32 %[t10], %[t1], %[t2]\n"            /* u + v */                                                                 \
  "v_mul_f32_e32 %[t3], " LT_R_E2X ", %[t4]\n"                                                                                  \
  "v_fmac_f32_e32 %[t3], " LT_R_E2Y ", %[t5]\n"                                                                                 \
  "v_fmac_f32_e32 %[t3], " LT_R_E2Z ", %[t6]\n"                                                                                 \
  "v_cmpx_ngt_f32_e64 " LT_R_HML ", 0, %[t2]\n"     /* !(v < 0) */                                                              \
  "v_add_f32_e32 %[t3], 0, %[t3]\n"                                                                                             \
  "v_cmpx_nlt_f32_e64 " LT_R_HML ", 1.0, %[t10]\n"  /* !(u + v > 1) */                                                          \
  "v_mul_f32_e32 %[t3], %[t3], %[t0]\n"             /* t */

// what a closest-hit walk does with a triangle's t (EXEC = the lanes that passed every other test)
#define LT_ASM_ACCEPT_CLOSEST(ROFF)                                                                                             \
  "v_cmp_eq_f32_e64 " LT_R_HMR ", %[t3], %[pt]\n"   /* the same t again, bit for bit? (rare) */                                 \
  "v_cmp_lt_f32_e64 " LT_R_HML ", %[t3], %[pt]\n"   /* t < payload.t (no t > 0 test in the reference) */                        \
  "s_cmp_lg_u64 " LT_R_HMR ", 0\n"                                                                                              \
  "s_cbranch_scc0 .Ltake%=\n"                                                                                                   \
  /* Two triangles, one t: the reference keeps the one its own depth-first order meets first; this walk has no order, so it     \
     asks the table (SceneDev::rank8: 8 ranks per primitive, one per direction-sign octant). */                                 \
  "s_mov_b64 " LT_R_TMPM ", exec\n"                                                                                             \
  "s_mov_b64 exec, " LT_R_HMR "\n"                                                                                              \
  "v_cmpx_eq_u32_e64 " LT_R_HMR ", 1, %[phit]\n"    /* ... against a hit the lane already holds */                              \
  "v_lshlrev_b32_e32 %[t4], 5, %[pprim]\n"                                                                                      \
  "global_load_dword %[t5], %[t4], %[ranks] offset:" ROFF "\n"                                                                  \
  "s_lshl_b32 " LT_R_LEAF ", " LT_R_PRIM ", 5\n"                                                                                \
  "s_load_dword " LT_R_LEAF ", %[ranks], " LT_R_LEAF " offset:" ROFF "\n"                                                       \
  "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                                             \
  "v_cmpx_gt_u32_e64 " LT_R_HMR ", %[t5], " LT_R_LEAF "\n" /* this triangle's leaf comes first */                               \
  "s_or_b64 " LT_R_HML ", " LT_R_HML ", exec\n"                                                                                 \
  "s_mov_b64 exec, " LT_R_TMPM "\n"                                                                                             \
  ".Ltake%=:\n"                                                                                                                 \
  "s_and_b64 exec, exec, " LT_R_HML "\n"                                                                                        \
  "v_mov_b32_e32 %[pt], %[t3]\n"                    /* the lanes still in EXEC take the hit */                                  \
  "v_mov_b32_e32 %[pu], %[t1]\n"                                                                                                \
  "v_mov_b32_e32 %[pv], %[t2]\n"                                                                                                \
  "v_mov_b32_e32 %[pprim], " LT_R_PRIM "\n"                                                                                     \
  "v_mov_b32_e32 %[phit], 1\n"

// ... and an any-hit walk: the lanes that accept are done; the walk ends when nobody is left
#define LT_ASM_ACCEPT_ANYHIT                                                                                                    \
  "v_cmpx_lt_f32_e64 " LT_R_HML ", %[t3], %[tmax]\n" /* t < payload.t: accepted */                                              \
  "s_andn2_b64 %[open], %[open], exec\n"            /* those lanes are done; SCC = anyone still looking */                      \
  "s_cbranch_scc0 .Ldone%=\n"

// The walk.  LIVE = the register pair holding the lanes that take part (EXEC on entry for a closest-hit walk, `open` for an
// any-hit walk); IGNORE = the any-hit walks' "not the primitive the ray starts on" (acc.cl:188); ACCEPT as above.
#define LT_ASM_WALK(NF, LIVE, IGNORE, ACCEPT)                                                                                   \
  "s_mov_b64 " LT_R_EXEC ", exec\n"                                                                                             \
  "s_mov_b32 " LT_R_M0 ", m0\n"                                                                                                 \
  /* v_writelane ignores EXEC: a stack slot is a LANE of the stack register, and where that lane is switched off (a pixel that  \
     ended at a light, a pixel outside the image) the same physical register may hold a live value of that lane's own path --   \
     the compiler allocates registers per thread.  So, unless every lane is on, all 64 lanes of the register are parked in the  \
     wave's LDS row for the duration of the walk (ds_write_addtid: address = M0 + 4 * lane, no address register needed). */     \
  "s_cmp_eq_u64 exec, -1\n"                                                                                                     \
  "s_cbranch_scc1 .Lsaved%=\n"                                                                                                  \
  "s_mov_b32 m0, %[ldsrow]\n"                                                                                                   \
  "s_mov_b64 exec, -1\n"                                                                                                        \
  "ds_write_addtid_b32 %[stk]\n"                                                                                                \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "s_mov_b64 exec, " LT_R_EXEC "\n"                                                                                             \
  ".Lsaved%=:\n"                                                                                                                \
  "s_mov_b32 m0, 0\n"                               /* the stack pointer lives in M0 */                                         \
  "s_mov_b32 " LT_R_CUR ", 0\n"                     /* the root: interior node 0 */                                             \
  "s_branch .Lnode%=\n"                                                                                                         \
  ".Lpop%=:\n"                                                                                                                  \
  "s_cmp_eq_u32 m0, 0\n"                                                                                                        \
  "s_cbranch_scc1 .Ldone%=\n"                                                                                                   \
  "s_sub_u32 m0, m0, 1\n"                                                                                                       \
  "v_readlane_b32 " LT_R_CUR ", %[stk], m0\n"                                                                                   \
  "s_cmp_lt_i32 " LT_R_CUR ", 0\n"                                                                                              \
  "s_cbranch_scc1 .Lleaf%=\n"                                                                                                   \
  ".Lnode%=:\n"                                                                                                                 \
  "s_lshl_b32 " LT_R_TMPLO ", " LT_R_CUR ", 6\n"                                                                                \
  "s_load_dwordx16 " LT_R_REC0 ", %[pairs], " LT_R_TMPLO "\n"                                                                   \
  "s_cmp_eq_u32 m0, 0\n"                            /* a second interior node to fetch beside it? */                            \
  "s_cbranch_scc1 .Lone%=\n"                                                                                                    \
  "s_sub_u32 m0, m0, 1\n"                                                                                                       \
  "v_readlane_b32 " LT_R_CUR2 ", %[stk], m0\n"                                                                                  \
  "s_cmp_lt_i32 " LT_R_CUR2 ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LoneBack%=\n"                    /* a leaf on top: it stays */                                               \
  "s_lshl_b32 " LT_R_TMPHI ", " LT_R_CUR2 ", 6\n"                                                                               \
  "s_load_dwordx16 " LT_R_REC1 ", %[pairs], " LT_R_TMPHI "\n"                                                                   \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "s_mov_b64 exec, " LIVE "\n"                                                                                                  \
  LT_BC(NF, LT_LOHI_0, LT_R_HML)                                                                                               \
  LT_BC(NF, LT_LOHI_1, LT_R_HMR)                                                                                               \
  LT_ASM_PUSH(LT_R_REF0L, LT_R_HML)                                                                                             \
  LT_ASM_PUSH(LT_R_REF0R, LT_R_HMR)                                                                                             \
  LT_BC(NF, LT_LOHI_2, LT_R_HML)                                                                                               \
  LT_BC(NF, LT_LOHI_3, LT_R_HMR)                                                                                               \
  LT_ASM_PUSH(LT_R_REF1L, LT_R_HML)                                                                                             \
  LT_ASM_PUSH(LT_R_REF1R, LT_R_HMR)                                                                                             \
  "s_branch .Lpop%=\n"                                                                                                          \
  ".LoneBack%=:\n"                                                                                                              \
  "s_add_u32 m0, m0, 1\n"                                                                                                       \
  ".Lone%=:\n"                                                                                                                  \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "s_mov_b64 exec, " LIVE "\n"                                                                                                  \
  LT_BC(NF, LT_LOHI_0, LT_R_HML)                                                                                               \
  LT_BC(NF, LT_LOHI_1, LT_R_HMR)                                                                                               \
  LT_ASM_PUSH(LT_R_REF0L, LT_R_HML)                                                                                             \
  LT_ASM_PUSH(LT_R_REF0R, LT_R_HMR)                                                                                             \
  "s_branch .Lpop%=\n"                                                                                                          \
  ".Lleaf%=:\n"                                                                                                                 \
  "s_lshl_b32 " LT_R_TMPLO ", " LT_R_CUR ", 6\n"    /* (bit 31 falls off the 32-bit byte offset) */                             \
  "s_load_dwordx16 " LT_R_REC0 ", %[pairs], " LT_R_TMPLO "\n"                                                                   \
  "s_mov_b64 exec, " LIVE "\n"                                                                                                  \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  IGNORE                                                                                                                        \
  LT_ASM_TRI_PART1                                                                                                              \
  LT_ASM_TRI_PART2                                                                                                              \
  "s_cbranch_execz .LleafEnd%=\n"                                                                                               \
  LT_BXL(NF, LT_LOHI_LEAF)                          /* the reference's own test of the leaf's own box, for the lanes that hit the triangle */ \
  ACCEPT                                                                                                                        \
  ".LleafEnd%=:\n"                                                                                                              \
  "s_branch .Lpop%=\n"                                                                                                          \
  ".Ldone%=:\n"                                                                                                                 \
  "s_cmp_eq_u64 " LT_R_EXEC ", -1\n"                                                                                            \
  "s_cbranch_scc1 .Lrestored%=\n"                                                                                               \
  "s_mov_b32 m0, %[ldsrow]\n"                                                                                                   \
  "s_mov_b64 exec, -1\n"                                                                                                        \
  "ds_read_addtid_b32 %[stk]\n"                     /* every lane of the stack register as it was */                            \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  ".Lrestored%=:\n"                                                                                                             \
  "s_mov_b32 m0, " LT_R_M0 "\n"                                                                                                 \
  "s_mov_b64 exec, " LT_R_EXEC "\n"

#define LT_ASM_IGNORE_ANYHIT "v_cmpx_ne_u32_e64 " LT_R_HML ", " LT_R_PRIM ", %[ign]\n"

// Per-ray constants of the conservative test.
struct PacketRay {
  float px, py, pz, mg;
};
__device__ __forceinline__ PacketRay packet_ray(float ox, float oy, float oz, float ix, float iy, float iz) {
  PacketRay r;
  r.px = ox * ix;
  r.py = oy * iy;
  r.pz = oz * iz;
  r.mg = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.px), __builtin_fabsf(r.py)), __builtin_fabsf(r.pz)) * 0x1p-19f + 0x1p-140f;
  return r;
}

// The any-hit walk: on return `open` holds the lanes that found no occluder.  NEG = the direction-sign octant all rays of the wave
// share; eps = the program's triangle epsilon as the float the reference's double compare amounts to (intersect_triangle_data).
template <int NEG>
__device__ __forceinline__ lt_u64 packet_anyhit_walk(const void* pairs, float ox, float oy, float oz, float ix, float iy, float iz, float dx,
                                                     float dy, float dz, float dw, float tmax, int ign, float eps, uint32_t fast, lt_u64 open,
                                                     uint32_t ldsrow) {
  const PacketRay pr = packet_ray(ox, oy, oz, ix, iy, iz);
  int stk = 0;
  float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
#define LT_ANYHIT_INSTANCE(NF)                                                                                                           \
  asm volatile(LT_ASM_WALK(NF, "%[open]", LT_ASM_IGNORE_ANYHIT, LT_ASM_ACCEPT_ANYHIT)                                                    \
               : [open] "+s"(open), [stk] "+v"(stk), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4),     \
                 [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9), [t10] "=&v"(t10)                        \
               : [pairs] "s"(pairs), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [px] "v"(pr.px),  \
                 [py] "v"(pr.py), [pz] "v"(pr.pz), [mg] "v"(pr.mg), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [dw] "v"(dw),                \
                 [tmax] "v"(tmax), [ign] "v"(ign), [eps] "s"(eps), [fast] "s"(fast), [ldsrow] "s"(ldsrow)                                 \
               : LT_ASM_CLOBBERS)
  if constexpr (NEG == 0) LT_ANYHIT_INSTANCE(LT_NF_0);
  else if constexpr (NEG == 1) LT_ANYHIT_INSTANCE(LT_NF_1);
  else if constexpr (NEG == 2) LT_ANYHIT_INSTANCE(LT_NF_2);
  else if constexpr (NEG == 3) LT_ANYHIT_INSTANCE(LT_NF_3);
  else if constexpr (NEG == 4) LT_ANYHIT_INSTANCE(LT_NF_4);
  else if constexpr (NEG == 5) LT_ANYHIT_INSTANCE(LT_NF_5);
  else if constexpr (NEG == 6) LT_ANYHIT_INSTANCE(LT_NF_6);
  else if constexpr (NEG == 7) LT_ANYHIT_INSTANCE(LT_NF_7);
  else LT_ANYHIT_INSTANCE(LT_NF_G);   // NEG < 0: the rays' direction signs differ
#undef LT_ANYHIT_INSTANCE
  return open;
}

// The closest-hit walk (camera rays).
template <int NEG>
__device__ __forceinline__ void packet_closest_walk(const void* pairs, const void* ranks, float ox, float oy, float oz, float ix, float iy,
                                                    float iz, float dx, float dy, float dz, float dw, float eps, uint32_t fast, float& pt,
                                                    float& pu, float& pv, int& pprim, int& phit, uint32_t ldsrow) {
  const PacketRay pr = packet_ray(ox, oy, oz, ix, iy, iz);
  int stk = 0;
  float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
#define LT_CLOSEST_INSTANCE(NF, ROFF)                                                                                                    \
  asm volatile(LT_ASM_WALK(NF, LT_R_EXEC, "", LT_ASM_ACCEPT_CLOSEST(ROFF))                                                               \
               : [pt] "+v"(pt), [pu] "+v"(pu), [pv] "+v"(pv), [pprim] "+v"(pprim), [phit] "+v"(phit), [stk] "+v"(stk), [t0] "=&v"(t0),   \
                 [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7),         \
                 [t8] "=&v"(t8), [t9] "=&v"(t9), [t10] "=&v"(t10)                                                                        \
               : [pairs] "s"(pairs), [ranks] "s"(ranks), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy),            \
                 [iz] "v"(iz), [px] "v"(pr.px), [py] "v"(pr.py), [pz] "v"(pr.pz), [mg] "v"(pr.mg), [dx] "v"(dx), [dy] "v"(dy),            \
                 [dz] "v"(dz), [dw] "v"(dw), [eps] "s"(eps), [fast] "s"(fast), [ldsrow] "s"(ldsrow)                                       \
               : LT_ASM_CLOBBERS)
  if constexpr (NEG == 0) LT_CLOSEST_INSTANCE(LT_NF_0, "0");
  else if constexpr (NEG == 1) LT_CLOSEST_INSTANCE(LT_NF_1, "4");
  else if constexpr (NEG == 2) LT_CLOSEST_INSTANCE(LT_NF_2, "8");
  else if constexpr (NEG == 3) LT_CLOSEST_INSTANCE(LT_NF_3, "12");
  else if constexpr (NEG == 4) LT_CLOSEST_INSTANCE(LT_NF_4, "16");
  else if constexpr (NEG == 5) LT_CLOSEST_INSTANCE(LT_NF_5, "20");
  else if constexpr (NEG == 6) LT_CLOSEST_INSTANCE(LT_NF_6, "24");
  else LT_CLOSEST_INSTANCE(LT_NF_7, "28");
#undef LT_CLOSEST_INSTANCE
}

}  // namespace lt
