// lt_walk_asm.hpp -- the packet walks below the root node, hand-written for gfx950 (CDNA4).
//
// The packet walks (lt_device.hpp: traverse_packet_pairs for camera rays, traverse_packet_pairs_anyhit for shadow rays) were
// bound by SCALAR instruction issue as hipcc compiled them: rocprofv3 on the bench workload showed 0.75 scalar instructions
// per CU-cycle against 0.62 of the VALU peak (profiles/r2/before_asm_issue_profile.json).  hipcc turns their wave-uniform,
// multi-exit loops into a flag-driven state machine (structurised control flow: s_mov -1 / s_andn2 exec / s_cbranch_vccnz
// chains, a loop-exit selector register, copies of every loop-carried mask and of the whole hit payload at each join): 25-30
// scalar and ~10 wasted vector instructions per visited pair of nodes, where the algorithm needs about a dozen scalar ones.
// This file is that dozen:
//
//   * one `s_load_dwordx16` per interior node brings the 64-byte child-pair record (lt_pair_kernel) into SGPRs;
//   * the wave's lane mask of the node becomes EXEC for the two slab tests, so each test ends in ONE `v_cmp_ge_f32` whose
//     SGPR-pair result already is "lanes of this node that hit the child" -- no s_and with the node's mask -- and the
//     reference's two conditions `tEnter <= tExit && tExit > 0` (acc.cl:113-130, in box_mask<NEG>'s octant form) fold into
//     `tExit >= max(tEnter, 0x00000001)`: the smallest positive float (denormals are kept: .amdhsa_float_denorm_mode_32 3)
//     stands for "> 0", exact for every non-NaN input, and a packet is only formed from rays that cannot produce a NaN
//     (traverse_camera / traverse: all origins and inverse directions finite);
//   * branches test SCC straight from the mask arithmetic; the wave-uniform stack (child reference + 64-bit lane mask per
//     entry, 16 bytes apart in the wave's LDS) is written / read with three ds_*_b32 of identical data per lane;
//   * the triangle test (acc.cl:72-111 on the re-tiled 48-byte triangle: cross = fma(a, b, -(c * d)), dot = fma chain + the
//     `w` terms, IEEE 1 / det by the div_scale / rcp / fma / div_fmas / div_fixup sequence hipcc emits for `1.0f / x` -- or,
//     for the as-shipped math flavour (operand `fast`), the 2.5-ulp form the reference's NULL build options give it -- each
//     reject written as the reference's negated compare) runs with EXEC = the lanes that reached the leaf, and every test
//     NARROWS EXEC (v_cmpx): what is left of EXEC at the end is the mask of lanes that accept the hit;
//
// Hazards (none of the instructions below needs a manually inserted wait state on gfx950 beyond these): v_div_fmas reads the
// VCC written by the second v_div_scale four VALU instructions earlier; v_rcp's result is first read three instructions
// later; SGPRs written by VALU (v_cmp, v_readfirstlane) are read by SALU / as SMEM offsets only (no VMEM, no v_readlane lane
// select, no DPP); DS results are waited for with s_waitcnt lgkmcnt(0) before v_readfirstlane; an SMEM instruction reads its
// address operands at issue, so the offset register is reused right after.  EXEC is saved on entry and restored on exit.
#pragma once

namespace lt {

typedef unsigned long long lt_u64;

// ---- fixed scalar registers (all clobbered; operands live elsewhere) -----------------------------------------------------
// child-pair record s[36:51] (lt_pair_kernel): a reference with bit 31 set is a leaf (0x80000000 | primitive offset), else an
// interior node (index | split axis << 29)
#define LT_R_REC "s[36:51]"
#define LT_R_LMINX "s36"
#define LT_R_LMINY "s37"
#define LT_R_LMINZ "s38"
#define LT_R_LMAXX "s39"
#define LT_R_LMAXY "s40"
#define LT_R_LMAXZ "s41"
#define LT_R_REFL "s42"
#define LT_R_RMINX "s44"
#define LT_R_RMINY "s45"
#define LT_R_RMINZ "s46"
#define LT_R_RMAXX "s47"
#define LT_R_RMAXY "s48"
#define LT_R_RMAXZ "s49"
#define LT_R_REFR "s50"
// triangle (lt_retile_kernel): A, e1 = B - A, e2 = C - A -- in the record's registers: a record is dead once its two child
// references and hit masks have been moved on (s32 / s33, the ABI's stack and frame pointers, are never touched)
#define LT_R_TRI8 "s[36:43]"
#define LT_R_TRI4 "s[44:47]"
#define LT_R_AX "s36"
#define LT_R_AY "s37"
#define LT_R_AZ "s38"
#define LT_R_E1X "s39"
#define LT_R_E1Y "s40"
#define LT_R_E1Z "s41"
#define LT_R_E2X "s42"
#define LT_R_E2Y "s43"
#define LT_R_E2Z "s44"
#define LT_R_TMPM "s[52:53]"   /* scratch mask / popped mask */
#define LT_R_TMPLO "s52"
#define LT_R_TMPHI "s53"
#define LT_R_PRIM "s54"        /* scratch; in the triangle test: the primitive offset */
#define LT_R_LEAF "s55"        /* pending leaf reference / scratch */
#define LT_R_EXEC "s[56:57]"   /* EXEC on entry */
#define LT_R_HML "s[58:59]"
#define LT_R_HMLLO "s58"
#define LT_R_HMLHI "s59"
#define LT_R_HMR "s[60:61]"
#define LT_R_HMRLO "s60"
#define LT_R_HMRHI "s61"
#define LT_R_LEAFM "s[62:63]"  /* lanes of the pending leaf */
#define LT_ASM_CLOBBERS                                                                                                         \
  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", \
  "s60", "s61", "s62", "s63", "vcc", "scc", "memory"

// slab test of one child: near / far plane registers chosen by the wave's direction-sign octant; result in the SGPR pair OUT
#define LT_ASM_BOX(NX, NY, NZ, FX, FY, FZ, OUT)      \
  "v_sub_f32_e32 %[t0], " NX ", %[ox]\n"            \
  "v_sub_f32_e32 %[t1], " NY ", %[oy]\n"            \
  "v_sub_f32_e32 %[t2], " NZ ", %[oz]\n"            \
  "v_mul_f32_e32 %[t0], %[t0], %[ix]\n"             \
  "v_mul_f32_e32 %[t1], %[t1], %[iy]\n"             \
  "v_mul_f32_e32 %[t2], %[t2], %[iz]\n"             \
  "v_max3_f32 %[t0], %[t0], %[t1], %[t2]\n"         \
  "v_sub_f32_e32 %[t1], " FX ", %[ox]\n"            \
  "v_sub_f32_e32 %[t2], " FY ", %[oy]\n"            \
  "v_sub_f32_e32 %[t3], " FZ ", %[oz]\n"            \
  "v_mul_f32_e32 %[t1], %[t1], %[ix]\n"             \
  "v_mul_f32_e32 %[t2], %[t2], %[iy]\n"             \
  "v_mul_f32_e32 %[t3], %[t3], %[iz]\n"             \
  "v_min3_f32 %[t1], %[t1], %[t2], %[t3]\n"         \
  "v_max_f32_e32 %[t0], 1, %[t0]\n"                 \
  "v_cmp_ge_f32_e64 " OUT ", %[t1], %[t0]\n"

// the two children's tests for the octant NEG (bit a set = direction component a negative: the near plane is the box's max)
#define LT_ASM_BOXES_0 LT_ASM_BOX(LT_R_LMINX, LT_R_LMINY, LT_R_LMINZ, LT_R_LMAXX, LT_R_LMAXY, LT_R_LMAXZ, LT_R_HML) LT_ASM_BOX(LT_R_RMINX, LT_R_RMINY, LT_R_RMINZ, LT_R_RMAXX, LT_R_RMAXY, LT_R_RMAXZ, LT_R_HMR)
#define LT_ASM_BOXES_1 LT_ASM_BOX(LT_R_LMAXX, LT_R_LMINY, LT_R_LMINZ, LT_R_LMINX, LT_R_LMAXY, LT_R_LMAXZ, LT_R_HML) LT_ASM_BOX(LT_R_RMAXX, LT_R_RMINY, LT_R_RMINZ, LT_R_RMINX, LT_R_RMAXY, LT_R_RMAXZ, LT_R_HMR)
#define LT_ASM_BOXES_2 LT_ASM_BOX(LT_R_LMINX, LT_R_LMAXY, LT_R_LMINZ, LT_R_LMAXX, LT_R_LMINY, LT_R_LMAXZ, LT_R_HML) LT_ASM_BOX(LT_R_RMINX, LT_R_RMAXY, LT_R_RMINZ, LT_R_RMAXX, LT_R_RMINY, LT_R_RMAXZ, LT_R_HMR)
#define LT_ASM_BOXES_3 LT_ASM_BOX(LT_R_LMAXX, LT_R_LMAXY, LT_R_LMINZ, LT_R_LMINX, LT_R_LMINY, LT_R_LMAXZ, LT_R_HML) LT_ASM_BOX(LT_R_RMAXX, LT_R_RMAXY, LT_R_RMINZ, LT_R_RMINX, LT_R_RMINY, LT_R_RMAXZ, LT_R_HMR)
#define LT_ASM_BOXES_4 LT_ASM_BOX(LT_R_LMINX, LT_R_LMINY, LT_R_LMAXZ, LT_R_LMAXX, LT_R_LMAXY, LT_R_LMINZ, LT_R_HML) LT_ASM_BOX(LT_R_RMINX, LT_R_RMINY, LT_R_RMAXZ, LT_R_RMAXX, LT_R_RMAXY, LT_R_RMINZ, LT_R_HMR)
#define LT_ASM_BOXES_5 LT_ASM_BOX(LT_R_LMAXX, LT_R_LMINY, LT_R_LMAXZ, LT_R_LMINX, LT_R_LMAXY, LT_R_LMINZ, LT_R_HML) LT_ASM_BOX(LT_R_RMAXX, LT_R_RMINY, LT_R_RMAXZ, LT_R_RMINX, LT_R_RMAXY, LT_R_RMINZ, LT_R_HMR)
#define LT_ASM_BOXES_6 LT_ASM_BOX(LT_R_LMINX, LT_R_LMAXY, LT_R_LMAXZ, LT_R_LMAXX, LT_R_LMINY, LT_R_LMINZ, LT_R_HML) LT_ASM_BOX(LT_R_RMINX, LT_R_RMAXY, LT_R_RMAXZ, LT_R_RMAXX, LT_R_RMINY, LT_R_RMINZ, LT_R_HMR)
#define LT_ASM_BOXES_7 LT_ASM_BOX(LT_R_LMAXX, LT_R_LMAXY, LT_R_LMAXZ, LT_R_LMINX, LT_R_LMINY, LT_R_LMINZ, LT_R_HML) LT_ASM_BOX(LT_R_RMAXX, LT_R_RMAXY, LT_R_RMAXZ, LT_R_RMINX, LT_R_RMINY, LT_R_RMINZ, LT_R_HMR)

// push (reference REF, lane mask LO:HI) on the wave-uniform stack
#define LT_ASM_PUSH(REF, LO, HI)                    \
  "v_lshl_add_u32 %[t0], %[sp], 4, %[lds]\n"        \
  "v_mov_b32_e32 %[t1], " REF "\n"                  \
  "v_mov_b32_e32 %[t2], " LO "\n"                   \
  "v_mov_b32_e32 %[t3], " HI "\n"                   \
  "ds_write_b32 %[t0], %[t1]\n"                     \
  "ds_write_b32 %[t0], %[t2] offset:4\n"            \
  "ds_write_b32 %[t0], %[t3] offset:8\n"            \
  "s_add_u32 %[sp], %[sp], 1\n"

// Load the node record of `cur` and wait for it.
#define LT_ASM_LOAD_NODE                                    \
  "s_and_b32 " LT_R_PRIM ", %[cur], 0x1fffffff\n"          \
  "s_lshl_b32 " LT_R_PRIM ", " LT_R_PRIM ", 6\n"           \
  "s_load_dwordx16 " LT_R_REC ", %[pairs], " LT_R_PRIM "\n" \
  "s_waitcnt lgkmcnt(0)\n"

// pop the top entry: reference -> LT_R_LEAF, mask -> LT_R_TMPM (EXEC = the entry EXEC: every lane reads the same address)
#define LT_ASM_POP_TOP                                      \
  "s_mov_b64 exec, " LT_R_EXEC "\n"                         \
  "s_sub_u32 %[sp], %[sp], 1\n"                             \
  "v_lshl_add_u32 %[t0], %[sp], 4, %[lds]\n"                \
  "ds_read_b32 %[t1], %[t0]\n"                              \
  "ds_read_b32 %[t2], %[t0] offset:4\n"                     \
  "ds_read_b32 %[t3], %[t0] offset:8\n"                     \
  "s_waitcnt lgkmcnt(0)\n"                                  \
  "v_readfirstlane_b32 " LT_R_LEAF ", %[t1]\n"              \
  "v_readfirstlane_b32 " LT_R_TMPLO ", %[t2]\n"             \
  "v_readfirstlane_b32 " LT_R_TMPHI ", %[t3]\n"

// Start of a triangle test: LT_R_LEAF = the leaf's reference, LT_R_LEAFM = its lanes, %[cur] = the interior node to go to
// afterwards or -1 = "pop".  Ends with EXEC = the leaf's lanes; LT_ASM_TRI_PART1 waits for the loads.
// (Issuing the NEXT node's record load beside the triangle's -- `cur`, or the peeked top of the stack, into registers of its
// own -- so that the node-to-node load latency hides behind the test was built and measured: 28.45 against 28.55 ms per
// 16-sample 4K frame, i.e. nothing.  Eight waves per SIMD already cover that latency; the walk is bound by instruction issue.)
#define LT_ASM_LEAF_PROLOGUE                                                                                                    \
  ".Lleaf%=:\n"                                                                                                                 \
  "s_and_b32 " LT_R_PRIM ", " LT_R_LEAF ", 0x7fffffff\n"                                                                         \
  "s_mul_i32 " LT_R_TMPLO ", " LT_R_PRIM ", 48\n"                                                                                \
  "s_load_dwordx8 " LT_R_TRI8 ", %[tris], " LT_R_TMPLO "\n"                                                                      \
  "s_load_dwordx4 " LT_R_TRI4 ", %[tris], " LT_R_TMPLO " offset:0x20\n"                                                          \
  "s_mov_b64 exec, " LT_R_LEAFM "\n"

// det, 1 / det, u (EXEC narrowed by the det and u tests); leaves: t0 = invDet, t1 = u, t7 t8 t9 = tvec
#define LT_ASM_TRI_PART1                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "v_mul_f32_e64 %[t4], %[dz], -" LT_R_E2Y "\n"     /* pvec = cross(d, e2) */                                                   \
  "v_fmac_f32_e32 %[t4], " LT_R_E2Z ", %[dy]\n"                                                                                 \
  "v_mul_f32_e64 %[t5], %[dx], -" LT_R_E2Z "\n"                                                                                 \
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

// after a triangle test: on to node `cur`, or pop
#define LT_ASM_LEAF_EPILOGUE                                \
  ".LleafEnd%=:\n"                                          \
  "s_cmp_eq_u32 %[cur], -1\n"                               \
  "s_cbranch_scc0 .Ltop%=\n"                                \
  "s_branch .Lpop%=\n"

// ---------------------------------------------------------------------------------------------------------------- any-hit
// Order-free walk for shadow rays (lt_device.hpp, traverse_packet_pairs_anyhit): a leaf is tested as soon as it is met (before
// descending into a sibling subtree), the other hit child is entered next or pushed.  `open` = lanes still looking for an
// occluder; nodes none of them reaches are skipped; the walk ends when `open` is empty or the stack is.
#define LT_ASM_ANYHIT_WALK(BOXES)                                                                                               \
  "s_mov_b64 " LT_R_EXEC ", exec\n"                                                                                             \
  ".Ltop%=:\n"                                                                                                                  \
  LT_ASM_LOAD_NODE                                                                                                              \
  "s_and_b64 " LT_R_TMPM ", %[mask], %[open]\n"     /* lanes of this node that still look: SCC = any */                          \
  "s_cbranch_scc0 .Lpop%=\n"                                                                                                    \
  "s_mov_b64 exec, " LT_R_TMPM "\n"                                                                                             \
  BOXES                                                                                                                         \
  "s_or_b64 " LT_R_TMPM ", " LT_R_HML ", " LT_R_HMR "\n"                                                                         \
  "s_cbranch_scc0 .Lpop%=\n"                        /* both children missed */                                                  \
  "s_cmp_lg_u64 " LT_R_HML ", 0\n"                                                                                              \
  "s_cbranch_scc0 .LonlyR%=\n"                                                                                                  \
  "s_cmp_lt_i32 " LT_R_REFL ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LleafL%=\n"                                                                                                  \
  /* left child: interior, hit */                                                                                               \
  "s_cmp_lg_u64 " LT_R_HMR ", 0\n"                                                                                              \
  "s_cbranch_scc0 .LdescL%=\n"                                                                                                  \
  "s_cmp_lt_i32 " LT_R_REFR ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LevR_thenL%=\n"                                                                                              \
  LT_ASM_PUSH(LT_R_REFR, LT_R_HMRLO, LT_R_HMRHI)    /* right child: interior, hit too: it waits */                              \
  ".LdescL%=:\n"                                                                                                                \
  "s_mov_b32 %[cur], " LT_R_REFL "\n"                                                                                           \
  "s_mov_b64 %[mask], " LT_R_HML "\n"                                                                                           \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LleafL%=:\n"                                    /* left child: a leaf some lane hit -> test it */                           \
  "s_mov_b32 " LT_R_LEAF ", " LT_R_REFL "\n"                                                                                    \
  "s_mov_b64 " LT_R_LEAFM ", " LT_R_HML "\n"                                                                                    \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  "s_cmp_lg_u64 " LT_R_HMR ", 0\n"                                                                                              \
  "s_cbranch_scc0 .Lleaf%=\n"                                                                                                   \
  "s_cmp_lt_i32 " LT_R_REFR ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LpushR_leaf%=\n"                                                                                             \
  "s_mov_b32 %[cur], " LT_R_REFR "\n"               /* then the right child (interior) */                                       \
  "s_mov_b64 %[mask], " LT_R_HMR "\n"                                                                                           \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LpushR_leaf%=:\n"                                                                                                           \
  LT_ASM_PUSH(LT_R_REFR, LT_R_HMRLO, LT_R_HMRHI)    /* right child: a second leaf, comes back through the stack */              \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LevR_thenL%=:\n"                                /* right leaf first (order is free), then into the left child */            \
  "s_mov_b32 " LT_R_LEAF ", " LT_R_REFR "\n"                                                                                    \
  "s_mov_b64 " LT_R_LEAFM ", " LT_R_HMR "\n"                                                                                    \
  "s_mov_b32 %[cur], " LT_R_REFL "\n"                                                                                           \
  "s_mov_b64 %[mask], " LT_R_HML "\n"                                                                                           \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LonlyR%=:\n"                                                                                                                \
  "s_cmp_lt_i32 " LT_R_REFR ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LevR_pop%=\n"                                                                                                \
  "s_mov_b32 %[cur], " LT_R_REFR "\n"                                                                                           \
  "s_mov_b64 %[mask], " LT_R_HMR "\n"                                                                                           \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LevR_pop%=:\n"                                                                                                              \
  "s_mov_b32 " LT_R_LEAF ", " LT_R_REFR "\n"                                                                                    \
  "s_mov_b64 " LT_R_LEAFM ", " LT_R_HMR "\n"                                                                                    \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".Lpop%=:\n"                                                                                                                  \
  "s_cmp_eq_u32 %[sp], 0\n"                                                                                                     \
  "s_cbranch_scc1 .Ldone%=\n"                                                                                                   \
  LT_ASM_POP_TOP                                                                                                                \
  "s_cmp_lt_i32 " LT_R_LEAF ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LpoppedLeaf%=\n"                                                                                             \
  "s_mov_b32 %[cur], " LT_R_LEAF "\n"                                                                                           \
  "s_mov_b64 %[mask], " LT_R_TMPM "\n"                                                                                          \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LpoppedLeaf%=:\n"                                                                                                           \
  "s_and_b64 " LT_R_LEAFM ", " LT_R_TMPM ", %[open]\n" /* lanes of it that still look */                                        \
  "s_cbranch_scc0 .Lpop%=\n"                                                                                                    \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  LT_ASM_LEAF_PROLOGUE                                                                                                          \
  "v_cmpx_ne_u32_e64 " LT_R_HML ", " LT_R_PRIM ", %[ign]\n" /* not the primitive the ray starts on (acc.cl:188) */              \
  LT_ASM_TRI_PART1                                                                                                              \
  LT_ASM_TRI_PART2                                                                                                              \
  "v_cmpx_lt_f32_e64 " LT_R_HML ", %[t3], %[tmax]\n" /* t < payload.t: accepted */                                              \
  "s_andn2_b64 %[open], %[open], exec\n"            /* those lanes are done; SCC = anyone still looking */                      \
  "s_cbranch_scc0 .Ldone%=\n"                                                                                                   \
  LT_ASM_LEAF_EPILOGUE                                                                                                          \
  ".Ldone%=:\n"                                                                                                                 \
  "s_mov_b64 exec, " LT_R_EXEC "\n"

// The whole any-hit walk below the root: on return `open` holds the lanes that found no occluder.  NEG = the direction-sign
// octant all rays of the wave share; eps = the program's triangle epsilon as the float the reference's double compare amounts
// to (intersect_triangle_data).
template <int NEG>
__device__ __forceinline__ lt_u64 packet_anyhit_walk(const void* pairs, const void* tris, float ox, float oy, float oz, float ix, float iy,
                                                     float iz, float dx, float dy, float dz, float dw, float tmax, int ign, float eps,
                                                     uint32_t fast, uint32_t lds, lt_u64 mask, lt_u64 open) {
  uint32_t cur = 0u, sp = 0u;
  float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
#define LT_ANYHIT_INSTANCE(BOXES)                                                                                                        \
  asm volatile(LT_ASM_ANYHIT_WALK(BOXES)                                                                                                 \
               : [cur] "+s"(cur), [mask] "+s"(mask), [sp] "+s"(sp), [open] "+s"(open), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),    \
                 [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9),         \
                 [t10] "=&v"(t10)                                                                                                        \
               : [pairs] "s"(pairs), [tris] "s"(tris), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), \
                 [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [dw] "v"(dw), [tmax] "v"(tmax), [ign] "v"(ign), [eps] "s"(eps), [fast] "s"(fast), [lds] "v"(lds) \
               : LT_ASM_CLOBBERS)
  if constexpr (NEG == 0) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_0);
  else if constexpr (NEG == 1) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_1);
  else if constexpr (NEG == 2) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_2);
  else if constexpr (NEG == 3) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_3);
  else if constexpr (NEG == 4) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_4);
  else if constexpr (NEG == 5) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_5);
  else if constexpr (NEG == 6) LT_ANYHIT_INSTANCE(LT_ASM_BOXES_6);
  else LT_ANYHIT_INSTANCE(LT_ASM_BOXES_7);
#undef LT_ANYHIT_INSTANCE
  return open;
}

// ------------------------------------------------------------------------------------------------------------ closest hit
// The camera-ray walk (lt_device.hpp, traverse_packet_pairs): per lane the reference's order -- the near child's subtree (or
// leaf) completely before the far child's, near = the child on the side the rays come from along the node's split axis
// (acc.cl:150-160: dirIsNeg[node->axis]) -- so a far child that is hit while the near one is entered waits on the stack, and a
// far LEAF behind a near leaf goes through the stack too (it is the top entry, popped at once).  The payload (t, u, v,
// primitive, hitType: RayPayload, acc.cl:55-61) lives in five VGPRs and is overwritten under the EXEC the triangle test ends
// with: the lanes whose `t < payload.t` held.
//   visit(N = near, F = far): TAG makes the labels of the two instances distinct
#define LT_ASM_VISIT(TAG, HMN, REFN, HMF, REFF, FLO, FHI)                                                                        \
  "s_cmp_lg_u64 " HMN ", 0\n"                                                                                                   \
  "s_cbranch_scc0 .LnMiss" TAG "%=\n"                                                                                           \
  "s_cmp_lt_i32 " REFN ", 0\n"                                                                                                  \
  "s_cbranch_scc1 .LnLeaf" TAG "%=\n"                                                                                           \
  "s_cmp_lg_u64 " HMF ", 0\n"                       /* near: interior, hit */                                                   \
  "s_cbranch_scc0 .Ldesc" TAG "%=\n"                                                                                            \
  LT_ASM_PUSH(REFF, FLO, FHI)                       /* far child (leaf or interior) waits */                                    \
  ".Ldesc" TAG "%=:\n"                                                                                                          \
  "s_mov_b32 %[cur], " REFN "\n"                                                                                                \
  "s_mov_b64 %[mask], " HMN "\n"                                                                                                \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LnLeaf" TAG "%=:\n"                             /* near: a leaf some lane hit -> test it now */                             \
  "s_mov_b32 " LT_R_LEAF ", " REFN "\n"                                                                                         \
  "s_mov_b64 " LT_R_LEAFM ", " HMN "\n"                                                                                         \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  "s_cmp_lg_u64 " HMF ", 0\n"                                                                                                   \
  "s_cbranch_scc0 .Lleaf%=\n"                                                                                                   \
  "s_cmp_lt_i32 " REFF ", 0\n"                                                                                                  \
  "s_cbranch_scc1 .LfLeaf" TAG "%=\n"                                                                                           \
  "s_mov_b32 %[cur], " REFF "\n"                    /* then the far child (interior) */                                         \
  "s_mov_b64 %[mask], " HMF "\n"                                                                                                \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LfLeaf" TAG "%=:\n"                                                                                                         \
  LT_ASM_PUSH(REFF, FLO, FHI)                       /* far leaf: next, through the stack */                                     \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LnMiss" TAG "%=:\n"                             /* only the far child was hit */                                            \
  "s_cmp_lt_i32 " REFF ", 0\n"                                                                                                  \
  "s_cbranch_scc1 .LfOnlyLeaf" TAG "%=\n"                                                                                       \
  "s_mov_b32 %[cur], " REFF "\n"                                                                                                \
  "s_mov_b64 %[mask], " HMF "\n"                                                                                                \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LfOnlyLeaf" TAG "%=:\n"                                                                                                     \
  "s_mov_b32 " LT_R_LEAF ", " REFF "\n"                                                                                         \
  "s_mov_b64 " LT_R_LEAFM ", " HMF "\n"                                                                                         \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  "s_branch .Lleaf%=\n"

#define LT_ASM_CLOSEST_WALK(BOXES, NEGBITS, ROFF)                                                                                    \
  "s_mov_b64 " LT_R_EXEC ", exec\n"                                                                                             \
  ".Ltop%=:\n"                                                                                                                  \
  LT_ASM_LOAD_NODE                                                                                                              \
  "s_mov_b64 exec, %[mask]\n"                                                                                                   \
  BOXES                                                                                                                         \
  "s_or_b64 " LT_R_TMPM ", " LT_R_HML ", " LT_R_HMR "\n"                                                                         \
  "s_cbranch_scc0 .Lpop%=\n"                        /* both children missed */                                                  \
  "s_lshr_b32 " LT_R_PRIM ", %[cur], 29\n"          /* the node's split axis */                                                 \
  "s_bitcmp1_b32 " NEGBITS ", " LT_R_PRIM "\n"      /* dirIsNeg[axis]: the right child is the near one */                       \
  "s_cbranch_scc1 .LnearR%=\n"                                                                                                  \
  LT_ASM_VISIT("a", LT_R_HML, LT_R_REFL, LT_R_HMR, LT_R_REFR, LT_R_HMRLO, LT_R_HMRHI)                                           \
  ".LnearR%=:\n"                                                                                                                \
  LT_ASM_VISIT("b", LT_R_HMR, LT_R_REFR, LT_R_HML, LT_R_REFL, LT_R_HMLLO, LT_R_HMLHI)                                           \
  ".Lpop%=:\n"                                                                                                                  \
  "s_cmp_eq_u32 %[sp], 0\n"                                                                                                     \
  "s_cbranch_scc1 .Ldone%=\n"                                                                                                   \
  LT_ASM_POP_TOP                                                                                                                \
  "s_cmp_lt_i32 " LT_R_LEAF ", 0\n"                                                                                             \
  "s_cbranch_scc1 .LpoppedLeaf%=\n"                                                                                             \
  "s_mov_b32 %[cur], " LT_R_LEAF "\n"                                                                                           \
  "s_mov_b64 %[mask], " LT_R_TMPM "\n"                                                                                          \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LpoppedLeaf%=:\n"                                                                                                           \
  "s_mov_b64 " LT_R_LEAFM ", " LT_R_TMPM "\n"                                                                                   \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  LT_ASM_LEAF_PROLOGUE                                                                                                          \
  LT_ASM_TRI_PART1                                                                                                              \
  LT_ASM_TRI_PART2                                                                                                              \
  "v_cmp_eq_f32_e64 " LT_R_HMR ", %[t3], %[pt]\n"   /* the same t again, bit for bit? (rare) */                                 \
  "v_cmp_lt_f32_e64 " LT_R_HML ", %[t3], %[pt]\n"   /* t < payload.t (no t > 0 test in the reference) */                        \
  "s_cmp_lg_u64 " LT_R_HMR ", 0\n"                                                                                              \
  "s_cbranch_scc0 .Ltake%=\n"                                                                                                   \
  /* Two triangles, one t: the reference keeps the one its own depth-first order meets first.  This walk follows the backend's   \
     tree, so the order comes from the table (SceneDev::rank8: 8 ranks per primitive, one per direction-sign octant). */         \
  "s_cmp_lg_u64 %[ranks], 0\n"                      /* no table: this IS the reference's order */                               \
  "s_cbranch_scc0 .Ltake%=\n"                                                                                                   \
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
  "v_mov_b32_e32 %[phit], 1\n"                                                                                                  \
  LT_ASM_LEAF_EPILOGUE                                                                                                          \
  ".Ldone%=:\n"                                                                                                                 \
  "s_mov_b64 exec, " LT_R_EXEC "\n"

// The whole closest-hit walk below the root.  `cur` = the root's reference (index 0 | its split axis << 29), `mask` = the lanes
// that hit the root's box.
template <int NEG>
__device__ __forceinline__ void packet_closest_walk(const void* pairs, const void* tris, const void* ranks, float ox, float oy, float oz, float ix, float iy,
                                                    float iz, float dx, float dy, float dz, float dw, float eps, uint32_t fast, uint32_t lds,
                                                    uint32_t cur, lt_u64 mask, float& pt, float& pu, float& pv, int& pprim, int& phit) {
  uint32_t sp = 0u;
  float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
#define LT_CLOSEST_INSTANCE(BOXES, NEGBITS, ROFF)                                                                                            \
  asm volatile(LT_ASM_CLOSEST_WALK(BOXES, NEGBITS, ROFF)                                                                                     \
               : [cur] "+s"(cur), [mask] "+s"(mask), [sp] "+s"(sp), [pt] "+v"(pt), [pu] "+v"(pu), [pv] "+v"(pv), [pprim] "+v"(pprim),    \
                 [phit] "+v"(phit), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),     \
                 [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9), [t10] "=&v"(t10)                                        \
               : [pairs] "s"(pairs), [tris] "s"(tris), [ranks] "s"(ranks), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), \
                 [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [dw] "v"(dw), [eps] "s"(eps), [fast] "s"(fast), [lds] "v"(lds)                  \
               : LT_ASM_CLOBBERS)
  if constexpr (NEG == 0) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_0, "0", "0");
  else if constexpr (NEG == 1) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_1, "1", "4");
  else if constexpr (NEG == 2) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_2, "2", "8");
  else if constexpr (NEG == 3) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_3, "3", "12");
  else if constexpr (NEG == 4) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_4, "4", "16");
  else if constexpr (NEG == 5) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_5, "5", "20");
  else if constexpr (NEG == 6) LT_CLOSEST_INSTANCE(LT_ASM_BOXES_6, "6", "24");
  else LT_CLOSEST_INSTANCE(LT_ASM_BOXES_7, "7", "28");
#undef LT_CLOSEST_INSTANCE
}

}  // namespace lt
