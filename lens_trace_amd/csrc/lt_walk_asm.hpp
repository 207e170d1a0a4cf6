// lt_walk_asm.hpp -- the interior-node loop of the packet walks, hand-written for gfx950.
//
// The packet walks (lt_device.hpp: traverse_packet_pairs, traverse_packet_pairs_anyhit) are bound by SCALAR instruction issue:
// rocprofv3 on the bench workload shows 0.75 scalar instructions per CU-cycle against 0.62 of the VALU peak
// (profiles/r2/issue_profile.json).  hipcc turns their wave-uniform, multi-exit loops into a flag-driven state machine
// (structurised control flow: s_mov -1 / s_andn2 exec / s_cbranch_vccnz chains, a loop-exit selector register, phi copies of
// every loop-carried mask at the latch): ~25-30 scalar instructions per visited pair of nodes, of which the algorithm needs
// about a dozen.  This file is that dozen:
//
//   * one `s_load_dwordx16` per interior node brings the 64-byte child-pair record (lt_pair_kernel) into s[36:51];
//   * the wave's lane mask of the node becomes EXEC for the two slab tests, so each test ends in ONE `v_cmp_ge_f32` whose
//     SGPR-pair result already is "lanes of this node that hit the child" -- no s_and with the node's mask, and the
//     reference's two conditions `tEnter <= tExit && tExit > 0` (acc.cl:113-130, in box_mask<NEG>'s octant form) fold into
//     `tExit >= max(tEnter, 0x00000001)`: the smallest positive float (denormals are kept: .amdhsa_float_denorm_mode_32 3)
//     stands for "> 0", exact for every non-NaN input, and a packet is only formed from rays that cannot produce a NaN
//     (traverse_camera / traverse: all origins and inverse directions finite);
//   * branches test SCC straight from the mask arithmetic; the wave-uniform stack (child reference + 64-bit lane mask per
//     entry, one row of the wave's LDS stack each) is written / read with three ds_*_b32 of identical data per lane.
//
// The loop runs until it reaches a LEAF some lane has to test, then leaves the asm block with that leaf's reference and lane
// mask ("event"); the triangle test stays compiler-generated C++ (lt_device.hpp), and the next call resumes the walk.  All
// state lives in the operands: `cur` (next interior node, or 0xffffffff = "pop the stack first"), `mask`, `sp` (stack rows in
// use).  EXEC is saved on entry and restored on every exit.  No instruction here needs a manually inserted wait state on
// gfx950 (plain VALU -> SGPR -> SALU, SALU -> SMEM address, DS + s_waitcnt lgkmcnt(0) before v_readfirstlane; no DPP, no
// v_readlane with a VALU-written select, no v_div_fmas, no VMEM).
#pragma once

namespace lt {

typedef unsigned long long lt_u64;

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

// push (reference REF, lane mask LO:HI) on the wave-uniform stack
#define LT_ASM_PUSH(REF, LO, HI)                    \
  "v_lshl_add_u32 %[t0], %[sp], 8, %[lds]\n"        \
  "v_mov_b32_e32 %[t1], " REF "\n"                  \
  "v_mov_b32_e32 %[t2], " LO "\n"                   \
  "v_mov_b32_e32 %[t3], " HI "\n"                   \
  "ds_write_b32 %[t0], %[t1]\n"                     \
  "ds_write_b32 %[t0], %[t2] offset:4\n"            \
  "ds_write_b32 %[t0], %[t3] offset:8\n"            \
  "s_add_u32 %[sp], %[sp], 1\n"

// Record layout in s[36:51] (lt_pair_kernel): left child  min = s36 s37 s38, max = s39 s40 s41, reference = s42;
//                                              right child min = s44 s45 s46, max = s47 s48 s49, reference = s50.
// A reference with bit 31 set is a leaf (0x80000000 | primitive offset), else an interior node (index | axis << 29).
// Fixed registers: s[52:53] scratch mask, s54 scratch, s[56:57] saved EXEC, s[58:59] / s[60:61] hit masks of the left / right
// child, s[62:63] popped mask.
#define LT_ASM_CLOBBERS                                                                                                         \
  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", \
  "s54", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "scc", "memory"

// ---------------------------------------------------------------------------------------------------------------- any-hit
// Order-free walk for shadow rays (lt_device.hpp, traverse_packet_pairs_anyhit): a leaf is tested as soon as it is met (before
// descending into a sibling subtree), the other hit child is entered next or pushed.  `open` = lanes still looking for an
// occluder; nodes none of them reaches are skipped; the walk ends when `open` is empty or the stack is.
//
// The triangle test (acc.cl:72-111 on the re-tiled 48-byte triangle, intersect_triangle_anyhit's arithmetic instruction for
// instruction: cross = fma(a, b, -(c * d)), dot = fma chain + the `w` terms, IEEE 1 / det by the div_scale / rcp / fma /
// div_fmas / div_fixup sequence hipcc emits, each reject written as the reference's negated compare) runs with EXEC = the
// lanes that reached the leaf, and every test NARROWS EXEC (v_cmpx): what is left of EXEC at the end is the mask of lanes that
// found their occluder.  v_div_fmas reads the VCC of the second v_div_scale four VALU instructions later (the required wait
// states); v_rcp's result is first read three instructions later.
#define LT_ASM_ANYHIT_WALK(LNX, LNY, LNZ, LFX, LFY, LFZ, RNX, RNY, RNZ, RFX, RFY, RFZ, EPS)                                       \
  "s_mov_b64 s[56:57], exec\n"                                                                                                  \
  ".Ltop%=:\n"                                                                                                                  \
  "s_and_b64 s[52:53], %[mask], %[open]\n"          /* lanes of this node that still look: SCC = any */                          \
  "s_cbranch_scc0 .Lpop%=\n"                                                                                                    \
  "s_and_b32 s54, %[cur], 0x1fffffff\n"                                                                                         \
  "s_lshl_b32 s54, s54, 6\n"                                                                                                    \
  "s_load_dwordx16 s[36:51], %[pairs], s54\n"                                                                                   \
  "s_mov_b64 exec, s[52:53]\n"                                                                                                  \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  LT_ASM_BOX(LNX, LNY, LNZ, LFX, LFY, LFZ, "s[58:59]")                                                                          \
  LT_ASM_BOX(RNX, RNY, RNZ, RFX, RFY, RFZ, "s[60:61]")                                                                          \
  "s_or_b64 s[52:53], s[58:59], s[60:61]\n"                                                                                     \
  "s_cbranch_scc0 .Lpop%=\n"                        /* both children missed */                                                  \
  "s_cmp_lg_u64 s[58:59], 0\n"                                                                                                  \
  "s_cbranch_scc0 .LonlyR%=\n"                                                                                                  \
  "s_cmp_lt_i32 s42, 0\n"                                                                                                       \
  "s_cbranch_scc1 .LleafL%=\n"                                                                                                  \
  /* left child: interior, hit */                                                                                               \
  "s_cmp_lg_u64 s[60:61], 0\n"                                                                                                  \
  "s_cbranch_scc0 .LdescL%=\n"                                                                                                  \
  "s_cmp_lt_i32 s50, 0\n"                                                                                                       \
  "s_cbranch_scc1 .LevR_thenL%=\n"                                                                                              \
  LT_ASM_PUSH("s50", "s60", "s61")                  /* right child: interior, hit too: it waits */                              \
  ".LdescL%=:\n"                                                                                                                \
  "s_mov_b32 %[cur], s42\n"                                                                                                     \
  "s_mov_b64 %[mask], s[58:59]\n"                                                                                               \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LleafL%=:\n"                                    /* left child: a leaf some lane hit -> test it */                           \
  "s_mov_b32 s55, s42\n"                                                                                                        \
  "s_mov_b64 s[62:63], s[58:59]\n"                                                                                              \
  "s_cmp_lg_u64 s[60:61], 0\n"                                                                                                  \
  "s_cbranch_scc0 .LleafThenPop%=\n"                                                                                            \
  "s_cmp_lt_i32 s50, 0\n"                                                                                                       \
  "s_cbranch_scc1 .LpushR_leafThenPop%=\n"                                                                                      \
  "s_mov_b32 %[cur], s50\n"                         /* then the right child (interior) */                                       \
  "s_mov_b64 %[mask], s[60:61]\n"                                                                                               \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LpushR_leafThenPop%=:\n"                                                                                                    \
  LT_ASM_PUSH("s50", "s60", "s61")                  /* right child: a second leaf, comes back through the stack */              \
  ".LleafThenPop%=:\n"                                                                                                          \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LevR_thenL%=:\n"                                /* right leaf first (order is free), then into the left child */            \
  "s_mov_b32 s55, s50\n"                                                                                                        \
  "s_mov_b64 s[62:63], s[60:61]\n"                                                                                              \
  "s_mov_b32 %[cur], s42\n"                                                                                                     \
  "s_mov_b64 %[mask], s[58:59]\n"                                                                                               \
  "s_branch .Lleaf%=\n"                                                                                                         \
  ".LonlyR%=:\n"                                                                                                                \
  "s_cmp_lt_i32 s50, 0\n"                                                                                                       \
  "s_cbranch_scc1 .LevR_pop%=\n"                                                                                                \
  "s_mov_b32 %[cur], s50\n"                                                                                                     \
  "s_mov_b64 %[mask], s[60:61]\n"                                                                                               \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LevR_pop%=:\n"                                                                                                              \
  "s_mov_b32 s55, s50\n"                                                                                                        \
  "s_mov_b64 s[62:63], s[60:61]\n"                                                                                              \
  "s_branch .LleafThenPop%=\n"                                                                                                  \
  ".Lpop%=:\n"                                                                                                                  \
  "s_cmp_eq_u32 %[sp], 0\n"                                                                                                     \
  "s_cbranch_scc1 .Ldone%=\n"                                                                                                   \
  "s_mov_b64 exec, s[56:57]\n"                                                                                                  \
  "s_sub_u32 %[sp], %[sp], 1\n"                                                                                                 \
  "v_lshl_add_u32 %[t0], %[sp], 8, %[lds]\n"                                                                                    \
  "ds_read_b32 %[t1], %[t0]\n"                                                                                                  \
  "ds_read_b32 %[t2], %[t0] offset:4\n"                                                                                         \
  "ds_read_b32 %[t3], %[t0] offset:8\n"                                                                                         \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "v_readfirstlane_b32 s55, %[t1]\n"                                                                                            \
  "v_readfirstlane_b32 s62, %[t2]\n"                                                                                            \
  "v_readfirstlane_b32 s63, %[t3]\n"                                                                                            \
  "s_cmp_lt_i32 s55, 0\n"                                                                                                       \
  "s_cbranch_scc1 .LpoppedLeaf%=\n"                                                                                             \
  "s_mov_b32 %[cur], s55\n"                                                                                                     \
  "s_mov_b64 %[mask], s[62:63]\n"                                                                                               \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LpoppedLeaf%=:\n"                                                                                                           \
  "s_and_b64 s[62:63], s[62:63], %[open]\n"         /* lanes of it that still look */                                           \
  "s_cbranch_scc0 .Lpop%=\n"                                                                                                    \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  /* ---- triangle s55 & 0x7fffffff for the lanes s[62:63]; afterwards: pop if cur == -1, else on to node cur ---- */           \
  ".Lleaf%=:\n"                                                                                                                 \
  "s_and_b32 s54, s55, 0x7fffffff\n"                                                                                            \
  "s_mul_i32 s52, s54, 48\n"                                                                                                    \
  "s_load_dwordx8 s[36:43], %[tris], s52\n"                                                                                     \
  "s_load_dwordx4 s[44:47], %[tris], s52 offset:0x20\n"                                                                         \
  "s_mov_b64 exec, s[62:63]\n"                                                                                                  \
  "v_cmpx_ne_u32_e64 s[58:59], s54, %[ign]\n"       /* not the primitive the ray starts on (acc.cl:188) */                      \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  /* A = s36 s37 s38, e1 = s39 s40 s41, e2 = s42 s43 s44;  pvec = cross(d, e2) */                                               \
  "v_mul_f32_e64 %[t4], %[dz], -s43\n"                                                                                          \
  "v_fmac_f32_e32 %[t4], s44, %[dy]\n"                                                                                          \
  "v_mul_f32_e64 %[t5], %[dx], -s44\n"                                                                                          \
  "v_fmac_f32_e32 %[t5], s42, %[dz]\n"                                                                                          \
  "v_mul_f32_e64 %[t6], %[dy], -s42\n"                                                                                          \
  "v_mul_f32_e32 %[t0], s39, %[t4]\n"                                                                                           \
  "v_fmac_f32_e32 %[t6], s43, %[dx]\n"                                                                                          \
  "v_fmac_f32_e32 %[t0], s40, %[t5]\n"                                                                                          \
  "v_fmac_f32_e32 %[t0], s41, %[t6]\n"                                                                                          \
  "v_add_f32_e32 %[t0], 0, %[t0]\n"                 /* det */                                                                   \
  "v_div_scale_f32 %[t1], s[58:59], %[t0], %[t0], 1.0\n"                                                                        \
  "v_rcp_f32_e32 %[t2], %[t1]\n"                                                                                                \
  "v_cmpx_nlt_f32_e64 s[58:59], |%[t0]|, " EPS "\n" /* !(fabs(det) < epsilon) */                                                \
  "v_subrev_f32_e32 %[t7], s36, %[ox]\n"            /* tvec = o - A */                                                          \
  "v_subrev_f32_e32 %[t8], s37, %[oy]\n"                                                                                        \
  "v_fma_f32 %[t3], -%[t1], %[t2], 1.0\n"                                                                                       \
  "v_fmac_f32_e32 %[t2], %[t3], %[t2]\n"                                                                                        \
  "v_div_scale_f32 %[t3], vcc, 1.0, %[t0], 1.0\n"                                                                               \
  "v_mul_f32_e32 %[t9], %[t3], %[t2]\n"                                                                                         \
  "v_fma_f32 %[t10], -%[t1], %[t9], %[t3]\n"                                                                                    \
  "v_fmac_f32_e32 %[t9], %[t10], %[t2]\n"                                                                                       \
  "v_fma_f32 %[t1], -%[t1], %[t9], %[t3]\n"                                                                                     \
  "v_div_fmas_f32 %[t1], %[t1], %[t2], %[t9]\n"                                                                                 \
  "v_div_fixup_f32 %[t0], %[t1], %[t0], 1.0\n"      /* invDet = 1 / det */                                                      \
  "v_subrev_f32_e32 %[t9], s38, %[oz]\n"                                                                                        \
  "v_mul_f32_e32 %[t1], %[t7], %[t4]\n"                                                                                         \
  "v_fmac_f32_e32 %[t1], %[t8], %[t5]\n"                                                                                        \
  "v_fmac_f32_e32 %[t1], %[t9], %[t6]\n"                                                                                        \
  "v_add_f32_e32 %[t1], 0, %[t1]\n"                                                                                             \
  "v_mul_f32_e32 %[t1], %[t1], %[t0]\n"             /* u */                                                                     \
  "v_cmpx_ngt_f32_e64 s[58:59], 0, %[t1]\n"         /* !(u < 0) */                                                              \
  "v_cmpx_nlt_f32_e64 s[58:59], 1.0, %[t1]\n"       /* !(u > 1) */                                                              \
  "s_cbranch_execz .LleafEnd%=\n"                                                                                               \
  /* qvec = cross(tvec, e1) */                                                                                                  \
  "v_mul_f32_e64 %[t4], %[t9], -s40\n"                                                                                          \
  "v_fmac_f32_e32 %[t4], s41, %[t8]\n"                                                                                          \
  "v_mul_f32_e64 %[t5], %[t7], -s41\n"                                                                                          \
  "v_mul_f32_e64 %[t6], %[t8], -s39\n"                                                                                          \
  "v_fmac_f32_e32 %[t5], s39, %[t9]\n"                                                                                          \
  "v_fmac_f32_e32 %[t6], s40, %[t7]\n"                                                                                          \
  "v_mul_f32_e32 %[t2], %[dx], %[t4]\n"                                                                                         \
  "v_fmac_f32_e32 %[t2], %[dy], %[t5]\n"                                                                                        \
  "v_fmac_f32_e32 %[t2], %[dz], %[t6]\n"                                                                                        \
  "v_fmac_f32_e32 %[t2], 0, %[dw]\n"                                                                                            \
  "v_mul_f32_e32 %[t2], %[t2], %[t0]\n"             /* v */                                                                     \
  "v_add_f32_e32 %[t1], %[t1], %[t2]\n"             /* u + v */                                                                 \
  "v_mul_f32_e32 %[t3], s42, %[t4]\n"                                                                                           \
  "v_fmac_f32_e32 %[t3], s43, %[t5]\n"                                                                                          \
  "v_fmac_f32_e32 %[t3], s44, %[t6]\n"                                                                                          \
  "v_cmpx_ngt_f32_e64 s[58:59], 0, %[t2]\n"         /* !(v < 0) */                                                              \
  "v_add_f32_e32 %[t3], 0, %[t3]\n"                                                                                             \
  "v_cmpx_nlt_f32_e64 s[58:59], 1.0, %[t1]\n"       /* !(u + v > 1) */                                                          \
  "v_mul_f32_e32 %[t3], %[t3], %[t0]\n"             /* t */                                                                     \
  "v_cmpx_lt_f32_e64 s[58:59], %[t3], %[tmax]\n"    /* t < payload.t: accepted */                                               \
  "s_andn2_b64 %[open], %[open], exec\n"            /* those lanes are done; SCC = anyone still looking */                      \
  "s_cbranch_scc0 .Ldone%=\n"                                                                                                   \
  ".LleafEnd%=:\n"                                                                                                              \
  "s_mov_b64 exec, s[56:57]\n"                                                                                                  \
  "s_cmp_eq_u32 %[cur], -1\n"                                                                                                   \
  "s_cbranch_scc0 .Ltop%=\n"                                                                                                    \
  "s_branch .Lpop%=\n"                                                                                                          \
  ".Ldone%=:\n"                                                                                                                 \
  "s_mov_b64 exec, s[56:57]\n"

// The whole any-hit walk below the root: on return `open` holds the lanes that found no occluder.  NEG = the direction-sign
// octant all rays of the wave share; eps = the program's triangle epsilon as the float the reference's double compare amounts
// to (intersect_triangle_data).
template <int NEG>
__device__ __forceinline__ lt_u64 packet_anyhit_walk(const void* pairs, const void* tris, float ox, float oy, float oz, float ix, float iy,
                                                     float iz, float dx, float dy, float dz, float dw, float tmax, int ign, float eps,
                                                     uint32_t lds, lt_u64 mask, lt_u64 open) {
  uint32_t cur = 0u, sp = 0u;
  float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
#define LT_ANYHIT_INSTANCE(LNX, LNY, LNZ, LFX, LFY, LFZ, RNX, RNY, RNZ, RFX, RFY, RFZ)                                                    \
  asm volatile(LT_ASM_ANYHIT_WALK(LNX, LNY, LNZ, LFX, LFY, LFZ, RNX, RNY, RNZ, RFX, RFY, RFZ, "%[eps]")                                   \
               : [cur] "+s"(cur), [mask] "+s"(mask), [sp] "+s"(sp), [open] "+s"(open), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),    \
                 [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9),         \
                 [t10] "=&v"(t10)                                                                                                        \
               : [pairs] "s"(pairs), [tris] "s"(tris), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), \
                 [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [dw] "v"(dw), [tmax] "v"(tmax), [ign] "v"(ign), [eps] "s"(eps), [lds] "v"(lds) \
               : LT_ASM_CLOBBERS, "s55", "vcc")
  // near plane of an axis = the box's max when the direction component is negative, else its min
  if constexpr (NEG == 0) LT_ANYHIT_INSTANCE("s36", "s37", "s38", "s39", "s40", "s41", "s44", "s45", "s46", "s47", "s48", "s49");
  else if constexpr (NEG == 1) LT_ANYHIT_INSTANCE("s39", "s37", "s38", "s36", "s40", "s41", "s47", "s45", "s46", "s44", "s48", "s49");
  else if constexpr (NEG == 2) LT_ANYHIT_INSTANCE("s36", "s40", "s38", "s39", "s37", "s41", "s44", "s48", "s46", "s47", "s45", "s49");
  else if constexpr (NEG == 3) LT_ANYHIT_INSTANCE("s39", "s40", "s38", "s36", "s37", "s41", "s47", "s48", "s46", "s44", "s45", "s49");
  else if constexpr (NEG == 4) LT_ANYHIT_INSTANCE("s36", "s37", "s41", "s39", "s40", "s38", "s44", "s45", "s49", "s47", "s48", "s46");
  else if constexpr (NEG == 5) LT_ANYHIT_INSTANCE("s39", "s37", "s41", "s36", "s40", "s38", "s47", "s45", "s49", "s44", "s48", "s46");
  else if constexpr (NEG == 6) LT_ANYHIT_INSTANCE("s36", "s40", "s41", "s39", "s37", "s38", "s44", "s48", "s49", "s47", "s45", "s46");
  else LT_ANYHIT_INSTANCE("s39", "s40", "s41", "s36", "s37", "s38", "s47", "s48", "s49", "s44", "s45", "s46");
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
  "s_mov_b32 s55, " REFN "\n"                                                                                                   \
  "s_mov_b64 s[62:63], " HMN "\n"                                                                                               \
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
  "s_mov_b32 s55, " REFF "\n"                                                                                                   \
  "s_mov_b64 s[62:63], " HMF "\n"                                                                                               \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  "s_branch .Lleaf%=\n"

#define LT_ASM_CLOSEST_WALK(LNX, LNY, LNZ, LFX, LFY, LFZ, RNX, RNY, RNZ, RFX, RFY, RFZ, NEGBITS)                                  \
  "s_mov_b64 s[56:57], exec\n"                                                                                                  \
  ".Ltop%=:\n"                                                                                                                  \
  "s_and_b32 s54, %[cur], 0x1fffffff\n"                                                                                         \
  "s_lshl_b32 s54, s54, 6\n"                                                                                                    \
  "s_load_dwordx16 s[36:51], %[pairs], s54\n"                                                                                   \
  "s_mov_b64 exec, %[mask]\n"                                                                                                   \
  "s_lshr_b32 s54, %[cur], 29\n"                    /* the node's split axis */                                                 \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  LT_ASM_BOX(LNX, LNY, LNZ, LFX, LFY, LFZ, "s[58:59]")                                                                          \
  LT_ASM_BOX(RNX, RNY, RNZ, RFX, RFY, RFZ, "s[60:61]")                                                                          \
  "s_or_b64 s[52:53], s[58:59], s[60:61]\n"                                                                                     \
  "s_cbranch_scc0 .Lpop%=\n"                        /* both children missed */                                                  \
  "s_bitcmp1_b32 " NEGBITS ", s54\n"                /* dirIsNeg[axis]: the right child is the near one */                       \
  "s_cbranch_scc1 .LnearR%=\n"                                                                                                  \
  LT_ASM_VISIT("a", "s[58:59]", "s42", "s[60:61]", "s50", "s60", "s61")                                                         \
  ".LnearR%=:\n"                                                                                                                \
  LT_ASM_VISIT("b", "s[60:61]", "s50", "s[58:59]", "s42", "s58", "s59")                                                         \
  ".Lpop%=:\n"                                                                                                                  \
  "s_cmp_eq_u32 %[sp], 0\n"                                                                                                     \
  "s_cbranch_scc1 .Ldone%=\n"                                                                                                   \
  "s_mov_b64 exec, s[56:57]\n"                                                                                                  \
  "s_sub_u32 %[sp], %[sp], 1\n"                                                                                                 \
  "v_lshl_add_u32 %[t0], %[sp], 8, %[lds]\n"                                                                                    \
  "ds_read_b32 %[t1], %[t0]\n"                                                                                                  \
  "ds_read_b32 %[t2], %[t0] offset:4\n"                                                                                         \
  "ds_read_b32 %[t3], %[t0] offset:8\n"                                                                                         \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "v_readfirstlane_b32 s55, %[t1]\n"                                                                                            \
  "v_readfirstlane_b32 s62, %[t2]\n"                                                                                            \
  "v_readfirstlane_b32 s63, %[t3]\n"                                                                                            \
  "s_cmp_lt_i32 s55, 0\n"                                                                                                       \
  "s_cbranch_scc1 .LpoppedLeaf%=\n"                                                                                             \
  "s_mov_b32 %[cur], s55\n"                                                                                                     \
  "s_mov_b64 %[mask], s[62:63]\n"                                                                                               \
  "s_branch .Ltop%=\n"                                                                                                          \
  ".LpoppedLeaf%=:\n"                                                                                                           \
  "s_mov_b32 %[cur], -1\n"                                                                                                      \
  /* ---- triangle s55 & 0x7fffffff for the lanes s[62:63]; afterwards: pop if cur == -1, else on to node cur ---- */           \
  ".Lleaf%=:\n"                                                                                                                 \
  "s_and_b32 s54, s55, 0x7fffffff\n"                                                                                            \
  "s_mul_i32 s52, s54, 48\n"                                                                                                    \
  "s_load_dwordx8 s[36:43], %[tris], s52\n"                                                                                     \
  "s_load_dwordx4 s[44:47], %[tris], s52 offset:0x20\n"                                                                         \
  "s_mov_b64 exec, s[62:63]\n"                                                                                                  \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                      \
  "v_mul_f32_e64 %[t4], %[dz], -s43\n"              /* pvec = cross(d, e2) */                                                   \
  "v_fmac_f32_e32 %[t4], s44, %[dy]\n"                                                                                          \
  "v_mul_f32_e64 %[t5], %[dx], -s44\n"                                                                                          \
  "v_fmac_f32_e32 %[t5], s42, %[dz]\n"                                                                                          \
  "v_mul_f32_e64 %[t6], %[dy], -s42\n"                                                                                          \
  "v_mul_f32_e32 %[t0], s39, %[t4]\n"                                                                                           \
  "v_fmac_f32_e32 %[t6], s43, %[dx]\n"                                                                                          \
  "v_fmac_f32_e32 %[t0], s40, %[t5]\n"                                                                                          \
  "v_fmac_f32_e32 %[t0], s41, %[t6]\n"                                                                                          \
  "v_add_f32_e32 %[t0], 0, %[t0]\n"                 /* det */                                                                   \
  "v_div_scale_f32 %[t1], s[58:59], %[t0], %[t0], 1.0\n"                                                                        \
  "v_rcp_f32_e32 %[t2], %[t1]\n"                                                                                                \
  "v_cmpx_nlt_f32_e64 s[58:59], |%[t0]|, %[eps]\n"  /* !(fabs(det) < epsilon) */                                                \
  "v_subrev_f32_e32 %[t7], s36, %[ox]\n"            /* tvec = o - A */                                                          \
  "v_subrev_f32_e32 %[t8], s37, %[oy]\n"                                                                                        \
  "v_fma_f32 %[t3], -%[t1], %[t2], 1.0\n"                                                                                       \
  "v_fmac_f32_e32 %[t2], %[t3], %[t2]\n"                                                                                        \
  "v_div_scale_f32 %[t3], vcc, 1.0, %[t0], 1.0\n"                                                                               \
  "v_mul_f32_e32 %[t9], %[t3], %[t2]\n"                                                                                         \
  "v_fma_f32 %[t10], -%[t1], %[t9], %[t3]\n"                                                                                    \
  "v_fmac_f32_e32 %[t9], %[t10], %[t2]\n"                                                                                       \
  "v_fma_f32 %[t1], -%[t1], %[t9], %[t3]\n"                                                                                     \
  "v_div_fmas_f32 %[t1], %[t1], %[t2], %[t9]\n"                                                                                 \
  "v_div_fixup_f32 %[t0], %[t1], %[t0], 1.0\n"      /* invDet = 1 / det */                                                      \
  "v_subrev_f32_e32 %[t9], s38, %[oz]\n"                                                                                        \
  "v_mul_f32_e32 %[t1], %[t7], %[t4]\n"                                                                                         \
  "v_fmac_f32_e32 %[t1], %[t8], %[t5]\n"                                                                                        \
  "v_fmac_f32_e32 %[t1], %[t9], %[t6]\n"                                                                                        \
  "v_add_f32_e32 %[t1], 0, %[t1]\n"                                                                                             \
  "v_mul_f32_e32 %[t1], %[t1], %[t0]\n"             /* u */                                                                     \
  "v_cmpx_ngt_f32_e64 s[58:59], 0, %[t1]\n"         /* !(u < 0) */                                                              \
  "v_cmpx_nlt_f32_e64 s[58:59], 1.0, %[t1]\n"       /* !(u > 1) */                                                              \
  "s_cbranch_execz .LleafEnd%=\n"                                                                                               \
  "v_mul_f32_e64 %[t4], %[t9], -s40\n"              /* qvec = cross(tvec, e1) */                                                \
  "v_fmac_f32_e32 %[t4], s41, %[t8]\n"                                                                                          \
  "v_mul_f32_e64 %[t5], %[t7], -s41\n"                                                                                          \
  "v_mul_f32_e64 %[t6], %[t8], -s39\n"                                                                                          \
  "v_fmac_f32_e32 %[t5], s39, %[t9]\n"                                                                                          \
  "v_fmac_f32_e32 %[t6], s40, %[t7]\n"                                                                                          \
  "v_mul_f32_e32 %[t2], %[dx], %[t4]\n"                                                                                         \
  "v_fmac_f32_e32 %[t2], %[dy], %[t5]\n"                                                                                        \
  "v_fmac_f32_e32 %[t2], %[dz], %[t6]\n"                                                                                        \
  "v_fmac_f32_e32 %[t2], 0, %[dw]\n"                                                                                            \
  "v_mul_f32_e32 %[t2], %[t2], %[t0]\n"             /* v */                                                                     \
  "v_add_f32_e32 %[t10], %[t1], %[t2]\n"            /* u + v */                                                                 \
  "v_mul_f32_e32 %[t3], s42, %[t4]\n"                                                                                           \
  "v_fmac_f32_e32 %[t3], s43, %[t5]\n"                                                                                          \
  "v_fmac_f32_e32 %[t3], s44, %[t6]\n"                                                                                          \
  "v_cmpx_ngt_f32_e64 s[58:59], 0, %[t2]\n"         /* !(v < 0) */                                                              \
  "v_add_f32_e32 %[t3], 0, %[t3]\n"                                                                                             \
  "v_cmpx_nlt_f32_e64 s[58:59], 1.0, %[t10]\n"      /* !(u + v > 1) */                                                          \
  "v_mul_f32_e32 %[t3], %[t3], %[t0]\n"             /* t */                                                                     \
  "v_cmpx_lt_f32_e64 s[58:59], %[t3], %[pt]\n"      /* t < payload.t (no t > 0 test in the reference) */                        \
  "v_mov_b32_e32 %[pt], %[t3]\n"                    /* the lanes still in EXEC take the hit */                                  \
  "v_mov_b32_e32 %[pu], %[t1]\n"                                                                                                \
  "v_mov_b32_e32 %[pv], %[t2]\n"                                                                                                \
  "v_mov_b32_e32 %[pprim], s54\n"                                                                                               \
  "v_mov_b32_e32 %[phit], 1\n"                                                                                                  \
  ".LleafEnd%=:\n"                                                                                                              \
  "s_mov_b64 exec, s[56:57]\n"                                                                                                  \
  "s_cmp_eq_u32 %[cur], -1\n"                                                                                                   \
  "s_cbranch_scc0 .Ltop%=\n"                                                                                                    \
  "s_branch .Lpop%=\n"                                                                                                          \
  ".Ldone%=:\n"                                                                                                                 \
  "s_mov_b64 exec, s[56:57]\n"

// The whole closest-hit walk below the root.  `cur` = the root's reference (index 0 | its split axis << 29), `mask` = the lanes
// that hit the root's box.
template <int NEG>
__device__ __forceinline__ void packet_closest_walk(const void* pairs, const void* tris, float ox, float oy, float oz, float ix, float iy,
                                                    float iz, float dx, float dy, float dz, float dw, float eps, uint32_t lds, uint32_t cur,
                                                    lt_u64 mask, float& pt, float& pu, float& pv, int& pprim, int& phit) {
  uint32_t sp = 0u;
  float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
#define LT_CLOSEST_INSTANCE(LNX, LNY, LNZ, LFX, LFY, LFZ, RNX, RNY, RNZ, RFX, RFY, RFZ, NEGBITS)                                          \
  asm volatile(LT_ASM_CLOSEST_WALK(LNX, LNY, LNZ, LFX, LFY, LFZ, RNX, RNY, RNZ, RFX, RFY, RFZ, NEGBITS)                                   \
               : [cur] "+s"(cur), [mask] "+s"(mask), [sp] "+s"(sp), [pt] "+v"(pt), [pu] "+v"(pu), [pv] "+v"(pv), [pprim] "+v"(pprim),    \
                 [phit] "+v"(phit), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),     \
                 [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9), [t10] "=&v"(t10)                                        \
               : [pairs] "s"(pairs), [tris] "s"(tris), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), \
                 [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [dw] "v"(dw), [eps] "s"(eps), [lds] "v"(lds)                                  \
               : LT_ASM_CLOBBERS, "s55", "vcc")
  if constexpr (NEG == 0) LT_CLOSEST_INSTANCE("s36", "s37", "s38", "s39", "s40", "s41", "s44", "s45", "s46", "s47", "s48", "s49", "0");
  else if constexpr (NEG == 1) LT_CLOSEST_INSTANCE("s39", "s37", "s38", "s36", "s40", "s41", "s47", "s45", "s46", "s44", "s48", "s49", "1");
  else if constexpr (NEG == 2) LT_CLOSEST_INSTANCE("s36", "s40", "s38", "s39", "s37", "s41", "s44", "s48", "s46", "s47", "s45", "s49", "2");
  else if constexpr (NEG == 3) LT_CLOSEST_INSTANCE("s39", "s40", "s38", "s36", "s37", "s41", "s47", "s48", "s46", "s44", "s45", "s49", "3");
  else if constexpr (NEG == 4) LT_CLOSEST_INSTANCE("s36", "s37", "s41", "s39", "s40", "s38", "s44", "s45", "s49", "s47", "s48", "s46", "4");
  else if constexpr (NEG == 5) LT_CLOSEST_INSTANCE("s39", "s37", "s41", "s36", "s40", "s38", "s47", "s45", "s49", "s44", "s48", "s46", "5");
  else if constexpr (NEG == 6) LT_CLOSEST_INSTANCE("s36", "s40", "s41", "s39", "s37", "s38", "s44", "s48", "s49", "s47", "s45", "s46", "6");
  else LT_CLOSEST_INSTANCE("s39", "s40", "s41", "s36", "s37", "s38", "s47", "s48", "s49", "s44", "s45", "s46", "7");
#undef LT_CLOSEST_INSTANCE
}

}  // namespace lt
