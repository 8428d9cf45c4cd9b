// queue_walk_k2: the lean walk of k_pass_queue (k_pass_queue.h) for k = 1 or 2, NumPartitions > 0 -- the passes of BASELINE
// config 5, and with the row of nodeToNodeCounts folded into the keys the runs of partitions that lost their primary in a
// rebalance -- as hand-written gfx950 assembly.  Part of tu_queue.hip.
//
// Why: a lone wave issues one instruction per ~4.5 cycles whatever its kind (tools/dev_lat_micro.hip), so a moving step is
// an instruction-count problem, and the compiler's rendering of the C++ walk (the twin of this text, `lean walk` in
// k_pass_queue.h: it is what the SIMT emulator runs and what takes every step this text leaves) costs ~4,300 cycles per
// moving step: ~500 executed instructions, 240 spilled scalars read back lane by lane, wave-uniform values carried in
// vector registers.  This text keeps the step's scalars in named SGPRs, compares (key, node) pairs with three-instruction
// borrow chains (s_sub_u32 / s_subb_u32 / s_subb_u32: the 96-bit number hi:lo:node), and has no spill.
//
// What it does, per iteration (same decisions as the C++ walk, in the same order):
//   front of the window -> which lanes certainly stay -> first lane f that does not; f needs the general code (slowmask,
//   stale) -> leave.  Else: f's weight, own nodes (sorted, exact keys), higher priority node; eligible window entries; their
//   nodeToNodeCounts bits from the batch's row bit maps in LDS; the two first clean entries t1, t2 with no set bit in
//   front of t2, or only such whose score with an entry of 1 -- a lower bound -- lies above t2 (else leave: the general
//   code reads the matrix); the two best of
//   (a, b, t1, t2); below THETA (else leave); no promotion (else leave); commit: lanes 0..3 settle the leaving / entering
//   nodes (new counters, new key: cvt, + ffT[tot], ldexp by the node's power-of-two weight, sortable image), the window
//   removes and re-inserts them (DPP wave shifts), lanes that hold a changed node are marked stale, f's choice goes to its
//   output registers.  Leaves with code 0 (the batch is done, cur = B), 1 (lane cur needs the general code) or 2 (lane cur may still be the C++ twin's).
//
// Registers are fixed by the operand constraints (an operand's halves have to be named):
//   v200:201 window key, v202 window node, v203 / v204 output nodes of the lanes that moved (in / out)
//   v206:207 lastK, v205 lastN, v208 / v209 sorted own nodes (-1: none), v210:211 / v212:213 their exact keys, v214 higher
//   priority node, v215 weight, v216 lower priority nodes (two 16-bit fields, 0xffff: none), v217 lane id (in)
//   s40 cur, s41 window count, s42:43 / s44 THETA, s46:47 stale lanes, s48:49 lanes that moved (in / out)
//   s50:51 lanes that never stay, s52:53 lanes for the general code, s54:55 active lanes, s56..59 LDS layout (packed; s59 bit 25:
//   the row is folded), s45 where the folded row and the table of c / NumPartitions lie (packed), s60 code,
//   s38:39 1 / NumPartitions (the nodeToNodeCounts term of an entry of 1, plan.go:638-644)
// Temporaries: v218..v239, s61..s101, vcc, m0.
#pragma once

namespace blance {

#ifndef BLANCE_SIMT_EMU

// (hi:lo:node of a) < (hi:lo:node of b) -> SCC
#define BLANCE_QW_LT96(an, alo, ahi, bn, blo, bhi) \
    "s_sub_u32 s101, " an ", " bn "\n\t"           \
    "s_subb_u32 s101, " alo ", " blo "\n\t"        \
    "s_subb_u32 s101, " ahi ", " bhi "\n\t"

// the window loses node X (if it holds it) and takes it back with the key of lane J of v234:235 (if that lies below THETA);
// lanes whose own nodes include X are stale.  X: an SGPR holding the node or a negative value (nothing to do).
#define BLANCE_QW_UPDATE(J, X)                                                   \
    "s_cmp_lt_i32 " X ", 0\n\t"                                                  \
    "s_cbranch_scc1 39" J "f\n\t"                                                \
    "v_readlane_b32 s84, v234, " J "\n\t"                                        \
    "v_readlane_b32 s85, v235, " J "\n\t"                                        \
    "v_cmp_eq_u32_e64 s[74:75], " X ", v202\n\t"                                 \
    "s_cmp_eq_u64 s[74:75], 0\n\t"                                               \
    "s_cbranch_scc1 31" J "f\n\t"                                                \
    "s_ff1_i32_b64 s86, s[74:75]\n\t"                                            \
    "s_lshl_b64 s[74:75], -1, s86\n\t"                                           \
    "v_mov_b32_e32 v237, -1\n\t"                                                 \
    "v_mov_b32_e32 v238, -1\n\t"                                                 \
    "v_bfrev_b32_e32 v239, -2\n\t"                                               \
    "s_nop 1\n\t"                                                                \
    "v_mov_b32_dpp v237, v200 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mov_b32_dpp v238, v201 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mov_b32_dpp v239, v202 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_cndmask_b32_e64 v200, v200, v237, s[74:75]\n\t"                           \
    "v_cndmask_b32_e64 v201, v201, v238, s[74:75]\n\t"                           \
    "v_cndmask_b32_e64 v202, v202, v239, s[74:75]\n\t"                           \
    "s_sub_u32 s41, s41, 1\n"                                                    \
    "31" J ":\n\t"                                                               \
    BLANCE_QW_LT96(X, "s84", "s85", "s44", "s42", "s43")                         \
    "s_cbranch_scc0 38" J "f\n\t"                                                \
    "v_cmp_lt_u64_e64 s[74:75], v[200:201], s[84:85]\n\t"                        \
    "v_cmp_eq_u64_e64 s[76:77], v[200:201], s[84:85]\n\t"                        \
    "v_cmp_lt_i32_e64 s[78:79], v202, " X "\n\t"                                 \
    "s_and_b64 s[76:77], s[76:77], s[78:79]\n\t"                                 \
    "s_or_b64 s[74:75], s[74:75], s[76:77]\n\t"                                  \
    "s_bcnt1_i32_b64 s86, s[74:75]\n\t"                                          \
    "s_cmp_lt_u32 s86, 64\n\t"                                                   \
    "s_cbranch_scc1 32" J "f\n\t"                                                \
    "s_mov_b64 s[42:43], s[84:85]\n\t"                                           \
    "s_mov_b32 s44, " X "\n\t"                                                   \
    "s_branch 38" J "f\n"                                                        \
    "32" J ":\n\t"                                                               \
    "s_cmp_lt_u32 s41, 64\n\t"                                                   \
    "s_cbranch_scc1 33" J "f\n\t"                                                \
    "v_readlane_b32 s42, v200, 63\n\t"                                           \
    "v_readlane_b32 s43, v201, 63\n\t"                                           \
    "v_readlane_b32 s44, v202, 63\n\t"                                           \
    "s_sub_u32 s41, s41, 1\n"                                                    \
    "33" J ":\n\t"                                                               \
    "s_lshl_b64 s[74:75], -2, s86\n\t"                                           \
    "s_nop 1\n\t"                                                                \
    "v_mov_b32_dpp v237, v200 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mov_b32_dpp v238, v201 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mov_b32_dpp v239, v202 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_cndmask_b32_e64 v200, v200, v237, s[74:75]\n\t"                           \
    "v_cndmask_b32_e64 v201, v201, v238, s[74:75]\n\t"                           \
    "v_cndmask_b32_e64 v202, v202, v239, s[74:75]\n\t"                           \
    "s_mov_b32 m0, s86\n\t"                                                      \
    "v_writelane_b32 v200, s84, m0\n\t"                                          \
    "v_writelane_b32 v201, s85, m0\n\t"                                          \
    "v_writelane_b32 v202, " X ", m0\n\t"                                        \
    "s_add_u32 s41, s41, 1\n"                                                    \
    "38" J ":\n\t"                                                               \
    "v_cmp_eq_u32_e64 s[74:75], " X ", v208\n\t"                                 \
    "v_cmp_eq_u32_e64 s[76:77], " X ", v209\n\t"                                 \
    "s_or_b64 s[46:47], s[46:47], s[74:75]\n\t"                                  \
    "s_or_b64 s[46:47], s[46:47], s[76:77]\n"                                    \
    "39" J ":\n\t"

#define BLANCE_QW_TEXT                                                           \
    /* LDS layout */                                                             \
    "s_and_b32 s61, s56, 0xffff\n\t"                                             \
    "s_lshl_b32 s61, s61, 2\n\t"                                                 \
    "s_lshr_b32 s62, s56, 16\n\t"                                                \
    "s_lshl_b32 s62, s62, 2\n\t"                                                 \
    "s_and_b32 s63, s57, 0xffff\n\t"                                             \
    "s_lshl_b32 s63, s63, 2\n\t"                                                 \
    "s_lshr_b32 s64, s57, 16\n\t"                                                \
    "s_lshl_b32 s64, s64, 2\n\t"                                                 \
    "s_and_b32 s65, s58, 0xffff\n\t"                                             \
    "s_lshl_b32 s65, s65, 2\n\t"                                                 \
    "s_lshr_b32 s66, s58, 16\n\t"                                                \
    "s_and_b32 s67, s59, 0xffff\n\t"                                             \
    "s_lshl_b32 s67, s67, 2\n\t"                                                 \
    "s_bfe_u32 s68, s59, 0x80010\n\t"           /* B: bits 16..23 */             \
    "s_bfe_u32 s69, s59, 0x10018\n"             /* bit 24: k = 2 (else 1) */      \
    /* ---- one step per iteration */                                            \
    "10:\n\t"                                                                    \
    "s_cmp_ge_u32 s40, s68\n\t"                                                  \
    "s_cbranch_scc1 80f\n\t"                                                     \
    "s_mov_b64 s[70:71], s[42:43]\n\t"                                           \
    "s_mov_b32 s72, s44\n\t"                                                     \
    "s_cmp_eq_u32 s41, 0\n\t"                                                    \
    "s_cbranch_scc1 11f\n\t"                                                     \
    "v_readfirstlane_b32 s70, v200\n\t"                                          \
    "v_readfirstlane_b32 s71, v201\n\t"                                          \
    "v_readfirstlane_b32 s72, v202\n\t"                                         \
    "s_nop 1\n"                                                                  \
    "11:\n\t"                                                                    \
    "v_cmp_gt_u64_e64 s[74:75], v[206:207], s[70:71]\n\t"                        \
    "v_cmp_eq_u64_e64 s[76:77], v[206:207], s[70:71]\n\t"                        \
    "v_cmp_ge_i32_e64 s[78:79], v205, s72\n\t"                                   \
    "s_and_b64 s[76:77], s[76:77], s[78:79]\n\t"                                 \
    "s_or_b64 s[74:75], s[74:75], s[76:77]\n\t"                                  \
    "s_or_b64 s[74:75], s[74:75], s[50:51]\n\t"                                  \
    "s_or_b64 s[74:75], s[74:75], s[46:47]\n\t"                                  \
    "s_and_b64 s[74:75], s[74:75], s[54:55]\n\t"                                 \
    "s_lshl_b64 s[76:77], -1, s40\n\t"                                           \
    "s_and_b64 s[74:75], s[74:75], s[76:77]\n\t"                                 \
    "s_cbranch_scc0 79f\n\t"                                                     \
    "s_ff1_i32_b64 s73, s[74:75]\n\t"                                            \
    "s_or_b64 s[76:77], s[52:53], s[46:47]\n\t"                                  \
    "s_bitcmp1_b64 s[76:77], s73\n\t"                                            \
    "s_cbranch_scc1 84f\n\t"                                                     \
    /* the step's own data */                                                    \
    "v_readlane_b32 s80, v215, s73\n\t"                                          \
    "v_readlane_b32 s81, v208, s73\n\t"                                          \
    "v_readlane_b32 s82, v209, s73\n\t"                                          \
    "v_readlane_b32 s83, v214, s73\n\t"                                          \
    "v_readlane_b32 s84, v210, s73\n\t"                                          \
    "v_readlane_b32 s85, v211, s73\n\t"                                          \
    "v_readlane_b32 s86, v212, s73\n\t"                                          \
    "v_readlane_b32 s87, v213, s73\n\t"                                          \
    "v_readlane_b32 s88, v216, s73\n\t"                                          \
    /* eligible window entries, and which of them have a nodeToNodeCounts entry in the step's row */ \
    "v_cmp_ne_u32_e64 s[74:75], s81, v202\n\t"                                   \
    "v_cmp_ne_u32_e64 s[76:77], s82, v202\n\t"                                   \
    "v_cmp_ne_u32_e64 s[78:79], s83, v202\n\t"                                   \
    "s_and_b64 s[74:75], s[74:75], s[76:77]\n\t"                                 \
    "s_and_b64 s[74:75], s[74:75], s[78:79]\n\t"                                 \
    "v_cmp_gt_u32_e64 s[76:77], s41, v217\n\t"                                   \
    "s_and_b64 s[74:75], s[74:75], s[76:77]\n\t"                                 \
    "v_and_b32_e32 v218, 0xfff, v202\n\t"                                        \
    "v_lshrrev_b32_e32 v218, 5, v218\n\t"                                        \
    "s_mul_i32 s89, s73, s66\n\t"                                                \
    "s_add_u32 s89, s89, s65\n\t"                                                \
    "v_lshl_add_u32 v218, v218, 2, s89\n\t"                                      \
    "ds_read_b32 v219, v218\n\t"                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                   \
    "v_lshrrev_b32_e32 v219, v202, v219\n\t"                                     \
    "v_and_b32_e32 v219, 1, v219\n\t"                                            \
    "v_cmp_eq_u32_e64 s[76:77], 1, v219\n\t"                                     \
    "s_and_b64 s[76:77], s[76:77], s[74:75]\n\t"                                 \
    "s_bitcmp1_b32 s59, 25\n\t"                 /* folded row (bit 25): every window key is exact, no entry to look up */ \
    "s_cselect_b64 s[76:77], 0, s[76:77]\n\t"                                    \
    "s_andn2_b64 s[78:79], s[74:75], s[76:77]\n\t"                               \
    "s_add_u32 s90, s78, -1\n\t"                                                 \
    "s_addc_u32 s91, s79, -1\n\t"                                                \
    "s_and_b64 s[90:91], s[90:91], s[78:79]\n\t"                                 \
    "s_cmp_eq_u32 s69, 0\n\t"                   /* k = 1: the first clean entry bounds the search */ \
    "s_cselect_b64 s[90:91], s[78:79], s[90:91]\n\t"                             \
    "s_cmp_eq_u64 s[90:91], 0\n\t"                                               \
    "s_cbranch_scc1 83f\n\t"                                                     \
    "s_ff1_i32_b64 s92, s[78:79]\n\t"                                            \
    "s_ff1_i32_b64 s93, s[90:91]\n\t"                                            \
    "s_lshl_b64 s[90:91], 2, s93\n\t"                                            \
    "s_sub_u32 s90, s90, 1\n\t"                                                  \
    "s_subb_u32 s91, s91, 0\n\t"                                                 \
    "s_and_b64 s[90:91], s[90:91], s[76:77]\n\t"                                 \
    "s_cbranch_scc0 15f\n\t"                                                     \
    /* entries with their bit set in front of t2: scored with an entry of 1 -- a lower bound of their exact score -- they   \
       are out of the race if even that lies above t2's key; else the caller reads the matrix */                          \
    "v_readlane_b32 s76, v200, s93\n\t"                                          \
    "v_readlane_b32 s77, v201, s93\n\t"                                          \
    "s_mov_b64 s[94:95], exec\n\t"                                               \
    "s_mov_b64 exec, s[90:91]\n\t"                                               \
    "v_lshlrev_b32_e32 v221, 2, v202\n\t"                                        \
    "v_add_u32_e32 v222, s61, v221\n\t"                                          \
    "v_add_u32_e32 v223, s62, v221\n\t"                                          \
    "v_add_u32_e32 v224, s63, v202\n\t"                                          \
    "ds_read_b32 v225, v222\n\t"                                                 \
    "ds_read_b32 v226, v223\n\t"                                                 \
    "ds_read_u8 v227, v224\n\t"                                                  \
    "s_waitcnt lgkmcnt(1)\n\t"                                                   \
    "v_cmp_le_u32_e32 vcc, 0x800, v226\n\t"                                      \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                    \
    "s_cbranch_scc1 82f\n\t"                                                     \
    "v_lshl_add_u32 v230, v226, 3, s64\n\t"                                      \
    "ds_read_b64 v[232:233], v230\n\t"                                           \
    "v_cvt_f64_i32_e32 v[234:235], v225\n\t"                                     \
    "v_add_f64 v[234:235], v[234:235], s[38:39]\n\t"                             \
    "s_waitcnt lgkmcnt(0)\n\t"                                                   \
    "v_sub_u32_e32 v231, 0, v227\n\t"                                            \
    "v_add_f64 v[234:235], v[234:235], v[232:233]\n\t"                           \
    "v_ldexp_f64 v[234:235], v[234:235], v231\n\t"                               \
    "v_cmp_eq_f64_e32 vcc, 0, v[234:235]\n\t"                                    \
    "v_cndmask_b32_e64 v234, v234, 0, vcc\n\t"                                   \
    "v_cndmask_b32_e64 v235, v235, 0, vcc\n\t"                                   \
    "v_ashrrev_i32_e32 v236, 31, v235\n\t"                                       \
    "v_xor_b32_e32 v234, v234, v236\n\t"                                         \
    "v_or_b32_e32 v236, 0x80000000, v236\n\t"                                    \
    "v_xor_b32_e32 v235, v235, v236\n\t"                                         \
    "v_cmp_ge_u64_e32 vcc, s[76:77], v[234:235]\n\t"                             \
    "s_mov_b64 exec, s[94:95]\n\t"                                               \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                    \
    "s_cbranch_scc1 84f\n"                                                        \
    "15:\n\t"                                                                    \
    "v_readlane_b32 s74, v200, s92\n\t"                                          \
    "v_readlane_b32 s75, v201, s92\n\t"                                          \
    "v_readlane_b32 s76, v202, s92\n\t"                                          \
    "v_readlane_b32 s77, v200, s93\n\t"                                          \
    "v_readlane_b32 s78, v201, s93\n\t"                                          \
    "v_readlane_b32 s79, v202, s93\n\t"                                          \
    /* the two best of (a, b) and (t1, t2): s96 / s97 the result, s92 / s93 leave, s90 / s91 enter, s98..100 the last one's (node, lo, hi) */ \
    "s_cmp_lt_i32 s81, 0\n\t"                                                    \
    "s_cselect_b32 s94, 0x7fffffff, s81\n\t"                                     \
    "s_cmp_lt_i32 s82, 0\n\t"                                                    \
    "s_cselect_b32 s95, 0x7fffffff, s82\n\t"                                     \
    "s_mov_b32 s91, -2\n\t"                                                      \
    "s_mov_b32 s93, -2\n\t"                                                      \
    "s_cmp_eq_u32 s69, 0\n\t"                                                    \
    "s_cbranch_scc1 25f\n\t"                                                     \
    BLANCE_QW_LT96("s94", "s84", "s85", "s76", "s74", "s75")                     \
    "s_cbranch_scc0 20f\n\t"                                                     \
    BLANCE_QW_LT96("s95", "s86", "s87", "s76", "s74", "s75")                     \
    "s_cbranch_scc1 70f\n\t"                     /* both own nodes stay, in this order: nothing changes */ \
    "s_mov_b32 s96, s81\n\t"                     /* (a, t1): t1 enters, b leaves */ \
    "s_mov_b32 s97, s76\n\t"                                                     \
    "s_mov_b32 s90, s76\n\t"                                                     \
    "s_mov_b32 s92, s82\n\t"                                                     \
    "s_mov_b32 s98, s76\n\t"                                                     \
    "s_mov_b32 s99, s74\n\t"                                                     \
    "s_mov_b32 s100, s75\n\t"                                                    \
    "s_branch 22f\n"                                                             \
    "20:\n\t"                                                                    \
    BLANCE_QW_LT96("s94", "s84", "s85", "s79", "s77", "s78")                     \
    "s_cbranch_scc0 21f\n\t"                                                     \
    "s_mov_b32 s96, s76\n\t"                     /* (t1, a): t1 enters, b leaves */ \
    "s_mov_b32 s97, s81\n\t"                                                     \
    "s_mov_b32 s90, s76\n\t"                                                     \
    "s_mov_b32 s92, s82\n\t"                                                     \
    "s_mov_b32 s98, s94\n\t"                                                     \
    "s_mov_b32 s99, s84\n\t"                                                     \
    "s_mov_b32 s100, s85\n\t"                                                    \
    "s_branch 22f\n"                                                             \
    "21:\n\t"                                                                    \
    "s_mov_b32 s96, s76\n\t"                     /* (t1, t2): both enter, a and b leave */ \
    "s_mov_b32 s97, s79\n\t"                                                     \
    "s_mov_b32 s90, s76\n\t"                                                     \
    "s_mov_b32 s91, s79\n\t"                                                     \
    "s_mov_b32 s92, s81\n\t"                                                     \
    "s_mov_b32 s93, s82\n\t"                                                     \
    "s_mov_b32 s98, s79\n\t"                                                     \
    "s_mov_b32 s99, s77\n\t"                                                     \
    "s_mov_b32 s100, s78\n\t"                                                    \
    "s_branch 22f\n"                                                             \
    "25:\n\t"                                   /* k = 1: the better of a and t1 */ \
    BLANCE_QW_LT96("s94", "s84", "s85", "s76", "s74", "s75")                     \
    "s_cbranch_scc1 70f\n\t"                                                     \
    "s_mov_b32 s96, s76\n\t"                    /* t1 enters, a leaves */         \
    "s_mov_b32 s97, -1\n\t"                                                      \
    "s_mov_b32 s90, s76\n\t"                                                     \
    "s_mov_b32 s92, s81\n\t"                                                     \
    "s_mov_b32 s98, s76\n\t"                                                     \
    "s_mov_b32 s99, s74\n\t"                                                     \
    "s_mov_b32 s100, s75\n"                                                      \
    "22:\n\t"                                                                    \
    BLANCE_QW_LT96("s98", "s99", "s100", "s44", "s42", "s43")                    \
    "s_cbranch_scc0 81f\n\t"                     /* not below THETA: the caller rebuilds the window or scores every node */ \
    "s_sext_i32_i16 s94, s88\n\t"                /* a node held in a lower priority state would be promoted: the caller's business */ \
    "s_ashr_i32 s95, s88, 16\n\t"                                                \
    "s_cmp_eq_u32 s90, s94\n\t"                                                  \
    "s_cbranch_scc1 81f\n\t"                                                     \
    "s_cmp_eq_u32 s90, s95\n\t"                                                  \
    "s_cbranch_scc1 81f\n\t"                                                     \
    "s_cmp_eq_u32 s91, s94\n\t"                                                  \
    "s_cbranch_scc1 81f\n\t"                                                     \
    "s_cmp_eq_u32 s91, s95\n\t"                                                  \
    "s_cbranch_scc1 81f\n\t"                                                     \
    /* ---- commit: lanes 0, 1 the leaving nodes, lanes 2, 3 the entering ones */ \
    "v_mov_b32_e32 v220, -1\n\t"                                                 \
    "v_writelane_b32 v220, s92, 0\n\t"                                           \
    "v_writelane_b32 v220, s93, 1\n\t"                                           \
    "v_writelane_b32 v220, s90, 2\n\t"                                           \
    "v_writelane_b32 v220, s91, 3\n\t"                                           \
    "v_cmp_le_i32_e32 vcc, 0, v220\n\t"                                          \
    "s_and_saveexec_b64 s[94:95], vcc\n\t"                                       \
    "v_lshlrev_b32_e32 v221, 2, v220\n\t"                                        \
    "v_add_u32_e32 v222, s61, v221\n\t"                                          \
    "v_add_u32_e32 v223, s62, v221\n\t"                                          \
    "v_add_u32_e32 v224, s63, v220\n\t"                                          \
    "ds_read_b32 v225, v222\n\t"                                                 \
    "ds_read_b32 v226, v223\n\t"                                                 \
    "ds_read_u8 v227, v224\n\t"                                                  \
    "v_mov_b32_e32 v228, s80\n\t"                                                \
    "v_sub_u32_e32 v229, 0, v228\n\t"                                            \
    "v_cmp_gt_u32_e32 vcc, 2, v217\n\t"                                          \
    "v_cndmask_b32_e32 v228, v228, v229, vcc\n\t"                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                   \
    "v_add_u32_e32 v225, v225, v228\n\t"                                         \
    "v_add_u32_e32 v226, v226, v228\n\t"                                         \
    "v_cmp_le_u32_e32 vcc, 0x800, v226\n\t"      /* beyond the table of (0.001 t) / NP: the caller divides */ \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                    \
    "s_cbranch_scc1 82f\n\t"                                                     \
    "v_lshl_add_u32 v230, v226, 3, s64\n\t"                                      \
    "ds_read_b64 v[232:233], v230\n\t"                                           \
    "v_cvt_f64_i32_e32 v[234:235], v225\n\t"                                     \
    "v_sub_u32_e32 v231, 0, v227\n\t"                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                   \
    /* folded row: the node's entry of row "" (u16 in LDS; one more for a chosen node, plan.go:238-245) is part of its key: \
       + lpT[entry] before + ffT[tot] -- the order of nodeSorter.Score.  Beyond the table: the caller divides */           \
    "s_bitcmp1_b32 s59, 25\n\t"                                                  \
    "s_cbranch_scc1 61f\n"                     /* (out of line, behind the loop: the plain case falls through) */ \
    "60:\n\t"                                                                    \
    "v_add_f64 v[234:235], v[234:235], v[232:233]\n\t"                           \
    "v_ldexp_f64 v[234:235], v[234:235], v231\n\t"                               \
    "v_cmp_eq_f64_e32 vcc, 0, v[234:235]\n\t"                                    \
    "v_cndmask_b32_e64 v234, v234, 0, vcc\n\t"                                   \
    "v_cndmask_b32_e64 v235, v235, 0, vcc\n\t"                                   \
    "v_ashrrev_i32_e32 v236, 31, v235\n\t"                                       \
    "v_xor_b32_e32 v234, v234, v236\n\t"                                         \
    "v_or_b32_e32 v236, 0x80000000, v236\n\t"                                    \
    "v_xor_b32_e32 v235, v235, v236\n\t"                                         \
    "ds_write_b32 v222, v225\n\t"                                                \
    "ds_write_b32 v223, v226\n\t"                                                \
    "v_lshl_add_u32 v230, v220, 3, s67\n\t"                                      \
    "ds_write_b64 v230, v[234:235]\n\t"                                          \
    "s_mov_b64 exec, s[94:95]\n\t"                                               \
    BLANCE_QW_UPDATE("0", "s92")                                                 \
    BLANCE_QW_UPDATE("1", "s93")                                                 \
    BLANCE_QW_UPDATE("2", "s90")                                                 \
    BLANCE_QW_UPDATE("3", "s91")                                                 \
    "s_mov_b32 m0, s73\n\t"                                                      \
    "v_writelane_b32 v203, s96, m0\n\t"                                          \
    "v_writelane_b32 v204, s97, m0\n\t"                                          \
    "s_bitset1_b64 s[48:49], s73\n"                                              \
    "70:\n\t"                                                                    \
    "s_add_u32 s40, s73, 1\n\t"                                                  \
    "s_branch 10b\n"                                                             \
    "61:\n\t"                                                                    \
    "s_and_b32 s89, s45, 0xffff\n\t"                                             \
    "s_lshl_b32 s89, s89, 2\n\t"                                                 \
    "v_lshl_add_u32 v218, v220, 1, s89\n\t"                                      \
    "ds_read_u16 v219, v218\n\t"                                                 \
    "v_cmp_lt_u32_e32 vcc, 1, v217\n\t"                                          \
    "v_cndmask_b32_e64 v229, 0, 1, vcc\n\t"                                      \
    "s_lshr_b32 s89, s45, 16\n\t"                                                \
    "s_lshl_b32 s89, s89, 2\n\t"                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                   \
    "v_add_u32_e32 v219, v219, v229\n\t"                                         \
    "v_cmp_le_u32_e32 vcc, 0x200, v219\n\t"                                      \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                    \
    "s_cbranch_scc1 82f\n\t"                                                     \
    "v_lshl_add_u32 v229, v219, 3, s89\n\t"                                      \
    "ds_read_b64 v[236:237], v229\n\t"                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                   \
    "v_add_f64 v[234:235], v[234:235], v[236:237]\n\t"                           \
    "ds_write_b16 v218, v219\n"                                                  \
    "s_branch 60b\n"                                                             \
    "79:\n\t"                                                                    \
    "s_mov_b32 s40, s68\n"                                                       \
    "80:\n\t"                                                                    \
    "s_mov_b32 s60, 0\n\t"                                                       \
    "s_branch 89f\n"                                                             \
    "82:\n\t"                                                                    \
    "s_mov_b64 exec, s[94:95]\n"                                                 \
    "81:\n\t"                                 /* the C++ twin may take this step */  \
    "s_mov_b32 s40, s73\n\t"                                                     \
    "s_mov_b32 s60, 2\n\t"                                                       \
    "s_branch 89f\n"                                                             \
    "83:\n\t"                                 /* no clean entry: a window run dry is the twin's to rebuild */ \
    "s_cmp_lt_u32 s41, 32\n\t"                                                   \
    "s_cbranch_scc1 81b\n"                                                       \
    "84:\n\t"                                 /* the general code */               \
    "s_mov_b32 s40, s73\n\t"                                                     \
    "s_mov_b32 s60, 1\n"                                                         \
    "89:\n"

struct QueueWalkState {
    unsigned long long wk;
    int wn, o1, o2;
    int cur, wcnt;
    unsigned long long thK;
    int thN;
    unsigned long long stale, moved;
    int code;
};

__device__ __forceinline__ void queue_walk_k2(QueueWalkState& st, unsigned long long lastK, int lastN, int own_a, int own_b,
                                              unsigned long long ka, unsigned long long kb, int h0, int w, int ov, int lane,
                                              unsigned long long always, unsigned long long slow, unsigned long long act,
                                              int cfa, int cfb, int cfc, int cfd, int cfe, unsigned long long lp1_bits) {
    asm volatile(BLANCE_QW_TEXT
                 : "+{v[200:201]}"(st.wk), "+{v202}"(st.wn), "+{v203}"(st.o1), "+{v204}"(st.o2),
                   "+{s40}"(st.cur), "+{s41}"(st.wcnt), "+{s[42:43]}"(st.thK), "+{s44}"(st.thN),
                   "+{s[46:47]}"(st.stale), "+{s[48:49]}"(st.moved), "=&{s60}"(st.code)
                 : "{v[206:207]}"(lastK), "{v205}"(lastN), "{v208}"(own_a), "{v209}"(own_b), "{v[210:211]}"(ka), "{v[212:213]}"(kb),
                   "{v214}"(h0), "{v215}"(w), "{v216}"(ov), "{v217}"(lane),
                   "{s[50:51]}"(always), "{s[52:53]}"(slow), "{s[54:55]}"(act),
                   "{s56}"(cfa), "{s57}"(cfb), "{s58}"(cfc), "{s59}"(cfd), "{s45}"(cfe), "{s[38:39]}"(lp1_bits)
                 : "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231",
                   "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",
                   "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77",
                   "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93",
                   "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "vcc", "m0", "scc", "memory");
}

#endif  // !BLANCE_SIMT_EMU

}  // namespace blance
