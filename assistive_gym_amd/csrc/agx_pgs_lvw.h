// agx_pgs_lvw.h -- K6, the WIDE row-local sweep: up to four rows per visit, one per 16-lane group of the wavefront (AGX_PGS_LV == 4, the
// default of the feeding variant).  Part of the stepper (see agx_step.h); included by agx_step.h only, after agx_pgs_lvs.h.
//
// Why (round 6).  The row-local sweep of agx_pgs_lvs.h visits ONE row at a time under EXEC = lanes 0..15: a quarter of the wavefront works,
// and at four wavefronts per SIMD the kernel is bound by the number of instructions it issues per row (38; profiles/r05/r05m_*).  Two rows of
// a Gauss-Seidel sweep that touch disjoint velocity slots COMMUTE EXACTLY -- neither reads what the other writes -- so they may be visited in
// either order, or at the same time on different lanes, without changing a single bit of the result.  Per substep and part of the sweep
// (non-contact + normal rows; friction rows) this file list-schedules the rows in their original order onto STEPS of up to four rows: a
// row goes to the earliest step behind every EARLIER row of the part it shares a velocity slot with (lvw_schedule: lane = step, the slots a
// step has taken as a 96-bit mask per lane; ~25 instructions per row, once per substep).  Any two rows that do not commute keep their order,
// so the schedule computes bit for bit what the sequential sweep computes: tests/test_gpu_solve_variants.py compares the states of this
// kernel with those of agx_pgs_lvs.h after rollouts, tests/test_emu_parity.py the two C++ twins.
// How far it goes (tests/diag/solve_schedule_study.py, FeedingJaco, the emulator's own row sets and no-op masks): 117 rows per sweep, of which
// the no-op re-test rule lets 77 through; their steps: 56 -- 1.37 rows per step.  The bound is the scene, not the width: every food-on-spoon
// contact row touches the spoon's six slots and has to wait for the one before it (two lane groups would do: 56.1 against 55.7 steps); re-scheduling
// the active rows whenever a mask changes would reach 49 (1.56) at ~20 re-schedules per substep, which costs more than it saves.
//
// A visit of a step (lane = 16 g + k: pair k of the row of group g):
//   * the step's four row indices come from a list in LDS (one byte per group; rows the no-op rule or the friction rule skips are replaced by
//     the idle row, steps without an active row are dropped when the list is compacted: lvw_compact);
//   * the row's header from the two compact tables build_rows() leaves in the scratch record (agx_ctx.h: Q -- 1/D, b, lo, hi, a lane reads
//     word k & 3 and takes the others from its quad through DPP operand selects; P -- pair offset, pair counts, velocity slot offsets, packed
//     so that SDWA byte / word selects pick the fields without unpacking), requested three steps ahead;
//   * its pair from the LDS window (or, a step with a row beyond it, every lane from the scratch record), its velocity slot from LDS, the
//     4-step butterfly inside the DPP row, the impulse update in all 16 lanes alike, the scatter; impulses live in LDS (one word per row: the
//     16 lanes of a group store the same value to the same address, no EXEC juggling).
// 36 instructions per step (the narrow sweep: 38 per row).  LDS: velocity deltas, impulses, three step lists, the window -- 10 KB per wave.
#pragma once

namespace agx {

constexpr bool LVW_COMPILED = AGX_PGS_LV == 4 && LV_COMPILED;
constexpr int LVW_NG = 4;                                            // lane groups = rows per step
#ifndef AGX_LVW_MAX_STEPS            // (tests: a small limit makes ordinary substeps overflow the scheduler and take the fallback)
#define AGX_LVW_MAX_STEPS 64
#endif
constexpr int LVW_MAX_STEPS = AGX_LVW_MAX_STEPS;                     // lane = step in the scheduler; a part that needs more takes the narrow sweep
static_assert(LVW_MAX_STEPS <= 64, "lane = step");
constexpr int LVW_LIST = 64 + 4;                          // words of a step list: the loop's look-ahead reads up to four entries behind the last step
constexpr int LVW_DV = 0, LVW_LAM = 128, LVW_LISTS = LVW_LAM + HW_ROWS, LVW_PAIRS = LVW_LISTS + 3 * LVW_LIST;      // LDS words: dv[128], lam[rows + idle], lists[3], pairs[2 x window]
static_assert(LVW_PAIRS % 2 == 0, "(J,B) pairs are read as 8-byte words");
static_assert(!HDR_WIDE || (HP_BASE % 2 == 0 && HQ_BASE % 4 == 0), "the compact tables are read with 8- and 4-byte loads");
static_assert(HW_DUMMY < 255, "a step list holds row indices as bytes");
constexpr uint32_t LVW_IDLE = 0x01010101u * (uint32_t)HW_DUMMY;      // a step without rows

AGX_DEV int lvw_window(int lds_words) {
  int w = (lds_words - LVW_PAIRS) / 2;
#ifdef AGX_LV_WINDOW_CAP
  if (w > AGX_LV_WINDOW_CAP) w = AGX_LV_WINDOW_CAP;
#endif
  return w < 0 ? 0 : w;
}
// the slots of a row are a 96-bit mask (second header table): nv <= 96
AGX_DEV bool lvw_eligible(const Ctx& c, int lds_words) { return lvs_eligible(c, lds_words) && c.nv <= 96 && lds_words > LVW_PAIRS + 64 && PRM(c, AGX_P_SOLVE_WIDE) != 0.f; }

struct LvwLay { float* lds; const float* H; const float* E; int dv_addr, lam_addr, pairs_addr, far8; };

// ---- the static schedule of the rows [r0, r0 + nr), in that order: lane = step.  ss: the (up to four) rows of this lane's step, a byte each,
// idle slots = HW_DUMMY.  Returns the number of steps, -1 when they do not fit LVW_MAX_STEPS.
AGX_DEV int lvw_schedule(const Ctx& c, int lane, int r0, int nr, uint32_t& ss) {
  // the slot masks of the rows r0 + lane and r0 + 64 + lane, read once (lane = row); the loop below takes a row's from its lane
  uint32_t a0 = 0u, a1 = 0u, a2 = 0u, b0 = 0u, b1 = 0u, b2 = 0u;
  if (lane < nr) { const int* Xi = (const int*)hx_row(c.H, r0 + lane); a0 = (uint32_t)Xi[H_MLO]; a1 = (uint32_t)Xi[H_MHI]; a2 = (uint32_t)Xi[H_M2]; }
  if (64 + lane < nr) { const int* Xi = (const int*)hx_row(c.H, r0 + 64 + lane); b0 = (uint32_t)Xi[H_MLO]; b1 = (uint32_t)Xi[H_MHI]; b2 = (uint32_t)Xi[H_M2]; }
  uint32_t u0 = 0u, u1 = 0u, u2 = 0u; int fill = 0; ss = LVW_IDLE;
  bool fits = true;
  auto place = [&](uint32_t m0, uint32_t m1, uint32_t m2, int r) {
    const uint64_t conflicts = wave_ballot(((u0 & m0) | (u1 & m1) | (u2 & m2)) != 0u);         // steps that hold a row sharing a slot with r
    const int e = conflicts ? 64 - clz64(conflicts) : 0;                                         // r goes behind the last of them
    const uint64_t room = wave_ballot(fill < LVW_NG && lane < LVW_MAX_STEPS);
    const uint64_t cand = e >= 64 ? 0ull : (room >> e) << e;
    if (!cand) fits = false;                                                                     // (wave uniform; the rows that follow are placed for nothing)
    const int s = cand ? ffs64(cand) : -1;
    if (lane == s) { const int sh = 8 * fill; ss = (ss & ~(0xffu << sh)) | ((uint32_t)r << sh); fill++; u0 |= m0; u1 |= m1; u2 |= m2; }
  };
  const int n0 = nr < 64 ? nr : 64;
  _Pragma("nounroll") for (int r = 0; r < n0; r++) place((uint32_t)wave_bcast_i((int)a0, r), (uint32_t)wave_bcast_i((int)a1, r), (uint32_t)wave_bcast_i((int)a2, r), r0 + r);
  _Pragma("nounroll") for (int r = 64; r < nr; r++) place((uint32_t)wave_bcast_i((int)b0, r - 64), (uint32_t)wave_bcast_i((int)b1, r - 64), (uint32_t)wave_bcast_i((int)b2, r - 64), r0 + r);
  if (!fits) return -1;
  const uint64_t used = wave_ballot(fill > 0);
  return used ? 64 - clz64(used) : 0;
}
AGX_DEV int lvw_row(uint32_t ss, int j) { return (int)((ss >> (8 * j)) & 255u); }
// bits of the slots of `ss` that hold a row
AGX_DEV int lvw_rows4(uint32_t ss) { int m = 0; for (int j = 0; j < LVW_NG; j++) if (lvw_row(ss, j) != HW_DUMMY) m |= 1 << j; return m; }
// the step list of a part for one sweep: the steps of `ss` (lane = step) with an active row (act4: bit j = the row in byte j is visited), in order,
// rows that are not visited replaced by the idle row; four idle steps behind the last one.  Returns the number of steps.
AGX_DEV int lvw_compact(float* lds, int list, int lane, uint32_t ss, int act4) {
  uint32_t w = ss;
  for (int j = 0; j < LVW_NG; j++) if (!((act4 >> j) & 1)) w = (w & ~(0xffu << (8 * j))) | ((uint32_t)HW_DUMMY << (8 * j));
  const uint64_t m = wave_ballot(act4 != 0);
  const int n = popc64(m);
  int* L = (int*)lds + list;
  if (act4 != 0) L[wave_rank(m)] = (int)w;
  if (lane < 4) L[n + lane] = (int)LVW_IDLE;
  wave_fence();
  return n;
}

#if !defined(__HIP_DEVICE_COMPILE__) || defined(AGX_PGS_LV_CPP)
// One step, the C++ statement of what the assembly loop does (what the emulator runs).
AGX_DEV void lvw_step(const LvwLay& Y, int lane, uint32_t w, bool fric) {
  const int g = lane >> 4, k = lane & (LV_G - 1), row = (int)((w >> (8 * g)) & 255u);
  const float* Q = Y.H + HQ_BASE + HQ_STRIDE * row; const int* Qi = (const int*)Q; const int* P = (const int*)(Y.H + HP_BASE + HP_STRIDE * row);
  const int off8 = P[0] & 0xffff, n = (P[0] >> 16) & 255, na = (P[0] >> 24) & 255, ab = P[1];
  const bool on = k < n;
  const int slot = 4 * k + (k < na ? (ab & 0xffff) : ((ab >> 16) & 0xffff)) - H_AB_BIAS;
  float J = 0.f, B = 0.f, v = 0.f;
  if (on) { if (off8 >= Y.far8) lv_ld2g(Y.E + (off8 >> 2) + 2 * k, J, B); else lv_ld2(Y.lds, Y.pairs_addr + off8 + 8 * k, J, B); v = lv_ld1(Y.lds, Y.dv_addr + slot); }
  const float jdv = wave_sum16(on ? J * v : 0.f);
  const float l0 = lv_ld1(Y.lds, Y.lam_addr + 4 * row);
  float lo = Q[2], hi = Q[3];
  if (fric) { hi = Q[3] * lv_ld1(Y.lds, Y.lam_addr + Qi[2]); lo = -hi; }
  const float nl = wave_clamp(l0 + (Q[1] - jdv) * Q[0], lo, hi);
  const float dl = nl - l0;
  wave_fence();                                                     // every group has gathered before any scatters (the rows of a step share no slot anyway)
  if (on) lv_st1(Y.lds, Y.dv_addr + slot, v + B * dl);
  if (k == 0) lv_st1(Y.lds, Y.lam_addr + 4 * row, nl);
  wave_fence();
}
#else
// ---- the step loop in gfx950 assembly.  A four-deep software pipeline over the step list, written out four steps long so that its register
// slots rotate without moves.  At step t (slot t & 3):
//   S1(t + 4)  the row index of this lane's group                      ds_read_u8
//   S2(t + 3)  the header words of that row, its impulse address       2 global loads (Q: one word per lane, P: two words)
//   S3(t + 1)  pair address, on-mask, velocity slot (SDWA selects on P), the pair (LDS; a step with a row beyond the window: global),
//              the row's impulse (friction: the normal impulse of its contact as well)
//   S4(t)      gather, butterfly, update (DPP selects on Q), scatter, impulse store
// Vector memory returns in order: at the top of a step everything but the two loads of the last S2 has landed (vmcnt 2) -- the pair a far
// S3 requested too, because S3 comes before S2 in program order.  LDS returns in order: the one wait of a step (lgkmcnt 1) sits behind the gather and
// the S3 reads, whose last one may stay in flight.
// Registers: a slot s = v[80 + 10 s ...]: P0, P1 | J, B | Q word | l0 | slot address | impulse address | row index | normal impulse; its
// on-mask s[52 + 2 s : 53 + 2 s]; v120..v127 temporaries; s[50:51] the caller's EXEC, s60 steps left.
#define LVW_SL0 "v80", "v81", "v[80:81]", "v82", "v83", "v[82:83]", "v84", "v85", "v86", "v87", "v88", "v89", "s[52:53]"
#define LVW_SL1 "v90", "v91", "v[90:91]", "v92", "v93", "v[92:93]", "v94", "v95", "v96", "v97", "v98", "v99", "s[54:55]"
#define LVW_SL2 "v100", "v101", "v[100:101]", "v102", "v103", "v[102:103]", "v104", "v105", "v106", "v107", "v108", "v109", "s[56:57]"
#define LVW_SL3 "v110", "v111", "v[110:111]", "v112", "v113", "v[112:113]", "v114", "v115", "v116", "v117", "v118", "v119", "s[58:59]"
#define LVW_QP(i) " quad_perm:[" #i "," #i "," #i "," #i "] row_mask:0xf bank_mask:0xf\n"
#define LVW_DPP(CTRL) "v_add_f32_dpp v120, v120, v120 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// S1, the row index of this lane's group for step t + 4: one byte of the step list in LDS.  (Tried, r06b: the list in a vector register, lane = step,
// a step's word through v_readlane + v_bfe -- one LDS instruction less, two VALU / SALU more: 588 k against 607 k env-steps/s, same box.  Not kept.)
#define LVW_S1(RID, OFF) "ds_read_u8 " RID ", %[sa] offset:" OFF "\n"
#define LVW_S1B(RID)
#define LVW_TAIL_LAST "v_add_u32_e32 %[sa], 16, %[sa]\n"
#define LVW_PRIME LVW_S1("v88", "0") LVW_S1("v98", "4") LVW_S1("v108", "8") LVW_S1("v118", "12") "s_waitcnt lgkmcnt(0)\n"
#define LVW_S2(P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) \
  "v_lshl_add_u32 v126, " RID ", 4, %[hbk]\n" \
  "v_lshlrev_b32_e32 v127, 3, " RID "\n" \
  "global_load_dword " HWA ", v126, %[hq]\n" \
  "v_lshl_add_u32 " LA ", " RID ", 2, %[lamb]\n" \
  "global_load_dwordx2 " PP ", v127, %[hp]\n"
// (the pair read comes last: a far step has none -- its pairs come from the scratch record in an out-of-line block, LVW_FAR -- and the wait behind S3
// allows one LDS read in flight; two instructions between a VALU write of VCC and the v_cndmask that reads it: gfx940 hazard)
#define LVW_S3(FRIC, LBL, P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) \
  "v_cmp_lt_u32_sdwa vcc, %[kreg], " P0 " src0_sel:DWORD src1_sel:BYTE_3\n" \
  "v_cmp_lt_u32_sdwa " SON ", %[kreg], " P0 " src0_sel:DWORD src1_sel:BYTE_2\n" \
  "ds_read_b32 " L0 ", " LA "\n" \
  FRIC("v_add_u32_dpp v125, " HWA ", %[lambv]" LVW_QP(2)) \
  FRIC("ds_read_b32 " LN ", v125\n") \
  "v_cndmask_b32_sdwa v124, " P1 ", " P1 ", vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n" \
  "v_add_u32_e32 " IA ", v124, %[k4dv]\n" \
  "v_cmp_le_u32_sdwa vcc, %[far8], " P0 " src0_sel:DWORD src1_sel:WORD_0\n" \
  "v_add_u32_sdwa v124, " P0 ", %[k8p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
  "s_cbranch_vccnz " LBL "1f\n" \
  "ds_read_b64 " JB ", v124\n" \
  LBL "2:\n"
// (AGX_LVW_FAR_ALL_LANES=1: the first version -- every lane of a far step took its pair from the scratch record, 64 loads of which two thirds fetched pairs the
// window holds, or nothing a row owns (k >= n).  Now: the lanes of rows inside the window read LDS as in a near step -- the LDS count of the step is the same --
// and only the on-lanes of the rows beyond it load from the record.  Same values either way.)
#ifndef AGX_LVW_FAR_ALL_LANES
#define AGX_LVW_FAR_ALL_LANES 0
#endif
#if AGX_LVW_FAR_ALL_LANES
#define LVW_FAR(LBL, P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) \
  LBL "1:\n" \
  "v_add_u32_sdwa v124, " P0 ", %[k8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
  "global_load_dwordx2 " JB ", v124, %[E]\n" \
  "s_branch " LBL "2b\n"
#else
#define LVW_FAR(LBL, P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) \
  LBL "1:\n" \
  "s_andn2_b64 exec, exec, vcc\n" \
  "ds_read_b64 " JB ", v124\n" \
  "s_and_b64 exec, vcc, " SON "\n" \
  "v_add_u32_sdwa v124, " P0 ", %[k8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
  "global_load_dwordx2 " JB ", v124, %[E]\n" \
  "s_mov_b64 exec, -1\n" \
  "s_branch " LBL "2b\n"
#endif
#define LVW_GATHER(P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) "ds_read_b32 v123, " IA "\n"
// The end of a step.  Default (AGX_LVW_EARLY_GATHER, round 6): the scatter, then the GATHER OF THE NEXT STEP -- it only has to follow the scatter (LDS executes a
// wave's accesses in order) -- and only then the impulse store; the loop counter is decremented in a wait state of the butterfly (nothing between there and the
// branch touches SCC).  -DAGX_LVW_EARLY_GATHER=0: the first version (impulse store, scatter, counter, branch; the gather opens the next step).
#ifndef AGX_LVW_EARLY_GATHER
#define AGX_LVW_EARLY_GATHER 1
#endif
#if AGX_LVW_EARLY_GATHER
#define LVW_END(LA, SON, IA, GNEXT, SUBEND) \
  "s_mov_b64 exec, " SON "\n" \
  "ds_write_b32 " IA ", v123\n" \
  "s_mov_b64 exec, -1\n" \
  GNEXT \
  "ds_write_b32 " LA ", v122\n" \
  SUBEND \
  "s_cbranch_scc1 9f\n"
#define LVW_TOP_GATHER(...)
#define LVW_PRIME_GATHER(...) LVS_APPLY(LVW_GATHER, __VA_ARGS__)
#define LVW_TAIL_STEP "s_sub_u32 s60, s60, 1\n"
#define LVW_SUBEND_STEP
#else
#define LVW_END(LA, SON, IA, GNEXT, SUBEND) \
  "ds_write_b32 " LA ", v122\n" \
  "s_mov_b64 exec, " SON "\n" \
  "ds_write_b32 " IA ", v123\n" \
  "s_mov_b64 exec, -1\n" \
  "s_sub_u32 s60, s60, 1\n" \
  "s_cbranch_scc1 9f\n"
#define LVW_TOP_GATHER(...) LVS_APPLY(LVW_GATHER, __VA_ARGS__)
#define LVW_PRIME_GATHER(...)
#define LVW_TAIL_STEP "s_nop 0\n"
#define LVW_SUBEND_STEP
#endif
#define LVW_SUBEND_LAST "s_sub_u32 s60, s60, 1\n"
// S4 with the S2 of step t + 3 and the S1 of step t + 4 in the wait states the butterfly needs (two between a write and a DPP read of it)
#define LVW_S4(FRIC, S2TEXT_A, S2TEXT_B, S2TEXT_C, S1TEXT, S1B, TAIL, GNEXT, SUBEND, P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) \
  "v_mul_f32_e32 v120, " J ", v123\n" \
  "v_cndmask_b32_e64 v120, 0, v120, " SON "\n" \
  S2TEXT_A \
  LVW_DPP("quad_perm:[1,0,3,2]") \
  S2TEXT_B \
  LVW_DPP("quad_perm:[2,3,0,1]") \
  S2TEXT_C S1TEXT \
  LVW_DPP("row_half_mirror") \
  "v_mov_b32_e32 v122, " L0 "\n" \
  TAIL \
  LVW_DPP("row_mirror") \
  S1B \
  "v_sub_f32_dpp v121, " HWA ", v120" LVW_QP(1) \
  "v_fmac_f32_dpp v122, " HWA ", v121" LVW_QP(0) \
  FRIC("v_mul_f32_dpp v121, " HWA ", " LN LVW_QP(3)) \
  FRIC("v_med3_f32 v122, v122, -v121, v121\n") \
  LVS_NOT_##FRIC("v_max_f32_dpp v121, " HWA ", v122" LVW_QP(2)) \
  LVS_NOT_##FRIC("v_min_f32_dpp v122, " HWA ", v121" LVW_QP(3)) \
  "v_sub_f32_e32 v121, v122, " L0 "\n" \
  "v_fmac_f32_e32 v123, " B ", v121\n" \
  LVW_END(LA, SON, IA, GNEXT, SUBEND)
#define LVW_S2_A(P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) "v_lshl_add_u32 v126, " RID ", 4, %[hbk]\n" "v_lshlrev_b32_e32 v127, 3, " RID "\n"
#define LVW_S2_B(P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) "global_load_dword " HWA ", v126, %[hq]\n" "v_lshl_add_u32 " LA ", " RID ", 2, %[lamb]\n"
#define LVW_S2_C(P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) "global_load_dwordx2 " PP ", v127, %[hp]\n"
#define LVW_RID(P0, P1, PP, J, B, JB, HWA, L0, IA, LA, RID, LN, SON) RID
// one step: C the slot of step t, N1 of t + 1, N3 of t + 3 (S1 of t + 4 re-uses C's row index register: its step has long passed S2)
#define LVW_ITER(FRIC, LBL, OFF, TAIL, SUBEND, C, N1, N3) \
  LVW_TOP_GATHER(C) \
  "s_waitcnt vmcnt(2)\n" \
  LVS_APPLY(LVW_S3, FRIC, LBL, N1) \
  "s_waitcnt lgkmcnt(1)\n" \
  LVS_APPLY(LVW_S4, FRIC, LVS_APPLY(LVW_S2_A, N3), LVS_APPLY(LVW_S2_B, N3), LVS_APPLY(LVW_S2_C, N3), LVW_S1(LVS_APPLY(LVW_RID, C), OFF), LVW_S1B(LVS_APPLY(LVW_RID, C)), TAIL, LVW_GNEXT(N1), SUBEND, C)
#if AGX_LVW_EARLY_GATHER
#define LVW_GNEXT(...) LVS_APPLY(LVW_GATHER, __VA_ARGS__)
#else
#define LVW_GNEXT(...)
#endif
#define LVW_BODY(FRIC) \
    "s_mov_b64 s[50:51], exec\n" \
    "s_mov_b64 exec, -1\n" \
    "s_mov_b32 s60, %[nst1]\n" \
    /* prime: row indices of steps 0..3, headers of 0..2, the S3 of step 0 before the S2 of step 2 (so that a far pair of step 0 is older than the two loads the first vmcnt(2) leaves in flight) */ \
    LVW_PRIME \
    LVS_APPLY(LVW_S2, LVW_SL0) LVS_APPLY(LVW_S2, LVW_SL1) \
    "s_waitcnt vmcnt(2)\n" \
    LVS_APPLY(LVW_S3, FRIC, "7", LVW_SL0) \
    LVW_PRIME_GATHER(LVW_SL0) \
    LVS_APPLY(LVW_S2, LVW_SL2) \
    "8:\n" \
    LVW_ITER(FRIC, "1", "16", LVW_TAIL_STEP, LVW_SUBEND_STEP, LVW_SL0, LVW_SL1, LVW_SL3) \
    LVW_ITER(FRIC, "2", "20", LVW_TAIL_STEP, LVW_SUBEND_STEP, LVW_SL1, LVW_SL2, LVW_SL0) \
    LVW_ITER(FRIC, "3", "24", LVW_TAIL_STEP, LVW_SUBEND_STEP, LVW_SL2, LVW_SL3, LVW_SL1) \
    LVW_ITER(FRIC, "4", "28", LVW_TAIL_LAST, LVW_SUBEND_LAST, LVW_SL3, LVW_SL0, LVW_SL2) \
    "s_branch 8b\n" \
    "9:\n" \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
    "s_mov_b64 exec, s[50:51]\n" \
    "s_branch 99f\n" \
    LVS_APPLY(LVW_FAR, "7", LVW_SL0) LVS_APPLY(LVW_FAR, "1", LVW_SL1) LVS_APPLY(LVW_FAR, "2", LVW_SL2) LVS_APPLY(LVW_FAR, "3", LVW_SL3) LVS_APPLY(LVW_FAR, "4", LVW_SL0) \
    "99:\n"
#define LVW_CLOBBERS \
      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
      "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", \
      "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", \
      "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "vcc", "scc", "memory"
#define LVW_ASM(FRIC) \
  asm volatile(LVW_BODY(FRIC) \
    : [sa] "+v"(sa) \
    : [nst1] "s"(nsteps - 1), [hq] "s"(Hq), [hp] "s"(Hp), [E] "s"(Y.E), [far8] "s"(Y.far8), [lamb] "s"(Y.lam_addr), [lambv] "v"(Y.lam_addr), \
      [kreg] "v"(k), [k8p] "v"(8 * k + Y.pairs_addr), [k8] "v"(8 * k), [k4dv] "v"(4 * k + Y.dv_addr - H_AB_BIAS), [hbk] "v"(4 * (lane & 3)) \
    : LVW_CLOBBERS)
AGX_DEV void lvw_part_asm(const LvwLay& Y, int lane, int list_addr, int nsteps, bool fric) {
  const int k = lane & (LV_G - 1);
  int sa = list_addr + (lane >> 4);                                 // this lane's byte of step 0
  const float* Hq = Y.H + HQ_BASE; const float* Hp = Y.H + HP_BASE;
  if (fric) LVW_ASM(LVS_YES); else LVW_ASM(LVS_NO);
}
#endif

// the steps of a list, in order
AGX_DEV void lvw_part(const LvwLay& Y, int lane, int list, int nsteps, bool fric) {
  if (nsteps <= 0) return;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_LV_CPP)
  lvw_part_asm(Y, lane, lv_addr(Y.lds, Y.lds + list), nsteps, fric);
#else
  for (int t = 0; t < nsteps; t++) lvw_step(Y, lane, (uint32_t)((const int*)Y.lds)[list + t], fric);
#endif
  wave_fence();
}

// Returns false (nothing touched but LDS) when a part of this substep does not fit the scheduler: the caller takes the narrow sweep.
AGX_DEV bool pgs_lvw(Ctx& c, float* lds, int lds_words, float& dv0, float& dv1) {
  const int lane = c.lane, iters = (int)PRM(c, AGX_P_NITER);
  const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc, R = c.nrows;       // rows: [0,nnc) non-contact, [nnc,nA) normals, then nc friction rows per direction
  const bool two_dirs = R > nA + nc;
  // ---- the static schedules: non-contact + normal rows; friction rows (the second direction's rows have the slots of the first's)
  uint32_t ssA, ssF;
  const int nsA = lvw_schedule(c, lane, 0, nA, ssA), nsF = nsA < 0 ? -1 : lvw_schedule(c, lane, nA, nc, ssF);
  if (nsA < 0 || nsF < 0) return false;
  LvwLay Y; Y.lds = lds; Y.H = c.H; Y.E = c.E; Y.dv_addr = lv_addr(lds, lds + LVW_DV); Y.lam_addr = lv_addr(lds, lds + LVW_LAM); Y.pairs_addr = lv_addr(lds, lds + LVW_PAIRS);
  const int LIST_FULL = LVW_LISTS, LIST_RED = LVW_LISTS + LVW_LIST, LIST_F = LVW_LISTS + 2 * LVW_LIST;
  float* LAM = lds + LVW_LAM;
  // ---- prologue: velocity deltas, impulses, the window of pairs, the first pair offset beyond it
  const int win = lvw_window(lds_words);
  lds[LVW_DV + lane] = 0.f; lds[LVW_DV + 64 + lane] = 0.f;
  for (int r = lane; r < HW_ROWS; r += 64) LAM[r] = 0.f;
  { const f2* src = (const f2*)c.E; f2* dst = (f2*)(lds + LVW_PAIRS); const int np = c.nent < win ? c.nent : win; for (int q = lane; q < np; q += 64) dst[q] = src[q]; }
  { int far_first = 0x2000;                                          // (pair offsets are below 2^11: 8 x this never matches)
    for (int r = lane; r < R; r += 64) { const int p0 = ((const int*)(c.H + HP_BASE))[HP_STRIDE * r]; const int off = (p0 & 0xffff) >> 3, n = (p0 >> 16) & 255; if (off + n > win && off < far_first) far_first = off; }
    Y.far8 = 8 * (int)wave_min((float)far_first); }
  wave_sync();
  const int rowsA4 = lane < nsA ? lvw_rows4(ssA) : 0, rowsF4 = lane < nsF ? lvw_rows4(ssF) : 0;
  const int nFull = lvw_compact(lds, LIST_FULL, lane, ssA, rowsA4);
  int nRed = 0, nF = 0, actF = -1;
  int stat_steps = 0, stat_rows = 0;                                // debug launches only: steps executed, rows visited (tools/gpu_solve_streams.py)
  const bool stats = c.dbg != nullptr;
  const int K = noop_period(c);                                     // the no-op re-test rule: see pgs()
  _Pragma("nounroll") for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0, use = K > 0 && !retest;
    float bef[LVW_NG];
    if (retest) for (int j = 0; j < LVW_NG; j++) bef[j] = LAM[lvw_row(ssA, j)];
    lvw_part(Y, lane, use ? LIST_RED : LIST_FULL, use ? nRed : nFull, false);          // (ONE call site per kind of part: the loop is ~290 instructions)
    if (stats) { stat_steps += use ? nRed : nFull; const int* L = (const int*)lds + (use ? LIST_RED : LIST_FULL); const int nn = use ? nRed : nFull;
      stat_rows += wave_sum_i(lane < nn ? lvw_rows4((uint32_t)L[lane < nn ? lane : 0]) == 0 ? 0 : __builtin_popcount(lvw_rows4((uint32_t)L[lane < nn ? lane : 0])) : 0); }
    if (retest) {                                                   // rows whose visit changed nothing are left out until the next re-test
      int act = 0;
      for (int j = 0; j < LVW_NG; j++) if (LAM[lvw_row(ssA, j)] != bef[j]) act |= 1 << j;
      nRed = lvw_compact(lds, LIST_RED, lane, ssA, act & rowsA4);
    }
    _Pragma("nounroll") for (int dir = 0; dir < (two_dirs ? 2 : 1); dir++) {
      // friction rows: bounds from the normal impulses as this sweep's normal pass left them (the loop reads them: Q word 2); a row whose
      // normal impulse and own impulse are both zero is an exact no-op and is not visited
      const uint32_t ss = ssF;
      int act = 0;
      for (int j = 0; j < LVW_NG; j++) {
        const int r = lvw_row(ss, j);
        if (r != HW_DUMMY) { const float lf = LAM[r + dir * nc], ln = LAM[r - nc]; if (ln != 0.f || lf != 0.f) act |= 1 << j; }
      }
      uint32_t ssd = ss;
      if (dir) for (int j = 0; j < LVW_NG; j++) if (lvw_row(ss, j) != HW_DUMMY) ssd += (uint32_t)nc << (8 * j);      // the same steps, the rows of the second direction
      act &= rowsF4;
      if (two_dirs || wave_any(act != actF)) { nF = lvw_compact(lds, LIST_F, lane, ssd, act); actF = act; }      // (the list of the last sweep serves while the same rows are due)
      lvw_part(Y, lane, LIST_F, nF, true);
      if (stats) { stat_steps += nF; stat_rows += wave_sum_i(__builtin_popcount(act)); }
    }
  }
  wave_sync();
  // velocity deltas back to their DoF lanes; solved normal impulses -> contact records (what getContactPoints reports until the next step)
  dv0 = lds[LVW_DV + lane]; dv1 = lds[LVW_DV + 64 + lane];
  if (lane < nc) c.gcon[CON_STRIDE * lane + C_LAM] = LAM[nnc + lane];
  if (stats && lane == 0) { c.dbg[DBG_TIME + 16] = (float)stat_steps; c.dbg[DBG_TIME + 17] = (float)stat_rows; c.dbg[DBG_TIME + 18] = (float)(nsA + nsF); c.dbg[DBG_TIME + 19] = (float)R; }
  wave_sync();
  return true;
}

}  // namespace agx
