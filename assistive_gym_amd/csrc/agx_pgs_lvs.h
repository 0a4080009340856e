// agx_pgs_lvs.h -- K6, the row-local sweep with nothing but velocities and pairs in LDS (AGX_PGS_LV == 3, the default of the feeding variant).
// Part of the stepper (see agx_step.h); included by agx_step.h only, after agx_pgs_lv.h (whose visit it restates on a leaner layout).
//
// agx_pgs_lv.h keeps an 8-word header per row and a 16-bit velocity slot per pair in LDS beside the pairs: 17.6 KB for an ordinary FeedingJaco
// substep (1,350 pairs, 123 rows), i.e. 8 solve waves per CU.  Here a visit gets
//   * the row's header -- 1/D, b, lo, hi, pair offset, pair counts, velocity slot offsets: the 32-byte row of the first header table build_rows()
//     leaves in the scratch record -- through the SCALAR cache: one s_load_dwordx8 three visits ahead, the values are used straight from
//     scalar registers;
//   * the row's impulse from a vector register (lane = row: v_readlane before, a one-lane v_mov after), so the no-op re-test and the
//     friction bounds are ordinary per-lane arithmetic between the parts;
//   * the velocity slot of a pair by arithmetic on one header word;
//   * its pairs from the LDS window, or -- the rows beyond it: the tail of the friction rows, most of which the no-op rule skips -- from the
//     scratch record (global_load, vmcnt).
// LDS holds the velocity deltas (128 words) and the window: 10 KB per solve wave (16 per CU), three LDS instructions per visit instead of
// seven.  Same rows, same order, same clamps, same arithmetic as pgs_lv(): the two are BIT-IDENTICAL on the GPU (tests/test_gpu_solve_variants.py,
// tools/gpu_lv_bits.py, profiles/r05/r05i_bits_*) and in the emulator, whatever the window.
//
// MEASURED (round 5, same box, 4096 FeedingJaco environments, 300 steps; profiles/r05/r05h..r05s).  env-steps/s: pgs_lv() at 20 KB 522 k;
// this file at 9.5 / 10 / 11 / 12 KB of LDS 561 / 566 / 555 / 547 k (with the headers as 64-byte rows of one table: 544 / 552 / 546 / 540 k --
// the 32-byte table halves the lines a sweep pulls through the scalar cache and lets the live headers of an XCD fit its L2: HBM-side
// traffic of a solve launch 70 -> 31 KB per environment).  Shader cycles per row and sweep with one / sixteen waves per CU: 159 / 213 at
// 10 KB (64-byte rows: 159 / 266; pgs_lv: 155 / 168 with eight).  What did NOT help: impulses and friction bounds in LDS arrays instead of the register
// (three fewer vector instructions, three more LDS instructions: 521 k; tools/experiments/agx_pgs_lvs_impulses_in_lds.h); the write-back of
// the impulse and the loop test moved into the shadow of the next gather (546 k).  What did: the scalar instructions of the header request
// in the wait states the DPP butterfly needs anyway (538 -> 552 k, measured on the 64-byte rows like the probes below); rows beyond the
// window waiting for their OWN pair only (vmcnt 1): 566 -> 569 k, and the 9.5 KB launch as fast as the 10 KB one.  Warming the scalar cache
// with the next part's first header lines at the start of a part: no difference (568 / 569 k, r05u_*).
// Marginal cost of ONE more instruction per visit, measured with redundant instructions (AGX_LVS_PROBE_*, r05m_*), one / sixteen waves
// per CU: vector 5.9 / 3.8 cycles, scalar 5.5 / 5.9, s_nop 5.5 / 4.2, LDS read 21 / 6, LDS write 11 / 9 -- of a visit of 268 / 448 cycles and
// 38 instructions.  A wave pays 4..6 cycles for every instruction it issues, of whatever kind, in both regimes: the sweep is bound by the
// per-wave issue rate of a dependent chain, not by a pipe; what is left is the instruction count per visit.
#pragma once

namespace agx {

constexpr bool LVS_COMPILED = (AGX_PGS_LV == 3 || AGX_PGS_LV == 4) && LV_COMPILED;      // (4: the wide sweep of agx_pgs_lvw.h, with this one as its fallback)
constexpr int LVS_SOLVE_LDS_BYTES = 10240;                          // LDS of a solve launch of that variant: 16 waves per CU; the window (1,216 pairs) holds the non-contact and normal rows of an ordinary substep and most friction rows
constexpr int LVS_DV = 0, LVS_PAIRS = 128;                           // LDS words: dv[128], pairs[2 x window]
static_assert(!LVS_COMPILED || HDR_STRIDE == 8, "the row-local sweep reads 32-byte headers");
static_assert(!HDR_WIDE || H_INVD == 0 && H_B == 1 && H_LO == 2 && H_HI == 3 && H_OFF == 4 && H_N == 5 && H_NA == 6 && H_AB == 7, "the scalar load of a visit is the 8 words of a row of the first header table");

//  The look-ahead of the last visit of a part runs on the "exhausted cursor" header -- row base - 1; for base 0 the last 32 bytes of the pair
// arena, arbitrary floats read as a pair offset.  On the FAR path that offset would be DEREFERENCED (a global load up to 4 GB behind the scratch
// record).  It cannot happen while rows 0..63 -- the only ones a part with base 0 visits -- lie inside the LDS window whatever the launch size:
// 64 rows x 16 pairs + the zero pair <= the smallest window a solve launch can have.  Device builds with a capped window (AGX_LV_WINDOW_CAP) are refused.
#if defined(__HIP_DEVICE_COMPILE__) && defined(AGX_LV_WINDOW_CAP) && !defined(AGX_PGS_LV_CPP)
#error "AGX_LV_WINDOW_CAP is for the C++ twin (emulator, -DAGX_PGS_LV_CPP): the assembly sweep's look-ahead relies on rows 0..63 lying inside the window"
#endif
static_assert(!LVS_COMPILED || (LDS_SOLVE_BYTES / 4 - LVS_PAIRS) / 2 >= 64 * LV_G + 1, "rows 0..63 must fit the smallest LDS window of a solve launch (look-ahead of the exhausted cursor)");
// pairs that fit a solve launch with lds_words of LDS
AGX_DEV int lvs_window(int lds_words) {
  int w = (lds_words - LVS_PAIRS) / 2;
#ifdef AGX_LV_WINDOW_CAP            // tests: a small window, so that most rows of an ordinary scene read their pairs from the scratch record
  if (w > AGX_LV_WINDOW_CAP) w = AGX_LV_WINDOW_CAP;
#endif
  return w;
}
AGX_DEV bool lvs_eligible(const Ctx& c, int lds_words) {
  (void)lds_words;
  return lv_eligible(c) && c.first_normal + c.ncon <= 128 && c.ncon <= 64;
}

struct LvsLay { float* lds; const float* H; const float* E; int dv_addr, pairs_addr, rfar; };   // rfar: first row whose pairs are not (all) inside the LDS window

#if !defined(__HIP_DEVICE_COMPILE__) || defined(AGX_PGS_LV_CPP)
// One visit, the C++ statement of what the assembly loop does: what the emulator runs (tests/test_emu_parity.py holds it bit-identical with
// pgs_lv()'s twin).  -DAGX_PGS_LV_CPP compiles it for the device as well (a debugging aid, never run on hardware).
// lam: this lane's impulse register (lane = row - base); hiv: friction parts, mu x the normal impulse of the lane's contact (else unused).
AGX_DEV void lvs_visit(const LvsLay& Y, int lane, int base, int bit, float& lam, bool fric, bool far, float hiv) {
  const float* H = Y.H + HDR_STRIDE * (base + bit); const int* Hi = (const int*)H;
  const int k = lane & (LV_G - 1);                                  // (the four 16-lane groups of the wave do the same visit)
  const int n = Hi[H_N], na = Hi[H_NA], ab = Hi[H_AB], off = Hi[H_OFF] & 0x0fffffff;
  const bool on = k < n;
  const int slot = 4 * k + ((ab >> (k < na ? 0 : 16)) & 1023) - H_AB_BIAS;
  float J = 0.f, B = 0.f, v = 0.f;
  if (on) { if (far) lv_ld2g(Y.E + 2 * (off + k), J, B); else lv_ld2(Y.lds, Y.pairs_addr + 8 * (off + k), J, B); v = lv_ld1(Y.lds, Y.dv_addr + slot); }
  const float jdv = wave_sum16(on ? J * v : 0.f);
  const float l0 = wave_bcast(lam, bit);
  float lo = H[H_LO], hi = H[H_HI];
  if (fric) { hi = wave_bcast(hiv, bit); lo = -hi; }
  const float nl = wave_clamp(l0 + (H[H_B] - jdv) * H[H_INVD], lo, hi);
  const float dl = nl - l0;
  if (lane == bit) lam = nl;
  wave_fence();                                                     // (emulator: the four groups have gathered before any of them scatters)
  if (on) lv_st1(Y.lds, Y.dv_addr + slot, v + B * dl);
  wave_fence();                                                     // the next visit gathers what this one scattered
}
#else
// ---- the visit loop in gfx950 assembly.  One 64-bit mask of rows (base + bit), at least one.  Header ring of four scalar octets (visit t,
// t + 1, t + 2 in registers, t + 3 requested), entry ring of two; written out four visits long so that both rings rotate without moves.
// Scalar loads return out of order with respect to LDS traffic, so the one wait of a visit is lgkmcnt(0): it sits after the gather and the
// next visit's pair load are issued and BEFORE the header request of the visit, which then has a whole visit to come back.
// A header: s[+0] 1/D, +1 b, +2 lo, +3 hi, +4 pair offset, +5 pairs, +6 pairs of the first range, +7 slot offsets; its bit index beside it.
// Registers: s[50:51] the caller's EXEC (the visits run under lanes 0..15; the impulse of the visited row is written under EXEC = its lane),
// s[52:83] headers, s84..s87 their bit indices, s[88:89] cursor, s[90:93] on-masks, s94..s97 impulses / friction bounds,
// s98 visits left, s99 scratch, vcc; v88..v93 entries (pair, slot address), v94..v99 temporaries.
#define LVS_DPP(CTRL) "v_add_f32_dpp v95, v95, v95 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// AGX_LVS_PROBE_*: marginal cost of one more instruction of a kind inside a visit -- redundant instructions that change no result
// (tools/gpu_r05_s13.sh; profiles/r05/r05m_*): n extra instructions per visit
#define LVS_REP0(x)
#define LVS_REP1(x) x
#define LVS_REP2(x) x x
#define LVS_REP4(x) x x x x
#define LVS_REPN(n, x) LVS_REPN_(n, x)
#define LVS_REPN_(n, x) LVS_REP##n(x)
#ifndef AGX_LVS_PROBE_VALU
#define AGX_LVS_PROBE_VALU 0
#endif
#ifndef AGX_LVS_PROBE_SALU
#define AGX_LVS_PROBE_SALU 0
#endif
#ifndef AGX_LVS_PROBE_NOP
#define AGX_LVS_PROBE_NOP 0
#endif
#ifndef AGX_LVS_PROBE_LDSR
#define AGX_LVS_PROBE_LDSR 0
#endif
#ifndef AGX_LVS_PROBE_LDSW
#define AGX_LVS_PROBE_LDSW 0
#endif
#define LVS_YES(x) x
#define LVS_NO(x)
#define LVS_NOT_LVS_YES(x)
#define LVS_NOT_LVS_NO(x) x
#define LVS_ENTRY(FRIC, FAR, N_OFF, N_N, N_NA, N_AB, N_BIT, EN_JB, EN_IA, EN_ON, EN_ONLO, EN_LAM, EN_HI) \
  "s_bfm_b32 " EN_ONLO ", " N_N ", 0\n" \
  "s_bfm_b32 vcc_lo, " N_NA ", 0\n" \
  "v_lshl_add_u32 v98, " N_OFF ", 3, %[k8p]\n" \
  "v_cndmask_b32_e64 v99, 16, 0, vcc\n" \
  FAR("global_load_dwordx2 " EN_JB ", v98, %[E]\n") LVS_NOT_##FAR("ds_read_b64 " EN_JB ", v98\n") \
  "v_bfe_u32 v99, " N_AB ", v99, 10\n" \
  "v_add_u32_e32 " EN_IA ", v99, %[k4dv]\n" \
  "v_readlane_b32 " EN_LAM ", %[lam], " N_BIT "\n" \
  FRIC("v_readlane_b32 " EN_HI ", %[hiv], " N_BIT "\n")
#define LVS_HEADER(N_OCT, N_BIT) \
  "s_ff1_i32_b64 " N_BIT ", s[88:89]\n" \
  "s_bitset0_b64 s[88:89], " N_BIT "\n" \
  "s_lshl_b32 s99, " N_BIT ", 5\n" \
  "s_add_u32 s99, s99, %[base64]\n" \
  "s_load_dwordx8 " N_OCT ", %[Hm], s99\n"
// the wait of a visit: its gather (LDS), the header requested a visit ago (scalar), and -- rows beyond the window -- its OWN pair, requested a visit
// ago: vector loads return in order, so the pair of the next visit, requested a moment ago, may stay in flight (vmcnt 1)
#define LVS_WAIT(FAR) FAR("s_waitcnt vmcnt(1) lgkmcnt(0)\n") LVS_NOT_##FAR("s_waitcnt lgkmcnt(0)\n")
#define LVS_WAIT_ALL(FAR) FAR("s_waitcnt vmcnt(0) lgkmcnt(0)\n") LVS_NOT_##FAR("s_waitcnt lgkmcnt(0)\n")
#define LVS_STEP(FRIC, FAR, C_INVD, C_B, C_LO, C_HI, C_BIT, N1_OFF, N1_N, N1_NA, N1_AB, N1_BIT, N3_OCT, N3_BIT, EC_J, EC_B, EC_IA, EC_ON, EC_LAM, EC_HI, EN_JB, EN_IA, EN_ON, EN_ONLO, EN_LAM, EN_HI) \
  "ds_read_b32 v94, " EC_IA "\n" \
  LVS_ENTRY(FRIC, FAR, N1_OFF, N1_N, N1_NA, N1_AB, N1_BIT, EN_JB, EN_IA, EN_ON, EN_ONLO, EN_LAM, EN_HI) \
  LVS_REPN(AGX_LVS_PROBE_LDSR, "ds_read_b32 v97, " EC_IA "\n") \
  LVS_WAIT(FAR) \
  LVS_REPN(AGX_LVS_PROBE_VALU, "v_mov_b32_e32 v97, 0\n") LVS_REPN(AGX_LVS_PROBE_SALU, "s_mov_b32 s99, 0\n") LVS_REPN(AGX_LVS_PROBE_NOP, "s_nop 0\n") \
  "v_mul_f32_e32 v95, " EC_J ", v94\n" \
  "v_cndmask_b32_e64 v95, 0, v95, " EC_ON "\n" \
  /* the header request of visit t + 3 and two moves fill the wait states a DPP read of a fresh register needs (2 each) */ \
  "s_ff1_i32_b64 " N3_BIT ", s[88:89]\n" \
  "s_bitset0_b64 s[88:89], " N3_BIT "\n" \
  LVS_DPP("quad_perm:[1,0,3,2]") \
  "s_lshl_b32 s99, " N3_BIT ", 5\n" \
  "s_add_u32 s99, s99, %[base64]\n" \
  LVS_DPP("quad_perm:[2,3,0,1]") \
  "s_load_dwordx8 " N3_OCT ", %[Hm], s99\n" \
  "v_mov_b32_e32 v99, " EC_LAM "\n" \
  LVS_DPP("row_half_mirror") \
  FRIC("v_mov_b32_e32 v98, " EC_HI "\n") LVS_NOT_##FRIC("v_mov_b32_e32 v98, " C_LO "\n") \
  "s_nop 0\n" \
  LVS_DPP("row_mirror") \
  "v_sub_f32_e32 v96, " C_B ", v95\n" \
  "v_fma_f32 v97, " C_INVD ", v96, v99\n" \
  FRIC("v_med3_f32 v96, v97, -v98, v98\n") LVS_NOT_##FRIC("v_med3_f32 v96, v97, v98, " C_HI "\n") \
  "v_sub_f32_e32 v97, v96, v99\n" \
  "v_fmac_f32_e32 v94, " EC_B ", v97\n" \
  "s_mov_b64 exec, " EC_ON "\n" \
  "ds_write_b32 " EC_IA ", v94\n" \
  LVS_REPN(AGX_LVS_PROBE_LDSW, "ds_write_b32 " EC_IA ", v94\n") \
  "v_readfirstlane_b32 s99, v96\n" \
  "s_lshl_b64 exec, 1, " C_BIT "\n" \
  "v_mov_b32_e32 %[lam], s99\n" \
  "s_mov_b64 exec, 0xffff\n" \
  "s_sub_u32 s98, s98, 1\n" \
  "s_cbranch_scc1 9f\n"
#define LVS_C_A "s52", "s53", "s54", "s55", "s84"
#define LVS_C_B "s60", "s61", "s62", "s63", "s85"
#define LVS_C_C "s68", "s69", "s70", "s71", "s86"
#define LVS_C_D "s76", "s77", "s78", "s79", "s87"
#define LVS_N1_A "s56", "s57", "s58", "s59", "s84"
#define LVS_N1_B "s64", "s65", "s66", "s67", "s85"
#define LVS_N1_C "s72", "s73", "s74", "s75", "s86"
#define LVS_N1_D "s80", "s81", "s82", "s83", "s87"
#define LVS_N3_A "s[52:59]", "s84"
#define LVS_N3_B "s[60:67]", "s85"
#define LVS_N3_C "s[68:75]", "s86"
#define LVS_N3_D "s[76:83]", "s87"
// entry slots: J, B, slot address, on-mask, impulse, friction bound / as targets: pair, slot address, on-mask (pair, low word), impulse, bound
#define LVS_EC_P "v88", "v89", "v90", "s[90:91]", "s94", "s96"
#define LVS_EC_Q "v92", "v93", "v91", "s[92:93]", "s95", "s97"
#define LVS_EN_P "v[88:89]", "v90", "s[90:91]", "s90", "s94", "s96"
#define LVS_EN_Q "v[92:93]", "v91", "s[92:93]", "s92", "s95", "s97"
#define LVS_APPLY(M, ...) M(__VA_ARGS__)
#define LVS_CALL(FRIC, FAR, C, N1, N3, EC, EN) LVS_APPLY(LVS_STEP, FRIC, FAR, C, N1, N3, EC, EN)
#define LVS_BODY(FRIC, FAR) \
    "s_mov_b64 s[50:51], exec\n" \
    "s_mov_b64 exec, 0xffff\n" \
    "s_mov_b64 s[88:89], %[mask]\n" \
    "s_mov_b32 s98, %[nvis1]\n" \
    "s_mov_b32 s91, 0\n" "s_mov_b32 s93, 0\n" "s_mov_b32 vcc_hi, 0\n" \
    /* prime: headers of visits 0, 1, 2; entry of visit 0 */ \
    LVS_HEADER("s[52:59]", "s84") LVS_HEADER("s[60:67]", "s85") LVS_HEADER("s[68:75]", "s86") \
    "s_waitcnt lgkmcnt(0)\n" \
    LVS_APPLY(LVS_ENTRY, FRIC, FAR, LVS_N1_A, LVS_EN_P) \
    "8:\n" \
    LVS_CALL(FRIC, FAR, LVS_C_A, LVS_N1_B, LVS_N3_D, LVS_EC_P, LVS_EN_Q) \
    LVS_CALL(FRIC, FAR, LVS_C_B, LVS_N1_C, LVS_N3_A, LVS_EC_Q, LVS_EN_P) \
    LVS_CALL(FRIC, FAR, LVS_C_C, LVS_N1_D, LVS_N3_B, LVS_EC_P, LVS_EN_Q) \
    LVS_CALL(FRIC, FAR, LVS_C_D, LVS_N1_A, LVS_N3_C, LVS_EC_Q, LVS_EN_P) \
    "s_branch 8b\n" \
    "9:\n" \
    LVS_WAIT_ALL(FAR) \
    "s_mov_b64 exec, s[50:51]\n"
#define LVS_CLOBBERS \
      "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
      "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", \
      "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc", "memory"
// an exhausted cursor gives bit -1: the header address is then base64 - 32 with base64 = 32 (base + 1) against Hm = H - 32 bytes, i.e. the row
// before `base` (or, for base 0, the last 32 bytes of the pair arena in front of the headers): loaded, never visited
#define LVS_ASM(FRIC, FAR, K8) \
  asm volatile(LVS_BODY(FRIC, FAR) \
    : [lam] "+v"(lam) \
    : [mask] "s"(mask), [nvis1] "s"(nvis1), [base64] "s"(base64), [Hm] "s"(Hm), [E] "s"(Y.E), [k8p] "v"(K8), [k4dv] "v"(4 * lane + Y.dv_addr - H_AB_BIAS), [hiv] "v"(hiv) \
    : LVS_CLOBBERS)
// far: the rows' pairs lie beyond the LDS window: loaded from the scratch record (vmcnt) instead
AGX_DEV void lvs_part_asm(const LvsLay& Y, int lane, uint64_t mask, int base, float& lam, bool fric, bool far, float hiv) {
  const int nvis1 = popc64(mask) - 1, base64 = 4 * HDR_STRIDE * (base + 1);
  const float* Hm = Y.H - HDR_STRIDE;
  if (!far) { if (fric) LVS_ASM(LVS_YES, LVS_NO, 8 * lane + Y.pairs_addr); else LVS_ASM(LVS_NO, LVS_NO, 8 * lane + Y.pairs_addr); }
  else { if (fric) LVS_ASM(LVS_YES, LVS_YES, 8 * lane); else LVS_ASM(LVS_NO, LVS_YES, 8 * lane); }
}
#endif

// the rows base + (set bits of mask), ascending.  lam: impulses, lane = row - base; fric: friction rows, bounds -+ hiv (lane = row - base).
// Pair offsets grow with the row index: the rows whose pairs lie beyond the LDS window are a suffix [Y.rfar, ...), visited after the others.
AGX_DEV void lvs_part(const LvsLay& Y, int lane, uint64_t mask, int base, float& lam, bool fric, float hiv) {
  if (!mask) return;
  const int nn = Y.rfar - base;
  const uint64_t near = nn >= 64 ? mask : (nn <= 0 ? 0ull : mask & pgs_range_mask(0, nn)), far = mask & ~near;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_LV_CPP)
  // (the loop narrows EXEC to lanes 0..15 itself and restores it; the impulse register is read and written by lane index, whatever EXEC)
  if (near) lvs_part_asm(Y, lane, near, base, lam, fric, false, hiv);
  if (far) lvs_part_asm(Y, lane, far, base, lam, fric, true, hiv);
#else
  for (uint64_t m = near; m; m &= m - 1ull) lvs_visit(Y, lane, base, ffs64(m), lam, fric, false, hiv);
  for (uint64_t m = far; m; m &= m - 1ull) lvs_visit(Y, lane, base, ffs64(m), lam, fric, true, hiv);
#endif
  wave_fence();
}

AGX_DEV void pgs_lvs(Ctx& c, float* lds, int lds_words, float& dv0, float& dv1) {
  const int lane = c.lane, iters = (int)PRM(c, AGX_P_NITER);
  const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc, R = c.nrows;       // rows: [0,nnc) non-contact, [nnc,nA) normals, then nc friction rows per direction
  LvsLay Y; Y.lds = lds; Y.H = c.H; Y.E = c.E; Y.dv_addr = lv_addr(lds, lds + LVS_DV); Y.pairs_addr = lv_addr(lds, lds + LVS_PAIRS);
  // ---- prologue (all 64 lanes): velocity deltas, the window of pairs, the first row beyond it
  const int win = lvs_window(lds_words);
  lds[LVS_DV + lane] = 0.f; lds[LVS_DV + 64 + lane] = 0.f;
  { const f2* src = (const f2*)c.E; f2* dst = (f2*)(lds + LVS_PAIRS); const int np = c.nent < win ? c.nent : win; for (int q = lane; q < np; q += 64) dst[q] = src[q]; }
  { int far_first = R;
    for (int r = lane; r < R; r += 64) { const int* Hi = (const int*)(c.H + HDR_STRIDE * r); if ((Hi[H_OFF] & 0x0fffffff) + Hi[H_N] > win && r < far_first) far_first = r; }
    Y.rfar = (int)wave_min((float)far_first); }
  // friction coefficient of this lane's contact (0 for a row without effective mass: pinned at zero impulse)
  const bool two_dirs = R > nA + nc;
  float mu1 = 0.f, mu2 = 0.f;
  if (lane < nc) {
    const float* H = c.H + HDR_STRIDE * (nA + lane); mu1 = H[H_INVD] != 0.f ? hx_row(c.H, nA + lane)[H_MU] : 0.f;
    if (two_dirs) { const float* H2 = c.H + HDR_STRIDE * (nA + nc + lane); mu2 = H2[H_INVD] != 0.f ? hx_row(c.H, nA + nc + lane)[H_MU] : 0.f; }
  }
  wave_sync();
  const uint64_t rowsA0 = pgs_range_mask(0, nA < 64 ? nA : 64), rowsA1 = nA > 64 ? pgs_range_mask(0, nA - 64) : 0ull;
  const int K = noop_period(c);                                     // the no-op re-test rule: see pgs()
  // impulses: rows 0..63 and 64..127 of the non-contact + normal block (lane = row, row - 64); friction rows of either direction (lane = contact)
  float lamA0 = 0.f, lamA1 = 0.f, lamF1 = 0.f, lamF2 = 0.f;
  uint64_t skip0 = 0ull, skip1 = 0ull;
  const int nsrc = (nnc + lane) & 63; const bool nhi = nnc + lane >= 64;           // where the normal impulse of this lane's contact lives
  float ln = 0.f;
#ifdef AGX_EMU_TRACE_SCHED     // tests/diag/solve_schedule_study.py: the DoF masks of the rows, then per sweep the rows visited
  if (lane == 0) { g_sched_trace[g_sched_n++] = -1; g_sched_trace[g_sched_n++] = R; g_sched_trace[g_sched_n++] = nnc; g_sched_trace[g_sched_n++] = nc;
    for (int r = 0; r < R; r++) { const int* Xi = (const int*)hx_row(c.H, r); g_sched_trace[g_sched_n++] = Xi[H_MLO]; g_sched_trace[g_sched_n++] = Xi[H_MHI]; g_sched_trace[g_sched_n++] = Xi[H_M2]; } }
#define AGX_TRACE_SCHED(kind, m) if (lane == 0 && g_sched_n + 3 < (1 << 24)) { g_sched_trace[g_sched_n++] = kind; g_sched_trace[g_sched_n++] = (int)(uint32_t)(m); g_sched_trace[g_sched_n++] = (int)(uint32_t)((m) >> 32); }
#else
#define AGX_TRACE_SCHED(kind, m)
#endif
  for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0, use = K > 0 && !retest;
    const float bef0 = lamA0, bef1 = lamA1;
    AGX_TRACE_SCHED(-2, rowsA0 & ~(use ? skip0 : 0ull)) AGX_TRACE_SCHED(-3, rowsA1 & ~(use ? skip1 : 0ull))
    lvs_part(Y, lane, rowsA0 & ~(use ? skip0 : 0ull), 0, lamA0, false, 0.f);
    lvs_part(Y, lane, rowsA1 & ~(use ? skip1 : 0ull), 64, lamA1, false, 0.f);
    if (retest) { skip0 = wave_ballot(lamA0 == bef0); skip1 = wave_ballot(lamA1 == bef1); }
    { const float s0 = wave_shfl(lamA0, nsrc), s1 = wave_shfl(lamA1, nsrc); ln = lane < nc ? (nhi ? s1 : s0) : 0.f; }
    // friction rows (lane = contact): bounds from the normal impulses as this sweep's normal pass left them; a row whose normal impulse
    // and own impulse are both zero is an exact no-op and is not visited
    { const uint64_t todo = wave_ballot(lane < nc && (ln != 0.f || lamF1 != 0.f));
      AGX_TRACE_SCHED(-4, todo)
      lvs_part(Y, lane, todo, nA, lamF1, true, mu1 * ln); }
    if (two_dirs) {
      const uint64_t todo = wave_ballot(lane < nc && (ln != 0.f || lamF2 != 0.f));
      lvs_part(Y, lane, todo, nA + nc, lamF2, true, mu2 * ln);
    }
  }
  wave_sync();
  // velocity deltas back to their DoF lanes; solved normal impulses -> contact records (what getContactPoints reports until the next step)
  dv0 = lds[LVS_DV + lane]; dv1 = lds[LVS_DV + 64 + lane];
  if (lane < nc) c.gcon[CON_STRIDE * lane + C_LAM] = ln;
  wave_sync();
}

}  // namespace agx
