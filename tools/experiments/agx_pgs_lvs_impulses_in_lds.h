// agx_pgs_lvs.h -- K6, the row-local sweep with nothing but velocities and pairs in LDS (AGX_PGS_LV == 3).
// Part of the stepper (see agx_step.h); included by agx_step.h only, after agx_pgs_lv.h (whose visit it restates on a leaner layout).
//
// agx_pgs_lv.h keeps an 8-word header per row and a 16-bit velocity slot per pair in LDS beside the pairs: 17.6 KB for an ordinary FeedingJaco
// substep, 8 solve waves per CU, and the sweep is a dependent chain per visit (260 cycles alone on a CU, 283 with 8 waves): latency bound.
// Here a visit gets
//   * the row's header -- 1/D, b, lo, hi, pair offset, pair counts, velocity slot offsets: words 0..7 of the 64-byte header build_rows()
//     leaves in the scratch record -- through the SCALAR cache: one s_load_dwordx8 three visits ahead, the values are used straight from
//     scalar registers;
//   * the row's impulse from a vector register (lane = row: v_readlane before, v_writelane after), so the no-op re-test and the friction
//     bounds are ordinary per-lane arithmetic between the parts;
//   * the velocity slot of a pair by arithmetic on two header words.
// LDS holds the velocity deltas (128 words) and the pairs: 10.5 KB for the ordinary substep, and per visit three LDS instructions (pairs,
// gather, scatter) instead of seven.  An environment whose pairs do not fit takes the register sweep (agx_pgs.h): wave uniform.
// Same rows, same order, same clamps, same no-op re-test rule and friction skipping as pgs() and pgs_lv().
#pragma once

namespace agx {

constexpr bool LVS_COMPILED = AGX_PGS_LV == 3 && LV_COMPILED;
constexpr int LVS_SOLVE_LDS_BYTES = LDS_SOLVE_BYTES;                // LDS of a solve launch of that variant: 16 waves per CU; the window holds the non-contact and normal rows of an ordinary substep and the first friction rows
constexpr int LVS_DV = 0, LVS_LAM = 128, LVS_HI = LVS_LAM + MAX_ROWS, LVS_PAIRS = LVS_HI + MAX_ROWS;   // LDS words: dv[128], impulses[MAX_ROWS], friction bounds[MAX_ROWS], pairs[2 x window]
static_assert(LVS_PAIRS % 2 == 0, "pairs are read as 8-byte words");
static_assert(HDR_STRIDE == 16 && H_INVD == 0 && H_B == 1 && H_LO == 2 && H_HI == 3 && H_OFF == 4 && H_N == 5 && H_NA == 6 && H_AB == 7, "the scalar load of a visit is words 0..7 of a 64-byte header");

// pairs that fit a solve launch with lds_words of LDS
AGX_DEV int lvs_window(int lds_words) {
  int w = (lds_words - LVS_PAIRS) / 2;
#ifdef AGX_LV_WINDOW_CAP            // tests: a small window, so that ordinary scenes take the fallback
  if (w > AGX_LV_WINDOW_CAP) w = AGX_LV_WINDOW_CAP;
#endif
  return w;
}
AGX_DEV bool lvs_eligible(const Ctx& c, int lds_words) {
  (void)lds_words;
  return lv_eligible(c) && c.first_normal + c.ncon <= 128 && c.ncon <= 64;
}

struct LvsLay { float* lds; const float* H; const float* E; int dv_addr, lam_addr, pairs_addr, rfar; };   // rfar: first row whose pairs are not (all) inside the LDS window

#if !defined(__HIP_DEVICE_COMPILE__) || defined(AGX_PGS_LV_CPP)
// One visit, the C++ statement of what the assembly loop does (the emulator runs this; on the device it is the -DAGX_PGS_LV_CPP build).
// lam: this lane's impulse register (lane = row - base); hiv: friction parts, mu x the normal impulse of the lane's contact (else unused).
AGX_DEV void lvs_visit(const LvsLay& Y, int lane, int row, bool fric, bool far) {
  const float* H = Y.H + HDR_STRIDE * row; const int* Hi = (const int*)H;
  const int k = lane & (LV_G - 1);                                  // (the four 16-lane groups of the wave do the same visit)
  const int n = Hi[H_N], na = Hi[H_NA], ab = Hi[H_AB], off = Hi[H_OFF] & 0x0fffffff;
  const bool on = k < n;
  const int slot = 4 * k + ((ab >> (k < na ? 0 : 16)) & 1023) - H_AB_BIAS;
  float J = 0.f, B = 0.f, v = 0.f;
  if (on) { if (far) lv_ld2g(Y.E + 2 * (off + k), J, B); else lv_ld2(Y.lds, Y.pairs_addr + 8 * (off + k), J, B); v = lv_ld1(Y.lds, Y.dv_addr + slot); }
  const float jdv = wave_sum16(on ? J * v : 0.f);
  const float l0 = lv_ld1(Y.lds, Y.lam_addr + 4 * row);
  float lo = H[H_LO], hi = H[H_HI];
  if (fric) { hi = lv_ld1(Y.lds, Y.lam_addr + 4 * (MAX_ROWS + row)); lo = -hi; }
  const float nl = wave_clamp(l0 + (H[H_B] - jdv) * H[H_INVD], lo, hi);
  const float dl = nl - l0;
  wave_fence();                                                     // (emulator: the four groups have read before any of them writes)
  if (on) { lv_st1(Y.lds, Y.lam_addr + 4 * row, nl); lv_st1(Y.lds, Y.dv_addr + slot, v + B * dl); }
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
#define LVS_YES(x) x
#define LVS_NO(x)
#define LVS_NOT_LVS_YES(x)
#define LVS_NOT_LVS_NO(x) x
#define LVS_ENTRY(FRIC, FAR, N_OFF, N_N, N_NA, N_AB, N_BIT, EN_JB, EN_IA, EN_ON, EN_ONLO, EN_LA, EN_LAM, EN_HI) \
  "s_bfm_b32 " EN_ONLO ", " N_N ", 0\n" \
  "s_bfm_b32 vcc_lo, " N_NA ", 0\n" \
  "v_lshl_add_u32 v98, " N_OFF ", 3, %[k8p]\n" \
  "v_cndmask_b32_e64 v99, 16, 0, vcc\n" \
  FAR("global_load_dwordx2 " EN_JB ", v98, %[E]\n") LVS_NOT_##FAR("ds_read_b64 " EN_JB ", v98\n") \
  "v_bfe_u32 v99, " N_AB ", v99, 10\n" \
  "v_add_u32_e32 " EN_IA ", v99, %[k4dv]\n" \
  "v_add_u32_e32 " EN_LA ", " N_BIT ", %[lamm]\n" \
  "ds_read_b32 " EN_LAM ", " EN_LA "\n" \
  FRIC("ds_read_b32 " EN_HI ", " EN_LA " offset:%c[hioff]\n")
#define LVS_HEADER(N_OCT, N_BIT) \
  "s_ff1_i32_b64 s99, s[88:89]\n" \
  "s_bitset0_b64 s[88:89], s99\n" \
  "s_lshl_b32 s99, s99, 6\n" \
  "s_add_u32 s99, s99, %[base64]\n" \
  "s_load_dwordx8 " N_OCT ", %[Hm], s99\n" \
  "s_lshr_b32 " N_BIT ", s99, 4\n"
#define LVS_WAIT(FAR) FAR("s_waitcnt vmcnt(0) lgkmcnt(0)\n") LVS_NOT_##FAR("s_waitcnt lgkmcnt(0)\n")
#define LVS_STEP(FRIC, FAR, C_INVD, C_B, C_LO, C_HI, C_BIT, N1_OFF, N1_N, N1_NA, N1_AB, N1_BIT, N3_OCT, N3_BIT, EC_J, EC_B, EC_IA, EC_ON, EC_LA, EC_LAM, EC_HI, EN_JB, EN_IA, EN_ON, EN_ONLO, EN_LA, EN_LAM, EN_HI) \
  "ds_read_b32 v94, " EC_IA "\n" \
  LVS_ENTRY(FRIC, FAR, N1_OFF, N1_N, N1_NA, N1_AB, N1_BIT, EN_JB, EN_IA, EN_ON, EN_ONLO, EN_LA, EN_LAM, EN_HI) \
  LVS_WAIT(FAR) \
  LVS_HEADER(N3_OCT, N3_BIT) \
  "v_mul_f32_e32 v95, " EC_J ", v94\n" \
  "v_cndmask_b32_e64 v95, 0, v95, " EC_ON "\n" \
  FRIC("s_nop 1\n") LVS_NOT_##FRIC("v_mov_b32_e32 v98, " C_LO "\n" "s_nop 0\n") \
  LVS_DPP("quad_perm:[1,0,3,2]") "s_nop 1\n" LVS_DPP("quad_perm:[2,3,0,1]") "s_nop 1\n" LVS_DPP("row_half_mirror") "s_nop 1\n" LVS_DPP("row_mirror") \
  "v_sub_f32_e32 v96, " C_B ", v95\n" \
  "v_fma_f32 v97, " C_INVD ", v96, " EC_LAM "\n" \
  FRIC("v_med3_f32 v96, v97, -" EC_HI ", " EC_HI "\n") LVS_NOT_##FRIC("v_med3_f32 v96, v97, v98, " C_HI "\n") \
  "v_sub_f32_e32 v97, v96, " EC_LAM "\n" \
  "v_fmac_f32_e32 v94, " EC_B ", v97\n" \
  "s_mov_b64 exec, " EC_ON "\n" \
  "ds_write_b32 " EC_IA ", v94\n" \
  "ds_write_b32 " EC_LA ", v96\n" \
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
#define LVS_EC_P "v88", "v89", "v90", "s[90:91]", "v100", "v101", "v102"
#define LVS_EC_Q "v92", "v93", "v91", "s[92:93]", "v103", "v104", "v105"
#define LVS_EN_P "v[88:89]", "v90", "s[90:91]", "s90", "v100", "v101", "v102"
#define LVS_EN_Q "v[92:93]", "v91", "s[92:93]", "s92", "v103", "v104", "v105"
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
    LVS_WAIT(FAR) \
    "s_mov_b64 exec, s[50:51]\n"
#define LVS_CLOBBERS \
      "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", \
      "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", \
      "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc", "memory"
// an exhausted cursor gives bit -1: the header address is then base64 - 64 with base64 = 64 (base + 1) against Hm = H - 64 bytes, i.e. the row
// before `base` (or, for base 0, the last 64 bytes of the pair arena in front of the headers): loaded, never visited
#define LVS_ASM(FRIC, FAR, K8) \
  asm volatile(LVS_BODY(FRIC, FAR) \
    : \
    : [mask] "s"(mask), [nvis1] "s"(nvis1), [base64] "s"(base64), [Hm] "s"(Hm), [E] "s"(Y.E), [k8p] "v"(K8), [k4dv] "v"(4 * lane + Y.dv_addr - H_AB_BIAS), \
      [lamm] "v"(Y.lam_addr - 4), [hioff] "n"(4 * MAX_ROWS) \
    : LVS_CLOBBERS)
// far: the rows' pairs lie beyond the LDS window: loaded from the scratch record (vmcnt) instead
AGX_DEV void lvs_part_asm(const LvsLay& Y, int lane, uint64_t mask, int base, bool fric, bool far) {
  const int nvis1 = popc64(mask) - 1, base64 = 64 * (base + 1);
  const float* Hm = Y.H - HDR_STRIDE;
  if (!far) { if (fric) LVS_ASM(LVS_YES, LVS_NO, 8 * lane + Y.pairs_addr); else LVS_ASM(LVS_NO, LVS_NO, 8 * lane + Y.pairs_addr); }
  else { if (fric) LVS_ASM(LVS_YES, LVS_YES, 8 * lane); else LVS_ASM(LVS_NO, LVS_YES, 8 * lane); }
}
#endif

// the rows base + (set bits of mask), ascending; fric: friction rows, bounds -+ the row's entry of the bound array.
// Pair offsets grow with the row index: the rows whose pairs lie beyond the LDS window are a suffix [Y.rfar, ...), visited after the others.
AGX_DEV void lvs_part(const LvsLay& Y, int lane, uint64_t mask, int base, bool fric) {
  if (!mask) return;
  const int nn = Y.rfar - base;
  const uint64_t near = nn >= 64 ? mask : (nn <= 0 ? 0ull : mask & pgs_range_mask(0, nn)), far = mask & ~near;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_LV_CPP)
  // (the loop narrows EXEC to lanes 0..15 itself and restores it)
  if (near) lvs_part_asm(Y, lane, near, base, fric, false);
  if (far) lvs_part_asm(Y, lane, far, base, fric, true);
#else
  for (uint64_t m = near; m; m &= m - 1ull) lvs_visit(Y, lane, base + ffs64(m), fric, false);
  for (uint64_t m = far; m; m &= m - 1ull) lvs_visit(Y, lane, base + ffs64(m), fric, true);
#endif
  wave_fence();
}

AGX_DEV void pgs_lvs(Ctx& c, float* lds, int lds_words, float& dv0, float& dv1) {
  const int lane = c.lane, iters = (int)PRM(c, AGX_P_NITER);
  const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc, R = c.nrows;       // rows: [0,nnc) non-contact, [nnc,nA) normals, then nc friction rows per direction
  LvsLay Y; Y.lds = lds; Y.H = c.H; Y.E = c.E; Y.dv_addr = lv_addr(lds, lds + LVS_DV); Y.lam_addr = lv_addr(lds, lds + LVS_LAM); Y.pairs_addr = lv_addr(lds, lds + LVS_PAIRS);
  // ---- prologue (all 64 lanes): velocity deltas, impulses, the window of pairs, the first row beyond it
  const int win = lvs_window(lds_words);
  lds[LVS_DV + lane] = 0.f; lds[LVS_DV + 64 + lane] = 0.f;
  for (int r = lane; r < 2 * MAX_ROWS; r += 64) lds[LVS_LAM + r] = 0.f;
  { const f2* src = (const f2*)c.E; f2* dst = (f2*)(lds + LVS_PAIRS); const int np = c.nent < win ? c.nent : win; for (int q = lane; q < np; q += 64) dst[q] = src[q]; }
  { int far_first = R;
    for (int r = lane; r < R; r += 64) { const int* Hi = (const int*)(c.H + HDR_STRIDE * r); if ((Hi[H_OFF] & 0x0fffffff) + Hi[H_N] > win && r < far_first) far_first = r; }
    Y.rfar = (int)wave_min((float)far_first); }
  // friction coefficient of this lane's contact (0 for a row without effective mass: pinned at zero impulse)
  const bool two_dirs = R > nA + nc;
  float mu1 = 0.f, mu2 = 0.f;
  if (lane < nc) {
    const float* H = c.H + HDR_STRIDE * (nA + lane); mu1 = H[H_INVD] != 0.f ? H[H_MU] : 0.f;
    if (two_dirs) { const float* H2 = c.H + HDR_STRIDE * (nA + nc + lane); mu2 = H2[H_INVD] != 0.f ? H2[H_MU] : 0.f; }
  }
  wave_sync();
  const uint64_t rowsA0 = pgs_range_mask(0, nA < 64 ? nA : 64), rowsA1 = nA > 64 ? pgs_range_mask(0, nA - 64) : 0ull;
  const int K = noop_period(c);                                     // the no-op re-test rule: see pgs()
  uint64_t skip0 = 0ull, skip1 = 0ull;
  float* LAM = lds + LVS_LAM; float* HI = lds + LVS_HI;
  for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0, use = K > 0 && !retest;
    float bef0 = 0.f, bef1 = 0.f;
    if (retest) { if (lane < nA) bef0 = LAM[lane]; if (64 + lane < nA) bef1 = LAM[64 + lane]; }
    lvs_part(Y, lane, rowsA0 & ~(use ? skip0 : 0ull), 0, false);
    lvs_part(Y, lane, rowsA1 & ~(use ? skip1 : 0ull), 64, false);
    if (retest) {
      const float af0 = lane < nA ? LAM[lane] : 0.f, af1 = 64 + lane < nA ? LAM[64 + lane] : 0.f;
      skip0 = wave_ballot(af0 == bef0); skip1 = wave_ballot(af1 == bef1);
    }
    for (int dir = 0; dir < (two_dirs ? 2 : 1); dir++) {
      // friction rows (lane = contact): bounds from the normal impulses as this sweep's normal pass left them; a row whose normal
      // impulse and own impulse are both zero is an exact no-op and is not visited
      const int f0 = nA + dir * nc;
      float ln = 0.f, lf = 0.f;
      if (lane < nc) { ln = LAM[nnc + lane]; lf = LAM[f0 + lane]; HI[f0 + lane] = (dir ? mu2 : mu1) * ln; }
      const uint64_t todo = wave_ballot(lane < nc && (ln != 0.f || lf != 0.f));
      wave_sync();
      lvs_part(Y, lane, todo, f0, true);
    }
  }
  wave_sync();
  // velocity deltas back to their DoF lanes; solved normal impulses -> contact records (what getContactPoints reports until the next step)
  dv0 = lds[LVS_DV + lane]; dv1 = lds[LVS_DV + 64 + lane];
  if (lane < nc) c.gcon[CON_STRIDE * lane + C_LAM] = LAM[nnc + lane];
  wave_sync();
}

}  // namespace agx
