// agx_pgs4.h -- K6 packed: FOUR environments per wavefront, one 16-lane DPP row each, one lane per 6-DoF velocity block.
// Part of the stepper (see agx_step.h); included by agx_step.h only.
//
// Why: with one wavefront per environment the Gauss-Seidel chain of an environment occupies a whole wave -- 64 lanes for a dot product
// over the 12 ... 22 velocity entries a row touches, a 6-step cross-lane reduction and two v_readlane round trips per row visit -- and
// at 4096 environments a SIMD has only four such chains to interleave (profiles/r02_solve_kernel_bound.md).  Here the generalised
// velocity of an environment is cut into blocks of 6 (articulated DoFs first, then one block per free body; agx_ctx.h "block rows")
// and lane j of a 16-lane group holds block j of ITS environment in six registers.  A row visit is then, per lane: its entry
// (J[6], B[6], or nothing), six FMAs, a 4-step butterfly inside the DPP row (bitwise identical in the 16 lanes), the impulse update
// evaluated redundantly in every lane, six FMAs -- no broadcast through scalar registers at all -- and one instruction stream serves
// four environments.  Each group walks its OWN visit list (the rows it has to visit this sweep: the no-op re-test rule and the exact
// skip of friction rows without load, see agx_pgs.h), so environments with different row sets stay in lock step only by list position.
// Same rows, same order, same clamps as pgs(); sums are associated differently (rounding only).
#pragma once

namespace agx {

// ---- 16-lane group primitives (device: DPP row operations; the emulator has its own in tests/emu/agx_wave.h) ----
#if defined(__HIP__)
#define AGX_DPP_ROW_HALF_MIRROR 0x141
#define AGX_DPP_ROW_MIRROR 0x140
// sum over the 16 lanes of this lane's DPP row; xor-butterfly (1, 2, half mirror, mirror) so that every lane adds the same two
// numbers at every step: the result is bitwise the same in all 16 lanes
AGX_DEV float g16_sum(float x) {
  x += dpp_mov<AGX_DPP_QUAD_1032>(0.f, x);
  x += dpp_mov<AGX_DPP_QUAD_2301>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_HALF_MIRROR>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_MIRROR>(0.f, x);
  return x;
}
// the 16 ballot bits of this lane's group
AGX_DEV uint32_t g16_ballot(bool p, int group) { return (uint32_t)(__ballot(p) >> (16 * group)) & 0xffffu; }
#endif

// LDS of the packed kernel (float words), 40 KB = a quarter of the CU's 160 KB: one wavefront per SIMD, every environment of a
// 4096-environment batch resident at once.  16 zero words (what a lane reads for a row that does not touch its block), then per group:
// its impulses, its visit list (16-bit entries: row | class << 8; 8 entries of slack for the look-ahead reads), its skip flags (bytes) and
// the row memory: the headers of all its rows (BRH_WORDS each), then a window of the first units of its rows -- a FeedingJaco scene at
// rest (53 contacts, 122 rows, 260 units) fits entirely; units beyond the window are read from the scratch record (L2).  When the sweeps are over, the per-environment epilogue reuses the bottom
// of this memory as [L_ST, L_VEL + 128) of the single-environment layout, and the top 4 x 128 words as the velocity deltas in DoF order.
constexpr int LDS_SOLVE4_WORDS = 10240, LDS_SOLVE4_BYTES = LDS_SOLVE4_WORDS * 4;
constexpr int P4_ZERO = 0, P4_G0 = 16, P4_GROUP_WORDS = (LDS_SOLVE4_WORDS - P4_G0) / 4;
constexpr int P4_LIST_PAD = 8;
// a group's memory is cut to its row count R8 (R rounded up to 8): impulses [R8], list [R8 + pad] (16 bit), skip flags [R8] (bytes), row memory
AGX_DEV constexpr int p4_rows_words(int r8) { return r8 + (r8 + P4_LIST_PAD) / 2 + r8 / 4; }
constexpr int P4_DV = LDS_SOLVE4_WORDS - 4 * 128;        // [4][128]
static_assert(P4_GROUP_WORDS - p4_rows_words(MAX_ROWS) >= MAX_ROWS * BRH_WORDS + 4 * BRU_WORDS && P4_G0 % 2 == 0 && P4_GROUP_WORDS % 2 == 0 && MAX_ROWS % 8 == 0 && P4_LIST_PAD % 8 == 0, "8-byte aligned row memory");
static_assert(L_VEL + 128 <= P4_DV && MAX_ROWS <= 255, "the epilogue's single-environment region stays below the velocity deltas; rows fit the list's low byte");

AGX_DEV void solve_tail(Ctx& c, float* gstate, Scratch& scr, int sw, int phase, float dv0, float dv1);

// One look-ahead slot of the pipeline: J[6] of this lane's block and B[6]; for a free body's lane both are its single unit S J (agx_ctx.h:
// in the scaled velocity S^-1 v of a free body, S = (M^-1)^(1/2), the row's B IS its J).  `far`: the units lie beyond the LDS window
// (then at unit index `unit` of the scratch record)
struct P4Ent { f2 j0, j1, j2, b0, b1, b2; int unit; bool far; };
// what a lane needs to find its units of a row: its nibble word and the first-unit word of the row's header (agx_ctx.h)
struct P4Na { uint32_t x, y; };
AGX_DEV P4Na p4_na(const float* HDR, int row, int j) {
  const uint32_t* h = (const uint32_t*)HDR + BRH_WORDS * row + (j >> 3);       // lanes 0..7: words (0, 1); lanes 8..15: words (1, 2)
  P4Na n; n.x = h[0]; n.y = h[1]; return n;
}
// Requests this lane's units of the row behind `na` into E.  LDS pointers (the window or the zero unit) and unconditional ds_reads, so
// that the loads stay in flight across the arithmetic of the steps before their use: LDS answers in order and the compiler can wait
// for "all but the youngest n" (a flat pointer covering LDS and L2 would count against lgkmcnt AND vmcnt: every LDS wait would then
// wait for memory too).
AGX_DEV void p4_load(P4Na na, bool on, int j, int wunits, const float* ENT, const float* ZERO, P4Ent& E) {
  const bool lo8 = j < 8, isart = j < NB_ART;
  const uint32_t nibw = lo8 ? na.x : na.y, eoffw = lo8 ? na.y : na.x;
  const int nib = (int)((nibw >> (4 * (j & 7))) & 15u);
  const int unit = (int)(eoffw & 0xffffu) + nib - 1;
  const bool has = on & (nib != 0), inwin = unit + (isart ? 1 : 0) < wunits;
  const float* pj = (has & inwin) ? ENT + BRU_WORDS * unit : ZERO;
  const float* pb = isart ? pj + BRU_WORDS : pj;                     // (ZERO is two units long)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float p4_v2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) p4_v2* p4_lp;
  const p4_lp qj = (p4_lp)pj, qb = (p4_lp)pb;
  const p4_v2 u0 = qj[0], u1 = qj[1], u2 = qj[2], u3 = qb[0], u4 = qb[1], u5 = qb[2];
  E.j0 = {u0.x, u0.y}; E.j1 = {u1.x, u1.y}; E.j2 = {u2.x, u2.y}; E.b0 = {u3.x, u3.y}; E.b1 = {u4.x, u4.y}; E.b2 = {u5.x, u5.y};
#else
  const f2* qj = (const f2*)pj; const f2* qb = (const f2*)pb;
  E.j0 = qj[0]; E.j1 = qj[1]; E.j2 = qj[2]; E.b0 = qb[0]; E.b1 = qb[1]; E.b2 = qb[2];
#endif
  E.unit = unit; E.far = has & !inwin;
}
// just before the slot is used, in the (rare) parts whose lists hold rows beyond the window: those units from L2, waited for on the spot
AGX_DEV void p4_fix(P4Ent& E, const float* BE, int j) {
  if (wave_any(E.far)) {
    if (E.far) {
      const f2* q = (const f2*)(BE + BRU_WORDS * E.unit);
      E.j0 = q[0]; E.j1 = q[1]; E.j2 = q[2];
      const f2* qb = j < NB_ART ? q + 3 : q;
      E.b0 = qb[0]; E.b1 = qb[1]; E.b2 = qb[2];
    }
  }
}

// env_first: environment of group 0; n_envs: end of the launch's environment range; active as in the single-environment kernels
AGX_DEV void env_solve4(const uint32_t* blob, float* gstate_all, float* gscratch_all, int env_first, int n_envs, int sw, const uint8_t* active, float* lds, int lane, int phase) {
  const int g = lane >> 4, j = lane & 15;
  const int env = env_first + g;
  const bool valid = env < n_envs && (!active || active[env]);
  const float* bf = (const float*)blob; const int* bi = (const int*)blob;
  float* scrb = gscratch_all + (size_t)(env < n_envs ? env : n_envs - 1) * SCR_WORDS;
  const int* meta = (const int*)(scrb + SCR_O_META);
  const int nnc = valid ? meta[META_NNC] : 0, nc = valid ? meta[META_NCON] : 0, nA = nnc + nc, R = nA + nc, nunits = valid ? meta[META_NBENT] : 0;
  const float* BH = scrb + SCR_O_BRH; const float* BE = scrb + SCR_O_BRE;
  float* G = lds + P4_G0 + g * P4_GROUP_WORDS;
  const int R8 = (R + 7) & ~7;
  float* LAM = G; uint16_t* LIST = (uint16_t*)(G + R8); uint8_t* SKIP = (uint8_t*)(G + R8 + (R8 + P4_LIST_PAD) / 2);
  float* HDR = G + p4_rows_words(R8); float* ENT = HDR + BRH_WORDS * R;       // headers, then the window of units (8-byte aligned: BRH_WORDS is even)
#ifdef AGX_P4_WINDOW_CAP          // tests: a small window, so that the path for units beyond it runs on ordinary scenes
  const int wunits = AGX_P4_WINDOW_CAP;
#else
  const int wunits = (P4_GROUP_WORDS - p4_rows_words(R8) - BRH_WORDS * R) / BRU_WORDS;
#endif
  for (int r = j; r < R8; r += 16) { LAM[r] = 0.f; SKIP[r] = 0; }
  for (int r = j; r < R8 + P4_LIST_PAD; r += 16) LIST[r] = 0;
  if (lane < 16) lds[P4_ZERO + lane] = 0.f;
  if (R == 0 && j < BRH_WORDS) HDR[j] = 0.f;                         // the look-ahead reads of an empty list land on row 0
  { const int nw = R * (BRH_WORDS / 2); for (int k = j; k < nw; k += 16) ((f2*)HDR)[k] = ((const f2*)BH)[k]; }
  { const int nw = (nunits < wunits ? nunits : wunits) * (BRU_WORDS / 2); for (int k = j; k < nw; k += 16) ((f2*)ENT)[k] = ((const f2*)BE)[k]; }
  // the first row with units beyond the window (rows are stored in unit order)
  int rfar = R;
  for (int base = 0; wave_any(base < R); base += 16) {
    const int r = base + j;
    const int end = r + 1 < R ? (int)(((const uint32_t*)BH)[BRH_WORDS * (r + 1) + BRH_EOFF] & 0xffffu) : nunits;
    const uint32_t m = g16_ballot(r < R && end > wunits, g);
    if (m && rfar == R) rfar = base + __builtin_ctz(m);
  }
  // S = (M^-1)^(1/2) of this lane's free body: its velocity delta is kept as S^-1 dv until the epilogue
  float sm = 0.f, sxx = 0.f, sxy = 0.f, sxz = 0.f, syy = 0.f, syz = 0.f, szz = 0.f;
  if (j >= NB_ART && j - NB_ART < bi[AGX_H_NFREE]) {
    const float* F = scrb + SCR_O_BRF + BRF_WORDS * (j - NB_ART);
    sm = F[0]; sxx = F[1]; sxy = F[2]; sxz = F[3]; syy = F[4]; syz = F[5]; szz = F[6];
  }
  float dv0 = 0.f, dv1 = 0.f, dv2 = 0.f, dv3 = 0.f, dv4 = 0.f, dv5 = 0.f;
  // (one no-op period for the four environments of the wave: AGX_P_NOOP_PEN, the product kernel's per-environment switch to plain sweeps, is not implemented here)
  const int iters = (int)bf[bi[AGX_H_OFF_PARAMS] + AGX_P_NITER], K = (int)bf[bi[AGX_H_OFF_PARAMS] + AGX_P_NOOP_RETEST];
  const float* ZERO = lds + P4_ZERO;
  int lenA = 0; bool farA = false;
  wave_sync();
  for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0;
    // the non-friction part of the visit list changes only around a re-test sweep (every row / the rows that are not skipped)
    if (it == 0 || (K > 0 && (it % K == 0 || it % K == 1))) {
      lenA = 0; farA = rfar < nA;                  // (conservative on the sweeps that skip rows)
      for (int base = 0; wave_any(base < nA); base += 16) {
        const int r = base + j;
        const bool take = r < nA && (K <= 0 || retest || !SKIP[r]);
        const uint32_t m = g16_ballot(take, g);
        if (take) LIST[lenA + __builtin_popcount(m & ((1u << j) - 1u))] = (uint16_t)(r | ((((const uint32_t*)HDR)[BRH_WORDS * r + BRH_EOFF] >> 16) << 8));
        lenA += __builtin_popcount(m);
      }
      wave_sync();
    }
    for (int part = 0; part < 2; part++) {
      int len = lenA, first = 0; bool anyfar = farA;
      if (part == 1) {
        // friction rows: a row whose normal impulse (as this sweep's normal pass left it) and own impulse are both zero is an exact no-op
        wave_sync();
        len = 0; first = lenA;
        for (int base = 0; wave_any(base < nc); base += 16) {
          const int r = nA + base + j;
          const bool take = base + j < nc && (LAM[r - nc] != 0.f || LAM[r] != 0.f);
          const uint32_t m = g16_ballot(take, g);
          if (take) LIST[first + len + __builtin_popcount(m & ((1u << j) - 1u))] = (uint16_t)(r | (BR_CLASS_FRIC << 8));
          len += __builtin_popcount(m);
        }
        anyfar = rfar < R;
        wave_sync();
      }
      const bool mark = retest && part == 0;
      // Software pipeline over the group's visit list (fixed for the part, so every address is known ahead): at step t the list entry
      // of step t + 4, the nibble / first-unit words of step t + 3, the units of step t + 2 and the header of step t + 1 are requested
      // before the arithmetic of step t.  Three unit slots rotate through "in use / next / being fetched" (the loop body is three steps).
      // List entries past the end are stale but valid rows; `on` keeps them from having any effect.
      const uint16_t* LP = LIST + first;
      int rq0 = LP[0], rq1 = LP[1], rq2 = LP[2], rq3 = LP[3];
      P4Ent EA, EB, EC;
      p4_load(p4_na(HDR, rq0 & 255, j), 0 < len, j, wunits, ENT, ZERO, EA);
      p4_load(p4_na(HDR, rq1 & 255, j), 1 < len, j, wunits, ENT, ZERO, EB);
      P4Na na2 = p4_na(HDR, rq2 & 255, j);
      f2 Hc01 = *(const f2*)(HDR + BRH_WORDS * (rq0 & 255) + BRH_INVD); float Hcb = HDR[BRH_WORDS * (rq0 & 255) + BRH_BOUND];      // (1/D, b), bound
      float lamP = LAM[rq0 & 255], lamnP = LAM[(rq0 >> 8) == BR_CLASS_FRIC ? (rq0 & 255) - nc : (rq0 & 255)];
      int t = 0;
#define P4_STEP(ECUR, ENEW, FIX) { \
        const bool on = t < len; \
        const int r = rq0 & 255, cls = rq0 >> 8; \
        if (FIX) p4_fix(ECUR, BE, j); \
        /* the impulses of step t + 1 (never the row this step rewrites).  Every lane reads them BEFORE the cross-lane sum: lane 0 of the */ \
        /* group writes LAM[r] below (lock step on the GPU; on the fibre emulator the sum is the rendezvous that orders reads and write) */ \
        const float lam = lamP, lamn = lamnP; \
        { const int r1 = rq1 & 255; lamP = LAM[r1]; lamnP = LAM[(rq1 >> 8) == BR_CLASS_FRIC ? r1 - nc : r1]; } \
        wave_fence(); \
        const int rn = LP[t + 4]; \
        const P4Na na3 = p4_na(HDR, rq3 & 255, j); \
        p4_load(na2, t + 2 < len, j, wunits, ENT, ZERO, ENEW); \
        const f2 Hn01 = *(const f2*)(HDR + BRH_WORDS * (rq1 & 255) + BRH_INVD); const float Hnb = HDR[BRH_WORDS * (rq1 & 255) + BRH_BOUND]; \
        const float J0 = ECUR.j0.x, J1 = ECUR.j0.y, J2 = ECUR.j1.x, J3 = ECUR.j1.y, J4 = ECUR.j2.x, J5 = ECUR.j2.y; \
        const float B0 = ECUR.b0.x, B1 = ECUR.b0.y, B2 = ECUR.b1.x, B3 = ECUR.b1.y, B4 = ECUR.b2.x, B5 = ECUR.b2.y; \
        const float x = ((J0 * dv0 + J1 * dv1) + (J2 * dv2 + J3 * dv3)) + (J4 * dv4 + J5 * dv5); \
        const float jdv = g16_sum(x); \
        const float hi = cls == BR_CLASS_SYM ? Hcb : (cls == BR_CLASS_POS ? 1e30f : Hcb * lamn), lo = cls == BR_CLASS_POS ? 0.f : -hi; \
        const float nl = wave_clamp(lam + (Hc01.y - jdv) * Hc01.x, lo, hi); \
        const float dl = on ? nl - lam : 0.f; \
        if (on && j == 0) { LAM[r] = nl; if (mark) SKIP[r] = dl == 0.f ? 1 : 0; } \
        dv0 += B0 * dl; dv1 += B1 * dl; dv2 += B2 * dl; dv3 += B3 * dl; dv4 += B4 * dl; dv5 += B5 * dl; \
        wave_fence();     /* the next visit of this group reads the impulse written above: program order inside one wavefront, no wait */ \
        rq0 = rq1; rq1 = rq2; rq2 = rq3; rq3 = rn; na2 = na3; Hc01 = Hn01; Hcb = Hnb; t++; \
        if (!wave_any(t < len)) break; }
      if (!wave_any(0 < len)) continue;
      if (!wave_any(anyfar)) for (;;) {             // every unit of every group's list in its LDS window: the common case
        P4_STEP(EA, EC, false)
        P4_STEP(EB, EA, false)
        P4_STEP(EC, EB, false)
      } else for (;;) {
        P4_STEP(EA, EC, true)
        P4_STEP(EB, EA, true)
        P4_STEP(EC, EB, true)
      }
#undef P4_STEP
    }
  }
  wave_sync();
  // solved normal impulses -> contact records (what getContactPoints reports until the next step)
  if (valid) { float* gcon = scrb + SCR_O_CON; for (int r = nnc + j; r < nA; r += 16) gcon[CON_STRIDE * (r - nnc) + C_LAM] = LAM[r]; }
  wave_sync();                 // the velocity deltas of group 3 land in (what was) its own row memory
  // velocity deltas in DoF order, then integration + hooks one environment at a time with the whole wave (the single-environment code)
  {
    const int ndof = bi[AGX_H_NDOF], nfree = bi[AGX_H_NFREE];
    float* DV = lds + P4_DV + 128 * g;
    for (int k = j; k < 128; k += 16) DV[k] = 0.f;
    wave_sync();
    float dv[6] = {dv0, dv1, dv2, dv3, dv4, dv5};
    if (j >= NB_ART) {       // a free body: back from the scaled velocity
      dv[0] = sm * dv0; dv[1] = sm * dv1; dv[2] = sm * dv2;
      dv[3] = sxx * dv3 + sxy * dv4 + sxz * dv5; dv[4] = sxy * dv3 + syy * dv4 + syz * dv5; dv[5] = sxz * dv3 + syz * dv4 + szz * dv5;
    }
    for (int s = 0; s < 6; s++) {
      int d = -1;
      if (j < NB_ART) { if (6 * j + s < ndof) d = 6 * j + s; }
      else if (j - NB_ART < nfree) d = ndof + 6 * (j - NB_ART) + s;
      if (d >= 0) DV[d] = dv[s];
    }
    wave_sync();
  }
  for (int q = 0; q < 4; q++) {
    const int e2 = env_first + q;
    if (e2 >= n_envs || (active && !active[e2])) continue;       // wave uniform
    Ctx c; ctx_init(c, blob, lds, lane);
    Scratch scr = scratch_of(gscratch_all + (size_t)e2 * SCR_WORDS);
    const float d0 = lds[P4_DV + 128 * q + lane], d1 = lds[P4_DV + 128 * q + 64 + lane];
    solve_tail(c, gstate_all + (size_t)e2 * sw, scr, sw, phase, d0, d1);
    wave_sync();
  }
}

}  // namespace agx
