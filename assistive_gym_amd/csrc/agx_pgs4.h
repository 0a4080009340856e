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

// which variants solve with the packed kernel (the others keep the one-wave-per-environment sweeps of agx_pgs.h)
#ifndef AGX_USE_SOLVE4
#define AGX_USE_SOLVE4 (AGX_TASK == 0)
#endif
constexpr bool USE_SOLVE4 = AGX_USE_SOLVE4;
// LDS of the packed kernel (float words): the per-environment epilogue reuses [L_ST, L_VEL + 128) of the single-environment layout; each
// of the four groups has its impulses, its visit list and skip flags (bytes) and P4_ROWMEM words of row memory: the headers of all its
// rows (4 words each), then a window of the first units of its rows -- units beyond the window are read from the scratch record (L2).
// 4 x 40 KB = the CU's 160 KB: one wavefront per SIMD, every environment of a 4096-environment batch resident at once.
constexpr int P4_BASE = L_VEL + 128;
constexpr int P4_LAM = 0, P4_LIST = P4_LAM + MAX_ROWS, P4_SKIP = P4_LIST + MAX_ROWS / 4, P4_ROWS = P4_SKIP + MAX_ROWS / 4;
constexpr int P4_ROWMEM = (10240 - 64 - P4_BASE - 4 * 128) / 4 - P4_ROWS;
constexpr int P4_GROUP_WORDS = P4_ROWS + P4_ROWMEM;
constexpr int P4_DV = P4_BASE + 4 * P4_GROUP_WORDS;      // [4][128] velocity deltas in DoF order for the epilogue
constexpr int LDS_SOLVE4_WORDS = P4_DV + 4 * 128;
constexpr int LDS_SOLVE4_BYTES = LDS_SOLVE4_WORDS * 4;
static_assert(LDS_SOLVE4_BYTES <= 40 * 1024 && P4_ROWMEM >= MAX_ROWS * BRH_WORDS && P4_ROWS % 2 == 0 && P4_BASE % 2 == 0 && P4_GROUP_WORDS % 2 == 0, "four wavefronts per CU; 8-byte aligned row memory");

AGX_DEV void solve_tail(Ctx& c, float* gstate, Scratch& scr, int sw, int phase, float dv0, float dv1);

struct P4Ent { f2 j0, j1, j2, b0, b1, b2; };      // J[6] of this lane's block, B[6] for articulated blocks
// this lane's part of the row described by `desc`: issues the loads (window in LDS or scratch in L2); lanes the row does not touch get zeros
AGX_DEV void p4_fetch(uint32_t desc, bool on, int j, int wunits, const float* ENT, const float* BE, P4Ent& E) {
  const f2 z = {0.f, 0.f};
  E.j0 = z; E.j1 = z; E.j2 = z; E.b0 = z; E.b1 = z; E.b2 = z;
  const int k0 = desc & 15, nart = (desc >> 4) & 15, fa = (desc >> 8) & 15, fb = (desc >> 12) & 15; const int eoff = (int)(desc >> 18);
  const bool isart = j < NB_ART; const int f1 = j - NB_ART + 1;
  const bool has = on && (isart ? (unsigned)(j - k0) < (unsigned)nart : (f1 == fa || f1 == fb));
  const int unit = isart ? eoff + 2 * (j - k0) : eoff + 2 * nart + ((f1 == fb && fa != 0) ? 1 : 0);
  if (has) {
    if (unit + (isart ? 1 : 0) < wunits) {
      const f2* p = (const f2*)(ENT + BRU_WORDS * unit);
      E.j0 = p[0]; E.j1 = p[1]; E.j2 = p[2];
      if (isart) { E.b0 = p[3]; E.b1 = p[4]; E.b2 = p[5]; }
    } else {
      const f2* p = (const f2*)(BE + BRU_WORDS * unit);
      E.j0 = p[0]; E.j1 = p[1]; E.j2 = p[2];
      if (isart) { E.b0 = p[3]; E.b1 = p[4]; E.b2 = p[5]; }
    }
  }
}

// env_first: environment of group 0; env_end: end of the launch's environment range; active as in the single-environment kernels
AGX_DEV void env_solve4(const uint32_t* blob, float* gstate_all, float* gscratch_all, int env_first, int n_envs, int sw, const uint8_t* active, float* lds, int lane, int phase) {
  const int g = lane >> 4, j = lane & 15;
  const int env = env_first + g;
  const bool valid = env < n_envs && (!active || active[env]);
  const float* bf = (const float*)blob; const int* bi = (const int*)blob;
  float* scrb = gscratch_all + (size_t)(env < n_envs ? env : n_envs - 1) * SCR_WORDS;
  const int* meta = (const int*)(scrb + SCR_O_META);
  const int nnc = valid ? meta[META_NNC] : 0, nc = valid ? meta[META_NCON] : 0, nA = nnc + nc, R = nA + nc, nunits = valid ? meta[META_NBENT] : 0;
  const float* BH = scrb + SCR_O_BRH; const float* BE = scrb + SCR_O_BRE;
  float* G = lds + P4_BASE + g * P4_GROUP_WORDS;
  float* LAM = G + P4_LAM; uint8_t* LIST = (uint8_t*)(G + P4_LIST); uint8_t* SKIP = (uint8_t*)(G + P4_SKIP);
  float* HDR = G + P4_ROWS; float* ENT = HDR + BRH_WORDS * ((R + 1) & ~1);       // headers, then the window of units (8-byte aligned)
  const int wunits = (P4_ROWMEM - BRH_WORDS * ((R + 1) & ~1)) / BRU_WORDS;
  static_assert(MAX_ROWS <= 255, "visit lists are bytes");
  for (int r = j; r < MAX_ROWS; r += 16) { LAM[r] = 0.f; SKIP[r] = 0; LIST[r] = 0; }
  for (int r = j; r < R; r += 16) *(float4*)(HDR + BRH_WORDS * r) = *(const float4*)(BH + BRH_WORDS * r);
  { const int nw = (nunits < wunits ? nunits : wunits) * (BRU_WORDS / 2);
    for (int k = j; k < nw; k += 16) ((f2*)ENT)[k] = ((const f2*)BE)[k]; }
  // inverse mass and world inverse inertia of this lane's free body (B = M^-1 J of its rows is formed from J)
  float im = 0.f, ixx = 0.f, ixy = 0.f, ixz = 0.f, iyy = 0.f, iyz = 0.f, izz = 0.f;
  const bool isart = j < NB_ART;
  if (!isart && j - NB_ART < bi[AGX_H_NFREE]) {
    const float* F = scrb + SCR_O_BRF + BRF_WORDS * (j - NB_ART);
    im = F[0]; ixx = F[1]; ixy = F[2]; ixz = F[3]; iyy = F[4]; iyz = F[5]; izz = F[6];
  }
  float dv0 = 0.f, dv1 = 0.f, dv2 = 0.f, dv3 = 0.f, dv4 = 0.f, dv5 = 0.f;
  const int iters = (int)bf[bi[AGX_H_OFF_PARAMS] + AGX_P_NITER], K = (int)bf[bi[AGX_H_OFF_PARAMS] + AGX_P_NOOP_RETEST];
  int lenA = 0;
  wave_sync();
  for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0;
    // the non-friction part of the visit list changes only around a re-test sweep (every row / the rows that are not skipped)
    if (it == 0 || (K > 0 && (it % K == 0 || it % K == 1))) {
      lenA = 0;
      for (int base = 0; wave_any(base < nA); base += 16) {
        const int r = base + j;
        const bool take = r < nA && (K <= 0 || retest || !SKIP[r]);
        const uint32_t m = g16_ballot(take, g);
        if (take) LIST[lenA + __builtin_popcount(m & ((1u << j) - 1u))] = (uint8_t)r;
        lenA += __builtin_popcount(m);
      }
      wave_sync();
    }
    for (int part = 0; part < 2; part++) {
      int len = lenA, first = 0;
      if (part == 1) {
        // friction rows: a row whose normal impulse (as this sweep's normal pass left it) and own impulse are both zero is an exact no-op
        wave_sync();
        len = 0; first = lenA;
        for (int base = 0; wave_any(base < nc); base += 16) {
          const int r = nA + base + j;
          const bool take = base + j < nc && (LAM[r - nc] != 0.f || LAM[r] != 0.f);
          const uint32_t m = g16_ballot(take, g);
          if (take) LIST[first + len + __builtin_popcount(m & ((1u << j) - 1u))] = (uint8_t)r;
          len += __builtin_popcount(m);
        }
        wave_sync();
      }
      // software pipeline: step t computes with the header and units fetched during step t - 1; the header of step t + 2 and the units
      // of step t + 1 are requested before the arithmetic of step t (a group's visit list is fixed for the part, so every address is known)
      int r0 = 0 < len ? LIST[first] : 0, r1 = 1 < len ? LIST[first + 1] : 0;
      float4 H0 = *(const float4*)(HDR + BRH_WORDS * r0), H1 = *(const float4*)(HDR + BRH_WORDS * r1);
      P4Ent E0; p4_fetch(__builtin_bit_cast(uint32_t, H0.w), 0 < len, j, wunits, ENT, BE, E0);
      for (int t = 0; wave_any(t < len); t++) {
        const bool on = t < len;
        const int r2 = t + 2 < len ? LIST[first + t + 2] : 0;
        const float4 H2 = *(const float4*)(HDR + BRH_WORDS * r2);
        P4Ent E1; p4_fetch(__builtin_bit_cast(uint32_t, H1.w), t + 1 < len, j, wunits, ENT, BE, E1);
        const int r = r0;
        const int cls = (int)((__builtin_bit_cast(uint32_t, H0.w) >> 16) & 3u);
        const float J0 = E0.j0.x, J1 = E0.j0.y, J2 = E0.j1.x, J3 = E0.j1.y, J4 = E0.j2.x, J5 = E0.j2.y;
        float B0, B1, B2, B3, B4, B5;
        if (isart) { B0 = E0.b0.x; B1 = E0.b0.y; B2 = E0.b1.x; B3 = E0.b1.y; B4 = E0.b2.x; B5 = E0.b2.y; }
        else { B0 = im * J0; B1 = im * J1; B2 = im * J2; B3 = ixx * J3 + ixy * J4 + ixz * J5; B4 = ixy * J3 + iyy * J4 + iyz * J5; B5 = ixz * J3 + iyz * J4 + izz * J5; }
        const float x = ((J0 * dv0 + J1 * dv1) + (J2 * dv2 + J3 * dv3)) + (J4 * dv4 + J5 * dv5);
        // every lane reads the impulses BEFORE the cross-lane sum: lane 0 of the group rewrites LAM[r] below (lock step on the GPU; on
        // the fibre emulator the sum is the rendezvous that orders these reads before that write)
        const float lam = LAM[r];
        const float lamn = LAM[(cls == BR_CLASS_FRIC && on) ? r - nc : r];
        const float jdv = g16_sum(x);
        const float hi = cls == BR_CLASS_SYM ? H0.z : (cls == BR_CLASS_POS ? 1e30f : H0.z * lamn), lo = cls == BR_CLASS_POS ? 0.f : -hi;
        const float nl = wave_clamp(lam + (H0.y - jdv) * H0.x, lo, hi);
        const float dl = on ? nl - lam : 0.f;
        if (on && j == 0) { LAM[r] = nl; if (retest && part == 0) SKIP[r] = dl == 0.f ? 1 : 0; }
        dv0 += B0 * dl; dv1 += B1 * dl; dv2 += B2 * dl; dv3 += B3 * dl; dv4 += B4 * dl; dv5 += B5 * dl;
        wave_fence();     // the next visit of this group reads the impulse written above: program order inside one wavefront, no wait
        r0 = r1; r1 = r2; H0 = H1; H1 = H2; E0 = E1;
      }
    }
  }
  wave_sync();
  // solved normal impulses -> contact records (what getContactPoints reports until the next step)
  if (valid) { float* gcon = scrb + SCR_O_CON; for (int r = nnc + j; r < nA; r += 16) gcon[CON_STRIDE * (r - nnc) + C_LAM] = LAM[r]; }
  // velocity deltas in DoF order, then integration + hooks one environment at a time with the whole wave (the single-environment code)
  {
    const int ndof = bi[AGX_H_NDOF], nfree = bi[AGX_H_NFREE];
    float* DV = lds + P4_DV + 128 * g;
    for (int k = j; k < 128; k += 16) DV[k] = 0.f;
    wave_sync();
    const float dv[6] = {dv0, dv1, dv2, dv3, dv4, dv5};
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
