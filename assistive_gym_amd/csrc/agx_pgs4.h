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
// LDS of the packed kernel (float words): the per-environment epilogue reuses [L_ST, L_VEL + 128) of the single-environment layout
constexpr int P4_BASE = L_VEL + 128;
constexpr int P4_LAM = 0, P4_LIST = P4_LAM + MAX_ROWS, P4_SKIP = P4_LIST + MAX_ROWS, P4_GROUP_WORDS = P4_SKIP + MAX_ROWS;
constexpr int P4_DV = P4_BASE + 4 * P4_GROUP_WORDS;      // [4][128] velocity deltas in DoF order for the epilogue
constexpr int LDS_SOLVE4_WORDS = P4_DV + 4 * 128;
constexpr int LDS_SOLVE4_BYTES = LDS_SOLVE4_WORDS * 4;

AGX_DEV void solve_tail(Ctx& c, float* gstate, Scratch& scr, int sw, int phase, float dv0, float dv1);

// env_first: environment of group 0; n_envs / active as in the single-environment kernels
AGX_DEV void env_solve4(const uint32_t* blob, float* gstate_all, float* gscratch_all, int env_first, int n_envs, int sw, const uint8_t* active, float* lds, int lane, int phase) {
  const int g = lane >> 4, j = lane & 15;
  const int env = env_first + g;
  const bool valid = env < n_envs && (!active || active[env]);
  const float* bf = (const float*)blob; const int* bi = (const int*)blob;
  float* scrb = gscratch_all + (size_t)(env < n_envs ? env : n_envs - 1) * SCR_WORDS;
  const int* meta = (const int*)(scrb + SCR_O_META);
  const int nnc = valid ? meta[META_NNC] : 0, nc = valid ? meta[META_NCON] : 0, nA = nnc + nc, R = nA + nc;
  const float* BH = scrb + SCR_O_BRH; const float* BE = scrb + SCR_O_BRE;
  float* LAM = lds + P4_BASE + g * P4_GROUP_WORDS + P4_LAM;
  int* LIST = (int*)(lds + P4_BASE + g * P4_GROUP_WORDS + P4_LIST);
  int* SKIP = (int*)(lds + P4_BASE + g * P4_GROUP_WORDS + P4_SKIP);
  for (int r = j; r < MAX_ROWS; r += 16) { LAM[r] = 0.f; SKIP[r] = 0; LIST[r] = 0; }
  float dv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        if (take) LIST[lenA + __builtin_popcount(m & ((1u << j) - 1u))] = r;
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
          if (take) LIST[first + len + __builtin_popcount(m & ((1u << j) - 1u))] = r;
          len += __builtin_popcount(m);
        }
        wave_sync();
      }
      for (int t = 0; wave_any(t < len); t++) {
        const bool on = t < len;
        const int r = on ? LIST[first + t] : 0;
        const float4 h0 = *(const float4*)(BH + BRH_WORDS * r);
        const int4 h1 = *(const int4*)(BH + BRH_WORDS * r + 4);
        const uint64_t map = (uint64_t)(uint32_t)h1.x | ((uint64_t)(uint32_t)h1.y << 32);
        const int e = on ? (int)((map >> (4 * j)) & 15ull) : 0;
        float J0 = 0.f, J1 = 0.f, J2 = 0.f, J3 = 0.f, J4 = 0.f, J5 = 0.f, B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f, B4 = 0.f, B5 = 0.f;
        if (e) {
          const float4* p = (const float4*)(BE + BRE_WORDS * (h1.z + e - 1));
          const float4 a = p[0], b = p[1], cc = p[2];
          J0 = a.x; J1 = a.y; J2 = a.z; J3 = a.w; J4 = b.x; J5 = b.y; B0 = b.z; B1 = b.w; B2 = cc.x; B3 = cc.y; B4 = cc.z; B5 = cc.w;
        }
        const float x = ((J0 * dv[0] + J1 * dv[1]) + (J2 * dv[2] + J3 * dv[3])) + (J4 * dv[4] + J5 * dv[5]);
        // every lane reads the impulses BEFORE the cross-lane sum: lane 0 of the group rewrites LAM[r] below (lock step on the GPU; on
        // the fibre emulator the sum is the rendezvous that orders these reads before that write)
        const float lam = LAM[r];
        const bool fric = part == 1;
        const float lamn = LAM[(fric && on) ? r - nc : r];
        const float jdv = g16_sum(x);
        const float hi = fric ? h0.w * lamn : h0.w, lo = fric ? -hi : h0.z;
        const float nl = wave_clamp(lam + (h0.y - jdv) * h0.x, lo, hi);
        const float dl = on ? nl - lam : 0.f;
        if (on && j == 0) { LAM[r] = nl; if (retest && !fric) SKIP[r] = dl == 0.f ? 1 : 0; }
        dv[0] += B0 * dl; dv[1] += B1 * dl; dv[2] += B2 * dl; dv[3] += B3 * dl; dv[4] += B4 * dl; dv[5] += B5 * dl;
        wave_sync();      // the next visit of this group reads the impulses written above (one wavefront: a fence, not a wait for other waves)
      }
    }
  }
  // solved normal impulses -> contact records (what getContactPoints reports until the next step)
  if (valid) { float* gcon = scrb + SCR_O_CON; for (int r = nnc + j; r < nA; r += 16) gcon[CON_STRIDE * (r - nnc) + C_LAM] = LAM[r]; }
  // velocity deltas in DoF order, then integration + hooks one environment at a time with the whole wave (the single-environment code)
  {
    const int ndof = bi[AGX_H_NDOF], nfree = bi[AGX_H_NFREE];
    float* DV = lds + P4_DV + 128 * g;
    for (int k = j; k < 128; k += 16) DV[k] = 0.f;
    wave_sync();
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
    const float dv0 = lds[P4_DV + 128 * q + lane], dv1 = lds[P4_DV + 128 * q + 64 + lane];
    solve_tail(c, gstate_all + (size_t)e2 * sw, scr, sw, phase, dv0, dv1);
    wave_sync();
  }
}

}  // namespace agx
