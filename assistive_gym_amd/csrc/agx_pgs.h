// agx_pgs.h -- K6 projected Gauss-Seidel: register-resident row sets, the row sweep in gfx950 assembly (and its C++ twin).
// Part of the FeedingJaco stepper (see agx_step.h for the overview); included by agx_step.h only.
#pragma once

namespace agx {

// ---- K6: projected Gauss-Seidel --------------------------------------------------------------------------
// Rows are visited in construction order (Gauss-Seidel is order dependent).  Row headers and the
// accumulated impulses live in registers, distributed over the lanes (lane r&63 owns row r of slot
// r>>6) and are broadcast with v_readlane (wave-uniform -> SGPRs, scalar control flow).  Per row
// every lane fetches its (J,B) pair from the LDS arena (lanes outside the row's two DoF ranges read
// nothing), one DPP reduction gives J.dv, the impulse update is uniform, every lane applies
// B*dlambda to the DoFs it owns.  The fetch of row r+1 is issued before the reduction of row r.
// Register sets: A0/A1 hold rows 0..127 of the non-contact + normal block (lane r&63 of set r>>6),
// B0/B1 hold the friction rows, placed in the SAME lane as the normal row of their contact so the
// friction bound mu*lambda_n is a lane-local product.  The impulse update is evaluated in every
// lane on its own row registers; only the owner lane's result is kept and its delta broadcast.
struct PgsSet { float invD, b, lo, hi, lam; int pack, off, mlo, mhi, m2; };   // one row per lane (hi = mu for friction sets); off carries OFF_TWO_BIT

// (J,B) pair of this lane for a row: lanes outside the row's two DoF ranges read arena entry 0 = (0,0).
// Addresses are 32-bit byte offsets from the (wave-uniform) entry base, so the loads use the
// SGPR-base + VGPR-offset form and need no 64-bit address arithmetic.
struct PgsBuf { float j0, c0, j1, c1; };
AGX_DEV void pgs_fetch(const float* E, int lane, int pack, int off, PgsBuf& X) {
  const unsigned a0 = pack & 255, na = (pack >> 8) & 255, b0 = (pack >> 16) & 255, nb = (unsigned)pack >> 24;
  const unsigned oa = 8u * (unsigned)off, ob = 8u * ((unsigned)off + na);
  const char* Eb = (const char*)E;
  unsigned ia = (unsigned)lane - a0, ib = (unsigned)lane - b0;
  unsigned e = ib < nb ? ob + 8u * ib : 0u;
  e = ia < na ? oa + 8u * ia : e;
  unsigned e1 = 0u;
  if (a0 + na > 64 || b0 + nb > 64) {   // wave-uniform: only rows touching DoFs 64.. have entries in the second slot
    ia = (unsigned)lane + 64u - a0; ib = (unsigned)lane + 64u - b0;
    e1 = ib < nb ? ob + 8u * ib : 0u;
    e1 = ia < na ? oa + 8u * ia : e1;
  }
  // both loads are always issued (the second one degenerates to a broadcast of the zero pair): the
  // number of loads in flight is then the same on every path and the waits can be exact
  const f2 p = *(const f2*)(Eb + e);
  const f2 q = *(const f2*)(Eb + e1);
  X.j0 = p.x; X.c0 = p.y; X.j1 = q.x; X.c1 = q.y;
}
// one Gauss-Seidel pass over the rows held in lanes [l0, l1) of one register set.  The (J,B) pairs
// stream from the per-env scratch (L2).  Three named buffers rotate through the roles "in use",
// "next" and "being fetched" (the loop is unrolled by three so that no register moves are needed and
// the loads of rows r+1 and r+2 stay in flight while row r is reduced).  Prefetches past the end
// re-read the last row instead of being skipped, again to keep the number of loads in flight fixed.
template <bool FRICTION>
AGX_DEV void pgs_row(PgsSet& S, const float& lam_normal, const PgsBuf& X, int lane, int rl, float& dv0, float& dv1) {
  const float jdv = wave_sum(X.j0 * dv0 + X.j1 * dv1);
  const float hi = FRICTION ? S.hi * lam_normal : S.hi, lo = FRICTION ? -hi : S.lo;
  const float nl = wave_clamp(S.lam + (S.b - jdv) * S.invD, lo, hi);
  const float dlo = nl - S.lam;
  S.lam = (lane == rl) ? nl : S.lam;
  const float dl = wave_bcast(dlo, rl);
  dv0 += X.c0 * dl;
  wave_opaque(dv0);       // keeps the two updates scalar: a packed FMA would need (c0, c1) in adjacent registers
  dv1 += X.c1 * dl;
}
AGX_DEV uint64_t pgs_range_mask(int l0, int l1) {
  const uint64_t hi = l1 >= 64 ? ~0ull : ((1ull << l1) - 1ull), lo = l0 >= 64 ? ~0ull : ((1ull << l0) - 1ull);
  return hi & ~lo;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_CPP)
// ---- the Gauss-Seidel sweep in gfx950 assembly ------------------------------------------------------
// The compiler's schedule of the loop above is poor in exactly the places that matter: it rotates
// the prefetch buffers with register moves and, because the number of loads in flight differs between
// paths, falls back to s_waitcnt vmcnt(0) right after issuing a prefetch.  Here the row loop is written
// out by hand: four register buffers W,X,Y,Z rotate through "in use / +1 / +2 / being fetched"
// (unrolled by four, no moves), every row issues exactly two global_load_dwordx2 (rows that do not
// reach DoFs 64.. load the zero pair for the second slot), so s_waitcnt vmcnt(6) is exact, and the
// prefetch index is clamped to the last row of the sweep instead of being skipped.
// Hazards (gfx940 family; the assembler inserts nothing in inline asm): VALU-written VGPR -> DPP 2
// wait states, VALU-written SGPR/VCC -> VALU read 2, -> v_readlane lane select 4; spacing below
// keeps to these with independent instructions or s_nop.
// Register map: v64..v79 buffers, v80..v88 temporaries, s80..s95 scalars.
#define AGX_STR2(x) #x
#define AGX_STR(x) AGX_STR2(x)
#if AGX_ST_WORDS == 336          // a literal: the offset is pasted into the assembly text
#define AGX_SOLVE_ENT_BYTES 1856
#elif AGX_ST_WORDS == 344
#define AGX_SOLVE_ENT_BYTES 1888
#endif
static_assert(AGX_SOLVE_ENT_BYTES == 4 * L_SOLVE_ENT, "LDS offset of the row window used by the assembly");
// the two sources of a row's pairs: the global scratch (vmcnt) or the LDS window (lgkmcnt)
#define AGX_LOAD_G(DST, ADDR) "global_load_dwordx2 " DST ", " ADDR ", %[E]\n"
#define AGX_LOAD_L(DST, ADDR) "ds_read_b64 " DST ", " ADDR " offset:" AGX_STR(AGX_SOLVE_ENT_BYTES) "\n"
#define AGX_WAIT_G(N) "s_waitcnt vmcnt(" N ")\n"
#define AGX_WAIT_L(N) "s_waitcnt lgkmcnt(" N ")\n"
// pairs of row IDX -> buffer (Z0: lanes 0..63, Z1: lanes 64..).  Bit 31 of the offset word (second-slot
// flag) needs no masking: the shift by 3 of the address arithmetic discards it.  The row's lane mask (precomputed by
// row_store) turns the address into "offset + rank of this lane among the row's lanes": 2 x v_mbcnt,
// 1 add-shift, 1 select with the mask itself as the condition.
#define AGX_PGS_FETCH(LOAD, IDX, Z0, Z1) \
  "v_readlane_b32 s84, %[off], " IDX "\n" \
  "v_readlane_b32 s82, %[mlo], " IDX "\n" \
  "v_readlane_b32 s83, %[mhi], " IDX "\n" \
  "s_bitcmp1_b32 s84, 31\n" \
  "v_mbcnt_lo_u32_b32 v81, s82, 0\n" \
  "v_mbcnt_hi_u32_b32 v81, s83, v81\n" \
  "v_add_lshl_u32 v85, v81, s84, 3\n" \
  "v_cndmask_b32_e64 v85, 0, v85, s[82:83]\n" \
  LOAD(Z0, "v85") \
  "s_cbranch_scc0 1f\n" \
  "v_readlane_b32 s86, %[m2], " IDX "\n" \
  "s_bcnt1_i32_b64 s85, s[82:83]\n" \
  "s_mov_b32 s87, 0\n" \
  "s_add_i32 s85, s85, s84\n" \
  "v_mbcnt_lo_u32_b32 v82, s86, 0\n" \
  "v_add_lshl_u32 v86, v82, s85, 3\n" \
  "v_cndmask_b32_e64 v86, 0, v86, s[86:87]\n" \
  LOAD(Z1, "v86") \
  "s_branch 2f\n" \
  "1:\n" \
  LOAD(Z1, "v88") \
  "2:\n"
// The address arithmetic of the row-ahead fetch, woven into the wait states of the reduction.  -DAGX_PGS_NO_ADDR (timing ablation, results
// meaningless: the fetch reads the zero pair): solve kernel 1.51 -> 1.11 ms per 4096 FeedingJaco environments -- these 3 v_readlane and
// 4 vector instructions cost 26 %.  Tried instead (round 2, correct on the GPU, slower): per-row fetch headers read with scalar loads and
// the pairs of a row's two contiguous DoF ranges loaded under EXEC = the range's lanes (2 vector instructions per row instead of 11, but
// 4 global loads per row and no LDS window): 2.7 ms -- a wave64 load costs the memory pipe the same whatever its EXEC mask.
#ifdef AGX_PGS_NO_ADDR
#define AGX_PGS_ADDR0 "s_mov_b32 s84, 0\n"
#define AGX_PGS_ADDR1 ""
#define AGX_PGS_ADDR2 ""
#define AGX_PGS_ADDR3 "v_mov_b32_e32 v85, 0\n"
#define AGX_PGS_ADDR4 "s_bitcmp1_b32 s84, 31\n"
#elif defined(AGX_PGS_NO_READLANE3)   // timing ablation (results meaningless): the three v_readlane of the row-ahead fetch as scalar moves -- what fetching the row descriptors some other way (scalar loads) could save at most
#define AGX_PGS_ADDR0 "s_mov_b32 s84, 8\n" "s_mov_b32 s82, 0xfff\n"
#define AGX_PGS_ADDR1 "s_mov_b32 s83, 0\n" "s_bitcmp1_b32 s84, 31\n"
#define AGX_PGS_ADDR2 "v_mbcnt_lo_u32_b32 v81, s82, 0\n"
#define AGX_PGS_ADDR3 "v_mbcnt_hi_u32_b32 v81, s83, v81\n" "v_add_lshl_u32 v85, v81, s84, 3\n"
#define AGX_PGS_ADDR4 "v_cndmask_b32_e64 v85, 0, v85, s[82:83]\n"
#else
#define AGX_PGS_ADDR0 "v_readlane_b32 s84, %[off], s80\n" "v_readlane_b32 s82, %[mlo], s80\n"
#define AGX_PGS_ADDR1 "v_readlane_b32 s83, %[mhi], s80\n" "s_bitcmp1_b32 s84, 31\n"
#define AGX_PGS_ADDR2 "v_mbcnt_lo_u32_b32 v81, s82, 0\n"
#define AGX_PGS_ADDR3 "v_mbcnt_hi_u32_b32 v81, s83, v81\n" "v_add_lshl_u32 v85, v81, s84, 3\n"
#define AGX_PGS_ADDR4 "v_cndmask_b32_e64 v85, 0, v85, s[82:83]\n"
#endif
#define AGX_PGS_DPP(CTRL) "v_add_f32_dpp v80, v80, v80 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// One row: the dependent chain (dot product, 6-step DPP reduction, impulse update, broadcast) with the
// address arithmetic of the prefetch for row r+3 woven into its wait states.
#define AGX_PGS_NEXT(IDX, MASK) "s_ff1_i32_b64 " IDX ", " MASK "\n" "s_bitset0_b64 " MASK ", " IDX "\n"
#define AGX_PGS_STEP(LOAD, WAIT, XJ0, XC0, XJ1, XC1, Z0, Z1) \
  AGX_PGS_NEXT("s94", "%[mask]") \
  AGX_PGS_NEXT("s80", "s[96:97]") \
  WAIT("4") \
  "v_mul_f32_e32 v80, " XJ0 ", %[dv0]\n" \
  "v_fmac_f32_e32 v80, " XJ1 ", %[dv1]\n" \
  AGX_PGS_ADDR0 \
  AGX_PGS_DPP("quad_perm:[1,0,3,2]") \
  AGX_PGS_ADDR1 \
  AGX_PGS_DPP("quad_perm:[2,3,0,1]") \
  AGX_PGS_ADDR2 \
  "v_cmp_eq_u32_e32 vcc, s94, %[lane]\n" \
  AGX_PGS_DPP("row_shr:4") \
  AGX_PGS_ADDR3 \
  AGX_PGS_DPP("row_shr:8") \
  AGX_PGS_ADDR4 \
  LOAD(Z0, "v85") \
  AGX_PGS_DPP("row_bcast:15") \
  "s_nop 1\n" \
  AGX_PGS_DPP("row_bcast:31") \
  "s_nop 0\n" \
  "v_readlane_b32 s92, v80, 63\n" \
  "s_nop 1\n" \
  "v_subrev_f32_e32 v80, s92, %[b]\n" \
  "v_fma_f32 v80, %[invD], v80, %[lam]\n" \
  "v_med3_f32 v80, v80, %[lo], %[hi]\n" \
  "v_sub_f32_e32 v87, v80, %[lam]\n" \
  "v_cndmask_b32_e32 %[lam], %[lam], v80, vcc\n" \
  "s_cbranch_scc0 1f\n" \
  "v_readlane_b32 s86, %[m2], s80\n" \
  "s_bcnt1_i32_b64 s85, s[82:83]\n" \
  "s_mov_b32 s87, 0\n" \
  "s_add_i32 s85, s85, s84\n" \
  "v_mbcnt_lo_u32_b32 v82, s86, 0\n" \
  "v_add_lshl_u32 v86, v82, s85, 3\n" \
  "v_cndmask_b32_e64 v86, 0, v86, s[86:87]\n" \
  LOAD(Z1, "v86") \
  "s_branch 2f\n" \
  "1:\n" \
  LOAD(Z1, "v88") \
  "2:\n" \
  "v_readlane_b32 s93, v87, s94\n" \
  "s_cmp_eq_u64 %[mask], 0\n" \
  "s_nop 0\n" \
  "v_fmac_f32_e32 %[dv0], s93, " XC0 "\n" \
  "v_fmac_f32_e32 %[dv1], s93, " XC1 "\n" \
  "s_cbranch_scc1 9f\n"
// The rows to visit are the set bits of %[mask] (lane = row slot), taken in ascending order with
// s_ff1 / s_bitset0; a second cursor (s[96:97]) runs three rows ahead for the prefetch (when it runs
// dry its index is -1, i.e. lane 63: a harmless extra fetch).
#define AGX_PGS_BODY(LOAD, WAIT) \
    "v_mov_b32_e32 v88, 0\n" \
    "s_mov_b64 s[96:97], %[mask]\n" \
    AGX_PGS_NEXT("s80", "s[96:97]") \
    AGX_PGS_FETCH(LOAD, "s80", "v[64:65]", "v[66:67]") \
    AGX_PGS_NEXT("s80", "s[96:97]") \
    AGX_PGS_FETCH(LOAD, "s80", "v[68:69]", "v[70:71]") \
    AGX_PGS_NEXT("s80", "s[96:97]") \
    AGX_PGS_FETCH(LOAD, "s80", "v[72:73]", "v[74:75]") \
    "8:\n" \
    AGX_PGS_STEP(LOAD, WAIT, "v64", "v65", "v66", "v67", "v[76:77]", "v[78:79]") \
    AGX_PGS_STEP(LOAD, WAIT, "v68", "v69", "v70", "v71", "v[64:65]", "v[66:67]") \
    AGX_PGS_STEP(LOAD, WAIT, "v72", "v73", "v74", "v75", "v[68:69]", "v[70:71]") \
    AGX_PGS_STEP(LOAD, WAIT, "v76", "v77", "v78", "v79", "v[72:73]", "v[74:75]") \
    "s_branch 8b\n" \
    "9:\n" \
    WAIT("0")
#define AGX_PGS_OPERANDS \
    : [lam] "+v"(S.lam), [dv0] "+v"(dv0), [dv1] "+v"(dv1), [mask] "+s"(mask) \
    : [off] "v"(S.off), [mlo] "v"(S.mlo), [mhi] "v"(S.mhi), [m2] "v"(S.m2), [invD] "v"(S.invD), [b] "v"(S.b), [lo] "v"(lo), [hi] "v"(hi), \
      [lane] "v"(lane), [E] "s"(E) \
    : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", \
      "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "vcc", "scc", "memory"
// lo/hi are the per-lane bounds of this sweep (for friction sets already scaled by the normal
// impulses).  `rows` has one bit per row slot to visit; rows below slot `ls` have all their pairs inside
// the LDS window, the others stream from global.
AGX_DEV void pgs_sweep_asm(PgsSet& S, float lo, float hi, const float* E, int lane, uint64_t rows, int ls, float& dv0, float& dv1) {
  const uint64_t in_lds = rows & pgs_range_mask(0, ls), in_glb = rows & ~pgs_range_mask(0, ls);
  if (in_lds) {
    uint64_t mask = in_lds;
    asm volatile(AGX_PGS_BODY(AGX_LOAD_L, AGX_WAIT_L) AGX_PGS_OPERANDS);
  }
  if (in_glb) {
    uint64_t mask = in_glb;
    asm volatile(AGX_PGS_BODY(AGX_LOAD_G, AGX_WAIT_G) AGX_PGS_OPERANDS);
  }
}
#endif
// `skip`: row slots of [l0, l1) that are not visited in this sweep (the no-op re-test rule of pgs(), AGX_P_NOOP_RETEST)
template <bool FRICTION>
AGX_DEV void pgs_sweep(PgsSet& S, const float& lam_normal, const float* E, int lane, int l0, int ls, int l1, float& dv0, float& dv1, uint64_t skip = 0ull) {
  if (l1 <= l0) return;
  uint64_t rows = pgs_range_mask(l0, l1) & ~skip;
  // A friction row whose normal impulse is zero has the bounds [0, 0]; if its own impulse is zero as
  // well its update is exactly "no change", so the visit is skipped.  The normal impulses do not
  // change during a friction sweep and a friction impulse only changes at its own visit, so the set
  // of rows to visit is known up front.  (More than half of the contacts are speculative and inactive.)
  if (FRICTION) rows &= wave_ballot(lam_normal != 0.f || S.lam != 0.f);
  if (!rows) return;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_CPP)
  const float hi = FRICTION ? S.hi * lam_normal : S.hi, lo = FRICTION ? -hi : S.lo;
  pgs_sweep_asm(S, lo, hi, E, lane, rows, ls, dv0, dv1);
#else
  (void)ls;
  while (rows) {
    const int rl = ffs64(rows); rows &= rows - 1ull;
    PgsBuf X; pgs_fetch(E, lane, wave_bcast_i(S.pack, rl), wave_bcast_i(S.off, rl) & 0x7fffffff, X);
    pgs_row<FRICTION>(S, lam_normal, X, lane, rl, dv0, dv1);
  }
#endif
}
AGX_DEV void pgs_load_set(const Ctx& c, int row, bool ok, bool friction, PgsSet& S) {
  ok = ok && row < MAX_ROWS;
  const float* H = c.H + HDR_STRIDE * (ok ? row : 0); const int* Hi = (const int*)H; const float* X = hx_row(c.H, ok ? row : 0); const int* Xi = (const int*)X;
  const float invD = ok ? H[H_INVD] : 0.f;
  // a row without effective mass (static-static, degenerate) is kept but pinned at zero impulse
  const bool live = ok && invD != 0.f;
  S.invD = invD; S.b = ok ? H[H_B] : 0.f; S.lam = 0.f;
  S.lo = live ? H[H_LO] : 0.f; S.hi = live ? (friction ? X[H_MU] : H[H_HI]) : 0.f;
  S.pack = ok ? Xi[H_PACK] : 0; S.off = ok ? Hi[H_OFF] : 0;
  S.mlo = ok ? Xi[H_MLO] : 0; S.mhi = ok ? Xi[H_MHI] : 0; S.m2 = ok ? Xi[H_M2] : 0;
}
// first lane of [l0, l1) whose row reaches beyond the LDS window of (J,B) pairs (l1 if none)
AGX_DEV int pgs_lds_split(const PgsSet& S, int lane, int l0, int l1) {
  if (l1 <= l0) return l0;
#ifdef AGX_NO_LDS_ROWS
  return l0;
#endif
  const int end = (S.off & 0x7fffffff) + ((S.pack >> 8) & 255) + (int)((unsigned)S.pack >> 24);
  const uint64_t m = wave_ballot(lane >= l0 && lane < l1 && end > SOLVE_LDS_PAIRS);
  return wave_uniform(m ? ffs64(m) : l1);
}

// ---- row-space sweep for environments with few rows -----------------------------------------------------
// The velocity-space sweep above pays a 6-step cross-lane reduction (J.dv) inside the dependent chain of every row visit.
// With at most RS_MAX_ROWS rows the coupling matrix A[i][j] = J_j . B_i (how much an impulse on row i changes the velocity
// row j measures) fits LDS next to the state: lane j then carries w_j = J_j . dv incrementally -- a visit of row i is
// "new impulse from (b_i - w_i), broadcast its change, w += A[i][:] * change": no reduction, one broadcast.  Measured on MI355X
// (tools/gpu_pgs_cycles.py): ~145 cycles per row visit instead of ~740, but the work area (30 KB) allows 5 environments per CU
// instead of 16, so the solve launch of BedBathingSawyer only went from 0.57 to 0.50 ms and its step rate from 890 k to 995 k.  Same rows, same order, same clamps as the velocity-space sweep; the results differ by rounding only
// (sums are associated differently).  dv = sum_i B_i lambda_i is formed once at the end.
// Returns false (nothing done) if the environment has more rows than fit or touches DoFs beyond lane 63.
// K of the no-op re-test rule for this environment and substep (see pgs()): AGX_P_NOOP_RETEST, or 0 = plain sweeps when one of the
// substep's contacts is pressed deeper than AGX_P_NOOP_PEN or is a contact of the robot / its tool with the person (include/agx_blob.h).  Wave-uniform; lane = contact (MAX_CON <= 64).
AGX_DEV int noop_period(const Ctx& c) {
  const int K = (int)PRM(c, AGX_P_NOOP_RETEST); const float pen = PRM(c, AGX_P_NOOP_PEN);
  if (K <= 0 || !(pen > 0.f)) return K;
  bool pressed = false;
  if (c.lane < c.ncon) {                                            // ... or is one of the robot / its tool with the person: the forces the task reports come from plain sweeps
    const float* o = c.gcon + CON_STRIDE * c.lane; const int* oi = (const int*)o;
    const int ta = CLI(c, oi[C_CA], AGX_C_TAG), tb = CLI(c, oi[C_CB], AGX_C_TAG);
    pressed = o[C_DIST] < -pen || (ta == AGX_TAG_HUMAN && (tb == AGX_TAG_ROBOT || tb == AGX_TAG_TOOL)) || (tb == AGX_TAG_HUMAN && (ta == AGX_TAG_ROBOT || ta == AGX_TAG_TOOL));
  }
  return wave_any(pressed) ? 0 : K;
}
AGX_DEV bool pgs_rowspace(Ctx& c, float* W, float& dv0, float& dv1) {
  if constexpr (RS_MAX_ROWS == 0) { (void)c; (void)W; (void)dv0; (void)dv1; return false; }
  else {
    const int lane = c.lane, iters = (int)PRM(c, AGX_P_NITER);
    const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc, R = c.nrows, nv = c.nv;       // R = nA + nc friction rows (AGX_P_FRICTION_DIRS = 2: + nc more)
    if (R > RS_MAX_ROWS || nv > 64 || nv > RS_NVP || R == 0 || c.nent > SOLVE_LDS_PAIRS) return false;
    const float* E = W;                                            // the (J,B) window holds every pair of this environment (env_solve copied them)
    PgsSet S; pgs_load_set(c, lane, lane < R, lane >= nA, S);      // lane r owns row r; friction rows: S.hi = mu
    float* LJ = W + RS_J; float* A = W + RS_A; float* ROW = W + RS_ROW; float* LAM = W + RS_LAM;
    const long long tA0 = c.dbg ? wave_clock() : 0;                 // debug runs: cycles of forming the dense Jacobians and A (DBG_TIME + 7)
    // dense Jacobians J[r][d] and response rows B[r][d] = (M^-1 J_r^T)[d]; B overlays the region of A until A is written
    float* LB = A;
    static_assert(RS_NVP <= RS_MAX_ROWS || RS_MAX_ROWS == 0, "the dense B rows fit the region of A");
    for (int r = 0; r < R; r++) {
      PgsBuf X; pgs_fetch(E, lane, wave_bcast_i(S.pack, r), wave_bcast_i(S.off, r) & 0x7fffffff, X);
      if (lane < nv) { LJ[RS_NVP * r + lane] = X.j0; LB[RS_NVP * r + lane] = X.c0; }
    }
    wave_sync();
    // A = B J^T (A[i][j] = sum_d B_i[d] J_j[d], symmetric up to the rows' DoF ranges) on the matrix cores: 32 x 32 tiles of
    // v_mfma_f32_32x32x2_f32 steps over d (exact f32), one tile for up to 32 rows, four for up to 56.  Formed by a per-lane loop over d
    // this was a quarter to a third of the whole solve (measured: 29 ... 50 k of 119 ... 160 k cycles per environment and substep).
    {
      const int li = lane & 31, hb = lane >> 5;
      const bool two = R > 32;
      Acc16 c00, c01, c10, c11; acc16_zero(c00); acc16_zero(c01); acc16_zero(c10); acc16_zero(c11);
      for (int d0 = 0; d0 < nv; d0 += 2) {
        const int d = d0 + hb; const bool dk = d < nv;
        const float b0 = (dk && li < R) ? LB[RS_NVP * li + d] : 0.f, j0 = (dk && li < R) ? LJ[RS_NVP * li + d] : 0.f;
        wave_mfma_32x32x2(b0, j0, c00);
        if (two) {
          const float b1 = (dk && li + 32 < R) ? LB[RS_NVP * (li + 32) + d] : 0.f, j1 = (dk && li + 32 < R) ? LJ[RS_NVP * (li + 32) + d] : 0.f;
          wave_mfma_32x32x2(b0, j1, c01); wave_mfma_32x32x2(b1, j0, c10); wave_mfma_32x32x2(b1, j1, c11);
        }
      }
      wave_sync();                                                  // every lane has read B: its region becomes A
      for (int v = 0; v < 16; v++) {
        const int i = (v & 3) + 8 * (v >> 2) + 4 * hb;
        if (i < R && li < R) A[RS_MAX_ROWS * i + li] = acc16_get(c00, v);
        if (two) {
          if (i < R && li + 32 < R) A[RS_MAX_ROWS * i + li + 32] = acc16_get(c01, v);
          if (i + 32 < R && li < R) A[RS_MAX_ROWS * (i + 32) + li] = acc16_get(c10, v);
          if (i + 32 < R && li + 32 < R) A[RS_MAX_ROWS * (i + 32) + li + 32] = acc16_get(c11, v);
        }
      }
    }
    wave_sync();
    if (c.dbg && lane == 0) c.dbg[DBG_TIME + 7] = (float)(wave_clock() - tA0);
    float w = 0.f;                                                  // J_r . dv of this lane's row
    if (PRM(c, AGX_P_WARMSTART) > 0.f) {                            // warm start: the normals begin at the impulses the build kernel seeded (warm_seed)
      const float l0 = (lane >= nnc && lane < nA) ? c.gcon[CON_STRIDE * (lane - nnc) + C_LAM] : 0.f;
      S.lam = l0;
      for (uint64_t m = wave_ballot(l0 != 0.f); m; m &= m - 1ull) { const int r = ffs64(m); w += A[RS_MAX_ROWS * r + lane] * wave_bcast(l0, r); }
    }
    const int fn = lane - (lane >= nA + nc ? 2 * nc : nc);          // normal row of this lane's friction row (first / second direction block)
    const int K = noop_period(c);                                   // the no-op re-test rule, see pgs()
    uint64_t skip = 0ull;
    for (int it = 0; it < iters; it++) {
      const bool retest = K > 0 && it % K == 0;
      const float before = S.lam;
      uint64_t rowsA = pgs_range_mask(0, nA) & ~((K > 0 && !retest) ? skip : 0ull);
      while (rowsA) {                                               // non-contact rows and contact normals
        const int r = ffs64(rowsA); rowsA &= rowsA - 1ull;
        const float arow = A[RS_MAX_ROWS * r + lane];
        const float nl = wave_clamp(S.lam + (S.b - w) * S.invD, S.lo, S.hi);
        const float dl = wave_bcast(nl - S.lam, r);
        if (lane == r) S.lam = nl;
        w += arow * dl;
      }
      if (retest) skip = wave_ballot(S.lam == before) & pgs_range_mask(0, nA);
      if (nc > 0) {
        wave_sync(); LAM[lane] = S.lam; wave_sync();
        const float ln = (lane >= nA && lane < R) ? LAM[fn] : 0.f;  // the normal impulses do not change during the friction pass
        const float hi = S.hi * ln, lo = -hi;
        // a friction row whose normal impulse and own impulse are both zero is an exact no-op (as in the velocity-space sweep)
        uint64_t todo = wave_ballot(lane >= nA && lane < R && (ln != 0.f || S.lam != 0.f));
        while (todo) {
          const int r = ffs64(todo); todo &= todo - 1ull;
          const float arow = A[RS_MAX_ROWS * r + lane];
          const float nl = wave_clamp(S.lam + (S.b - w) * S.invD, lo, hi);
          const float dl = wave_bcast(nl - S.lam, r);
          if (lane == r) S.lam = nl;
          w += arow * dl;
        }
      }
    }
    // dv = sum_r B_r lambda_r, in row order
    dv0 = 0.f; dv1 = 0.f;
    for (int r = 0; r < R; r++) {
      PgsBuf X; pgs_fetch(E, lane, wave_bcast_i(S.pack, r), wave_bcast_i(S.off, r) & 0x7fffffff, X);
      dv0 += X.c0 * wave_bcast(S.lam, r);
    }
    if (lane >= nnc && lane < nA) c.gcon[CON_STRIDE * (lane - nnc) + C_LAM] = S.lam;
    return true;
  }
}
AGX_DEV void pgs(Ctx& c, float& dv0, float& dv1) {
  const int lane = c.lane; const int iters = (int)PRM(c, AGX_P_NITER);
  const float* E = c.E;
  const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc;      // rows: [0,nnc) non-contact, [nnc,nA) normals, [nA,nA+nc) friction
  static_assert(MAX_ROWS <= 256 && MAX_CON <= 64, "two register sets per block");
  PgsSet A0, A1, B0, B1;
  pgs_load_set(c, lane, lane < nA, false, A0);
  pgs_load_set(c, 64 + lane, 64 + lane < nA, false, A1);
  { const int c0 = lane - nnc, c1 = 64 + lane - nnc;
    pgs_load_set(c, nA + c0, c0 >= 0 && c0 < nc, true, B0);
    pgs_load_set(c, nA + c1, c1 >= 0 && c1 < nc, true, B1); }
  const int a0n = nA < 64 ? nA : 64, a1n = nA - 64;
  const int f0a = nnc < 64 ? nnc : 64, f0b = nA < 64 ? nA : 64;          // friction rows in B0: lanes [nnc, min(nA,64))
  const int f1a = nnc > 64 ? nnc - 64 : 0, f1b = nA - 64;                // friction rows in B1: lanes [max(nnc-64,0), nA-64)
  dv0 = 0.f; dv1 = 0.f;
  // rows whose pairs lie inside the LDS window (offsets grow with the row index, so per register set
  // this is a prefix of its lane range)
  const int s0 = pgs_lds_split(A0, lane, 0, a0n), s1 = pgs_lds_split(A1, lane, 0, a1n);
  const int t0 = pgs_lds_split(B0, lane, f0a, f0b), t1 = pgs_lds_split(B1, lane, f1a, f1b);
  // AGX_P_NOOP_RETEST = K > 0: a row of the non-friction block (motors, limits, tool rows, contact normals) whose visit in a re-test
  // sweep (sweep index divisible by K) changed nothing -- an inactive contact or limit (0 -> 0), a motor sitting on its force bound --
  // is not visited in the K - 1 sweeps that follow (42 % of these visits are no-ops in FeedingJaco; the oracle applies the same
  // rule, pgs() in oracle/agx_oracle.c; sensitivity: profiles/r03/noop_retest_sensitivity.json).  K = 0: every row in every sweep.
  if (PRM(c, AGX_P_WARMSTART) > 0.f) {                              // warm start (see pgs_rowspace): impulses of the memory, dv = sum_r B_r lambda_r in row order
    const int r0 = lane, r1 = 64 + lane;
    const float l0 = (r0 >= nnc && r0 < nA) ? c.gcon[CON_STRIDE * (r0 - nnc) + C_LAM] : 0.f, l1 = (r1 >= nnc && r1 < nA) ? c.gcon[CON_STRIDE * (r1 - nnc) + C_LAM] : 0.f;
    A0.lam = l0; A1.lam = l1;
    for (uint64_t m = wave_ballot(l0 != 0.f); m; m &= m - 1ull) {
      const int r = ffs64(m); PgsBuf X; pgs_fetch(E, lane, wave_bcast_i(A0.pack, r), wave_bcast_i(A0.off, r) & 0x7fffffff, X);
      const float dl = wave_bcast(l0, r); dv0 += X.c0 * dl; dv1 += X.c1 * dl;
    }
    for (uint64_t m = wave_ballot(l1 != 0.f); m; m &= m - 1ull) {
      const int r = ffs64(m); PgsBuf X; pgs_fetch(E, lane, wave_bcast_i(A1.pack, r), wave_bcast_i(A1.off, r) & 0x7fffffff, X);
      const float dl = wave_bcast(l1, r); dv0 += X.c0 * dl; dv1 += X.c1 * dl;
    }
  }
  const int K = noop_period(c);
  uint64_t skip0 = 0ull, skip1 = 0ull;
  const bool two_dirs = c.nrows > nA + nc;
  float lamC0 = 0.f, lamC1 = 0.f;
  for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0, use = K > 0 && !retest;
    const float b0 = A0.lam, b1 = A1.lam;
    pgs_sweep<false>(A0, A0.lam, E, lane, 0, s0, a0n, dv0, dv1, use ? skip0 : 0ull);
    pgs_sweep<false>(A1, A1.lam, E, lane, 0, s1, a1n, dv0, dv1, use ? skip1 : 0ull);
    if (retest) { skip0 = wave_ballot(A0.lam == b0); skip1 = wave_ballot(A1.lam == b1); }
    pgs_sweep<true>(B0, A0.lam, E, lane, f0a, t0, f0b, dv0, dv1);
    pgs_sweep<true>(B1, A1.lam, E, lane, f1a, t1, f1b, dv0, dv1);
    if (two_dirs) {
      // AGX_P_FRICTION_DIRS = 2: the block of second directions, one row per contact again in the lane of its normal row.  Its row sets are
      // re-read from the headers in every sweep (only the impulses stay in registers): the switch is for parity studies, the default path
      // pays two registers for it
      PgsSet C0, C1;
      { const int c0 = lane - nnc, c1 = 64 + lane - nnc;
        pgs_load_set(c, nA + nc + c0, c0 >= 0 && c0 < nc, true, C0);
        pgs_load_set(c, nA + nc + c1, c1 >= 0 && c1 < nc, true, C1); }
      C0.lam = lamC0; C1.lam = lamC1;
      const int u0 = pgs_lds_split(C0, lane, f0a, f0b), u1 = pgs_lds_split(C1, lane, f1a, f1b);
      pgs_sweep<true>(C0, A0.lam, E, lane, f0a, u0, f0b, dv0, dv1);
      pgs_sweep<true>(C1, A1.lam, E, lane, f1a, u1, f1b, dv0, dv1);
      lamC0 = C0.lam; lamC1 = C1.lam;
    }
  }
  // solved normal impulses -> contact records (what getContactPoints reports until the next step)
  { const int r0 = lane, r1 = 64 + lane;
    if (r0 >= nnc && r0 < nA) c.gcon[CON_STRIDE * (r0 - nnc) + C_LAM] = A0.lam;
    if (r1 >= nnc && r1 < nA) c.gcon[CON_STRIDE * (r1 - nnc) + C_LAM] = A1.lam; }
}

}  // namespace agx
