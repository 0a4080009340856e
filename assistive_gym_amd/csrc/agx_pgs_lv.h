// agx_pgs_lv.h -- K6, the row-local sweep: velocity deltas in LDS, lane = ENTRY of the visited row.
// Part of the stepper (see agx_step.h); included by agx_step.h only.
//
// Why (round 5).  The sweep of agx_pgs.h keeps the velocity deltas in registers (lane = DoF) and pays for that layout at every row visit:
// the (J,B) pairs of a row have to be found by the rank of the lane inside the row's DoF mask (3 v_readlane + 4 vector instructions), the
// dot product is a 6-step reduction over all 64 lanes followed by a v_readlane, the impulse change travels back through another
// v_readlane, and FeedingJaco's 74 velocity entries need a second register slot -- 25 vector instructions per visit, five of them the
// slow cross-lane-to-scalar kind; at four waves per SIMD the kernel is bound by the vector issue port (profiles/r03/solve_kernels_sq_counters.md).
// build_rows() already stores the pairs of a row CONTIGUOUSLY (articulated range, then the free bodies in ascending order), and a row of
// the feeding scenes has at most 16 of them (10 robot DoFs + 6 of a free body; two free bodies: 12).  So here lane k < 16 holds entry k of
// the visited row: it gathers dv[d_k] from LDS, the dot product is a 4-step xor butterfly inside ONE 16-lane DPP row whose result is
// bitwise the same in the 16 lanes, every lane evaluates the impulse update from a broadcast LDS read of the row header, and scatters
// dv[d_k] += B_k dlambda.  No v_readlane, no lane masks, no second slot; the sweeps run under EXEC = lanes 0..15.
// Row headers (1/D, b, lo, hi | lambda, byte offset, n, pack) live in LDS, 8 words per row, read three visits ahead; the pairs of the rows
// inside the LDS window and their dv byte offsets (16 bit) two visits ahead; rows beyond the window stream their pairs from the scratch
// record (L2), requested two visits ahead as well.
// Same rows, same order, same clamps, same no-op re-test rule and friction skipping as pgs(); the sums are associated differently
// (rounding only).  AGX_P_WARMSTART > 0 and rows longer than 16 entries keep the register sweep (lv_eligible).
#pragma once

namespace agx {

constexpr int LV_G = 16;                                            // lanes of a visit = the longest row this path takes
// MEASURED (round 5; profiles/r05/).  AGX_PGS_LV = 1, the visit loop as hipcc compiles it from the C++ below: 430 k env-steps/s against 467 k with the
// register sweep -- ~110 instructions per visit (16 vector, 7 LDS, the rest scalar bookkeeping, exec-mask branches, waits): 268 / 331 shader
// cycles per row and sweep with one / sixteen wavefronts per CU (register sweep 192 / 278).  AGX_PGS_LV = 2, the same visit on the same
// LDS tables written out in gfx950 assembly (lv_part_asm, 45 instructions): 155 / 168 cycles with every row inside the window, FeedingJaco
// 522 k env-steps/s (r05f_*).  Rows beyond the window still go through the C++ loop, so the solve launch of that build takes 20 KB of LDS
// (agx_kernels.hip: 8 solve waves per CU) -- with the register sweep's 9.5 KB a third of the visits are such rows: 437 k.
// AGX_PGS_LV = 3 (the default of round 5): agx_pgs_lvs.h, the same visit with the row headers in scalar registers and 10 KB
// of LDS: 566 k.  AGX_PGS_LV = 4 (the DEFAULT since round 6): agx_pgs_lvw.h, up to four rows with disjoint velocity slots per visit, one per 16-lane
// group (same bits); agx_pgs_lvs.h stays compiled in as its fallback and A/B partner (AGX_P_SOLVE_WIDE = 0).  -DAGX_PGS_LV=0: the register sweep of agx_pgs.h.  Emulator variants 'feeding_lv2' / 'feeding_reg' keep the C++ twins tested.
#ifndef AGX_PGS_LV
#define AGX_PGS_LV 4
#endif
constexpr bool LV_COMPILED = AGX_PGS_LV && HDR_WIDE;       // the `feeding` variant (Jaco, Panda); rows of at most 16 pairs are checked per environment (lv_eligible)
constexpr int LV_SOLVE_LDS_BYTES = 20480;                           // LDS of a solve launch of that variant: every row of an ordinary substep inside the window
constexpr int LV_HDR_WORDS = 8;                                     // invD, b, lo, hi | lam, off8, n, pack
constexpr int LV_H_LAM = 4, LV_H_LO = 2, LV_H_HI = 3;
constexpr int LV_DV = 0, LV_HDR = 128;                              // LDS words: dv[128], headers[8 R8], pairs[2 (win + 16)], dv slot addresses[(win + 16) / 2] (16 bit)

// LDS accessors of the hot loop.  On the device they take ABSOLUTE LDS byte addresses (32 bit, what a ds_read wants in its address register):
// the prologue folds the array bases into the row headers and the dv offsets, so that a visit adds nothing but its lane offset.  Explicit
// address spaces (a flat access would count against vmcnt AND lgkmcnt).  On the emulator an "address" is a byte offset from the LDS array.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float lv_v4 __attribute__((ext_vector_type(4)));
typedef float lv_v2 __attribute__((ext_vector_type(2)));
#define LV_LDS(T) __attribute__((address_space(3))) T*
#define LV_GLB(T) const __attribute__((address_space(1))) T*
AGX_DEV int lv_addr(const float* lds, const void* p) { (void)lds; return (int)(uintptr_t)(LV_LDS(const char))(const char*)p; }
AGX_DEV void lv_ld4(const float* lds, int a, float& x, float& y, float& z, float& w) { (void)lds; const lv_v4 v = *(LV_LDS(const lv_v4))(uintptr_t)a; x = v.x; y = v.y; z = v.z; w = v.w; }
AGX_DEV void lv_ld2(const float* lds, int a, float& x, float& y) { (void)lds; const lv_v2 v = *(LV_LDS(const lv_v2))(uintptr_t)a; x = v.x; y = v.y; }
AGX_DEV int lv_ld16(const float* lds, int a) { (void)lds; return (int)*(LV_LDS(const uint16_t))(uintptr_t)a; }
AGX_DEV float lv_ld1(const float* lds, int a) { (void)lds; return *(LV_LDS(const float))(uintptr_t)a; }
AGX_DEV void lv_st1(float* lds, int a, float v) { (void)lds; *(LV_LDS(float))(uintptr_t)a = v; }
AGX_DEV void lv_ld2g(const float* p, float& x, float& y) { const lv_v2 v = *(LV_GLB(lv_v2))p; x = v.x; y = v.y; }
#else
AGX_DEV int lv_addr(const float* lds, const void* p) { return (int)((const char*)p - (const char*)lds); }
AGX_DEV void lv_ld4(const float* lds, int a, float& x, float& y, float& z, float& w) { const float* p = (const float*)((const char*)lds + a); x = p[0]; y = p[1]; z = p[2]; w = p[3]; }
AGX_DEV void lv_ld2(const float* lds, int a, float& x, float& y) { const float* p = (const float*)((const char*)lds + a); x = p[0]; y = p[1]; }
AGX_DEV int lv_ld16(const float* lds, int a) { return (int)*(const uint16_t*)((const char*)lds + a); }
AGX_DEV float lv_ld1(const float* lds, int a) { return *(const float*)((const char*)lds + a); }
AGX_DEV void lv_st1(float* lds, int a, float v) { *(float*)((char*)lds + a) = v; }
AGX_DEV void lv_ld2g(const float* p, float& x, float& y) { x = p[0]; y = p[1]; }
#endif

struct LvLay { float* lds; int hdr_addr; const float* Eg; int dv_addr, rfar; };
// header of a row as the visit needs it.  pa: address of the row's pairs (LDS, absolute) or their byte offset in the scratch record (rows beyond
// the window); xa: address of the row's dv offsets (LDS), or the row's DoF ranges (a0 | na << 8 | b0 << 16 | nb << 24) beyond the window
struct LvHdr { float invD, b, lo, hi, lam; int pa, n, xa, addr; bool far; };
// this lane's entry of a row: J, B, address of its dv slot; on = the row has an entry k
struct LvEnt { float J, B; int ia; bool on; };

AGX_DEV void lv_hdr_load(const LvLay& Y, int row, LvHdr& h) {
  const int a = Y.hdr_addr + 4 * LV_HDR_WORDS * row;
  float l, o, n, x;
  lv_ld4(Y.lds, a, h.invD, h.b, h.lo, h.hi); lv_ld4(Y.lds, a + 16, l, o, n, x);
  h.lam = l; h.pa = __builtin_bit_cast(int, o); h.n = __builtin_bit_cast(int, n); h.xa = __builtin_bit_cast(int, x); h.addr = a; h.far = row >= Y.rfar;
}
AGX_DEV void lv_ent_load(const LvLay& Y, const LvHdr& h, int k, LvEnt& e) {
  e.on = k < h.n;
  if (!h.far) {                                                     // wave uniform: the row's pairs and dv offsets are in the LDS window
    lv_ld2(Y.lds, h.pa + 8 * k, e.J, e.B);
    e.ia = lv_ld16(Y.lds, h.xa + 2 * k);
  } else {                                                          // ... or stream from the scratch record; the dv slot from the row's two DoF ranges
    lv_ld2g((const float*)((const char*)Y.Eg + (unsigned)(h.pa + 8 * k)), e.J, e.B);
    const int a0 = h.xa & 255, na = (h.xa >> 8) & 255, b0 = (h.xa >> 16) & 255;
    e.ia = Y.dv_addr + 4 * (k < na ? a0 + k : b0 + k - na);
  }
}
// one visit in two halves: the gather is issued before the look-ahead loads of the step (LDS answers in order: what is requested
// first arrives first), then the dot product, the impulse update (identical in the 16 lanes) and the scatter.  Lane 0 keeps the row's impulse.
AGX_DEV float lv_gather(const LvLay& Y, const LvEnt& e) {
#if defined(AGX_LV_ABL_NOGATHER)
  return e.J;
#elif defined(__HIP_DEVICE_COMPILE__)
  return lv_ld1(Y.lds, e.ia);                                       // lanes beyond the row read a 16-bit address of some later row (or nothing: LDS reads beyond the allocation return 0); masked in lv_update
#else
  return e.on ? lv_ld1(Y.lds, e.ia) : 0.f;
#endif
}
AGX_DEV void lv_update(const LvLay& Y, const LvHdr& h, const LvEnt& e, float v, int k) {
  const float x = e.on ? e.J * v : 0.f;
  const float jdv = wave_sum16(x);
  const float nl = wave_clamp(h.lam + (h.b - jdv) * h.invD, h.lo, h.hi);
  const float dl = nl - h.lam;
#ifndef AGX_LV_ABL_NOLAMW        // AGX_LV_ABL_*: timing experiments (results meaningless), what each piece of a visit costs
  if (k == 0) lv_st1(Y.lds, h.addr + 4 * LV_H_LAM, nl);
#endif
#ifndef AGX_LV_ABL_NOSCATTER
  if (e.on) lv_st1(Y.lds, e.ia, v + e.B * dl);
#else
  if (e.on && v + e.B * dl == 12345.f) lv_st1(Y.lds, e.ia, 0.f);
#endif
  wave_fence();                                                     // the next visit gathers what this one scattered: program order inside one wavefront
}
#if defined(__HIP_DEVICE_COMPILE__) && AGX_PGS_LV == 2
AGX_DEV void lv_part_asm(const LvLay& Y, int lane, uint64_t mask, int base);
#endif
// the rows base + (set bits of m0, then 64 + set bits of m1), ascending
struct LvIt { uint64_t m0, m1; int base, last; };
AGX_DEV int lv_next(LvIt& it) {
  if (it.m0) { const int r = ffs64(it.m0); it.m0 &= it.m0 - 1ull; it.last = it.base + r; }
  else if (it.m1) { const int r = ffs64(it.m1); it.m1 &= it.m1 - 1ull; it.last = it.base + 64 + r; }
  return it.last;                                                   // exhausted: the look-ahead re-reads the last row (never used)
}
// One part of a sweep.  Software pipeline, written out four visits long so that the slots rotate without register moves: at visit t the
// header of visit t + 3 and the entry of visit t + 2 are requested before the arithmetic of visit t.  A row is visited once per part, so a
// header (with its impulse) read three visits early is current.
AGX_DEV void lv_part(const LvLay& Y, int lane, uint64_t m0, uint64_t m1, int base) {
  int nvis = popc64(m0) + popc64(m1);
  if (nvis == 0) return;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_LV_ABL_FULL_EXEC)
  if (lane < LV_G)                                                  // the sweeps run under EXEC = lanes 0..15 (the emulator's collectives need all 64 fibres)
#endif
  {
#if defined(__HIP_DEVICE_COMPILE__) && AGX_PGS_LV == 2
    // the assembly loop takes a 64-bit mask of rows inside the LDS window; rows beyond it (a suffix, Y.rfar) go through the C++ loop below
    { const uint64_t n0 = base >= Y.rfar ? 0ull : (base + 64 <= Y.rfar ? ~0ull : pgs_range_mask(0, Y.rfar - base));
      const uint64_t n1 = base + 64 >= Y.rfar ? 0ull : (base + 128 <= Y.rfar ? ~0ull : pgs_range_mask(0, Y.rfar - base - 64));
      if ((m0 & ~n0) == 0ull && (m1 & ~n1) == 0ull) {
        if (m0) lv_part_asm(Y, lane, m0, base);
        if (m1) lv_part_asm(Y, lane, m1, base + 64);
        m0 = 0ull; m1 = 0ull;
      } }
    nvis = popc64(m0) + popc64(m1);
#endif
    if (nvis > 0) {
    const int k = lane & (LV_G - 1);
    LvIt it; it.m0 = m0; it.m1 = m1; it.base = base; it.last = base;
    LvHdr h0, h1, h2, h3; LvEnt e0, e1, e2, e3;
    lv_hdr_load(Y, lv_next(it), h0); lv_hdr_load(Y, lv_next(it), h1); lv_hdr_load(Y, lv_next(it), h2);
    lv_ent_load(Y, h0, k, e0); lv_ent_load(Y, h1, k, e1);
    int t = 0;
#ifdef AGX_LV_ABL_NOHDR
#define LV_ABL_HDR(x) (void)lv_next(it);
#else
#define LV_ABL_HDR(x) x
#endif
#ifdef AGX_LV_ABL_NOENT
#define LV_ABL_ENT(x)
#else
#define LV_ABL_ENT(x) x
#endif
#define LV_STEP(HC, EC, HN3, HN2, EN2) { \
      const float v = lv_gather(Y, EC); \
      wave_fence(); \
      LV_ABL_HDR(lv_hdr_load(Y, lv_next(it), HN3);) \
      LV_ABL_ENT(lv_ent_load(Y, HN2, k, EN2);) \
      lv_update(Y, HC, EC, v, k); \
      if (++t == nvis) break; }
    for (;;) {
      LV_STEP(h0, e0, h3, h2, e2)
      LV_STEP(h1, e1, h0, h3, e3)
      LV_STEP(h2, e2, h1, h0, e0)
      LV_STEP(h3, e3, h2, h1, e1)
    }
#undef LV_STEP
    }
  }
  wave_fence();
}

#if defined(__HIP_DEVICE_COMPILE__) && AGX_PGS_LV == 2
// ---- the visit loop in gfx950 assembly (-DAGX_PGS_LV=2): the same visit as lv_gather / lv_update on the same LDS tables, 45 instructions instead of
// the compiler's ~110.  One 64-bit mask of rows whose pairs all lie inside the LDS window (the caller checks).  Header ring of three
// (visit t, t + 1, t + 2 in flight), entry ring of two; the body is written out six visits long so that both rings rotate without moves.
// LDS answers in order, so the waits are exact: at the top of a visit the two stores of the previous one may still be in flight (lgkmcnt 2),
// after the look-ahead loads the gather must be in (lgkmcnt 4).  Registers: v64..v87 headers, v88..v93 entries, v94..v99 temporaries,
// v100..v102 header addresses; s80..s91.
#define LVA_H0 "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v[64:67]", "v[68:71]", "v100"
// step(HC: invD,b,lo,hi,lam of the current visit + its header address | HN1: pa, n, xa of the next | HN2: two b128 targets + address register
//      of the one after | EC: J, B, ia, on-mask of the current | EN: pair target, ia, on-mask of the next)
#define LVA_STEP(C_INVD, C_B, C_LO, C_HI, C_LAM, C_HA, N1_PA, N1_N, N1_XA, N2_LO4, N2_HI4, N2_HA, EC_J, EC_B, EC_IA, EC_ON, EN_JB, EN_IA, EN_ON) \
  "s_waitcnt lgkmcnt(2)\n" \
  "ds_read_b32 v94, " EC_IA "\n" \
  "s_ff1_i32_b64 s82, s[80:81]\n" \
  "s_bitset0_b64 s[80:81], s82\n" \
  "s_lshl_b32 s83, s82, 5\n" \
  "s_add_u32 s83, s83, %[hdr]\n" \
  "v_mov_b32_e32 " N2_HA ", s83\n" \
  "v_add_u32_e32 v98, " N1_PA ", %[k8]\n" \
  "v_add_u32_e32 v99, " N1_XA ", %[k2]\n" \
  "ds_read_b64 " EN_JB ", v98\n" \
  "ds_read_u16 " EN_IA ", v99\n" \
  "v_cmp_lt_u32_e64 " EN_ON ", %[k], " N1_N "\n" \
  "ds_read_b128 " N2_LO4 ", " N2_HA "\n" \
  "ds_read_b128 " N2_HI4 ", " N2_HA " offset:16\n" \
  "s_waitcnt lgkmcnt(4)\n" \
  "v_mul_f32_e32 v95, " EC_J ", v94\n" \
  "v_cndmask_b32_e64 v95, 0, v95, " EC_ON "\n" \
  "s_nop 1\n" \
  "v_add_f32_dpp v95, v95, v95 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "s_nop 1\n" \
  "v_add_f32_dpp v95, v95, v95 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "s_nop 1\n" \
  "v_add_f32_dpp v95, v95, v95 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "s_nop 1\n" \
  "v_add_f32_dpp v95, v95, v95 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "v_sub_f32_e32 v96, " C_B ", v95\n" \
  "v_fma_f32 v96, " C_INVD ", v96, " C_LAM "\n" \
  "v_med3_f32 v96, v96, " C_LO ", " C_HI "\n" \
  "v_sub_f32_e32 v97, v96, " C_LAM "\n" \
  "v_fmac_f32_e32 v94, " EC_B ", v97\n" \
  "s_mov_b64 exec, 1\n" \
  "ds_write_b32 " C_HA ", v96 offset:16\n" \
  "s_mov_b64 exec, " EC_ON "\n" \
  "ds_write_b32 " EC_IA ", v94\n" \
  "s_mov_b64 exec, 0xffff\n" \
  "s_sub_u32 s84, s84, 1\n" \
  "s_cbranch_scc1 9f\n" \
  "s_cmp_eq_u32 s84, 0\n" \
  "s_cbranch_scc1 9f\n"
// header slots: A = v64..v71 (address v100), B = v72..v79 (v101), C = v80..v87 (v102); entry slots: P = v[88:89] + v90 (mask s[86:87]), Q = v[92:93] + v91 (s[88:89]); register tuples must be even aligned on gfx950
#define LVA_HCUR_A "v64", "v65", "v66", "v67", "v68", "v100"
#define LVA_HCUR_B "v72", "v73", "v74", "v75", "v76", "v101"
#define LVA_HCUR_C "v80", "v81", "v82", "v83", "v84", "v102"
#define LVA_HN1_A "v69", "v70", "v71"
#define LVA_HN1_B "v77", "v78", "v79"
#define LVA_HN1_C "v85", "v86", "v87"
#define LVA_HN2_A "v[64:67]", "v[68:71]", "v100"
#define LVA_HN2_B "v[72:75]", "v[76:79]", "v101"
#define LVA_HN2_C "v[80:83]", "v[84:87]", "v102"
#define LVA_EC_P "v88", "v89", "v90", "s[86:87]"
#define LVA_EC_Q "v92", "v93", "v91", "s[88:89]"
#define LVA_EN_P "v[88:89]", "v90", "s[86:87]"
#define LVA_EN_Q "v[92:93]", "v91", "s[88:89]"
#define LVA_S2(a, b, c, d, e) LVA_STEP a, b, c, d, e)
#define LVA_CALL(HC, HN1, HN2, EC, EN) LVA_APPLY(LVA_STEP, HC, HN1, HN2, EC, EN)
#define LVA_APPLY(M, ...) M(__VA_ARGS__)
AGX_DEV void lv_part_asm(const LvLay& Y, int lane, uint64_t mask, int base) {
  const int k = lane, k8 = 8 * lane, k2 = 2 * lane;
  const int hdr = Y.hdr_addr + 4 * LV_HDR_WORDS * base;
  const int nvis = popc64(mask);
  asm volatile(
    "s_mov_b64 s[80:81], %[mask]\n"
    "s_mov_b32 s84, %[nvis]\n"
    // prime: headers of visits 0 and 1, entry of visit 0
    "s_ff1_i32_b64 s82, s[80:81]\n" "s_bitset0_b64 s[80:81], s82\n" "s_lshl_b32 s83, s82, 5\n" "s_add_u32 s83, s83, %[hdr]\n" "v_mov_b32_e32 v100, s83\n"
    "ds_read_b128 v[64:67], v100\n" "ds_read_b128 v[68:71], v100 offset:16\n"
    "s_ff1_i32_b64 s82, s[80:81]\n" "s_bitset0_b64 s[80:81], s82\n" "s_lshl_b32 s83, s82, 5\n" "s_add_u32 s83, s83, %[hdr]\n" "v_mov_b32_e32 v101, s83\n"
    "ds_read_b128 v[72:75], v101\n" "ds_read_b128 v[76:79], v101 offset:16\n"
    "s_waitcnt lgkmcnt(0)\n"
    "v_add_u32_e32 v98, v69, %[k8]\n" "v_add_u32_e32 v99, v71, %[k2]\n"
    "ds_read_b64 v[88:89], v98\n" "ds_read_u16 v90, v99\n"
    "v_cmp_lt_u32_e64 s[86:87], %[k], v70\n"
    "s_waitcnt lgkmcnt(0)\n"
    "8:\n"
    LVA_CALL(LVA_HCUR_A, LVA_HN1_B, LVA_HN2_C, LVA_EC_P, LVA_EN_Q)
    LVA_CALL(LVA_HCUR_B, LVA_HN1_C, LVA_HN2_A, LVA_EC_Q, LVA_EN_P)
    LVA_CALL(LVA_HCUR_C, LVA_HN1_A, LVA_HN2_B, LVA_EC_P, LVA_EN_Q)
    LVA_CALL(LVA_HCUR_A, LVA_HN1_B, LVA_HN2_C, LVA_EC_Q, LVA_EN_P)
    LVA_CALL(LVA_HCUR_B, LVA_HN1_C, LVA_HN2_A, LVA_EC_P, LVA_EN_Q)
    LVA_CALL(LVA_HCUR_C, LVA_HN1_A, LVA_HN2_B, LVA_EC_Q, LVA_EN_P)
    "s_branch 8b\n"
    "9:\n"
    "s_waitcnt lgkmcnt(0)\n"
    :
    : [mask] "s"(mask), [nvis] "s"(nvis), [hdr] "s"(hdr), [k] "v"(k), [k8] "v"(k8), [k2] "v"(k2)
    : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87",
      "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102",
      "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "vcc", "scc", "memory");
}
#endif

// may this environment take the row-local sweep?  (wave uniform)
AGX_DEV bool lv_eligible(const Ctx& c) {
  return c.ndof <= LV_G && c.nrobot + 6 <= LV_G && c.nhdof + 6 <= LV_G && c.nv <= 128 && !(PRM(c, AGX_P_WARMSTART) > 0.f) && c.nrows > 0;
}

// lds: the solve kernel's LDS, lds_words long (state copy and velocity vector are loaded by solve_tail() afterwards: all of it is free here)
AGX_DEV void pgs_lv(Ctx& c, float* lds, int lds_words, float& dv0, float& dv1) {
  const int lane = c.lane, iters = (int)PRM(c, AGX_P_NITER);
  const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc, R = c.nrows;       // rows: [0,nnc) non-contact, [nnc,nA) normals, then nc friction rows per direction
  const int R8 = (R + 7) & ~7;
  float* HDR = lds + LV_HDR;
  int win = ((lds_words - LV_HDR - LV_HDR_WORDS * R8) * 2 / 5 - LV_G) & ~3; if (win < 0) win = 0; if (win > c.nent) win = (c.nent + 3) & ~3;
#ifdef AGX_LV_WINDOW_CAP            // tests: a small window, so that the path for rows beyond it runs on ordinary scenes
  if (win > AGX_LV_WINDOW_CAP) win = AGX_LV_WINDOW_CAP;
#endif
  float* PAIRS = HDR + LV_HDR_WORDS * R8; uint16_t* IDX = (uint16_t*)(PAIRS + 2 * (win + LV_G));
  LvLay Y; Y.lds = lds; Y.hdr_addr = lv_addr(lds, HDR); Y.Eg = c.E; Y.dv_addr = lv_addr(lds, lds + LV_DV);
  const int pairs_addr = lv_addr(lds, PAIRS), idx_addr = lv_addr(lds, IDX);
  // ---- prologue (all 64 lanes): velocity deltas, headers, the window of pairs, their dv slots
  lds[LV_DV + lane] = 0.f; lds[LV_DV + 64 + lane] = 0.f;
  int far_first = R;
  for (int r = lane; r < R8; r += 64) {
    const bool ok = r < R;
    const float* H = c.H + HDR_STRIDE * (ok ? r : 0); const int* Hi = (const int*)H;
    const float invD = ok ? H[H_INVD] : 0.f;
    const bool live = ok && invD != 0.f, fric = r >= nA;          // friction bounds are rewritten before every friction part
    const int pack = ok ? ((const int*)hx_row(c.H, r))[H_PACK] : 0, off = ok ? (Hi[H_OFF] & 0x7fffffff) : 0;
    const int na = (pack >> 8) & 255, nb = (int)((unsigned)pack >> 24), n = na + nb;
    const bool near = ok && off + n <= win;
    float* o = HDR + LV_HDR_WORDS * r; int* oi = (int*)o;
    o[0] = invD; o[1] = ok ? H[H_B] : 0.f; o[LV_H_LO] = (live && !fric) ? H[H_LO] : 0.f; o[LV_H_HI] = (live && !fric) ? H[H_HI] : 0.f;
    o[LV_H_LAM] = 0.f; oi[5] = near ? pairs_addr + 8 * off : 8 * off; oi[6] = n; oi[7] = near ? idx_addr + 2 * off : pack;
    if (ok && !near && r < far_first) far_first = r;
    if (near) {
      const int a0 = pack & 255, b0 = (pack >> 16) & 255;
      for (int j = 0; j < LV_G; j++) if (j < n) IDX[off + j] = (uint16_t)(Y.dv_addr + 4 * (j < na ? a0 + j : b0 + j - na));
    }
  }
  Y.rfar = (int)wave_min((float)far_first);                         // offsets grow with the row index: the rows beyond the window are a suffix
  { const f2* src = (const f2*)c.E; f2* dst = (f2*)PAIRS; const int np = win < c.nent ? win : c.nent;
    for (int q = lane; q < np; q += 64) dst[q] = src[q]; }
  // friction coefficient of this lane's contact (0 for a row without effective mass: pinned at zero impulse)
  const bool two_dirs = R > nA + nc;
  float mu1 = 0.f, mu2 = 0.f;
  if (lane < nc) {
    const float* H = c.H + HDR_STRIDE * (nA + lane); mu1 = H[H_INVD] != 0.f ? hx_row(c.H, nA + lane)[H_MU] : 0.f;
    if (two_dirs) { const float* H2 = c.H + HDR_STRIDE * (nA + nc + lane); mu2 = H2[H_INVD] != 0.f ? hx_row(c.H, nA + nc + lane)[H_MU] : 0.f; }
  }
  wave_sync();
  const int a0n = nA < 64 ? nA : 64, a1n = nA - 64;
  const uint64_t rowsA0 = pgs_range_mask(0, a0n), rowsA1 = a1n > 0 ? pgs_range_mask(0, a1n) : 0ull;
  const int K = noop_period(c);                                     // the no-op re-test rule: see pgs()
  uint64_t skip0 = 0ull, skip1 = 0ull;
  const float* LAM = HDR + LV_H_LAM;
  for (int it = 0; it < iters; it++) {
    const bool retest = K > 0 && it % K == 0, use = K > 0 && !retest;
    float bef0 = 0.f, bef1 = 0.f;
    if (retest) { if (lane < nA) bef0 = LAM[LV_HDR_WORDS * lane]; if (64 + lane < nA) bef1 = LAM[LV_HDR_WORDS * (64 + lane)]; }
    lv_part(Y, lane, rowsA0 & ~(use ? skip0 : 0ull), rowsA1 & ~(use ? skip1 : 0ull), 0);
    if (retest) {
      const float af0 = lane < nA ? LAM[LV_HDR_WORDS * lane] : 0.f, af1 = 64 + lane < nA ? LAM[LV_HDR_WORDS * (64 + lane)] : 0.f;
      skip0 = wave_ballot(af0 == bef0); skip1 = wave_ballot(af1 == bef1);
    }
    for (int dir = 0; dir < (two_dirs ? 2 : 1); dir++) {
      // friction rows (lane = contact): bounds from the normal impulses as this sweep's normal pass left them; a row whose normal
      // impulse and own impulse are both zero is an exact no-op and is not visited
      const int f0 = nA + dir * nc;
      float ln = 0.f, lf = 0.f;
      if (lane < nc) {
        ln = LAM[LV_HDR_WORDS * (nnc + lane)]; lf = LAM[LV_HDR_WORDS * (f0 + lane)];
        const float hi = (dir ? mu2 : mu1) * ln;
        HDR[LV_HDR_WORDS * (f0 + lane) + LV_H_LO] = -hi; HDR[LV_HDR_WORDS * (f0 + lane) + LV_H_HI] = hi;
      }
      const uint64_t todo = wave_ballot(lane < nc && (ln != 0.f || lf != 0.f));
      wave_fence();
      lv_part(Y, lane, todo, 0ull, f0);
    }
  }
  wave_sync();
  // velocity deltas back to their DoF lanes; solved normal impulses -> contact records (what getContactPoints reports until the next step)
  dv0 = lds[LV_DV + lane]; dv1 = lds[LV_DV + 64 + lane];
  if (lane < nc) c.gcon[CON_STRIDE * lane + C_LAM] = LAM[LV_HDR_WORDS * (nnc + lane)];
  wave_sync();
}

}  // namespace agx
