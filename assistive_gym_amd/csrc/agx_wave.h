// agx_wave.h -- wave64 primitives for gfx950 (CDNA4).  One wavefront = one environment.
//
// Everything cross-lane in the stepper goes through this header: DPP reductions (no LDS
// traffic), ballots, uniform broadcasts.  Workgroups are a single wavefront, so wave_sync() is a
// plain LDS fence for the wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AGX_DEV __device__ __forceinline__
#define AGX_DEV_NOINLINE __device__ __noinline__
#define AGX_WAVE 64

AGX_DEV int wave_lane() { return (int)(threadIdx.x & 63u); }
#ifdef AGX_WAVE_SYNC_LDS_ONLY   // timing experiment only (NOT safe: some sync points order global scratch traffic between lanes): how much of the
                                // kernels' time is the workgroup fence of __syncthreads() waiting for outstanding global loads / stores?
                                // Measured (same box, profiles/r04/r04t_ab_feeding_lds_only_wave_sync.txt): none -- 473.3 vs 473.2 k env-steps/s.
AGX_DEV void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
AGX_DEV void wave_sync() { __syncthreads(); }
#endif
// orders this wavefront's own LDS accesses in the compiler; the hardware executes a wavefront's LDS instructions in order, so no wait
AGX_DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// DPP controls (cdna4 ISA: DPP_CTRL)
#define AGX_DPP_QUAD_1032 0xb1
#define AGX_DPP_QUAD_2301 0x4e
#define AGX_DPP_ROW_SHR4 0x114
#define AGX_DPP_ROW_SHR8 0x118
#define AGX_DPP_ROW_BCAST15 0x142
#define AGX_DPP_ROW_BCAST31 0x143

template <int CTRL>
AGX_DEV float dpp_mov(float identity, float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
AGX_DEV int dpp_mov_i(int identity, int x) { return __builtin_amdgcn_update_dpp(identity, x, CTRL, 0xf, 0xf, false); }

// sum over the 64 lanes, result uniform in every lane
AGX_DEV float wave_sum(float x) {
  x += dpp_mov<AGX_DPP_QUAD_1032>(0.f, x);
  x += dpp_mov<AGX_DPP_QUAD_2301>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_SHR4>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_SHR8>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_BCAST15>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_BCAST31>(0.f, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
AGX_DEV float wave_min(float x) {
  const float id = 3.0e38f;
  x = fminf(x, dpp_mov<AGX_DPP_QUAD_1032>(id, x));
  x = fminf(x, dpp_mov<AGX_DPP_QUAD_2301>(id, x));
  x = fminf(x, dpp_mov<AGX_DPP_ROW_SHR4>(id, x));
  x = fminf(x, dpp_mov<AGX_DPP_ROW_SHR8>(id, x));
  x = fminf(x, dpp_mov<AGX_DPP_ROW_BCAST15>(id, x));
  x = fminf(x, dpp_mov<AGX_DPP_ROW_BCAST31>(id, x));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
AGX_DEV float wave_max(float x) { return -wave_min(-x); }
AGX_DEV int wave_sum_i(int x) {
  x += dpp_mov_i<AGX_DPP_QUAD_1032>(0, x);
  x += dpp_mov_i<AGX_DPP_QUAD_2301>(0, x);
  x += dpp_mov_i<AGX_DPP_ROW_SHR4>(0, x);
  x += dpp_mov_i<AGX_DPP_ROW_SHR8>(0, x);
  x += dpp_mov_i<AGX_DPP_ROW_BCAST15>(0, x);
  x += dpp_mov_i<AGX_DPP_ROW_BCAST31>(0, x);
  return __builtin_amdgcn_readlane(x, 63);
}
// sum over the 16 lanes of this lane's DPP row as an xor butterfly (1, 2, half mirror, mirror): every lane adds the same two numbers at
// every step, so the result is bitwise the same in all 16 lanes -- no broadcast afterwards (the row-local sweep, agx_pgs_lv.h)
#define AGX_DPP_ROW_HALF_MIRROR 0x141
#define AGX_DPP_ROW_MIRROR 0x140
AGX_DEV float wave_sum16(float x) {
  x += dpp_mov<AGX_DPP_QUAD_1032>(0.f, x);
  x += dpp_mov<AGX_DPP_QUAD_2301>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_HALF_MIRROR>(0.f, x);
  x += dpp_mov<AGX_DPP_ROW_MIRROR>(0.f, x);
  return x;
}
AGX_DEV uint64_t wave_ballot(bool p) { return __ballot(p); }
AGX_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }
// value of lane `src` (src may differ per lane)
AGX_DEV float wave_shfl(float x, int src) { return __shfl(x, src, 64); }
AGX_DEV int wave_shfl_i(int x, int src) { return __shfl(x, src, 64); }
// value of lane `src`, src wave-uniform
AGX_DEV float wave_bcast(float x, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); }
AGX_DEV int wave_bcast_i(int x, int src) { return __builtin_amdgcn_readlane(x, src); }
// number of set bits of `mask` strictly below this lane
AGX_DEV int wave_rank(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
AGX_DEV int popc64(uint64_t m) { return __popcll(m); }
AGX_DEV int ffs64(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }
AGX_DEV int clz64(uint64_t m) { return __clzll((long long)m); }      // leading zero bits (m != 0)
// exclusive prefix sum over lanes (Hillis-Steele on ds_bpermute; used a few times per substep only)
AGX_DEV int wave_scan_excl(int x) {
  const int lane = wave_lane();
  int incl = x;
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  return incl - x;
}
// 1 / sqrt(x), v_rsq_f32 (1 ulp)
AGX_DEV float wave_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
// shader clock (s_memtime), for the per-phase cycle counters of the debug path
AGX_DEV long long wave_clock() { return (long long)__builtin_readcyclecounter(); }
// clamp to [lo, hi] (lo <= hi) in one v_med3_f32
AGX_DEV float wave_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// a value known to be the same in every lane, moved to a scalar register
AGX_DEV int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// optimisation barrier on one register value (no instruction is emitted)
AGX_DEV void wave_opaque(float& x) { asm volatile("" : "+v"(x)); }
// 32 x 32 x 2 f32 matrix-core step: c += a_mat (32 x 2) * b_mat (2 x 32), exact f32.  Lane l supplies a_mat[l & 31][l >> 5] and
// b_mat[l >> 5][l & 31]; register v of lane l holds c[(v & 3) + 8 (v >> 2) + 4 (l >> 5)][l & 31] (cdna_hip_programming.md, fragment layout)
typedef float agx_f32x16 __attribute__((ext_vector_type(16)));
struct Acc16 { agx_f32x16 v; };
AGX_DEV void acc16_zero(Acc16& c) { for (int k = 0; k < 16; k++) c.v[k] = 0.f; }
AGX_DEV float acc16_get(const Acc16& c, int k) { return c.v[k]; }
AGX_DEV void wave_mfma_32x32x2(float a, float b, Acc16& c) { c.v = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c.v, 0, 0, 0); }

