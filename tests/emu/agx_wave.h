// CPU wave64 emulator for the stepper kernel sources (TEST HARNESS ONLY -- never part of the
// product library).  The same agx_step.h / agx_gjk.h that hipcc compiles for gfx950 is compiled
// here with g++; the 64 lanes of one wavefront run as cooperative fibers (ucontext) and every
// cross-lane primitive of csrc/agx_wave.h is implemented as a rendezvous of all 64 lanes.
// A non-uniform collective (a kernel bug on the GPU) shows up here as a deadlock assertion.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <math.h>

#define AGX_DEV static inline
#define AGX_DEV_NOINLINE static
#define AGX_WAVE 64

namespace emu {
struct Wave {
  ucontext_t main_ctx, ctx[64];
  char* stacks[64];
  bool done[64];
  int cur;
  int arrived; unsigned gen;
  uint32_t stage[2][64];
};
extern Wave* W;
inline void yield() { swapcontext(&W->ctx[W->cur], &W->main_ctx); }
// all 64 lanes deposit a word, then read the full vector
inline const uint32_t* exchange(uint32_t mine) {
  Wave* w = W; const int lane = w->cur; const unsigned g = w->gen;
  w->stage[g & 1][lane] = mine;
  if (++w->arrived == 64) { w->arrived = 0; w->gen++; }
  else while (w->gen == g) yield();
  return w->stage[g & 1];
}
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
}  // namespace emu

AGX_DEV int wave_lane() { return emu::W->cur; }
AGX_DEV void wave_sync() { emu::exchange(0); }
AGX_DEV void wave_fence() { emu::exchange(0); }
AGX_DEV float wave_sum(float x) {
  const uint32_t* s = emu::exchange(emu::f2u(x));
  // same association as the DPP tree of csrc/agx_wave.h: quads, rows of 16, then rows combined
  float q[16]; for (int i = 0; i < 16; i++) q[i] = (emu::u2f(s[4 * i + 1]) + emu::u2f(s[4 * i])) + (emu::u2f(s[4 * i + 3]) + emu::u2f(s[4 * i + 2]));
  float r[4]; for (int i = 0; i < 4; i++) r[i] = (q[4 * i + 3] + q[4 * i + 2]) + (q[4 * i + 1] + q[4 * i]);
  return (r[3] + r[2]) + (r[1] + r[0]);
}
AGX_DEV float wave_min(float x) { const uint32_t* s = emu::exchange(emu::f2u(x)); float m = emu::u2f(s[0]); for (int i = 1; i < 64; i++) m = fminf(m, emu::u2f(s[i])); return m; }
AGX_DEV float wave_max(float x) { return -wave_min(-x); }
AGX_DEV int wave_sum_i(int x) { const uint32_t* s = emu::exchange((uint32_t)x); int t = 0; for (int i = 0; i < 64; i++) t += (int)s[i]; return t; }
AGX_DEV uint64_t wave_ballot(bool p) { const uint32_t* s = emu::exchange(p ? 1u : 0u); uint64_t m = 0; for (int i = 0; i < 64; i++) if (s[i]) m |= 1ull << i; return m; }
AGX_DEV bool wave_any(bool p) { return wave_ballot(p) != 0ull; }
AGX_DEV float wave_shfl(float x, int src) { const uint32_t* s = emu::exchange(emu::f2u(x)); return emu::u2f(s[src & 63]); }
AGX_DEV int wave_shfl_i(int x, int src) { const uint32_t* s = emu::exchange((uint32_t)x); return (int)s[src & 63]; }
AGX_DEV float wave_bcast(float x, int src) { return wave_shfl(x, src); }
AGX_DEV int wave_bcast_i(int x, int src) { return wave_shfl_i(x, src); }
AGX_DEV int wave_rank(uint64_t mask) { return __builtin_popcountll(mask & ((1ull << emu::W->cur) - 1ull)); }
AGX_DEV int popc64(uint64_t m) { return __builtin_popcountll(m); }
AGX_DEV int ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : -1; }
AGX_DEV int clz64(uint64_t m) { return m ? __builtin_clzll(m) : 64; }
AGX_DEV int wave_scan_excl(int x) { const uint32_t* s = emu::exchange((uint32_t)x); int t = 0; for (int i = 0; i < emu::W->cur; i++) t += (int)s[i]; return t; }
AGX_DEV long long wave_clock() { return 0; }
AGX_DEV float wave_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
AGX_DEV void wave_opaque(float&) {}
AGX_DEV int wave_uniform(int x) { return x; }
AGX_DEV float wave_rsqrt(float x) { return 1.0f / sqrtf(x); }

// the 32 x 32 x 2 f32 matrix-core step of csrc/agx_wave.h: the same lane -> element maps, an fmaf chain over k = 0, 1
struct Acc16 { float v[16]; };
AGX_DEV void acc16_zero(Acc16& c) { for (int k = 0; k < 16; k++) c.v[k] = 0.f; }
AGX_DEV float acc16_get(const Acc16& c, int k) { return c.v[k]; }
AGX_DEV void wave_mfma_32x32x2(float a, float b, Acc16& c) {
  float A[64], B[64];
  { const uint32_t* s = emu::exchange(emu::f2u(a)); for (int i = 0; i < 64; i++) A[i] = emu::u2f(s[i]); }
  { const uint32_t* s = emu::exchange(emu::f2u(b)); for (int i = 0; i < 64; i++) B[i] = emu::u2f(s[i]); }
  const int l = emu::W->cur, j = l & 31, hb = l >> 5;
  for (int v = 0; v < 16; v++) { const int i = (v & 3) + 8 * (v >> 2) + 4 * hb; c.v[v] = fmaf(A[i + 32], B[j + 32], fmaf(A[i], B[j], c.v[v])); }
}

// ---- 16-lane group primitives of the packed solve kernel (csrc/agx_pgs4.h) ----
// same association as the DPP butterfly of the device code: pairs, quads, halves of the 16-lane row, the row
AGX_DEV float g16_sum(float x) {
  const uint32_t* s = emu::exchange(emu::f2u(x));
  const int b = emu::W->cur & ~15;
  float q[4]; for (int i = 0; i < 4; i++) q[i] = (emu::u2f(s[b + 4 * i]) + emu::u2f(s[b + 4 * i + 1])) + (emu::u2f(s[b + 4 * i + 2]) + emu::u2f(s[b + 4 * i + 3]));
  return (q[0] + q[1]) + (q[2] + q[3]);
}
AGX_DEV float wave_sum16(float x) { return g16_sum(x); }
AGX_DEV uint32_t g16_ballot(bool p, int group) { return (uint32_t)(wave_ballot(p) >> (16 * group)) & 0xffffu; }
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
