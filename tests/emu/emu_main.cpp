// Host harness that runs the product kernel sources (csrc/agx_step.h) for ONE environment on the
// CPU wave emulator.  Built by tests/emu_lib.py into tests/emu/libagx_emu.so.  Test-only.
#include "agx_wave.h"
#ifdef AGX_EMU_TRACE_GJK      // tests/diag/narrowphase_passes.py: per gjk_distance call and lane (call, lane, iterations, |A|, |B|, box)
extern "C" { int g_gjk_trace[1 << 22]; int g_gjk_n = 0, g_gjk_call = 0; }
#define AGX_TRACE_GJK(has, iters, na, nb, box, far_out, dist, far) { if (wave_lane() == 0) g_gjk_call++; wave_sync(); \
  if ((has) && g_gjk_n + 9 <= (1 << 22)) { int* t = g_gjk_trace + g_gjk_n; g_gjk_n += 9; t[0] = g_gjk_call; t[1] = wave_lane(); t[2] = (iters); t[3] = (na); t[4] = (nb); t[5] = (box) ? 1 : 0; \
    t[6] = (far_out) ? 1 : 0; t[7] = (int)((dist) * 1e6f); t[8] = (int)((far) * 1e6f); } }
#endif
#ifdef AGX_EMU_TRACE_SCHED    // tests/diag/solve_schedule_study.py: row DoF masks and per-sweep visit masks of the row-local sweep
extern "C" { int g_sched_trace[1 << 24]; int g_sched_n = 0; }
#endif
#include "agx_step.h"
#include "agx_water.h"      // not yet part of a kernel variant: the source is checked here against the oracle first
#include <functional>
#include <cstdio>

namespace emu { Wave* W = nullptr; }

static std::function<void(int)> g_body;
static void fiber_entry() {
  const int lane = emu::W->cur;
  g_body(lane);
  emu::W->done[lane] = true;
  swapcontext(&emu::W->ctx[lane], &emu::W->main_ctx);
}
// one "kernel launch" of one workgroup: 64 fresh fibers, LDS filled with garbage (the GPU does not
// clear LDS between workgroups, and nothing may rely on it)
static int run_wave(float* lds, size_t lds_words, std::function<void(int)> body) {
  static emu::Wave wave;
  emu::W = &wave;
  memset(&wave, 0, sizeof wave);
  for (size_t k = 0; k < lds_words; k++) { uint32_t g = 0x7fc00000u | (uint32_t)(k * 2654435761u >> 10); memcpy(&lds[k], &g, 4); }
  g_body = body;
  const size_t STK = 1 << 20;
  for (int l = 0; l < 64; l++) {
    wave.stacks[l] = (char*)malloc(STK);
    getcontext(&wave.ctx[l]);
    wave.ctx[l].uc_stack.ss_sp = wave.stacks[l]; wave.ctx[l].uc_stack.ss_size = STK; wave.ctx[l].uc_link = &wave.main_ctx;
    makecontext(&wave.ctx[l], fiber_entry, 0);
  }
  int rc = 0;
  for (unsigned long spins = 0;; spins++) {
    int remaining = 0; unsigned gen_before = wave.gen; int arrived_before = wave.arrived;
    for (int l = 0; l < 64; l++) if (!wave.done[l]) { remaining++; wave.cur = l; swapcontext(&wave.main_ctx, &wave.ctx[l]); }
    if (!remaining) break;
    int rem2 = 0; for (int l = 0; l < 64; l++) if (!wave.done[l]) rem2++;
    // a sweep without progress while some lanes have finished: the lanes disagree on a collective
    if (wave.gen == gen_before && wave.arrived == arrived_before && rem2 == remaining && rem2 != 64 && spins > 4) {
      fprintf(stderr, "agx_emu: lanes diverged around a collective (%d lanes finished early)\n", 64 - rem2); rc = -1; break;
    }
  }
  for (int l = 0; l < 64; l++) free(wave.stacks[l]);
  return rc;
}

// the scratch record of agx_emu_run's environment persists between calls, like the device's (the warm-start memory of AGX_P_WARMSTART
// lives in it); agx_emu_forget_warm() = what agx_set_state / the resets do to it
static float g_scratch[agx::SCR_WORDS];
extern "C" void agx_emu_forget_warm() { ((int*)g_scratch)[agx::SCR_O_META + agx::META_NWARM] = 0; ((int*)g_scratch)[agx::SCR_O_META + agx::META_NMAN] = 0; }
// test hooks of the persistent manifold: the cache of the emulated environment's scratch record, rows as in oracle_lib.Oracle.manifold_get
extern "C" int agx_emu_manifold_get(double* out, int max_out) {
  int n = ((int*)g_scratch)[agx::SCR_O_META + agx::META_NMAN]; if (n > max_out) n = max_out;
  for (int p = 0; p < n; p++) { const float* q = g_scratch + agx::SCR_O_MAN + agx::MP_STRIDE * p; double* o = out + 12 * p; const int key = ((const int*)q)[agx::MP_KEY];
    o[0] = key & 511; o[1] = (key >> 9) & 511; for (int k = 0; k < 3; k++) { o[2 + k] = q[agx::MP_LA + k]; o[5 + k] = q[agx::MP_LB + k]; o[8 + k] = q[agx::MP_N + k]; } o[11] = q[agx::MP_MU]; }
  return n;
}
extern "C" void agx_emu_manifold_set(const double* in, int n) {
  if (n > agx::MAX_CON) n = agx::MAX_CON;
  ((int*)g_scratch)[agx::SCR_O_META + agx::META_NMAN] = n;
  for (int p = 0; p < n; p++) { float* q = g_scratch + agx::SCR_O_MAN + agx::MP_STRIDE * p; const double* o = in + 12 * p;
    ((int*)q)[agx::MP_KEY] = (int)o[0] | ((int)o[1] << 9); for (int k = 0; k < 3; k++) { q[agx::MP_LA + k] = (float)o[2 + k]; q[agx::MP_LB + k] = (float)o[5 + k]; q[agx::MP_N + k] = (float)o[8 + k]; } q[agx::MP_DIST] = 0.f; q[agx::MP_MU] = (float)o[11]; }
}
// LDS of a solve launch, as libagx sizes it (agx_kernels.hip, g_solve_lds_bytes)
constexpr int EMU_SOLVE_WORDS = agx::LVS_COMPILED && agx::LVS_SOLVE_LDS_BYTES / 4 > agx::LDS_SOLVE_WORDS ? agx::LVS_SOLVE_LDS_BYTES / 4 : agx::LDS_SOLVE_WORDS;
extern "C" int agx_emu_run(const uint32_t* blob, float* state, const float* action, float* obs, float* reward, uint8_t* done,
                           float* info, float* debug, int mode, int nsettle) {
  static float lds[agx::LDS_WORDS > EMU_SOLVE_WORDS ? agx::LDS_WORDS : EMU_SOLVE_WORDS];
  float* const scratch = g_scratch;
  const int frame_skip = (int)((const float*)blob)[((const int*)blob)[AGX_H_OFF_PARAMS] + AGX_P_FRAME_SKIP];
  int rc = 0;
  if (mode == 2) return run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_observe(blob, state, obs, lds, lane); });
  const int sim_sub = ((const int*)blob)[AGX_H_SIM_SUBSTEPS] > 1 ? ((const int*)blob)[AGX_H_SIM_SUBSTEPS] : 1;   // internal substeps per stepSimulation
  const int nsub = (mode == 1 ? nsettle : frame_skip) * sim_sub;
  for (int k = 0; k < nsub && !rc; k++) {
    const float* act = (mode == 0 && k == 0) ? action : nullptr; float* dbg = (k == 0) ? debug : nullptr;
    rc = run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_build<true>(blob, state, act, scratch, dbg, lds, lane); });
    const int ph = mode == 1 ? (k | AGX_PHASE_SETTLE) : k;
    if (!rc && getenv("AGX_EMU_PRINT_META")) { const int* m = (const int*)(scratch + agx::SCR_O_META); fprintf(stderr, "meta: rows %d nnc %d contacts %d pairs %d\n", m[agx::META_NROWS], m[agx::META_NNC], m[agx::META_NCON], m[agx::META_NENT]); }
    if (!rc) rc = run_wave(lds, EMU_SOLVE_WORDS, [&](int lane) { agx::env_solve(blob, state, scratch, dbg, lds, lane, ph, EMU_SOLVE_WORDS); });
  }
  if (mode == 0 && !rc) rc = run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_finish(blob, state, action, scratch, obs, reward, done, info, lds, lane); });
  return rc;
}
#if AGX_HAS_SAMPLER
// settled (may be null): the rag-doll model's settled state record of this environment (bed bathing, AGX_X_FLAGS bit 4)
// fell (may be null): the fall model's record of this environment after the arm's fall (arm manipulation, AGX_X_FLAGS bit 7)
extern "C" int agx_emu_sample(const uint32_t* blob, float* state, uint64_t seed, int impairment_mode, int gender_mode, float* info4, const float* settled, const float* fell) {
  // the same schedule as libagx's launch_sample: sample, then AGX_X_COLLISION_TRIES rounds of [build, verdict, re-sample from the next restart]
  static float lds[agx::LDS_WORDS > agx::LDS_SOLVE_WORDS ? agx::LDS_WORDS : agx::LDS_SOLVE_WORDS];
  static float scratch[agx::SCR_WORDS];
  int chosen = -1, first = 0;
  int rc = run_wave(lds, 64, [&](int lane) { int r = agx::env_sample(blob, state, (uint32_t)seed, (uint32_t)(seed >> 32), impairment_mode, gender_mode, info4, lane, 0, settled, fell); if (lane == 0) chosen = r; });
  const int tries = ((const int*)blob)[((const int*)blob)[AGX_H_OFF_RESET] + AGX_X_COLLISION_TRIES];
  for (int t = 0; t < tries && !rc && chosen >= 0; t++) {
    rc = run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_build<true>(blob, state, nullptr, scratch, nullptr, lds, lane); });
    bool again = false;
    if (!rc) rc = run_wave(lds, 64, [&](int lane) { bool a = agx::reset_collides(blob, scratch, lane); if (lane == 0) again = a; });
    if (!again) break;
    first = chosen + 1;
    if (!rc) rc = run_wave(lds, 64, [&](int lane) { int r = agx::env_sample(blob, state, (uint32_t)seed, (uint32_t)(seed >> 32), impairment_mode, gender_mode, info4, lane, first, settled, fell); if (lane == 0) chosen = r; });
  }
  return rc;
}
#endif
// agx_check_collisions: the build pass on the state as it is, then the flags (returns them, -1 on divergent control flow)
extern "C" int agx_emu_check_collisions(const uint32_t* blob, float* state) {
  static float lds[agx::LDS_WORDS > agx::LDS_SOLVE_WORDS ? agx::LDS_WORDS : agx::LDS_SOLVE_WORDS];
  static float scratch[agx::SCR_WORDS];
  int flags = 0;
  int rc = run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_build<true>(blob, state, nullptr, scratch, nullptr, lds, lane); });
  if (!rc) rc = run_wave(lds, 64, [&](int lane) { int f = agx::collision_flags(blob, scratch, lane); if (lane == 0) flags = f; });
  return rc ? -1 : flags;
}
#if AGX_TASK == 5
// one env step of the drinking scene as libagx schedules it: frame_skip x sim_sub x [build (leaving the frames of the substep in the trace), solve],
// the water kernel over the trace, the finish kernel with the water buffer and the water kernel's report
extern "C" int agx_emu_step_water(const uint32_t* blob, float* state, float* water, const float* action, float* obs, float* reward, uint8_t* done, float* info) {
  static float lds[(agx::LDS_WORDS > agx::LDS_SOLVE_WORDS ? agx::LDS_WORDS : agx::LDS_SOLVE_WORDS) > agxw::LDS_WORDS ? (agx::LDS_WORDS > agx::LDS_SOLVE_WORDS ? agx::LDS_WORDS : agx::LDS_SOLVE_WORDS) : agxw::LDS_WORDS];
  static float scratch[agx::SCR_WORDS];
  const int* bi = (const int*)blob;
  const int frame_skip = (int)((const float*)blob)[bi[AGX_H_OFF_PARAMS] + AGX_P_FRAME_SKIP], sim_sub = bi[AGX_H_SIM_SUBSTEPS] > 1 ? bi[AGX_H_SIM_SUBSTEPS] : 1;
  const int nsub = frame_skip * sim_sub, slot = 12 * (bi[AGX_H_NDOF] + bi[AGX_H_NFREE]);
  float* trace = (float*)calloc((size_t)nsub * slot, 4); float report[agxw::REPORT_WORDS] = {0};
  int rc = 0;
  for (int k = 0; k < nsub && !rc; k++) {
    const float* act = k == 0 ? action : nullptr;
    rc = run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_build<true>(blob, state, act, scratch, nullptr, lds, lane, trace + (size_t)k * slot); });
    if (!rc) rc = run_wave(lds, agx::LDS_SOLVE_WORDS, [&](int lane) { agx::env_solve(blob, state, scratch, nullptr, lds, lane, k); });
  }
  if (!rc) rc = run_wave(lds, agxw::LDS_WORDS, [&](int lane) { agxw::water_env(blob, state, trace, water, report, nsub, lds, lane); });
  if (!rc) rc = run_wave(lds, agx::LDS_WORDS, [&](int lane) { agx::env_finish(blob, state, action, scratch, obs, reward, done, info, lds, lane, report, water); });
  free(trace);
  return rc;
}
#endif
// the water kernel body (csrc/agx_water.h) for one environment: `nsub` internal substeps over the given trace
extern "C" int agx_emu_water(const uint32_t* blob, const float* state, const float* trace, float* water, float* report, int nsub) {
  static float lds[agxw::LDS_WORDS];
  return run_wave(lds, agxw::LDS_WORDS, [&](int lane) { agxw::water_env(blob, state, trace, water, report, nsub, lds, lane); });
}
#if AGX_PGS_LV == 4
// the list scheduler of the wide row-local sweep (csrc/agx_pgs_lvw.h lvw_schedule) on given slot masks: masks3[3 r ...] = the 96-bit mask of row r;
// ss64[step] = the (up to four) rows of a step, a byte each (agx::HW_DUMMY = idle); returns the number of steps, -1 when they do not fit
extern "C" int agx_emu_lvw_schedule(const uint32_t* masks3, int nr, uint32_t* ss64) {
  if constexpr (!agx::LVW_COMPILED) { (void)masks3; (void)nr; (void)ss64; return -2; }
  else {
    static float hdr[agx::SCR_HDR_WORDS];
    for (int r = 0; r < nr && r < agx::MAX_ROWS; r++) { int* Xi = (int*)agx::hx_row(hdr, r); Xi[agx::H_MLO] = (int)masks3[3 * r]; Xi[agx::H_MHI] = (int)masks3[3 * r + 1]; Xi[agx::H_M2] = (int)masks3[3 * r + 2]; }
    static float lds[64];
    int ns = 0;
    const int rc = run_wave(lds, 64, [&](int lane) { agx::Ctx c; c.H = hdr; c.lane = lane; uint32_t ss; const int n = agx::lvw_schedule(c, lane, 0, nr, ss); ss64[lane] = ss; if (lane == 0) ns = n; });
    return rc ? -3 : ns;
  }
}
extern "C" int agx_emu_lvw_dummy_row() { return agx::HW_DUMMY; }
#endif
extern "C" int agx_emu_lds_bytes() { return agx::LDS_BYTES; }
// debug record layout of this variant (same order as agx_debug_layout of the product library)
extern "C" void agx_emu_debug_layout(int* out8) {
  out8[0] = agx::DBG_WORDS; out8[1] = agx::DBG_CON; out8[2] = agx::DBG_MINV; out8[3] = agx::MAX_DOF; out8[4] = agx::DBG_HDR; out8[5] = agx::DBG_LAM;
  out8[6] = agx::DBG_TIME; out8[7] = agx::DBG_QDD;
}
