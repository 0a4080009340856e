// Host harness that runs the product kernel source (csrc/agx_step.h) for ONE environment on the
// CPU wave emulator.  Built by tests/emu_lib.py into tests/emu/libagx_emu.so.  Test-only.
#include "agx_wave.h"
#include "agx_step.h"

namespace emu { Wave* W = nullptr; }

struct Args { const uint32_t* blob; float* state; const float* action; float* obs; float* reward; uint8_t* done; float* info; float* debug; float* lds; int mode, nsettle; };
static Args g_args;

static void fiber_entry() {
  const int lane = emu::W->cur;
  agx::env_step(g_args.blob, g_args.state, g_args.action, g_args.obs, g_args.reward, g_args.done, g_args.info, g_args.debug, g_args.lds, lane, g_args.mode, g_args.nsettle);
  emu::W->done[lane] = true;
  // a finished lane still has to take part in nothing: env_step ends uniformly for all lanes
  swapcontext(&emu::W->ctx[lane], &emu::W->main_ctx);
}

extern "C" int agx_emu_run(const uint32_t* blob, float* state, const float* action, float* obs, float* reward, uint8_t* done,
                           float* info, float* debug, int mode, int nsettle) {
  static emu::Wave wave;
  emu::W = &wave;
  memset(&wave, 0, sizeof wave);
  static float lds[agx::LDS_WORDS];
  memset(lds, 0, sizeof lds);
  g_args = Args{blob, state, action, obs, reward, done, info, debug, lds, mode, nsettle};
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
    // progress check: a full sweep that neither completed a rendezvous nor finished a lane nor
    // changed the arrival count means the lanes disagree on a collective (non-uniform control flow)
    int rem2 = 0; for (int l = 0; l < 64; l++) if (!wave.done[l]) rem2++;
    if (wave.gen == gen_before && wave.arrived == arrived_before && rem2 == remaining && rem2 != 64 - 0 && wave.arrived != 0 && spins > 4) {
      bool mixed = rem2 != 64 && rem2 != 0;
      if (mixed) { fprintf(stderr, "agx_emu: lanes diverged around a collective (%d lanes finished early)\n", 64 - rem2); rc = -1; break; }
    }
  }
  for (int l = 0; l < 64; l++) free(wave.stacks[l]);
  return rc;
}
extern "C" int agx_emu_lds_bytes() { return agx::LDS_BYTES; }
