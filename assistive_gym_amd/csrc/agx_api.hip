// agx_api.hip -- libagx: C ABI (include/agx.h), handles, streams and launch sequencing for gfx950.
// The kernels are compiled per variant (limits + task layer) in agx_kernels.hip; agx_create picks the variant whose
// task and limits fit the model blob.  One workgroup = one wavefront = one environment (64 threads); an env.step() is
// frame_skip x [build kernel, solve kernel] + finish kernel per chunk of environments.
#include "agx_wave.h"
#include "agx_variant.h"
#include "../../include/agx_blob.h"
#include "../../include/agx.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e)); else snprintf(buf, sizeof buf, "%s", what);
  g_err = buf; return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(AGX_E_HIP, #x, e_); } while (0)

// the warm-start memory (AGX_P_WARMSTART) of the environments whose state is replaced from outside is forgotten: their scratch record's
// entry count is zeroed (mask null = every environment)
extern "C" __global__ void __launch_bounds__(256)
agx_forget_warm_kernel(float* scratch, int scr_words, int word, const uint8_t* mask, int n_envs) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env < n_envs && (!mask || mask[env])) { int* m = (int*)scratch + (size_t)env * scr_words + word; m[0] = 0; m[1] = 0; }      // META_NWARM, META_NMAN
}

// done envs take a fresh record from the pool; coalesced copy, one wave per env
extern "C" __global__ void __launch_bounds__(64)
agx_reset_kernel(float* state, const float* pool, int pool_n, const uint8_t* done, int* episode, int n_envs, int sw, long long env_offset, int iter_word, int iteration) {
  const int env = blockIdx.x;
  if (env >= n_envs || !done[env]) return;
  int ep = episode[env] + 1;
  // the pool entry is a function of the GLOBAL environment index: the same env draws the same states on any GPU count
  const float* src = pool + (size_t)((env_offset + env + 977 * (long long)ep) % pool_n) * sw;
  for (int k = threadIdx.x; k < sw; k += 64) state[(size_t)env * sw + k] = src[k];
  if (threadIdx.x == 0) {
    episode[env] = ep;
    // agx_reset_done_at: the replacement joins the lock-stepped batch at its current iteration (env.py:185), so that it ends with the batch
    if (iteration >= 0) ((int*)state)[(size_t)env * sw + iter_word] = iteration;
  }
}

// ... and, for models with a cloth, the garment that belongs to that pool record (launched after agx_reset_kernel: episode[] is already advanced)
extern "C" __global__ void __launch_bounds__(256)
agx_reset_cloth_kernel(float* cloth, const float* pool, int pool_n, const uint8_t* done, const int* episode, int n_envs, int cw, long long env_offset) {
  const int env = blockIdx.x;
  if (env >= n_envs || !done[env]) return;
  const float* src = pool + (size_t)((env_offset + env + 977 * (long long)episode[env]) % pool_n) * cw;
  for (int k = threadIdx.x; k < cw; k += 256) cloth[(size_t)env * cw + k] = src[k];
}

// drinking, device-side reset: the 4 x 4 x 4 grid of water spheres above the cup's base frame position, world axes, at rest (drinking.py:160-167)
extern "C" __global__ void __launch_bounds__(64)
agx_place_water_kernel(float* water, const float* state, const uint32_t* blob, const uint8_t* mask, int n_envs, int sw, int cw) {
  const int env = blockIdx.x;
  if (env >= n_envs || (mask && !mask[env])) return;
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  const int oc = bi[AGX_H_OFF_CLOTH]; const int nn = bi[oc + AGX_CL_NN];
  const float* x0 = bf + oc + bi[oc + AGX_CL_OFF_X0];
  const float* cup = state + (size_t)env * sw + bi[AGX_H_S_FREE] + 13 * bi[AGX_H_TOOL_BODY];      // the cup's record holds its base frame (REFPOS = 0, compile_feeding)
  float* c = water + (size_t)env * cw;
  for (int k = threadIdx.x; k < 3 * nn; k += 64) { c[k] = x0[k] + cup[k % 3]; c[3 * nn + k] = 0.f; }
}

// models with a cloth, device-side reset: the garment of a freshly sampled environment is the loaded mesh shifted to the end effector
// (dressing.py:146-153: x = X0 + offset; the reset generator left the offset in the task words, AGX_DR_CLOTH_OFF), at rest
extern "C" __global__ void __launch_bounds__(256)
agx_place_cloth_kernel(float* cloth, const float* state, const uint32_t* blob, const uint8_t* mask, int n_envs, int sw, int cw) {
  const int env = blockIdx.x;
  if (env >= n_envs || (mask && !mask[env])) return;
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  const int oc = bi[AGX_H_OFF_CLOTH]; const int nn = bi[oc + AGX_CL_NN];
  const float* x0 = bf + oc + bi[oc + AGX_CL_OFF_X0];
  const float* off = state + (size_t)env * sw + bi[AGX_H_S_TASK] + AGX_DR_CLOTH_OFF;
  float* c = cloth + (size_t)env * cw;
  for (int k = threadIdx.x; k < 3 * nn; k += 256) { c[k] = x0[k] + off[k % 3]; c[3 * nn + k] = 0.f; }
}
// ... and when the settle of reset() is over the garment feels full gravity (dressing.py:195)
extern "C" __global__ void __launch_bounds__(64)
agx_cloth_gravity_kernel(float* state, const uint32_t* blob, const uint8_t* mask, int n_envs, int sw) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= n_envs || (mask && !mask[env])) return;
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  state[(size_t)env * sw + bi[AGX_H_S_TASK] + AGX_DR_CLOTH_GRAVITY] = bf[bi[AGX_H_OFF_RESET] + AGX_X_CLOTH_GRAVITY];
}

extern "C" __global__ void __launch_bounds__(64) agx_selftest_kernel(float* out_f, int* out_i, unsigned long long* out_m) {
  const int lane = wave_lane();
  float x = (float)(lane * lane % 17) * 0.25f - 1.0f;
  out_f[lane] = wave_sum(x);
  out_f[64 + lane] = wave_min(x + (float)((lane * 7) % 5));
  out_f[128 + lane] = wave_bcast(x, 37);
  out_f[192 + lane] = wave_shfl(x, (lane * 13 + 5) & 63);
  out_i[lane] = wave_sum_i(lane % 7);
  out_i[64 + lane] = wave_scan_excl(lane % 5);
  unsigned long long m = wave_ballot((lane % 3) == 1);
  out_m[lane] = m;
  out_i[128 + lane] = wave_rank(m);
  out_f[256 + lane] = wave_max(x);
}

}  // namespace

struct agx_handle_s {
  const agx_variant* V;   // the compiled kernel variant serving this model
  int* overflow_dev;      // contacts dropped by a budget since creation (all envs)
  int collision_tries;    // reset generator: successful IK restarts that may be rejected for a collision (AGX_X_COLLISION_TRIES)
  int *first_restart_dev, *chosen_dev; uint8_t* work_dev;   // reset generator: per-env retry bookkeeping
  long long env_offset;   // global index of env 0 of this handle (multi-GPU sharding), see agx_set_env_offset
  int device, n_envs, act_dim, obs_dim, sw;
  int iter_word;          // index of AGX_E_ITERATION in a state record (agx_reset_done_at)
  uint32_t* blob_dev;
  float* state_dev;
  float* scratch_dev;   // [n_envs][SCR_WORDS]: rows, predicted velocities, contacts handed between the kernels
  int* episode_dev;
  int frame_skip;
  int sim_sub;          // internal substeps per p.stepSimulation() (AGX_H_SIM_SUBSTEPS)
  // models with a cloth section (DressingBaxter): garments [n_envs][2][NN][3], the link frames of every substep of an env step for the
  // cloth kernel, its report to the finish kernel, and the garments of the reset pool (agx_set_cloth_pool)
  int cloth_nn, cloth_words, trace_words, report_words, cloth_lds;
  bool particles;       // the "cloth" section holds the water particles of the drinking scene (AGX_CL_PARTICLES), not a garment
  float *cloth_dev, *trace_dev, *report_dev; const float* cloth_pool_dev;
  bool can_sample;      // the variant has a reset generator (agx_reset.h) and the blob fits it
  bool manifold;        // AGX_P_MANIFOLD > 0 in the blob: the step path launches the variant's build kernel with the manifold stage
  const uint8_t* active;// per-env mask honoured by the build / solve launches (agx_reset's masked settle), normally null
  // staging for the *_host convenience calls
  float *act_dev, *obs_dev, *rew_dev, *info_dev; uint8_t* done_dev;
  hipEvent_t ev0, ev1;
  hipEvent_t kev[8][16];   // agx_step_timed: boundaries of the launches of one step, per chunk
  // The environments are stepped in independent chunks on internal streams (AGX_CHUNKS overrides the
  // count): every chunk runs its own build -> solve -> ... chain, so the tail of one chunk's kernel
  // (environments with many rows finish last) overlaps with the next kernel of another chunk and the
  // lean solve kernel shares the CUs with the LDS-heavy build kernel.
  int n_chunks; hipStream_t cs[8]; hipEvent_t fork_ev, join_ev[8];
  int reset_flags;      // AGX_X_FLAGS of the blob's reset section (0 without one)
  agx_handle_s* settle; // bed bathing (AGX_X_FLAGS bit 4): the handle of the rag-doll model whose settled records the sampler reads (agx_attach_settle_model; not owned)
  int settle_substeps;  // substeps of that settle (100 simulation steps: bed_bathing.py:130-131)
};

static void forget_warm(agx_handle_s* h, const uint8_t* mask_dev, hipStream_t st) {
  hipLaunchKernelGGL(agx_forget_warm_kernel, dim3((h->n_envs + 255) / 256), dim3(256), 0, st, h->scratch_dev, h->V->scr_words, h->V->scr_warm_word, mask_dev, h->n_envs);
}

extern "C" {

const char* agx_version(void) { return "libagx 0.2 (gfx950, wave-per-env stepper; variants: feeding, bed_bathing, scratch_itch, bed_settle, dressing)"; }
const char* agx_last_error(void) { return g_err.c_str(); }
int agx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int agx_lds_bytes_per_env(void) { return agx_variant_feeding()->lds_bytes; }
int agx_debug_words(void) { return agx_variant_feeding()->dbg_words; }   // FeedingJaco variant; agx_debug_layout(h) for a handle

static int create_fill(agx_handle h, const void* blob, size_t blob_bytes, int n_envs, int device);
void agx_destroy(agx_handle h);
int agx_create(const void* blob, size_t blob_bytes, int n_envs, int device, agx_handle* out) {
  if (!blob || !out || n_envs <= 0 || blob_bytes < sizeof(uint32_t) * AGX_H_COUNT) return fail(AGX_E_ARG, "agx_create: bad argument");
  const uint32_t* w = (const uint32_t*)blob; const int32_t* hi = (const int32_t*)blob;
  if (w[AGX_H_MAGIC] != AGX_BLOB_MAGIC || w[AGX_H_VERSION] != AGX_BLOB_VERSION || (size_t)hi[AGX_H_NWORDS] * 4 != blob_bytes)
    return fail(AGX_E_BLOB, "agx_create: not a model blob of this version");
  const agx_variant* V = nullptr;
  {
    // the first (smallest) variant with the model's task layer whose limits hold the model
    const agx_variant* all[17] = {agx_variant_feeding(), agx_variant_feeding_l(), agx_variant_feeding_m(), agx_variant_bed_bathing(), agx_variant_bed_bathing_l(), agx_variant_bed_bathing_m(),
                                 agx_variant_scratch_itch(), agx_variant_scratch_itch_m(), agx_variant_bed_settle(),
                                 agx_variant_dressing(), agx_variant_dressing_l(), agx_variant_dressing_m(), agx_variant_arm_manipulation(), agx_variant_arm_manipulation_l(), agx_variant_drinking(), agx_variant_drinking_l(), agx_variant_drinking_m()};
    bool task_seen = false;
    for (const agx_variant* v : all) {
      if (v->task_kind != hi[AGX_H_TASK_KIND]) continue;
      task_seen = true;
      if (hi[AGX_H_NDOF] > v->max_dof || hi[AGX_H_NFREE] > v->max_free || hi[AGX_H_NHUMAN] > v->max_human || hi[AGX_H_NCOLL] > v->max_coll ||
          hi[AGX_H_STATE_WORDS] > v->st_words || hi[AGX_H_NROBOT] > v->max_block || hi[AGX_H_NHDOF] > v->max_block) continue;
      V = v; break;
    }
    if (!task_seen) return fail(AGX_E_LIMIT, "agx_create: no kernel variant is compiled for the task of this model");
    if (!V || hi[AGX_H_NDOF] + 6 * hi[AGX_H_NFREE] > 128 || hi[AGX_H_NGROUP] > 64 || hi[AGX_H_NDOF] > 64)
      return fail(AGX_E_LIMIT, "agx_create: model exceeds the limits of the compiled kernel variants");
  }
  for (int g = 0; g < hi[AGX_H_NGROUP]; g++) {   // the broadphase compacts each collider range of a pair group into a 128-entry list
    const int32_t* G = hi + hi[AGX_H_OFF_GROUP] + g * AGX_G_STRIDE;
    if (G[AGX_G_A1] - G[AGX_G_A0] > 128 || G[AGX_G_B1] - G[AGX_G_B0] > 128 || (G[AGX_G_B0F] >= 0 && G[AGX_G_B1F] - G[AGX_G_B0F] > 128))
      return fail(AGX_E_LIMIT, "agx_create: a pair group has a collider range of more than 128 colliders");
  }
  bool can_sample = V->sample != nullptr;   // the reset generator's IK is compiled for a serial 7-DoF arm carrying the end effector
  if (can_sample) {
    const int32_t* X = hi + hi[AGX_H_OFF_RESET];
    const int32_t* T = hi + hi[AGX_H_OFF_TASK];
    // the arm: rs_narm joints in a serial chain (AGX_X_CHAIN: each joint's parent is the one before, the first hangs off the base), the
    // k-th one driven by action k, the last one carrying the end effector
    const bool ragdoll = (X[AGX_X_FLAGS] & 32) != 0;  // the rag-doll model of bed bathing: its sampler writes the drop record, there is no robot
    const bool mobile = (X[AGX_X_FLAGS] & 8) != 0 || ragdoll;    // a robot on wheels: no arm chain to solve (its NARM only says that the blob has a reset section)
    if (X[AGX_X_NARM] != V->rs_narm || (!ragdoll && hi[AGX_H_NROBOT] < V->rs_narm) || hi[AGX_H_NHUMAN] >= 64 || hi[AGX_H_NDOF] > 64 || X[AGX_X_TOC_ATTEMPTS] > 64 ||
        X[AGX_X_TOC_NGOALS] > 4 || X[AGX_X_PED_N] > 2) can_sample = false;
    if (ragdoll && hi[AGX_H_NDOF] != 6 + X[AGX_X_NJOINT] - 1) can_sample = false;    // virtual joints + every joint of the tree but the fixed waist
    if (mobile && !ragdoll && (X[AGX_X_MOBILE_LIFT_DOF] < 0 || X[AGX_X_MOBILE_LIFT_DOF] >= hi[AGX_H_NROBOT] || X[AGX_X_TOC_ATTEMPTS] != 0 || X[AGX_X_IK_RESTARTS] != 0)) can_sample = false;
    for (int k = 0; can_sample && !mobile && k < V->rs_narm; k++) {
      const int d = X[AGX_X_CHAIN + k];
      if (d < 0 || d >= hi[AGX_H_NROBOT]) { can_sample = false; break; }
      const int32_t* R = hi + hi[AGX_H_OFF_ROBOT] + d * AGX_R_STRIDE;
      const int dup = hi[AGX_H_TASK_KIND] == AGX_TASK_ARM_MANIPULATION ? T[AGX_T_DUP_ACT] : 0;     // robot_arm = 'both' lists a single arm twice: the second copy's actions drive it (robot.py:16)
      if (R[AGX_R_PARENT] != (k ? X[AGX_X_CHAIN + k - 1] : -1) || R[AGX_R_ACT] != k + dup) can_sample = false;
      if (k == V->rs_narm - 1 && T[AGX_T_EE_LINK] != d) can_sample = false;
    }
    for (int k = 0; can_sample && (X[AGX_X_FLAGS] & 512) && k < V->rs_narm; k++) {     // a two-armed robot: the second chain, driven by the actions behind the first arm's
      const int d = X[AGX_X_CHAIN2 + k];
      if (d < 0 || d >= hi[AGX_H_NROBOT]) { can_sample = false; break; }
      const int32_t* R = hi + hi[AGX_H_OFF_ROBOT] + d * AGX_R_STRIDE;
      if (R[AGX_R_PARENT] != (k ? X[AGX_X_CHAIN2 + k - 1] : -1) || R[AGX_R_ACT] != V->rs_narm + k) can_sample = false;
      if (k == V->rs_narm - 1 && (T[AGX_T_EE2_LINK] != d || T[AGX_T_TOOL2_BODY] <= 0 || T[AGX_T_TOOL2_BODY] >= hi[AGX_H_NFREE])) can_sample = false;
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(AGX_E_NOGPU, "agx_create: no HIP device (libagx has no CPU path)");
  if (device < 0 || device >= ndev) return fail(AGX_E_ARG, "agx_create: bad device index");
  HIPCHK(hipSetDevice(device));
  agx_handle h = new agx_handle_s();
  memset(h, 0, sizeof *h);
  h->can_sample = can_sample; h->V = V;
  h->manifold = ((const float*)blob)[hi[AGX_H_OFF_PARAMS] + AGX_P_MANIFOLD] > 0.f;
  if (h->manifold && !V->build_mf) { delete h; return fail(AGX_E_LIMIT, "agx_create: AGX_P_MANIFOLD is set, but this kernel variant is compiled without the manifold stage (feeding, bed_bathing, scratch_itch, arm_manipulation have it)"); }
  h->settle = nullptr; h->settle_substeps = 0;
  h->reset_flags = (hi[AGX_H_OFF_TARGETS] - hi[AGX_H_OFF_RESET] >= AGX_X_COUNT || hi[AGX_H_NWORDS] - hi[AGX_H_OFF_RESET] >= AGX_X_COUNT) ? hi[hi[AGX_H_OFF_RESET] + AGX_X_FLAGS] : 0;
  const int rc = create_fill(h, blob, blob_bytes, n_envs, device);
  if (rc != AGX_OK) { const std::string keep = g_err; agx_destroy(h); g_err = keep; return rc; }   // every error exit releases what was allocated
  *out = h;
  return AGX_OK;
}

// the allocations of agx_create; on failure the caller destroys the partly filled handle
static int create_fill(agx_handle h, const void* blob, size_t blob_bytes, int n_envs, int device) {
  const int32_t* hi = (const int32_t*)blob; const agx_variant* V = h->V; const bool can_sample = h->can_sample;
  h->iter_word = hi[AGX_H_S_ENV] + AGX_E_ITERATION;
  h->device = device; h->n_envs = n_envs; h->act_dim = hi[AGX_H_ACT_DIM]; h->obs_dim = hi[AGX_H_OBS_DIM]; h->sw = hi[AGX_H_STATE_WORDS];
  HIPCHK(hipMalloc(&h->blob_dev, blob_bytes));
  HIPCHK(hipMemcpy(h->blob_dev, blob, blob_bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&h->state_dev, (size_t)n_envs * h->sw * 4));
  HIPCHK(hipMemset(h->state_dev, 0, (size_t)n_envs * h->sw * 4));
  HIPCHK(hipMalloc(&h->scratch_dev, (size_t)n_envs * V->scr_words * 4));
  HIPCHK(hipMemset(h->scratch_dev, 0, (size_t)n_envs * V->scr_words * 4));
  HIPCHK(hipMalloc(&h->overflow_dev, 4));
  HIPCHK(hipMemset(h->overflow_dev, 0, 4));
  h->collision_tries = can_sample ? hi[hi[AGX_H_OFF_RESET] + AGX_X_COLLISION_TRIES] : 0;
  HIPCHK(hipMalloc(&h->first_restart_dev, (size_t)n_envs * 4)); HIPCHK(hipMalloc(&h->chosen_dev, (size_t)n_envs * 4)); HIPCHK(hipMalloc(&h->work_dev, (size_t)n_envs));
  h->frame_skip = (int)((const float*)blob)[hi[AGX_H_OFF_PARAMS] + AGX_P_FRAME_SKIP];
  h->sim_sub = hi[AGX_H_SIM_SUBSTEPS] > 1 ? hi[AGX_H_SIM_SUBSTEPS] : 1;
  if (hi[AGX_H_OFF_CLOTH]) {
    if (!V->cloth || !V->cloth_lds_bytes) return fail(AGX_E_LIMIT, "agx_create: the model has a cloth but the kernel variant of its task has no cloth kernel");
    const int32_t* cl = hi + hi[AGX_H_OFF_CLOTH];
    h->cloth_nn = cl[AGX_CL_NN];
    h->cloth_words = 6 * h->cloth_nn; h->report_words = AGX_CLOTH_REPORT_WORDS(h->cloth_nn) + AGX_CLOTH_SCRATCH_WORDS(h->cloth_nn);
    h->trace_words = h->frame_skip * h->sim_sub * (hi[AGX_H_NDOF] + hi[AGX_H_NFREE]) * 12;      // per substep: the frames of the moving links, then of the free bodies
    h->cloth_lds = V->cloth_lds_bytes(h->cloth_nn);          // agxc::lds_words of the variant's own cloth kernel
    h->particles = cl[AGX_CL_PARTICLES] != 0;
    if (cl[AGX_CL_PARTICLES]) {      // the water of the drinking scene (agx_water.h): one wavefront, lane = particle; report = one word per particle
      h->report_words = 64;
      if (h->cloth_nn > 64 || cl[AGX_CL_NSHAPE] > 192 || hi[AGX_H_NDOF] + hi[AGX_H_NHUMAN] + hi[AGX_H_NFREE] + 2 > 64 || hi[AGX_H_TASK_KIND] != AGX_TASK_DRINKING)
        return fail(AGX_E_LIMIT, "agx_create: particles exceed the limits of the water kernel");
    } else if (h->cloth_lds > 160 * 1024 || h->cloth_nn > 4096 || cl[AGX_CL_NCOLOR] - (AGX_CLOTH_THREADS / 64) * cl[AGX_CL_NPATCH_COLOR] > AGX_CLOTH_MAX_COLORS || cl[AGX_CL_NPATCH_COLOR] < 0 || cl[AGX_CL_MAX_LINKS_PER_COLOR] > 1024 ||
        cl[AGX_CL_NSHAPE] > 192 || hi[AGX_H_NDOF] + hi[AGX_H_NHUMAN] + 2 > 64 || cl[AGX_CL_NN] > 65535 || hi[AGX_H_NFREE] != 0) return fail(AGX_E_LIMIT, "agx_create: cloth exceeds the limits of the cloth kernel");
    HIPCHK(hipMalloc(&h->cloth_dev, (size_t)n_envs * h->cloth_words * 4)); HIPCHK(hipMemset(h->cloth_dev, 0, (size_t)n_envs * h->cloth_words * 4));
    HIPCHK(hipMalloc(&h->trace_dev, (size_t)n_envs * h->trace_words * 4)); HIPCHK(hipMemset(h->trace_dev, 0, (size_t)n_envs * h->trace_words * 4));
    HIPCHK(hipMalloc(&h->report_dev, (size_t)n_envs * h->report_words * 4)); HIPCHK(hipMemset(h->report_dev, 0, (size_t)n_envs * h->report_words * 4));
  }
  HIPCHK(hipMalloc(&h->episode_dev, (size_t)n_envs * 4));
  HIPCHK(hipMemset(h->episode_dev, 0, (size_t)n_envs * 4));
  HIPCHK(hipMalloc(&h->act_dev, (size_t)n_envs * h->act_dim * 4));
  HIPCHK(hipMalloc(&h->obs_dev, (size_t)n_envs * h->obs_dim * 4));
  HIPCHK(hipMalloc(&h->rew_dev, (size_t)n_envs * 4));
  HIPCHK(hipMalloc(&h->info_dev, (size_t)n_envs * AGX_INFO_DIM * 4));
  HIPCHK(hipMalloc(&h->done_dev, (size_t)n_envs));
  HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1));
  for (int c = 0; c < 8; c++) for (int k = 0; k < 16; k++) HIPCHK(hipEventCreate(&h->kev[c][k]));
  {
    const char* e = getenv("AGX_CHUNKS");
    // default: 3 chunks for large batches (3 chunk streams + the caller's stream = the 4 hardware queues HIP uses
    // by default; measured 335 k -> 386 k env-steps/s at 4096 envs, 4 chunks need GPU_MAX_HW_QUEUES=8), 1 otherwise
    // (the drinking scenes run unchunked: 20 light build / solve pairs and one long water launch per step -- measured 295 k env-steps/s in one chunk
    // under the profiler against 244 k in three)
    int nc = e ? atoi(e) : (h->particles ? 1 : (n_envs >= 2048 ? 3 : 1)); if (nc < 1) nc = 1; if (nc > 8) nc = 8; if (n_envs < 64 * nc) nc = 1;
    h->n_chunks = nc;
    HIPCHK(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
    for (int k = 0; k < nc; k++) { HIPCHK(hipStreamCreateWithFlags(&h->cs[k], hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&h->join_ev[k], hipEventDisableTiming)); }
  }
  HIPCHK(V->init());
  return AGX_OK;
}

void agx_destroy(agx_handle h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipFree(h->scratch_dev); hipFree(h->overflow_dev); hipFree(h->first_restart_dev); hipFree(h->chosen_dev); hipFree(h->work_dev); hipFree(h->blob_dev); hipFree(h->state_dev); hipFree(h->episode_dev); hipFree(h->act_dev); hipFree(h->obs_dev);
  hipFree(h->rew_dev); hipFree(h->info_dev); hipFree(h->done_dev);
  if (h->cloth_dev) { hipFree(h->cloth_dev); hipFree(h->trace_dev); hipFree(h->report_dev); }
  if (h->ev0) hipEventDestroy(h->ev0); if (h->ev1) hipEventDestroy(h->ev1);      // a handle whose creation failed half way holds nulls
  for (int c = 0; c < 8; c++) for (int k = 0; k < 16; k++) if (h->kev[c][k]) hipEventDestroy(h->kev[c][k]);
  if (h->fork_ev) hipEventDestroy(h->fork_ev);
  for (int k = 0; k < h->n_chunks; k++) { if (h->cs[k]) hipStreamDestroy(h->cs[k]); if (h->join_ev[k]) hipEventDestroy(h->join_ev[k]); }
  delete h;
}

int agx_dims(agx_handle h, int* n_envs, int* act_dim, int* obs_dim, int* state_words) {
  if (!h) return fail(AGX_E_ARG, "agx_dims: null handle");
  if (n_envs) *n_envs = h->n_envs; if (act_dim) *act_dim = h->act_dim; if (obs_dim) *obs_dim = h->obs_dim; if (state_words) *state_words = h->sw;
  return AGX_OK;
}

int agx_set_state(agx_handle h, const float* host_states) {
  if (!h || !host_states) return fail(AGX_E_ARG, "agx_set_state: bad argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->state_dev, host_states, (size_t)h->n_envs * h->sw * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(h->episode_dev, 0, (size_t)h->n_envs * 4));
  forget_warm(h, nullptr, nullptr); HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
  return AGX_OK;
}
int agx_get_state(agx_handle h, float* host_states) {
  if (!h || !host_states) return fail(AGX_E_ARG, "agx_get_state: bad argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(host_states, h->state_dev, (size_t)h->n_envs * h->sw * 4, hipMemcpyDeviceToHost));
  return AGX_OK;
}
int agx_state_dev(agx_handle h, float** out_dev) { if (!h || !out_dev) return fail(AGX_E_ARG, "agx_state_dev: bad argument"); *out_dev = h->state_dev; return AGX_OK; }

// one p.stepSimulation() for the environments [e0, e0+ne): build + solve
static int launch_substep(agx_handle h, const float* act, float* dbg, int e0, int ne, hipStream_t st, int phase, bool settle) {
  (h->manifold ? h->V->build_mf : h->V->build)(st, ne, h->blob_dev, h->state_dev, act, h->scratch_dev, dbg, e0, h->n_envs, h->sw, h->act_dim, h->active, h->overflow_dev, h->trace_dev, h->trace_words, phase);
  HIPCHK(hipGetLastError());
  const int ph = settle ? (phase | AGX_PHASE_SETTLE) : phase;
  // the packed kernel (four environments per wavefront) where the variant has one; the debug path keeps the single-environment kernel
  // (its per-phase cycle counters); AGX_SOLVE=old selects it for same-box A/B runs
  h->V->solve(st, ne, h->blob_dev, h->state_dev, h->scratch_dev, dbg, e0, h->n_envs, h->sw, h->active, ph);
  HIPCHK(hipGetLastError());
  return AGX_OK;
}
// fork the caller's stream into the chunk streams, run `n_substeps` (and optionally the finish
// kernel) per chunk, join back.  Work of one chunk is ordered; chunks are independent.
static int launch_chunked(agx_handle h, int n_substeps, const float* act, float* obs, float* rew, uint8_t* done, float* info, float* dbg, bool finish, void* stream) {
  HIPCHK(hipSetDevice(h->device));
  hipStream_t user = (hipStream_t)stream;
  const int nc = h->n_chunks, per = (h->n_envs + nc - 1) / nc;
  HIPCHK(hipEventRecord(h->fork_ev, user));
  for (int c = 0; c < nc; c++) {
    const int e0 = c * per, ne = (e0 + per <= h->n_envs ? per : h->n_envs - e0);
    if (ne <= 0) continue;
    hipStream_t st = nc == 1 ? user : h->cs[c];
    if (nc > 1) HIPCHK(hipStreamWaitEvent(st, h->fork_ev, 0));
    // n_substeps counts p.stepSimulation() calls, each sim_sub internal substeps long.  With a cloth, the rigid substeps of up to frame_skip
    // calls run first (leaving their link frames in the trace), then one cloth launch replays them (one-way coupling, agx_cloth.h).
    for (int g0 = 0; g0 < n_substeps; g0 += h->frame_skip) {
      const int gs = n_substeps - g0 < h->frame_skip ? n_substeps - g0 : h->frame_skip, nsub = gs * h->sim_sub;
      for (int k = 0; k < nsub; k++) { const bool first = g0 == 0 && k == 0; int rc = launch_substep(h, first ? act : nullptr, first ? dbg : nullptr, e0, ne, st, k, !finish); if (rc) return rc; }
      if (h->cloth_dev) {
        h->V->cloth(st, ne, h->blob_dev, h->state_dev, h->trace_dev, h->cloth_dev, h->report_dev, e0, h->n_envs, h->sw, h->trace_words, h->cloth_words, h->report_words, nsub,
                    h->active, h->cloth_lds);
        HIPCHK(hipGetLastError());
      }
    }
    if (finish) {
      h->V->finish(st, ne, h->blob_dev, h->state_dev, act, h->scratch_dev, obs, rew, done, info, e0, h->n_envs, h->sw, h->act_dim, h->obs_dim, h->report_dev, h->report_words,
                   h->cloth_dev, h->cloth_words);
      HIPCHK(hipGetLastError());
    }
    if (nc > 1) { HIPCHK(hipEventRecord(h->join_ev[c], st)); HIPCHK(hipStreamWaitEvent(user, h->join_ev[c], 0)); }
  }
  return AGX_OK;
}
static int launch_step(agx_handle h, const float* act, float* obs, float* rew, uint8_t* done, float* info, float* dbg, void* stream) {
  return launch_chunked(h, h->frame_skip, act, obs, rew, done, info, dbg, true, stream);
}
int agx_settle(agx_handle h, int n_substeps, void* stream) {
  if (!h || n_substeps < 0) return fail(AGX_E_ARG, "agx_settle: bad argument");
  return launch_chunked(h, n_substeps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, stream);
}
int agx_settle_debug(agx_handle h, int n_substeps, float* dbg, void* stream) {
  if (!h || n_substeps < 1 || !dbg) return fail(AGX_E_ARG, "agx_settle_debug: bad argument");
  return launch_chunked(h, n_substeps, nullptr, nullptr, nullptr, nullptr, nullptr, dbg, false, stream);
}
int agx_step(agx_handle h, const float* a, float* obs, float* rew, uint8_t* done, float* info, void* stream) {
  if (!h || !a || !obs || !rew || !done) return fail(AGX_E_ARG, "agx_step: bad argument");
  return launch_step(h, a, obs, rew, done, info, nullptr, stream);
}
int agx_step_debug(agx_handle h, const float* a, float* obs, float* rew, uint8_t* done, float* info, float* dbg, void* stream) {
  if (!h || !a || !obs || !rew || !done || !dbg) return fail(AGX_E_ARG, "agx_step_debug: bad argument");
  return launch_step(h, a, obs, rew, done, info, dbg, stream);
}
int agx_step_timed(agx_handle h, const float* a, float* obs, float* rew, uint8_t* done, float* info, void* stream, float* ms3, int* launches3) {
  if (!h || !a || !obs || !rew || !done || !ms3) return fail(AGX_E_ARG, "agx_step_timed: bad argument");
  if (2 * h->frame_skip * h->sim_sub + 2 > 16 || h->cloth_dev) return fail(AGX_E_LIMIT, "agx_step_timed: too many launches per step for the event table (frame_skip x substeps), or a model with a cloth");
  HIPCHK(hipSetDevice(h->device));
  // the same chunked launch sequence as agx_step, with an event after every launch on its chunk stream
  hipStream_t user = (hipStream_t)stream;
  const int nc = h->n_chunks, per = (h->n_envs + nc - 1) / nc;
  int nev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipEventRecord(h->fork_ev, user));
  for (int c = 0; c < nc; c++) {
    const int e0 = c * per, ne = (e0 + per <= h->n_envs ? per : h->n_envs - e0);
    if (ne <= 0) continue;
    hipStream_t st = nc == 1 ? user : h->cs[c];
    if (nc > 1) HIPCHK(hipStreamWaitEvent(st, h->fork_ev, 0));
    int e = 0;
    HIPCHK(hipEventRecord(h->kev[c][e++], st));
    for (int k = 0; k < h->frame_skip; k++) {
      (h->manifold ? h->V->build_mf : h->V->build)(st, ne, h->blob_dev, h->state_dev, k == 0 ? a : nullptr, h->scratch_dev, nullptr, e0, h->n_envs, h->sw, h->act_dim, (const uint8_t*)nullptr, h->overflow_dev, nullptr, 0, k);
      HIPCHK(hipEventRecord(h->kev[c][e++], st));
      h->V->solve(st, ne, h->blob_dev, h->state_dev, h->scratch_dev, nullptr, e0, h->n_envs, h->sw, (const uint8_t*)nullptr, k);
      HIPCHK(hipEventRecord(h->kev[c][e++], st));
    }
    h->V->finish(st, ne, h->blob_dev, h->state_dev, a, h->scratch_dev, obs, rew, done, info, e0, h->n_envs, h->sw, h->act_dim, h->obs_dim, nullptr, 0, nullptr, 0);
    HIPCHK(hipEventRecord(h->kev[c][e++], st));
    HIPCHK(hipGetLastError());
    nev[c] = e;
    if (nc > 1) { HIPCHK(hipEventRecord(h->join_ev[c], st)); HIPCHK(hipStreamWaitEvent(user, h->join_ev[c], 0)); }
  }
  HIPCHK(hipStreamSynchronize(user));
  ms3[0] = ms3[1] = ms3[2] = 0.f;
  int cnt[3] = {0, 0, 0};
  for (int c = 0; c < nc; c++) for (int k = 0; k + 1 < nev[c]; k++) {
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, h->kev[c][k], h->kev[c][k + 1]));
    const int t = k == nev[c] - 2 ? 2 : (k & 1);
    ms3[t] += ms; cnt[t]++;
  }
  if (launches3) { launches3[0] = cnt[0]; launches3[1] = cnt[1]; launches3[2] = cnt[2]; }
  return AGX_OK;
}
int agx_observe(agx_handle h, float* obs, void* stream) { return agx_observe_masked(h, obs, nullptr, stream); }
int agx_observe_masked(agx_handle h, float* obs, const uint8_t* mask_dev, void* stream) {
  if (!h || !obs) return fail(AGX_E_ARG, "agx_observe: bad argument");
  HIPCHK(hipSetDevice(h->device));
  h->V->observe((hipStream_t)stream, h->n_envs, h->blob_dev, h->state_dev, obs, h->sw, h->obs_dim, mask_dev);
  HIPCHK(hipGetLastError());
  return AGX_OK;
}
static int launch_sample(agx_handle h, uint64_t seed, const uint64_t* seeds_dev, const uint8_t* mask_dev, int impairment_mode, int gender_mode, float* ik_info_dev, void* stream) {
  if (impairment_mode < -2 || impairment_mode > 3 || gender_mode < -1 || gender_mode > 1) return fail(AGX_E_ARG, "reset: bad impairment / gender mode");
  if (!h->can_sample) return fail(AGX_E_LIMIT, "reset: no device-side reset generator for this model (FeedingJaco-type scenes with a serial 7-DoF arm only); "
                                               "provide post-reset states with agx_set_state / a pool for agx_reset_done");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  forget_warm(h, mask_dev, st);          // the sampled environments start without a warm-start memory
  const float* settled = nullptr; int settled_sw = 0; const float* fell = nullptr;
  if (h->reset_flags & 16) {
    // bed bathing (bed_bathing.py:119-137): the human is a rag doll dropped onto the bed -- the attached model samples its drop record from the
    // same seeds, settles for 100 simulation steps, and its state records tell this model's sampler where the human lies.
    // Arm manipulation (arm_manipulation.py:117-146; FLAGS bit 7): the attached model is the FALL model -- this blob at gravity -1, its
    // sampler writing the record the posed arm falls from -- and the rag-doll model is attached to that one: rag doll, then the fall, then here.
    agx_handle_s* hs = h->settle;
    if (!hs) return fail(AGX_E_ARG, "reset: this model's human comes out of a rag-doll settle; attach that model first (agx_attach_settle_model)");
    const bool two = (h->reset_flags & 128) && !(h->reset_flags & 256);
    if (two && !(hs->reset_flags & 256)) return fail(AGX_E_ARG, "reset: this model's reset lets the arm fall in a second handle (ModelBlob.fall_model()); attach that one, and the rag-doll model to it");
    agx_handle_s* hr = two ? hs->settle : hs;
    if (!hr) return fail(AGX_E_ARG, "reset: the fall model has no rag-doll model attached (agx_attach_settle_model)");
    hr->V->sample(st, hr->n_envs, hr->blob_dev, hr->state_dev, (unsigned long long)seed, (const unsigned long long*)seeds_dev, mask_dev, impairment_mode,
                  gender_mode, nullptr, hr->episode_dev, hr->sw, nullptr, hr->chosen_dev, nullptr, 0, nullptr);
    HIPCHK(hipGetLastError());
    hr->active = mask_dev;
    int rc = launch_chunked(hr, two ? hs->settle_substeps : h->settle_substeps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, stream);
    hr->active = nullptr;
    if (rc) return rc;
    settled = hr->state_dev; settled_sw = hr->sw;
    if (two) {
      forget_warm(hs, mask_dev, st);
      hs->V->sample(st, hs->n_envs, hs->blob_dev, hs->state_dev, (unsigned long long)seed, (const unsigned long long*)seeds_dev, mask_dev, impairment_mode,
                    gender_mode, nullptr, hs->episode_dev, hs->sw, nullptr, hs->chosen_dev, settled, settled_sw, nullptr);
      HIPCHK(hipGetLastError());
      hs->active = mask_dev;
      rc = launch_chunked(hs, h->settle_substeps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, stream);
      hs->active = nullptr;
      if (rc) return rc;
      fell = hs->state_dev;
    }
  }
  h->V->sample(st, h->n_envs, h->blob_dev, h->state_dev, (unsigned long long)seed, (const unsigned long long*)seeds_dev, mask_dev, impairment_mode,
               gender_mode, ik_info_dev, h->episode_dev, h->sw, nullptr, h->chosen_dev, settled, settled_sw, fell);
  HIPCHK(hipGetLastError());
  // collision rejection (robot.py:105-112, env.py:299-308): the contacts of the sampled state come from the stepper's own build kernel;
  // a state whose arm / tool touches the human, the table or the wheelchair is re-sampled from the next IK restart.  Fixed schedule
  // of `collision_tries` rounds (no host round trip): the kernels of a round exit at once for the environments that are settled.
  for (int t = 0; t < h->collision_tries; t++) {
    const uint8_t* active = t == 0 ? mask_dev : h->work_dev;
    h->V->build(st, h->n_envs, h->blob_dev, h->state_dev, nullptr, h->scratch_dev, nullptr, 0, h->n_envs, h->sw, h->act_dim, active, h->overflow_dev, nullptr, 0, 0);
    HIPCHK(hipGetLastError());
    h->V->verdict(st, h->n_envs, h->blob_dev, h->scratch_dev, active, h->work_dev, h->first_restart_dev, h->chosen_dev);
    HIPCHK(hipGetLastError());
    h->V->sample(st, h->n_envs, h->blob_dev, h->state_dev, (unsigned long long)seed, (const unsigned long long*)seeds_dev, h->work_dev, impairment_mode,
                 gender_mode, ik_info_dev, h->episode_dev, h->sw, h->first_restart_dev, h->chosen_dev, settled, settled_sw, fell);
    HIPCHK(hipGetLastError());
  }
  if (h->cloth_dev && h->particles) {       // the water goes into the sampled cup
    hipLaunchKernelGGL(agx_place_water_kernel, dim3(h->n_envs), dim3(64), 0, st, h->cloth_dev, h->state_dev, h->blob_dev, mask_dev, h->n_envs, h->sw, h->cloth_words);
    HIPCHK(hipGetLastError());
  }
  if (h->cloth_dev && !h->particles) {      // the garment goes where the sampled end effector is (dressing task words; the water has its own placement)
    hipLaunchKernelGGL(agx_place_cloth_kernel, dim3(h->n_envs), dim3(256), 0, st, h->cloth_dev, h->state_dev, h->blob_dev, mask_dev, h->n_envs, h->sw, h->cloth_words);
    HIPCHK(hipGetLastError());
  }
  return AGX_OK;
}
int agx_attach_settle_model(agx_handle h, agx_handle settle, int n_substeps) {
  if (!h || n_substeps < 0) return fail(AGX_E_ARG, "agx_attach_settle_model: bad argument");
  if (!settle) { h->settle = nullptr; return AGX_OK; }
  if (!(h->reset_flags & 16)) return fail(AGX_E_ARG, "agx_attach_settle_model: this model's reset does not read a settle record");
  const bool wants_fall = (h->reset_flags & 128) && !(h->reset_flags & 256);      // arm manipulation: the fall model goes here, the rag doll behind it
  if (wants_fall ? !(settle->reset_flags & 256) : !(settle->reset_flags & 32)) return fail(AGX_E_ARG, wants_fall ? "agx_attach_settle_model: this model takes its fall model (the same blob with AGX_X_FLAGS bit 8)" : "agx_attach_settle_model: the second handle is not a rag-doll model with a drop sampler");
  if (!settle->can_sample) return fail(AGX_E_ARG, "agx_attach_settle_model: the second handle has no sampler");
  if (settle->n_envs != h->n_envs || settle->device != h->device) return fail(AGX_E_ARG, "agx_attach_settle_model: both handles must hold the same number of environments on the same device");
  h->settle = settle; h->settle_substeps = n_substeps;
  return AGX_OK;
}
int agx_sample_reset(agx_handle h, uint64_t seed, int impairment_mode, int gender_mode, float* ik_info_dev, void* stream) {
  if (!h) return fail(AGX_E_ARG, "agx_sample_reset: null handle");
  return launch_sample(h, seed, nullptr, nullptr, impairment_mode, gender_mode, ik_info_dev, stream);
}
int agx_reset(agx_handle h, const uint8_t* mask_dev, const uint64_t* seeds_dev, uint64_t seed, int impairment_mode, int gender_mode, int settle_substeps, void* stream) {
  if (!h || settle_substeps < 0) return fail(AGX_E_ARG, "agx_reset: bad argument");
  int rc = launch_sample(h, seed, seeds_dev, mask_dev, impairment_mode, gender_mode, nullptr, stream);
  if (rc) return rc;
  h->active = mask_dev;   // the settle substeps touch the masked environments only
  rc = launch_chunked(h, settle_substeps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, stream);
  h->active = nullptr;
  if (!rc && h->cloth_dev && !h->particles) {
    hipLaunchKernelGGL(agx_cloth_gravity_kernel, dim3((h->n_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->state_dev, h->blob_dev, mask_dev, h->n_envs, h->sw);
    HIPCHK(hipGetLastError());
  }
  return rc;
}
int agx_reset_done(agx_handle h, const float* pool_dev, int pool_n, const uint8_t* done_dev, void* stream) { return agx_reset_done_at(h, pool_dev, pool_n, done_dev, -1, stream); }
int agx_reset_done_at(agx_handle h, const float* pool_dev, int pool_n, const uint8_t* done_dev, int iteration, void* stream) {
  if (!h || !pool_dev || pool_n <= 0 || !done_dev) return fail(AGX_E_ARG, "agx_reset_done: bad argument");
  HIPCHK(hipSetDevice(h->device));
  if (h->cloth_dev && !h->cloth_pool_dev) return fail(AGX_E_ARG, "agx_reset_done: the model has a cloth; give the garments of the pool with agx_set_cloth_pool first");
  hipLaunchKernelGGL(agx_reset_kernel, dim3(h->n_envs), dim3(64), 0, (hipStream_t)stream, h->state_dev, pool_dev, pool_n, done_dev, h->episode_dev, h->n_envs, h->sw, h->env_offset,
                     h->iter_word, iteration);
  forget_warm(h, done_dev, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  if (h->cloth_dev) {
    hipLaunchKernelGGL(agx_reset_cloth_kernel, dim3(h->n_envs), dim3(256), 0, (hipStream_t)stream, h->cloth_dev, h->cloth_pool_dev, pool_n, done_dev, h->episode_dev, h->n_envs, h->cloth_words, h->env_offset);
    HIPCHK(hipGetLastError());
  }
  return AGX_OK;
}
// ---- garments of models with a cloth section: float[n_envs][2][NN][3], node positions then node velocities
int agx_cloth_nodes(agx_handle h, int* nodes) { if (!h || !nodes) return fail(AGX_E_ARG, "agx_cloth_nodes: bad argument"); *nodes = h->cloth_nn; return AGX_OK; }
int agx_set_cloth(agx_handle h, const float* host_cloth) {
  if (!h || !host_cloth || !h->cloth_dev) return fail(AGX_E_ARG, "agx_set_cloth: bad argument or a model without a cloth");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->cloth_dev, host_cloth, (size_t)h->n_envs * h->cloth_words * 4, hipMemcpyHostToDevice));
  return AGX_OK;
}
int agx_get_cloth(agx_handle h, float* host_cloth) {
  if (!h || !host_cloth || !h->cloth_dev) return fail(AGX_E_ARG, "agx_get_cloth: bad argument or a model without a cloth");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(host_cloth, h->cloth_dev, (size_t)h->n_envs * h->cloth_words * 4, hipMemcpyDeviceToHost));
  return AGX_OK;
}
int agx_cloth_dev(agx_handle h, float** out_dev) { if (!h || !out_dev || !h->cloth_dev) return fail(AGX_E_ARG, "agx_cloth_dev: bad argument or a model without a cloth"); *out_dev = h->cloth_dev; return AGX_OK; }
int agx_get_cloth_report(agx_handle h, float* host_report, int* words_per_env) {
  if (!h || !h->cloth_dev || !h->report_dev) return fail(AGX_E_ARG, "agx_get_cloth_report: bad argument or a model without a cloth");
  const int words = AGX_CLOTH_REPORT_WORDS(h->cloth_nn);
  if (words_per_env) *words_per_env = words;
  if (!host_report) return AGX_OK;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy2D(host_report, (size_t)words * 4, h->report_dev, (size_t)h->report_words * 4, (size_t)words * 4, (size_t)h->n_envs, hipMemcpyDeviceToHost));
  return AGX_OK;
}
int agx_set_cloth_pool(agx_handle h, const float* pool_cloth_dev) {
  if (!h || !h->cloth_dev) return fail(AGX_E_ARG, "agx_set_cloth_pool: bad argument or a model without a cloth");
  h->cloth_pool_dev = pool_cloth_dev; return AGX_OK;
}

int agx_check_collisions(agx_handle h, uint8_t* flags_host, void* stream) {
  if (!h || !flags_host) return fail(AGX_E_ARG, "agx_check_collisions: bad argument");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  // the stepper's own collision pass on the states as they are (no motor targets, nothing is integrated: the states do not change)
  h->V->build(st, h->n_envs, h->blob_dev, h->state_dev, nullptr, h->scratch_dev, nullptr, 0, h->n_envs, h->sw, h->act_dim, nullptr, h->overflow_dev, nullptr, 0, 0);
  HIPCHK(hipGetLastError());
  h->V->collision_flags(st, h->n_envs, h->blob_dev, h->scratch_dev, h->work_dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(flags_host, h->work_dev, (size_t)h->n_envs, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return AGX_OK;
}

int agx_set_env_offset(agx_handle h, long long env_offset) {
  if (!h || env_offset < 0) return fail(AGX_E_ARG, "agx_set_env_offset: bad argument");
  h->env_offset = env_offset; return AGX_OK;
}
int agx_overflow_count(agx_handle h, int* out) {
  if (!h || !out) return fail(AGX_E_ARG, "agx_overflow_count: bad argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(out, h->overflow_dev, 4, hipMemcpyDeviceToHost));
  return AGX_OK;
}
int agx_debug_layout(agx_handle h, int* out8) {
  if (!h || !out8) return fail(AGX_E_ARG, "agx_debug_layout: bad argument");
  const agx_variant* V = h->V;
  out8[0] = V->dbg_words; out8[1] = V->dbg_con; out8[2] = V->dbg_minv; out8[3] = V->max_dof; out8[4] = V->dbg_hdr; out8[5] = V->dbg_lam; out8[6] = V->dbg_time; out8[7] = V->dbg_qdd;
  return AGX_OK;
}
const char* agx_variant_name(agx_handle h) { return h ? h->V->name : ""; }

// ---- observation all-gather over RCCL: the only exchange the sharded path has -------------------------------
// RCCL is bound at run time (dlopen), so a process that never gathers does not need it; a library instance that is already
// loaded (e.g. the one PyTorch ships) is preferred so that the process ends up with ONE RCCL.
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, const void*, int) = nullptr;   // ncclUniqueId is passed by value: 128 bytes, see call site
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
struct agx_unique_id { char bytes[128]; };   // layout of ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
RcclApi g_rccl;
int rccl_load() {
  if (g_rccl.lib) return AGX_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // already in the process?
  for (const char* n : names) { if (lib) break; lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
  if (!lib) return fail(AGX_E_HIP, "agx_allgather: librccl.so not found");
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, const void*, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) return fail(AGX_E_HIP, "agx_allgather: RCCL symbols missing");
  g_rccl.lib = lib;
  return AGX_OK;
}
int rccl_fail(const char* what, int rc) {
  char buf[256]; snprintf(buf, sizeof buf, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  return fail(AGX_E_HIP, buf);
}
}  // namespace

// [n_envs][obs_dim + 4] = observation | reward | done (0 / 1) | info[0] (total_force_on_human) | info[1] (task_success): what one consumer of the
// whole batch reads per step (learn.py:26,72: the sampler's obs, reward, done, info) as ONE record per environment -- one all-gather per step
__global__ void agx_pack_kernel(const float* __restrict__ obs, const float* __restrict__ rew, const uint8_t* __restrict__ done, const float* __restrict__ info,
                                float* __restrict__ out, int n, int od) {
  const int w = od + 4, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * w) return;
  const int e = i / w, c = i - e * w;
  out[i] = c < od ? obs[(size_t)e * od + c] : c == od ? rew[e] : c == od + 1 ? (done[e] ? 1.f : 0.f) : info[(size_t)e * AGX_INFO_DIM + (c - od - 2)];
}
int agx_pack_step(agx_handle h, const float* obs_dev, const float* reward_dev, const uint8_t* done_dev, const float* info_dev, float* packed_dev, void* stream) {
  if (!h || !obs_dev || !reward_dev || !done_dev || !info_dev || !packed_dev) return fail(AGX_E_ARG, "agx_pack_step: bad argument");
  HIPCHK(hipSetDevice(h->device));
  const int total = h->n_envs * (h->obs_dim + 4);
  hipLaunchKernelGGL(agx_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs_dev, reward_dev, done_dev, info_dev, packed_dev, h->n_envs, h->obs_dim);
  HIPCHK(hipGetLastError());
  return AGX_OK;
}

int agx_comm_unique_id(void* out128) {
  if (!out128) return fail(AGX_E_ARG, "agx_comm_unique_id: null");
  int rc = rccl_load(); if (rc) return rc;
  rc = g_rccl.GetUniqueId(out128);
  return rc ? rccl_fail("ncclGetUniqueId", rc) : AGX_OK;
}
int agx_comm_init_rank(int device, int rank, int world, const void* unique_id128, void** comm_out) {
  if (!unique_id128 || !comm_out || rank < 0 || rank >= world) return fail(AGX_E_ARG, "agx_comm_init_rank: bad argument");
  int rc = rccl_load(); if (rc) return rc;
  HIPCHK(hipSetDevice(device));
  // ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank): the id is a 128-byte struct passed by value
  typedef int (*init_fn)(void**, int, agx_unique_id, int);
  agx_unique_id id; memcpy(id.bytes, unique_id128, sizeof id.bytes);
  rc = ((init_fn)(void*)g_rccl.CommInitRank)(comm_out, world, id, rank);
  return rc ? rccl_fail("ncclCommInitRank", rc) : AGX_OK;
}
int agx_comm_destroy(void* comm) {
  if (!comm) return AGX_OK;
  int rc = rccl_load(); if (rc) return rc;
  rc = g_rccl.CommDestroy(comm);
  return rc ? rccl_fail("ncclCommDestroy", rc) : AGX_OK;
}
int agx_allgather(agx_handle h, const float* local_dev, float* gathered_dev, size_t floats_per_rank, void* comm, void* stream) {
  if (!h || !local_dev || !gathered_dev) return fail(AGX_E_ARG, "agx_allgather: bad argument");
  HIPCHK(hipSetDevice(h->device));
  if (!comm) {   // a single rank: the gathered batch is the local shard
    if (gathered_dev != local_dev) HIPCHK(hipMemcpyAsync(gathered_dev, local_dev, floats_per_rank * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AGX_OK;
  }
  int rc = rccl_load(); if (rc) return rc;
  rc = g_rccl.AllGather(local_dev, gathered_dev, floats_per_rank, 7 /* ncclFloat32 */, comm, (hipStream_t)stream);
  return rc ? rccl_fail("ncclAllGather", rc) : AGX_OK;
}

int agx_step_host(agx_handle h, const float* a, float* obs, float* rew, uint8_t* done, float* info) {
  if (!h || !a || !obs || !rew || !done) return fail(AGX_E_ARG, "agx_step_host: bad argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->act_dev, a, (size_t)h->n_envs * h->act_dim * 4, hipMemcpyHostToDevice));
  int rc = launch_step(h, h->act_dev, h->obs_dev, h->rew_dev, h->done_dev, h->info_dev, nullptr, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(obs, h->obs_dev, (size_t)h->n_envs * h->obs_dim * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rew, h->rew_dev, (size_t)h->n_envs * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(done, h->done_dev, (size_t)h->n_envs, hipMemcpyDeviceToHost));
  if (info) HIPCHK(hipMemcpy(info, h->info_dev, (size_t)h->n_envs * AGX_INFO_DIM * 4, hipMemcpyDeviceToHost));
  return AGX_OK;
}
int agx_observe_host(agx_handle h, float* obs) {
  if (!h || !obs) return fail(AGX_E_ARG, "agx_observe_host: bad argument");
  int rc = agx_observe(h, h->obs_dev, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(obs, h->obs_dev, (size_t)h->n_envs * h->obs_dim * 4, hipMemcpyDeviceToHost));
  return AGX_OK;
}

int agx_profile_begin(agx_handle h, void* stream) { if (!h) return fail(AGX_E_ARG, "agx_profile_begin: null handle"); HIPCHK(hipSetDevice(h->device)); HIPCHK(hipEventRecord(h->ev0, (hipStream_t)stream)); return AGX_OK; }
int agx_profile_end(agx_handle h, void* stream, float* ms) {
  if (!h || !ms) return fail(AGX_E_ARG, "agx_profile_end: bad argument");
  HIPCHK(hipSetDevice(h->device)); HIPCHK(hipEventRecord(h->ev1, (hipStream_t)stream)); HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return AGX_OK;
}
int agx_synchronize(agx_handle h, void* stream) { if (!h) return fail(AGX_E_ARG, "agx_synchronize: null handle"); HIPCHK(hipSetDevice(h->device)); HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return AGX_OK; }

int agx_selftest(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(AGX_E_NOGPU, "agx_selftest: no HIP device");
  HIPCHK(hipSetDevice(device));
  float* of; int* oi; unsigned long long* om;
  HIPCHK(hipMalloc(&of, 320 * 4)); HIPCHK(hipMalloc(&oi, 192 * 4)); HIPCHK(hipMalloc(&om, 64 * 8));
  hipLaunchKernelGGL(agx_selftest_kernel, dim3(1), dim3(64), 0, 0, of, oi, om);
  HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
  std::vector<float> f(320); std::vector<int> i(192); std::vector<unsigned long long> m(64);
  HIPCHK(hipMemcpy(f.data(), of, 320 * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(i.data(), oi, 192 * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(m.data(), om, 64 * 8, hipMemcpyDeviceToHost));
  hipFree(of); hipFree(oi); hipFree(om);
  float x[64]; double sum = 0; float mn = 1e30f, mx = -1e30f; int isum = 0; unsigned long long bm = 0;
  for (int l = 0; l < 64; l++) { x[l] = (float)(l * l % 17) * 0.25f - 1.0f; sum += x[l]; float y = x[l] + (float)((l * 7) % 5); if (y < mn) mn = y; if (x[l] > mx) mx = x[l]; isum += l % 7; if (l % 3 == 1) bm |= 1ull << l; }
  int bad = 0, scan = 0;
  for (int l = 0; l < 64; l++) {
    if (fabs(f[l] - (float)sum) > 1e-4f) bad |= 1;
    if (f[64 + l] != mn) bad |= 2;
    if (f[128 + l] != x[37]) bad |= 4;
    if (f[192 + l] != x[(l * 13 + 5) & 63]) bad |= 8;
    if (i[l] != isum) bad |= 16;
    if (i[64 + l] != scan) bad |= 32;
    scan += l % 5;
    if (m[l] != bm) bad |= 64;
    if (i[128 + l] != __builtin_popcountll(bm & ((1ull << l) - 1ull))) bad |= 128;
    if (f[256 + l] != mx) bad |= 256;
  }
  if (bad) { char b[96]; snprintf(b, sizeof b, "agx_selftest: wave primitive mismatch, mask 0x%x", bad); return fail(AGX_E_HIP, b); }
  return AGX_OK;
}

}  // extern "C"
