// agx_kernels.hip -- the kernels of ONE variant of the stepper for gfx950.  Built once per variant by
// assistive_gym_amd/build.py (-DAGX_VARIANT_FEEDING / -DAGX_VARIANT_BED_BATHING select the limits and the task layer);
// agx_api.hip picks the variant whose task and limits fit the model blob.
// One workgroup = one wavefront = one environment (64 threads); an env.step() is
// frame_skip x [build kernel, solve kernel] + finish kernel on one stream.
#if defined(AGX_VARIANT_BED_BATHING)
// BedBathingSawyer: 10 Sawyer DoFs + the 10 joints of the human's right arm (dynamic when the impairment is tremor), one free body (wiper)
#define AGX_MAX_DOF 20
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 10
#define AGX_TASK 1
#define AGX_VNAME bed_bathing
#define AGX_K(name) name##_bb
#elif defined(AGX_VARIANT_BED_BATHING_L)
// bed bathing with a robot of up to 12 dynamic joints (PR2: 7 arm + 4 finger joints): the limits of the scratch_itch variant
#define AGX_MAX_DOF 24
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 12
#define AGX_ARENA_WORDS 4096
#define AGX_TASK 1
#define AGX_VNAME bed_bathing_l
#define AGX_K(name) name##_bbl
#elif defined(AGX_VARIANT_BED_SETTLE)
// the rag-doll settle of BedBathingEnv.reset (bed_bathing.py:119-137): the whole human as ONE articulated body of 47 DoFs (6 for the
// floating base + 41 joints) falling onto the bed; reset time only (65 KB of LDS per environment -- two per CU -- since the M^-1 columns are computed
// in two batches of 24 lanes, agx_ctx.h COLS_LANES; 97 KB and one per CU before round 4)
#define AGX_MAX_DOF 48
#define AGX_MAX_FREE 1
#define AGX_MAX_BLOCK 48
#define AGX_ARENA_WORDS 11968
#define AGX_SCR_ENT 16384
#define AGX_TASK 1
#define AGX_VNAME bed_settle
#define AGX_K(name) name##_bs
#elif defined(AGX_VARIANT_SCRATCH_ITCH)
// ScratchItchPR2: the PR2's left arm branch (7 arm joints + 4 finger joints; its other branches start at rest with zero gravity and are
// compiled as static) + the 10 joints of the human's right arm, one free body (scratcher)
#define AGX_MAX_DOF 24
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 12
#define AGX_ARENA_WORDS 4096
#define AGX_TASK 2
#define AGX_VNAME scratch_itch
#define AGX_K(name) name##_si
#elif defined(AGX_VARIANT_DRESSING_L)
// DressingPR2: 7 arm + 4 finger joints + the 10 joints of the human's left arm; plus the cloth kernel
#define AGX_MAX_DOF 24
#define AGX_MAX_FREE 1
#define AGX_MAX_BLOCK 12
#define AGX_ARENA_WORDS 4096
#define AGX_TASK 3
#define AGX_VNAME dressing_l
#define AGX_K(name) name##_drl
#elif defined(AGX_VARIANT_DRESSING)
// DressingBaxter: Baxter's left arm (7 arm joints + 2 finger joints; the rest of the robot static) + the 10 joints of the human's left
// arm, no free body; plus the cloth kernel (agx_cloth.h)
#define AGX_MAX_DOF 20
#define AGX_MAX_FREE 1
#define AGX_MAX_BLOCK 10
#define AGX_TASK 3
#define AGX_VNAME dressing
#define AGX_K(name) name##_dr
#elif defined(AGX_VARIANT_ARM_MANIPULATION)
// ArmManipulationSawyer: 10 Sawyer DoFs + the 10 joints of the human's right arm (always dynamic: it hangs limp beside the bed), one free
// body (the scooper, 12 hulls)
#define AGX_MAX_DOF 20
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 10
#define AGX_TASK 4
#define AGX_VNAME arm_manipulation
#define AGX_K(name) name##_am
#elif defined(AGX_VARIANT_ARM_MANIPULATION_L)
// arm manipulation with a two-armed robot (PR2: 2 x (7 arm + 4 finger joints); Baxter: 2 x 9): both arms are ONE articulated body of up to
// 22 DoFs, plus the 10 joints of the human's arm and two free bodies (tool_right, tool_left).  40 KB of LDS per environment in the build kernel.
#define AGX_MAX_DOF 32
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 22
#define AGX_ARENA_WORDS 7552
#define AGX_TASK 4
#define AGX_VNAME arm_manipulation_l
#define AGX_K(name) name##_aml
#elif defined(AGX_VARIANT_FEEDING_L)
// the feeding scene with a free-standing robot (FeedingSawyer, FeedingBaxter, FeedingPR2): the pedestal / torso / other arm add up to 320 colliders
#define AGX_MAX_COLL 320
#define AGX_MAX_BLOCK 12      // the PR2's arm: 7 arm + 4 finger joints
#define AGX_ARENA_WORDS 4040
#define AGX_VNAME feeding_l
#define AGX_K(name) name##_fl
#elif defined(AGX_VARIANT_BED_BATHING_M)
// bed bathing with the mobile manipulator (BedBathingStretch): 16 robot DoFs on a floating base + the 10 joints of the human's arm
#define AGX_MAX_DOF 28
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 16
#define AGX_ARENA_WORDS 5632
#define AGX_TASK 1
#define AGX_VNAME bed_bathing_m
#define AGX_K(name) name##_bbm
#elif defined(AGX_VARIANT_SCRATCH_ITCH_M)
// ScratchItchStretch: as bed_bathing_m with the scratch-itch task layer
#define AGX_MAX_DOF 28
#define AGX_MAX_FREE 2
#define AGX_MAX_BLOCK 16
#define AGX_ARENA_WORDS 5632
#define AGX_TASK 2
#define AGX_VNAME scratch_itch_m
#define AGX_K(name) name##_sim
#elif defined(AGX_VARIANT_DRESSING_M)
// DressingStretch: 16 robot DoFs on a floating base + the 10 joints of the human's left arm; plus the cloth kernel
#define AGX_MAX_DOF 28
#define AGX_MAX_FREE 1
#define AGX_MAX_BLOCK 16
#define AGX_ARENA_WORDS 5632
#define AGX_TASK 3
#define AGX_VNAME dressing_m
#define AGX_K(name) name##_drm
#elif defined(AGX_VARIANT_FEEDING_M)
// the feeding scene with a mobile manipulator (FeedingStretch): a floating base (6 virtual joints) + 2 wheels + lift + 4 telescoping joints +
// wrist + 2 fingers = 16 DoFs in ONE articulated body, plus the 4 head joints
#define AGX_MAX_DOF 20
#define AGX_MAX_BLOCK 16
#define AGX_MAX_COLL 320
#define AGX_ST_WORDS 344
#define AGX_ARENA_WORDS 4040
#define AGX_VNAME feeding_m
#define AGX_K(name) name##_fm
#elif defined(AGX_VARIANT_DRINKING_L)
// the drinking scene with the PR2's arm: 7 arm + 4 finger joints in one articulated body
#define AGX_MAX_FREE 1
#define AGX_MAX_BLOCK 12
#define AGX_ARENA_WORDS 4040
#define AGX_TASK 5
#define AGX_VNAME drinking_l
#define AGX_K(name) name##_dkl
#elif defined(AGX_VARIANT_DRINKING_M)
// the drinking scene with the mobile manipulator (DrinkingStretch): as feeding_m -- 16 DoFs on a floating base + the 4 head joints
#define AGX_MAX_FREE 1
#define AGX_MAX_DOF 20
#define AGX_MAX_BLOCK 16
#define AGX_ARENA_WORDS 4040
#define AGX_TASK 5
#define AGX_VNAME drinking_m
#define AGX_K(name) name##_dkm
#elif defined(AGX_VARIANT_DRINKING)
// DrinkingJaco: the feeding scene's robot and person, the cup as the one free body (68 hulls), no food; plus the water kernel (agx_water.h)
#define AGX_MAX_FREE 1
#define AGX_TASK 5
#define AGX_VNAME drinking
#define AGX_K(name) name##_dk
#elif defined(AGX_VARIANT_FEEDING)
#define AGX_VNAME feeding
#define AGX_K(name) name
#else
#error "build with -DAGX_VARIANT_<name>, see assistive_gym_amd/build.py"
#endif
// the variants that also carry the build kernel with the persistent-manifold stage (AGX_P_MANIFOLD; agx_env.h env_build<true>): the four
// tasks whose rewards read contact forces, with their default robots
#if defined(AGX_VARIANT_FEEDING) || defined(AGX_VARIANT_BED_BATHING) || defined(AGX_VARIANT_SCRATCH_ITCH) || defined(AGX_VARIANT_ARM_MANIPULATION)
#define AGX_HAS_MANIFOLD 1
#else
#define AGX_HAS_MANIFOLD 0
#endif

#include "agx_wave.h"
#include "agx_step.h"
#include "agx_variant.h"
#if AGX_TASK == 3
#include "agx_cloth.h"
#endif

namespace {

// build: kinematics, ABA, collision, constraint rows -> scratch.  Register- and LDS-heavy.
extern "C" __global__ void __launch_bounds__(64, 2)
AGX_K(agx_build_kernel)(const uint32_t* __restrict__ blob, float* state, const float* actions, float* scratch, float* debug, int env0, int n_envs, int sw, int act_dim,
                        const uint8_t* __restrict__ active, int* overflow_total, float* trace, int trace_words, int phase) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = env0 + blockIdx.x;
  if (env >= n_envs || (active && !active[env])) return;   // `active`: masked settle of agx_reset, null on the step path
  const int dropped = agx::env_build(blob, state + (size_t)env * sw, actions ? actions + (size_t)env * act_dim : nullptr, scratch + (size_t)env * agx::SCR_WORDS,
                                     debug ? debug + (size_t)env * agx::DBG_WORDS : nullptr, lds, (int)threadIdx.x,
                                     trace ? trace + (size_t)env * trace_words + (size_t)phase * 12 * (((const int*)blob)[AGX_H_NDOF] + ((const int*)blob)[AGX_H_NFREE]) : nullptr);
  if (dropped > 0 && threadIdx.x == 0) atomicAdd(overflow_total, dropped);   // contacts dropped by a budget (rare; agx_overflow_count)
}
#if AGX_HAS_MANIFOLD
// build with the persistent-manifold stage between collision and rows (blobs with AGX_P_MANIFOLD > 0 only)
extern "C" __global__ void __launch_bounds__(64, 2)
AGX_K(agx_build_mf_kernel)(const uint32_t* __restrict__ blob, float* state, const float* actions, float* scratch, float* debug, int env0, int n_envs, int sw, int act_dim,
                           const uint8_t* __restrict__ active, int* overflow_total, float* trace, int trace_words, int phase) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = env0 + blockIdx.x;
  if (env >= n_envs || (active && !active[env])) return;
  const int dropped = agx::env_build<true>(blob, state + (size_t)env * sw, actions ? actions + (size_t)env * act_dim : nullptr, scratch + (size_t)env * agx::SCR_WORDS,
                                           debug ? debug + (size_t)env * agx::DBG_WORDS : nullptr, lds, (int)threadIdx.x,
                                           trace ? trace + (size_t)env * trace_words + (size_t)phase * 12 * (((const int*)blob)[AGX_H_NDOF] + ((const int*)blob)[AGX_H_NFREE]) : nullptr);
  if (dropped > 0 && threadIdx.x == 0) atomicAdd(overflow_total, dropped);
}
#endif
// solve: 50 PGS sweeps streaming the rows from the scratch record (L2), integration.  Lean.
extern "C" __global__ void __launch_bounds__(64, 4)
AGX_K(agx_solve_kernel)(const uint32_t* __restrict__ blob, float* state, float* scratch, float* debug, int env0, int n_envs, int sw, const uint8_t* __restrict__ active, int phase,
                        int lds_words) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = env0 + blockIdx.x;
  if (env >= n_envs || (active && !active[env])) return;
  agx::env_solve(blob, state + (size_t)env * sw, scratch + (size_t)env * agx::SCR_WORDS, debug ? debug + (size_t)env * agx::DBG_WORDS : nullptr, lds, (int)threadIdx.x, phase,
                 lds_words);
}
// finish: forces, observation, task state machine, reward, done, info
extern "C" __global__ void __launch_bounds__(64, 2)
AGX_K(agx_finish_kernel)(const uint32_t* __restrict__ blob, float* state, const float* actions, float* scratch, float* obs, float* reward, uint8_t* done,
                         float* info, int env0, int n_envs, int sw, int act_dim, int obs_dim, const float* report, int report_words, float* cloth, int cloth_words) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = env0 + blockIdx.x;
  if (env >= n_envs) return;
  agx::env_finish(blob, state + (size_t)env * sw, actions + (size_t)env * act_dim, scratch + (size_t)env * agx::SCR_WORDS, obs + (size_t)env * obs_dim,
                  reward + env, done + env, info ? info + (size_t)env * AGX_INFO_COUNT : nullptr, lds, (int)threadIdx.x,
                  report ? report + (size_t)env * report_words : nullptr, cloth ? cloth + (size_t)env * cloth_words : nullptr);
}
#if AGX_TASK == 5
// the water: one wavefront per environment, lane = particle (agx_water.h); replays the frames the build kernels left in the trace
extern "C" __global__ void __launch_bounds__(64)
AGX_K(agx_water_kernel)(const uint32_t* __restrict__ blob, const float* __restrict__ state, const float* __restrict__ trace, float* water, float* report, int env0, int n_envs,
                        int sw, int trace_words, int cloth_words, int report_words, int nsub, const uint8_t* __restrict__ active) {
  __shared__ __attribute__((aligned(16))) float lds[agxw::LDS_WORDS];
  const int env = env0 + blockIdx.x;
  if (env >= n_envs || (active && !active[env])) return;
  agxw::water_env(blob, state + (size_t)env * sw, trace + (size_t)env * trace_words, water + (size_t)env * cloth_words, report ? report + (size_t)env * report_words : nullptr, nsub,
                  lds, (int)threadIdx.x);
}
#endif
#if AGX_TASK == 3
// the garment: one workgroup of AGX_CLOTH_THREADS threads per environment, positions resident in LDS for all substeps of the launch
extern "C" __global__ void __launch_bounds__(AGX_CLOTH_THREADS)
AGX_K(agx_cloth_kernel)(const uint32_t* __restrict__ blob, const float* __restrict__ state, const float* __restrict__ trace, float* cloth, float* report, int env0, int n_envs,
                        int sw, int trace_words, int cloth_words, int report_words, int nsub, const uint8_t* __restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = env0 + blockIdx.x;
  if (env >= n_envs || (active && !active[env])) return;
  agxc::cloth_env(blob, state + (size_t)env * sw, trace + (size_t)env * trace_words, cloth + (size_t)env * cloth_words, report ? report + (size_t)env * report_words : nullptr, nsub, lds);
}
#endif
extern "C" __global__ void __launch_bounds__(64, 2)
AGX_K(agx_observe_kernel)(const uint32_t* __restrict__ blob, float* state, float* obs, int n_envs, int sw, int obs_dim, const uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = blockIdx.x;
  if (env >= n_envs || (mask && !mask[env])) return;
  agx::env_observe(blob, state + (size_t)env * sw, obs + (size_t)env * obs_dim, lds, (int)threadIdx.x);
}
// after a build-kernel pass: the collision flags of every environment's state (agx_check_collisions)
extern "C" __global__ void __launch_bounds__(64)
AGX_K(agx_collision_flags_kernel)(const uint32_t* __restrict__ blob, const float* __restrict__ scratch, uint8_t* flags, int n_envs) {
  const int env = blockIdx.x;
  if (env >= n_envs) return;
  const int f = agx::collision_flags(blob, scratch + (size_t)env * agx::SCR_WORDS, (int)threadIdx.x);
  if (threadIdx.x == 0) flags[env] = (uint8_t)f;
}
#if AGX_HAS_SAMPLER
// reset generator: FeedingEnv.reset's sampling incl. the IK restarts (64 per round, one per lane), float64
extern "C" __global__ void __launch_bounds__(64)
AGX_K(agx_sample_kernel)(const uint32_t* __restrict__ blob, float* state, unsigned long long seed0, const unsigned long long* __restrict__ seeds, const uint8_t* __restrict__ mask,
                         int impairment_mode, int gender_mode, float* info4, int* episode, int n_envs, int sw, const int* __restrict__ first_restart, int* chosen,
                         const float* __restrict__ settled, int settled_sw, const float* __restrict__ fell) {
  const int env = blockIdx.x;
  if (env >= n_envs || (mask && !mask[env])) return;
  const unsigned long long seed = seeds ? seeds[env] : seed0 + (unsigned long long)env;
  if (threadIdx.x == 0) episode[env] = 0;
  const int r = agx::env_sample(blob, state + (size_t)env * sw, (uint32_t)seed, (uint32_t)(seed >> 32), impairment_mode, gender_mode,
                                info4 ? info4 + (size_t)env * 4 : nullptr, (int)threadIdx.x, first_restart ? first_restart[env] : 0,
                                settled ? settled + (size_t)env * settled_sw : nullptr, fell ? fell + (size_t)env * sw : nullptr);
  if (chosen && threadIdx.x == 0) chosen[env] = r;
}
// after a build-kernel pass over the freshly sampled states: which of them start in collision (and have a restart left to try)?
// work[env] = 1 and first_restart[env] = chosen + 1 for those, work[env] = 0 for the others
extern "C" __global__ void __launch_bounds__(64)
AGX_K(agx_reset_verdict_kernel)(const uint32_t* __restrict__ blob, const float* __restrict__ scratch, const uint8_t* __restrict__ active, uint8_t* work,
                                int* first_restart, const int* __restrict__ chosen, int n_envs) {
  const int env = blockIdx.x;
  if (env >= n_envs) return;
  bool again = false;
  if ((!active || active[env]) && chosen[env] >= 0) again = agx::reset_collides(blob, scratch + (size_t)env * agx::SCR_WORDS, (int)threadIdx.x);
  if (threadIdx.x == 0) { work[env] = again ? 1 : 0; if (again) first_restart[env] = chosen[env] + 1; }
}
#endif

// LDS of a solve launch.  AGX_SOLVE_LDS_BYTES (tuning knob, read once): the row-local sweep (agx_pgs_lvs.h, agx_pgs_lv.h) sizes its window of resident rows
// from it -- more LDS = fewer rows streamed from L2, fewer wavefronts per CU; never below what the other sweeps and solve_tail() need
constexpr int SWEEP_LDS_BYTES = agx::LVS_COMPILED ? agx::LVS_SOLVE_LDS_BYTES : (agx::LV_COMPILED ? agx::LV_SOLVE_LDS_BYTES : 0);      // what the row-local sweep of this variant wants (agx_pgs_lvs.h / agx_pgs_lv.h)
int g_solve_lds_bytes = SWEEP_LDS_BYTES > agx::LDS_SOLVE_BYTES ? SWEEP_LDS_BYTES : agx::LDS_SOLVE_BYTES;
hipError_t v_init(void) {
  if (const char* e = getenv("AGX_SOLVE_LDS_BYTES")) { int b = atoi(e) & ~15; if (b >= agx::LDS_SOLVE_BYTES && b <= 64 * 1024) g_solve_lds_bytes = b; }
  hipError_t e = hipFuncSetAttribute((const void*)AGX_K(agx_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, g_solve_lds_bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)AGX_K(agx_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, agx::LDS_BYTES);
#if AGX_HAS_MANIFOLD
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)AGX_K(agx_build_mf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, agx::LDS_BYTES);
#endif
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)AGX_K(agx_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, agx::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)AGX_K(agx_observe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, agx::LDS_BYTES);
#if AGX_TASK == 3
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)AGX_K(agx_cloth_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
  return e;
}
void v_build(hipStream_t st, int ne, const uint32_t* blob, float* state, const float* actions, float* scratch, float* debug, int e0, int n_envs, int sw, int act_dim,
             const uint8_t* active, int* overflow_total, float* trace, int trace_words, int phase) {
  hipLaunchKernelGGL(AGX_K(agx_build_kernel), dim3(ne), dim3(64), agx::LDS_BYTES, st, blob, state, actions, scratch, debug, e0, n_envs, sw, act_dim, active, overflow_total,
                     trace, trace_words, phase);
}
#if AGX_HAS_MANIFOLD
void v_build_mf(hipStream_t st, int ne, const uint32_t* blob, float* state, const float* actions, float* scratch, float* debug, int e0, int n_envs, int sw, int act_dim,
                const uint8_t* active, int* overflow_total, float* trace, int trace_words, int phase) {
  hipLaunchKernelGGL(AGX_K(agx_build_mf_kernel), dim3(ne), dim3(64), agx::LDS_BYTES, st, blob, state, actions, scratch, debug, e0, n_envs, sw, act_dim, active, overflow_total,
                     trace, trace_words, phase);
}
#endif
void v_solve(hipStream_t st, int ne, const uint32_t* blob, float* state, float* scratch, float* debug, int e0, int n_envs, int sw, const uint8_t* active, int phase) {
  hipLaunchKernelGGL(AGX_K(agx_solve_kernel), dim3(ne), dim3(64), g_solve_lds_bytes, st, blob, state, scratch, debug, e0, n_envs, sw, active, phase, g_solve_lds_bytes / 4);
}
void v_finish(hipStream_t st, int ne, const uint32_t* blob, float* state, const float* actions, float* scratch, float* obs, float* reward, uint8_t* done, float* info,
              int e0, int n_envs, int sw, int act_dim, int obs_dim, const float* report, int report_words, float* cloth, int cloth_words) {
  hipLaunchKernelGGL(AGX_K(agx_finish_kernel), dim3(ne), dim3(64), agx::LDS_BYTES, st, blob, state, actions, scratch, obs, reward, done, info, e0, n_envs, sw, act_dim, obs_dim,
                     report, report_words, cloth, cloth_words);
}
#if AGX_TASK == 5
void v_cloth(hipStream_t st, int ne, const uint32_t* blob, const float* state, const float* trace, float* cloth, float* report, int e0, int n_envs, int sw,
             int trace_words, int cloth_words, int report_words, int nsub, const uint8_t* active, int lds_bytes) {
  (void)lds_bytes;
  hipLaunchKernelGGL(AGX_K(agx_water_kernel), dim3(ne), dim3(64), 0, st, blob, state, trace, cloth, report, e0, n_envs, sw, trace_words, cloth_words, report_words, nsub, active);
}
int v_cloth_lds_bytes(int nn) { (void)nn; return 4 * agxw::LDS_WORDS; }
#endif
#if AGX_TASK == 3
void v_cloth(hipStream_t st, int ne, const uint32_t* blob, const float* state, const float* trace, float* cloth, float* report, int e0, int n_envs, int sw,
             int trace_words, int cloth_words, int report_words, int nsub, const uint8_t* active, int lds_bytes) {
  hipLaunchKernelGGL(AGX_K(agx_cloth_kernel), dim3(ne), dim3(AGX_CLOTH_THREADS), lds_bytes, st, blob, state, trace, cloth, report, e0, n_envs, sw, trace_words, cloth_words,
                     report_words, nsub, active);
}
#endif
void v_observe(hipStream_t st, int n_envs, const uint32_t* blob, float* state, float* obs, int sw, int obs_dim, const uint8_t* mask) {
  hipLaunchKernelGGL(AGX_K(agx_observe_kernel), dim3(n_envs), dim3(64), agx::LDS_BYTES, st, blob, state, obs, n_envs, sw, obs_dim, mask);
}
#if AGX_HAS_SAMPLER
void v_sample(hipStream_t st, int n_envs, const uint32_t* blob, float* state, unsigned long long seed0, const unsigned long long* seeds, const uint8_t* mask,
              int impairment_mode, int gender_mode, float* info4, int* episode, int sw, const int* first_restart, int* chosen, const float* settled, int settled_sw, const float* fell) {
  hipLaunchKernelGGL(AGX_K(agx_sample_kernel), dim3(n_envs), dim3(64), 0, st, blob, state, seed0, seeds, mask, impairment_mode, gender_mode, info4, episode, n_envs, sw,
                     first_restart, chosen, settled, settled_sw, fell);
}
void v_verdict(hipStream_t st, int n_envs, const uint32_t* blob, const float* scratch, const uint8_t* active, uint8_t* work, int* first_restart, const int* chosen) {
  hipLaunchKernelGGL(AGX_K(agx_reset_verdict_kernel), dim3(n_envs), dim3(64), 0, st, blob, scratch, active, work, first_restart, chosen, n_envs);
}
#endif

#if AGX_TASK == 3
int v_cloth_lds_bytes(int nn) { return 4 * agxc::lds_words(nn); }
#endif
void v_collision_flags(hipStream_t st, int n_envs, const uint32_t* blob, const float* scratch, uint8_t* flags) {
  hipLaunchKernelGGL(AGX_K(agx_collision_flags_kernel), dim3(n_envs), dim3(64), 0, st, blob, scratch, flags, n_envs);
}

#define AGX_STR2_(x) #x
#define AGX_STR_(x) AGX_STR2_(x)
const agx_variant g_variant = {
  AGX_STR_(AGX_VNAME), agx::TASK,
  agx::MAX_DOF, agx::MAX_FREE, agx::MAX_BLOCK, agx::MAX_HUMAN, agx::MAX_COLL, agx::ST_WORDS, agx::MAX_CON, agx::MAX_ROWS,
  agx::LDS_BYTES, agx::LDS_SOLVE_BYTES, agx::SCR_WORDS, agx::DBG_WORDS,
  agx::DBG_CON, agx::DBG_MINV, agx::DBG_HDR, agx::DBG_LAM, agx::DBG_TIME, agx::DBG_QDD,
#if AGX_HAS_SAMPLER
  agx::RS_NARM,
#else
  0,
#endif
  v_init, v_build, v_solve,
#if AGX_HAS_MANIFOLD
  v_build_mf,
#else
  nullptr,
#endif
  v_finish, v_observe,
#if AGX_HAS_SAMPLER
  v_sample,
#else
  nullptr,
#endif
#if AGX_TASK == 3 || AGX_TASK == 5
  v_cloth,
#else
  nullptr,
#endif
#if AGX_HAS_SAMPLER
  v_verdict,
#else
  nullptr,
#endif
  v_collision_flags,
#if AGX_TASK == 3 || AGX_TASK == 5
  v_cloth_lds_bytes,
#else
  nullptr,
#endif
  agx::SCR_O_META + agx::META_NWARM
};

}  // namespace

#define AGX_CAT2_(a, b) a##b
#define AGX_CAT_(a, b) AGX_CAT2_(a, b)
extern "C" const agx_variant* AGX_CAT_(agx_variant_, AGX_VNAME)(void) { return &g_variant; }
