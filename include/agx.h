/* agx.h -- C ABI of libagx: the MI355X-native batched stepper behind Assistive Gym's
 * env.step() hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *
 *   agx_create        p.connect(p.DIRECT) + the world build of reset()
 *                     (assistive_gym/envs/env.py:34,91-134; envs/feeding.py:114-172), for N envs.
 *                     The world is handed over as a compiled model blob (include/agx_blob.h).
 *   agx_set_state /   no reference equivalent (the reference has no p.saveState/restoreState,
 *   agx_get_state     SURVEY section 5): state injection for parity tests, checkpoints, reset pools.
 *   agx_settle        the settle loop `for _ in range(25): p.stepSimulation()` (feeding.py:178-179).
 *   (the citations are those of the FeedingJaco model; a BedBathingSawyer blob is served by the bed_bathing kernel
 *    variant, whose task layer replaces BedBathingEnv.step/_get_obs/get_total_force/update_targets,
 *    assistive_gym/envs/bed_bathing.py:12-110,190-203, and the non-feeding branch of human_preferences, env.py:244-247)
 *   agx_step          FeedingEnv.step(action) for every env: AssistiveEnv.take_step
 *                     (envs/env.py:174-235) incl. 5x p.stepSimulation() (env.py:226), _get_obs
 *                     (feeding.py:85-112), get_food_rewards (feeding.py:50-83), human_preferences
 *                     (env.py:237-274), reward / done / info (feeding.py:25-37).
 *   agx_observe       the `return self._get_obs()` of reset() (feeding.py:182).
 *   agx_sample_reset  FeedingEnv.reset() up to its settle loop (feeding.py:114-177) for every env, on the device:
 *                     Human.init draws (agents/human.py:72-92), the posed human (feeding.py:124-126), the mouth
 *                     target (feeding.py:184-196), init_robot_pose -> Robot.ik_random_restarts (env.py:276-310,
 *                     agents/robot.py:84-121, incl. the rejection of IK solutions that touch the human / table / wheelchair,
 *                     robot.py:105-112, env.py:299-308), tool / bowl / food placement (feeding.py:143-166).
 *   agx_reset         the same followed by the settle loop, for the envs selected by a mask (a caller resetting
 *                     the envs that are done), per-env seeds optional.
 *   agx_reset_done    gym's TimeLimit/auto-reset on done (assistive_gym/__init__.py:11), drawing
 *                     the new post-reset state from a caller-provided pool.
 *
 * All `*_dev` pointers are DEVICE pointers (HBM) owned by the caller; `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Calls on one handle must come from one thread at a
 * time; handles are independent.  Every function returns 0 on success or a negative AGX_E_* code
 * and never calls exit(); agx_last_error() describes the last failure of the calling thread.
 */
#ifndef AGX_H
#define AGX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct agx_handle_s* agx_handle;

enum { AGX_OK = 0, AGX_E_ARG = -1, AGX_E_BLOB = -2, AGX_E_HIP = -3, AGX_E_NOGPU = -4, AGX_E_LIMIT = -5 };

#define AGX_INFO_DIM 8   /* see AGX_INFO_* in agx_blob.h */

/* library / build information */
const char* agx_version(void);
const char* agx_last_error(void);
int agx_device_count(void);
int agx_lds_bytes_per_env(void);

int agx_create(const void* blob, size_t blob_bytes, int n_envs, int device, agx_handle* out);
void agx_destroy(agx_handle h);
int agx_dims(agx_handle h, int* n_envs, int* act_dim, int* obs_dim, int* state_words);

/* host <-> device copies of the per-env state records, [n_envs][state_words] float32 */
int agx_set_state(agx_handle h, const float* host_states);
int agx_get_state(agx_handle h, float* host_states);
/* device address of the state records (e.g. to fill them from a device-resident pool) */
int agx_state_dev(agx_handle h, float** out_dev);

/* n_substeps p.stepSimulation() calls (each AGX_H_SIM_SUBSTEPS internal substeps) without actions, task layer or hooks of env.step */
int agx_settle(agx_handle h, int n_substeps, void* stream);
int agx_step(agx_handle h, const float* actions_dev, float* obs_dev, float* reward_dev,
             uint8_t* done_dev, float* info_dev, void* stream);
/* same as agx_step, additionally dumping first-substep internals ([n_envs][agx_debug_words()]) */
int agx_step_debug(agx_handle h, const float* actions_dev, float* obs_dev, float* reward_dev,
                   uint8_t* done_dev, float* info_dev, float* debug_dev, void* stream);
/* agx_settle, additionally dumping first-substep internals (models without a task layer to finish a step with, e.g. bed_settle) */
int agx_settle_debug(agx_handle h, int n_substeps, float* debug_dev, void* stream);
/* models with a cloth section (DressingBaxter; assistive_gym/envs/dressing.py:149-157, p.loadCloth / p.getSoftBodyData): the garment of
 * every environment is a float[2][nodes][3] record next to its state record -- node positions, node velocities.  agx_step / agx_settle
 * advance it; agx_reset_done also replaces the garment of a finished environment, from the device array [pool_n][2][nodes][3] given once
 * with agx_set_cloth_pool (same pool index as the state record). */
int agx_cloth_nodes(agx_handle h, int* nodes);   /* 0 for models without a cloth */
int agx_set_cloth(agx_handle h, const float* host_cloth);
int agx_get_cloth(agx_handle h, float* host_cloth);
int agx_cloth_dev(agx_handle h, float** out_dev);
int agx_set_cloth_pool(agx_handle h, const float* pool_cloth_dev);
/* models with a cloth: copies the cloth kernel's report of the LAST agx_step to the host -- per environment float[18] the six sleeve vertices
 * (util.py:156-157), 2 unused, then per node and contact slot {height of the node, |contact force| of the last substep, -1 = no contact}
 * (what dressing.py:25,35-43 reads from getSoftBodyData; AGX_CLOTH_REPORT_WORDS in agx_blob.h).  words_per_env (may be NULL) receives the row
 * length; host_report NULL = query the length only.  Synchronises the device.  For parity tests of the cloth-force term (which node contacts
 * are in the sum), not on the step path. */
int agx_get_cloth_report(agx_handle h, float* host_report, int* words_per_env);
int agx_debug_words(void);   /* of the FeedingJaco kernel variant; agx_debug_layout for the variant serving a handle */
/* layout of the debug record of the kernel variant serving this handle: out8 = {words per env, contacts offset, M^-1 offset,
 * M^-1 row stride, row headers offset, impulses offset, phase timers offset, qdd offset} */
int agx_debug_layout(agx_handle h, int* out8);
/* name of the compiled kernel variant (limits + task layer) that serves this handle: "feeding", "bed_bathing" */
const char* agx_variant_name(agx_handle h);
/* contacts dropped since agx_create because a budget was exceeded (contact, row or coefficient cap), summed over all envs:
 * they are missing from the dynamics and from total_force_on_human; 0 in a healthy run */
int agx_overflow_count(agx_handle h, int* out);
/* same as agx_step (same chunk streams) with a HIP event after every kernel launch; blocks until the step
 * is done and returns the summed launch durations of one step in ms3 = {build, solve, finish kernels} and
 * the number of launches of each in launches3 (may be NULL) */
int agx_step_timed(agx_handle h, const float* actions_dev, float* obs_dev, float* reward_dev,
                   uint8_t* done_dev, float* info_dev, void* stream, float* ms3, int* launches3);
int agx_observe(agx_handle h, float* obs_dev, void* stream);
/* ... for the environments with mask_dev[i] != 0 only (the rows of the others are left as they are); mask_dev NULL = all.
 * The first observation of environments replaced mid-batch by agx_reset_done / agx_reset */
int agx_observe_masked(agx_handle h, float* obs_dev, const uint8_t* mask_dev, void* stream);
/* Overwrites the state record of EVERY env of the handle with a freshly sampled pre-settle reset state; env i
 * is a pure function of (seed + i) (counter-based Philox4x32-10 draws, so the result does not depend on
 * how envs are spread over handles or GPUs).  impairment_mode: -1 = random over none/limits/weakness/tremor
 * (human.py:80), -2 = random without tremor, 0..3 = fixed; gender_mode: -1 random, 0 male, 1 female.
 * ik_info_dev (may be NULL): [n_envs][4] float = {IK met the thresholds, restarts used, position error,
 * impairment drawn}; for a free-standing robot, whose base pose is searched as Robot.position_robot_toc does
 * (robot.py:123-215: AGX_X_TOC_ATTEMPTS candidate poses per round, one per lane): {a candidate reached the start
 * pose, rounds used, goals reached by the chosen candidate incl. the start pose, impairment drawn}.
 * Models with a cloth: the garment is placed at the sampled end effector (dressing.py:146-153), at rest.
 * Follow with agx_settle(h, 25, stream) (feeding.py:178-179) and agx_observe -- or use agx_reset, which also
 * switches the garment of a dressing scene from its settle gravity to full gravity when the settle is over.
 * Also zeroes the handle's episode counters. */
int agx_sample_reset(agx_handle h, uint64_t seed, int impairment_mode, int gender_mode, float* ik_info_dev, void* stream);
/* reset() of a SUBSET of the envs (gym semantics: the caller resets the envs that are done): envs with
 * mask_dev[i] != 0 (NULL = all) are re-sampled as by agx_sample_reset -- from seeds_dev[i] if seeds_dev is
 * given, else from seed + i -- and then run `settle_substeps` substeps (25 in feeding.py:178-179) while all
 * other envs are left untouched.  Their episode counters restart at 0. */
int agx_reset(agx_handle h, const uint8_t* mask_dev, const uint64_t* seeds_dev, uint64_t seed, int impairment_mode,
              int gender_mode, int settle_substeps, void* stream);
/* Bed bathing (reference: BedBathingEnv.reset, assistive_gym/envs/bed_bathing.py:119-137): the human of such a model is a rag doll dropped
 * onto the bed and left to settle for 100 simulation steps before the rest of the reset is sampled.  `settle` is a second handle, created
 * by the caller on the rag-doll model blob (bed_settle) with the same number of environments on the same device; agx_sample_reset / agx_reset
 * of `h` then sample that model's drop records from the same seeds, settle them for n_substeps substeps and read the human's resting pose
 * from its state records.  The second handle is not owned (destroy it after `h`); a null `settle` detaches.  Without an attachment the
 * sampler of such a model refuses to run.
 * Arm manipulation (ArmManipulationEnv.reset, assistive_gym/envs/arm_manipulation.py:117-165): two settles.  `settle` is
 * then a handle on the task's FALL model (the same blob with HUMAN_GRAVITY_Z = -1 and AGX_X_FLAGS bit 8: its sampler writes the record the
 * posed arm falls from), n_substeps the length of the fall (100); the rag-doll handle is attached to the fall handle by a second call.
 * agx_sample_reset / agx_reset of `h` run rag doll, fall and the task's sampler in that order. */
int agx_attach_settle_model(agx_handle h, agx_handle settle, int n_substeps);

/* envs with done != 0 get a fresh state from pool_dev ([pool_n][state_words]); the pool entry is
 * (env_offset + env_index + 977 * episode_count) mod pool_n with env_offset = the global index of this handle's first env
 * (agx_set_env_offset, default 0), so results do not depend on how the envs are spread over GPUs */
int agx_reset_done(agx_handle h, const float* pool_dev, int pool_n, const uint8_t* done_dev, void* stream);
/* the same for an environment that ends OUTSIDE the batch's episode boundary (the non-finite guard of agx_step reports done = 1 mid-episode;
 * the reference's nearest analogue is the forced reconnect of env.py:93-97): the replacement state joins the lock-stepped batch at
 * `iteration` (env.py:185; < 0: keep the pool record's own), so that it reaches `done` (feeding.py:37) together with the others */
int agx_reset_done_at(agx_handle h, const float* pool_dev, int pool_n, const uint8_t* done_dev, int iteration, void* stream);
/* Collision rejection for resets sampled on the host (env.py:276-310 init_robot_pose, robot.py:103-108): runs the stepper's collision
 * pass on the states as they are (they are not advanced) and writes one byte of AGX_COLLIDE_* flags per environment to the HOST buffer
 * flags_host[n_envs].  Synchronises the stream. */
#ifndef AGX_COLLIDE_FLAGS
#define AGX_COLLIDE_FLAGS
enum { AGX_COLLIDE_ENV = 1,    /* a robot link or the tool touches (distance <= 0) the human or the furniture               */
       AGX_COLLIDE_SELF = 2 }; /* two robot links, or a robot link and the tool, interpenetrate by more than 1 cm           */
#endif
int agx_check_collisions(agx_handle h, uint8_t* flags_host, void* stream);
int agx_set_env_offset(agx_handle h, long long env_offset);

/* ---- whole-batch observation collation across GPUs: environments are sharded by contiguous index ranges, one
 * process per GPU; the only exchange is an all-gather of the [n_envs, obs_dim] shards over RCCL / xGMI, and only when one
 * consumer wants the whole batch.  RCCL is bound at run time (an instance already loaded into the process is reused).
 * agx_comm_unique_id: rank 0 obtains the 128-byte id and hands it to the other ranks by the host's own means (MPI, TCP, file);
 * agx_comm_init_rank: collective over all ranks; agx_allgather: gathered_dev[rank * floats_per_rank ...] = the shard of `rank`,
 * enqueued on `stream` (use a side stream and an event after agx_step to overlap it with the next step); comm == NULL = one rank. */
int agx_comm_unique_id(void* out128);
/* What is gathered: agx_pack_step writes packed_dev[n_envs][obs_dim + 4] = observation | reward | done (0 / 1) | total_force_on_human |
 * task_success per environment (the sampler's obs, reward, done, info: assistive_gym/learn.py:26,72), enqueued on `stream` after the agx_step
 * that produced them; agx_allgather of n_envs * (obs_dim + 4) floats then collates the whole batch in ONE collective per step. */
int agx_pack_step(agx_handle h, const float* obs_dev, const float* reward_dev, const uint8_t* done_dev, const float* info_dev, float* packed_dev, void* stream);
int agx_comm_init_rank(int device, int rank, int world, const void* unique_id128, void** comm_out);
int agx_comm_destroy(void* comm);
int agx_allgather(agx_handle h, const float* local_dev, float* gathered_dev, size_t floats_per_rank, void* comm, void* stream);

/* convenience wrappers with HOST buffers (copies included; not the timed path) */
int agx_step_host(agx_handle h, const float* actions, float* obs, float* reward, uint8_t* done, float* info);
int agx_observe_host(agx_handle h, float* obs);

/* HIP-event timing of the kernels launched between begin and end on `stream` */
int agx_profile_begin(agx_handle h, void* stream);
int agx_profile_end(agx_handle h, void* stream, float* elapsed_ms);

int agx_synchronize(agx_handle h, void* stream);

/* wave-primitive self test (DPP reductions, ballots, scans): returns 0 if the device matches the
 * host-computed expectations */
int agx_selftest(int device);

#ifdef __cplusplus
}
#endif
#endif
