/* agx_oracle.h -- CPU oracle (TEST INFRASTRUCTURE ONLY, not part of the product path).
 *
 * Double-precision, single-environment, deliberately plain restatement of the FeedingJaco
 * step():  AssistiveEnv.take_step (assistive_gym/envs/env.py:174-235) + the physics that
 * p.stepSimulation() performs for this scene + FeedingEnv.step/_get_obs/get_food_rewards
 * (assistive_gym/envs/feeding.py:12-112,192-196) + human_preferences (env.py:237-274).
 *
 * PARITY UNPINNED: the physics half of the reference lives in PyBullet / Bullet3 (Zackory fork,
 * unpinned, setup.py:21), which is not installable here and whose source is not on this box; the
 * reference ships no tests or golden vectors for this path.  The Bullet-side conventions restated
 * here are from the published algorithm descriptions (Featherstone ABA; projected Gauss-Seidel
 * sequential impulses; GJK) and are marked [BULLET-UNVERIFIED] where a Bullet default is assumed.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef AGX_ORACLE_H
#define AGX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct agxo_model agxo_model;

agxo_model* agxo_load(const uint32_t* blob, size_t nwords);
void agxo_free(agxo_model* m);
int agxo_state_words(const agxo_model* m);
int agxo_ndof(const agxo_model* m);

/* one full env step on one environment. state: float32 record (in/out). */
void agxo_step(const agxo_model* m, float* state, const float* action, float* obs, float* reward,
               int* done, float* info);
/* n physics substeps without action processing / rewards (reset-time settling, feeding.py:178-179) */
void agxo_settle(const agxo_model* m, float* state, int n_substeps);
/* models with a cloth section (DressingBaxter): the garment travels next to the state record as float[2][NN][3] (node positions,
 * node velocities); n_sim_steps counts p.stepSimulation() calls (SIM_SUBSTEPS internal substeps each).  cloth may be NULL: the
 * rigid scene alone is stepped */
int agxo_cloth_nodes(const agxo_model* m);
void agxo_step_cloth(const agxo_model* m, float* state, float* cloth, const float* action, float* obs, float* reward, int* done, float* info);
void agxo_settle_cloth(const agxo_model* m, float* state, float* cloth, int n_sim_steps);
int agxo_cloth_contacts(double* out, int max_out);
/* observation only (reset() return value, feeding.py:182) */
void agxo_observe(const agxo_model* m, const float* state, float* obs);

/* logit of the arm-limit classifier for the four remapped angles (class 1 <=> logit > 0), human.py:146 */
double agxo_arm_limit_logit(const agxo_model* m, const double* in4);

/* ---- building blocks exposed for unit tests ------------------------------------------------ */
/* link world frames: pos[ndof*3], rot[ndof*9] row-major */
void agxo_fk(const agxo_model* m, const float* state, double* pos, double* rot);
/* end-effector (PyBullet link 8) world pose */
void agxo_ee_pose(const agxo_model* m, const float* state, double* pos3, double* quat4);
/* joint-space mass matrix by CRBA, row-major ndof*ndof */
void agxo_crba(const agxo_model* m, const float* state, double* M);
/* unconstrained joint accelerations by ABA with joint torques tau (may be NULL = 0); damping on/off */
void agxo_aba(const agxo_model* m, const float* state, const double* tau, int with_damping, double* qdd);
/* bias term h(q,qd) = tau needed for zero acceleration (RNEA), damping excluded */
void agxo_rnea_bias(const agxo_model* m, const float* state, double* h);
/* M^-1 by ABA unit-impulse responses, row-major ndof*ndof */
void agxo_minv(const agxo_model* m, const float* state, double* Minv);
/* GJK distance between two point sets (cores). returns 0 separated, 1 penetrating */
int agxo_gjk(const double* a, int na, const double* b, int nb, double tol, int maxit,
             double* dist, double* pa, double* pb, int* iters);
/* contacts of the current state: out rows of 12 doubles
 * [colliderA, colliderB, pA(3), pB(3), n(3) (from B to A), distance]; returns count */
int agxo_collide(const agxo_model* m, const float* state, double* out, int max_out);
/* one substep returning the solved contact impulses (same row layout + impulse appended = 13) */
int agxo_rows_debug(const agxo_model* m, float* state, double* out, int max_out);
int agxo_substep_debug(const agxo_model* m, float* state, double* contacts_out, int max_out);

/* ---- world API (tests/refbridge): a persistent f64 simulation driven call by call through a PyBullet-shaped facade, so that the
 * reference's own Python half runs on this oracle's physics.  One agxo_world_step = one p.stepSimulation(), nothing else. */
typedef struct agxo_world agxo_world;
agxo_world* agxo_world_create(const agxo_model* m, const float* state, const float* cloth);
void agxo_world_store(agxo_world* w, float* state, float* cloth);
void agxo_world_free(agxo_world* w);
void agxo_world_joints(agxo_world* w, double* q, double* qd, double* qt);
void agxo_world_reset_joint(agxo_world* w, int dof, double q, double qd);
void agxo_world_set_target(agxo_world* w, int dof, double qt);
void agxo_world_set_free_base(agxo_world* w, int body, const double* pos3, const double* quat4);
void agxo_world_set_anchor(agxo_world* w, const double* pos3);
void agxo_world_set_cloth_gravity(agxo_world* w, double gz);
void agxo_world_step(agxo_world* w);
int agxo_world_frame(agxo_world* w, int kind, int index, double* pos3, double* quat4, double* lin3, double* ang3);
int agxo_world_contacts(agxo_world* w, double* out16, int max_out);
int agxo_world_closest(agxo_world* w, const int* ca, int na, const int* cb, int nb, double dist, double* out9, int max_out);
int agxo_world_cloth(agxo_world* w, double* x, double* contacts6, int max_contacts);
/* a water particle as a body (DrinkingEnv.get_water_rewards, drinking.py:52-91): pose / velocity, teleport, and the two proximity
 * queries (bit 0: within `dist` of the cup, bit 1: touched the person in the last internal substep) */
/* test hook: record the frames of the moving links and the free bodies at the start of every internal substep of the following calls
 * ([substep][NDOF + NFREE][12] floats: p, R row major); NULL stops */
void agxo_trace_into(float* buf);
int agxo_world_particle(agxo_world* w, int k, double* pos, double* vel);
void agxo_world_set_particle(agxo_world* w, int k, const double* pos);
int agxo_world_particle_query(agxo_world* w, int k, double dist);
void agxo_sleeve_reward(const double* pts6, const double* shoulder, const double* elbow, const double* wrist, double rad, double* out9);

#ifdef __cplusplus
}
#endif
#endif
