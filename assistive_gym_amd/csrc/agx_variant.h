// agx_variant.h -- what one compiled variant of the kernels (agx_kernels.hip built with one set of limits and one task
// layer) exposes to the handle code of agx_api.hip: its limits, its LDS / scratch / debug layout and its launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct agx_variant {
  const char* name;
  int task_kind;                         // AGX_TASK_* the finish / observe kernels are compiled for
  int max_dof, max_free, max_block, max_human, max_coll, st_words, max_con, max_rows;
  int lds_bytes, lds_solve_bytes, scr_words, dbg_words;
  int dbg_con, dbg_minv, dbg_hdr, dbg_lam, dbg_time, dbg_qdd;     // debug record layout (agx_debug_layout)
  int rs_narm;                           // arm DoFs the reset generator's IK is compiled for, 0 = no reset generator in this variant
  // one-time kernel attribute set-up (dynamic LDS sizes)
  hipError_t (*init)(void);
  // launchers: grid = ne workgroups of one wavefront, environments [e0, e0 + ne)
  // phase: index of the substep within the env step (hooks after every SIM_SUBSTEPS-th, slot of the link-frame trace);
  // trace: [n_envs][trace_words] link frames per substep for the cloth kernel, null for models without a cloth
  void (*build)(hipStream_t st, int ne, const uint32_t* blob, float* state, const float* actions, float* scratch, float* debug, int e0, int n_envs, int sw,
                int act_dim, const uint8_t* active, int* overflow_total, float* trace, int trace_words, int phase);
  void (*solve)(hipStream_t st, int ne, const uint32_t* blob, float* state, float* scratch, float* debug, int e0, int n_envs, int sw, const uint8_t* active, int phase);
  // the build kernel with the persistent-manifold stage (blobs with AGX_P_MANIFOLD > 0); null in variants compiled without it
  void (*build_mf)(hipStream_t st, int ne, const uint32_t* blob, float* state, const float* actions, float* scratch, float* debug, int e0, int n_envs, int sw, int act_dim,
                   const uint8_t* active, int* overflow_total, float* trace, int trace_words, int phase);
  void (*finish)(hipStream_t st, int ne, const uint32_t* blob, float* state, const float* actions, float* scratch, float* obs, float* reward, uint8_t* done,
                 float* info, int e0, int n_envs, int sw, int act_dim, int obs_dim, const float* report, int report_words, float* cloth, int cloth_words);   // cloth: the water buffer for the drinking task layer (teleports drunk particles), unused elsewhere
  void (*observe)(hipStream_t st, int n_envs, const uint32_t* blob, float* state, float* obs, int sw, int obs_dim, const uint8_t* mask);   // mask: null = every environment
  void (*sample)(hipStream_t st, int n_envs, const uint32_t* blob, float* state, unsigned long long seed0, const unsigned long long* seeds, const uint8_t* mask,
                 int impairment_mode, int gender_mode, float* info4, int* episode, int sw, const int* first_restart, int* chosen,
                 const float* settled, int settled_sw, const float* fell);   // null without a reset generator; settled: [n_envs][settled_sw] records of the attached rag-doll model (or null); fell: [n_envs][sw] records of the attached fall model (arm manipulation, or null)
  // the garment (agx_cloth.h): nsub substeps replaying the trace; null in variants without a cloth
  void (*cloth)(hipStream_t st, int ne, const uint32_t* blob, const float* state, const float* trace, float* cloth, float* report, int e0, int n_envs, int sw,
                int trace_words, int cloth_words, int report_words, int nsub, const uint8_t* active, int lds_bytes);
  // collision verdict on freshly sampled states after a build pass (see agx_reset.h reset_collides)
  void (*verdict)(hipStream_t st, int n_envs, const uint32_t* blob, const float* scratch, const uint8_t* active, uint8_t* work, int* first_restart, const int* chosen);
  // collision flags (AGX_COLLIDE_*) of every environment's state after a build pass (agx_check_collisions)
  void (*collision_flags)(hipStream_t st, int n_envs, const uint32_t* blob, const float* scratch, uint8_t* flags);
  // dynamic LDS bytes of the cloth kernel for a garment of nn nodes (agxc::lds_words); null without a cloth kernel
  int (*cloth_lds_bytes)(int nn);
  // word of the per-environment scratch record that counts the entries of the warm-start memory (AGX_P_WARMSTART): agx_api.hip zeroes it
  // for every environment whose state is replaced from outside (set_state, resets)
  int scr_warm_word;
};

extern "C" const agx_variant* agx_variant_feeding(void);
extern "C" const agx_variant* agx_variant_bed_bathing(void);
extern "C" const agx_variant* agx_variant_scratch_itch(void);
extern "C" const agx_variant* agx_variant_bed_settle(void);
extern "C" const agx_variant* agx_variant_dressing(void);
extern "C" const agx_variant* agx_variant_arm_manipulation(void);
extern "C" const agx_variant* agx_variant_bed_bathing_l(void);
extern "C" const agx_variant* agx_variant_feeding_l(void);
extern "C" const agx_variant* agx_variant_feeding_m(void);
extern "C" const agx_variant* agx_variant_bed_bathing_m(void);
extern "C" const agx_variant* agx_variant_scratch_itch_m(void);
extern "C" const agx_variant* agx_variant_dressing_m(void);
extern "C" const agx_variant* agx_variant_dressing_l(void);
extern "C" const agx_variant* agx_variant_arm_manipulation_l(void);
extern "C" const agx_variant* agx_variant_drinking(void);
extern "C" const agx_variant* agx_variant_drinking_l(void);
extern "C" const agx_variant* agx_variant_drinking_m(void);
