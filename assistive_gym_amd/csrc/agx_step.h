// agx_step.h -- the batched stepper: ONE WAVEFRONT PER ENVIRONMENT.  (Compiled once per variant: limits + task layer, see
// agx_kernels.hip; the citations below are for the FeedingJaco variant, bed_bathing.py has the same structure.)
//
// Replaces, for N lock-stepped environments, what the reference does per env.step():
//   AssistiveEnv.take_step         assistive_gym/envs/env.py:174-235   (action -> motor targets,
//                                   5x [p.stepSimulation, human limit clamp, update_targets])
//   FeedingEnv.step/_get_obs       assistive_gym/envs/feeding.py:12-48,85-112
//   FeedingEnv.get_food_rewards    assistive_gym/envs/feeding.py:50-83
//   AssistiveEnv.human_preferences assistive_gym/envs/env.py:237-274
// and the physics inside p.stepSimulation() for this scene.
//
// Layout: the environment's state record (AGX_H_STATE_WORDS floats, contiguous in HBM) is loaded
// with consecutive lanes reading consecutive words (coalesced), lives in LDS for all frame_skip
// substeps, and is written back once.  The model blob (tree, inertias, hull vertices, pair table)
// is shared by every environment and is read through L1/L2.  Generalised velocity deltas of the
// PGS solve live in registers (lane l owns DoF l and l+64); rows are visited in order, each row's
// dot product is a DPP wave reduction.
//
// Files: agx_ctx.h (limits, LDS / scratch layouts, per-lane context), agx_dyn.h (kinematics, ABA, M^-1),
// agx_collide.h (broadphase, GJK narrowphase, contact selection), agx_rows.h (constraint rows),
// agx_pgs.h (Gauss-Seidel sweeps, gfx950 assembly), agx_env.h (integration, task layer, kernel bodies),
// agx_reset.h (device-side reset generator: sampling + IK restarts, float64).
#pragma once
#include "agx_math.h"
#include "agx_gjk.h"
#include "../../include/agx_blob.h"
#include "agx_ctx.h"
#include "agx_dyn.h"
#include "agx_collide.h"
#include "agx_rows.h"
#include "agx_pgs.h"
#include "agx_pgs_lv.h"
#include "agx_pgs_lvs.h"
#include "agx_pgs_lvw.h"
#if AGX_TASK == 5   /* AGX_TASK_DRINKING (an enum: not visible to the preprocessor) */
#include "agx_water.h"
#endif
#include "agx_env.h"
#if AGX_HAS_SAMPLER
#include "agx_reset.h"
#endif
