/* agx_blob.h -- flat model blob + per-environment state record layout.
 *
 * The blob is the compiled form of the scene the reference builds at reset() through PyBullet
 * (assistive_gym/envs/feeding.py:114-182, envs/env.py:114-134, envs/agents/jaco.py:52-54,
 * envs/agents/tool.py:10-47, envs/agents/furniture.py:10-40, envs/human_creation.py:58-316):
 * kinematic tree, inertias, joint limits, motor gains, convex collision geometry, the static
 * collision-pair table, friction, the tool constraint and the task constants.  It is produced
 * by assistive_gym_amd/model (Python) and consumed, as DATA only, by libagx (HIP) and by the CPU
 * oracle.  Everything is 32-bit little-endian words: int32 or IEEE float32.
 *
 * This header defines a data format, not an algorithm.
 */
#ifndef AGX_BLOB_H
#define AGX_BLOB_H

#define AGX_BLOB_MAGIC 0x31584741 /* "AGX1" */
#define AGX_BLOB_VERSION 16
/* Agent.enforce_joint_limits (agent.py:240-250) resets a human joint found beyond a limit (q = limit, qd = 0).  A joint
 * stopped by its limit row arrives EXACTLY on the limit up to rounding, where `q < lower` is a coin flip of the arithmetic
 * (f32 here, f64 in the oracle / in Bullet); the reset is therefore applied only beyond this tolerance (radians). */
#define AGX_LIMIT_EPS 1e-6f
/* launch flag OR-ed into the `phase` argument of the solve kernel: this substep belongs to a reset-time settle loop (plain
 * p.stepSimulation() calls: feeding.py:178-179, bed_bathing.py:130-131, arm_manipulation.py:145-146, dressing.py:190-193), so none of
 * the hooks take_step runs between its stepSimulation calls (env.py:227-231) apply */
#define AGX_PHASE_SETTLE 0x40000000
#define AGX_BOX_CLIP 0.05f /* static world boxes are clipped to the other collider's AABB grown by this */
/* face manifold on static world boxes (table top, ground): besides the closest point, up to AGX_FACE_EXTRA more
 * vertices of the other collider become contact candidates -- those within AGX_FACE_BAND of its lowest vertex,
 * each at least AGX_FACE_SPREAD (horizontally) away from the points already chosen, farthest first */
#define AGX_FACE_EXTRA 3
#define AGX_FACE_BAND 0.0001f
#define AGX_FACE_SPREAD 0.01f

/* ---- header: int32[AGX_H_COUNT] at word 0 ------------------------------------------------ */
enum {
  AGX_H_MAGIC = 0, AGX_H_VERSION, AGX_H_NWORDS,
  AGX_H_NDOF,      /* 1-DoF moving links of all articulated bodies: robot (fixed links merged into
                      their parents) followed by the human joints that can be dynamic              */
  AGX_H_NFREE,     /* free rigid bodies (tool, bowl, food particles)                         */
  AGX_H_NHUMAN,    /* human collision bodies with a per-env world transform                  */
  AGX_H_NCOLL, AGX_H_NVERT, AGX_H_NGROUP,
  AGX_H_NFOOD, AGX_H_ACT_DIM, AGX_H_OBS_DIM,
  AGX_H_OFF_PARAMS, AGX_H_OFF_ROBOT, AGX_H_OFF_FREE, AGX_H_OFF_COLL, AGX_H_OFF_VERT,
  AGX_H_OFF_GROUP, AGX_H_OFF_TASK,
  AGX_H_STATE_WORDS,  /* words per environment state record                                  */
  AGX_H_S_Q, AGX_H_S_QD, AGX_H_S_QT, AGX_H_S_FREE, AGX_H_S_BASE, AGX_H_S_HUMAN, AGX_H_S_ENV,
  AGX_H_FOOD0,        /* index of the first food particle among the free bodies              */
  AGX_H_TOOL_BODY,    /* free-body index of the tool                                          */
  AGX_H_NDIR,         /* number of penetration-sampling directions (stored after the verts)  */
  AGX_H_OFF_DIRS,
  AGX_H_OFF_RESET,    /* reset section: sampling ranges + posed-human tree (AGX_X_*, AGX_XJ_*)                */
  AGX_H_NROBOT,       /* DoFs [0, NROBOT) belong to the robot, [NROBOT, NDOF) to the human                */
  AGX_H_NHDOF,        /* human DoFs; their link records exist per gender: record = dof + gender * NHDOF   */
  AGX_H_S_TREMOR,     /* state: float[NHDOF] tremor amplitude, float[NHDOF] tremor-free target (human.py:89-92) */
  AGX_H_TASK_KIND,    /* AGX_TASK_*: which task layer (observation, reward, force bookkeeping) the blob is compiled for */
  AGX_H_S_TASK,       /* state: task-specific words after the ENV block (bed bathing: bitmask of the targets not wiped yet) */
  AGX_H_TASK_WORDS,
  AGX_H_OFF_TARGETS,  /* bed bathing: float[2 genders][NT_MAX][4] = target position in its arm link frame + arm (0 upper, 1 fore) */
  AGX_H_OFF_MLP,      /* arm-limit classifier of Human.enforce_realistic_joint_limits (human.py:134-152): float W1[4][64], b1[64],
                         W2[64][64], b2[64], W3[64][64], b3[64], W4[64], b4[1] (assets/realistic_arm_limits_model.h5); 0 = none */
  AGX_H_OFF_CLOTH,    /* cloth section (AGX_CL_*), 0 = the scene has no cloth (dressing.py:153-154)                              */
  AGX_H_SIM_SUBSTEPS, /* internal substeps per p.stepSimulation(): numSubSteps of setPhysicsEngineParameter (dressing.py:184), else 1.
                         The stepper's substep is then DT / SIM_SUBSTEPS long, an env step has FRAME_SKIP * SIM_SUBSTEPS of them, and
                         what the reference does between two stepSimulation calls (limit reset, arm-limit classifier, env.py:226-232)
                         runs after every SIM_SUBSTEPS-th substep                                                                    */
  AGX_H_BASE_LINK,    /* 1 + the moving link that IS the base link of a robot with a floating base (Stretch: the last of six virtual
                         joints hanging off the anchor pose AGX_H_S_BASE): Robot.get_base_pos_orient() / convert_to_realworld
                         (agent.py:142-150, 165-171) read this link's pose.  0 = fixed base, the pose of the state record            */
  AGX_H_COUNT = 48
};

enum { AGX_TASK_FEEDING = 0,      /* assistive_gym/envs/feeding.py      */
       AGX_TASK_BED_BATHING = 1,  /* assistive_gym/envs/bed_bathing.py  */
       AGX_TASK_SCRATCH_ITCH = 2, /* assistive_gym/envs/scratch_itch.py */
       AGX_TASK_DRESSING = 3,     /* assistive_gym/envs/dressing.py     */
       AGX_TASK_ARM_MANIPULATION = 4,    /* assistive_gym/envs/arm_manipulation.py (single-arm robots) */
       AGX_TASK_DRINKING = 5 };          /* assistive_gym/envs/drinking.py: the `drinking` kernel variant; the water is a particle section in the
                                          * garment's format (AGX_CL_PARTICLES) stepped by agx_water.h.  Round 3: checked on the CPU wave emulator
                                          * only, NOT YET RUN ON A GPU; no env ids are registered for it                                */

/* ---- PARAMS: float[AGX_P_COUNT] ----------------------------------------------------------- */
enum {
  AGX_P_DT = 0,          /* physics time step, env.py:21 (0.02)                               */
  AGX_P_FRAME_SKIP,      /* substeps per env step, env.py:21 (5)                              */
  AGX_P_NITER,           /* PGS sweeps per substep (PyBullet default 50, feeding.py:155)      */
  AGX_P_ERP,             /* joint / constraint error reduction                                 */
  AGX_P_CONTACT_ERP,
  AGX_P_CONTACT_BREAK,   /* contact rows are built for separation < this                      */
  AGX_P_LIN_DAMP, AGX_P_ANG_DAMP,
  AGX_P_FRIC_EPS,        /* squared lateral speed above which friction follows the slip dir   */
  AGX_P_LIMIT_ACT,       /* joint-limit rows are built when the gap is below this             */
  AGX_P_ACTION_SCALE,    /* env.py:174 action_multiplier (0.05)                               */
  AGX_P_GRAVITY_Z,       /* env.py:104                                                        */
  AGX_P_GJK_TOL, AGX_P_GJK_MAXIT,
  AGX_P_MAX_CONTACTS, AGX_P_MAX_ROWS,
  AGX_P_ROBOT_GRAVITY_Z, /* per-body gravity of the robot (feeding.py:150-151 sets 0)             */
  AGX_P_HUMAN_GRAVITY_Z,
  AGX_P_CONTACT_SLACK,   /* solver rows only for contacts that could close within one substep:
                            dist + v_n*dt < slack (rows that stay inactive have no effect)         */
  AGX_P_MAX_ENTRIES,     /* cap on the summed (J,B) coefficient pairs of all rows of a substep    */
  AGX_P_NOOP_RETEST,     /* K > 0: rows of the non-friction block whose visit in a re-test sweep (every K-th) was a no-op are skipped until
                            the next re-test sweep; 0 = every row in every sweep (agx_pgs.h, oracle pgs())                 */
  /* [BULLET-UNVERIFIED] switches (all off = the default conventions; tests/diag/bullet_unknowns_sensitivity.py measures what each would change).
   * RESIDUAL_EPS is evaluated by the CPU oracle only; FRICTION_DIRS and WARMSTART by the oracle AND the device (round 4): */
  AGX_P_ORACLE_RESIDUAL_EPS = 21, /* > 0: the sweeps stop once max_rows (delta lambda x D)^2 <= eps (btSequentialImpulseConstraintSolver's
                                     m_leastSquaresResidualThreshold; PyBullet's default solverResidualThreshold is believed to be 1e-7)   */
  AGX_P_FRICTION_DIRS = 22,       /* 2: a second friction row per contact along n x t (SOLVER_USE_2_FRICTION_DIRECTIONS), each bounded by
                                     mu x the normal impulse (the friction pyramid); 0 / 1: one.  Row order: ... normals, the first directions of all
                                     contacts, the second directions of all contacts (Bullet interleaves the two per contact; with the blocked
                                     order the device sweeps one register set after the other).  Costs a third row per contact in the row and
                                     coefficient budgets (AGX_P_MAX_ROWS: 47 instead of 71 contacts next to FeedingJaco's 17 other rows)     */
  AGX_P_WARMSTART = 23,           /* > 0: contact normals start from this factor x the impulse the SAME contact -- (collider a, collider b,
                                     ordinal inside the pair) -- was solved to in the previous substep (SOLVER_USE_WARMSTARTING,
                                     m_warmstartingFactor 0.85).  Device: the solve kernel leaves (key, impulse) of its contacts in the
                                     environment's scratch record, the next build kernel seeds C_LAM from it, the solve starts its sweeps
                                     from those impulses (agx_env.h warm_*); the memory is per environment and is cleared whenever the
                                     environment's state is replaced (agx_set_state, agx_reset*, agx_sample_reset).                        */
  AGX_P_NOOP_PEN = 24,   /* > 0: the no-op re-test rule is switched OFF (plain sweeps) for an environment in every substep that has a contact
                            penetrating deeper than this (metres).  The rule delays the wake-up of a skipped row by up to K - 1 sweeps; with a tool
                            PRESSED onto skin -- where the force terms of the rewards come from -- that moved total_force_on_human by up to 5e-2 N
                            and single steps' rewards through flipped wiping events against the plain 50-sweep solve (f64 oracle with the rule vs
                            without, BedBathingSawyer wiping workload: 5 of 96 steps beyond 1e-3).  With 0.2 mm the same 96 steps are identical to
                            the plain solve, and FeedingJaco's resting food pile (penetrations of ~0.05 mm) keeps the rule: 75.5 -> 76.5 row
                            visits per sweep.  With the switch on (> 0) a contact of the robot or its tool with the person switches the rule off as well:
                            a robot link RESTING on the person sits at dist ~ 0 and carries newtons (wiping workload, f64: total_force_on_human
                            5.1404 N with the rule vs 5.1480 N plain, 1.5e-3 relative) -- every force the tasks report is a force on the person,
                            so all of them come from plain sweeps.  0 = the rule applies regardless (round 3) */
  AGX_P_MANIFOLD = 25,   /* > 0: persistent contact manifold for the hull pairs that are not resting on a static world box (those have the face
                            manifold): up to 4 points per collider pair live across substeps and steps (btPersistentManifold [BULLET-UNVERIFIED],
                            default off).  Per substep: (1) every cached point is refreshed -- world positions from its two body-local points,
                            distance along its stored world normal -- and dropped when that distance or its lateral drift exceeds the
                            contact-break distance (refreshContactPoints); (2) the substep's GJK contact of a pair replaces the cached point
                            whose local point on A is nearest within the break distance (getCacheEntry / replaceContactPoint), else it is
                            appended, else -- four cached -- it replaces the point whose removal keeps the largest area while the deepest point
                            stays (sortCachedPoints); (3) every cached point whose predicted gap is below the slack becomes a contact: the
                            pairs of the substep's own contacts in their order, each with its cached points in cache order, then the cached
                            pairs without a new point.  The memory (64 points per environment) lives with the warm-start memory in the
                            scratch record and is cleared with it.  Oracle and device; agx_env.h manifold_*                               */
  AGX_P_SPLIT_PEN = 26,  /* > 0: a contact penetrating deeper than this (metres) gets NO positional term in its row (b = -v_rel.n only).  Bullet's
                            split impulse ([BULLET-UNVERIFIED]; m_splitImpulse = true, m_splitImpulsePenetrationThreshold = -0.04): below the
                            threshold the penetration recovery leaves the velocity solve (m_rhs = velocity impulse, m_rhsPenetration = the rest) and
                            is applied to the POSITIONS of rigid bodies by a separate push solve; for multibodies -- what PyBullet makes of every
                            URDF and of createMultiBody -- btMultiBodyConstraintSolver sets up the same split and never solves the push part, i.e.
                            such a contact only stops the approach.  0 = off: the Baumgarte term on the full depth (deeply penetrating
                            START poses are pushed out within one substep); 0.04 = Bullet's value.  Oracle and device (agx_rows.h)            */
  AGX_P_SOLVE_WIDE = 27, /* != 0 (the default, 1): the feeding variant's solve kernel visits up to four rows with disjoint velocity slots at once, one per
                            16-lane group (csrc/agx_pgs_lvw.h); 0: one row per visit (csrc/agx_pgs_lvs.h).  The two compute the same bits (rows that share no
                            slot commute exactly): a device-only switch for same-process A/B runs and the bit-for-bit test; the oracle ignores it          */
  AGX_P_COUNT = 28
};

/* ---- ROBOT: one record per moving link, stride AGX_R_STRIDE ------------------------------- */
enum {
  AGX_R_PARENT = 0,      /* int: parent moving link, -1 = robot base, -2 = human base (chest)    */
  AGX_R_TPOS = 1,        /* float[3] parent-link frame -> joint frame (q = 0)                 */
  AGX_R_TQUAT = 4,       /* float[4] (x,y,z,w)                                                */
  AGX_R_AXIS = 8,        /* float[3] joint axis in the link frame                             */
  AGX_R_COM = 11,        /* float[3] centre of mass in the link frame                         */
  AGX_R_MASS = 14,
  AGX_R_INERTIA = 15,    /* float[6] xx,yy,zz,xy,xz,yz about the COM, link-frame axes         */
  AGX_R_LOWER = 21, AGX_R_UPPER = 22, AGX_R_HAS_LIMIT = 23, /* int                            */
  AGX_R_KP = 24, AGX_R_KD = 25, AGX_R_MAXF = 26,
  AGX_R_ACT = 27,        /* int: action component driving this joint, -1 = none               */
  AGX_R_QT0 = 28,        /* initial motor target (gripper joints)                             */
  AGX_R_JDAMP = 29,
  AGX_R_PB_INDEX = 30,   /* int: PyBullet joint index (jaco.py:8-17, human.py:5-58)              */
  AGX_R_KIND = 31,       /* int bits: bit0 human link (human gravity; hard limit reset after each substep, agent.py:240-250; limits scaled
                          * by the impairment), bit1 limits NOT scaled (legs, waist: human_creation.py:249-278), bit2 no hard limit
                          * reset (the rag-doll settle of bed_bathing.py:129-131 is plain stepSimulation) */
  AGX_R_JTYPE = 32,      /* int: 0 revolute, 1 prismatic (Sawyer gripper fingers, assets/sawyer/sawyer.urdf)  */
  AGX_R_ACT_MULT = 33,   /* float: Robot.action_multiplier of this joint's action (env.py:196-197; stretch.py:52); 0 = 1          */
  AGX_R_ACT_SRC = 34,    /* int: 1 + the DoF whose angle and limits the motor target is accumulated from -- Robot.action_duplication
                          * (env.py:218-220, stretch.py:51): the four telescoping joints of the Stretch's arm are all told to go where
                          * the first one's action leads; 0 = the joint itself                                                       */
  AGX_R_OBS_SKIP = 35,   /* int: 1 = an actuated joint whose angle the observation leaves out (the wheels of a mobile robot,
                          * feeding.py:90-92, and the duplicates of ACT_SRC, which are not controllable_joint_indices)               */
  AGX_R_STRIDE = 36
};

/* ---- FREE bodies: stride AGX_F_STRIDE ----------------------------------------------------- */
enum {
  AGX_F_MASS = 0,
  AGX_F_INERTIA = 1,     /* float[3] principal moments in the COM frame                       */
  AGX_F_GRAVITY = 4,     /* per-body gravity z (agent.py:196-197)                             */
  AGX_F_REFPOS = 5,      /* float[3] base(link) frame expressed in the COM frame              */
  AGX_F_REFQUAT = 8,     /* float[4]                                                          */
  AGX_F_KIND = 12,       /* int: AGX_KIND_*                                                   */
  AGX_F_RADIUS = 13,
  AGX_F_STRIDE = 16
};
enum { AGX_KIND_TOOL = 1, AGX_KIND_BOWL = 2, AGX_KIND_FOOD = 3 };

/* ---- COLLIDERS: convex core (vertex list) + radius, stride AGX_C_STRIDE ------------------- */
enum {
  AGX_C_BODY = 0,        /* int: body code, see below                                         */
  AGX_C_NVERT = 1, AGX_C_VOFF = 2, /* int: vertices are float[3] in the body frame            */
  AGX_C_RADIUS = 3,      /* sphere/capsule radius, or hull collision margin                   */
  AGX_C_FRICTION = 4,
  AGX_C_TAG = 5,         /* int: AGX_TAG_*                                                    */
  AGX_C_AABB_C = 6,      /* float[3] body-frame AABB centre of the core                       */
  AGX_C_AABB_H = 9,      /* float[3] half extents                                             */
  AGX_C_LINK = 12,       /* int: PyBullet link index of the collider on its body (-1 = base), what getContactPoints
                          * reports as linkIndexA/B (agent.py:100-116; bed_bathing.py:48,51)     */
  AGX_C_STRIDE = 16
};
/* body codes */
#define AGX_BODY_WORLD (-1)
#define AGX_PARENT_ROBOT_BASE (-1)
#define AGX_PARENT_HUMAN_BASE (-2)
#define AGX_BODY_ROBOT_BASE 100
#define AGX_BODY_FREE0 200
#define AGX_BODY_HUMAN0 300
/* flags of agx_check_collisions (agx.h) */
#ifndef AGX_COLLIDE_FLAGS
#define AGX_COLLIDE_FLAGS
enum { AGX_COLLIDE_ENV = 1, AGX_COLLIDE_SELF = 2 };
#endif
enum { AGX_TAG_ROBOT = 1, AGX_TAG_TOOL = 2, AGX_TAG_HUMAN = 3, AGX_TAG_FOOD = 4, AGX_TAG_BOWL = 5,
       AGX_TAG_TABLE = 6, AGX_TAG_PLANE = 7, AGX_TAG_WHEELCHAIR = 8, AGX_TAG_BED = 9 };

/* ---- pair GROUPS (static broadphase table): stride AGX_G_STRIDE --------------------------- */
enum {
  AGX_G_A0 = 0, AGX_G_A1 = 1,   /* collider range A [a0,a1)                                   */
  AGX_G_B0 = 2, AGX_G_B1 = 3,   /* collider range B (male human / default)                    */
  AGX_G_B0F = 4, AGX_G_B1F = 5, /* collider range B for a female human, -1 = same as default  */
  AGX_G_FLAGS = 6,              /* bit0: A and B are the same range (i<j only); bit1: the task asks
                                   whether a manifold point exists (broadphase margin = CONTACT_BREAK);
                                   bit2: skip pairs on the same link or on parent/child links
                                   (URDF_USE_SELF_COLLISION semantics, jaco.py:53);
                                   bit3 / bit4: the group exists for a male / female human only (both collider ranges
                                   are gender specific, e.g. the human's arm against the rest of its body);
                                   bit5: the group is skipped while every human DoF is frozen (both sides static);
                                   bit6: a solver row for EVERY contact of the group inside CONTACT_BREAK, not only those whose
                                   predicted gap is below CONTACT_SLACK (the prediction knows nothing of the motor rows: a tool
                                   driven onto the person closes gaps it calls open).  Set on the groups whose forces the tasks
                                   report -- robot / tool against the person; not a tool compiled as a compound of dozens of convex
                                   pieces (spoon, cup), whose speculative contacts would starve the substep's contact budget -- together
                                   with KEEP = 0: against the oracle without
                                   any budget these two were worth up to 0.5 relative on total_force_on_human in 1-2 % of the
                                   contact-rich steps (round 5, profiles/r05/approximation_budget.json)                     */
  AGX_G_KEEP = 7,               /* per A collider keep only the KEEP contacts with the smallest
                                   predicted gap (0 = keep all)                               */
  AGX_G_STRIDE = 8
};

/* ---- TASK constants: float[AGX_T_COUNT] --------------------------------------------------- */
enum {
  AGX_T_W_DISTANCE = 0, AGX_T_W_ACTION, AGX_T_W_FOOD,           /* config.ini:15-18           */
  AGX_T_C_V, AGX_T_C_F, AGX_T_C_HF, AGX_T_C_FD, AGX_T_C_FDV,   /* config.ini:40-44           */
  AGX_T_SUCCESS_FRAC,                                           /* config.ini:19              */
  AGX_T_MOUTH_DIST, AGX_T_SPILL_DIST,                           /* feeding.py:61,71           */
  AGX_T_SI_LIMB_DIMS = 9, /* scratch itch blobs reuse the feeding-only words 9..16: float[2 genders][2 limbs][length, radius] of the upper arm and
                          * the forearm (scratch_itch.py:136-139), read by the device-side reset generator                                          */
  AGX_T_MOUTH_M = 11,    /* float[3] mouth offset in the head frame, male (feeding.py:186)    */
  AGX_T_MOUTH_F = 14,    /* float[3] female                                                   */
  AGX_T_HEAD_LINK = 17,  /* int: moving link that is the human head (human.head = 23)         */
  AGX_T_EE_LINK = 18,    /* int: moving link carrying the end-effector frame                  */
  AGX_T_EE_POS = 19,     /* float[3] end-effector (PyBullet link 8) frame in that link frame  */
  AGX_T_EE_QUAT = 22,    /* float[4]                                                          */
  AGX_T_TOOL_POS = 26,   /* float[3] tool pivot in the end-effector frame (jaco.py:26)        */
  AGX_T_TOOL_QUAT = 29,  /* float[4] tool frame in the end-effector frame (jaco.py:31)        */
  AGX_T_TOOL_MAXF = 33,  /* tool.py:47                                                        */
  AGX_T_EPISODE_LEN = 34,/* feeding.py:37                                                     */
  AGX_T_COOP = 35,       /* int: 1 = the human is controllable (<Task><Robot>HumanEnv, feeding_envs.py:44-69):
                          * ACT_DIM / OBS_DIM of the header include the human's action and observation    */
  AGX_T_TOOL_OBS_POS = 36, /* float[3] frame whose pose the observation reports, in the tool base frame: identity for the
                            * spoon (feeding.py:86), link 1 of the wiper (bed_bathing.py:81)                        */
  AGX_T_TOOL_OBS_QUAT = 39,/* float[4]                                                                              */
  /* ---- bed bathing (assistive_gym/envs/bed_bathing.py, config.ini:9-13) ---- */
  AGX_T_W_WIPE = 43,       /* wiping_reward_weight                                                                  */
  AGX_T_TARGET_RADIUS = 44,/* a target is wiped when a (tool link 1, human) contact lies within this (bed_bathing.py:57) */
  AGX_T_CLOSEST_DIST = 45, /* range of the tool <-> human closest-point query (bed_bathing.py:23)                   */
  AGX_T_PAD_LINK = 46,     /* int: bitmask over (tool link + 1) of the tool links whose contacts count: the wiping pad, link 1
                            * (bed_bathing.py:51 `linkA in [1]`); the scratcher's links 0 and 1 (scratch_itch.py:54 `linkA in [0, 1]`) */
  AGX_T_ARM_LINK = 47,     /* int[2]: moving links carrying the upper-arm / forearm targets (human.right_shoulder, right_elbow) */
  AGX_T_OBS_LINK = 49,     /* int[3]: moving links whose positions the observation reports (shoulder, elbow, wrist) */
  AGX_T_NT = 52,           /* int[2 genders][2 arms]: number of targets (bed_bathing.py:173-188)                    */
  AGX_T_NT_MAX = 56,       /* int: row count of the per-gender target table                                         */
  /* ---- pose-dependent arm limits (human.py:134-152), active in co-op envs whose controllable joints contain a shoulder ---- */
  AGX_T_ARM_LIMIT_ON = 57, /* int: 1 = run the classifier after every substep (env.py:230-231)                      */
  AGX_T_ARM_LIMIT_DOF = 58,/* int[4]: DoFs of shoulder x, y, z and elbow of the movable arm (human.py:139)          */
  AGX_T_ARM_LIMIT_SIGN = 62,/* float: -1 right arm, +1 left arm (human.py:142-145)                                   */
  /* ---- dressing (assistive_gym/envs/dressing.py, config.ini:28-31; W_WIPE = dressing_reward_weight, SUCCESS_FRAC = task_success_threshold) ---- */
  AGX_T_C_D = 64,           /* dressing_force_weight (config.ini:45)                                                 */
  AGX_T_ARM_RADIUS = 65,    /* float[2 genders]: hand_radius = elbow_radius = shoulder_radius (human_creation.py:89,140; util.py:134-138) */
  /* ---- arm manipulation (assistive_gym/envs/arm_manipulation.py, config.ini:33-37; W_DISTANCE = distance_human_weight, W_WIPE =
   * distance_end_effector_weight, SUCCESS_FRAC = task_success_threshold) ---- */
  AGX_T_C_P = 67,           /* high_pressures_weight (config.ini:46)                                                  */
  AGX_T_STOMACH_BODY = 68,  /* int: static human collision body whose frame is human.stomach (link 24)                */
  AGX_T_WAIST_BODY = 69,    /* int: ... human.waist (link 27)                                                         */
  AGX_T_DUP_ACT = 70,       /* int: a single-arm robot driven as 'both' arms (arm_manipulation_envs.py:13, robot.py:16) lists its arm joints
                             * twice: ACT_DIM counts both copies, the second copy's targets win, the observation reports the angles twice */
  AGX_T_PRESSURE_DIST = 71, /* float: range of tool.get_closest_points(human) that counts contact points for the pressure term (env.py:262) */
  /* a second tool in the robot's other hand (two-armed robots in arm manipulation: tool_left, arm_manipulation.py:15-16; the first
   * tool -- AGX_H_TOOL_BODY, AGX_T_EE_LINK ... -- is tool_right) */
  AGX_T_TOOL2_BODY = 72,    /* int: free-body index of the second tool, 0 = there is none                                            */
  AGX_T_EE2_LINK = 73,      /* int: moving link carrying the second end-effector frame (robot.left_end_effector)                     */
  AGX_T_EE2_POS = 74,       /* float[3] that frame in the link frame                                                                 */
  AGX_T_EE2_QUAT = 77,      /* float[4]                                                                                              */
  AGX_T_TOOL2_POS = 81,     /* float[3] pivot of the second tool in its end-effector frame                                           */
  AGX_T_TOOL2_QUAT = 84,    /* float[4]                                                                                              */
  AGX_T_COUNT = 88
};

/* ---- RESET section (offset AGX_H_OFF_RESET): what FeedingEnv.reset samples (feeding.py:114-172, human.py:72-102,
 * env.py:120, furniture.py:33, robot.py:84-121) and the kinematic tree of the posed, static human
 * (human_creation.py:188-278).  Consumed by the device-side reset generator (csrc/agx_reset.h) and by its
 * host mirror (assistive_gym_amd/host/reset.py). ------------------------------------------------------- */
enum {
  AGX_X_NJOINT = 0,      /* int: joints of the human tree (42)                                  */
  AGX_X_NARM = 1,        /* int: robot arm DoFs solved by the IK (= robot ACT slots)            */
  AGX_X_BASE_POS = 2,    /* float[3] robot base, world (jaco.py:47)                             */
  AGX_X_BASE_QUAT = 5,   /* float[4]                                                            */
  AGX_X_EE_QUAT = 9,     /* float[4] end-effector orientation the IK aims for (jaco.py:43)      */
  AGX_X_EE_TARGET = 13,  /* float[3] centre of the end-effector start position (feeding.py:139) */
  AGX_X_EE_RANGE = 16,   /* +- uniform offset per axis                                          */
  AGX_X_BOWL_POS = 17,   /* float[3] bowl base position (furniture.py:33)                       */
  AGX_X_BOWL_RANGE = 20, /* +- uniform offset in x and y                                        */
  AGX_X_HBASE_M = 21,    /* float[3] human base position, male (human.py:102)                   */
  AGX_X_HBASE_F = 24,    /* float[3] female                                                     */
  AGX_X_FOOD_R = 27,     /* particle radius of the 2x2x2 grid above the spoon (feeding.py:158-166) */
  AGX_X_HEAD_RANGE = 28, /* head joint angles ~ U(-r, r) radians (feeding.py:125)               */
  AGX_X_IK_ITERS = 29,   /* int: damped-least-squares iterations per restart                    */
  AGX_X_IK_DAMP = 30, AGX_X_IK_MAXSTEP = 31,
  AGX_X_IK_THRESH = 32,  /* position and orientation acceptance threshold (robot.py:84)         */
  AGX_X_IK_RESTARTS = 33,/* int: max restarts (env.py:289 max_ik_random_restarts)               */
  AGX_X_IK_TOL = 34,     /* early exit of the iteration                                         */
  AGX_X_IK_RANDLIM_FROM = 35, /* int: restarts >= this randomise the IK limits (robot.py:91)    */
  AGX_X_FRIC_LO = 36, AGX_X_FRIC_HI = 37,   /* plane lateral friction range (env.py:120)        */
  AGX_X_LIMIT_LO = 38,   /* impairment 'limits': scale ~ U(lo, 1) (human.py:85)                 */
  AGX_X_TREMOR_RANGE = 39,/* impairment 'tremor': amplitude ~ U(-r, r) radians (human.py:89-90) */
  AGX_X_BOWL_BODY = 40,  /* int: free-body index of the bowl                                    */
  AGX_X_OFF_JOINTS = 41, /* int: joint table [2 genders][NJOINT][AGX_XJ_STRIDE], section-relative */
  AGX_X_OFF_BODIES = 42, /* int: int[NHUMAN] link of every static human collision body, -1 = base  */
  AGX_X_OFF_DYN = 43,    /* int: int[NHDOF] human joint behind every human DoF                  */
  AGX_X_STRENGTH_LO = 44,/* impairment 'weakness': strength ~ U(lo, 1) (human.py:86; unused by Feeding) */
  AGX_X_FOOD_OFF = 45,   /* float[3] offset of the food grid from the tool position (feeding.py:162) */
  AGX_X_COLLISION_TRIES = 48, /* int: how many successful IK restarts may be rejected because the robot / tool touches the human,
                          * the table or the wheelchair (robot.py:105-112, env.py:299-308) before one is accepted unchecked */
  AGX_X_REACTIVE_KP = 49,   /* gain of the reactive hold of a human that is not an agent (setup_joints reactive_gain, scratch_itch.py:105); 0 = none (Feeding) */
  AGX_X_REACTIVE_MAXF = 50, /* its force limit before the strength factor (reactive_force, human.py:126)                                                    */
  AGX_X_FLAGS = 51,         /* int: bit 0 = the human's controllable joints stay dynamic whatever the impairment (human.py:108 with a reactive force);
                             * bit 1 = scratch itch: draw the limb and the target on it (scratch_itch.py:134-146; dimensions in AGX_T_SI_LIMB_DIMS);
                             * bit 2 = dressing: settle gravity and garment offset into the task words (AGX_DR_CLOTH_GRAVITY, AGX_DR_CLOTH_OFF);
                             * bit 3 = a robot on wheels (env.py:282-293): no IK -- the base at BASE_POS + U(-r, r)^2 (r = TOC_POS_RANGE), yaw TOC_YAW0
                             *         + U(-r, r) (r = TOC_YAW_RANGE; roll and pitch of BASE_QUAT's rpy are zero), joint MOBILE_LIFT_DOF at MOBILE_LIFT +
                             *         U(-0.1, 0.1) (stretch.py:58-62), every other joint at its QT0; a colliding placement is drawn again (env.py:299-308);
                             * bit 4 = bed bathing: the human's pose (base + joint angles) is where the rag doll of a second model came to rest
                             *         (bed_bathing.py:119-137): agx_sample_reset samples that model's drop record, settles it and hands its state
                             *         record to this sampler (agx_attach_settle_model);
                             * bit 5 = THIS blob is that rag-doll model: its sampler writes the drop record -- base at HBASE_M / HBASE_F with the virtual
                             *         joint angles (yaw, pitch, roll) EE_TARGET, every joint U(-EE_RANGE, EE_RANGE) clamped to its limits;
                             * bit 6 = bed bathing: all wiping targets alive, TOTAL_FOOD = their number (bed_bathing.py:173-188);
                             * bit 7 = arm manipulation (arm_manipulation.py:110-180; with bit 4): between the rag doll's settle and the base pose search the
                             *         human's right arm is posed (the joints with AGX_XJ_FLAGS bit 2 at their PRESET) and FALLS for 100 simulation steps
                             *         at gravity -1 while everything else of the human is static (:139-146) -- in a THIRD handle on the same model
                             *         ("fall model": this blob with HUMAN_GRAVITY_Z = -1 and bit 8 set, ModelBlob.fall_model()), attached to this one;
                             *         the rag-doll model is attached to the fall model.  This sampler reads the human's bodies, the arm's joint angles
                             *         and velocities from the fall model's settled record and the four goals of the base pose search (wrist, waist,
                             *         elbow, stomach: TOC_GOAL_LINKS[0..2], TOC_GOAL_LINK3) from the tree with those angles;
                             * bit 9 = a two-armed robot (arm_manipulation.py:165; Robot.position_robot_toc with arm = ['right', 'left']): ONE base pose for both
                             *         arms -- each arm solves its own start pose (EE_TARGET / EE_TARGET2, orientation EE_QUAT) and two of the four goals
                             *         (right: TOC_GOAL_LINKS[0..1] = wrist, waist; left: TOC_GOAL_LINKS[2], TOC_GOAL_LINK3 = elbow, stomach); both start
                             *         poses must be reached, goals and manipulability add up over the arms; the second tool goes into the second hand;
                             * bit 8 = THIS blob is the fall model: its sampler writes the record the arm falls from -- the human where the rag doll lies
                             *         with the preset arm, the robot parked at FALL_PARK with its arm at the middle of its joint ranges               */
  /* base pose search of a free-standing robot (Robot.position_robot_toc, robot.py:123-215); TOC_ATTEMPTS = 0: the base is fixed (BASE_POS / BASE_QUAT) */
  AGX_X_TOC_ATTEMPTS = 52,  /* int: candidate base poses per round (<= 64: one per lane)                                                   */
  AGX_X_TOC_ROUNDS = 53,    /* int: rounds of new candidates while no candidate reaches the start pose                                   */
  AGX_X_TOC_POS_RANGE = 54, /* candidate position = BASE_POS + (x, y, 0), x ~ U(0, r) x TOC_X_SIGN, y ~ U(-r, r) (random_position)          */
  AGX_X_TOC_YAW_RANGE = 55, /* candidate yaw = TOC_YAW0 + U(-r, r) (random_rotation)                                                      */
  AGX_X_TOC_YAW0 = 56, AGX_X_TOC_X_SIGN = 57,
  AGX_X_TOC_IK_ITERS = 58,  /* int: damped-least-squares iterations per goal (max_ik_iterations)                                          */
  AGX_X_TOC_THRESH = 59,    /* a goal counts as reached below this position (start pose: and orientation) error (robot.py:97)             */
  AGX_X_TOC_GOAL_LINKS = 60,/* int[3]: human tree links whose origins (+ TOC_GOAL_OFF) are the goals besides the start pose (scratch_itch.py:107-109, dressing.py:126-132) */
  AGX_X_TOC_GOAL_ORIENT = 63, /* int: 1 = the goals carry the end-effector orientations TOC_GOAL_QUAT (dressing.py:132), 0 = position only        */
  AGX_X_TOC_GOAL_OFF = 64,  /* float[3] offset added to every goal position                                                                */
  AGX_X_CLOTH_GRAVITY_SETTLE = 67, /* dressing (FLAGS bit 2): gravity on the garment during the settle of reset() (dressing.py:178) ...          */
  AGX_X_CLOTH_GRAVITY = 68,        /* ... and afterwards (dressing.py:195): agx_reset writes it when its settle is over                       */
  AGX_X_CLOTH_ORIG_POS = 69,/* float[3]: the garment is loaded shifted by (end effector position - this) (dressing.py:146-149)              */
  AGX_X_TOC_GOAL_QUAT = 72, /* float[3][4]                                                                                                  */
  AGX_X_CHAIN = 84,         /* int[7]: the DoFs of the arm's joints in chain order (each one's parent is the one before; the Sawyer's skip its head pan) */
  AGX_X_TOC_NGOALS = 91,    /* int: goals besides the start pose (1 ... 3)                                                                 */
  AGX_X_TOC_GOAL_KIND = 92, /* int: 0 = origins of TOC_GOAL_LINKS (+ TOC_GOAL_OFF); 1 = the mouth target (feeding.py:142)                  */
  AGX_X_PED_N = 93,         /* int: boxes of the robot's own pedestal (0 ... 2), base frame, already grown by a link radius: a candidate whose
                               start pose puts a joint origin past the shoulder, a link midpoint or the end effector inside one is rejected */
  AGX_X_MOBILE_LIFT = 94,   /* float: centre of the lift joint's start height (FLAGS bit 3)                                                 */
  AGX_X_MOBILE_LIFT_DOF = 95, /* int: its DoF                                                                                              */
  AGX_X_PED_BOX = 96,       /* float[PED_N][6]: min corner, max corner                                                                     */
  AGX_X_TOC_GOAL_LINK3 = 108, /* int: a fourth goal link (arm manipulation: wrist, waist, elbow, stomach, arm_manipulation.py:162)                 */
  AGX_X_FALL_PARK = 109,    /* float[3]: where the robot stands while the arm falls (FLAGS bit 8; it is placed afterwards, :162)                    */
  AGX_X_CHAIN2 = 112,       /* int[7]: FLAGS bit 9 -- the DoFs of the SECOND arm's joints in chain order (arm manipulation with a two-armed robot: CHAIN is
                               the right arm holding tool_right, CHAIN2 the left arm holding tool_left: AGX_T_EE2_*, AGX_T_TOOL2_*)                   */
  AGX_X_EE_TARGET2 = 119,   /* float[3]: centre of the second arm's end-effector start position (arm_manipulation.py:159), +- EE_RANGE                */
  AGX_X_COUNT = 124
};
enum {
  AGX_XJ_PARENT = 0,     /* int: parent joint (PyBullet link numbering), -1 = base              */
  AGX_XJ_OFF = 1,        /* float[3] joint frame in the parent link frame                       */
  AGX_XJ_AXIS = 4,       /* float[3]                                                            */
  AGX_XJ_LOWER = 7, AGX_XJ_UPPER = 8,  /* limits at limit_scale 1                                */
  AGX_XJ_FLAGS = 9,      /* int: bit0 revolute (else fixed), bit1 limits scale with the impairment, bit2 the fall model poses this joint at its
                            PRESET instead of where the rag doll left it (arm_manipulation.py:139) */
  AGX_XJ_PRESET = 10,    /* joint angle set by the task at reset (feeding.py:124), radians      */
  AGX_XJ_DRAW = 11,      /* int: index of the head-angle draw added to the preset, -1 = none    */
  AGX_XJ_STRIDE = 12
};

/* ---- per-env ENV block inside the state record (offset AGX_H_S_ENV) ----------------------- */
enum {
  AGX_E_PLANE_FRICTION = 0, /* env.py:120                                                     */
  AGX_E_GENDER = 1,         /* int 0 male, 1 female (human.py:76-78)                          */
  AGX_E_TARGET = 2,         /* float[3] mouth target, world (feeding.py:192-196)              */
  AGX_E_FOOD_ALIVE = 5,     /* int bitmask: particles still in self.foods (feeding.py:50-83)  */
  AGX_E_FOOD_ACTIVE = 6,    /* int bitmask: particles still in self.foods_active              */
  AGX_E_ITERATION = 7,      /* int, env.py:185                                                */
  AGX_E_TASK_SUCCESS = 8,   /* int, feeding.py:64                                             */
  AGX_E_RNG = 9,            /* uint32[2] per-env counter RNG for the teleport draw            */
  AGX_E_TOTAL_FOOD = 11,    /* int                                                            */
  AGX_E_FROZEN = 12,        /* int bitmask of DoFs made static (mass 0 links, human.py:104-110) */
  AGX_E_LIMIT_SCALE = 13,   /* scale of the human joint limits (impairment 'limits', human.py:85,
                             * human_creation.py:199-200)                                          */
  AGX_E_HUMAN_KP = 14,      /* > 0: position gain of the human's joint motors in this environment, overriding the blob's:
                             * the reactive hold of a non-controllable human, setup_joints(reactive_gain) (human.py:124-127,
                             * scratch_itch.py:106), vs motor_gains when the human is an agent (env.py:175-182)          */
  AGX_E_HUMAN_MAXF = 15,    /* with it: the motors' force limit, reactive_force * strength (human.py:126)               */
  AGX_E_COUNT = 16
};
/* bed bathing reuses AGX_E_TASK_SUCCESS (targets wiped, bed_bathing.py:60) and AGX_E_TOTAL_FOOD (total_target_count,
 * bed_bathing.py:187); AGX_E_TARGET / FOOD_* / RNG are unused.  Its task words (offset AGX_H_S_TASK): */
enum { AGX_BB_ALIVE = 0,        /* int[AGX_BB_ALIVE_WORDS] bitmask of the targets not wiped yet, upper-arm targets first */
       AGX_BB_ALIVE_WORDS = 6,
       AGX_BB_PREV = 6,         /* float[4] arm_previous_valid_pose (human.py:147-149): shoulder x, y, z, elbow          */
       AGX_BB_HAS_PREV = 10,    /* int: a valid pose has been seen (arm_previous_valid_pose is not None)                 */
       AGX_BB_WORDS = 12 };
/* scratch itch (offset AGX_H_S_TASK): the target point on the arm (scratch_itch.py:134-146), where the tool last scratched
 * (prev_target_contact_pos, :30,96) and, at the same offsets as bed bathing, the arm-limit classifier's remembered pose */
enum { AGX_SI_TARGET = 0,       /* float[3] target_on_arm, in the frame of its limb                                        */
       AGX_SI_LIMB = 3,         /* int: 0 upper arm (human.right_shoulder), 1 forearm (human.right_elbow)                  */
       AGX_SI_PREV_CONTACT = 12,/* float[3]                                                                                */
       AGX_SI_WORDS = 16 };
/* arm manipulation (offset AGX_H_S_TASK): float task_success = best reward_distance_human so far, 0 = none yet (arm_manipulation.py:46-47);
 * the arm-limit words at AGX_BB_PREV / AGX_BB_HAS_PREV */
enum { AGX_AM_BEST = 0, AGX_AM_WORDS = 12 };
/* drinking (offset AGX_H_S_TASK): the water particles still in the scene (self.waters, drinking.py:52-91) and those that have not hit the
 * person yet (self.waters_active), 64-bit masks; the task constants reuse words no drinking scene needs otherwise: AGX_T_W_WIPE = cup_tilt_weight
 * (config.ini:24), AGX_T_TARGET_RADIUS = the radius of the cup's cylinder test (0.05, drinking.py:64), AGX_T_TOOL_OBS_POS / _QUAT = the frame
 * the reward reads the cup in ([0, 0.06, 0], rpy (pi/2, 0, 0): drinking.py:24,56), AGX_T_EE2_POS / AGX_T_TOOL2_POS = cup_top_center_offset /
 * cup_bottom_center_offset in that frame (drinking.py:138-139) */
enum { AGX_DK_ALIVE = 0, AGX_DK_ACTIVE = 2, AGX_DK_WORDS = 12 };   /* (words 6 .. 10 stay what they are in every task: AGX_BB_PREV / AGX_BB_HAS_PREV) */
#define AGX_T_W_TILT AGX_T_W_WIPE
#define AGX_T_DK_TOP AGX_T_EE2_POS
#define AGX_T_DK_BOTTOM AGX_T_TOOL2_POS
/* dressing (offset AGX_H_S_TASK): at AGX_BB_PREV / AGX_BB_HAS_PREV the arm-limit classifier's remembered pose, like the others */
enum { AGX_DR_CLOTH_GRAVITY = 0, /* float: world gravity z acting on the cloth: -9.81 / 2 while it settles in reset, then -9.81 (dressing.py:178,195) */
       AGX_DR_FORCE_SUM = 1,     /* float: cloth_force_sum of the last step (dressing.py:96), an observation input                                */
       AGX_DR_BEST = 2,          /* float: self.task_success, the best reward_dressing so far (dressing.py:62-63)                                 */
       AGX_DR_CLOTH_OFF = 3,     /* float[3]: offset of the garment's load position, written by the device-side reset generator for the launch that
                                    places the garment (x = X0 + offset); unused afterwards                                                      */
       AGX_DR_WORDS = 12 };

/* ---- CLOTH section (offset AGX_H_OFF_CLOTH): the garment of DressingEnv.reset (dressing.py:153-154: p.loadCloth of
 * assets/clothing/hospitalgown_reduced.obj, scale 1.4, mass 0.16, collisionMargin 0.04, four anchor nodes; p.clothParams).
 * The fork's cloth API is Bullet's btSoftBody [BULLET-UNVERIFIED]; what is restated is its position-based solver:
 * nodes, links (one per mesh edge), node normals / areas for the aerodynamic drag, anchors, node-vs-rigid contacts.
 * int32 header [AGX_CL_HDR], then the arrays at the section-relative word offsets it names. ------------------------ */
enum {
  AGX_CL_NN = 0,         /* nodes: the OBJ's vertices in order of first appearance in its face list (tinyobj re-indexing)   */
  AGX_CL_NL = 1,         /* link SLOTS = unique mesh edges sorted by colour, plus empty slots (-1) of the bank schedule          */
  AGX_CL_NCOLOR = 2,     /* colour classes: links of one class share no node and are relaxed in parallel; classes in order.  The first
                            16 x NPATCH_COLOR classes are patch classes (below), the others hold the links between patches       */
  AGX_CL_NANCHOR = 3,
  AGX_CL_NSHAPE = 4,     /* rigid colliders the cloth is tested against                                                      */
  AGX_CL_OFF_COLOR = 5,  /* int[NCOLOR + 1] first link of every class                                                        */
  AGX_CL_OFF_LINK = 6,   /* {int a | b << 16 (-1: empty slot), float rest length squared}[NL]                                 */
  AGX_CL_OFF_NODE = 7,   /* {int first incident face entry, float area}[NN] (+ one terminating entry): node area = mean rest
                            area of the incident faces (btSoftBody::updateArea)                                             */
  AGX_CL_OFF_FACE = 8,   /* int[entries] j | k << 16: the other two vertices of each incident face, in the face's winding      */
  AGX_CL_OFF_X0 = 9,     /* float[NN][3] node positions of the loaded garment for cloth_offset = 0 (dressing.py:149-153)       */
  AGX_CL_OFF_ANCHOR = 10,/* {int node, float[3] offset from the attachment body}[NANCHOR]                                     */
  AGX_CL_OFF_SHAPE = 11, /* int[NSHAPE][4]: collider index, first plane, plane count (0: capsule / sphere core + radius), gender
                            (0 always present, 1 / 2: part of the male / female human only)                                    */
  AGX_CL_OFF_PLANE = 12, /* float[planes][4]: outward unit normal and offset of the hull's faces in the body frame; a hull's list is
                            padded to a multiple of four planes with copies of its last one (the cloth kernel reads four at a time)    */
  AGX_CL_TRI = 13,       /* int[6]: the two vertex triples around the opening of the left sleeve (dressing.py:156-157)         */
  AGX_CL_OFF_PARAM = 19, /* float[AGX_CP_COUNT]                                                                                */
  AGX_CL_MAX_LINKS_PER_COLOR = 20, /* int: size of the largest class between patches (<= 1,024: one link per thread of the cloth kernel)   */
  AGX_CL_OFF_PERM = 21,  /* int[4096]: node owned by slot t of the cloth kernel (thread t % threads, t / threads-th node of that thread), -1 = none:
                            the nodes in Morton order of their rest positions, so that the 64 nodes of a wave lie close together    */
  AGX_CL_NPATCH_COLOR = 22, /* K: the link table starts with 16 x K classes of exactly 64 slots, class w K + c = colour c of the links whose
                            two nodes both belong to patch w (the 256 nodes that wave w of the cloth kernel owns, OFF_PERM): patches
                            share no node, so a wave relaxes its K classes in order without waiting for any other wave            */
  AGX_CL_PARTICLES = 23,    /* 1 = the nodes are free particles (the water of the drinking task: no links, faces or anchors): spheres of radius MARGIN
                             * that also collide with each other (a Jacobi pass per solver iteration, neighbours in ascending order), under the world's
                             * gravity AGX_P_GRAVITY_Z                                                                                                  */
  AGX_CL_HDR = 24
};
enum {
  AGX_CP_KLST = 0,       /* material linear stiffness (clothParams kLST)                                                      */
  AGX_CP_KDP = 1,        /* damping                                                                                           */
  AGX_CP_KDG = 2,        /* drag                                                                                              */
  AGX_CP_KDF = 3,        /* dynamic friction                                                                                  */
  AGX_CP_KCHR = 4, AGX_CP_KKHR = 5, AGX_CP_KAHR = 6,   /* rigid / kinetic contact hardness, anchor hardness                   */
  AGX_CP_PITER = 7,      /* position solver iterations                                                                        */
  AGX_CP_MARGIN = 8,     /* collisionMargin                                                                                   */
  AGX_CP_NODE_IM = 9,    /* inverse mass of a node (total mass / NN each)                                                     */
  AGX_CP_AIR_DENSITY = 10,/* btSoftBodyWorldInfo::air_density (1.2)                                                           */
  AGX_CP_FORCE_SCALE = 11,/* dressing.py:35: forces * 10                                                                      */
  AGX_CP_FORCE_MAX = 12, /* dressing.py:42: only forces below 20 count                                                        */
  AGX_CP_EE_BELOW = 13,  /* dressing.py:42: only contacts more than 0.05 below the end effector count                         */
  AGX_CP_COUNT = 16
};
#define AGX_CLOTH_MAX_COLORS 16
#ifndef AGX_CLOTH_THREADS      /* -DAGX_CLOTH_THREADS=512 with a blob compiled for it (AGX_CLOTH_THREADS=512 python -m ...compiler): same-box A/B builds */
#define AGX_CLOTH_THREADS 1024
#endif
#define AGX_CLOTH_NODE_CONTACTS 2   /* node-vs-rigid contacts kept per node and substep: the first ones in shape order */
/* per-environment cloth report written by the cloth kernel for the finish kernel: float[18] the six sleeve vertices, 2 unused,
 * then per node and contact slot {height of the node, |force|} of the last substep's contacts (|force| = -1: empty slot) */
#define AGX_CLOTH_REPORT_WORDS(nn) (20 + 2 * AGX_CLOTH_NODE_CONTACTS * (nn))
/* behind the report, per environment: the cloth kernel's contact records of the current substep, 8 floats per (node, contact slot):
 * normal (3), offset, friction factor, impulse sum of the last substep (3) */
#define AGX_CLOTH_SCRATCH_WORDS(nn) (8 * AGX_CLOTH_NODE_CONTACTS * (nn))

#define AGX_MLP_HIDDEN 64
#define AGX_MLP_WORDS (4 * 64 + 64 + 64 * 64 + 64 + 64 * 64 + 64 + 64 + 1)

/* ---- per-step outputs --------------------------------------------------------------------- */
enum { AGX_INFO_TOTAL_FORCE = 0, AGX_INFO_TASK_SUCCESS = 1, AGX_INFO_ROBOT_FORCE = 2,
       AGX_INFO_TOOL_FORCE = 3,  /* feeding: spoon force on the human; bed bathing: tool_force_on_human (link 1) */
       AGX_INFO_FOOD_REWARD = 4, /* feeding: food reward; bed bathing: new_contact_points of this step          */
       AGX_INFO_PREF = 5,
       AGX_INFO_NCONTACT = 6,    /* solver contacts of the last substep + 1000 * contacts dropped by a budget (overflow) */
       AGX_INFO_NROWS = 7, AGX_INFO_COUNT = 8 };
/* AGX_INFO_NCONTACT of an environment whose state was found non-finite after the step: observation and reward zeroed, done set */
#define AGX_INFO_NONFINITE 1.0e6f

#endif
