// agx_ctx.h -- compile-time limits, the LDS / scratch layouts, the per-lane context and blob accessors.
// Part of the stepper (see agx_step.h for the overview); included by agx_step.h only.
#pragma once

namespace agx {

// A variant of the kernels is compiled per task (agx_kernels.hip is built once per variant); the limits below size
// its LDS / register footprint.  Defaults = FeedingJaco (10 robot + 4 head DoFs, tool + bowl + 8 particles).
#ifndef AGX_MAX_DOF
#define AGX_MAX_DOF 16
#endif
#ifndef AGX_MAX_FREE
#define AGX_MAX_FREE 10
#endif
#ifndef AGX_MAX_BLOCK      // DoFs of the largest articulated body (robot or human chain): M^-1 is block diagonal
#define AGX_MAX_BLOCK 10
#endif
#ifndef AGX_TASK           // AGX_TASK_* of include/agx_blob.h: the task layer compiled into the finish / observe kernels
#define AGX_TASK 0
#endif
#define AGX_HAS_SAMPLER 1   // the device-side reset generator (agx_reset.h): every task (arm manipulation: the single-arm robots)
constexpr int MAX_DOF = AGX_MAX_DOF;
constexpr int MAX_FREE = AGX_MAX_FREE;
constexpr int MAX_BLOCK = AGX_MAX_BLOCK;
constexpr int TASK = AGX_TASK;
constexpr int MAX_HUMAN = 20;
constexpr int MAX_CON = 64;
constexpr int MAX_ROWS = 160;
#ifndef AGX_ST_WORDS
#define AGX_ST_WORDS 336
#endif
constexpr int ST_WORDS = AGX_ST_WORDS;
constexpr int CON_STRIDE = 16;
// row headers of the scratch record.  Variants whose solve kernel has the row-local sweep (agx_pgs_lv.h: LV_COMPILED -- the same condition) keep TWO
// tables: 32 bytes per row with exactly what a visit of that sweep needs (one 8-word scalar load, agx_pgs_lvs.h; two rows per cache line), and
// behind it 8 words per row for the register sweep (DoF lane masks, mu).  The others keep the one 10-word table of rounds 1-4.
constexpr bool HDR_WIDE = AGX_MAX_DOF <= 16 && AGX_MAX_BLOCK <= 10 && AGX_TASK == AGX_TASK_FEEDING;
constexpr int HDR_STRIDE = HDR_WIDE ? 8 : 10;                              // words per row of the (first) table
#ifndef AGX_ARENA_WORDS   // LDS arena reused per phase: dynamics workspace, then collider AABB table + worklist + candidates
#define AGX_ARENA_WORDS 3592
#endif
constexpr int ARENA_WORDS = AGX_ARENA_WORDS;
constexpr int MAX_QPT = 16;                              // manifold points of the (wiping pad, human) pairs handed to the finish kernel
// collider table stride: world AABB (6) + the collider's own travel distance (1).  -DAGX_AABB_WITHOUT_TRAVEL (A/B knob): without the seventh word
// -- rel_travel() (agx_collide.h) then bounds the narrowphase limit alone -- the worklist gets 256 words of the arena back: 200 entries instead
// of 182, one flush instead of two for a FeedingJaco substep (23 instead of 28 passes per step on the emulator, bit-identical contacts).  Measured
// on the last GPU seconds of round 4 (profiles/r04/r04v_ab_feeding_worklist_200.txt): 469.5 / 468.2 k against 468.8 k -- no difference, so the
// layout the GPU suite ran on stays the default.
#if defined(AGX_AABB_WITHOUT_TRAVEL) && !defined(AGX_NO_REL_TRAVEL)
constexpr int ABS = 6;
#else
constexpr int ABS = 7;
#endif

// ---- LDS layout (float words) -------------------------------------------------------------
constexpr int L_ST = 0;
constexpr int L_VEL = L_ST + ST_WORDS;                   // [128] generalised velocities v*
constexpr int L_LINKP = L_VEL + 128;                     // [MAX_DOF][3] world
constexpr int L_LINKR = L_LINKP + MAX_DOF * 3;           // [MAX_DOF][9]
constexpr int L_S = L_LINKR + MAX_DOF * 9;               // [MAX_DOF][6] joint screw about the ref point
constexpr int L_MINV = L_S + MAX_DOF * 6;                // [MAX_DOF*MAX_DOF]
constexpr int L_FREER = L_MINV + MAX_DOF * MAX_DOF;      // [MAX_FREE][9]
constexpr int L_FIINV = L_FREER + MAX_FREE * 9;          // [MAX_FREE][9]
constexpr int L_BASE = L_FIINV + MAX_FREE * 9;           // p(3) R(9)
constexpr int L_HUMAN = L_BASE + 12;                     // [MAX_HUMAN][12] p(3) R(9)
constexpr int L_MISC = L_HUMAN + MAX_HUMAN * 12;         // ref(3), ee p(3), ee R(9), anc masks (MAX_DOF ints)
constexpr int ANC_WORDS = MAX_DOF > 32 ? 2 : 1;            // ancestor bitmask of a moving link: 1 or 2 ints
constexpr int MISC_WORDS = (15 + ANC_WORDS * MAX_DOF + 7) / 8 * 8;
constexpr int L_WMAG = L_MISC + MISC_WORDS;                      // |angular velocity| per moving body: links [MAX_DOF], free bodies [MAX_FREE]
constexpr int WMAG_WORDS = MAX_DOF + MAX_FREE > 32 ? (MAX_DOF + MAX_FREE + 7) / 8 * 8 : 32;
constexpr int L_ARENA = L_WMAG + WMAG_WORDS;                     // contact records live in the per-env global scratch, not in LDS
static_assert(MAX_DOF + MAX_FREE <= 64, "angular speed table: one lane per moving body");
constexpr int LDS_WORDS = L_ARENA + ARENA_WORDS;
static_assert(L_ARENA % 2 == 0, "(J,B) pairs are read as 8-byte words");
constexpr int LDS_BYTES = LDS_WORDS * 4;
// the solve kernel keeps the state copy, the velocity vector and a window of the first
// SOLVE_LDS_PAIRS (J,B) pairs of the environment's rows in LDS (16 waves x 9.5 KB fill a CU's 160 KB);
// rows beyond the window stream from the global scratch (L2)
constexpr int L_SOLVE_ENT = L_VEL + 128;
// Build-time knobs for same-box A/B runs (tools/ab_build.sh): -DAGX_SOLVE_LDS_PAIRS=n (size of the solve kernel's
// LDS row window), -DAGX_NO_LDS_ROWS (all rows from global memory), -DAGX_PGS_CPP (the C++ twin of the assembly sweep).
#ifndef AGX_SOLVE_LDS_PAIRS
#define AGX_SOLVE_LDS_PAIRS 960
#endif
constexpr int SOLVE_LDS_PAIRS = AGX_SOLVE_LDS_PAIRS;
static_assert(L_SOLVE_ENT % 2 == 0, "(J,B) pairs are read as 8-byte words");
// Row-space solve (agx_pgs.h pgs_rowspace; the variants other than FeedingJaco, whose environments have few solver rows): the dense
// Jacobians of up to RS_MAX_ROWS rows and their coupling matrix A = J M^-1 J^T replace the (J,B) window in LDS
constexpr int RS_MAX_ROWS = TASK != AGX_TASK_FEEDING && MAX_DOF <= 32 ? 56 : 0, RS_NVP = (MAX_DOF + 6 * MAX_FREE) | 1;     // row stride of the dense Jacobian: odd = conflict free
// layout of the work area behind the (J,B) window, which the row-space path fills with ALL pairs of the environment (it bails out otherwise)
constexpr int RS_J = 2 * SOLVE_LDS_PAIRS, RS_A = RS_J + RS_MAX_ROWS * RS_NVP, RS_ROW = RS_A + RS_MAX_ROWS * RS_MAX_ROWS, RS_LAM = RS_ROW + 64, RS_WORDS = RS_MAX_ROWS ? RS_LAM + 64 : 0;
constexpr int LDS_SOLVE_WORDS = L_SOLVE_ENT + (2 * SOLVE_LDS_PAIRS > RS_WORDS ? 2 * SOLVE_LDS_PAIRS : RS_WORDS);
constexpr int LDS_SOLVE_BYTES = LDS_SOLVE_WORDS * 4;
// arena, dynamics phase
constexpr int A_COMW = 0;                                // [MAX_DOF][3] rel. ref
constexpr int A_IW = A_COMW + MAX_DOF * 3;               // [MAX_DOF][9]
constexpr int A_VSP = A_IW + MAX_DOF * 9;                // [MAX_DOF][6]
constexpr int A_CVP = A_VSP + MAX_DOF * 6;
constexpr int A_IA = A_CVP + MAX_DOF * 6;                // [MAX_DOF][36]
constexpr int A_U = A_IA + MAX_DOF * 36;
constexpr int A_PA = A_U + MAX_DOF * 6;
constexpr int A_ACC = A_PA + MAX_DOF * 6;
constexpr int A_DINV = A_ACC + MAX_DOF * 6;              // [MAX_DOF]
constexpr int A_UU = A_DINV + MAX_DOF;
constexpr int A_QDD = A_UU + MAX_DOF;
// M^-1 column workspace: [COLS_LANES lanes][MAX_BLOCK][6] + [COLS_LANES][MAX_BLOCK] (links of the lane's own articulated body).  A lane computes
// the columns lane, lane + COLS_LANES, ...: one batch for the task models; the 47-DoF rag doll of the bed-bathing reset takes two batches of
// 24 lanes, which halves its 64 KB workspace -- 97 -> 65 KB of LDS per environment, two environments per CU instead of one
constexpr int COLS_LANES = MAX_DOF > 32 ? (MAX_DOF + 1) / 2 : MAX_DOF;
constexpr int A_COLS = A_QDD + MAX_DOF;
constexpr int A_DYN_END = A_COLS + COLS_LANES * MAX_BLOCK * 6 + COLS_LANES * MAX_BLOCK;
static_assert(A_DYN_END <= ARENA_WORDS, "dynamics workspace exceeds the arena");
// arena, collision phase: world AABBs [ncoll][6]
#ifndef AGX_MAX_COLL
#define AGX_MAX_COLL 256
#endif
constexpr int MAX_COLL = AGX_MAX_COLL;
static_assert(MAX_COLL * 6 <= ARENA_WORDS, "AABB table exceeds the arena");
// misc words
constexpr int M_REF = 0, M_EEP = 3, M_EER = 6, M_ANC = 15;
// contact record
constexpr int C_CA = 0, C_CB = 1, C_BA = 2, C_BB = 3, C_PA = 4, C_PB = 7, C_N = 10, C_DIST = 13, C_MU = 14, C_LAM = 15;
// row header
// ... and, for the wide row-local sweep (agx_pgs_lvw.h: up to four rows with disjoint velocity slots per visit, one per 16-lane group, so every
// lane fetches the header of ITS group's row), two compact tables behind them -- what a visit of that sweep reads per row is 24 bytes:
//   Q, 4 words: 1/D, b, lo, hi (friction rows: 1/D, b, 4 x the row of the contact's normal impulse, mu -- 0 without effective mass); a lane reads
//      word (lane & 3) and takes the others from its quad by DPP;
//   P, 2 words: 8 off | n << 16 | na << 24, and the H_AB word; read whole by every lane, fields picked by SDWA byte / word selects.
// Row HW_DUMMY of both is the idle row (n = 0, 1/D = 0) a lane group without work visits.
constexpr int HW_ROWS = MAX_ROWS + 8, HW_DUMMY = MAX_ROWS, HQ_STRIDE = 4, HP_STRIDE = 2;
constexpr int HDR_TABLE_WORDS = MAX_ROWS * (HDR_WIDE ? 16 : 10);
constexpr int HQ_BASE = HDR_TABLE_WORDS, HP_BASE = HQ_BASE + (HDR_WIDE ? HQ_STRIDE * HW_ROWS : 0), SCR_HDR_WORDS = HP_BASE + (HDR_WIDE ? HP_STRIDE * HW_ROWS : 0);
constexpr int DBG_CON = 16, DBG_MINV = DBG_CON + MAX_CON * CON_STRIDE, DBG_HDR = DBG_MINV + MAX_DOF * MAX_DOF, DBG_LAM = DBG_HDR + SCR_HDR_WORDS,
              DBG_TIME = DBG_LAM + MAX_ROWS, DBG_QDD = DBG_TIME + 24,      // (timers: 16 words of the build kernel's phases, 5..7 the solve kernel's; 16..19: steps / rows / scheduled steps / rows per sweep of the wide sweep)
              DBG_WORDS = DBG_QDD + MAX_DOF;
// per-environment scratch record in HBM (L2-resident while its environment is being solved)
#ifndef AGX_SCR_ENT        // floats of the per-env (J,B) coefficient store: a row keeps one pair per DoF of each articulated block it touches
#define AGX_SCR_ENT 4096
#endif
constexpr int SCR_ENT = AGX_SCR_ENT, SCR_HDR = SCR_HDR_WORDS, SCR_VEL = 128, SCR_CON = MAX_CON * CON_STRIDE, SCR_META = 16;
constexpr int QPT_STRIDE = 4;                            // manifold query point: position on the human (3), PyBullet link of the human collider (int)
constexpr int SCR_QPT = TASK != AGX_TASK_FEEDING ? MAX_QPT * QPT_STRIDE : 0;
constexpr int SCR_O_ENT = 0, SCR_O_HDR = SCR_O_ENT + SCR_ENT, SCR_O_VEL = SCR_O_HDR + SCR_HDR, SCR_O_CON = SCR_O_VEL + SCR_VEL, SCR_O_META = SCR_O_CON + SCR_CON;
constexpr int SCR_O_QPT = SCR_O_META + SCR_META;
// warm-start memory (AGX_P_WARMSTART): per contact of the last solved substep its key -- collider a | collider b << 9 | ordinal inside the
// pair << 18 -- and its solved normal impulse; META_NWARM entries, 0 = none / invalidated
constexpr int SCR_WARM = 2 * MAX_CON, SCR_O_WARM = SCR_O_QPT + SCR_QPT;
// persistent manifold (AGX_P_MANIFOLD, agx_env.h manifold_update): META_NMAN cached points of MP_STRIDE words -- key (collider a | collider b << 9),
// local point on A (3), on B (3), world normal (3), distance, friction -- in cache order
constexpr int MP_STRIDE = 12, MP_KEY = 0, MP_LA = 1, MP_LB = 4, MP_N = 7, MP_DIST = 10, MP_MU = 11;
constexpr int SCR_MAN = MP_STRIDE * MAX_CON, SCR_O_MAN = SCR_O_WARM + SCR_WARM;
constexpr int SCR_WORDS = SCR_O_MAN + SCR_MAN;
constexpr int META_NWARM = 8, META_NMAN = 9;      // (NWARM and NMAN are cleared together: agx_forget_warm_kernel)
constexpr int META_NCON = 0, META_NROWS = 1, META_NNC = 2, META_NEAR = 3, META_OVERFLOW = 4, META_NENT = 5, META_NQPT = 6;
constexpr int H_INVD = 0, H_B = 1, H_LO = 2, H_HI = 3, H_OFF = HDR_WIDE ? 4 : 5, H_N = 5, H_NA = 6, H_AB = 7;      // (H_N, H_NA, H_AB: two-table layout only)
// words of the second table (two-table layout) / of the same row (one table): hx_row()
constexpr int HX_STRIDE = HDR_WIDE ? 8 : 10, HX_BASE = HDR_WIDE ? MAX_ROWS * HDR_STRIDE : 0;
constexpr int H_PACK = HDR_WIDE ? 0 : 4, H_M2 = HDR_WIDE ? 1 : 6, H_MU = HDR_WIDE ? 2 : 7, H_MLO = HDR_WIDE ? 3 : 8, H_MHI = HDR_WIDE ? 4 : 9;
// H_N: pairs of the row, H_NA: of its first DoF range; H_AB: byte offsets of the velocity slots of the two ranges relative to the pair index,
// (4 a0 + H_AB_BIAS) | (4 (b0 - na) + H_AB_BIAS) << 16 -- the slot of pair k is 4 k + (k < na ? A : B) - H_AB_BIAS
constexpr int H_AB_BIAS = 64;
AGX_DEV float* hx_row(float* H, int row) { return H + HX_BASE + HX_STRIDE * row; }
AGX_DEV const float* hx_row(const float* H, int row) { return H + HX_BASE + HX_STRIDE * row; }
constexpr int OFF_TWO_BIT = 31;   // H_OFF bit 31: the row also touches DoFs 64.. (second lane slot)

struct Ctx {
  const float* bf; const int* bi;   // model blob
  float* lds; int* ldsi;
  int lane;
  int ndof, nfree, nhuman, ncoll, ngroup, nfood, nv;
  int nrobot, nhdof, gender, frozen, s_tremor;   // articulated set: robot DoFs [0,nrobot), human DoFs [nrobot,ndof)
  float limit_scale;    // scale of the human joint limits of this environment (impairment 'limits')
  bool coop;            // the human is controllable (TASK.COOP)
  int o_params, o_robot, o_free, o_coll, o_vert, o_group, o_task, o_dirs;
  int s_q, s_qd, s_qt, s_free, s_base, s_human, s_env;
  float dt;             // one internal substep: DT / SIM_SUBSTEPS
  bool hooks;           // this substep ends a p.stepSimulation() call: the limit reset and the arm-limit classifier run (env.py:226-232)
  int ncon, nrows, first_normal, near_mask, overflow;
  float* gqpt; int nqpt;   // bed bathing: manifold points of the (wiping pad, human) pairs (bed_bathing.py:47-58), per-env scratch
  float* dbg;   // optional debug sink (parity tests)
  float* E; float* H;   // constraint rows: (J,B) coefficient pairs and row headers (per-env scratch in HBM/L2)
  int nent;             // (J,B) pairs written by build_rows (entry 0 is the zero pair)
  float* gcon;          // contact records handed from the build kernel to the solve / finish kernels
  long long tm[16]; bool timing;   // per-phase shader-clock totals (debug path only)
};

#define PRM(c, k) ((c).bf[(c).o_params + (k)])
// link record of DoF d: human DoFs have one record per gender
#define RREC(c, d) ((d) < (c).nrobot ? (d) : (d) + (c).gender * (c).nhdof)
#define RBF(c, d, k) ((c).bf[(c).o_robot + RREC(c, d) * AGX_R_STRIDE + (k)])
#define RBI(c, d, k) ((c).bi[(c).o_robot + RREC(c, d) * AGX_R_STRIDE + (k)])
// joint limits of DoF d; the human's are scaled per environment (human_creation.py:199-200)
#define DLO(c, d) (RBF(c, d, AGX_R_LOWER) * ((RBI(c, d, AGX_R_KIND) & 3) == 1 ? (c).limit_scale : 1.f))
#define DHI(c, d) (RBF(c, d, AGX_R_UPPER) * ((RBI(c, d, AGX_R_KIND) & 3) == 1 ? (c).limit_scale : 1.f))
// DoF d made static for this environment (mass-0 links, human.py:104-110): a 32-bit mask, DoFs 32.. are never frozen
#define FROZEN(c, d) ((d) < 32 && (((c).frozen >> (d)) & 1))
#define FBF(c, b, k) ((c).bf[(c).o_free + (b) * AGX_F_STRIDE + (k)])
#define CLF(c, i, k) ((c).bf[(c).o_coll + (i) * AGX_C_STRIDE + (k)])
#define CLI(c, i, k) ((c).bi[(c).o_coll + (i) * AGX_C_STRIDE + (k)])
#define GRI(c, g, k) ((c).bi[(c).o_group + (g) * AGX_G_STRIDE + (k)])
#define TKF(c, k) ((c).bf[(c).o_task + (k)])
#define TKI(c, k) ((c).bi[(c).o_task + (k)])

AGX_DEV void ctx_init(Ctx& c, const uint32_t* blob, float* lds, int lane) {
  c.bf = (const float*)blob; c.bi = (const int*)blob; c.lds = lds; c.ldsi = (int*)lds; c.lane = lane;
  const int* h = c.bi;
  c.ndof = h[AGX_H_NDOF]; c.nfree = h[AGX_H_NFREE]; c.nhuman = h[AGX_H_NHUMAN]; c.ncoll = h[AGX_H_NCOLL];
  c.ngroup = h[AGX_H_NGROUP]; c.nfood = h[AGX_H_NFOOD]; c.nv = c.ndof + 6 * c.nfree;
  c.o_params = h[AGX_H_OFF_PARAMS]; c.o_robot = h[AGX_H_OFF_ROBOT]; c.o_free = h[AGX_H_OFF_FREE]; c.o_coll = h[AGX_H_OFF_COLL];
  c.o_vert = h[AGX_H_OFF_VERT]; c.o_group = h[AGX_H_OFF_GROUP]; c.o_task = h[AGX_H_OFF_TASK]; c.o_dirs = h[AGX_H_OFF_DIRS];
  c.s_q = h[AGX_H_S_Q]; c.s_qd = h[AGX_H_S_QD]; c.s_qt = h[AGX_H_S_QT]; c.s_free = h[AGX_H_S_FREE]; c.s_base = h[AGX_H_S_BASE];
  c.s_human = h[AGX_H_S_HUMAN]; c.s_env = h[AGX_H_S_ENV]; c.s_tremor = h[AGX_H_S_TREMOR];
  c.nrobot = h[AGX_H_NROBOT]; c.nhdof = h[AGX_H_NHDOF]; c.gender = 0; c.frozen = 0; c.limit_scale = 1.f; c.coop = false;
  c.dt = PRM(c, AGX_P_DT) / (float)(h[AGX_H_SIM_SUBSTEPS] > 1 ? h[AGX_H_SIM_SUBSTEPS] : 1); c.hooks = true;
  c.ncon = 0; c.nrows = 0; c.first_normal = 0; c.near_mask = 0; c.overflow = 0; c.nent = 0; c.dbg = nullptr; c.E = nullptr; c.H = nullptr; c.gcon = nullptr; c.gqpt = nullptr; c.nqpt = 0;
  c.timing = false; for (int k = 0; k < 16; k++) c.tm[k] = 0;
}

// ---- small helpers ------------------------------------------------------------------------
AGX_DEV float dot6p(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
// body transform lookup (world rotation + world position) from the LDS tables
AGX_DEV void body_xf(const Ctx& c, int code, m3& R, v3& p) {
  const float* L = c.lds;
  if (code == AGX_BODY_WORLD) { R.a[0] = 1; R.a[1] = 0; R.a[2] = 0; R.a[3] = 0; R.a[4] = 1; R.a[5] = 0; R.a[6] = 0; R.a[7] = 0; R.a[8] = 1; p = mk3(0, 0, 0); }
  else if (code >= AGX_BODY_HUMAN0) { const float* h = L + L_HUMAN + 12 * (code - AGX_BODY_HUMAN0); p = ld3(h); R = ldm3(h + 3); }
  else if (code >= AGX_BODY_FREE0) { int b = code - AGX_BODY_FREE0; p = ld3(L + L_ST + c.s_free + 13 * b); R = ldm3(L + L_FREER + 9 * b); }
  else if (code == AGX_BODY_ROBOT_BASE) { p = ld3(L + L_BASE); R = ldm3(L + L_BASE + 3); }
  else { p = ld3(L + L_LINKP + 3 * code); R = ldm3(L + L_LINKR + 9 * code); }
}

}  // namespace agx
