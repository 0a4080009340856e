// agx_step.h -- the batched FeedingJaco stepper: ONE WAVEFRONT PER ENVIRONMENT.
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
#pragma once
#include "agx_math.h"
#include "agx_gjk.h"
#include "../../include/agx_blob.h"

namespace agx {

constexpr int MAX_DOF = 16;
constexpr int MAX_FREE = 10;
constexpr int MAX_HUMAN = 20;
constexpr int MAX_CON = 64;
constexpr int MAX_ROWS = 160;
constexpr int ST_WORDS = 336;
constexpr int CON_STRIDE = 16;
constexpr int HDR_STRIDE = 10;
constexpr int ARENA_WORDS = 3592;
constexpr int ABS = 7;                                   // collider table stride: world AABB (6) + speculative growth (1)

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
constexpr int L_WMAG = L_MISC + 32;                      // |angular velocity| per moving body: links [MAX_DOF], free bodies [MAX_FREE]
constexpr int L_ARENA = L_WMAG + 32;                     // contact records live in the per-env global scratch, not in LDS
static_assert(MAX_DOF + MAX_FREE <= 32, "angular speed table");
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
constexpr int LDS_SOLVE_WORDS = L_SOLVE_ENT + 2 * SOLVE_LDS_PAIRS;
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
constexpr int A_COLS = A_QDD + MAX_DOF;                  // [MAX_DOF lanes][MAX_DOF][6] M^-1 column workspace
constexpr int A_DYN_END = A_COLS + MAX_DOF * MAX_DOF * 6 + MAX_DOF * MAX_DOF;
static_assert(A_DYN_END <= ARENA_WORDS, "dynamics workspace exceeds the arena");
// arena, collision phase: world AABBs [ncoll][6]
constexpr int MAX_COLL = 256;
static_assert(MAX_COLL * 6 <= ARENA_WORDS, "AABB table exceeds the arena");
// misc words
constexpr int M_REF = 0, M_EEP = 3, M_EER = 6, M_ANC = 15;
// contact record
constexpr int C_CA = 0, C_CB = 1, C_BA = 2, C_BB = 3, C_PA = 4, C_PB = 7, C_N = 10, C_DIST = 13, C_MU = 14, C_LAM = 15;
// row header
constexpr int DBG_HDR = 16 + MAX_CON * CON_STRIDE + MAX_DOF * MAX_DOF, DBG_LAM = DBG_HDR + MAX_ROWS * HDR_STRIDE, DBG_TIME = DBG_LAM + MAX_ROWS, DBG_WORDS = DBG_TIME + 16;
// per-environment scratch record in HBM (L2-resident while its environment is being solved)
constexpr int SCR_ENT = 4096, SCR_HDR = MAX_ROWS * HDR_STRIDE, SCR_VEL = 128, SCR_CON = MAX_CON * CON_STRIDE, SCR_META = 16;
constexpr int SCR_O_ENT = 0, SCR_O_HDR = SCR_O_ENT + SCR_ENT, SCR_O_VEL = SCR_O_HDR + SCR_HDR, SCR_O_CON = SCR_O_VEL + SCR_VEL, SCR_O_META = SCR_O_CON + SCR_CON;
constexpr int SCR_WORDS = SCR_O_META + SCR_META;
constexpr int META_NCON = 0, META_NROWS = 1, META_NNC = 2, META_NEAR = 3, META_OVERFLOW = 4, META_NENT = 5;
constexpr int H_INVD = 0, H_B = 1, H_LO = 2, H_HI = 3, H_PACK = 4, H_OFF = 5, H_M2 = 6, H_MU = 7, H_MLO = 8, H_MHI = 9;
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
  float dt;
  int ncon, nrows, first_normal, near_mask, overflow;
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
#define DLO(c, d) (RBF(c, d, AGX_R_LOWER) * (RBI(c, d, AGX_R_KIND) == 1 ? (c).limit_scale : 1.f))
#define DHI(c, d) (RBF(c, d, AGX_R_UPPER) * (RBI(c, d, AGX_R_KIND) == 1 ? (c).limit_scale : 1.f))
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
  c.dt = PRM(c, AGX_P_DT);
  c.ncon = 0; c.nrows = 0; c.first_normal = 0; c.near_mask = 0; c.overflow = 0; c.nent = 0; c.dbg = nullptr; c.E = nullptr; c.H = nullptr; c.gcon = nullptr;
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

// ---- K1: kinematics (agent.py:52 getLinkState(computeForwardKinematics)) ---------------------
AGX_DEV void kinematics(Ctx& c) {
  float* L = c.lds; const int lane = c.lane, n = c.ndof;
  // chain walk: every lane computes the same link frame, lane 0 publishes it
  for (int d = 0; d < n; d++) {
    int par = RBI(c, d, AGX_R_PARENT);
    v3 pp; m3 PR;
    if (par == AGX_PARENT_HUMAN_BASE) { pp = ld3(L + L_HUMAN); PR = ldm3(L + L_HUMAN + 3); }
    else if (par < 0) { pp = ld3(L + L_BASE); PR = ldm3(L + L_BASE + 3); } else { pp = ld3(L + L_LINKP + 3 * par); PR = ldm3(L + L_LINKR + 9 * par); }
    v3 tp = mk3(RBF(c, d, AGX_R_TPOS), RBF(c, d, AGX_R_TPOS + 1), RBF(c, d, AGX_R_TPOS + 2));
    m3 Rt = quat_to_m3(RBF(c, d, AGX_R_TQUAT), RBF(c, d, AGX_R_TQUAT + 1), RBF(c, d, AGX_R_TQUAT + 2), RBF(c, d, AGX_R_TQUAT + 3));
    v3 ax = mk3(RBF(c, d, AGX_R_AXIS), RBF(c, d, AGX_R_AXIS + 1), RBF(c, d, AGX_R_AXIS + 2));
    m3 Rq = axis_angle_m3(ax, L[L_ST + c.s_q + d]);
    m3 R = mul(mul(PR, Rt), Rq);
    v3 p = mul(PR, tp) + pp;
    wave_sync();
    if (lane == 0) { st3(L + L_LINKP + 3 * d, p); stm3(L + L_LINKR + 9 * d, R); }
    wave_sync();
  }
  // end-effector frame and the reference point for the spatial algebra (keeps f32 lever arms short)
  {
    int ee = TKI(c, AGX_T_EE_LINK);
    v3 lp = ld3(L + L_LINKP + 3 * ee); m3 LR = ldm3(L + L_LINKR + 9 * ee);
    v3 ep = mul(LR, mk3(TKF(c, AGX_T_EE_POS), TKF(c, AGX_T_EE_POS + 1), TKF(c, AGX_T_EE_POS + 2))) + lp;
    m3 ER = mul(LR, quat_to_m3(TKF(c, AGX_T_EE_QUAT), TKF(c, AGX_T_EE_QUAT + 1), TKF(c, AGX_T_EE_QUAT + 2), TKF(c, AGX_T_EE_QUAT + 3)));
    if (lane == 0) { st3(L + L_MISC + M_REF, lp); st3(L + L_MISC + M_EEP, ep); stm3(L + L_MISC + M_EER, ER); }
  }
  wave_sync();
  const v3 ref = ld3(L + L_MISC + M_REF);
  float* A = L + L_ARENA;
  if (lane < n) {
    const int d = lane;
    v3 p = ld3(L + L_LINKP + 3 * d) - ref; m3 R = ldm3(L + L_LINKR + 9 * d);
    v3 aw = mul(R, mk3(RBF(c, d, AGX_R_AXIS), RBF(c, d, AGX_R_AXIS + 1), RBF(c, d, AGX_R_AXIS + 2)));
    v3 pxa = cross(p, aw);
    st3(L + L_S + 6 * d, aw); st3(L + L_S + 6 * d + 3, pxa);
    v3 cw = mul(R, mk3(RBF(c, d, AGX_R_COM), RBF(c, d, AGX_R_COM + 1), RBF(c, d, AGX_R_COM + 2))) + p;
    st3(A + A_COMW + 3 * d, cw);
    m3 Il;
    Il.a[0] = RBF(c, d, AGX_R_INERTIA); Il.a[4] = RBF(c, d, AGX_R_INERTIA + 1); Il.a[8] = RBF(c, d, AGX_R_INERTIA + 2);
    Il.a[1] = Il.a[3] = RBF(c, d, AGX_R_INERTIA + 3); Il.a[2] = Il.a[6] = RBF(c, d, AGX_R_INERTIA + 4); Il.a[5] = Il.a[7] = RBF(c, d, AGX_R_INERTIA + 5);
    stm3(A + A_IW + 9 * d, mul_bt(mul(R, Il), R));
  }
  wave_sync();
  if (lane < n) {
    const int d = lane;
    float v[6] = {0, 0, 0, 0, 0, 0};
    for (int k = d; k >= 0; k = RBI(c, k, AGX_R_PARENT)) { float qd = L[L_ST + c.s_qd + k]; for (int j = 0; j < 6; j++) v[j] += L[L_S + 6 * k + j] * qd; }
    float qd = L[L_ST + c.s_qd + d];
    v3 w = mk3(v[0], v[1], v[2]), vo = mk3(v[3], v[4], v[5]);
    v3 sw = qd * ld3(L + L_S + 6 * d), sv = qd * ld3(L + L_S + 6 * d + 3);
    v3 ca = cross(w, sw), cl = cross(w, sv) + cross(vo, sw);
    for (int j = 0; j < 6; j++) A[A_VSP + 6 * d + j] = v[j];
    st3(A + A_CVP + 6 * d, ca); st3(A + A_CVP + 6 * d + 3, cl);
  }
  // free bodies: rotation matrices and world inverse inertia
  if (lane < c.nfree) {
    const int b = lane; const float* r = L + L_ST + c.s_free + 13 * b;
    m3 R = quat_to_m3(r[3], r[4], r[5], r[6]);
    stm3(L + L_FREER + 9 * b, R);
    m3 Di; for (int k = 0; k < 9; k++) Di.a[k] = 0;
    for (int k = 0; k < 3; k++) { float I = FBF(c, b, AGX_F_INERTIA + k); Di.a[4 * k] = I > 0 ? 1.0f / I : 0.0f; }
    stm3(L + L_FIINV + 9 * b, mul_bt(mul(R, Di), R));
  }
  wave_sync();
}

// ---- K4: articulated-body algorithm, world-frame spatial algebra about the ref point ---------
AGX_DEV float skewc(v3 c, int i, int j) {   // [c]x entry (i,j)
  if (i == j) return 0.f;
  int k = 3 - i - j; float s = ((j - i + 3) % 3 == 1) ? -1.f : 1.f;
  return s * comp(c, k);
}
AGX_DEV void aba_and_minv(Ctx& c) {
  float* L = c.lds; float* A = L + L_ARENA; const int lane = c.lane, n = c.ndof;
  const float kl = PRM(c, AGX_P_LIN_DAMP), ka = PRM(c, AGX_P_ANG_DAMP);
  // spatial inertias -> IA (lanes = matrix entries), bias forces -> pA (lanes = links)
  if (lane < 36) {
    const int r = lane / 6, cc = lane % 6;
    for (int d = 0; d < n; d++) {
      float m = RBF(c, d, AGX_R_MASS); v3 cw = ld3(A + A_COMW + 3 * d);
      float val;
      if (r < 3 && cc < 3) val = A[A_IW + 9 * d + 3 * r + cc] + m * ((r == cc ? dot(cw, cw) : 0.f) - comp(cw, r) * comp(cw, cc));
      else if (r < 3) val = m * skewc(cw, r, cc - 3);
      else if (cc < 3) val = m * skewc(cw, cc, r - 3);
      else val = (r == cc) ? m : 0.f;
      A[A_IA + 36 * d + lane] = val;
    }
  }
  if (lane < n) {
    const int d = lane;
    float m = RBF(c, d, AGX_R_MASS); v3 cw = ld3(A + A_COMW + 3 * d); m3 Iw = ldm3(A + A_IW + 9 * d);
    v3 w = ld3(A + A_VSP + 6 * d), vo = ld3(A + A_VSP + 6 * d + 3);
    v3 vc = vo + cross(w, cw);
    v3 hl = m * vc, ha = mul(Iw, w) + cross(cw, hl);            // momentum about the ref point
    v3 pa_ang = cross(w, ha) + cross(vo, hl), pa_lin = cross(w, hl);
    // external force: gravity + velocity damping [BULLET-UNVERIFIED, see oracle]
    float sl = kl + kl * sqrtf(dot(vc, vc)), sa = ka + ka * sqrtf(dot(w, w));
    const float gz = PRM(c, RBI(c, d, AGX_R_KIND) == 1 ? AGX_P_HUMAN_GRAVITY_Z : AGX_P_ROBOT_GRAVITY_Z);
    v3 f = mk3(0, 0, m * gz) - (m * sl) * vc;
    v3 tau = -(sa * mul(Iw, w));
    v3 fa = tau + cross(cw, f);
    st3(A + A_PA + 6 * d, pa_ang - fa); st3(A + A_PA + 6 * d + 3, pa_lin - f);
  }
  wave_sync();
  // pass 2: leaves -> root
  for (int d = n - 1; d >= 0; d--) {
    if (lane < 6) { float s = 0; for (int k = 0; k < 6; k++) s += A[A_IA + 36 * d + 6 * lane + k] * L[L_S + 6 * d + k]; A[A_U + 6 * d + lane] = s; }
    wave_sync();
    float D = dot6p(L + L_S + 6 * d, A + A_U + 6 * d);
    float Dinv = (D > 1e-30f && !(c.frozen >> d & 1)) ? 1.0f / D : 0.0f;   // frozen DoF: static link (mass 0, human.py:104-110)
    float u = -RBF(c, d, AGX_R_JDAMP) * L[L_ST + c.s_qd + d] - dot6p(L + L_S + 6 * d, A + A_PA + 6 * d);
    if (lane == 0) { A[A_DINV + d] = Dinv; A[A_UU + d] = u; }
    int par = RBI(c, d, AGX_R_PARENT);
    if (par >= 0) {
      float addp = 0.f;
      if (lane < 6) {
        float s = 0;
        for (int j = 0; j < 6; j++) s += (A[A_IA + 36 * d + 6 * lane + j] - A[A_U + 6 * d + lane] * A[A_U + 6 * d + j] * Dinv) * A[A_CVP + 6 * d + j];
        addp = A[A_PA + 6 * d + lane] + s + A[A_U + 6 * d + lane] * (u * Dinv);
      }
      if (lane < 36) { const int r = lane / 6, cc = lane % 6; A[A_IA + 36 * par + lane] += A[A_IA + 36 * d + lane] - A[A_U + 6 * d + r] * A[A_U + 6 * d + cc] * Dinv; }
      if (lane < 6) A[A_PA + 6 * par + lane] += addp;
    }
    wave_sync();
  }
  // pass 3: root -> leaves (every lane computes the same chain; lane 0 publishes)
  for (int d = 0; d < n; d++) {
    int par = RBI(c, d, AGX_R_PARENT);
    float ap[6];
    for (int k = 0; k < 6; k++) ap[k] = (par < 0 ? 0.f : A[A_ACC + 6 * par + k]) + A[A_CVP + 6 * d + k];
    float qdd = (A[A_UU + d] - dot6p(A + A_U + 6 * d, ap)) * A[A_DINV + d];
    wave_sync();
    if (lane == 0) { A[A_QDD + d] = qdd; for (int k = 0; k < 6; k++) A[A_ACC + 6 * d + k] = ap[k] + L[L_S + 6 * d + k] * qdd; }
    wave_sync();
  }
  // M^-1: lane j = response to a unit force on joint j (Bullet: calcAccelerationDeltasMultiDof)
  if (lane < n) {
    const int j = lane; float* P = A + A_COLS + j * (MAX_DOF * 6);
    for (int k = 0; k < n * 6; k++) P[k] = 0.f;
    float* UU = A + A_COLS + MAX_DOF * MAX_DOF * 6 + j * MAX_DOF;   // per-lane u[] next to the column workspaces
    for (int d = n - 1; d >= 0; d--) {
      float u = (d == j ? 1.f : 0.f) - dot6p(L + L_S + 6 * d, P + 6 * d);
      UU[d] = u;
      int par = RBI(c, d, AGX_R_PARENT);
      if (par >= 0) { float s = u * A[A_DINV + d]; for (int k = 0; k < 6; k++) P[6 * par + k] += P[6 * d + k] + A[A_U + 6 * d + k] * s; }
    }
    // reuse P as the acceleration workspace
    for (int d = 0; d < n; d++) {
      int par = RBI(c, d, AGX_R_PARENT);
      float ap[6]; for (int k = 0; k < 6; k++) ap[k] = par < 0 ? 0.f : P[6 * par + k];
      float qdd = (UU[d] - dot6p(A + A_U + 6 * d, ap)) * A[A_DINV + d];
      for (int k = 0; k < 6; k++) P[6 * d + k] = ap[k] + L[L_S + 6 * d + k] * qdd;
      L[L_MINV + d * MAX_DOF + j] = qdd;
    }
  }
  wave_sync();
  if (c.dbg && lane < n) { c.dbg[4 + lane] = A[A_QDD + lane]; }
}

// ---- unconstrained velocity update -------------------------------------------------------------
AGX_DEV void predict_velocities(Ctx& c) {
  float* L = c.lds; float* A = L + L_ARENA; const int lane = c.lane, n = c.ndof; const float dt = c.dt;
  for (int k = lane; k < 128; k += 64) L[L_VEL + k] = 0.f;
  wave_sync();
  if (lane < n) L[L_VEL + lane] = L[L_ST + c.s_qd + lane] + dt * A[A_QDD + lane];
  if (lane < c.nfree) {
    const int b = lane, o = n + 6 * b; const float* r = L + L_ST + c.s_free + 13 * b;
    v3 v = ld3(r + 7), w = ld3(r + 10);
    float kl = PRM(c, AGX_P_LIN_DAMP), ka = PRM(c, AGX_P_ANG_DAMP);
    float sl = kl + kl * sqrtf(dot(v, v)), sa = ka + ka * sqrtf(dot(w, w));
    v3 g = mk3(0, 0, FBF(c, b, AGX_F_GRAVITY));
    st3(L + L_VEL + o, v + dt * (g - sl * v));
    m3 R = ldm3(L + L_FREER + 9 * b), Ii = ldm3(L + L_FIINV + 9 * b);
    v3 wl = tmul(R, w);
    v3 Iw = mul(R, mk3(FBF(c, b, AGX_F_INERTIA) * wl.x, FBF(c, b, AGX_F_INERTIA + 1) * wl.y, FBF(c, b, AGX_F_INERTIA + 2) * wl.z));
    v3 acc = mul(Ii, -cross(w, Iw));
    st3(L + L_VEL + o + 3, w + dt * (acc - sa * w));
  }
  wave_sync();
}

// velocity of the material point of body `code` at world point x from the generalised velocities
AGX_DEV v3 point_velocity(const Ctx& c, int code, v3 x) {
  const float* L = c.lds;
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) {
    float sv[6] = {0, 0, 0, 0, 0, 0};
    for (int d = code; d >= 0; d = RBI(c, d, AGX_R_PARENT)) { float q = L[L_VEL + d]; for (int k = 0; k < 6; k++) sv[k] += L[L_S + 6 * d + k] * q; }
    v3 xr = x - ld3(L + L_MISC + M_REF);
    return mk3(sv[3], sv[4], sv[5]) + cross(mk3(sv[0], sv[1], sv[2]), xr);
  } else if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) {
    int b = code - AGX_BODY_FREE0, o = c.ndof + 6 * b;
    v3 r = x - ld3(L + L_ST + c.s_free + 13 * b);
    return ld3(L + L_VEL + o) + cross(ld3(L + L_VEL + o + 3), r);
  }
  return mk3(0, 0, 0);
}

// ---- K2/K3: collision ----------------------------------------------------------------------------
struct Cand { v3 pa, pb, n; float dist, gap; };

AGX_DEV void make_shape(const Ctx& c, int col, v3 shift, gjk_shape& s) {
  s.n = CLI(c, col, AGX_C_NVERT);
  s.v = c.bf + c.o_vert + 3 * CLI(c, col, AGX_C_VOFF);
  v3 p; body_xf(c, CLI(c, col, AGX_C_BODY), s.R, p);
  s.p = p - shift; s.box = false;
}
// closest features of colliders (ca, cb); true if the separation (radii included) is below limit
AGX_DEV bool narrowphase(const Ctx& c, int ca, int cb, float limit, Cand& out) {
  const float* AB = c.lds + L_ARENA;
  v3 shift = mk3(0.5f * (AB[ABS * ca] + AB[ABS * ca + 3]), 0.5f * (AB[ABS * ca + 1] + AB[ABS * ca + 4]), 0.5f * (AB[ABS * ca + 2] + AB[ABS * ca + 5]));
  gjk_shape sa, sb; make_shape(c, ca, shift, sa); make_shape(c, cb, shift, sb);
  // large static world boxes (table top, ground): clip to the neighbourhood of A (see oracle)
  if (CLI(c, cb, AGX_C_BODY) == AGX_BODY_WORLD && sb.n == 8 && (CLI(c, cb, AGX_C_TAG) == AGX_TAG_TABLE || CLI(c, cb, AGX_C_TAG) == AGX_TAG_PLANE)) {
    sb.box = true;
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
      lo[k] = fmaxf(AB[ABS * cb + k], AB[ABS * ca + k] - AGX_BOX_CLIP); hi[k] = fminf(AB[ABS * cb + 3 + k], AB[ABS * ca + 3 + k] + AGX_BOX_CLIP);
      if (hi[k] < lo[k]) return false;
    }
    sb.lo = mk3(lo[0], lo[1], lo[2]) - shift; sb.hi = mk3(hi[0], hi[1], hi[2]) - shift;
  }
  float ra = CLF(c, ca, AGX_C_RADIUS), rb = CLF(c, cb, AGX_C_RADIUS);
  float d; v3 pa, pb, n;
  bool pen = gjk_distance(sa, sb, PRM(c, AGX_P_GJK_TOL), (int)PRM(c, AGX_P_GJK_MAXIT), d, pa, pb);
  if (!pen) {
    if (d - ra - rb >= limit) return false;
    n = (1.0f / d) * (pa - pb);
  } else {
    float depth; gjk_penetration(sa, sb, c.bf + c.o_dirs, c.bi[AGX_H_NDIR], depth, n, pa, pb);
    d = -depth;
  }
  out.pa = pa - ra * n + shift; out.pb = pb + rb * n + shift; out.n = n; out.dist = d - ra - rb;
  return true;
}
AGX_DEV float pair_mu(const Ctx& c, int ca, int cb) {
  float plane_mu = c.lds[L_ST + c.s_env + AGX_E_PLANE_FRICTION];
  float mua = CLI(c, ca, AGX_C_TAG) == AGX_TAG_PLANE ? plane_mu : CLF(c, ca, AGX_C_FRICTION);
  float mub = CLI(c, cb, AGX_C_TAG) == AGX_TAG_PLANE ? plane_mu : CLF(c, cb, AGX_C_FRICTION);
  return mua * mub;
}
AGX_DEV void emit_contact(Ctx& c, int slot, int ca, int cb, const Cand& k) {
  float* o = c.gcon + CON_STRIDE * slot; int* oi = (int*)o;
  oi[C_CA] = ca; oi[C_CB] = cb; oi[C_BA] = CLI(c, ca, AGX_C_BODY); oi[C_BB] = CLI(c, cb, AGX_C_BODY);
  st3(o + C_PA, k.pa); st3(o + C_PB, k.pb); st3(o + C_N, k.n); o[C_DIST] = k.dist; o[C_MU] = pair_mu(c, ca, cb); o[C_LAM] = 0.f;
}
// Collision pipeline per substep (K2 + K3), all inside the wave:
//   1. world AABB of every collider (lanes over colliders) -> LDS table
//   2. per static pair group: body-level cull (union AABBs), then a lane-parallel sweep over the
//      |A|x|B| pair grid appends the overlapping pairs to an LDS worklist IN ENUMERATION ORDER
//   3. narrowphase over the worklist, 64 pairs per pass, every lane running its own GJK
//   4. selection: per A collider the `keep` candidates with the smallest predicted gap (or all of
//      them, in order) become contacts.
// The contact order (group, a, selection order) is what the oracle produces, so the solver rows
// are identical.
constexpr int WL_MAX = 200, CAND_STRIDE = 8;
constexpr int A_WL = ABS * MAX_COLL;                      // int[WL_MAX]: a | b << 9 | group << 18
constexpr int A_CAND = A_WL + WL_MAX;                   // float[WL_MAX][CAND_STRIDE]: gap, pa, n, dist (pb = pa - dist n)
static_assert(A_CAND + WL_MAX * CAND_STRIDE <= ARENA_WORDS, "collision workspace exceeds the arena");
static_assert(MAX_COLL <= 512, "collider indices are packed in 9 bits");
constexpr int WL_CAP = WL_MAX - 16;   // the candidate words of the last 16 entries (128 ints) hold the A-collider list of a sweep

AGX_DEV void emit_from_cand(Ctx& c, int slot, int idx) {
  const float* L = c.lds; const int pr = c.ldsi[L_ARENA + A_WL + idx]; const float* cd = L + L_ARENA + A_CAND + CAND_STRIDE * idx;
  Cand k; k.gap = cd[0]; k.pa = ld3(cd + 1); k.n = ld3(cd + 4); k.dist = cd[7]; k.pb = k.pa - k.dist * k.n;
  emit_contact(c, slot, pr & 511, (pr >> 9) & 511, k);
}
struct CollideState { int ncon, near_mask, overflow, maxc; };
// the pair-group table, one group per lane (lane g = group g): read from the blob once per substep and
// broadcast with v_readlane where a group's parameters are needed (a dependent blob load costs an L2 trip)
struct GroupRegs { int a0, a1, b0, b1, flags, keep; float alo[3], ahi[3], blo[3], bhi[3]; };   // + union boxes of the two collider ranges
// |angular velocity| of the body a collider is attached to (0 for the static ones), from the table
// filled at the start of collide()
AGX_DEV float body_wmag(const Ctx& c, int code) {
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) return c.lds[L_WMAG + code];
  if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) return c.lds[L_WMAG + MAX_DOF + (code - AGX_BODY_FREE0)];
  return 0.f;
}
// conservative separation test: every point of collider x lies within |half extents| + radius of
// the centre of its box; collider y lies within its body-frame box inflated by its radius.  True if
// the two are certainly further apart than `reach`.
AGX_DEV bool sphere_box_apart(const Ctx& c, int x, int y, float reach) {
  const float* AB = c.lds + L_ARENA;
  const v3 cx = mk3(0.5f * (AB[ABS * x] + AB[ABS * x + 3]), 0.5f * (AB[ABS * x + 1] + AB[ABS * x + 4]), 0.5f * (AB[ABS * x + 2] + AB[ABS * x + 5]));
  const v3 hx = mk3(CLF(c, x, AGX_C_AABB_H), CLF(c, x, AGX_C_AABB_H + 1), CLF(c, x, AGX_C_AABB_H + 2));
  const float rx = sqrtf(dot(hx, hx)) + CLF(c, x, AGX_C_RADIUS);
  m3 R; v3 p; body_xf(c, CLI(c, y, AGX_C_BODY), R, p);
  const v3 pl = tmul(R, cx - p) - mk3(CLF(c, y, AGX_C_AABB_C), CLF(c, y, AGX_C_AABB_C + 1), CLF(c, y, AGX_C_AABB_C + 2));
  const float dx = fmaxf(fabsf(pl.x) - CLF(c, y, AGX_C_AABB_H), 0.f), dy = fmaxf(fabsf(pl.y) - CLF(c, y, AGX_C_AABB_H + 1), 0.f),
              dz = fmaxf(fabsf(pl.z) - CLF(c, y, AGX_C_AABB_H + 2), 0.f);
  const float lb = sqrtf(dx * dx + dy * dy + dz * dz) - rx - CLF(c, y, AGX_C_RADIUS);
  return lb > reach;
}

// narrowphase + selection over the current worklist (entries of one or several whole groups, in
// enumeration order); appends the resulting contacts
AGX_DEV void collide_flush(Ctx& c, int wn, CollideState& cs, float brk, float slack, const GroupRegs& G) {
  float* L = c.lds; int* WL = c.ldsi + L_ARENA + A_WL; float* CD = L + L_ARENA + A_CAND; const int lane = c.lane;
  const int food0 = c.bi[AGX_H_FOOD0];
  if (wn == 0) return;
  wave_sync();
  long long ct0 = c.timing ? wave_clock() : 0;
  if (c.timing) { c.tm[13] += wn; c.tm[14] += (wn + 63) / 64; }   // debug: narrowphase pairs / passes
  // 3. narrowphase, 64 pairs per pass
  bool any_manifold_query = false;
  for (int base = 0; base < wn; base += 64) {
    const int i = base + lane; const bool has = i < wn;
    Cand k; k.gap = 3.0e38f; bool near = false; int a = 0, g = 0;
    if (has) {
      const int pr = WL[i]; a = pr & 511; const int b = (pr >> 9) & 511; g = pr >> 18;
      if (narrowphase(c, a, b, brk, k)) {
        near = true;
        v3 vr = point_velocity(c, CLI(c, a, AGX_C_BODY), k.pa) - point_velocity(c, CLI(c, b, AGX_C_BODY), k.pb);
        float pg = k.dist + dot(vr, k.n) * c.dt;
        if (pg < slack) k.gap = pg;
      }
      float* cd = CD + CAND_STRIDE * i;
      cd[0] = k.gap; st3(cd + 1, k.pa); st3(cd + 4, k.n); cd[7] = k.dist;
    }
    // a manifold point exists: what getContactPoints(food, human) reports (agent.py:100-116)
    const bool mq = has && near && (GRI(c, g, AGX_G_FLAGS) & 2) && CLI(c, a, AGX_C_TAG) == AGX_TAG_FOOD;
    if (wave_any(mq)) { any_manifold_query = true; for (int f = 0; f < c.nfood; f++) if (wave_any(mq && CLI(c, a, AGX_C_BODY) - AGX_BODY_FREE0 - food0 == f)) cs.near_mask |= 1 << f; }
  }
  (void)any_manifold_query;
  wave_sync();
  if (c.timing) { long long t = wave_clock(); c.tm[11] += t - ct0; ct0 = t; }
  // 4. selection, one (group, A collider) segment at a time.  The loop only decides which candidate
  // becomes which contact slot (SEL[]); the contact records are then written by one lane per contact, so
  // that their blob reads (bodies, friction) overlap instead of queueing up behind each other.
  int* SEL = c.ldsi + L_ARENA + A_CAND + CAND_STRIDE * WL_CAP;      // the sweep's lists are dead by now
  const int ncon0 = cs.ncon;
  int cur = 0;
  while (cur < wn) {
    const int key = WL[cur] & ~(511 << 9);            // group and A collider
    const int g = key >> 18, keep = wave_bcast_i(G.keep, g);
    const int i0 = cur + lane, i1 = cur + 64 + lane;
    const bool s0 = i0 < wn && (WL[i0 < wn ? i0 : 0] & ~(511 << 9)) == key, s1 = i1 < wn && (WL[i1 < wn ? i1 : 0] & ~(511 << 9)) == key;
    const uint64_t b0 = wave_ballot(s0), b1 = wave_ballot(s1);
    // segments are contiguous: the run of matching entries starting at cur
    const int len0 = (~b0) ? ffs64(~b0) : 64;
    const int len = len0 < 64 ? len0 : 64 + ((~b1) ? ffs64(~b1) : 64);
    const bool in0 = lane < len, in1 = 64 + lane < len;
    float g0 = in0 ? CD[CAND_STRIDE * i0] : 3.0e38f, g1 = in1 ? CD[CAND_STRIDE * i1] : 3.0e38f;
    if (keep == 0) {   // keep everything, in enumeration order
      for (int pass = 0; pass < 2; pass++) {
        const bool has = (pass ? g1 : g0) < 1.0e38f;
        const uint64_t m = wave_ballot(has);
        int cnt = popc64(m); const int slot = cs.ncon + wave_rank(m);
        if (has && slot < cs.maxc) SEL[slot - ncon0] = pass ? i1 : i0;
        int room = cs.maxc - cs.ncon; if (room < 0) room = 0;
        if (cnt > room) { cs.overflow += cnt - room; cnt = room; }
        cs.ncon += cnt;
      }
    } else {           // the `keep` smallest predicted gaps, in selection order
      for (int q = 0; q < keep; q++) {
        const float mg = wave_min(fminf(g0, g1));
        if (mg > 1.0e38f) break;
        const uint64_t m0 = wave_ballot(g0 == mg);
        int slot1 = 0, win;
        if (m0) win = ffs64(m0); else { win = ffs64(wave_ballot(g1 == mg)); slot1 = 1; }
        if (cs.ncon < cs.maxc) { if (lane == win) SEL[cs.ncon - ncon0] = slot1 ? i1 : i0; cs.ncon++; } else cs.overflow++;
        if (lane == win) { if (slot1) g1 = 3.0e38f; else g0 = 3.0e38f; }
      }
    }
    cur += len;
  }
  wave_sync();
  if (ncon0 + lane < cs.ncon) emit_from_cand(c, ncon0 + lane, SEL[lane]);      // at most MAX_CON = 64 new contacts
  wave_sync();
  if (c.timing) { long long t = wave_clock(); c.tm[12] += t - ct0; }
}

// broadphase sweep of A colliders [aa, ab) x B range of group g, appended to the worklist at wn.
// returns the new count (may exceed WL_MAX: entries beyond it are not stored)
AGX_DEV int collide_sweep(Ctx& c, int g, int aa, int ab, int b0, int b1, int gflags, float mg, int wn, const GroupRegs& G) {
  const float* AB = c.lds + L_ARENA; int* WL = c.ldsi + L_ARENA + A_WL; const int lane = c.lane;
  const bool same = gflags & 1, no_adjacent = gflags & 4;   // bit2, self-collision: not the same link, not parent and child
  // level 1: the A colliders whose box reaches the union box of the B range and vice versa (the union
  // boxes were computed by the group cull, lane g holds them), compacted in ascending order into two
  // 16-bit lists in the tail of the candidate area (unused until the flush).  Filtering both sides
  // matters for pairs of compounds (64 spoon pieces x 44 wheelchair pieces: a handful of each are close).
  float ulo[2][3], uhi[2][3];
  for (int q = 0; q < 3; q++) { ulo[0][q] = wave_bcast(G.blo[q], g); uhi[0][q] = wave_bcast(G.bhi[q], g); ulo[1][q] = wave_bcast(G.alo[q], g); uhi[1][q] = wave_bcast(G.ahi[q], g); }
  unsigned short* LIST = (unsigned short*)(c.ldsi + L_ARENA + A_CAND + CAND_STRIDE * WL_CAP);   // [0,128): A side, [128,256): B side
  int nlive[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    const int r0 = side == 0 ? aa : b0, r1 = side == 0 ? ab : b1;
    for (int base = r0; base < r1; base += 64) {
      const int x = base + lane; bool ok = x < r1;
      if (ok) for (int q = 0; q < 3; q++) if (AB[ABS * x + q] > uhi[side][q] + mg || ulo[side][q] > AB[ABS * x + 3 + q] + mg) ok = false;
      const uint64_t m = wave_ballot(ok);
      if (ok) LIST[128 * side + nlive[side] + wave_rank(m)] = (unsigned short)x;
      nlive[side] += popc64(m);
    }
  }
  const int na_live = nlive[0], nb = nlive[1];
  wave_sync();
  // level 2: the pair grid of the surviving A colliders, in enumeration order
  const int npairs = na_live * nb;
  for (int base = 0; base < npairs; base += 64) {
    const int p = base + lane; bool ok = p < npairs;
    const int ai = ok ? p / nb : 0; const int a = LIST[ai], b = LIST[128 + (ok ? p - ai * nb : 0)];
    ok = ok && (!same || b > a);
    if (ok && no_adjacent) {
      const int la = CLI(c, a, AGX_C_BODY), lb = CLI(c, b, AGX_C_BODY);
      if (la == lb) ok = false;
      else if (la >= 0 && la < AGX_BODY_ROBOT_BASE && lb >= 0 && lb < AGX_BODY_ROBOT_BASE && (RBI(c, la, AGX_R_PARENT) == lb || RBI(c, lb, AGX_R_PARENT) == la)) ok = false;
    }
    if (ok) for (int q = 0; q < 3; q++) if (AB[ABS * a + q] > AB[ABS * b + 3 + q] + mg || AB[ABS * b + q] > AB[ABS * a + 3 + q] + mg) ok = false;
    // level 3: bounding sphere of one collider against the body-frame box of the other, both ways
    if (ok) { const float reach = mg + AB[ABS * a + 6] + AB[ABS * b + 6] + 1e-5f; ok = !sphere_box_apart(c, a, b, reach) && !sphere_box_apart(c, b, a, reach); }
    const uint64_t m = wave_ballot(ok);
    const int slot = wn + wave_rank(m);
    if (ok && slot < WL_CAP) WL[slot] = a | (b << 9) | (g << 18);
    wn += popc64(m);
  }
  wave_sync();
  return wn;
}

AGX_DEV void collide(Ctx& c) {
  float* L = c.lds; float* AB = L + L_ARENA; const int lane = c.lane;
  const float brk = PRM(c, AGX_P_CONTACT_BREAK), slack = PRM(c, AGX_P_CONTACT_SLACK);
  CollideState cs; cs.ncon = 0; cs.near_mask = 0; cs.overflow = 0;
  cs.maxc = (int)PRM(c, AGX_P_MAX_CONTACTS); if (cs.maxc > MAX_CON) cs.maxc = MAX_CON;
  long long ct0 = c.timing ? wave_clock() : 0, ct1;
#define AGX_CTICK(k) if (c.timing) { ct1 = wave_clock(); c.tm[k] += ct1 - ct0; ct0 = ct1; }
  // 1. world AABBs, grown by the distance the collider can travel in this substep (speculative):
  //    the broadphase margin then only has to cover the solver slack
  if (lane < MAX_DOF + MAX_FREE) {     // angular speed of every moving body, once (the chain walk is per body, not per collider)
    v3 w = mk3(0, 0, 0);
    if (lane < MAX_DOF) { if (lane < c.ndof) for (int d = lane; d >= 0; d = RBI(c, d, AGX_R_PARENT)) w = w + L[L_VEL + d] * ld3(L + L_S + 6 * d); }
    else if (lane - MAX_DOF < c.nfree) w = ld3(L + L_VEL + c.ndof + 6 * (lane - MAX_DOF) + 3);
    L[L_WMAG + lane] = sqrtf(dot(w, w));
  }
  wave_sync();
  for (int col = lane; col < c.ncoll; col += 64) {
    const int code = CLI(c, col, AGX_C_BODY);
    m3 R; v3 p; body_xf(c, code, R, p);
    v3 cl = mk3(CLF(c, col, AGX_C_AABB_C), CLF(c, col, AGX_C_AABB_C + 1), CLF(c, col, AGX_C_AABB_C + 2));
    v3 hl = mk3(CLF(c, col, AGX_C_AABB_H), CLF(c, col, AGX_C_AABB_H + 1), CLF(c, col, AGX_C_AABB_H + 2));
    v3 cw = mul(R, cl) + p; float r = CLF(c, col, AGX_C_RADIUS);
    v3 vc = point_velocity(c, code, cw);
    const float wmag = body_wmag(c, code);
    const float grow = (sqrtf(dot(vc, vc)) + wmag * (sqrtf(dot(hl, hl)) + r)) * c.dt;
    for (int k = 0; k < 3; k++) {
      float h = fabsf(R.a[3 * k]) * hl.x + fabsf(R.a[3 * k + 1]) * hl.y + fabsf(R.a[3 * k + 2]) * hl.z + r;
      AB[ABS * col + k] = comp(cw, k) - h - grow; AB[ABS * col + 3 + k] = comp(cw, k) + h + grow;
    }
    AB[ABS * col + 6] = grow;
  }
  wave_sync();
  AGX_CTICK(8)
  const int gender = c.ldsi[L_ST + c.s_env + AGX_E_GENDER];
  // body-level cull of every group at once: lane g scans both collider ranges of group g
  uint64_t live_groups = 0;
  GroupRegs G; G.a0 = 0; G.a1 = 0; G.b0 = 0; G.b1 = 0; G.flags = 0; G.keep = 0;
  for (int k = 0; k < 3; k++) { G.alo[k] = 0.f; G.ahi[k] = 0.f; G.blo[k] = 0.f; G.bhi[k] = 0.f; }
  {
    const int g = lane; bool live = false;
    if (g < c.ngroup) {
      G.a0 = GRI(c, g, AGX_G_A0); G.a1 = GRI(c, g, AGX_G_A1); G.b0 = GRI(c, g, AGX_G_B0); G.b1 = GRI(c, g, AGX_G_B1);
      if (gender == 1 && GRI(c, g, AGX_G_B0F) >= 0) { G.b0 = GRI(c, g, AGX_G_B0F); G.b1 = GRI(c, g, AGX_G_B1F); }
      G.flags = GRI(c, g, AGX_G_FLAGS); G.keep = GRI(c, g, AGX_G_KEEP);
      const int a0 = G.a0, a1 = G.a1, b0 = G.b0, b1 = G.b1;
      const float mg = (G.flags & 2) ? brk : slack;
      if (a1 > a0 && b1 > b0) {
        float alo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ahi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, blo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int i = a0; i < a1; i++) for (int k = 0; k < 3; k++) { alo[k] = fminf(alo[k], AB[ABS * i + k]); ahi[k] = fmaxf(ahi[k], AB[ABS * i + 3 + k]); }
        for (int i = b0; i < b1; i++) for (int k = 0; k < 3; k++) { blo[k] = fminf(blo[k], AB[ABS * i + k]); bhi[k] = fmaxf(bhi[k], AB[ABS * i + 3 + k]); }
        live = true;
        for (int k = 0; k < 3; k++) if (alo[k] > bhi[k] + mg || blo[k] > ahi[k] + mg) live = false;
        for (int k = 0; k < 3; k++) { G.alo[k] = alo[k]; G.ahi[k] = ahi[k]; G.blo[k] = blo[k]; G.bhi[k] = bhi[k]; }
      }
    }
    live_groups = wave_ballot(live);   // the pair table has at most 64 groups (checked in agx_create)
  }
  AGX_CTICK(9)
  // 2.-4. The work is a sequence of units (group, range of its A colliders), normally one unit per
  // live group.  Units are swept into the shared worklist until one does not fit behind the pending
  // entries; then the pending entries are flushed (narrowphase + selection) and the unit is retried
  // on the empty list, split into batches of whole A colliders if it still does not fit.  One sweep
  // and one flush call site keep the kernel's code size down.
  int wn = 0, g = 0, ab = -1, abatch = 0;
  while (true) {
    while (g < c.ngroup && !(live_groups >> g & 1)) g++;
    if (g >= c.ngroup) {
      if (wn == 0) break;
    } else {
      const int a0 = wave_bcast_i(G.a0, g), a1 = wave_bcast_i(G.a1, g), b0 = wave_bcast_i(G.b0, g), b1 = wave_bcast_i(G.b1, g);
      const int gflags = wave_bcast_i(G.flags, g);
      const float mg = (gflags & 2) ? brk : slack;   // bit1: getContactPoints-style existence query
      const int nb = b1 - b0;
      if (ab < 0) { ab = a0; abatch = a1 - a0; }
      const int ae = ab + abatch < a1 ? ab + abatch : a1;
      int wn2 = collide_sweep(c, g, ab, ae, b0, b1, gflags, mg, wn, G);
      AGX_CTICK(10)
      bool fits = wn2 <= WL_CAP;
      if (!fits && wn == 0) {
        const int small = WL_CAP / nb > 0 ? WL_CAP / nb : 1;   // a batch of WL_CAP / nb colliders cannot overflow
        if (abatch > small) { abatch = small; continue; }       // retry this unit in smaller batches
        cs.overflow += wn2 - WL_CAP; wn2 = WL_CAP; fits = true;  // a single A collider with more than WL_CAP partners
      }
      if (fits) {
        wn = wn2; ab = ae;
        if (ab >= a1) { g++; ab = -1; }
        // a unit that was split is flushed batch by batch; whole groups keep accumulating
        if (ab < 0) continue;
      }
    }
    collide_flush(c, wn, cs, brk, slack, G); wn = 0;
    ct0 = c.timing ? wave_clock() : 0;
  }
  c.ncon = cs.ncon; c.near_mask = cs.near_mask; c.overflow = cs.overflow;
  wave_sync();
#undef AGX_CTICK
}

// ---- K5: constraint rows -----------------------------------------------------------------------------
// accumulate the Jacobian of a unit force f / unit torque t applied to body `code` at world point x
AGX_DEV void add_jac(const Ctx& c, int code, v3 x, v3 f, v3 t, float sign, float* Jr, float* Jf) {
  const float* L = c.lds;
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) {
    v3 xr = x - ld3(L + L_MISC + M_REF);
    v3 Fa = cross(xr, f) + t;
    float F[6] = {Fa.x, Fa.y, Fa.z, f.x, f.y, f.z};
    const int anc = c.ldsi[L_MISC + M_ANC + code];
    _Pragma("unroll") for (int d = 0; d < MAX_DOF; d++) if (d < c.ndof && (anc >> d & 1)) Jr[d] += sign * dot6p(L + L_S + 6 * d, F);
  } else if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) {
    int b = code - AGX_BODY_FREE0;
    v3 r = x - ld3(L + L_ST + c.s_free + 13 * b);
    v3 ta = cross(r, f) + t;
    Jf[0] += sign * f.x; Jf[1] += sign * f.y; Jf[2] += sign * f.z; Jf[3] += sign * ta.x; Jf[4] += sign * ta.y; Jf[5] += sign * ta.z;
  }
}
struct RowGeom { float Jr[MAX_DOF]; float Ja[6]; float Jb[6]; int fa, fb; bool robot, human; };   // fa/fb: free body index or -1; robot/human: articulated blocks touched
AGX_DEV void row_clear(RowGeom& r) { for (int k = 0; k < MAX_DOF; k++) r.Jr[k] = 0.f; for (int k = 0; k < 6; k++) { r.Ja[k] = 0.f; r.Jb[k] = 0.f; } r.fa = -1; r.fb = -1; r.robot = false; r.human = false; }
// force +f (torque +t) on body A at xa, -f (-t) on body B at xb
AGX_DEV void row_pair(const Ctx& c, RowGeom& r, int codeA, v3 xa, int codeB, v3 xb, v3 f, v3 t) {
  row_clear(r);
  if (codeA >= 0 && codeA < AGX_BODY_ROBOT_BASE) { if (codeA < c.nrobot) r.robot = true; else r.human = true; }
  if (codeB >= 0 && codeB < AGX_BODY_ROBOT_BASE) { if (codeB < c.nrobot) r.robot = true; else r.human = true; }
  if (codeA >= AGX_BODY_FREE0 && codeA < AGX_BODY_HUMAN0) r.fa = codeA - AGX_BODY_FREE0;
  if (codeB >= AGX_BODY_FREE0 && codeB < AGX_BODY_HUMAN0) r.fb = codeB - AGX_BODY_FREE0;
  add_jac(c, codeA, xa, f, t, 1.f, r.Jr, r.Ja);
  add_jac(c, codeB, xb, f, t, -1.f, r.Jr, r.Jb);
}
AGX_DEV float row_velocity(const Ctx& c, const RowGeom& r) {
  const float* L = c.lds; float s = 0.f;
  if (r.robot || r.human) { _Pragma("unroll") for (int d = 0; d < MAX_DOF; d++) if (d < c.ndof) s += r.Jr[d] * L[L_VEL + d]; }
  if (r.fa >= 0) for (int k = 0; k < 6; k++) s += r.Ja[k] * L[L_VEL + c.ndof + 6 * r.fa + k];
  if (r.fb >= 0) for (int k = 0; k < 6; k++) s += r.Jb[k] * L[L_VEL + c.ndof + 6 * r.fb + k];
  return s;
}
// articulated DoF range a row stores: the robot block, the human block, or both (contiguous)
AGX_DEV void row_art_range(const Ctx& c, const RowGeom& r, int& lo, int& n) {
  lo = r.robot ? 0 : c.nrobot;
  n = (r.robot && r.human) ? c.ndof : (r.robot ? c.nrobot : (r.human ? c.nhdof : 0));
}
AGX_DEV int row_entries(const Ctx& c, const RowGeom& r) { int lo, n; row_art_range(c, r, lo, n); return n + (r.fa >= 0 ? 6 : 0) + (r.fb >= 0 ? 6 : 0); }
// B = M^-1 J^T, D = J B; stores the (J,B) pairs and the header of row `row` at entry offset `off`.
// A row addresses at most two contiguous DoF ranges: [a0,a0+na) and [b0,b0+nb).
AGX_DEV void row_store(const Ctx& c, const RowGeom& r, int row, int off, float bterm, float lo, float hi, int fric_of, float mu) {
  float* L = c.lds; float* E = c.E + 2 * off; const int n = c.ndof;
  float D = 0.f; int e = 0;
  int a0 = 0, na = 0, b0 = 0, nb = 0;
  int alo, an; row_art_range(c, r, alo, an);
  const bool art = an > 0;
  if (art) {
    a0 = alo; na = an;
    _Pragma("unroll") for (int i = 0; i < MAX_DOF; i++) if (i >= alo && i < alo + an) {
      float acc = 0.f;
      _Pragma("unroll") for (int j = 0; j < MAX_DOF; j++) if (j >= alo && j < alo + an) acc += L[L_MINV + i * MAX_DOF + j] * r.Jr[j];
      E[2 * e] = r.Jr[i]; E[2 * e + 1] = acc; D += r.Jr[i] * acc; e++;
    }
  }
  // free bodies in ascending DoF order, so that the pairs of a row are stored in lane order (the
  // solver addresses them by the rank of the lane inside the row's lane mask)
  const int first = (r.fa >= 0 && r.fb >= 0 && r.fb < r.fa) ? 1 : 0;
  for (int s2 = 0; s2 < 2; s2++) {
    const int side = s2 ^ first;
    int fb = side == 0 ? r.fa : r.fb; if (fb < 0) continue;
    const float* J = side == 0 ? r.Ja : r.Jb;
    float mass = FBF(c, fb, AGX_F_MASS), im = mass > 0 ? 1.0f / mass : 0.f;
    v3 Ba = mul(ldm3(L + L_FIINV + 9 * fb), mk3(J[3], J[4], J[5]));
    float B[6] = {im * J[0], im * J[1], im * J[2], Ba.x, Ba.y, Ba.z};
    int base = n + 6 * fb;
    if (na == 0 && nb == 0 && !art) { a0 = base; na = 6; } else if (nb == 0) { b0 = base; nb = 6; } else { /* third range cannot occur */ }
    for (int k = 0; k < 6; k++) { E[2 * e] = J[k]; E[2 * e + 1] = B[k]; D += J[k] * B[k]; e++; }
  }
  // a robot + two free bodies would need three ranges; the scene has no such row (checked at build time)
  float* H = c.H + HDR_STRIDE * row; int* Hi = (int*)H;
  H[H_INVD] = D > 1e-12f ? 1.0f / D : 0.f; H[H_B] = bterm; H[H_LO] = lo; H[H_HI] = hi;
  // lane masks of the two DoF ranges: bits 0..63 (first lane slot) and 64.. (second slot)
  const uint64_t ra = na > 0 ? ((~0ull >> (64 - na)) ) : 0ull, rb = nb > 0 ? ((~0ull >> (64 - nb))) : 0ull;
  uint64_t mlo = 0ull, mhi = 0ull;
  if (na > 0) { if (a0 < 64) mlo |= ra << a0; if (a0 + na > 64) mhi |= a0 >= 64 ? ra << (a0 - 64) : ra >> (64 - a0); }
  if (nb > 0) { if (b0 < 64) mlo |= rb << b0; if (b0 + nb > 64) mhi |= b0 >= 64 ? rb << (b0 - 64) : rb >> (64 - b0); }
  Hi[H_PACK] = a0 | (na << 8) | (b0 << 16) | (nb << 24); Hi[H_OFF] = off | (mhi ? (int)(1u << OFF_TWO_BIT) : 0);
  Hi[H_M2] = (int)(uint32_t)mhi; H[H_MU] = mu; Hi[H_MLO] = (int)(uint32_t)mlo; Hi[H_MHI] = (int)(uint32_t)(mlo >> 32);
  (void)fric_of;
}
AGX_DEV void plane_space(v3 n, v3& p) {
  if (fabsf(n.z) > 0.70710678f) { float a = n.y * n.y + n.z * n.z, k = 1.0f / sqrtf(a); p = mk3(0, -n.z * k, n.y * k); }
  else { float a = n.x * n.x + n.y * n.y, k = 1.0f / sqrtf(a); p = mk3(-n.y * k, n.x * k, 0); }
}
AGX_DEV void m3_to_euler_xyz(const m3& M, float* e) {
  const float* R = M.a; float fi = R[2];
  if (fi < 1.0f) { if (fi > -1.0f) { e[0] = atan2f(-R[5], R[8]); e[1] = asinf(R[2]); e[2] = atan2f(-R[1], R[0]); }
    else { e[0] = -atan2f(R[3], R[4]); e[1] = -1.57079632679f; e[2] = 0; } }
  else { e[0] = atan2f(R[3], R[4]); e[1] = 1.57079632679f; e[2] = 0; }
}

AGX_DEV void build_rows(Ctx& c) {
  float* L = c.lds; const int lane = c.lane, n = c.ndof; const float dt = c.dt;
  const float erp = PRM(c, AGX_P_ERP), cerp = PRM(c, AGX_P_CONTACT_ERP);
  int maxrows = (int)PRM(c, AGX_P_MAX_ROWS); if (maxrows > MAX_ROWS) maxrows = MAX_ROWS;
  int maxent = (int)PRM(c, AGX_P_MAX_ENTRIES); if (maxent > SCR_ENT / 2) maxent = SCR_ENT / 2;
  // --- non-contact rows: lanes 0..15 motors, 16..47 joint limits (dof, side), 48..53 tool constraint
  RowGeom r; row_clear(r);
  bool active = false; float bterm = 0.f, lo = 0.f, hi = 0.f;
  if (lane < 16) {
    const int d = lane;
    if (d < n && RBF(c, d, AGX_R_MAXF) > 0.f && !(c.frozen >> d & 1)) {
      active = true; if (d < c.nrobot) r.robot = true; else r.human = true;
      _Pragma("unroll") for (int q = 0; q < MAX_DOF; q++) r.Jr[q] = (q == d) ? 1.f : 0.f;
      // Agent.control (agent.py:28-33): POSITION_CONTROL motor, target dv = kp (q*-q)/dt + kd (0 - qd)
      bterm = RBF(c, d, AGX_R_KP) * (L[L_ST + c.s_qt + d] - L[L_ST + c.s_q + d]) / dt + RBF(c, d, AGX_R_KD) * (0.f - L[L_VEL + d]);
      float lim = RBF(c, d, AGX_R_MAXF) * dt; lo = -lim; hi = lim;
    }
  } else if (lane < 48) {
    const int d = (lane - 16) >> 1, side = (lane - 16) & 1;
    if (d < n && RBI(c, d, AGX_R_HAS_LIMIT) && !(c.frozen >> d & 1)) {
      float q = L[L_ST + c.s_q + d];
      float gap = side == 0 ? q - DLO(c, d) : DHI(c, d) - q;
      if (gap < PRM(c, AGX_P_LIMIT_ACT)) {
        active = true; if (d < c.nrobot) r.robot = true; else r.human = true;
        const float sg = side == 0 ? 1.f : -1.f;
        _Pragma("unroll") for (int q = 0; q < MAX_DOF; q++) r.Jr[q] = (q == d) ? sg : 0.f;
        float rv = sg * L[L_VEL + d];
        bterm = gap > 0 ? (-gap / dt - rv) : (-gap * erp / dt - rv);
        lo = 0.f; hi = 1e30f;
      }
    }
  } else if (lane < 54) {
    // tool fixed constraint (tool.py:46-47)
    const int k = lane - 48; active = true;
    v3 eep = ld3(L + L_MISC + M_EEP); m3 eeR = ldm3(L + L_MISC + M_EER);
    v3 pivA = mul(eeR, mk3(TKF(c, AGX_T_TOOL_POS), TKF(c, AGX_T_TOOL_POS + 1), TKF(c, AGX_T_TOOL_POS + 2))) + eep;
    m3 frameA = mul(eeR, quat_to_m3(TKF(c, AGX_T_TOOL_QUAT), TKF(c, AGX_T_TOOL_QUAT + 1), TKF(c, AGX_T_TOOL_QUAT + 2), TKF(c, AGX_T_TOOL_QUAT + 3)));
    const int tb = c.bi[AGX_H_TOOL_BODY];
    v3 pivB = ld3(L + L_ST + c.s_free + 13 * tb); m3 frameB = ldm3(L + L_FREER + 9 * tb);
    float lim = TKF(c, AGX_T_TOOL_MAXF) * dt; lo = -lim; hi = lim;
    const int link = TKI(c, AGX_T_EE_LINK);
    if (k < 3) {
      v3 nrm = mk3(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f);
      row_pair(c, r, link, pivA, AGX_BODY_FREE0 + tb, pivB, nrm, mk3(0, 0, 0));
      bterm = -comp(pivA - pivB, k) * erp / dt - row_velocity(c, r);
    } else {
      float ang[3]; m3_to_euler_xyz(mul_at(frameA, frameB), ang);
      const int q = k - 3;
      v3 axw = mk3(frameA.a[q], frameA.a[3 + q], frameA.a[6 + q]);
      row_pair(c, r, link, pivA, AGX_BODY_FREE0 + tb, pivB, mk3(0, 0, 0), axw);
      bterm = ang[q] * erp / dt - row_velocity(c, r);
    }
  }
  int cnt = active ? row_entries(c, r) : 0;
  uint64_t am = wave_ballot(active);
  int row = wave_rank(am), off = 1 + wave_scan_excl(cnt);      // entry 0 of the arena is the zero pair
  int nnc = popc64(am), ent = 1 + wave_sum_i(cnt);
  wave_sync();
  if (lane == 0) { c.E[0] = 0.f; c.E[1] = 0.f; }
  // --- contact rows: lane = contact; normal rows first, then one friction row per contact
  int nc = c.ncon;
  const bool has = lane < nc;
  int ba = 0, bb = 0; v3 pa = mk3(0, 0, 0), pb = pa, nn = pa; float dist = 0.f, mu = 0.f;
  RowGeom rn; row_clear(rn);
  if (has) {
    const float* k = c.gcon + CON_STRIDE * lane; const int* ki = (const int*)k;
    ba = ki[C_BA]; bb = ki[C_BB]; pa = ld3(k + C_PA); pb = ld3(k + C_PB); nn = ld3(k + C_N); dist = k[C_DIST]; mu = k[C_MU];
    row_pair(c, rn, ba, pa, bb, pb, nn, mk3(0, 0, 0));
  }
  int ccnt = has ? row_entries(c, rn) : 0;
  int cincl = wave_scan_excl(ccnt) + ccnt;
  // largest prefix of the contact list that fits the row and coefficient budgets
  bool fits = has && (nnc + 2 * (lane + 1) <= maxrows) && (ent + 2 * cincl <= maxent);
  nc = popc64(wave_ballot(fits));
  const int tot = wave_bcast_i(cincl, nc > 0 ? nc - 1 : 0);
  const int entN = ent, entF = ent + (nc > 0 ? tot : 0);
  // the three row kinds go through ONE row_store call site (its M^-1 J^T product is the bulk of this
  // phase's code): kind 0 non-contact, 1 contact normal, 2 contact friction
  _Pragma("nounroll") for (int kind = 0; kind < 3; kind++) {
    RowGeom R; int rrow = 0, roff = 0, rfric = -1; float rb = 0.f, rlo = 0.f, rhi = 0.f, rmu = 0.f; bool go = false;
    if (kind == 0) {
      R = r; go = active; rrow = row; roff = off; rb = bterm; rlo = lo; rhi = hi;
    } else if (kind == 1) {
      R = rn; go = lane < nc; rrow = nnc + lane; roff = entN + cincl - ccnt; rlo = 0.f; rhi = 1e30f;
      if (go) { const float rv = row_velocity(c, rn); rb = dist > 0 ? (-dist / dt - rv) : (-dist * cerp / dt - rv); }
    } else {
      go = lane < nc; rrow = nnc + nc + lane; roff = entF + cincl - ccnt; rfric = nnc + lane; rmu = mu;
      row_clear(R);
      if (go) {
        // friction direction: lateral slip direction if it is resolvable, else the first plane-space tangent
        v3 vr = point_velocity(c, ba, pa) - point_velocity(c, bb, pb);
        v3 t = vr - dot(vr, nn) * nn;
        float l2 = dot(t, t);
        if (l2 > PRM(c, AGX_P_FRIC_EPS)) t = (1.0f / sqrtf(l2)) * t; else plane_space(nn, t);
        row_pair(c, R, ba, pa, bb, pb, t, mk3(0, 0, 0));
        rb = -row_velocity(c, R);
      }
    }
    if (go) row_store(c, R, rrow, roff, rb, rlo, rhi, rfric, rmu);
  }
  c.ncon = nc; c.first_normal = nnc; c.nrows = nnc + 2 * nc; c.nent = entF + (nc > 0 ? tot : 0);
  wave_sync();
}

// ---- K6: projected Gauss-Seidel --------------------------------------------------------------------------
// Rows are visited in construction order (Gauss-Seidel is order dependent).  Row headers and the
// accumulated impulses live in registers, distributed over the lanes (lane r&63 owns row r of slot
// r>>6) and are broadcast with v_readlane (wave-uniform -> SGPRs, scalar control flow).  Per row
// every lane fetches its (J,B) pair from the LDS arena (lanes outside the row's two DoF ranges read
// nothing), one DPP reduction gives J.dv, the impulse update is uniform, every lane applies
// B*dlambda to the DoFs it owns.  The fetch of row r+1 is issued before the reduction of row r.
// Register sets: A0/A1 hold rows 0..127 of the non-contact + normal block (lane r&63 of set r>>6),
// B0/B1 hold the friction rows, placed in the SAME lane as the normal row of their contact so the
// friction bound mu*lambda_n is a lane-local product.  The impulse update is evaluated in every
// lane on its own row registers; only the owner lane's result is kept and its delta broadcast.
struct PgsSet { float invD, b, lo, hi, lam; int pack, off, mlo, mhi, m2; };   // one row per lane (hi = mu for friction sets); off carries OFF_TWO_BIT

// (J,B) pair of this lane for a row: lanes outside the row's two DoF ranges read arena entry 0 = (0,0).
// Addresses are 32-bit byte offsets from the (wave-uniform) entry base, so the loads use the
// SGPR-base + VGPR-offset form and need no 64-bit address arithmetic.
struct PgsBuf { float j0, c0, j1, c1; };
AGX_DEV void pgs_fetch(const float* E, int lane, int pack, int off, PgsBuf& X) {
  const unsigned a0 = pack & 255, na = (pack >> 8) & 255, b0 = (pack >> 16) & 255, nb = (unsigned)pack >> 24;
  const unsigned oa = 8u * (unsigned)off, ob = 8u * ((unsigned)off + na);
  const char* Eb = (const char*)E;
  unsigned ia = (unsigned)lane - a0, ib = (unsigned)lane - b0;
  unsigned e = ib < nb ? ob + 8u * ib : 0u;
  e = ia < na ? oa + 8u * ia : e;
  unsigned e1 = 0u;
  if (a0 + na > 64 || b0 + nb > 64) {   // wave-uniform: only rows touching DoFs 64.. have entries in the second slot
    ia = (unsigned)lane + 64u - a0; ib = (unsigned)lane + 64u - b0;
    e1 = ib < nb ? ob + 8u * ib : 0u;
    e1 = ia < na ? oa + 8u * ia : e1;
  }
  // both loads are always issued (the second one degenerates to a broadcast of the zero pair): the
  // number of loads in flight is then the same on every path and the waits can be exact
  const f2 p = *(const f2*)(Eb + e);
  const f2 q = *(const f2*)(Eb + e1);
  X.j0 = p.x; X.c0 = p.y; X.j1 = q.x; X.c1 = q.y;
}
// one Gauss-Seidel pass over the rows held in lanes [l0, l1) of one register set.  The (J,B) pairs
// stream from the per-env scratch (L2).  Three named buffers rotate through the roles "in use",
// "next" and "being fetched" (the loop is unrolled by three so that no register moves are needed and
// the loads of rows r+1 and r+2 stay in flight while row r is reduced).  Prefetches past the end
// re-read the last row instead of being skipped, again to keep the number of loads in flight fixed.
template <bool FRICTION>
AGX_DEV void pgs_row(PgsSet& S, const float& lam_normal, const PgsBuf& X, int lane, int rl, float& dv0, float& dv1) {
  const float jdv = wave_sum(X.j0 * dv0 + X.j1 * dv1);
  const float hi = FRICTION ? S.hi * lam_normal : S.hi, lo = FRICTION ? -hi : S.lo;
  const float nl = wave_clamp(S.lam + (S.b - jdv) * S.invD, lo, hi);
  const float dlo = nl - S.lam;
  S.lam = (lane == rl) ? nl : S.lam;
  const float dl = wave_bcast(dlo, rl);
  dv0 += X.c0 * dl;
  wave_opaque(dv0);       // keeps the two updates scalar: a packed FMA would need (c0, c1) in adjacent registers
  dv1 += X.c1 * dl;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_CPP)
// ---- the Gauss-Seidel sweep in gfx950 assembly ------------------------------------------------------
// The compiler's schedule of the loop above is poor in exactly the places that matter: it rotates
// the prefetch buffers with register moves and, because the number of loads in flight differs between
// paths, falls back to s_waitcnt vmcnt(0) right after issuing a prefetch.  Here the row loop is written
// out by hand: four register buffers W,X,Y,Z rotate through "in use / +1 / +2 / being fetched"
// (unrolled by four, no moves), every row issues exactly two global_load_dwordx2 (rows that do not
// reach DoFs 64.. load the zero pair for the second slot), so s_waitcnt vmcnt(6) is exact, and the
// prefetch index is clamped to the last row of the sweep instead of being skipped.
// Hazards (gfx940 family; the assembler inserts nothing in inline asm): VALU-written VGPR -> DPP 2
// wait states, VALU-written SGPR/VCC -> VALU read 2, -> v_readlane lane select 4; spacing below
// keeps to these with independent instructions or s_nop.
// Register map: v64..v79 buffers, v80..v88 temporaries, s80..s95 scalars.
#define AGX_STR2(x) #x
#define AGX_STR(x) AGX_STR2(x)
#define AGX_SOLVE_ENT_BYTES 1856
static_assert(AGX_SOLVE_ENT_BYTES == 4 * L_SOLVE_ENT, "LDS offset of the row window used by the assembly");
// the two sources of a row's pairs: the global scratch (vmcnt) or the LDS window (lgkmcnt)
#define AGX_LOAD_G(DST, ADDR) "global_load_dwordx2 " DST ", " ADDR ", %[E]\n"
#define AGX_LOAD_L(DST, ADDR) "ds_read_b64 " DST ", " ADDR " offset:" AGX_STR(AGX_SOLVE_ENT_BYTES) "\n"
#define AGX_WAIT_G(N) "s_waitcnt vmcnt(" N ")\n"
#define AGX_WAIT_L(N) "s_waitcnt lgkmcnt(" N ")\n"
// pairs of row IDX -> buffer (Z0: lanes 0..63, Z1: lanes 64..).  Bit 31 of the offset word (second-slot
// flag) needs no masking: the shift by 3 of the address arithmetic discards it.  The row's lane mask (precomputed by
// row_store) turns the address into "offset + rank of this lane among the row's lanes": 2 x v_mbcnt,
// 1 add-shift, 1 select with the mask itself as the condition.
#define AGX_PGS_FETCH(LOAD, IDX, Z0, Z1) \
  "v_readlane_b32 s84, %[off], " IDX "\n" \
  "v_readlane_b32 s82, %[mlo], " IDX "\n" \
  "v_readlane_b32 s83, %[mhi], " IDX "\n" \
  "s_bitcmp1_b32 s84, 31\n" \
  "v_mbcnt_lo_u32_b32 v81, s82, 0\n" \
  "v_mbcnt_hi_u32_b32 v81, s83, v81\n" \
  "v_add_lshl_u32 v85, v81, s84, 3\n" \
  "v_cndmask_b32_e64 v85, 0, v85, s[82:83]\n" \
  LOAD(Z0, "v85") \
  "s_cbranch_scc0 1f\n" \
  "v_readlane_b32 s86, %[m2], " IDX "\n" \
  "s_bcnt1_i32_b64 s85, s[82:83]\n" \
  "s_mov_b32 s87, 0\n" \
  "s_add_i32 s85, s85, s84\n" \
  "v_mbcnt_lo_u32_b32 v82, s86, 0\n" \
  "v_add_lshl_u32 v86, v82, s85, 3\n" \
  "v_cndmask_b32_e64 v86, 0, v86, s[86:87]\n" \
  LOAD(Z1, "v86") \
  "s_branch 2f\n" \
  "1:\n" \
  LOAD(Z1, "v88") \
  "2:\n"
#define AGX_PGS_DPP(CTRL) "v_add_f32_dpp v80, v80, v80 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// One row: the dependent chain (dot product, 6-step DPP reduction, impulse update, broadcast) with the
// address arithmetic of the prefetch for row r+3 woven into its wait states.
#define AGX_PGS_NEXT(IDX, MASK) "s_ff1_i32_b64 " IDX ", " MASK "\n" "s_bitset0_b64 " MASK ", " IDX "\n"
#define AGX_PGS_STEP(LOAD, WAIT, XJ0, XC0, XJ1, XC1, Z0, Z1) \
  AGX_PGS_NEXT("s94", "%[mask]") \
  AGX_PGS_NEXT("s80", "s[96:97]") \
  WAIT("4") \
  "v_mul_f32_e32 v80, " XJ0 ", %[dv0]\n" \
  "v_fmac_f32_e32 v80, " XJ1 ", %[dv1]\n" \
  "v_readlane_b32 s84, %[off], s80\n" \
  "v_readlane_b32 s82, %[mlo], s80\n" \
  AGX_PGS_DPP("quad_perm:[1,0,3,2]") \
  "v_readlane_b32 s83, %[mhi], s80\n" \
  "s_bitcmp1_b32 s84, 31\n" \
  AGX_PGS_DPP("quad_perm:[2,3,0,1]") \
  "v_mbcnt_lo_u32_b32 v81, s82, 0\n" \
  "v_cmp_eq_u32_e32 vcc, s94, %[lane]\n" \
  AGX_PGS_DPP("row_shr:4") \
  "v_mbcnt_hi_u32_b32 v81, s83, v81\n" \
  "v_add_lshl_u32 v85, v81, s84, 3\n" \
  AGX_PGS_DPP("row_shr:8") \
  "v_cndmask_b32_e64 v85, 0, v85, s[82:83]\n" \
  LOAD(Z0, "v85") \
  AGX_PGS_DPP("row_bcast:15") \
  "s_nop 1\n" \
  AGX_PGS_DPP("row_bcast:31") \
  "s_nop 0\n" \
  "v_readlane_b32 s92, v80, 63\n" \
  "s_nop 1\n" \
  "v_subrev_f32_e32 v80, s92, %[b]\n" \
  "v_fma_f32 v80, %[invD], v80, %[lam]\n" \
  "v_med3_f32 v80, v80, %[lo], %[hi]\n" \
  "v_sub_f32_e32 v87, v80, %[lam]\n" \
  "v_cndmask_b32_e32 %[lam], %[lam], v80, vcc\n" \
  "s_cbranch_scc0 1f\n" \
  "v_readlane_b32 s86, %[m2], s80\n" \
  "s_bcnt1_i32_b64 s85, s[82:83]\n" \
  "s_mov_b32 s87, 0\n" \
  "s_add_i32 s85, s85, s84\n" \
  "v_mbcnt_lo_u32_b32 v82, s86, 0\n" \
  "v_add_lshl_u32 v86, v82, s85, 3\n" \
  "v_cndmask_b32_e64 v86, 0, v86, s[86:87]\n" \
  LOAD(Z1, "v86") \
  "s_branch 2f\n" \
  "1:\n" \
  LOAD(Z1, "v88") \
  "2:\n" \
  "v_readlane_b32 s93, v87, s94\n" \
  "s_cmp_eq_u64 %[mask], 0\n" \
  "s_nop 0\n" \
  "v_fmac_f32_e32 %[dv0], s93, " XC0 "\n" \
  "v_fmac_f32_e32 %[dv1], s93, " XC1 "\n" \
  "s_cbranch_scc1 9f\n"
// The rows to visit are the set bits of %[mask] (lane = row slot), taken in ascending order with
// s_ff1 / s_bitset0; a second cursor (s[96:97]) runs three rows ahead for the prefetch (when it runs
// dry its index is -1, i.e. lane 63: a harmless extra fetch).
#define AGX_PGS_BODY(LOAD, WAIT) \
    "v_mov_b32_e32 v88, 0\n" \
    "s_mov_b64 s[96:97], %[mask]\n" \
    AGX_PGS_NEXT("s80", "s[96:97]") \
    AGX_PGS_FETCH(LOAD, "s80", "v[64:65]", "v[66:67]") \
    AGX_PGS_NEXT("s80", "s[96:97]") \
    AGX_PGS_FETCH(LOAD, "s80", "v[68:69]", "v[70:71]") \
    AGX_PGS_NEXT("s80", "s[96:97]") \
    AGX_PGS_FETCH(LOAD, "s80", "v[72:73]", "v[74:75]") \
    "8:\n" \
    AGX_PGS_STEP(LOAD, WAIT, "v64", "v65", "v66", "v67", "v[76:77]", "v[78:79]") \
    AGX_PGS_STEP(LOAD, WAIT, "v68", "v69", "v70", "v71", "v[64:65]", "v[66:67]") \
    AGX_PGS_STEP(LOAD, WAIT, "v72", "v73", "v74", "v75", "v[68:69]", "v[70:71]") \
    AGX_PGS_STEP(LOAD, WAIT, "v76", "v77", "v78", "v79", "v[72:73]", "v[74:75]") \
    "s_branch 8b\n" \
    "9:\n" \
    WAIT("0")
#define AGX_PGS_OPERANDS \
    : [lam] "+v"(S.lam), [dv0] "+v"(dv0), [dv1] "+v"(dv1), [mask] "+s"(mask) \
    : [off] "v"(S.off), [mlo] "v"(S.mlo), [mhi] "v"(S.mhi), [m2] "v"(S.m2), [invD] "v"(S.invD), [b] "v"(S.b), [lo] "v"(lo), [hi] "v"(hi), \
      [lane] "v"(lane), [E] "s"(E) \
    : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", \
      "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "vcc", "scc", "memory"
// lo/hi are the per-lane bounds of this sweep (for friction sets already scaled by the normal
// impulses).  `rows` has one bit per row slot to visit; rows below slot `ls` have all their pairs inside
// the LDS window, the others stream from global.
AGX_DEV uint64_t pgs_range_mask(int l0, int l1) {
  const uint64_t hi = l1 >= 64 ? ~0ull : ((1ull << l1) - 1ull), lo = l0 >= 64 ? ~0ull : ((1ull << l0) - 1ull);
  return hi & ~lo;
}
AGX_DEV void pgs_sweep_asm(PgsSet& S, float lo, float hi, const float* E, int lane, uint64_t rows, int ls, float& dv0, float& dv1) {
  const uint64_t in_lds = rows & pgs_range_mask(0, ls), in_glb = rows & ~pgs_range_mask(0, ls);
  if (in_lds) {
    uint64_t mask = in_lds;
    asm volatile(AGX_PGS_BODY(AGX_LOAD_L, AGX_WAIT_L) AGX_PGS_OPERANDS);
  }
  if (in_glb) {
    uint64_t mask = in_glb;
    asm volatile(AGX_PGS_BODY(AGX_LOAD_G, AGX_WAIT_G) AGX_PGS_OPERANDS);
  }
}
#endif
template <bool FRICTION>
AGX_DEV void pgs_sweep(PgsSet& S, const float& lam_normal, const float* E, int lane, int l0, int ls, int l1, float& dv0, float& dv1) {
  if (l1 <= l0) return;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AGX_PGS_CPP)
  const float hi = FRICTION ? S.hi * lam_normal : S.hi, lo = FRICTION ? -hi : S.lo;
  uint64_t rows = pgs_range_mask(l0, l1);
  // A friction row whose normal impulse is zero has the bounds [0, 0]; if its own impulse is zero as
  // well its update is exactly "no change", so the visit is skipped.  The normal impulses do not
  // change during a friction sweep and a friction impulse only changes at its own visit, so the set
  // of rows to visit is known up front.  (More than half of the contacts are speculative and inactive.)
  if (FRICTION) rows &= wave_ballot(lam_normal != 0.f || S.lam != 0.f);
  pgs_sweep_asm(S, lo, hi, E, lane, rows, ls, dv0, dv1);
#else
  (void)ls;
  PgsBuf A, B, C;
  const int last = l1 - 1;
#define AGX_PGS_FETCH_C(X, r) { const int rr_ = (r) < last ? (r) : last; pgs_fetch(E, lane, wave_bcast_i(S.pack, rr_), wave_bcast_i(S.off, rr_) & 0x7fffffff, X); }
  AGX_PGS_FETCH_C(A, l0);
  AGX_PGS_FETCH_C(B, l0 + 1);
  for (int rl = l0;; rl += 3) {
    AGX_PGS_FETCH_C(C, rl + 2);
    pgs_row<FRICTION>(S, lam_normal, A, lane, rl, dv0, dv1);
    if (rl + 1 >= l1) break;
    AGX_PGS_FETCH_C(A, rl + 3);
    pgs_row<FRICTION>(S, lam_normal, B, lane, rl + 1, dv0, dv1);
    if (rl + 2 >= l1) break;
    AGX_PGS_FETCH_C(B, rl + 4);
    pgs_row<FRICTION>(S, lam_normal, C, lane, rl + 2, dv0, dv1);
    if (rl + 3 >= l1) break;
  }
#undef AGX_PGS_FETCH_C
#endif
}
AGX_DEV void pgs_load_set(const Ctx& c, int row, bool ok, bool friction, PgsSet& S) {
  ok = ok && row < MAX_ROWS;
  const float* H = c.H + HDR_STRIDE * (ok ? row : 0); const int* Hi = (const int*)H;
  const float invD = ok ? H[H_INVD] : 0.f;
  // a row without effective mass (static-static, degenerate) is kept but pinned at zero impulse
  const bool live = ok && invD != 0.f;
  S.invD = invD; S.b = ok ? H[H_B] : 0.f; S.lam = 0.f;
  S.lo = live ? H[H_LO] : 0.f; S.hi = live ? (friction ? H[H_MU] : H[H_HI]) : 0.f;
  S.pack = ok ? Hi[H_PACK] : 0; S.off = ok ? Hi[H_OFF] : 0;
  S.mlo = ok ? Hi[H_MLO] : 0; S.mhi = ok ? Hi[H_MHI] : 0; S.m2 = ok ? Hi[H_M2] : 0;
}
// first lane of [l0, l1) whose row reaches beyond the LDS window of (J,B) pairs (l1 if none)
AGX_DEV int pgs_lds_split(const PgsSet& S, int lane, int l0, int l1) {
  if (l1 <= l0) return l0;
#ifdef AGX_NO_LDS_ROWS
  return l0;
#endif
  const int end = (S.off & 0x7fffffff) + ((S.pack >> 8) & 255) + (int)((unsigned)S.pack >> 24);
  const uint64_t m = wave_ballot(lane >= l0 && lane < l1 && end > SOLVE_LDS_PAIRS);
  return wave_uniform(m ? ffs64(m) : l1);
}
AGX_DEV void pgs(Ctx& c, float& dv0, float& dv1) {
  const int lane = c.lane; const int iters = (int)PRM(c, AGX_P_NITER);
  const float* E = c.E;
  const int nnc = c.first_normal, nc = c.ncon, nA = nnc + nc;      // rows: [0,nnc) non-contact, [nnc,nA) normals, [nA,nA+nc) friction
  static_assert(MAX_ROWS <= 256 && MAX_CON <= 64, "two register sets per block");
  PgsSet A0, A1, B0, B1;
  pgs_load_set(c, lane, lane < nA, false, A0);
  pgs_load_set(c, 64 + lane, 64 + lane < nA, false, A1);
  { const int c0 = lane - nnc, c1 = 64 + lane - nnc;
    pgs_load_set(c, nA + c0, c0 >= 0 && c0 < nc, true, B0);
    pgs_load_set(c, nA + c1, c1 >= 0 && c1 < nc, true, B1); }
  const int a0n = nA < 64 ? nA : 64, a1n = nA - 64;
  const int f0a = nnc < 64 ? nnc : 64, f0b = nA < 64 ? nA : 64;          // friction rows in B0: lanes [nnc, min(nA,64))
  const int f1a = nnc > 64 ? nnc - 64 : 0, f1b = nA - 64;                // friction rows in B1: lanes [max(nnc-64,0), nA-64)
  dv0 = 0.f; dv1 = 0.f;
  // rows whose pairs lie inside the LDS window (offsets grow with the row index, so per register set
  // this is a prefix of its lane range)
  const int s0 = pgs_lds_split(A0, lane, 0, a0n), s1 = pgs_lds_split(A1, lane, 0, a1n);
  const int t0 = pgs_lds_split(B0, lane, f0a, f0b), t1 = pgs_lds_split(B1, lane, f1a, f1b);
  for (int it = 0; it < iters; it++) {
    pgs_sweep<false>(A0, A0.lam, E, lane, 0, s0, a0n, dv0, dv1);
    pgs_sweep<false>(A1, A1.lam, E, lane, 0, s1, a1n, dv0, dv1);
    pgs_sweep<true>(B0, A0.lam, E, lane, f0a, t0, f0b, dv0, dv1);
    pgs_sweep<true>(B1, A1.lam, E, lane, f1a, t1, f1b, dv0, dv1);
  }
  // solved normal impulses -> contact records (what getContactPoints reports until the next step)
  { const int r0 = lane, r1 = 64 + lane;
    if (r0 >= nnc && r0 < nA) c.gcon[CON_STRIDE * (r0 - nnc) + C_LAM] = A0.lam;
    if (r1 >= nnc && r1 < nA) c.gcon[CON_STRIDE * (r1 - nnc) + C_LAM] = A1.lam; }
}

// ---- K7 + post-substep hooks -----------------------------------------------------------------------------
AGX_DEV void integrate(Ctx& c, const float* gvel, float dv0, float dv1) {
  float* L = c.lds; const int lane = c.lane, n = c.ndof; const float dt = c.dt;
  L[L_VEL + lane] = gvel[lane] + dv0;
  L[L_VEL + lane + 64] = gvel[lane + 64] + dv1;
  wave_sync();
  if (lane < n) {
    const int d = lane;
    float qd = L[L_VEL + d], q = L[L_ST + c.s_q + d] + dt * qd;
    // Agent.enforce_joint_limits on the human after every stepSimulation (env.py:229, agent.py:240-250)
    if (RBI(c, d, AGX_R_KIND) == 1 && !(c.frozen >> d & 1)) {
      const float lo = DLO(c, d), hi = DHI(c, d);
      if (q < lo) { q = lo; qd = 0.f; } else if (q > hi) { q = hi; qd = 0.f; }
    }
    L[L_ST + c.s_qd + d] = qd; L[L_ST + c.s_q + d] = q;
  }
  if (lane < c.nfree) {
    const int b = lane, o = n + 6 * b; float* r = L + L_ST + c.s_free + 13 * b;
    v3 v = ld3(L + L_VEL + o), w = ld3(L + L_VEL + o + 3);
    st3(r + 7, v); st3(r + 10, w); st3(r, ld3(r) + dt * v);
    float wn = sqrtf(dot(w, w)), th = wn * dt; float dq[4];
    if (th > 1e-12f) { float sc = sinf(0.5f * th) / wn; dq[0] = w.x * sc; dq[1] = w.y * sc; dq[2] = w.z * sc; dq[3] = cosf(0.5f * th); }
    else { dq[0] = 0.5f * dt * w.x; dq[1] = 0.5f * dt * w.y; dq[2] = 0.5f * dt * w.z; dq[3] = 1.f; }
    const float ax = dq[0], ay = dq[1], az = dq[2], aw = dq[3], bx = r[3], by = r[4], bz = r[5], bw = r[6];
    float x = aw * bx + ax * bw + ay * bz - az * by, y = aw * by - ax * bz + ay * bw + az * bx;
    float z = aw * bz + ax * by - ay * bx + az * bw, w2 = aw * bw - ax * bx - ay * by - az * bz;
    float nn = 1.0f / sqrtf(x * x + y * y + z * z + w2 * w2);
    r[3] = x * nn; r[4] = y * nn; r[5] = z * nn; r[6] = w2 * nn;
  }
  wave_sync();
}
// FeedingEnv.update_targets (feeding.py:192-196): mouth = head pose o mouth offset.  Needs the link
// frames of a preceding kinematics(); the target is only consumed by the observation / reward code.
AGX_DEV void update_target(Ctx& c) {
  float* L = c.lds;
  wave_sync();
  if (c.lane == 0) {
    const int hl = TKI(c, AGX_T_HEAD_LINK), o = c.gender == 1 ? AGX_T_MOUTH_F : AGX_T_MOUTH_M;
    st3(L + L_ST + c.s_env + AGX_E_TARGET, mul(ldm3(L + L_LINKR + 9 * hl), mk3(TKF(c, o), TKF(c, o + 1), TKF(c, o + 2))) + ld3(L + L_LINKP + 3 * hl));
  }
  wave_sync();
}

// ---- state load / store ---------------------------------------------------------------------------------
AGX_DEV void load_env(Ctx& c, const float* gstate, int sw) {
  float* L = c.lds; const int lane = c.lane;
  for (int k = lane; k < sw; k += 64) L[L_ST + k] = gstate[k];
  wave_sync();
  c.gender = c.ldsi[L_ST + c.s_env + AGX_E_GENDER]; c.frozen = c.ldsi[L_ST + c.s_env + AGX_E_FROZEN];
  { const float ls = c.lds[L_ST + c.s_env + AGX_E_LIMIT_SCALE]; c.limit_scale = ls > 0.f ? ls : 1.f; }   // records written before v6 carry 0
  c.coop = TKI(c, AGX_T_COOP) == 1;
  if (lane == 0) { const float* r = L + L_ST + c.s_base; st3(L + L_BASE, ld3(r)); stm3(L + L_BASE + 3, quat_to_m3(r[3], r[4], r[5], r[6])); }
  if (lane < c.nhuman) { const float* r = L + L_ST + c.s_human + 7 * lane; float* h = L + L_HUMAN + 12 * lane; st3(h, ld3(r)); stm3(h + 3, quat_to_m3(r[3], r[4], r[5], r[6])); }
  if (lane < c.ndof) { int m = 0; for (int d = lane; d >= 0; d = RBI(c, d, AGX_R_PARENT)) m |= 1 << d; c.ldsi[L_MISC + M_ANC + lane] = m; }
  wave_sync();
}
AGX_DEV void store_env(Ctx& c, float* gstate, int sw) {
  wave_sync();
  for (int k = c.lane; k < sw; k += 64) gstate[k] = c.lds[L_ST + k];
}

// ---- task layer ------------------------------------------------------------------------------------------
AGX_DEV uint32_t rng_next(uint32_t& s0, uint32_t& s1) {
  uint64_t x = ((uint64_t)s1 << 32) | s0;
  x = x * 6364136223846793005ULL + 1442695040888963407ULL;
  s0 = (uint32_t)x; s1 = (uint32_t)(x >> 32);
  return (uint32_t)(x >> 33) ^ (uint32_t)(x >> 11);
}
AGX_DEV void tool_base_pose(const Ctx& c, v3& p, m3& R) {
  const float* L = c.lds; const int tb = c.bi[AGX_H_TOOL_BODY];
  m3 FR = ldm3(L + L_FREER + 9 * tb); v3 fp = ld3(L + L_ST + c.s_free + 13 * tb);
  p = mul(FR, mk3(FBF(c, tb, AGX_F_REFPOS), FBF(c, tb, AGX_F_REFPOS + 1), FBF(c, tb, AGX_F_REFPOS + 2))) + fp;
  R = mul(FR, quat_to_m3(FBF(c, tb, AGX_F_REFQUAT), FBF(c, tb, AGX_F_REFQUAT + 1), FBF(c, tb, AGX_F_REFQUAT + 2), FBF(c, tb, AGX_F_REFQUAT + 3)));
}
// FeedingEnv._get_obs (feeding.py:85-112), robot part; every lane computes, lane 0 writes
AGX_DEV void observe(const Ctx& c, float robot_force, float tool_force, float* gobs) {
  const float* L = c.lds;
  v3 bp = ld3(L + L_BASE); m3 BR = ldm3(L + L_BASE + 3);
  v3 sp; m3 sR; tool_base_pose(c, sp, sR);
  v3 spr = tmul(BR, sp - bp); q4 sq = m3_to_quat(mul_at(BR, sR));
  const int hl = TKI(c, AGX_T_HEAD_LINK);
  v3 hpr = tmul(BR, ld3(L + L_LINKP + 3 * hl) - bp); q4 hq = m3_to_quat(mul_at(BR, ldm3(L + L_LINKR + 9 * hl)));
  v3 tpr = tmul(BR, ld3(L + L_ST + c.s_env + AGX_E_TARGET) - bp);
  if (c.lane == 0) {
    int o = 0;
    gobs[o++] = spr.x; gobs[o++] = spr.y; gobs[o++] = spr.z;
    gobs[o++] = sq.x; gobs[o++] = sq.y; gobs[o++] = sq.z; gobs[o++] = sq.w;
    gobs[o++] = spr.x - tpr.x; gobs[o++] = spr.y - tpr.y; gobs[o++] = spr.z - tpr.z;
    for (int d = 0; d < c.nrobot; d++) if (RBI(c, d, AGX_R_ACT) >= 0) {
      float a = L[L_ST + c.s_q + d] + 3.14159265358979f;
      gobs[o++] = (a - 6.28318530717959f * floorf(a / 6.28318530717959f)) - 3.14159265358979f;
    }
    gobs[o++] = hpr.x; gobs[o++] = hpr.y; gobs[o++] = hpr.z;
    gobs[o++] = hq.x; gobs[o++] = hq.y; gobs[o++] = hq.z; gobs[o++] = hq.w;
    gobs[o++] = tool_force;
    if (c.coop) {   // human_obs (feeding.py:102-108): the same quantities in the frame of the human's base (collision body 0)
      const v3 hb = ld3(L + L_HUMAN); const m3 HR = ldm3(L + L_HUMAN + 3);
      const v3 sph = tmul(HR, sp - hb); const q4 sqh = m3_to_quat(mul_at(HR, sR));
      const v3 hph = tmul(HR, ld3(L + L_LINKP + 3 * hl) - hb); const q4 hqh = m3_to_quat(mul_at(HR, ldm3(L + L_LINKR + 9 * hl)));
      const v3 tph = tmul(HR, ld3(L + L_ST + c.s_env + AGX_E_TARGET) - hb);
      gobs[o++] = sph.x; gobs[o++] = sph.y; gobs[o++] = sph.z;
      gobs[o++] = sqh.x; gobs[o++] = sqh.y; gobs[o++] = sqh.z; gobs[o++] = sqh.w;
      gobs[o++] = sph.x - tph.x; gobs[o++] = sph.y - tph.y; gobs[o++] = sph.z - tph.z;
      for (int d = c.nrobot; d < c.ndof; d++) if (RBI(c, d, AGX_R_ACT) >= 0) gobs[o++] = L[L_ST + c.s_q + d];
      gobs[o++] = hph.x; gobs[o++] = hph.y; gobs[o++] = hph.z;
      gobs[o++] = hqh.x; gobs[o++] = hqh.y; gobs[o++] = hqh.z; gobs[o++] = hqh.w;
      gobs[o++] = robot_force; gobs[o++] = tool_force;
    }
  }
}

// ============================================================================================
// Kernel bodies.  One env.step() = frame_skip x [build, solve] + finish:
//   build  (register/LDS heavy, ~1/3 of the time): state -> kinematics, ABA + M^-1, predicted
//          velocities, collision, constraint rows -> per-env scratch record (rows, v*, contacts)
//   solve  (lean: ~64 VGPRs, 5 KB LDS -> many waves per SIMD): 50 PGS sweeps streaming the rows
//          from L2, integration, mouth-target update -> state
//   finish (once per step): forces, observation, food state machine, preferences, reward, done.
// ============================================================================================
struct Scratch { float* ent; float* hdr; float* vel; float* con; int* meta; };
AGX_DEV Scratch scratch_of(float* base) {
  Scratch s; s.ent = base + SCR_O_ENT; s.hdr = base + SCR_O_HDR; s.vel = base + SCR_O_VEL; s.con = base + SCR_O_CON; s.meta = (int*)(base + SCR_O_META);
  return s;
}

// build: `gaction` non-null on the first substep of an env.step() (take_step, env.py:174-222)
AGX_DEV void env_build(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gdebug, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  c.timing = gdebug != nullptr; c.dbg = gdebug;
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS];
  Scratch scr = scratch_of(gscratch);
  c.E = scr.ent; c.H = scr.hdr; c.gcon = scr.con;
  load_env(c, gstate, sw);
  if (gaction) {
    const int nsub = (int)PRM(c, AGX_P_FRAME_SKIP);
    // clip, scale, 5x accumulate against the joint limits -> motor targets (kept in the state record)
    const int iteration = Li[L_ST + c.s_env + AGX_E_ITERATION] + 1;       // env.py:185
    wave_sync();
    if (lane == 0) { Li[L_ST + c.s_env + AGX_E_ITERATION] = iteration; ((int*)gstate)[c.s_env + AGX_E_ITERATION] = iteration; }
    if (lane < c.ndof) {
      const int d = lane, ai = RBI(c, d, AGX_R_ACT);
      const bool is_human = d >= c.nrobot;
      const int k2 = is_human ? d - c.nrobot : 0;
      const float tsign = (iteration % 2 == 0) ? 1.f : -1.f;
      bool tremor_on = false;                         // impairment == 'tremor'
      for (int k = 0; k < c.nhdof; k++) if (L[L_ST + c.s_tremor + k] != 0.f) tremor_on = true;
      if (ai >= 0 && (!is_human || c.coop)) {
        // the limit test of take_step is discontinuous (an action that would cross a limit is zeroed,
        // env.py:206-211); it is evaluated in double like the reference's numpy code so that a joint
        // resting exactly on a limit takes the same branch
        const float a32 = fminf(fmaxf(gaction[ai], -1.f), 1.f) * PRM(c, AGX_P_ACTION_SCALE);
        double a = (double)a32, qa = (double)L[L_ST + c.s_q + d]; const double lo = (double)DLO(c, d), hi = (double)DHI(c, d);
        double tt = (double)L[L_ST + c.s_tremor + c.nhdof + k2];
        for (int k = 0; k < nsub; k++) {
          bool below = qa + a < lo, above = qa + a > hi;
          if (below || above) a = 0.0;
          if (below) qa = lo; if (above) qa = hi;
          if (is_human && tremor_on) { tt += a; qa = tt + (double)(L[L_ST + c.s_tremor + k2] * tsign); }   // env.py:212-215
          else qa += a;
        }
        L[L_ST + c.s_qt + d] = (float)qa; gstate[c.s_qt + d] = (float)qa;
        if (is_human && tremor_on) { L[L_ST + c.s_tremor + c.nhdof + k2] = (float)tt; gstate[c.s_tremor + c.nhdof + k2] = (float)tt; }
      }
      if (is_human && !c.coop) {   // tremor without control (env.py:212-215): target + tremors * (+1 on even iterations, -1 on odd)
        const float qt = L[L_ST + c.s_tremor + c.nhdof + k2] + L[L_ST + c.s_tremor + k2] * tsign;
        L[L_ST + c.s_qt + d] = qt; gstate[c.s_qt + d] = qt;
      }
    }
    wave_sync();
  }
  long long t0 = c.timing ? wave_clock() : 0, t1;
#define AGX_TICK(k) if (c.timing) { t1 = wave_clock(); c.tm[k] += t1 - t0; t0 = t1; }
  kinematics(c); AGX_TICK(0)
  aba_and_minv(c); AGX_TICK(1)
  predict_velocities(c); AGX_TICK(2)
  collide(c); AGX_TICK(3)
  build_rows(c); AGX_TICK(4)
#undef AGX_TICK
  // hand-over to the solve kernel
  for (int k = lane; k < SCR_VEL; k += 64) scr.vel[k] = L[L_VEL + k];
  if (lane == 0) { scr.meta[META_NCON] = c.ncon; scr.meta[META_NROWS] = c.nrows; scr.meta[META_NNC] = c.first_normal; scr.meta[META_NEAR] = c.near_mask; scr.meta[META_OVERFLOW] = c.overflow; scr.meta[META_NENT] = c.nent; }
  if (gdebug) {   // first-substep internals for the parity tests and the phase cycle counters
    if (lane == 0) { gdebug[0] = (float)c.ncon; gdebug[1] = (float)c.nrows; gdebug[2] = (float)c.overflow; gdebug[3] = (float)c.first_normal; }   // [4..4+ndof) = qdd
    for (int q = lane; q < MAX_CON * CON_STRIDE; q += 64) gdebug[16 + q] = scr.con[q];
    for (int q = lane; q < MAX_DOF * MAX_DOF; q += 64) gdebug[16 + MAX_CON * CON_STRIDE + q] = L[L_MINV + q];
    wave_sync();
    for (int q = lane; q < MAX_ROWS * HDR_STRIDE; q += 64) gdebug[DBG_HDR + q] = scr.hdr[q];
    if (lane == 0) { for (int k = 0; k < 16; k++) if (k != 5 && k != 6 && k != 7) gdebug[DBG_TIME + k] = (float)c.tm[k]; }
  }
}

// solve: PGS + integration + post-substep hooks of one p.stepSimulation() (env.py:226-232)
AGX_DEV void env_solve(const uint32_t* blob, float* gstate, float* gscratch, float* gdebug, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  const int sw = c.bi[AGX_H_STATE_WORDS];
  Scratch scr = scratch_of(gscratch);
  c.E = scr.ent; c.H = scr.hdr; c.gcon = scr.con;
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC]; c.nent = scr.meta[META_NENT];
  // state copy only (the frame tables of load_env are not needed here and their LDS is the row window)
  for (int k = lane; k < sw; k += 64) lds[L_ST + k] = gstate[k];
  { const int np = c.nent < SOLVE_LDS_PAIRS ? c.nent : SOLVE_LDS_PAIRS;
    const f2* src = (const f2*)scr.ent; f2* dst = (f2*)(lds + L_SOLVE_ENT);
    for (int k = lane; k < np; k += 64) dst[k] = src[k]; }
  wave_sync();
  c.gender = c.ldsi[L_ST + c.s_env + AGX_E_GENDER]; c.frozen = c.ldsi[L_ST + c.s_env + AGX_E_FROZEN];
  { const float ls = c.lds[L_ST + c.s_env + AGX_E_LIMIT_SCALE]; c.limit_scale = ls > 0.f ? ls : 1.f; }   // records written before v6 carry 0
  c.coop = TKI(c, AGX_T_COOP) == 1;
  const long long t0 = gdebug ? wave_clock() : 0;
  float dv0, dv1;
  pgs(c, dv0, dv1);
  const long long t1 = gdebug ? wave_clock() : 0;
  integrate(c, scr.vel, dv0, dv1);
  store_env(c, gstate, sw);
  if (gdebug && lane == 0) { gdebug[DBG_TIME + 5] = (float)(t1 - t0); gdebug[DBG_TIME + 6] = (float)(wave_clock() - t1); }
}

AGX_DEV void env_observe(const uint32_t* blob, float* gstate, float* gobs, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  load_env(c, gstate, c.bi[AGX_H_STATE_WORDS]);
  kinematics(c); update_target(c); observe(c, 0.f, 0.f, gobs);
}

// finish: everything FeedingEnv.step does after take_step (feeding.py:17-43)
AGX_DEV void env_finish(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                        float* ginfo, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS], act_dim = c.bi[AGX_H_ACT_DIM];
  Scratch scr = scratch_of(gscratch);
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC]; c.near_mask = scr.meta[META_NEAR];
  load_env(c, gstate, sw);
  float an2 = 0.f;
  for (int k = 0; k < act_dim; k++) an2 += gaction[k] * gaction[k];
  wave_sync();
  kinematics(c);   // poses as the getters of _get_obs see them after the last stepSimulation
  update_target(c);
  // get_total_force (feeding.py:45-48) from the last substep's contact impulses
  float rf = 0.f, tf = 0.f;
  if (lane < c.ncon) {
    const float* k = scr.con + CON_STRIDE * lane; const int* ki = (const int*)k;
    int ta = CLI(c, ki[C_CA], AGX_C_TAG), tb = CLI(c, ki[C_CB], AGX_C_TAG);
    if (ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN) {
      int other = ta == AGX_TAG_HUMAN ? tb : ta; float f = k[C_LAM] / c.dt;
      if (other == AGX_TAG_ROBOT) rf = f;
      if (other == AGX_TAG_TOOL) tf = f;
    }
  }
  const float robot_f = wave_sum(rf), tool_f = wave_sum(tf), total_f = robot_f + tool_f;
  observe(c, robot_f, tool_f, gobs);
  // get_food_rewards (feeding.py:50-83)
  float food_reward = 0.f, food_hit = 0.f, vel_sum = 0.f;
  int alive = Li[L_ST + c.s_env + AGX_E_FOOD_ALIVE], active = Li[L_ST + c.s_env + AGX_E_FOOD_ACTIVE];
  int success = Li[L_ST + c.s_env + AGX_E_TASK_SUCCESS];
  uint32_t r0 = (uint32_t)Li[L_ST + c.s_env + AGX_E_RNG], r1 = (uint32_t)Li[L_ST + c.s_env + AGX_E_RNG + 1];
  const int active_on_entry = active, hit_mask = c.near_mask, food0 = c.bi[AGX_H_FOOD0];
  const v3 target = ld3(L + L_ST + c.s_env + AGX_E_TARGET);
  // world AABBs of the tool colliders + particles for the 0.1 m closest-point query (agent.py:118-130)
  {
    float* AB = L + L_ARENA;
    for (int col = lane; col < c.ncoll; col += 64) {
      int tag = CLI(c, col, AGX_C_TAG);
      if (tag != AGX_TAG_TOOL && tag != AGX_TAG_FOOD) continue;
      m3 R; v3 p; body_xf(c, CLI(c, col, AGX_C_BODY), R, p);
      v3 cl = mk3(CLF(c, col, AGX_C_AABB_C), CLF(c, col, AGX_C_AABB_C + 1), CLF(c, col, AGX_C_AABB_C + 2));
      v3 hl = mk3(CLF(c, col, AGX_C_AABB_H), CLF(c, col, AGX_C_AABB_H + 1), CLF(c, col, AGX_C_AABB_H + 2));
      v3 cw = mul(R, cl) + p; float r = CLF(c, col, AGX_C_RADIUS);
      for (int k = 0; k < 3; k++) {
        float hh = fabsf(R.a[3 * k]) * hl.x + fabsf(R.a[3 * k + 1]) * hl.y + fabsf(R.a[3 * k + 2]) * hl.z + r;
        AB[ABS * col + k] = comp(cw, k) - hh; AB[ABS * col + 3 + k] = comp(cw, k) + hh;
      }
    }
    wave_sync();
  }
  int tool0 = -1, tool1 = -1, foodc0 = -1;
  for (int g = 0; g < c.ngroup; g++) {   // the (food, tool) group carries both collider ranges
    int a0 = GRI(c, g, AGX_G_A0), b0 = GRI(c, g, AGX_G_B0);
    if (CLI(c, a0, AGX_C_TAG) == AGX_TAG_FOOD && CLI(c, b0, AGX_C_TAG) == AGX_TAG_TOOL) { foodc0 = a0; tool0 = b0; tool1 = GRI(c, g, AGX_G_B1); break; }
  }
  const float spill = TKF(c, AGX_T_SPILL_DIST);
  for (int k = 0; k < c.nfood; k++) {
    if (!(alive >> k & 1)) continue;
    const int b = food0 + k; float* r = L + L_ST + c.s_free + 13 * b;
    v3 d = target - ld3(r);
    if (sqrtf(dot(d, d)) < TKF(c, AGX_T_MOUTH_DIST)) {
      food_reward += 20.f; success += 1; vel_sum += sqrtf(dot(ld3(r + 7), ld3(r + 7)));
      alive &= ~(1 << k); active &= ~(1 << k);
      float px = 1000.0f + 1000.0f * (float)(rng_next(r0, r1) >> 8) * (1.0f / 16777216.0f);
      float py = 1000.0f + 1000.0f * (float)(rng_next(r0, r1) >> 8) * (1.0f / 16777216.0f);
      float pz = 1000.0f + 1000.0f * (float)(rng_next(r0, r1) >> 8) * (1.0f / 16777216.0f);
      wave_sync();
      if (lane == 0) { r[0] = px; r[1] = py; r[2] = pz; r[3] = 0.f; r[4] = 0.f; r[5] = 0.f; r[6] = 1.f; }
      wave_sync();
      continue;
    }
    bool near = false;
    {
      const int fc = foodc0 + k; const float* AB = L + L_ARENA;
      for (int base = tool0; base < tool1; base += 64) {
        const int tc = base + lane; bool hitl = false;
        if (tc < tool1) {
          bool sep = false;
          for (int q = 0; q < 3; q++) if (AB[ABS * fc + q] > AB[ABS * tc + 3 + q] + spill || AB[ABS * tc + q] > AB[ABS * fc + 3 + q] + spill) sep = true;
          Cand tmp; if (!sep) hitl = narrowphase(c, fc, tc, spill, tmp);
        }
        if (wave_any(hitl)) near = true;
      }
    }
    if (!near) { food_reward -= 5.f; alive &= ~(1 << k); }
  }
  for (int k = 0; k < c.nfood; k++) if ((active_on_entry >> k & 1) && (hit_mask >> k & 1)) { food_hit -= 1.f; active &= ~(1 << k); }
  // end-effector speed (feeding.py:22), human_preferences (env.py:237-274, feeding branch), reward
  const float* A = L + L_ARENA; (void)A;
  float ee_speed;
  {
    const int ee = TKI(c, AGX_T_EE_LINK);
    float sv[6] = {0, 0, 0, 0, 0, 0};
    for (int d = ee; d >= 0; d = RBI(c, d, AGX_R_PARENT)) { float qd = L[L_ST + c.s_qd + d]; for (int j = 0; j < 6; j++) sv[j] += L[L_S + 6 * d + j] * qd; }
    v3 xr = ld3(L + L_MISC + M_EEP) - ld3(L + L_MISC + M_REF);
    v3 v = mk3(sv[3], sv[4], sv[5]) + cross(mk3(sv[0], sv[1], sv[2]), xr);
    ee_speed = sqrtf(dot(v, v));
  }
  float pref = TKF(c, AGX_T_C_V) * (-ee_speed) + TKF(c, AGX_T_C_F) * (-total_f) + TKF(c, AGX_T_C_HF) * (tool_f < 10.f ? 0.f : -tool_f)
             + TKF(c, AGX_T_C_FD) * food_hit + TKF(c, AGX_T_C_FDV) * (-vel_sum);
  v3 sp; m3 sR; tool_base_pose(c, sp, sR);
  v3 dd = target - sp;
  float reward = TKF(c, AGX_T_W_DISTANCE) * (-sqrtf(dot(dd, dd))) + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + TKF(c, AGX_T_W_FOOD) * food_reward + pref;
  const int iteration = Li[L_ST + c.s_env + AGX_E_ITERATION];
  wave_sync();
  if (lane == 0) {
    Li[L_ST + c.s_env + AGX_E_FOOD_ALIVE] = alive; Li[L_ST + c.s_env + AGX_E_FOOD_ACTIVE] = active;
    Li[L_ST + c.s_env + AGX_E_TASK_SUCCESS] = success; Li[L_ST + c.s_env + AGX_E_RNG] = (int)r0; Li[L_ST + c.s_env + AGX_E_RNG + 1] = (int)r1;
    *greward = reward;
    *gdone = (uint8_t)(iteration >= (int)TKF(c, AGX_T_EPISODE_LEN));
    if (ginfo) {
      ginfo[AGX_INFO_TOTAL_FORCE] = total_f;
      ginfo[AGX_INFO_TASK_SUCCESS] = (float)(success >= Li[L_ST + c.s_env + AGX_E_TOTAL_FOOD] * TKF(c, AGX_T_SUCCESS_FRAC));
      ginfo[AGX_INFO_ROBOT_FORCE] = robot_f; ginfo[AGX_INFO_TOOL_FORCE] = tool_f; ginfo[AGX_INFO_FOOD_REWARD] = food_reward;
      ginfo[AGX_INFO_PREF] = pref; ginfo[AGX_INFO_NCONTACT] = (float)c.ncon; ginfo[AGX_INFO_NROWS] = (float)c.nrows;
    }
  }
  store_env(c, gstate, sw);
}

}  // namespace agx
