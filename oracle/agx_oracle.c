/* agx_oracle.c -- CPU oracle, TEST INFRASTRUCTURE ONLY (see agx_oracle.h; PARITY UNPINNED).
 *
 * Plain C, double precision, one environment at a time, no SIMD, no threads.  Every function
 * names the reference lines it follows; the physics inside p.stepSimulation() (env.py:226) is a
 * restatement of the published algorithms Bullet's multibody pipeline is built from:
 *   - forward kinematics + Featherstone articulated-body algorithm (world-frame spatial algebra)
 *   - convex narrowphase by GJK on (vertex core + radius) shapes, 42-direction penetration sampling
 *   - velocity-level constraint rows (joint motors, joint limits, 6-row fixed constraint, contact
 *     normal + one friction direction) solved by projected Gauss-Seidel, fixed sweep count
 *   - semi-implicit Euler integration.
 */
#include "agx_oracle.h"
#include "../include/agx_blob.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXDOF 64
#define MAXFREE 16
#define MAXHUMAN 24
#define NVMAX (MAXDOF + 6 * MAXFREE)
/* limits; the PLAIN build of the approximation study (oracle/Makefile `plain`, tests/diag/approximation_budget.py) raises them: full hulls,
 * every candidate contact kept, every contact inside the break distance solved */
#ifndef MAXC
#define MAXC 96
#endif
#ifndef MAXROWS
#define MAXROWS 320
#endif
#ifndef MAXV
#define MAXV 80 /* max vertices of one collider core */
#endif
#ifndef MAXCAND
#define MAXCAND 128 /* candidate contacts of one collider against one pair group */
#endif
#define MAXCOLL 512
#define MAXQPT 16
#ifndef MAXMAN
#define MAXMAN 192
#endif
/* narrowphase calls whose CORES overlapped since the last agxo_stat_core_overlaps(): the 42-direction penetration sampling ran instead of
 * the exact GJK distance (how often does the approximate depth matter at all?) */
static long g_stat_core_overlap = 0, g_stat_narrowphase = 0;
void agxo_stat_core_overlaps(long* out2) { out2[0] = g_stat_core_overlap; out2[1] = g_stat_narrowphase; g_stat_core_overlap = 0; g_stat_narrowphase = 0; }

typedef struct { double p[3]; double R[9]; } xf_t;

struct agxo_model {
  uint32_t* w; const float* f; const int32_t* i; size_t nwords;
  int ndof, nfree, nhuman, ncoll, ngroup, nfood, act_dim, obs_dim, state_words, food0, tool_body, ndir;
  int o_params, o_robot, o_free, o_coll, o_vert, o_group, o_task, o_dirs;
  int s_q, s_qd, s_qt, s_free, s_base, s_human, s_env, s_tremor;
  int nrobot, nhdof;
  int task_kind, s_task, o_targets;
  int o_cloth, sim_sub;      /* cloth section (0 = none); internal substeps per stepSimulation (numSubSteps) */
  double dt;                 /* length of one internal substep: DT / sim_sub */
};

/* ------------------------------------------------------------------------------------ math */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void add3(const double* a, const double* b, double* o) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static void axpy3(double s, const double* a, double* o) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static void mv3(const double* R, const double* v, double* o) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
         z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mtv3(const double* R, const double* v, double* o) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
         z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mm3(const double* A, const double* B, double* O) {
  double t[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
  memcpy(O, t, sizeof t);
}
static void mmt3(const double* A, const double* B, double* O) { /* A * B^T */
  double t[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
  memcpy(O, t, sizeof t);
}
static void quat_to_mat(const double* q, double* R) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void mat_to_quat(const double* R, double* q) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) { double s = sqrt(t + 1.0) * 2; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; q[3] = 0.25 * s; }
  else if (R[0] > R[4] && R[0] > R[8]) { double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; q[3] = (R[7] - R[5]) / s; }
  else if (R[4] > R[8]) { double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; q[3] = (R[2] - R[6]) / s; }
  else { double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; q[3] = (R[3] - R[1]) / s; }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; k++) q[k] /= n;
}
static void quat_mul(const double* a, const double* b, double* o) {
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void axis_angle_mat(const double* a, double th, double* R) {
  double c = cos(th), s = sin(th), t = 1 - c, x = a[0], y = a[1], z = a[2];
  R[0] = t * x * x + c; R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
  R[3] = t * x * y + s * z; R[4] = t * y * y + c; R[5] = t * y * z - s * x;
  R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}
static void xf_apply(const xf_t* X, const double* v, double* o) { double t[3]; mv3(X->R, v, t); add3(t, X->p, o); }

/* spatial vectors: [angular(3); linear(3)] referred to the world origin */
static double dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
static void crm(const double* v, const double* m, double* o) { /* motion x motion */
  double a[3], b[3], c[3];
  cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(const double* v, const double* f, double* o) { /* motion x* force */
  double a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
static void mv6(const double* I, const double* v, double* o) {
  double t[6];
  for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += I[6 * r + c] * v[c]; t[r] = s; }
  memcpy(o, t, sizeof t);
}
/* spatial inertia about the world origin from mass, world COM c, world-axes inertia Ic about the COM */
static void spatial_inertia(double m, const double* c, const double* Ic, double* I) {
  double cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
  double cc = dot3(c, c);
  for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) {
    I[6 * r + k] = Ic[3 * r + k] + m * ((r == k ? cc : 0.0) - c[r] * c[k]);
    I[6 * r + 3 + k] = m * cx[3 * r + k];
    I[6 * (r + 3) + k] = m * cx[3 * k + r];
    I[6 * (r + 3) + 3 + k] = (r == k) ? m : 0.0;
  }
}

/* ------------------------------------------------------------------------------------ model */
agxo_model* agxo_load(const uint32_t* blob, size_t nwords) {
  if (nwords < AGX_H_COUNT || blob[AGX_H_MAGIC] != AGX_BLOB_MAGIC || blob[AGX_H_VERSION] != AGX_BLOB_VERSION) return NULL;
  agxo_model* m = (agxo_model*)calloc(1, sizeof *m);
  m->w = (uint32_t*)malloc(nwords * 4);
  memcpy(m->w, blob, nwords * 4);
  m->f = (const float*)m->w; m->i = (const int32_t*)m->w; m->nwords = nwords;
  const int32_t* h = m->i;
  m->ndof = h[AGX_H_NDOF]; m->nfree = h[AGX_H_NFREE]; m->nhuman = h[AGX_H_NHUMAN]; m->ncoll = h[AGX_H_NCOLL];
  m->ngroup = h[AGX_H_NGROUP]; m->nfood = h[AGX_H_NFOOD]; m->act_dim = h[AGX_H_ACT_DIM]; m->obs_dim = h[AGX_H_OBS_DIM];
  m->state_words = h[AGX_H_STATE_WORDS]; m->food0 = h[AGX_H_FOOD0]; m->tool_body = h[AGX_H_TOOL_BODY]; m->ndir = h[AGX_H_NDIR];
  m->o_params = h[AGX_H_OFF_PARAMS]; m->o_robot = h[AGX_H_OFF_ROBOT]; m->o_free = h[AGX_H_OFF_FREE]; m->o_coll = h[AGX_H_OFF_COLL];
  m->o_vert = h[AGX_H_OFF_VERT]; m->o_group = h[AGX_H_OFF_GROUP]; m->o_task = h[AGX_H_OFF_TASK]; m->o_dirs = h[AGX_H_OFF_DIRS];
  m->s_q = h[AGX_H_S_Q]; m->s_qd = h[AGX_H_S_QD]; m->s_qt = h[AGX_H_S_QT]; m->s_free = h[AGX_H_S_FREE]; m->s_base = h[AGX_H_S_BASE];
  m->s_human = h[AGX_H_S_HUMAN]; m->s_env = h[AGX_H_S_ENV]; m->s_tremor = h[AGX_H_S_TREMOR];
  m->nrobot = h[AGX_H_NROBOT]; m->nhdof = h[AGX_H_NHDOF];
  m->task_kind = h[AGX_H_TASK_KIND]; m->s_task = h[AGX_H_S_TASK]; m->o_targets = h[AGX_H_OFF_TARGETS];
  m->o_cloth = h[AGX_H_OFF_CLOTH]; m->sim_sub = h[AGX_H_SIM_SUBSTEPS] > 1 ? h[AGX_H_SIM_SUBSTEPS] : 1;
  m->dt = (double)m->f[m->o_params + AGX_P_DT] / m->sim_sub;
  if (m->ndof > MAXDOF || m->nfree > MAXFREE || m->nhuman > MAXHUMAN || m->ncoll > MAXCOLL) { agxo_free(m); return NULL; }
  return m;
}
void agxo_free(agxo_model* m) { if (m) { free(m->w); free(m); } }
int agxo_state_words(const agxo_model* m) { return m->state_words; }
int agxo_ndof(const agxo_model* m) { return m->ndof; }

#define PARAM(m, k) ((double)(m)->f[(m)->o_params + (k)])
/* DoF d made static for this environment: a 32-bit mask, DoFs 32.. are never frozen */
#define FROZEN(s, d) ((d) < 32 && (((s)->frozen >> (d)) & 1))
/* link record of DoF d: human DoFs have one record per gender (agx_blob.h AGX_H_NHDOF) */
static int g_gender = 0; /* gender of the environment currently being stepped (the oracle is single threaded per process; see sim_load) */
#define REC(m, d) ((d) < (m)->nrobot ? (d) : (d) + g_gender * (m)->nhdof)
#define RF(m, d, k) ((double)(m)->f[(m)->o_robot + REC(m, d) * AGX_R_STRIDE + (k)])
#define RI(m, d, k) ((m)->i[(m)->o_robot + REC(m, d) * AGX_R_STRIDE + (k)])
#define FF(m, b, k) ((double)(m)->f[(m)->o_free + (b) * AGX_F_STRIDE + (k)])
#define FI(m, b, k) ((m)->i[(m)->o_free + (b) * AGX_F_STRIDE + (k)])
#define CF(m, c, k) ((double)(m)->f[(m)->o_coll + (c) * AGX_C_STRIDE + (k)])
#define CI(m, c, k) ((m)->i[(m)->o_coll + (c) * AGX_C_STRIDE + (k)])
#define GI(m, g, k) ((m)->i[(m)->o_group + (g) * AGX_G_STRIDE + (k)])
#define TF(m, k) ((double)(m)->f[(m)->o_task + (k)])
#define TI(m, k) ((m)->i[(m)->o_task + (k)])

/* ------------------------------------------------------------------------------------ sim state */
typedef struct {
  int ca, cb;          /* collider indices */
  int ba, bb;          /* body codes */
  double pa[3], pb[3], n[3], dist, mu;
  double lambda_n;     /* solved normal impulse */
  double t1[3];        /* first friction direction of the substep's row (build_rows) */
} contact_t;

typedef struct {
  double J[NVMAX], B[NVMAX];
  double invD, b, lo, hi, lambda;
  int fric_of;         /* >=0: friction row bound by mu * lambda of that row */
  double mu;
} row_t;

typedef struct {
  const agxo_model* m;
  int ndof, nfree, nv;
  double q[MAXDOF], qd[MAXDOF], qt[MAXDOF];
  double fpos[MAXFREE][3], fquat[MAXFREE][4], fv[MAXFREE][3], fw[MAXFREE][3];
  xf_t base, human[MAXHUMAN];
  double plane_mu, target[3];
  int gender, alive, active, iteration, success, total_food, frozen;
  double tremor[MAXDOF], tremor_target[MAXDOF];
  double limit_scale;    /* scale of the human joint limits (impairment 'limits') */
  double human_kp, human_maxf;   /* per-env motor gain / force of the human's joints, 0 = the blob's (AGX_E_HUMAN_KP) */
  int coop;              /* the human is controllable (TASK.COOP) */
  int human_agent;       /* the human is in env.agents: controllable or tremor (env.py:130-131).  Only then does take_step's loop reach
                          * Agent.enforce_joint_limits / enforce_realistic_joint_limits (env.py:227-231): a human whose arm is dynamic only
                          * because of a reactive hold (scratch itch, arm manipulation, dressing; human.py:108,124-127) is never limit-reset */
  int settling;          /* reset-time stepSimulation loops (feeding.py:178-179, bed_bathing.py:130-131, arm_manipulation.py:145-146,
                          * dressing.py:190-193) are plain engine steps: no hooks */
  uint32_t rng[2];
  /* derived, per substep */
  xf_t link[MAXDOF], freex[MAXFREE];
  double comw[MAXDOF][3], Iw[MAXDOF][9], S[MAXDOF][6], vsp[MAXDOF][6], cvp[MAXDOF][6];
  double I6[MAXDOF][36], IA[MAXDOF][36], U[MAXDOF][6], Dinv[MAXDOF];
  double Minv[MAXDOF * MAXDOF];
  double fIinv[MAXFREE][9];
  double vel[NVMAX];
  contact_t con[MAXC]; int ncon;
  int food_near_human;   /* particles with a (food, human) manifold point: separation < CONTACT_BREAK */
  double qpt[MAXQPT][3]; int qpt_link[MAXQPT]; int nqpt;   /* bed bathing: manifold points of the wiping pad on the human (bed_bathing.py:47-58) */
  uint32_t bb_alive[AGX_BB_ALIVE_WORDS];                  /* bed bathing: targets not wiped yet */
  double arm_prev[4]; int arm_has_prev;                    /* arm_previous_valid_pose (human.py:147-149) */
  double si_target[3], si_prev[3]; int si_limb;            /* scratch itch: target_on_arm, prev_target_contact_pos, limb (scratch_itch.py:134-146,96) */
  int contact_overflow;
  contact_t man[MAXMAN]; int nman;                         /* every narrowphase hit (separation < CONTACT_BREAK) of the groups whose manifold the task reads (flag bit 1), in collide order: what getContactPoints lists for them, with or without force (world API below) */
  row_t* rows; int nrows;
  /* dressing: the cloth (node positions / velocities live with the caller), its attachment point and the contacts of the last substep */
  double dr_gravity, dr_force_sum, dr_best;
  uint64_t dk_alive, dk_active;                            /* drinking: the water particles still in the scene / that have not hit the person yet */
  int* ccon_node; int* ccon_shape;                         /* node and shape-table entry of every contact of ccon */
  double am_best;                                          /* arm manipulation: task_success = best reward_distance_human so far (arm_manipulation.py:48-49) */
  double* cx; double* cv; double* cq;                      /* [NN][3] each; NULL = no cloth attached to this call */
  double anchor[3]; int anchor_set;
  double* ccon; int nccon;                                 /* {x, y, z, fx, fy, fz} per node-vs-rigid contact */
} sim_t;

static void sim_load(sim_t* s, const agxo_model* m, const float* st) {
  memset(s, 0, sizeof *s);
  s->m = m; s->ndof = m->ndof; s->nfree = m->nfree; s->nv = m->ndof + 6 * m->nfree;
  for (int d = 0; d < m->ndof; d++) { s->q[d] = st[m->s_q + d]; s->qd[d] = st[m->s_qd + d]; s->qt[d] = st[m->s_qt + d]; }
  for (int b = 0; b < m->nfree; b++) {
    const float* r = st + m->s_free + 13 * b;
    for (int k = 0; k < 3; k++) { s->fpos[b][k] = r[k]; s->fv[b][k] = r[7 + k]; s->fw[b][k] = r[10 + k]; }
    for (int k = 0; k < 4; k++) s->fquat[b][k] = r[3 + k];
  }
  { const float* r = st + m->s_base; double qd[4] = {r[3], r[4], r[5], r[6]};
    for (int k = 0; k < 3; k++) s->base.p[k] = r[k]; quat_to_mat(qd, s->base.R); }
  for (int h = 0; h < m->nhuman; h++) {
    const float* r = st + m->s_human + 7 * h; double qd[4] = {r[3], r[4], r[5], r[6]};
    for (int k = 0; k < 3; k++) s->human[h].p[k] = r[k]; quat_to_mat(qd, s->human[h].R);
  }
  const float* e = st + m->s_env; const int32_t* ei = (const int32_t*)e;
  s->plane_mu = e[AGX_E_PLANE_FRICTION]; s->gender = ei[AGX_E_GENDER]; g_gender = s->gender;
  s->frozen = ei[AGX_E_FROZEN];
  s->limit_scale = e[AGX_E_LIMIT_SCALE] > 0 ? e[AGX_E_LIMIT_SCALE] : 1.0;   /* records written before v6 carry 0 */
  s->coop = TI(m, AGX_T_COOP) == 1;
  s->human_agent = s->coop;
  for (int k = 0; k < m->nhdof; k++) if (st[m->s_tremor + k] != 0) s->human_agent = 1;          /* impairment == 'tremor' */
  s->human_kp = e[AGX_E_HUMAN_KP]; s->human_maxf = e[AGX_E_HUMAN_MAXF];
  for (int k = 0; k < m->nhdof; k++) { s->tremor[k] = st[m->s_tremor + k]; s->tremor_target[k] = st[m->s_tremor + m->nhdof + k]; }
  for (int k = 0; k < 3; k++) s->target[k] = e[AGX_E_TARGET + k];
  s->alive = ei[AGX_E_FOOD_ALIVE]; s->active = ei[AGX_E_FOOD_ACTIVE]; s->iteration = ei[AGX_E_ITERATION];
  s->success = ei[AGX_E_TASK_SUCCESS]; s->total_food = ei[AGX_E_TOTAL_FOOD];
  s->rng[0] = (uint32_t)ei[AGX_E_RNG]; s->rng[1] = (uint32_t)ei[AGX_E_RNG + 1];
  if (m->task_kind != AGX_TASK_FEEDING) {
    if (m->task_kind == AGX_TASK_BED_BATHING) for (int k = 0; k < AGX_BB_ALIVE_WORDS; k++) s->bb_alive[k] = (uint32_t)((const int32_t*)st)[m->s_task + AGX_BB_ALIVE + k];
    for (int k = 0; k < 4; k++) s->arm_prev[k] = st[m->s_task + AGX_BB_PREV + k];
    s->arm_has_prev = ((const int32_t*)st)[m->s_task + AGX_BB_HAS_PREV];
    if (m->task_kind == AGX_TASK_SCRATCH_ITCH) {
      for (int k = 0; k < 3; k++) { s->si_target[k] = st[m->s_task + AGX_SI_TARGET + k]; s->si_prev[k] = st[m->s_task + AGX_SI_PREV_CONTACT + k]; }
      s->si_limb = ((const int32_t*)st)[m->s_task + AGX_SI_LIMB];
    }
    if (m->task_kind == AGX_TASK_ARM_MANIPULATION) s->am_best = st[m->s_task + AGX_AM_BEST];
    if (m->task_kind == AGX_TASK_DRESSING) { s->dr_gravity = st[m->s_task + AGX_DR_CLOTH_GRAVITY]; s->dr_force_sum = st[m->s_task + AGX_DR_FORCE_SUM]; s->dr_best = st[m->s_task + AGX_DR_BEST]; }
    if (m->task_kind == AGX_TASK_DRINKING) {
      const uint32_t* u = (const uint32_t*)st + m->s_task;
      s->dk_alive = (uint64_t)u[AGX_DK_ALIVE] | ((uint64_t)u[AGX_DK_ALIVE + 1] << 32); s->dk_active = (uint64_t)u[AGX_DK_ACTIVE] | ((uint64_t)u[AGX_DK_ACTIVE + 1] << 32);
      s->dr_gravity = PARAM(m, AGX_P_GRAVITY_Z);           /* the water falls under the world's gravity (env.py:104) */
    }
  }
}
static void sim_store(const sim_t* s, float* st) {
  const agxo_model* m = s->m;
  for (int d = 0; d < m->ndof; d++) { st[m->s_q + d] = (float)s->q[d]; st[m->s_qd + d] = (float)s->qd[d]; st[m->s_qt + d] = (float)s->qt[d]; }
  for (int b = 0; b < m->nfree; b++) {
    float* r = st + m->s_free + 13 * b;
    for (int k = 0; k < 3; k++) { r[k] = (float)s->fpos[b][k]; r[7 + k] = (float)s->fv[b][k]; r[10 + k] = (float)s->fw[b][k]; }
    for (int k = 0; k < 4; k++) r[3 + k] = (float)s->fquat[b][k];
  }
  for (int k = 0; k < m->nhdof; k++) st[m->s_tremor + m->nhdof + k] = (float)s->tremor_target[k];
  float* e = st + m->s_env; int32_t* ei = (int32_t*)e;
  for (int k = 0; k < 3; k++) e[AGX_E_TARGET + k] = (float)s->target[k];
  ei[AGX_E_FOOD_ALIVE] = s->alive; ei[AGX_E_FOOD_ACTIVE] = s->active; ei[AGX_E_ITERATION] = s->iteration;
  ei[AGX_E_TASK_SUCCESS] = s->success; ei[AGX_E_RNG] = (int32_t)s->rng[0]; ei[AGX_E_RNG + 1] = (int32_t)s->rng[1];
  if (m->task_kind != AGX_TASK_FEEDING) {
    if (m->task_kind == AGX_TASK_BED_BATHING) for (int k = 0; k < AGX_BB_ALIVE_WORDS; k++) ((int32_t*)st)[m->s_task + AGX_BB_ALIVE + k] = (int32_t)s->bb_alive[k];
    for (int k = 0; k < 4; k++) st[m->s_task + AGX_BB_PREV + k] = (float)s->arm_prev[k];
    ((int32_t*)st)[m->s_task + AGX_BB_HAS_PREV] = s->arm_has_prev;
    if (m->task_kind == AGX_TASK_SCRATCH_ITCH) for (int k = 0; k < 3; k++) st[m->s_task + AGX_SI_PREV_CONTACT + k] = (float)s->si_prev[k];
    if (m->task_kind == AGX_TASK_ARM_MANIPULATION) st[m->s_task + AGX_AM_BEST] = (float)s->am_best;
    if (m->task_kind == AGX_TASK_DRESSING) { st[m->s_task + AGX_DR_FORCE_SUM] = (float)s->dr_force_sum; st[m->s_task + AGX_DR_BEST] = (float)s->dr_best; }
    if (m->task_kind == AGX_TASK_DRINKING) {
      uint32_t* u = (uint32_t*)st + m->s_task;
      u[AGX_DK_ALIVE] = (uint32_t)s->dk_alive; u[AGX_DK_ALIVE + 1] = (uint32_t)(s->dk_alive >> 32); u[AGX_DK_ACTIVE] = (uint32_t)s->dk_active; u[AGX_DK_ACTIVE + 1] = (uint32_t)(s->dk_active >> 32);
    }
  }
}

/* joint limits of DoF d; the human's are scaled per environment (human_creation.py:199-200) */
static double dof_lower(const sim_t* s, int d) { return RF(s->m, d, AGX_R_LOWER) * ((RI(s->m, d, AGX_R_KIND) & 3) == 1 ? s->limit_scale : 1.0); }
static double dof_upper(const sim_t* s, int d) { return RF(s->m, d, AGX_R_UPPER) * ((RI(s->m, d, AGX_R_KIND) & 3) == 1 ? s->limit_scale : 1.0); }

/* ------------------------------------------------------------------------------------ kinematics
 * K1 of SURVEY 2.2: what getLinkState(computeForwardKinematics=True) (agent.py:52) reads back. */
static void kinematics(sim_t* s) {
  const agxo_model* m = s->m;
  for (int d = 0; d < s->ndof; d++) {
    int par = RI(m, d, AGX_R_PARENT);
    const xf_t* P = par == AGX_PARENT_HUMAN_BASE ? &s->human[0] : (par < 0 ? &s->base : &s->link[par]);
    double tp[3] = {RF(m, d, AGX_R_TPOS), RF(m, d, AGX_R_TPOS + 1), RF(m, d, AGX_R_TPOS + 2)};
    double tq[4] = {RF(m, d, AGX_R_TQUAT), RF(m, d, AGX_R_TQUAT + 1), RF(m, d, AGX_R_TQUAT + 2), RF(m, d, AGX_R_TQUAT + 3)};
    double ax[3] = {RF(m, d, AGX_R_AXIS), RF(m, d, AGX_R_AXIS + 1), RF(m, d, AGX_R_AXIS + 2)};
    double Rt[9], Rq[9], R0[9];
    const int prismatic = RI(m, d, AGX_R_JTYPE) == 1;   /* Sawyer gripper fingers (assets/sawyer/sawyer.urdf) */
    quat_to_mat(tq, Rt); mm3(P->R, Rt, R0);
    xf_apply(P, tp, s->link[d].p);
    if (prismatic) { memcpy(s->link[d].R, R0, sizeof R0); double aw0[3]; mv3(R0, ax, aw0); axpy3(s->q[d], aw0, s->link[d].p); }
    else { axis_angle_mat(ax, s->q[d], Rq); mm3(R0, Rq, s->link[d].R); }
    double com[3] = {RF(m, d, AGX_R_COM), RF(m, d, AGX_R_COM + 1), RF(m, d, AGX_R_COM + 2)};
    xf_apply(&s->link[d], com, s->comw[d]);
    double aw[3]; mv3(s->link[d].R, ax, aw);
    double px[3]; cross3(s->link[d].p, aw, px);
    /* joint screw (angular; linear at the world origin): revolute (a, p x a), prismatic (0, a) */
    for (int k = 0; k < 3; k++) { s->S[d][k] = prismatic ? 0.0 : aw[k]; s->S[d][3 + k] = prismatic ? aw[k] : px[k]; }
    const double ixx = RF(m, d, AGX_R_INERTIA), iyy = RF(m, d, AGX_R_INERTIA + 1), izz = RF(m, d, AGX_R_INERTIA + 2),
                 ixy = RF(m, d, AGX_R_INERTIA + 3), ixz = RF(m, d, AGX_R_INERTIA + 4), iyz = RF(m, d, AGX_R_INERTIA + 5);
    double Il[9] = {ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz}, T[9];
    mm3(s->link[d].R, Il, T); mmt3(T, s->link[d].R, s->Iw[d]);
    spatial_inertia(RF(m, d, AGX_R_MASS), s->comw[d], s->Iw[d], s->I6[d]);
    /* spatial velocity */
    for (int k = 0; k < 6; k++) s->vsp[d][k] = (par < 0 ? 0.0 : s->vsp[par][k]) + s->S[d][k] * s->qd[d];
    double sq[6]; for (int k = 0; k < 6; k++) sq[k] = s->S[d][k] * s->qd[d];
    crm(s->vsp[d], sq, s->cvp[d]);
  }
  for (int b = 0; b < s->nfree; b++) {
    memcpy(s->freex[b].p, s->fpos[b], sizeof(double) * 3);
    quat_to_mat(s->fquat[b], s->freex[b].R);
    double Ii[9] = {0}; double T[9];
    for (int k = 0; k < 3; k++) { double I = FF(m, b, AGX_F_INERTIA + k); Ii[4 * k] = I > 0 ? 1.0 / I : 0.0; }
    mm3(s->freex[b].R, Ii, T); mmt3(T, s->freex[b].R, s->fIinv[b]);
  }
}

/* external spatial force on moving link d (gravity + velocity damping), world origin reference.
 * [BULLET-UNVERIFIED] damping: f = -m v_c (k + k|v_c|), tau = -Ic w (k + k|w|) per link. */
static void link_external_force(const sim_t* s, int d, int with_damping, double* f6) {
  const agxo_model* m = s->m;
  double mass = RF(m, d, AGX_R_MASS);
  double f[3] = {0, 0, mass * PARAM(m, (RI(m, d, AGX_R_KIND) & 1) ? AGX_P_HUMAN_GRAVITY_Z : AGX_P_ROBOT_GRAVITY_Z)}, tau[3] = {0, 0, 0};
  if (with_damping) {
    const double* w = s->vsp[d]; double vc[3], t[3];
    cross3(w, s->comw[d], t); add3(s->vsp[d] + 3, t, vc);
    double kl = PARAM(m, AGX_P_LIN_DAMP), ka = PARAM(m, AGX_P_ANG_DAMP);
    double sl = kl + kl * sqrt(dot3(vc, vc)), sa = ka + ka * sqrt(dot3(w, w));
    axpy3(-mass * sl, vc, f);
    double Iw[3]; mv3(s->Iw[d], w, Iw); axpy3(-sa, Iw, tau);
  }
  double cf[3]; cross3(s->comw[d], f, cf);
  for (int k = 0; k < 3; k++) { f6[k] = tau[k] + cf[k]; f6[3 + k] = f[k]; }
}

/* Featherstone ABA, world-frame spatial quantities.  Leaves IA/U/Dinv cached for minv(). */
static void aba(sim_t* s, const double* tau, int with_damping, double* qdd) {
  const agxo_model* m = s->m; int n = s->ndof;
  double pA[MAXDOF][6], u[MAXDOF], a[MAXDOF][6];
  for (int d = 0; d < n; d++) {
    memcpy(s->IA[d], s->I6[d], sizeof(double) * 36);
    double h[6], fe[6]; mv6(s->I6[d], s->vsp[d], h); crf(s->vsp[d], h, pA[d]);
    link_external_force(s, d, with_damping, fe);
    for (int k = 0; k < 6; k++) pA[d][k] -= fe[k];
  }
  for (int d = n - 1; d >= 0; d--) {
    mv6(s->IA[d], s->S[d], s->U[d]);
    double D = dot6(s->S[d], s->U[d]);
    s->Dinv[d] = (D > 1e-300 && !FROZEN(s, d)) ? 1.0 / D : 0.0;   /* frozen DoF: static link (mass 0, human.py:104-110) */
    double t = (tau ? tau[d] : 0.0) - RF(m, d, AGX_R_JDAMP) * s->qd[d];
    u[d] = t - dot6(s->S[d], pA[d]);
    int par = RI(m, d, AGX_R_PARENT);
    if (par >= 0) {
      double Ia[36];
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Ia[6 * r + c] = s->IA[d][6 * r + c] - s->U[d][r] * s->U[d][c] * s->Dinv[d];
      double Iac[6]; mv6(Ia, s->cvp[d], Iac);
      for (int k = 0; k < 36; k++) s->IA[par][k] += Ia[k];
      for (int k = 0; k < 6; k++) pA[par][k] += pA[d][k] + Iac[k] + s->U[d][k] * (u[d] * s->Dinv[d]);
    }
  }
  for (int d = 0; d < n; d++) {
    int par = RI(m, d, AGX_R_PARENT);
    double ap[6];
    for (int k = 0; k < 6; k++) ap[k] = (par < 0 ? 0.0 : a[par][k]) + s->cvp[d][k];
    qdd[d] = (u[d] - dot6(s->U[d], ap)) * s->Dinv[d];
    for (int k = 0; k < 6; k++) a[d][k] = ap[k] + s->S[d][k] * qdd[d];
  }
}
/* M^-1 column by column: ABA response to a unit joint force (what Bullet's
 * calcAccelerationDeltasMultiDof evaluates per constraint row). Requires aba() caches. */
static void minv_from_aba(sim_t* s) {
  const agxo_model* m = s->m; int n = s->ndof;
  for (int j = 0; j < n; j++) {
    double pA[MAXDOF][6], u[MAXDOF], a[MAXDOF][6];
    memset(pA, 0, sizeof pA);
    for (int d = n - 1; d >= 0; d--) {
      u[d] = (d == j ? 1.0 : 0.0) - dot6(s->S[d], pA[d]);
      int par = RI(m, d, AGX_R_PARENT);
      if (par >= 0) for (int k = 0; k < 6; k++) pA[par][k] += pA[d][k] + s->U[d][k] * (u[d] * s->Dinv[d]);
    }
    for (int d = 0; d < n; d++) {
      int par = RI(m, d, AGX_R_PARENT);
      double ap[6]; for (int k = 0; k < 6; k++) ap[k] = par < 0 ? 0.0 : a[par][k];
      double qdd = (u[d] - dot6(s->U[d], ap)) * s->Dinv[d];
      for (int k = 0; k < 6; k++) a[d][k] = ap[k] + s->S[d][k] * qdd;
      s->Minv[d * n + j] = qdd;
    }
  }
}

/* ------------------------------------------------------------------------------------ GJK */
static int support(const double* v, int n, const double* d) {
  int best = 0; double bd = dot3(v, d);
  for (int k = 1; k < n; k++) { double t = dot3(v + 3 * k, d); if (t > bd) { bd = t; best = k; } }
  return best;
}
/* closest point to the origin on triangle (a,b,c): barycentric weights out (Ericson, RTCD 5.1.5) */
static void closest_tri(const double* a, const double* b, const double* c, double* wa, double* wb, double* wc) {
  double ab[3], ac[3], ap[3], bp[3], cp[3];
  sub3(b, a, ab); sub3(c, a, ac);
  for (int k = 0; k < 3; k++) { ap[k] = -a[k]; bp[k] = -b[k]; cp[k] = -c[k]; }
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { *wa = 1; *wb = 0; *wc = 0; return; }
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { *wa = 0; *wb = 1; *wc = 0; return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); *wa = 1 - v; *wb = v; *wc = 0; return; }
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { *wa = 0; *wb = 0; *wc = 1; return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2 / (d2 - d6); *wa = 1 - w; *wb = 0; *wc = w; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); *wa = 0; *wb = 1 - w; *wc = w; return; }
  double den = 1.0 / (va + vb + vc);
  double v = vb * den, w = vc * den;
  *wa = 1 - v - w; *wb = v; *wc = w;
}
/* simplex of n points W (Minkowski difference) -> closest point to origin; compacts the simplex to
 * the supporting sub-simplex; returns 1 if the origin is enclosed (n==4 case). */
static int simplex_solve(double W[4][3], double A[4][3], double B[4][3], int* pn, double* lam, double* v) {
  int n = *pn;
  double l[4] = {0, 0, 0, 0};
  if (n == 1) { l[0] = 1; }
  else if (n == 2) {
    double d[3]; sub3(W[1], W[0], d);
    double dd = dot3(d, d), t = dd > 0 ? -dot3(W[0], d) / dd : 0.0;
    if (t <= 0) { l[0] = 1; } else if (t >= 1) { l[1] = 1; } else { l[0] = 1 - t; l[1] = t; }
  } else if (n == 3) {
    closest_tri(W[0], W[1], W[2], &l[0], &l[1], &l[2]);
  } else {
    static const int faces[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
    double best = 1e300; int any = 0;
    for (int f = 0; f < 4; f++) {
      const double *a = W[faces[f][0]], *b = W[faces[f][1]], *c = W[faces[f][2]], *d = W[faces[f][3]];
      double ab[3], ac[3], nrm[3], ad[3];
      sub3(b, a, ab); sub3(c, a, ac); cross3(ab, ac, nrm); sub3(d, a, ad);
      double sp = -dot3(a, nrm), sd = dot3(ad, nrm);
      if (sp * sd > 0) continue; /* origin on the same side as the 4th vertex: not outside this face */
      any = 1;
      double wa, wb, wc; closest_tri(a, b, c, &wa, &wb, &wc);
      double p[3]; for (int k = 0; k < 3; k++) p[k] = wa * a[k] + wb * b[k] + wc * c[k];
      double d2 = dot3(p, p);
      if (d2 < best) { best = d2; l[0] = l[1] = l[2] = l[3] = 0; l[faces[f][0]] = wa; l[faces[f][1]] = wb; l[faces[f][2]] = wc; }
    }
    if (!any) return 1;
  }
  int k2 = 0;
  for (int k = 0; k < n; k++) if (l[k] > 0) {
    if (k2 != k) { memcpy(W[k2], W[k], 24); memcpy(A[k2], A[k], 24); memcpy(B[k2], B[k], 24); }
    lam[k2] = l[k]; k2++;
  }
  *pn = k2;
  v[0] = v[1] = v[2] = 0;
  for (int k = 0; k < k2; k++) axpy3(lam[k], W[k], v);
  return 0;
}
/* d0 (may be NULL): a vector from B towards A, typically centre(A) - centre(B).  The first simplex point is the
 * support point of A - B along -d0 (a point of the Minkowski difference that already faces the origin) instead of
 * the difference of the first vertices; it saves about one iteration per pair. */
static int gjk_core(const double* a, int na, const double* b, int nb, double tol, int maxit, const double* d0, double* dist, double* pa, double* pb, int* iters) {
  double W[4][3], A[4][3], B[4][3], lam[4] = {1, 0, 0, 0}, v[3];
  int n = 0, pen = 0, it, ia0 = 0, ib0 = 0;
  if (d0) {
    double d[3] = {d0[0], d0[1], d0[2]};
    if (dot3(d, d) < 1e-12) { d[0] = 1; d[1] = 0; d[2] = 0; }
    double nd[3] = {-d[0], -d[1], -d[2]};
    ia0 = support(a, na, nd); ib0 = support(b, nb, d);
  }
  sub3(a + 3 * ia0, b + 3 * ib0, v);
  double vv = dot3(v, v);
  /* seed simplex so witness points are always defined */
  memcpy(A[0], a + 3 * ia0, 24); memcpy(B[0], b + 3 * ib0, 24); memcpy(W[0], v, 24); n = 1;
  memcpy(pa, a + 3 * ia0, 24); memcpy(pb, b + 3 * ib0, 24);
  for (it = 0; it < maxit; it++) {
    if (vv < 1e-12) { pen = 1; break; }   /* cores closer than 1 micron: treat as overlapping */
    double nv[3] = {-v[0], -v[1], -v[2]};
    int ia = support(a, na, nv), ib = support(b, nb, v);
    double w[3]; sub3(a + 3 * ia, b + 3 * ib, w);
    double vw = dot3(v, w);
    if (vv - vw <= tol * vv) break;
    int dup = 0;
    for (int k = 0; k < n; k++) if (W[k][0] == w[0] && W[k][1] == w[1] && W[k][2] == w[2]) dup = 1;
    if (dup) break;
    memcpy(W[n], w, 24); memcpy(A[n], a + 3 * ia, 24); memcpy(B[n], b + 3 * ib, 24); n++;
    double vn[3];
    /* an enclosed origin contradicts a separating plane (v.w > 0: every point x of A - B has v.x >= v.w > 0): a flat
     * tetrahedron passed the side tests by rounding; the closest points found so far stand (same rule as the device code) */
    if (simplex_solve(W, A, B, &n, lam, vn)) { pen = !(vw > 0); break; }
    double vvn = dot3(vn, vn);
    /* no progress (a degenerate sub-simplex solve can even move away): keep the closest points found so far */
    if (vvn >= vv) break;
    memcpy(v, vn, 24); vv = vvn;
    pa[0] = pa[1] = pa[2] = pb[0] = pb[1] = pb[2] = 0;
    for (int k = 0; k < n; k++) { axpy3(lam[k], A[k], pa); axpy3(lam[k], B[k], pb); }
  }
  if (iters) *iters = it;
  if (pen) { *dist = 0; return 1; }
  *dist = sqrt(vv);
  return 0;
}
int agxo_gjk(const double* a, int na, const double* b, int nb, double tol, int maxit, double* dist, double* pa, double* pb, int* iters) {
  return gjk_core(a, na, b, nb, tol, maxit, NULL, dist, pa, pb, iters);
}

/* ------------------------------------------------------------------------------------ collision */
static const xf_t XF_IDENT = {{0, 0, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1}};
static const xf_t* body_xf(const sim_t* s, int code) {
  if (code == AGX_BODY_WORLD) return &XF_IDENT;
  if (code >= AGX_BODY_HUMAN0) return &s->human[code - AGX_BODY_HUMAN0];
  if (code >= AGX_BODY_FREE0) return &s->freex[code - AGX_BODY_FREE0];
  if (code == AGX_BODY_ROBOT_BASE) return &s->base;
  return &s->link[code];
}
static void collider_aabb(const sim_t* s, int c, double* lo, double* hi) {
  const agxo_model* m = s->m; const xf_t* X = body_xf(s, CI(m, c, AGX_C_BODY));
  double cl[3] = {CF(m, c, AGX_C_AABB_C), CF(m, c, AGX_C_AABB_C + 1), CF(m, c, AGX_C_AABB_C + 2)};
  double hl[3] = {CF(m, c, AGX_C_AABB_H), CF(m, c, AGX_C_AABB_H + 1), CF(m, c, AGX_C_AABB_H + 2)};
  double cw[3]; xf_apply(X, cl, cw);
  double r = CF(m, c, AGX_C_RADIUS);
  for (int k = 0; k < 3; k++) {
    double h = fabs(X->R[3 * k]) * hl[0] + fabs(X->R[3 * k + 1]) * hl[1] + fabs(X->R[3 * k + 2]) * hl[2] + r;
    lo[k] = cw[k] - h; hi[k] = cw[k] + h;
  }
}
static int collider_world_verts(const sim_t* s, int c, const double* shift, double* out) {
  const agxo_model* m = s->m; const xf_t* X = body_xf(s, CI(m, c, AGX_C_BODY));
  int n = CI(m, c, AGX_C_NVERT), off = CI(m, c, AGX_C_VOFF);
  for (int k = 0; k < n; k++) {
    double v[3] = {m->f[m->o_vert + 3 * (off + k)], m->f[m->o_vert + 3 * (off + k) + 1], m->f[m->o_vert + 3 * (off + k) + 2]};
    double w[3]; xf_apply(X, v, w); sub3(w, shift, out + 3 * k);
  }
  return n;
}
/* closest features of two colliders: returns 1 and fills the contact geometry when the separation
 * (radii included) is below `limit`. */
static int narrowphase_ab(const sim_t* s, int ca, int cb, double limit, contact_t* out, const double* alo_in, const double* ahi_in) {
  const agxo_model* m = s->m;
  double va[3 * MAXV], vb[3 * MAXV], lo[3], hi[3], shift[3];
  if (alo_in) { memcpy(lo, alo_in, 24); memcpy(hi, ahi_in, 24); } else collider_aabb(s, ca, lo, hi);
  for (int k = 0; k < 3; k++) shift[k] = 0.5 * (lo[k] + hi[k]);
  int na = collider_world_verts(s, ca, shift, va), nb = collider_world_verts(s, cb, shift, vb);
  /* a large static world box (table top, ground) is replaced by its intersection with the other
   * collider's AABB grown by BOX_CLIP: same closest points (they lie within `limit` of A), but all
   * vertices stay near A so that single-precision GJK remains well conditioned */
  int clipped = 0;
  if (CI(m, cb, AGX_C_BODY) == AGX_BODY_WORLD && nb == 8 && (CI(m, cb, AGX_C_TAG) == AGX_TAG_TABLE || CI(m, cb, AGX_C_TAG) == AGX_TAG_PLANE)) {
    double blo[3], bhi[3], alo[3], ahi[3];
    clipped = 1;
    collider_aabb(s, cb, blo, bhi); memcpy(alo, lo, 24); memcpy(ahi, hi, 24);
    for (int k = 0; k < 3; k++) {
      double lo2 = alo[k] - AGX_BOX_CLIP, hi2 = ahi[k] + AGX_BOX_CLIP;
      if (lo2 > blo[k]) blo[k] = lo2; if (hi2 < bhi[k]) bhi[k] = hi2;
      if (bhi[k] < blo[k]) return 0;
    }
    for (int q = 0; q < 8; q++) for (int k = 0; k < 3; k++) vb[3 * q + k] = ((q >> (2 - k)) & 1 ? bhi[k] : blo[k]) - shift[k];
  }
  double ra = CF(m, ca, AGX_C_RADIUS), rb = CF(m, cb, AGX_C_RADIUS);
  double d, pa[3], pb[3], n[3];
  /* GJK starts along centre(A) - centre(B): body-frame AABB centres of the cores (the clipped box's own centre) */
  double cen[2][3], d0[3];
  for (int q = 0; q < 2; q++) {
    int cc = q ? cb : ca; const xf_t* X = body_xf(s, CI(m, cc, AGX_C_BODY));
    double cl[3] = {CF(m, cc, AGX_C_AABB_C), CF(m, cc, AGX_C_AABB_C + 1), CF(m, cc, AGX_C_AABB_C + 2)};
    xf_apply(X, cl, cen[q]); for (int k = 0; k < 3; k++) cen[q][k] -= shift[k];
  }
  if (clipped) for (int k = 0; k < 3; k++) cen[1][k] = 0.5 * (vb[k] + vb[3 * 7 + k]);
  sub3(cen[0], cen[1], d0);
  int pen = gjk_core(va, na, vb, nb, PARAM(m, AGX_P_GJK_TOL), (int)PARAM(m, AGX_P_GJK_MAXIT), d0, &d, pa, pb, NULL);
  g_stat_narrowphase++; if (pen) g_stat_core_overlap++;
  if (!pen) {
    if (d - ra - rb >= limit) return 0;
    sub3(pa, pb, n); for (int k = 0; k < 3; k++) n[k] /= d;
  } else {
    /* cores overlap: 42-direction penetration sampling (btMinkowskiPenetrationDepthSolver-style) */
    double best = 1e300; int bi = 0;
    for (int k = 0; k < m->ndir; k++) {
      double dir[3] = {m->f[m->o_dirs + 3 * k], m->f[m->o_dirs + 3 * k + 1], m->f[m->o_dirs + 3 * k + 2]};
      double nd[3] = {-dir[0], -dir[1], -dir[2]};
      int ia = support(va, na, nd), ib = support(vb, nb, dir);
      double depth = dot3(vb + 3 * ib, dir) - dot3(va + 3 * ia, dir);
      if (depth < best) { best = depth; bi = k; }
    }
    for (int k = 0; k < 3; k++) n[k] = m->f[m->o_dirs + 3 * bi + k];
    double nd[3] = {-n[0], -n[1], -n[2]};
    int ia = support(va, na, nd);
    memcpy(pa, va + 3 * ia, 24); memcpy(pb, pa, 24); axpy3(best, n, pb);
    d = -best;
  }
  for (int k = 0; k < 3; k++) { out->pa[k] = pa[k] - ra * n[k] + shift[k]; out->pb[k] = pb[k] + rb * n[k] + shift[k]; out->n[k] = n[k]; }
  out->dist = d - ra - rb; out->ca = ca; out->cb = cb;
  out->ba = CI(m, ca, AGX_C_BODY); out->bb = CI(m, cb, AGX_C_BODY);
  double mua = CI(m, ca, AGX_C_TAG) == AGX_TAG_PLANE ? s->plane_mu : CF(m, ca, AGX_C_FRICTION);
  double mub = CI(m, cb, AGX_C_TAG) == AGX_TAG_PLANE ? s->plane_mu : CF(m, cb, AGX_C_FRICTION);
  out->mu = mua * mub; /* [BULLET-UNVERIFIED] combined friction = product */
  out->lambda_n = 0;
  return 1;
}
static int narrowphase(const sim_t* s, int ca, int cb, double limit, contact_t* out);
/* velocity of the material point of body `code` at world point x, from the generalized velocities s->vel */
static void point_velocity(const sim_t* s, int code, const double* x, double* v) {
  const agxo_model* m = s->m;
  v[0] = v[1] = v[2] = 0;
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) {
    double sv[6] = {0, 0, 0, 0, 0, 0};
    for (int d = code; d >= 0; d = RI(m, d, AGX_R_PARENT)) for (int k = 0; k < 6; k++) sv[k] += s->S[d][k] * s->vel[d];
    double wx[3]; cross3(sv, x, wx); add3(sv + 3, wx, v);
  } else if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) {
    int b = code - AGX_BODY_FREE0, o = s->ndof + 6 * b; double r[3], wr[3];
    sub3(x, s->fpos[b], r); cross3(s->vel + o + 3, r, wr); add3(s->vel + o, wr, v);
  }
}
static int narrowphase(const sim_t* s, int ca, int cb, double limit, contact_t* out) { return narrowphase_ab(s, ca, cb, limit, out, NULL, NULL); }
/* Face manifold: a collider resting on the top face of a static world box (table, ground) touches it along a face or
 * an edge, where the closest point of GJK is not unique and a single contact point makes the body rock.  Bullet
 * accumulates up to 4 points per pair in its persistent manifold (btPersistentManifold [BULLET-UNVERIFIED]); without
 * per-pair state the same support polygon is rebuilt every substep from the vertices of A: among the vertices above
 * the box's footprint and within AGX_FACE_BAND of the lowest one, the one farthest (horizontally) from the points
 * chosen so far is added, as long as it is at least AGX_FACE_SPREAD away; up to AGX_FACE_EXTRA times.  The band makes
 * the choice insensitive to which of several (nearly) coplanar vertices is the lowest by a rounding error.
 * k: the GJK contact of the pair (normal +z); returns the number of extra contacts written to out[]. */
static int face_manifold(const sim_t* s, int ca, int cb, contact_t* k, const double* blo, const double* bhi, contact_t* out) {
  const agxo_model* m = s->m;
  if (!(CI(m, cb, AGX_C_BODY) == AGX_BODY_WORLD && CI(m, cb, AGX_C_NVERT) == 8 && (CI(m, cb, AGX_C_TAG) == AGX_TAG_TABLE || CI(m, cb, AGX_C_TAG) == AGX_TAG_PLANE))) return 0;
  int na = CI(m, ca, AGX_C_NVERT);
  if (na < 2 || k->n[2] <= 0.999) return 0;
  double zero[3] = {0, 0, 0}, va[3 * MAXV]; collider_world_verts(s, ca, zero, va);
  double ra = CF(m, ca, AGX_C_RADIUS), top = bhi[2];        /* bhi: the box's AABB, radius included */
  double zmin = 1e300;
  for (int v = 0; v < na; v++) { const double* p = va + 3 * v; if (p[0] >= blo[0] && p[0] <= bhi[0] && p[1] >= blo[1] && p[1] <= bhi[1] && p[2] < zmin) zmin = p[2]; }
  if (zmin > 1e299) return 0;                               /* no vertex above the footprint: keep the GJK contact */
  /* On a face the closest point of GJK is an arbitrary point of the face (it differs between f32 and f64 and from
   * substep to substep); the pair's first contact is therefore re-anchored at a vertex too: the first vertex (in
   * model order) inside the band.  For a vertex or edge contact that is GJK's own witness point. */
  for (int v = 0; v < na; v++) {
    const double* p = va + 3 * v;
    if (!(p[0] >= blo[0] && p[0] <= bhi[0] && p[1] >= blo[1] && p[1] <= bhi[1]) || p[2] > zmin + (double)AGX_FACE_BAND) continue;
    k->pa[0] = p[0]; k->pa[1] = p[1]; k->pa[2] = p[2] - ra; k->pb[0] = p[0]; k->pb[1] = p[1]; k->pb[2] = top;
    k->n[0] = 0; k->n[1] = 0; k->n[2] = 1; k->dist = p[2] - ra - top;
    break;
  }
  double cx[1 + AGX_FACE_EXTRA], cy[1 + AGX_FACE_EXTRA]; int nch = 1, nout = 0;
  cx[0] = k->pa[0]; cy[0] = k->pa[1];
  for (int e = 0; e < AGX_FACE_EXTRA; e++) {
    int bi = -1; double bd = (double)AGX_FACE_SPREAD * (double)AGX_FACE_SPREAD;
    for (int v = 0; v < na; v++) {
      const double* p = va + 3 * v;
      if (!(p[0] >= blo[0] && p[0] <= bhi[0] && p[1] >= blo[1] && p[1] <= bhi[1]) || p[2] > zmin + (double)AGX_FACE_BAND) continue;
      double dmin = 1e300;
      for (int c = 0; c < nch; c++) { double dx = p[0] - cx[c], dy = p[1] - cy[c], d2 = dx * dx + dy * dy; if (d2 < dmin) dmin = d2; }
      if (dmin > bd) { bd = dmin; bi = v; }   /* strictly farther: the lowest index wins ties */
    }
    if (bi < 0) break;
    const double* p = va + 3 * bi;
    contact_t* o = &out[nout++]; *o = *k;
    o->pa[0] = p[0]; o->pa[1] = p[1]; o->pa[2] = p[2] - ra; o->pb[0] = p[0]; o->pb[1] = p[1]; o->pb[2] = top;
    o->n[0] = 0; o->n[1] = 0; o->n[2] = 1; o->dist = p[2] - ra - top; o->lambda_n = 0;
    cx[nch] = p[0]; cy[nch] = p[1]; nch++;
  }
  return nout;
}
/* speed bound of any material point of collider c under the predicted velocities: |v(centre)| + |w| * rho */
static double collider_speed(const sim_t* s, int c) {
  const agxo_model* m = s->m; int code = CI(m, c, AGX_C_BODY);
  const xf_t* X = body_xf(s, code);
  double cl[3] = {CF(m, c, AGX_C_AABB_C), CF(m, c, AGX_C_AABB_C + 1), CF(m, c, AGX_C_AABB_C + 2)};
  double hl[3] = {CF(m, c, AGX_C_AABB_H), CF(m, c, AGX_C_AABB_H + 1), CF(m, c, AGX_C_AABB_H + 2)};
  double cw[3], v[3], w[3] = {0, 0, 0}; xf_apply(X, cl, cw); point_velocity(s, code, cw, v);
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) { for (int d = code; d >= 0; d = RI(m, d, AGX_R_PARENT)) for (int k = 0; k < 3; k++) w[k] += s->S[d][k] * s->vel[d]; }
  else if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) { int b = code - AGX_BODY_FREE0; memcpy(w, s->vel + s->ndof + 6 * b + 3, 24); }
  double rho = sqrt(dot3(hl, hl)) + CF(m, c, AGX_C_RADIUS);
  return sqrt(dot3(v, v)) + sqrt(dot3(w, w)) * rho;
}
/* K2+K3: static pair table -> AABB cull -> GJK; contact order = (group, a, b) enumeration order */
static void collide(sim_t* s) {
  const agxo_model* m = s->m;
  double brk = PARAM(m, AGX_P_CONTACT_BREAK);
  int maxc = (int)PARAM(m, AGX_P_MAX_CONTACTS); if (maxc > MAXC) maxc = MAXC;
  double lo[MAXCOLL][3], hi[MAXCOLL][3];
  double dt0 = m->dt;
  for (int c = 0; c < m->ncoll && c < MAXCOLL; c++) {
    /* speculative AABB: grown by the distance the collider can travel in this substep, so the
     * broadphase margin only has to cover the solver slack (pairs farther apart cannot yield a row) */
    collider_aabb(s, c, lo[c], hi[c]);
    double g = collider_speed(s, c) * dt0;
    for (int k = 0; k < 3; k++) { lo[c][k] -= g; hi[c][k] += g; }
  }
  s->ncon = 0; s->food_near_human = 0; s->contact_overflow = 0; s->nqpt = 0; s->nman = 0;
  double dt = m->dt, slack = PARAM(m, AGX_P_CONTACT_SLACK);
  for (int g = 0; g < m->ngroup; g++) {
    int a0 = GI(m, g, AGX_G_A0), a1 = GI(m, g, AGX_G_A1), b0 = GI(m, g, AGX_G_B0), b1 = GI(m, g, AGX_G_B1);
    if (s->gender == 1 && GI(m, g, AGX_G_B0F) >= 0) { b0 = GI(m, g, AGX_G_B0F); b1 = GI(m, g, AGX_G_B1F); }
    int same = GI(m, g, AGX_G_FLAGS) & 1, keep = GI(m, g, AGX_G_KEEP);
    {   /* bit3 / bit4: male / female only; bit5: only while some human DoF is dynamic */
      const int fl = GI(m, g, AGX_G_FLAGS);
      if (((fl & 8) && s->gender != 0) || ((fl & 16) && s->gender != 1)) continue;
      if ((fl & 32) && m->ndof <= 32 && ((~s->frozen >> m->nrobot) & ((1u << m->nhdof) - 1u)) == 0) continue;
    }
    /* bit6: a solver row for EVERY contact of the group inside the break distance, not only those that can close within the substep by
     * their predicted velocity (the groups whose forces the task reports: robot / tool against the person).  The predicted velocity knows
     * nothing of the motor rows: a pad driven onto the arm closes gaps the prediction calls open (profiles/r05/approximation_budget.json) */
    const double gslack = (GI(m, g, AGX_G_FLAGS) & 64) ? brk : slack;
    const double mg = (GI(m, g, AGX_G_FLAGS) & (2 | 64)) ? brk : slack;   /* bit1: getContactPoints-style existence query */
    for (int a = a0; a < a1; a++) {
      contact_t cand[MAXCAND]; double gap[MAXCAND]; int nc = 0;
      for (int b = (same ? a + 1 : b0); b < b1; b++) {
        if (GI(m, g, AGX_G_FLAGS) & 4) {   /* self-collision: not the same link, not parent and child */
          int la = CI(m, a, AGX_C_BODY), lb = CI(m, b, AGX_C_BODY);
          if (la == lb || (la >= 0 && la < AGX_BODY_ROBOT_BASE && lb >= 0 && lb < AGX_BODY_ROBOT_BASE && (RI(m, la, AGX_R_PARENT) == lb || RI(m, lb, AGX_R_PARENT) == la))) continue;
        }
        int sep = 0;
        for (int k = 0; k < 3; k++) if (lo[a][k] > hi[b][k] + mg || lo[b][k] > hi[a][k] + mg) sep = 1;
        if (sep) continue;
        contact_t k;
        if (!narrowphase_ab(s, a, b, brk, &k, lo[a], hi[a])) continue;
        if ((GI(m, g, AGX_G_FLAGS) & 2) && s->nman < MAXMAN) s->man[s->nman++] = k;
        /* a manifold point exists (what getContactPoints reports, agent.py:100-116) */
        if (m->task_kind == AGX_TASK_FEEDING && CI(m, a, AGX_C_TAG) == AGX_TAG_FOOD && CI(m, b, AGX_C_TAG) == AGX_TAG_HUMAN)
          s->food_near_human |= 1 << (CI(m, a, AGX_C_BODY) - AGX_BODY_FREE0 - m->food0);
        /* bed bathing: manifold points of tool link 1 on the human, with or without force (bed_bathing.py:47-58) */
        if (m->task_kind != AGX_TASK_FEEDING && (GI(m, g, AGX_G_FLAGS) & 2) && CI(m, a, AGX_C_TAG) == AGX_TAG_TOOL && (TI(m, AGX_T_PAD_LINK) >> (CI(m, a, AGX_C_LINK) + 1) & 1) &&
            CI(m, b, AGX_C_TAG) == AGX_TAG_HUMAN && s->nqpt < MAXQPT) {
          memcpy(s->qpt[s->nqpt], k.pb, 24); s->qpt_link[s->nqpt] = CI(m, b, AGX_C_LINK); s->nqpt++;
        }
        /* resting on a static world box: vertex contacts (re-anchors k, adds up to AGX_FACE_EXTRA more) */
        contact_t extra[AGX_FACE_EXTRA];
        int ne = face_manifold(s, a, b, &k, lo[b], hi[b], extra);
        /* solver row only if the gap can close within this substep */
        double va[3], vb[3], vr[3]; point_velocity(s, k.ba, k.pa, va); point_velocity(s, k.bb, k.pb, vb); sub3(va, vb, vr);
        double pg = k.dist + dot3(vr, k.n) * dt;
        if (pg < gslack && nc < MAXCAND) { cand[nc] = k; gap[nc] = pg; nc++; }
        for (int e = 0; e < ne; e++) {
          if (extra[e].dist >= brk) continue;
          point_velocity(s, extra[e].ba, extra[e].pa, va); point_velocity(s, extra[e].bb, extra[e].pb, vb); sub3(va, vb, vr);
          double pg2 = extra[e].dist + dot3(vr, extra[e].n) * dt;
          if (pg2 < gslack && nc < MAXCAND) { cand[nc] = extra[e]; gap[nc] = pg2; nc++; }
        }
      }
      /* keep the `keep` candidates with the smallest predicted gap (ties: lower B index), emitted in
       * selection order; keep == 0 keeps all in B order */
      int nsel = (keep > 0 && keep < nc) ? keep : nc;
      for (int q = 0; q < nsel; q++) {
        int bi = q;
        if (keep > 0) { bi = -1; for (int c = 0; c < nc; c++) if (gap[c] < 1e299 && (bi < 0 || gap[c] < gap[bi])) bi = c; }
        if (s->ncon >= maxc) { s->contact_overflow++; } else s->con[s->ncon++] = cand[bi];
        gap[bi] = 1e300;
      }
    }
  }
}

/* persistent manifold of the [BULLET-UNVERIFIED] switch AGX_P_MANIFOLD (include/agx_blob.h): cached contact points of one environment, in
 * cache order; cleared together with the warm-start memory */
#define MAXMP 64
typedef struct { int ca, cb; double la[3], lb[3], n[3], dist, mu; } mpoint_t;
static int g_mp_n = 0; static mpoint_t g_mp[MAXMP];
static int g_mp_stats[4];      /* replaced (nearest), appended, replaced by the area rule, dropped at the refresh -- since the last agxo_manifold_stats() */
/* test hooks: the cache as rows of 12 doubles {collider a, collider b, local point on A (3), on B (3), world normal (3), friction} */
int agxo_manifold_get(double* out, int max_out) {
  int n = g_mp_n < max_out ? g_mp_n : max_out;
  for (int p = 0; p < n; p++) { double* o = out + 12 * p; o[0] = g_mp[p].ca; o[1] = g_mp[p].cb; memcpy(o + 2, g_mp[p].la, 24); memcpy(o + 5, g_mp[p].lb, 24); memcpy(o + 8, g_mp[p].n, 24); o[11] = g_mp[p].mu; }
  return n;
}
void agxo_manifold_set(const double* in, int n) {
  g_mp_n = n < MAXMP ? n : MAXMP;
  for (int p = 0; p < g_mp_n; p++) { const double* o = in + 12 * p; g_mp[p].ca = (int)o[0]; g_mp[p].cb = (int)o[1]; memcpy(g_mp[p].la, o + 2, 24); memcpy(g_mp[p].lb, o + 5, 24); memcpy(g_mp[p].n, o + 8, 24); g_mp[p].mu = o[11]; g_mp[p].dist = 0; }
}
void agxo_manifold_stats(int* out4) { memcpy(out4, g_mp_stats, sizeof g_mp_stats); memset(g_mp_stats, 0, sizeof g_mp_stats); }
/* AGX_P_MANIFOLD: refresh the cached points, merge the substep's GJK contacts into them, rebuild the contact list (see include/agx_blob.h) */
static int mp_face_pair(const agxo_model* m, int cb) {
  return CI(m, cb, AGX_C_BODY) == AGX_BODY_WORLD && CI(m, cb, AGX_C_NVERT) == 8 && (CI(m, cb, AGX_C_TAG) == AGX_TAG_TABLE || CI(m, cb, AGX_C_TAG) == AGX_TAG_PLANE);
}
static void xf_apply_inv(const xf_t* X, const double* w, double* o) {
  double d[3]; sub3(w, X->p, d);
  for (int k = 0; k < 3; k++) o[k] = X->R[k] * d[0] + X->R[3 + k] * d[1] + X->R[6 + k] * d[2];
}
static void manifold_update(sim_t* s) {
  const agxo_model* m = s->m;
  const double brk = PARAM(m, AGX_P_CONTACT_BREAK), slack = PARAM(m, AGX_P_CONTACT_SLACK), dt = m->dt;
  int maxc = (int)PARAM(m, AGX_P_MAX_CONTACTS); if (maxc > MAXC) maxc = MAXC;
  /* (1) refresh */
  int n = 0;
  for (int p = 0; p < g_mp_n; p++) {
    mpoint_t q = g_mp[p];
    double pa[3], pb[3], d[3];
    xf_apply(body_xf(s, CI(m, q.ca, AGX_C_BODY)), q.la, pa); xf_apply(body_xf(s, CI(m, q.cb, AGX_C_BODY)), q.lb, pb);
    sub3(pa, pb, d);
    q.dist = dot3(d, q.n);
    if (q.dist > brk) { g_mp_stats[3]++; continue; }
    double drift[3]; for (int k = 0; k < 3; k++) drift[k] = pb[k] - (pa[k] - q.n[k] * q.dist);
    if (dot3(drift, drift) > brk * brk) { g_mp_stats[3]++; continue; }
    g_mp[n++] = q;
  }
  g_mp_n = n;
  /* (2) merge the substep's contacts (not those of the face manifold) */
  for (int c = 0; c < s->ncon; c++) {
    const contact_t* k = &s->con[c];
    if (mp_face_pair(m, k->cb)) continue;
    mpoint_t q; q.ca = k->ca; q.cb = k->cb; memcpy(q.n, k->n, 24); q.dist = k->dist; q.mu = k->mu;
    xf_apply_inv(body_xf(s, k->ba), k->pa, q.la); xf_apply_inv(body_xf(s, k->bb), k->pb, q.lb);
    int idx[4], cnt = 0, nearest = -1; double shortest = brk * brk;
    for (int p = 0; p < g_mp_n; p++) {
      if (g_mp[p].ca != q.ca || g_mp[p].cb != q.cb) continue;
      double d[3]; sub3(g_mp[p].la, q.la, d); const double d2 = dot3(d, d);
      if (d2 < shortest) { shortest = d2; nearest = p; }
      if (cnt < 4) idx[cnt] = p;
      cnt++;
    }
    if (nearest >= 0) { g_mp[nearest] = q; g_mp_stats[0]++; continue; }
    if (cnt < 4) { if (g_mp_n < MAXMP) g_mp[g_mp_n++] = q; g_mp_stats[1]++; continue; }
    g_mp_stats[2]++;
    /* four cached: sortCachedPoints -- the deepest of the five stays, the replacement leaves the largest area */
    int deepest = -1; double pen = q.dist;
    for (int i = 0; i < 4; i++) if (g_mp[idx[i]].dist < pen) { deepest = i; pen = g_mp[idx[i]].dist; }
    double res[4] = {0, 0, 0, 0};
    static const int other[4][3] = {{1, 3, 2}, {0, 3, 2}, {0, 3, 1}, {0, 2, 1}};     /* res_i = |(new - p[o0]) x (p[o1] - p[o2])|^2 */
    for (int i = 0; i < 4; i++) {
      if (deepest == i) continue;
      double a[3], b[3], x[3];
      sub3(q.la, g_mp[idx[other[i][0]]].la, a); sub3(g_mp[idx[other[i][1]]].la, g_mp[idx[other[i][2]]].la, b); cross3(a, b, x);
      res[i] = dot3(x, x);
    }
    int bi = -1; double bv = -1e300;
    for (int i = 0; i < 4; i++) if (fabs(res[i]) > bv) { bv = fabs(res[i]); bi = i; }
    g_mp[idx[bi]] = q;
  }
  /* (3) the contact list */
  contact_t out[MAXC]; int no = 0, overflow = 0;
  unsigned char used[MAXMP]; memset(used, 0, sizeof used);
  for (int c = 0; c <= s->ncon; c++) {
    const contact_t* k = c < s->ncon ? &s->con[c] : NULL;
    if (k && mp_face_pair(m, k->cb)) { if (no < maxc) out[no++] = *k; else overflow++; continue; }
    for (int p = 0; p < g_mp_n; p++) {
      if (used[p] || (k && (g_mp[p].ca != k->ca || g_mp[p].cb != k->cb))) continue;      /* k == NULL: the cached pairs without a new point */
      used[p] = 1;
      const mpoint_t* q = &g_mp[p];
      contact_t e; e.ca = q->ca; e.cb = q->cb; e.ba = CI(m, q->ca, AGX_C_BODY); e.bb = CI(m, q->cb, AGX_C_BODY);
      xf_apply(body_xf(s, e.ba), q->la, e.pa); xf_apply(body_xf(s, e.bb), q->lb, e.pb);
      memcpy(e.n, q->n, 24); e.dist = q->dist; e.mu = q->mu; e.lambda_n = 0; memset(e.t1, 0, sizeof e.t1);
      double va[3], vb[3], vr[3]; point_velocity(s, e.ba, e.pa, va); point_velocity(s, e.bb, e.pb, vb); sub3(va, vb, vr);
      if (!(e.dist + dot3(vr, e.n) * dt < slack)) continue;
      if (no < maxc) out[no++] = e; else overflow++;
    }
  }
  memcpy(s->con, out, sizeof(contact_t) * no); s->ncon = no; s->contact_overflow += overflow;
}

/* ------------------------------------------------------------------------------------ rows */
/* Jacobian contribution of a unit force `f` (and unit torque `t`) applied on body `code` at world
 * point x (lever arms about the world origin for links, about the COM for free bodies). */
static void body_jacobian(const sim_t* s, int code, const double* x, const double* f, const double* t, double sign, double* J) {
  const agxo_model* m = s->m;
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) {
    double F[6], xf[3] = {0, 0, 0};
    if (f) cross3(x, f, xf);
    for (int k = 0; k < 3; k++) { F[k] = xf[k] + (t ? t[k] : 0.0); F[3 + k] = f ? f[k] : 0.0; }
    for (int d = code; d >= 0; d = RI(m, d, AGX_R_PARENT)) J[d] += sign * dot6(s->S[d], F);
  } else if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) {
    int b = code - AGX_BODY_FREE0, o = s->ndof + 6 * b;
    double r[3], rf[3] = {0, 0, 0};
    if (f) { sub3(x, s->fpos[b], r); cross3(r, f, rf); }
    for (int k = 0; k < 3; k++) { J[o + k] += sign * (f ? f[k] : 0.0); J[o + 3 + k] += sign * (rf[k] + (t ? t[k] : 0.0)); }
  }
}
static void finish_row(sim_t* s, row_t* r) {
  const agxo_model* m = s->m; int n = s->ndof;
  memset(r->B, 0, sizeof r->B);
  for (int i = 0; i < n; i++) { double acc = 0; for (int j = 0; j < n; j++) acc += s->Minv[i * n + j] * r->J[j]; r->B[i] = acc; }
  for (int b = 0; b < s->nfree; b++) {
    int o = n + 6 * b; double mass = FF(m, b, AGX_F_MASS), im = mass > 0 ? 1.0 / mass : 0.0;
    for (int k = 0; k < 3; k++) r->B[o + k] = im * r->J[o + k];
    mv3(s->fIinv[b], r->J + o + 3, r->B + o + 3);
  }
  double D = 0; for (int k = 0; k < s->nv; k++) D += r->J[k] * r->B[k];
  r->invD = D > 1e-12 ? 1.0 / D : 0.0;
  r->lambda = 0; r->fric_of = -1; r->mu = 0;
}
static double row_vel(const sim_t* s, const row_t* r) { double a = 0; for (int k = 0; k < s->nv; k++) a += r->J[k] * s->vel[k]; return a; }

/* rotation matrix -> XYZ Euler angles (small relative rotations of the fixed constraint) */
static void mat_to_euler_xyz(const double* R, double* e) {
  double fi = R[2];
  if (fi < 1.0) { if (fi > -1.0) { e[0] = atan2(-R[5], R[8]); e[1] = asin(R[2]); e[2] = atan2(-R[1], R[0]); }
    else { e[0] = -atan2(R[3], R[4]); e[1] = -M_PI / 2; e[2] = 0; } }
  else { e[0] = atan2(R[3], R[4]); e[1] = M_PI / 2; e[2] = 0; }
}
static void plane_space(const double* n, double* p) { /* first tangent of btPlaneSpace1 */
  if (fabs(n[2]) > 0.7071067811865475244) { double a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a); p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k; }
  else { double a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a); p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0; }
}
/* end-effector frame of tool t: 0 = robot.right_end_effector (AGX_T_EE_*), 1 = the second tool's (AGX_T_EE2_*) */
static void ee_frame_of(const sim_t* s, int t, xf_t* ee) {
  const agxo_model* m = s->m; int L = TI(m, t ? AGX_T_EE2_LINK : AGX_T_EE_LINK), op = t ? AGX_T_EE2_POS : AGX_T_EE_POS, oq = t ? AGX_T_EE2_QUAT : AGX_T_EE_QUAT;
  double p[3] = {TF(m, op), TF(m, op + 1), TF(m, op + 2)};
  double q[4] = {TF(m, oq), TF(m, oq + 1), TF(m, oq + 2), TF(m, oq + 3)}, Rq[9];
  quat_to_mat(q, Rq); xf_apply(&s->link[L], p, ee->p); mm3(s->link[L].R, Rq, ee->R);
}
static void ee_frame(const sim_t* s, xf_t* ee) { ee_frame_of(s, 0, ee); }
static int n_tools(const agxo_model* m) { return m->nfree > 0 ? (TI(m, AGX_T_TOOL2_BODY) > 0 ? 2 : 1) : 0; }

/* number of articulated DoF entries a row stores: the robot block, the human block, or both */
static int art_entries_of(const sim_t* s, int has_robot, int has_human) {
  const agxo_model* m = s->m;
  if (has_robot && has_human) return m->ndof; if (has_robot) return m->nrobot; if (has_human) return m->nhdof; return 0;
}
static int art_range_entries(const sim_t* s, int ba, int bb) {
  const agxo_model* m = s->m; int r = 0, h = 0;
  if (ba >= 0 && ba < AGX_BODY_ROBOT_BASE) { if (ba < m->nrobot) r = 1; else h = 1; }
  if (bb >= 0 && bb < AGX_BODY_ROBOT_BASE) { if (bb < m->nrobot) r = 1; else h = 1; }
  return art_entries_of(s, r, h);
}
static int row_art_entries(const sim_t* s, const row_t* r) {
  const agxo_model* m = s->m; int rr = 0, hh = 0;
  for (int d = 0; d < m->nrobot; d++) if (r->J[d] != 0) rr = 1;
  for (int d = m->nrobot; d < m->ndof; d++) if (r->J[d] != 0) hh = 1;
  return art_entries_of(s, rr, hh);
}
/* warm-start memory of the [BULLET-UNVERIFIED] switch AGX_P_WARMSTART: normal impulses of the last substep that was solved, by
 * (collider a, collider b, ordinal inside the pair); one environment at a time (the sensitivity study), cleared by agxo_warm_clear() */
static int g_warm_n = 0, g_warm_key[MAXC][3]; static double g_warm_lam[MAXC];
void agxo_warm_clear(void) { g_warm_n = 0; g_mp_n = 0; }
/* test hooks (tests/diag/resting_contact_sensitivity.py): the warm-start memory as rows of 4 doubles {collider a, collider b, ordinal, impulse} */
int agxo_warm_get(double* out, int max_out) {
  int n = g_warm_n < max_out ? g_warm_n : max_out;
  for (int p = 0; p < n; p++) { out[4 * p] = g_warm_key[p][0]; out[4 * p + 1] = g_warm_key[p][1]; out[4 * p + 2] = g_warm_key[p][2]; out[4 * p + 3] = g_warm_lam[p]; }
  return n;
}
void agxo_warm_set(const double* in, int n) {
  g_warm_n = n < MAXC ? n : MAXC;
  for (int p = 0; p < g_warm_n; p++) { g_warm_key[p][0] = (int)in[4 * p]; g_warm_key[p][1] = (int)in[4 * p + 1]; g_warm_key[p][2] = (int)in[4 * p + 2]; g_warm_lam[p] = in[4 * p + 3]; }
}
static void build_rows(sim_t* s) {
  const agxo_model* m = s->m; int n = s->ndof;
  double dt = m->dt, erp = PARAM(m, AGX_P_ERP), cerp = PARAM(m, AGX_P_CONTACT_ERP);
  int maxrows = (int)PARAM(m, AGX_P_MAX_ROWS); if (maxrows > MAXROWS) maxrows = MAXROWS;
  s->nrows = 0;
#define NEWROW() (memset(&s->rows[s->nrows], 0, sizeof(row_t)), &s->rows[s->nrows++])
  /* joint motors: Agent.control (agent.py:28-33) -> POSITION_CONTROL velocity-level row.
   * [BULLET-UNVERIFIED] target dv = kp (q*-q)/dt + kd (0 - qd), impulse clamp maxForce*dt */
  for (int d = 0; d < n; d++) {
    double maxf = RF(m, d, AGX_R_MAXF); if (maxf <= 0 || FROZEN(s, d)) continue;
    row_t* r = NEWROW(); r->J[d] = 1.0; finish_row(s, r);
    double kp = RF(m, d, AGX_R_KP);
    if (d >= m->nrobot && s->human_kp > 0) { kp = s->human_kp; maxf = s->human_maxf; }   /* reactive hold of a human that is not an agent (human.py:124-127) */
    r->b = kp * (s->qt[d] - s->q[d]) / dt + RF(m, d, AGX_R_KD) * (0.0 - s->vel[d]);
    r->lo = -maxf * dt; r->hi = maxf * dt;
  }
  /* joint limits (URDF lower/upper): unilateral rows, built only when the gap is small */
  for (int d = 0; d < n; d++) {
    if (!RI(m, d, AGX_R_HAS_LIMIT) || FROZEN(s, d)) continue;
    for (int side = 0; side < 2; side++) {
      double gap = side == 0 ? s->q[d] - dof_lower(s, d) : dof_upper(s, d) - s->q[d];
      if (gap >= PARAM(m, AGX_P_LIMIT_ACT)) continue;
      row_t* r = NEWROW(); r->J[d] = side == 0 ? 1.0 : -1.0; finish_row(s, r);
      double rv = row_vel(s, r);
      r->b = gap > 0 ? (-gap / dt - rv) : (-gap * erp / dt - rv);
      r->lo = 0; r->hi = 1e30;
    }
  }
  /* tool fixed constraint (tool.py:46-47): 3 linear rows along world axes at the pivots,
   * 3 angular rows about the parent frame axes; impulse clamp maxForce*dt */
  const int ntool = n_tools(m), has_tool = ntool > 0;
  for (int t = 0; t < ntool; t++) {
    xf_t ee; ee_frame_of(s, t, &ee);
    int L = TI(m, t ? AGX_T_EE2_LINK : AGX_T_EE_LINK), tb = t ? TI(m, AGX_T_TOOL2_BODY) : m->tool_body, code_b = AGX_BODY_FREE0 + tb;
    const int otp = t ? AGX_T_TOOL2_POS : AGX_T_TOOL_POS, otq = t ? AGX_T_TOOL2_QUAT : AGX_T_TOOL_QUAT;
    double tp[3] = {TF(m, otp), TF(m, otp + 1), TF(m, otp + 2)};
    double tq[4] = {TF(m, otq), TF(m, otq + 1), TF(m, otq + 2), TF(m, otq + 3)};
    double pivA[3], Rt[9], frameA[9];
    xf_apply(&ee, tp, pivA); quat_to_mat(tq, Rt); mm3(ee.R, Rt, frameA);
    /* child frame = the tool's base (URDF root link) frame: COM frame o REF */
    double pivB[3], frameB[9];
    { double rp[3] = {FF(m, tb, AGX_F_REFPOS), FF(m, tb, AGX_F_REFPOS + 1), FF(m, tb, AGX_F_REFPOS + 2)};
      double rq[4] = {FF(m, tb, AGX_F_REFQUAT), FF(m, tb, AGX_F_REFQUAT + 1), FF(m, tb, AGX_F_REFQUAT + 2), FF(m, tb, AGX_F_REFQUAT + 3)}, Rr[9];
      quat_to_mat(rq, Rr); xf_apply(&s->freex[tb], rp, pivB); mm3(s->freex[tb].R, Rr, frameB); }
    double rel[9], At[9];
    for (int r0 = 0; r0 < 3; r0++) for (int c = 0; c < 3; c++) At[3 * r0 + c] = frameA[3 * c + r0];
    mm3(At, frameB, rel);
    double ang[3]; mat_to_euler_xyz(rel, ang);
    double lim = TF(m, AGX_T_TOOL_MAXF) * dt;
    for (int k = 0; k < 3; k++) {
      double nrm[3] = {0, 0, 0}; nrm[k] = 1.0;
      row_t* r = NEWROW();
      body_jacobian(s, L, pivA, nrm, NULL, 1.0, r->J); body_jacobian(s, code_b, pivB, nrm, NULL, -1.0, r->J);
      finish_row(s, r);
      double perr = pivA[k] - pivB[k];
      r->b = -perr * erp / dt - row_vel(s, r); r->lo = -lim; r->hi = lim;
    }
    for (int k = 0; k < 3; k++) {
      double axw[3] = {frameA[k], frameA[3 + k], frameA[6 + k]};
      row_t* r = NEWROW();
      body_jacobian(s, L, pivA, NULL, axw, 1.0, r->J); body_jacobian(s, code_b, pivB, NULL, axw, -1.0, r->J);
      finish_row(s, r);
      r->b = ang[k] * erp / dt - row_vel(s, r); r->lo = -lim; r->hi = lim;
    }
  }
  /* contact normals, then one friction row per contact.  The contact list is truncated to the
   * longest prefix that fits the row budget and the (J,B) coefficient budget: a row stores one
   * pair per DoF of each dynamic body it touches (all robot DoFs, 6 per free body). */
  int first_normal = s->nrows, nc = 0;
  const int fdirs = (int)PARAM(m, AGX_P_FRICTION_DIRS) == 2 ? 2 : 1;   /* [BULLET-UNVERIFIED] switch, oracle and device */
  const double wsf = PARAM(m, AGX_P_WARMSTART);
  {
    int ent = 1, maxent = (int)PARAM(m, AGX_P_MAX_ENTRIES);
    /* non-contact rows: motors and limits address the robot; the 6 tool rows (last) robot + tool */
    for (int r0 = 0; r0 < first_normal; r0++) ent += row_art_entries(s, &s->rows[r0]) + (has_tool && r0 >= first_normal - 6 * ntool ? 6 : 0);
    int acc = 0;
    for (int c = 0; c < s->ncon; c++) {
      const contact_t* k = &s->con[c];
      int e = art_range_entries(s, k->ba, k->bb) + ((k->ba >= AGX_BODY_FREE0 && k->ba < AGX_BODY_HUMAN0) ? 6 : 0) + ((k->bb >= AGX_BODY_FREE0 && k->bb < AGX_BODY_HUMAN0) ? 6 : 0);
      acc += e;
      if (first_normal + (1 + fdirs) * (c + 1) > maxrows || ent + (1 + fdirs) * acc > maxent) break;
      nc = c + 1;
    }
  }
  for (int c = 0; c < nc; c++) {
    contact_t* k = &s->con[c]; row_t* r = NEWROW();
    body_jacobian(s, k->ba, k->pa, k->n, NULL, 1.0, r->J); body_jacobian(s, k->bb, k->pb, k->n, NULL, -1.0, r->J);
    finish_row(s, r);
    double rv = row_vel(s, r);
    r->b = k->dist > 0 ? (-k->dist / dt - rv) : (-k->dist * cerp / dt - rv);
    { const double sp = PARAM(m, AGX_P_SPLIT_PEN); if (sp > 0 && k->dist < -sp) r->b = -rv; }   /* split impulse: no positional term below the threshold */
    r->lo = 0; r->hi = 1e30;
    if (wsf > 0) {   /* warm start: the same contact (collider pair, ordinal inside the pair) of the previous substep */
      int ord = 0; for (int c2 = 0; c2 < c; c2++) if (s->con[c2].ca == k->ca && s->con[c2].cb == k->cb) ord++;
      for (int p = 0; p < g_warm_n; p++) if (g_warm_key[p][0] == k->ca && g_warm_key[p][1] == k->cb && g_warm_key[p][2] == ord) { r->lambda = wsf * g_warm_lam[p]; break; }
    }
  }
  for (int c = 0; c < nc; c++) {
    contact_t* k = &s->con[c];
    /* relative velocity of the contact points, lateral part */
    row_t tmp; double vr[3];
    for (int ax = 0; ax < 3; ax++) {
      double e[3] = {0, 0, 0}; e[ax] = 1.0; memset(tmp.J, 0, sizeof tmp.J);
      body_jacobian(s, k->ba, k->pa, e, NULL, 1.0, tmp.J); body_jacobian(s, k->bb, k->pb, e, NULL, -1.0, tmp.J);
      vr[ax] = row_vel(s, &tmp);
    }
    double vn = dot3(vr, k->n), t[3] = {vr[0] - vn * k->n[0], vr[1] - vn * k->n[1], vr[2] - vn * k->n[2]};
    double l2 = dot3(t, t);
    if (l2 > PARAM(m, AGX_P_FRIC_EPS)) { double il = 1.0 / sqrt(l2); for (int q = 0; q < 3; q++) t[q] *= il; }
    else plane_space(k->n, t);
    row_t* r = NEWROW();
    body_jacobian(s, k->ba, k->pa, t, NULL, 1.0, r->J); body_jacobian(s, k->bb, k->pb, t, NULL, -1.0, r->J);
    finish_row(s, r);
    r->b = -row_vel(s, r); r->fric_of = first_normal + c; r->mu = k->mu; r->lo = 0; r->hi = 0;
    memcpy(k->t1, t, sizeof t);
  }
  if (fdirs == 2) for (int c = 0; c < nc; c++) {   /* the second direction of the friction pyramid: a block of its own behind the first directions */
    contact_t* k = &s->con[c];
    double t2[3]; cross3(k->n, k->t1, t2);
    row_t* r2 = NEWROW();
    body_jacobian(s, k->ba, k->pa, t2, NULL, 1.0, r2->J); body_jacobian(s, k->bb, k->pb, t2, NULL, -1.0, r2->J);
    finish_row(s, r2);
    r2->b = -row_vel(s, r2); r2->fric_of = first_normal + c; r2->mu = k->mu; r2->lo = 0; r2->hi = 0;
  }
  s->contact_overflow += s->ncon - nc;   /* dropped by the row / coefficient budgets */
  s->ncon = nc;
#undef NEWROW
}

/* K6: projected Gauss-Seidel, fixed number of sweeps, rows in construction order */
double g_pgs_stats[8];
/* AGX_P_NOOP_RETEST = K > 0: a row of the non-friction block (motors, limits, tool rows, contact normals) whose visit in a re-test
 * sweep (sweep index divisible by K) changed nothing -- an inactive contact or limit (0 -> 0), a motor sitting on its force bound --
 * is not visited in the K - 1 sweeps that follow.  Skipped visits that would have been no-ops too change nothing; the others (a row
 * that would have woken up between two re-tests) are the approximation, quantified in profiles/r03/noop_retest_sensitivity.json.
 * K = 0: plain projected Gauss-Seidel, every row in every sweep.  Friction rows: a row whose bound is zero and whose impulse is zero
 * is an exact no-op and is skipped by the device in any case. */
static void pgs(sim_t* s, double* dv) {
  int iters = (int)PARAM(s->m, AGX_P_NITER);
  int K = (int)PARAM(s->m, AGX_P_NOOP_RETEST);
  { const double pen = PARAM(s->m, AGX_P_NOOP_PEN);          /* a pressed contact (deeper than AGX_P_NOOP_PEN): plain sweeps in this substep */
    if (K > 0 && pen > 0) for (int c = 0; c < s->ncon; c++) if (s->con[c].dist < -pen) K = 0;
    /* ... and so does any contact of the robot or its tool with the person (resting contacts sit at dist ~ 0 and still carry the forces the
     * task reports) */
    if (K > 0 && pen > 0) for (int c = 0; c < s->ncon; c++) {
      const int ta = CI(s->m, s->con[c].ca, AGX_C_TAG), tb = CI(s->m, s->con[c].cb, AGX_C_TAG);
      if ((ta == AGX_TAG_HUMAN && (tb == AGX_TAG_ROBOT || tb == AGX_TAG_TOOL)) || (tb == AGX_TAG_HUMAN && (ta == AGX_TAG_ROBOT || ta == AGX_TAG_TOOL))) K = 0;
    } }
  unsigned char skip[MAXROWS]; memset(skip, 0, sizeof skip);
  memset(dv, 0, sizeof(double) * NVMAX);
  g_pgs_stats[5] += 1;
  const double reps = PARAM(s->m, AGX_P_ORACLE_RESIDUAL_EPS);
  for (int i = 0; i < s->nrows; i++) if (s->rows[i].lambda != 0) for (int k = 0; k < s->nv; k++) dv[k] += s->rows[i].B[k] * s->rows[i].lambda;   /* warm-started rows */
  for (int it = 0; it < iters; it++) {
   double res = 0;
   for (int i = 0; i < s->nrows; i++) {
    row_t* r = &s->rows[i];
    if (r->invD == 0) continue;
    const int retest = K > 0 && it % K == 0;
    if (K > 0 && !retest && r->fric_of < 0 && skip[i]) continue;
    double lo = r->lo, hi = r->hi;
    if (r->fric_of >= 0) { double ln = s->rows[r->fric_of].lambda; hi = r->mu * ln; lo = -hi; }
    double jdv = 0; for (int k = 0; k < s->nv; k++) jdv += r->J[k] * dv[k];
    double nl = r->lambda + (r->b - jdv) * r->invD;
    if (nl < lo) nl = lo; if (nl > hi) nl = hi;
    double dl = nl - r->lambda; r->lambda = nl;
    if (retest && r->fric_of < 0) skip[i] = dl == 0;
    if (r->fric_of < 0) { g_pgs_stats[0] += 1; if (dl == 0) g_pgs_stats[1] += 1; }
    else { g_pgs_stats[2] += 1; if (dl == 0) g_pgs_stats[3] += 1; if (hi == 0 && nl == 0 && dl == 0) g_pgs_stats[4] += 1; }
    if (dl != 0) for (int k = 0; k < s->nv; k++) dv[k] += r->B[k] * dl;
    { const double e = dl / r->invD; if (e * e > res) res = e * e; }
   }
   g_pgs_stats[6] += 1;
   if (reps > 0 && res <= reps) break;   /* [BULLET-UNVERIFIED] switch: residual early-out */
  }
}

/* FeedingEnv.update_targets (feeding.py:192-196): mouth = head pose o mouth offset */
static void update_target(sim_t* s) {
  const agxo_model* m = s->m; if (m->task_kind != AGX_TASK_FEEDING && m->task_kind != AGX_TASK_DRINKING) return;   /* feeding.py:192-196, drinking.py:192-196 */
  int hl = TI(m, AGX_T_HEAD_LINK), o = s->gender == 1 ? AGX_T_MOUTH_F : AGX_T_MOUTH_M;
  double mp[3] = {TF(m, o), TF(m, o + 1), TF(m, o + 2)};
  xf_apply(&s->link[hl], mp, s->target);   /* link frames of the CURRENT kinematics() call */
}

/* Human.enforce_realistic_joint_limits (human.py:134-152): Keras Sequential[Dense(4->64,tanh) x3, Dense(64->1,sigmoid)]
 * (assets/realistic_arm_limits_model.h5; weights = the blob's MLP section) on the remapped arm angles (human.py:142-145);
 * class 1 remembers the pose, class 0 puts the four joints back to the last valid pose with zero velocity. */
static double wrap_2pi(double x) { return x - 2 * M_PI * floor(x / (2 * M_PI)); }   /* Python's % for a positive modulus */
double agxo_arm_limit_logit(const agxo_model* m, const double* in4) {
  const float* W1 = m->f + m->i[AGX_H_OFF_MLP]; const float* W2 = W1 + 4 * 64 + 64; const float* W3 = W2 + 64 * 64 + 64; const float* W4 = W3 + 64 * 64 + 64;
  double h[64], g[64];
  for (int j = 0; j < 64; j++) { double a = W1[256 + j]; for (int k = 0; k < 4; k++) a += in4[k] * W1[64 * k + j]; h[j] = tanh(a); }
  for (int layer = 0; layer < 2; layer++) {
    const float* W = layer == 0 ? W2 : W3;
    for (int j = 0; j < 64; j++) { double a = W[4096 + j]; for (int k = 0; k < 64; k++) a += h[k] * W[64 * k + j]; g[j] = tanh(a); }
    memcpy(h, g, sizeof h);
  }
  double z = W4[64]; for (int k = 0; k < 64; k++) z += h[k] * W4[k];
  return z;   /* class = sigmoid(z) > 0.5 <=> z > 0 */
}
static void arm_limits(sim_t* s) {
  const agxo_model* m = s->m;
  if (m->task_kind == AGX_TASK_FEEDING || !TI(m, AGX_T_ARM_LIMIT_ON)) return;
  double sg = TF(m, AGX_T_ARM_LIMIT_SIGN), a[4]; int dof[4];
  for (int k = 0; k < 4; k++) {
    dof[k] = TI(m, AGX_T_ARM_LIMIT_DOF + k); a[k] = s->q[dof[k]];
    /* read as the reference's strict limit reset leaves them (the reset here has the tolerance AGX_LIMIT_EPS) */
    if (a[k] < dof_lower(s, dof[k])) a[k] = dof_lower(s, dof[k]); if (a[k] > dof_upper(s, dof[k])) a[k] = dof_upper(s, dof[k]);
  }
  double in4[4] = {wrap_2pi(sg * a[0] + 2 * M_PI), wrap_2pi(a[1] + 2 * M_PI), sg * a[2], wrap_2pi(-a[3] + 2 * M_PI)};   /* human.py:142-145 */
  if (agxo_arm_limit_logit(m, in4) > 0) { for (int k = 0; k < 4; k++) s->arm_prev[k] = s->q[dof[k]]; s->arm_has_prev = 1; }
  else if (s->arm_has_prev) for (int k = 0; k < 4; k++) {
    double v = s->arm_prev[k]; if (v < dof_lower(s, dof[k])) v = dof_lower(s, dof[k]); if (v > dof_upper(s, dof[k])) v = dof_upper(s, dof[k]);
    s->q[dof[k]] = v; s->qd[dof[k]] = 0;
  }
}

/* one p.stepSimulation() (env.py:226) + the post-substep hooks (env.py:227-232) */

/* ------------------------------------------------------------------------------------ cloth (dressing.py:149-157,184; model/cloth.py)
 * One internal substep of the garment: Bullet's btSoftBody position-based step [BULLET-UNVERIFIED, restated from the published
 * source layout: btSoftBody::predictMotion / solveConstraints / PSolve_Anchors / PSolve_RContacts / PSolve_Links]:
 *   1. v += g dt; aerodynamic drag (aero model V_Point: only while a node moves along its normal), clamped so that it cannot
 *      reverse the node; q = x; x += v dt
 *   2. node-vs-rigid contacts: every node within the collision margin of a shape (anchored nodes excepted) gets a contact
 *      plane at the margin surface; friction state c3 from the displacement of this substep
 *   3. piterations x [anchors, rigid contacts, links]; links class by class (classes share no node)
 *   4. v = (x - q) / dt (1 - kDP)
 * The rigid bodies do not feel the cloth: the human and the robot are multibodies, which btSoftBody's rigid-contact and anchor
 * solvers do not push (their impulses go to btRigidBody only), and the attachment sphere has no mass. */
#define CLH(m, k) ((m)->i[(m)->o_cloth + (k)])
#define CLPAR(m, k) ((double)(m)->f[(m)->o_cloth + CLH(m, AGX_CL_OFF_PARAM) + (k)])
#define CLOTH_NODE_CONTACTS AGX_CLOTH_NODE_CONTACTS
typedef struct { double n[3], offset, c3, imp[3]; int sh; } ccontact_t;

/* signed distance of world point x to the surface of shape sh (negative inside) and the outward normal, world frame */
static double cloth_shape_distance(const sim_t* s, int sh, const double* x, double* nw) {
  const agxo_model* m = s->m; const int32_t* S = m->i + m->o_cloth + CLH(m, AGX_CL_OFF_SHAPE) + 4 * sh;
  const int c = S[0], p0 = S[1], np = S[2];
  const xf_t* X = body_xf(s, CI(m, c, AGX_C_BODY));
  double d[3], xl[3]; sub3(x, X->p, d); mtv3(X->R, d, xl);
  const double rad = CF(m, c, AGX_C_RADIUS);
  double nl[3], dist;
  if (np == 0) {           /* sphere / capsule: distance to the core point or segment */
    const float* v = m->f + m->o_vert + 3 * CI(m, c, AGX_C_VOFF);
    double a[3] = {v[0], v[1], v[2]}, cp[3] = {a[0], a[1], a[2]};
    if (CI(m, c, AGX_C_NVERT) == 2) {
      double b[3] = {v[3], v[4], v[5]}, ab[3], ax[3]; sub3(b, a, ab); sub3(xl, a, ax);
      double t = dot3(ab, ab) > 0 ? dot3(ax, ab) / dot3(ab, ab) : 0; t = t < 0 ? 0 : (t > 1 ? 1 : t);
      for (int k = 0; k < 3; k++) cp[k] = a[k] + t * ab[k];
    }
    sub3(xl, cp, nl); double len = sqrt(dot3(nl, nl));
    if (len > 1e-12) { for (int k = 0; k < 3; k++) nl[k] /= len; } else { nl[0] = 0; nl[1] = 0; nl[2] = 1; }
    dist = len - rad;
  } else {                 /* hull: the largest face-plane distance (exact inside and in front of a face) */
    const float* P = m->f + m->o_cloth + CLH(m, AGX_CL_OFF_PLANE) + 4 * p0;
    int best = 0; double bd = -1e300;
    for (int k = 0; k < np; k++) { double t = P[4 * k] * xl[0] + P[4 * k + 1] * xl[1] + P[4 * k + 2] * xl[2] - P[4 * k + 3]; if (t > bd) { bd = t; best = k; } }
    nl[0] = P[4 * best]; nl[1] = P[4 * best + 1]; nl[2] = P[4 * best + 2];
    dist = bd - rad;
  }
  mv3(X->R, nl, nw);
  return dist;
}

/* the 0.1 m closest-point query of a water particle against the cup (drinking.py:77: w.get_closest_points(self.tool, distance=0.1) empty
 * -> spilled): is the particle's surface within `dist` of a piece of the tool?  At that range the distance of a point to a centimetre-sized
 * convex piece is its distance to the piece's nearest VERTEX to within (piece size)^2 / (8 d) ~ 0.5 mm, whereas the face-plane value of
 * cloth_shape_distance is a lower bound that runs up to ~10-30 % low off edges (the penalty fired late).  Vertices it is. */
static int particle_near_tool(const sim_t* s, const double* x, double dist) {
  const agxo_model* m = s->m; const int NS = m->i[m->o_cloth + AGX_CL_NSHAPE];
  const double rw = CLPAR(m, AGX_CP_MARGIN);
  for (int sh = 0; sh < NS; sh++) {
    const int c = m->i[m->o_cloth + m->i[m->o_cloth + AGX_CL_OFF_SHAPE] + 4 * sh];
    if (CI(m, c, AGX_C_TAG) != AGX_TAG_TOOL) continue;
    const xf_t* X = body_xf(s, CI(m, c, AGX_C_BODY));
    double d[3], xl[3]; sub3(x, X->p, d); mtv3(X->R, d, xl);
    const double lim = dist + rw + CF(m, c, AGX_C_RADIUS);
    const float* v = m->f + m->o_vert + 3 * CI(m, c, AGX_C_VOFF);
    for (int k = 0; k < CI(m, c, AGX_C_NVERT); k++) {
      const double e0 = xl[0] - v[3 * k], e1 = xl[1] - v[3 * k + 1], e2 = xl[2] - v[3 * k + 2];
      if (e0 * e0 + e1 * e1 + e2 * e2 <= lim * lim) return 1;
    }
  }
  return 0;
}

static void cloth_substep(sim_t* s) {
  const agxo_model* m = s->m; const int oc = m->o_cloth;
  const int NN = CLH(m, AGX_CL_NN), NCOL = CLH(m, AGX_CL_NCOLOR), NA = CLH(m, AGX_CL_NANCHOR), NS = CLH(m, AGX_CL_NSHAPE);
  const int32_t* color = m->i + oc + CLH(m, AGX_CL_OFF_COLOR);
  const int32_t* linki = m->i + oc + CLH(m, AGX_CL_OFF_LINK); const float* linkf = m->f + oc + CLH(m, AGX_CL_OFF_LINK);
  const int32_t* nodei = m->i + oc + CLH(m, AGX_CL_OFF_NODE); const float* nodef = m->f + oc + CLH(m, AGX_CL_OFF_NODE);
  const int32_t* face = m->i + oc + CLH(m, AGX_CL_OFF_FACE);
  const int32_t* anci = m->i + oc + CLH(m, AGX_CL_OFF_ANCHOR); const float* ancf = m->f + oc + CLH(m, AGX_CL_OFF_ANCHOR);
  const double dt = m->dt, kLST = CLPAR(m, AGX_CP_KLST), kDP = CLPAR(m, AGX_CP_KDP), kDG = CLPAR(m, AGX_CP_KDG), kDF = CLPAR(m, AGX_CP_KDF);
  const double kCHR = CLPAR(m, AGX_CP_KCHR), kAHR = CLPAR(m, AGX_CP_KAHR), mrg = CLPAR(m, AGX_CP_MARGIN), im = CLPAR(m, AGX_CP_NODE_IM);
  const double rho = CLPAR(m, AGX_CP_AIR_DENSITY); const int piter = (int)CLPAR(m, AGX_CP_PITER);
  double (*x)[3] = (double (*)[3])s->cx, (*v)[3] = (double (*)[3])s->cv, (*q)[3] = (double (*)[3])s->cq;
  /* the attachment sphere sits where the end effector was when the current stepSimulation call began (dressing.py:200-210) */
  if (!s->anchor_set) { xf_t ee; ee_frame(s, &ee); memcpy(s->anchor, ee.p, 24); s->anchor_set = 1; }
  /* world AABB of every shape, grown by the margin: nodes outside cannot be in contact */
  double (*slo)[3] = (double (*)[3])malloc(sizeof(double) * 3 * NS), (*shi)[3] = (double (*)[3])malloc(sizeof(double) * 3 * NS);
  for (int sh = 0; sh < NS; sh++) {
    const int c = m->i[oc + CLH(m, AGX_CL_OFF_SHAPE) + 4 * sh];
    /* the core's body-frame box rotated into the world, grown by radius + margin (the same conservative box as the device code: a
     * hull's face-plane distance can accept points beyond any box, so both sides must cull alike) */
    const xf_t* X = body_xf(s, CI(m, c, AGX_C_BODY));
    double cl_[3] = {CF(m, c, AGX_C_AABB_C), CF(m, c, AGX_C_AABB_C + 1), CF(m, c, AGX_C_AABB_C + 2)}, cw[3];
    xf_apply(X, cl_, cw);
    const double r = (double)(float)((float)CF(m, c, AGX_C_RADIUS) + (float)mrg + 1e-6f);
    for (int k = 0; k < 3; k++) {
      const double h = fabs(X->R[3 * k]) * CF(m, c, AGX_C_AABB_H) + fabs(X->R[3 * k + 1]) * CF(m, c, AGX_C_AABB_H + 1) + fabs(X->R[3 * k + 2]) * CF(m, c, AGX_C_AABB_H + 2) + r;
      slo[sh][k] = cw[k] - h; shi[sh][k] = cw[k] + h;
    }
  }
  uint8_t* attached = (uint8_t*)calloc(NN, 1);
  for (int a = 0; a < NA; a++) attached[anci[4 * a]] = 1;
  ccontact_t* con = (ccontact_t*)malloc(sizeof(ccontact_t) * CLOTH_NODE_CONTACTS * NN);
  int* ncon = (int*)calloc(NN, sizeof(int));
  /* 1. forces and prediction */
  for (int i = 0; i < NN; i++) {
    double nrm[3] = {0, 0, 0};
    for (int e = nodei[2 * i]; e < nodei[2 * i + 2]; e++) {
      const int j = face[e] & 0xffff, k = (face[e] >> 16) & 0xffff; double a[3], b[3], cr[3];
      sub3(x[j], x[i], a); sub3(x[k], x[i], b); cross3(a, b, cr); add3(nrm, cr, nrm);
    }
    double nl = sqrt(dot3(nrm, nrm)); if (nl > 1.1920929e-7) for (int k = 0; k < 3; k++) nrm[k] /= nl;   /* btSoftBody::updateNormals */
    double vi[3] = {v[i][0], v[i][1], v[i][2] + s->dr_gravity * dt};
    const double v2 = dot3(vi, vi);
    if (kDG > 0 && v2 > 1.1920929e-7) {
      const double dvn = dot3(vi, nrm);
      if (dvn > 0) {
        const double vl = sqrt(v2), c1 = nodef[2 * i + 1] * dvn * v2 / 2 * rho, fmag = c1 * kDG;   /* force = -v/|v| * fmag */
        const double dtim = dt * im;
        if (fmag * dtim * fmag * dtim > v2) { vi[0] = 0; vi[1] = 0; vi[2] = 0; }                 /* ApplyClampedForce: it may stop the node, not reverse it */
        else for (int k = 0; k < 3; k++) vi[k] -= vi[k] / vl * fmag * dtim;
      }
    }
    for (int k = 0; k < 3; k++) { q[i][k] = x[i][k]; v[i][k] = vi[k]; }
  }
  for (int i = 0; i < NN; i++) for (int k = 0; k < 3; k++) x[i][k] = q[i][k] + v[i][k] * dt;
  /* 2. contacts with the rigid shapes (CollideSDF_RS::DoNode) */
  for (int i = 0; i < NN; i++) {
    if (attached[i]) continue;
    for (int sh = 0; sh < NS && ncon[i] < CLOTH_NODE_CONTACTS; sh++) {
      const int c = m->i[oc + CLH(m, AGX_CL_OFF_SHAPE) + 4 * sh];
      const int only = m->i[oc + CLH(m, AGX_CL_OFF_SHAPE) + 4 * sh + 3];
      if (only && only != s->gender + 1) continue;        /* the other gender's colliders are not in the world */
      int out = 0; for (int k = 0; k < 3; k++) if (x[i][k] < slo[sh][k] || x[i][k] > shi[sh][k]) out = 1;
      if (out) continue;
      double nw[3]; const double dst = cloth_shape_distance(s, sh, x[i], nw) - mrg;
      if (dst >= 0) continue;
      if (getenv("AGXO_TRACE_NODE") && atoi(getenv("AGXO_TRACE_NODE")) == i) fprintf(stderr, "node %d shape %d collider %d tag %d body %d dst %g n %g %g %g x %g %g %g\n", i, sh, c, CI(m, c, AGX_C_TAG), CI(m, c, AGX_C_BODY), dst, nw[0], nw[1], nw[2], x[i][0], x[i][1], x[i][2]);
      ccontact_t* k = &con[CLOTH_NODE_CONTACTS * i + ncon[i]++];
      memcpy(k->n, nw, 24); k->offset = -dot3(nw, x[i]) + dst; k->imp[0] = k->imp[1] = k->imp[2] = 0; k->sh = sh;
      double vr[3]; sub3(x[i], q[i], vr); const double dn = dot3(vr, nw); double fv[3] = {vr[0] - nw[0] * dn, vr[1] - nw[1] * dn, vr[2] - nw[2] * dn};
      const double fc = kDF * CF(m, c, AGX_C_FRICTION);
      k->c3 = dot3(fv, fv) < (dn * fc * dn * fc) ? 0 : 1 - fc;
    }
  }
  /* 3. position solver */
  for (int it = 0; it < piter; it++) {
    for (int a = 0; a < NA; a++) {                       /* PSolve_Anchors: x += -(x - q) + (target - x) kAHR  (static attachment body) */
      const int i = anci[4 * a];
      for (int k = 0; k < 3; k++) { const double wa = s->anchor[k] + ancf[4 * a + 1 + k]; x[i][k] += -(x[i][k] - q[i][k]) + (wa - x[i][k]) * kAHR; }
    }
    for (int i = 0; i < NN; i++) for (int cc = 0; cc < ncon[i]; cc++) {   /* PSolve_RContacts */
      ccontact_t* k = &con[CLOTH_NODE_CONTACTS * i + cc];
      double vr[3]; sub3(x[i], q[i], vr); const double dn = dot3(vr, k->n);
      if (dn <= 1.1920929e-7) {
        double dp = dot3(x[i], k->n) + k->offset; if (dp > mrg) dp = mrg;
        for (int a = 0; a < 3; a++) {
          const double fva = vr[a] - k->n[a] * dn, corr = vr[a] - fva * k->c3 + k->n[a] * dp * kCHR;
          x[i][a] -= corr; k->imp[a] += corr / (dt * im);                /* impulse = c0 * corr, c0 = 1 / (dt im) for a static partner */
        }
      }
    }
    for (int c = 0; c < NCOL; c++) for (int l = color[c]; l < color[c + 1]; l++) {   /* PSolve_Links */
      if (linki[2 * l] < 0) continue;                                  /* an empty slot of the kernel's bank schedule (model/cloth.py) */
      const int a = linki[2 * l] & 0xffff, b = (linki[2 * l] >> 16) & 0xffff; const double c1 = linkf[2 * l + 1];
      double del[3]; sub3(x[b], x[a], del); const double len = dot3(del, del);
      if (c1 + len > 1.1920929e-7) {
        const double k = (c1 - len) / (c1 + len) * kLST * 0.5;           /* ((c1 - len) / (c0 (c1 + len))) * im with c0 = 2 im / kLST */
        for (int t = 0; t < 3; t++) { x[a][t] -= del[t] * k; x[b][t] += del[t] * k; }
      }
    }
  }
  /* 4. velocities; contact report (position of the node, force = summed impulse / dt) */
  for (int i = 0; i < NN; i++) for (int k = 0; k < 3; k++) v[i][k] = (x[i][k] - q[i][k]) / dt * (1 - kDP);
  s->nccon = 0;
  for (int i = 0; i < NN; i++) for (int cc = 0; cc < ncon[i]; cc++) {
    const ccontact_t* k = &con[CLOTH_NODE_CONTACTS * i + cc]; s->ccon_node[s->nccon] = i; s->ccon_shape[s->nccon] = k->sh; double* o = s->ccon + 6 * s->nccon++;
    for (int a = 0; a < 3; a++) { o[a] = x[i][a]; o[3 + a] = k->imp[a] / dt; }
  }
  free(slo); free(shi); free(attached); free(con); free(ncon);
}
/* The water of the drinking task (AGX_CL_PARTICLES; drinking.py:160-177): NN free spheres of radius r = MARGIN, one-way coupled to the rigid
 * scene like the garment (they see the cup, the gripper and the person where this substep starts).  Position-based, every step local to a
 * particle or a Jacobi pass, so that a device kernel with one lane per particle reproduces it:
 *   1. v += g dt, x = q + v dt;
 *   2. candidate shapes per particle (at most WATER_CONTACTS, in shape order): those within 2 r + |v| dt of where the substep starts; each
 *      contributes the half space of the face (or tangent plane) the particle is in front of THERE;
 *   3. PITER iterations: (a) particle <-> particle: every particle sums, over its overlapping neighbours in ascending index order and from
 *      the positions the pass starts with, half of each overlap along the centre line, divides by their number and moves by that; (b) each
 *      particle against the half spaces of its candidates in order (the shapes have the last word of an iteration: the pile cannot press a
 *      particle through a wall);
 *   4. v = (x - q) / dt (1 - KDP); a particle that touched a shape loses the share KDF x friction of its tangential velocity.
 * [deviation: Bullet solves the spheres as rigid bodies inside its sequential-impulse solve, with rolling] */
#define WATER_CONTACTS 12
static void water_substep(sim_t* s) {
  const agxo_model* m = s->m; const int oc = m->o_cloth;
  const int NN = CLH(m, AGX_CL_NN), NS = CLH(m, AGX_CL_NSHAPE);
  const double dt = m->dt, r = CLPAR(m, AGX_CP_MARGIN), kDP = CLPAR(m, AGX_CP_KDP), kDF = CLPAR(m, AGX_CP_KDF); const int piter = (int)CLPAR(m, AGX_CP_PITER);
  double (*x)[3] = (double (*)[3])s->cx, (*v)[3] = (double (*)[3])s->cv, (*q)[3] = (double (*)[3])s->cq;
  double (*slo)[3] = (double (*)[3])malloc(sizeof(double) * 3 * NS), (*shi)[3] = (double (*)[3])malloc(sizeof(double) * 3 * NS);
  for (int sh = 0; sh < NS; sh++) {                         /* world boxes of the shapes (as cloth_substep builds them), grown per particle below */
    const int c = m->i[oc + CLH(m, AGX_CL_OFF_SHAPE) + 4 * sh];
    const xf_t* X = body_xf(s, CI(m, c, AGX_C_BODY));
    double cl_[3] = {CF(m, c, AGX_C_AABB_C), CF(m, c, AGX_C_AABB_C + 1), CF(m, c, AGX_C_AABB_C + 2)}, cw[3];
    xf_apply(X, cl_, cw);
    const double g = (double)(float)((float)CF(m, c, AGX_C_RADIUS) + 1e-6f);
    for (int k = 0; k < 3; k++) {
      const double h = fabs(X->R[3 * k]) * CF(m, c, AGX_C_AABB_H) + fabs(X->R[3 * k + 1]) * CF(m, c, AGX_C_AABB_H + 1) + fabs(X->R[3 * k + 2]) * CF(m, c, AGX_C_AABB_H + 2) + g;
      slo[sh][k] = cw[k] - h; shi[sh][k] = cw[k] + h;
    }
  }
  typedef struct { double n[3], off; int sh, hit; } wplane_t;                  /* half space n . p >= off + r of a candidate shape */
  wplane_t (*cand)[WATER_CONTACTS] = (wplane_t (*)[WATER_CONTACTS])malloc(sizeof(wplane_t) * WATER_CONTACTS * NN); int* ncand = (int*)calloc(NN, sizeof(int));
  for (int i = 0; i < NN; i++) {
    for (int k = 0; k < 3; k++) q[i][k] = x[i][k];
    v[i][2] += s->dr_gravity * dt;
    const double reach = 2 * r + sqrt(dot3(v[i], v[i])) * dt;
    for (int sh = 0; sh < NS && ncand[i] < WATER_CONTACTS; sh++) {
      const int only = m->i[oc + CLH(m, AGX_CL_OFF_SHAPE) + 4 * sh + 3];
      if (only && only != s->gender + 1) continue;
      int out = 0; for (int k = 0; k < 3; k++) if (q[i][k] < slo[sh][k] - reach || q[i][k] > shi[sh][k] + reach) out = 1;
      if (out) continue;
      /* the plane is taken where the substep STARTS (the face the particle is in front of there): re-evaluating the nearest face while the pile
       * presses a particle into a thin piece would flip it to the far side */
      double nw[3]; const double d = cloth_shape_distance(s, sh, q[i], nw);
      if (d >= reach) continue;
      wplane_t* c = &cand[i][ncand[i]++]; memcpy(c->n, nw, 24); c->off = dot3(nw, q[i]) - d; c->sh = sh; c->hit = 0;
    }
    for (int k = 0; k < 3; k++) x[i][k] = q[i][k] + v[i][k] * dt;
  }
  double (*dx)[3] = (double (*)[3])malloc(sizeof(double) * 3 * NN);
  for (int it = 0; it < piter; it++) {
    for (int i = 0; i < NN; i++) {
      int cnt = 0; dx[i][0] = dx[i][1] = dx[i][2] = 0;
      for (int j = 0; j < NN; j++) {
        if (j == i) continue;
        double del[3]; sub3(x[i], x[j], del); const double d2 = dot3(del, del);
        if (d2 >= 4 * r * r) continue;
        const double d = sqrt(d2);
        if (d > 1.1920929e-7) { const double sc = 0.5 * (2 * r - d) / d; for (int t = 0; t < 3; t++) dx[i][t] += del[t] * sc; }
        else dx[i][2] += (i > j ? 1.0 : -1.0) * r;          /* coincident centres: apart along z, the higher index up */
        cnt++;
      }
      if (cnt > 1) for (int t = 0; t < 3; t++) dx[i][t] /= cnt;
    }
    for (int i = 0; i < NN; i++) for (int t = 0; t < 3; t++) x[i][t] += dx[i][t];
    for (int i = 0; i < NN; i++) for (int cc = 0; cc < ncand[i]; cc++) {
      wplane_t* c = &cand[i][cc]; const double d = dot3(c->n, x[i]) - c->off - r;
      if (d < 0) { for (int k = 0; k < 3; k++) x[i][k] -= c->n[k] * d; c->hit = 1; }
    }
  }
  s->nccon = 0;
  for (int i = 0; i < NN; i++) {
    for (int k = 0; k < 3; k++) v[i][k] = (x[i][k] - q[i][k]) / dt * (1 - kDP);
    for (int cc = 0; cc < ncand[i]; cc++) {
      const wplane_t* c = &cand[i][cc];
      if (!c->hit) continue;
      const int col = m->i[oc + CLH(m, AGX_CL_OFF_SHAPE) + 4 * c->sh];
      const double vn = dot3(v[i], c->n), fc = kDF * CF(m, col, AGX_C_FRICTION);
      for (int k = 0; k < 3; k++) v[i][k] -= (v[i][k] - c->n[k] * vn) * (fc < 1 ? fc : 1);
      s->ccon_node[s->nccon] = i; s->ccon_shape[s->nccon] = c->sh; double* o = s->ccon + 6 * s->nccon++;
      for (int k = 0; k < 3; k++) { o[k] = x[i][k]; o[3 + k] = 0; }
    }
  }
  free(slo); free(shi); free(cand); free(ncand); free(dx);
}
/* one internal substep; `hooks`: this substep ends a p.stepSimulation() call, after which the reference enforces the human's joint
 * limits and the pose-dependent arm limits (env.py:226-232) */
/* test hook: the world frames of the moving links and the free bodies where each internal substep starts, in the layout the build kernel
 * leaves them in for the cloth / water kernels (agx_blob.h, trace: [substep][NDOF + NFREE][p(3), R(9)]); NULL switches it off */
static float* g_trace_out = NULL; static int g_trace_k = 0;
void agxo_trace_into(float* buf) { g_trace_out = buf; g_trace_k = 0; }
static void substep_h(sim_t* s, int hooks) {
  const agxo_model* m = s->m; int n = s->ndof; double dt = m->dt;
  kinematics(s);
  if (g_trace_out) {
    float* o = g_trace_out + (size_t)12 * (n + m->nfree) * g_trace_k++;
    for (int d = 0; d < n + m->nfree; d++) { const xf_t* X = d < n ? &s->link[d] : &s->freex[d - n]; for (int k = 0; k < 3; k++) o[12 * d + k] = (float)X->p[k]; for (int k = 0; k < 9; k++) o[12 * d + 3 + k] = (float)X->R[k]; }
  }
  if (s->cx && m->i[m->o_cloth + AGX_CL_PARTICLES]) water_substep(s);
  else if (s->cx) cloth_substep(s);   /* one-way coupling: the cloth sees the rigid bodies where this substep starts (btSoftBody::predictMotion precedes the rigid solve) */
  double qdd[MAXDOF];
  aba(s, NULL, 1, qdd); minv_from_aba(s);
  for (int d = 0; d < n; d++) s->vel[d] = s->qd[d] + dt * qdd[d];
  for (int b = 0; b < s->nfree; b++) {
    int o = n + 6 * b; const double *v = s->fv[b], *w = s->fw[b];
    double kl = PARAM(m, AGX_P_LIN_DAMP), ka = PARAM(m, AGX_P_ANG_DAMP);
    double sl = kl + kl * sqrt(dot3(v, v)), sa = ka + ka * sqrt(dot3(w, w));
    double g[3] = {0, 0, FF(m, b, AGX_F_GRAVITY)};
    for (int k = 0; k < 3; k++) s->vel[o + k] = v[k] + dt * (g[k] - sl * v[k]);
    /* w' = w + dt * Iinv (-w x (I w)) - dt * sa * w  (world frame; damping torque = -I w sa) */
    double wl[3], Iwl[3], Iw[3], gy[3], acc[3];
    mtv3(s->freex[b].R, w, wl);
    for (int k = 0; k < 3; k++) Iwl[k] = FF(m, b, AGX_F_INERTIA + k) * wl[k];
    mv3(s->freex[b].R, Iwl, Iw); cross3(w, Iw, gy);
    double ng[3] = {-gy[0], -gy[1], -gy[2]}; mv3(s->fIinv[b], ng, acc);
    for (int k = 0; k < 3; k++) s->vel[o + 3 + k] = w[k] + dt * (acc[k] - sa * w[k]);
  }
  collide(s);
  if (PARAM(m, AGX_P_MANIFOLD) > 0) manifold_update(s);
  build_rows(s);
  double dv[NVMAX];
  pgs(s, dv);
  {
    const int fdirs = (int)PARAM(m, AGX_P_FRICTION_DIRS) == 2 ? 2 : 1;
    /* normal rows follow the non-contact rows in construction order */
    const int first_normal = s->nrows - (1 + fdirs) * s->ncon;
    for (int c = 0; c < s->ncon; c++) s->con[c].lambda_n = s->rows[first_normal + c].lambda;
    if (PARAM(m, AGX_P_WARMSTART) > 0) {
      g_warm_n = s->ncon < MAXC ? s->ncon : MAXC;
      for (int c = 0; c < g_warm_n; c++) {
        int ord = 0; for (int c2 = 0; c2 < c; c2++) if (s->con[c2].ca == s->con[c].ca && s->con[c2].cb == s->con[c].cb) ord++;
        g_warm_key[c][0] = s->con[c].ca; g_warm_key[c][1] = s->con[c].cb; g_warm_key[c][2] = ord; g_warm_lam[c] = s->con[c].lambda_n;
      }
    }
  }
  for (int d = 0; d < n; d++) {
    s->qd[d] = s->vel[d] + dv[d]; s->q[d] += dt * s->qd[d];
    /* Agent.enforce_joint_limits on the human after every stepSimulation (env.py:229, agent.py:240-250) */
    if (hooks && (RI(m, d, AGX_R_KIND) & 5) == 1 && !FROZEN(s, d)) {
      if (s->q[d] < dof_lower(s, d) - (double)AGX_LIMIT_EPS) { s->q[d] = dof_lower(s, d); s->qd[d] = 0; }
      else if (s->q[d] > dof_upper(s, d) + (double)AGX_LIMIT_EPS) { s->q[d] = dof_upper(s, d); s->qd[d] = 0; }
    }
  }
  for (int b = 0; b < s->nfree; b++) {
    int o = n + 6 * b;
    for (int k = 0; k < 3; k++) { s->fv[b][k] = s->vel[o + k] + dv[o + k]; s->fw[b][k] = s->vel[o + 3 + k] + dv[o + 3 + k]; s->fpos[b][k] += dt * s->fv[b][k]; }
    double* w = s->fw[b]; double th = sqrt(dot3(w, w)) * dt, dq[4];
    if (th > 1e-12) { double sc = sin(th / 2) / (th / dt); dq[0] = w[0] * sc; dq[1] = w[1] * sc; dq[2] = w[2] * sc; dq[3] = cos(th / 2); }
    else { dq[0] = w[0] * dt / 2; dq[1] = w[1] * dt / 2; dq[2] = w[2] * dt / 2; dq[3] = 1; }
    double qn[4]; quat_mul(dq, s->fquat[b], qn);
    double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int k = 0; k < 4; k++) s->fquat[b][k] = qn[k] / nn;
  }
  if (hooks) arm_limits(s);   /* env.py:230-231, after the limit reset above */
}
static void substep(sim_t* s) { substep_h(s, s->human_agent && !s->settling); }
/* one p.stepSimulation(): sim_sub internal substeps (numSubSteps, dressing.py:184), the hooks after the last one */
static void sim_step(sim_t* s) {
  s->anchor_set = 0;   /* DressingEnv.update_targets after the previous call moved the cloth's attachment to the end effector (dressing.py:200-210) */
  for (int k = 0; k < s->m->sim_sub; k++) substep_h(s, k == s->m->sim_sub - 1 && s->human_agent && !s->settling);
}

/* ------------------------------------------------------------------------------------ task layer */
static uint32_t rng_next(uint32_t* st) { /* 64-bit LCG in two words, xorshifted output */
  uint64_t x = ((uint64_t)st[1] << 32) | st[0];
  x = x * 6364136223846793005ULL + 1442695040888963407ULL;
  st[0] = (uint32_t)x; st[1] = (uint32_t)(x >> 32);
  uint32_t o = (uint32_t)(x >> 33) ^ (uint32_t)(x >> 11);
  return o;
}
static void to_base_frame(const sim_t* s, const double* p, const double* R, double* po, double* qo) {
  /* Agent.convert_to_realworld (agent.py:60-64): invertTransform(base) o (p, R) */
  /* the base of a robot on a floating base is a moving link (AGX_H_BASE_LINK) */
  const int bl = s->m->i[AGX_H_BASE_LINK];
  const xf_t* B = bl > 0 ? &s->link[bl - 1] : &s->base;
  double d[3]; sub3(p, B->p, d); mtv3(B->R, d, po);
  if (R && qo) { double Bt[9], Rr[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Bt[3 * r + c] = B->R[3 * c + r];
    mm3(Bt, R, Rr); mat_to_quat(Rr, qo); }
}
static void tool_base_pose_of(const sim_t* s, int tb, double* p, double* R);
static void tool_base_pose(const sim_t* s, double* p, double* R) { tool_base_pose_of(s, s->m->tool_body, p, R); }
static void tool_base_pose_of(const sim_t* s, int tb, double* p, double* R) {
  const agxo_model* m = s->m;
  double rp[3] = {FF(m, tb, AGX_F_REFPOS), FF(m, tb, AGX_F_REFPOS + 1), FF(m, tb, AGX_F_REFPOS + 2)};
  double rq[4] = {FF(m, tb, AGX_F_REFQUAT), FF(m, tb, AGX_F_REFQUAT + 1), FF(m, tb, AGX_F_REFQUAT + 2), FF(m, tb, AGX_F_REFQUAT + 3)}, Rr[9];
  quat_to_mat(rq, Rr); xf_apply(&s->freex[tb], rp, p); mm3(s->freex[tb].R, Rr, R);
  if (m->task_kind != AGX_TASK_FEEDING && m->task_kind != AGX_TASK_DRINKING) {   /* the frame the task reads: link 1 of the wiper (bed_bathing.py:81); drinking observes the cup's base frame (drinking.py:94) and applies its offset in the reward */
    double op[3] = {TF(m, AGX_T_TOOL_OBS_POS), TF(m, AGX_T_TOOL_OBS_POS + 1), TF(m, AGX_T_TOOL_OBS_POS + 2)};
    double oq[4] = {TF(m, AGX_T_TOOL_OBS_QUAT), TF(m, AGX_T_TOOL_OBS_QUAT + 1), TF(m, AGX_T_TOOL_OBS_QUAT + 2), TF(m, AGX_T_TOOL_OBS_QUAT + 3)}, Ro[9], R2[9], t[3];
    mv3(R, op, t); add3(p, t, p); quat_to_mat(oq, Ro); mm3(R, Ro, R2); memcpy(R, R2, sizeof R2);
  }
}
static void to_human_frame(const sim_t* s, const double* p, const double* R, double* po, double* qo) {
  /* human.convert_to_realworld: the human's base is collision body 0 (link -1) */
  const xf_t* B = &s->human[0];
  double d[3]; sub3(p, B->p, d); mtv3(B->R, d, po);
  if (R && qo) { double Bt[9], Rr[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Bt[3 * r + c] = B->R[3 * c + r];
    mm3(Bt, R, Rr); mat_to_quat(Rr, qo); }
}
/* FeedingEnv._get_obs (feeding.py:85-112): robot part, followed by the human part in co-op */
static void observe(sim_t* s, double robot_force, double tool_force, float* obs) {
  const agxo_model* m = s->m;
  double sp[3], sR[9], spr[3], sqr[4], hpr[3], hqr[4], tpr[3];
  tool_base_pose(s, sp, sR);
  to_base_frame(s, sp, sR, spr, sqr);
  int hl = TI(m, AGX_T_HEAD_LINK);
  to_base_frame(s, s->link[hl].p, s->link[hl].R, hpr, hqr);
  to_base_frame(s, s->target, NULL, tpr, NULL);
  int o = 0;
  for (int k = 0; k < 3; k++) obs[o++] = (float)spr[k];
  for (int k = 0; k < 4; k++) obs[o++] = (float)sqr[k];
  for (int k = 0; k < 3; k++) obs[o++] = (float)(spr[k] - tpr[k]);
  for (int d = 0; d < m->nrobot; d++) if (RI(m, d, AGX_R_ACT) >= 0 && !RI(m, d, AGX_R_OBS_SKIP)) {
    double a = s->q[d] + M_PI, w = a - 2 * M_PI * floor(a / (2 * M_PI)); obs[o++] = (float)(w - M_PI);
  }
  for (int k = 0; k < 3; k++) obs[o++] = (float)hpr[k];
  for (int k = 0; k < 4; k++) obs[o++] = (float)hqr[k];
  obs[o++] = (float)tool_force;
  if (s->coop) {   /* human_obs, feeding.py:102-108 */
    double sph[3], sqh[4], hph[3], hqh[4], tph[3];
    to_human_frame(s, sp, sR, sph, sqh);
    to_human_frame(s, s->link[hl].p, s->link[hl].R, hph, hqh);
    to_human_frame(s, s->target, NULL, tph, NULL);
    for (int k = 0; k < 3; k++) obs[o++] = (float)sph[k];
    for (int k = 0; k < 4; k++) obs[o++] = (float)sqh[k];
    for (int k = 0; k < 3; k++) obs[o++] = (float)(sph[k] - tph[k]);
    for (int d = m->nrobot; d < s->ndof; d++) if (RI(m, d, AGX_R_ACT) >= 0) obs[o++] = (float)s->q[d];
    for (int k = 0; k < 3; k++) obs[o++] = (float)hph[k];
    for (int k = 0; k < 4; k++) obs[o++] = (float)hqh[k];
    obs[o++] = (float)robot_force; obs[o++] = (float)tool_force;
  }
}
/* BedBathingEnv._get_obs (bed_bathing.py:80-110): robot part, followed by the human part in co-op */
static void observe_bed(sim_t* s, double tool_force, double total_force, double pad_force, float* obs) {
  const agxo_model* m = s->m;
  double sp[3], sR[9], spr[3], sqr[4];
  tool_base_pose(s, sp, sR);                 /* tool.get_pos_orient(1) */
  to_base_frame(s, sp, sR, spr, sqr);
  int o = 0;
  for (int k = 0; k < 3; k++) obs[o++] = (float)spr[k];
  for (int k = 0; k < 4; k++) obs[o++] = (float)sqr[k];
  for (int d = 0; d < m->nrobot; d++) if (RI(m, d, AGX_R_ACT) >= 0 && !RI(m, d, AGX_R_OBS_SKIP)) {
    double a = s->q[d] + M_PI, w = a - 2 * M_PI * floor(a / (2 * M_PI)); obs[o++] = (float)(w - M_PI);
  }
  for (int j = 0; j < 3; j++) {              /* shoulder, elbow, wrist positions (bed_bathing.py:89-94) */
    double pr[3]; to_base_frame(s, s->link[TI(m, AGX_T_OBS_LINK + j)].p, NULL, pr, NULL);
    for (int k = 0; k < 3; k++) obs[o++] = (float)pr[k];
  }
  obs[o++] = (float)tool_force;
  if (s->coop) {                             /* human_obs, bed_bathing.py:100-106 */
    double sph[3], sqh[4];
    to_human_frame(s, sp, sR, sph, sqh);
    for (int k = 0; k < 3; k++) obs[o++] = (float)sph[k];
    for (int k = 0; k < 4; k++) obs[o++] = (float)sqh[k];
    for (int d = m->nrobot; d < s->ndof; d++) if (RI(m, d, AGX_R_ACT) >= 0) obs[o++] = (float)s->q[d];
    for (int j = 0; j < 3; j++) {
      double ph[3]; to_human_frame(s, s->link[TI(m, AGX_T_OBS_LINK + j)].p, NULL, ph, NULL);
      for (int k = 0; k < 3; k++) obs[o++] = (float)ph[k];
    }
    obs[o++] = (float)total_force; obs[o++] = (float)pad_force;
  }
}
/* everything BedBathingEnv.step does after take_step (bed_bathing.py:15-39) */
static void finish_bed(sim_t* s, const float* action, float* obs, float* reward, int* done, float* info) {
  const agxo_model* m = s->m; double dt = m->dt;
  /* get_total_force (bed_bathing.py:41-78) */
  double robot_f = 0, tool_f = 0, tool_human_f = 0, pad_f = 0;
  for (int c = 0; c < s->ncon; c++) {
    const contact_t* k = &s->con[c];
    int ta = CI(m, k->ca, AGX_C_TAG), tb = CI(m, k->cb, AGX_C_TAG);
    int human = ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN, tool = ta == AGX_TAG_TOOL || tb == AGX_TAG_TOOL, robot = ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT;
    double f = k->lambda_n / dt;
    if (tool) tool_f += f;                                   /* :43 every contact of the tool */
    if (human && robot) robot_f += f;                        /* :42 */
    if (human && tool) { tool_human_f += f;                  /* :47-48 */
      int tc = ta == AGX_TAG_TOOL ? k->ca : k->cb; if (TI(m, AGX_T_PAD_LINK) >> (CI(m, tc, AGX_C_LINK) + 1) & 1) pad_f += f; }   /* :49-50 */
  }
  double total_f = robot_f + tool_human_f;
  observe_bed(s, tool_f, total_f, pad_f, obs);
  /* targets within TARGET_RADIUS of a manifold point of the pad on a link of the human (:52-74); world positions as
   * update_targets leaves them after the last substep (:190-203) */
  int g = s->gender, nt = TI(m, AGX_T_NT + 2 * g) + TI(m, AGX_T_NT + 2 * g + 1), new_points = 0;
  const float* TT = m->f + m->o_targets + 4 * g * TI(m, AGX_T_NT_MAX);
  double r2 = TF(m, AGX_T_TARGET_RADIUS) * TF(m, AGX_T_TARGET_RADIUS);
  for (int t = 0; t < nt; t++) {
    if (!(s->bb_alive[t >> 5] >> (t & 31) & 1)) continue;
    int link = TI(m, AGX_T_ARM_LINK + ((const int32_t*)TT)[4 * t + 3]);
    double tl[3] = {TT[4 * t], TT[4 * t + 1], TT[4 * t + 2]}, w[3]; xf_apply(&s->link[link], tl, w);
    int hit = 0;
    for (int q = 0; q < s->nqpt && !hit; q++) {
      if (s->qpt_link[q] < 0) continue;                      /* not the human's base link (:52) */
      double d[3]; sub3(s->qpt[q], w, d); if (dot3(d, d) < r2) hit = 1;
    }
    if (hit) { new_points++; s->bb_alive[t >> 5] &= ~(1u << (t & 31)); }
  }
  s->success += new_points;
  /* reward_distance = -min(closest distance tool <-> human within 5 m) (:23) */
  double dmin = TF(m, AGX_T_CLOSEST_DIST);
  {
    int t0 = -1, t1 = -1, h0 = -1, h1 = -1;
    for (int gg = 0; gg < m->ngroup; gg++) {
      int a0 = GI(m, gg, AGX_G_A0), b0 = GI(m, gg, AGX_G_B0);
      if (CI(m, a0, AGX_C_TAG) == AGX_TAG_TOOL && CI(m, b0, AGX_C_TAG) == AGX_TAG_HUMAN) {
        t0 = a0; t1 = GI(m, gg, AGX_G_A1); h0 = b0; h1 = GI(m, gg, AGX_G_B1);
        if (s->gender == 1 && GI(m, gg, AGX_G_B0F) >= 0) { h0 = GI(m, gg, AGX_G_B0F); h1 = GI(m, gg, AGX_G_B1F); }
        break;
      }
    }
    const double lim = dmin;
    for (int a = t0; a < t1; a++) for (int b = h0; b < h1; b++) {
      contact_t k; if (narrowphase(s, a, b, lim, &k) && k.dist < dmin) dmin = k.dist;
    }
  }
  double act_norm2 = 0; for (int k = 0; k < m->act_dim; k++) act_norm2 += (double)action[k] * action[k];
  xf_t ee; ee_frame(s, &ee);
  int L = TI(m, AGX_T_EE_LINK); double wxp[3], vee[3];
  cross3(s->vsp[L], ee.p, wxp); add3(s->vsp[L] + 3, wxp, vee);
  double ee_speed = sqrt(dot3(vee, vee));
  /* human_preferences (env.py:237-274), non-feeding branch (:244-247) */
  double pref = TF(m, AGX_T_C_V) * (-ee_speed) + TF(m, AGX_T_C_F) * (-(total_f - pad_f)) + TF(m, AGX_T_C_HF) * (pad_f < 10 ? 0.0 : -pad_f);
  double r = TF(m, AGX_T_W_DISTANCE) * (-dmin) + TF(m, AGX_T_W_ACTION) * (-sqrt(act_norm2)) + TF(m, AGX_T_W_WIPE) * new_points + pref;
  *reward = (float)r;
  *done = s->iteration >= (int)TF(m, AGX_T_EPISODE_LEN);
  if (info) {
    info[AGX_INFO_TOTAL_FORCE] = (float)total_f;
    info[AGX_INFO_TASK_SUCCESS] = (float)(s->success >= s->total_food * TF(m, AGX_T_SUCCESS_FRAC));
    info[AGX_INFO_ROBOT_FORCE] = (float)robot_f; info[AGX_INFO_TOOL_FORCE] = (float)pad_f;
    info[AGX_INFO_FOOD_REWARD] = (float)new_points; info[AGX_INFO_PREF] = (float)pref;
    info[AGX_INFO_NCONTACT] = (float)s->ncon; info[AGX_INFO_NROWS] = (float)s->nrows;
  }
}
/* ArmManipulationEnv._get_obs (arm_manipulation.py:71-110).  Single-arm robot: tool_left IS tool_right (:12-14), so the tool pose and
 * the tool forces appear twice, and the arm joints are listed twice (robot_arm = 'both', robot.py:16).  Two-armed robot: tool 0 is
 * tool_right, AGX_T_TOOL2_BODY is tool_left; the 14 joint angles are the right arm's, then the left arm's.
 * tf[2] = {tool_right_force, tool_left_force} (all contacts of the tool), thf[2] = the same on the human */
static void observe_arm(sim_t* s, const double* tf, double total_force, const double* thf, float* obs) {
  const agxo_model* m = s->m;
  const int dual = TI(m, AGX_T_TOOL2_BODY) > 0;
  double sp[2][3], sR[2][9], spr[2][3], sqr[2][4];
  for (int t = 0; t < 2; t++) {
    tool_base_pose_of(s, (t && dual) ? TI(m, AGX_T_TOOL2_BODY) : m->tool_body, sp[t], sR[t]);   /* tool.get_base_pos_orient() */
    to_base_frame(s, sp[t], sR[t], spr[t], sqr[t]);
  }
  const double* pts[5] = {s->link[TI(m, AGX_T_OBS_LINK)].p, s->link[TI(m, AGX_T_OBS_LINK + 1)].p, s->link[TI(m, AGX_T_OBS_LINK + 2)].p,
                          s->human[TI(m, AGX_T_STOMACH_BODY)].p, s->human[TI(m, AGX_T_WAIST_BODY)].p};     /* :85-89 */
  int o = 0;
  for (int t = 0; t < 2; t++) { for (int k = 0; k < 3; k++) obs[o++] = (float)spr[t][k]; for (int k = 0; k < 4; k++) obs[o++] = (float)sqr[t][k]; }   /* right, then left (:92) */
  for (int rep = 0; rep < (dual ? 1 : 2); rep++)
    for (int d = 0; d < m->nrobot; d++) if (RI(m, d, AGX_R_ACT) >= 0 && !RI(m, d, AGX_R_OBS_SKIP)) {
      double a = s->q[d] + M_PI, w = a - 2 * M_PI * floor(a / (2 * M_PI)); obs[o++] = (float)(w - M_PI);
    }
  for (int j = 0; j < 5; j++) { double pr[3]; to_base_frame(s, pts[j], NULL, pr, NULL); for (int k = 0; k < 3; k++) obs[o++] = (float)pr[k]; }
  obs[o++] = (float)tf[1]; obs[o++] = (float)tf[0];                     /* [tool_left_force, tool_right_force] (:92) */
  if (s->coop) {                             /* human_obs, :98-107 */
    for (int t = 0; t < 2; t++) {
      double sph[3], sqh[4];
      to_human_frame(s, sp[t], sR[t], sph, sqh);
      for (int k = 0; k < 3; k++) obs[o++] = (float)sph[k];
      for (int k = 0; k < 4; k++) obs[o++] = (float)sqh[k];
    }
    for (int d = m->nrobot; d < s->ndof; d++) if (RI(m, d, AGX_R_ACT) >= 0) obs[o++] = (float)s->q[d];
    for (int j = 0; j < 5; j++) { double ph[3]; to_human_frame(s, pts[j], NULL, ph, NULL); for (int k = 0; k < 3; k++) obs[o++] = (float)ph[k]; }
    obs[o++] = (float)total_force; obs[o++] = (float)thf[1]; obs[o++] = (float)thf[0];      /* total, tool_left_on_human, tool_right_on_human (:107) */
  }
}
/* everything ArmManipulationEnv.step does after take_step (arm_manipulation.py:18-60) */
static void finish_arm(sim_t* s, const float* action, float* obs, float* reward, int* done, float* info) {
  const agxo_model* m = s->m; double dt = m->dt;
  const int dual = TI(m, AGX_T_TOOL2_BODY) > 0, tb2 = AGX_BODY_FREE0 + TI(m, AGX_T_TOOL2_BODY);
  /* get_total_force (:62-69); the one tool of a single-arm robot is counted as tool_right and as tool_left */
  double robot_f = 0, tf[2] = {0, 0}, thf[2] = {0, 0};
  for (int c = 0; c < s->ncon; c++) {
    const contact_t* k = &s->con[c];
    int ta = CI(m, k->ca, AGX_C_TAG), tb = CI(m, k->cb, AGX_C_TAG);
    int human = ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN, robot = ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT;
    double f = k->lambda_n / dt;
    if (human && robot) robot_f += f;
    for (int side = 0; side < 2; side++) {                    /* a contact between the two tools counts for both */
      int tag = side ? tb : ta, body = side ? k->bb : k->ba;
      if (tag != AGX_TAG_TOOL) continue;
      int t = (dual && body == tb2) ? 1 : 0;
      tf[t] += f;
      if (human) thf[t] += f;
    }
  }
  if (!dual) { tf[1] = tf[0]; thf[1] = thf[0]; }
  double total_f = robot_f + thf[0] + thf[1];                 /* :68 */
  observe_arm(s, tf, total_f, thf, obs);
  /* tool.get_closest_points(human, distance=0.01): one point per (hull of the tool, shape of the human) pair that close, at the
   * poses after the last substep (env.py:264-265) */
  int near_pts[2] = {0, 0};
  {
    int t0 = -1, t1 = -1, h0 = -1, h1 = -1;
    for (int gg = 0; gg < m->ngroup; gg++) {
      int a0 = GI(m, gg, AGX_G_A0), b0 = GI(m, gg, AGX_G_B0);
      if (CI(m, a0, AGX_C_TAG) == AGX_TAG_TOOL && CI(m, b0, AGX_C_TAG) == AGX_TAG_HUMAN) {
        t0 = a0; t1 = GI(m, gg, AGX_G_A1); h0 = b0; h1 = GI(m, gg, AGX_G_B1);
        if (s->gender == 1 && GI(m, gg, AGX_G_B0F) >= 0) { h0 = GI(m, gg, AGX_G_B0F); h1 = GI(m, gg, AGX_G_B1F); }
        break;
      }
    }
    const double lim = TF(m, AGX_T_PRESSURE_DIST);
    for (int a = t0; a < t1; a++) for (int b = h0; b < h1; b++) {
      contact_t k; if (narrowphase(s, a, b, lim, &k) && k.dist < lim) near_pts[(dual && CI(m, a, AGX_C_BODY) == tb2) ? 1 : 0]++;
    }
    if (!dual) near_pts[1] = near_pts[0];
  }
  double act_norm2 = 0; for (int k = 0; k < m->act_dim; k++) act_norm2 += (double)action[k] * action[k];
  double ee_speed = 0;                                        /* right + left end effector (:26-27): the same link on a single-arm robot */
  for (int t = 0; t < 2; t++) {
    xf_t ee; ee_frame_of(s, dual ? t : 0, &ee);
    int L = TI(m, (dual && t) ? AGX_T_EE2_LINK : AGX_T_EE_LINK); double wxp[3], vee[3];
    cross3(s->vsp[L], ee.p, wxp); add3(s->vsp[L] + 3, wxp, vee);
    ee_speed += sqrt(dot3(vee, vee));
  }
  double pressure = 0;                                        /* env.py:266-269 */
  for (int t = 0; t < 2; t++) pressure += near_pts[t] <= 0 ? 0.0 : thf[t] / near_pts[t];
  /* human_preferences (env.py:237-274): reward_force_nontarget = -(total - (right + left)) = -robot_f, tool_force_at_target = 0 */
  double pref = TF(m, AGX_T_C_V) * (-ee_speed) + TF(m, AGX_T_C_F) * (-(total_f - thf[0] - thf[1])) + TF(m, AGX_T_C_P) * (-pressure);
  double spr[3], spl[3], sR[9], d[3];
  tool_base_pose_of(s, m->tool_body, spr, sR);
  tool_base_pose_of(s, dual ? TI(m, AGX_T_TOOL2_BODY) : m->tool_body, spl, sR);
  const double *elbow = s->link[TI(m, AGX_T_OBS_LINK + 1)].p, *wrist = s->link[TI(m, AGX_T_OBS_LINK + 2)].p;
  const double *stomach = s->human[TI(m, AGX_T_STOMACH_BODY)].p, *waist = s->human[TI(m, AGX_T_WAIST_BODY)].p;
  sub3(spl, elbow, d); double rd_left = -sqrt(dot3(d, d));                                      /* :36 */
  sub3(spr, wrist, d); double rd_right = -sqrt(dot3(d, d));                                     /* :37 */
  sub3(elbow, stomach, d); double rd_human = -sqrt(dot3(d, d));
  sub3(wrist, waist, d); rd_human -= sqrt(dot3(d, d));                                          /* :38 */
  double we = TF(m, AGX_T_W_WIPE);                                                              /* distance_end_effector_weight */
  double r = TF(m, AGX_T_W_DISTANCE) * rd_human + (dual ? we * rd_left + we * rd_right : 2 * we * rd_left) + TF(m, AGX_T_W_ACTION) * (-sqrt(act_norm2)) + pref;   /* :41-44 */
  if (s->am_best == 0 || rd_human > s->am_best) s->am_best = rd_human;                          /* :47-48 */
  *reward = (float)r;
  *done = s->iteration >= (int)TF(m, AGX_T_EPISODE_LEN);
  if (info) {
    info[AGX_INFO_TOTAL_FORCE] = (float)total_f;
    info[AGX_INFO_TASK_SUCCESS] = (float)((float)s->am_best >= TF(m, AGX_T_SUCCESS_FRAC));
    info[AGX_INFO_ROBOT_FORCE] = (float)robot_f; info[AGX_INFO_TOOL_FORCE] = (float)(dual ? thf[0] + thf[1] : thf[0]);
    info[AGX_INFO_FOOD_REWARD] = (float)(dual ? near_pts[0] + near_pts[1] : near_pts[0]); info[AGX_INFO_PREF] = (float)pref;
    info[AGX_INFO_NCONTACT] = (float)s->ncon; info[AGX_INFO_NROWS] = (float)s->nrows;
  }
}
/* scratch itch: world position of the target, limb frame o target_on_arm (scratch_itch.py:148-152) */
static void scratch_target(const sim_t* s, double* t) {
  const agxo_model* m = s->m; xf_apply(&s->link[TI(m, AGX_T_ARM_LINK + s->si_limb)], s->si_target, t);
}
/* ScratchItchEnv._get_obs (scratch_itch.py:59-91) */
static void observe_scratch(sim_t* s, double tool_force, double total_force, double target_force, float* obs) {
  const agxo_model* m = s->m;
  double sp[3], sR[9], spr[3], sqr[4], tg[3], tgr[3];
  tool_base_pose(s, sp, sR);                 /* tool.get_pos_orient(1) */
  to_base_frame(s, sp, sR, spr, sqr);
  scratch_target(s, tg); to_base_frame(s, tg, NULL, tgr, NULL);
  int o = 0;
  for (int k = 0; k < 3; k++) obs[o++] = (float)spr[k];
  for (int k = 0; k < 4; k++) obs[o++] = (float)sqr[k];
  for (int k = 0; k < 3; k++) obs[o++] = (float)(spr[k] - tgr[k]);
  for (int k = 0; k < 3; k++) obs[o++] = (float)tgr[k];
  for (int d = 0; d < m->nrobot; d++) if (RI(m, d, AGX_R_ACT) >= 0 && !RI(m, d, AGX_R_OBS_SKIP)) {
    double a = s->q[d] + M_PI, w = a - 2 * M_PI * floor(a / (2 * M_PI)); obs[o++] = (float)(w - M_PI);
  }
  for (int j = 0; j < 3; j++) {
    double pr[3]; to_base_frame(s, s->link[TI(m, AGX_T_OBS_LINK + j)].p, NULL, pr, NULL);
    for (int k = 0; k < 3; k++) obs[o++] = (float)pr[k];
  }
  obs[o++] = (float)tool_force;
  if (s->coop) {                             /* human_obs, scratch_itch.py:79-88 */
    double sph[3], sqh[4], tgh[3];
    to_human_frame(s, sp, sR, sph, sqh); to_human_frame(s, tg, NULL, tgh, NULL);
    for (int k = 0; k < 3; k++) obs[o++] = (float)sph[k];
    for (int k = 0; k < 4; k++) obs[o++] = (float)sqh[k];
    for (int k = 0; k < 3; k++) obs[o++] = (float)(sph[k] - tgh[k]);
    for (int k = 0; k < 3; k++) obs[o++] = (float)tgh[k];
    for (int d = m->nrobot; d < s->ndof; d++) if (RI(m, d, AGX_R_ACT) >= 0) obs[o++] = (float)s->q[d];
    for (int j = 0; j < 3; j++) {
      double ph[3]; to_human_frame(s, s->link[TI(m, AGX_T_OBS_LINK + j)].p, NULL, ph, NULL);
      for (int k = 0; k < 3; k++) obs[o++] = (float)ph[k];
    }
    obs[o++] = (float)total_force; obs[o++] = (float)target_force;
  }
}
/* everything ScratchItchEnv.step does after take_step (scratch_itch.py:14-44) */
static void finish_scratch(sim_t* s, const float* action, float* obs, float* reward, int* done, float* info) {
  const agxo_model* m = s->m; double dt = m->dt;
  double target[3]; scratch_target(s, target);
  double r2 = TF(m, AGX_T_TARGET_RADIUS) * TF(m, AGX_T_TARGET_RADIUS);
  /* get_total_force (scratch_itch.py:46-57) */
  double robot_f = 0, tool_f = 0, tool_human_f = 0, target_f = 0;
  for (int c = 0; c < s->ncon; c++) {
    const contact_t* k = &s->con[c];
    int ta = CI(m, k->ca, AGX_C_TAG), tb = CI(m, k->cb, AGX_C_TAG);
    int human = ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN, tool = ta == AGX_TAG_TOOL || tb == AGX_TAG_TOOL, robot = ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT;
    double f = k->lambda_n / dt;
    if (tool) tool_f += f;                                   /* :48 */
    if (human && robot) robot_f += f;                        /* :47 */
    if (human && tool) {
      tool_human_f += f;                                     /* :52 */
      int tool_is_a = ta == AGX_TAG_TOOL, tc = tool_is_a ? k->ca : k->cb;
      const double* on_human = tool_is_a ? k->pb : k->pa; double d[3]; sub3(on_human, target, d);
      if ((TI(m, AGX_T_PAD_LINK) >> (CI(m, tc, AGX_C_LINK) + 1) & 1) && dot3(d, d) < r2) target_f += f;   /* :54-55 */
    }
  }
  double total_f = robot_f + tool_human_f;
  observe_scratch(s, tool_f, total_f, target_f, obs);
  /* target_contact_pos = posB of the last manifold point of tool links 0 / 1 on the human near the target (:54-56) */
  int have = 0; double cp[3] = {0, 0, 0};
  for (int q = 0; q < s->nqpt; q++) { double d[3]; sub3(s->qpt[q], target, d); if (dot3(d, d) < r2) { have = 1; memcpy(cp, s->qpt[q], 24); } }
  double scratch_reward = 0;
  if (have) {
    double dp[3]; sub3(cp, s->si_prev, dp);
    if (sqrt(dot3(dp, dp)) > 0.01 && target_f < 10) { scratch_reward = 5; memcpy(s->si_prev, cp, 24); s->success += 1; }   /* :28-32 */
  }
  double act_norm2 = 0; for (int k = 0; k < m->act_dim; k++) act_norm2 += (double)action[k] * action[k];
  double sp[3], sR[9], dd[3]; tool_base_pose(s, sp, sR); sub3(target, sp, dd);      /* :25-26 tool.get_pos_orient(1) */
  xf_t ee; ee_frame(s, &ee);
  int L = TI(m, AGX_T_EE_LINK); double wxp[3], vee[3];
  cross3(s->vsp[L], ee.p, wxp); add3(s->vsp[L] + 3, wxp, vee);
  double ee_speed = sqrt(dot3(vee, vee));
  double pref = TF(m, AGX_T_C_V) * (-ee_speed) + TF(m, AGX_T_C_F) * (-(total_f - target_f)) + TF(m, AGX_T_C_HF) * (target_f < 10 ? 0.0 : -target_f);
  double r = TF(m, AGX_T_W_DISTANCE) * (-sqrt(dot3(dd, dd))) + TF(m, AGX_T_W_ACTION) * (-sqrt(act_norm2)) + TF(m, AGX_T_W_WIPE) * scratch_reward + pref;
  *reward = (float)r;
  *done = s->iteration >= (int)TF(m, AGX_T_EPISODE_LEN);
  if (info) {
    info[AGX_INFO_TOTAL_FORCE] = (float)total_f;
    info[AGX_INFO_TASK_SUCCESS] = (float)(s->success >= s->total_food * TF(m, AGX_T_SUCCESS_FRAC));
    info[AGX_INFO_ROBOT_FORCE] = (float)robot_f; info[AGX_INFO_TOOL_FORCE] = (float)target_f;
    info[AGX_INFO_FOOD_REWARD] = (float)scratch_reward; info[AGX_INFO_PREF] = (float)pref;
    info[AGX_INFO_NCONTACT] = (float)s->ncon; info[AGX_INFO_NROWS] = (float)s->nrows;
  }
}

/* ---- dressing (assistive_gym/envs/dressing.py) ---- */
/* _get_obs (dressing.py:78-110) */
static void observe_dressing(sim_t* s, double cloth_force_sum, double robot_force, float* obs) {
  const agxo_model* m = s->m;
  xf_t ee; ee_frame(s, &ee);
  double pr[3], qr[4]; to_base_frame(s, ee.p, ee.R, pr, qr);
  int o = 0;
  for (int k = 0; k < 3; k++) obs[o++] = (float)pr[k];
  for (int k = 0; k < 4; k++) obs[o++] = (float)qr[k];
  for (int d = 0; d < m->nrobot; d++) if (RI(m, d, AGX_R_ACT) >= 0 && !RI(m, d, AGX_R_OBS_SKIP)) {
    double a = s->q[d] + M_PI, w = a - 2 * M_PI * floor(a / (2 * M_PI)); obs[o++] = (float)(w - M_PI);   /* :84 */
  }
  for (int j = 0; j < 3; j++) {
    double p[3]; to_base_frame(s, s->link[TI(m, AGX_T_OBS_LINK + j)].p, NULL, p, NULL);
    for (int k = 0; k < 3; k++) obs[o++] = (float)p[k];
  }
  obs[o++] = (float)cloth_force_sum;
  if (s->coop) {                             /* human_obs, :100-106 */
    double ph[3], qh[4]; to_human_frame(s, ee.p, ee.R, ph, qh);
    for (int k = 0; k < 3; k++) obs[o++] = (float)ph[k];
    for (int k = 0; k < 4; k++) obs[o++] = (float)qh[k];
    for (int d = m->nrobot; d < s->ndof; d++) if (RI(m, d, AGX_R_ACT) >= 0) obs[o++] = (float)s->q[d];
    for (int j = 0; j < 3; j++) {
      double p[3]; to_human_frame(s, s->link[TI(m, AGX_T_OBS_LINK + j)].p, NULL, p, NULL);
      for (int k = 0; k < 3; k++) obs[o++] = (float)p[k];
    }
    obs[o++] = (float)cloth_force_sum; obs[o++] = (float)robot_force;
  }
}
static double signed_volume(const double* a, const double* b, const double* c, const double* d) {
  double ba[3], ca[3], da[3], cr[3]; sub3(b, a, ba); sub3(c, a, ca); sub3(d, a, da); cross3(ba, ca, cr); return dot3(cr, da) / 6.0;
}
static int sgn(double x) { return (x > 0) - (x < 0); }   /* np.sign */
/* Util.line_intersects_triangle (util.py:125-132) */
static int line_intersects_triangle(const double* p0, const double* p1, const double* p2, const double* q0, const double* q1) {
  if (sgn(signed_volume(q0, p0, p1, p2)) != sgn(signed_volume(q1, p0, p1, p2))) {
    int a = sgn(signed_volume(q0, q1, p0, p1)), b = sgn(signed_volume(q0, q1, p1, p2)), c = sgn(signed_volume(q0, q1, p2, p0));
    if (a == b && b == c) return 1;
  }
  return 0;
}
/* does the point set straddle both planes through `origin` spanned by the arm axis and one of two perpendiculars (util.py:144-160)? */
static int points_around_axis(const double (*pts)[3], int n, const double* axis_from, const double* axis_to, const double* origin) {
  double nrm[3], tan[3], bin[3], c110[3] = {1, 1, 0};
  sub3(axis_to, axis_from, nrm); double l = sqrt(dot3(nrm, nrm)); for (int k = 0; k < 3; k++) nrm[k] /= l;
  cross3(c110, nrm, tan); l = sqrt(dot3(tan, tan)); for (int k = 0; k < 3; k++) tan[k] /= l;
  cross3(tan, nrm, bin); l = sqrt(dot3(bin, bin)); for (int k = 0; k < 3; k++) bin[k] /= l;
  int tp = 0, tn = 0, bp = 0, bn = 0;
  for (int i = 0; i < n; i++) { double d[3]; sub3(pts[i], origin, d); double t = dot3(tan, d), b = dot3(bin, d); tp |= t > 0; tn |= t < 0; bp |= b > 0; bn |= b < 0; }
  return tp && tn && bp && bn;
}
/* everything DressingEnv.step does after take_step (dressing.py:20-76) */
static void finish_dressing(sim_t* s, const float* action, float* obs, float* reward, int* done, float* info) {
  const agxo_model* m = s->m; const double dt = m->dt;
  const double* shoulder = s->link[TI(m, AGX_T_OBS_LINK)].p; const double* elbow = s->link[TI(m, AGX_T_OBS_LINK + 1)].p; const double* wrist = s->link[TI(m, AGX_T_OBS_LINK + 2)].p;
  double pts[6][3] = {{0}};
  if (s->cx) for (int k = 0; k < 6; k++) memcpy(pts[k], s->cx + 3 * m->i[m->o_cloth + AGX_CL_TRI + k], 24);
  /* Util.sleeve_on_arm_reward (util.py:134-202) */
  const double rad = TF(m, AGX_T_ARM_RADIUS + s->gender);     /* hand_radius = elbow_radius = shoulder_radius */
  double we[3], es[3]; sub3(wrist, elbow, we); sub3(shoulder, elbow, es);
  const double lwe = sqrt(dot3(we, we)), les = sqrt(dot3(es, es));
  double hand_end[3], elbow_end[3], shoulder_end[3];
  for (int k = 0; k < 3; k++) { hand_end[k] = wrist[k] + we[k] / lwe * rad * 2; elbow_end[k] = elbow[k] - we[k] / lwe * rad; shoulder_end[k] = shoulder[k] + es[k] / les * rad; }
  const int around_fore = points_around_axis(pts, 6, elbow_end, hand_end, hand_end);              /* :144-160 */
  const int around_upper = points_around_axis(pts, 6, shoulder_end, elbow_end, shoulder_end);    /* :162-171 */
  const int f1 = line_intersects_triangle(pts[0], pts[1], pts[2], hand_end, elbow_end), f2 = line_intersects_triangle(pts[3], pts[4], pts[5], hand_end, elbow_end);
  const int u1 = line_intersects_triangle(pts[0], pts[1], pts[2], elbow_end, shoulder_end), u2 = line_intersects_triangle(pts[3], pts[4], pts[5], elbow_end, shoulder_end);
  double centre[3] = {0, 0, 0}; for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) centre[k] += pts[i][k] / 6.0;
  double d[3];
  sub3(hand_end, centre, d); const double distance_to_hand = sqrt(dot3(d, d)), distance_along_forearm = distance_to_hand;   /* :181,189 */
  sub3(centre, elbow, d); const double distance_along_upperarm = sqrt(dot3(d, d));                                          /* :190 */
  sub3(hand_end, elbow_end, d); const double forearm_length = sqrt(dot3(d, d));
  const double upperarm_length = les;                                                                                        /* |elbow_pos - shoulder_pos| */
  const int forearm_in = around_fore && (f1 || f2), upperarm_in = around_upper && (u1 || u2);
  /* cloth forces (:34-46): x 10, only below the end effector and below 20 */
  xf_t ee; ee_frame(s, &ee);
  double cloth_force_sum = 0;
  for (int c = 0; c < s->nccon; c++) {
    const double* k = s->ccon + 6 * c; double f[3] = {k[3] * CLPAR(m, AGX_CP_FORCE_SCALE), k[4] * CLPAR(m, AGX_CP_FORCE_SCALE), k[5] * CLPAR(m, AGX_CP_FORCE_SCALE)};
    const double fn = sqrt(dot3(f, f));
    if (k[2] < ee.p[2] - CLPAR(m, AGX_CP_EE_BELOW) && fn < CLPAR(m, AGX_CP_FORCE_MAX)) cloth_force_sum += fn;
  }
  int L = TI(m, AGX_T_EE_LINK); double wxp[3], vee[3];
  cross3(s->vsp[L], ee.p, wxp); add3(s->vsp[L] + 3, wxp, vee);
  const double ee_speed = sqrt(dot3(vee, vee));
  /* human_preferences(end_effector_velocity, dressing_forces) (env.py:237-274): the force terms see their defaults, 0 */
  const double pref = TF(m, AGX_T_C_V) * (-ee_speed) + TF(m, AGX_T_C_D) * (-cloth_force_sum);
  double act_norm2 = 0; for (int k = 0; k < m->act_dim; k++) act_norm2 += (double)action[k] * action[k];
  double reward_dressing;
  if (upperarm_in) { reward_dressing = forearm_length; if (distance_along_upperarm < upperarm_length) reward_dressing += distance_along_upperarm; }
  else if (forearm_in && distance_along_forearm < forearm_length) reward_dressing = distance_along_forearm;
  else reward_dressing = -distance_to_hand;
  const double r = TF(m, AGX_T_W_WIPE) * reward_dressing + TF(m, AGX_T_W_ACTION) * (-sqrt(act_norm2)) + pref;
  /* _get_obs (:95-96): robot_force_on_human from the contacts of the last substep */
  double robot_f = 0;
  for (int c = 0; c < s->ncon; c++) {
    const contact_t* k = &s->con[c]; int ta = CI(m, k->ca, AGX_C_TAG), tb = CI(m, k->cb, AGX_C_TAG);
    if ((ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN) && (ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT)) robot_f += k->lambda_n / dt;
  }
  s->dr_force_sum = cloth_force_sum;
  observe_dressing(s, cloth_force_sum, robot_f, obs);
  if (reward_dressing > s->dr_best) s->dr_best = reward_dressing;                 /* :62-63 */
  *reward = (float)r;
  *done = s->iteration >= (int)TF(m, AGX_T_EPISODE_LEN);
  if (info) {
    info[AGX_INFO_TOTAL_FORCE] = (float)(robot_f + cloth_force_sum);
    info[AGX_INFO_TASK_SUCCESS] = (float)(s->dr_best >= TF(m, AGX_T_SUCCESS_FRAC));
    info[AGX_INFO_ROBOT_FORCE] = (float)robot_f; info[AGX_INFO_TOOL_FORCE] = (float)cloth_force_sum;
    info[AGX_INFO_FOOD_REWARD] = (float)reward_dressing; info[AGX_INFO_PREF] = (float)pref;
    info[AGX_INFO_NCONTACT] = (float)s->ncon; info[AGX_INFO_NROWS] = (float)s->nrows;
  }
}
static void contact_forces(const sim_t* s, double* robot_f, double* tool_f, int* food_hit_mask) {
  const agxo_model* m = s->m; double dt = m->dt;
  *robot_f = 0; *tool_f = 0; *food_hit_mask = s->food_near_human;
  for (int c = 0; c < s->ncon; c++) {
    const contact_t* k = &s->con[c];
    int ta = CI(m, k->ca, AGX_C_TAG), tb = CI(m, k->cb, AGX_C_TAG);
    if (tb == AGX_TAG_HUMAN || ta == AGX_TAG_HUMAN) {
      int other = ta == AGX_TAG_HUMAN ? tb : ta;
      if (other == AGX_TAG_ROBOT) *robot_f += k->lambda_n / dt;     /* feeding.py:46 */
      if (other == AGX_TAG_TOOL) *tool_f += k->lambda_n / dt;       /* feeding.py:47 */
    }
  }
}

/* DrinkingEnv.step after take_step (drinking.py:12-50): observation, get_water_rewards (:52-91), the cup's distance / tilt terms, preferences */
static void finish_drinking(sim_t* s, double act_norm2, float* obs, float* reward, int* done, float* info) {
  const agxo_model* m = s->m;
  update_target(s);
  double robot_f, tool_f; int hm; contact_forces(s, &robot_f, &tool_f, &hm);
  observe(s, robot_f, tool_f, obs);                       /* DrinkingEnv._get_obs = FeedingEnv._get_obs with the cup (drinking.py:93-121) */
  const double total_f = robot_f + tool_f;
  /* the frame the reward reads the cup in: base frame o ([0, 0.06, 0], rpy (pi/2, 0, 0)) (drinking.py:24,56), top and bottom centres in it */
  double cp[3], cR[9], p2[3], R2[9], Ro[9], t[3], top[3], bot[3];
  tool_base_pose(s, cp, cR);
  const double op[3] = {TF(m, AGX_T_TOOL_OBS_POS), TF(m, AGX_T_TOOL_OBS_POS + 1), TF(m, AGX_T_TOOL_OBS_POS + 2)};
  const double oq[4] = {TF(m, AGX_T_TOOL_OBS_QUAT), TF(m, AGX_T_TOOL_OBS_QUAT + 1), TF(m, AGX_T_TOOL_OBS_QUAT + 2), TF(m, AGX_T_TOOL_OBS_QUAT + 3)};
  mv3(cR, op, t); add3(cp, t, p2); quat_to_mat(oq, Ro); mm3(cR, Ro, R2);
  const double to[3] = {TF(m, AGX_T_DK_TOP), TF(m, AGX_T_DK_TOP + 1), TF(m, AGX_T_DK_TOP + 2)}, bo[3] = {TF(m, AGX_T_DK_BOTTOM), TF(m, AGX_T_DK_BOTTOM + 1), TF(m, AGX_T_DK_BOTTOM + 2)};
  mv3(R2, to, t); add3(p2, t, top); mv3(R2, bo, t); add3(p2, t, bot);
  double water_reward = 0, water_hit = 0, vel_sum = 0;
  const int NN = s->cx ? agxo_cloth_nodes(m) : 0;
  const uint64_t active_on_entry = s->dk_active;
  double axis[3]; sub3(bot, top, axis); const double cyl = TF(m, AGX_T_TARGET_RADIUS) * sqrt(dot3(axis, axis));
  for (int k = 0; k < NN; k++) {
    if (!(s->dk_alive >> k & 1)) continue;
    const double* x = s->cx + 3 * k; double a[3], b[3], cr[3];
    sub3(x, top, a); sub3(x, bot, b); cross3(a, axis, cr);
    const int inside = dot3(a, axis) >= 0 && dot3(b, axis) <= 0 && sqrt(dot3(cr, cr)) <= cyl;       /* Util.points_in_cylinder (util.py:53-56) */
    if (inside) continue;
    double d[3]; sub3(s->target, x, d);
    if (sqrt(dot3(d, d)) < TF(m, AGX_T_MOUTH_DIST)) {                                               /* in the mouth (drinking.py:66-75) */
      water_reward += 10; s->success += 1; vel_sum += sqrt(dot3(s->cv + 3 * k, s->cv + 3 * k));
      s->dk_alive &= ~(1ull << k); s->dk_active &= ~(1ull << k);
      for (int q = 0; q < 3; q++) s->cx[3 * k + q] = 1000.0 + 1000.0 * (rng_next(s->rng) >> 8) * (1.0 / 16777216.0);
      continue;
    }
    /* w.get_closest_points(self.tool, distance=0.1) empty -> spilled (drinking.py:77-80): the particle against the cup's pieces */
    const int near = particle_near_tool(s, x, TF(m, AGX_T_SPILL_DIST));
    if (!near) { water_reward -= 1; s->dk_alive &= ~(1ull << k); }
  }
  /* waters_active as it was on entry (drinking.py:81-85; the list is only filtered after both loops): a particle that touches the person */
  uint64_t counted = 0;
  for (int c = 0; c < s->nccon; c++) {
    const int k = s->ccon_node[c], col = m->i[m->o_cloth + m->i[m->o_cloth + AGX_CL_OFF_SHAPE] + 4 * s->ccon_shape[c]];
    if (CI(m, col, AGX_C_TAG) == AGX_TAG_HUMAN && (active_on_entry >> k & 1) && !(counted >> k & 1)) { water_hit -= 1; counted |= 1ull << k; s->dk_active &= ~(1ull << k); }
  }
  xf_t ee; ee_frame(s, &ee);
  int L = TI(m, AGX_T_EE_LINK); double wxp[3], vee[3];
  cross3(s->vsp[L], ee.p, wxp); add3(s->vsp[L] + 3, wxp, vee);
  const double ee_speed = sqrt(dot3(vee, vee));
  const double pref = TF(m, AGX_T_C_V) * (-ee_speed) + TF(m, AGX_T_C_F) * (-total_f) + TF(m, AGX_T_C_HF) * (tool_f < 10 ? 0.0 : -tool_f)
                    + TF(m, AGX_T_C_FD) * water_hit + TF(m, AGX_T_C_FDV) * (-vel_sum);              /* env.py:249-256, the feeding / drinking branch */
  double dd[3]; sub3(s->target, top, dd);
  /* cup_euler[0]: roll of the offset frame, btQuaternion::getEulerZYX as p.getEulerFromQuaternion returns it (drinking.py:30-31) */
  double q[4]; mat_to_quat(R2, q);
  const double sarg = -2.0 * (q[0] * q[2] - q[3] * q[1]);
  const double roll = (sarg <= -0.99999 || sarg >= 0.99999) ? 0.0 : atan2(2 * (q[1] * q[2] + q[3] * q[0]), q[3] * q[3] - q[0] * q[0] - q[1] * q[1] + q[2] * q[2]);
  const double r = TF(m, AGX_T_W_DISTANCE) * (-sqrt(dot3(dd, dd))) + TF(m, AGX_T_W_ACTION) * (-sqrt(act_norm2)) + TF(m, AGX_T_W_TILT) * (-fabs(roll - M_PI / 2))
                 + TF(m, AGX_T_W_FOOD) * water_reward + pref;
  *reward = (float)r;
  *done = s->iteration >= (int)TF(m, AGX_T_EPISODE_LEN);
  if (info) {
    info[AGX_INFO_TOTAL_FORCE] = (float)total_f;
    info[AGX_INFO_TASK_SUCCESS] = (float)(s->success >= s->total_food * TF(m, AGX_T_SUCCESS_FRAC));
    info[AGX_INFO_ROBOT_FORCE] = (float)robot_f; info[AGX_INFO_TOOL_FORCE] = (float)tool_f;
    info[AGX_INFO_FOOD_REWARD] = (float)water_reward; info[AGX_INFO_PREF] = (float)pref;
    info[AGX_INFO_NCONTACT] = (float)s->ncon; info[AGX_INFO_NROWS] = (float)s->nrows;
  }
}

void agxo_observe(const agxo_model* m, const float* state, float* obs) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s); update_target(s);
  if (m->task_kind == AGX_TASK_BED_BATHING) observe_bed(s, 0, 0, 0, obs);
  else if (m->task_kind == AGX_TASK_SCRATCH_ITCH) observe_scratch(s, 0, 0, 0, obs);
  else if (m->task_kind == AGX_TASK_DRESSING) observe_dressing(s, s->dr_force_sum, 0, obs);
  else if (m->task_kind == AGX_TASK_ARM_MANIPULATION) { const double z[2] = {0, 0}; observe_arm(s, z, 0, z, obs); }
  else observe(s, 0, 0, obs);
  free(s);
}

/* the cloth travels next to the state record: float[2][NN][3], node positions then node velocities */
int agxo_cloth_nodes(const agxo_model* m) { return m->o_cloth ? m->i[m->o_cloth + AGX_CL_NN] : 0; }
static void cloth_attach(sim_t* s, const float* cloth) {
  if (!cloth || !s->m->o_cloth) return;
  const int n3 = 3 * agxo_cloth_nodes(s->m);
  s->cx = (double*)malloc(sizeof(double) * n3); s->cv = (double*)malloc(sizeof(double) * n3); s->cq = (double*)malloc(sizeof(double) * n3);
  const int per = CLOTH_NODE_CONTACTS > 12 ? CLOTH_NODE_CONTACTS : 12;    /* (water: WATER_CONTACTS candidates per particle) */
  s->ccon = (double*)malloc(sizeof(double) * 6 * per * (n3 / 3));
  s->ccon_node = (int*)malloc(sizeof(int) * per * (n3 / 3)); s->ccon_shape = (int*)malloc(sizeof(int) * per * (n3 / 3));
  for (int k = 0; k < n3; k++) { s->cx[k] = cloth[k]; s->cv[k] = cloth[n3 + k]; }
}
static void cloth_detach(sim_t* s, float* cloth) {
  if (!s->cx) return;
  const int n3 = 3 * agxo_cloth_nodes(s->m);
  for (int k = 0; k < n3; k++) { cloth[k] = (float)s->cx[k]; cloth[n3 + k] = (float)s->cv[k]; }
  free(s->cx); free(s->cv); free(s->cq); free(s->ccon); free(s->ccon_node); free(s->ccon_shape); s->cx = NULL;
}
void agxo_settle_cloth(const agxo_model* m, float* state, float* cloth, int n_sim_steps) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state);
  s->rows = (row_t*)malloc(sizeof(row_t) * MAXROWS);
  cloth_attach(s, cloth);
  s->settling = 1;
  for (int k = 0; k < n_sim_steps; k++) sim_step(s);
  kinematics(s); update_target(s);
  cloth_detach(s, cloth);
  sim_store(s, state); free(s->rows); free(s);
}
void agxo_settle(const agxo_model* m, float* state, int n_substeps) { agxo_settle_cloth(m, state, NULL, n_substeps); }
/* contacts of the cloth with the rigid shapes in the last substep of the last call: {x, y, z, fx, fy, fz} each (tests) */
static double g_ccon[6 * 4096]; static int g_ccon_node[4096]; static int g_nccon = 0;
int agxo_cloth_contacts(double* out, int max_out) { int n = g_nccon < max_out ? g_nccon : max_out; memcpy(out, g_ccon, sizeof(double) * 6 * n); return g_nccon; }
/* ... and the garment node of each of them (tests/test_gpu_bench_size.py: which node contacts are in the cloth-force sum on either side) */
int agxo_cloth_contact_nodes(int* out, int max_out) { int n = g_nccon < max_out ? g_nccon : max_out; memcpy(out, g_ccon_node, sizeof(int) * n); return g_nccon; }

void agxo_step(const agxo_model* m, float* state, const float* action, float* obs, float* reward, int* done, float* info) {
  agxo_step_cloth(m, state, NULL, action, obs, reward, done, info);
}
void agxo_step_cloth(const agxo_model* m, float* state, float* cloth, const float* action, float* obs, float* reward, int* done, float* info) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state);
  s->rows = (row_t*)malloc(sizeof(row_t) * MAXROWS);
  cloth_attach(s, cloth);
  int nsub = (int)PARAM(m, AGX_P_FRAME_SKIP);
  /* take_step (env.py:174-222): clip, scale (float32 arithmetic as numpy does for a float32 action),
   * 5x accumulate with per-joint limit clamp, set motor targets */
  s->iteration += 1;
  double act_norm2 = 0;
  int tremor_on = 0;
  for (int k = 0; k < m->nhdof; k++) if (s->tremor[k] != 0) tremor_on = 1;          /* impairment == 'tremor' */
  const double tsign = (s->iteration % 2 == 0) ? 1.0 : -1.0;
  for (int d = 0; d < s->ndof; d++) {
    int ai = RI(m, d, AGX_R_ACT); if (ai < 0) continue;
    const int is_human = d >= m->nrobot;
    if (is_human && !s->coop) continue;                /* the human only takes actions when controllable */
    float a32 = action[ai]; if (a32 < -1.0f) a32 = -1.0f; if (a32 > 1.0f) a32 = 1.0f;
    a32 *= (float)PARAM(m, AGX_P_ACTION_SCALE);
    /* Robot.action_multiplier (env.py:196-197) and Robot.action_duplication (env.py:218-220): see AGX_R_ACT_MULT / AGX_R_ACT_SRC */
    const float mult = (float)RF(m, d, AGX_R_ACT_MULT); if (mult != 0.0f) a32 *= mult;
    const int ds = RI(m, d, AGX_R_ACT_SRC) > 0 ? RI(m, d, AGX_R_ACT_SRC) - 1 : d;
    double a = a32, qa = s->q[ds], lo = dof_lower(s, ds), hi = dof_upper(s, ds);
    const int k2 = d - m->nrobot;
    for (int k = 0; k < nsub; k++) {
      int below = qa + a < lo, above = qa + a > hi;
      if (below || above) a = 0;
      if (below) qa = lo; if (above) qa = hi;
      if (is_human && tremor_on) { s->tremor_target[k2] += a; qa = s->tremor_target[k2] + s->tremor[k2] * tsign; }   /* env.py:212-215 */
      else qa += a;
    }
    s->qt[d] = qa;
  }
  /* tremor without control (env.py:212-215): target + tremors * (+1 on even iterations, -1 on odd) */
  if (!s->coop) for (int k = 0; k < m->nhdof; k++) s->qt[m->nrobot + k] = s->tremor_target[k] + s->tremor[k] * tsign;
  for (int k = 0; k < m->act_dim; k++) act_norm2 += (double)action[k] * action[k];
  for (int k = 0; k < nsub; k++) sim_step(s);
  kinematics(s); /* poses after the last integration, as the getters in _get_obs see them */
  if (m->task_kind == AGX_TASK_DRESSING) {
    finish_dressing(s, action, obs, reward, done, info);
    g_nccon = s->cx ? (s->nccon < 4096 ? s->nccon : 4096) : 0; if (g_nccon) { memcpy(g_ccon, s->ccon, sizeof(double) * 6 * g_nccon); memcpy(g_ccon_node, s->ccon_node, sizeof(int) * g_nccon); }
    cloth_detach(s, cloth); sim_store(s, state); free(s->rows); free(s); return;
  }
  if (m->task_kind == AGX_TASK_DRINKING) { finish_drinking(s, act_norm2, obs, reward, done, info); cloth_detach(s, cloth); sim_store(s, state); free(s->rows); free(s); return; }
  if (m->task_kind == AGX_TASK_BED_BATHING) { finish_bed(s, action, obs, reward, done, info); sim_store(s, state); free(s->rows); free(s); return; }
  if (m->task_kind == AGX_TASK_SCRATCH_ITCH) { finish_scratch(s, action, obs, reward, done, info); sim_store(s, state); free(s->rows); free(s); return; }
  if (m->task_kind == AGX_TASK_ARM_MANIPULATION) { finish_arm(s, action, obs, reward, done, info); sim_store(s, state); free(s->rows); free(s); return; }
  update_target(s); /* FeedingEnv.update_targets (feeding.py:192-196) */
  double robot_f, tool_f; int hit_mask;
  contact_forces(s, &robot_f, &tool_f, &hit_mask);
  observe(s, robot_f, tool_f, obs);
  double total_f = robot_f + tool_f;
  /* get_food_rewards (feeding.py:50-83) */
  double food_reward = 0, food_hit = 0, vel_sum = 0;
  /* the second loop of get_food_rewards walks foods_active as it was on entry (feeding.py:74-82):
   * a particle eaten in the first loop is still in it, and getContactPoints still returns the
   * contacts of the last stepSimulation */
  const int active_on_entry = s->active;
  for (int k = 0; k < m->nfood; k++) {
    if (!(s->alive >> k & 1)) continue;
    int b = m->food0 + k; double d[3]; sub3(s->target, s->fpos[b], d);
    if (sqrt(dot3(d, d)) < TF(m, AGX_T_MOUTH_DIST)) {
      food_reward += 20; s->success += 1; vel_sum += sqrt(dot3(s->fv[b], s->fv[b]));
      s->alive &= ~(1 << k); s->active &= ~(1 << k);
      for (int q = 0; q < 3; q++) s->fpos[b][q] = 1000.0 + 1000.0 * (rng_next(s->rng) >> 8) * (1.0 / 16777216.0);
      s->fquat[b][0] = s->fquat[b][1] = s->fquat[b][2] = 0; s->fquat[b][3] = 1;
      continue;
    }
    /* getClosestPoints(food, tool, distance=0.1) empty?  (agent.py:118-130) */
    int near = 0;
    {
      /* find this particle's collider and test it against every tool collider */
      int fc = -1;
      for (int c = 0; c < m->ncoll; c++) if (CI(m, c, AGX_C_BODY) == AGX_BODY_FREE0 + b) { fc = c; break; }
      for (int c = 0; c < m->ncoll && !near; c++) {
        if (CI(m, c, AGX_C_TAG) != AGX_TAG_TOOL) continue;
        double lo1[3], hi1[3], lo2[3], hi2[3]; int sep = 0;
        collider_aabb(s, fc, lo1, hi1); collider_aabb(s, c, lo2, hi2);
        for (int q = 0; q < 3; q++) if (lo1[q] > hi2[q] + TF(m, AGX_T_SPILL_DIST) || lo2[q] > hi1[q] + TF(m, AGX_T_SPILL_DIST)) sep = 1;
        if (sep) continue;
        contact_t tmp; if (narrowphase(s, fc, c, TF(m, AGX_T_SPILL_DIST), &tmp)) near = 1;
      }
    }
    if (!near) { food_reward -= 5; s->alive &= ~(1 << k); }
  }
  for (int k = 0; k < m->nfood; k++) if ((active_on_entry >> k & 1) && (hit_mask >> k & 1)) { food_hit -= 1; s->active &= ~(1 << k); }
  /* end-effector linear velocity (feeding.py:22, agent.py:69-72) */
  xf_t ee; ee_frame(s, &ee);
  int L = TI(m, AGX_T_EE_LINK); double wxp[3], vee[3];
  cross3(s->vsp[L], ee.p, wxp); add3(s->vsp[L] + 3, wxp, vee);
  double ee_speed = sqrt(dot3(vee, vee));
  /* human_preferences (env.py:237-274), feeding branch */
  double pref = TF(m, AGX_T_C_V) * (-ee_speed) + TF(m, AGX_T_C_F) * (-total_f) + TF(m, AGX_T_C_HF) * (tool_f < 10 ? 0.0 : -tool_f)
              + TF(m, AGX_T_C_FD) * food_hit + TF(m, AGX_T_C_FDV) * (-vel_sum);
  double sp[3], sR[9], dd[3]; tool_base_pose(s, sp, sR); sub3(s->target, sp, dd);
  double r = TF(m, AGX_T_W_DISTANCE) * (-sqrt(dot3(dd, dd))) + TF(m, AGX_T_W_ACTION) * (-sqrt(act_norm2)) + TF(m, AGX_T_W_FOOD) * food_reward + pref;
  *reward = (float)r;
  *done = s->iteration >= (int)TF(m, AGX_T_EPISODE_LEN);
  if (info) {
    info[AGX_INFO_TOTAL_FORCE] = (float)total_f;
    info[AGX_INFO_TASK_SUCCESS] = (float)(s->success >= s->total_food * TF(m, AGX_T_SUCCESS_FRAC));
    info[AGX_INFO_ROBOT_FORCE] = (float)robot_f; info[AGX_INFO_TOOL_FORCE] = (float)tool_f;
    info[AGX_INFO_FOOD_REWARD] = (float)food_reward; info[AGX_INFO_PREF] = (float)pref;
    info[AGX_INFO_NCONTACT] = (float)s->ncon; info[AGX_INFO_NROWS] = (float)s->nrows;
  }
  sim_store(s, state); free(s->rows); free(s);
}

/* ------------------------------------------------------------------------------------ test hooks */
void agxo_fk(const agxo_model* m, const float* state, double* pos, double* rot) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s);
  for (int d = 0; d < m->ndof; d++) { memcpy(pos + 3 * d, s->link[d].p, 24); memcpy(rot + 9 * d, s->link[d].R, 72); }
  free(s);
}
void agxo_ee_pose(const agxo_model* m, const float* state, double* pos3, double* quat4) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s);
  xf_t ee; ee_frame(s, &ee); memcpy(pos3, ee.p, 24); mat_to_quat(ee.R, quat4); free(s);
}
void agxo_crba(const agxo_model* m, const float* state, double* M) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s);
  int n = m->ndof; double Ic[MAXDOF][36];
  for (int d = 0; d < n; d++) memcpy(Ic[d], s->I6[d], sizeof Ic[d]);
  for (int d = n - 1; d >= 0; d--) { int par = RI(m, d, AGX_R_PARENT); if (par >= 0) for (int k = 0; k < 36; k++) Ic[par][k] += Ic[d][k]; }
  memset(M, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; i++) {
    double F[6]; mv6(Ic[i], s->S[i], F);
    for (int j = i; j >= 0; j = RI(m, j, AGX_R_PARENT)) { double v = dot6(s->S[j], F); M[i * n + j] = v; M[j * n + i] = v; }
  }
  free(s);
}
void agxo_aba(const agxo_model* m, const float* state, const double* tau, int with_damping, double* qdd) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s); aba(s, tau, with_damping, qdd); free(s);
}
void agxo_rnea_bias(const agxo_model* m, const float* state, double* h) {
  /* recursive Newton-Euler with qdd = 0, world-frame spatial algebra */
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s);
  int n = m->ndof; double a[MAXDOF][6], f[MAXDOF][6];
  for (int d = 0; d < n; d++) {
    int par = RI(m, d, AGX_R_PARENT);
    for (int k = 0; k < 6; k++) a[d][k] = (par < 0 ? 0.0 : a[par][k]) + s->cvp[d][k];
    double Ia[6], Iv[6], vIv[6], fe[6];
    mv6(s->I6[d], a[d], Ia); mv6(s->I6[d], s->vsp[d], Iv); crf(s->vsp[d], Iv, vIv); link_external_force(s, d, 0, fe);
    for (int k = 0; k < 6; k++) f[d][k] = Ia[k] + vIv[k] - fe[k];
  }
  for (int d = n - 1; d >= 0; d--) {
    h[d] = dot6(s->S[d], f[d]);
    int par = RI(m, d, AGX_R_PARENT); if (par >= 0) for (int k = 0; k < 6; k++) f[par][k] += f[d][k];
  }
  free(s);
}
void agxo_minv(const agxo_model* m, const float* state, double* Minv) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s);
  double qdd[MAXDOF]; aba(s, NULL, 0, qdd); minv_from_aba(s);
  memcpy(Minv, s->Minv, sizeof(double) * m->ndof * m->ndof); free(s);
}
int agxo_collide(const agxo_model* m, const float* state, double* out, int max_out) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state); kinematics(s); collide(s);
  int n = s->ncon < max_out ? s->ncon : max_out;
  for (int c = 0; c < n; c++) { double* o = out + 12 * c; const contact_t* k = &s->con[c];
    o[0] = k->ca; o[1] = k->cb; memcpy(o + 2, k->pa, 24); memcpy(o + 5, k->pb, 24); memcpy(o + 8, k->n, 24); o[11] = k->dist; }
  free(s); return n;
}
/* rows of one substep: [b, lo, hi, invD, lambda] per row; returns the row count */
int agxo_rows_debug(const agxo_model* m, float* state, double* out, int max_out) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state);
  s->rows = (row_t*)malloc(sizeof(row_t) * MAXROWS);
  substep(s);
  int n = s->nrows < max_out ? s->nrows : max_out;
  for (int r = 0; r < n; r++) { double* o = out + 5 * r; o[0] = s->rows[r].b; o[1] = s->rows[r].lo; o[2] = s->rows[r].hi; o[3] = s->rows[r].invD; o[4] = s->rows[r].lambda; }
  sim_store(s, state); free(s->rows); free(s); return n;
}
int agxo_substep_debug(const agxo_model* m, float* state, double* out, int max_out) {
  sim_t* s = (sim_t*)malloc(sizeof *s); sim_load(s, m, state);
  s->rows = (row_t*)malloc(sizeof(row_t) * MAXROWS);
  substep(s);
  int n = s->ncon < max_out ? s->ncon : max_out;
  for (int c = 0; c < n; c++) { double* o = out + 13 * c; const contact_t* k = &s->con[c];
    o[0] = k->ca; o[1] = k->cb; memcpy(o + 2, k->pa, 24); memcpy(o + 5, k->pb, 24); memcpy(o + 8, k->n, 24); o[11] = k->dist; o[12] = k->lambda_n; }
  sim_store(s, state); free(s->rows); free(s); return n;
}

/* ------------------------------------------------------------------------------------ world API
 * A persistent double-precision simulation that a PyBullet-shaped facade (tests/refbridge/) drives call by call, so that the
 * REFERENCE'S OWN Python (AssistiveEnv.take_step, <Task>Env.step / _get_obs / get_total_force / get_food_rewards,
 * human_preferences, Agent.enforce_joint_limits, Human.enforce_realistic_joint_limits) runs on top of this oracle's physics:
 * one agxo_world_step() is one p.stepSimulation() and nothing else -- no action processing, no limit reset, no arm-limit
 * classifier, no task layer; those stay with the caller.  Test infrastructure, like the rest of this file. */
struct agxo_world { sim_t s; int fresh; int anchor_given; double anchor[3]; };

static void world_refresh(agxo_world* w) { if (!w->fresh) { kinematics(&w->s); w->fresh = 1; } }

agxo_world* agxo_world_create(const agxo_model* m, const float* state, const float* cloth) {
  agxo_world* w = (agxo_world*)calloc(1, sizeof *w);
  sim_load(&w->s, m, state);
  w->s.rows = (row_t*)malloc(sizeof(row_t) * MAXROWS);
  cloth_attach(&w->s, cloth);
  w->fresh = 0; w->anchor_given = 0;
  return w;
}
void agxo_world_store(agxo_world* w, float* state, float* cloth) {
  sim_store(&w->s, state);
  if (cloth && w->s.cx) { const int n3 = 3 * agxo_cloth_nodes(w->s.m); for (int k = 0; k < n3; k++) { cloth[k] = (float)w->s.cx[k]; cloth[n3 + k] = (float)w->s.cv[k]; } }
}
void agxo_world_free(agxo_world* w) {
  if (!w) return;
  if (w->s.cx) { free(w->s.cx); free(w->s.cv); free(w->s.cq); free(w->s.ccon); free(w->s.ccon_node); free(w->s.ccon_shape); }
  free(w->s.rows); free(w);
}
/* getJointStates / resetJointState / setJointMotorControlArray(targetPositions) by DoF */
void agxo_world_joints(agxo_world* w, double* q, double* qd, double* qt) {
  for (int d = 0; d < w->s.ndof; d++) { if (q) q[d] = w->s.q[d]; if (qd) qd[d] = w->s.qd[d]; if (qt) qt[d] = w->s.qt[d]; }
}
void agxo_world_reset_joint(agxo_world* w, int d, double q, double qd) { w->s.q[d] = q; w->s.qd[d] = qd; w->fresh = 0; }
void agxo_world_set_target(agxo_world* w, int d, double qt) { w->s.qt[d] = qt; }
/* resetBasePositionAndOrientation of a free body given its BASE (URDF root link) frame; velocities kept (as Bullet does) */
void agxo_world_set_free_base(agxo_world* w, int b, const double* pos, const double* quat) {
  sim_t* s = &w->s; const agxo_model* m = s->m;
  double rp[3] = {FF(m, b, AGX_F_REFPOS), FF(m, b, AGX_F_REFPOS + 1), FF(m, b, AGX_F_REFPOS + 2)};
  double rq[4] = {FF(m, b, AGX_F_REFQUAT), FF(m, b, AGX_F_REFQUAT + 1), FF(m, b, AGX_F_REFQUAT + 2), FF(m, b, AGX_F_REFQUAT + 3)};
  /* base = com o ref  =>  com = base o ref^-1 */
  double Rb[9], Rr[9], Rc[9], t[3]; quat_to_mat(quat, Rb); quat_to_mat(rq, Rr); mmt3(Rb, Rr, Rc);
  mv3(Rc, rp, t); for (int k = 0; k < 3; k++) s->fpos[b][k] = pos[k] - t[k];
  mat_to_quat(Rc, s->fquat[b]); w->fresh = 0;
}
/* the cloth's attachment body (dressing.py:200-210: teleported to the end effector by update_targets); until it is given the
 * oracle's own rule applies (the end effector where the stepSimulation call began) */
void agxo_world_set_anchor(agxo_world* w, const double* pos) { memcpy(w->anchor, pos, 24); w->anchor_given = 1; }
/* one p.stepSimulation(): SIM_SUBSTEPS internal substeps, no hooks */
void agxo_world_step(agxo_world* w) {
  sim_t* s = &w->s;
  if (w->anchor_given) { memcpy(s->anchor, w->anchor, 24); s->anchor_set = 1; } else s->anchor_set = 0;
  for (int k = 0; k < s->m->sim_sub; k++) substep_h(s, 0);
  w->fresh = 0;
}
/* world frames and velocities (getLinkState(computeForwardKinematics, computeLinkVelocity) / getBasePositionAndOrientation / getBaseVelocity).
 * kind 0: URDF link frame of moving link `index`; 1: base frame of free body `index`; 2: static human collision body `index`;
 * 3: robot base; 4: end-effector frame of tool `index` (robot.right/left_end_effector); 5: the frame of the tool the task observes
 * (AGX_T_TOOL_OBS_*, e.g. link 1 of the wiper) for free body `index`.  lin = velocity of the frame origin.  Returns 0 on a bad kind */
int agxo_world_frame(agxo_world* w, int kind, int index, double* pos, double* quat, double* lin, double* ang) {
  world_refresh(w);
  sim_t* s = &w->s; const agxo_model* m = s->m;
  double p[3] = {0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, v[3] = {0, 0, 0}, om[3] = {0, 0, 0};
  if (kind == 0 || kind == 4) {
    int L = kind == 0 ? index : TI(m, index ? AGX_T_EE2_LINK : AGX_T_EE_LINK);
    if (L < 0 || L >= s->ndof) return 0;
    if (kind == 0) { memcpy(p, s->link[L].p, 24); memcpy(R, s->link[L].R, 72); }
    else { xf_t ee; ee_frame_of(s, index, &ee); memcpy(p, ee.p, 24); memcpy(R, ee.R, 72); }
    double wxp[3]; cross3(s->vsp[L], p, wxp); add3(s->vsp[L] + 3, wxp, v); memcpy(om, s->vsp[L], 24);
  } else if (kind == 1 || kind == 5) {
    if (index < 0 || index >= s->nfree) return 0;
    if (kind == 5) tool_base_pose_of(s, index, p, R);
    else { double rp[3] = {FF(m, index, AGX_F_REFPOS), FF(m, index, AGX_F_REFPOS + 1), FF(m, index, AGX_F_REFPOS + 2)};
      double rq[4] = {FF(m, index, AGX_F_REFQUAT), FF(m, index, AGX_F_REFQUAT + 1), FF(m, index, AGX_F_REFQUAT + 2), FF(m, index, AGX_F_REFQUAT + 3)}, Rr[9];
      quat_to_mat(rq, Rr); xf_apply(&s->freex[index], rp, p); mm3(s->freex[index].R, Rr, R); }
    double r[3], wr[3]; sub3(p, s->fpos[index], r); cross3(s->fw[index], r, wr); add3(s->fv[index], wr, v); memcpy(om, s->fw[index], 24);
  } else if (kind == 2) { if (index < 0 || index >= m->nhuman) return 0; memcpy(p, s->human[index].p, 24); memcpy(R, s->human[index].R, 72); }
  else if (kind == 3) { const int bl = m->i[AGX_H_BASE_LINK]; const xf_t* B = bl > 0 ? &s->link[bl - 1] : &s->base; memcpy(p, B->p, 24); memcpy(R, B->R, 72);
    if (bl > 0) { double wxp[3]; cross3(s->vsp[bl - 1], p, wxp); add3(s->vsp[bl - 1] + 3, wxp, v); memcpy(om, s->vsp[bl - 1], 24); } }   /* the twist of a floating base */
  else return 0;
  if (pos) memcpy(pos, p, 24); if (quat) mat_to_quat(R, quat); if (lin) memcpy(lin, v, 24); if (ang) memcpy(ang, om, 24);
  return 1;
}
/* getContactPoints after the last agxo_world_step: rows of 16 doubles {colliderA, colliderB, pA(3), pB(3), n(3) from B to A,
 * distance, normal force, 1 = solver contact / 0 = manifold point without a solver row}.  First the manifold points of the groups
 * whose manifold the task reads (collide order; force of the solver contact of the same pair, if any), then the remaining solver
 * contacts.  Returns the count */
int agxo_world_contacts(agxo_world* w, double* out, int max_out) {
  sim_t* s = &w->s; const double dt = s->m->dt; int n = 0;
  unsigned char used[MAXC]; memset(used, 0, sizeof used);
  for (int q = 0; q < s->nman; q++) {
    const contact_t* k = &s->man[q]; double f = 0; int solver = 0;
    for (int c = 0; c < s->ncon; c++) if (!used[c] && s->con[c].ca == k->ca && s->con[c].cb == k->cb) { f = s->con[c].lambda_n / dt; solver = 1; used[c] = 1; break; }
    if (n < max_out) { double* o = out + 16 * n; o[0] = k->ca; o[1] = k->cb; memcpy(o + 2, k->pa, 24); memcpy(o + 5, k->pb, 24); memcpy(o + 8, k->n, 24); o[11] = k->dist; o[12] = f; o[13] = solver; o[14] = o[15] = 0; }
    n++;
  }
  for (int c = 0; c < s->ncon; c++) {
    if (used[c]) continue;
    const contact_t* k = &s->con[c];
    if (n < max_out) { double* o = out + 16 * n; o[0] = k->ca; o[1] = k->cb; memcpy(o + 2, k->pa, 24); memcpy(o + 5, k->pb, 24); memcpy(o + 8, k->n, 24); o[11] = k->dist; o[12] = k->lambda_n / dt; o[13] = 1; o[14] = o[15] = 0; }
    n++;
  }
  return n;
}
/* getClosestPoints(distance): one row {colliderA, colliderB, pA(3), pB(3), distance} per collider pair closer than `dist`, at the current poses */
int agxo_world_closest(agxo_world* w, const int* ca, int na, const int* cb, int nb, double dist, double* out, int max_out) {
  world_refresh(w);
  sim_t* s = &w->s; int n = 0;
  for (int a = 0; a < na; a++) for (int b = 0; b < nb; b++) {
    contact_t k; if (!narrowphase(s, ca[a], cb[b], dist, &k) || k.dist >= dist) continue;
    if (n < max_out) { double* o = out + 9 * n; o[0] = ca[a]; o[1] = cb[b]; memcpy(o + 2, k.pa, 24); memcpy(o + 5, k.pb, 24); o[8] = k.dist; }
    n++;
  }
  return n;
}
/* getSoftBodyData: node positions x[NN][3] and the contacts of the last internal substep {x, y, z, fx, fy, fz}; returns the contact count */
int agxo_world_cloth(agxo_world* w, double* x, double* contacts, int max_contacts) {
  sim_t* s = &w->s; if (!s->cx) return -1;
  const int n3 = 3 * agxo_cloth_nodes(s->m);
  if (x) memcpy(x, s->cx, sizeof(double) * n3);
  int n = s->nccon < max_contacts ? s->nccon : max_contacts;
  if (contacts) memcpy(contacts, s->ccon, sizeof(double) * 6 * n);
  return s->nccon;
}
void agxo_world_set_cloth_gravity(agxo_world* w, double gz) { w->s.dr_gravity = gz; }
/* a water particle as the body the reference holds it as (drinking.py:52-91): getBasePositionAndOrientation / getBaseVelocity,
 * resetBasePositionAndOrientation (velocity kept, as Bullet does), and the two proximity queries of get_water_rewards --
 * bit 0: getClosestPoints(water, tool, dist) is not empty (surface distance of the sphere to a piece of the cup <= dist),
 * bit 1: getContactPoints(water, human) is not empty (a person's shape among the particle's contacts of the last internal substep) */
int agxo_world_particle(agxo_world* w, int k, double* pos, double* vel) {
  sim_t* s = &w->s; if (!s->cx || k < 0 || k >= agxo_cloth_nodes(s->m)) return 0;
  if (pos) memcpy(pos, s->cx + 3 * k, 24);
  if (vel) memcpy(vel, s->cv + 3 * k, 24);
  return 1;
}
void agxo_world_set_particle(agxo_world* w, int k, const double* pos) {
  sim_t* s = &w->s; if (!s->cx || k < 0 || k >= agxo_cloth_nodes(s->m)) return;
  memcpy(s->cx + 3 * k, pos, 24);
}
int agxo_world_particle_query(agxo_world* w, int k, double dist) {
  sim_t* s = &w->s; const agxo_model* m = s->m; if (!s->cx || k < 0 || k >= agxo_cloth_nodes(m)) return 0;
  world_refresh(w);
  int out = particle_near_tool(s, s->cx + 3 * k, dist);
  for (int c = 0; c < s->nccon; c++) {
    if (s->ccon_node[c] != k) continue;
    const int col = m->i[m->o_cloth + m->i[m->o_cloth + AGX_CL_OFF_SHAPE] + 4 * s->ccon_shape[c]];
    if (CI(m, col, AGX_C_TAG) == AGX_TAG_HUMAN) out |= 2;
  }
  return out;
}
/* Util.sleeve_on_arm_reward as finish_dressing evaluates it (util.py:134-202), for direct comparison with the reference's function:
 * pts[6][3], shoulder / elbow / wrist, the common radius -> out[9] = {forearm_in_sleeve, upperarm_in_sleeve, distance_along_forearm,
 * distance_along_upperarm, distance_to_hand, distance_to_elbow, distance_to_shoulder, forearm_length, upperarm_length} */
void agxo_sleeve_reward(const double* pts6, const double* shoulder, const double* elbow, const double* wrist, double rad, double* out) {
  const double (*pts)[3] = (const double (*)[3])pts6;
  double we[3], es[3]; sub3(wrist, elbow, we); sub3(shoulder, elbow, es);
  const double lwe = sqrt(dot3(we, we)), les = sqrt(dot3(es, es));
  double hand_end[3], elbow_end[3], shoulder_end[3];
  for (int k = 0; k < 3; k++) { hand_end[k] = wrist[k] + we[k] / lwe * rad * 2; elbow_end[k] = elbow[k] - we[k] / lwe * rad; shoulder_end[k] = shoulder[k] + es[k] / les * rad; }
  const int around_fore = points_around_axis(pts, 6, elbow_end, hand_end, hand_end), around_upper = points_around_axis(pts, 6, shoulder_end, elbow_end, shoulder_end);
  const int f1 = line_intersects_triangle(pts[0], pts[1], pts[2], hand_end, elbow_end), f2 = line_intersects_triangle(pts[3], pts[4], pts[5], hand_end, elbow_end);
  const int u1 = line_intersects_triangle(pts[0], pts[1], pts[2], elbow_end, shoulder_end), u2 = line_intersects_triangle(pts[3], pts[4], pts[5], elbow_end, shoulder_end);
  double centre[3] = {0, 0, 0}, d[3]; for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) centre[k] += pts[i][k] / 6.0;
  out[0] = around_fore && (f1 || f2); out[1] = around_upper && (u1 || u2);
  sub3(hand_end, centre, d); out[2] = out[4] = sqrt(dot3(d, d));
  sub3(centre, elbow, d); out[3] = sqrt(dot3(d, d));
  sub3(elbow_end, centre, d); out[5] = sqrt(dot3(d, d));
  sub3(shoulder_end, centre, d); out[6] = sqrt(dot3(d, d));
  sub3(hand_end, elbow_end, d); out[7] = sqrt(dot3(d, d));
  out[8] = les;
}
