// agx_reset.h -- device-side reset generator: FeedingEnv.reset's sampling for one environment per wavefront.
//
// What it replaces (reference paths): FeedingEnv.reset up to the settle loop (assistive_gym/envs/feeding.py:114-177):
// plane friction (envs/env.py:120), Human.init draws (agents/human.py:72-92), the posed static human
// (feeding.py:124-126, human.py:104-127, tree of human_creation.py:188-278), the mouth target (feeding.py:184-196),
// init_robot_pose -> Robot.ik_random_restarts (env.py:276-310, agents/robot.py:84-121), gripper / tool / bowl / food
// placement (feeding.py:143-166, tool.py:49-62, furniture.py:32-34).  The 25 settle substeps (feeding.py:178-179)
// are the stepper's own kernels (agx_settle).
//
// Mapping to the wavefront:
//   * the scalar draws and the human pose are cheap and computed redundantly / one body per lane;
//   * the IK restarts of robot.py:88-99 -- a sequential loop in the reference -- run 64 AT A TIME, one restart
//     per lane, each lane iterating its own damped-least-squares problem in registers.  Every random number has
//     a fixed (stream, slot) address in a counter-based generator (Philox4x32-10 keyed by the env seed), so
//     "the first successful restart" is simply the lowest lane of the first round whose ballot is non-empty;
//   * all of it in float64: it runs once per episode, MI355X has full-rate FP64 vector units, and the result is
//     then independent of evaluation order to ~1e-13, which is what makes it checkable against the numpy
//     restatement (oracle/reset_oracle.py) through the discontinuous accept / reject decisions.
// Bullet's own IK is not reproduced; the acceptance test of ik_random_restarts is: position / orientation
// thresholds (robot.py:97) AND the collision rejection (robot.py:105-112: a solution whose arm touches the human, the table or the
// wheelchair is skipped; env.py:299-308: so is one whose tool does).  The contacts of a sampled state come from the stepper's own
// build kernel (libagx launch_sample: sample -> [build, verdict, re-sample from the next restart] x AGX_X_COLLISION_TRIES); the
// reference's outer loop re-draws all restarts up to 3 times instead of continuing with the next restart.
#pragma once

namespace agx {

struct d3 { double x, y, z; };
struct dq { double x, y, z, w; };

AGX_DEV d3 dmk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
AGX_DEV d3 operator+(d3 a, d3 b) { return dmk(a.x + b.x, a.y + b.y, a.z + b.z); }
AGX_DEV d3 operator-(d3 a, d3 b) { return dmk(a.x - b.x, a.y - b.y, a.z - b.z); }
AGX_DEV double ddot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
AGX_DEV d3 dcross(d3 a, d3 b) { return dmk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
AGX_DEV d3 dld3(const float* p) { return dmk((double)p[0], (double)p[1], (double)p[2]); }
AGX_DEV dq dld4(const float* p) { dq q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
AGX_DEV dq dq_ident() { dq q; q.x = 0; q.y = 0; q.z = 0; q.w = 1; return q; }
AGX_DEV dq dqmul(dq a, dq b) {
  dq r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}
AGX_DEV d3 dqrot(dq q0, d3 v) {
  const double n = 1.0 / sqrt(q0.x * q0.x + q0.y * q0.y + q0.z * q0.z + q0.w * q0.w);
  const double x = q0.x * n, y = q0.y * n, z = q0.z * n, w = q0.w * n;
  return dmk((1 - 2 * (y * y + z * z)) * v.x + 2 * (x * y - z * w) * v.y + 2 * (x * z + y * w) * v.z,
             2 * (x * y + z * w) * v.x + (1 - 2 * (x * x + z * z)) * v.y + 2 * (y * z - x * w) * v.z,
             2 * (x * z - y * w) * v.x + 2 * (y * z + x * w) * v.y + (1 - 2 * (x * x + y * y)) * v.z);
}
AGX_DEV dq dq_axis_angle(d3 a, double th) {
  const double n = sqrt(ddot(a, a));
  if (n < 1e-12) return dq_ident();
  const double s = sin(0.5 * th) / n;
  dq q; q.x = a.x * s; q.y = a.y * s; q.z = a.z * s; q.w = cos(0.5 * th);
  return q;
}
// (pa, qa) o (pb, qb)
AGX_DEV void dcompose(d3 pa, dq qa, d3 pb, dq qb, d3& p, dq& q) { p = pa + dqrot(qa, pb); q = dqmul(qa, qb); }

AGX_DEV double wave_bcast_d(double x, int src) {
  int w[2]; __builtin_memcpy(w, &x, 8);
  w[0] = wave_bcast_i(w[0], src); w[1] = wave_bcast_i(w[1], src);
  double r; __builtin_memcpy(&r, w, 8); return r;
}

// ---- counter-based random numbers: Philox4x32-10, counter (slot, stream, 0, 0), key = env seed ----------
AGX_DEV double rs_u01(uint32_t k0, uint32_t k1, uint32_t stream, uint32_t slot) {
  uint32_t c0 = slot, c1 = stream, c2 = 0u, c3 = 0u;
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (double)((((uint64_t)(c0 >> 5)) << 26) | (uint64_t)(c1 >> 6)) * (1.0 / 9007199254740992.0);
}
// stream 0 slots, restart-stream slots (+ DoF)
enum { RS_FRICTION = 0, RS_GENDER = 1, RS_IMPAIRMENT = 2, RS_LIMIT = 3, RS_STRENGTH = 4, RS_HEAD = 8, RS_EE = 12, RS_BOWL = 16, RS_EE2 = 20,
       RS_TREMOR = 32, RS_LIMB = 48, RS_TARGET_LEN = 49, RS_TARGET_TH = 50, RS_R_REST = 0, RS_R_LO = 16, RS_R_HI = 32,
       RS_T_STREAM0 = 2000, RS_T_X = 0, RS_T_Y = 1, RS_T_YAW = 2, RS_T_REST = 16 };      // base pose search: stream RS_T_STREAM0 + 64 (try x rounds + round) + candidate; rest pose of goal g at RS_T_REST + 8 g + DoF
enum { RS_IMP_NONE = 0, RS_IMP_LIMITS = 1, RS_IMP_WEAKNESS = 2, RS_IMP_TREMOR = 3, RS_MODE_RANDOM = -1, RS_MODE_NO_TREMOR = -2 };
constexpr int RS_NARM = 7;   // arm DoFs solved by the IK; agx_create checks the blob (a serial chain 0..6 carrying the end effector)

struct ResetCtx {
  const float* bf; const int* bi;
  const float* xf; const int* xi;     // reset section
  const float* rob; const float* task;
  int nj, gender;
  double ls;                          // limit scale of the sampled human
  double head[3];                     // head joint angle draws
  d3 base_p; dq base_q;               // robot base: the blob's fixed pose, or this lane's candidate of the base pose search
  int chain[7];                       // DoF of the arm's k-th joint (AGX_X_CHAIN; the second arm of a two-armed robot: AGX_X_CHAIN2, see rs_second_arm)
  int ee_pos, ee_quat;                // task words of that arm's end-effector frame (AGX_T_EE_POS / _QUAT, or AGX_T_EE2_*)
  const float* settled;               // AGX_X_FLAGS bit 4: the joint angles (q) of the rag-doll record the human's pose is read from; else null
  const float* fell;                  // AGX_X_FLAGS bit 7: the fall model's record after the arm's fall (same layout as this blob's records); else null
  bool fall_stage;                    // AGX_X_FLAGS bit 8: this blob is the fall model
  int fell_q;                         // offset of the human DoFs' joint angles inside `fell`
  int nhdof;
};
constexpr int RS_RAGDOLL = 64;        // stream 0 slots RS_RAGDOLL + k: the jitter of the rag doll's k-th joint (bed_bathing.py:126)
constexpr int RS_SETTLE_VIRTUAL = 6;  // the rag-doll model's virtual root joints come first (x, y, z, yaw, pitch, roll), then its joints in PyBullet
constexpr int RS_SETTLE_FIXED = 24;   // order without the fixed waist joint 24 (model/compiler.py compile_bed_settle)
#define XF(c, k) ((double)(c).xf[(k)])
#define XI(c, k) ((c).xi[(k)])

// joint angle of human joint j: task preset (+ head draw), clamped to the limits (agent.py:240-250)
AGX_DEV double rs_joint_angle(const ResetCtx& c, int j) {
  const int base = XI(c, AGX_X_OFF_JOINTS) + (c.gender * c.nj + j) * AGX_XJ_STRIDE;
  const int flags = c.xi[base + AGX_XJ_FLAGS];
  if (!(flags & 1)) return 0.0;
  if (c.fell)                                                              // a joint of the arm that fell: where it came to rest (arm_manipulation.py:145-151)
    for (int k = 0; k < c.nhdof; k++) if (c.xi[XI(c, AGX_X_OFF_DYN) + k] == j) return (double)c.fell[c.fell_q + k];
  const double s = (flags & 2) ? c.ls : 1.0;
  if (c.settled && !(c.fall_stage && (flags & 4))) {                       // where the rag doll came to rest (bed_bathing.py:129-137)
    const double a = (double)c.settled[RS_SETTLE_VIRTUAL + j - (j > RS_SETTLE_FIXED ? 1 : 0)];
    // the fall model: setup_joints -> enforce_joint_limits clamps every joint once more (arm_manipulation.py:139-140, human.py:121)
    // ... and the task's own sampler (c.fell: the stage after the fall) reads the same clamped pose: its goal poses (wrist, elbow, waist, stomach) must come
    // from the tree the fall model wrote the human bodies from, not from the rag doll's unclamped angles (arm_manipulation.py:139-151)
    return (c.fall_stage || c.fell) ? fmin(fmax(a, (double)c.xf[base + AGX_XJ_LOWER] * s), (double)c.xf[base + AGX_XJ_UPPER] * s) : a;
  }
  double a = (double)c.xf[base + AGX_XJ_PRESET];
  const int k = c.xi[base + AGX_XJ_DRAW];
  if (k >= 0) a += (k == 0 ? c.head[0] : (k == 1 ? c.head[1] : c.head[2]));
  return fmin(fmax(a, (double)c.xf[base + AGX_XJ_LOWER] * s), (double)c.xf[base + AGX_XJ_UPPER] * s);
}
// world pose of a human link frame (-1 = base): from the link up to the base, then the base transform
AGX_DEV void rs_link_pose(const ResetCtx& c, int link, d3& p, dq& q) {
  p = dmk(0, 0, 0); q = dq_ident();
  for (int j = link; j >= 0;) {
    const int base = XI(c, AGX_X_OFF_JOINTS) + (c.gender * c.nj + j) * AGX_XJ_STRIDE;
    const dq jq = dq_axis_angle(dld3(c.xf + base + AGX_XJ_AXIS), rs_joint_angle(c, j));
    dcompose(dld3(c.xf + base + AGX_XJ_OFF), jq, p, q, p, q);
    j = c.xi[base + AGX_XJ_PARENT];
  }
  if (c.settled) {      // the rag doll's base: position q[0..2], orientation Rz(q[3]) Ry(q[4]) Rx(q[5])
    const dq bq = dqmul(dqmul(dq_axis_angle(dmk(0, 0, 1), (double)c.settled[3]), dq_axis_angle(dmk(0, 1, 0), (double)c.settled[4])), dq_axis_angle(dmk(1, 0, 0), (double)c.settled[5]));
    dcompose(dmk((double)c.settled[0], (double)c.settled[1], (double)c.settled[2]), bq, p, q, p, q);
  } else
  dcompose(dld3(c.xf + (c.gender ? AGX_X_HBASE_F : AGX_X_HBASE_M)), dq_ident(), p, q, p, q);
}

// arm forward kinematics: joint origins, world joint axes, end-effector pose
// end-effector pose of a robot on wheels (AGX_X_FLAGS bit 3): the chain from the link that carries the end effector up to the base
// (virtual joints, lift, telescoping joints, wrist: any joint types), every joint at zero but the lift
AGX_DEV void rs_mobile_fk(const ResetCtx& c, int lift_dof, double lift_q, d3& pe, dq& oe) {
  int links[24], n = 0;
  for (int d = ((const int*)c.task)[AGX_T_EE_LINK]; d >= 0 && n < 24; d = ((const int*)c.rob)[d * AGX_R_STRIDE + AGX_R_PARENT]) links[n++] = d;
  d3 pp = c.base_p; dq pq = c.base_q;
  for (int k = n - 1; k >= 0; k--) {
    const int d = links[k];
    const float* r = c.rob + d * AGX_R_STRIDE;
    d3 jp; dq jq;
    dcompose(pp, pq, dld3(r + AGX_R_TPOS), dld4(r + AGX_R_TQUAT), jp, jq);
    const double qd = d == lift_dof ? lift_q : fmin(fmax((double)r[AGX_R_QT0], (double)r[AGX_R_LOWER]), (double)r[AGX_R_UPPER]);
    const d3 ax = dld3(r + AGX_R_AXIS);
    if (((const int*)r)[AGX_R_JTYPE] == 1) { pp = jp + dqrot(jq, dmk(ax.x * qd, ax.y * qd, ax.z * qd)); pq = jq; }
    else { pp = jp; pq = dqmul(jq, dq_axis_angle(ax, qd)); }
  }
  dcompose(pp, pq, dld3(c.task + c.ee_pos), dld4(c.task + c.ee_quat), pe, oe);
}
AGX_DEV void rs_arm_fk(const ResetCtx& c, const double* q, d3* pos, d3* axw, d3& pe, dq& oe) {
  d3 pp = c.base_p; dq pq = c.base_q;
#pragma unroll
  for (int d = 0; d < RS_NARM; d++) {
    const float* r = c.rob + c.chain[d] * AGX_R_STRIDE;
    d3 jp; dq jq;
    dcompose(pp, pq, dld3(r + AGX_R_TPOS), dld4(r + AGX_R_TQUAT), jp, jq);
    const d3 ax = dld3(r + AGX_R_AXIS);
    pq = dqmul(jq, dq_axis_angle(ax, q[d]));
    pp = jp;
    pos[d] = jp; axw[d] = dqrot(pq, ax);
  }
  dcompose(pp, pq, dld3(c.task + c.ee_pos), dld4(c.task + c.ee_quat), pe, oe);
}
// the context of the SECOND arm of a two-armed robot (AGX_X_FLAGS bit 9): its chain and its end-effector frame, the same base
AGX_DEV ResetCtx rs_second_arm(const ResetCtx& c) {
  ResetCtx c2 = c;
  for (int d = 0; d < RS_NARM; d++) c2.chain[d] = c.xi[AGX_X_CHAIN2 + d];
  c2.ee_pos = AGX_T_EE2_POS; c2.ee_quat = AGX_T_EE2_QUAT;
  return c2;
}

// damped least squares from q (in place), joint box [lo, hi]; ORIENT = false: position only (a 3 x 3 system)
template <bool ORIENT>
AGX_DEV void rs_ik(const ResetCtx& c, double* q, const double* lo, const double* hi, d3 tpos, dq tquat, int iters) {
  const double lam2 = XF(c, AGX_X_IK_DAMP) * XF(c, AGX_X_IK_DAMP), tol = XF(c, AGX_X_IK_TOL), maxstep = XF(c, AGX_X_IK_MAXSTEP);
  constexpr int NE = ORIENT ? 6 : 3;
  for (int it = 0; it < iters; it++) {
    d3 pos[RS_NARM], axw[RS_NARM], pe; dq oe;
    rs_arm_fk(c, q, pos, axw, pe, oe);
    const d3 ep = tpos - pe;
    d3 er = dmk(0, 0, 0);
    if (ORIENT) {
      dq oc; oc.x = -oe.x; oc.y = -oe.y; oc.z = -oe.z; oc.w = oe.w;
      dq qe = dqmul(tquat, oc);
      if (qe.w < 0) { qe.x = -qe.x; qe.y = -qe.y; qe.z = -qe.z; }
      er = dmk(2.0 * qe.x, 2.0 * qe.y, 2.0 * qe.z);
    }
    if (sqrt(ddot(ep, ep)) < tol && sqrt(ddot(er, er)) < tol) break;
    // columns of the 6 x NARM Jacobian: [axis x (pe - origin); axis]
    double J[NE][RS_NARM];
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) {
      const d3 l = dcross(axw[d], pe - pos[d]);
      J[0][d] = l.x; J[1][d] = l.y; J[2][d] = l.z;
      if (ORIENT) { J[NE - 3][d] = axw[d].x; J[NE - 2][d] = axw[d].y; J[NE - 1][d] = axw[d].z; }
    }
    // A = J J^T + lam^2 I, Cholesky A = L L^T, solve A y = e
    double A[NE][NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double s = (i == j) ? lam2 : 0.0;
#pragma unroll
        for (int d = 0; d < RS_NARM; d++) s += J[i][d] * J[j][d];
        A[i][j] = s;
      }
    }
    double y[6] = {ep.x, ep.y, ep.z, er.x, er.y, er.z};
#pragma unroll
    for (int j = 0; j < NE; j++) {
      double s = A[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= A[j][k] * A[j][k];
      const double ljj = sqrt(s), inv = 1.0 / ljj;
      A[j][j] = ljj;
#pragma unroll
      for (int i = j + 1; i < NE; i++) {
        double t = A[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k];
        A[i][j] = t * inv;
      }
    }
#pragma unroll
    for (int i = 0; i < NE; i++) {
      double t = y[i];
#pragma unroll
      for (int k = 0; k < i; k++) t -= A[i][k] * y[k];
      y[i] = t / A[i][i];
    }
#pragma unroll
    for (int i = NE - 1; i >= 0; i--) {
      double t = y[i];
#pragma unroll
      for (int k = i + 1; k < NE; k++) t -= A[k][i] * y[k];
      y[i] = t / A[i][i];
    }
    double dqv[RS_NARM], step = 0.0;
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < NE; i++) s += J[i][d] * y[i];
      dqv[d] = s; step = fmax(step, fabs(s));
    }
    const double scale = step > maxstep ? maxstep / step : 1.0;
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) q[d] = fmin(fmax(q[d] + (step > maxstep ? dqv[d] * scale : dqv[d]), lo[d]), hi[d]);
  }
}

// Robot.joint_limited_weighting (robot.py:217-228) and the joint-limit-weighted kinematic isotropy of a solution (robot.py:186-191):
// M = J diag(w) J^T, JLWKI = det(M)^(1/6) / (trace(M) / 6); det through the Cholesky factor (0 when M is not positive definite)
AGX_DEV double rs_jlwki(const ResetCtx& c, const double* q) {
  d3 pos[RS_NARM], axw[RS_NARM], pe; dq oe;
  rs_arm_fk(c, q, pos, axw, pe, oe);
  double J[6][RS_NARM], w[RS_NARM];
#pragma unroll
  for (int d = 0; d < RS_NARM; d++) {
    const d3 l = dcross(axw[d], pe - pos[d]);
    J[0][d] = l.x; J[1][d] = l.y; J[2][d] = l.z; J[3][d] = axw[d].x; J[4][d] = axw[d].y; J[5][d] = axw[d].z;
    const double lower = (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_LOWER], upper = (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_UPPER];
    const double qr = 0.5 * (upper - lower);
    const double wd = 1.0 - pow(0.5, (qr - fabs(qr - q[d] + lower)) / (0.05 * qr) + 1.0);
    w[d] = fmax(wd, 0.001);
  }
  double M[6][6], tr = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < RS_NARM; d++) s += J[i][d] * w[d] * J[j][d];
      M[i][j] = s;
    }
    tr += M[i][i];
  }
  double det = 1.0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double s = M[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) s -= M[j][k] * M[j][k];
    if (!(s > 0.0)) return 0.0;
    const double ljj = sqrt(s), inv = 1.0 / ljj;
    M[j][j] = ljj; det *= s;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = M[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) t -= M[i][k] * M[j][k];
      M[i][j] = t * inv;
    }
  }
  return pow(det, 1.0 / 6.0) / (tr / 6.0);
}

// One environment.  gstate: this env's state record (fully overwritten).  ginfo (may be null): float[4] =
// {IK succeeded, restarts used, end-effector position error, impairment index}.
// first_restart: successful IK restarts below this index were rejected because the arm / tool touched the human, the table or the
// wheelchair there (robot.py:105-112 `continue`s to the next restart); they neither succeed again nor count as the closest attempt.
// Returns the index of the accepted restart, -1 if no restart met the thresholds (the closest attempt is used, robot.py:114-117).
// settled (may be null): the state record of this environment in the rag-doll model after its settle (AGX_X_FLAGS bit 4 of this blob)
AGX_DEV int env_sample(const uint32_t* __restrict__ blob, float* __restrict__ gstate, uint32_t seed_lo, uint32_t seed_hi,
                       int impairment_mode, int gender_mode, float* __restrict__ ginfo, int lane, int first_restart = 0, const float* __restrict__ settled = nullptr,
                       const float* __restrict__ fell = nullptr) {
  ResetCtx c;
  c.bf = (const float*)blob; c.bi = (const int*)blob;
  c.xf = c.bf + c.bi[AGX_H_OFF_RESET]; c.xi = c.bi + c.bi[AGX_H_OFF_RESET];
  c.rob = c.bf + c.bi[AGX_H_OFF_ROBOT]; c.task = c.bf + c.bi[AGX_H_OFF_TASK];
  c.nj = XI(c, AGX_X_NJOINT);
  c.base_p = dld3(c.xf + AGX_X_BASE_POS); c.base_q = dld4(c.xf + AGX_X_BASE_QUAT);
  for (int d = 0; d < RS_NARM; d++) c.chain[d] = XI(c, AGX_X_CHAIN + d);
  c.ee_pos = AGX_T_EE_POS; c.ee_quat = AGX_T_EE_QUAT;
  const int ndof = c.bi[AGX_H_NDOF], nrobot = c.bi[AGX_H_NROBOT], nhdof = c.bi[AGX_H_NHDOF], nfree = c.bi[AGX_H_NFREE];
  const int nhuman = c.bi[AGX_H_NHUMAN], nfood = c.bi[AGX_H_NFOOD], state_words = c.bi[AGX_H_STATE_WORDS];
  const int sQ = c.bi[AGX_H_S_Q], sQT = c.bi[AGX_H_S_QT], sFREE = c.bi[AGX_H_S_FREE], sBASE = c.bi[AGX_H_S_BASE];
  const int sHUMAN = c.bi[AGX_H_S_HUMAN], sENV = c.bi[AGX_H_S_ENV], sTREMOR = c.bi[AGX_H_S_TREMOR];
  int* gstate_i = (int*)gstate;

  // ---- scalar draws (every lane computes the same values) ------------------------------------------
  const double friction = XF(c, AGX_X_FRIC_LO) + (XF(c, AGX_X_FRIC_HI) - XF(c, AGX_X_FRIC_LO)) * rs_u01(seed_lo, seed_hi, 0, RS_FRICTION);
  c.gender = gender_mode >= 0 ? gender_mode : (rs_u01(seed_lo, seed_hi, 0, RS_GENDER) < 0.5 ? 0 : 1);     // human.py:76-78
  int imp = impairment_mode;
  if (imp < 0) {                                                                                             // human.py:80-81
    const int nchoice = impairment_mode == RS_MODE_RANDOM ? 4 : 3;
    imp = (int)(rs_u01(seed_lo, seed_hi, 0, RS_IMPAIRMENT) * nchoice);
    if (imp > nchoice - 1) imp = nchoice - 1;
  }
  c.ls = imp != RS_IMP_LIMITS ? 1.0 : XF(c, AGX_X_LIMIT_LO) + (1.0 - XF(c, AGX_X_LIMIT_LO)) * rs_u01(seed_lo, seed_hi, 0, RS_LIMIT);   // human.py:85
  for (int k = 0; k < 3; k++) c.head[k] = (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_HEAD + k) - 1.0) * XF(c, AGX_X_HEAD_RANGE);          // feeding.py:125
  const double strength = imp != RS_IMP_WEAKNESS ? 1.0 : XF(c, AGX_X_STRENGTH_LO) + (1.0 - XF(c, AGX_X_STRENGTH_LO)) * rs_u01(seed_lo, seed_hi, 0, RS_STRENGTH);   // human.py:86
  const int xflags = XI(c, AGX_X_FLAGS);
  c.settled = (xflags & 16) ? settled : nullptr;      // (agx_sample_reset refuses to run such a blob without its rag-doll model attached)
  c.fall_stage = (xflags & 256) != 0;
  c.fell = ((xflags & 128) && !c.fall_stage) ? fell : nullptr;
  c.fell_q = sQ + nrobot; c.nhdof = nhdof;
  const int sQD = c.bi[AGX_H_S_QD];

  for (int w = lane; w < state_words; w += AGX_WAVE) gstate[w] = 0.f;
  wave_sync();

  if (xflags & 32) {
    // ---- this blob IS the rag-doll model (bed_settle): the record it is dropped from (bed_bathing.py:119-127) -- base in the air, every
    // joint U(-r, r) clamped to its limits (Agent.set_joint_angles + enforce_joint_limits), at rest; friction / gender / limit scale as the
    // task blob's sampler draws them (same seed, same stream-0 slots).  One joint per lane.
    if (lane < ndof) {
      double qv;
      if (lane < 3) qv = XF(c, (c.gender ? AGX_X_HBASE_F : AGX_X_HBASE_M) + lane);
      else if (lane < RS_SETTLE_VIRTUAL) qv = XF(c, AGX_X_EE_TARGET + lane - 3);                      // yaw, pitch, roll of the drop pose
      else {
        const int k = lane - RS_SETTLE_VIRTUAL, j = k + (k >= RS_SETTLE_FIXED ? 1 : 0);
        const int jb = XI(c, AGX_X_OFF_JOINTS) + (c.gender * c.nj + j) * AGX_XJ_STRIDE;
        const double sc = (c.xi[jb + AGX_XJ_FLAGS] & 2) ? c.ls : 1.0;
        qv = (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_RAGDOLL + k) - 1.0) * XF(c, AGX_X_EE_RANGE);
        qv = fmin(fmax(qv, (double)c.xf[jb + AGX_XJ_LOWER] * sc), (double)c.xf[jb + AGX_XJ_UPPER] * sc);
      }
      gstate[sQ + lane] = (float)qv; gstate[sQT + lane] = (float)qv;
    }
    if (lane == 0) {
      gstate[sHUMAN + 6] = 1.f; gstate[sBASE + 6] = 1.f;                                               // the world anchor of the virtual joints
      gstate[sENV + AGX_E_PLANE_FRICTION] = (float)friction; gstate_i[sENV + AGX_E_GENDER] = c.gender; gstate[sENV + AGX_E_LIMIT_SCALE] = (float)c.ls;
      if (ginfo) { ginfo[0] = 1.f; ginfo[1] = 0.f; ginfo[2] = 0.f; ginfo[3] = (float)imp; }
    }
    return 0;
  }

  // ---- posed human: one static collision body per lane, lane NHUMAN the head (mouth target) --------
  const int head_link = ((const int*)c.task)[AGX_T_HEAD_LINK];       // -1: the task has no mouth target (scratch itch)
  if (lane < nhuman || (lane == nhuman && head_link >= 0)) {
    const int link = lane < nhuman ? c.xi[XI(c, AGX_X_OFF_BODIES) + lane]
                                   : c.xi[XI(c, AGX_X_OFF_DYN) + head_link - nrobot];
    d3 p; dq q;
    rs_link_pose(c, link, p, q);
    if (lane < nhuman && c.fell) {                      // the bodies did not move while the arm fell: the fall model's record has them
      for (int k = 0; k < 7; k++) gstate[sHUMAN + 7 * lane + k] = c.fell[sHUMAN + 7 * lane + k];
    } else if (lane < nhuman) {
      float* o = gstate + sHUMAN + 7 * lane;
      o[0] = (float)p.x; o[1] = (float)p.y; o[2] = (float)p.z; o[3] = (float)q.x; o[4] = (float)q.y; o[5] = (float)q.z; o[6] = (float)q.w;
    } else {                                                                                                 // feeding.py:184-196
      const d3 t = p + dqrot(q, dld3(c.task + (c.gender ? AGX_T_MOUTH_F : AGX_T_MOUTH_M)));
      float* o = gstate + sENV + AGX_E_TARGET;
      o[0] = (float)t.x; o[1] = (float)t.y; o[2] = (float)t.z;
    }
  }

  // ---- robot start pose: IK restarts, 64 per round (robot.py:84-121) -------------------------------
  d3 tpos = dld3(c.xf + AGX_X_EE_TARGET);                                                                    // feeding.py:139
  tpos.x += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_EE + 0) - 1.0) * XF(c, AGX_X_EE_RANGE);
  tpos.y += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_EE + 1) - 1.0) * XF(c, AGX_X_EE_RANGE);
  tpos.z += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_EE + 2) - 1.0) * XF(c, AGX_X_EE_RANGE);
  const dq tquat = dld4(c.xf + AGX_X_EE_QUAT);
  const int max_restarts = XI(c, AGX_X_IK_RESTARTS), randlim_from = XI(c, AGX_X_IK_RANDLIM_FROM);
  const double thresh = XF(c, AGX_X_IK_THRESH);
  double best_q[RS_NARM], best_q2[RS_NARM], best_d = 1e300;                                                   // best_q2: the second arm of a two-armed robot
  int best_r = 0x7fffffff, restarts = max_restarts, ok = 0;
#pragma unroll
  for (int d = 0; d < RS_NARM; d++) { best_q[d] = 0.0; best_q2[d] = 0.0; }
  const bool two_arms = (xflags & 512) != 0;
  d3 tpos2 = dld3(c.xf + AGX_X_EE_TARGET2);                                                                  // arm_manipulation.py:159
  tpos2.x += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_EE2 + 0) - 1.0) * XF(c, AGX_X_EE_RANGE);
  tpos2.y += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_EE2 + 1) - 1.0) * XF(c, AGX_X_EE_RANGE);
  tpos2.z += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_EE2 + 2) - 1.0) * XF(c, AGX_X_EE_RANGE);
  const int toc_attempts = XI(c, AGX_X_TOC_ATTEMPTS);
  const bool mobile = (xflags & 8) != 0;
  double lift_q = 0.0;
  if (c.fall_stage) {
    // ---- the fall model: the robot is not placed yet (arm_manipulation.py:162 comes after the fall) -- parked out of reach, its arm at the
    // middle of its joint ranges (host/reset_arm.py arm_fall_record)
    c.base_p = dld3(c.xf + AGX_X_FALL_PARK); c.base_q = dq_ident();
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) {
      const double lower = (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_LOWER], upper = (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_UPPER];
      best_q[d] = (lower > -1e9 && upper < 1e9) ? 0.5 * (lower + upper) : 0.0;
      if (two_arms) {
        const int d2 = c.xi[AGX_X_CHAIN2 + d];
        const double lower2 = (double)c.rob[d2 * AGX_R_STRIDE + AGX_R_LOWER], upper2 = (double)c.rob[d2 * AGX_R_STRIDE + AGX_R_UPPER];
        best_q2[d] = (lower2 > -1e9 && upper2 < 1e9) ? 0.5 * (lower2 + upper2) : 0.0;
      }
    }
    ok = 1; restarts = 0; best_d = 0.0;
  }
  else if (mobile) {
    // ---- a robot on wheels (env.py:282-293): three draws for the base, one for the lift (stretch.py:58-62); placement `first_restart` has its
    // own stream, as the placements of the base pose search have
    const uint32_t stream = (uint32_t)RS_T_STREAM0 + (uint32_t)first_restart;
    const double prange = XF(c, AGX_X_TOC_POS_RANGE), yrange = XF(c, AGX_X_TOC_YAW_RANGE);
    c.base_p = dld3(c.xf + AGX_X_BASE_POS) + dmk((2.0 * rs_u01(seed_lo, seed_hi, stream, RS_T_X) - 1.0) * prange, (2.0 * rs_u01(seed_lo, seed_hi, stream, RS_T_Y) - 1.0) * prange, 0.0);
    c.base_q = dq_axis_angle(dmk(0, 0, 1), XF(c, AGX_X_TOC_YAW0) + (2.0 * rs_u01(seed_lo, seed_hi, stream, RS_T_YAW) - 1.0) * yrange);
    lift_q = XF(c, AGX_X_MOBILE_LIFT) + (2.0 * rs_u01(seed_lo, seed_hi, stream, RS_T_REST) - 1.0) * 0.1;
    ok = 1; restarts = 0; best_d = 0.0;
  }
  else if (toc_attempts > 0) {
    // ---- a free-standing robot: Robot.position_robot_toc (robot.py:123-215), one candidate base pose per lane.  A candidate solves the
    // IK for the start pose (position + orientation) and for the position goals on the human's arm from random rest poses; candidates that
    // reach the start pose compete by (goals reached, summed JLWKI of the reached goals), the earliest one wins ties (robot.py:204: >).
    // `first_restart` counts the placements that collided so far (env.py:281-308 re-draws the placement): every placement has its own streams.
    const int rounds = XI(c, AGX_X_TOC_ROUNDS), titers = XI(c, AGX_X_TOC_IK_ITERS);
    const double tthr = XF(c, AGX_X_TOC_THRESH), prange = XF(c, AGX_X_TOC_POS_RANGE), yrange = XF(c, AGX_X_TOC_YAW_RANGE);
    d3 goals[4]; const bool goal_orient = XI(c, AGX_X_TOC_GOAL_ORIENT) != 0;
    const int ngoals = XI(c, AGX_X_TOC_NGOALS);
    for (int k = 0; k < 4; k++) goals[k] = dmk(0, 0, 0);
    const int goal_kind = XI(c, AGX_X_TOC_GOAL_KIND);
    if (goal_kind == 1 || goal_kind == 2) {      // feeding: the mouth (feeding.py:142, 184-196)
      d3 hp; dq hq; rs_link_pose(c, c.xi[XI(c, AGX_X_OFF_DYN) + head_link - nrobot], hp, hq);
      goals[0] = hp + dqrot(hq, dld3(c.task + (c.gender ? AGX_T_MOUTH_F : AGX_T_MOUTH_M)));
      // drinking (kind 2, drinking.py:143): start_pos_orient = [(start pose), (mouth, None)] -- BOTH must be reached for a base pose to count
      // (robot.py:196-200) -- and the mouth with the start pose's end-effector orientation as the one further goal
      goals[1] = goals[0];
    } else
      for (int k = 0; k < ngoals; k++) { dq gq; rs_link_pose(c, k < 3 ? XI(c, AGX_X_TOC_GOAL_LINKS + k) : XI(c, AGX_X_TOC_GOAL_LINK3), goals[k], gq); goals[k] = goals[k] + dld3(c.xf + AGX_X_TOC_GOAL_OFF); }
    const int nped = XI(c, AGX_X_PED_N);
    const d3 base0 = dld3(c.xf + AGX_X_BASE_POS);
    restarts = 0;
    for (int round = 0; round < rounds && !ok; round++) {
      const uint32_t stream = (uint32_t)RS_T_STREAM0 + 64u * (uint32_t)(first_restart * rounds + round) + (uint32_t)lane;
      const bool active = lane < toc_attempts;
      const double ux = rs_u01(seed_lo, seed_hi, stream, RS_T_X), uy = rs_u01(seed_lo, seed_hi, stream, RS_T_Y), uw = rs_u01(seed_lo, seed_hi, stream, RS_T_YAW);
      const double yaw = XF(c, AGX_X_TOC_YAW0) + (2.0 * uw - 1.0) * yrange;
      c.base_p = base0 + dmk(XF(c, AGX_X_TOC_X_SIGN) * prange * ux, (2.0 * uy - 1.0) * prange, 0.0);
      c.base_q = dq_axis_angle(dmk(0, 0, 1), yaw);
      int reached = 0; double manip = 0.0, qs[RS_NARM], qs2[RS_NARM];
#pragma unroll
      for (int d = 0; d < RS_NARM; d++) { qs[d] = 0.0; qs2[d] = 0.0; }
      // a two-armed robot (arm_manipulation.py:165): arm 0 = the right arm with the start pose and the first half of the goals, arm 1 = the left arm
      // (CHAIN2, EE2) with its own start pose tpos2 and the second half; bit 3 arm + g of `reached`, rest-pose draws at RS_T_REST + 8 (3 arm + g)
      const int narms = two_arms ? 2 : 1, gpa = two_arms ? ngoals / 2 : ngoals;      // goals per arm
      if (active) for (int arm = 0; arm < narms; arm++) {
        ResetCtx ca = arm ? rs_second_arm(c) : c;
        ca.base_p = c.base_p; ca.base_q = c.base_q;
        double lo[RS_NARM], hi[RS_NARM];
#pragma unroll
        for (int d = 0; d < RS_NARM; d++) {
          const double lower = (double)c.rob[ca.chain[d] * AGX_R_STRIDE + AGX_R_LOWER], upper = (double)c.rob[ca.chain[d] * AGX_R_STRIDE + AGX_R_UPPER];
          lo[d] = lower < -1e9 ? -6.283185307179586 : lower; hi[d] = upper > 1e9 ? 6.283185307179586 : upper;     // agent.py:223-231
        }
        for (int g = 0; g <= gpa; g++) {
          const int slot = (two_arms ? 3 * arm : 0) + g;
          double q[RS_NARM];
#pragma unroll
          for (int d = 0; d < RS_NARM; d++) q[d] = lo[d] + (hi[d] - lo[d]) * rs_u01(seed_lo, seed_hi, stream, RS_T_REST + 8 * slot + d);     // agent.py:263
          const d3 tp = g == 0 ? (arm ? tpos2 : tpos) : goals[arm * gpa + g - 1];
          const bool orient = g == 0 || goal_orient || (goal_kind == 2 && g == 2);
          const dq tq = (g == 0 || goal_kind == 2 || !goal_orient) ? tquat : dld4(c.xf + AGX_X_TOC_GOAL_QUAT + 4 * (g - 1));       // (position-only goals do not read it)
          if (orient) rs_ik<true>(ca, q, lo, hi, tp, tq, titers); else rs_ik<false>(ca, q, lo, hi, tp, tq, titers);
          d3 pos[RS_NARM], axw[RS_NARM], pe; dq oe;
          rs_arm_fk(ca, q, pos, axw, pe, oe);
          bool hit = sqrt(ddot(tp - pe, tp - pe)) < tthr;                                                          // robot.py:97
          if (orient) {
            const double mx = tq.x - oe.x, my = tq.y - oe.y, mz = tq.z - oe.z, mw = tq.w - oe.w;
            const double px = tq.x + oe.x, py = tq.y + oe.y, pz = tq.z + oe.z, pw = tq.w + oe.w;
            hit = hit && fmin(sqrt(mx * mx + my * my + mz * mz + mw * mw), sqrt(px * px + py * py + pz * pz + pw * pw)) < tthr;
          }
          if (g == 0) {
#pragma unroll
            for (int d = 0; d < RS_NARM; d++) { if (arm) qs2[d] = q[d]; else qs[d] = q[d]; }
            // the pedestal guard: joint origins past the shoulder, the midpoints between them and the end effector, in the base frame
            for (int b = 0; b < nped && hit; b++) {
              const float* bx = c.xf + AGX_X_PED_BOX + 6 * b;
              dq bi_ = c.base_q; bi_.x = -bi_.x; bi_.y = -bi_.y; bi_.z = -bi_.z;
              for (int k = 0; k < 10 && hit; k++) {
                const d3 pt = k < 5 ? pos[2 + k] : (k < 9 ? dmk(0.5 * (pos[k - 3].x + pos[k - 2].x), 0.5 * (pos[k - 3].y + pos[k - 2].y), 0.5 * (pos[k - 3].z + pos[k - 2].z)) : pe);
                const d3 l = dqrot(bi_, pt - c.base_p);
                if (l.x >= bx[0] && l.y >= bx[1] && l.z >= bx[2] && l.x <= bx[3] && l.y <= bx[4] && l.z <= bx[5]) hit = false;
              }
            }
          }
          if (hit) { reached |= 1 << slot; manip += rs_jlwki(ca, q); }
        }
      }
      const int must = goal_kind == 2 ? 3 : (two_arms ? 9 : 1);                                    // the start goals must be reachable (robot.py:196-200; both arms': bits 0 and 3)
      const int ngoal = (active && (reached & must) == must) ? __builtin_popcount(reached) : -1;
      int bl = 0, bn = -2; double bm = -1e300;
      for (int l = 0; l < AGX_WAVE; l++) {
        const int nl = wave_bcast_i(ngoal, l); const double ml = wave_bcast_d(manip, l);
        if (nl > bn || (nl == bn && nl > 0 && ml > bm)) { bn = nl; bm = ml; bl = l; }
      }
      restarts++;
      if (bn > 0) ok = 1;
      if (bn > 0 || round == rounds - 1) {           // the winner (or, if nobody reaches the start pose in any round, candidate 0 of the last one)
        const d3 bp = c.base_p; const dq bq = c.base_q;
        c.base_p = dmk(wave_bcast_d(bp.x, bl), wave_bcast_d(bp.y, bl), wave_bcast_d(bp.z, bl));
        c.base_q.x = wave_bcast_d(bq.x, bl); c.base_q.y = wave_bcast_d(bq.y, bl); c.base_q.z = wave_bcast_d(bq.z, bl); c.base_q.w = wave_bcast_d(bq.w, bl);
#pragma unroll
        for (int d = 0; d < RS_NARM; d++) { best_q[d] = wave_bcast_d(qs[d], bl); best_q2[d] = wave_bcast_d(qs2[d], bl); }
        best_d = (double)bn; best_r = first_restart;
      }
    }
  } else
  for (int r0 = 0; r0 < max_restarts && !ok; r0 += AGX_WAVE) {
    const int r = r0 + lane;
    const bool active = r < max_restarts;
    double q[RS_NARM], dpos = 1e300, dor = 1e300;
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) q[d] = 0.0;
    if (active) {
      double lo[RS_NARM], hi[RS_NARM];
#pragma unroll
      for (int d = 0; d < RS_NARM; d++) {
        const double lower = (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_LOWER], upper = (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_UPPER];
        double l = lower < -1e9 ? -6.283185307179586 : lower, h = upper > 1e9 ? 6.283185307179586 : upper;       // agent.py:223-231
        if (r >= randlim_from) {                                                                                 // robot.py:91
          l *= rs_u01(seed_lo, seed_hi, 1u + (uint32_t)r, RS_R_LO + d);
          h *= rs_u01(seed_lo, seed_hi, 1u + (uint32_t)r, RS_R_HI + d);
        }
        q[d] = l + (h - l) * rs_u01(seed_lo, seed_hi, 1u + (uint32_t)r, RS_R_REST + d);                            // agent.py:263
        lo[d] = fmin(l, h); hi[d] = fmax(l, h);
      }
      rs_ik<true>(c, q, lo, hi, tpos, tquat, XI(c, AGX_X_IK_ITERS));
#pragma unroll
      for (int d = 0; d < RS_NARM; d++)                                                                          // set_joint_angles(use_limits=True)
        q[d] = fmin(fmax(q[d], (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_LOWER]), (double)c.rob[c.chain[d] * AGX_R_STRIDE + AGX_R_UPPER]);
      d3 pos[RS_NARM], axw[RS_NARM], pe; dq oe;
      rs_arm_fk(c, q, pos, axw, pe, oe);
      dpos = sqrt(ddot(tpos - pe, tpos - pe));
      const double mx = tquat.x - oe.x, my = tquat.y - oe.y, mz = tquat.z - oe.z, mw = tquat.w - oe.w;
      const double px = tquat.x + oe.x, py = tquat.y + oe.y, pz = tquat.z + oe.z, pw = tquat.w + oe.w;
      dor = fmin(sqrt(mx * mx + my * my + mz * mz + mw * mw), sqrt(px * px + py * py + pz * pz + pw * pw));
      const bool rejected = r < first_restart && dpos < thresh && dor < thresh;      // met the thresholds earlier, collided
      if (dpos < best_d && !rejected) {
        best_d = dpos; best_r = r;
#pragma unroll
        for (int d = 0; d < RS_NARM; d++) best_q[d] = q[d];
      }
    }
    const uint64_t hit = wave_ballot(active && r >= first_restart && dpos < thresh && dor < thresh);            // robot.py:97
    if (hit) {
      const int l = ffs64(hit);
      ok = 1; restarts = r0 + l + 1;
      best_d = wave_bcast_d(dpos, l);
#pragma unroll
      for (int d = 0; d < RS_NARM; d++) best_q[d] = wave_bcast_d(q[d], l);
    }
  }
  if (!ok && toc_attempts == 0) {   // no restart met the thresholds: the one with the smallest position error, earliest on ties (robot.py:100-103)
    double md = 1e300; int mr = 0x7fffffff, ml = 0;
    for (int l = 0; l < AGX_WAVE; l++) {
      const double dl = wave_bcast_d(best_d, l); const int rl = wave_bcast_i(best_r, l);
      if (dl < md || (dl == md && rl < mr)) { md = dl; mr = rl; ml = l; }
    }
    best_d = md;
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) best_q[d] = wave_bcast_d(best_q[d], ml);
  }

  // ---- write the record (every lane now holds the chosen arm pose) ---------------------------------
  d3 pos[RS_NARM], axw[RS_NARM], pe, tp; dq oe, tq;
  const int lift_dof = mobile ? XI(c, AGX_X_MOBILE_LIFT_DOF) : -1;
  if (mobile) rs_mobile_fk(c, lift_dof, lift_q, pe, oe);
  else rs_arm_fk(c, best_q, pos, axw, pe, oe);
  dcompose(pe, oe, dld3(c.task + AGX_T_TOOL_POS), dld4(c.task + AGX_T_TOOL_QUAT), tp, tq);                   // tool.py:49-62
  d3 tp2 = tp; dq tq2 = tq;                                                                                  // tool_left in the left hand (arm_manipulation.py:156)
  if (two_arms) {
    ResetCtx c2 = rs_second_arm(c);
    d3 pos2[RS_NARM], axw2[RS_NARM], pe2; dq oe2;
    rs_arm_fk(c2, best_q2, pos2, axw2, pe2, oe2);
    dcompose(pe2, oe2, dld3(c.task + AGX_T_TOOL2_POS), dld4(c.task + AGX_T_TOOL2_QUAT), tp2, tq2);
  }
  if (lane < ndof) {
    double qv;
    int ck = -1, ck2 = -1;
#pragma unroll
    for (int d = 0; d < RS_NARM; d++) { ck = (!mobile && c.chain[d] == lane) ? d : ck; ck2 = (two_arms && c.xi[AGX_X_CHAIN2 + d] == lane) ? d : ck2; }
    if (lane == lift_dof) qv = lift_q;
    else if (ck >= 0) {
      qv = best_q[0];
#pragma unroll
      for (int d = 1; d < RS_NARM; d++) qv = ck == d ? best_q[d] : qv;
    } else if (ck2 >= 0) {
      qv = best_q2[0];
#pragma unroll
      for (int d = 1; d < RS_NARM; d++) qv = ck2 == d ? best_q2[d] : qv;
    } else if (lane < nrobot) {                                                                              // gripper (and joints outside the arm), feeding.py:143-144
      const float* r = c.rob + lane * AGX_R_STRIDE;
      qv = fmin(fmax((double)r[AGX_R_QT0], (double)r[AGX_R_LOWER]), (double)r[AGX_R_UPPER]);
    } else {
      const int k = lane - nrobot;
      qv = rs_joint_angle(c, c.xi[XI(c, AGX_X_OFF_DYN) + k]);
      gstate[sTREMOR + k] = imp == RS_IMP_TREMOR ? (float)((2.0 * rs_u01(seed_lo, seed_hi, 0, RS_TREMOR + k) - 1.0) * XF(c, AGX_X_TREMOR_RANGE)) : 0.f;   // human.py:89-90
      gstate[sTREMOR + nhdof + k] = (float)qv;                                                               // human.py:123
    }
    gstate[sQ + lane] = (float)qv;
    gstate[sQT + lane] = (float)qv;
    if (c.fell && lane >= nrobot) {     // the arm keeps the velocity it fell with, its hold the pose it was given before the fall (host/reset_arm.py post_fall)
      gstate[sQD + lane] = c.fell[sQD + lane]; gstate[sQT + lane] = c.fell[sQT + lane];
      gstate[sTREMOR + nhdof + lane - nrobot] = c.fell[sTREMOR + nhdof + lane - nrobot];
    }
  }
  if (lane < nfree) {
    float* o = gstate + sFREE + 13 * lane;
    const int tool_body = c.bi[AGX_H_TOOL_BODY], bowl_body = XI(c, AGX_X_BOWL_BODY), food0 = c.bi[AGX_H_FOOD0];
    d3 p = dmk(0, 0, 0); dq q = dq_ident();
    const int tool2_body = two_arms ? ((const int*)c.task)[AGX_T_TOOL2_BODY] : -1;
    if (lane == tool_body || lane == tool2_body) {
      const d3 tpx = lane == tool2_body ? tp2 : tp; const dq tqx = lane == tool2_body ? tq2 : tq;
      p = tpx; q = tqx;
      const float* fb = c.bf + c.bi[AGX_H_OFF_FREE] + lane * AGX_F_STRIDE;
      if (fb[AGX_F_REFPOS] != 0.f || fb[AGX_F_REFPOS + 1] != 0.f || fb[AGX_F_REFPOS + 2] != 0.f || fb[AGX_F_REFQUAT + 3] != 1.f) {
        // a tool whose URDF base frame is not its centre-of-mass frame (wiper, scratcher): the record holds the COM frame
        dq qi = dld4(fb + AGX_F_REFQUAT); qi.x = -qi.x; qi.y = -qi.y; qi.z = -qi.z;
        const d3 ip = dqrot(qi, dld3(fb + AGX_F_REFPOS));
        dcompose(tpx, tqx, dmk(-ip.x, -ip.y, -ip.z), qi, p, q);
      }
    }
    else if (lane == bowl_body) {                                                                            // furniture.py:32-34
      const double br = XF(c, AGX_X_BOWL_RANGE);
      d3 b = dld3(c.xf + AGX_X_BOWL_POS);
      b.x += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_BOWL + 0) - 1.0) * br;
      b.y += (2.0 * rs_u01(seed_lo, seed_hi, 0, RS_BOWL + 1) - 1.0) * br;
      const float* fb = c.bf + c.bi[AGX_H_OFF_FREE] + lane * AGX_F_STRIDE;
      dq qi = dld4(fb + AGX_F_REFQUAT); qi.x = -qi.x; qi.y = -qi.y; qi.z = -qi.z;
      const d3 ip = dqrot(qi, dld3(fb + AGX_F_REFPOS));
      dcompose(b, dq_ident(), dmk(-ip.x, -ip.y, -ip.z), qi, p, q);
    } else if (lane >= food0 && lane < food0 + nfood) {                                                      // feeding.py:158-166
      const int k = lane - food0;
      const double two_r = 2.0 * XF(c, AGX_X_FOOD_R);
      p = dmk((double)(k >> 2 & 1) * two_r, (double)(k >> 1 & 1) * two_r, (double)(k & 1) * two_r) + dld3(c.xf + AGX_X_FOOD_OFF) + tp;
    }
    o[0] = (float)p.x; o[1] = (float)p.y; o[2] = (float)p.z; o[3] = (float)q.x; o[4] = (float)q.y; o[5] = (float)q.z; o[6] = (float)q.w;
  }
  if (lane == 0) {
    float* o = gstate + sBASE;
    o[0] = (float)c.base_p.x; o[1] = (float)c.base_p.y; o[2] = (float)c.base_p.z; o[3] = (float)c.base_q.x; o[4] = (float)c.base_q.y; o[5] = (float)c.base_q.z; o[6] = (float)c.base_q.w;
  }
  if (lane == 0) {
    float* e = gstate + sENV; int* ei = gstate_i + sENV;
    const uint64_t seed = ((uint64_t)seed_hi << 32) | seed_lo;
    e[AGX_E_PLANE_FRICTION] = (float)friction;
    ei[AGX_E_GENDER] = c.gender;
    ei[AGX_E_FOOD_ALIVE] = (1 << nfood) - 1; ei[AGX_E_FOOD_ACTIVE] = (1 << nfood) - 1;
    ei[AGX_E_ITERATION] = 0; ei[AGX_E_TASK_SUCCESS] = 0;
    ei[AGX_E_RNG] = (int)((seed * 2654435761ull + 12345ull) & 0x7FFFFFFFull);
    ei[AGX_E_RNG + 1] = (int)((seed ^ 0x5bd1e995ull) & 0x7FFFFFFFull);
    ei[AGX_E_TOTAL_FOOD] = (xflags & (6 | 128)) ? 1 : nfood;       // scratch itch, dressing, arm manipulation: task_success >= 1 x task_success_threshold (scratch_itch.py:37)
    if (c.bi[AGX_H_TASK_KIND] == AGX_TASK_DRINKING) {               // self.waters / self.waters_active: every particle (drinking.py:168-172)
      const int nw = c.bi[c.bi[AGX_H_OFF_CLOTH] + AGX_CL_NN];
      ei[AGX_E_TOTAL_FOOD] = nw;                                    // total_water_count
      unsigned* tw = (unsigned*)(gstate + c.bi[AGX_H_S_TASK]);
      for (int t = 0; t < 2; t++) { const unsigned m = nw >= 32 * (t + 1) ? 0xffffffffu : (nw > 32 * t ? (1u << (nw - 32 * t)) - 1u : 0u); tw[AGX_DK_ALIVE + t] = m; tw[AGX_DK_ACTIVE + t] = m; }
    }
    if (xflags & 64) {   // bed bathing: generate_targets (bed_bathing.py:173-188) -- every target of this gender's two tables is alive
      const int* nt4 = (const int*)c.task + AGX_T_NT;
      const int nt = nt4[2 * c.gender] + nt4[2 * c.gender + 1];
      ei[AGX_E_TOTAL_FOOD] = nt;                                    // total_target_count (bed_bathing.py:187)
      unsigned* alive = (unsigned*)(gstate + c.bi[AGX_H_S_TASK]) + AGX_BB_ALIVE;
      for (int t = 0; t < AGX_BB_ALIVE_WORDS; t++) alive[t] = nt >= 32 * (t + 1) ? 0xffffffffu : (nt > 32 * t ? (1u << (nt - 32 * t)) - 1u : 0u);
    }
    const bool coop = ((const int*)c.task)[AGX_T_COOP] == 1;
    const bool agent = imp == RS_IMP_TREMOR || coop;                // then take_step drives the human's motors (env.py:130-131)
    ei[AGX_E_FROZEN] = (agent || (xflags & 1)) ? 0 : (((1 << nhdof) - 1) << nrobot);                          // human.py:104-110
    e[AGX_E_LIMIT_SCALE] = (float)c.ls;
    if (!agent && XF(c, AGX_X_REACTIVE_KP) > 0.0) {                 // the reactive hold of setup_joints (human.py:124-127)
      e[AGX_E_HUMAN_KP] = c.xf[AGX_X_REACTIVE_KP]; e[AGX_E_HUMAN_MAXF] = (float)(XF(c, AGX_X_REACTIVE_MAXF) * strength);
    }
    if (xflags & 2) {   // generate_target (scratch_itch.py:134-146): limb, then Util.point_on_capsule (util.py:58-78) along (0, 0, -length)
      const int limb = rs_u01(seed_lo, seed_hi, 0, RS_LIMB) < 0.5 ? 0 : 1;
      const double length = (double)c.task[AGX_T_SI_LIMB_DIMS + 4 * c.gender + 2 * limb], radius = (double)c.task[AGX_T_SI_LIMB_DIMS + 4 * c.gender + 2 * limb + 1];
      const double rl = radius + (length - radius) * rs_u01(seed_lo, seed_hi, 0, RS_TARGET_LEN);
      const double th = 6.283185307179586 * rs_u01(seed_lo, seed_hi, 0, RS_TARGET_TH);
      // axis (0, 0, -1), Util.orthogonal_vector -> (0, -1, 0), normal = axis x ortho = (-1, 0, 0)
      float* tw = gstate + c.bi[AGX_H_S_TASK];
      tw[AGX_SI_TARGET] = (float)(-radius * sin(th)); tw[AGX_SI_TARGET + 1] = (float)(-radius * cos(th)); tw[AGX_SI_TARGET + 2] = (float)(-rl);
      ((int*)tw)[AGX_SI_LIMB] = limb;
    }
    if (xflags & 4) {   // dressing: the garment is loaded shifted to the end effector and settles under half gravity (dressing.py:146-149,178)
      float* tw = gstate + c.bi[AGX_H_S_TASK];
      tw[AGX_DR_CLOTH_GRAVITY] = c.xf[AGX_X_CLOTH_GRAVITY_SETTLE];
      tw[AGX_DR_CLOTH_OFF] = (float)(pe.x - XF(c, AGX_X_CLOTH_ORIG_POS)); tw[AGX_DR_CLOTH_OFF + 1] = (float)(pe.y - XF(c, AGX_X_CLOTH_ORIG_POS + 1));
      tw[AGX_DR_CLOTH_OFF + 2] = (float)(pe.z - XF(c, AGX_X_CLOTH_ORIG_POS + 2));
    }
    if (ginfo) { ginfo[0] = (float)ok; ginfo[1] = (float)restarts; ginfo[2] = (float)best_d; ginfo[3] = (float)imp; }
  }
  if (toc_attempts > 0 || mobile) return ok ? first_restart : -1;   // the placement that was accepted (the next one, if it collides, is first_restart + 1)
  return ok ? restarts - 1 : -1;
}

// Collision verdict on a freshly sampled state whose contacts the build kernel has just written to the per-env scratch record:
// does a robot link (robot.py:105-112: get_closest_points(obj, distance=0)) or the tool (env.py:300-304) touch one of
// collision_objects = [human, table, wheelchair] (feeding.py:141)?  Wave-uniform result.
AGX_DEV bool reset_collides(const uint32_t* __restrict__ blob, const float* __restrict__ gscratch, int lane) {
  const int* bi = (const int*)blob;
  const int* meta = (const int*)(gscratch + SCR_O_META);
  const int ncon = meta[META_NCON];
  bool bad = false;
  if (lane < ncon) {
    const float* k = gscratch + SCR_O_CON + CON_STRIDE * lane; const int* ki = (const int*)k;
    const int ta = bi[bi[AGX_H_OFF_COLL] + ki[C_CA] * AGX_C_STRIDE + AGX_C_TAG], tb = bi[bi[AGX_H_OFF_COLL] + ki[C_CB] * AGX_C_STRIDE + AGX_C_TAG];
    const bool ra = ta == AGX_TAG_ROBOT || ta == AGX_TAG_TOOL, rb = tb == AGX_TAG_ROBOT || tb == AGX_TAG_TOOL;
    const bool oa = ta == AGX_TAG_HUMAN || ta == AGX_TAG_TABLE || ta == AGX_TAG_WHEELCHAIR, ob = tb == AGX_TAG_HUMAN || tb == AGX_TAG_TABLE || tb == AGX_TAG_WHEELCHAIR;
    bad = ((ra && ob) || (rb && oa)) && k[C_DIST] <= 0.f;
  }
  return wave_any(bad);
}

#undef XF
#undef XI

}  // namespace agx
