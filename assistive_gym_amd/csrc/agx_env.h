// agx_env.h -- K7 integration and hooks, state load / store, the task layer (observation, food state machine, reward) and the kernel bodies.
// Part of the stepper (see agx_step.h for the overview); included by agx_step.h only.
#pragma once

namespace agx {

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
    if (c.hooks && (RBI(c, d, AGX_R_KIND) & 5) == 1 && !FROZEN(c, d)) {
      const float lo = DLO(c, d), hi = DHI(c, d);
      if (q < lo - AGX_LIMIT_EPS) { q = lo; qd = 0.f; } else if (q > hi + AGX_LIMIT_EPS) { q = hi; qd = 0.f; }
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
// Human.enforce_realistic_joint_limits (human.py:134-152) after Agent.enforce_joint_limits (env.py:229-231): the four arm angles,
// remapped as the training data was (human.py:142-145), go through the Keras classifier Dense(4->64, tanh) x3 -> Dense(64->1,
// sigmoid) (assets/realistic_arm_limits_model.h5, weights in the blob's MLP section).  Class 1 (sigma > 0.5 <=> logit > 0): the pose
// is remembered; class 0: the four joints are put back to the last valid pose with zero velocity (set_joint_angles, agent.py:154-156).
// lane = hidden unit; the activations of a layer are exchanged through 64 words of LDS (`X`, dead storage of the caller).
AGX_DEV void arm_limits(Ctx& c, float* X) {
  if (!TKI(c, AGX_T_ARM_LIMIT_ON) || !c.hooks) return;
  float* L = c.lds; int* Li = c.ldsi; const int lane = c.lane;
  const float* W1 = c.bf + c.bi[AGX_H_OFF_MLP]; const float* W2 = W1 + 4 * 64 + 64; const float* W3 = W2 + 64 * 64 + 64; const float* W4 = W3 + 64 * 64 + 64;
  const int s_task = c.bi[AGX_H_S_TASK];
  const float sg = TKF(c, AGX_T_ARM_LIMIT_SIGN), TWO_PI = 6.28318530717959f;
  int dof[4]; float a[4];
  for (int k = 0; k < 4; k++) {
    dof[k] = TKI(c, AGX_T_ARM_LIMIT_DOF + k);
    // the reference reads the angles after its strict limit reset; here the reset has the tolerance AGX_LIMIT_EPS, so clamp what is read
    a[k] = wave_clamp(L[L_ST + c.s_q + dof[k]], DLO(c, dof[k]), DHI(c, dof[k]));
  }
  const float x0 = sg * a[0] + TWO_PI, x1 = a[1] + TWO_PI, x2 = sg * a[2], x3 = -a[3] + TWO_PI;
  const float in0 = x0 - TWO_PI * floorf(x0 / TWO_PI), in1 = x1 - TWO_PI * floorf(x1 / TWO_PI), in3 = x3 - TWO_PI * floorf(x3 / TWO_PI);
  float h = W1[256 + lane] + in0 * W1[lane] + in1 * W1[64 + lane] + x2 * W1[128 + lane] + in3 * W1[192 + lane];
  h = tanhf(h);
  for (int layer = 0; layer < 2; layer++) {
    const float* W = layer == 0 ? W2 : W3;
    wave_sync(); X[lane] = h; wave_sync();
    float acc = W[4096 + lane];
    for (int k = 0; k < 64; k++) acc += X[k] * W[64 * k + lane];
    h = tanhf(acc);
  }
  const float z = wave_sum(h * W4[lane]) + W4[64];
  wave_sync();
  if (lane == 0) {
    if (z > 0.f) {
      for (int k = 0; k < 4; k++) L[L_ST + s_task + AGX_BB_PREV + k] = L[L_ST + c.s_q + dof[k]];
      Li[L_ST + s_task + AGX_BB_HAS_PREV] = 1;
    } else if (Li[L_ST + s_task + AGX_BB_HAS_PREV]) {
      for (int k = 0; k < 4; k++) {
        L[L_ST + c.s_q + dof[k]] = wave_clamp(L[L_ST + s_task + AGX_BB_PREV + k], DLO(c, dof[k]), DHI(c, dof[k]));
        L[L_ST + c.s_qd + dof[k]] = 0.f;
      }
    }
  }
  wave_sync();
}
// FeedingEnv.update_targets (feeding.py:192-196): mouth = head pose o mouth offset.  Needs the link
// frames of a preceding kinematics(); the target is only consumed by the observation / reward code.
AGX_DEV void update_target(Ctx& c) {
  float* L = c.lds;
  wave_sync();
  if ((TASK == AGX_TASK_FEEDING || TASK == AGX_TASK_DRINKING) && c.lane == 0) {      // (drinking.py:192-196: the same target, and the head moves there)
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
  if (lane < c.ndof) {
    uint64_t m = 0; for (int d = lane; d >= 0; d = RBI(c, d, AGX_R_PARENT)) m |= 1ull << d;
    c.ldsi[L_MISC + M_ANC + ANC_WORDS * lane] = (int)(uint32_t)m;
    if (ANC_WORDS == 2) c.ldsi[L_MISC + M_ANC + 2 * lane + 1] = (int)(uint32_t)(m >> 32);
  }
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
// Robot.get_base_pos_orient() (agent.py:142-150), the frame of convert_to_realworld: the pose of the state record, or the moving link
// that is the base link of a robot on a floating base (AGX_H_BASE_LINK)
AGX_DEV void robot_base_pose(const Ctx& c, v3& p, m3& R) {
  const float* L = c.lds; const int bl = c.bi[AGX_H_BASE_LINK];
  if (bl > 0) { p = ld3(L + L_LINKP + 3 * (bl - 1)); R = ldm3(L + L_LINKR + 9 * (bl - 1)); } else { p = ld3(L + L_BASE); R = ldm3(L + L_BASE + 3); }
}
// pose of the tool frame the task reads: the base frame of the spoon (feeding.py:86), link 1 of the wiper (bed_bathing.py:81)
AGX_DEV void tool_base_pose_of(const Ctx& c, int tb, v3& p, m3& R);
AGX_DEV void tool_base_pose(const Ctx& c, v3& p, m3& R) { tool_base_pose_of(c, c.bi[AGX_H_TOOL_BODY], p, R); }
AGX_DEV void tool_base_pose_of(const Ctx& c, int tb, v3& p, m3& R) {
  const float* L = c.lds;
  m3 FR = ldm3(L + L_FREER + 9 * tb); v3 fp = ld3(L + L_ST + c.s_free + 13 * tb);
  p = mul(FR, mk3(FBF(c, tb, AGX_F_REFPOS), FBF(c, tb, AGX_F_REFPOS + 1), FBF(c, tb, AGX_F_REFPOS + 2))) + fp;
  R = mul(FR, quat_to_m3(FBF(c, tb, AGX_F_REFQUAT), FBF(c, tb, AGX_F_REFQUAT + 1), FBF(c, tb, AGX_F_REFQUAT + 2), FBF(c, tb, AGX_F_REFQUAT + 3)));
  if constexpr (TASK != AGX_TASK_FEEDING && TASK != AGX_TASK_DRINKING) {      // (the cup is observed in its base frame; AGX_T_TOOL_OBS_* is the frame of its reward terms)
    p = mul(R, mk3(TKF(c, AGX_T_TOOL_OBS_POS), TKF(c, AGX_T_TOOL_OBS_POS + 1), TKF(c, AGX_T_TOOL_OBS_POS + 2))) + p;
    R = mul(R, quat_to_m3(TKF(c, AGX_T_TOOL_OBS_QUAT), TKF(c, AGX_T_TOOL_OBS_QUAT + 1), TKF(c, AGX_T_TOOL_OBS_QUAT + 2), TKF(c, AGX_T_TOOL_OBS_QUAT + 3)));
  }
}
// BedBathingEnv._get_obs (bed_bathing.py:80-110); every lane computes, lane 0 writes.
// tool_force = all contacts of the tool, total_force = total_force_on_human, pad_force = tool_force_on_human
AGX_DEV void observe_bed(const Ctx& c, float tool_force, float total_force, float pad_force, float* gobs) {
  const float* L = c.lds;
  v3 bp; m3 BR; robot_base_pose(c, bp, BR);
  v3 sp; m3 sR; tool_base_pose(c, sp, sR);
  v3 spr = tmul(BR, sp - bp); q4 sq = m3_to_quat(mul_at(BR, sR));
  v3 jp[3], jpr[3];
  for (int k = 0; k < 3; k++) { jp[k] = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + k)); jpr[k] = tmul(BR, jp[k] - bp); }
  if (c.lane == 0) {
    int o = 0;
    gobs[o++] = spr.x; gobs[o++] = spr.y; gobs[o++] = spr.z;
    gobs[o++] = sq.x; gobs[o++] = sq.y; gobs[o++] = sq.z; gobs[o++] = sq.w;
    for (int d = 0; d < c.nrobot; d++) if (RBI(c, d, AGX_R_ACT) >= 0 && !RBI(c, d, AGX_R_OBS_SKIP)) {
      float a = L[L_ST + c.s_q + d] + 3.14159265358979f;
      gobs[o++] = (a - 6.28318530717959f * floorf(a / 6.28318530717959f)) - 3.14159265358979f;
    }
    for (int k = 0; k < 3; k++) { gobs[o++] = jpr[k].x; gobs[o++] = jpr[k].y; gobs[o++] = jpr[k].z; }
    gobs[o++] = tool_force;
    if (c.coop) {   // human_obs (bed_bathing.py:101-106), in the frame of the human's base (collision body 0)
      const v3 hb = ld3(L + L_HUMAN); const m3 HR = ldm3(L + L_HUMAN + 3);
      const v3 sph = tmul(HR, sp - hb); const q4 sqh = m3_to_quat(mul_at(HR, sR));
      gobs[o++] = sph.x; gobs[o++] = sph.y; gobs[o++] = sph.z;
      gobs[o++] = sqh.x; gobs[o++] = sqh.y; gobs[o++] = sqh.z; gobs[o++] = sqh.w;
      for (int d = c.nrobot; d < c.ndof; d++) if (RBI(c, d, AGX_R_ACT) >= 0) gobs[o++] = L[L_ST + c.s_q + d];
      for (int k = 0; k < 3; k++) { const v3 h = tmul(HR, jp[k] - hb); gobs[o++] = h.x; gobs[o++] = h.y; gobs[o++] = h.z; }
      gobs[o++] = total_force; gobs[o++] = pad_force;
    }
  }
}
// ArmManipulationEnv._get_obs (arm_manipulation.py:71-110).  Single-arm robot: the one tool is tool_right AND tool_left (:12-14) and the arm
// joints are listed twice (robot_arm = 'both', robot.py:16).  Two-armed robot (AGX_T_TOOL2_BODY > 0): tool 0 = tool_right, the second tool
// = tool_left, 14 joint angles = right arm then left arm.  Every lane computes, lane 0 writes.
// tf_r / tf_l = all contacts of the right / left tool, thf_r / thf_l = the same on the human, total_force = total_force_on_human
AGX_DEV void observe_arm(const Ctx& c, float tf_r, float tf_l, float total_force, float thf_r, float thf_l, float* gobs) {
  const float* L = c.lds;
  const int tb2 = TKI(c, AGX_T_TOOL2_BODY); const bool dual = tb2 > 0;
  v3 bp; m3 BR; robot_base_pose(c, bp, BR);
  v3 sp[2]; m3 sR[2];
  tool_base_pose(c, sp[0], sR[0]);
  if (dual) tool_base_pose_of(c, tb2, sp[1], sR[1]); else { sp[1] = sp[0]; sR[1] = sR[0]; }
  v3 jp[5], jpr[5];
  for (int k = 0; k < 3; k++) jp[k] = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + k));        // shoulder, elbow, wrist
  jp[3] = ld3(L + L_HUMAN + 12 * TKI(c, AGX_T_STOMACH_BODY)); jp[4] = ld3(L + L_HUMAN + 12 * TKI(c, AGX_T_WAIST_BODY));
  for (int k = 0; k < 5; k++) jpr[k] = tmul(BR, jp[k] - bp);
  if (c.lane == 0) {
    int o = 0;
    for (int t = 0; t < 2; t++) {
      const v3 spr = tmul(BR, sp[t] - bp); const q4 sq = m3_to_quat(mul_at(BR, sR[t]));
      gobs[o++] = spr.x; gobs[o++] = spr.y; gobs[o++] = spr.z; gobs[o++] = sq.x; gobs[o++] = sq.y; gobs[o++] = sq.z; gobs[o++] = sq.w;
    }
    for (int rep = 0; rep < (dual ? 1 : 2); rep++)
      for (int d = 0; d < c.nrobot; d++) if (RBI(c, d, AGX_R_ACT) >= 0 && !RBI(c, d, AGX_R_OBS_SKIP)) {
        float a = L[L_ST + c.s_q + d] + 3.14159265358979f;
        gobs[o++] = (a - 6.28318530717959f * floorf(a / 6.28318530717959f)) - 3.14159265358979f;
      }
    for (int k = 0; k < 5; k++) { gobs[o++] = jpr[k].x; gobs[o++] = jpr[k].y; gobs[o++] = jpr[k].z; }
    gobs[o++] = tf_l; gobs[o++] = tf_r;                 // [tool_left_force, tool_right_force] (:92)
    if (c.coop) {   // human_obs (:98-107), in the frame of the human's base (collision body 0)
      const v3 hb = ld3(L + L_HUMAN); const m3 HR = ldm3(L + L_HUMAN + 3);
      for (int t = 0; t < 2; t++) {
        const v3 sph = tmul(HR, sp[t] - hb); const q4 sqh = m3_to_quat(mul_at(HR, sR[t]));
        gobs[o++] = sph.x; gobs[o++] = sph.y; gobs[o++] = sph.z; gobs[o++] = sqh.x; gobs[o++] = sqh.y; gobs[o++] = sqh.z; gobs[o++] = sqh.w;
      }
      for (int d = c.nrobot; d < c.ndof; d++) if (RBI(c, d, AGX_R_ACT) >= 0) gobs[o++] = L[L_ST + c.s_q + d];
      for (int k = 0; k < 5; k++) { const v3 h = tmul(HR, jp[k] - hb); gobs[o++] = h.x; gobs[o++] = h.y; gobs[o++] = h.z; }
      gobs[o++] = total_force; gobs[o++] = thf_l; gobs[o++] = thf_r;
    }
  }
}
// world position of the scratch-itch target: limb frame o target_on_arm (scratch_itch.py:148-152)
AGX_DEV v3 scratch_target(const Ctx& c) {
  const float* L = c.lds; const int s_task = c.bi[AGX_H_S_TASK];
  const int link = TKI(c, AGX_T_ARM_LINK + c.ldsi[L_ST + s_task + AGX_SI_LIMB]);
  return mul(ldm3(L + L_LINKR + 9 * link), ld3(L + L_ST + s_task + AGX_SI_TARGET)) + ld3(L + L_LINKP + 3 * link);
}
// ScratchItchEnv._get_obs (scratch_itch.py:59-91); every lane computes, lane 0 writes.
// tool_force = all contacts of the tool, total_force = total_force_on_human, target_force = tool_force_at_target
AGX_DEV void observe_scratch(const Ctx& c, float tool_force, float total_force, float target_force, float* gobs) {
  const float* L = c.lds;
  v3 bp; m3 BR; robot_base_pose(c, bp, BR);
  v3 sp; m3 sR; tool_base_pose(c, sp, sR);                       // tool.get_pos_orient(1)
  v3 spr = tmul(BR, sp - bp); q4 sq = m3_to_quat(mul_at(BR, sR));
  const v3 tg = scratch_target(c), tgr = tmul(BR, tg - bp);
  v3 jp[3], jpr[3];
  for (int k = 0; k < 3; k++) { jp[k] = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + k)); jpr[k] = tmul(BR, jp[k] - bp); }
  if (c.lane == 0) {
    int o = 0;
    gobs[o++] = spr.x; gobs[o++] = spr.y; gobs[o++] = spr.z;
    gobs[o++] = sq.x; gobs[o++] = sq.y; gobs[o++] = sq.z; gobs[o++] = sq.w;
    gobs[o++] = spr.x - tgr.x; gobs[o++] = spr.y - tgr.y; gobs[o++] = spr.z - tgr.z;
    gobs[o++] = tgr.x; gobs[o++] = tgr.y; gobs[o++] = tgr.z;
    for (int d = 0; d < c.nrobot; d++) if (RBI(c, d, AGX_R_ACT) >= 0 && !RBI(c, d, AGX_R_OBS_SKIP)) {
      float a = L[L_ST + c.s_q + d] + 3.14159265358979f;
      gobs[o++] = (a - 6.28318530717959f * floorf(a / 6.28318530717959f)) - 3.14159265358979f;
    }
    for (int k = 0; k < 3; k++) { gobs[o++] = jpr[k].x; gobs[o++] = jpr[k].y; gobs[o++] = jpr[k].z; }
    gobs[o++] = tool_force;
    if (c.coop) {   // human_obs (scratch_itch.py:79-88), in the frame of the human's base (collision body 0)
      const v3 hb = ld3(L + L_HUMAN); const m3 HR = ldm3(L + L_HUMAN + 3);
      const v3 sph = tmul(HR, sp - hb); const q4 sqh = m3_to_quat(mul_at(HR, sR)); const v3 tgh = tmul(HR, tg - hb);
      gobs[o++] = sph.x; gobs[o++] = sph.y; gobs[o++] = sph.z;
      gobs[o++] = sqh.x; gobs[o++] = sqh.y; gobs[o++] = sqh.z; gobs[o++] = sqh.w;
      gobs[o++] = sph.x - tgh.x; gobs[o++] = sph.y - tgh.y; gobs[o++] = sph.z - tgh.z;
      gobs[o++] = tgh.x; gobs[o++] = tgh.y; gobs[o++] = tgh.z;
      for (int d = c.nrobot; d < c.ndof; d++) if (RBI(c, d, AGX_R_ACT) >= 0) gobs[o++] = L[L_ST + c.s_q + d];
      for (int k = 0; k < 3; k++) { const v3 h = tmul(HR, jp[k] - hb); gobs[o++] = h.x; gobs[o++] = h.y; gobs[o++] = h.z; }
      gobs[o++] = total_force; gobs[o++] = target_force;
    }
  }
}
// FeedingEnv._get_obs (feeding.py:85-112), robot part; every lane computes, lane 0 writes
AGX_DEV void observe(const Ctx& c, float robot_force, float tool_force, float* gobs) {
  const float* L = c.lds;
  v3 bp; m3 BR; robot_base_pose(c, bp, BR);
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
    for (int d = 0; d < c.nrobot; d++) if (RBI(c, d, AGX_R_ACT) >= 0 && !RBI(c, d, AGX_R_OBS_SKIP)) {
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
struct Scratch { float* ent; float* hdr; float* vel; float* con; int* meta; float* qpt; float* warm; float* man; };
AGX_DEV Scratch scratch_of(float* base) {
  Scratch s; s.ent = base + SCR_O_ENT; s.hdr = base + SCR_O_HDR; s.vel = base + SCR_O_VEL; s.con = base + SCR_O_CON; s.meta = (int*)(base + SCR_O_META); s.qpt = base + SCR_O_QPT; s.warm = base + SCR_O_WARM; s.man = base + SCR_O_MAN;
  return s;
}

// ---- warm starting (AGX_P_WARMSTART, a [BULLET-UNVERIFIED] switch, default off) -------------------------------------------------------
// key of this lane's contact: collider a | collider b << 9 | ordinal among the contacts of the same pair << 18 (the face manifold gives a pair
// up to four); lanes >= ncon return -1.  Wave-uniform call.
AGX_DEV int warm_key(const Ctx& c, int lane) {
  const bool has = lane < c.ncon;
  const int* ki = (const int*)(c.gcon + CON_STRIDE * (has ? lane : 0));
  const int pair = has ? (ki[C_CA] | (ki[C_CB] << 9)) : -1;
  int ord = 0;
  for (int j = 0; j < c.ncon; j++) { const int pj = wave_bcast_i(pair, j); if (has && j < lane && pj == pair) ord++; }
  return has ? (pair | (ord << 18)) : -1;
}
// build kernel, after the contacts of the substep are in the scratch record: C_LAM := factor x the impulse of the same contact in the memory
AGX_DEV void warm_seed(const Ctx& c, const Scratch& scr, int lane) {
  const float wsf = PRM(c, AGX_P_WARMSTART);
  if (!(wsf > 0.f)) return;
  const int nw = scr.meta[META_NWARM];
  const int key = warm_key(c, lane);
  float lam = 0.f;
  for (int p = 0; p < nw && p < MAX_CON; p++) if (((const int*)scr.warm)[p] == key) { lam = wsf * scr.warm[MAX_CON + p]; break; }
  if (lane < c.ncon) c.gcon[CON_STRIDE * lane + C_LAM] = lam;
}
// solve kernel, after the sweeps: this substep's contacts and their solved impulses become the memory
AGX_DEV void warm_remember(const Ctx& c, const Scratch& scr, int lane) {
  if (!(PRM(c, AGX_P_WARMSTART) > 0.f)) return;
  const int key = warm_key(c, lane);
  if (lane < c.ncon) { ((int*)scr.warm)[lane] = key; scr.warm[MAX_CON + lane] = c.gcon[CON_STRIDE * lane + C_LAM]; }
  if (lane == 0) scr.meta[META_NWARM] = c.ncon;
}

// ---- persistent contact manifold (AGX_P_MANIFOLD, a [BULLET-UNVERIFIED] switch, default off; include/agx_blob.h) ---------------------------
// Build kernel, after collide(): the substep's GJK contacts (c.gcon[0 .. c.ncon)) update the cached points of the environment (scratch record,
// lane = cached point while they are in registers) and the contact list is rebuilt from the cache.  Contacts on a static world box keep their
// face manifold and pass through.  Mirrors manifold_update() of oracle/agx_oracle.c step by step (same order, same tie breaks).
struct MPoint { int key; v3 la, lb, n; float dist, mu; };
AGX_DEV MPoint mp_bcast(const MPoint& e, int src) {
  MPoint r; r.key = wave_bcast_i(e.key, src);
  r.la = mk3(wave_bcast(e.la.x, src), wave_bcast(e.la.y, src), wave_bcast(e.la.z, src)); r.lb = mk3(wave_bcast(e.lb.x, src), wave_bcast(e.lb.y, src), wave_bcast(e.lb.z, src));
  r.n = mk3(wave_bcast(e.n.x, src), wave_bcast(e.n.y, src), wave_bcast(e.n.z, src)); r.dist = wave_bcast(e.dist, src); r.mu = wave_bcast(e.mu, src);
  return r;
}
AGX_DEV void manifold_update(Ctx& c, const Scratch& scr) {
  const int lane = c.lane;
  const float brk = PRM(c, AGX_P_CONTACT_BREAK), slack = PRM(c, AGX_P_CONTACT_SLACK);
  int maxc = (int)PRM(c, AGX_P_MAX_CONTACTS); if (maxc > MAX_CON) maxc = MAX_CON;
  float* MP = scr.man;
  int nm = scr.meta[META_NMAN]; if (nm > MAX_CON) nm = MAX_CON;
  // (1) refresh (lane = cached point), order-preserving compaction through the scratch record
  MPoint e; e.key = -1; e.la = mk3(0, 0, 0); e.lb = e.la; e.n = e.la; e.dist = 0.f; e.mu = 0.f;
  bool keep = false;
  if (lane < nm) {
    const float* q = MP + MP_STRIDE * lane;
    e.key = ((const int*)q)[MP_KEY]; e.la = ld3(q + MP_LA); e.lb = ld3(q + MP_LB); e.n = ld3(q + MP_N); e.mu = q[MP_MU];
    m3 Ra, Rb; v3 oa, ob; body_xf(c, CLI(c, e.key & 511, AGX_C_BODY), Ra, oa); body_xf(c, CLI(c, (e.key >> 9) & 511, AGX_C_BODY), Rb, ob);
    const v3 pa = mul(Ra, e.la) + oa, pb = mul(Rb, e.lb) + ob;
    e.dist = dot(pa - pb, e.n);
    const v3 drift = pb - (pa - e.dist * e.n);
    keep = e.dist <= brk && dot(drift, drift) <= brk * brk;
  }
  wave_sync();
  {
    const uint64_t km = wave_ballot(keep);
    if (keep) { float* q = MP + MP_STRIDE * wave_rank(km); ((int*)q)[MP_KEY] = e.key; st3(q + MP_LA, e.la); st3(q + MP_LB, e.lb); st3(q + MP_N, e.n); q[MP_DIST] = e.dist; q[MP_MU] = e.mu; }
    nm = popc64(km);
  }
  wave_sync();
  e.key = -1;
  if (lane < nm) { const float* q = MP + MP_STRIDE * lane; e.key = ((const int*)q)[MP_KEY]; e.la = ld3(q + MP_LA); e.lb = ld3(q + MP_LB); e.n = ld3(q + MP_N); e.dist = q[MP_DIST]; e.mu = q[MP_MU]; }
  // (2) merge the substep's contacts, one after the other (every lane reads the same record)
  const int nnew = c.ncon;
  for (int i = 0; i < nnew; i++) {
    const float* k = c.gcon + CON_STRIDE * i; const int* ki = (const int*)k;
    const int ca = ki[C_CA], cb = ki[C_CB];
    if (face_box(c, cb)) continue;
    MPoint q; q.key = ca | (cb << 9); q.n = ld3(k + C_N); q.dist = k[C_DIST]; q.mu = k[C_MU];
    { m3 Ra, Rb; v3 oa, ob; body_xf(c, ki[C_BA], Ra, oa); body_xf(c, ki[C_BB], Rb, ob); q.la = tmul(Ra, ld3(k + C_PA) - oa); q.lb = tmul(Rb, ld3(k + C_PB) - ob); }
    const bool match = lane < nm && e.key == q.key;
    const v3 dl = e.la - q.la; const float d2 = dot(dl, dl);
    const bool close = match && d2 < brk * brk;
    const float dmin = wave_min(close ? d2 : 3.0e38f);
    int target = -1;
    const uint64_t mm = wave_ballot(match);
    if (dmin < 3.0e38f) target = ffs64(wave_ballot(close && d2 == dmin));      // getCacheEntry: the nearest cached point (lowest index on ties)
    else if (popc64(mm) < 4) { if (nm < MAX_CON) { target = nm; nm++; } }       // append
    else {
      // four cached (sortCachedPoints): the deepest of the five stays, the replacement leaves the largest area
      int idx[4]; uint64_t t = mm; for (int j = 0; j < 4; j++) { idx[j] = ffs64(t); t &= t - 1ull; }
      const MPoint p0 = mp_bcast(e, idx[0]), p1 = mp_bcast(e, idx[1]), p2 = mp_bcast(e, idx[2]), p3 = mp_bcast(e, idx[3]);
      int deepest = -1; float pen = q.dist;
      if (p0.dist < pen) { deepest = 0; pen = p0.dist; }
      if (p1.dist < pen) { deepest = 1; pen = p1.dist; }
      if (p2.dist < pen) { deepest = 2; pen = p2.dist; }
      if (p3.dist < pen) { deepest = 3; pen = p3.dist; }
      float res[4] = {0.f, 0.f, 0.f, 0.f};
      if (deepest != 0) { const v3 x = cross(q.la - p1.la, p3.la - p2.la); res[0] = dot(x, x); }
      if (deepest != 1) { const v3 x = cross(q.la - p0.la, p3.la - p2.la); res[1] = dot(x, x); }
      if (deepest != 2) { const v3 x = cross(q.la - p0.la, p3.la - p1.la); res[2] = dot(x, x); }
      if (deepest != 3) { const v3 x = cross(q.la - p0.la, p2.la - p1.la); res[3] = dot(x, x); }
      int bi = -1; float bv = -3.0e38f;
      for (int j = 0; j < 4; j++) if (fabsf(res[j]) > bv) { bv = fabsf(res[j]); bi = j; }
      target = idx[bi];
    }
    if (lane == target) e = q;
  }
  // (3) the contact list: face-manifold contacts in place, every pair of the substep's contacts with its cached points in cache order, then the
  // cached pairs without a new point; a cached point becomes a contact when its predicted gap is below the slack
  bool live = false; v3 wpa = mk3(0, 0, 0), wpb = wpa; int eba = 0, ebb = 0;
  if (lane < nm) {
    eba = CLI(c, e.key & 511, AGX_C_BODY); ebb = CLI(c, (e.key >> 9) & 511, AGX_C_BODY);
    m3 Ra, Rb; v3 oa, ob; body_xf(c, eba, Ra, oa); body_xf(c, ebb, Rb, ob);
    wpa = mul(Ra, e.la) + oa; wpb = mul(Rb, e.lb) + ob;
    const v3 vr = point_velocity(c, eba, wpa) - point_velocity(c, ebb, wpb);
    live = e.dist + dot(vr, e.n) * c.dt < slack;
  }
  float rec[CON_STRIDE];                                       // lane i < nnew: the i-th contact of the substep, kept while the list is rewritten
  for (int w = 0; w < CON_STRIDE; w++) rec[w] = lane < nnew ? c.gcon[CON_STRIDE * lane + w] : 0.f;
  int my_face_slot = -1, out_slot = -1, no = 0, overflow = 0;
  bool used = false;
  for (int i = 0; i <= nnew; i++) {
    int key_i = -1; bool face = false;
    if (i < nnew) { const int* ki = (const int*)(c.gcon + CON_STRIDE * i); key_i = ki[C_CA] | (ki[C_CB] << 9); face = face_box(c, ki[C_CB]); }
    if (face) { if (no < maxc) { if (lane == i) my_face_slot = no; no++; } else overflow++; continue; }
    const bool sel = lane < nm && !used && (i == nnew || e.key == key_i);
    used = used || sel;
    const bool em = sel && live;
    const uint64_t m = wave_ballot(em);
    const int slot = no + wave_rank(m), cnt = popc64(m);
    if (em && slot < maxc) out_slot = slot;
    const int room = maxc - no;
    if (cnt > room) { overflow += cnt - room; no = maxc; } else no += cnt;
  }
  wave_sync();
  if (my_face_slot >= 0) for (int w = 0; w < CON_STRIDE; w++) c.gcon[CON_STRIDE * my_face_slot + w] = rec[w];
  if (out_slot >= 0) {
    float* o = c.gcon + CON_STRIDE * out_slot; int* oi = (int*)o;
    oi[C_CA] = e.key & 511; oi[C_CB] = (e.key >> 9) & 511; oi[C_BA] = eba; oi[C_BB] = ebb;
    st3(o + C_PA, wpa); st3(o + C_PB, wpb); st3(o + C_N, e.n); o[C_DIST] = e.dist; o[C_MU] = e.mu; o[C_LAM] = 0.f;
  }
  if (lane < nm) { float* q = MP + MP_STRIDE * lane; ((int*)q)[MP_KEY] = e.key; st3(q + MP_LA, e.la); st3(q + MP_LB, e.lb); st3(q + MP_N, e.n); q[MP_DIST] = e.dist; q[MP_MU] = e.mu; }
  if (lane == 0) scr.meta[META_NMAN] = nm;
  c.ncon = no; c.overflow += overflow;
  wave_sync();
}

// build: `gaction` non-null on the first substep of an env.step() (take_step, env.py:174-222)
// returns the number of contacts dropped by a budget (contact, row or coefficient cap)
// MANIFOLD: the build kernel with the persistent-manifold stage (AGX_P_MANIFOLD).  A second kernel, not a branch: compiled into the default build
// kernel -- even as a function that is never called -- the stage costs the DEFAULT path 9.5 % (same-box A/B, session r04k: 467 -> 423 k
// env-steps/s, build kernel 8.2 -> 10.8 ms per step; scratch 176 -> 736 bytes per lane).  libagx launches it for blobs with the switch on.
template <bool MANIFOLD = false>
AGX_DEV int env_build(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gdebug, float* lds, int lane, float* gtrace = nullptr) {
  Ctx c; ctx_init(c, blob, lds, lane);
  c.timing = gdebug != nullptr; c.dbg = gdebug;
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS];
  Scratch scr = scratch_of(gscratch);
  c.E = scr.ent; c.H = scr.hdr; c.gcon = scr.con; c.gqpt = scr.qpt;
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
        // Robot.action_multiplier scales a joint's action (env.py:196-197), Robot.action_duplication hands one joint's new target to
        // several (env.py:218-220): such a joint accumulates from the angle and the limits of its source joint (AGX_R_ACT_SRC)
        const float mult = RBF(c, d, AGX_R_ACT_MULT);
        const int ds = RBI(c, d, AGX_R_ACT_SRC) > 0 ? RBI(c, d, AGX_R_ACT_SRC) - 1 : d;
        const float a32 = fminf(fmaxf(gaction[ai], -1.f), 1.f) * PRM(c, AGX_P_ACTION_SCALE) * (mult != 0.f ? mult : 1.f);
        double a = (double)a32, qa = (double)L[L_ST + c.s_q + ds]; const double lo = (double)DLO(c, ds), hi = (double)DHI(c, ds);
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
  // models with a cloth: the world frames of the moving links where this substep starts, for the cloth kernel (one-way coupling)
  if (gtrace) {
    for (int k = lane; k < 3 * c.ndof; k += 64) gtrace[12 * (k / 3) + k % 3] = L[L_LINKP + k]; for (int k = lane; k < 9 * c.ndof; k += 64) gtrace[12 * (k / 9) + 3 + k % 9] = L[L_LINKR + k];
    // ... and of the free bodies behind them (the water's cup; the garment's scene has none): a trace slot is 12 (NDOF + NFREE) words
    for (int k = lane; k < 12 * c.nfree; k += 64) { const int b = k / 12, t = k % 12; gtrace[12 * (c.ndof + b) + t] = t < 3 ? L[L_ST + c.s_free + 13 * b + t] : L[L_FREER + 9 * b + t - 3]; }
  }
  aba_and_minv(c); AGX_TICK(1)
  predict_velocities(c); AGX_TICK(2)
  collide(c); AGX_TICK(3)
  if constexpr (MANIFOLD) { if (PRM(c, AGX_P_MANIFOLD) > 0.f) manifold_update(c, scr); }
  warm_seed(c, scr, lane);
  build_rows(c); AGX_TICK(4)
#undef AGX_TICK
  // hand-over to the solve kernel
  for (int k = lane; k < SCR_VEL; k += 64) scr.vel[k] = L[L_VEL + k];
  if (lane == 0) { scr.meta[META_NCON] = c.ncon; scr.meta[META_NROWS] = c.nrows; scr.meta[META_NNC] = c.first_normal; scr.meta[META_NEAR] = c.near_mask; scr.meta[META_OVERFLOW] = c.overflow; scr.meta[META_NENT] = c.nent; scr.meta[META_NQPT] = c.nqpt; }
  if (gdebug) {   // first-substep internals for the parity tests and the phase cycle counters
    if (lane == 0) { gdebug[0] = (float)c.ncon; gdebug[1] = (float)c.nrows; gdebug[2] = (float)c.overflow; gdebug[3] = (float)c.first_normal; gdebug[4] = (float)c.nent; }   // qdd at DBG_QDD
    for (int q = lane; q < MAX_CON * CON_STRIDE; q += 64) gdebug[DBG_CON + q] = scr.con[q];
    for (int q = lane; q < MAX_DOF * MAX_DOF; q += 64) gdebug[DBG_MINV + q] = L[L_MINV + q];
    wave_sync();
    for (int q = lane; q < SCR_HDR; q += 64) gdebug[DBG_HDR + q] = scr.hdr[q];
    if (lane == 0) { for (int k = 0; k < 16; k++) if (k != 5 && k != 6 && k != 7) gdebug[DBG_TIME + k] = (float)c.tm[k]; }
  }
  return c.overflow;
}

// Collision flags of a state whose contacts the build kernel has just written to the per-env scratch record (agx_check_collisions):
//   AGX_COLLIDE_ENV  -- a robot link or the tool touches (distance <= 0) the human or the furniture: what init_robot_pose and
//                       ik_random_restarts reject (env.py:299-308, robot.py:103-108: get_closest_points(obj, distance=0));
//   AGX_COLLIDE_SELF -- two robot links, or a robot link and the tool, interpenetrate by more than 1 cm (not checked by the reference,
//                       whose null-space IK keeps away from folded arms; the host-side least-squares IK needs it).
// Wave-uniform result.
AGX_DEV int collision_flags(const uint32_t* __restrict__ blob, const float* __restrict__ gscratch, int lane) {
  const int* bi = (const int*)blob;
  const int* meta = (const int*)(gscratch + SCR_O_META);
  const int ncon = meta[META_NCON];
  bool env = false, self = false;
  if (lane < ncon) {
    const float* k = gscratch + SCR_O_CON + CON_STRIDE * lane; const int* ki = (const int*)k;
    const int ta = bi[bi[AGX_H_OFF_COLL] + ki[C_CA] * AGX_C_STRIDE + AGX_C_TAG], tb = bi[bi[AGX_H_OFF_COLL] + ki[C_CB] * AGX_C_STRIDE + AGX_C_TAG];
    const bool ra = ta == AGX_TAG_ROBOT || ta == AGX_TAG_TOOL, rb = tb == AGX_TAG_ROBOT || tb == AGX_TAG_TOOL;
    const bool oa = ta == AGX_TAG_HUMAN || ta == AGX_TAG_TABLE || ta == AGX_TAG_WHEELCHAIR || ta == AGX_TAG_BED;
    const bool ob = tb == AGX_TAG_HUMAN || tb == AGX_TAG_TABLE || tb == AGX_TAG_WHEELCHAIR || tb == AGX_TAG_BED;
    env = ((ra && ob) || (rb && oa)) && k[C_DIST] <= 0.f;
    self = ra && rb && k[C_DIST] < -0.01f;
  }
  return (wave_any(env) ? AGX_COLLIDE_ENV : 0) | (wave_any(self) ? AGX_COLLIDE_SELF : 0);
}

// the part of the solve kernels after the Gauss-Seidel sweeps: state copy, hooks decision, integration, arm limits, store
AGX_DEV void solve_tail(Ctx& c, float* gstate, Scratch& scr, int sw, int phase, float dv0, float dv1) {
  float* lds = c.lds; const int lane = c.lane;
  for (int k = lane; k < sw; k += 64) lds[L_ST + k] = gstate[k];
  wave_sync();
  c.gender = c.ldsi[L_ST + c.s_env + AGX_E_GENDER]; c.frozen = c.ldsi[L_ST + c.s_env + AGX_E_FROZEN];
  { const float ls = c.lds[L_ST + c.s_env + AGX_E_LIMIT_SCALE]; c.limit_scale = ls > 0.f ? ls : 1.f; }   // records written before v6 carry 0
  c.coop = TKI(c, AGX_T_COOP) == 1;
  // hooks (env.py:227-231) run after the last internal substep of a p.stepSimulation() call of take_step -- not in the reset-time settle
  // loops, which are plain engine steps (AGX_PHASE_SETTLE), and only while the human is in env.agents: controllable or tremor
  // (env.py:130-131).  A human arm that is dynamic because of a reactive hold only (scratch itch, arm manipulation, dressing;
  // human.py:108,124-127) is never limit-reset by the reference.
  { const int S = c.bi[AGX_H_SIM_SUBSTEPS], ph = phase & ~AGX_PHASE_SETTLE;
    const bool tremor_on = wave_any(lane < c.nhdof && lds[L_ST + c.s_tremor + lane] != 0.f);
    c.hooks = (S <= 1 || (ph + 1) % S == 0) && !(phase & AGX_PHASE_SETTLE) && (c.coop || tremor_on); }
  integrate(c, scr.vel, dv0, dv1);
  if constexpr (TASK != AGX_TASK_FEEDING) arm_limits(c, lds + L_VEL);   // the velocity vector is dead after the integration
  store_env(c, gstate, sw);
}
// solve: PGS + integration + post-substep hooks of one p.stepSimulation() (env.py:226-232)
// lds_words: the solve kernel's LDS (LDS_SOLVE_WORDS, or more / less when libagx was told so: the row-local sweep sizes its window from it)
AGX_DEV void env_solve(const uint32_t* blob, float* gstate, float* gscratch, float* gdebug, float* lds, int lane, int phase = 0, int lds_words = LDS_SOLVE_WORDS) {
  Ctx c; ctx_init(c, blob, lds, lane);
  const int sw = c.bi[AGX_H_STATE_WORDS];
  Scratch scr = scratch_of(gscratch);
  c.E = scr.ent; c.H = scr.hdr; c.gcon = scr.con; c.dbg = gdebug;
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC]; c.nent = scr.meta[META_NENT];
  const long long t0 = gdebug ? wave_clock() : 0;
  float dv0 = 0.f, dv1 = 0.f;
  bool solved = false;
  if constexpr (LVW_COMPILED) { if (lvw_eligible(c, lds_words)) solved = pgs_lvw(c, lds, lds_words, dv0, dv1); }              // the wide row-local sweep: up to four rows per visit (agx_pgs_lvw.h)
  if constexpr (LVS_COMPILED) { if (!solved && lvs_eligible(c, lds_words)) { pgs_lvs(c, lds, lds_words, dv0, dv1); solved = true; } }   // the row-local sweep, pairs and velocities in LDS (agx_pgs_lvs.h)
  else if constexpr (LV_COMPILED) { if (lv_eligible(c)) { pgs_lv(c, lds, lds_words, dv0, dv1); solved = true; } }      // the row-local sweep, headers in LDS too (agx_pgs_lv.h)
  if (!solved) {
    // environments with few rows take the row-space sweep (pgs_rowspace; it needs all pairs inside the window)
    const bool rowspace = RS_MAX_ROWS > 0 && c.nrows <= RS_MAX_ROWS && c.nv <= 64 && c.nv <= RS_NVP && c.nrows > 0 && c.nent <= SOLVE_LDS_PAIRS;
    { const int np = c.nent < SOLVE_LDS_PAIRS ? c.nent : SOLVE_LDS_PAIRS;
      const f2* src = (const f2*)scr.ent; f2* dst = (f2*)(lds + L_SOLVE_ENT);
      for (int k = lane; k < np; k += 64) dst[k] = src[k]; }
    wave_sync();
    if (!(rowspace && pgs_rowspace(c, lds + L_SOLVE_ENT, dv0, dv1))) pgs(c, dv0, dv1);
  }
  warm_remember(c, scr, lane);
  const long long t1 = gdebug ? wave_clock() : 0;
  solve_tail(c, gstate, scr, sw, phase, dv0, dv1);
  if (gdebug && lane == 0) { gdebug[DBG_TIME + 5] = (float)(t1 - t0); gdebug[DBG_TIME + 6] = (float)(wave_clock() - t1); }
}

AGX_DEV void observe_dressing(const Ctx& c, float cloth_force_sum, float robot_force, float* gobs);
AGX_DEV void env_observe(const uint32_t* blob, float* gstate, float* gobs, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  load_env(c, gstate, c.bi[AGX_H_STATE_WORDS]);
  kinematics(c); update_target(c);
  if constexpr (TASK == AGX_TASK_BED_BATHING) observe_bed(c, 0.f, 0.f, 0.f, gobs);
  else if constexpr (TASK == AGX_TASK_SCRATCH_ITCH) observe_scratch(c, 0.f, 0.f, 0.f, gobs);
  else if constexpr (TASK == AGX_TASK_DRESSING) observe_dressing(c, c.lds[L_ST + c.bi[AGX_H_S_TASK] + AGX_DR_FORCE_SUM], 0.f, gobs);
  else if constexpr (TASK == AGX_TASK_ARM_MANIPULATION) observe_arm(c, 0.f, 0.f, 0.f, 0.f, 0.f, gobs);
  else observe(c, 0.f, 0.f, gobs);
}

// end-effector speed: norm of getLinkState(ee, computeLinkVelocity)[6] (feeding.py:22, bed_bathing.py:19)
AGX_DEV float frame_speed(const Ctx& c, int link, v3 p) {
  const float* L = c.lds;
  float sv[6] = {0, 0, 0, 0, 0, 0};
  for (int d = link; d >= 0; d = RBI(c, d, AGX_R_PARENT)) { float qd = L[L_ST + c.s_qd + d]; for (int j = 0; j < 6; j++) sv[j] += L[L_S + 6 * d + j] * qd; }
  v3 xr = p - ld3(L + L_MISC + M_REF);
  v3 v = mk3(sv[3], sv[4], sv[5]) + cross(mk3(sv[0], sv[1], sv[2]), xr);
  return sqrtf(dot(v, v));
}
AGX_DEV float ee_speed_of(const Ctx& c) { return frame_speed(c, TKI(c, AGX_T_EE_LINK), ld3(c.lds + L_MISC + M_EEP)); }

// finish, bed bathing: everything BedBathingEnv.step does after take_step (bed_bathing.py:15-39)
AGX_DEV void env_finish_bed(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                            float* ginfo, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS], act_dim = c.bi[AGX_H_ACT_DIM];
  Scratch scr = scratch_of(gscratch);
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC]; c.nqpt = scr.meta[META_NQPT];
  load_env(c, gstate, sw);
  float an2 = 0.f;
  for (int k = 0; k < act_dim; k++) an2 += gaction[k] * gaction[k];
  wave_sync();
  kinematics(c);   // poses as the getters of _get_obs see them after the last stepSimulation
  // get_total_force (bed_bathing.py:41-78) from the last substep's contact impulses:
  //   total_force_on_human = robot-human + tool-human, tool_force = every contact of the tool,
  //   tool_force_on_human = (tool link 1, human)
  float rf = 0.f, tf = 0.f, thf = 0.f, pf = 0.f;
  if (lane < c.ncon) {
    const float* k = scr.con + CON_STRIDE * lane; const int* ki = (const int*)k;
    const int ca = ki[C_CA], cb = ki[C_CB], ta = CLI(c, ca, AGX_C_TAG), tb = CLI(c, cb, AGX_C_TAG);
    const float f = k[C_LAM] / c.dt;
    const bool human = ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN, tool = ta == AGX_TAG_TOOL || tb == AGX_TAG_TOOL, robot = ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT;
    if (tool) tf = f;
    if (human && robot) rf = f;
    if (human && tool) { thf = f; const int tc = ta == AGX_TAG_TOOL ? ca : cb; if (TKI(c, AGX_T_PAD_LINK) >> (CLI(c, tc, AGX_C_LINK) + 1) & 1) pf = f; }
  }
  const float robot_f = wave_sum(rf), tool_f = wave_sum(tf), pad_f = wave_sum(pf), total_f = robot_f + wave_sum(thf);
  observe_bed(c, tool_f, total_f, pad_f, gobs);
  // targets wiped by the manifold points of the pad on the human's links (not its base, bed_bathing.py:52-53): lane = target
  int success = Li[L_ST + c.s_env + AGX_E_TASK_SUCCESS];
  const int s_task = c.bi[AGX_H_S_TASK];
  int new_points = 0;
  {
    const int g = c.gender, nt = TKI(c, AGX_T_NT + 2 * g) + TKI(c, AGX_T_NT + 2 * g + 1);
    const float* TT = c.bf + c.bi[AGX_H_OFF_TARGETS] + 4 * g * TKI(c, AGX_T_NT_MAX);
    const float r2 = TKF(c, AGX_T_TARGET_RADIUS) * TKF(c, AGX_T_TARGET_RADIUS);
    for (int base = 0; base < nt; base += 64) {
      const int t = base + lane; bool hit = false;
      const uint32_t w0 = (uint32_t)Li[L_ST + s_task + AGX_BB_ALIVE + (base >> 5)], w1 = (uint32_t)Li[L_ST + s_task + AGX_BB_ALIVE + (base >> 5) + 1];
      const uint64_t alive = (uint64_t)w0 | ((uint64_t)w1 << 32);
      if (t < nt && (alive >> lane & 1)) {
        const int link = TKI(c, AGX_T_ARM_LINK + ((const int*)TT)[4 * t + 3]);
        const v3 w = mul(ldm3(L + L_LINKR + 9 * link), ld3(TT + 4 * t)) + ld3(L + L_LINKP + 3 * link);   // update_targets (bed_bathing.py:190-203)
        for (int q = 0; q < c.nqpt; q++) {
          const float* o = scr.qpt + QPT_STRIDE * q; const int lb = ((const int*)o)[3];
          if (lb < 0) continue;
          const v3 d = ld3(o) - w;
          if (dot(d, d) < r2) hit = true;
        }
      }
      const uint64_t hm = wave_ballot(hit);
      new_points += popc64(hm);
      wave_sync();
      if (lane == 0) { const uint64_t na = alive & ~hm; Li[L_ST + s_task + AGX_BB_ALIVE + (base >> 5)] = (int)(uint32_t)na; Li[L_ST + s_task + AGX_BB_ALIVE + (base >> 5) + 1] = (int)(uint32_t)(na >> 32); }
      wave_sync();
    }
  }
  success += new_points;
  // reward_distance = -min distance tool <-> human within CLOSEST_DIST (bed_bathing.py:23): lane = (tool collider, human collider)
  float dmin = TKF(c, AGX_T_CLOSEST_DIST);
  {
    int t0 = -1, t1 = -1, h0 = -1, h1 = -1;
    for (int g = 0; g < c.ngroup; g++) {   // the (tool, human) group carries both collider ranges
      const int a0 = GRI(c, g, AGX_G_A0), b0 = GRI(c, g, AGX_G_B0);
      if (CLI(c, a0, AGX_C_TAG) == AGX_TAG_TOOL && CLI(c, b0, AGX_C_TAG) == AGX_TAG_HUMAN) {
        t0 = a0; t1 = GRI(c, g, AGX_G_A1); h0 = b0; h1 = GRI(c, g, AGX_G_B1);
        if (c.gender == 1 && GRI(c, g, AGX_G_B0F) >= 0) { h0 = GRI(c, g, AGX_G_B0F); h1 = GRI(c, g, AGX_G_B1F); }
        break;
      }
    }
    float* AB = L + L_ARENA;
    for (int col = t0 + lane; col < t1; col += 64) {   // world AABBs of the tool colliders (narrowphase centres its arithmetic there)
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
    const int nh = h1 - h0, np = (t1 - t0) * nh;
    const float lim = dmin;
    for (int base = 0; base < np; base += 64) {
      const int p = base + lane; const bool has = p < np;
      const int ti = has ? p / nh : 0, hi = has ? p - ti * nh : 0;
      Cand k; k.dist = lim; k.n = mk3(0.f, 0.f, 0.f); k.pa = k.n; k.pb = k.n; k.gap = 0.f;
      const bool hit = narrowphase(c, t0 + ti, h0 + hi, lim, k, has);
      dmin = fminf(dmin, wave_min(hit ? k.dist : lim));
    }
  }
  // human_preferences (env.py:237-274), non-feeding branch: forces away from the target area, high forces at the target
  const float ee_speed = ee_speed_of(c);
  const float pref = TKF(c, AGX_T_C_V) * (-ee_speed) + TKF(c, AGX_T_C_F) * (-(total_f - pad_f)) + TKF(c, AGX_T_C_HF) * (pad_f < 10.f ? 0.f : -pad_f);
  const float reward = TKF(c, AGX_T_W_DISTANCE) * (-dmin) + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + TKF(c, AGX_T_W_WIPE) * (float)new_points + pref;
  const int iteration = Li[L_ST + c.s_env + AGX_E_ITERATION];
  wave_sync();
  if (lane == 0) {
    Li[L_ST + c.s_env + AGX_E_TASK_SUCCESS] = success;
    *greward = reward;
    *gdone = (uint8_t)(iteration >= (int)TKF(c, AGX_T_EPISODE_LEN));
    if (ginfo) {
      ginfo[AGX_INFO_TOTAL_FORCE] = total_f;
      ginfo[AGX_INFO_TASK_SUCCESS] = (float)(success >= Li[L_ST + c.s_env + AGX_E_TOTAL_FOOD] * TKF(c, AGX_T_SUCCESS_FRAC));
      ginfo[AGX_INFO_ROBOT_FORCE] = robot_f; ginfo[AGX_INFO_TOOL_FORCE] = pad_f; ginfo[AGX_INFO_FOOD_REWARD] = (float)new_points;
      ginfo[AGX_INFO_PREF] = pref; ginfo[AGX_INFO_NCONTACT] = (float)c.ncon; ginfo[AGX_INFO_NROWS] = (float)c.nrows;
    }
  }
  store_env(c, gstate, sw);
}

// finish, arm manipulation: everything ArmManipulationEnv.step does after take_step (arm_manipulation.py:18-60), single-arm robot
AGX_DEV void env_finish_arm(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                            float* ginfo, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS], act_dim = c.bi[AGX_H_ACT_DIM];
  Scratch scr = scratch_of(gscratch);
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC]; c.nqpt = 0;
  load_env(c, gstate, sw);
  float an2 = 0.f;
  for (int k = 0; k < act_dim; k++) an2 += gaction[k] * gaction[k];
  wave_sync();
  kinematics(c);
  // get_total_force (:62-69): the one tool of a single-arm robot is counted as tool_right and as tool_left
  const int tb2 = TKI(c, AGX_T_TOOL2_BODY); const bool dual = tb2 > 0; const int code2 = AGX_BODY_FREE0 + tb2;
  float rf = 0.f, tf0 = 0.f, tf1 = 0.f, thf0 = 0.f, thf1 = 0.f;
  if (lane < c.ncon) {
    const float* k = scr.con + CON_STRIDE * lane; const int* ki = (const int*)k;
    const int ta = CLI(c, ki[C_CA], AGX_C_TAG), tb = CLI(c, ki[C_CB], AGX_C_TAG);
    const float f = k[C_LAM] / c.dt;
    const bool human = ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN, robot = ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT;
    if (human && robot) rf = f;
    for (int side = 0; side < 2; side++) {              // a contact between the two tools counts for both
      const int tag = side ? tb : ta, body = side ? ki[C_BB] : ki[C_BA];
      if (tag != AGX_TAG_TOOL) continue;
      if (dual && body == code2) { tf1 += f; if (human) thf1 += f; } else { tf0 += f; if (human) thf0 += f; }
    }
  }
  const float robot_f = wave_sum(rf); float tf_r = wave_sum(tf0), tf_l = wave_sum(tf1), thf_r = wave_sum(thf0), thf_l = wave_sum(thf1);
  if (!dual) { tf_l = tf_r; thf_l = thf_r; }
  const float total_f = robot_f + thf_r + thf_l;
  observe_arm(c, tf_r, tf_l, total_f, thf_r, thf_l, gobs);
  // tool.get_closest_points(human, distance=0.01) (env.py:264-265): one point per (hull of the tool, shape of the human) pair that close,
  // at the poses after the last substep: lane = pair
  int near_r = 0, near_l = 0;
  {
    int t0 = -1, t1 = -1, h0 = -1, h1 = -1;
    for (int g = 0; g < c.ngroup; g++) {
      const int a0 = GRI(c, g, AGX_G_A0), b0 = GRI(c, g, AGX_G_B0);
      if (CLI(c, a0, AGX_C_TAG) == AGX_TAG_TOOL && CLI(c, b0, AGX_C_TAG) == AGX_TAG_HUMAN) {
        t0 = a0; t1 = GRI(c, g, AGX_G_A1); h0 = b0; h1 = GRI(c, g, AGX_G_B1);
        if (c.gender == 1 && GRI(c, g, AGX_G_B0F) >= 0) { h0 = GRI(c, g, AGX_G_B0F); h1 = GRI(c, g, AGX_G_B1F); }
        break;
      }
    }
    float* AB = L + L_ARENA;
    for (int col = t0 + lane; col < t1; col += 64) {   // world AABBs of the tool colliders (narrowphase centres its arithmetic there)
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
    const int nh = h1 - h0, np = (t1 - t0) * nh;
    const float lim = TKF(c, AGX_T_PRESSURE_DIST);
    for (int base = 0; base < np; base += 64) {
      const int p = base + lane; const bool has = p < np;
      const int ti = has ? p / nh : 0, hi = has ? p - ti * nh : 0;
      Cand k; k.dist = lim; k.n = mk3(0.f, 0.f, 0.f); k.pa = k.n; k.pb = k.n; k.gap = 0.f;
      const bool hit = narrowphase(c, t0 + ti, h0 + hi, lim, k, has) && k.dist < lim;
      const bool second = dual && CLI(c, t0 + ti, AGX_C_BODY) == code2;
      near_r += popc64(wave_ballot(hit && !second)); near_l += popc64(wave_ballot(hit && second));
    }
    if (!dual) near_l = near_r;
  }
  // right + left end effector (:26-27): the same link on a single-arm robot
  float ee_speed = ee_speed_of(c);
  if (dual) {
    const int l2 = TKI(c, AGX_T_EE2_LINK);
    const v3 e2 = mul(ldm3(L + L_LINKR + 9 * l2), mk3(TKF(c, AGX_T_EE2_POS), TKF(c, AGX_T_EE2_POS + 1), TKF(c, AGX_T_EE2_POS + 2))) + ld3(L + L_LINKP + 3 * l2);
    ee_speed += frame_speed(c, l2, e2);
  } else ee_speed *= 2.f;
  const float pressure = (near_r <= 0 ? 0.f : thf_r / (float)near_r) + (near_l <= 0 ? 0.f : thf_l / (float)near_l);     // env.py:266-269
  // human_preferences (env.py:237-274): reward_force_nontarget = -(total - (right + left)), tool_force_at_target = 0
  const float pref = TKF(c, AGX_T_C_V) * (-ee_speed) + TKF(c, AGX_T_C_F) * (-(total_f - thf_r - thf_l)) + TKF(c, AGX_T_C_P) * (-pressure);
  v3 spr; m3 sRr; tool_base_pose(c, spr, sRr);
  v3 spl = spr; if (dual) { m3 sRl; tool_base_pose_of(c, tb2, spl, sRl); }
  const v3 elbow = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + 1)), wrist = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + 2));
  const v3 stomach = ld3(L + L_HUMAN + 12 * TKI(c, AGX_T_STOMACH_BODY)), waist = ld3(L + L_HUMAN + 12 * TKI(c, AGX_T_WAIST_BODY));
  const v3 d0 = spl - elbow, d3 = spr - wrist, d1 = elbow - stomach, d2 = wrist - waist;
  const float rd_left = -sqrtf(dot(d0, d0)), rd_right = -sqrtf(dot(d3, d3)), rd_human = -sqrtf(dot(d1, d1)) - sqrtf(dot(d2, d2));      // :36-38
  const float we = TKF(c, AGX_T_W_WIPE);                                                                   // distance_end_effector_weight
  const float reward = TKF(c, AGX_T_W_DISTANCE) * rd_human + (dual ? we * rd_left + we * rd_right : 2.f * we * rd_left) + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + pref;   // :41-44
  const float th_f = dual ? thf_r + thf_l : thf_r; const int near_pts = dual ? near_r + near_l : near_r;
  const int s_task = c.bi[AGX_H_S_TASK];
  const float best0 = L[L_ST + s_task + AGX_AM_BEST], best = (best0 == 0.f || rd_human > best0) ? rd_human : best0;   // :47-48
  const int iteration = Li[L_ST + c.s_env + AGX_E_ITERATION];
  wave_sync();
  if (lane == 0) {
    L[L_ST + s_task + AGX_AM_BEST] = best;
    *greward = reward;
    *gdone = (uint8_t)(iteration >= (int)TKF(c, AGX_T_EPISODE_LEN));
    if (ginfo) {
      ginfo[AGX_INFO_TOTAL_FORCE] = total_f;
      ginfo[AGX_INFO_TASK_SUCCESS] = (float)(best >= TKF(c, AGX_T_SUCCESS_FRAC));
      ginfo[AGX_INFO_ROBOT_FORCE] = robot_f; ginfo[AGX_INFO_TOOL_FORCE] = th_f; ginfo[AGX_INFO_FOOD_REWARD] = (float)near_pts;
      ginfo[AGX_INFO_PREF] = pref; ginfo[AGX_INFO_NCONTACT] = (float)c.ncon; ginfo[AGX_INFO_NROWS] = (float)c.nrows;
    }
  }
  store_env(c, gstate, sw);
}

// finish, scratch itch: everything ScratchItchEnv.step does after take_step (scratch_itch.py:14-44)
AGX_DEV void env_finish_scratch(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                                float* ginfo, float* lds, int lane) {
  Ctx c; ctx_init(c, blob, lds, lane);
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS], act_dim = c.bi[AGX_H_ACT_DIM];
  Scratch scr = scratch_of(gscratch);
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC]; c.nqpt = scr.meta[META_NQPT];
  load_env(c, gstate, sw);
  float an2 = 0.f;
  for (int k = 0; k < act_dim; k++) an2 += gaction[k] * gaction[k];
  wave_sync();
  kinematics(c);   // poses as the getters of _get_obs see them after the last stepSimulation
  const v3 target = scratch_target(c);                            // update_targets after the last substep
  const float r2 = TKF(c, AGX_T_TARGET_RADIUS) * TKF(c, AGX_T_TARGET_RADIUS);
  // get_total_force (scratch_itch.py:46-57) from the last substep's contact impulses
  float rf = 0.f, tf = 0.f, thf = 0.f, gf = 0.f;
  if (lane < c.ncon) {
    const float* k = scr.con + CON_STRIDE * lane; const int* ki = (const int*)k;
    const int ca = ki[C_CA], cb = ki[C_CB], ta = CLI(c, ca, AGX_C_TAG), tb = CLI(c, cb, AGX_C_TAG);
    const float f = k[C_LAM] / c.dt;
    const bool human = ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN, tool = ta == AGX_TAG_TOOL || tb == AGX_TAG_TOOL, robot = ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT;
    if (tool) tf = f;
    if (human && robot) rf = f;
    if (human && tool) {
      thf = f;
      const bool tool_is_a = ta == AGX_TAG_TOOL; const int tc = tool_is_a ? ca : cb;
      const v3 on_human = ld3(k + (tool_is_a ? C_PB : C_PA)), d = on_human - target;
      if ((TKI(c, AGX_T_PAD_LINK) >> (CLI(c, tc, AGX_C_LINK) + 1) & 1) && dot(d, d) < r2) gf = f;      // close to the target (:54)
    }
  }
  const float robot_f = wave_sum(rf), tool_f = wave_sum(tf), target_f = wave_sum(gf), total_f = robot_f + wave_sum(thf);
  observe_scratch(c, tool_f, total_f, target_f, gobs);
  // target_contact_pos: the LAST manifold point of the tool's links 0 / 1 on the human within TARGET_RADIUS of the target (:52-56)
  bool near = false; v3 qp = mk3(0.f, 0.f, 0.f);
  if (lane < c.nqpt) { qp = ld3(scr.qpt + QPT_STRIDE * lane); const v3 d = qp - target; near = dot(d, d) < r2; }
  const uint64_t nm = wave_ballot(near);
  const int s_task = c.bi[AGX_H_S_TASK];
  int success = Li[L_ST + c.s_env + AGX_E_TASK_SUCCESS];
  float scratch_reward = 0.f;
  if (nm) {
    const int last = 63 - __builtin_clzll((unsigned long long)nm);
    const v3 cp = mk3(wave_bcast(qp.x, last), wave_bcast(qp.y, last), wave_bcast(qp.z, last));
    const v3 dp = cp - ld3(L + L_ST + s_task + AGX_SI_PREV_CONTACT);
    if (sqrtf(dot(dp, dp)) > 0.01f && target_f < 10.f) {         // the tool moved along the skin and does not press too hard (:28-32)
      scratch_reward = 5.f; success += 1;
      wave_sync();
      if (lane == 0) st3(L + L_ST + s_task + AGX_SI_PREV_CONTACT, cp);
      wave_sync();
    }
  }
  v3 sp; m3 sR; tool_base_pose(c, sp, sR);
  const v3 dd = target - sp;
  const float ee_speed = ee_speed_of(c);
  const float pref = TKF(c, AGX_T_C_V) * (-ee_speed) + TKF(c, AGX_T_C_F) * (-(total_f - target_f)) + TKF(c, AGX_T_C_HF) * (target_f < 10.f ? 0.f : -target_f);
  const float reward = TKF(c, AGX_T_W_DISTANCE) * (-sqrtf(dot(dd, dd))) + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + TKF(c, AGX_T_W_WIPE) * scratch_reward + pref;
  const int iteration = Li[L_ST + c.s_env + AGX_E_ITERATION];
  wave_sync();
  if (lane == 0) {
    Li[L_ST + c.s_env + AGX_E_TASK_SUCCESS] = success;
    *greward = reward;
    *gdone = (uint8_t)(iteration >= (int)TKF(c, AGX_T_EPISODE_LEN));
    if (ginfo) {
      ginfo[AGX_INFO_TOTAL_FORCE] = total_f;
      ginfo[AGX_INFO_TASK_SUCCESS] = (float)(success >= Li[L_ST + c.s_env + AGX_E_TOTAL_FOOD] * TKF(c, AGX_T_SUCCESS_FRAC));
      ginfo[AGX_INFO_ROBOT_FORCE] = robot_f; ginfo[AGX_INFO_TOOL_FORCE] = target_f; ginfo[AGX_INFO_FOOD_REWARD] = scratch_reward;
      ginfo[AGX_INFO_PREF] = pref; ginfo[AGX_INFO_NCONTACT] = (float)c.ncon; ginfo[AGX_INFO_NROWS] = (float)c.nrows;
    }
  }
  store_env(c, gstate, sw);
}


// DressingEnv._get_obs (dressing.py:78-110); every lane computes, lane 0 writes
AGX_DEV void observe_dressing(const Ctx& c, float cloth_force_sum, float robot_force, float* gobs) {
  const float* L = c.lds;
  v3 bp; m3 BR; robot_base_pose(c, bp, BR);
  const v3 ep = ld3(L + L_MISC + M_EEP); const m3 eR = ldm3(L + L_MISC + M_EER);
  const v3 epr = tmul(BR, ep - bp); const q4 eq = m3_to_quat(mul_at(BR, eR));
  v3 jp[3], jpr[3];
  for (int k = 0; k < 3; k++) { jp[k] = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + k)); jpr[k] = tmul(BR, jp[k] - bp); }
  if (c.lane == 0) {
    int o = 0;
    gobs[o++] = epr.x; gobs[o++] = epr.y; gobs[o++] = epr.z;
    gobs[o++] = eq.x; gobs[o++] = eq.y; gobs[o++] = eq.z; gobs[o++] = eq.w;
    for (int d = 0; d < c.nrobot; d++) if (RBI(c, d, AGX_R_ACT) >= 0 && !RBI(c, d, AGX_R_OBS_SKIP)) {
      float a = L[L_ST + c.s_q + d] + 3.14159265358979f;
      gobs[o++] = (a - 6.28318530717959f * floorf(a / 6.28318530717959f)) - 3.14159265358979f;
    }
    for (int k = 0; k < 3; k++) { gobs[o++] = jpr[k].x; gobs[o++] = jpr[k].y; gobs[o++] = jpr[k].z; }
    gobs[o++] = cloth_force_sum;
    if (c.coop) {   // human_obs (dressing.py:100-106), in the frame of the human's base
      const v3 hb = ld3(L + L_HUMAN); const m3 HR = ldm3(L + L_HUMAN + 3);
      const v3 eph = tmul(HR, ep - hb); const q4 eqh = m3_to_quat(mul_at(HR, eR));
      gobs[o++] = eph.x; gobs[o++] = eph.y; gobs[o++] = eph.z;
      gobs[o++] = eqh.x; gobs[o++] = eqh.y; gobs[o++] = eqh.z; gobs[o++] = eqh.w;
      for (int d = c.nrobot; d < c.ndof; d++) if (RBI(c, d, AGX_R_ACT) >= 0) gobs[o++] = L[L_ST + c.s_q + d];
      for (int k = 0; k < 3; k++) { const v3 h = tmul(HR, jp[k] - hb); gobs[o++] = h.x; gobs[o++] = h.y; gobs[o++] = h.z; }
      gobs[o++] = cloth_force_sum; gobs[o++] = robot_force;
    }
  }
}
AGX_DEV float signed_volume6(v3 a, v3 b, v3 c, v3 d) { return dot(cross(b - a, c - a), d - a); }   // 6 x the signed volume: only its sign is used
AGX_DEV int sgnf(float x) { return (x > 0.f) - (x < 0.f); }
// Util.line_intersects_triangle (util.py:125-132)
AGX_DEV bool line_intersects_triangle(v3 p0, v3 p1, v3 p2, v3 q0, v3 q1) {
  if (sgnf(signed_volume6(q0, p0, p1, p2)) == sgnf(signed_volume6(q1, p0, p1, p2))) return false;
  const int a = sgnf(signed_volume6(q0, q1, p0, p1)), b = sgnf(signed_volume6(q0, q1, p1, p2)), cc = sgnf(signed_volume6(q0, q1, p2, p0));
  return a == b && b == cc;
}
// do the six sleeve vertices straddle both planes through `origin` that contain the arm axis (util.py:144-171)?
AGX_DEV bool points_around_axis(const v3* pts, v3 from, v3 to, v3 origin) {
  v3 nrm = to - from; nrm = (1.0f / sqrtf(dot(nrm, nrm))) * nrm;
  v3 tan = cross(mk3(1.f, 1.f, 0.f), nrm); tan = (1.0f / sqrtf(dot(tan, tan))) * tan;
  v3 bin = cross(tan, nrm); bin = (1.0f / sqrtf(dot(bin, bin))) * bin;
  bool tp = false, tn = false, bp = false, bn = false;
  for (int i = 0; i < 6; i++) { const v3 d = pts[i] - origin; const float t = dot(tan, d), b = dot(bin, d); tp |= t > 0.f; tn |= t < 0.f; bp |= b > 0.f; bn |= b < 0.f; }
  return tp && tn && bp && bn;
}
// finish, dressing: everything DressingEnv.step does after take_step (dressing.py:20-76).  greport: what the cloth kernel left for
// this environment -- the six sleeve vertices and, per node and contact slot, {height, |force|} of the node-vs-rigid contacts of the last substep
AGX_DEV void env_finish_dressing(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                                 float* ginfo, float* lds, int lane, const float* greport) {
  Ctx c; ctx_init(c, blob, lds, lane);
  float* L = c.lds; int* Li = c.ldsi;
  const int sw = c.bi[AGX_H_STATE_WORDS], act_dim = c.bi[AGX_H_ACT_DIM];
  Scratch scr = scratch_of(gscratch);
  c.ncon = scr.meta[META_NCON]; c.nrows = scr.meta[META_NROWS]; c.first_normal = scr.meta[META_NNC];
  load_env(c, gstate, sw);
  float an2 = 0.f;
  for (int k = 0; k < act_dim; k++) an2 += gaction[k] * gaction[k];
  wave_sync();
  kinematics(c);
  const int* cl = c.bi + c.bi[AGX_H_OFF_CLOTH]; const float* clp = c.bf + c.bi[AGX_H_OFF_CLOTH] + cl[AGX_CL_OFF_PARAM];
  const v3 shoulder = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK)), elbow = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + 1)), wrist = ld3(L + L_LINKP + 3 * TKI(c, AGX_T_OBS_LINK + 2));
  v3 pts[6];
  for (int k = 0; k < 6; k++) pts[k] = greport ? ld3(greport + 3 * k) : mk3(0.f, 0.f, 0.f);
  // Util.sleeve_on_arm_reward (util.py:134-202)
  const float rad = TKF(c, AGX_T_ARM_RADIUS + c.gender);
  const v3 we = wrist - elbow, es = shoulder - elbow; const float lwe = sqrtf(dot(we, we)), les = sqrtf(dot(es, es));
  const v3 hand_end = wrist + (rad * 2.f / lwe) * we, elbow_end = elbow - (rad / lwe) * we, shoulder_end = shoulder + (rad / les) * es;
  const bool around_fore = points_around_axis(pts, elbow_end, hand_end, hand_end), around_upper = points_around_axis(pts, shoulder_end, elbow_end, shoulder_end);
  const bool f1 = line_intersects_triangle(pts[0], pts[1], pts[2], hand_end, elbow_end), f2 = line_intersects_triangle(pts[3], pts[4], pts[5], hand_end, elbow_end);
  const bool u1 = line_intersects_triangle(pts[0], pts[1], pts[2], elbow_end, shoulder_end), u2 = line_intersects_triangle(pts[3], pts[4], pts[5], elbow_end, shoulder_end);
  v3 centre = mk3(0.f, 0.f, 0.f); for (int k = 0; k < 6; k++) centre = centre + (1.0f / 6.0f) * pts[k];
  const v3 dh = hand_end - centre, de = centre - elbow, dfl = hand_end - elbow_end;
  const float distance_to_hand = sqrtf(dot(dh, dh)), distance_along_forearm = distance_to_hand, distance_along_upperarm = sqrtf(dot(de, de));
  const float forearm_length = sqrtf(dot(dfl, dfl)), upperarm_length = les;
  const bool forearm_in = around_fore && (f1 || f2), upperarm_in = around_upper && (u1 || u2);
  // cloth forces (dressing.py:34-46): x 10, only below the end effector and below 20
  const v3 ep = ld3(L + L_MISC + M_EEP);
  float fs = 0.f;
  if (greport) {
    const int entries = 2 * cl[AGX_CL_NN];   // AGX_CLOTH_NODE_CONTACTS slots per node
    const float scale = clp[AGX_CP_FORCE_SCALE], fmax = clp[AGX_CP_FORCE_MAX], below = clp[AGX_CP_EE_BELOW];
    for (int k = lane; k < entries; k += 64) {
      const float z = greport[20 + 2 * k], f = greport[20 + 2 * k + 1] * scale;
      if (f >= 0.f && z < ep.z - below && f < fmax) fs += f;
    }
  }
  const float cloth_force_sum = wave_sum(fs);
  const float ee_speed = ee_speed_of(c);
  const float pref = TKF(c, AGX_T_C_V) * (-ee_speed) + TKF(c, AGX_T_C_D) * (-cloth_force_sum);   // human_preferences(end_effector_velocity, dressing_forces)
  float reward_dressing;
  if (upperarm_in) { reward_dressing = forearm_length; if (distance_along_upperarm < upperarm_length) reward_dressing += distance_along_upperarm; }
  else if (forearm_in && distance_along_forearm < forearm_length) reward_dressing = distance_along_forearm;
  else reward_dressing = -distance_to_hand;
  const float reward = TKF(c, AGX_T_W_WIPE) * reward_dressing + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + pref;
  float rf = 0.f;
  if (lane < c.ncon) {
    const float* k = scr.con + CON_STRIDE * lane; const int* ki = (const int*)k;
    const int ta = CLI(c, ki[C_CA], AGX_C_TAG), tb = CLI(c, ki[C_CB], AGX_C_TAG);
    if ((ta == AGX_TAG_HUMAN || tb == AGX_TAG_HUMAN) && (ta == AGX_TAG_ROBOT || tb == AGX_TAG_ROBOT)) rf = k[C_LAM] / c.dt;
  }
  const float robot_f = wave_sum(rf);
  observe_dressing(c, cloth_force_sum, robot_f, gobs);
  const int s_task = c.bi[AGX_H_S_TASK];
  const float best0 = L[L_ST + s_task + AGX_DR_BEST], best = reward_dressing > best0 ? reward_dressing : best0;
  const int iteration = Li[L_ST + c.s_env + AGX_E_ITERATION];
  wave_sync();
  if (lane == 0) {
    L[L_ST + s_task + AGX_DR_BEST] = best; L[L_ST + s_task + AGX_DR_FORCE_SUM] = cloth_force_sum;
    *greward = reward;
    *gdone = (uint8_t)(iteration >= (int)TKF(c, AGX_T_EPISODE_LEN));
    if (ginfo) {
      ginfo[AGX_INFO_TOTAL_FORCE] = robot_f + cloth_force_sum;
      ginfo[AGX_INFO_TASK_SUCCESS] = (float)(best >= TKF(c, AGX_T_SUCCESS_FRAC));
      ginfo[AGX_INFO_ROBOT_FORCE] = robot_f; ginfo[AGX_INFO_TOOL_FORCE] = cloth_force_sum; ginfo[AGX_INFO_FOOD_REWARD] = reward_dressing;
      ginfo[AGX_INFO_PREF] = pref; ginfo[AGX_INFO_NCONTACT] = (float)c.ncon; ginfo[AGX_INFO_NROWS] = (float)c.nrows;
    }
  }
  wave_sync();
  store_env(c, gstate, sw);
}

// finish: everything FeedingEnv.step does after take_step (feeding.py:17-43)
AGX_DEV void env_finish_feeding(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                        float* ginfo, float* lds, int lane, const float* greport = nullptr, float* gwater = nullptr) {
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
        const int tc = base + lane;
        bool query = tc < tool1;
        if (query) for (int q = 0; q < 3; q++) if (AB[ABS * fc + q] > AB[ABS * tc + 3 + q] + spill || AB[ABS * tc + q] > AB[ABS * fc + 3 + q] + spill) query = false;
        Cand tmp;
        const bool hitl = narrowphase(c, fc, query ? tc : tool0, spill, tmp, query);
        if (wave_any(hitl)) near = true;
      }
    }
    if (!near) { food_reward -= 5.f; alive &= ~(1 << k); }
  }
  for (int k = 0; k < c.nfood; k++) if ((active_on_entry >> k & 1) && (hit_mask >> k & 1)) { food_hit -= 1.f; active &= ~(1 << k); }
#if AGX_TASK == 5   /* AGX_TASK_DRINKING (an enum: not visible to the preprocessor) */
  // DrinkingEnv.get_water_rewards (drinking.py:52-91), lane = particle: outside the cup's cylinder and within 0.03 m of the mouth -> drunk
  // (+10, teleported), else further than 0.1 m from every piece of the cup -> spilled (-1); a particle of waters_active (as it was on
  // entry) that touched the person in the last internal substep (the water kernel's report) counts for the preferences
  float tilt_term = 0.f; v3 cup_top = mk3(0.f, 0.f, 0.f);
  {
    float* body = L + L_ARENA;                                   // frames in agx_water.h's slot layout, where this env step ends
    for (int k = lane; k < 3 * c.ndof; k += 64) body[12 * (k / 3) + k % 3] = L[L_LINKP + k];
    for (int k = lane; k < 9 * c.ndof; k += 64) body[12 * (k / 9) + 3 + k % 9] = L[L_LINKR + k];
    agxw::static_frames(blob, L + L_ST, body, lane, true);
    wave_sync();
    v3 cp; m3 cR; tool_base_pose(c, cp, cR);
    const v3 p2 = mul(cR, mk3(TKF(c, AGX_T_TOOL_OBS_POS), TKF(c, AGX_T_TOOL_OBS_POS + 1), TKF(c, AGX_T_TOOL_OBS_POS + 2))) + cp;
    const m3 R2 = mul(cR, quat_to_m3(TKF(c, AGX_T_TOOL_OBS_QUAT), TKF(c, AGX_T_TOOL_OBS_QUAT + 1), TKF(c, AGX_T_TOOL_OBS_QUAT + 2), TKF(c, AGX_T_TOOL_OBS_QUAT + 3)));
    const v3 top = mul(R2, mk3(TKF(c, AGX_T_DK_TOP), TKF(c, AGX_T_DK_TOP + 1), TKF(c, AGX_T_DK_TOP + 2))) + p2;
    const v3 bot = mul(R2, mk3(TKF(c, AGX_T_DK_BOTTOM), TKF(c, AGX_T_DK_BOTTOM + 1), TKF(c, AGX_T_DK_BOTTOM + 2))) + p2;
    cup_top = top;
    const q4 q = m3_to_quat(R2);                                 // cup_euler[0] as p.getEulerFromQuaternion returns it (drinking.py:30-31)
    const float sarg = -2.0f * (q.x * q.z - q.w * q.y);
    const float roll = (sarg <= -0.99999f || sarg >= 0.99999f) ? 0.f : atan2f(2 * (q.y * q.z + q.w * q.x), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z);
    tilt_term = -fabsf(roll - 3.14159265358979f * 0.5f);
    const int* cl = c.bi + c.bi[AGX_H_OFF_CLOTH]; const float* clf = c.bf + c.bi[AGX_H_OFF_CLOTH];
    const int NN = cl[AGX_CL_NN], NS = cl[AGX_CL_NSHAPE]; const float margin = clf[cl[AGX_CL_OFF_PARAM] + AGX_CP_MARGIN];
    const int s_task = c.bi[AGX_H_S_TASK];
    const bool mine = lane < NN;
    const bool is_alive = mine && ((uint32_t)Li[L_ST + s_task + AGX_DK_ALIVE + (lane >> 5)] >> (lane & 31) & 1u);
    const bool is_active = mine && ((uint32_t)Li[L_ST + s_task + AGX_DK_ACTIVE + (lane >> 5)] >> (lane & 31) & 1u);
    float x[3] = {0.f, 0.f, 0.f}; float speed = 0.f;
    if (mine) { for (int k = 0; k < 3; k++) x[k] = gwater[3 * lane + k]; const float* vv = gwater + 3 * NN + 3 * lane; speed = sqrtf(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]); }
    bool drunk = false, spilled = false;
    if (is_alive) {
      const v3 xp = mk3(x[0], x[1], x[2]), axis = bot - top, a = xp - top, b = xp - bot, cr = cross(a, axis);
      const float cyl = TKF(c, AGX_T_TARGET_RADIUS) * sqrtf(dot(axis, axis));
      const bool inside = dot(a, axis) >= 0.f && dot(b, axis) <= 0.f && sqrtf(dot(cr, cr)) <= cyl;          // Util.points_in_cylinder (util.py:53-56)
      if (!inside) {
        const v3 d = target - xp;
        if (sqrtf(dot(d, d)) < TKF(c, AGX_T_MOUTH_DIST)) drunk = true;
        else {
          spilled = true;      // decided below: the 0.1 m query against the cup's vertices (every lane takes part in the wave-uniform loops)
        }
      }
    }
    // w.get_closest_points(self.tool, distance=0.1) (drinking.py:77): a candidate is near when one of the cup's VERTICES is within the limit
    // of the particle (the oracle's particle_near_tool: accurate to ~0.5 mm at this range, where the face-plane bound ran 10-30 % low).
    // The shape and vertex loops are wave uniform (scalar / broadcast loads); lanes without a candidate idle.
    if (wave_any(spilled)) {
      for (int sh = 0; sh < NS; sh++) {
        const int col = cl[cl[AGX_CL_OFF_SHAPE] + 4 * sh];
        const int* ci = c.bi + c.bi[AGX_H_OFF_COLL] + col * AGX_C_STRIDE; const float* cf = c.bf + c.bi[AGX_H_OFF_COLL] + col * AGX_C_STRIDE;
        if (ci[AGX_C_TAG] != AGX_TAG_TOOL) continue;
        const float* B = body + 12 * agxw::body_slot(ci[AGX_C_BODY], c.ndof, c.nhuman); const float* R = B + 3;
        const float d0 = x[0] - B[0], d1 = x[1] - B[1], d2 = x[2] - B[2];
        const float xl0 = R[0] * d0 + R[3] * d1 + R[6] * d2, xl1 = R[1] * d0 + R[4] * d1 + R[7] * d2, xl2 = R[2] * d0 + R[5] * d1 + R[8] * d2;
        const float lim = spill + margin + cf[AGX_C_RADIUS], lim2 = lim * lim;
        const float* v = c.bf + c.bi[AGX_H_OFF_VERT] + 3 * ci[AGX_C_VOFF];
        for (int k = 0; k < ci[AGX_C_NVERT]; k++) {
          const float e0 = xl0 - v[3 * k], e1 = xl1 - v[3 * k + 1], e2 = xl2 - v[3 * k + 2];
          if (e0 * e0 + e1 * e1 + e2 * e2 <= lim2) spilled = false;
        }
      }
    }
    const bool hit = is_active && greport && ((const int*)greport)[lane] != 0;
    const uint64_t m_drunk = wave_ballot(drunk), m_spilled = wave_ballot(spilled), m_hit = wave_ballot(hit);
    food_reward = 10.f * (float)popc64(m_drunk) - (float)popc64(m_spilled);
    food_hit = -(float)popc64(m_hit);
    vel_sum = wave_sum(drunk ? speed : 0.f);
    success += popc64(m_drunk);
    for (uint64_t m = m_drunk; m; m &= m - 1) {                   // teleported in ascending order, three draws each (drinking.py:74)
      const int e = ffs64(m);
      const float px = 1000.0f + 1000.0f * (float)(rng_next(r0, r1) >> 8) * (1.0f / 16777216.0f);
      const float py = 1000.0f + 1000.0f * (float)(rng_next(r0, r1) >> 8) * (1.0f / 16777216.0f);
      const float pz = 1000.0f + 1000.0f * (float)(rng_next(r0, r1) >> 8) * (1.0f / 16777216.0f);
      if (lane == e) { gwater[3 * lane] = px; gwater[3 * lane + 1] = py; gwater[3 * lane + 2] = pz; }
    }
    wave_sync();
    if (lane < 2) {
      const uint32_t gone = (uint32_t)((m_drunk | m_spilled) >> (32 * lane)), off = (uint32_t)((m_drunk | m_hit) >> (32 * lane));
      Li[L_ST + s_task + AGX_DK_ALIVE + lane] = (int)((uint32_t)Li[L_ST + s_task + AGX_DK_ALIVE + lane] & ~gone);
      Li[L_ST + s_task + AGX_DK_ACTIVE + lane] = (int)((uint32_t)Li[L_ST + s_task + AGX_DK_ACTIVE + lane] & ~off);
    }
    wave_sync();
  }
#endif
  // end-effector speed (feeding.py:22), human_preferences (env.py:237-274, feeding branch), reward
  const float ee_speed = ee_speed_of(c);
  float pref = TKF(c, AGX_T_C_V) * (-ee_speed) + TKF(c, AGX_T_C_F) * (-total_f) + TKF(c, AGX_T_C_HF) * (tool_f < 10.f ? 0.f : -tool_f)
             + TKF(c, AGX_T_C_FD) * food_hit + TKF(c, AGX_T_C_FDV) * (-vel_sum);
  v3 sp; m3 sR; tool_base_pose(c, sp, sR);
  v3 dd = target - sp;
#if AGX_TASK == 5   /* AGX_TASK_DRINKING (an enum: not visible to the preprocessor) */
  dd = target - cup_top;                                           // the top centre of the cup against the mouth (drinking.py:25-26), and the tilt term (:30-31)
  float reward = TKF(c, AGX_T_W_DISTANCE) * (-sqrtf(dot(dd, dd))) + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + TKF(c, AGX_T_W_TILT) * tilt_term + TKF(c, AGX_T_W_FOOD) * food_reward + pref;
#else
  float reward = TKF(c, AGX_T_W_DISTANCE) * (-sqrtf(dot(dd, dd))) + TKF(c, AGX_T_W_ACTION) * (-sqrtf(an2)) + TKF(c, AGX_T_W_FOOD) * food_reward + pref;
#endif
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

AGX_DEV void env_finish(const uint32_t* blob, float* gstate, const float* gaction, float* gscratch, float* gobs, float* greward, uint8_t* gdone,
                        float* ginfo, float* lds, int lane, const float* greport = nullptr, float* gwater = nullptr) {
  if constexpr (TASK == AGX_TASK_DRESSING) env_finish_dressing(blob, gstate, gaction, gscratch, gobs, greward, gdone, ginfo, lds, lane, greport);
  else if constexpr (TASK == AGX_TASK_BED_BATHING) env_finish_bed(blob, gstate, gaction, gscratch, gobs, greward, gdone, ginfo, lds, lane);
  else if constexpr (TASK == AGX_TASK_SCRATCH_ITCH) env_finish_scratch(blob, gstate, gaction, gscratch, gobs, greward, gdone, ginfo, lds, lane);
  else if constexpr (TASK == AGX_TASK_ARM_MANIPULATION) env_finish_arm(blob, gstate, gaction, gscratch, gobs, greward, gdone, ginfo, lds, lane);
  else env_finish_feeding(blob, gstate, gaction, gscratch, gobs, greward, gdone, ginfo, lds, lane, greport, gwater);
  // Non-finite guard (the reference's nearest analogue is the forced reconnect of env.py:93-97): an environment whose joint or
  // free-body state has become NaN / Inf must not poison a training batch.  Its observation and reward are zeroed, it is reported done
  // -- so the auto-reset (agx_reset_done / the masked agx_reset) replaces its state -- and AGX_INFO_NCONTACT carries AGX_INFO_NONFINITE.
  {
    const int* bi = (const int*)blob; const float* st = lds + L_ST;    // the state copy of the task layer above
    const int ndof = bi[AGX_H_NDOF], nfree = bi[AGX_H_NFREE], s_q = bi[AGX_H_S_Q], s_free = bi[AGX_H_S_FREE], od = bi[AGX_H_OBS_DIM];
    // "not finite" includes velocities nothing in these scenes can reach (1e3 rad/s, m/s): a deep start-up penetration pushed out in one
    // substep (no recovery clamp) blows up over a few steps before it overflows -- it is ended as soon as it is implausible.
    // The outputs themselves are checked too (a finite state can still give an overflowing reward term).
    wave_sync();                       // (the observation was written by whichever lane computed an entry)
    bool bad = false;
    for (int k = lane; k < 2 * ndof; k += 64) { const float v = st[(k < ndof ? s_q : bi[AGX_H_S_QD] - ndof) + k]; bad = bad || !(fabsf(v) < (k < ndof ? 3.0e38f : 1.0e3f)); }
    for (int k = lane; k < 13 * nfree; k += 64) { const float v = st[s_free + k]; bad = bad || !(fabsf(v) < (k % 13 >= 7 ? 1.0e3f : 3.0e38f)); }
    for (int k = lane; k < od; k += 64) bad = bad || !(fabsf(gobs[k]) < 3.0e38f);
    if (lane == 0) bad = bad || !(fabsf(*greward) < 3.0e38f);
    if (wave_any(bad)) {
      for (int k = lane; k < od; k += 64) gobs[k] = 0.f;
      if (lane == 0) { *greward = 0.f; *gdone = 1; }
      if (ginfo && lane < AGX_INFO_COUNT) ginfo[lane] = lane == AGX_INFO_NCONTACT ? AGX_INFO_NONFINITE : 0.f;
    }
  }
}

}  // namespace agx
