// agx_dyn.h -- K1 kinematics, K4 articulated-body algorithm + M^-1, unconstrained velocity update.
// Part of the stepper (see agx_step.h for the overview); included by agx_step.h only.
#pragma once

namespace agx {

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
    const bool prismatic = RBI(c, d, AGX_R_JTYPE) == 1;
    const float qj = L[L_ST + c.s_q + d];
    m3 R = mul(PR, Rt);
    v3 p = mul(PR, tp) + pp;
    if (prismatic) p = p + qj * mul(R, ax); else R = mul(R, axis_angle_m3(ax, qj));
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
    // joint screw (angular; linear at the ref point): revolute (a, p x a), prismatic (0, a)
    if (RBI(c, d, AGX_R_JTYPE) == 1) { st3(L + L_S + 6 * d, mk3(0.f, 0.f, 0.f)); st3(L + L_S + 6 * d + 3, aw); }
    else { st3(L + L_S + 6 * d, aw); st3(L + L_S + 6 * d + 3, cross(p, aw)); }
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
    const float gz = PRM(c, (RBI(c, d, AGX_R_KIND) & 1) ? AGX_P_HUMAN_GRAVITY_Z : AGX_P_ROBOT_GRAVITY_Z);
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
    float Dinv = (D > 1e-30f && !FROZEN(c, d)) ? 1.0f / D : 0.0f;   // frozen DoF: static link (mass 0, human.py:104-110)
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
  // M^-1: lane j = response to a unit force on joint j (Bullet: calcAccelerationDeltasMultiDof).  M^-1 is block diagonal
  // (robot, human chain): a lane walks the links of its own articulated body only; cross-block entries are zero.
  if (lane < COLS_LANES) for (int j = lane; j < n; j += COLS_LANES) {
    const int b0 = j < c.nrobot ? 0 : c.nrobot, b1 = j < c.nrobot ? c.nrobot : n;
    float* P = A + A_COLS + lane * (MAX_BLOCK * 6);
    for (int k = 0; k < (b1 - b0) * 6; k++) P[k] = 0.f;
    float* UU = A + A_COLS + COLS_LANES * MAX_BLOCK * 6 + lane * MAX_BLOCK;   // per-lane u[] next to the column workspaces
    for (int d = b1 - 1; d >= b0; d--) {
      float u = (d == j ? 1.f : 0.f) - dot6p(L + L_S + 6 * d, P + 6 * (d - b0));
      UU[d - b0] = u;
      int par = RBI(c, d, AGX_R_PARENT);
      if (par >= 0) { float s = u * A[A_DINV + d]; for (int k = 0; k < 6; k++) P[6 * (par - b0) + k] += P[6 * (d - b0) + k] + A[A_U + 6 * d + k] * s; }
    }
    // reuse P as the acceleration workspace
    for (int d = b0; d < b1; d++) {
      int par = RBI(c, d, AGX_R_PARENT);
      float ap[6]; for (int k = 0; k < 6; k++) ap[k] = par < 0 ? 0.f : P[6 * (par - b0) + k];
      float qdd = (UU[d - b0] - dot6p(A + A_U + 6 * d, ap)) * A[A_DINV + d];
      for (int k = 0; k < 6; k++) P[6 * (d - b0) + k] = ap[k] + L[L_S + 6 * d + k] * qdd;
      L[L_MINV + d * MAX_DOF + j] = qdd;
    }
    for (int d = 0; d < n; d++) if (d < b0 || d >= b1) L[L_MINV + d * MAX_DOF + j] = 0.f;
  }
  wave_sync();
  if (c.dbg && lane < n) { c.dbg[DBG_QDD + lane] = A[A_QDD + lane]; }
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

}  // namespace agx
