// agx_rows.h -- K5 constraint rows: motors, joint limits, tool constraint, contact normal + friction; B = M^-1 J^T.
// Part of the stepper (see agx_step.h for the overview); included by agx_step.h only.
#pragma once

namespace agx {

// ---- K5: constraint rows -----------------------------------------------------------------------------
// accumulate the Jacobian of a unit force f / unit torque t applied to body `code` at world point x
AGX_DEV void add_jac(const Ctx& c, int code, v3 x, v3 f, v3 t, float sign, float* Jr, float* Jf) {
  const float* L = c.lds;
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) {
    v3 xr = x - ld3(L + L_MISC + M_REF);
    v3 Fa = cross(xr, f) + t;
    float F[6] = {Fa.x, Fa.y, Fa.z, f.x, f.y, f.z};
    uint64_t anc = (uint32_t)c.ldsi[L_MISC + M_ANC + ANC_WORDS * code];
    if (ANC_WORDS == 2) anc |= (uint64_t)(uint32_t)c.ldsi[L_MISC + M_ANC + 2 * code + 1] << 32;
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
AGX_DEV void row_store(const Ctx& c, const RowGeom& r, int row, int off, float bterm, float lo, float hi, int fric_of, float mu, const float* Bd) {
  float* L = c.lds; float* E = c.E + 2 * off; const int n = c.ndof;
  float D = 0.f; int e = 0;
  int a0 = 0, na = 0, b0 = 0, nb = 0;
  int alo, an; row_art_range(c, r, alo, an);
  const bool art = an > 0;
  if (art) {
    a0 = alo; na = an;
    _Pragma("unroll") for (int i = 0; i < MAX_DOF; i++) if (i >= alo && i < alo + an) {
      float acc = 0.f;
      if (Bd) acc = Bd[i];                 // (M^-1 J^T)[i] of this row, formed for the whole pass on the matrix cores (build_rows)
      else { _Pragma("unroll") for (int j = 0; j < MAX_DOF; j++) if (j >= alo && j < alo + an) acc += L[L_MINV + i * MAX_DOF + j] * r.Jr[j]; }
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
  float* H = c.H + HDR_STRIDE * row; int* Hi = (int*)H; float* X = hx_row(c.H, row); int* Xi = (int*)X;
  H[H_INVD] = D > 1e-12f ? 1.0f / D : 0.f; H[H_B] = bterm; H[H_LO] = lo; H[H_HI] = hi;
  // lane masks of the two DoF ranges: bits 0..63 (first lane slot) and 64.. (second slot)
  const uint64_t ra = na > 0 ? ((~0ull >> (64 - na)) ) : 0ull, rb = nb > 0 ? ((~0ull >> (64 - nb))) : 0ull;
  uint64_t mlo = 0ull, mhi = 0ull;
  if (na > 0) { if (a0 < 64) mlo |= ra << a0; if (a0 + na > 64) mhi |= a0 >= 64 ? ra << (a0 - 64) : ra >> (64 - a0); }
  if (nb > 0) { if (b0 < 64) mlo |= rb << b0; if (b0 + nb > 64) mhi |= b0 >= 64 ? rb << (b0 - 64) : rb >> (64 - b0); }
  Xi[H_PACK] = a0 | (na << 8) | (b0 << 16) | (nb << 24); Hi[H_OFF] = off | (mhi ? (int)(1u << OFF_TWO_BIT) : 0);
  if constexpr (HDR_WIDE) {
    Hi[H_N] = na + nb; Hi[H_NA] = na; Hi[H_AB] = (4 * a0 + H_AB_BIAS) | ((4 * (b0 - na) + H_AB_BIAS) << 16);
    // the two compact tables of the wide row-local sweep (agx_ctx.h, agx_pgs_lvw.h)
    float* Q = c.H + HQ_BASE + HQ_STRIDE * row; int* Qi = (int*)Q; int* P = (int*)(c.H + HP_BASE + HP_STRIDE * row);
    Q[0] = H[H_INVD]; Q[1] = bterm;
    if (fric_of >= 0) { Qi[2] = 4 * fric_of; Q[3] = H[H_INVD] != 0.f ? mu : 0.f; } else { Q[2] = lo; Q[3] = hi; }
    P[0] = (8 * off) | ((na + nb) << 16) | (na << 24); P[1] = Hi[H_AB];
  }
  Xi[H_M2] = (int)(uint32_t)mhi; X[H_MU] = mu; Xi[H_MLO] = (int)(uint32_t)mlo; Xi[H_MHI] = (int)(uint32_t)(mlo >> 32);
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

// Non-contact row slots, in row order (the oracle builds them in the same order): motors of DoF 0..MAX_DOF-1, joint limits
// (DoF, side), the 6 rows of the tool constraint.  One lane per slot, NC_PASSES passes of 64 slots.
constexpr int NC_TOOLS = TASK == AGX_TASK_ARM_MANIPULATION ? 2 : 1;            // fixed constraints of tools (two-armed robots hold a second one, AGX_T_TOOL2_BODY)
constexpr int NC_SLOTS = 3 * MAX_DOF + 6 * NC_TOOLS, NC_PASSES = (NC_SLOTS + 63) / 64;
AGX_DEV void build_rows(Ctx& c) {
  float* L = c.lds; const int lane = c.lane, n = c.ndof; const float dt = c.dt;
  const float erp = PRM(c, AGX_P_ERP), cerp = PRM(c, AGX_P_CONTACT_ERP);
  int maxrows = (int)PRM(c, AGX_P_MAX_ROWS); if (maxrows > MAX_ROWS) maxrows = MAX_ROWS;
  int maxent = (int)PRM(c, AGX_P_MAX_ENTRIES); if (maxent > SCR_ENT / 2) maxent = SCR_ENT / 2;
  if (lane == 0) { c.E[0] = 0.f; c.E[1] = 0.f; }               // entry 0 of the arena is the zero pair
  if constexpr (HDR_WIDE) { if (lane < HQ_STRIDE + HP_STRIDE) {                   // the idle row of the wide sweep's tables: no pairs, no effective mass
    if (lane < HQ_STRIDE) c.H[HQ_BASE + HQ_STRIDE * HW_DUMMY + lane] = 0.f;
    else ((int*)c.H)[HP_BASE + HP_STRIDE * HW_DUMMY + lane - HQ_STRIDE] = lane == HQ_STRIDE ? 0 : (H_AB_BIAS | (H_AB_BIAS << 16)); } }
  int nnc = 0, ent = 1;                                         // non-contact rows / coefficient pairs so far
  // contact rows: lane = contact; normal rows first, then one friction row per contact (AGX_P_FRICTION_DIRS = 2: a second block of
  // friction rows along n x t behind the first)
  const int fd = (int)PRM(c, AGX_P_FRICTION_DIRS) == 2 ? 2 : 1;
  v3 tdir = mk3(0, 0, 0);
  int nc = c.ncon, ccnt = 0, cincl = 0, tot = 0, entN = 0, entF = 0;
  int ba = 0, bb = 0; v3 pa = mk3(0, 0, 0), pb = pa, nn = pa; float dist = 0.f, mu = 0.f;
  RowGeom rn; row_clear(rn);
  // the row kinds go through ONE row_store call site (its M^-1 J^T product is the bulk of this phase's code):
  // phases [0, NC_PASSES) non-contact slots, NC_PASSES contact normals, NC_PASSES + 1 contact friction
  _Pragma("nounroll") for (int ph = 0; ph < NC_PASSES + 1 + fd; ph++) {
    RowGeom R; row_clear(R);
    int rrow = 0, roff = 0, rfric = -1; float rb = 0.f, rlo = 0.f, rhi = 0.f, rmu = 0.f; bool go = false;
    if (ph < NC_PASSES) {
      const int slot = 64 * ph + lane;
      if (slot < MAX_DOF) {
        const int d = slot;
        if (d < n && RBF(c, d, AGX_R_MAXF) > 0.f && !FROZEN(c, d)) {
          go = true; if (d < c.nrobot) R.robot = true; else R.human = true;
          _Pragma("unroll") for (int q = 0; q < MAX_DOF; q++) R.Jr[q] = (q == d) ? 1.f : 0.f;
          // Agent.control (agent.py:28-33): POSITION_CONTROL motor, target dv = kp (q*-q)/dt + kd (0 - qd)
          // a human that is not an agent holds its pose with the per-env reactive gain / force (human.py:124-127), if there is one
          float kp = RBF(c, d, AGX_R_KP), maxf = RBF(c, d, AGX_R_MAXF);
          if (d >= c.nrobot && L[L_ST + c.s_env + AGX_E_HUMAN_KP] > 0.f) { kp = L[L_ST + c.s_env + AGX_E_HUMAN_KP]; maxf = L[L_ST + c.s_env + AGX_E_HUMAN_MAXF]; }
          rb = kp * (L[L_ST + c.s_qt + d] - L[L_ST + c.s_q + d]) / dt + RBF(c, d, AGX_R_KD) * (0.f - L[L_VEL + d]);
          float lim = maxf * dt; rlo = -lim; rhi = lim;
        }
      } else if (slot < 3 * MAX_DOF) {
        const int d = (slot - MAX_DOF) >> 1, side = (slot - MAX_DOF) & 1;
        if (d < n && RBI(c, d, AGX_R_HAS_LIMIT) && !FROZEN(c, d)) {
          float q = L[L_ST + c.s_q + d];
          float gap = side == 0 ? q - DLO(c, d) : DHI(c, d) - q;
          if (gap < PRM(c, AGX_P_LIMIT_ACT)) {
            go = true; if (d < c.nrobot) R.robot = true; else R.human = true;
            const float sg = side == 0 ? 1.f : -1.f;
            _Pragma("unroll") for (int q2 = 0; q2 < MAX_DOF; q2++) R.Jr[q2] = (q2 == d) ? sg : 0.f;
            float rv = sg * L[L_VEL + d];
            rb = gap > 0 ? (-gap / dt - rv) : (-gap * erp / dt - rv);
            rlo = 0.f; rhi = 1e30f;
          }
        }
      } else if (slot < NC_SLOTS && c.nfree > 0) {
        // tool fixed constraint (tool.py:46-47): parent frame = end-effector frame o tool offset, child frame = the tool's
        // base (URDF root link) frame
        const int kk = slot - 3 * MAX_DOF, second = kk >= 6, k = second ? kk - 6 : kk;
        go = !second || TKI(c, AGX_T_TOOL2_BODY) > 0;
        v3 eep = ld3(L + L_MISC + M_EEP); m3 eeR = ldm3(L + L_MISC + M_EER);
        int o_tp = AGX_T_TOOL_POS, o_tq = AGX_T_TOOL_QUAT, tb = c.bi[AGX_H_TOOL_BODY], link = TKI(c, AGX_T_EE_LINK);
        if (second && go) {   // the second tool hangs from the other end effector
          link = TKI(c, AGX_T_EE2_LINK); tb = TKI(c, AGX_T_TOOL2_BODY); o_tp = AGX_T_TOOL2_POS; o_tq = AGX_T_TOOL2_QUAT;
          const v3 lp = ld3(L + L_LINKP + 3 * link); const m3 LR = ldm3(L + L_LINKR + 9 * link);
          eep = mul(LR, mk3(TKF(c, AGX_T_EE2_POS), TKF(c, AGX_T_EE2_POS + 1), TKF(c, AGX_T_EE2_POS + 2))) + lp;
          eeR = mul(LR, quat_to_m3(TKF(c, AGX_T_EE2_QUAT), TKF(c, AGX_T_EE2_QUAT + 1), TKF(c, AGX_T_EE2_QUAT + 2), TKF(c, AGX_T_EE2_QUAT + 3)));
        }
        v3 pivA = mul(eeR, mk3(TKF(c, o_tp), TKF(c, o_tp + 1), TKF(c, o_tp + 2))) + eep;
        m3 frameA = mul(eeR, quat_to_m3(TKF(c, o_tq), TKF(c, o_tq + 1), TKF(c, o_tq + 2), TKF(c, o_tq + 3)));
        const m3 FR = ldm3(L + L_FREER + 9 * tb);
        v3 pivB = mul(FR, mk3(FBF(c, tb, AGX_F_REFPOS), FBF(c, tb, AGX_F_REFPOS + 1), FBF(c, tb, AGX_F_REFPOS + 2))) + ld3(L + L_ST + c.s_free + 13 * tb);
        m3 frameB = mul(FR, quat_to_m3(FBF(c, tb, AGX_F_REFQUAT), FBF(c, tb, AGX_F_REFQUAT + 1), FBF(c, tb, AGX_F_REFQUAT + 2), FBF(c, tb, AGX_F_REFQUAT + 3)));
        float lim = TKF(c, AGX_T_TOOL_MAXF) * dt; rlo = -lim; rhi = lim;
        if (k < 3) {
          v3 nrm = mk3(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f);
          row_pair(c, R, link, pivA, AGX_BODY_FREE0 + tb, pivB, nrm, mk3(0, 0, 0));
          rb = -comp(pivA - pivB, k) * erp / dt - row_velocity(c, R);
        } else {
          float ang[3]; m3_to_euler_xyz(mul_at(frameA, frameB), ang);
          const int q = k - 3;
          // (column q of frameA by selects: a dynamically indexed private array would live in scratch memory)
          v3 axw = q == 0 ? mk3(frameA.a[0], frameA.a[3], frameA.a[6]) : (q == 1 ? mk3(frameA.a[1], frameA.a[4], frameA.a[7]) : mk3(frameA.a[2], frameA.a[5], frameA.a[8]));
          row_pair(c, R, link, pivA, AGX_BODY_FREE0 + tb, pivB, mk3(0, 0, 0), axw);
          rb = (q == 0 ? ang[0] : (q == 1 ? ang[1] : ang[2])) * erp / dt - row_velocity(c, R);
        }
      }
      const int cnt = go ? row_entries(c, R) : 0;
      const uint64_t am = wave_ballot(go);
      rrow = nnc + wave_rank(am); roff = ent + wave_scan_excl(cnt);
      nnc += popc64(am); ent += wave_sum_i(cnt);
    } else if (ph == NC_PASSES) {
      const bool has = lane < nc;
      if (has) {
        const float* k = c.gcon + CON_STRIDE * lane; const int* ki = (const int*)k;
        ba = ki[C_BA]; bb = ki[C_BB]; pa = ld3(k + C_PA); pb = ld3(k + C_PB); nn = ld3(k + C_N); dist = k[C_DIST]; mu = k[C_MU];
        row_pair(c, rn, ba, pa, bb, pb, nn, mk3(0, 0, 0));
      }
      ccnt = has ? row_entries(c, rn) : 0;
      cincl = wave_scan_excl(ccnt) + ccnt;
      // largest prefix of the contact list that fits the row and coefficient budgets; the contacts beyond it are
      // dropped and counted as overflow
      const bool fits = has && (nnc + (1 + fd) * (lane + 1) <= maxrows) && (ent + (1 + fd) * cincl <= maxent);
      const int kept = popc64(wave_ballot(fits));
      c.overflow += nc - kept; nc = kept;
      tot = wave_bcast_i(cincl, nc > 0 ? nc - 1 : 0);
      entN = ent; entF = ent + (nc > 0 ? tot : 0);
      R = rn; go = lane < nc; rrow = nnc + lane; roff = entN + cincl - ccnt; rlo = 0.f; rhi = 1e30f;
      if (go) {
        const float rv = row_velocity(c, rn); rb = dist > 0 ? (-dist / dt - rv) : (-dist * cerp / dt - rv);
        const float sp = PRM(c, AGX_P_SPLIT_PEN); if (sp > 0.f && dist < -sp) rb = -rv;      // split impulse: no positional term below the threshold (agx_blob.h)
      }
    } else {
      const int k2 = ph - NC_PASSES - 1;                            // 0: first friction direction, 1: the second (n x t)
      go = lane < nc; rrow = nnc + (1 + k2) * nc + lane; roff = entF + k2 * tot + cincl - ccnt; rfric = nnc + lane; rmu = mu;
      if (go) {
        // friction direction: lateral slip direction if it is resolvable, else the first plane-space tangent
        v3 t;
        if (k2 == 0) {
          v3 vr = point_velocity(c, ba, pa) - point_velocity(c, bb, pb);
          t = vr - dot(vr, nn) * nn;
          float l2 = dot(t, t);
          if (l2 > PRM(c, AGX_P_FRIC_EPS)) t = (1.0f / sqrtf(l2)) * t; else plane_space(nn, t);
          tdir = t;
        } else t = cross(nn, tdir);
        row_pair(c, R, ba, pa, bb, pb, t, mk3(0, 0, 0));
        rb = -row_velocity(c, R);
      }
    }
    // B = J M^-1 of the articulated part of this pass's rows (lane = row) on the matrix cores: the rows' Jacobians go through LDS (the
    // collision arena is dead by now), v_mfma_f32_32x32x2_f32 steps over the DoFs into one 32 x MAX_DOF tile per 32 rows, and every lane
    // reads its row of the product back.  As a per-lane double loop over the DoFs this product was the bulk of this phase: 72 k of the
    // build kernel's 255 k cycles for a BedBathingSawyer environment (tools/gpu_build_phases.py).
    const float* Bd = nullptr;
    if constexpr (MAX_DOF <= 32) {
      if (wave_any(go && (R.robot || R.human))) {
        constexpr int JS = MAX_DOF | 1;                              // odd stride: lane = row reads / writes without bank conflicts
        static_assert(64 * JS <= ARENA_WORDS, "the dense Jacobians of a pass fit the arena");
        float* JD = L + L_ARENA;
        _Pragma("unroll") for (int q = 0; q < MAX_DOF; q++) JD[JS * lane + q] = go ? R.Jr[q] : 0.f;
        wave_sync();
        const bool two = (wave_ballot(go) >> 32) != 0ull;
        const int li = lane & 31, hb = lane >> 5;
        Acc16 t0, t1; acc16_zero(t0); acc16_zero(t1);
        for (int k0 = 0; k0 < n; k0 += 2) {
          const int k = k0 + hb; const bool kk = k < n;
          const float mb = (kk && li < n) ? L[L_MINV + k * MAX_DOF + li] : 0.f;
          wave_mfma_32x32x2(kk ? JD[JS * li + k] : 0.f, mb, t0);
          if (two) wave_mfma_32x32x2(kk ? JD[JS * (li + 32) + k] : 0.f, mb, t1);
        }
        wave_sync();                                                  // every lane has read the Jacobians: their place takes the product
        for (int v = 0; v < 16; v++) {
          const int i = (v & 3) + 8 * (v >> 2) + 4 * hb;
          if (li < n) { JD[JS * i + li] = acc16_get(t0, v); if (two) JD[JS * (i + 32) + li] = acc16_get(t1, v); }
        }
        wave_sync();
        Bd = JD + JS * lane;
      }
    }
    if (go) row_store(c, R, rrow, roff, rb, rlo, rhi, rfric, rmu, Bd);
  }
  c.ncon = nc; c.first_normal = nnc; c.nrows = nnc + (1 + fd) * nc; c.nent = entF + (nc > 0 ? fd * tot : 0);
  wave_sync();
}

}  // namespace agx
