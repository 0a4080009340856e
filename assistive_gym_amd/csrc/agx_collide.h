// agx_collide.h -- K2/K3 collision: speculative AABBs, pair-group broadphase, per-lane GJK narrowphase, contact selection.
// Part of the stepper (see agx_step.h for the overview); included by agx_step.h only.
#pragma once

namespace agx {

// ---- K2/K3: collision ----------------------------------------------------------------------------
struct Cand { v3 pa, pb, n; float dist, gap; };
constexpr float GJK_FAR_MARGIN = 1e-4f;   // the separating-axis early-out of gjk_distance only fires this far beyond the contact limit

AGX_DEV void make_shape(const Ctx& c, int col, v3 shift, gjk_shape& s) {
  s.n = CLI(c, col, AGX_C_NVERT);
  s.v = c.bf + c.o_vert + 3 * CLI(c, col, AGX_C_VOFF);
  v3 p; body_xf(c, CLI(c, col, AGX_C_BODY), s.R, p);
  s.p = p - shift; s.box = false;
  s.c = mul(s.R, mk3(CLF(c, col, AGX_C_AABB_C), CLF(c, col, AGX_C_AABB_C + 1), CLF(c, col, AGX_C_AABB_C + 2))) + s.p;
}
// closest features of colliders (ca, cb); true if the separation (radii included) is below limit.
// Wave-uniform: every lane calls it, lanes without a pair pass has = false.
AGX_DEV bool narrowphase(const Ctx& c, int ca, int cb, float limit, Cand& out, bool has) {
  const float* AB = c.lds + L_ARENA;
  gjk_shape sa, sb;
  sa.v = nullptr; sa.n = 0; sa.box = false; sa.c = mk3(0.f, 0.f, 0.f); sb.v = nullptr; sb.n = 0; sb.box = false; sb.c = sa.c;
  v3 shift = mk3(0.f, 0.f, 0.f);
  float ra = 0.f, rb = 0.f;
  bool ok = has;
  if (has) {
    shift = mk3(0.5f * (AB[ABS * ca] + AB[ABS * ca + 3]), 0.5f * (AB[ABS * ca + 1] + AB[ABS * ca + 4]), 0.5f * (AB[ABS * ca + 2] + AB[ABS * ca + 5]));
    make_shape(c, ca, shift, sa); make_shape(c, cb, shift, sb);
    // large static world boxes (table top, ground): clip to the neighbourhood of A (see oracle)
    if (CLI(c, cb, AGX_C_BODY) == AGX_BODY_WORLD && sb.n == 8 && (CLI(c, cb, AGX_C_TAG) == AGX_TAG_TABLE || CLI(c, cb, AGX_C_TAG) == AGX_TAG_PLANE)) {
      sb.box = true;
      float lo[3], hi[3];
      for (int k = 0; k < 3; k++) {
        lo[k] = fmaxf(AB[ABS * cb + k], AB[ABS * ca + k] - AGX_BOX_CLIP); hi[k] = fminf(AB[ABS * cb + 3 + k], AB[ABS * ca + 3 + k] + AGX_BOX_CLIP);
        if (hi[k] < lo[k]) ok = false;
      }
      sb.lo = mk3(lo[0], lo[1], lo[2]) - shift; sb.hi = mk3(hi[0], hi[1], hi[2]) - shift; sb.c = 0.5f * (sb.lo + sb.hi);
    }
    ra = CLF(c, ca, AGX_C_RADIUS); rb = CLF(c, cb, AGX_C_RADIUS);
  }
  float d; v3 pa, pb, n;
  const bool pen = gjk_distance(sa, sb, PRM(c, AGX_P_GJK_TOL), (int)PRM(c, AGX_P_GJK_MAXIT), limit + ra + rb + GJK_FAR_MARGIN, ok, d, pa, pb);
  if (!ok) return false;
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
constexpr int CAND_STRIDE = 8;
constexpr int TW_WORDS = 6 * (MAX_DOF + MAX_FREE);        // twist (w, v at the reference point M_REF) of every moving body, see rel_travel()
constexpr int WL_FIT = (ARENA_WORDS - ABS * MAX_COLL - TW_WORDS) / (1 + CAND_STRIDE);
constexpr int WL_MAX = WL_FIT < 200 ? WL_FIT : 200;      // worklist entries the arena has room for
constexpr int A_WL = ABS * MAX_COLL;                      // int[WL_MAX]: a | b << 9 | group << 18 | face-manifold point << 24
constexpr int WL_KEY_MASK = ~((511 << 9) | (3 << 24));    // (group, A collider) of a worklist entry
constexpr int A_CAND = A_WL + WL_MAX;                   // float[WL_MAX][CAND_STRIDE]: gap, pa, n, dist (pb = pa - dist n)
constexpr int A_TW = A_CAND + WL_MAX * CAND_STRIDE;     // float[MAX_DOF + MAX_FREE][6]
static_assert(A_TW + TW_WORDS <= ARENA_WORDS, "collision workspace exceeds the arena");
static_assert(MAX_COLL <= 512, "collider indices are packed in 9 bits");
// the candidate words of the last entries hold the two collider lists of a sweep (128 + 128 indices; afterwards the selection's slot list, 64 ints):
// bytes where collider indices fit a byte -- 8 entries instead of 16.  For the feeding variant that is 174 usable entries instead of 166, and an
// ordinary FeedingJaco substep has ~170 candidate pairs (150 of them food x spoon pieces): one flush of three narrowphase passes instead of
// 163 + 6 = four (profiles/r05/r05x_*)
constexpr bool LIST_BYTES = MAX_COLL <= 256;
constexpr int LIST_TAIL = LIST_BYTES ? 8 : 16;
constexpr int WL_CAP = WL_MAX - LIST_TAIL;
template <bool B> struct ListIndex { typedef unsigned short type; };
template <> struct ListIndex<true> { typedef uint8_t type; };
typedef ListIndex<LIST_BYTES>::type list_t;
static_assert(WL_CAP >= 128, "a flush of at least two full passes");

AGX_DEV void emit_from_cand(Ctx& c, int slot, int idx) {
  const float* L = c.lds; const int pr = c.ldsi[L_ARENA + A_WL + idx]; const float* cd = L + L_ARENA + A_CAND + CAND_STRIDE * idx;
  Cand k; k.gap = cd[0]; k.pa = ld3(cd + 1); k.n = ld3(cd + 4); k.dist = cd[7]; k.pb = k.pa - k.dist * k.n;
  emit_contact(c, slot, pr & 511, (pr >> 9) & 511, k);
}
// Face manifold (see face_manifold in oracle/agx_oracle.c and AGX_FACE_* in agx_blob.h): a collider resting on the top
// face of a static world box touches it along a face or an edge, where GJK's closest point is not unique and a single
// contact point makes the body rock.  Such a pair occupies 1 + AGX_FACE_EXTRA consecutive worklist entries: entry 0
// is the pair's first contact (GJK decides whether the pair is in contact and gives the normal; the point itself is
// re-anchored at the first vertex, in model order, within AGX_FACE_BAND of the lowest one, because GJK's witness point
// on a flat face is arbitrary), entry e the e-th additional vertex contact -- among the vertices of A above the box's
// footprint and inside the band, the one farthest (horizontally) from the points chosen before it, if that is at
// least AGX_FACE_SPREAD.
AGX_DEV bool face_box(const Ctx& c, int cb) {
  return CLI(c, cb, AGX_C_BODY) == AGX_BODY_WORLD && CLI(c, cb, AGX_C_NVERT) == 8 && (CLI(c, cb, AGX_C_TAG) == AGX_TAG_TABLE || CLI(c, cb, AGX_C_TAG) == AGX_TAG_PLANE);
}
// contact e of the face manifold of pair (ca, cb): e = 0 the re-anchored first contact, e >= 1 the e-th extra one given
// the first contact point p0; false if there is none
AGX_DEV bool face_point(const Ctx& c, int ca, int cb, int e, v3 p0, Cand& out) {
  const float* AB = c.lds + L_ARENA;
  const int n = CLI(c, ca, AGX_C_NVERT);
  const float* V = c.bf + c.o_vert + 3 * CLI(c, ca, AGX_C_VOFF);
  m3 R; v3 p; body_xf(c, CLI(c, ca, AGX_C_BODY), R, p);
  const float x0 = AB[ABS * cb], y0 = AB[ABS * cb + 1], x1 = AB[ABS * cb + 3], y1 = AB[ABS * cb + 4], top = AB[ABS * cb + 5];   // radius included
  float zmin = 3.0e38f;
  for (int v = 0; v < n; v++) { const v3 w = mul(R, mk3(V[3 * v], V[3 * v + 1], V[3 * v + 2])) + p; if (w.x >= x0 && w.x <= x1 && w.y >= y0 && w.y <= y1 && w.z < zmin) zmin = w.z; }
  float cx[1 + AGX_FACE_EXTRA], cy[1 + AGX_FACE_EXTRA];
  cx[0] = p0.x; cy[0] = p0.y;
#pragma unroll
  for (int q = 1; q <= AGX_FACE_EXTRA; q++) { cx[q] = 0.f; cy[q] = 0.f; }
  v3 pick = mk3(0.f, 0.f, 0.f);
  if (e == 0) {   // the pair's first contact, re-anchored at the first vertex (model order) inside the band
    bool found = false;
    for (int v = 0; v < n && !found; v++) {
      const v3 w = mul(R, mk3(V[3 * v], V[3 * v + 1], V[3 * v + 2])) + p;
      if (w.x >= x0 && w.x <= x1 && w.y >= y0 && w.y <= y1 && w.z <= zmin + AGX_FACE_BAND) { pick = w; found = true; }
    }
    if (!found) return false;
  }
#pragma unroll
  for (int q = 1; q <= AGX_FACE_EXTRA; q++) {
    if (q > e) break;
    int bi = -1; float bd = AGX_FACE_SPREAD * AGX_FACE_SPREAD; v3 bw = mk3(0.f, 0.f, 0.f);
    for (int v = 0; v < n; v++) {
      const v3 w = mul(R, mk3(V[3 * v], V[3 * v + 1], V[3 * v + 2])) + p;
      if (!(w.x >= x0 && w.x <= x1 && w.y >= y0 && w.y <= y1) || w.z > zmin + AGX_FACE_BAND) continue;
      float dmin = 3.0e38f;
#pragma unroll
      for (int k = 0; k <= AGX_FACE_EXTRA; k++) if (k < q) { const float dx = w.x - cx[k], dy = w.y - cy[k]; dmin = fminf(dmin, dx * dx + dy * dy); }
      if (dmin > bd) { bd = dmin; bi = v; bw = w; }
    }
    if (bi < 0) return false;
    cx[q] = bw.x; cy[q] = bw.y; pick = bw;
  }
  const float ra = CLF(c, ca, AGX_C_RADIUS);
  out.n = mk3(0.f, 0.f, 1.f); out.pa = mk3(pick.x, pick.y, pick.z - ra); out.pb = mk3(pick.x, pick.y, top); out.dist = pick.z - ra - top;
  return true;
}
struct CollideState { int ncon, near_mask, overflow, maxc, nqpt; };
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
// twist of the body a collider is attached to, about the common reference point M_REF: velocity field u(x) = v + w x (x - ref); zero for
// the static bodies.  Filled at the start of collide().
AGX_DEV void body_twist(const Ctx& c, int code, v3& w, v3& v) {
  int slot = -1;
  if (code >= 0 && code < AGX_BODY_ROBOT_BASE) slot = code;
  else if (code >= AGX_BODY_FREE0 && code < AGX_BODY_HUMAN0) slot = MAX_DOF + (code - AGX_BODY_FREE0);
  if (slot < 0) { w = mk3(0.f, 0.f, 0.f); v = w; return; }
  const float* T = c.lds + L_ARENA + A_TW + 6 * slot;
  w = ld3(T); v = ld3(T + 3);
}
// Upper bound on |v_n| dt of a contact of the pair (a, b) from the RELATIVE motion of the two bodies.  With the relative velocity field
// u(x) = v_A(x) - v_B(x) = dv + dw x (x - ref) the contact's normal velocity is v_n = n . (v_A(pa) - v_B(pb)) = n . u(pa), because
// pa - pb is parallel to n (closest points, or the penetration direction).  pa lies within ext_a (half the diagonal of a's world box) of
// the box centre ca:  |v_n| <= |u(ca)| + |dw| ext_a.  For a sphere (one-vertex core) pa - ca = -r n and the rotation term has no normal
// component at all: |v_n| <= |u(c)| at the sphere's centre, whichever side the sphere is on.
// The per-collider travel distances (AB[.][6], absolute speeds) bound the same quantity by |v_A| + |v_B|: for bodies that move TOGETHER
// -- the food riding on the spoon while the arm swings it, 180 of the ~225 narrowphase pairs of a FeedingJaco substep -- that is the speed
// of the arm, this is ~0.  Used wherever a pair is dropped because it cannot produce a solver row (predicted gap = dist + v_n dt >=
// slack): the set of contacts is unchanged, only the work is.
AGX_DEV float rel_travel(const Ctx& c, int a, int b) {
  const float* AB = c.lds + L_ARENA;
#ifdef AGX_NO_REL_TRAVEL   // build-time knob for same-box A/B runs and the bit-identity check of tests/test_emu_parity.py
  return AB[ABS * a + 6] + AB[ABS * b + 6];
#endif
  v3 wa, va, wb, vb;
  body_twist(c, CLI(c, a, AGX_C_BODY), wa, va); body_twist(c, CLI(c, b, AGX_C_BODY), wb, vb);
  const bool sa = CLI(c, a, AGX_C_NVERT) == 1, sb = CLI(c, b, AGX_C_NVERT) == 1;
  const int e = (!sa && sb) ? b : a;                               // evaluate at the sphere's centre if there is one
  const v3 lo = ld3(AB + ABS * e), hi = ld3(AB + ABS * e + 3);
  const v3 ce = 0.5f * (lo + hi), d = hi - lo;
  const v3 dw = wa - wb;
  const v3 u = (va - vb) + cross(dw, ce - ld3(c.lds + L_MISC + M_REF));
  const float rot = (sa || sb) ? 0.f : sqrtf(dot(dw, dw)) * 0.5f * sqrtf(dot(d, d));
  const float t = 1.001f * (sqrtf(dot(u, u)) + rot) * c.dt + 1e-6f;
  if constexpr (ABS == 7) return fminf(t, AB[ABS * a + 6] + AB[ABS * b + 6]);       // (the old layout: never above the sum of the absolute travels)
  else return t;
}
#ifdef AGX_SWEEP_TWO_SIDED
#define AGX_SWEEP_ONE_SIDED_COND false
#else
#define AGX_SWEEP_ONE_SIDED_COND (CLI(c, a, AGX_C_NVERT) == 1)
#endif
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
  const float* AB = L + L_ARENA;
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
    int b = 0, sub = 0;
    float lim = brk;
    if (has) {
      const int pr = WL[i]; a = pr & 511; b = (pr >> 9) & 511; g = (pr >> 18) & 63; sub = (pr >> 24) & 3;
      // A row is only built for predicted gap = dist + v_n dt < slack, and |v_n| dt is bounded by the travel distances
      // the AABBs were grown by.  Unless the task asks whether a manifold point exists (group flag bit 1), a pair
      // further apart than that is of no interest and its GJK may stop at the first separating axis that proves it
      // (pairs such as two idle fingers 4 mm apart otherwise run to full convergence every substep).
      if (!(GRI(c, g, AGX_G_FLAGS) & (2 | 64))) lim = fminf(brk, slack + rel_travel(c, a, b) + 1e-5f);
    }
    k.n = mk3(0.f, 0.f, 0.f); k.pa = k.n; k.pb = k.n; k.dist = 0.f;
    bool hit = narrowphase(c, a, b, lim, k, has && sub == 0);
#ifdef AGX_NARROWPHASE_TWICE   // timing experiment: the narrowphase of every pass a second time (same result) -- the slowdown of the step is what ONE
    { Cand k2; k2.gap = 3.0e38f; k2.n = mk3(0.f, 0.f, 0.f); k2.pa = k2.n; k2.pb = k2.n; k2.dist = 0.f;     // narrowphase costs under the chunk overlap,
      const bool h2 = narrowphase(c, a, b, lim + 1e-9f, k2, has && sub == 0);                               // i.e. the ceiling of any gain there
      if (h2 && k2.dist == 12345.678f) k.gap = 0.f; }
#endif
    // on a face GJK's closest point is an arbitrary point of the face: the first contact of a pair resting on a static
    // world box is re-anchored at a vertex as well (oracle: face_manifold)
    if (has && sub == 0 && hit && k.n.z > 0.999f && face_box(c, b) && CLI(c, a, AGX_C_NVERT) >= 2) { Cand k0; k0.gap = k.gap; if (face_point(c, a, b, 0, k.pa, k0)) k = k0; }
    if (wave_any(has && sub > 0)) {   // face-manifold entries: the GJK contact of the pair is `sub` entries back
      if (has && sub == 0) { float* cd = CD + CAND_STRIDE * i; st3(cd + 1, k.pa); st3(cd + 4, k.n); }
      wave_sync();
      if (has && sub > 0) {
        const float* sd = CD + CAND_STRIDE * (i - sub);
        hit = sd[6] > 0.999f && face_point(c, a, b, sub, ld3(sd + 1), k) && k.dist < brk;
      }
    }
    if (has) {
      if (hit) {
        near = true;
        v3 vr = point_velocity(c, CLI(c, a, AGX_C_BODY), k.pa) - point_velocity(c, CLI(c, b, AGX_C_BODY), k.pb);
        float pg = k.dist + dot(vr, k.n) * c.dt;
        if (pg < ((GRI(c, g, AGX_G_FLAGS) & 64) ? brk : slack)) k.gap = pg;      // bit6: a row for every contact of the group inside the break distance (agx_blob.h)
      }
      float* cd = CD + CAND_STRIDE * i;
      cd[0] = k.gap; st3(cd + 1, k.pa); st3(cd + 4, k.n); cd[7] = k.dist;
    }
    // a manifold point exists: what getContactPoints(food, human) reports (agent.py:100-116)
    if constexpr (TASK == AGX_TASK_FEEDING) {
      const bool mq = has && near && (GRI(c, g, AGX_G_FLAGS) & 2) && CLI(c, a, AGX_C_TAG) == AGX_TAG_FOOD;
      if (wave_any(mq)) { any_manifold_query = true; for (int f = 0; f < c.nfood; f++) if (wave_any(mq && CLI(c, a, AGX_C_BODY) - AGX_BODY_FREE0 - food0 == f)) cs.near_mask |= 1 << f; }
    }
    if constexpr (TASK != AGX_TASK_FEEDING) {
      // manifold points of the wiping pad / the scratcher's tool links on the human: what tool.get_contact_points(human) reports for
      // those linkA, whether or not they carry force (bed_bathing.py:47-58, scratch_itch.py:51-57); the point on the human and the
      // human's link go to the finish kernel
      const bool qp = has && near && (GRI(c, g, AGX_G_FLAGS) & 2) && CLI(c, a, AGX_C_TAG) == AGX_TAG_TOOL && (TKI(c, AGX_T_PAD_LINK) >> (CLI(c, a, AGX_C_LINK) + 1) & 1) &&
                      CLI(c, b, AGX_C_TAG) == AGX_TAG_HUMAN;
      const uint64_t qm = wave_ballot(qp);
      const int slot = cs.nqpt + wave_rank(qm);
      if (qp && slot < MAX_QPT) { float* o = c.gqpt + QPT_STRIDE * slot; st3(o, k.pb); ((int*)o)[3] = CLI(c, b, AGX_C_LINK); }
      cs.nqpt += popc64(qm);
    }
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
    const int key = WL[cur] & WL_KEY_MASK;            // group and A collider
    const int g = (key >> 18) & 63, keep = wave_bcast_i(G.keep, g);
    const int i0 = cur + lane, i1 = cur + 64 + lane;
    const bool s0 = i0 < wn && (WL[i0 < wn ? i0 : 0] & WL_KEY_MASK) == key, s1 = i1 < wn && (WL[i1 < wn ? i1 : 0] & WL_KEY_MASK) == key;
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
AGX_DEV int collide_sweep(Ctx& c, int g, int aa, int ab, int b0, int b1, int gflags, float mg, int wn, int rep, const GroupRegs& G) {
  const float* AB = c.lds + L_ARENA; int* WL = c.ldsi + L_ARENA + A_WL; const int lane = c.lane;
  const bool same = gflags & 1, no_adjacent = gflags & 4;   // bit2, self-collision: not the same link, not parent and child
  // level 1: the A colliders whose box reaches the union box of the B range and vice versa (the union
  // boxes were computed by the group cull, lane g holds them), compacted in ascending order into two
  // index lists (list_t) in the tail of the candidate area (unused until the flush).  Filtering both sides
  // matters for pairs of compounds (64 spoon pieces x 44 wheelchair pieces: a handful of each are close).
  float ulo[2][3], uhi[2][3];
  for (int q = 0; q < 3; q++) { ulo[0][q] = wave_bcast(G.blo[q], g); uhi[0][q] = wave_bcast(G.bhi[q], g); ulo[1][q] = wave_bcast(G.alo[q], g); uhi[1][q] = wave_bcast(G.ahi[q], g); }
  list_t* LIST = (list_t*)(c.ldsi + L_ARENA + A_CAND + CAND_STRIDE * WL_CAP);   // [0,128): A side, [128,256): B side
  int nlive[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    const int r0 = side == 0 ? aa : b0, r1 = side == 0 ? ab : b1;
    for (int base = r0; base < r1; base += 64) {
      const int x = base + lane; bool ok = x < r1;
      if (ok) for (int q = 0; q < 3; q++) if (AB[ABS * x + q] > uhi[side][q] + mg || ulo[side][q] > AB[ABS * x + 3 + q] + mg) ok = false;
      const uint64_t m = wave_ballot(ok);
      if (ok) LIST[128 * side + nlive[side] + wave_rank(m)] = (list_t)x;
      nlive[side] += popc64(m);
    }
  }
  const int na_live = nlive[0], nb = nlive[1];
  wave_sync();
  // level 2: the pair grid of the surviving A colliders, in enumeration order
  // (groups against static world boxes: every pair is enumerated rep = 1 + AGX_FACE_EXTRA times, entry `sub` > 0
  // standing for the sub-th extra point of the face manifold)
  const int npairs = na_live * nb * rep;
  for (int base = 0; base < npairs; base += 64) {
    const int p = base + lane; bool ok = p < npairs;
    const int pq = ok ? p / rep : 0, sub = ok ? p - pq * rep : 0;
    const int ai = pq / nb; const int a = LIST[ai], b = LIST[128 + pq - ai * nb];
    ok = ok && (!same || b > a) && (sub == 0 || (face_box(c, b) && CLI(c, a, AGX_C_NVERT) >= 2));
    if (ok && no_adjacent) {
      const int la = CLI(c, a, AGX_C_BODY), lb = CLI(c, b, AGX_C_BODY);
      if (la == lb) ok = false;
      else if (la >= 0 && la < AGX_BODY_ROBOT_BASE && lb >= 0 && lb < AGX_BODY_ROBOT_BASE && (RBI(c, la, AGX_R_PARENT) == lb || RBI(c, lb, AGX_R_PARENT) == la)) ok = false;
    }
    if (ok) for (int q = 0; q < 3; q++) if (AB[ABS * a + q] > AB[ABS * b + 3 + q] + mg || AB[ABS * b + q] > AB[ABS * a + 3 + q] + mg) ok = false;
    // level 3: bounding sphere of one collider against the body-frame box of the other, both ways
    // (the second test -- b's bounding sphere against a's body-frame box -- cannot reject what the first one passed when a is a sphere: its
    // box is the point itself, so the test degenerates to two bounding spheres; skipped then: AGX_SWEEP_ONE_SIDED)
    if (ok) { const float reach = mg + rel_travel(c, a, b) + 1e-5f; ok = !sphere_box_apart(c, a, b, reach) && (AGX_SWEEP_ONE_SIDED_COND || !sphere_box_apart(c, b, a, reach)); }
    const uint64_t m = wave_ballot(ok);
    const int slot = wn + wave_rank(m);
    if (ok && slot < WL_CAP) WL[slot] = a | (b << 9) | (g << 18) | (sub << 24);
    wn += popc64(m);
  }
  wave_sync();
  return wn;
}

AGX_DEV void collide(Ctx& c) {
  float* L = c.lds; float* AB = L + L_ARENA; const int lane = c.lane;
  const float brk = PRM(c, AGX_P_CONTACT_BREAK), slack = PRM(c, AGX_P_CONTACT_SLACK);
  CollideState cs; cs.ncon = 0; cs.near_mask = 0; cs.overflow = 0; cs.nqpt = 0;
  cs.maxc = (int)PRM(c, AGX_P_MAX_CONTACTS); if (cs.maxc > MAX_CON) cs.maxc = MAX_CON;
  long long ct0 = c.timing ? wave_clock() : 0, ct1;
#define AGX_CTICK(k) if (c.timing) { ct1 = wave_clock(); c.tm[k] += ct1 - ct0; ct0 = ct1; }
  // 1. world AABBs, grown by the distance the collider can travel in this substep (speculative):
  //    the broadphase margin then only has to cover the solver slack
  if (lane < MAX_DOF + MAX_FREE) {     // twist and angular speed of every moving body, once (the chain walk is per body, not per collider)
    v3 w = mk3(0, 0, 0), v = w;
    if (lane < MAX_DOF) { if (lane < c.ndof) for (int d = lane; d >= 0; d = RBI(c, d, AGX_R_PARENT)) { const float qd = L[L_VEL + d]; w = w + qd * ld3(L + L_S + 6 * d); v = v + qd * ld3(L + L_S + 6 * d + 3); } }
    else if (lane - MAX_DOF < c.nfree) {
      const int fb = lane - MAX_DOF, o = c.ndof + 6 * fb;
      w = ld3(L + L_VEL + o + 3);
      v = ld3(L + L_VEL + o) + cross(w, ld3(L + L_MISC + M_REF) - ld3(L + L_ST + c.s_free + 13 * fb));
    }
    L[L_WMAG + lane] = sqrtf(dot(w, w));
    st3(AB + A_TW + 6 * lane, w); st3(AB + A_TW + 6 * lane + 3, v);
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
    if constexpr (ABS == 7) AB[ABS * col + 6] = grow;
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
      const float mg = (G.flags & (2 | 64)) ? brk : slack;
      // bit3 / bit4: male / female only; bit5: only while some human DoF is dynamic
      const bool wanted = !((G.flags & 8) && gender != 0) && !((G.flags & 16) && gender != 1) &&
                          !((G.flags & 32) && c.nrobot + c.nhdof <= 32 && ((~c.frozen >> c.nrobot) & ((1u << c.nhdof) - 1u)) == 0);
      if (a1 > a0 && b1 > b0 && wanted) {
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
      const float mg = (gflags & (2 | 64)) ? brk : slack;   // bit1: getContactPoints-style existence query; bit6: every contact inside the break distance is solved
      const int rep = face_box(c, b0) ? 1 + AGX_FACE_EXTRA : 1;   // B ranges are homogeneous (table boxes, the ground plane, ...)
      const int nb = (b1 - b0) * rep;
      if (ab < 0) { ab = a0; abatch = a1 - a0; }
      const int ae = ab + abatch < a1 ? ab + abatch : a1;
      int wn2 = collide_sweep(c, g, ab, ae, b0, b1, gflags, mg, wn, rep, G);
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
  c.ncon = cs.ncon; c.near_mask = cs.near_mask; c.overflow = cs.overflow; c.nqpt = cs.nqpt < MAX_QPT ? cs.nqpt : MAX_QPT;
  wave_sync();
#undef AGX_CTICK
}

}  // namespace agx
