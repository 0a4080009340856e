// agx_gjk.h -- per-lane convex narrowphase: GJK distance between two (vertex core + radius)
// colliders, with a fixed-direction penetration sampler when the cores overlap.
// Each lane works on its own collider pair; vertices are read from the model blob (L1/L2
// resident, shared by all environments) in the body frame and transformed on the fly.
// The GJK iteration itself is wave-uniform (lanes that are done are predicated off), so that the
// support scan of the few pairs with large hulls can be served by the whole wave (gjk_support_wave).
#pragma once

struct gjk_shape {
  const float* v;   // body-frame vertices (x,y,z)*n in the model blob
  int n;
  m3 R;             // body rotation (world)
  v3 p;             // body position minus the pair's shift point
  v3 c;             // centre of the core's box in the shifted frame (start direction of the iteration)
  bool box;         // axis-aligned world box given by lo/hi (shifted frame) instead of vertices
  v3 lo, hi;
};

// vertex . direction with a fixed rounding sequence, so that the per-lane scan and the wave-wide scan of
// gjk_support_wave rank the vertices identically
AGX_DEV float gjk_dot3(float x, float y, float z, v3 d) { return fmaf(z, d.z, fmaf(y, d.y, x * d.x)); }
// R^T d, the search direction in the body frame, same remark
AGX_DEV v3 gjk_local_dir(const m3& R, v3 d) {
  return mk3(fmaf(R.a[6], d.z, fmaf(R.a[3], d.y, R.a[0] * d.x)), fmaf(R.a[7], d.z, fmaf(R.a[4], d.y, R.a[1] * d.x)), fmaf(R.a[8], d.z, fmaf(R.a[5], d.y, R.a[2] * d.x)));
}
AGX_DEV v3 gjk_vertex0(const gjk_shape& s) {
  if (s.box) return s.lo;
  return mul(s.R, mk3(s.v[0], s.v[1], s.v[2])) + s.p;
}
// AGX_GJK_SCAN_WIDE (round 6).  The vertex scan of gjk_support is a chain of memory round trips, not of arithmetic: no address depends on a comparison,
// but each round of the loop waits for its own loads (the compiler keeps the rounds of a loop with a per-lane trip count apart, and un-does a source-level
// software pipeline).  0: the scan of rounds 3-5 -- vertex 0, then four vertices per round, then the winner loaded by index: five round trips for a
// 16-vertex spoon piece.  2 (the DEFAULT): vertex 0 inside the first round and EIGHT vertices per round while more than four are left -- three round trips.
// Same comparisons in the same order, and `R v + p` still works on a fresh 12-byte load: BIT-IDENTICAL with 0 on 13 task / robot combinations
// (512-1,024 environments x 30-40 steps each, profiles/r06/r06w_*), 615.2 / 615.7 k against 610.6 / 610.5 k env-steps/s, same box, interleaved.
// (Sixteen per round: bit-identical too, 66 instead of 62 spilled registers, 609.7 / 610.0 k against 616.5 / 615.3 k -- eight it is.)
// 1 (an A/B knob): the winner's coordinates carried along instead of re-loaded -- one round trip less, 614-621 k -- but the winner then sits in three
// separate registers, and `R v + p`, which the compiler rounds differently at every inlined call site (packed products, partly fused chains), comes out with
// other last bits whatever sequence is pinned (gjk_xf).  Two pinned sequences were run through the GPU suite: 203 of 205 tests passed each time, and each
// time two OTHER tests on ill-conditioned states (a threshold contact of a crafted pressed state; the co-op arm one step after a classifier roll-back; a
// wiping force) missed margins the suite's own rounding meets.  Not worth re-basing those margins (profiles/r06/r06u_ab_gjk_scan_wide.txt).
#ifndef AGX_GJK_SCAN_WIDE
#define AGX_GJK_SCAN_WIDE 2
#endif
#ifndef AGX_GJK_SCAN_ONE
#define AGX_GJK_SCAN_ONE 1
#endif

// R v + p of the scan's winner with ONE rounding sequence, p + fma(R2, z, fma(R1, y, [R0 x])), written with a product and a sum the compiler may not
// contract or re-associate.  (Left to the compiler, `mul(R, v) + p` came out differently at every inlined call site -- packed products, partly fused
// chains, another pattern for the third component -- so the support points of A and of B were rounded by different rules, and neither like the CPU wave
// emulator.  With the wide scan the winner sits in three separate registers and the patterns would have shifted once more: pinned instead.)
#ifdef __HIPCC__
AGX_DEV float gjk_nc_mul(float a, float b) { return __fmul_rn(a, b); }
AGX_DEV float gjk_nc_add(float a, float b) { return __fadd_rn(a, b); }
#else
AGX_DEV float gjk_nc_mul(float a, float b) { volatile float r = a * b; return r; }
AGX_DEV float gjk_nc_add(float a, float b) { volatile float r = a + b; return r; }
#endif
AGX_DEV float gjk_xf1(float r0, float r1, float r2, float x, float y, float z, float p) { return gjk_nc_add(p, fmaf(r2, z, fmaf(r1, y, gjk_nc_mul(r0, x)))); }
AGX_DEV v3 gjk_xf(const m3& R, v3 p, float x, float y, float z) {
  return mk3(gjk_xf1(R.a[0], R.a[1], R.a[2], x, y, z, p.x), gjk_xf1(R.a[3], R.a[4], R.a[5], x, y, z, p.y), gjk_xf1(R.a[6], R.a[7], R.a[8], x, y, z, p.z));
}
AGX_DEV v3 gjk_support(const gjk_shape& s, v3 d) {
  if (s.box) {
    // vertex order of the 8-corner enumeration (x major): first maximum wins, like the vertex scan
    v3 best = s.lo; float bd = dot(s.lo, d);
    for (int q = 1; q < 8; q++) {
      v3 c = mk3((q & 4) ? s.hi.x : s.lo.x, (q & 2) ? s.hi.y : s.lo.y, (q & 1) ? s.hi.z : s.lo.z);
      float t = dot(c, d);
      if (t > bd) { bd = t; best = c; }
    }
    return best;
  }
  const v3 dl = gjk_local_dir(s.R, d);
  // (AGX_GJK_SCAN_WIDE above.)  Indices are clamped to n-1: a repeated vertex never wins the strict comparison, so the first maximum is still the one
  // returned -- the same vertex as the 4-per-round scan, to the bit.
  const float* V = s.v;
#if AGX_GJK_SCAN_WIDE
  const int last = s.n - 1;
  float bd = -3.0e38f, bx = 0.f, by = 0.f, bz = 0.f; int best = 0;
#define GJK_LDV(j, kk) const int i##j = (kk) < last ? (kk) : last; const float x##j = V[3 * i##j], y##j = V[3 * i##j + 1], z##j = V[3 * i##j + 2];
#if AGX_GJK_SCAN_WIDE == 2     // the winner by INDEX, loaded again at the end (one round trip more): `R v + p` keeps the operands -- a fresh 12-byte load -- it has in the 4-per-round scan
#define GJK_CMP(j) { const float t = gjk_dot3(x##j, y##j, z##j, dl); if (t > bd) { bd = t; best = i##j; } }
#else
#define GJK_CMP(j) { const float t = gjk_dot3(x##j, y##j, z##j, dl); if (t > bd) { bd = t; bx = x##j; by = y##j; bz = z##j; } }
#endif
  int k = 0;
#if AGX_GJK_SCAN_ONE
  if (s.n > 1)       // a one-vertex core (a food particle, a bead of the tool) has nothing to scan: its support point is one round trip, not two
#endif
  {
  for (; s.n - k > 4; k += 8) {
    GJK_LDV(0, k) GJK_LDV(1, k + 1) GJK_LDV(2, k + 2) GJK_LDV(3, k + 3) GJK_LDV(4, k + 4) GJK_LDV(5, k + 5) GJK_LDV(6, k + 6) GJK_LDV(7, k + 7)
    GJK_CMP(0) GJK_CMP(1) GJK_CMP(2) GJK_CMP(3) GJK_CMP(4) GJK_CMP(5) GJK_CMP(6) GJK_CMP(7)
  }
  if (k < s.n) {
    GJK_LDV(0, k) GJK_LDV(1, k + 1) GJK_LDV(2, k + 2) GJK_LDV(3, k + 3)
    GJK_CMP(0) GJK_CMP(1) GJK_CMP(2) GJK_CMP(3)
  }
  }
#undef GJK_LDV
#undef GJK_CMP
#if AGX_GJK_SCAN_WIDE == 2
  (void)bx; (void)by; (void)bz;
  return mul(s.R, mk3(V[3 * best], V[3 * best + 1], V[3 * best + 2])) + s.p;
#else
  (void)best;
  return gjk_xf(s.R, s.p, bx, by, bz);
#endif
#else
  int best = 0;
  float bd = gjk_dot3(V[0], V[1], V[2], dl);
  const int last = s.n - 1;
  for (int k = 1; k < s.n; k += 4) {
    const int k1 = k + 1 < last ? k + 1 : last, k2 = k + 2 < last ? k + 2 : last, k3 = k + 3 < last ? k + 3 : last;
    const float x0 = V[3 * k], y0 = V[3 * k + 1], z0 = V[3 * k + 2], x1 = V[3 * k1], y1 = V[3 * k1 + 1], z1 = V[3 * k1 + 2];
    const float x2 = V[3 * k2], y2 = V[3 * k2 + 1], z2 = V[3 * k2 + 2], x3 = V[3 * k3], y3 = V[3 * k3 + 1], z3 = V[3 * k3 + 2];
    const float t0 = gjk_dot3(x0, y0, z0, dl), t1 = gjk_dot3(x1, y1, z1, dl), t2 = gjk_dot3(x2, y2, z2, dl), t3 = gjk_dot3(x3, y3, z3, dl);
    if (t0 > bd) { bd = t0; best = k; }
    if (t1 > bd) { bd = t1; best = k1; }
    if (t2 > bd) { bd = t2; best = k2; }
    if (t3 > bd) { bd = t3; best = k3; }
  }
  return mul(s.R, mk3(V[3 * best], V[3 * best + 1], V[3 * best + 2])) + s.p;
#endif
}

// Support points for all lanes of the wave at once.  Lanes with a small hull (or a box) scan their own
// vertices; a lane whose hull has more than GJK_COOP_MIN vertices would make the whole wave wait for its
// scan, so -- when there are only a few of them -- the wave serves those lanes one at a time: 64 vertices per
// step, one per lane, argmax by wave_max + ballot.  The lowest index among equal maxima wins, like the
// sequential scan's strict comparison, so both paths return the same vertex.
// (Tried at the end of round 6: two served lanes per turn and the winners loaded by their own lanes in one load behind the loop -- 618.0 / 618.4 k against
// 616.8 / 616.6 k, and NOT bit-identical: `R v + p` moved out of the loop is rounded by another pattern.  The served scans are coalesced loads; they are
// not where a pass waits.  Dropped: profiles/r06/r06z_ab_coop_scan_batched.txt.)
#ifdef AGX_GJK_NO_COOP   // build-time knob for A/B runs: every lane scans its own hull
constexpr int GJK_COOP_MIN = 32, GJK_COOP_MAX_LANES = 0;
#else
constexpr int GJK_COOP_MIN = 32, GJK_COOP_MAX_LANES = 16;
#endif
AGX_DEV v3 gjk_support_wave(const gjk_shape& s, v3 d, bool active) {
  const bool big = active && !s.box && s.n > GJK_COOP_MIN;
  uint64_t hm = wave_ballot(big);
  const bool coop = hm != 0ull && popc64(hm) <= GJK_COOP_MAX_LANES;
  v3 out = mk3(0.f, 0.f, 0.f);
  if (active && !(coop && big)) out = gjk_support(s, d);
  if (!coop) return out;
  const v3 dl = gjk_local_dir(s.R, d);
  const int lane = wave_lane();
  int lo32, hi32;
  { const unsigned long long a = (unsigned long long)(uintptr_t)s.v; lo32 = (int)(unsigned)(a & 0xffffffffull); hi32 = (int)(unsigned)(a >> 32); }
  while (hm) {
    const int h = ffs64(hm); hm &= hm - 1ull;
    const float dx = wave_bcast(dl.x, h), dy = wave_bcast(dl.y, h), dz = wave_bcast(dl.z, h);
    const int n = wave_bcast_i(s.n, h);
    const float* V = (const float*)(uintptr_t)(((unsigned long long)(unsigned)wave_bcast_i(hi32, h) << 32) | (unsigned long long)(unsigned)wave_bcast_i(lo32, h));
    int best = 0; float bd = -3.0e38f;
    for (int base = 0; base < n; base += AGX_WAVE) {
      const int k = base + lane;
      const float t = k < n ? gjk_dot3(V[3 * k], V[3 * k + 1], V[3 * k + 2], mk3(dx, dy, dz)) : -3.0e38f;
      const float m = wave_max(t);
      if (m > bd) { bd = m; best = base + ffs64(wave_ballot(t == m)); }
    }
    if (lane == h) out = mul(s.R, mk3(V[3 * best], V[3 * best + 1], V[3 * best + 2])) + s.p;
  }
  return out;
}

// iteration statistics of the narrowphase on the CPU wave emulator (tests/diag/narrowphase_passes.py); nothing on the device
#ifndef AGX_TRACE_GJK
#define AGX_TRACE_GJK(has, iters, na, nb, box, far_out, dist, far)
#endif

// closest point to the origin on triangle (a,b,c): barycentric weights
AGX_DEV void gjk_closest_tri(v3 a, v3 b, v3 c, float& wa, float& wb, float& wc) {
  v3 ab = b - a, ac = c - a, ap = -a, bp = -b, cp = -c;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { wa = 1; wb = 0; wc = 0; return; }
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { wa = 0; wb = 1; wc = 0; return; }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { float v = d1 / (d1 - d3); wa = 1 - v; wb = v; wc = 0; return; }
  float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { wa = 0; wb = 0; wc = 1; return; }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { float w = d2 / (d2 - d6); wa = 1 - w; wb = 0; wc = w; return; }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { float w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); wa = 0; wb = 1 - w; wc = w; return; }
  float den = 1.0f / (va + vb + vc);
  float v = vb * den, w = vc * den;
  wa = 1 - v - w; wb = v; wc = w;
}

// The simplex lives in registers: every access below uses compile-time indices (new vertices are
// appended with a switch on the current size, the reduction to the supporting sub-simplex is a
// chain of selects), so nothing is spilled to scratch.  Vertex order matches the oracle's
// (append at the end, order-preserving compaction): ties are broken identically.
struct gjk_pt { v3 w, a, b; };
struct gjk_simplex { gjk_pt p0, p1, p2, p3; float l0, l1, l2, l3; int n; };

// i-th of four values as a chain of register selects.  The operands are first pinned to registers:
// a conditional expression over struct members is an lvalue, which the optimiser turns into a select of
// ADDRESSES and so forces the whole simplex into scratch memory with dynamically indexed loads.
AGX_DEV float gjk_pick(int i, float a, float b, float c, float d) {
  wave_opaque(a); wave_opaque(b); wave_opaque(c); wave_opaque(d);
  float r = d;
  r = i == 2 ? c : r; r = i == 1 ? b : r; r = i == 0 ? a : r;
  return r;
}
AGX_DEV gjk_pt gjk_sel(int i, const gjk_pt& a, const gjk_pt& b, const gjk_pt& c, const gjk_pt& d) {
  gjk_pt r;
#define AGX_SEL3(f) r.f.x = gjk_pick(i, a.f.x, b.f.x, c.f.x, d.f.x); r.f.y = gjk_pick(i, a.f.y, b.f.y, c.f.y, d.f.y); r.f.z = gjk_pick(i, a.f.z, b.f.z, c.f.z, d.f.z);
  AGX_SEL3(w) AGX_SEL3(a) AGX_SEL3(b)
#undef AGX_SEL3
  return r;
}
AGX_DEV float gjk_self(int i, float a, float b, float c, float d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }

template <int A, int B, int C, int D>
AGX_DEV void gjk_face(const v3 (&W)[4], float& best, bool& any, float (&l)[4]) {
  const v3 a = W[A], b = W[B], c = W[C], d = W[D];
  v3 nrm = cross(b - a, c - a);
  float sp = -dot(a, nrm), sd = dot(d - a, nrm);
  if (sp * sd > 0) return;
  any = true;
  float wa, wb, wc; gjk_closest_tri(a, b, c, wa, wb, wc);
  v3 p = wa * a + wb * b + wc * c;
  float d2 = dot(p, p);
  if (d2 < best) { best = d2; l[0] = 0; l[1] = 0; l[2] = 0; l[3] = 0; l[A] = wa; l[B] = wb; l[C] = wc; }
}

// closest point of the simplex to the origin; compacts to the supporting sub-simplex.
// returns true if the origin is enclosed (tetrahedron case).
AGX_DEV bool gjk_solve(gjk_simplex& s, v3& v) {
  float l[4] = {0, 0, 0, 0};
  const int n = s.n;
  if (n == 1) { l[0] = 1; }
  else if (n == 2) {
    v3 d = s.p1.w - s.p0.w;
    float dd = dot(d, d), t = dd > 0 ? -dot(s.p0.w, d) / dd : 0.0f;
    if (t <= 0) l[0] = 1; else if (t >= 1) l[1] = 1; else { l[0] = 1 - t; l[1] = t; }
  } else if (n == 3) {
    gjk_closest_tri(s.p0.w, s.p1.w, s.p2.w, l[0], l[1], l[2]);
  } else {
    const v3 W[4] = {s.p0.w, s.p1.w, s.p2.w, s.p3.w};
    float best = 3.0e38f; bool any = false;
    gjk_face<0, 1, 2, 3>(W, best, any, l);
    gjk_face<0, 2, 3, 1>(W, best, any, l);
    gjk_face<0, 3, 1, 2>(W, best, any, l);
    gjk_face<1, 3, 2, 0>(W, best, any, l);
    if (!any) return true;
  }
  // order-preserving compaction: position j takes the j-th vertex with a positive weight
  const bool k0 = l[0] > 0, k1 = l[1] > 0, k2 = l[2] > 0, k3 = l[3] > 0;
  const int c0 = k0 ? 1 : 0, c1 = c0 + (k1 ? 1 : 0), c2 = c1 + (k2 ? 1 : 0), cnt = c2 + (k3 ? 1 : 0);
  // index of the j-th kept vertex
  const int i0 = k0 ? 0 : (k1 ? 1 : (k2 ? 2 : 3));
  const int i1 = (k1 && c1 == 2) ? 1 : ((k2 && c2 == 2) ? 2 : 3);
  const int i2 = (k2 && c2 == 3) ? 2 : 3;
  const gjk_pt q0 = gjk_sel(i0, s.p0, s.p1, s.p2, s.p3), q1 = gjk_sel(i1, s.p0, s.p1, s.p2, s.p3), q2 = gjk_sel(i2, s.p0, s.p1, s.p2, s.p3);
  const float m0 = gjk_self(i0, l[0], l[1], l[2], l[3]), m1 = gjk_self(i1, l[0], l[1], l[2], l[3]), m2 = gjk_self(i2, l[0], l[1], l[2], l[3]);
  s.p0 = q0; s.p1 = q1; s.p2 = q2;
  s.l0 = m0; s.l1 = cnt > 1 ? m1 : 0.f; s.l2 = cnt > 2 ? m2 : 0.f; s.l3 = 0.f;
  s.n = cnt;   // cnt <= 3 whenever the origin is not enclosed
  v = s.l0 * s.p0.w;
  if (cnt > 1) v = v + s.l1 * s.p1.w;
  if (cnt > 2) v = v + s.l2 * s.p2.w;
  return false;
}

// returns true when the cores overlap; otherwise dist / witness points (shifted frame).  Called by ALL lanes of the
// wave; lanes with `has` false only take part in the collectives.
// `far`: core distance beyond which the caller discards the pair.  Every support point w yields the lower bound
// v.w / |v| on the distance (w is the extreme point of A - B along -v); once that bound exceeds `far` the iteration
// stops and the bound is returned as dist (> far, witness points meaningless).  The caller adds a margin to `far` that
// dwarfs the rounding error of the bound, so accept / reject decisions are those of the converged distance.
AGX_DEV bool gjk_distance(const gjk_shape& sa, const gjk_shape& sb, float tol, int maxit, float far, bool has, float& dist, v3& pa, v3& pb) {
  gjk_simplex s;
  // first simplex point: the support point of A - B along -(centre(A) - centre(B)), a point of the Minkowski
  // difference that already faces the origin (about one iteration fewer per pair than starting from two arbitrary
  // vertices); the same direction gives a first lower bound on the distance for free
  v3 d0 = sa.c - sb.c;
  float dd = dot(d0, d0);
  if (!(dd >= 1e-12f)) { d0 = mk3(1.f, 0.f, 0.f); dd = 1.f; }
  const v3 a0 = gjk_support_wave(sa, -d0, has), b0 = gjk_support_wave(sb, d0, has);
  v3 v = a0 - b0;
  float vv = dot(v, v);
  s.p0.a = a0; s.p0.b = b0; s.p0.w = v; s.p1 = s.p0; s.p2 = s.p0; s.p3 = s.p0;
  s.n = 1; s.l0 = 1; s.l1 = 0; s.l2 = 0; s.l3 = 0;
  pa = a0; pb = b0;
  bool pen = false, far_out = false, active = has;
  float lb = 0.f;
  int my_iters = 0; (void)my_iters;
  { const float vw0 = dot(d0, v); if (has && vw0 > 0.f && vw0 * vw0 > far * far * dd) { far_out = true; lb = vw0 / sqrtf(dd); active = false; } }
  for (int it = 0; it < maxit; it++) {
    if (active && vv < 1e-12f) { pen = true; active = false; }   /* cores closer than 1 micron: treat as overlapping */
    if (!wave_any(active)) break;
    const v3 wa = gjk_support_wave(sa, -v, active), wb = gjk_support_wave(sb, v, active);
    if (active) {
      my_iters++;
      const v3 w = wa - wb;
      const float vw = dot(v, w);
      const bool e0 = s.p0.w.x == w.x && s.p0.w.y == w.y && s.p0.w.z == w.z;
      const bool e1 = s.n > 1 && s.p1.w.x == w.x && s.p1.w.y == w.y && s.p1.w.z == w.z;
      const bool e2 = s.n > 2 && s.p2.w.x == w.x && s.p2.w.y == w.y && s.p2.w.z == w.z;
      if (vw > 0.f && vw * vw > far * far * vv) { far_out = true; lb = vw / sqrtf(vv); active = false; }
      else if (vv - vw <= tol * vv || e0 || e1 || e2) active = false;
      else {
        gjk_pt np; np.w = w; np.a = wa; np.b = wb;
        if (s.n == 1) s.p1 = np; else if (s.n == 2) s.p2 = np; else s.p3 = np;
        s.n++;
        v3 vn;
        // An enclosed origin contradicts a separating plane: v.w > 0 proves that every point x of A - B has v.x >= v.w > 0.  A flat
        // tetrahedron (four nearly coplanar support points: a capsule's segment along a hull facet) can pass the four side tests
        // by rounding alone -- seen on the device, with its contracted multiply-adds, for a forearm lying along the mattress
        // edge, 4 cm clear of it.  Then the closest points found so far stand, as in the no-progress case below.
        if (gjk_solve(s, vn)) { pen = !(vw > 0.f); active = false; }
        else {
          const float vvn = dot(vn, vn);
          // no progress (a degenerate sub-simplex solve can even move away): keep the closest points found so far
          if (vvn >= vv) active = false;
          else {
            v = vn; vv = vvn;
            pa = s.l0 * s.p0.a; pb = s.l0 * s.p0.b;
            if (s.n > 1) { pa = pa + s.l1 * s.p1.a; pb = pb + s.l1 * s.p1.b; }
            if (s.n > 2) { pa = pa + s.l2 * s.p2.a; pb = pb + s.l2 * s.p2.b; }
          }
        }
      }
    }
  }
  AGX_TRACE_GJK(has, my_iters, sa.n, sb.n, sb.box, far_out, far_out ? lb : sqrtf(vv), far)
  if (pen) { dist = 0; return true; }
  dist = far_out ? lb : sqrtf(vv);
  return false;
}

// cores overlap: minimum over a fixed direction set of the separation needed along that direction
AGX_DEV void gjk_penetration(const gjk_shape& sa, const gjk_shape& sb, const float* dirs, int ndir, float& depth, v3& n, v3& pa, v3& pb) {
  float best = 3.0e38f; int bi = 0;
  for (int k = 0; k < ndir; k++) {
    v3 d = mk3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
    v3 a = gjk_support(sa, -d), b = gjk_support(sb, d);
    float dep = dot(b, d) - dot(a, d);
    if (dep < best) { best = dep; bi = k; }
  }
  n = mk3(dirs[3 * bi], dirs[3 * bi + 1], dirs[3 * bi + 2]);
  pa = gjk_support(sa, -n);
  pb = pa + best * n;
  depth = best;
}
