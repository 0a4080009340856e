// agx_gjk.h -- per-lane convex narrowphase: GJK distance between two (vertex core + radius)
// colliders, with a fixed-direction penetration sampler when the cores overlap.
// Each lane works on its own collider pair; vertices are read from the model blob (L1/L2
// resident, shared by all environments) in the body frame and transformed on the fly.
#pragma once

struct gjk_shape {
  const float* v;   // body-frame vertices (x,y,z)*n in the model blob
  int n;
  m3 R;             // body rotation (world)
  v3 p;             // body position minus the pair's shift point
  bool box;         // axis-aligned world box given by lo/hi (shifted frame) instead of vertices
  v3 lo, hi;
};

AGX_DEV v3 gjk_vertex0(const gjk_shape& s) {
  if (s.box) return s.lo;
  return mul(s.R, mk3(s.v[0], s.v[1], s.v[2])) + s.p;
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
  v3 dl = tmul(s.R, d);
  int best = 0;
  float bd = s.v[0] * dl.x + s.v[1] * dl.y + s.v[2] * dl.z;
  for (int k = 1; k < s.n; k++) {
    float t = s.v[3 * k] * dl.x + s.v[3 * k + 1] * dl.y + s.v[3 * k + 2] * dl.z;
    if (t > bd) { bd = t; best = k; }
  }
  return mul(s.R, mk3(s.v[3 * best], s.v[3 * best + 1], s.v[3 * best + 2])) + s.p;
}

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

struct gjk_simplex { v3 W[4], A[4], B[4]; float lam[4]; int n; };

// closest point of the simplex to the origin; compacts to the supporting sub-simplex.
// returns true if the origin is enclosed (tetrahedron case).
AGX_DEV bool gjk_solve(gjk_simplex& s, v3& v) {
  float l[4] = {0, 0, 0, 0};
  const int n = s.n;
  if (n == 1) { l[0] = 1; }
  else if (n == 2) {
    v3 d = s.W[1] - s.W[0];
    float dd = dot(d, d), t = dd > 0 ? -dot(s.W[0], d) / dd : 0.0f;
    if (t <= 0) l[0] = 1; else if (t >= 1) l[1] = 1; else { l[0] = 1 - t; l[1] = t; }
  } else if (n == 3) {
    gjk_closest_tri(s.W[0], s.W[1], s.W[2], l[0], l[1], l[2]);
  } else {
    const int faces[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
    float best = 3.0e38f; bool any = false;
    for (int f = 0; f < 4; f++) {
      v3 a = s.W[faces[f][0]], b = s.W[faces[f][1]], c = s.W[faces[f][2]], d = s.W[faces[f][3]];
      v3 nrm = cross(b - a, c - a);
      float sp = -dot(a, nrm), sd = dot(d - a, nrm);
      if (sp * sd > 0) continue;
      any = true;
      float wa, wb, wc; gjk_closest_tri(a, b, c, wa, wb, wc);
      v3 p = wa * a + wb * b + wc * c;
      float d2 = dot(p, p);
      if (d2 < best) { best = d2; l[0] = l[1] = l[2] = l[3] = 0; l[faces[f][0]] = wa; l[faces[f][1]] = wb; l[faces[f][2]] = wc; }
    }
    if (!any) return true;
  }
  int k2 = 0;
  for (int k = 0; k < n; k++) if (l[k] > 0) {
    if (k2 != k) { s.W[k2] = s.W[k]; s.A[k2] = s.A[k]; s.B[k2] = s.B[k]; }
    s.lam[k2] = l[k]; k2++;
  }
  s.n = k2;
  v = mk3(0, 0, 0);
  for (int k = 0; k < k2; k++) v = v + s.lam[k] * s.W[k];
  return false;
}

// returns true when the cores overlap; otherwise dist / witness points (shifted frame)
AGX_DEV bool gjk_distance(const gjk_shape& sa, const gjk_shape& sb, float tol, int maxit, float& dist, v3& pa, v3& pb) {
  gjk_simplex s;
  v3 a0 = gjk_vertex0(sa), b0 = gjk_vertex0(sb);
  v3 v = a0 - b0;
  float vv = dot(v, v);
  s.A[0] = a0; s.B[0] = b0; s.W[0] = v; s.n = 1; s.lam[0] = 1; s.lam[1] = s.lam[2] = s.lam[3] = 0;
  bool pen = false;
  for (int it = 0; it < maxit; it++) {
    if (vv < 1e-12f) { pen = true; break; }   /* cores closer than 1 micron: treat as overlapping */
    v3 wa = gjk_support(sa, -v), wb = gjk_support(sb, v), w = wa - wb;
    float vw = dot(v, w);
    if (vv - vw <= tol * vv) break;
    bool dup = false;
    for (int k = 0; k < s.n; k++) if (s.W[k].x == w.x && s.W[k].y == w.y && s.W[k].z == w.z) dup = true;
    if (dup) break;
    s.W[s.n] = w; s.A[s.n] = wa; s.B[s.n] = wb; s.n++;
    v3 vn;
    if (gjk_solve(s, vn)) { pen = true; break; }
    float vvn = dot(vn, vn);
    v = vn;
    if (vvn >= vv) { vv = vvn; break; }
    vv = vvn;
  }
  if (pen) { dist = 0; return true; }
  pa = mk3(0, 0, 0); pb = mk3(0, 0, 0);
  for (int k = 0; k < s.n; k++) { pa = pa + s.lam[k] * s.A[k]; pb = pb + s.lam[k] * s.B[k]; }
  dist = sqrtf(vv);
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
