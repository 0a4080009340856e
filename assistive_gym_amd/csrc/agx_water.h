// agx_water.h -- the water of DrinkingEnv (drinking.py:160-172: 64 spheres of 5 mm radius, 1 g each, in the cup): ONE WAVEFRONT per
// environment, lane = particle.  Follows oracle/agx_oracle.c, water_substep, step for step (same order of operations, float32 here):
// position-based, one-way coupled -- the water sees the cup, the robot and the person where each internal substep STARTS (the build
// kernel leaves the world frames of the moving links and the free bodies of every substep in the per-environment trace), the rigid
// bodies do not feel the water (64 g against a cup held by a 500 N constraint).  [deviation: Bullet solves the spheres as rigid bodies
// inside its sequential-impulse solve, with rolling and friction]
//
// Per substep: (a) frames and shape boxes into LDS; (b) per particle: gravity, the candidate half spaces -- at most CONTACTS, in shape
// order, of the shapes within 2 r + |v| dt of where the substep starts, each the face (or tangent plane) the particle is in front of
// THERE --, prediction; (c) PITER iterations of [a Jacobi particle-particle pass over ascending neighbours, from the positions the
// pass starts with | the half-space projections in candidate order]; (d) v = (x - q) / dt (1 - kDP), tangential damping where a shape
// was touched.  The rigid substeps of an env step all run first, then this kernel replays their poses (as the garment's kernel does).
#pragma once

namespace agxw {
constexpr int CONTACTS = 12;                 // WATER_CONTACTS of the oracle
constexpr int MAX_BODIES = 64, MAX_SHAPES = 192, MAX_PARTICLES = 64;
// LDS (floats): body frames [MAX_BODIES][p(3), R(9)], shape boxes [MAX_SHAPES][lo(3), hi(3)], particle positions [64][3], the shape table
// [MAX_SHAPES][SHAPE_WORDS] (filled once per launch: what shape_distance and the boxes need of a shape's cloth-section and collider records --
// read from the blob per particle and substep these were chains of four dependent L2 loads per shape, 17 ms per 4096-environment step), and
// the list of the shapes whose box meets the water's (per substep, in shape order).  A particle's candidate half spaces live in registers.
constexpr int SHAPE_WORDS = 14;              // body slot, plane count, first plane, radius, vertex offset, vertex count, min(kDF x friction, 1), only-gender | human << 8, AABB centre (3), half extents (3)
constexpr int L_BODY = 0, L_BOX = L_BODY + 12 * MAX_BODIES, L_X = L_BOX + 6 * MAX_SHAPES, L_SHAPE = L_X + 3 * MAX_PARTICLES, L_LIST = L_SHAPE + SHAPE_WORDS * MAX_SHAPES;
constexpr int LDS_WORDS = L_LIST + MAX_SHAPES;
struct P4 { float x, y, z, w; };            // a face plane: unit normal, offset
constexpr int TRACE_BODY_WORDS = 12;         // per body and substep: p(3), R(9) row major
constexpr int REPORT_WORDS = MAX_PARTICLES;  // per particle: 1 = touched a shape of the person in the last internal substep (drinking.py:84-88)

// body slots: moving link d -> d, robot base -> ndof, human body h -> ndof + 1 + h, world -> ndof + 1 + nhuman, free body b -> ndof + 2 + nhuman + b
AGX_DEV int body_slot(int code, int ndof, int nhuman) {
  if (code == AGX_BODY_WORLD) return ndof + 1 + nhuman;
  if (code >= AGX_BODY_HUMAN0) return ndof + 1 + (code - AGX_BODY_HUMAN0);
  if (code >= AGX_BODY_FREE0) return ndof + 2 + nhuman + (code - AGX_BODY_FREE0);
  if (code == AGX_BODY_ROBOT_BASE) return ndof;
  return code;
}
AGX_DEV void quat_to_rows(const float* q, float* R) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
// robot base, the person's static bodies, the world and (from the state record: where the env step ENDS) the free bodies
AGX_DEV void static_frames(const uint32_t* blob, const float* gstate, float* body, int lane, bool with_free) {
  const int* bi = (const int*)blob;
  const int ndof = bi[AGX_H_NDOF], nhuman = bi[AGX_H_NHUMAN], nfree = bi[AGX_H_NFREE];
  for (int k = lane; k < 2 + nhuman + (with_free ? nfree : 0); k += 64) {
    float* B = body + 12 * (ndof + k);
    if (k == 1 + nhuman) { for (int t = 0; t < 12; t++) B[t] = (t == 3 || t == 7 || t == 11) ? 1.f : 0.f; continue; }
    const float* r = k == 0 ? gstate + bi[AGX_H_S_BASE] : k <= nhuman ? gstate + bi[AGX_H_S_HUMAN] + 7 * (k - 1) : gstate + bi[AGX_H_S_FREE] + 13 * (k - 2 - nhuman);
    B[0] = r[0]; B[1] = r[1]; B[2] = r[2]; quat_to_rows(r + 3, B + 3);
  }
}
// the shape table (lane = shape): everything the kernel reads of a shape per substep, once per launch
AGX_DEV void shape_table(const uint32_t* blob, float* table, float kDF, int lane) {
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  const int* cl = bi + bi[AGX_H_OFF_CLOTH];
  const int NS = cl[AGX_CL_NSHAPE], ndof = bi[AGX_H_NDOF], nhuman = bi[AGX_H_NHUMAN];
  for (int sh = lane; sh < NS; sh += 64) {
    const int* rec = cl + cl[AGX_CL_OFF_SHAPE] + 4 * sh; const int c = rec[0];
    const int* ci = bi + bi[AGX_H_OFF_COLL] + c * AGX_C_STRIDE; const float* cf = bf + bi[AGX_H_OFF_COLL] + c * AGX_C_STRIDE;
    float* o = table + SHAPE_WORDS * sh; int* oi = (int*)o;
    oi[0] = body_slot(ci[AGX_C_BODY], ndof, nhuman); oi[1] = rec[2]; oi[2] = rec[1]; o[3] = cf[AGX_C_RADIUS]; oi[4] = ci[AGX_C_VOFF]; oi[5] = ci[AGX_C_NVERT];
    const float fr = kDF * cf[AGX_C_FRICTION]; o[6] = fr < 1.f ? fr : 1.f;
    oi[7] = rec[3] | ((ci[AGX_C_TAG] == AGX_TAG_HUMAN ? 1 : 0) << 8);
    for (int k = 0; k < 3; k++) { o[8 + k] = cf[AGX_C_AABB_C + k]; o[11 + k] = cf[AGX_C_AABB_H + k]; }
  }
}
// signed distance of world point x to the surface of cloth shape `sh` (negative inside) and the outward normal nw (world frame): capsule /
// sphere cores exactly, hulls through their face planes (the largest plane distance: exact inside and in front of a face).  `sh` is the same
// in every lane (wave-uniform loop of the caller): the shape's record comes from the LDS table as broadcast reads, its planes / core vertices
// from the blob at a wave-uniform address (scalar loads).
AGX_DEV float shape_distance(const uint32_t* blob, const float* body, const float* table, int sh, const float* x, float* nw) {
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  const float* clf = bf + bi[AGX_H_OFF_CLOTH]; const int* cl = bi + bi[AGX_H_OFF_CLOTH];
  const float* rec = table + SHAPE_WORDS * sh; const int* reci = (const int*)rec;
  const int np = wave_uniform(reci[1]), p0 = wave_uniform(reci[2]);
  const float* B = body + 12 * wave_uniform(reci[0]); const float* R = B + 3;
  const float d0 = x[0] - B[0], d1 = x[1] - B[1], d2 = x[2] - B[2];
  const float xl0 = R[0] * d0 + R[3] * d1 + R[6] * d2, xl1 = R[1] * d0 + R[4] * d1 + R[7] * d2, xl2 = R[2] * d0 + R[5] * d1 + R[8] * d2;
  const float rad = rec[3];
  float n0, n1, n2, dist;
  if (np == 0) {
    const float* v = bf + bi[AGX_H_OFF_VERT] + 3 * wave_uniform(reci[4]);
    float c0 = v[0], c1 = v[1], c2 = v[2];
    if (wave_uniform(reci[5]) == 2) {
      const float ab0 = v[3] - v[0], ab1 = v[4] - v[1], ab2 = v[5] - v[2], l2 = ab0 * ab0 + ab1 * ab1 + ab2 * ab2;
      float t = l2 > 0.f ? ((xl0 - v[0]) * ab0 + (xl1 - v[1]) * ab1 + (xl2 - v[2]) * ab2) / l2 : 0.f; t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
      c0 += t * ab0; c1 += t * ab1; c2 += t * ab2;
    }
    n0 = xl0 - c0; n1 = xl1 - c1; n2 = xl2 - c2; const float len = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
    if (len > 1e-12f) { n0 /= len; n1 /= len; n2 /= len; } else { n0 = 0.f; n1 = 0.f; n2 = 1.f; }
    dist = len - rad;
  } else {
    // Plane lists are padded to a multiple of four and start at a multiple of four (model/cloth.py): FOUR planes = 64 bytes per scalar load
    // at a wave-uniform address, the next four requested before these are evaluated.  One plane per load and iteration left the loop waiting
    // for a scalar-cache round trip per plane: ~23 planes x 68 cup pieces x 20 substeps = most of the kernel's 10.6 ms (round 4, second pass).
    const P4* P = (const P4*)(clf + cl[AGX_CL_OFF_PLANE] + 4 * p0);
    float bd = -3.0e38f; n0 = 0.f; n1 = 0.f; n2 = 1.f;
    P4 a0 = P[0], a1 = P[1], a2 = P[2], a3 = P[3];
    for (int k = 0; k < np; k += 4) {
      const P4 c0 = a0, c1 = a1, c2 = a2, c3 = a3;
      if (k + 4 < np) { a0 = P[k + 4]; a1 = P[k + 5]; a2 = P[k + 6]; a3 = P[k + 7]; }
      const float t0 = c0.x * xl0 + c0.y * xl1 + c0.z * xl2 - c0.w, t1 = c1.x * xl0 + c1.y * xl1 + c1.z * xl2 - c1.w;
      const float t2 = c2.x * xl0 + c2.y * xl1 + c2.z * xl2 - c2.w, t3 = c3.x * xl0 + c3.y * xl1 + c3.z * xl2 - c3.w;
      if (t0 > bd) { bd = t0; n0 = c0.x; n1 = c0.y; n2 = c0.z; }
      if (t1 > bd) { bd = t1; n0 = c1.x; n1 = c1.y; n2 = c1.z; }
      if (t2 > bd) { bd = t2; n0 = c2.x; n1 = c2.y; n2 = c2.z; }
      if (t3 > bd) { bd = t3; n0 = c3.x; n1 = c3.y; n2 = c3.z; }
    }
    dist = bd - rad;
  }
  nw[0] = R[0] * n0 + R[1] * n1 + R[2] * n2; nw[1] = R[3] * n0 + R[4] * n1 + R[5] * n2; nw[2] = R[6] * n0 + R[7] * n1 + R[8] * n2;
  return dist;
}

// one env step of the water: `nsub` internal substeps, substep k reading the frames of trace slot k ([NDOF + NFREE][12]).
// gwater: float[2][NN][3] positions then velocities (in/out); greport: REPORT_WORDS ints, written after the last substep
AGX_DEV void water_env(const uint32_t* blob, const float* gstate, const float* gtrace, float* gwater, float* greport, int nsub, float* lds, int lane) {
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  const int* cl = bi + bi[AGX_H_OFF_CLOTH]; const float* clf = bf + bi[AGX_H_OFF_CLOTH];
  const int NN = cl[AGX_CL_NN], NS = cl[AGX_CL_NSHAPE];
  const int ndof = bi[AGX_H_NDOF], nhuman = bi[AGX_H_NHUMAN], nfree = bi[AGX_H_NFREE], S_ = bi[AGX_H_SIM_SUBSTEPS] > 1 ? bi[AGX_H_SIM_SUBSTEPS] : 1;
  const float* par = clf + cl[AGX_CL_OFF_PARAM];
  const float dt = bf[bi[AGX_H_OFF_PARAMS] + AGX_P_DT] / (float)S_, grav = bf[bi[AGX_H_OFF_PARAMS] + AGX_P_GRAVITY_Z];
  const float r = par[AGX_CP_MARGIN], kDP = par[AGX_CP_KDP], kDF = par[AGX_CP_KDF]; const int piter = (int)par[AGX_CP_PITER];
  const int gender = ((const int*)gstate)[bi[AGX_H_S_ENV] + AGX_E_GENDER];
  float* body = lds + L_BODY; float* box = lds + L_BOX; float* X = lds + L_X; float* table = lds + L_SHAPE; int* list = (int*)(lds + L_LIST);
  const bool mine = lane < NN;
  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f}, q[3];
  if (mine) for (int k = 0; k < 3; k++) { x[k] = gwater[3 * lane + k]; v[k] = gwater[3 * NN + 3 * lane + k]; }
  static_frames(blob, gstate, body, lane, false);
  shape_table(blob, table, kDF, lane);
  // candidate half spaces of this lane's particle: normal (3), offset, shape | touched << 16 -- registers, walked by unrolled predicated loops
  float cn0[CONTACTS], cn1[CONTACTS], cn2[CONTACTS], cof[CONTACTS]; int csh[CONTACTS];
  int human_hit = 0;
  for (int sub = 0; sub < nsub; sub++) {
    // (a) frames of the moving links and the free bodies at the start of this substep; world boxes of the shapes
    const float* tr = gtrace + (size_t)sub * (ndof + nfree) * TRACE_BODY_WORDS;
    wave_sync();
    for (int k = lane; k < 12 * ndof; k += 64) body[k] = tr[k];
    for (int k = lane; k < 12 * nfree; k += 64) body[12 * (ndof + 2 + nhuman) + k] = tr[12 * ndof + k];
    wave_sync();
    for (int sh = lane; sh < NS; sh += 64) {
      const float* rec = table + SHAPE_WORDS * sh;
      const float* B = body + 12 * ((const int*)rec)[0]; const float* R = B + 3;
      const float g = rec[3] + 1e-6f;
      for (int k = 0; k < 3; k++) {
        const float cw = B[k] + R[3 * k] * rec[8] + R[3 * k + 1] * rec[9] + R[3 * k + 2] * rec[10];
        const float h = fabsf(R[3 * k]) * rec[11] + fabsf(R[3 * k + 1]) * rec[12] + fabsf(R[3 * k + 2]) * rec[13] + g;
        box[6 * sh + k] = cw - h; box[6 * sh + 3 + k] = cw + h;
      }
    }
    // (b) gravity; the shapes whose box meets the box of the water (each particle grown by its own reach) in shape order; a particle's
    // candidates where the substep starts; prediction
    float reach = 0.f;
    if (mine) {
      for (int k = 0; k < 3; k++) q[k] = x[k];
      v[2] += grav * dt;
      reach = 2 * r + sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * dt;
    }
    // (a drunk particle waits thousands of metres away, drinking.py:70: it meets no shape's box -- the ground box ends at 15 m -- and stays out of the water's)
    const bool here = mine && fabsf(q[0]) < 500.f && fabsf(q[1]) < 500.f && fabsf(q[2]) < 500.f;
    float wlo[3], whi[3];
    for (int k = 0; k < 3; k++) { wlo[k] = wave_min(here ? q[k] - reach : 3.0e38f); whi[k] = wave_max(here ? q[k] + reach : -3.0e38f); }
    wave_sync();
    int nlist = 0;
    for (int base = 0; base < NS; base += 64) {
      const int sh = base + lane; bool ok = sh < NS;
      if (ok) { const int only = ((const int*)table)[SHAPE_WORDS * sh + 7] & 0xff; if (only && only != gender + 1) ok = false; }
      if (ok) { const float* b6 = box + 6 * sh; for (int k = 0; k < 3; k++) if (b6[k] > whi[k] || b6[3 + k] < wlo[k]) ok = false; }
      const uint64_t m = wave_ballot(ok);
      if (ok) list[nlist + wave_rank(m)] = sh;
      nlist += popc64(m);
    }
    wave_sync();
    int ncand = 0;
    for (int e = 0; e < nlist; e++) {                       // wave-uniform loop; a lane takes part while it has room and its particle is within reach of the shape's box
      const int sh = wave_uniform(list[e]);
      const float* b6 = box + 6 * sh;
      const bool near = here && ncand < CONTACTS && !(q[0] < b6[0] - reach || q[0] > b6[3] + reach || q[1] < b6[1] - reach || q[1] > b6[4] + reach || q[2] < b6[2] - reach || q[2] > b6[5] + reach);
      if (!wave_any(near)) continue;
      float nw[3]; const float d = shape_distance(blob, body, table, sh, q, nw);
      if (near && d < reach) {
        const float off = (nw[0] * q[0] + nw[1] * q[1] + nw[2] * q[2]) - d;
#pragma unroll
        for (int cc = 0; cc < CONTACTS; cc++) if (cc == ncand) { cn0[cc] = nw[0]; cn1[cc] = nw[1]; cn2[cc] = nw[2]; cof[cc] = off; csh[cc] = sh; }
        ncand++;
      }
    }
    if (mine) for (int k = 0; k < 3; k++) x[k] = q[k] + v[k] * dt;
    // (c) projection iterations
    for (int it = 0; it < piter; it++) {
      wave_sync();
      if (mine) { X[3 * lane] = x[0]; X[3 * lane + 1] = x[1]; X[3 * lane + 2] = x[2]; }
      wave_sync();
      if (mine) {
        float dx[3] = {0.f, 0.f, 0.f}; int cnt = 0;
        // branch-free over the 64 neighbours (selects instead of divergent skips; the positions come as LDS broadcasts): this loop is 640 passes
        // per substep and was 40 % of the kernel's vector instructions with a branch, an IEEE square root and a division in each
        const float rr4 = 4 * r * r;
        for (int j = 0; j < NN; j++) {
          const float e0 = x[0] - X[3 * j], e1 = x[1] - X[3 * j + 1], e2 = x[2] - X[3 * j + 2], d2 = e0 * e0 + e1 * e1 + e2 * e2;
          const bool in = d2 < rr4 && j != lane;
          const bool apart = d2 > 1.1920929e-7f * 1.1920929e-7f;
          // 0.5 (2 r - d) / d = r / d - 0.5 with 1 / d from the hardware reciprocal square root (1 ulp; the oracle's arithmetic is float64 anyway)
          const float sc = (in && apart) ? r * wave_rsqrt(d2) - 0.5f : 0.f;
          dx[0] += e0 * sc; dx[1] += e1 * sc; dx[2] += e2 * sc;
          dx[2] += (in && !apart) ? (lane > j ? r : -r) : 0.f;   // coincident centres: apart along z, the higher index up
          cnt += in ? 1 : 0;
        }
        if (cnt > 1) for (int k = 0; k < 3; k++) dx[k] /= (float)cnt;
        for (int k = 0; k < 3; k++) x[k] += dx[k];
#pragma unroll
        for (int cc = 0; cc < CONTACTS; cc++) if (cc < ncand) {
          const float d = (cn0[cc] * x[0] + cn1[cc] * x[1] + cn2[cc] * x[2]) - cof[cc] - r;
          if (d < 0.f) { x[0] -= cn0[cc] * d; x[1] -= cn1[cc] * d; x[2] -= cn2[cc] * d; csh[cc] |= 1 << 16; }
        }
      }
    }
    // (d) velocities; a particle that touched a shape loses the share kDF x friction of its tangential velocity
    human_hit = 0;
    if (mine) {
      for (int k = 0; k < 3; k++) v[k] = (x[k] - q[k]) / dt * (1 - kDP);
#pragma unroll
      for (int cc = 0; cc < CONTACTS; cc++) if (cc < ncand && (csh[cc] >> 16)) {
        const float* rec = table + SHAPE_WORDS * (csh[cc] & 0xffff);
        const float fc = rec[6];
        const float vn = v[0] * cn0[cc] + v[1] * cn1[cc] + v[2] * cn2[cc];
        v[0] -= (v[0] - cn0[cc] * vn) * fc; v[1] -= (v[1] - cn1[cc] * vn) * fc; v[2] -= (v[2] - cn2[cc] * vn) * fc;
        if (((const int*)rec)[7] >> 8) human_hit = 1;
      }
    }
  }
  if (mine) for (int k = 0; k < 3; k++) { gwater[3 * lane + k] = x[k]; gwater[3 * NN + 3 * lane + k] = v[k]; }
  if (greport && lane < REPORT_WORDS) ((int*)greport)[lane] = mine ? human_hit : 0;
}
}  // namespace agxw
