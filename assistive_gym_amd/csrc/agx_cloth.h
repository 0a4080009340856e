// agx_cloth.h -- the garment of DressingEnv (dressing.py:149-157,184): one WORKGROUP of AGX_CLOTH_THREADS threads per environment,
// node positions x and start-of-substep positions q of all 3,966 nodes resident in LDS (95 KB: one workgroup per CU) for ALL
// internal substeps of an env step; the velocities are implicit between substeps (v = (x - q)(1 - kDP) / dt).
//
// Coupling is one way (see oracle/agx_oracle.c, cloth_substep, which this file follows step for step): the cloth sees the rigid
// bodies where each substep starts -- the build kernel leaves the world frames of the moving links of every substep in the
// per-environment trace -- and the rigid bodies do not feel the cloth.  The rigid substeps of an env step therefore all run
// first, then this kernel replays their poses: x and q are read from and written to HBM once per env step (95 KB each way)
// instead of once per substep.
//
// Per substep: (a) body frames, shape boxes (world AABB grown by the margin) and the attachment point into LDS; the shapes whose
// box meets the cloth's bounding box form the candidate list, in shape order; (b) per node: normal from the incident faces,
// gravity, clamped aerodynamic drag, q = x, x += v dt; (c) per node: contacts with the candidate shapes (capsule / sphere cores
// exactly, hulls through their face planes), at most AGX_CLOTH_NODE_CONTACTS per node, kept in registers -- a node's contacts
// only move that node; (d) piterations x [anchors | rigid contacts, the links inside each wave's patch (256 neighbouring nodes: 8 of 9
// links) colour by colour without a workgroup barrier | the links between patches, one workgroup-wide colour class at a time].
// A wave owns the nodes of its patch (AGX_CL_OFF_PERM, Morton order), lane l its l-th, (64 + l)-th, ...
#pragma once
#include "../../include/agx_blob.h"

namespace agxc {

constexpr int T = AGX_CLOTH_THREADS;
constexpr int NPT = 4096 / T;              // nodes per thread (garments of up to 4,096 nodes)
constexpr int LPT = 1024 / T;              // links per thread and colour class (classes hold at most 1,024 links)
#ifndef AGXC_PF
#define AGXC_PF 1
#endif
constexpr int PF = AGXC_PF;                // colour classes of link records in flight (see PSolve_Links)
constexpr int NODE_CONTACTS = 2;          // AGX_CLOTH_NODE_CONTACTS: contacts kept per node (the first ones in shape order)
constexpr int MAX_BODIES = 64, MAX_SHAPES = 192;
constexpr float EPS = 1.1920929e-7f;      // SIMD_EPSILON

struct f3 { float x, y, z; };
__device__ inline f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ inline f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ inline f3 ld(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ inline void st(float* p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
// Node positions in LDS: NS4 words per node.  3 (the default): x, y, z packed, three ds_read_b32 / ds_write_b32 per node.  -DAGXC_NODE_STRIDE=4 (round 6, A/B):
// padded to 16 bytes, ONE ds_read_b128 and a ds_write_b64 + ds_write_b32 per node -- fewer LDS instructions, and by the guide's table fewer LDS cycles per
// link (28 against 36 + conflicts; the link tables of the garment give ~2.1-way conflicts on the packed layout, measured on the blob).  MEASURED, same box,
// DressingBaxter 4096 x 30 steps: padded 50.4 / 51.6 k env-steps/s against packed 54.1 k (profiles/r06/r06d_ab_cloth_node_stride.txt) -- slower: the kernel
// is not bound by LDS array cycles (as the counters of round 5 already said: 63 % of the wave cycles at s_waitcnt, i.e. latency between 17 barriers per
// solver iteration).  Results are bit-identical either way (tests/test_gpu_dressing.py passes with both).  Not kept as the default.
#ifndef AGXC_NODE_STRIDE
#define AGXC_NODE_STRIDE 3
#endif
constexpr int NS4 = AGXC_NODE_STRIDE;
__device__ inline f3 ldn(const float* p) {
  if constexpr (NS4 == 4) { const float4 v = *(const float4*)p; return mk(v.x, v.y, v.z); } else return mk(p[0], p[1], p[2]);
}
__device__ inline void stn(float* p, f3 a) {
  if constexpr (NS4 == 4) { *(float2*)p = make_float2(a.x, a.y); p[2] = a.z; } else { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
}
// R (row major, 9 floats) times v, and R^T times v
__device__ inline f3 rot(const float* R, f3 v) { return mk(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z); }
__device__ inline f3 rot_t(const float* R, f3 v) { return mk(R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z); }
__device__ inline void quat_to_mat(const float* q, float* R) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// Workgroup barrier for phases that exchange data through LDS only.  __syncthreads() is a workgroup-scope fence + s_barrier: the fence makes
// every wave wait for ALL its outstanding memory operations -- including the link records of the NEXT colour class, which are requested one
// class ahead precisely so that their L2 latency overlaps this class's LDS work.  With 17 barriers per solver iteration and 200 iterations
// per env step that wait (a full global-memory round trip per phase) was most of the kernel.  Here: wait for this wave's LDS operations only
// (lgkmcnt), then the barrier; global loads stay in flight (the link table is read-only, and the contact records in the HBM scratch are only
// ever touched by the thread that owns the node).  -DAGXC_FULL_BARRIERS restores __syncthreads() for same-box A/B runs.
__device__ inline void lds_barrier() {
#ifdef AGXC_FULL_BARRIERS
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// LDS layout (floats)
struct Lds {
  float* x; float* q;            // [NN][NS4] each (x, y, z, pad)
  float* body;                   // [MAX_BODIES][12]: p(3), R(9); index: moving link d, ndof = robot base, ndof + 1 + h = human body h, last = world
  float* box;                    // [MAX_SHAPES][6] world AABB of the shape grown by the margin
  int* cand; int* ncand;         // candidate shapes of this substep, in shape order
  float* red;                    // [16 waves][6] bounding-box reduction
  float* anchor;                 // [3]
  float* shape;                  // [MAX_SHAPES][12]: body slot (int), face planes (int count, int first), radius, core vertex 0 (3), core vertex 1 (3), kDF x friction, unused
};
constexpr int SHAPE_WORDS = 12;
constexpr int lds_words(int nn) { return 2 * NS4 * nn + 12 * MAX_BODIES + 6 * MAX_SHAPES + MAX_SHAPES + 4 + 6 * (T / 64) + 4 + SHAPE_WORDS * MAX_SHAPES; }

__device__ inline int body_slot(int code, int ndof, int nhuman) {
  if (code == AGX_BODY_WORLD) return ndof + 1 + nhuman;
  if (code >= AGX_BODY_HUMAN0) return ndof + 1 + (code - AGX_BODY_HUMAN0);
  if (code == AGX_BODY_ROBOT_BASE) return ndof;
  return code;                   // moving link (the dressing scene has no free bodies)
}

// signed distance of world point x to the surface of shape `sh` (negative inside) and the outward normal, world frame; the shape's
// record comes from the LDS table filled once per launch, only the face planes of hulls are read from the blob.  `sh` is the same in
// every lane of the wave (the candidate loop is workgroup-uniform): plane count and first plane go to scalar registers, so that the
// planes arrive through the scalar cache (s_load) instead of as 64-lane vector loads of one address each -- the texture path, not the
// arithmetic, was what the contact phase waited for.
__device__ inline float shape_distance(const float* clf, const int* cl, const Lds& S, int sh, f3 x, f3& nw) {
  const float* rec = S.shape + SHAPE_WORDS * sh; const int* reci = (const int*)rec;
  const float* B = S.body + 12 * reci[0];
  const f3 xl = rot_t(B + 3, x - ld(B));
  const int np = __builtin_amdgcn_readfirstlane(reci[1]); const float rad = rec[3];
  f3 nl; float dist;
  if (np == 0) {
    const f3 a = ld(rec + 4), ab = ld(rec + 7) - a, ax = xl - a; const float l2 = dot(ab, ab);
    float t = l2 > 0.f ? dot(ax, ab) / l2 : 0.f; t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    nl = xl - (a + t * ab); const float len = sqrtf(dot(nl, nl));
    if (len > 1e-12f) nl = (1.0f / len) * nl; else nl = mk(0.f, 0.f, 1.f);
    dist = len - rad;
  } else {
    const float4* P = (const float4*)(clf + cl[AGX_CL_OFF_PLANE]) + __builtin_amdgcn_readfirstlane(reci[2]);   // 16-byte aligned by the model compiler
    float bd = -3.0e38f; nl = mk(0.f, 0.f, 1.f);
    for (int k = 0; k < np; k += 4) {          // plane lists are padded to a multiple of four (model/cloth.py): one scalar load of 64 bytes
      const float4 p0 = P[k], p1 = P[k + 1], p2 = P[k + 2], p3 = P[k + 3];
      const float t0 = p0.x * xl.x + p0.y * xl.y + p0.z * xl.z - p0.w, t1 = p1.x * xl.x + p1.y * xl.y + p1.z * xl.z - p1.w;
      const float t2 = p2.x * xl.x + p2.y * xl.y + p2.z * xl.z - p2.w, t3 = p3.x * xl.x + p3.y * xl.y + p3.z * xl.z - p3.w;
      if (t0 > bd) { bd = t0; nl = mk(p0.x, p0.y, p0.z); }
      if (t1 > bd) { bd = t1; nl = mk(p1.x, p1.y, p1.z); }
      if (t2 > bd) { bd = t2; nl = mk(p2.x, p2.y, p2.z); }
      if (t3 > bd) { bd = t3; nl = mk(p3.x, p3.y, p3.z); }
    }
    dist = bd - rad;
  }
  nw = rot(B + 3, nl);
  return dist;
}

// A node-vs-rigid contact of the current substep lives in the environment's contact scratch in HBM (behind its report, agx_blob.h
// AGX_CLOTH_SCRATCH_WORDS), slot (node, k): normal (3), offset, c3 (friction factor), impulse sum (3; last substep only).  Only the
// thread that owns the node touches the slot.  In registers the eight records of a thread cost 48 VGPRs that the compiler could not
// find (121 spilled registers, spill traffic in every loop of the kernel); the thread keeps the contact COUNT of its nodes only.
constexpr int CREC = 8;

// one env step of the garment: `nsub` internal substeps, substep k reading the link frames of trace slot k.
// gcloth: float[2][NN][3] positions then velocities (in/out); greport: see agx_blob.h AGX_CLOTH_REPORT (written after the last substep)
__device__ inline void cloth_env(const uint32_t* blob, const float* gstate, const float* gtrace, float* gcloth, float* greport, int nsub, float* lds) {
  const int* bi = (const int*)blob; const float* bf = (const float*)blob;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int oc = bi[AGX_H_OFF_CLOTH]; const int* cl = bi + oc; const float* clf = bf + oc;
  const int NN = cl[AGX_CL_NN], NCOL = cl[AGX_CL_NCOLOR], NA = cl[AGX_CL_NANCHOR], NS = cl[AGX_CL_NSHAPE], KP = cl[AGX_CL_NPATCH_COLOR];
  const int ndof = bi[AGX_H_NDOF], nhuman = bi[AGX_H_NHUMAN], S_ = bi[AGX_H_SIM_SUBSTEPS] > 1 ? bi[AGX_H_SIM_SUBSTEPS] : 1;
  const float* par = clf + cl[AGX_CL_OFF_PARAM];
  const float dt = bf[bi[AGX_H_OFF_PARAMS] + AGX_P_DT] / (float)S_;
  const float kLST = par[AGX_CP_KLST], kDP = par[AGX_CP_KDP], kDG = par[AGX_CP_KDG], kDF = par[AGX_CP_KDF], kCHR = par[AGX_CP_KCHR], kAHR = par[AGX_CP_KAHR];
  const float mrg = par[AGX_CP_MARGIN], im = par[AGX_CP_NODE_IM], rho = par[AGX_CP_AIR_DENSITY]; const int piter = (int)par[AGX_CP_PITER];
  const int s_env = bi[AGX_H_S_ENV], s_task = bi[AGX_H_S_TASK];
  const int gender = ((const int*)gstate)[s_env + AGX_E_GENDER];
  const float grav = gstate[s_task + AGX_DR_CLOTH_GRAVITY];
  Lds S; S.x = lds; S.q = S.x + NS4 * NN; S.body = S.q + NS4 * NN; S.box = S.body + 12 * MAX_BODIES; S.cand = (int*)(S.box + 6 * MAX_SHAPES);
  S.ncand = S.cand + MAX_SHAPES; S.red = (float*)(S.ncand + 4); S.anchor = S.red + 6 * (T / 64); S.shape = S.anchor + 4;
  const int* nodei = cl + cl[AGX_CL_OFF_NODE]; const float* nodef = clf + cl[AGX_CL_OFF_NODE];
  const int* face = cl + cl[AGX_CL_OFF_FACE];
  const int* anci = cl + cl[AGX_CL_OFF_ANCHOR]; const float* ancf = clf + cl[AGX_CL_OFF_ANCHOR];
  const int* color = cl + cl[AGX_CL_OFF_COLOR];
  // links: thread tid relaxes link number tid (+ T, ...) of every colour class.  The link table (93 KB, shared by all environments: L2)
  // is streamed, one class ahead of its use, instead of being held in registers (30 VGPRs that the contact phase needs).
  const int2* links = (const int2*)(cl + cl[AGX_CL_OFF_LINK]);
  // static frames: robot base, human bodies, world; load x, and q := x - v dt / (1 - kDP) so that the implicit velocity of the first substep is v
  if (tid == 0) {
    float* B = S.body + 12 * ndof; const float* r = gstate + bi[AGX_H_S_BASE]; st(B, ld(r)); quat_to_mat(r + 3, B + 3);
    float* W = S.body + 12 * (ndof + 1 + nhuman); st(W, mk(0.f, 0.f, 0.f)); W[3] = 1; W[4] = 0; W[5] = 0; W[6] = 0; W[7] = 1; W[8] = 0; W[9] = 0; W[10] = 0; W[11] = 1;
  }
  if (tid >= 64 && tid < 64 + nhuman) { const int h = tid - 64; float* B = S.body + 12 * (ndof + 1 + h); const float* r = gstate + bi[AGX_H_S_HUMAN] + 7 * h; st(B, ld(r)); quat_to_mat(r + 3, B + 3); }
  if (tid >= 128 && tid < 128 + NS) {     // shape table
    const int sh = tid - 128; const int* rec = cl + cl[AGX_CL_OFF_SHAPE] + 4 * sh; const int c = rec[0];
    const int* ci = bi + bi[AGX_H_OFF_COLL] + c * AGX_C_STRIDE; const float* cf = bf + bi[AGX_H_OFF_COLL] + c * AGX_C_STRIDE;
    float* o = S.shape + SHAPE_WORDS * sh; int* oi = (int*)o;
    oi[0] = body_slot(ci[AGX_C_BODY], ndof, nhuman); oi[1] = rec[2]; oi[2] = rec[1]; o[3] = cf[AGX_C_RADIUS];
    const float* v = bf + bi[AGX_H_OFF_VERT] + 3 * ci[AGX_C_VOFF];
    st(o + 4, ld(v)); st(o + 7, ci[AGX_C_NVERT] == 2 ? ld(v + 3) : ld(v)); o[10] = kDF * cf[AGX_C_FRICTION]; o[11] = 0.f;
  }
  const float vscale = dt / (1.0f - kDP);
  for (int k = tid; k < 3 * NN; k += T) { const float xv = gcloth[k]; const int kk = NS4 * (k / 3) + k % 3; S.x[kk] = xv; S.q[kk] = xv - gcloth[3 * NN + k] * vscale; }
  // ownership: a wave owns 64 NPT consecutive nodes of the Morton-ordered list (-1: none), lane l its l-th, (64 + l)-th, ...: the nodes
  // of a wave are one patch of the garment
  int own[NPT]; bool attached[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) { const int i = cl[cl[AGX_CL_OFF_PERM] + wave * (64 * NPT) + j * 64 + lane]; own[j] = i; attached[j] = false; for (int a = 0; a < NA; a++) if (anci[4 * a] == i) attached[j] = true; }
  int ncon[NPT];
  float* const grec = greport + AGX_CLOTH_REPORT_WORDS(NN);
  lds_barrier();
  for (int sub = 0; sub < nsub; sub++) {
    // (a) frames of the moving links at the start of this substep; attachment point at the start of the current stepSimulation call
    const float* tr = gtrace + (size_t)sub * ndof * 12;
    for (int k = tid; k < 12 * ndof; k += T) S.body[k] = tr[k];
    lds_barrier();
    if (sub % S_ == 0 && tid == 0) {     // end-effector frame origin = link frame * EE_POS (dressing.py:200-210)
      const int ot = bi[AGX_H_OFF_TASK]; const float* B = S.body + 12 * bi[ot + AGX_T_EE_LINK];
      st(S.anchor, ld(B) + rot(B + 3, ld(bf + ot + AGX_T_EE_POS)));
    }
    // cloth bounding box
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int j = 0; j < NPT; j++) { const int i = own[j]; if (i >= 0) for (int a = 0; a < 3; a++) { const float v = S.x[NS4 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); } }
    for (int a = 0; a < 3; a++) for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
    if (lane == 0) for (int a = 0; a < 3; a++) { S.red[6 * wave + a] = lo[a]; S.red[6 * wave + 3 + a] = hi[a]; }
    // shape boxes: world AABB of the collider (core box rotated + radius) grown by the margin
    if (tid < NS) {
      const int* rec = cl + cl[AGX_CL_OFF_SHAPE] + 4 * tid; const int c = rec[0];
      const float* cf = bf + bi[AGX_H_OFF_COLL] + c * AGX_C_STRIDE;
      const float* B = S.body + 12 * ((const int*)S.shape)[SHAPE_WORDS * tid];
      const f3 cw = ld(B) + rot(B + 3, ld(cf + AGX_C_AABB_C)); const f3 h = ld(cf + AGX_C_AABB_H); const float r = cf[AGX_C_RADIUS] + mrg + 1e-6f;
      const float* R = B + 3;
      const float hx = fabsf(R[0]) * h.x + fabsf(R[1]) * h.y + fabsf(R[2]) * h.z + r, hy = fabsf(R[3]) * h.x + fabsf(R[4]) * h.y + fabsf(R[5]) * h.z + r,
                  hz = fabsf(R[6]) * h.x + fabsf(R[7]) * h.y + fabsf(R[8]) * h.z + r;
      float* bx = S.box + 6 * tid; bx[0] = cw.x - hx; bx[1] = cw.y - hy; bx[2] = cw.z - hz; bx[3] = cw.x + hx; bx[4] = cw.y + hy; bx[5] = cw.z + hz;
    }
    lds_barrier();
    if (wave == 0) {                     // candidate list: shapes of this gender whose box meets the cloth's, in shape order
      float clo[3], chi[3];
      for (int a = 0; a < 3; a++) { clo[a] = S.red[a]; chi[a] = S.red[3 + a]; for (int w = 1; w < T / 64; w++) { clo[a] = fminf(clo[a], S.red[6 * w + a]); chi[a] = fmaxf(chi[a], S.red[6 * w + 3 + a]); } }
      int n = 0;
      for (int base = 0; base < NS; base += 64) {
        const int sh = base + lane; bool ok = sh < NS;
        if (ok) { const int only = cl[cl[AGX_CL_OFF_SHAPE] + 4 * sh + 3]; if (only && only != gender + 1) ok = false; }
        if (ok) { const float* bx = S.box + 6 * sh; for (int a = 0; a < 3; a++) if (bx[a] > chi[a] || bx[3 + a] < clo[a]) ok = false; }
        const unsigned long long m = __ballot(ok);
        if (ok) S.cand[n + __popcll(m & ((1ull << lane) - 1ull))] = sh;
        n += __popcll(m);
      }
      if (lane == 0) *S.ncand = n;
    }
    // (b) forces and prediction (normals read every node's position before any is moved: two passes with a barrier)
    f3 vnew[NPT];
#pragma unroll
    for (int j = 0; j < NPT; j++) {
      const int i = own[j]; vnew[j] = mk(0.f, 0.f, 0.f);
      if (i >= 0) {
        const f3 xi = ldn(S.x + NS4 * i);
        f3 nrm = mk(0.f, 0.f, 0.f);
#ifndef AGXC_NO_NORMALS
        for (int e = nodei[2 * i]; e < nodei[2 * i + 2]; e++) { const int fe = face[e]; nrm = nrm + cross(ldn(S.x + NS4 * (fe & 0xffff)) - xi, ldn(S.x + NS4 * ((fe >> 16) & 0xffff)) - xi); }
#endif
        const float nl = sqrtf(dot(nrm, nrm)); if (nl > EPS) nrm = (1.0f / nl) * nrm;
        f3 v = ((1.0f - kDP) / dt) * (xi - ldn(S.q + NS4 * i));
        v.z += grav * dt;
        const float v2 = dot(v, v);
        if (kDG > 0.f && v2 > EPS) {
          const float dvn = dot(v, nrm);
          if (dvn > 0.f) {
            const float fmag = nodef[2 * i + 1] * dvn * v2 * 0.5f * rho * kDG, dtim = dt * im;
            if (fmag * dtim * fmag * dtim > v2) v = mk(0.f, 0.f, 0.f); else v = v - (fmag * dtim / sqrtf(v2)) * v;
          }
        }
        vnew[j] = v;
      }
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < NPT; j++) { const int i = own[j]; if (i >= 0) { const f3 xi = ldn(S.x + NS4 * i); stn(S.q + NS4 * i, xi); stn(S.x + NS4 * i, xi + dt * vnew[j]); } }
    lds_barrier();
    // (c) contacts of this thread's nodes (CollideSDF_RS::DoNode)
#ifdef AGXC_NO_CONTACTS
    const int ncand = 0;
#else
    const int ncand = *S.ncand;
#endif
    // candidates outermost (their boxes are read once per thread, the next one while the current one is tested), this thread's nodes
    // innermost.  A shape whose box misses the box of the wave's 64 nodes of slot j (wave uniform, kept in scalar registers) is skipped
    // for all of them with one test: the nodes of a wave are neighbours on the garment (Morton ownership order).
    bool mine[NPT];
    float wlo[NPT][3], whi[NPT][3];
#pragma unroll
    for (int j = 0; j < NPT; j++) {
      const int i = own[j]; ncon[j] = 0; mine[j] = i >= 0 && !attached[j];
      const f3 xj = mine[j] ? ldn(S.x + NS4 * i) : mk(0.f, 0.f, 0.f);
      float l3[3] = {mine[j] ? xj.x : 3.0e38f, mine[j] ? xj.y : 3.0e38f, mine[j] ? xj.z : 3.0e38f}, h3[3] = {mine[j] ? xj.x : -3.0e38f, mine[j] ? xj.y : -3.0e38f, mine[j] ? xj.z : -3.0e38f};
      for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) { l3[a] = fminf(l3[a], __shfl_xor(l3[a], o)); h3[a] = fmaxf(h3[a], __shfl_xor(h3[a], o)); }
        wlo[j][a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, l3[a])));
        whi[j][a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h3[a])));
      }
    }
    float ulo[3], uhi[3];
    for (int a = 0; a < 3; a++) { ulo[a] = wlo[0][a]; uhi[a] = whi[0][a]; for (int j = 1; j < NPT; j++) { ulo[a] = fminf(ulo[a], wlo[j][a]); uhi[a] = fmaxf(uhi[a], whi[j][a]); } }
    int shn = ncand > 0 ? S.cand[0] : 0;
    float nb[6]; for (int a = 0; a < 6; a++) nb[a] = S.box[6 * shn + a];
    for (int k = 0; k < ncand; k++) {
      const int sh = __builtin_amdgcn_readfirstlane(shn); float bx[6]; for (int a = 0; a < 6; a++) bx[a] = nb[a];
      if (k + 1 < ncand) { shn = S.cand[k + 1]; for (int a = 0; a < 6; a++) nb[a] = S.box[6 * shn + a]; }
      if (ulo[0] > bx[3] || ulo[1] > bx[4] || ulo[2] > bx[5] || uhi[0] < bx[0] || uhi[1] < bx[1] || uhi[2] < bx[2]) continue;   // the whole patch misses the shape
#ifdef AGXC_NO_HULLS
      if (((const int*)S.shape)[SHAPE_WORDS * sh + 1] > 0) continue;
#endif
#pragma unroll
      for (int j = 0; j < NPT; j++) {
        if (wlo[j][0] > bx[3] || wlo[j][1] > bx[4] || wlo[j][2] > bx[5] || whi[j][0] < bx[0] || whi[j][1] < bx[1] || whi[j][2] < bx[2]) continue;   // wave uniform
        if (!mine[j] || ncon[j] >= NODE_CONTACTS) continue;
        const f3 xi = ldn(S.x + NS4 * own[j]);           // (positions do not move during this phase: re-read rather than kept in registers)
        if (xi.x < bx[0] || xi.y < bx[1] || xi.z < bx[2] || xi.x > bx[3] || xi.y > bx[4] || xi.z > bx[5]) continue;
#ifdef AGXC_NO_EVAL
        continue;
#endif
        f3 nw; const float dst = shape_distance(clf, cl, S, sh, xi, nw) - mrg;
        if (dst >= 0.f) continue;
        const f3 vr = xi - ldn(S.q + NS4 * own[j]); const float dn = dot(vr, nw); const f3 fv = vr - dn * nw;
        const float fc = S.shape[SHAPE_WORDS * sh + 10];
        float* rec = grec + CREC * (NODE_CONTACTS * own[j] + ncon[j]);
        *(float4*)rec = make_float4(nw.x, nw.y, nw.z, -dot(nw, xi) + dst);
        *(float4*)(rec + 4) = make_float4(dot(fv, fv) < (dn * fc * dn * fc) ? 0.f : 1.f - fc, 0.f, 0.f, 0.f);
        ncon[j]++;
      }
    }
    // (d) position solver
    for (int it = 0; it < piter; it++) {
      if (tid < NA) {                     // PSolve_Anchors
        const int i = anci[4 * tid]; const f3 wa = ld(S.anchor) + ld(ancf + 4 * tid + 1), xi = ldn(S.x + NS4 * i), qi = ldn(S.q + NS4 * i);
        stn(S.x + NS4 * i, xi + (-1.0f) * (xi - qi) + kAHR * (wa - xi));
      }
      lds_barrier();
#pragma unroll
      for (int j = 0; j < NPT; j++) {     // PSolve_RContacts
        const int i = own[j];
        if (i >= 0 && ncon[j] > 0) {
          f3 xi = ldn(S.x + NS4 * i); const f3 qi = ldn(S.q + NS4 * i);
#pragma unroll
          for (int cc = 0; cc < NODE_CONTACTS; cc++) if (cc < ncon[j]) {
            float* rec = grec + CREC * (NODE_CONTACTS * i + cc);
            const float4 r0 = *(const float4*)rec; const float c3 = rec[4];
            const f3 cn = mk(r0.x, r0.y, r0.z);
            const f3 vr = xi - qi; const float dn = dot(vr, cn);
            if (dn <= EPS) {
              float dp = dot(xi, cn) + r0.w; if (dp > mrg) dp = mrg;
              const f3 fv = vr - dn * cn, corr = vr - c3 * fv + (dp * kCHR) * cn;
              xi = xi - corr;
              if (sub == nsub - 1) { const float w = 1.0f / (dt * im); rec[5] += w * corr.x; rec[6] += w * corr.y; rec[7] += w * corr.z; }
            }
          }
          stn(S.x + NS4 * i, xi);
        }
      }
      // PSolve_Links.  (1) The links inside this wave's patch, colour by colour, no workgroup barrier: patches share no node, the LDS
      // executes a wave's accesses in order, and nothing but this wave has touched the patch since the anchors' barrier (a node's contacts
      // move that node only).  Link table: class w KP + c = 64 slots, lane l relaxes slot l; streamed one class ahead.
      {
        // link records are requested PF classes ahead of their use (a ring of PF registers, statically indexed: the loop is unrolled by PF).
        // Measured (round 4, same box, DressingBaxter 30 steps): PF = 1 with __syncthreads 53.04 k, PF = 1 with the LDS-only barrier 52.97 k,
        // PF = 3 with it 51.69 k env-steps/s -- neither the barriers' global-memory waits nor the prefetch distance is what the kernel waits for
        const int2* pl = links + (size_t)wave * KP * 64 + lane;
        int2 ring[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) ring[u] = u < KP ? pl[u * 64] : make_int2(-1, 0);
        for (int c0 = 0; c0 < KP; c0 += PF) {
#ifdef AGXC_NO_LINKS
          break;
#endif
#pragma unroll
          for (int u = 0; u < PF; u++) {
            const int c = c0 + u;
            if (c >= KP) break;
            const int2 cur = ring[u];
            ring[u] = c + PF < KP ? pl[(c + PF) * 64] : make_int2(-1, 0);
            if (cur.x >= 0) {
              const int a = cur.x & 0xffff, b = (cur.x >> 16) & 0xffff;
              const f3 xa = ldn(S.x + NS4 * a), xb = ldn(S.x + NS4 * b), del = xb - xa; const float len = dot(del, del), c1 = __int_as_float(cur.y);
              if (c1 + len > EPS) { const float k = (c1 - len) / (c1 + len) * kLST * 0.5f; stn(S.x + NS4 * a, xa - k * del); stn(S.x + NS4 * b, xb + k * del); }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next class reads what this one wrote (other lanes of this wave)
          }
        }
      }
      lds_barrier();
      // (2) the links between patches, one colour class at a time across the workgroup
      const int* xcolor = color + (T / 64) * KP; const int NX = NCOL - (T / 64) * KP;
      int2 nxt[PF][LPT];
#pragma unroll
      for (int p = 0; p < PF; p++)
#pragma unroll
        for (int u = 0; u < LPT; u++) { const int l = (p < NX ? xcolor[p] : 0) + tid + u * T; nxt[p][u] = (p < NX && l < xcolor[p + 1]) ? links[l] : make_int2(-1, 0); }
      for (int c0 = 0; c0 < NX; c0 += PF) {
#ifdef AGXC_NO_LINKS
        break;
#endif
#pragma unroll
        for (int p = 0; p < PF; p++) {
          const int c = c0 + p;
          if (c >= NX) break;
          int2 cur[LPT];
#pragma unroll
          for (int u = 0; u < LPT; u++) {
            cur[u] = nxt[p][u];
            const int l = (c + PF < NX ? xcolor[c + PF] : 0) + tid + u * T;
            nxt[p][u] = (c + PF < NX && l < xcolor[c + PF + 1]) ? links[l] : make_int2(-1, 0);
          }
#pragma unroll
          for (int u = 0; u < LPT; u++) if (cur[u].x >= 0) {
            const int a = cur[u].x & 0xffff, b = (cur[u].x >> 16) & 0xffff;
            const f3 xa = ldn(S.x + NS4 * a), xb = ldn(S.x + NS4 * b), del = xb - xa; const float len = dot(del, del), c1 = __int_as_float(cur[u].y);
            if (c1 + len > EPS) { const float k = (c1 - len) / (c1 + len) * kLST * 0.5f; stn(S.x + NS4 * a, xa - k * del); stn(S.x + NS4 * b, xb + k * del); }
          }
          lds_barrier();
        }
      }
    }
  }
  // report for the finish kernel; write back the positions and the velocities of the last substep
  if (greport) {
    if (tid < 6) st(greport + 3 * tid, ldn(S.x + NS4 * cl[AGX_CL_TRI + tid]));
#pragma unroll
    for (int j = 0; j < NPT; j++) {
      const int i = own[j];
      if (i >= 0)
#pragma unroll
      for (int cc = 0; cc < NODE_CONTACTS; cc++) {
        float* o = greport + 20 + 2 * (NODE_CONTACTS * i + cc);
        if (nsub > 0 && cc < ncon[j]) { const f3 f = (1.0f / dt) * ld(grec + CREC * (NODE_CONTACTS * i + cc) + 5); o[0] = S.x[NS4 * i + 2]; o[1] = sqrtf(dot(f, f)); } else { o[0] = 0.f; o[1] = -1.f; }
      }
    }
  }
  lds_barrier();
  const float vc = (1.0f - kDP) / dt;
  for (int k = tid; k < 3 * NN; k += T) { const int kk = NS4 * (k / 3) + k % 3; gcloth[k] = S.x[kk]; gcloth[3 * NN + k] = (S.x[kk] - S.q[kk]) * vc; }
}

}  // namespace agxc
