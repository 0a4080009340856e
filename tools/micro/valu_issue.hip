// valu_issue.hip -- micro-benchmark behind the "what bounds the solve kernel" analysis: issue cost and dependent-chain
// latency of the instruction kinds the PGS row update is made of, measured with s_memtime on ONE wave (1 wave on a SIMD) and with
// 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.001f + 1.0f, b = 1.000001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f;
  int lane = threadIdx.x & 63;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {          // dependent v_fma_f32 chain
      REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    } else if (MODE == 1) {   // 4 independent v_fma_f32 chains interleaved
      REP16(REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));))
    } else if (MODE == 2) {   // dependent chain through DPP (row_shr:1) adds
      REP256(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
    } else if (MODE == 3) {   // v_readlane -> SGPR -> v_fmac using the SGPR (the broadcast step of a row visit)
      REP256(asm volatile("v_readlane_b32 s40, %0, 5\n s_nop 1\n v_fmac_f32 %0, s40, %1" : "+v"(a) : "v"(c) : "s40");)
    } else if (MODE == 4) {   // the row-space visit chain: sub, fma, med3, sub, readlane, fmac
      REP256(asm volatile("v_sub_f32 %1, %3, %0\n v_fma_f32 %1, %1, %4, %2\n v_med3_f32 %1, %1, %5, %6\n v_sub_f32 %1, %1, %2\n v_readlane_b32 s40, %1, 7\n s_nop 1\n v_fmac_f32 %0, s40, %4"
                          : "+v"(a), "+v"(d) : "v"(e), "v"(f), "v"(b), "v"(c), "v"(f) : "s40");)
    } else if (MODE == 5) {   // packed: v_pk_fma_f32 dependent chain
      REP256(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&a) : "v"(*(double*)&b), "v"(*(double*)&c));)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> double run(int threads, int iters, int per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[64]; hipMemcpy(h, cyc, (threads / 64) * 8, hipMemcpyDeviceToHost);
  long long mx = 0; for (int w = 0; w < threads / 64; w++) if (h[w] > mx) mx = h[w];
  hipFree(out); hipFree(cyc);
  return (double)mx / ((double)iters * per_iter);
}
int main() {
  // s_memtime counts at a fixed 100 MHz-class "shader clock"?  Calibrate against wall time with a long run of MODE 0.
  const int it = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* out; long long* cyc; hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 20000);
  hipEventRecord(e0, 0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 20000); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
  printf("{\"counter_ticks_per_us\": %.2f, \"dependent_fma_ns\": %.3f,\n", (double)c0 / (ms * 1e3), ms * 1e6 / (20000.0 * 256));
  const char* names[6] = {"dependent v_fma_f32", "4 independent v_fma_f32 chains", "dependent v_add_f32_dpp (+s_nop 1)", "v_readlane -> s_nop 1 -> v_fmac(sgpr)",
                          "row-space visit chain (sub,fma,med3,sub,readlane,nop,fmac)", "dependent v_pk_fma_f32"};
  const int per[6] = {256, 1024, 256, 256, 256, 256};
  printf(" \"ticks_per_item\": {\n");
  double r;
#define LINE(M) r = run<M>(64, it, per[M]); printf("  \"%s | 1 wave per SIMD\": %.3f,\n", names[M], r); r = run<M>(1024, it, per[M]); printf("  \"%s | 16 waves on one CU (4 per SIMD)\": %.3f,\n", names[M], r);
  LINE(0) LINE(1) LINE(2) LINE(3) LINE(4) LINE(5)
  printf("  \"end\": 0}}\n");
  return 0;
}
