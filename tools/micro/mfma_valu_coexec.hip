// Do fp32 MFMA and ordinary fp32 VALU work from two different waves of one SIMD overlap on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 issue v_mfma_f32_32x32x2_f32 only, waves 4-7 v_fma_f32 only
// (wave w and w+4 share a SIMD).  Times: MFMA waves alone, VALU waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu_coexec tools/micro/mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void k(float* out, int mfma_iters, int valu_iters, int trans) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = threadIdx.x * 1e-3f + 0.5f, b = blockIdx.x * 1e-3f - 0.25f;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float c = 0.999f, d = 1e-3f;
    for (int it = 0; it < valu_iters; ++it) {
      if (trans) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_rcpf(v[i] + 1.5f);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], c, d);
      }
    }
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float run(float* out, int mi, int vi, int trans) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mi, vi, trans);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mi, vi, trans);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main() {
  float* out; (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  const int mi = 20000;                     // 160k MFMAs per wave = 10.24 M cycles
  for (int trans = 0; trans < 2; ++trans) {
    const int vi = trans ? 10000 : 40000;   // 16 VALU per iteration
    const float ta = run(out, mi, 0, trans), tb = run(out, 0, vi, trans), tc = run(out, mi, vi, trans);
    printf("%s: MFMA alone %.3f ms (%.1f TFLOP/s), VALU alone %.3f ms, both %.3f ms  -> %s\n", trans ? "v_rcp_f32" : "v_fma_f32",
           ta, 256.0 * 4 * mi * 8 * 4096.0 / ta / 1e9, tb, tc, tc < 0.5f * (ta + tb) + 0.5f * (ta > tb ? ta : tb) ? "overlap" : "serialised");
  }
  return 0;
}
