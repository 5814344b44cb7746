// Attainable fp32 MFMA rate on this board: waves issue nothing but v_mfma_f32_32x32x2_f32 on register operands.
// Prints TFLOP/s and the shader clock observed inside the kernel (s_memtime ticks per 100 MHz s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_peak tools/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, unsigned long long* clk, int iters, int mode) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // operand bits matter for power: mode 0 = near-zero constants, 1 = pseudo-random operands that change every MFMA
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float av[8], bv[8];
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; av[i] = mode ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : threadIdx.x * 1e-3f;
    h = h * 1664525u + 1013904223u; bv[i] = mode ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : blockIdx.x * 1e-3f;
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[(i + j) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20000, reps = 20, mode = argc > 3 ? atoi(argv[3]) : 1;
  constexpr int NACC = 8;
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, sizeof(float) * blocks * 256); (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, mode);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, mode);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * blocks);
  (void)hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double ghz = 0; for (int i = 0; i < blocks; ++i) ghz += (double)h[2 * i] / ((double)h[2 * i + 1] * 10e-9) / 1e9;
  ghz /= blocks;
  const double flops = (double)blocks * 4 /*waves*/ * iters * NACC * 2.0 * 32 * 32 * 2;
  printf("mode %d blocks %d iters %d: %.3f ms/launch  %.1f TFLOP/s  shader clock %.3f GHz  (ideal at that clock: %.1f TFLOP/s)\n",
         mode, blocks, iters, ms, flops / ms / 1e9, ghz, 256 * 4 * (2.0 * 32 * 32 * 2 / 64) * ghz / 1e3);
  return 0;
}
