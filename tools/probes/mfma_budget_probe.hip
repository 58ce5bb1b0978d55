// What the chip sustains on a bare bf16 MFMA stream: one wave per SIMD (or two), every CU, v_mfma_f32_32x32x16_bf16 back to
// back on four accumulators for ~1 ms.  Prints MFMAs per second per SIMD as "GHz-equivalents of a fully busy matrix pipe"
// (MFMAs / s x 32 cycles): the yardstick for busy x clock of the SIREN kernels.
//   hipcc -O3 --offload-arch=gfx950 -w tools/probes/mfma_budget_probe.hip -o tools/probes/mfma_budget_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// RANDOM: eight operand pairs of random bits per lane, rotated through (the switching activity of real data); else one constant pair
template <bool RANDOM>
__global__ __launch_bounds__(512) void burn(float* sink, int iters, float seed, const bf16x8* rnd) {
  bf16x8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 8; ++j) a[q][j] = (__bf16)(seed + threadIdx.x * 0.001f + j), b[q][j] = (__bf16)(seed * 0.5f + j * 0.25f);
  if (RANDOM)
    for (int q = 0; q < 4; ++q) a[q] = rnd[(threadIdx.x * 8 + q) % 4096], b[q] = rnd[(threadIdx.x * 8 + 4 + q) % 4096];
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b[3], c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  float* sink;
  hipMalloc(&sink, 4);
  bf16x8* rnd;
  hipMalloc(&rnd, 4096 * 16);
  {
    unsigned short h[4096 * 8];
    unsigned x = 12345u;
    for (int i = 0; i < 4096 * 8; ++i) x = x * 1664525u + 1013904223u, h[i] = (unsigned short)(((x >> 9) & 0x7fff) % 0x7f00 | ((x >> 3) & 0x8000));  // finite bf16
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int random = 0; random < 2; ++random)
  for (int waves : {4, 8}) {
    for (int iters : {20000, 80000}) {
      auto launch = [&] {
        if (random)
          hipLaunchKernelGGL(burn<true>, dim3(256), dim3(64 * waves), 0, 0, sink, iters, 1.0f, rnd);
        else
          hipLaunchKernelGGL(burn<false>, dim3(256), dim3(64 * waves), 0, 0, sink, iters, 1.0f, rnd);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double mfma_per_simd = 4.0 * iters * (waves / 4);
      printf("%s operands, %d waves per CU, %6d x 4 MFMAs per wave: %.3f ms, %.2f GHz-equivalents of a busy matrix pipe (%.0f TF dense bf16)\n",
             random ? "random" : "constant", waves, iters, ms, mfma_per_simd * 32 / (ms * 1e-3) / 1e9, 256.0 * waves * 4.0 * iters * 32768 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
