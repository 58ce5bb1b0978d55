// Micro-benchmark: LDS atomic throughput on gfx950 (design input for hashgrid backward).
// hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE, int ACTIVE_SHIFT>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ idx, int iters, float* out) {
  __shared__ float accf[32768];
  uint32_t* accu = reinterpret_cast<uint32_t*>(accf);
  unsigned long long* accl = reinterpret_cast<unsigned long long*>(accf);
  for (int s = threadIdx.x; s < 32768; s += 1024) accf[s] = 0.f;
  __syncthreads();
  const bool active = (threadIdx.x & ((1 << ACTIVE_SHIFT) - 1)) == 0;
  uint32_t h = idx[threadIdx.x + blockIdx.x * 1024];
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    const uint32_t slot = (h >> 8);
    if (active) {
      if (MODE == 0) atomicAdd(&accf[slot & 32767], 1.0f);
      if (MODE == 1) atomicAdd(&accu[slot & 32767], 1u);
      if (MODE == 2) atomicAdd(&accl[slot & 16383], 1ull);
      if (MODE == 3) accf[slot & 32767] += 1.0f;  // racy plain RMW, for comparison
      if (MODE == 4) { atomicAdd(&accf[(slot & 16383) * 2], 1.0f); atomicAdd(&accf[(slot & 16383) * 2 + 1], 1.0f); }
      if (MODE == 5) atomicAdd(&accf[(slot & 1023) * 32 + (threadIdx.x & 31)], 1.0f);  // conflict-free banks
    }
  }
  __syncthreads();
  float s = 0;
  for (int t = threadIdx.x; t < 32768; t += 1024) s += accf[t];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int SH>
void run(const char* name, const uint32_t* idx, float* out) {
  const int iters = 2000, blocks = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, SH>), dim3(blocks), dim3(1024), 0, 0, idx, iters, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, SH>), dim3(blocks), dim3(1024), 0, 0, idx, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double lane_ops = (double)blocks * 1024 * iters / (1 << SH) * (MODE == 4 ? 2 : 1);
  printf("%-34s active 1/%d: %8.3f ms  %8.2f G lane-atomics/s chip  (%.2f per CU per ns)\n", name, 1 << SH, ms,
         lane_ops / ms / 1e6, lane_ops / ms / 1e6 / 256);
}

int main() {
  uint32_t* idx; float* out;
  hipMalloc(&idx, 256 * 1024 * 4); hipMalloc(&out, 4);
  uint32_t* h = (uint32_t*)malloc(256 * 1024 * 4);
  for (int i = 0; i < 256 * 1024; ++i) h[i] = i * 2654435761u + 12345u;
  hipMemcpy(idx, h, 256 * 1024 * 4, hipMemcpyHostToDevice);
  run<0, 0>("ds_add_f32 random", idx, out);
  run<1, 0>("ds_add_u32 random", idx, out);
  run<2, 0>("ds_add_u64 random", idx, out);
  run<3, 0>("plain RMW random (racy)", idx, out);
  run<4, 0>("2x ds_add_f32 adjacent (F=2)", idx, out);
  run<5, 0>("ds_add_f32 conflict-free banks", idx, out);
  run<0, 5>("ds_add_f32 random", idx, out);
  run<1, 5>("ds_add_u32 random", idx, out);
  run<2, 5>("ds_add_u64 random", idx, out);
  run<4, 5>("2x ds_add_f32 adjacent (F=2)", idx, out);
  return 0;
}
