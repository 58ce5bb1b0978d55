// How fast can one persistent workgroup per CU stream two (n, 256) f32 arrays from HBM -- the weight-gradient kernels'
// access pattern (1 KiB rows, chunk c of both arrays, chunks interleaved over the workgroups) -- by LDS-DMA into a ring,
// and by plain 16-byte loads into registers?  No compute: the yardstick for siren_wgrad_kernel (2.15 GB per launch).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o tools/probes/stream_probe && tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kH = 256, kRows = 16, kChunkFloats = 2 * kRows * kH;

template <int DEPTH, int WAVES>  // chunks in flight; waves per workgroup (each moves 32 / WAVES pieces of a chunk)
__global__ __launch_bounds__(64 * WAVES) void dma_stream(const float* a, const float* b, long n, float* sink) {
  __shared__ float ring[DEPTH + 1][kChunkFloats];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long chunks = n / kRows, steps = (chunks - blockIdx.x + gridDim.x - 1) / gridDim.x;
  constexpr int per = 2 * kRows / WAVES;  // pieces per wave and chunk
  auto issue = [&](long s, int slot) {
    const long c = blockIdx.x + s * gridDim.x < chunks ? blockIdx.x + s * gridDim.x : chunks - 1;
#pragma unroll
    for (int i = 0; i < per / 2; ++i) {
      const int r = wave + WAVES * i;
      const long src = (c * kRows + r) * kH + 4 * lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + src),
                                       (__attribute__((address_space(3))) void*)(ring[slot] + r * kH), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + src),
                                       (__attribute__((address_space(3))) void*)(ring[slot] + (kRows + r) * kH), 16, 0, 0);
    }
  };
  for (int s = 0; s < DEPTH; ++s) issue(s, s);
  float acc = 0.f;
  for (long s = 0; s < steps; ++s) {
    // chunk s has landed (behind it: DEPTH - 1 chunks)
    if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(per) : "memory");
    if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * per) : "memory");
    if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * per) : "memory");
    __builtin_amdgcn_s_barrier();
    acc += ring[s % (DEPTH + 1)][threadIdx.x];  // (a token read)
    issue(s + DEPTH, (int)((s + DEPTH) % (DEPTH + 1)));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 12345.678f) sink[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void reg_stream(const float4* a, const float4* b, long n4, float* sink) {
  float acc = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    float4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) va[u] = a[i + u * stride], vb[u] = b[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += va[u].x + vb[u].y;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const long n = 1 << 20;
  float *a, *b, *sink;
  hipMalloc(&a, n * kH * 4), hipMalloc(&b, n * kH * 4), hipMalloc(&sink, 4);
  hipMemset(a, 0, n * kH * 4), hipMemset(b, 0, n * kH * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const double gb = 2.0 * n * kH * 4 / 1e9;
  auto time = [&](const char* name, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms / 5, gb / (ms / 5));
  };
  time("LDS-DMA ring, 1 chunk in flight, 4 waves", [&] { hipLaunchKernelGGL((dma_stream<1, 4>), dim3(256), dim3(256), 0, 0, a, b, n, sink); });
  time("LDS-DMA ring, 2 chunks in flight, 4 waves", [&] { hipLaunchKernelGGL((dma_stream<2, 4>), dim3(256), dim3(256), 0, 0, a, b, n, sink); });
  time("LDS-DMA ring, 3 chunks in flight, 4 waves", [&] { hipLaunchKernelGGL((dma_stream<3, 4>), dim3(256), dim3(256), 0, 0, a, b, n, sink); });
  time("LDS-DMA ring, 4 chunks in flight, 4 waves", [&] { hipLaunchKernelGGL((dma_stream<4, 4>), dim3(256), dim3(256), 0, 0, a, b, n, sink); });
  time("LDS-DMA ring, 3 chunks in flight, 8 waves", [&] { hipLaunchKernelGGL((dma_stream<3, 8>), dim3(256), dim3(512), 0, 0, a, b, n, sink); });
  time("LDS-DMA ring, 3 chunks, 2 workgroups per CU", [&] { hipLaunchKernelGGL((dma_stream<3, 4>), dim3(512), dim3(256), 0, 0, a, b, n, sink); });
  time("register loads, 1024 x 256 threads, unroll 4", [&] { hipLaunchKernelGGL((reg_stream<4>), dim3(1024), dim3(256), 0, 0, (const float4*)a, (const float4*)b, n * kH / 4, sink); });
  time("register loads, 2048 x 256 threads, unroll 8", [&] { hipLaunchKernelGGL((reg_stream<8>), dim3(2048), dim3(256), 0, 0, (const float4*)a, (const float4*)b, n * kH / 4, sink); });
  time("register loads, 256 x 256 threads, unroll 8", [&] { hipLaunchKernelGGL((reg_stream<8>), dim3(256), dim3(256), 0, 0, (const float4*)a, (const float4*)b, n * kH / 4, sink); });
  return 0;
}
