// How fast can the chip start workgroups?  An (almost) empty kernel with the scatter kernel's launch shape:
// `wgs` workgroups of `threads` threads and `lds` bytes of LDS, each writing one word of LDS and leaving.
//   hipcc -O3 --offload-arch=gfx950 dispatch_probe.hip -o dispatch_probe && ./dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int LDS>
__global__ void probe(int* out) {
  __shared__ int buf[LDS / 4];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (buf[(threadIdx.x + 1) % blockDim.x] == -1) out[0] = 1;
}

template <int LDS>
float run(int wgs, int threads, int* out) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<LDS>, dim3(wgs), dim3(threads), 0, 0, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<LDS>, dim3(wgs), dim3(threads), 0, 0, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1e3f;
}

int main() {
  int* out;
  hipMalloc(&out, 4);
  const int shapes[][2] = {{6656, 512}, {13312, 256}, {26624, 128}, {3328, 1024}, {6656, 256}, {820, 1024}};
  for (auto& s : shapes) {
    printf("%6d workgroups x %4d threads: LDS 4 KB %7.1f us | 51 KB %7.1f us | 64 KB %7.1f us\n", s[0], s[1],
           run<4096>(s[0], s[1], out), run<52224>(s[0], s[1], out), run<65536>(s[0], s[1], out));
  }
  return 0;
}
