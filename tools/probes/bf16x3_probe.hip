// Accuracy probe: C = A B^T (K = 256) on one wave with (a) the f32 MFMA, (b) the bf16 MFMA on an
// exact three-way split of each f32 operand (x = h + m + l, 6 of the 9 cross products), against
// float64 on the host.   hipcc -O3 --offload-arch=gfx950 bf16x3_probe.hip -o bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 256;

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// A: [tiles][32][K], B: [tiles][32][K] (both k-contiguous); C: [tiles][variant][32][32]
__global__ __launch_bounds__(64) void probe(const float* A, const float* B, float* C, int order) {
  const int lane = threadIdx.x, i = lane & 31, half = lane >> 5;
  const float* a = A + ((size_t)blockIdx.x * 32 + i) * K;
  const float* b = B + ((size_t)blockIdx.x * 32 + i) * K;
  f32x16 c0 = {0}, c1 = {0}, c2 = {0};
  for (int k = 0; k < K; k += 2)
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + half], b[k + half], c0, 0, 0, 0);
  for (int k = 0; k < K; k += 16) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
      __bf16 h, m, l;
      split3(a[k + 8 * half + j], h, m, l);
      ah[j] = h, am[j] = m, al[j] = l;
      split3(b[k + 8 * half + j], h, m, l);
      bh[j] = h, bm[j] = m, bl[j] = l;
    }
    if (order == 0) {  // small terms first
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
    } else {
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c1, 0, 0, 0);
    }
    // two accumulators: the large term alone, the five corrections together, summed at the end
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c2, 0, 0, 0);
  }
  float* c = C + (size_t)blockIdx.x * 3 * 1024;
  for (int r = 0; r < 16; ++r) {
    const int row = (r / 4) * 8 + half * 4 + (r % 4);
    c[row * 32 + i] = c0[r];
    c[1024 + row * 32 + i] = c1[r];
    c[2048 + row * 32 + i] = c2[r];
  }
}

int main(int argc, char** argv) {
  const int tiles = 256;
  for (int dist = 0; dist < 3; ++dist)
    for (int order = 0; order < 2; ++order) {
      std::vector<float> A((size_t)tiles * 32 * K), B(A.size());
      srand(17 + dist);
      auto u = [] { return 2.0 * rand() / RAND_MAX - 1.0; };
      for (size_t e = 0; e < A.size(); ++e) {
        if (dist == 0) A[e] = (float)u(), B[e] = (float)u();
        if (dist == 1) A[e] = (float)sin(30.0 * u()), B[e] = (float)(u() * 0.005);  // SIREN layer
        if (dist == 2) A[e] = (float)(u() > 0 ? u() * u() : 0.0), B[e] = (float)(u() * 0.3);  // ReLU decoder
      }
      float *dA, *dB, *dC;
      hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, (size_t)tiles * 3 * 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, dim3(tiles), dim3(64), 0, 0, dA, dB, dC, order);
      std::vector<float> C((size_t)tiles * 3 * 1024);
      if (hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
      double e_f32 = 0, e_x3 = 0, e_hh = 0, r_f32 = 0, r_x3 = 0, cmax = 0;
      for (int t = 0; t < tiles; ++t)
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) {
              const double p = (double)A[((size_t)t * 32 + i) * K + k] * B[((size_t)t * 32 + j) * K + k];
              ref += p, mag += fabs(p);
            }
            const float* c = &C[(size_t)t * 3 * 1024];
            cmax = fmax(cmax, fabs(ref));
            e_f32 = fmax(e_f32, fabs(c[i * 32 + j] - ref) / mag);
            e_x3 = fmax(e_x3, fabs(c[1024 + i * 32 + j] - ref) / mag);
            e_hh = fmax(e_hh, fabs(c[2048 + i * 32 + j] - ref) / mag);
            r_f32 += pow((c[i * 32 + j] - ref) / mag, 2), r_x3 += pow((c[1024 + i * 32 + j] - ref) / mag, 2);
          }
      const double cnt = (double)tiles * 1024;
      printf("dist %d order %d: max err / sum|ab|: f32 mfma %.3e  bf16x3 %.3e  (bf16 hh only %.3e); rms f32 %.3e bf16x3 %.3e\n",
             dist, order, e_f32, e_x3, e_hh, sqrt(r_f32 / cnt), sqrt(r_x3 / cnt));
      hipFree(dA), hipFree(dB), hipFree(dC);
    }
  return 0;
}
