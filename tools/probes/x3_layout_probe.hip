// Checks the LDS image layout and the operand maps of csrc/bf16x3.h with one wave, against float64:
//   chain : Yt[n][b] = sum_k W[n][k] X[b][k]     (A = weight fragments in registers, B = row reads)
//   wgrad : D[n][k]  = sum_b Z[b][n] X[b][k]     (both operands by transposed reads)
//   narrow: the same two on the [32][32] input image
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../mri_interpolation_amd/csrc x3_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bf16x3.h"
using namespace mri::x3;

// lane (b = li + 16 t, g) stores columns 4 g' .. : the way the kernel's epilogues write an image
// (a lane owns 4 consecutive columns of one row): here one wave walks all column quads.
__device__ void fill_image(char* img, const float* X, int cols, int lane, bool narrow) {
  const int li = lane & 15, g = lane >> 4;
  for (int t = 0; t < 2; ++t)
    for (int c4 = g; c4 < cols / 4; c4 += 4) {
      const int row = 16 * t + li;
      const float* x = X + row * cols + 4 * c4;
      uint32_t h0, m0, l0, h1, m1, l1;
      split2(x[0], x[1], h0, m0, l0);
      split2(x[2], x[3], h1, m1, l1);
      const int off = (narrow ? img32_off(row, c4 >> 1) : img_off(row, c4 >> 1)) + 8 * (c4 & 1);
      const int term = narrow ? kImg32Bytes : kImgBytes;
      *reinterpret_cast<u32x2*>(img + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(img + term + off) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(img + 2 * term + off) = u32x2{l0, l1};
    }
}

__global__ __launch_bounds__(64) void probe(const float* W, const float* X, const float* Z,
                                            const float* Xn, const float* Wn, float* out) {
  __shared__ __attribute__((aligned(16))) char xi[3 * kImgBytes];
  __shared__ __attribute__((aligned(16))) char zi[3 * kImgBytes];
  __shared__ __attribute__((aligned(16))) char ni[3 * kImg32Bytes];
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  fill_image(xi, X, 128, lane, false);
  fill_image(zi, Z, 128, lane, false);
  fill_image(ni, Xn, 32, lane, true);
  __syncthreads();
  // ---- chain: strip of 16 rows of W (n = li), K = 128 in 4 steps; two 16-row sub-tiles of X
  for (int t = 0; t < 2; ++t) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 4; ++s) {
      float wv[8];
      for (int j = 0; j < 8; ++j) wv[j] = W[li * 128 + 32 * s + 8 * g + j];
      const Frag a = split8(wv);
      Frag b;
      const int off = img_off(16 * t + li, 4 * s + g);
      b.h = lds_read_b128(xi + off), b.m = lds_read_b128(xi + kImgBytes + off),
      b.l = lds_read_b128(xi + 2 * kImgBytes + off);
      c = mma6(a, b, c);
    }
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 32 + 16 * t + li] = c[r];  // Yt[n][b]
  }
  // ---- wgrad: D[n][k], n = columns 16..31 of Z (chunk 2), k-tiles 0..7 of X
  {
    Frag a;
    auto off = [](int row, int ch) { return img_off(row, ch); };
    a.h = tr_frag(zi, 2, lane, off), a.m = tr_frag(zi + kImgBytes, 2, lane, off),
    a.l = tr_frag(zi + 2 * kImgBytes, 2, lane, off);
    for (int kt = 0; kt < 8; ++kt) {
      Frag b;
      b.h = tr_frag(xi, 2 * kt, lane, off), b.m = tr_frag(xi + kImgBytes, 2 * kt, lane, off),
      b.l = tr_frag(xi + 2 * kImgBytes, 2 * kt, lane, off);
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = mma6(a, b, c);
      for (int r = 0; r < 4; ++r) out[1024 + (4 * g + r) * 128 + 16 * kt + li] = c[r];
    }
  }
  // ---- narrow image: chain Y1t[n][b] = sum_kin Wn[n][kin] Xn[b][kin] (one step), and
  //      D1[n][kin] = sum_b Z[b][16 + n] Xn[b][kin]
  {
    float wv[8];
    for (int j = 0; j < 8; ++j) wv[j] = Wn[li * 32 + 8 * g + j];
    const Frag a = split8(wv);
    for (int t = 0; t < 2; ++t) {
      Frag b;
      const int off = img32_off(16 * t + li, g);
      b.h = lds_read_b128(ni + off), b.m = lds_read_b128(ni + kImg32Bytes + off),
      b.l = lds_read_b128(ni + 2 * kImg32Bytes + off);
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = mma6(a, b, c);
      for (int r = 0; r < 4; ++r) out[4096 + (4 * g + r) * 32 + 16 * t + li] = c[r];
    }
    auto off = [](int row, int ch) { return img_off(row, ch); };
    auto offn = [](int row, int ch) { return img32_off(row, ch); };
    Frag za;
    za.h = tr_frag(zi, 2, lane, off), za.m = tr_frag(zi + kImgBytes, 2, lane, off),
    za.l = tr_frag(zi + 2 * kImgBytes, 2, lane, off);
    for (int kt = 0; kt < 2; ++kt) {
      Frag b;
      b.h = tr_frag(ni, 2 * kt, lane, offn), b.m = tr_frag(ni + kImg32Bytes, 2 * kt, lane, offn),
      b.l = tr_frag(ni + 2 * kImg32Bytes, 2 * kt, lane, offn);
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = mma6(za, b, c);
      for (int r = 0; r < 4; ++r) out[8192 + (4 * g + r) * 32 + 16 * kt + li] = c[r];
    }
  }
}

int main() {
  std::vector<float> W(16 * 128), X(32 * 128), Z(32 * 128), Xn(32 * 32), Wn(16 * 32), out(16384, 0.f);
  srand(3);
  auto u = [] { return (float)(2.0 * rand() / RAND_MAX - 1.0); };
  for (auto& v : W) v = u();
  for (auto& v : X) v = u();
  for (auto& v : Z) v = u() * 1e-6f;
  for (auto& v : Xn) v = u() * 1e-4f;
  for (auto& v : Wn) v = u();
  float *dW, *dX, *dZ, *dXn, *dWn, *dO;
  hipMalloc(&dW, W.size() * 4), hipMalloc(&dX, X.size() * 4), hipMalloc(&dZ, Z.size() * 4);
  hipMalloc(&dXn, Xn.size() * 4), hipMalloc(&dWn, Wn.size() * 4), hipMalloc(&dO, out.size() * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dZ, Z.data(), Z.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dXn, Xn.data(), Xn.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dWn, Wn.data(), Wn.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dO, 0, out.size() * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dW, dX, dZ, dXn, dWn, dO);
  if (hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  double e1 = 0, e2 = 0, e3 = 0, e4 = 0;
  for (int n = 0; n < 16; ++n)
    for (int b = 0; b < 32; ++b) {
      double r = 0, mag = 0;
      for (int k = 0; k < 128; ++k) r += (double)W[n * 128 + k] * X[b * 128 + k], mag += fabs((double)W[n * 128 + k] * X[b * 128 + k]);
      e1 = fmax(e1, fabs(out[n * 32 + b] - r) / mag);
      r = 0, mag = 0;
      for (int k = 0; k < 32; ++k) r += (double)Wn[n * 32 + k] * Xn[b * 32 + k], mag += fabs((double)Wn[n * 32 + k] * Xn[b * 32 + k]);
      e3 = fmax(e3, fabs(out[4096 + n * 32 + b] - r) / mag);
    }
  for (int n = 0; n < 16; ++n) {
    for (int k = 0; k < 128; ++k) {
      double r = 0, mag = 0;
      for (int b = 0; b < 32; ++b) r += (double)Z[b * 128 + 16 + n] * X[b * 128 + k], mag += fabs((double)Z[b * 128 + 16 + n] * X[b * 128 + k]);
      e2 = fmax(e2, fabs(out[1024 + n * 128 + k] - r) / mag);
    }
    for (int k = 0; k < 32; ++k) {
      double r = 0, mag = 0;
      for (int b = 0; b < 32; ++b) r += (double)Z[b * 128 + 16 + n] * Xn[b * 32 + k], mag += fabs((double)Z[b * 128 + 16 + n] * Xn[b * 32 + k]);
      e4 = fmax(e4, fabs(out[8192 + n * 32 + k] - r) / mag);
    }
  }
  printf("max err / sum|ab|: chain %.3e  wgrad %.3e  narrow chain %.3e  narrow wgrad %.3e  (f32 level: ~2e-7)\n", e1, e2, e3, e4);
  return (e1 < 1e-6 && e2 < 1e-6 && e3 < 1e-6 && e4 < 1e-6) ? 0 : 2;
}
