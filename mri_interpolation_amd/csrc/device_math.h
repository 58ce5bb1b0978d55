// Device math shared by the layer kernels (linear.hip, linear_small.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mri {

// sin and cos of one argument, ~1 ulp for |u| < 2^13 (SIREN pre-activations are a few tens):
// three-term Cody-Waite reduction by pi/2 with fused multiply-adds, Cephes minimax polynomials on
// [-pi/4, pi/4], quadrant fix-up.  About a third of the instructions of the library sincosf,
// which the epilogue of every SIREN layer pays per output element.
__device__ __forceinline__ void sincos_fast(float u, float* s_out, float* c_out) {
  if (fabsf(u) > 8192.0f) {
    sincosf(u, s_out, c_out);
    return;
  }
  const float k = rintf(u * 0.636619772367581343f);
  float r = __builtin_fmaf(k, -1.57079625129699707031e+00f, u);
  r = __builtin_fmaf(k, -7.54978941586159635335e-08f, r);
  r = __builtin_fmaf(k, -5.39030252995776476554e-15f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  const float sn = __builtin_fmaf(ps * z, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  const float cs = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const int q = (int)k & 3;
  const float s1 = (q & 1) ? cs : sn, c1 = (q & 1) ? sn : cs;
  *s_out = (q & 2) ? -s1 : s1;
  *c_out = ((q + 1) & 2) ? -c1 : c1;
}

__device__ __forceinline__ float gelu_f(float z) {
  return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float z) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
  return cdf + z * pdf;
}

}  // namespace mri
