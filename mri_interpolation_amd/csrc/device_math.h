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

typedef float f32x2 __attribute__((ext_vector_type(2)));
// The branch-free body of sincos_fast2 (|u| <= 8192 is the CALLER's business: siren_rows.hip interleaves this
// with matrix instructions, checks the range per tile and repairs out-of-range values on a cold path).
__device__ __forceinline__ void sincos_fast2_core(float u0, float u1, float* s0, float* c0, float* s1,
                                                  float* c1) {
  const f32x2 u = {u0, u1};
  const f32x2 t = u * 0.636619772367581343f;
  const f32x2 k = {rintf(t.x), rintf(t.y)};
  f32x2 r = __builtin_elementwise_fma(k, (f32x2)(-1.57079625129699707031e+00f), u);
  r = __builtin_elementwise_fma(k, (f32x2)(-7.54978941586159635335e-08f), r);
  r = __builtin_elementwise_fma(k, (f32x2)(-5.39030252995776476554e-15f), r);
  const f32x2 z = r * r;
  f32x2 ps = __builtin_elementwise_fma((f32x2)(-1.9515295891e-4f), z, (f32x2)(8.3321608736e-3f));
  ps = __builtin_elementwise_fma(ps, z, (f32x2)(-1.6666654611e-1f));
  const f32x2 sn = __builtin_elementwise_fma(ps * z, r, r);
  f32x2 pc = __builtin_elementwise_fma((f32x2)(2.443315711809948e-5f), z,
                                       (f32x2)(-1.388731625493765e-3f));
  pc = __builtin_elementwise_fma(pc, z, (f32x2)(4.166664568298827e-2f));
  const f32x2 half = __builtin_elementwise_fma((f32x2)(-0.5f), z, (f32x2)(1.0f));
  const f32x2 cs = __builtin_elementwise_fma(pc * z, z, half);
  {
    const int q = (int)k.x & 3;
    const float a = (q & 1) ? cs.x : sn.x, b = (q & 1) ? sn.x : cs.x;
    *s0 = (q & 2) ? -a : a;
    *c0 = ((q + 1) & 2) ? -b : b;
  }
  {
    const int q = (int)k.y & 3;
    const float a = (q & 1) ? cs.y : sn.y, b = (q & 1) ? sn.y : cs.y;
    *s1 = (q & 2) ? -a : a;
    *c1 = ((q + 1) & 2) ? -b : b;
  }
}

// Two arguments at once on the packed f32 FMA / multiply (v_pk_fma_f32, v_pk_mul_f32: two elements
// of f32 per instruction slot): the reduction and both polynomials are FMA chains, only the
// rounding, the quadrant logic and the range check stay per element.  Same arithmetic as
// sincos_fast, operation for operation.
__device__ __forceinline__ void sincos_fast2(float u0, float u1, float* s0, float* c0, float* s1,
                                             float* c1) {
#ifdef MRI_SCALAR_SINCOS  // A/B builds: beside bf16 MFMAs packed f32 VALU can be the slower form
  sincos_fast(u0, s0, c0);
  sincos_fast(u1, s1, c1);
  return;
#endif
  if (fabsf(u0) > 8192.0f || fabsf(u1) > 8192.0f) {
    sincos_fast(u0, s0, c0);
    sincos_fast(u1, s1, c1);
    return;
  }
  sincos_fast2_core(u0, u1, s0, c0, s1, c1);
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
