// NeRF-style frequency encoding, reference encoding.py:43-66 (`Frequency`):
//   out[r][d*2L + l]     = sin(x[r][d] * 2^l)
//   out[r][d*2L + L + l] = cos(x[r][d] * 2^l)          l = 0..L-1
// and its gradient dx[r][d] = sum_l 2^l (cos * g_sin - sin * g_cos).  Pure streaming work: one
// thread per (row, axis) computes the 2L outputs of that axis, a wave writes 64 consecutive
// (row, axis) runs, so stores of one instruction cover one contiguous span of the output.
// 2^l is exact in f32, so x * 2^l is the same float the reference's `x * freqs` produces; sine
// and cosine are the full-range OCML versions (arguments reach 2^(L-1)), not the SIREN fast path.
#include <math.h>

#include "common.h"

namespace mri {
namespace {

__global__ __launch_bounds__(256) void frequency_forward_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t n, int dim, int n_levels,
    float* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * dim) return;
  const int64_t r = i / dim;
  const int d = (int)(i - r * dim);
  const float v = x[r * ldx + d];
  float* __restrict__ o = out + r * ldo + (int64_t)d * 2 * n_levels;
  float f = 1.0f;
  for (int l = 0; l < n_levels; ++l, f *= 2.0f) {
    float s, c;
    sincosf(v * f, &s, &c);
    o[l] = s;
    o[n_levels + l] = c;
  }
}

__global__ __launch_bounds__(256) void frequency_backward_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ g, int64_t ldg, int64_t n,
    int dim, int n_levels, float* __restrict__ dx, int64_t lddx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * dim) return;
  const int64_t r = i / dim;
  const int d = (int)(i - r * dim);
  const float v = x[r * ldx + d];
  const float* __restrict__ gr = g + r * ldg + (int64_t)d * 2 * n_levels;
  // torch: d(x*freqs)[l] = g_sin[l]*cos + g_cos[l]*(-sin); dx = sum_l d[l]*freqs[l] (ascending l)
  float acc = 0.f, f = 1.0f;
  for (int l = 0; l < n_levels; ++l, f *= 2.0f) {
    float s, c;
    sincosf(v * f, &s, &c);
    acc += (gr[l] * c + gr[n_levels + l] * (-s)) * f;
  }
  dx[r * lddx + d] = acc;
}

}  // namespace
}  // namespace mri

extern "C" {

int mri_frequency_forward(const float* x, int64_t ldx, int64_t n, int32_t dim, int32_t n_levels,
                          float* out, int64_t ldo, void* stream) {
  using namespace mri;
  MRI_REQUIRE(n >= 0 && dim >= 1 && n_levels >= 1 && n_levels <= 64, "frequency: bad sizes");
  MRI_REQUIRE(ldx >= dim && ldo >= (int64_t)dim * 2 * n_levels, "frequency: bad strides");
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && out, "frequency: null pointer");
  hipLaunchKernelGGL(frequency_forward_kernel, dim3((unsigned)ceil_div(n * dim, 256)), dim3(256),
                     0, (hipStream_t)stream, x, ldx, n, dim, n_levels, out, ldo);
  return check_launch("frequency_forward");
}

int mri_frequency_backward(const float* x, int64_t ldx, const float* d_out, int64_t ldg, int64_t n,
                           int32_t dim, int32_t n_levels, float* dx, int64_t lddx, void* stream) {
  using namespace mri;
  MRI_REQUIRE(n >= 0 && dim >= 1 && n_levels >= 1 && n_levels <= 64, "frequency: bad sizes");
  MRI_REQUIRE(ldx >= dim && lddx >= dim && ldg >= (int64_t)dim * 2 * n_levels,
              "frequency: bad strides");
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && d_out && dx, "frequency: null pointer");
  hipLaunchKernelGGL(frequency_backward_kernel, dim3((unsigned)ceil_div(n * dim, 256)), dim3(256),
                     0, (hipStream_t)stream, x, ldx, d_out, ldg, n, dim, n_levels, dx, lddx);
  return check_launch("frequency_backward");
}

}  // extern "C"
