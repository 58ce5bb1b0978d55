// Fully connected layers whose input or output width is tiny (<= 8 / <= 4): the first layer of a
// coordinate network (K = 2, 3, 4 coordinates) and its last layer (N = 1 intensity), reference
// models.py:199-228 (SirenNet first / last layer) and :46-56 (BaseMLP).  These products have
// no reuse to speak of -- a 32x32 MFMA tile would be >= 90 % padding -- and are bound by streaming
// the wide side once (M x 256 floats = 1 GiB at M = 2^20), so forward and backward-data run on the
// VALU with 16-byte accesses and wave shuffles instead of going through linear.hip's tiles.
// (The weight gradient of a tiny OUTPUT stays on the split-batch MFMA kernel, which measured 2-3x
// faster than a VALU column sweep: 0.35 ms vs 0.97 ms for N = 1, K = 256, M = 2^20; the weight
// gradient of a tiny INPUT has its own streaming kernel below.)
#include <math.h>

#include <algorithm>

#include "common.h"
#include "device_math.h"

namespace mri {
namespace {

constexpr int kSmallK = 8;  // widest "tiny input" handled here
constexpr int kSmallN = 4;  // widest "tiny output"

__device__ __forceinline__ void activate(int act, float w0, float z, float& y, float& d) {
  y = z;
  d = 1.0f;
  if (act == MRI_ACT_RELU) {
    y = fmaxf(z, 0.f);
  } else if (act == MRI_ACT_SINE) {
    float s, c;
    sincos_fast(w0 * z, &s, &c);
    y = s;
    d = w0 * c;
  } else if (act == MRI_ACT_GELU) {
    d = gelu_grad_f(z);
    y = gelu_f(z);
  }
}

// ---- K <= 8: y[m][n] = act(w0 (sum_k x[m][k] W[n][k] + b[n])), thread <-> output column ------
constexpr int kRowsPerBlock = 64;

__global__ __launch_bounds__(256) void small_k_forward_kernel(
    const float* __restrict__ x, int64_t xrs, int64_t xcs, const float* __restrict__ w,
    const float* __restrict__ b, int64_t m, int n, int k, int act, float w0,
    float* __restrict__ y, int64_t ldy, float* __restrict__ deriv, int64_t ldd) {
  const int col = blockIdx.y * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock, r1 = min(m, r0 + kRowsPerBlock);
  float wr[kSmallK];
#pragma unroll
  for (int kk = 0; kk < kSmallK; ++kk) wr[kk] = (kk < k && col < n) ? w[col * k + kk] : 0.f;
  const float bias = (b && col < n) ? b[col] : 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    float z = 0.f;
#pragma unroll
    for (int kk = 0; kk < kSmallK; ++kk)
      if (kk < k) z += x[r * xrs + kk * xcs] * wr[kk];  // row-uniform loads
    float yv, dv;
    activate(act, w0, z + bias, yv, dv);
    if (col < n) {
      y[r * ldy + col] = yv;
      if (deriv) deriv[r * ldd + col] = dv;
    }
  }
}

// ---- N <= 4: y[m][n] = act(...), one wave per row, lanes stride the K axis by float4 -----------
__global__ __launch_bounds__(256) void small_n_forward_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
    const float* __restrict__ b, int64_t m, int n, int k, int act, float w0,
    float* __restrict__ y, int64_t ldy, float* __restrict__ deriv, int64_t ldd) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (int64_t)gridDim.x * 4;
  const bool vec = (k % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  for (int64_t r = wave; r < m; r += waves) {
    float acc[kSmallN] = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ xr = x + r * ldx;
    if (vec) {
      for (int k4 = lane * 4; k4 < k; k4 += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + k4);
#pragma unroll
        for (int j = 0; j < kSmallN; ++j)
          if (j < n) {
            const float4 wv = *reinterpret_cast<const float4*>(w + j * k + k4);
            acc[j] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
          }
      }
    } else {
      for (int kk = lane; kk < k; kk += 64)
#pragma unroll
        for (int j = 0; j < kSmallN; ++j)
          if (j < n) acc[j] += xr[kk] * w[j * k + kk];
    }
#pragma unroll
    for (int j = 0; j < kSmallN; ++j)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < kSmallN; ++j)
        if (j < n) {
          float yv, dv;
          activate(act, w0, acc[j] + (b ? b[j] : 0.f), yv, dv);
          y[r * ldy + j] = yv;
          if (deriv) deriv[r * ldd + j] = dv;
        }
    }
  }
}

// ---- N <= 4: dx[m][k] = (sum_n dy[m][n] W[n][k]) (.) g[m][k], thread <-> 4 columns of a row ----
__global__ __launch_bounds__(256) void small_n_backward_data_kernel(
    const float* __restrict__ dy, int64_t lddy, const float* __restrict__ w, int64_t m, int n,
    int k, int mode, const float* __restrict__ g, int64_t ldg, float* __restrict__ dx,
    int64_t lddx) {
  const int k4 = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (k4 >= k) return;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock, r1 = min(m, r0 + kRowsPerBlock);
  const bool full = k4 + 3 < k;
  const bool vec = full && (lddx % 4 == 0) && ((reinterpret_cast<uintptr_t>(dx) & 15) == 0) &&
                   (mode == MRI_DERIV_NONE ||
                    ((ldg % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0)));
  float wv[kSmallN][4];
#pragma unroll
  for (int j = 0; j < kSmallN; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) wv[j][c] = (j < n && k4 + c < k) ? w[j * k + k4 + c] : 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kSmallN; ++j)
      if (j < n) {
        const float d = dy[r * lddy + j];  // row-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += d * wv[j][c];
      }
    if (vec) {
      if (mode != MRI_DERIV_NONE) {
        const float4 gv = *reinterpret_cast<const float4*>(g + r * ldg + k4);
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          v[c] = mode == MRI_DERIV_MUL ? v[c] * gg[c] : (gg[c] > 0.f ? v[c] : 0.f);
      }
      *reinterpret_cast<float4*>(dx + r * lddx + k4) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (k4 + c < k) {
          float o = v[c];
          if (mode != MRI_DERIV_NONE) {
            const float gg = g[r * ldg + k4 + c];
            o = mode == MRI_DERIV_MUL ? o * gg : (gg > 0.f ? o : 0.f);
          }
          dx[r * lddx + k4 + c] = o;
        }
    }
  }
}

// ---- K <= 8: dW[n][k] += sum_m dy[m][n] x[m][k], db[n] += sum_m dy[m][n] -------------------------
// The first layer of a coordinate network: the contraction is the whole batch, the output N x 3.
// A 128 x 32 MFMA tile would multiply 29 columns of zeros and streams dy at 2.4 TB/s; here a
// thread owns one column n, a workgroup a range of rows whose (few) x values sit in LDS and are
// read as broadcasts, and dy is read once with row-contiguous loads.
constexpr int kWeightRows = 1024;  // rows per workgroup

__global__ __launch_bounds__(256) void small_k_backward_weight_kernel(
    const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t xrs,
    int64_t xcs, int64_t m, int n, int k, float* __restrict__ d_weight,
    float* __restrict__ d_bias) {
  __shared__ float xs[kWeightRows * kSmallK];
  const int64_t r0 = (int64_t)blockIdx.x * kWeightRows;
  const int rows = (int)min((int64_t)kWeightRows, m - r0);
  for (int e = threadIdx.x; e < rows * k; e += 256) {
    const int r = e / k, kk = e - r * k;
    xs[r * kSmallK + kk] = x[(r0 + r) * xrs + kk * xcs];
  }
  __syncthreads();
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= n) return;
  float acc[kSmallK], sum = 0.f;
#pragma unroll
  for (int kk = 0; kk < kSmallK; ++kk) acc[kk] = 0.f;
  const float* __restrict__ p = dy + r0 * lddy + col;
  int r = 0;
  for (; r + 8 <= rows; r += 8) {
    float d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = p[(int64_t)(r + j) * lddy];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sum += d[j];
#pragma unroll
      for (int kk = 0; kk < kSmallK; ++kk)
        if (kk < k) acc[kk] += d[j] * xs[(r + j) * kSmallK + kk];
    }
  }
  for (; r < rows; ++r) {
    const float d = p[(int64_t)r * lddy];
    sum += d;
#pragma unroll
    for (int kk = 0; kk < kSmallK; ++kk)
      if (kk < k) acc[kk] += d * xs[r * kSmallK + kk];
  }
#pragma unroll
  for (int kk = 0; kk < kSmallK; ++kk)
    if (kk < k) atomicAdd(d_weight + (int64_t)col * k + kk, acc[kk]);
  if (d_bias) atomicAdd(d_bias + col, sum);
}

}  // namespace

// ---- entry points used by linear.hip's extern "C" functions -------------------------------------
bool small_backward_weight(const float* dy, int64_t lddy, const float* x, int64_t xrs, int64_t xcs,
                           int64_t m, int n, int k, float* d_weight, float* d_bias,
                           hipStream_t st) {
  if (k > kSmallK || n < 64) return false;  // narrow outputs stay on the MFMA kernel
  hipLaunchKernelGGL(small_k_backward_weight_kernel,
                     dim3((unsigned)ceil_div(m, kWeightRows), (unsigned)ceil_div(n, 256)),
                     dim3(256), 0, st, dy, lddy, x, xrs, xcs, m, n, k, d_weight, d_bias);
  return true;
}

bool small_forward(const float* x, int64_t xrs, int64_t xcs, const float* w, const float* b,
                   int64_t m, int n, int k, int act, float w0, float* y, int64_t ldy, float* deriv,
                   int64_t ldd, hipStream_t st) {
  if (n <= kSmallN && xcs == 1) {
    const int blocks = (int)std::min<int64_t>(ceil_div(m, 4), 8192);
    hipLaunchKernelGGL(small_n_forward_kernel, dim3(blocks), dim3(256), 0, st, x, xrs, w, b, m, n,
                       k, act, w0, y, ldy, deriv, ldd);
    return true;
  }
  if (k <= kSmallK) {
    hipLaunchKernelGGL(small_k_forward_kernel,
                       dim3((unsigned)ceil_div(m, kRowsPerBlock), (unsigned)ceil_div(n, 256)),
                       dim3(256), 0, st, x, xrs, xcs, w, b, m, n, k, act, w0, y, ldy, deriv, ldd);
    return true;
  }
  return false;
}

bool small_backward_data(const float* dy, int64_t lddy, const float* w, int64_t m, int n, int k,
                         int mode, const float* g, int64_t ldg, float* dx, int64_t lddx,
                         hipStream_t st) {
  if (n > kSmallN) return false;
  hipLaunchKernelGGL(small_n_backward_data_kernel,
                     dim3((unsigned)ceil_div(m, kRowsPerBlock), (unsigned)ceil_div(k, 1024)),
                     dim3(256), 0, st, dy, lddy, w, m, n, k, mode, g, ldg, dx, lddx);
  return true;
}

}  // namespace mri
