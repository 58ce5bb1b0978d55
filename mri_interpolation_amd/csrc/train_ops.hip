// Loss, optimiser and the on-device coordinate-batch producer for gfx950, plus the library's
// small host-side state (last error, tuning options).
//
//   mse_loss      F.mse_loss(y, y_pred) and its gradient          reference models.py:61-66
//   adam_step     torch.optim.Adam defaults, one flat buffer      reference models.py:68-70
//   sample/gather MriImage.__getitem__ + shuffled DataLoader      reference datamodules.py:140-172,198-205
#include <math.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "hashgrid_common.h"  // FinTab: the table gradient's int64 sums, converted by the Adam kernel

namespace mri {

char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

Options& options() {
  static Options o;
  return o;
}

namespace {

// ------------------------------------------------------------------------------------ loss
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ pred,
                                                       const float* __restrict__ target,
                                                       int64_t count, float inv_count,
                                                       float grad_scale,
                                                       float* __restrict__ loss_out,
                                                       float* __restrict__ d_pred) {
  float local = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count;
       e += (int64_t)gridDim.x * 256) {
    const float diff = pred[e] - target[e];
    local += diff * diff;
    if (d_pred) d_pred[e] = diff * grad_scale;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  __shared__ float partial[4];
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(loss_out, (partial[0] + partial[1] + partial[2] + partial[3]) * inv_count);
}

// ------------------------------------------------------------------------------------ Adam
// Same operation order as torch's single-tensor Adam:
//   m = m + (g - m) * (1 - b1)            (Tensor.lerp_)
//   v = v * b2 + (1 - b2) * g * g         (mul_ then addcmul_)
//   p = p + (-step_size * m) / (sqrt(v) / sqrt(bc2) + eps)    (addcdiv_)
// FIN (mri_fused_step): where a segment of `fin` covers an element, its gradient is the table gradient's int64 sum,
// converted here exactly as bin_finalize_kernel would have (hashgrid_common.h) -- that launch then never happens.
template <bool FIN>
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g,
                                          float* __restrict__ m, float* __restrict__ v,
                                          int64_t count, float one_minus_b1, float b2,
                                          float one_minus_b2, float neg_step_size,
                                          float bc2_sqrt, float eps, float grad_scale,
                                          int head, const FinTab& fin) {
  // p/g/m/v point at the first 16-byte aligned element of the range; the `head` (< 4) elements in
  // front of it (a range that starts inside a float4, e.g. a level group of the hash table) are
  // taken one each by the first threads of block 0
  if (blockIdx.x == 0 && (int)threadIdx.x < head) {
    const int64_t e = (int64_t)threadIdx.x - head;
    const float gr = g[e] * grad_scale;
    m[e] = m[e] + (gr - m[e]) * one_minus_b1;
    v[e] = v[e] * b2 + (one_minus_b2 * gr) * gr;
    const float denom = sqrtf(v[e]) / bc2_sqrt + eps;
    p[e] = p[e] + (neg_step_size * m[e]) / denom;
  }
  const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (base >= count) return;
  if (base + 3 < count) {
    float4 pv = *reinterpret_cast<float4*>(p + base);
    // gradient and moments are touched once per step: non-temporal accesses keep them from
    // evicting the parameters (the next forward pass gathers from them) -- 10 us per step
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v g_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g + base));
    if (FIN && base + 3 >= fin.lo && base < fin.hi) {
      for (int s = 0; s < fin.n; ++s) {
        const int64_t b0 = fin.seg[s].begin, b1 = b0 + fin.seg[s].words;
        if (base + 3 < b0 || base >= b1) continue;
        const double inv = fin_inv_scale(fin, s);
        if (base >= b0 && base < b1) g_.x = fin_value(fin, s, base, inv);
        if (base + 1 >= b0 && base + 1 < b1) g_.y = fin_value(fin, s, base + 1, inv);
        if (base + 2 >= b0 && base + 2 < b1) g_.z = fin_value(fin, s, base + 2, inv);
        if (base + 3 >= b0 && base + 3 < b1) g_.w = fin_value(fin, s, base + 3, inv);
      }
    }
    const f4v m_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m + base));
    const f4v v_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v + base));
    const float4 gv4 = make_float4(g_.x, g_.y, g_.z, g_.w);
    float4 mv = make_float4(m_.x, m_.y, m_.z, m_.w);
    float4 vv = make_float4(v_.x, v_.y, v_.z, v_.w);
    float* pp = &pv.x;
    const float* gp = &gv4.x;
    float* mp = &mv.x;
    float* vp = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gp[e] * grad_scale;
      mp[e] = mp[e] + (gr - mp[e]) * one_minus_b1;
      vp[e] = vp[e] * b2 + (one_minus_b2 * gr) * gr;
      const float denom = sqrtf(vp[e]) / bc2_sqrt + eps;
      pp[e] = pp[e] + (neg_step_size * mp[e]) / denom;
    }
    *reinterpret_cast<float4*>(p + base) = pv;
    f4v mo, vo;
    mo.x = mv.x, mo.y = mv.y, mo.z = mv.z, mo.w = mv.w;
    vo.x = vv.x, vo.y = vv.y, vo.z = vv.z, vo.w = vv.w;
    __builtin_nontemporal_store(mo, reinterpret_cast<f4v*>(m + base));
    __builtin_nontemporal_store(vo, reinterpret_cast<f4v*>(v + base));
  } else {
    for (int64_t e = base; e < count; ++e) {
      float ge = g[e];
      if (FIN && e >= fin.lo && e < fin.hi)
        for (int s = 0; s < fin.n; ++s)
          if (e >= fin.seg[s].begin && e < fin.seg[s].begin + fin.seg[s].words)
            ge = fin_value(fin, s, e, fin_inv_scale(fin, s));
      const float gr = ge * grad_scale;
      m[e] = m[e] + (gr - m[e]) * one_minus_b1;
      v[e] = v[e] * b2 + (one_minus_b2 * gr) * gr;
      const float denom = sqrtf(v[e]) / bc2_sqrt + eps;
      p[e] = p[e] + (neg_step_size * m[e]) / denom;
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int64_t count, float one_minus_b1, float b2,
                                                   float one_minus_b2, float neg_step_size,
                                                   float bc2_sqrt, float eps, float grad_scale,
                                                   int head) {
  adam_body<false>(p, g, m, v, count, one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps, grad_scale, head,
                   FinTab{});
}
__global__ __launch_bounds__(256) void adam_fin_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       int64_t count, float one_minus_b1, float b2,
                                                       float one_minus_b2, float neg_step_size,
                                                       float bc2_sqrt, float eps, float grad_scale,
                                                       const FinTab fin) {
  adam_body<true>(p, g, m, v, count, one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps, grad_scale, 0,
                  fin);
}

// ---------------------------------------------------------------------------------- sampler
// Keyed bijection of [0, range): a 4-round Feistel network on the smallest even-width bit
// field that covers `range`, with cycle walking for values that fall outside.  One key =
// one shuffle of the data set (what DataLoader(shuffle=True) draws per epoch).
__device__ __forceinline__ uint32_t mix32(uint32_t x, uint32_t k) {
  x ^= k;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint64_t feistel(uint64_t v, int half_bits, uint64_t key) {
  const uint64_t half_mask = (1ull << half_bits) - 1;
  uint32_t left = (uint32_t)(v >> half_bits), right = (uint32_t)(v & half_mask);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t k = (uint32_t)(key >> (16 * r)) ^ (0x9e3779b9u * (r + 1));
    const uint32_t f = mix32(right, k) & (uint32_t)half_mask;
    const uint32_t nl = right;
    right = left ^ f;
    left = nl;
  }
  return ((uint64_t)left << half_bits) | right;
}

__global__ __launch_bounds__(256) void sample_kernel(uint64_t key, int64_t first, int64_t lo,
                                                     int64_t range, int half_bits, int64_t n,
                                                     int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t v = (uint64_t)((first + i) % range);
  do {
    v = feistel(v, half_bits, key);
  } while (v >= (uint64_t)range);  // cycle walking: the field is < 4x range, so ~1-4 rounds
  out[i] = lo + (int64_t)v;
}

struct ShapeTab {
  int64_t shape[MRI_MAX_DIM];
  int64_t axis_offset[MRI_MAX_DIM];
};

__global__ __launch_bounds__(256) void gather_batch_kernel(const int64_t* __restrict__ idx,
                                                           int64_t n, int dim, ShapeTab tab,
                                                           const float* __restrict__ axes,
                                                           const float* __restrict__ volume,
                                                           float* __restrict__ coords,
                                                           float* __restrict__ target) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t flat = idx[i];
  if (target) target[i] = volume[flat];
  int64_t rest = flat;
  for (int d = dim - 1; d >= 0; --d) {  // C order: last axis fastest (datamodules.py:162-163)
    const int64_t pos = rest % tab.shape[d];
    rest /= tab.shape[d];
    coords[i * dim + d] = axes[tab.axis_offset[d] + pos];
  }
}

}  // namespace

int adam_step_fin(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, double lr,
                  double beta1, double beta2, double eps, int32_t step, float grad_scale, const FinTab& fin,
                  hipStream_t stream) {
  MRI_REQUIRE(count >= 1 && step >= 1 && param && grad && exp_avg && exp_avg_sq, "adam_step_fin: bad arguments");
  MRI_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
              "adam_step_fin: 16-byte aligned buffers");
  const double bc1 = 1.0 - pow(beta1, (double)step);  // scalar prefactors in double, as torch computes them on the host
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const int64_t blocks = std::max<int64_t>(1, ceil_div(ceil_div(count, 4), 256));
  hipLaunchKernelGGL(adam_fin_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq,
                     count, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(-(lr / bc1)),
                     (float)sqrt(bc2), (float)eps, grad_scale, fin);
  return check_launch("adam_fin_kernel");
}
}  // namespace mri

using namespace mri;

extern "C" const char* mri_version(void) { return "mri_inr 0.1.0 gfx950"; }
extern "C" const char* mri_last_error(void) { return error_buffer(); }

extern "C" int mri_set_option(const char* name, int32_t value) {
  MRI_REQUIRE(name != nullptr, "option name is NULL");
  if (!strcmp(name, "xcd_affinity")) {
    options().xcd_affinity = value != 0;
  } else if (!strcmp(name, "bwd_lds_max_parts")) {
    options().bwd_lds_max_parts = value;
  } else if (!strcmp(name, "bwd_fuse_dense")) {
    options().bwd_fuse_dense = value != 0;
  } else if (!strcmp(name, "bwd_dense_blocks")) {
    options().bwd_dense_blocks = value < 1 ? 1 : value;
  } else if (!strcmp(name, "bwd_dense_max_parts")) {
    options().bwd_dense_max_parts = value;
  } else if (!strcmp(name, "fwd_pair")) {
    options().fwd_pair = value != 0;
  } else if (!strcmp(name, "mlp_x3")) {
    options().mlp_x3 = value;
  } else if (!strcmp(name, "mlp_stagger")) {
    options().mlp_stagger = value;
  } else if (!strcmp(name, "bwd_records")) {
    MRI_REQUIRE(value == 0 || value == 1, "bwd_records %d not in {0, 1}", value);
    options().bwd_records = value;
  } else if (!strcmp(name, "siren_rows")) {
    options().siren_rows = value != 0;
  } else if (!strcmp(name, "bwd_blocks_per_level")) {
    options().bwd_blocks_per_level = value < 1 ? 1 : value;
  } else {
    return fail(MRI_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
  }
  return MRI_OK;
}

extern "C" int mri_mse_loss(const float* pred, const float* target, int64_t count,
                            float grad_divisor, float* loss_out, float* d_pred, void* stream) {
  MRI_REQUIRE(count >= 0, "negative count");
  MRI_REQUIRE(grad_divisor > 0.f, "grad_divisor must be positive");
  if (count == 0) return MRI_OK;
  MRI_REQUIRE(pred && target && loss_out, "NULL device pointer");
  const int blocks = (int)std::min<int64_t>(ceil_div(count, 256), 2048);
  const float inv = (float)(1.0 / (double)count);
  const float gs = (float)(2.0 / ((double)count * (double)grad_divisor));
  hipLaunchKernelGGL(mse_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred,
                     target, count, inv, gs, loss_out, d_pred);
  return check_launch("mse_loss_kernel");
}

extern "C" int mri_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             int64_t count, double lr, double beta1, double beta2,
                             double eps, int32_t step, float grad_scale, void* stream) {
  MRI_REQUIRE(count >= 0 && step >= 1, "bad count/step");
  if (count == 0) return MRI_OK;
  MRI_REQUIRE(param && grad && exp_avg && exp_avg_sq, "NULL device pointer");
  // any 4-byte aligned range of the flat buffers, as long as the four pointers sit at the same
  // offset inside a 16-byte line (they do when they are the same slice of four aligned buffers)
  const uintptr_t mis = reinterpret_cast<uintptr_t>(param) & 15;
  MRI_REQUIRE((mis & 3) == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == mis &&
                  (reinterpret_cast<uintptr_t>(exp_avg) & 15) == mis &&
                  (reinterpret_cast<uintptr_t>(exp_avg_sq) & 15) == mis,
              "Adam buffers must share one 4-byte aligned offset within a 16-byte line");
  const int head = (int)std::min<int64_t>(count, mis ? (int64_t)(16 - mis) / 4 : 0);
  param += head, grad += head, exp_avg += head, exp_avg_sq += head, count -= head;
  // scalar prefactors in double, as torch computes them on the host
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float neg_step_size = (float)(-(lr / bc1));
  const float bc2_sqrt = (float)sqrt(bc2);
  const int64_t blocks = std::max<int64_t>(1, ceil_div(ceil_div(count, 4), 256));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     param, grad, exp_avg, exp_avg_sq, count, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), neg_step_size, bc2_sqrt, (float)eps, grad_scale, head);
  return check_launch("adam_kernel");
}

namespace {
uint64_t sample_key(uint64_t seed) {  // scramble the user seed so that nearby seeds give unrelated keys
  uint64_t key = seed + 0x9E3779B97F4A7C15ull;
  key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
  key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
  key ^= key >> 31;
  return key;
}
}  // namespace

extern "C" int mri_sample_indices(uint64_t seed, int64_t first, int64_t lo, int64_t hi,
                                  int64_t n, int64_t* idx_out, void* stream) {
  MRI_REQUIRE(hi > lo && first >= 0 && n >= 0, "bad range [%lld, %lld)", (long long)lo,
              (long long)hi);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(idx_out, "NULL device pointer");
  const int64_t range = hi - lo;
  int bits = 2;
  while ((1ll << bits) < range) ++bits;
  if (bits & 1) ++bits;
  MRI_REQUIRE(bits <= 62, "range too large");
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, sample_key(seed), first, lo, range, bits / 2, n, idx_out);
  return check_launch("sample_kernel");
}

extern "C" int mri_gather_batch(const int64_t* idx, int64_t n, int32_t dim, const int64_t* shape,
                                const float* axes, const int64_t* axis_offset,
                                const float* volume, float* coords_out, float* target_out,
                                void* stream) {
  MRI_REQUIRE(n >= 0 && dim >= 1 && dim <= MRI_MAX_DIM, "bad n/dim");
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(idx && shape && axes && axis_offset && coords_out, "NULL pointer");
  MRI_REQUIRE(!target_out || volume, "target_out needs a volume");
  ShapeTab tab{};
  for (int d = 0; d < dim; ++d) {
    MRI_REQUIRE(shape[d] >= 1, "shape[%d] < 1", d);
    tab.shape[d] = shape[d];
    tab.axis_offset[d] = axis_offset[d];
  }
  hipLaunchKernelGGL(gather_batch_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, idx, n, (int)dim, tab, axes, volume, coords_out,
                     target_out);
  return check_launch("gather_batch_kernel");
}
