// Pieces shared by the hash-grid forward (hashgrid.hip) and backward (hashgrid_bwd.hip) kernels.
#pragma once
#include "common.h"

namespace mri {

constexpr uint32_t kPrimes[MRI_MAX_DIM] = {1u,          2654435761u, 805459861u, 3674653429u,
                                           2097192037u, 1434869437u, 2165219737u};

struct LevelTab {
  float res[MRI_MAX_LEVELS][MRI_MAX_DIM + 1];
  uint32_t size[MRI_MAX_LEVELS];
  uint32_t magic[MRI_MAX_LEVELS];  // floor(2^32 / size) when size is not a power of two
  uint32_t pow2[MRI_MAX_LEVELS];   // 1 -> slot = h & (size-1)
  uint64_t offset[MRI_MAX_LEVELS];
};

__device__ __forceinline__ uint32_t slot_of(uint32_t h, uint32_t size, uint32_t magic, bool pow2) {
  if (pow2) return h & (size - 1u);
  // q in {floor(h/size) - 1, floor(h/size)} since magic = floor(2^32/size)
  uint32_t q = __umulhi(h, magic);
  uint32_t r = h - q * size;
  return r >= size ? r - size : r;
}

// Exponent of the 64-bit fixed point a level's table gradient is summed in (hashgrid_bwd.hip): e with
// n * max|g| * 2^e < 2^61: max|g| < 2^(E+1) (E = unbiased exponent of the level's max |d_out|), n < 2^log_n
__device__ __forceinline__ int level_exponent(uint32_t max_bits, int64_t n) {
  const int E = (int)((max_bits >> 23) & 255u) - 127;
  const int log_n = 64 - __clzll((unsigned long long)n);
  const int e = 61 - (E + 1) - log_n;
  return max(-90, min(e, 120));
}

// Levels whose gradient sums meet in the int64 area of the table gradient's workspace (the dense levels, binned
// levels cut over entry ranges) are converted to f32 by bin_finalize_kernel -- or, when the optimizer step follows
// at once (mri_fused_step), by the Adam kernel itself as it fetches the gradient: the 8 us launch between the
// accumulation and Adam goes.  One segment per such level, in elements of the buffer the consumer walks.
struct FinSeg {
  int64_t begin, words;
  const unsigned long long* src;
  int level, pad;
};
struct FinTab {
  int n = 0, pad = 0;
  int64_t lo = 0, hi = 0;            // [lo, hi) covers every segment
  const uint32_t* max_bits = nullptr;  // per level: bit pattern of max |d_out|
  int64_t n_coords = 0;
  FinSeg seg[MRI_MAX_LEVELS];
};
// the f32 gradient of element e (inside segment s), as bin_finalize_kernel computes it
__device__ __forceinline__ double fin_inv_scale(const FinTab& f, int s) {
  return __builtin_ldexp(1.0, -level_exponent(f.max_bits[f.seg[s].level], f.n_coords));
}
__device__ __forceinline__ float fin_value(const FinTab& f, int s, int64_t e, double inv_scale) {
  return (float)((double)(long long)f.seg[s].src[e - f.seg[s].begin] * inv_scale);
}
// hashgrid_bwd.hip: the next table-gradient call on this thread that would launch bin_finalize_kernel (overwrite
// mode) fills *out instead (segments in elements of d_table) and leaves the conversion to the caller
void set_finalize_export(FinTab* out);
// train_ops.hip: mri_adam_step over a 16-byte aligned range whose gradient comes from `fin` where a segment covers
// it (segments in elements of the range), from `grad` elsewhere
int adam_step_fin(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, double lr,
                  double beta1, double beta2, double eps, int32_t step, float grad_scale, const FinTab& fin,
                  hipStream_t stream);

// corner loops are fully unrolled up to 4-D (16 corners), by 2 beyond
template <int D>
constexpr int kCornerUnroll = D <= 4 ? (1 << D) : 2;

template <int D>
struct Cell {
  uint32_t h0[D];  // hash term of the floor vertex on axis d; the ceil vertex adds the prime
  float f[D];      // fractional position
};

// Cell index of a scaled coordinate: truncation toward zero as torch .long() (int64), of which
// the hash keeps the low 32 bits (reference encoding.py:73: `(ind * prime) & 0xFFFFFFFF`).  Below
// 2^31 that is the hardware float -> int32 conversion; a per-axis resolution may grow far beyond
// that (V2's growth exponent), where the saturating conversion would differ: such
// positions are integer-valued floats, converted through int64 on a path no usual level takes.
__device__ __forceinline__ uint32_t cell_low32(float pos) {
  if (__builtin_expect(fabsf(pos) < 2147483648.0f, 1)) return (uint32_t)(int)pos;
  return (uint32_t)(unsigned long long)(long long)pos;
}

template <int D>
__device__ __forceinline__ Cell<D> locate(const float* __restrict__ x, int64_t i, const float* res) {
  Cell<D> c;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float pos = x[i * D + d] * res[d];
    // pos - float(cell): for |pos| >= 2^24 the position is its own cell (fraction 0)
    c.f[d] = fabsf(pos) < 16777216.0f ? pos - (float)(int)pos : 0.0f;
    c.h0[d] = cell_low32(pos) * kPrimes[d];
  }
  return c;
}

template <int D>
__device__ __forceinline__ void corner(const Cell<D>& c, int n, uint32_t& h, float& w) {
  h = 0;
  w = 1.0f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const bool hi = (n >> d) & 1;
    h ^= hi ? c.h0[d] + kPrimes[d] : c.h0[d];
    const float wd = hi ? c.f[d] : 1.0f - c.f[d];
    w = (d == 0) ? wd : w * wd;
  }
}

inline int validate(const mri_grid_desc* g) {
  MRI_REQUIRE(g != nullptr, "grid descriptor is NULL");
  MRI_REQUIRE(g->dim >= 1 && g->dim <= MRI_MAX_DIM, "dim %d not in 1..%d", g->dim, MRI_MAX_DIM);
  MRI_REQUIRE(g->n_levels >= 1 && g->n_levels <= MRI_MAX_LEVELS, "n_levels %d not in 1..%d",
              g->n_levels, MRI_MAX_LEVELS);
  MRI_REQUIRE(g->n_features == 1 || g->n_features == 2 || g->n_features == 4 ||
                  g->n_features == 8,
              "n_features %d not in {1,2,4,8}", g->n_features);
  for (int l = 0; l < g->n_levels; ++l)
    MRI_REQUIRE(g->table_size[l] >= 1 && g->table_size[l] <= (1u << 30),
                "table_size[%d] = %u not in 1..2^30", l, g->table_size[l]);
  return MRI_OK;
}

inline LevelTab make_tab(const mri_grid_desc* g) {
  LevelTab t{};
  for (int l = 0; l < g->n_levels; ++l) {
    for (int d = 0; d < g->dim; ++d) t.res[l][d] = g->resolution[l][d];
    const uint32_t s = g->table_size[l];
    t.size[l] = s;
    t.pow2[l] = (s & (s - 1)) == 0;
    t.magic[l] = t.pow2[l] ? 0u : (uint32_t)((1ull << 32) / s);
    t.offset[l] = g->table_offset[l];
  }
  return t;
}

template <template <int, int> class Launch, typename... Args>
int dispatch(int dim, int feats, Args&&... args) {
#define MRI_CASE_F(DD)                                                   \
  switch (feats) {                                                       \
    case 1: return Launch<DD, 1>::run(args...);                          \
    case 2: return Launch<DD, 2>::run(args...);                          \
    case 4: return Launch<DD, 4>::run(args...);                          \
    case 8: return Launch<DD, 8>::run(args...);                          \
  }                                                                      \
  break;
  switch (dim) {
    case 1: MRI_CASE_F(1)
    case 2: MRI_CASE_F(2)
    case 3: MRI_CASE_F(3)
    case 4: MRI_CASE_F(4)
    case 5: MRI_CASE_F(5)
    case 6: MRI_CASE_F(6)
    case 7: MRI_CASE_F(7)
  }
#undef MRI_CASE_F
  return fail(MRI_ERR_UNSUPPORTED, "no kernel for dim %d, n_features %d", dim, feats);
}

// global-atomic backward for the levels selected by `level_mask` (hashgrid.hip)
int launch_backward_atomic(const mri_grid_desc* grid, uint32_t level_mask, const float* x,
                           const float* d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
                           float* d_table, hipStream_t st);

}  // namespace mri
