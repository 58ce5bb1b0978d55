// Multiresolution hash-grid encoding for gfx950: forward gather, backward scatter.
//
// Replaces the per-level ATen op chain of the reference (encoding.py:108-128 / 232-270 and
// fast_hash encoding.py:69-78) and its autograd (aten::embedding_dense_backward, SURVEY.md 8a
// row a11).  Arithmetic follows the reference exactly where it is observable:
//   pos = x * res (one f32 multiply), cell = trunc(pos), frac = pos - float(cell),
//   vertex index on axis d = cell or cell+1, weight = prod_d (1-frac | frac) in axis order,
//   slot = (xor_d uint32(vertex_d * PRIME_d)) % T   for EVERY level (no dense fast path, Q9),
//   out = sum over corners (in corner order) of table[slot] * w  (separate multiply and add).
//
// Work decomposition (MI355X-first):
//   forward / atomic backward: one thread per (coordinate, level), 256-thread blocks, and the
//     block -> level map keeps each level on ONE XCD (blockIdx % 8) so that a level's table
//     (<= 4 MiB at T = 2^19, F = 2) is served from that XCD's 4 MiB L2 instead of every L2
//     thrashing over all 40-60 MB of tables.
//   LDS backward: "owner computes" -- a 1024-thread workgroup owns a slice of one level's
//     table as f32 accumulators in LDS (128 KiB), scans the whole batch, and adds only the
//     corners that hash into its slice (ds_add_f32), then writes the slice back with
//     coalesced stores.  Scattered global float atomics run ~17x below the coalesced atomic
//     rate on gfx950 (MI355X_MICROARCH.md, Global float atomics); LDS atomics do not.
#include "common.h"

namespace mri {
namespace {

constexpr uint32_t kPrimes[MRI_MAX_DIM] = {1u,          2654435761u, 805459861u, 3674653429u,
                                           2097192037u, 1434869437u, 2165219737u};

struct LevelTab {
  float res[MRI_MAX_LEVELS][MRI_MAX_DIM + 1];
  uint32_t size[MRI_MAX_LEVELS];
  uint32_t magic[MRI_MAX_LEVELS];  // floor(2^32 / size) when size is not a power of two
  uint32_t pow2[MRI_MAX_LEVELS];   // 1 -> slot = h & (size-1)
  uint64_t offset[MRI_MAX_LEVELS];
};

struct Sched {  // block -> (level, chunk) map, see decode()
  int affinity;
  int n_levels;
  int virtual_levels;  // 8 * ceil(L / 8)
  int chunks_per_slot;
  int chunks;
};

struct PartTab {  // LDS backward: blocks [start[l], start[l+1]) own level-l slices
  int32_t start[MRI_MAX_LEVELS + 1];
  int32_t level_of[MRI_MAX_LEVELS];  // compacted list of levels handled by this launch
  int32_t n_entries;
  int32_t slots_per_block;
};

__device__ __forceinline__ uint32_t slot_of(uint32_t h, uint32_t size, uint32_t magic, bool pow2) {
  if (pow2) return h & (size - 1u);
  // q in {floor(h/size) - 1, floor(h/size)} since magic = floor(2^32/size)
  uint32_t q = __umulhi(h, magic);
  uint32_t r = h - q * size;
  return r >= size ? r - size : r;
}

// corner loops are fully unrolled up to 4-D (16 corners), by 2 beyond
template <int D>
constexpr int kCornerUnroll = D <= 4 ? (1 << D) : 2;

template <int D>
struct Cell {
  uint32_t h0[D];  // hash term of the floor vertex on axis d; the ceil vertex adds the prime
  float f[D];      // fractional position
};

template <int D>
__device__ __forceinline__ Cell<D> locate(const float* __restrict__ x, int64_t i, const float* res) {
  Cell<D> c;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float pos = x[i * D + d] * res[d];
    int cell = (int)pos;  // truncation toward zero, as torch .long()
    c.f[d] = pos - (float)cell;
    c.h0[d] = (uint32_t)cell * kPrimes[d];
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

__device__ __forceinline__ bool decode(const Sched& s, int& level, int& chunk) {
  const int b = blockIdx.x;
  if (s.affinity) {
    const int xcd = b & 7, r = b >> 3;
    const int q = r / s.chunks_per_slot, j = r - q * s.chunks_per_slot;
    const int slot = xcd + 8 * q;
    level = slot % s.n_levels;
    const int rank = slot / s.n_levels;
    const int cnt = (s.virtual_levels - 1 - level) / s.n_levels + 1;
    chunk = j * cnt + rank;
  } else {
    level = b % s.n_levels;
    chunk = b / s.n_levels;
  }
  return chunk < s.chunks;
}

// ---------------------------------------------------------------------------------- forward
template <int D, int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(
    const LevelTab tab, const Sched sched, const float* __restrict__ x, int64_t n,
    const float* __restrict__ table, float* __restrict__ out, int64_t sl, int64_t sr,
    int64_t sf) {
  int level, chunk;
  if (!decode(sched, level, chunk)) return;
  const int64_t i = (int64_t)chunk * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  const float* __restrict__ rows = table + tab.offset[level] * F;
  const Cell<D> c = locate<D>(x, i, tab.res[level]);

  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0f;
#pragma unroll kCornerUnroll<D>
  for (int nb = 0; nb < (1 << D); ++nb) {
    uint32_t h;
    float w;
    corner<D>(c, nb, h, w);
    const float* __restrict__ row = rows + (uint64_t)slot_of(h, size, magic, pow2) * F;
    float v[F];
    if constexpr (F == 2) {
      const float2 t = *reinterpret_cast<const float2*>(row);
      v[0] = t.x, v[1] = t.y;
    } else if constexpr (F == 4) {
      const float4 t = *reinterpret_cast<const float4*>(row);
      v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) v[f] = row[f];
    }
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = acc[f] + v[f] * w;
  }
  float* __restrict__ o = out + (int64_t)level * sl + i * sr;
  if constexpr (F == 2) {
    if (sf == 1 && ((reinterpret_cast<uintptr_t>(o)) & 7) == 0) {
      *reinterpret_cast<float2*>(o) = make_float2(acc[0], acc[1]);
      return;
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) o[f * sf] = acc[f];
}

// ------------------------------------------------------------------- backward, global atomics
template <int D, int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_kernel(
    const LevelTab tab, const Sched sched, const uint32_t level_mask, const float* __restrict__ x,
    const float* __restrict__ d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
    float* __restrict__ d_table) {
  int level, chunk;
  if (!decode(sched, level, chunk)) return;
  if (!((level_mask >> level) & 1u)) return;
  const int64_t i = (int64_t)chunk * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  float* __restrict__ rows = d_table + tab.offset[level] * F;
  const Cell<D> c = locate<D>(x, i, tab.res[level]);
  float g[F];
  const float* __restrict__ gp = d_out + (int64_t)level * sl + i * sr;
#pragma unroll
  for (int f = 0; f < F; ++f) g[f] = gp[f * sf];
#pragma unroll kCornerUnroll<D>
  for (int nb = 0; nb < (1 << D); ++nb) {
    uint32_t h;
    float w;
    corner<D>(c, nb, h, w);
    float* __restrict__ row = rows + (uint64_t)slot_of(h, size, magic, pow2) * F;
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(row + f, g[f] * w);
  }
}

// ------------------------------------------------------------- backward, LDS owner-computes
constexpr int kLdsFloats = 32768;  // 128 KiB of f32 accumulators per workgroup
constexpr int kLdsThreads = 1024;

template <int D, int F>
__global__ __launch_bounds__(kLdsThreads) void hashgrid_bwd_lds_kernel(
    const LevelTab tab, const PartTab parts, const float* __restrict__ x,
    const float* __restrict__ d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
    float* __restrict__ d_table) {
  __shared__ float acc[kLdsFloats];
  const int b = blockIdx.x;
  int e = 0;
  while (e + 1 < parts.n_entries && b >= parts.start[e + 1]) ++e;
  const int level = parts.level_of[e];
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  const uint32_t base = (uint32_t)(b - parts.start[e]) * (uint32_t)parts.slots_per_block;
  const uint32_t count = min((uint32_t)parts.slots_per_block, size - base);

  for (uint32_t s = threadIdx.x; s < count * F; s += kLdsThreads) acc[s] = 0.0f;
  __syncthreads();

  const float* __restrict__ res = tab.res[level];
  const float* __restrict__ gl = d_out + (int64_t)level * sl;
  for (int64_t i = threadIdx.x; i < n; i += kLdsThreads) {
    const Cell<D> c = locate<D>(x, i, res);
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = gl[i * sr + f * sf];
#pragma unroll kCornerUnroll<D>
    for (int nb = 0; nb < (1 << D); ++nb) {
      uint32_t h = 0;
#pragma unroll
      for (int d = 0; d < D; ++d) h ^= ((nb >> d) & 1) ? c.h0[d] + kPrimes[d] : c.h0[d];
      const uint32_t rel = slot_of(h, size, magic, pow2) - base;
      if (rel < count) {
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const float wd = ((nb >> d) & 1) ? c.f[d] : 1.0f - c.f[d];
          w = (d == 0) ? wd : w * wd;
        }
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(&acc[rel * F + f], g[f] * w);
      }
    }
  }
  __syncthreads();
  float* __restrict__ dst = d_table + (tab.offset[level] + base) * F;
  for (uint32_t s = threadIdx.x; s < count * F; s += kLdsThreads) dst[s] += acc[s];
}

// ------------------------------------------------------------------------------ host side
int validate(const mri_grid_desc* g) {
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

LevelTab make_tab(const mri_grid_desc* g) {
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

Sched make_sched(int n_levels, int64_t n) {
  Sched s{};
  s.affinity = options().xcd_affinity;
  s.n_levels = n_levels;
  s.chunks = (int)ceil_div(n, 256);
  const int lpx = (n_levels + 7) / 8;
  s.virtual_levels = 8 * lpx;
  const int min_cnt = s.virtual_levels / n_levels;  // replicas of the least replicated level
  s.chunks_per_slot = (int)ceil_div(s.chunks, min_cnt);
  return s;
}

int64_t grid_blocks(const Sched& s) {
  return s.affinity ? (int64_t)s.virtual_levels * s.chunks_per_slot
                    : (int64_t)s.n_levels * s.chunks;
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

template <int D, int F>
struct FwdLaunch {
  static int run(const LevelTab& tab, const Sched& sched, const float* x, int64_t n,
                 const float* table, float* out, int64_t sl, int64_t sr, int64_t sf,
                 hipStream_t st) {
    hipLaunchKernelGGL((hashgrid_fwd_kernel<D, F>), dim3((unsigned)grid_blocks(sched)), dim3(256),
                       0, st, tab, sched, x, n, table, out, sl, sr, sf);
    return check_launch("hashgrid_fwd_kernel");
  }
};

template <int D, int F>
struct BwdAtomicLaunch {
  static int run(const LevelTab& tab, const Sched& sched, uint32_t mask, const float* x,
                 const float* d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
                 float* d_table, hipStream_t st) {
    hipLaunchKernelGGL((hashgrid_bwd_atomic_kernel<D, F>), dim3((unsigned)grid_blocks(sched)),
                       dim3(256), 0, st, tab, sched, mask, x, d_out, n, sl, sr, sf, d_table);
    return check_launch("hashgrid_bwd_atomic_kernel");
  }
};

template <int D, int F>
struct BwdLdsLaunch {
  static int run(const LevelTab& tab, const PartTab& parts, int blocks, const float* x,
                 const float* d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
                 float* d_table, hipStream_t st) {
    hipLaunchKernelGGL((hashgrid_bwd_lds_kernel<D, F>), dim3((unsigned)blocks),
                       dim3(kLdsThreads), 0, st, tab, parts, x, d_out, n, sl, sr, sf, d_table);
    return check_launch("hashgrid_bwd_lds_kernel");
  }
};

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int mri_hashgrid_forward(const mri_grid_desc* grid, const float* x, int64_t n,
                                    const float* table, float* out, int64_t out_level_stride,
                                    int64_t out_row_stride, int64_t out_feat_stride,
                                    void* stream) {
  if (int rc = validate(grid)) return rc;
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && table && out, "NULL device pointer");
  const LevelTab tab = make_tab(grid);
  const Sched sched = make_sched(grid->n_levels, n);
  MRI_REQUIRE(grid_blocks(sched) < (1ll << 31), "grid too large");
  return dispatch<FwdLaunch>(grid->dim, grid->n_features, tab, sched, x, n, table, out,
                             out_level_stride, out_row_stride, out_feat_stride,
                             (hipStream_t)stream);
}

extern "C" int mri_hashgrid_backward(const mri_grid_desc* grid, const float* x,
                                     const float* d_out, int64_t n, int64_t dout_level_stride,
                                     int64_t dout_row_stride, int64_t dout_feat_stride,
                                     float* d_table, int32_t method, void* stream) {
  if (int rc = validate(grid)) return rc;
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  MRI_REQUIRE(method >= 0 && method <= 2, "method %d not in 0..2", method);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && d_out && d_table, "NULL device pointer");
  const LevelTab tab = make_tab(grid);
  const int F = grid->n_features;

  // Split levels between the LDS owner-computes kernel and the atomic kernel.
  PartTab parts{};
  parts.slots_per_block = kLdsFloats / F;
  uint32_t atomic_mask = 0;
  int blocks = 0;
  for (int l = 0; l < grid->n_levels; ++l) {
    const int need = (int)ceil_div(grid->table_size[l], parts.slots_per_block);
    const bool lds = method == 2 || (method == 0 && need <= options().bwd_lds_max_parts);
    if (lds) {
      parts.level_of[parts.n_entries] = l;
      parts.start[parts.n_entries] = blocks;
      blocks += need;
      parts.n_entries++;
      parts.start[parts.n_entries] = blocks;
    } else {
      atomic_mask |= 1u << l;
    }
  }
  if (blocks > 0) {
    int rc = dispatch<BwdLdsLaunch>(grid->dim, F, tab, parts, blocks, x, d_out, n,
                                    dout_level_stride, dout_row_stride, dout_feat_stride, d_table,
                                    (hipStream_t)stream);
    if (rc) return rc;
  }
  if (atomic_mask) {
    const Sched sched = make_sched(grid->n_levels, n);
    return dispatch<BwdAtomicLaunch>(grid->dim, F, tab, sched, atomic_mask, x, d_out, n,
                                     dout_level_stride, dout_row_stride, dout_feat_stride,
                                     d_table, (hipStream_t)stream);
  }
  return MRI_OK;
}
