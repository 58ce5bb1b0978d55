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
//   forward / atomic backward: one thread (forward, F = 2: two lanes) per (coordinate, level),
//     256-thread blocks, and the
//     block -> level map keeps each level on ONE XCD (blockIdx % 8) so that a level's table
//     (<= 4 MiB at T = 2^19, F = 2) is served from that XCD's 4 MiB L2 instead of every L2
//     thrashing over all 40-60 MB of tables.
//   the table gradient proper lives in hashgrid_bwd.hip (binning + fixed-point LDS accumulation);
//     the global-atomic kernel here is its fallback for levels too large to bin, D > 4 or F > 4,
//     and the cross-check of the tests (method 1).  Scattered global float atomics run ~17x below
//     the coalesced atomic rate on gfx950 (MI355X_MICROARCH.md, Global float atomics): 3.6 ms for
//     BASELINE config 4, against 0.23 ms for the binned path.
#include "hashgrid_common.h"

namespace mri {
namespace {

struct Sched {  // block -> (level, chunk) map, see decode()
  int affinity;
  int n_levels;
  int virtual_levels;  // 8 * ceil(L / 8)
  int chunks_per_slot;
  int chunks;
};

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
  if constexpr (F == 2) {
    // The two corners that differ only on axis 0 (PRIME_0 = 1) hash to h and h ^ (x ^ (x+1)):
    // for an even cell index the slots are 2k and 2k+1, i.e. one aligned 16-byte pair of rows.
    // The kernel is bound by the number of divergent lanes the address unit has to walk
    // (77 % of wave cycles in vmem issue stalls), so fetch such pairs with ONE load.
#pragma unroll kCornerUnroll<D>
    for (int nb = 0; nb < (1 << D); nb += 2) {
      uint32_t h0, h1;
      float w0, w1;
      corner<D>(c, nb, h0, w0);
      corner<D>(c, nb + 1, h1, w1);
      const uint32_t s0 = slot_of(h0, size, magic, pow2), s1 = slot_of(h1, size, magic, pow2);
      float2 v0, v1;
      if ((s0 ^ 1u) == s1) {
        const float4 t = *reinterpret_cast<const float4*>(rows + (uint64_t)(s0 & ~1u) * 2);
        const float2 lo = make_float2(t.x, t.y), hi = make_float2(t.z, t.w);
        v0 = (s0 & 1u) ? hi : lo;
        v1 = (s0 & 1u) ? lo : hi;
      } else {
        v0 = *reinterpret_cast<const float2*>(rows + (uint64_t)s0 * 2);
        v1 = *reinterpret_cast<const float2*>(rows + (uint64_t)s1 * 2);
      }
      acc[0] = acc[0] + v0.x * w0;
      acc[1] = acc[1] + v0.y * w0;
      acc[0] = acc[0] + v1.x * w1;
      acc[1] = acc[1] + v1.y * w1;
    }
  } else {
#pragma unroll kCornerUnroll<D>
    for (int nb = 0; nb < (1 << D); ++nb) {
      uint32_t h;
      float w;
      corner<D>(c, nb, h, w);
      const float* __restrict__ row = rows + (uint64_t)slot_of(h, size, magic, pow2) * F;
      float v[F];
      if constexpr (F == 4) {
        const float4 t = *reinterpret_cast<const float4*>(row);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
      } else {
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = row[f];
      }
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = acc[f] + v[f] * w;
    }
  }
  float* __restrict__ o = out + (int64_t)level * sl + i * sr;
  if constexpr (F == 2) {
    if (sf == 1 && ((reinterpret_cast<uintptr_t>(o)) & 7) == 0) {
      typedef float f2v __attribute__((ext_vector_type(2)));
      f2v v2;
      v2.x = acc[0], v2.y = acc[1];
      __builtin_nontemporal_store(v2, reinterpret_cast<f2v*>(o));
      return;
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) {
    // written once, read by another kernel: keep it out of the L2 that holds the tables
    __builtin_nontemporal_store(acc[f], o + f * sf);
  }
}

// ------------------------------------------------------------------- backward, global atomics
// F = 2, two lanes per (coordinate, level): lane 2i takes the corners with the LOWER vertex on
// axis 0, lane 2i+1 those with the upper one.  PRIME_0 = 1, so the two slots of such a pair are
// h and h ^ (x ^ (x+1)): they differ in the low bits only and share a 128-byte line unless
// x = 15 (mod 16).  The kernel is bound by L1 line fills (one 128-byte fill per divergent lane
// for 8 useful bytes); as neighbours of ONE load instruction the two requests are served by one
// fill.  Each lane ends up owning one feature (its partial sum + the neighbour's, one DPP swap).
template <int D>
__global__ __launch_bounds__(256) void hashgrid_fwd_pair_kernel(
    const LevelTab tab, const Sched sched, const float* __restrict__ x, int64_t n,
    const float* __restrict__ table, float* __restrict__ out, int64_t sl, int64_t sr,
    int64_t sf) {
  int level, chunk;
  if (!decode(sched, level, chunk)) return;
  const int64_t i = (int64_t)chunk * 128 + (threadIdx.x >> 1);
  const int xc = threadIdx.x & 1;
  const bool live = i < n;
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  const float* __restrict__ rows = table + tab.offset[level] * 2;
  const Cell<D> c = locate<D>(x, live ? i : n - 1, tab.res[level]);
  float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
  for (int nb = 0; nb < (1 << (D - 1)); ++nb) {
    uint32_t h;
    float w;
    corner<D>(c, (nb << 1) | xc, h, w);
    const float2 v = *reinterpret_cast<const float2*>(
        rows + (uint64_t)slot_of(h, size, magic, pow2) * 2);
    p0 = p0 + v.x * w;
    p1 = p1 + v.y * w;
  }
  const float mine = xc ? p1 : p0, send = xc ? p0 : p1;
  const float total = mine + __shfl_xor(send, 1, 64);
  if (live) __builtin_nontemporal_store(total, out + (int64_t)level * sl + i * sr + xc * sf);
}

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

Sched make_sched(int n_levels, int64_t n, int coords_per_block = 256) {
  Sched s{};
  s.affinity = options().xcd_affinity;
  s.n_levels = n_levels;
  s.chunks = (int)ceil_div(n, coords_per_block);
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

template <int D, int F>
struct FwdLaunch {
  static int run(const LevelTab& tab, const Sched& sched, const float* x, int64_t n,
                 const float* table, float* out, int64_t sl, int64_t sr, int64_t sf,
                 hipStream_t st) {
    if constexpr (F == 2 && D >= 2) {
      if (options().fwd_pair) {
        const Sched ps = make_sched(sched.n_levels, n, 128);
        hipLaunchKernelGGL((hashgrid_fwd_pair_kernel<D>), dim3((unsigned)grid_blocks(ps)),
                           dim3(256), 0, st, tab, ps, x, n, table, out, sl, sr, sf);
        return check_launch("hashgrid_fwd_pair_kernel");
      }
    }
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

}  // namespace

int launch_backward_atomic(const mri_grid_desc* grid, uint32_t level_mask, const float* x,
                           const float* d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
                           float* d_table, hipStream_t st) {
  const LevelTab tab = make_tab(grid);
  const Sched sched = make_sched(grid->n_levels, n);
  return dispatch<BwdAtomicLaunch>(grid->dim, grid->n_features, tab, sched, level_mask, x, d_out,
                                   n, sl, sr, sf, d_table, st);
}

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

