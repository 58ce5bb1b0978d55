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
  // (cache-policy bits on these gathers -- sc0, sc1, sc0 sc1 through inline-asm loads -- measured no
  // change, 0.108 ms in every form on one box; nt: 0.303, EXPERIMENTS.md Part II 4.5)
  {
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
  }
  const float mine = xc ? p1 : p0, send = xc ? p0 : p1;
  const float total = mine + __shfl_xor(send, 1, 64);
  if (live) __builtin_nontemporal_store(total, out + (int64_t)level * sl + i * sr + xc * sf);
}

// The same lookup for a CONSUMER THAT RUNS BESIDE IT (the fused decoder kernel of mlp_fused.hip,
// launched on another stream): blocks are ordered slice-major -- a slice = `slice_rows`
// consecutive coordinates = what one round of the consumer's workgroups reads -- and every block
// adds 1 to its slice's counter when its features are visible device-wide.  The consumer polls
// the counter before it touches a slice, so the MFMA-bound decoder starts as soon as the first
// slices are encoded and the rest of the lookup (address-path bound, no LDS, 30 VGPRs) runs in
// the issue slots and the register space the decoder leaves free.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): features leave through
// write-through (sc0 sc1) 16-byte stores -- four neighbouring coordinates of one feature,
// collected with three lane shuffles -- every storing wave drains its stores (vmcnt(0)), the
// workgroup meets at a barrier, ONE lane then adds to the counter at agent scope.  The consumer
// polls relaxed, fences once (agent acquire) and reads with plain loads.
constexpr int kSignalChunks = 8;

template <int D>
__global__ __launch_bounds__(256) void hashgrid_fwd_pair_signal_kernel(
    const LevelTab tab, const Sched sched, int blocks_per_slice, int64_t slice_rows,
    const float* __restrict__ x, int64_t n, const float* __restrict__ table,
    float* __restrict__ out, int64_t ld, unsigned long long* __restrict__ ready) {
  // beside the decoder's MFMA waves (f32 MFMAs run on the vector ALUs) this kernel's ~150 vector
  // instructions per coordinate and level only issue when they win the arbitration
  __builtin_amdgcn_s_setprio(3);
  const int slice = blockIdx.x / blocks_per_slice;
  const int b = blockIdx.x - slice * blocks_per_slice;
  int level, chunk;
  {  // decode() on the block index within the slice
    const int xcd = b & 7, r = b >> 3;
    const int q = r / sched.chunks_per_slot, j = r - q * sched.chunks_per_slot;
    const int slot = xcd + 8 * q;
    level = slot % sched.n_levels;
    const int rank = slot / sched.n_levels;
    const int cnt = (sched.virtual_levels - 1 - level) / sched.n_levels + 1;
    chunk = j * cnt + rank;
  }
  const int64_t row0 = (int64_t)slice * slice_rows;
  const int64_t rows = min(slice_rows, n - row0);
  // a block walks kSignalChunks chunks of 128 coordinates: the drain + barrier + counter update
  // at its end (1-2 us: the write-through stores must be acknowledged) is paid once per
  // kSignalChunks x 128 coordinates -- with one chunk per block it tripled the kernel's time
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  const float* __restrict__ rows_p = table + tab.offset[level] * 2;
  const int xc = threadIdx.x & 1;
  const bool aligned = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  for (int sub = 0; sub < kSignalChunks; ++sub) {
    const int64_t c0 = ((int64_t)chunk * kSignalChunks + sub) * 128;
    if (chunk >= sched.chunks || c0 >= rows) break;  // block-uniform
    const int64_t i = row0 + c0 + (threadIdx.x >> 1);
    const bool live = i < row0 + rows;
    const Cell<D> c = locate<D>(x, live ? i : n - 1, tab.res[level]);
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int nb = 0; nb < (1 << (D - 1)); ++nb) {
      uint32_t h;
      float w;
      corner<D>(c, (nb << 1) | xc, h, w);
      const float2 v = *reinterpret_cast<const float2*>(
          rows_p + (uint64_t)slot_of(h, size, magic, pow2) * 2);
      p0 = p0 + v.x * w;
      p1 = p1 + v.y * w;
    }
    const float mine = xc ? p1 : p0, send = xc ? p0 : p1;
    const float total = mine + __shfl_xor(send, 1, 64);
    // feature-major destination: row (2 level + xc) of `out`, column i
    float* __restrict__ dst = out + ((int64_t)level * 2 + xc) * ld + i;
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v quad;
    quad.x = total;
    quad.y = __shfl_down(total, 2, 64);
    quad.z = __shfl_down(total, 4, 64);
    quad.w = __shfl_down(total, 6, 64);
    const bool quad_inside = aligned && (i & ~int64_t(3)) + 3 < row0 + rows;
    if (quad_inside) {
      if ((i & 3) == 0)  // one lane of four stores the four coordinates' values, 16 bytes
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(quad) : "memory");
    } else if (live) {
      asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(dst), "v"(total) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add(ready + slice, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ------------------------------------------------------------- gradient w.r.t. the coordinates
// The reference keeps the path x -> xf = x res - trunc(x res) -> weights differentiable
// (encoding.py:111-113,120-122: `.detach()` only on the integer part), so a caller that asks for
// d out / d x gets  dx[d] = res_d * sum_levels sum_corners (+-1) prod_{e != d} w_e * <d_out, row>.
// Nothing on the training path needs it (coordinates carry no gradient there): one thread per
// coordinate walks all levels, no tuning beyond coalesced stores.
template <int D, int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_input_kernel(
    const LevelTab tab, int n_levels, const float* __restrict__ x, const float* __restrict__ d_out,
    int64_t n, int64_t sl, int64_t sr, int64_t sf, const float* __restrict__ table,
    float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.0f;
  for (int level = 0; level < n_levels; ++level) {
    const uint32_t size = tab.size[level], magic = tab.magic[level];
    const bool pow2 = tab.pow2[level] != 0;
    const float* __restrict__ rows = table + tab.offset[level] * F;
    const Cell<D> c = locate<D>(x, i, tab.res[level]);
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = d_out[(int64_t)level * sl + i * sr + f * sf];
#pragma unroll kCornerUnroll<D>
    for (int nb = 0; nb < (1 << D); ++nb) {
      uint32_t h;
      float w;
      corner<D>(c, nb, h, w);
      const float* __restrict__ row = rows + (uint64_t)slot_of(h, size, magic, pow2) * F;
      float dot = 0.0f;
#pragma unroll
      for (int f = 0; f < F; ++f) dot += g[f] * row[f];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        float others = 1.0f;
#pragma unroll
        for (int e = 0; e < D; ++e)
          if (e != d) others *= ((nb >> e) & 1) ? c.f[e] : 1.0f - c.f[e];
        const float signed_w = ((nb >> d) & 1) ? others : -others;
        acc[d] += (signed_w * dot) * tab.res[level][d];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) dx[i * D + d] = acc[d];
}

template <int D, int F>
struct BwdInputLaunch {
  static int run(const LevelTab& tab, int n_levels, const float* x, const float* d_out, int64_t n,
                 int64_t sl, int64_t sr, int64_t sf, const float* table, float* dx,
                 hipStream_t st) {
    hipLaunchKernelGGL((hashgrid_bwd_input_kernel<D, F>), dim3((unsigned)ceil_div(n, 256)),
                       dim3(256), 0, st, tab, n_levels, x, d_out, n, sl, sr, sf, table, dx);
    return check_launch("hashgrid_bwd_input_kernel");
  }
};

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

// blocks of one slice of the signalling forward kernel: every (virtual level, chunk slot)
int signal_blocks_per_slice(const Sched& s) { return s.virtual_levels * s.chunks_per_slot; }

template <int D, int F>
struct FwdSignalLaunch {
  static int run(const LevelTab& tab, const Sched& sched, int64_t slice_rows, int slices,
                 const float* x, int64_t n, const float* table, float* out, int64_t ld,
                 unsigned long long* ready, hipStream_t st) {
    if constexpr (F == 2 && D >= 2 && D <= 4) {
      const int per = signal_blocks_per_slice(sched);
      hipLaunchKernelGGL((hashgrid_fwd_pair_signal_kernel<D>), dim3((unsigned)(per * slices)),
                         dim3(256), 0, st, tab, sched, per, slice_rows, x, n, table, out, ld, ready);
      return check_launch("hashgrid_fwd_pair_signal_kernel");
    } else {
      return fail(MRI_ERR_UNSUPPORTED, "signalling forward needs 2 features, 2 <= dim <= 4");
    }
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


extern "C" int64_t mri_hashgrid_forward_signal_blocks(const mri_grid_desc* grid, int64_t slice_rows) {
  if (validate(grid) || slice_rows < 1) return -1;
  if (grid->n_features != 2 || grid->dim < 2 || grid->dim > 4) return -1;
  Sched sched = make_sched(grid->n_levels, slice_rows, 128 * kSignalChunks);
  sched.affinity = 1;
  return signal_blocks_per_slice(sched);
}

extern "C" int mri_hashgrid_forward_signal(const mri_grid_desc* grid, const float* x, int64_t n,
                                           const float* table, float* out, int64_t out_ld,
                                           int64_t slice_rows, uint64_t* ready, void* stream) {
  if (int rc = validate(grid)) return rc;
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  MRI_REQUIRE(slice_rows >= 1, "slice_rows %lld", (long long)slice_rows);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && table && out && ready && out_ld >= n, "NULL device pointer / short leading dimension");
  const LevelTab tab = make_tab(grid);
  Sched sched = make_sched(grid->n_levels, slice_rows, 128 * kSignalChunks);
  sched.affinity = 1;  // the block order inside a slice is the XCD-aware one
  const int64_t slices = ceil_div(n, slice_rows);
  MRI_REQUIRE(slices * signal_blocks_per_slice(sched) < (1ll << 31), "grid too large");
  return dispatch<FwdSignalLaunch>(grid->dim, grid->n_features, tab, sched, slice_rows,
                                   (int)slices, x, n, table, out, out_ld,
                                   reinterpret_cast<unsigned long long*>(ready),
                                   (hipStream_t)stream);
}

extern "C" int mri_hashgrid_backward_input(const mri_grid_desc* grid, const float* x,
                                           const float* d_out, int64_t n,
                                           int64_t dout_level_stride, int64_t dout_row_stride,
                                           int64_t dout_feat_stride, const float* table,
                                           float* d_x, void* stream) {
  if (int rc = validate(grid)) return rc;
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && d_out && table && d_x, "NULL device pointer");
  const LevelTab tab = make_tab(grid);
  return dispatch<BwdInputLaunch>(grid->dim, grid->n_features, tab, grid->n_levels, x, d_out, n,
                                  dout_level_stride, dout_row_stride, dout_feat_stride, table,
                                  d_x, (hipStream_t)stream);
}
