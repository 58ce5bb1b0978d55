// Spatial order of a training batch (mri_order_batch): which voxels a batch holds is the shuffle's business
// (mri_sample_indices; reference datamodules.py:198-205, DataLoader(shuffle=True)); the ORDER of its rows is free --
// the loss is a mean over the batch, the table gradient a sum -- and the kernels care:
//
//   * rows sorted along a Morton (Z) curve put neighbouring voxels next to each other: the lookup finds the
//     cache lines of the coarse and middle levels again a moment later (BASELINE config 4: 0.099 -> 0.078 ms) --
//     but the 64 lanes of a wave then add into the SAME few slots of the coarse levels' gradient, and the
//     LDS atomics of the dense-level pass serialise (table gradient 0.22 -> 0.39 ms, all of it on levels 0-2:
//     tools/sorted_batch_probe.py);
//   * so the sorted rows are TRANSPOSED inside blocks of 16384: a wave takes every 256th row of a block (64
//     different neighbourhoods: no equal slots inside a wave, as in a shuffled batch), consecutive waves take
//     their Morton neighbours (the same lines, a moment later, on the same CU).  Measured on the same batch:
//     lookup 0.099 -> 0.090 ms, table gradient 0.222 -> 0.214 ms (its scatter writes longer runs);
//   * the 12 highest bits of the Morton key carry that gain (4096 cells; `--coarse` of the probe: 9 bits 0.0947 /
//     0.2199 ms, 12 bits 0.0908 / 0.2119, all 24 bits 0.0899 / 0.2107), so the sort is ONE stable counting pass
//     over 4096 buckets: rows keep their shuffle order inside a bucket.
//
// Four launches (count per 1024-row chunk -> per-bucket scan -> bucket bases -> place), deterministic: a row's rank
// inside its bucket comes from chunk, wave and lane order, never from the order atomics happen in.
//
// State at the end of round 3: an OPTION (BatchPipeline(order="morton"), bench.py --batch-order morton), not the
// default.  On ordered batches every kernel of BASELINE config 4's step is faster under rocprofv3 (lookup 105.1 ->
// 96.6 us, scatter 75.6 -> 74.5, dense + accumulate 86.6 -> 82.7, count 60.2 -> 50.7, gather 20.1 -> 10.7) and the
// step is not: the ordering is 32 us of GPU work per step (this file, alone on the GPU) that has nowhere to hide.
// Beside the lookup -- whose workgroups hold every wave slot, so that a side kernel gets CU time only as they retire --
// the four launches stretch to ~280 us, the counting stage of the table gradient lands behind the decoder and beside
// the scatter: 0.535 against 0.512 ms (the lookup itself, alone at last: 0.0965 instead of 0.108 ms).  Produced two
// batches ahead, behind everything the next step waits for, the ordering ran beside the table gradient: 0.522
// against 0.515 (that form is in the history, commit "one stable counting pass").  Earlier sorts:
// rocprim::radix_sort_pairs takes its merge-sort path for 2^18 pairs (17 launches, ~120 us: 0.536 ms; it also loses
// values when the sorted bit window ends at bit 32 of a 32-bit key, tools/probes/rocprim_sort_probe.hip), a two-pass
// 8-bit radix sort of six naive launches (0.59 ms), this pass with its scan in ONE workgroup (0.529: one CU's ~10
// bytes per cycle over a 6 MB table).
#include <hip/hip_runtime.h>

#include "common.h"

namespace mri {
namespace {

constexpr int kBlockRows = 16384;  // rows of a transposition block (a multiple of 64)
constexpr int kChunk = 1024;       // rows (= threads) of a workgroup of the count and place kernels
constexpr int kKeyBits = 12;
constexpr int kBuckets = 1 << kKeyBits;
constexpr int kWaves = kChunk / 64;

struct OrderShape {
  int64_t shape[MRI_MAX_DIM];
  int64_t axis_offset[MRI_MAX_DIM];  // (the gathering form: start of axis d in `axes`)
};

// Morton key: up to 8 bits per axis (position / extent, so that axes of different lengths weigh the same),
// interleaved with the LAST axis in the lowest bit; up to 4 axes (more: the first four decide).  The bucket is its
// kKeyBits highest bits (fewer key bits than that -- one axis: all of them).
__device__ __forceinline__ uint32_t bucket_of(int64_t flat, int dim, const OrderShape& s, bool small) {
  uint32_t q[MRI_MAX_DIM];
  if (small) {  // fewer than 2^24 positions per axis and 2^32 voxels: 32-bit divisions (wave-uniform branch)
    uint32_t rest = (uint32_t)flat;
    for (int d = dim - 1; d >= 0; --d) {  // C order: last axis fastest (as mri_gather_batch)
      const uint32_t extent = (uint32_t)s.shape[d], pos = rest % extent;
      rest /= extent;
      q[d] = (pos << 8) / extent;
    }
  } else {
    int64_t rest = flat;
    for (int d = dim - 1; d >= 0; --d) {
      const int64_t pos = rest % s.shape[d];
      rest /= s.shape[d];
      q[d] = (uint32_t)((pos << 8) / s.shape[d]);
    }
  }
  const int used = dim < 4 ? dim : 4;
  uint32_t k = 0;
  for (int b = 7; b >= 0; --b)
    for (int d = 0; d < used; ++d) k = (k << 1) | ((q[d] >> b) & 1u);
  const int bits = 8 * used;
  return bits > kKeyBits ? k >> (bits - kKeyBits) : k;
}

// 1. per chunk of 1024 rows: every row's bucket and its rank among the chunk's rows of the same bucket (in row
//    order), the chunk's count per bucket, and a copy of the indices (the place kernel writes `idx` in place)
__global__ __launch_bounds__(kChunk) void order_count_kernel(const int64_t* __restrict__ idx, int64_t n, int dim,
                                                             OrderShape s, int small, int64_t* __restrict__ idx_copy,
                                                             uint16_t* __restrict__ bucket, uint16_t* __restrict__ rank,
                                                             uint16_t* __restrict__ chunk_count) {
  __shared__ uint8_t wave_cnt[kWaves][kBuckets];  // rows of each wave per bucket (<= 64)
  {
    uint4* z = reinterpret_cast<uint4*>(&wave_cnt[0][0]);
    for (int e = threadIdx.x; e < kWaves * kBuckets / 16; e += kChunk) z[e] = uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kChunk + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = i < n;
  uint32_t b = 0;
  if (live) {
    const int64_t v = idx[i];
    idx_copy[i] = v;
    b = bucket_of(v, dim, s, small != 0);
  }
  unsigned long long peers = __ballot(live);  // lanes of this wave in the same bucket: one ballot per key bit
#pragma unroll
  for (int k = 0; k < kKeyBits; ++k) {
    const unsigned long long set = __ballot(live && ((b >> k) & 1u));
    peers &= ((b >> k) & 1u) ? set : ~set;
  }
  const int in_wave = __popcll(peers & ((1ull << lane) - 1ull));
  if (live && in_wave == 0) wave_cnt[wave][b] = (uint8_t)__popcll(peers);
  __syncthreads();
  if (live) {
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w][b];
    bucket[i] = (uint16_t)b;
    rank[i] = (uint16_t)(before + in_wave);
  }
  uint16_t* __restrict__ row = chunk_count + (int64_t)blockIdx.x * kBuckets;
  for (int e = threadIdx.x; e < kBuckets; e += kChunk) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) c += wave_cnt[w][e];
    row[e] = (uint16_t)c;
  }
}

// 2a. per bucket the exclusive prefix of its counts over the chunks (chunk order = row order) and its total: one
//     thread per bucket, 64-thread workgroups (a single workgroup walking the whole 6 MB table is bound by ONE CU's
//     ~10 bytes per cycle: 100 us)
__global__ __launch_bounds__(64) void order_scan_kernel(const uint16_t* __restrict__ chunk_count, int chunks,
                                                        uint32_t* __restrict__ chunk_prefix,
                                                        uint32_t* __restrict__ bucket_total) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  uint32_t run = 0;
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {  // eight loads in flight
    uint16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = chunk_count[(int64_t)(c + j) * kBuckets + b];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      chunk_prefix[(int64_t)(c + j) * kBuckets + b] = run;
      run += v[j];
    }
  }
  for (; c < chunks; ++c) {
    const uint32_t v = chunk_count[(int64_t)c * kBuckets + b];
    chunk_prefix[(int64_t)c * kBuckets + b] = run;
    run += v;
  }
  bucket_total[b] = run;
}

// 2b. the buckets' bases: exclusive scan of the 4096 totals, one workgroup, four buckets per thread
__global__ __launch_bounds__(1024) void order_base_kernel(const uint32_t* __restrict__ bucket_total,
                                                          uint32_t* __restrict__ bucket_base) {
  __shared__ uint32_t part[1024];
  static_assert(kBuckets == 4096, "four buckets per thread");
  const uint4 t = reinterpret_cast<const uint4*>(bucket_total)[threadIdx.x];
  part[threadIdx.x] = t.x + t.y + t.z + t.w;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan of the 1024 sums
    const uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const uint32_t base = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  reinterpret_cast<uint4*>(bucket_base)[threadIdx.x] = uint4{base, base + t.x, base + t.x + t.y, base + t.x + t.y + t.z};
}

// 3. sorted position = bucket base + rows of the bucket in earlier chunks + rank inside the chunk; inside full
//    blocks of kBlockRows the transposition, the tail as it is
//    With `coords` the row is gathered on the spot (what mri_gather_batch would do with the ordered indices: the
//    same coordinates and targets, one launch less on a side stream that has none to spare).
__global__ __launch_bounds__(kChunk) void order_place_kernel(const int64_t* __restrict__ idx_copy, int64_t n,
                                                             const uint16_t* __restrict__ bucket,
                                                             const uint16_t* __restrict__ rank,
                                                             const uint32_t* __restrict__ chunk_prefix,
                                                             const uint32_t* __restrict__ bucket_base,
                                                             int64_t* __restrict__ idx, int dim, OrderShape s,
                                                             const float* __restrict__ axes,
                                                             const float* __restrict__ volume,
                                                             float* __restrict__ coords, float* __restrict__ target) {
  const int64_t i = (int64_t)blockIdx.x * kChunk + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = bucket[i];
  const int64_t q = (int64_t)bucket_base[b] + chunk_prefix[(int64_t)blockIdx.x * kBuckets + b] + rank[i];
  const int64_t inner = q % kBlockRows, base = q - inner;
  constexpr int w = kBlockRows / 64;
  const int64_t pos = base + kBlockRows <= n ? base + (inner % w) * 64 + inner / w : q;
  const int64_t flat = idx_copy[i];
  idx[pos] = flat;
  if (coords) {
    if (target) target[pos] = volume[flat];
    int64_t rest = flat;
    for (int d = dim - 1; d >= 0; --d) {  // C order: last axis fastest (datamodules.py:162-163), as mri_gather_batch
      const int64_t at = rest % s.shape[d];
      rest /= s.shape[d];
      coords[pos * dim + d] = axes[s.axis_offset[d] + at];
    }
  }
}

struct OrderWs {
  int64_t* idx_copy;
  uint16_t *bucket, *rank, *chunk_count;
  uint32_t *chunk_prefix, *bucket_total, *bucket_base;
  int64_t total;
};

OrderWs carve_order(void* base, int64_t n) {
  OrderWs w{};
  const int64_t chunks = ceil_div(n, kChunk);
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  char* p = static_cast<char*>(base);
  int64_t off = 0;
  w.idx_copy = reinterpret_cast<int64_t*>(p + off), off += up(n * 8);
  w.bucket = reinterpret_cast<uint16_t*>(p + off), off += up(n * 2);
  w.rank = reinterpret_cast<uint16_t*>(p + off), off += up(n * 2);
  w.chunk_count = reinterpret_cast<uint16_t*>(p + off), off += up(chunks * kBuckets * 2);
  w.chunk_prefix = reinterpret_cast<uint32_t*>(p + off), off += up(chunks * kBuckets * 4);
  w.bucket_total = reinterpret_cast<uint32_t*>(p + off), off += up(kBuckets * 4);
  w.bucket_base = reinterpret_cast<uint32_t*>(p + off), off += up(kBuckets * 4);
  w.total = off;
  return w;
}

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int64_t mri_order_batch_workspace_bytes(int64_t n, int32_t dim) {
  if (n < 1 || n >= (1ll << 31) || dim < 1 || dim > MRI_MAX_DIM) return -1;
  return carve_order(nullptr, n).total;
}

namespace {
int order_impl(int64_t* idx, int64_t n, int32_t dim, const int64_t* shape, const float* axes, const int64_t* axis_offset,
               const float* volume, float* coords_out, float* target_out, void* workspace, int64_t workspace_bytes,
               void* stream) {
  MRI_REQUIRE(n >= 0 && n < (1ll << 31) && dim >= 1 && dim <= MRI_MAX_DIM, "bad n / dim");
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(idx && shape && workspace, "NULL pointer");
  MRI_REQUIRE(!coords_out || (axes && axis_offset), "coords_out needs the axis tables");
  MRI_REQUIRE(!target_out || (volume && coords_out), "target_out needs a volume (and coords_out)");
  MRI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  const OrderWs w = carve_order(workspace, n);
  MRI_REQUIRE(workspace_bytes >= w.total, "mri_order_batch needs a workspace of %lld bytes (mri_order_batch_workspace_bytes)",
              (long long)w.total);
  OrderShape s{};
  for (int d = 0; d < dim; ++d) {
    MRI_REQUIRE(shape[d] >= 1, "shape[%d] < 1", d);
    s.shape[d] = shape[d];
    s.axis_offset[d] = axis_offset ? axis_offset[d] : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)ceil_div(n, kChunk);
  int small = 1;  // 32-bit index arithmetic on the device where it is exact
  double voxels = 1.0;
  for (int d = 0; d < dim; ++d) {
    voxels *= (double)shape[d];
    if (shape[d] >= (1ll << 24)) small = 0;
  }
  if (voxels >= 4294967296.0) small = 0;
  hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)chunks), dim3(kChunk), 0, st, (const int64_t*)idx, n, (int)dim, s,
                     small, w.idx_copy, w.bucket, w.rank, w.chunk_count);
  hipLaunchKernelGGL(order_scan_kernel, dim3(kBuckets / 64), dim3(64), 0, st, (const uint16_t*)w.chunk_count, chunks,
                     w.chunk_prefix, w.bucket_total);
  hipLaunchKernelGGL(order_base_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t*)w.bucket_total, w.bucket_base);
  hipLaunchKernelGGL(order_place_kernel, dim3((unsigned)chunks), dim3(kChunk), 0, st, (const int64_t*)w.idx_copy, n,
                     (const uint16_t*)w.bucket, (const uint16_t*)w.rank, (const uint32_t*)w.chunk_prefix,
                     (const uint32_t*)w.bucket_base, idx, (int)dim, s, axes, volume, coords_out, target_out);
  return check_launch("mri_order_batch");
}
}  // namespace

extern "C" int mri_order_batch(int64_t* idx, int64_t n, int32_t dim, const int64_t* shape, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  return order_impl(idx, n, dim, shape, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int mri_order_gather_batch(int64_t* idx, int64_t n, int32_t dim, const int64_t* shape, const float* axes,
                                      const int64_t* axis_offset, const float* volume, float* coords_out,
                                      float* target_out, void* workspace, int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(coords_out != nullptr, "NULL coords_out");
  return order_impl(idx, n, dim, shape, axes, axis_offset, volume, coords_out, target_out, workspace, workspace_bytes,
                    stream);
}
