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
//     lookup 0.099 -> 0.090 ms, table gradient 0.222 -> 0.214 ms (its scatter writes longer runs).
//
// The sort is rocPRIM's radix_sort_pairs (stable, deterministic: the same batch gives the same order every run) on a
// 16-bit prefix of the Morton key; keys and the permutation live in the caller's workspace.
//
// State at the end of round 3: an OPTION (BatchPipeline(order="morton"), bench.py --batch-order morton), not the
// default.  With ordered batches every kernel of BASELINE config 4's step is faster under rocprofv3 (lookup 105.1 ->
// 96.6 us, scatter 75.6 -> 74.5, dense + accumulate 86.6 -> 82.7, count 60.2 -> 50.7, gather 20.1 -> 10.7) and the
// step is slower, 0.536 against 0.513 ms: for 2^18 pairs rocPRIM takes its merge-sort path -- one block sort and
// eight merge passes, 17 launches, ~120 us on the side stream -- which does not fit beside the lookup (indices are
// therefore produced two batches ahead, mri_fused_step_args::next2_idx) and then runs beside the table gradient,
// costing it 17 us.  What it needs is a two-pass counting sort of its own (~6 launches).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace mri {
namespace {

constexpr int kBlockRows = 16384;  // rows of a transposition block (a multiple of 64)

struct OrderShape {
  int64_t shape[MRI_MAX_DIM];
};

// Up to 8 bits per axis (position / extent, so that axes of different lengths weigh the same), interleaved with
// the LAST axis in the lowest bit; up to 4 axes (more: the first four decide) in at most 28 bits -- 7 bits per axis
// for 4 axes: rocPRIM 3.x loses values when the sorted bit window ends at bit 32 of a 32-bit key
// (tools/probes/rocprim_sort_probe.hip: windows [0, 16), [8, 24) fine, [16, 32): 99,998 of 100,000 values lost).
__host__ __device__ inline int axes_used(int dim) { return dim < 4 ? dim : 4; }
__host__ __device__ inline int axis_bits(int dim) { return axes_used(dim) < 4 ? 8 : 7; }

__global__ __launch_bounds__(256) void order_key_kernel(const int64_t* __restrict__ idx, int64_t n, int dim,
                                                        OrderShape s, uint32_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t rest = idx[i];
  uint32_t q[MRI_MAX_DIM];
  const int bits = axis_bits(dim);
  for (int d = dim - 1; d >= 0; --d) {  // C order: last axis fastest (as mri_gather_batch)
    const int64_t pos = rest % s.shape[d];
    rest /= s.shape[d];
    q[d] = (uint32_t)((pos << bits) / s.shape[d]);
  }
  const int used = axes_used(dim);
  uint32_t k = 0;
  for (int b = bits - 1; b >= 0; --b)
    for (int d = 0; d < used; ++d) k = (k << 1) | ((q[d] >> b) & 1u);
  key[i] = k;
}

// sorted row q -> position: inside full blocks of kBlockRows the transposition, the tail as it is
__global__ __launch_bounds__(256) void order_place_kernel(const int64_t* __restrict__ sorted, int64_t n,
                                                          int64_t* __restrict__ idx) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const int64_t inner = q % kBlockRows, base = q - inner;
  constexpr int w = kBlockRows / 64;
  const int64_t pos = base + kBlockRows <= n ? base + (inner % w) * 64 + inner / w : q;
  idx[pos] = sorted[q];
}

struct OrderWs {
  uint32_t *key_in, *key_out;
  int64_t* val_out;
  void* temp;
  size_t temp_bytes;
  int64_t total;
};

int key_bits(int dim) { return axis_bits(dim) * axes_used(dim); }

OrderWs carve_order(void* base, int64_t n, int dim) {
  OrderWs w{};
  size_t temp = 0;
  const int end = key_bits(dim), begin = end > 16 ? end - 16 : 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (const int64_t*)nullptr, (int64_t*)nullptr, (size_t)n, begin, end, (hipStream_t)0);
  const int64_t keys = (n * 4 + 255) / 256 * 256;
  char* p = static_cast<char*>(base);
  w.key_in = reinterpret_cast<uint32_t*>(p);
  w.key_out = reinterpret_cast<uint32_t*>(p + keys);
  w.val_out = reinterpret_cast<int64_t*>(p + 2 * keys);
  w.temp = p + 2 * keys + (n * 8 + 255) / 256 * 256;
  w.temp_bytes = temp;
  w.total = 2 * keys + (n * 8 + 255) / 256 * 256 + (int64_t)temp + 256;
  return w;
}

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int64_t mri_order_batch_workspace_bytes(int64_t n, int32_t dim) {
  if (n < 1 || dim < 1 || dim > MRI_MAX_DIM) return -1;
  return carve_order(nullptr, n, dim).total;
}

extern "C" int mri_order_batch(int64_t* idx, int64_t n, int32_t dim, const int64_t* shape, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(n >= 0 && n < (1ll << 31) && dim >= 1 && dim <= MRI_MAX_DIM, "bad n / dim");
  if (n < 2) return MRI_OK;
  MRI_REQUIRE(idx && shape && workspace, "NULL pointer");
  MRI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  const OrderWs w = carve_order(workspace, n, dim);
  MRI_REQUIRE(workspace_bytes >= w.total, "mri_order_batch needs a workspace of %lld bytes (mri_order_batch_workspace_bytes)",
              (long long)w.total);
  OrderShape s{};
  for (int d = 0; d < dim; ++d) {
    MRI_REQUIRE(shape[d] >= 1, "shape[%d] < 1", d);
    s.shape[d] = shape[d];
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL(order_key_kernel, dim3(blocks), dim3(256), 0, st, idx, n, (int)dim, s, w.key_in);
  const int end = key_bits(dim), begin = end > 16 ? end - 16 : 0;
  size_t temp = w.temp_bytes;
  if (rocprim::radix_sort_pairs(w.temp, temp, w.key_in, w.key_out, (const int64_t*)idx, w.val_out, (size_t)n, begin, end,
                                st) != hipSuccess)
    return fail(MRI_ERR_LAUNCH, "mri_order_batch: radix sort");
  hipLaunchKernelGGL(order_place_kernel, dim3(blocks), dim3(256), 0, st, w.val_out, n, idx);
  return check_launch("order_place_kernel");
}
