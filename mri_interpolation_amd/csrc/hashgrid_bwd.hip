// Hash-grid backward for gfx950: table-gradient scatter by BINNING + LDS fixed-point accumulation.
//
// Replaces aten::embedding_dense_backward x L plus the mul/sum backward (autograd of reference
// encoding.py:127-128; 73 % of the reference's CPU step).
//
// Why not atomics.  B * L * 2^D (33.5 M at BASELINE config 2/4) gradient rows go to hashed,
// i.e. random, slots.  Measured on MI355X:
//   * scattered global_atomic_add_f32: 20 G atomics/s chip-wide (they execute at the memory
//     side, 64 B per request)                       -> 3.6 ms per step;
//   * ds_add_f32 (LDS): 0.8 lane-atomics/ns/CU regardless of addresses; ds_add_u64: 13; ds_add_u32: 22
//     (tools/lds_atomic_bench.hip)                   -> float LDS atomics are no way out either;
//   * every workgroup scanning the whole batch for the corners of "its" table slice redoes the
//     hashing 64x                                    -> 1.3 ms.
// So the contributions are first ROUTED to the workgroup that owns their table slice, then
// summed there with 64-bit integer LDS atomics:
//   1. (in 4.)    per-level max|d_out|  -> per-level power-of-two scale 2^e (on device)
//   2. count      per 512-coordinate chunk: corners per bin (bin = level x slice of kAccWords/F
//                 slots); then per bin an exclusive scan over the chunks (2b)
//   3. prefix     exclusive scan of the bin totals -> bin offsets
//   4. scatter    recompute the corners, stage (slot, w*g[0..F)) records in LDS grouped by
//                 bin, copy them out in bin-contiguous runs (coalesced)
//   5. accumulate one workgroup per bin (big bins: per entry range) adds its records into LDS as
//                 fixed point (v * 2^e as int64, ds_add_u64), converts once and adds the slice
//                 to d_table with coalesced accesses
//   6. finalize   bins that were cut into several entry ranges meet in an int64 workspace.
// Coarse levels (few slices) skip 2-5: dense_level_kernel evaluates every corner per slice instead.
// Integer addition is associative: the table gradient is BITWISE REPRODUCIBLE (independent of
// scheduling and of the order the records landed in); the sums keep >= 40 fraction bits below
// max|g| (|sum| <= n * max|g| < 2^61 cannot overflow).
// Records, written once and read once: F = 2 (the grids of this project): ONE 8-byte word, 13-bit slot
// within the slice + two values of 3-bit class and 22-bit mantissa relative to the level's max|g|
// (pack_record below: each value rounded to 18-21 significant bits before the exact sum); other F:
// 2 + 4 F bytes (16-bit slot + F float values).
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "hashgrid_common.h"

namespace mri {
namespace {

// Per-workgroup time stamps of the scatter kernel for tools/bwd_segments.py (a tools-only build with
// -DMRI_BWD_PROFILE; the shipped library has none of this).
#ifdef MRI_BWD_PROFILE
__device__ long long* g_bwd_profile = nullptr;
#define BWDP(k)                                                                                   \
  if (g_bwd_profile && threadIdx.x == 0)                                                          \
    g_bwd_profile[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = (k) == 0 || (k) == 7 \
                                                                                 ? wall_clock64() \
                                                                                 : clock64();
// (accumulate workgroups: slots of their own behind the scatter's)
#define BWDA(k)                                                                                  \
  if (g_bwd_profile && threadIdx.x == 0)                                                         \
    g_bwd_profile[(65536 + (int64_t)blockIdx.x) * 8 + (k)] = (k) == 0 || (k) == 7 ? wall_clock64() : clock64();
// (dense workgroups: behind the accumulate ones)
#define BWDD(k)                                                                                  \
  if (g_bwd_profile && threadIdx.x == 0)                                                         \
    g_bwd_profile[(65536 + 4096 + (int64_t)b) * 8 + (k)] = (k) == 0 || (k) == 7 ? wall_clock64() : clock64();
#else
#define BWDP(k)
#define BWDA(k)
#define BWDD(k)
#endif

#ifndef MRI_ACC_WORDS  // A/B builds: accumulators (and threads) of an accumulate workgroup
#define MRI_ACC_WORDS 16384
#endif
#ifndef MRI_ACC_THREADS
#define MRI_ACC_THREADS 1024
#endif
constexpr int kAccWords = MRI_ACC_WORDS;  // 128 KiB of u64 accumulators per accumulate workgroup
static_assert(kAccWords <= 65536, "record slots are stored in 16 bits");
constexpr int kAccThreads = MRI_ACC_THREADS;
#ifndef MRI_BIN_THREADS  // A/B builds (tools/build_variant.py): threads = coordinates of a scatter workgroup
#define MRI_BIN_THREADS 512
#endif
constexpr int kBinThreads = MRI_BIN_THREADS;
constexpr int kStageWords = 24 * kBinThreads;  // LDS staging buffer of the scatter kernel: 48 KiB, 3 workgroups/CU
constexpr int kPackedStageWords = 18 * kBinThreads;  // ... with packed 8-byte records: 4608 of them
constexpr int kMaxParts = 256;        // slices per level handled by the binned path
constexpr int kHeaderWords = 64;      // per-level max|g| bits
constexpr int kMaxBins = MRI_MAX_LEVELS * kMaxParts;

struct BinPlan {
  int32_t n_entries;                 // levels handled by the binned path
  int32_t log2_slots;                // slice = 2^log2_slots table slots
  int32_t coords_per_block;          // of the count / scatter kernels
  int32_t total_bins;
  int32_t level_of[MRI_MAX_LEVELS];
  int32_t parts[MRI_MAX_LEVELS];     // slices (bins) of the level
  int32_t bin_start[MRI_MAX_LEVELS];  // first bin of the level
  int32_t splits[MRI_MAX_LEVELS];    // accumulate workgroups per bin
  int32_t acc_start[MRI_MAX_LEVELS + 1];  // first accumulate workgroup of the level
  int64_t ws_offset[MRI_MAX_LEVELS];      // int64 words; -1 = single owner, direct flush
};

struct Workspace {  // carved out of the caller's buffer
  uint32_t* max_bits;   // [kHeaderWords]       cleared at the start of every call
  uint32_t* cursor;     // [kMaxBins]           cleared at the start of every call
  uint32_t* offsets;    // [kMaxBins + 1]       bin starts, multiples of 4 records
  uint32_t* counts;     // [kMaxBins]           records per bin
  uint32_t* chunk_hist;  // [total_bins][chunks] corners of a chunk per bin
  uint32_t* chunk_base;  // [total_bins][chunks] where the chunk's run starts inside the bin
  unsigned long long* partial;  // [ws_words]   cleared at the start of every call
  int64_t partial_words;
  uint16_t* rec_slot;   // [records] slot within the bin's slice (< 2^14: kAccWords / F slots)
  float* rec_val;       // [F][records]
  int64_t records;
};

// (level_exponent: hashgrid_common.h)

__device__ __forceinline__ long long to_fixed(float v, float scale_hi) {
  // v * 2^e as int64 (e = log2(scale_hi) + 32): integer part of v * 2^(e-32) in the high
  // word, 31 more bits from the remainder.  Every step but the final truncation is exact.
  const float t = v * scale_hi;
  const int hi = (int)t;
  const float rem = t - (float)hi;
  const int lo = (int)(rem * 2147483648.0f);
  return ((long long)hi << 32) + ((long long)lo << 1);
}

// Packed records (F = 2): ONE 8-byte word = 13-bit slot | two values of 3 + 22 bits.  A value v = w g is
// stored as a 22-bit signed integer m and a 3-bit class c: v ~ m 2^-(es0 + 4 c), es0 = 16 - E for a level
// whose max |g| < 2^(E+1) (known BEFORE the scatter: level_absmax pre-pass), c = min((E + 4 - exponent(v)) / 4, 7).
// Class 0 is for |v| >= 2^(E+1): a coordinate outside the grid has interpolation weights up to 2 per axis
// (the reference extrapolates, encoding.py:113), so |w g| < 2^D max|g| <= 2^(E+5) for D <= 4.  Classes 1..7:
// a value keeps 18 to 21 significant bits whatever its size down to 2^-24 of the level's maximum (below
// that the resolution stays 2^(E-44); what is smaller than 2^(E-45) becomes 0), and values near the
// maximum are within 2^-22 of it of their f32 value.  The integer sum of the records is exact, so the
// gradient stays bitwise reproducible; per record it is rounded to those 18-21 bits (the tests' 1e-5 is
// on the level's largest gradient, where the rounding is 2.4e-7).  Plain 24-bit fixed point would lose
// every contribution below 2^-23 of the level's maximum, which Adam, scale-free per parameter, still sees.
// Record formats of the binned path.  Grids with two features per level choose at RUN time
// (mri_set_option("bwd_records", ...), read per call):
//   kRecSoA    (option 0, default, and every other F) 16-bit slot + F f32 values in F + 1 arrays: the
//              products are the f32 values the reference's autograd forms (encoding.py:127-128), and their
//              sum is exact -- at least the reference's precision;
//   kRecPacked (option 1, F = 2) the 8-byte word above: each product rounded to 18-21 significant bits
//              first (17x the median per-slot error of the f32 records, 31x the rms, measured against
//              float64: tests/test_gpu_round3.py), 13 us faster at BASELINE config 4.
// (Round 3 also built 12-byte records {slot, f32, f32} -- one staged store, one 12-byte global store and
// 16- or 12-byte loads per record, exact like kRecSoA: scatter 81 us against 76, accumulate 132 (151 with
// 12-byte loads) against 86 on one box: 96-bit LDS and global accesses are the slow path here.  Removed.)
enum RecFmt { kRecSoA = 0, kRecPacked = 1 };
template <int F>
inline int record_format() {
  return F == 2 && options().bwd_records == 1 ? kRecPacked : kRecSoA;
}
static_assert(kAccWords / 2 <= 8192, "packed records hold the slot within a slice in 13 bits");
__device__ __forceinline__ int level_E(uint32_t max_bits) { return (int)((max_bits >> 23) & 255u) - 127; }
__device__ __forceinline__ int rec_exponent(uint32_t max_bits) {  // es0
  return max(-126, min(16 - level_E(max_bits), 98));
}
__device__ __forceinline__ uint32_t pack_value(float v, int E, int es0) {  // -> c (3 bits) << 22 | m (22 bits)
  const int ev = (int)((__float_as_uint(v) >> 23) & 255u) - 127;
  const int c = min(max(E + 4 - ev, 0) >> 2, 7);
  const float scale = __uint_as_float((uint32_t)(es0 + 4 * c + 127) << 23);  // 2^(es0 + 4 c), <= 2^126
  const int m = max(-2097151, min(__float2int_rn(v * scale), 2097151));
  return ((uint32_t)c << 22) | ((uint32_t)m & 0x3fffffu);
}
__device__ __forceinline__ uint2 pack_record(uint32_t slot, float v0, float v1, int E, int es0) {
  const uint32_t a = pack_value(v0, E, es0), b = pack_value(v1, E, es0);  // 25 bits each
  // lo: slot [12:0] | c0 [15:13] | m0 low half [31:16];  hi: m0 high 6 bits [5:0] | c1 [8:6] | m1 [30:9]
  return make_uint2((slot & 0x1fffu) | ((a >> 22) << 13) | (a << 16),
                    ((a >> 16) & 0x3fu) | ((b >> 22) << 6) | ((b & 0x3fffffu) << 9));
}
// -> slot and the two values as integers in units of 2^-(es0 + 28) ... shifted by the caller: m and class
__device__ __forceinline__ void unpack_record(uint32_t lo, uint32_t hi, uint32_t& slot, int& m0, int& c0,
                                              int& m1, int& c1) {
  slot = lo & 0x1fffu;
  c0 = (lo >> 13) & 7, c1 = (hi >> 6) & 7;
  m0 = (int)(__builtin_amdgcn_alignbit(hi, lo, 16) << 10) >> 10;  // 22 bits from bit 16, sign-extended
  m1 = (int)(hi << 1) >> 10;                                       // bits 9..30
}

// Exclusive prefix sum of `count` <= 64 * kPerLane values in LDS by ONE wave (call from wave 0):
// out[i] = sum of in[0..i), out[count] = total.
template <int kPerLane>
__device__ __forceinline__ void wave_exclusive_scan(const uint32_t* in, uint32_t* out, int count) {
  const int lane = threadIdx.x & 63;
  uint32_t v[kPerLane], sum = 0;
#pragma unroll
  for (int j = 0; j < kPerLane; ++j) {
    const int idx = lane * kPerLane + j;
    v[j] = idx < count ? in[idx] : 0u;
    sum += v[j];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  uint32_t run = incl - sum;
#pragma unroll
  for (int j = 0; j < kPerLane; ++j) {
    const int idx = lane * kPerLane + j;
    if (idx < count) out[idx] = run;
    run += v[j];
  }
  if (lane == 63) out[count] = incl;
}

// coordinates of a count / scatter workgroup, and threads per coordinate: the staging buffer holds
// 2^D (1 + F) words per coordinate, so 4-D grids take 256 (128 with F = 4) coordinates per workgroup --
// their 16 corners are then split over 2 (4) threads instead of leaving half the workgroup idle
template <int D, int F>
struct BinGeometry {
  static constexpr int per_coord = (1 << D) * (1 + F);
  static constexpr int coords = kStageWords / per_coord / 64 * 64 < kBinThreads
                                    ? kStageWords / per_coord / 64 * 64 : kBinThreads;
  static constexpr int tpc = coords >= 64 ? kBinThreads / coords : 1;  // threads per coordinate
  static constexpr int corners = (1 << D) / tpc;                        // corners per thread
};

// Chunk of a count / scatter workgroup.  Workgroups go to the 8 XCDs round-robin by linear id, and
// chunk c's run inside a bin is followed by chunk c + 1's: with chunk = blockIdx.x the two halves of
// every shared 128-byte line would be written through two different L2s.  This map gives XCD k a
// contiguous range of chunks, in dispatch order (a bijection of [0, gridDim.x) for any grid width).
__device__ __forceinline__ int xcd_chunk() {
#if defined(MRI_BWD_PLAIN_CHUNKS)
  return blockIdx.x;
#else
  const int k = blockIdx.x & 7, j = blockIdx.x >> 3, q = gridDim.x >> 3, r = gridDim.x & 7;
  return (k < r ? k * (q + 1) : r * (q + 1) + (k - r) * q) + j;
#endif
}

// ------------------------------------------------------------------------ 2. count / 4. scatter
// One workgroup = (chunk of coords_per_block coordinates, level).  Both kernels walk the same
// corners in the same way; `SCATTER` selects what is done with them.
template <int D, int F, bool SCATTER, int R = kRecSoA>
__global__ __launch_bounds__(kBinThreads) void bin_kernel(
    const LevelTab tab, const BinPlan plan, const float* __restrict__ x,
    const float* __restrict__ d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
    uint32_t* __restrict__ chunk_hist, const uint32_t* __restrict__ chunk_base,
    const uint32_t* __restrict__ offsets, int chunks, uint16_t* __restrict__ rec_slot,
    float* __restrict__ rec_val, int64_t records, uint32_t* __restrict__ max_bits,
    uint32_t absmax_levels) {
  __shared__ uint32_t hist[kMaxParts];      // contributions of this workgroup per bin
  __shared__ uint32_t local_off[kMaxParts + 1];
  __shared__ uint32_t global_base[kMaxParts];
  // (packed records: 8 instead of 12 bytes per corner, room for an eighth more -> 36 KiB, four workgroups per CU)
  __shared__ __attribute__((aligned(16))) uint32_t stage[SCATTER ? (R == kRecPacked ? kPackedStageWords : kStageWords) : 1];
  __shared__ uint32_t wg_max;

  const int e = blockIdx.y;
  if (SCATTER && e >= plan.n_entries) {
    // Extra rows of the scatter grid: max |d_out| of the dense levels (`absmax_levels`, one row
    // per set bit), which the dense workgroups of the next launch need for their scale -- a
    // launch of its own cost 6 us.
    int level = -1;
    for (int l = 0, k = e - plan.n_entries; l < MRI_MAX_LEVELS; ++l)
      if ((absmax_levels >> l) & 1u) {
        if (k-- == 0) {
          level = l;
          break;
        }
      }
    if (level < 0) return;
    if (threadIdx.x == 0) wg_max = 0u;
    __syncthreads();
    const float* __restrict__ gl = d_out + (int64_t)level * sl;
    const int64_t i0 = (int64_t)blockIdx.x * plan.coords_per_block;
    const int64_t i1 = min(n, i0 + plan.coords_per_block);
    float gmax = 0.0f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kBinThreads)
#pragma unroll
      for (int f = 0; f < F; ++f) gmax = fmaxf(gmax, fabsf(gl[i * sr + f * sf]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_down(gmax, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&wg_max, __float_as_uint(gmax));
    __syncthreads();
    if (threadIdx.x == 0 && wg_max > __hip_atomic_load(max_bits + level, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(max_bits + level, wg_max);
    return;
  }
  const int level = plan.level_of[e];
  const int parts = plan.parts[e];
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  const uint32_t slot_mask = (1u << plan.log2_slots) - 1u;
  const float* __restrict__ res = tab.res[level];
  const int chunk = xcd_chunk();
  const int64_t i_begin = (int64_t)chunk * plan.coords_per_block;
  const int64_t i_end = min(n, i_begin + plan.coords_per_block);
  // per-(bin, chunk) tables, bin-major: entry (bin, chunk) at bin * chunks + chunk
  const uint64_t row0 = (uint64_t)plan.bin_start[e] * chunks + chunk;

  if (!SCATTER) {
    // count: how many corners of this chunk fall into each slice of the level
    for (int p = threadIdx.x; p < parts; p += kBinThreads) hist[p] = 0u;
    __syncthreads();
    using G = BinGeometry<D, F>;
    const int sub = threadIdx.x % G::tpc;  // this thread's share of the corners
    for (int64_t i = i_begin + threadIdx.x / G::tpc; i < i_end; i += kBinThreads / G::tpc) {
      uint32_t h0[D];
#pragma unroll
      for (int d = 0; d < D; ++d) h0[d] = cell_low32(x[i * D + d] * res[d]) * kPrimes[d];
      if constexpr (F == 2) {
        // PAIRS (see the scatter below): the two corners that differ on axis 0 are routed together, two
        // record positions per pair -- in ONE bin when both slots lie in the same slice (always, unless
        // the axis-0 cell index ends in log2_slots ones or the wrap of `% T` falls between them), else
        // two positions in EACH of the two bins (a record and a zero pad)
#pragma unroll
        for (int q = 0; q < G::corners; q += 2) {
          const int nb = sub * G::corners + q;
          uint32_t ha = h0[0];
#pragma unroll
          for (int d = 1; d < D; ++d) ha ^= ((nb >> d) & 1) ? h0[d] + kPrimes[d] : h0[d];
          const uint32_t hb = ha ^ h0[0] ^ (h0[0] + 1u);  // kPrimes[0] = 1
          const uint32_t pa = slot_of(ha, size, magic, pow2) >> plan.log2_slots;
          const uint32_t pb = slot_of(hb, size, magic, pow2) >> plan.log2_slots;
          atomicAdd(&hist[pa], 2u);
          if (pb != pa) atomicAdd(&hist[pb], 2u);
        }
      } else {
#pragma unroll
        for (int q = 0; q < G::corners; ++q) {
          const int nb = sub * G::corners + q;
          uint32_t h = 0;
#pragma unroll
          for (int d = 0; d < D; ++d) h ^= ((nb >> d) & 1) ? h0[d] + kPrimes[d] : h0[d];
          atomicAdd(&hist[slot_of(h, size, magic, pow2) >> plan.log2_slots], 1u);
        }
      }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < parts; p += kBinThreads)
      chunk_hist[row0 + (uint64_t)p * chunks] = hist[p];
    return;
  }

  // scatter: this chunk's run inside every bin was fixed by the count + scan stages (no
  // atomics, and the record order is the same on every run).
  // A workgroup lives for a chain of dependent global round trips (run tables -> scan -> coordinate
  // and gradient -> records -> stores), and only three fit a CU: the coordinate's loads are issued
  // first, so that they travel beside the run tables instead of behind them.
  BWDP(0) BWDP(1)
  using G = BinGeometry<D, F>;
  const int sub = threadIdx.x % G::tpc;  // this thread's share of the corners
  const int64_t i = i_begin + threadIdx.x / G::tpc;  // one coordinate per thread (group)
  const bool live = i < i_end;
  const float* __restrict__ gl = d_out + (int64_t)level * sl;
  float xi[D], g[F];
#pragma unroll
  for (int d = 0; d < D; ++d) xi[d] = live ? x[i * D + d] : 0.0f;
#pragma unroll
  for (int f = 0; f < F; ++f) g[f] = live ? gl[i * sr + f * sf] : 0.0f;
  for (int p = threadIdx.x; p < parts; p += kBinThreads) {
    hist[p] = chunk_hist[row0 + (uint64_t)p * chunks];
    global_base[p] = offsets[plan.bin_start[e] + p] + chunk_base[row0 + (uint64_t)p * chunks];
  }
  if (threadIdx.x == 0) wg_max = 0u;
  __syncthreads();
  BWDP(2)
  if (threadIdx.x < 64) wave_exclusive_scan<kMaxParts / 64>(hist, local_off, parts);
  __syncthreads();
  for (int p = threadIdx.x; p < parts; p += kBinThreads) hist[p] = 0u;  // fill counters
  __syncthreads();
  BWDP(3)

  // pass B: recompute the corners and stage (slot in slice, w * g[f]) grouped by bin
  const uint32_t total = local_off[parts];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (F == 2) {
    // Two features per level: the two corners that differ on axis 0 (PRIME_0 = 1: their slots differ in the
    // low bits only) are routed as a PAIR -- one bin lookup, one LDS counter atomic, one slot word and two
    // 8-byte value stores per pair instead of two atomics and six 4-byte stores, and half as many, twice
    // as wide stores in the copy-out.  A pair occupies two ADJACENT record positions, so the record
    // area holds exactly what it held before (slot16[2], val[f][2]) and the accumulate kernel reads it
    // unchanged; the sums are the same integers, the gradient bit-identical.  A pair whose slots lie in
    // two slices (axis-0 cell ending in log2_slots ones, cell -1, or the wrap of `% T` between them:
    // 0.1 % of the pairs of a non-power-of-two level, none of a power-of-two level of BASELINE's grids)
    // leaves a record and a zero pad in each of the two bins: the count stage reserved both.
    const uint32_t mb = R == kRecPacked ? max_bits[level] : 0u;  // (complete: level_absmax ran before this kernel)
    const int E = level_E(mb), es0 = rec_exponent(mb);
    // staging capacity in records: the usual 2^D per coordinate plus a sixth for pads; a workgroup beyond
    // it (a batch of coordinates outside the grid) writes its records straight to HBM
    constexpr uint32_t kCap = R == kRecPacked ? kPackedStageWords / 2 : kStageWords / 5 * 2 - 2;
    const bool staged = total <= kCap;
    // f32 records, staged per PAIR position pp = record position / 2: slot word | v[0] pair | v[1] pair
    const uint32_t pairs = total >> 1, pairs_even = (pairs + 1u) & ~1u;
    uint32_t* __restrict__ st_slot = stage;
    float2* __restrict__ st_v0 = reinterpret_cast<float2*>(stage + pairs_even);
    float2* __restrict__ st_v1 = reinterpret_cast<float2*>(stage + pairs_even + 2 * pairs);
    uint2* __restrict__ stage2 = reinterpret_cast<uint2*>(stage);  // packed records, per record position
    uint32_t* __restrict__ out_slot = reinterpret_cast<uint32_t*>(rec_slot);  // record area, per pair position
    float2* __restrict__ out_v0 = reinterpret_cast<float2*>(rec_val);
    float2* __restrict__ out_v1 = reinterpret_cast<float2*>(rec_val + records);  // (`records` is a multiple of 4)
    uint2* __restrict__ out_rec = reinterpret_cast<uint2*>(rec_val);
    // bin id in the spare bits of the staged slot word (two 16-bit halves holding log2_slots-bit slots): the
    // copy-out then walks the staging buffer linearly, every lane busy; a level with more slices than the
    // spare bits can name (64 for 13-bit slots) copies out bin by bin
    const int spare = 16 - plan.log2_slots;  // per half
    const bool flat = spare >= 1 && parts <= (1 << (2 * spare));
    const uint32_t half_mask = (1u << plan.log2_slots) - 1u, spare_mask = (1u << spare) - 1u;
    float gmax = 0.0f;
    if (live) gmax = fmaxf(fabsf(g[0]), fabsf(g[1]));
    if (R != kRecPacked && max_bits != nullptr) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_down(gmax, off, 64));
      // non-negative floats order like their bit patterns; one atomic per workgroup, and only if
      // it can still raise the level's maximum (thousands of workgroups share 16 addresses)
      if ((threadIdx.x & 63) == 0) atomicMax(&wg_max, __float_as_uint(gmax));
    }
    // (two instantiations of the loop, staged and straight to HBM: with one loop and a run-time choice of the
    // destination hipcc selects between an LDS and a global pointer through scratch and stores with flat_store)
    auto route = [&](auto staged_tag) {
      constexpr bool STAGED = decltype(staged_tag)::value;
      auto emit = [&](uint32_t p, uint32_t sa, uint32_t sb, float a0, float a1, float b0, float b1) {
        const uint32_t r = atomicAdd(&hist[p], 2u);  // (even: every add is 2)
        if constexpr (R == kRecPacked) {
          const uint2 ra = pack_record(sa, a0, a1, E, es0), rb = pack_record(sb, b0, b1, E, es0);
          if constexpr (STAGED) {
            stage2[local_off[p] + r] = ra, stage2[local_off[p] + r + 1] = rb;
          } else {
            out_rec[(uint64_t)global_base[p] + r] = ra, out_rec[(uint64_t)global_base[p] + r + 1] = rb;
          }
        } else {
          const uint32_t word = sa | (sb << 16);
          if constexpr (STAGED) {
            const uint32_t pp = (local_off[p] + r) >> 1;
            st_slot[pp] = flat ? word | ((p & spare_mask) << plan.log2_slots) | ((p >> spare) << (16 + plan.log2_slots))
                               : word;
            st_v0[pp] = make_float2(a0, b0);
            st_v1[pp] = make_float2(a1, b1);
          } else {
            const uint64_t pp = ((uint64_t)global_base[p] + r) >> 1;
            out_slot[pp] = word;
            out_v0[pp] = make_float2(a0, b0);
            out_v1[pp] = make_float2(a1, b1);
          }
        }
      };
      const Cell<D> c = locate<D>(xi, 0, res);
#pragma unroll
      for (int q = 0; q < G::corners; q += 2) {
        const int nb = sub * G::corners + q;
        uint32_t ha, hb;
        float wa, wb;
        corner<D>(c, nb, ha, wa);
        corner<D>(c, nb + 1, hb, wb);
        const uint32_t sa = slot_of(ha, size, magic, pow2), sb = slot_of(hb, size, magic, pow2);
        const uint32_t pa = sa >> plan.log2_slots, pb = sb >> plan.log2_slots;
        const float a0 = g[0] * wa, a1 = g[1] * wa, b0 = g[0] * wb, b1 = g[1] * wb;
        if (pa == pb) {
          emit(pa, sa & slot_mask, sb & slot_mask, a0, a1, b0, b1);
        } else {  // (rare, see above) a record and a zero pad on its own slot in each bin
          emit(pa, sa & slot_mask, sa & slot_mask, a0, a1, 0.0f, 0.0f);
          emit(pb, sb & slot_mask, sb & slot_mask, b0, b1, 0.0f, 0.0f);
        }
      }
    };
    if (live) {
      if (staged)
        route(std::true_type{});
      else
        route(std::false_type{});
    }
    __syncthreads();
    BWDP(4)
    if (R != kRecPacked && max_bits != nullptr && threadIdx.x == 0 &&
        wg_max > __hip_atomic_load(max_bits + level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(max_bits + level, wg_max);
    if (staged) {
      if constexpr (R == kRecPacked) {
        // copy out: each wave takes whole bins, lanes walk a bin's run -> contiguous 8-byte stores
        for (int p = wave; p < parts; p += kBinThreads / 64) {
          const uint32_t lo = local_off[p], cnt = local_off[p + 1] - lo;
          const uint64_t dst = (uint64_t)global_base[p];
          for (uint32_t k = lane; k < cnt; k += 64) out_rec[dst + k] = stage2[lo + k];
        }
      } else if (flat) {
        // linear over the staging buffer: lane k copies pair k, its bin is in the slot word
        for (uint32_t k = threadIdx.x; k < pairs; k += kBinThreads) {
          const uint32_t word = st_slot[k];
          const float2 v0 = st_v0[k], v1 = st_v1[k];
          const uint32_t p = ((word >> plan.log2_slots) & spare_mask) | ((word >> (16 + plan.log2_slots)) << spare);
          const uint64_t dst = ((uint64_t)global_base[p] >> 1) + (k - (local_off[p] >> 1));
          out_slot[dst] = word & (half_mask * 0x10001u);
          out_v0[dst] = v0;
          out_v1[dst] = v1;
        }
      } else {
        for (int p = wave; p < parts; p += kBinThreads / 64) {
          const uint32_t lo = local_off[p] >> 1, cnt = (local_off[p + 1] >> 1) - lo;
          const uint64_t dst = (uint64_t)global_base[p] >> 1;
          for (uint32_t k = lane; k < cnt; k += 64) {
            out_slot[dst + k] = st_slot[lo + k];
            out_v0[dst + k] = st_v0[lo + k];
            out_v1[dst + k] = st_v1[lo + k];
          }
        }
      }
    }
  } else {
    // other feature counts: one record per corner, 16-bit slot + F f32 values in F + 1 arrays.  max |g|
    // seen by this thread feeds the level's fixed-point scale, which only the accumulate launch needs
    float gmax = 0.0f;
    if (live) {
#pragma unroll
      for (int f = 0; f < F; ++f) gmax = fmaxf(gmax, fabsf(g[f]));
    }
    if (max_bits != nullptr) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_down(gmax, off, 64));
      if ((threadIdx.x & 63) == 0) atomicMax(&wg_max, __float_as_uint(gmax));
    }
    if (live) {
      const Cell<D> c = locate<D>(xi, 0, res);
#pragma unroll
      for (int q = 0; q < G::corners; ++q) {
        const int nb = sub * G::corners + q;
        uint32_t h;
        float w;
        corner<D>(c, nb, h, w);
        const uint32_t slot = slot_of(h, size, magic, pow2);
        const uint32_t p = slot >> plan.log2_slots;
        const uint32_t pos = local_off[p] + atomicAdd(&hist[p], 1u);
        stage[pos] = slot & slot_mask;
#pragma unroll
        for (int f = 0; f < F; ++f) stage[(1 + f) * total + pos] = __float_as_uint(g[f] * w);
      }
    }
    __syncthreads();
    BWDP(4)
    if (max_bits != nullptr && threadIdx.x == 0 &&
        wg_max > __hip_atomic_load(max_bits + level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(max_bits + level, wg_max);
    // copy out: each wave takes whole bins, lanes walk a bin's run -> contiguous global stores
    for (int p = wave; p < parts; p += kBinThreads / 64) {
      const uint32_t lo = local_off[p], cnt = local_off[p + 1] - lo;
      const uint64_t dst = (uint64_t)global_base[p];
      for (uint32_t k = lane; k < cnt; k += 64) {
        rec_slot[dst + k] = (uint16_t)stage[lo + k];
#pragma unroll
        for (int f = 0; f < F; ++f)
          rec_val[(uint64_t)f * records + dst + k] = __uint_as_float(stage[(1 + f) * total + lo + k]);
      }
    }
  }
#ifdef MRI_BWD_PROFILE
  BWDP(5)
  __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
  BWDP(6) BWDP(7)
#endif
}

// ----------------------------------------------------------------------------- 2b. chunk scan
// One wave per bin: exclusive scan of the bin's per-chunk counts (where each chunk's run starts
// inside the bin) and the bin total.
__global__ __launch_bounds__(64) void bin_chunk_scan_kernel(const uint32_t* __restrict__ chunk_hist,
                                                            uint32_t* __restrict__ chunk_base,
                                                            uint32_t* __restrict__ cursor,
                                                            int chunks) {
  const uint64_t row = (uint64_t)blockIdx.x * chunks;
  const int lane = threadIdx.x;
  uint32_t carry = 0;
  for (int c0 = 0; c0 < chunks; c0 += 64) {
    const int c = c0 + lane;
    const uint32_t v = c < chunks ? chunk_hist[row + c] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (c < chunks) chunk_base[row + c] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) cursor[blockIdx.x] = carry;
}

// ------------------------------------------------------------------------------ 3. prefix
// Bin counts (in `cursor`) -> exclusive offsets; `cursor` then holds each bin's write cursor.
// One 256-thread workgroup and 32 bytes of LDS: this kernel is queued on the side stream and must
// find room beside the decoder kernel, whose workgroups hold all but 2.5 KiB of every CU's LDS.
__global__ __launch_bounds__(256) void bin_prefix_kernel(uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ offsets,
                                                         uint32_t* __restrict__ counts,
                                                         int total_bins) {
  __shared__ uint32_t wave_total[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (total_bins + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(total_bins, lo + per);
  uint32_t s = 0;
  for (int b = lo; b < hi; ++b) s += (cursor[b] + 3u) & ~3u;  // bins start on 16-byte boundaries
  uint32_t incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  uint32_t run = incl - s;
  for (int w = 0; w < wave; ++w) run += wave_total[w];
  for (int b = lo; b < hi; ++b) {
    const uint32_t v = cursor[b];
    offsets[b] = run;
    counts[b] = v;
    run += (v + 3u) & ~3u;
  }
  if (threadIdx.x == 255) offsets[total_bins] = run;
}

// Adam fused into the last stage (one rank, no gradient accumulation): where a table gradient entry
// is complete -- a slice's sole accumulate workgroup, or the finalize pass -- the update of
// torch.optim.Adam (train_ops.hip: adam_kernel, the same operations in the same order) is applied to
// the parameter and its two moments right there; the gradient never goes to HBM (8 bytes per table
// parameter and step) and the optimiser launch shrinks to the decoder's parameters.
struct AdamFuse {
  float* p;  // the table itself (null: not fused, the gradient is written as before)
  float* m;
  float* v;
  float one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps, grad_scale;
};

__device__ __forceinline__ void adam_apply(const AdamFuse& a, uint64_t idx, float g) {
  const float gr = g * a.grad_scale;
  float m = a.m[idx], v = a.v[idx];
  m = m + (gr - m) * a.one_minus_b1;
  v = v * a.b2 + (a.one_minus_b2 * gr) * gr;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  a.m[idx] = m;
  a.v[idx] = v;
  a.p[idx] = a.p[idx] + (a.neg_step_size * m) / denom;
}

// ------------------------------------------------------------------------------ 5. accumulate
template <int F, int R>
__device__ __forceinline__ void bin_accumulate_body(
    unsigned long long* __restrict__ acc, int b, const LevelTab& tab, const BinPlan& plan,
    int64_t n, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
    const uint16_t* __restrict__ rec_slot, const float* __restrict__ rec_val, int64_t records,
    const uint32_t* __restrict__ max_bits, float* __restrict__ d_table,
    unsigned long long* __restrict__ partial, int overwrite, const AdamFuse& ad) {
  int e = 0;
  while (e + 1 < plan.n_entries && b >= plan.acc_start[e + 1]) ++e;
  const int level = plan.level_of[e];
  const int splits = plan.splits[e];
  const int part = (b - plan.acc_start[e]) / splits, split = (b - plan.acc_start[e]) % splits;
  const uint32_t slots = 1u << plan.log2_slots;
  const uint32_t base = (uint32_t)part * slots;
  const uint32_t count = min(slots, tab.size[level] - base);
  const int bin = plan.bin_start[e] + part;
  const uint32_t r_lo = offsets[bin], r_cnt = counts[bin];  // r_lo is a multiple of 4
  const uint32_t per = ((r_cnt + splits - 1) / splits + 3u) & ~3u;
  const uint32_t k_lo = min(r_cnt, (uint32_t)split * per), k_hi = min(r_cnt, k_lo + per);
  if (k_lo >= k_hi) {  // nothing routed here (uniform for the workgroup)
    if (ad.p && plan.ws_offset[e] < 0) {  // zero gradient: the moments still decay, the parameter still moves
      const uint64_t first = (tab.offset[level] + base) * F;
      for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads) adam_apply(ad, first + s, 0.0f);
    } else if (overwrite && plan.ws_offset[e] < 0) {
      float* __restrict__ dst = d_table + (tab.offset[level] + base) * F;
      for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads) dst[s] = 0.0f;
    }
    return;
  }

  BWDA(0) BWDA(1)
  // packed records: two sets of kPG x 4 records per lane, the next set's loads in flight while this
  // one's values go into LDS -- and the first set requested BEFORE the slice is zeroed: with one
  // workgroup per CU every latency a workgroup waits out is idle CU time (tools/bwd_segments.py)
  constexpr int kPG = 2;
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const uint2* __restrict__ rec = reinterpret_cast<const uint2*>(rec_val) + (uint64_t)r_lo;
  const uint32_t k_vec = k_lo + ((k_hi - k_lo) & ~3u), k_step = 4 * kAccThreads * kPG;
  u4v ra[2][kPG], rb[2][kPG];
  auto load_set = [&](int set, uint32_t k0) {
#pragma unroll
    for (int g = 0; g < kPG; ++g) {
      const uint32_t k = k0 + g * 4 * kAccThreads;
      if (k < k_vec) {  // read once: non-temporal loads leave the L2 to the gradient slices written below
        ra[set][g] = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(rec + k));
        rb[set][g] = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(rec + k + 2));
      }
    }
  };
  if constexpr (R == kRecPacked) load_set(0, k_lo + 4 * threadIdx.x);
  for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads) acc[s] = 0ull;
  // fused Adam: this slice's parameters and moments are fetched NOW, before the record phase (read at
  // the end they would add their full HBM latency to every workgroup: 0.21 -> 0.28 ms measured)
  constexpr int kPre = kAccWords / kAccThreads;
  const bool fuse_here = ad.p != nullptr && plan.ws_offset[e] < 0;
  const uint64_t first_word = (tab.offset[level] + base) * F;
  float pre_p[kPre], pre_m[kPre], pre_v[kPre];
  if (fuse_here) {
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const uint32_t s = threadIdx.x + j * kAccThreads;
      if (s < count * F) {
        pre_p[j] = ad.p[first_word + s];
        pre_m[j] = __builtin_nontemporal_load(ad.m + first_word + s);
        pre_v[j] = __builtin_nontemporal_load(ad.v + first_word + s);
      }
    }
  }
  __syncthreads();
  BWDA(2)
  const int ex = level_exponent(max_bits[level], n);
  const float scale_hi = __builtin_ldexpf(1.0f, ex - 32);
  if constexpr (R == kRecPacked) {
    // packed 8-byte records (pack_record): value = m 2^-(es0 + 4 c), accumulator unit 2^-ex: a shift by
    // ex - es0 - 4 c (44 - log2 n - 4 c)
    const int sh = ex - rec_exponent(max_bits[level]);  // for class 0; 4 less per class
    auto add = [&](uint32_t lo, uint32_t hi) {
      uint32_t slot;
      int m0, c0, m1, c1;
      unpack_record(lo, hi, slot, m0, c0, m1, c1);
      const int s0 = sh - 4 * c0, s1 = sh - 4 * c1;
      atomicAdd(&acc[slot * 2], (unsigned long long)(((long long)m0 << max(s0, 0)) >> max(-s0, 0)));
      atomicAdd(&acc[slot * 2 + 1], (unsigned long long)(((long long)m1 << max(s1, 0)) >> max(-s1, 0)));
    };
    auto add_set = [&](int set, uint32_t k0) {
#pragma unroll
      for (int g = 0; g < kPG; ++g) {
        const uint32_t k = k0 + g * 4 * kAccThreads;
        if (k < k_vec) {
          add(ra[set][g].x, ra[set][g].y), add(ra[set][g].z, ra[set][g].w);
          add(rb[set][g].x, rb[set][g].y), add(rb[set][g].z, rb[set][g].w);
        }
      }
    };
    for (uint32_t k0 = k_lo + 4 * threadIdx.x; k0 < k_vec; k0 += 2 * k_step) {  // (set 0 of the first round is loaded)
      load_set(1, k0 + k_step);
      add_set(0, k0);
      load_set(0, k0 + 2 * k_step);
      add_set(1, k0 + k_step);
    }
    for (uint32_t kt = k_vec + threadIdx.x; kt < k_hi; kt += kAccThreads) add(rec[kt].x, rec[kt].y);
  } else {
    // 4 records per lane and load (8- and 16-byte accesses: r_lo, k_lo and `records` are multiples of 4),
    // kGroups such loads per array in flight before the first LDS atomic: the kernel is latency
    // bound (78 % of wave cycles in s_waitcnt), not LDS bound
    constexpr int kGroups = 4;
    const uint16_t* __restrict__ slot_ptr = rec_slot + (uint64_t)r_lo;
    const float* __restrict__ val_ptr = rec_val + (uint64_t)r_lo;
    for (uint32_t k0 = k_lo + 4 * threadIdx.x; k0 < k_vec; k0 += 4 * kAccThreads * kGroups) {
      uint2 rel[kGroups];  // four 16-bit slots
      float4 val[kGroups][F];
  #pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        const uint32_t k = k0 + g * 4 * kAccThreads;
        if (k < k_vec) {
          // read once: non-temporal loads leave the L2 to the gradient slices written below (25 us)
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          typedef float f4v __attribute__((ext_vector_type(4)));
          const u2v r_ = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(slot_ptr + k));
          rel[g] = make_uint2(r_.x, r_.y);
  #pragma unroll
          for (int f = 0; f < F; ++f) {
            const f4v v_ = __builtin_nontemporal_load(
                reinterpret_cast<const f4v*>(val_ptr + (uint64_t)f * records + k));
            val[g][f] = make_float4(v_.x, v_.y, v_.z, v_.w);
          }
        }
      }
  #pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        const uint32_t k = k0 + g * 4 * kAccThreads;
        if (k < k_vec) {
          const uint32_t r4[4] = {rel[g].x & 0xffffu, rel[g].x >> 16, rel[g].y & 0xffffu,
                                  rel[g].y >> 16};
  #pragma unroll
          for (int f = 0; f < F; ++f) {
            const float v4[4] = {val[g][f].x, val[g][f].y, val[g][f].z, val[g][f].w};
  #pragma unroll
            for (int j = 0; j < 4; ++j)
              atomicAdd(&acc[r4[j] * F + f], (unsigned long long)to_fixed(v4[j], scale_hi));
          }
        }
      }
    }
    for (uint32_t kt = k_vec + threadIdx.x; kt < k_hi; kt += kAccThreads) {
      const uint32_t rel_t = slot_ptr[kt];
  #pragma unroll
      for (int f = 0; f < F; ++f)
        atomicAdd(&acc[rel_t * F + f],
                  (unsigned long long)to_fixed(val_ptr[(uint64_t)f * records + kt], scale_hi));
    }
  }
  BWDA(3)
  __syncthreads();
  BWDA(4)
  const int64_t ws_off = plan.ws_offset[e];
  if (ws_off < 0) {  // sole owner of the slice: convert once, add to the f32 gradient
    const double inv_scale = __builtin_ldexp(1.0, -ex);
    float* __restrict__ dst = d_table + (tab.offset[level] + base) * F;
    if (fuse_here) {
#pragma unroll
      for (int j = 0; j < kPre; ++j) {
        const uint32_t s = threadIdx.x + j * kAccThreads;
        if (s < count * F) {
          const float gr = (float)((double)(long long)acc[s] * inv_scale) * ad.grad_scale;
          const float m = pre_m[j] + (gr - pre_m[j]) * ad.one_minus_b1;
          const float v = pre_v[j] * ad.b2 + (ad.one_minus_b2 * gr) * gr;
          const float denom = sqrtf(v) / ad.bc2_sqrt + ad.eps;
          __builtin_nontemporal_store(m, ad.m + first_word + s);
          __builtin_nontemporal_store(v, ad.v + first_word + s);
          ad.p[first_word + s] = pre_p[j] + (ad.neg_step_size * m) / denom;
        }
      }
    } else if (overwrite) {
      for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads)
        dst[s] = (float)((double)(long long)acc[s] * inv_scale);
    } else {
      for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads) {
        const long long v = (long long)acc[s];
        if (v) dst[s] += (float)((double)v * inv_scale);
      }
    }
  } else {  // one entry range of a big bin: meet the other ranges in the integer workspace
    unsigned long long* __restrict__ dst = partial + ws_off + (uint64_t)base * F;
    for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads)
      if (acc[s]) atomicAdd(dst + s, acc[s]);
  }
#ifdef MRI_BWD_PROFILE
  BWDA(5)
  __builtin_amdgcn_s_waitcnt(0);
  BWDA(6) BWDA(7)
#endif
}

template <int F, int R>
__global__ __launch_bounds__(kAccThreads) void bin_accumulate_kernel(
    const LevelTab tab, const BinPlan plan, int64_t n, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ counts, const uint16_t* __restrict__ rec_slot,
    const float* __restrict__ rec_val, int64_t records,
    const uint32_t* __restrict__ max_bits, float* __restrict__ d_table,
    unsigned long long* __restrict__ partial, int overwrite, const AdamFuse ad) {
  __shared__ unsigned long long acc[kAccWords];
  bin_accumulate_body<F, R>(acc, blockIdx.x, tab, plan, n, offsets, counts, rec_slot, rec_val,
                            records, max_bits, d_table, partial, overwrite, ad);
}

// ------------------------------------------------------------------------- coarse levels
// A level whose table is cut into only a few slices does not need records at all: a workgroup
// (level, slice, coordinate range) can afford to evaluate the corners of every coordinate of its
// range and add the ones that fall into its slice straight into LDS -- the redundancy is the
// number of slices (<= bwd_dense_max_parts), against a record written and read per corner.
// Ranges meet in the int64 workspace like the entry ranges of the binned levels.
template <int F>
__global__ __launch_bounds__(256) void dense_absmax_kernel(const BinPlan plan,
                                                           const float* __restrict__ d_out,
                                                           int64_t n, int64_t sl, int64_t sr,
                                                           int64_t sf,
                                                           uint32_t* __restrict__ max_bits) {
  __shared__ uint32_t wg_max;
  const int level = plan.level_of[blockIdx.y];
  const float* __restrict__ gl = d_out + (int64_t)level * sl;
  if (threadIdx.x == 0) wg_max = 0u;
  __syncthreads();
  float m = 0.0f;
  if (sr == 1 && (sf & 3) == 0 && (reinterpret_cast<uintptr_t>(gl) & 15) == 0) {
    // feature-major block (the fused step's layout): rows of n contiguous values, 16 bytes per load
    const int64_t n4 = n >> 2;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const float4* __restrict__ row = reinterpret_cast<const float4*>(gl + f * sf);
      // four loads in flight per thread: one after the other the kernel is a chain of HBM latencies (15 us)
      const int64_t step = (int64_t)gridDim.x * 256;
      for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * step) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = i + j * step < n4 ? row[i + j * step] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
      }
      if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(gl[f * sf + 4 * n4 + threadIdx.x]));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
      for (int f = 0; f < F; ++f) m = fmaxf(m, fabsf(gl[i * sr + f * sf]));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(&wg_max, __float_as_uint(m));
  __syncthreads();
  if (threadIdx.x == 0 && wg_max > __hip_atomic_load(max_bits + level, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(max_bits + level, wg_max);
}

template <int D, int F>
__device__ __forceinline__ void dense_level_body(
    unsigned long long* __restrict__ acc, int b, const LevelTab& tab, const BinPlan& plan,
    const float* __restrict__ x, const float* __restrict__ d_out, int64_t n, int64_t sl,
    int64_t sr, int64_t sf, const uint32_t* __restrict__ max_bits,
    unsigned long long* __restrict__ partial) {
  int e = 0;
  while (e + 1 < plan.n_entries && b >= plan.acc_start[e + 1]) ++e;
  const int level = plan.level_of[e];
  const int splits = plan.splits[e];
  const int part = (b - plan.acc_start[e]) / splits, split = (b - plan.acc_start[e]) % splits;
  const uint32_t size = tab.size[level], magic = tab.magic[level];
  const bool pow2 = tab.pow2[level] != 0;
  const uint32_t slots = 1u << plan.log2_slots;
  const uint32_t base = (uint32_t)part * slots;
  const uint32_t count = min(slots, size - base);
  const int64_t per = (n + splits - 1) / splits;
  const int64_t i_begin = (int64_t)split * per, i_end = min(n, i_begin + per);

  BWDD(0) BWDD(1)
  for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads) acc[s] = 0ull;
  __syncthreads();
  BWDD(2)
  const int ex = level_exponent(max_bits[level], n);
  const float scale_hi = __builtin_ldexpf(1.0f, ex - 32);
  const float* __restrict__ res = tab.res[level];
  const float* __restrict__ gl = d_out + (int64_t)level * sl;
  // A thread walks ~20 coordinates; the next one's coordinate and gradient are requested before this one's
  // corners are hashed: loaded at the top of each iteration the walk was a chain of 20 global round trips
  // (tools/bwd_segments.py: 46 us per workgroup, vector ALU a good half busy)
  float xn[D], gn[F];
  int64_t i = i_begin + threadIdx.x;
  if (i < i_end) {
#pragma unroll
    for (int d = 0; d < D; ++d) xn[d] = x[i * D + d];
#pragma unroll
    for (int f = 0; f < F; ++f) gn[f] = gl[i * sr + f * sf];
  }
  for (; i < i_end; i += kAccThreads) {
    float xi[D], g[F];
#pragma unroll
    for (int d = 0; d < D; ++d) xi[d] = xn[d];
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = gn[f];
    const int64_t j = i + kAccThreads;
    if (j < i_end) {
#pragma unroll
      for (int d = 0; d < D; ++d) xn[d] = x[j * D + d];
#pragma unroll
      for (int f = 0; f < F; ++f) gn[f] = gl[j * sr + f * sf];
    }
    const Cell<D> c = locate<D>(xi, 0, res);
#pragma unroll
    for (int nb = 0; nb < (1 << D); ++nb) {
      uint32_t h;
      float w;
      corner<D>(c, nb, h, w);
      const uint32_t rel = slot_of(h, size, magic, pow2) - base;
      if (rel < count) {
#pragma unroll
        for (int f = 0; f < F; ++f)
          atomicAdd(&acc[rel * F + f], (unsigned long long)to_fixed(g[f] * w, scale_hi));
      }
    }
  }
  BWDD(3)
  __syncthreads();
  BWDD(4)
  unsigned long long* __restrict__ dst = partial + plan.ws_offset[e] + (uint64_t)base * F;
  for (uint32_t s = threadIdx.x; s < count * F; s += kAccThreads)
    if (acc[s]) atomicAdd(dst + s, acc[s]);
#ifdef MRI_BWD_PROFILE
  BWDD(5)
  __builtin_amdgcn_s_waitcnt(0);
  BWDD(6) BWDD(7)
#endif
}

template <int D, int F>
__global__ __launch_bounds__(kAccThreads) void dense_level_kernel(
    const LevelTab tab, const BinPlan plan, const float* __restrict__ x,
    const float* __restrict__ d_out, int64_t n, int64_t sl, int64_t sr, int64_t sf,
    const uint32_t* __restrict__ max_bits, unsigned long long* __restrict__ partial) {
  __shared__ unsigned long long acc[kAccWords];
  dense_level_body<D, F>(acc, blockIdx.x, tab, plan, x, d_out, n, sl, sr, sf, max_bits, partial);
}

// Dense levels and record accumulation in ONE launch: both bodies want a whole CU (1024 threads,
// 128 KiB of LDS).  The accumulate workgroups of BASELINE config 4 (820 on 256 CUs) leave their
// fourth round 80 % empty and the dense launch (192 workgroups) leaves a quarter of the CUs idle;
// together they fill four rounds, the (longer) dense workgroups first.
template <int D, int F, int R>
__global__ __launch_bounds__(kAccThreads) void dense_and_accumulate_kernel(
    const LevelTab tab, const BinPlan dense, int dense_blocks, const BinPlan plan,
    const float* __restrict__ x, const float* __restrict__ d_out, int64_t n, int64_t sl,
    int64_t sr, int64_t sf, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ counts, const uint16_t* __restrict__ rec_slot,
    const float* __restrict__ rec_val, int64_t records, const uint32_t* __restrict__ max_bits,
    float* __restrict__ d_table, unsigned long long* __restrict__ partial, int overwrite,
    const AdamFuse ad) {
  __shared__ unsigned long long acc[kAccWords];
  const int b = blockIdx.x;
#if defined(MRI_BWD_EXP) && (MRI_BWD_EXP & 4)  // timing experiment: the records are not read
  if (b >= dense_blocks) return;
#endif
#if defined(MRI_BWD_EXP) && (MRI_BWD_EXP & 8)  // timing experiment: the dense levels are skipped
  if (b < dense_blocks) return;
#endif
  if (b < dense_blocks)
    dense_level_body<D, F>(acc, b, tab, dense, x, d_out, n, sl, sr, sf, max_bits, partial);
  else
    bin_accumulate_body<F, R>(acc, b - dense_blocks, tab, plan, n, offsets, counts, rec_slot,
                              rec_val, records, max_bits, d_table, partial, overwrite, ad);
}

// ------------------------------------------------------------------------------ 6. finalize
__global__ __launch_bounds__(256) void bin_finalize_kernel(const LevelTab tab, const BinPlan plan,
                                                           int F, int64_t n,
                                                           float* __restrict__ d_table,
                                                           const uint32_t* __restrict__ max_bits,
                                                           const unsigned long long* __restrict__ partial,
                                                           int overwrite, const AdamFuse ad) {
  const int e = blockIdx.y;
  if (plan.ws_offset[e] < 0) return;
  const int level = plan.level_of[e];
  const double inv_scale = __builtin_ldexp(1.0, -level_exponent(max_bits[level], n));
  const uint64_t words = (uint64_t)tab.size[level] * F;
  const unsigned long long* __restrict__ src = partial + plan.ws_offset[e];
  float* __restrict__ dst = d_table + tab.offset[level] * F;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < words;
       s += (uint64_t)gridDim.x * 256) {
    const long long v = (long long)src[s];
    if (ad.p)
      adam_apply(ad, tab.offset[level] * F + s, (float)((double)v * inv_scale));
    else if (overwrite)
      dst[s] = (float)((double)v * inv_scale);
    else if (v)
      dst[s] += (float)((double)v * inv_scale);
  }
}

// ------------------------------------------------------------------------------ host side
// Which levels take the binned path and how the kernels are cut; returns false if none does.
bool make_plan(const mri_grid_desc* g, int64_t n, int method, BinPlan& plan, BinPlan& dense,
               int& dense_blocks, uint32_t& atomic_mask, int64_t& ws_words, int64_t& records,
               int& acc_blocks) {
  const int F = g->n_features, D = g->dim;
  plan = BinPlan{};
  dense = BinPlan{};
  dense_blocks = 0;
  atomic_mask = 0;
  ws_words = records = 0;
  acc_blocks = 0;
  const int slots = kAccWords / F;
  plan.log2_slots = 31 - __builtin_clz((unsigned)slots);
  const int per_coord = (1 + F) << D;  // staging words per coordinate
  plan.coords_per_block = std::min(kBinThreads, kStageWords / per_coord / 64 * 64);
  const bool supported = D <= 4 && F <= 4 && plan.coords_per_block >= 64;
  const int target = options().bwd_blocks_per_level;
  for (int l = 0; l < g->n_levels; ++l) {
    const int parts = (int)ceil_div(g->table_size[l], slots);
    const bool binned = supported && parts <= kMaxParts &&
                        (method == 2 || (method == 0 && parts <= options().bwd_lds_max_parts));
    if (!binned) {
      atomic_mask |= 1u << l;
      continue;
    }
    if (parts <= options().bwd_dense_max_parts) {  // coarse level: no records, see dense_level_kernel
      const int e = dense.n_entries++;
      dense.level_of[e] = l;
      dense.parts[e] = parts;
      dense.ws_offset[e] = ws_words;
      ws_words += (int64_t)g->table_size[l] * F;
      continue;
    }
    const int e = plan.n_entries++;
    // a level has n * 2^D records; aim at `target` accumulate workgroups per level
    int splits = std::max(1, target / parts);
    splits = (int)std::min<int64_t>(splits, std::max<int64_t>(1, (n << D) / parts / 4096));
    plan.level_of[e] = l;
    plan.parts[e] = parts;
    plan.bin_start[e] = plan.total_bins;
    plan.total_bins += parts;
    plan.splits[e] = splits;
    plan.acc_start[e] = acc_blocks;
    acc_blocks += parts * splits;
    plan.acc_start[e + 1] = acc_blocks;
    plan.ws_offset[e] = -1;
    if (splits > 1) {
      plan.ws_offset[e] = ws_words;
      ws_words += (int64_t)g->table_size[l] * F;
    }
    // two features per level: pairs whose slots lie in two slices leave a pad in each (bin_kernel); every
    // pair of a batch can be one (coordinates one cell outside the grid on axis 0), so room for twice the corners
    records += (n << D) * (F == 2 ? 2 : 1);
  }
  // Dense levels: every workgroup hashes ALL corners of its coordinate range and keeps those of
  // its slice, so a workgroup's time is set by the length of that range alone: one range length
  // for all dense levels (the same number of splits).  Their workgroups run in the launch of the
  // record accumulation (dense_and_accumulate_kernel), ahead of its workgroups; ~96 of them
  // measured best there (fewer: they become the long pole; more: more int64 merges).
  int dense_parts = 0;
  for (int e = 0; e < dense.n_entries; ++e) dense_parts += dense.parts[e];
  if (dense_parts > 0) {
    int splits = std::max(1, options().bwd_dense_blocks / dense_parts);
    splits = (int)std::min<int64_t>(splits, std::max<int64_t>(1, n / 2048));
    for (int e = 0; e < dense.n_entries; ++e) {
      dense.splits[e] = splits;
      dense.acc_start[e] = dense_blocks;
      dense_blocks += dense.parts[e] * splits;
      dense.acc_start[e + 1] = dense_blocks;
    }
  }
  records = (records + 4 * (int64_t)plan.total_bins + 3) / 4 * 4;  // bins are padded to 4 records
  dense.log2_slots = plan.log2_slots;
  return plan.n_entries > 0 || dense.n_entries > 0;
}

int64_t chunk_table_words(const BinPlan& plan, int64_t n) {
  const int64_t chunks = plan.coords_per_block > 0 ? ceil_div(n, plan.coords_per_block) : 0;
  return (int64_t)plan.total_bins * chunks + 4;
}

// bytes per record of the largest format the grid may choose at run time (16-bit slot + F f32 values)
inline int64_t record_bytes(int F) { return 2 + 4 * F; }

int64_t workspace_bytes(const BinPlan& plan, int64_t n, int64_t ws_words, int64_t records, int F) {
  return (int64_t)kHeaderWords * 4 + 2 * (int64_t)kMaxBins * 4 + (int64_t)(kMaxBins + 1) * 4 + 12 +
         2 * chunk_table_words(plan, n) * 4 + ws_words * 8 + records * record_bytes(F) + 80;
}

// Fixed-position regions first, the record area next, the int64 area at the END of the buffer.
Workspace carve(void* base, int64_t total_bytes, const BinPlan& plan, int64_t n, int64_t ws_words,
                int64_t records, int F) {
  Workspace w{};
  char* p = static_cast<char*>(base);
  w.max_bits = reinterpret_cast<uint32_t*>(p);
  p += kHeaderWords * 4;
  w.cursor = reinterpret_cast<uint32_t*>(p);
  p += (int64_t)kMaxBins * 4;
  w.offsets = reinterpret_cast<uint32_t*>(p);
  p += (int64_t)(kMaxBins + 1) * 4 + 12;  // keeps the following fields 16-byte aligned
  w.counts = reinterpret_cast<uint32_t*>(p);
  p += (int64_t)kMaxBins * 4;
  const int64_t table = chunk_table_words(plan, n) / 4 * 4;  // keeps 16-byte alignment
  w.chunk_hist = reinterpret_cast<uint32_t*>(p);
  p += table * 4;
  w.chunk_base = reinterpret_cast<uint32_t*>(p);
  p += table * 4;
  w.rec_slot = reinterpret_cast<uint16_t*>(p);
  p += (records * 2 + 15) / 16 * 16;
  w.rec_val = reinterpret_cast<float*>(p);
  w.records = records;
  const int64_t tail = (total_bytes - ws_words * 8) & ~int64_t(15);
  w.partial = reinterpret_cast<unsigned long long*>(static_cast<char*>(base) + tail);
  w.partial_words = ws_words;
  return w;
}

// The entries of `plan` whose level is in `mask`, for the launches of one level group: positions
// in the record / bin / int64 areas stay those of the full plan (so one prepare call serves every
// group), only the workgroup -> entry map (acc_start) is renumbered.
BinPlan select_levels(const BinPlan& plan, uint32_t mask, int& blocks) {
  BinPlan sel = plan;
  sel.n_entries = 0;
  blocks = 0;
  for (int e = 0; e < plan.n_entries; ++e) {
    if (!((mask >> plan.level_of[e]) & 1u)) continue;
    const int k = sel.n_entries++;
    sel.level_of[k] = plan.level_of[e];
    sel.parts[k] = plan.parts[e];
    sel.bin_start[k] = plan.bin_start[e];
    sel.splits[k] = plan.splits[e];
    sel.ws_offset[k] = plan.ws_offset[e];
    sel.acc_start[k] = blocks;
    blocks += plan.parts[e] * plan.splits[e];
    sel.acc_start[k + 1] = blocks;
  }
  return sel;
}

thread_local FinTab* tl_fin_out = nullptr;  // set_finalize_export(): taken by the next finalize that would launch

template <int D, int F>
struct BinnedLaunch {
  static int run(const LevelTab& tab, const BinPlan& plan, const BinPlan& dense_all,
                 const Workspace& w, int n_levels, uint32_t level_mask, int phase, int overwrite,
                 const float* x, const float* d_out, int64_t n, int64_t sl, int64_t sr,
                 int64_t sf, float* d_table, const AdamFuse& ad, const uint32_t* ext_max, hipStream_t st) {
    // phase 0: everything; 1: count + prefix only (needs x alone, so it can run beside the
    // forward pass); 2: the rest, after a phase-1 call on the same workspace
    if constexpr (D <= 4 && F <= 4) {
      // The zero-on-entry regions are cleared here, per call: a buffer shared by calls with
      // different level sets or batch sizes then needs no invariant across calls.
      // (the int64 area is cleared with the header, i.e. by the prepare call when there is one:
      // on the side stream it costs nothing, on the main stream 6 us)
      if (phase != 2) {
        (void)hipMemsetAsync(w.max_bits, 0, kHeaderWords * 4, st);
        if (w.partial_words > 0)
          (void)hipMemsetAsync(w.partial, 0, (size_t)w.partial_words * 8, st);
      }
      // phase 1 counts every level; the gradient launches below cover the levels of the mask
      int dense_blocks = 0, acc_blocks = 0;
      const BinPlan dense = select_levels(dense_all, level_mask, dense_blocks);
      const BinPlan sel = select_levels(plan, level_mask, acc_blocks);
      // dense levels: their launch is merged with the record accumulation when the call has both
      const bool fuse_dense = dense.n_entries > 0 && sel.n_entries > 0 && options().bwd_fuse_dense;
      // max |d_out| per level (float bit patterns): the workspace header, filled by a pass over d_out -- or,
      // two features per level (feature pair = level), the caller's array (the decoder kernel that produced
      // d_out knows it already)
      const int fmt = record_format<F>();
      const bool given = ext_max != nullptr && F == 2;
      uint32_t* const mx = given ? const_cast<uint32_t*>(ext_max) : w.max_bits;
      if (dense.n_entries > 0 && phase != 1 && !fuse_dense) {
        if (!given)
          hipLaunchKernelGGL((dense_absmax_kernel<F>), dim3(128, dense.n_entries), dim3(256), 0, st,
                             dense, d_out, n, sl, sr, sf, mx);
        if (!fuse_dense)
          hipLaunchKernelGGL((dense_level_kernel<D, F>), dim3((unsigned)dense_blocks),
                             dim3(kAccThreads), 0, st, tab, dense, x, d_out, n, sl, sr, sf,
                             mx, w.partial);
      }
      // ONE finalize launch for everything that met in the int64 area: the dense levels and the
      // binned levels whose bins were cut over entry ranges
      BinPlan fin{};
      for (int e = 0; e < dense.n_entries && phase != 1; ++e) {
        fin.level_of[fin.n_entries] = dense.level_of[e];
        fin.ws_offset[fin.n_entries++] = dense.ws_offset[e];
      }
      for (int e = 0; e < sel.n_entries && phase != 1; ++e)
        if (sel.ws_offset[e] >= 0) {
          fin.level_of[fin.n_entries] = sel.level_of[e];
          fin.ws_offset[fin.n_entries++] = sel.ws_offset[e];
        }
      auto finalize = [&]() {
        if (fin.n_entries > 0 && tl_fin_out && overwrite && !ad.p) {
          // the caller converts (set_finalize_export: the Adam kernel of mri_fused_step, as it fetches the gradient)
          FinTab& t = *tl_fin_out;
          tl_fin_out = nullptr;
          t.n = fin.n_entries, t.max_bits = mx, t.n_coords = n;
          for (int e = 0; e < fin.n_entries; ++e) {
            const int level = fin.level_of[e];
            t.seg[e].begin = (int64_t)tab.offset[level] * F, t.seg[e].words = (int64_t)tab.size[level] * F;
            t.seg[e].src = w.partial + fin.ws_offset[e], t.seg[e].level = level;
          }
          return;
        }
        if (fin.n_entries > 0)
          hipLaunchKernelGGL(bin_finalize_kernel, dim3(256, fin.n_entries), dim3(256), 0, st, tab,
                             fin, F, n, d_table, mx, w.partial, overwrite, ad);
      };
      if (plan.n_entries == 0) {
        finalize();
        return check_launch("hashgrid backward (dense levels)");
      }
      const int chunks = (int)ceil_div(n, plan.coords_per_block);
      const dim3 bin_grid((unsigned)chunks, plan.n_entries);
      if (phase != 2) {
        hipLaunchKernelGGL((bin_kernel<D, F, false>), bin_grid, dim3(kBinThreads), 0, st, tab, plan,
                           x, d_out, n, sl, sr, sf, w.chunk_hist, w.chunk_base, w.offsets, chunks,
                           w.rec_slot, w.rec_val, w.records, mx, 0u);
        hipLaunchKernelGGL(bin_chunk_scan_kernel, dim3((unsigned)plan.total_bins), dim3(64), 0, st,
                           w.chunk_hist, w.chunk_base, w.cursor, chunks);
        hipLaunchKernelGGL(bin_prefix_kernel, dim3(1), dim3(256), 0, st, w.cursor, w.offsets,
                           w.counts, plan.total_bins);
      }
      if (phase == 1) return check_launch("hashgrid backward (count)");
      if (sel.n_entries == 0) {
        finalize();
        return check_launch("hashgrid backward (dense levels)");
      }
      uint32_t absmax_levels = 0;  // with the fused launch: dense absmax rides on the scatter grid
      if (fuse_dense)
        for (int e = 0; e < dense.n_entries; ++e) absmax_levels |= 1u << dense.level_of[e];
      if (fmt == kRecPacked || given) {
        // packed records are scaled by their level's max |g|, which must therefore be complete before the
        // scatter stages its first record: one pass over d_out for the binned (and fused dense) levels
        BinPlan both{};
        for (int e = 0; e < sel.n_entries; ++e) both.level_of[both.n_entries++] = sel.level_of[e];
        for (int e = 0; fuse_dense && e < dense.n_entries; ++e) both.level_of[both.n_entries++] = dense.level_of[e];
        if (!given)
          hipLaunchKernelGGL((dense_absmax_kernel<F>), dim3(64, both.n_entries), dim3(256), 0, st, both, d_out,
                             n, sl, sr, sf, mx);
        absmax_levels = 0;  // (f32 records with the maxima given: the scatter has none to find)
      }
      auto launch = [&](auto fmt_tag) {
        constexpr int R = decltype(fmt_tag)::value;
        // f32 records: the scatter finds the maxima on the way (extra rows for the fused dense levels),
        // unless they are given; packed records read them
        uint32_t* const scatter_max = (R != kRecPacked && given) ? nullptr : mx;
        const int extra_rows = (fuse_dense && R != kRecPacked && !given) ? dense.n_entries : 0;
        hipLaunchKernelGGL((bin_kernel<D, F, true, R>), dim3((unsigned)chunks, sel.n_entries + extra_rows),
                           dim3(kBinThreads), 0, st, tab, sel, x, d_out, n, sl, sr, sf,
                           w.chunk_hist, w.chunk_base, w.offsets, chunks, w.rec_slot, w.rec_val,
                           w.records, scatter_max, absmax_levels);
        if (fuse_dense)
          hipLaunchKernelGGL((dense_and_accumulate_kernel<D, F, R>),
                             dim3((unsigned)(dense_blocks + acc_blocks)), dim3(kAccThreads), 0, st,
                             tab, dense, dense_blocks, sel, x, d_out, n, sl, sr, sf, w.offsets,
                             w.counts, w.rec_slot, w.rec_val, w.records, mx, d_table,
                             w.partial, overwrite, ad);
        else
          hipLaunchKernelGGL((bin_accumulate_kernel<F, R>), dim3((unsigned)acc_blocks),
                             dim3(kAccThreads), 0, st, tab, sel, n, w.offsets, w.counts,
                             w.rec_slot, w.rec_val, w.records, mx, d_table, w.partial,
                             overwrite, ad);
      };
      if constexpr (F == 2) {
        if (fmt == kRecPacked)
          launch(std::integral_constant<int, kRecPacked>{});
        else
          launch(std::integral_constant<int, kRecSoA>{});
      } else {
        launch(std::integral_constant<int, kRecSoA>{});
      }
      finalize();
      return check_launch("hashgrid backward (binned)");
    } else {
      return fail(MRI_ERR_UNSUPPORTED, "binned backward supports dim <= 4, n_features <= 4");
    }
  }
};

}  // namespace
void set_finalize_export(FinTab* out) { tl_fin_out = out; }
}  // namespace mri

using namespace mri;

extern "C" int64_t mri_hashgrid_backward_workspace_bytes(const mri_grid_desc* grid, int64_t n) {
  if (validate(grid)) return -1;
  BinPlan plan, dense;
  uint32_t mask;
  int64_t words, records;
  int acc_blocks, dense_blocks;
  make_plan(grid, std::max<int64_t>(n, 1), 2, plan, dense, dense_blocks, mask, words, records,
            acc_blocks);
  return workspace_bytes(plan, std::max<int64_t>(n, 1), words, records, grid->n_features);
}

namespace {
int backward_impl(const mri_grid_desc* grid, const float* x, const float* d_out, int64_t n,
                  int64_t sl, int64_t sr, int64_t sf, float* d_table, int32_t method, int phase,
                  int overwrite, uint32_t level_mask, void* workspace,
                  int64_t workspace_bytes_given, void* stream, const AdamFuse* fuse = nullptr,
                  const float* level_absmax = nullptr) {
  if (int rc = validate(grid)) return rc;
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  MRI_REQUIRE(method >= 0 && method <= 2, "method %d not in 0..2", method);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && (phase == 1 || (d_out && (d_table || fuse))), "NULL device pointer");
  const int F = grid->n_features;
  const AdamFuse ad = fuse ? *fuse : AdamFuse{};
  BinPlan plan, dense;
  uint32_t atomic_mask;
  int64_t ws_words, records;
  int acc_blocks, dense_blocks;
  const bool planned = make_plan(grid, n, method, plan, dense, dense_blocks, atomic_mask, ws_words,
                                 records, acc_blocks);
  if (fuse && (!planned || atomic_mask != 0))
    return fail(MRI_ERR_UNSUPPORTED, "fused Adam needs every level on the binned path "
                                     "(a level of this grid takes the atomic kernel)");
  if (planned) {
    MRI_REQUIRE(records < (1ll << 32), "too many gradient records (%lld)", (long long)records);
    const int64_t need = workspace_bytes(plan, n, ws_words, records, F);
    MRI_REQUIRE(workspace != nullptr && workspace_bytes_given >= need,
                "hashgrid backward needs a workspace of %lld bytes "
                "(mri_hashgrid_backward_workspace_bytes), got %lld",
                (long long)need, (long long)workspace_bytes_given);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                "workspace must be 16-byte aligned");
    const Workspace w = carve(workspace, workspace_bytes_given, plan, n, ws_words, records, F);
    const LevelTab tab = make_tab(grid);
    int rc = dispatch<BinnedLaunch>(grid->dim, F, tab, plan, dense, w, grid->n_levels,
                                    level_mask, phase, overwrite, x, d_out, n, sl, sr, sf,
                                    d_table, ad, reinterpret_cast<const uint32_t*>(level_absmax),
                                    (hipStream_t)stream);
    if (rc) return rc;
  }
  atomic_mask &= level_mask;
  if (atomic_mask && phase != 1) {
    if (overwrite)  // the atomic kernel can only add: clear its levels first
      for (int l = 0; l < grid->n_levels; ++l)
        if ((atomic_mask >> l) & 1u)
          (void)hipMemsetAsync(d_table + grid->table_offset[l] * F, 0,
                               (size_t)grid->table_size[l] * F * 4, (hipStream_t)stream);
    return launch_backward_atomic(grid, atomic_mask, x, d_out, n, sl, sr, sf, d_table,
                                  (hipStream_t)stream);
  }
  return MRI_OK;
}
}  // namespace

extern "C" int mri_hashgrid_backward_prepare(const mri_grid_desc* grid, const float* x, int64_t n,
                                             int32_t method, void* workspace,
                                             int64_t workspace_bytes, void* stream) {
  return backward_impl(grid, x, nullptr, n, 0, 0, 0, nullptr, method, 1, 0, 0xffffffffu, workspace,
                       workspace_bytes, stream);
}

extern "C" int mri_hashgrid_backward(const mri_grid_desc* grid, const float* x,
                                     const float* d_out, int64_t n, int64_t dout_level_stride,
                                     int64_t dout_row_stride, int64_t dout_feat_stride,
                                     float* d_table, int32_t method, void* workspace,
                                     int64_t workspace_bytes_given, void* stream) {
  return mri_hashgrid_backward_levels(grid, x, d_out, n, dout_level_stride, dout_row_stride,
                                      dout_feat_stride, d_table, method, 0xffffffffu, workspace,
                                      workspace_bytes_given, stream);
}

extern "C" int mri_hashgrid_backward_levels(const mri_grid_desc* grid, const float* x,
                                            const float* d_out, int64_t n,
                                            int64_t dout_level_stride, int64_t dout_row_stride,
                                            int64_t dout_feat_stride, float* d_table,
                                            int32_t method, uint32_t level_mask, void* workspace,
                                            int64_t workspace_bytes_given, void* stream) {
  const int phase = (method & MRI_BWD_PREPARED) ? 2 : 0;
  const int overwrite = (method & MRI_BWD_OVERWRITE) ? 1 : 0;
  return backward_impl(grid, x, d_out, n, dout_level_stride, dout_row_stride, dout_feat_stride,
                       d_table, method & ~(MRI_BWD_PREPARED | MRI_BWD_OVERWRITE), phase, overwrite,
                       level_mask, workspace, workspace_bytes_given, stream);
}

extern "C" int mri_hashgrid_backward_scaled(const mri_grid_desc* grid, const float* x,
                                            const float* d_out, int64_t n,
                                            int64_t dout_level_stride, int64_t dout_row_stride,
                                            int64_t dout_feat_stride, float* d_table,
                                            int32_t method, uint32_t level_mask,
                                            const float* level_absmax, void* workspace,
                                            int64_t workspace_bytes_given, void* stream) {
  const int phase = (method & MRI_BWD_PREPARED) ? 2 : 0;
  const int overwrite = (method & MRI_BWD_OVERWRITE) ? 1 : 0;
  return backward_impl(grid, x, d_out, n, dout_level_stride, dout_row_stride, dout_feat_stride,
                       d_table, method & ~(MRI_BWD_PREPARED | MRI_BWD_OVERWRITE), phase, overwrite,
                       level_mask, workspace, workspace_bytes_given, stream, nullptr, level_absmax);
}

extern "C" int mri_hashgrid_backward_adam(const mri_grid_desc* grid, const float* x,
                                          const float* d_out, int64_t n,
                                          int64_t dout_level_stride, int64_t dout_row_stride,
                                          int64_t dout_feat_stride, float* table, float* exp_avg,
                                          float* exp_avg_sq, double lr, double beta1, double beta2,
                                          double eps, int32_t step, float grad_scale, int32_t method,
                                          void* workspace, int64_t workspace_bytes_given,
                                          void* stream) {
  MRI_REQUIRE(table && exp_avg && exp_avg_sq && step >= 1, "NULL parameter / moment pointer or step < 1");
  AdamFuse ad{};
  ad.p = table, ad.m = exp_avg, ad.v = exp_avg_sq;
  // scalar prefactors in double, as torch computes them on the host (mri_adam_step)
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  ad.one_minus_b1 = (float)(1.0 - beta1), ad.b2 = (float)beta2, ad.one_minus_b2 = (float)(1.0 - beta2);
  ad.neg_step_size = (float)(-(lr / bc1)), ad.bc2_sqrt = (float)sqrt(bc2), ad.eps = (float)eps;
  ad.grad_scale = grad_scale;
  const int phase = (method & MRI_BWD_PREPARED) ? 2 : 0;
  return backward_impl(grid, x, d_out, n, dout_level_stride, dout_row_stride, dout_feat_stride,
                       nullptr, method & ~(MRI_BWD_PREPARED | MRI_BWD_OVERWRITE), phase, 1, 0xffffffffu,
                       workspace, workspace_bytes_given, stream, &ad);
}

#ifdef MRI_BWD_PROFILE
extern "C" int mri_debug_set_bwd_profile(long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mri::g_bwd_profile), &device_buffer, sizeof(device_buffer)) ==
                 hipSuccess
             ? 0
             : -1;
}
#endif
