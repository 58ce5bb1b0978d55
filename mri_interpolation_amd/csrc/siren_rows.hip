// Fused SIREN chain for H = 256 with the activations of a wave's rows IN REGISTERS across all layers (round 4).
//
// Same contract as siren_chain.hip (reference models.py:153-156 SirenLayer.forward, :230-233 SirenNet.forward:
// `F.linear` + `sin(w0 .)` per layer, then the linear head), same arguments, same split-weight workspace.  What
// changes is who owns what.  siren_chain.hip keeps a 64-row tile's activations as an f32 image in LDS: eight waves
// share it, every wave re-reads and re-splits the image fragments the other column blocks also read, every
// 16-deep weight chunk is fenced by a workgroup barrier (64 per tile), the sincos epilogue of a layer runs with the
// matrix pipe idle, and the 66 KB image leaves room for two 24 KB weight chunks only -- the whole H x H matrix is
// streamed from L2 once per 64 rows.  Here:
//
//   * the product is computed TRANSPOSED, Z^T = W A^T: the weights are the MFMA's A operand (32 output features
//     per tile), a wave's 32 batch rows are the N dimension.  The 32x32 accumulator then holds, per lane, one batch
//     row (lane & 31) and the features (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile -- which is exactly the
//     B-operand layout of the NEXT layer's products over the contraction order the split weights already use
//     (siren_chain.h: slot q of a row holds k = 16 kc + 4 q + e and 16 kc + 8 + 4 q + e).  A layer's output becomes
//     the next layer's operand with no exchange at all: every activation is split into its three bf16 terms ONCE,
//     by the lane that computed it, and is never written to LDS;
//   * one workgroup = 4 waves, one per SIMD, 128 rows; a wave owns its 32 rows through the whole chain (up to 512
//     registers per lane: 192 for the layer's split input, 128 for its f32 output, 32 for two accumulator tiles);
//   * LDS holds a ring of weight chunks (4 x 24 KB; a chunk = the 32 features of one output tile x 128 contraction
//     indices x three terms) filled by LDS-DMA three chunks ahead, the small parameters and a 9 KB staging image
//     per wave; the matrix is streamed once per 128 rows (half the L2 traffic per row), one workgroup barrier per
//     chunk = per 48 MFMAs of a wave, placed two k-steps before the chunk's end so that the next chunk's first
//     fragments are requested early;
//   * the sincos epilogue of output tile i runs beside the MFMAs of tile i + 1 IN THE SAME WAVE: the order (one
//     MFMA, five or six vector instructions, ...) is written out gap by gap between sched_barrier(0) fences (a
//     sched_group_barrier pipeline is dropped by the scheduler when it judges the interleaved order's register
//     pressure too high).  The branch of sincos_fast2 (|u| > 8192 -> library sincosf) would cut those regions:
//     the interleaved code is branch-free, the range is checked once per tile and out-of-range values are
//     repaired on a cold path;
//   * a lane owns a ROW, 1 KB apart in HBM from its neighbours': activations leave (and the backward kernel's
//     derivative tiles arrive) through per-wave LDS images, transposed to 4 lanes x 16 bytes per half row;
//   * the backward chain (siren_backward_rows_kernel, below) has the same ownership from the head down; the
//     loss-mode forward leaves it w0 cos of the last sine layer and dLoss/dy per row.
// DESIGN.md 4.5 has the cycle budget; EXPERIMENTS.md (Part I) what the compiler needed to get there.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "bf16x3.h"
#include "common.h"
#include "device_math.h"
#include "siren_chain.h"

namespace mri {
namespace {
namespace rr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kH = 256;
constexpr int kThreadsR = 256;                   // 4 waves, one per SIMD
constexpr int kRows = 128;                       // rows of a group: 32 per wave
constexpr int kTiles = kH / 32;                  // output tiles of 32 features
constexpr int kSteps = kH / kKc;                 // 16-deep k-steps per layer
constexpr int kChunkSteps = 8;                   // k-steps per weight chunk: a chunk = (tile, half of K)
constexpr int kChunksPerLayer = kTiles * 2;
constexpr int kPiece = 32 * 32;                  // bytes of one term plane of one k-step of one tile
constexpr int kChunk = kChunkSteps * 3 * kPiece; // 24,576
constexpr int kRing = 4;
constexpr int kStgLd = 36;                        // floats per staged row: 144 bytes, conflict-free 16-byte writes
constexpr int kPlane = kH * 32;                  // one term plane of one k-step of the whole matrix (siren_chain.h)
constexpr int kStepBytes = 3 * kPlane;           // one k-step of the whole matrix: 24,576

// The small arrays first: their addresses then fit the 16-bit offset field of the DS instructions (beyond 64 KB hipcc
// keeps one address register per access, hoists hundreds of them out of the loops and spills them).
struct Smem {
  float bias[kMaxSine][kH] __attribute__((aligned(16)));
  float w_first[kH][kMaxIn];
  float w_last[kH];
  // Per wave: a tile's a and w0 cos on their way to HBM, [array][row][32 features + pad].  A lane owns a ROW (1 KB apart
  // in HBM): stored from registers, every 16-byte piece of a store instruction lands in another row (~180 cycles per
  // instruction measured).  Through this image the pieces go out as the rows lie: 4 lanes per 64-byte half row.  The cold
  // path (arguments beyond the fast sincos' range) borrows the same bytes as [value][lane].
  float stg[4][2][32 * kStgLd];
  // Per wave: a group's coordinates (one row of 64 lanes per input) and, by group parity, its targets: requested by
  // LDS-DMA a whole group ahead.  (Loaded into registers at the group's start they sit behind a wait that hipcc makes
  // vmcnt(0) -- an LDS-DMA is always in flight -- i.e. behind the previous group's last stores.)
  float xin[4][kMaxIn + 2][64];
  char ring[kRing][kChunk] __attribute__((aligned(16)));
};

__device__ __forceinline__ f32x16 mfma(const x3::u32x4& a, const x3::u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3::bf16x8, a),
                                                 __builtin_bit_cast(x3::bf16x8, b), c, 0, 0, 0);
}
// (a k-step = six of these, weight term x activation term: h l, l h, m m, h m, m h, h h -- the six products of
// siren_chain.hip's mma6_32, smallest first; written out in the tile loops)

// Queue this wave's share (6 of the 24 one-KiB pieces) of chunk (tile, half) of one split matrix: piece pc =
// k-step pc / 3 of the half, term plane pc % 3; lane = 16-byte slot, fetched with the half bit XORed by bit 3 of
// the row (the involution the fragment reads undo, as siren_chain.hip's issue_chunk).
__device__ __forceinline__ void issue(const char* __restrict__ mat, int tile, int half, char* dst, int wave, int lane) {
  const char* src = mat + (int64_t)(kChunkSteps * half) * kStepBytes + tile * kPiece + 16 * (lane ^ ((lane >> 4) & 1));
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pc = wave + 4 * j, ksl = pc / 3, p = pc - 3 * ksl;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ksl * kStepBytes + p * kPlane),
                                     (__attribute__((address_space(3))) void*)(dst + pc * kPiece), 16, 0, 0);
  }
}

__device__ __forceinline__ x3::Frag read_wfrag(const char* __restrict__ p) {
  x3::Frag f;
  f.h = *reinterpret_cast<const x3::u32x4*>(p);
  f.m = *reinterpret_cast<const x3::u32x4*>(p + kPiece);
  f.l = *reinterpret_cast<const x3::u32x4*>(p + 2 * kPiece);
  return f;
}

// Loops whose index must be a compile-time constant (register arrays, scheduling fences): `#pragma unroll` is a
// request the optimizer may decline (it did, and indexed the register file dynamically).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Sum each of 16 registers over the 16 lanes that share (lane >> 4): a halving butterfly -- per step a lane keeps one
// register of a pair and sends the other to its partner; lane (lane & 15) = r ends up with the sum of register r.  The
// exchanges are DPP modifiers (lane ^ 1, ^ 2: quad permutes; ^ 4 = half-row mirror then quad reverse; ^ 8 = row mirror then
// half-row mirror): no LDS round trip -- with __shfl_xor (ds_bpermute, every result behind an lgkmcnt(0)) one call took
// 1.4 k cycles, and a group's tail makes 8 to 64 of them.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float reduce16(const float (&v)[16], int lane) {
  const bool u1 = (lane & 1) != 0, u2 = (lane & 2) != 0, u4 = (lane & 4) != 0, u8 = (lane & 8) != 0;
  // (per level: every select first, then the exchanges -- a DPP operand written by the instruction in front of it costs
  // two wait states)
  float k8[8], s8[8], a8[8], k4[4], s4[4], a4[4], k2[2], s2[2], a2[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) k8[j] = u1 ? v[2 * j + 1] : v[2 * j], s8[j] = u1 ? v[2 * j] : v[2 * j + 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) a8[j] = k8[j] + dpp<0xB1>(s8[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) k4[j] = u2 ? a8[2 * j + 1] : a8[2 * j], s4[j] = u2 ? a8[2 * j] : a8[2 * j + 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = k4[j] + dpp<0x4E>(s4[j]);
#pragma unroll
  for (int j = 0; j < 2; ++j) k2[j] = u4 ? a4[2 * j + 1] : a4[2 * j], s2[j] = dpp<0x141>(u4 ? a4[2 * j] : a4[2 * j + 1]);
#pragma unroll
  for (int j = 0; j < 2; ++j) a2[j] = k2[j] + dpp<0x1B>(s2[j]);
  const float k1 = u8 ? a2[1] : a2[0], s1 = dpp<0x140>(u8 ? a2[0] : a2[1]);
  return k1 + dpp<0x141>(s1);
}
// ... and over bit 4 of the lane for a pair of tiles: lanes with the bit clear end up with the even tile's sum over the
// wave's 32 rows, the others with the odd tile's (v_permlane16_swap exchanges the odd rows of one register with the even
// rows of the other)
__device__ __forceinline__ float pair32(float even, float odd) {
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(even), __float_as_uint(odd), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}

// Cold path: sin / cos of 32 arguments per lane staged in LDS, by the library routine.
__device__ __attribute__((noinline)) void repair_values(float* fix, int count) {
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int r = 0; r < count; ++r) {
    float s, c;
    sincosf(fix[r * 64 + lane], &s, &c);
    fix[r * 64 + lane] = s;
    fix[(16 + r) * 64 + lane] = c;
  }
}

// Phase timing for tools/rows_phases.py (a tools-only build with -DSIREN_PROFILE; the shipped library compiles these
// to nothing): shader-clock cycles per phase, per wave.
#ifdef SIREN_PROFILE
__device__ long long* g_rows_profile = nullptr;
__device__ long long* g_rows_profile_bwd = nullptr;
#define RP_BEGIN long long rp_t = clock64(); long long rp_acc[8] = {};
#define RP_MARK(i) { const long long rp_n = clock64(); rp_acc[i] += rp_n - rp_t; rp_t = rp_n; }
#define RP_END_TO(buf)                                                       \
  if (buf && (threadIdx.x & 63) == 0) {                                        \
    long long* dst = buf + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; \
    for (int q = 0; q < 8; ++q) dst[q] = rp_acc[q];                            \
  }
#define RP_END RP_END_TO(g_rows_profile)
#else
#define RP_BEGIN
#define RP_MARK(i)
#define RP_END
#define RP_END_TO(buf)
#endif

// MODE 0: inference; 1: training (a and w0 cos of every layer leave for HBM); 2: training with the loss (siren_chain.hip's
// MODE 2: MSE against `target` and the head's backward in the group's tail; the last sine layer's a never leaves, its
// w0 cos waits in the dz_last buffer until the row's dLoss/dy is known and is replaced by dz there)
template <int MODE>
__global__ __launch_bounds__(kThreadsR) void siren_forward_rows_kernel(const ChainArgs a) {
  __shared__ Smem sm;
  constexpr bool STORE = MODE >= 1, LOSS = MODE == 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int n_mm = a.n_sine - 1;  // H x H layers (>= 1)

  for (int l = 0; l < a.n_sine; ++l)
    for (int e = tid; e < kH; e += kThreadsR) sm.bias[l][e] = a.b[l][e];
  for (int e = tid; e < kH; e += kThreadsR) sm.w_last[e] = a.w[a.n_sine][e];
  for (int e = tid; e < kH * kMaxIn; e += kThreadsR) {
    const int f = e / kMaxIn, d = e % kMaxIn;
    sm.w_first[f][d] = d < a.dim_in ? a.w[0][f * a.dim_in + d] : 0.f;
  }
  const float b_last = a.b[a.n_sine][0];
  const int64_t groups = (a.n + kRows - 1) / kRows;
  // this lane's slot inside a piece, per ring slot: four opaque bases, every fragment read a 16-bit offset from one
  // (opaque OFFSETS, not pointers: a laundered pointer loses its LDS address space and the reads become flat loads)
  int wslot[kRing];
#pragma unroll
  for (int q = 0; q < kRing; ++q) {
    wslot[q] = q * kChunk + 32 * l31 + 16 * (lh ^ ((l31 >> 3) & 1));
    asm volatile("" : "+v"(wslot[q]));
  }
  const char* const ring0 = &sm.ring[0][0];
  const int feat0 = 4 * lh;  // feature of accumulator register r of tile i: 32 i + (r & 3) + 8 (r >> 2) + feat0

  // a group's coordinates (and targets, slot kMaxIn + parity) on their way into xin: 4 bytes per lane and input
  auto request = [&](int64_t gg, int parity) {
    const int64_t r = std::min<int64_t>(gg * kRows + 32 * wave + l31, a.n - 1);
    for (int d = 0; d < a.dim_in; ++d)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.x + r * a.dim_in + d),
                                       (__attribute__((address_space(3))) void*)&sm.xin[wave][d][0], 4, 0, 0);
    if constexpr (LOSS)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.target + r),
                                       (__attribute__((address_space(3))) void*)&sm.xin[wave][kMaxIn + parity][0], 4, 0, 0);
  };
  // the chunk stream: chunks 0, 1, 2 of the first group
  if ((int64_t)blockIdx.x < groups) {
    request(blockIdx.x, 0);
    issue(a.wsplit, 0, 0, sm.ring[0], wave, lane);
    issue(a.wsplit, 0, 1, sm.ring[1], wave, lane);
    issue(a.wsplit, 1, 0, sm.ring[2], wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // parameters and chunk 0 .. 2 are in LDS

  float g_wh[4] = {0.f, 0.f, 0.f, 0.f}, g_bhead = 0.f, g_loss = 0.f;  // loss mode: running sums of this lane
  // (parked in accumulation registers between the groups' tails: the tile loop has no vector register to spare)
  auto park = [&]() {
    if constexpr (LOSS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(g_wh[j]));
      asm volatile("" : "+a"(g_bhead), "+a"(g_loss));
    }
  };
  park();
  RP_BEGIN
  int gpar = 0;  // parity of this workgroup's group count: which target slot is this group's
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x, gpar ^= 1) {
    // rows beyond n repeat row n - 1: the same values to the same addresses, no exec-masked store (a divergent branch
    // would cut the scheduling regions below)
    const int64_t row = std::min<int64_t>(g * kRows + 32 * wave + l31, a.n - 1);
    float out[kTiles][16];  // the layer's output, accumulator layout: the next layer's operand before its split
    // staging (STORE): this lane writes quad q of its row at wr + 8 q, reads piece (lane & 3) of rows (lane >> 2) + 16 j
    float* const stg_w = &sm.stg[wave][0][0];
    const int wr = l31 * kStgLd + feat0, rd = (lane >> 2) * kStgLd + 4 * (lane & 3);
    // element offsets of those two rows' pieces in an (n, H) array, from the group's first row (32 bits per lane; the
    // group's base is wave-uniform: two 64-bit offsets per lane were the registers the loss mode did not have)
    const int64_t gbase = g * kRows * kH;
    int goff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      goff[j] = (int)std::min<int64_t>(32 * wave + (lane >> 2) + 16 * j, a.n - 1 - g * kRows) * kH + 4 * (lane & 3);
    auto stage = [&](int arr, int q, const f32x4& v) { *reinterpret_cast<f32x4*>(stg_w + arr * 32 * kStgLd + wr + 8 * q) = v; };
    auto unstage = [&](int arr, int h, f32x4 (&v)[2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j) v[j] = *reinterpret_cast<const f32x4*>(stg_w + arr * 32 * kStgLd + rd + 16 * h + 16 * j * kStgLd);
    };
    auto put = [&](float* base, int tile, int h, const f32x4 (&v)[2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(base + goff[j] + 32 * tile + 16 * h) = v[j];
    };

    // ---- first layer on the VALU (K = dim_in) -----------------------------------------------------------
    {
      // this group's coordinates were requested a group ago (the first group's in the prologue): more than 63 requests
      // lie behind them in this wave's queue
      asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      float xv[kMaxIn];
#pragma unroll
      for (int d = 0; d < kMaxIn; ++d) xv[d] = d < a.dim_in ? sm.xin[wave][d][lane] : 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read: the next group's may land
      if (g + gridDim.x < groups) request(g + gridDim.x, gpar ^ 1);
      const bool wide = a.dim_in > 4;
#pragma unroll
      for (int i = 0; i < kTiles; ++i) {
        float u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(&sm.w_first[32 * i + (r & 3) + 8 * (r >> 2) + feat0][0]);
          float z = 0.f;
          z += xv[0] * wa.x, z += xv[1] * wa.y, z += xv[2] * wa.z, z += xv[3] * wa.w;
          u[r] = z;
        }
        if (wide) {  // dim_in > 4 (wave-uniform): the sums go on in the same order
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const f32x4 wb = *reinterpret_cast<const f32x4*>(&sm.w_first[32 * i + (r & 3) + 8 * (r >> 2) + feat0][4]);
            float z = u[r];
            z += xv[4] * wb.x, z += xv[5] * wb.y, z += xv[6] * wb.z, z += xv[7] * wb.w;
            u[r] = z;
          }
        }
        float amax = 0.f, c[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          u[r] = a.w0_first * (u[r] + sm.bias[0][32 * i + (r & 3) + 8 * (r >> 2) + feat0]);
          amax = fmaxf(amax, fabsf(u[r]));
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) sincos_fast2_core(u[r], u[r + 1], &out[i][r], &c[r], &out[i][r + 1], &c[r + 1]);
        if (__builtin_amdgcn_ballot_w64(!(amax <= 8192.0f)) != 0) {  // wave-uniform, cold
          float* fix = &sm.stg[wave][0][0];
#pragma unroll
          for (int r = 0; r < 16; ++r) fix[r * 64 + lane] = u[r];
          repair_values(fix, 16);
#pragma unroll
          for (int r = 0; r < 16; ++r) out[i][r] = fix[r * 64 + lane], c[r] = fix[(16 + r) * 64 + lane];
        }
        if (STORE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            stage(0, q, f32x4{out[i][4 * q], out[i][4 * q + 1], out[i][4 * q + 2], out[i][4 * q + 3]});
            stage(1, q, f32x4{a.w0_first * c[4 * q], a.w0_first * c[4 * q + 1], a.w0_first * c[4 * q + 2], a.w0_first * c[4 * q + 3]});
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 va[2], vd[2];
            unstage(0, h, va), unstage(1, h, vd);
            put(a.act[0] + gbase, i, h, va), put(a.deriv[0] + gbase, i, h, vd);
          }
        }
      }
    }

    RP_MARK(0)  // first layer
    // ---- H x H layers ------------------------------------------------------------------------------------
    for (int l = 1; l <= n_mm; ++l) {
      const char* mat = a.wsplit + (int64_t)(l - 1) * split_matrix_bytes(kH);
      const char* mat_next = l < n_mm ? mat + split_matrix_bytes(kH) : a.wsplit;
      // The last sine layer of the loss mode: a stays on chip, w0 cos waits in dz_last.  Its a is stored all the same -- into
      // dz_last, a slice ahead of the w0 cos that overwrites it (the same lane, the same address, in order): a wave-uniform
      // branch around those stores would cut the regions below into blocks, and the compiler then sinks the range check's
      // running maximum behind them, keeping (spilling) every slice's arguments until the tile's end.  (Sending them to a
      // 2 KB dummy instead saved 0.03 ms and cost the two registers this kernel form does not have.)
      const bool keep_a = LOSS && l == n_mm;
      float* gd = STORE ? (keep_a ? a.dz_last : a.deriv[l]) + gbase : nullptr;  // (the group's first row)
      float* ga = STORE ? (keep_a ? a.dz_last : a.act[l]) + gbase : nullptr;
      const float w0 = a.w0;

      // the layer's operand: k-step ks contracts features 16 ks + 4 lh + (j & 3) + 8 (j >> 2) = registers 8 (ks & 1) + j of tile ks / 2
      // (only the first region's two fragments here: the other fourteen are split in the gaps of tile 0's MFMAs, a region
      // ahead of their use -- tile 0 has no epilogue to carry, and the split of all sixteen in front of the tile loop was
      // 5 k cycles per layer with the matrix pipe idle)
      x3::Frag cur[kSteps];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = out[ks >> 1][8 * (ks & 1) + j];
        cur[ks] = x3::split8(v);
        // pinned to the accumulation registers (the MFMA reads its B operand there): left to itself the allocator keeps
        // these 192 registers as VGPRs, spills them to AGPRs and copies every fragment back in front of its MFMAs
        asm("" : "+a"(cur[ks].h), "+a"(cur[ks].m), "+a"(cur[ks].l));
      }

      RP_MARK(1)  // the operand's split
      // One accumulator per tile (two tiles in flight: the epilogue of tile i - 1 reads its accumulator while tile i fills
      // the other); a single chain of this MFMA issues at its full rate (MI355X_MICROARCH.md; two interleaved chains
      // measured the same).  A region = the 12 MFMAs of two k-steps.  WHEN the fragments are requested matters: hipcc
      // turns every LDS wait into lgkmcnt(0) while an LDS-DMA is in flight (always, here), so a request issued just in
      // front of a wait is paid in full by the MFMA behind that wait.  The second k-step's fragment is requested behind
      // the region's first MFMA (six MFMAs before its use), the NEXT region's first fragment behind the seventh: both
      // waits (in front of MFMA 1 and MFMA 7) then find only requests that are five MFMAs old.
      f32x16 acc[2];
      x3::Frag wf[2];
      wf[0] = read_wfrag(ring0 + wslot[0]);  // chunk 0 of the layer sits in ring slot 0 (16 chunks per layer)
      float bnx = 0.f, bny = 0.f;  // the next slice's bias pair, requested half a region ahead
      float amax = 0.f;  // largest |argument| of the tile whose epilogue is running
      float smax = 0.f;  // largest |sine| (never above 1: see the epilogue)
      float dd[4];       // w0 cos of the quad being assembled

      static_for<kTiles + 1>([&](auto i_c) {  // iteration i: MFMAs of tile i, epilogue of tile i - 1
        constexpr int i = decltype(i_c)::value;
        static_for<2>([&](auto hf_c) {
          constexpr int hf = decltype(hf_c)::value;
          constexpr int p = 2 * i + hf;  // chunk of the layer
          static_for<4>([&](auto rg_c) {
            constexpr int rg = decltype(rg_c)::value;
            constexpr int ks = kChunkSteps * hf + 2 * rg;  // this region's k-steps: ks, ks + 1
            if constexpr (i < kTiles && rg == 3) {
              // ---- chunk p + 1 has landed for everyone, chunk p - 1's slot is free (chunk p + 3 goes there, below) ----
              __builtin_amdgcn_sched_barrier(0);
              RP_MARK(2)  // MFMAs + interleaved epilogue
              // Chunk p + 1 was queued three syncs ago; what this wave has queued since (never fewer): chunk p + 2's six
              // pieces and, in the training forms, the two stores of each half-row flush in between (slices 1, 2 from the
              // layer's third tile on, slices 5, 6 from its second).  Counting them keeps the wait off stores that are a
              // region old (a plain vmcnt(6) made every sync wait for those to be acknowledged).
              constexpr int younger = 6 + (STORE ? (i >= 2 ? 8 : (i == 1 && rg == 3 && hf == 1 ? 4 : 0)) : 0);
              if constexpr (younger == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
              if constexpr (younger == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
              if constexpr (younger == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              RP_MARK(3)  // chunk wait + barrier
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < kTiles && rg == 3) {
              // Always queued, also behind the kernel's last chunk (the first layer's chunks again, never read): the
              // stream has no tail, every wait above is the same count, no branch cuts this region.
              constexpr int q = p + 3;
              issue(q < kChunksPerLayer ? mat : mat_next, (q & (kChunksPerLayer - 1)) >> 1, q & 1, sm.ring[q & 3], wave, lane);
            }
            // ---- the region: the 12 MFMAs of k-steps ks and ks + 1, and in the gaps between them the epilogue slice of
            // tile i - 1 (accumulator registers 2 pj, 2 pj + 1: bias, w0, the sincos of sincos_fast2_core operation for
            // operation, the derivative, the stores).  The order is written out and fenced gap by gap: a
            // sched_group_barrier pipeline was dropped by the scheduler whenever it judged the interleaved order's
            // register pressure too high (back to all-VALU-then-all-MFMA).  Everything is SCALAR f32: one v_pk_fma_f32
            // beside MFMAs costs 22 cycles more than two v_fma_f32 (MI355X_MICROARCH.md), and a gap hides five to six
            // 4-cycle instructions -- 54 (inference) / 68 (training) of them per slice over 12 gaps.
            constexpr int ti = i > 0 ? i - 1 : 0, pj = 4 * hf + rg, r0 = 2 * pj;
            constexpr bool mm = i < kTiles, ep = i > 0;
            constexpr bool first = hf == 0 && rg == 0;
            constexpr bool next_chunk = rg == 3;
            constexpr bool have_next = mm && !(next_chunk && p + 1 >= kChunksPerLayer);  // the next layer requests its own
            // the next region's first fragment: k-step 2 rg + 2 of this chunk, or 0 of the next (synced above)
            const char* nA = ring0 + wslot[(p + (next_chunk ? 1 : 0)) & 3] + 3 * kPiece * (next_chunk ? 0 : 2 * rg + 2);
            const char* tB = ring0 + wslot[p & 3] + 3 * kPiece * (2 * rg + 1);  // this region's second fragment
            float bx = 0.f, by = 0.f, ux = 0.f, uy = 0.f, kx = 0.f, ky = 0.f, rx = 0.f, ry = 0.f, zx = 0.f, zy = 0.f;
            float psx = 0.f, psy = 0.f, pcx = 0.f, pcy = 0.f, hx = 0.f, hy = 0.f, snx = 0.f, sny = 0.f, csx = 0.f, csy = 0.f;
            float ax = 0.f, ay = 0.f, sx = 0.f, sy = 0.f, cx = 0.f, cy = 0.f;
            int qx = 0, qy = 0;
            f32x4 fl[2];  // a half row on its way out
            bool ex = false, ey = false, nx = false, ny = false;  // quadrant bits 0 / 1 clear
            constexpr int kOps = STORE ? 31 : 27;
            auto op = [&](auto n_c) {
              constexpr int n = decltype(n_c)::value;
              if constexpr (n == 0) ux = acc[ti & 1][r0], uy = acc[ti & 1][r0 + 1];
              if constexpr (n == 1) ux += bx, uy += by;
              if constexpr (n == 2) ux *= w0, uy *= w0;
              if constexpr (n == 3) amax = __builtin_fmaxf(__builtin_fmaxf(amax, fabsf(ux)), fabsf(uy));  // (one v_max3 with |.| modifiers)
              if constexpr (n == 4) kx = ux * 0.636619772367581343f, ky = uy * 0.636619772367581343f;
              if constexpr (n == 5) kx = rintf(kx), ky = rintf(ky);
              if constexpr (n == 6) qx = (int)kx, qy = (int)ky;
              if constexpr (n == 7) rx = __builtin_fmaf(kx, -1.57079625129699707031e+00f, ux), ry = __builtin_fmaf(ky, -1.57079625129699707031e+00f, uy);
              if constexpr (n == 8) rx = __builtin_fmaf(kx, -7.54978941586159635335e-08f, rx), ry = __builtin_fmaf(ky, -7.54978941586159635335e-08f, ry);
              if constexpr (n == 9) rx = __builtin_fmaf(kx, -5.39030252995776476554e-15f, rx), ry = __builtin_fmaf(ky, -5.39030252995776476554e-15f, ry);
              if constexpr (n == 10) zx = rx * rx, zy = ry * ry;
              if constexpr (n == 11) psx = __builtin_fmaf(-1.9515295891e-4f, zx, 8.3321608736e-3f), psy = __builtin_fmaf(-1.9515295891e-4f, zy, 8.3321608736e-3f);
              if constexpr (n == 12) pcx = __builtin_fmaf(2.443315711809948e-5f, zx, -1.388731625493765e-3f), pcy = __builtin_fmaf(2.443315711809948e-5f, zy, -1.388731625493765e-3f);
              if constexpr (n == 13) psx = __builtin_fmaf(psx, zx, -1.6666654611e-1f), psy = __builtin_fmaf(psy, zy, -1.6666654611e-1f);
              if constexpr (n == 14) pcx = __builtin_fmaf(pcx, zx, 4.166664568298827e-2f), pcy = __builtin_fmaf(pcy, zy, 4.166664568298827e-2f);
              if constexpr (n == 15) psx *= zx, psy *= zy;
              if constexpr (n == 16) pcx *= zx, pcy *= zy;
              if constexpr (n == 17) hx = __builtin_fmaf(-0.5f, zx, 1.0f), hy = __builtin_fmaf(-0.5f, zy, 1.0f);
              if constexpr (n == 18) snx = __builtin_fmaf(psx, rx, rx), sny = __builtin_fmaf(psy, ry, ry);
              if constexpr (n == 19) csx = __builtin_fmaf(pcx, zx, hx), csy = __builtin_fmaf(pcy, zy, hy);
              if constexpr (n == 20) ex = (qx & 1) == 0, ey = (qy & 1) == 0;
              // (the compares a slot ahead of the selects that read them: no wait states in between)
              if constexpr (n == 21) nx = (qx & 2) == 0, ny = (qy & 2) == 0;
              if constexpr (n == 22) ax = ex ? snx : csx, ay = ey ? sny : csy;
              if constexpr (n == 23) sx = nx ? ax : -ax, sy = ny ? ay : -ay;
              // (a use in front of the range check: nothing sinks behind its branch.  Of the SIGNED values: |s| does not need
              // the quadrant's sign, and the inference form then kept -- spilled -- sign bits and magnitudes for later)
              if constexpr (n == 25 && !STORE) smax = fmaxf(smax, fmaxf(sx, sy));  // (the training forms' stores are such a use)
              if constexpr (n == 24) out[ti][r0] = sx, out[ti][r0 + 1] = sy;
              // (26: a spare slot of the inference form)
              if constexpr (n == 27) ax = ex ? csx : snx, ay = ey ? csy : sny;
              if constexpr (n == 28) nx = ((qx + 1) & 2) == 0, ny = ((qy + 1) & 2) == 0;
              if constexpr (n == 29) cx = nx ? ax : -ax, cy = ny ? ay : -ay;
              if constexpr (n == 30) dd[r0 & 3] = w0 * cx, dd[(r0 & 3) + 1] = w0 * cy;
            };
            if constexpr (ep) bx = bnx, by = bny;
            static_for<12>([&](auto k_c) {
              constexpr int k = decltype(k_c)::value;
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (mm) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                constexpr int j = k / 6, t = k % 6, kk = mm ? ks + j : 0;
                const x3::Frag& w = wf[j];
                const x3::Frag& v = cur[kk];
                f32x16& c = acc[i & 1];
                if constexpr (t == 0) c = mfma(w.h, v.l, first && j == 0 ? zero : c);
                if constexpr (t == 1) c = mfma(w.l, v.h, c);
                if constexpr (t == 2) c = mfma(w.m, v.m, c);
                if constexpr (t == 3) c = mfma(w.h, v.m, c);
                if constexpr (t == 4) c = mfma(w.m, v.h, c);
                if constexpr (t == 5) c = mfma(w.h, v.h, c);
              }
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (k == 6) {
                // the bias pair of the NEXT slice (its first use is a region away: no wait catches the fragment reads
                // that follow it in the LDS queue)
                constexpr int npj = (pj + 1) & 7, nti = pj == 7 ? i : ti, nr0 = 2 * npj;
                if constexpr (nti < kTiles && (ep || pj == 7)) {
                  const f32x2 b01 = *reinterpret_cast<const f32x2*>(&sm.bias[l][32 * nti + (nr0 & 3) + 8 * (nr0 >> 2) + feat0]);
                  bnx = b01.x, bny = b01.y;
                }
              }
              if constexpr (STORE) {
                // The stores, a region behind the values.  A quad is complete at the end of an odd slice: it is staged
                // in the next region, behind the wait in front of MFMA 7 (no wait then sees a young LDS write); a half
                // row (two quads) leaves two regions later: a in one region, w0 cos in the next, read behind MFMA 2 and
                // stored behind MFMA 6 (the wait in front of the store finds requests that are four gaps old).
                constexpr int qt = pj == 0 ? i - 2 : ti, qq = pj == 0 ? 3 : (pj >> 1) - 1;  // the quad to stage here
                if constexpr (k == 7 && (pj & 1) == 0 && qt >= 0 && (pj > 0 || i >= 2)) {
                  stage(0, qq, f32x4{out[qt][4 * qq], out[qt][4 * qq + 1], out[qt][4 * qq + 2], out[qt][4 * qq + 3]});
                  stage(1, qq, f32x4{dd[0], dd[1], dd[2], dd[3]});
                }
                // flushes: half 0 of tile ti in slices 5 (a) and 6 (w0 cos); half 1 of tile i - 2 in slices 1 and 2
                constexpr bool f0 = ep && (pj == 5 || pj == 6), f1 = i >= 2 && (pj == 1 || pj == 2);
                constexpr int ft = f0 ? ti : i - 2, fh = f0 ? 0 : 1, fa = (pj == 5 || pj == 1) ? 0 : 1;
                if constexpr ((f0 || f1) && k == 1) unstage(fa, fh, fl);
                if constexpr ((f0 || f1) && k == 5) {
                  if constexpr (fa)
                    put(gd, ft, fh, fl);
                  else
                    put(ga, ft, fh, fl);
                }
              }
              if constexpr (mm && k == 0) wf[1] = read_wfrag(tB);
              if constexpr (have_next && k == 6) wf[0] = read_wfrag(nA);
              if constexpr (i == 0 && 2 * pj + 2 < kSteps) {
                // ---- tile 0: the operand fragments of the NEXT region's two k-steps, a pair of values per gap ----------
                constexpr int sk = 2 * pj + 2 + (k >> 2 & 1), sq = k & 3;  // gaps 0-3: k-step 2 pj + 2, gaps 4-7: 2 pj + 3
                if constexpr (k < 8) {
                  uint32_t h, m, l2;
                  x3::split2(out[sk >> 1][8 * (sk & 1) + 2 * sq], out[sk >> 1][8 * (sk & 1) + 2 * sq + 1], h, m, l2);
                  cur[sk].h[sq] = h, cur[sk].m[sq] = m, cur[sk].l[sq] = l2;
                }
                if constexpr (k == 8) asm("" : "+a"(cur[2 * pj + 2].h), "+a"(cur[2 * pj + 2].m), "+a"(cur[2 * pj + 2].l));
                if constexpr (k == 9) asm("" : "+a"(cur[2 * pj + 3].h), "+a"(cur[2 * pj + 3].m), "+a"(cur[2 * pj + 3].l));
              }
#ifndef RR_EXPERIMENT_NO_EPILOGUE  // timing experiment only (results are then wrong)
              if constexpr (ep) static_for<kOps>([&](auto n_c) {
                if constexpr (decltype(n_c)::value * 12 / kOps == k) op(n_c);
              });
#else
              if constexpr (ep && k == 11) out[ti][r0] = acc[ti & 1][r0], out[ti][r0 + 1] = acc[ti & 1][r0 + 1];
#endif
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        if constexpr (i == kTiles) RP_MARK(4)  // the last tile's epilogue, alone
        if constexpr (i > 0) {
          // ---- range check of tile i - 1 (wave-uniform, never taken by a trained network) --------------------------
          constexpr int ti = i - 1;
          if (__builtin_amdgcn_ballot_w64(!(amax <= 8192.0f) || !(smax <= 2.0f)) != 0) {
            float* fix = &sm.stg[wave][0][0];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = 32 * ti + (r & 3) + 8 * (r >> 2) + feat0;
              float v = acc[ti & 1][r];
              asm volatile("" : "+v"(v));  // (opaque: or the slices' arguments are kept alive -- spilled -- for this path)
              fix[r * 64 + lane] = w0 * (v + sm.bias[l][f]);
            }
            repair_values(fix, 16);
#pragma unroll
            for (int r = 0; r < 16; ++r) out[ti][r] = fix[r * 64 + lane];
            if (STORE) {
              // where the normal path stands: quads 0, 1 have left (wrong), quad 2 is staged (these bytes: gone), quad 3 waits
              // in `out` / `dd`.  Quads 0, 1 again, straight from the registers; quad 2 staged again; dd corrected.
              float dq[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) dq[r] = w0 * fix[(16 + r) * 64 + lane];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<f32x4*>(ga + (row * kH - gbase) + feat0 + 32 * ti + 8 * q) =
                    f32x4{out[ti][4 * q], out[ti][4 * q + 1], out[ti][4 * q + 2], out[ti][4 * q + 3]};
                *reinterpret_cast<f32x4*>(gd + (row * kH - gbase) + feat0 + 32 * ti + 8 * q) =
                    f32x4{dq[4 * q], dq[4 * q + 1], dq[4 * q + 2], dq[4 * q + 3]};
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch has been read: its bytes are the image again
              stage(0, 2, f32x4{out[ti][8], out[ti][9], out[ti][10], out[ti][11]});
              stage(1, 2, f32x4{dq[8], dq[9], dq[10], dq[11]});
#pragma unroll
              for (int e = 0; e < 4; ++e) dd[e] = dq[12 + e];
            }
          }
          amax = 0.f, smax = 0.f;
        }
      });
      if (STORE) {  // the last tile's quad 3 and second half row
        constexpr int t = kTiles - 1;
        stage(0, 3, f32x4{out[t][12], out[t][13], out[t][14], out[t][15]});
        stage(1, 3, f32x4{dd[0], dd[1], dd[2], dd[3]});
        f32x4 va[2], vd[2];
        unstage(0, 1, va), unstage(1, 1, vd);
        put(ga, t, 1, va), put(gd, t, 1, vd);
      }
    }

    RP_MARK(2)
    // ---- head: y[row] = a_last[row] . w_last + b_last; the lane halves hold the two halves of the features -----------
    {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < kTiles; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(&sm.w_last[32 * i + 8 * q + feat0]);
          part += out[i][4 * q] * w4.x, part += out[i][4 * q + 1] * w4.y;
          part += out[i][4 * q + 2] * w4.z, part += out[i][4 * q + 3] * w4.w;
        }
      part += __shfl_xor(part, 32, 64);
      const float yv = part + b_last;
      if (lh == 0) a.y[row] = yv;
      if constexpr (LOSS) {
        // ---- loss (models.py:64 F.mse_loss: mean((y - target)^2); dLoss/dy = 2 (y - t) / N) and what of the head's
        // backward needs a: dW_head += dy^T a.  dy goes to the workspace: siren_backward_rows_kernel starts from it and
        // from the w0 cos this kernel left in dz_last (dz = dy w_head (.) w0 cos, db_last: there -- re-reading w0 cos
        // here, a tile per round trip, cost 0.6 ms at config 3).  Rows beyond n repeat row n - 1 and add nothing.
        const bool live = g * kRows + 32 * wave + l31 < a.n;
        const float diff = yv - sm.xin[wave][kMaxIn + gpar][lane], dyv = diff * a.grad_scale, dys = live ? dyv : 0.f;
        if (lh == 0) a.dy_ws[row] = dyv;
        if (live && lh == 0) g_loss += diff * diff, g_bhead += dyv;
        float pw = 0.f;  // the even tile's reduced sums, waiting for the odd one
#pragma unroll
        for (int i = 0; i < kTiles; ++i) {
          float P[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) P[r] = dys * out[i][r];
          // sums over the wave's 32 rows: a halving butterfly over the lanes (reduce16), then tile i - 1 / tile i over bit 4
          const float sw = reduce16(P, lane);
          if (i & 1)
            g_wh[i >> 1] += pair32(pw, sw);
          else
            pw = sw;
        }
        park();
      }
    }
    RP_MARK(5)  // head (+ loss tail)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunks queued behind the last one land before the LDS is released
  RP_END
  if constexpr (LOSS) {
    // ---- this workgroup's slab: dW_head [H] | db_last [H] (zero: the backward kernel's) | db_head, loss.  A lane holds, for j = 0 .. 3, the sums of
    // accumulator register (lane & 15) of tile 2 j + ((lane >> 4) & 1): feature 32 tile + (r & 3) + 8 (r >> 2) + 4 lh.
    __syncthreads();  // (the ring is free: every chunk has landed, every wave is past its last read)
    float* red = reinterpret_cast<float*>(&sm.ring[0][0]);  // [wave][2][H], then [2][4 waves]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = lane & 15, f = 32 * (2 * j + ((lane >> 4) & 1)) + (r & 3) + 8 * (r >> 2) + 4 * lh;
      red[(wave * 2 + 0) * kH + f] = g_wh[j];
    }
    // lane halves: every lane of a half added its own row's terms; lh == 0 lanes hold them (see above)
    float sb = g_bhead, sl = g_loss;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) sb += __shfl_xor(sb, off, 64), sl += __shfl_xor(sl, off, 64);
    if (lane == 0) red[8 * kH + wave] = sb, red[8 * kH + 4 + wave] = sl;
    __syncthreads();
    float* slab = a.partial + (int64_t)blockIdx.x * fwd_slab_floats(kH);
    {
      const int f = tid;  // 256 threads = H
      slab[f] = ((red[0 * kH + f] + red[2 * kH + f]) + red[4 * kH + f]) + red[6 * kH + f];
      slab[kH + f] = 0.f;
    }
    if (tid == 0) {
      slab[2 * kH] = ((red[8 * kH] + red[8 * kH + 1]) + red[8 * kH + 2]) + red[8 * kH + 3];
      slab[2 * kH + 1] = (((red[8 * kH + 4] + red[8 * kH + 5]) + red[8 * kH + 6]) + red[8 * kH + 7]) * a.inv_n;
      slab[2 * kH + 2] = 0.f, slab[2 * kH + 3] = 0.f;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward chain, same ownership: a wave's 32 rows of dz stay in registers from the last sine layer down to the first.
// Per H x H layer l: da_{l-1}^T = W_l^T dz_l^T (the split W^T planes are the MFMA's A operand, streamed through the same
// chunk ring; dz_l in the accumulator layout is the B operand after its split), dz_{l-1} = da_{l-1} (.) w0 cos_{l-1} in the
// gaps of the next tile's MFMAs -- the derivative tile arrives by LDS-DMA as the rows lie in HBM (16 lanes x 64 bytes per
// instruction) and is read back a row per lane; dz_{l-1} (l - 1 >= 1: the weight gradient kernel contracts it with
// a_{l-2} over the batch) leaves through the staging image.  Bias gradients = column sums over the rows = sums over
// the LANES: a halving butterfly per tile after the layer's tile loop, accumulated by sole owners in LDS per (layer, wave);
// the first layer's weight gradient dz_0^T x likewise, in registers.  Serves the training path only: the head's backward
// was done by the loss-mode forward kernel (head_done), dim_in <= 4.
struct SmemB {
  float gb[kMaxSine - 1][4][kH];  // bias-gradient column sums of sine layers 0 .. L-2 per wave
  float w_last[kH];
  float stg[4][32 * kStgLd];      // dz of a tile on its way out
  char din[4][4096];              // w0 cos of a tile on its way in: [half][16-row block][16 rows][64 bytes]
  char ring[kRing][kChunk] __attribute__((aligned(16)));
};

__global__ __launch_bounds__(kThreadsR) void siren_backward_rows_kernel(const BwdArgs a) {
  __shared__ SmemB sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int L = a.n_sine, n_mm = L - 1;  // >= 1 H x H layers
  for (int e = tid; e < (kMaxSine - 1) * 4 * kH; e += kThreadsR) (&sm.gb[0][0][0])[e] = 0.f;
  for (int e = tid; e < kH; e += kThreadsR) sm.w_last[e] = a.w[L][e];
  const int64_t groups = (a.n + kRows - 1) / kRows;
  int wslot[kRing];
#pragma unroll
  for (int q = 0; q < kRing; ++q) {
    wslot[q] = q * kChunk + 32 * l31 + 16 * (lh ^ ((l31 >> 3) & 1));
    asm volatile("" : "+v"(wslot[q]));
  }
  const char* const ring0 = &sm.ring[0][0];
  const int feat0 = 4 * lh;
  const char* const mat_top = a.wtsplit + (int64_t)(L - 2) * split_matrix_bytes(kH);  // W^T of the last sine layer: a group's first

  if ((int64_t)blockIdx.x < groups) {
    issue(mat_top, 0, 0, sm.ring[0], wave, lane);
    issue(mat_top, 0, 1, sm.ring[1], wave, lane);
    issue(mat_top, 1, 0, sm.ring[2], wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // first layer's weight gradient [input d][pair of tiles] and the last sine layer's bias gradient, parked in
  // accumulation registers between their phases
  float g_wf[4][4], g_bl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int j = 0; j < 4; ++j) g_wf[d][j] = 0.f;
  auto park = [&]() {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(g_wf[d][j]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(g_bl[j]));
  };
  park();
  // this lane's piece of the derivative image: DMA side (16 lanes x 64 bytes per instruction) and read side (its row)
  float* const stg_w = &sm.stg[wave][0];
  char* const din_w = &sm.din[wave][0];
  const int wr = l31 * kStgLd + feat0, rd = (lane >> 2) * kStgLd + 4 * (lane & 3);
  const int din_rd = (l31 >> 4) * 1024 + (l31 & 15) * 64 + 16 * lh;  // + 2048 (q >> 1) + 32 (q & 1)

  // A group's first requests -- its rows' dLoss/dy and x, the first two tiles of w0 cos of the last sine layer (by LDS-DMA
  // into this wave's derivative image and its share of ring slot 3) -- are made a phase ahead: for the first group here, for
  // the others in front of the previous group's first-layer phase, when both images are free.  (Made at the group's own
  // start they cost a full round trip each, the dy load behind a vmcnt(0) that also drains the previous group's stores.)
  float dy_pre = 0.f, x_pre[4] = {0.f, 0.f, 0.f, 0.f};
  char* const buf1 = &sm.ring[3][0] + 4096 * wave;
  auto fetch_tile = [&](const int64_t (&off)[2], int i) {
    char* dst = (i & 1) ? buf1 : din_w;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a.dz[n_mm] + off[e & 1] + 32 * i + 16 * (e >> 1)),
          (__attribute__((address_space(3))) void*)(dst + 1024 * e), 16, 0, 0);
  };
  auto prefetch = [&](int64_t g) {
    const int64_t row = std::min<int64_t>(g * kRows + 32 * wave + l31, a.n - 1);
    dy_pre = a.dy_ws[row];
#pragma unroll
    for (int d = 0; d < 4; ++d) x_pre[d] = d < a.dim_in ? a.x[row * a.dim_in + d] : 0.f;
    int64_t off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      off[j] = std::min<int64_t>(g * kRows + 32 * wave + (lane >> 2) + 16 * j, a.n - 1) * kH + 4 * (lane & 3);
    fetch_tile(off, 0), fetch_tile(off, 1);
  };
  if ((int64_t)blockIdx.x < groups) prefetch(blockIdx.x);
  RP_BEGIN
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t row = std::min<int64_t>(g * kRows + 32 * wave + l31, a.n - 1);
    const bool live = g * kRows + 32 * wave + l31 < a.n;
    int64_t goff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      goff[j] = std::min<int64_t>(g * kRows + 32 * wave + (lane >> 2) + 16 * j, a.n - 1) * kH + 4 * (lane & 3);
    float xv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) xv[d] = x_pre[d];

    // ---- the head: dz_{L-1} = dy w_head (.) w0 cos.  The loss-mode forward kernel left w0 cos of the last sine layer
    // in dz[L-1] and dLoss/dy in the workspace; the tiles come in by LDS-DMA, two in flight (this wave's derivative
    // image and its share of ring slot 3, free until the group's first chunk sync; the first two were queued a phase ago),
    // and leave as dz the way they came.
    float out[kTiles][16];
    {
      const float dyv = dy_pre;  // (its wait also covers the two tiles queued with it: a phase old by now)
      float* const gzs = a.dz[n_mm];
      auto fetch = [&](int i) { fetch_tile(goff, i); };
      float pb = 0.f;
#pragma unroll
      for (int i = 0; i < kTiles; ++i) {
        // tile i has landed: behind it in the queue the four pieces of tile i + 1 (and stores of tile i - 1, maybe)
        if (i + 1 < kTiles)
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* src = ((i & 1) ? buf1 : din_w) + din_rd;
        f32x4 dq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dq[q] = *reinterpret_cast<const f32x4*>(src + 2048 * (q >> 1) + 32 * (q & 1));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image's bytes are free
        if (i + 2 < kTiles) fetch(i + 2);
        float Q[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(&sm.w_last[32 * i + 8 * q + feat0]);
          f32x4 z;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            z[e] = (dyv * w4[e]) * dq[q][e];
            out[i][4 * q + e] = z[e];
            Q[4 * q + e] = live ? z[e] : 0.f;
          }
          *reinterpret_cast<f32x4*>(stg_w + wr + 8 * q) = z;
        }
        // (only rows below n are stored: a repeated row may find dz where it expects w0 cos -- another wave owns that row)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg_w + rd + 16 * h + 16 * j * kStgLd);
            if (g * kRows + 32 * wave + (lane >> 2) + 16 * j < a.n) *reinterpret_cast<f32x4*>(gzs + goff[j] + 32 * i + 16 * h) = v;
          }
        const float sb = reduce16(Q, lane);  // db_{L-1} += colsum(dz)
        if (i & 1)
          g_bl[i >> 1] += pair32(pb, sb);
        else
          pb = sb;
      }
      park();
    }

    RP_MARK(0)  // head phase
    for (int l = n_mm; l >= 1; --l) {
      const char* mat = a.wtsplit + (int64_t)(l - 1) * split_matrix_bytes(kH);
      const char* mat_next = l > 1 ? mat - split_matrix_bytes(kH) : mat_top;
      const float* __restrict__ gd = a.deriv[l - 1];
      // dz_{l-1} leaves for l - 1 >= 1; for l - 1 = 0 the same stores go to a dummy (the head of this workgroup's slab,
      // written again at the kernel's end): no branch in the tile loop
      const bool keep = l - 1 >= 1;
      float* const zbase = keep ? a.dz[l - 1] : a.partial + (int64_t)blockIdx.x * bwd_slab_floats(kH, L);
      const int tmul = keep ? 1 : 0;
      int64_t zoff[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) zoff[j] = keep ? goff[j] : (int64_t)(4 * lane + 256 * j);

      // (only the first region's two fragments here: the other fourteen are split in the gaps of tile 0's MFMAs, a region
      // ahead of their use -- tile 0 has no epilogue to carry, and the split of all sixteen in front of the tile loop was
      // 5 k cycles per layer with the matrix pipe idle)
      x3::Frag cur[kSteps];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = out[ks >> 1][8 * (ks & 1) + j];
        cur[ks] = x3::split8(v);
        asm("" : "+a"(cur[ks].h), "+a"(cur[ks].m), "+a"(cur[ks].l));
      }

      RP_MARK(1)  // the operand's split
      f32x16 acc[2];
      x3::Frag wf[2];
      wf[0] = read_wfrag(ring0 + wslot[0]);
      f32x4 dv[4];  // w0 cos of the tile whose epilogue is running, a row per lane
      f32x4 fl[2];  // a half row of dz on its way out

      static_for<kTiles + 1>([&](auto i_c) {  // iteration i: MFMAs of tile i, epilogue of tile i - 1
        constexpr int i = decltype(i_c)::value;
        static_for<2>([&](auto hf_c) {
          constexpr int hf = decltype(hf_c)::value;
          constexpr int p = 2 * i + hf;
          static_for<4>([&](auto rg_c) {
            constexpr int rg = decltype(rg_c)::value;
            constexpr int ks = kChunkSteps * hf + 2 * rg;
            constexpr int ti = i > 0 ? i - 1 : 0, pj = 4 * hf + rg, r0 = 2 * pj;
            constexpr bool mm = i < kTiles, ep = i > 0;
            constexpr bool first = hf == 0 && rg == 0;
            constexpr bool next_chunk = rg == 3;
            constexpr bool have_next = mm && !(next_chunk && p + 1 >= kChunksPerLayer);
            if constexpr (mm && rg == 3) {
              // chunk p + 1 has landed for everyone, chunk p - 1's slot is free
              __builtin_amdgcn_sched_barrier(0);
              RP_MARK(2)  // MFMAs + epilogue
              // (what this wave has queued behind chunk p + 1, never fewer: chunk p + 2's six pieces, the four of a
              // derivative tile, and the two stores of each half-row flush in between)
              constexpr int younger = 10 + (i >= 2 ? 4 : (i == 1 && hf == 1 ? 2 : 0));
              if constexpr (younger == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
              if constexpr (younger == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
              if constexpr (younger == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              RP_MARK(3)  // chunk wait + barrier
            }
            if constexpr (ep && first) {
              // the derivative tile of tile i - 1 has landed (queued an iteration ago, twelve or more requests behind it)
              __builtin_amdgcn_sched_barrier(0);
              RP_MARK(2)
              asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
              RP_MARK(6)  // derivative tile wait
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (mm && rg == 3) {
              constexpr int q = p + 3;
              issue(q < kChunksPerLayer ? mat : mat_next, (q & (kChunksPerLayer - 1)) >> 1, q & 1, sm.ring[q & 3], wave, lane);
            }
            const char* nA = ring0 + wslot[(p + (next_chunk ? 1 : 0)) & 3] + 3 * kPiece * (next_chunk ? 0 : 2 * rg + 2);
            const char* tB = ring0 + wslot[p & 3] + 3 * kPiece * (2 * rg + 1);
            static_for<12>([&](auto k_c) {
              constexpr int k = decltype(k_c)::value;
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (mm) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                constexpr int j = k / 6, t = k % 6, kk = mm ? ks + j : 0;
                const x3::Frag& w = wf[j];
                const x3::Frag& v = cur[kk];
                f32x16& c = acc[i & 1];
                if constexpr (t == 0) c = mfma(w.h, v.l, first && j == 0 ? zero : c);
                if constexpr (t == 1) c = mfma(w.l, v.h, c);
                if constexpr (t == 2) c = mfma(w.m, v.m, c);
                if constexpr (t == 3) c = mfma(w.h, v.m, c);
                if constexpr (t == 4) c = mfma(w.m, v.h, c);
                if constexpr (t == 5) c = mfma(w.h, v.h, c);
              }
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (mm && k == 0) wf[1] = read_wfrag(tB);
              if constexpr (have_next && k == 6) wf[0] = read_wfrag(nA);
              if constexpr (i == 0 && 2 * pj + 2 < kSteps) {
                // ---- tile 0: the operand fragments of the NEXT region's two k-steps, a pair of values per gap ----------
                constexpr int sk = 2 * pj + 2 + (k >> 2 & 1), sq = k & 3;  // gaps 0-3: k-step 2 pj + 2, gaps 4-7: 2 pj + 3
                if constexpr (k < 8) {
                  uint32_t h, m, l2;
                  x3::split2(out[sk >> 1][8 * (sk & 1) + 2 * sq], out[sk >> 1][8 * (sk & 1) + 2 * sq + 1], h, m, l2);
                  cur[sk].h[sq] = h, cur[sk].m[sq] = m, cur[sk].l[sq] = l2;
                }
                if constexpr (k == 8) asm("" : "+a"(cur[2 * pj + 2].h), "+a"(cur[2 * pj + 2].m), "+a"(cur[2 * pj + 2].l));
                if constexpr (k == 9) asm("" : "+a"(cur[2 * pj + 3].h), "+a"(cur[2 * pj + 3].m), "+a"(cur[2 * pj + 3].l));
              }
              // ---- the derivative image: read for tile i - 1 (first region, behind MFMA 2), then -- its bytes are free --
              // the next tile's four pieces queued (behind MFMA 6)
              if constexpr (ep && first && k == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dv[q] = *reinterpret_cast<const f32x4*>(din_w + din_rd + 2048 * (q >> 1) + 32 * (q & 1));
              }
              if constexpr (mm && first && k == 5) {
                if constexpr (ep) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int e = 0; e < 4; ++e) {  // piece e: half e >> 1, rows 16 (e & 1) + (lane >> 2), 16 bytes (lane & 3)
                  __builtin_amdgcn_global_load_lds(
                      (const __attribute__((address_space(1))) void*)(gd + goff[e & 1] + 32 * i + 16 * (e >> 1)),
                      (__attribute__((address_space(3))) void*)(din_w + 1024 * e), 16, 0, 0);
                }
              }
              // ---- the epilogue slice: dz_{l-1} = da (.) w0 cos, accumulator registers 2 pj, 2 pj + 1 (behind MFMA 9)
              if constexpr (ep && k == 8) {
                out[ti][r0] = acc[ti & 1][r0] * dv[r0 >> 2][r0 & 3];
                out[ti][r0 + 1] = acc[ti & 1][r0 + 1] * dv[r0 >> 2][(r0 & 3) + 1];
              }
              // ---- the stores, as the forward kernel's: a quad staged a region behind its values, a half row out two
              // regions later
              constexpr int qt = pj == 0 ? i - 2 : ti, qq = pj == 0 ? 3 : (pj >> 1) - 1;
              if constexpr (k == 7 && (pj & 1) == 0 && qt >= 0 && (pj > 0 || i >= 2))
                *reinterpret_cast<f32x4*>(stg_w + wr + 8 * qq) =
                    f32x4{out[qt][4 * qq], out[qt][4 * qq + 1], out[qt][4 * qq + 2], out[qt][4 * qq + 3]};
              constexpr bool f0 = ep && pj == 5, f1 = i >= 2 && pj == 1;
              constexpr int ft = f0 ? ti : i - 2, fh = f0 ? 0 : 1;
              if constexpr ((f0 || f1) && k == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j) fl[j] = *reinterpret_cast<const f32x4*>(stg_w + rd + 16 * fh + 16 * j * kStgLd);
              }
              if constexpr ((f0 || f1) && k == 5) {
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(zbase + zoff[j] + (32 * ft + 16 * fh) * tmul) = fl[j];
              }
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      });
      RP_MARK(2)
      {  // the last tile's quad 3 and second half row
        constexpr int t = kTiles - 1;
        *reinterpret_cast<f32x4*>(stg_w + wr + 24) = f32x4{out[t][12], out[t][13], out[t][14], out[t][15]};
#pragma unroll
        for (int j = 0; j < 2; ++j) fl[j] = *reinterpret_cast<const f32x4*>(stg_w + rd + 16 + 16 * j * kStgLd);
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(zbase + zoff[j] + (32 * t + 16) * tmul) = fl[j];
      }
      // ---- bias gradient of sine layer l - 1: column sums of dz_{l-1} over the wave's live rows ----------------------
      {
#pragma unroll
        for (int j = 0; j < kTiles / 2; ++j) {
          float Qa[16], Qb[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) Qa[r] = live ? out[2 * j][r] : 0.f, Qb[r] = live ? out[2 * j + 1][r] : 0.f;
          const int r = lane & 15, f = 32 * (2 * j + ((lane >> 4) & 1)) + (r & 3) + 8 * (r >> 2) + feat0;
          sm.gb[l - 1][wave][f] += pair32(reduce16(Qa, lane), reduce16(Qb, lane));  // sole owner of (layer, wave, feature)
        }
      }
    }

    RP_MARK(4)  // tail stores + bias gradient sums
    if (g + gridDim.x < groups) {  // (workgroup-uniform)
      // ring slot 3 held the layer's last chunk: every wave is done reading it behind this barrier (one per group); the
      // wave's own derivative image is free since its last tile was read
      __builtin_amdgcn_s_barrier();
      prefetch(g + gridDim.x);
    }
    // ---- first layer: dW_first[k][d] += sum_rows dz_0[row][k] x[row][d] ----------------------------------------------
    {
      // (inputs beyond dim_in are zeros: the first three always, the fourth behind ONE wave-uniform branch -- a branch per
      // butterfly kept the compiler from interleaving the independent ones, and a DPP chain alone is mostly wait states)
      auto input = [&](int d) {
#pragma unroll
        for (int j = 0; j < kTiles / 2; ++j) {
          float Pa[16], Pb[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) Pa[r] = live ? out[2 * j][r] * xv[d] : 0.f, Pb[r] = live ? out[2 * j + 1][r] * xv[d] : 0.f;
          g_wf[d][j] += pair32(reduce16(Pa, lane), reduce16(Pb, lane));
        }
      };
      input(0), input(1), input(2);
      if (a.dim_in > 3) input(3);
      park();
    }
    RP_MARK(5)  // first layer's weight gradient
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RP_END_TO(g_rows_profile_bwd)

  // ---- this workgroup's slab: dW_head [H] | db_head [4] | db_l [L][H] | dW_first [H][kMaxIn] ---------------------------------
  // (head weight and head bias: the forward kernel's; zeros here)
  float* slab = a.partial + (int64_t)blockIdx.x * bwd_slab_floats(kH, L);
  __syncthreads();
  float* red = reinterpret_cast<float*>(&sm.ring[0][0]);  // [wave][4 inputs + last bias][H]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = lane & 15, f = 32 * (2 * j + ((lane >> 4) & 1)) + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
    for (int d = 0; d < 4; ++d) red[(wave * 5 + d) * kH + f] = g_wf[d][j];
    red[(wave * 5 + 4) * kH + f] = g_bl[j];
  }
  __syncthreads();
  {
    const int f = tid;  // 256 threads = H
    slab[f] = 0.f;
    if (f < 4) slab[kH + f] = 0.f;
    float* p_b = slab + kH + 4;
    for (int l = 0; l + 1 < L; ++l)
      p_b[l * kH + f] = ((sm.gb[l][0][f] + sm.gb[l][1][f]) + sm.gb[l][2][f]) + sm.gb[l][3][f];
    p_b[(L - 1) * kH + f] = ((red[(0 * 5 + 4) * kH + f] + red[(1 * 5 + 4) * kH + f]) + red[(2 * 5 + 4) * kH + f]) + red[(3 * 5 + 4) * kH + f];
    float* p_wf = p_b + L * kH;
#pragma unroll
    for (int d = 0; d < kMaxIn; ++d)
      p_wf[f * kMaxIn + d] =
          d < 4 ? ((red[(0 * 5 + d) * kH + f] + red[(1 * 5 + d) * kH + f]) + red[(2 * 5 + d) * kH + f]) + red[(3 * 5 + d) * kH + f]
                : 0.f;
  }
}

}  // namespace rr
}  // namespace

int rows_blocks(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, rr::kRows), 256); }
bool rows_supported(int hidden, int n_sine) { return hidden == rr::kH && n_sine >= 2; }
// dLoss/dy per row between the loss-mode forward and the backward kernel: behind both kernels' slabs in the slab region
int64_t rows_dy_offset(int64_t n, int hidden, int n_sine) {
  const int64_t slabs = (int64_t)rows_blocks(n) * std::max(fwd_slab_floats(hidden), bwd_slab_floats(hidden, n_sine)) * 4;
  return (slabs + 255) / 256 * 256;
}

int forward_rows(const ChainArgs& a, int mode, hipStream_t st) {
  const int blocks = rows_blocks(a.n);  // one workgroup per CU
  if (mode == 2)
    hipLaunchKernelGGL((rr::siren_forward_rows_kernel<2>), dim3(blocks), dim3(rr::kThreadsR), 0, st, a);
  else if (mode == 1)
    hipLaunchKernelGGL((rr::siren_forward_rows_kernel<1>), dim3(blocks), dim3(rr::kThreadsR), 0, st, a);
  else if (mode == 0)
    hipLaunchKernelGGL((rr::siren_forward_rows_kernel<0>), dim3(blocks), dim3(rr::kThreadsR), 0, st, a);
  else
    return fail(MRI_ERR_INVALID_ARGUMENT, "forward_rows: mode %d", mode);
  return check_launch("siren_forward_rows_kernel");
}

#ifdef SIREN_PROFILE
extern "C" int mri_debug_set_rows_profile(long long* device_buffer, int backward) {
  const hipError_t e = backward ? hipMemcpyToSymbol(HIP_SYMBOL(rr::g_rows_profile_bwd), &device_buffer, sizeof(device_buffer))
                                : hipMemcpyToSymbol(HIP_SYMBOL(rr::g_rows_profile), &device_buffer, sizeof(device_buffer));
  return e == hipSuccess ? 0 : -1;
}
#endif

bool rows_backward_supported(int hidden, int n_sine, int dim_in, int head_done) {
  return hidden == rr::kH && n_sine >= 2 && dim_in <= 4 && head_done != 0;
}

int backward_rows(const BwdArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(rr::siren_backward_rows_kernel, dim3(rows_blocks(a.n)), dim3(rr::kThreadsR), 0, st, a);
  return check_launch("siren_backward_rows_kernel");
}

}  // namespace mri
