// Fused sine-MLP (SIREN) chain kernels for gfx950: a row tile's activations stay ON CHIP across
// all layers.
//
// Replaces, for SirenNet(dim_in <= 8 -> H x n -> 1) with H in {32, 64, 128, 256} (reference
// models.py:153-156 SirenLayer.forward, :230-233 SirenNet.forward: `F.linear` + `sin(w0 .)` per
// layer, then the linear head), the layer-by-layer GEMM launches of linear.hip, which re-read
// every (n, H) activation from HBM as the next layer's operand and pay a prologue / epilogue
// bubble per output tile:
//
//   siren_forward_kernel   one persistent 512-thread workgroup per CU walks row tiles; per tile the
//     first layer (K = dim_in) runs on the VALU straight into an LDS activation image, every
//     H x H layer multiplies that image (A operand, ds_read_b128) with the layer's weights
//     streamed from L2 in 16-deep chunks by LDS-DMA (global_load_lds_dwordx4, double buffered,
//     XOR-swizzled on the source side so that the B-operand ds_read_b128 are conflict-free) on
//     the bf16 matrix pipe with three-term operands (f32-accurate, see MRI_SIREN_X3 below), applies
//     bias / w0 / sincos in registers and overwrites the image; the 1-wide head is a wave
//     reduction over the image.  For training the activation a_l = sin(.) and its derivative
//     w0 cos(.) leave for HBM once (the backward kernels need them); they are dripped out of
//     registers beside the NEXT layer's MFMAs, a chunk ahead of the next wait.
//   siren_backward_kernel  the same walk from the head down (see there).
//   siren_wgrad_kernel     one H x H weight gradient, the whole result in MFMA accumulators.
//
// Data layout: activations / derivatives row-major (n, H) per layer, as linear.hip writes them,
// so the layer-wise kernels and the chain kernels are interchangeable per layer.
#include <algorithm>

#include "bf16x3.h"
#include "common.h"
#include "device_math.h"
#include "siren_chain.h"

// The H x H products run on the bf16 matrix pipe, f32-accurate (bf16x3.h): every operand is split
// exactly into three bf16 terms and a 16-deep step is six v_mfma_f32_32x32x16_bf16 -- 6/16 of the f32
// MFMA's cycles, and VALU work (sincos, the splits) runs beside a bf16 MFMA, which it cannot beside
// the f32 one.  The WEIGHTS are split once per call by siren_split_weights_kernel into chunk-major
// term planes that the chunk DMA streams as they are (both orientations: W for the forward products,
// W^T for dz W); the ACTIVATIONS are split by the consuming lane right after its LDS read (11 VALU
// instructions per pair: a pre-split image would not fit beside the weight chunks).  The accumulator
// layout of the 32x32x16 bf16 MFMA is that of the 32x32x2 f32 one, so epilogues are unchanged.
// The batch-contracting weight gradient splits both operands after the read (MRI_SIREN_X3 = 0
// builds its f32-MFMA form for A/B runs).
#ifndef MRI_SIREN_X3
#define MRI_SIREN_X3 1
#endif

namespace mri {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 512;  // 8 waves
constexpr int kWr = 32;        // batch rows per chunk of the weight-gradient kernel

// Geometry for hidden width H: a wave owns a 32 x CT tile of the (rows x H) layer output, the 8
// waves are RB row blocks x CB column blocks, so narrower networks take taller tiles and every
// shape keeps ~64 KiB of activation image and 32 MFMAs per wave and chunk (16 for H = 32).
template <int HH>
struct Shape {
  static constexpr int H = HH;
  static constexpr int CT = H < 64 ? H : 64;        // columns of a wave's tile
  static constexpr int NT = CT / 32;                // 32 x 32 MFMA tiles per wave
  static constexpr int CB = H / CT, RB = 8 / CB;    // column / row blocks of the 8 waves
  static constexpr int rows = 32 * RB;              // rows of a tile: 64, 128, 256, 256
  static constexpr int ld = H + 4;                  // image row stride: rows 4 banks apart (mod 64)
  static constexpr int chunks = H / kKc;            // weight chunks per layer
  static constexpr int chunk_bytes = 3 * H * 32;    // a weight chunk: three term planes of [H][16] bf16
  static constexpr int groups = kThreads / H;       // row groups of the (column, row group) phases
  static constexpr int rpt = rows / groups;         // rows per thread there: 32 (16 for H = 32)
  static constexpr int drip = 16 / chunks;          // accumulator registers dripped per chunk
  static_assert(H == 32 || H == 64 || H == 128 || H == 256, "hidden width");
};

template <class S>
struct FwdSmem {
  char wbuf[2][S::chunk_bytes] __attribute__((aligned(16)));  // weight chunks: term planes [n][2 slots of 8 bf16]
  float img[S::rows * S::ld];       // activation image of the tile
  float xs[S::rows * kMaxIn];
  float bias[kMaxSine][S::H];
  float w_last[S::H];
  float tgt[S::rows];               // loss mode: the tile's targets, then dLoss / dy
};

// row of register r of a 32x32 accumulator: (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// acc += A B over a 16-deep step, operands in their three bf16 terms: the six products, smallest first
__device__ __forceinline__ f32x16 mfma32x3(const x3::u32x4& a, const x3::u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3::bf16x8, a),
                                                 __builtin_bit_cast(x3::bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma6_32(const x3::Frag& a, const x3::Frag& b, f32x16 c) {
  c = mfma32x3(a.l, b.h, c);
  c = mfma32x3(a.h, b.l, c);
  c = mfma32x3(a.m, b.m, c);
  c = mfma32x3(a.m, b.h, c);
  c = mfma32x3(a.h, b.m, c);
  c = mfma32x3(a.h, b.h, c);
  return c;
}
__device__ __forceinline__ x3::Frag split_octets(const float4& lo, const float4& hi) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return x3::split8(v);
}

// Queue the LDS-DMA of chunk kc of one split matrix into `dst`: 16-byte slots, lane = slot.  The
// slot a lane FETCHES is its LDS slot with the half bit XORed by bit 3 of the row, the involution
// the fragment reads undo: the 16 lanes of a ds_read_b128 group then touch 16 different slots.
template <class S>
__device__ __forceinline__ void issue_chunk(const char* __restrict__ wsplit, int kc, char* dst,
                                            int wave, int lane) {
  constexpr int slots = 6 * S::H;  // 3 planes x H rows x 2
  const char* src = wsplit + (int64_t)kc * S::chunk_bytes;
#pragma unroll
  for (int i = 0; i < (slots + kThreads - 1) / kThreads; ++i) {
    const int base = (wave + 8 * i) * 64;
    if (slots % kThreads != 0 && base >= slots) break;
    const int slot = base + lane, n = (slot >> 1) % S::H;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(src + 16 * (slot ^ ((n >> 3) & 1))),
        (__attribute__((address_space(3))) void*)(dst + 16 * base), 16, 0, 0);
  }
}

// One 16-deep chunk of acc[t] += img[rows][k] * W[cols_t][k] for the wave's 32 x CT tile.  boff[t]: byte offset of
// the lane's slot of its column 32 t in a term plane.  The image operand is software-pipelined over the chunks of a
// layer (round 4): `fa` holds THIS chunk's fragment, already split (the image is complete when a layer starts, only
// the weights arrive chunk by chunk); the f32 words of the NEXT chunk's fragment (`a_next`: the lane's image row at
// that chunk's first k, + 4 lh; null for a layer's last chunk) are requested together with this chunk's weight
// fragments and split while this chunk's MFMAs execute.  Measured neutral against reading and splitting in front of
// the chunk's own MFMAs (config 3 12.82 against 12.81 ms, same flags, same box; EXPERIMENTS.md corrects the first
// claim); the weight fragments a chunk ahead as well (a three-deep DMA ring) measured slower.
template <int NT, int H>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NT], x3::Frag& fa, const float* __restrict__ a_next,
                                          const char* __restrict__ wb, const int (&boff)[NT]) {
  x3::Frag fb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    fb[t].h = *reinterpret_cast<const x3::u32x4*>(wb + boff[t]);
    fb[t].m = *reinterpret_cast<const x3::u32x4*>(wb + H * 32 + boff[t]);
    fb[t].l = *reinterpret_cast<const x3::u32x4*>(wb + 2 * H * 32 + boff[t]);
  }
  float4 n_lo = {0.f, 0.f, 0.f, 0.f}, n_hi = n_lo;
  if (a_next) n_lo = *reinterpret_cast<const float4*>(a_next), n_hi = *reinterpret_cast<const float4*>(a_next + 8);
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = mma6_32(fa, fb[t], acc[t]);
  if (a_next) fa = split_octets(n_lo, n_hi);
}
// Scheduling pattern for a region of N MFMAs with LDS reads and vector work to hide beside them: after each of the
// first four MFMAs READS_PER LDS reads, after each of the others VALU_PER vector instructions.
template <int N, int READS_PER, int VALU_PER, int I = 0>
__device__ __forceinline__ void sched_interleave() {
  if constexpr (I < N) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (I < 4)
      __builtin_amdgcn_sched_group_barrier(0x100, READS_PER, 0);
    else
      __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);
    sched_interleave<N, READS_PER, VALU_PER, I + 1>();
  }
}
// the first fragment of a layer (behind the barrier that completes the image)
__device__ __forceinline__ x3::Frag first_fragment(const float* __restrict__ a_k) {
  return split_octets(*reinterpret_cast<const float4*>(a_k), *reinterpret_cast<const float4*>(a_k + 8));
}

// Phase timing for tools/siren_phases.py (a tools-only build with -DSIREN_PROFILE; the shipped
// library compiles these to nothing): shader-clock cycles per phase of the forward kernel, per wave.
#ifdef SIREN_PROFILE
__device__ long long* g_siren_profile = nullptr;
#define SP_BEGIN long long sp_t = clock64(); long long sp_acc[8] = {};
#define SP_MARK(i) { const long long sp_n = clock64(); sp_acc[i] += sp_n - sp_t; sp_t = sp_n; }
#define SP_END                                                                         \
  if (g_siren_profile && (threadIdx.x & 63) == 0) {                                    \
    long long* dst = g_siren_profile + ((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; \
    for (int q = 0; q < 8; ++q) dst[q] = sp_acc[q];                                    \
  }
#else
#define SP_BEGIN
#define SP_MARK(i)
#define SP_END
#endif

// MODE 0: inference (nothing but y is written); 1: training (every layer's activation and
// derivative leave for HBM); 2: training with the loss: MSE against `target`, and the head's
// backward (dz of the last sine layer, dW_head, db_head, that layer's bias gradient) in the tile's
// tail, where a and w0 cos of the last sine layer are still in registers -- they never reach HBM,
// and the backward kernel starts from dz instead of reloading both.  Needs >= 2 sine layers.
template <int MODE, class S>
__global__ __launch_bounds__(kThreads) void siren_forward_kernel(const ChainArgs a) {
  __shared__ FwdSmem<S> sm;
  constexpr int H = S::H, NT = S::NT;
  constexpr bool STORE = MODE >= 1, LOSS = MODE == 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int rb = wave / S::CB, cb = wave % S::CB;
  const int n_mm = a.n_sine - 1;  // H x H layers

  // ---- small resident parameters ------------------------------------------------------------
  for (int l = 0; l < a.n_sine; ++l)
    for (int e = tid; e < H; e += kThreads) sm.bias[l][e] = a.b[l][e];
  for (int e = tid; e < H; e += kThreads) sm.w_last[e] = a.w[a.n_sine][e];
  const float b_last = a.b[a.n_sine][0];

  const int64_t tiles = (a.n + S::rows - 1) / S::rows;
  // this lane's fragment addresses
  const float* a_row = sm.img + (rb * 32 + l31) * S::ld + 4 * lh;
  const int n0 = cb * S::CT + l31;  // column of tile 0; tile t: + 32 t
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) boff[t] = 32 * (n0 + 32 * t) + 16 * (lh ^ (((n0 + 32 * t) >> 3) & 1));

  // chunk stream: chunk s (layer 1 + (s / chunks) % n_mm, columns 32 (s % chunks)) lives in wbuf[s & 1]
  int s = 0;
  if (n_mm > 0 && (int64_t)blockIdx.x < tiles) issue_chunk<S>(a.wsplit, 0, sm.wbuf[0], wave, lane);

  // activations waiting to leave for HBM (STORE): the outputs of MFMA layer `pend_l` of the tile
  // at `pend_m0`, dripped out a few registers per chunk beside the next layer's MFMAs
  float pa[NT][16], pd[NT][16];
  int pend_l = -1;
  int64_t pend_m0 = 0;
  float g_wh[NT], g_bl[NT], g_bhead = 0.f, g_loss = 0.f;  // loss mode: running sums of the workgroup
#pragma unroll
  for (int t = 0; t < NT; ++t) g_wh[t] = g_bl[t] = 0.f;
  // lane's element offset inside a tile's (rows, H) block for accumulator register 0 of tile 0
  const int lane_off = (rb * 32 + 4 * lh) * H + n0;
  auto drip = [&](int r_lo, int r_hi, bool tile_full) {
    if (!STORE || pend_l < 0) return;
    // uniform bases + a 32-bit lane offset; the offset is re-derived from an opaque copy per call,
    // or hipcc hoists all the store offsets out of the layer loop and spills
    float* __restrict__ ga = a.act[pend_l] + pend_m0 * H;
    float* __restrict__ gd = a.deriv[pend_l] + pend_m0 * H;
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const int64_t rows_left = a.n - pend_m0 - rb * 32 - 4 * lh;  // rows below this one are live
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < r_lo || r >= r_hi) continue;
        const int dr = (r & 3) + 8 * (r >> 2);
        if (tile_full || dr < rows_left) {  // tile_full is wave-uniform: no per-store branches
          ga[off + dr * H + t * 32] = pa[t][r];
          gd[off + dr * H + t * 32] = pd[t][r];
        }
      }
  };

  SP_BEGIN
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * S::rows;
    const bool full_tile = m0 + S::rows <= a.n;  // wave-uniform: stores need no per-lane row check
    // ---- x tile -> LDS -------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // previous tile is done with img / xs
    for (int e = tid; e < S::rows * kMaxIn; e += kThreads) {
      const int row = e / kMaxIn, d = e % kMaxIn;
      sm.xs[e] = (d < a.dim_in && m0 + row < a.n) ? a.x[(m0 + row) * a.dim_in + d] : 0.f;
    }
    if (LOSS && tid < S::rows) sm.tgt[tid] = m0 + tid < a.n ? a.target[m0 + tid] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SP_MARK(0)  // tile top: x tile
    // ---- first layer on the VALU: thread <-> (column, group of rows) ---------------------------
    {
      const int col = tid % H, r0 = (tid / H) * S::rpt;
      float wr[kMaxIn];
#pragma unroll
      for (int d = 0; d < kMaxIn; ++d) wr[d] = d < a.dim_in ? a.w[0][col * a.dim_in + d] : 0.f;
      const float bias = sm.bias[0][col];
      float* __restrict__ ga = STORE ? a.act[0] : nullptr;
      float* __restrict__ gd = STORE ? a.deriv[0] : nullptr;
#pragma unroll 8  // 8 independent sincos chains in flight
      for (int r = 0; r < S::rpt; r += 2) {
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int d = 0; d < kMaxIn; ++d) {
          if (d < a.dim_in) {
            z0 += sm.xs[(r0 + r) * kMaxIn + d] * wr[d];
            z1 += sm.xs[(r0 + r + 1) * kMaxIn + d] * wr[d];
          }
        }
        float s0, c0, s1, c1;
        sincos_fast2(a.w0_first * (z0 + bias), a.w0_first * (z1 + bias), &s0, &c0, &s1, &c1);
        sm.img[(r0 + r) * S::ld + col] = s0;
        sm.img[(r0 + r + 1) * S::ld + col] = s1;
        if (STORE) {
          const int64_t row = m0 + r0 + r;
          if (row < a.n) ga[row * H + col] = s0, gd[row * H + col] = a.w0_first * c0;
          if (row + 1 < a.n) ga[(row + 1) * H + col] = s1, gd[(row + 1) * H + col] = a.w0_first * c1;
        }
      }
    }
    SP_MARK(1)  // first layer
    // ---- H x H layers ---------------------------------------------------------------------------
    for (int l = 1; l <= n_mm; ++l) {
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      x3::Frag fa;
#pragma unroll
      for (int kc = 0; kc < S::chunks; ++kc, ++s) {
        // Chunk s has landed (this wave's own pieces; vmcnt counts loads, stores and LDS-DMA
        // together).  The stores dripped at the start of the previous chunk have had a whole
        // chunk to retire, so waiting for everything costs nothing: counted waits that left them
        // in flight measured 6.27 against 6.29 ms per 2^20-row pass.
#ifndef SIREN_EXPERIMENT_NO_DMA_WAIT  // timing experiment only (results are then wrong)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... for every wave, and wbuf[(s + 1) & 1] is free
        SP_MARK(2)  // chunk wait + barrier
        {
          const bool more_k = kc + 1 < S::chunks;
          const int nl = more_k ? l : (l < n_mm ? l + 1 : 1);
          if (more_k || l < n_mm || tile + gridDim.x < tiles)
            issue_chunk<S>(a.wsplit + (nl - 1) * split_matrix_bytes(H), more_k ? kc + 1 : 0,
                           sm.wbuf[(s + 1) & 1], wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        drip(kc * S::drip, (kc + 1) * S::drip, full_tile);
        SP_MARK(3)  // DMA issue + dripped stores
        if (kc == 0) fa = first_fragment(a_row);  // (the barrier above completed the image)
        mma_chunk<NT, H>(acc, fa, kc + 1 < S::chunks ? a_row + (kc + 1) * kKc : nullptr, sm.wbuf[s & 1], boff);
        SP_MARK(4)  // fragment reads + MFMAs
      }
      pend_l = -1;  // fully dripped
      // ---- epilogue: bias, w0, sincos; the image becomes this layer's output -------------------
      const float w0 = a.w0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bj = sm.bias[l][n0 + 32 * t];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float s0, c0, s1, c1;
          sincos_fast2(w0 * (acc[t][r] + bj), w0 * (acc[t][r + 1] + bj), &s0, &c0, &s1, &c1);
          pa[t][r] = s0, pa[t][r + 1] = s1;
          pd[t][r] = w0 * c0, pd[t][r + 1] = w0 * c1;
        }
      }
      SP_MARK(5)  // epilogue arithmetic
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave has read the image for the last time
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sm.img[(rb * 32 + acc_row(r, lh)) * S::ld + n0 + t * 32] = pa[t][r];
      if (STORE) {
        pend_l = l, pend_m0 = m0;
        if (l == n_mm) {  // no next MFMA layer in this tile: leave now (loss mode: stay on chip)
          if (!LOSS) drip(0, 16, full_tile);
          pend_l = -1;
        }
      }
      SP_MARK(6)  // epilogue barrier + image write
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // image complete
    // ---- head: y[row] = img[row] . w_last + b_last, one wave per row ---------------------------
    {
      constexpr int kPer = H >= 64 ? H / 64 : 1;  // elements per lane (H = 32: half the lanes)
      float wv[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) wv[j] = lane + 64 * j < H ? sm.w_last[lane + 64 * j] : 0.f;
#pragma unroll 4
      for (int i = 0; i < S::rows / 8; ++i) {
        const int row = wave * (S::rows / 8) + i;
        float acc1 = 0.f;
#pragma unroll
        for (int j = 0; j < kPer; ++j)
          if (lane + 64 * j < H) acc1 += sm.img[row * S::ld + lane + 64 * j] * wv[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc1 += __shfl_down(acc1, off, 64);
        if (lane == 0 && m0 + row < a.n) {
          const float yv = acc1 + b_last;
          a.y[m0 + row] = yv;
          if (LOSS) {  // models.py:64 F.mse_loss: mean((y - target)^2); dLoss/dy = 2 (y - t) / N
            const float diff = yv - sm.tgt[row];
            g_loss += diff * diff;
            const float dyv = diff * a.grad_scale;
            g_bhead += dyv;
            sm.tgt[row] = dyv;
          }
        } else if (LOSS && lane == 0) {
          sm.tgt[row] = 0.f;
        }
      }
    }
    if (LOSS) {
      // ---- head backward: dz = dy w_head (.) w0 cos, dW_head += dy^T a, db_last += colsum(dz) -----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every row's dLoss/dy is in LDS
      float* __restrict__ gz = a.dz_last + m0 * H;
      int off = lane_off;
      asm volatile("" : "+v"(off));
      const int64_t rows_left = a.n - m0 - rb * 32 - 4 * lh;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float wl = sm.w_last[n0 + 32 * t];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const float dyv = sm.tgt[rb * 32 + 4 * lh + dr];
          g_wh[t] += dyv * pa[t][r];
          const float dz = (dyv * wl) * pd[t][r];
          g_bl[t] += dz;
          if (full_tile || dr < rows_left) gz[off + dr * H + t * 32] = dz;
        }
      }
    }
    SP_MARK(7)  // head
  }
  SP_END
  if (LOSS) {
    // ---- this workgroup's slab: dW_head, db of the last sine layer, db_head, loss -----------------
    float* slab = a.partial + (int64_t)blockIdx.x * fwd_slab_floats(H);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = sm.img;  // scratch: [row block][H] per quantity
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float wh = g_wh[t] + __shfl_xor(g_wh[t], 32, 64);  // the lane halves hold different rows
      const float bl = g_bl[t] + __shfl_xor(g_bl[t], 32, 64);
      if (lh == 0) {
        red[rb * H + n0 + 32 * t] = wh;
        red[(S::RB + rb) * H + n0 + 32 * t] = bl;
      }
    }
    __syncthreads();
    if (tid < H) {
      float wh = 0.f, bl = 0.f;
#pragma unroll
      for (int q = 0; q < S::RB; ++q) wh += red[q * H + tid], bl += red[(S::RB + q) * H + tid];
      slab[tid] = wh;
      slab[H + tid] = bl;
    }
    __syncthreads();
    red[tid] = lane == 0 ? g_bhead : 0.f;  // lane 0 of each wave holds its rows' sums
    red[kThreads + tid] = lane == 0 ? g_loss : 0.f;
    __syncthreads();
    if (tid == 0) {
      float sb = 0.f, sl = 0.f;
      for (int w = 0; w < 8; ++w) sb += red[w * 64], sl += red[kThreads + w * 64];
      slab[2 * H] = sb, slab[2 * H + 1] = sl * a.inv_n, slab[2 * H + 2] = 0.f, slab[2 * H + 3] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward chain: dz of a row tile walks the layers from the head down to the first layer
// inside LDS.  Per tile:
//   head     dz_L = dy w_head^T (.) d_L  on the VALU into the image; dW_head, db_head partials
//   layer l  da_{l-1} = dz_l W_l on the MFMAs: the image (dz_l) is the A operand, W_l streams
//            from L2 UNTRANSPOSED in 32-row chunks (row n of W_l = contraction index n, its H
//            columns = output columns: B fragments are conflict-free ds_read_b32, no swizzle);
//            dz_{l-1} = da_{l-1} (.) d_{l-1} in the epilogue (d prefetched at the layer's first
//            chunk), bias-gradient column sums in LDS, the image becomes dz_{l-1}
//   first    dW_first = dz_0^T x on the VALU (K = dim_in)
// dz_l (l >= 1) leaves for HBM once: siren_wgrad_kernel contracts it with a_{l-1} over the batch.
// Partial sums (biases, head, first layer) leave through one slab per workgroup, summed in a
// fixed order by siren_bwd_reduce_kernel (bitwise reproducible, no float atomics).
template <class S>
struct BwdSmem {
  char wbuf[2][S::chunk_bytes] __attribute__((aligned(16)));  // chunks of the split W^T: term planes [k][2 slots]
  float img[S::rows * S::ld];
  float xs[S::rows * kMaxIn];
  float w_last[S::H];
  float dy[S::rows];
  float gb[kMaxSine][S::RB][S::H];  // bias-gradient column sums per layer and row block (sole owners)
};

template <class S>
__global__ __launch_bounds__(kThreads) void siren_backward_kernel(const BwdArgs a) {
  __shared__ BwdSmem<S> sm;
  constexpr int H = S::H, NT = S::NT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int rb = wave / S::CB, cb = wave % S::CB;
  const int L = a.n_sine;  // sine layers 0 .. L-1, head = layer L
  for (int e = tid; e < H; e += kThreads) sm.w_last[e] = a.w[L][e];

  const int64_t tiles = (a.n + S::rows - 1) / S::rows;
  const float* a_row = sm.img + (rb * 32 + l31) * S::ld + 4 * lh;
  const int n0 = cb * S::CT + l31;
  const int lane_off = (rb * 32 + 4 * lh) * H + n0;  // element (row of register 0, column n0)
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) boff[t] = 32 * (n0 + 32 * t) + 16 * (lh ^ (((n0 + 32 * t) >> 3) & 1));
  const int col = tid % H, r0 = (tid / H) * S::rpt;  // VALU phases: (column, group of rows)

  // running sums of this workgroup
  float g_whead = 0.f, g_bhead = 0.f, g_blast = 0.f;  // (col, group) mapping; g_bhead: tid < rows
  float g_wfirst[kMaxIn];                             // (col, group) mapping
  for (int e = tid; e < kMaxSine * S::RB * H; e += kThreads) (&sm.gb[0][0][0])[e] = 0.f;
#pragma unroll
  for (int d = 0; d < kMaxIn; ++d) g_wfirst[d] = 0.f;

  int s = 0;
  if (L > 1 && (int64_t)blockIdx.x < tiles)
    issue_chunk<S>(a.wtsplit + (L - 2) * split_matrix_bytes(H), 0, sm.wbuf[0], wave, lane);

  float pz[NT][16];  // dz waiting to leave for HBM, dripped beside the next layer's MFMAs
  int pend_l = -1;
  int64_t pend_m0 = 0;
  auto drip = [&](int r_lo, int r_hi, bool tile_full) {
    if (pend_l < 0) return;
    float* __restrict__ gz = a.dz[pend_l] + pend_m0 * H;
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const int64_t rows_left = a.n - pend_m0 - rb * 32 - 4 * lh;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < r_lo || r >= r_hi) continue;
        const int dr = (r & 3) + 8 * (r >> 2);
        if (tile_full || dr < rows_left) gz[off + dr * H + t * 32] = pz[t][r];
      }
  };
  // derivative of a layer's activation in the accumulator layout (zeros beyond n)
  auto load_deriv = [&](const float* __restrict__ d, int64_t m0, bool tile_full, float (&dv)[NT][16]) {
    const float* __restrict__ g = d + m0 * H;
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const int64_t rows_left = a.n - m0 - rb * 32 - 4 * lh;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        dv[t][r] = (tile_full || dr < rows_left) ? g[off + dr * H + t * 32] : 0.f;
      }
  };

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * S::rows;
    const bool full_tile = m0 + S::rows <= a.n;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // previous tile is done with img / xs / dy
    for (int e = tid; e < S::rows * kMaxIn; e += kThreads) {
      const int row = e / kMaxIn, d = e % kMaxIn;
      sm.xs[e] = (d < a.dim_in && m0 + row < a.n) ? a.x[(m0 + row) * a.dim_in + d] : 0.f;
    }
    if (!a.head_done && tid < S::rows) {
      const float v = m0 + tid < a.n ? a.dy[m0 + tid] : 0.f;
      sm.dy[tid] = v;
      g_bhead += v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (a.head_done) {
      // ---- the forward kernel's tail did the head: the image starts as dz_{L-1} ------------------
      const float* __restrict__ gz = a.dz[L - 1] + m0 * H + col;
      float zv[S::rpt];  // every load in flight before the first LDS store
#pragma unroll
      for (int i = 0; i < S::rpt; ++i)
        zv[i] = (full_tile || m0 + r0 + i < a.n) ? gz[(r0 + i) * H] : 0.f;
#pragma unroll
      for (int i = 0; i < S::rpt; ++i) sm.img[(r0 + i) * S::ld + col] = zv[i];
    } else
    // ---- head: dz_{L-1} = dy w_head (.) d_{L-1}; dW_head += dy^T a_{L-1} ------------------------
    {
      const float wl = sm.w_last[col];
      const float* __restrict__ ga = a.act_last + m0 * H + col;
      const float* __restrict__ gd = a.deriv[L - 1] + m0 * H + col;
      float* __restrict__ gz = L > 1 ? a.dz[L - 1] + m0 * H + col : nullptr;
#pragma unroll 2
      for (int rr = 0; rr < S::rpt; rr += 8) {
        float av[8], dv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = r0 + rr + i;
          const bool live = full_tile || m0 + row < a.n;
          av[i] = live ? ga[row * H] : 0.f;
          dv[i] = live ? gd[row * H] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = r0 + rr + i;
          const float dyv = sm.dy[row];
          g_whead += dyv * av[i];
          const float dz = (dyv * wl) * dv[i];
          g_blast += dz;
          sm.img[row * S::ld + col] = dz;
          if (gz && (full_tile || m0 + row < a.n)) gz[row * H] = dz;
        }
      }
    }
    // ---- layers L-1 .. 1: da_{l-1} = dz_l W_l, dz_{l-1} = da_{l-1} (.) d_{l-1} ------------------
    for (int l = L - 1; l >= 1; --l) {
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      float dv[NT][16];
      x3::Frag fa;
#pragma unroll
      for (int kc = 0; kc < S::chunks; ++kc, ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // chunk s landed for every wave; the other buffer is free
        {
          const bool more_k = kc + 1 < S::chunks;
          const int nl = more_k ? l : (l > 1 ? l - 1 : L - 1);
          if (more_k || l > 1 || tile + gridDim.x < tiles)
            issue_chunk<S>(a.wtsplit + (nl - 1) * split_matrix_bytes(H), more_k ? kc + 1 : 0,
                           sm.wbuf[(s + 1) & 1], wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        drip(kc * S::drip, (kc + 1) * S::drip, full_tile);
        if (kc == 0) load_deriv(a.deriv[l - 1], m0, full_tile, dv);  // lands beside the MFMAs
        if (kc == 0) fa = first_fragment(a_row);  // (the barrier above completed the image)
        mma_chunk<NT, H>(acc, fa, kc + 1 < S::chunks ? a_row + (kc + 1) * kKc : nullptr, sm.wbuf[s & 1], boff);
      }
      pend_l = -1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pz[t][r] = acc[t][r] * dv[t][r];
          sum += pz[t][r];
        }
        sum += __shfl_xor(sum, 32, 64);  // the two lane halves hold different rows of one column
        // (layer, row block, column) has exactly one owner lane: plain read-add-write
        if (lh == 0) sm.gb[l - 1][rb][n0 + 32 * t] += sum;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave has read the image for the last time
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sm.img[(rb * 32 + acc_row(r, lh)) * S::ld + n0 + t * 32] = pz[t][r];
      if (l - 1 >= 1) {  // dz_{l-1} feeds the weight gradient of layer l-1
        pend_l = l - 1, pend_m0 = m0;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // image = dz_0
    // ---- first layer: dW_first[col][d] += sum_rows dz_0[row][col] x[row][d] --------------------
#pragma unroll 4
    for (int r = 0; r < S::rpt; ++r) {
      const float dz = sm.img[(r0 + r) * S::ld + col];
#pragma unroll
      for (int d = 0; d < kMaxIn; ++d)
        if (d < a.dim_in) g_wfirst[d] += dz * sm.xs[(r0 + r) * kMaxIn + d];
    }
  }

  // ---- this workgroup's slab -------------------------------------------------------------------
  float* slab = a.partial + (int64_t)blockIdx.x * bwd_slab_floats(H, L);
  float* p_whead = slab;
  float* p_bhead = slab + H;
  float* p_b = slab + H + 4;
  float* p_wfirst = p_b + L * H;
  __syncthreads();
  float* red = sm.img;  // scratch; (col, group) sums: `groups` partial sums per column
  auto column_sum = [&](const float* v) {
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < S::groups; ++g) sum += v[g * H + tid];
    return sum;
  };
  red[tid] = g_whead;
  red[kThreads + tid] = g_blast;
  __syncthreads();
  if (tid < H) {
    p_whead[tid] = column_sum(red);
    p_b[(L - 1) * H + tid] = column_sum(red + kThreads);
    for (int l = 0; l + 1 < L; ++l) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < S::RB; ++q) sum += sm.gb[l][q][tid];
      p_b[l * H + tid] = sum;
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < kMaxIn; ++d) {
    red[tid] = g_wfirst[d];
    __syncthreads();
    if (tid < H) p_wfirst[tid * kMaxIn + d] = column_sum(red);
    __syncthreads();
  }
  red[tid] = tid < S::rows ? g_bhead : 0.f;
  __syncthreads();
  if (tid == 0) {
    float sb = 0.f;
    for (int c = 0; c < S::rows; ++c) sb += red[c];
    p_bhead[0] = sb, p_bhead[1] = 0.f, p_bhead[2] = 0.f, p_bhead[3] = 0.f;
  }
}

// Sum the backward slabs in a fixed order and add them to the gradient tensors.
struct BwdReduceArgs {
  const float* partial;
  int slabs, hidden, n_sine, dim_in;
  float* d_w[kMaxSine + 1];
  float* d_b[kMaxSine + 1];
};

__global__ __launch_bounds__(256) void siren_bwd_reduce_kernel(const BwdReduceArgs r) {
  const int H = r.hidden, L = r.n_sine;
  const int slab = bwd_slab_floats(H, L);
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= slab) return;
  float sum = 0.f;
  const float* p = r.partial + e;
  int b = 0;
  for (; b + 8 <= r.slabs; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(b + j) * slab];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  for (; b < r.slabs; ++b) sum += p[(int64_t)b * slab];
  if (e < H) {
    r.d_w[L][e] += sum;                                      // head weight (1, H)
  } else if (e < H + 4) {
    if (e == H) r.d_b[L][0] += sum;                          // head bias
  } else if (e < H + 4 + L * H) {
    const int q = e - H - 4;
    r.d_b[q / H][q % H] += sum;                              // sine-layer biases
  } else {
    const int q = e - H - 4 - L * H, o = q / kMaxIn, d = q % kMaxIn;
    if (d < r.dim_in) r.d_w[0][o * r.dim_in + d] += sum;     // first layer (H, dim_in)
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of an H x H layer: dW[n][k] = sum_rows dz[row][n] a[row][k].
// One persistent workgroup per CU holds the WHOLE H x H result in MFMA accumulators (H = 256:
// 8 waves x 2 x 4 tiles of 32 x 32 = 128 registers) and streams its share of the batch through
// LDS in 32-row chunks of dz and a (LDS-DMA, double buffered, rows as they lie in HBM: both
// operands are read along their contiguous axis, conflict-free ds_read_b32, no padding, no
// swizzle).  A chunk feeds 128 MFMAs per wave at H = 256, so the chunk barrier costs ~2 %.
// Narrow layers have fewer tiles than waves: the waves of one tile then split the chunk's row
// pairs among themselves and each writes its own slab.  The per-workgroup (per-wave-group)
// results meet in a slab workspace and are summed in a fixed order (slab_sum_kernel).

template <int HH>
struct WgradShape {
  static constexpr int H = HH, TD = H / 32, tiles = TD * TD;
  static constexpr bool wide = tiles >= 8;             // every wave owns TI x TJ tiles
  static constexpr int TI = wide ? TD / 4 : 1, TJ = wide ? TD / 2 : 1;
  static constexpr int RS = wide ? 1 : 8 / tiles;      // waves per tile, splitting the row pairs
  static constexpr int pieces = H / 8;                 // 1-KiB DMA pieces of a 32 x H chunk
};

template <class W>
struct WgradSmem {
  float z[2][kWr * W::H];
  float a[2][kWr * W::H];
};

template <class W>
__global__ __launch_bounds__(kThreads) void siren_wgrad_kernel(const WgradArgs g) {
  __shared__ WgradSmem<W> sm;
  constexpr int H = W::H, TI = W::TI, TJ = W::TJ, RS = W::RS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  // rows (n) / columns (k) of dW this wave owns, and its share of a chunk's row pairs
  const int tile = W::wide ? 0 : wave % W::tiles, split = W::wide ? 0 : wave / W::tiles;
  const int n_base = W::wide ? (wave >> 1) * TI * 32 : (tile / W::TD) * 32;
  const int k_base = W::wide ? (wave & 1) * TJ * 32 : (tile % W::TD) * 32;
  const int64_t chunks = (g.n + kWr - 1) / kWr;
  const int64_t per = (chunks + gridDim.x - 1) / gridDim.x;
  const int64_t c_lo = (int64_t)blockIdx.x * per, c_hi = c_lo + per < chunks ? c_lo + per : chunks;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  auto issue = [&](int64_t c, int buf) {  // 32 rows of dz and of a, contiguous in HBM
#pragma unroll
    for (int i = 0; i < (W::pieces + 7) / 8; ++i) {
      const int piece = wave + 8 * i;
      if (W::pieces % 8 != 0 && piece >= W::pieces) break;
      // a piece is 256 / H rows of the chunk (one row at H = 256: the row index is then wave-uniform)
      int64_t row = c * kWr + piece * (256 / H) + (lane * 4) / H;
      if (row >= g.n) row = g.n - 1;  // stays inside the buffers; such rows are zeroed below
      const int64_t src = row * H + (lane * 4) % H;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(g.dz + src),
          (__attribute__((address_space(3))) void*)(sm.z[buf] + piece * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(g.act + src),
          (__attribute__((address_space(3))) void*)(sm.a[buf] + piece * 256), 16, 0, 0);
    }
  };
  if (c_lo < c_hi) issue(c_lo, 0);
  for (int64_t c = c_lo; c < c_hi; ++c) {
    const int buf = (int)(c - c_lo) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // chunk c landed for every wave; the other buffer is free
    if (c + 1 < c_hi) issue(c + 1, buf ^ 1);
    if ((c + 1) * kWr > g.n) {  // the batch ends inside this chunk: rows beyond it contribute 0
      const int live = (int)(g.n - c * kWr);
      for (int e = tid; e < (kWr - live) * H; e += kThreads) sm.z[buf][live * H + e] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const float* zp = sm.z[buf] + lh * H + n_base + l31;
    const float* ap = sm.a[buf] + lh * H + k_base + l31;
    constexpr int kPairs = kWr / 2 / RS;  // row pairs of this wave: split, split + RS, ...
#if MRI_SIREN_X3
    if constexpr (kPairs % 8 == 0) {
      // a 16-deep step contracts eight of the wave's row pairs: lane half lh holds row 2 pair + lh,
      // element j of both operands = pair split + (8 s + j) RS
      // Both operands are split a tile ahead, beside the MFMAs of the tile before (one MFMA, then a share of the
      // reads / of the split's 44 vector instructions per fragment: a bf16 MFMA leaves the issue port free for 24 of
      // its 32 cycles).  Split in front of their own MFMAs, as until round 4, the matrix pipe idled meanwhile
      // (weight gradient 0.75 -> 0.64 ms per layer at config 3).
      float raw[8];
      auto fetch_a = [&](int s, int tj) {
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = ap[2 * (split + (8 * s + j) * RS) * H + tj * 32];
      };
      float rz[TI][8];
      auto fetch_z = [&](int s) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int j = 0; j < 8; ++j) rz[ti][j] = zp[2 * (split + (8 * s + j) * RS) * H + ti * 32];
      };
      x3::Frag fz[TI], fa, fz_keep0;
      fetch_z(0);
      fetch_a(0, 0);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) fz[ti] = x3::split8(rz[ti]);
      fa = x3::split8(raw);
      constexpr int kSteps = kPairs / 8, kM = 6 * TI;  // (kM MFMAs per tile: the first four carry reads, the others vector work)
      // where the NEXT step's dz fragments are read and split: spread over the last tiles of a step when it has
      // three or more (H = 256: read beside tile 1, split beside tiles 2 and 3), else all beside its last tile
      constexpr bool kSpread = TJ >= 3 && TI == 2;
#pragma unroll
      for (int s = 0; s < kSteps; ++s) {
        const bool next_step = s + 1 < kSteps;
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const bool last_tile = tj + 1 == TJ, more = !last_tile || next_step;
          __builtin_amdgcn_sched_barrier(0);
          if (!last_tile) fetch_a(s, tj + 1);
          if (last_tile && next_step) fetch_a(s + 1, 0);
          const bool z_read = next_step && (kSpread ? tj == TJ - 3 : last_tile);
          if (z_read) fetch_z(s + 1);
          x3::Frag fz_use[TI];
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) fz_use[ti] = fz[ti];
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) acc[ti][tj] = mma6_32(fz_use[ti], fa, acc[ti][tj]);
          if (more) fa = x3::split8(raw);
          x3::Frag fz_next[TI];
          int z_splits = 0;
          if (next_step) {
            if (kSpread) {
              if (tj == TJ - 2) fz_next[0] = x3::split8(rz[0]), z_splits = 1;
              if (tj == TJ - 1) fz_next[TI - 1] = x3::split8(rz[TI - 1]), z_splits = 1;
            } else if (last_tile) {
#pragma unroll
              for (int ti = 0; ti < TI; ++ti) fz_next[ti] = x3::split8(rz[ti]);
              z_splits = TI;
            }
          }
          // (fz of this step is still in use by the tiles behind this one: the new fragments take over at the step's end)
          if (more && !z_read && z_splits == 0) sched_interleave<kM, 2, (44 + kM - 5) / (kM - 4)>();
          if (more && z_read && z_splits == 0) sched_interleave<kM, 2 * (1 + TI), (44 + kM - 5) / (kM - 4)>();
          if (more && !z_read && z_splits == 1) sched_interleave<kM, 2, (88 + kM - 5) / (kM - 4)>();
          if (more && z_read && z_splits > 0) sched_interleave<kM, 2 * (1 + TI), (44 * (1 + TI) + kM - 5) / (kM - 4)>();
          if (kSpread) {
            if (next_step && tj == TJ - 2) fz_keep0 = fz_next[0];
            if (next_step && last_tile) fz[0] = fz_keep0, fz[TI - 1] = fz_next[TI - 1];
          } else if (next_step && last_tile) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) fz[ti] = fz_next[ti];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      continue;
    }
#endif
    float zv[2][TI], av[2][TJ];
    auto fetch = [&](int b, int rp) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) zv[b][ti] = zp[2 * rp * H + ti * 32];
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) av[b][tj] = ap[2 * rp * H + tj * 32];
    };
    auto compute = [&](int b) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[b][ti], av[b][tj], acc[ti][tj], 0, 0, 0);
    };
    fetch(0, split);
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      if (q + 1 < kPairs) fetch((q + 1) & 1, split + (q + 1) * RS);
      __builtin_amdgcn_sched_barrier(0);
      compute(q & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* slab = g.partial + ((int64_t)blockIdx.x * RS + split) * H * H;
  // (narrow layers: the `tiles` waves that share a split cover the whole H x H slab between them)
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[(n_base + ti * 32 + acc_row(r, lh)) * H + k_base + tj * 32 + l31] = acc[ti][tj][r];
}

// dst[e] += sum over slabs of partial[slab][e], fixed order
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ partial, int slabs,
                                                       int count, float* __restrict__ dst) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const float* p = partial + e;
  float sum = 0.f;
  int b = 0;
  for (; b + 8 <= slabs; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(b + j) * count];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  for (; b < slabs; ++b) sum += p[(int64_t)b * count];
  dst[e] += sum;
}

// Three-term split of the H x H weights into the chunk-major planes issue_chunk streams (layout:
// split_matrix_bytes).  transposed: the planes of W^T (row = input unit k, contraction = output unit).
struct SplitArgs {
  const float* w[kMaxSine];
  int count, H, transposed;
  char* out;
};

__global__ __launch_bounds__(256) void siren_split_weights_kernel(const SplitArgs g) {
  const int H = g.H, per = H * (H / kKc) * 2;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= g.count * per) return;
  const int m = id / per, e = id % per;
  const int q = e & 1, row = (e >> 1) % H, kc = (e >> 1) / H;
  const float* w = g.w[m];
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kKc * kc + 8 * (j >> 2) + 4 * q + (j & 3);
    v[j] = g.transposed ? w[k * H + row] : w[row * H + k];
  }
  const x3::Frag f = x3::split8(v);
  char* dst = g.out + m * split_matrix_bytes(H) + (int64_t)kc * 3 * H * 32 + row * 32 + 16 * q;
  *reinterpret_cast<x3::u32x4*>(dst) = f.h;
  *reinterpret_cast<x3::u32x4*>(dst + H * 32) = f.m;
  *reinterpret_cast<x3::u32x4*>(dst + 2 * H * 32) = f.l;
}

int split_weights(const float* const* weight, int n_sine, int hidden, bool transposed, char* out,
                  hipStream_t st) {
  if (n_sine < 2) return MRI_OK;
  SplitArgs g{};
  g.count = n_sine - 1, g.H = hidden, g.transposed = transposed ? 1 : 0, g.out = out;
  for (int l = 1; l < n_sine; ++l) g.w[l - 1] = weight[l];
  const int threads = g.count * hidden * (hidden / kKc) * 2;
  hipLaunchKernelGGL(siren_split_weights_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0,
                     st, g);
  return check_launch("siren_split_weights_kernel");
}

int64_t split_region_bytes(int hidden, int n_sine) {
  return n_sine > 1 ? (n_sine - 1) * split_matrix_bytes(hidden) : 0;
}

// workspace of the training entry points: [slabs of partial sums][split weights]
int64_t slab_region_bytes(int64_t n, int hidden, int n_sine);

// ------------------------------------------------------------------------------ host side
bool chain_supported(int dim_in, int hidden, int n_sine, int dim_out) {
  return (hidden == 32 || hidden == 64 || hidden == 128 || hidden == 256) && dim_in >= 1 &&
         dim_in <= kMaxIn && n_sine >= 1 && n_sine <= kMaxSine && dim_out == 1;
}

int tile_rows(int hidden) { return hidden >= 256 ? 64 : hidden == 128 ? 128 : 256; }
int wgrad_split(int hidden) { return hidden >= 128 ? 1 : hidden == 64 ? 2 : 8; }
int chain_blocks(int hidden, int64_t n) { return (int)std::min<int64_t>(ceil_div(n, tile_rows(hidden)), 256); }
int wgrad_blocks(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, kWr), 256); }

template <int H>
int launch_forward(const ChainArgs& a, int mode, hipStream_t st) {
  using S = Shape<H>;
  const int blocks = chain_blocks(H, a.n);  // one workgroup per CU
  if (mode == 2)
    hipLaunchKernelGGL((siren_forward_kernel<2, S>), dim3(blocks), dim3(kThreads), 0, st, a);
  else if (mode == 1)
    hipLaunchKernelGGL((siren_forward_kernel<1, S>), dim3(blocks), dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL((siren_forward_kernel<0, S>), dim3(blocks), dim3(kThreads), 0, st, a);
  return check_launch("siren_forward_kernel");
}

template <int H>
int launch_backward(const BwdArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((siren_backward_kernel<Shape<H>>), dim3(chain_blocks(H, a.n)), dim3(kThreads),
                     0, st, a);
  return check_launch("siren_backward_kernel");
}

template <int H>
int launch_wgrad(const WgradArgs& g, float* d_weight, hipStream_t st) {
  using W = WgradShape<H>;
  const int wb = wgrad_blocks(g.n);
  hipLaunchKernelGGL((siren_wgrad_kernel<W>), dim3(wb), dim3(kThreads), 0, st, g);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)ceil_div(H * H, 256)), dim3(256), 0, st,
                     g.partial, wb * W::RS, H * H, d_weight);
  return check_launch("siren_wgrad_kernel");
}

int forward_any(int hidden, const ChainArgs& a, int mode, hipStream_t st) {
  // (the loss mode hands over to the rows backward kernel: both or neither)
  if (options().siren_rows && rows_supported(hidden, a.n_sine) &&
      (mode != 2 || rows_backward_supported(hidden, a.n_sine, a.dim_in, 1)))
    return forward_rows(a, mode, st);
  switch (hidden) {
    case 32: return launch_forward<32>(a, mode, st);
    case 64: return launch_forward<64>(a, mode, st);
    case 128: return launch_forward<128>(a, mode, st);
    default: return launch_forward<256>(a, mode, st);
  }
}

// loss-mode slabs -> head weight / bias gradient, last sine layer's bias gradient, loss
struct FwdReduceArgs {
  const float* partial;
  int slabs, hidden;
  float* d_w_head;
  float* d_b_head;
  float* d_b_last;
  float* loss_out;
};

__global__ __launch_bounds__(256) void siren_fwd_reduce_kernel(const FwdReduceArgs r) {
  const int H = r.hidden, slab = fwd_slab_floats(H);
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= slab) return;
  // (eight loads in flight, the additions in slab order: a load per addition was 63 us of dependent round trips)
  float sum = 0.f;
  const float* p = r.partial + e;
  int b = 0;
  for (; b + 8 <= r.slabs; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(b + j) * slab];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  for (; b < r.slabs; ++b) sum += p[(int64_t)b * slab];
  if (e < H)
    r.d_w_head[e] += sum;
  else if (e < 2 * H)
    r.d_b_last[e - H] += sum;
  else if (e == 2 * H)
    r.d_b_head[0] += sum;
  else if (e == 2 * H + 1)
    r.loss_out[0] += sum;
}

int backward_any(int hidden, const BwdArgs& a, hipStream_t st) {
  switch (hidden) {
    case 32: return launch_backward<32>(a, st);
    case 64: return launch_backward<64>(a, st);
    case 128: return launch_backward<128>(a, st);
    default: return launch_backward<256>(a, st);
  }
}

int wgrad_any(int hidden, const WgradArgs& g, float* d_weight, hipStream_t st) {
  switch (hidden) {
    case 32: return launch_wgrad<32>(g, d_weight, st);
    case 64: return launch_wgrad<64>(g, d_weight, st);
    case 128: return launch_wgrad<128>(g, d_weight, st);
    default: return launch_wgrad<256>(g, d_weight, st);
  }
}

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int mri_siren_supported(int32_t dim_in, int32_t hidden, int32_t n_sine_layers,
                                   int32_t dim_out) {
  return chain_supported(dim_in, hidden, n_sine_layers, dim_out) ? 1 : 0;
}

extern "C" int mri_siren_forward(const float* x, int64_t n, int32_t dim_in, int32_t hidden,
                                 int32_t n_sine_layers, const float* const* weight,
                                 const float* const* bias, float w0_first, float w0,
                                 float* const* act, float* const* deriv, float* y, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(chain_supported(dim_in, hidden, n_sine_layers, 1),
              "fused SIREN chain: %d -> %d x %d -> 1 is not supported (hidden 32 / 64 / 128 / 256, "
              "dim_in <= 8, <= %d sine layers)", dim_in, hidden, n_sine_layers, kMaxSine);
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && weight && bias && y, "NULL pointer");
  const int64_t need = split_region_bytes(hidden, n_sine_layers);
  MRI_REQUIRE(need == 0 || (workspace && workspace_bytes >= need &&
                            (reinterpret_cast<uintptr_t>(workspace) & 15) == 0),
              "SIREN forward needs a 16-byte aligned workspace of %lld bytes "
              "(mri_siren_forward_workspace_bytes)", (long long)need);
  MRI_REQUIRE((act == nullptr) == (deriv == nullptr), "act and deriv go together");
  ChainArgs a{};
  a.x = x, a.n = n, a.dim_in = dim_in, a.n_sine = n_sine_layers;
  a.w0_first = w0_first, a.w0 = w0, a.y = y;
  for (int l = 0; l <= n_sine_layers; ++l) {
    MRI_REQUIRE(weight[l] && bias[l], "NULL parameter pointer (layer %d)", l);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(weight[l]) & 15) == 0,
                "weights must be 16-byte aligned (layer %d)", l);
    a.w[l] = weight[l], a.b[l] = bias[l];
  }
  if (act)
    for (int l = 0; l < n_sine_layers; ++l) {
      MRI_REQUIRE(act[l] && deriv[l], "NULL activation buffer (layer %d)", l);
      a.act[l] = act[l], a.deriv[l] = deriv[l];
    }
  a.wsplit = static_cast<const char*>(workspace);
  if (int rc = split_weights(weight, n_sine_layers, hidden, false, static_cast<char*>(workspace),
                             (hipStream_t)stream))
    return rc;
  return forward_any(hidden, a, act != nullptr ? 1 : 0, (hipStream_t)stream);
}

extern "C" int64_t mri_siren_forward_workspace_bytes(int32_t hidden, int32_t n_sine_layers) {
  if (!chain_supported(1, hidden, n_sine_layers, 1)) return -1;
  return split_region_bytes(hidden, n_sine_layers);
}

namespace mri {
namespace {
// What the last mri_siren_forward_loss of this host thread left behind: mri_siren_backward(head_done = 1) continues exactly
// that call (the two kernel families park different things in dz_last and the workspace), so it checks instead of trusting.
struct LossCall {
  const void* workspace = nullptr;
  const void* dz_last = nullptr;
  int64_t n = -1;
  int hidden = 0, n_sine = 0, rows = -1;
};
// (the last eight calls, by dz_last buffer: several networks may interleave their forward / backward pairs)
thread_local LossCall g_loss_calls[8];
thread_local int g_loss_call_next = 0;
void remember_loss_call(const LossCall& c) {
  for (LossCall& o : g_loss_calls)
    if (o.dz_last == c.dz_last) {
      o = c;
      return;
    }
  g_loss_calls[g_loss_call_next] = c;
  g_loss_call_next = (g_loss_call_next + 1) % 8;
}
const LossCall* find_loss_call(const void* dz_last) {
  for (const LossCall& o : g_loss_calls)
    if (o.dz_last == dz_last && o.n >= 0) return &o;
  return nullptr;
}
}  // namespace
}  // namespace mri

extern "C" int mri_siren_forward_loss(const float* x, const float* target, int64_t n,
                                      int64_t n_total, int32_t dim_in, int32_t hidden,
                                      int32_t n_sine_layers, const float* const* weight,
                                      const float* const* bias, float w0_first, float w0,
                                      float grad_divisor, float* const* act, float* const* deriv,
                                      float* dz_last, float* y, float* d_w_head, float* d_b_head,
                                      float* d_b_last, float* loss_out, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(chain_supported(dim_in, hidden, n_sine_layers, 1) && n_sine_layers >= 2,
              "fused SIREN chain with loss: %d -> %d x %d -> 1 is not supported (>= 2 sine layers)",
              dim_in, hidden, n_sine_layers);
  MRI_REQUIRE(n >= 0 && n < (1ll << 31) && n_total >= n && grad_divisor > 0.f, "bad n / divisor");
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && target && weight && bias && act && deriv && dz_last && y && d_w_head &&
                  d_b_head && d_b_last && loss_out, "NULL pointer");
  const int blocks = chain_blocks(hidden, n);
  const int64_t need = mri_siren_backward_workspace_bytes(n, hidden, n_sine_layers);
  MRI_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "SIREN forward with loss needs a 16-byte aligned workspace of %lld bytes "
              "(mri_siren_backward_workspace_bytes)", (long long)need);
  char* const wsplit = static_cast<char*>(workspace) + slab_region_bytes(n, hidden, n_sine_layers);
  ChainArgs a{};
  a.x = x, a.n = n, a.dim_in = dim_in, a.n_sine = n_sine_layers;
  a.w0_first = w0_first, a.w0 = w0, a.y = y;
  a.target = target, a.dz_last = dz_last, a.partial = static_cast<float*>(workspace);
  a.dy_ws = reinterpret_cast<float*>(static_cast<char*>(workspace) + rows_dy_offset(n, hidden, n_sine_layers));
  a.grad_scale = (float)(2.0 / ((double)n_total * (double)grad_divisor));
  a.inv_n = (float)(1.0 / (double)n_total);
  for (int l = 0; l <= n_sine_layers; ++l) {
    MRI_REQUIRE(weight[l] && bias[l], "NULL parameter pointer (layer %d)", l);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(weight[l]) & 15) == 0,
                "weights must be 16-byte aligned (layer %d)", l);
    a.w[l] = weight[l], a.b[l] = bias[l];
  }
  for (int l = 0; l + 1 < n_sine_layers; ++l) {  // the last sine layer's a / w0 cos stay on chip
    MRI_REQUIRE(act[l] && deriv[l], "NULL activation buffer (layer %d)", l);
    a.act[l] = act[l], a.deriv[l] = deriv[l];
  }
  hipStream_t st = (hipStream_t)stream;
  a.wsplit = wsplit;
  remember_loss_call(LossCall{workspace, dz_last, n, hidden, n_sine_layers,
                              options().siren_rows && rows_backward_supported(hidden, n_sine_layers, dim_in, 1) ? 1 : 0});
  if (int rc = split_weights(weight, n_sine_layers, hidden, false, wsplit, st)) return rc;
  if (int rc = forward_any(hidden, a, 2, st)) return rc;
  FwdReduceArgs r{};
  r.partial = a.partial, r.hidden = hidden;
  r.slabs = options().siren_rows && rows_backward_supported(hidden, n_sine_layers, dim_in, 1) ? rows_blocks(n) : blocks;
  r.d_w_head = d_w_head, r.d_b_head = d_b_head, r.d_b_last = d_b_last, r.loss_out = loss_out;
  hipLaunchKernelGGL(siren_fwd_reduce_kernel, dim3((unsigned)ceil_div(fwd_slab_floats(hidden), 256)),
                     dim3(256), 0, st, r);
  return check_launch("siren_fwd_reduce_kernel");
}

namespace mri {
namespace {
int64_t slab_region_bytes(int64_t n, int hidden, int n_sine) {
  const int64_t chain = (int64_t)chain_blocks(hidden, n) * bwd_slab_floats(hidden, n_sine);
  const int64_t wgrad = n_sine > 1 ? (int64_t)wgrad_blocks(n) * wgrad_split(hidden) * hidden * hidden : 0;
  int64_t bytes = std::max(chain, wgrad) * 4;
  // siren_rows.hip: dLoss/dy per row travels from the loss-mode forward to the backward kernel behind the slabs
  if (rows_supported(hidden, n_sine)) bytes = std::max(bytes, rows_dy_offset(n, hidden, n_sine) + n * 4);
  return (bytes + 255) / 256 * 256;
}
}  // namespace
}  // namespace mri

extern "C" int64_t mri_siren_backward_workspace_bytes(int64_t n, int32_t hidden,
                                                      int32_t n_sine_layers) {
  if (n < 1 || !chain_supported(1, hidden, n_sine_layers, 1)) return -1;
  return slab_region_bytes(n, hidden, n_sine_layers) + split_region_bytes(hidden, n_sine_layers);
}

extern "C" int mri_siren_backward(const float* x, const float* dy, int64_t n, int32_t dim_in,
                                  int32_t hidden, int32_t n_sine_layers,
                                  const float* const* weight, const float* const* act,
                                  const float* const* deriv, float* const* dz,
                                  float* const* d_weight, float* const* d_bias,
                                  int32_t head_done, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  MRI_REQUIRE(chain_supported(dim_in, hidden, n_sine_layers, 1),
              "fused SIREN chain: %d -> %d x %d -> 1 is not supported", dim_in, hidden,
              n_sine_layers);
  MRI_REQUIRE(!head_done || n_sine_layers >= 2, "head_done needs >= 2 sine layers");
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && (dy || head_done) && weight && act && deriv && dz && d_weight && d_bias,
              "NULL pointer");
  const int L = n_sine_layers;
  const int64_t need = mri_siren_backward_workspace_bytes(n, hidden, L);
  MRI_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "SIREN backward needs a 16-byte aligned workspace of %lld bytes "
              "(mri_siren_backward_workspace_bytes)", (long long)need);
  hipStream_t st = (hipStream_t)stream;
  char* const wtsplit = static_cast<char*>(workspace) + slab_region_bytes(n, hidden, L);
  BwdArgs a{};
  a.x = x, a.dy = dy, a.n = n, a.dim_in = dim_in, a.n_sine = L;
  a.partial = static_cast<float*>(workspace);
  a.dy_ws = reinterpret_cast<const float*>(static_cast<const char*>(workspace) + rows_dy_offset(n, hidden, L));
  for (int l = 0; l <= L; ++l) {
    MRI_REQUIRE(weight[l] && d_weight[l] && d_bias[l], "NULL parameter / gradient pointer (layer %d)", l);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(weight[l]) & 15) == 0, "weights must be 16-byte aligned");
    a.w[l] = weight[l];
  }
  for (int l = 0; l < L; ++l) {
    const bool on_chip = head_done && l == L - 1;  // the loss-mode forward never stored these
    MRI_REQUIRE((on_chip || (act[l] && deriv[l])) && (l == 0 || dz[l]),
                "NULL activation buffer (layer %d)", l);
    MRI_REQUIRE((on_chip || (reinterpret_cast<uintptr_t>(act[l]) & 15) == 0) &&
                    (l == 0 || (reinterpret_cast<uintptr_t>(dz[l]) & 15) == 0),
                "activation buffers must be 16-byte aligned");
    a.deriv[l] = on_chip ? nullptr : deriv[l];
    a.dz[l] = dz[l];
  }
  a.act_last = head_done ? nullptr : act[L - 1];
  a.head_done = head_done ? 1 : 0;
  a.wtsplit = wtsplit;
  if (int rc = split_weights(weight, L, hidden, true, wtsplit, st)) return rc;
  const bool rows = options().siren_rows && rows_backward_supported(hidden, L, dim_in, head_done);
  if (head_done) {
    const LossCall* found = find_loss_call(dz[L - 1]);
    const LossCall c = found ? *found : LossCall{};
    MRI_REQUIRE(found && c.workspace == workspace && c.n == n && c.hidden == hidden && c.n_sine == L &&
                    c.rows == (rows ? 1 : 0),
                "mri_siren_backward(head_done = 1) continues the mri_siren_forward_loss call before it: same n, network, "
                "dz[n_sine_layers - 1] = dz_last, workspace and \"siren_rows\" option (got n %lld / %lld, hidden %d / %d, "
                "layers %d / %d, rows kernels %d / %d)", (long long)n, (long long)c.n, hidden, c.hidden, L, c.n_sine,
                rows ? 1 : 0, c.rows);
  }
  if (int rc = rows ? backward_rows(a, st) : backward_any(hidden, a, st)) return rc;
  BwdReduceArgs r{};
  r.partial = a.partial, r.slabs = rows ? rows_blocks(n) : chain_blocks(hidden, n), r.hidden = hidden, r.n_sine = L;
  r.dim_in = dim_in;
  for (int l = 0; l <= L; ++l) r.d_w[l] = d_weight[l], r.d_b[l] = d_bias[l];
  hipLaunchKernelGGL(siren_bwd_reduce_kernel,
                     dim3((unsigned)ceil_div(bwd_slab_floats(hidden, L), 256)), dim3(256), 0, st, r);
  if (int rc = check_launch("siren_bwd_reduce_kernel")) return rc;
  for (int l = L - 1; l >= 1; --l) {  // dW_l = dz_l^T a_{l-1}
    WgradArgs g{};
    g.dz = dz[l], g.act = act[l - 1], g.n = n, g.partial = static_cast<float*>(workspace);
    if (int rc = wgrad_any(hidden, g, d_weight[l], st)) return rc;
  }
  return MRI_OK;
}

#ifdef SIREN_PROFILE
extern "C" int mri_debug_set_siren_profile(long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mri::g_siren_profile), &device_buffer, sizeof(device_buffer)) ==
                 hipSuccess
             ? 0
             : -1;
}
#endif
