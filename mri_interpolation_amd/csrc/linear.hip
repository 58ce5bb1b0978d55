// Fully connected layers on the f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces F.linear + activation and its autograd (reference models.py:46-56, 153-156,
// 230-233, 730-736).  gfx950 has no xf32/TF32, and BASELINE.json asks for 1e-5 relative
// fp32 parity, so every product runs on the exact-f32 MFMA (k-ordered fmaf chain).
//
// One kernel template covers the three GEMMs of a layer.  It computes
//        C[i][j] = sum_c P(i, c) * Q(j, c)
// for operands addressed with two strides each (element (o, c) at base + o*os + c*cs), so
// that row-major activations, feature-major encoder output, W and W^T all use the same code:
//   forward          i = batch row m, j = output unit n, c = input unit k
//   backward (data)  i = m,           j = k,             c = n       (Q = W read "transposed")
//   backward (weight) i = n,          j = k,             c = m       (contraction over the batch,
//                                                                    split over blockIdx.z, f32 atomics)
// A 256-thread workgroup (4 waves) owns a BI x BJ tile; each wave owns TI x TJ MFMA tiles of
// 32x32 and keeps them in registers across the contraction loop.  Both operand tiles are
// staged through LDS in 32-deep chunks -- two LDS stages, global loads of chunk c+1 issued into
// registers before the MFMAs of chunk c and written to the other stage after them, one barrier
// per chunk -- in whichever of two images makes the staging stores and the fragment reads
// conflict-free for that operand's memory order:
//   "row" image  [o][c], leading dim 33   (operand contiguous along the contraction)
//   "col" image  [c][o], leading dim O+4  (operand contiguous along the outer index)
#include <math.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "device_math.h"

namespace mri {

// VALU kernels for layers with a tiny input (K <= 8) or output (N <= 4) width (linear_small.hip);
// they return false when the shape is not theirs.
bool small_forward(const float* x, int64_t xrs, int64_t xcs, const float* w, const float* b,
                   int64_t m, int n, int k, int act, float w0, float* y, int64_t ldy, float* deriv,
                   int64_t ldd, hipStream_t st);
bool small_backward_weight(const float* dy, int64_t lddy, const float* x, int64_t xrs, int64_t xcs,
                           int64_t m, int n, int k, float* d_weight, float* d_bias,
                           hipStream_t st);
bool small_backward_data(const float* dy, int64_t lddy, const float* w, int64_t m, int n, int k,
                         int mode, const float* g, int64_t ldg, float* dx, int64_t lddx,
                         hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KB = 32;        // contraction chunk staged per iteration
constexpr int kThreads = 256;

enum Epilogue { EPI_FORWARD = 0, EPI_BACKWARD_DATA = 1, EPI_ATOMIC = 2 };

struct Operand {
  const float* ptr;
  int64_t os, cs;  // strides of the outer and the contraction index (elements)
  int mode;        // 0 scalar loads, 1 float4 along c ("row" image), 2 float4 along o ("col" image)
};

struct GemmArgs {
  Operand p, q;
  int64_t I, J, C;   // extents
  int64_t c_per_split;
  float* out;        // C[i*ldo + j]
  int64_t ldo;
  // EPI_FORWARD
  const float* bias;
  int act;
  float w0;
  float* deriv_out;
  int64_t ldd_out;
  // EPI_BACKWARD_DATA
  int deriv_mode;
  const float* deriv_in;
  int64_t ldd_in;
  // EPI_ATOMIC
  float* rowsum_out;  // += sum_c P(i, c) for the workgroups of column tile 0 (d_bias)
  int gj;             // number of column tiles (set by launch_cfg)
  int xcd_order;      // tile order described in gemm_kernel (set by launch_cfg)
};

template <int O>
struct TileImage {
  static constexpr int kRowLd = KB + 1;
  static constexpr int kColLd = O + 4;
  static constexpr int kFloats = (O * kRowLd > KB * kColLd) ? O * kRowLd : KB * kColLd;
};

// An operand tile is O (outer) x KB (contraction) floats = O*KB/256 per thread.  It travels in two
// halves so that the global loads of chunk c+1 are in flight while chunk c is on the MFMAs:
//   load_tile   global -> registers (zero-filling out-of-range elements)
//   store_tile  registers -> LDS image
template <int O>
struct TileRegs {
  static constexpr int kPer = O * KB / kThreads;  // floats per thread
  float v[kPer];
};

// hipcc wraps every conditionally executed load in its own branch and waits for it before the
// join, which serialises the loads of a tile (one HBM latency each).  So: tiles that lie fully
// inside the operand (almost all of them) take a path with NO per-element conditions, and edge
// tiles read clamped, always-valid addresses and zero the out-of-range values afterwards.
template <int O>
__device__ __forceinline__ void load_tile(TileRegs<O>& r, const Operand& op, int64_t o0,
                                          int64_t o_end, int64_t c0, int64_t c_end) {
  const int t = threadIdx.x;
  constexpr int kPer = TileRegs<O>::kPer;
  const bool interior = o0 + O <= o_end && c0 + KB <= c_end;  // workgroup-uniform
  if (interior && op.mode == 1) {  // float4 along the contraction
    // one 64-bit address per thread and chunk; the passes differ by a workgroup-uniform offset
    const float* __restrict__ p0 = op.ptr + (o0 + t / (KB / 4)) * op.os + c0 + (t % (KB / 4)) * 4;
    const int64_t pass = (int64_t)(kThreads / (KB / 4)) * op.os;
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(p0 + q * pass);
      r.v[4 * q] = x.x, r.v[4 * q + 1] = x.y, r.v[4 * q + 2] = x.z, r.v[4 * q + 3] = x.w;
    }
  } else if (interior && op.mode == 2) {  // float4 along the outer index
    static_assert(kThreads % (O / 4) == 0, "a pass covers whole contraction rows");
    const float* __restrict__ p0 = op.ptr + (c0 + t / (O / 4)) * op.cs + o0 + (t % (O / 4)) * 4;
    const int64_t pass = (int64_t)(kThreads / (O / 4)) * op.cs;
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(p0 + q * pass);
      r.v[4 * q] = x.x, r.v[4 * q + 1] = x.y, r.v[4 * q + 2] = x.z, r.v[4 * q + 3] = x.w;
    }
  } else {
    // element (o, c) of the tile for register slot q, in the order store_tile expects
    const int64_t o_last = o_end - 1, c_last = c_end - 1;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      int o, c;
      if (op.mode == 1) {
        const int v = t + (q / 4) * kThreads;
        o = v / (KB / 4), c = (v % (KB / 4)) * 4 + (q & 3);
      } else if (op.mode == 2) {
        const int v = t + (q / 4) * kThreads;
        c = v / (O / 4), o = (v % (O / 4)) * 4 + (q & 3);
      } else {
        const int v = t + q * kThreads;
        o = v / KB, c = v % KB;
      }
      const int64_t go = o0 + o, gc = c0 + c;
      const float x = op.ptr[min(go, o_last) * op.os + min(gc, c_last) * op.cs];
      r.v[q] = (go <= o_last && gc <= c_last) ? x : 0.f;
    }
  }
}

template <int O>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const TileRegs<O>& r,
                                           int mode) {
  const int t = threadIdx.x;
  constexpr int kPer = TileRegs<O>::kPer;
  if (mode == 1) {
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q) {
      const int v = t + q * kThreads;
      const int o = v / (KB / 4), c4 = (v % (KB / 4)) * 4;
      float* dst = lds + o * TileImage<O>::kRowLd + c4;
      dst[0] = r.v[4 * q], dst[1] = r.v[4 * q + 1], dst[2] = r.v[4 * q + 2], dst[3] = r.v[4 * q + 3];
    }
  } else if (mode == 2) {
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q) {
      const int v = t + q * kThreads;
      const int c = v / (O / 4), o4 = (v % (O / 4)) * 4;
      *reinterpret_cast<float4*>(lds + c * TileImage<O>::kColLd + o4) =
          make_float4(r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int v = t + q * kThreads;
      lds[(v / KB) * TileImage<O>::kRowLd + (v % KB)] = r.v[q];
    }
  }
}

// Phase timing for tools/gemm_phases.py (a tools-only build with -DMRI_GEMM_PROFILE; nothing in the
// shipped library): shader-clock cycles per wave, summed over the chunks of its tile.
#ifdef MRI_GEMM_PROFILE
__device__ long long* g_gemm_profile = nullptr;
#define GP_BEGIN long long gp_t = clock64(); long long gp_acc[8] = {};
#define GP_MARK(i) { const long long gp_n = clock64(); gp_acc[i] += gp_n - gp_t; gp_t = gp_n; }
#define GP_END                                                                              \
  if (g_gemm_profile && (threadIdx.x & 63) == 0) {                                          \
    long long* dst = g_gemm_profile + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;   \
    for (int q = 0; q < 8; ++q) dst[q] = gp_acc[q];                                         \
  }
#else
#define GP_BEGIN
#define GP_MARK(i)
#define GP_END
#endif

template <int WI, int WJ, int TI, int TJ, int EPI>
__global__ __launch_bounds__(kThreads, 2) void gemm_kernel(const GemmArgs a) {
  constexpr int BI = WI * TI * 32, BJ = WJ * TJ * 32;
  static_assert(WI * WJ * kWave == kThreads, "4 waves per workgroup");
  constexpr int kStage = TileImage<BI>::kFloats + TileImage<BJ>::kFloats;
  __shared__ float lds[2 * kStage];  // two stages: chunk c is read while chunk c+1 is written

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = wave / WJ, wj = wave % WJ;
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (speed only): workgroups b and b + 8 share an XCD (round-robin
  // dispatch), so the gj column tiles of one row block are given ids b, b + 8, ...: they run
  // at about the same time on the SAME XCD and the second one finds the row block's operand
  // tile in that L2 instead of fetching it from HBM again.
  // Only for many row blocks: with a handful (weight gradients), padding to groups of 8 would
  // leave whole XCDs without work (and ordering the tiles of one batch split onto one XCD
  // measured no gain there).
  const int64_t b = blockIdx.x;
  const int gj = a.gj;
  int64_t ti_blk;
  const int64_t split = blockIdx.z;
  int tj_blk;
  if (a.xcd_order == 1) {
    ti_blk = (b / (8 * gj)) * 8 + b % 8;
    tj_blk = (int)((b / 8) % gj);
    if (ti_blk * BI >= a.I) return;  // padding of the last group of 8 row blocks
  } else {
    ti_blk = b / gj;
    tj_blk = (int)(b % gj);
  }
  const int64_t i0 = ti_blk * BI, j0 = (int64_t)tj_blk * BJ;
  const int64_t c_begin = split * a.c_per_split;
  const int64_t c_end = min(a.C, c_begin + a.c_per_split);

  // per-lane fragment addresses inside the two images
  const int p_so = a.p.mode == 2 ? 1 : TileImage<BI>::kRowLd;
  const int p_sc = a.p.mode == 2 ? TileImage<BI>::kColLd : 1;
  const int q_so = a.q.mode == 2 ? 1 : TileImage<BJ>::kRowLd;
  const int q_sc = a.q.mode == 2 ? TileImage<BJ>::kColLd : 1;
  const int pf_off = (wi * TI * 32 + l31) * p_so + lh * p_sc;
  const int qf_off = TileImage<BI>::kFloats + (wj * TJ * 32 + l31) * q_so + lh * q_sc;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  float rowsum = 0.f;
  const bool want_rowsum = EPI == EPI_ATOMIC && a.rowsum_out != nullptr && tj_blk == 0;

  GP_BEGIN
  TileRegs<BI> pr;
  TileRegs<BJ> qr;
  if (c_begin < c_end) {
    load_tile<BI>(pr, a.p, i0, a.I, c_begin, c_end);
    load_tile<BJ>(qr, a.q, j0, a.J, c_begin, c_end);
    store_tile<BI>(lds, pr, a.p.mode);
    store_tile<BJ>(lds + TileImage<BI>::kFloats, qr, a.q.mode);
  }
  __syncthreads();
  GP_MARK(0)  // prologue: first chunk load + store + barrier
  int stage_id = 0;
  for (int64_t c0 = c_begin; c0 < c_end; c0 += KB) {
    const float* __restrict__ cur = lds + stage_id * kStage;
    float* __restrict__ nxt = lds + (stage_id ^ 1) * kStage;
    const bool more = c0 + KB < c_end;
    if (more) {  // global loads of the next chunk fly while this chunk is on the matrix cores
      load_tile<BI>(pr, a.p, i0, a.I, c0 + KB, c_end);
      load_tile<BJ>(qr, a.q, j0, a.J, c0 + KB, c_end);
    }
    GP_MARK(1)  // issue of the next chunk's global loads
    const float* __restrict__ pf = cur + pf_off;
    const float* __restrict__ qf = cur + qf_off;
    const int kc = (int)min((int64_t)KB, c_end - c0);
    if (want_rowsum && threadIdx.x < BI) {
      const float* row = cur + threadIdx.x * p_so;
      for (int c = 0; c < kc; ++c) rowsum += row[c * p_sc];
    }
    // Fragment reads software-pipelined in groups of U contraction pairs (the reads of group
    // g+1 are issued before the MFMAs of group g): bounded register use -- a fully unrolled
    // chunk lets the compiler hoist all 64 fragment reads and spill.
    constexpr int U = 4;
    auto fetch = [&](float (&pa)[U][TI], float (&qb)[U][TJ], int kk0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) pa[u][ti] = pf[ti * 32 * p_so + (kk0 + 2 * u) * p_sc];
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) qb[u][tj] = qf[tj * 32 * q_so + (kk0 + 2 * u) * q_sc];
      }
    };
    auto compute = [&](const float (&pa)[U][TI], const float (&qb)[U][TJ], int n_steps) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < n_steps) {
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
              acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[u][ti], qb[u][tj],
                                                                 acc[ti][tj], 0, 0, 0);
        }
    };
    {
      // kc <= KB = 32 -> at most 16 pairs = 4 groups; reads beyond kc stay inside the (zero
      // filled) stage, their products are skipped
      const int pairs = (kc + 1) / 2;
      float pa0[U][TI], qb0[U][TJ], pa1[U][TI], qb1[U][TJ];
      fetch(pa0, qb0, 0);
      fetch(pa1, qb1, 2 * U);
      compute(pa0, qb0, pairs);
      fetch(pa0, qb0, 4 * U);
      compute(pa1, qb1, pairs - U);
      fetch(pa1, qb1, 6 * U);
      compute(pa0, qb0, pairs - 2 * U);
      compute(pa1, qb1, pairs - 3 * U);
    }
    GP_MARK(2)  // fragment reads + MFMAs of this chunk
    if (more) {
      store_tile<BI>(nxt, pr, a.p.mode);
      store_tile<BJ>(nxt + TileImage<BI>::kFloats, qr, a.q.mode);
    }
    GP_MARK(3)  // wait for the loads, store them to the other stage
    __syncthreads();  // next stage complete, and everyone is done reading the current one
    GP_MARK(4)  // barrier
    stage_id ^= 1;
  }

  // ---- epilogue: C/D layout of the 32x32 f32 MFMA: col = lane & 31,
  //      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // Tiles that lie fully inside the output (workgroup-uniform test) run without per-element
  // bounds checks, so the loads of the derivative matrix are issued back to back.
  auto emit = [&](auto checked_tag) {
    constexpr bool kChecked = decltype(checked_tag)::value;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const int64_t j = j0 + (wj * TJ + tj) * 32 + l31;
        if (kChecked && j >= a.J) continue;
        float bj = 0.f;
        if (EPI == EPI_FORWARD && a.bias) bj = a.bias[j];
        const int64_t i_first = i0 + (wi * TI + ti) * 32 + 4 * lh;
        float g[16];
        if (EPI == EPI_BACKWARD_DATA && a.deriv_mode != MRI_DERIV_NONE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t i = i_first + (r & 3) + 8 * (r >> 2);
            g[r] = a.deriv_in[(kChecked ? min(i, a.I - 1) : i) * a.ldd_in + j];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t i = i_first + (r & 3) + 8 * (r >> 2);
          if (kChecked && i >= a.I) continue;
          float v = acc[ti][tj][r];
          if (EPI == EPI_FORWARD) {
            v += bj;
            float d = 1.0f;
            switch (a.act) {
              case MRI_ACT_RELU:
                v = fmaxf(v, 0.f);
                break;
              case MRI_ACT_SINE: {
                const float u = a.w0 * v;
                float sn, cs;
                sincos_fast(u, &sn, &cs);
                v = sn;
                d = a.w0 * cs;
              } break;
              case MRI_ACT_GELU:
                d = gelu_grad_f(v);
                v = gelu_f(v);
                break;
              default:
                break;
            }
            a.out[i * a.ldo + j] = v;
            if (a.deriv_out) a.deriv_out[i * a.ldd_out + j] = d;
          } else if (EPI == EPI_BACKWARD_DATA) {
            if (a.deriv_mode == MRI_DERIV_MUL) {
              v *= g[r];
            } else if (a.deriv_mode == MRI_DERIV_RELU_MASK) {
              v = g[r] > 0.f ? v : 0.f;
            }
            a.out[i * a.ldo + j] = v;
          } else {
            atomicAdd(a.out + i * a.ldo + j, v);
          }
        }
      }
    }
  };
  // Forward tiles inside the output whose rows can be written 16 bytes at a time go through LDS
  // (free by now): the accumulator layout gives each lane single floats of 32-wide row pieces,
  // i.e. BI*BJ/256 four-byte store instructions per thread; staged, a thread writes float4s and
  // a wave 512-byte runs of a row.  The per-float stores were 27 % (ReLU) / 43 % (sine) of a
  // wave's time in a 256 x 256 layer (tools/gemm_phases.py).
  constexpr int kOutLd = BJ + 4;
  constexpr bool kCanStage = BI * kOutLd <= 2 * kStage && BJ % 4 == 0 && (BI * BJ / 4) % kThreads == 0;
  const bool inside = i0 + BI <= a.I && j0 + BJ <= a.J;
  bool staged = false;
  if constexpr (EPI == EPI_FORWARD && kCanStage) {
    const bool vec_ok = a.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                        (!a.deriv_out || (a.ldd_out % 4 == 0 &&
                                          (reinterpret_cast<uintptr_t>(a.deriv_out) & 15) == 0));
    if (inside && vec_ok) {
      staged = true;
      float dv[TI][TJ][16];  // derivative, kept while the activation is on its way out
      auto stage = [&](auto&& value_of) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (wi * TI + ti) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
              lds[row * kOutLd + (wj * TJ + tj) * 32 + l31] = value_of(ti, tj, r);
            }
      };
      auto flush = [&](float* __restrict__ dst, int64_t ld) {
        typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < BI * BJ / 4 / kThreads; ++q) {
          const int v = threadIdx.x + q * kThreads;
          const int row = v / (BJ / 4), c4 = (v % (BJ / 4)) * 4;
          const f4v x = *reinterpret_cast<const f4v*>(lds + row * kOutLd + c4);
          *reinterpret_cast<f4v*>(dst + (i0 + row) * ld + j0 + c4) = x;
        }
      };
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const float bj = a.bias ? a.bias[j0 + (wj * TJ + tj) * 32 + l31] : 0.f;
          if (a.act == MRI_ACT_SINE) {  // two elements per packed-f32 instruction slot
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              float s0, c0, s1, c1;
              sincos_fast2(a.w0 * (acc[ti][tj][r] + bj), a.w0 * (acc[ti][tj][r + 1] + bj), &s0,
                           &c0, &s1, &c1);
              acc[ti][tj][r] = s0, acc[ti][tj][r + 1] = s1;
              dv[ti][tj][r] = a.w0 * c0, dv[ti][tj][r + 1] = a.w0 * c1;
            }
            continue;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[ti][tj][r] + bj, d = 1.0f;
            switch (a.act) {
              case MRI_ACT_RELU:
                v = fmaxf(v, 0.f);
                break;
              case MRI_ACT_SINE: {
                float sn, cs;
                sincos_fast(a.w0 * v, &sn, &cs);
                v = sn;
                d = a.w0 * cs;
              } break;
              case MRI_ACT_GELU:
                d = gelu_grad_f(v);
                v = gelu_f(v);
                break;
              default:
                break;
            }
            acc[ti][tj][r] = v;
            dv[ti][tj][r] = d;
          }
        }
      __syncthreads();  // every wave is done with the operand stages
      stage([&](int ti, int tj, int r) { return acc[ti][tj][r]; });
      __syncthreads();
      flush(a.out, a.ldo);
      if (a.deriv_out) {
        __syncthreads();
        stage([&](int ti, int tj, int r) { return dv[ti][tj][r]; });
        __syncthreads();
        flush(a.deriv_out, a.ldd_out);
      }
    }
  }
  if (!staged) {
    if (inside)
      emit(std::false_type{});
    else
      emit(std::true_type{});
  }
  if (want_rowsum && threadIdx.x < BI && i0 + threadIdx.x < a.I)
    atomicAdd(a.rowsum_out + i0 + threadIdx.x, rowsum);
  GP_MARK(5)  // epilogue
  GP_END
}

// elementwise dy *= g
__global__ __launch_bounds__(256) void apply_deriv_kernel(float* __restrict__ dy, int64_t lddy,
                                                          int mode, const float* __restrict__ g,
                                                          int64_t ldd, int64_t m, int n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * n) return;
  const int64_t r = e / n, c = e % n;
  const float gv = g[r * ldd + c];
  float v = dy[r * lddy + c];
  if (mode == MRI_DERIV_MUL)
    v *= gv;
  else if (mode == MRI_DERIV_RELU_MASK)
    v = gv > 0.f ? v : 0.f;
  dy[r * lddy + c] = v;
}

// ------------------------------------------------------------------------------ host side
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

Operand make_operand(const float* ptr, int64_t os, int64_t cs) {
  Operand o{ptr, os, cs, 0};
  if (cs == 1 && os % 4 == 0 && aligned16(ptr))
    o.mode = 1;
  else if (os == 1 && cs % 4 == 0 && aligned16(ptr))
    o.mode = 2;
  return o;
}

template <int WI, int WJ, int TI, int TJ, int EPI>
int launch_cfg(GemmArgs& a, int splits, hipStream_t st) {
  constexpr int BI = WI * TI * 32, BJ = WJ * TJ * 32;
  const int64_t gi = ceil_div(a.I, BI), gj = ceil_div(a.J, BJ);
  a.gj = (int)gj;
  a.xcd_order = (gi >= 64 && gj > 1) ? 1 : 0;
  const int64_t blocks = a.xcd_order ? ceil_div(gi, 8) * 8 * gj : gi * gj;
  if (blocks >= (1ll << 31) || splits > 65535)
    return fail(MRI_ERR_INVALID_ARGUMENT, "gemm grid too large");
  hipLaunchKernelGGL((gemm_kernel<WI, WJ, TI, TJ, EPI>), dim3((unsigned)blocks, 1, splits),
                     dim3(kThreads), 0, st, a);
  return check_launch("gemm_kernel");
}

// Pick the tile that wastes the least MFMA work for an I x J output.
template <int EPI>
int launch(GemmArgs& a, int splits, hipStream_t st) {
  const int64_t I = a.I, J = a.J;
  if (I <= 32) return launch_cfg<1, 4, 1, 1, EPI>(a, splits, st);  // 32 x 128
  if (J <= 32) return launch_cfg<4, 1, 1, 1, EPI>(a, splits, st);  // 128 x 32
  if (J <= 64) {
    if (I <= 64) return launch_cfg<2, 2, 1, 1, EPI>(a, splits, st);  // 64 x 64
    return launch_cfg<2, 2, 2, 1, EPI>(a, splits, st);               // 128 x 64
  }
  // Forward layers that also write their derivative (sine, GELU) take 64 x 128 tiles even when
  // 128 x 128 would fit: their epilogue (activation, derivative, two staged stores) is as long as
  // the main loop of a 256-deep contraction, and twice as many, half as long workgroups
  // interleave it better with their neighbours' MFMAs (SIREN 5 x 256 forward 9.1 -> 7.2 ms).
  // ReLU / identity layers and the backward GEMMs measured 7 % slower that way.
  const bool long_epilogue = EPI == EPI_FORWARD && a.deriv_out != nullptr;
  if (I <= 64 || long_epilogue) return launch_cfg<2, 2, 1, 2, EPI>(a, splits, st);  // 64 x 128
  return launch_cfg<2, 2, 2, 2, EPI>(a, splits, st);                                     // 128 x 128
}

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int mri_linear_forward(const float* x, int64_t x_row_stride, int64_t x_col_stride,
                                  const float* weight, const float* bias, int64_t m, int32_t n,
                                  int32_t k, int32_t activation, float w0, float* y, int64_t ldy,
                                  float* deriv, int64_t ldd, void* stream) {
  MRI_REQUIRE(m >= 0 && n >= 1 && k >= 1, "bad shape m=%lld n=%d k=%d", (long long)m, n, k);
  MRI_REQUIRE(activation >= MRI_ACT_IDENTITY && activation <= MRI_ACT_GELU, "bad activation %d",
              activation);
  if (m == 0) return MRI_OK;
  MRI_REQUIRE(x && weight && y, "NULL device pointer");
  MRI_REQUIRE(ldy >= n && (!deriv || ldd >= n), "leading dimension smaller than n");
  if (small_forward(x, x_row_stride, x_col_stride, weight, bias, m, n, k, activation, w0, y, ldy,
                    deriv, ldd, (hipStream_t)stream))
    return check_launch("small forward kernel");
  GemmArgs a{};
  a.p = make_operand(x, x_row_stride, x_col_stride);
  a.q = make_operand(weight, k, 1);
  a.I = m, a.J = n, a.C = k, a.c_per_split = k;
  a.out = y, a.ldo = ldy;
  a.bias = bias, a.act = activation, a.w0 = w0;
  a.deriv_out = deriv, a.ldd_out = ldd;
  return launch<EPI_FORWARD>(a, 1, (hipStream_t)stream);
}

extern "C" int mri_linear_backward_data(const float* dy, int64_t lddy, const float* weight,
                                        int64_t m, int32_t n, int32_t k, int32_t deriv_mode,
                                        const float* deriv, int64_t ldd, float* dx,
                                        int64_t dx_row_stride, int64_t dx_col_stride,
                                        void* stream) {
  MRI_REQUIRE(m >= 0 && n >= 1 && k >= 1, "bad shape m=%lld n=%d k=%d", (long long)m, n, k);
  MRI_REQUIRE(deriv_mode >= MRI_DERIV_NONE && deriv_mode <= MRI_DERIV_RELU_MASK,
              "bad deriv_mode %d", deriv_mode);
  if (m == 0) return MRI_OK;
  MRI_REQUIRE(dy && weight && dx, "NULL device pointer");
  MRI_REQUIRE(deriv_mode == MRI_DERIV_NONE || deriv, "deriv_mode %d needs a deriv matrix",
              deriv_mode);
  if (dx_col_stride == 1 &&
      small_backward_data(dy, lddy, weight, m, n, k, deriv_mode, deriv, ldd, dx, dx_row_stride,
                          (hipStream_t)stream))
    return check_launch("small backward-data kernel");
  GemmArgs a{};
  a.C = n, a.c_per_split = n;
  if (dx_col_stride == 1) {
    a.p = make_operand(dy, lddy, 1);    // P(i = m, c = n)
    a.q = make_operand(weight, 1, k);   // Q(j = k, c = n) = W[n][k]
    a.I = m, a.J = k;
    a.out = dx, a.ldo = dx_row_stride;
    a.deriv_mode = deriv_mode, a.deriv_in = deriv, a.ldd_in = ldd;
    return launch<EPI_BACKWARD_DATA>(a, 1, (hipStream_t)stream);
  }
  // feature-major dx: compute the transposed product so that stores stay coalesced
  MRI_REQUIRE(dx_row_stride == 1, "dx must have a unit stride");
  MRI_REQUIRE(deriv_mode == MRI_DERIV_NONE, "deriv_mode needs row-major dx");
  a.p = make_operand(weight, 1, k);   // P(i = k, c = n)
  a.q = make_operand(dy, lddy, 1);    // Q(j = m, c = n)
  a.I = k, a.J = m;
  a.out = dx, a.ldo = dx_col_stride;
  a.act = MRI_ACT_IDENTITY;
  return launch<EPI_FORWARD>(a, 1, (hipStream_t)stream);
}

extern "C" int mri_linear_backward_weight(const float* dy, int64_t lddy, const float* x,
                                          int64_t x_row_stride, int64_t x_col_stride, int64_t m,
                                          int32_t n, int32_t k, float* d_weight, float* d_bias,
                                          void* stream) {
  MRI_REQUIRE(m >= 0 && n >= 1 && k >= 1, "bad shape m=%lld n=%d k=%d", (long long)m, n, k);
  if (m == 0) return MRI_OK;
  MRI_REQUIRE(dy && x && d_weight, "NULL device pointer");
  if (small_backward_weight(dy, lddy, x, x_row_stride, x_col_stride, m, n, k, d_weight, d_bias,
                            (hipStream_t)stream))
    return check_launch("small_k_backward_weight_kernel");
  GemmArgs a{};
  a.p = make_operand(dy, 1, lddy);                     // P(i = n, c = m)
  a.q = make_operand(x, x_col_stride, x_row_stride);   // Q(j = k, c = m)
  a.I = n, a.J = k, a.C = m;
  a.out = d_weight, a.ldo = k;
  a.rowsum_out = d_bias;
  // split the batch so that ~2 workgroups per CU are in flight; each split is a multiple of KB
  const int64_t tiles = ceil_div(n, 128) * ceil_div(k, 128);
  int64_t splits = std::max<int64_t>(1, std::min<int64_t>(512 / tiles, ceil_div(m, 4 * KB)));
  a.c_per_split = ceil_div(ceil_div(m, splits), KB) * KB;
  splits = ceil_div(m, a.c_per_split);
  return launch<EPI_ATOMIC>(a, (int)splits, (hipStream_t)stream);
}

extern "C" int mri_apply_deriv(float* dy, int64_t lddy, int32_t deriv_mode, const float* deriv,
                               int64_t ldd, int64_t m, int32_t n, void* stream) {
  MRI_REQUIRE(m >= 0 && n >= 1, "bad shape");
  MRI_REQUIRE(deriv_mode >= MRI_DERIV_NONE && deriv_mode <= MRI_DERIV_RELU_MASK,
              "bad deriv_mode %d", deriv_mode);
  if (m == 0 || deriv_mode == MRI_DERIV_NONE) return MRI_OK;
  MRI_REQUIRE(dy && deriv, "NULL device pointer");
  const int64_t blocks = ceil_div(m * n, 256);
  hipLaunchKernelGGL(apply_deriv_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, dy, lddy, deriv_mode, deriv, ldd, m, (int)n);
  return check_launch("apply_deriv_kernel");
}

#ifdef MRI_GEMM_PROFILE
extern "C" int mri_debug_set_gemm_profile(long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mri::g_gemm_profile), &device_buffer, sizeof(device_buffer)) ==
                 hipSuccess
             ? 0
             : -1;
}
#endif

