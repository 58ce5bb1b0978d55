// Shared helpers for the gfx950 hot-path library (see include/mri_inr.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "mri_inr.h"

namespace mri {

constexpr int kWave = 64;  // CDNA wavefront width

// Thread-local last-error string, set by MRI_FAIL and read through mri_last_error().
char* error_buffer();

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define MRI_REQUIRE(cond, ...)                                        \
  do {                                                                \
    if (!(cond)) return ::mri::fail(MRI_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MRI_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MRI_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Tuning knobs (process-wide, speed only -- never change results beyond fp32 summation order), with ONE
// exception that trades accuracy: bwd_records.
struct Options {
  int xcd_affinity = 1;       // map hash-grid levels to XCDs (blockIdx % 8) so a level's table stays in one L2
  int bwd_lds_max_parts = 256;    // levels cut into more LDS slices than this use global atomics
  int bwd_blocks_per_level = 64;  // target workgroups per level of the LDS backward (hit balance)
  int fwd_pair = 1;               // F = 2 forward: two lanes per (coordinate, level), see hashgrid.hip
  int mlp_stagger = 0;            // fused tiny MLP (H = 128): segments team 1 runs behind team 0 (0 = lockstep: measured best)
  int mlp_x3 = 1;                 // 128-wide decoder on the bf16 matrix pipe with three-term operands (mlp_x3.hip); 0: f32 MFMA team kernel
  int bwd_fuse_dense = 1;         // dense levels share the launch of the record accumulation (fills its last round)
  int bwd_dense_blocks = 96;      // workgroups of the dense-level launch (all dense levels together)
  int bwd_dense_max_parts = 4;    // levels with at most this many table slices skip the records (measured optimum)
  // Table-gradient records of grids with two features per level (hashgrid_bwd.hip): 0 = f32 products
  // (the reference's precision: f32 products, here even summed exactly); 1 = packed 8-byte records, every
  // product rounded to 18-21 significant bits before the exact sum (13 us faster at config 4, NOT f32).
  int bwd_records = 0;
  int siren_rows = 1;             // SirenNet H = 256: the chain kernels that keep a wave's rows in registers (siren_rows.hip)
};
Options& options();

}  // namespace mri
