// Wave-level (64-lane) and row-level (16-lane) sums built on DPP row operations instead of
// ds_bpermute shuffles (measured on MI355X: a 3-value segmented scan by shuffles costs
// ~8 us over 1M observations, the dominant term of the point pass).
#pragma once
#include <hip/hip_runtime.h>

namespace ba {

// DPP controls (GFX9 encoding)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// With every row enabled (ROW_MASK 0xf) the "lanes without a source read 0" rule is the instruction's own
// bound_ctrl:1, so no zeroed destination has to be set up in front of each move (2 of the 5 instructions of a
// 64-bit reduction step; the 27-sum kernels spend a third of their vector instructions in these steps).  The
// broadcast steps enable only some rows: the others must keep a 0, which is the `old` operand.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_f64(double src) {   // lanes without a source read 0.0
  constexpr bool all_rows = ROW_MASK == 0xf;
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, ROW_MASK, 0xf, all_rows);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, ROW_MASK, 0xf, all_rows);
  return __hiloint2double(hi, lo);
}

// Inclusive scan-sum over the wave; lane 63 ends with the total.  Fixed order.
__device__ inline double wave_scan_sum_dpp(double x) {
  x += dpp_f64<DPP_ROW_SHR1, 0xf>(x);
  x += dpp_f64<DPP_ROW_SHR2, 0xf>(x);
  x += dpp_f64<DPP_ROW_SHR4, 0xf>(x);
  x += dpp_f64<DPP_ROW_SHR8, 0xf>(x);
  x += dpp_f64<DPP_ROW_BCAST15, 0xa>(x);
  x += dpp_f64<DPP_ROW_BCAST31, 0xc>(x);
  return x;
}
// total of the wave, uniform in every lane
// value of lane `src` (the same in every lane: it goes through a scalar register) in all lanes
__device__ inline double readlane_f64(double v, int src) {
  const int s = __builtin_amdgcn_readfirstlane(src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), s), hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
  return __hiloint2double(hi, lo);
}

__device__ inline double wave_total_dpp(double x) {
  x = wave_scan_sum_dpp(x);
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
  return __hiloint2double(hi, lo);
}

}  // namespace ba
