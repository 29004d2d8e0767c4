// wave_ops.hpp -- cross-lane primitives shared by the solve kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "lpvmpc_device.hpp"

namespace lpvmpc {

// ---- cross-lane primitives (CDNA4) ------------------------------------------------------------
// A tile element [i][j] lives in lane 8*i+j.  Summing over j stays inside an 8-lane group: two quad
// permutes and a half-row mirror, all DPP modifiers on v_mov (a 64-bit value moves as two dwords).
// Summing over i crosses 16-lane DPP rows: row_ror:8 for lane^8, then v_permlane16_swap /
// v_permlane32_swap (gfx950) for lane^16 / lane^32.  -DLPVMPC_USE_SHFL selects plain ds_bpermute
// shuffles instead (validation only).
template <int CTRL>
__device__ inline double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane has a valid source under these controls; bound_ctrl lets the destination start undefined
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double xor16_sum(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ inline double xor32_sum(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// all-reduce over the column index j (lane bits 0..2) / the row index i (lane bits 3..5) of a tile
__device__ inline double red_j(double v) {
#ifdef LPVMPC_USE_SHFL
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
#else
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror (quads are uniform by now)
#endif
    return v;
}
__device__ inline double red_i(double v) {
#ifdef LPVMPC_USE_SHFL
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
#else
    v += dpp_mov<0x128>(v);   // row_ror:8  == lane ^ 8
    v = xor16_sum(v);
    v = xor32_sum(v);
#endif
    return v;
}
// whole-wavefront all-reduces (sum / max): 8-lane group, row_ror:8, then the two row-crossing swaps
__device__ inline void xor16_pair(double v, double &x, double &y) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}
__device__ inline void xor32_pair(double v, double &x, double &y) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}
__device__ inline double wave_sum(double v) {
#ifdef LPVMPC_USE_SHFL
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
#else
    return red_i(red_j(v));
#endif
}
__device__ inline double wave_max(double v) {
#ifdef LPVMPC_USE_SHFL
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
#else
    v = fmax(v, dpp_mov<0xB1>(v)); v = fmax(v, dpp_mov<0x4E>(v)); v = fmax(v, dpp_mov<0x141>(v));
    v = fmax(v, dpp_mov<0x128>(v));
    double x, y;
    xor16_pair(v, x, y); v = fmax(x, y);
    xor32_pair(v, x, y); v = fmax(x, y);
    return v;
#endif
}
// value of lane SRC in every lane (uniform source lane): two v_readlane
template <int SRC>
__device__ inline double bcast_lane(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC), hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}
// tile element [i][T] in every lane (i, .): broadcast lane T of each 8-lane group.  gfx90a+ moves 64 bits per DPP instruction under
// the row_newbcast controls (lane N of every 16-lane row to the whole row): one v_mov_b64_dpp per half row, bank-masked to its own
// 8 lanes -- two instructions where quad_perm + row_half_mirror on the two dwords took four (tools/microbench/dpp64_neg_probe.hip
// checks the move against the old form on the device)
template <int T>
__device__ inline double bcast_row(double v) {
    const long x = __double_as_longlong(v);
    long y = __builtin_amdgcn_update_dpp(x, x, 0x150 + T, 0xF, 0x3, false);         // row_newbcast:T     -> lanes 0..7 of the row
    y = __builtin_amdgcn_update_dpp(y, x, 0x158 + T, 0xF, 0xC, false);              // row_newbcast:8 + T -> lanes 8..15
    return __longlong_as_double(y);
}
// two independent all-reduces of the same kind, interleaved step by step (fills the DPP hazard slots
// and the dependent-add latency of one with the other)
__device__ inline void red_j2(double &a, double &b) {
#ifdef LPVMPC_USE_SHFL
    a = red_j(a); b = red_j(b);
#else
    double ta = dpp_mov<0xB1>(a), tb = dpp_mov<0xB1>(b); a += ta; b += tb;
    ta = dpp_mov<0x4E>(a); tb = dpp_mov<0x4E>(b); a += ta; b += tb;
    ta = dpp_mov<0x141>(a); tb = dpp_mov<0x141>(b); a += ta; b += tb;
#endif
}
__device__ inline void red_i2(double &a, double &b) {
#ifdef LPVMPC_USE_SHFL
    a = red_i(a); b = red_i(b);
#else
    double ta = dpp_mov<0x128>(a), tb = dpp_mov<0x128>(b); a += ta; b += tb;
    a = xor16_sum(a); b = xor16_sum(b);
    a = xor32_sum(a); b = xor32_sum(b);
#endif
}
// ---- matrix-core primitives (v_mfma_f64_4x4x4_4b_f64, layouts measured on gfx950 by tools/microbench/mfma_f64_probe.hip) ----
// Four independent 4x4x4 products per instruction; with r = lane >> 4, b = (lane >> 2) & 3, c = lane & 3:
//   A[b][i][k] sits in lane (r = k, b, c = i),  B[b][k][j] in lane (r = k, b, c = j),  D[b][i][j] in lane (r = i, b, c = j).
// A stage vector v[0..7] lives in "V layout": lane (r, b, c) holds v[4 (b & 1) + r] -- replicated over c and over b >> 1 --
// so it is at once a B operand (every column j carries the vector) and, coming out as D, the next step's operand.
__device__ inline double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// c - A B: the f64 MFMAs of gfx940+ read the BLGP field as NEG[2:0] (bit 0: the A operand) -- the negation costs no instruction
// and no second copy of a replicated operand (checked against a negated operand on the device: tools/microbench/dpp64_neg_probe.hip)
__device__ inline double mfma4_nega(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1); }
template <int CTRL, int BANK>
__device__ inline double dpp_into(double old, double src) {          // src moved by CTRL into the quads of BANK, other quads keep old
    int ol = __double2loint(old), oh = __double2hiint(old);
    const int sl = __double2loint(src), sh = __double2hiint(src);
    ol = __builtin_amdgcn_update_dpp(ol, sl, CTRL, 0xF, BANK, false);
    oh = __builtin_amdgcn_update_dpp(oh, sh, CTRL, 0xF, BANK, false);
    return __hiloint2double(oh, ol);
}
// lane ^ 4: swaps the two halves of a V-layout vector (row_shl:4 into quads 0, 2; row_shr:4 into quads 1, 3)
__device__ inline double swap_half(double v) {
#ifdef LPVMPC_USE_SHFL
    return __shfl(v, (threadIdx.x & 63) ^ 4);
#else
    double any;                                      // every quad is written by one of the two moves: the start value is irrelevant
    asm volatile("" : "=v"(any));
    return dpp_into<0x114, 0xA>(dpp_into<0x104, 0x5>(any, v), v);
#endif
}
// quads 2, 3 <- quads 0, 1 (lane & ~8): re-replicates the chain result over the second block pair
__device__ inline double dup01(double v) {
#ifdef LPVMPC_USE_SHFL
    return __shfl(v, (threadIdx.x & 63) & ~8);
#else
    return dpp_into<0x128, 0xC>(v, v);
#endif
}
__device__ inline double limit_scaling(double v) {
    v = v < kMinScaling ? 1.0 : v;
    return v > kMaxScaling ? kMaxScaling : v;
}
// 1/sqrt(x) to double precision: the hardware estimate y0 (relative error <= 2^-24.2 measured over [2^-14, 2^14]) and ONE third-order
// step, y = y0 (1 + h/2 + 3 h^2/8) with h = 1 - x y0^2: five instructions on a four-deep chain where two Newton steps took seven on
// a seven-deep one, and closer to the exact value (max relative error 2^-52.7 against 2^-51.9 over 2^20 arguments:
// tools/microbench/dpp64_neg_probe.hip).  Eight of them sit on the dependent chain of every 8x8 Cholesky of the factorisation.
__device__ inline double inv_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double h = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * h, __builtin_fma(0.375, h, 0.5), y);
}
// projection on [lo, hi] (lo <= hi): two instructions (v_max_f64, v_min_f64) instead of compare / select pairs; equal to
// t < lo ? lo : (t > hi ? hi : t) for every non-NaN t (a NaN maps to lo, as OSQP's c_min(c_max(z, l), u) does)
// Written as the two instructions themselves: through fmax / fmin the compiler first quiets each operand that comes from memory
// (a v_max_f64 x, x apiece), which the hardware's IEEE-mode max / min do anyway.
__device__ inline double clipd(double t, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(t), "v"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "v"(hi));
    return r;
}


}  // namespace lpvmpc
