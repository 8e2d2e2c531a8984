// Shared device/host helpers for libtecogan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tecogan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

void tg_set_error(const char* fmt, ...);

// TG_DETERMINISTIC=1 (read once): ordered-reduction PARITY mode.  Every kernel that accumulates with floating-point atomics is
// launched so that the order of its additions is fixed: reductions (losses, batch-norm sums, column sums) as ONE workgroup (their
// in-block reduction order is fixed by the code, a single atomic per output remains), weight gradients without split-K (one
// workgroup owns an output block over ALL pixels: each dW element receives exactly one add per launch), scatter kernels (the
// three warp / D-input backward passes) as ONE wavefront.  Slow (the fp32 parity step: seconds instead of 60 ms) and only meant
// for the parity tests: a regression is then distinguishable from summation-order noise (tools/c3_repeat.py).
bool tg_det();
// Compute units of the current device (hipDeviceAttributeMultiprocessorCount, cached per device; 256 on MI355X): the persistent
// kernels size their grids from it instead of a literal 256.
int tg_num_cus();
#define TG_DET_GRID(g) (tg_det() ? dim3(1) : dim3(g))
#define TG_DET_WAVE(b) (tg_det() ? dim3(64) : dim3(b))

#define TG_CHECK_ARG(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      tg_set_error("%s: %s", __func__, msg);                      \
      return TG_EINVAL;                                           \
    }                                                             \
  } while (0)

// ---- instrumented launch (csrc/runtime.hip): identical to hipLaunchKernelGGL unless tg_prof_enable(1) was called, in
// which case the dispatch carries a start/stop event pair and is booked under NAME with its algorithmic FLOPs / bytes.
bool tg_prof_slot(const char* name, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1);
#define TG_LAUNCH(NAME, FLOPS, BYTES, kern, grid, block, lds, st, ...)                                   \
  do {                                                                                                   \
    hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                           \
    if (tg_prof_slot(NAME, (double)(FLOPS), (double)(BYTES), &pe0_, &pe1_))                              \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, pe0_, pe1_, 0, __VA_ARGS__);                     \
    else                                                                                                 \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                       \
  } while (0)

#define TG_CHECK_LAUNCH()                                         \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess) {                                       \
      tg_set_error("%s: %s", __func__, hipGetErrorString(e_));    \
      return TG_ELAUNCH;                                          \
    }                                                             \
    return TG_OK;                                                 \
  } while (0)

// ---- bf16 <-> f32 (round to nearest even; NaN kept quiet) ------------------
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round to nearest even, two values per instruction -- hipcc pairs
// adjacent conversions).  The software sequence it replaces was 5 VALU ops per value: ~40 % of the instructions of a
// 64-element conv epilogue.
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, static_cast<__bf16>(f)); }

// 8 bf16 <-> 8 f32 (one 16-byte vector): the unit of the vectorised pointwise kernels
__device__ __forceinline__ void bf8_unpack(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(w[e] << 16);
    f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 bf8_pack(const float (&f)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = (uint32_t)f2bf(f[2 * e]) | ((uint32_t)f2bf(f[2 * e + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<u16> {
  static __device__ __forceinline__ float ld(const u16* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(u16* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float act_fwd(float v, int act, float alpha) {
  switch (act) {
    case TG_ACT_RELU: return v > 0.f ? v : 0.f;
    case TG_ACT_LRELU: return v > 0.f ? v : v * alpha;
    case TG_ACT_TANH: return tanhf(v) * alpha;
    case TG_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
// derivative factor from the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act, float alpha) {
  switch (act) {
    case TG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case TG_ACT_LRELU: return y > 0.f ? 1.f : alpha;
    case TG_ACT_TANH: return alpha - y * y / alpha;
    case TG_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int grid_1d(int64_t work, int block, int cap = 256 * 16) {
  int64_t g = cdiv64(work, block);
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}
