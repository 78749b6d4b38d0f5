// Shared device/host helpers for the vlni operator library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VLNI_F32 0
#define VLNI_BF16 1
#define VLNI_F16 2

// error codes returned by every extern "C" entry point
#define VLNI_OK 0
#define VLNI_EINVAL (-1)    // bad argument (shape / alignment / dtype)
#define VLNI_ELAUNCH (-2)   // hipLaunch failed
#define VLNI_EUNSUP (-3)    // shape outside what the kernels cover

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void vlni_set_error(const char* fmt, ...);
const unsigned* vlni_seed_base();      // device pointer set by vlni_set_dropout_seed_base (or null)

#define VLNI_CHECK(cond, code, ...)                 \
  do {                                              \
    if (!(cond)) {                                  \
      vlni_set_error(__VA_ARGS__);                  \
      return (code);                                \
    }                                               \
  } while (0)

#define VLNI_LAUNCH_CHECK()                                             \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      vlni_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,     \
                     hipGetErrorString(e_));                            \
      return VLNI_ELAUNCH;                                              \
    }                                                                   \
  } while (0)

// ---- dtype helpers -------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(__bf16 x) { return (float)x; }
__device__ __forceinline__ __bf16 f32_to_bf16(float x) { return (__bf16)x; }   // v_cvt_pk_bf16_f32, RNE, NaN kept

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int id = VLNI_F32;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  // 4 consecutive elements
  __device__ static __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
  __device__ static __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct DT<__bf16> {
  static constexpr int id = VLNI_BF16;
  __device__ static __forceinline__ float ld(const __bf16* p) { return (float)*p; }
  __device__ static __forceinline__ void st(__bf16* p, float v) { *p = (__bf16)v; }
  __device__ static __forceinline__ f32x4 ld4(const __bf16* p) {
    bf16x4 t = *(const bf16x4*)p;
    f32x4 r = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    return r;
  }
  __device__ static __forceinline__ void st4(__bf16* p, f32x4 v) {
    bf16x4 t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *(bf16x4*)p = t;
  }
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <> struct DT<_Float16> {
  static constexpr int id = VLNI_F16;
  __device__ static __forceinline__ float ld(const _Float16* p) { return (float)*p; }
  __device__ static __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
  __device__ static __forceinline__ f32x4 ld4(const _Float16* p) {
    f16x4 t = *(const f16x4*)p;
    f32x4 r = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    return r;
  }
  __device__ static __forceinline__ void st4(_Float16* p, f32x4 v) {
    f16x4 t = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *(f16x4*)p = t;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16 path: Abramowitz-Stegun 7.1.26 erf (|err| <= 1.5e-7, far below bf16 resolution): 1 exp + 1 rcp + 6 fma
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __expf(-ax * ax);
  return copysignf(e, x);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
  if constexpr (sizeof(T) == 2) return x * 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
  else return gelu_erf(x);
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) {
  if constexpr (sizeof(T) == 2) {
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
  } else {
    return gelu_erf_grad(x);
  }
}

// Counter-based dropout: keep(seed, idx) is a pure function, so the backward pass regenerates the forward mask from
// (seed, element index) instead of storing it. idx = row * row_length + col of the tensor the mask applies to.
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned idx, unsigned seed) {
  unsigned x = idx * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// returns 0 (dropped) or 1/(1-p) (kept); thr = p * 2^32
__device__ __forceinline__ unsigned eff_seed(unsigned seed, const unsigned* base) { return base ? seed + *base : seed; }
__device__ __forceinline__ float drop_scale(unsigned idx, unsigned seed, unsigned thr, float inv_keep) {
  return drop_hash(idx, seed) >= thr ? inv_keep : 0.f;
}
static inline unsigned drop_thr(float p) { return p <= 0.f ? 0u : (p >= 1.f ? 0xFFFFFFFFu : (unsigned)((double)p * 4294967296.0)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
