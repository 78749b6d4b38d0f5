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
int* vlni_index_error_counter();        // device counter of out-of-range table indices (or null), vlni_set_index_error_counter
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

// 16-bit paths: Phi(x) = 1 / (1 + exp(-x (c0 + c1 x^2 + c2 x^4))), a minimax fit of the normal CDF (tools/fit_gelu.py): max |error| 3.1e-5 in
// Phi and in x Phi over all x - an order of magnitude below the rounding of a bfloat16 / float16 pre-activation (2^-9 / 2^-12 relative),
// at 9 VALU instructions per GELU instead of 16 (A&S 7.1.26 erf before) and 12 instead of 25 per GELU'. The 256 x 256 GEMM tiles run
// their epilogue with nothing to hide it behind, so those instructions are wall time. float32 (parity) keeps erff.
__device__ __forceinline__ float phi_cdf16(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);           // the quartic turns around at |x| ~ 7.4 .. 11; Phi(+-8) is 0 / 1 to 1e-15
  const float x2 = xc * xc;
  // coefficients pre-multiplied by -log2(e): exp(-t) = exp2(t')
  const float t = xc * (-2.30146679f + x2 * (-0.106544594f + x2 * 9.84420674e-4f));
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));      // v_rcp_f32 (1 ulp); __frcp_rn expands to the 10-instruction IEEE division
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
  if constexpr (sizeof(T) == 2) return x * phi_cdf16(x);
  else return gelu_erf(x);
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) {
  if constexpr (sizeof(T) == 2) {
    const float pdf_x = (x * 0.3989422804014327f) * __builtin_amdgcn_exp2f(x * x * -0.72134752044f);   // x phi(x), exp(-x^2/2) = exp2(-x^2 log2(e)/2)
    return phi_cdf16(x) + pdf_x;
  } else {
    return gelu_erf_grad(x);
  }
}

// GELU and GELU' of one pre-activation together (act == 3: the forward epilogue stores GELU'(z) where act == 1 stores z, and the dgrad epilogue's
// dact == 3 is then ONE multiply by the stored value instead of GELU' of z - three transcendentals per element of a 256 x 256 tile's
// epilogue that nothing hides). The two share Phi(x).
template <typename T> __device__ __forceinline__ float gelu_with_grad_t(float x, float& grad) {
  if constexpr (sizeof(T) == 2) {
    const float cdf = phi_cdf16(x);
    grad = cdf + (x * 0.3989422804014327f) * __builtin_amdgcn_exp2f(x * x * -0.72134752044f);
    return x * cdf;
  } else {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    grad = cdf + x * (0.3989422804014327f * __expf(-0.5f * x * x));
    return x * cdf;
  }
}

// Counter-based dropout: keep(seed, idx) is a pure function, so the backward pass regenerates the forward mask from
// (seed, element index) instead of storing it. idx = row * row_length + col of the tensor the mask applies to.
constexpr unsigned DROP_MUL = 0x9E3779B1u;
__host__ __device__ __forceinline__ unsigned drop_mix(unsigned x) {      // x = idx * DROP_MUL + seed
#ifdef VLNI_DROP_ONE_MUL            // timing-only A/B build (tools/build_variant.sh): the bound of a cheaper hash, not a product option
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
#else
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
#endif
  return x;
}
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned idx, unsigned seed) { return drop_mix(idx * DROP_MUL + seed); }
// returns 0 (dropped) or 1/(1-p) (kept); thr = p * 2^32
__device__ __forceinline__ unsigned eff_seed(unsigned seed, const unsigned* base) { return base ? seed + *base : seed; }
__device__ __forceinline__ float drop_scale(unsigned idx, unsigned seed, unsigned thr, float inv_keep) {
  return drop_hash(idx, seed) >= thr ? inv_keep : 0.f;
}
// the same with the caller's running x = idx * DROP_MUL + seed (an index that moves by a fixed stride moves x by stride * DROP_MUL: one add
// per element instead of two quarter-rate 32-bit multiplies)
__device__ __forceinline__ float drop_scale_x(unsigned x, unsigned thr, float inv_keep) { return drop_mix(x) >= thr ? inv_keep : 0.f; }
static inline unsigned drop_thr(float p) { return p <= 0.f ? 0u : (p >= 1.f ? 0xFFFFFFFFu : (unsigned)((double)p * 4294967296.0)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
