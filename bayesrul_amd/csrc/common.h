// Device-side building blocks shared by every kernel of the SVI/ELBO path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define WAVE 64
constexpr int HALO = 2;                 // zero rows kept before row 0 of a window image (k=5 'same' conv)
constexpr int TILE_ROWS = 32;           // one window = one 32-row MFMA column block (rows >= L are dummies)
constexpr int IMG_ROWS = TILE_ROWS + 2 * HALO;  // rows -2 .. 33

// ------------------------------------------------------------------------------------------
// bf16 helpers (RNE through the hardware convert)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

// ------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator; one call = 128 random bits = 4 normals.
// Streams: key = (seed lo, seed hi); counter = (index lo, index hi | particle, kind | layer<<8, step)
// ------------------------------------------------------------------------------------------
enum : uint32_t { NK_EPSW = 1, NK_RADIAL_R = 2, NK_LRT = 3, NK_SIGN_IN = 4, NK_SIGN_OUT = 5, NK_DROPOUT = 6 };

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: both are quarter rate
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// Box-Muller on 24-bit uniforms: (0,1) open on both sides, |z| <= 5.9
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(u1));
  // v_sin_f32 / v_cos_f32 take revolutions
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ f32x4 philox_normal4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed) {
  const uint4 u = philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
  f32x4 z;
  float a, b;
  box_muller(u.x, u.y, a, b);
  z[0] = a;
  z[1] = b;
  box_muller(u.z, u.w, a, b);
  z[2] = a;
  z[3] = b;
  return z;
}

// ------------------------------------------------------------------------------------------
// softplus exactly as torch (beta=1, threshold=20) and its derivative
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_t(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float dsoftplus_t(float x) { return x > 20.0f ? 1.0f : sigmoid_t(x); }

// ------------------------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LDS row stride (elements) of a window image with `cwp` channels.
//   f32  : RS = 2 (mod 4)   -> the 16 rows of a ds_read_b32 fragment hit 16 distinct even banks
//   bf16 : RS/8 odd         -> the 16 rows of a ds_read_b128 fragment hit 16 distinct 16-B slots
__host__ __device__ inline int img_row_stride(int cwp, bool bf) {
  if (bf) return ((cwp / 8) & 1) ? cwp : cwp + 8;
  return cwp + 2;  // cwp is a multiple of 8
}
