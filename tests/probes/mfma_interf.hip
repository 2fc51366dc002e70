// Diagnostics (not part of the product): how much does a wave that runs epilogue-like work (VALU, DPP, LDS writes, global
// stores) slow down the MFMA stream of the OTHER wave of its SIMD, and vice versa?  Wave 0 issues v_mfma_f32_16x16x4_f32 back
// to back; wave 4 (same SIMD) or wave 1 (another SIMD) runs one kind of side work.  Prints cycles per MFMA of wave 0 and cycles
// per side-work instruction of the other wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_interf tests/probes/mfma_interf.hip && /tmp/mfma_interf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { SIDE_NONE = 0, SIDE_VALU = 1, SIDE_DPP = 2, SIDE_LDSW = 3, SIDE_STORE16 = 4, SIDE_STOREB = 5, SIDE_MIX = 6 };

template <int SIDE>
__global__ __launch_bounds__(512) void interf_kernel(float* out, float* sink, unsigned long long* ticks, int iters, int side_wave) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    const float x = 1.f + lane * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.25f, a1, 0, 0, 0);
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
    if (lane == 0) ticks[0] = t1 - t0;
  } else if (wave == side_wave && SIDE != SIDE_NONE) {
    float v0 = lane, v1 = lane * 2.f, v2 = 1.f, v3 = 3.f;
    float* g = sink + (size_t)lane * 128;       // 512-byte row stride like the activation rows
    unsigned char* gb = (unsigned char*)(sink + 64 * 128) + lane * 32;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // the side work runs for about as long as the MFMA stream: the count is per kind
    const int n = SIDE == SIDE_STORE16 || SIDE == SIDE_STOREB || SIDE == SIDE_MIX ? iters * 8 : iters * 32;
    for (int it = 0; it < n; ++it) {
      if constexpr (SIDE == SIDE_VALU) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v0 = fmaxf(v0 * 1.0001f, v1);
          v1 = fmaxf(v1 * 0.9999f, v2);
          v2 = fmaxf(v2 + v3, v0);
          v3 = fmaxf(v3 * 1.00001f, 0.f);
        }
      } else if constexpr (SIDE == SIDE_DPP) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v0 = fmaxf(v0, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v1), 0x121, 0xf, 0xf, false)));
          v1 = fmaxf(v1, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), 0x12F, 0xf, 0xf, false)));
          v2 = fmaxf(v2, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v3), 0x121, 0xf, 0xf, false)));
          v3 = fmaxf(v3, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0x12F, 0xf, 0xf, false)));
        }
      } else if constexpr (SIDE == SIDE_LDSW) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          *(f32x4*)(lds + ((lane & 15) * 136 + (lane >> 4) * 4 + u * 16)) = f32x4{v0, v1, v2, v3};
          v0 += 1.f;
        }
      } else if constexpr (SIDE == SIDE_STORE16) {
        *(f32x4*)(g + (it & 7) * 4) = f32x4{v0, v1, v2, v3};
        v0 += 1.f;
      } else if constexpr (SIDE == SIDE_STOREB) {
        gb[it & 15] = (unsigned char)it;
      } else if constexpr (SIDE == SIDE_MIX) {
        // one activation-tile epilogue: ReLU, LDS image write, vector store, nibble mask, byte store
        v0 = fmaxf(v0 + 1.f, 0.f); v1 = fmaxf(v1 - 1.f, 0.f); v2 = fmaxf(v2 * 1.01f, 0.f); v3 = fmaxf(v3 - 0.5f, 0.f);
        *(f32x4*)(lds + ((lane & 15) * 136 + (lane >> 4) * 4)) = f32x4{v0, v1, v2, v3};
        *(f32x4*)(g + (it & 7) * 4) = f32x4{v0, v1, v2, v3};
        const unsigned bits = (v0 > 0.f ? 1u : 0u) | (v1 > 0.f ? 2u : 0u) | (v2 > 0.f ? 4u : 0u) | (v3 > 0.f ? 8u : 0u);
        gb[it & 15] = (unsigned char)bits;
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[64 + lane] = v0 + v1 + v2 + v3 + lds[lane];
    if (lane == 0) { ticks[1] = t1 - t0; ticks[2] = (unsigned long long)n; }
  }
}

// one wave: an MFMA stream with its OWN stores / epilogue work placed between the MFMAs (what deferring a job's global
// stores into the next job's k loop would look like)
template <int KIND>
__global__ __launch_bounds__(64) void own_kernel(float* out, float* sink, unsigned long long* ticks, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const float x = 1.f + lane * 1e-3f;
  float* g = sink + (size_t)lane * 128;
  unsigned char* gb = (unsigned char*)(sink + 64 * 128) + lane * 32;
  float v0 = lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.25f, a1, 0, 0, 0);
      if (KIND >= 1 && (u == 2 || u == 10)) {   // 2 vector stores per 32 MFMAs
        __builtin_amdgcn_sched_barrier(0);
        *(f32x4*)(g + ((it + u) & 7) * 4) = f32x4{v0, v0, v0, v0};
        v0 += 1.f;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (KIND >= 2 && (u == 5 || u == 13)) {   // + 2 byte stores
        __builtin_amdgcn_sched_barrier(0);
        gb[(it + u) & 15] = (unsigned char)it;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3] + v0;
  if (lane == 0) ticks[0] = t1 - t0;
}
template <int KIND>
static void run_own(const char* what, float* out, float* sink, unsigned long long* ticks) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    own_kernel<KIND><<<1, 64>>>(out, sink, ticks, iters);
    hipDeviceSynchronize();
  }
  unsigned long long h[2];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-60s %6.1f cycles/MFMA\n", what, h[0] / (iters * 32.0));
}

template <int SIDE>
static void run(const char* what, int side_wave, float* out, float* sink, unsigned long long* ticks) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(ticks, 0, 8 * sizeof(unsigned long long));
    interf_kernel<SIDE><<<1, 512>>>(out, sink, ticks, iters, side_wave);
    hipDeviceSynchronize();
  }
  unsigned long long h[8];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s other wave = %d (%s SIMD): %6.1f cycles/MFMA", what, side_wave, side_wave == 4 ? "same" : "other", h[0] / (iters * 32.0));
  if (h[2]) printf("   side loop: %7.1f cycles per iteration (%llu iterations)", (double)h[1] / h[2], h[2]);
  printf("\n");
}

int main() {
  float *out, *sink;
  unsigned long long* ticks;
  hipMalloc(&out, 512 * sizeof(float));
  hipMalloc(&sink, (64 * 128 + 4096) * sizeof(float));
  hipMalloc(&ticks, 8 * sizeof(unsigned long long));
  run<SIDE_NONE>("MFMA stream alone", 4, out, sink, ticks);
  for (int sw : {4, 1}) {
    run<SIDE_VALU>("+ VALU (16 v_max/v_mul per it.)", sw, out, sink, ticks);
    run<SIDE_DPP>("+ DPP row rotations + v_max", sw, out, sink, ticks);
    run<SIDE_LDSW>("+ 4 ds_write_b128 per it.", sw, out, sink, ticks);
    run<SIDE_STORE16>("+ 1 global_store_dwordx4 per it.", sw, out, sink, ticks);
    run<SIDE_STOREB>("+ 1 global_store_byte per it.", sw, out, sink, ticks);
    run<SIDE_MIX>("+ a tile epilogue per it.", sw, out, sink, ticks);
  }
  run_own<0>("one wave, MFMAs only", out, sink, ticks);
  run_own<1>("one wave, + 2 of its own vector stores per 32 MFMAs", out, sink, ticks);
  run_own<2>("one wave, + 2 vector + 2 byte stores per 32 MFMAs", out, sink, ticks);
  return 0;
}
