// Diagnostics (not part of the product): issue rate of v_mfma_f32_16x16x4_f32 from ONE wave as a function of the number of
// independent accumulator chains, and from two waves that share a SIMD.  Prints cycles (s_memtime ticks) per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tests/probes/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void chain_kernel(float* out, unsigned long long* ticks, int iters, unsigned wave_mask) {
  const int wave = threadIdx.x >> 6;
  if (!((wave_mask >> wave) & 1u)) return;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.f + threadIdx.x * 1e-3f, b = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
}

template <int NACC>
static void run(const char* what, unsigned mask, float* out, unsigned long long* ticks) {
  const int iters = 2000;
  hipMemset(ticks, 0, 8 * sizeof(unsigned long long));
  chain_kernel<NACC><<<1, 512>>>(out, ticks, iters, mask);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  chain_kernel<NACC><<<1, 512>>>(out, ticks, iters, mask);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * NACC;
  printf("%-28s chains %d:", what, NACC);
  for (int w = 0; w < 8; ++w)
    if ((mask >> w) & 1u) printf("  wave %d %.1f ticks/MFMA", w, h[w] / n);
  printf("   (kernel %.3f ms = %.1f ns per MFMA per wave)\n", ms, ms * 1e6 / n);
}

int main() {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 512 * sizeof(float));
  hipMalloc(&ticks, 8 * sizeof(unsigned long long));
  run<1>("one wave", 0x01, out, ticks);
  run<2>("one wave", 0x01, out, ticks);
  run<3>("one wave", 0x01, out, ticks);
  run<4>("one wave", 0x01, out, ticks);
  run<1>("two waves, same SIMD (0, 4)", 0x11, out, ticks);
  run<2>("two waves, same SIMD (0, 4)", 0x11, out, ticks);
  run<4>("two waves, same SIMD (0, 4)", 0x11, out, ticks);
  run<2>("two waves, SIMDs 0 and 1", 0x03, out, ticks);
  run<2>("eight waves", 0xff, out, ticks);
  return 0;
}
