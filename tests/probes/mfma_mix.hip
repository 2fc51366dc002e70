// Diagnostics (not part of the product): what keeps ONE wave from issuing v_mfma_f32_16x16x4_f32 every 32 cycles when its
// instruction stream looks like a k-block of tf_fwd_kernel (16 MFMAs on two accumulators, sign folds, LDS operand reads one
// block ahead)?  Prints s_memtime ticks per MFMA for each ingredient.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_mix tests/probes/mfma_mix.hip && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 xor4(f32x4 v, u32x4 m) { return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, v) ^ m); }

enum { NKB = 8 };
// FLAGS: 1 = second contraction through a sign fold (v_xor per 4 MFMAs), 2 = operands from LDS one block ahead,
//        4 = the fold mask from LDS too (table lookup), 8 = scheduling barriers around each block
template <int FLAGS>
__global__ __launch_bounds__(512) void mix_kernel(float* out, unsigned long long* ticks, int iters, unsigned wave_mask) {
  __shared__ __attribute__((aligned(16))) char smem[32 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8 * 1024; i += 512) ((float*)smem)[i] = 1.f + i * 1e-4f;
  __syncthreads();
  if (!((wave_mask >> wave) & 1u)) return;
  f32x4 wa[NKB], wb[NKB];
  for (int k = 0; k < NKB; ++k) {
    wa[k] = f32x4{1.f + k, 2.f, 3.f, 4.f + lane};
    wb[k] = f32x4{0.5f + k, 0.25f, 0.125f, 1.f + lane};
  }
  const char* lb = smem + (lane & 15) * 544 + (lane >> 4) * 16;
  const u32x4* lut = (const u32x4*)(smem + 24 * 1024);
  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  struct Op { f32x4 x[2]; u32x4 m; };
  auto fetch = [&](int kb, int it, Op& o) __attribute__((always_inline)) {
    if constexpr (FLAGS & 2) {
      o.x[0] = *(const f32x4*)(lb + kb * 64);
      o.x[1] = *(const f32x4*)(lb + 16 * 544 + kb * 64);
    } else {
      o.x[0] = f32x4{1.f, 2.f, 3.f, (float)it};
      o.x[1] = f32x4{2.f, 3.f, 4.f, (float)it};
    }
    if constexpr (FLAGS & 4) o.m = lut[(lane + kb + it) & 31];
    else o.m = u32x4{0x80000000u, 0u, 0x80000000u, (unsigned)it << 31};
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    Op cur, nxt;
    fetch(0, it, cur);
    unsigned swl0 = 0, swl16 = 0;
    if constexpr (FLAGS & 32) {
      const unsigned sw = ((const unsigned*)(smem + 28 * 1024))[it & 63];   // the window's sign word (uniform)
      swl0 = sw << (28 - 4 * (lane >> 4));
      swl16 = sw << (12 - 4 * (lane >> 4));
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if constexpr (FLAGS & 16) {
        // staggered: the next block's table address, its LDS reads and this block's sign folds are placed BEHIND the first
        // MFMAs of the block, so no VALU write has to wait for an operand register of an MFMA that has not started yet
#define MF(j) acc[0] = mfma4(wa[kb][j], cur.x[0][j], acc[0]); acc[1] = mfma4(wa[kb][j], cur.x[1][j], acc[1])
#define MG(j) acc[0] = mfma4(wbm[j], cur.x[0][j], acc[0]); acc[1] = mfma4(wbm[j], cur.x[1][j], acc[1])
        __builtin_amdgcn_sched_barrier(0);
        MF(0);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* lp = lut + ((lane + kb + 1 + it) & 31);   // address arithmetic here
        asm volatile("" : "+v"(lp));
        __builtin_amdgcn_sched_barrier(0);
        MF(1);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < NKB) {
          nxt.x[0] = *(const f32x4*)(lb + (kb + 1) * 64);
          nxt.x[1] = *(const f32x4*)(lb + 16 * 544 + (kb + 1) * 64);
          nxt.m = *lp;
        }
        __builtin_amdgcn_sched_barrier(0);
        MF(2);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 wbm = xor4(wb[kb], cur.m);   // the four folds here
        asm volatile("" ::"v"(wbm));
        __builtin_amdgcn_sched_barrier(0);
        MF(3);
        MG(0); MG(1); MG(2); MG(3);
        __builtin_amdgcn_sched_barrier(0);
#undef MF
#undef MG
        cur = nxt;
        continue;
      }
      if constexpr (FLAGS & 32) {
        // fold masks without the LDS table: the block's sign bits sit in a register word (pre-shifted per lane once per
        // job), element j's bit is moved to bit 31 with a constant shift and xor-ed into the operand's sign with one v_bitop3
        if (kb + 1 < NKB) {
          nxt.x[0] = *(const f32x4*)(lb + (kb + 1) * 64);
          nxt.x[1] = *(const f32x4*)(lb + 16 * 544 + (kb + 1) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned w = (kb & 1) ? swl16 : swl0;
        f32x4 wbm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned sh = w << (3 - j);
          wbm[j] = __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(__builtin_bit_cast(unsigned, wb[kb][j]), sh, 0x80000000u, 0x6a));   // a ^ (b & c)
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wa[kb][j], cur.x[mt][j], acc[mt]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wbm[j], cur.x[mt][j], acc[mt]);
        // all eight VALU operations first (own registers, issued in the shadow of the previous block's last MFMAs)
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        continue;
      }
      if (kb + 1 < NKB) fetch(kb + 1, it, nxt);
      if constexpr (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wa[kb][j], cur.x[mt][j], acc[mt]);
      f32x4 wbm = wb[kb];
      if constexpr (FLAGS & 1) wbm = xor4(wb[kb], cur.m);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wbm[j], cur.x[mt][j], acc[mt]);
      if constexpr (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
  if (lane == 0) ticks[wave] = t1 - t0;
}

template <int FLAGS>
static void run(const char* what, unsigned mask, float* out, unsigned long long* ticks) {
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(ticks, 0, 8 * sizeof(unsigned long long));
    mix_kernel<FLAGS><<<1, 512>>>(out, ticks, iters, mask);
    hipDeviceSynchronize();
  }
  unsigned long long h[8];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * NKB * 16;
  printf("%-64s", what);
  for (int w = 0; w < 8; ++w)
    if ((mask >> w) & 1u) printf("  w%d %.1f", w, h[w] / n);
  printf("   ticks/MFMA\n");
}

int main() {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 512 * sizeof(float));
  hipMalloc(&ticks, 8 * sizeof(unsigned long long));
  run<0>("one wave: MFMAs only, 2 chains", 0x01, out, ticks);
  run<8>("one wave: + scheduling barriers", 0x01, out, ticks);
  run<1 | 8>("one wave: + sign folds", 0x01, out, ticks);
  run<2 | 8>("one wave: + LDS operands", 0x01, out, ticks);
  run<1 | 2 | 8>("one wave: + sign folds + LDS operands", 0x01, out, ticks);
  run<1 | 2 | 4 | 8>("one wave: + sign folds + LDS operands + LDS fold table", 0x01, out, ticks);
  run<1 | 2 | 4 | 8>("two waves of one SIMD, all of it", 0x11, out, ticks);
  run<1 | 2 | 4 | 8>("eight waves, all of it", 0xff, out, ticks);
  run<1 | 2 | 4>("eight waves, all of it, no scheduling barriers", 0xff, out, ticks);
  run<1 | 2 | 32>("one wave: folds by shift + v_bitop3 (no table)", 0x01, out, ticks);
  run<1 | 2 | 32>("two waves of one SIMD: shift + v_bitop3", 0x11, out, ticks);
  run<1 | 2 | 32>("eight waves: shift + v_bitop3", 0xff, out, ticks);
  run<1 | 2 | 4 | 16>("one wave: staggered fetch / folds", 0x01, out, ticks);
  run<1 | 2 | 4 | 16>("two waves of one SIMD: staggered", 0x11, out, ticks);
  run<1 | 2 | 4 | 16>("eight waves: staggered", 0xff, out, ticks);
  return 0;
}
