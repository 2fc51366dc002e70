// Shared device helpers of the hand-scheduled kernels (LDS-DMA issue, counted vmcnt waits, DPP row rotation, bf16
// packing / sign folding, the Flipout sign LUT) and the bf16-plane kernels of one-branch DENSE layers (the Linear net's
// Flipout / radial / plain-sampling passes on the split-bf16 plan; the Inception net's dense layers run the K-split
// kernels of kernels_dense_ks.h):
//
//   dense_dx_bf_kernel : dX of a dense layer, one workgroup per 32-row window
//   dense_dw_bf_kernel : dW of a dense layer, K chunks x row splits, operands through ds_read_b64_tr_b16
#pragma once
#include "kernels_group.h"

__device__ __forceinline__ uint2 pack_bf4(f32x4 v) {
  return make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16),
                    (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
}

__device__ __forceinline__ bf16x8 xor_sign(bf16x8 v, bool neg) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 u = __builtin_bit_cast(u32x4, v);
  const unsigned m = neg ? 0x80008000u : 0u;
  u[0] ^= m; u[1] ^= m; u[2] ^= m; u[3] ^= m;
  return __builtin_bit_cast(bf16x8, u);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) char lds_char_t;

// LDS-DMA issued through inline asm so that hipcc does not model it: no compiler-inserted
// vmcnt(0) before later LDS reads / barriers; completion is waited for with explicit counted
// s_waitcnt vmcnt(N) by the issuing (loader) wave.  M0 carries the wave-uniform LDS byte address
// and is restored (cdna_hip_programming.md §5.7).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_char_t*)p; }
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// two floats -> packed bf16 pair (one v_cvt_pk_bf16_f32)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// hi / lo bf16 planes of 4 floats: hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void split4(f32x4 v, uint2& hi, uint2& lo) {
  const uint32_t h01 = cvt_pk(v[0], v[1]), h23 = cvt_pk(v[2], v[3]);
  const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xffff0000u);
  const float r2 = v[2] - __uint_as_float(h23 << 16), r3 = v[3] - __uint_as_float(h23 & 0xffff0000u);
  hi = make_uint2(h01, h23);
  lo = make_uint2(cvt_pk(r0, r1), cvt_pk(r2, r3));
}
// 256-entry table: sign byte (8 channels) -> XOR mask of a bf16x8 fragment (one ds_read_b128)
__device__ __forceinline__ void build_sign_lut(uint4* lut, int tid, int nthreads) {
  for (int b = tid; b < 256; b += nthreads) {
    uint32_t m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = (((b >> (2 * q)) & 1u) << 15) | (((b >> (2 * q + 1)) & 1u) << 31);
    lut[b] = make_uint4(m[0], m[1], m[2], m[3]);
  }
}

// diagnostics only (GroupArgs::dbg, null in production): lane 0 of every wave of workgroup 0 records
// s_memtime at phase `ph` of window iteration `k`
#define BNN_STAMP_DECL(A)                                                                                   \
  const bool stamp_on = (A).dbg != nullptr && blockIdx.x == (unsigned)(A).dbg_block && (threadIdx.x & 63) == 0;                   \
  const int stamp_wave = threadIdx.x >> 6;                                                                  \
  auto stamp = [&](int k, int ph) {                                                                         \
    if (stamp_on && k < 48) (A).dbg[((size_t)stamp_wave * 48 + k) * 8 + ph] = __builtin_amdgcn_s_memtime(); \
  }

// workgroup barrier that only waits for this wave's LDS traffic (never for VMEM)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// value of the lane `rotation` positions away inside the 16-lane DPP row (one VALU move, no LDS crossbar)
template <int CTRL>
__device__ __forceinline__ float rot16(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

#define BNN_WAIT_VMCNT_WIDE(N)                                          \
  do {                                                                  \
    switch (N) {                                                        \
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   \
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;   \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;   \
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;   \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;   \
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;   \
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;   \
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;   \
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;   \
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;   \
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break; \
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break; \
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break; \
      case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break; \
      case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break; \
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break; \
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break; \
      case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break; \
      case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break; \
      case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break; \
      case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break; \
      case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break; \
      case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break; \
      case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break; \
      case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break; \
      case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break; \
      case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break; \
      case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break; \
      case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break; \
      case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break; \
      case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break; \
      case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break; \
      case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break; \
      case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break; \
      case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break; \
      case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break; \
      case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break; \
      case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break; \
      case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break; \
      case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break; \
      case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break; \
      case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break; \
      case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break; \
      case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break; \
      case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break; \
      case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break; \
      case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break; \
      case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break; \
      case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break; \
      case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break; \
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  \
    }                                                                   \
  } while (0)

enum { DN_CH = 128, DN_ROWS = 32 };   // dense layers: channels per K chunk, example rows per window

// ==========================================================================================
// dense_dx_bf_kernel : dX of a dense layer (K = cout <= 64 is tiny, the output is wide).
// One workgroup = one 32-row window; dz (= dY [Y>0]) and its second image (LRT: dz*q, Flipout:
// dz*s_out per row) are staged once into LDS; each of the 8 waves then walks its own 16-channel
// output tiles in batches of 4 (weights from L2, no barrier in the tile loop).
// ==========================================================================================
enum { DDX_WAVES = 8 };   // 8 waves: 256 VGPRs each, room for a batch of 4 tiles' fragments
template <int EM>
__global__ __launch_bounds__(DDX_WAVES * 64) void dense_dx_bf_kernel(const GroupArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const BranchDesc& br = G.br[0];
  const LayerDesc ly = A.layers[br.layer];   // device-resident table: read once, not inside the tile loop
  const Win W = decode_win(G, A.cg, blockIdx.x);
  const int s = W.s;
  const uint32_t* sgi = A.nz.sign_in + ly.sign_in_off * A.nz.examples + (long)W.ex0 * ly.sign_in_words;   // [row][words]
  const int sgi_stride = ly.sign_in_words;
  const int zw = (br.cout + 31) & ~31;          // K of the transposed contraction (multiple of 32)
  const int RS = zw + 8;                        // (zw/8 is even) -> conflict-free b128 rows
  u16* dz = (u16*)smem;
  u16* dz2 = dz + DN_ROWS * RS;
  const TensorRef tg = A.t[br.out_t + T_GRAD], ty = A.t[br.out_t], tq = A.t[br.q_t];
  const TensorRef tin = A.t[G.in_t], tdx = A.t[br.dx_t];
  // ---- stage dz / dz2 (4 channels per unit) ----
  for (int U = tid; U < DN_ROWS * (zw >> 2); U += DDX_WAVES * 64) {
    const int row = U / (zw >> 2), c = (U - row * (zw >> 2)) * 4;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f};
    if (row < W.nvalid && c < br.cout) {
      const long o = (long)(W.out_row0 + row) * tg.ctot + br.out_off + c;
      const int nv = br.cout - c;
      const bool vec = ((tg.ctot & 3) == 0) && ((br.out_off & 3) == 0);
      g = tload4(tg, o, nv, vec);
      if (br.relu) {
        const f32x4 y = tload4(ty, o, nv, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.f ? g[k] : 0.f;
      }
      if constexpr (LRT) {
        const f32x4 q = tload4(tq, o, nv, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) g2[k] = g[k] * q[k];
      } else if constexpr (EM == EM_FLIPOUT) {
        const int bit0 = br.n_off + c;
        const uint32_t word = A.nz.sign_out[ly.sign_out_off * A.nz.examples + (long)(W.ex0 + row) * ly.sign_out_words + (bit0 >> 5)];
        const uint32_t bits = word >> (bit0 & 31);
#pragma unroll
        for (int k = 0; k < 4; ++k) g2[k] = ((bits >> k) & 1u) ? -g[k] : g[k];
      }
    }
    *(uint2*)&dz[row * RS + c] = pack_bf4(g);
    if constexpr (DUAL) *(uint2*)&dz2[row * RS + c] = pack_bf4(g2);
  }
  __syncthreads();
  const int i16 = lane & 15, g4 = lane >> 4;
  const int nks = zw >> 5;                       // 1 or 2 k-steps
  // B fragments of the window (both m-tiles, all k-steps) stay in registers for every tile
  bf16x8 bz[2][2], bz2[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      bz[ks][mt] = bz2[ks][mt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (ks < nks) {
        const int o = (mt * 16 + i16) * RS + ks * 32 + g4 * 8;
        bz[ks][mt] = *(const bf16x8*)&dz[o];
        if constexpr (DUAL) bz2[ks][mt] = *(const bf16x8*)&dz2[o];
      }
    }
  const int ntile = br.cin_p >> 4;
  const long sa = A.ws.slott_stride_a * s, sb = A.ws.slott_stride_b * s;
  const u16* wat = (const u16*)A.ws.at + sa + ly.wt_off;
  const u16* wbt = (const u16*)A.ws.bt + sb + ly.wt_off;
  const int KPt = ly.KPt;
  // Tiles in batches of TB: all weight fragments of a batch are fetched first, then its MFMAs and stores.  A
  // wave that loads and stores shares ONE in-order vmcnt, so the wait for the next batch's fragments also waits
  // for this batch's stores; paying that round trip once per TB tiles instead of once per tile is the point.
  // A wave owns TB CONSECUTIVE tiles (its stores of a row then cover 128 contiguous bytes), and the group order is
  // rotated by the window index so that the workgroups of a particle do not walk the same weight rows in lockstep.
  constexpr int TB = 4;
  const int ngrp = (ntile + TB - 1) / TB;
  const int rot = (int)(blockIdx.x % (unsigned)ngrp);
  for (int g0 = wave; g0 < ngrp; g0 += DDX_WAVES) {
    int grp = g0 + rot;
    if (grp >= ngrp) grp -= ngrp;
    const int t0 = grp * TB;
    bf16x8 wa[TB][2], wb[TB][2];
    uint32_t sw[TB][2];   // flipout: sign_in word of (row, 4 channels of this lane), per m-tile
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wa[j][ks] = wb[j][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (t < ntile && ks < nks) {
          const long wo = (long)(t * 16 + i16) * KPt + br.n_off + ks * 32 + g4 * 8;
          wa[j][ks] = *(const bf16x8*)(wat + wo);
          if constexpr (DUAL) wb[j][ks] = *(const bf16x8*)(wbt + wo);
        }
      }
      if constexpr (EM == EM_FLIPOUT) {
        const int och = t * 16 + 4 * g4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = min(mt * 16 + i16, W.nvalid - 1);
          sw[j][mt] = 0u;
          if (t < ntile) sw[j][mt] = sgi[(long)row * sgi_stride + (och >> 5)] >> (och & 31);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j;
      if (t >= ntile) break;
      const int c0 = t * 16;
      f32x4 acc_a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 acc_b[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks < nks) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][ks], bz[ks][mt], acc_a[mt], 0, 0, 0);
          if constexpr (DUAL) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][ks], bz2[ks][mt], acc_b[mt], 0, 0, 0);
          }
        }
      }
      const int och = c0 + 4 * g4;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        if (row >= W.nvalid) continue;
        f32x4 v = acc_a[mt];
        if constexpr (LRT) {
          f32x4 xv = tload4(tin, (long)(W.in_row0 + row) * tin.ctot + br.in_off + och, 4, true);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += 2.f * bf2f(f2bf(xv[r])) * acc_b[mt][r];
        } else if constexpr (EM == EM_FLIPOUT) {
          const uint32_t bits = sw[j][mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += ((bits >> r) & 1u) ? -acc_b[mt][r] : acc_b[mt][r];
        }
        tstore4(tdx, (long)(W.in_row0 + row) * tdx.ctot + br.in_off + och, v, 4, true);
      }
    }
  }
}

// ==========================================================================================
// dense_dw_bf_kernel : dW of a dense layer.  One workgroup = (particle, 128-channel chunk of the
// input, split of the particle's 32-row windows); its 32 (x2) dW tiles stay in registers while the
// windows stream through LDS (register-prefetched one window ahead).  Flipout signs are per
// example row here, so the sign-multiplied copies (X*s_in, dz*s_out) are explicit images.
// ==========================================================================================
template <int EM>
__global__ __launch_bounds__(512) void dense_dw_bf_kernel(const GroupArgs A, int nchunk, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const BranchDesc& br = G.br[0];
  const LayerDesc& ly = A.layers[br.layer];
  int bid = blockIdx.x;
  const int split = bid % nsplit; bid /= nsplit;
  const int chunk = bid % nchunk;
  const int s = bid / nchunk;
  const int c0 = chunk * DN_CH;
  const int cw = min(DN_CH, br.cin_p - c0);          // multiple of 16
  const int zw = (br.cout + 15) & ~15;               // <= 64
  constexpr int RSX = DN_CH + 8, RSZ = 64 + 8;
  u16* xi = (u16*)smem;
  u16* x2 = xi + DN_ROWS * RSX;
  u16* dz = x2 + DN_ROWS * RSX;
  u16* dz2 = dz + DN_ROWS * RSZ;
  const TensorRef tin = A.t[G.in_t], tg = A.t[br.out_t + T_GRAD], ty = A.t[br.out_t], tq = A.t[br.q_t];
  const u16* g_x = (const u16*)tin.p;
  const int pp = A.cg.per_particle, B = A.cg.B;
  // staging plan: X unit = thread (row = tid >> 4, c8 = tid & 15); dz unit = thread < 256 (row = tid >> 3, c8 = tid & 7)
  const int xr = tid >> 4, xc8 = tid & 15;
  const bool x_on = xc8 * 8 < cw;
  const int zr = tid >> 3, zc8 = tid & 7;
  const bool z_on = tid < 256 && zc8 * 8 < br.cout;
  uint4 px = make_uint4(0, 0, 0, 0), pz = make_uint4(0, 0, 0, 0), py = make_uint4(0, 0, 0, 0), pq = make_uint4(0, 0, 0, 0);
  uint32_t psi = 0, pso = 0;
  int p_nvalid = 0;
  auto prefetch = [&](int wl) {
    const int row0 = s * B + wl * DN_ROWS;
    const int nvalid = min(DN_ROWS, B - wl * DN_ROWS);
    p_nvalid = nvalid;
    px = pz = py = pq = make_uint4(0, 0, 0, 0);
    psi = pso = 0;
    if (x_on && xr < nvalid) {
      px = *(const uint4*)(g_x + (long)(row0 + xr) * tin.ctot + br.in_off + c0 + xc8 * 8);
      if constexpr (EM == EM_FLIPOUT) {
        const int bit0 = c0 + xc8 * 8;
        psi = A.nz.sign_in[ly.sign_in_off * A.nz.examples + (long)(row0 + xr) * ly.sign_in_words + (bit0 >> 5)] >> (bit0 & 31);
      }
    }
    if (z_on && zr < nvalid) {
      const long o = (long)(row0 + zr) * tg.ctot + br.out_off + zc8 * 8;
      pz = *(const uint4*)((const u16*)tg.p + o);
      if (br.relu) py = *(const uint4*)((const u16*)ty.p + o);
      if constexpr (LRT) pq = *(const uint4*)((const u16*)tq.p + o);
      if constexpr (EM == EM_FLIPOUT) {
        const int bit0 = br.n_off + zc8 * 8;
        pso = A.nz.sign_out[ly.sign_out_off * A.nz.examples + (long)(row0 + zr) * ly.sign_out_words + (bit0 >> 5)] >> (bit0 & 31);
      }
    }
  };
  // tiles: t = wave + 8 m, (nt, ct) = (t / 8, t % 8)
  f32x4 acc_a[4], acc_b[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    acc_a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gb_a = 0.f, gb_b = 0.f;
  const int ntl = zw >> 4, ctl = (cw + 15) >> 4;
  auto sgn8 = [](uint4 v, uint32_t bits) {   // flip the sign of bf16 element e where bit e is set
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] ^= (((bits >> (2 * e)) & 1u) << 15) | (((bits >> (2 * e + 1)) & 1u) << 31);
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  auto mul8 = [](uint4 a, uint4 b) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float l = bf2f((u16)(x[e] & 0xffff)) * bf2f((u16)(y[e] & 0xffff));
      const float h = bf2f((u16)(x[e] >> 16)) * bf2f((u16)(y[e] >> 16));
      o[e] = (uint32_t)f2bf(l) | ((uint32_t)f2bf(h) << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  int wl = split;
  if (wl < pp) prefetch(wl);
  const int gq = lane >> 4, qq = (lane >> 2) & 3, pq4 = lane & 3;
  for (; wl < pp; wl += nsplit) {
    __syncthreads();
    // ---- images of this window ----
    {
      *(uint4*)&xi[xr * RSX + xc8 * 8] = px;
      if constexpr (LRT) *(uint4*)&x2[xr * RSX + xc8 * 8] = mul8(px, px);
      if constexpr (EM == EM_FLIPOUT) *(uint4*)&x2[xr * RSX + xc8 * 8] = sgn8(px, psi);
      if (tid < 256) {
        uint4 g = pz;
        if (br.relu) {
          auto msk = [](uint32_t yy) {
            const uint32_t lo = ((yy & 0x8000u) == 0 && (yy & 0x7fffu) != 0) ? 0xffffu : 0u;
            const uint32_t hi = ((yy & 0x80000000u) == 0 && (yy & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
            return lo | hi;
          };
          g.x &= msk(py.x); g.y &= msk(py.y); g.z &= msk(py.z); g.w &= msk(py.w);
        }
        *(uint4*)&dz[zr * RSZ + zc8 * 8] = g;
        if constexpr (LRT) *(uint4*)&dz2[zr * RSZ + zc8 * 8] = mul8(g, pq);
        if constexpr (EM == EM_FLIPOUT) *(uint4*)&dz2[zr * RSZ + zc8 * 8] = sgn8(g, pso);
      }
    }
    __syncthreads();
    if (wl + nsplit < pp) prefetch(wl + nsplit);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int t = wave + 8 * m;
      const int nt = t >> 3, ct = t & 7;
      if (nt < ntl && ct < ctl) {
        const int r0 = 8 * gq + qq;
        const u16* a0 = &dz[r0 * RSZ + nt * 16 + 4 * pq4];
        const u16* b0 = &xi[r0 * RSX + ct * 16 + 4 * pq4];
        acc_a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a0, a0 + 4 * RSZ), tr_frag(b0, b0 + 4 * RSX), acc_a[m], 0, 0, 0);
        if constexpr (DUAL) {
          const u16* a2 = &dz2[r0 * RSZ + nt * 16 + 4 * pq4];
          const u16* b2 = &x2[r0 * RSX + ct * 16 + 4 * pq4];
          acc_b[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a2, a2 + 4 * RSZ), tr_frag(b2, b2 + 4 * RSX), acc_b[m], 0, 0, 0);
        }
      }
    }
    if (chunk == 0 && tid < br.cout) {
      float sa = 0.f, sb = 0.f;
      for (int r = 0; r < DN_ROWS; ++r) {
        sa += bf2f(dz[r * RSZ + tid]);
        // bias gradients: LRT needs sum(dz*q); Flipout's bias gradient arrives through slot A only
        if constexpr (LRT) sb += bf2f(dz2[r * RSZ + tid]);
      }
      gb_a += sa;
      gb_b += sb;
    }
  }
  float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
  float* gwb = A.gw_b + A.gw_stride * s + ly.w_off;
  const int i4 = 4 * (lane >> 4), jc = lane & 15;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int t = wave + 8 * m;
    const int nt = t >> 3, ct = t & 7;
    if (nt >= ntl || ct >= ctl) continue;
    const int c = c0 + ct * 16 + jc;
    if (c >= br.cin_p) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      const long o = (long)(br.n_off + n) * ly.KP + c;
      atomicAdd(gwa + o, acc_a[m][r]);
      if constexpr (DUAL) atomicAdd(gwb + o, acc_b[m][r]);
    }
  }
  if (chunk == 0 && tid < br.cout) {
    atomicAdd(A.gb_a + (long)A.gb_stride * s + ly.bias_off + br.n_off + tid, gb_a);
    if constexpr (LRT) atomicAdd(A.gb_b + (long)A.gb_stride * s + ly.bias_off + br.n_off + tid, gb_b);
  }
}

