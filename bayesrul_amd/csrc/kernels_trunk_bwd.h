// Backward of the convolutional trunk of the Inception net, fused (bf16 planes, gfx950).
//
// trunk_dx_kernel : d loss / d (pre-activation) of block 2's 1x1 level (MID) and of block 1 (ACT1) in ONE launch.
//   A window flows through a two-stage pipeline inside the workgroup, ONE barrier per step:
//     step t : loaders : window t+1's dY(ACT2) regs -> LDS (masked with [ACT2 > 0] on the way, plus Flipout's
//                        dz o s_out copy), window t+3's global loads issued (two loader waves alternate windows, so
//                        every load has two steps to land).  The mask sources of the two stages (MID / ACT1 hi planes,
//                        arg-max codes) are LDS-DMA'd by a third loader wave, two steps ahead, behind a counted vmcnt.
//              stage A : dMID  = W6^T dz6 + W8^T dz8  (k3 / k5 level), masked with [MID > 0]      window t
//              stage B : dACT1 = W4^T dz4 + W5^T dz5 + W7^T dz7 + scatter(W9^T dz9), masked       window t-1
//   dz of MID never leaves LDS between the stages; both masked gradients are written once to HBM for the dW kernels
//   (which may re-apply the same mask: idempotent).  Weights: transposed + tap-flipped images, register-stationary;
//   Flipout folds s_in into the rows of the dW fragment and s_out into the dz fragment (sign-bit XOR).
#pragma once
#include "kernels_trunk.h"

// diagnostics builds only (tests/probes/ablate_gpu.sh): timing with parts of the kernel removed; results are wrong.
// The product library is built with TX_ABL == 0: every `if constexpr` below folds away.
#ifndef TX_ABL
#define TX_ABL 0
#endif

enum { TX_NW = 15, TX_THREADS = TX_NW * 64, TX_NJ = 1 };   // 8 stage-B + 4 stage-A (TX_NJ tiles each) + 2 register loaders + 1 LDS-DMA loader
enum {
  TX_RS2 = 160,                        // bytes per row of the dz(ACT2) image: 80 channels, no pad: pitch 10 = 2 (mod 4) slots is
                                       // conflict-free for the ds_read_b128 lane groups (see TR_RSB in kernels_trunk.h)
  TX_P2 = IMG_ROWS * TX_RS2,           // 5,760
  TX_PM = IMG_ROWS * TR_RSB,           // 10,368: dz(MID) image, 128 channels + 32 pad
  TX_SLOT = 2 * TX_P2,                 // dz2 | dz2 o s_out
  TX_O_SLOT = 0,                       // [3 slots]
  TX_O_DZM = 3 * TX_SLOT,              // [2 bufs][dz(MID), dz(MID) o s_out]
  TX_O_SGN = TX_O_DZM + 2 * 2 * TX_PM, // [3 slots][80 words]
  TX_O_LUT = TX_O_SGN + 3 * 80 * 4,    // 256 x 16 B sign-byte -> XOR mask
  // mask sources, LDS-DMA'd five / six steps ahead into a ring of 8: bit masks [MID > 0] [32 rows][16 B] | [ACT1 > 0] [32][16 B] |
  // 2-bit arg-max codes [32][32 B] (planes written by the forward).  Not part of the zero fill: the DMA may land first.
  TX_PD = 512, TX_PC = 1024,
  TX_DSLOT = 2 * TX_PD + TX_PC,
  TX_O_DMA = TX_O_LUT + 4096,
  TX_O_MLUT = TX_O_DMA + 8 * TX_DSLOT,   // 16 x 8 B: mask nibble -> AND masks of 4 bf16 values
  TX_LDS = TX_O_MLUT + 128
};

struct TrunkDxArgs {
  u16* g_act2;             // [S*B*L][80]  in: dY of ACT2 (from the dense layer's dX); out: masked with [ACT2 > 0] in place
  const u16* act2_hi;      // [S*B*L][80]
  const unsigned char* m_mid;    // [S*B*L][16] bit masks [MID > 0]
  const unsigned char* m_act1;   // [S*B*L][16] bit masks [ACT1 > 0]
  const unsigned char* amax;     // [S*B*L][32] 2-bit arg-max codes
  u16* g_mid;              // [S*B*L][128]  out: dz of block 2's 1x1 outputs (masked)
  u16* g_act1;             // [S*B*L][128]  out: dz of block 1's outputs (masked)
  WeightSlots ws;
  const LayerDesc* layers;
  const uint32_t* sign_in;
  const uint32_t* sign_out;
  long examples;
  int S, B, L, nsplit;
};

// Y is a ReLU output (never negative): Y > 0  <=>  magnitude bits non-zero.  Packed 16-bit min / mul through inline asm:
// hipcc turns the vector-extension form into v_cmp_ne_u16 + v_cndmask pairs (7 instructions per dword instead of 3)
__device__ __forceinline__ uint32_t relu_mask2(uint32_t yy) {
  uint32_t m = yy & 0x7fff7fffu;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(0x00010001u));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(0xffffffffu));
  return m;
}

// PRE: dY of ACT2 arrives already masked with [ACT2 > 0] (dense_ks_bwd_kernel stores it that way): the loaders neither read
// ACT2 nor write the gradient back
template <int EM, bool PRE>
__global__ __launch_bounds__(TX_THREADS) void trunk_dx_kernel(const TrunkDxArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  const int L = A.L;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TX_O_LUT / 4; k += TX_THREADS) z[k] = 0u;
    build_sign_lut((uint4*)(smem + TX_O_LUT), tid, TX_THREADS);
    if (tid < 16)
      *(uint2*)(smem + TX_O_MLUT + tid * 8) = make_uint2(((tid & 1) ? 0xffffu : 0u) | ((tid & 2) ? 0xffff0000u : 0u),
                                                         ((tid & 4) ? 0xffffu : 0u) | ((tid & 8) ? 0xffff0000u : 0u));
  }
  const long Rs = ((long)s * A.B + split) * L, Rstep = (long)A.nsplit * L;
  const int nsteps = nwin + 1;
  const uint4* lut = (const uint4*)(smem + TX_O_LUT);

  if (wave == 14) {
    // =========================== LDS-DMA loader: mask sources ===========================
    // step t issues MID hi of window t+3 and ACT1 hi + codes of window t+2, then waits until only the DMAs of this step
    // and the previous one are in flight: what the stages read at step t+1 (MID hi of t+1, ACT1 hi / codes of t) landed.
    // one instruction per plane: the window's rows are contiguous (L x 16 B masks, L x 32 B codes)
    const uint32_t lds0 = lds_addr(smem) + TX_O_DMA;
    auto dma_rows = [&](const unsigned char* plane, long R0, int rowbytes, uint32_t dst) __attribute__((always_inline)) {
      if constexpr (!(TX_ABL & 4)) {
        if (lane * 16 < L * rowbytes) dma16(plane + R0 * rowbytes + lane * 16, __builtin_amdgcn_readfirstlane(dst));
      }
    };
    auto issue_mid = [&](int k) __attribute__((always_inline)) {
      if (k < nwin) dma_rows(A.m_mid, Rs + k * Rstep, 16, lds0 + (uint32_t)((k & 7) * TX_DSLOT));
    };
    auto issue_a1 = [&](int k) __attribute__((always_inline)) {
      if (k < nwin) {
        dma_rows(A.m_act1, Rs + k * Rstep, 16, lds0 + (uint32_t)((k & 7) * TX_DSLOT + TX_PD));
        dma_rows(A.amax, Rs + k * Rstep, 32, lds0 + (uint32_t)((k & 7) * TX_DSLOT + 2 * TX_PD));
      }
    };
    constexpr int per_mid = 1, per_a1 = 2;
    // step t issues the MID masks of window t+6 and the ACT1 masks / codes of window t+5 and waits until only the DMAs of
    // the last four steps are in flight: what the stages read at step t+1 (MID of t+1, ACT1 / codes of t) was issued at
    // least five steps (>= 10 us) ago - HBM latency under load is 2-3 us
    for (int k = 0; k < 6; ++k) issue_mid(k);
    for (int k = 0; k < 5; ++k) issue_a1(k);
    if constexpr (!(TX_ABL & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int p1 = 0, p2 = 0, p3 = 0;   // instructions issued one / two / three steps ago
    __syncthreads();
    lds_barrier();
    for (int t = 0; t < nsteps; ++t) {
      issue_mid(t + 6);
      issue_a1(t + 5);
      const int cur = (t + 6 < nwin ? per_mid : 0) + (t + 5 < nwin ? per_a1 : 0);
      if constexpr (!(TX_ABL & 4)) BNN_WAIT_VMCNT_WIDE(cur + p1 + p2 + p3);
      p3 = p2; p2 = p1; p1 = cur;
      lds_barrier();
    }
    return;
  }
  if (wave >= 12) {
    // =========================== loaders ===========================
    // wave 12 + p stages the windows k = p (mod 2): dz(ACT2) (+ its s_out copy) and the window's sign words
    const int p = wave - 12;
    const int n2 = L * 10;   // 16-byte chunks of an 80-channel bf16 plane
    const uint32_t* sg0 = nullptr;
    const uint32_t* sg1 = nullptr;
    long sst0 = 0, sst1 = 0;
    if constexpr (FO) {
      auto setup = [&](int layer, int kk, const uint32_t*& q, long& stride) {
        const LayerDesc ly = A.layers[layer];
        if (kk < 4 && kk < ly.sign_in_words) {
          q = A.sign_in + ly.sign_in_off * A.examples + ((long)s * A.B + split) * ly.sign_in_words + kk;
          stride = (long)A.nsplit * ly.sign_in_words;
        } else if (kk >= 4 && kk - 4 < ly.sign_out_words && kk < 6) {
          q = A.sign_out + ly.sign_out_off * A.examples + ((long)s * A.B + split) * ly.sign_out_words + (kk - 4);
          stride = (long)A.nsplit * ly.sign_out_words;
        }
      };
      setup(lane >> 3, lane & 7, sg0, sst0);
      if (lane < 16) setup(8 + (lane >> 3), lane & 7, sg1, sst1);
    }
    uint4 g0, g1, g2, g3, g4_;
    uint4 y0 = make_uint4(0, 0, 0, 0), y1 = y0, y2 = y0, y3 = y0, y4 = y0;
    uint32_t sb0 = 0, sb1 = 0;
    int qo[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int q = j * 64 + lane;
      qo[j] = (q < n2 ? q : 0) * 16;
    }
#define TX_FETCH(K)                                                            \
  do {                                                                         \
    const char* gp = (const char*)A.g_act2 + (Rs + (K) * Rstep) * 160;         \
    const char* yp = (const char*)A.act2_hi + (Rs + (K) * Rstep) * 160;        \
    g0 = *(const uint4*)(gp + qo[0]); g1 = *(const uint4*)(gp + qo[1]);        \
    g2 = *(const uint4*)(gp + qo[2]); g3 = *(const uint4*)(gp + qo[3]);        \
    g4_ = *(const uint4*)(gp + qo[4]);                                         \
    if constexpr (!PRE) {                                                      \
      y0 = *(const uint4*)(yp + qo[0]); y1 = *(const uint4*)(yp + qo[1]);      \
      y2 = *(const uint4*)(yp + qo[2]); y3 = *(const uint4*)(yp + qo[3]);      \
      y4 = *(const uint4*)(yp + qo[4]);                                        \
    }                                                                          \
    if constexpr (FO) {                                                        \
      if (sg0) sb0 = sg0[(long)(K) * sst0];                                    \
      if (sg1) sb1 = sg1[(long)(K) * sst1];                                    \
    }                                                                          \
  } while (0)
    auto put1 = [&](char* sl, char* gdst, int j, uint4 g, const uint4 y) __attribute__((always_inline)) {
      const int q = j * 64 + lane, qq = q < n2 ? q : 0;
      const int row = qq / 10, c = qq - row * 10;
      uint4 fm = make_uint4(0, 0, 0, 0);
      if constexpr (FO) {
        // s_out of the chunk's 8 couts: channels 0-15 layer 4, 16-31 layer 6, 32-47 layer 8, 48-79 layer 9; the word
        // sits in the registers of the lane that loaded it (all lanes take part in the shuffle)
        const int layer = c < 2 ? 4 : (c < 4 ? 6 : (c < 6 ? 8 : 9));
        const int sh = c < 6 ? (c & 1) * 8 : (c - 6) * 8;
        const int widx = layer * 8 + 4;
        const uint32_t w0 = __shfl(sb0, widx & 63, 64), w1 = __shfl(sb1, widx & 63, 64);
        fm = lut[((widx < 64 ? w0 : w1) >> sh) & 0xffu];
      }
      if (q < n2) {
        if constexpr (!PRE) { g.x &= relu_mask2(y.x); g.y &= relu_mask2(y.y); g.z &= relu_mask2(y.z); g.w &= relu_mask2(y.w); }
        const int o = (row + HALO) * TX_RS2 + c * 16;
        *(uint4*)(sl + o) = g;
        if constexpr (!(TX_ABL & 1) && !PRE) *(uint4*)(gdst + q * 16) = g;   // the dW kernels read dz(ACT2) from HBM: masked once, here
        if constexpr (FO) *(uint4*)(sl + TX_P2 + o) = make_uint4(g.x ^ fm.x, g.y ^ fm.y, g.z ^ fm.z, g.w ^ fm.w);
      }
    };
#define TX_PUT(K)                                                              \
  do {                                                                         \
    char* sl = smem + TX_O_SLOT + ((K) % 3) * TX_SLOT;                         \
    if constexpr (FO) {                                                        \
      uint32_t* sgw = (uint32_t*)(smem + TX_O_SGN) + ((K) % 3) * 80;           \
      sgw[lane] = sb0;                                                         \
      if (lane < 16) sgw[64 + lane] = sb1;                                     \
    }                                                                          \
    char* gd = (char*)A.g_act2 + (Rs + (K) * Rstep) * 160;                     \
    put1(sl, gd, 0, g0, y0); put1(sl, gd, 1, g1, y1); put1(sl, gd, 2, g2, y2); \
    put1(sl, gd, 3, g3, y3); put1(sl, gd, 4, g4_, y4);                         \
  } while (0)
    // window k is fetched at the start of step k-3 (or in the prologue) and put during step k-1
    if (p < nwin) TX_FETCH(p);       // windows 0 / 1
    __syncthreads();                 // zero fill + table
    if (p == 0 && nwin > 0) TX_PUT(0);
    if (p == 0 && 2 < nwin) TX_FETCH(2);
    lds_barrier();                   // window 0 staged
    for (int t = 0; t < nsteps; ++t) {
      const int k = t + 1;           // window to stage during this step
      if ((k & 1) == p && !(TX_ABL & 8)) {
        if (k < nwin) TX_PUT(k);
        if (k + 2 < nwin) TX_FETCH(k + 2);
      }
      lds_barrier();
    }
#undef TX_FETCH
#undef TX_PUT
    return;
  }

  const int i16 = lane & 15, g4 = lane >> 4;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  if (wave >= 8) {
    // =========================== stage A: dz(MID) tiles j (layer 6) and 4 + j (layer 8), j = TX_NJ (wave - 8) + jj ===========================
    const int jb = (wave - 8) * TX_NJ;
    const LayerDesc l6 = A.layers[6], l8 = A.layers[8];
    const long sa = A.ws.slott_stride_a * s, sb = A.ws.slott_stride_b * s;
    bf16x8 a6[TX_NJ][2], b6[TX_NJ][2], a8[TX_NJ][3], b8[TX_NJ][3];
#pragma unroll
    for (int jj = 0; jj < TX_NJ; ++jj) {
      const long r6 = (long)((jb + jj) * 16 + i16) * l6.KPt + l6.wt_off + g4 * 8;
      const long r8 = (long)((jb + jj) * 16 + i16) * l8.KPt + l8.wt_off + g4 * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        a6[jj][ks] = *(const bf16x8*)((const u16*)A.ws.at + sa + r6 + ks * 32);
        b6[jj][ks] = FO ? *(const bf16x8*)((const u16*)A.ws.bt + sb + r6 + ks * 32) : a6[jj][ks];
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        a8[jj][ks] = *(const bf16x8*)((const u16*)A.ws.at + sa + r8 + ks * 32);
        b8[jj][ks] = FO ? *(const bf16x8*)((const u16*)A.ws.bt + sb + r8 + ks * 32) : a8[jj][ks];
      }
    }
    // B fragment of k-step ks: K index gg = ks*4 + g4 -> (flipped) tap gg >> 1, 8-cout chunk gg & 1
    const int lane_b = i16 * TX_RS2 + (g4 >> 1) * TX_RS2 + (g4 & 1) * 16;
    __syncthreads();
    lds_barrier();
    for (int t = 0; t < nsteps; ++t) {
      const int k = t;
      if (k < nwin && !(TX_ABL & 64)) {
        const char* sl = smem + TX_O_SLOT + (k % 3) * TX_SLOT;
        const uint32_t* sg = (const uint32_t*)(smem + TX_O_SGN) + (k % 3) * 80;
        const unsigned R0 = (unsigned)(Rs + k * Rstep);
        char* dzm = smem + TX_O_DZM + (k & 1) * 2 * TX_PM;
        const char* dsl = smem + TX_O_DMA + (k & 7) * TX_DSLOT;   // MID hi of this window (LDS-DMA'd)
#pragma unroll
        for (int jj = 0; jj < TX_NJ; ++jj) {
          const int j = jb + jj;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            // half 0: layer 6 (k3, pad 1, dz channels 16..31, MID channels 0..63); half 1: layer 8 (k5, pad 2, 32..47, 64..127)
            const int NKS = half == 0 ? 2 : 3, PAD = half == 0 ? 1 : 2, TAPS = half == 0 ? 3 : 5, CH0 = half == 0 ? 2 : 4;
            const int ly = half == 0 ? 6 : 8;
            __builtin_amdgcn_sched_barrier(0);   // one tile at a time (the scheduler would hoist every tile's operand reads)
            f32x4 acc[2];
            acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            uint32_t rsgn = 0;
            if constexpr (FO) {
              const int ci = j * 16 + i16;   // input channel of the layer = row of the transposed fragment
              rsgn = ((sg[ly * 8 + (ci >> 5)] >> (ci & 31)) & 1u) ? 0x80008000u : 0u;
            }
            // all operand reads of the tile first, then its MFMAs (one LDS latency per tile)
            bf16x8 bz[3][2], bs[FO ? 3 : 1][2];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
              if (ks >= NKS) break;
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                // taps past the last one carry zero weights: keep their rows inside the image
                const int off = (mt * 16 + ks * 2 - PAD + HALO) * TX_RS2 + CH0 * 16;
                const int tap_hi = ks * 2 + 1;   // the tap of lanes g4 >= 2
                const char* bp = sl + lane_b + off;
                if (tap_hi >= TAPS) bp -= (g4 >> 1) * TX_RS2;
                bz[ks][mt] = *(const bf16x8*)bp;
                if constexpr (FO) bs[ks][mt] = *(const bf16x8*)(bp + TX_P2);
              }
            }
            // the ReLU mask of the tile's outputs (mask byte -> AND masks: two dependent LDS reads) is fetched with the
            // operands, not after the MFMAs: this role is one wave per SIMD and its per-tile latency chain is the step
            const int och = (half * 4 + j) * 16 + 4 * g4;   // MID channel
            uint2 mk[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const uint32_t mb = *(const unsigned char*)(dsl + (mt * 16 + i16) * 16 + (och >> 3));
              mk[mt] = *(const uint2*)(smem + TX_O_MLUT + ((mb >> (och & 4)) & 15u) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);   // reads first, then the MFMAs
            f32x4 accb[2];   // Flipout: the perturbation path accumulates on its own (two dependent chains per m-tile, not one)
            accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
              if (ks >= NKS) break;
              const bf16x8 wa = half == 0 ? a6[jj][ks < 2 ? ks : 0] : a8[jj][ks];
              bf16x8 wb = half == 0 ? b6[jj][ks < 2 ? ks : 0] : b8[jj][ks];
              if constexpr (FO) wb = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, wb) ^ u32x4{rsgn, rsgn, rsgn, rsgn});
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bz[ks][mt], acc[mt], 0, 0, 0);
                if constexpr (FO) accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, bs[ks][mt], accb[mt], 0, 0, 0);
              }
            }
            if constexpr (FO) {
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][r] += accb[mt][r];
            }
            // epilogue: mask with [MID > 0], dz -> LDS (+ s_out copy) and HBM
            const int oly = half == 0 ? 5 : 7;              // the 1x1 layer that produced these MID channels
            uint32_t so = 0;
            if constexpr (FO) {
              const int b0 = och & 63;
              so = (sg[oly * 8 + 4 + (b0 >> 5)] >> (b0 & 31)) & 0xfu;
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const int row = mt * 16 + i16;
              if (row < L) {
                uint2 d = make_uint2(cvt_pk(acc[mt][0], acc[mt][1]), cvt_pk(acc[mt][2], acc[mt][3]));
                d.x &= mk[mt].x;
                d.y &= mk[mt].y;
                const int o = (row + HALO) * TR_RSB + och * 2;
                *(uint2*)(dzm + o) = d;
                if constexpr (FO) {
                  const uint32_t m0 = ((so & 1u) << 15) | ((so & 2u) << 30), m1 = ((so & 4u) << 13) | ((so & 8u) << 28);
                  *(uint2*)(dzm + TX_PM + o) = make_uint2(d.x ^ m0, d.y ^ m1);
                }
                if constexpr (!(TX_ABL & 1)) *(uint2*)((char*)A.g_mid + ((R0 + (unsigned)row) * 256u + (unsigned)(och * 2))) = d;
              }
            }
          }
        }
      }
      lds_barrier();
    }
    return;
  }

  // =========================== stage B: dz(ACT1) tile ct ===========================
  {
    const int ct = wave;
    const long sa = A.ws.slott_stride_a * s, sb = A.ws.slott_stride_b * s;
    // k-steps: 0 layer 4 (dz2 chunks 0..3), 1-2 layer 5 (dzm chunks 0..7), 3-4 layer 7 (dzm chunks 8..15), 5 layer 9 (pooled; dz2 chunks 6..9)
    bf16x8 wa[6], wb[6];
    {
      const int lys[6] = {4, 5, 5, 7, 7, 9};
      const int kss[6] = {0, 0, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const LayerDesc ly = A.layers[lys[q]];
        const long r = (long)(ct * 16 + i16) * ly.KPt + ly.wt_off + kss[q] * 32 + g4 * 8;
        wa[q] = *(const bf16x8*)((const u16*)A.ws.at + sa + r);
        wb[q] = FO ? *(const bf16x8*)((const u16*)A.ws.bt + sb + r) : wa[q];
      }
    }
    const int och = ct * 16 + 4 * g4;
    const int ci = ct * 16 + i16;
    __syncthreads();
    lds_barrier();
    for (int t = 0; t < nsteps; ++t) {
      const int k = t - 1;
      if (k >= 0 && k < nwin) {
        const char* sl = smem + TX_O_SLOT + (k % 3) * TX_SLOT;
        const uint32_t* sg = (const uint32_t*)(smem + TX_O_SGN) + (k % 3) * 80;
        const char* dzm = smem + TX_O_DZM + (k & 1) * 2 * TX_PM;
        const char* dsl = smem + TX_O_DMA + (k & 7) * TX_DSLOT;   // ACT1 hi + codes of this window (LDS-DMA'd)
        const unsigned R0 = (unsigned)(Rs + k * Rstep);
        f32x4 acc[2], accp[2];
        acc[0] = acc[1] = accp[0] = accp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* b2 = sl + (i16 + HALO) * TX_RS2 + g4 * 16;
        const char* bm = dzm + (i16 + HALO) * TR_RSB + g4 * 16;
        constexpr int lys[6] = {4, 5, 5, 7, 7, 9};
        // all operand reads of an m-tile are issued before its MFMAs (one LDS latency per m-tile, not one per k-step)
        bf16x8 w2[6];
        __builtin_amdgcn_sched_barrier(0);
        // the ReLU masks and pool codes of the tile's outputs are fetched here, ahead of the MFMAs (dependent LDS reads)
        uint2 mk[2];
        uint32_t code[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + i16;
          const uint32_t mb = *(const unsigned char*)(dsl + TX_PD + row * 16 + (och >> 3));
          mk[mt] = *(const uint2*)(smem + TX_O_MLUT + ((mb >> (och & 4)) & 15u) * 8);
          code[mt] = row < L ? *(const unsigned char*)(dsl + 2 * TX_PD + row * 32 + (och >> 2)) : 0x55u;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          w2[q] = wb[q];
          if constexpr (FO) {
            const uint32_t rsgn = ((sg[lys[q] * 8 + (ci >> 5)] >> (ci & 31)) & 1u) ? 0x80008000u : 0u;
            w2[q] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, wb[q]) ^ u32x4{rsgn, rsgn, rsgn, rsgn});
          }
        }
#pragma unroll
        for (int mt = 0; mt < ((TX_ABL & 16) ? 0 : 2); ++mt) {
#pragma unroll
          for (int qb = 0; qb < 6; qb += 3) {   // three k-steps per batch: 6 fragments in flight
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 bz[3], bs[FO ? 3 : 1];
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) {
              const int q = qb + qq;
              const char* bp;
              int second;
              if (q == 0) { bp = b2 + mt * 16 * TX_RS2; second = TX_P2; }
              else if (q == 5) { bp = b2 + mt * 16 * TX_RS2 + 6 * 16; second = TX_P2; }
              else { bp = bm + mt * 16 * TR_RSB + (q - 1) * 64; second = TX_PM; }
              bz[qq] = *(const bf16x8*)bp;
              if constexpr (FO) bs[qq] = *(const bf16x8*)(bp + second);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads together: hipcc otherwise sinks each one next to its MFMA
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) {
              const int q = qb + qq;
              f32x4& ta = q == 5 ? accp[mt] : acc[mt];
              ta = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q], bz[qq], ta, 0, 0, 0);
              if constexpr (FO) ta = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[q], bs[qq], ta, 0, 0, 0);
            }
          }
        }
        if constexpr (TX_ABL & 32) {
          asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(accp[0]), "v"(accp[1]));
        } else {
        // pooled branch: this row's gradient goes to row + code - 1 (codes of the forward's MaxPool1d(3,1,1))
        f32x4 v[2], up[2], dn[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + i16;
          v[mt] = acc[mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t cd = (code[mt] >> (2 * r)) & 3u;
            const float g = row < L ? accp[mt][r] : 0.f;
            v[mt][r] += cd == 1u ? g : 0.f;
            up[mt][r] = cd == 0u ? g : 0.f;
            dn[mt][r] = cd == 2u ? g : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // row r receives `dn` of row r-1 and `up` of row r+1 (DPP rotations inside the 16-lane row; m-tile seam 15 | 16)
          const float a0 = rot16<0x121>(dn[0][r]), a1 = rot16<0x121>(dn[1][r]);
          const float c0 = rot16<0x12F>(up[0][r]), c1 = rot16<0x12F>(up[1][r]);
          v[0][r] += (i16 == 0 ? 0.f : a0) + (i16 == 15 ? c1 : c0);
          v[1][r] += (i16 == 0 ? a0 : a1) + (i16 == 15 ? 0.f : c1);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + i16;
          if (row < L) {
            uint2 d = make_uint2(cvt_pk(v[mt][0], v[mt][1]), cvt_pk(v[mt][2], v[mt][3]));
            d.x &= mk[mt].x;
            d.y &= mk[mt].y;
            if constexpr (!(TX_ABL & 1)) *(uint2*)((char*)A.g_act1 + ((R0 + (unsigned)row) * 256u + (unsigned)(och * 2))) = d;
            else asm volatile("" ::"v"(d.x), "v"(d.y));
          }
        }
        }
      }
      lds_barrier();
    }
  }
}
