// Weight gradients of the convolutional trunk of the Inception net (bf16 planes, gfx950).
//
// dW[n][tap][c] += sum_rows dz[row][n] * X[row + tap - pad][c]: the contraction index is the window row, so both
// MFMA operands are read TRANSPOSED from [row][channel] LDS images (ds_read_b64_tr_b16).  Images use XOR layouts
// that make every transposed read bank-conflict free for every tap shift (cdna_hip_programming.md T10):
//   128-channel image : byte(row, chunk16) = 256 row + 16 (chunk16 ^ f(row)),  f = ((row & 3) << 2) | ((row >> 2) & 3)
//   32-channel image  : byte(row, chunk16) =  64 row + 16 (chunk16 ^ ((row >> 2) & 3))
// A workgroup owns a particle and a strided set of its windows; tiles stay in registers across the windows and are
// added to the per-particle fp32 gradient images with atomics at the end.  Flipout's second product
// (dz o s_out)^T (x o s_in) flips the sign bit of the transposed fragments per lane (a channel per lane).
// Two loader waves alternate windows (global -> registers -> LDS, two steps ahead); one barrier per window.
#pragma once
#include "kernels_trunk_bwd.h"

__device__ __forceinline__ int f128(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

enum { TW1_NC = 8, TW1_NW = 10, TW1_THREADS = TW1_NW * 64 };
enum {
  TW_PZ = 32 * 256,                 // dz image: rows 0..31 (rows >= L stay zero), 128 channels
  TW_PX = IMG_ROWS * 64,            // x image: rows -2..33, 32 channels
  TW1_SLOT = TW_PZ + 2 * TW_PX,     // dz(ACT1) | x hi | pooled x hi
  TW1_O_SGN = 3 * TW1_SLOT,         // [3 slots][4 layers][2 words]: sign_in word, sign_out word
  TW1_LDS = TW1_O_SGN + 3 * 8 * 4
};

struct TrunkDw1Args {
  const u16* x_hi;         // [B*L][32]
  const u16* xp_hi;        // [B*L][32] pooled
  const u16* g_act1;       // [S*B*L][128] dz of block 1 (masked)
  const LayerDesc* layers;
  const uint32_t* sign_in;
  const uint32_t* sign_out;
  long examples;
  float* gw_a; float* gw_b; float* gb_a;
  long gw_stride; int gb_stride;
  int S, B, L, nsplit;
};

// transposed 16 x 32 fragment: element offsets of the two 4-row blocks this lane addresses
__device__ __forceinline__ bf16x8 tr_frag2(const char* p0, const char* p1) {
  return tr_frag((const u16*)p0, (const u16*)p1);
}

// one (layer, n-tile) job over the c-tiles [CT0, CT0 + NCT) of the 32-channel input
template <int EM, int LAYER, int NT, int CT0, int NCT, bool BIAS>
struct Dw1Job {
  static constexpr bool FO = (EM == EM_FLIPOUT);
  static constexpr int TAPS = tl_taps(LAYER), PAD = (TAPS - 1) / 2;
  f32x4 acc_a[TAPS][NCT], acc_b[FO ? TAPS : 1][FO ? NCT : 1], acc_bias;

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        acc_a[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (FO) acc_b[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    acc_bias = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  __device__ __forceinline__ void run(const char* sl, const uint32_t* sg, int lane, bf16x8 ones) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = 8 * g + q;
    // A: dz channels LAYER*32 + NT*16 .. +15, rows r0 / r0 + 4
    constexpr int n0 = LAYER * 32 + NT * 16;
    const int ca = (n0 >> 3) + (p >> 1);
    const char* a0 = sl + r0 * 256 + ((ca ^ f128(r0)) << 4) + 8 * (p & 1);
    const char* a1 = sl + (r0 + 4) * 256 + ((ca ^ f128(r0 + 4)) << 4) + 8 * (p & 1);
    const bf16x8 fa = tr_frag2(a0, a1);
    bf16x8 fas = fa;
    if constexpr (FO) {
      const bool no = (sg[LAYER * 2 + 1] >> (NT * 16 + (lane & 15))) & 1u;
      fas = xor_sign(fa, no);
    }
    if constexpr (BIAS) acc_bias = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, acc_bias, 0, 0, 0);
    const char* xi = sl + TW_PZ + (tl_pool(LAYER) ? TW_PX : 0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int ra = r0 + t - PAD + HALO, rb = ra + 4;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int cb = (CT0 + c) * 2 + (p >> 1);
        const char* b0 = xi + ra * 64 + ((cb ^ ((ra >> 2) & 3)) << 4) + 8 * (p & 1);
        const char* b1 = xi + rb * 64 + ((cb ^ ((rb >> 2) & 3)) << 4) + 8 * (p & 1);
        const bf16x8 fb = tr_frag2(b0, b1);
        acc_a[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc_a[t][c], 0, 0, 0);
        if constexpr (FO) {
          const bool ni = (sg[LAYER * 2] >> ((CT0 + c) * 16 + (lane & 15))) & 1u;
          acc_b[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fas, xor_sign(fb, ni), acc_b[t][c], 0, 0, 0);
        }
      }
    }
  }

  __device__ __forceinline__ void flush(const TrunkDw1Args& A, int s, int lane) const {
    const LayerDesc ly = A.layers[LAYER];
    const int i4 = 4 * (lane >> 4), jc = lane & 15;
    float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
    float* gwb = A.gw_b + A.gw_stride * s + ly.w_off;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int ch = (CT0 + c) * 16 + jc;
        if (ch >= ly.cin) continue;   // channel pads of the x image
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = NT * 16 + i4 + r;
          if (n >= ly.cout) continue;
          const long o = (long)n * ly.KP + (long)t * ly.cin_img + ch;
          atomicAdd(gwa + o, acc_a[t][c][r]);
          if constexpr (FO) atomicAdd(gwb + o, acc_b[t][c][r]);
        }
      }
    if constexpr (BIAS) {
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = NT * 16 + i4 + r;
          if (n < ly.cout) atomicAdd(A.gb_a + (long)A.gb_stride * s + ly.bias_off + n, acc_bias[r]);
        }
      }
    }
  }
};

template <int EM, class J0, class J1>
__device__ __forceinline__ void dw1_role(const TrunkDw1Args& A, char* smem, int s, int nwin, int lane) {
  J0 j0;
  J1 j1;
  j0.init();
  j1.init();
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  __syncthreads();
  lds_barrier();
  for (int k = 0; k < nwin; ++k) {
    const char* sl = smem + (k % 3) * TW1_SLOT;
    const uint32_t* sg = (const uint32_t*)(smem + TW1_O_SGN) + (k % 3) * 8;
    j0.run(sl, sg, lane, ones);
    j1.run(sl, sg, lane, ones);
    lds_barrier();
  }
  j0.flush(A, s, lane);
  j1.flush(A, s, lane);
}

template <int EM>
struct Dw1Empty {
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void run(const char*, const uint32_t*, int, bf16x8) {}
  __device__ __forceinline__ void flush(const TrunkDw1Args&, int, int) const {}
};

template <int EM>
__global__ __launch_bounds__(TW1_THREADS) void trunk_dw1_kernel(const TrunkDw1Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  const int L = A.L;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TW1_LDS / 4; k += TW1_THREADS) z[k] = 0u;
  }
  if (wave >= TW1_NC) {
    // =========================== loaders: wave 8 + p stages the windows k = p (mod 2) ===========================
    const int p = wave - TW1_NC;
    const int nz = L * 16, nx = L * 4;
    // dz chunks: q = j*64 + lane over [row][16 chunks]; x / pooled x chunks: lanes 0..119 each (two instructions)
    int zo[8], zd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = j * 64 + lane, qq = q < nz ? q : 0;
      const int row = qq >> 4, c = qq & 15;
      zo[j] = qq * 16;
      zd[j] = row * 256 + ((c ^ f128(row)) << 4);
    }
    int xo[2], xd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = j * 64 + lane, qq = q < nx ? q : 0;
      const int row = qq >> 2, c = qq & 3, ri = row + HALO;
      xo[j] = qq * 16;
      xd[j] = TW_PZ + ri * 64 + ((c ^ ((ri >> 2) & 3)) << 4);
    }
    const uint32_t* sgp = nullptr;
    long sgs = 0;
    if (FO && lane < 8) {   // lane -> (layer = lane >> 1, in / out)
      const LayerDesc ly = A.layers[lane >> 1];
      if (lane & 1) {
        sgp = A.sign_out + ly.sign_out_off * A.examples + ((long)s * A.B + split) * ly.sign_out_words;
        sgs = (long)A.nsplit * ly.sign_out_words;
      } else {
        sgp = A.sign_in + ly.sign_in_off * A.examples + ((long)s * A.B + split) * ly.sign_in_words;
        sgs = (long)A.nsplit * ly.sign_in_words;
      }
    }
    uint4 z0, z1, z2, z3, z4, z5, z6, z7, x0, x1, y0, y1;
    uint32_t sw = 0;
#define TW1_FETCH(K)                                                                                       \
  do {                                                                                                     \
    const char* zp = (const char*)A.g_act1 + ((long)s * A.B + split + (long)(K) * A.nsplit) * L * 256;     \
    const long xw = (split + (long)(K) * A.nsplit) * L * 64;                                               \
    z0 = *(const uint4*)(zp + zo[0]); z1 = *(const uint4*)(zp + zo[1]);                                    \
    z2 = *(const uint4*)(zp + zo[2]); z3 = *(const uint4*)(zp + zo[3]);                                    \
    z4 = *(const uint4*)(zp + zo[4]); z5 = *(const uint4*)(zp + zo[5]);                                    \
    z6 = *(const uint4*)(zp + zo[6]); z7 = *(const uint4*)(zp + zo[7]);                                    \
    x0 = *(const uint4*)((const char*)A.x_hi + xw + xo[0]); x1 = *(const uint4*)((const char*)A.x_hi + xw + xo[1]);     \
    y0 = *(const uint4*)((const char*)A.xp_hi + xw + xo[0]); y1 = *(const uint4*)((const char*)A.xp_hi + xw + xo[1]);   \
    if constexpr (FO) { if (sgp) sw = sgp[(long)(K) * sgs]; }                                              \
  } while (0)
#define TW1_PUT(K)                                                                                         \
  do {                                                                                                     \
    char* sl = smem + ((K) % 3) * TW1_SLOT;                                                                \
    *(uint4*)(sl + zd[0]) = z0; *(uint4*)(sl + zd[1]) = z1; *(uint4*)(sl + zd[2]) = z2;                    \
    *(uint4*)(sl + zd[3]) = z3; *(uint4*)(sl + zd[4]) = z4; *(uint4*)(sl + zd[5]) = z5;                    \
    *(uint4*)(sl + zd[6]) = z6;                                                                            \
    if (7 * 64 + lane < nz) *(uint4*)(sl + zd[7]) = z7;                                                    \
    *(uint4*)(sl + xd[0]) = x0; *(uint4*)(sl + TW_PX + xd[0]) = y0;                                        \
    if (64 + lane < nx) { *(uint4*)(sl + xd[1]) = x1; *(uint4*)(sl + TW_PX + xd[1]) = y1; }                \
    if constexpr (FO) { if (lane < 8) ((uint32_t*)(smem + TW1_O_SGN))[((K) % 3) * 8 + lane] = sw; }        \
  } while (0)
    // window k: fetched at step k-3 (or in the prologue), put during step k-1; step k computes window k
    if (p < nwin) TW1_FETCH(p);
    __syncthreads();
    if (p == 0 && nwin > 0) TW1_PUT(0);
    if (p == 0 && 2 < nwin) TW1_FETCH(2);
    lds_barrier();
    for (int t = 0; t < nwin; ++t) {
      const int k = t + 1;
      if ((k & 1) == p) {
        if (k < nwin) TW1_PUT(k);
        if (k + 2 < nwin) TW1_FETCH(k + 2);
      }
      lds_barrier();
    }
#undef TW1_FETCH
#undef TW1_PUT
    return;
  }
  // =========================== compute: (layer, n-tile[, c-tile]) jobs ===========================
  switch (wave) {
    case 0: dw1_role<EM, Dw1Job<EM, 2, 0, 0, 1, true>, Dw1Job<EM, 0, 0, 0, 2, true>>(A, smem, s, nwin, lane); break;
    case 1: dw1_role<EM, Dw1Job<EM, 2, 0, 1, 1, false>, Dw1Job<EM, 0, 1, 0, 2, true>>(A, smem, s, nwin, lane); break;
    case 2: dw1_role<EM, Dw1Job<EM, 2, 1, 0, 1, true>, Dw1Empty<EM>>(A, smem, s, nwin, lane); break;
    case 3: dw1_role<EM, Dw1Job<EM, 2, 1, 1, 1, false>, Dw1Empty<EM>>(A, smem, s, nwin, lane); break;
    case 4: dw1_role<EM, Dw1Job<EM, 1, 0, 0, 2, true>, Dw1Empty<EM>>(A, smem, s, nwin, lane); break;
    case 5: dw1_role<EM, Dw1Job<EM, 1, 1, 0, 2, true>, Dw1Empty<EM>>(A, smem, s, nwin, lane); break;
    case 6: dw1_role<EM, Dw1Job<EM, 3, 0, 0, 2, true>, Dw1Empty<EM>>(A, smem, s, nwin, lane); break;
    default: dw1_role<EM, Dw1Job<EM, 3, 1, 0, 2, true>, Dw1Empty<EM>>(A, smem, s, nwin, lane); break;
  }
}
