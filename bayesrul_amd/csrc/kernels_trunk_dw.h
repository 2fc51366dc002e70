// Weight gradients of the convolutional trunk of the Inception net (bf16 planes, gfx950).
//
// dW[n][tap][c] += sum_rows dz[row][n] * X[row + tap - pad][c]: the contraction index is the window row, so both
// MFMA operands are read TRANSPOSED from [row][channel] LDS images (ds_read_b64_tr_b16).  Images use XOR layouts
// that make every transposed read bank-conflict free for every tap shift (cdna_hip_programming.md T10):
//   128-channel image : byte(row, chunk16) = 256 row + 16 (chunk16 ^ f(row)),  f = ((row & 3) << 2) | ((row >> 2) & 3)
//   32-channel image  : byte(row, chunk16) =  64 row + 16 (chunk16 ^ ((row >> 2) & 3))
// A workgroup owns a particle and a strided set of its windows; tiles stay in registers across the windows and are
// written ONCE, with plain stores, into the workgroup's own partial image ("slab", forward image layout); the chain
// rule kernel sums the slabs of a particle (grad_finalize_kernel).  No atomics: the sums are bitwise reproducible, and
// plain stores run at ~4x the chip-wide float-atomic rate (MI355X_MICROARCH.md, Global float atomics).  Flipout's second product
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
  float* gw_a; float* gw_b; float* gb_a;   // slabs of this kernel: [S * nsplit][gw_stride] / [S * nsplit][gb_stride]
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
        // operands swapped: the tile is accumulated TRANSPOSED (rows = input channels, columns = couts), so a lane ends up
        // with 4 consecutive input channels of one cout = one 16-byte store in flush()
        acc_a[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc_a[t][c], 0, 0, 0);
        if constexpr (FO) {
          const bool ni = (sg[LAYER * 2] >> ((CT0 + c) * 16 + (lane & 15))) & 1u;
          acc_b[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xor_sign(fb, ni), fas, acc_b[t][c], 0, 0, 0);
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
        // transposed tile: this lane holds input channels ch .. ch+3 of cout n (channel pads of the x image are zero
        // columns: their sums are zeros)
        const int ch = (CT0 + c) * 16 + i4, n = NT * 16 + jc;
        if (n >= ly.cout) continue;
        const long o = (long)n * ly.KP + (long)t * ly.cin_img + ch;
        *(f32x4*)(gwa + o) = acc_a[t][c];
        if constexpr (FO) *(f32x4*)(gwb + o) = acc_b[t][c];
      }
    if constexpr (BIAS) {
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = NT * 16 + i4 + r;
          if (n < ly.cout) A.gb_a[(long)A.gb_stride * s + ly.bias_off + n] = acc_bias[r];
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
  j0.flush(A, (int)blockIdx.x, lane);   // slab = this workgroup's (particle, split)
  j1.flush(A, (int)blockIdx.x, lane);
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

// ==========================================================================================
// Block 2: trunk_dw2a_kernel (1x1 level: layers 4, 5, 7 and the pooled layer 9; X = ACT1 hi / its max-pool) and
// trunk_dw2b_kernel (k3 / k5 level: layers 6, 8; X = MID hi).  dz comes pre-masked: dz(MID) from trunk_dx, dz(ACT2)
// masked in place by trunk_dx's loader.
// ==========================================================================================
enum {
  TW_PH = IMG_ROWS * 256,           // X image with halo rows: 36 x 128 channels
  TW2A_SLOT = 2 * TW_PH + 2 * TW_PZ,   // ACT1 hi | pooled ACT1 hi | dz(MID) | dz(ACT2) (80 channels in a 128-channel image)
  TW2A_O_SGN = 3 * TW2A_SLOT,          // [3 slots][6 layers 4..9][8 words]
  TW2A_LDS = TW2A_O_SGN + 3 * 48 * 4,
  TW2A_NC = 11, TW2A_NW = 15, TW2A_THREADS = TW2A_NW * 64,
  TW2B_SLOT = TW_PH + TW_PZ,           // MID hi | dz(ACT2)
  TW2B_O_SGN = 3 * TW2B_SLOT,
  TW2B_LDS = TW2B_O_SGN + 3 * 48 * 4,
  TW2B_NC = 4, TW2B_NW = 6, TW2B_THREADS = TW2B_NW * 64
};

struct TrunkDw2Args {
  const u16* x_hi;         // [S*B*L][128]: ACT1 hi (a) or MID hi (b)
  const u16* g_mid;        // [S*B*L][128] dz(MID) (a)
  const u16* g_act2;       // [S*B*L][80]  dz(ACT2)
  const LayerDesc* layers;
  const uint32_t* sign_in;
  const uint32_t* sign_out;
  long examples;
  float* gw_a; float* gw_b; float* gb_a;   // slabs of this kernel: [S * nsplit][gw_stride] / [S * nsplit][gb_stride]
  long gw_stride; int gb_stride;
  int S, B, L, nsplit;
};

// per-layer image geometry inside a slot (byte offset of the image, first channel)
template <bool KA> __host__ __device__ constexpr int dw2_zoff(int l) { return KA ? ((l == 5 || l == 7) ? 2 * TW_PH : 2 * TW_PH + TW_PZ) : TW_PH; }
__host__ __device__ constexpr int dw2_zch(int l) { return l == 7 ? 64 : (l == 9 ? 48 : (l == 6 ? 16 : (l == 8 ? 32 : 0))); }
template <bool KA> __host__ __device__ constexpr int dw2_xoff(int l) { return (KA && l == 9) ? TW_PH : 0; }
__host__ __device__ constexpr int dw2_xch(int l) { return l == 8 ? 64 : 0; }

// one (layer, n-tile) job over the c-tiles [CT0, CT0 + NCT) of the layer's input channels, all taps
template <int EM, bool KA, int LAYER, int NT, int CT0, int NCT, bool BIAS>
struct Dw2Job {
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

  __device__ __forceinline__ void run(const char* sl, const uint32_t* sgs, int lane, bf16x8 ones) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = 8 * g + q;
    const uint32_t* sg = sgs + (LAYER - 4) * 8;
    constexpr int n0 = dw2_zch(LAYER) + NT * 16;
    const char* zi = sl + dw2_zoff<KA>(LAYER);
    const int ca = (n0 >> 3) + (p >> 1);
    const char* a0 = zi + r0 * 256 + ((ca ^ f128(r0)) << 4) + 8 * (p & 1);
    const char* a1 = zi + (r0 + 4) * 256 + ((ca ^ f128(r0 + 4)) << 4) + 8 * (p & 1);
    const bf16x8 fa = tr_frag2(a0, a1);
    bf16x8 fas = fa;
    if constexpr (FO) {
      constexpr int nb = NT * 16;
      const bool no = (sg[4 + (nb >> 5)] >> ((nb & 31) + (lane & 15))) & 1u;
      fas = xor_sign(fa, no);
    }
    if constexpr (BIAS) acc_bias = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, acc_bias, 0, 0, 0);
    const char* xi = sl + dw2_xoff<KA>(LAYER);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int ra = r0 + t - PAD + HALO, rb = ra + 4;
      const int fa_ = f128(ra), fb_ = f128(rb);
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
          const int cb = ((dw2_xch(LAYER) + (CT0 + c) * 16) >> 3) + (p >> 1);
        const char* b0 = xi + ra * 256 + ((cb ^ fa_) << 4) + 8 * (p & 1);
        const char* b1 = xi + rb * 256 + ((cb ^ fb_) << 4) + 8 * (p & 1);
        const bf16x8 fb = tr_frag2(b0, b1);
        // operands swapped: the tile is accumulated TRANSPOSED (rows = input channels, columns = couts), so a lane ends up
        // with 4 consecutive input channels of one cout = one 16-byte store in flush()
        acc_a[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc_a[t][c], 0, 0, 0);
        if constexpr (FO) {
          const int cbit = (CT0 + c) * 16;   // bit of the layer's own input channel
          const bool ni = (sg[cbit >> 5] >> ((cbit & 31) + (lane & 15))) & 1u;
          acc_b[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xor_sign(fb, ni), fas, acc_b[t][c], 0, 0, 0);
        }
      }
    }
  }

  __device__ __forceinline__ void flush(const TrunkDw2Args& A, int s, int lane) const {
    const LayerDesc ly = A.layers[LAYER];
    const int i4 = 4 * (lane >> 4), jc = lane & 15;
    float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
    float* gwb = A.gw_b + A.gw_stride * s + ly.w_off;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        // transposed tile: this lane holds image channels ch .. ch+3 of cout n (block pads 27 -> 32 of the input are zero
        // columns: their sums are zeros)
        const int ch = (CT0 + c) * 16 + i4, n = NT * 16 + jc;
        if (n >= ly.cout) continue;
        const long o = (long)n * ly.KP + (long)t * ly.cin_img + ch;
        *(f32x4*)(gwa + o) = acc_a[t][c];
        if constexpr (FO) *(f32x4*)(gwb + o) = acc_b[t][c];
      }
    if constexpr (BIAS) {
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = NT * 16 + i4 + r;
          if (n < ly.cout) A.gb_a[(long)A.gb_stride * s + ly.bias_off + n] = acc_bias[r];
        }
      }
    }
  }
};

template <int EM, int SLOT, int O_SGN, class J0, class J1>
__device__ __forceinline__ void dw2_role(const TrunkDw2Args& A, char* smem, int s, int nwin, int lane) {
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
    const char* sl = smem + (k % 3) * SLOT;
    const uint32_t* sg = (const uint32_t*)(smem + O_SGN) + (k % 3) * 48;
    j0.run(sl, sg, lane, ones);
    j1.run(sl, sg, lane, ones);
    lds_barrier();
  }
  j0.flush(A, (int)blockIdx.x, lane);   // slab = this workgroup's (particle, split)
  j1.flush(A, (int)blockIdx.x, lane);
}

struct Dw2Empty {
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void run(const char*, const uint32_t*, int, bf16x8) {}
  __device__ __forceinline__ void flush(const TrunkDw2Args&, int, int) const {}
};

// sign words of layers 4..9 of one window: lane -> (layer = 4 + (lane >> 3), word k = lane & 7), lanes 0..47
__device__ __forceinline__ void dw2_sign_setup(const TrunkDw2Args& A, int s, int split, int lane, const uint32_t*& p, long& stride) {
  p = nullptr;
  stride = 0;
  if (lane >= 48) return;
  const LayerDesc ly = A.layers[4 + (lane >> 3)];
  const int kk = lane & 7;
  if (kk < 4 && kk < ly.sign_in_words) {
    p = A.sign_in + ly.sign_in_off * A.examples + ((long)s * A.B + split) * ly.sign_in_words + kk;
    stride = (long)A.nsplit * ly.sign_in_words;
  } else if (kk >= 4 && kk - 4 < ly.sign_out_words && kk < 6) {
    p = A.sign_out + ly.sign_out_off * A.examples + ((long)s * A.B + split) * ly.sign_out_words + (kk - 4);
    stride = (long)A.nsplit * ly.sign_out_words;
  }
}

// packed max of non-negative bf16 pairs (ReLU outputs: the bit patterns order like the values)
__device__ __forceinline__ uint32_t max2_bf16_pos(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

template <int EM>
__global__ __launch_bounds__(TW2A_THREADS) void trunk_dw2a_kernel(const TrunkDw2Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  const int L = A.L;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TW2A_LDS / 4; k += TW2A_THREADS) z[k] = 0u;
  }
  if (wave >= TW2A_NC) {
    // =========================== loaders ===========================
    // EVERY loader wave stages a quarter (8 rows) of EVERY window: the ACT1 hi plane and its MaxPool1d(3,1,1) copy (the
    // wave loads its rows plus one halo row on each side, so the pooled rows come out of registers), dz(MID), dz(ACT2);
    // wave 0 also the sign words.  Loads are issued two windows ahead (compile-time ring of 2 register sets).  With one wave per window the staging work of a window (27 LDS stores, the pooled copy) WAS the step.
    const int w = wave - TW2A_NC;
    const int c = lane & 15, a0 = 8 * w + 2 * (lane >> 4);   // this lane's rows a0, a0 + 1 of piece c
    const int zrow = 8 * w + lane / 6, zcc = lane % 6, zc = zcc < 2 ? zcc : zcc + 4;   // dz(ACT2): 6 of the row's 10 pieces
    const bool z_on = lane < 48 && zrow < L;
    auto wrow = [&](int k) __attribute__((always_inline)) { return ((long)s * A.B + split + (long)k * A.nsplit) * L; };
    const uint32_t* sgp = nullptr;
    long sgs = 0;
    if (w == 0) dw2_sign_setup(A, s, split, lane, sgp, sgs);
    struct Set {
      tr_u32x4 x[4], g[2], z;
      uint32_t sw;
    };
    auto load = [&](int k, Set& S) __attribute__((always_inline)) {
      const long R0 = wrow(k);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = min(max(a0 - 1 + j, 0), L - 1);   // clamped; rows outside the window are zeroed in put()
        S.x[j] = *(const tr_u32x4*)((const char*)A.x_hi + (R0 + row) * 256 + c * 16);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) S.g[j] = *(const tr_u32x4*)((const char*)A.g_mid + (R0 + min(a0 + j, L - 1)) * 256 + c * 16);
      S.z = *(const tr_u32x4*)((const char*)A.g_act2 + (R0 + (z_on ? zrow : 0)) * 160 + (z_on ? zc : 0) * 16);
      S.sw = 0u;
      if constexpr (FO) {
        if (sgp) S.sw = sgp[(long)k * sgs];
      }
    };
    auto put = [&](int k, const Set& S) __attribute__((always_inline)) {
      char* sl = smem + (k % 3) * TW2A_SLOT;
      tr_u32x4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = a0 - 1 + j;
        x[j] = (row >= 0 && row < L) ? S.x[j] : tr_u32x4{0u, 0u, 0u, 0u};   // values are >= 0: zero is the pool's identity
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = a0 + j, ri = row + HALO;
        if (row < L) {
          *(tr_u32x4*)(sl + ri * 256 + ((c ^ f128(ri)) << 4)) = x[1 + j];
          tr_u32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = max2_bf16_pos(max2_bf16_pos(x[j][e], x[1 + j][e]), x[2 + j][e]);
          *(tr_u32x4*)(sl + TW_PH + ri * 256 + ((c ^ f128(ri)) << 4)) = m;
          *(tr_u32x4*)(sl + 2 * TW_PH + row * 256 + ((c ^ f128(row)) << 4)) = S.g[j];
        }
      }
      if (z_on) *(tr_u32x4*)(sl + 2 * TW_PH + TW_PZ + zrow * 256 + ((zc ^ f128(zrow)) << 4)) = S.z;
      if constexpr (FO) {
        if (w == 0 && lane < 48) ((uint32_t*)(smem + TW2A_O_SGN))[(k % 3) * 48 + lane] = S.sw;
      }
    };
    Set s0, s1;
    if (0 < nwin) load(0, s0);
    if (1 < nwin) load(1, s1);
    __syncthreads();   // zero fill visible
    if (nwin > 0) {
      put(0, s0);
      if (2 < nwin) load(2, s0);
    }
    lds_barrier();     // window 0 staged
    // step t (compute waves: window t): window t+1 from register set (t+1) % 2, which is then reloaded with window t+3
#define TW2A_STEP(T, SET)                        \
  do {                                           \
    if ((T) + 1 < nwin) {                        \
      put((T) + 1, SET);                         \
      if ((T) + 3 < nwin) load((T) + 3, SET);    \
    }                                            \
    lds_barrier();                               \
  } while (0)
    for (int t = 0; t < nwin; t += 2) {
      TW2A_STEP(t, s1);
      if (t + 1 < nwin) TW2A_STEP(t + 1, s0);
    }
#undef TW2A_STEP
    return;
  }
#define DW2A_ROLE(LY, NT) dw2_role<EM, TW2A_SLOT, TW2A_O_SGN, Dw2Job<EM, true, LY, NT, 0, 8, true>, Dw2Empty>(A, smem, s, nwin, lane)
  switch (wave) {
    case 0: DW2A_ROLE(4, 0); break;
    case 1: DW2A_ROLE(5, 0); break;
    case 2: DW2A_ROLE(5, 1); break;
    case 3: DW2A_ROLE(5, 2); break;
    case 4: DW2A_ROLE(5, 3); break;
    case 5: DW2A_ROLE(7, 0); break;
    case 6: DW2A_ROLE(7, 1); break;
    case 7: DW2A_ROLE(7, 2); break;
    case 8: DW2A_ROLE(7, 3); break;
    case 9: DW2A_ROLE(9, 0); break;
    default: DW2A_ROLE(9, 1); break;
  }
#undef DW2A_ROLE
}

template <int EM>
__global__ __launch_bounds__(TW2B_THREADS) void trunk_dw2b_kernel(const TrunkDw2Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  const int L = A.L;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TW2B_LDS / 4; k += TW2B_THREADS) z[k] = 0u;
  }
  if (wave >= TW2B_NC) {
    // loader wave 4 + p stages the windows k = p (mod 2): MID hi, dz(ACT2), sign words
    const int p = wave - TW2B_NC;
    const int nz = L * 16, n2 = L * 10;
    tr_u32x4 b[13];   // ext_vector registers (arrays of HIP's uint4 struct are not split into registers)
    uint32_t sw = 0;
    const uint32_t* sgp;
    long sgs;
    dw2_sign_setup(A, s, split, lane, sgp, sgs);
    auto fetch = [&](int k) __attribute__((always_inline)) {
      const long R0 = ((long)s * A.B + split + (long)k * A.nsplit) * L;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = j * 64 + lane, qq = q < nz ? q : 0;
        b[j] = *(const tr_u32x4*)((const char*)A.x_hi + R0 * 256 + qq * 16);
      }
      // dz(ACT2): only the channels of layers 6 and 8 (pieces 2 .. 5 of the row's 10)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = j * 64 + lane, qq = q < L * 4 ? q : 0;
        b[8 + j] = *(const tr_u32x4*)((const char*)A.g_act2 + (R0 + (qq >> 2)) * 160 + (2 + (qq & 3)) * 16);
      }
      if constexpr (FO) {
        if (sgp) sw = sgp[(long)k * sgs];
      }
    };
    auto put = [&](int k) __attribute__((always_inline)) {
      char* sl = smem + (k % 3) * TW2B_SLOT;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = j * 64 + lane;
        if (q < nz) {
          const int ri = (q >> 4) + HALO, c = q & 15;
          *(tr_u32x4*)(sl + ri * 256 + ((c ^ f128(ri)) << 4)) = b[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = j * 64 + lane;
        if (q < L * 4) {
          const int row = q >> 2, c = 2 + (q & 3);
          *(tr_u32x4*)(sl + TW_PH + row * 256 + ((c ^ f128(row)) << 4)) = b[8 + j];
        }
      }
      if constexpr (FO) {
        if (lane < 48) ((uint32_t*)(smem + TW2B_O_SGN))[(k % 3) * 48 + lane] = sw;
      }
    };
    if (p < nwin) fetch(p);
    __syncthreads();
    if (p == 0 && nwin > 0) put(0);
    if (p == 0 && 2 < nwin) fetch(2);
    lds_barrier();
    for (int t = 0; t < nwin; ++t) {
      const int k = t + 1;
      if ((k & 1) == p) {
        if (k < nwin) put(k);
        if (k + 2 < nwin) fetch(k + 2);
      }
      lds_barrier();
    }
    return;
  }
  // wave j: c-tile j of both layers (k3: 3 tiles, k5: 5 tiles); wave 0 also owns the bias sums
#define DW2B_ROLE(CT, BS) dw2_role<EM, TW2B_SLOT, TW2B_O_SGN, Dw2Job<EM, false, 6, 0, CT, 1, BS>, Dw2Job<EM, false, 8, 0, CT, 1, BS>>(A, smem, s, nwin, lane)
  switch (wave) {
    case 0: DW2B_ROLE(0, true); break;
    case 1: DW2B_ROLE(1, false); break;
    case 2: DW2B_ROLE(2, false); break;
    default: DW2B_ROLE(3, false); break;
  }
#undef DW2B_ROLE
}
