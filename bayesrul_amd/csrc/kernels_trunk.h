// trunk_fwd_kernel : the whole convolutional trunk of the Inception net (nets/inception.py:10-132: both
// inception blocks, 10 variational Conv1d layers) as ONE launch.  bf16 hi/lo planes, gfx950.
//
// One workgroup owns a particle and walks the windows split, split + nsplit, ...; a window flows through a
// three-stage software pipeline inside the workgroup, one stage per step, ONE barrier per step:
//     step t :  loader  : window t+1's x planes (regs -> LDS), window t+2's global loads issued
//               stage 0 : block 1 (4 branches, x -> ACT1 + its MaxPool1d(3,1,1) copy)         window t
//               stage 1 : block 2, 1x1 level (ACT1 -> MID, ACT2 channels 0..15, 48..79)       window t-1
//               stage 2 : block 2, k3 / k5 level (MID -> ACT2 channels 16..47)                window t-2
// Every inter-stage image lives in LDS only (double-buffered by window parity): ACT1, its pooled copy and MID
// never reach HBM as operands of this pass.  What IS written to HBM: ACT2 (hi + lo planes, the dense layer's
// input) and, for a training step, the hi planes of ACT1 / MID and the arg-max codes the backward kernels read.
//
// 11 compute waves + 1 loader wave; every compute wave owns a fixed set of (layer, n-tile) jobs whose weight
// fragments (mean hi / lo, Flipout dW) stay in registers for the whole launch.  The jobs are dealt so that the
// three waves that share a SIMD carry equal MFMA work (the step time is the busiest SIMD's).  LDS images are
// padded rows (row stride = channels * 2 + 32 bytes): a B fragment is `lane base + compile-time offset`.
#pragma once
// diagnostics builds only (tests/probes/ablate_gpu.sh): timing with parts of the kernel removed; results are wrong.
// The product library is built with TR_ABL == 0.
#ifndef TR_ABL
#define TR_ABL 0
#endif
#ifndef TR_SWAP
#define TR_SWAP 0
#endif
#include "kernels_conv_bf.h"

enum { TR_NC = 11, TR_NW = 12, TR_THREADS = TR_NW * 64 };
enum {
  // Row pitches in 16-byte slots are = 2 (mod 4): ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, ... of
  // MI355X_MICROARCH.md (LDS), i.e. 8 rows of k-group g and 8 rows of k-group g+1 at once; with an even pitch p the
  // rows of a k-group fall on slots of one parity (p * {0..3, 12..15} = 8 distinct even residues mod 16), the
  // neighbouring k-group on the other: conflict-free for every tap shift.  (A 16-byte pad, pitch 5 or 17, is 2-way.)
  TR_RSX = 96,                       // bytes per row of an x image ([36][32] bf16 + 32 pad): pitch 6
  TR_RSB = 288,                      // bytes per row of a 128-channel image (+ 32 pad): pitch 18
  TR_PX = IMG_ROWS * TR_RSX,         // 3,456
  TR_PB = IMG_ROWS * TR_RSB,         // 10,368
  TR_PA = TILE_ROWS * TR_RSB,        // 9,216: ACT1 planes have no halo rows (every reader is a 1x1 conv); addressed from
  TR_O_X = 0,                        //        TR_A1B so that image row r + HALO is row r, like the other images
  TR_O_A1 = TR_O_X + 2 * 4 * TR_PX,  // [2 bufs][ACT1 hi, lo, pooled hi, pooled lo]
  TR_A1B = TR_O_A1 - HALO * TR_RSB,
  TR_O_MID = TR_O_A1 + 2 * 4 * TR_PA,  // [2 bufs][MID hi, lo]
  TR_O_SGN = TR_O_MID + 2 * 2 * TR_PB,  // [4 slots][10 layers][8 words]
  TR_O_LUT = TR_O_SGN + 4 * 80 * 4,     // 512 x 16 B: XOR mask of a dW fragment = s_in byte of its 8 K channels, s_out bit of its row
  TR_O_WL = TR_O_LUT + 8192,         // lo weight fragments of the one job too wide for the register file: [10 k-steps][64 lanes][16 B]
  TR_LDS = TR_O_WL + 10 * 1024
};

// compile-time layer table of the trunk (layer ids of kInception in plan.hip)
__host__ __device__ constexpr int tl_taps(int l) { return (l == 1 || l == 3 || l == 6) ? 3 : ((l == 2 || l == 8) ? 5 : 1); }
__host__ __device__ constexpr int tl_g8(int l) { return l < 4 ? 4 : ((l == 6 || l == 8) ? 8 : 16); }   // 8-channel chunks per tap
__host__ __device__ constexpr int tl_inch(int l) { return l == 8 ? 8 : 0; }                            // first chunk inside the input image
__host__ __device__ constexpr int tl_pool(int l) { return (l == 3 || l == 9) ? 1 : 0; }
__host__ __device__ constexpr int tl_cout(int l) { return l < 4 ? 27 : ((l == 5 || l == 7) ? 64 : (l == 9 ? 32 : 16)); }
__host__ __device__ constexpr int tl_stage(int l) { return l < 4 ? 0 : ((l == 6 || l == 8) ? 2 : 1); }
__host__ __device__ constexpr int tl_outk(int l) { return l < 4 ? 0 : ((l == 5 || l == 7) ? 1 : 2); }   // 0 ACT1, 1 MID, 2 ACT2
__host__ __device__ constexpr int tl_ooff(int l) {
  return l == 1 ? 32 : (l == 2 ? 64 : (l == 3 ? 96 : (l == 6 ? 16 : (l == 7 ? 64 : (l == 8 ? 32 : (l == 9 ? 48 : 0))))));
}
__host__ __device__ constexpr int tl_nks(int l) { return tl_taps(l) * tl_g8(l) / 4; }

struct TrunkArgs {
  const u16* xp[4];          // x hi, x lo, pooled x hi, pooled x lo: [B * L][32] bf16
  WeightSlots ws;
  const LayerDesc* layers;   // device table
  const uint32_t* sign_in;   // packed Flipout signs, all layers (NoiseRefs layout)
  const uint32_t* sign_out;
  long examples;             // S * B of the call
  u16* act1_hi;              // [S*B*L][128]  (training step only, else null)
  u16* mid_hi;               // [S*B*L][128]
  u16* act2_hi;              // [S*B*L][80]
  u16* act2_lo;
  unsigned char* amax;       // [S*B*L][32] 2-bit arg-max codes of block 2's pooled branch, 4 channels per byte (training step only)
  unsigned char* m_act1;     // [S*B*L][16] bit masks [ACT1 > 0], 8 channels per byte (training step only): what the backward
  unsigned char* m_mid;      // [S*B*L][16] needs of ACT1 / MID besides the dW operands
  int S, B, L, nsplit;
};

// WL_LDS: the job's lo weight fragments live in LDS (lane-linear, conflict-free) instead of registers
template <int LAYER, int NT, bool WL_LDS = false>
struct TJob {
  static constexpr int layer = LAYER, nt = NT, nks = tl_nks(LAYER);
  static constexpr bool wl_lds = WL_LDS;
};
struct TNone {
  static constexpr int layer = -1, nt = 0, nks = 0;
  static constexpr bool wl_lds = false;
};

typedef unsigned int tr_u32x4 __attribute__((ext_vector_type(4)));

// 512-entry table: index = s_out bit of the fragment's row << 8 | s_in byte of its 8 K channels -> XOR mask
__device__ __forceinline__ void build_sign_lut2(uint4* lut, int tid, int nthreads) {
  for (int e = tid; e < 512; e += nthreads) {
    const uint32_t rs = (uint32_t)(e >> 8) & 1u;
    uint32_t m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = ((((e >> (2 * q)) & 1u) ^ rs) << 15) | ((((e >> (2 * q + 1)) & 1u) ^ rs) << 31);
    lut[e] = make_uint4(m[0], m[1], m[2], m[3]);
  }
}

// register state + step body of one job
template <int EM, bool TRAIN, class J>
struct TrunkJobRun {
  static constexpr int LY = J::layer, NT = J::nt, NKS = J::nks;
  static constexpr bool FO = (EM == EM_FLIPOUT);
  static constexpr int STAGE = tl_stage(LY), TAPS = tl_taps(LY), PAD = (TAPS - 1) / 2, G8 = tl_g8(LY);
  static constexpr int RS = STAGE == 0 ? TR_RSX : TR_RSB;
  static constexpr bool WLL = J::wl_lds;
  bf16x8 wh[NKS], wl[WLL ? 1 : NKS], wb[FO ? NKS : 1];
  f32x4 bias;

  __device__ __forceinline__ void init(const TrunkArgs& A, char* smem, int s, int lane) {
    const int i16 = lane & 15, g4 = lane >> 4;
    const LayerDesc ly = A.layers[LY];
    const long sa = A.ws.slot_stride_a * s, sb = A.ws.slot_stride_b * s;
    const long row = (long)(NT * 16 + i16) * ly.KP + ly.w_off + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      wh[ks] = *(const bf16x8*)((const u16*)A.ws.a_hi + sa + row + ks * 32);
      const bf16x8 l = *(const bf16x8*)((const u16*)A.ws.a_lo + sa + row + ks * 32);
      if constexpr (WLL) *(bf16x8*)(smem + TR_O_WL + ks * 1024 + lane * 16) = l;
      else wl[ks] = l;
      if constexpr (FO) wb[ks] = *(const bf16x8*)((const u16*)A.ws.b + sb + row + ks * 32);
    }
    const int chb = NT * 16 + 4 * g4;
    const int nv = tl_cout(LY) - chb;
    const float* ba = A.ws.bias_a + (long)A.ws.bias_stride_a * s + ly.bias_off + chb;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = r < nv ? ba[r] : 0.f;
  }

  // k = index of the window inside this workgroup's list; R0 = first row of the window in the [S*B*L] row space
  __device__ __forceinline__ void run(const TrunkArgs& A, char* smem, int k, unsigned R0, int lane) const {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int par = k & 1;
    const char* in_hi;
    if constexpr (STAGE == 0) in_hi = smem + TR_O_X + par * 4 * TR_PX + (tl_pool(LY) ? 2 * TR_PX : 0);
    else if constexpr (STAGE == 1) in_hi = smem + TR_A1B + par * 4 * TR_PA + (tl_pool(LY) ? 2 * TR_PA : 0);
    else in_hi = smem + TR_O_MID + par * 2 * TR_PB;
    constexpr int PLANE = STAGE == 0 ? TR_PX : (STAGE == 1 ? TR_PA : TR_PB);
    const char* lb = in_hi + i16 * RS + g4 * 16;
    const uint32_t* sg = (const uint32_t*)(smem + TR_O_SGN) + (k & 3) * 80 + LY * 8;
    const uint4* lut = (const uint4*)(smem + TR_O_LUT);
    constexpr int chb0 = NT * 16;
    // one accumulator: out = bias + W_mu x  (+ Flipout: (s_out o dW o s_in) x, both signs folded into the dW fragment)
    f32x4 acc[2];
    if constexpr (TR_ABL & 2) acc[0] = acc[1] = bias;
    uint32_t rs8 = 0;
    if constexpr (FO) rs8 = ((sg[4 + (chb0 >> 5)] >> ((chb0 & 31) + i16)) & 1u) << 8;   // s_out of this lane's fragment row
#pragma unroll
    for (int ks = 0; ks < ((TR_ABL & 2) ? 0 : NKS); ++ks) {
      const int tap = (ks * 4) / G8, c0 = (ks * 4) % G8;
      bf16x8 wlk;
      if constexpr (WLL) wlk = *(const bf16x8*)(smem + TR_O_WL + ks * 1024 + lane * 16);
      else wlk = wl[ks];
      bf16x8 wbm = wh[ks];
      if constexpr (FO) {
        const int cl = c0 + g4;   // 8-channel chunk of the layer's own input channels
        const uint32_t byte = (sg[cl >> 2] >> ((cl & 3) * 8)) & 0xffu;
        const tr_u32x4 fm = __builtin_bit_cast(tr_u32x4, lut[rs8 | byte]);
        wbm = __builtin_bit_cast(bf16x8, __builtin_bit_cast(tr_u32x4, wb[ks]) ^ fm);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int off = (mt * 16 + tap - PAD + HALO) * RS + (tl_inch(LY) + c0) * 16;
        const bf16x8 bh = *(const bf16x8*)(lb + off);
        const bf16x8 bl = *(const bf16x8*)(lb + PLANE + off);
        // the first MFMA of a tile takes the bias registers as its C operand (no copy of the bias into the accumulator)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bh, ks == 0 ? bias : acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlk, bh, acc[mt], 0, 0, 0);
        if constexpr (FO) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbm, bh, acc[mt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the next job's operand reads out of this job's epilogue (register pressure)
    // ---------------- epilogue: ReLU, hi/lo split ----------------
    const int chb = chb0 + 4 * g4;
    f32x4 v[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[mt][r] = fmaxf(acc[mt][r], 0.f);   // every conv of the trunk is followed by ReLU (inception.py:48-60, 118-131)
    const int L = A.L;
    constexpr int OOFF = tl_ooff(LY), OUTK = tl_outk(LY);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      uint2 hv, lv;
      split4(v[mt], hv, lv);
      if (row < L) {
        if constexpr (OUTK == 0 || OUTK == 1) {
          char* img = smem + (OUTK == 0 ? TR_A1B + par * 4 * TR_PA : TR_O_MID + par * 2 * TR_PB);
          const int o = (row + HALO) * TR_RSB + (OOFF + chb) * 2;
          *(uint2*)(img + o) = hv;
          *(uint2*)(img + (OUTK == 0 ? TR_PA : TR_PB) + o) = lv;
          if constexpr (TRAIN && !(TR_ABL & 1)) {
            // 32-bit byte offsets from the (uniform) plane base: one scalar base + one vector offset per store
            char* g = (char*)(OUTK == 0 ? A.act1_hi : A.mid_hi);
            *(uint2*)(g + ((R0 + (unsigned)row) * 256u + (unsigned)((OOFF + chb) * 2))) = hv;
          }
        } else {
          const unsigned oo = (R0 + (unsigned)row) * 160u + (unsigned)((OOFF + chb) * 2);
          if constexpr (!(TR_ABL & 1)) {
            *(uint2*)((char*)A.act2_hi + oo) = hv;
            *(uint2*)((char*)A.act2_lo + oo) = lv;
          } else {
            asm volatile("" ::"v"(hv.x), "v"(hv.y), "v"(lv.x), "v"(lv.y), "v"(oo));
          }
        }
      }
    }
    // ---------------- block 1 only: MaxPool1d(3,1,1) of the output rows (inception.py:99-104 reads it) ----------------
    if constexpr (OUTK == 0 && !(TR_ABL & 4)) {
      // rows live on the 16 lanes of a DPP row: row-1 / row+1 are one lane away; the seam between the two
      // m-tiles (rows 15 | 16) takes the other accumulator.  torch keeps the FIRST maximum of (row-1, row, row+1).
      f32x4 p[2];
      uint32_t code[2] = {0u, 0u};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a0 = v[0][r], a1 = v[1][r];
        const float up0 = rot16<0x121>(a0), up1 = rot16<0x121>(a1);   // lane i <- lane i-1 (mod 16)
        const float dn0 = rot16<0x12F>(a0), dn1 = rot16<0x12F>(a1);   // lane i <- lane i+1 (mod 16)
        {   // m-tile 0: row = i16
          float best = a0;
          uint32_t c = 1u;
          if (i16 > 0 && up0 >= best) { best = up0; c = 0u; }
          const float d = i16 == 15 ? dn1 : dn0;
          if (i16 + 1 < L && d > best) { best = d; c = 2u; }
          p[0][r] = best;
          if constexpr (TRAIN) code[0] |= c << (2 * r);
        }
        {   // m-tile 1: row = 16 + i16
          float best = a1;
          uint32_t c = 1u;
          const float u = i16 == 0 ? up0 : up1;
          if (u >= best) { best = u; c = 0u; }
          if (17 + i16 < L && dn1 > best) { best = dn1; c = 2u; }
          p[1][r] = best;
          if constexpr (TRAIN) code[1] |= c << (2 * r);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        uint2 hv, lv;
        split4(p[mt], hv, lv);
        if (row < L) {
          char* img = smem + TR_A1B + par * 4 * TR_PA + 2 * TR_PA;
          const int o = (row + HALO) * TR_RSB + (OOFF + chb) * 2;
          *(uint2*)(img + o) = hv;
          *(uint2*)(img + TR_PA + o) = lv;
          if constexpr (TRAIN && !(TR_ABL & 1)) A.amax[(R0 + (unsigned)row) * 32u + (unsigned)((OOFF + chb) >> 2)] = (unsigned char)code[mt];
        }
      }
    }
  }
};

template <int EM, bool TRAIN>
struct TrunkJobRun<EM, TRAIN, TNone> {
  __device__ __forceinline__ void init(const TrunkArgs&, char*, int, int) {}
  __device__ __forceinline__ void run(const TrunkArgs&, char*, int, unsigned, int) const {}
};

template <class J>
__device__ __forceinline__ constexpr int tj_stage() {
  if constexpr (J::layer < 0) return 0;
  else return tl_stage(J::layer);
}

template <int EM, bool TRAIN, class J0, class J1, class J2>
__device__ __forceinline__ void trunk_role(const TrunkArgs& A, char* smem, int s, int split, int nwin, int lane) {
  TrunkJobRun<EM, TRAIN, J0> r0;
  TrunkJobRun<EM, TRAIN, J1> r1;
  TrunkJobRun<EM, TRAIN, J2> r2;
  r0.init(A, smem, s, lane);
  r1.init(A, smem, s, lane);
  r2.init(A, smem, s, lane);
  __syncthreads();   // zero fill + sign table
  lds_barrier();     // window 0 staged
  const int nsteps = nwin + 2;
  // row indices fit 32 bits (the host refuses launches whose planes exceed 4 GiB)
  const unsigned Rs = (unsigned)(((long)s * A.B + split) * A.L), Rstep = (unsigned)(A.nsplit * A.L);
  for (int t = 0; t < nsteps; ++t) {
    {
      const int k = t - tj_stage<J0>();
      if (J0::layer >= 0 && k >= 0 && k < nwin) r0.run(A, smem, k, Rs + k * Rstep, lane);
    }
    {
      const int k = t - tj_stage<J1>();
      if (J1::layer >= 0 && k >= 0 && k < nwin) r1.run(A, smem, k, Rs + k * Rstep, lane);
    }
    {
      const int k = t - tj_stage<J2>();
      if (J2::layer >= 0 && k >= 0 && k < nwin) r2.run(A, smem, k, Rs + k * Rstep, lane);
    }
    lds_barrier();
  }
}

// loader wave: x planes (and Flipout sign words) of the next windows, global -> registers -> LDS
// bit mask [v > 0] of the 8 bf16 values of a 16-byte piece (ReLU outputs: > 0 <=> non-zero)
__device__ __forceinline__ uint32_t tr_mask8(tr_u32x4 d) {
  uint32_t b = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t t;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(d[i]), "v"(0x00010001u));
    b |= ((t | (t >> 15)) & 3u) << (2 * i);
  }
  return b;
}

// mask plane rows of one window from its hi image in LDS: lane = (row, half of the 128 channels) -> 8 bytes
__device__ __forceinline__ void trunk_mask_rows(const char* img, unsigned char* plane, unsigned R0, int L, int lane) {
  const int row = lane >> 1, half = lane & 1;
  if (row >= L) return;
  const char* p = img + (row + HALO) * TR_RSB + half * 128;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lo |= tr_mask8(*(const tr_u32x4*)(p + j * 16)) << (8 * j);
    hi |= tr_mask8(*(const tr_u32x4*)(p + (4 + j) * 16)) << (8 * j);
  }
  *(uint2*)(plane + (R0 + (unsigned)row) * 16u + (unsigned)(half * 8)) = make_uint2(lo, hi);
}

template <int EM, bool TRAIN>
__device__ __forceinline__ void trunk_loader(const TrunkArgs& A, char* smem, int s, int split, int nwin, int lane) {
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int L = A.L;
  const int nch = L * 4;                 // 16-byte chunks of one plane of one window
  // chunk q of the 4 planes: q = j * 64 + lane over [4 planes][nch]
  const int tot = 4 * nch;
  // per-lane source offset / LDS destination of chunk q = j * 64 + lane (planes are `pstride` bytes apart in the
  // workspace); only the last instruction (j = 7) can run past the end: its lanes load chunk 0 and do not store
  const long pstride = (const char*)A.xp[1] - (const char*)A.xp[0];
  const char* base = (const char*)A.xp[0] + ((long)split * L * 32) * 2;
  long soff[8];
  int dst[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q = j * 64 + lane;
    const int qq = q < tot ? q : 0;
    const int pl = qq / nch, c = qq - pl * nch;
    soff[j] = pl * pstride + (long)c * 16;
    dst[j] = pl * TR_PX + ((c >> 2) + HALO) * TR_RSX + (c & 3) * 16;
  }
  const bool on7 = 7 * 64 + lane < tot;
  const long wstep = (long)A.nsplit * L * 32 * 2;
  // sign words: lane -> (layer = lane >> 3, word k = lane & 7) for layers 0..7; lanes 0..15 also layers 8, 9
  const uint32_t* sg0 = nullptr;
  const uint32_t* sg1 = nullptr;
  long sst0 = 0, sst1 = 0;
  if constexpr (FO) {
    auto setup = [&](int layer, int kk, const uint32_t*& p, long& stride) {
      const LayerDesc ly = A.layers[layer];
      if (kk < 4 && kk < ly.sign_in_words) {
        p = A.sign_in + ly.sign_in_off * A.examples + ((long)s * A.B + split) * ly.sign_in_words + kk;
        stride = (long)A.nsplit * ly.sign_in_words;
      } else if (kk >= 4 && kk - 4 < ly.sign_out_words && kk < 6) {
        p = A.sign_out + ly.sign_out_off * A.examples + ((long)s * A.B + split) * ly.sign_out_words + (kk - 4);
        stride = (long)A.nsplit * ly.sign_out_words;
      }
    };
    setup(lane >> 3, lane & 7, sg0, sst0);
    if (lane < 16) setup(8 + (lane >> 3), lane & 7, sg1, sst1);
  }
  uint4 b0, b1, b2, b3, b4, b5, b6, b7;
  uint32_t sb0 = 0, sb1 = 0;
#define TR_FETCH()                                              \
  do {                                                          \
    b0 = *(const uint4*)(base + soff[0]);                       \
    b1 = *(const uint4*)(base + soff[1]);                       \
    b2 = *(const uint4*)(base + soff[2]);                       \
    b3 = *(const uint4*)(base + soff[3]);                       \
    b4 = *(const uint4*)(base + soff[4]);                       \
    b5 = *(const uint4*)(base + soff[5]);                       \
    b6 = *(const uint4*)(base + soff[6]);                       \
    b7 = *(const uint4*)(base + soff[7]);                       \
    base += wstep;                                              \
    if constexpr (FO) {                                         \
      if (sg0) { sb0 = *sg0; sg0 += sst0; }                     \
      if (sg1) { sb1 = *sg1; sg1 += sst1; }                     \
    }                                                           \
  } while (0)
#define TR_PUT(K)                                                         \
  do {                                                                    \
    char* xs = smem + TR_O_X + ((K) & 1) * 4 * TR_PX;                     \
    *(uint4*)(xs + dst[0]) = b0;                                          \
    *(uint4*)(xs + dst[1]) = b1;                                          \
    *(uint4*)(xs + dst[2]) = b2;                                          \
    *(uint4*)(xs + dst[3]) = b3;                                          \
    *(uint4*)(xs + dst[4]) = b4;                                          \
    *(uint4*)(xs + dst[5]) = b5;                                          \
    *(uint4*)(xs + dst[6]) = b6;                                          \
    if (on7) *(uint4*)(xs + dst[7]) = b7;                                 \
    if constexpr (FO) {                                                   \
      uint32_t* sgw = (uint32_t*)(smem + TR_O_SGN) + ((K) & 3) * 80;      \
      sgw[lane] = sb0;                                                    \
      if (lane < 16) sgw[64 + lane] = sb1;                                \
    }                                                                     \
  } while (0)
  if (nwin > 0) TR_FETCH();
  __syncthreads();   // zero fill + sign table
  if (nwin > 0) TR_PUT(0);
  if (nwin > 1) TR_FETCH();
  lds_barrier();     // window 0 staged
  const int nsteps = nwin + 2;
  const unsigned Rs = (unsigned)(((long)s * A.B + split) * A.L), Rstep = (unsigned)(A.nsplit * A.L);
  for (int t = 0; t < nsteps; ++t) {
    if (t + 1 < nwin) TR_PUT(t + 1);
    if (t + 2 < nwin) TR_FETCH();
    if constexpr (TRAIN) {
      // the backward's ReLU masks of ACT1 (window t-1: its image is complete and is being read by the 1x1 level) and of
      // MID (window t-2), from the hi images in LDS: 16 bytes per row instead of the 256-byte hi rows the dX would re-read
      if (t >= 1 && t - 1 < nwin) trunk_mask_rows(smem + TR_A1B + ((t - 1) & 1) * 4 * TR_PA, A.m_act1, Rs + (t - 1) * Rstep, L, lane);
      if (t >= 2 && t - 2 < nwin) trunk_mask_rows(smem + TR_O_MID + ((t - 2) & 1) * 2 * TR_PB, A.m_mid, Rs + (t - 2) * Rstep, L, lane);
    }
    lds_barrier();
  }
#undef TR_FETCH
#undef TR_PUT
}

template <int EM, bool TRAIN>
__global__ __launch_bounds__(TR_THREADS) void trunk_fwd_kernel(const TrunkArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TR_O_LUT / 4; k += TR_THREADS) z[k] = 0u;
    build_sign_lut2((uint4*)(smem + TR_O_LUT), tid, TR_THREADS);
  }
  // jobs: (layer, n-tile).  Dealt by cost (MFMAs + epilogue, the block-1 epilogue pools) so that every wave, and the
  // three waves that share a SIMD (w, w+4, w+8), carry about the same work per step.
#define TR_ROLE(...) trunk_role<EM, TRAIN, __VA_ARGS__>(A, smem, s, split, nwin, lane)
  // TR_SWAP (diagnostics): bit w = wave w runs its two jobs in the other order (the order inside a step is free: every
  // job reads images of earlier steps and writes images of later ones)
#define TR_ROLE2(W, JA, JB) do { if constexpr ((TR_SWAP >> (W)) & 1) TR_ROLE(JB, JA, TNone); else TR_ROLE(JA, JB, TNone); } while (0)
#define TJ(...) TJob<__VA_ARGS__>
#ifdef TR_PERM
  // diagnostics: role r runs on wave perm[r] (search for a better placement of the roles on the SIMDs)
  {
    constexpr int perm[12] = {TR_PERM};
    if (wave == perm[0]) TR_ROLE(TJob<8, 0, true>, TNone, TNone);
    else if (wave == perm[1]) TR_ROLE2(1, TJ(6, 0), TJ(0, 0));
    else if (wave == perm[2]) TR_ROLE2(2, TJ(2, 0), TJ(4, 0));
    else if (wave == perm[3]) TR_ROLE2(3, TJ(2, 1), TJ(9, 0));
    else if (wave == perm[4]) TR_ROLE2(4, TJ(5, 0), TJ(1, 0));
    else if (wave == perm[5]) TR_ROLE2(5, TJ(5, 3), TJ(1, 1));
    else if (wave == perm[6]) TR_ROLE2(6, TJ(7, 2), TJ(3, 0));
    else if (wave == perm[7]) TR_ROLE2(7, TJ(9, 1), TJ(3, 1));
    else if (wave == perm[8]) TR_ROLE2(8, TJ(5, 1), TJ(5, 2));
    else if (wave == perm[9]) TR_ROLE2(9, TJ(7, 0), TJ(7, 1));
    else if (wave == perm[10]) TR_ROLE2(10, TJ(7, 3), TJ(0, 1));
    else trunk_loader<EM, TRAIN>(A, smem, s, split, nwin, lane);
  }
#else
  switch (wave) {
    // the k5 64->16 job (10 k-steps) and the loader trade places on the training step: its loader also builds the two
    // ReLU mask planes, and the placement search (TR_PERM) found wave 0 for it 2-3 % faster there, 3 % slower without
    case 0:
      if constexpr (TRAIN) trunk_loader<EM, TRAIN>(A, smem, s, split, nwin, lane);
      else TR_ROLE(TJob<8, 0, true>, TNone, TNone);
      break;
    case 4: TR_ROLE2(4, TJ(5, 0), TJ(1, 0)); break;
    case 8: TR_ROLE2(8, TJ(5, 1), TJ(5, 2)); break;
    case 1: TR_ROLE2(1, TJ(6, 0), TJ(0, 0)); break;           // k3 64->16 (6 k-steps) + block-1 k1
    case 5: TR_ROLE2(5, TJ(5, 3), TJ(1, 1)); break;
    case 9: TR_ROLE2(9, TJ(7, 0), TJ(7, 1)); break;
    case 2: TR_ROLE2(2, TJ(2, 0), TJ(4, 0)); break;           // block-1 k5 (5 k-steps) + a 1x1 tile
    case 6: TR_ROLE2(6, TJ(7, 2), TJ(3, 0)); break;
    case 10: TR_ROLE2(10, TJ(7, 3), TJ(0, 1)); break;
    case 3: TR_ROLE2(3, TJ(2, 1), TJ(9, 0)); break;
    case 7: TR_ROLE2(7, TJ(9, 1), TJ(3, 1)); break;
    default:
      if constexpr (TRAIN) TR_ROLE(TJob<8, 0, true>, TNone, TNone);
      else trunk_loader<EM, TRAIN>(A, smem, s, split, nwin, lane);
      break;
  }
#endif
#undef TJ
#undef TR_ROLE2
#undef TR_ROLE
}

// bf16 hi/lo planes [rows][32] of the raw fp32 windows and of their MaxPool1d(3,1,1) copy (block 1's pooled
// branch, inception.py:41-46); channel pads zero.  rows = B * L, pooling stays inside a window.
// 8 channels per thread (one 16-byte store per plane): idx8 = row * 4 + channel group
__device__ __forceinline__ void x_planes4_dev8(const float* x, u16* hi, u16* lo, u16* phi, u16* plo, long rows, int L, int F, long idx8) {
  if (idx8 >= rows * 4) return;
  const long r = idx8 >> 2;
  const int c0 = (int)(idx8 & 3) * 8;
  const int l = (int)(r % L);
  const bool up = l > 0, dn = l + 1 < L;
  float v[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    const bool on = c < F;
    const int cc = on ? c : 0;
    const float a = x[r * F + cc];
    const float b = x[(up ? r - 1 : r) * F + cc], d = x[(dn ? r + 1 : r) * F + cc];
    v[k] = on ? a : 0.f;
    q[k] = on ? fmaxf(fmaxf(a, b), d) : 0.f;   // rows outside the window are replaced by the row itself
  }
  uint32_t h[4], lo4[4], ph[4], pl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u16 h0 = f2bf(v[2 * k]), h1 = f2bf(v[2 * k + 1]);
    const u16 p0 = f2bf(q[2 * k]), p1 = f2bf(q[2 * k + 1]);
    h[k] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    lo4[k] = (uint32_t)f2bf(v[2 * k] - bf2f(h0)) | ((uint32_t)f2bf(v[2 * k + 1] - bf2f(h1)) << 16);
    ph[k] = (uint32_t)p0 | ((uint32_t)p1 << 16);
    pl[k] = (uint32_t)f2bf(q[2 * k] - bf2f(p0)) | ((uint32_t)f2bf(q[2 * k + 1] - bf2f(p1)) << 16);
  }
  const long o = r * 32 + c0;
  *(uint4*)(hi + o) = make_uint4(h[0], h[1], h[2], h[3]);
  *(uint4*)(lo + o) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
  *(uint4*)(phi + o) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
  *(uint4*)(plo + o) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}
__global__ void x_planes4_kernel(const float* x, u16* hi, u16* lo, u16* phi, u16* plo, long rows, int L, int F) {
  x_planes4_dev8(x, hi, lo, phi, plo, rows, L, F, (long)blockIdx.x * blockDim.x + threadIdx.x);   // rows * 4 threads
}
