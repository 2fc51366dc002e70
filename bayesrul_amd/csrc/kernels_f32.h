// Exact-fp32 fused kernels of the Inception net (the reference's arithmetic precision: conf/trainer/default.yaml:8-12
// trains in fp32).  v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain and runs at 1/16 of the bf16 matrix rate, so these
// kernels are MATRIX-PIPE bound by construction: what matters is (a) issuing no MFMA that is not needed (K walks 20 of
// the 32 padded x channels; Flipout's dW is one contraction + a sign outer product, see tf_dw_kernel), (b) keeping the pipe
// fed: weight fragments stay in registers for the whole launch, every activation operand comes from LDS, two waves per
// SIMD so that one wave's epilogue hides under the other's MFMAs.
//
// An fp32 image of C channels has the shape of a bf16 image of 2 C channels: a lane's ds_read_b128 delivers 4
// consecutive channels = its K element of FOUR 16x16x4 k-steps (k-step j of a 16-channel block takes channel 4 g + j from
// lane group g, in the weight fragment and in the activation fragment alike), so the conflict-free row pitches of
// kernels_trunk.h (pitch = 2 mod 4 sixteen-byte slots) carry over.
//
// Reference arithmetic: nets/inception.py:10-132 (both inception blocks), [3P] tyxe.poutine.flipout (bayesian.py:68-69).
#pragma once
#include "kernels_trunk.h"

enum { TF_WAVES = 8, TF_THREADS = TF_WAVES * 64 };
enum {
  TF_XC = 20,                          // channels of an x image row: 18 features + 2 zero pads (5 k-steps instead of 8)
  TF_RSX = TF_XC * 4 + 16,             // 96 bytes: pitch 6
  TF_RSB = 128 * 4 + 32,               // 544 bytes: pitch 34
  TF_PX = IMG_ROWS * TF_RSX,           // 3,456
  TF_PA = TILE_ROWS * TF_RSB,          // 17,408: ACT1 images have no halo rows (every reader is a 1x1 conv)
  TF_PB = IMG_ROWS * TF_RSB,           // 19,584
  TF_O_X = 0,                          // [2 bufs][x, pooled x]
  TF_O_A1 = TF_O_X + 2 * 2 * TF_PX,    // [2 bufs][ACT1, pooled ACT1]
  TF_A1B = TF_O_A1 - HALO * TF_RSB,    // image row r + HALO is row r, as in the other images
  TF_O_MID = TF_O_A1 + 2 * 2 * TF_PA,  // [2 bufs][MID]
  TF_O_SGN = TF_O_MID + 2 * TF_PB,     // [4 slots][10 layers][8 words]
  TF_O_LUT = TF_O_SGN + 4 * 80 * 4,    // 32 x 16 B: (s_out bit, 4 s_in bits) -> sign masks of a 4-channel fragment
  TF_LDS = TF_O_LUT + 32 * 16
};
static_assert(TF_A1B >= 0, "ACT1 image base");

// per-layer geometry of the fp32 trunk kernels (layer ids of kInception in plan.hip; taps / stage / output of tl_*)
__host__ __device__ constexpr int tf_cb(int l) { return l < 4 ? 1 : ((l == 6 || l == 8) ? 4 : 8); }      // full 16-channel blocks per tap
__host__ __device__ constexpr int tf_tail(int l) { return l < 4 ? 1 : 0; }                                 // + one 4-channel k-step (x channels 16..19)
__host__ __device__ constexpr int tf_inch(int l) { return l == 8 ? 64 : 0; }                               // first channel inside the input image
__host__ __device__ constexpr int tf_cimg(int l) { return l < 4 ? 32 : ((l == 6 || l == 8) ? 64 : 128); }  // LayerDesc::cin_img

struct TfArgs {
  const float* xp[2];        // x, MaxPool1d(3,1,1) of x: [B * L][20] fp32 (xf_planes_kernel)
  WeightSlots ws;
  const LayerDesc* layers;   // device table
  const uint32_t* sign_in;   // packed Flipout signs, all layers (NoiseRefs layout)
  const uint32_t* sign_out;
  long examples;             // S * B of the call
  float* act1;               // [S*B*L][128]  (training step only: dW operand)
  float* mid;                // [S*B*L][128]  (training step only)
  float* act2;               // [S*B*L][80]
  unsigned char* amax;       // [S*B*L][32] 2-bit arg-max codes of block 2's pooled branch, 4 channels per byte (training step only)
  unsigned char* m_act1;     // [S*B*L][32] ReLU masks [ACT1 > 0], 4 channels per byte (low nibble)
  unsigned char* m_mid;      // [S*B*L][32] [MID > 0]
  int S, B, L, nsplit;
};

typedef unsigned int tf_u32x4 __attribute__((ext_vector_type(4)));

// index = s_out bit << 4 | 4 s_in bits -> XOR masks (sign bits) of the 4 fp32 values
__device__ __forceinline__ void build_sign_lut_f32(uint4* lut, int tid) {
  if (tid < 32) {
    const uint32_t rs = (uint32_t)(tid >> 4) & 1u;
    lut[tid] = make_uint4((((tid >> 0) & 1u) ^ rs) << 31, (((tid >> 1) & 1u) ^ rs) << 31, (((tid >> 2) & 1u) ^ rs) << 31,
                          (((tid >> 3) & 1u) ^ rs) << 31);
  }
}
__device__ __forceinline__ f32x4 xor4(f32x4 v, uint4 m) {
  return __builtin_bit_cast(f32x4, __builtin_bit_cast(tf_u32x4, v) ^ tf_u32x4{m.x, m.y, m.z, m.w});
}
__device__ __forceinline__ float xor1(float v, uint32_t m) { return __uint_as_float(__float_as_uint(v) ^ m); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ==========================================================================================
// tf_fwd_kernel : block 1 + block 2 of the Inception net in ONE launch (exact fp32).
//   workgroup = (particle, split) walks its windows through the three-stage pipeline of trunk_fwd_kernel (block 1 ->
//   1x1 level -> k3 / k5 level on windows t, t-1, t-2; inter-stage images in LDS only; one barrier per step).  8 waves
//   = 2 per SIMD, up to 256 registers each: the 22 (layer, n-tile) jobs are dealt so that the four SIMDs carry equal MFMA
//   counts (596 / 608 / 592 / 604 per window with Flipout); wave 6 also stages the next window's x planes and sign words.
// ==========================================================================================
template <int EM, bool TRAIN, class J>
struct TfJobRun {
  static constexpr int LY = J::layer, NT = J::nt;
  static constexpr bool FO = (EM == EM_FLIPOUT);
  static constexpr int STAGE = tl_stage(LY), TAPS = tl_taps(LY), PAD = (TAPS - 1) / 2, CB = tf_cb(LY), TAIL = tf_tail(LY);
  static constexpr int NKB = TAPS * CB, NTL = TAPS * TAIL;
  static constexpr int RS = STAGE == 0 ? TF_RSX : TF_RSB;
  f32x4 wa[NKB], wb[FO ? NKB : 1];
  float ta[NTL ? NTL : 1], tb[(FO && NTL) ? NTL : 1];
  f32x4 bias;

  __device__ __forceinline__ void init(const TfArgs& A, int s, int lane) {
    const int i16 = lane & 15, g4 = lane >> 4;
    const LayerDesc ly = A.layers[LY];
    const float* pa = (const float*)A.ws.a_hi + A.ws.slot_stride_a * s + ly.w_off + (long)(NT * 16 + i16) * ly.KP;
    const float* pb = (const float*)A.ws.b + A.ws.slot_stride_b * s + ly.w_off + (long)(NT * 16 + i16) * ly.KP;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int o = tap * tf_cimg(LY) + cb * 16 + g4 * 4;
        wa[tap * CB + cb] = *(const f32x4*)(pa + o);
        if constexpr (FO) wb[tap * CB + cb] = *(const f32x4*)(pb + o);
      }
      if constexpr (TAIL) {
        ta[tap] = pa[tap * tf_cimg(LY) + 16 + g4];
        if constexpr (FO) tb[tap] = pb[tap * tf_cimg(LY) + 16 + g4];
      }
    }
    const int chb = NT * 16 + 4 * g4;
    const int nv = tl_cout(LY) - chb;
    const float* ba = A.ws.bias_a + (long)A.ws.bias_stride_a * s + ly.bias_off + chb;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = r < nv ? ba[r] : 0.f;
  }

  // k = index of the window inside this workgroup's list; R0 = first row of the window in the [S*B*L] row space
  __device__ __forceinline__ void run(const TfArgs& A, char* smem, int k, unsigned R0, int lane) const {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int par = k & 1;
    const char* in;
    if constexpr (STAGE == 0) in = smem + TF_O_X + par * 2 * TF_PX + (tl_pool(LY) ? TF_PX : 0);
    else if constexpr (STAGE == 1) in = smem + TF_A1B + par * 2 * TF_PA + (tl_pool(LY) ? TF_PA : 0);
    else in = smem + TF_O_MID + par * TF_PB;
    const char* lr = in + i16 * RS;
    const char* lb = lr + g4 * 16;
    const uint32_t* sg = (const uint32_t*)(smem + TF_O_SGN) + (k & 3) * 80 + LY * 8;
    const uint4* lut = (const uint4*)(smem + TF_O_LUT);
    constexpr int chb0 = NT * 16;
    // one accumulator per m-tile: out = bias + W_mu x  (+ Flipout: (s_out o dW o s_in) x, both signs folded into the dW fragment)
    f32x4 acc[2] = {bias, bias};
    uint32_t so = 0;
    if constexpr (FO) so = (sg[4 + (chb0 >> 5)] >> ((chb0 & 31) + i16)) & 1u;   // s_out of this lane's fragment row (cout)
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int kb = tap * CB + cb;
        f32x4 x[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) x[mt] = *(const f32x4*)(lb + (mt * 16 + tap - PAD + HALO) * RS + (tf_inch(LY) + cb * 16) * 4);
        f32x4 wbm = wa[kb];
        if constexpr (FO) {
          // s_in of the layer's own input channels cb*16 + 4 g4 .. + 3
          const uint32_t nib = (sg[(cb * 16) >> 5] >> (((cb * 16) & 31) + 4 * g4)) & 15u;
          wbm = xor4(wb[kb], lut[(so << 4) | nib]);
        }
        // consecutive MFMAs alternate between the two accumulators (dependent issue distance 64 cycles > 40 latency)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wa[kb][j], x[mt][j], acc[mt]);
        if constexpr (FO) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wbm[j], x[mt][j], acc[mt]);
        }
      }
      if constexpr (TAIL) {
        float xt[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xt[mt] = *(const float*)(lr + (mt * 16 + tap - PAD + HALO) * RS + (16 + g4) * 4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(ta[tap], xt[mt], acc[mt]);
        if constexpr (FO) {
          const uint32_t m = (((sg[0] >> (16 + g4)) & 1u) ^ so) << 31;
          const float tbm = xor1(tb[tap], m);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(tbm, xt[mt], acc[mt]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- epilogue: ReLU (every conv of the trunk is followed by one: inception.py:48-60, 118-131) ----------------
    const int chb = chb0 + 4 * g4;
    f32x4 v[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[mt][r] = fmaxf(acc[mt][r], 0.f);
    const int L = A.L;
    constexpr int OOFF = tl_ooff(LY), OUTK = tl_outk(LY);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      if (row < L) {
        if constexpr (OUTK == 0 || OUTK == 1) {
          char* img = smem + (OUTK == 0 ? TF_A1B + par * 2 * TF_PA : TF_O_MID + par * TF_PB);
          *(f32x4*)(img + (row + HALO) * TF_RSB + (OOFF + chb) * 4) = v[mt];
          if constexpr (TRAIN) {
            char* g = (char*)(OUTK == 0 ? A.act1 : A.mid);
            *(f32x4*)(g + ((R0 + (unsigned)row) * 512u + (unsigned)((OOFF + chb) * 4))) = v[mt];
            const uint32_t bits = (v[mt][0] > 0.f ? 1u : 0u) | (v[mt][1] > 0.f ? 2u : 0u) | (v[mt][2] > 0.f ? 4u : 0u) | (v[mt][3] > 0.f ? 8u : 0u);
            (OUTK == 0 ? A.m_act1 : A.m_mid)[(R0 + (unsigned)row) * 32u + (unsigned)((OOFF + chb) >> 2)] = (unsigned char)bits;
          }
        } else {
          *(f32x4*)((char*)A.act2 + ((R0 + (unsigned)row) * 320u + (unsigned)((OOFF + chb) * 4))) = v[mt];
        }
      }
    }
    // ---------------- block 1 only: MaxPool1d(3,1,1) of the output rows (inception.py:99-104 reads it) ----------------
    if constexpr (OUTK == 0) {
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
        if (row < L) {
          char* img = smem + TF_A1B + par * 2 * TF_PA + TF_PA;
          *(f32x4*)(img + (row + HALO) * TF_RSB + (OOFF + chb) * 4) = p[mt];
          if constexpr (TRAIN) A.amax[(R0 + (unsigned)row) * 32u + (unsigned)((OOFF + chb) >> 2)] = (unsigned char)code[mt];
        }
      }
    }
  }
};

template <int EM, bool TRAIN>
struct TfJobRun<EM, TRAIN, TNone> {
  __device__ __forceinline__ void init(const TfArgs&, int, int) {}
  __device__ __forceinline__ void run(const TfArgs&, char*, int, unsigned, int) const {}
};

// x planes (and Flipout sign words) of the next windows: global -> registers (one step ahead) -> LDS
template <bool FO>
struct TfLoader {
  const char* base;
  long wstep;
  int soff[5], dst[5];
  bool on[5];
  tf_u32x4 b[5];   // ext_vector registers (arrays of HIP's uint4 struct are not split into registers)
  const uint32_t* sg0 = nullptr;
  const uint32_t* sg1 = nullptr;
  long sst0 = 0, sst1 = 0;
  uint32_t sb0 = 0, sb1 = 0;

  __device__ __forceinline__ void setup(const TfArgs& A, int s, int split, int lane) {
    const int L = A.L;
    const int nch = L * 5, tot = 2 * nch;   // 16-byte chunks of one plane / both planes of one window
    const long pstride = (const char*)A.xp[1] - (const char*)A.xp[0];
    base = (const char*)A.xp[0] + (long)split * L * (TF_XC * 4);
    wstep = (long)A.nsplit * L * (TF_XC * 4);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int q = j * 64 + lane;
      on[j] = q < tot;
      const int qq = on[j] ? q : 0;
      const int pl = qq / nch, c = qq - pl * nch;
      soff[j] = (int)(pl * pstride) + c * 16;
      dst[j] = pl * TF_PX + (c / 5 + HALO) * TF_RSX + (c % 5) * 16;
    }
    if constexpr (FO) {
      // lane -> (layer = lane >> 3, word k = lane & 7) for layers 0..7; lanes 0..15 also layers 8, 9.  Words 0..3: s_in, 4..5: s_out
      auto one = [&](int layer, int kk, const uint32_t*& p, long& stride) {
        const LayerDesc ly = A.layers[layer];
        if (kk < 4 && kk < ly.sign_in_words) {
          p = A.sign_in + ly.sign_in_off * A.examples + ((long)s * A.B + split) * ly.sign_in_words + kk;
          stride = (long)A.nsplit * ly.sign_in_words;
        } else if (kk >= 4 && kk - 4 < ly.sign_out_words && kk < 6) {
          p = A.sign_out + ly.sign_out_off * A.examples + ((long)s * A.B + split) * ly.sign_out_words + (kk - 4);
          stride = (long)A.nsplit * ly.sign_out_words;
        }
      };
      one(lane >> 3, lane & 7, sg0, sst0);
      if (lane < 16) one(8 + (lane >> 3), lane & 7, sg1, sst1);
    }
  }
  __device__ __forceinline__ void fetch() {
#pragma unroll
    for (int j = 0; j < 5; ++j) b[j] = *(const tf_u32x4*)(base + soff[j]);
    base += wstep;
    if constexpr (FO) {
      if (sg0) { sb0 = *sg0; sg0 += sst0; }
      if (sg1) { sb1 = *sg1; sg1 += sst1; }
    }
  }
  __device__ __forceinline__ void put(char* smem, int k, int lane) {
    char* xs = smem + TF_O_X + (k & 1) * 2 * TF_PX;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (on[j]) *(tf_u32x4*)(xs + dst[j]) = b[j];
    if constexpr (FO) {
      uint32_t* sgw = (uint32_t*)(smem + TF_O_SGN) + (k & 3) * 80;
      sgw[lane] = sb0;
      if (lane < 16) sgw[64 + lane] = sb1;
    }
  }
};

template <class J>
__device__ __forceinline__ constexpr int tfj_stage() {
  if constexpr (J::layer < 0) return 0;
  else return tl_stage(J::layer);
}

template <int EM, bool TRAIN, bool LOADER, class J0, class J1, class J2>
__device__ __forceinline__ void tf_role(const TfArgs& A, char* smem, int s, int split, int nwin, int lane) {
  TfJobRun<EM, TRAIN, J0> r0;
  TfJobRun<EM, TRAIN, J1> r1;
  TfJobRun<EM, TRAIN, J2> r2;
  r0.init(A, s, lane);
  r1.init(A, s, lane);
  r2.init(A, s, lane);
  TfLoader<EM == EM_FLIPOUT> ld;
  if constexpr (LOADER) {
    ld.setup(A, s, split, lane);
    if (nwin > 0) ld.fetch();
  }
  __syncthreads();   // zero fill + sign table
  if constexpr (LOADER) {
    if (nwin > 0) ld.put(smem, 0, lane);
    if (nwin > 1) ld.fetch();
  }
  lds_barrier();     // window 0 staged
  const int nsteps = nwin + 2;
  // row indices fit 32 bits (the host refuses launches whose planes exceed 4 GiB)
  const unsigned Rs = (unsigned)(((long)s * A.B + split) * A.L), Rstep = (unsigned)(A.nsplit * A.L);
  for (int t = 0; t < nsteps; ++t) {
    if constexpr (LOADER) {
      if (t + 1 < nwin) ld.put(smem, t + 1, lane);
      if (t + 2 < nwin) ld.fetch();
    }
    {
      const int k = t - tfj_stage<J0>();
      if (J0::layer >= 0 && k >= 0 && k < nwin) r0.run(A, smem, k, Rs + k * Rstep, lane);
    }
    {
      const int k = t - tfj_stage<J1>();
      if (J1::layer >= 0 && k >= 0 && k < nwin) r1.run(A, smem, k, Rs + k * Rstep, lane);
    }
    {
      const int k = t - tfj_stage<J2>();
      if (J2::layer >= 0 && k >= 0 && k < nwin) r2.run(A, smem, k, Rs + k * Rstep, lane);
    }
    lds_barrier();
  }
}

template <int EM, bool TRAIN>
__global__ __launch_bounds__(TF_THREADS) void tf_fwd_kernel(const TfArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TF_O_LUT / 4; k += TF_THREADS) z[k] = 0u;
    build_sign_lut_f32((uint4*)(smem + TF_O_LUT), tid);
  }
#define TF_ROLE(LD, ...) tf_role<EM, TRAIN, LD, __VA_ARGS__>(A, smem, s, split, nwin, lane)
#define TJ(...) TJob<__VA_ARGS__>
  // MFMAs per window with Flipout (plain: half): wave w and w + 4 share SIMD w
  switch (wave) {
    case 0: TF_ROLE(false, TJ(8, 0), TNone, TNone); break;               // k5 64->16: 320
    case 4: TF_ROLE(false, TJ(5, 0), TJ(5, 1), TJ(0, 0)); break;         // 128 + 128 + 20
    case 1: TF_ROLE(false, TJ(6, 0), TJ(2, 0), TNone); break;            // 192 + 100
    case 5: TF_ROLE(false, TJ(5, 2), TJ(5, 3), TJ(1, 0)); break;         // 128 + 128 + 60
    case 2: TF_ROLE(false, TJ(7, 0), TJ(7, 1), TJ(1, 1)); break;         // 128 + 128 + 60
    case 6: TF_ROLE(true, TJ(7, 2), TJ(7, 3), TJ(0, 1)); break;          // 128 + 128 + 20, + the loader
    case 3: TF_ROLE(false, TJ(9, 0), TJ(9, 1), TJ(3, 0)); break;         // 128 + 128 + 60
    default: TF_ROLE(false, TJ(4, 0), TJ(2, 1), TJ(3, 1)); break;        // 128 + 100 + 60
  }
#undef TJ
#undef TF_ROLE
}

// fp32 planes [rows][20] of the raw windows and of their MaxPool1d(3,1,1) copy (block 1's pooled branch,
// inception.py:41-46); channels 18, 19 zero.  rows = B * L, pooling stays inside a window.  One thread per 4 channels.
__global__ void xf_planes_kernel(const float* x, float* xp, float* xpp, long rows, int L, int F) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * 5) return;
  const long r = idx / 5;
  const int c0 = (int)(idx - r * 5) * 4;
  const int l = (int)(r % L);
  const bool up = l > 0, dn = l + 1 < L;
  f32x4 v, q;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + k;
    const bool on = c < F;
    const int cc = on ? c : 0;
    const float a = x[r * F + cc];
    const float b = x[(up ? r - 1 : r) * F + cc], d = x[(dn ? r + 1 : r) * F + cc];
    v[k] = on ? a : 0.f;
    q[k] = on ? fmaxf(fmaxf(a, b), d) : 0.f;   // rows outside the window are replaced by the row itself
  }
  *(f32x4*)(xp + r * TF_XC + c0) = v;
  *(f32x4*)(xpp + r * TF_XC + c0) = q;
}

// ==========================================================================================
// tf_dx_kernel : d loss / d (pre-activation) of block 2's 1x1 level (MID) and of block 1 (ACT1) in ONE launch (exact fp32).
//   Two-stage pipeline inside the workgroup, one barrier per step (the structure of trunk_dx_kernel):
//     stage A : dMID  = W6^T dz6 + W8^T dz8 (k3 / k5 level), masked with [MID > 0]            window t
//     stage B : dACT1 = W4^T dz4 + W5^T dz5 + W7^T dz7 + scatter(W9^T dz9), masked             window t-1
//   dz(MID) goes from stage A to stage B through LDS; both masked gradients are written once to HBM for the dW kernels.
//   Weights: transposed + tap-flipped images, register-stationary.  Flipout: BOTH sign vectors are folded into the dW^T
//   fragment (row = input channel -> s_in bit, K = cout -> s_out nibble; the forward's table), so both contractions
//   share the dz operand and the accumulator.  Wave w owns tile w of stage B (176 MFMAs per window with Flipout) and one
//   tile of stage A (layer 6: 48, waves 0..3; layer 8: 80, waves 4..7): every SIMD carries 480.  Waves 0 and 1 also stage
//   the next window's dY(ACT2) (global -> registers, one step ahead -> LDS), masking it with [ACT2 > 0] unless PRE.
// ==========================================================================================
enum {
  TD_RS2 = 80 * 4 + 32,                // 352 bytes: pitch 22
  TD_P2 = IMG_ROWS * TD_RS2,           // 12,672: dz(ACT2) image
  TD_PM = IMG_ROWS * TF_RSB,           // 19,584: dz(MID) image
  TD_O_DZ2 = 0,                        // [3 slots]
  TD_O_DZM = 3 * TD_P2,                // [2 bufs]
  TD_O_SGN = TD_O_DZM + 2 * TD_PM,     // [3 slots][80 words]
  TD_O_LUT = TD_O_SGN + 3 * 80 * 4,
  TD_LDS = TD_O_LUT + 32 * 16
};

struct TfDxArgs {
  const float* g_act2;           // [S*B*L][80] dY of ACT2 (from the dense layer's dX)
  const float* act2;             // [S*B*L][80] (read unless PRE)
  const unsigned char* m_mid;    // [S*B*L][32] nibble masks [MID > 0]
  const unsigned char* m_act1;   // [S*B*L][32]
  const unsigned char* amax;     // [S*B*L][32] 2-bit arg-max codes
  float* g_mid;                  // [S*B*L][128]  out: dz of block 2's 1x1 outputs (masked)
  float* g_act1;                 // [S*B*L][128]  out: dz of block 1's outputs (masked)
  float* g_act2m;                // [S*B*L][80]   out (unless PRE): dY(ACT2) masked with [ACT2 > 0] (dW operand); may alias g_act2
  WeightSlots ws;
  const LayerDesc* layers;
  const uint32_t* sign_in;
  const uint32_t* sign_out;
  long examples;
  int S, B, L, nsplit;
};

__device__ __forceinline__ f32x4 mask4(f32x4 v, uint32_t nib) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = ((nib >> r) & 1u) ? v[r] : 0.f;
  return o;
}

// stage A tile J of layer LY (6: k3, MID channels 0..63, dz = ACT2 channels 16..31; 8: k5, 64..127, 32..47)
template <int EM, int LY, int J>
struct TdJobA {
  static constexpr bool FO = (EM == EM_FLIPOUT);
  static constexpr int TAPS = tl_taps(LY), PAD = (TAPS - 1) / 2, CH0 = LY == 6 ? 16 : 32, MCH = (LY == 6 ? 0 : 64) + J * 16;
  f32x4 wa[TAPS], wb[FO ? TAPS : 1];
  __device__ __forceinline__ void init(const TfDxArgs& A, int s, int lane) {
    const int i16 = lane & 15, g4 = lane >> 4;
    const LayerDesc ly = A.layers[LY];
    const long ro = ly.wt_off + (long)(J * 16 + i16) * ly.KPt + 4 * g4;
    const float* pa = (const float*)A.ws.at + A.ws.slott_stride_a * s + ro;
    const float* pb = (const float*)A.ws.bt + A.ws.slott_stride_b * s + ro;
#pragma unroll
    for (int tf = 0; tf < TAPS; ++tf) {
      wa[tf] = *(const f32x4*)(pa + tf * 16);
      if constexpr (FO) wb[tf] = *(const f32x4*)(pb + tf * 16);
    }
  }
  __device__ __forceinline__ void run(const TfDxArgs& A, char* smem, int k, unsigned R0, int lane) const {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int L = A.L;
    const char* sl = smem + TD_O_DZ2 + (k % 3) * TD_P2;
    const uint32_t* sg = (const uint32_t*)(smem + TD_O_SGN) + (k % 3) * 80 + LY * 8;
    char* dzm = smem + TD_O_DZM + (k & 1) * TD_PM;
    const int och = MCH + 4 * g4;   // MID channel of this lane's 4 outputs
    // the ReLU masks of the tile's outputs are fetched ahead of the MFMAs
    uint32_t mb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = min(mt * 16 + i16, L - 1);
      mb[mt] = A.m_mid[(R0 + (unsigned)row) * 32u + (unsigned)(och >> 2)];
    }
    uint4 fm = make_uint4(0, 0, 0, 0);
    if constexpr (FO) {
      const int ci = J * 16 + i16;   // input channel of the layer = row of the transposed fragment
      const uint32_t si = (sg[ci >> 5] >> (ci & 31)) & 1u;
      const uint32_t nib = (sg[4] >> (4 * g4)) & 15u;
      fm = ((const uint4*)(smem + TD_O_LUT))[(si << 4) | nib];
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* lb = sl + i16 * TD_RS2 + (CH0 + 4 * g4) * 4;
#pragma unroll
    for (int tf = 0; tf < TAPS; ++tf) {
      f32x4 x[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) x[mt] = *(const f32x4*)(lb + (mt * 16 + tf - PAD + HALO) * TD_RS2);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wa[tf][j], x[mt][j], acc[mt]);
      if constexpr (FO) {
        const f32x4 wbm = xor4(wb[tf], fm);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wbm[j], x[mt][j], acc[mt]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      if (row < L) {
        const f32x4 d = mask4(acc[mt], mb[mt]);
        *(f32x4*)(dzm + (row + HALO) * TF_RSB + och * 4) = d;
        *(f32x4*)((char*)A.g_mid + ((R0 + (unsigned)row) * 512u + (unsigned)(och * 4))) = d;
      }
    }
  }
};

// stage B tile CT (ACT1 channels CT*16 .. +15): k-blocks 0: layer 4, 1-4: layer 5, 5-8: layer 7, 9-10: layer 9 (pooled)
template <int EM, int CT>
struct TdJobB {
  static constexpr bool FO = (EM == EM_FLIPOUT);
  f32x4 wa[11], wb[FO ? 11 : 1];
  __host__ __device__ static constexpr int q_layer(int q) { return q == 0 ? 4 : (q < 5 ? 5 : (q < 9 ? 7 : 9)); }
  __host__ __device__ static constexpr int q_kb(int q) { return q == 0 ? 0 : (q < 5 ? q - 1 : (q < 9 ? q - 5 : q - 9)); }
  __device__ __forceinline__ void init(const TfDxArgs& A, int s, int lane) {
    const int i16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      const LayerDesc ly = A.layers[q_layer(q)];
      const long ro = ly.wt_off + (long)(CT * 16 + i16) * ly.KPt + q_kb(q) * 16 + 4 * g4;
      wa[q] = *(const f32x4*)((const float*)A.ws.at + A.ws.slott_stride_a * s + ro);
      if constexpr (FO) wb[q] = *(const f32x4*)((const float*)A.ws.bt + A.ws.slott_stride_b * s + ro);
    }
  }
  __device__ __forceinline__ void run(const TfDxArgs& A, char* smem, int k, unsigned R0, int lane) const {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int L = A.L;
    const char* sl = smem + TD_O_DZ2 + (k % 3) * TD_P2;
    const uint32_t* sg = (const uint32_t*)(smem + TD_O_SGN) + (k % 3) * 80;
    const char* dzm = smem + TD_O_DZM + (k & 1) * TD_PM;
    const uint4* lut = (const uint4*)(smem + TD_O_LUT);
    const int och = CT * 16 + 4 * g4, ci = CT * 16 + i16;
    uint32_t mb[2], code[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      const unsigned o = (R0 + (unsigned)min(row, L - 1)) * 32u + (unsigned)(och >> 2);
      mb[mt] = A.m_act1[o];
      code[mt] = row < L ? (uint32_t)A.amax[o] : 0x55u;
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 accp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* b2 = sl + (i16 + HALO) * TD_RS2 + g4 * 16;
    const char* bm = dzm + (i16 + HALO) * TF_RSB + g4 * 16;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      constexpr int dummy = 0;
      (void)dummy;
      f32x4 x[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const char* bp;
        if (q == 0) bp = b2 + mt * 16 * TD_RS2;
        else if (q >= 9) bp = b2 + mt * 16 * TD_RS2 + (48 + (q - 9) * 16) * 4;
        else bp = bm + mt * 16 * TF_RSB + (q - 1) * 64;
        x[mt] = *(const f32x4*)bp;
      }
      f32x4 wbm = wa[q];
      if constexpr (FO) {
        const int ly = q_layer(q);
        const uint32_t si = (sg[ly * 8 + (ci >> 5)] >> (ci & 31)) & 1u;
        const int bit = q_kb(q) * 16;
        const uint32_t nib = (sg[ly * 8 + 4 + (bit >> 5)] >> ((bit & 31) + 4 * g4)) & 15u;
        wbm = xor4(wb[q], lut[(si << 4) | nib]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4& ta = q >= 9 ? accp[mt] : acc[mt];
          ta = mfma4(wa[q][j], x[mt][j], ta);
        }
      if constexpr (FO) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x4& ta = q >= 9 ? accp[mt] : acc[mt];
            ta = mfma4(wbm[j], x[mt][j], ta);
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
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
      if (row < L) *(f32x4*)((char*)A.g_act1 + ((R0 + (unsigned)row) * 512u + (unsigned)(och * 4))) = mask4(v[mt], mb[mt]);
    }
  }
};

// dY(ACT2) of the next windows (and the Flipout sign words): loader wave p of 2 takes the chunks (2 j + p) * 64 + lane
template <bool FO, bool PRE>
struct TdLoader {
  int qo[5], dst[5];
  bool on[5];
  tf_u32x4 g[5], y[PRE ? 1 : 5];
  const uint32_t* sg0 = nullptr;
  const uint32_t* sg1 = nullptr;
  long sst0 = 0, sst1 = 0;
  uint32_t sb0 = 0, sb1 = 0;
  long Rs, Rstep;
  int p;
  __device__ __forceinline__ void setup(const TfDxArgs& A, int s, int split, int lane, int p_) {
    p = p_;
    const int n2 = A.L * 20;   // 16-byte chunks of an 80-channel fp32 window
    Rs = ((long)s * A.B + split) * A.L;
    Rstep = (long)A.nsplit * A.L;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int q = (2 * j + p) * 64 + lane;
      on[j] = q < n2;
      const int qq = on[j] ? q : 0;
      qo[j] = qq * 16;
      dst[j] = (qq / 20 + HALO) * TD_RS2 + (qq % 20) * 16;
    }
    if constexpr (FO) {
      if (p == 0) {
        auto one = [&](int layer, int kk, const uint32_t*& q, long& stride) {
          const LayerDesc ly = A.layers[layer];
          if (kk < 4 && kk < ly.sign_in_words) {
            q = A.sign_in + ly.sign_in_off * A.examples + ((long)s * A.B + split) * ly.sign_in_words + kk;
            stride = (long)A.nsplit * ly.sign_in_words;
          } else if (kk >= 4 && kk - 4 < ly.sign_out_words && kk < 6) {
            q = A.sign_out + ly.sign_out_off * A.examples + ((long)s * A.B + split) * ly.sign_out_words + (kk - 4);
            stride = (long)A.nsplit * ly.sign_out_words;
          }
        };
        one(lane >> 3, lane & 7, sg0, sst0);
        if (lane < 16) one(8 + (lane >> 3), lane & 7, sg1, sst1);
      }
    }
  }
  __device__ __forceinline__ void fetch(const TfDxArgs& A, int k) {
    const char* gp = (const char*)A.g_act2 + (Rs + k * Rstep) * 320;
#pragma unroll
    for (int j = 0; j < 5; ++j) g[j] = *(const tf_u32x4*)(gp + qo[j]);
    if constexpr (!PRE) {
      const char* yp = (const char*)A.act2 + (Rs + k * Rstep) * 320;
#pragma unroll
      for (int j = 0; j < 5; ++j) y[j] = *(const tf_u32x4*)(yp + qo[j]);
    }
    if constexpr (FO) {
      if (sg0) sb0 = sg0[(long)k * sst0];
      if (sg1) sb1 = sg1[(long)k * sst1];
    }
  }
  __device__ __forceinline__ void put(const TfDxArgs& A, char* smem, int k, int lane) {
    char* sl = smem + TD_O_DZ2 + (k % 3) * TD_P2;
    char* gd = (char*)A.g_act2m + (Rs + k * Rstep) * 320;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (on[j]) {
        tf_u32x4 d = g[j];
        if constexpr (!PRE) {
          const f32x4 yy = __builtin_bit_cast(f32x4, y[j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = yy[e] > 0.f ? d[e] : 0u;
          *(tf_u32x4*)(gd + qo[j]) = d;   // the dW kernels read dz(ACT2) from HBM: masked once, here
        }
        *(tf_u32x4*)(sl + dst[j]) = d;
      }
    }
    if constexpr (FO) {
      if (p == 0) {
        uint32_t* sgw = (uint32_t*)(smem + TD_O_SGN) + (k % 3) * 80;
        sgw[lane] = sb0;
        if (lane < 16) sgw[64 + lane] = sb1;
      }
    }
  }
};

template <int EM, bool PRE, int W>
__device__ __forceinline__ void td_role(const TfDxArgs& A, char* smem, int s, int split, int nwin, int lane) {
  constexpr bool FO = (EM == EM_FLIPOUT);
  TdJobB<EM, W> jb;
  TdJobA<EM, (W < 4 ? 6 : 8), (W & 3)> ja;
  jb.init(A, s, lane);
  ja.init(A, s, lane);
  constexpr bool LOADER = W < 2;
  TdLoader<FO, PRE> ld;
  if constexpr (LOADER) {
    ld.setup(A, s, split, lane, W);
    if (nwin > 0) ld.fetch(A, 0);
  }
  __syncthreads();   // zero fill + sign table
  if constexpr (LOADER) {
    if (nwin > 0) ld.put(A, smem, 0, lane);
    if (nwin > 1) ld.fetch(A, 1);
  }
  lds_barrier();
  const unsigned Rs = (unsigned)(((long)s * A.B + split) * A.L), Rstep = (unsigned)(A.nsplit * A.L);
  const int nsteps = nwin + 1;
  for (int t = 0; t < nsteps; ++t) {
    if constexpr (LOADER) {
      if (t + 1 < nwin) ld.put(A, smem, t + 1, lane);
      if (t + 2 < nwin) ld.fetch(A, t + 2);
    }
    if (t < nwin) ja.run(A, smem, t, Rs + t * Rstep, lane);
    if (t >= 1) jb.run(A, smem, t - 1, Rs + (t - 1) * Rstep, lane);
    lds_barrier();
  }
}

template <int EM, bool PRE>
__global__ __launch_bounds__(TF_THREADS) void tf_dx_kernel(const TfDxArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TD_O_LUT / 4; k += TF_THREADS) z[k] = 0u;
    build_sign_lut_f32((uint4*)(smem + TD_O_LUT), tid);
  }
  switch (wave) {
    case 0: td_role<EM, PRE, 0>(A, smem, s, split, nwin, lane); break;
    case 1: td_role<EM, PRE, 1>(A, smem, s, split, nwin, lane); break;
    case 2: td_role<EM, PRE, 2>(A, smem, s, split, nwin, lane); break;
    case 3: td_role<EM, PRE, 3>(A, smem, s, split, nwin, lane); break;
    case 4: td_role<EM, PRE, 4>(A, smem, s, split, nwin, lane); break;
    case 5: td_role<EM, PRE, 5>(A, smem, s, split, nwin, lane); break;
    case 6: td_role<EM, PRE, 6>(A, smem, s, split, nwin, lane); break;
    default: td_role<EM, PRE, 7>(A, smem, s, split, nwin, lane); break;
  }
}
