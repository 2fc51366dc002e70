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
// diagnostics builds only (tests/probes/ablate_gpu.sh, results wrong): TFV bit 1 = dW without the fold arithmetic, 2 = forward
// without epilogue, 4 = dW without its DMAs, 8 = forward without LDS operand reads in the k loop, 32 = forward without global
// stores, 64 = forward without max-pooling; the product library is built with TFV == 0.
#ifndef TF_STAMPS
#define TF_STAMPS 0   // diagnostics builds (tests/probes/ablate_gpu.sh): s_memtime stamps of workgroup 0's waves
#endif
#ifndef TFV
#define TFV 0
#endif
#include "kernels_misc.h"
#include "kernels_trunk_dw.h"   // kernels_trunk.h (tl_* tables, rot16, lds_barrier) and what kernels_dense_ks.h needs
#include "kernels_dense_ks.h"   // HL_ROWS, DenseKsFinArgs

enum { TF_WAVES = 8, TF_THREADS = TF_WAVES * 64 };

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): the workgroups with equal blockIdx % 8 share an
// XCD.  This bijection hands each of those groups a CONTIGUOUS range of logical ids, so that the workgroups of one
// particle (the same weight fragments: 0.3 MB per workgroup of the trunk forward) fill ONE L2 instead of all eight.
// Speed only - nothing depends on the placement.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
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
  TF_O_LUT = TF_O_SGN + 4 * 80 * 4,    // 32 x 16 B: sign-mask table of the first version (the folds are register arithmetic now; kept as a layout slot)
  TF_O_WB = TF_O_LUT + 32 * 16,        // LRT: sigma^2 fragments of the one job too wide for the register file: [20 k-blocks][64 lanes][16 B]
  TF_LDS = TF_O_WB + 20 * 1024
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
  unsigned char* m_act2;     // [S*B*L][20] [ACT2 > 0] (the dense layer's dX applies it)
  // MC-dropout of the frequentist sibling (DROP instantiations only; nets/inception.py:48-52,119-123: nn.Dropout(p / 4) behind
  // every branch of both blocks): an element is kept with probability 1 - rate and scaled by 1 / (1 - rate)
  float drop_rate, drop_scale;
  uint64_t drop_seed;
  uint32_t drop_step;
  const float* keep1;        // injected keep masks (1 keep / 0 drop) [B*L][128], [B*L][80]; null: Philox
  const float* keep2;
  // local reparameterisation (EM_LRT; [3P] tyxe.poutine.local_reparameterization, bayesian.py:66-67): slot A = mu, slot B =
  // sigma^2 (shared by the particles), out = loc + sqrt(var) eps with var = sigma^2 . x^2; eps from Philox or injected
  NoiseRefs nz;
  CallGeom cg;
  float* q1; float* qm; float* q2;   // q = eps / (2 sd) of ACT1 / MID / ACT2 outputs (training step: the backward's d out / d var)
  int S, B, L, nsplit;
#if TF_STAMPS
  unsigned long long* dbg;   // [8 waves][48 steps][16 phases]
#endif
};
#if TF_STAMPS
#define TF_STAMP_AT(P, STEP, PH)                                                                                              \
  do {                                                                                                                        \
    if ((P) && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (STEP) < 48)                                                      \
      (P)[(((threadIdx.x >> 6) * 48 + (STEP)) * 16) + (PH)] = __builtin_amdgcn_s_memtime();                                   \
  } while (0)
#else
#define TF_STAMP_AT(P, STEP, PH) do {} while (0)
#endif

// keep mask of 4 consecutive channels of one row: injected floats, or one Philox call (4 x 32 uniform bits)
__device__ __forceinline__ uint32_t drop_keep4(const float* inj, long o, float rate, uint32_t row, uint32_t quad, uint32_t which,
                                               uint32_t step, uint64_t seed) {
  if (inj) {
    const f32x4 k = *(const f32x4*)(inj + o);
    return (k[0] > 0.5f ? 1u : 0u) | (k[1] > 0.5f ? 2u : 0u) | (k[2] > 0.5f ? 4u : 0u) | (k[3] > 0.5f ? 8u : 0u);
  }
  const uint4 u = philox4x32_10(row, quad, NK_DROPOUT | (which << 8), step, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t thr = (uint32_t)fminf(rate * 4294967296.f, 4294967040.f);   // drop iff u < rate * 2^32
  return (u.x >= thr ? 1u : 0u) | (u.y >= thr ? 2u : 0u) | (u.z >= thr ? 4u : 0u) | (u.w >= thr ? 8u : 0u);
}

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
template <int EM, bool TRAIN, class J, bool DROP = false>
struct TfJobRun {
  static constexpr int LY = J::layer, NT = J::nt;
  static constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT), TWO = FO || LRT;   // TWO: a second weight operand (dW | sigma^2)
  static constexpr int STAGE = tl_stage(LY), TAPS = tl_taps(LY), PAD = (TAPS - 1) / 2, CB = tf_cb(LY), TAIL = tf_tail(LY);
  static constexpr int NKB = TAPS * CB, NTL = TAPS * TAIL;
  static constexpr int RS = STAGE == 0 ? TF_RSX : TF_RSB;
  static constexpr bool WBL = J::wl_lds && LRT;   // the second operand's fragments live in LDS (lane-linear: conflict-free)
  f32x4 wa[NKB], wb[(TWO && !WBL) ? NKB : 1];
  float ta[NTL ? NTL : 1], tb[(TWO && NTL) ? NTL : 1];
  f32x4 bias, biasv;

  __device__ __forceinline__ void init(const TfArgs& A, char* smem, int s, int lane) {
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
        if constexpr (WBL) *(f32x4*)(smem + TF_O_WB + (tap * CB + cb) * 1024 + lane * 16) = *(const f32x4*)(pb + o);
        else if constexpr (TWO) wb[tap * CB + cb] = *(const f32x4*)(pb + o);
      }
      if constexpr (TAIL) {
        ta[tap] = pa[tap * tf_cimg(LY) + 16 + g4];
        if constexpr (TWO) tb[tap] = pb[tap * tf_cimg(LY) + 16 + g4];
      }
    }
    const int chb = NT * 16 + 4 * g4;
    const int nv = tl_cout(LY) - chb;
    const float* ba = A.ws.bias_a + (long)A.ws.bias_stride_a * s + ly.bias_off + chb;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = r < nv ? ba[r] : 0.f;
    biasv = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LRT) {
      const float* bb = A.ws.bias_b + ly.bias_off + chb;   // sigma_b^2
#pragma unroll
      for (int r = 0; r < 4; ++r) biasv[r] = r < nv ? bb[r] : 0.f;
    }
  }

  // k = index of the window inside this workgroup's list; R0 = first row of the window in the [S*B*L] row space
  __device__ __forceinline__ void run(const TfArgs& A, char* smem, int k, unsigned R0, int lane, int st_step = 0, int st_ph = 0) const {
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
    constexpr int chb0 = NT * 16;
    // one accumulator per m-tile: out = bias + W_mu x  (+ Flipout: (s_out o dW o s_in) x, both signs folded into the dW fragment)
    f32x4 acc[2] = {bias, bias};
    f32x4 accv[2] = {biasv, biasv};   // LRT: var = sigma_b^2 + sigma_W^2 . x^2
    // the window's sign words of this layer, once per job, with s_out of this lane's fragment row (cout) folded in and
    // pre-shifted per lane: the sign of element j of k-block cb is bit 28 + j of sl0 / sl16 [word] (the block starts at bit 0
    // or 16 of its word), so a fold is one constant shift + one bit operation (w ^ (shifted & 0x80000000)) per element.
    // (A 32-entry LDS table of masks cost a dependent address chain + an LDS read per k-block in front of its MFMAs: one
    // wave then issued an MFMA every 39.5 cycles instead of every 34.7 - tests/probes/mfma_mix.hip.)
    uint32_t sl0[4] = {0u, 0u, 0u, 0u}, sl16[4] = {0u, 0u, 0u, 0u}, stl = 0;
    if constexpr (FO) {
      const uint32_t so = 0u - ((sg[4 + (chb0 >> 5)] >> ((chb0 & 31) + i16)) & 1u);   // all ones: s_out = -1
#pragma unroll
      for (int w = 0; w < tf_cimg(LY) / 32; ++w) {
        const uint32_t x = sg[w] ^ so;
        sl0[w] = x << (28 - 4 * g4);
        sl16[w] = x << (12 - 4 * g4);
        if (w == 0) stl = x << (15 - g4);   // the tail channel 16 + g4 of a 20-channel input
      }
    }
    // hipcc sinks every LDS read next to its first use (read, wait, 4 MFMAs, read, wait, ...): the operands of k-block
    // kb + 1 are fetched explicitly BEFORE the MFMAs of k-block kb, scheduling barriers keep the two groups apart
    struct Op {
      f32x4 x[2];
      float xt[2];
      f32x4 wbk;
    };
    auto fetch = [&](int kb, Op& o) __attribute__((always_inline)) {
      const int tap = kb / CB, cb = kb - tap * CB;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) o.x[mt] = *(const f32x4*)(lb + (mt * 16 + tap - PAD + HALO) * RS + (tf_inch(LY) + cb * 16) * 4);
      if constexpr (WBL) o.wbk = *(const f32x4*)(smem + TF_O_WB + kb * 1024 + lane * 16);
      if constexpr (TAIL) {
        if (cb == CB - 1) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) o.xt[mt] = *(const float*)(lr + (mt * 16 + tap - PAD + HALO) * RS + (16 + g4) * 4);
        }
      }
    };
    Op cur, nxt;
    fetch(0, cur);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int tap = kb / CB, cb = kb - tap * CB;
      if constexpr (TFV & 8) nxt = cur;   // diagnostics: no LDS operand traffic inside the k loop (results wrong)
      else if (kb + 1 < NKB) fetch(kb + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
#if TF_STAMPS
      if (st_ph == 4 && kb < 7) TF_STAMP_AT(A.dbg, st_step, 9 + kb);
#endif
      // consecutive MFMAs alternate between the two accumulators (dependent issue distance 64 cycles > 40 latency)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wa[kb][j], cur.x[mt][j], acc[mt]);
      if constexpr (FO) {
        // s_in of the layer's own input channels cb*16 + 4 g4 + j (and s_out of the row) into the dW fragment
        const uint32_t sgw = ((cb * 16) & 31) ? sl16[(cb * 16) >> 5] : sl0[(cb * 16) >> 5];
        f32x4 wbm;
#pragma unroll
        for (int j = 0; j < 4; ++j) wbm[j] = xor1(wb[kb][j], (sgw << (3 - j)) & 0x80000000u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wbm[j], cur.x[mt][j], acc[mt]);
        // the folds first, each into its own register, in the shadow of the previous k-block's last MFMAs (hipcc otherwise
        // sinks every fold next to its use)
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      if constexpr (LRT) {
        f32x4 x2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) x2[mt] = cur.x[mt] * cur.x[mt];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) accv[mt] = mfma4(WBL ? cur.wbk[j] : wb[WBL ? 0 : kb][j], x2[mt][j], accv[mt]);
      }
      if constexpr (TAIL) {
        if (cb == CB - 1) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(ta[tap], cur.xt[mt], acc[mt]);
          if constexpr (LRT) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) accv[mt] = mfma4(tb[tap], cur.xt[mt] * cur.xt[mt], accv[mt]);
          }
          if constexpr (FO) {
            const float tbm = xor1(tb[tap], stl & 0x80000000u);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(tbm, cur.xt[mt], acc[mt]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    __builtin_amdgcn_sched_barrier(0);
#if TF_STAMPS
    asm volatile("" ::"v"(acc[0]), "v"(acc[1]));   // (issue of the last MFMA; its result is waited for by the epilogue)
    TF_STAMP_AT(A.dbg, st_step, st_ph);
#endif
    // ---------------- epilogue: ReLU (every conv of the trunk is followed by one: inception.py:48-60, 118-131) ----------------
    if constexpr (TFV & 2) {
      asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
      return;
    }
    if constexpr (LRT) {
      // out = loc + sqrt(var) eps, q = eps / (2 sd) kept for the backward (group_fwd_kernel's epilogue; U5 / U6: a negative
      // variance is replaced by 1e-6).  Pad channels have loc = var = 0: out = 0 whatever eps is.
      constexpr int OOFFq = tl_ooff(LY), OUTKq = tl_outk(LY), CTq = OUTKq == 2 ? 80 : 128;
      const int cout = tl_cout(LY), lch0 = NT * 16 + 4 * g4;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        const unsigned R = R0 + (unsigned)min(row, A.L - 1);
        f32x4 eps;
        if (A.nz.use_philox_lrt) {
          const long Rg = global_row(A.cg, A.L, (int)R);
          const uint64_t idx = (uint64_t)Rg * (uint64_t)(((cout + 15) & ~15) >> 2) + (uint64_t)(lch0 >> 2);
          eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)LY << 8), A.nz.step, A.nz.seed);
        } else {
          const float* e = A.nz.lrt_eps[LY] + (long)R * cout + lch0;
#pragma unroll
          for (int r = 0; r < 4; ++r) eps[r] = (lch0 + r < cout) ? e[r] : 0.f;
        }
        f32x4 qv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float var = accv[mt][r];
          if (var < 0.f) var = 1e-6f;
          const float sd = sqrtf(var);
          acc[mt][r] = acc[mt][r] + sd * eps[r];
          qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
        }
        if constexpr (TRAIN) {
          if (row < A.L) {
            float* qp = OUTKq == 0 ? A.q1 : (OUTKq == 1 ? A.qm : A.q2);
            *(f32x4*)(qp + (long)R * CTq + OOFFq + lch0) = qv;
          }
        }
      }
    }
    const int chb = chb0 + 4 * g4;
    f32x4 v[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[mt][r] = fmaxf(acc[mt][r], 0.f);
    const int L = A.L;
    constexpr int OOFF = tl_ooff(LY), OUTK = tl_outk(LY);
    if constexpr (DROP && OUTK != 1) {
      // nn.Dropout behind the branch's last ReLU (block outputs only: MID is inside a branch); the pooled copy, the ReLU
      // masks and every consumer see the dropped values
      constexpr int CT = OUTK == 0 ? 128 : 80;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const unsigned R = R0 + (unsigned)min(mt * 16 + i16, L - 1);
        const uint32_t kb = drop_keep4(OUTK == 0 ? A.keep1 : A.keep2, (long)R * CT + OOFF + chb, A.drop_rate, R,
                                       (uint32_t)((OOFF + chb) >> 2), (uint32_t)OUTK, A.drop_step, A.drop_seed);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[mt][r] = ((kb >> r) & 1u) ? v[mt][r] * A.drop_scale : 0.f;
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      if (row < L) {
        if constexpr (OUTK == 0 || OUTK == 1) {
          char* img = smem + (OUTK == 0 ? TF_A1B + par * 2 * TF_PA : TF_O_MID + par * TF_PB);
          *(f32x4*)(img + (row + HALO) * TF_RSB + (OOFF + chb) * 4) = v[mt];
          if constexpr (TRAIN && !(TFV & 32)) {
            char* g = (char*)(OUTK == 0 ? A.act1 : A.mid);
            if constexpr (!(TFV & 256)) *(f32x4*)(g + ((R0 + (unsigned)row) * 512u + (unsigned)((OOFF + chb) * 4))) = v[mt];
            const uint32_t bits = (v[mt][0] > 0.f ? 1u : 0u) | (v[mt][1] > 0.f ? 2u : 0u) | (v[mt][2] > 0.f ? 4u : 0u) | (v[mt][3] > 0.f ? 8u : 0u);
            if constexpr (!(TFV & 128)) (OUTK == 0 ? A.m_act1 : A.m_mid)[(R0 + (unsigned)row) * 32u + (unsigned)((OOFF + chb) >> 2)] = (unsigned char)bits;
          }
        } else {
          if constexpr (!(TFV & (32 | 256))) *(f32x4*)((char*)A.act2 + ((R0 + (unsigned)row) * 320u + (unsigned)((OOFF + chb) * 4))) = v[mt];
          if constexpr (TRAIN && !(TFV & (32 | 128))) {
            const uint32_t bits = (v[mt][0] > 0.f ? 1u : 0u) | (v[mt][1] > 0.f ? 2u : 0u) | (v[mt][2] > 0.f ? 4u : 0u) | (v[mt][3] > 0.f ? 8u : 0u);
            A.m_act2[(R0 + (unsigned)row) * 20u + (unsigned)((OOFF + chb) >> 2)] = (unsigned char)bits;
          }
        }
      }
    }
    // ---------------- block 1 only: MaxPool1d(3,1,1) of the output rows (inception.py:99-104 reads it) ----------------
    if constexpr (OUTK == 0 && !(TFV & 64)) {
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
          if constexpr (TRAIN && !(TFV & (32 | 128))) A.amax[(R0 + (unsigned)row) * 32u + (unsigned)((OOFF + chb) >> 2)] = (unsigned char)code[mt];
        }
      }
    }
  }
};

template <int EM, bool TRAIN, bool DROP>
struct TfJobRun<EM, TRAIN, TNone, DROP> {
  __device__ __forceinline__ void init(const TfArgs&, char*, int, int) {}
  __device__ __forceinline__ void run(const TfArgs&, char*, int, unsigned, int, int = 0, int = 0) const {}
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

template <int EM, bool TRAIN, bool DROP, bool LOADER, class J0, class J1, class J2>
__device__ __forceinline__ void tf_role(const TfArgs& A, char* smem, int s, int split, int nwin, int lane) {
  TfJobRun<EM, TRAIN, J0, DROP> r0;
  TfJobRun<EM, TRAIN, J1, DROP> r1;
  TfJobRun<EM, TRAIN, J2, DROP> r2;
  r0.init(A, smem, s, lane);
  r1.init(A, smem, s, lane);
  r2.init(A, smem, s, lane);
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
    TF_STAMP_AT(A.dbg, t, 0);
    if constexpr (LOADER) {
      if (t + 1 < nwin) ld.put(smem, t + 1, lane);
      if (t + 2 < nwin) ld.fetch();
    }
    TF_STAMP_AT(A.dbg, t, 1);
    {
      const int k = t - tfj_stage<J0>();
      if (J0::layer >= 0 && k >= 0 && k < nwin) r0.run(A, smem, k, Rs + k * Rstep, lane, t, 2);
    }
    TF_STAMP_AT(A.dbg, t, 3);
    {
      const int k = t - tfj_stage<J1>();
      if (J1::layer >= 0 && k >= 0 && k < nwin) r1.run(A, smem, k, Rs + k * Rstep, lane, t, 4);
    }
    TF_STAMP_AT(A.dbg, t, 5);
    {
      const int k = t - tfj_stage<J2>();
      if (J2::layer >= 0 && k >= 0 && k < nwin) r2.run(A, smem, k, Rs + k * Rstep, lane, t, 6);
    }
    TF_STAMP_AT(A.dbg, t, 7);
    lds_barrier();
    TF_STAMP_AT(A.dbg, t, 8);
  }
}

template <int EM, bool TRAIN, bool DROP = false>
__global__ __launch_bounds__(TF_THREADS) void tf_fwd_kernel(const TfArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int s = wg / A.nsplit, split = wg - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TF_O_LUT / 4; k += TF_THREADS) z[k] = 0u;
  }
#define TF_ROLE(LD, ...) tf_role<EM, TRAIN, DROP, LD, __VA_ARGS__>(A, smem, s, split, nwin, lane)
#define TJ(...) TJob<__VA_ARGS__>
  // MFMAs per window with Flipout (plain: half): wave w and w + 4 share SIMD w
  switch (wave) {
    case 0: TF_ROLE(EM == EM_LRT, TJ(8, 0, true), TNone, TNone); break;  // k5 64->16: 320 (LRT: its sigma^2 fragments in LDS, + the loader)
    case 4: TF_ROLE(false, TJ(5, 0), TJ(5, 1), TJ(0, 0)); break;         // 128 + 128 + 20
    case 1: TF_ROLE(false, TJ(6, 0), TJ(2, 0), TNone); break;            // 192 + 100
    case 5: TF_ROLE(false, TJ(5, 2), TJ(5, 3), TJ(1, 0)); break;         // 128 + 128 + 60
    case 2: TF_ROLE(false, TJ(7, 0), TJ(7, 1), TJ(1, 1)); break;         // 128 + 128 + 60
    case 6: TF_ROLE(EM != EM_LRT, TJ(7, 2), TJ(7, 3), TJ(0, 1)); break;  // 128 + 128 + 20, + the loader (Flipout / plain)
    case 3: TF_ROLE(false, TJ(9, 0), TJ(9, 1), TJ(3, 0)); break;         // 128 + 128 + 60
    default: TF_ROLE(false, TJ(4, 0), TJ(2, 1), TJ(3, 1)); break;        // 128 + 100 + 60
  }
#undef TJ
#undef TF_ROLE
}

// fp32 planes [rows][20] of the raw windows and of their MaxPool1d(3,1,1) copy (block 1's pooled branch,
// inception.py:41-46); channels 18, 19 zero.  rows = B * L, pooling stays inside a window.  One thread per 4 channels.
__device__ __forceinline__ void xf_planes_dev(const float* x, float* xp, float* xpp, long rows, int L, int F, long idx) {
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
__global__ void xf_planes_kernel(const float* x, float* xp, float* xpp, long rows, int L, int F) {
  xf_planes_dev(x, xp, xpp, rows, L, F, (long)blockIdx.x * blockDim.x + threadIdx.x);   // rows * 5 threads
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
//   tile of stage A (layer 6: 48, waves 0..3; layer 8: 80, waves 4..7): every SIMD carries 480.  Waves 0..3 also stage
//   the next window's dY(ACT2) (global -> registers, one step ahead -> LDS), masking it with [ACT2 > 0] unless PRE.
// ==========================================================================================
enum {
  TD_RS2 = 80 * 4 + 32,                // 352 bytes: pitch 22
  TD_P2 = IMG_ROWS * TD_RS2,           // 12,672: dz(ACT2) image
  TD_PM = IMG_ROWS * TF_RSB,           // 19,584: dz(MID) image
  TD_O_DZ2 = 0,                        // [3 slots]
  TD_O_DZM = 3 * TD_P2,                // [2 bufs]
  TD_O_SGN = TD_O_DZM + 2 * TD_PM,     // [3 slots][80 words]
  TD_O_LUT = TD_O_SGN + 3 * 80 * 4,    // (sign-mask table of the first version: unused, kept as a layout slot)
  TD_O_MSK = TD_O_LUT + 32 * 16,       // [3 slots][m_mid | m_act1 | amax][1024]: the window's mask / code bytes (L x 32 B each)
  TD_LDS = TD_O_MSK + 3 * 3 * 1024
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
  float drop_scale;              // MC-dropout: 1 / (1 - p/4) on dz(ACT1) (its ReLU mask carries the keep mask); else 1
  // LRT (tf_dx_lrt_kernel): q = eps / (2 sd) of the three outputs, and the layers' input values (dX += 2 x o (sigma^2^T dVar))
  const float* q2; const float* qm;
  const float* act1; const float* mid;
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
    const unsigned char* msl = (const unsigned char*)(smem + TD_O_MSK + (k % 3) * 3072);
    uint32_t mb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) mb[mt] = msl[(mt * 16 + i16) * 32 + (och >> 2)];
    uint4 fm = make_uint4(0, 0, 0, 0);
    if constexpr (FO) {
      // sign masks of the lane's 4 k elements (couts 4 g4 .. + 3 of the layer's 16) with s_in of its fragment row (the
      // layer's input channel) folded in: register arithmetic, no LDS table
      const int ci = J * 16 + i16;
      const uint32_t si = 0u - ((sg[ci >> 5] >> (ci & 31)) & 1u);
      const uint32_t w = (sg[4] ^ si) << (28 - 4 * g4);
      fm = make_uint4((w << 3) & 0x80000000u, (w << 2) & 0x80000000u, (w << 1) & 0x80000000u, w & 0x80000000u);
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* lb = sl + i16 * TD_RS2 + (CH0 + 4 * g4) * 4;
    // operands of tap tf + 1 are fetched before the MFMAs of tap tf (see TfJobRun::run)
    f32x4 xc[2], xn[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xc[mt] = *(const f32x4*)(lb + (mt * 16 - PAD + HALO) * TD_RS2);
#pragma unroll
    for (int tf = 0; tf < TAPS; ++tf) {
      if (tf + 1 < TAPS) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xn[mt] = *(const f32x4*)(lb + (mt * 16 + tf + 1 - PAD + HALO) * TD_RS2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wa[tf][j], xc[mt][j], acc[mt]);
      if constexpr (FO) {
        const f32x4 wbm = xor4(wb[tf], fm);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma4(wbm[j], xc[mt][j], acc[mt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      xc[0] = xn[0];
      xc[1] = xn[1];
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
    const int och = CT * 16 + 4 * g4, ci = CT * 16 + i16;
    const unsigned char* msl = (const unsigned char*)(smem + TD_O_MSK + (k % 3) * 3072);
    uint32_t mb[2], code[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      mb[mt] = msl[1024 + row * 32 + (och >> 2)];
      code[mt] = row < L ? (uint32_t)msl[2048 + row * 32 + (och >> 2)] : 0x55u;
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 accp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* b2 = sl + (i16 + HALO) * TD_RS2 + g4 * 16;
    const char* bm = dzm + (i16 + HALO) * TF_RSB + g4 * 16;
    // sign words of the four layers once per job (registers): the s_out words (the contraction index here is the layer's
    // cout) with s_in of this lane's fragment row (the layer's input channel) folded in; a fold is then a per-lane shift, a
    // constant shift and one bit operation per element (tf_fwd_kernel's scheme: no LDS table read in front of a k-block)
    uint32_t sox[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
    const int sh0 = 28 - 4 * g4, sh16 = 12 - 4 * g4;
    if constexpr (FO) {
      constexpr int lys[4] = {4, 5, 7, 9};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t si = 0u - ((sg[lys[u] * 8 + (ci >> 5)] >> (ci & 31)) & 1u);
        sox[u][0] = sg[lys[u] * 8 + 4] ^ si;
        sox[u][1] = sg[lys[u] * 8 + 5] ^ si;
      }
    }
    struct Op {
      f32x4 x[2];
    };
    auto fetch = [&](int q, Op& o) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const char* bp;
        if (q == 0) bp = b2 + mt * 16 * TD_RS2;
        else if (q >= 9) bp = b2 + mt * 16 * TD_RS2 + (48 + (q - 9) * 16) * 4;
        else bp = bm + mt * 16 * TF_RSB + (q - 1) * 64;
        o.x[mt] = *(const f32x4*)bp;
      }
    };
    Op cur, nxt;
    fetch(0, cur);
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      if (q + 1 < 11) fetch(q + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4& ta = q >= 9 ? accp[mt] : acc[mt];
          ta = mfma4(wa[q][j], cur.x[mt][j], ta);
        }
      if constexpr (FO) {
        const int u = q == 0 ? 0 : (q < 5 ? 1 : (q < 9 ? 2 : 3));
        const int bit = q_kb(q) * 16;
        const uint32_t sgw = sox[u][bit >> 5] << ((bit & 31) ? sh16 : sh0);
        f32x4 wbm;
#pragma unroll
        for (int j = 0; j < 4; ++j) wbm[j] = xor1(wb[q][j], (sgw << (3 - j)) & 0x80000000u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x4& ta = q >= 9 ? accp[mt] : acc[mt];
            ta = mfma4(wbm[j], cur.x[mt][j], ta);
          }
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // the folds first (own registers), then the 16 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
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
      if (row < L) *(f32x4*)((char*)A.g_act1 + ((R0 + (unsigned)row) * 512u + (unsigned)(och * 4))) = mask4(v[mt], mb[mt]) * A.drop_scale;
    }
  }
};

// dY(ACT2) of the next windows (and the Flipout sign words): loader wave p of 4 takes the chunks (4 j + p) * 64 + lane
template <bool FO, bool PRE>
struct TdLoader {
  int qo[3], dst[3];
  bool on[3];
  tf_u32x4 g[3], y[PRE ? 1 : 3], mk;
  const unsigned char* msrc;
  int mdst;
  bool mon;
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
    for (int j = 0; j < 3; ++j) {
      const int q = (4 * j + p) * 64 + lane;
      on[j] = q < n2;
      const int qq = on[j] ? q : 0;
      qo[j] = qq * 16;
      dst[j] = (qq / 20 + HALO) * TD_RS2 + (qq % 20) * 16;
    }
    {
      // mask / code bytes of a window: 3 planes x L x 32 B, one 16-byte chunk per loader lane
      const int ll = p * 64 + lane, pl = min(ll / 60, 2), c = ll - (ll / 60) * 60;
      mon = ll < 180 && c * 16 < A.L * 32;
      msrc = (pl == 0 ? A.m_mid : (pl == 1 ? A.m_act1 : A.amax)) + (mon ? c * 16 : 0);
      mdst = pl * 1024 + c * 16;
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
    for (int j = 0; j < 3; ++j) g[j] = *(const tf_u32x4*)(gp + qo[j]);
    if constexpr (!PRE) {
      const char* yp = (const char*)A.act2 + (Rs + k * Rstep) * 320;
#pragma unroll
      for (int j = 0; j < 3; ++j) y[j] = *(const tf_u32x4*)(yp + qo[j]);
    }
    mk = *(const tf_u32x4*)(msrc + (Rs + k * Rstep) * 32);
    if constexpr (FO) {
      if (sg0) sb0 = sg0[(long)k * sst0];
      if (sg1) sb1 = sg1[(long)k * sst1];
    }
  }
  __device__ __forceinline__ void put(const TfDxArgs& A, char* smem, int k, int lane) {
    char* sl = smem + TD_O_DZ2 + (k % 3) * TD_P2;
    char* gd = (char*)A.g_act2m + (Rs + k * Rstep) * 320;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
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
    if (mon) *(tf_u32x4*)(smem + TD_O_MSK + (k % 3) * 3072 + mdst) = mk;
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
  constexpr bool LOADER = W < 4;
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
  const int wg = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int s = wg / A.nsplit, split = wg - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TD_O_LUT / 4; k += TF_THREADS) z[k] = 0u;
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

// ==========================================================================================
// tf_dx_lrt_kernel : the same two-stage pipeline under local reparameterisation.  Every layer has TWO gradient operands,
// dLoc = dz and dVar = dz q (q = eps / (2 sd), kept by the forward), and
//     dX = mu^T dLoc + 2 x o (sigma^2^T dVar)            (x = the layer's input; group_dx_kernel's EM_LRT branch)
// so each stage keeps a mean and a variance accumulator, reads dLoc / dVar image pairs, and multiplies by the layer's
// input in its epilogue (MID for stage A; ACT1 for stage B, its max-pooled copy - rebuilt from ACT1 with the forward's
// DPP row rotations - for the pooled branch).  dVar(ACT2) = dY q2 is built by the loaders, dVar(MID) = dz(MID) q(MID) by
// stage A's epilogue into a second LDS image.  The dz(MID) image pair has no halo rows (stage B's layers are 1x1).
// ==========================================================================================
enum {
  TL_P2 = IMG_ROWS * TD_RS2,             // 12,672
  TL_PM = TILE_ROWS * TF_RSB,            // 17,408: no halo rows
  TL_O_DZ2 = 0,                          // [3 slots][dLoc, dVar]
  TL_O_DZM = 3 * 2 * TL_P2,              // [2 bufs][dLoc, dVar]
  TL_O_MSK = TL_O_DZM + 2 * 2 * TL_PM,   // [3 slots][m_mid | m_act1 | amax][1024]
  TL_LDS = TL_O_MSK + 3 * 3 * 1024
};
static_assert(TL_LDS <= 160 * 1024, "LDS budget of tf_dx_lrt_kernel");

template <int LY, int J>
struct TlJobA {
  static constexpr int TAPS = tl_taps(LY), PAD = (TAPS - 1) / 2, CH0 = LY == 6 ? 16 : 32, MCH = (LY == 6 ? 0 : 64) + J * 16;
  f32x4 wa[TAPS], wb[TAPS];
  __device__ __forceinline__ void init(const TfDxArgs& A, int s, int lane) {
    const int i16 = lane & 15, g4 = lane >> 4;
    const LayerDesc ly = A.layers[LY];
    const long ro = ly.wt_off + (long)(J * 16 + i16) * ly.KPt + 4 * g4;
    const float* pa = (const float*)A.ws.at + A.ws.slott_stride_a * s + ro;
    const float* pb = (const float*)A.ws.bt + A.ws.slott_stride_b * s + ro;
#pragma unroll
    for (int tf = 0; tf < TAPS; ++tf) {
      wa[tf] = *(const f32x4*)(pa + tf * 16);
      wb[tf] = *(const f32x4*)(pb + tf * 16);
    }
  }
  __device__ __forceinline__ void run(const TfDxArgs& A, char* smem, int k, unsigned R0, int lane) const {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int L = A.L;
    const char* sl = smem + TL_O_DZ2 + (k % 3) * 2 * TL_P2;
    char* dzm = smem + TL_O_DZM + (k & 1) * 2 * TL_PM;
    const int och = MCH + 4 * g4;
    const unsigned char* msl = (const unsigned char*)(smem + TL_O_MSK + (k % 3) * 3072);
    uint32_t mb[2];
    f32x4 xin[2], qo[2];   // the layer's input (MID) and q(MID) of this lane's outputs: fetched ahead of the MFMAs
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      mb[mt] = msl[(mt * 16 + i16) * 32 + (och >> 2)];
      const long o = (long)(R0 + (unsigned)min(mt * 16 + i16, L - 1)) * 128 + och;
      xin[mt] = *(const f32x4*)(A.mid + o);
      qo[mt] = *(const f32x4*)(A.qm + o);
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 accv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* lb = sl + i16 * TD_RS2 + (CH0 + 4 * g4) * 4;
#pragma unroll
    for (int tf = 0; tf < TAPS; ++tf) {
      f32x4 x[2], xv[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        x[mt] = *(const f32x4*)(lb + (mt * 16 + tf - PAD + HALO) * TD_RS2);
        xv[mt] = *(const f32x4*)(lb + TL_P2 + (mt * 16 + tf - PAD + HALO) * TD_RS2);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[mt] = mfma4(wa[tf][j], x[mt][j], acc[mt]);
          accv[mt] = mfma4(wb[tf][j], xv[mt][j], accv[mt]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      if (row < L) {
        f32x4 d;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = acc[mt][r] + 2.f * xin[mt][r] * accv[mt][r];
        d = mask4(d, mb[mt]);
        *(f32x4*)(dzm + row * TF_RSB + och * 4) = d;
        *(f32x4*)(dzm + TL_PM + row * TF_RSB + och * 4) = d * qo[mt];   // dVar of the 1x1 layer that produced these channels
        *(f32x4*)((char*)A.g_mid + ((R0 + (unsigned)row) * 512u + (unsigned)(och * 4))) = d;
      }
    }
  }
};

template <int CT>
struct TlJobB {
  f32x4 wa[11], wb[11];
  __host__ __device__ static constexpr int q_layer(int q) { return q == 0 ? 4 : (q < 5 ? 5 : (q < 9 ? 7 : 9)); }
  __host__ __device__ static constexpr int q_kb(int q) { return q == 0 ? 0 : (q < 5 ? q - 1 : (q < 9 ? q - 5 : q - 9)); }
  __device__ __forceinline__ void init(const TfDxArgs& A, int s, int lane) {
    const int i16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      const LayerDesc ly = A.layers[q_layer(q)];
      const long ro = ly.wt_off + (long)(CT * 16 + i16) * ly.KPt + q_kb(q) * 16 + 4 * g4;
      wa[q] = *(const f32x4*)((const float*)A.ws.at + A.ws.slott_stride_a * s + ro);
      wb[q] = *(const f32x4*)((const float*)A.ws.bt + A.ws.slott_stride_b * s + ro);
    }
  }
  __device__ __forceinline__ void run(const TfDxArgs& A, char* smem, int k, unsigned R0, int lane) const {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int L = A.L;
    const char* sl = smem + TL_O_DZ2 + (k % 3) * 2 * TL_P2;
    const char* dzm = smem + TL_O_DZM + (k & 1) * 2 * TL_PM;
    const int och = CT * 16 + 4 * g4;
    const unsigned char* msl = (const unsigned char*)(smem + TL_O_MSK + (k % 3) * 3072);
    uint32_t mb[2], code[2];
    f32x4 a1[2];   // ACT1 values of this lane's outputs (the 1x1 layers' input)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      mb[mt] = msl[1024 + row * 32 + (och >> 2)];
      code[mt] = row < L ? (uint32_t)msl[2048 + row * 32 + (och >> 2)] : 0x55u;
      a1[mt] = *(const f32x4*)(A.act1 + (long)(R0 + (unsigned)min(row, L - 1)) * 128 + och);
      if (row >= L) a1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};   // rows past the window: the pool's identity (ACT1 >= 0)
    }
    f32x4 acc[2], accv[2], accp[2], accpv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = accv[mt] = accp[mt] = accpv[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* b2 = sl + (i16 + HALO) * TD_RS2 + g4 * 16;
    const char* bm = dzm + i16 * TF_RSB + g4 * 16;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      f32x4 x[2], xv[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const char* bp;
        int second;
        if (q == 0) { bp = b2 + mt * 16 * TD_RS2; second = TL_P2; }
        else if (q >= 9) { bp = b2 + mt * 16 * TD_RS2 + (48 + (q - 9) * 16) * 4; second = TL_P2; }
        else { bp = bm + mt * 16 * TF_RSB + (q - 1) * 64; second = TL_PM; }
        x[mt] = *(const f32x4*)bp;
        xv[mt] = *(const f32x4*)(bp + second);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4& ta = q >= 9 ? accp[mt] : acc[mt];
          f32x4& tv = q >= 9 ? accpv[mt] : accv[mt];
          ta = mfma4(wa[q][j], x[mt][j], ta);
          tv = mfma4(wb[q][j], xv[mt][j], tv);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // MaxPool1d(3,1,1) of ACT1 at this lane's rows (the pooled layer's input), as the forward builds it: rows live on the 16
    // lanes of a DPP row; first maximum of (row-1, row, row+1)
    f32x4 pa[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x0 = a1[0][r], x1 = a1[1][r];
      const float up0 = rot16<0x121>(x0), up1 = rot16<0x121>(x1);
      const float dn0 = rot16<0x12F>(x0), dn1 = rot16<0x12F>(x1);
      float b0 = x0, b1 = x1;
      if (i16 > 0) b0 = fmaxf(b0, up0);
      b0 = fmaxf(b0, i16 == 15 ? dn1 : dn0);          // rows >= L hold 0 <= every ACT1 value
      b1 = fmaxf(b1, i16 == 0 ? up0 : up1);
      if (i16 < 15) b1 = fmaxf(b1, dn1);
      pa[0][r] = b0;
      pa[1][r] = b1;
    }
    f32x4 v[2], up[2], dn[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[mt][r] = acc[mt][r] + 2.f * a1[mt][r] * accv[mt][r];
        const uint32_t cd = (code[mt] >> (2 * r)) & 3u;
        const float g = row < L ? accp[mt][r] + 2.f * pa[mt][r] * accpv[mt][r] : 0.f;
        v[mt][r] += cd == 1u ? g : 0.f;
        up[mt][r] = cd == 0u ? g : 0.f;
        dn[mt][r] = cd == 2u ? g : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a0 = rot16<0x121>(dn[0][r]), a1r = rot16<0x121>(dn[1][r]);
      const float c0 = rot16<0x12F>(up[0][r]), c1 = rot16<0x12F>(up[1][r]);
      v[0][r] += (i16 == 0 ? 0.f : a0) + (i16 == 15 ? c1 : c0);
      v[1][r] += (i16 == 0 ? a0 : a1r) + (i16 == 15 ? 0.f : c1);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 16 + i16;
      if (row < L) *(f32x4*)((char*)A.g_act1 + ((R0 + (unsigned)row) * 512u + (unsigned)(och * 4))) = mask4(v[mt], mb[mt]);
    }
  }
};

// dY(ACT2) (premasked by the dense dX) and q(ACT2) of the next windows -> dLoc / dVar images; mask / code bytes
struct TlLoader {
  int qo[3], dst[3];
  bool on[3];
  tf_u32x4 g[3], q[3], mk;
  const unsigned char* msrc;
  int mdst;
  bool mon;
  long Rs, Rstep;
  __device__ __forceinline__ void setup(const TfDxArgs& A, int s, int split, int lane, int p) {
    const int n2 = A.L * 20;
    Rs = ((long)s * A.B + split) * A.L;
    Rstep = (long)A.nsplit * A.L;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = (4 * j + p) * 64 + lane;
      on[j] = c < n2;
      const int qq = on[j] ? c : 0;
      qo[j] = qq * 16;
      dst[j] = (qq / 20 + HALO) * TD_RS2 + (qq % 20) * 16;
    }
    const int ll = p * 64 + lane, pl = min(ll / 60, 2), c = ll - (ll / 60) * 60;
    mon = ll < 180 && c * 16 < A.L * 32;
    msrc = (pl == 0 ? A.m_mid : (pl == 1 ? A.m_act1 : A.amax)) + (mon ? c * 16 : 0);
    mdst = pl * 1024 + c * 16;
  }
  __device__ __forceinline__ void fetch(const TfDxArgs& A, int k) {
    const char* gp = (const char*)A.g_act2 + (Rs + k * Rstep) * 320;
    const char* qp = (const char*)A.q2 + (Rs + k * Rstep) * 320;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      g[j] = *(const tf_u32x4*)(gp + qo[j]);
      q[j] = *(const tf_u32x4*)(qp + qo[j]);
    }
    mk = *(const tf_u32x4*)(msrc + (Rs + k * Rstep) * 32);
  }
  __device__ __forceinline__ void put(char* smem, int k) {
    char* sl = smem + TL_O_DZ2 + (k % 3) * 2 * TL_P2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (on[j]) {
        *(tf_u32x4*)(sl + dst[j]) = g[j];
        const f32x4 dv = __builtin_bit_cast(f32x4, g[j]) * __builtin_bit_cast(f32x4, q[j]);
        *(f32x4*)(sl + TL_P2 + dst[j]) = dv;
      }
    }
    if (mon) *(tf_u32x4*)(smem + TL_O_MSK + (k % 3) * 3072 + mdst) = mk;
  }
};

template <int W>
__device__ __forceinline__ void tl_role(const TfDxArgs& A, char* smem, int s, int split, int nwin, int lane) {
  TlJobB<W> jb;
  TlJobA<(W < 4 ? 6 : 8), (W & 3)> ja;
  jb.init(A, s, lane);
  ja.init(A, s, lane);
  constexpr bool LOADER = W < 4;
  TlLoader ld;
  if constexpr (LOADER) {
    ld.setup(A, s, split, lane, W);
    if (nwin > 0) ld.fetch(A, 0);
  }
  __syncthreads();   // zero fill
  if constexpr (LOADER) {
    if (nwin > 0) ld.put(smem, 0);
    if (nwin > 1) ld.fetch(A, 1);
  }
  lds_barrier();
  const unsigned Rs = (unsigned)(((long)s * A.B + split) * A.L), Rstep = (unsigned)(A.nsplit * A.L);
  const int nsteps = nwin + 1;
  for (int t = 0; t < nsteps; ++t) {
    if constexpr (LOADER) {
      if (t + 1 < nwin) ld.put(smem, t + 1);
      if (t + 2 < nwin) ld.fetch(A, t + 2);
    }
    if (t < nwin) ja.run(A, smem, t, Rs + t * Rstep, lane);
    if (t >= 1) jb.run(A, smem, t - 1, Rs + (t - 1) * Rstep, lane);
    lds_barrier();
  }
}

__global__ __launch_bounds__(TF_THREADS) void tf_dx_lrt_kernel(const TfDxArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int s = wg / A.nsplit, split = wg - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < TL_LDS / 4; k += TF_THREADS) z[k] = 0u;
  }
  switch (wave) {
    case 0: tl_role<0>(A, smem, s, split, nwin, lane); break;
    case 1: tl_role<1>(A, smem, s, split, nwin, lane); break;
    case 2: tl_role<2>(A, smem, s, split, nwin, lane); break;
    case 3: tl_role<3>(A, smem, s, split, nwin, lane); break;
    case 4: tl_role<4>(A, smem, s, split, nwin, lane); break;
    case 5: tl_role<5>(A, smem, s, split, nwin, lane); break;
    case 6: tl_role<6>(A, smem, s, split, nwin, lane); break;
    default: tl_role<7>(A, smem, s, split, nwin, lane); break;
  }
}

// ==========================================================================================
// tf_dw_kernel<EM, KIND> : weight gradients of the conv trunk (exact fp32), two launches:
//   KIND 0 : block 1 (layers 0-3: x / pooled x, dz(ACT1)) and the k3 / k5 level (layers 6, 8: MID, dz(ACT2) 16..47)
//   KIND 1 : the 1x1 level (layers 4, 5, 7 and the pooled layer 9: ACT1, dz(MID), dz(ACT2) 0..15 / 48..79)
// dW[n][tap][c] += sum_rows dz[row][n] * X[row + tap - pad][c]: the contraction index is the window row; both operands
// are COLUMN reads of [row][channel] LDS images (ds_read_b32; row pitches of 16 (mod 32) words put the 4 rows of a k-step
// on disjoint bank groups).  Tiles are accumulated transposed (rows = input channels, columns = couts) and stay in
// registers across the workgroup's windows; they are written once into the workgroup's partial image ("slab", forward
// image layout), summed by slab_reduce_kernel in a fixed order.
// Flipout: the signs of a conv layer are per (window, channel), constant over the rows of a window, so the second
// product is  (dz o s_out)^T (x o s_in) = (s_in (x) s_out) o (dz^T x):  ONE contraction T per tile and window,
// acc_mean += T, acc_dW += signs o T (8 VALU instructions instead of 8 MFMAs).
// Layer 9's operand MaxPool1d(3,1,1)(ACT1) is the maximum of three row-shifted reads of the ACT1 image (ACT1 >= 0, the
// zero halo rows are the pool's identity).  All 8 waves issue a share of the next window's LDS-DMA instructions (TwDma).
// ==========================================================================================
enum {
  TFW_RX = 192, TFW_RB = 576, TFW_RZA = 192, TFW_RZB = 320,
  TFW0_O_X = 0, TFW0_O_XP = TFW0_O_X + IMG_ROWS * TFW_RX, TFW0_O_DZ1 = TFW0_O_XP + IMG_ROWS * TFW_RX,
  TFW0_O_MID = TFW0_O_DZ1 + TILE_ROWS * TFW_RB, TFW0_O_DZ2 = TFW0_O_MID + IMG_ROWS * TFW_RB, TFW0_SLOT = TFW0_O_DZ2 + TILE_ROWS * TFW_RZA,
  TFW1_O_A1 = 0, TFW1_O_DZM = TFW1_O_A1 + IMG_ROWS * TFW_RB, TFW1_O_DZ2 = TFW1_O_DZM + TILE_ROWS * TFW_RB, TFW1_SLOT = TFW1_O_DZ2 + TILE_ROWS * TFW_RZB
};
// LRT: + the q images (q = eps / (2 sd): dVar = dz q is formed on read); x images at a 96-byte pitch so that two slots fit
enum {
  TFWL_RX = 96,
  TFWL0_O_X = 0, TFWL0_O_XP = TFWL0_O_X + IMG_ROWS * TFWL_RX, TFWL0_O_DZ1 = TFWL0_O_XP + IMG_ROWS * TFWL_RX,
  TFWL0_O_MID = TFWL0_O_DZ1 + TILE_ROWS * TFW_RB, TFWL0_O_DZ2 = TFWL0_O_MID + IMG_ROWS * TFW_RB,
  TFWL0_O_Q1 = TFWL0_O_DZ2 + TILE_ROWS * TFW_RZA, TFWL0_O_Q2 = TFWL0_O_Q1 + TILE_ROWS * TFW_RB, TFWL0_SLOT = TFWL0_O_Q2 + TILE_ROWS * TFW_RZA,
  TFWL1_O_QM = TFW1_SLOT, TFWL1_O_Q2 = TFWL1_O_QM + TILE_ROWS * TFW_RB, TFWL1_SLOT = TFWL1_O_Q2 + TILE_ROWS * TFW_RZB
};
template <int KIND, bool LRT = false> __host__ __device__ constexpr int tw_slot() { return LRT ? (KIND == 0 ? TFWL0_SLOT : TFWL1_SLOT) : (KIND == 0 ? TFW0_SLOT : TFW1_SLOT); }
// ring of window slots: the 1x1 kind fits three (its DMAs run two steps ahead), the other kind two
template <int KIND, bool LRT = false> __host__ __device__ constexpr int tw_nslot() { return (KIND == 0 || LRT) ? 2 : 3; }
template <int KIND, bool LRT = false> __host__ __device__ constexpr int tw_lds() { return tw_nslot<KIND, LRT>() * (tw_slot<KIND, LRT>() + 80 * 4); }

struct TfDwArgs {
  const float* xp[2];      // [B*L][20] x, pooled x
  const float* act1;       // [S*B*L][128]
  const float* mid;        // [S*B*L][128]
  const float* g_act1;     // dz (masked) of ACT1 / MID / ACT2
  const float* g_mid;
  const float* g_act2;     // [S*B*L][80]
  const float* q1; const float* qm; const float* q2;   // LRT: q planes of ACT1 / MID / ACT2
  const LayerDesc* layers;
  const uint32_t* sign_in;
  const uint32_t* sign_out;
  long examples;
  float* gw_a; float* gw_b; float* gb_a;   // slabs: [S * nsplit][gw_stride] / [S * nsplit][gb_stride]
  float* gb_b;                             // LRT: slab of the sigma_b^2 gradients
  long gw_stride; int gb_stride;
  int S, B, L, nsplit;
};

// image geometry of a layer's operands inside a slot
template <int KIND, bool LRT = false> __host__ __device__ constexpr int tw_zoff(int l) {
  return KIND == 0 ? (l < 4 ? (LRT ? TFWL0_O_DZ1 : TFW0_O_DZ1) : (LRT ? TFWL0_O_DZ2 : TFW0_O_DZ2)) : ((l == 5 || l == 7) ? TFW1_O_DZM : TFW1_O_DZ2);
}
// LRT: the q image that goes with a layer's dz image (same pitch, same channels)
template <int KIND> __host__ __device__ constexpr int tw_qoff(int l) {
  return KIND == 0 ? (l < 4 ? TFWL0_O_Q1 : TFWL0_O_Q2) : ((l == 5 || l == 7) ? TFWL1_O_QM : TFWL1_O_Q2);
}
template <int KIND> __host__ __device__ constexpr int tw_zpitch(int l) { return KIND == 0 ? (l < 4 ? TFW_RB : TFW_RZA) : ((l == 5 || l == 7) ? TFW_RB : TFW_RZB); }
template <int KIND> __host__ __device__ constexpr int tw_zch(int l) { return KIND == 0 ? (l < 4 ? l * 32 : (l == 6 ? 0 : 16)) : (l == 7 ? 64 : (l == 9 ? 48 : 0)); }
template <int KIND, bool LRT = false> __host__ __device__ constexpr int tw_xoff(int l) {
  return KIND == 0 ? (l < 3 ? (LRT ? TFWL0_O_X : TFW0_O_X) : (l == 3 ? (LRT ? TFWL0_O_XP : TFW0_O_XP) : (LRT ? TFWL0_O_MID : TFW0_O_MID))) : TFW1_O_A1;
}
template <int KIND, bool LRT = false> __host__ __device__ constexpr int tw_xpitch(int l) { return KIND == 0 ? (l < 4 ? (LRT ? TFWL_RX : TFW_RX) : TFW_RB) : TFW_RB; }
__host__ __device__ constexpr int tw_xch(int l) { return l == 8 ? 64 : 0; }

// one (layer, n-tile) job over the c-tiles [CT0, CT0 + NCT) of the layer's own input channels, all taps
struct TwPre {
  float bz[8], ax[8];
};

template <int EM, int KIND, int LY, int NT, int CT0, int NCT, bool BIAS>
struct TwJob {
  static constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT), TWO = FO || LRT;
  static constexpr int TAPS = tl_taps(LY), PAD = (TAPS - 1) / 2, NTILE = TAPS * NCT;
  static constexpr bool POOL3 = (KIND == 1 && LY == 9);
  f32x4 acc_a[NTILE], acc_b[TWO ? NTILE : 1];
  float bsum, bsumv;

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      acc_a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (TWO) acc_b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bsum = bsumv = 0.f;
  }

  __device__ __forceinline__ void load_a(const char* sl, int tt, int lane, float (&ax)[8]) const {
    const int i16 = lane & 15, g4 = lane >> 4;
    const int tap = tt / NCT, c = tt - tap * NCT;
    const char* xi = sl + tw_xoff<KIND, LRT>(LY) + (tw_xch(LY) + (CT0 + c) * 16 + i16) * 4 + (g4 + tap - PAD + HALO) * tw_xpitch<KIND, LRT>(LY);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const char* q = xi + 4 * ks * tw_xpitch<KIND, LRT>(LY);
      if constexpr (POOL3) ax[ks] = fmaxf(fmaxf(*(const float*)(q - TFW_RB), *(const float*)q), *(const float*)(q + TFW_RB));
      else ax[ks] = *(const float*)q;
    }
  }

  // the operands a job needs before its first MFMA: the dz column fragments of its cout tile and the X fragments of its first
  // tile.  The PREVIOUS job of the wave issues these reads in front of its last tile (tw_role), so that their LDS latency
  // hides behind eight MFMAs instead of opening every job (measured: 9 + 12 us of the two launches)
  __device__ __forceinline__ void preload(const char* sl, int lane, TwPre& q) const {
    const int i16 = lane & 15, g4 = lane >> 4;
    constexpr int n0 = tw_zch<KIND>(LY) + NT * 16;
    const char* zi = sl + tw_zoff<KIND, LRT>(LY) + g4 * tw_zpitch<KIND>(LY) + (n0 + i16) * 4;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) q.bz[ks] = (TFV & 2048) ? 1.f + (float)lane : *(const float*)(zi + 4 * ks * tw_zpitch<KIND>(LY));   // (2048: diagnostics)
    load_a(sl, 0, lane, q.ax);
  }

  // acc_mean += T, acc_dW += (s_in (x) s_out) o T.  siw = the window's s_in words of this layer (registers), sobx = 15 where
  // this lane's cout has s_out = -1, else 0.  The sign masks are computed first: they do not depend on T, whose last MFMA
  // is still in flight when the fold starts (it rides in the shadow of the NEXT tile's MFMAs).
  __device__ __forceinline__ void fold(int tt, f32x4 T, const uint32_t (&siw)[4], uint32_t sobx, int lane) {
    if constexpr (TFV & 1) {   // diagnostics: no fold arithmetic, every tile's MFMAs stay live
      acc_a[tt] = T;
      return;
    }
    if constexpr (FO) {
      const int g4 = lane >> 4;
      const int c = tt % NCT;
      const int cbit = (CT0 + c) * 16;   // the layer's own input channel of row 0 of the tile
      const uint32_t nibx = ((siw[cbit >> 5] >> ((cbit & 31) + 4 * g4)) & 15u) ^ sobx;
      uint32_t m[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = (nibx >> r) << 31;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_b[tt][r] += xor1(T[r], m[r]);
    }
    acc_a[tt] += T;
  }

  // q: this job's preloaded operands; next(): issues the next job's preload (called in front of the last tile)
  template <class NX>
  __device__ __forceinline__ void run(const char* sl, const uint32_t* sg, int lane, const TwPre& q, NX&& next) {
    __builtin_amdgcn_sched_barrier(0);
    const int i16 = lane & 15, g4 = lane >> 4;
    constexpr int n0 = tw_zch<KIND>(LY) + NT * 16;
    float bz[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bz[ks] = q.bz[ks];
    if constexpr (LRT) {
      // d mu = dLoc^T x, d sigma^2 = dVar^T x^2 (dLoc = dz, dVar = dz q): two contractions with their own operands on both
      // sides, chained straight into the accumulators (no sign fold)
      float bzv[8];
      const char* qi = sl + tw_qoff<KIND>(LY) + g4 * tw_zpitch<KIND>(LY) + (n0 + i16) * 4;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bzv[ks] = bz[ks] * *(const float*)(qi + 4 * ks * tw_zpitch<KIND>(LY));
      if constexpr (BIAS) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          bsum += bz[ks];
          bsumv += bzv[ks];
        }
      }
      float axc[8], axn[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) axc[ks] = q.ax[ks];
#pragma unroll
      for (int tt = 0; tt < NTILE; ++tt) {
        if (tt + 1 < NTILE) load_a(sl, tt + 1, lane, axn);
        else next();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          acc_a[tt] = mfma4(axc[ks], bz[ks], acc_a[tt]);
          acc_b[tt] = mfma4(axc[ks] * axc[ks], bzv[ks], acc_b[tt]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) axc[ks] = axn[ks];
      }
      return;
    }
    // the window's sign words of this layer, once per job (registers: no LDS read inside the MFMA stream)
    uint32_t siw[4] = {0u, 0u, 0u, 0u}, sobx = 0;
    if constexpr (FO) {
#pragma unroll
      for (int w = 0; w < tf_cimg(LY) / 32; ++w) siw[w] = (TFV & 2048) ? (uint32_t)lane * 0x9e3779b9u : sg[LY * 8 + w];
      sobx = (TFV & 2048) ? (lane & 1 ? 15u : 0u)
                          : (((sg[LY * 8 + 4 + ((NT * 16) >> 5)] >> (((NT * 16) & 31) + i16)) & 1u) ? 15u : 0u);   // s_out of this lane's column (cout)
    }
    if constexpr (BIAS) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bsum += bz[ks];
    }
    // Software pipeline inside the wave: the column reads of tile t + 1 are issued before the MFMAs of tile t, and the
    // fold (VALU) of tile t - 1 is interleaved with them (1 MFMA : up to 4 VALU) - the two waves of a SIMD run the same job
    // structure in lockstep, so a fold phase between the MFMA phases would leave the matrix pipe idle in both.
    float axc[8], axn[8];
    f32x4 Tp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) axc[ks] = q.ax[ks];
#pragma unroll
    for (int tt = 0; tt < NTILE; ++tt) {
      if (tt + 1 < NTILE) load_a(sl, tt + 1, lane, axn);
      else next();   // the next job's first operands, behind this job's last eight MFMAs
      __builtin_amdgcn_sched_barrier(0);
      f32x4 T = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) T = mfma4(axc[ks], bz[ks], T);
      if (tt > 0) fold(tt - 1, Tp, siw, sobx, lane);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      Tp = T;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) axc[ks] = axn[ks];
    }
    fold(NTILE - 1, Tp, siw, sobx, lane);
  }

  __device__ __forceinline__ void flush(const TfDwArgs& A, int slab, int lane) const {
    const LayerDesc ly = A.layers[LY];
    const int g4 = lane >> 4, n = NT * 16 + (lane & 15);
    float* gwa = A.gw_a + A.gw_stride * slab + ly.w_off;
    float* gwb = A.gw_b + A.gw_stride * slab + ly.w_off;
    if (n < ly.cout) {
#pragma unroll
      for (int tt = 0; tt < NTILE; ++tt) {
        const int tap = tt / NCT, c = tt - tap * NCT;
        // transposed tile: this lane holds the image channels ch .. ch+3 of cout n (channel pads are zero columns)
        const long o = (long)n * ly.KP + (long)tap * ly.cin_img + (CT0 + c) * 16 + 4 * g4;
        *(f32x4*)(gwa + o) = acc_a[tt];
        if constexpr (TWO) *(f32x4*)(gwb + o) = acc_b[tt];
      }
    }
    if constexpr (BIAS) {
      float t = bsum;   // lane group g4 summed the rows g4 (mod 4)
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (g4 == 0 && n < ly.cout) A.gb_a[(long)A.gb_stride * slab + ly.bias_off + n] = t;
      if constexpr (LRT) {
        float tv = bsumv;
        tv += __shfl_xor(tv, 16, 64);
        tv += __shfl_xor(tv, 32, 64);
        if (g4 == 0 && n < ly.cout) A.gb_b[(long)A.gb_stride * slab + ly.bias_off + n] = tv;
      }
    }
  }
};

struct TwNone {
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void preload(const char*, int, TwPre&) const {}
  template <class NX>
  __device__ __forceinline__ void run(const char*, const uint32_t*, int, const TwPre&, NX&&) {}
  __device__ __forceinline__ void flush(const TfDwArgs&, int, int) const {}
};

// Staging: LDS-DMA (global_load_lds_dwordx4: no data registers, no LDS store instructions).  A DMA instruction writes
// 64 lanes x 16 B = 1 KB of contiguous LDS, so an image is addressed linearly in 16-byte slots INCLUDING the pad slots of
// its rows (slot q -> row q / slots_per_row, piece q % slots_per_row; lanes on pad slots or past the last row are masked
// off).  The instructions of a window are dealt round-robin to the 8 waves (at most TFW_NDMA each); window t + 1 is
// issued at the start of step t and waited for (vmcnt(0): these waves issue no other VMEM in the loop) before the
// barrier that ends it.
struct TwStream {
  const char* base;   // first row of window 0 of this workgroup (+ first byte of the column range)
  int rowbytes;       // global row stride
  int cpr;            // data pieces (16 B) per row
  int spr;            // LDS slots per row (pitch / 16)
  int dstoff;         // byte offset of row 0 inside a slot of the ring
};
enum { TFW_NDMA = 7, TFWL_NDMA = 9 };   // DMA instructions per wave and window (LRT stages the q images too)
template <bool FO, int ND>
struct TwDma {
  const char* src[ND];
  int wstep[ND];
  uint32_t dst[ND];
  bool on[ND];
  const uint32_t* sg0 = nullptr;
  const uint32_t* sg1 = nullptr;
  long sst0 = 0, sst1 = 0;
  __device__ __forceinline__ void setup(const TfDwArgs& A, const TwStream* st, int nst, int s, int split, int wave, int lane) {
    const int L = A.L;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      int g = j * TF_WAVES + wave;   // instruction index inside the window
      on[j] = false;
      src[j] = st[0].base;
      wstep[j] = 0;
      dst[j] = 0;
      for (int i = 0; i < nst; ++i) {
        const int n = L * st[i].spr, ni = (n + 63) >> 6;
        if (g >= 0 && g < ni) {
          const int q = g * 64 + lane;
          const int row = q / st[i].spr, c = q - row * st[i].spr;
          on[j] = q < n && c < st[i].cpr;
          src[j] = st[i].base + (on[j] ? (long)row * st[i].rowbytes + c * 16 : 0);
          wstep[j] = A.nsplit * L * st[i].rowbytes;
          dst[j] = (uint32_t)(st[i].dstoff + g * 1024);
        }
        g -= ni;
      }
    }
    if constexpr (FO) {
      if (wave == 0) {
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
  // DMA instructions this wave issues per window (wave-uniform): the counted wait of the ring
  __device__ __forceinline__ int count() const {
    int n = 0;
#pragma unroll
    for (int j = 0; j < ND; ++j) n += __builtin_amdgcn_ballot_w64(on[j]) != 0 ? 1 : 0;
    if constexpr (FO) n += (__builtin_amdgcn_ballot_w64(sg0 != nullptr) != 0 ? 1 : 0) + (__builtin_amdgcn_ballot_w64(sg1 != nullptr) != 0 ? 1 : 0);
    return n;
  }
  // window k -> ring slot at LDS byte address `slot`, its sign words at `sgw`
  __device__ __forceinline__ void issue(int k, uint32_t slot, uint32_t sgw) {
#pragma unroll
    for (int j = 0; j < ND; ++j)
      if (on[j]) dma16(src[j] + (long)k * wstep[j], __builtin_amdgcn_readfirstlane(slot + dst[j]));
    if constexpr (FO) {
      if (sg0) dma4(sg0 + (long)k * sst0, __builtin_amdgcn_readfirstlane(sgw));
      if (sg1) dma4(sg1 + (long)k * sst1, __builtin_amdgcn_readfirstlane(sgw + 256));
    }
  }
};

template <int EM, int KIND, class J0, class J1, class J2>
__device__ __forceinline__ void tw_role(const TfDwArgs& A, char* smem, const TwStream* st, int nst, int s, int split, int nwin, int tid) {
  constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT);
  constexpr int SLOT = tw_slot<KIND, LRT>();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  J0 j0;
  J1 j1;
  J2 j2;
  j0.init();
  j1.init();
  j2.init();
  TwDma<FO, (LRT ? TFWL_NDMA : TFW_NDMA)> ld;
  ld.setup(A, st, nst, s, split, wave, lane);
  constexpr int NS = tw_nslot<KIND, LRT>(), AHEAD = NS - 1;
  const int nper = ld.count();
  const uint32_t lds0 = lds_addr(smem), sgb0 = lds0 + NS * SLOT;
  const uint32_t* sgb = (const uint32_t*)(smem + NS * SLOT);
  __syncthreads();   // zero fill
  for (int k = 0; k < AHEAD; ++k)
    if (k < nwin) ld.issue(k, lds0 + k * SLOT, sgb0 + k * 320);
  // window 0 landed: everything but the DMAs of the windows issued after it
  if (AHEAD == 2 && 1 < nwin) BNN_WAIT_VMCNT_WIDE(nper);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  for (int t = 0; t < nwin; ++t) {
    const int ka = t + AHEAD;
    if (ka < nwin && !(TFV & 4)) ld.issue(ka, lds0 + (ka % NS) * SLOT, sgb0 + (ka % NS) * 320);
    const char* sl = smem + (t % NS) * SLOT;
    const uint32_t* sg = sgb + (t % NS) * 80;
    TwPre q0, q1, q2;
    j0.preload(sl, lane, q0);
    if constexpr (KIND == 0) {
      // each job issues its successor's first operand reads in front of its own last tile
      j0.run(sl, sg, lane, q0, [&]() __attribute__((always_inline)) { j1.preload(sl, lane, q1); });
      j1.run(sl, sg, lane, q1, [&]() __attribute__((always_inline)) { j2.preload(sl, lane, q2); });
      j2.run(sl, sg, lane, q2, []() {});
    } else {
      // (the 1x1 kind is at 235 registers and its pooled layer's operands are three reads + two v_max each: chaining
      // measured slower there - 181 instead of 178 us)
      j0.run(sl, sg, lane, q0, []() {});
      j1.preload(sl, lane, q1);
      j1.run(sl, sg, lane, q1, []() {});
      j2.preload(sl, lane, q2);
      j2.run(sl, sg, lane, q2, []() {});
    }
    // window t + 1 must have landed before the barrier: only the DMAs of later windows may stay in flight
    if (AHEAD == 2 && t + 2 < nwin) BNN_WAIT_VMCNT_WIDE(nper);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  j0.flush(A, (int)blockIdx.x, lane);   // slab = this workgroup's (particle, split)
  j1.flush(A, (int)blockIdx.x, lane);
  j2.flush(A, (int)blockIdx.x, lane);
}

template <int EM, int KIND>
__global__ __launch_bounds__(TF_THREADS) void tf_dw_kernel(const TfDwArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / A.nsplit, split = blockIdx.x - s * A.nsplit;
  const int nwin = (A.B - split + A.nsplit - 1) / A.nsplit;
  const int L = A.L;
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < tw_lds<KIND, EM == EM_LRT>() / 4; k += TF_THREADS) z[k] = 0u;
  }
  constexpr bool LRT = (EM == EM_LRT);
  const long pr = ((long)s * A.B + split) * L;   // first row of this workgroup's window 0 in the [S*B*L] row space
  TwStream st[7];
  int nst;
  if constexpr (KIND == 0 && LRT) {
    st[0] = TwStream{(const char*)A.xp[0] + (long)split * L * (TF_XC * 4), TF_XC * 4, 5, TFWL_RX / 16, TFWL0_O_X + HALO * TFWL_RX};
    st[1] = TwStream{(const char*)A.xp[1] + (long)split * L * (TF_XC * 4), TF_XC * 4, 5, TFWL_RX / 16, TFWL0_O_XP + HALO * TFWL_RX};
    st[2] = TwStream{(const char*)A.g_act1 + pr * 512, 512, 32, TFW_RB / 16, TFWL0_O_DZ1};
    st[3] = TwStream{(const char*)A.mid + pr * 512, 512, 32, TFW_RB / 16, TFWL0_O_MID + HALO * TFW_RB};
    st[4] = TwStream{(const char*)A.g_act2 + pr * 320 + 64, 320, 8, TFW_RZA / 16, TFWL0_O_DZ2};
    st[5] = TwStream{(const char*)A.q1 + pr * 512, 512, 32, TFW_RB / 16, TFWL0_O_Q1};
    st[6] = TwStream{(const char*)A.q2 + pr * 320 + 64, 320, 8, TFW_RZA / 16, TFWL0_O_Q2};
    nst = 7;
  } else if constexpr (KIND == 1 && LRT) {
    st[0] = TwStream{(const char*)A.act1 + pr * 512, 512, 32, TFW_RB / 16, TFW1_O_A1 + HALO * TFW_RB};
    st[1] = TwStream{(const char*)A.g_mid + pr * 512, 512, 32, TFW_RB / 16, TFW1_O_DZM};
    st[2] = TwStream{(const char*)A.g_act2 + pr * 320, 320, 20, TFW_RZB / 16, TFW1_O_DZ2};
    st[3] = TwStream{(const char*)A.qm + pr * 512, 512, 32, TFW_RB / 16, TFWL1_O_QM};
    st[4] = TwStream{(const char*)A.q2 + pr * 320, 320, 20, TFW_RZB / 16, TFWL1_O_Q2};
    nst = 5;
  } else if constexpr (KIND == 0) {
    st[0] = TwStream{(const char*)A.xp[0] + (long)split * L * (TF_XC * 4), TF_XC * 4, 5, TFW_RX / 16, TFW0_O_X + HALO * TFW_RX};
    st[1] = TwStream{(const char*)A.xp[1] + (long)split * L * (TF_XC * 4), TF_XC * 4, 5, TFW_RX / 16, TFW0_O_XP + HALO * TFW_RX};
    st[2] = TwStream{(const char*)A.g_act1 + pr * 512, 512, 32, TFW_RB / 16, TFW0_O_DZ1};
    st[3] = TwStream{(const char*)A.mid + pr * 512, 512, 32, TFW_RB / 16, TFW0_O_MID + HALO * TFW_RB};
    st[4] = TwStream{(const char*)A.g_act2 + pr * 320 + 64, 320, 8, TFW_RZA / 16, TFW0_O_DZ2};
    nst = 5;
  } else {
    st[0] = TwStream{(const char*)A.act1 + pr * 512, 512, 32, TFW_RB / 16, TFW1_O_A1 + HALO * TFW_RB};
    st[1] = TwStream{(const char*)A.g_mid + pr * 512, 512, 32, TFW_RB / 16, TFW1_O_DZM};
    st[2] = TwStream{(const char*)A.g_act2 + pr * 320, 320, 20, TFW_RZB / 16, TFW1_O_DZ2};
    nst = 3;
  }
#define TWJ(...) TwJob<EM, KIND, __VA_ARGS__>
#define TW_ROLE(...) tw_role<EM, KIND, __VA_ARGS__>(A, smem, st, nst, s, split, nwin, tid)
  if constexpr (KIND == 0) {
    // tiles per wave: 10 10 10 10 9 11 11 9 (8 MFMAs each per window); wave w and w + 4 share SIMD w
    switch (wave) {
      case 0: TW_ROLE(TWJ(2, 0, 0, 2, true), TwNone, TwNone); break;
      case 1: TW_ROLE(TWJ(2, 1, 0, 2, true), TwNone, TwNone); break;
      case 2: TW_ROLE(TWJ(8, 0, 0, 2, true), TwNone, TwNone); break;
      case 3: TW_ROLE(TWJ(8, 0, 2, 2, false), TwNone, TwNone); break;
      case 4: TW_ROLE(TWJ(1, 0, 0, 2, true), TWJ(6, 0, 3, 1, false), TwNone); break;
      case 5: TW_ROLE(TWJ(1, 1, 0, 2, true), TWJ(6, 0, 0, 1, true), TWJ(0, 0, 0, 2, true)); break;
      case 6: TW_ROLE(TWJ(3, 0, 0, 2, true), TWJ(6, 0, 1, 1, false), TWJ(0, 1, 0, 2, true)); break;
      default: TW_ROLE(TWJ(3, 1, 0, 2, true), TWJ(6, 0, 2, 1, false), TwNone); break;
    }
  } else {
    // 11 tiles per wave
    switch (wave) {
      case 0: TW_ROLE(TWJ(5, 0, 0, 8, true), TWJ(4, 0, 0, 3, true), TwNone); break;
      case 1: TW_ROLE(TWJ(5, 1, 0, 8, true), TWJ(4, 0, 3, 3, false), TwNone); break;
      case 2: TW_ROLE(TWJ(5, 2, 0, 8, true), TWJ(4, 0, 6, 2, false), TWJ(9, 0, 0, 1, true)); break;
      case 3: TW_ROLE(TWJ(5, 3, 0, 8, true), TWJ(9, 0, 1, 3, false), TwNone); break;
      case 4: TW_ROLE(TWJ(7, 0, 0, 8, true), TWJ(9, 0, 4, 3, false), TwNone); break;
      case 5: TW_ROLE(TWJ(7, 1, 0, 8, true), TWJ(9, 0, 7, 1, false), TWJ(9, 1, 0, 2, true)); break;
      case 6: TW_ROLE(TWJ(7, 2, 0, 8, true), TWJ(9, 1, 2, 3, false), TwNone); break;
      default: TW_ROLE(TWJ(7, 3, 0, 8, true), TWJ(9, 1, 5, 3, false), TwNone); break;
    }
  }
#undef TW_ROLE
#undef TWJ
}

// ==========================================================================================
// Flatten -> Linear(80 L, 64) of the Inception net (inception.py:200-216) in exact fp32: K-split, weight-stationary.
//   workgroup = (particle, 240-channel K chunk, row range); the chunk's weight fragments (mean | per-particle W, Flipout
//   dW) live in registers for the whole launch, the example rows stream through LDS in 32-row steps (all waves stage a
//   share: registers, one step ahead).  Flipout's signs are per EXAMPLE here, i.e. per row: s_in is folded into the
//   activation fragment (sign words of the step's rows in LDS), s_out is applied to the perturbation accumulator in the
//   epilogue.  densef_fwd writes per-chunk partial pre-activations; dense_ks_fin_kernel sums them in order (+ bias, ReLU).
// ==========================================================================================
enum {
  FDF_CH = 240, FDF_KB = FDF_CH / 16, FDF_ROWS = 32,
  FDF_RSX = FDF_CH * 4 + 32,                   // 992 bytes: pitch 62 (row reads of the forward)
  FDF_SGW = 12,                               // sign words per row: 9 s_in words covering the chunk + 2 s_out + pad
  FDF_O_SG = FDF_ROWS * FDF_RSX,
  FDF_SLOT = FDF_O_SG + FDF_ROWS * FDF_SGW * 4,  // 33,280
  FDF_NSLOT = 3,
  FDF_O_LUT = FDF_NSLOT * FDF_SLOT,
  FDF_LDS = FDF_O_LUT + 32 * 16
};

// Work items of the dense forward / dX = (particle, chunk, 32-row step), numbered pair-major (pair = particle * nchunk + chunk,
// SP steps each).  Two ways to deal them to the workgroups (the host takes the cheaper one, densef_schedule in plan.hip):
//   q == 0 : workgroup g takes the contiguous items [items g / G, items (g + 1) / G): equal step counts whatever S * nchunk
//            is; a range that crosses a pair boundary reloads the weight fragments (a second ~7 us load for most workgroups)
//   q > 0  : every pair is cut into k = SP / q full ranges of q steps (one workgroup, one fragment load each); the remainders
//            (r = SP - k q steps per pair) are packed m = q / r to a workgroup.  S = 10, B = 1000: 200 workgroups with 13
//            steps and one load, 50 with 2 x 6 steps.
struct DfSeg {
  int pair, t0, n;
};
__host__ __device__ __forceinline__ bool df_segment(int wg, int nwg, int seg, int pairs, int SP, int q, long& item, DfSeg& o) {
  if (q == 0) {
    const long items = (long)pairs * SP;
    if (seg == 0) item = items * wg / nwg;
    const long item_end = items * (wg + 1) / nwg;
    if (item >= item_end) return false;
    o.pair = (int)(item / SP);
    o.t0 = (int)(item - (long)o.pair * SP);
    const long left = item_end - item;
    o.n = (long)(SP - o.t0) < left ? SP - o.t0 : (int)left;
    item += o.n;
    return true;
  }
  const int k = SP / q, r = SP - k * q;
  if (wg < pairs * k) {
    if (seg) return false;
    o.pair = wg / k;
    o.t0 = (wg - o.pair * k) * q;
    o.n = q;
    return true;
  }
  if (r == 0) return false;
  const int m = q / r > 1 ? q / r : 1;
  const int p = (wg - pairs * k) * m + seg;
  if (seg >= m || p >= pairs) return false;
  o.pair = p;
  o.t0 = k * q;
  o.n = r;
  return true;
}

struct DfArgs {
  const float* x;                     // [S*B][x_ctot] fp32: the layer's input rows
  int x_ctot;
  const float* wa; const float* wb;   // forward images of the layer [64][KP] (slot A: mean | sampled W, slot B: dW), offset to the layer
  long stride_a, stride_b;            // elements between particles (0: shared)
  int KP;
  const uint32_t* sg_in; const uint32_t* sg_out;   // packed signs of the layer [S*B][siw] / [S*B][sow]
  int siw, sow;
  float* slab;                        // [nchunk][S*B][64] partial pre-activations
  float* slabv;                       // LRT: partial variances sigma^2 . x^2 (same layout)
  long slab_stride;
  int S, B, nchunk, nrs, rows_per_wg;
  int q;                              // item schedule (df_segment)
};

template <int EM>
__global__ __launch_bounds__(TF_THREADS) void densef_fwd_kernel(const DfArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT), TWO = FO || LRT;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave & 3, mh = wave >> 2;
  build_sign_lut_f32((uint4*)(smem + FDF_O_LUT), tid);
  // this workgroup's (pair, row step range) segments: df_segment (100 pairs x 2 row ranges used to leave 56 of 256 CUs idle)
  const int SP = (A.B + FDF_ROWS - 1) / FDF_ROWS;
  const int wg = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);   // neighbours in item space (the same weight fragments) share an L2
  long item = 0;
  DfSeg sg_;
  for (int seg = 0; df_segment(wg, (int)gridDim.x, seg, A.S * A.nchunk, SP, A.q, item, sg_); ++seg) {
  const int pair = sg_.pair, t_first = sg_.t0, nsteps = sg_.n;
  const int s = pair / A.nchunk, chunk = pair - s * A.nchunk;
  const int b0 = t_first * FDF_ROWS, b1 = min(A.B, b0 + nsteps * FDF_ROWS);
  const int ch0 = chunk * FDF_CH;
  const int w0 = ch0 >> 5;   // first s_in word of the chunk
  // ---- weight fragments: rows nt*16 + i16 of the forward image, k-blocks of the chunk ----
  f32x4 wa[FDF_KB], wb[TWO ? FDF_KB : 1];
  {
    const float* pa = A.wa + A.stride_a * s + (long)(nt * 16 + i16) * A.KP + ch0 + 4 * g4;
    const float* pb = A.wb + A.stride_b * s + (long)(nt * 16 + i16) * A.KP + ch0 + 4 * g4;
#pragma unroll
    for (int kb = 0; kb < FDF_KB; ++kb) {
      if ((TFV & 512) || ((TFV & 1024) && mh == 1)) {   // diagnostics: no weight loads (1024: by one row half's waves only)
        wa[kb] = f32x4{1.f, 2.f, 3.f, (float)lane};
        if constexpr (TWO) wb[kb] = f32x4{1.f, 2.f, 3.f, (float)lane};
        continue;
      }
      wa[kb] = *(const f32x4*)(pa + kb * 16);
      if constexpr (TWO) wb[kb] = *(const f32x4*)(pb + kb * 16);
    }
  }
  // ---- staging plan: 4 chunks of the step's [32][240] rows per thread, sign words by the first 352 threads ----
  int xq_row[4], xq_off[4], xq_dst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = j * TF_THREADS + tid;   // < 2048; rows 0..31 x 60 chunks = 1920
    const int row = min(q / 60, FDF_ROWS - 1), c = q % 60;
    xq_row[j] = q < FDF_ROWS * 60 ? row : -1;
    xq_off[j] = ch0 * 4 + c * 16;
    xq_dst[j] = row * FDF_RSX + c * 16;
  }
  const int sq_row = tid / 11, sq_w = tid - sq_row * 11;
  const bool sq_on = FO && tid < FDF_ROWS * 11;
  tf_u32x4 xr[4];
  uint32_t sw = 0;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int bb = b0 + t * FDF_ROWS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long R = (long)s * A.B + min(bb + max(xq_row[j], 0), A.B - 1);
      xr[j] = *(const tf_u32x4*)((const char*)A.x + R * A.x_ctot * 4 + xq_off[j]);
    }
    if constexpr (FO) {
      if (sq_on) {
        const long R = (long)s * A.B + min(bb + sq_row, A.B - 1);
        sw = sq_w < 9 ? A.sg_in[R * A.siw + min(w0 + sq_w, A.siw - 1)] : A.sg_out[R * A.sow + min(sq_w - 9, A.sow - 1)];
      }
    }
  };
  auto put = [&](int t) __attribute__((always_inline)) {
    char* sl = smem + (t % FDF_NSLOT) * FDF_SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (xq_row[j] >= 0) *(tf_u32x4*)(sl + xq_dst[j]) = xr[j];
    if constexpr (FO) {
      if (sq_on) ((uint32_t*)(sl + FDF_O_SG))[sq_row * FDF_SGW + sq_w] = sw;
    }
  };
  if (nsteps > 0) fetch(0);
  __syncthreads();   // LUT
  if (nsteps > 0) put(0);
  if (nsteps > 1) fetch(1);
  lds_barrier();
  const uint4* lut = (const uint4*)(smem + FDF_O_LUT);
  for (int t = 0; t < nsteps; ++t) {
    const char* sl = smem + (t % FDF_NSLOT) * FDF_SLOT;
    const uint32_t* sgl = (const uint32_t*)(sl + FDF_O_SG) + (mh * 16 + i16) * FDF_SGW;
    const char* lb = sl + (mh * 16 + i16) * FDF_RSX + g4 * 16;
    f32x4 accm = {0.f, 0.f, 0.f, 0.f}, accp = {0.f, 0.f, 0.f, 0.f};
    // this row's s_in words of the chunk, once per step (registers); operands of k-block kb + 1 are fetched before the
    // MFMAs of k-block kb (see TfJobRun::run)
    uint32_t swr[9];
    const int hi16 = (ch0 >> 4) & 1;
    if constexpr (FO) {
#pragma unroll
      for (int w = 0; w < 9; ++w) swr[w] = sgl[w];
    }
    struct Op {
      f32x4 x;
    };
    auto fetchop = [&](int kb, Op& o) __attribute__((always_inline)) { o.x = *(const f32x4*)(lb + kb * 64); };
    // s_in of this row into the activation fragment: image channel of the lane's 4 values c = ch0 + kb*16 + 4 g4, ch0 = 0 or 16
    // (mod 32), so a 16-channel block starts at bit 0 or 16 of its sign word; the block's bits go to the top of a register with
    // a per-lane shift, element j's bit to bit 31 with a constant one (tf_fwd_kernel's scheme: no LDS table read per k-block)
    const int shE = hi16 ? 12 - 4 * g4 : 28 - 4 * g4, shO = hi16 ? 28 - 4 * g4 : 12 - 4 * g4;   // even / odd k-blocks
    Op cur, nxt;
    fetchop(0, cur);
#pragma unroll
    for (int kb = 0; kb < FDF_KB; ++kb) {
      if (kb + 1 < FDF_KB) fetchop(kb + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 xs = cur.x;
      if constexpr (FO) {
        const uint32_t word = hi16 ? swr[(kb + 1) >> 1] : swr[kb >> 1];
        const uint32_t sgw = word << ((kb & 1) ? shO : shE);
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = xor1(cur.x[j], (sgw << (3 - j)) & 0x80000000u);
      }
      if constexpr (LRT) xs = cur.x * cur.x;   // var = sigma^2 . x^2
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        accm = mfma4(wa[kb][j], cur.x[j], accm);
        if constexpr (TWO) accp = mfma4(wb[kb][j], xs[j], accp);
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    // next step's rows: written after this step's operand reads were issued, loads of the step after that behind them;
    // the partial-sum stores come last (the staging never waits on a store it has just issued)
    if (t + 1 < nsteps) put(t + 1);
    if (t + 2 < nsteps) fetch(t + 2);
    const int b = b0 + t * FDF_ROWS + mh * 16 + i16;
    f32x4 v = accm;
    if constexpr (FO) {
      const int n = nt * 16 + 4 * g4;
      const uint32_t nib = (sgl[9 + (n >> 5)] >> (n & 31)) & 15u;
      const f32x4 ps = xor4(accp, lut[nib]);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += ps[r];
    }
    if (b < b1) {
      *(f32x4*)(A.slab + chunk * A.slab_stride + ((long)s * A.B + b) * 64 + nt * 16 + 4 * g4) = v;
      if constexpr (LRT) *(f32x4*)(A.slabv + chunk * A.slab_stride + ((long)s * A.B + b) * 64 + nt * 16 + 4 * g4) = accp;
    }
    lds_barrier();
  }
  }   // next (pair, step range) of this workgroup
}

// ==========================================================================================
// backward of that layer, two launches over the same (particle, chunk, row range) grid:
//   densef_dx_kernel : dX[R][c] = sum_n dz[R][n] W[n][c]  (+ Flipout: s_in[R][c] sum_n (s_out dz)[R][n] dW[n][c]);
//                      dz = dH [H > 0] is built while staging; transposed weight fragments in registers
//   densef_dw_kernel : dW[n][c] = sum_R dz[R][n] X[R][c]  (+ Flipout: sum_R (s_out dz)[R][n] (s_in X)[R][c]: the signs are
//                      per row here, so the second product is a contraction of its own); tiles in registers across the rows
// ==========================================================================================
struct DfBwdArgs {
  const float* x;                       // [S*B][x_ctot]
  int x_ctot;
  const float* h; const float* dh;      // [S*B][64]: the layer's output (after ReLU) and its gradient
  const float* wat; const float* wbt;   // transposed images [cin][KPt] (offset to the layer)
  long stride_at, stride_bt;
  int KPt;
  const uint32_t* sg_in; const uint32_t* sg_out;
  int siw, sow;
  float* dx;                            // [S*B][x_ctot]
  const unsigned char* m_x;             // [S*B][x_ctot / 4] nibble masks [X > 0] of the layer's input (null: dX is stored unmasked)
  float x_scale, h_scale;               // MC-dropout: 1 / (1 - p/4) on dX, 1 / (1 - p) on dz = dH [H > 0]; else 1
  float* gw_a; float* gw_b; float* gb_a;   // per-particle gradient images of the layer (forward layout [64][KP]) / bias gradients
  float* gw2_a; float* gw2_b; float* gb2_a;   // partial images of row ranges 1 .. nrs - 1: image (rs - 1) * S + s, same strides (summed by dense_addn_kernel)
  const float* qh;                      // LRT: q = eps / (2 sd) of the layer's output [S*B][64]: dVar = dz q
  float* gb_b; float* gb2_b;            // LRT: gradient of sigma_b^2 (+ its second-range partial)
  long gw_stride; int gb_stride;
  int KP;
  int S, B, nchunk, nrs, rows_per_wg;
  int q;                                // item schedule of the dX launch (df_segment)
  int q_dw;                             // steps per full row range of the dW launch (df_segment with q > 0)
};

enum {
  FDX_RSZ = 64 * 4 + 32,                        // 288 bytes: pitch 18 (row reads)
  FDX_O_DZS = FDF_ROWS * FDX_RSZ, FDX_O_SG = 2 * FDF_ROWS * FDX_RSZ,
  FDX_SLOT = FDX_O_SG + FDF_ROWS * FDF_SGW * 4,    // 19,968
  FDX_O_LUT = FDF_NSLOT * FDX_SLOT,
  FDX_LDS = FDX_O_LUT + 32 * 16,
  DWF_RSX = FDF_CH * 4,                         // 960 bytes = 240 words = 48 (mod 64): column reads hit disjoint bank groups
  DWF_RSZ = 80 * 4,                            // 320 bytes = 80 words = 16 (mod 64)
  DWF_O_DZ = FDF_ROWS * DWF_RSX, DWF_O_DZS = DWF_O_DZ + FDF_ROWS * DWF_RSZ, DWF_O_SG = DWF_O_DZS + FDF_ROWS * DWF_RSZ,
  DWF_SLOT = DWF_O_SG + FDF_ROWS * FDF_SGW * 4,  // 52,736
  DWF_LDS = 2 * DWF_SLOT
};

// dz = dH [H > 0] of 4 couts of one row, and the second product's operand (Flipout: dz o s_out; LRT: dVar = dz q): the
// staging unit of both backward kernels
struct DzStage {
  tf_u32x4 g, y, qv;
  uint32_t so;
  int row, c4;
  __device__ __forceinline__ void setup(int tid) { row = tid >> 4; c4 = tid & 15; }
  template <int EM>
  __device__ __forceinline__ void fetch(const DfBwdArgs& A, int s, int bb) {
    const long R = (long)s * A.B + min(bb + row, A.B - 1);
    g = *(const tf_u32x4*)(A.dh + R * 64 + c4 * 4);
    y = *(const tf_u32x4*)(A.h + R * 64 + c4 * 4);
    if constexpr (EM == EM_FLIPOUT) so = A.sg_out[R * A.sow + (c4 >> 3)];
    if constexpr (EM == EM_LRT) qv = *(const tf_u32x4*)(A.qh + R * 64 + c4 * 4);
  }
  template <int EM>
  __device__ __forceinline__ void put(char* dz, char* dzs, int pitch, bool live, float hs) const {
    tf_u32x4 d = g;
    const f32x4 yy = __builtin_bit_cast(f32x4, y);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (live && yy[e] > 0.f) ? __float_as_uint(__uint_as_float(d[e]) * hs) : 0u;
    *(tf_u32x4*)(dz + row * pitch + c4 * 16) = d;
    if constexpr (EM == EM_FLIPOUT) {
      const uint32_t nib = (so >> ((c4 & 7) * 4)) & 15u;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] ^= ((nib >> e) & 1u) << 31;
      *(tf_u32x4*)(dzs + row * pitch + c4 * 16) = d;
    }
    if constexpr (EM == EM_LRT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = __float_as_uint(__uint_as_float(d[e]) * __uint_as_float(qv[e]));
      *(tf_u32x4*)(dzs + row * pitch + c4 * 16) = d;
    }
  }
};

template <int EM>
__global__ __launch_bounds__(TF_THREADS) void densef_dx_kernel(const DfBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT), TWO = FO || LRT;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0t = wave & 3, mh = wave >> 2;   // c-tiles c0t, c0t + 4, c0t + 8, c0t + 12 (< 15); row half
  build_sign_lut_f32((uint4*)(smem + FDX_O_LUT), tid);
  // this workgroup's (pair, row step range) segments: df_segment
  const int SP = (A.B + FDF_ROWS - 1) / FDF_ROWS;
  const int wg = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  long item = 0;
  DfSeg sg_;
  for (int seg = 0; df_segment(wg, (int)gridDim.x, seg, A.S * A.nchunk, SP, A.q, item, sg_); ++seg) {
  const int pair = sg_.pair, t_first = sg_.t0, nsteps = sg_.n;
  const int s = pair / A.nchunk, chunk = pair - s * A.nchunk;
  const int b0 = t_first * FDF_ROWS, b1 = min(A.B, b0 + nsteps * FDF_ROWS);
  const int ch0 = chunk * FDF_CH, w0 = ch0 >> 5;
  f32x4 wa[4][4], wb[TWO ? 4 : 1][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ct = min(c0t + 4 * m, FDF_KB - 1);
    const long ro = (long)(ch0 + ct * 16 + i16) * A.KPt + 4 * g4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      wa[m][kb] = *(const f32x4*)(A.wat + A.stride_at * s + ro + kb * 16);
      if constexpr (TWO) wb[m][kb] = *(const f32x4*)(A.wbt + A.stride_bt * s + ro + kb * 16);
    }
  }
  DzStage zs;
  zs.setup(tid);
  const int sq_row = tid / 9, sq_w = tid - sq_row * 9;
  const bool sq_on = FO && tid < FDF_ROWS * 9;
  uint32_t sw = 0;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int bb = b0 + t * FDF_ROWS;
    zs.fetch<EM>(A, s, bb);
    if constexpr (FO) {
      if (sq_on) sw = A.sg_in[((long)s * A.B + min(bb + sq_row, A.B - 1)) * A.siw + min(w0 + sq_w, A.siw - 1)];
    }
  };
  auto put = [&](int t) __attribute__((always_inline)) {
    char* sl = smem + (t % FDF_NSLOT) * FDX_SLOT;
    zs.put<EM>(sl, sl + FDX_O_DZS, FDX_RSZ, b0 + t * FDF_ROWS + zs.row < b1, A.h_scale);
    if constexpr (FO) {
      if (sq_on) ((uint32_t*)(sl + FDX_O_SG))[sq_row * FDF_SGW + sq_w] = sw;
    }
  };
  if (nsteps > 0) fetch(0);
  __syncthreads();
  if (nsteps > 0) put(0);
  if (nsteps > 1) fetch(1);
  lds_barrier();
  const uint4* lut = (const uint4*)(smem + FDX_O_LUT);
  for (int t = 0; t < nsteps; ++t) {
    const char* sl = smem + (t % FDF_NSLOT) * FDX_SLOT;
    const char* lb = sl + (mh * 16 + i16) * FDX_RSZ + g4 * 16;
    uint32_t mxb[4] = {15u, 15u, 15u, 15u};
    if (A.m_x) {
      const long R = (long)s * A.B + min(b0 + t * FDF_ROWS + mh * 16 + i16, A.B - 1);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        mxb[m] = A.m_x[R * (A.x_ctot >> 2) + ((ch0 + min(c0t + 4 * m, FDF_KB - 1) * 16) >> 2) + g4];
    }
    f32x4 accm[4], accp[TWO ? 4 : 1];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      accm[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (TWO) accp[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LRT: dX = mu^T dLoc + 2 X o (sigma^2^T dVar): the layer's input values of this lane's outputs, fetched ahead of the MFMAs
    f32x4 xv[LRT ? 4 : 1];
    if constexpr (LRT) {
      const long R = (long)s * A.B + min(b0 + t * FDF_ROWS + mh * 16 + i16, A.B - 1);
#pragma unroll
      for (int m = 0; m < 4; ++m) xv[m] = *(const f32x4*)(A.x + R * A.x_ctot + ch0 + min(c0t + 4 * m, FDF_KB - 1) * 16 + 4 * g4);
    }
    // all operand reads of the step first (K = 64 couts: four k-blocks), then its MFMAs
    f32x4 z[4], zz[TWO ? 4 : 1];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      z[kb] = *(const f32x4*)(lb + kb * 64);
      if constexpr (TWO) zz[kb] = *(const f32x4*)(lb + FDX_O_DZS + kb * 64);   // Flipout: dz o s_out; LRT: dVar = dz q
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (m < 3 || c0t + 4 * m < FDF_KB) {
            accm[m] = mfma4(wa[m][kb][j], z[kb][j], accm[m]);
            if constexpr (TWO) accp[m] = mfma4(wb[m][kb][j], zz[kb][j], accp[m]);
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nsteps) put(t + 1);
    if (t + 2 < nsteps) fetch(t + 2);
    const int b = b0 + t * FDF_ROWS + mh * 16 + i16;
    const uint32_t* sgl = (const uint32_t*)(sl + FDX_O_SG) + (mh * 16 + i16) * FDF_SGW;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int ct = c0t + 4 * m;
      if (ct < FDF_KB) {
        f32x4 v = accm[m];
        const int c = ch0 + ct * 16 + 4 * g4;
        if constexpr (FO) {
          const uint32_t nib = (sgl[(c >> 5) - w0] >> (c & 31)) & 15u;
          const f32x4 ps = xor4(accp[m], lut[nib]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += ps[r];
        }
        if constexpr (LRT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += 2.f * xv[m][r] * accp[m][r];
        }
        if (A.m_x) v = mask4(v, mxb[m]) * A.x_scale;   // the layer's input is a ReLU output: its gradient is wanted where it is positive only
        if (b < b1) *(f32x4*)(A.dx + ((long)s * A.B + b) * A.x_ctot + c) = v;
      }
    }
    lds_barrier();
  }
  }   // next (pair, step range) of this workgroup
}

template <int EM>
__global__ __launch_bounds__(TF_THREADS) void densef_dw_kernel(const DfBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT), TWO = FO || LRT;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave & 3, cpar = wave >> 2;   // c-tiles cpar, cpar + 2, ... (< 15)
  {
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < DWF_LDS / 4; k += TF_THREADS) z[k] = 0u;
  }
  // this workgroup's (pair, row step range) segments (df_segment, q > 0: full ranges of q steps + packed remainders): the
  // range starting at step j q of its pair is the pair's row range j - range 0 stores the gradient image, the others partial
  // images (dense_addn_kernel)
  const int SP = (A.B + FDF_ROWS - 1) / FDF_ROWS;
  long item_ = 0;
  DfSeg sg_;
  for (int seg = 0; df_segment((int)blockIdx.x, (int)gridDim.x, seg, A.S * A.nchunk, SP, A.q_dw, item_, sg_); ++seg) {
  const int rs = sg_.t0 / A.q_dw;
  const int s = sg_.pair / A.nchunk, chunk = sg_.pair - s * A.nchunk;
  const int b0 = sg_.t0 * FDF_ROWS, b1 = min(A.B, b0 + sg_.n * FDF_ROWS);
  const int nsteps = sg_.n;
  const int ch0 = chunk * FDF_CH, w0 = ch0 >> 5;
  f32x4 acc_a[8], acc_b[TWO ? 8 : 1];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    acc_a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (TWO) acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float bsum = 0.f, bsumv = 0.f;
  // staging: X chunk [32][240] (4 chunks per thread), dz, sign words
  int xq_row[4], xq_off[4], xq_dst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = j * TF_THREADS + tid;
    const int row = min(q / 60, FDF_ROWS - 1), c = q % 60;
    xq_row[j] = q < FDF_ROWS * 60 ? row : -1;
    xq_off[j] = ch0 * 4 + c * 16;
    xq_dst[j] = row * DWF_RSX + c * 16;
  }
  DzStage zs;
  zs.setup(tid);
  const int sq_row = tid / 9, sq_w = tid - sq_row * 9;
  const bool sq_on = FO && tid < FDF_ROWS * 9;
  tf_u32x4 xr[4];
  uint32_t sw = 0;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int bb = b0 + t * FDF_ROWS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long R = (long)s * A.B + min(bb + max(xq_row[j], 0), A.B - 1);
      xr[j] = *(const tf_u32x4*)((const char*)A.x + R * A.x_ctot * 4 + xq_off[j]);
    }
    zs.fetch<EM>(A, s, bb);
    if constexpr (FO) {
      if (sq_on) sw = A.sg_in[((long)s * A.B + min(bb + sq_row, A.B - 1)) * A.siw + min(w0 + sq_w, A.siw - 1)];
    }
  };
  auto put = [&](int t) __attribute__((always_inline)) {
    char* sl = smem + (t & 1) * DWF_SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (xq_row[j] >= 0) *(tf_u32x4*)(sl + xq_dst[j]) = xr[j];
    // rows past the end of the range carry dz = 0: their X rows (clamped copies) do not contribute
    zs.put<EM>(sl + DWF_O_DZ, sl + DWF_O_DZS, DWF_RSZ, b0 + t * FDF_ROWS + zs.row < b1, A.h_scale);
    if constexpr (FO) {
      if (sq_on) ((uint32_t*)(sl + DWF_O_SG))[sq_row * FDF_SGW + sq_w] = sw;
    }
  };
  if (nsteps > 0) fetch(0);
  __syncthreads();
  if (nsteps > 0) put(0);
  if (nsteps > 1) fetch(1);
  lds_barrier();
  for (int t = 0; t < nsteps; ++t) {
    const char* sl = smem + (t & 1) * DWF_SLOT;
    // the previous step's slot is free (barrier): stage the next rows first, the loads of the step after behind them
    if (t + 1 < nsteps) put(t + 1);
    if (t + 2 < nsteps) fetch(t + 2);
    const char* zi = sl + DWF_O_DZ + g4 * DWF_RSZ + (nt * 16 + i16) * 4;
    float bz[8], bzs[TWO ? 8 : 1];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bz[ks] = *(const float*)(zi + 4 * ks * DWF_RSZ);
      if constexpr (TWO) bzs[ks] = *(const float*)(zi + (DWF_O_DZS - DWF_O_DZ) + 4 * ks * DWF_RSZ);
    }
    if (chunk == 0 && cpar == 0) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bsum += bz[ks];
      if constexpr (LRT) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bsumv += bzs[ks];
      }
    }
    const uint32_t* sgl = (const uint32_t*)(sl + DWF_O_SG) + g4 * FDF_SGW;
    // tile m = c-tile cpar + 2 m (tile 7 exists for cpar = 0 only); the column reads (and sign words) of tile m + 1 are
    // issued before the MFMAs of tile m
    struct Op {
      float ax[8];
      uint32_t sw[FO ? 8 : 1];
    };
    auto fetchop = [&](int m, Op& o) __attribute__((always_inline)) {
      const int ct = min(cpar + 2 * m, FDF_KB - 1);
      const char* xi = sl + g4 * DWF_RSX + (ct * 16 + i16) * 4;
      const int c = ch0 + ct * 16 + i16;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        o.ax[ks] = *(const float*)(xi + 4 * ks * DWF_RSX);
        if constexpr (FO) o.sw[ks] = sgl[4 * ks * FDF_SGW + (c >> 5) - w0];
      }
    };
    Op cur, nxt;
    fetchop(0, cur);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m + 1 < 8) fetchop(m + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (m < 7 || cpar == 0) {
        const int c = ch0 + (cpar + 2 * m) * 16 + i16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          acc_a[m] = mfma4(cur.ax[ks], bz[ks], acc_a[m]);
          if constexpr (FO) {
            const uint32_t bit = (cur.sw[ks] >> (c & 31)) & 1u;
            acc_b[m] = mfma4(xor1(cur.ax[ks], bit << 31), bzs[ks], acc_b[m]);
          }
          if constexpr (LRT) acc_b[m] = mfma4(cur.ax[ks] * cur.ax[ks], bzs[ks], acc_b[m]);   // d sigma^2 = dVar^T x^2
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    lds_barrier();
  }
  // ---- flush: transposed tiles (rows = input channels, columns = couts), plain stores.  Row range 0 writes the particle's
  // gradient image, ranges 1 .. nrs - 1 partial images that dense_addn_kernel adds afterwards in a fixed order: float
  // atomics from 200 workgroups onto the same 12 MB cost more than the kernel's MFMAs ----
  const int n = nt * 16 + i16;
  const long pimg = rs == 0 ? s : (long)(rs - 1) * A.S + s;
  float* gwa = (rs == 0 ? A.gw_a : A.gw2_a) + A.gw_stride * pimg + (long)n * A.KP + ch0 + 4 * g4;
  float* gwb = (rs == 0 ? A.gw_b : A.gw2_b) + A.gw_stride * pimg + (long)n * A.KP + ch0 + 4 * g4;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int ct = cpar + 2 * m;
    if (ct < FDF_KB) {
      *(f32x4*)(gwa + ct * 16) = acc_a[m];
      if constexpr (TWO) *(f32x4*)(gwb + ct * 16) = acc_b[m];
    }
  }
  if (chunk == 0 && cpar == 0) {
    float t = bsum;
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (g4 == 0) (rs == 0 ? A.gb_a : A.gb2_a)[(long)A.gb_stride * pimg + n] = t;
    if constexpr (LRT) {
      float tv = bsumv;
      tv += __shfl_xor(tv, 16, 64);
      tv += __shfl_xor(tv, 32, 64);
      if (g4 == 0) (rs == 0 ? A.gb_b : A.gb2_b)[(long)A.gb_stride * pimg + n] = tv;
    }
  }
  __syncthreads();   // the staging slots are reused by the next segment
  }
}

// dst[s][0 .. n) += sum over k < np of src[k * S + s][0 .. n), in that order (strides in floats), for up to four buffers in
// one launch (blockIdx.z: weight images of slots A and B, bias sums); 4 floats per thread
struct DenseAddJobs {
  float* dst[4];
  const float* src[4];
  long n[4], stride[4];
  int S, np;
};
__global__ void dense_addn_kernel(const DenseAddJobs J) {
  const int job = blockIdx.z;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int s = blockIdx.y;
  if (i >= J.n[job] || s >= J.S) return;
  const long stride = J.stride[job];
  f32x4* d = (f32x4*)(J.dst[job] + stride * s + i);
  f32x4 v = *d;
  for (int k = 0; k < J.np; ++k) {
    const f32x4 a = *(const f32x4*)(J.src[job] + stride * ((long)k * J.S + s) + i);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += a[r];
  }
  *d = v;
}

// ==========================================================================================
// densef_fin_kernel : H = relu(bias + sum_chunk partials) of the wide dense layer (chunks summed in order: reproducible),
// and the net's last layer Linear(64, 2) (inception.py:217) on the row while it is in the 16 lanes' registers:
// z = W2 h + b2 (+ Flipout: s_out o (dW2 (h o s_in))).  fp32 twin of dense_ks_fin_kernel.
// ==========================================================================================
struct DenseFinF32Args {
  const float* slab;
  long slab_stride;
  int nchunk;
  int rows, B;          // S*B rows, rows per particle
  const float* bias;    // [S or 1][bias_stride]: offset to the dense layer
  const float* b2;      // same array, offset to the last layer
  int bias_stride;
  float* h;             // [rows][64]
  const float* w2a; const float* w2b;   // forward images of the last layer (slot A | slot B), offset to its first row
  long w2_stride_a, w2_stride_b;
  int w2_KP;
  const uint32_t* sg_in; const uint32_t* sg_out;   // its Flipout sign words [rows][siw] / [rows][sow]
  int siw, sow;
  float* z;             // [rows][2]
  // MC-dropout behind the hidden layer's ReLU (nets/inception.py:205-207: nn.Dropout(p)): rate 0 = off
  float drop_rate, drop_scale;
  uint64_t drop_seed;
  uint32_t drop_step;
  const float* keep_h;  // injected keep mask [rows][64]; null: Philox
  // the last layer's gradient elements, zeroed here for the head launch's atomics (training step): per particle 2 rows of
  // KP in the weight images (slots A and B) and 2 biases; null: nothing to zero
  float* g2_a; float* g2_b; float* g2_ba;
  long g2_stride; int g2_bstride;
  int S;
  // LRT (EM_LRT): partial variances, sigma_b^2 of both layers, the output noise and where q = eps / (2 sd) goes
  const float* slabv;
  const float* biasv; const float* b2v;   // [bias_total] arrays offset to the two layers (shared by the particles)
  NoiseRefs nz; CallGeom cg;
  int layer1, layer2;                     // layer ids (noise streams)
  float* qh; float* qz;                   // [rows][64], [rows][2] (training step; null otherwise)
  float* g2_bb;                           // the last layer's sigma_b^2 gradient elements (zeroed with the others)
};

template <int EM>
__global__ __launch_bounds__(256) void densef_fin_kernel(const DenseFinF32Args F) {
  constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT);
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int row = min(t >> 4, F.rows - 1), ch = (t & 15) * 4;   // surplus threads redo the last row (the shuffles need full groups)
  const bool live = (t >> 4) < F.rows;
  const int s = row / F.B;
  f32x4 v = *(const f32x4*)(F.bias + (long)F.bias_stride * s + ch);
  const float* p = F.slab + (long)row * 64 + ch;
  for (int c = 0; c < F.nchunk; ++c) {
    const f32x4 a = *(const f32x4*)(p + c * F.slab_stride);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += a[r];
  }
  long Rg = 0;
  if constexpr (LRT) {
    // hidden layer: out = loc + sqrt(var) eps (group_fwd_kernel's epilogue), q for the backward
    f32x4 var = *(const f32x4*)(F.biasv + ch);
    const float* pv = F.slabv + (long)row * 64 + ch;
    for (int c = 0; c < F.nchunk; ++c) {
      const f32x4 a = *(const f32x4*)(pv + c * F.slab_stride);
#pragma unroll
      for (int r = 0; r < 4; ++r) var[r] += a[r];
    }
    Rg = global_row(F.cg, 1, row);
    f32x4 eps;
    if (F.nz.use_philox_lrt) {
      const uint64_t idx = (uint64_t)Rg * 16ull + (uint64_t)(ch >> 2);   // cout_p16 / 4 = 16 quads per row
      eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)F.layer1 << 8), F.nz.step, F.nz.seed);
    } else {
      eps = *(const f32x4*)(F.nz.lrt_eps[F.layer1] + (long)row * 64 + ch);
    }
    f32x4 qv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float vv = var[r];
      if (vv < 0.f) vv = 1e-6f;
      const float sd = sqrtf(vv);
      v[r] += sd * eps[r];
      qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
    }
    if (live && F.qh) *(f32x4*)(F.qh + (long)row * 64 + ch) = qv;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  if (F.drop_rate > 0.f) {
    const uint32_t kb = drop_keep4(F.keep_h, (long)row * 64 + ch, F.drop_rate, (uint32_t)row, (uint32_t)(ch >> 2), 3u, F.drop_step, F.drop_seed);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ((kb >> r) & 1u) ? v[r] * F.drop_scale : 0.f;
  }
  if (live) *(f32x4*)(F.h + (long)row * 64 + ch) = v;
  if (F.g2_a && t < F.S * 2 * F.w2_KP) {
    const int zs = t / (2 * F.w2_KP), ze = t - zs * 2 * F.w2_KP;
    F.g2_a[F.g2_stride * zs + ze] = 0.f;
    F.g2_b[F.g2_stride * zs + ze] = 0.f;
    if (ze < 2) F.g2_ba[(long)F.g2_bstride * zs + ze] = 0.f;
    if (LRT && ze < 2) F.g2_bb[(long)F.g2_bstride * zs + ze] = 0.f;
  }
  float m[2] = {0.f, 0.f}, pz[2] = {0.f, 0.f};
  uint32_t bits = 0;
  if constexpr (FO) bits = F.sg_in[(long)row * F.siw + (ch >> 5)] >> (ch & 31);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long wo = (long)k * F.w2_KP + ch;
    const f32x4 wa = *(const f32x4*)(F.w2a + F.w2_stride_a * s + wo);
#pragma unroll
    for (int r = 0; r < 4; ++r) m[k] += v[r] * wa[r];
    if constexpr (FO) {
      const f32x4 wb = *(const f32x4*)(F.w2b + F.w2_stride_b * s + wo);
#pragma unroll
      for (int r = 0; r < 4; ++r) pz[k] += (((bits >> r) & 1u) ? -v[r] : v[r]) * wb[r];
    }
    if constexpr (LRT) {   // var_k = sum_c h_c^2 sigma^2[k][c]
      const f32x4 wb = *(const f32x4*)(F.w2b + F.w2_stride_b * s + wo);
#pragma unroll
      for (int r = 0; r < 4; ++r) pz[k] += v[r] * v[r] * wb[r];
    }
  }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      m[k] += __shfl_xor(m[k], d, 16);
      if constexpr (FO || LRT) pz[k] += __shfl_xor(pz[k], d, 16);
    }
  }
  if (live && (t & 15) == 0) {
    uint32_t so = 0;
    if constexpr (FO) so = F.sg_out[(long)row * F.sow];
    f32x4 eps2 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LRT) {
      if (F.nz.use_philox_lrt) {
        const uint64_t idx = (uint64_t)Rg * 4ull;   // cout_p16 / 4 = 4 quads per row, couts 0, 1 in quad 0
        eps2 = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)F.layer2 << 8), F.nz.step, F.nz.seed);
      } else {
        eps2[0] = F.nz.lrt_eps[F.layer2][(long)row * 2];
        eps2[1] = F.nz.lrt_eps[F.layer2][(long)row * 2 + 1];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float zk = m[k] + F.b2[(long)F.bias_stride * s + k];
      if constexpr (FO) zk += ((so >> k) & 1u) ? -pz[k] : pz[k];
      if constexpr (LRT) {
        float vv = pz[k] + F.b2v[k];
        if (vv < 0.f) vv = 1e-6f;
        const float sd = sqrtf(vv);
        zk += sd * eps2[k];
        if (F.qz) F.qz[(long)row * 2 + k] = sd > 0.f ? eps2[k] / (2.f * sd) : 0.f;
      }
      F.z[(long)row * 2 + k] = zk;
    }
  }
}

// ==========================================================================================
// headf_last_kernel : head + backward of the last layer Linear(64, 2) in one launch (fp32 twin of head_last_kernel):
// per example row the head's d(-ll)/dz, then dH = dz W2 (+ Flipout: s_in o ((dz o s_out) dW2)) written as the fp32
// gradient of H, and per workgroup the layer's weight / bias gradient sums over its rows (fp32 atomics into the zeroed
// per-particle images).
// ==========================================================================================
struct HeadLastF32Args {
  HeadArgs H;
  const float* wa; const float* wb;      // forward images of the layer (slot A | slot B), offset to its first row
  long stride_a, stride_b;
  int KP;
  const float* h;                        // [S*B][64]
  float* dh;                             // [S*B][64]
  const uint32_t* sg_in; const uint32_t* sg_out;
  int siw, sow;
  float* gw_a; float* gw_b; float* gb_a;   // offset to the layer; per particle strides below
  long gw_stride; int gb_stride;
  const float* qz;                         // LRT: q of the net outputs [S*B][2]; dVar = dz q
  float* gb_b;                             // LRT: gradient of sigma_b^2
};

template <int EM>
__global__ __launch_bounds__(256) void headf_last_kernel(const HeadLastF32Args A) {
  constexpr bool FO = (EM == EM_FLIPOUT), LRT = (EM == EM_LRT), TWO = FO || LRT;
  const int tid = threadIdx.x;
  const int blk0 = blockIdx.x * HL_ROWS;
  const int idx = blk0 + tid;   // rows: threads 0 .. HL_ROWS-1
  const int s = blockIdx.y;
  const int B = A.H.B;
  __shared__ float dzs[4][HL_ROWS];
  __shared__ float wsh[4][64];   // W rows 0, 1 | dW rows 0, 1
  {
    const int kk = tid >> 6, c = tid & 63;
    const float* src = (kk < 2 ? A.wa + A.stride_a * s : A.wb + A.stride_b * s) + (long)(kk & 1) * A.KP + c;
    wsh[kk][c] = (kk < 2 || TWO) ? *src : 0.f;
  }
  __syncthreads();
  if (tid < HL_ROWS) {
    double ll = 0.0;
    float g0 = 0.f, g1 = 0.f;
    if (idx < B) ll = head_row(A.H, s, idx, g0, g1);
    ll = wave_sum_d(ll);   // HL_ROWS = one wave
    if (tid == 0 && A.H.with_obs) atomicAdd(A.H.ll_acc + s, ll);
    const long r = (long)s * B + min(idx, B - 1);
    uint32_t so = 0;
    if constexpr (FO) so = A.sg_out[r * A.sow];
    float h0 = (so & 1u) ? -g0 : g0, h1 = (so & 2u) ? -g1 : g1;   // dz o s_out
    if constexpr (LRT) {   // dVar = dz q
      h0 = g0 * A.qz[r * 2];
      h1 = g1 * A.qz[r * 2 + 1];
    }
    dzs[0][tid] = g0;
    dzs[1][tid] = g1;
    dzs[2][tid] = h0;
    dzs[3][tid] = h1;
    if (idx < B) {
      uint32_t si[2] = {0u, 0u};
      if constexpr (FO) {
        si[0] = A.sg_in[r * A.siw];
        si[1] = A.sg_in[r * A.siw + 1];
      }
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) {
        f32x4 d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c4 * 4 + q;
          d[q] = g0 * wsh[0][c] + g1 * wsh[1][c];
          if constexpr (LRT) d[q] += 2.f * A.h[r * 64 + c] * (h0 * wsh[2][c] + h1 * wsh[3][c]);   // + 2 h o (sigma^2^T dVar)
          if constexpr (FO) {
            const float pp = h0 * wsh[2][c] + h1 * wsh[3][c];
            d[q] += ((si[c >> 5] >> (c & 31)) & 1u) ? -pp : pp;
          }
        }
        *(f32x4*)(A.dh + r * 64 + c4 * 4) = d;
      }
    }
  }
  __syncthreads();
  // ---- weight / bias gradient sums of this workgroup's rows: thread = (kind kk, channel c) ----
  const int kk = tid >> 6, c = tid & 63;   // kk 0, 1: d/dW_a rows 0, 1; kk 2, 3: d/dW_b rows 0, 1 (Flipout)
  const int nrow = min(HL_ROWS, B - blk0);
  if (kk < 2 || TWO) {
    float acc = 0.f;
    const float* hp = A.h + ((long)s * B + blk0) * 64 + c;
    const uint32_t* sp = A.sg_in + ((long)s * B + blk0) * A.siw + (c >> 5);
    for (int r0 = 0; r0 < HL_ROWS; r0 += 16) {   // 16 rows of loads in flight
      float hv[16];
      uint32_t sw[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int rr = min(r0 + j, nrow - 1);
        hv[j] = hp[(long)rr * 64];
        sw[j] = (FO && kk >= 2) ? sp[(long)rr * A.siw] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float h = hv[j];
        if ((sw[j] >> (c & 31)) & 1u) h = -h;
        if (LRT && kk >= 2) h = h * h;   // d sigma^2 = dVar^T h^2
        if (r0 + j < nrow) acc += dzs[kk][r0 + j] * h;
      }
    }
    float* g = (kk < 2 ? A.gw_a : A.gw_b) + A.gw_stride * s + (long)(kk & 1) * A.KP + c;
    atomicAdd(g, acc);
  }
  if (tid < 2) {   // bias gradient: sum of dz (Flipout's arrives through slot A only)
    float acc = 0.f;
    for (int rr = 0; rr < nrow; ++rr) acc += dzs[tid][rr];
    atomicAdd(A.gb_a + (long)A.gb_stride * s + tid, acc);
  }
  if (LRT && tid >= 2 && tid < 4) {   // gradient of sigma_b^2: sum of dVar
    float acc = 0.f;
    for (int rr = 0; rr < nrow; ++rr) acc += dzs[tid][rr];
    atomicAdd(A.gb_b + (long)A.gb_stride * s + (tid - 2), acc);
  }
}
