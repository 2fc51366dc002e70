// Group kernels: variational forward, dX and dW of one branch group (see desc.h).
#pragma once
#include "kernels_core.h"

struct GroupArgs {
  GroupDesc g;
  CallGeom cg;
  WeightSlots ws;
  NoiseRefs nz;
  TensorRef t[BNN_MAX_TENSORS];
  const LayerDesc* layers;  // device table
  // dW outputs (per particle): fp32 images in the forward layout + bias gradients
  float* gw_a;
  float* gw_b;
  float* gb_a;
  float* gb_b;
  long gw_stride;  // elements between particles
  int gb_stride;
  int pool_sel;    // dx: 0 = branches reading the tensor directly, 1 = pooled branches
  int nsplit;      // dw: window splits per (job, particle)
  int lds_per_wave;  // bytes
  unsigned char* amax;     // conv groups with a pooled branch: arg-max code (0 row-1, 1 row, 2 row+1) of
                           // MaxPool1d(3,1,1) per element of the input tensor, [windows * L][in_cin_p]; written by the
                           // forward, read by the fused dX
  int dbg_block;           // diagnostics: workgroup whose waves record stamps
  unsigned long long* dbg;  // diagnostics only: phase time stamps of workgroup 0 ([wave][48 windows][8 phases]); null = off
};

template <class P>
struct WaveLds {
  typename P::elem* img[4];
  __device__ WaveLds(char* base, int n_img, int RSmax) {
    const int bytes = IMG_ROWS * RSmax * (int)sizeof(typename P::elem);
    const int stride = (bytes + 15) & ~15;
    for (int k = 0; k < 4; ++k) img[k] = (typename P::elem*)(base + (k < n_img ? k : 0) * stride);
  }
};

template <class P>
__device__ __forceinline__ const typename P::elem* slot_ptr(const void* base, long per_particle, int s, long layer_off,
                                                            long row0, int KP) {
  return (const typename P::elem*)base + per_particle * s + layer_off + row0 * KP;
}

// ==========================================================================================
// forward
// ==========================================================================================
template <class P, int EM>
__global__ __launch_bounds__(256) void group_fwd_kernel(const GroupArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool SPLIT = P::BF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const GroupDesc& G = A.g;
  const int RSmax = img_row_stride(DENSE_CHUNK, P::BF);
  WaveLds<P> L(smem + wave * A.lds_per_wave, 4, RSmax);
  typename P::elem* x0 = L.img[0];
  typename P::elem* x1 = P::BF ? L.img[1] : nullptr;
  typename P::elem* x2 = P::BF ? L.img[2] : L.img[1];

  // work item = (window, branch): a dense group of the Linear net has only B/32 windows per particle, so the
  // branches of a window go to different waves (each stages the window chunk it needs itself)
  const int nitem = A.cg.nwin * G.n_branch;
  for (int item = blockIdx.x * nwaves + wave; item < nitem; item += gridDim.x * nwaves) {
    const int win = item / G.n_branch;
    const Win W = decode_win(G, A.cg, win);
    const TensorRef tin = A.t[G.in_t];
    int staged_off = -1, staged_pool = -1, staged_c0 = -1;
    {
      const int b = item - win * G.n_branch;
      const BranchDesc& br = G.br[b];
      const LayerDesc& ly = A.layers[br.layer];
      f32x4 acc_a[4][2], acc_b[4][2];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc_a[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc_b[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      const typename P::elem* wa = slot_ptr<P>(A.ws.a_hi, A.ws.slot_stride_a, W.s, ly.w_off, br.n_off, ly.KP);
      const typename P::elem* wlo =
          P::BF ? slot_ptr<P>(A.ws.a_lo, A.ws.slot_stride_a, W.s, ly.w_off, br.n_off, ly.KP) : nullptr;
      const typename P::elem* wb =
          DUAL ? slot_ptr<P>(A.ws.b, A.ws.slot_stride_b, W.s, ly.w_off, br.n_off, ly.KP) : nullptr;
      for (int c0 = 0; c0 < br.cin_p; c0 += DENSE_CHUNK) {
        const int cwp = min(DENSE_CHUNK, br.cin_p - c0);
        const int RS = img_row_stride(cwp, P::BF);
        const bool need = (EM == EM_FLIPOUT) || staged_off != br.in_off || staged_pool != br.pool || staged_c0 != c0;
        if (need) {
          StageSpec sp;
          sp.src = tin;
          sp.row0 = W.in_row0;
          sp.ctot = tin.ctot;
          sp.coff = br.in_off + c0;
          sp.cw = br.cin_real - c0;
          sp.cwp = cwp;
          sp.nvalid = W.nvalid;
          sp.pool = br.pool;
          sp.mask = TensorRef{nullptr, nullptr, 0, 0};
          sp.mul = TensorRef{nullptr, nullptr, 0, 0};
          sp.sign = nullptr;
          sp.sign_stride = 0;
          sp.sign_per_row = G.is_dense;
          sp.sign_coff = c0;
          sp.second = SEC_NONE;
          if (EM == EM_LRT) sp.second = SEC_SQUARE;
          if (EM == EM_FLIPOUT) {
            sp.second = SEC_SIGN;
            sp.sign = A.nz.sign_in + ly.sign_in_off * A.nz.examples + (long)W.ex0 * ly.sign_in_words;
            sp.sign_stride = ly.sign_in_words;
          }
          wave_lds_sync();  // previous readers of the images are done
          stage_window<P>(sp, x0, x1, x2, RS, lane);
          wave_lds_sync();
          staged_off = br.in_off;
          staged_pool = br.pool;
          staged_c0 = c0;
        }
        gemm_f<P, 4, DUAL, SPLIT>(acc_a, acc_b, br.ntiles, wa, wlo, wb, ly.KP, c0, ly.taps, ly.pad, cwp, x0, x1, x2, RS,
                                  lane);
      }
      // ---------------- epilogue ----------------
      const TensorRef tout = A.t[br.out_t];
      const float* bias_a = A.ws.bias_a + (long)A.ws.bias_stride_a * W.s + ly.bias_off + br.n_off;
      const float* bias_b = (EM == EM_LRT) ? A.ws.bias_b + ly.bias_off + br.n_off : nullptr;
      const int j = lane & 15, g = lane >> 4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= br.ntiles) continue;
        const int chb = nt * 16 + 4 * g;
        if (chb >= br.cout) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + j;
          if (row >= W.nvalid) continue;
          const int R = W.out_row0 + row;
          f32x4 v = acc_a[nt][mt];
          f32x4 qv = {0.f, 0.f, 0.f, 0.f};
          if constexpr (EM == EM_LRT) {
            f32x4 eps;
            const int lch = br.n_off + chb;  // channel inside the layer
            if (A.nz.use_philox_lrt) {
              const int Lrows = G.is_dense ? 1 : G.L;
              const long Rg = global_row(A.cg, Lrows, R);
              const uint64_t idx = (uint64_t)Rg * (uint64_t)(ly.cout_p16 >> 2) + (uint64_t)(lch >> 2);
              eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)br.layer << 8),
                                   A.nz.step, A.nz.seed);
            } else {
              const float* e = A.nz.lrt_eps[br.layer] + (long)R * ly.cout + lch;
#pragma unroll
              for (int r = 0; r < 4; ++r) eps[r] = (chb + r < br.cout) ? e[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (chb + r < br.cout) {
                const float loc = v[r] + bias_a[chb + r];
                float var = acc_b[nt][mt][r] + bias_b[chb + r];
                if (var < 0.f) var = 1e-6f;  // var + (var<0)*(|var|+1e-6)
                const float sd = sqrtf(var);
                v[r] = loc + sd * eps[r];
                qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
              }
            }
          } else if constexpr (EM == EM_FLIPOUT) {
            const int bit0 = br.n_off + chb;
            const int ex = G.is_dense ? (W.ex0 + row) : W.ex0;
            const uint32_t word = A.nz.sign_out[ly.sign_out_off * A.nz.examples + (long)ex * ly.sign_out_words + (bit0 >> 5)];
            const uint32_t bits = word >> (bit0 & 31);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (chb + r < br.cout) {
                const float pb = acc_b[nt][mt][r];
                v[r] = v[r] + bias_a[chb + r] + (((bits >> r) & 1u) ? -pb : pb);
              }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (chb + r < br.cout) v[r] += bias_a[chb + r];
          }
          if (br.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          const long oo = (long)R * tout.ctot + br.out_off + chb;
          const bool full = ((tout.ctot & 3) == 0) && ((br.out_off & 3) == 0);
          tstore4(tout, oo, v, br.cout - chb, full);
          if constexpr (EM == EM_LRT) {
            if (br.q_t >= 0) {
              const TensorRef tq = A.t[br.q_t];
              tstore4(tq, (long)R * tq.ctot + br.out_off + chb, qv, br.cout - chb, full);
            }
          }
        }
      }
    }
  }
}

// ==========================================================================================
// dX : gradient w.r.t. the group's input tensor (or its pooled copy)
// ==========================================================================================
template <class P, int EM>
__global__ __launch_bounds__(256) void group_dx_kernel(const GroupArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const GroupDesc& G = A.g;
  const int RSmax = img_row_stride(DENSE_CHUNK, P::BF);
  WaveLds<P> L(smem + wave * A.lds_per_wave, 2, RSmax);
  typename P::elem* z0 = L.img[0];
  typename P::elem* z2 = L.img[1];
  const int j = lane & 15, g = lane >> 4;

  // target tensor = dx_t of the selected branches
  int dx_t = -1;
  for (int b = 0; b < G.n_branch; ++b)
    if (G.br[b].pool == A.pool_sel && G.br[b].dx_t >= 0) dx_t = G.br[b].dx_t;
  if (dx_t < 0) return;
  const TensorRef tdx = A.t[dx_t];
  const TensorRef tin = A.t[G.in_t];

  for (int win = blockIdx.x * nwaves + wave; win < A.cg.nwin; win += gridDim.x * nwaves) {
    const Win W = decode_win(G, A.cg, win);
    int staged_b = -1;
    for (int oc0 = 0; oc0 < G.in_cin_p; oc0 += DENSE_CHUNK) {
      const int ocw = min(DENSE_CHUNK, G.in_cin_p - oc0);
      f32x4 acc_t[8][2];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc_t[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

      for (int b = 0; b < G.n_branch; ++b) {
        const BranchDesc& br = G.br[b];
        if (br.pool != A.pool_sel || br.dx_t < 0) continue;
        // intersection of the branch input range with this output chunk
        const int lo = max(br.in_off, oc0), hi = min(br.in_off + br.cin_p, oc0 + ocw);
        if (lo >= hi) continue;
        const LayerDesc& ly = A.layers[br.layer];
        const int cwp = G.is_dense ? ((br.cout + 31) & ~31) : ((br.cout + 7) & ~7);
        const int RS = img_row_stride(cwp, P::BF);
        if (staged_b != b) {
          const TensorRef tg = A.t[br.out_t + T_GRAD];
          const TensorRef ty = A.t[br.out_t];
          StageSpec sp;
          sp.src = tg;
          sp.row0 = W.out_row0;
          sp.ctot = tg.ctot;
          sp.coff = br.out_off;
          sp.cw = br.cout;
          sp.cwp = cwp;
          sp.nvalid = W.nvalid;
          sp.pool = 0;
          sp.mask = br.relu ? ty : TensorRef{nullptr, nullptr, 0, 0};
          sp.mul = TensorRef{nullptr, nullptr, 0, 0};
          sp.sign = nullptr;
          sp.sign_stride = 0;
          sp.sign_per_row = G.is_dense;
          sp.sign_coff = br.n_off;
          sp.second = SEC_NONE;
          if (EM == EM_LRT) {
            const TensorRef tq = A.t[br.q_t];
            sp.second = SEC_MUL;
            sp.mul = tq;
          }
          if (EM == EM_FLIPOUT) {
            sp.second = SEC_SIGN;
            sp.sign = A.nz.sign_out + ly.sign_out_off * A.nz.examples + (long)W.ex0 * ly.sign_out_words;
            sp.sign_stride = ly.sign_out_words;
          }
          wave_lds_sync();
          stage_window<P>(sp, z0, nullptr, z2, RS, lane);
          wave_lds_sync();
          staged_b = b;
        }
        // rows of the transposed image = input channels of the layer
        const int crow0 = lo - br.in_off;          // first layer-input channel handled here
        const int tile0 = (lo - oc0) >> 4;         // first accumulator tile
        const int ntl = (hi - lo + 15) >> 4;
        const typename P::elem* wat = slot_ptr<P>(A.ws.at, A.ws.slott_stride_a, W.s, ly.wt_off, crow0, ly.KPt);
        const typename P::elem* wbt =
            DUAL ? slot_ptr<P>(A.ws.bt, A.ws.slott_stride_b, W.s, ly.wt_off, crow0, ly.KPt) : nullptr;
        f32x4 acc_a[8][2], acc_b[8][2];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            acc_a[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_b[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        gemm_f<P, 8, DUAL, false>(acc_a, acc_b, ntl, wat, nullptr, wbt, ly.KPt, br.n_off, ly.taps, ly.pad, cwp, z0,
                                  nullptr, z2, RS, lane);
        // fold into the chunk accumulator
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          if (nt >= ntl) continue;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x4 add = acc_a[nt][mt];
            if constexpr (DUAL) {
              const int row = mt * 16 + j;
              const int cimg = lo + nt * 16 + 4 * g;  // channel inside the input tensor view
              if (row < W.nvalid) {
                if constexpr (EM == EM_LRT) {
                  // dX += 2 * X * (Wb^T dB);  X = (pooled) input value
                  const long o = (long)(W.in_row0 + row) * tin.ctot + cimg;
                  const bool ok = (cimg + 4 <= br.in_off + br.cin_real);
                  f32x4 xv = {0.f, 0.f, 0.f, 0.f};
                  if (ok) {
                    xv = tload4(tin, o, 4, true);
                    if (br.pool) {
                      if (row > 0) {
                        const f32x4 a = tload4(tin, o - tin.ctot, 4, true);
#pragma unroll
                        for (int r = 0; r < 4; ++r) xv[r] = fmaxf(xv[r], a[r]);
                      }
                      if (row + 1 < W.nvalid) {
                        const f32x4 a = tload4(tin, o + tin.ctot, 4, true);
#pragma unroll
                        for (int r = 0; r < 4; ++r) xv[r] = fmaxf(xv[r], a[r]);
                      }
                    }
                  }
                  if constexpr (P::BF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[r] = bf2f(f2bf(xv[r]));
                  }
#pragma unroll
                  for (int r = 0; r < 4; ++r) add[r] += 2.f * xv[r] * acc_b[nt][mt][r];
                } else {
                  const int bit0 = cimg - br.in_off;
                  const int ex = G.is_dense ? (W.ex0 + row) : W.ex0;
                  const uint32_t word = A.nz.sign_in[ly.sign_in_off * A.nz.examples + (long)ex * ly.sign_in_words + (bit0 >> 5)];
                  const uint32_t bits = word >> (bit0 & 31);
#pragma unroll
                  for (int r = 0; r < 4; ++r) add[r] += ((bits >> r) & 1u) ? -acc_b[nt][mt][r] : acc_b[nt][mt][r];
                }
              }
            }
            // static index into acc_t: tile0 is wave-uniform but not a constant -> select
#pragma unroll
            for (int tt = 0; tt < 8; ++tt)
              if (tt == tile0 + nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_t[tt][mt][r] += add[r];
              }
          }
        }
      }
      // store the chunk
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int ch = oc0 + nt * 16 + 4 * g;
        if (nt * 16 >= ocw || ch + 4 > tdx.ctot) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + j;
          if (row >= W.nvalid) continue;
          tstore4(tdx, (long)(W.in_row0 + row) * tdx.ctot + ch, acc_t[nt][mt], 4, true);
        }
      }
    }
  }
}

// ==========================================================================================
// dW : weight-image gradients, reduced over the windows of one particle
// ==========================================================================================
template <class P, int EM, int NW>
__global__ __launch_bounds__(NW * 64) void group_dw_kernel(const GroupArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr int TPW = 32 / NW;  // tiles per wave (<= 32 tiles per job)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const GroupDesc& G = A.g;

  // decode block -> (job = branch x cin-chunk, particle, split)
  int bid = blockIdx.x;
  const int split = bid % A.nsplit;
  bid /= A.nsplit;
  const int s = bid % A.cg.S;
  int job = bid / A.cg.S;
  int b = 0, c0 = 0;
  for (b = 0; b < G.n_branch; ++b) {
    const int nch = (G.br[b].cin_p + DENSE_CHUNK - 1) / DENSE_CHUNK;
    if (job < nch) {
      c0 = job * DENSE_CHUNK;
      break;
    }
    job -= nch;
  }
  if (b >= G.n_branch) return;
  const BranchDesc& br = G.br[b];
  const LayerDesc& ly = A.layers[br.layer];
  const int cwp = min(DENSE_CHUNK, br.cin_p - c0);
  const int ctiles = (cwp + 15) >> 4;
  const int ntiles_total = br.ntiles * ly.taps * ctiles;
  const int zwp = (br.cout + 15) & ~15;
  const int RSz = img_row_stride(zwp, P::BF);
  const int RSx = img_row_stride((cwp + 15) & ~15, P::BF);

  // LDS carve: per wave [dz | dz2 | x | x2]
  const int zbytes = (IMG_ROWS * img_row_stride(64, P::BF) * (int)sizeof(typename P::elem) + 15) & ~15;
  const int xbytes = (IMG_ROWS * img_row_stride(DENSE_CHUNK, P::BF) * (int)sizeof(typename P::elem) + 15) & ~15;
  auto wbase = [&](int w) { return smem + w * A.lds_per_wave; };
  auto dz_of = [&](int w) { return (typename P::elem*)(wbase(w)); };
  auto dz2_of = [&](int w) { return (typename P::elem*)(wbase(w) + zbytes); };
  auto x_of = [&](int w) { return (typename P::elem*)(wbase(w) + 2 * zbytes); };
  auto x2_of = [&](int w) { return (typename P::elem*)(wbase(w) + 2 * zbytes + xbytes); };

  f32x4 acc_a[TPW], acc_b[TPW];
#pragma unroll
  for (int m = 0; m < TPW; ++m) {
    acc_a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gb_a = 0.f, gb_b = 0.f;  // bias gradient of channel threadIdx.x (c0 == 0 jobs only)

  const TensorRef tin = A.t[G.in_t];
  const TensorRef tg = A.t[br.out_t + T_GRAD];
  const TensorRef ty = A.t[br.out_t];
  const int pp = A.cg.per_particle;
  const int rounds = (pp + A.nsplit * NW - 1) / (A.nsplit * NW);
  for (int rd = 0; rd < rounds; ++rd) {
    const int wl = (rd * A.nsplit + split) * NW + wave;  // window inside the particle
    __syncthreads();  // images of the previous round fully consumed
    {
      Win W;
      StageSpec sz, sx;
      const bool live = wl < pp;
      if (live) W = decode_win(G, A.cg, s * pp + wl);
      // dZ (+ dZ*q | dZ*s_out)
      sz.src = tg;
      sz.row0 = live ? W.out_row0 : 0;
      sz.ctot = tg.ctot;
      sz.coff = br.out_off;
      sz.cw = br.cout;
      sz.cwp = zwp;
      sz.nvalid = live ? W.nvalid : 0;
      sz.pool = 0;
      sz.mask = (br.relu && live) ? ty : TensorRef{nullptr, nullptr, 0, 0};
      sz.mul = TensorRef{nullptr, nullptr, 0, 0};
      sz.sign = nullptr;
      sz.sign_stride = 0;
      sz.sign_per_row = G.is_dense;
      sz.sign_coff = br.n_off;
      sz.second = SEC_NONE;
      if (EM == EM_LRT && live) {
        const TensorRef tq = A.t[br.q_t];
        sz.second = SEC_MUL;
        sz.mul = tq;
      }
      if (EM == EM_FLIPOUT && live) {
        sz.second = SEC_SIGN;
        sz.sign = A.nz.sign_out + ly.sign_out_off * A.nz.examples + (long)W.ex0 * ly.sign_out_words;
        sz.sign_stride = ly.sign_out_words;
      }
      if (DUAL && !live) sz.second = SEC_SQUARE;  // zeros
      stage_window<P>(sz, dz_of(wave), nullptr, dz2_of(wave), RSz, lane);
      // X (+ X^2 | X*s_in)
      sx.src = tin;
      sx.row0 = live ? W.in_row0 : 0;
      sx.ctot = tin.ctot;
      sx.coff = br.in_off + c0;
      sx.cw = br.cin_real - c0;
      sx.cwp = (cwp + 15) & ~15;
      sx.nvalid = live ? W.nvalid : 0;
      sx.pool = br.pool;
      sx.mask = TensorRef{nullptr, nullptr, 0, 0};
      sx.mul = TensorRef{nullptr, nullptr, 0, 0};
      sx.sign = nullptr;
      sx.sign_stride = 0;
      sx.sign_per_row = G.is_dense;
      sx.sign_coff = c0;
      sx.second = SEC_NONE;
      if (EM == EM_LRT) sx.second = SEC_SQUARE;
      if (EM == EM_FLIPOUT) {
        if (live) {
          sx.second = SEC_SIGN;
          sx.sign = A.nz.sign_in + ly.sign_in_off * A.nz.examples + (long)W.ex0 * ly.sign_in_words;
          sx.sign_stride = ly.sign_in_words;
        } else {
          sx.second = SEC_SQUARE;
        }
      }
      stage_window<P>(sx, x_of(wave), nullptr, x2_of(wave), RSx, lane);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TPW; ++m) {
      const int t = wave + NW * m;
      if (t < ntiles_total) {
        const int ct = t % ctiles;
        const int tap = (t / ctiles) % ly.taps;
        const int nt = t / (ctiles * ly.taps);
        for (int w = 0; w < NW; ++w) {
          acc_a[m] = gemm_w_tile<P>(acc_a[m], dz_of(w), RSz, nt * 16, x_of(w), RSx, ct * 16, tap - ly.pad, lane);
          if constexpr (DUAL)
            acc_b[m] = gemm_w_tile<P>(acc_b[m], dz2_of(w), RSz, nt * 16, x2_of(w), RSx, ct * 16, tap - ly.pad, lane);
        }
      }
    }
    if (c0 == 0 && (int)threadIdx.x < br.cout) {
      for (int w = 0; w < NW; ++w) {
        const typename P::elem* dz = dz_of(w);
        const typename P::elem* dz2 = dz2_of(w);
        for (int r = 0; r < TILE_ROWS; ++r) {
          const int o = (r + HALO) * RSz + threadIdx.x;
          if constexpr (P::BF) {
            gb_a += bf2f(dz[o]);
            if (DUAL) gb_b += bf2f(dz2[o]);
          } else {
            gb_a += dz[o];
            if (DUAL) gb_b += dz2[o];
          }
        }
      }
    }
  }
  // ---- write out (atomics: several splits / jobs of other particles never collide) ----
  float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
  float* gwb = DUAL ? A.gw_b + A.gw_stride * s + ly.w_off : nullptr;
  const int i4 = 4 * (lane >> 4), jc = lane & 15;
#pragma unroll
  for (int m = 0; m < TPW; ++m) {
    const int t = wave + NW * m;
    if (t >= ntiles_total) continue;
    const int ct = t % ctiles;
    const int tap = (t / ctiles) % ly.taps;
    const int nt = t / (ctiles * ly.taps);
    const int c = c0 + ct * 16 + jc;
    if (c >= br.cin_p) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      const long o = (long)(br.n_off + n) * ly.KP + (long)tap * ly.cin_img + c;
      atomicAdd(gwa + o, acc_a[m][r]);
      if constexpr (DUAL) atomicAdd(gwb + o, acc_b[m][r]);
    }
  }
  if (c0 == 0 && (int)threadIdx.x < br.cout) {
    atomicAdd(A.gb_a + (long)A.gb_stride * s + ly.bias_off + br.n_off + threadIdx.x, gb_a);
    if (EM == EM_LRT) atomicAdd(A.gb_b + (long)A.gb_stride * s + ly.bias_off + br.n_off + threadIdx.x, gb_b);
  }
}
