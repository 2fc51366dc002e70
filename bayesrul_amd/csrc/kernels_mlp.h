// Fused LRT kernels of the Linear net (bayesrul/models/nets/linear.py:44-71: Flatten -> 540 -> 256 -> 128 -> 128 -> 32
// -> 2, ReLU between), bf16 planes, gfx950.
//
// One SVI step on this net is a few MFLOP per window: the per-layer kernels of the generic path spend their time in
// launches, LDS fills and pipeline prologues.  Here the step's dense work is four launches:
//   mlp_l0_kernel  : layer 0 (72 % of the weights) for one (32-row window, 64-cout group): the group's weight fragments
//                    (mu hi / lo, sigma^2) are ALL in registers before the first MFMA, x is converted fp32 -> hi / lo /
//                    squared planes in LDS on the way;
//   mlp_l14_kernel : layers 1..4 for one window, activations never leave LDS between layers; every wave holds the
//                    fragments of its n-tile of every layer from the start (straight-line code: hipcc counts vmcnt
//                    exactly, the loads of layer l+1 land under layer l);
//   mlp_dx_kernel  : the dX chain 4 -> 1 for one window (transposed mu / sigma^2 fragments in registers), writes the
//                    masked dz = dY [Y > 0] and dz q of every layer for the dW kernel;
//   mlp_dw_kernel  : dW of all five layers in one launch, a (layer, 64-cout group, 128-cin chunk, window split) per
//                    workgroup: the dense_ks_bwd structure (X ring by LDS-DMA, builder waves, transposed reads).
// LRT estimator (bayesian.py / tyxe LocalReparameterization): out = x mu^T + b + sqrt(x^2 (sigma^2)^T + sigma_b^2) eps.
// Mean path split-bf16 (hi*hi + hi*lo + lo*hi), variance path and the backward single bf16, fp32 accumulation.
#pragma once
// diagnostics builds only (tests/probes/ablate_gpu.sh): timing with parts of the kernels removed; results are wrong.
#ifndef ML_ABL
#define ML_ABL 0
#endif

// forward and dX chain: windows of MLF_ROWS rows (their cost is VALU issue in the epilogues, so more, smaller workgroups);
// dW: windows of ML_ROWS rows (one MFMA k-step of the row contraction)
enum { MLF_MT = 1, MLF_ROWS = 16 * MLF_MT };
enum { ML_ROWS = 32, ML_K0 = 544, ML_N0 = 256, ML_N1 = 128, ML_N2 = 128, ML_N3 = 32, ML_N4 = 2 };
// LDS activation image: three planes (hi | lo | squared hi), ML_ROWS rows, pitch = K * 2 + 32 bytes: pitch / 16 = 2 (mod 4)
// for every K here, conflict-free for the ds_read_b128 lane groups (see TR_RSB in kernels_trunk.h)
__host__ __device__ constexpr int ml_pitch(int K) { return K * 2 + 32; }
__host__ __device__ constexpr int ml_plane(int K) { return ML_ROWS * ml_pitch(K); }
__host__ __device__ constexpr int mlf_plane(int K) { return MLF_ROWS * ml_pitch(K); }

struct MlpPlan {
  LayerDesc ly[5];     // by value: kernel arguments, not global loads (see DenseKsPlan)
  TensorRef h[5];      // outputs of the five layers (h[4] = z, fp32), bf16 hi / lo planes
  TensorRef g[5];      // their gradients: the backward stores the MASKED gradient dz of layers 0..3 here
  TensorRef q[5];      // q = eps / (2 sd) of the forward; overwritten with dz q by the backward
  u16* xhi;            // [B][544] hi plane of x (written by layer 0's forward, read by its dW)
  u16* dz4;            // [rows][8] dz of the last layer as bf16, followed by [rows][8] dz q
  long dz4_plane;      // elements between the two
  // backward only: gradient images the dX kernel zeroes for the dW kernel (16-byte units), and the head it runs first
  float* zero_p[4];
  long zero_n[4];
  HeadArgs head;
  int fuse_head;
};

template <int KS>
struct MlFrag {
  bf16x8 hi[KS], lo[KS], b[KS];
};

// fragments of n-tile `nt` of a layer (forward images: [cout][KP])
template <int KS>
__device__ __forceinline__ void ml_load(MlFrag<KS>& f, const WeightSlots& ws, const LayerDesc& ly, int nt, int lane) {
  const int i16 = lane & 15, g4 = lane >> 4;
  const long row = (long)(nt * 16 + i16) * ly.KP + ly.w_off + g4 * 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if constexpr (ML_ABL & 8) {
      f.hi[ks] = f.lo[ks] = f.b[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      continue;
    }
    f.hi[ks] = *(const bf16x8*)((const u16*)ws.a_hi + row + ks * 32);
    f.lo[ks] = *(const bf16x8*)((const u16*)ws.a_lo + row + ks * 32);
    f.b[ks] = *(const bf16x8*)((const u16*)ws.b + row + ks * 32);
  }
}

// mean (three independent accumulation chains, summed by the caller) and variance contractions of one n-tile over the
// window image at `in` (hi plane; lo and squared planes `plane` and 2 * `plane` bytes further)
template <int KS>
__device__ __forceinline__ void ml_mma(const MlFrag<KS>& f, const char* in, int pitch, int plane, int lane, f32x4 (&am)[MLF_MT], f32x4 (&av)[MLF_MT]) {
  const int i16 = lane & 15, g4 = lane >> 4;
  const char* lb = in + i16 * pitch + g4 * 16;
  f32x4 a1[MLF_MT], a2[MLF_MT];
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt) am[mt] = av[mt] = a1[mt] = a2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < ((ML_ABL & 2) ? 0 : KS); ++ks)
#pragma unroll
    for (int mt = 0; mt < MLF_MT; ++mt) {
      const char* p = lb + mt * 16 * pitch + ks * 64;
      const bf16x8 bh = *(const bf16x8*)p;
      const bf16x8 bl = *(const bf16x8*)(p + plane);
      const bf16x8 bs = *(const bf16x8*)(p + 2 * plane);
      am[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[ks], bh, am[mt], 0, 0, 0);
      av[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[ks], bs, av[mt], 0, 0, 0);
      a1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[ks], bl, a1[mt], 0, 0, 0);
      a2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.lo[ks], bh, a2[mt], 0, 0, 0);
    }
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) am[mt][r] += a1[mt][r] + a2[mt][r];
}

// 4 bias values of a layer starting at channel chb (channels beyond cout read the last one; their results are dropped)
__device__ __forceinline__ f32x4 ml_bias4(const float* b, const LayerDesc& ly, int chb) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = b[ly.bias_off + min(chb + r, ly.cout - 1)];
  return v;
}

// squares of the 4 bf16 values of a hi pair, rounded to bf16 (what the variance contraction sees)
__device__ __forceinline__ uint2 ml_sq4(uint2 h) {
  const float a0 = __uint_as_float(h.x << 16), a1 = __uint_as_float(h.x & 0xffff0000u);
  const float a2 = __uint_as_float(h.y << 16), a3 = __uint_as_float(h.y & 0xffff0000u);
  return make_uint2(cvt_pk(a0 * a0, a1 * a1), cvt_pk(a2 * a2, a3 * a3));
}

// LRT epilogue of one n-tile: noise, ReLU, the next layer's LDS planes (TO_LDS) and the global planes / q
template <bool TO_LDS>
__device__ __forceinline__ void ml_epilogue(const GroupArgs& A, const LayerDesc& ly, int layer, int nt, const f32x4 (&am)[MLF_MT],
                                            const f32x4 (&av)[MLF_MT], f32x4 ba, f32x4 bb, int out_row0, int nvalid,
                                            const TensorRef& tout, const TensorRef& tq, bool relu, char* out_hi, int opitch,
                                            int oplane, int lane) {
  const int i16 = lane & 15, g4 = lane >> 4;
  const int chb = nt * 16 + 4 * g4;
  const int nv = ly.cout - chb;
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt) {
    const int row = mt * 16 + i16;
    const int R = out_row0 + min(row, nvalid - 1);
    f32x4 eps = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ML_ABL & 1) {
      eps = f32x4{0.5f, -0.5f, 0.25f, 1.f};
    } else if (A.nz.use_philox_lrt) {
      const long Rg = global_row(A.cg, 1, R);
      const uint64_t idx = (uint64_t)Rg * (uint64_t)(ly.cout_p16 >> 2) + (uint64_t)(chb >> 2);
      eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)layer << 8), A.nz.step, A.nz.seed);
    } else {
      const float* e = A.nz.lrt_eps[layer] + (long)R * ly.cout + chb;
#pragma unroll
      for (int r = 0; r < 4; ++r) eps[r] = (r < nv) ? e[r] : 0.f;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < nv) {
        const float loc = am[mt][r] + ba[r];
        float var = av[mt][r] + bb[r];
        if (var < 0.f) var = 1e-6f;
        const float sd = sqrtf(var);
        v[r] = loc + sd * eps[r];
        qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
        if (relu) v[r] = fmaxf(v[r], 0.f);
      }
    if constexpr (TO_LDS) {
      uint2 hv, lv;
      split4(v, hv, lv);
      char* o = out_hi + row * opitch + chb * 2;
      *(uint2*)o = hv;
      *(uint2*)(o + oplane) = lv;
      *(uint2*)(o + 2 * oplane) = ml_sq4(hv);
    }
    if (row < nvalid && nv > 0 && !(ML_ABL & 4)) {
      const bool vec = (tout.ctot & 3) == 0;
      if constexpr (TO_LDS) {
        // the next layer reads the LDS planes and the backward only the hi plane: no lo plane in HBM
        TensorRef thi = tout;
        thi.lo = nullptr;
        tstore4(thi, (long)R * tout.ctot + chb, v, nv, vec);
      } else {
        tstore4(tout, (long)R * tout.ctot + chb, v, nv, vec);
      }
      tstore4(tq, (long)R * tq.ctot + chb, qv, nv, vec);
    }
  }
}

// The same for a hidden layer (cout a multiple of 16, bf16 planes): no per-channel tails, packed stores
template <bool TO_LDS>
__device__ __forceinline__ void ml_epilogue_h(const GroupArgs& A, const LayerDesc& ly, int layer, int nt, const f32x4 (&am)[MLF_MT],
                                              const f32x4 (&av)[MLF_MT], f32x4 ba, f32x4 bb, int out_row0, int nvalid,
                                              const TensorRef& tout, const TensorRef& tq, char* out_hi, int opitch, int oplane,
                                              int lane) {
  const int i16 = lane & 15, g4 = lane >> 4;
  const int chb = nt * 16 + 4 * g4;
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt) {
    const int row = mt * 16 + i16;
    const int R = out_row0 + min(row, nvalid - 1);
    f32x4 eps;
    if constexpr (ML_ABL & 1) {
      eps = f32x4{0.5f, -0.5f, 0.25f, 1.f};
    } else if (A.nz.use_philox_lrt) {
      const long Rg = global_row(A.cg, 1, R);
      const uint64_t idx = (uint64_t)Rg * (uint64_t)(ly.cout_p16 >> 2) + (uint64_t)(chb >> 2);
      eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)layer << 8), A.nz.step, A.nz.seed);
    } else {
      const float* e = A.nz.lrt_eps[layer] + (long)R * ly.cout + chb;
#pragma unroll
      for (int r = 0; r < 4; ++r) eps[r] = e[r];
    }
    f32x4 v, qv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float loc = am[mt][r] + ba[r];
      float var = av[mt][r] + bb[r];
      if (var < 0.f) var = 1e-6f;
      const float sd = sqrtf(var);
      v[r] = fmaxf(loc + sd * eps[r], 0.f);
      qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
    }
    uint2 hv, lv;
    split4(v, hv, lv);
    if constexpr (TO_LDS) {
      char* o = out_hi + row * opitch + chb * 2;
      *(uint2*)o = hv;
      *(uint2*)(o + oplane) = lv;
      *(uint2*)(o + 2 * oplane) = ml_sq4(hv);
    }
    if (row < nvalid && !(ML_ABL & 4)) {
      const long o = (long)R * tout.ctot + chb;
      *(uint2*)((u16*)tout.p + o) = hv;
      // the next layer reads the LDS planes and the backward only the hi plane: no lo plane in HBM when TO_LDS
      if constexpr (!TO_LDS) *(uint2*)((u16*)tout.lo + o) = lv;
      *(uint2*)((u16*)tq.p + (long)R * tq.ctot + chb) = make_uint2(cvt_pk(qv[0], qv[1]), cvt_pk(qv[2], qv[3]));
    }
  }
}

// window index -> (particle, first row inside the particle, first row in the [S*B] row space, valid rows)
struct MlWin {
  int s, r0, R0, nvalid;
};
template <int ROWS>
__device__ __forceinline__ MlWin ml_win(const CallGeom& cg, int win) {
  const int pp = (cg.B + ROWS - 1) / ROWS;
  MlWin w;
  w.s = win / pp;
  w.r0 = (win - w.s * pp) * ROWS;
  w.R0 = w.s * cg.B + w.r0;
  w.nvalid = min(ROWS, cg.B - w.r0);
  return w;
}

enum { ML0_LDS = 3 * MLF_ROWS * (ML_K0 * 2 + 32) };

// ==========================================================================================
// layer 0: workgroup = (window, 64-cout group); 4 waves = the group's 4 n-tiles, 17 k-steps each
// ==========================================================================================
__global__ __launch_bounds__(256) void mlp_l0_kernel(const GroupArgs A, const MlpPlan M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = blockIdx.x & 3;
  const MlWin W = ml_win<MLF_ROWS>(A.cg, blockIdx.x >> 2);
  const LayerDesc& ly = M.ly[0];
  constexpr int KS = ML_K0 / 32, PITCH = ml_pitch(ML_K0), PLANE = mlf_plane(ML_K0);
  constexpr int NU = ML_K0 / 4;   // 4-channel units per row (the last one is channel padding)
  // ---- x window: fp32 rows (broadcast over particles) -> hi / lo / squared planes ----
  const float* x = (const float*)A.t[T_X].p;
  const int xc = A.t[T_X].ctot;   // 540
  constexpr int NIT = (MLF_ROWS * NU + 255) / 256;
  f32x4 xv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int u = min(it * 256 + tid, MLF_ROWS * NU - 1);
    const int row = u / NU, c4 = u - row * NU;
    const int srow = min(row, W.nvalid - 1), sc = min(c4 * 4, xc - 4);
    xv[it] = *(const f32x4*)(x + (long)(W.r0 + srow) * xc + sc);
  }
  const int nt = ng * 4 + wave;
  MlFrag<KS> f;
  ml_load<KS>(f, A.ws, ly, nt, lane);
  const int chb = nt * 16 + 4 * (lane >> 4);
  const f32x4 ba = ml_bias4(A.ws.bias_a, ly, chb), bb = ml_bias4(A.ws.bias_b, ly, chb);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int u = min(it * 256 + tid, MLF_ROWS * NU - 1);   // (the clamped tail rewrites the last unit with the same values)
    const int row = u / NU, c4 = u - row * NU;
    f32x4 v = xv[it];
    if (row >= W.nvalid || c4 * 4 >= xc) v = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 hv, lv;
    split4(v, hv, lv);
    char* o = smem + row * PITCH + c4 * 8;
    *(uint2*)o = hv;
    *(uint2*)(o + PLANE) = lv;
    *(uint2*)(o + 2 * PLANE) = ml_sq4(hv);
    if (ng == 0 && row < W.nvalid && W.s == 0) *(uint2*)(M.xhi + (long)(W.r0 + row) * ML_K0 + c4 * 4) = hv;
  }
  lds_barrier();
  f32x4 am[MLF_MT], av[MLF_MT];
  ml_mma<KS>(f, smem, PITCH, PLANE, lane, am, av);
  ml_epilogue_h<false>(A, ly, 0, nt, am, av, ba, bb, W.R0, W.nvalid, M.h[0], M.q[0], nullptr, 0, 0, lane);
}

enum { ML14_O_H1 = 0, ML14_O_H2 = ML14_O_H1 + 3 * MLF_ROWS * (ML_N0 * 2 + 32), ML14_O_H3 = ML14_O_H2 + 3 * MLF_ROWS * (ML_N1 * 2 + 32),
       ML14_O_H4 = ML14_O_H3 + 3 * MLF_ROWS * (ML_N2 * 2 + 32), ML14_LDS = ML14_O_H4 + 3 * MLF_ROWS * (ML_N3 * 2 + 32) };

// ==========================================================================================
// layers 1..4: workgroup = window; 8 waves; layer 1, 2: one n-tile per wave; layer 3: waves 0, 1; layer 4: wave 0
// ==========================================================================================
__global__ __launch_bounds__(512) void mlp_l14_kernel(const GroupArgs A, const MlpPlan M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const MlWin W = ml_win<MLF_ROWS>(A.cg, blockIdx.x);
  // ---- h1 window: bf16 hi / lo planes -> LDS (+ squares): 32 rows x 32 pieces of 16 B per plane ----
  constexpr int NJ = MLF_ROWS * 32 / 512;
  static_assert(NJ * 512 == MLF_ROWS * 32, "h1 staging");
  tr_u32x4 ph[NJ], pl[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int u = j * 512 + tid;
    const int row = u >> 5, c = u & 31;
    const long o = (long)(W.R0 + min(row, W.nvalid - 1)) * M.h[0].ctot + c * 8;
    ph[j] = *(const tr_u32x4*)((const u16*)M.h[0].p + o);
    pl[j] = *(const tr_u32x4*)((const u16*)M.h[0].lo + o);
  }
  // every fragment this wave will need, issued now (the tiles of layers 3 / 4 that are not this wave's are loaded and
  // never used: a load under a branch would be waited for at the join)
  MlFrag<ML_N0 / 32> f1;
  MlFrag<ML_N1 / 32> f2;
  MlFrag<ML_N2 / 32> f3;
  MlFrag<ML_N3 / 32> f4;
  ml_load(f1, A.ws, M.ly[1], wave, lane);
  ml_load(f2, A.ws, M.ly[2], wave, lane);
  const int g4 = lane >> 4;
  const f32x4 ba1 = ml_bias4(A.ws.bias_a, M.ly[1], wave * 16 + 4 * g4), bb1 = ml_bias4(A.ws.bias_b, M.ly[1], wave * 16 + 4 * g4);
  const f32x4 ba2 = ml_bias4(A.ws.bias_a, M.ly[2], wave * 16 + 4 * g4), bb2 = ml_bias4(A.ws.bias_b, M.ly[2], wave * 16 + 4 * g4);
  const f32x4 ba3 = ml_bias4(A.ws.bias_a, M.ly[3], (wave & 1) * 16 + 4 * g4), bb3 = ml_bias4(A.ws.bias_b, M.ly[3], (wave & 1) * 16 + 4 * g4);
  const f32x4 ba4 = ml_bias4(A.ws.bias_a, M.ly[4], 4 * g4), bb4 = ml_bias4(A.ws.bias_b, M.ly[4], 4 * g4);
  constexpr int P1 = ml_pitch(ML_N0), L1 = mlf_plane(ML_N0), P2 = ml_pitch(ML_N1), L2 = mlf_plane(ML_N1);
  constexpr int P3 = ml_pitch(ML_N2), L3 = mlf_plane(ML_N2), P4 = ml_pitch(ML_N3), L4 = mlf_plane(ML_N3);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int u = j * 512 + tid;
    const int row = u >> 5, c = u & 31;
    tr_u32x4 h = ph[j], l = pl[j], q;
    if (row >= W.nvalid) h = l = tr_u32x4{0u, 0u, 0u, 0u};
    const uint2 s0 = ml_sq4(make_uint2(h[0], h[1])), s1 = ml_sq4(make_uint2(h[2], h[3]));
    q = tr_u32x4{s0.x, s0.y, s1.x, s1.y};
    char* o = smem + ML14_O_H1 + row * P1 + c * 16;
    *(tr_u32x4*)o = h;
    *(tr_u32x4*)(o + L1) = l;
    *(tr_u32x4*)(o + 2 * L1) = q;
  }
  lds_barrier();
  // the fragments of layers 3 / 4 are issued here (register pressure: the staging registers are dead now); they land
  // under layers 1 and 2
  ml_load(f3, A.ws, M.ly[3], wave & 1, lane);
  ml_load(f4, A.ws, M.ly[4], 0, lane);
  f32x4 am[MLF_MT], av[MLF_MT];
  ml_mma(f1, smem + ML14_O_H1, P1, L1, lane, am, av);
  ml_epilogue_h<true>(A, M.ly[1], 1, wave, am, av, ba1, bb1, W.R0, W.nvalid, M.h[1], M.q[1], smem + ML14_O_H2, P2, L2, lane);
  lds_barrier();
  ml_mma(f2, smem + ML14_O_H2, P2, L2, lane, am, av);
  ml_epilogue_h<true>(A, M.ly[2], 2, wave, am, av, ba2, bb2, W.R0, W.nvalid, M.h[2], M.q[2], smem + ML14_O_H3, P3, L3, lane);
  lds_barrier();
  if (wave < 2) {
    ml_mma(f3, smem + ML14_O_H3, P3, L3, lane, am, av);
    ml_epilogue_h<true>(A, M.ly[3], 3, wave, am, av, ba3, bb3, W.R0, W.nvalid, M.h[3], M.q[3], smem + ML14_O_H4, P4, L4, lane);
  }
  lds_barrier();
  if (wave == 0) {
    ml_mma(f4, smem + ML14_O_H4, P4, L4, lane, am, av);
    ml_epilogue<false>(A, M.ly[4], 4, 0, am, av, ba4, bb4, W.R0, W.nvalid, M.h[4], M.q[4], false, nullptr, 0, 0, lane);
  }
}

// ==========================================================================================
// dX chain.  Layer l's contraction: dH_{l-1} = dz_l W_l + 2 H_{l-1} o ((dz_l q_l) sigma_l^2), then
// dz_{l-1} = dH_{l-1} [H_{l-1} > 0].  Transposed images: [cin][KPt].
// ==========================================================================================
enum { MX_PZ = MLF_ROWS * (ML_N1 * 2 + 32),   // a dz image of up to 128 channels (pitch 288)
       MX_LDS = 4 * MX_PZ };                 // two buffers x (dz | dz q)

// one 16-channel tile of dH: KS k-steps over the dz images at `zi` (dz) and `zi + MX_PZ` (dz q), pitch `pitch`
template <int KS>
__device__ __forceinline__ void mx_tile(const bf16x8 (&wa)[KS], const bf16x8 (&wb)[KS], const char* zi, int pitch, int lane,
                                        f32x4 (&acc_a)[MLF_MT], f32x4 (&acc_b)[MLF_MT]) {
  const int i16 = lane & 15, g4 = lane >> 4;
  const char* lb = zi + i16 * pitch + g4 * 16;
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt) acc_a[mt] = acc_b[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MLF_MT; ++mt) {
      const char* p = lb + mt * 16 * pitch + ks * 64;
      acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], *(const bf16x8*)p, acc_a[mt], 0, 0, 0);
      acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], *(const bf16x8*)(p + MX_PZ), acc_b[mt], 0, 0, 0);
    }
}

// fragments of cin-tile t of a layer's transposed images
template <int KS>
__device__ __forceinline__ void mx_load(bf16x8 (&wa)[KS], bf16x8 (&wb)[KS], const WeightSlots& ws, const LayerDesc& ly, int t, int lane) {
  const int i16 = lane & 15, g4 = lane >> 4;
  const long row = (long)(t * 16 + i16) * ly.KPt + ly.wt_off + g4 * 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    wa[ks] = *(const bf16x8*)((const u16*)ws.at + row + ks * 32);
    wb[ks] = *(const bf16x8*)((const u16*)ws.bt + row + ks * 32);
  }
}

// this lane's 4 channels of H (hi plane) and q of a dH tile, both m-tiles: [mt] -> (H, q) as bf16 pairs
struct MxEp {
  uint2 h[MLF_MT], q[MLF_MT];
};
__device__ __forceinline__ void mx_ep_load(MxEp& e, const TensorRef& th, const TensorRef& tq, int R0, int nvalid, int t, int lane) {
  const int i16 = lane & 15, g4 = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt) {
    const long o = (long)(R0 + min(mt * 16 + i16, nvalid - 1)) * th.ctot + t * 16 + 4 * g4;
    e.h[mt] = *(const uint2*)((const u16*)th.p + o);
    e.q[mt] = *(const uint2*)((const u16*)tq.p + o);
  }
}

// dH tile -> masked dz and dz q: into the next contraction's LDS images and the global planes for dW
template <bool TO_LDS>
__device__ __forceinline__ void mx_finish(const f32x4 (&acc_a)[MLF_MT], const f32x4 (&acc_b)[MLF_MT], const MxEp& e, int t, int R0, int nvalid,
                                          const TensorRef& tg, const TensorRef& tq, char* zo, int opitch, int lane) {
  const int i16 = lane & 15, g4 = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MLF_MT; ++mt) {
    const int row = mt * 16 + i16;
    const f32x4 xv = unpack_bf4(e.h[mt]), qv = unpack_bf4(e.q[mt]);
    f32x4 g, g2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc_a[mt][r] + 2.f * xv[r] * acc_b[mt][r];
      g[r] = (xv[r] > 0.f && row < nvalid) ? v : 0.f;
      g2[r] = g[r] * qv[r];
    }
    const uint2 dz = make_uint2(cvt_pk(g[0], g[1]), cvt_pk(g[2], g[3]));
    const uint2 dz2 = make_uint2(cvt_pk(g2[0], g2[1]), cvt_pk(g2[2], g2[3]));
    if constexpr (TO_LDS) {
      char* o = zo + row * opitch + (t * 16 + 4 * g4) * 2;
      *(uint2*)o = dz;
      *(uint2*)(o + MX_PZ) = dz2;
    }
    if (row < nvalid) {
      const long o = (long)(R0 + row) * tg.ctot + t * 16 + 4 * g4;
      *(uint2*)((u16*)tg.p + o) = dz;
      *(uint2*)((u16*)tq.p + o) = dz2;
    }
  }
}

__global__ __launch_bounds__(512) void mlp_dx_kernel(const GroupArgs A, const MlpPlan M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const MlWin W = ml_win<MLF_ROWS>(A.cg, blockIdx.x);
  // the gradient images of this call start from zero (no separate fill launch)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    tr_u32x4* z = (tr_u32x4*)M.zero_p[k];
    for (long i = (long)blockIdx.x * 512 + tid; i < M.zero_n[k]; i += (long)gridDim.x * 512) z[i] = tr_u32x4{0u, 0u, 0u, 0u};
  }
  // dz of the last layer (d loss / d z, no ReLU) and its q: fp32 [rows][2].  On the training step the head itself runs
  // here (predictions, log-likelihood sum, dz), otherwise its output is read.
  float z0 = 0.f, z1 = 0.f, q0 = 0.f, q1 = 0.f;
  {
    const long R = W.R0 + min(tid & (MLF_ROWS - 1), W.nvalid - 1);
    q0 = ((const float*)M.q[4].p)[R * 2];
    q1 = ((const float*)M.q[4].p)[R * 2 + 1];
    if (M.fuse_head) {
      if (tid < 64) {
        double ll = 0.0;
        if (tid < W.nvalid) ll = head_row(M.head, W.s, W.r0 + tid, z0, z1);
        ll = wave_sum_d(ll);
        if (tid == 0 && M.head.with_obs) atomicAdd(M.head.ll_acc + W.s, ll);
      }
    } else {
      z0 = ((const float*)M.g[4].p)[R * 2];
      z1 = ((const float*)M.g[4].p)[R * 2 + 1];
    }
  }
  // fragments: layer 4 (cin 32: tile wave & 1), 3 (cin 128: tile wave), 2 (cin 128: tile wave), 1 (cin 256: tiles 2 wave, 2 wave + 1)
  bf16x8 wa4[1], wb4[1], wa3[1], wb3[1], wa2[4], wb2[4], wa1[2][4], wb1[2][4];
  mx_load<1>(wa4, wb4, A.ws, M.ly[4], wave & 1, lane);
  mx_load<1>(wa3, wb3, A.ws, M.ly[3], wave, lane);
  mx_load<4>(wa2, wb2, A.ws, M.ly[2], wave, lane);
  mx_load<4>(wa1[0], wb1[0], A.ws, M.ly[1], 2 * wave, lane);
  mx_load<4>(wa1[1], wb1[1], A.ws, M.ly[1], 2 * wave + 1, lane);
  MxEp e3, e2, e1, e0[2];   // H / q of the tiles this wave finishes: h[3] (32 ch), h[2], h[1] (128), h[0] (256)
  mx_ep_load(e3, M.h[3], M.q[3], W.R0, W.nvalid, wave & 1, lane);
  mx_ep_load(e2, M.h[2], M.q[2], W.R0, W.nvalid, wave, lane);
  mx_ep_load(e1, M.h[1], M.q[1], W.R0, W.nvalid, wave, lane);
  mx_ep_load(e0[0], M.h[0], M.q[0], W.R0, W.nvalid, 2 * wave, lane);
  mx_ep_load(e0[1], M.h[0], M.q[0], W.R0, W.nvalid, 2 * wave + 1, lane);
  char* za = smem;                 // buffer A: dz | dz q
  char* zb = smem + 2 * MX_PZ;     // buffer B
  constexpr int P32 = ml_pitch(32), P128 = ml_pitch(128);
  // ---- stage dz_4 (K = 2 couts, padded to one 32-channel k-step) ----
  if (tid < MLF_ROWS) {
    const bool on = tid < W.nvalid;
    const float a0 = on ? z0 : 0.f, a1 = on ? z1 : 0.f;
    const tr_u32x4 d = {cvt_pk(a0, a1), 0u, 0u, 0u}, d2 = {cvt_pk(a0 * q0, a1 * q1), 0u, 0u, 0u}, zz = {0u, 0u, 0u, 0u};
    char* o = za + tid * P32;
    *(tr_u32x4*)o = d;
    *(tr_u32x4*)(o + MX_PZ) = d2;
#pragma unroll
    for (int c = 1; c < 4; ++c) {
      *(tr_u32x4*)(o + c * 16) = zz;
      *(tr_u32x4*)(o + MX_PZ + c * 16) = zz;
    }
    if (on) {
      *(tr_u32x4*)(M.dz4 + (long)(W.R0 + tid) * 8) = d;
      *(tr_u32x4*)(M.dz4 + M.dz4_plane + (long)(W.R0 + tid) * 8) = d2;
    }
  }
  lds_barrier();
  f32x4 acc_a[MLF_MT], acc_b[MLF_MT];
  // ---- layer 4: dH_3 (32 channels): waves 0, 1 ----
  if (wave < 2) {
    mx_tile<1>(wa4, wb4, za, P32, lane, acc_a, acc_b);
    mx_finish<true>(acc_a, acc_b, e3, wave, W.R0, W.nvalid, M.g[3], M.q[3], zb, P32, lane);
  }
  lds_barrier();
  // ---- layer 3: dH_2 (128 channels), K = 32 ----
  mx_tile<1>(wa3, wb3, zb, P32, lane, acc_a, acc_b);
  mx_finish<true>(acc_a, acc_b, e2, wave, W.R0, W.nvalid, M.g[2], M.q[2], za, P128, lane);
  lds_barrier();
  // ---- layer 2: dH_1 (128 channels), K = 128 ----
  mx_tile<4>(wa2, wb2, za, P128, lane, acc_a, acc_b);
  mx_finish<true>(acc_a, acc_b, e1, wave, W.R0, W.nvalid, M.g[1], M.q[1], zb, P128, lane);
  lds_barrier();
  // ---- layer 1: dH_0 (256 channels), K = 128: two tiles per wave; only the global planes (layer 0 has no dX) ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    mx_tile<4>(wa1[j], wb1[j], zb, P128, lane, acc_a, acc_b);
    mx_finish<false>(acc_a, acc_b, e0[j], 2 * wave + j, W.R0, W.nvalid, M.g[0], M.q[0], nullptr, 0, lane);
  }
}

// ==========================================================================================
// dW of all layers: workgroup = (job = (layer, 64-cout group, 128-cin chunk), window split).
//   waves 0..3 : one n-tile x the chunk's 8 c-tiles each (mu and sigma^2 gradient tiles in registers across the windows)
//   waves 4..7 : image builders, a quarter of the next window each: dz | dz q copied from the planes the dX chain wrote,
//                x^2 from the landed X ring slot
//   wave  8    : LDS-DMA of the X chunk into a ring of 6 windows, four ahead
// (the structure of dense_ks_bwd_kernel without the dX waves; see there for the barrier protocol)
// ==========================================================================================
enum { MW_ND = 4, MW_NB = 4, MW_WAVES = MW_ND + MW_NB + 1, MW_RING = 6, MW_AHEAD = 4, MW_IMG = ML_ROWS * 256,
       MW_O_XR = 0, MW_O_X2 = MW_O_XR + MW_RING * MW_IMG, MW_O_Z = MW_O_X2 + 2 * MW_IMG, MW_LDS = MW_O_Z + 2 * MW_IMG };

struct MlpDwJob {
  const u16* x;            // input plane of the layer (hi), already offset to the chunk's first channel
  const u16* dz;           // masked gradient plane, offset to the group's first cout
  const u16* dz2;          // dz q plane
  float* gwa; float* gwb;  // mu / sigma^2 gradient images, offset to (first cout of the group, first cin of the chunk)
  float* gba; float* gbb;  // bias gradients of the group (null unless this job owns them: chunk 0)
  int x_ctot, z_ctot;      // row strides (elements)
  int cw;                  // valid channels of the chunk (multiple of 16)
  int cout;                // valid couts of the group (<= 64)
  int KP;                  // row stride of the gradient images
  int x_bcast;             // 1: the input plane has no particle dimension (layer 0 reads x)
};
struct MlpDwPlan {
  int njobs, nsplit;
  MlpDwJob job[30];
};

// squares of 8 bf16 values, rounded to bf16
__device__ __forceinline__ tr_u32x4 ml_sq8(tr_u32x4 v) {
  const uint2 a = ml_sq4(make_uint2(v[0], v[1])), b = ml_sq4(make_uint2(v[2], v[3]));
  return tr_u32x4{a.x, a.y, b.x, b.y};
}

__global__ __launch_bounds__(MW_WAVES * 64) void mlp_dw_kernel(const CallGeom cg, const MlpDwPlan D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ji = blockIdx.x / D.nsplit, split = blockIdx.x - ji * D.nsplit;
  const MlpDwJob& J = D.job[ji];
  const int B = cg.B, pp = (cg.B + ML_ROWS - 1) / ML_ROWS;
  const int nwin = cg.S * pp;   // windows of the whole call: the gradient is summed over particles
  const int nwl = split < nwin ? (nwin - split + D.nsplit - 1) / D.nsplit : 0;
  const int cw = J.cw;

  if (wave == MW_WAVES - 1) {
    // =========================== X loader (LDS-DMA) ===========================
    const int cw8 = cw >> 3;
    const int r4 = lane >> 4, pc = lane & 15;
    const uint32_t rowb = (uint32_t)J.x_ctot * 2u;
    uint32_t cb[4], act = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // instruction i covers rows 4i .. 4i+3: f128(row) = (r4 << 2) | (i & 3)
      const int c = pc ^ ((r4 << 2) | i);
      cb[i] = (uint32_t)c * 16u;
      if (c < cw8) act |= 1u << i;
    }
    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int k, int slot) {
      const MlWin W = ml_win<ML_ROWS>(cg, split + k * D.nsplit);
      const int row0 = J.x_bcast ? W.r0 : W.R0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = min(4 * i + r4, W.nvalid - 1);   // pad rows re-read the last valid one (their dz rows are zero)
        const uint32_t off = (uint32_t)(row0 + row) * rowb + cb[i & 3];
        if ((act >> (i & 3)) & 1u) dma16((const char*)J.x + off, lds0 + (uint32_t)(MW_O_XR + slot * MW_IMG + i * 1024));
      }
    };
    constexpr int n_issue = 8;
    for (int j = 0; j < MW_AHEAD; ++j)
      if (j < nwl) issue(j, j);
    BNN_WAIT_VMCNT_WIDE(max(0, min(nwl, MW_AHEAD) - 1) * n_issue);   // window 0 landed
    lds_barrier();   // P
    lds_barrier();   // Q
    int slot_i = MW_AHEAD % MW_RING;
    for (int k = 0; k < nwl; ++k) {
      const int fly = max(0, min(k + MW_AHEAD - 1, nwl - 1) - (k + 1)) * n_issue;
      BNN_WAIT_VMCNT_WIDE(fly);
      lds_barrier();   // B(k)
      if (k + MW_AHEAD < nwl) issue(k + MW_AHEAD, slot_i);
      slot_i = slot_i + 1 == MW_RING ? 0 : slot_i + 1;
    }
    return;
  }
  if (wave >= MW_ND) {
    // =========================== image builders ===========================
    asm volatile("" ::: "memory");
    const int bw = wave - MW_ND;
    const int c8 = lane & 7, zrow = bw * 8 + (lane >> 3);
    const bool c_on = c8 * 8 < J.cout;
    const int cc8 = c_on ? c8 : 0;
    const int zo = zrow * 256 + ((c8 ^ f128(zrow)) << 4);
    auto load = [&](int k, tr_u32x4& dz, tr_u32x4& dz2, int& nvw) {
      const MlWin W = ml_win<ML_ROWS>(cg, split + k * D.nsplit);
      nvw = W.nvalid;
      const long o = (long)(W.R0 + min(zrow, nvw - 1)) * J.z_ctot + cc8 * 8;
      dz = *(const tr_u32x4*)(J.dz + o);
      dz2 = *(const tr_u32x4*)(J.dz2 + o);
    };
    auto make = [&](int k, int ring, tr_u32x4 dz, tr_u32x4 dz2, int nvw) {
      char* zi = smem + MW_O_Z + (k & 1) * MW_IMG;
      if (zrow >= nvw || !c_on) dz = dz2 = tr_u32x4{0u, 0u, 0u, 0u};
      *(tr_u32x4*)(zi + zo) = dz;
      *(tr_u32x4*)(zi + (zo ^ 128)) = dz2;
      const char* xr = smem + MW_O_XR + ring * MW_IMG;
      char* x2 = smem + MW_O_X2 + (k & 1) * MW_IMG;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = (bw * 2 + j) * 64 + lane;
        *(tr_u32x4*)(x2 + e * 16) = ml_sq8(*(const tr_u32x4*)(xr + e * 16));
      }
    };
    tr_u32x4 z0 = {0u, 0u, 0u, 0u}, y0 = z0, z1 = z0, y1 = z0, z2 = z0, y2 = z0;
    int nv0 = 0, nv1 = 0, nv2 = 0;
    if (0 < nwl) load(0, z0, y0, nv0);
    if (1 < nwl) load(1, z1, y1, nv1);
    if (2 < nwl) load(2, z2, y2, nv2);
    lds_barrier();   // P: window 0's X landed
    if (nwl > 0) {
      make(0, 0, z0, y0, nv0);
      if (3 < nwl) load(3, z0, y0, nv0);
    }
    lds_barrier();   // Q
    int ring = 1;
#define MW_STEP(K, DZ, DZ2, NV)                        \
  do {                                                 \
    lds_barrier();                                     \
    if ((K) + 1 < nwl) {                               \
      make((K) + 1, ring, DZ, DZ2, NV);                \
      if ((K) + 4 < nwl) load((K) + 4, DZ, DZ2, NV);   \
    }                                                  \
    ring = ring + 1 == MW_RING ? 0 : ring + 1;         \
  } while (0)
    for (int k = 0; k < nwl; k += 3) {
      MW_STEP(k, z1, y1, nv1);
      if (k + 1 < nwl) MW_STEP(k + 1, z2, y2, nv2);
      if (k + 2 < nwl) MW_STEP(k + 2, z0, y0, nv0);
    }
#undef MW_STEP
    asm volatile("" ::"v"(z0), "v"(y0), "v"(z1), "v"(y1), "v"(z2), "v"(y2));   // close the vmcnt state (see dense_ks_bwd)
    return;
  }

  // =========================== dW waves ===========================
  asm volatile("" ::: "memory");
  const int nt = wave;
  f32x4 acc_a[8], acc_b[8], bias_a = {0.f, 0.f, 0.f, 0.f}, bias_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c) acc_a[c] = acc_b[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  const bool bias_job = J.gba != nullptr;
  const bool tile_on = nt * 16 < J.cout;
  const int gq = lane >> 4, qq = (lane >> 2) & 3, p4 = lane & 3;
  const int r0 = 8 * gq + qq;
  const int f0 = f128(r0), f1 = f128(r0 + 4);
  const int ho = 8 * (p4 & 1);
  const int ca = nt * 2 + (p4 >> 1);
  const int a0 = r0 * 256 + ((ca ^ f0) << 4) + ho, a1 = (r0 + 4) * 256 + ((ca ^ f1) << 4) + ho;
  lds_barrier();   // P
  lds_barrier();   // Q
  int ring = 0;
  for (int k = 0; k < nwl; ++k) {
    const char* zi = smem + MW_O_Z + (k & 1) * MW_IMG;
    const char* xi = smem + MW_O_XR + ring * MW_IMG;
    const char* x2 = smem + MW_O_X2 + (k & 1) * MW_IMG;
    lds_barrier();
    if (tile_on) {
      const bf16x8 fa = tr_frag2(zi + a0, zi + a1);
      const bf16x8 fa2 = tr_frag2(zi + (a0 ^ 128), zi + (a1 ^ 128));
      if (bias_job) {
        bias_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, bias_a, 0, 0, 0);
        bias_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa2, ones, bias_b, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c * 16 >= cw) break;
        const int cb = c * 2 + (p4 >> 1);
        const int b0 = r0 * 256 + ((cb ^ f0) << 4) + ho, b1 = (r0 + 4) * 256 + ((cb ^ f1) << 4) + ho;
        acc_a[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, tr_frag2(xi + b0, xi + b1), acc_a[c], 0, 0, 0);
        acc_b[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa2, tr_frag2(x2 + b0, x2 + b1), acc_b[c], 0, 0, 0);
      }
    }
    ring = ring + 1 == MW_RING ? 0 : ring + 1;
  }
  if (!tile_on) return;
  const int i4 = 4 * (lane >> 4), jc = lane & 15;
  const bool direct = D.nsplit == 1;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c * 16 >= cw) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= J.cout) continue;
      const long o = (long)n * J.KP + c * 16 + jc;
      if (direct) {
        J.gwa[o] = acc_a[c][r];
        J.gwb[o] = acc_b[c][r];
      } else {
        atomicAdd(J.gwa + o, acc_a[c][r]);
        atomicAdd(J.gwb + o, acc_b[c][r]);
      }
    }
  }
  if (bias_job && jc == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= J.cout) continue;
      if (direct) {
        J.gba[n] = bias_a[r];
        J.gbb[n] = bias_b[r];
      } else {
        atomicAdd(J.gba + n, bias_a[r]);
        atomicAdd(J.gbb + n, bias_b[r]);
      }
    }
  }
}
