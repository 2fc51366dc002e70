// Round-1 optimised bf16 kernels for the CONV groups (inception blocks): one 8-wave
// workgroup owns a contiguous set of windows of ONE particle and processes ALL branches of
// the group per window, so every window is staged into LDS once (register-prefetched global
// loads: the loads of window i+1 are in flight while window i is computed).
//
//   conv_dw_bf_kernel : dW tiles of all branches live in registers across the windows of the
//                       workgroup; operands come through ds_read_b64_tr_b16; Flipout's sign
//                       products are applied to the fragments (the sign of a conv channel is
//                       constant over the rows of a window), LRT reads its own dZ*q / X^2 images.
#pragma once
#include "kernels_group.h"

enum { CV_WAVES = 8, CV_THREADS = 512 };

struct DwTile {
  signed char b, nt, tap, ct;
};

struct ConvDwPlan {
  int ntiles;
  int zw;                 // channels of the concatenated dZ image (branches padded to 16)
  int zoff[BNN_MAX_BRANCH];
  int nsplit;             // workgroups per particle
  int has_pool;
  DwTile tile[96];
};

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

// MAXT: dW tiles per wave; XU / ZU: float4 staging units per thread for X / dZ
template <int EM, int MAXT, int XU, int ZU>
__global__ __launch_bounds__(CV_THREADS) void conv_dw_bf_kernel(const GroupArgs A, const ConvDwPlan D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> tile data in SGPRs
  const GroupDesc& G = A.g;
  const int s = blockIdx.x / D.nsplit, split = blockIdx.x - s * D.nsplit;
  const int L = G.L, B = A.cg.B;
  const int cwp = G.in_cin_p;               // all channels of the input tensor view
  const int xw16 = (cwp + 15) & ~15;
  const int RSx = img_row_stride(xw16, true), RSz = img_row_stride(D.zw, true);
  const int xbytes = (IMG_ROWS * RSx * 2 + 15) & ~15, zbytes = (IMG_ROWS * RSz * 2 + 15) & ~15;
  u16* x_hi = (u16*)smem;
  u16* xp_hi = (u16*)(smem + xbytes);
  u16* x_sq = (u16*)(smem + 2 * xbytes);
  u16* xp_sq = (u16*)(smem + 3 * xbytes);
  u16* dz = (u16*)(smem + 4 * xbytes);
  u16* dz2 = (u16*)(smem + 4 * xbytes + zbytes);
  // zero everything once: halo rows / pad channels are never written again
  {
    const int total = (4 * xbytes + 2 * zbytes) >> 2;
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < total; k += CV_THREADS) z[k] = 0u;
  }
  const TensorRef tin = A.t[G.in_t];
  const int xc4 = cwp >> 2, zc4 = D.zw >> 2;
  const int xunits = L * xc4, zunits = L * zc4;

  // ---- per-thread staging plan (fixed for all windows) ----
  int x_src[XU], x_dst[XU];
  bool x_ok[XU];
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int unit = tid + u * CV_THREADS;
    x_ok[u] = unit < xunits;
    const int row = unit / xc4, c = (unit - row * xc4) * 4;
    x_src[u] = row * tin.ctot + c;
    x_dst[u] = (row + HALO) * RSx + c;
    if (c + 4 > tin.ctot && x_ok[u]) x_ok[u] = (c < tin.ctot);  // partial handled in loader
  }
  int z_src[ZU], z_dst[ZU], z_nv[ZU];
  int z_tg[ZU], z_ty[ZU], z_tq[ZU];   // tensor ids of dY / Y (relu mask; -1 none) / q (LRT)
  int z_ct[ZU];
#pragma unroll
  for (int u = 0; u < ZU; ++u) {
    const int unit = tid + u * CV_THREADS;
    z_nv[u] = 0;
    z_src[u] = z_dst[u] = z_ct[u] = 0;
    z_tg[u] = z_ty[u] = z_tq[u] = -1;
    if (unit < zunits) {
      const int row = unit / zc4, zc = (unit - row * zc4) * 4;
      int b = 0;
      for (int k = 1; k < G.n_branch; ++k)
        if (zc >= D.zoff[k]) b = k;
      const BranchDesc& br = G.br[b];
      const int c = zc - D.zoff[b];
      const int nv = min(4, br.cout - c);
      if (nv > 0) {
        const TensorRef tg = A.t[br.out_t + T_GRAD];
        z_nv[u] = nv;
        z_ct[u] = tg.ctot;
        z_src[u] = row * tg.ctot + br.out_off + c;
        z_dst[u] = (row + HALO) * RSz + zc;
        z_tg[u] = br.out_t + T_GRAD;
        z_ty[u] = br.relu ? br.out_t : -1;
        z_tq[u] = LRT ? br.q_t : -1;
      }
    }
  }
  // ---- per-wave tiles: everything that does not depend on the window is hoisted ----
  f32x4 acc_a[MAXT], acc_b[MAXT];
  int t_a[MAXT], t_b[MAXT];        // LDS element offsets of the first tr-read of A / B
  // flipout (all wave-uniform): pointer to the sign word of example 0 that holds this tile's 16
  // lanes, words per example, bit of lane 0
  const uint32_t* t_sop[MAXT];
  const uint32_t* t_sip[MAXT];
  int t_sow[MAXT], t_siw[MAXT], t_so[MAXT], t_si[MAXT];
  bool t_ok[MAXT];
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    acc_a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t = wave + CV_WAVES * m;
    t_ok[m] = t < D.ntiles;
    t_a[m] = t_b[m] = t_so[m] = t_si[m] = t_sow[m] = t_siw[m] = 0;
    t_sop[m] = t_sip[m] = nullptr;
    if (t_ok[m]) {
      const DwTile T = D.tile[t];
      const BranchDesc& br = G.br[T.b];
      const LayerDesc& ly = A.layers[br.layer];
      const int n0 = D.zoff[T.b] + T.nt * 16;
      const int c0 = br.in_off + T.ct * 16;
      const int tshift = T.tap - ly.pad;
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const int r0 = 8 * g + q;
      t_a[m] = (r0 + HALO) * RSz + n0 + 4 * p;
      t_b[m] = (br.pool ? (xbytes >> 1) : 0) + (r0 + tshift + HALO) * RSx + c0 + 4 * p;  // xp_hi follows x_hi
      const int nbit = br.n_off + T.nt * 16;  // lane l adds (l & 15): same 32-bit word for all lanes
      const int cbit = T.ct * 16;
      t_so[m] = nbit & 31;
      t_si[m] = cbit & 31;
      t_sop[m] = A.nz.sign_out + ly.sign_out_off * A.nz.examples + (nbit >> 5);
      t_sip[m] = A.nz.sign_in + ly.sign_in_off * A.nz.examples + (cbit >> 5);
      t_sow[m] = ly.sign_out_words;
      t_siw[m] = ly.sign_in_words;
    }
  }
  float gb_a = 0.f, gb_b = 0.f;
  uint32_t sg_o[MAXT], sg_i[MAXT];  // prefetched sign words of the next window

  f32x4 px[XU], pz[ZU], pq[ZU];
  const bool x_vec = ((tin.ctot & 3) == 0);
  auto prefetch = [&](int wl) {
    const int w = s * B + wl;
    const long xrow0 = (long)(G.in_bcast ? wl : w) * L;
    const long zrow0 = (long)w * L;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      px[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (x_ok[u]) {
        const int c = x_src[u] % tin.ctot;
        px[u] = tload4(tin, xrow0 * tin.ctot + x_src[u], tin.ctot - c, x_vec);
      }
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      pz[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      pq[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (z_nv[u] > 0) {
        const long o = zrow0 * z_ct[u] + z_src[u];
        const bool v = ((z_ct[u] & 3) == 0) && ((z_src[u] & 3) == 0);
        f32x4 g = tload4(A.t[z_tg[u]], o, z_nv[u], v);
        if (z_ty[u] >= 0) {
          const f32x4 y = tload4(A.t[z_ty[u]], o, z_nv[u], v);
#pragma unroll
          for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.f ? g[k] : 0.f;
        }
        pz[u] = g;
        if constexpr (LRT) pq[u] = tload4(A.t[z_tq[u]], o, z_nv[u], v);
      }
    }
    if constexpr (EM == EM_FLIPOUT) {
#pragma unroll
      for (int m = 0; m < MAXT; ++m) {
        sg_o[m] = sg_i[m] = 0u;
        if (t_ok[m]) {
          sg_o[m] = t_sop[m][(long)w * t_sow[m]];
          sg_i[m] = t_sip[m][(long)w * t_siw[m]];
        }
      }
    }
  };

  const int pp = B;  // windows per particle (conv groups)
  int wl = split;
  if (wl < pp) prefetch(wl);
  for (; wl < pp; wl += D.nsplit) {
    __syncthreads();  // previous window fully consumed (and the initial zero fill is visible)
#pragma unroll
    for (int u = 0; u < XU; ++u)
      if (tid + u * CV_THREADS < xunits) {
        *(uint2*)&x_hi[x_dst[u]] = pack_bf4(px[u]);
        if constexpr (LRT) {
          f32x4 q;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xb = bf2f(f2bf(px[u][k]));
            q[k] = xb * xb;
          }
          *(uint2*)&x_sq[x_dst[u]] = pack_bf4(q);
        }
      }
#pragma unroll
    for (int u = 0; u < ZU; ++u)
      if (tid + u * CV_THREADS < zunits) {
        *(uint2*)&dz[z_dst[u]] = pack_bf4(pz[u]);
        if constexpr (LRT) {
          f32x4 q;
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = pz[u][k] * pq[u][k];
          *(uint2*)&dz2[z_dst[u]] = pack_bf4(q);
        }
      }
    __syncthreads();
    if (D.has_pool) {
      // pooled image from the bf16 image: rounding is monotone, so max commutes with it
      for (int unit = tid; unit < xunits; unit += CV_THREADS) {
        const int row = unit / xc4, c = (unit - row * xc4) * 4;
        const int o = (row + HALO) * RSx + c;
        uint2 v0 = *(const uint2*)&x_hi[o];
        u16 h[4] = {(u16)(v0.x & 0xffff), (u16)(v0.x >> 16), (u16)(v0.y & 0xffff), (u16)(v0.y >> 16)};
        float m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = bf2f(h[k]);
        if (row > 0) {
          const uint2 a = *(const uint2*)&x_hi[o - RSx];
          m[0] = fmaxf(m[0], bf2f((u16)(a.x & 0xffff))); m[1] = fmaxf(m[1], bf2f((u16)(a.x >> 16)));
          m[2] = fmaxf(m[2], bf2f((u16)(a.y & 0xffff))); m[3] = fmaxf(m[3], bf2f((u16)(a.y >> 16)));
        }
        if (row + 1 < L) {
          const uint2 a = *(const uint2*)&x_hi[o + RSx];
          m[0] = fmaxf(m[0], bf2f((u16)(a.x & 0xffff))); m[1] = fmaxf(m[1], bf2f((u16)(a.x >> 16)));
          m[2] = fmaxf(m[2], bf2f((u16)(a.y & 0xffff))); m[3] = fmaxf(m[3], bf2f((u16)(a.y >> 16)));
        }
        *(uint2*)&xp_hi[o] = pack_bf4(f32x4{m[0], m[1], m[2], m[3]});
        if constexpr (LRT) *(uint2*)&xp_sq[o] = pack_bf4(f32x4{m[0] * m[0], m[1] * m[1], m[2] * m[2], m[3] * m[3]});
      }
      __syncthreads();
    }
    // sign words of THIS window were prefetched with its activations; keep a copy because the
    // prefetch of the next window (issued below) overwrites the registers
    bool neg_o[MAXT], neg_i[MAXT];
    if constexpr (EM == EM_FLIPOUT) {
#pragma unroll
      for (int m = 0; m < MAXT; ++m) {
        neg_o[m] = (sg_o[m] >> (t_so[m] + (lane & 15))) & 1u;
        neg_i[m] = (sg_i[m] >> (t_si[m] + (lane & 15))) & 1u;
      }
    }
    if (wl + D.nsplit < pp) prefetch(wl + D.nsplit);  // in flight during the MFMAs below
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      if (t_ok[m]) {
        const u16* a0 = dz + t_a[m];
        const u16* b0 = x_hi + t_b[m];
        const bf16x8 fa = tr_frag(a0, a0 + 4 * RSz);
        const bf16x8 fb = tr_frag(b0, b0 + 4 * RSx);
        acc_a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc_a[m], 0, 0, 0);
        if constexpr (LRT) {
          const u16* a2 = dz2 + t_a[m];
          const u16* b2 = x_sq + t_b[m];  // xp_sq follows x_sq at the same distance
          acc_b[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a2, a2 + 4 * RSz), tr_frag(b2, b2 + 4 * RSx),
                                                             acc_b[m], 0, 0, 0);
        } else if constexpr (EM == EM_FLIPOUT) {
          acc_b[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xor_sign(fa, neg_o[m]), xor_sign(fb, neg_i[m]), acc_b[m],
                                                             0, 0, 0);
        }
      }
    }
    if (tid < D.zw) {
      float sa = 0.f, sb = 0.f;
      for (int r = 0; r < L; ++r) {
        sa += bf2f(dz[(r + HALO) * RSz + tid]);
        if constexpr (LRT) sb += bf2f(dz2[(r + HALO) * RSz + tid]);
      }
      gb_a += sa;
      gb_b += sb;
    }
  }
  // ---- write out ----
  const int i4 = 4 * (lane >> 4), jc = lane & 15;
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    const int t = wave + CV_WAVES * m;
    if (t >= D.ntiles) continue;
    const DwTile T = D.tile[t];
    const BranchDesc& br = G.br[T.b];
    const LayerDesc& ly = A.layers[br.layer];
    const int c = T.ct * 16 + jc;
    if (c >= br.cin_p) continue;
    float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
    float* gwb = A.gw_b + A.gw_stride * s + ly.w_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = T.nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      const long o = (long)(br.n_off + n) * ly.KP + (long)T.tap * ly.cin_img + c;
      atomicAdd(gwa + o, acc_a[m][r]);
      if constexpr (DUAL) atomicAdd(gwb + o, acc_b[m][r]);
    }
  }
  if (tid < D.zw) {
    int b = 0;
    for (int k = 1; k < G.n_branch; ++k)
      if (tid >= D.zoff[k]) b = k;
    const BranchDesc& br = G.br[b];
    const int n = tid - D.zoff[b];
    if (n < br.cout) {
      const LayerDesc& ly = A.layers[br.layer];
      atomicAdd(A.gb_a + (long)A.gb_stride * s + ly.bias_off + br.n_off + n, gb_a);
      if constexpr (LRT) atomicAdd(A.gb_b + (long)A.gb_stride * s + ly.bias_off + br.n_off + n, gb_b);
    }
  }
}
