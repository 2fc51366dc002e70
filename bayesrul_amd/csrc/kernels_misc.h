// Parameter-side and elementwise kernels: noise generation, weight sampling + KL, head / NLL,
// gradient chain rule to (mu, rho), ClippedAdam, max-pool backward, predictive aggregation.
#pragma once
#include "kernels_core.h"

struct SiteDesc {
  long off;      // offset in the flat (mu, rho) buffers
  long numel;
  int layer;
  int is_bias;
};

struct ParamTable {
  int n_sites;
  int n_layers;
  long P;
  SiteDesc site[BNN_MAX_SITES];
};

__device__ __forceinline__ int find_site(const ParamTable& T, long e) {
  int s = 0;
#pragma unroll 1
  for (int k = 1; k < T.n_sites; ++k)
    if (e >= T.site[k].off) s = k;
  return s;
}

__device__ __forceinline__ int map_cin(const LayerDesc& L, int ci) {
  if (L.cmap == CM_BLOCK) return (ci / L.cmap_a) * L.cmap_b + (ci % L.cmap_a);
  if (L.cmap == CM_FLATTEN) return (ci % L.cmap_b) * L.cmap_a + (ci / L.cmap_b);  // c*L + l -> l*C + c
  return ci;
}

// canonical weight element -> (n, forward-image index, transposed-image index)
__device__ __forceinline__ void map_weight(const LayerDesc& L, long e, int& n, long& fi, long& ti) {
  const int per_n = L.cin * L.taps;
  n = (int)(e / per_n);
  const int rem = (int)(e - (long)n * per_n);
  const int ci = rem / L.taps;
  const int t = rem - ci * L.taps;
  const int cimg = map_cin(L, ci);
  fi = (long)n * L.KP + (long)t * L.cin_img + cimg;
  ti = (long)cimg * L.KPt + (long)(L.taps - 1 - t) * L.cout_p8 + n;
}

// ------------------------------------------------------------------------------------------
// noise generation (Philox) — the same streams bnn_export_noise writes out
// ------------------------------------------------------------------------------------------
__global__ void gen_eps_w_kernel(float* eps, long P, int S, uint64_t seed, uint32_t step) {
  const long n4 = (P + 3) >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n4 * S) return;
  const int s = (int)(idx / n4);
  const long q = idx - (long)s * n4;
  const f32x4 z = philox_normal4((uint32_t)q, (uint32_t)s, NK_EPSW, step, seed);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (q * 4 + r < P) eps[(long)s * P + q * 4 + r] = z[r];
}

__global__ void gen_radial_r_kernel(float* rr, int n_sites, int S, uint64_t seed, uint32_t step) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_sites * S) return;
  const int s = idx / n_sites, site = idx - s * n_sites;
  const f32x4 z = philox_normal4((uint32_t)site, (uint32_t)s, NK_RADIAL_R, step, seed);
  rr[idx] = z[0];
}

// packed sign words for one layer: [S*B examples][words]; example index is global (DP-invariant)
__global__ void gen_signs_kernel(uint32_t* dst, int words, int S, int B, int Bglob, int goff, int layer, uint32_t kind,
                                 uint64_t seed, uint32_t step) {
  const int w4 = (words + 3) >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)S * B * w4) return;
  const int q = (int)(idx % w4);
  const long ex = idx / w4;
  const int s = (int)(ex / B), b = (int)(ex - (long)s * B);
  const uint32_t gex = (uint32_t)(goff + b);
  const uint4 u = philox4x32_10(gex * (uint32_t)w4 + (uint32_t)q, (uint32_t)s, kind | ((uint32_t)layer << 8), step,
                                (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t v[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (q * 4 + r < words) dst[ex * words + q * 4 + r] = v[r];
}

// all layers' sign arrays in one launch: blockIdx.y = entry (layer x {in, out})
struct SignGenArgs {
  int n;
  int S, B, Bglob, goff;
  uint64_t seed;
  uint32_t step;
  uint32_t* dst[2 * BNN_MAX_LAYERS];
  int words[2 * BNN_MAX_LAYERS];
  int layer[2 * BNN_MAX_LAYERS];
  uint32_t kind[2 * BNN_MAX_LAYERS];
};
__global__ void gen_signs_all_kernel(const SignGenArgs A) {
  const int e = blockIdx.y;
  const int words = A.words[e];
  const int w4 = (words + 3) >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)A.S * A.B * w4) return;
  const int q = (int)(idx % w4);
  const long ex = idx / w4;
  const int s = (int)(ex / A.B), b = (int)(ex - (long)s * A.B);
  const uint32_t gex = (uint32_t)(A.goff + b);
  const uint4 u = philox4x32_10(gex * (uint32_t)w4 + (uint32_t)q, (uint32_t)s, A.kind[e] | ((uint32_t)A.layer[e] << 8),
                                A.step, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
  const uint32_t v[4] = {u.x, u.y, u.z, u.w};
  uint32_t* dst = A.dst[e];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (q * 4 + r < words) dst[ex * words + q * 4 + r] = v[r];
}

// Everything a Flipout training step on the conv trunk generates before the weights are sampled - the weight noise, every
// layer's sign words and the bf16 planes of the input windows - in ONE launch: three launches of a few microseconds each cost
// more in ramp-up and drain than in work.  Block ranges: [0, b_x) weight noise, [b_x, b_sg[0]) planes of x, then one range per
// sign entry (b_sg[e] .. b_sg[e + 1]).  The streams are those of gen_eps_w_kernel / gen_signs_all_kernel.
struct StepInputsArgs {
  float* eps; long P; int S; uint64_t eps_seed; uint32_t step;
  float* rad_r; int n_sites;   // radial guide: the S * n_sites radial distances ride on the last weight-noise block
  const float* x; u16* xp[4]; long rows; int L, F;
  float* xf[2];                // fp32 plan: x | pooled x planes [rows][20] instead of the four bf16 planes (null: bf16)
  SignGenArgs sg;
  unsigned b_x;
  unsigned b_sg[2 * BNN_MAX_LAYERS + 1];
};
__device__ __forceinline__ void x_planes4_dev8(const float* x, u16* hi, u16* lo, u16* phi, u16* plo, long rows, int L, int F, long idx8);
__device__ __forceinline__ void xf_planes_dev(const float* x, float* xp, float* xpp, long rows, int L, int F, long idx);

__global__ __launch_bounds__(256) void step_inputs_kernel(const StepInputsArgs A) {
  const unsigned blk = blockIdx.x;
  if (blk < A.b_x) {
    const long n4 = (A.P + 3) >> 2;
    const long idx = (long)blk * 256 + threadIdx.x;
    if (A.rad_r && blk == A.b_x - 1) {
      // (an extra block appended to the weight-noise range)
      for (int i = threadIdx.x; i < A.n_sites * A.S; i += 256) {
        const int rs = i / A.n_sites, site = i - rs * A.n_sites;
        A.rad_r[i] = philox_normal4((uint32_t)site, (uint32_t)rs, NK_RADIAL_R, A.step, A.eps_seed)[0];
      }
      return;
    }
    if (idx >= n4 * A.S) return;
    const int s = (int)(idx / n4);
    const long q = idx - (long)s * n4;
    const f32x4 z = philox_normal4((uint32_t)q, (uint32_t)s, NK_EPSW, A.step, A.eps_seed);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (q * 4 + r < A.P) A.eps[(long)s * A.P + q * 4 + r] = z[r];
    return;
  }
  if (blk < A.b_sg[0]) {
    if (A.xf[0]) xf_planes_dev(A.x, A.xf[0], A.xf[1], A.rows, A.L, A.F, (long)(blk - A.b_x) * 256 + threadIdx.x);
    else x_planes4_dev8(A.x, A.xp[0], A.xp[1], A.xp[2], A.xp[3], A.rows, A.L, A.F, (long)(blk - A.b_x) * 256 + threadIdx.x);
    return;
  }
  int e = 0;
  for (int k = 1; k < A.sg.n; ++k)
    if (blk >= A.b_sg[k]) e = k;
  const int words = A.sg.words[e];
  const int w4 = (words + 3) >> 2;
  const long idx = (long)(blk - A.b_sg[e]) * 256 + threadIdx.x;
  if (idx >= (long)A.sg.S * A.sg.B * w4) return;
  const int q = (int)(idx % w4);
  const long ex = idx / w4;
  const int s = (int)(ex / A.sg.B), b = (int)(ex - (long)s * A.sg.B);
  const uint32_t gex = (uint32_t)(A.sg.goff + b);
  const uint4 u = philox4x32_10(gex * (uint32_t)w4 + (uint32_t)q, (uint32_t)s, A.sg.kind[e] | ((uint32_t)A.sg.layer[e] << 8),
                                A.sg.step, (uint32_t)A.sg.seed, (uint32_t)(A.sg.seed >> 32));
  const uint32_t v[4] = {u.x, u.y, u.z, u.w};
  uint32_t* dst = A.sg.dst[e];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (q * 4 + r < words) dst[ex * words + q * 4 + r] = v[r];
}

// floats (+1/-1) [rows][C] -> packed bits (1 = negative)
__global__ void pack_signs_kernel(const float* src, uint32_t* dst, long rows, int C, int words) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * words) return;
  const long r = idx / words;
  const int w = (int)(idx - r * words);
  uint32_t bits = 0;
  for (int k = 0; k < 32; ++k) {
    const int c = w * 32 + k;
    if (c < C && src[r * C + c] < 0.f) bits |= (1u << k);
  }
  dst[idx] = bits;
}

__global__ void unpack_signs_kernel(const uint32_t* src, float* dst, long rows, int C, int words) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C) return;
  const long r = idx / C;
  const int c = (int)(idx - r * C);
  dst[idx] = ((src[r * words + (c >> 5)] >> (c & 31)) & 1u) ? -1.f : 1.f;
}

// LRT eps of one layer, written in the injected layout [rows][cout]
__global__ void export_lrt_eps_kernel(float* dst, long rows, int cout, int cout_p16, int Lrows, CallGeom cg, int layer,
                                      uint64_t seed, uint32_t step) {
  const int c4n = (cout + 3) >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * c4n) return;
  const long R = idx / c4n;
  const int c4 = (int)(idx - R * c4n);
  const long Rg = global_row(cg, Lrows, (int)R);
  const uint64_t gi = (uint64_t)Rg * (uint64_t)(cout_p16 >> 2) + (uint64_t)c4;
  const f32x4 z = philox_normal4((uint32_t)gi, (uint32_t)(gi >> 32), NK_LRT | ((uint32_t)layer << 8), step, seed);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (c4 * 4 + r < cout) dst[R * cout + c4 * 4 + r] = z[r];
}

// ------------------------------------------------------------------------------------------
// radial: per (particle, site) L2 norm of eps  (guides/radial.py:38)
// ------------------------------------------------------------------------------------------
// Two launches: partial sums of squares over chunks of SN_CHUNK elements (a site of 153,600 elements as ONE workgroup was a
// 150 us chain of dependent loads at S = 20: 15 % of the radial step), then one thread per (particle, site) adds the
// chunks in order (fixed order: reproducible).
enum { SN_CHUNK = 8192 };
struct SiteChunks {
  int total;                       // chunks per particle
  int start[BNN_MAX_SITES + 1];    // first chunk of site i
};

__global__ __launch_bounds__(256) void site_norm_part_kernel(const float* eps_w, long P, ParamTable T, SiteChunks C, double* part) {
  const int s = blockIdx.x / C.total, ch = blockIdx.x - s * C.total;
  int site = 0;
  for (int i = 1; i < T.n_sites; ++i)
    if (ch >= C.start[i]) site = i;
  const long k0 = (long)(ch - C.start[site]) * SN_CHUNK;
  const long n = min((long)SN_CHUNK, T.site[site].numel - k0);
  const float* e = eps_w + (long)s * P + T.site[site].off + k0;
  double acc = 0.0;
  for (int j0 = 0; j0 < SN_CHUNK / 256; j0 += 8) {   // 8 loads in flight per thread
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long k = (long)(j0 + u) * 256 + threadIdx.x;
      v[u] = k < n ? e[k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (double)v[u] * (double)v[u];
  }
  __shared__ double red[4];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)s * C.total + ch] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void site_norm_fin_kernel(const double* part, ParamTable T, SiteChunks C, int S, float* norms) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * T.n_sites) return;
  const int s = i / T.n_sites, site = i - s * T.n_sites;
  double t = 0.0;
  for (int c = C.start[site]; c < C.start[site + 1]; ++c) t += part[(long)s * C.total + c];
  norms[i] = (float)sqrt(t);
}

// ------------------------------------------------------------------------------------------
// weight sampling + image construction + KL / (log q - log p)
// ------------------------------------------------------------------------------------------
struct PrepArgs {
  ParamTable T;
  const LayerDesc* layers;
  const float* mu;
  const float* rho;
  const float* eps_w;   // [S][P]
  const float* rad_r;   // [S][n_sites]
  const float* norms;   // [S][n_sites]
  int mode;             // BNN_MODE_* (0 normal, 1 lrt, 2 flipout, 3 radial)
  int S;
  int want_t;           // build transposed images
  void* a_hi; void* a_lo; void* b; void* at; void* bt;
  long slot_stride;     // elements between particles (fwd images)
  long slott_stride;
  float* bias_a; float* bias_b;
  int bias_total;
  double* kl_acc;       // [S] (radial) or [1]
  float prior_loc, prior_scale;
  long skip_lo, skip_hi; // elements [skip_lo, skip_hi) of the flat buffers are handled by prep_flat_dense_kernel (0, 0: none)
};

template <class P>
__device__ __forceinline__ void put_img(void* hi, void* lo, long idx, float v) {
  if constexpr (!P::BF) {
    ((float*)hi)[idx] = v;
  } else {
    const u16 h = f2bf(v);
    ((u16*)hi)[idx] = h;
    if (lo) ((u16*)lo)[idx] = f2bf(v - bf2f(h));
  }
}

template <class P>
__device__ __forceinline__ void prep_weights_body(const PrepArgs& A, unsigned bid) {
  long e = (long)bid * blockDim.x + threadIdx.x;
  if (e >= A.skip_lo) e += A.skip_hi - A.skip_lo;   // the grid covers the elements outside the skipped range only
  const bool live = e < A.T.P;
  const int lane = threadIdx.x & 63;
  __shared__ double red[16];   // up to 1024 threads per workgroup
  constexpr int RS_MAX = 128;
  __shared__ double red_s[RS_MAX];   // radial: per-particle partial sums of this workgroup
  const int radial = (A.mode == 3);
  if (radial) {
    for (int k = threadIdx.x; k < RS_MAX; k += blockDim.x) red_s[k] = 0.0;
    __syncthreads();
  }
  double kl_local = 0.0;  // mean-field KL contribution of this element
  int si = 0, n = 0, is_bias = 0, bias_idx = 0;
  long fi = 0, ti = 0;
  float mu = 0.f, rho = 0.f, sigma = 1.f;
  if (live) {
    si = find_site(A.T, e);
    const SiteDesc sd = A.T.site[si];
    const LayerDesc& ly = A.layers[sd.layer];
    const long le = e - sd.off;
    is_bias = sd.is_bias;
    mu = A.mu[e];
    rho = A.rho[e];
    sigma = expf(rho);
    n = (int)le;
    if (!is_bias) map_weight(ly, le, n, fi, ti);
    fi += ly.w_off;
    ti += ly.wt_off;
    bias_idx = ly.bias_off + n;
    if (!radial) {
      const float vr = (sigma / A.prior_scale) * (sigma / A.prior_scale);
      const float t1 = ((mu - A.prior_loc) / A.prior_scale) * ((mu - A.prior_loc) / A.prior_scale);
      kl_local = 0.5 * ((double)vr + (double)t1 - 1.0 - (double)logf(vr));
    }
    if (A.mode == 1) {  // LRT: shared (mu, sigma^2)
      if (is_bias) {
        A.bias_a[bias_idx] = mu;
        A.bias_b[bias_idx] = sigma * sigma;
      } else {
        put_img<P>(A.a_hi, A.a_lo, fi, mu);
        put_img<P>(A.b, nullptr, fi, sigma * sigma);
        if (A.want_t) {
          put_img<P>(A.at, nullptr, ti, mu);
          put_img<P>(A.bt, nullptr, ti, sigma * sigma);
        }
      }
    } else if (A.mode == 2 && !is_bias) {  // flipout: shared mean image
      put_img<P>(A.a_hi, A.a_lo, fi, mu);
      if (A.want_t) put_img<P>(A.at, nullptr, ti, mu);
    }
  }
  if (A.mode != 1) {
    // the particles' noise values of this element: loads of up to 16 particles in flight (a load per iteration followed
    // by its dependent stores made the kernel latency-bound: S round trips per thread)
    constexpr int NB = 16;   // particles per batch of loads
    for (int s0 = 0; s0 < A.S; s0 += NB) {
      float ev[NB], rv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int s = min(s0 + u, A.S - 1);
        ev[u] = 0.f;
        rv[u] = 1.f;
        if (live) {
          ev[u] = A.eps_w[(long)s * A.T.P + e];
          if (radial) rv[u] = A.rad_r[s * A.T.n_sites + si] / A.norms[s * A.T.n_sites + si];
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
      const int s = s0 + u;
      if (s >= A.S) break;
      double term = 0.0;
      if (live) {
        float eps = ev[u];
        if (radial) eps = eps * rv[u];
        const float dw = sigma * eps;
        const float w = mu + dw;
        if (is_bias) {
          A.bias_a[(long)s * A.bias_total + bias_idx] = w;
        } else if (A.mode == 2) {
          put_img<P>(A.b, nullptr, A.slot_stride * s + fi, dw);
          if (A.want_t) put_img<P>(A.bt, nullptr, A.slott_stride * s + ti, dw);
        } else {
          put_img<P>(A.a_hi, A.a_lo, A.slot_stride * s + fi, w);
          if (A.want_t) put_img<P>(A.at, nullptr, A.slott_stride * s + ti, w);
        }
        if (radial) {
          // log q(w) - log p(w), Normal.log_prob of the radial sample (A6)
          const float zp = (w - A.prior_loc) / A.prior_scale;
          term = -0.5 * (double)eps * (double)eps - (double)rho + 0.5 * (double)zp * (double)zp +
                 (double)logf(A.prior_scale);
        }
      }
      if (radial) {  // all lanes of the wave take part in the reduction
        const double t = wave_sum_d(term);
        if (lane == 0) {
          // one global fp64 atomic per (workgroup, particle) instead of one per wave: S*P/64 atomics on S addresses
          // used to cost 0.4 ms at S = 20
          if (s < RS_MAX) atomicAdd(&red_s[s], t);
          else atomicAdd(A.kl_acc + s, t);
        }
      }
      }
    }
    if (radial) {
      __syncthreads();
      for (int s = threadIdx.x; s < A.S && s < RS_MAX; s += blockDim.x) atomicAdd(A.kl_acc + s, red_s[s]);
    }
  }
  if (!radial) {
    const double t = wave_sum_d(kl_local);
    if (lane == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tt = 0;
      for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tt += red[k];
      atomicAdd(A.kl_acc, tt);
    }
  }
}

template <class P>
__global__ void prep_weights_kernel(const PrepArgs A) {
  prep_weights_body<P>(A, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// The same for the weights of a dense layer that follows nn.Flatten of [C][L] activations kept as [L][C] rows (the Inception
// net's Linear(2400, 64): 80 % of its parameters), bf16 plans, every estimator but LRT.  In the canonical order (cout, c, l)
// a thread per element stores 2 bytes at stride 2 C in the forward image [cout][l C + c] and at stride 2 KPt in the
// transposed one [l C + c][cout]: 64 write requests per wave store, which - not bytes - is what bounded the one-thread-per-
// element kernel.  Here a workgroup takes a (16 cout, 8 c, all l) tile of one particle: coalesced runs of 8 L floats in,
// through LDS, 16-byte pieces out (8 channels of a forward row, 8 couts of a transposed row).
// grid.x = (cout / 16) * (C / 8) tiles, grid.y = particles (+ 1 for Flipout: the shared mean images and the KL).
// ------------------------------------------------------------------------------------------
// C, Lw are template parameters: the index arithmetic is divisions by them (by a run-time value: ~40 instructions each,
// and the kernel was bound by exactly that).
enum { PF_TN = 16, PF_TC = 8 };
template <class P, int C, int Lw>
__device__ __forceinline__ void prep_flat_dense_body(const PrepArgs& A, int layer_id, int site, int tile_blk, int s) {
  typedef typename std::conditional<P::BF, u16, float>::type elem_t;   // image element: bf16 (hi [+ lo]) or fp32
  constexpr int EPV = 16 / (int)sizeof(elem_t);                        // elements per 16-byte store
  extern __shared__ __attribute__((aligned(16))) char pf_smem[];
  const LayerDesc ly = A.layers[layer_id];
  constexpr int ncg = C / PF_TC;
  const int ng = tile_blk / ncg, cg = tile_blk - ng * ncg;
  const int n0 = ng * PF_TN, c0 = cg * PF_TC;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool mean_pass = (A.mode == 2 && s == A.S);
  const bool radial = (A.mode == 3);
  const bool want_kl = mean_pass || (A.mode == 0 && s == 0);
  constexpr int run = PF_TC * Lw, tile = PF_TN * run;
  elem_t* th = (elem_t*)pf_smem;
  elem_t* tl = th + tile;   // (bf16 only: the lo plane)
  const long off = A.T.site[site].off;
  float rr = 1.f;
  if (radial) rr = A.rad_r[s * A.T.n_sites + site] / A.norms[s * A.T.n_sites + site];
  const long so = mean_pass ? 0 : (long)s * A.T.P;
  double acc = 0.0;
  // all loads of the tile first (PF_IT iterations of 256 elements, L <= 32), then the arithmetic: a load per iteration
  // followed by its dependent LDS stores would be PF_IT round trips
  constexpr int PF_IT = (tile + 255) / 256;
  float mu_r[PF_IT], rho_r[PF_IT], eps_r[PF_IT];
#pragma unroll
  for (int it = 0; it < PF_IT; ++it) {
    const int idx = min(it * 256 + tid, tile - 1);
    const int nl = idx / run, rem = idx - nl * run;
    const int cl = rem / Lw, l = rem - cl * Lw;
    const long e = off + (long)(n0 + nl) * ly.cin + (long)(c0 + cl) * Lw + l;
    mu_r[it] = A.mu[e];
    rho_r[it] = A.rho[e];
    eps_r[it] = mean_pass ? 0.f : A.eps_w[so + e];
  }
#pragma unroll
  for (int it = 0; it < PF_IT; ++it) {
    const int idx = it * 256 + tid;
    if (idx >= tile) break;
    const int nl = idx / run, rem = idx - nl * run;
    const int cl = rem / Lw, l = rem - cl * Lw;
    const float mu = mu_r[it], rho = rho_r[it];
    const float sigma = expf(rho);
    float v = mu;
    if (!mean_pass) {
      float eps = eps_r[it];
      if (radial) eps = eps * rr;
      const float dw = sigma * eps;
      const float w = mu + dw;
      v = A.mode == 2 ? dw : w;
      if (radial) {
        const float zp = (w - A.prior_loc) / A.prior_scale;
        acc += -0.5 * (double)eps * (double)eps - (double)rho + 0.5 * (double)zp * (double)zp + (double)logf(A.prior_scale);
      }
    }
    if (want_kl) {
      const float vr = (sigma / A.prior_scale) * (sigma / A.prior_scale);
      const float t1 = ((mu - A.prior_loc) / A.prior_scale) * ((mu - A.prior_loc) / A.prior_scale);
      acc += 0.5 * ((double)vr + (double)t1 - 1.0 - (double)logf(vr));
    }
    const int o = nl * run + l * PF_TC + cl;
    if constexpr (P::BF) {
      const u16 h = f2bf(v);
      th[o] = h;
      tl[o] = f2bf(v - bf2f(h));
    } else {
      th[o] = v;
    }
  }
  __syncthreads();
  // destinations: Flipout particle pass -> slot B (single bf16); mean pass -> the shared slot A; normal / radial -> slot A of s
  elem_t *d_hi, *d_lo, *d_t;
  if (A.mode == 2 && !mean_pass) {
    d_hi = (elem_t*)A.b + A.slot_stride * s; d_lo = nullptr; d_t = (elem_t*)A.bt + A.slott_stride * s;
  } else if (mean_pass) {
    d_hi = (elem_t*)A.a_hi; d_lo = (elem_t*)A.a_lo; d_t = (elem_t*)A.at;
  } else {
    d_hi = (elem_t*)A.a_hi + A.slot_stride * s; d_lo = P::BF ? (elem_t*)A.a_lo + A.slot_stride * s : nullptr; d_t = (elem_t*)A.at + A.slott_stride * s;
  }
  constexpr int FV = PF_TC / EPV;   // 16-byte pieces of a forward row's 8 channels
  for (int u = tid; u < PF_TN * Lw * FV; u += 256) {
    const int piece = u % FV, r = u / FV;
    const int nl = r / Lw, l = r - nl * Lw;
    const long fi = ly.w_off + (long)(n0 + nl) * ly.KP + (long)l * C + c0 + piece * EPV;
    *(uint4*)(d_hi + fi) = *(const uint4*)(th + nl * run + l * PF_TC + piece * EPV);
    if constexpr (P::BF) {
      if (d_lo) *(uint4*)(d_lo + fi) = *(const uint4*)(tl + nl * run + l * PF_TC + piece * EPV);
    }
  }
  if constexpr (!P::BF) {
    if (A.want_t) {
      for (int u = tid; u < run * 4; u += 256) {
        const int quarter = u & 3, q = u >> 2;   // q = l * TC + cl
        const int l = q / PF_TC, cl = q - l * PF_TC;
        f32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = th[(quarter * 4 + j) * run + q];
        const long ti = ly.wt_off + (long)(l * C + c0 + cl) * ly.KPt + n0 + quarter * 4;
        *(f32x4*)(d_t + ti) = w;
      }
    }
  } else if (A.want_t) {
    for (int u = tid; u < run * 2; u += 256) {
      const int half = u & 1, q = u >> 1;   // q = l * TC + cl
      const int l = q / PF_TC, cl = q - l * PF_TC;
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int na = half * 8 + 2 * j;
        w[j] = (uint32_t)th[na * run + q] | ((uint32_t)th[(na + 1) * run + q] << 16);
      }
      const long ti = ly.wt_off + (long)(l * C + c0 + cl) * ly.KPt + n0 + half * 8;
      *(uint4*)(d_t + ti) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  if (want_kl || radial) {
    __shared__ double red[4];
    const double t = wave_sum_d(acc);
    if (lane == 0) red[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) atomicAdd(A.kl_acc + (radial ? s : 0), (red[0] + red[1]) + (red[2] + red[3]));
  }
}

// One launch for both: workgroups [0, n_tiles * ny) are (tile, particle) pairs of the wide dense layer, the rest walk the
// remaining elements one thread each (two launches of ~15 us of mostly latency each were slower than the single
// one-thread-per-element launch they replaced; side by side they are not).
template <class P, int C, int Lw>
__global__ __launch_bounds__(256) void prep_fused_kernel(const PrepArgs A, int layer_id, int site, unsigned n_tiles, unsigned ny) {
  if (blockIdx.x < n_tiles * ny) prep_flat_dense_body<P, C, Lw>(A, layer_id, site, (int)(blockIdx.x % n_tiles), (int)(blockIdx.x / n_tiles));
  else prep_weights_body<P>(A, blockIdx.x - n_tiles * ny);
}

// ------------------------------------------------------------------------------------------
// head: softplus -> Threshold(1e-9) on both outputs, likelihood softplus on the scale,
// Gaussian log-lik and d(-ll)/dz  (A11)
// ------------------------------------------------------------------------------------------
struct HeadArgs {
  const float* z;   // [S*B][2]
  const float* y;   // [B]
  float* dz;        // [S*B][2] or null
  float* preds;     // [S*B][2] or null
  double* ll_acc;   // [S]
  int S, B;
  int with_obs;
  int objective;    // 0: TyXe HeteroskedasticGaussian (A11); 1: HNN gaussian_nll_loss; 2: NN mse  (frequentist.py:39-48,173-178)
};

// one example row of the head: predictions, log-likelihood term (returned) and d(-ll)/dz (g0, g1; also stored when A.dz)
__device__ __forceinline__ double head_row(const HeadArgs& A, int s, int idx, float& g0o, float& g1o) {
  double ll = 0.0;
  g0o = g1o = 0.f;
  const long r = (long)s * A.B + idx;
  const float z0 = A.z[r * 2], z1 = A.z[r * 2 + 1];
  const float sp0 = softplus_t(z0), sp1 = softplus_t(z1);
  const bool p0 = sp0 > 1e-9f, p1 = sp1 > 1e-9f;
  const float o0 = p0 ? sp0 : 1e-9f, o1 = p1 ? sp1 : 1e-9f;
  if (A.preds) {
    A.preds[r * 2] = o0;
    A.preds[r * 2 + 1] = o1;
  }
  if (A.with_obs && A.objective != 0) {
    // frequentist siblings: ll = -loss_b; scale = o1 (the net's own softplus + threshold only)
    const float d = o0 - A.y[idx];
    float g0, g1 = 0.f;
    if (A.objective == 1) {
      const float var = fmaxf(o1 * o1, 1e-6f);   // F.gaussian_nll_loss: var.clamp_(min=eps) under no_grad
      ll = -0.5 * ((double)logf(var) + (double)(d * d) / (double)var);
      g0 = d / var;
      g1 = (0.5f / var - 0.5f * d * d / (var * var)) * 2.f * o1;   // the clamp carries no gradient: d var / d o1 = 2 o1
    } else {
      ll = -(double)(d * d);
      g0 = 2.f * d;
    }
    g0o = p0 ? g0 * dsoftplus_t(z0) : 0.f;
    g1o = p1 ? g1 * dsoftplus_t(z1) : 0.f;
  } else if (A.with_obs) {
    const float sc = softplus_t(o1);
    const float d = A.y[idx] - o0;
    ll = -(double)(d * d) / (2.0 * (double)sc * (double)sc) - (double)logf(sc) - 0.9189385332046727;
    // d(-ll)/d o0 = -(y - o0)/s^2 ; d(-ll)/d s = -(y-o0)^2/s^3 + 1/s
    const float g0 = -d / (sc * sc);
    const float gs = -(d * d) / (sc * sc * sc) + 1.f / sc;
    const float g1 = gs * dsoftplus_t(o1);
    g0o = p0 ? g0 * dsoftplus_t(z0) : 0.f;
    g1o = p1 ? g1 * dsoftplus_t(z1) : 0.f;
  }
  if (A.dz) {
    A.dz[r * 2] = g0o;
    A.dz[r * 2 + 1] = g1o;
  }
  return ll;
}

__global__ void head_nll_kernel(const HeadArgs A) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  double ll = 0.0;
  float g0, g1;
  if (idx < A.B) ll = head_row(A, s, idx, g0, g1);
  __shared__ double red[4];
  ll = wave_sum_d(ll);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ll;
  __syncthreads();
  if (threadIdx.x == 0 && A.with_obs) {
    double t = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
    atomicAdd(A.ll_acc + s, t);
  }
}

// loss = (1/S) sum_s [ c*KL_s - c*(N/B)*ll_s ]   (A5 / A6); written to out + grad[2P], grad[2P+1]
struct LossArgs {
  double* kl_acc;
  const double* ll_acc;
  int S;
  int radial;
  double c, n_over_b;
  double kl_weight;   // 1: ELBO; 0: the frequentist objectives (loss = -c * n_over_b * ll)
  float* loss; float* kl; float* ll;
  float* grad_tail;  // &grad[2P] or null
  int n_acc;         // doubles to zero after use (all accumulators: 2 * (max_particles + 1))
};

// one thread: the loss scalars from the accumulators.  REARM: the accumulators are zeroed for the next call (the host
// then skips its memset: plan.hip acc_clean)
__device__ __forceinline__ void finish_loss_dev(const LossArgs& A) {
  double kl = 0, ll = 0;
  for (int s = 0; s < A.S; ++s) {
    kl += A.radial ? A.kl_acc[s] : A.kl_acc[0];
    ll += A.ll_acc[s];
  }
  kl /= A.S;
  ll /= A.S;
  const double loss = A.kl_weight * A.c * kl - A.c * A.n_over_b * ll;
  if (A.loss) *A.loss = (float)loss;
  if (A.kl) *A.kl = (float)kl;
  if (A.ll) *A.ll = (float)ll;
  if (A.grad_tail) {
    A.grad_tail[0] = (float)loss;
    A.grad_tail[1] = (float)kl;
  }
  for (int k = 0; k < A.n_acc; ++k) A.kl_acc[k] = 0.0;   // kl | ll accumulators are back to back
}

__global__ void finish_loss_kernel(const LossArgs A) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  finish_loss_dev(A);
}

// ------------------------------------------------------------------------------------------
// ClippedAdam on the flat (mu, rho) buffer  (A12)
// ------------------------------------------------------------------------------------------
struct AdamArgs {
  float* mu; float* rho; float* m; float* v; const float* grad;
  long P;
  float lr, beta1, beta2, eps, clip, wd, step_size, grad_scale;
  int freeze_loc, freeze_scale;
};

// element i of the flat (mu | rho) buffer, raw gradient g0
__device__ __forceinline__ void adam_update(const AdamArgs& A, long i, float g0) {
  if (i < A.P ? A.freeze_loc : A.freeze_scale) return;
  float* p = i < A.P ? A.mu + i : A.rho + (i - A.P);
  // explicitly rounded operations (no mul+add contraction left to the optimiser): the update is the same arithmetic whether
  // it is inlined behind the chain rule (grad_finalize_kernel) or runs from the gradient buffer (clipped_adam_kernel, the DP
  // path) - test_dp_step_world1_equals_plain_step holds the two bit for bit
  float g = __fmul_rn(g0, A.grad_scale);
  g = fminf(fmaxf(g, -A.clip), A.clip);
  if (A.wd != 0.f) g = __fmaf_rn(A.wd, *p, g);
  const float m = __fmaf_rn(A.beta1, A.m[i], __fmul_rn(1.f - A.beta1, g));
  const float v = __fmaf_rn(A.beta2, A.v[i], __fmul_rn(__fmul_rn(1.f - A.beta2, g), g));
  A.m[i] = m;
  A.v[i] = v;
  *p = __fsub_rn(*p, __fdiv_rn(__fmul_rn(A.step_size, m), __fadd_rn(__fsqrt_rn(v), A.eps)));
}

__global__ void clipped_adam_kernel(const AdamArgs A) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * A.P) return;
  adam_update(A, i, A.grad[i]);
}

// ------------------------------------------------------------------------------------------
// chain rule: weight-image gradients -> d loss / d (mu, rho)   (+ KL / prior terms)
//   G = d(-sum_b ll_b)/dW per particle (unscaled); scale_ll = c*(N/B)/S
// ------------------------------------------------------------------------------------------
struct FinalizeArgs {
  ParamTable T;
  const LayerDesc* layers;
  const float* mu; const float* rho;
  const float* eps_w; const float* rad_r; const float* norms;
  const float* gw_a; const float* gw_b; const float* gb_a; const float* gb_b;
  long gw_stride; int gb_stride;
  int mode, S;
  float scale_ll;   // c * (N/B) / S   (0 when with_obs == 0)
  float c;          // KL scale
  float prior_loc, prior_scale;
  float* grad;      // [2P]
  LossArgs loss;    // fused_loss: thread 0 of workgroup 0 also finishes the loss scalars (the accumulators are complete:
  int fused_loss;   // the head ran earlier on the stream)
  AdamArgs adam;    // fused_adam: the optimizer update of the element follows at once (single-GPU step: no all-reduce
  int fused_adam;   // between the two)
};

__global__ void grad_finalize_kernel(const FinalizeArgs A) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (A.fused_loss && e == 0) finish_loss_dev(A.loss);
  if (e >= A.T.P) return;
  const int si = find_site(A.T, e);
  const SiteDesc sd = A.T.site[si];
  const LayerDesc& ly = A.layers[sd.layer];
  const long le = e - sd.off;
  const float mu = A.mu[e], rho = A.rho[e];
  const float sigma = expf(rho);
  int n = (int)le;
  long fi = 0, ti = 0;
  if (!sd.is_bias) map_weight(ly, le, n, fi, ti);
  const long gi = sd.is_bias ? (long)(ly.bias_off + n) : (ly.w_off + fi);
  const float* ga = sd.is_bias ? A.gb_a : A.gw_a;
  const float* gb = sd.is_bias ? A.gb_b : A.gw_b;
  const long gs = sd.is_bias ? (long)A.gb_stride : A.gw_stride;
  const float inv_s0sq = 1.f / (A.prior_scale * A.prior_scale);
  float dmu = 0.f, drho = 0.f;
  if (A.mode == 1) {  // LRT
    float sa = 0.f, sb = 0.f;
    for (int s = 0; s < A.S; ++s) {
      sa += ga[gs * s + gi];
      sb += gb[gs * s + gi];
    }
    dmu = A.scale_ll * sa;
    drho = A.scale_ll * sb * 2.f * sigma * sigma;
  } else if (A.mode == 2) {  // flipout: mean path shared, perturbation dW_s = sigma*eps_s
    float sa = 0.f, sb = 0.f;
#pragma unroll 4
    for (int s = 0; s < A.S; ++s) {
      const float eps = A.eps_w[(long)s * A.T.P + e];
      const float a = ga[gs * s + gi];
      sa += a;
      // bias: b_s = mu_b + sigma_b*eps, its gradient arrives through slot A
      sb += (sd.is_bias ? a : gb[gs * s + gi]) * eps;
    }
    dmu = A.scale_ll * sa;
    drho = A.scale_ll * sb * sigma;
  } else if (A.mode == 0) {  // plain normal sampling
    float sa = 0.f, sb = 0.f;
#pragma unroll 4
    for (int s = 0; s < A.S; ++s) {
      const float a = ga[gs * s + gi];
      sa += a;
      sb += a * A.eps_w[(long)s * A.T.P + e];
    }
    dmu = A.scale_ll * sa;
    drho = A.scale_ll * sb * sigma;
  } else {  // radial + Trace_ELBO: pathwise through w, log p(w); d log q/d rho = -1
    const float cs = A.c / (float)A.S;
#pragma unroll 4
    for (int s = 0; s < A.S; ++s) {
      const float er = A.eps_w[(long)s * A.T.P + e] * (A.rad_r[s * A.T.n_sites + si] / A.norms[s * A.T.n_sites + si]);
      const float w = mu + sigma * er;
      const float gwt = A.scale_ll * ga[gs * s + gi] + cs * (w - A.prior_loc) * inv_s0sq;
      dmu += gwt;
      drho += gwt * sigma * er - cs;
    }
  }
  if (A.mode != 3) {  // closed-form KL(N(mu,sigma)||N(mu0,sigma0)) gradient
    dmu += A.c * (mu - A.prior_loc) * inv_s0sq;
    drho += A.c * (sigma * sigma * inv_s0sq - 1.f);
  }
  A.grad[e] = dmu;
  A.grad[A.T.P + e] = drho;
  if (A.fused_adam) {
    adam_update(A.adam, e, dmu);
    adam_update(A.adam, A.T.P + e, drho);
  }
}

// ------------------------------------------------------------------------------------------
// Sum of the partial gradient images ("slabs") the fused trunk dW kernels write, one per workgroup, into the
// per-particle images grad_finalize_kernel reads.  Layer group g (0: layers 0-3, 1: layers 4, 5, 7, 9, 2: layers 6, 8)
// owns n[g] slabs per particle; a slab has the forward image layout of the conv layers (stride floats).  Fixed summation
// order: bitwise reproducible.  One thread per (particle, image element); eight independent partial sums keep eight
// loads in flight.
// ------------------------------------------------------------------------------------------
struct SlabReduceArgs {
  const float* slab[3];     // [S * n[g]][stride]
  int n[3];
  long stride;              // floats per slab
  long lay_end[10];         // end offset of conv layer l inside a slab (images are in layer order)
  float* out;               // [S][out_stride]
  long out_stride;
  long elems;               // elements to reduce (= lay_end[9])
  int S;
};

// the (up to) three reductions of a step - slot A, slot B, bias sums - in one launch: blockIdx.z picks the job
struct SlabReduceJobs {
  SlabReduceArgs job[4];
};

__global__ void slab_reduce_kernel(const SlabReduceJobs J) {
  const SlabReduceArgs& A = J.job[blockIdx.z];
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (e >= A.elems) return;
  int layer = 0;
#pragma unroll
  for (int l = 0; l < 9; ++l)
    if (e >= A.lay_end[l]) layer = l + 1;
  const int g = layer < 4 ? 0 : ((layer == 6 || layer == 8) ? 2 : 1);
  const int n = A.n[g];
  const float* p = A.slab[g] + (long)s * n * A.stride + e;
  float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= n; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += p[(long)(k + u) * A.stride];
  }
  for (; k < n; ++k) t[0] += p[(long)k * A.stride];
  A.out[(long)s * A.out_stride + e] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

// ------------------------------------------------------------------------------------------
// MaxPool1d(3,1,1) backward: dX[r][c] += sum_{r' in {r-1,r,r+1}} dP[r'][c] * [argmax(r') == r]
// (first maximum wins on ties, as torch's max_pool backward)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float tget(const TensorRef& t, long o) {
  if (t.fmt == TF_F32) return ((const float*)t.p)[o];
  return bf2f(((const u16*)t.p)[o]) + (t.lo ? bf2f(((const u16*)t.lo)[o]) : 0.f);
}
__device__ __forceinline__ void tput(const TensorRef& t, long o, float v) {
  if (t.fmt == TF_F32) {
    ((float*)t.p)[o] = v;
    return;
  }
  const u16 h = f2bf(v);
  ((u16*)t.p)[o] = h;
  if (t.lo) ((u16*)t.lo)[o] = f2bf(v - bf2f(h));
}

__global__ void pool_bwd_kernel(const TensorRef X, const TensorRef dP, const TensorRef dX, long nwin, int L, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nwin * L * C) return;
  const int c = (int)(idx % C);
  const long rw = idx / C;
  const int r = (int)(rw % L);
  const long w = rw / L;
  const long base = w * L * C + c;
  float acc = 0.f;
  for (int rp = max(r - 1, 0); rp <= min(r + 1, L - 1); ++rp) {
    int am = -1;
    float best = 0.f;
    for (int k = max(rp - 1, 0); k <= min(rp + 1, L - 1); ++k) {
      const float v = tget(X, base + (long)k * C);
      if (am < 0 || v > best) {
        best = v;
        am = k;
      }
    }
    if (am == r) acc += tget(dP, base + (long)rp * C);
  }
  tput(dX, idx, tget(dX, idx) + acc);
}


// ------------------------------------------------------------------------------------------
// predictive aggregation over particles (A16), two-pass variance
// ------------------------------------------------------------------------------------------
__global__ void predict_finish_kernel(const float* preds /*[S][B][2]*/, int B, int S, float* out4) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s1 = 0.f, sv = 0.f;
  const float2* pp = (const float2*)preds + b;   // (loc, scale) pairs; the sums keep the order s = 0, 1, ...
#pragma unroll 10
  for (int s = 0; s < S; ++s) {
    const float2 v = pp[(long)s * B];
    s1 += v.x;
    sv += v.y * v.y;
  }
  const float mean = s1 / S;
  float ss = 0.f;
#pragma unroll 10
  for (int s = 0; s < S; ++s) {
    const float d = pp[(long)s * B].x - mean;
    ss += d * d;
  }
  // unbiased variance (torch.var default); NaN at S == 1 like the reference
  const float ep = ss / (float)(S - 1);
  const float al = sv / S;
  out4[b] = mean;
  out4[B + b] = sqrtf(al + ep);
  out4[2 * B + b] = ep;
  out4[3 * B + b] = al;
}

// ------------------------------------------------------------------------------------------
// window store: gather a batch of windows by index from an HBM-resident set.
// feature_major = 1: a stored window is the reference's LMDB value, [F][W] fp32 (data/lmdb_utils.py:190-191 reads
// it with reshape(n_features, -1).T), and is transposed to the [W][F] the step consumes.
// ------------------------------------------------------------------------------------------
__global__ void gather_windows_kernel(const float* x_all, const float* y_all, const long* idx, long n, long n_windows, int W,
                                      int F, int feature_major, float* x_out, float* y_out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int wf = W * F;
  if (e >= n * wf) return;
  const long i = e / wf;
  const int r = (int)(e - i * wf);
  const long j = idx[i];
  const bool ok = j >= 0 && j < n_windows;   // a bad index is never dereferenced: NaN marks it
  const int w = r / F, f = r - w * F;
  x_out[e] = ok ? x_all[j * wf + (feature_major ? f * W + w : r)] : __int_as_float(0x7fc00000);
  if (r == 0 && y_all) y_out[i] = ok ? y_all[j] : __int_as_float(0x7fc00000);
}

