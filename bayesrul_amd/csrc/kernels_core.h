// Window staging + MFMA contraction cores (device functions) for gfx950.
//
//   image  : one window of a channels-last activation tensor in LDS, IMG_ROWS x RS elements,
//            row r of the window at image row r + HALO; rows outside [0, nvalid) are zero, so a
//            'same'-padded k-tap conv is a plain dot product of K = taps * C contiguous-per-tap
//            elements and needs no masking in the inner loop.
//   gemm_f : out[n][row] = sum_{tap,c} W[n][tap][c] * X[row + tap - pad][c]
//            (forward conv / linear; with the transposed+flipped weight image it is dX).
//            MFMA orientation: A = weights (i = cout), B = window rows (j = row), so a lane ends
//            up with 4 consecutive channels of one row -> one 16-byte store.
//   gemm_w : dW[n][tap][c] += sum_row dZ[row][n] * X[row + tap - pad][c]   (k = rows; the
//            bf16 form feeds both operands through ds_read_b64_tr_b16).
//
// Two arithmetic policies:
//   PrecF32 : v_mfma_f32_16x16x4_f32, exact fp32 (parity path)
//   PrecBF  : v_mfma_f32_16x16x32_bf16; forward mean path split hi+lo (3 MFMAs), everything
//             else single bf16; fp32 accumulation
#pragma once
#include "common.h"
#include "desc.h"

struct PrecF32 {
  static constexpr bool BF = false;
  using elem = float;
};
struct PrecBF {
  static constexpr bool BF = true;
  using elem = u16;
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ------------------------------------------------------------------------------------------
// window geometry
// ------------------------------------------------------------------------------------------
struct Win {
  int s;         // particle
  int in_row0;   // first row of the window in the input tensor
  int out_row0;  // first row in the [S*B*L] (conv) / [S*B] (dense) row space of outputs
  int nvalid;    // valid rows
  int ex0;       // example index (s*B + b) of row 0; dense: rows are consecutive examples
};

__device__ __forceinline__ Win decode_win(const GroupDesc& g, const CallGeom& cg, int win) {
  Win w;
  w.s = win / cg.per_particle;
  const int r = win - w.s * cg.per_particle;
  if (!g.is_dense) {
    w.in_row0 = (g.in_bcast ? r : w.s * cg.B + r) * g.L;
    w.out_row0 = (w.s * cg.B + r) * g.L;
    w.nvalid = g.L;
    w.ex0 = w.s * cg.B + r;
  } else {
    w.in_row0 = (g.in_bcast ? 0 : w.s * cg.B) + r * TILE_ROWS;
    w.out_row0 = w.s * cg.B + r * TILE_ROWS;
    w.nvalid = min(TILE_ROWS, cg.B - r * TILE_ROWS);
    w.ex0 = w.out_row0;
  }
  return w;
}

// global (DP-invariant) row index of local output row R: rows are [S][B][L]; the rank holds
// examples [goff, goff+B) of a global batch of Bglob.
__device__ __forceinline__ long global_row(const CallGeom& cg, int L, int R) {
  const int per_s = cg.B * L;
  const int s = R / per_s;
  const int rem = R - s * per_s;
  return ((long)s * cg.Bglob + cg.goff) * L + rem;
}

// ------------------------------------------------------------------------------------------
// staging: global (fp32, channels-last) -> LDS image(s)
// ------------------------------------------------------------------------------------------
enum { SEC_NONE = 0, SEC_SQUARE = 1, SEC_SIGN = 2, SEC_MUL = 3 };

struct StageSpec {
  TensorRef src;      // source tensor
  long row0;          // row 0 of the window in the source / mask / mul tensors
  int ctot;           // source row stride
  int coff;           // first source channel
  int cw;             // real channels available from coff (rest of cwp is zero)
  int cwp;            // image channels (multiple of 8)
  int nvalid;         // valid rows
  int pool;           // MaxPool1d(3,1,1) of the source
  TensorRef mask;     // relu mask source (same geometry as src); mask.p == nullptr: none
  TensorRef mul;      // SEC_MUL: multiplier tensor (same geometry as src)
  const uint32_t* sign;  // SEC_SIGN: packed sign words of example 0 of this window
  int sign_stride;       // words per example
  int sign_per_row;      // 1: example = row (dense); 0: example = window (conv)
  int sign_coff;         // bit index of channel 0 of this chunk
  int second;            // SEC_*
};

__device__ __forceinline__ f32x4 load4(const float* p, int nvalid_c, bool vec_ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (vec_ok && nvalid_c >= 4) {
    v = *(const f32x4*)p;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nvalid_c) v[k] = p[k];
  }
  return v;
}

__device__ __forceinline__ f32x4 unpack_bf4(uint2 h) {
  f32x4 v;
  v[0] = bf2f((u16)(h.x & 0xffff));
  v[1] = bf2f((u16)(h.x >> 16));
  v[2] = bf2f((u16)(h.y & 0xffff));
  v[3] = bf2f((u16)(h.y >> 16));
  return v;
}

// 4 consecutive channels at element offset `o` of a typed tensor (fp32 rows or bf16 hi[/lo] planes)
__device__ __forceinline__ f32x4 tload4(const TensorRef& t, long o, int nvalid_c, bool vec_ok) {
  if (t.fmt == TF_F32) return load4((const float*)t.p + o, nvalid_c, vec_ok);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const u16* h = (const u16*)t.p + o;
  const u16* l = t.lo ? (const u16*)t.lo + o : nullptr;
  if (vec_ok && nvalid_c >= 4) {
    v = unpack_bf4(*(const uint2*)h);
    if (l) {
      const f32x4 w = unpack_bf4(*(const uint2*)l);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += w[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nvalid_c) v[k] = bf2f(h[k]) + (l ? bf2f(l[k]) : 0.f);
  }
  return v;
}

__device__ __forceinline__ void tstore4(const TensorRef& t, long o, f32x4 v, int nvalid_c, bool vec_ok) {
  if (t.fmt == TF_F32) {
    float* p = (float*)t.p + o;
    if (vec_ok && nvalid_c >= 4) {
      *(f32x4*)p = v;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nvalid_c) p[k] = v[k];
    }
    return;
  }
  u16 h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = f2bf(v[k]);
    l[k] = f2bf(v[k] - bf2f(h[k]));
  }
  u16* ph = (u16*)t.p + o;
  u16* pl = t.lo ? (u16*)t.lo + o : nullptr;
  if (vec_ok && nvalid_c >= 4) {
    *(uint2*)ph = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    if (pl) *(uint2*)pl = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nvalid_c) {
        ph[k] = h[k];
        if (pl) pl[k] = l[k];
      }
  }
}

// P::BF == false: img0 = primary (fp32), img2 = secondary.  P::BF: img0 = hi, img1 = lo (may be
// nullptr: hi only), img2 = secondary.
template <class P>
__device__ __forceinline__ void stage_window(const StageSpec& sp, typename P::elem* img0, typename P::elem* img1,
                                             typename P::elem* img2, int RS, int lane) {
  const int nc4 = sp.cwp >> 2;
  const int units = IMG_ROWS * nc4;
  const bool vec_ok = ((sp.ctot & 3) == 0) && ((sp.coff & 3) == 0) && ((((uintptr_t)sp.src.p) & 15) == 0);
  for (int u = lane; u < units; u += WAVE) {
    const int rr = u / nc4;
    const int c = (u - rr * nc4) * 4;
    const int row = rr - HALO;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    f32x4 w = {0.f, 0.f, 0.f, 0.f};
    const int nvc = sp.cw - c;  // valid channels in this unit (may be <= 0)
    if (row >= 0 && row < sp.nvalid && nvc > 0) {
      const long o = (sp.row0 + row) * sp.ctot + sp.coff + c;
      v = tload4(sp.src, o, nvc, vec_ok);
      if (sp.pool) {
        if (row > 0) {
          const f32x4 a = tload4(sp.src, o - sp.ctot, nvc, vec_ok);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], a[k]);
        }
        if (row + 1 < sp.nvalid) {
          const f32x4 a = tload4(sp.src, o + sp.ctot, nvc, vec_ok);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], a[k]);
        }
      }
      if (sp.mask.p) {
        const f32x4 m = tload4(sp.mask, o, nvc, vec_ok);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = m[k] > 0.f ? v[k] : 0.f;
      }
      if (sp.second == SEC_SQUARE) {
        if (P::BF) {
          // the contraction sees bf16(x): square what the MFMA will see
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xb = bf2f(f2bf(v[k]));
            w[k] = xb * xb;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = v[k] * v[k];
        }
      } else if (sp.second == SEC_SIGN) {
        const int bit0 = sp.sign_coff + c;
        const uint32_t word = sp.sign[(long)(sp.sign_per_row ? row : 0) * sp.sign_stride + (bit0 >> 5)];
        const uint32_t bits = word >> (bit0 & 31);
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = ((bits >> k) & 1u) ? -v[k] : v[k];
      } else if (sp.second == SEC_MUL) {
        const f32x4 m = tload4(sp.mul, o, nvc, vec_ok);
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = v[k] * m[k];
      }
    }
    const int io = rr * RS + c;
    if constexpr (!P::BF) {
      *(float2*)&img0[io] = make_float2(v[0], v[1]);
      *(float2*)&img0[io + 2] = make_float2(v[2], v[3]);
      if (sp.second != SEC_NONE) {
        *(float2*)&img2[io] = make_float2(w[0], w[1]);
        *(float2*)&img2[io + 2] = make_float2(w[2], w[3]);
      }
    } else {
      u16 h[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) h[k] = f2bf(v[k]);
      *(uint2*)&img0[io] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      if (img1) {
        u16 l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) l[k] = f2bf(v[k] - bf2f(h[k]));
        *(uint2*)&img1[io] =
            make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
      }
      if (sp.second != SEC_NONE) {
        u16 s2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s2[k] = f2bf(w[k]);
        *(uint2*)&img2[io] =
            make_uint2((uint32_t)s2[0] | ((uint32_t)s2[1] << 16), (uint32_t)s2[2] | ((uint32_t)s2[3] << 16));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// gemm_f
// ------------------------------------------------------------------------------------------
// acc_a[nt][mt] += Wa[n][k] * X[row][k]  (+ split terms);  acc_b[nt][mt] += Wb[n][k] * X2[row][k]
//   wa/wlo/wb : weight image rows, row length KP, this branch's first row already applied
//   wk0       : first k index of this chunk inside a weight row
//   x0/x1/x2  : LDS images (primary|hi, lo, secondary)
template <class P, int NT, bool DUAL, bool SPLIT>
__device__ __forceinline__ void gemm_f(f32x4 (&acc_a)[NT][2], f32x4 (&acc_b)[NT][2], int nt_count,
                                       const typename P::elem* __restrict__ wa,
                                       const typename P::elem* __restrict__ wlo,
                                       const typename P::elem* __restrict__ wb, int KP, int wk0, int taps, int pad,
                                       int cwp, const typename P::elem* x0, const typename P::elem* x1,
                                       const typename P::elem* x2, int RS, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!P::BF) {
    int tap = 0, c = 0;
    const int ksteps = (taps * cwp) >> 2;
    for (int ks = 0; ks < ksteps; ++ks) {
      float bx[2], bx2[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int off = (mt * 16 + i + tap - pad + HALO) * RS + c + g;
        bx[mt] = x0[off];
        if constexpr (DUAL) bx2[mt] = x2[off];
      }
      const long wcol = wk0 + ks * 4 + g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt < nt_count) {
          const long wo = (long)(nt * 16 + i) * KP + wcol;
          const float a = wa[wo];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc_a[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx[mt], acc_a[nt][mt], 0, 0, 0);
          if constexpr (DUAL) {
            const float b = wb[wo];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc_b[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, bx2[mt], acc_b[nt][mt], 0, 0, 0);
          }
        }
      }
      c += 4;
      if (c >= cwp) {
        c = 0;
        ++tap;
      }
    }
  } else {
    const int G8 = cwp >> 3;
    const int total = taps * G8;
    const int ksteps = (total + 3) >> 2;
    int c8 = g, tap = 0;
    while (c8 >= G8) {
      c8 -= G8;
      ++tap;
    }
    for (int ks = 0; ks < ksteps; ++ks) {
      const bool valid = tap < taps;
      const int roff = valid ? ((tap - pad + HALO) * RS + c8 * 8) : (HALO * RS);
      bf16x8 bh[2], bl[2], b2[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int off = (mt * 16 + i) * RS + roff;
        bh[mt] = *(const bf16x8*)&x0[off];
        if constexpr (SPLIT) bl[mt] = *(const bf16x8*)&x1[off];
        if constexpr (DUAL) b2[mt] = *(const bf16x8*)&x2[off];
      }
      const long wcol = wk0 + ks * 32 + g * 8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt < nt_count) {
          const long wo = (long)(nt * 16 + i) * KP + wcol;
          const bf16x8 ah = *(const bf16x8*)&wa[wo];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc_a[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[mt], acc_a[nt][mt], 0, 0, 0);
          if constexpr (SPLIT) {
            const bf16x8 al = *(const bf16x8*)&wlo[wo];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              acc_a[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[mt], acc_a[nt][mt], 0, 0, 0);
              acc_a[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[mt], acc_a[nt][mt], 0, 0, 0);
            }
          }
          if constexpr (DUAL) {
            const bf16x8 ab = *(const bf16x8*)&wb[wo];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc_b[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, b2[mt], acc_b[nt][mt], 0, 0, 0);
          }
        }
      }
      c8 += 4;
      while (c8 >= G8) {
        c8 -= G8;
        ++tap;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// gemm_w : one 16(cout) x 16(cin) tile of dW for one tap, reduced over the 32 rows of a window
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const u16* p0, const u16* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

//   dz : image of dZ (IMG_ROWS x RSz), columns = couts;  x : image of X (IMG_ROWS x RSx)
template <class P>
__device__ __forceinline__ f32x4 gemm_w_tile(f32x4 acc, const typename P::elem* dz, int RSz, int n0,
                                             const typename P::elem* x, int RSx, int c0, int tshift, int lane) {
  if constexpr (!P::BF) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < TILE_ROWS / 4; ++ks) {
      const int k = ks * 4 + g;
      const float a = dz[(k + HALO) * RSz + n0 + i];
      const float b = x[(k + tshift + HALO) * RSx + c0 + i];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
  } else {
    // lane 16g + 4q + p supplies row (8g + q [+4]), columns 4p..4p+3 of the 16-column block;
    // lane i of the group receives column i of the 4 rows (ds_read_b64_tr_b16)
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = 8 * g + q;
    const u16* a0 = &dz[(r0 + HALO) * RSz + n0 + 4 * p];
    const u16* b0 = &x[(r0 + tshift + HALO) * RSx + c0 + 4 * p];
    const bf16x8 a = tr_frag(a0, a0 + 4 * RSz);
    const bf16x8 b = tr_frag(b0, b0 + 4 * RSx);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
}
