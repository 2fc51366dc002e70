// conv_dx2_kernel : d loss / d input of a conv group, pooled branches included (bf16 planes, gfx950).
//
// One workgroup = NC compute waves + 2 loader waves (DMA instructions dealt alternately); it walks the windows
// split, split + nsplit, ... of one particle and produces NC of the 16-channel tiles of the input tensor
// (workgroup KIND = which NC tiles; every kind loads the whole dY / Y slices).  NC = 8 covers a 128-channel
// tensor with ONE kind (each slice loaded and masked once; measured 1.7x faster than two 6-wave kinds that
// share a CU but load everything twice); NC = 4 is kept for narrow tensors.
//   loader  : LDS-DMA (global_load_lds_dwordx4) of the dY / Y(hi) [/ q] channel slices of every branch
//             (one dense XOR-swizzled sub-image per branch and plane type), the arg-max plane of the
//             pooled input and, for LRT, the X hi plane, `nslots - 1` windows ahead.  The DMA table is
//             resolved on the host (ConvDx2Plan::inst); every instruction keeps its per-lane source
//             address in a VGPR pair that is advanced by a constant after each issue, so the window
//             loop issues ~10 instructions per DMA: no load, no division, no address arithmetic.
//   all     : dz = dY * [Y > 0] IN PLACE in the dY plane (LRT: q plane <- dz * q; Flipout: Y plane <-
//             dz * s_out, the second contraction's operand).
//   compute : wave ct owns the 16-channel tile ct of the input tensor; K runs over the k-steps of all
//             branches (transposed + flipped weight fragments in registers, B fragments from LDS at
//             window-invariant precomputed offsets).
//               direct : dX  = Wa^T dz + { LRT: 2 X (Wb^T dz2) | Flipout: (s_in Wb)^T (dz s_out) }
//               pooled : dP  = the same w.r.t. the max-pooled input; scattered through the arg-max
//                        codes written by the forward (0: row-1, 1: row, 2: row+1) with lane permutes:
//                        no pooled-gradient tensor, no separate scatter kernel.
//             Flipout folds s_in into the rows of the weight fragment and s_out into the dz fragment
//             (sign-bit XOR), so every branch accumulates into one accumulator.
#pragma once

enum { DX2_KS = 6, DX2_MAXI = 48, DX2_NL = 2, DX2_LI = DX2_MAXI / DX2_NL, DX2_MU = 2 };   // DX2_KS >= KD + KP of every instantiation

struct Dx2Inst {          // one LDS-DMA instruction of a window (64 lanes x 16 bytes), host-resolved
  const void* base;       // plane base + first channel of the stream
  uint32_t wstride;       // bytes between windows
  uint32_t rstride;       // bytes between rows
  uint32_t dst;           // LDS byte offset inside a slot of chunk q0
  uint32_t geom;          // q0 | lg << 16 (log2 chunks per row) | sh << 24 (swizzle: c ^= (r >> sh) & (cb - 1))
  uint32_t qn;            // chunks of the stream (rows * chunks per row)
  uint32_t pad_;
};

struct ConvDx2Plan {
  int nsplit, nks, ntile, nslots, nkinds;
  int zbase[BNN_MAX_BRANCH];      // element offset of branch b's sub-image inside a plane (-1: no dX wanted)
  int zelems;                     // elements of one plane (all sub-images incl. halo rows)
  int has_pool, ninst, slot_bytes;
  int o_x, o_am;                  // byte offsets of the X hi plane / arg-max plane inside a slot
  int dx_t;
  signed char ks_b[DX2_KS], ks_i[DX2_KS];   // k-step -> (branch, 32-wide K slice); [0, KD) direct, [KD, KD+KP) pooled; b < 0: unused
  Dx2Inst inst[DX2_MAXI];
};

// value of the lane `rotation` positions away inside the 16-lane DPP row (one VALU move, no LDS crossbar)
template <int CTRL>
__device__ __forceinline__ float rot16(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

#define BNN_WAIT_VMCNT_WIDE(N)                                          \
  do {                                                                  \
    switch (N) {                                                        \
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   \
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;   \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;   \
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;   \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;   \
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;   \
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;   \
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;   \
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;   \
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;   \
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break; \
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break; \
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break; \
      case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break; \
      case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break; \
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break; \
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break; \
      case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break; \
      case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break; \
      case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break; \
      case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break; \
      case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break; \
      case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break; \
      case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break; \
      case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break; \
      case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break; \
      case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break; \
      case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break; \
      case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break; \
      case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break; \
      case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break; \
      case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break; \
      case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break; \
      case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break; \
      case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break; \
      case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break; \
      case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break; \
      case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break; \
      case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break; \
      case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break; \
      case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break; \
      case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break; \
      case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break; \
      case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break; \
      case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break; \
      case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break; \
      case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break; \
      case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break; \
      case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break; \
      case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break; \
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  \
    }                                                                   \
  } while (0)

// NC compute waves (tiles per workgroup kind): 4 -> two kinds, two workgroups per CU; 8 -> one kind, one workgroup
// per CU that loads and masks every dY / Y slice once
template <int EM, int KD, int KP, int NC>
__global__ __launch_bounds__((NC + DX2_NL) * 64) void conv_dx2_kernel(const GroupArgs A, const ConvDx2Plan D) {
  constexpr int DX2_NC = NC, DX2_NW = NC + DX2_NL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  constexpr bool FO = (EM == EM_FLIPOUT);
  constexpr int NTHR = DX2_NW * 64;
  constexpr int NKS = KD + KP;
  static_assert(NKS <= DX2_KS, "k-step table too small");
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const int kind = blockIdx.x % D.nkinds, bid = blockIdx.x / D.nkinds;
  const int s = bid / D.nsplit, split = bid - s * D.nsplit;
  const int L = G.L, B = A.cg.B;
  const int zbytes = D.zelems * 2;
  const int xw = G.in_cin_p, x8n = xw >> 3;
  const int nslots = D.nslots, slot_bytes = D.slot_bytes;
  // LDS: slot[nslots] { dY | Y | [q] | [X hi] | [arg-max] } | sign words [nslots][64] | sign-mask table
  uint32_t* sgn = (uint32_t*)(smem + nslots * slot_bytes);
  uint4* lut = (uint4*)(smem + nslots * slot_bytes + nslots * 64 * 4);
  {
    const int total = (nslots * slot_bytes + nslots * 64 * 4) >> 2;
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < total; k += NTHR) z[k] = 0u;
  }
  if constexpr (FO) build_sign_lut(lut, tid, NTHR);
  const bool is_loader = wave >= DX2_NC;
  const int lw = wave - DX2_NC;
  const int nwin = (B - split + D.nsplit - 1) / D.nsplit;
  auto win_of = [&](int k) { return split + k * D.nsplit; };

  // ---- mask pass plan: <= DX2_MU 16-byte chunks per thread (same position in the dY / Y / q planes) ----
  int m_o[DX2_MU];      // element offset inside a plane (-1: none)
  int m_f[DX2_MU];      // relu | flipout: sign_out byte of the chunk's 8 couts: word index << 8 | shift << 16
#pragma unroll
  for (int u = 0; u < DX2_MU; ++u) {
    m_o[u] = -1;
    m_f[u] = 0;
    int U = tid + u * NTHR;
    for (int b = 0; b < G.n_branch; ++b) {
      if (D.zbase[b] < 0) continue;
      const int cb8 = G.br[b].cout >> 3;
      const int n = L * cb8;
      if (U >= 0 && U < n) {
        m_o[u] = D.zbase[b] + HALO * G.br[b].cout + U * 8;   // rows HALO .. HALO+L-1 of a sub-image are contiguous
        const int row = U / cb8, p = U - row * cb8;
        const int n0 = G.br[b].n_off + swz(p, row + HALO, cb8) * 8;   // channel chunk stored at position p
        m_f[u] = (G.br[b].relu ? 1 : 0) | ((b * 8 + 4 + (n0 >> 5)) << 8) | ((n0 & 31) << 16);
        U = -1;
      } else if (U >= n) {
        U -= n;
      }
    }
  }
  auto mask_pass = [&](char* slot, const uint32_t* sg) {
    u16* p_dy = (u16*)slot;
    u16* p_y = (u16*)(slot + zbytes);
    u16* p_q = (u16*)(slot + 2 * zbytes);
    uint4 g[DX2_MU], y[DX2_MU], qq[DX2_MU], fm[DX2_MU];
#pragma unroll
    for (int u = 0; u < DX2_MU; ++u) {   // all reads first: one LDS latency for the whole pass
      g[u] = y[u] = qq[u] = fm[u] = make_uint4(0, 0, 0, 0);
      if (m_o[u] >= 0) {
        g[u] = *(const uint4*)&p_dy[m_o[u]];
        if (m_f[u] & 1) y[u] = *(const uint4*)&p_y[m_o[u]];
        if constexpr (LRT) qq[u] = *(const uint4*)&p_q[m_o[u]];
        if constexpr (FO) fm[u] = lut[(sg[(m_f[u] >> 8) & 0xff] >> (m_f[u] >> 16)) & 0xffu];
      }
    }
#pragma unroll
    for (int u = 0; u < DX2_MU; ++u) {
      if (m_o[u] < 0) continue;
      uint4 gg = g[u];
      if (m_f[u] & 1) {
        // Y is a ReLU output (never negative): Y > 0  <=>  magnitude bits non-zero; packed 16-bit min / mul
        auto msk = [](uint32_t yy) {
          typedef unsigned short us2 __attribute__((ext_vector_type(2)));
          const us2 v = __builtin_bit_cast(us2, yy & 0x7fff7fffu);
          const us2 m = __builtin_elementwise_min(v, us2{1, 1}) * us2{0xffff, 0xffff};
          return __builtin_bit_cast(uint32_t, m);
        };
        gg.x &= msk(y[u].x); gg.y &= msk(y[u].y); gg.z &= msk(y[u].z); gg.w &= msk(y[u].w);
        *(uint4*)&p_dy[m_o[u]] = gg;
      }
      if constexpr (FO) {
        *(uint4*)&p_y[m_o[u]] = make_uint4(gg.x ^ fm[u].x, gg.y ^ fm[u].y, gg.z ^ fm[u].z, gg.w ^ fm[u].w);
      }
      if constexpr (LRT) {
        const uint32_t ga[4] = {gg.x, gg.y, gg.z, gg.w}, qv[4] = {qq[u].x, qq[u].y, qq[u].z, qq[u].w};
        uint32_t out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = bf2f((u16)(ga[e] & 0xffff)) * bf2f((u16)(qv[e] & 0xffff));
          const float a1 = bf2f((u16)(ga[e] >> 16)) * bf2f((u16)(qv[e] >> 16));
          out[e] = cvt_pk(a0, a1);
        }
        *(uint4*)&p_q[m_o[u]] = make_uint4(out[0], out[1], out[2], out[3]);
      }
    }
  };
  BNN_STAMP_DECL(A);

  if (is_loader) {
    // =========================== loader wave ===========================
    // lane i keeps the wave-uniform fields of DMA instruction i; a_src[i] = this lane's source address of
    // instruction i for the next window to issue (null: lane inactive), advanced by t_step after each issue
    const int my_n = (D.ninst - lw + DX2_NL - 1) / DX2_NL;   // this loader's instructions: lw, lw + NL, ...
    const int li = lane < my_n ? lw + lane * DX2_NL : 0;
    const Dx2Inst mine = D.inst[li];
    uint32_t t_dst = mine.dst, t_step = mine.wstride * (uint32_t)D.nsplit;
    const char* a_src[DX2_LI];
    uint32_t a_on = 0;   // bit i: this lane takes part in this loader's i-th instruction
    const long w0 = (long)s * B + split;
#pragma unroll
    for (int i = 0; i < DX2_LI; ++i) {
      a_src[i] = nullptr;
      if (i < my_n) {
        const Dx2Inst I = D.inst[lw + i * DX2_NL];
        const uint32_t lg = (I.geom >> 16) & 0xffu, sh = I.geom >> 24, cm = (1u << lg) - 1u;
        const uint32_t q = (I.geom & 0xffffu) + (uint32_t)lane;
        const uint32_t row = q >> lg, p = q & cm;
        const uint32_t c = p ^ (((row + HALO) >> sh) & cm);
        a_src[i] = (const char*)I.base + (uint64_t)w0 * I.wstride + (uint64_t)(row * I.rstride + c * 16u);
        if (q < I.qn) a_on |= 1u << i;
      }
    }
    // flipout sign words of a window: [branch][8] = 4 words sign_in + 2 words sign_out
    const uint32_t* sg_src = nullptr;
    long sg_step = 0;
    if (FO && lw == 0 && lane < 8 * G.n_branch) {
      const int b = lane >> 3, k = lane & 7;
      const LayerDesc& ly = A.layers[G.br[b].layer];
      if (k < 4 && k < ly.sign_in_words) {
        sg_src = A.nz.sign_in + ly.sign_in_off * A.nz.examples + k + w0 * ly.sign_in_words;
        sg_step = (long)ly.sign_in_words * D.nsplit;
      } else if (k >= 4 && k - 4 < ly.sign_out_words && k < 6) {
        sg_src = A.nz.sign_out + ly.sign_out_off * A.nz.examples + (k - 4) + w0 * ly.sign_out_words;
        sg_step = (long)ly.sign_out_words * D.nsplit;
      }
    }
    // every ordinary load of this wave is consumed HERE: the compiler puts its s_waitcnt vmcnt(0) at the first
    // use of a loaded value, and a wait inside the issue sequence would serialise the (unmodelled) DMAs
    asm volatile("" : "+v"(t_dst), "+v"(t_step));
    asm volatile("" : "+v"(sg_src), "+v"(sg_step));
    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int slot) {   // the next window of this workgroup -> slot
      const uint32_t sbase = lds0 + (uint32_t)(slot * slot_bytes);
#pragma unroll
      for (int i = 0; i < DX2_LI; ++i) {
        if (i >= my_n) break;
        const uint32_t dst = __builtin_amdgcn_readlane(t_dst, i), step = __builtin_amdgcn_readlane(t_step, i);
        uint32_t on = a_on;
        asm volatile("" : "+v"(on));   // keeps the lane test here: hoisted, its 24 exec masks would spill
        if ((on >> i) & 1u) dma16(a_src[i], sbase + dst);
        a_src[i] += step;
      }
      if constexpr (FO) {
        if (lw == 0) {
          if (sg_src) dma4(sg_src, lds0 + (uint32_t)(nslots * slot_bytes + slot * 256));
          sg_src += sg_step;
        }
      }
    };
    const int n_issue = my_n + ((FO && lw == 0) ? 1 : 0);   // every instruction has active lanes
    __syncthreads();   // zero fill + table visible
    const int ahead = nslots - 1;
    for (int j = 0; j < ahead; ++j)
      if (j < nwin) issue(j);
    int slot = 0;
    for (int k = 0; k < nwin; ++k) {
      stamp(k, 0);
      // windows k+1 .. k+ahead-1 may stay in flight
      const int fly = min(ahead - 1, nwin - 1 - k) * n_issue;
      BNN_WAIT_VMCNT_WIDE(fly);
      stamp(k, 1);
      lds_barrier();   // B1: planes of window k landed; every wave finished window k-1
      stamp(k, 2);
      mask_pass(smem + slot * slot_bytes, sgn + slot * 64);
      stamp(k, 3);
      lds_barrier();   // B2: dz visible
      stamp(k, 4);
      // slot of window k + ahead = slot of window k - 1: its readers passed B1
      if (k + ahead < nwin) issue(slot == 0 ? nslots - 1 : slot - 1);
      stamp(k, 5);
      slot = slot + 1 == nslots ? 0 : slot + 1;
    }
    return;
  }

  // =========================== compute waves ===========================
  const int i16 = lane & 15, g4 = lane >> 4;
  const int ct = kind * DX2_NC + wave;   // 16-channel tile of the input tensor
  const bool has_job = ct < D.ntile;
  const TensorRef tdx = A.t[D.dx_t];
  bf16x8 w_a[NKS], w_b[NKS];
  int k_o[NKS];         // LDS element offset of this lane's dz fragment inside the dY plane, m-tile 0
  int k_d16[NKS];       // wave-uniform: element distance of m-tile 1 (16 image rows; the swizzle has period 16)
  int k_si[NKS];        // flipout: sign_in bit of the lane's input channel: word index | shift << 8
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    w_a[ks] = w_b[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    k_o[ks] = 0;
    k_d16[ks] = 0;
    k_si[ks] = 0;
    const int b = D.ks_b[ks];
    if (has_job && b >= 0) {
      const BranchDesc& br = G.br[b];
      const LayerDesc& ly = A.layers[br.layer];
      const int c0 = ct * 16 - br.in_off;                 // first layer-input channel of this tile
      if (c0 >= 0 && c0 < br.cin_p) {
        const int G8 = ly.cout_p8 >> 3;                   // K groups per tap in the transposed image
        const int gg = D.ks_i[ks] * 4 + g4;
        const int tap = gg / G8, c8 = gg - tap * G8;
        const bool valid = tap < ly.taps && c8 * 8 < br.cout;
        const long wo = (long)(c0 + i16) * ly.KPt + (long)D.ks_i[ks] * 32 + g4 * 8;
        const long sa = A.ws.slott_stride_a * s, sb = A.ws.slott_stride_b * s;
        if (valid) {
          w_a[ks] = *(const bf16x8*)((const u16*)A.ws.at + sa + ly.wt_off + wo);
          if constexpr (DUAL) w_b[ks] = *(const bf16x8*)((const u16*)A.ws.bt + sb + ly.wt_off + wo);
        }
        const int cb8 = br.cout >> 3;
        const int rr = (valid ? (tap - ly.pad + HALO) : HALO) + i16;
        k_o[ks] = D.zbase[b] + rr * (cb8 * 8) + swz(valid ? c8 : 0, rr, cb8) * 8;
        k_d16[ks] = 16 * cb8 * 8;
        const int ci = c0 + i16;
        k_si[ks] = (b * 8 + (ci >> 5)) | ((ci & 31) << 8);
      }
    }
  }
  const int och = ct * 16 + 4 * g4;   // channel of the target tensor held by this lane
  const int zel = D.zelems;

  __syncthreads();   // zero fill + table visible
  int slot = 0;
  for (int k = 0; k < nwin; ++k) {
    stamp(k, 0);
    char* sl = smem + slot * slot_bytes;
    const u16* dzi = (const u16*)sl;
    const u16* r_x = (const u16*)(sl + D.o_x);
    const unsigned char* r_am = (const unsigned char*)(sl + D.o_am);
    const uint32_t* sg = sgn + slot * 64;
    stamp(k, 1);
    lds_barrier();   // B1
    stamp(k, 2);
    mask_pass(sl, sg);
    stamp(k, 3);
    lds_barrier();   // B2
    stamp(k, 4);
    if (has_job) {
      // ---------------- MFMA: branch-free, every LDS read independent of the others ----------------
      f32x4 acc_a[2], acc_b[2], acc_pa[2], acc_pb[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc_a[mt] = acc_b[mt] = acc_pa[mt] = acc_pb[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        bf16x8 wb = w_b[ks];
        if constexpr (FO) {
          // s_in of this lane's input channel folded into the weight row
          const uint32_t sb = ((sg[k_si[ks] & 0xff] >> (k_si[ks] >> 8)) & 1u) ? 0x80008000u : 0u;
          const u32x4 wx = __builtin_bit_cast(u32x4, wb) ^ u32x4{sb, sb, sb, sb};
          wb = __builtin_bit_cast(bf16x8, wx);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int o = k_o[ks] + mt * k_d16[ks];
          const bf16x8 bz = *(const bf16x8*)&dzi[o];
          f32x4& ta = ks < KD ? acc_a[mt] : acc_pa[mt];
          ta = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_a[ks], bz, ta, 0, 0, 0);
          if constexpr (LRT) {
            const bf16x8 b2 = *(const bf16x8*)&dzi[o + 2 * zel];   // q plane <- dz * q
            f32x4& tb = ks < KD ? acc_b[mt] : acc_pb[mt];
            tb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, b2, tb, 0, 0, 0);
          } else if constexpr (FO) {
            const bf16x8 bs = *(const bf16x8*)&dzi[o + zel];       // Y plane <- dz * s_out
            ta = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, bs, ta, 0, 0, 0);
          }
        }
      }
      stamp(k, 5);
      // ---------------- epilogue ----------------
      const long w = (long)s * B + win_of(k);
      auto xat = [&](int r) {   // X hi, 4 channels of image row r (LRT only)
        const int ri = r + HALO;
        return unpack_bf4(*(const uint2*)&r_x[ri * xw + swz(och >> 3, ri, x8n) * 8 + (och & 7)]);
      };
      f32x4 v[2], up[2], dn[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        v[mt] = acc_a[mt];
        up[mt] = dn[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (LRT) {
          if (row < L) {
            const f32x4 xv = xat(row);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[mt][r] += 2.f * xv[r] * acc_b[mt][r];
          }
        }
        if constexpr (KP > 0) {
          // this row's pooled gradient goes to row + code - 1
          f32x4 dp = acc_pa[mt];
          uint32_t code = 0x01010101u;
          if (row < L) {
            const int ri = row + HALO;
            code = *(const uint32_t*)&r_am[ri * xw + (((och >> 4) ^ (ri & 7)) << 4) + (och & 15)];
          }
          if constexpr (LRT) {
            if (row < L) {
              const f32x4 x0 = xat(row - 1), x1 = xat(row), x2 = xat(row + 1);   // halo rows are zero and never selected
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const uint32_t cd = (code >> (8 * r)) & 3u;
                const float xp = cd == 0u ? x0[r] : (cd == 1u ? x1[r] : x2[r]);
                dp[r] += 2.f * xp * acc_pb[mt][r];
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t cd = (code >> (8 * r)) & 3u;
            const float g = row < L ? dp[r] : 0.f;
            v[mt][r] += cd == 1u ? g : 0.f;
            up[mt][r] = cd == 0u ? g : 0.f;
            dn[mt][r] = cd == 2u ? g : 0.f;
          }
        }
      }
      if constexpr (KP > 0) {
        // row r receives `dn` of row r-1 and `up` of row r+1: rows live in the 16 lanes of a lane group
        // (i16) and in the two m-tiles
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // DPP row rotations (rows = the 16 lanes of a lane group): ror:1 reads lane i16-1, ror:15 lane i16+1 (mod 16)
          const float a0 = rot16<0x121>(dn[0][r]), a1 = rot16<0x121>(dn[1][r]);
          const float b0 = rot16<0x12F>(up[0][r]), b1 = rot16<0x12F>(up[1][r]);
          v[0][r] += (i16 == 0 ? 0.f : a0) + (i16 == 15 ? b1 : b0);
          v[1][r] += (i16 == 0 ? a0 : a1) + (i16 == 15 ? 0.f : b1);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        if (row < L) *(uint2*)((u16*)tdx.p + (w * L + row) * tdx.ctot + och) = pack_bf4(v[mt]);
      }
    }
    stamp(k, 6);
    slot = slot + 1 == nslots ? 0 : slot + 1;
  }
}
