// dense_fwd2_kernel : variational forward of a dense (Linear) layer, bf16 hi/lo planes (gfx950).
//
// One workgroup = one 32-row window of example rows of one particle = DF_NC compute waves (one 16-cout
// n-tile each, every K chunk) + 2 loader waves.  Small workgroups: two share a CU (70 KB of LDS each), so
// one's DMA waits and barriers overlap with the other's MFMAs.  No K split across waves, hence no
// reduction: a wave walks all 128-channel chunks with its weight fragments double-buffered in registers
// (chunk c+1's fragments are fetched from L2 while chunk c's MFMAs run).
//   loaders : LDS-DMA of the hi / lo planes of a chunk (XOR swizzle on the per-lane source address) and,
//             for Flipout, its sign_in words; DF_SLOTS - 1 chunks ahead; running per-lane addresses.
//   compute : mean path split-bf16 (hi*hi + hi*lo + lo*hi), second contraction single bf16
//             (LRT: x^2 * sigma^2 from a squared plane built per chunk; Flipout: (x s_in) * dW through a
//             sign-mask table), epilogue = bias, LRT noise / Flipout s_out, ReLU, typed store.
#pragma once

enum { DF_NC = 4, DF_NL = 2, DF_NW = DF_NC + DF_NL, DF_SLOTS = 4, DF_CH = 128, DF_ROWS = 32 };

struct DenseFwd2Plan {
  int ntile;    // n-tiles of the branch (<= DF_NC)
  int nchunk;   // 128-channel K chunks
};

template <int EM>
__global__ __launch_bounds__(DF_NW * 64) void dense_fwd2_kernel(const GroupArgs A, const DenseFwd2Plan F) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  constexpr bool FO = (EM == EM_FLIPOUT);
  constexpr int NTHR = DF_NW * 64;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const BranchDesc& br = G.br[0];
  // the layer table lives in device memory: copy what the chunk loop needs into registers ONCE (a reload inside
  // load_w would be a VMEM load whose wait also drains the weight fragments in flight)
  const LayerDesc ly = A.layers[br.layer];
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so XCD x takes the contiguous window range
  // [x * nx, (x + 1) * nx) = at most two particles, whose sampled weights then stay in that XCD's L2
  const int nx = (A.cg.nwin + 7) >> 3;
  const int win = (blockIdx.x & 7) * nx + (blockIdx.x >> 3);
  if (win >= A.cg.nwin) return;
  const Win W = decode_win(G, A.cg, win);
  const int s = W.s;
  constexpr int pbytes = DF_ROWS * DF_CH * 2;   // one plane of a chunk: dense 256-byte rows
  constexpr int slot_bytes = 2 * pbytes + DF_ROWS * 4 * 4;   // hi | lo | sign words [32 rows][4]
  // LDS: slot[DF_SLOTS] | squared plane (LRT) | sign-mask table (Flipout)
  u16* sqi = (u16*)(smem + DF_SLOTS * slot_bytes);
  uint4* lut = (uint4*)(smem + DF_SLOTS * slot_bytes + (LRT ? pbytes : 0));
  {
    const int total = (DF_SLOTS * slot_bytes + (LRT ? pbytes : 0)) >> 2;
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < total; k += NTHR) z[k] = 0u;
  }
  if constexpr (FO) build_sign_lut(lut, tid, NTHR);
  const TensorRef tin = A.t[G.in_t];
  const bool is_loader = wave >= DF_NC;
  const int lw = wave - DF_NC;
  const int nchunk = F.nchunk;
  // K chunks are walked in an order rotated by the window index: the workgroups of a particle would otherwise fetch
  // the same weight lines from L2 in lockstep.  Step c works on chunk kc(c).
  const int c_rot = win % nchunk;
  auto kc = [&](int c) { const int k = c + c_rot; return k >= nchunk ? k - nchunk : k; };
  BNN_STAMP_DECL(A);

  if (is_loader) {
    // =========================== loader waves ===========================
    // loader lw streams plane lw (0 hi, 1 lo): 8 instructions per chunk (32 rows x 16 chunks of 16 B), and half of
    // the sign words.  Rows beyond the last valid one re-read that row (every instruction keeps active lanes;
    // their results are never stored).
    const char* a_src[8];
    uint32_t on_last = 0;
    const int cw8_last = (br.cin_p - (nchunk - 1) * DF_CH) >> 3;   // valid 16-byte chunks per row of the last chunk
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = i * 64 + lane;
      const int row = q >> 4, p = q & 15;
      const int c8 = p ^ (row & 15);
      const int srow = min(row, W.nvalid - 1);
      a_src[i] = (const char*)((const u16*)(lw ? tin.lo : tin.p) + (long)(W.in_row0 + srow) * tin.ctot + c8 * 8 + c_rot * DF_CH);
      if (c8 < cw8_last) on_last |= 1u << i;
    }
    const uint32_t* sg_src = nullptr;
    int sg_n = 0;   // chunks for which this lane's sign word exists
    if constexpr (FO) {
      const int q = lw * 64 + lane;
      const int row = q >> 2, k = q & 3;
      const int srow = min(row, W.nvalid - 1);
      sg_src = A.nz.sign_in + ly.sign_in_off * A.nz.examples + (long)(W.ex0 + srow) * ly.sign_in_words + k;
      sg_n = (ly.sign_in_words - k + 3) >> 2;
    }
    asm volatile("" : "+v"(sg_src), "+v"(sg_n));   // every ordinary load consumed before the DMA sequence
    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int c, int slot) {   // steps must be issued in order 0, 1, 2, ...
      const uint32_t sbase = lds0 + (uint32_t)(slot * slot_bytes + lw * pbytes);
      const int k = kc(c);
      const bool last = k == nchunk - 1;
      const int adv = last ? -(nchunk - 1) * DF_CH * 2 : DF_CH * 2;   // wrap to chunk 0 after the last one
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t on = on_last;
        asm volatile("" : "+v"(on));
        if (!last || ((on >> i) & 1u)) dma16(a_src[i], sbase + (uint32_t)(i * 1024));
        a_src[i] += adv;
      }
      if constexpr (FO) {
        if (k < sg_n) dma4(sg_src + 4 * k, lds0 + (uint32_t)(slot * slot_bytes + 2 * pbytes + lw * 256));
      }
    };
    constexpr int n_issue = 8 + (FO ? 1 : 0);
    __syncthreads();   // zero fill + table visible
    constexpr int ahead = DF_SLOTS - 1;
    for (int j = 0; j < ahead; ++j)
      if (j < nchunk) issue(j, j);
    int slot = 0;
    for (int c = 0; c < nchunk; ++c) {
      stamp(c, 0);
      const int fly = min(ahead - 1, nchunk - 1 - c) * n_issue;   // chunks c+1 .. c+ahead-1 may stay in flight
      BNN_WAIT_VMCNT_WIDE(fly);
      stamp(c, 1);
      lds_barrier();   // B1: chunk c landed; chunk c-1 consumed
      stamp(c, 2);
      if constexpr (LRT) lds_barrier();   // B2 (squared plane built by the compute waves)
      if (c + ahead < nchunk) issue(c + ahead, slot == 0 ? DF_SLOTS - 1 : slot - 1);   // slot of chunk c-1
      stamp(c, 3);
      slot = slot + 1 == DF_SLOTS ? 0 : slot + 1;
    }
    return;
  }

  // =========================== compute waves ===========================
  const int i16 = lane & 15, g4 = lane >> 4;
  const int nt = wave;
  const bool has_job = nt < F.ntile;
  const long sa = A.ws.slot_stride_a * s, sb = A.ws.slot_stride_b * s;
  // weight fragments of a chunk (4 k-steps), double-buffered
  bf16x8 w_hi[2][4], w_lo[2][4], w_b[2][4];
  // this lane's weight-image row (n-tile row i16, k group g4); chunk c / k-step ks add c*128 + ks*32 elements.
  // Rows of waves without a job point at row 0 (loaded, never used).
  const long w_row = (long)(br.n_off + (has_job ? nt : 0) * 16 + i16) * ly.KP + ly.w_off + g4 * 8;
  const u16* p_hi = (const u16*)A.ws.a_hi + sa + w_row;
  const u16* p_lo = (const u16*)A.ws.a_lo + sa + w_row;
  const u16* p_b = (const u16*)A.ws.b + sb + w_row;
  const int cin_p = br.cin_p;
  asm volatile("" : "+v"(p_hi), "+v"(p_lo), "+v"(p_b));   // table loads consumed here
  auto load_w = [&](int c, bf16x8* whi, bf16x8* wlo, bf16x8* wb) {
    const int nks = min(DF_CH, cin_p - c * DF_CH) >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // k-steps past the end of K read the image's zero padding or the next row: their B operand rows are
      // never accumulated (the MFMA loop stops at nks)
      const int o = c * DF_CH + (ks < nks ? ks : 0) * 32;
      whi[ks] = *(const bf16x8*)(p_hi + o);
      wlo[ks] = *(const bf16x8*)(p_lo + o);
      if constexpr (DUAL) wb[ks] = *(const bf16x8*)(p_b + o);
    }
  };
  // window-invariant LDS element offsets of this lane's B fragments inside a plane: [ks][mt]
  int k_o[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int rr = mt * 16 + i16;
      k_o[ks][mt] = rr * DF_CH + (((ks * 4 + g4) ^ (rr & 15)) * 8);
    }
  f32x4 acc_a[2], acc_b[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) acc_a[mt] = acc_b[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  load_w(c_rot, w_hi[0], w_lo[0], w_b[0]);

  __syncthreads();   // zero fill + table visible
  // Chunk c with the fragment set (whi, wlo, wb); the next chunk's fragments go to (nhi, nlo, nb).  The next
  // loads are issued AFTER the first use of the current set: hipcc waits with vmcnt(0) for loads that crossed
  // the loop back-edge, so any newer load in flight at that point would be waited for as well.
  auto chunk_step = [&](int c, int slot, bf16x8* whi, bf16x8* wlo, bf16x8* wb, bf16x8* nhi, bf16x8* nlo, bf16x8* nb) {
    stamp(c, 0);
    const char* sl = smem + slot * slot_bytes;
    const u16* r_hi = (const u16*)sl;
    const u16* r_lo = (const u16*)(sl + pbytes);
    const uint32_t* sg = (const uint32_t*)(sl + 2 * pbytes);
    lds_barrier();   // B1
    stamp(c, 2);
    if constexpr (LRT) {
      // squares of the bf16 hi plane (what the variance contraction sees)
      for (int U = tid; U < DF_ROWS * 16; U += DF_NC * 64) {
        const uint4 h = *(const uint4*)&r_hi[U * 8];
        const uint32_t hh[4] = {h.x, h.y, h.z, h.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = bf2f((u16)(hh[e] & 0xffff)), b = bf2f((u16)(hh[e] >> 16));
          o[e] = cvt_pk(a * a, b * b);
        }
        *(uint4*)&sqi[U * 8] = make_uint4(o[0], o[1], o[2], o[3]);
      }
      lds_barrier();   // B2
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {   // first use of the current set (the compiler's wait lands here)
      asm volatile("" : "+v"(whi[ks]), "+v"(wlo[ks]));
      if constexpr (DUAL) asm volatile("" : "+v"(wb[ks]));
    }
    if (c + 1 < nchunk) load_w(kc(c + 1), nhi, nlo, nb);   // lands during this chunk's MFMAs
    if (has_job) {
      const int nks = min(DF_CH, cin_p - kc(c) * DF_CH) >> 5;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= nks) break;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int o = k_o[ks][mt];
          const bf16x8 bh = *(const bf16x8*)&r_hi[o];
          const bf16x8 bl = *(const bf16x8*)&r_lo[o];
          acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[ks], bh, acc_a[mt], 0, 0, 0);
          acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[ks], bl, acc_a[mt], 0, 0, 0);
          acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[ks], bh, acc_a[mt], 0, 0, 0);
          if constexpr (LRT) {
            const bf16x8 b2 = *(const bf16x8*)&sqi[o];
            acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], b2, acc_b[mt], 0, 0, 0);
          } else if constexpr (FO) {
            // sign word of (row, k-step): byte g4 = this lane's 8 channels
            const uint32_t word = sg[(mt * 16 + i16) * 4 + ks];
            const u32x4 fm = __builtin_bit_cast(u32x4, lut[(word >> (8 * g4)) & 0xffu]);
            const u32x4 xb = __builtin_bit_cast(u32x4, bh) ^ fm;
            acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], __builtin_bit_cast(bf16x8, xb), acc_b[mt], 0, 0, 0);
          }
        }
      }
    }
    stamp(c, 3);
  };
  int slot = 0;
  for (int c = 0; c < nchunk; c += 2) {
    chunk_step(c, slot, w_hi[0], w_lo[0], w_b[0], w_hi[1], w_lo[1], w_b[1]);
    slot = slot + 1 == DF_SLOTS ? 0 : slot + 1;
    if (c + 1 < nchunk) {
      chunk_step(c + 1, slot, w_hi[1], w_lo[1], w_b[1], w_hi[0], w_lo[0], w_b[0]);
      slot = slot + 1 == DF_SLOTS ? 0 : slot + 1;
    }
  }
  // ---------------- epilogue ----------------
  if (has_job) {
    const int chb = nt * 16 + 4 * g4;
    const int nv = br.cout - chb;
    if (nv > 0) {
      const TensorRef tout = A.t[br.out_t];
      const float* ba = A.ws.bias_a + (long)A.ws.bias_stride_a * s + ly.bias_off + br.n_off + chb;
      const float* bb = A.ws.bias_b + ly.bias_off + br.n_off + chb;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        if (row >= W.nvalid) continue;
        const int R = W.out_row0 + row;
        f32x4 v = acc_a[mt];
        f32x4 qv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LRT) {
          f32x4 eps;
          const int lch = br.n_off + chb;
          if (A.nz.use_philox_lrt) {
            const long Rg = global_row(A.cg, 1, R);
            const uint64_t idx = (uint64_t)Rg * (uint64_t)(ly.cout_p16 >> 2) + (uint64_t)(lch >> 2);
            eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)br.layer << 8), A.nz.step, A.nz.seed);
          } else {
            const float* e = A.nz.lrt_eps[br.layer] + (long)R * ly.cout + lch;
#pragma unroll
            for (int r = 0; r < 4; ++r) eps[r] = (r < nv) ? e[r] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nv) {
              const float loc = v[r] + ba[r];
              float var = acc_b[mt][r] + bb[r];
              if (var < 0.f) var = 1e-6f;
              const float sd = sqrtf(var);
              v[r] = loc + sd * eps[r];
              qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
            }
        } else if constexpr (FO) {
          const int bit0 = br.n_off + chb;
          const uint32_t word = A.nz.sign_out[ly.sign_out_off * A.nz.examples + (long)(W.ex0 + row) * ly.sign_out_words + (bit0 >> 5)];
          const uint32_t bits = word >> (bit0 & 31);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nv) {
              const float pb = acc_b[mt][r];
              v[r] = v[r] + ba[r] + (((bits >> r) & 1u) ? -pb : pb);
            }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nv) v[r] += ba[r];
        }
        if (br.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const long oo = (long)R * tout.ctot + br.out_off + chb;
        const bool vec = ((tout.ctot & 3) == 0) && ((br.out_off & 3) == 0);
        tstore4(tout, oo, v, nv, vec);
        if constexpr (LRT) {
          const TensorRef tq = A.t[br.q_t];
          tstore4(tq, (long)R * tq.ctot + br.out_off + chb, qv, nv, vec);
        }
      }
    }
  }
}
