// K-split kernels of the wide dense layer (Flatten -> Linear(2400, 64) of the Inception net), gfx950.
//
// The layer has few output channels and few rows per particle (B = 1000) but a long contraction: a workgroup that
// owns a 32-row window and walks all of K re-reads the particle's weight images from L2 for every window (48 KB per
// 128-channel chunk against 16 KB of activations), and the per-CU L2 rate is what bounds it.  Here a workgroup owns a
// (particle, K chunk) pair instead: the chunk's weight fragments are loaded ONCE into registers and the particle's
// windows stream through LDS.
//
//   dense_ks_fwd_kernel : (particle, 256-channel chunk, window split).  Partial pre-activations (mean path split-bf16,
//                         Flipout perturbation with both sign vectors applied) go to a per-chunk fp32 slab
//                         [chunk][S*B][64] with plain stores;
//   dense_ks_fin_kernel : sums the slabs in chunk order, adds the bias, ReLU, writes the hi / lo planes.
//   dense_ks_bwd_kernel : (particle, 128-channel chunk, window split): dX of the chunk's input channels (weights of the
//                         transposed images in registers) AND the chunk's dW tiles (in registers across the windows) from
//                         one staging of dz = dY [Y > 0] and of the X chunk.
// Plain and Flipout estimators (LRT keeps the row-stationary kernels: its variance needs the full sum before the noise).
#pragma once
// diagnostics builds only (tests/probes/ablate_gpu.sh): timing with parts of the kernels removed; results are wrong.
// The product library is built with DK_ABL == 0.
#ifndef DK_ABL
#define DK_ABL 0
#endif

// forward: a step = 16 rows (half a window); ring of 4 step slots, 3 steps (51 KB per workgroup, two workgroups per
// CU) in flight: HBM latency under load is 2-3 us, a step's MFMAs 0.2 us
enum { DK_KS = 8, DK_CH = DK_KS * 32, DK_ROWS = 16, DK_NC = 4, DK_NL = 2, DK_NW = DK_NC + DK_NL, DK_SLOTS = 4 };
enum { DK_PLANE = DK_ROWS * DK_CH * 2,                         // one plane of a step: 512-byte rows
       DK_SLOT = 2 * DK_PLANE + DK_ROWS * DK_KS * 4 + DK_ROWS * 2 * 4,   // hi | lo | sign_in [16][8] | sign_out [16][2]
       DK_FWD_LDS = DK_SLOTS * DK_SLOT + 4096 };

struct DenseKsPlan {
  int nchunk;         // K chunks
  int nsplit;         // window splits per (particle, chunk)
  float* slab;        // forward: [nchunk][rows][64] partial pre-activations
  long slab_stride;   // rows * 64
  int mask_x;         // backward: the layer's input is a ReLU output - store dX already masked with [X > 0] (the X chunk is
                      // in LDS anyway), so that the consumer neither reads X again nor rewrites the gradient
  LayerDesc ly;       // the layer's table entry BY VALUE: a kernel argument (scalar registers), not a global load.  A
                      // vector load in the common prologue stays "pending" in hipcc's vmcnt bookkeeping of every role
                      // that does not use it, and turns into vmcnt(N) waits in front of the loaders' unmodelled DMAs
};

template <int EM>
__global__ __launch_bounds__(DK_NW * 64) void dense_ks_fwd_kernel(const GroupArgs A, const DenseKsPlan F) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT);
  constexpr int NTHR = DK_NW * 64;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const BranchDesc& br = A.g.br[0];
  const LayerDesc& ly = F.ly;
  // XCD-aware order: the splits of a (particle, chunk) pair are neighbours in u, hence on the same XCD (its L2 keeps
  // the pair's weight lines)
  const int total = A.cg.S * F.nchunk * F.nsplit;
  const int per = (total + 7) >> 3;
  const int u = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (u >= total) return;
  const int split = u % F.nsplit;
  const int q = u / F.nsplit;
  const int chunk = q % F.nchunk, s = q / F.nchunk;
  const int B = A.cg.B, pp = A.cg.per_particle;
  const int cw8 = min(DK_CH, br.cin_p - chunk * DK_CH) >> 3;   // valid 16-byte pieces per row of this chunk
  const int nks = cw8 >> 2;
  uint4* lut = (uint4*)(smem + DK_SLOTS * DK_SLOT);
  if constexpr (FO) build_sign_lut(lut, tid, NTHR);
  const TensorRef tin = A.t[A.g.in_t];
  const int nwl = split < pp ? (pp - split + F.nsplit - 1) / F.nsplit : 0;   // 32-row windows of this workgroup
  const int nst = 2 * nwl;   // steps: (window, half)

  if (wave >= DK_NC) {
    // =========================== loader waves ===========================
    // loader lw moves plane lw of a step (0 hi, 1 lo): 8 LDS-DMA instructions of 2 rows each; piece position pc of
    // image row r holds channel piece pc ^ r (low 4 bits; swizzle applied to the SOURCE address): the 16-lane groups
    // of ds_read_b128 (8 rows of k-group g, 8 rows of k-group g+1) then cover 16 distinct slots.  Loader 0 also moves the
    // step's sign_in words, loader 1 its sign_out words.
    const int lw = wave - DK_NC;
    const int l5 = lane >> 5, pc = lane & 31;
    const char* plane = (const char*)(lw ? tin.lo : tin.p);
    const uint32_t rowb = (uint32_t)tin.ctot * 2u;
    const uint32_t cb0 = (uint32_t)(br.in_off + chunk * DK_CH) * 2u;
    const uint32_t* sgi = A.nz.sign_in + ly.sign_in_off * A.nz.examples;
    const uint32_t* sgo = A.nz.sign_out + ly.sign_out_off * A.nz.examples;
    int siw = ly.sign_in_words, sow = ly.sign_out_words;
    // every ordinary load (the layer table) is consumed HERE: hipcc does not model the DMA instructions below, and a
    // wait it inserts later for a table field (vmcnt(N) counted over ITS loads) would drain the DMAs in flight
    asm volatile("" : "+v"(sgi), "+v"(sgo), "+v"(siw), "+v"(sow));
    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int k, int slot) {
      const int wl = split + (k >> 1) * F.nsplit;
      const int r_lo = wl * 32 + (k & 1) * DK_ROWS;                 // first row of the step inside the particle
      const int row0 = s * B + r_lo;
      const int nv = max(1, min(DK_ROWS, B - r_lo));                // an all-pad step re-reads its first row
      const int rfix = B - r_lo < 1 ? B - 1 - r_lo : 0;             // ... which is then the particle's last row
      const uint32_t sbase = lds0 + (uint32_t)(slot * DK_SLOT + lw * DK_PLANE);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = pc ^ (2 * i + l5);   // channel piece held at position pc of image row 2i + l5
        const int row = min(2 * i + l5, nv - 1) + rfix;
        const uint32_t off = (uint32_t)(row0 + row) * rowb + cb0 + (uint32_t)(c * 16);
        if (c < cw8 && !(DK_ABL & 1)) dma16(plane + off, sbase + (uint32_t)(i * 1024));
      }
      if constexpr (FO) {
        if (lw == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {   // [16 rows][8 words]
            const int e = i * 64 + lane;
            const int row = min(e >> 3, nv - 1) + rfix, w = e & 7;
            if (chunk * DK_KS + w < siw && !(DK_ABL & 1))
              dma4(sgi + (long)(row0 + row) * siw + chunk * DK_KS + w, lds0 + (uint32_t)(slot * DK_SLOT + 2 * DK_PLANE + i * 256));
          }
        } else {
          const int row = min((lane >> 1) & 15, nv - 1) + rfix, w = lane & 1;
          if (lane < 32 && w < sow && !(DK_ABL & 1)) dma4(sgo + (long)(row0 + row) * sow + w, lds0 + (uint32_t)(slot * DK_SLOT + 2 * DK_PLANE + DK_ROWS * DK_KS * 4));
        }
      }
    };
    const int n_issue = 8 + (FO ? (lw == 0 ? 2 : 1) : 0);
    __syncthreads();   // table visible
    constexpr int ahead = DK_SLOTS - 1;
    for (int j = 0; j < ahead; ++j)
      if (j < nst) issue(j, j);
    for (int k = 0; k < nst; ++k) {
      const int fly = min(ahead - 1, nst - 1 - k) * n_issue;   // steps k+1 .. k+ahead-1 may stay in flight
      if constexpr (DK_ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
      BNN_WAIT_VMCNT_WIDE(fly);
      lds_barrier();                                           // step k landed; step k-1 consumed: its slot is free
      if (k + ahead < nst) issue(k + ahead, (k + ahead) & (DK_SLOTS - 1));
    }
    return;
  }

  // =========================== compute waves ===========================
  asm volatile("" ::: "memory");   // the fragment loads below stay on this side of the role branch (hipcc hoists them into
                                   // the common path otherwise, where they sit in front of the loaders' DMAs in vmcnt order)
  const int i16 = lane & 15, g4 = lane >> 4;
  const int nt = wave;
  const bool has_job = nt < br.ntiles;
  const long sa = A.ws.slot_stride_a * s, sb = A.ws.slot_stride_b * s;
  bf16x8 w_hi[DK_KS], w_lo[DK_KS], w_b[DK_KS];
  {
    const long w_row = (long)(br.n_off + (has_job ? nt : 0) * 16 + i16) * ly.KP + ly.w_off + chunk * DK_CH + g4 * 8;
    const u16* p_hi = (const u16*)A.ws.a_hi + sa + w_row;
    const u16* p_lo = (const u16*)A.ws.a_lo + sa + w_row;
    const u16* p_b = (const u16*)A.ws.b + sb + w_row;
#pragma unroll
    for (int ks = 0; ks < DK_KS; ++ks) {
      const int o = (ks < nks ? ks : 0) * 32;   // k-steps past the end of K are never accumulated
      w_hi[ks] = *(const bf16x8*)(p_hi + o);
      w_lo[ks] = *(const bf16x8*)(p_lo + o);
      if constexpr (FO) w_b[ks] = *(const bf16x8*)(p_b + o);
    }
  }
  // byte offsets of this lane's B fragments inside a plane, per k-step
  int k_o[DK_KS];
#pragma unroll
  for (int ks = 0; ks < DK_KS; ++ks) k_o[ks] = i16 * (DK_CH * 2) + (((ks * 4 + g4) ^ i16) << 4);
  const int chb = nt * 16 + 4 * g4;
  const int bit0 = br.n_off + chb;
  float* slab = F.slab + (long)chunk * F.slab_stride + chb;
  // every fragment has landed before the first barrier: hipcc merges this role's zero-trip exit with the loaders' entry
  // block, and loads it still counts as pending there turn into vmcnt(N) waits in front of the loaders' DMAs
#pragma unroll
  for (int ks = 0; ks < DK_KS; ++ks) {
    asm volatile("" : "+v"(w_hi[ks]), "+v"(w_lo[ks]));
    if constexpr (FO) asm volatile("" : "+v"(w_b[ks]));
  }
  __syncthreads();   // table visible
  for (int k = 0; k < nst; ++k) {
    const int wl = split + (k >> 1) * F.nsplit;
    const int r_lo = wl * 32 + (k & 1) * DK_ROWS;
    const char* sl = smem + (k & (DK_SLOTS - 1)) * DK_SLOT;
    const uint32_t* sg = (const uint32_t*)(sl + 2 * DK_PLANE);
    const uint32_t* so = sg + DK_ROWS * DK_KS;
    lds_barrier();
    if (has_job && r_lo < B) {
      // four independent accumulation chains (hi*hi, hi*lo, lo*hi, perturbation): a chain's MFMAs are 4 issues apart
      f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_hl = {0.f, 0.f, 0.f, 0.f}, acc_lh = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
      uint32_t sw[DK_KS];
      if constexpr (FO) {   // the row's 8 sign words of this chunk: two 16-byte reads
        const u32x4 s0 = *(const u32x4*)(sg + i16 * DK_KS), s1 = *(const u32x4*)(sg + i16 * DK_KS + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sw[e] = s0[e]; sw[4 + e] = s1[e]; }
      }
      auto rd = [&](int ks, bf16x8& bh, bf16x8& bl, u32x4& fm) {
        bh = *(const bf16x8*)(sl + k_o[ks]);
        bl = *(const bf16x8*)(sl + DK_PLANE + k_o[ks]);
        if constexpr (FO) fm = __builtin_bit_cast(u32x4, lut[(sw[ks] >> (8 * g4)) & 0xffu]);   // byte g4 = this lane's 8 channels
      };
      auto mm = [&](int ks, bf16x8 bh, bf16x8 bl, u32x4 fm) {
        acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[ks], bh, acc_a, 0, 0, 0);
        acc_hl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[ks], bl, acc_hl, 0, 0, 0);
        acc_lh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo[ks], bh, acc_lh, 0, 0, 0);
        if constexpr (FO) {
          const u32x4 xb = __builtin_bit_cast(u32x4, bh) ^ fm;
          acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_b[ks], __builtin_bit_cast(bf16x8, xb), acc_b, 0, 0, 0);
        }
      };
      if constexpr (!(DK_ABL & 2)) {
        bf16x8 bh0, bl0, bh1, bl1;
        u32x4 fm0 = {0u, 0u, 0u, 0u}, fm1 = {0u, 0u, 0u, 0u};
        if (nks == DK_KS) {
          // full chunk: the operand reads of k-step ks+1 are in flight while the MFMAs of k-step ks issue (hipcc would
          // sink every read next to its MFMA: two exposed LDS latencies per k-step)
          rd(0, bh0, bl0, fm0);
#pragma unroll
          for (int ks = 0; ks < DK_KS; ks += 2) {
            rd(ks + 1, bh1, bl1, fm1);
            __builtin_amdgcn_sched_barrier(0);
            mm(ks, bh0, bl0, fm0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < DK_KS) rd(ks + 2, bh0, bl0, fm0);
            __builtin_amdgcn_sched_barrier(0);
            mm(ks + 1, bh1, bl1, fm1);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < DK_KS; ++ks) {   // last, shorter chunk (compile-time fragment indices: no scratch)
            if (ks >= nks) break;
            rd(ks, bh0, bl0, fm0);
            mm(ks, bh0, bl0, fm0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_a[r] += acc_hl[r] + acc_lh[r];
      f32x4 v = acc_a;
      if constexpr (FO) {
        const uint32_t bits = so[i16 * 2 + (bit0 >> 5)] >> (bit0 & 31);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += ((bits >> r) & 1u) ? -acc_b[r] : acc_b[r];
      }
      if (r_lo + i16 < B && !(DK_ABL & 4)) *(f32x4*)(slab + ((long)s * B + r_lo + i16) * 64) = v;
      if constexpr (DK_ABL & 4) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    }
  }
}

// out = relu(bias + sum_chunk slab): one thread per (row, 4 channels), chunks summed in order (reproducible).
// FUSE2: the net's last layer Linear(64, 2) (inception.py:217) is evaluated here as well - its input row is in the 16
// lanes' registers - instead of in a launch of its own: z = W2 h + b2 (+ Flipout: s_out o (dW2 (h o s_in))).
struct DenseKsFinArgs {
  const float* slab;
  long slab_stride;
  int nchunk;
  int rows, B;          // S*B rows, rows per particle
  const float* bias;    // [S or 1][bias_stride]: already offset to the layer / branch
  int bias_stride;
  int relu;
  TensorRef out;
  int out_off;
  // last layer (fuse2)
  int fuse2;
  const u16* w2_hi; const u16* w2_lo; const u16* w2_b;   // images of the last layer, offset to its first row
  long w2_stride_a, w2_stride_b;                          // elements between particles (0: shared)
  int w2_KP;
  const float* b2;      // bias of the last layer [S or 1][bias_stride], offset to it
  const uint32_t* sg_in; const uint32_t* sg_out;           // its Flipout sign words [rows][siw] / [rows][sow]
  int siw, sow;
  float* z;             // [rows][2]
  // the last layer's gradient elements, zeroed here for the head launch's atomics: per particle 2 rows of KP in the
  // weight images (slots A and B) and 2 biases
  float* g2_a; float* g2_b; float* g2_ba;
  long g2_stride; int g2_bstride;
  int S;
};

template <int EM>
__global__ __launch_bounds__(256) void dense_ks_fin_kernel(const DenseKsFinArgs F) {
  constexpr bool FO = (EM == EM_FLIPOUT);
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
  if (F.relu) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (live) tstore4(F.out, (long)row * F.out.ctot + F.out_off + ch, v, 4, true);
  if (!F.fuse2) return;
  if (t < F.S * 2 * F.w2_KP) {
    const int zs = t / (2 * F.w2_KP), ze = t - zs * 2 * F.w2_KP;
    F.g2_a[F.g2_stride * zs + ze] = 0.f;
    F.g2_b[F.g2_stride * zs + ze] = 0.f;
    if (ze < 2) F.g2_ba[(long)F.g2_bstride * zs + ze] = 0.f;
  }
  uint2 hv, lv;
  split4(v, hv, lv);
  const f32x4 xb = unpack_bf4(hv);   // what the perturbation path sees (single bf16)
  const f32x4 xm = unpack_bf4(lv);
  float m[2] = {0.f, 0.f}, pz[2] = {0.f, 0.f};
  uint32_t bits = 0;
  if constexpr (FO) bits = F.sg_in[(long)row * F.siw + (ch >> 5)] >> (ch & 31);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long wo = (long)k * F.w2_KP + ch;
    const f32x4 wh = unpack_bf4(*(const uint2*)(F.w2_hi + F.w2_stride_a * s + wo));
    const f32x4 wl = unpack_bf4(*(const uint2*)(F.w2_lo + F.w2_stride_a * s + wo));
#pragma unroll
    for (int r = 0; r < 4; ++r) m[k] += (xb[r] + xm[r]) * (wh[r] + wl[r]);   // x = hi + lo, w = hi + lo
    if constexpr (FO) {
      const f32x4 wb = unpack_bf4(*(const uint2*)(F.w2_b + F.w2_stride_b * s + wo));
#pragma unroll
      for (int r = 0; r < 4; ++r) pz[k] += (((bits >> r) & 1u) ? -xb[r] : xb[r]) * wb[r];
    }
  }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      m[k] += __shfl_xor(m[k], d, 16);
      if constexpr (FO) pz[k] += __shfl_xor(pz[k], d, 16);
    }
  }
  if (live && (t & 15) == 0) {
    uint32_t so = 0;
    if constexpr (FO) so = F.sg_out[(long)row * F.sow];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float zk = m[k] + F.b2[(long)F.bias_stride * s + k];
      if constexpr (FO) zk += ((so >> k) & 1u) ? -pz[k] : pz[k];
      F.z[(long)row * 2 + k] = zk;
    }
  }
}

// ------------------------------------------------------------------------------------------
// head + backward of the net's last layer Linear(64, 2) in one launch (the Inception net's training step): per example row
// the head's d(-ll)/dz, then dH = dz W2 (+ Flipout: s_in o ((dz o s_out) dW2)) written as the bf16 gradient plane of H, and
// per workgroup the layer's weight / bias gradient sums over its rows (fp32 atomics into the zeroed per-particle images).
// Replaces head_nll_kernel + the layer's dX and dW launches.
// ------------------------------------------------------------------------------------------
struct HeadLastArgs {
  HeadArgs H;
  const u16* w_hi; const u16* w_b;      // forward images of the layer (bf16 hi of W | dW), offset to its first row
  long stride_a, stride_b;              // elements between particles (0: shared)
  int KP;
  const u16* h_hi; int h_ctot;          // the layer's input, hi plane [S*B][64]
  u16* dh; int dh_ctot;                 // gradient plane of H
  const uint32_t* sg_in; const uint32_t* sg_out;
  int siw, sow;
  float* gw_a; float* gw_b; float* gb_a;   // offset to the layer; per particle strides below
  long gw_stride; int gb_stride;
};

enum { HL_ROWS = 64 };   // example rows per workgroup

template <int EM>
__global__ __launch_bounds__(256) void head_last_kernel(const HeadLastArgs A) {
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int tid = threadIdx.x;
  const int blk0 = blockIdx.x * HL_ROWS;
  const int idx = blk0 + tid;   // rows: threads 0 .. HL_ROWS-1
  const int s = blockIdx.y;
  const int B = A.H.B;
  __shared__ float dzs[4][HL_ROWS];
  // the layer's two weight rows (W | dW: 4 x 128 B) once per workgroup: 32 dependent global loads per row thread otherwise
  __shared__ uint4 wsh[4][8];
  if (tid < 32) {
    const int kk = tid >> 3, c8 = tid & 7;
    const u16* src = (kk < 2 ? A.w_hi + A.stride_a * s : A.w_b + A.stride_b * s) + (long)(kk & 1) * A.KP + c8 * 8;
    wsh[kk][c8] = (kk < 2 || FO) ? *(const uint4*)src : make_uint4(0u, 0u, 0u, 0u);
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
    const float h0 = (so & 1u) ? -g0 : g0, h1 = (so & 2u) ? -g1 : g1;   // dz o s_out
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
      for (int c8 = 0; c8 < 8; ++c8) {
        const uint4 w0 = wsh[0][c8], w1 = wsh[1][c8];
        const f32x4 a0 = unpack_bf4(make_uint2(w0.x, w0.y)), a1 = unpack_bf4(make_uint2(w0.z, w0.w));
        const f32x4 b0 = unpack_bf4(make_uint2(w1.x, w1.y)), b1 = unpack_bf4(make_uint2(w1.z, w1.w));
        f32x4 d0, d1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d0[q] = g0 * a0[q] + g1 * b0[q];
          d1[q] = g0 * a1[q] + g1 * b1[q];
        }
        if constexpr (FO) {
          const uint4 v0 = wsh[2][c8], v1 = wsh[3][c8];
          const f32x4 e0 = unpack_bf4(make_uint2(v0.x, v0.y)), e1 = unpack_bf4(make_uint2(v0.z, v0.w));
          const f32x4 f0 = unpack_bf4(make_uint2(v1.x, v1.y)), f1 = unpack_bf4(make_uint2(v1.z, v1.w));
          const uint32_t byte = (si[c8 >> 2] >> ((c8 & 3) * 8)) & 0xffu;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float p0 = h0 * e0[q] + h1 * f0[q], p1 = h0 * e1[q] + h1 * f1[q];
            d0[q] += ((byte >> q) & 1u) ? -p0 : p0;
            d1[q] += ((byte >> (4 + q)) & 1u) ? -p1 : p1;
          }
        }
        *(uint4*)(A.dh + r * A.dh_ctot + c8 * 8) =
            make_uint4(cvt_pk(d0[0], d0[1]), cvt_pk(d0[2], d0[3]), cvt_pk(d1[0], d1[1]), cvt_pk(d1[2], d1[3]));
      }
    }
  }
  __syncthreads();
  // ---- weight / bias gradient sums of this workgroup's rows: thread = (kind kk, channel c) ----
  const int kk = tid >> 6, c = tid & 63;   // kk 0, 1: d/dW_a rows 0, 1; kk 2, 3: d/dW_b rows 0, 1 (Flipout)
  const int nrow = min(HL_ROWS, B - blk0);
  if (kk < 2 || FO) {
    float acc = 0.f;
    const u16* hp = A.h_hi + ((long)s * B + blk0) * A.h_ctot + c;
    const uint32_t* sp = A.sg_in + ((long)s * B + blk0) * A.siw + (c >> 5);
    for (int r0 = 0; r0 < HL_ROWS; r0 += 16) {   // 16 rows of loads in flight
      u16 hv[16];
      uint32_t sw[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int rr = min(r0 + j, nrow - 1);
        hv[j] = hp[(long)rr * A.h_ctot];
        sw[j] = (FO && kk >= 2) ? sp[(long)rr * A.siw] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float h = bf2f(hv[j]);
        if ((sw[j] >> (c & 31)) & 1u) h = -h;
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
}

// ==========================================================================================
// dense_ks_bwd_kernel : dX and dW of the wide dense layer for one (particle, 128-channel chunk, window split).
//   13 waves, one barrier per 32-row window:
//   waves 0..3   dX : two 16-channel tiles each, fragments of the transposed weight images in registers;
//                     dX[row][c] = dz W^T + s_in (dz s_out) dW^T, stored as bf16 (a wave's two tiles are neighbours);
//   waves 4..7   dW : one n-tile x the chunk's 8 c-tiles each: 8 (x2 for Flipout) tiles in registers across the
//                     windows, both operands through transposed LDS reads; the chunk-0 workgroups also sum the bias
//                     gradient;
//   waves 8..11     : image builders, EVERY step a quarter of the next window each (a VALU instruction costs its wave
//                     4 cycles: one wave building a whole window was the step time): dz = dY [Y > 0] and dz s_out from
//                     register loads three windows ahead (compile-time ring of 3 register sets), X s_in from the landed
//                     X ring slot;
//   wave  12        : LDS-DMA of the X chunk and its sign_in words into a ring of 6 windows, four windows ahead (HBM
//                     latency under load is 2-3 us, a step far shorter).
//   Images are 256-byte rows with the f128 XOR swizzle (for the DMA'd X: applied to the source address): conflict-free
//   for the transposed reads of dW; the row reads of dX are 2-way.
// ==========================================================================================
enum { DB_CH = 128, DB_ROWS = 32, DB_NX = 4, DB_ND = 4, DB_NB = 4, DB_WAVES = DB_NX + DB_ND + DB_NB + 1, DB_RING = 6, DB_AHEAD = 4 };
enum { DB_IMG = DB_ROWS * 256, DB_SGB = DB_ROWS * 4 * 4,
       DB_O_XR = 0, DB_O_SGR = DB_O_XR + DB_RING * DB_IMG, DB_O_X2 = DB_O_SGR + DB_RING * DB_SGB, DB_O_Z = DB_O_X2 + 2 * DB_IMG,
       DB_LDS = DB_O_Z + 2 * DB_IMG };

// flip the sign of bf16 element e of a 16-byte piece where bit e of `bits` is set
__device__ __forceinline__ tr_u32x4 sgn8v(tr_u32x4 v, uint32_t bits) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] ^= (((bits >> (2 * e)) & 1u) << 15) | (((bits >> (2 * e + 1)) & 1u) << 31);
  return v;
}
// keep the bf16 elements of g whose y (a ReLU output) is positive
__device__ __forceinline__ tr_u32x4 relu_mask8(tr_u32x4 g, tr_u32x4 y) {
#pragma unroll
  for (int e = 0; e < 4; ++e) g[e] &= relu_mask2(y[e]);
  return g;
}

template <int EM>
__global__ __launch_bounds__(DB_WAVES * 64) void dense_ks_bwd_kernel(const GroupArgs A, const DenseKsPlan F) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FO = (EM == EM_FLIPOUT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const BranchDesc& br = A.g.br[0];
  const LayerDesc& ly = F.ly;
  const int total = A.cg.S * F.nchunk * F.nsplit;
  const int per = (total + 7) >> 3;
  const int u = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (u >= total) return;
  const int split = u % F.nsplit;
  const int q = u / F.nsplit;
  const int chunk = q % F.nchunk, s = q / F.nchunk;
  const int B = A.cg.B, pp = A.cg.per_particle;
  const int cw = min(DB_CH, br.cin_p - chunk * DB_CH);   // multiple of 16
  const int nwl = split < pp ? (pp - split + F.nsplit - 1) / F.nsplit : 0;
  // Barriers: P (window 0's X landed, before the builders make its images), Q (window 0's images visible), then B(k)
  // per window.  Every role executes exactly these.

  if (wave == DB_WAVES - 1) {
    // =========================== X loader (LDS-DMA) ===========================
    const TensorRef tin = A.t[A.g.in_t];
    const uint32_t* sgi = A.nz.sign_in + ly.sign_in_off * A.nz.examples;
    const int siw = ly.sign_in_words;
    const int cw8 = cw >> 3;
    const int r4 = lane >> 4, pc = lane & 15;
    const uint32_t rowb = (uint32_t)tin.ctot * 2u;
    uint32_t cb[4], act = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // instruction i covers rows 4i .. 4i+3: f128(row) = (r4 << 2) | (i & 3)
      const int c = pc ^ ((r4 << 2) | i);
      cb[i] = (uint32_t)(br.in_off + chunk * DB_CH + c * 8) * 2u;
      if (c < cw8) act |= 1u << i;
    }
    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int k, int slot) {
      const int wl = split + k * F.nsplit;
      const int row0 = s * B + wl * DB_ROWS;
      const int nv = min(DB_ROWS, B - wl * DB_ROWS);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = min(4 * i + r4, nv - 1);   // pad rows re-read the last valid one (their dz rows are zero)
        const uint32_t off = (uint32_t)(row0 + row) * rowb + cb[i & 3];
        if (((act >> (i & 3)) & 1u) && !(DK_ABL & 16)) dma16((const char*)tin.p + off, lds0 + (uint32_t)(DB_O_XR + slot * DB_IMG + i * 1024));
      }
      if constexpr (FO) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // [32 rows][4 words]
          const int e = i * 64 + lane;
          const int row = min(e >> 2, nv - 1), w = e & 3;
          if (chunk * 4 + w < siw && !(DK_ABL & 16))
            dma4(sgi + (long)(row0 + row) * siw + chunk * 4 + w, lds0 + (uint32_t)(DB_O_SGR + slot * DB_SGB + i * 256));
        }
      }
    };
    constexpr int n_issue = 8 + (FO ? 2 : 0);
    for (int j = 0; j < DB_AHEAD; ++j)
      if (j < nwl) issue(j, j);
    if constexpr (DK_ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
    BNN_WAIT_VMCNT_WIDE(max(0, min(nwl, DB_AHEAD) - 1) * n_issue);   // window 0 landed
    lds_barrier();   // P
    lds_barrier();   // Q
    int slot_i = DB_AHEAD % DB_RING;     // ring slot of window k + DB_AHEAD
    for (int k = 0; k < nwl; ++k) {
      // window k+1 landed (the builders read it during step k); windows k+2 .. k+3 may stay in flight
      const int fly = max(0, min(k + DB_AHEAD - 1, nwl - 1) - (k + 1)) * n_issue;
      if constexpr (DK_ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
      BNN_WAIT_VMCNT_WIDE(fly);
      lds_barrier();   // B(k)
      if (k + DB_AHEAD < nwl) issue(k + DB_AHEAD, slot_i);   // slot of window k-2: consumed
      slot_i = slot_i + 1 == DB_RING ? 0 : slot_i + 1;
    }
    return;
  }
  if (wave >= DB_NX + DB_ND) {
    // =========================== image builders ===========================
    asm volatile("" ::: "memory");
    const int bw = wave - (DB_NX + DB_ND);
    const TensorRef tg = A.t[br.out_t + T_GRAD], ty = A.t[br.out_t];
    const uint32_t* sgo = A.nz.sign_out + ly.sign_out_off * A.nz.examples;
    const int sow = ly.sign_out_words;
    // dz unit of this lane: (row zrow, 8-channel piece c8).  Loads are unconditional (pad rows re-read the last valid
    // one, masked when the image is written): a load under a branch is waited for at the join, i.e. immediately.
    const int c8 = lane & 7, zrow = bw * 8 + (lane >> 3);
    const bool c_on = c8 * 8 < br.cout;
    const int cc8 = c_on ? c8 : 0;
    const int bit0 = br.n_off + cc8 * 8;
    const int zo = zrow * 256 + ((c8 ^ f128(zrow)) << 4);
    auto load = [&](int k, tr_u32x4& dy, tr_u32x4& yy, uint32_t& so, int& nvw) {
      const int wl = split + k * F.nsplit;
      const long row0 = (long)s * B + wl * DB_ROWS;
      nvw = min(DB_ROWS, B - wl * DB_ROWS);
      const long o = (row0 + min(zrow, nvw - 1)) * tg.ctot + br.out_off + cc8 * 8;
      if constexpr (!(DK_ABL & 64)) {
        dy = *(const tr_u32x4*)((const u16*)tg.p + o);
        yy = *(const tr_u32x4*)((const u16*)ty.p + o);
        if constexpr (FO) so = sgo[(row0 + min(zrow, nvw - 1)) * sow + (bit0 >> 5)];
      }
    };
    // images of window k: the lane's dz unit from a register set, two pieces of X s_in from the window's ring slot
    auto make = [&](int k, int ring, tr_u32x4 dy, tr_u32x4 yy, uint32_t so, int nvw) {
      char* zi = smem + DB_O_Z + (k & 1) * DB_IMG;
      tr_u32x4 g = dy;
      if (br.relu) g = relu_mask8(g, yy);
      if (zrow >= nvw || !c_on) g = tr_u32x4{0u, 0u, 0u, 0u};
      *(tr_u32x4*)(zi + zo) = g;
      if constexpr (FO) {
        *(tr_u32x4*)(zi + (zo ^ 128)) = sgn8v(g, so >> (bit0 & 31));
        if constexpr (!(DK_ABL & 32)) {
          const char* xr = smem + DB_O_XR + ring * DB_IMG;
          const uint32_t* sg = (const uint32_t*)(smem + DB_O_SGR + ring * DB_SGB);
          char* x2 = smem + DB_O_X2 + (k & 1) * DB_IMG;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int e = (bw * 2 + j) * 64 + lane;
            const int row = e >> 4, c = (e & 15) ^ f128(row);
            const tr_u32x4 v = *(const tr_u32x4*)(xr + e * 16);
            const uint32_t word = sg[row * 4 + (c >> 2)];
            *(tr_u32x4*)(x2 + e * 16) = sgn8v(v, word >> (8 * (c & 3)));
          }
        }
      }
    };
    tr_u32x4 dy0 = {0u, 0u, 0u, 0u}, yy0 = dy0, dy1 = dy0, yy1 = dy0, dy2 = dy0, yy2 = dy0;
    uint32_t so0 = 0, so1 = 0, so2 = 0;
    int nv0 = 0, nv1 = 0, nv2 = 0;
    if (0 < nwl) load(0, dy0, yy0, so0, nv0);
    if (1 < nwl) load(1, dy1, yy1, so1, nv1);
    if (2 < nwl) load(2, dy2, yy2, so2, nv2);
    lds_barrier();   // P: window 0's X landed
    if (nwl > 0) {
      make(0, 0, dy0, yy0, so0, nv0);
      if (3 < nwl) load(3, dy0, yy0, so0, nv0);
    }
    lds_barrier();   // Q
    // step k (after B(k)): images of window k+1 from register set (k+1) % 3, then that set is reloaded with window k+4
    int ring = 1;
#define DB_STEP(K, DY, YY, SO, NV)                          \
  do {                                                      \
    lds_barrier();                                          \
    if ((K) + 1 < nwl) {                                    \
      make((K) + 1, ring, DY, YY, SO, NV);                  \
      if ((K) + 4 < nwl) load((K) + 4, DY, YY, SO, NV);     \
    }                                                       \
    ring = ring + 1 == DB_RING ? 0 : ring + 1;              \
  } while (0)
    for (int k = 0; k < nwl; k += 3) {
      DB_STEP(k, dy1, yy1, so1, nv1);
      if (k + 1 < nwl) DB_STEP(k + 1, dy2, yy2, so2, nv2);
      if (k + 2 < nwl) DB_STEP(k + 2, dy0, yy0, so0, nv0);
    }
#undef DB_STEP
    // nothing is in flight here, but hipcc's bookkeeping says otherwise (the loads of the last loop body), and its
    // structurised control flow lets that state reach the DMA role's code: close it with a use of every destination
    asm volatile("" ::"v"(dy0), "v"(yy0), "v"(so0), "v"(dy1), "v"(yy1), "v"(so1), "v"(dy2), "v"(yy2), "v"(so2));
    return;
  }

  asm volatile("" ::: "memory");   // keep the compute roles' loads out of the common path (see the forward)
  const int i16 = lane & 15, g4 = lane >> 4;
  const long sat = A.ws.slott_stride_a * s, sbt = A.ws.slott_stride_b * s;
  if (wave < DB_NX) {
    // =========================== dX waves ===========================
    bf16x8 wa[2][2], wb[2][2];
    const u16* wat = (const u16*)A.ws.at + sat + ly.wt_off;
    const u16* wbt = (const u16*)A.ws.bt + sbt + ly.wt_off;
    const int KPt = ly.KPt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = 2 * wave + j;
      const bool on = t * 16 < cw;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const long wo = (long)(chunk * DB_CH + (on ? t : 0) * 16 + i16) * KPt + br.n_off + ks * 32 + g4 * 8;
        wa[j][ks] = *(const bf16x8*)(wat + wo);
        if constexpr (FO) wb[j][ks] = *(const bf16x8*)(wbt + wo);
      }
    }
    const TensorRef tdx = A.t[br.dx_t];
    int z_o[2][2];   // [ks][mt] byte offsets of this lane's dz fragments; dz s_out sits 8 pieces further
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        z_o[ks][mt] = row * 256 + (((ks * 4 + g4) ^ f128(row)) << 4);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {   // fragments landed before the first barrier (see the forward)
        asm volatile("" : "+v"(wa[j][ks]));
        if constexpr (FO) asm volatile("" : "+v"(wb[j][ks]));
      }
    lds_barrier();   // P
    lds_barrier();   // Q
    int ring = 0;
    for (int k = 0; k < nwl; ++k) {
      const int wl = split + k * F.nsplit;
      const char* zi = smem + DB_O_Z + (k & 1) * DB_IMG;
      const uint32_t* sg = (const uint32_t*)(smem + DB_O_SGR + ring * DB_SGB);
      lds_barrier();
      bf16x8 bz[2][2], bz2[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          bz[ks][mt] = *(const bf16x8*)(zi + z_o[ks][mt]);
          if constexpr (FO) bz2[ks][mt] = *(const bf16x8*)(zi + (z_o[ks][mt] ^ 128));
        }
      const int nv = min(DB_ROWS, B - wl * DB_ROWS);
      const long row0 = (long)s * B + wl * DB_ROWS;
#pragma unroll
      for (int j = 0; j < ((DK_ABL & 512) ? 0 : 2); ++j) {
        const int t = 2 * wave + j;
        if (t * 16 >= cw) break;
        f32x4 acc_a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 acc_b[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][ks], bz[ks][mt], acc_a[mt], 0, 0, 0);
            if constexpr (FO) acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][ks], bz2[ks][mt], acc_b[mt], 0, 0, 0);
          }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + i16;
          f32x4 v = acc_a[mt];
          if constexpr (FO) {
            const uint32_t bits = sg[row * 4 + (t >> 1)] >> ((t & 1) * 16 + 4 * g4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += ((bits >> r) & 1u) ? -acc_b[mt][r] : acc_b[mt][r];
          }
          if (F.mask_x) {
            const int ch = t * 16 + 4 * g4;
            const uint2 xv = *(const uint2*)(smem + DB_O_XR + ring * DB_IMG + row * 256 + (((ch >> 3) ^ f128(row)) << 4) + (ch & 7) * 2);
            const f32x4 xf = unpack_bf4(xv);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = xf[r] > 0.f ? v[r] : 0.f;
          }
          if (row < nv && !(DK_ABL & 128)) tstore4(tdx, (row0 + row) * tdx.ctot + br.in_off + chunk * DB_CH + t * 16 + 4 * g4, v, 4, true);
        }
      }
      ring = ring + 1 == DB_RING ? 0 : ring + 1;
    }
    return;
  }

  // =========================== dW waves ===========================
  const int nt = wave - DB_NX;
  const bool bias_job = chunk == 0;
  f32x4 acc_a[8], acc_b[FO ? 8 : 1], acc_bias = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    acc_a[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (FO) acc_b[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
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
    const char* zi = smem + DB_O_Z + (k & 1) * DB_IMG;
    const char* xi = smem + DB_O_XR + ring * DB_IMG;
    const char* x2 = smem + DB_O_X2 + (k & 1) * DB_IMG;
    lds_barrier();
    const bf16x8 fa = tr_frag2(zi + a0, zi + a1);
    bf16x8 fa2 = fa;
    if constexpr (FO) fa2 = tr_frag2(zi + (a0 ^ 128), zi + (a1 ^ 128));
    if (bias_job) acc_bias = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, acc_bias, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < ((DK_ABL & 256) ? 0 : 8); ++c) {
      if (c * 16 >= cw) break;
      const int cb = c * 2 + (p4 >> 1);
      const int b0 = r0 * 256 + ((cb ^ f0) << 4) + ho, b1 = (r0 + 4) * 256 + ((cb ^ f1) << 4) + ho;
      const bf16x8 fb = tr_frag2(xi + b0, xi + b1);
      acc_a[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc_a[c], 0, 0, 0);
      if constexpr (FO) {
        const bf16x8 fb2 = tr_frag2(x2 + b0, x2 + b1);
        acc_b[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa2, fb2, acc_b[c], 0, 0, 0);
      }
    }
    ring = ring + 1 == DB_RING ? 0 : ring + 1;
  }
  float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
  float* gwb = A.gw_b + A.gw_stride * s + ly.w_off;
  const int i4 = 4 * (lane >> 4), jc = lane & 15;
  const bool direct = F.nsplit == 1;   // one workgroup per (particle, chunk): its tiles ARE the gradient
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int cc = chunk * DB_CH + c * 16 + jc;
    if (c * 16 >= cw) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      const long o = (long)(br.n_off + n) * ly.KP + cc;
      if (direct) {
        gwa[o] = acc_a[c][r];
        if constexpr (FO) gwb[o] = acc_b[c][r];
      } else {
        atomicAdd(gwa + o, acc_a[c][r]);
        if constexpr (FO) atomicAdd(gwb + o, acc_b[c][r]);
      }
    }
  }
  if (bias_job && jc == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      float* gb = A.gb_a + (long)A.gb_stride * s + ly.bias_off + br.n_off + n;
      if (direct) *gb = acc_bias[r];
      else atomicAdd(gb, acc_bias[r]);
    }
  }
}
