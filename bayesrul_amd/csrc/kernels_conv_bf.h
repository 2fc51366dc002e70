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
  int no_bias;            // 1: this launch covers part of the group's tiles and another launch sums the bias gradients
  float* slab_a;          // non-null: the workgroup stores its tiles into its own partial image (slab blockIdx.x, forward image
  float* slab_b;          // layout) instead of adding them to the gradient images with atomics; slab_reduce_kernel sums the
  long slab_stride;       // slabs in a fixed order
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

// one compute wave of the role-specialised conv forward: an (n-tile, k-step range) job; jobs may split K
// across waves (block-2 k3/k5 branches have only one n-tile each): partial accumulators are reduced
// through LDS by the owning wave
struct FwdJob {
  signed char b, nt, ks0, ks1;   // branch (-1: none), n-tile, k-step range [ks0, ks1)
  signed char grp, owner, member, nmember;  // K-split reduction group (-1: none)
};

// bf16 hi/lo planes [rows][CP] of the raw fp32 windows (channel pads zero)
__global__ void x_planes_kernel(const float* x, u16* hi, u16* lo, long rows, int F, int CP) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * CP) return;
  const long r = idx / CP;
  const int c = (int)(idx - r * CP);
  const float v = c < F ? x[r * F + c] : 0.f;
  const u16 h = f2bf(v);
  hi[idx] = h;
  lo[idx] = f2bf(v - bf2f(h));
}

// ==========================================================================================
// conv_fwd_dma_kernel : role-specialised forward of a conv group (bf16 planes).
//   16 waves: waves [0, FW_NC) compute (one (n-tile, k-range) job each, weight fragments in
//   registers), waves [FW_NC, 16) are LOADERS that move the next windows' hi/lo planes
//   global -> LDS with LDS-DMA (global_load_lds_dwordx4), two windows ahead, and never touch a
//   register or a store: their vmcnt counts nothing but their own DMAs, so the counted wait
//   is exact.  Compute waves never wait on memory inside the window loop (their epilogue
//   stores are fire-and-forget).  LDS images are dense copies of the global rows; bank
//   conflicts are removed by an XOR swizzle applied to the per-lane SOURCE address of the DMA
//   and to every read (chunk position p of image row r holds channel chunk p ^ (r & mask)).
// ==========================================================================================
enum { FW_WAVES = 16, FW_THREADS = 1024, FW_NC = 12, FW_NL = 4, FW_KS = 4, FW_SLOTS = 3 };
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) char lds_char_t;

// LDS-DMA issued through inline asm so that hipcc does not model it: no compiler-inserted
// vmcnt(0) before later LDS reads / barriers; completion is waited for with explicit counted
// s_waitcnt vmcnt(N) by the issuing (loader) wave.  M0 carries the wave-uniform LDS byte address
// and is restored (cdna_hip_programming.md §5.7).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_char_t*)p; }
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// two floats -> packed bf16 pair (one v_cvt_pk_bf16_f32)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// hi / lo bf16 planes of 4 floats: hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void split4(f32x4 v, uint2& hi, uint2& lo) {
  const uint32_t h01 = cvt_pk(v[0], v[1]), h23 = cvt_pk(v[2], v[3]);
  const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xffff0000u);
  const float r2 = v[2] - __uint_as_float(h23 << 16), r3 = v[3] - __uint_as_float(h23 & 0xffff0000u);
  hi = make_uint2(h01, h23);
  lo = make_uint2(cvt_pk(r0, r1), cvt_pk(r2, r3));
}
// 256-entry table: sign byte (8 channels) -> XOR mask of a bf16x8 fragment (one ds_read_b128)
__device__ __forceinline__ void build_sign_lut(uint4* lut, int tid, int nthreads) {
  for (int b = tid; b < 256; b += nthreads) {
    uint32_t m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = (((b >> (2 * q)) & 1u) << 15) | (((b >> (2 * q + 1)) & 1u) << 31);
    lut[b] = make_uint4(m[0], m[1], m[2], m[3]);
  }
}

// diagnostics only (GroupArgs::dbg, null in production): lane 0 of every wave of workgroup 0 records
// s_memtime at phase `ph` of window iteration `k`
#define BNN_STAMP_DECL(A)                                                                                   \
  const bool stamp_on = (A).dbg != nullptr && blockIdx.x == (unsigned)(A).dbg_block && (threadIdx.x & 63) == 0;                   \
  const int stamp_wave = threadIdx.x >> 6;                                                                  \
  auto stamp = [&](int k, int ph) {                                                                         \
    if (stamp_on && k < 48) (A).dbg[((size_t)stamp_wave * 48 + k) * 8 + ph] = __builtin_amdgcn_s_memtime(); \
  }

// workgroup barrier that only waits for this wave's LDS traffic (never for VMEM)
// a pointer known to be the same in every lane, moved to scalar registers
__device__ __forceinline__ void* uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (void*)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#define BNN_WAIT_VMCNT(N)                                                     \
  do {                                                                        \
    switch (N) {                                                              \
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;         \
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;         \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;         \
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;         \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;         \
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;         \
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;         \
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;         \
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;         \
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;         \
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;       \
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;       \
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;       \
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;        \
    }                                                                         \
  } while (0)

// The branches of a group may be dealt to FW_KINDS workgroup kinds (two 8-wave workgroups per CU
// instead of one 16-wave workgroup: their barrier / DMA stalls overlap).
enum { FW_KINDS = 2 };
struct ConvFwd2Plan {
  int nsplit, nkinds, pad0_, pad1_;
  int has_pool[FW_KINDS], n_red_groups[FW_KINDS];
  FwdJob job[FW_KINDS][FW_NC];
};

template <int EM, int NC, int NL>
__global__ __launch_bounds__((NC + NL) * 64, 4) void conv_fwd_dma_kernel(const GroupArgs A, const ConvFwd2Plan F) {
  constexpr int NTHR = (NC + NL) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  constexpr int KS = FW_KS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const int kind = blockIdx.x % F.nkinds;
  const int bid = blockIdx.x / F.nkinds;
  const int s = bid / F.nsplit, split = bid - s * F.nsplit;
  const int has_pool = F.has_pool[kind], n_red_groups = F.n_red_groups[kind];
  const int L = G.L, B = A.cg.B;
  const int cwp = G.in_cin_p, c8n = cwp >> 3;
  const int swm = (c8n - 1) & 15;           // swizzle mask (c8n is a power of two: 4 or 16)
  const int RS = cwp;                       // dense rows
  const int pbytes = IMG_ROWS * RS * 2;     // one plane incl. halo rows (multiple of 16)
  // LDS: raw[FW_SLOTS][hi|lo] | derived: pooled hi | pooled lo | sq | pooled sq | sign words | red
  u16* raw = (u16*)smem;
  u16* der = (u16*)(smem + FW_SLOTS * 2 * pbytes);
  uint32_t* sgn = (uint32_t*)(smem + (FW_SLOTS * 2 + 4) * pbytes);   // [FW_SLOTS][64]
  uint4* lut = (uint4*)(smem + (FW_SLOTS * 2 + 4) * pbytes + FW_SLOTS * 64 * 4);   // 4 KB sign-mask table
  float* red = (float*)(smem + (FW_SLOTS * 2 + 4) * pbytes + FW_SLOTS * 64 * 4 + 4096);
  {
    const int total = ((FW_SLOTS * 2 + 4) * pbytes + FW_SLOTS * 64 * 4) >> 2;
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < total; k += NTHR) z[k] = 0u;
  }
  const TensorRef tin = A.t[G.in_t];
  const int nchunk = L * c8n;                       // 16-byte chunks per plane per window
  const int ninst = (nchunk + 63) >> 6;             // DMA instructions per plane
  const bool is_loader = wave >= NC;
  const int lw = wave - NC;
  const int pp = B;
  const int nwin = (pp - split + F.nsplit - 1) / F.nsplit;   // windows of this workgroup
  auto win_of = [&](int k) { return split + k * F.nsplit; };

  // =========================== loader state ===========================
  // DMA instructions of a window: 2 planes x ninst, dealt round-robin to the 4 loader waves;
  // addresses are recomputed per issue (no per-thread arrays: they would land in scratch)
  const int my_ninst = is_loader ? max(0, (2 * ninst - lw + NL - 1) / NL) : 0;
  // flipout sign words of a window: [branch][8] = 4 words sign_in + 2 words sign_out, by loader 0
  const uint32_t* sg_src = nullptr;
  long sg_stride = 0;
  bool sg_ok = false;
  if (EM == EM_FLIPOUT && is_loader && lw == 0 && lane < 8 * G.n_branch) {
    const int b = lane >> 3, k = lane & 7;
    const BranchDesc& br = G.br[b];
    const LayerDesc& ly = A.layers[br.layer];
    if (k < 4 && k < ly.sign_in_words) {
      sg_src = A.nz.sign_in + ly.sign_in_off * A.nz.examples + k;
      sg_stride = ly.sign_in_words;
      sg_ok = true;
    } else if (k >= 4 && k - 4 < ly.sign_out_words && k < 6) {
      sg_src = A.nz.sign_out + ly.sign_out_off * A.nz.examples + (k - 4);
      sg_stride = ly.sign_out_words;
      sg_ok = true;
    }
  }
  // =========================== loader waves: own code path (own register allocation) ===========================
  if (is_loader) {
    // Every DMA instruction of this loader keeps its per-lane source address (window 0 of the workgroup) in a
    // VGPR pair that advances by `a_step` per issue: the window loop contains no address arithmetic.
    // issue() must be called for windows 0, 1, 2, ... in order.
    constexpr int LI = 8;   // instructions per loader (2 planes x ninst <= 16, NL >= 2)
    const char* a_src[LI];
    uint32_t a_dst[LI];
    uint32_t a_on = 0;
    {
      const int wl0 = split;
      const long row0 = (long)(G.in_bcast ? wl0 : s * B + wl0) * L * tin.ctot;
#pragma unroll
      for (int i = 0; i < LI; ++i) {
        a_src[i] = nullptr;
        a_dst[i] = 0;
        const int inst = lw + i * NL;
        if (is_loader && inst < 2 * ninst) {
          const int plane = inst >= ninst ? 1 : 0;
          const int q0 = (inst - plane * ninst) * 64;
          const int q = q0 + lane;
          const int row = q / c8n, pz = q - row * c8n;
          const int c8 = pz ^ ((row + HALO) & swm);
          a_src[i] = (const char*)((const u16*)(plane ? tin.lo : tin.p) + row0 + (long)row * tin.ctot + c8 * 8);
          a_dst[i] = (uint32_t)(plane * pbytes + (HALO * RS * 2) + q0 * 16);
          if (q < nchunk) a_on |= 1u << i;
        }
      }
    }
    const long a_step = (long)F.nsplit * L * tin.ctot * 2;
    const long sg_step = (long)F.nsplit * sg_stride;
    if (sg_ok) sg_src += ((long)s * B + split) * sg_stride;
    // all ordinary loads of the loader state are consumed here (a compiler-placed vmcnt wait inside the
    // issue sequence would serialise the unmodelled DMAs)
    asm volatile("" : "+v"(sg_src));
    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int k) {   // DMA the next window (k) of this workgroup into slot k % FW_SLOTS
      const uint32_t sbase = lds0 + (uint32_t)((k % FW_SLOTS) * 2 * pbytes);
#pragma unroll
      for (int i = 0; i < LI; ++i) {
        if (lw + i * NL >= 2 * ninst) break;
        uint32_t on = a_on;
        asm volatile("" : "+v"(on));
        if ((on >> i) & 1u) dma16(a_src[i], __builtin_amdgcn_readfirstlane(sbase + a_dst[i]));
        a_src[i] += a_step;
      }
      if (EM == EM_FLIPOUT && lw == 0) {
        if (sg_ok) dma4(sg_src, lds0 + (uint32_t)((FW_SLOTS * 2 + 4) * pbytes + (k % FW_SLOTS) * 256));
        sg_src += sg_step;
      }
    };
    if (EM == EM_FLIPOUT) build_sign_lut(lut, tid, NTHR);
    __syncthreads();  // zero fill visible
    BNN_STAMP_DECL(A);
    if (nwin > 0) issue(0);
    if (nwin > 1) issue(1);
    const bool derive = (has_pool || LRT) && !(A.pool_sel & 2);
    const int nfly = my_ninst + ((EM == EM_FLIPOUT && lw == 0) ? 1 : 0);   // DMAs of one window
    for (int k = 0; k < nwin; ++k) {
      stamp(k, 0);
      // window k landed: all but the DMAs of window k+1 are complete
      if (k + 1 < nwin) BNN_WAIT_VMCNT(nfly);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(k, 1);
      lds_barrier();                                // B1
      stamp(k, 2);
      if (derive) lds_barrier();                    // B2 (derived planes are built by the compute waves)
      stamp(k, 4);
      if (k + 2 < nwin) issue(k + 2);               // slot (k+2)%3 == (k-1)%3: free since B1
      stamp(k, 5);
      if (n_red_groups > 0) lds_barrier();          // K-split reduction barrier
    }
    return;
  }

  // =========================== compute state ===========================
  const int i16 = lane & 15, g4 = lane >> 4;
  bf16x8 w_hi[KS], w_lo[KS], w_b[KS];
  int k_o[KS];     // LDS element offset of this lane's B fragment, m-tile 0 (m-tile 1: 16 rows further - the swizzle
                   // mask is < 16, so the chunk position is the same): window invariant
  int k_sb[KS];    // flipout: sign word index | bit shift << 8 of the lane's 8 channels
  FwdJob J = FwdJob{-1, 0, 0, 0, -1, 0, 0, 0};
  int j_nks = 0, j_pool = 0;
  f32x4 e_ba = {0.f, 0.f, 0.f, 0.f}, e_bb = {0.f, 0.f, 0.f, 0.f};
  int e_nv = 0, e_ooff = 0, e_octot = 0, e_relu = 0, e_layer = 0, e_lch = 0, e_cout = 0, e_c4n = 0, e_sob = 0;
  u16 *e_ohi = nullptr, *e_olo = nullptr, *e_q = nullptr;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    w_hi[ks] = w_lo[ks] = w_b[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    k_o[ks] = HALO * RS;
    k_sb[ks] = 0;
  }
  if (!is_loader) {
    J = F.job[kind][wave];
    if (J.b >= 0) {
      const BranchDesc& br = G.br[J.b];
      const LayerDesc& ly = A.layers[br.layer];
      j_nks = J.ks1 - J.ks0;
      j_pool = br.pool;
      const int G8 = br.cin_p >> 3;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < j_nks) {
          const int gg = (J.ks0 + ks) * 4 + g4;
          const int tap = gg / G8, c8 = gg - tap * G8;
          const bool valid = tap < ly.taps;
          const long wo = (long)(br.n_off + J.nt * 16 + i16) * ly.KP + (long)(J.ks0 + ks) * 32 + g4 * 8;
          const long sa = A.ws.slot_stride_a * s, sb = A.ws.slot_stride_b * s;
          w_hi[ks] = *(const bf16x8*)((const u16*)A.ws.a_hi + sa + ly.w_off + wo);
          w_lo[ks] = *(const bf16x8*)((const u16*)A.ws.a_lo + sa + ly.w_off + wo);
          if constexpr (DUAL) w_b[ks] = *(const bf16x8*)((const u16*)A.ws.b + sb + ly.w_off + wo);
          {
            const int rb = valid ? (tap - ly.pad + HALO) : HALO;
            const int cg = valid ? ((br.in_off >> 3) + c8) : 0;
            const int rr = rb + i16;
            k_o[ks] = rr * RS + ((cg ^ (rr & swm)) * 8);
            const int cl = valid ? c8 : 0;
            k_sb[ks] = (cl >> 2) | (((cl & 3) * 8) << 8);
          }
        }
      }
      const int chb = J.nt * 16 + 4 * g4;
      e_nv = br.cout - chb;
      const float* ba = A.ws.bias_a + (long)A.ws.bias_stride_a * s + ly.bias_off + br.n_off + chb;
      const float* bb = A.ws.bias_b + ly.bias_off + br.n_off + chb;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < e_nv) {
          e_ba[r] = ba[r];
          if constexpr (LRT) e_bb[r] = bb[r];
        }
      const TensorRef tout = A.t[br.out_t];
      // wave-uniform epilogue state, pinned to scalar registers (the vector file is the tight one here)
      e_ohi = (u16*)uniform_ptr(tout.p);
      e_olo = (u16*)uniform_ptr(tout.lo);
      e_q = (u16*)uniform_ptr(A.t[br.q_t].p);
      e_octot = __builtin_amdgcn_readfirstlane(tout.ctot);
      e_ooff = br.out_off + chb;
      e_relu = __builtin_amdgcn_readfirstlane(br.relu);
      e_layer = __builtin_amdgcn_readfirstlane(br.layer);
      e_lch = br.n_off + chb;
      e_cout = __builtin_amdgcn_readfirstlane(ly.cout);
      e_c4n = __builtin_amdgcn_readfirstlane(ly.cout_p16 >> 2);
      e_sob = br.n_off + J.nt * 16;   // first sign_out bit of this n-tile
    }
  }

  // =========================== prologue ===========================
  if (EM == EM_FLIPOUT) build_sign_lut(lut, tid, NTHR);
  __syncthreads();  // zero fill visible

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  BNN_STAMP_DECL(A);
  for (int k = 0; k < nwin; ++k) {
    stamp(k, 0);
    const int slot = k % FW_SLOTS;
    u16* r_hi = raw + slot * pbytes;             // 2 planes per slot: elements = 2*pbytes/2
    u16* r_lo = r_hi + (pbytes >> 1);
    stamp(k, 1);
    lds_barrier();                                // B1: raw planes of window k visible; compute(k-1) finished
    stamp(k, 2);
    // ---- derived planes (all 16 waves) ----
    if ((has_pool || LRT) && !(A.pool_sel & 2)) {
      // compute waves only: the loaders' vmcnt must count nothing but their own DMAs (the arg-max store)
      if (!is_loader)
      for (int U = tid; U < nchunk; U += NC * 64) {
        const int row = U / c8n, p = U - row * c8n;
        const int ri = row + HALO;
        const int c8 = p ^ (ri & swm);
        const int o = ri * RS + p * 8;
        const uint4 h0 = *(const uint4*)&r_hi[o], l0 = *(const uint4*)&r_lo[o];
        const uint32_t hh[4] = {h0.x, h0.y, h0.z, h0.w}, ll[4] = {l0.x, l0.y, l0.z, l0.w};
        uint32_t ha[4] = {0, 0, 0, 0}, la[4] = {0, 0, 0, 0}, hb[4] = {0, 0, 0, 0}, lb[4] = {0, 0, 0, 0};
        const bool up = row > 0, dn = row + 1 < L;
        if (has_pool) {
          if (up) {
            const int o2 = (ri - 1) * RS + ((c8 ^ ((ri - 1) & swm)) * 8);
            const uint4 a = *(const uint4*)&r_hi[o2], b = *(const uint4*)&r_lo[o2];
            ha[0] = a.x; ha[1] = a.y; ha[2] = a.z; ha[3] = a.w;
            la[0] = b.x; la[1] = b.y; la[2] = b.z; la[3] = b.w;
          }
          if (dn) {
            const int o2 = (ri + 1) * RS + ((c8 ^ ((ri + 1) & swm)) * 8);
            const uint4 a = *(const uint4*)&r_hi[o2], b = *(const uint4*)&r_lo[o2];
            hb[0] = a.x; hb[1] = a.y; hb[2] = a.z; hb[3] = a.w;
            lb[0] = b.x; lb[1] = b.y; lb[2] = b.z; lb[3] = b.w;
          }
        }
        uint32_t ph[4], pl[4], sq[4], psq[4];
        uint32_t am[2] = {0u, 0u};   // arg-max code of each of the 8 channels: 0 = row-1, 1 = row, 2 = row+1
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t oh = 0, ol = 0, os = 0, ops = 0;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int sh = 16 * e;
            const u16 h = (u16)(hh[q] >> sh), l = (u16)(ll[q] >> sh);
            u16 bh_ = h, bl_ = l;
            if (has_pool) {
              // torch scans row-1, row, row+1 and keeps the FIRST maximum
              float best = bf2f(h) + bf2f(l);
              uint32_t code = 1u;
              if (up) {
                const u16 h2 = (u16)(ha[q] >> sh), l2 = (u16)(la[q] >> sh);
                const float v = bf2f(h2) + bf2f(l2);
                if (v >= best) { best = v; bh_ = h2; bl_ = l2; code = 0u; }
              }
              if (dn) {
                const u16 h2 = (u16)(hb[q] >> sh), l2 = (u16)(lb[q] >> sh);
                const float v = bf2f(h2) + bf2f(l2);
                if (v > best) { best = v; bh_ = h2; bl_ = l2; code = 2u; }
              }
              am[q >> 1] |= code << (8 * (2 * (q & 1) + e));
            }
            oh |= (uint32_t)bh_ << sh;
            ol |= (uint32_t)bl_ << sh;
            if constexpr (LRT) {
              const float x = bf2f(h), xp = bf2f(bh_);
              os |= (uint32_t)f2bf(x * x) << sh;
              ops |= (uint32_t)f2bf(xp * xp) << sh;
            }
          }
          ph[q] = oh; pl[q] = ol; sq[q] = os; psq[q] = ops;
        }
        if (has_pool) {
          *(uint4*)&der[o] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          *(uint4*)&der[(pbytes >> 1) + o] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          if (A.amax) {
            const long R = (long)(s * B + win_of(k)) * L + row;
            *(uint2*)(A.amax + R * cwp + c8 * 8) = make_uint2(am[0], am[1]);
          }
        }
        if constexpr (LRT) {
          *(uint4*)&der[pbytes + o] = make_uint4(sq[0], sq[1], sq[2], sq[3]);
          if (has_pool) *(uint4*)&der[pbytes + (pbytes >> 1) + o] = make_uint4(psq[0], psq[1], psq[2], psq[3]);
        }
      }
      stamp(k, 3);
      lds_barrier();                              // B2: derived planes visible
      stamp(k, 4);
    }
    // ---------------- MFMA (compute waves) ----------------
    f32x4 acc_a[2], acc_b[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      acc_a[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc_b[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    uint32_t so_bits = 0;
    if (!is_loader && J.b >= 0 && !(A.pool_sel & 4)) {
      const u16* x_hi = j_pool ? der : r_hi;
      const u16* x_lo = j_pool ? der + (pbytes >> 1) : r_lo;
      const u16* x_sq = der + pbytes + (j_pool ? (pbytes >> 1) : 0);
      const uint32_t* sg = sgn + slot * 64 + J.b * 8;
      if constexpr (EM == EM_FLIPOUT) so_bits = sg[4 + (e_sob >> 5)] >> (e_sob & 31);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < j_nks) {
          u32x4 fm = {0u, 0u, 0u, 0u};
          if constexpr (EM == EM_FLIPOUT) {
            const uint32_t byte = (sg[k_sb[ks] & 0xff] >> (k_sb[ks] >> 8)) & 0xffu;
            fm = __builtin_bit_cast(u32x4, lut[byte]);
          }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const int o = k_o[ks] + mt * 16 * RS;
            const bf16x8 bh = *(const bf16x8*)&x_hi[o];
            const bf16x8 bl = *(const bf16x8*)&x_lo[o];
            acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[ks], bh, acc_a[mt], 0, 0, 0);
            acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[ks], bl, acc_a[mt], 0, 0, 0);
            acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo[ks], bh, acc_a[mt], 0, 0, 0);
            if constexpr (LRT) {
              const bf16x8 b2 = *(const bf16x8*)&x_sq[o];
              acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_b[ks], b2, acc_b[mt], 0, 0, 0);
            } else if constexpr (EM == EM_FLIPOUT) {
              const u32x4 xb = __builtin_bit_cast(u32x4, bh) ^ fm;
              acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_b[ks], __builtin_bit_cast(bf16x8, xb), acc_b[mt], 0,
                                                                 0, 0);
            }
          }
        }
      }
    }
    stamp(k, 5);
    // ---------------- K-split reduction ----------------
    if (n_red_groups > 0) {
      // (the lane index is made opaque here and in the epilogue: offsets derived from it are recomputed per window
      // instead of being kept in registers across the loop - the vector file is full)
      int rl = lane;
      asm volatile("" : "+v"(rl));
      if (!is_loader && J.b >= 0 && J.grp >= 0 && !J.owner) {
        float* r = red + (size_t)wave * (2 * 2 * 256);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          *(f32x4*)&r[(mt * 2 + 0) * 256 + rl * 4] = acc_a[mt];
          if constexpr (DUAL) *(f32x4*)&r[(mt * 2 + 1) * 256 + rl * 4] = acc_b[mt];
        }
      }
      lds_barrier();
      if (!is_loader && J.b >= 0 && J.grp >= 0 && J.owner) {
        for (int m = 1; m < J.nmember; ++m) {
          const float* r = red + (size_t)(wave + m) * (2 * 2 * 256);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const f32x4 pa = *(const f32x4*)&r[(mt * 2 + 0) * 256 + rl * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc_a[mt][q] += pa[q];
            if constexpr (DUAL) {
              const f32x4 pb = *(const f32x4*)&r[(mt * 2 + 1) * 256 + rl * 4];
#pragma unroll
              for (int q = 0; q < 4; ++q) acc_b[mt][q] += pb[q];
            }
          }
        }
      }
    }
    stamp(k, 6);
    // ---------------- epilogue ----------------
    if (!is_loader && J.b >= 0 && e_nv > 0 && (J.grp < 0 || J.owner) && !(A.pool_sel & 1)) {
      const int w = s * B + win_of(k);
      int lch = e_lch, ooff = e_ooff;
      asm volatile("" : "+v"(lch), "+v"(ooff));
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        if (row >= L) continue;
        const int R = w * L + row;
        f32x4 v = acc_a[mt];
        f32x4 qv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LRT) {
          f32x4 eps;
          if (A.nz.use_philox_lrt) {
            const long Rg = global_row(A.cg, L, R);
            const uint64_t idx = (uint64_t)Rg * (uint64_t)e_c4n + (uint64_t)(lch >> 2);
            eps = philox_normal4((uint32_t)idx, (uint32_t)(idx >> 32), NK_LRT | ((uint32_t)e_layer << 8), A.nz.step,
                                 A.nz.seed);
          } else {
            const float* e = A.nz.lrt_eps[e_layer] + (long)R * e_cout + lch;
#pragma unroll
            for (int r = 0; r < 4; ++r) eps[r] = (r < e_nv) ? e[r] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float loc = v[r] + e_ba[r];
            float var = acc_b[mt][r] + e_bb[r];
            if (var < 0.f) var = 1e-6f;
            const float sd = sqrtf(var);
            v[r] = loc + sd * eps[r];
            qv[r] = sd > 0.f ? eps[r] / (2.f * sd) : 0.f;
          }
        } else if constexpr (EM == EM_FLIPOUT) {
          const uint32_t bits = so_bits >> (4 * g4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pb = acc_b[mt][r];
            v[r] = v[r] + e_ba[r] + (((bits >> r) & 1u) ? -pb : pb);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += e_ba[r];
        }
        if (e_relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const long oo = (long)R * e_octot + ooff;
        // channels beyond cout inside the 4-group are channel pads of the output tensor (zero
        // weights and bias -> exact zeros), so the whole group is always stored
        uint2 hv, lv;
        split4(v, hv, lv);
        *(uint2*)(e_ohi + oo) = hv;
        *(uint2*)(e_olo + oo) = lv;
        if constexpr (LRT) *(uint2*)(e_q + oo) = make_uint2(cvt_pk(qv[0], qv[1]), cvt_pk(qv[2], qv[3]));
      }
    }
    stamp(k, 7);
  }
}

// chunk position of channel chunk c8 in image row r of a dense image with cb8 chunks per row
__device__ __forceinline__ int swz(int c8, int r, int cb8) {
  return c8 ^ ((cb8 >= 16 ? r : (r / (16 / cb8))) & (cb8 - 1));
}

enum { DN_CH = 128, DN_ROWS = 32 };   // dense layers: channels per K chunk, example rows per window

// ==========================================================================================
// dense_dx_bf_kernel : dX of a dense layer (K = cout <= 64 is tiny, the output is wide).
// One workgroup = one 32-row window; dz (= dY [Y>0]) and its second image (LRT: dz*q, Flipout:
// dz*s_out per row) are staged once into LDS; each of the 8 waves then walks its own 16-channel
// output tiles in batches of 4 (weights from L2, no barrier in the tile loop).
// ==========================================================================================
enum { DDX_WAVES = 8 };   // 8 waves: 256 VGPRs each, room for a batch of 4 tiles' fragments
template <int EM>
__global__ __launch_bounds__(DDX_WAVES * 64) void dense_dx_bf_kernel(const GroupArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const BranchDesc& br = G.br[0];
  const LayerDesc ly = A.layers[br.layer];   // device-resident table: read once, not inside the tile loop
  const Win W = decode_win(G, A.cg, blockIdx.x);
  const int s = W.s;
  const uint32_t* sgi = A.nz.sign_in + ly.sign_in_off * A.nz.examples + (long)W.ex0 * ly.sign_in_words;   // [row][words]
  const int sgi_stride = ly.sign_in_words;
  const int zw = (br.cout + 31) & ~31;          // K of the transposed contraction (multiple of 32)
  const int RS = zw + 8;                        // (zw/8 is even) -> conflict-free b128 rows
  u16* dz = (u16*)smem;
  u16* dz2 = dz + DN_ROWS * RS;
  const TensorRef tg = A.t[br.out_t + T_GRAD], ty = A.t[br.out_t], tq = A.t[br.q_t];
  const TensorRef tin = A.t[G.in_t], tdx = A.t[br.dx_t];
  // ---- stage dz / dz2 (4 channels per unit) ----
  for (int U = tid; U < DN_ROWS * (zw >> 2); U += DDX_WAVES * 64) {
    const int row = U / (zw >> 2), c = (U - row * (zw >> 2)) * 4;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f};
    if (row < W.nvalid && c < br.cout) {
      const long o = (long)(W.out_row0 + row) * tg.ctot + br.out_off + c;
      const int nv = br.cout - c;
      const bool vec = ((tg.ctot & 3) == 0) && ((br.out_off & 3) == 0);
      g = tload4(tg, o, nv, vec);
      if (br.relu) {
        const f32x4 y = tload4(ty, o, nv, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.f ? g[k] : 0.f;
      }
      if constexpr (LRT) {
        const f32x4 q = tload4(tq, o, nv, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) g2[k] = g[k] * q[k];
      } else if constexpr (EM == EM_FLIPOUT) {
        const int bit0 = br.n_off + c;
        const uint32_t word = A.nz.sign_out[ly.sign_out_off * A.nz.examples + (long)(W.ex0 + row) * ly.sign_out_words + (bit0 >> 5)];
        const uint32_t bits = word >> (bit0 & 31);
#pragma unroll
        for (int k = 0; k < 4; ++k) g2[k] = ((bits >> k) & 1u) ? -g[k] : g[k];
      }
    }
    *(uint2*)&dz[row * RS + c] = pack_bf4(g);
    if constexpr (DUAL) *(uint2*)&dz2[row * RS + c] = pack_bf4(g2);
  }
  __syncthreads();
  const int i16 = lane & 15, g4 = lane >> 4;
  const int nks = zw >> 5;                       // 1 or 2 k-steps
  // B fragments of the window (both m-tiles, all k-steps) stay in registers for every tile
  bf16x8 bz[2][2], bz2[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      bz[ks][mt] = bz2[ks][mt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (ks < nks) {
        const int o = (mt * 16 + i16) * RS + ks * 32 + g4 * 8;
        bz[ks][mt] = *(const bf16x8*)&dz[o];
        if constexpr (DUAL) bz2[ks][mt] = *(const bf16x8*)&dz2[o];
      }
    }
  const int ntile = br.cin_p >> 4;
  const long sa = A.ws.slott_stride_a * s, sb = A.ws.slott_stride_b * s;
  const u16* wat = (const u16*)A.ws.at + sa + ly.wt_off;
  const u16* wbt = (const u16*)A.ws.bt + sb + ly.wt_off;
  const int KPt = ly.KPt;
  // Tiles in batches of TB: all weight fragments of a batch are fetched first, then its MFMAs and stores.  A
  // wave that loads and stores shares ONE in-order vmcnt, so the wait for the next batch's fragments also waits
  // for this batch's stores; paying that round trip once per TB tiles instead of once per tile is the point.
  // A wave owns TB CONSECUTIVE tiles (its stores of a row then cover 128 contiguous bytes), and the group order is
  // rotated by the window index so that the workgroups of a particle do not walk the same weight rows in lockstep.
  constexpr int TB = 4;
  const int ngrp = (ntile + TB - 1) / TB;
  const int rot = (int)(blockIdx.x % (unsigned)ngrp);
  for (int g0 = wave; g0 < ngrp; g0 += DDX_WAVES) {
    int grp = g0 + rot;
    if (grp >= ngrp) grp -= ngrp;
    const int t0 = grp * TB;
    bf16x8 wa[TB][2], wb[TB][2];
    uint32_t sw[TB][2];   // flipout: sign_in word of (row, 4 channels of this lane), per m-tile
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wa[j][ks] = wb[j][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (t < ntile && ks < nks) {
          const long wo = (long)(t * 16 + i16) * KPt + br.n_off + ks * 32 + g4 * 8;
          wa[j][ks] = *(const bf16x8*)(wat + wo);
          if constexpr (DUAL) wb[j][ks] = *(const bf16x8*)(wbt + wo);
        }
      }
      if constexpr (EM == EM_FLIPOUT) {
        const int och = t * 16 + 4 * g4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = min(mt * 16 + i16, W.nvalid - 1);
          sw[j][mt] = 0u;
          if (t < ntile) sw[j][mt] = sgi[(long)row * sgi_stride + (och >> 5)] >> (och & 31);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j;
      if (t >= ntile) break;
      const int c0 = t * 16;
      f32x4 acc_a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 acc_b[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks < nks) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc_a[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][ks], bz[ks][mt], acc_a[mt], 0, 0, 0);
          if constexpr (DUAL) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc_b[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][ks], bz2[ks][mt], acc_b[mt], 0, 0, 0);
          }
        }
      }
      const int och = c0 + 4 * g4;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + i16;
        if (row >= W.nvalid) continue;
        f32x4 v = acc_a[mt];
        if constexpr (LRT) {
          f32x4 xv = tload4(tin, (long)(W.in_row0 + row) * tin.ctot + br.in_off + och, 4, true);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += 2.f * bf2f(f2bf(xv[r])) * acc_b[mt][r];
        } else if constexpr (EM == EM_FLIPOUT) {
          const uint32_t bits = sw[j][mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += ((bits >> r) & 1u) ? -acc_b[mt][r] : acc_b[mt][r];
        }
        tstore4(tdx, (long)(W.in_row0 + row) * tdx.ctot + br.in_off + och, v, 4, true);
      }
    }
  }
}

// ==========================================================================================
// dense_dw_bf_kernel : dW of a dense layer.  One workgroup = (particle, 128-channel chunk of the
// input, split of the particle's 32-row windows); its 32 (x2) dW tiles stay in registers while the
// windows stream through LDS (register-prefetched one window ahead).  Flipout signs are per
// example row here, so the sign-multiplied copies (X*s_in, dz*s_out) are explicit images.
// ==========================================================================================
template <int EM>
__global__ __launch_bounds__(512) void dense_dw_bf_kernel(const GroupArgs A, int nchunk, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const BranchDesc& br = G.br[0];
  const LayerDesc& ly = A.layers[br.layer];
  int bid = blockIdx.x;
  const int split = bid % nsplit; bid /= nsplit;
  const int chunk = bid % nchunk;
  const int s = bid / nchunk;
  const int c0 = chunk * DN_CH;
  const int cw = min(DN_CH, br.cin_p - c0);          // multiple of 16
  const int zw = (br.cout + 15) & ~15;               // <= 64
  constexpr int RSX = DN_CH + 8, RSZ = 64 + 8;
  u16* xi = (u16*)smem;
  u16* x2 = xi + DN_ROWS * RSX;
  u16* dz = x2 + DN_ROWS * RSX;
  u16* dz2 = dz + DN_ROWS * RSZ;
  const TensorRef tin = A.t[G.in_t], tg = A.t[br.out_t + T_GRAD], ty = A.t[br.out_t], tq = A.t[br.q_t];
  const u16* g_x = (const u16*)tin.p;
  const int pp = A.cg.per_particle, B = A.cg.B;
  // staging plan: X unit = thread (row = tid >> 4, c8 = tid & 15); dz unit = thread < 256 (row = tid >> 3, c8 = tid & 7)
  const int xr = tid >> 4, xc8 = tid & 15;
  const bool x_on = xc8 * 8 < cw;
  const int zr = tid >> 3, zc8 = tid & 7;
  const bool z_on = tid < 256 && zc8 * 8 < br.cout;
  uint4 px = make_uint4(0, 0, 0, 0), pz = make_uint4(0, 0, 0, 0), py = make_uint4(0, 0, 0, 0), pq = make_uint4(0, 0, 0, 0);
  uint32_t psi = 0, pso = 0;
  int p_nvalid = 0;
  auto prefetch = [&](int wl) {
    const int row0 = s * B + wl * DN_ROWS;
    const int nvalid = min(DN_ROWS, B - wl * DN_ROWS);
    p_nvalid = nvalid;
    px = pz = py = pq = make_uint4(0, 0, 0, 0);
    psi = pso = 0;
    if (x_on && xr < nvalid) {
      px = *(const uint4*)(g_x + (long)(row0 + xr) * tin.ctot + br.in_off + c0 + xc8 * 8);
      if constexpr (EM == EM_FLIPOUT) {
        const int bit0 = c0 + xc8 * 8;
        psi = A.nz.sign_in[ly.sign_in_off * A.nz.examples + (long)(row0 + xr) * ly.sign_in_words + (bit0 >> 5)] >> (bit0 & 31);
      }
    }
    if (z_on && zr < nvalid) {
      const long o = (long)(row0 + zr) * tg.ctot + br.out_off + zc8 * 8;
      pz = *(const uint4*)((const u16*)tg.p + o);
      if (br.relu) py = *(const uint4*)((const u16*)ty.p + o);
      if constexpr (LRT) pq = *(const uint4*)((const u16*)tq.p + o);
      if constexpr (EM == EM_FLIPOUT) {
        const int bit0 = br.n_off + zc8 * 8;
        pso = A.nz.sign_out[ly.sign_out_off * A.nz.examples + (long)(row0 + zr) * ly.sign_out_words + (bit0 >> 5)] >> (bit0 & 31);
      }
    }
  };
  // tiles: t = wave + 8 m, (nt, ct) = (t / 8, t % 8)
  f32x4 acc_a[4], acc_b[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    acc_a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gb_a = 0.f, gb_b = 0.f;
  const int ntl = zw >> 4, ctl = (cw + 15) >> 4;
  auto sgn8 = [](uint4 v, uint32_t bits) {   // flip the sign of bf16 element e where bit e is set
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] ^= (((bits >> (2 * e)) & 1u) << 15) | (((bits >> (2 * e + 1)) & 1u) << 31);
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  auto mul8 = [](uint4 a, uint4 b) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float l = bf2f((u16)(x[e] & 0xffff)) * bf2f((u16)(y[e] & 0xffff));
      const float h = bf2f((u16)(x[e] >> 16)) * bf2f((u16)(y[e] >> 16));
      o[e] = (uint32_t)f2bf(l) | ((uint32_t)f2bf(h) << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  int wl = split;
  if (wl < pp) prefetch(wl);
  const int gq = lane >> 4, qq = (lane >> 2) & 3, pq4 = lane & 3;
  for (; wl < pp; wl += nsplit) {
    __syncthreads();
    // ---- images of this window ----
    {
      *(uint4*)&xi[xr * RSX + xc8 * 8] = px;
      if constexpr (LRT) *(uint4*)&x2[xr * RSX + xc8 * 8] = mul8(px, px);
      if constexpr (EM == EM_FLIPOUT) *(uint4*)&x2[xr * RSX + xc8 * 8] = sgn8(px, psi);
      if (tid < 256) {
        uint4 g = pz;
        if (br.relu) {
          auto msk = [](uint32_t yy) {
            const uint32_t lo = ((yy & 0x8000u) == 0 && (yy & 0x7fffu) != 0) ? 0xffffu : 0u;
            const uint32_t hi = ((yy & 0x80000000u) == 0 && (yy & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
            return lo | hi;
          };
          g.x &= msk(py.x); g.y &= msk(py.y); g.z &= msk(py.z); g.w &= msk(py.w);
        }
        *(uint4*)&dz[zr * RSZ + zc8 * 8] = g;
        if constexpr (LRT) *(uint4*)&dz2[zr * RSZ + zc8 * 8] = mul8(g, pq);
        if constexpr (EM == EM_FLIPOUT) *(uint4*)&dz2[zr * RSZ + zc8 * 8] = sgn8(g, pso);
      }
    }
    __syncthreads();
    if (wl + nsplit < pp) prefetch(wl + nsplit);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int t = wave + 8 * m;
      const int nt = t >> 3, ct = t & 7;
      if (nt < ntl && ct < ctl) {
        const int r0 = 8 * gq + qq;
        const u16* a0 = &dz[r0 * RSZ + nt * 16 + 4 * pq4];
        const u16* b0 = &xi[r0 * RSX + ct * 16 + 4 * pq4];
        acc_a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a0, a0 + 4 * RSZ), tr_frag(b0, b0 + 4 * RSX), acc_a[m], 0, 0, 0);
        if constexpr (DUAL) {
          const u16* a2 = &dz2[r0 * RSZ + nt * 16 + 4 * pq4];
          const u16* b2 = &x2[r0 * RSX + ct * 16 + 4 * pq4];
          acc_b[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a2, a2 + 4 * RSZ), tr_frag(b2, b2 + 4 * RSX), acc_b[m], 0, 0, 0);
        }
      }
    }
    if (chunk == 0 && tid < br.cout) {
      float sa = 0.f, sb = 0.f;
      for (int r = 0; r < DN_ROWS; ++r) {
        sa += bf2f(dz[r * RSZ + tid]);
        // bias gradients: LRT needs sum(dz*q); Flipout's bias gradient arrives through slot A only
        if constexpr (LRT) sb += bf2f(dz2[r * RSZ + tid]);
      }
      gb_a += sa;
      gb_b += sb;
    }
  }
  float* gwa = A.gw_a + A.gw_stride * s + ly.w_off;
  float* gwb = A.gw_b + A.gw_stride * s + ly.w_off;
  const int i4 = 4 * (lane >> 4), jc = lane & 15;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int t = wave + 8 * m;
    const int nt = t >> 3, ct = t & 7;
    if (nt >= ntl || ct >= ctl) continue;
    const int c = c0 + ct * 16 + jc;
    if (c >= br.cin_p) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      const long o = (long)(br.n_off + n) * ly.KP + c;
      atomicAdd(gwa + o, acc_a[m][r]);
      if constexpr (DUAL) atomicAdd(gwb + o, acc_b[m][r]);
    }
  }
  if (chunk == 0 && tid < br.cout) {
    atomicAdd(A.gb_a + (long)A.gb_stride * s + ly.bias_off + br.n_off + tid, gb_a);
    if constexpr (LRT) atomicAdd(A.gb_b + (long)A.gb_stride * s + ly.bias_off + br.n_off + tid, gb_b);
  }
}

// ==========================================================================================
// conv_dw_mw_kernel : dW of a conv group, NWI windows per iteration.
// The per-iteration costs (3 workgroup barriers, the exposed part of the global-load latency)
// are amortised over NWI windows; all loads of an iteration are issued back to back (16-byte
// units straight from the bf16 planes), so the memory pipe sees NWI windows of traffic at once.
// Tiles (all branches of the group) stay in registers across the windows of the workgroup.
// ==========================================================================================
template <int EM, int MAXT, int NWI, int NWV>
__global__ __launch_bounds__(NWV * 64) void conv_dw_mw_kernel(const GroupArgs A, const ConvDwPlan D) {
  constexpr int CV_THREADS = NWV * 64, CV_WAVES = NWV;   // shadows the 8-wave defaults
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DUAL = (EM != EM_PLAIN);
  constexpr bool LRT = (EM == EM_LRT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GroupDesc& G = A.g;
  const int s = blockIdx.x / D.nsplit, split = blockIdx.x - s * D.nsplit;
  const int L = G.L, B = A.cg.B;
  const int cwp = G.in_cin_p;
  const int xw16 = (cwp + 15) & ~15;
  const int RSx = img_row_stride(xw16, true), RSz = img_row_stride(D.zw, true);
  const int xbytes = (IMG_ROWS * RSx * 2 + 15) & ~15, zbytes = (IMG_ROWS * RSz * 2 + 15) & ~15;
  // per window: x_hi | [xp] | dz | [x_sq | xp_sq | dz2]
  const int o_xp = xbytes;
  const int o_dz = o_xp + (D.has_pool ? xbytes : 0);
  const int o_xsq = o_dz + zbytes;
  const int o_xpsq = o_xsq + (LRT ? xbytes : 0);
  const int o_dz2 = o_xpsq + ((LRT && D.has_pool) ? xbytes : 0);
  const int o_sg = o_dz2 + (LRT ? zbytes : 0);              // flipout sign words [branch][8]
  const int wbytes = o_sg + ((EM == EM_FLIPOUT) ? 128 : 0);
  {
    const int total = (NWI * wbytes) >> 2;
    uint32_t* z = (uint32_t*)smem;
    for (int k = tid; k < total; k += CV_THREADS) z[k] = 0u;
  }
  const TensorRef tin = A.t[G.in_t];
  const int x8 = cwp >> 3, z8 = D.zw >> 3;
  const int xunits = L * x8, zunits = L * z8;

  // ---- per-thread staging plan: <= 1 X unit and <= 2 dz units per window (16 bytes each) ----
  const bool x_on = tid < xunits;
  const int x_row = tid / x8, x_c = (tid - x_row * x8) * 8;
  const int x_src = x_row * tin.ctot + x_c, x_dst = (x_row + HALO) * RSx + x_c;
  int z_src[2], z_dst[2], z_ct[2];
  const u16 *z_g[2], *z_y[2], *z_q[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int unit = tid + u * CV_THREADS;
    z_src[u] = z_dst[u] = z_ct[u] = 0;
    z_g[u] = z_y[u] = z_q[u] = nullptr;
    if (unit < zunits) {
      const int row = unit / z8, zc = (unit - row * z8) * 8;
      int b = 0;
      for (int k = 1; k < G.n_branch; ++k)
        if (zc >= D.zoff[k]) b = k;
      const BranchDesc& br = G.br[b];
      const int c = zc - D.zoff[b];
      if (c < br.cout) {   // 8-channel units: the tensors' channel pads are zero and in bounds
        const TensorRef tg = A.t[br.out_t + T_GRAD];
        z_ct[u] = tg.ctot;
        z_src[u] = row * tg.ctot + br.out_off + c;
        z_dst[u] = (row + HALO) * RSz + zc;
        z_g[u] = (const u16*)tg.p;
        z_y[u] = br.relu ? (const u16*)A.t[br.out_t].p : nullptr;
        z_q[u] = LRT ? (const u16*)A.t[br.q_t].p : nullptr;
      }
    }
  }
  // ---- tiles ----
  f32x4 acc_a[MAXT], acc_b[MAXT];
  int t_a[MAXT], t_b[MAXT], t_so[MAXT], t_si[MAXT];   // t_so / t_si: LDS sign word index | bit of lane 0 << 8
  bool t_ok[MAXT];
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    acc_a[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t = wave + CV_WAVES * m;
    t_ok[m] = t < D.ntiles;
    t_a[m] = t_b[m] = t_so[m] = t_si[m] = 0;
    if (t_ok[m]) {
      const DwTile T = D.tile[t];
      const BranchDesc& br = G.br[T.b];
      const LayerDesc& ly = A.layers[br.layer];
      const int n0 = D.zoff[T.b] + T.nt * 16;
      const int c0 = br.in_off + T.ct * 16;
      const int tshift = T.tap - ly.pad;
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const int r0 = 8 * g + q;
      t_a[m] = (o_dz >> 1) + (r0 + HALO) * RSz + n0 + 4 * p;
      t_b[m] = (br.pool ? (o_xp >> 1) : 0) + (r0 + tshift + HALO) * RSx + c0 + 4 * p;
      const int nbit = br.n_off + T.nt * 16, cbit = T.ct * 16;
      t_so[m] = (T.b * 8 + 4 + (nbit >> 5)) | ((nbit & 31) << 8);
      t_si[m] = (T.b * 8 + (cbit >> 5)) | ((cbit & 31) << 8);
    }
  }
  // bias gradients: column sums of dz through one MFMA against an all-ones B fragment; wave w owns
  // the 16-channel blocks w, w+8 of the concatenated dz image
  f32x4 acc_ba[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 acc_bb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int nzt = D.zw >> 4;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  // flipout sign words of a window: thread tid < 8 * n_branch fetches word (branch, k)
  const uint32_t* sg_src = nullptr;
  long sg_stride = 0;
  if (EM == EM_FLIPOUT && tid < 8 * G.n_branch) {
    const int b = tid >> 3, k = tid & 7;
    const LayerDesc& ly = A.layers[G.br[b].layer];
    if (k < 4 && k < ly.sign_in_words) {
      sg_src = A.nz.sign_in + ly.sign_in_off * A.nz.examples + k;
      sg_stride = ly.sign_in_words;
    } else if (k >= 4 && k < 6 && k - 4 < ly.sign_out_words) {
      sg_src = A.nz.sign_out + ly.sign_out_off * A.nz.examples + (k - 4);
      sg_stride = ly.sign_out_words;
    }
  }
  const int pp = B;
  const int my_nwin = (pp - split + D.nsplit - 1) / D.nsplit;
  const u16* g_x = (const u16*)tin.p;
  auto msk = [](uint32_t yy) {   // bf16 > 0
    const uint32_t lo = ((yy & 0x8000u) == 0 && (yy & 0x7fffu) != 0) ? 0xffffu : 0u;
    const uint32_t hi = ((yy & 0x80000000u) == 0 && (yy & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
    return lo | hi;
  };
  auto sq8 = [](uint4 a) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float l = bf2f((u16)(x[e] & 0xffff)), h = bf2f((u16)(x[e] >> 16));
      o[e] = (uint32_t)f2bf(l * l) | ((uint32_t)f2bf(h * h) << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  auto mul8 = [](uint4 a, uint4 b) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float l = bf2f((u16)(x[e] & 0xffff)) * bf2f((u16)(y[e] & 0xffff));
      const float h = bf2f((u16)(x[e] >> 16)) * bf2f((u16)(y[e] >> 16));
      o[e] = (uint32_t)f2bf(l) | ((uint32_t)f2bf(h) << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  auto max8 = [](uint4 a, uint4 b) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float l = fmaxf(bf2f((u16)(x[e] & 0xffff)), bf2f((u16)(y[e] & 0xffff)));
      const float h = fmaxf(bf2f((u16)(x[e] >> 16)), bf2f((u16)(y[e] >> 16)));
      o[e] = (uint32_t)f2bf(l) | ((uint32_t)f2bf(h) << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };

  BNN_STAMP_DECL(A);
  // registers of one iteration's loads; the NEXT iteration's loads are issued right after the staging
  // stores, so their latency hides behind the pooled-copy pass and the MFMA phase
  uint4 px[NWI], pz[NWI][2], py[NWI][2], pq[NWI][2];
  uint32_t psg[NWI];
  // running element offsets of window k0 of this workgroup (advanced by NWI windows per call): the loads of an
  // iteration cost a handful of adds, not 64-bit multiplies per pointer (1,400 cycles per call before)
  const long w_first = (long)s * B + split;
  long o_x = (G.in_bcast ? (long)split : w_first) * L * tin.ctot + x_src;
  const long st_x = (long)D.nsplit * L * tin.ctot;
  long o_z[2] = {w_first * L * z_ct[0] + z_src[0], w_first * L * z_ct[1] + z_src[1]};
  const long st_z[2] = {(long)D.nsplit * L * z_ct[0], (long)D.nsplit * L * z_ct[1]};
  long r_sg = w_first * sg_stride;
  const long st_sg = (long)D.nsplit * sg_stride;
  auto load_iter = [&](int k0) {   // must be called for k0 = 0, NWI, 2 NWI, ... in order
    const int nw = min(NWI, my_nwin - k0);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      px[i] = make_uint4(0, 0, 0, 0);
      psg[i] = 0u;
#pragma unroll
      for (int u = 0; u < 2; ++u) pz[i][u] = py[i][u] = pq[i][u] = make_uint4(0, 0, 0, 0);
      if (i < nw && !(A.pool_sel & 8)) {
        if (x_on) px[i] = *(const uint4*)(g_x + o_x + i * st_x);
        if (sg_src) psg[i] = sg_src[r_sg + i * st_sg];
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (z_g[u]) {
            const long o = o_z[u] + i * st_z[u];
            pz[i][u] = *(const uint4*)(z_g[u] + o);
            if (z_y[u]) py[i][u] = *(const uint4*)(z_y[u] + o);
            if constexpr (LRT) pq[i][u] = *(const uint4*)(z_q[u] + o);
          }
      }
    }
    o_x += NWI * st_x;
    r_sg += NWI * st_sg;
    o_z[0] += NWI * st_z[0];
    o_z[1] += NWI * st_z[1];
  };
  // 11 tiles per wave leave no registers for the early loads of TWO windows (they would spill): one window per
  // iteration there, or the loads at the top
  constexpr bool PF = MAXT < 11 || NWI == 1;
  if (PF && my_nwin > 0) load_iter(0);
  for (int k0 = 0; k0 < my_nwin; k0 += NWI) {
    const int nw = min(NWI, my_nwin - k0);
    const int kst = k0 / NWI;
    stamp(kst, 0);
    if (!PF) load_iter(k0);
    stamp(kst, 1);
    lds_barrier();   // raw s_barrier: __syncthreads() would also drain the prefetched global loads (vmcnt(0))   // previous iteration's images consumed
    stamp(kst, 2);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      u16* base = (u16*)(smem + i * wbytes);
      if (x_on) {
        *(uint4*)&base[x_dst] = px[i];
        if constexpr (LRT) *(uint4*)&base[(o_xsq >> 1) + x_dst] = sq8(px[i]);
      }
      if constexpr (EM == EM_FLIPOUT) {
        if (tid < 32) ((uint32_t*)((char*)base + o_sg))[tid] = psg[i];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (z_g[u]) {
          uint4 g = pz[i][u];
          if (z_y[u]) {
            g.x &= msk(py[i][u].x); g.y &= msk(py[i][u].y); g.z &= msk(py[i][u].z); g.w &= msk(py[i][u].w);
          }
          *(uint4*)&base[(o_dz >> 1) + z_dst[u]] = g;
          if constexpr (LRT) *(uint4*)&base[(o_dz2 >> 1) + z_dst[u]] = mul8(g, pq[i][u]);
        }
    }
    stamp(kst, 3);
    if (PF && k0 + NWI < my_nwin) load_iter(k0 + NWI);
    lds_barrier();   // raw s_barrier: __syncthreads() would also drain the prefetched global loads (vmcnt(0))
    stamp(kst, 4);
    if (D.has_pool && !(A.pool_sel & 2)) {
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        u16* base = (u16*)(smem + i * wbytes);
        if (x_on) {
          uint4 m = *(const uint4*)&base[x_dst];
          if (x_row > 0) m = max8(m, *(const uint4*)&base[x_dst - RSx]);
          if (x_row + 1 < L) m = max8(m, *(const uint4*)&base[x_dst + RSx]);
          *(uint4*)&base[(o_xp >> 1) + x_dst] = m;
          if constexpr (LRT) *(uint4*)&base[(o_xpsq >> 1) + x_dst] = sq8(m);
        }
      }
      lds_barrier();   // raw s_barrier: __syncthreads() would also drain the prefetched global loads (vmcnt(0))
    }
    stamp(kst, 5);
    // ---- tiles ----
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      if (i < nw) {
        const u16* base = (const u16*)(smem + i * wbytes);
        const uint32_t* sg = (const uint32_t*)((const char*)base + o_sg);
#pragma unroll
        for (int m = 0; m < MAXT; ++m) {
          if (t_ok[m] && !(A.pool_sel & 1)) {
            const u16* a0 = base + t_a[m];
            const u16* b0 = base + t_b[m];
            const bf16x8 fa = tr_frag(a0, a0 + 4 * RSz);
            const bf16x8 fb = tr_frag(b0, b0 + 4 * RSx);
            acc_a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc_a[m], 0, 0, 0);
            if constexpr (LRT) {
              const u16* a2 = a0 + ((o_dz2 - o_dz) >> 1);
              const u16* b2 = b0 + ((o_xsq) >> 1);   // x_sq follows x_hi, xp_sq follows xp at the same distance
              acc_b[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a2, a2 + 4 * RSz), tr_frag(b2, b2 + 4 * RSx),
                                                                 acc_b[m], 0, 0, 0);
            } else if constexpr (EM == EM_FLIPOUT) {
              const bool no = (sg[t_so[m] & 0xff] >> ((t_so[m] >> 8) + (lane & 15))) & 1u;
              const bool ni = (sg[t_si[m] & 0xff] >> ((t_si[m] >> 8) + (lane & 15))) & 1u;
              acc_b[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xor_sign(fa, no), xor_sign(fb, ni), acc_b[m], 0, 0, 0);
            }
          }
        }
        if (!(A.pool_sel & 4)) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int zt = wave + CV_WAVES * q;
            if (zt < nzt) {
              const int g = lane >> 4, qq = (lane >> 2) & 3, p = lane & 3;
              const u16* a0 = base + (o_dz >> 1) + (8 * g + qq + HALO) * RSz + zt * 16 + 4 * p;
              acc_ba[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a0, a0 + 4 * RSz), ones, acc_ba[q], 0, 0, 0);
              if constexpr (LRT) {
                const u16* a2 = a0 + ((o_dz2 - o_dz) >> 1);
                acc_bb[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(a2, a2 + 4 * RSz), ones, acc_bb[q], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    stamp(kst, 6);
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
    const bool slab = D.slab_a != nullptr;
    float* gwa = (slab ? D.slab_a + D.slab_stride * blockIdx.x : A.gw_a + A.gw_stride * s) + ly.w_off;
    float* gwb = (slab ? D.slab_b + D.slab_stride * blockIdx.x : A.gw_b + A.gw_stride * s) + ly.w_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = T.nt * 16 + i4 + r;
      if (n >= br.cout) continue;
      const long o = (long)(br.n_off + n) * ly.KP + (long)T.tap * ly.cin_img + c;
      if (slab) {
        gwa[o] = acc_a[m][r];
        if constexpr (DUAL) gwb[o] = acc_b[m][r];
      } else {
        atomicAdd(gwa + o, acc_a[m][r]);
        if constexpr (DUAL) atomicAdd(gwb + o, acc_b[m][r]);
      }
    }
  }
  if ((lane & 15) == 0 && !D.no_bias) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int zt = wave + CV_WAVES * q;
      if (zt >= nzt) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int zc = zt * 16 + 4 * (lane >> 4) + r;
        int b = 0;
        for (int k = 1; k < G.n_branch; ++k)
          if (zc >= D.zoff[k]) b = k;
        const BranchDesc& br = G.br[b];
        const int n = zc - D.zoff[b];
        if (n < br.cout) {
          const LayerDesc& ly = A.layers[br.layer];
          atomicAdd(A.gb_a + (long)A.gb_stride * s + ly.bias_off + br.n_off + n, acc_ba[q][r]);
          if constexpr (LRT) atomicAdd(A.gb_b + (long)A.gb_stride * s + ly.bias_off + br.n_off + n, acc_bb[q][r]);
        }
      }
    }
  }
}
