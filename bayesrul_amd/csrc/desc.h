// Plain-data descriptors shared by the host plan (plan.hip) and the kernels (kernels.hip).
// Vocabulary: a *window* is one N-CMAPSS sequence window (W rows x C channels, channels-last);
// dense layers see the [S*B] example rows chopped into 32-row windows.  A *group* is a set of
// branches (variational Conv1d / Linear layers) that read the same input tensor.
#pragma once
#include <stdint.h>

enum { BNN_MAX_BRANCH = 4, BNN_MAX_TENSORS = 32, BNN_MAX_LAYERS = 12, BNN_MAX_SITES = 24 };
// tensor ids: activation t in [0,10); its gradient at t + T_GRAD; its LRT q at t + T_Q;
// T_POOLGRAD: gradient w.r.t. the max-pooled copy of an activation (scattered by pool_bwd)
enum { T_X = 9, T_GRAD = 10, T_Q = 20, T_POOLGRAD = 30 };
enum { DENSE_CHUNK = 128 };  // channels per K-chunk of a dense layer (and of a dx output chunk)

// estimator of a contraction (kernel template parameter)
enum { EM_PLAIN = 0, EM_LRT = 1, EM_FLIPOUT = 2 };

// fmt 0: fp32 rows in `p`.  fmt 1: bf16 planes, `p` = hi plane, `lo` = lo plane (value = hi + lo;
// nullptr: single bf16 plane).  Row stride = ctot elements in every plane.
enum { TF_F32 = 0, TF_BF16 = 1 };
struct TensorRef {
  void* p;
  void* lo;
  int ctot;  // channels per row (row stride, elements)
  int fmt;
};

// How a layer's canonical input-channel index maps to the image channel index of its input
// tensor.  CM_BLOCK: groups of `a` channels stored with stride `b` (27 -> 28 padding of
// block-1 branch outputs).  CM_FLATTEN: canonical k = c*L + l (nn.Flatten of [C=a, L=b])
// -> image k = l*C + c.
enum { CM_IDENT = 0, CM_BLOCK = 1, CM_FLATTEN = 2 };

struct LayerDesc {
  int is_conv;
  int cout, cin, taps, pad;   // canonical sizes (taps = 1 for linear)
  int cin_img;                // channels of the input image (incl. padding)
  int cmap, cmap_a, cmap_b;   // channel map
  int cout_p16, cout_p8;
  int KP;                     // fwd image row length  = roundup32(taps * cin_img)
  int KPt;                    // transposed image row length = roundup32(taps * cout_p8)
  int cin_p16;                // rows of the transposed image
  long w_off, wt_off;         // element offsets of the images inside one slot
  int bias_off;               // offset inside the bias arrays
  long canon_w, canon_b;      // offsets of weight / bias sites in the flat (mu, rho) buffers
  int site_w, site_b;         // site indices
  int sign_in_words, sign_out_words;  // u32 words per example in the packed sign arrays
  long sign_in_off, sign_out_off;     // word offsets PER EXAMPLE of this layer's rows inside the packed sign buffers
};

struct BranchDesc {
  int layer;
  int n_off;     // first cout of the layer handled by this branch (wide layers are split)
  int cout;      // real couts of this branch (<= 64)
  int ntiles;    // ceil(cout / 16)
  int in_off;    // channel offset of the branch input inside the group input tensor
  int cin_p;     // image channels per tap seen by this branch (mult. of 8; dense: of 32)
  int cin_real;  // channels really present in the source tensor starting at in_off
  int pool;      // 1: input is MaxPool1d(3,1,1) of the tensor
  int relu;
  int out_t, out_off;  // output tensor / channel offset
  int q_t;             // tensor that keeps q = eps / (2 sd) for the LRT backward (-1: none)
  int dx_t;            // tensor receiving d loss / d input of this branch (-1: not needed)
};

struct GroupDesc {
  int n_branch;
  int is_dense;   // windows are 32-row chunks of example rows
  int in_t;       // input tensor id
  int in_bcast;   // input has no particle dimension (x)
  int L;          // rows per window (conv: win_length; dense: 32)
  int in_cin_p;   // image channels of the whole input tensor view (for dx chunking)
  BranchDesc br[BNN_MAX_BRANCH];
};

// per-call geometry
struct CallGeom {
  int S, B;           // particles, local batch
  int Bglob, goff;    // global batch / offset of this rank (DP-invariant noise)
  int nwin;           // windows of the group launch
  int per_particle;   // windows per particle
};

// pointers to the weight-image slots of the current call
struct WeightSlots {
  const void* a_hi;   // slot A (mu | sampled W): f32 image, or bf16 hi part
  const void* a_lo;   // bf16 lo part (split-bf16 only)
  const void* b;      // slot B (sigma^2 | dW = sigma*eps)
  const void* at;     // transposed slot A (backward dx)
  const void* bt;     // transposed slot B
  const float* bias_a;  // [S or 1][bias_total]
  const float* bias_b;  // LRT: sigma_b^2
  long slot_stride_a;   // elements between particles (0: shared)
  long slot_stride_b;
  long slott_stride_a, slott_stride_b;
  int bias_stride_a;    // floats between particles (0: shared)
  int bias_total;
};

struct NoiseRefs {
  uint64_t seed;
  uint32_t step;
  int use_philox_lrt;                       // 1: draw LRT eps in-kernel
  long examples;                            // S*B of this call (layer sign arrays are back to back)
  const float* lrt_eps[BNN_MAX_LAYERS];     // injected [S][B][L][Cout] or null
  const uint32_t* sign_in;                  // packed bits, all layers
  const uint32_t* sign_out;
};
