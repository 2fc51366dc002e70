/*
 * bayesrul_amd.h — C ABI of the MI355X-native SVI/ELBO hot path of lbasora/bayesrul.
 *
 * The reference has NO native interface for this path: the boundary is the
 * Python class `bayesrul.models.bayesian.BNN` which delegates the arithmetic to Pyro/TyXe.
 * Each entry point below cites the reference call site(s) whose arithmetic it replaces.
 * The Python host (`bayesrul_amd/`) binds these with ctypes; no torch
 * types cross this boundary: plain device pointers, sizes and a hipStream_t (as void*).
 *
 * Conventions
 *   - every function returns 0 on success, a negative BNN_E_* code on failure; the message is
 *     available from bnn_last_error() (thread-local).  No C++ exception crosses the ABI.
 *   - all device memory is owned by the caller (PyTorch's caching allocator in the Python
 *     host); the library allocates nothing on the device.  Functions only enqueue work on the
 *     given stream and never synchronise.
 *   - parameters live in ONE flat fp32 buffer in the reference's own layout
 *     (`named_parameters` order, PyTorch [Cout,Cin,k] / [out,in] element order):
 *     mu[P], rho[P] (= log sigma, unconstrained), Adam m[2P], v[2P], grad[2P+2].
 */
#ifndef BAYESRUL_AMD_H
#define BAYESRUL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNN_ABI_VERSION 3

enum {
  BNN_OK = 0,
  BNN_E_INVALID = -1,   /* bad argument / unsupported configuration */
  BNN_E_UNBOUND = -2,   /* plan used before bnn_plan_bind */
  BNN_E_HIP = -3,       /* a HIP runtime call failed */
  BNN_E_NO_DEVICE = -4  /* no gfx950 device / kernel image not loadable */
};

/* network: bayesrul/models/nets/inception.py:142-217, nets/linear.py:10-72 (out_size=2) */
enum { BNN_NET_INCEPTION = 0, BNN_NET_LINEAR = 1 };
/* estimator: bayesian.py:66-85 (fit_context lrt|flipout|null, guide normal|radial) */
enum { BNN_MODE_NORMAL = 0, BNN_MODE_LRT = 1, BNN_MODE_FLIPOUT = 2, BNN_MODE_RADIAL = 3 };
/* contraction arithmetic: exact f32 MFMA (the reference trains in fp32: conf/trainer/default.yaml:8-12), or split-bf16
 * (hi+lo, 3 MFMAs on the mean path) with fp32 accumulation.  BNN_MODE_LRT on BNN_NET_INCEPTION is implemented on
 * BNN_PREC_F32 only: bnn_plan_create (and a per-call mode override) returns BNN_E_INVALID for it on a BNN_PREC_BF16X3 plan */
enum { BNN_PREC_F32 = 0, BNN_PREC_BF16X3 = 1 };

typedef struct BnnPlan BnnPlan;

typedef struct BnnPlanDesc {
  int32_t net;            /* BNN_NET_* */
  int32_t mode;           /* BNN_MODE_* used by bnn_elbo_* (training estimator) */
  int32_t prec;           /* BNN_PREC_* */
  int32_t max_particles;  /* S_max: largest mc_samples_{train,eval} the plan must hold */
  int32_t max_batch;      /* B_max: largest per-call batch (windows) */
  int32_t win_length;     /* W (30) */
  int32_t n_features;     /* F (18) */
  int32_t max_windows;    /* capacity S*B of one launch (0: max_particles*max_batch); bnn_predict
                             walks the particles in chunks of max_windows / batch */
} BnnPlanDesc;

/* device buffers the plan works on; all caller-owned */
typedef struct BnnBuffers {
  float* mu;       /* [P]  variational means, reference layout */
  float* rho;      /* [P]  log sigma */
  float* adam_m;   /* [2P] */
  float* adam_v;   /* [2P] */
  float* grad;     /* [2P+2]: d loss/d mu, d loss/d rho, loss, kl  (the DP all-reduce payload) */
  void*  workspace;
  size_t workspace_bytes;
} BnnBuffers;

/* per-call noise: either Philox (all pointers NULL) or injected (parity tests) */
typedef struct BnnNoise {
  uint64_t seed;           /* Philox key */
  uint64_t step;           /* Philox stream offset (the optimiser step) */
  const float* eps_w;      /* [S][P] weight noise, reference layout, or NULL */
  const float* radial_r;   /* [S][n_sites] radial distances, or NULL */
  const float* const* lrt_eps;   /* [n_layers] -> [S][B][L][Cout] (channels-last) or NULL */
  const float* const* sign_in;   /* [n_layers] -> [S][B][Cin_img] of +-1, or NULL */
  const float* const* sign_out;  /* [n_layers] -> [S][B][Cout]    of +-1, or NULL */
} BnnNoise;

typedef struct BnnElboArgs {
  const float* x;          /* [B][W][F] fp32 windows (NCMAPSSLmdbDataset layout, dataset.py:13-16) */
  const float* y;          /* [B] labels */
  int32_t batch;           /* B (local) */
  int32_t particles;       /* S = mc_samples_train */
  int32_t global_batch_offset; /* index of this rank's first window in the global batch (DP noise) */
  int32_t global_batch;    /* B_global (== batch on one GPU) */
  double dataset_size;     /* N */
  double prior_loc, prior_scale;
  int32_t mode_override;   /* -1: plan mode; else BNN_MODE_* (validation = plain sampling) */
  int32_t with_obs;        /* 0: svi_no_obs (KL only), 1: full ELBO */
  int32_t scaled;          /* 1: poutine.scale 1/(N*W*F) applied (bayesian.py:111-129) */
  int32_t reserved;
} BnnElboArgs;

typedef struct BnnAdamArgs {
  double lr, beta1, beta2, eps, clip_norm, weight_decay;
  int64_t step;            /* 1-based step count after increment */
  double grad_scale;       /* multiplies grad before the clamp (1/world_size after a sum all-reduce) */
  int32_t freeze_loc;      /* != 0: mu is not updated (guide option train_loc=False, guides/radial.py:74-76) */
  int32_t freeze_scale;    /* != 0: rho is not updated (train_scale=False, :90-94; the frequentist siblings) */
  int32_t torch_eps;       /* != 0: torch.optim.Adam's epsilon placement, sqrt(v / bc2) + eps (frequentist siblings) */
  int32_t reserved;
} BnnAdamArgs;

/* scalars + predictions produced by a step; device pointers into caller memory */
typedef struct BnnElboOut {
  float* loss;     /* [1]  the value svi.step returns */
  float* kl;       /* [1]  unscaled KL (or log q - log p), particle mean */
  float* loglik;   /* [1]  sum_b log N(y_b; ...), particle mean */
  float* preds;    /* [S][B][2] net outputs (loc, scale after softplus+threshold) or NULL */
} BnnElboOut;

/* ---- library ---- */
int bnn_version(void);
const char* bnn_last_error(void);
/* sizeof() of the ABI structs as compiled into the library, for binding self-checks:
 * 0 BnnPlanDesc, 1 BnnBuffers, 2 BnnNoise, 3 BnnElboArgs, 4 BnnAdamArgs, 5 BnnElboOut, 6 BnnDetArgs, 7 BnnDropout */
size_t bnn_abi_sizeof(int which);

/* ---- plan: replaces BNN.define_bnn / on_fit_start bookkeeping (bayesian.py:45-132) ---- */
int bnn_plan_create(const BnnPlanDesc* desc, BnnPlan** out);
void bnn_plan_destroy(BnnPlan* plan);
int bnn_plan_num_params(const BnnPlan* plan, int64_t* P);
int bnn_plan_num_sites(const BnnPlan* plan, int32_t* n_sites);
int bnn_plan_num_layers(const BnnPlan* plan, int32_t* n_layers);
int bnn_plan_workspace_bytes(const BnnPlan* plan, size_t* bytes);
int bnn_plan_bind(BnnPlan* plan, const BnnBuffers* bufs);
/* site table: name / offset / numel of site i in the flat buffers */
int bnn_plan_site(const BnnPlan* plan, int32_t i, const char** name, int64_t* offset, int64_t* numel);
/* layer table: image-channel count an injected sign_in row must have, and Cout */
int bnn_plan_layer(const BnnPlan* plan, int32_t i, const char** name, int32_t* cin_img, int32_t* cout,
                   int32_t* is_conv);
/* Host-side validation of every launch the plan would make for a call of this geometry (mode = BNN_MODE_* or -1 for the
 * plan's own, train != 0: forward + backward): DMA instruction counts vs LDS plane / slot sizes, slot rings vs windows in
 * flight, counted-wait ranges, LDS budgets.  No device work: usable without a GPU and before bnn_plan_bind. */
int bnn_plan_validate(BnnPlan* plan, int32_t mode, int32_t particles, int32_t batch, int32_t train);
/* debug/test access to intermediate activations of the last forward: tensor `which`
 * (see BNN_T_*), returns device pointer + row count + channel count */
enum { BNN_T_ACT1 = 0, BNN_T_MID = 1, BNN_T_ACT2 = 2, BNN_T_H = 3, BNN_T_Z = 4,
       BNN_T_H2 = 5, BNN_T_H3 = 6, BNN_T_H4 = 7 };
int bnn_plan_tensor(const BnnPlan* plan, int32_t which, float** ptr, int64_t* rows, int32_t* ctot);

/* ---- kernels, one entry per family ---- */

/* A7/A8/A10: AutoNormal / RadialNormal.rsample (guides/radial.py:31-41) / flipout dW: builds
 * the per-particle weight images the contraction kernels consume and the KL (A5) or
 * log q - log p (A6) reductions. */
int bnn_sample_weights(BnnPlan* plan, const BnnElboArgs* a, const BnnNoise* noise, void* stream);

/* A9/A10/A13/A14: S x B variational forwards of the net (LRT / Flipout / plain sampled) */
int bnn_forward(BnnPlan* plan, const BnnElboArgs* a, const BnnNoise* noise, void* stream);

/* A11: softplus -> Threshold -> (likelihood softplus) -> Gaussian log-lik, and its backward */
int bnn_head_nll(BnnPlan* plan, const BnnElboArgs* a, const BnnElboOut* out, void* stream);

/* backward of bnn_forward: activation grads + weight-image grads */
int bnn_backward(BnnPlan* plan, const BnnElboArgs* a, const BnnNoise* noise, void* stream);

/* chain rule to (mu, rho) + KL / prior terms -> grad[2P+2] (A5/A6) */
int bnn_grad_finalize(BnnPlan* plan, const BnnElboArgs* a, const BnnNoise* noise, void* stream);

/* A12: pyro.optim.ClippedAdam on the flat (mu, rho) buffer (conf/model/bnn.yaml:6-10) */
int bnn_clipped_adam(BnnPlan* plan, const BnnAdamArgs* a, void* stream);

/* A4: svi.step = sample -> forward -> head -> backward -> finalize [-> adam if adam != NULL].
 * With adam == NULL the caller all-reduces grad[2P+2] (RCCL) and calls bnn_clipped_adam. */
int bnn_elbo_step(BnnPlan* plan, const BnnElboArgs* a, const BnnNoise* noise, const BnnAdamArgs* adam,
                  const BnnElboOut* out, void* stream);

/* A4: svi.evaluate_loss (no grad): sample -> forward -> head */
int bnn_elbo_evaluate(BnnPlan* plan, const BnnElboArgs* a, const BnnNoise* noise, const BnnElboOut* out,
                      void* stream);

/* SURVEY.md 8(f) rank 4, the frequentist siblings on the same kernels: one deterministic training step of the net
 * with weights = mu (no sampling, no KL): forward -> loss -> backward -> d loss / d mu in grad[0..P) [-> optimiser].
 *   objective 1: HNN.step (bayesrul/models/frequentist.py:39-48): mean_b F.gaussian_nll_loss(loc, y, scale^2)
 *                (eps = 1e-6, var clamped without gradient), scale = the net's second output (single softplus)
 *   objective 2: NN.step (:173-178): mean_b (loc - y)^2
 * out->loss receives the mean loss; out->preds [1][B][2] the net outputs.  adam == NULL leaves the gradient in place. */
/* MC-dropout of the frequentist sibling (conf/experiment/ncmapss_mcd.yaml; nets/inception.py:48-52,119-123,205-207): nn.Dropout
 * behind every branch of both inception blocks (rate p / 4) and behind the hidden layer's ReLU (rate p); an element is kept
 * with probability 1 - rate and scaled by 1 / (1 - rate).  Exact-fp32 Inception plans only.  The keep masks come from the
 * Philox stream (seed, step) or, for parity runs, from injected buffers (1 keep / 0 drop; all three or none). */
typedef struct BnnDropout {
  double p;                /* net.dropout; 0 = off */
  uint64_t seed, step;
  const float* keep_act1;  /* [B*W][128]: block 1, channel c of branch b at 32 b + c */
  const float* keep_act2;  /* [B*W][80]:  block 2, the concatenation order of its branches (16, 16, 16, 32) */
  const float* keep_h;     /* [B][64] */
} BnnDropout;

typedef struct BnnDetArgs {
  const float* x;          /* [B][W][F] */
  const float* y;          /* [B] */
  int32_t batch;
  int32_t objective;       /* 1 gaussian NLL (HNN), 2 MSE (NN) */
  const BnnDropout* dropout;   /* NULL: none (HNN.training_step runs the net in train mode: dropout active, frequentist.py:50) */
} BnnDetArgs;
int bnn_det_step(BnnPlan* plan, const BnnDetArgs* a, const BnnAdamArgs* adam, const BnnElboOut* out, void* stream);
/* net(x) with weights = mu (frequentist.py:39-42, the forward of HNN.step), dropout active when given: one pass of
 * HNN.mc_sampling (frequentist.py:60-81).  preds_b2 = [B][2] net outputs (loc, scale). */
int bnn_det_forward(BnnPlan* plan, const float* x, int32_t batch, const BnnDropout* dropout, float* preds_b2, void* stream);

/* A16: bnn.predict(x, num_predictions=S, aggregate=False) + the aggregation of
 * predict_step / test_step (bayesian.py:203-250): out4 = [4][B] (preds, stds, ep_vars, al_vars);
 * preds_sb2 (optional) = [S][B][2] raw net outputs. */
int bnn_predict(BnnPlan* plan, const float* x, int32_t batch, int32_t particles, const BnnNoise* noise,
                float* preds_sb2, float* out4, void* stream);

/* test helper: writes the Philox noise the kernels would draw for (seed, step) into caller
 * buffers with the layouts of BnnNoise, so an oracle can replay identical noise. */
int bnn_export_noise(BnnPlan* plan, const BnnElboArgs* a, uint64_t seed, uint64_t step, float* eps_w,
                     float* radial_r, float* const* lrt_eps, float* const* sign_in, float* const* sign_out,
                     void* stream);

/* ---- window store: batch gather from an HBM-resident window set ----
 * Replaces LmdbDataset.__getitem__ (bayesrul/data/lmdb_utils.py:184-194), NCMAPSSLmdbDataset.__getitem__
 * (bayesrul/data/ncmapss/dataset.py:13-16) and the DataLoader collate: x_out[i] = window idx[i], y_out[i] = its RUL.
 * feature_major != 0: stored windows are the reference's LMDB values, [n_features][win_length] fp32, and are
 * transposed to [win_length][n_features] (lmdb_utils.py:190-191).  y_all / y_out may be NULL.  Device pointers.
 * n_windows = number of windows stored in x_all: an index outside [0, n_windows) is never dereferenced (the output
 * window is filled with NaN and the label with NaN, so a bad index is visible, not a fault). */
int bnn_gather_windows(const float* x_all, const float* y_all, const int64_t* idx, int64_t n, int64_t n_windows,
                       int32_t win_length, int32_t n_features, int32_t feature_major, float* x_out, float* y_out,
                       void* stream);

/* ---- measurement: per-kernel durations from HIP events recorded on the launch stream ----
 * tag = kind * 16 + group; kind: 0 group forward, 1 group dX, 2 group dW, 3 weight sampling,
 * 4 head/NLL, 5 gradient finalize, 6 ClippedAdam, 7 max-pool backward. */
int bnn_profile_enable(BnnPlan* plan, int on);
/* record only the launches carrying one of `tags` (n = 0: all launches).  Each recorded launch puts two
 * events on the stream, so a timed region selects the dominant kernel's tags only. */
int bnn_profile_select(BnnPlan* plan, const int32_t* tags, int32_t n);
/* kernel symbol (as rocprofv3 prints it, without the argument list) last recorded under `tag` */
int bnn_profile_name(BnnPlan* plan, int32_t tag, char* buf, int32_t cap);
int bnn_profile_read(BnnPlan* plan, int32_t* tags, double* ms, int64_t* count, int32_t cap, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* BAYESRUL_AMD_H */
