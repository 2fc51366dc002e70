// Host side of the C ABI (include/bayesrul_amd.h): plan construction (layer / group /
// tensor tables, workspace layout) and the launch sequences of the SVI/ELBO step.
// gfx950 only; compiled with hipcc into libbayesrul_amd.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/bayesrul_amd.h"
#include "kernels_group.h"
#include "kernels_misc.h"
#include "kernels_conv_bf.h"
#include "kernels_dense_fwd.h"
#include "kernels_trunk.h"
#include "kernels_trunk_bwd.h"
#include "kernels_trunk_dw.h"
#include "kernels_dense_ks.h"
#include "kernels_mlp.h"
#include "kernels_f32.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) return fail(BNN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)
#define BNN_TRY(expr)        \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != 0) return rc__; \
  } while (0)

// bnn_plan_validate: the launch functions run their geometry checks and return before touching the device
static thread_local bool g_dry = false;
#define BNN_DRY_RETURN() do { if (g_dry) return 0; } while (0)

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }
static inline long rupl(long v, long m) { return (v + m - 1) / m * m; }

// internal tensor ids (activation ids; +T_GRAD / +T_Q for twins)
enum { TI_ACT1 = 0, TI_MID = 1, TI_ACT2 = 2, TI_H = 3, TI_Z = 4, TI_H2 = 5, TI_H3 = 6, TI_H4 = 7, TI_ACT2F = 8 };

struct TensorSpec {
  int ctot = 0;
  int rows_per_example = 0;  // L for conv tensors, 1 for dense
  int fmt = TF_F32;          // storage of the activation / its gradient / its LRT q
  size_t off = 0, off_lo = 0;  // BYTE offsets inside the tensor region: act (hi | f32), act lo plane
  size_t goff = 0, qoff = 0;   // gradient, q (single plane)
  int alias = -1;            // shares memory with this tensor id
};

// optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> tag;         // kind * 16 + group
  size_t used = 0;
  bool filter = false;          // record only the tags set in `sel`
  bool sel[256] = {};
  std::string names[256];       // kernel symbol (as rocprofv3 prints it, without the argument list) last launched under a tag
  hipEvent_t* next(int t) {
    if (!on) return nullptr;
    if (filter && !(t >= 0 && t < 256 && sel[t])) return nullptr;
    if (used + 2 > ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
      ev.push_back(a);
      ev.push_back(b);
      tag.push_back(t);
    } else {
      tag[used / 2] = t;
    }
    used += 2;
    return &ev[used - 2];
  }
};
struct ProfScope {
  Prof* pf;
  int tag;
  hipEvent_t* e;
  hipStream_t st;
  ProfScope(Prof* p, int kind, int group, hipStream_t s)
      : pf(p), tag(kind * 16 + group), e(p ? p->next(kind * 16 + group) : nullptr), st(s) {
    if (e) (void)hipEventRecord(e[0], st);
  }
  void name(const char* fmt, ...) {
    if (!e || !pf) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    pf->names[tag] = buf;
  }
  ~ProfScope() {
    if (e) (void)hipEventRecord(e[1], st);
  }
};
enum { DENSEF_DW_PARTS = 25 };   // partial images of densef_dw_kernel: (row ranges - 1) x particles <= 256 / 10 chunks
enum { PK_FWD = 0, PK_DX = 1, PK_DW = 2, PK_SAMPLE = 3, PK_HEAD = 4, PK_FINALIZE = 5, PK_ADAM = 6, PK_POOLBWD = 7, PK_NOISE = 8 };


struct BnnPlan {
  BnnPlanDesc d;
  Prof prof;
  int n_layers = 0, n_sites = 0, n_groups = 0;
  long P = 0;
  std::vector<std::string> site_names, layer_names;
  LayerDesc layers[BNN_MAX_LAYERS];
  ParamTable ptab;
  GroupDesc groups[8];
  TensorSpec tens[10];
  long img_total = 0, imgt_total = 0;
  int bias_total = 0;
  long sign_in_words_total = 0, sign_out_words_total = 0;  // per example
  long cap_windows = 0;
  int z_t = TI_Z;
  int x_ctot = 18;
  // workspace layout (byte offsets)
  size_t ws_bytes = 0;
  size_t o_slab_a[3], o_slab_b[3], o_slab_ba[3];   // partial images of the fused trunk dW kernels
  size_t o_slab_bb = 0;                            // fp32 LRT: partial sums of the sigma_b^2 gradients
  size_t o_dw2_a = 0, o_dw2_b = 0, o_dw2_ba = 0, o_dw2_bb = 0;   // fp32 plan: partial images of the wide dense layer's dW (DENSEF_DW_PARTS x (row range, particle))
  size_t o_dksv = 0;                               // fp32 LRT plan: partial variances of the K-split dense forward
  size_t o_mact2 = 0;                              // fp32 plan: nibble masks [ACT2 > 0] ([rows][20 B])
  size_t o_mact1 = 0, o_mmid = 0;                  // bit masks [ACT1 > 0] / [MID > 0] of the trunk kernels: [rows][16 B]
  size_t o_dks = 0;                                // partial pre-activations of the K-split dense forward [chunk][rows][64]
  long dks_rows = 0;
  bool fwd_fused_last = false;                     // the last do_forward evaluated the last layer in the fin kernel (and zeroed its gradients)
  bool acc_clean = false;                          // the loss accumulators are zero (the last finish_loss re-armed them)
  size_t o_mlp_x = 0, o_mlp_dz4 = 0;               // fused Linear-net kernels: x hi plane [B][544]; dz / dz q of the last layer
  int mlp = 0;
  long slab_stride = 0;
  int slab_bstride = 0;
  int slab_slots[3] = {0, 0, 0};                     // slabs allocated per group
  size_t o_layers, o_a_hi, o_a_lo, o_b, o_at, o_bt, o_bias_a, o_bias_b, o_gw_a, o_gw_b, o_gb_a, o_gb_b, o_eps, o_radr,
      o_norms, o_norm_part, o_sign_in, o_sign_out, o_acc, o_scal, o_preds, o_poolgrad, o_xplanes, o_amax, o_tens;
  size_t elem = 4;
  bool bound = false;
  BnnBuffers bufs{};
  // last call geometry (for bnn_plan_tensor)
  int last_S = 0, last_B = 0;
};

// ------------------------------------------------------------------------------------------
// network tables
// ------------------------------------------------------------------------------------------
struct LSpec {
  const char* name;
  int is_conv, cout, cin, taps;
  int cin_img, cmap, ca, cb;
};

static const LSpec kInception[] = {
    {"layers.0.conv1.0", 1, 27, 18, 1, 32, CM_IDENT, 0, 0},
    {"layers.0.conv3.0", 1, 27, 18, 3, 32, CM_IDENT, 0, 0},
    {"layers.0.conv5.0", 1, 27, 18, 5, 32, CM_IDENT, 0, 0},
    {"layers.0.convpool.1", 1, 27, 18, 3, 32, CM_IDENT, 0, 0},
    {"layers.1.branch1.0", 1, 16, 108, 1, 128, CM_BLOCK, 27, 32},
    {"layers.1.branch2.0", 1, 64, 108, 1, 128, CM_BLOCK, 27, 32},
    {"layers.1.branch2.2", 1, 16, 64, 3, 64, CM_IDENT, 0, 0},
    {"layers.1.branch3.0", 1, 64, 108, 1, 128, CM_BLOCK, 27, 32},
    {"layers.1.branch3.2", 1, 16, 64, 5, 64, CM_IDENT, 0, 0},
    {"layers.1.branch4.1", 1, 32, 108, 1, 128, CM_BLOCK, 27, 32},
    {"layers.3", 0, 64, 2400, 1, 2400, CM_FLATTEN, 80, 30},
    {"last", 0, 2, 64, 1, 64, CM_IDENT, 0, 0},
};
static const LSpec kLinear[] = {
    {"layers.1", 0, 256, 540, 1, 544, CM_IDENT, 0, 0}, {"layers.3", 0, 128, 256, 1, 256, CM_IDENT, 0, 0},
    {"layers.5", 0, 128, 128, 1, 128, CM_IDENT, 0, 0}, {"layers.7", 0, 32, 128, 1, 128, CM_IDENT, 0, 0},
    {"last", 0, 2, 32, 1, 32, CM_IDENT, 0, 0},
};

static BranchDesc mk_branch(int layer, int n_off, int cout, int in_off, int cin_p, int cin_real, int pool, int relu,
                            int out_t, int out_off, int dx_t) {
  BranchDesc b{};
  b.layer = layer;
  b.n_off = n_off;
  b.cout = cout;
  b.ntiles = (cout + 15) / 16;
  b.in_off = in_off;
  b.cin_p = cin_p;
  b.cin_real = cin_real;
  b.pool = pool;
  b.relu = relu;
  b.out_t = out_t;
  b.out_off = out_off;
  b.q_t = out_t + T_Q;
  b.dx_t = dx_t;
  return b;
}

static int build_tables(BnnPlan* p) {
  const bool inc = p->d.net == BNN_NET_INCEPTION;
  const LSpec* ls = inc ? kInception : kLinear;
  p->n_layers = inc ? 12 : 5;
  const int L = p->d.win_length;
  if (inc && (L < 1 || L > 30 + 0) && L != 30)
    return fail(BNN_E_INVALID, "Inception needs 1 <= win_length <= 30 (one window per 32-row MFMA tile), got %d", L);
  if (inc && L > TILE_ROWS - 2) return fail(BNN_E_INVALID, "win_length %d too long for a 32-row window tile", L);
  if (p->d.n_features != 18) return fail(BNN_E_INVALID, "n_features must be 18 (nets/inception.py:160-162)");
  long canon = 0, woff = 0, wtoff = 0, si_w = 0, so_w = 0;
  int boff = 0;
  p->ptab = ParamTable{};
  for (int i = 0; i < p->n_layers; ++i) {
    LayerDesc& l = p->layers[i];
    l = LayerDesc{};
    l.is_conv = ls[i].is_conv;
    l.cout = ls[i].cout;
    l.cin = ls[i].cin;
    l.taps = ls[i].taps;
    l.pad = (l.taps - 1) / 2;
    l.cin_img = ls[i].cin_img;
    l.cmap = ls[i].cmap;
    l.cmap_a = ls[i].ca;
    l.cmap_b = ls[i].cb;
    if (inc && i == 10) {  // Flatten of [80, L]
      l.cin = 80 * L;
      l.cin_img = 80 * L;
      l.cmap_b = L;
      if (l.cin_img % 32) return fail(BNN_E_INVALID, "80*win_length must be a multiple of 32");
    }
    if (!inc && i == 0) {
      l.cin = L * 18;
      l.cin_img = rup(l.cin, 32);
    }
    l.cout_p16 = rup(l.cout, 16);
    l.cout_p8 = rup(l.cout, 8);
    if (!l.is_conv) l.cout_p8 = rup(l.cout, 32);  // dense dZ images are 32-wide chunks
    l.KP = rup(l.taps * l.cin_img, 32);
    l.KPt = rup(l.taps * l.cout_p8, 32);
    l.cin_p16 = rup(l.cin_img, 16);
    l.w_off = woff;
    l.wt_off = wtoff;
    woff += (long)l.cout_p16 * l.KP;
    wtoff += (long)l.cin_p16 * l.KPt;
    l.bias_off = boff;
    boff += l.cout_p16;
    l.canon_w = canon;
    canon += (long)l.cout * l.cin * l.taps;
    l.canon_b = canon;
    canon += l.cout;
    l.site_w = 2 * i;
    l.site_b = 2 * i + 1;
    l.sign_in_words = (l.cin_img + 31) / 32;
    l.sign_out_words = (l.cout + 31) / 32;
    l.sign_in_off = si_w;   // multiplied by examples at call time
    l.sign_out_off = so_w;
    si_w += l.sign_in_words;
    so_w += l.sign_out_words;
    p->layer_names.push_back(ls[i].name);
    p->site_names.push_back(std::string(ls[i].name) + ".weight");
    p->site_names.push_back(std::string(ls[i].name) + ".bias");
    SiteDesc& sw = p->ptab.site[2 * i];
    sw.off = l.canon_w;
    sw.numel = (long)l.cout * l.cin * l.taps;
    sw.layer = i;
    sw.is_bias = 0;
    SiteDesc& sb = p->ptab.site[2 * i + 1];
    sb.off = l.canon_b;
    sb.numel = l.cout;
    sb.layer = i;
    sb.is_bias = 1;
  }
  p->n_sites = 2 * p->n_layers;
  p->P = canon;
  p->ptab.n_sites = p->n_sites;
  p->ptab.n_layers = p->n_layers;
  p->ptab.P = canon;
  p->img_total = rupl(woff, 64);
  p->imgt_total = rupl(wtoff, 64);
  p->bias_total = rup(boff, 16);
  p->sign_in_words_total = si_w;
  p->sign_out_words_total = so_w;

  // tensors
  for (auto& t : p->tens) t = TensorSpec{};
  auto T = [&](int id, int ctot, int rpe, int alias = -1) {
    p->tens[id].ctot = ctot;
    p->tens[id].rows_per_example = rpe;
    p->tens[id].alias = alias;
  };
  p->n_groups = 0;
  if (inc) {
    T(TI_ACT1, 128, L);
    T(TI_MID, 128, L);
    T(TI_ACT2, 80, L);
    T(TI_ACT2F, 80 * L, 1, TI_ACT2);
    T(TI_H, 64, 1);
    T(TI_Z, 2, 1);
    p->x_ctot = 18;
    GroupDesc g{};
    // block 1 (nets/inception.py:10-61)
    g = GroupDesc{};
    g.n_branch = 4; g.is_dense = 0; g.in_t = T_X; g.in_bcast = 1; g.L = L; g.in_cin_p = 32;
    g.br[0] = mk_branch(0, 0, 27, 0, 32, 18, 0, 1, TI_ACT1, 0, -1);
    g.br[1] = mk_branch(1, 0, 27, 0, 32, 18, 0, 1, TI_ACT1, 32, -1);
    g.br[2] = mk_branch(2, 0, 27, 0, 32, 18, 0, 1, TI_ACT1, 64, -1);
    g.br[3] = mk_branch(3, 0, 27, 0, 32, 18, 1, 1, TI_ACT1, 96, -1);
    p->groups[p->n_groups++] = g;
    // block 2, 1x1 level (nets/inception.py:71-132)
    g = GroupDesc{};
    g.n_branch = 4; g.is_dense = 0; g.in_t = TI_ACT1; g.in_bcast = 0; g.L = L; g.in_cin_p = 128;
    g.br[0] = mk_branch(4, 0, 16, 0, 128, 128, 0, 1, TI_ACT2, 0, TI_ACT1 + T_GRAD);
    g.br[1] = mk_branch(5, 0, 64, 0, 128, 128, 0, 1, TI_MID, 0, TI_ACT1 + T_GRAD);
    g.br[2] = mk_branch(7, 0, 64, 0, 128, 128, 0, 1, TI_MID, 64, TI_ACT1 + T_GRAD);
    g.br[3] = mk_branch(9, 0, 32, 0, 128, 128, 1, 1, TI_ACT2, 48, T_POOLGRAD);
    p->groups[p->n_groups++] = g;
    // block 2, k3 / k5 level
    g = GroupDesc{};
    g.n_branch = 2; g.is_dense = 0; g.in_t = TI_MID; g.in_bcast = 0; g.L = L; g.in_cin_p = 128;
    g.br[0] = mk_branch(6, 0, 16, 0, 64, 64, 0, 1, TI_ACT2, 16, TI_MID + T_GRAD);
    g.br[1] = mk_branch(8, 0, 16, 64, 64, 64, 0, 1, TI_ACT2, 32, TI_MID + T_GRAD);
    p->groups[p->n_groups++] = g;
    // Flatten + Linear(80*L -> 64) + ReLU
    g = GroupDesc{};
    g.n_branch = 1; g.is_dense = 1; g.in_t = TI_ACT2F; g.in_bcast = 0; g.L = TILE_ROWS; g.in_cin_p = 80 * L;
    g.br[0] = mk_branch(10, 0, 64, 0, 80 * L, 80 * L, 0, 1, TI_H, 0, TI_ACT2F + T_GRAD);
    p->groups[p->n_groups++] = g;
    // last Linear(64 -> 2)
    g = GroupDesc{};
    g.n_branch = 1; g.is_dense = 1; g.in_t = TI_H; g.in_bcast = 0; g.L = TILE_ROWS; g.in_cin_p = 64;
    g.br[0] = mk_branch(11, 0, 2, 0, 64, 64, 0, 0, TI_Z, 0, TI_H + T_GRAD);
    p->groups[p->n_groups++] = g;
  } else {
    const int F = L * 18, Fp = rup(F, 32);
    T(TI_H, 256, 1);
    T(TI_H2, 128, 1);
    T(TI_H3, 128, 1);
    T(TI_H4, 32, 1);
    T(TI_Z, 2, 1);
    p->x_ctot = F;
    GroupDesc g{};
    g.n_branch = 4; g.is_dense = 1; g.in_t = T_X; g.in_bcast = 1; g.L = TILE_ROWS; g.in_cin_p = Fp;
    for (int k = 0; k < 4; ++k) g.br[k] = mk_branch(0, 64 * k, 64, 0, Fp, F, 0, 1, TI_H, 64 * k, -1);
    p->groups[p->n_groups++] = g;
    g = GroupDesc{};
    g.n_branch = 2; g.is_dense = 1; g.in_t = TI_H; g.L = TILE_ROWS; g.in_cin_p = 256;
    for (int k = 0; k < 2; ++k) g.br[k] = mk_branch(1, 64 * k, 64, 0, 256, 256, 0, 1, TI_H2, 64 * k, TI_H + T_GRAD);
    p->groups[p->n_groups++] = g;
    g = GroupDesc{};
    g.n_branch = 2; g.is_dense = 1; g.in_t = TI_H2; g.L = TILE_ROWS; g.in_cin_p = 128;
    for (int k = 0; k < 2; ++k) g.br[k] = mk_branch(2, 64 * k, 64, 0, 128, 128, 0, 1, TI_H3, 64 * k, TI_H2 + T_GRAD);
    p->groups[p->n_groups++] = g;
    g = GroupDesc{};
    g.n_branch = 1; g.is_dense = 1; g.in_t = TI_H3; g.L = TILE_ROWS; g.in_cin_p = 128;
    g.br[0] = mk_branch(3, 0, 32, 0, 128, 128, 0, 1, TI_H4, 0, TI_H3 + T_GRAD);
    p->groups[p->n_groups++] = g;
    g = GroupDesc{};
    g.n_branch = 1; g.is_dense = 1; g.in_t = TI_H4; g.L = TILE_ROWS; g.in_cin_p = 32;
    g.br[0] = mk_branch(4, 0, 2, 0, 32, 32, 0, 0, TI_Z, 0, TI_H4 + T_GRAD);
    p->groups[p->n_groups++] = g;
  }
  return 0;
}

static void layout_workspace(BnnPlan* p) {
  const size_t el = p->d.prec == BNN_PREC_BF16X3 ? 2 : 4;
  p->elem = el;
  const long S = p->d.max_particles;
  const long cap = p->cap_windows;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (bytes + 255) & ~(size_t)255;
    return r;
  };
  p->o_layers = take(sizeof(LayerDesc) * BNN_MAX_LAYERS);
  p->o_a_hi = take((size_t)S * p->img_total * el);
  p->o_a_lo = take((size_t)S * p->img_total * el);
  p->o_b = take((size_t)S * p->img_total * el);
  p->o_at = take((size_t)S * p->imgt_total * el);
  p->o_bt = take((size_t)S * p->imgt_total * el);
  p->o_bias_a = take((size_t)S * p->bias_total * 4);
  p->o_bias_b = take((size_t)p->bias_total * 4);
  p->o_gw_a = take((size_t)S * p->img_total * 4);
  p->o_gw_b = take((size_t)S * p->img_total * 4);
  p->o_gb_a = take((size_t)S * p->bias_total * 4);
  p->o_gb_b = take((size_t)S * p->bias_total * 4);
  p->o_eps = take((size_t)S * p->P * 4);
  p->o_radr = take((size_t)S * p->n_sites * 4);
  p->o_norms = take((size_t)S * p->n_sites * 4);
  p->o_norm_part = take((size_t)S * (p->n_sites + p->P / SN_CHUNK + 1) * 8);   // radial: partial sums of squares per chunk
  p->o_sign_in = take((size_t)cap * p->sign_in_words_total * 4);
  p->o_sign_out = take((size_t)cap * p->sign_out_words_total * 4);
  p->o_acc = take(sizeof(double) * 2 * (S + 1));
  p->o_scal = take(64);
  p->o_preds = take((size_t)S * p->d.max_batch * 2 * 4);
  p->o_poolgrad = take((size_t)cap * p->d.win_length * 128 * 4);
  p->o_amax = take(p->d.prec == BNN_PREC_BF16X3 && p->d.net == BNN_NET_INCEPTION ? (size_t)cap * p->d.win_length * 128 : 0);
  if (p->d.prec == BNN_PREC_F32 && p->d.net == BNN_NET_INCEPTION) {
    // fused fp32 trunk (kernels_f32.h): 2-bit arg-max codes and ReLU masks of ACT1 / MID, one byte per 4 channels
    p->o_amax = take((size_t)cap * p->d.win_length * 32);
    p->o_mact1 = take((size_t)cap * p->d.win_length * 32);
    p->o_mmid = take((size_t)cap * p->d.win_length * 32);
    p->o_mact2 = take((size_t)cap * p->d.win_length * 20);
  }
  // fused trunk dW: one partial image per workgroup, S * nsplit <= 256 (512 for the k3 / k5 kernel) whatever the call
  p->slab_stride = 0;
  if (p->d.net == BNN_NET_INCEPTION) {
    // S * nsplit slabs per launch, nsplit = max(1, min(B, 256 (512) / S)): at most max(256 (512), S) whatever the call.
    // The fp32 plan's two dW launches write disjoint layers of ONE slab set (group 0).
    p->slab_stride = rupl(p->layers[10].w_off, 64);
    p->slab_bstride = rup(p->layers[10].bias_off, 16);
    const int ngrp = p->d.prec == BNN_PREC_BF16X3 ? 3 : 1;
    for (int g = 0; g < 3; ++g) {
      p->slab_slots[g] = g < ngrp ? std::max(g == 2 ? 512 : 256, (int)p->d.max_particles) : 0;
      const int gs = g < ngrp ? g : 0;
      p->o_slab_a[g] = g < ngrp ? take((size_t)p->slab_slots[g] * p->slab_stride * 4) : p->o_slab_a[gs];
      p->o_slab_b[g] = g < ngrp ? take((size_t)p->slab_slots[g] * p->slab_stride * 4) : p->o_slab_b[gs];
      p->o_slab_ba[g] = g < ngrp ? take((size_t)p->slab_slots[g] * p->slab_bstride * 4) : p->o_slab_ba[gs];
      if (g >= ngrp) p->slab_slots[g] = p->slab_slots[gs];
    }
    if (p->d.prec == BNN_PREC_F32) p->o_slab_bb = take((size_t)p->slab_slots[0] * p->slab_bstride * 4);
  }
  if (p->d.prec == BNN_PREC_BF16X3 && p->d.net == BNN_NET_LINEAR && p->layers[0].KP == ML_K0 && p->layers[0].cout == ML_N0 &&
      p->layers[1].cout == ML_N1 && p->layers[2].cout == ML_N2 && p->layers[3].cout == ML_N3 && p->layers[4].cout == ML_N4) {
    p->mlp = 1;
    p->o_mlp_x = take((size_t)p->d.max_batch * ML_K0 * 2);
    p->o_mlp_dz4 = take((size_t)cap * 8 * 2 * 2);
  }
  if (p->d.prec == BNN_PREC_F32 && p->d.net == BNN_NET_INCEPTION && p->layers[10].cin % FDF_CH == 0) {
    // partial images of densef_dw_kernel's row ranges 1, 2, ...: the dense layer's part of slot A | slot B, and its biases
    // (laid out like the gradient images themselves: same offsets and particle strides)
    p->o_dw2_a = take((size_t)DENSEF_DW_PARTS * p->img_total * 4);
    p->o_dw2_b = take((size_t)DENSEF_DW_PARTS * p->img_total * 4);
    p->o_dw2_ba = take((size_t)DENSEF_DW_PARTS * p->bias_total * 4);
    p->o_dw2_bb = take((size_t)DENSEF_DW_PARTS * p->bias_total * 4);
    p->dks_rows = cap;
    p->o_dks = take((size_t)(p->layers[10].cin / FDF_CH) * cap * 64 * 4);   // partial pre-activations of densef_fwd_kernel
    p->o_dksv = p->d.mode == BNN_MODE_LRT ? take((size_t)(p->layers[10].cin / FDF_CH) * cap * 64 * 4) : 0;   // LRT: partial variances
  }
  if (p->d.prec == BNN_PREC_BF16X3 && p->d.net == BNN_NET_INCEPTION) {
    p->o_mact1 = take((size_t)cap * p->d.win_length * 16);
    p->o_mmid = take((size_t)cap * p->d.win_length * 16);
    p->dks_rows = cap;
    p->o_dks = take((size_t)((p->layers[10].cin + DK_CH - 1) / DK_CH) * cap * 64 * 4);
  }
  p->o_xplanes = take((size_t)4 * p->d.max_batch * p->d.win_length * 32 * 2);   // x hi | lo | pooled hi | pooled lo
  p->o_tens = o;
  // activations / grads / q.  f32 plans keep fp32 rows; bf16x3 plans keep bf16 planes
  // (activation = hi + lo planes, gradient and q = one plane) except the net output z.
  size_t fo = 0;
  auto takeb = [&](size_t bytes) {
    size_t r = fo;
    fo += (bytes + 255) & ~(size_t)255;
    return r;
  };
  for (int t = 0; t < 10; ++t) {
    TensorSpec& ts = p->tens[t];
    if (ts.ctot == 0 || ts.alias >= 0) continue;
    const size_t n = (size_t)cap * ts.rows_per_example * ts.ctot;
    ts.fmt = (p->d.prec == BNN_PREC_BF16X3 && t != TI_Z) ? TF_BF16 : TF_F32;
    const size_t eb = ts.fmt == TF_BF16 ? 2 : 4;
    ts.off = takeb(n * eb);
    ts.off_lo = ts.fmt == TF_BF16 ? takeb(n * eb) : 0;
    ts.goff = takeb(n * eb);
    ts.qoff = takeb(n * eb);
  }
  for (int t = 0; t < 10; ++t) {
    TensorSpec& ts = p->tens[t];
    if (ts.alias >= 0) {
      const TensorSpec& o = p->tens[ts.alias];
      ts.fmt = o.fmt;
      ts.off = o.off;
      ts.off_lo = o.off_lo;
      ts.goff = o.goff;
      ts.qoff = o.qoff;
    }
  }
  p->ws_bytes = p->o_tens + fo;
}

// ------------------------------------------------------------------------------------------
// API: library / plan
// ------------------------------------------------------------------------------------------
extern "C" int bnn_version(void) { return BNN_ABI_VERSION; }
extern "C" const char* bnn_last_error(void) { return g_err; }
extern "C" size_t bnn_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(BnnPlanDesc);
    case 1: return sizeof(BnnBuffers);
    case 2: return sizeof(BnnNoise);
    case 3: return sizeof(BnnElboArgs);
    case 4: return sizeof(BnnAdamArgs);
    case 5: return sizeof(BnnElboOut);
    case 6: return sizeof(BnnDetArgs);
    case 7: return sizeof(BnnDropout);
  }
  return 0;
}

extern "C" int bnn_plan_create(const BnnPlanDesc* desc, BnnPlan** out) {
  if (!desc || !out) return fail(BNN_E_INVALID, "null argument");
  if (desc->net != BNN_NET_INCEPTION && desc->net != BNN_NET_LINEAR) return fail(BNN_E_INVALID, "unknown net %d", desc->net);
  if (desc->mode < 0 || desc->mode > 3) return fail(BNN_E_INVALID, "unknown mode %d", desc->mode);
  if (desc->prec != BNN_PREC_F32 && desc->prec != BNN_PREC_BF16X3) return fail(BNN_E_INVALID, "unknown prec %d", desc->prec);
  if (desc->max_particles < 1 || desc->max_batch < 1) return fail(BNN_E_INVALID, "max_particles / max_batch must be >= 1");
  if (desc->mode == BNN_MODE_LRT && desc->prec == BNN_PREC_BF16X3 && desc->net == BNN_NET_INCEPTION)
    return fail(BNN_E_INVALID, "LRT on the Inception net is implemented on the exact-fp32 plan (BNN_PREC_F32); the split-bf16 plan "
                               "covers Flipout, radial and plain sampling there, and the Linear net's LRT");
  BnnPlan* p = new BnnPlan();
  p->d = *desc;
  p->cap_windows = desc->max_windows > 0 ? desc->max_windows : (long)desc->max_particles * desc->max_batch;
  if (p->cap_windows < desc->max_batch) p->cap_windows = desc->max_batch;
  int rc = build_tables(p);
  if (rc) {
    delete p;
    return rc;
  }
  layout_workspace(p);
  *out = p;
  return 0;
}
extern "C" void bnn_plan_destroy(BnnPlan* plan) {
  if (plan)
    for (hipEvent_t e : plan->prof.ev) (void)hipEventDestroy(e);
  delete plan;
}
extern "C" int bnn_plan_num_params(const BnnPlan* p, int64_t* P) {
  if (!p || !P) return fail(BNN_E_INVALID, "null argument");
  *P = p->P;
  return 0;
}
extern "C" int bnn_plan_num_sites(const BnnPlan* p, int32_t* n) {
  if (!p || !n) return fail(BNN_E_INVALID, "null argument");
  *n = p->n_sites;
  return 0;
}
extern "C" int bnn_plan_num_layers(const BnnPlan* p, int32_t* n) {
  if (!p || !n) return fail(BNN_E_INVALID, "null argument");
  *n = p->n_layers;
  return 0;
}
extern "C" int bnn_plan_workspace_bytes(const BnnPlan* p, size_t* bytes) {
  if (!p || !bytes) return fail(BNN_E_INVALID, "null argument");
  *bytes = p->ws_bytes;
  return 0;
}
extern "C" int bnn_plan_site(const BnnPlan* p, int32_t i, const char** name, int64_t* offset, int64_t* numel) {
  if (!p || i < 0 || i >= p->n_sites) return fail(BNN_E_INVALID, "site index out of range");
  if (name) *name = p->site_names[i].c_str();
  if (offset) *offset = p->ptab.site[i].off;
  if (numel) *numel = p->ptab.site[i].numel;
  return 0;
}
extern "C" int bnn_plan_layer(const BnnPlan* p, int32_t i, const char** name, int32_t* cin_img, int32_t* cout,
                              int32_t* is_conv) {
  if (!p || i < 0 || i >= p->n_layers) return fail(BNN_E_INVALID, "layer index out of range");
  if (name) *name = p->layer_names[i].c_str();
  if (cin_img) *cin_img = p->layers[i].cin_img;
  if (cout) *cout = p->layers[i].cout;
  if (is_conv) *is_conv = p->layers[i].is_conv;
  return 0;
}

static float* ws_f(const BnnPlan* p, size_t off) { return (float*)((char*)p->bufs.workspace + off); }
static TensorRef tens_ref(const BnnPlan* p, int t, int which /*0 act,1 grad,2 q*/) {
  const TensorSpec& ts = p->tens[t];
  char* base = (char*)p->bufs.workspace + p->o_tens;
  TensorRef r{};
  r.ctot = ts.ctot;
  r.fmt = ts.fmt;
  r.p = base + (which == 0 ? ts.off : (which == 1 ? ts.goff : ts.qoff));
  r.lo = (which == 0 && ts.fmt == TF_BF16) ? base + ts.off_lo : nullptr;
  return r;
}
static float* tens_ptr(const BnnPlan* p, int t, int which) { return (float*)tens_ref(p, t, which).p; }

extern "C" int bnn_plan_bind(BnnPlan* p, const BnnBuffers* b) {
  if (!p || !b) return fail(BNN_E_INVALID, "null argument");
  if (!b->mu || !b->rho || !b->grad || !b->workspace) return fail(BNN_E_INVALID, "mu/rho/grad/workspace must be set");
  if (b->workspace_bytes < p->ws_bytes)
    return fail(BNN_E_INVALID, "workspace too small: %zu < %zu", b->workspace_bytes, p->ws_bytes);
  if (((uintptr_t)b->workspace) & 255) return fail(BNN_E_INVALID, "workspace must be 256-byte aligned");
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(BNN_E_NO_DEVICE, "device is %s, this library is built for gfx950 only", prop.gcnArchName);
  p->bufs = *b;
  // zero everything once: image pads, activation pads (e.g. channel 27 of each block-1 branch)
  HIP_TRY(hipMemset(b->workspace, 0, p->ws_bytes));
  HIP_TRY(hipMemcpy((char*)b->workspace + p->o_layers, p->layers, sizeof(LayerDesc) * BNN_MAX_LAYERS,
                    hipMemcpyHostToDevice));
  p->bound = true;
  return 0;
}

extern "C" int bnn_plan_tensor(const BnnPlan* p, int32_t which, float** ptr, int64_t* rows, int32_t* ctot) {
  if (!p || !p->bound) return fail(BNN_E_UNBOUND, "plan not bound");
  if (which < 0 || which > 7 || p->tens[which].ctot == 0) return fail(BNN_E_INVALID, "tensor %d not part of this net", which);
  if (p->tens[which].fmt != TF_F32) return fail(BNN_E_INVALID, "tensor %d is stored as bf16 planes (debug access needs an f32 plan)", which);
  if (ptr) *ptr = tens_ptr(p, which, 0);
  if (rows) *rows = (int64_t)p->last_S * p->last_B * p->tens[which].rows_per_example;
  if (ctot) *ctot = p->tens[which].ctot;
  return 0;
}

// ------------------------------------------------------------------------------------------
// per-call context
// ------------------------------------------------------------------------------------------
struct Ctx {
  int mode;  // BNN_MODE_*
  int em;    // EM_*
  int S, B;
  bool radial;
  bool train;
  hipStream_t st;
  const float* eps_w = nullptr;
  const float* rad_r = nullptr;
  float c, scale_ll, n_over_b;
  int objective = 0;   // 0 ELBO; 1 / 2: frequentist objectives (bnn_det_step)
  NoiseRefs nz{};
  int s_base = 0;  // particle offset for noise streams (predict chunks)
  bool x_planes_ready = false;   // predictive pass, chunks after the first: the planes of x are already in the workspace
  const float* fuse_x = nullptr; // training step on the trunk path: prepare_noise may generate the planes of these windows in
                                 // the launch that generates the noise (step_inputs_kernel)
  const BnnDropout* drop = nullptr;   // MC-dropout of this call (frequentist siblings; fp32 Inception plans)
  bool direct_assumed = false;   // grads_zeroed was set WITHOUT a fill: the K-split dense backward must store every element
  bool grads_zeroed = false;     // the gradient images need no fill in do_backward (done earlier, or every element is stored)
  bool last_fused = false;       // Inception trunk path: the last layer Linear(64, 2) runs inside the fin / head kernels
  bool head_fused = false;       // bnn_elbo_step on the fused Linear-net path: the head runs inside the backward's first kernel
  float* head_preds = nullptr;
};

static int em_of(int mode) { return mode == BNN_MODE_LRT ? EM_LRT : (mode == BNN_MODE_FLIPOUT ? EM_FLIPOUT : EM_PLAIN); }

static int make_ctx(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, void* stream, bool train, Ctx* c) {
  if (!p || !a) return fail(BNN_E_INVALID, "null argument");
  if (!p->bound) return fail(BNN_E_UNBOUND, "plan not bound (call bnn_plan_bind)");
  c->mode = a->mode_override >= 0 ? a->mode_override : p->d.mode;
  if (c->mode < 0 || c->mode > 3) return fail(BNN_E_INVALID, "bad mode %d", c->mode);
  c->em = em_of(c->mode);
  c->radial = c->mode == BNN_MODE_RADIAL;
  if (c->em == EM_LRT && p->d.prec == BNN_PREC_BF16X3 && p->d.net == BNN_NET_INCEPTION)
    return fail(BNN_E_INVALID, "LRT on the Inception net is implemented on the exact-fp32 plan (BNN_PREC_F32); the split-bf16 plan "
                               "covers Flipout, radial and plain sampling there, and the Linear net's LRT");
  c->S = a->particles;
  c->B = a->batch;
  c->train = train;
  c->st = (hipStream_t)stream;
  if (c->S < 1 || c->S > p->d.max_particles) return fail(BNN_E_INVALID, "particles %d outside [1, %d]", c->S, p->d.max_particles);
  if (c->B < 1 || c->B > p->d.max_batch) return fail(BNN_E_INVALID, "batch %d outside [1, %d]", c->B, p->d.max_batch);
  if ((long)c->S * c->B > p->cap_windows)
    return fail(BNN_E_INVALID, "S*B = %ld exceeds the plan capacity %ld windows", (long)c->S * c->B, p->cap_windows);
  if (a->with_obs && (!a->x || !a->y)) return fail(BNN_E_INVALID, "x / y must be set when with_obs != 0");
  const double cc = a->scaled ? 1.0 / (a->dataset_size * p->d.win_length * p->d.n_features) : 1.0;
  c->c = (float)cc;
  c->n_over_b = (float)(a->dataset_size / a->batch);
  c->scale_ll = a->with_obs ? (float)(cc * (a->dataset_size / a->batch) / c->S) : 0.f;
  return 0;
}

template <class K>
static int set_lds(K kernel, int bytes) {
  HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

// ------------------------------------------------------------------------------------------
// noise preparation: eps_w / radial r / packed signs
// ------------------------------------------------------------------------------------------
static int prepare_noise(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, Ctx* c) {
  const uint64_t seed = nz ? nz->seed : 0;
  const uint32_t step = nz ? (uint32_t)nz->step : 0;
  c->nz.seed = seed;
  c->nz.step = step;
  const int S = c->S, B = c->B;
  const bool need_eps = c->mode != BNN_MODE_LRT;
  // one launch for weight noise + sign words + planes of x when all of them are generated here (nothing injected)
  const bool fused_fo = c->fuse_x && c->mode == BNN_MODE_FLIPOUT && !g_dry && !(nz && (nz->eps_w || nz->sign_in || nz->sign_out));
  const bool fused_rad = c->fuse_x && c->radial && !g_dry && !(nz && (nz->eps_w || nz->radial_r));
  const bool fused = fused_fo || fused_rad;
  static thread_local StepInputsArgs SI;
  auto fill_inputs = [&](const SignGenArgs* SG, long ex) {
    SI = StepInputsArgs{};
    SI.eps = ws_f(p, p->o_eps); SI.P = p->P; SI.S = S; SI.eps_seed = seed + 0x9E37ull * c->s_base; SI.step = step;
    const int L = p->d.win_length;
    const size_t plane = (size_t)p->d.max_batch * L * 32;
    u16* xp = (u16*)((char*)p->bufs.workspace + p->o_xplanes);
    SI.x = c->fuse_x; SI.rows = (long)B * L; SI.L = L; SI.F = p->d.n_features;
    for (int k = 0; k < 4; ++k) SI.xp[k] = xp + k * plane;
    const bool f32p = p->d.prec == BNN_PREC_F32;   // fp32 plan: x | pooled x as fp32 [rows][20] (launch_tf_fwd's layout)
    if (f32p) {
      SI.xf[0] = (float*)xp;
      SI.xf[1] = (float*)xp + (size_t)p->d.max_batch * L * TF_XC;
    }
    unsigned nb = (unsigned)((((p->P + 3) / 4) * S + 255) / 256);
    if (fused_rad) {
      SI.rad_r = ws_f(p, p->o_radr); SI.n_sites = p->n_sites;
      nb += 1;
    }
    SI.b_x = nb;
    nb += (unsigned)((SI.rows * (f32p ? 5 : 4) + 255) / 256);   // 8 (bf16 planes) / 4 (fp32 planes) channels per thread
    if (SG) SI.sg = *SG;
    for (int e = 0; e < SI.sg.n; ++e) {
      SI.b_sg[e] = nb;
      nb += (unsigned)((ex * ((SI.sg.words[e] + 3) / 4) + 255) / 256);
    }
    SI.b_sg[SI.sg.n] = nb;
    if (SI.sg.n == 0) SI.b_sg[0] = nb;
    step_inputs_kernel<<<dim3(nb), dim3(256), 0, c->st>>>(SI);
    c->x_planes_ready = true;
  };
  if (need_eps) {
    if (nz && nz->eps_w) {
      c->eps_w = nz->eps_w;
    } else {
      float* e = ws_f(p, p->o_eps);
      const long n = ((p->P + 3) / 4) * S;
      // particle offset is folded into the step word for predict chunks
      if (!fused)
        gen_eps_w_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->st>>>(e, p->P, S, seed + 0x9E37ull * c->s_base, step);
      c->eps_w = e;
    }
  }
  if (c->radial) {
    if (nz && nz->radial_r) {
      c->rad_r = nz->radial_r;
    } else {
      float* r = ws_f(p, p->o_radr);
      if (fused_rad)
        fill_inputs(nullptr, 0);   // weight noise + radial distances + planes of x: one launch
      else
        gen_radial_r_kernel<<<dim3((S * p->n_sites + 255) / 256), dim3(256), 0, c->st>>>(r, p->n_sites, S, seed + 0x9E37ull * c->s_base, step);
      c->rad_r = r;
    }
    {
      SiteChunks C{};
      for (int i = 0; i < p->n_sites; ++i) C.start[i + 1] = C.start[i] + (int)((p->ptab.site[i].numel + SN_CHUNK - 1) / SN_CHUNK);
      C.total = C.start[p->n_sites];
      double* part = (double*)((char*)p->bufs.workspace + p->o_norm_part);
      site_norm_part_kernel<<<dim3(S * C.total), dim3(256), 0, c->st>>>(c->eps_w, p->P, p->ptab, C, part);
      site_norm_fin_kernel<<<dim3((S * p->n_sites + 255) / 256), dim3(256), 0, c->st>>>(part, p->ptab, C, S, ws_f(p, p->o_norms));
    }
  }
  // layer sign arrays live back to back: layer i at (sign_*_off * S*B) words
  uint32_t* si = (uint32_t*)ws_f(p, p->o_sign_in);
  uint32_t* so = (uint32_t*)ws_f(p, p->o_sign_out);
  c->nz.sign_in = si;
  c->nz.sign_out = so;
  if (c->mode == BNN_MODE_FLIPOUT) {
    const long ex = (long)S * B;
    SignGenArgs SG{};
    SG.S = S; SG.B = B; SG.Bglob = a->global_batch; SG.goff = a->global_batch_offset;
    SG.seed = seed; SG.step = step;
    int maxw4 = 1;
    for (int i = 0; i < p->n_layers; ++i) {
      const LayerDesc& l = p->layers[i];
      uint32_t* di = si + l.sign_in_off * ex;
      uint32_t* d_o = so + l.sign_out_off * ex;
      if (nz && nz->sign_in && nz->sign_in[i]) {
        const long n = ex * l.sign_in_words;
        pack_signs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->st>>>(nz->sign_in[i], di, ex, l.cin_img, l.sign_in_words);
      } else {
        SG.dst[SG.n] = di; SG.words[SG.n] = l.sign_in_words; SG.layer[SG.n] = i; SG.kind[SG.n] = NK_SIGN_IN;
        maxw4 = std::max(maxw4, (l.sign_in_words + 3) / 4);
        SG.n++;
      }
      if (nz && nz->sign_out && nz->sign_out[i]) {
        const long n = ex * l.sign_out_words;
        pack_signs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->st>>>(nz->sign_out[i], d_o, ex, l.cout, l.sign_out_words);
      } else {
        SG.dst[SG.n] = d_o; SG.words[SG.n] = l.sign_out_words; SG.layer[SG.n] = i; SG.kind[SG.n] = NK_SIGN_OUT;
        maxw4 = std::max(maxw4, (l.sign_out_words + 3) / 4);
        SG.n++;
      }
    }
    if (fused_fo) {
      fill_inputs(&SG, ex);
    } else if (SG.n > 0) {
      const long n = ex * maxw4;
      gen_signs_all_kernel<<<dim3((unsigned)((n + 255) / 256), SG.n), dim3(256), 0, c->st>>>(SG);
    }
  }
  c->nz.use_philox_lrt = 1;
  c->nz.examples = (long)S * B;
  for (int i = 0; i < BNN_MAX_LAYERS; ++i) c->nz.lrt_eps[i] = nullptr;
  if (c->mode == BNN_MODE_LRT && nz && nz->lrt_eps) {
    c->nz.use_philox_lrt = 0;
    for (int i = 0; i < p->n_layers; ++i) {
      c->nz.lrt_eps[i] = nz->lrt_eps[i];
      if (!nz->lrt_eps[i]) return fail(BNN_E_INVALID, "lrt_eps[%d] is null (inject all layers or none)", i);
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// sample weights
// ------------------------------------------------------------------------------------------
static int do_sample(BnnPlan* p, const BnnElboArgs* a, Ctx* c) {
  if (!p->acc_clean) HIP_TRY(hipMemsetAsync(ws_f(p, p->o_acc), 0, sizeof(double) * 2 * (p->d.max_particles + 1), c->st));
  p->acc_clean = false;
  PrepArgs A{};
  A.T = p->ptab;
  A.layers = (const LayerDesc*)((char*)p->bufs.workspace + p->o_layers);
  A.mu = p->bufs.mu;
  A.rho = p->bufs.rho;
  A.eps_w = c->eps_w;
  A.rad_r = c->rad_r;
  A.norms = ws_f(p, p->o_norms);
  A.mode = c->mode;
  A.S = c->S;
  A.want_t = c->train ? 1 : 0;
  char* w = (char*)p->bufs.workspace;
  A.a_hi = w + p->o_a_hi;
  A.a_lo = p->d.prec == BNN_PREC_BF16X3 ? w + p->o_a_lo : nullptr;
  A.b = w + p->o_b;
  A.at = w + p->o_at;
  A.bt = w + p->o_bt;
  A.slot_stride = p->img_total;
  A.slott_stride = p->imgt_total;
  A.bias_a = ws_f(p, p->o_bias_a);
  A.bias_b = ws_f(p, p->o_bias_b);
  A.bias_total = p->bias_total;
  A.kl_acc = (double*)(w + p->o_acc);
  A.prior_loc = (float)a->prior_loc;
  A.prior_scale = (float)a->prior_scale;
  ProfScope ps_(&p->prof, PK_SAMPLE, 0, c->st);
  // the wide dense layer behind nn.Flatten goes through the tile kernel (16-byte image stores instead of 2- / 4-byte ones)
  int flat_layer = -1;
  const bool bfp = p->d.prec == BNN_PREC_BF16X3;
  if (c->mode != BNN_MODE_LRT)
    for (int i = 0; i < p->n_layers; ++i) {
      const LayerDesc& l = p->layers[i];
      if (!l.is_conv && l.taps == 1 && l.cmap == CM_FLATTEN && l.cmap_a % PF_TC == 0 && l.cout % PF_TN == 0 &&
          l.cin == l.cmap_a * l.cmap_b && l.w_off % 8 == 0 && l.wt_off % 8 == 0 && l.KP % 8 == 0 && l.KPt % 8 == 0 &&
          p->img_total % 8 == 0 && p->imgt_total % 8 == 0 && l.cmap_a == 80 && l.cmap_b == 30)
        flat_layer = i;
    }
  if (flat_layer >= 0) {
    const LayerDesc& l = p->layers[flat_layer];
    const SiteDesc& sd = p->ptab.site[l.site_w];
    A.skip_lo = sd.off;
    A.skip_hi = sd.off + sd.numel;
    const int lds = bfp ? 2 * PF_TN * PF_TC * l.cmap_b * 2 : PF_TN * PF_TC * l.cmap_b * 4;
    const unsigned tiles = (unsigned)((l.cout / PF_TN) * (l.cmap_a / PF_TC));
    const unsigned ny = (unsigned)(c->S + (c->mode == BNN_MODE_FLIPOUT ? 1 : 0));
    const unsigned rest = (unsigned)((p->P - sd.numel + 255) / 256);
    if (bfp) prep_fused_kernel<PrecBF, 80, 30><<<dim3(tiles * ny + rest), dim3(256), lds, c->st>>>(A, flat_layer, l.site_w, tiles, ny);
    else prep_fused_kernel<PrecF32, 80, 30><<<dim3(tiles * ny + rest), dim3(256), lds, c->st>>>(A, flat_layer, l.site_w, tiles, ny);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // 1024-thread workgroups: four times fewer fp64 atomics on the KL sum
  const unsigned grid = (unsigned)((p->P + 1023) / 1024);
  if (p->d.prec == BNN_PREC_F32)
    prep_weights_kernel<PrecF32><<<dim3(grid), dim3(1024), 0, c->st>>>(A);
  else
    prep_weights_kernel<PrecBF><<<dim3(grid), dim3(1024), 0, c->st>>>(A);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// group launches
// ------------------------------------------------------------------------------------------
static void fill_group_args(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, int gi, const float* x, GroupArgs* A) {
  *A = GroupArgs{};
  A->g = p->groups[gi];
  A->cg.S = c->S;
  A->cg.B = c->B;
  A->cg.Bglob = a ? a->global_batch : c->B;
  A->cg.goff = a ? a->global_batch_offset : 0;
  if (A->cg.Bglob < c->B) A->cg.Bglob = c->B;
  A->cg.per_particle = A->g.is_dense ? (c->B + TILE_ROWS - 1) / TILE_ROWS : c->B;
  A->cg.nwin = A->cg.per_particle * c->S;
  char* w = (char*)p->bufs.workspace;
  const bool per_particle_a = (c->mode == BNN_MODE_NORMAL || c->mode == BNN_MODE_RADIAL);
  const bool per_particle_b = (c->mode == BNN_MODE_FLIPOUT);
  A->ws.a_hi = w + p->o_a_hi;
  A->ws.a_lo = w + p->o_a_lo;
  A->ws.b = w + p->o_b;
  A->ws.at = w + p->o_at;
  A->ws.bt = w + p->o_bt;
  A->ws.bias_a = ws_f(p, p->o_bias_a);
  A->ws.bias_b = ws_f(p, p->o_bias_b);
  A->ws.slot_stride_a = per_particle_a ? p->img_total : 0;
  A->ws.slot_stride_b = per_particle_b ? p->img_total : 0;
  A->ws.slott_stride_a = per_particle_a ? p->imgt_total : 0;
  A->ws.slott_stride_b = per_particle_b ? p->imgt_total : 0;
  A->ws.bias_stride_a = (c->mode == BNN_MODE_LRT) ? 0 : p->bias_total;
  A->ws.bias_total = p->bias_total;
  A->nz = c->nz;
  for (int t = 0; t < 10; ++t) {
    if (p->tens[t].ctot == 0) continue;
    A->t[t] = tens_ref(p, t, 0);
    A->t[t + T_GRAD] = tens_ref(p, t, 1);
    A->t[t + T_Q] = tens_ref(p, t, 2);
  }
  A->t[T_X] = TensorRef{const_cast<float*>(x), nullptr, p->x_ctot, TF_F32};
  A->t[T_POOLGRAD] = TensorRef{ws_f(p, p->o_poolgrad), nullptr, 128, p->d.prec == BNN_PREC_BF16X3 ? TF_BF16 : TF_F32};
  A->amax = nullptr;
  if (p->d.prec == BNN_PREC_BF16X3 && p->d.net == BNN_NET_INCEPTION && c->train)   // no backward, no codes (evaluate / predict)
    for (int b = 0; b < A->g.n_branch; ++b)
      if (A->g.br[b].pool && A->g.br[b].dx_t >= 0) A->amax = (unsigned char*)w + p->o_amax;
  A->layers = (const LayerDesc*)(w + p->o_layers);
  A->gw_a = ws_f(p, p->o_gw_a);
  A->gw_b = ws_f(p, p->o_gw_b);
  A->gb_a = ws_f(p, p->o_gb_a);
  A->gb_b = ws_f(p, p->o_gb_b);
  A->gw_stride = p->img_total;
  A->gb_stride = p->bias_total;
}

static int img_bytes(int ch, bool bf) { return (IMG_ROWS * img_row_stride(ch, bf) * (bf ? 2 : 4) + 15) & ~15; }

template <class P>
static int launch_fwd(const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  ProfScope ps_(pf, PK_FWD, gi, st);
  ps_.name("group_fwd_kernel<%s, %d>", P::BF ? "PrecBF" : "PrecF32", em);
  const int nimg = P::BF ? 3 : 2;
  A.lds_per_wave = nimg * img_bytes(DENSE_CHUNK, P::BF);
  const int lds = 4 * A.lds_per_wave;
  const unsigned grid = (unsigned)std::min((A.cg.nwin * A.g.n_branch + 3) / 4, 2048);   // items = (window, branch)
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(group_fwd_kernel<P, EM_PLAIN>, lds));
    group_fwd_kernel<P, EM_PLAIN><<<dim3(grid), dim3(256), lds, st>>>(A);
  } else if (em == EM_LRT) {
    BNN_TRY(set_lds(group_fwd_kernel<P, EM_LRT>, lds));
    group_fwd_kernel<P, EM_LRT><<<dim3(grid), dim3(256), lds, st>>>(A);
  } else {
    BNN_TRY(set_lds(group_fwd_kernel<P, EM_FLIPOUT>, lds));
    group_fwd_kernel<P, EM_FLIPOUT><<<dim3(grid), dim3(256), lds, st>>>(A);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

template <class P>
static int launch_dx(const GroupArgs& A0, int em, int pool_sel, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  ProfScope ps_(pf, PK_DX, gi, st);
  ps_.name("group_dx_kernel<%s, %d>", P::BF ? "PrecBF" : "PrecF32", em);
  A.pool_sel = pool_sel;
  A.lds_per_wave = 2 * img_bytes(DENSE_CHUNK, P::BF);
  const int lds = 4 * A.lds_per_wave;
  const unsigned grid = (unsigned)std::min((A.cg.nwin + 3) / 4, 2048);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(group_dx_kernel<P, EM_PLAIN>, lds));
    group_dx_kernel<P, EM_PLAIN><<<dim3(grid), dim3(256), lds, st>>>(A);
  } else if (em == EM_LRT) {
    BNN_TRY(set_lds(group_dx_kernel<P, EM_LRT>, lds));
    group_dx_kernel<P, EM_LRT><<<dim3(grid), dim3(256), lds, st>>>(A);
  } else {
    BNN_TRY(set_lds(group_dx_kernel<P, EM_FLIPOUT>, lds));
    group_dx_kernel<P, EM_FLIPOUT><<<dim3(grid), dim3(256), lds, st>>>(A);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

template <class P, int NW>
static int launch_dw(const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  ProfScope ps_(pf, PK_DW, gi, st);
  ps_.name("group_dw_kernel<%s, %d, %d>", P::BF ? "PrecBF" : "PrecF32", em, NW);
  A.lds_per_wave = 2 * img_bytes(64, P::BF) + 2 * img_bytes(DENSE_CHUNK, P::BF);
  const int lds = NW * A.lds_per_wave;
  int njobs = 0;
  for (int b = 0; b < A.g.n_branch; ++b) njobs += (A.g.br[b].cin_p + DENSE_CHUNK - 1) / DENSE_CHUNK;
  const int max_split = std::max(1, (A.cg.per_particle + NW - 1) / NW);
  int nsplit = std::max(1, 768 / std::max(1, njobs * A.cg.S));
  nsplit = std::min(nsplit, max_split);
  A.nsplit = nsplit;
  const unsigned grid = (unsigned)(njobs * A.cg.S * nsplit);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(group_dw_kernel<P, EM_PLAIN, NW>, lds));
    group_dw_kernel<P, EM_PLAIN, NW><<<dim3(grid), dim3(NW * 64), lds, st>>>(A);
  } else if (em == EM_LRT) {
    BNN_TRY(set_lds(group_dw_kernel<P, EM_LRT, NW>, lds));
    group_dw_kernel<P, EM_LRT, NW><<<dim3(grid), dim3(NW * 64), lds, st>>>(A);
  } else {
    BNN_TRY(set_lds(group_dw_kernel<P, EM_FLIPOUT, NW>, lds));
    group_dw_kernel<P, EM_FLIPOUT, NW><<<dim3(grid), dim3(NW * 64), lds, st>>>(A);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- bf16-plane kernels of the dense layers (kernels_conv_bf.h, kernels_dense_fwd.h) ----
static int check_slab_slots(const BnnPlan* p, const Ctx* c, int g, int nsplit) {
  if ((long)c->S * nsplit > p->slab_slots[g])
    return fail(BNN_E_INVALID, "dW partial images: %d particles x %d splits exceed the %d slabs of this plan", c->S, nsplit, p->slab_slots[g]);
  return 0;
}
static int launch_dense_dw_bf(const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  const BranchDesc& br = A.g.br[0];
  const int nchunk = (br.cin_p + DN_CH - 1) / DN_CH;
  int nsplit = std::max(1, 384 / std::max(1, A.cg.S * nchunk));
  nsplit = std::min(nsplit, A.cg.per_particle);
  const int lds = (2 * DN_ROWS * (DN_CH + 8) + 2 * DN_ROWS * (64 + 8)) * 2;
  const unsigned grid = (unsigned)(A.cg.S * nchunk * nsplit);
  ProfScope ps_(pf, PK_DW, gi, st);
  ps_.name("dense_dw_bf_kernel<%d>", em);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) dense_dw_bf_kernel<EM_PLAIN><<<dim3(grid), dim3(512), lds, st>>>(A, nchunk, nsplit);
  else if (em == EM_LRT) dense_dw_bf_kernel<EM_LRT><<<dim3(grid), dim3(512), lds, st>>>(A, nchunk, nsplit);
  else dense_dw_bf_kernel<EM_FLIPOUT><<<dim3(grid), dim3(512), lds, st>>>(A, nchunk, nsplit);
  HIP_TRY(hipGetLastError());
  return 0;
}

// dense layers that fit the role-specialised kernel: one branch, bf16-plane input, cin multiple of 32
static bool dense_dma_ok(const GroupArgs& A) {
  if (!A.g.is_dense || A.g.n_branch != 1 || A.g.in_bcast) return false;
  const BranchDesc& br = A.g.br[0];
  if (A.t[A.g.in_t].fmt != TF_BF16 || A.t[A.g.in_t].lo == nullptr) return false;
  return (br.cin_p % 32) == 0 && br.ntiles <= 4 && br.cin_real == br.cin_p && br.in_off == 0 && (A.t[A.g.in_t].ctot % 8) == 0;
}

// one-branch dense forward, 6-wave workgroups without K split (kernels_dense_fwd.h)
static int launch_dense_fwd2(const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  DenseFwd2Plan F{};
  const BranchDesc& br = A.g.br[0];
  F.ntile = br.ntiles;
  F.nchunk = (br.cin_p + DF_CH - 1) / DF_CH;
  if (F.ntile > DF_NC) return fail(BNN_E_INVALID, "dense fwd: %d n-tiles", F.ntile);
  const int slot_bytes = 2 * DF_ROWS * DF_CH * 2 + DF_ROWS * 4 * 4;
  static_assert(DF_ROWS * DF_CH * 2 == 8 * 1024, "a chunk plane is exactly the 8 DMA instructions a loader issues for it");
  static_assert((DF_SLOTS - 2) * 9 <= 49 && DF_SLOTS >= 3, "chunks in flight: counted-wait range / slot ring");
  const int lds = DF_SLOTS * slot_bytes + (em == EM_LRT ? DF_ROWS * DF_CH * 2 : 0) + (em == EM_FLIPOUT ? 4096 : 0);
  const unsigned grid = (unsigned)(((A.cg.nwin + 7) / 8) * 8);   // XCD-aware window order inside the kernel
  ProfScope ps_(pf, PK_FWD, gi, st);
  ps_.name("dense_fwd2_kernel<%d>", em);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(dense_fwd2_kernel<EM_PLAIN>, lds));
    dense_fwd2_kernel<EM_PLAIN><<<dim3(grid), dim3(DF_NW * 64), lds, st>>>(A, F);
  } else if (em == EM_LRT) {
    BNN_TRY(set_lds(dense_fwd2_kernel<EM_LRT>, lds));
    dense_fwd2_kernel<EM_LRT><<<dim3(grid), dim3(DF_NW * 64), lds, st>>>(A, F);
  } else {
    BNN_TRY(set_lds(dense_fwd2_kernel<EM_FLIPOUT>, lds));
    dense_fwd2_kernel<EM_FLIPOUT><<<dim3(grid), dim3(DF_NW * 64), lds, st>>>(A, F);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// the wide dense layer of the Inception net: K-split, weight-stationary (kernels_dense_ks.h)
static bool dense_ks_ok(const BnnPlan* p, const GroupArgs& A, int em) {
  if (em == EM_LRT || p->o_dks == 0 || !dense_dma_ok(A)) return false;
  const BranchDesc& br = A.g.br[0];
  return br.cout == 64 && br.ntiles == 4 && br.cin_p >= 4 * DK_CH && (long)A.cg.S * A.cg.B <= p->dks_rows &&
         (size_t)A.cg.S * A.cg.B * A.t[A.g.in_t].ctot * 2 < ((size_t)1 << 32);
}

// the last layer Linear(64, 2) of the Inception net can ride on the K-split dense kernels' fin / head launches
static bool last_fused_ok(const BnnPlan* p, int em, int gi, bool allow_lrt = false) {
  if ((em == EM_LRT && !allow_lrt) || p->o_dks == 0 || gi + 2 != p->n_groups) return false;
  const GroupDesc& g = p->groups[gi + 1];
  const BranchDesc& br = g.br[0];
  const LayerDesc& ly = p->layers[br.layer];
  return g.is_dense && g.n_branch == 1 && br.cout == 2 && br.cin_p == 64 && br.n_off == 0 && br.in_off == 0 && !br.relu &&
         ly.KP == 64 && g.in_t == p->groups[gi].br[0].out_t && p->groups[gi].br[0].out_off == 0 && br.out_t == p->z_t &&
         p->tens[g.in_t].ctot == 64 && ly.sign_in_words == 2 && ly.sign_out_words == 1;
}

static int launch_dense_ks_fwd(BnnPlan* p, const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi, bool fuse_last) {
  GroupArgs A = A0;
  const BranchDesc& br = A.g.br[0];
  const LayerDesc& ly = p->layers[br.layer];
  DenseKsPlan F{};
  F.nchunk = (br.cin_p + DK_CH - 1) / DK_CH;
  F.nsplit = std::min(std::max(1, 512 / std::max(1, A.cg.S * F.nchunk)), A.cg.per_particle);
  F.slab = ws_f(p, p->o_dks);
  F.slab_stride = (long)A.cg.S * A.cg.B * 64;
  F.ly = ly;
  static_assert(DK_PLANE == 8 * 1024 && (DK_SLOTS & (DK_SLOTS - 1)) == 0, "a step plane is exactly the 8 DMA instructions a loader issues for it");
  static_assert((DK_SLOTS - 2) * 10 <= 49, "counted-wait range");
  static_assert(2 * DK_FWD_LDS <= 160 * 1024, "two workgroups per CU");
  if (ly.sign_out_words > 2 || (br.cin_p & 31)) return fail(BNN_E_INVALID, "dense K-split forward: layer shape");
  const int total = A.cg.S * F.nchunk * F.nsplit;
  const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
  ProfScope ps_(pf, PK_FWD, gi, st);
  ps_.name("dense_ks_fwd_kernel<%d>", em);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(dense_ks_fwd_kernel<EM_PLAIN>, DK_FWD_LDS));
    dense_ks_fwd_kernel<EM_PLAIN><<<dim3(grid), dim3(DK_NW * 64), DK_FWD_LDS, st>>>(A, F);
  } else {
    BNN_TRY(set_lds(dense_ks_fwd_kernel<EM_FLIPOUT>, DK_FWD_LDS));
    dense_ks_fwd_kernel<EM_FLIPOUT><<<dim3(grid), dim3(DK_NW * 64), DK_FWD_LDS, st>>>(A, F);
  }
  DenseKsFinArgs R{};
  R.slab = F.slab;
  R.slab_stride = F.slab_stride;
  R.nchunk = F.nchunk;
  R.rows = A.cg.S * A.cg.B;
  R.B = A.cg.B;
  R.bias = A.ws.bias_a + ly.bias_off + br.n_off;
  R.bias_stride = A.ws.bias_stride_a;
  R.relu = br.relu;
  R.out = A.t[br.out_t];
  R.out_off = br.out_off;
  if (fuse_last) {
    const LayerDesc& l2 = p->layers[p->groups[gi + 1].br[0].layer];
    R.fuse2 = 1;
    R.w2_hi = (const u16*)A.ws.a_hi + l2.w_off;
    R.w2_lo = (const u16*)A.ws.a_lo + l2.w_off;
    R.w2_b = (const u16*)A.ws.b + l2.w_off;
    R.w2_stride_a = A.ws.slot_stride_a;
    R.w2_stride_b = A.ws.slot_stride_b;
    R.w2_KP = l2.KP;
    R.b2 = A.ws.bias_a + l2.bias_off;
    R.sg_in = A.nz.sign_in + l2.sign_in_off * A.nz.examples;
    R.sg_out = A.nz.sign_out + l2.sign_out_off * A.nz.examples;
    R.siw = l2.sign_in_words;
    R.sow = l2.sign_out_words;
    R.z = tens_ptr(p, p->z_t, 0);
    R.g2_a = A.gw_a + l2.w_off;
    R.g2_b = A.gw_b + l2.w_off;
    R.g2_ba = A.gb_a + l2.bias_off;
    R.g2_stride = A.gw_stride;
    R.g2_bstride = A.gb_stride;
    R.S = A.cg.S;
  }
  const unsigned fgrid = (unsigned)((R.rows * 16 + 255) / 256);
  if (em == EM_PLAIN) dense_ks_fin_kernel<EM_PLAIN><<<dim3(fgrid), dim3(256), 0, st>>>(R);
  else dense_ks_fin_kernel<EM_FLIPOUT><<<dim3(fgrid), dim3(256), 0, st>>>(R);
  HIP_TRY(hipGetLastError());
  return 0;
}

// window splits per (particle, chunk) of the K-split dense backward: 1 = every gradient element is STORED by exactly one
// workgroup (no zero fill needed: prepare_fused_tail relies on this very rule), else float atomics onto zeroed images
static int dense_ks_bwd_nsplit(int S, int cin_p, int per_particle) {
  const int pairs = S * ((cin_p + DB_CH - 1) / DB_CH);
  return pairs >= 128 ? 1 : std::min(per_particle, (256 + pairs - 1) / pairs);
}

// dX + dW of the wide dense layer in one launch (kernels_dense_ks.h)
static int launch_dense_ks_bwd(BnnPlan* p, const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  const BranchDesc& br = A.g.br[0];
  DenseKsPlan F{};
  F.nchunk = (br.cin_p + DB_CH - 1) / DB_CH;
  F.nsplit = dense_ks_bwd_nsplit(A.cg.S, br.cin_p, A.cg.per_particle);
  const int pairs = A.cg.S * F.nchunk;
  F.ly = p->layers[br.layer];
  F.mask_x = 1;   // the layer's input (ACT2) is the concatenation of ReLU outputs (inception.py:118-131)
  static_assert(DB_LDS <= 160 * 1024 && DB_WAVES <= 16, "one workgroup per CU");
  static_assert(DB_AHEAD * 10 <= 49 && DB_RING >= DB_AHEAD + 2, "counted-wait range / ring: windows k-1 .. k+4 are live");
  if ((br.cin_p & 15) || (A.t[br.dx_t].ctot & 3) || p->layers[br.layer].KPt < 64)
    return fail(BNN_E_INVALID, "dense K-split backward: layer shape");
  const int total = pairs * F.nsplit;
  const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
  ProfScope ps_(pf, PK_DX, gi, st);
  ps_.name("dense_ks_bwd_kernel<%d>", em);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(dense_ks_bwd_kernel<EM_PLAIN>, DB_LDS));
    dense_ks_bwd_kernel<EM_PLAIN><<<dim3(grid), dim3(DB_WAVES * 64), DB_LDS, st>>>(A, F);
  } else {
    BNN_TRY(set_lds(dense_ks_bwd_kernel<EM_FLIPOUT>, DB_LDS));
    dense_ks_bwd_kernel<EM_FLIPOUT><<<dim3(grid), dim3(DB_WAVES * 64), DB_LDS, st>>>(A, F);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_dense_dx_bf(const GroupArgs& A0, int em, hipStream_t st, Prof* pf, int gi) {
  GroupArgs A = A0;
  const int zw = rup(A.g.br[0].cout, 32);
  const int lds = 2 * DN_ROWS * (zw + 8) * 2;
  const unsigned grid = (unsigned)A.cg.nwin;
  ProfScope ps_(pf, PK_DX, gi, st);
  ps_.name("dense_dx_bf_kernel<%d>", em);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) dense_dx_bf_kernel<EM_PLAIN><<<dim3(grid), dim3(DDX_WAVES * 64), lds, st>>>(A);
  else if (em == EM_LRT) dense_dx_bf_kernel<EM_LRT><<<dim3(grid), dim3(DDX_WAVES * 64), lds, st>>>(A);
  else dense_dx_bf_kernel<EM_FLIPOUT><<<dim3(grid), dim3(DDX_WAVES * 64), lds, st>>>(A);
  HIP_TRY(hipGetLastError());
  return 0;
}

// fused LRT kernels of the Linear net (kernels_mlp.h)
static bool mlp_ok(const BnnPlan* p, const Ctx* c) { return p->mlp && c->em == EM_LRT; }

static void fill_mlp_plan(const BnnPlan* p, MlpPlan* M) {
  const int tid[5] = {TI_H, TI_H2, TI_H3, TI_H4, TI_Z};
  for (int l = 0; l < 5; ++l) {
    M->ly[l] = p->layers[l];
    M->h[l] = tens_ref(p, tid[l], 0);
    M->g[l] = tens_ref(p, tid[l], 1);
    M->q[l] = tens_ref(p, tid[l], 2);
  }
  M->xhi = (u16*)((char*)p->bufs.workspace + p->o_mlp_x);
  M->dz4 = (u16*)((char*)p->bufs.workspace + p->o_mlp_dz4);
  M->dz4_plane = p->cap_windows * 8;
}

static int launch_mlp_fwd(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const float* x) {
  GroupArgs A;
  fill_group_args(p, a, c, 0, x, &A);
  static thread_local MlpPlan M;
  fill_mlp_plan(p, &M);
  static_assert(ML0_LDS <= 160 * 1024 && ML14_LDS <= 160 * 1024 && MX_LDS <= 160 * 1024 && MW_LDS <= 160 * 1024, "LDS budgets");
  if (A.t[T_X].ctot != 540 || (A.t[T_X].ctot & 3)) return fail(BNN_E_INVALID, "fused Linear net: %d input features", A.t[T_X].ctot);
  const int nwf = c->S * ((c->B + MLF_ROWS - 1) / MLF_ROWS);   // forward windows (the dW kernel walks ML_ROWS-row windows)
  {
    ProfScope ps_(&p->prof, PK_FWD, 0, c->st);
    ps_.name("mlp_l0_kernel");
    if (!g_dry) {
      BNN_TRY(set_lds(mlp_l0_kernel, ML0_LDS));
      mlp_l0_kernel<<<dim3((unsigned)nwf * 4), dim3(256), ML0_LDS, c->st>>>(A, M);
    }
  }
  {
    ProfScope ps_(&p->prof, PK_FWD, 1, c->st);
    ps_.name("mlp_l14_kernel");
    if (!g_dry) {
      BNN_TRY(set_lds(mlp_l14_kernel, ML14_LDS));
      mlp_l14_kernel<<<dim3((unsigned)nwf), dim3(512), ML14_LDS, c->st>>>(A, M);
    }
  }
  if (!g_dry) HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_mlp_bwd(BnnPlan* p, const BnnElboArgs* a, const Ctx* c) {
  GroupArgs A;
  fill_group_args(p, a, c, 0, a->x, &A);
  static thread_local MlpPlan M;
  fill_mlp_plan(p, &M);
  {
    // the gradient images of this call are zeroed by the dX kernel (the dW kernel adds into them)
    const size_t gwb = (size_t)c->S * p->img_total * 4, gbb = (size_t)c->S * p->bias_total * 4;
    float* zp[4] = {ws_f(p, p->o_gw_a), ws_f(p, p->o_gw_b), ws_f(p, p->o_gb_a), ws_f(p, p->o_gb_b)};
    const size_t zb[4] = {gwb, gwb, gbb, gbb};
    for (int k = 0; k < 4; ++k) {
      M.zero_p[k] = zp[k];
      M.zero_n[k] = (long)((zb[k] + 15) / 16);   // regions are 256-byte aligned and padded
    }
  }
  M.fuse_head = c->head_fused ? 1 : 0;
  if (c->head_fused) {
    M.head = HeadArgs{};
    M.head.z = tens_ptr(p, p->z_t, 0);
    M.head.y = a->y;
    M.head.dz = nullptr;
    M.head.preds = c->head_preds;
    M.head.ll_acc = (double*)((char*)p->bufs.workspace + p->o_acc) + (p->d.max_particles + 1);
    M.head.S = c->S;
    M.head.B = c->B;
    M.head.with_obs = a->with_obs && a->y;
    M.head.objective = c->objective;
  }
  {
    ProfScope ps_(&p->prof, PK_DX, 1, c->st);
    ps_.name("mlp_dx_kernel");
    if (!g_dry) {
      BNN_TRY(set_lds(mlp_dx_kernel, MX_LDS));
      mlp_dx_kernel<<<dim3((unsigned)(c->S * ((c->B + MLF_ROWS - 1) / MLF_ROWS))), dim3(512), MX_LDS, c->st>>>(A, M);
    }
  }
  static thread_local MlpDwPlan D;
  D = MlpDwPlan{};
  for (int l = 0; l < 5; ++l) {
    const LayerDesc& ly = p->layers[l];
    const int K = ly.KP;
    for (int n0 = 0; n0 < ly.cout; n0 += 64)
      for (int c0 = 0; c0 < K; c0 += 128) {
        if (D.njobs >= 30) return fail(BNN_E_INVALID, "fused Linear net: dW job table");
        MlpDwJob& J = D.job[D.njobs++];
        if (l == 0) {
          J.x = M.xhi + c0; J.x_ctot = ML_K0; J.x_bcast = 1;
        } else {
          J.x = (const u16*)M.h[l - 1].p + c0; J.x_ctot = M.h[l - 1].ctot; J.x_bcast = 0;
        }
        if (l < 4) {
          J.dz = (const u16*)M.g[l].p + n0; J.dz2 = (const u16*)M.q[l].p + n0; J.z_ctot = M.g[l].ctot;
        } else {
          J.dz = M.dz4; J.dz2 = M.dz4 + M.dz4_plane; J.z_ctot = 8;
        }
        J.gwa = A.gw_a + ly.w_off + (long)n0 * ly.KP + c0;
        J.gwb = A.gw_b + ly.w_off + (long)n0 * ly.KP + c0;
        J.gba = c0 == 0 ? A.gb_a + ly.bias_off + n0 : nullptr;
        J.gbb = c0 == 0 ? A.gb_b + ly.bias_off + n0 : nullptr;
        J.cw = std::min(128, K - c0);
        J.cout = std::min(64, ly.cout - n0);
        J.KP = ly.KP;
        if ((J.cw & 15) || (J.x_ctot & 7) || (J.z_ctot & 7)) return fail(BNN_E_INVALID, "fused Linear net: dW job shape");
      }
  }
  D.nsplit = std::max(1, std::min(A.cg.nwin, 128 / std::max(1, D.njobs)));
  static_assert(MW_AHEAD * 8 <= 49 && MW_RING >= MW_AHEAD + 2, "counted-wait range / ring");
  {
    ProfScope ps_(&p->prof, PK_DW, 0, c->st);
    ps_.name("mlp_dw_kernel");
    if (!g_dry) {
      BNN_TRY(set_lds(mlp_dw_kernel, MW_LDS));
      mlp_dw_kernel<<<dim3((unsigned)(D.njobs * D.nsplit)), dim3(MW_WAVES * 64), MW_LDS, c->st>>>(A.cg, D);
    }
  }
  if (!g_dry) HIP_TRY(hipGetLastError());
  return 0;
}

// fused conv trunk (kernels_trunk.h): groups 0..2 of the Inception net in one launch
static bool trunk_ok(const BnnPlan* p, const Ctx* c) {
  return p->d.prec == BNN_PREC_BF16X3 && p->d.net == BNN_NET_INCEPTION && c->em != EM_LRT;
}

static int launch_trunk_fwd(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const float* x) {
  const int L = p->d.win_length;
  const long rows = (long)c->B * L;
  const size_t plane = (size_t)p->d.max_batch * L * 32;
  u16* xp = (u16*)((char*)p->bufs.workspace + p->o_xplanes);
  if (!g_dry && !c->x_planes_ready) {
    x_planes4_kernel<<<dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, c->st>>>(x, xp, xp + plane, xp + 2 * plane,
                                                                                        xp + 3 * plane, rows, L, p->d.n_features);
    HIP_TRY(hipGetLastError());
  }
  GroupArgs G;
  fill_group_args(p, a, c, 0, x, &G);
  TrunkArgs T{};
  for (int k = 0; k < 4; ++k) T.xp[k] = xp + k * plane;
  T.ws = G.ws;
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  const TensorRef a1 = tens_ref(p, TI_ACT1, 0), md = tens_ref(p, TI_MID, 0), a2 = tens_ref(p, TI_ACT2, 0);
  T.act1_hi = c->train ? (u16*)a1.p : nullptr;
  T.mid_hi = c->train ? (u16*)md.p : nullptr;
  T.act2_hi = (u16*)a2.p;
  T.act2_lo = (u16*)a2.lo;
  T.amax = c->train ? (unsigned char*)p->bufs.workspace + p->o_amax : nullptr;
  T.m_act1 = c->train ? (unsigned char*)p->bufs.workspace + p->o_mact1 : nullptr;
  T.m_mid = c->train ? (unsigned char*)p->bufs.workspace + p->o_mmid : nullptr;
  T.S = c->S;
  T.B = c->B;
  T.L = L;
  T.nsplit = std::max(1, std::min(c->B, 256 / std::max(1, c->S)));
  static_assert(TR_LDS <= 160 * 1024 && TX_LDS <= 160 * 1024 && TW2A_LDS <= 160 * 1024 && 2 * TW2B_LDS <= 160 * 1024, "LDS budgets");
  static_assert(2 * (2 * ((30 * 16 + 63) / 64) + (30 * 8 + 63) / 64) <= 49, "trunk dX: DMA instructions of two steps within the counted-wait range");
  if (L > 30 || L < 1 || p->d.n_features != 18) return fail(BNN_E_INVALID, "trunk kernels: windows of 1..30 rows x 18 features");
  if ((long)c->S * c->B * L * 256 >= (1L << 32))
    return fail(BNN_E_INVALID, "trunk kernels address rows with 32-bit byte offsets: S*B*L = %ld rows exceed 2^24", (long)c->S * c->B * L);
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_FWD, 0, c->st);
  ps_.name("trunk_fwd_kernel<%d, %s>", c->em, c->train ? "true" : "false");
#define LAUNCH_TRUNK(EMV, TRV)                                                                  \
  do {                                                                                          \
    BNN_TRY(set_lds(trunk_fwd_kernel<EMV, TRV>, TR_LDS));                                       \
    trunk_fwd_kernel<EMV, TRV><<<dim3(grid), dim3(TR_THREADS), TR_LDS, c->st>>>(T);             \
  } while (0)
  BNN_DRY_RETURN();
  if (c->em == EM_FLIPOUT) {
    if (c->train) LAUNCH_TRUNK(EM_FLIPOUT, true); else LAUNCH_TRUNK(EM_FLIPOUT, false);
  } else {
    if (c->train) LAUNCH_TRUNK(EM_PLAIN, true); else LAUNCH_TRUNK(EM_PLAIN, false);
  }
#undef LAUNCH_TRUNK
  HIP_TRY(hipGetLastError());
  return 0;
}

// workgroups per particle of the trunk dW kernels (group 0: block 1, 1: block 2's 1x1 level, 2: its k3 / k5 level: small
// workgroups, two per CU); S * nsplit slabs are written and summed by grad_finalize_kernel
static int trunk_dw_nsplit(const Ctx* c, int g) { return std::max(1, std::min(c->B, (g == 2 ? 512 : 256) / std::max(1, c->S))); }

// dW of block 1 (kernels_trunk_dw.h): transposed-read tiles in registers across a particle's windows
static int launch_trunk_dw1(BnnPlan* p, const BnnElboArgs* a, const Ctx* c) {
  GroupArgs G;
  fill_group_args(p, a, c, 0, a->x, &G);
  TrunkDw1Args T{};
  const size_t plane = (size_t)p->d.max_batch * p->d.win_length * 32;
  const u16* xp = (const u16*)((char*)p->bufs.workspace + p->o_xplanes);
  T.x_hi = xp;
  T.xp_hi = xp + 2 * plane;
  T.g_act1 = (const u16*)tens_ref(p, TI_ACT1, 1).p;
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  T.gw_a = ws_f(p, p->o_slab_a[0]); T.gw_b = ws_f(p, p->o_slab_b[0]); T.gb_a = ws_f(p, p->o_slab_ba[0]);
  T.gw_stride = p->slab_stride; T.gb_stride = p->slab_bstride;
  T.S = c->S; T.B = c->B; T.L = p->d.win_length;
  T.nsplit = trunk_dw_nsplit(c, 0);
  BNN_TRY(check_slab_slots(p, c, 0, T.nsplit));
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_DW, 0, c->st);
  ps_.name("trunk_dw1_kernel<%d>", c->em);
  BNN_DRY_RETURN();
  if (c->em == EM_FLIPOUT) trunk_dw1_kernel<EM_FLIPOUT><<<dim3(grid), dim3(TW1_THREADS), TW1_LDS, c->st>>>(T);
  else trunk_dw1_kernel<EM_PLAIN><<<dim3(grid), dim3(TW1_THREADS), TW1_LDS, c->st>>>(T);
  HIP_TRY(hipGetLastError());
  return 0;
}

// dW of block 2 (kernels_trunk_dw.h): kind 0 = 1x1 level (layers 4, 5, 7, 9), kind 1 = k3 / k5 level (layers 6, 8)
static int launch_trunk_dw2(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, int kind) {
  GroupArgs G;
  fill_group_args(p, a, c, kind == 0 ? 1 : 2, a->x, &G);
  TrunkDw2Args T{};
  T.x_hi = (const u16*)tens_ref(p, kind == 0 ? TI_ACT1 : TI_MID, 0).p;
  T.g_mid = (const u16*)tens_ref(p, TI_MID, 1).p;
  T.g_act2 = (const u16*)tens_ref(p, TI_ACT2, 1).p;
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  const int sg = kind == 0 ? 1 : 2;
  T.gw_a = ws_f(p, p->o_slab_a[sg]); T.gw_b = ws_f(p, p->o_slab_b[sg]); T.gb_a = ws_f(p, p->o_slab_ba[sg]);
  T.gw_stride = p->slab_stride; T.gb_stride = p->slab_bstride;
  T.S = c->S; T.B = c->B; T.L = p->d.win_length;
  // kind 1 workgroups are small (6 waves, 52 KB of LDS): two share a CU
  T.nsplit = trunk_dw_nsplit(c, sg);
  BNN_TRY(check_slab_slots(p, c, sg, T.nsplit));
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_DW, kind == 0 ? 1 : 2, c->st);
  ps_.name(kind == 0 ? "trunk_dw2a_kernel<%d>" : "trunk_dw2b_kernel<%d>", c->em);
  BNN_DRY_RETURN();
  if (kind == 0) {
    if (c->em == EM_FLIPOUT) {
      BNN_TRY(set_lds(trunk_dw2a_kernel<EM_FLIPOUT>, TW2A_LDS));
      trunk_dw2a_kernel<EM_FLIPOUT><<<dim3(grid), dim3(TW2A_THREADS), TW2A_LDS, c->st>>>(T);
    } else {
      BNN_TRY(set_lds(trunk_dw2a_kernel<EM_PLAIN>, TW2A_LDS));
      trunk_dw2a_kernel<EM_PLAIN><<<dim3(grid), dim3(TW2A_THREADS), TW2A_LDS, c->st>>>(T);
    }
  } else {
    if (c->em == EM_FLIPOUT) trunk_dw2b_kernel<EM_FLIPOUT><<<dim3(grid), dim3(TW2B_THREADS), TW2B_LDS, c->st>>>(T);
    else trunk_dw2b_kernel<EM_PLAIN><<<dim3(grid), dim3(TW2B_THREADS), TW2B_LDS, c->st>>>(T);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// fused conv-trunk dX (kernels_trunk_bwd.h): dz of MID and of ACT1 from dY(ACT2), groups 2 and 1 in one launch
static int launch_trunk_dx(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, bool premasked) {
  GroupArgs G;
  fill_group_args(p, a, c, 1, a->x, &G);
  TrunkDxArgs T{};
  T.g_act2 = (u16*)tens_ref(p, TI_ACT2, 1).p;
  T.act2_hi = (const u16*)tens_ref(p, TI_ACT2, 0).p;
  T.amax = (const unsigned char*)p->bufs.workspace + p->o_amax;
  T.m_act1 = (const unsigned char*)p->bufs.workspace + p->o_mact1;
  T.m_mid = (const unsigned char*)p->bufs.workspace + p->o_mmid;
  T.g_mid = (u16*)tens_ref(p, TI_MID, 1).p;
  T.g_act1 = (u16*)tens_ref(p, TI_ACT1, 1).p;
  T.ws = G.ws;
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  T.S = c->S;
  T.B = c->B;
  T.L = p->d.win_length;
  T.nsplit = std::max(1, std::min(c->B, 256 / std::max(1, c->S)));
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_DX, 1, c->st);
  ps_.name("trunk_dx_kernel<%d, %s>", c->em, premasked ? "true" : "false");
  BNN_DRY_RETURN();
  if (c->em == EM_FLIPOUT) {
    if (premasked) {
      BNN_TRY(set_lds((trunk_dx_kernel<EM_FLIPOUT, true>), TX_LDS));
      trunk_dx_kernel<EM_FLIPOUT, true><<<dim3(grid), dim3(TX_THREADS), TX_LDS, c->st>>>(T);
    } else {
      BNN_TRY(set_lds((trunk_dx_kernel<EM_FLIPOUT, false>), TX_LDS));
      trunk_dx_kernel<EM_FLIPOUT, false><<<dim3(grid), dim3(TX_THREADS), TX_LDS, c->st>>>(T);
    }
  } else {
    if (premasked) {
      BNN_TRY(set_lds((trunk_dx_kernel<EM_PLAIN, true>), TX_LDS));
      trunk_dx_kernel<EM_PLAIN, true><<<dim3(grid), dim3(TX_THREADS), TX_LDS, c->st>>>(T);
    } else {
      BNN_TRY(set_lds((trunk_dx_kernel<EM_PLAIN, false>), TX_LDS));
      trunk_dx_kernel<EM_PLAIN, false><<<dim3(grid), dim3(TX_THREADS), TX_LDS, c->st>>>(T);
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static bool drop_on(const Ctx* c) { return c->drop && c->drop->p > 0.0; }
// MC-dropout runs on the fused fp32 kernels only: deterministic weights (plain estimator), one particle
static int check_dropout(const BnnPlan* p, const Ctx* c);

// fused fp32 conv trunk (kernels_f32.h): groups 0..2 of the Inception net in one launch, exact fp32 MFMA
static bool tf_ok(const BnnPlan* p, const Ctx* c) {
  return p->d.prec == BNN_PREC_F32 && p->d.net == BNN_NET_INCEPTION && p->d.win_length <= 30 && p->d.n_features == 18;
}
// which parts of the fused fp32 path exist for the call's estimator (LRT: built stage by stage; the generic per-group
// kernels take over where a stage is missing - they share the tensors' layout)
static bool tf_dw_ok(const BnnPlan* p, const Ctx* c) { return tf_ok(p, c); }
static bool tf_bwd_ok(const BnnPlan* p, const Ctx* c) { return tf_dw_ok(p, c); }   // every gradient element is stored (no fill needed)
static bool tf_dense_ok(const BnnPlan* p, const Ctx* c, bool bwd = true) { (void)bwd; return tf_ok(p, c); }

static int tf_check_tables(const BnnPlan* p) {
  for (int l = 0; l < 10; ++l) {
    const LayerDesc& ly = p->layers[l];
    if (ly.cin_img != tf_cimg(l) || ly.taps != tl_taps(l) || ly.cout != tl_cout(l) || ly.KP != rup(ly.taps * ly.cin_img, 32) ||
        (ly.w_off & 3) || (ly.wt_off & 3) || (ly.KPt & 3))
      return fail(BNN_E_INVALID, "fp32 trunk kernels: layer %d does not match the compiled geometry", l);
  }
  if ((p->img_total & 3) || (p->imgt_total & 3)) return fail(BNN_E_INVALID, "fp32 trunk kernels: image strides");
  return 0;
}

static int launch_tf_fwd(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const float* x) {
  const int L = p->d.win_length;
  const long rows = (long)c->B * L;
  float* xp = (float*)((char*)p->bufs.workspace + p->o_xplanes);
  const size_t plane = (size_t)p->d.max_batch * L * TF_XC;
  BNN_TRY(tf_check_tables(p));
  if (!g_dry && !c->x_planes_ready) {
    xf_planes_kernel<<<dim3((unsigned)((rows * 5 + 255) / 256)), dim3(256), 0, c->st>>>(x, xp, xp + plane, rows, L, p->d.n_features);
    HIP_TRY(hipGetLastError());
  }
  GroupArgs G;
  fill_group_args(p, a, c, 0, x, &G);
  TfArgs T{};
  T.xp[0] = xp;
  T.xp[1] = xp + plane;
  T.ws = G.ws;
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  T.act1 = c->train ? tens_ptr(p, TI_ACT1, 0) : nullptr;
  T.mid = c->train ? tens_ptr(p, TI_MID, 0) : nullptr;
  T.act2 = tens_ptr(p, TI_ACT2, 0);
  T.amax = c->train ? (unsigned char*)p->bufs.workspace + p->o_amax : nullptr;
  T.m_act1 = c->train ? (unsigned char*)p->bufs.workspace + p->o_mact1 : nullptr;
  T.m_mid = c->train ? (unsigned char*)p->bufs.workspace + p->o_mmid : nullptr;
  T.m_act2 = c->train ? (unsigned char*)p->bufs.workspace + p->o_mact2 : nullptr;
  T.nz = c->nz;
  T.cg = G.cg;
  T.q1 = c->train ? tens_ptr(p, TI_ACT1, 2) : nullptr;
  T.qm = c->train ? tens_ptr(p, TI_MID, 2) : nullptr;
  T.q2 = c->train ? tens_ptr(p, TI_ACT2, 2) : nullptr;
  T.S = c->S;
  T.B = c->B;
  T.L = L;
  T.nsplit = std::max(1, std::min(c->B, 256 / std::max(1, c->S)));
  static_assert(TF_LDS <= 160 * 1024, "LDS budget");
  if ((long)c->S * c->B * L * 512 >= (1L << 32))
    return fail(BNN_E_INVALID, "fp32 trunk kernels address rows with 32-bit byte offsets: S*B*L = %ld rows exceed 2^23", (long)c->S * c->B * L);
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_FWD, 0, c->st);
  ps_.name("tf_fwd_kernel<%d, %s, %s>", c->em, c->train ? "true" : "false", drop_on(c) ? "true" : "false");
#define LAUNCH_TF(EMV, TRV)                                                               \
  do {                                                                                    \
    BNN_TRY(set_lds(tf_fwd_kernel<EMV, TRV>, TF_LDS));                                    \
    tf_fwd_kernel<EMV, TRV><<<dim3(grid), dim3(TF_THREADS), TF_LDS, c->st>>>(T);          \
  } while (0)
  BNN_DRY_RETURN();
#if TF_STAMPS
  static unsigned long long* tf_dbg = nullptr;   // diagnostics build: stamps of the last launch -> gpurun_out/tf_stamps.bin
  const size_t tf_dbg_bytes = 8 * 48 * 16 * sizeof(unsigned long long);
  if (!tf_dbg) HIP_TRY(hipMalloc((void**)&tf_dbg, tf_dbg_bytes));
  HIP_TRY(hipMemsetAsync(tf_dbg, 0, tf_dbg_bytes, c->st));
  T.dbg = tf_dbg;
  struct TfDbgDump {
    unsigned long long* d; size_t n; hipStream_t st;
    ~TfDbgDump() {
      std::vector<char> h(n);
      if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), d, n, hipMemcpyDeviceToHost) != hipSuccess) return;
      if (FILE* f = fopen("gpurun_out/tf_stamps.bin", "wb")) { fwrite(h.data(), 1, n, f); fclose(f); }
    }
  } tf_dbg_dump{tf_dbg, tf_dbg_bytes, c->st};
#endif
  if (drop_on(c)) {
    T.drop_rate = (float)(c->drop->p / 4);
    T.drop_scale = (float)(1.0 / (1.0 - c->drop->p / 4));
    T.drop_seed = c->drop->seed;
    T.drop_step = (uint32_t)c->drop->step;
    T.keep1 = c->drop->keep_act1;
    T.keep2 = c->drop->keep_act2;
    if (c->train) {
      BNN_TRY(set_lds((tf_fwd_kernel<EM_PLAIN, true, true>), TF_LDS));
      tf_fwd_kernel<EM_PLAIN, true, true><<<dim3(grid), dim3(TF_THREADS), TF_LDS, c->st>>>(T);
    } else {
      BNN_TRY(set_lds((tf_fwd_kernel<EM_PLAIN, false, true>), TF_LDS));
      tf_fwd_kernel<EM_PLAIN, false, true><<<dim3(grid), dim3(TF_THREADS), TF_LDS, c->st>>>(T);
    }
  } else if (c->em == EM_FLIPOUT) {
    if (c->train) LAUNCH_TF(EM_FLIPOUT, true); else LAUNCH_TF(EM_FLIPOUT, false);
  } else if (c->em == EM_LRT) {
    if (c->train) LAUNCH_TF(EM_LRT, true); else LAUNCH_TF(EM_LRT, false);
  } else {
    if (c->train) LAUNCH_TF(EM_PLAIN, true); else LAUNCH_TF(EM_PLAIN, false);
  }
#undef LAUNCH_TF
  HIP_TRY(hipGetLastError());
  return 0;
}

static bool densef_ok(const BnnPlan* p, const Ctx* c, const GroupArgs& A, bool bwd);
static int launch_densef_fwd(BnnPlan* p, const GroupArgs& A, int em, hipStream_t st, Prof* pf, int gi, bool fuse_last, const BnnDropout* drop, bool train);

static int do_forward(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const float* x) {
  const bool bf = p->d.prec == BNN_PREC_BF16X3;
  const bool tf = tf_ok(p, c);
  const bool trunk = trunk_ok(p, c) || tf;
  BNN_TRY(check_dropout(p, c));
  p->fwd_fused_last = false;
  if (mlp_ok(p, c)) {
    BNN_TRY(launch_mlp_fwd(p, a, c, x));
    p->last_S = c->S;
    p->last_B = c->B;
    return 0;
  }
  if (tf) BNN_TRY(launch_tf_fwd(p, a, c, x));
  else if (trunk) BNN_TRY(launch_trunk_fwd(p, a, c, x));
  for (int gi = trunk ? 3 : 0; gi < p->n_groups; ++gi) {
    GroupArgs A;
    fill_group_args(p, a, c, gi, x, &A);
    if (!bf && densef_ok(p, c, A, false)) {
      const bool fl = last_fused_ok(p, c->em, gi, true);   // the fp32 fin kernel evaluates the last layer under LRT too
      BNN_TRY(launch_densef_fwd(p, A, c->em, c->st, &p->prof, gi, fl, drop_on(c) ? c->drop : nullptr, c->train));
      p->fwd_fused_last = fl;
      if (fl) ++gi;   // the last layer was evaluated by the fin kernel
    }
    else if (!bf)
      BNN_TRY(launch_fwd<PrecF32>(A, c->em, c->st, &p->prof, gi));
    else if (!A.g.is_dense)
      return fail(BNN_E_INVALID, "internal: a conv group of the split-bf16 plan outside the fused trunk kernels");
    else if (dense_ks_ok(p, A, c->em)) {
      const bool fl = last_fused_ok(p, c->em, gi);
      BNN_TRY(launch_dense_ks_fwd(p, A, c->em, c->st, &p->prof, gi, fl));
      p->fwd_fused_last = fl;
      if (fl) ++gi;   // the last layer was evaluated by the fin kernel
    }
    else if (dense_dma_ok(A))
      BNN_TRY(launch_dense_fwd2(A, c->em, c->st, &p->prof, gi));
    else
      BNN_TRY(launch_fwd<PrecBF>(A, c->em, c->st, &p->prof, gi));
  }
  p->last_S = c->S;
  p->last_B = c->B;
  return 0;
}

// fused fp32 trunk dX (kernels_f32.h): dz of MID and of ACT1 from dY(ACT2), groups 2 and 1 in one launch
static int tf_nsplit(const Ctx* c) { return std::max(1, std::min(c->B, 256 / std::max(1, c->S))); }

static int launch_tf_dx(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, bool premasked) {
  GroupArgs G;
  fill_group_args(p, a, c, 1, a->x, &G);
  BNN_TRY(tf_check_tables(p));
  TfDxArgs T{};
  T.g_act2 = tens_ptr(p, TI_ACT2, 1);
  T.g_act2m = tens_ptr(p, TI_ACT2, 1);
  T.act2 = tens_ptr(p, TI_ACT2, 0);
  T.amax = (const unsigned char*)p->bufs.workspace + p->o_amax;
  T.m_act1 = (const unsigned char*)p->bufs.workspace + p->o_mact1;
  T.m_mid = (const unsigned char*)p->bufs.workspace + p->o_mmid;
  T.g_mid = tens_ptr(p, TI_MID, 1);
  T.g_act1 = tens_ptr(p, TI_ACT1, 1);
  T.ws = G.ws;
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  T.S = c->S;
  T.B = c->B;
  T.L = p->d.win_length;
  T.nsplit = tf_nsplit(c);
  T.drop_scale = drop_on(c) ? (float)(1.0 / (1.0 - c->drop->p / 4)) : 1.f;
  T.q2 = tens_ptr(p, TI_ACT2, 2);
  T.qm = tens_ptr(p, TI_MID, 2);
  T.act1 = tens_ptr(p, TI_ACT1, 0);
  T.mid = tens_ptr(p, TI_MID, 0);
  static_assert(TD_LDS <= 160 * 1024, "LDS budget");
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_DX, 1, c->st);
  if (c->em == EM_LRT) {
    ps_.name("tf_dx_lrt_kernel");
    if (!premasked) return fail(BNN_E_INVALID, "fp32 LRT trunk dX expects dY(ACT2) premasked by the dense dX kernel");
    BNN_DRY_RETURN();
    BNN_TRY(set_lds(tf_dx_lrt_kernel, TL_LDS));
    tf_dx_lrt_kernel<<<dim3(grid), dim3(TF_THREADS), TL_LDS, c->st>>>(T);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  ps_.name("tf_dx_kernel<%d, %s>", c->em, premasked ? "true" : "false");
  BNN_DRY_RETURN();
#define LAUNCH_TFDX(EMV, PREV)                                                             \
  do {                                                                                     \
    BNN_TRY(set_lds((tf_dx_kernel<EMV, PREV>), TD_LDS));                                   \
    tf_dx_kernel<EMV, PREV><<<dim3(grid), dim3(TF_THREADS), TD_LDS, c->st>>>(T);           \
  } while (0)
  if (c->em == EM_FLIPOUT) {
    if (premasked) LAUNCH_TFDX(EM_FLIPOUT, true); else LAUNCH_TFDX(EM_FLIPOUT, false);
  } else {
    if (premasked) LAUNCH_TFDX(EM_PLAIN, true); else LAUNCH_TFDX(EM_PLAIN, false);
  }
#undef LAUNCH_TFDX
  HIP_TRY(hipGetLastError());
  return 0;
}

// fp32 trunk dW (kernels_f32.h): kind 0 = block 1 + the k3 / k5 level, kind 1 = the 1x1 level; one slab set
static int launch_tf_dw(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, int kind) {
  GroupArgs G;
  fill_group_args(p, a, c, 0, a->x, &G);
  TfDwArgs T{};
  const float* xp = (const float*)((char*)p->bufs.workspace + p->o_xplanes);
  T.xp[0] = xp;
  T.xp[1] = xp + (size_t)p->d.max_batch * p->d.win_length * TF_XC;
  T.act1 = tens_ptr(p, TI_ACT1, 0);
  T.mid = tens_ptr(p, TI_MID, 0);
  T.g_act1 = tens_ptr(p, TI_ACT1, 1);
  T.g_mid = tens_ptr(p, TI_MID, 1);
  T.g_act2 = tens_ptr(p, TI_ACT2, 1);
  T.layers = G.layers;
  T.sign_in = c->nz.sign_in;
  T.sign_out = c->nz.sign_out;
  T.examples = (long)c->S * c->B;
  T.gw_a = ws_f(p, p->o_slab_a[0]); T.gw_b = ws_f(p, p->o_slab_b[0]); T.gb_a = ws_f(p, p->o_slab_ba[0]);
  T.gb_b = ws_f(p, p->o_slab_bb);
  T.q1 = tens_ptr(p, TI_ACT1, 2); T.qm = tens_ptr(p, TI_MID, 2); T.q2 = tens_ptr(p, TI_ACT2, 2);
  T.gw_stride = p->slab_stride; T.gb_stride = p->slab_bstride;
  T.S = c->S; T.B = c->B; T.L = p->d.win_length;
  T.nsplit = tf_nsplit(c);
  BNN_TRY(check_slab_slots(p, c, 0, T.nsplit));
  static_assert(tw_lds<0>() <= 160 * 1024 && tw_lds<1>() <= 160 * 1024 && tw_lds<0, true>() <= 160 * 1024 && tw_lds<1, true>() <= 160 * 1024, "LDS budgets");
  {
    // LDS-DMA instructions per window (64 slots of 16 B each, pad slots included) against the 8 x TFW_NDMA the waves issue
    auto ni = [&](int spr) { return (T.L * spr + 63) / 64; };
    const int n0 = 3 * ni(TFW_RX / 16) + 2 * ni(TFW_RB / 16), n1 = 2 * ni(TFW_RB / 16) + ni(TFW_RZB / 16);
    if (std::max(n0, n1) > TFW_NDMA * TF_WAVES) return fail(BNN_E_INVALID, "fp32 trunk dW: window too long for the staging plan");
    const int l0 = 2 * ni(TFWL_RX / 16) + 3 * ni(TFW_RB / 16) + 2 * ni(TFW_RZA / 16), l1 = 3 * ni(TFW_RB / 16) + 2 * ni(TFW_RZB / 16);
    if (c->em == EM_LRT && std::max(l0, l1) > TFWL_NDMA * TF_WAVES) return fail(BNN_E_INVALID, "fp32 LRT trunk dW: window too long for the staging plan");
  }
  const unsigned grid = (unsigned)(c->S * T.nsplit);
  ProfScope ps_(&p->prof, PK_DW, kind == 0 ? 0 : 1, c->st);
  ps_.name("tf_dw_kernel<%d, %d>", c->em, kind);
  BNN_DRY_RETURN();
#define LAUNCH_TFDW(EMV, KV)                                                                        \
  do {                                                                                              \
    BNN_TRY(set_lds((tf_dw_kernel<EMV, KV>), tw_lds<KV>()));                                        \
    tf_dw_kernel<EMV, KV><<<dim3(grid), dim3(TF_THREADS), tw_lds<KV>(), c->st>>>(T);                \
  } while (0)
  if (c->em == EM_LRT) {
#define LAUNCH_TFDWL(KV)                                                                               \
  do {                                                                                               \
    BNN_TRY(set_lds((tf_dw_kernel<EM_LRT, KV>), (tw_lds<KV, true>())));                              \
    tf_dw_kernel<EM_LRT, KV><<<dim3(grid), dim3(TF_THREADS), (tw_lds<KV, true>()), c->st>>>(T);      \
  } while (0)
    if (kind == 0) LAUNCH_TFDWL(0); else LAUNCH_TFDWL(1);
#undef LAUNCH_TFDWL
  } else if (c->em == EM_FLIPOUT) {
    if (kind == 0) LAUNCH_TFDW(EM_FLIPOUT, 0); else LAUNCH_TFDW(EM_FLIPOUT, 1);
  } else {
    if (kind == 0) LAUNCH_TFDW(EM_PLAIN, 0); else LAUNCH_TFDW(EM_PLAIN, 1);
  }
#undef LAUNCH_TFDW
  HIP_TRY(hipGetLastError());
  return 0;
}

// the wide dense layer of the Inception net on the fp32 plan: K-split, weight-stationary (kernels_f32.h)
static bool densef_ok(const BnnPlan* p, const Ctx* c, const GroupArgs& A, bool bwd) {
  if (!tf_dense_ok(p, c, bwd) || p->o_dks == 0 || !A.g.is_dense || A.g.n_branch != 1 || A.g.in_bcast) return false;
  const BranchDesc& br = A.g.br[0];
  const LayerDesc& ly = p->layers[br.layer];
  if (c->em == EM_LRT && p->o_dksv == 0) return false;   // (a plan created for another estimator, LRT by override)
  return br.cout == 64 && br.n_off == 0 && br.in_off == 0 && br.out_off == 0 && br.cin_p == ly.cin && ly.cin % FDF_CH == 0 &&
         ly.KP == ly.cin && ly.KPt == 64 && A.t[A.g.in_t].ctot == ly.cin && A.t[br.out_t].ctot == 64 && br.relu &&
         (long)A.cg.S * A.cg.B <= p->dks_rows && ly.sign_out_words == 2 && (ly.w_off & 3) == 0 && (ly.wt_off & 3) == 0;
}

static void densef_geometry(const GroupArgs& A, int nchunk, int max_rs, int* nrs, int* rows_per_wg) {
  const int steps = (A.cg.B + FDF_ROWS - 1) / FDF_ROWS;
  int r = std::max(1, std::min(steps, 256 / std::max(1, A.cg.S * nchunk)));
  r = std::min(r, max_rs);
  *nrs = r;
  *rows_per_wg = ((steps + r - 1) / r) * FDF_ROWS;
}

// forward / dX of the wide dense layer: how the (particle, chunk, 32-row step) items are dealt to at most 256 workgroups
// (df_segment in kernels_f32.h).  *q = 0: equal contiguous item ranges (most workgroups load weight fragments twice); *q > 0:
// full ranges of q steps inside a pair + packed remainders.  Cost model: a row step ~5 us, a fragment load ~7 us (measured).
static unsigned densef_schedule(const GroupArgs& A, int nchunk, int* q) {
  const int SP = (A.cg.B + FDF_ROWS - 1) / FDF_ROWS, pairs = A.cg.S * nchunk;
  const long items = (long)pairs * SP;
  const unsigned gbal = (unsigned)std::max(1L, std::min(items, 256L));
  const double c = 5.0, P = 7.0;
  const long per = (items + gbal - 1) / gbal;
  const double cost_bal = per * c + (per >= SP ? (double)((per + SP - 1) / SP + 1) : 2.0) * P;
  *q = 0;
  unsigned best = gbal;
  double best_cost = cost_bal;
  for (int qq = (int)std::max(1L, (items + 255) / 256); qq <= SP; ++qq) {
    const int k = SP / qq, r = SP - k * qq, m = r ? std::max(1, qq / r) : 0;
    const long G = (long)pairs * k + (r ? (pairs + m - 1) / m : 0);
    if (G > 256) continue;
    const double cost = std::max(k ? qq * c + P : 0.0, r ? m * (r * c + P) : 0.0);
    if (cost < best_cost) {
      best_cost = cost;
      best = (unsigned)G;
      *q = qq;
    }
    if (k == 1 && r == 0) break;
  }
  return best;
}

// host-side check of a schedule (dry-run validation): every (pair, row step) item dealt exactly once
static int densef_check_schedule(const GroupArgs& A, int nchunk, unsigned grid, int q) {
  const int SP = (A.cg.B + FDF_ROWS - 1) / FDF_ROWS, pairs = A.cg.S * nchunk;
  std::vector<unsigned char> seen((size_t)pairs * SP, 0);
  for (unsigned wg = 0; wg < grid; ++wg) {
    long item = 0;
    DfSeg sg;
    for (int seg = 0; df_segment((int)wg, (int)grid, seg, pairs, SP, q, item, sg); ++seg) {
      if (sg.pair < 0 || sg.pair >= pairs || sg.t0 < 0 || sg.n < 1 || sg.t0 + sg.n > SP)
        return fail(BNN_E_INVALID, "internal: dense item schedule out of range (S %d B %d q %d)", A.cg.S, A.cg.B, q);
      for (int t = sg.t0; t < sg.t0 + sg.n; ++t)
        if (seen[(size_t)sg.pair * SP + t]++) return fail(BNN_E_INVALID, "internal: dense item schedule deals an item twice (S %d B %d q %d)", A.cg.S, A.cg.B, q);
      if (seg > 4096) return fail(BNN_E_INVALID, "internal: dense item schedule does not terminate");
    }
  }
  for (unsigned char v : seen)
    if (!v) return fail(BNN_E_INVALID, "internal: dense item schedule leaves an item out (S %d B %d q %d)", A.cg.S, A.cg.B, q);
  return 0;
}

static int launch_densef_fwd(BnnPlan* p, const GroupArgs& A, int em, hipStream_t st, Prof* pf, int gi, bool fuse_last, const BnnDropout* drop, bool train) {
  if (em == EM_LRT && !fuse_last) return fail(BNN_E_INVALID, "fp32 LRT dense forward needs the fused hidden / last layer launch");
  if (drop && !fuse_last) return fail(BNN_E_INVALID, "MC-dropout needs the fused hidden / last layer launch");
  const BranchDesc& br = A.g.br[0];
  const LayerDesc& ly = p->layers[br.layer];
  DfArgs F{};
  F.x = (const float*)A.t[A.g.in_t].p;
  F.x_ctot = A.t[A.g.in_t].ctot;
  F.wa = (const float*)A.ws.a_hi + ly.w_off;
  F.wb = (const float*)A.ws.b + ly.w_off;
  F.stride_a = A.ws.slot_stride_a;
  F.stride_b = A.ws.slot_stride_b;
  F.KP = ly.KP;
  F.sg_in = A.nz.sign_in + ly.sign_in_off * A.nz.examples;
  F.sg_out = A.nz.sign_out + ly.sign_out_off * A.nz.examples;
  F.siw = ly.sign_in_words;
  F.sow = ly.sign_out_words;
  F.slab = ws_f(p, p->o_dks);
  F.slabv = ws_f(p, p->o_dksv);
  F.slab_stride = (long)A.cg.S * A.cg.B * 64;
  F.S = A.cg.S;
  F.B = A.cg.B;
  F.nchunk = ly.cin / FDF_CH;
  static_assert(FDF_LDS <= 160 * 1024 && FDX_LDS <= 160 * 1024 && DWF_LDS <= 160 * 1024, "LDS budgets");
  const unsigned grid = densef_schedule(A, F.nchunk, &F.q);
  if (g_dry) BNN_TRY(densef_check_schedule(A, F.nchunk, grid, F.q));
  ProfScope ps_(pf, PK_FWD, gi, st);
  ps_.name("densef_fwd_kernel<%d>", em);
  BNN_DRY_RETURN();
  if (em == EM_PLAIN) {
    BNN_TRY(set_lds(densef_fwd_kernel<EM_PLAIN>, FDF_LDS));
    densef_fwd_kernel<EM_PLAIN><<<dim3(grid), dim3(TF_THREADS), FDF_LDS, st>>>(F);
  } else if (em == EM_LRT) {
    BNN_TRY(set_lds(densef_fwd_kernel<EM_LRT>, FDF_LDS));
    densef_fwd_kernel<EM_LRT><<<dim3(grid), dim3(TF_THREADS), FDF_LDS, st>>>(F);
  } else {
    BNN_TRY(set_lds(densef_fwd_kernel<EM_FLIPOUT>, FDF_LDS));
    densef_fwd_kernel<EM_FLIPOUT><<<dim3(grid), dim3(TF_THREADS), FDF_LDS, st>>>(F);
  }
  if (fuse_last) {
    // + the last layer Linear(64, 2) on the row while it is in registers
    const LayerDesc& l2 = p->layers[p->groups[gi + 1].br[0].layer];
    DenseFinF32Args R{};
    R.slab = F.slab;
    R.slab_stride = F.slab_stride;
    R.nchunk = F.nchunk;
    R.rows = A.cg.S * A.cg.B;
    R.B = A.cg.B;
    R.bias = A.ws.bias_a + ly.bias_off + br.n_off;
    R.b2 = A.ws.bias_a + l2.bias_off;
    R.bias_stride = A.ws.bias_stride_a;
    R.h = (float*)A.t[br.out_t].p;
    R.w2a = (const float*)A.ws.a_hi + l2.w_off;
    R.w2b = (const float*)A.ws.b + l2.w_off;
    R.w2_stride_a = A.ws.slot_stride_a;
    R.w2_stride_b = A.ws.slot_stride_b;
    R.w2_KP = l2.KP;
    R.sg_in = A.nz.sign_in + l2.sign_in_off * A.nz.examples;
    R.sg_out = A.nz.sign_out + l2.sign_out_off * A.nz.examples;
    R.siw = l2.sign_in_words;
    R.sow = l2.sign_out_words;
    R.z = tens_ptr(p, p->z_t, 0);
    R.g2_a = A.gw_a + l2.w_off;   // zeroed for the head launch's atomics (every other gradient element of a step is stored)
    R.g2_b = A.gw_b + l2.w_off;
    R.g2_ba = A.gb_a + l2.bias_off;
    R.g2_stride = A.gw_stride;
    R.g2_bstride = A.gb_stride;
    R.S = A.cg.S;
    if (em == EM_LRT) {
      R.slabv = F.slabv;
      R.biasv = A.ws.bias_b + ly.bias_off + br.n_off;
      R.b2v = A.ws.bias_b + l2.bias_off;
      R.nz = A.nz;
      R.cg = A.cg;
      R.layer1 = br.layer;
      R.layer2 = p->groups[gi + 1].br[0].layer;
      R.qh = train ? (float*)A.t[br.q_t].p : nullptr;
      R.qz = train ? (float*)A.t[p->groups[gi + 1].br[0].q_t].p : nullptr;
      R.g2_bb = A.gb_b + l2.bias_off;
    }
    if (drop) {
      R.drop_rate = (float)drop->p;
      R.drop_scale = (float)(1.0 / (1.0 - drop->p));
      R.drop_seed = drop->seed;
      R.drop_step = (uint32_t)drop->step;
      R.keep_h = drop->keep_h;
    }
    // one thread per (row, 4 channels); at least S * 2 * KP threads for the zeroing (surplus threads redo the last row)
    const unsigned fgrid = (unsigned)((std::max((long)R.rows * 16, (long)R.S * 2 * l2.KP) + 255) / 256);
    if (em == EM_PLAIN) densef_fin_kernel<EM_PLAIN><<<dim3(fgrid), dim3(256), 0, st>>>(R);
    else if (em == EM_LRT) densef_fin_kernel<EM_LRT><<<dim3(fgrid), dim3(256), 0, st>>>(R);
    else densef_fin_kernel<EM_FLIPOUT><<<dim3(fgrid), dim3(256), 0, st>>>(R);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  DenseKsFinArgs R{};
  R.slab = F.slab;
  R.slab_stride = F.slab_stride;
  R.nchunk = F.nchunk;
  R.rows = A.cg.S * A.cg.B;
  R.B = A.cg.B;
  R.bias = A.ws.bias_a + ly.bias_off + br.n_off;
  R.bias_stride = A.ws.bias_stride_a;
  R.relu = br.relu;
  R.out = A.t[br.out_t];
  R.out_off = br.out_off;
  const unsigned fgrid = (unsigned)((R.rows * 16 + 255) / 256);
  dense_ks_fin_kernel<EM_PLAIN><<<dim3(fgrid), dim3(256), 0, st>>>(R);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_densef_bwd(BnnPlan* p, const GroupArgs& A, int em, hipStream_t st, Prof* pf, int gi, const BnnDropout* drop) {
  const BranchDesc& br = A.g.br[0];
  const LayerDesc& ly = p->layers[br.layer];
  DfBwdArgs F{};
  F.x = (const float*)A.t[A.g.in_t].p;
  F.x_ctot = A.t[A.g.in_t].ctot;
  F.h = (const float*)A.t[br.out_t].p;
  F.dh = (const float*)A.t[br.out_t + T_GRAD].p;
  F.wat = (const float*)A.ws.at + ly.wt_off;
  F.wbt = (const float*)A.ws.bt + ly.wt_off;
  F.stride_at = A.ws.slott_stride_a;
  F.stride_bt = A.ws.slott_stride_b;
  F.KPt = ly.KPt;
  F.sg_in = A.nz.sign_in + ly.sign_in_off * A.nz.examples;
  F.sg_out = A.nz.sign_out + ly.sign_out_off * A.nz.examples;
  F.siw = ly.sign_in_words;
  F.sow = ly.sign_out_words;
  F.dx = (float*)A.t[br.dx_t].p;
  F.m_x = (const unsigned char*)p->bufs.workspace + p->o_mact2;   // written by tf_fwd_kernel: dX(ACT2) is stored masked
  F.qh = (const float*)A.t[br.q_t].p;
  F.gb_b = A.gb_b + ly.bias_off;
  F.x_scale = drop ? (float)(1.0 / (1.0 - drop->p / 4)) : 1.f;
  F.h_scale = drop ? (float)(1.0 / (1.0 - drop->p)) : 1.f;
  F.gw_a = A.gw_a + ly.w_off;
  F.gw_b = A.gw_b + ly.w_off;
  F.gb_a = A.gb_a + ly.bias_off;
  F.gw_stride = A.gw_stride;
  F.gb_stride = A.gb_stride;
  F.KP = ly.KP;
  F.S = A.cg.S;
  F.B = A.cg.B;
  F.nchunk = ly.cin / FDF_CH;
  if (br.dx_t < 0 || A.t[br.dx_t].ctot != F.x_ctot) return fail(BNN_E_INVALID, "fp32 dense backward: gradient tensor shape");
  {
    // dW: a (particle, chunk) pair's row steps are cut into full ranges of q steps (one workgroup each) + a remainder (packed
    // m pairs to a workgroup): up to 8 ranges per pair, every range but the first into a partial image of its own; (ranges - 1)
    // * S <= 25 partial images exist in the workspace.  The cheapest admissible q by a small cost model:
    unsigned grid = 0;
    {
      const int SP = (F.B + FDF_ROWS - 1) / FDF_ROWS, pairs = F.S * F.nchunk;
      double best = 1e30;
      for (int qq = 1; qq <= SP; ++qq) {
        const int k = SP / qq, r = SP - k * qq, m = r ? std::max(1, qq / r) : 0, ranges = k + (r ? 1 : 0);
        const long G = (long)pairs * k + (r ? (pairs + m - 1) / m : 0);
        if ((G > 256 && ranges > 1) || ranges > 8 || (long)(ranges - 1) * F.S > DENSEF_DW_PARTS) continue;   // (one range per pair always exists)
        // measured: a row step 5.5 us with two contractions (Flipout, LRT), 2.6 us with one; 8 us per range for the zero fill,
        // the first loads and the flush; 8 us for the add launch + 1.5 us per partial image it reads
        const double cs = em == EM_PLAIN ? 2.6 : 5.5;
        const double cost = (std::max(k ? qq * cs + 8.0 : 0.0, r ? m * (r * cs + 8.0) : 0.0) + (ranges > 1 ? 8.0 + (ranges - 1) * 1.5 : 0.0)) *
                            (double)((G + 255) / 256);
        if (cost < best) {
          best = cost;
          grid = (unsigned)G;
          F.q_dw = qq;
          F.nrs = ranges;
        }
      }
      if (!grid) return fail(BNN_E_INVALID, "internal: no dense dW schedule for S %d B %d", F.S, F.B);
      if (g_dry) BNN_TRY(densef_check_schedule(A, F.nchunk, grid, F.q_dw));
    }
    ProfScope ps_(pf, PK_DW, gi, st);
    ps_.name("densef_dw_kernel<%d>", em);
    // the second row range's partial images: laid out like the gradient images (same layer offset and particle strides)
    F.gw2_a = ws_f(p, p->o_dw2_a) + ly.w_off;
    F.gw2_b = ws_f(p, p->o_dw2_b) + ly.w_off;
    F.gb2_a = ws_f(p, p->o_dw2_ba) + ly.bias_off;
    F.gb2_b = ws_f(p, p->o_dw2_bb) + ly.bias_off;
    if (!g_dry) {
      if (em == EM_PLAIN) {
        BNN_TRY(set_lds(densef_dw_kernel<EM_PLAIN>, DWF_LDS));
        densef_dw_kernel<EM_PLAIN><<<dim3(grid), dim3(TF_THREADS), DWF_LDS, st>>>(F);
      } else if (em == EM_LRT) {
        BNN_TRY(set_lds(densef_dw_kernel<EM_LRT>, DWF_LDS));
        densef_dw_kernel<EM_LRT><<<dim3(grid), dim3(TF_THREADS), DWF_LDS, st>>>(F);
      } else {
        BNN_TRY(set_lds(densef_dw_kernel<EM_FLIPOUT>, DWF_LDS));
        densef_dw_kernel<EM_FLIPOUT><<<dim3(grid), dim3(TF_THREADS), DWF_LDS, st>>>(F);
      }
      if (F.nrs > 1) {
        const long n = (long)ly.cout * ly.KP;
        DenseAddJobs J{};
        int nj = 0;
        auto job = [&](float* dst, const float* src, long cnt, long stride) {
          J.dst[nj] = dst; J.src[nj] = src; J.n[nj] = cnt; J.stride[nj] = stride;
          ++nj;
        };
        job(F.gw_a, F.gw2_a, n, F.gw_stride);
        if (em != EM_PLAIN) job(F.gw_b, F.gw2_b, n, F.gw_stride);
        job(F.gb_a, F.gb2_a, 64, F.gb_stride);
        if (em == EM_LRT) job(F.gb_b, F.gb2_b, 64, F.gb_stride);
        J.S = F.S;
        J.np = F.nrs - 1;
        dense_addn_kernel<<<dim3((unsigned)((n / 4 + 255) / 256), (unsigned)F.S, (unsigned)nj), dim3(256), 0, st>>>(J);
      }
    }
  }
  {
    const unsigned grid = densef_schedule(A, F.nchunk, &F.q);
    if (g_dry) BNN_TRY(densef_check_schedule(A, F.nchunk, grid, F.q));
    ProfScope ps_(pf, PK_DX, gi, st);
    ps_.name("densef_dx_kernel<%d>", em);
    if (!g_dry) {
      if (em == EM_PLAIN) {
        BNN_TRY(set_lds(densef_dx_kernel<EM_PLAIN>, FDX_LDS));
        densef_dx_kernel<EM_PLAIN><<<dim3(grid), dim3(TF_THREADS), FDX_LDS, st>>>(F);
      } else if (em == EM_LRT) {
        BNN_TRY(set_lds(densef_dx_kernel<EM_LRT>, FDX_LDS));
        densef_dx_kernel<EM_LRT><<<dim3(grid), dim3(TF_THREADS), FDX_LDS, st>>>(F);
      } else {
        BNN_TRY(set_lds(densef_dx_kernel<EM_FLIPOUT>, FDX_LDS));
        densef_dx_kernel<EM_FLIPOUT><<<dim3(grid), dim3(TF_THREADS), FDX_LDS, st>>>(F);
      }
    }
  }
  if (!g_dry) HIP_TRY(hipGetLastError());
  return 0;
}

static int do_head(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, float* preds, bool want_dz) {
  HeadArgs H{};
  H.z = tens_ptr(p, p->z_t, 0);
  H.y = a->y;
  H.dz = want_dz ? tens_ptr(p, p->z_t, 1) : nullptr;
  H.preds = preds;
  H.ll_acc = (double*)((char*)p->bufs.workspace + p->o_acc) + (p->d.max_particles + 1);
  H.S = c->S;
  H.B = c->B;
  H.with_obs = a->with_obs && a->y;
  H.objective = c->objective;
  ProfScope ps_(&p->prof, PK_HEAD, 0, c->st);
  if (c->last_fused && want_dz && p->d.prec == BNN_PREC_F32) {
    ps_.name("headf_last_kernel<%d>", c->em);
    GroupArgs A;
    fill_group_args(p, a, c, p->n_groups - 1, a->x, &A);
    const BranchDesc& br = A.g.br[0];
    const LayerDesc& ly = p->layers[br.layer];
    HeadLastF32Args L{};
    L.H = H;
    L.H.dz = nullptr;
    L.wa = (const float*)A.ws.a_hi + ly.w_off;
    L.wb = (const float*)A.ws.b + ly.w_off;
    L.stride_a = A.ws.slot_stride_a;
    L.stride_b = A.ws.slot_stride_b;
    L.KP = ly.KP;
    L.h = (const float*)A.t[A.g.in_t].p;
    L.dh = (float*)A.t[br.dx_t].p;
    L.sg_in = A.nz.sign_in + ly.sign_in_off * A.nz.examples;
    L.sg_out = A.nz.sign_out + ly.sign_out_off * A.nz.examples;
    L.siw = ly.sign_in_words;
    L.sow = ly.sign_out_words;
    L.gw_a = A.gw_a + ly.w_off;
    L.gw_b = A.gw_b + ly.w_off;
    L.gb_a = A.gb_a + ly.bias_off;
    L.gw_stride = A.gw_stride;
    L.gb_stride = A.gb_stride;
    L.qz = (const float*)A.t[br.q_t].p;
    L.gb_b = A.gb_b + ly.bias_off;
    if (A.t[A.g.in_t].ctot != 64 || A.t[br.dx_t].ctot != 64) return fail(BNN_E_INVALID, "fp32 head: hidden width");
    if (g_dry) return 0;
    const dim3 hgrid((unsigned)((c->B + HL_ROWS - 1) / HL_ROWS), (unsigned)c->S);
    if (c->em == EM_PLAIN) headf_last_kernel<EM_PLAIN><<<hgrid, dim3(256), 0, c->st>>>(L);
    else if (c->em == EM_LRT) headf_last_kernel<EM_LRT><<<hgrid, dim3(256), 0, c->st>>>(L);
    else headf_last_kernel<EM_FLIPOUT><<<hgrid, dim3(256), 0, c->st>>>(L);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (c->last_fused && want_dz) {
    // head + backward of the last layer in one launch (its gradient images were zeroed by do_backward's fill, which
    // therefore runs BEFORE the head on this path: see bnn_elbo_step)
    ps_.name("head_last_kernel<%d>", c->em);
    GroupArgs A;
    fill_group_args(p, a, c, p->n_groups - 1, a->x, &A);
    const BranchDesc& br = A.g.br[0];
    const LayerDesc& ly = p->layers[br.layer];
    HeadLastArgs L{};
    L.H = H;
    L.H.dz = nullptr;
    L.w_hi = (const u16*)A.ws.a_hi + ly.w_off;
    L.w_b = (const u16*)A.ws.b + ly.w_off;
    L.stride_a = A.ws.slot_stride_a;
    L.stride_b = A.ws.slot_stride_b;
    L.KP = ly.KP;
    L.h_hi = (const u16*)A.t[A.g.in_t].p;
    L.h_ctot = A.t[A.g.in_t].ctot;
    L.dh = (u16*)A.t[br.dx_t].p;
    L.dh_ctot = A.t[br.dx_t].ctot;
    L.sg_in = A.nz.sign_in + ly.sign_in_off * A.nz.examples;
    L.sg_out = A.nz.sign_out + ly.sign_out_off * A.nz.examples;
    L.siw = ly.sign_in_words;
    L.sow = ly.sign_out_words;
    L.gw_a = A.gw_a + ly.w_off;
    L.gw_b = A.gw_b + ly.w_off;
    L.gb_a = A.gb_a + ly.bias_off;
    L.gw_stride = A.gw_stride;
    L.gb_stride = A.gb_stride;
    if (g_dry) return 0;
    const dim3 hgrid((unsigned)((c->B + HL_ROWS - 1) / HL_ROWS), (unsigned)c->S);
    if (c->em == EM_PLAIN) head_last_kernel<EM_PLAIN><<<hgrid, dim3(256), 0, c->st>>>(L);
    else head_last_kernel<EM_FLIPOUT><<<hgrid, dim3(256), 0, c->st>>>(L);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  head_nll_kernel<<<dim3((c->B + 255) / 256, c->S), dim3(256), 0, c->st>>>(H);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int reduce_trunk_slabs(BnnPlan* p, const Ctx* c);

static int zero_grad_images(BnnPlan* p, const Ctx* c) {
  const size_t gwb = (size_t)c->S * p->img_total * 4, gbb = (size_t)c->S * p->bias_total * 4;
  if (g_dry) {
  } else if (c->S == p->d.max_particles) {
    // gw_a | gw_b | gb_a | gb_b are back to back in the workspace: one fill
    HIP_TRY(hipMemsetAsync(ws_f(p, p->o_gw_a), 0, (p->o_gb_b - p->o_gw_a) + gbb, c->st));
  } else {
    HIP_TRY(hipMemsetAsync(ws_f(p, p->o_gw_a), 0, gwb, c->st));
    HIP_TRY(hipMemsetAsync(ws_f(p, p->o_gw_b), 0, gwb, c->st));
    HIP_TRY(hipMemsetAsync(ws_f(p, p->o_gb_a), 0, gbb, c->st));
    HIP_TRY(hipMemsetAsync(ws_f(p, p->o_gb_b), 0, gbb, c->st));
  }
  return 0;
}

// Training-step tail on the Inception trunk path: the last layer's backward rides on the head launch (its atomics need
// zeroed images BEFORE the head), and when every other element of the gradient images is stored, not added (slabs for the
// conv layers, one workgroup per (particle, chunk) for the wide dense layer), the fill is not needed at all: the fin
// kernel of the forward zeroed the last layer's few elements.
static int prepare_fused_tail(BnnPlan* p, const BnnElboArgs* a, Ctx* c) {
  const bool tfd = tf_dense_ok(p, c);   // fp32 plan: the dense layer's fused backward reads dH from the fused head launch
  c->last_fused = (trunk_ok(p, c) || tfd) && p->fwd_fused_last && last_fused_ok(p, c->em, p->n_groups - 2, tfd);
  if (!c->last_fused) return 0;
  if (tfd && !tf_bwd_ok(p, c)) {
    // (the conv groups still run the generic kernels for this estimator: their dW adds with atomics)
    BNN_TRY(zero_grad_images(p, c));
    c->grads_zeroed = true;
    return 0;
  }
  if (tfd) {
    // fp32 plan: every gradient element of the conv layers (slab reduction) and of the wide dense layer (densef_dw_kernel,
    // dense_addn_kernel) is STORED; the last layer's few elements, which the head launch adds to, were zeroed by the fin
    // kernel of the forward: no fill
    c->grads_zeroed = true;
    return 0;
  }
  const BranchDesc& br = p->groups[p->n_groups - 2].br[0];
  const bool direct = dense_ks_bwd_nsplit(c->S, br.cin_p, (c->B + TILE_ROWS - 1) / TILE_ROWS) == 1;
  if (!direct) BNN_TRY(zero_grad_images(p, c));
  c->grads_zeroed = true;
  c->direct_assumed = direct;   // do_backward refuses to run if the dense group does not take the K-split backward after all
  return 0;
}

static int do_backward(BnnPlan* p, const BnnElboArgs* a, const Ctx* c) {
  if (mlp_ok(p, c)) return launch_mlp_bwd(p, a, c);   // zeroes the gradient images itself
  if (!c->grads_zeroed) BNN_TRY(zero_grad_images(p, c));
  bool act2_premasked = false;
  for (int gi = p->n_groups - 1 - (c->last_fused ? 1 : 0); gi >= 0; --gi) {
    GroupArgs A;
    fill_group_args(p, a, c, gi, a->x, &A);
    if (!A.g.is_dense && tf_ok(p, c)) {
      // fp32 conv trunk: dz of MID / ACT1 (and the masked dz of ACT2), then the two dW launches and the slab reduction
      if (gi == 2) {
        BNN_TRY(launch_tf_dx(p, a, c, act2_premasked));
        if (tf_dw_ok(p, c)) {
          BNN_TRY(launch_tf_dw(p, a, c, 0));
          BNN_TRY(launch_tf_dw(p, a, c, 1));
          BNN_TRY(reduce_trunk_slabs(p, c));
        }
      }
      // (an estimator without fused dW kernels yet: the generic kernel of this group, on the masked gradients above)
      if (!tf_dw_ok(p, c)) BNN_TRY((launch_dw<PrecF32, 2>(A, c->em, c->st, &p->prof, gi)));
      continue;
    }
    if (!A.g.is_dense && trunk_ok(p, c)) {
      // conv trunk: dz of MID / ACT1 (and the masked dz of ACT2) first, then the three dW kernels
      if (gi == 2) {
        BNN_TRY(launch_trunk_dx(p, a, c, act2_premasked));
        BNN_TRY(launch_trunk_dw2(p, a, c, 1));
      } else if (gi == 1) {
        BNN_TRY(launch_trunk_dw2(p, a, c, 0));
      } else {
        BNN_TRY(launch_trunk_dw1(p, a, c));
        BNN_TRY(reduce_trunk_slabs(p, c));
      }
      continue;
    }
    if (A.g.is_dense && dense_ks_ok(p, A, c->em) && A.g.br[0].dx_t >= 0 && A.t[A.g.br[0].out_t].fmt == TF_BF16) {
      BNN_TRY(launch_dense_ks_bwd(p, A, c->em, c->st, &p->prof, gi));
      act2_premasked = true;   // its dX is stored masked with [input > 0]
      continue;
    }
    if (A.g.is_dense && c->direct_assumed && gi == p->n_groups - 2)
      return fail(BNN_E_INVALID, "internal: the gradient images were left unfilled for the K-split dense backward, which this call does not take");
    if (p->d.prec == BNN_PREC_F32 && densef_ok(p, c, A, true) && A.g.br[0].dx_t >= 0) {
      BNN_TRY(launch_densef_bwd(p, A, c->em, c->st, &p->prof, gi, drop_on(c) ? c->drop : nullptr));
      act2_premasked = true;   // its dX is stored masked with [input > 0]
      continue;
    }
    if (p->d.prec == BNN_PREC_F32)
      BNN_TRY((launch_dw<PrecF32, 2>(A, c->em, c->st, &p->prof, gi)));
    else if (!A.g.is_dense)
      return fail(BNN_E_INVALID, "internal: a conv group of the split-bf16 plan outside the fused trunk kernels");
    else if (A.g.n_branch == 1 && !A.g.in_bcast && A.g.br[0].cout <= 64 && (A.g.br[0].cout % 8) == 0 &&
             (A.g.br[0].cin_p % 16) == 0 && A.g.br[0].cin_real == A.g.br[0].cin_p && A.t[A.g.in_t].fmt == TF_BF16 &&
             (A.t[A.g.in_t].ctot % 8) == 0 && (A.t[A.g.br[0].out_t].ctot % 8) == 0 && A.t[A.g.br[0].out_t].fmt == TF_BF16)
      BNN_TRY(launch_dense_dw_bf(A, c->em, c->st, &p->prof, gi));
    else
      BNN_TRY((launch_dw<PrecBF, 4>(A, c->em, c->st, &p->prof, gi)));
    bool any_direct = false, any_pool = false;
    for (int b = 0; b < A.g.n_branch; ++b) {
      if (A.g.br[b].dx_t < 0) continue;
      (A.g.br[b].pool ? any_pool : any_direct) = true;
    }
    if (any_direct) {
      if (p->d.prec == BNN_PREC_F32)
        BNN_TRY(launch_dx<PrecF32>(A, c->em, 0, c->st, &p->prof, gi));
      else if (A.g.is_dense && A.g.n_branch == 1 && A.g.br[0].cout <= 64 && (A.g.br[0].cin_p % 16) == 0 &&
               A.g.br[0].cin_real == A.g.br[0].cin_p && (A.t[A.g.br[0].dx_t].ctot % 4) == 0)
        BNN_TRY(launch_dense_dx_bf(A, c->em, c->st, &p->prof, gi));
      else
        BNN_TRY(launch_dx<PrecBF>(A, c->em, 0, c->st, &p->prof, gi));
    }
    if (any_pool) {
      if (p->d.prec == BNN_PREC_F32)
        BNN_TRY(launch_dx<PrecF32>(A, c->em, 1, c->st, &p->prof, gi));
      else
        BNN_TRY(launch_dx<PrecBF>(A, c->em, 1, c->st, &p->prof, gi));
      // scatter through the arg-max of MaxPool1d(3,1,1) into the direct gradient
      const int tin = A.g.in_t;
      const long nwin = (long)c->S * c->B;
      const int C = p->tens[tin].ctot, L = A.g.L;
      const long n = nwin * L * C;
      if (g_dry) continue;
      ProfScope ps_(&p->prof, PK_POOLBWD, gi, c->st);
      ps_.name("pool_bwd_kernel");
      pool_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->st>>>(
          A.t[tin], A.t[T_POOLGRAD], A.t[tin + T_GRAD], nwin, L, C);
      HIP_TRY(hipGetLastError());
    }
  }
  return 0;
}

// partial images of the trunk dW kernels -> the per-particle gradient images (weights: slots A and B, biases)
static int reduce_trunk_slabs(BnnPlan* p, const Ctx* c) {
  if (g_dry) return 0;
  ProfScope ps_(&p->prof, PK_FINALIZE, 1, c->st);
  ps_.name("slab_reduce_kernel");
  static thread_local SlabReduceJobs J;
  int nj = 0;
  long max_elems = 0;
  const bool f32p = p->d.prec == BNN_PREC_F32;
  for (int which = 0; which < 4; ++which) {   // 0: slot A, 1: slot B (Flipout's dW | LRT's sigma^2 part), 2: bias sums, 3: LRT's sigma_b^2 sums
    if (which == 1 && c->em == EM_PLAIN) continue;
    if (which == 3 && !(c->em == EM_LRT && f32p)) continue;
    SlabReduceArgs& R = J.job[nj++];
    R = SlabReduceArgs{};
    for (int g = 0; g < 3; ++g) {
      R.slab[g] = ws_f(p, which == 0 ? p->o_slab_a[g] : (which == 1 ? p->o_slab_b[g] : (which == 2 ? p->o_slab_ba[g] : p->o_slab_bb)));
      R.n[g] = p->d.prec == BNN_PREC_F32 ? tf_nsplit(c) : trunk_dw_nsplit(c, g);
    }
    R.stride = which >= 2 ? p->slab_bstride : p->slab_stride;
    for (int l = 0; l < 10; ++l) R.lay_end[l] = which >= 2 ? p->layers[l + 1].bias_off : p->layers[l + 1].w_off;
    R.elems = R.lay_end[9];
    R.out = ws_f(p, which == 0 ? p->o_gw_a : (which == 1 ? p->o_gw_b : (which == 2 ? p->o_gb_a : p->o_gb_b)));
    R.out_stride = which >= 2 ? p->bias_total : p->img_total;
    R.S = c->S;
    max_elems = std::max(max_elems, R.elems);
  }
  slab_reduce_kernel<<<dim3((unsigned)((max_elems + 255) / 256), c->S, nj), dim3(256), 0, c->st>>>(J);
  HIP_TRY(hipGetLastError());
  return 0;
}

static void fill_loss_args(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const BnnElboOut* out, bool to_grad, LossArgs* L);

static int fill_adam_args(BnnPlan* p, const BnnAdamArgs* ad, AdamArgs* out);

static int do_finalize(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const BnnElboOut* fused_out = nullptr, bool fuse_loss = false,
                       const BnnAdamArgs* fused_adam = nullptr) {
  FinalizeArgs F{};
  if (fused_adam) {
    BNN_TRY(fill_adam_args(p, fused_adam, &F.adam));
    F.fused_adam = 1;
  }
  F.T = p->ptab;
  F.layers = (const LayerDesc*)((char*)p->bufs.workspace + p->o_layers);
  F.mu = p->bufs.mu;
  F.rho = p->bufs.rho;
  F.eps_w = c->eps_w;
  F.rad_r = c->rad_r;
  F.norms = ws_f(p, p->o_norms);
  F.gw_a = ws_f(p, p->o_gw_a);
  F.gw_b = ws_f(p, p->o_gw_b);
  F.gb_a = ws_f(p, p->o_gb_a);
  F.gb_b = ws_f(p, p->o_gb_b);
  F.gw_stride = p->img_total;
  F.gb_stride = p->bias_total;
  F.mode = c->mode;
  F.S = c->S;
  F.scale_ll = c->scale_ll;
  F.c = c->c;
  F.prior_loc = (float)a->prior_loc;
  F.prior_scale = (float)a->prior_scale;
  F.grad = p->bufs.grad;
  if (fuse_loss) {
    fill_loss_args(p, a, c, fused_out, true, &F.loss);
    F.fused_loss = 1;
    p->acc_clean = true;
  }
  ProfScope ps_(&p->prof, PK_FINALIZE, 0, c->st);
  grad_finalize_kernel<<<dim3((unsigned)((p->P + 255) / 256)), dim3(256), 0, c->st>>>(F);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int do_loss(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const BnnElboOut* out, bool to_grad) {
  LossArgs LA{};
  fill_loss_args(p, a, c, out, to_grad, &LA);
  finish_loss_kernel<<<dim3(1), dim3(64), 0, c->st>>>(LA);
  HIP_TRY(hipGetLastError());
  p->acc_clean = true;
  return 0;
}

static void fill_loss_args(BnnPlan* p, const BnnElboArgs* a, const Ctx* c, const BnnElboOut* out, bool to_grad, LossArgs* L) {
  LossArgs& LA = *L;
  LA = LossArgs{};
  double* acc = (double*)((char*)p->bufs.workspace + p->o_acc);
  LA.n_acc = 2 * (p->d.max_particles + 1);
  LA.kl_acc = acc;
  LA.ll_acc = acc + (p->d.max_particles + 1);
  LA.S = c->S;
  LA.radial = c->radial;
  LA.c = c->c;
  LA.n_over_b = a->with_obs ? c->n_over_b : 0.0;
  LA.kl_weight = c->objective ? 0.0 : 1.0;
  float* scal = ws_f(p, p->o_scal);
  LA.loss = (out && out->loss) ? out->loss : scal;
  LA.kl = (out && out->kl) ? out->kl : scal + 1;
  LA.ll = (out && out->loglik) ? out->loglik : scal + 2;
  LA.grad_tail = to_grad ? p->bufs.grad + 2 * p->P : nullptr;
}

static int fill_adam_args(BnnPlan* p, const BnnAdamArgs* ad, AdamArgs* out) {
  if (!p->bufs.adam_m || !p->bufs.adam_v) return fail(BNN_E_INVALID, "adam_m / adam_v not bound");
  AdamArgs& A = *out;
  A = AdamArgs{};
  A.mu = p->bufs.mu;
  A.rho = p->bufs.rho;
  A.m = p->bufs.adam_m;
  A.v = p->bufs.adam_v;
  A.grad = p->bufs.grad;
  A.P = p->P;
  A.lr = (float)ad->lr;
  A.beta1 = (float)ad->beta1;
  A.beta2 = (float)ad->beta2;
  A.eps = (float)ad->eps;
  A.clip = (float)ad->clip_norm;
  A.wd = (float)ad->weight_decay;
  const double bc1 = 1.0 - std::pow(ad->beta1, (double)ad->step);
  const double bc2 = 1.0 - std::pow(ad->beta2, (double)ad->step);
  A.step_size = (float)(ad->lr * std::sqrt(bc2) / bc1);
  A.grad_scale = (float)(ad->grad_scale == 0.0 ? 1.0 : ad->grad_scale);
  A.freeze_loc = ad->freeze_loc;
  A.freeze_scale = ad->freeze_scale;
  // torch.optim.Adam: p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)  ==  step_size * m / (sqrt(v) + eps * sqrt(bc2))
  if (ad->torch_eps) A.eps = (float)(ad->eps * std::sqrt(bc2));
  return 0;
}

static int do_adam(BnnPlan* p, const BnnAdamArgs* ad, hipStream_t st) {
  AdamArgs A{};
  BNN_TRY(fill_adam_args(p, ad, &A));
  ProfScope ps_(&p->prof, PK_ADAM, 0, st);
  clipped_adam_kernel<<<dim3((unsigned)((2 * p->P + 255) / 256)), dim3(256), 0, st>>>(A);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// API: ops and steps
// ------------------------------------------------------------------------------------------
// Host-side plan validation (no device work, usable without a GPU): walks every launch function of a forward
// [+ backward] pass for the given call geometry with the dry-run guard set, i.e. runs all their geometry checks
// (DMA instructions vs plane / slot sizes, slot rings vs windows in flight, counted-wait ranges, LDS budgets).
extern "C" int bnn_plan_validate(BnnPlan* p, int32_t mode, int32_t particles, int32_t batch, int32_t train) {
  if (!p) return fail(BNN_E_INVALID, "null plan");
  BnnElboArgs a{};
  a.batch = batch;
  a.particles = particles;
  a.global_batch = batch;
  a.dataset_size = 1.0;
  a.prior_scale = 1.0;
  a.mode_override = mode;
  a.with_obs = 0;
  const bool was_bound = p->bound;
  p->bound = true;   // geometry only: no buffer is dereferenced under the guard
  Ctx c;
  int rc = make_ctx(p, &a, nullptr, nullptr, train != 0, &c);
  if (rc == 0) {
    g_dry = true;
    rc = do_forward(p, &a, &c, nullptr);
    if (rc == 0 && train) rc = do_backward(p, &a, &c);
    g_dry = false;
  }
  p->bound = was_bound;
  return rc;
}

extern "C" int bnn_sample_weights(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nz, stream, true, &c));
  BNN_TRY(prepare_noise(p, a, nz, &c));
  return do_sample(p, a, &c);
}

extern "C" int bnn_forward(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nz, stream, true, &c));
  if (!a->x) return fail(BNN_E_INVALID, "x is null");
  BNN_TRY(prepare_noise(p, a, nz, &c));
  return do_forward(p, a, &c, a->x);
}

extern "C" int bnn_head_nll(BnnPlan* p, const BnnElboArgs* a, const BnnElboOut* out, void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nullptr, stream, true, &c));
  HIP_TRY(hipMemsetAsync((double*)((char*)p->bufs.workspace + p->o_acc) + (p->d.max_particles + 1), 0,
                         sizeof(double) * (p->d.max_particles + 1), c.st));
  BNN_TRY(do_head(p, a, &c, out ? out->preds : nullptr, true));
  return do_loss(p, a, &c, out, false);
}

extern "C" int bnn_backward(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nz, stream, true, &c));
  BNN_TRY(prepare_noise(p, a, nz, &c));
  return do_backward(p, a, &c);
}

extern "C" int bnn_grad_finalize(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nz, stream, true, &c));
  BNN_TRY(prepare_noise(p, a, nz, &c));
  return do_finalize(p, a, &c);
}

extern "C" int bnn_clipped_adam(BnnPlan* p, const BnnAdamArgs* ad, void* stream) {
  if (!p || !ad) return fail(BNN_E_INVALID, "null argument");
  if (!p->bound) return fail(BNN_E_UNBOUND, "plan not bound");
  return do_adam(p, ad, (hipStream_t)stream);
}

extern "C" int bnn_elbo_step(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, const BnnAdamArgs* adam,
                             const BnnElboOut* out, void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nz, stream, true, &c));
  if (!a->with_obs) return fail(BNN_E_INVALID, "bnn_elbo_step needs with_obs = 1");
  if (trunk_ok(p, &c) || tf_ok(p, &c)) c.fuse_x = a->x;
  BNN_TRY(prepare_noise(p, a, nz, &c));
  BNN_TRY(do_sample(p, a, &c));
  BNN_TRY(do_forward(p, a, &c, a->x));
  c.head_fused = mlp_ok(p, &c);
  c.head_preds = out ? out->preds : nullptr;
  BNN_TRY(prepare_fused_tail(p, a, &c));
  if (!c.head_fused) BNN_TRY(do_head(p, a, &c, c.head_preds, true));
  BNN_TRY(do_backward(p, a, &c));
  BNN_TRY(do_finalize(p, a, &c, out, true, adam));   // + the loss scalars and the optimizer update (two launches less)
  return 0;
}

static int check_dropout(const BnnPlan* p, const Ctx* c) {
  if (!drop_on(c)) return 0;
  if (c->drop->p >= 1.0) return fail(BNN_E_INVALID, "dropout rate %g", c->drop->p);
  if (!tf_ok(p, c) || c->em != EM_PLAIN || c->S != 1 || p->o_dks == 0)
    return fail(BNN_E_INVALID, "MC-dropout runs on the exact-fp32 Inception plan with deterministic weights (prec f32, one particle)");
  const int ninj = (c->drop->keep_act1 != nullptr) + (c->drop->keep_act2 != nullptr) + (c->drop->keep_h != nullptr);
  if (ninj != 0 && ninj != 3) return fail(BNN_E_INVALID, "inject all three dropout keep masks or none");
  return 0;
}

// frequentist siblings (frequentist.py:39-48,173-178) on the plain-contraction kernels: weights = mu (zero noise), no KL
extern "C" int bnn_det_step(BnnPlan* p, const BnnDetArgs* d, const BnnAdamArgs* adam, const BnnElboOut* out, void* stream) {
  if (!p || !d) return fail(BNN_E_INVALID, "null argument");
  if (d->objective != 1 && d->objective != 2) return fail(BNN_E_INVALID, "objective must be 1 (gaussian NLL) or 2 (MSE)");
  if (!d->x || !d->y) return fail(BNN_E_INVALID, "x / y must be set");
  BnnElboArgs a{};
  a.x = d->x;
  a.y = d->y;
  a.batch = d->batch;
  a.particles = 1;
  a.global_batch = d->batch;
  a.dataset_size = (double)d->batch;   // N / B = 1: the likelihood term is a plain sum over the batch
  a.prior_loc = 0.0;
  a.prior_scale = 1.0;
  a.mode_override = BNN_MODE_NORMAL;
  a.with_obs = 1;
  a.scaled = 0;
  Ctx c;
  BNN_TRY(make_ctx(p, &a, nullptr, stream, true, &c));
  c.objective = d->objective;
  c.drop = d->dropout;
  c.c = (float)(1.0 / d->batch);   // mean over the batch ...
  c.n_over_b = 1.f;
  c.scale_ll = (float)(1.0 / d->batch);
  // weights = mu: the weight noise of the one particle is zero
  float* eps = ws_f(p, p->o_eps);
  HIP_TRY(hipMemsetAsync(eps, 0, (size_t)p->P * 4, c.st));
  BnnNoise nz{};
  nz.eps_w = eps;
  BNN_TRY(prepare_noise(p, &a, &nz, &c));
  BNN_TRY(do_sample(p, &a, &c));
  BNN_TRY(do_forward(p, &a, &c, a.x));
  BNN_TRY(prepare_fused_tail(p, &a, &c));
  BNN_TRY(do_head(p, &a, &c, out ? out->preds : nullptr, true));
  BNN_TRY(do_backward(p, &a, &c));
  const float c_kl = c.c;
  c.c = 0.f;   // ... and no KL / prior term in the gradient
  BNN_TRY(do_finalize(p, &a, &c));
  c.c = c_kl;
  BNN_TRY(do_loss(p, &a, &c, out, true));
  if (adam) BNN_TRY(do_adam(p, adam, c.st));
  return 0;
}

extern "C" int bnn_det_forward(BnnPlan* p, const float* x, int32_t batch, const BnnDropout* dropout, float* preds_b2, void* stream) {
  if (!p || !x || !preds_b2) return fail(BNN_E_INVALID, "null argument");
  BnnElboArgs a{};
  a.x = x;
  a.batch = batch;
  a.particles = 1;
  a.global_batch = batch;
  a.dataset_size = 1.0;
  a.prior_scale = 1.0;
  a.mode_override = BNN_MODE_NORMAL;
  Ctx c;
  BNN_TRY(make_ctx(p, &a, nullptr, stream, false, &c));
  c.drop = dropout;
  float* eps = ws_f(p, p->o_eps);
  HIP_TRY(hipMemsetAsync(eps, 0, (size_t)p->P * 4, c.st));   // weights = mu
  BnnNoise nz{};
  nz.eps_w = eps;
  BNN_TRY(prepare_noise(p, &a, &nz, &c));
  BNN_TRY(do_sample(p, &a, &c));
  BNN_TRY(do_forward(p, &a, &c, x));
  return do_head(p, &a, &c, preds_b2, false);
}

extern "C" int bnn_elbo_evaluate(BnnPlan* p, const BnnElboArgs* a, const BnnNoise* nz, const BnnElboOut* out,
                                 void* stream) {
  Ctx c;
  BNN_TRY(make_ctx(p, a, nz, stream, false, &c));
  BNN_TRY(prepare_noise(p, a, nz, &c));
  BNN_TRY(do_sample(p, a, &c));
  if (a->with_obs) {
    BNN_TRY(do_forward(p, a, &c, a->x));
    BNN_TRY(do_head(p, a, &c, out ? out->preds : nullptr, false));
  }
  return do_loss(p, a, &c, out, false);
}

extern "C" int bnn_predict(BnnPlan* p, const float* x, int32_t batch, int32_t particles, const BnnNoise* nz,
                           float* preds_sb2, float* out4, void* stream) {
  if (!p || !x) return fail(BNN_E_INVALID, "null argument");
  if (!p->bound) return fail(BNN_E_UNBOUND, "plan not bound");
  if (batch < 1 || batch > p->d.max_batch) return fail(BNN_E_INVALID, "batch %d outside [1, %d]", batch, p->d.max_batch);
  if (particles < 1) return fail(BNN_E_INVALID, "particles must be >= 1");
  int chunk = (int)std::min<long>(p->d.max_particles, p->cap_windows / batch);
  if (chunk < 1) return fail(BNN_E_INVALID, "plan capacity (%ld windows) below one particle of batch %d", p->cap_windows, batch);
  float* preds_all = preds_sb2;
  if (!preds_all) {
    if (particles > p->d.max_particles)
      return fail(BNN_E_INVALID, "without a preds buffer particles must be <= max_particles (%d)", p->d.max_particles);
    preds_all = ws_f(p, p->o_preds);
  }
  if (nz && (nz->lrt_eps || nz->sign_in || nz->sign_out))
    return fail(BNN_E_INVALID, "bnn_predict samples weights plainly (bayesian.py:231-250): only eps_w / radial_r may be injected");
  for (int s0 = 0; s0 < particles; s0 += chunk) {
    const int sc = std::min(chunk, particles - s0);
    BnnElboArgs a{};
    a.x = x;
    a.y = nullptr;
    a.batch = batch;
    a.particles = sc;
    a.global_batch = batch;
    a.global_batch_offset = 0;
    a.dataset_size = 1.0;
    a.prior_loc = 0.0;
    a.prior_scale = 1.0;
    a.mode_override = p->d.mode == BNN_MODE_RADIAL ? BNN_MODE_RADIAL : BNN_MODE_NORMAL;
    a.with_obs = 0;
    a.scaled = 0;
    BnnNoise n2{};
    if (nz) n2 = *nz;
    if (n2.eps_w) n2.eps_w += (long)s0 * p->P;
    if (n2.radial_r) n2.radial_r += (long)s0 * p->n_sites;
    Ctx c;
    BNN_TRY(make_ctx(p, &a, &n2, stream, false, &c));
    c.s_base = s0;
    c.x_planes_ready = s0 > 0;   // same windows for every chunk of particles
    BNN_TRY(prepare_noise(p, &a, &n2, &c));
    if (s0 > 0) p->acc_clean = true;   // the KL / log-likelihood accumulators are not read on this path: no fill per chunk
    BNN_TRY(do_sample(p, &a, &c));
    BNN_TRY(do_forward(p, &a, &c, x));
    BNN_TRY(do_head(p, &a, &c, preds_all + (long)s0 * batch * 2, false));
  }
  if (out4) {
    predict_finish_kernel<<<dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(preds_all, batch, particles, out4);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

extern "C" int bnn_export_noise(BnnPlan* p, const BnnElboArgs* a, uint64_t seed, uint64_t step, float* eps_w,
                                float* radial_r, float* const* lrt_eps, float* const* sign_in, float* const* sign_out,
                                void* stream) {
  Ctx c;
  BnnNoise nz{};
  nz.seed = seed;
  nz.step = step;
  BNN_TRY(make_ctx(p, a, &nz, stream, false, &c));
  const int S = c.S, B = c.B;
  const uint32_t st32 = (uint32_t)step;
  if (eps_w) {
    const long n = ((p->P + 3) / 4) * S;
    gen_eps_w_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st>>>(eps_w, p->P, S, seed, st32);
  }
  if (radial_r)
    gen_radial_r_kernel<<<dim3((S * p->n_sites + 255) / 256), dim3(256), 0, c.st>>>(radial_r, p->n_sites, S, seed, st32);
  CallGeom cg{};
  cg.S = S;
  cg.B = B;
  cg.Bglob = std::max(a->global_batch, B);
  cg.goff = a->global_batch_offset;
  const long ex = (long)S * B;
  uint32_t* tmp = (uint32_t*)ws_f(p, p->o_sign_in);
  for (int i = 0; i < p->n_layers; ++i) {
    const LayerDesc& l = p->layers[i];
    const int Lrows = l.is_conv ? p->d.win_length : 1;
    if (lrt_eps && lrt_eps[i]) {
      const long rows = ex * Lrows;
      const long n = rows * ((l.cout + 3) / 4);
      export_lrt_eps_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st>>>(lrt_eps[i], rows, l.cout, l.cout_p16, Lrows, cg, i, seed, st32);
    }
    if (sign_in && sign_in[i]) {
      const long n = ex * ((l.sign_in_words + 3) / 4);
      gen_signs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st>>>(tmp, l.sign_in_words, S, B, cg.Bglob, cg.goff, i, NK_SIGN_IN, seed, st32);
      const long m = ex * l.cin_img;
      unpack_signs_kernel<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c.st>>>(tmp, sign_in[i], ex, l.cin_img, l.sign_in_words);
    }
    if (sign_out && sign_out[i]) {
      const long n = ex * ((l.sign_out_words + 3) / 4);
      gen_signs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st>>>(tmp, l.sign_out_words, S, B, cg.Bglob, cg.goff, i, NK_SIGN_OUT, seed, st32);
      const long m = ex * l.cout;
      unpack_signs_kernel<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c.st>>>(tmp, sign_out[i], ex, l.cout, l.sign_out_words);
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// per-kernel timing (HIP events on the launch stream)
// ------------------------------------------------------------------------------------------
extern "C" int bnn_profile_enable(BnnPlan* p, int on) {
  if (!p) return fail(BNN_E_INVALID, "null plan");
  p->prof.on = on != 0;
  p->prof.used = 0;
  return 0;
}

// window store (data feed): x_out[i] = window idx[i] of an HBM-resident set, y_out[i] = its label.  Replaces the
// per-item reads of LmdbDataset.__getitem__ (data/lmdb_utils.py:184-194) + NCMAPSSLmdbDataset.__getitem__
// (data/ncmapss/dataset.py:13-16) + the DataLoader collate for a whole batch.
extern "C" int bnn_gather_windows(const float* x_all, const float* y_all, const int64_t* idx, int64_t n, int64_t n_windows,
                                  int32_t win_length, int32_t n_features, int32_t feature_major, float* x_out, float* y_out,
                                  void* stream) {
  if (n < 0 || n_windows < 0 || win_length <= 0 || n_features <= 0) return fail(BNN_E_INVALID, "bad geometry");
  if (n == 0) return 0;   // an empty batch carries no buffers
  if (!x_all || !idx || !x_out || (y_all && !y_out)) return fail(BNN_E_INVALID, "null argument");
  const long total = (long)n * win_length * n_features;
  gather_windows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      x_all, y_all, (const long*)idx, (long)n, (long)n_windows, win_length, n_features, feature_major, x_out, y_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Restricts the recorder to the given tags (n = 0: every launch).  Two events per recorded launch sit
// between the kernels on the stream, so bench.py's timed region records the dominant symbol only.
extern "C" int bnn_profile_select(BnnPlan* p, const int32_t* tags, int32_t n) {
  if (!p || (n > 0 && !tags)) return fail(BNN_E_INVALID, "null argument");
  p->prof.filter = n > 0;
  for (bool& b : p->prof.sel) b = false;
  for (int i = 0; i < n; ++i)
    if (tags[i] >= 0 && tags[i] < 256) p->prof.sel[tags[i]] = true;
  return 0;
}

// Kernel symbol last launched under `tag` while recording (as rocprofv3 prints it, without the argument
// list); empty if none.
extern "C" int bnn_profile_name(BnnPlan* p, int32_t tag, char* buf, int32_t cap) {
  if (!p || !buf || cap <= 0 || tag < 0 || tag >= 256) return fail(BNN_E_INVALID, "bad argument");
  snprintf(buf, (size_t)cap, "%s", p->prof.names[tag].c_str());
  return 0;
}

// Synchronises the recorded events and returns, per (kind, group) tag, the summed duration in
// milliseconds and the launch count.  tags/ms/count must hold `cap` entries; *n receives the
// number of distinct tags.  Resets the recorder.
extern "C" int bnn_profile_read(BnnPlan* p, int32_t* tags, double* ms, int64_t* count, int32_t cap, int32_t* n) {
  if (!p || !tags || !ms || !count || !n) return fail(BNN_E_INVALID, "null argument");
  int m = 0;
  for (size_t k = 0; k + 1 < p->prof.used; k += 2) {
    HIP_TRY(hipEventSynchronize(p->prof.ev[k + 1]));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, p->prof.ev[k], p->prof.ev[k + 1]));
    const int tg = p->prof.tag[k / 2];
    int j = 0;
    for (; j < m; ++j)
      if (tags[j] == tg) break;
    if (j == m) {
      if (m >= cap) return fail(BNN_E_INVALID, "profile table too small");
      tags[m] = tg;
      ms[m] = 0;
      count[m] = 0;
      ++m;
    }
    ms[j] += t;
    count[j] += 1;
  }
  *n = m;
  p->prof.used = 0;
  return 0;
}
