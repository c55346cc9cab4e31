/* tfusion.h -- C ABI of libtfusion_hip.so: the MI355X (gfx950) kernels and encoder runtime behind the
 * TransFusion cross-fusion block.
 *
 * The reference has no FFI seam: its boundary is the Python nn.Module contract of
 *   modeling/cross_fusion/ego_fusion/cross_f_box_layers.py:13-108   (CrossTransformerModuleBox)
 *   modeling/cross_fusion/ego_fusion/cross_f_box_wrapper.py:165-230 (CrossFusionBoxWrapper.forward)
 * whose arithmetic lives in torch 1.9.1 (restated in .../ego_fusion/torch18_adapters.py).  The entry points below
 * are what transfusion_amd's modules of the same names bind (via ctypes) underneath that contract; each entry
 * cites the reference lines whose work it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); nothing here allocates or frees;
 *   - tf_stream_t is a hipStream_t; every call only enqueues work on it (safe under graph capture);
 *   - return 0 = ok, < 0 = argument error, > 0 = hipError_t; tf_last_error() gives a message (thread-local);
 *   - entries are re-entrant and may be called from any thread (autograd backward thread included);
 *   - bf16 tensors are row-major with a leading dimension in ELEMENTS that is a multiple of 8.
 */
#ifndef TFUSION_H_
#define TFUSION_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tf_stream_t;   /* hipStream_t */
#define TF_ABI_VERSION 10
#define TF_MAX_LAYERS 16
#define TF_MAX_GROUPS 8     /* ragged groups (ABI v8): at most this many row ranges of unequal size in one grouped launch */

enum TfEpilogue {
  TF_EPI_NONE = 0,            // C = acc
  TF_EPI_BIAS = 1,            // C = acc + bias
  TF_EPI_BIAS_GELU_DROP = 2,  // C = U = acc + bias ; C2 = dropout(gelu(U))
  TF_EPI_BIAS_DROP_RES = 3,   // C = R + dropout(acc + bias)
  TF_EPI_ADD = 4,             // C = acc + R
  TF_EPI_DGELU_DROP = 5,      // C = acc * mask/(1-p) * gelu'(R)
  TF_EPI_BIAS_GELU_DROP_G = 6,// C = G = mask/(1-p) * gelu'(acc + bias) ; C2 = dropout(gelu(acc + bias)): the forward already holds erf
                              //   and exp, so the activation derivative costs 3 more operations here instead of ~20 in the backward
  TF_EPI_MUL = 7              // C = acc * R   (FFN-down dgrad against the stored G)
};

typedef struct TfGemmArgs {
  const void* A; int lda;     // [M,K] bf16
  const void* W; int ldw;     // [N,K] bf16
  void* C; int ldc;           // [M,N] bf16
  const float* bias;          // [N] fp32 or null
  const void* R; int ldr;     // [M,N] bf16 aux input (residual / pre-activation)
  void* C2; int ldc2;         // second output (EPI_BIAS_GELU_DROP)
  int M, N, K;
  int epilogue;
  unsigned drop_thr, drop_key; float drop_scale;   // drop_thr == 0 -> no dropout
  // fp8 operands (BASELINE configs[4]: fp8 MFMA for the QKV / FFN projections, fp32 accumulate).  fp8 != 0: A and W hold OCP
  // e4m3 bytes (lda / ldw / K in elements = bytes, K % 64 == 0) as produced by tf_quant_rows_fp8, and the result is
  // acc * scale_a[m] * scale_w[n] (+ bias ...).  Epilogues NONE, BIAS, BIAS_DROP_RES, BIAS_GELU_DROP_G.
  int fp8; const float* scale_a; const float* scale_w;
  // activation of the two *_GELU_* epilogues and of DGELU: 0 = exact (erf) GELU, 1 = ReLU -- the reference constructor's
  // activ_f ("gelu" in the shipped YAMLs, "relu" its default: cross_f_box_layers.py:26,56)
  int act;
  // fp32-accuracy mode (BASELINE configs[2], run.precision: 32): every tensor is a PAIR of bf16 planes, value = hi + lo
  // (16 significant bits), and the product is three bf16 MFMA passes into one fp32 accumulator:
  // A_hi.W_hi + A_lo.W_hi + A_hi.W_lo.  A_lo != null selects it; then W_lo and C_lo are required, and R_lo / C2_lo
  // wherever R / C2 are.  Planes share the leading dimension of their hi plane.  Not combinable with fp8.
  const void* A_lo; const void* W_lo; void* C_lo; const void* R_lo; void* C2_lo;
  // grouped launch (no reference counterpart: the wrapper's FPN levels, cross_f_box_wrapper.py:177-212, run identical encoders with
  // different weights): groups > 1 splits the M rows into `groups` equal ranges; range g multiplies by the weight / bias / scale_w
  // tensors that start g * w_gstride BYTES after W / bias / scale_w (W_lo likewise).  A, C, R, C2, scale_a are indexed by global row.
  int groups; long long w_gstride;
  // fp32-accuracy mode, epilogues NONE / BIAS (ABI v7): c_is_f32 != 0 -> C is an fp32 [M, ldc] tensor written directly (C_lo unused):
  // the tokens K1 hands the encoder and the gradient K9 hands back are fp32 tensors, not plane pairs
  int c_is_f32;
  // RAGGED groups (ABI v8; the reference's real FPN geometry gives level 0 four times the tokens of levels 1 - 3,
  // cross_fusion_config_sym_ego_res50.yml:8-17): group_rows[0] > 0 -> range g holds group_rows[g] rows (groups <= TF_MAX_GROUPS, the
  // counts add up to M) instead of M / groups; all zero = equal ranges.  Tiles never straddle two ranges either way.
  int group_rows[TF_MAX_GROUPS];
} TfGemmArgs;

/* row-wise fp8 (e4m3) quantisation: dst[r][c] = fp8(src[r][c] / scale[r]), scale[r] = max|src[r][:]| / 448 (1 for an all-zero row);
 * src bf16 [rows, ld_src] (cols valid, cols % 8 == 0), dst bytes [rows, ld_dst] with columns [cols, ld_dst) zero-filled */
int tf_quant_rows_fp8(const void* src, int ld_src, void* dst, int ld_dst, float* scale, int rows, int cols, tf_stream_t s);

typedef struct TfWgradArgs {
  const void* dY; int ldy;    // [M,N] bf16
  const void* X; int ldx;     // [M,K] bf16
  float* dW; int lddw;        // [n_src, k_src] fp32, accumulated with atomics
  float* db;                  // [n_src] fp32 or null
  const void* zeros;          // >= 256 B of zeros (reduction-tail source)
  int M, N, K;                // padded extents of the bf16 operands
  int rg, rgp, n_src;         // padded row n -> source row (n/rgp)*rg + n%rgp, valid iff n%rgp < rg
  int cg, cgp, k_src;         // same for columns
  int m_chunk;                // rows of M per block (0 = auto)
  const void* dY_lo; const void* X_lo;   // fp32-accuracy mode (see TfGemmArgs): lo planes, dW += dY_hi^T X_hi + dY_lo^T X_hi + dY_hi^T X_lo
  // grouped launch (see TfGemmArgs.groups): `groups` equal row ranges of dY / X, range g accumulates into the dW / db that start
  // g * dw_gstride BYTES after dW / db; m_chunk counts rows inside a group
  int groups; long long dw_gstride;
  int group_rows[TF_MAX_GROUPS];          // ragged row ranges (see TfGemmArgs.group_rows; tf_gemm_wgrad_multi only), all zero = equal
} TfWgradArgs;
/* Several weight gradients as ONE launch (tf_gemm_wgrad_multi): a layer's four linears (torch18_adapters.py:683-685,608,111: in-proj,
 * out-proj, linear1, linear2) are ready together at the end of the layer's backward and fill the chip together at two row chunks per
 * output tile, where each alone needs 5 - 14 (every chunk adds one fp32 |dW| of atomic traffic).  Problems may differ in every extent
 * (M included: the wrapper's FPN levels); `groups` of a problem expands into that many problems; zeros and m_chunk are ignored.
 * At most TF_WGRAD_MULTI_MAX problems after the expansion; one arithmetic mode (lo planes on all or on none). */
#define TF_WGRAD_MULTI_MAX 16


typedef struct TfAttnArgs {
  const void* qkv; int ld_qkv;   // [B*S, 3*H*HDP] bf16, column = (which*H + head)*HDP + e
  void* out; int ld_out;         // [B*S, H*HDP] bf16
  float* lse;                    // [B,H,S] fp32, log2 domain: m*scale*log2e + log2(l)
  const uint8_t* key_mask;       // [B,S] 1 = ignore key, or null
  int B, S, H, HDP;
  float scale;                   // 1/sqrt(true head dim)
  unsigned drop_thr, drop_key; float drop_scale;
  const void* drop_bits;         // [B*H*S, ceil(S/64)] u64 keep-bitmask from tf_attn_dropmask (required when drop_thr != 0)
  const void* block_bits;        // optional [S, ceil(S/64)] u64, bit k of row q = 1: query q does not attend key k (all batches and
                                 //   heads) -- the bool attn_mask the reference builds from vis_tokens_mask, cross_f_box_layers.py:87-95
  // backward only
  const void* dout; int ld_dout; // [B*S, H*HDP] bf16
  void* dqkv; int ld_dqkv;       // [B*S, 3*H*HDP] bf16
  float* delta;                  // [B,H,S] fp32 workspace: rowsum(dO . O)
  // fp32-accuracy mode (see TfGemmArgs): lo planes of qkv / out / dout / dqkv, same leading dimensions.  qkv_lo != null selects
  // the split kernels (attn_x3.hip): every S x S x hd product is three bf16 MFMA passes, probabilities are split in registers.
  const void* qkv_lo; void* out_lo; const void* dout_lo; void* dqkv_lo;
  // Cross attention with its own query set (QKVEncoder, modeling/cross_fusion/cross_qkv_layers.py:51-81: queries from one modality,
  // keys / values from the concatenation).  q != null: queries are q [B*Sq, ld_q] (column = head*HDP + e) instead of the Q third of
  // qkv, whose K and V thirds [B*S, .] still hold the S keys; out / lse / delta / dout / the dropout bitmask rows then have Sq rows
  // per (batch, head), and the query gradient goes to dq [B*Sq, ld_dq] (dqkv keeps dK, dV).  block_bits must be null.
  const void* q; int ld_q; int Sq; void* dq; int ld_dq;
  const void* q_lo; void* dq_lo;          // their lo planes in the fp32-accuracy mode
  // Packed (ragged) batches: cu_rows != null -> sample b owns rows cu_rows[b] .. cu_rows[b+1]-1 of qkv / out / dout / dqkv (its length
  // S_b = cu_rows[b+1] - cu_rows[b] <= S keys and queries, all of them attended: key_mask and q must be null); lse / delta / drop_bits
  // keep their dense [B,H,S] row indexing with sample-local row numbers.  S stays the maximum length (grid and bitmask geometry).
  const int* cu_rows;                     // [B+1] int32 device array, or null: every sample has S rows starting at b*S
  // Backward, optional workspace of tf_attn_ds_bytes(B, H, S) bytes.  With it (head dims <= 192, self attention) S and dP are computed
  // ONCE: a small kernel forms delta, the dK / dV kernel writes its dS tiles here (bf16, one coalesced 1-KiB chunk per 16 keys x 32
  // queries) and a thin kernel forms dQ = dS . K from them -- instead of a dQ kernel that recomputes S and dP.  null: the two-kernel form.
  // fp32-accuracy mode (qkv_lo != null): 2 x tf_attn_ds_bytes(B, H, S) bytes -- a hi and a lo plane of dS, any head dim the mode serves.
  void* ds_work;
  // how many tf_attn_ds_bytes-sized planes ds_work holds.  0 = the mode's minimum (1 in bf16, 2 in the fp32-accuracy mode).  4 in the
  // fp32-accuracy mode: the dK launch also leaves Pd = P . keep / (1 - p) there (hi + lo), and dV is formed from it without recomputing S.
  int ds_planes;
  // block-sparse tiles of a block mask (ABI v7; vis_mask_type "local_k", reference utils.py:14-30: a visual token attends a (2k+1)^2
  // window, so most 64-key tiles of a visual query block are blocked for every one of its queries).  Derived from block_bits by
  // tf_attn_block_skip; null = every tile is visited and masked element-wise (same results, bit for bit: a skipped tile holds only
  // probabilities that are exactly 0).
  //   block_skip_q [ceil(S/128)] u64: bit t of word qb = every query of block qb has all 64 keys of tile t blocked (forward, dQ);
  //   block_skip_k [ceil(S/128)] u64: bit j of word kb = every key of block kb is blocked for all 32 queries of tile j (dK / dV).
  // (S <= 4096 / 2048 respectively; beyond that the words are zero)
  // Both must be null when block_bits is null (tf_attn_fwd / tf_attn_bwd return -9 otherwise).  tf_attn_block_skip never marks EVERY tile
  // of a query block: a block whose rows attend nothing keeps tile 0, so its rows come out as with element-wise masking, bit for bit.
  const void* block_skip_q; const void* block_skip_k;
} TfAttnArgs;
/* the two maps above from block_bits [S, ceil(S/64)] u64: skip_q and skip_k each ceil(S/128) u64 words (device memory) */
int tf_attn_block_skip(const void* block_bits, int S, void* skip_q, void* skip_k, tf_stream_t s);


// ---- row-wise kernels (rowops.hip) ----
// LayerNorm over the first d of ld columns; rows are addressed as (r / rows_per_group) * group_stride +
// r % rows_per_group so the final LN can pick the visual rows of [B,S,*].
typedef struct TfLnArgs {
  const void* x; int ldx;        // bf16 in
  void* y; int ldy; int y_is_f32;// bf16 (or fp32) out; pad columns [d, ldy) are zero-filled when bf16
  const float* gamma; const float* beta;
  float* mean; float* rstd;      // [rows] fp32 (may be null in inference)
  int rows, d;
  int rows_per_group, x_group_stride, y_group_stride;   // in rows
  float eps;
  // backward
  const void* dy; int lddy; int dy_is_f32;
  void* dx; int lddx;            // bf16 dz
  void* dx_drop; int lddxd;      // optional second output: dz * mask/(1-p) (dropout that preceded the residual add)
  unsigned drop_thr, drop_key; float drop_scale; int drop_ld;
  float* dgamma; float* dbeta;   // fp32, atomically accumulated
  const void* dres; int lddres;  // optional bf16 tensor added to dy before the backward (residual-path gradient)
  // fp32-accuracy mode: lo planes of the bf16 tensors above (value = hi + lo); each is used iff its hi plane is bf16 and it is non-null
  const void* x_lo; void* y_lo; const void* dy_lo; void* dx_lo; void* dx_drop_lo; const void* dres_lo;
  // packed batches: group g of the x side (x, dx, dx_drop, dres) starts at row x_group_row0[g] instead of g * x_group_stride
  const int* x_group_row0;       // [ceil(rows / rows_per_group)] int32 device array or null
  // parameter groups (see TfGemmArgs.groups): pgroups equal ranges of the `rows`, range g normalised with the gamma / beta that start
  // g * p_gstride BYTES after gamma / beta (dgamma / dbeta likewise)
  int pgroups; long long p_gstride;
  int group_rows[TF_MAX_GROUPS];          // ragged parameter groups (see TfGemmArgs.group_rows): rows of range g, all zero = equal ranges
  // x-side row of EVERY row given explicitly (ragged visual rows of a packed batch: row r of the output is token r of the concatenated
  // visual tokens, x_row_map[r] its packed row); overrides rows_per_group / x_group_stride / x_group_row0 on the x side.  null: unused
  const int* x_row_map;
} TfLnArgs;

typedef struct TfAssembleArgs {
  const void* vis; int vis_is_f32; int ld_vis;   // [B,Nv,d]
  const void* lang; int lang_is_f32; int ld_lang;// [B,Nl,d]
  const float* pe;                               // [>=Nv, d] fp32, or null (nothing added)
  const float* pe_lang;                          // [>=Nl, d] fp32 or null: lang_pos_embedding (cross_f_box_layers.py:77-78)
  const float* kind_v; const float* kind_l;      // [d]
  void* out; int ld_out;                         // [B,S,ld_out] bf16
  int B, Nv, Nl, d;
  unsigned drop_thr, drop_key; float drop_scale;
  // backward
  const void* dout; int ld_dout;                 // bf16
  void* dvis; int dvis_is_f32; int ld_dvis;
  void* dlang; int dlang_is_f32; int ld_dlang;
  float* dkind_v; float* dkind_l;
  void* out_lo; const void* dout_lo;             // fp32-accuracy mode: lo planes of out / dout
  // packed batches (padded language tokens dropped): out / dout hold `rows` token rows and row_map[m] = b * (Nv + Nl) + s names the
  // (sample, position) row m was gathered from; dlang rows of dropped tokens are NOT written (the caller zero-fills dlang).
  const int* row_map; int rows;                  // null / 0: dense, row m = b * S + s
  // parameter groups (see TfGemmArgs.groups): sample b belongs to group b / (B / pgroups); group g adds the kind embeddings that start
  // g * p_gstride BYTES after kind_v / kind_l (dkind_v / dkind_l likewise); with a row_map the rows must be group-major
  int pgroups; long long p_gstride;
  // ragged groups (ABI v8, packed batches only): the samples of group g carry group_nv[g] <= Nv visual tokens -- vis / dvis are then the
  // CONCATENATION [sum_g (B / pgroups) group_nv[g], d] (group-major, sample, token), positions >= group_nv[g] of the dense index space
  // (b * (Nv + Nl) + s) do not exist -- and group g owns group_rows[g] of the packed rows (backward: whose kind-embedding gradients
  // a workgroup accumulates).  group_nv[0] == 0: every sample has Nv visual tokens, equal ranges.
  int group_nv[TF_MAX_GROUPS]; int group_rows[TF_MAX_GROUPS];
} TfAssembleArgs;

// rowsum(dO . O) per (b, head, s)

// fp32 [rows, cols] (param layout) -> bf16 padded/grouped shadow [rows_p, cols_p] and/or its transpose
typedef struct TfPackArgs {
  const float* src; int rows, cols;      // source (parameter layout)
  void* dst; int ld_dst;                 // [rows_p, ld_dst] bf16 or null
  void* dst_t; int ld_dst_t;             // [cols_p, ld_dst_t] bf16 (transpose) or null
  int rows_p, cols_p;
  int rg, rgp, cg, cgp;                  // group maps as in TfWgradArgs
  int dst_is_f32;                        // biases keep fp32
  int residual;                          // 1: store bf16(w - bf16(w)) instead of bf16(w) -- the lo plane of the fp32-accuracy mode
} TfPackArgs;

// dst[map(r), 0:cols] = src[map(r), 0:cols] with dtype conversion; columns [cols, ld_dst) of a bf16 dst are zeroed.
// src == null writes zeros.
typedef struct TfCopyRowsArgs {
  const void* src; int src_is_f32; int ld_src; int src_rpg, src_gstride;
  void* dst; int dst_is_f32; int ld_dst; int dst_rpg, dst_gstride;
  int rows, cols;
  const void* src_lo; void* dst_lo;      // fp32-accuracy mode: lo plane of a bf16 src (added) / of a bf16 dst (residual written)
  // optional per-row indirections (int32 [rows], device): row r reads src row src_row_map[r] (< 0: zeros) instead of its group-mapped
  // row, and writes dst row dst_row_map[r] (< 0: nothing is written)
  const int* src_row_map; const int* dst_row_map;
  // or per-GROUP starts (packed batches): group g of src / dst begins at row src_group_row0[g] / dst_group_row0[g] instead of g * gstride
  const int* src_group_row0; const int* dst_group_row0;
} TfCopyRowsArgs;
// key_mask[b, s] = s < Nv ? 0 : lang_pad_mask[b, s - Nv]


// fused RAdam over a flat fp32 buffer (runner/metrics_losses/radam_optim.py:30-104 arithmetic)
typedef struct TfRadamArgs {
  float* p; const float* g; float* m; float* v; long long n;
  float lr, beta1, beta2, eps, weight_decay;
  float beta2_t;   // beta2^step
  float bias1;     // 1 - beta1^step
  float n_sma; float step_size; int rectified;   // host-computed schedule terms
  float grad_scale;                               // multiplies g first (loss-scale / 1/world)
  const float* sumsq; float clip;                 // optional device scalar sum(g^2): global-norm clip without a host sync
  // optional (a step captured in a HIP graph, see tf_clock_ptr): the step number is step0 + *step_clock and beta2_t, bias1, n_sma,
  // step_size, rectified are formed on the device from it (degenerated_to_sgd: radam_optim.py:31,80-84)
  const uint32_t* step_clock; long long step0; int degenerated_to_sgd;
  int zero_grad;                                  // != 0: g[i] = 0 after it has been read (the next step's zero fill, fused: g is writable then)
  const float* lr_dev;                            // optional device scalar: the learning rate is *lr_dev instead of lr -- a step captured in
                                                  // a HIP graph would otherwise replay the rate it was captured with under the reference's
                                                  // warm-up / multi-step schedulers (runner/nao/abc_nao_trainer.py:197-212)
} TfRadamArgs;

// language auxiliary head, pooling stage (modeling/cross_fusion/ego_fusion/lm_layers.py:59-72, PoolPredictor.forward):
//   pooled[b, c] = mean_l / max_l ( x[b, l, c] * mask[b, l] )      (the mean divides by L, padded rows count as zeros)
//   feat         = GELU( LayerNorm(pooled) )                        (LN iff ln_w != null, GELU iff gelu != 0)
// The two Linear heads that follow (mlp_noun / mlp_verb, :74-77) are tf_gemm_fwd calls.
typedef struct TfLmPoolArgs {
  const void* x; int x_is_f32;       // [B, L, d] dense
  const uint8_t* mask;               // [B, L], 1 = real token (HF convention, wrapper :226), or null
  int B, L, d;                       // d % 8 == 0, d <= 2048
  int type;                          // 0 = mean, 1 = max
  const float* ln_w; const float* ln_b; float eps;
  int gelu;
  float* pooled;                     // [B, d] fp32, saved for backward
  int* arg;                          // [B, d] arg-max row (type 1 only), saved for backward
  float* feat;                       // forward out [B, d] fp32
  const float* dfeat;                // backward in  [B, d] fp32
  void* dx; int dx_is_f32;           // backward out [B, L, d]
  float* dln_w; float* dln_b;        // backward out [d] (overwritten), when ln_w != null
  float* scratch;                    // backward work [2, B, d] fp32, when ln_w != null
} TfLmPoolArgs;

// tensor-in narration pooling layer, everything behind its out_mlp GEMM in one kernel each way (SlowFastPooling.forward,
// modeling/narration_embeds/datasets/slowfast_features_dsets.py:229-235; the same tail as SBertLayer's, narr_pooling_layers.py:193-199):
//   u = tanh(x) (iff use_tanh) with the rows t >= lens[b] zeroed (ragged batches: right padding),
//   n[b, c] = max(||u[b, :, c]||_2, 1e-12), z = u / n when T > 1 (F.normalize(p=2, dim=1): over the TOKEN axis), else z = u,
//   y = dropout(z) (keep(i) of the element index i = (b * T + t) * d + c, as everywhere in this library).
// backward: gz = dropout'(gy); gu = (gz - z * sum_t(gz * z)) / n (T > 1); gx = gu * (1 - u^2) (tanh) on the kept rows, 0 on the padded ones.
typedef struct TfPoolNormArgs {
  const void* x; int x_is_f32; int ldx;  // [B, T, ldx] bf16 or fp32 (the out_mlp GEMM's output, or the raw embeddings)
  const int* lens;                       // [B] int32 valid tokens per sample, or null: all T
  int B, T, d;
  int use_tanh;
  unsigned drop_thr, drop_key; float drop_scale;
  float* y;                              // forward out [B, T, d] fp32
  float* z; float* n;                    // saved for backward: z [B, T, d] fp32 (may alias y when drop_thr == 0), n [B, d] fp32
  const float* gy;                       // backward in  [B, T, d] fp32
  void* gx; int gx_is_f32; int ldgx;     // backward out [B, T, ldgx] (pad columns zeroed when bf16)
} TfPoolNormArgs;

// RoI heads' losses.  The four Linears (faster_rcnn_wrapper.py:93-100: box_regressor, noun_classifier,
// verb_classifier; roi_wrappers.py:306: ttc_pred_layer) are tf_gemm_fwd calls: `box` = box_regression [R, 4*Cn] and `cls` =
// noun | verb | ttc pre-activation concatenated [R, Cn + Cv + 1 (+ pad)], both bf16 (+ lo plane in the fp32-accuracy mode).
// One wave per RoI then computes, from one read of its logits (reference lines in csrc/heads.hip):
//   box  = sum over positive RoIs (noun label > 0) of smooth_l1(box[r, 4y..4y+3] - reg_targets[r], beta 1/9) / max(R, 1)
//   noun = class-weighted mean CE of (logits + 1e-6);  verb = the same with background RoIs (label == verb_ignore) mapped to the last
//          class (verb_bg) or dropped;  ttc = mean smooth_l1(ttcs - target, ttc_beta) over non-background RoIs (or, ttc_bg, all RoIs
//          with background targets replaced by ttc_bg_val);  ttcs = softplus(cls[:, Cn + Cv]) (tf_softplus_col).
typedef struct TfHeadsLossArgs {
  const void* cls; const void* cls_lo; int ld_cls;
  const void* box; const void* box_lo; int ld_box;      // box == null: no box loss
  const float* ttcs;                                     // [R] fp32 softplus outputs, or null: no TTC loss
  int R, Cn, Cv;
  const long long* noun_labels;                          // [R] int64, 0 = background
  const long long* verb_labels;                          // [R] int64 (verb_ignore = background), or null
  const float* ttc_targets;                              // [R] fp32
  const float* reg_targets;                              // [R, 4] fp32
  const float* noun_w; const float* verb_w;              // class weights [Cn] / [Cv]; null switches that head's loss off
  // Labels are range-checked on the device: a noun label outside [0, Cn) or a verb label that is neither verb_ignore nor in [0, Cv)
  // contributes nothing (no memory is read through it); unless it equals -100 (torch's ignore_index) it is counted in sums[7].
  long long verb_ignore; int verb_bg; int ttc_bg; float ttc_bg_val; float ttc_beta; float box_beta;
  float* sums;                                           // [8] fp32, zeroed by the caller: numerators / normalisers, kept for the backward;
                                                         // [7] = number of out-of-range labels seen (error flag for the host, read lazily)
  float* lse;                                            // [2, R] fp32 work: log-sum-exp of the noun / verb rows, kept for the backward
  float* losses;                                         // forward out [4]: box, noun, verb, ttc
  const float* gscale;                                   // backward in [4]: d(total) / d(box, noun, verb, ttc loss)
  void* d_cls; void* d_cls_lo;                           // backward out, layout of cls (every column written; ttc column and pad zero)
  void* d_box; void* d_box_lo;                           // backward out, layout of box (every column written)
  float* d_ttcs;                                         // backward out [R]: d(total) / d(ttcs)
} TfHeadsLossArgs;

// patch <-> token permutations for K1 / K9
typedef struct TfPatchArgs {
  const void* feat; int feat_is_f32;     // [B,C,H,W]
  void* cols; int ld_cols;               // [B*Hp*Wp, C*ph*pw (padded to ld_cols)] bf16
  int B, C, H, W, ph, pw;
  // fp32-accuracy mode (ABI v7): cols as a hi + lo bf16 plane pair (value = hi + lo, 16 significant bits; same ld).  The gather
  // (tf_patchify_fwd / tf_regroup_bwd) splits an fp32 feature map into the two planes, the scatter (tf_regroup_fwd / tf_patchify_bwd)
  // adds them back: K1 / K9 at run.precision 32 without an fp32 im2col matrix in between
  void* cols_lo;
} TfPatchArgs;

/* fp32 [rows, cols] -> the operand planes of the fp32-accuracy mode: hi = bf16(v), lo = bf16(v - hi), both [rows, ld_dst] with the
 * columns [cols, ld_dst) zero-filled (ld_dst % 8 == 0; any cols, any row stride: 16-B accesses where the rows allow them) -- what torch
 * spelt as three elementwise passes and two pads per GEMM operand.  Optional input
 * dropout first (drop_thr != 0: element (r, c) is kept iff tf_dropout_mask says so at index r * drop_ld + c, kept values scaled by
 * drop_scale -- BEFORE the split, so that the pair still carries 16 bits of the scaled value).  dst_f32 != null: the (dropped) fp32
 * values are also / only written there (ld_f32; may alias src: the backward's in-place mask); hi / lo may then be null. */
typedef struct TfPlanesArgs {
  const float* src; int ld_src;
  void* hi; void* lo; int ld_dst;
  float* dst_f32; int ld_f32;
  int rows, cols;
  unsigned drop_thr, drop_key; float drop_scale; int drop_ld;
} TfPlanesArgs;
int tf_split_planes(const TfPlanesArgs* a, tf_stream_t s);


/* ---- dropout key: every dropout site draws keep(i) = hash32(i, key) >= p * 2^32, key = f(seed, site) ---- */
uint32_t tf_drop_key(uint64_t seed, uint32_t site);
uint32_t tf_drop_threshold(float p);   /* 16-bit threshold: round(p * 65536) */
float tf_drop_scale(float p);          /* 1 / (1 - threshold/65536): the exact inverse keep probability */
/* bytes of the attention dropout bitmask for (B, H, S) */
size_t tf_attn_dropmask_bytes(int B, int H, int S);
int tf_attn_dropmask(void* bits, int B, int H, int S, uint32_t key, uint32_t thr, tf_stream_t s);
/* cross attention (TfAttnArgs.q): nrows = B * H * Sq query rows of S key bits, [nrows, ceil(S/64)] u64 */
int tf_attn_dropmask_rows(void* bits, long long nrows, int S, uint32_t key, uint32_t thr, tf_stream_t s);
/* bytes of TfAttnArgs.ds_work for (B, H, S) */
size_t tf_attn_ds_bytes(int B, int H, int S);

/* ---- library ---- */
int tf_version(void);
const char* tf_last_error(void);

/* ---- per-op entries (thin validated wrappers over the kernels) ---------------------------------------------
 * tf_gemm_fwd        nn.Linear / F.linear call sites: torch18_adapters.py:683-685 (QKV in-proj), :608 (out-proj),
 *                    :111 (linear1, gelu, dropout, linear2) with the residual/dropout of :109,:112 fused as epilogues;
 *                    also the dgrad GEMMs of their autograd backward (W^T shadows) and K1/K9's GEMMs
 *                    (cross_f_box_wrapper.py:268-274 Conv2d k=s=p; cross_fusion/utils.py:116 RegroupPatchesLayerBox.linear)
 * tf_gemm_wgrad      autograd weight/bias gradients of the same Linear layers
 * tf_attn_fwd/bwd    torch18_adapters.py:756-799 (_scaled_dot_product_attention) + :578-597 (key-padding mask merge)
 * tf_layernorm_*     norm1/norm2 (torch18_adapters.py:110,113) and final_norm_layer (cross_f_box_layers.py:59-60,107)
 * tf_assemble_*      cross_f_box_layers.py:72-86 (pos-emb add utils.py:209-214, kind-emb adds, patch dropout, concat)
 * tf_patchify_*      cross_fusion/utils.py:35-39 (patchify_image) fused with the im2col of the k=s=p Conv2d
 * tf_regroup_*       cross_fusion/utils.py:42-46 (regroup_patches: transpose + F.fold, kernel == stride)
 * tf_radam_step      runner/metrics_losses/radam_optim.py:30-104
 * tf_lm_pool_*       ego_fusion/lm_layers.py:59-72 (PoolPredictor: masked mean/max pooling, LayerNorm, GELU)
 * tf_heads_loss_*    runner/metrics_losses/losses.py:98-135 (box_loss), runner/nao/ego_nao_trainer.py:307-359 (noun / verb CE, TTC)
 * tf_softplus_col    modeling/obj_detection/roi_wrappers.py:228-229 (ttcs = softplus(ttc_pred_layer(box_features)))
 */
/* planning hint, process-wide: n independent launch sequences (the wrapper's feature levels on their own streams) share the chip, so a
 * GEMM plans its tile grid for 1 / n of the CUs.  1 (default): a launch has the chip to itself.  Affects speed only, never results. */
void tf_set_gemm_concurrency(int n);
int tf_gemm_fwd(const TfGemmArgs* a, tf_stream_t s);
int tf_gemm_wgrad(const TfWgradArgs* a, tf_stream_t s);
/* probs[0 .. count): see TF_WGRAD_MULTI_MAX.  blocks > 0: that many workgroups (256 x 128 output tiles x row chunks) in flight;
 * 0 = sized by the library for a launch that has the chip to itself; -1 = sized for a launch that runs beside other kernels (0 / -1:
 * where the products make one well-filled round of 192 x 192 tiles, the two-quad form with its in-workgroup reduction is taken);
 * -2 = that form wherever the row count allows it (tests) */
int tf_gemm_wgrad_multi(const TfWgradArgs* probs, int count, int blocks, tf_stream_t s);
int tf_attn_fwd(const TfAttnArgs* a, tf_stream_t s);
int tf_attn_bwd(const TfAttnArgs* a, tf_stream_t s);       /* dQ (which also fills `delta`), then dK, dV */
int tf_layernorm_fwd(const TfLnArgs* a, tf_stream_t s);
int tf_layernorm_bwd(const TfLnArgs* a, tf_stream_t s);
int tf_assemble_fwd(const TfAssembleArgs* a, tf_stream_t s);
int tf_assemble_bwd(const TfAssembleArgs* a, tf_stream_t s);
int tf_patchify_fwd(const TfPatchArgs* a, tf_stream_t s);                 /* feat -> token-major im2col rows */
int tf_patchify_bwd(const TfPatchArgs* a, int out_is_f32, tf_stream_t s); /* d(cols) -> d(feat) */
int tf_regroup_fwd(const TfPatchArgs* a, int out_is_f32, tf_stream_t s);  /* token rows -> [B,C,H,W], zero border */
int tf_regroup_bwd(const TfPatchArgs* a, tf_stream_t s);                  /* d(feat) -> d(token rows) */
int tf_pack_weight(const TfPackArgs* a, tf_stream_t s);
int tf_copy_rows(const TfCopyRowsArgs* a, tf_stream_t s);
int tf_radam_step(const TfRadamArgs* a, tf_stream_t s);
/* The step clock (no reference counterpart; what replaces torch's advancing Philox offset when a whole training step is captured in a
 * HIP graph).  Every kernel that draws a dropout mask folds mix(*clock) into the key it was launched with; the word is 0 until the
 * caller advances it and mix(0) = 0, so callers that never touch it see the keys they pass.  A captured step puts tf_clock_advance
 * first: each replay then draws fresh masks (forward and backward of one replay agree), and tf_radam_step with step_clock = the same
 * word counts its steps.  tf_clock_ptr: the device word (allocated on first use, one per process); tf_clock_advance: *clock += by on
 * the stream; tf_clock_set: *clock = value (tests). */
const uint32_t* tf_clock_ptr(void);
int tf_clock_advance(uint32_t by, tf_stream_t s);
int tf_clock_set(uint32_t value, tf_stream_t s);
int tf_sumsq(const float* x, long long n, float* out, tf_stream_t s);
/* ABI v10.  out[0] = sum(x^2) -- tf_sumsq without the accumulation: the global gradient norm of a step over ONE flat gradient buffer is
 * then a single launch (no zero fill in front, no launch per parameter range).  Same deterministic block sum. */
int tf_sumsq_set(const float* x, long long n, float* out, tf_stream_t s);
/* ABI v9.  One term of the synthetic training loss of the benchmark (mean(vis^2) + mean(lang[valid]^2); the reference
 * has no counterpart -- its losses sit behind the detector, losses.py:98-135) and its gradient, so that a timed step holds no
 * framework elementwise kernel.  x: contiguous fp32 [rows, d], d % 4 == 0, 16-B aligned; row_w: optional [rows] weights (the 0 / 1
 * valid-token mask).  fwd: out[0] = (accumulate ? out[0] : 0) + scale * sum_r row_w[r]^2 |x[r, :]|^2 -- block partials summed in index
 * order by the last block to arrive, like tf_sumsq: the same bits on every run and rank.  bwd: dx[r, :] = g[0] * 2 * scale *
 * row_w[r]^2 * x[r, :] (g: optional device scalar, the upstream gradient; null = 1). */
typedef struct TfSqLossArgs {
  const float* x; long long rows; int d;
  const float* row_w;
  float scale;
  float* out; int accumulate;         /* forward */
  const float* g; float* dx;          /* backward */
} TfSqLossArgs;
int tf_sq_loss_fwd(const TfSqLossArgs* a, tf_stream_t s);
int tf_sq_loss_bwd(const TfSqLossArgs* a, tf_stream_t s);
int tf_heads_loss_fwd(const TfHeadsLossArgs* a, tf_stream_t s);
int tf_heads_loss_bwd(const TfHeadsLossArgs* a, tf_stream_t s);
/* dy == null: y[r] = softplus(x[r, col]) (F.softplus defaults, roi_wrappers.py:229); dy != null: dx[r, col] += dy[r] * sigmoid(x[r, col]).
 * x / dx: bf16 [R, ld] (+ lo planes or null) */
int tf_softplus_col(const void* x, const void* x_lo, int ld, int col, float* y, const float* dy, void* dx, void* dx_lo, int R, tf_stream_t s);
int tf_pool_norm_fwd(const TfPoolNormArgs* a, tf_stream_t s);
int tf_pool_norm_bwd(const TfPoolNormArgs* a, tf_stream_t s);
int tf_lm_pool_fwd(const TfLmPoolArgs* a, tf_stream_t s);
int tf_lm_pool_bwd(const TfLmPoolArgs* a, tf_stream_t s);
/* y = dropout(x) over a dense bf16 array of n (multiple of 8) elements; the same call is its backward (utils.py:115) */
int tf_dropout_apply(const void* x, void* y, long long n, uint32_t key, uint32_t thr, float scale, tf_stream_t s);
int tf_dropout_mask(uint8_t* out, long long n, uint32_t key, uint32_t thr, tf_stream_t s);   /* test hook */
int tf_cast_f32_bf16(const float* src, void* dst, long long n, tf_stream_t s);
int tf_cast_bf16_f32(const void* src, float* dst, long long n, tf_stream_t s);

/* ---- whole-encoder runtime: one call = forward (or backward) of CrossTransformerModuleBox ------------------
 * cross_f_box_layers.py:69-108.  The caller plans once (tf_encoder_plan), allocates `wpack` and `work` (both
 * zero-initialised), and then calls fwd / bwd; parameter pointers are the reference's state_dict tensors
 *  in fp32. */
typedef struct TfLayerParams {
  float *in_w, *in_b, *out_w, *out_b, *w1, *b1, *w2, *b2, *n1_w, *n1_b, *n2_w, *n2_b;
} TfLayerParams;

typedef struct TfEncoderPlan {
  int hd, hdp, dp, ffp, ldq, S, M;
  size_t wpack_bytes, work_bytes;
} TfEncoderPlan;

/* Optional side stream for the weight-gradient GEMMs of tf_encoder_bwd.  The backward chain (LN bwd -> dgrad -> attention
 * bwd -> dgrad ...) is strictly serial and every kernel in it has a fill and a drain phase in which CUs idle; the four
 * wgrad GEMMs of a layer depend only on tensors the chain has already produced, so with a TfOverlap they are issued on
 * `stream` (forked from / joined back into the caller's stream with the events) and fill those bubbles.  The caller
 * still sees single-stream semantics: tf_encoder_bwd joins before it returns.  Create once per device, reuse for every call
 * (calls that share one TfOverlap must be issued from one thread).  No reference counterpart (autograd engine detail). */
typedef struct TfOverlap {
  void* stream;                 /* hipStream_t, non-blocking */
  void* ev[8];                  /* hipEvent_t: [0..3] fork (chain -> side), [4..7] done (side -> chain); see tf_api.hip */
  unsigned pending, reserved;   /* owned by the library: done events (by layer parity) of tf_encoder_bwd(defer_join) calls not yet joined,
                                 * and the parity of the most recent one */
} TfOverlap;
int tf_overlap_create(TfOverlap* o);
int tf_overlap_destroy(TfOverlap* o);
/* makes stream s wait for everything the side stream has been given so far (after tf_encoder_bwd calls with defer_join) */
int tf_overlap_join(TfOverlap* o, tf_stream_t s);

typedef struct TfEncoderDesc {
  int B, Nv, Nl, d, H, L, ff;
  int training;                 /* dropout on (p_token, p_patch) */
  int final_norm;               /* final_norm == "ln" */
  float p_token, p_patch;
  uint64_t seed;                /* dropout stream for THIS forward; backward must pass the same value */
  TfLayerParams p[TF_MAX_LAYERS];   /* parameters */
  TfLayerParams g[TF_MAX_LAYERS];   /* gradients (fp32, accumulated into) */
  float *kind_v, *kind_l, *fn_w, *fn_b;
  float *g_kind_v, *g_kind_l, *g_fn_w, *g_fn_b;
  const float* pe;              /* pos_embedding [>=Nv, d] fp32 */
  void* wpack; void* work;
  const void* vis; int vis_is_f32;          /* [B,Nv,d] */
  const void* lang; int lang_is_f32;        /* [B,Nl,d] */
  const uint8_t* lang_pad_mask;             /* [B,Nl] 1 = ignore, or null */
  const void* attn_block_bits;              /* TfAttnArgs.block_bits for every layer (vis_mask_type "local_k"), or null */
  void* vis_out; int vis_out_is_f32;        /* [B,Nv,d] */
  void* lang_out; int lang_out_is_f32;      /* [B,Nl,d] or null */
  const void* d_vis_out; int d_vis_out_is_f32;
  const void* d_lang_out; int d_lang_out_is_f32;   /* may be null (zero) */
  void* d_vis; int d_vis_is_f32;            /* may be null */
  void* d_lang; int d_lang_is_f32;          /* may be null */
  /* partial backward (gradient all-reduce overlap): process bwd_nlayers layers starting at layer bwd_hi and going DOWN;
   * bwd_nlayers == 0 means the whole encoder.  The output-side part (final LayerNorm) runs with the chunk that contains
   * layer L-1, the input-side part (token assemble) with the chunk that contains layer 0; chunks must be issued in
   * descending order on one stream. */
  int bwd_hi, bwd_nlayers;
  TfOverlap* overlap;           /* null: everything on the caller's stream */
  int defer_join;               /* tf_encoder_bwd with an overlap handle: return WITHOUT joining the side stream; its pending wgrads
                                 * are recorded in the handle, guarded by the next tf_encoder_bwd call on the same handle and joined by
                                 * tf_overlap_join (per-layer calls of the data-parallel reducer: the collective of a layer waits for
                                 * the side stream's events on its own stream instead of stalling the backward chain) */
  int fp8_proj;                 /* forward QKV / FFN-up / FFN-down projections with fp8 (e4m3) operands and fp32 accumulation
                                 * (BASELINE configs[4]): activations quantised per token in front of each of the three GEMMs,
                                 * weights per output channel when the shadows are packed; backward unchanged (bf16) */
  int repack;                   /* tf_encoder_fwd only: refresh the bf16 weight shadows first (what tf_encoder_pack does); with an
                                 * overlap handle only layer 0 is packed on the caller's stream, layers >= 1 and the attention
                                 * dropout masks are produced on the side stream while the chain already runs layer 0 */
  int precision;                /* 0: bf16 compute (run.precision 16 / bf16).  1: fp32-accuracy mode for run.precision: 32
                                 * (ego_nao_res50_ego4dv2.yml:124, BASELINE configs[2]): activations and weight shadows are hi + lo
                                 * bf16 plane pairs and every contraction runs as three bf16 MFMA passes with fp32 accumulation
                                 * (results within 1e-3 of the fp32 reference; ~1e-5 measured).  wpack / work sizes: tf_encoder_plan_ex */
  int act;                      /* FFN activation: 0 = GELU (activ_f: "gelu"), 1 = ReLU (the reference constructor's default) */
  const float* pe_lang;         /* lang_pos_embedding table [>=Nl, d] fp32 or null (cross_f_box_layers.py:77-78) */
  int packed_rows;              /* > 0: run on the PACKED token rows only.  The value is the number of rows that take part: B * Nv plus the
                                 * number of language tokens lang_pad_mask leaves un-masked, counted by the host (the tokeniser / the
                                 * length list that built the mask knows it; no device read-back).  Masked language tokens are then never
                                 * gathered: they are attended by nobody (key padding) and, in every use the reference makes of the
                                 * fused language tokens (lm_layers.py:59-61 multiplies them by the mask), their outputs are discarded --
                                 * so this mode writes ZEROS to their lang_out rows, ignores their d_lang_out rows and returns zero d_lang
                                 * rows for them; every other output and every parameter gradient is what the dense mode computes.
                                 * All row-wise work (GEMMs, LayerNorm, attention queries, weight gradients) shrinks to the real tokens.
                                 * 0: dense (bit-for-bit the reference's semantics, padded rows included).  A count that disagrees with
                                 * the mask is reported through tf_encoder_packed_error. */
  int groups;                   /* > 1: the batch holds the samples of `groups` INDEPENDENT encoders of identical shape, B / groups samples
                                 * each (group g = samples [g B / groups, (g+1) B / groups)) -- the wrapper's FPN levels
                                 * (cross_f_box_wrapper.py:177-212), which run the same [Nv + Nl]-token encoder with four weight sets, as ONE
                                 * launch sequence: a quarter of the launches, grids four times the size.  Every parameter and gradient
                                 * pointer of this struct (p[], g[], kind_*, fn_*, g_*) names group 0's tensor; group g's tensor starts
                                 * g * param_gstride BYTES further (true for every tensor alike: the parameters of the encoders sit at one
                                 * common stride, as in FusionTrainStep's flat buffer).  wpack holds `groups` shadow blocks back to back
                                 * (tf_encoder_plan_ex sizes it).  pe / pe_lang are shared by the groups; attn_block_bits must be null;
                                 * with packed_rows every group must drop the same tokens (same lang_pad_mask rows).  0 / 1: one encoder. */
  long long param_gstride;
  int group_nv[TF_MAX_GROUPS];  /* RAGGED groups (ABI v8): group_nv[0] > 0 -> the samples of group g carry group_nv[g] visual tokens (Nv is the
                                 * LARGEST of them and shapes the dense index space b * (Nv + Nl) + s, the attention grid, the dropout
                                 * bitmasks).  The reference's real FPN geometry: 28 x 28 tokens on level 0, 14 x 14 on levels 1 - 3
                                 * (cross_fusion_config_sym_ego_res50.yml:8-17) -- with this all four levels are ONE launch sequence.
                                 * Needs packed_rows > 0 = sum_g (B / groups) group_nv[g] + the un-masked language tokens of all groups
                                 * (every group drops the same tokens).  vis / vis_out / d_vis_out / d_vis are then the CONCATENATION
                                 * [sum_g (B / groups) group_nv[g], d], group-major; pe is read by its first group_nv[g] rows (the tables
                                 * of the levels must agree on their common prefix: the sin1d tables do).  All zero: every group has Nv. */
  int* packed_error_host;       /* ABI v10, optional: a device-accessible word of PINNED HOST memory (hipHostMalloc / torch pin_memory).  The
                                 * row-map kernel of a packed forward stores this forward's verdict there too (0, or the row count the mask
                                 * gave): the host presets the word to -1 and polls it later -- no device-to-host copy and no event on the
                                 * forward's stream (tf_encoder_packed_error costs the chain a 4-us copy and a barrier packet per step). */
} TfEncoderDesc;

int tf_encoder_plan(int B, int Nv, int Nl, int d, int H, int L, int ff, TfEncoderPlan* out);   /* precision 0 */
int tf_encoder_plan_ex(const TfEncoderDesc* e, TfEncoderPlan* out);   /* reads B, Nv, Nl, d, H, L, ff and precision of *e */
int tf_encoder_pack(const TfEncoderDesc* e, tf_stream_t s);   /* fp32 parameters -> bf16 shadows in wpack */
int tf_encoder_fwd(const TfEncoderDesc* e, tf_stream_t s);
int tf_encoder_bwd(const TfEncoderDesc* e, tf_stream_t s);
/* packed mode self-check: copies the device-side count mismatch word of the workspace of *e (0 = the host's packed_rows agreed with
 * the mask in every forward run on this workspace so far; otherwise the row count the mask gave) into host memory *out.  Enqueues a
 * 4-byte device-to-host copy on s; the caller synchronises s before reading (tests, debugging; not needed per step). */
int tf_encoder_packed_error(const TfEncoderDesc* e, int* out, tf_stream_t s);
/* test hook: copies an internal activation by name ("x<l>", "qkv<l>", "o<l>", "z1_<l>", "x1_<l>", "u<l>" (holds G = d h / d u), "h<l>",
 * "z2_<l>", "dqkv", ...) as fp32 [rows, cols] into dst; returns rows*cols (cols = padded width) or < 0 */
long long tf_encoder_peek(const TfEncoderDesc* e, const char* name, float* dst, long long cap, tf_stream_t s);

/* ---- data-parallel gradient exchange (SURVEY.md 8(b) `tf_allreduce_bucket`, 8(e)) ---------------------------------------
 * What the reference gets from Lightning's strategy="ddp" (runner/run_experiment.py:452: torch DDP's bucketed gradient
 * all-reduce over NCCL): one RCCL communicator per process (one process per GPU, created on the calling thread's current
 * HIP device), and an in-place SUM all-reduce of a contiguous fp32 slice of the flat gradient buffer, enqueued on the
 * stream the caller names (the host side puts it behind events of the backward's two streams, so a layer's exchange
 * overlaps the layers below it; the 1 / world average is folded into tf_radam_step's grad_scale).
 * RCCL is bound at run time (dlopen of librccl.so.1): without it every entry returns TF_ERR_NO_RCCL.
 * Bootstrap: rank 0 calls tf_comm_unique_id, the host broadcasts the 128 bytes (torch.distributed store, MPI, a file ...),
 * every rank calls tf_comm_create (collective: returns when all `world` ranks have joined). */
#define TF_COMM_ID_BYTES 128
#define TF_ERR_NO_RCCL (-100)   /* librccl.so.1 not loadable in this process */
#define TF_ERR_RCCL (-101)      /* an RCCL call failed; tf_last_error() carries ncclGetErrorString */
typedef struct TfComm TfComm;
int tf_comm_unique_id(void* id /* TF_COMM_ID_BYTES, host memory */);
int tf_comm_create(TfComm** comm, const void* id, int world, int rank);
int tf_allreduce_bucket(TfComm* comm, float* buf, long long n, tf_stream_t s);    /* buf[0..n) <- sum over ranks, in place */
/* collectives issued / fp32 elements reduced so far on this rank (all ranks must agree); any out pointer may be null */
int tf_comm_stats(const TfComm* comm, int* world, int* rank, long long* calls, long long* elems);
int tf_comm_destroy(TfComm* comm);

/* ---- launch tracer (measurement only) ----
 * Between tf_trace_start and tf_trace_stop every kernel launch of this library is bracketed by a HIP event pair on
 * the stream it is launched on, so durations are the kernels' own, in situ (side-stream overlap included) -- the same
 * quantity rocprofv3 --kernel-trace reports.  bench.py uses it for the `roofline` object.  Process-wide, one tracer. */
typedef struct TfTraceRecord {
  char name[56];                /* kernel symbol as rocprofv3 prints it (short form) */
  float us;                     /* duration of this launch */
  float start_us;               /* start of this launch relative to the start of the first traced launch */
  int side;                     /* 1: launched on a TfOverlap side stream */
  double flops, bytes;          /* algorithmic work of this launch (SURVEY.md 8(d) accounting; 0 where not applicable) */
} TfTraceRecord;
int tf_trace_start(void);
long long tf_trace_stop(TfTraceRecord* out, long long cap);   /* device sync; returns the number of launches recorded (<= cap are written) */

/* ---- stream-ordering probes (tests; no reference counterpart) ----
 * The weight gradients of this library are produced on OTHER streams than the chain (TfOverlap side streams, the wrapper's level
 * streams) and consumed by collectives and the optimiser behind events.  A missing event edge shows as a wrong gradient once in many
 * runs; behind a spin it shows every time.  tf_debug_spin: a kernel that occupies stream s for `us` microseconds (one wave, capped at
 * 50 ms, always terminates).  tf_debug_delay_wgrad: from now on every weight-gradient launch tf_encoder_bwd puts on a TfOverlap side
 * stream is preceded by such a spin (process-wide; 0 = off, the default); returns the previous value. */
int tf_debug_spin(int us, tf_stream_t s);
int tf_debug_delay_wgrad(int us);

#ifdef __cplusplus
}
#endif
#endif  /* TFUSION_H_ */
