/*
 * lmx.h — C-ABI of liblmx.so, the MI355X (gfx950) feature-extraction kernels that sit behind the
 * yolo-pipeline / sam3-pipeline / dinov3-pipeline services of UBC-AWP/vision-sam3-yolo-lameless.
 *
 * The reference has no FFI of its own (it is pure Python: SURVEY.md §8b); the seam it offers is the three
 * third-party call sites
 *     services/yolo-pipeline/app/main.py:76      self.yolo_model(frame, verbose=False, conf=...)
 *     services/sam3-pipeline/app/main.py:80-88   predictor.set_image(image); predictor.predict(box=...)
 *     services/dinov3-pipeline/app/main.py:107-113  processor(images=...); model(**inputs).last_hidden_state.mean(1)
 * Every entry point below replaces a piece of the arithmetic that runs under one of those calls; the comment
 * on each one says which.  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no C++ or torch types; every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue work;
 *   - return 0 on success, <0 on error; lmx_last_error() returns the thread-local message;
 *   - activations are NHWC / token-major with the channel dimension contiguous, f16 unless said otherwise,
 *     accumulation is always f32 (MFMA f32 accumulators), the ViT residual stream is f32;
 *   - all "ld*" strides are in ELEMENTS of the tensor's dtype.
 */
#ifndef LMX_H
#define LMX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LMX_VERSION 100

/* error codes */
#define LMX_OK 0
#define LMX_EINVAL (-1)  /* bad argument / unsupported shape (message says which) */
#define LMX_EHIP (-2)    /* a HIP runtime call failed */

typedef void* lmx_stream_t;

int lmx_version(void);
const char* lmx_last_error(void);
/* number of visible HIP devices, <0 on error (used by the loader to fail loudly on a GPU-less box) */
int lmx_device_count(void);

/* ---- dtypes / activations -------------------------------------------------------------------------- */
enum { LMX_F16 = 0, LMX_F32 = 1 };
enum { LMX_ACT_NONE = 0, LMX_ACT_SILU = 1, LMX_ACT_GELU = 2 /* erf form */, LMX_ACT_RELU = 3 };

/* ---- K3/K12/K2: GEMM / 1x1 conv / 3x3 conv (implicit GEMM), f16 in, f32 MFMA accumulate ------------
 * C[m][n] = res[m][n] + scale[n] * act( sum_k A[m][k] * W[n][k] + bias[n] )
 * Replaces: torch Linear / Conv2d(+folded BN)+SiLU under ultralytics' C2f/Conv/Detect (yolo main.py:76),
 * the ViT qkv/proj/fc1/fc2 Linears under segment_anything's ImageEncoderViT (sam3 main.py:80) and under
 * transformers' Dinov2/DINOv3 layers (dinov3 main.py:110-111).
 *   a_mode 0: A is a row-major [M][K] f16 matrix with row stride lda.
 *   a_mode 1: A is generated on the fly from an NHWC f16 image batch X[n][H][W][*] (pixel stride lda,
 *             Cin channels used): kernel 3x3, pad 1, stride conv_stride; M = n*Ho*Wo, K = 9*Cin,
 *             k = (ky*3+kx)*Cin + ci  (weights must be packed in that order).
 *   a_mode 2: "pooled rows" — A is a row-major [M][K] matrix whose rows are an [n][H][W_] token grid (H, W_ even), and C is
 *             the 2 x 2 max-pool of the product over that grid: [M/4][N] (f32 or f16) in [n][H/2][W_/2] order, the bits of a_mode 0
 *             followed by lmx_k_maxpool2 without the full-size intermediate (Hiera's `do_pool(self.proj(x))` at the stage
 *             transitions, TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock, and of the pooled queries).  No
 *             residual / scale / activation; M >= 512, N >= 96, N%8==0 (the LDS-DMA kernel); A smaller than 2 GB.
 * W is f16 [N][K] (K contiguous).  Requirements: K%8==0, N%4==0, lda%8==0, Cin%8==0, 16-byte aligned bases.
 */
typedef struct {
  const void* A;
  const void* W;
  const float* bias;   /* [N] or NULL */
  const float* scale;  /* [N] or NULL (LayerScale) */
  const void* res;     /* [M][N] residual of dtype out_dtype, row stride ldr, or NULL; may alias C */
  void* C;
  int64_t lda, ldc, ldr;
  int32_t M, N, K;
  int32_t act;
  int32_t out_dtype;   /* LMX_F16 / LMX_F32 */
  int32_t a_mode;
  /* a_mode 1 (all six); a_mode 2 (H, W_: the token grid) */
  int32_t H, W_, Cin, conv_stride, Ho, Wo;
  /* residual broadcast: if res_rows > 0 the residual row is (m % res_rows) — e.g. a position table [tokens][N]
   * added to every image of the batch (Hiera patch embed + pos_embed, TF sam2 :661-662) */
  int32_t res_rows;
  /* a_mode 0 only: A's K columns are read a_rep times (0 / 1: once): K = a_rep * Ka, k-tile kt takes A columns (kt*BK) mod Ka
   * and W columns kt*BK.  With W = [whi | wlo] (a_rep 2) ONE launch accumulates a.whi + a.wlo: f16 activations against
   * 22-bit weights (the exact plan of the SAM ViT encoder, lmx/sam.py).  Ka % 64 == 0; the LDS-DMA kernel's shapes only. */
  int32_t a_rep;
  /* split-K for long-K, few-tile problems (the exact plan's 3 x 3 convolutions at 10 frames: K = 27 Cin against a few dozen
   * tiles on 256 CUs): split_k = S >= 2 cuts the k range into S parts, part s writes ITS partial sum to C + s * split_stride
   * (elements).  f32 output, no activation, a_mode 0 or 1 on the LDS-DMA kernel's shapes only; bias and residual enter partial 0,
   * `scale` applies to every partial (the epilogue is linear), and the consumer adds the S partials in a fixed order
   * (lmx_k_split3's `nsum`): deterministic, unlike atomics.  0 / 1: no split. */
  int32_t split_k;
  int64_t split_stride;
} lmx_gemm_desc;
int lmx_k_gemm(const lmx_gemm_desc* d, lmx_stream_t stream);
/* development hook (tools/gemm_sweep.py): force one tiling of the LDS-DMA GEMM for every following launch; v = 0 restores
 * the launcher's per-shape choice, otherwise one of the letters documented at lmx_gemm2_launch (csrc/gemm2.hip) */
void lmx_dbg_set_gemm2_variant(int v);

/* ---- K11: LayerNorm over the last dim, f32 or f16 in -> f16 or f32 out ------------------------------
 * Replaces torch.nn.LayerNorm inside the ViT blocks (TF:models/dinov3_vit/modeling_dinov3_vit.py:400-445,
 * TF:models/sam/modeling_sam.py:891-972).  One 64-lane wave per row, wave-shuffle reductions, two-pass
 * variance (mean first) in f32.  D%4==0, D<=4096.
 */
int lmx_k_layernorm(const void* x, int in_dtype, int64_t ldx, const float* gamma, const float* beta,
                    void* y, int out_dtype, int64_t ldy, int rows, int D, float eps, int act, lmx_stream_t stream);
/* (act = LMX_ACT_NONE or LMX_ACT_GELU applied after the affine: the SAM decoder's LayerNorm2d -> GELU, TF sam :523) */

/* ---- K11 + K8-K10 + K14 for Hiera's 8 x 8-token windows: x += proj(window_attention(qkv(layer_norm1(x)))) as one kernel ---------
 * Replaces, for the blocks of Hiera-B+ stage 1 (D = 112, 2 heads; TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock.forward,
 * Sam2MultiScaleAttention :350-409 with window_partition / window_unpartition :412-455), the four launches LayerNorm -> qkv GEMM ->
 * window attention -> projection GEMM (+ residual) and their f16 intermediates: per token it reads the f32 stream (4D bytes) and
 * writes it (4D).  x f32 [rows, ldx] updated in place; rows = n_img * Gh * Gw, Gh and Gw multiples of 8 (no padded windows).
 * h = NULL: the kernel normalises x itself (gamma, beta f32 [D], eps: layer_norm1); h = f16 [rows, D] contiguous: layer_norm1(x) as
 * written by the previous block's lmx_k_ln_mlp (h_next), read instead (+ 2D bytes per token; wqkv_p's columns then in natural order:
 * cheaper when the rows exist anyway, because the kernel's first products then do not wait for the f32 rows).  Packed operands (lmx/sam.py pack_hiera_attn): wqkv_p f16 [3*heads*64, 128] — sections
 * q | k | v, each head padded from 56 to 64 rows; v's row 63 of every head is zero with bias 1 (the softmax sum rides the PV
 * product); columns in MFMA k-slot order (position 32s + 8g + 4h + i holds input feature 16(2s+h) + 4g + i, zeros past D) —
 * bqkv_p f32 [3*heads*64], wo_p f16 [D, heads*64] with the 64 columns of a head in the same order, bo f32 [D].
 * scale = head_dim ** -0.5.  Rounding points as in the unfused chain: LayerNorm output, q, k, v, P, attention output f16; sums f32. */
int lmx_k_hiera_attn8(const void* h, float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* wqkv_p,
                      const float* bqkv_p, const void* wo_p, const float* bo, int n_img, int Gh, int Gw, int D, int heads, float scale,
                      lmx_stream_t stream);

/* The same for Hiera-B+ stage 2 (D = 224, 4 heads of 56, 4 x 4-token windows; blocks whose input and output widths are equal): the
 * three launches qkv GEMM -> window attention -> projection GEMM (+ residual) as one kernel whose weights stream through an LDS ring
 * (csrc/hiera.hip).  h f16 [rows, D] contiguous = layer_norm1(x) (the previous block's lmx_k_ln_mlp leaves it as h_next); x f32
 * [rows, ldx] updated in place; rows = n_img * Gh * Gw, Gh and Gw multiples of 4.  w_img f16 [16][16384]: per head h the LDS
 * images of its q, k, v (64 rows x 512 B, 16-byte chunk c of row r at c ^ (r & 15), rows >= 56 zero, v's row 63 zero with bias 1)
 * and projection (256 rows x 128 B, chunk c of row r at c ^ ((r >> 1) & 7), the head's 64 input columns in MFMA k-slot order)
 * matrices at index 4 h + {0, 1, 2, 3}; bias f32 [4][q | k | v][64] then the projection's [224] (lmx/sam.py pack_hiera_attn4). */
int lmx_k_hiera_attn4(const void* h, float* x, int64_t ldx, const void* w_img, const float* bias, int n_img, int Gh, int Gw, int D,
                      int heads, float scale, lmx_stream_t stream);

/* The blocks that open Hiera-B+ stage 2 (112 -> 224 channels, 4 heads of 56; keys / values the 64 tokens of an 8 x 8 window) and stage 3
 * (224 -> 448, 8 heads; 16 tokens of a 4 x 4 window), queries and shortcut the 2 x 2 max-pools of the window's tokens
 * (TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock.forward with dim != dim_out and q_stride): shortcut GEMM + pool, q GEMM + pool,
 * k | v GEMM, window attention with pooled queries, projection GEMM + residual as one kernel (csrc/hiera.hip; weights streamed).
 * h f16 [n_img*Gh*Gw, Din] contiguous = layer_norm1(x); out f32 [n_img*(Gh/2)*(Gw/2), Dout] contiguous (written, not accumulated);
 * Gh, Gw multiples of the window side.  w_img f16 [14 | 47][16384]: LDS images in streaming order — 112 -> 224: shortcut rows
 * 0..127, shortcut rows 128..223, then per head [q | k] (2 x 64 rows of 256 B), [v | unused], projection columns of the head (256
 * rows of 128 B, k-slot order); 224 -> 448: shortcut in 7 images of 64 rows (512 B rows), then per head q, k, v (64 rows each) and
 * the head's projection columns for output rows 0..223 and 224..447 —, bias f32 [shortcut + projection: Dout | heads x (q | k | v)
 * x 64 (| Dout zeros for 112 -> 224)] (lmx/sam.py pack_hiera_attn_pool). */
int lmx_k_hiera_attn_pool(const void* h, float* out, const void* w_img, const float* bias, int n_img, int Gh, int Gw, int Din, int Dout,
                          int heads, float scale, lmx_stream_t stream);

/* ---- K11+K12 for narrow widths: x += fc2(gelu(fc1(LayerNorm(x)))) without the 4D-wide hidden tensor ever reaching HBM ----
 * Replaces `hidden_states + self.mlp(self.layer_norm2(hidden_states))` of the Hiera blocks whose width is 112 or 224
 * (TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock.forward; stages 1-2 of Hiera-B+), where the unfused
 * LN -> GEMM -> GEMM chain is bound by its HBM intermediates (32*D bytes/token against 16*D here).
 * Two launches: the LayerNorm kernel (f32 stream -> f16 rows in `workspace`, rows*D*2 bytes), then the fused MLP + residual.
 * x f32 [rows, ldx] updated in place; w1 f16 [4D, D], b1 f32 [4D], w2 f16 [D, 4D], b2 f32 [D] (torch Linear layouts).
 * Rounding points match the unfused kernels: LN output and GELU output are rounded to f16, accumulation is f32.
 */
int lmx_k_ln_mlp(float* x, int64_t ldx, const float* gamma, const float* beta, const void* w1, const float* b1,
                 const void* w2, const float* b2, int64_t rows, int D, float eps, void* workspace, void* x16,
                 const float* gamma_next, const float* beta_next, void* h_next, lmx_stream_t stream);
/* The same operation with the weights as host-made LDS images streamed in steps of 64 hidden units by a workgroup of 8 waves x 32
 * tokens (csrc/hiera.hip hiera_mlp_kernel: half the L2 -> LDS traffic and a quarter to a half of the barriers of the kernel above;
 * layer_norm2 always inside).  w_img f16 [7 | 28][16384] for D = 112 | 224: per step the 64 rows of W1 (256 | 512-byte rows, the D
 * input columns in MFMA k-slot order, chunk c of row r at c ^ (r & 15)) and the D rows x 64 columns of W2 (128-byte rows, columns
 * in k-slot order, chunk c of row r at c ^ ((r >> 1) & 7); D = 112: in the same image at byte 16384, D = 224: the next image);
 * bias f32 [b1 (4D) | b2 | gamma2 | beta2 | gamma_next | beta_next (D each; the last two unused without h_next)] (lmx/sam.py
 * pack_ln_mlp).  x16, h_next: as below. */
int lmx_k_ln_mlp_img(float* x, int64_t ldx, const void* w_img, const float* bias, int64_t rows, int D, float eps, void* x16, void* h_next,
                     lmx_stream_t stream);
/* x16: NULL, or f16 [rows, D] (contiguous) that receives a copy of the updated x — the input of the FPN's lateral 1x1
 * convolution after a stage's last block, which otherwise costs a cast pass over the f32 stream.
 * h_next: NULL, or f16 [rows, D] that receives LayerNorm(updated x; gamma_next, beta_next, eps) — the NEXT block's
 * `layer_norm1(hidden_states)`, computed on the rows while they are still in registers instead of by a LayerNorm launch
 * that reads the f32 stream again */

/* ---- K13/K14: attention (flash-style, online softmax in f32, S and PV on MFMA) ----------------------
 * O[b,t,h,:] = softmax_j( scale * Q[b,t,h,:] . K[b,j,h,:] ) V[b,j,h,:]
 * Q/K/V/O are f16 with the head dim contiguous; element (row r, head h, d) sits at base + r*ld + h*hd + d
 * where r is the token row.  geometry:
 *   mode 0 (flat):    row = b*T + t                     (Tq queries, Tk keys per batch element)
 *   mode 1 (window):  K/V tokens live on a [Gh][Gw] grid per image, partitioned into ws x ws windows
 *                     (grid zero-padded up to a multiple of ws, as ImageEncoderViT.window_partition does);
 *                     b = (img, wy, wx); key t -> (wy*ws + t/ws, wx*ws + t%ws).  Keys that fall in the
 *                     padding take K = pad_k[h], V = pad_v[h] (= the qkv bias: what Linear(0) yields).
 *                     Queries use the same map on a grid subsampled by q_stride (1, or 2 for Hiera Q-pool):
 *                     grid [Gh/q][Gw/q]... see DESIGN.md §attention.
 * hd % 8 == 0, hd <= 96 (head dims up to 64 use 64-half LDS rows; 72..96 — SAM ViT-H's 80 — the 128-half class).
 * Three kernels behind the one entry point, chosen from the shape: tiles of 64 keys with an online softmax (any length;
 * LDS-DMA ring for long flat sequences), whole-sequence tiles for 128 < Tk <= 208 (single-pass softmax), one wave per
 * item for Tq, Tk <= 16.
 */
typedef struct {
  const void* Q; const void* K; const void* V; void* O;
  int64_t ldq, ldk, ldv, ldo;   /* token-row strides (elements) */
  int32_t B, H, Tq, Tk, hd;
  float scale;
  int32_t mode;
  /* mode 1 */
  int32_t Gh, Gw, ws, q_stride;
  const void* pad_k; const void* pad_v; /* f16 [H*hd] or NULL (zeros) */
  /* decomposed relative-position bias of SAM v1's ImageEncoderViT (TF sam :761-801): rel f16 [B*H*Tq][2*rel_S] from
   * lmx_k_relpos_tables; score(q, key (ky,kx)) += rel[q][ky] + rel[q][rel_S + kx], key t -> (t / rel_S, t % rel_S).
   * NULL = no bias.  Requires Tk == rel_S*rel_S and q_stride 1. */
  const void* rel; int32_t rel_S;
} lmx_attn_desc;
int lmx_k_attention(const lmx_attn_desc* d, lmx_stream_t stream);
/* rel[(b*H+h)*T + t][j]      = sum_c q[b,t,h,c] * rel_pos_h[ty - j + S-1][c]        (j < S)
 * rel[(b*H+h)*T + t][S + j]  = sum_c q[b,t,h,c] * rel_pos_w[tx - j + S-1][c]        (t = ty*S + tx, T = S*S)
 * q addressed with the attention geometry of `d` (mode 0 or window mode; d->Q, ldq, B, H, Tq, hd, mode, Gh, Gw, ws used),
 * rel_pos_h / rel_pos_w f32 [2S-1][hd] (get_rel_pos without interpolation: q_size == k_size), out f16. */
int lmx_k_relpos_tables(const lmx_attn_desc* d, const float* rel_pos_h, const float* rel_pos_w, int S, void* out,
                        lmx_stream_t stream);

/* ---- DINOv3 RoPE on the patch tokens of Q and K, in place (TF dinov3_vit :238-268) ------------------
 * x[b, t, h, :] for t >= n_prefix is rotated: x' = x*cos + rotate_half(x)*sin with cos/sin f32 [T-n_prefix][hd].
 */
int lmx_k_rope(void* x, int64_t ld, int B, int T, int H, int hd, int n_prefix, const float* cos_t,
               const float* sin_t, lmx_stream_t stream);

/* ---- K9/K21: Pillow-exact separable u8 resize (two passes, 8bpc fixed-point coefficients) -----------
 * Reproduces PIL.Image.resize(..., BICUBIC|BILINEAR, reducing_gap=None) on RGB u8 — the resampler behind
 * AutoImageProcessor (dinov3 main.py:107) and SamPredictor.set_image -> ResizeLongestSide (sam3 main.py:80).
 * Coefficient tables are computed on the host by the caller (lmx/resample.py restates Pillow's
 * precompute_coeffs / normalize_coeffs_8bpc): bounds[2*out] = (xmin, xsize), kk[out*ksize] int32.
 * Pass H: src [n][sh][sw][3] u8 -> tmp [n][sh][dw][3] u8 ; pass V: tmp -> dst [n][dh][dw][3] u8.
 * swap_rb swaps channel 0 and 2 while reading src (cv2 BGR frame -> RGB, dinov3 main.py:98-99).
 */
int lmx_k_pil_resize_h(const uint8_t* src, uint8_t* dst, int n, int sh, int sw, int dw, const int32_t* bounds,
                       const int32_t* kk, int ksize, int swap_rb, lmx_stream_t stream);
int lmx_k_pil_resize_v(const uint8_t* src, uint8_t* dst, int n, int sh, int dh, int w, const int32_t* bounds,
                       const int32_t* kk, int ksize, lmx_stream_t stream);

/* crop + /255 + (x-mean)/std + write the ViT patch matrix directly (im2col of the k=P,s=P patch conv):
 * out[(n*gh + py)*gw + px][(ky*P + kx)*3 + c] f16 (row stride ldo >= P*P*3; columns beyond P*P*3 are not
 * written — the caller zeroes them once when ldo pads K to a multiple of 8), from u8 img [n][ih][iw][3] cropped
 * at (top,left) to gh*P x gw*P.  (BitImageProcessor center_crop/rescale/normalize + Dinov*PatchEmbeddings' conv as a GEMM.)
 * lut: DEVICE f32 [3][256], lut[c][u] = normalised value of byte u in channel c (built on the host with the
 * processor's own expression, so rescale+normalize is exact by construction). */
int lmx_k_patchify_norm(const uint8_t* img, void* out, int n, int ih, int iw, int top, int left, int gh, int gw,
                        int P, int64_t ldo, const float* lut, lmx_stream_t stream);

/* tokens: out[b][0..n_prefix) = prefix[t][:] (+pos), out[b][n_prefix + p] = patch[b*np + p][:] + pos[n_prefix+p]
 * (f32 residual stream; pos may be NULL — DINOv3 has no learned position table). */
int lmx_k_assemble_tokens(const void* patch_f16, const float* prefix, const float* pos, float* out, int B, int np,
                          int n_prefix, int D, lmx_stream_t stream);

/* K22: mean over all T tokens of a f16/f32 [B][T][D] tensor -> f32 [B][D]  (dinov3 main.py:113) */
int lmx_k_token_mean(const void* x, int in_dtype, float* out, int B, int T, int D, lmx_stream_t stream);

/* ---- K8: NMS (ultralytics non_max_suppression as invoked at yolo main.py:76) ------------------------
 * pred f32 [n][A][4+nc] rows = (cx,cy,w,h, cls scores...) in letterboxed pixels.
 * Per image: keep candidates with max class score > conf; best class; sort by score descending (ties: lower
 * anchor index first); boxes offset by cls*max_wh; greedy IoU > iou suppression (torchvision.ops.nms
 * semantics, IoU computed exactly as torchvision's CPU kernel: f32, no FMA contraction, compared against the
 * DOUBLE threshold as `ovr > iou_threshold` does there); first max_det.  A <= 16384.
 * Outputs (row-major, caller allocated): boxes f32 [n][max_det][4] xyxy in letterboxed pixels (before
 * scale_boxes), scores f32 [n][max_det], cls int32 [n][max_det], src int32 [n][max_det] (anchor index),
 * counts int32 [n].  workspace: lmx_nms_workspace_bytes(n, A) bytes.
 */
int64_t lmx_nms_workspace_bytes(int n, int A);
int lmx_k_nms(const float* pred, int n, int A, int nc, float conf, double iou, int max_det, float max_wh,
              float* boxes, float* scores, int32_t* cls, int32_t* src, int32_t* counts, void* workspace,
              lmx_stream_t stream);

/* ---- YOLOv8 non-GEMM pieces (ultralytics predictor under yolo main.py:76) ---------- */
/* K1: LetterBox = cv2.resize(INTER_LINEAR, u8 fixed point) to rh x rw + constant-114 border to oh x ow at
 * (top,left) + BGR->RGB (swap_rb).  src u8 [n][sh][sw][3] -> dst u8 [n][oh][ow][3].  Tables from the host
 * (lmx/letterbox.py restates OpenCV's resizeGeneric_ table build): xofs i32 [rw], ialpha i16 [rw][2], yofs i32 [rh],
 * ibeta i16 [rh][2]; ignored (may be NULL) when rh==sh && rw==sw (LetterBox then skips the resize). */
int lmx_k_letterbox(const uint8_t* src, uint8_t* dst, int n, int sh, int sw, int rh, int rw, int top, int left,
                    int oh, int ow, const int32_t* xofs, const int16_t* ialpha, const int32_t* yofs,
                    const int16_t* ibeta, int swap_rb, lmx_stream_t stream);
/* stem Conv(3->Cout,k3,s2,p1)+bias+SiLU straight from the u8 letterboxed RGB frame (x = u8/255 in f32):
 * w f32 [3][3][3][Cout] (ky,kx,c,co), out f16 NHWC [n][H/2][W/2][Cout]; Cout%8==0. */
int lmx_k_stem_conv(const uint8_t* img, const float* w, const float* bias, void* out, int n, int H, int W, int Cout,
                    lmx_stream_t stream);
/* K5: max_pool2d(5, stride 1, pad 2) on an NHWC f16 channel slice (pixel strides lds/ldd in elements). */
int lmx_k_maxpool5(const void* src, int64_t lds, void* dst, int64_t ldd, int n, int H, int W, int C,
                   lmx_stream_t stream);
/* K6: nearest x2 upsample of an NHWC f16 slice [n][H][W][C] into a slice of a [n][2H][2W][*] buffer. */
int lmx_k_upsample2(const void* src, int64_t lds, void* dst, int64_t ldd, int n, int H, int W, int C,
                    lmx_stream_t stream);
/* K7: Detect decode of one level: head f32 [n][H][W][ldh >= 64+nc] (box logits side*16+bin, then class logits)
 * -> pred f32 [n][A][4+nc] rows a_off + y*W + x: DFL softmax expectation, dist2bbox(xywh), *stride, sigmoid. */
int lmx_k_detect_decode(const float* head, int64_t ldh, float* pred, int n, int H, int W, int nc, float stride,
                        int a_off, int A, lmx_stream_t stream);
/* ops.scale_boxes: boxes f32 [total][4] xyxy in place: (x - pad)/gain clipped to [0,w] x [0,h]. */
int lmx_k_scale_boxes(float* boxes, int total, float padx, float pady, float gain, float w, float h,
                      lmx_stream_t stream);

/* ---- EXACT-precision plan of the YOLO conv stack (lmx/yolo.py precision="exact"; csrc/exact.hip) -------------------------
 * north_star asks for NMS keep-sets bit-exact against the fp32 CPU path of yolo main.py:76; f16 activations deviate 3e-3 in
 * score.  The exact plan keeps lmx_k_gemm and feeds it operands with 22 mantissa bits: a value x travels as the f16 channel
 * triple [hi | lo | hi] per channel group of width g ("x3": hi = f16(x), lo = f16((x - hi) * 2048)), a weight row as
 * [whi | whi/2048 | wlo] over the same groups (rows pre-scaled by a power of two, undone by lmx_k_gemm's `scale`), so ONE
 * lmx_k_gemm launch over K' = 3K with f32 output yields x.w up to the dropped (x-hi)(w-whi) term (2^-22 relative).
 * lmx_k_split3: y = act(x) (+ the value of the x3 residual res3), written as x3 groups: f32 x [rows][ldx], N logical
 *   channels in groups of g (N % g == 0, g % 8 == 0): logical channel n = q*g + r sits at out3[m*ldo + q*3g + r] (hi),
 *   + g (lo), + 2g (hi again); res3 has the same grouping with pixel stride ldr.  act: NONE / SILU / GELU / RELU in torch's CPU forms: SiLU as
 *   x / (1 + exp(-x)) with a true division, GELU as 0.5 x (1 + erf(x / sqrt 2)).  Replaces the activation half of ultralytics' Conv and the
 *   shortcut add of Bottleneck under the exact plan.
 * lmx_k_maxpool5_x3: lmx_k_maxpool5 on an x3 slice of C logical channels (maximum by value; the pair travels with it).
 * lmx_k_stem_conv_x3: lmx_k_stem_conv writing x3: out3 f16 [n][H/2][W/2][3*Cout].
 * (nearest upsampling of an x3 slice is lmx_k_upsample2 over 3C channels.) */
int lmx_k_split3(const float* x, int64_t ldx, int act, const void* res3, int64_t ldr, void* out3, int64_t ldo, int64_t rows,
                 int N, int g, int nsum, int64_t sum_stride, lmx_stream_t stream);
/* (nsum > 1: x is the sum of nsum partial tensors x + s * sum_stride (elements), s = 0 .. nsum-1 added in that order — the
 * partial outputs of a split-K lmx_k_gemm launch) */
int lmx_k_maxpool5_x3(const void* src3, int64_t lds, void* dst3, int64_t ldd, int n, int H, int W, int C, lmx_stream_t stream);
int lmx_k_stem_conv_x3(const uint8_t* img, const float* w, const float* bias, void* out3, int n, int H, int W, int Cout,
                       lmx_stream_t stream);
/* The exact plan of the SAM mask decoder (lmx/sam_decoder.py precision="exact": f32 activations, x3 operands for every Linear):
 * lmx_k_attention_f32: SamAttention's softmax(q k^T * scale) v in plain f32 (TF:models/sam/modeling_sam.py:205-268) for the
 *   decoder's tiny problems — 7 tokens against 4096 image positions and back, head dim 16 or 32; q/k/v/o f32 token rows
 *   (row = b*T + t, head h at columns h*hd ..), flat geometry.  exp is expf, the division a true one.
 * lmx_k_hyper_mask_f32: lmx_k_hyper_mask on an f32 upscaled embedding `up` f32 [n][G*G][4][4][C], with the upscaler's last
 *   activation (act = LMX_ACT_GELU: exact erf form, or LMX_ACT_NONE) applied on load. */
int lmx_k_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o,
                        int64_t ldo, int B, int H, int Tq, int Tk, int hd, float scale, lmx_stream_t stream);
int lmx_k_hyper_mask_f32(const float* up, const float* hyper, float* logits, int n, int G, int C, int act, lmx_stream_t stream);

/* Pose head post-processing for the detections lmx_k_nms kept (ultralytics Pose.kpts_decode + ops.scale_coords +
 * clip_coords; the YOLOv8-pose consumer is services/tleap-pipeline/app/main.py:142-163, `result.keypoints[j].data`).
 * raw0..2: the three levels' cv4 outputs, f32 [n][h_l][w_l][ldk] (ldk >= K*ndim); hw = {h0,w0,h1,w1,h2,w2} and strides[3]
 * are HOST arrays; src / counts are lmx_k_nms's device outputs (anchor index per kept detection, detections per image).
 * out f32 [n][max_det][K][ndim]: x, y in frame pixels ((v*2 + cell)*stride, minus the UNROUNDED letterbox padding,
 * / gain, clipped to [0,w] x [0,h]) and sigmoid(visibility) when ndim == 3; rows >= counts[b] are zero. */
int lmx_k_pose_gather(const float* raw0, const float* raw1, const float* raw2, int64_t ldk, const int32_t* hw,
                      const float* strides, const int32_t* src, const int32_t* counts, int n, int max_det, int K, int ndim,
                      float padx, float pady, float gain, float w, float h, float* out, lmx_stream_t stream);

/* ---- SAM / Hiera non-GEMM pieces ------------------------------------------------------------------------ */
/* K9+K10: im2col of the Hiera patch-embed conv (k7 s4 p3, TF sam2 :120-136) fused with SamPredictor's
 * normalisation and zero padding: img u8 [n][rh][rw][3] (the PIL-resized frame) sits at the top-left of an
 * IH x IW canvas of zeros (in NORMALISED space); out f16 [n*OH*OW][ldo], column (ky*KW + kx)*3 + c =
 * lut[c][u8] or 0 outside the image.  OH = (IH + 2*pad - KH)/stride + 1.  ldo%8==0, ldo >= KH*KW*3 (tail zeroed). */
int lmx_k_im2col_u8(const uint8_t* img, const float* lut, void* out, int n, int rh, int rw, int IH, int IW, int KH,
                    int KW, int stride, int pad, int64_t ldo, lmx_stream_t stream);
/* 2x2 / stride-2 max pool on an NHWC grid (Hiera do_pool, TF sam2 :291-300), dtype f16 or f32, channel slices
 * allowed on both sides (pixel strides lds/ldd in elements).  H, W even. */
int lmx_k_maxpool2(const void* src, int64_t lds, void* dst, int64_t ldd, int dtype, int n, int H, int W, int C,
                   lmx_stream_t stream);
/* f32 -> f16 row-wise convert (stage outputs of the f32 residual stream feeding the FPN 1x1 convs). */
int lmx_k_cast_f32_f16(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int cols,
                       lmx_stream_t stream);

/* ---- SAM mask decoder glue (TF:models/sam/modeling_sam.py:432-543; K18/K19) --------------------------------------- */
/* out[r][:] = a[r][:] + b[r % b_rows][:]  (a, b f32; out f32 or f16): queries + point embeddings, keys + image PE,
 * image embedding + no-mask dense embedding. */
int lmx_k_add_bcast(const void* a, int a_dtype, int64_t lda, const float* b, int64_t ldb, int b_rows, void* out,
                    int out_dtype, int64_t ldo, int64_t rows, int D, lmx_stream_t stream);
/* SamPromptEncoder._embed_boxes (TF sam :650-659) on device: boxes f32 [n][4] xyxy in FRAME pixels are scaled to the
 * resized image (ResizeLongestSide.apply_boxes: x*nw/w, y*nh/h in double), +0.5, /S, 2c-1, @gauss [2][F], *2pi,
 * [sin|cos] -> sparse f32 [n][2][2F], + corner[0|1] (the point_embed[2|3] rows, f32 [2][2F]). */
int lmx_k_prompt_box(const float* boxes, int64_t ldb, float* sparse, int n, double sx, double sy, float S,
                     const float* gauss, const float* corner, int F, lmx_stream_t stream);
/* masks[:,0] = hyper_in[:,0] @ upscaled: up f16 [n][G*G][4][4][C] (the two ConvTranspose2d(k2,s2) outputs kept in
 * nested quadrant order: pixel (y,x) of the GxG grid, then (dy1,dx1), then (dy2,dx2)), hyper f32 [n][C] ->
 * logits f32 [n][4G][4G] in spatial order (Y = 4y + 2dy1 + dy2, X = 4x + 2dx1 + dx2).  C%8==0, C<=64. */
int lmx_k_hyper_mask(const void* up, const float* hyper, float* logits, int n, int G, int C, lmx_stream_t stream);
/* Sam.postprocess_masks + threshold + mask statistics: logits f32 [n][L][L] -> bilinear (align_corners=False) to
 * TxT, crop [:nh,:nw], bilinear to h x w, > 0  => mask u8 [n][h][w] (0/1);  stats int64 [n][8] =
 * (area, sum_x, sum_y, min_x, min_y, max_x, max_y, 0) over mask pixels (min/max = +-big when empty).
 * workspace: n*nh*nw floats (the cropped TxT intermediate, so each output pixel costs 4 taps instead of 16). */
int lmx_k_mask_post(const float* logits, int n, int L, int T, int nh, int nw, int h, int w, uint8_t* mask,
                    int64_t* stats, float* workspace, lmx_stream_t stream);
/* Bit-pack a 0/non-0 byte image: dst[r][c] holds pixels 8c..8c+7 of row r, first pixel in the most significant bit
 * (numpy.packbits order); rows are padded to ceil(w/8) bytes.  Used for the mask persisted / gathered per frame
 * (services/sam3-pipeline/app/main.py:83-89 returns a bool[H,W] mask; SURVEY.md §8b `mask_bits [n, h, ceil(w/8)]`):
 * 8x less D2H and xGMI traffic than the byte mask. */
int lmx_k_pack_bits(const uint8_t* src, int64_t rows, int w, uint8_t* dst, lmx_stream_t stream);

/* ---- contour features ON THE DEVICE ---------------------------------------------------------
 * The cv2 part of extract_segmentation_features (sam3 main.py:118-135): findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE),
 * max(contours, key=contourArea), arcLength(closed), boundingRect — for n masks [n][h][w] (0 / non-0 bytes) in HBM.
 * out int64 [n][8] = { 2 * contourArea (exact), unit steps, diagonal steps of the border chain (arcLength = unit + diag * sqrt 2),
 * min x, min y, max x, max y of the largest external contour, number of external contours (0: all other fields 0) }.
 * Parallel restatement of host_mask.cpp's border following as sums over boundary cracks (csrc/contour.hip); equal to it bit
 * for bit (tests/test_gpu_contour.py).  workspace: lmx_contour_workspace_bytes(n, h, w) bytes, 16-byte aligned. */
int64_t lmx_contour_workspace_bytes(int n, int h, int w);
int lmx_k_contour_features(const uint8_t* mask, int n, int h, int w, int64_t* out, void* workspace, lmx_stream_t stream);

/* ---- HOST function (mask pointer is HOST memory) ----------------------------------------------------------------------
 * extract_segmentation_features (sam3 main.py:102-145) on a 0/1 byte mask [h][w]: out[7] = mask_area, area_ratio,
 * circularity, aspect_ratio, centroid_x, centroid_y, perimeter.  Restates cv2.findContours(RETR_EXTERNAL,
 * CHAIN_APPROX_SIMPLE) + contourArea/arcLength/boundingRect of the largest contour + cv2.moments (cv2 absent: parity
 * unpinned).  Sequential border following on the host; lmx_k_contour_features is the device form (same numbers: the
 * perimeter is unit steps + diagonal steps * sqrt 2 in both). */
int lmx_h_mask_features(const uint8_t* mask_host, int h, int w, double* out_host);

/* ---- HOST functions of the tracking service (all pointers are HOST memory) -------------------
 * The association arithmetic of services/tracking-service/app/tracker/matching.py, the consumer of pipeline.yolo /
 * pipeline.dinov3: per frame a detections x tracks IoU matrix and one minimum-cost assignment on it (tens of boxes,
 * sequential per frame: host code in the reference, host code here; lmx/services/tracking.py is the caller).
 * lmx_h_iou_matrix: iou_batch (matching.py:12-44): a [n][4], b [m][4] xyxy boxes -> out [n][m] = inter / (union + 1e-6).
 * lmx_h_assign: linear_assignment (matching.py:69-101) = lap.lapjv(cost, extend_cost=True, cost_limit=100000) on a finite
 *   cost [n][m]: the minimum-cost assignment matching min(n, m) pairs; row_to_col [n] / col_to_row [m] hold the partner or -1.
 *   (lap is absent: parity unpinned; optimality is tested against scipy.optimize.linear_sum_assignment.) */
int lmx_h_iou_matrix(const double* a, int n, const double* b, int m, double* out);
int lmx_h_assign(const double* cost, int n, int m, int* row_to_col, int* col_to_row);

#ifdef __cplusplus
}
#endif
#endif
