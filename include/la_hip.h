/*
 * la_hip.h - C ABI of libla_hip.so, the MI355X (gfx950) kernels behind the LabelAnything hot path.
 *
 * The reference (pasqualedem/LabelAnything) is pure Python on torch and has NO FFI boundary of its
 * own: every function below replaces a chain of stock torch ops inside
 *   label_anything/models/{image_encoder,common,transformer,prompt_encoder,mask_decoder,lam}.py
 * and is called only from the Python host mirror (labelanything_amd/), which keeps the reference's
 * module API.  Each entry point cites the reference lines it stands in for.
 *
 * Conventions
 *  - plain C, raw DEVICE pointers (tensor.data_ptr()), explicit sizes / leading dimensions in ELEMENTS;
 *  - `dt` selects the 16-bit storage / MFMA operand type: LA_F16 or LA_BF16 (accumulation is fp32);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden syncs,
 *    no allocation, the caller owns every buffer and workspace;
 *  - returns 0 on success, <0 on error; la_last_error() gives the message (thread local).
 */
#ifndef LA_HIP_H
#define LA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

enum { LA_F16 = 0, LA_BF16 = 1 };
enum { LA_ACT_NONE = 0, LA_ACT_GELU = 1, LA_ACT_RELU = 2 };
/* output row mappings of la_gemm (see LaGemmEpilogue.map) */
enum { LA_MAP_NONE = 0, LA_MAP_GROUP = 1, LA_MAP_WINDOW_MERGE = 2, LA_MAP_CONVT2X2 = 3 };
/* la_attn_fwd modes */
enum { LA_ATTN_PLAIN = 0, LA_ATTN_RELPOS = 1 };

const char* la_last_error(void);
int la_version(void);

/* Epilogue of la_gemm: out = map( act(A.W^T + bias) + residual ).
 *  map LA_MAP_GROUP        dst_row = (row / p0) * p1 + row % p0 + p2        (CLS-gap insertion for the HF ViT)
 *  map LA_MAP_WINDOW_MERGE rows are window-partitioned tokens (p0 = window, p1 = #win y, p2 = #win x,
 *                          p3 = H, p4 = W): dst_row = (b*H + y)*W + x, padded tokens are dropped
 *                          (window_unpartition, image_encoder.py:282-304)
 *  map LA_MAP_CONVT2X2     ConvTranspose2d(k=2,s=2) as a GEMM with pixel shuffle: rows are (b, y, x) on a p1 x p0
 *                          (H x W) grid, cols are (ky, kx, cout) with p2 = cout: dst_row = (b*2H + 2y+ky)*2W + 2x+kx,
 *                          dst_col = cout   (mask_decoder.py:206-222)
 *  residual is fp32, indexed by (res_mod ? dst_row % res_mod : dst_row), dst_col.
 *  vt != NULL: columns >= vt_col0 are NOT written to out16 but transposed into
 *  vt[((row / vt_T) * vt_heads + head) * vt_hd + d][vt_Tpad] at token row % vt_T  (V operand of la_attn_fwd).
 */
typedef struct LaGemmEpilogue {
  const float* bias;   /* [N] or NULL (for LA_MAP_CONVT2X2: [cout]) */
  const float* res;    /* fp32 residual or NULL */
  int ldr;
  int res_mod;
  float* out32;        /* optional fp32 output */
  int ld32;
  void* out16;         /* optional 16-bit output */
  int ld16;
  int act;
  int map;
  int p0, p1, p2, p3, p4;
  void* vt;
  int vt_col0, vt_T, vt_Tpad, vt_hd, vt_heads;
} LaGemmEpilogue;

/* C[M,N] = A[M,K] . W[N,K]^T (nn.Linear layout), 16-bit operands, fp32 accumulate on MFMA.
 * K % 8 == 0.  Replaces every nn.Linear / 1x1 conv / im2col'd conv / k=s ConvTranspose of the path:
 * image_encoder.py:242,254 (qkv, proj), common.py:37 (MLPBlock), image_encoder.py:399-410 (PatchEmbed),
 * build_lam.py:154-170 (neck), common.py:103-105,146 (decoder projections), mask_decoder.py:206-229. */
int la_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K,
            const LaGemmEpilogue* epi, int dt, void* stream);

/* Row LayerNorm over the last dim (biased variance):  y = LN(x [+ x2]) * gamma + beta  [-> GELU].
 * x, x2 fp32 [rows, E] (ldx).  Outputs (each optional): out32 fp32, out16, out16_pe = y + pe[(row % pe_mod)]
 * (pe fp32 [pe_mod, E]).  window > 0: rows are (b, y, x) tokens on an H x W grid and the 16-bit outputs are written
 * in window-partitioned order (pad rows are left untouched: the caller zero-fills them once)
 * (image_encoder.py:179-187,258-279; nn.LayerNorm; LayerNorm2d common.py:42-54 in NHWC). */
int la_layernorm(const float* x, const float* x2, int ldx, int rows, int E, const float* gamma, const float* beta,
                 float eps, int gelu, float* out32, void* out16, void* out16_pe, const float* pe, int pe_mod,
                 int window, int H, int W, int dt, void* stream);

/* Patch-embed im2col: image fp32 NCHW [Bn,3,S,S] -> A16 [Bn*g*g, 3*p*p], k = c*p*p + ky*p + kx
 * (image_encoder.py:399-410, Conv2d k=s=patch). */
int la_im2col_patch(const float* img, int Bn, int S, int patch, void* out16, int dt, void* stream);

/* 3x3 / pad 1 im2col on an NHWC 16-bit map [B,H,W,C] -> [B*H*W, 9*C], k = (ky*3+kx)*C + c  (C % 8 == 0)
 * (neck conv image_encoder.py:100-106, spatial convs mask_decoder.py:236-255). */
int la_im2col_3x3(const void* in16, int B, int H, int W, int C, void* out16, int dt, void* stream);

/* Decomposed relative-position terms (image_encoder.py:340-376) from the UNSCALED q of a fused qkv buffer:
 *   relh[bh][q][kh] = q . Rh[qy - kh + G-1],   relw[bh][q][kw] = q . Rw[qx - kw + G-1],  fp32 [B*heads, T, G], T = G*G.
 * qkv: 16-bit [B*T, 3E] (q at column head*64); tabh/tabw: 16-bit [(2G-1), 64]. head_dim == 64. */
int la_relpos_terms(const void* qkv, int B, int heads, int G, int E, const void* tabh, const void* tabw,
                    float* relh, float* relw, int dt, void* stream);

/* Flash attention over fused qkv (head_dim 64): out[B*T, E] = softmax(q k^T * scale + bias) v.
 * qkv 16-bit [B*T, 3E] supplies q and k; vt is V transposed per (b, head): [B*heads, 64, Tpad] (zero padded,
 * written by la_gemm's vt epilogue).  mode LA_ATTN_RELPOS adds relh[q][k / G] + relw[q][k % G]
 * (image_encoder.py:246-253); LA_ATTN_PLAIN is the HF ViT softmax(qk^T/sqrt(d))v.  The T x T score matrix is
 * never materialised. */
int la_attn_fwd(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw,
                int B, int heads, int T, int Tpad, int G, int E, float scale, int mode, int dt, void* stream);

#ifdef __cplusplus
}
#endif
#endif
