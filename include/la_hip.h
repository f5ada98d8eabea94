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
 *  - `dt` selects the storage / MFMA operand type of the "16-bit" buffers: LA_F16, LA_BF16 (accumulation is fp32)
 *    or LA_F32 (the same buffers then hold fp32; supported by every entry point except the encoder attention pair);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden syncs,
 *    no allocation, the caller owns every buffer and workspace;
 *  - returns 0 on success, <0 on error; la_last_error() gives the message (thread local).
 */
#ifndef LA_HIP_H
#define LA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* LA_F32: operands stay fp32 and la_gemm runs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the 16-bit rate);
 * used for the small decoder stages where 16-bit operand rounding would dominate the logit error. */
/* LA_F16X2 (la_layernorm, la_add_cast, la_attn_small, la_mask_embed outputs only): the "16-bit" buffers hold TWO fp16 planes per row,
 * [hi (E) | lo (E)] with hi = rn(v), lo = rn(v - hi) (row stride 2 E).  Fed to la_gemm as A with a_kmod = 2 K against weights packed as
 * [W_hi | W_hi | W_lo] (K' = 3 K) the product carries ~21 mantissa bits through the fast MFMA: the decoder-side (P, hw, D) stream at
 * fp32-level accuracy for 3 fp16 passes instead of the 16 passes an exact-fp32 MFMA costs. */
enum { LA_F16 = 0, LA_BF16 = 1, LA_F32 = 2, LA_F16X2 = 3 };
enum { LA_ACT_NONE = 0, LA_ACT_GELU = 1, LA_ACT_RELU = 2, LA_ACT_GELU_BWD = 3 /* la_gemm only, see LaGemmEpilogue.aux16 */ };
/* output row mappings of la_gemm (see LaGemmEpilogue.map) */
enum { LA_MAP_NONE = 0, LA_MAP_GROUP = 1, LA_MAP_WINDOW_MERGE = 2, LA_MAP_CONVT2X2 = 3, LA_MAP_WINDOW_PART = 4, LA_MAP_CONV3X3 = 5 /* amap only */ };
/* la_attn_fwd modes */
/* LA_ATTN_RELPOS_WIN16: SAM window attention with the keys held in a 16-wide padded slot order (see la_attn_fwd) */
enum { LA_ATTN_PLAIN = 0, LA_ATTN_RELPOS = 1, LA_ATTN_RELPOS_WIN16 = 2 };

const char* la_last_error(void);
int la_version(void);

/* Tuning hook of la_gemm (measurement A/B only, results are bit-identical): 1 = 64-deep quadrant-phase main loop where it
 * applies (default), 0 = the 32-deep persistent kernel everywhere; v < 0 only queries.  Returns the previous value.
 * Only bit 0 is honoured by the product library; the timing ablations behind the higher bits and every environment
 * override of kernel selection are compiled into the -DLA_DEBUG library alone (`make -C labelanything_amd/csrc DEBUG=1`). */
int la_gemm_variant(int v);

/* Epilogue of la_gemm: out = map( act(A.W^T + bias) + residual ).
 *  map LA_MAP_GROUP        dst_row = (row / p0) * p1 + row % p0 + p2        (CLS-gap insertion for the HF ViT)
 *  map LA_MAP_WINDOW_MERGE rows are window-partitioned tokens (p0 = window, p1 = #win y, p2 = #win x,
 *                          p3 = H, p4 = W): dst_row = (b*H + y)*W + x, padded tokens are dropped
 *                          (window_unpartition, image_encoder.py:282-304)
 *  map LA_MAP_CONVT2X2     ConvTranspose2d(k=2,s=2) as a GEMM with pixel shuffle: rows are (b, y, x) on a p1 x p0
 *                          (H x W) grid, cols are (ky, kx, cout) with p2 = cout: dst_row = (b*2H + 2y+ky)*2W + 2x+kx,
 *                          dst_col = cout   (mask_decoder.py:206-222)
 *  map LA_MAP_WINDOW_PART  the inverse: rows are (b, y, x) tokens of an H x W grid, dst_row = their place in the
 *                          window-partitioned order (same p0..p4; window_partition, image_encoder.py:258-279).  The padded
 *                          tokens of the destination are never written: their q, k, v are the bias vector, which the
 *                          caller stores there once.
 *  amap (same encodings, same p0..p4, at most one of map / amap non-zero): a map applied to the SOURCE rows instead -
 *                          GEMM row m reads A[amap(m)] and writes row m.  With LA_MAP_WINDOW_PART the proj GEMM of a SAM
 *                          window block walks the H x W tokens only and gathers its input from the window-ordered
 *                          attention output, so neither window GEMM touches the 16 % padded tokens.
 *  amap LA_MAP_CONV3X3 (round 6; p0 = padded row width W + 2, p1 = channels C per plane, C % 64 == 0, p2 = lda = 2 C): an IMPLICIT 3 x 3 / pad 1
 *                          convolution on a zero-bordered NHWC map of fp16 plane pairs - A rows are the pixels of [B, H + 2, W + 2] maps
 *                          ([hi (C) | lo (C)] per pixel, W + 3 zero rows in front of and behind the buffer), GEMM row m = padded pixel m
 *                          (border rows compute values nobody reads), K = 27 C against W = [W_hi | W_hi | W_lo] in (ky, kx, c) order:
 *                          k-tile k0 reads plane (k0 % 18 C) / 9 C, tap t = (k0 % 9 C) / C at row offset (t / 3 - 1) p0 + (t % 3 - 1) -
 *                          a wave-uniform shift of the source base per k-tile, no im2col buffer (the SAM neck's second convolution,
 *                          image_encoder.py:100-106: 3.6 GB written and read again for a 0.4 GB map).
 *  a_kmod > 0 (16-bit operands, a_kmod % 64 == 0, K % a_kmod == 0): the A columns REPEAT with period a_kmod while W runs over all
 *                          K columns, i.e. C = A[:, :a_kmod] . (W[:, 0:a_kmod] + W[:, a_kmod:2 a_kmod] + ...)^T accumulated in fp32 with
 *                          every partial product formed separately.  With W = [W_hi | W_lo] (W_hi = the 16-bit rounding of an fp32
 *                          weight, W_lo = the 16-bit rounding of what it lost) the weights enter with ~22 mantissa bits: the
 *                          split-precision mode that brings the encoder inside the 1e-3 logit tolerance.
 *  residual is fp32, indexed by (res_mod ? dst_row % res_mod : dst_row), dst_col.
 *  vt != NULL: columns >= vt_col0 are NOT written to out16 but transposed into
 *  vt[((row / vt_T) * vt_heads + head) * vt_hd + d][vt_Tpad] at token row % vt_T  (V operand of la_attn_fwd), row = dst_row.
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
  int vt_ws;           /* > 0: token t of a ws x ws window goes to slot (t / ws) * 16 + t % ws of the V^T row (LA_ATTN_RELPOS_WIN16) */
  int amap;            /* source-row map (LA_MAP_NONE or LA_MAP_WINDOW_PART), see above */
  int a_kmod;          /* > 0: period of the A columns (split-precision weights [W_hi | W_lo]), see above */
  int ksplit;          /* != 0 (16-bit operands): out32 += A . W^T - the bare product ADDED with fp32 atomics, the K range cut into
                          independent chunks so that a product with few output tiles and a very long K (a weight gradient
                          dW[N, K] = dY^T X over all tokens, operands passed transposed) still fills the chip.  N % 256 == 0, K % 64 == 0;
                          no bias / residual / activation / out16; row map LA_MAP_NONE or LA_MAP_GROUP (several weight gradients a fixed
                          stride apart in one flat gradient buffer - HF's query / key / value weights - from ONE launch). */
  void* aux16;         /* training (models/common.py:36-37 MLPBlock / transformers ViTIntermediate when the backbone trains, lam.py:321-347): a second
                          16-bit matrix [M, N] (row stride ldaux).  act == LA_ACT_GELU: WRITTEN - the pre-activation A W^T + b beside
                          out16 = GELU of it (what gelu' needs in the backward).  act == LA_ACT_GELU_BWD: READ - out16 = (A W^T) * gelu'(aux16),
                          the data gradient through a GELU in one pass (A = dY, W = the transposed weight).  Only shapes for which
                          la_gemm_fused_act_ok() says 1 (the persistent four-wave kernel); out16 only, no residual / maps / V^T. */
  int ldaux;
  /* LayerNorm folded into the GEMMs on both sides of it (round 6; image_encoder.py:181-197 `x = x + attn(norm1(x))`, `x + mlp(norm2(x))`,
   * transformers ViTLayer layernorm_before / layernorm_after): the LayerNorm pass over the residual stream (24 launches, 8 % of a cfg2
   * step) disappears.  16-bit fp16 operands, N % 256 == 0, K % 64 == 0, K >= 128, no row maps / V^T / ksplit / aux16: the direct epilogue
   * of the persistent four-wave kernel.
   * PRODUCER (the GEMM that writes the stream; needs res or bias, out32 AND out16): nstat_out != NULL - beside out32 = acc + bias + res
   *   (+ rvec[row / rvec_rpg][col], a per-group fp32 vector [groups, N]: the token-mean correction of the block, added to the stream HERE
   *   instead of being carried as a pending vector) and out16 = its 16-bit rounding, every epilogue round leaves the partial row sums of its 64
   *   columns: nstat_out[(row * (N / 64) + col / 64) * 2 + {0, 1}] = sum x, sum x^2 over those columns (fp32, fixed order: deterministic).
   *   la_norm_finalize turns the N / 64 partials of a row into (mean, rstd).  out16 saturates at the fp16 range (it is an MFMA operand; an
   *   un-normalised stream has no range guarantee).  aux16 != NULL (producer only): the lo plane rn16(x - out16) of the same rows - with
   *   out16 / aux16 pointing at the two halves of [rows, 2 N] rows the stream leaves the block stack as an LA_F16X2 operand (the SAM neck's
   *   1 x 1 convolution) without a pass of its own.  With out32 == NULL and res == NULL the STREAM ITSELF is that pair of planes, updated in
   *   place: out16 / aux16 are read as the residual (hi + lo, ~22 mantissa bits per value: 25 read-modify-writes cost 1e-6) and written
   *   back - same bytes per row as the fp32 read-modify-write, and the hi plane is the next GEMM's operand at no extra bytes.
   * CONSUMER (the GEMM behind the LayerNorm; out16 only, act NONE or GELU): nstat_in != NULL - A is the UN-normalised 16-bit stream, W the
   *   16-bit rounding of W diag(gamma), and the epilogue applies the normalisation to the product:
   *     out = act( rstd[row] * (acc - mean[row] * ncol[col]) + bias[col] ),   ncol[col] = sum_k W16[col][k],  bias = b + W beta
   *   which is LayerNorm(x16) W^T + b with the row statistics of the fp32 stream.  nstat_in: fp32 [ceil(M / 256) * 256][2] = (mean, rstd). */
  float* nstat_out;
  const float* rvec;
  int rvec_rpg;
  const float* nstat_in;
  const float* ncol;
} LaGemmEpilogue;

/* 1 when la_gemm takes LaGemmEpilogue.aux16 for this shape (N % 256 == 0, K % 64 == 0, K >= 128, at least one round of 256 x 256 tiles). */
int la_gemm_fused_act_ok(int M, int N, int K);

/* C[M,N] = A[M,K] . W[N,K]^T (nn.Linear layout), 16-bit operands, fp32 accumulate on MFMA.
 * K % 8 == 0.  Replaces every nn.Linear / 1x1 conv / im2col'd conv / k=s ConvTranspose of the path:
 * image_encoder.py:242,254 (qkv, proj), common.py:37 (MLPBlock), image_encoder.py:399-410 (PatchEmbed),
 * build_lam.py:154-170 (neck), common.py:103-105,146 (decoder projections), mask_decoder.py:206-229. */
int la_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K,
            const LaGemmEpilogue* epi, int dt, void* stream);

/* 3x3 / pad 1 convolution as an IMPLICIT GEMM on the exact-fp32 MFMA (no im2col buffer): in fp32 NHWC [B,H,W,Cin]
 * (Cin % 32 == 0), wt fp32 [Cout, (ky,kx,cin)], bias fp32 [Cout] or NULL -> out32 fp32 NHWC [B*H*W, Cout]
 * (spatial convs of the mask decoder, mask_decoder.py:236-255). */
int la_conv3x3_f32(const float* in, int B, int H, int W, int Cin, const float* wt, const float* bias, int Cout, float* out32,
                   void* stream);

/* The same convolution for Cin == Cout == 32 (the D = 256 mask decoder's spatial convolutions, mask_decoder.py:236-255) in SPLIT precision on the
 * 16-bit MFMA: input pixels and weights enter as fp16 plane pairs (hi = rn16(x), lo = rn16(x - hi)), three products per k-step
 * (A_hi W_hi + A_lo W_hi + A_hi W_lo, ~22 mantissa bits: the accuracy class of LA_F16X2, 1e-6 relative), the 10 x 34 halo of an 8 x 32
 * pixel tile staged once in LDS.  Same arguments and layouts as la_conv3x3_f32; la_conv3x3_split_ok says whether the channel counts are taken.
 * For ACTIVATIONS (magnitudes around one): an fp16 plane pair resolves nothing below 6e-8 and saturates at 65504 - gradient maps (1e-8 ...
 * 1e-5 entries under a mean-reduced objective) keep la_conv3x3_f32. */
int la_conv3x3_split_ok(int Cin, int Cout);
int la_conv3x3_split(const float* in, int B, int H, int W, int Cin, const float* wt, const float* bias, int Cout, float* out32, void* stream);

/* Row LayerNorm over the last dim (biased variance):  y = LN(x [+ x2]) * gamma + beta  [-> GELU].
 * x, x2 fp32 [rows, E] (ldx).  Outputs (each optional): out32 fp32, out16, out16_pe = y + pe[(row % pe_mod)]
 * (pe fp32 [pe_mod, E]).  window > 0: rows are (b, y, x) tokens on an H x W grid and the 16-bit outputs are written
 * in window-partitioned order (pad rows are left untouched: the caller zero-fills them once); window == -1: the 16-bit outputs go to the
 * interior of zero-bordered [B, H + 2, W + 2] maps (row (b (H + 2) + y + 1)(W + 2) + x + 1: the operand layout of LA_MAP_CONV3X3, borders
 * untouched); window == -2: the INPUT rows are read from that layout, outputs in plain (b, y, x) order
 * (image_encoder.py:179-187,258-279; nn.LayerNorm; LayerNorm2d common.py:42-54 in NHWC). */
int la_layernorm(const float* x, const float* x2, int ldx, int rows, int E, const float* gamma, const float* beta,
                 float eps, int gelu, float* out32, void* out16, void* out16_pe, const float* pe, int pe_mod,
                 int window, int H, int W, int dt, void* stream);

/* Patch-embed im2col: image fp32 NCHW [Bn,3,S,S] -> A16 [Bn*g*g, 3*p*p], k = c*p*p + ky*p + kx
 * (image_encoder.py:399-410, Conv2d k=s=patch). */
int la_im2col_patch(const float* img, int Bn, int S, int patch, void* out16, int dt, void* stream);

/* 3x3 / pad 1 im2col on an NHWC 16-bit map [B,H,W,C] -> [B*H*W, 9*C], k = (ky*3+kx)*C + c  (C % 8 == 0)
 * (neck conv image_encoder.py:100-106, spatial convs mask_decoder.py:236-255).  dt = LA_F16X2: pixel rows are fp16 plane pairs
 * [hi (C) | lo (C)] and the output rows [9 taps of hi | 9 taps of lo] (18 C columns). */
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
 * never materialised.  For G <= 16 (SAM windows) and G == 64 (SAM global blocks at 1024 px) pass the 16-bit rel-pos tables
 * tabh/tabw [(2G-1), 64] instead of relh/relw: the decomposed terms are then computed inside the kernel (a few MFMAs
 * per query tile) and la_relpos_terms is not needed.
 * mode LA_ATTN_RELPOS_WIN16 (G <= 16, tables required): keys live in slot order s = kh*16 + kw (16*G slots, Tpad >= 16*G,
 * V^T written with LaGemmEpilogue.vt_ws = G, K rows gathered by the kernel).  A 64-slot tile is then exactly four key
 * rows, so the bias of every score register is bh[4 per tile] + bw[8 per lane] instead of three LDS lookups. */
int la_attn_fwd(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw,
                const void* tabh, const void* tabw, int B, int heads, int T, int Tpad, int G, int E, float scale, int mode,
                int dt, void* stream);

/* ---- decoder side (prompt encoder + mask decoder) ------------------------------------------------------------- */

/* Dense positional encoding of a g x g grid: out fp32 [g*g, D], channels [sin | cos]
 * (PositionEmbeddingRandom.forward, prompt_encoder.py:213-224).  gauss: fp32 [2, D/2]. */
int la_dense_pe(const float* gauss, int g, int D, float* out, void* stream);

/* Sparse prompt tokens (points / box corners), prompt_encoder.py:83-114,599-611,648-669.
 * xy fp32 [n,2] in input-image pixels; kind[n]: 0 NULL -> not_a_point_embed, 1 negative point, 2 positive point,
 * 3 / 4 box corners, 5 the "no sparse prompt" token; shift[n] != 0 adds the half-pixel offset.
 * type_emb fp32 [4, D] = point_embeddings.{0..3}.weight.  out32 fp32 [n, D]. */
int la_point_embed(const float* xy, const int* kind, const int* shift, int n, int D, int image_size, const float* gauss,
                   const float* type_emb, const float* not_a_point, const float* no_sparse, float* out32, void* stream);

/* Fused dense-prompt path: mask_downscaling (2x conv2x2s2 + LN2d + GELU, conv1x1), bilinear 64->g, not_a_mask /
 * no_mask replacement, + support features + class encoding (prompt_encoder.py:516-540,787-814, :250-262).
 * masks fp32 [P, Hm, Hm] or NULL (no mask prompts); flags int32 [P] or NULL; P = B*M*C pairs, class = p % C,
 * support image = p / C.  w: 12 device pointers {conv0.w, conv0.b, ln1.w, ln1.b, conv3.w, conv3.b, ln4.w, ln4.b,
 * conv6.w [D,16], conv6.b, not_a_mask [D], no_mask [D]}.  support fp32 [P/C, g*g, D] (NHWC) or NULL;
 * class_enc fp32 [C, D] or NULL; pe fp32 [g*g, D].  Outputs [P*g*g, D]: src32 (fp32 stream), src16, srcpe16 = src+pe. */
int la_mask_embed(const float* masks, const int* flags, int P, int C, int Hm, int g, int D, const float* const* w,
                  const float* support, const float* class_enc, const float* pe, float* src32, void* src16, void* srcpe16,
                  int dt, void* stream);

/* Decoder attention core after the q/k/v projections (common.py:126-144): softmax(q k^T / sqrt(hd)) v in fp32.
 * q [B,Nq,ldq], k/v [B,Nk,ld*] fp32, head h at column h*hd; hd in {4,8,16,32,64}.  Output [B,Nq,ldo] 16-bit and/or fp32. */
int la_attn_small(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int B, int Nq, int Nk, int heads,
                  int hd, void* out16, float* out32, int ldo, int dt, void* stream);

/* Mean over the hw rows of each [hw, D] slab: x fp32 [P, hw, D] -> out fp32 [P, D] (prompt_encoder.py:735-736).
 * scratch: fp32 [P, 16, D] workspace for the per-chunk partial sums (deterministic two-pass reduction). D % 4 == 0. */
int la_colmean(const float* x, int P, int hw, int D, float* out, float* scratch, void* stream);

/* Class prototypes: masked mean over supports, divisor clamped to >= 1 (prompt_encoder.py:738-745).
 * emb fp32 [B,M,C,D], flags u8 [B,M,C] -> out fp32 [B,C,D]. */
int la_class_mean(const float* emb, const unsigned char* flags, int B, int M, int C, int D, float* out, void* stream);

/* seg[b][c][pix] = protos[b][c] . feat[b][pix]; feat fp32 NHWC [B,Npix,Cf], protos fp32 [B,C,Cf], seg fp32 [B,C,Npix]
 * (mask_decoder.py:299-314). Cf in {8,16,32,64}. */
int la_classify(const float* feat, const float* protos, int B, int Npix, int C, int Cf, float* seg, void* stream);

/* out = x + y[row % ymod] (y optional) as fp32 and/or 16-bit, contiguous [rows, D] (q = tokens + pe, transformer.py:305-326). */
int la_add_cast(const float* x, const float* y, int ymod, long rows, int D, float* out32, void* out16, int dt, void* stream);

/* Layout shuffles at the module boundary: NCHW fp32 <-> NHWC (the kernels work on [pixels, channels]). */
int la_nchw_to_nhwc(const float* in, int N, int C, int HW, float* out32, void* out16, int dt, void* stream);
int la_nhwc_to_nchw(const float* in, int N, int C, int HW, float* out, void* stream);

/* Many small contiguous fp32 matrices transposed by one launch: dst[c][r] = src[r][c].  tile_table (device memory): 4 x int64 per 32 x 32
 * tile - (src pointer, dst pointer, rows << 32 | cols, r0 << 32 | c0).  The W^T copies of every nn.Linear of the decoder for its data
 * gradient dX = dY W (autograd of models/common.py:19-37, transformer.py, prompt_encoder.py, mask_decoder.py under experiment/run.py:247-331):
 * once per training step instead of one launch per layer. */
int la_transpose_many(const long long* tile_table, int ntiles, void* stream);

/* F.interpolate(mode="bilinear", align_corners=False) on fp32 planes [N,h,w] -> [N,H,W] (lam.py:408-413). */
int la_bilinear(const float* in, int N, int h, int w, int H, int W, float* out, void* stream);

/* Second half of Lam.postprocess_masks + flag_gts masking + argmax (lam.py:415-453,92-93; run.py:697).
 * big fp32 [B,C,S,S]; sizes int32 [B,4] = (orig_h, orig_w, crop_h, crop_w); flag_gts u8 [B,C] or NULL.
 * logits fp32 [B,C,Hmax,Wmax] and/or argmax int64 [B,Hmax,Wmax]. */
int la_post_final(const float* big, int B, int C, int S, const int* sizes, const unsigned char* flag_gts, int Hmax, int Wmax,
                  float* logits, long long* argmax, void* stream);

/* Evaluation metrics without the label maps leaving HBM (experiment/run.py:697-704, utils/metrics.py:28-53,
 * data/utils.py:567-590): accumulates the K x K multiclass confusion matrix (row = target, column = prediction, pixels with
 * target == ignore_index dropped) and the 2 x 2 foreground/background matrix (labels > 0 -> 1) of B label maps of HW pixels.
 * pred, gt int64 [B, HW]; lut int32 [B, L] or NULL maps episode-local labels 0..L-1 of item b to dataset labels (the chained
 * replacement of to_global_multiclass, collapsed on the host); confmat u64 [K*K], confbin u64 [4], counters u64 [1]
 * (counters[0] += number of labels outside [0, K), which torchmetrics rejects) are ACCUMULATED - zero them first. */
int la_confmat_update(const long long* pred, const long long* gt, int B, long HW, const int* lut, int L, int K,
                      long long ignore_index, unsigned long long* confmat, unsigned long long* confbin,
                      unsigned long long* counters, void* stream);

/* Image preprocessing on the device (the step before the path; data/transforms.py:14-46 via torchvision -> PIL).
 * la_resample_u8: ONE axis of PIL's antialiased 8-bit resample (Image.resize, BILINEAR): in u8 [n_outer, in_size, inner] ->
 * out u8 [n_outer, out_size, inner]; bounds int32 [out_size, 2] = (first tap, tap count), kk int32 [out_size, ksize] =
 * coefficients with 22 fractional bits, both computed on the host exactly like PIL's precompute_coeffs /
 * normalize_coeffs_8bpc.  Horizontal pass: (n_outer, in_size, inner) = (H, W, C); vertical: (1, H, W*C).  Bit-exact with PIL.
 * la_u8_to_chw_norm: u8 HWC [h, w, 3] -> fp32 CHW [3, SH, SW]: ToTensor (x / 255), (x - mean) / std, zero padding on the
 * right / bottom (CustomNormalize); mean3 / std3 are HOST pointers. */
int la_resample_u8(const unsigned char* in, long n_outer, int in_size, int inner, int out_size, const int* bounds, const int* kk,
                   int ksize, unsigned char* out, void* stream);
int la_u8_to_chw_norm(const unsigned char* in, int h, int w, int SH, int SW, const float* mean3, const float* std3, float* out,
                      void* stream);

/* PromptsProcessor.apply_masks + the mask branch of annotations_to_tensor (data/transforms.py:203-224, data/utils.py:219-223)
 * for P prompt slots that share one image geometry: slot p ORs the instance masks index[first[p] .. first[p] + count[p])
 * (u8 [n, H, W], non-zero = set), resizes nearest to (nh, nw), zero-pads to S x S and resizes nearest to Mo x Mo (torch
 * "nearest" index rule); nh == 0 skips the first resize + pad (custom_preprocess off).  out fp32 [P, Mo, Mo] in {0, 1};
 * flags u8 [P] (zeroed by the caller) is set to 1 where the slot's result has a set pixel. */
int la_prompt_masks(const unsigned char* masks, const int* first, const int* count, const int* index, int P, int H, int W, int nh,
                    int nw, int S, int Mo, float* out, unsigned char* flags, void* stream);

/* Training objective, first link: the focal term of LabelAnythingLoss with class weighting (loss/__init__.py:67-89,
 * loss/focal.py:17-26, loss/utils.py:17-43) fused with its gradient.  logits fp32 [B, C, HW] (-inf padding allowed where the
 * target is ignore_index), target int64 [B, HW]; loss fp32 [1] = scale * mean over ALL B*HW pixels of (1 - pt)^gamma * w[t] * ce;
 * dlogits fp32 [B, C, HW] or NULL; class_weights fp32 [C] or NULL (the per-batch weights 1 / log(1.1 + share), 1 for absent
 * classes; all 1 when class_weighting == 0).  scratch: device workspace, (C + 2) * 8 + 2048 * 8 bytes; after the call its 64-bit word
 * C + 1 holds the number of targets outside [0, C) other than ignore_index (torch raises on those; they contribute nothing here). */
int la_focal_loss(const float* logits, const long long* target, int B, int C, long HW, float gamma, int class_weighting, float scale,
                  long long ignore_index, float* loss, float* dlogits, float* class_weights, void* scratch, long scratch_bytes,
                  void* stream);

/* One torch.optim.AdamW step on flat fp32 buffers of n elements (experiment/utils.py:53-76; decoupled weight decay, bias
 * correction with the 1-based `step`), gradients scaled by grad_scale first (1 / world size after the data-parallel SUM
 * all-reduce). */
int la_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ---- backward kernels of the decoder-side training step (fp32 like the decoder forward).  Together with the forward
 * entry points above they are the autograd graph of label_anything/models/{common,transformer,prompt_encoder,mask_decoder,lam}.py
 * under experiment/utils.py:266-303 (WrapperModule: model forward + LabelAnythingLoss) and experiment/run.py:359-361 (backward). ---- */

/* dW[N,K] += dY[M,N]^T . X[M,K]: weight gradient of nn.Linear / 1x1 conv / k = s conv(-transpose) (exact-fp32 MFMA, split over M with
 * atomic accumulation: the caller zero-fills or pre-loads dW). */
int la_gemm_tn(const float* dy, int ldy, const float* x, int ldx, float* dw, int ldw, int M, int N, int K, void* stream);
/* la_gemm_tn that also accumulates the bias gradient db[n] += sum_m dy[m][n] from the dY operands it loads (db fp32 [N]; NULL = la_gemm_tn):
 * the weight and bias gradient of an nn.Linear / 1x1 conv in one pass over dY (autograd of models/common.py:19-37, transformer.py). */
int la_gemm_tn_db(const float* dy, int ldy, const float* x, int ldx, float* dw, int ldw, int M, int N, int K, float* db, void* stream);

/* out[N] += column sums of dY[M,N] (fp32): the bias gradient of nn.Linear / the conv layers (autograd of the `+ bias` in
 * models/common.py, transformer.py, mask_decoder.py). */
int la_colsum_acc(const float* dy, int ldy, long M, int N, float* out, void* stream);

/* Backward of la_layernorm (nn.LayerNorm / LayerNorm2d common.py:42-54, optionally followed by GELU): x, dy fp32 [rows, E] contiguous ->
 * dx (written), dgamma / dbeta fp32 [E] (ACCUMULATED). */
int la_layernorm_bwd(const float* x, const float* dy, long rows, int E, const float* gamma, const float* beta, float eps, int gelu,
                     float* dx, float* dgamma, float* dbeta, void* stream);
/* The same with the block's skip connection folded in: dx = LayerNorm backward + add (add may be dx itself - the running fp32 gradient of
 * the residual stream, image_encoder.py:178-197 / transformers ViTLayer: x + f(LN(x))), and an optional 16-bit copy of dx (out16, dt16 =
 * LA_F16 / LA_BF16; NULL = none) for the next backward GEMM.  Wave-per-row form only (not the E <= 32, rows >= 65536 LayerNorm2d form). */
int la_layernorm_bwd_res(const float* x, const float* dy, long rows, int E, const float* gamma, const float* beta, float eps, int gelu,
                         const float* add, float* dx, void* out16, int dt16, float* dgamma, float* dbeta, void* stream);

/* y = act(x) and dx = dy * act'(x), act = LA_ACT_GELU (erf form) or LA_ACT_RELU, n contiguous fp32 elements. */
int la_act_fwd(const float* x, float* y, long n, int kind, void* stream);
int la_act_bwd(const float* x, const float* dy, float* dx, long n, int kind, void* stream);

/* Row statistics lse[(b * Nq + q) * heads + h] = log sum_j exp(q . k_j / sqrt(hd)) of la_attn_small (needed by the backward when
 * the QUERIES are the few tokens). */
int la_attn_small_lse(const float* q, int ldq, const float* k, int ldk, int B, int Nq, int Nk, int heads, int hd, float* lse, void* stream);

/* Backward of la_attn_small (Attention of models/common.py:57-148, score scale 1/sqrt(hd)): layouts as in the forward; dq / dk / dv have
 * the leading dimensions of q / k / v.  Nk <= 256 (and Nk <= Nq or Nq > 256): dq is written, dk / dv ACCUMULATED (zero-fill them);
 * otherwise Nq <= 256, o (saved forward output) and lse are required, dk / dv are written and dq ACCUMULATED. */
int la_attn_small_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* dout, int ldo,
                      const float* lse, int B, int Nq, int Nk, int heads, int hd, float* dq, float* dk, float* dv, void* stream);

/* Adjoint of la_bilinear (F.interpolate bilinear, align_corners=False): dx[n, ih, iw] += taps * dy[n, oh, ow]; plane strides and row
 * strides explicit so that cropped sources / padded destinations of Lam.postprocess_masks (lam.py:405-449) need no copies. */
int la_bilinear_bwd(const float* dy, int n, int oh, int ow, long dy_plane, int dy_ld, float* dx, int ih, int iw, long dx_plane, int dx_ld,
                    void* stream);

/* The same adjoint for REDUCTIONS (oh <= ih, ow <= iw: the 64 x 64 -> grid resize of a dense mask embedding, prompt_encoder.py:528-540), as a
 * gather: every dx entry is a sum of <= 2 x 2 taps in a fixed order - no atomics, and dx is WRITTEN (no zero-filled destination).
 * la_bilinear_bwd_set_ok says whether a shape qualifies (ih, iw <= 128, oh * ow <= 4096). */
int la_bilinear_bwd_set_ok(int oh, int ow, int ih, int iw);
int la_bilinear_bwd_set(const float* dy, int n, int oh, int ow, long dy_plane, int dy_ld, float* dx, int ih, int iw, long dx_plane, int dx_ld,
                        void* stream);

/* The same resize on NHWC rows, forward and (reductions, la_bilinear_bwd_set_ok) backward: in [N, h * w, C] -> out [N, H * W, C], C % 4 == 0 -
 * the dense mask embedding of the TRAINING graph (prompt_encoder.py:528-540) without plane transposes around the resize; taps and blend
 * are those of la_bilinear per channel.  la_bilinear_rows_bwd_set WRITES dx [n, ih * iw, C] from dy [n, oh * ow, C]. */
int la_bilinear_rows(const float* in, int N, int h, int w, int C, float* out, int H, int W, void* stream);
int la_bilinear_rows_bwd_set(const float* dy, int n, int oh, int ow, int C, float* dx, int ih, int iw, void* stream);

/* Backward of la_classify: dfeat[b, pix, f] (written), dprotos[b, c, f] (ACCUMULATED).  C <= 32, cf in {8, 16, 32, 64}. */
int la_classify_bwd(const float* dseg, const float* feat, const float* protos, int B, int npix, int C, int cf, float* dfeat, float* dprotos,
                    void* stream);

/* out[(g * rep + r), :] = scale * src[g, :]  (backward of the mean over the hw axis, prompt_encoder.py:696-701). */
int la_row_broadcast(const float* src, long groups, int rep, int D, float scale, float* out, void* stream);

/* ---- fused image-side kernels of the TwoWayTransformer (models/transformer.py:255-329): the (groups, hw, D) stream is read once per
 * attention.  8 heads; D = 256 (internal width DI = 128, head width 16: every published LabelAnything decoder) or D = 512 (DI = 256, head
 * width 32: the published SAM-1024 decoder geometry).  Weight planes are fp16 [rows, cols] pairs hi = rn(W), lo = rn(W - hi);
 * the activations are split the same way on the fly, so every projection carries ~21 mantissa bits through the fast MFMA. ---- */

/* The positional encoding enters as a constant table per layer: (img + pe) W^T + b = img W^T + (pe W^T + b); pek / peq = pe W^T + b,
 * fp32 [hw, DI], computed once by the caller and handed over in the kernels' own (accumulator) order: la_twoway_pe_layout(table [hw, DI],
 * hw, DI, out [ceil(hw / 64) * 64 * DI]) rewrites a row-major table once per layer and grid. */
int la_twoway_pe_layout(const float* table, int hw, int DI, float* out, void* stream);

/*
 *
 * tokens -> image attention (cross_attn_token_to_image / final_attn_token_to_image without q_proj / out_proj, which act on the few tokens):
 * out[g, t, :] = softmax_hw(q[g, t] . K_g^T / sqrt(hd)) V_g per head, K = img Wk^T + pek, V = img Wv^T + bv computed tile by tile (64 rows)
 * and never written.  img fp32 [G*hw, D], wk / wv planes [DI, D], q fp32 [G*nt, DI] (projected, bias included),
 * part = scratch fp32 [G * ceil(hw / 64) * 2 * nt * 8 * (2 + hd)], out fp32 [G*nt, DI]. */
int la_twoway_t2i(const float* img, const void* wk_hi, const void* wk_lo, const void* wv_hi, const void* wv_lo, const float* pek,
                  const float* bv, const float* q, int G, int hw, int nt, int D, int heads, float* part, float* out, void* stream);

/* image -> tokens attention + out_proj + residual + LayerNorm (cross_attn_image_to_token + norm4), IN PLACE:
 * img <- LN(img + softmax_t((img Wq^T + peq) . k^T / sqrt(hd)) v Wo^T + bo).  k, v fp32 [G*nt, DI] (projected token keys / values),
 * wq planes [DI, D], wo planes [D, DI], nt <= 32. */
int la_twoway_i2t(float* img, const void* wq_hi, const void* wq_lo, const float* peq, const float* k, const float* v, const void* wo_hi,
                  const void* wo_lo, const float* bo, const float* gamma, const float* beta, float eps, int G, int hw, int nt, int D,
                  int heads, void* stream);

/* ---- image-encoder backward (SURVEY 8f row 1 with a TRAINABLE backbone: parameters/trainval/coco20i/mae_noembs.yaml has no
 * freeze_backbone, so models/lam.py:321-347 hands every ViT parameter to the optimizer).  Plain HF ViT attention, head_dim 64. ---- */

/* la_attn_fwd in LA_ATTN_PLAIN mode that also writes the log2-domain log-sum-exp of every query row: lse fp32 [B*heads, Tpad]
 * (entries t >= T are not touched here; la_attn_bwd overwrites them with +1e30 itself, so that those rows carry zero probability). */
int la_attn_fwd_lse(const void* qkv, const void* vt, void* out16, float* lse, int B, int heads, int T, int Tpad, int E, float scale,
                    int dt, void* stream);

/* dst[(b * heads + h) * 64 + d][t] = src[b * T + t][col0 + h * 64 + d] (16-bit), zero for T <= t < Tpad: the token-contiguous operand
 * copy of la_attn_fwd's V^T form (the backward kernels and la_attn_fwd_rows need none). */
int la_head_transpose(const void* src, int ld, int col0, int B, int heads, int T, int Tpad, void* dst, int dt, void* stream);

/* Gradient of O = softmax(Q K^T scale) V per (image, head) (transformers ViTSelfAttention.forward under build_encoder.py:83-100):
 * qkv [B*T, 3E] (q | k | v), out16 = O, dout16 = dO [B*T, E]; kt / qt / dot: UNUSED since round 5 (may be NULL) - the kernels read K^T,
 * Q^T and dO^T out of the row-major tiles with LDS transpose reads (ds_read_b64_tr_b16), the la_head_transpose copies of rounds 3 - 4
 * are gone; lse from la_attn_fwd_lse;
 * dvec fp32 [B*heads, Tpad] workspace (need not be initialised: receives rowsum(dO * O), and 0 in [T, Tpad), where lse is set to
 * +1e30 as well); dqkv [B*T, 3E] receives dq | dk | dv.  All 16-bit
 * tensors share dt; dO may be pre-scaled (loss scaling), dq / dk / dv then carry the same factor. */
int la_attn_bwd(const void* qkv, const void* out16, const void* dout16, const void* kt, const void* qt, const void* dot, float* lse,
                float* dvec, void* dqkv, int B, int heads, int T, int Tpad, int E, float scale, int dt, void* stream);

/* ---- backward of the SAM ViTDet attention (window / global attention with the decomposed relative-position bias,
 * /root/reference/label_anything/models/image_encoder.py:200-255,340-376; trainable through models/lam.py:321-347 when nothing is frozen) ---- */

/* la_attn_fwd in rel-pos mode with the terms from la_relpos_terms (relh, relw fp32 [B*heads, T, G], T == G*G, G <= 64) that also writes
 * the log2-domain log-sum-exp of every query row like la_attn_fwd_lse: the saved-activation forward of a trainable SAM block. */
int la_attn_fwd_relpos_lse(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw, float* lse, int B, int heads,
                           int T, int Tpad, int G, int E, float scale, int dt, void* stream);

/* la_attn_bwd with S = scale q.k + relh[q][key / G] + relw[q][key % G] (G <= 32, or G == 64): the same outputs, plus
 * drelh[q][kh] = sum_kw dS[q][(kh, kw)] and drelw[q][kw] = sum_kh dS[q][(kh, kw)] (fp32 [B*heads, T, G], every entry written). */
int la_attn_bwd_relpos(const void* qkv, const void* out16, const void* dout16, const void* kt, const void* qt, const void* dot, float* lse,
                       float* dvec, void* dqkv, const float* relh, const float* relw, float* drelh, float* drelw, int B, int heads, int T,
                       int Tpad, int G, int E, float scale, int dt, void* stream);

/* Backward of the rel-pos terms themselves (relh[q][kh] = q . Rh[qy - kh + G - 1], relw[q][kw] = q . Rw[qx - kw + G - 1], q unscaled;
 * image_encoder.py:340-376): dq rows of dqkv += the terms' share (read-modify-write of the 16-bit rows la_attn_bwd_relpos wrote);
 * dtabh / dtabw fp32 [(2G - 1), hd] += gscale * table gradients (atomics over images, heads and rows).  tabh / tabw fp32 [(2G - 1), hd];
 * hd = E / heads = 64 or 128 (other head widths zero-padded by the caller, like the forward). */
int la_relpos_bwd(const void* qkv, void* dqkv, const float* drelh, const float* drelw, const float* tabh, const float* tabw, float* dtabh,
                  float* dtabw, int B, int heads, int G, int E, float gscale, int dt, void* stream);

/* dst = scale * src, n contiguous elements, between LA_F32 and LA_F16 / LA_BF16 (either direction) or LA_F32 -> LA_F32 (in place allowed). */
int la_cast(const void* src, int src_dt, void* dst, int dst_dt, long n, float scale, void* stream);

/* dst[c][r] = src[r][c] (fp32 or 16-bit [R, ld] -> 16-bit [C, Rp], zero for R <= r < Rp, Rp % 64 == 0): the token-contiguous operands of
 * a split-K weight-gradient la_gemm (LaGemmEpilogue.ksplit).  colsum != NULL (fp32 [C]): colsum[c] += sum_r dst[c][r] (atomics) - the
 * bias gradient db = colsum(dY) of an nn.Linear from the same pass over dY that prepares its weight gradient. */
int la_transpose16(const void* src, int src_dt, int ld, int R, int C, void* dst, int dst_dt, int Rp, float* colsum, void* stream);

/* dW[N, K] += dY[R, N]^T . X[R, K] on the 16-bit MFMA from ROW-major 16-bit operands (no transposed copies: both MFMA operands are read out of
 * the row-major LDS tiles with ds_read_b64_tr_b16), split over R with fp32 atomics; db != NULL (fp32 [N]): db[n] += sum_r dY[r][n] from the
 * same pass.  The weight / bias gradient of nn.Linear when the backbone trains (models/common.py:19-37, image_encoder.py:200-255,
 * transformers ViT layers under build_encoder.py:83-100; lam.py:321-347).  N % 256 == 0, K % 256 == 0, ldy / ldx in elements (% 8 == 0).
 * gsize > 0: output row n is written to dW row (n / gsize) * gstride + n % gsize (HF's separate query / key / value weights, a fixed
 * number of rows apart in the flat gradient buffer, from ONE product - la_gemm's LA_MAP_GROUP). */
int la_gemm_tn16(const void* dy, int ldy, const void* x, int ldx, float* dw, int lddw, int R, int N, int K, int gsize, int gstride,
                 float* db, int dt, void* stream);

/* y += a * x, n contiguous fp32 elements (loss-scaled encoder gradients folded into the flat gradient buffer). */
int la_axpy(const float* x, float* y, long n, float a, void* stream);

/* post = GELU(pre) (erf form) on n contiguous 16-bit values, n % 8 == 0: the training forward derives the MLP activation from the saved
 * pre-activation (transformers ViTIntermediate; models/common.py:36-37). */
int la_gelu_fwd16(const void* pre16, void* post16, long n, int dt, void* stream);

/* dpre = dh * gelu'(pre) (erf form, transformers ViTIntermediate): pre 16-bit, dh fp32, outputs fp32 and / or 16-bit (either may be NULL). */
int la_gelu_bwd16(const void* pre16, const float* dh, float* d32, void* d16, long n, int dt, void* stream);

/* ---- fp8 QK^T attention (BASELINE configs[4]; opt-in: Lam(attn_fp8=True), never the parity configuration) ----
 * la_qk_fp8: the q | k columns of qkv16 [rows, 3E] as OCP e4m3 bytes [rows, 2E].
 * la_attn_fwd_fp8: la_attn_fwd in LA_ATTN_PLAIN mode (head_dim 64) with S = Q K^T on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block
 * scales) from those bytes; softmax and P V as in the 16-bit kernel (vt = V^T 16-bit, out 16-bit). */
int la_qk_fp8(const void* qkv, long rows, int E, void* qk8, int dt, void* stream);
int la_attn_fwd_fp8(const void* qk8, const void* vt, void* out16, int B, int heads, int T, int Tpad, int E, float scale, int dt,
                    void* stream);

/* ---- token-mean correction of single-plane weights ("mean planes") ----
 * A weight rounded to one 16-bit plane errs by a_i . W_lo^T on token i; the part of that which survives attention and pooling is the
 * part that is the same for every token of an image: mean_i(a_i) . W_lo^T.  It is added back as a per-image vector: la_colmean16 takes
 * the token means of the 16-bit A operand, a few-row la_gemm forms the vectors, la_layernorm_g / la_add_rowvec fold them into the
 * residual stream.  (image_encoder.py:134-255, build_encoder.py:83-100: same products, ~22-bit weights on the part that matters.) */

/* out[g][c] = mean over the rows_per_group rows of group g of src[row][c] (16-bit [groups * rows_per_group, ld]; out fp32 [groups, D]).
 * Deterministic: fixed 128-row chunks into scratch (fp32 [groups * ceil(rows_per_group / 128) * D]), folded in order - a group's mean
 * does not depend on the other groups of the launch.  wpart > 0: src is window-partitioned (LA_MAP_WINDOW_PART order of an H x W grid,
 * ws = wpart): the rows of group g are gathered in image order, pad slots are skipped. */
int la_colmean16(const void* src, int ld, int groups, int rows_per_group, int D, float* out, float* scratch, int wpart, int H, int W, int dt,
                 void* stream);

/* la_layernorm of x[r] + xg[r / rows_per_group] (xg fp32 [rows / rows_per_group, E]): the pending per-image corrections enter every
 * consumer of the residual stream without being written back.  colsum_part != NULL (fp32 [rows / rows_per_group * ceil(rows_per_group
 * / 32), E]): the pass also leaves the column sums of the rows it stored, per fixed 1 / ceil(rows_per_group / 32) share of a group - the token means of
 * the next GEMM's operand without a second pass over it (la_colsum_fold turns the chunks into means). */
int la_layernorm_g(const float* x, const float* xg, int rows_per_group, int ldx, int rows, int E, const float* gamma, const float* beta,
                   float eps, float* out32, void* out16, int window, int H, int W, float* colsum_part, int dt, void* stream);

/* la_attn_fwd that also writes the column sums of every 128-query block of its 16-bit output: cspart fp32 [B * ceil(T / 128), E]
 * (models/image_encoder.py:225-255 - the attention output is the proj operand whose token means the correction needs).
 * LA_ATTN_RELPOS_WIN16 with csH x csW = the image's token grid: B = images * windows per image, rows of padded window slots are left
 * out of the sums.  Not available on the G <= 16 LA_ATTN_RELPOS window path. */
int la_attn_fwd_cs(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw, const void* tabh, const void* tabw,
                   int B, int heads, int T, int Tpad, int G, int E, float scale, int mode, float* cspart, int csH, int csW, int dt,
                   void* stream);

/* la_attn_fwd / la_attn_fwd_cs WITHOUT a V^T copy: the V tiles are staged row-major from the v columns of qkv, like the K tiles, and reach
 * the MFMA through LDS transpose reads (ds_read_b64_tr_b16) - the q | k | v GEMM keeps its plain row-major epilogue on every column
 * (models/image_encoder.py:225-255: ``qkv = self.qkv(x)`` has no transposed copy either).  Modes: LA_ATTN_PLAIN; LA_ATTN_RELPOS with the
 * 16-bit tables on the 64 x 64 grid; LA_ATTN_RELPOS_WIN16 (G <= 16, Tpad >= 16 G) - with imgH > 0 the windows are addressed in IMAGE
 * order (window_partition / window_unpartition of image_encoder.py:258-304 as address arithmetic): qkv [images * imgH * imgW, 3E] and
 * out16 [images * imgH * imgW, E] hold the image's tokens, B = images * ceil(imgH / G) * ceil(imgW / G) windows, and a window token
 * beyond the image is ``padrow`` (16-bit [3E]: q | k | v of a pad-after-norm token = the qkv bias, image_encoder.py:160-172); its
 * output row does not exist.  cspart (optional) as in la_attn_fwd_cs: fp32 [B * ceil(T / 128), E], padded tokens left out. */
int la_attn_fwd_rows(const void* qkv, void* out16, const void* tabh, const void* tabw, int B, int heads, int T, int Tpad, int G, int E,
                     float scale, int mode, float* cspart, int imgH, int imgW, const void* padrow, int dt, void* stream);

/* out[g * ldo + c] = inv * sum over the chunks j of part[(g * chunks + j) * D + c], added in index order (fp32). */
int la_colsum_fold(const float* part, int groups, int chunks, int D, float inv, float* out, int ldo, void* stream);

/* x[r] += v[r / rows_per_group] in place (fp32 [rows, D]). */
int la_add_rowvec(float* x, const float* v, long rows, int rows_per_group, int D, void* stream);
/* The same pass (v optional: nullptr leaves x as it is) that also writes the result as LA_F16X2 operand rows split16 fp16 [rows, 2 D] =
 * [hi | lo]: the fp32 token stream leaving the block stack becomes the A operand of the SAM neck's 1 x 1 convolution (image_encoder.py:92-108)
 * as three fp16 MFMA products instead of the exact-fp32 MFMA, without another pass over it. */
int la_add_rowvec_split(float* x, const float* v, long rows, int rows_per_group, int D, void* split16, void* stream);

/* ---- LayerNorm folded into its neighbour GEMMs (LaGemmEpilogue.nstat_out / nstat_in; image_encoder.py:181-197) --------------------
 * la_norm_finalize: the row statistics a consumer GEMM's epilogue applies, and the token means the V correction needs.
 *   part != NULL (fp32 [M][nslots][2], written by producer GEMMs): mr[row] = (mean, rstd) with mean = sum_s part[row][s][0] / E,
 *     var = sum_s part[row][s][1] / E - mean^2 (biased, like nn.LayerNorm), rstd = 1 / sqrt(var + eps); slots added in index order.
 *   part == NULL: mr is an input (la_norm_stats wrote it).
 *   cs_part != NULL (fp32 [M / rows_per_group * ceil(rows_per_group / 128), E]; needs x16, 16-bit [M, E], row stride ld16): per 128-row
 *     chunk of a group the column sums of rstd[row] * (x16[row] - mean[row]) - the normalised rows BEFORE gamma / beta, which live in the
 *     folded weights - for la_colsum_fold (the token means of the qkv operand: LamEngine "vmean").  M % rows_per_group == 0.
 *   mr has ceil(M / 256) * 256 rows; rows beyond M are written as (0, 0). */
int la_norm_finalize(const float* part, int M, int nslots, int E, float eps, float* mr, const void* x16, int ld16, int rows_per_group,
                     float* cs_part, int dt, void* stream);
/* la_norm_stats: the same statistics from the fp32 stream itself, one pass: x16 = 16-bit rounding of x, mr[row] = (mean, rstd)
 * (two-pass variance on the row in registers).  The entry of a block stack whose first rows do not come from a producer GEMM
 * (HF ViT: CLS row + patch embedding through a row map). */
int la_norm_stats(const float* x, int ldx, int M, int E, float eps, void* x16, float* mr, int dt, void* stream);

#ifdef __cplusplus
}
#endif
#endif
