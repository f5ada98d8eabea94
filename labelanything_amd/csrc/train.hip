// Backward kernels of the decoder-side training step: everything the forward kernels of dec.hip / gemm.hip /
// norm.hip / post.hip need to be differentiated - fp32 throughout, like the decoder forward.
//   la_gemm_tn          dW[N,K] += dY[M,N]^T . X[M,K]      (weight gradient of every nn.Linear / 1x1 / k=s conv: fp32 MFMA 32x32x2, operands
//                                                            straight from row-major global memory, split over M with atomic accumulation)
//   la_layernorm_bwd    LayerNorm / LayerNorm2d (+ fused GELU) backward: dx, dgamma, dbeta
//   la_act_fwd / _bwd   GELU(erf) / ReLU
//   la_attn_small_lse / la_attn_small_bwd   softmax attention backward of the decoder attentions (one side is a handful of tokens)
//   la_bilinear_bwd     adjoint of F.interpolate(mode="bilinear", align_corners=False) (post-processing, mask resize)
//   la_classify_bwd     prototype classification backward
//   la_row_broadcast    backward of the mean over the hw axis
// Reference: the autograd graph of label_anything/models/{common,transformer,prompt_encoder,mask_decoder,lam}.py under
// experiment/utils.py:266-303 (WrapperModule) + loss/__init__.py:67-89.
#include <cstdlib>
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

// ---------------------------------------------------------------------------------------------------------------------------
// dW[n][k] += sum_m dY[m][n] X[m][k].  The 32x32x2 fp32 MFMA takes A[i][kk] / B[kk][j] with lane l holding (i or j = l & 31,
// kk = l >> 5): for a reduction over ROWS of two row-major matrices that is one coalesced 128-byte read of row m + (l >> 5) per
// half-wave and operand - no LDS, no transposes.  Workgroup = 4 waves (2 x 2) of 64 x 64 -> a 128 x 128 tile of dW over one
// M-chunk; chunks are folded with fp32 atomics (the summation order is therefore not fixed; 1e-7-level run-to-run noise).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                      float* __restrict__ dw, int ldw, int M, int N, int K, int mchunk, int tiles_k,
                                                      float* __restrict__ db) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
  const int n0 = tn * 128 + (wave >> 1) * 64, k0 = tk * 128 + (wave & 1) * 64;
  const int mbeg = blockIdx.y * mchunk, mend = min(M, mbeg + mchunk);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // clamped column indices + validity masks (loads never fault, invalid lanes contribute zeros)
  int cn[2], ck[2];
  float vn[2], vk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + i * 32 + fr, k = k0 + i * 32 + fr;
    vn[i] = n < N ? 1.f : 0.f;
    vk[i] = k < K ? 1.f : 0.f;
    cn[i] = min(n, N - 1);
    ck[i] = min(k, K - 1);
  }
  // software pipeline: the 16 loads of the next 8 rows are in flight while the 16 MFMAs of the current 8 rows issue (with one
  // iteration in flight the loop ran at one memory latency per 8 rows)
  auto fetch = [&](int m, float (&a)[4][2], float (&b)[4][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = m + 2 * u + fh;
      const float vr = row < mend ? 1.f : 0.f;
      const size_t rc = (size_t)min(row, M - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float ta = dy[rc * ldy + cn[i]], tb = x[rc * ldx + ck[i]];
        a[u][i] = (vr * vn[i] != 0.f) ? ta : 0.f;      // (select: 0 * Inf would poison the masked lanes)
        b[u][i] = (vr * vk[i] != 0.f) ? tb : 0.f;
      }
    }
  };
  // db != nullptr: the bias gradient db[n] += sum_m dY[m][n] on the way - the waves of the first k-tile column that own a dY column
  // block (wave & 1 == 0) add up the operands they load anyway
  const bool do_db = db != nullptr && tk == 0 && (wave & 1) == 0;
  float bsum[2] = {0.f, 0.f};
  auto mma = [&](const float (&a)[4][2], const float (&b)[4][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (do_db) bsum[i] += a[u][i];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
      }
  };
  float a0[4][2], b0[4][2], a1[4][2], b1[4][2];
  fetch(mbeg, a0, b0);
  for (int m = mbeg; m < mend; m += 16) {
    fetch(m + 8, a1, b1);                            // (rows >= mend contribute zeros)
    mma(a0, b0);
    fetch(m + 16, a0, b0);
    mma(a1, b1);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j * 32 + fr;
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (n < N) atomicAdd(&dw[(size_t)n * ldw + k], acc[i][j][r]);
      }
    }
  if (do_db) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + i * 32 + fr;                // (both half waves hold partial sums of column n: rows m + 2 u + fh)
      if (n < N) atomicAdd(&db[n], bsum[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same product for long reductions (M >= 4096, 16-byte aligned operands), with the rows staged through LDS by LDS-DMA.  The
// register-fed loop above keeps ONE 8-row fetch in flight beside 16 MFMAs (1024 cycles): every iteration waits out the rest of a
// 2 - 4 k-cycle memory latency, and the matrix pipe is busy 30 % of the time (270000 x 128 x 256: 377 us against 113 us of fp32 MFMA
// issue).  Here a ring of NS stages of R rows ([R][TN] of dY | [R][TK] of X, row-major fp32, filled by 1 KiB pieces of
// global_load_lds_dwordx4: each wave stages R / 4 rows of both operands) runs NS - 1 stages ahead of the MFMAs without holding a
// register; the fragments are 4-byte LDS reads (lane = column, half-wave = row parity).
//   WN x WK waves over the (n, k) tile, SN x SK 32 x 32 sub-tiles per wave; the remaining 4 / (WN WK) waves split the rows of a stage
//   (every wave adds its partial tile with atomics at the end, as the M-chunks do anyway).  Shapes: 128 x 128 (2 x 2 waves of 64 x 64),
//   256 x 32 / 32 x 256 for one narrow side (mask_downscaling / output_upscaling: 16 channels against 256), 32 x 32 split four ways over
//   the rows when both sides are narrow.
// ---------------------------------------------------------------------------------------------------------------------------
template <int WN, int WK, int SN, int SK, int R, int NS>
__global__ __launch_bounds__(256) void gemm_tn_dma_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                          float* __restrict__ dw, int ldw, int M, int N, int K, int mchunk, int tiles_k,
                                                          float* __restrict__ db) {
  constexpr int WM = 4 / (WN * WK), TN = WN * SN * 32, TK = WK * SK * 32, RW = R / 4;
  constexpr int PN = RW * TN * 4 / 1024, PK = RW * TK * 4 / 1024;          // pieces per wave and stage
  static_assert(WM * WN * WK == 4 && PN >= 1 && PK >= 1 && (RW * TN * 4) % 1024 == 0 && (RW * TK * 4) % 1024 == 0, "piece layout");
  static_assert((R / 2) % WM == 0 && (NS - 2) * (PN + PK) < 64, "stage layout");
  constexpr int STAGE = R * (TN + TK) * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
  const int n0 = tn * TN, k0 = tk * TK;
  const int mbeg = blockIdx.y * mchunk, mend = min(M, mbeg + mchunk);
  const unsigned lds0 = lds_addr_of(smem);

  f32x16 acc[SN][SK];
#pragma unroll
  for (int i = 0; i < SN; ++i)
#pragma unroll
    for (int j = 0; j < SK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float vn[SN], vk[SK];
#pragma unroll
  for (int i = 0; i < SN; ++i) vn[i] = n0 + (wn * SN + i) * 32 + fr < N ? 1.f : 0.f;
#pragma unroll
  for (int j = 0; j < SK; ++j) vk[j] = k0 + (wk * SK + j) * 32 + fr < K ? 1.f : 0.f;

  // stage `st` of this workgroup's rows -> ring slot: rows clamped to M - 1, 4-column groups clamped to the last group of the matrix
  // (N, K are multiples of 4: a group is valid or masked as a whole; masked columns / rows are multiplied by zero when they are read)
  auto issue = [&](int st, int slot) {
    const int m = mbeg + st * R + wave * RW;
    const unsigned base = lds0 + slot * STAGE;
#pragma unroll
    for (int p = 0; p < PN; ++p) {
      const int o = p * 1024 + lane * 16, row = o / (TN * 4), col = min(n0 + (o % (TN * 4)) / 4, N - 4);
      dma16(dy + (size_t)min(m + row, M - 1) * ldy + col, base + wave * (RW * TN * 4) + p * 1024);
    }
#pragma unroll
    for (int p = 0; p < PK; ++p) {
      const int o = p * 1024 + lane * 16, row = o / (TK * 4), col = min(k0 + (o % (TK * 4)) / 4, K - 4);
      dma16(x + (size_t)min(m + row, M - 1) * ldx + col, base + R * TN * 4 + wave * (RW * TK * 4) + p * 1024);
    }
  };
  const bool do_db = db != nullptr && tk == 0 && wk == 0;
  float bsum[SN];
#pragma unroll
  for (int i = 0; i < SN; ++i) bsum[i] = 0.f;

  const int nstages = (mend - mbeg + R - 1) / R;
#pragma unroll
  for (int st = 0; st < NS - 1; ++st) issue(st, st);           // (stages beyond the chunk re-read clamped rows: every wave issues the same count)
  for (int it = 0; it < nstages; ++it) {
    dma_wait<(NS - 2) * (PN + PK)>();                          // this wave's pieces of stage `it` have landed ...
    __syncthreads();                                           // ... so have everyone's, and everyone is done reading stage it - 1
    issue(it + NS - 1, (it + NS - 1) % NS);
    const float* sdy = reinterpret_cast<const float*>(smem + (it % NS) * STAGE);
    const float* sx = sdy + R * TN;
    const int mrow = mbeg + it * R;
#pragma unroll
    for (int kk = 0; kk < R / 2 / WM; ++kk) {
      const int row = 2 * (kk * WM + wm) + fh;
      const float vr = mrow + row < mend ? 1.f : 0.f;
      float a[SN], b[SK];
#pragma unroll
      for (int i = 0; i < SN; ++i) {      // select, not multiply: an Inf / NaN in a clamped row must not reach the masked lanes (0 * Inf)
        const float t = sdy[row * TN + (wn * SN + i) * 32 + fr];
        a[i] = (vr * vn[i] != 0.f) ? t : 0.f;
      }
#pragma unroll
      for (int j = 0; j < SK; ++j) {
        const float t = sx[row * TK + (wk * SK + j) * 32 + fr];
        b[j] = (vk[j] != 0.f) ? t : 0.f;
      }
#pragma unroll
      for (int i = 0; i < SN; ++i) {
        if (do_db) bsum[i] += a[i];
#pragma unroll
        for (int j = 0; j < SK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  dma_wait<0>();                                               // (the look-ahead pieces of stages that do not exist)
#pragma unroll
  for (int i = 0; i < SN; ++i)
#pragma unroll
    for (int j = 0; j < SK; ++j) {
      const int k = k0 + (wk * SK + j) * 32 + fr;
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * SN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (n < N) atomicAdd(&dw[(size_t)n * ldw + k], acc[i][j][r]);
      }
    }
  if (do_db) {
#pragma unroll
    for (int i = 0; i < SN; ++i) {
      const int n = n0 + (wn * SN + i) * 32 + fr;
      if (n < N) atomicAdd(&db[n], bsum[i]);
    }
  }
}

template <int WN, int WK, int SN, int SK, int R, int NS>
static void launch_gemm_tn_dma(const float* dy, int ldy, const float* x, int ldx, float* dw, int ldw, int M, int N, int K, float* db,
                               int target_wgs, hipStream_t st) {
  constexpr int TN = WN * SN * 32, TK = WK * SK * 32, LDS = NS * R * (TN + TK) * 4;
  const int tiles_n = (N + TN - 1) / TN, tiles_k = (K + TK - 1) / TK;
  int chunks = target_wgs / (tiles_n * tiles_k);
  if (chunks < 1) chunks = 1;
  int mchunk = (M + chunks - 1) / chunks;
  if (mchunk < 8 * R) mchunk = 8 * R;                          // at least 8 stages behind every atomic epilogue
  mchunk = (mchunk + R - 1) / R * R;
  chunks = (M + mchunk - 1) / mchunk;
  static unsigned long long mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_tn_dma_kernel<WN, WK, SN, SK, R, NS>), LDS, mask);
  hipLaunchKernelGGL((gemm_tn_dma_kernel<WN, WK, SN, SK, R, NS>), dim3(tiles_n * tiles_k, chunks), dim3(256), LDS, st, dy, ldy, x, ldx, dw,
                     ldw, M, N, K, mchunk, tiles_k, db);
}

// ---------------------------------------------------------------------------------------------------------------------------
// out[n] += sum_m dY[m][n]: the bias gradient of every linear / conv layer.  (As a product with a ones vector through the 128 x 128
// MFMA tiles above it cost as much as the weight gradient itself: 5 ms of a 41 ms training step.)  Threads are laid out
// (rows_par x NW) with NW = the column count rounded up to a power of two (<= 256): coalesced row reads, per-thread partial sums
// over a strided row set, LDS fold over the row lanes, one atomic per column and workgroup.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, int ldy, long M, int N, int nw, long rows_per_block,
                                                     float* __restrict__ out) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int c = tid % nw, ty = tid / nw, rp = 256 / nw;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int cb = 0; cb < N; cb += nw) {
    const int col = cb + c;
    float acc = 0.f;
    if (col < N) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // four rows in flight per thread
      long r = r0 + ty;
      for (; r + 3L * rp < r1; r += 4L * rp) {
        a0 += dy[(size_t)r * ldy + col];
        a1 += dy[(size_t)(r + rp) * ldy + col];
        a2 += dy[(size_t)(r + 2L * rp) * ldy + col];
        a3 += dy[(size_t)(r + 3L * rp) * ldy + col];
      }
      for (; r < r1; r += rp) a0 += dy[(size_t)r * ldy + col];
      acc = (a0 + a1) + (a2 + a3);
    }
    red[tid] = acc;
    __syncthreads();
    for (int s2 = rp >> 1; s2 > 0; s2 >>= 1) {
      if (ty < s2) red[tid] += red[tid + s2 * nw];
      __syncthreads();
    }
    if (ty == 0 && col < N) atomicAdd(&out[col], red[tid]);
    __syncthreads();
  }
}

// dW[n][k] += sum_m dY[m][n] X[m][k] for tiny outputs (N, K <= 8: e.g. the 4 x 4 taps of a k = s conv over millions of pixels): every
// thread keeps the whole N x K block over a strided row set, DPP row sums, one atomic per entry and 16-lane row.
template <int NN, int KK>
__global__ __launch_bounds__(256) void gemm_tn_tiny_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ dw, int ldw, long M, int N, int K) {
  float acc[NN][KK];
#pragma unroll
  for (int n = 0; n < NN; ++n)
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[n][k] = 0.f;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    float a[NN], b[KK];
#pragma unroll
    for (int n = 0; n < NN; ++n) a[n] = n < N ? dy[(size_t)r * ldy + n] : 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k) b[k] = k < K ? x[(size_t)r * ldx + k] : 0.f;
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
      for (int k = 0; k < KK; ++k) acc[n][k] = fmaf(a[n], b[k], acc[n][k]);
  }
  __shared__ float red[16][NN * KK];
#pragma unroll
  for (int n = 0; n < NN; ++n)
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      const float v = row16_sum(acc[n][k]);
      if ((threadIdx.x & 15) == 0) red[threadIdx.x >> 4][n * KK + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < NN * KK) {
    const int n = threadIdx.x / KK, k = threadIdx.x % KK;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += red[g][threadIdx.x];
    if (n < N && k < K) atomicAdd(&dw[(size_t)n * ldw + k], v);      // one atomic per entry and workgroup
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (E <= 2048): y = [GELU](xhat * gamma + beta), xhat = (x - mu) * rstd.
//   dz = dy [* GELU'(z)];  dgamma += dz * xhat;  dbeta += dz;  g = dz * gamma;
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
// dgamma / dbeta: every lane keeps the partial sums of its columns over all the rows its wave walks, one atomic per column and
// wave at the end.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_grad(float z) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  return cdf + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
}

template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long rows, int E,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            int gelu, float* dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, const float* add, void* out16, int dt16) {
  // add != nullptr: dx = LN backward + add (the gradient that arrives over the block's skip connection; add may be dx itself);
  // out16 != nullptr: a 16-bit copy of dx on the way (the operand of the next backward GEMM)
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  float gm[VPL], bt[VPL], sg[VPL], sb[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    gm[v] = c < E ? gamma[c] : 0.f;
    bt[v] = c < E ? beta[c] : 0.f;
    sg[v] = sb[v] = 0.f;
  }
  const float inv_e = 1.0f / (float)E;
  for (long r = wave; r < rows; r += nwaves) {
    float xv[VPL], dv[VPL];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      xv[v] = c < E ? x[r * E + c] : 0.f;
      dv[v] = c < E ? dy[r * E + c] : 0.f;
      s += xv[v];
    }
    const float mu = wave_sum_dpp(s) * inv_e;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      const float d = c < E ? xv[v] - mu : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum_dpp(q) * inv_e + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      const float xh = c < E ? (xv[v] - mu) * rstd : 0.f;
      float dz = dv[v];
      if (gelu) dz *= gelu_grad(xh * gm[v] + bt[v]);
      sg[v] += dz * xh;
      sb[v] += dz;
      const float g = dz * gm[v];
      s1 += g;
      s2 += g * xh;
      xv[v] = xh;
      dv[v] = g;
    }
    s1 = wave_sum_dpp(s1) * inv_e;
    s2 = wave_sum_dpp(s2) * inv_e;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      if (c < E) {
        float o = rstd * (dv[v] - s1 - xv[v] * s2);
        if (add) o += add[r * E + c];
        dx[r * E + c] = o;
        if (out16) {
          if (dt16 == LA_F16) reinterpret_cast<f16_t*>(out16)[r * E + c] = (f16_t)o;
          else reinterpret_cast<bf16_t*>(out16)[r * E + c] = (bf16_t)o;
        }
      }
    }
  }
  // the four waves of the workgroup meet in LDS first: a quarter of the atomics (8192 waves adding to the same 2 E addresses
  // serialise in L2 - a third of the kernel's time at E = 768 before)
  __shared__ float red[2][4][64 * VPL];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    red[0][wv][lane + 64 * v] = sg[v];
    red[1][wv][lane + 64 * v] = sb[v];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += 256) {
    atomicAdd(&dgamma[c], (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
    atomicAdd(&dbeta[c], (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
  }
}

// The same pass for E % 256 == 0 (the encoder widths: 768 / 1024 / 1280, the decoder's 256 / 512) with 16-byte accesses: a lane owns four
// consecutive columns per vector (V4 = E / 256 vectors) - float4 loads of x / dy / add, float4 stores of dx, 8-byte stores of the 16-bit
// copy - instead of 4-byte loads and 2-byte stores at a 64-column stride (46852 x 768: 171 us for 0.47 GB of traffic, round 6).
template <int V4>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy, long rows, int E,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                int gelu, float* dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                const float* add, void* out16, int dt16) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  float4 gm[V4], bt[V4], sg[V4], sb[V4];
#pragma unroll
  for (int v = 0; v < V4; ++v) {
    gm[v] = reinterpret_cast<const float4*>(gamma)[lane + 64 * v];
    bt[v] = reinterpret_cast<const float4*>(beta)[lane + 64 * v];
    sg[v] = sb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_e = 1.0f / (float)E;
  const int e4 = E >> 2;
  for (long r = wave; r < rows; r += nwaves) {
    float4 xv[V4], dv[V4];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      xv[v] = reinterpret_cast<const float4*>(x)[r * e4 + lane + 64 * v];
      dv[v] = reinterpret_cast<const float4*>(dy)[r * e4 + lane + 64 * v];
      s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
    }
    const float mu = wave_sum_dpp(s) * inv_e;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const float a = xv[v].x - mu, b = xv[v].y - mu, c = xv[v].z - mu, d = xv[v].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum_dpp(q) * inv_e + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      float* xp = reinterpret_cast<float*>(&xv[v]);
      float* dp = reinterpret_cast<float*>(&dv[v]);
      const float* gp = reinterpret_cast<const float*>(&gm[v]);
      const float* bp = reinterpret_cast<const float*>(&bt[v]);
      float* sgp = reinterpret_cast<float*>(&sg[v]);
      float* sbp = reinterpret_cast<float*>(&sb[v]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xp[k] - mu) * rstd;
        float dz = dp[k];
        if (gelu) dz *= gelu_grad(xh * gp[k] + bp[k]);
        sgp[k] += dz * xh;
        sbp[k] += dz;
        const float g = dz * gp[k];
        s1 += g;
        s2 += g * xh;
        xp[k] = xh;
        dp[k] = g;
      }
    }
    s1 = wave_sum_dpp(s1) * inv_e;
    s2 = wave_sum_dpp(s2) * inv_e;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      float4 o;
      o.x = rstd * (dv[v].x - s1 - xv[v].x * s2);
      o.y = rstd * (dv[v].y - s1 - xv[v].y * s2);
      o.z = rstd * (dv[v].z - s1 - xv[v].z * s2);
      o.w = rstd * (dv[v].w - s1 - xv[v].w * s2);
      if (add) {
        const float4 a = reinterpret_cast<const float4*>(add)[r * e4 + lane + 64 * v];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      reinterpret_cast<float4*>(dx)[r * e4 + lane + 64 * v] = o;
      if (out16) {
        uint2 pk;
        if (dt16 == LA_F16) { pk.x = pack2<f16_t>(o.x, o.y); pk.y = pack2<f16_t>(o.z, o.w); }
        else { pk.x = pack2<bf16_t>(o.x, o.y); pk.y = pack2<bf16_t>(o.z, o.w); }
        reinterpret_cast<uint2*>(out16)[r * e4 + lane + 64 * v] = pk;
      }
    }
  }
  __shared__ float4 red[2][4][64 * V4];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < V4; ++v) {
    red[0][wv][lane + 64 * v] = sg[v];
    red[1][wv][lane + 64 * v] = sb[v];
  }
  __syncthreads();
  const float* r0 = reinterpret_cast<const float*>(red[0]);
  const float* r1 = reinterpret_cast<const float*>(red[1]);
  for (int c = threadIdx.x; c < E; c += 256) {
    atomicAdd(&dgamma[c], (r0[c] + r0[E + c]) + (r0[2 * E + c] + r0[3 * E + c]));
    atomicAdd(&dbeta[c], (r1[c] + r1[E + c]) + (r1[2 * E + c] + r1[3 * E + c]));
  }
}

// LayerNorm2d over a handful of channels (E = 4 / 16 / 32 of the mask_downscaling and upscaling stacks, millions of pixels): one
// THREAD per row - a wave per 4-element row left 60 lanes idle (2.4 M rows of 4: 0.73 ms).  Same formulas as above; dgamma / dbeta
// partials are folded over the wave with DPP, one atomic per channel and wave.
template <int EMAX>
__global__ __launch_bounds__(256) void layernorm_bwd_narrow_kernel(const float* __restrict__ x, const float* __restrict__ dy, long rows, int E,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                   int gelu, float* __restrict__ dx, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta) {
  float gm[EMAX], bt[EMAX], sg[EMAX], sb[EMAX];
#pragma unroll
  for (int c = 0; c < EMAX; ++c) {
    gm[c] = c < E ? gamma[c] : 0.f;
    bt[c] = c < E ? beta[c] : 0.f;
    sg[c] = sb[c] = 0.f;
  }
  const float inv_e = 1.0f / (float)E;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    float xv[EMAX], dv[EMAX];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < EMAX; ++c) {
      xv[c] = c < E ? x[r * E + c] : 0.f;
      dv[c] = c < E ? dy[r * E + c] : 0.f;
      s += xv[c];
    }
    const float mu = s * inv_e;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < EMAX; ++c) {
      const float d = c < E ? xv[c] - mu : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(q * inv_e + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < EMAX; ++c) {
      const float xh = c < E ? (xv[c] - mu) * rstd : 0.f;
      float dz = dv[c];
      if (gelu) dz *= gelu_grad(xh * gm[c] + bt[c]);
      sg[c] += dz * xh;
      sb[c] += dz;
      const float g = dz * gm[c];
      s1 += g;
      s2 += g * xh;
      xv[c] = xh;
      dv[c] = g;
    }
    s1 *= inv_e;
    s2 *= inv_e;
#pragma unroll
    for (int c = 0; c < EMAX; ++c)
      if (c < E) dx[r * E + c] = rstd * (dv[c] - s1 - xv[c] * s2);
  }
  // wave sums -> LDS -> ONE atomic per channel and workgroup (per wave, 16 k atomics landed on each of the 2 E addresses and
  // serialised in L2: 3.4 ms of a 0.1 ms kernel on 1.2 M rows of 16 channels)
  __shared__ float red[2][4][EMAX];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < EMAX; ++c) {
    const float a = wave_sum_dpp(sg[c]), b = wave_sum_dpp(sb[c]);
    if ((threadIdx.x & 63) == 0) {
      red[0][wv][c] = a;
      red[1][wv][c] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < E) {
    const int c = threadIdx.x;
    atomicAdd(&dgamma[c], (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
    atomicAdd(&dbeta[c], (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int kind) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = kind == LA_ACT_GELU ? gelu_erf(v) : fmaxf(v, 0.f);
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n,
                                                      int kind) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    dx[i] = dy[i] * (kind == LA_ACT_GELU ? gelu_grad(v) : (v > 0.f ? 1.f : 0.f));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Attention backward.  Layout as la_attn_small: q [B, Nq, ldq] (head h at column h * HD), k / v [B, Nk, ld], o / do [B, Nq, ldo],
// scores = q.k * scale.  One side of every decoder attention is a handful of tokens:
//   few keys  (Nk <= 256): one thread per (b, query, head); the softmax is local to the thread (three passes over the keys,
//                          nothing stored); dq is written directly, dk / dv are reduced over the wave and added atomically.
//   few queries (Nq <= 256): one thread per (b, key, head); needs the row statistics lse[b, head, q] (la_attn_small_lse) and the
//                          saved output o (delta = do . o); dk / dv are written directly, dq is reduced over the wave.
// ---------------------------------------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const float *q, *k, *v, *o, *dout, *lse;
  float *dq, *dk, *dv;
  int ldq, ldk, ldv, ldo;        // dq / dk / dv use the same leading dimensions as q / k / v
  int B, Nq, Nk, heads;
  float scale;
  int vec;                       // every pointer 16-byte aligned and every leading dimension a multiple of 4: rows move as float4
};

// one thread's head slice of a row (HDIM consecutive floats; rows of neighbouring threads are a leading dimension apart, so a 4-byte
// access touches 64 cache lines per instruction: float4 quarters the instruction count)
template <int HDIM> __device__ __forceinline__ void load_row(const float* p, float (&v)[HDIM], bool vec) {
  if (vec) {
#pragma unroll
    for (int d = 0; d < HDIM; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + d);
      v[d] = t.x; v[d + 1] = t.y; v[d + 2] = t.z; v[d + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int d = 0; d < HDIM; ++d) v[d] = p[d];
  }
}
template <int HDIM> __device__ __forceinline__ void store_row(float* p, const float (&v)[HDIM], bool vec) {
  if (vec) {
#pragma unroll
    for (int d = 0; d < HDIM; d += 4) *reinterpret_cast<float4*>(p + d) = make_float4(v[d], v[d + 1], v[d + 2], v[d + 3]);
  } else {
#pragma unroll
    for (int d = 0; d < HDIM; ++d) p[d] = v[d];
  }
}

template <int HDIM>
__global__ __launch_bounds__(256) void attn_lse_kernel(AttnBwdArgs a, float* __restrict__ lse) {
  // one workgroup per (b, q, head): lse = log sum_j exp(q.k_j scale)
  __shared__ float red[2][4];
  const int i = blockIdx.x;
  const int h = i % a.heads, bq = i / a.heads, b = bq / a.Nq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float qv[HDIM];
#pragma unroll
  for (int d = 0; d < HDIM; ++d) qv[d] = a.q[(size_t)bq * a.ldq + h * HDIM + d] * a.scale;
  const float* kp = a.k + (size_t)b * a.Nk * a.ldk + h * HDIM;
  float m = -3.0e38f, l = 0.f;
  for (int j = tid; j < a.Nk; j += 256) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HDIM; ++d) s += qv[d] * kp[(size_t)j * a.ldk + d];
    const float mn = fmaxf(m, s);
    l = l * expf(m - mn) + expf(s - mn);
    m = mn;
  }
  const float mw = wave_max(m);
  l = wave_sum_dpp(l * expf(m - mw));
  if (lane == 0) {
    red[0][wave] = mw;
    red[1][wave] = l;
  }
  __syncthreads();
  if (tid == 0) {
    const float mg = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float lt = 0.f;
    for (int w = 0; w < 4; ++w) lt += red[1][w] * expf(red[0][w] - mg);
    lse[i] = mg + logf(lt);       // index (b * Nq + q) * heads + h
  }
}

// Backward of the decoder attentions.  One workgroup = up to QW rows of the LONG side of one (group, head), one row per thread with its
// q / dO (or k / v) vectors in registers; the rows of the short side are the same for every thread of the workgroup (uniform addresses:
// scalar loads).  The gradients of the short side are sums over the threads: per chunk of 8 short-side rows every thread leaves its
// dS (and P) values in LDS, then the workgroup re-partitions - thread = (short row, channel) - and adds up its column over the
// workgroup's rows with the q / dO (k) vectors that were parked in LDS at the start: one atomic per output element and workgroup.
// (Round 3 reduced every (short row, channel) pair over the wave with DPP and added it with a single-lane atomic: 4800 dependent
// reductions per wave for the 150 x 150 class-example attention, 0.73 ms for 38 waves of work.)
template <int HDIM> struct AttnSmallCfg {
  static constexpr int QW = HDIM <= 16 ? 256 : HDIM <= 32 ? 128 : 64;      // threads = long-side rows per workgroup (32 KiB of parked vectors)
  static constexpr int JC = 8;                                             // short-side rows per chunk (static LDS <= 64 KiB)
};

template <int HDIM>
__global__ __launch_bounds__(AttnSmallCfg<HDIM>::QW) void attn_bwd_fewkeys_kernel(AttnBwdArgs a) {
  constexpr int QW = AttnSmallCfg<HDIM>::QW, JC = AttnSmallCfg<HDIM>::JC;
  __shared__ float sq[QW][HDIM + 1], sdo[QW][HDIM + 1];
  __shared__ float sds[JC][QW], sp[JC][QW];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int qi = blockIdx.y * QW + tid;
  const bool live = qi < a.Nq;
  const long bq = (long)b * a.Nq + min(qi, a.Nq - 1);
  float qv[HDIM], dov[HDIM], dqv[HDIM];
  const float lv = live ? 1.f : 0.f;           // (threads beyond Nq carry dO = 0: dS = 0 and P dO = 0, no masks further down)
  load_row<HDIM>(a.q + bq * a.ldq + h * HDIM, qv, a.vec);
  load_row<HDIM>(a.dout + bq * a.ldo + h * HDIM, dov, a.vec);
#pragma unroll
  for (int d = 0; d < HDIM; ++d) {
    dov[d] *= lv;
    dqv[d] = 0.f;
    sq[tid][d] = qv[d];
    sdo[tid][d] = dov[d];
  }
  const float* kp = a.k + (size_t)b * a.Nk * a.ldk + h * HDIM;
  const float* vp = a.v + (size_t)b * a.Nk * a.ldv + h * HDIM;
  float mx = -3.0e38f;
  for (int j = 0; j < a.Nk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HDIM; ++d) s += qv[d] * kp[(size_t)j * a.ldk + d];
    mx = fmaxf(mx, s * a.scale);
  }
  float l = 0.f, delta = 0.f;      // delta = sum_j p_j (do . v_j) = do . o
  for (int j = 0; j < a.Nk; ++j) {
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < HDIM; ++d) {
      s += qv[d] * kp[(size_t)j * a.ldk + d];
      dp += dov[d] * vp[(size_t)j * a.ldv + d];
    }
    const float pe = expf(s * a.scale - mx);
    l += pe;
    delta += pe * dp;
  }
  const float inv = 1.0f / l;
  delta *= inv;
  for (int j0 = 0; j0 < a.Nk; j0 += JC) {
#pragma unroll 1
    for (int jj = 0; jj < JC; ++jj) {
      const int j = j0 + jj;
      float p = 0.f, ds = 0.f;
      if (j < a.Nk) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < HDIM; ++d) {
          s += qv[d] * kp[(size_t)j * a.ldk + d];
          dp += dov[d] * vp[(size_t)j * a.ldv + d];
        }
        p = expf(s * a.scale - mx) * inv;
        ds = p * (dp - delta) * a.scale;
#pragma unroll
        for (int d = 0; d < HDIM; ++d) dqv[d] += ds * kp[(size_t)j * a.ldk + d];
      }
      sds[jj][tid] = ds;
      sp[jj][tid] = p;
    }
    __syncthreads();
    const int nj = min(JC, a.Nk - j0);
    for (int o = tid; o < 2 * nj * HDIM; o += QW) {
      const int which = o / (nj * HDIM), r = o % (nj * HDIM), jj = r / HDIM, d = r % HDIM;
      const float* src = which ? sp[jj] : sds[jj];
      const float(*mat)[HDIM + 1] = which ? sdo : sq;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll 4
      for (int q = 0; q < QW; q += 4) {
        t0 += src[q] * mat[q][d];
        t1 += src[q + 1] * mat[q + 1][d];
        t2 += src[q + 2] * mat[q + 2][d];
        t3 += src[q + 3] * mat[q + 3][d];
      }
      float* dst = which ? a.dv + ((size_t)b * a.Nk + j0 + jj) * a.ldv : a.dk + ((size_t)b * a.Nk + j0 + jj) * a.ldk;
      atomicAdd(&dst[h * HDIM + d], (t0 + t1) + (t2 + t3));
    }
    __syncthreads();
  }
  if (live) store_row<HDIM>(a.dq + bq * a.ldq + h * HDIM, dqv, a.vec);
}

template <int HDIM>
__global__ __launch_bounds__(AttnSmallCfg<HDIM>::QW) void attn_bwd_fewqueries_kernel(AttnBwdArgs a) {
  // thread = key of one (group, head); the queries are the short side (their softmax statistics come from la_attn_small_lse)
  constexpr int KW = AttnSmallCfg<HDIM>::QW, TC = AttnSmallCfg<HDIM>::JC;
  __shared__ float sk[KW][HDIM + 1];
  __shared__ float sds[TC][KW];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int kj = blockIdx.y * KW + tid;
  const bool live = kj < a.Nk;
  const float lv = live ? 1.f : 0.f;
  const size_t krow = (size_t)b * a.Nk + min(kj, a.Nk - 1);
  float kv[HDIM], vv[HDIM], dkv[HDIM], dvv[HDIM];
  load_row<HDIM>(a.k + krow * a.ldk + h * HDIM, kv, a.vec);
  load_row<HDIM>(a.v + krow * a.ldv + h * HDIM, vv, a.vec);
#pragma unroll
  for (int d = 0; d < HDIM; ++d) {
    dkv[d] = dvv[d] = 0.f;
    sk[tid][d] = kv[d];
  }
  for (int t0 = 0; t0 < a.Nq; t0 += TC) {
#pragma unroll 1
    for (int tt = 0; tt < TC; ++tt) {
      const int t = t0 + tt;
      float ds = 0.f;
      if (t < a.Nq) {
        const size_t qrow = (size_t)b * a.Nq + t;
        const float* qp = a.q + qrow * a.ldq + h * HDIM;
        const float* dop = a.dout + qrow * a.ldo + h * HDIM;
        const float* op = a.o + qrow * a.ldo + h * HDIM;
        float s = 0.f, dp = 0.f, delta = 0.f;
#pragma unroll
        for (int d = 0; d < HDIM; ++d) {
          s += qp[d] * kv[d];
          dp += dop[d] * vv[d];
          delta += dop[d] * op[d];
        }
        const float p = expf(s * a.scale - a.lse[qrow * a.heads + h]) * lv;
        ds = p * (dp - delta) * a.scale;
#pragma unroll
        for (int d = 0; d < HDIM; ++d) {
          dvv[d] += p * dop[d];
          dkv[d] += ds * qp[d];
        }
      }
      sds[tt][tid] = ds;
    }
    __syncthreads();
    const int nt = min(TC, a.Nq - t0);
    for (int o = tid; o < nt * HDIM; o += KW) {
      const int tt = o / HDIM, d = o % HDIM;
      const float* src = sds[tt];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
      for (int kk = 0; kk < KW; kk += 4) {
        s0 += src[kk] * sk[kk][d];
        s1 += src[kk + 1] * sk[kk + 1][d];
        s2 += src[kk + 2] * sk[kk + 2][d];
        s3 += src[kk + 3] * sk[kk + 3][d];
      }
      atomicAdd(&a.dq[((size_t)b * a.Nq + t0 + tt) * a.ldq + h * HDIM + d], (s0 + s1) + (s2 + s3));
    }
    __syncthreads();
  }
  if (live) {
    store_row<HDIM>(a.dk + krow * a.ldk + h * HDIM, dkv, a.vec);
    store_row<HDIM>(a.dv + krow * a.ldv + h * HDIM, dvv, a.vec);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Adjoint of bilinear resampling (align_corners = False, no antialias; index arithmetic identical to bilinear_kernel of
// post.hip): dx[n, iy, ix] += w * dy[n, oy, ox] over the four taps of every output pixel.  Explicit plane strides so that a
// cropped source region / padded destination frame (Lam.postprocess_masks, lam.py:405-449) needs no copies.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int n, int oh, int ow, long dy_plane, int dy_ld,
                                                           float* __restrict__ dx, int ih, int iw, long dx_plane, int dx_ld) {
  const long total = (long)n * oh * ow;
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const long pl = i / ((long)ow * oh);
    const float g = dy[pl * dy_plane + (long)oy * dy_ld + ox];
    if (g == 0.f) continue;
    float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, ih - 1), x0 = min((int)fx, iw - 1);
    const int y1 = min(y0 + 1, ih - 1), x1 = min(x0 + 1, iw - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    float* base = dx + pl * dx_plane;
    atomicAdd(&base[(long)y0 * dx_ld + x0], g * (1.f - ly) * (1.f - lx));
    atomicAdd(&base[(long)y0 * dx_ld + x1], g * (1.f - ly) * lx);
    atomicAdd(&base[(long)y1 * dx_ld + x0], g * ly * (1.f - lx));
    atomicAdd(&base[(long)y1 * dx_ld + x1], g * ly * lx);
  }
}

// The same adjoint as a GATHER, for reductions (oh <= ih, ow <= iw: the 64 x 64 -> grid resize of a dense mask embedding, 76800 planes
// per cfg3 step): with a source step >= 1 an input row is the upper tap of at most one output row and the lower tap of at most one, so
// every dx entry is a sum of <= 2 x 2 products - no atomics, no zero-filled destination (dx is WRITTEN), one fixed order.  One workgroup
// per plane: the plane of dy and the tap tables of both axes (entries in increasing output index; the two taps of an output pixel that
// fall on the same clamped input index are one entry) sit in LDS.
constexpr int BG_KMAX = 4, BG_MAXDIM = 128, BG_MAXPLANE = 4096;      // 25 KiB of LDS: six workgroups per CU
__device__ __forceinline__ void bg_taps(int i, int isz, int osz, float s, int* cnt, int* idx, float* wt) {
  const int lo = max(0, (int)floorf(((float)i - 0.5f) / s - 0.5f) - 1), hi = min(osz - 1, (int)ceilf(((float)i + 1.5f) / s - 0.5f) + 1);
  int k = 0;
  for (int o = lo; o <= hi; ++o) {
    const float f = fmaxf(((float)o + 0.5f) * s - 0.5f, 0.f);
    const int i0 = min((int)f, isz - 1), i1 = min(i0 + 1, isz - 1);
    const float l = f - (float)i0;
    float w = 0.f;
    if (i0 == i) w += 1.f - l;
    if (i1 == i) w += l;
    if ((i0 == i || i1 == i) && k < BG_KMAX) {
      idx[i * BG_KMAX + k] = o;
      wt[i * BG_KMAX + k] = w;
      ++k;
    }
  }
  cnt[i] = k;
}
__global__ __launch_bounds__(256) void bilinear_bwd_gather_kernel(const float* __restrict__ dy, int oh, int ow, long dy_plane, int dy_ld,
                                                                  float* __restrict__ dx, int ih, int iw, long dx_plane, int dx_ld) {
  __shared__ float g[BG_MAXPLANE];
  __shared__ int ycnt[BG_MAXDIM], xcnt[BG_MAXDIM], yidx[BG_MAXDIM * BG_KMAX], xidx[BG_MAXDIM * BG_KMAX];
  __shared__ float ywt[BG_MAXDIM * BG_KMAX], xwt[BG_MAXDIM * BG_KMAX];
  const long pl = blockIdx.x;
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (int i = threadIdx.x; i < oh * ow; i += 256) g[i] = dy[pl * dy_plane + (long)(i / ow) * dy_ld + (i % ow)];
  for (int i = threadIdx.x; i < ih + iw; i += 256) {
    if (i < ih) bg_taps(i, ih, oh, sy, ycnt, yidx, ywt);
    else bg_taps(i - ih, iw, ow, sx, xcnt, xidx, xwt);
  }
  __syncthreads();
  float* base = dx + pl * dx_plane;
  // (workgroups that walk several planes with the tables built once and the next plane prefetched measured SLOWER - 612 against 442 us on
  // 76800 planes of 64 x 64 <- 30 x 30: fewer workgroups in flight and a barrier per plane cost more than 128 lanes of table building)
  if ((256 % iw) == 0) {                 // a thread keeps its column: the x taps live in registers, rows 256 / iw apart
    const int ix = threadIdx.x % iw, rstep = 256 / iw;
    const int nx = xcnt[ix];
    if (nx <= 2) {
      const int xo0 = nx > 0 ? xidx[ix * BG_KMAX] : 0, xo1 = nx > 1 ? xidx[ix * BG_KMAX + 1] : xo0;
      const float xw0 = nx > 0 ? xwt[ix * BG_KMAX] : 0.f, xw1 = nx > 1 ? xwt[ix * BG_KMAX + 1] : 0.f;
      for (int iy = threadIdx.x / iw; iy < ih; iy += rstep) {
        float acc = 0.f;
        for (int ky = 0; ky < ycnt[iy]; ++ky) {
          const float* row = g + yidx[iy * BG_KMAX + ky] * ow;
          float r = xw0 * row[xo0];
          if (nx > 1) r += xw1 * row[xo1];
          acc += ywt[iy * BG_KMAX + ky] * r;
        }
        base[(long)iy * dx_ld + ix] = acc;
      }
      return;
    }
  }
  for (int i = threadIdx.x; i < ih * iw; i += 256) {
    const int iy = i / iw, ix = i % iw;
    float acc = 0.f;
    for (int ky = 0; ky < ycnt[iy]; ++ky) {
      const float* row = g + yidx[iy * BG_KMAX + ky] * ow;
      float r = 0.f;
      for (int kx = 0; kx < xcnt[ix]; ++kx) r += xwt[ix * BG_KMAX + kx] * row[xidx[ix * BG_KMAX + kx]];
      acc += ywt[iy * BG_KMAX + ky] * r;
    }
    base[(long)iy * dx_ld + ix] = acc;
  }
}

// ... and on NHWC rows (adjoint of la_bilinear_rows): workgroup = one input row (n, iy) of dx [N, ih * iw, C]; its <= 2 output rows and the x
// taps of every column come from the same tables, a thread adds the <= 2 x 2 float4 of dy for four channels of one pixel.
__global__ __launch_bounds__(256) void bilinear_rows_bwd_gather_kernel(const float* __restrict__ dy, int oh, int ow, int C, float* __restrict__ dx,
                                                                       int ih, int iw) {
  __shared__ int ycnt[BG_MAXDIM], xcnt[BG_MAXDIM], yidx[BG_MAXDIM * BG_KMAX], xidx[BG_MAXDIM * BG_KMAX];
  __shared__ float ywt[BG_MAXDIM * BG_KMAX], xwt[BG_MAXDIM * BG_KMAX];
  const long n = blockIdx.x / ih;
  const int iy = blockIdx.x % ih;
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (int i = threadIdx.x; i < iw + 1; i += 256) {
    if (i < iw) bg_taps(i, iw, ow, sx, xcnt, xidx, xwt);
    else bg_taps(iy, ih, oh, sy, ycnt, yidx, ywt);
  }
  __syncthreads();
  const int c4 = C >> 2, ny = ycnt[iy];
  const float4* g = reinterpret_cast<const float4*>(dy) + n * oh * ow * c4;
  float4* out = reinterpret_cast<float4*>(dx) + (n * ih + iy) * iw * c4;
  for (int e = threadIdx.x; e < iw * c4; e += 256) {
    const int ix = e / c4, c = e % c4;
    const int nx = xcnt[ix];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < ny; ++ky) {
      const float4* row = g + (long)yidx[iy * BG_KMAX + ky] * ow * c4 + c;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int kx = 0; kx < nx; ++kx) {
        const float wx = xwt[ix * BG_KMAX + kx];
        const float4 v = row[(long)xidx[ix * BG_KMAX + kx] * c4];
        r.x += wx * v.x; r.y += wx * v.y; r.z += wx * v.z; r.w += wx * v.w;
      }
      const float wy = ywt[iy * BG_KMAX + ky];
      acc.x += wy * r.x; acc.y += wy * r.y; acc.z += wy * r.z; acc.w += wy * r.w;
    }
    out[e] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// seg[b, c, pix] = sum_f protos[b, c, f] feat[b, pix, f]  ->  dfeat[b, pix, f] = sum_c dseg[b, c, pix] protos[b, c, f],
// dprotos[b, c, f] += sum_pix dseg[b, c, pix] feat[b, pix, f]  (wave reduction + one atomic per wave).  C <= 32, CF <= 64.
// ---------------------------------------------------------------------------------------------------------------------------
template <int CF>
__global__ __launch_bounds__(256) void classify_bwd_kernel(const float* __restrict__ dseg, const float* __restrict__ feat,
                                                           const float* __restrict__ protos, int B, int npix, int C, float* __restrict__ dfeat,
                                                           float* __restrict__ dprotos) {
  __shared__ float pr[32 * CF];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < C * CF; i += 256) pr[i] = protos[(size_t)b * C * CF + i];
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < npix;
  const int pc = live ? pix : 0;
  float f[CF], df[CF];
#pragma unroll
  for (int d = 0; d < CF; ++d) {
    f[d] = live ? feat[((size_t)b * npix + pc) * CF + d] : 0.f;
    df[d] = 0.f;
  }
  for (int c = 0; c < C; ++c) {
    const float g = live ? dseg[((size_t)b * C + c) * npix + pc] : 0.f;
#pragma unroll
    for (int d = 0; d < CF; ++d) {
      df[d] += g * pr[c * CF + d];
      const float gp = wave_sum_dpp(g * f[d]);
      if ((threadIdx.x & 63) == 0) atomicAdd(&dprotos[((size_t)b * C + c) * CF + d], gp);
    }
  }
  if (live) {
#pragma unroll
    for (int d = 0; d < CF; ++d) dfeat[((size_t)b * npix + pix) * CF + d] = df[d];
  }
}

__global__ __launch_bounds__(256) void row_broadcast_kernel(const float* __restrict__ src, long groups, int rep, int D, float scale,
                                                            float* __restrict__ out) {
  const long total = groups * rep * (long)(D / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (D / 4));
    const long g = i / ((long)(D / 4) * rep);
    float4 v = reinterpret_cast<const float4*>(src + g * D)[c4];
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

static int grid_for_n(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}


// ---------------------------------------------------------------------------------------------------------------------------
// Image-encoder backward helpers (train_encoder.py).  la_cast: dst = scale * src between fp32 and 16-bit (or fp32 in place: the
// un-scaling of loss-scaled gradients); la_gelu_bwd16: dpre = dh * gelu'(pre) with the pre-activation kept in 16 bit.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n, float scale) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 256 * 4) {
    if (i + 3 < n) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)src[i + k] * scale;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[i + k] = (TD)v[k];
    } else {
      for (long k = i; k < n; ++k) dst[k] = (TD)((float)src[k] * scale);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd16_kernel(const T* __restrict__ pre, const float* __restrict__ dh, float* __restrict__ d32,
                                                         T* __restrict__ d16, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = dh[i] * gelu_grad((float)pre[i]);
    if (d32) d32[i] = g;
    if (d16) d16[i] = (T)g;
  }
}

// the same, 8 elements per thread (n % 8 == 0, 16-byte aligned arrays): one 16-byte and two 16-byte loads, 16-byte stores
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd16_v8_kernel(const T* __restrict__ pre, const float* __restrict__ dh, float* __restrict__ d32,
                                                            T* __restrict__ d16, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 pv = reinterpret_cast<const uint4*>(pre)[i];
    const float4 a = reinterpret_cast<const float4*>(dh)[2 * i], b = reinterpret_cast<const float4*>(dh)[2 * i + 1];
    const T* pe = reinterpret_cast<const T*>(&pv);
    const float d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = d[k] * gelu_grad((float)pe[k]);
    if (d32) {
      reinterpret_cast<float4*>(d32)[2 * i] = make_float4(g[0], g[1], g[2], g[3]);
      reinterpret_cast<float4*>(d32)[2 * i + 1] = make_float4(g[4], g[5], g[6], g[7]);
    }
    if (d16) {
      uint4 o;
      T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
      for (int k = 0; k < 8; ++k) oe[k] = (T)g[k];
      reinterpret_cast<uint4*>(d16)[i] = o;
    }
  }
}

// post = GELU(pre) on 16-bit rows, 8 elements per thread (the training forward keeps the pre-activation for gelu' and derives the
// activation from it: one fc1 GEMM instead of two)
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd16_kernel(const T* __restrict__ pre, T* __restrict__ post, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(pre)[i];
    const T* e = reinterpret_cast<const T*>(&v);
    uint4 o;
    T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int k = 0; k < 8; ++k) oe[k] = (T)gelu_erf((float)e[k]);
    reinterpret_cast<uint4*>(post)[i] = o;
  }
}

// dst[c][r] = (T)src[r][c] for r < R, 0 for R <= r < Rp: the token-contiguous 16-bit operands of a split-K weight-gradient GEMM
// (la_gemm ksplit: dW[N, K] = dY^T X runs as A = dY^T [N, Rp], W = X^T [K, Rp]).  64 x 64 tiles through LDS.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose16_kernel(const TS* __restrict__ src, int ld, int R, int Cn, TD* __restrict__ dst, int Rp,
                                                          float* __restrict__ colsum, int rtiles) {
  // 64 x 64 tile as 32-bit words of two adjacent COLUMNS: tile[r][c / 2].  Output row c needs (r, c), (r + 1, c), ...: two words
  // of rows r, r + 1 give the pairs of columns c and c + 1 with one v_perm each - half the LDS instructions of a 16-bit tile.
  // A workgroup walks `rtiles` row tiles of its column tile (column sums: one atomic per column and workgroup, not per tile - 732
  // row tiles adding to the same 3072 addresses serialise in L2).
  __shared__ unsigned tile[64][32 + 1];
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  float sa = 0.f, sb = 0.f;
  for (int rt = 0; rt < rtiles; ++rt) {
    const int r0 = (blockIdx.x * rtiles + rt) * 64;
    if (r0 >= Rp) break;
    {
      const int r = tid >> 2, cb = (tid & 3) * 16;
      const int rr = r0 + r;
      TD v[16];
      if (rr < R && c0 + cb + 15 < Cn && (ld % 8) == 0 && sizeof(TS) == 2) {
        const uint4 a = *reinterpret_cast<const uint4*>(src + (size_t)rr * ld + c0 + cb);
        const uint4 b2 = *reinterpret_cast<const uint4*>(src + (size_t)rr * ld + c0 + cb + 8);
        *reinterpret_cast<uint4*>(&v[0]) = a;
        *reinterpret_cast<uint4*>(&v[8]) = b2;
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int cc = c0 + cb + i;
          v[i] = (rr < R && cc < Cn) ? (TD)(float)src[(size_t)rr * ld + cc] : (TD)0.f;
        }
      }
      const unsigned* w = reinterpret_cast<const unsigned*>(v);
      if (rt > 0) __syncthreads();                   // the previous tile has been read out
#pragma unroll
      for (int i = 0; i < 8; ++i) tile[r][cb / 2 + i] = w[i];
    }
    __syncthreads();
    {
      // thread -> column pair cp (0..31) and 8 consecutive row pairs: writes 16 rows x 2 columns
      const int cp = tid >> 3, rb = (tid & 7) * 8;
      unsigned lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned a = tile[rb + 2 * i][cp], b2 = tile[rb + 2 * i + 1][cp];
        lo[i] = __builtin_amdgcn_perm(b2, a, 0x05040100);     // (row, c) | (row + 1, c) << 16
        hi[i] = __builtin_amdgcn_perm(b2, a, 0x07060302);     // (row, c + 1) | (row + 1, c + 1) << 16
      }
      const int c = c0 + 2 * cp;
      if (c < Cn) *reinterpret_cast<uint4*>(dst + (size_t)c * Rp + r0 + rb) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      if (c + 1 < Cn) *reinterpret_cast<uint4*>(dst + (size_t)(c + 1) * Rp + r0 + rb) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      if (colsum != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          sa += (float)from_bits16<TD>((uint16_t)(lo[i] & 0xffffu)) + (float)from_bits16<TD>((uint16_t)(lo[i] >> 16));
          sb += (float)from_bits16<TD>((uint16_t)(hi[i] & 0xffffu)) + (float)from_bits16<TD>((uint16_t)(hi[i] >> 16));
        }
      }
    }
  }
  if (colsum != nullptr) {
    // colsum[c] += the column sums of what this workgroup wrote (the bias gradient of the layer whose output gradient is being
    // transposed for its weight gradient: one pass over dY instead of two): the 8 threads of a column pair folded with DPP
    sa += dpp_mov<0xB1>(sa); sb += dpp_mov<0xB1>(sb);        // lanes ^ 1
    sa += dpp_mov<0x4E>(sa); sb += dpp_mov<0x4E>(sb);        // lanes ^ 2
    sa += dpp_mov<0x141>(sa); sb += dpp_mov<0x141>(sb);      // row_half_mirror: lane i <-> 7 - i of its 8-lane group = the other quad
    const int c = c0 + 2 * (tid >> 3);
    if ((tid & 7) == 0) {
      if (c < Cn) atomicAdd(colsum + c, sa);
      if (c + 1 < Cn) atomicAdd(colsum + c + 1, sb);
    }
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = fmaf(a, x[i], y[i]);
}

}  // namespace la

extern "C" int la_gemm_tn_db(const float* dy, int ldy, const float* x, int ldx, float* dw, int ldw, int M, int N, int K, float* db, void* stream);

extern "C" int la_gemm_tn(const float* dy, int ldy, const float* x, int ldx, float* dw, int ldw, int M, int N, int K, void* stream) {
  return la_gemm_tn_db(dy, ldy, x, ldx, dw, ldw, M, N, K, nullptr, stream);
}

extern "C" int la_gemm_tn_db(const float* dy, int ldy, const float* x, int ldx, float* dw, int ldw, int M, int N, int K, float* db, void* stream) {
  LA_CHECK_ARG(dy && x && dw, "la_gemm_tn: null pointer");
  LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && ldw >= K, "la_gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
  hipStream_t st0 = reinterpret_cast<hipStream_t>(stream);
  if (N <= 8 && K <= 8 && M >= 4096) {               // tiny output, long reduction
    const int blocks = (int)(((long)M + 255) / 256 < 1024 ? ((long)M + 255) / 256 : 1024);
    if (N <= 4 && K <= 4) hipLaunchKernelGGL((la::gemm_tn_tiny_kernel<4, 4>), dim3(blocks), dim3(256), 0, st0, dy, ldy, x, ldx, dw, ldw, (long)M, N, K);
    else hipLaunchKernelGGL((la::gemm_tn_tiny_kernel<8, 8>), dim3(blocks), dim3(256), 0, st0, dy, ldy, x, ldx, dw, ldw, (long)M, N, K);
    LA_CHECK_LAUNCH("la_gemm_tn");
    if (db) return la_colsum_acc(dy, ldy, (long)M, N, db, stream);      // (tiny outputs: the separate column-sum pass)
    return 0;
  }
  const bool aligned = ((N | K | ldy | ldx) & 3) == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  static const char* dmaenv = la_dbg_env("LA_TN_NODMA");   // debugging: the register-fed kernel everywhere
  static const char* smallenv = la_dbg_env("LA_TN_NOSMALL");   // debugging: short reductions on the register-fed kernel
  if (aligned && M >= 256 && M < 4096 && !dmaenv && !smallenv) {
    // a few hundred rows (the token-side layers of the decoder): one 128 x 128 tile per workgroup of the register-fed kernel walks all of
    // them alone (23 us for 300 x 256 x 256 on four workgroups); 32 x 32 tiles with the four waves splitting the rows are 64+ workgroups
    la::launch_gemm_tn_dma<1, 1, 1, 1, 32, 4>(dy, ldy, x, ldx, dw, ldw, M, N, K, db, 256, st0);
    LA_CHECK_LAUNCH("la_gemm_tn");
    return 0;
  }
  if (aligned && M >= 4096 && !dmaenv) {
    // workgroups: one resident set (2 per CU) - every workgroup ends in an atomic per output element, and with the rows prefetched
    // through LDS a long chunk costs nothing (256 / 512 / 1024 / 2048 / 4096 workgroups on 270000 x 128 x 256: 234 / 211 / 207 / 235 /
    // 271 us; the 32 x 32 shape on 1228800 rows: 76 / 81 / 119 / 208 / 377 us)
    static const char* wgdma = la_dbg_env("LA_TN_DMA_WGS");
    const int tw = wgdma ? atoi(wgdma) : 512;
    if (N <= 32 && K <= 32) la::launch_gemm_tn_dma<1, 1, 1, 1, 32, 4>(dy, ldy, x, ldx, dw, ldw, M, N, K, db, wgdma ? tw : 256, st0);
    else if (K <= 32) la::launch_gemm_tn_dma<4, 1, 2, 1, 32, 3>(dy, ldy, x, ldx, dw, ldw, M, N, K, db, tw, st0);
    else if (N <= 32) la::launch_gemm_tn_dma<1, 4, 1, 2, 32, 3>(dy, ldy, x, ldx, dw, ldw, M, N, K, db, tw, st0);
    else la::launch_gemm_tn_dma<2, 2, 2, 2, 16, 4>(dy, ldy, x, ldx, dw, ldw, M, N, K, db, tw, st0);
    LA_CHECK_LAUNCH("la_gemm_tn");
    return 0;
  }
  const int tiles_n = (N + 127) / 128, tiles_k = (K + 127) / 128;
  // M-chunks: every chunk ends in one atomic per output element: ~2048 workgroups (256 / 512 / 1024 / 2048 measured 10.4 / 7.2 / 7.2 /
  // 6.9 ms over the decoder's 85 launches: the row loads are latency-bound), at least 128 rows each (multiple of 16)
  static const char* wgenv = la_dbg_env("LA_TN_WGS");       // debugging: target number of workgroups
  int chunks = (wgenv ? atoi(wgenv) : 2048) / (tiles_n * tiles_k);
  if (chunks < 1) chunks = 1;
  int mchunk = (M + chunks - 1) / chunks;
  if (mchunk < 128) mchunk = 128;
  mchunk = (mchunk + 15) / 16 * 16;
  chunks = (M + mchunk - 1) / mchunk;
  hipLaunchKernelGGL(la::gemm_tn_kernel, dim3(tiles_n * tiles_k, chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, ldy, x, ldx,
                     dw, ldw, M, N, K, mchunk, tiles_k, db);
  LA_CHECK_LAUNCH("la_gemm_tn");
  return 0;
}

extern "C" int la_colsum_acc(const float* dy, int ldy, long M, int N, float* out, void* stream) {
  LA_CHECK_ARG(dy && out && M > 0 && N > 0 && ldy >= N, "la_colsum_acc: bad arguments M=%ld N=%d ldy=%d", M, N, ldy);
  int nw = 1;
  while (nw < N && nw < 256) nw <<= 1;
  const int rp = 256 / nw;
  // ~1024 workgroups, at least 64 rows per row lane
  long rows_per_block = (M + 1023) / 1024;
  if (rows_per_block < 64L * rp) rows_per_block = 64L * rp;
  const long blocks = (M + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(la::colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, ldy, M, N, nw,
                     rows_per_block, out);
  LA_CHECK_LAUNCH("la_colsum_acc");
  return 0;
}

extern "C" int la_layernorm_bwd(const float* x, const float* dy, long rows, int E, const float* gamma, const float* beta, float eps, int gelu,
                                float* dx, float* dgamma, float* dbeta, void* stream) {
  return la_layernorm_bwd_res(x, dy, rows, E, gamma, beta, eps, gelu, nullptr, dx, nullptr, LA_F16, dgamma, dbeta, stream);
}

extern "C" int la_layernorm_bwd_res(const float* x, const float* dy, long rows, int E, const float* gamma, const float* beta, float eps, int gelu,
                                    const float* add, float* dx, void* out16, int dt16, float* dgamma, float* dbeta, void* stream) {
  LA_CHECK_ARG(x && dy && gamma && beta && dx && dgamma && dbeta, "la_layernorm_bwd: null pointer");
  LA_CHECK_ARG(dt16 == LA_F16 || dt16 == LA_BF16, "la_layernorm_bwd: bad 16-bit dtype %d", dt16);
  LA_CHECK_ARG(!(add || out16) || !(E <= 32 && rows >= 65536), "la_layernorm_bwd_res: the narrow (E <= 32) form has no skip / 16-bit output");
  LA_CHECK_ARG(rows > 0 && E > 0 && E <= 2048, "la_layernorm_bwd: E=%d out of range (1..2048)", E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (E <= 32 && rows >= 65536) {                    // 4 / 16 / 32 channels over very many pixels (LayerNorm2d stacks): one thread per row
    long nb = (rows + 255) / 256;
    if (nb > 1024) nb = 1024;
    const dim3 g2((unsigned)nb), b2(256);
    if (E <= 4) hipLaunchKernelGGL(la::layernorm_bwd_narrow_kernel<4>, g2, b2, 0, st, x, dy, rows, E, gamma, beta, eps, gelu, dx, dgamma, dbeta);
    else if (E <= 16) hipLaunchKernelGGL(la::layernorm_bwd_narrow_kernel<16>, g2, b2, 0, st, x, dy, rows, E, gamma, beta, eps, gelu, dx, dgamma, dbeta);
    else hipLaunchKernelGGL(la::layernorm_bwd_narrow_kernel<32>, g2, b2, 0, st, x, dy, rows, E, gamma, beta, eps, gelu, dx, dgamma, dbeta);
    LA_CHECK_LAUNCH("la_layernorm_bwd");
    return 0;
  }
  if ((E % 256) == 0 && E <= 1280 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                                       reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(out16) | reinterpret_cast<uintptr_t>(gamma) |
                                       reinterpret_cast<uintptr_t>(beta)) & 15) == 0) {
    long vb = (rows + 3) / 4;
    const long vcap = 256L * (E <= 256 ? 8 : E <= 512 ? 5 : E <= 1024 ? 3 : 2);      // one resident set of 4-wave workgroups (58 / 96 / 136 / 164 / 198 VGPRs)
    if (vb > vcap) vb = vcap;
    const dim3 vg((unsigned)vb), vblk(256);
#define LA_LNV(V) \
  hipLaunchKernelGGL(la::layernorm_bwd_vec_kernel<V>, vg, vblk, 0, st, x, dy, rows, E, gamma, beta, eps, gelu, dx, dgamma, dbeta, add, out16, dt16)
    switch (E / 256) {
      case 1: LA_LNV(1); break;
      case 2: LA_LNV(2); break;
      case 3: LA_LNV(3); break;
      case 4: LA_LNV(4); break;
      default: LA_LNV(5); break;
    }
#undef LA_LNV
    LA_CHECK_LAUNCH("la_layernorm_bwd");
    return 0;
  }
  long blocks = (rows + 3) / 4;
  // workgroup cap = ONE resident set: a workgroup is one wave per SIMD, and the instance's register count (54 / 86 / 158 / 295 for 4 / 8 /
  // 16 / 32 values per lane) allows 8 / 5 / 3 / 1 of them per CU.  A partial second round of the grid-stride loop costs a quarter of the
  // pass (46852 x 768: 1024 workgroups 217 us, 768 172 us; 270000 x 256: 1024 282 us, 2048 256 us)
  static const char* lnb = la_dbg_env("LA_LNB_BLOCKS");    // debugging: workgroup cap
  const long cap = lnb ? atol(lnb) : 256L * (E <= 256 ? 8 : E <= 512 ? 5 : E <= 1024 ? 3 : 1);
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned)blocks), block(256);
#define LA_LNB(V) \
  hipLaunchKernelGGL(la::layernorm_bwd_kernel<V>, grid, block, 0, st, x, dy, rows, E, gamma, beta, eps, gelu, dx, dgamma, dbeta, add, out16, dt16)
  if (E <= 64) LA_LNB(1);
  else if (E <= 128) LA_LNB(2);
  else if (E <= 256) LA_LNB(4);
  else if (E <= 512) LA_LNB(8);
  else if (E <= 1024) LA_LNB(16);
  else LA_LNB(32);
#undef LA_LNB
  LA_CHECK_LAUNCH("la_layernorm_bwd");
  return 0;
}

extern "C" int la_act_fwd(const float* x, float* y, long n, int kind, void* stream) {
  LA_CHECK_ARG(x && y && n > 0 && (kind == LA_ACT_GELU || kind == LA_ACT_RELU), "la_act_fwd: bad arguments");
  hipLaunchKernelGGL(la::act_fwd_kernel, dim3(la::grid_for_n(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n, kind);
  LA_CHECK_LAUNCH("la_act_fwd");
  return 0;
}

extern "C" int la_act_bwd(const float* x, const float* dy, float* dx, long n, int kind, void* stream) {
  LA_CHECK_ARG(x && dy && dx && n > 0 && (kind == LA_ACT_GELU || kind == LA_ACT_RELU), "la_act_bwd: bad arguments");
  hipLaunchKernelGGL(la::act_bwd_kernel, dim3(la::grid_for_n(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, dy, dx, n, kind);
  LA_CHECK_LAUNCH("la_act_bwd");
  return 0;
}

extern "C" int la_attn_small_lse(const float* q, int ldq, const float* k, int ldk, int B, int Nq, int Nk, int heads, int hd, float* lse,
                                 void* stream) {
  LA_CHECK_ARG(q && k && lse && B > 0 && Nq > 0 && Nk > 0 && heads > 0, "la_attn_small_lse: bad arguments");
  la::AttnBwdArgs a{};
  a.q = q; a.k = k; a.ldq = ldq; a.ldk = ldk; a.B = B; a.Nq = Nq; a.Nk = Nk; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)hd);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(B * Nq * heads), block(256);
  switch (hd) {
    case 4: hipLaunchKernelGGL(la::attn_lse_kernel<4>, grid, block, 0, st, a, lse); break;
    case 8: hipLaunchKernelGGL(la::attn_lse_kernel<8>, grid, block, 0, st, a, lse); break;
    case 16: hipLaunchKernelGGL(la::attn_lse_kernel<16>, grid, block, 0, st, a, lse); break;
    case 32: hipLaunchKernelGGL(la::attn_lse_kernel<32>, grid, block, 0, st, a, lse); break;
    case 64: hipLaunchKernelGGL(la::attn_lse_kernel<64>, grid, block, 0, st, a, lse); break;
    default: LA_CHECK_ARG(false, "la_attn_small_lse: unsupported head dim %d (4, 8, 16, 32, 64)", hd);
  }
  LA_CHECK_LAUNCH("la_attn_small_lse");
  return 0;
}

extern "C" int la_attn_small_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* dout,
                                 int ldo, const float* lse, int B, int Nq, int Nk, int heads, int hd, float* dq, float* dk, float* dv,
                                 void* stream) {
  LA_CHECK_ARG(q && k && v && dout && dq && dk && dv, "la_attn_small_bwd: null pointer");
  LA_CHECK_ARG(B > 0 && Nq > 0 && Nk > 0 && heads > 0, "la_attn_small_bwd: bad shape");
  la::AttnBwdArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.lse = lse; a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.Nq = Nq; a.Nk = Nk; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)hd);
  a.vec = ((ldq | ldk | ldv | ldo | hd) & 3) == 0 &&
          ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout) |
            reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 15) == 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool fewkeys = Nk <= 256 && (Nk <= Nq || Nq > 256);
  if (fewkeys) {
    // dq is written, dk / dv are accumulated: the caller zero-fills dk / dv
    const unsigned bh = (unsigned)(B * heads);
#define LA_FK(HD) hipLaunchKernelGGL(la::attn_bwd_fewkeys_kernel<HD>, dim3(bh, (Nq + la::AttnSmallCfg<HD>::QW - 1) / la::AttnSmallCfg<HD>::QW), \
                                     dim3(la::AttnSmallCfg<HD>::QW), 0, st, a)
    switch (hd) {
      case 4: LA_FK(4); break;
      case 8: LA_FK(8); break;
      case 16: LA_FK(16); break;
      case 32: LA_FK(32); break;
      case 64: LA_FK(64); break;
      default: LA_CHECK_ARG(false, "la_attn_small_bwd: unsupported head dim %d (4, 8, 16, 32, 64)", hd);
    }
#undef LA_FK
  } else {
    LA_CHECK_ARG(Nq <= 256, "la_attn_small_bwd: one side of the attention must have <= 256 rows (Nq=%d Nk=%d)", Nq, Nk);
    LA_CHECK_ARG(o && lse, "la_attn_small_bwd: the few-queries form needs the saved output and la_attn_small_lse statistics");
    // dk / dv are written, dq is accumulated: the caller zero-fills dq
    const unsigned bh = (unsigned)(B * heads);
#define LA_FQ(HD) hipLaunchKernelGGL(la::attn_bwd_fewqueries_kernel<HD>, dim3(bh, (Nk + la::AttnSmallCfg<HD>::QW - 1) / la::AttnSmallCfg<HD>::QW), \
                                     dim3(la::AttnSmallCfg<HD>::QW), 0, st, a)
    switch (hd) {
      case 4: LA_FQ(4); break;
      case 8: LA_FQ(8); break;
      case 16: LA_FQ(16); break;
      case 32: LA_FQ(32); break;
      case 64: LA_FQ(64); break;
      default: LA_CHECK_ARG(false, "la_attn_small_bwd: unsupported head dim %d (4, 8, 16, 32, 64)", hd);
    }
#undef LA_FQ
  }
  LA_CHECK_LAUNCH("la_attn_small_bwd");
  return 0;
}

extern "C" int la_bilinear_bwd(const float* dy, int n, int oh, int ow, long dy_plane, int dy_ld, float* dx, int ih, int iw, long dx_plane,
                               int dx_ld, void* stream) {
  LA_CHECK_ARG(dy && dx && n > 0 && oh > 0 && ow > 0 && ih > 0 && iw > 0, "la_bilinear_bwd: bad arguments");
  hipLaunchKernelGGL(la::bilinear_bwd_kernel, dim3(la::grid_for_n((long)n * oh * ow)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, n,
                     oh, ow, dy_plane, dy_ld, dx, ih, iw, dx_plane, dx_ld);
  LA_CHECK_LAUNCH("la_bilinear_bwd");
  return 0;
}

extern "C" int la_bilinear_bwd_set_ok(int oh, int ow, int ih, int iw) {
  return oh > 0 && ow > 0 && oh <= ih && ow <= iw && ih <= la::BG_MAXDIM && iw <= la::BG_MAXDIM && oh * ow <= la::BG_MAXPLANE;
}

extern "C" int la_bilinear_bwd_set(const float* dy, int n, int oh, int ow, long dy_plane, int dy_ld, float* dx, int ih, int iw, long dx_plane,
                                   int dx_ld, void* stream) {
  LA_CHECK_ARG(dy && dx && n > 0, "la_bilinear_bwd_set: bad arguments");
  LA_CHECK_ARG(la_bilinear_bwd_set_ok(oh, ow, ih, iw), "la_bilinear_bwd_set: a reduction with ih, iw <= %d and oh * ow <= %d (got %d x %d -> %d x %d)",
               la::BG_MAXDIM, la::BG_MAXPLANE, ih, iw, oh, ow);
  hipLaunchKernelGGL(la::bilinear_bwd_gather_kernel, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, oh, ow, dy_plane, dy_ld, dx,
                     ih, iw, dx_plane, dx_ld);
  LA_CHECK_LAUNCH("la_bilinear_bwd_set");
  return 0;
}

extern "C" int la_bilinear_rows_bwd_set(const float* dy, int n, int oh, int ow, int C, float* dx, int ih, int iw, void* stream) {
  LA_CHECK_ARG(dy && dx && n > 0 && C > 0 && (C % 4) == 0, "la_bilinear_rows_bwd_set: bad arguments (C %% 4 == 0)");
  LA_CHECK_ARG(la_bilinear_bwd_set_ok(oh, ow, ih, iw), "la_bilinear_rows_bwd_set: a reduction with ih, iw <= %d (got %d x %d -> %d x %d)", la::BG_MAXDIM,
               ih, iw, oh, ow);
  hipLaunchKernelGGL(la::bilinear_rows_bwd_gather_kernel, dim3((unsigned)((long)n * ih)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, oh,
                     ow, C, dx, ih, iw);
  LA_CHECK_LAUNCH("la_bilinear_rows_bwd_set");
  return 0;
}

extern "C" int la_classify_bwd(const float* dseg, const float* feat, const float* protos, int B, int npix, int C, int cf, float* dfeat,
                               float* dprotos, void* stream) {
  LA_CHECK_ARG(dseg && feat && protos && dfeat && dprotos, "la_classify_bwd: null pointer");
  LA_CHECK_ARG(B > 0 && npix > 0 && C > 0 && C <= 32, "la_classify_bwd: C=%d out of range (1..32)", C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((npix + 255) / 256, B), block(256);
  switch (cf) {
    case 8: hipLaunchKernelGGL(la::classify_bwd_kernel<8>, grid, block, 0, st, dseg, feat, protos, B, npix, C, dfeat, dprotos); break;
    case 16: hipLaunchKernelGGL(la::classify_bwd_kernel<16>, grid, block, 0, st, dseg, feat, protos, B, npix, C, dfeat, dprotos); break;
    case 32: hipLaunchKernelGGL(la::classify_bwd_kernel<32>, grid, block, 0, st, dseg, feat, protos, B, npix, C, dfeat, dprotos); break;
    case 64: hipLaunchKernelGGL(la::classify_bwd_kernel<64>, grid, block, 0, st, dseg, feat, protos, B, npix, C, dfeat, dprotos); break;
    default: LA_CHECK_ARG(false, "la_classify_bwd: unsupported feature width %d (8, 16, 32, 64)", cf);
  }
  LA_CHECK_LAUNCH("la_classify_bwd");
  return 0;
}

extern "C" int la_row_broadcast(const float* src, long groups, int rep, int D, float scale, float* out, void* stream) {
  LA_CHECK_ARG(src && out && groups > 0 && rep > 0 && D > 0 && (D % 4) == 0, "la_row_broadcast: bad arguments (D %% 4 == 0)");
  hipLaunchKernelGGL(la::row_broadcast_kernel, dim3(la::grid_for_n(groups * rep * (D / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, groups, rep, D, scale, out);
  LA_CHECK_LAUNCH("la_row_broadcast");
  return 0;
}

extern "C" int la_cast(const void* src, int src_dt, void* dst, int dst_dt, long n, float scale, void* stream) {
  LA_CHECK_ARG(src && dst && n > 0, "la_cast: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(la::grid_for_n((n + 3) / 4)), blk(256);
#define LA_CAST(TS, TD) hipLaunchKernelGGL((la::cast_kernel<TS, TD>), grid, blk, 0, st, (const TS*)src, (TD*)dst, n, scale)
  if (src_dt == LA_F32 && dst_dt == LA_F16) LA_CAST(float, la::f16_t);
  else if (src_dt == LA_F32 && dst_dt == LA_BF16) LA_CAST(float, la::bf16_t);
  else if (src_dt == LA_F16 && dst_dt == LA_F32) LA_CAST(la::f16_t, float);
  else if (src_dt == LA_BF16 && dst_dt == LA_F32) LA_CAST(la::bf16_t, float);
  else if (src_dt == LA_F32 && dst_dt == LA_F32) LA_CAST(float, float);
  else LA_CHECK_ARG(false, "la_cast: unsupported conversion %d -> %d", src_dt, dst_dt);
#undef LA_CAST
  LA_CHECK_LAUNCH("la_cast");
  return 0;
}

extern "C" int la_gelu_bwd16(const void* pre16, const float* dh, float* d32, void* d16, long n, int dt, void* stream) {
  LA_CHECK_ARG(pre16 && dh && (d32 || d16) && n > 0 && (dt == LA_F16 || dt == LA_BF16), "la_gelu_bwd16: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool v8 = (n % 8) == 0 && ((reinterpret_cast<uintptr_t>(pre16) | reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(d32) |
                                    reinterpret_cast<uintptr_t>(d16)) & 15) == 0;
  const dim3 grid(la::grid_for_n(v8 ? n / 8 : n)), blk(256);
  if (v8) {
    if (dt == LA_F16) hipLaunchKernelGGL(la::gelu_bwd16_v8_kernel<la::f16_t>, grid, blk, 0, st, (const la::f16_t*)pre16, dh, d32, (la::f16_t*)d16, n / 8);
    else hipLaunchKernelGGL(la::gelu_bwd16_v8_kernel<la::bf16_t>, grid, blk, 0, st, (const la::bf16_t*)pre16, dh, d32, (la::bf16_t*)d16, n / 8);
  } else if (dt == LA_F16) hipLaunchKernelGGL(la::gelu_bwd16_kernel<la::f16_t>, grid, blk, 0, st, (const la::f16_t*)pre16, dh, d32, (la::f16_t*)d16, n);
  else hipLaunchKernelGGL(la::gelu_bwd16_kernel<la::bf16_t>, grid, blk, 0, st, (const la::bf16_t*)pre16, dh, d32, (la::bf16_t*)d16, n);
  LA_CHECK_LAUNCH("la_gelu_bwd16");
  return 0;
}

extern "C" int la_gelu_fwd16(const void* pre16, void* post16, long n, int dt, void* stream) {
  LA_CHECK_ARG(pre16 && post16 && n > 0 && (n % 8) == 0 && (dt == LA_F16 || dt == LA_BF16), "la_gelu_fwd16: bad arguments (n %% 8 == 0)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(la::grid_for_n(n / 8)), blk(256);
  if (dt == LA_F16) hipLaunchKernelGGL(la::gelu_fwd16_kernel<la::f16_t>, grid, blk, 0, st, (const la::f16_t*)pre16, (la::f16_t*)post16, n / 8);
  else hipLaunchKernelGGL(la::gelu_fwd16_kernel<la::bf16_t>, grid, blk, 0, st, (const la::bf16_t*)pre16, (la::bf16_t*)post16, n / 8);
  LA_CHECK_LAUNCH("la_gelu_fwd16");
  return 0;
}

extern "C" int la_axpy(const float* x, float* y, long n, float a, void* stream) {
  LA_CHECK_ARG(x && y && n > 0, "la_axpy: bad arguments");
  hipLaunchKernelGGL(la::axpy_kernel, dim3(la::grid_for_n(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n, a);
  LA_CHECK_LAUNCH("la_axpy");
  return 0;
}

extern "C" int la_transpose16(const void* src, int src_dt, int ld, int R, int Cn, void* dst, int dst_dt, int Rp, float* colsum, void* stream) {
  LA_CHECK_ARG(src && dst && R > 0 && Cn > 0 && ld >= Cn && Rp >= R && (Rp % 64) == 0, "la_transpose16: bad arguments (R=%d C=%d ld=%d Rp=%d)", R, Cn,
               ld, Rp);
  LA_CHECK_ARG(dst_dt == LA_F16 || dst_dt == LA_BF16, "la_transpose16: 16-bit destination expected");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rtiles = 8;                             // (also without column sums: 8 tiles per workgroup measured faster than one)
  const dim3 grid((Rp / 64 + rtiles - 1) / rtiles, (Cn + 63) / 64), blk(256);
#define LA_TR(TS, TD) hipLaunchKernelGGL((la::transpose16_kernel<TS, TD>), grid, blk, 0, st, (const TS*)src, ld, R, Cn, (TD*)dst, Rp, colsum, rtiles)
  if (src_dt == LA_F32 && dst_dt == LA_F16) LA_TR(float, la::f16_t);
  else if (src_dt == LA_F32 && dst_dt == LA_BF16) LA_TR(float, la::bf16_t);
  else if (src_dt == LA_F16 && dst_dt == LA_F16) LA_TR(la::f16_t, la::f16_t);
  else if (src_dt == LA_BF16 && dst_dt == LA_BF16) LA_TR(la::bf16_t, la::bf16_t);
  else LA_CHECK_ARG(false, "la_transpose16: unsupported conversion %d -> %d", src_dt, dst_dt);
#undef LA_TR
  LA_CHECK_LAUNCH("la_transpose16");
  return 0;
}
