// Decoder-side kernels of the LabelAnything hot path (prompt encoder + mask decoder glue).
// These stages are HBM / latency bound: fp32 VALU arithmetic, coalesced streaming of the
// (P, hw, D) support stream, one pass per kernel.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

// ---------------------------------------------------------------------------------------------------------
// positional encodings (prompt_encoder.py:187-233)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pe_value(const float* __restrict__ gm, int D, int d, float x01, float y01) {
  const int half = D >> 1;
  const int j = d < half ? d : d - half;
  float c = (2.0f * x01 - 1.0f) * gm[j] + (2.0f * y01 - 1.0f) * gm[half + j];
  c = 6.283185307179586f * c;
  return d < half ? sinf(c) : cosf(c);
}

__global__ void dense_pe_kernel(const float* __restrict__ gm, int g, int D, float* __restrict__ out) {
  const int total = g * g * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int d = i % D, pix = i / D;
    const float x = ((float)(pix % g) + 0.5f) / (float)g, y = ((float)(pix / g) + 0.5f) / (float)g;
    out[i] = pe_value(gm, D, d, x, y);
  }
}

// sparse prompt tokens: kind 0 NULL -> not_a_point; 1 negative point; 2 positive point; 3/4 box corners;
// 5 "no sparse prompt" token.  shift != 0 adds the half-pixel offset (prompt_encoder.py:90,107).
__global__ void point_embed_kernel(const float* __restrict__ xy, const int* __restrict__ kind, const int* __restrict__ shift, int n,
                                   int D, float inv_size, const float* __restrict__ gm, const float* __restrict__ type_emb /*[4,D]*/,
                                   const float* __restrict__ not_a_point, const float* __restrict__ no_sparse,
                                   float* __restrict__ out32) {
  const int total = n * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int d = i % D, t = i / D;
    const int k = kind[t];
    float v;
    if (k == 0) v = not_a_point[d];
    else if (k == 5) v = no_sparse[d];
    else {
      const float s = shift[t] ? 0.5f : 0.0f;
      const float x = (xy[2 * t] + s) * inv_size, y = (xy[2 * t + 1] + s) * inv_size;
      v = pe_value(gm, D, d, x, y) + type_emb[(k - 1) * D + d];
    }
    out32[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// la_mask_embed: mask_downscaling (conv2x2s2 -> LN2d -> GELU -> conv2x2s2 -> LN2d -> GELU -> conv1x1),
// bilinear resample hg -> g (commuted in front of the 1x1 conv: both are linear and the taps sum to 1),
// not_a_mask replacement, + support features + class encoding; emits the fp32 stream and the two 16-bit
// GEMM operands (v = src, k = src + pe).  prompt_encoder.py:516-540,61-69,787-814.
// ---------------------------------------------------------------------------------------------------------
struct MaskEmbedArgs {
  const float* masks;   // [P, Hm, Wm] or null
  const int* flags;     // [P] or null
  int P, C, Hm, g, D;
  const float *w0, *b0, *g1, *be1, *w3, *b3, *g4, *be4, *w6, *b6;
  const float* not_a_mask;   // [D]
  const float* no_mask;      // [D] (used when masks == null)
  const float* support;      // [P / C, hw, D] fp32 or null
  const float* class_enc;    // [C, D] or null
  const float* pe;           // [hw, D]
  float* src32;
  void* src16;
  void* srcpe16;
  int split;            // LA_F16X2: [hi | lo] plane pairs
};

constexpr int ME_PIX = 32;

__device__ void mask_hidden16(const MaskEmbedArgs& a, const float* __restrict__ m, int iy, int ix, float* h /*16*/) {
  // 4x4 mask patch -> 2x2 positions x 4 channels
  float patch[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(m + (size_t)(4 * iy + r) * a.Hm + 4 * ix);
    patch[r][0] = v.x; patch[r][1] = v.y; patch[r][2] = v.z; patch[r][3] = v.w;
  }
  float h0[2][2][4];
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      float t[4];
      float mu = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = a.b0[c];
#pragma unroll
        for (int ky = 0; ky < 2; ++ky)
#pragma unroll
          for (int kx = 0; kx < 2; ++kx) s += a.w0[c * 4 + ky * 2 + kx] * patch[2 * py + ky][2 * px + kx];
        t[c] = s;
        mu += s;
      }
      mu *= 0.25f;
      float var = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) var += (t[c] - mu) * (t[c] - mu);
      const float rstd = 1.0f / sqrtf(var * 0.25f + 1e-6f);
#pragma unroll
      for (int c = 0; c < 4; ++c) h0[py][px][c] = gelu_erf((t[c] - mu) * rstd * a.g1[c] + a.be1[c]);
    }
  float mu = 0.f;
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float s = a.b3[o];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int ky = 0; ky < 2; ++ky)
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) s += a.w3[((o * 4 + c) * 2 + ky) * 2 + kx] * h0[ky][kx][c];
    h[o] = s;
    mu += s;
  }
  mu *= (1.0f / 16.0f);
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < 16; ++o) var += (h[o] - mu) * (h[o] - mu);
  const float rstd = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
#pragma unroll
  for (int o = 0; o < 16; ++o) h[o] = gelu_erf((h[o] - mu) * rstd * a.g4[o] + a.be4[o]);
}

__device__ __forceinline__ void bilinear_tap(int dst, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = (float)in / (float)out;
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void mask_embed_kernel(MaskEmbedArgs a) {
  __shared__ float hb[ME_PIX][4][16];
  const int hw = a.g * a.g;
  const int p = blockIdx.y;
  const int pix0 = blockIdx.x * ME_PIX;
  const int tid = threadIdx.x;
  const bool have_mask = a.masks != nullptr;
  const bool valid = have_mask && (a.flags == nullptr || a.flags[p] != 0);
  const int hg = a.Hm >> 2;
  if (valid && tid < ME_PIX * 4) {
    const int pl = tid >> 2, tap = tid & 3;
    const int pix = pix0 + pl;
    float wgt = 0.f;
    int iy = 0, ix = 0;
    if (pix < hw) {
      const int oy = pix / a.g, ox = pix % a.g;
      if (hg == a.g) {
        wgt = tap == 0 ? 1.f : 0.f;
        iy = oy; ix = ox;
      } else {
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_tap(oy, hg, a.g, y0, y1, ly);
        bilinear_tap(ox, hg, a.g, x0, x1, lx);
        iy = (tap & 2) ? y1 : y0;
        ix = (tap & 1) ? x1 : x0;
        wgt = ((tap & 2) ? ly : 1.f - ly) * ((tap & 1) ? lx : 1.f - lx);
      }
    }
    float h[16];
    if (wgt != 0.f) {
      mask_hidden16(a, a.masks + (size_t)p * a.Hm * a.Hm, iy, ix, h);
#pragma unroll
      for (int o = 0; o < 16; ++o) hb[pl][tap][o] = h[o] * wgt;
    } else {
#pragma unroll
      for (int o = 0; o < 16; ++o) hb[pl][tap][o] = 0.f;
    }
  }
  __syncthreads();
  // fold the four bilinear taps once per (pixel, hidden channel): the 1x1 conv below then reads 16 LDS values per output, not 64
  if (valid)
    for (int i = tid; i < ME_PIX * 16; i += 256) {
      const int pl = i >> 4, o = i & 15;
      hb[pl][0][o] = (hb[pl][0][o] + hb[pl][1][o]) + (hb[pl][2][o] + hb[pl][3][o]);
    }
  __syncthreads();
  const int c = p % a.C;
  const int sup = p / a.C;
  T* s16 = reinterpret_cast<T*>(a.src16);
  T* spe16 = reinterpret_cast<T*>(a.srcpe16);
  // a thread owns output channel d for every pixel of the block: its row of the 1x1 conv weight lives in registers
  for (int d = tid; d < a.D; d += 256) {
    float w6r[16];
    float v0;                                        // value before the per-pixel terms: conv bias, or the "no mask" embedding
    if (valid) {
      v0 = a.b6[d];
#pragma unroll
      for (int o = 0; o < 16; ++o) w6r[o] = a.w6[d * 16 + o];
    } else {
      v0 = have_mask ? a.not_a_mask[d] : a.no_mask[d];
#pragma unroll
      for (int o = 0; o < 16; ++o) w6r[o] = 0.f;
    }
    // the support features of all the block's pixels are requested up front: fetched one per iteration, every pixel waited out a
    // memory latency before its store (731 us for 240 pairs x 4096 x 256, a 1.35 GB pass)
    float sv[ME_PIX];
#pragma unroll
    for (int pl = 0; pl < ME_PIX; ++pl)
      sv[pl] = (a.support && pix0 + pl < hw) ? a.support[((size_t)sup * hw + pix0 + pl) * a.D + d] : 0.f;
    if (a.class_enc) v0 += a.class_enc[c * a.D + d];
#pragma unroll
    for (int pl = 0; pl < ME_PIX; ++pl) {
      const int pix = pix0 + pl;
      if (pix >= hw) continue;
      float v = v0;
      if (valid) {
#pragma unroll
        for (int o = 0; o < 16; ++o) v += w6r[o] * hb[pl][0][o];
      }
      v += sv[pl];
      const size_t o = ((size_t)p * hw + pix) * a.D + d;
      a.src32[o] = v;
      if (a.split) {
        const size_t r2 = ((size_t)p * hw + pix) * 2 * a.D;
        if (s16) store_split<T>(s16 + r2, a.D, d, v);
        if (spe16) store_split<T>(spe16 + r2, a.D, d, v + a.pe[(size_t)pix * a.D + d]);
      } else {
        if (s16) s16[o] = (T)v;
        if (spe16) spe16[o] = (T)(v + a.pe[(size_t)pix * a.D + d]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// small attentions of the decoder (common.py:57-148): fp32, head dims 4..64, one side tiny.
//   q [B, Nq, ldq] (head h at column h*HD), k/v [B, Nk, ld], out16/out32 [B, Nq, ldo].
// ---------------------------------------------------------------------------------------------------------
struct SmallAttnArgs {
  const float *q, *k, *v;
  int ldq, ldk, ldv, ldo;
  int B, Nq, Nk, heads;
  float scale;
  void* out16;
  float* out32;
  int split;            // LA_F16X2: out16 rows are [hi | lo] plane pairs of ldo halves each (row stride 2 ldo)
};

// few keys, many queries: one thread per (b, q, head), two passes over the keys (max, then exp-sum).
template <typename T, int HDIM>
__global__ __launch_bounds__(256) void attn_fewkeys_kernel(SmallAttnArgs a) {
  const long total = (long)a.B * a.Nq * a.heads;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)(i % a.heads);
    const long bq = i / a.heads;
    const int b = (int)(bq / a.Nq);
    float qv[HDIM];
    const float* qp = a.q + bq * a.ldq + h * HDIM;
#pragma unroll
    for (int d = 0; d < HDIM; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qp + d);
      qv[d] = t.x; qv[d + 1] = t.y; qv[d + 2] = t.z; qv[d + 3] = t.w;
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
    float acc[HDIM];
#pragma unroll
    for (int d = 0; d < HDIM; ++d) acc[d] = 0.f;
    float l = 0.f;
    for (int j = 0; j < a.Nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HDIM; ++d) s += qv[d] * kp[(size_t)j * a.ldk + d];
      const float pexp = expf(s * a.scale - mx);
      l += pexp;
#pragma unroll
      for (int d = 0; d < HDIM; ++d) acc[d] += pexp * vp[(size_t)j * a.ldv + d];
    }
    const float inv = 1.0f / l;
    if (a.out32) {
#pragma unroll
      for (int d = 0; d < HDIM; ++d) a.out32[bq * a.ldo + h * HDIM + d] = acc[d] * inv;
    }
    if (a.out16 && a.split) {
      T* op = reinterpret_cast<T*>(a.out16) + bq * 2 * a.ldo;
#pragma unroll
      for (int d = 0; d < HDIM; ++d) store_split<T>(op, a.ldo, h * HDIM + d, acc[d] * inv);
    } else if (a.out16) {
      T* op = reinterpret_cast<T*>(a.out16) + bq * a.ldo + h * HDIM;
#pragma unroll
      for (int d = 0; d < HDIM; d += 2) store2<T>(op + d, acc[d] * inv, acc[d + 1] * inv);
    }
  }
}

// many keys, few queries: one workgroup per (b, q, head); threads stride over keys with an online softmax,
// then the 256 partial (m, l, acc) states are merged through LDS.
template <typename T, int HDIM>
__global__ __launch_bounds__(256) void attn_manykeys_kernel(SmallAttnArgs a) {
  __shared__ float red[4][HDIM + 2];
  const int i = blockIdx.x;
  const int h = i % a.heads;
  const int bq = i / a.heads;
  const int b = bq / a.Nq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float qv[HDIM];
  const float* qp = a.q + (size_t)bq * a.ldq + h * HDIM;
#pragma unroll
  for (int d = 0; d < HDIM; ++d) qv[d] = qp[d] * a.scale;
  const float* kp = a.k + (size_t)b * a.Nk * a.ldk + h * HDIM;
  const float* vp = a.v + (size_t)b * a.Nk * a.ldv + h * HDIM;
  float m = -3.0e38f, l = 0.f;
  float acc[HDIM];
#pragma unroll
  for (int d = 0; d < HDIM; ++d) acc[d] = 0.f;
  for (int j = tid; j < a.Nk; j += 256) {
    const float* kr = kp + (size_t)j * a.ldk;
    const float* vr = vp + (size_t)j * a.ldv;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HDIM; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(kr + d);
      s += qv[d] * t.x + qv[d + 1] * t.y + qv[d + 2] * t.z + qv[d + 3] * t.w;
    }
    const float mn = fmaxf(m, s);
    const float al = expf(m - mn), pe = expf(s - mn);
    l = l * al + pe;
#pragma unroll
    for (int d = 0; d < HDIM; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(vr + d);
      acc[d] = acc[d] * al + pe * t.x;
      acc[d + 1] = acc[d + 1] * al + pe * t.y;
      acc[d + 2] = acc[d + 2] * al + pe * t.z;
      acc[d + 3] = acc[d + 3] * al + pe * t.w;
    }
    m = mn;
  }
  // wave merge
  const float mw = wave_max(m);
  const float f = expf(m - mw);  // threads without keys: m = -3e38 -> f = 0
  l = wave_sum(l * f);
#pragma unroll
  for (int d = 0; d < HDIM; ++d) acc[d] = wave_sum(acc[d] * f);
  if (lane == 0) {
    red[wave][0] = mw;
    red[wave][1] = l;
#pragma unroll
    for (int d = 0; d < HDIM; ++d) red[wave][2 + d] = acc[d];
  }
  __syncthreads();
  if (tid < HDIM) {
    const float mg = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    float lt = 0.f, at = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float fw = expf(red[w][0] - mg);
      lt += red[w][1] * fw;
      at += red[w][2 + tid] * fw;
    }
    const float o = at / lt;
    if (a.out32) a.out32[(size_t)bq * a.ldo + h * HDIM + tid] = o;
    if (a.out16 && a.split) store_split<T>(reinterpret_cast<T*>(a.out16) + (size_t)bq * 2 * a.ldo, a.ldo, h * HDIM + tid, o);
    else if (a.out16) reinterpret_cast<T*>(a.out16)[(size_t)bq * a.ldo + h * HDIM + tid] = (T)o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// pooling / prototypes / classification
// ---------------------------------------------------------------------------------------------------------
// mean over the hw rows of each [hw, D] slab (adaptive_avg_pool1d, prompt_encoder.py:735-736).  Few slabs (P = pairs), long
// columns: the rows of a slab are split over COLMEAN_SPLIT workgroups whose partial sums go to a scratch row each and are
// folded in a fixed order by the last pass (deterministic - no float atomics), so the whole GPU streams the slab.
constexpr int COLMEAN_SPLIT = 16;
__global__ __launch_bounds__(256) void colmean_partial_kernel(const float* __restrict__ x, int hw, int D, float* __restrict__ part) {
  __shared__ float4 red[256];
  const int p = blockIdx.y, chunk = blockIdx.x;
  const int nv = D >> 2;                                   // float4 columns
  const int lanes = min(nv, 256), groups = 256 / lanes;    // threads per row, rows in flight (D % 4 == 0, D <= 1024)
  const int c = threadIdx.x % lanes, g = threadIdx.x / lanes;
  const int r0 = (int)((long)hw * chunk / COLMEAN_SPLIT), r1 = (int)((long)hw * (chunk + 1) / COLMEAN_SPLIT);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < groups)
    for (int r = r0 + g; r < r1; r += groups) {
      const float4 v = reinterpret_cast<const float4*>(x + ((size_t)p * hw + r) * D)[c];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0) {
    for (int i = 1; i < groups; ++i) {
      const float4 v = red[i * lanes + c];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(part + ((size_t)p * COLMEAN_SPLIT + chunk) * D)[c] = s;
  }
}
__global__ __launch_bounds__(256) void colmean_final_kernel(const float* __restrict__ part, int P, int hw, int D, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P * D) return;
  const int p = i / D, d = i % D;
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < COLMEAN_SPLIT; ++k) t += part[((size_t)p * COLMEAN_SPLIT + k) * D + d];
  out[i] = t / (float)hw;
}

// class prototypes = masked mean over the M supports, divisor clamped to >= 1 (prompt_encoder.py:738-745)
__global__ void class_mean_kernel(const float* __restrict__ emb /*[B,M,C,D]*/, const uint8_t* __restrict__ flags /*[B,M,C]*/, int B, int M,
                                  int C, int D, float* __restrict__ out /*[B,C,D]*/) {
  const int total = B * C * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int d = i % D, c = (i / D) % C, b = i / (D * C);
    float s = 0.f, n = 0.f;
    for (int m = 0; m < M; ++m) {
      const float f = (float)flags[(b * M + m) * C + c];
      s += emb[(((size_t)b * M + m) * C + c) * D + d] * f;
      n += f;
    }
    out[i] = s / (n == 0.f ? 1.f : n);
  }
}

// seg[b][c][pix] = protos[b][c] . feat[b][pix]   (mask_decoder.py:299-314), feat NHWC fp32 [B, Npix, Cf]
template <int CF>
__global__ __launch_bounds__(256) void classify_kernel(const float* __restrict__ feat, const float* __restrict__ protos, int B, int Npix,
                                                       int C, float* __restrict__ seg) {
  const long total = (long)B * Npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / Npix);
    const long pix = i % Npix;
    float f[CF];
    const float4* fp = reinterpret_cast<const float4*>(feat + i * CF);
#pragma unroll
    for (int d = 0; d < CF / 4; ++d) {
      const float4 t = fp[d];
      f[4 * d] = t.x; f[4 * d + 1] = t.y; f[4 * d + 2] = t.z; f[4 * d + 3] = t.w;
    }
    for (int c = 0; c < C; ++c) {
      const float* pp = protos + ((size_t)b * C + c) * CF;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < CF; ++d) s += f[d] * pp[d];
      seg[((size_t)b * C + c) * Npix + pix] = s;
    }
  }
}

// out = x (+ y[row % ymod]) as fp32 and/or 16 bit; contiguous [rows, D]
template <typename T>
__global__ void add_cast_kernel(const float* __restrict__ x, const float* __restrict__ y, int ymod, long rows, int D, float* __restrict__ out32,
                                T* __restrict__ out16, int split) {
  const long total = rows * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (y) {
      const long r = i / D;
      v += y[(ymod ? r % ymod : r) * D + (i % D)];
    }
    if (out32) out32[i] = v;
    if (out16 && split) store_split<T>(out16 + (i / D) * 2 * D, D, (int)(i % D), v);
    else if (out16) out16[i] = (T)v;
  }
}

// NCHW fp32 -> NHWC 16-bit / fp32 (precomputed-embedding inputs), and NHWC -> NCHW fp32 (API outputs)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int N, int Cc, int HW, float* __restrict__ out32, T* __restrict__ out16) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    tile[r][tx] = (c < Cc && p < HW) ? in[((size_t)n * Cc + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < Cc) {
      const float v = tile[tx][r];
      if (out32) out32[((size_t)n * HW + p) * Cc + c] = v;
      if (out16) out16[((size_t)n * HW + p) * Cc + c] = (T)v;
    }
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int N, int Cc, int HW, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (c < Cc && p < HW) ? in[((size_t)n * HW + p) * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (c < Cc && p < HW) out[((size_t)n * Cc + c) * HW + p] = tile[tx][r];
  }
}

// Many small fp32 matrices transposed by ONE launch (the W^T copies that the data-gradient products dX = dY W of every nn.Linear of the
// decoder need: 81 launches of the kernel above per training step, 16 us each).  tab: 4 x int64 per 32 x 32 tile -
// (src, dst, rows << 32 | cols, r0 << 32 | c0); dst[c][r] = src[r][c].
__global__ __launch_bounds__(256) void transpose_many_kernel(const long long* __restrict__ tab) {
  __shared__ float tile[32][33];
  const long long* t = tab + (size_t)blockIdx.x * 4;
  const float* src = reinterpret_cast<const float*>(t[0]);
  float* dst = reinterpret_cast<float*>(t[1]);
  const int rows = (int)(t[2] >> 32), cols = (int)(t[2] & 0xffffffffll), r0 = (int)(t[3] >> 32), c0 = (int)(t[3] & 0xffffffffll);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < rows && cc < cols) ? src[(size_t)rr * cols + cc] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int cc = c0 + r, rr = r0 + tx;
    if (cc < cols && rr < rows) dst[(size_t)cc * rows + rr] = tile[tx][r];
  }
}

static inline int grid_for(long total, int block = 256, int cap = 16384) {
  long b = (total + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace la

using namespace la;

extern "C" int la_transpose_many(const long long* tile_table, int ntiles, void* stream) {
  LA_CHECK_ARG(tile_table && ntiles > 0, "la_transpose_many: bad arguments");
  hipLaunchKernelGGL(transpose_many_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, tile_table);
  LA_CHECK_LAUNCH("la_transpose_many");
  return 0;
}

extern "C" int la_dense_pe(const float* gauss, int g, int D, float* out, void* stream) {
  LA_CHECK_ARG(gauss && out && g > 0 && D > 0 && (D % 2) == 0, "la_dense_pe: bad arguments");
  hipLaunchKernelGGL(dense_pe_kernel, dim3(grid_for((long)g * g * D)), dim3(256), 0, (hipStream_t)stream, gauss, g, D, out);
  LA_CHECK_LAUNCH("la_dense_pe");
  return 0;
}

extern "C" int la_point_embed(const float* xy, const int* kind, const int* shift, int n, int D, int image_size, const float* gauss,
                              const float* type_emb, const float* not_a_point, const float* no_sparse, float* out32, void* stream) {
  LA_CHECK_ARG(xy && kind && shift && gauss && type_emb && not_a_point && no_sparse && out32 && n > 0, "la_point_embed: bad arguments");
  hipLaunchKernelGGL(point_embed_kernel, dim3(grid_for((long)n * D)), dim3(256), 0, (hipStream_t)stream, xy, kind, shift, n, D,
                     1.0f / (float)image_size, gauss, type_emb, not_a_point, no_sparse, out32);
  LA_CHECK_LAUNCH("la_point_embed");
  return 0;
}

extern "C" int la_mask_embed(const float* masks, const int* flags, int P, int C, int Hm, int g, int D, const float* const* w,
                             const float* support, const float* class_enc, const float* pe, float* src32, void* src16, void* srcpe16,
                             int dt, void* stream) {
  LA_CHECK_ARG(w && pe && src32, "la_mask_embed: null pointer");
  LA_CHECK_ARG(P > 0 && C > 0 && g > 0 && D > 0, "la_mask_embed: bad shape");
  LA_CHECK_ARG(!masks || (Hm > 0 && (Hm % 4) == 0), "la_mask_embed: mask side %d must be a multiple of 4", Hm);
  MaskEmbedArgs a;
  a.masks = masks; a.flags = flags; a.P = P; a.C = C; a.Hm = Hm; a.g = g; a.D = D;
  a.w0 = w[0]; a.b0 = w[1]; a.g1 = w[2]; a.be1 = w[3]; a.w3 = w[4]; a.b3 = w[5]; a.g4 = w[6]; a.be4 = w[7]; a.w6 = w[8]; a.b6 = w[9];
  a.not_a_mask = w[10]; a.no_mask = w[11];
  a.support = support; a.class_enc = class_enc; a.pe = pe; a.src32 = src32; a.src16 = src16; a.srcpe16 = srcpe16;
  a.split = dt == LA_F16X2 ? 1 : 0;
  const int hw = g * g;
  dim3 grid((hw + ME_PIX - 1) / ME_PIX, P);
  if (dt == LA_F16 || dt == LA_F16X2) hipLaunchKernelGGL(mask_embed_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (dt == LA_BF16) hipLaunchKernelGGL(mask_embed_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (dt == LA_F32) hipLaunchKernelGGL(mask_embed_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else LA_CHECK_ARG(false, "la_mask_embed: bad dtype %d", dt);
  LA_CHECK_LAUNCH("la_mask_embed");
  return 0;
}

template <typename T, int HDIM>
static void launch_small(const SmallAttnArgs& a, hipStream_t st) {
  if (a.Nk <= 96) {
    const long total = (long)a.B * a.Nq * a.heads;
    hipLaunchKernelGGL((attn_fewkeys_kernel<T, HDIM>), dim3(grid_for(total)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((attn_manykeys_kernel<T, HDIM>), dim3(a.B * a.Nq * a.heads), dim3(256), 0, st, a);
  }
}

template <typename T>
static int dispatch_small(const SmallAttnArgs& a, int hd, hipStream_t st) {
  switch (hd) {
    case 4: launch_small<T, 4>(a, st); break;
    case 8: launch_small<T, 8>(a, st); break;
    case 16: launch_small<T, 16>(a, st); break;
    case 32: launch_small<T, 32>(a, st); break;
    case 64: launch_small<T, 64>(a, st); break;
    default: return -1;
  }
  return 0;
}

extern "C" int la_attn_small(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int B, int Nq, int Nk, int heads,
                             int hd, void* out16, float* out32, int ldo, int dt, void* stream) {
  LA_CHECK_ARG(q && k && v && (out16 || out32), "la_attn_small: null pointer");
  LA_CHECK_ARG(B > 0 && Nq > 0 && Nk > 0 && heads > 0, "la_attn_small: bad shape");
  LA_CHECK_ARG((ldq % 4) == 0 && (ldk % 4) == 0 && (ldv % 4) == 0 && (ldo % 2) == 0, "la_attn_small: leading dims must be multiples of 4");
  SmallAttnArgs a{q, k, v, ldq, ldk, ldv, ldo, B, Nq, Nk, heads, 1.0f / sqrtf((float)hd), out16, out32, dt == LA_F16X2 ? 1 : 0};
  int rc = (dt == LA_BF16) ? dispatch_small<bf16_t>(a, hd, (hipStream_t)stream)
           : (dt == LA_F32) ? dispatch_small<float>(a, hd, (hipStream_t)stream) : dispatch_small<f16_t>(a, hd, (hipStream_t)stream);
  LA_CHECK_ARG(rc == 0, "la_attn_small: unsupported head dim %d (4, 8, 16, 32, 64)", hd);
  LA_CHECK_LAUNCH("la_attn_small");
  return 0;
}

extern "C" int la_colmean(const float* x, int P, int hw, int D, float* out, float* scratch, void* stream) {
  LA_CHECK_ARG(x && out && scratch && P > 0 && hw > 0 && D > 0 && (D % 4) == 0 && D <= 1024, "la_colmean: bad arguments (D=%d must be a multiple of 4, <= 1024)", D);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colmean_partial_kernel, dim3(COLMEAN_SPLIT, P), dim3(256), 0, st, x, hw, D, scratch);
  hipLaunchKernelGGL(colmean_final_kernel, dim3((P * D + 255) / 256), dim3(256), 0, st, scratch, P, hw, D, out);
  LA_CHECK_LAUNCH("la_colmean");
  return 0;
}

extern "C" int la_class_mean(const float* emb, const unsigned char* flags, int B, int M, int C, int D, float* out, void* stream) {
  LA_CHECK_ARG(emb && flags && out && B > 0 && M > 0 && C > 0 && D > 0, "la_class_mean: bad arguments");
  hipLaunchKernelGGL(class_mean_kernel, dim3(grid_for((long)B * C * D)), dim3(256), 0, (hipStream_t)stream, emb, flags, B, M, C, D, out);
  LA_CHECK_LAUNCH("la_class_mean");
  return 0;
}

extern "C" int la_classify(const float* feat, const float* protos, int B, int Npix, int C, int Cf, float* seg, void* stream) {
  LA_CHECK_ARG(feat && protos && seg && B > 0 && Npix > 0 && C > 0, "la_classify: bad arguments");
  const int grid = grid_for((long)B * Npix);
  hipStream_t st = (hipStream_t)stream;
  switch (Cf) {
    case 8: hipLaunchKernelGGL(classify_kernel<8>, dim3(grid), dim3(256), 0, st, feat, protos, B, Npix, C, seg); break;
    case 16: hipLaunchKernelGGL(classify_kernel<16>, dim3(grid), dim3(256), 0, st, feat, protos, B, Npix, C, seg); break;
    case 32: hipLaunchKernelGGL(classify_kernel<32>, dim3(grid), dim3(256), 0, st, feat, protos, B, Npix, C, seg); break;
    case 64: hipLaunchKernelGGL(classify_kernel<64>, dim3(grid), dim3(256), 0, st, feat, protos, B, Npix, C, seg); break;
    default: LA_CHECK_ARG(false, "la_classify: unsupported feature width %d (8, 16, 32, 64)", Cf);
  }
  LA_CHECK_LAUNCH("la_classify");
  return 0;
}

extern "C" int la_add_cast(const float* x, const float* y, int ymod, long rows, int D, float* out32, void* out16, int dt, void* stream) {
  LA_CHECK_ARG(x && (out32 || out16) && rows > 0 && D > 0, "la_add_cast: bad arguments");
  const int grid = grid_for(rows * D);
  if (dt == LA_BF16) hipLaunchKernelGGL(add_cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, ymod, rows, D, out32, (bf16_t*)out16, 0);
  else if (dt == LA_F32) hipLaunchKernelGGL(add_cast_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, ymod, rows, D, out32, (float*)out16, 0);
  else hipLaunchKernelGGL(add_cast_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, ymod, rows, D, out32, (f16_t*)out16, dt == LA_F16X2 ? 1 : 0);
  LA_CHECK_LAUNCH("la_add_cast");
  return 0;
}

extern "C" int la_nchw_to_nhwc(const float* in, int N, int C, int HW, float* out32, void* out16, int dt, void* stream) {
  LA_CHECK_ARG(in && (out32 || out16) && N > 0 && C > 0 && HW > 0, "la_nchw_to_nhwc: bad arguments");
  dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
  if (dt == LA_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, in, N, C, HW, out32, (bf16_t*)out16);
  else if (dt == LA_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, in, N, C, HW, out32, (float*)out16);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, in, N, C, HW, out32, (f16_t*)out16);
  LA_CHECK_LAUNCH("la_nchw_to_nhwc");
  return 0;
}

extern "C" int la_nhwc_to_nchw(const float* in, int N, int C, int HW, float* out, void* stream) {
  LA_CHECK_ARG(in && out && N > 0 && C > 0 && HW > 0, "la_nhwc_to_nchw: bad arguments");
  dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, N, C, HW, out);
  LA_CHECK_LAUNCH("la_nhwc_to_nchw");
  return 0;
}
