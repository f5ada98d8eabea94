// Logit post-processing (lam.py:383-453, 92-93) and the caller's argmax (experiment/run.py:697).
// fp32 throughout; index arithmetic follows F.interpolate(mode="bilinear", align_corners=False) exactly
// (source = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out as float).  Compiled with -ffp-contract=off so the
// blend is evaluated as written: t = w0*a + w1*b per row, out = wy0*t0 + wy1*t1.
#include "la_common.h"
#include "../../include/la_hip.h"

#pragma clang fp contract(off)

namespace la {

__device__ __forceinline__ void tap(int dst, float scale, int in, int& i0, int& i1, float& w0, float& w1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.0f - w1;
}

// planes [N, h, w] -> [N, H, W]
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ in, int N, int h, int w, int H, int W, float* __restrict__ out) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long total = (long)N * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long n = i / ((long)W * H);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    tap(y, sy, h, y0, y1, wy0, wy1);
    tap(x, sx, w, x0, x1, wx0, wx1);
    const float* p = in + n * h * w;
    const float t0 = wx0 * p[y0 * w + x0] + wx1 * p[y0 * w + x1];
    const float t1 = wx0 * p[y1 * w + x0] + wx1 * p[y1 * w + x1];
    out[i] = wy0 * t0 + wy1 * t1;
  }
}

// NHWC rows [N, h * w, C] -> [N, H * W, C]: the same taps and the same blend per channel, four channels per thread - the resize of a dense mask
// embedding in the training graph without the plane transposes around it (prompt_encoder.py:528-540; 2 x 1.3 GB per direction on cfg3)
__global__ __launch_bounds__(256) void bilinear_rows_kernel(const float* __restrict__ in, int N, int h, int w, int C, int H, int W,
                                                            float* __restrict__ out) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int c4 = C >> 2;
  const long total = (long)N * H * W * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4);
    const long pix = i / c4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long n = pix / ((long)W * H);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    tap(y, sy, h, y0, y1, wy0, wy1);
    tap(x, sx, w, x0, x1, wx0, wx1);
    const float4* p = reinterpret_cast<const float4*>(in) + n * h * w * c4 + c;
    const float4 a00 = p[(long)(y0 * w + x0) * c4], a01 = p[(long)(y0 * w + x1) * c4];
    const float4 a10 = p[(long)(y1 * w + x0) * c4], a11 = p[(long)(y1 * w + x1) * c4];
    float4 o;
    o.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
    o.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
    o.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
    o.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// Second half of postprocess_masks for a batch: per item b the S x S logits are cropped to (ph, pw), resampled to
// the original (oh, ow), written into the (Hmax, Wmax) frame padded with -inf (class 0 padded with 0), classes
// without ground truth forced to -inf, and the class argmax is taken (first maximal index, as torch.argmax).
// sizes: int32 [B, 4] = (oh, ow, ph, pw).
__global__ __launch_bounds__(256) void post_final_kernel(const float* __restrict__ big, int B, int C, int S, const int* __restrict__ sizes,
                                                         const uint8_t* __restrict__ flag_gts, int Hmax, int Wmax, float* __restrict__ logits,
                                                         int64_t* __restrict__ argmax) {
  const long total = (long)B * Hmax * Wmax;
  const float ninf = -__builtin_inff();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wmax), y = (int)((i / Wmax) % Hmax), b = (int)(i / ((long)Wmax * Hmax));
    const int oh = sizes[4 * b], ow = sizes[4 * b + 1], ph = sizes[4 * b + 2], pw = sizes[4 * b + 3];
    const bool inside = (y < oh) && (x < ow);
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    float wy0 = 0.f, wy1 = 0.f, wx0 = 0.f, wx1 = 0.f;
    if (inside) {
      tap(y, (float)ph / (float)oh, ph, y0, y1, wy0, wy1);
      tap(x, (float)pw / (float)ow, pw, x0, x1, wx0, wx1);
    }
    float best = 0.f;
    int besti = 0;
    for (int c = 0; c < C; ++c) {
      float v;
      if (inside) {
        const float* p = big + ((size_t)b * C + c) * S * S;
        const float t0 = wx0 * p[y0 * S + x0] + wx1 * p[y0 * S + x1];
        const float t1 = wx0 * p[y1 * S + x0] + wx1 * p[y1 * S + x1];
        v = wy0 * t0 + wy1 * t1;
      } else {
        v = (c == 0) ? 0.f : ninf;
      }
      if (flag_gts && !flag_gts[b * C + c]) v = ninf;
      if (logits) logits[(((size_t)b * C + c) * Hmax + y) * Wmax + x] = v;
      if (c == 0 || v > best) {
        best = v;
        besti = c;
      }
    }
    if (argmax) argmax[i] = besti;
  }
}

}  // namespace la

extern "C" int la_bilinear(const float* in, int N, int h, int w, int H, int W, float* out, void* stream) {
  LA_CHECK_ARG(in && out && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "la_bilinear: bad arguments");
  const long total = (long)N * H * W;
  const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(la::bilinear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, N, h, w, H, W, out);
  LA_CHECK_LAUNCH("la_bilinear");
  return 0;
}

extern "C" int la_bilinear_rows(const float* in, int N, int h, int w, int C, float* out, int H, int W, void* stream) {
  LA_CHECK_ARG(in && out && N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "la_bilinear_rows: bad arguments (C %% 4 == 0)");
  const long total = (long)N * H * W * (C / 4);
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(la::bilinear_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, N, h, w, C, H, W, out);
  LA_CHECK_LAUNCH("la_bilinear_rows");
  return 0;
}

extern "C" int la_post_final(const float* big, int B, int C, int S, const int* sizes, const unsigned char* flag_gts, int Hmax, int Wmax,
                             float* logits, long long* argmax, void* stream) {
  LA_CHECK_ARG(big && sizes && (logits || argmax) && B > 0 && C > 0 && S > 0 && Hmax > 0 && Wmax > 0, "la_post_final: bad arguments");
  const long total = (long)B * Hmax * Wmax;
  const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(la::post_final_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, big, B, C, S, sizes, flag_gts, Hmax, Wmax, logits,
                     reinterpret_cast<int64_t*>(argmax));
  LA_CHECK_LAUNCH("la_post_final");
  return 0;
}
