// Gather kernels that turn the path's convolutions into la_gemm operands (16-bit row-major A matrices).
// Both are pure streaming: every thread moves one 16-byte output chunk.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

// image fp32 NCHW [Bn,3,S,S] -> [Bn*g*g, 3*p*p]; column k = c*p*p + ky*p + kx (Conv2d weight flattening order).
// SPLIT: rows are [hi (K) | lo (K)] fp16 plane pairs (LA_F16X2: the patch-embed GEMM then runs on three fp16 products).
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, int Bn, int S, int p, T* __restrict__ out) {
  const int g = S / p;
  const int K = 3 * p * p;
  const int cpr = K >> 3;  // 16-B chunks per output row
  const long total = (long)Bn * g * g * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const long row = i / cpr;
    const int k = ch << 3;
    const int c = k / (p * p), ky = (k / p) % p, kx = k % p;  // p % 8 == 0 -> the 8 elements share (c, ky)
    const int px = (int)(row % g), py = (int)((row / g) % g), b = (int)(row / ((long)g * g));
    const float* src = img + (((long)b * 3 + c) * S + (py * p + ky)) * S + px * p + kx;
    const float4 v0 = *reinterpret_cast<const float4*>(src);
    const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
    if constexpr (sizeof(T) == 4) {        // fp32 patches for the exact-fp32 patch-embed GEMM (split-precision group "patch")
      reinterpret_cast<float4*>(out + row * K + k)[0] = v0;
      reinterpret_cast<float4*>(out + row * K + k)[1] = v1;
    } else if constexpr (SPLIT) {
      store4_split<T>(out + row * 2 * K, K, k, v0.x, v0.y, v0.z, v0.w);
      store4_split<T>(out + row * 2 * K, K, k + 4, v1.x, v1.y, v1.z, v1.w);
    } else {
      uint4 o;
      o.x = pack2<T>(v0.x, v0.y);
      o.y = pack2<T>(v0.z, v0.w);
      o.z = pack2<T>(v1.x, v1.y);
      o.w = pack2<T>(v1.z, v1.w);
      *reinterpret_cast<uint4*>(out + row * K + k) = o;
    }
  }
}

// NHWC 16-bit [B,H,W,C] -> [B*H*W, 9*C]; column k = (ky*3+kx)*C + c, zero padding 1.  planes = 2 (LA_F16X2): the pixel rows are
// [hi (C) | lo (C)] plane pairs and the output rows [9 taps of hi | 9 taps of lo], i.e. again a plane pair of the whole patch.
__global__ __launch_bounds__(256) void im2col_3x3_kernel(const uint4* __restrict__ in, int B, int H, int W, int C8, int planes,
                                                         uint4* __restrict__ out) {
  const long total = (long)B * H * W * planes * 9 * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long r = i / C8;
    const int tap = (int)(r % 9);
    r /= 9;
    const int pl = (int)(r % planes);
    r /= planes;
    const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((long)W * H));
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[((((long)b * H + yy) * W + xx) * planes + pl) * C8 + c];
    out[i] = v;
  }
}

}  // namespace la

extern "C" int la_im2col_patch(const float* img, int Bn, int S, int patch, void* out16, int dt, void* stream) {
  LA_CHECK_ARG(img && out16, "la_im2col_patch: null pointer");
  LA_CHECK_ARG(Bn > 0 && patch > 0 && (patch % 8) == 0 && (S % patch) == 0, "la_im2col_patch: bad geometry S=%d patch=%d", S, patch);
  const int g = S / patch;
  const long total = (long)Bn * g * g * (3 * patch * patch / 8);
  int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_F16) hipLaunchKernelGGL(la::im2col_patch_kernel<la::f16_t>, dim3(blocks), dim3(256), 0, st, img, Bn, S, patch, (la::f16_t*)out16);
  else if (dt == LA_BF16) hipLaunchKernelGGL(la::im2col_patch_kernel<la::bf16_t>, dim3(blocks), dim3(256), 0, st, img, Bn, S, patch, (la::bf16_t*)out16);
  else if (dt == LA_F32) hipLaunchKernelGGL(la::im2col_patch_kernel<float>, dim3(blocks), dim3(256), 0, st, img, Bn, S, patch, (float*)out16);
  else if (dt == LA_F16X2)
    hipLaunchKernelGGL((la::im2col_patch_kernel<la::f16_t, true>), dim3(blocks), dim3(256), 0, st, img, Bn, S, patch, (la::f16_t*)out16);
  else LA_CHECK_ARG(false, "la_im2col_patch: bad dtype %d", dt);
  LA_CHECK_LAUNCH("la_im2col_patch");
  return 0;
}

extern "C" int la_im2col_3x3(const void* in16, int B, int H, int W, int C, void* out16, int dt, void* stream) {
  LA_CHECK_ARG(in16 && out16, "la_im2col_3x3: null pointer");
  const int per16 = (dt == LA_F32) ? 4 : 8;   // elements per 16-byte chunk
  const int planes = (dt == LA_F16X2) ? 2 : 1;
  LA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && (C % per16) == 0, "la_im2col_3x3: bad geometry (C=%d must be a multiple of %d)", C, per16);
  const long total = (long)B * H * W * planes * 9 * (C / per16);
  int blocks = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(la::im2col_3x3_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint4*>(in16), B, H, W, C / per16, planes, reinterpret_cast<uint4*>(out16));
  LA_CHECK_LAUNCH("la_im2col_3x3");
  return 0;
}
