// la_conv3x3_split: 3 x 3 / pad 1 convolution of an NHWC fp32 map with 32 input and 32 output channels (the spatial convolutions of the
// mask decoder at D = 256: mask_decoder.py:236-255, three of them on the 4x-upscaled 256 x 256 map) as an implicit GEMM on the 16-bit
// MFMA in split precision: both operands enter as fp16 plane pairs (hi = rn16(x), lo = rn16(x - hi)) and every k-step issues three
// products A_hi W_hi + A_lo W_hi + A_hi W_lo (~22 mantissa bits, the accuracy class of LA_F16X2) - 54 MFMAs 32x32x16 per 32 pixels
// instead of the 288 exact-fp32 MFMAs 32x32x2 (64 cycles each) of la_conv3x3_f32, which is bound by the fp32 matrix pipe at ~0.75 ms
// per 48 x 256 x 256 x 32 map (0.2 of what its 0.8 GB of traffic would take).
//
// Workgroup = 8 x 32 output pixels, four waves x 2 output rows.  The 10 x 34 pixel halo is read ONCE from global memory (every input
// pixel 1.33 times per launch instead of 9 gathers), split into the two planes on the way into LDS (zero outside the image = the
// padding); the 32 x 288 weight is split the same way per workgroup (36 KiB from L2).  LDS images:
//   A plane: [10 rows][34 pixels][64 B], the four 16-byte chunks of a pixel XOR-swizzled by (pixel >> 2) & 3: the 16 lanes of a ds_read_b128
//            phase (16 consecutive pixels, one chunk) cover 16 distinct 16-byte slots of a 256-byte bank row
//   W plane: [32 couts][592 B] (288 fp16 + 16 B of padding: 37 chunks per row, 37 mod 16 = 5 is odd - 16 consecutive rows, 16 slots)
// 79.5 KiB per workgroup: two per CU, one staging while the other computes.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int C3_TH = 8, C3_TW = 32, C3_HR = C3_TH + 2, C3_HC = C3_TW + 2;
constexpr int C3_AROW = C3_HC * 64;                 // bytes per halo row of one plane
constexpr int C3_APLANE = C3_HR * C3_AROW;          // 21760
constexpr int C3_WROW = 592;
constexpr int C3_WPLANE = 32 * C3_WROW;             // 18944
constexpr int C3_LDS = 2 * C3_APLANE + 2 * C3_WPLANE;

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const f16_t h0 = (f16_t)v.x, h1 = (f16_t)v.y, h2 = (f16_t)v.z, h3 = (f16_t)v.w;
  hi.x = pack2<f16_t>((float)h0, (float)h1);
  hi.y = pack2<f16_t>((float)h2, (float)h3);
  lo.x = pack2<f16_t>(v.x - (float)h0, v.y - (float)h1);
  lo.y = pack2<f16_t>(v.z - (float)h2, v.w - (float)h3);
}

__global__ __launch_bounds__(256, 2) void conv3x3_split_kernel(const float* __restrict__ in, int B, int H, int W, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_hi = smem;
  char* a_lo = smem + C3_APLANE;
  char* w_hi = smem + 2 * C3_APLANE;
  char* w_lo = w_hi + C3_WPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntx = (W + C3_TW - 1) / C3_TW, nty = (H + C3_TH - 1) / C3_TH;
  const int b = blockIdx.x / (ntx * nty), t_ = blockIdx.x % (ntx * nty);
  const int ty0 = (t_ / ntx) * C3_TH, tx0 = (t_ % ntx) * C3_TW;

  // ---- weights: [32][288] fp32 -> two fp16 planes ------------------------------------------------------------------------------------
  for (int i = tid; i < 32 * 72; i += 256) {
    const int co = i / 72, k4 = i % 72;
    uint2 hi, lo;
    split4(*reinterpret_cast<const float4*>(wt + (size_t)co * 288 + k4 * 4), hi, lo);
    *reinterpret_cast<uint2*>(w_hi + co * C3_WROW + k4 * 8) = hi;
    *reinterpret_cast<uint2*>(w_lo + co * C3_WROW + k4 * 8) = lo;
  }
  // ---- halo: 10 x 34 pixels x 32 channels --------------------------------------------------------------------------------------------
  const float* img = in + (size_t)b * H * W * 32;
  for (int i = tid; i < C3_HR * C3_HC * 8; i += 256) {
    const int pix = i >> 3, c4 = i & 7;
    const int hy = pix / C3_HC, hx = pix % C3_HC;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < H && x >= 0 && x < W) v = *reinterpret_cast<const float4*>(img + ((size_t)y * W + x) * 32 + c4 * 4);
    uint2 hi, lo;
    split4(v, hi, lo);
    const int off = hy * C3_AROW + hx * 64 + ((((c4 >> 1) ^ (hx >> 2)) & 3) << 4) + (c4 & 1) * 8;
    *reinterpret_cast<uint2*>(a_hi + off) = hi;
    *reinterpret_cast<uint2*>(a_lo + off) = lo;
  }
  __syncthreads();

  // ---- 18 k-steps of 16 (tap = ks / 2, channels 16 (ks & 1) ..): wave w owns output rows 2 w, 2 w + 1 ------------------------------
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) {
    const int tap = ks >> 1, ky = tap / 3, kx = tap % 3;
    const int wo = fr * C3_WROW + (ks * 16 + fh * 8) * 2;
    const uint4 wh = *reinterpret_cast<const uint4*>(w_hi + wo);
    const uint4 wl = *reinterpret_cast<const uint4*>(w_lo + wo);
    const int hx = kx + fr;
    const int ao = hx * 64 + (((((ks & 1) * 2 + fh) ^ (hx >> 2)) & 3) << 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int off = (2 * wave + t + ky) * C3_AROW + ao;
      const uint4 ah = *reinterpret_cast<const uint4*>(a_hi + off);
      const uint4 al = *reinterpret_cast<const uint4*>(a_lo + off);
      acc[t] = Half16<f16_t>::mfma32(al, wh, acc[t]);
      acc[t] = Half16<f16_t>::mfma32(ah, wl, acc[t]);
      acc[t] = Half16<f16_t>::mfma32(ah, wh, acc[t]);
    }
  }
  // ---- store: accumulator register r of lane (fr, fh) = pixel (r & 3) + 8 (r >> 2) + 4 fh, output channel fr --------------------------
  const float bv = bias ? bias[fr] : 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int y = ty0 + 2 * wave + t;
    if (y >= H) continue;
    float* orow = out + (((size_t)b * H + y) * W + tx0) * 32 + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (tx0 + px < W) orow[(size_t)px * 32] = acc[t][r] + bv;
    }
  }
}

}  // namespace la

extern "C" int la_conv3x3_split_ok(int Cin, int Cout) { return Cin == 32 && Cout == 32; }

extern "C" int la_conv3x3_split(const float* in, int B, int H, int W, int Cin, const float* wt, const float* bias, int Cout, float* out32,
                                void* stream) {
  LA_CHECK_ARG(in && wt && out32, "la_conv3x3_split: null pointer");
  LA_CHECK_ARG(B > 0 && H > 0 && W > 0 && la_conv3x3_split_ok(Cin, Cout), "la_conv3x3_split: 32 input and 32 output channels only (Cin=%d Cout=%d)", Cin,
               Cout);
  LA_CHECK_ARG(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(wt)) & 15) == 0, "la_conv3x3_split: 16-byte aligned input / weight");
  const long tiles = (long)B * ((H + la::C3_TH - 1) / la::C3_TH) * ((W + la::C3_TW - 1) / la::C3_TW);
  LA_CHECK_ARG(tiles < (1l << 31), "la_conv3x3_split: too many tiles");
  static unsigned long long attr_mask = 0;
  la::ensure_dyn_lds(reinterpret_cast<const void*>(la::conv3x3_split_kernel), la::C3_LDS, attr_mask);
  hipLaunchKernelGGL(la::conv3x3_split_kernel, dim3((unsigned)tiles), dim3(256), la::C3_LDS, reinterpret_cast<hipStream_t>(stream), in, B, H, W, wt,
                     bias, out32);
  LA_CHECK_LAUNCH("la_conv3x3_split");
  return 0;
}
