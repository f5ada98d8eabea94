// la_conv3x3_split: 3 x 3 / pad 1 convolution of an NHWC fp32 map with 32 input and 32 output channels (the spatial convolutions of the
// mask decoder at D = 256: mask_decoder.py:236-255, three of them on the 4x-upscaled 256 x 256 map) as an implicit GEMM on the 16-bit
// MFMA in split precision: both operands enter as fp16 plane pairs (hi = rn16(x), lo = rn16(x - hi)) and every k-step issues three
// products A_hi W_hi + A_lo W_hi + A_hi W_lo (~22 mantissa bits, the accuracy class of LA_F16X2) - 54 MFMAs 32x32x16 per 32 pixels
// instead of the 288 exact-fp32 MFMAs 32x32x2 (64 cycles each) of la_conv3x3_f32, which is bound by the fp32 matrix pipe at ~0.75 ms
// per 48 x 256 x 256 x 32 map (0.2 of what its 0.8 GB of traffic would take).
//
// Workgroup = 8 x 32 output pixels, four waves x 2 output rows.  The 10 x 34 pixel halo is read ONCE from global memory (every input
// pixel 1.33 times per launch instead of 9 gathers), split into the two planes on the way into LDS (zero outside the image = the
// padding).  LDS image of a plane: [10 rows][34 pixels][64 B], the four 16-byte chunks of a pixel XOR-swizzled by (pixel >> 2) & 3: the 16
// lanes of a ds_read_b128 phase (16 consecutive pixels, one chunk) cover 16 distinct 16-byte slots of a 256-byte bank row.
// PERSISTENT: two workgroups per CU walk the tiles; the 32 x 288 weight lives in REGISTERS for the whole walk (a lane's B-operand
// fragments of all 18 k-steps of the hi plane: 72 VGPRs; the lo plane, used once per k-step, stays in LDS) - the first version re-split it per tile from L2 (36 KiB per workgroup = as many
// bytes as the pixels) and read it back from LDS per k-step, which put the LDS at its bandwidth.  61 KiB per workgroup, two per CU,
// one staging while the other computes.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int C3_TH = 8, C3_TW = 32, C3_HR = C3_TH + 2, C3_HC = C3_TW + 2;
constexpr int C3_AROW = C3_HC * 64;                 // bytes per halo row of one plane
constexpr int C3_APLANE = C3_HR * C3_AROW;          // 21760
constexpr int C3_WROW = 592;                        // bytes per cout row of the lo weight plane (288 fp16 + 16 B: 37 chunks, odd modulo 16)
constexpr int C3_WPLANE = 32 * C3_WROW;             // 18944
constexpr int C3_LDS = 2 * C3_APLANE + C3_WPLANE;   // 61 KiB per workgroup

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const f16_t h0 = (f16_t)v.x, h1 = (f16_t)v.y, h2 = (f16_t)v.z, h3 = (f16_t)v.w;
  hi.x = pack2<f16_t>((float)h0, (float)h1);
  hi.y = pack2<f16_t>((float)h2, (float)h3);
  lo.x = pack2<f16_t>(v.x - (float)h0, v.y - (float)h1);
  lo.y = pack2<f16_t>(v.z - (float)h2, v.w - (float)h3);
}

__global__ __launch_bounds__(256, 2) void conv3x3_split_kernel(const float* __restrict__ in, int B, int H, int W, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, float* __restrict__ out, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_hi = smem;
  char* a_lo = smem + C3_APLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntx = (W + C3_TW - 1) / C3_TW, nty = (H + C3_TH - 1) / C3_TH;

  // ---- weights: the lane's B-operand fragments of all 18 k-steps, both planes, in registers for the workgroup's whole tile walk
  //      (lane (fr, fh) holds W[cout = fr][k = 16 ks + 8 fh .. + 7]) - no weight traffic per tile
  uint4 wh[18];
  char* w_lo = smem + 2 * C3_APLANE;
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) {
    const float* wp = wt + (size_t)fr * 288 + ks * 16 + fh * 8;
    uint2 h0, l0, h1, l1;
    split4(*reinterpret_cast<const float4*>(wp), h0, l0);
    split4(*reinterpret_cast<const float4*>(wp + 4), h1, l1);
    wh[ks] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    // (the lo plane goes to LDS: 72 more registers would spill; every wave writes the same bytes - the first barrier of the walk orders them)
    *reinterpret_cast<uint4*>(w_lo + fr * C3_WROW + (ks * 16 + fh * 8) * 2) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
  const float bv = bias ? bias[fr] : 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (ntx * nty), t_ = tile % (ntx * nty);
    const int ty0 = (t_ / ntx) * C3_TH, tx0 = (t_ % ntx) * C3_TW;
    // ---- halo: 10 x 34 pixels x 32 channels, split into the two planes on the way into LDS (zero outside the image) ----------------
    const float* img = in + (size_t)b * H * W * 32;
    // (2720 float4 per tile = 10.6 per thread: two batches of six requests in flight per thread - a rolled loop issued them one at a time)
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      float4 v[6];
      int off[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = tid + (bt * 6 + u) * 256;
        const int pix = i >> 3, c4 = i & 7;
        const int hy = pix / C3_HC, hx = pix % C3_HC;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        off[u] = i < C3_HR * C3_HC * 8 ? hy * C3_AROW + hx * 64 + ((((c4 >> 1) ^ (hx >> 2)) & 3) << 4) + (c4 & 1) * 8 : -1;
        if (off[u] >= 0 && y >= 0 && y < H && x >= 0 && x < W) v[u] = *reinterpret_cast<const float4*>(img + ((size_t)y * W + x) * 32 + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        uint2 hi, lo;
        split4(v[u], hi, lo);
        if (off[u] >= 0) {
          *reinterpret_cast<uint2*>(a_hi + off[u]) = hi;
          *reinterpret_cast<uint2*>(a_lo + off[u]) = lo;
        }
      }
    }
    __syncthreads();

    // ---- 18 k-steps of 16 (tap = ks / 2, channels 16 (ks & 1) ..): wave w owns output rows 2 w, 2 w + 1 ----------------------------
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int tap = ks >> 1, ky = tap / 3, kx = tap % 3;
      const int hx = kx + fr;
      const int ao = hx * 64 + (((((ks & 1) * 2 + fh) ^ (hx >> 2)) & 3) << 4);
      const uint4 wl = *reinterpret_cast<const uint4*>(w_lo + fr * C3_WROW + (ks * 16 + fh * 8) * 2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int off = (2 * wave + t + ky) * C3_AROW + ao;
        const uint4 ah = *reinterpret_cast<const uint4*>(a_hi + off);
        const uint4 al = *reinterpret_cast<const uint4*>(a_lo + off);
        acc[t] = Half16<f16_t>::mfma32(al, wh[ks], acc[t]);
        acc[t] = Half16<f16_t>::mfma32(ah, wl, acc[t]);
        acc[t] = Half16<f16_t>::mfma32(ah, wh[ks], acc[t]);
      }
    }
    __syncthreads();      // every wave is done with the planes: the next tile's halo may land
    // ---- store: accumulator register r of lane (fr, fh) = pixel (r & 3) + 8 (r >> 2) + 4 fh, output channel fr ----------------------
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
  static int ncu_of[64] = {0};                        // per device: a process may drive several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& ncu = ncu_of[dev & 63];
  if (ncu == 0 && (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)) ncu = 256;
  const long grid = tiles < 2l * ncu ? tiles : 2l * ncu;      // persistent: two workgroups per CU walk the tiles (weights stay in registers)
  hipLaunchKernelGGL(la::conv3x3_split_kernel, dim3((unsigned)grid), dim3(256), la::C3_LDS, reinterpret_cast<hipStream_t>(stream), in, B, H, W, wt,
                     bias, out32, (int)tiles);
  LA_CHECK_LAUNCH("la_conv3x3_split");
  return 0;
}
