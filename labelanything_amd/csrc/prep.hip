// Image preprocessing on the device (the step before the path): the reference resizes PIL images with
// torchvision.transforms.functional.resize (data/transforms.py:14-25 -> PIL Image.resize, BILINEAR = antialiased triangle
// filter in 8-bit fixed point), then ToTensor -> (x - mean) / std -> zero pad to S x S (data/transforms.py:28-46).
//   la_resample_u8     one axis of PIL's two-pass resample (ImagingResampleHorizontal/Vertical_8bpc): integer coefficients with
//                      22 fractional bits computed on the host exactly like precompute_coeffs + normalize_coeffs_8bpc,
//                      ss = 2^21 + sum(pixel * k), out = clip8(ss >> 22) - bit-exact with PIL.
//   la_u8_to_chw_norm  uint8 HWC -> fp32 CHW: x / 255, (x - mean) / std in IEEE fp32 (correctly rounded divisions, like
//                      torch on the CPU), zero padding on the right / bottom.
// Pure byte streaming: one thread per output element.
#include "la_common.h"
#include "../../include/la_hip.h"

#pragma clang fp contract(off)

namespace la {

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

// in [n_outer, in_size, inner] -> out [n_outer, out_size, inner]; (axis 0 of an [H, W*C] view = vertical; for the horizontal
// pass n_outer = H, in_size = W, inner = C)
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ in, long n_outer, int in_size, int inner,
                                                          int out_size, const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          int ksize, unsigned char* __restrict__ out) {
  const long total = n_outer * out_size * inner;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % inner);
    const int o = (int)((i / inner) % out_size);
    const long n = i / ((long)inner * out_size);
    const int x0 = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* k = kk + (long)o * ksize;
    const unsigned char* p = in + (n * in_size + x0) * inner + c;
    int ss = 1 << (PIL_PRECISION_BITS - 1);
    for (int j = 0; j < cnt; ++j) ss += (int)p[(long)j * inner] * k[j];
    ss >>= PIL_PRECISION_BITS;
    out[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

__global__ __launch_bounds__(256) void u8_to_chw_norm_kernel(const unsigned char* __restrict__ in, int h, int w, int SH, int SW, float m0,
                                                             float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
  const long total = 3L * SH * SW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % SW), y = (int)((i / SW) % SH), c = (int)(i / ((long)SW * SH));
    float v = 0.f;
    if (y < h && x < w) {
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = ((float)in[((long)y * w + x) * 3 + c] / 255.0f - mean) / sd;
    }
    out[i] = v;
  }
}

}  // namespace la

extern "C" int la_resample_u8(const unsigned char* in, long n_outer, int in_size, int inner, int out_size, const int* bounds, const int* kk,
                              int ksize, unsigned char* out, void* stream) {
  LA_CHECK_ARG(in && bounds && kk && out, "la_resample_u8: null pointer");
  LA_CHECK_ARG(n_outer > 0 && in_size > 0 && inner > 0 && out_size > 0 && ksize > 0, "la_resample_u8: bad shape");
  const long total = n_outer * out_size * inner;
  long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(la::resample_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, n_outer, in_size, inner, out_size,
                     bounds, kk, ksize, out);
  LA_CHECK_LAUNCH("la_resample_u8");
  return 0;
}

extern "C" int la_u8_to_chw_norm(const unsigned char* in, int h, int w, int SH, int SW, const float* mean3, const float* std3, float* out,
                                 void* stream) {
  LA_CHECK_ARG(in && mean3 && std3 && out, "la_u8_to_chw_norm: null pointer");
  LA_CHECK_ARG(h > 0 && w > 0 && SH >= h && SW >= w, "la_u8_to_chw_norm: the %d x %d image does not fit the %d x %d canvas", h, w, SH, SW);
  const long total = 3L * SH * SW;
  long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(la::u8_to_chw_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, h, w, SH, SW, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], out);
  LA_CHECK_LAUNCH("la_u8_to_chw_norm");
  return 0;
}
