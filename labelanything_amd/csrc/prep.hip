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

// torch "nearest" source index (ATen UpSampleKernel nearest_idx): identity / exact halving shortcuts, else
// min(floorf(dst * (float)in / out), in - 1) in fp32
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
  if (out_size == in_size) return dst;
  if (out_size == 2 * in_size) return dst >> 1;
  const float scale = (float)in_size / (float)out_size;
  return min((int)floorf((float)dst * scale), in_size - 1);
}

// PromptsProcessor.apply_masks (data/transforms.py:203-224) for P prompt slots at once: OR of the slot's instance masks
// (u8 [H, W] each, listed by [first, first + count) in `index`), nearest resize to (nh, nw), zero pad to S x S, nearest resize
// to Mo x Mo - composed per output pixel.  nh == 0: no custom preprocessing (one resize (H, W) -> (Mo, Mo)).
// out fp32 [P, Mo, Mo] in {0, 1}; flags u8 [P] = any pixel set (annotations_to_tensor, data/utils.py:219-223).
__global__ __launch_bounds__(256) void prompt_mask_kernel(const unsigned char* __restrict__ masks, const int* __restrict__ first,
                                                          const int* __restrict__ count, const int* __restrict__ index, int H, int W, int nh,
                                                          int nw, int S, int Mo, float* __restrict__ out, unsigned char* __restrict__ flags) {
  const int p = blockIdx.y;
  const int f = first[p], n = count[p];
  __shared__ int any_set;
  if (threadIdx.x == 0) any_set = 0;
  __syncthreads();
  int mine = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Mo * Mo; i += gridDim.x * 256) {
    const int ox = i % Mo, oy = i / Mo;
    int sy, sx;
    bool inside = true;
    if (nh > 0) {
      const int py = nearest_src(oy, S, Mo), px = nearest_src(ox, S, Mo);      // position on the padded S x S canvas
      inside = py < nh && px < nw;
      sy = nearest_src(min(py, nh - 1), H, nh);
      sx = nearest_src(min(px, nw - 1), W, nw);
    } else {
      sy = nearest_src(oy, H, Mo);
      sx = nearest_src(ox, W, Mo);
    }
    int v = 0;
    if (inside)
      for (int k = 0; k < n && !v; ++k) v = masks[((size_t)index[f + k] * H + sy) * W + sx] != 0;
    out[(size_t)p * Mo * Mo + i] = (float)v;
    mine |= v;
  }
  if (mine) atomicOr(&any_set, 1);
  __syncthreads();
  if (threadIdx.x == 0 && any_set) flags[p] = 1;        // flags are zeroed by the caller; several blocks may set the same 1
}

}  // namespace la

extern "C" int la_prompt_masks(const unsigned char* masks, const int* first, const int* count, const int* index, int P, int H, int W, int nh,
                               int nw, int S, int Mo, float* out, unsigned char* flags, void* stream) {
  LA_CHECK_ARG(masks && first && count && index && out && flags, "la_prompt_masks: null pointer");
  LA_CHECK_ARG(P > 0 && H > 0 && W > 0 && Mo > 0 && (nh == 0 || (nh > 0 && nw > 0 && S >= nh && S >= nw)), "la_prompt_masks: bad geometry");
  hipLaunchKernelGGL(la::prompt_mask_kernel, dim3(16, P), dim3(256), 0, (hipStream_t)stream, masks, first, count, index, H, W, nh, nw, S, Mo, out,
                     flags);
  LA_CHECK_LAUNCH("la_prompt_masks");
  return 0;
}

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
