// Can one wave's MFMA stream and ANOTHER wave's VALU stream share a SIMD?  8 waves per workgroup, one workgroup per CU: waves 0-3 (one
// per SIMD) issue dependent-free 32x32x16 f16 MFMAs, waves 4-7 (the second wave of each SIMD) issue packed FMAs / v_exp_f32.
// Prints the time of each stream alone and of both together.   hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>   // VALU stream: 0 v_pk_fma_f32, 1 v_exp_f32, 2 v_fma_f32
__global__ __launch_bounds__(512, 2) void probe(float* out, int n_mfma, int n_valu, int mode) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < n_mfma; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    if (!(mode & 2)) return;
    f2 x0 = {threadIdx.x * 1e-3f, 0.5f}, x1 = x0 * 1.1f, x2 = x0 * 1.2f, x3 = x0 * 1.3f, x4 = x0 * 1.4f, x5 = x0 * 1.5f, x6 = x0 * 1.6f, x7 = x0 * 1.7f;
    const f2 m = {1.0001f, 0.9999f}, ad = {1e-4f, -1e-4f};
    for (int i = 0; i < n_valu; ++i) {
      if (KIND == 0) {
        x0 = __builtin_elementwise_fma(x0, m, ad); x1 = __builtin_elementwise_fma(x1, m, ad); x2 = __builtin_elementwise_fma(x2, m, ad);
        x3 = __builtin_elementwise_fma(x3, m, ad); x4 = __builtin_elementwise_fma(x4, m, ad); x5 = __builtin_elementwise_fma(x5, m, ad);
        x6 = __builtin_elementwise_fma(x6, m, ad); x7 = __builtin_elementwise_fma(x7, m, ad);
      } else if (KIND == 1) {
        x0.x = __builtin_amdgcn_exp2f(x0.x); x1.x = __builtin_amdgcn_exp2f(x1.x); x2.x = __builtin_amdgcn_exp2f(x2.x); x3.x = __builtin_amdgcn_exp2f(x3.x);
        x4.x = __builtin_amdgcn_exp2f(x4.x); x5.x = __builtin_amdgcn_exp2f(x5.x); x6.x = __builtin_amdgcn_exp2f(x6.x); x7.x = __builtin_amdgcn_exp2f(x7.x);
      } else {
        x0.x = fmaf(x0.x, 1.0001f, 1e-4f); x1.x = fmaf(x1.x, 1.0001f, 1e-4f); x2.x = fmaf(x2.x, 1.0001f, 1e-4f); x3.x = fmaf(x3.x, 1.0001f, 1e-4f);
        x4.x = fmaf(x4.x, 1.0001f, 1e-4f); x5.x = fmaf(x5.x, 1.0001f, 1e-4f); x6.x = fmaf(x6.x, 1.0001f, 1e-4f); x7.x = fmaf(x7.x, 1.0001f, 1e-4f);
      }
    }
    const f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y;
  }
}

template <int KIND>
static float run(float* d, int nm, int nv, int mode) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, d, nm, nv, mode);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, d, nm, nv, mode);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int nm = 20000;                  // 80000 MFMAs per wave = 2.56 M cycles
  const char* names[3] = {"v_pk_fma_f32", "v_exp_f32", "v_fma_f32"};
  for (int k = 0; k < 3; ++k) {
    const int nv = k == 1 ? 20000 : 40000;
    float tm, tv, tb;
    if (k == 0) { tm = run<0>(d, nm, nv, 1); tv = run<0>(d, nm, nv, 2); tb = run<0>(d, nm, nv, 3); }
    else if (k == 1) { tm = run<1>(d, nm, nv, 1); tv = run<1>(d, nm, nv, 2); tb = run<1>(d, nm, nv, 3); }
    else { tm = run<2>(d, nm, nv, 1); tv = run<2>(d, nm, nv, 2); tb = run<2>(d, nm, nv, 3); }
    printf("%-14s MFMA alone %8.1f us (%.1f cycles/MFMA at 2.4 GHz)   VALU alone %8.1f us (%.2f cycles/instr)   both %8.1f us   (sum %.1f, max %.1f)\n", names[k], tm,
           tm * 2400.0 / (4.0 * nm), tv, tv * 2400.0 / (8.0 * nv), tb, tm + tv, tm > tv ? tm : tv);
  }
  return 0;
}
