// Micro-benchmark: L2-resident global -> LDS streaming per CU on gfx950 by instruction form (1 workgroup per CU, NW waves):
//   F0 global_load_lds_dwordx4 (SGPR base + VGPR offset)      F1 buffer_load_dwordx4 ... offen lds (SRD + VGPR offset + SGPR soffset)
//   F2 global_load_dwordx4 into VGPRs (no LDS write)            F3 global_load_dwordx4 + ds_write_b128
//   hipcc --offload-arch=gfx950 -O3 dma_forms.hip -o dma_forms && ./dma_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int INFLIGHT>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, size_t window, int row_stride, int iters, char* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (INFLIGHT * 1024);
  const char* base = src + (size_t)blockIdx.x * window;
  const unsigned off = (lane >> 3) * row_stride + (lane & 7) * 16;
  const unsigned piece_span = 8u * row_stride;
  const unsigned npieces = (unsigned)(window / piece_span);
  unsigned p = wave;
  u32x4 acc = {0, 0, 0, 0};
  // SRD for F1
  const uint64_t b64 = (uint64_t)base;
  u32x4 srd = {(uint32_t)b64, (uint32_t)(b64 >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  srd.x = __builtin_amdgcn_readfirstlane(srd.x);
  srd.y = __builtin_amdgcn_readfirstlane(srd.y);
  for (int it = 0; it < iters; ++it) {
    if (FORM == 0) {
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) {
        const unsigned po = (p % npieces) * piece_span;
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off + po), "s"(base), "s"(lds0), "n"(0) : "memory", "scc");
        p += nw;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
    } else if (FORM == 1) {
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) {
        const unsigned po = __builtin_amdgcn_readfirstlane((p % npieces) * piece_span);
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %4 offen lds" ::"v"(off), "s"(srd), "s"(lds0), "n"(0), "s"(po) : "memory", "scc");
        p += nw;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
    } else {
      u32x4 v[INFLIGHT];
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) {
        const unsigned po = (p % npieces) * piece_span;
        v[i] = *reinterpret_cast<const u32x4*>(base + off + po);
        p += nw;
      }
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) {
        if (FORM == 3) *reinterpret_cast<u32x4*>(smem + wave * (INFLIGHT * 1024) + i * 1024 + lane * 16) = v[i];
        else acc ^= v[i];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (FORM == 2 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
  if (smem[tid] == 123 && iters < 0) sink[0] = 1;
}

int main() {
  const size_t bytes = 256ull << 20;
  char* d;
  hipMalloc(&d, bytes);
  hipMemset(d, 1, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int threads, int inflight, size_t window, int row_stride) {
    const int iters = 400, blocks = 256;
    const size_t lds = (size_t)(threads / 64) * inflight * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, window, row_stride, iters, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, window, row_stride, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)blocks * (threads / 64) * iters * inflight * 1024.0;
    printf("%-46s %d waves  window %4zu KiB stride %5d: %6.2f TB/s = %5.1f GB/s/CU (%4.1f B/clk/CU at 2.0 GHz)\n", name, threads / 64, window >> 10,
           row_stride, total / (ms * 1e-3) / 1e12, total / (ms * 1e-3) / 256 / 1e9, total / (ms * 1e-3) / 256 / 2.0e9);
  };
  for (int threads : {256, 512}) {
    for (size_t win : {size_t(64) << 10, size_t(512) << 10}) {
      run(stream<0, 16>, "F0 global_load_lds saddr, 16 in flight", threads, 16, win, 1536);
      run(stream<1, 16>, "F1 buffer_load offen lds, 16 in flight", threads, 16, win, 1536);
      run(stream<2, 16>, "F2 global_load_dwordx4 -> VGPR, 16 in flight", threads, 16, win, 1536);
      run(stream<3, 16>, "F3 global_load_dwordx4 + ds_write_b128", threads, 16, win, 1536);
      run(stream<0, 8>, "F0 global_load_lds saddr, 8 in flight", threads, 8, win, 1536);
      run(stream<1, 8>, "F1 buffer_load offen lds, 8 in flight", threads, 8, win, 1536);
    }
  }
  return 0;
}
