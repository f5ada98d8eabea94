// Micro-benchmark: peak global -> LDS streaming rate of LDS-DMA (global_load_lds_dwordx4) per CU on gfx950, source resident
// in L2 / Infinity Cache.  Each wave streams 1 KiB pieces (64 lanes x 16 B) from a per-workgroup window of a large buffer.
//   hipcc --offload-arch=gfx950 -O3 -I../../labelanything_amd/csrc -I../../include dma_bw.hip -o dma_bw && ./dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "la_common.h"

using namespace la;

// MODE 0: 8 rows x 128 B per piece (full cache lines), MODE 1: 16 rows x 64 B (half lines, like the BK = 32 kernels)
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(512) void dma_stream(const char* __restrict__ src, size_t window, int row_stride, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const unsigned lds0 = lds_addr_of(smem) + wave * (INFLIGHT * 1024);
  const char* base = src + (size_t)blockIdx.x * window;
  unsigned off;
  if (MODE == 0) off = (lane >> 3) * row_stride + (lane & 7) * 16;
  else off = (lane >> 2) * row_stride + (lane & 3) * 16;
  const int rows_per_piece = MODE == 0 ? 8 : 16;
  const size_t piece_span = (size_t)rows_per_piece * row_stride;
  const size_t npieces = window / piece_span;
  size_t p = wave;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < INFLIGHT; ++i) {
      const unsigned po = (unsigned)((p % npieces) * piece_span + (MODE == 1 ? ((p / npieces) & 1) * 64 : 0));
      dma16s(base, off + po, lds0 + i * 1024);
      p += nw;
    }
    dma_wait<INFLIGHT / 2>();
  }
  dma_wait<0>();
  if (smem[tid] == 123 && iters < 0) ((char*)src)[0] = 1;
}

int main() {
  const size_t bytes = 256ull << 20;
  char* d;
  hipMalloc(&d, bytes);
  hipMemset(d, 1, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int blocks, int threads, int inflight, size_t window, int row_stride, int piece_rows) {
    const int iters = 400;
    const size_t lds = (size_t)(threads / 64) * inflight * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, window, row_stride, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, window, row_stride, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)blocks * (threads / 64) * iters * inflight * 1024.0;
    printf("%-44s blocks %4d x %3d thr  window %6zu KiB  stride %5d: %7.2f TB/s  (%5.1f B/clk/CU at 2.4 GHz, %d blocks/CU)\n", name, blocks, threads,
           window >> 10, row_stride, total / (ms * 1e-3) / 1e12, total / (ms * 1e-3) / 256 / 2.4e9, blocks / 256);
  };
  // per-workgroup windows small enough to live in L2 (256 x 64 KiB = 16 MiB) and bigger (Infinity Cache)
  for (size_t win : {size_t(64) << 10, size_t(512) << 10}) {
    run(dma_stream<0, 8>, "full lines 8x128B, 8 in flight/wave", 256, 512, 8, win, 1536, 8);
    run(dma_stream<0, 16>, "full lines 8x128B, 16 in flight/wave", 256, 512, 16, win, 1536, 8);
    run(dma_stream<0, 8>, "full lines, 2 blocks/CU x 4 waves", 512, 256, 8, win / 2, 1536, 8);
    run(dma_stream<1, 8>, "half lines 16x64B, 8 in flight/wave", 256, 512, 8, win, 1536, 16);
    run(dma_stream<1, 16>, "half lines 16x64B, 16 in flight/wave", 256, 512, 16, win, 1536, 16);
    run(dma_stream<0, 8>, "full lines, contiguous rows (stride 128)", 256, 512, 8, win, 128, 8);
  }
  return 0;
}
