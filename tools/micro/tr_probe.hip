// ds_read_b64_tr_b16 semantics probe: every lane passes its own address; prints which LDS element (index in 16-bit units) each lane's
// four result elements came from.  hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* lane_addr_bytes, uint16_t* out) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)lds;
  const unsigned addr = base + (unsigned)lane_addr_bytes[threadIdx.x];
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff;
  out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff;
  out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int h_addr[64];
  uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: linear (lane l -> byte 8 l); pass 1: lane p = 4 j + c of a 16-lane group g -> row (4 g + j) of 128 bytes, 8-byte chunk c
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, p = l & 15, j = p >> 2, c = p & 3;
      h_addr[l] = pass == 0 ? 8 * l : (4 * g + j) * 128 + c * 8;
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pass %d\n", pass);
    for (int l = 0; l < 64; ++l) printf("lane %2d (addr %4d B = elem %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
