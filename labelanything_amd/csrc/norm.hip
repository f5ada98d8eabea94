// la_layernorm: row LayerNorm (biased variance) with fused residual add, GELU, positional-encoding add,
// 16-bit down-conversion and SAM window partitioning of the output rows.  Pure HBM streaming: one pass over
// x (+x2), rows live in registers, reductions by wave shuffles.  A row is handled by LPR lanes
// (LPR = 64 for E >= 256, fewer for narrow rows so no lane idles), 256 threads per workgroup.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int LN_MAXV = 8;  // float4 per lane -> E <= 4 * 64 * 8 = 2048

struct LnArgs {
  const float* x;
  const float* x2;
  int ldx, rows, E;
  const float* gamma;
  const float* beta;
  float eps;
  int gelu;
  float* out32;
  void* out16;
  void* out16_pe;
  const float* pe;
  int pe_mod;
  int window, H, W;
  int split;          // LA_F16X2: the 16-bit outputs are [hi | lo] plane pairs (row stride 2 E)
};

template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d) { store4v<T>(p, a, b, c, d); }

// NVT = float4 vectors per lane (compile time, so the row loads are issued back to back without per-vector predicates);
// EXACT = every lane has all NVT vectors (E == 4 * LPR * NVT), otherwise the tail vectors are predicated.
template <typename T, int LPR, int NVT, bool EXACT>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int tid = threadIdx.x;
  const int rl = tid / LPR, lane = tid % LPR;
  const int nv = a.E >> 2;
  for (int row = blockIdx.x * RPB + rl; row < a.rows; row += gridDim.x * RPB) {
    float4 v[NVT];
    const float4* xp = reinterpret_cast<const float4*>(a.x + (size_t)row * a.ldx);
    const float4* yp = a.x2 ? reinterpret_cast<const float4*>(a.x2 + (size_t)row * a.ldx) : nullptr;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
      const int c = lane + i * LPR;
      if (EXACT || c < nv) {
        float4 t = xp[c];
        if (yp) {
          const float4 u = yp[c];
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        v[i] = t;
        s += (t.x + t.y) + (t.z + t.w);
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    s = wave_sum(s, LPR);
    const float mean = s / (float)a.E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
      const int c = lane + i * LPR;
      if (EXACT || c < nv) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    q = wave_sum(q, LPR);
    const float rstd = 1.0f / sqrtf(q / (float)a.E + a.eps);

    int drow = row;
    if (a.window > 0) {  // (b, y, x) -> window-partitioned row
      const int ws = a.window;
      const int x = row % a.W, y = (row / a.W) % a.H, b = row / (a.W * a.H);
      const int nwx = (a.W + ws - 1) / ws, nwy = (a.H + ws - 1) / ws;
      drow = ((b * nwy + y / ws) * nwx + x / ws) * ws * ws + (y % ws) * ws + (x % ws);
    }
    T* o16 = reinterpret_cast<T*>(a.out16);
    T* o16pe = reinterpret_cast<T*>(a.out16_pe);
    const float4* gp = reinterpret_cast<const float4*>(a.gamma);
    const float4* bp = reinterpret_cast<const float4*>(a.beta);
    const float4* pp = a.pe ? reinterpret_cast<const float4*>(a.pe + (size_t)(a.pe_mod ? row % a.pe_mod : row) * a.E) : nullptr;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
      const int c = lane + i * LPR;
      if (EXACT || c < nv) {
        const float4 g = gp[c], be = bp[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + be.x;
        o.y = (v[i].y - mean) * rstd * g.y + be.y;
        o.z = (v[i].z - mean) * rstd * g.z + be.z;
        o.w = (v[i].w - mean) * rstd * g.w + be.w;
        if (a.gelu) {
          o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w);
        }
        if (a.out32) reinterpret_cast<float4*>(a.out32 + (size_t)row * a.E)[c] = o;
        if (o16) {
          if (a.split) store4_split<T>(o16 + (size_t)drow * 2 * a.E, a.E, c * 4, o.x, o.y, o.z, o.w);
          else store4<T>(o16 + (size_t)drow * a.E + c * 4, o.x, o.y, o.z, o.w);
        }
        if (o16pe) {
          const float4 p = pp[c];
          if (a.split) store4_split<T>(o16pe + (size_t)drow * 2 * a.E, a.E, c * 4, o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
          else store4<T>(o16pe + (size_t)drow * a.E + c * 4, o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
        }
      }
    }
  }
}

template <typename T, int LPR, int NVT>
static void launch_ln_nv(const LnArgs& a, int blocks, hipStream_t st) {
  if ((a.E >> 2) == LPR * NVT) hipLaunchKernelGGL((layernorm_kernel<T, LPR, NVT, true>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((layernorm_kernel<T, LPR, NVT, false>), dim3(blocks), dim3(256), 0, st, a);
}

template <typename T>
static void launch_ln(const LnArgs& a, hipStream_t st) {
  const int nv = a.E >> 2;
  int lpr = 64;
  while (lpr > 1 && (lpr >> 1) >= nv) lpr >>= 1;
  const int rpb = 256 / lpr;
  int blocks = (a.rows + rpb - 1) / rpb;
  if (blocks > 8192) blocks = 8192;
  if (lpr == 64) {
    const int nvt = (nv + 63) / 64;
    if (nvt == 1) launch_ln_nv<T, 64, 1>(a, blocks, st);
    else if (nvt == 2) launch_ln_nv<T, 64, 2>(a, blocks, st);
    else if (nvt == 3) launch_ln_nv<T, 64, 3>(a, blocks, st);        // E = 768
    else if (nvt == 4) launch_ln_nv<T, 64, 4>(a, blocks, st);        // E = 1024
    else if (nvt == 5) launch_ln_nv<T, 64, 5>(a, blocks, st);        // E = 1280
    else launch_ln_nv<T, 64, LN_MAXV>(a, blocks, st);
    return;
  }
  switch (lpr) {          // narrow rows: one vector per lane
    case 32: launch_ln_nv<T, 32, 1>(a, blocks, st); break;
    case 16: launch_ln_nv<T, 16, 1>(a, blocks, st); break;
    case 8: launch_ln_nv<T, 8, 1>(a, blocks, st); break;
    case 4: launch_ln_nv<T, 4, 1>(a, blocks, st); break;
    case 2: launch_ln_nv<T, 2, 1>(a, blocks, st); break;
    default: launch_ln_nv<T, 1, 1>(a, blocks, st); break;
  }
}

}  // namespace la

extern "C" int la_layernorm(const float* x, const float* x2, int ldx, int rows, int E, const float* gamma, const float* beta,
                            float eps, int gelu, float* out32, void* out16, void* out16_pe, const float* pe, int pe_mod,
                            int window, int H, int W, int dt, void* stream) {
  LA_CHECK_ARG(x && gamma && beta, "la_layernorm: null pointer");
  LA_CHECK_ARG(rows > 0 && E > 0 && (E % 4) == 0 && E <= 4 * 64 * la::LN_MAXV && (ldx % 4) == 0,
               "la_layernorm: bad shape rows=%d E=%d ldx=%d", rows, E, ldx);
  LA_CHECK_ARG(out32 || out16 || out16_pe, "la_layernorm: no output");
  LA_CHECK_ARG(!out16_pe || pe, "la_layernorm: out16_pe needs pe");
  LA_CHECK_ARG(window == 0 || (H > 0 && W > 0 && rows % (H * W) == 0), "la_layernorm: bad window geometry");
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16 || dt == LA_F32 || dt == LA_F16X2, "la_layernorm: bad dtype %d", dt);
  la::LnArgs a{x, x2, ldx, rows, E, gamma, beta, eps, gelu, out32, out16, out16_pe, pe, pe_mod, window, H, W, dt == LA_F16X2 ? 1 : 0};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_F16 || dt == LA_F16X2) la::launch_ln<la::f16_t>(a, st);
  else if (dt == LA_BF16) la::launch_ln<la::bf16_t>(a, st);
  else la::launch_ln<float>(a, st);
  LA_CHECK_LAUNCH("la_layernorm");
  return 0;
}
