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
  int x2_group;       // > 0: x2 is [rows / x2_group, E] and row r adds x2[r / x2_group] (a per-image vector: the pending token-mean
                      // corrections of single-plane weights, LamEngine mean planes)
  float* cs_part;     // != nullptr: workgroup (group g, chunk c of ceil(cs_rpg / LN_CS_ROWS)) handles a fixed 1 / chunks share of the rows of group g
  int cs_rpg;         // (cs_rpg rows per group) and writes the column sums of what it stored to cs_part[(g * chunks + c) * E ..]: the token means
                      // of the qkv operand come out of the pass that writes it (la_colsum_fold adds the chunks in a fixed order)
};
constexpr int CM_CHUNK = 128;      // rows per partial of la_colmean16 / la_attn_fwd_cs
#ifndef LA_LN_CS_ROWS
#define LA_LN_CS_ROWS 32
#endif
constexpr int LN_CS_ROWS = LA_LN_CS_ROWS;     // rows per partial of the LayerNorm's column sums (la_layernorm_g): 64 images x 901 rows in 128-row shares
                                   // were 512 workgroups on 256 CUs - 2.7 TB/s instead of 4.6

template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d) { store4v<T>(p, a, b, c, d); }

// NVT = float4 vectors per lane (compile time, so the row loads are issued back to back without per-vector predicates);
// EXACT = every lane has all NVT vectors (E == 4 * LPR * NVT), otherwise the tail vectors are predicated.
// CS = the launch also leaves column sums (LnArgs.cs_part): its own instance, so that the twelve accumulator registers and the 4 NVT KiB
// of LDS do not cost the plain launches their eighth wave per SIMD (measured: +22 % on every LayerNorm of the step).
template <typename T, int LPR, int NVT, bool EXACT, bool CS>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int tid = threadIdx.x;
  const int rl = tid / LPR, lane = tid % LPR;
  const int nv = a.E >> 2;
  __shared__ float4 cs_red[CS ? 256 * NVT : 1];
  float4 cs[CS ? NVT : 1];
#pragma unroll
  for (int i = 0; i < (CS ? NVT : 1); ++i) cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int row_begin = blockIdx.x * RPB + rl, row_end = a.rows, row_step = gridDim.x * RPB;
  if (CS) {
    const int chunks = (a.cs_rpg + LN_CS_ROWS - 1) / LN_CS_ROWS;
    const int g = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    // "chunk" ch = the rows RPB ch + rl + it (RPB chunks) of the group, it = 0, 1, ...: the workgroups of a group sweep it TOGETHER, RPB
    // rows each per step, like the plain launch does (2048 workgroups each walking its own contiguous 128 rows measured 46 % slower)
    row_begin = g * a.cs_rpg + ch * RPB + rl;
    row_end = (g + 1) * a.cs_rpg;
    row_step = chunks * RPB;
  }
  for (int row = row_begin; row < row_end; row += row_step) {
    float4 v[NVT];
    // window == -1 / -2: the 16-bit outputs / the input rows live in the interior of zero-bordered [B, H + 2, W + 2] maps (LA_MAP_CONV3X3)
    int prow = row;
    if (a.window < 0) {
      const int x = row % a.W, y = (row / a.W) % a.H, b = row / (a.W * a.H);
      prow = (b * (a.H + 2) + y + 1) * (a.W + 2) + x + 1;
    }
    const float4* xp = reinterpret_cast<const float4*>(a.x + (size_t)(a.window == -2 ? prow : row) * a.ldx);
    const float4* yp = a.x2 ? reinterpret_cast<const float4*>(a.x2 + (a.x2_group > 0 ? (size_t)(row / a.x2_group) * a.E : (size_t)row * a.ldx))
                            : nullptr;
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
    s = LPR == 64 ? wave_sum_dpp(s) : wave_sum(s, LPR);       // (DPP row sums + 4 readlanes: no ds_bpermute round trips in the row's chain)
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
    q = LPR == 64 ? wave_sum_dpp(q) : wave_sum(q, LPR);
    const float rstd = 1.0f / sqrtf(q / (float)a.E + a.eps);

    int drow = a.window == -1 ? prow : row;
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
        // (the fp32 values, not their 16-bit roundings: the mean rounding error of >= 196 tokens is far below what the correction is
        // accurate to.  Accumulators in registers: thread-private LDS slots updated with ds_add_f32 measured 2.3x the kernel time.)
        if (CS) { cs[i].x += o.x; cs[i].y += o.y; cs[i].z += o.z; cs[i].w += o.w; }
        if (o16pe) {
          const float4 p = pp[c];
          if (a.split) store4_split<T>(o16pe + (size_t)drow * 2 * a.E, a.E, c * 4, o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
          else store4<T>(o16pe + (size_t)drow * a.E + c * 4, o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
        }
      }
    }
  }
  if (CS) {                                           // fold the RPB row lanes of a column in index order
#pragma unroll
    for (int i = 0; i < NVT; ++i) cs_red[tid * NVT + i] = cs[i];      // (tid == rl * LPR + lane)
    __syncthreads();
    if (rl == 0) {
      float4* dst = reinterpret_cast<float4*>(a.cs_part + (size_t)blockIdx.x * a.E);
#pragma unroll
      for (int i = 0; i < NVT; ++i) {
        const int c = lane + i * LPR;
        if (EXACT || c < nv) {
          float4 t = cs_red[lane * NVT + i];
#pragma unroll 1
          for (int j = 1; j < RPB; ++j) {         // (rolled: unrolled, its loads set the register count of the whole kernel)
            const float4 u = cs_red[(j * LPR + lane) * NVT + i];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
          }
          dst[c] = t;
        }
      }
    }
  }
}

template <typename T, int LPR, int NVT>
static void launch_ln_nv(const LnArgs& a, int blocks, hipStream_t st) {
  const bool exact = (a.E >> 2) == LPR * NVT;
  if (a.cs_part) {
    if (exact) hipLaunchKernelGGL((layernorm_kernel<T, LPR, NVT, true, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((layernorm_kernel<T, LPR, NVT, false, true>), dim3(blocks), dim3(256), 0, st, a);
  } else {
    if (exact) hipLaunchKernelGGL((layernorm_kernel<T, LPR, NVT, true, false>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((layernorm_kernel<T, LPR, NVT, false, false>), dim3(blocks), dim3(256), 0, st, a);
  }
}

template <typename T>
static void launch_ln(const LnArgs& a, hipStream_t st) {
  const int nv = a.E >> 2;
  int lpr = 64;
  while (lpr > 1 && (lpr >> 1) >= nv) lpr >>= 1;
  const int rpb = 256 / lpr;
  int blocks = (a.rows + rpb - 1) / rpb;
  if (blocks > 8192) blocks = 8192;
  if (a.cs_part) blocks = (a.rows / a.cs_rpg) * ((a.cs_rpg + LN_CS_ROWS - 1) / LN_CS_ROWS);
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

static int ln_impl(const char* name, const float* x, const float* x2, int x2_group, int ldx, int rows, int E, const float* gamma,
                   const float* beta, float eps, int gelu, float* out32, void* out16, void* out16_pe, const float* pe, int pe_mod, int window,
                   int H, int W, int dt, void* stream, float* cs_part = nullptr, int cs_rpg = 0) {
  LA_CHECK_ARG(x && gamma && beta, "%s: null pointer", name);
  LA_CHECK_ARG(rows > 0 && E > 0 && (E % 4) == 0 && E <= 4 * 64 * la::LN_MAXV && (ldx % 4) == 0, "%s: bad shape rows=%d E=%d ldx=%d", name, rows, E,
               ldx);
  LA_CHECK_ARG(out32 || out16 || out16_pe, "%s: no output", name);
  LA_CHECK_ARG(!out16_pe || pe, "%s: out16_pe needs pe", name);
  LA_CHECK_ARG(window == 0 || (window >= -2 && H > 0 && W > 0 && rows % (H * W) == 0), "%s: bad window geometry", name);
  LA_CHECK_ARG(window >= 0 || (!cs_part && !out16_pe && x2_group == 0 && (window == -1 ? (out16 && !out32) : true)),
               "%s: the padded-map forms (window -1 / -2) take no column sums / pe output / per-group vector; -1 writes out16 only", name);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16 || dt == LA_F32 || dt == LA_F16X2, "%s: bad dtype %d", name, dt);
  LA_CHECK_ARG(x2_group >= 0 && (x2_group == 0 || x2), "%s: x2_group needs x2", name);
  LA_CHECK_ARG(!cs_part || (cs_rpg > 0 && rows % cs_rpg == 0 && out16 && dt != LA_F16X2), "%s: column sums need rows %% rows_per_group == 0 and a plain 16-bit output", name);
  la::LnArgs a{x, x2, ldx, rows, E, gamma, beta, eps, gelu, out32, out16, out16_pe, pe, pe_mod, window, H, W, dt == LA_F16X2 ? 1 : 0, x2_group,
               cs_part, cs_rpg};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_F16 || dt == LA_F16X2) la::launch_ln<la::f16_t>(a, st);
  else if (dt == LA_BF16) la::launch_ln<la::bf16_t>(a, st);
  else la::launch_ln<float>(a, st);
  LA_CHECK_LAUNCH(name);
  return 0;
}

extern "C" int la_layernorm(const float* x, const float* x2, int ldx, int rows, int E, const float* gamma, const float* beta,
                            float eps, int gelu, float* out32, void* out16, void* out16_pe, const float* pe, int pe_mod,
                            int window, int H, int W, int dt, void* stream) {
  return ln_impl("la_layernorm", x, x2, 0, ldx, rows, E, gamma, beta, eps, gelu, out32, out16, out16_pe, pe, pe_mod, window, H, W, dt, stream);
}

extern "C" int la_layernorm_g(const float* x, const float* xg, int rows_per_group, int ldx, int rows, int E, const float* gamma,
                              const float* beta, float eps, float* out32, void* out16, int window, int H, int W, float* colsum_part, int dt,
                              void* stream) {
  return ln_impl("la_layernorm_g", x, xg, rows_per_group, ldx, rows, E, gamma, beta, eps, 0, out32, out16, nullptr, nullptr, 0, window, H, W, dt,
                 stream, colsum_part, rows_per_group);
}

// ---- token means per image of a 16-bit operand, and the in-place add of a per-image vector (LamEngine mean planes) -----------------
namespace la {
// part[g][chunk][c] = sum over rows [chunk * CM_CHUNK, +CM_CHUNK) of group g of src[row][c]; colmean16_fold_kernel adds the chunks of a
// group in index order and scales by 1 / rpg.  Chunk size and order are fixed, so a group's mean does not depend on how many groups
// the launch has or on timing: the encoder stays bit-reproducible and episode-shard invariant (an atomic version moved the last bit
// of the means from run to run, which the 16-bit roundings downstream turned into 1-ulp flips: 3e-4 between two runs).
// wpart > 0: group g's rows are the H x W tokens of image g in IMAGE order while src is window-partitioned (ws = wpart,
// LA_MAP_WINDOW_PART order, pad rows skipped).
template <typename T>
__global__ __launch_bounds__(256) void colmean16_kernel(const T* __restrict__ src, int ld, int rpg, int D, float* __restrict__ part, int wpart,
                                                        int H, int W) {
  __shared__ float red[256 * 8];
  const int g = blockIdx.x, c0 = blockIdx.y * CM_CHUNK;
  const int cols8 = D / 8, rl = 256 / cols8;              // row lanes
  const int tid = threadIdx.x, cv = tid % cols8, rlane = tid / cols8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rlane < rl) {
    const int r1 = min(c0 + CM_CHUNK, rpg);
    for (int r = c0 + rlane; r < r1; r += rl) {
      size_t srow;
      if (wpart > 0) {
        const int x = r % W, y = r / W;
        const int nwx = (W + wpart - 1) / wpart, nwy = (H + wpart - 1) / wpart;
        srow = ((size_t)(g * nwy + y / wpart) * nwx + x / wpart) * wpart * wpart + (y % wpart) * wpart + (x % wpart);
      } else {
        srow = (size_t)g * rpg + r;
      }
      const uint4 v = *reinterpret_cast<const uint4*>(src + srow * ld + cv * 8);
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += (float)e[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[tid * 8 + k] = acc[k];
  __syncthreads();
  if (rlane == 0) {
    float* dst = part + ((size_t)g * gridDim.y + blockIdx.y) * D + cv * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
      for (int j = 0; j < rl; ++j) s += red[(j * cols8 + cv) * 8 + k];
      dst[k] = s;
    }
  }
}

// workgroup (column tile of 64, group): chunk lane k adds chunks k, k + 4, ... (independent loads), the four lanes of a column are then
// added in index order - a fixed tree, whatever the launch
__global__ __launch_bounds__(256) void colmean16_fold_kernel(const float* __restrict__ part, int chunks, int D, float inv, float* __restrict__ out,
                                                             int ldo) {
  __shared__ float red[4][64];
  const int g = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), k = threadIdx.x >> 6;
  float s = 0.f;
  if (c < D)
    for (int j = k; j < chunks; j += 4) s += part[((size_t)g * chunks + j) * D + c];
  red[k][threadIdx.x & 63] = s;
  __syncthreads();
  if (k == 0 && c < D) out[(size_t)g * ldo + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * inv;
}

// SPLIT: also leaves the sum as an LA_F16X2 operand row [hi (D) | lo (D)] (the A operand of a three-product fp16 GEMM)
template <bool SPLIT>
__global__ __launch_bounds__(256) void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ v, long rows, int rpg, int D,
                                                         f16_t* __restrict__ split) {
  const int d4 = D / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * d4; i += (long)gridDim.x * 256) {
    const long r = i / d4;
    const int c = (int)(i % d4);
    float4 a = reinterpret_cast<float4*>(x)[i];
    if (v != nullptr) {
      const float4 b = reinterpret_cast<const float4*>(v + (r / rpg) * D)[c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      reinterpret_cast<float4*>(x)[i] = a;
    }
    if (SPLIT) {
      // fp16 planes hold |x| <= 65504 in hi and as much again in lo: the planes SATURATE (|x| up to 131008 stays exact to 2^-11, beyond
      // that the pair clamps) instead of turning into (inf, -inf) -> NaN in the three-product GEMM.  The fp32 stream itself is unchanged.
      constexpr float H = 65504.0f, H2 = 131008.0f;
      const float sx = __builtin_amdgcn_fmed3f(a.x, -H2, H2), sy = __builtin_amdgcn_fmed3f(a.y, -H2, H2);
      const float sz = __builtin_amdgcn_fmed3f(a.z, -H2, H2), sw = __builtin_amdgcn_fmed3f(a.w, -H2, H2);
      const float hx = (float)(f16_t)__builtin_amdgcn_fmed3f(sx, -H, H), hy = (float)(f16_t)__builtin_amdgcn_fmed3f(sy, -H, H);
      const float hz = (float)(f16_t)__builtin_amdgcn_fmed3f(sz, -H, H), hw = (float)(f16_t)__builtin_amdgcn_fmed3f(sw, -H, H);
      f16_t* row = split + (size_t)r * 2 * D;
      store4v<f16_t>(row + c * 4, hx, hy, hz, hw);
      store4v<f16_t>(row + D + c * 4, sx - hx, sy - hy, sz - hz, sw - hw);
    }
  }
}
}  // namespace la

extern "C" int la_colmean16(const void* src, int ld, int groups, int rows_per_group, int D, float* out, float* scratch, int wpart, int H, int W,
                            int dt, void* stream) {
  LA_CHECK_ARG(src && out && scratch && groups > 0 && rows_per_group > 0 && D > 0 && (D % 8) == 0 && D <= 2048 && (ld % 8) == 0 &&
                   (dt == LA_F16 || dt == LA_BF16),
               "la_colmean16: bad arguments (D %% 8, D <= 2048, ld %% 8)");
  LA_CHECK_ARG(wpart == 0 || (H > 0 && W > 0 && H * W == rows_per_group), "la_colmean16: window gather needs H * W == rows_per_group");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int chunks = (rows_per_group + la::CM_CHUNK - 1) / la::CM_CHUNK;
  const dim3 grid(groups, chunks), blk(256);
  if (dt == LA_F16) hipLaunchKernelGGL(la::colmean16_kernel<la::f16_t>, grid, blk, 0, st, (const la::f16_t*)src, ld, rows_per_group, D, scratch, wpart, H, W);
  else hipLaunchKernelGGL(la::colmean16_kernel<la::bf16_t>, grid, blk, 0, st, (const la::bf16_t*)src, ld, rows_per_group, D, scratch, wpart, H, W);
  hipLaunchKernelGGL(la::colmean16_fold_kernel, dim3((D + 63) / 64, groups), blk, 0, st, scratch, chunks, D, 1.0f / (float)rows_per_group, out, D);
  LA_CHECK_LAUNCH("la_colmean16");
  return 0;
}

extern "C" int la_colsum_fold(const float* part, int groups, int chunks, int D, float inv, float* out, int ldo, void* stream) {
  LA_CHECK_ARG(part && out && groups > 0 && chunks > 0 && D > 0 && ldo >= D, "la_colsum_fold: bad arguments");
  hipLaunchKernelGGL(la::colmean16_fold_kernel, dim3((D + 63) / 64, groups), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), part, chunks, D, inv, out,
                     ldo);
  LA_CHECK_LAUNCH("la_colsum_fold");
  return 0;
}

extern "C" int la_add_rowvec(float* x, const float* v, long rows, int rows_per_group, int D, void* stream) {
  LA_CHECK_ARG(x && v && rows > 0 && rows_per_group > 0 && D > 0 && (D % 4) == 0, "la_add_rowvec: bad arguments");
  const long n = rows * (D / 4);
  hipLaunchKernelGGL(la::add_rowvec_kernel<false>, dim3((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, v, rows, rows_per_group, D, nullptr);
  LA_CHECK_LAUNCH("la_add_rowvec");
  return 0;
}

extern "C" int la_add_rowvec_split(float* x, const float* v, long rows, int rows_per_group, int D, void* split16, void* stream) {
  LA_CHECK_ARG(x && split16 && rows > 0 && (v == nullptr || rows_per_group > 0) && D > 0 && (D % 4) == 0, "la_add_rowvec_split: bad arguments");
  const long n = rows * (D / 4);
  hipLaunchKernelGGL(la::add_rowvec_kernel<true>, dim3((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, v, rows, rows_per_group > 0 ? rows_per_group : 1, D,
                     reinterpret_cast<la::f16_t*>(split16));
  LA_CHECK_LAUNCH("la_add_rowvec_split");
  return 0;
}

// ---- LayerNorm folded into its neighbour GEMMs (LaGemmEpilogue.nstat_out / nstat_in) ---------------------------------------------------
namespace la {
constexpr int NF_ROWS = 128;      // rows per workgroup of la_norm_finalize = rows per column-sum partial (as CM_CHUNK)

// workgroup (group g, chunk c): rows [c * 128, +128) of group g.  Phase A: thread t < 128 folds the partial sums of its row (slots in index
// order) into (mean, rstd).  Phase B (cs_part): column sums of rstd (x16 - mean) over the chunk, laid out like colmean16_kernel's partials.
template <typename T>
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ part, int M, int nslots, int E, float eps, float* __restrict__ mr,
                                                            const T* __restrict__ x16, int ld16, int rpg, float* __restrict__ cs_part) {
  __shared__ float2 smr[NF_ROWS];
  __shared__ float red[256 * 8];
  const int g = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const int r0 = c * NF_ROWS, r1 = min(r0 + NF_ROWS, rpg);
  const size_t base = (size_t)g * rpg;
  if (tid < NF_ROWS && r0 + tid < r1) {
    const size_t row = base + r0 + tid;
    float2 v;
    if (part != nullptr) {
      // (slot pairs as float4: nslots = E / 64 is even for every E the producer epilogue takes - N % 256 == 0; slots still added in index order)
      const float4* pp = reinterpret_cast<const float4*>(part) + row * (nslots >> 1);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int j = 0; j < (nslots >> 1); ++j) {
        const float4 t = pp[j];
        s1 = (s1 + t.x) + t.z;
        s2 = (s2 + t.y) + t.w;
      }
      const float mean = s1 / (float)E;
      const float var = fmaxf(s2 / (float)E - mean * mean, 0.f);
      v = make_float2(mean, 1.0f / sqrtf(var + eps));
      reinterpret_cast<float2*>(mr)[row] = v;
    } else {
      v = reinterpret_cast<const float2*>(mr)[row];
    }
    smr[tid] = v;
  }
  // the padding rows of mr (a consumer GEMM's last row tile reads them unpredicated)
  if (part != nullptr && g == (int)gridDim.x - 1 && c == (int)gridDim.y - 1)
    for (int r = M + tid; r < ((M + 255) / 256) * 256; r += 256) reinterpret_cast<float2*>(mr)[r] = make_float2(0.f, 0.f);
  if (cs_part == nullptr) return;
  __syncthreads();
  const int cols8 = E / 8, rl = 256 / cols8;
  const int cv = tid % cols8, rlane = tid / cols8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rlane < rl) {
    for (int r = r0 + rlane; r < r1; r += rl) {
      const float2 m = smr[r - r0];
      const uint4 v = *reinterpret_cast<const uint4*>(x16 + (base + r) * ld16 + cv * 8);
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += ((float)e[k] - m.x) * m.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[tid * 8 + k] = acc[k];
  __syncthreads();
  if (rlane == 0) {
    float* dst = cs_part + ((size_t)g * gridDim.y + c) * E + cv * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s_ = 0.f;
      for (int j = 0; j < rl; ++j) s_ += red[(j * cols8 + cv) * 8 + k];
      dst[k] = s_;
    }
  }
}

// one wave per row: x16 = rn16(x), mr = (mean, rstd) with the two-pass variance of la_layernorm (the row lives in registers)
template <typename T>
__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, int ldx, int M, int E, float eps, T* __restrict__ x16,
                                                         float* __restrict__ mr) {
  const int lane = threadIdx.x & 63, nv = E >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const float4* xp = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
    float4 v[LN_MAXV];
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      v[i] = c < nv ? xp[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s_ += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum_dpp(s_) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        store4v<T>(x16 + (size_t)row * E + c * 4, v[i].x, v[i].y, v[i].z, v[i].w);
      }
    }
    q = wave_sum_dpp(q);
    if (lane == 0) reinterpret_cast<float2*>(mr)[row] = make_float2(mean, 1.0f / sqrtf(q / (float)E + eps));
  }
}
}  // namespace la

extern "C" int la_norm_finalize(const float* part, int M, int nslots, int E, float eps, float* mr, const void* x16, int ld16, int rows_per_group,
                                float* cs_part, int dt, void* stream) {
  LA_CHECK_ARG(mr && M > 0 && E > 0 && (E % 8) == 0 && E <= 2048 && (part == nullptr || (nslots > 0 && (nslots % 2) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0)),
               "la_norm_finalize: bad arguments (E %% 8, E <= 2048, an even number of 16-byte aligned slots)");
  LA_CHECK_ARG(part != nullptr || cs_part != nullptr, "la_norm_finalize: nothing to do (no partial sums and no column sums)");
  LA_CHECK_ARG(cs_part == nullptr || (x16 && rows_per_group > 0 && (M % rows_per_group) == 0 && (ld16 % 8) == 0 && (dt == LA_F16 || dt == LA_BF16)),
               "la_norm_finalize: column sums need x16 (16-bit, ld %% 8) and M %% rows_per_group == 0");
  const int rpg = cs_part ? rows_per_group : M;
  const dim3 grid(M / rpg, (rpg + la::NF_ROWS - 1) / la::NF_ROWS);
  LA_CHECK_ARG(grid.y <= 65535, "la_norm_finalize: more than 65535 chunks of 128 rows per group");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_BF16) hipLaunchKernelGGL(la::norm_finalize_kernel<la::bf16_t>, grid, dim3(256), 0, st, part, M, nslots, E, eps, mr, (const la::bf16_t*)x16, ld16, rpg, cs_part);
  else hipLaunchKernelGGL(la::norm_finalize_kernel<la::f16_t>, grid, dim3(256), 0, st, part, M, nslots, E, eps, mr, (const la::f16_t*)x16, ld16, rpg, cs_part);
  LA_CHECK_LAUNCH("la_norm_finalize");
  return 0;
}

extern "C" int la_norm_stats(const float* x, int ldx, int M, int E, float eps, void* x16, float* mr, int dt, void* stream) {
  LA_CHECK_ARG(x && x16 && mr && M > 0 && E > 0 && (E % 4) == 0 && E <= 4 * 64 * la::LN_MAXV && (ldx % 4) == 0 && (dt == LA_F16 || dt == LA_BF16),
               "la_norm_stats: bad arguments (E %% 4, E <= 2048, 16-bit output)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int blocks = (M + 3) / 4 > 16384 ? 16384 : (M + 3) / 4;
  if (dt == LA_BF16) hipLaunchKernelGGL(la::norm_stats_kernel<la::bf16_t>, dim3(blocks), dim3(256), 0, st, x, ldx, M, E, eps, (la::bf16_t*)x16, mr);
  else hipLaunchKernelGGL(la::norm_stats_kernel<la::f16_t>, dim3(blocks), dim3(256), 0, st, x, ldx, M, E, eps, (la::f16_t*)x16, mr);
  LA_CHECK_LAUNCH("la_norm_stats");
  return 0;
}
