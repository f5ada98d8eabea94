// Device code shared by the la_gemm translation units (gemm.hip, gemm_w4.hip): row maps, the A-operand maps and the wave-private
// epilogue of the persistent 256 x 256 kernels (a 128 x 64 accumulator block through a 2 KiB slab).
#pragma once
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

// V^T token -> slot (identity, or the 16-wide padded window order)
__device__ __forceinline__ int vt_slot(int t, int ws) { return ws > 0 ? (t / ws) * 16 + (t % ws) : t; }

struct RowMap {
  int mode, p0, p1, p2, p3, p4;
};

// returns destination row or -1 (dropped)
__device__ __forceinline__ int map_row(const RowMap& m, int row) {
  switch (m.mode) {
    case LA_MAP_GROUP:
      return (row / m.p0) * m.p1 + (row % m.p0) + m.p2;
    case LA_MAP_WINDOW_MERGE: {
      const int ws = m.p0, nwy = m.p1, nwx = m.p2, H = m.p3, W = m.p4;
      const int tok = row % (ws * ws);
      int win = row / (ws * ws);
      const int wx = win % nwx;
      win /= nwx;
      const int wy = win % nwy;
      const int b = win / nwy;
      const int y = wy * ws + tok / ws, x = wx * ws + tok % ws;
      return (y < H && x < W) ? (b * H + y) * W + x : -1;
    }
    case LA_MAP_WINDOW_PART: {
      const int ws = m.p0, nwy = m.p1, nwx = m.p2, H = m.p3, W = m.p4;
      const int x = row % W, y = (row / W) % H, b = row / (W * H);
      return ((b * nwy + y / ws) * nwx + x / ws) * ws * ws + (y % ws) * ws + (x % ws);
    }
    case LA_MAP_CONVT2X2: {
      const int W = m.p0, H = m.p1;
      const int x = row % W;
      const int y = (row / W) % H;
      const int b = row / (W * H);
      return (b * 2 * H + 2 * y) * (2 * W) + 2 * x;  // + ky*2W + kx added per column
    }
    default:
      return row;
  }
}

// k offset of the A operand for k-tile start k0: with a_kmod > 0 the A columns repeat with that period while W keeps running
// (W = [W_hi | W_lo] against one A: split-precision weights, LaGemmEpilogue.a_kmod)
// amap LA_MAP_CONV3X3: the k-tile's (plane, tap, channel block) as a wave-uniform element offset from the centre pixel's row - an implicit
// 3 x 3 convolution on a zero-bordered map of plane pairs (p0 = padded row width, p1 = C, p2 = lda)
__device__ __forceinline__ int a_koff(const LaGemmEpilogue& e, int k0) {
  if (e.amap == LA_MAP_CONV3X3) {
    const int C = e.p1, kk = k0 % (18 * C), plane = kk / (9 * C), rem = kk % (9 * C), tap = rem / C;
    return ((tap / 3 - 1) * e.p0 + (tap % 3 - 1)) * e.p2 + plane * C + rem % C;
  }
  return e.a_kmod > 0 ? k0 % e.a_kmod : k0;
}

// source row of GEMM row m (LaGemmEpilogue.amap)
__device__ __forceinline__ int a_row(const LaGemmEpilogue& e, int m) {
  if (e.amap == LA_MAP_NONE || e.amap == LA_MAP_CONV3X3) return m;
  return map_row(RowMap{e.amap, e.p0, e.p1, e.p2, e.p3, e.p4}, m);
}

// rare path of the V^T store (a quad that straddles a window row or the end of M): token by token
template <typename T>
__device__ __noinline__ void vt_store_slow(T* vt, size_t colbase, size_t bstride, int vt_T, int ws, int row, int M, float v0, float v1, float v2,
                                           float v3) {
  const float vv[4] = {v0, v1, v2, v3};
  for (int j = 0; j < 4; ++j) {
    const int rj = row + j;
    if (rj >= M) break;
    const int bj = rj / vt_T, tj = rj % vt_T;
    vt[(size_t)bj * bstride + colbase + vt_slot(tj, ws)] = (T)vv[j];
  }
}

// chunk swizzle of the 16-bit epilogue slab: one store instruction writes rows k (even lanes) and k + 1 (odd lanes) at the SAME
// four chunks; a 128-byte row spans all banks, so the two rows must land in different halves of it (bit 2), the row pairs in
// different chunks of the half (bits 0-1).  (r & 7 put both rows on the same banks: 2-way conflicts, 9 % of the LDS cycles.)
__device__ __forceinline__ int slab_swz(int r) { return ((r & 1) << 2) | ((r >> 1) & 3); }

template <typename T, int EPI>
__device__ __forceinline__ void epilogue_wave(char* slab, unsigned* rtab, f32x16 (&acc)[4][2], int row0, int col0, int n0, int M,
                                              const LaGemmEpilogue& e, int lane, bool nostore = false) {
  // everything below that depends only on the lane (slab addresses, output offsets) is loop-invariant over the tiles of a persistent
  // kernel; hoisted, it would sit in registers through the main loop, which has none to spare (spills there reload through
  // scratch_load + s_waitcnt vmcnt(0)) - an opaque copy of the lane id keeps those few dozen integer operations in the epilogue
  if (EPI != 3) asm volatile("" : "+v"(lane));       // (the fp32 form: its spills then move INTO the main loop - measured on the ISA)
  const int fr = lane & 31, fh = lane >> 5;
  const float bias0 = e.bias ? e.bias[col0 + fr] : 0.f, bias1 = e.bias ? e.bias[col0 + 32 + fr] : 0.f;
  // consume the two loads HERE: hipcc does not see the LDS-DMA pieces, and a load whose first use sits on only some of the paths
  // below leaves "pending" state at the tile loop's back edge - the compiler then drops an s_waitcnt vmcnt(0) into the first k-step
  // of the NEXT tile, which in reality drains the finished tile's whole store burst and the pieces just issued
  asm volatile("" ::"v"(bias0), "v"(bias1));
  const bool vtile = EPI == 1 && e.vt != nullptr && n0 >= e.vt_col0;
  constexpr unsigned NOROW = 0xffffffffu;
  if (EPI == 1 && rtab != nullptr) {
    // output row map (LA_MAP_WINDOW_PART: image-order GEMM rows scattered into window order): the wave tabulates the destination
    // of its 128 rows once - the row itself, or for a V^T tile the position b * (heads * hd * Tpad) + slot of the token
    const RowMap rm{e.map, e.p0, e.p1, e.p2, e.p3, e.p4};
    const unsigned bstride = (unsigned)(e.vt_heads * e.vt_hd * e.vt_Tpad);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int rr = lane + 64 * k, row = row0 + rr;
      const int d = row < M ? map_row(rm, row) : -1;
      unsigned v = NOROW;
      if (d >= 0) v = vtile ? (unsigned)(d / e.vt_T) * bstride + (unsigned)vt_slot(d % e.vt_T, e.vt_ws) : (unsigned)d;
      rtab[rr] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (vtile && rtab != nullptr) {
    T* vt = reinterpret_cast<T*>(e.vt);
    const int cv0 = col0 + fr - e.vt_col0, cv1 = cv0 + 32;
    const size_t cb0 = (size_t)((cv0 / e.vt_hd) * e.vt_hd + cv0 % e.vt_hd) * e.vt_Tpad;
    const size_t cb1 = (size_t)((cv1 / e.vt_hd) * e.vt_hd + cv1 % e.vt_hd) * e.vt_Tpad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int rb = i * 32 + 8 * g4 + 4 * fh;
        const unsigned p0 = rtab[rb], p1 = rtab[rb + 1], p2 = rtab[rb + 2], p3 = rtab[rb + 3];
        // 4 tokens in consecutive slots: one 8-byte store when the run is aligned, 2- / 4-byte pieces when it is not (14-wide windows:
        // every window of an odd column starts 2 slots off the token quads of the 64-wide image row)
        const bool run = p0 != NOROW && p1 == p0 + 1 && p2 == p0 + 2 && p3 == p0 + 3 && (e.vt_Tpad & 1) == 0;
        const bool fast = run && (p0 & 3) == 0 && (e.vt_Tpad & 3) == 0;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const float bias = tj ? bias1 : bias0;
          const float v0 = acc[i][tj][g4 * 4] + bias, v1 = acc[i][tj][g4 * 4 + 1] + bias, v2 = acc[i][tj][g4 * 4 + 2] + bias,
                      v3 = acc[i][tj][g4 * 4 + 3] + bias;
          T* cp = vt + (tj ? cb1 : cb0);
          if (fast) {
            store4v<T>(cp + p0, v0, v1, v2, v3);
          } else if (run) {
            T* pp = cp + p0;
            if (p0 & 1) {
              pp[0] = (T)v0;
              store2<T>(pp + 1, v1, v2);
              pp[3] = (T)v3;
            } else {
              store2<T>(pp, v0, v1);
              store2<T>(pp + 2, v2, v3);
            }
          } else {
            if (p0 != NOROW) cp[p0] = (T)v0;
            if (p1 != NOROW) cp[p1] = (T)v1;
            if (p2 != NOROW) cp[p2] = (T)v2;
            if (p3 != NOROW) cp[p3] = (T)v3;
          }
        }
      }
    return;
  }
  if (vtile) {
    // V^T columns: a lane owns one column, registers 4 g .. 4 g + 3 are 4 consecutive tokens -> one 8-byte store into
    // vt[(b, head, d)][slot] when the quad stays inside a window row.  (b, t) of the quad's first token is carried along
    // incrementally (rows advance by 8 per quad): no division in the unrolled body.
    T* vt = reinterpret_cast<T*>(e.vt);
    const bool quad_ok = (e.vt_T & 3) == 0;
    const int ws = e.vt_ws;
    const int cv0 = col0 + fr - e.vt_col0, cv1 = cv0 + 32;
    const size_t cb0 = (size_t)((cv0 / e.vt_hd) * e.vt_hd + cv0 % e.vt_hd) * e.vt_Tpad;      // (vhead * hd + vd) * Tpad
    const size_t cb1 = (size_t)((cv1 / e.vt_hd) * e.vt_hd + cv1 % e.vt_hd) * e.vt_Tpad;
    const size_t bstride = (size_t)e.vt_heads * e.vt_hd * e.vt_Tpad;
    int row = row0 + 4 * fh;
    int b = row / e.vt_T, t = row % e.vt_T;
    int tq = ws > 0 ? t / ws : 0, tw = ws > 0 ? t % ws : t;                                    // t = tq * ws + tw
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (row < M) {
          const int slot = ws > 0 ? tq * 16 + tw : t;
          // the quad's four tokens sit in consecutive slots of one V^T row: always when T % 4 == 0 (then the slot is 8-byte aligned as
          // well); with T = 901 (HF ViT + CLS) every image but each fourth starts off the 4-token grid - same row, 2- or 4-byte pieces
          const bool inside = row + 3 < M && t + 3 < e.vt_T && (ws == 0 || (ws & 3) == 0 || tw <= ws - 4);
          const bool fast = inside && (quad_ok || (slot & 3) == 0);
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) {
            const float bias = tj ? bias1 : bias0;
            const float v0 = acc[i][tj][g4 * 4] + bias, v1 = acc[i][tj][g4 * 4 + 1] + bias, v2 = acc[i][tj][g4 * 4 + 2] + bias,
                        v3 = acc[i][tj][g4 * 4 + 3] + bias;
            T* rowp = vt + (size_t)b * bstride + (tj ? cb1 : cb0);
            if (fast) {
              store4v<T>(rowp + slot, v0, v1, v2, v3);
            } else if (inside) {
              T* p = rowp + slot;
              if (slot & 1) {
                p[0] = (T)v0;
                store2<T>(p + 1, v1, v2);
                p[3] = (T)v3;
              } else {
                store2<T>(p, v0, v1);
                store2<T>(p + 2, v2, v3);
              }
            } else {
              vt_store_slow<T>(vt, (tj ? cb1 : cb0), bstride, e.vt_T, ws, row, M, v0, v1, v2, v3);
            }
          }
        }
        row += 8;
        t += 8;
        tw += 8;
        if (ws > 0) {
          while (tw >= ws) {
            tw -= ws;
            ++tq;
          }
        }
        if (t >= e.vt_T) {                       // next image / window (vt_T >= 8 on this path: host side)
          t -= e.vt_T;
          ++b;
          tq = ws > 0 ? t / ws : 0;
          tw = ws > 0 ? t % ws : t;
        }
      }
    return;
  }
  if ((EPI == 1 || EPI == 2) && row0 + 128 <= M) {
    // 16-bit output of an interior tile, the common case.  Registers 2k, 2k + 1 of an accumulator are rows 2p, 2p + 1 of ONE column:
    // bias (+ GELU) on the pair in packed fp32, one cvt_pk, one 32-bit LDS store - no lane exchange.  Slab = [8 row pairs][64
    // columns] of such words (256 B = one pass over the 64 LDS banks); 16-byte unit u of row pair p sits at u ^ (p & 1) ^ 8 ((p >> 1) & 1):
    //   a ds_write_b32 stores pairs kp + 4 q + 2 fh - its two half waves land in different 128-byte halves of the bank row;
    //   a ds_read_b128 pass serves 16 lanes = two pairs p, p + 1 reading the units 2 ch - the odd pair's sit on the odd units.
    // Read side: lane (pair p, 8-column group ch) fetches the pair's 8 columns as two 16-byte units
    // and unzips them with v_perm into the two 16-byte row segments it stores: whole 128-byte lines per 8 lanes.
    // The LDS queue of a wave is in order: chunk c + 1 is written right behind the READ INSTRUCTIONS of chunk c, and the read data is
    // waited for (counted) only after those writes have been issued - no round trip is exposed between chunks.
    T* out = reinterpret_cast<T*>(e.out16);
    const int rp = lane >> 3, rch = lane & 7;
    // write bases [tj][kp]: pair 2 fh (+ kp + 4 q added below), logical unit 8 tj + (fr >> 2), physical ^ kp ^ 8 fh
    char* wb[2][2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int kp = 0; kp < 2; ++kp)
        wb[tj][kp] = slab + (2 * fh + kp) * 256 + ((((fr >> 2) + 8 * tj) ^ kp ^ (8 * fh)) << 4) + (fr & 3) * 4;
    const int rx = (rp & 1) ^ (8 * ((rp >> 1) & 1));
    const char* rb0 = slab + rp * 256 + (((2 * rch) ^ rx) << 4);
    const char* rb1 = slab + rp * 256 + (((2 * rch + 1) ^ rx) << 4);
    T* op = out + (size_t)(row0 + 2 * rp) * e.ld16 + col0 + rch * 8;
    const size_t cstride = (size_t)16 * e.ld16;
    const f32x2 b0 = {bias0, bias0}, b1 = {bias1, bias1};
    auto wr = [&](int c) {
      const int i = c >> 1, h = c & 1;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int kp = 0; kp < 2; ++kp) {
            const int r = h * 8 + q * 4 + 2 * kp;
            f32x2 v = f32x2{acc[i][tj][r], acc[i][tj][r + 1]} + (tj ? b1 : b0);
            if (EPI == 2) v = gelu_erf_pk(v);
            *reinterpret_cast<uint32_t*>(wb[tj][kp] + (4 * q) * 256) = pack2<T>(v.x, v.y);
          }
    };
    uint4 u0, u1;
    auto rd = [&]() {
      u0 = *reinterpret_cast<const uint4*>(rb0);
      u1 = *reinterpret_cast<const uint4*>(rb1);
    };
    wr(0);
    asm volatile("" ::: "memory");
    rd();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      asm volatile("" ::: "memory");
      if (c + 1 < 8) wr(c + 1);
      asm volatile("" ::: "memory");
      uint4 ev, od;
      ev.x = __builtin_amdgcn_perm(u0.y, u0.x, 0x05040100u);
      ev.y = __builtin_amdgcn_perm(u0.w, u0.z, 0x05040100u);
      ev.z = __builtin_amdgcn_perm(u1.y, u1.x, 0x05040100u);
      ev.w = __builtin_amdgcn_perm(u1.w, u1.z, 0x05040100u);
      od.x = __builtin_amdgcn_perm(u0.y, u0.x, 0x07060302u);
      od.y = __builtin_amdgcn_perm(u0.w, u0.z, 0x07060302u);
      od.z = __builtin_amdgcn_perm(u1.y, u1.x, 0x07060302u);
      od.w = __builtin_amdgcn_perm(u1.w, u1.z, 0x07060302u);
      if (EPI == 1 && rtab != nullptr) {             // output row map: the two rows go where the wave's table says (pad rows: nowhere)
        const unsigned d0 = rtab[c * 16 + 2 * rp], d1 = rtab[c * 16 + 2 * rp + 1];
        if (d0 != NOROW) *reinterpret_cast<uint4*>(out + (size_t)d0 * e.ld16 + col0 + rch * 8) = ev;
        if (d1 != NOROW) *reinterpret_cast<uint4*>(out + (size_t)d1 * e.ld16 + col0 + rch * 8) = od;
      } else if (!nostore) {
        *reinterpret_cast<uint4*>(op) = ev;
        *reinterpret_cast<uint4*>(op + e.ld16) = od;
      }
      op += cstride;
      if (c + 1 < 8) rd();
    }
    return;
  }
  if (EPI == 1 || EPI == 2) {
    // 16-bit output, general form (edge tiles).
    // Slab = [16 rows][128 B], 16-B chunk c of row r at chunk slot c ^ slab_swz(r).  Lane pairs (fr, fr ^ 1) trade one
    // value per register pair over DPP: the even lane ends up with columns (c, c + 1) of row k, the odd lane with the same columns
    // of row k + 1 - one 32-bit LDS store each.
    T* out = reinterpret_cast<T*>(e.out16);
    const bool odd = fr & 1;
    const int rrow = lane >> 3, rch = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const float bias = tj ? bias1 : bias0;
          const int c = tj * 32 + (fr & ~1);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
              float a = acc[i][tj][h * 8 + q * 4 + 2 * kp] + bias, b = acc[i][tj][h * 8 + q * 4 + 2 * kp + 1] + bias;
              if (EPI == 2) {
                const f32x2 gv = gelu_erf_pk(f32x2{a, b});
                a = gv.x;
                b = gv.y;
              }
              const float y = dpp_mov<0xB1>(odd ? a : b);
              const uint32_t w = odd ? pack2<T>(y, b) : pack2<T>(a, y);
              const int srow = 2 * kp + (odd ? 1 : 0) + 8 * q + 4 * fh;
              *reinterpret_cast<uint32_t*>(slab + srow * 128 + (((c >> 3) ^ slab_swz(srow)) << 4) + (c & 7) * 2) = w;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int sr = rrow + 8 * j;
          const uint4 v = *reinterpret_cast<const uint4*>(slab + sr * 128 + ((rch ^ slab_swz(sr)) << 4));
          const int row = row0 + i * 32 + h * 16 + sr;
          if (EPI == 1 && rtab != nullptr) {
            const unsigned d = rtab[i * 32 + h * 16 + sr];
            if (d != NOROW) *reinterpret_cast<uint4*>(out + (size_t)d * e.ld16 + col0 + rch * 8) = v;
          } else if (row < M && !nostore) {
            *reinterpret_cast<uint4*>(out + (size_t)row * e.ld16 + col0 + rch * 8) = v;
          }
        }
      }
    return;
  }
  // EPI 3: fp32 slab = [8 rows][256 B] (one register quad of both column tiles), 16-B chunk c of row r at slot c ^ (r & 7); the
  // residual (often the output buffer itself) is fetched HALF a wave tile ahead - 16 float4 per lane in the registers the operand
  // fragments occupied - so that its latency is paid twice per tile, not once per slab.
  {
    T* out16 = reinterpret_cast<T*>(e.out16);
    const int rr4 = lane >> 4, rch = lane & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 rv[2][4][2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = min(row0 + (half * 2 + ii) * 32 + 8 * g4 + rr4 + 4 * j, M - 1);
            rv[ii][g4][j] = e.res ? *reinterpret_cast<const float4*>(e.res + (size_t)row * e.ldr + col0 + rch * 4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);            // (no residual: plain fp32 output, e.g. a data gradient)
          }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int i = half * 2 + ii;
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) {
            const float bias = tj ? bias1 : bias0;
            const int c = tj * 32 + fr;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int srow = k + 4 * fh;
              *reinterpret_cast<float*>(slab + srow * 256 + (((c >> 2) ^ (srow & 7)) << 4) + (c & 3) * 4) = acc[i][tj][g4 * 4 + k] + bias;
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int sr = rr4 + 4 * j;
            float4 v = *reinterpret_cast<const float4*>(slab + sr * 256 + ((rch ^ (sr & 7)) << 4));
            const float4 r = rv[ii][g4][j];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            const int row = row0 + i * 32 + 8 * g4 + sr;
            if (row < M) {
              *reinterpret_cast<float4*>(e.out32 + (size_t)row * e.ld32 + col0 + rch * 4) = v;
              if (out16) store4v<T>(out16 + (size_t)row * e.ld16 + col0 + rch * 4, v.x, v.y, v.z, v.w);
            }
          }
        }
    }
  }
}

// gemm_w4.hip: the four-wave persistent 256 x 256 x 64 kernel (EPI 1 / 2 / 3 as epilogue_wave; K % 64 == 0, K >= 128, N % 256 == 0)
template <typename T, int EPI>
void launch_t256w(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, int gm, hipStream_t st);
template <typename T, int EPI>
void launch_t256w_fused(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, int gm, hipStream_t st);

}  // namespace la
