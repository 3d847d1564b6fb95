// Fused multi-head attention for the ViT bottleneck (VisionTransformer.py:38-50, TBI_TransUNet.py:44-62), gfx950, head size 128:
//
//   O = softmax(scale * Q K^T) V            scale = 1/sqrt(num_heads) in the reference (:42), any value here
//
// forward + backward without the [B, heads, N, N] score tensors in HBM (the unfused path writes and re-reads them 3x per direction).
// Q, K, V are channel slices of the fused projection output [B, N, 3*hidden] (head h = columns h*128 .. h*128+127 of each slice),
// O / dO of [B, N, hidden]: no head transposes exist anywhere.
//
// All three kernels compute the TRANSPOSED score tile with the MFMA so that the accumulator layout (lane = one column, 4
// consecutive rows) IS the B-operand layout of the second GEMM - probabilities never leave registers:
//   forward   S^T = K Q^T  (rows keys, cols queries)  ->  P^T  ->  O^T  += V^T P^T        (V^T fragments: transposed LDS reads)
//   dQ        S^T, dP^T = V dO^T  ->  dS^T  ->  dQ^T += K^T dS^T
//   dK, dV    S = Q K^T (rows queries, cols keys), dP = dO V^T  ->  P, dS  ->  dV^T += dO^T P,  dK^T += Q^T dS
// The MFMA row i of a 16-row score tile t is mapped to tile row 32*(t>>1) + 8*(i>>2) + 4*(t&1) + (i&3): two tiles then fill the
// 32 contraction slots of the second GEMM in natural order (slot s = row 32c + s), and the transposed reads of a 32-lane half hit
// blocks 8 rows apart (conflict-free on the image below).
// LDS image of a [64][128] bf16 tile (256-byte rows): 16-byte chunk ch of row r at 256*r + 16*(ch ^ (((r&3)<<2) | ((r>>2)&3))) -
// one image serves the row reads (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16).
// Softmax statistics: running maximum m and PER-LANE partial sums (the four 16-lane groups hold disjoint key
// subsets of a query column; the rescale factor is per query, so the partial sums are only added across the groups once, at the end).
// The backward is two kernels (dQ by query tile, dK/dV by key tile) that both recompute the probabilities from the saved
// log-sum-exp: no atomics, bitwise reproducible.
#include "common.h"

struct FlashParams {
  const bf16_t *q, *k, *v;   // head 0 of image 0; row stride ld, image stride N*ld, head stride 128
  const bf16_t *o, *d_o;     // [B][N][ldo] (backward)
  bf16_t *out;               // forward: O; dQ kernel: dQ (row stride ldq_out); dKV kernel: unused
  bf16_t *dk, *dv;
  float *lse, *delta;        // [B*H][N], log2 units: lse2 = m2 + log2(l)
  float *o32;                // optional fp32 copy of O, dense [B][N][H*128]: delta = sum dO*O without O's bf16 rounding
  int32_t B, N, H, ld, ldo, ld_out;
  float scale, scale_log2;
};

#define FA_NEG (-1.0e30f)

__device__ __forceinline__ int fa_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// 64 rows x 128 columns -> registers (1024 / NTHR x 16 B per thread), rows >= N read as zero
template <int NTHR>
__device__ __forceinline__ void fa_load_tile(uint4 (&r)[1024 / NTHR], const bf16_t* base, int64_t ld, int row0, int N, int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / NTHR; ++it) {
    int idx = tid + NTHR * it, row = idx >> 4, ch = idx & 15;
    int gr = row0 + row;
    r[it] = make_uint4(0, 0, 0, 0);
    if (gr < N) r[it] = *reinterpret_cast<const uint4*>(base + (int64_t)gr * ld + ch * 8);
  }
}
template <int NTHR>
__device__ __forceinline__ void fa_store_tile(char* lds, const uint4 (&r)[1024 / NTHR], int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / NTHR; ++it) {
    int idx = tid + NTHR * it, row = idx >> 4, ch = idx & 15;
    *reinterpret_cast<uint4*>(lds + fa_off(row, ch)) = r[it];
  }
}
// row read: operand element [row][k = 32*ks + 8g .. +7]
__device__ __forceinline__ bf16x8_t fa_row(const char* lds, int row, int ks, int g) {
  return *reinterpret_cast<const bf16x8_t*>(lds + fa_off(row, ks * 4 + g));
}
// transposed read: operand element [free = 16*dt + li][slot 8g + j] = tile[32c + 8g + j][16*dt + li]
__device__ __forceinline__ bf16x8_t fa_tr(const char* lds, int c, int dt, int g, int tq, int tp) {
  const int r0 = 32 * c + 8 * g + tq;
  const char* a0 = lds + fa_off(r0, 2 * dt + (tp >> 1)) + 8 * (tp & 1);
  const char* a1 = lds + fa_off(r0 + 4, 2 * dt + (tp >> 1)) + 8 * (tp & 1);
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a1));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t fa_pack(const f32x4_t& a, const f32x4_t& b) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  u32x4_t v = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t fa_gload(const bf16_t* base, int64_t ld, int row, int N, int col) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < N) v = *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + col);
  return __builtin_bit_cast(bf16x8_t, v);
}
// MFMA row i of score tile t -> row of the 64-row LDS tile
__device__ __forceinline__ int fa_tile_row(int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }

// ---------------------------------------------------------------------------------------------------------------- forward
// grid (ceil(N/128), H, B), 256 threads: wave w owns queries q0 + 32w .. +31 (two 16-query column tiles).
// Online softmax over the key tiles.  The probabilities enter the second GEMM as TWO bf16 terms, e = hi + lo with hi = bf16(e) and
// lo = bf16(e - hi) (relative error 2^-17 instead of 2^-9): O is then the fp32-accurate sum_k P V.  That is what makes the
// recomputing backward exact - delta = sum_d dO*O (from the fp32 copy of O) equals sum_k P dP, so the rows of dS sum to zero as
// they do in exact arithmetic; with single-bf16 probabilities delta carries a per-row error of relative size 2^-9 that is
// correlated over the keys and shows up 1.5x in dQ / dK (measured against fp64).  Cost: 32 extra MFMAs per 64 in the key loop.
template <int QT, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void flash_fwd_kernel(const FlashParams p) {
  constexpr int NTHR = 64 * NW;             // QT = 16-query column tiles per wave, NW waves: 16*QT*NW queries per workgroup
  __shared__ __attribute__((aligned(16))) char lds_k[64 * 256];
  __shared__ __attribute__((aligned(16))) char lds_v[64 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int b = blockIdx.z, h = blockIdx.y, N = p.N;
  const int64_t ioff = (int64_t)b * N * p.ld + h * 128;
  const bf16_t *qb = p.q + ioff, *kb = p.k + ioff, *vb = p.v + ioff;
  const int q0 = blockIdx.x * (16 * QT * NW) + wv * 16 * QT;

  bf16x8_t qf[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = fa_gload(qb, p.ld, q0 + qt * 16 + li, N, ks * 32 + 8 * g);

  f32x4_t o[8][QT];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) o[dt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m[QT], l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m[qt] = FA_NEG; l[qt] = 0.f; }

  const int ntiles = (N + 63) >> 6;
  uint4 rk[1024 / NTHR], rv[1024 / NTHR];
  fa_load_tile<NTHR>(rk, kb, p.ld, 0, N, tid);
  fa_load_tile<NTHR>(rv, vb, p.ld, 0, N, tid);
  fa_store_tile<NTHR>(lds_k, rk, tid);
  fa_store_tile<NTHR>(lds_v, rv, tid);
  __syncthreads();
  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles;
    if (more) {
      fa_load_tile<NTHR>(rk, kb, p.ld, (j + 1) * 64, N, tid);
      fa_load_tile<NTHR>(rv, vb, p.ld, (j + 1) * 64, N, tid);
    }
    f32x4_t st[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) st[kt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const int krow = fa_tile_row(kt, li);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t kf = fa_row(lds_k, krow, ks, g);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], st[kt][qt], 0, 0, 0);
      }
    }
    // lane holds keys j*64 + 32*(kt>>1) + 8g + 4*(kt&1) + r of query column qt*16 + li
    const bool tail = (j + 1) * 64 > N;
    bf16x8_t ph[2][QT], pl[2][QT];
    if (tail) {           // keys past N (last tile only): a uniform branch, not a select per element in every tile
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j * 64 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r >= N) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) st[kt][qt][r] = FA_NEG;
          }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = FA_NEG;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = st[kt][qt][r] * p.scale_log2;
          st[kt][qt][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mn = fmaxf(m[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);
      m[qt] = mn;
      float ls = 0.f;
      f32x4_t lo[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = __builtin_amdgcn_exp2f(st[kt][qt][r] - mn);
          ls += e;
          st[kt][qt][r] = e;
          lo[kt][r] = e - bf2f(f2bf(e));
        }
      l[qt] = l[qt] * alpha + ls;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) o[dt][qt] *= alpha;
      ph[0][qt] = fa_pack(st[0][qt], st[1][qt]);
      ph[1][qt] = fa_pack(st[2][qt], st[3][qt]);
      pl[0][qt] = fa_pack(lo[0], lo[1]);
      pl[1][qt] = fa_pack(lo[2], lo[3]);
    }
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8_t vf = fa_tr(lds_v, c, dt, g, tq, tp);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ph[c][qt], o[dt][qt], 0, 0, 0);
          o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl[c][qt], o[dt][qt], 0, 0, 0);
        }
      }
    __syncthreads();
    if (more) {
      fa_store_tile<NTHR>(lds_k, rk, tid);
      fa_store_tile<NTHR>(lds_v, rv, tid);
    }
    __syncthreads();
  }
  // lane holds O[d = 16*dt + 4g + r][q = qt*16 + li]
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16);
    lt += __shfl_xor(lt, 32);
    const float inv = 1.0f / lt;
    const int q = q0 + qt * 16 + li;
    if (q < N) {
      bf16_t* orow = p.out + ((int64_t)b * N + q) * p.ld_out + h * 128;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        uint2 w;
        w.x = pack2bf(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
        w.y = pack2bf(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) = w;
      }
      if (p.o32) {
        float* frow = p.o32 + ((int64_t)b * N + q) * (p.H * 128) + h * 128;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(frow + dt * 16 + 4 * g) = o[dt][qt] * inv;
      }
      if (g == 0) p.lse[((int64_t)b * p.H + h) * N + q] = m[qt] + __log2f(lt);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ backward: dQ
// Same tiling as the forward.  Also writes delta[q] = sum_d dO[q][d] * O[q][d] for the dK/dV kernel (which runs after it).
template <int QT, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void flash_bwd_dq_kernel(const FlashParams p) {
  constexpr int NTHR = 64 * NW;
  __shared__ __attribute__((aligned(16))) char lds_k[64 * 256];
  __shared__ __attribute__((aligned(16))) char lds_v[64 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int b = blockIdx.z, h = blockIdx.y, N = p.N;
  const int64_t ioff = (int64_t)b * N * p.ld + h * 128, ooff = (int64_t)b * N * p.ldo + h * 128;
  const bf16_t *qb = p.q + ioff, *kb = p.k + ioff, *vb = p.v + ioff, *ob = p.o + ooff, *dob = p.d_o + ooff;
  const int q0 = blockIdx.x * (16 * QT * NW) + wv * 16 * QT;
  const int64_t srow = ((int64_t)b * p.H + h) * N;

  bf16x8_t qf[QT][4], dof[QT][4];
  float lse[QT], delta[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + li;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[qt][ks] = fa_gload(qb, p.ld, q, N, ks * 32 + 8 * g);
      dof[qt][ks] = fa_gload(dob, p.ldo, q, N, ks * 32 + 8 * g);
      if (p.o32) {
        if (q < N) {
          const float* frow = p.o32 + ((int64_t)b * N + q) * (p.H * 128) + h * 128 + ks * 32 + 8 * g;
          f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(frow), f1 = *reinterpret_cast<const f32x4_t*>(frow + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) dsum += (float)dof[qt][ks][e] * f0[e] + (float)dof[qt][ks][4 + e] * f1[e];
        }
      } else {
        bf16x8_t of = fa_gload(ob, p.ldo, q, N, ks * 32 + 8 * g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)dof[qt][ks][e] * (float)of[e];
      }
    }
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 32);
    delta[qt] = dsum;
    lse[qt] = q < N ? p.lse[srow + q] : 0.f;
    if (g == 0 && q < N) p.delta[srow + q] = dsum;
  }

  f32x4_t acc[8][QT];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) acc[dt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (N + 63) >> 6;
  uint4 rk[1024 / NTHR], rv[1024 / NTHR];
  fa_load_tile<NTHR>(rk, kb, p.ld, 0, N, tid);
  fa_load_tile<NTHR>(rv, vb, p.ld, 0, N, tid);
  fa_store_tile<NTHR>(lds_k, rk, tid);
  fa_store_tile<NTHR>(lds_v, rv, tid);
  __syncthreads();
  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles;
    if (more) {
      fa_load_tile<NTHR>(rk, kb, p.ld, (j + 1) * 64, N, tid);
      fa_load_tile<NTHR>(rv, vb, p.ld, (j + 1) * 64, N, tid);
    }
    const bool tail = (j + 1) * 64 > N;
    bf16x8_t dsb[2][QT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4_t st[2][QT], dp[2][QT];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * c + t;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st[t][qt] = dp[t][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int krow = fa_tile_row(kt, li);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          bf16x8_t kf = fa_row(lds_k, krow, ks, g);
          bf16x8_t vf = fa_row(lds_v, krow, ks, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            st[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], st[t][qt], 0, 0, 0);
            dp[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qt][ks], dp[t][qt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(st[t][qt][r] * p.scale_log2 - lse[qt]);
            if (tail && j * 64 + 32 * c + 8 * g + 4 * t + r >= N) pr = 0.f;
            st[t][qt][r] = pr * (dp[t][qt][r] - delta[qt]) * p.scale;
          }
        dsb[c][qt] = fa_pack(st[0][qt], st[1][qt]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8_t kf = fa_tr(lds_k, c, dt, g, tq, tp);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsb[c][qt], acc[dt][qt], 0, 0, 0);
      }
    __syncthreads();
    if (more) {
      fa_store_tile<NTHR>(lds_k, rk, tid);
      fa_store_tile<NTHR>(lds_v, rv, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + li;
    if (q < N) {
      bf16_t* orow = p.out + ((int64_t)b * N + q) * p.ld_out + h * 128;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        uint2 w;
        w.x = pack2bf(acc[dt][qt][0], acc[dt][qt][1]);
        w.y = pack2bf(acc[dt][qt][2], acc[dt][qt][3]);
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) = w;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------------- backward: dK, dV
// grid (ceil(N/128), H, B): wave w owns keys k0 + 32w .. +31 (two 16-key column tiles) and walks all query tiles of 64.
template <int KT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void flash_bwd_dkv_kernel(const FlashParams p) {
  constexpr int NTHR = 64 * NW;
  __shared__ __attribute__((aligned(16))) char lds_q[64 * 256];
  __shared__ __attribute__((aligned(16))) char lds_do[64 * 256];
  __shared__ __attribute__((aligned(16))) float lds_lse[64], lds_delta[64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int b = blockIdx.z, h = blockIdx.y, N = p.N;
  const int64_t ioff = (int64_t)b * N * p.ld + h * 128, ooff = (int64_t)b * N * p.ldo + h * 128;
  const bf16_t *qb = p.q + ioff, *kb = p.k + ioff, *vb = p.v + ioff, *dob = p.d_o + ooff;
  const int k0 = blockIdx.x * (16 * KT * NW) + wv * 16 * KT;
  const int64_t srow = ((int64_t)b * p.H + h) * N;

  bf16x8_t kf[KT][4], vf[KT][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[kt][ks] = fa_gload(kb, p.ld, k0 + kt * 16 + li, N, ks * 32 + 8 * g);
      vf[kt][ks] = fa_gload(vb, p.ld, k0 + kt * 16 + li, N, ks * 32 + 8 * g);
    }
  f32x4_t adv[8][KT], adk[8][KT];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) adv[dt][kt] = adk[dt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (N + 63) >> 6;
  uint4 rq[1024 / NTHR], rd[1024 / NTHR];
  float rl = 0.f, rdl = 0.f;
  auto load_stats = [&](int row0) {
    if (tid < 64) {
      int q = row0 + tid;
      rl = q < N ? p.lse[srow + q] : 0.f;
      rdl = q < N ? p.delta[srow + q] : 0.f;
    }
  };
  fa_load_tile<NTHR>(rq, qb, p.ld, 0, N, tid);
  fa_load_tile<NTHR>(rd, dob, p.ldo, 0, N, tid);
  load_stats(0);
  fa_store_tile<NTHR>(lds_q, rq, tid);
  fa_store_tile<NTHR>(lds_do, rd, tid);
  if (tid < 64) { lds_lse[tid] = rl; lds_delta[tid] = rdl; }
  __syncthreads();
  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles;
    if (more) {
      fa_load_tile<NTHR>(rq, qb, p.ld, (j + 1) * 64, N, tid);
      fa_load_tile<NTHR>(rd, dob, p.ldo, (j + 1) * 64, N, tid);
      load_stats((j + 1) * 64);
    }
    const bool tail = (j + 1) * 64 > N;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      // two 16-query row tiles (t = 0, 1) of chunk c: lane holds queries 32c + 8g + 4t + r, key column kt*16 + li
      f32x4_t st[2][KT], dp[2][KT];
      bf16x8_t pbf[KT], dsb[KT];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) st[t][kt] = dp[t][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int qrow = fa_tile_row(2 * c + t, li);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          bf16x8_t qa = fa_row(lds_q, qrow, ks, g);
          bf16x8_t da = fa_row(lds_do, qrow, ks, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            st[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], st[t][kt], 0, 0, 0);
            dp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[t][kt], 0, 0, 0);
          }
        }
      }
      f32x4_t lq[2], dq[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        lq[t] = *reinterpret_cast<const f32x4_t*>(&lds_lse[32 * c + 8 * g + 4 * t]);
        dq[t] = *reinterpret_cast<const f32x4_t*>(&lds_delta[32 * c + 8 * g + 4 * t]);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4_t pt[2], dt_[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(st[t][kt][r] * p.scale_log2 - lq[t][r]);
            if (tail && j * 64 + 32 * c + 8 * g + 4 * t + r >= N) pr = 0.f;
            pt[t][r] = pr;
            dt_[t][r] = pr * (dp[t][kt][r] - dq[t][r]) * p.scale;
          }
        pbf[kt] = fa_pack(pt[0], pt[1]);
        dsb[kt] = fa_pack(dt_[0], dt_[1]);
      }
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        bf16x8_t dot = fa_tr(lds_do, c, dt, g, tq, tp);
        bf16x8_t qt_ = fa_tr(lds_q, c, dt, g, tq, tp);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          adv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pbf[kt], adv[dt][kt], 0, 0, 0);
          adk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_, dsb[kt], adk[dt][kt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      fa_store_tile<NTHR>(lds_q, rq, tid);
      fa_store_tile<NTHR>(lds_do, rd, tid);
      if (tid < 64) { lds_lse[tid] = rl; lds_delta[tid] = rdl; }
    }
    __syncthreads();
  }
  // lane holds dV[key = k0 + kt*16 + li][d = 16*dt + 4g + r]
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = k0 + kt * 16 + li;
    if (key < N) {
      bf16_t* krow = p.dk + ((int64_t)b * N + key) * p.ld_out + h * 128;
      bf16_t* vrow = p.dv + ((int64_t)b * N + key) * p.ld_out + h * 128;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        uint2 w;
        w.x = pack2bf(adk[dt][kt][0], adk[dt][kt][1]);
        w.y = pack2bf(adk[dt][kt][2], adk[dt][kt][3]);
        *reinterpret_cast<uint2*>(krow + dt * 16 + 4 * g) = w;
        w.x = pack2bf(adv[dt][kt][0], adv[dt][kt][1]);
        w.y = pack2bf(adv[dt][kt][2], adv[dt][kt][3]);
        *reinterpret_cast<uint2*>(vrow + dt * 16 + 4 * g) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ host
// Tiling (USSEG_FLASH_TILE, default automatic): 1 = 8 waves x 16 queries (keys), 128 per workgroup - two waves per SIMD, the VALU
// softmax of one overlapping the MFMAs of the other (the kernels are VALU- and LDS-bound at head size 128): 44 / 118 us forward /
// backward at 8 x 1024 tokens x 4 heads against 70 / 144 us for 2 = 4 waves x 32; 0 = 4 waves x 16, 64 per workgroup, when 128-query
// workgroups would not fill the 256 CUs (16 images x 256 tokens: 13.8 / 30.5 us against 16.5 / 36.5).
static int fa_variant(const UssegFlashDesc* d) {
  static const int v = getenv("USSEG_FLASH_TILE") ? atoi(getenv("USSEG_FLASH_TILE")) : -1;
  if (v >= 0) return v;
  return (int64_t)d->B * d->H * ((d->N + 127) / 128) >= 256 ? 1 : 0;
}
static int fa_check(const UssegFlashDesc* d) {
  USSEG_CHECK_ARG(d != nullptr, "null descriptor");
  USSEG_CHECK_ARG(d->head_dim == 128, "fused attention is built for head size 128 (got %d)", d->head_dim);
  USSEG_CHECK_ARG(d->B > 0 && d->N > 0 && d->H > 0, "empty problem");
  USSEG_CHECK_ARG(d->B <= 65535 && d->H <= 65535, "grid limit");
  USSEG_CHECK_ARG(d->ld_qkv % 8 == 0 && d->ld_o % 8 == 0 && d->ld_qkv >= d->H * 128 && d->ld_o >= d->H * 128, "row strides must be multiples of 8 elements");
  return USSEG_OK;
}
static FlashParams fa_params(const UssegFlashDesc* d, const void* q, const void* k, const void* v) {
  FlashParams p{};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v;
  p.B = d->B; p.N = d->N; p.H = d->H; p.ld = d->ld_qkv; p.ldo = d->ld_o;
  p.scale = d->scale; p.scale_log2 = d->scale * 1.4426950408889634f;
  return p;
}

extern "C" int usseg_flash_attn_fwd(const UssegFlashDesc* d, const void* q, const void* k, const void* v, void* o, float* o32, float* lse,
                                    usseg_stream_t stream) {
  if (int e = fa_check(d)) return e;
  USSEG_CHECK_ARG(q && k && v && o && lse, "null operand");
  FlashParams p = fa_params(d, q, k, v);
  p.out = (bf16_t*)o; p.ld_out = d->ld_o; p.lse = lse; p.o32 = o32;
  const int tv = fa_variant(d);
#define FA_LAUNCH(K_, QT_, NW_) hipLaunchKernelGGL((K_<QT_, NW_>), dim3((d->N + 16 * QT_ * NW_ - 1) / (16 * QT_ * NW_), d->H, d->B), dim3(64 * NW_), 0, (hipStream_t)stream, p)
  if (tv == 2) FA_LAUNCH(flash_fwd_kernel, 2, 4); else if (tv == 1) FA_LAUNCH(flash_fwd_kernel, 1, 8); else FA_LAUNCH(flash_fwd_kernel, 1, 4);
  return usseg_check_launch("flash_attn_fwd");
}

extern "C" int usseg_flash_attn_bwd(const UssegFlashDesc* d, const void* q, const void* k, const void* v, const void* o, const float* o32,
                                    const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv, usseg_stream_t stream) {
  if (int e = fa_check(d)) return e;
  USSEG_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "null operand");
  FlashParams p = fa_params(d, q, k, v);
  p.o = (const bf16_t*)o; p.d_o = (const bf16_t*)d_o; p.lse = const_cast<float*>(lse); p.delta = delta; p.o32 = const_cast<float*>(o32);
  p.out = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.ld_out = d->ld_qkv;
  const int tv = fa_variant(d);
  if (tv == 2) FA_LAUNCH(flash_bwd_dq_kernel, 2, 4); else if (tv == 1) FA_LAUNCH(flash_bwd_dq_kernel, 1, 8); else FA_LAUNCH(flash_bwd_dq_kernel, 1, 4);
  if (int e = usseg_check_launch("flash_attn_bwd_dq")) return e;
  if (tv == 2) FA_LAUNCH(flash_bwd_dkv_kernel, 2, 4); else if (tv == 1) FA_LAUNCH(flash_bwd_dkv_kernel, 1, 8); else FA_LAUNCH(flash_bwd_dkv_kernel, 1, 4);
  return usseg_check_launch("flash_attn_bwd_dkv");
}
