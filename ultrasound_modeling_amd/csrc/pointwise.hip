// HBM-bound kernels of the hot path for gfx950: normalisation+activation (LayerNorm per pixel / BatchNorm
// inference affine) forward and backward, activation, 2x2 average pooling, channel-slice copies, input cast,
// column sums, operand packing.  All activations are bf16 NHWC rows read and written 16 bytes (8 channels) per
// lane; all arithmetic is fp32.
//
// "Row-chunk" mapping used by every kernel with a per-pixel or per-channel reduction: a pixel's Cphys/8 chunks
// are spread over LPP = pow2 lanes of one wave (64/LPP pixels per wave, 4 waves per workgroup), so per-pixel
// reductions are xor-shuffles inside LPP lanes and a lane keeps the SAME 8 channels for its whole grid-stride
// loop, which makes per-channel sums register accumulators that are combined once at the end.
#include "common.h"

static inline int lanes_per_pixel(int chunks) {
  int l = 1;
  while (l < chunks) l <<= 1;
  return l;
}
static inline unsigned grid_for(int64_t work_items, int per_block, int cap = 2048) {
  int64_t g = cdiv64(work_items, per_block);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// Per-channel sums without atomics: each workgroup reduces its lanes' register accumulators (xor-shuffles over the
// lanes that hold the same chunk, then LDS over the 4 waves) and stores ONE partial row ws[Cphys]; a finishing kernel
// adds the partial rows of all workgroups to the destination.  (Atomics on a handful of addresses serialise at the
// memory side: 4096 waves adding into 64 addresses cost 0.8 ms on MI355X; this form costs a few microseconds.)
__device__ __forceinline__ void block_chunk_partial(float* acc8, int LPP, int chunk, bool chunk_ok, float* ws_row, int Cphys,
                                                     float* s_red /* [4][512] */) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = acc8[j];
    for (int msk = LPP; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
    acc8[j] = v;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane < LPP && chunk_ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[wv * 512 + chunk * 8 + j] = acc8[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cphys; c += 256) ws_row[c] = s_red[c] + s_red[512 + c] + s_red[1024 + c] + s_red[1536 + c];
}

extern "C" int64_t usseg_reduce_ws_floats(void) { return (int64_t)USSEG_REDUCE_MAX_BLOCKS * 3 * 512; }

// ------------------------------------------------------------------------------------------ norm + act
#ifndef NORM_OCC_G3
#define NORM_OCC_G3 1   /* 3 waves (162-168 VGPRs) measured no faster on the latency-bound per-group LayerNorm launches, +7 us with the 24-byte spill of the fused variant */
#endif
// Butterfly sum of NG per-group partials over the LPP lanes of a pixel.  NORM_LADDER selects how the cross-lane results are consumed
// (diagnostic builds of tools/diag_norm_variants.sh, which cleared this ladder: variants 1 and 2 fail exactly like 0 - the cause of the
// load-dependent results was the unpadded packed-fp32 producer -> consumer pairs of the sum-of-squares chain):
//   0: plain C++ (`s[g] += __shfl_xor(s[g], msk)`): with packed-fp32 code generation the compiler pairs two groups into
//      ds_bpermute x3 -> s_waitcnt lgkmcnt(1) -> v_pk_add_f32 v[a:a+1], v[a:a+1], v[t:t+1]  (pair t:t+1 written by TWO LDS returns)
//   1: every returned value passes through an empty asm first: the compiler must wait lgkmcnt(0) before any add
//   2: the adds are single v_add_f32 (inline asm), whatever the rest of the kernel is compiled to
#ifndef NORM_LADDER
#define NORM_LADDER 0
#endif
template <int NG>
__device__ __forceinline__ void ladder_sum(float (&s)[NG], int LPP) {
  for (int msk = 1; msk < LPP; msk <<= 1) {
    float t[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) t[g] = __shfl_xor(s[g], msk, 64);
#if NORM_LADDER == 1
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(t[g]));
#endif
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#if NORM_LADDER == 2
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[g]) : "v"(t[g]));
#else
      s[g] += t[g];
#endif
    }
  }
}

struct NormParams {
  const bf16_t* x; const bf16_t* dy; bf16_t* y; bf16_t* dx;
  const float *gamma, *beta, *mean, *var;
  const bf16_t* mask;  // optional multiplicative mask on the activated output (dropout: 0 or 1/keep), stride ldm
  int32_t ldm;
  float *dgamma, *dbeta, *dbias, *ws;
  int64_t M;
  int32_t C, Cphys, ldx, ldy, lddy, lddx, G, Cg, mode, act, LPP;
  float eps, alpha;
  // FUSE (split attention, ResNest.py:171-199, folded into the norm that feeds / follows it; grid = (blocks per image, images)):
  //   forward:  gap_ws[(b*gridDim.x + block)*Cphys + c] = this block's sum over its pixels of the (bf16-rounded) activated output
  //             - the partial rows of the global average pool (ResNest.py:179), summed by the split-attention MLP kernel;
  //   backward: the incoming gradient is dout*sa_s[b][c] + sa_dg[b][c] (the re-weighting's backward, ResNest.py:194-197 with
  //             identical radix branches), formed in registers instead of by a separate pass over the tensor.
  int64_t HW;
  int32_t pool_ws, pool_hs;  // log2 of pool_w / pool_h when BOTH are powers of two, else -1
  int32_t pool_w, pool_h;   // POOL (backward): dy is the gradient of the 2x2-average-POOLED output [B][pool_h/2][pool_w/2]; x / dx are [B][pool_h][pool_w]
  float* gap_ws;
  const float *sa_s, *sa_dg;
  int32_t sa_cy;
  float sa_mult;
  const bf16_t* dres;   // RES (backward): dx = bf16(bf16(LN backward) + dres) - the residual branch around a pre-norm block
  int32_t lddres;
};

// MODE 0: per-pixel LayerNormalization over NG groups of channels; MODE 1: per-channel affine with given statistics.
// NG = 1 is the plain LayerNormalization (no group bookkeeping at all); NG = 4 handles up to four groups with 0/1
// membership masks folded into FMAs (the cardinal paths' LN: 3 groups of 3..85 channels, not aligned to the 8-channel
// chunks - per-element compare/select chains made that form VALU bound at ~1 TB/s).
// Waves per SIMD the register allocator is held to (measured: a variant two registers over a boundary runs 1.3-1.6x slower, and
// the allocator finds the smaller allocation without spilling when asked).
constexpr int norm_occ(bool bwd, int mode, int ng, bool fuse, bool pool) {
  if (!bwd) return ng >= 4 ? 2 : (ng == 3 ? 4 : 1);
  if (ng >= 4) return 2;
  if (ng == 3) return NORM_OCC_G3;
  return 4;
}
// (POOL sits at 130 VGPRs without the bound: two registers over the four-waves-per-SIMD line, 1.6x slower)
#ifdef NORM_NO_OCC
#define NORM_BOUNDS __launch_bounds__(256)
#else
#define NORM_BOUNDS __launch_bounds__(256, norm_occ(BWD, MODE, NG, FUSE, POOL))
#endif
template <bool BWD, int MODE, int NG, bool FUSE = false, bool POOL = false, bool RES = false>
__global__ NORM_BOUNDS void norm_act_kernel(const NormParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int LPP = p.LPP, ppw = 64 / LPP;
  const int chunk = lane & (LPP - 1), slot = lane / LPP;
  const int CH = p.Cphys >> 3;
  const bool chunk_ok = chunk < CH;
  const int c0 = chunk * 8;
  // per-lane channel constants: unconditional 16-byte loads, masked afterwards (a load under a per-element condition
  // compiles to a branch + wait per element: 32 dependent L2 round trips, ~10 us before the first pixel)
  float ga[8], be[8], mu_c[8], rs_c[8], okf[8];
  float gm[NG][8];   // gm[g][j] = 1 if channel c0+j belongs to group g (NG > 1 only)
  {
    const int cl = chunk_ok ? c0 : 0;
    float gv[8], bv[8], mv[8], vv[8];
    *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(p.gamma + cl);
    *reinterpret_cast<float4*>(gv + 4) = *reinterpret_cast<const float4*>(p.gamma + cl + 4);
    *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(p.beta + cl);
    *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(p.beta + cl + 4);
    if (MODE >= 1) {
      *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(p.mean + cl);
      *reinterpret_cast<float4*>(mv + 4) = *reinterpret_cast<const float4*>(p.mean + cl + 4);
      *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(p.var + cl);
      *reinterpret_cast<float4*>(vv + 4) = *reinterpret_cast<const float4*>(p.var + cl + 4);
    }
    const int g0 = c0 / p.Cg, r0 = c0 - g0 * p.Cg;   // group of the chunk's first channel, offset inside it
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const bool ok = chunk_ok && c < p.C;
      okf[j] = ok ? 1.f : 0.f;
      ga[j] = ok ? gv[j] : 0.f;
      be[j] = ok ? bv[j] : 0.f;
      mu_c[j] = (MODE == 1 && ok) ? mv[j] : 0.f;
      if (MODE == 2) mu_c[j] = (ok && fabsf(gv[j]) > 1e-20f) ? 1.f / gv[j] : 0.f;   // mode 2 keeps 1/gamma here (the mean is not needed)
      rs_c[j] = (MODE >= 1 && ok) ? rsqrtf(vv[j] + p.eps) : 0.f;
      if (NG > 1) {
        int gj = g0 + (r0 + j) / p.Cg;   // Cg >= 3 in every model, so a chunk spans at most 4 groups; general anyway
#pragma unroll
        for (int g = 0; g < NG; ++g) gm[g][j] = (ok && gj == g) ? 1.f : 0.f;
      }
    }
  }
  __shared__ float s_red[(BWD || FUSE) ? 4 * 512 : 1];
  float dga[8], dbe[8], dbi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { dga[j] = 0.f; dbe[j] = 0.f; dbi[j] = 0.f; }
  const float inv_cg = 1.f / (float)p.Cg;
  float sa_sv[8], sa_dgv[8];        // FUSE backward: this image's re-weighting (times mult) and pooled-path gradient per channel
  if (FUSE && BWD) {
    const int cl = chunk_ok ? c0 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cl + j < p.sa_cy ? cl + j : p.sa_cy - 1;       // clamped, unconditional loads
      sa_sv[j] = okf[j] * p.sa_mult * p.sa_s[(int64_t)blockIdx.y * p.sa_cy + c];
      sa_dgv[j] = okf[j] * p.sa_dg[(int64_t)blockIdx.y * p.sa_cy + c];
    }
  }

  const int64_t ppb = 4 * ppw;
  const int64_t m_lo = FUSE ? (int64_t)blockIdx.y * p.HW : 0, m_hi = FUSE ? m_lo + p.HW : p.M;
  for (int64_t base = m_lo + (int64_t)blockIdx.x * ppb; base < m_hi; base += (int64_t)gridDim.x * ppb) {
    const int64_t m = base + wv * ppw + slot;
    const bool valid = chunk_ok && m < m_hi;
    float xv[8], dyv[8];
    uint4 rawx = make_uint4(0, 0, 0, 0), rawd = make_uint4(0, 0, 0, 0);
    if (valid) {
      rawx = *reinterpret_cast<const uint4*>(p.x + m * p.ldx + c0);
      if (BWD && !POOL) rawd = *reinterpret_cast<const uint4*>(p.dy + m * p.lddy + c0);
      if (BWD && POOL) {     // average-pool backward folded in: every full-resolution pixel reads its pooled pixel's gradient (x 0.25 below)
        const uint32_t mm = (uint32_t)m;
        uint32_t t, xx, b, yy;
        if (p.pool_ws >= 0) {   // power-of-two image (the usual case): shifts and masks instead of two ~40-instruction integer divisions per item
          t = mm >> p.pool_ws; xx = mm & ((1u << p.pool_ws) - 1u);
          b = t >> p.pool_hs; yy = t & ((1u << p.pool_hs) - 1u);
        } else {
          t = mm / (uint32_t)p.pool_w; xx = mm - t * (uint32_t)p.pool_w;
          b = t / (uint32_t)p.pool_h; yy = t - b * (uint32_t)p.pool_h;
        }
        const int64_t pm = ((int64_t)b * (p.pool_h >> 1) + (yy >> 1)) * (p.pool_w >> 1) + (xx >> 1);
        rawd = *reinterpret_cast<const uint4*>(p.dy + pm * p.lddy + c0);
      }
    }
    unpack8(rawx, xv);
    if (BWD) unpack8(rawd, dyv);
    if (BWD && POOL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dyv[j] *= 0.25f;    // exact in bf16: what avgpool2_bwd would have stored
    }
    if (FUSE && BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dyv[j] = valid ? fmaf(dyv[j], sa_sv[j], sa_dgv[j]) : 0.f;
    }
    float xh[8];        // normalised value
    float rstd_j[8];    // 1/sigma of each element's group (MODE 0)
    if (MODE == 0) {
      float s[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) s[g] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (NG == 1) s[0] += xv[j] * okf[j];
        else
#pragma unroll
          for (int g = 0; g < NG; ++g) {
#ifdef NORM_SUMS_SCALAR   /* diagnostic: the per-group sums as single v_fmac_f32, the rest of the kernel as the compiler likes */
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[g]) : "v"(gm[g][j]), "v"(xv[j]));
#else
            s[g] = fmaf(gm[g][j], xv[j], s[g]);
#endif
          }
      }
      ladder_sum<NG>(s, LPP);
      float d[8], ss[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) ss[g] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float mean;
        if (NG == 1) mean = s[0] * inv_cg;
        else {
          mean = 0.f;
#pragma unroll
          for (int g = 0; g < NG; ++g) mean = fmaf(gm[g][j], s[g], mean);
          mean *= inv_cg;
        }
        d[j] = (xv[j] - mean) * okf[j];
        if (NG == 1) ss[0] = fmaf(d[j], d[j], ss[0]);
        else
#pragma unroll
          for (int g = 0; g < NG; ++g) {
#ifdef NORM_SUMS_SCALAR
            float gd;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(gd) : "v"(gm[g][j]), "v"(d[j]));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ss[g]) : "v"(gd), "v"(d[j]));
#else
            ss[g] = fmaf(gm[g][j] * d[j], d[j], ss[g]);
#endif
          }
      }
      ladder_sum<NG>(ss, LPP);
      float rstd_g[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) rstd_g[g] = rsqrtf(ss[g] * inv_cg + p.eps);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float r;
        if (NG == 1) r = rstd_g[0];
        else {
          r = 0.f;
#pragma unroll
          for (int g = 0; g < NG; ++g) r = fmaf(gm[g][j], rstd_g[g], r);
        }
        rstd_j[j] = r;
        xh[j] = d[j] * r;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xh[j] = (xv[j] - mu_c[j]) * rs_c[j];
    } else {
      // MODE 2 (backward only): x is the ACTIVATED output y of a conv whose epilogue applied the folded affine + activation
      // (usseg_conv2d_fwd_affine); the pre-activation and the normalised value are recovered from it (gamma != 0)
      // (LeakyReLU / ReLU / none only - checked by the launcher: an ELU inverse (log1p) in this kernel costs the registers that
      // take it from 4 to 3 waves per SIMD, 1.34x slower on every launch)
      const float inv_neg = (p.act == USSEG_ACT_LRELU && p.alpha != 0.f) ? 1.f / p.alpha : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float y = xv[j];
        xh[j] = ((y >= 0.f ? y : y * inv_neg) - be[j]) * mu_c[j];
      }
    }

    if (!BWD) {
      float o[8];
      // one uniform branch on the activation per 8 elements (apply_act's switch costs ~10 scalar branches per element)
      if (p.act == USSEG_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = ga[j] * xh[j] + be[j]; o[j] = okf[j] * (v >= 0.f ? v : p.alpha * v); }
      } else if (p.act == USSEG_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = okf[j] * (ga[j] * xh[j] + be[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = okf[j] * apply_act(ga[j] * xh[j] + be[j], p.act, p.alpha);
      }
      if (p.mask && valid) {
        float mk[8];
        unpack8(*reinterpret_cast<const uint4*>(p.mask + m * p.ldm + c0), mk);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= mk[j];
      }
      const uint4 packed = pack8(o);
      if (valid) *reinterpret_cast<uint4*>(p.y + m * p.ldy + c0) = packed;
      if (FUSE && valid) {          // global-average-pool partial of the STORED (bf16) values, as a pass over y would read them
        float r8[8];
        unpack8(packed, r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dga[j] += r8[j];
      }
    } else {
      if (p.mask && valid) {
        float mk[8];
        unpack8(*reinterpret_cast<const uint4*>(p.mask + m * p.ldm + c0), mk);
#pragma unroll
        for (int j = 0; j < 8; ++j) dyv[j] *= mk[j];
      }
      float dxh[8];
      if (MODE == 2) {
        // the activation's slope follows from the sign of the ACTIVATED value itself: one uniform branch for the 8 elements, no
        // pre-activation rebuilt through act_grad's per-element switch (this path was 1.4x slower than mode 1 with it)
        const float neg = p.act == USSEG_ACT_LRELU ? p.alpha : (p.act == USSEG_ACT_RELU ? 0.f : 1.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dh = okf[j] * dyv[j] * (xv[j] > 0.f ? 1.f : neg);
          dga[j] = fmaf(dh, xh[j], dga[j]);
          dbe[j] += dh;
          dxh[j] = dh * ga[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float pre = ga[j] * xh[j] + be[j];
          float dh = okf[j] * dyv[j] * act_grad(pre, p.act, p.alpha);
          dga[j] = fmaf(dh, xh[j], dga[j]);
          dbe[j] += dh;
          dxh[j] = dh * ga[j];
        }
      }
      float o[8];
      if (MODE == 0) {
        float s1[NG], s2[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { s1[g] = 0.f; s2[g] = 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (NG == 1) { s1[0] += dxh[j]; s2[0] = fmaf(dxh[j], xh[j], s2[0]); }
          else
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              s1[g] = fmaf(gm[g][j], dxh[j], s1[g]);
              s2[g] = fmaf(gm[g][j] * dxh[j], xh[j], s2[g]);
            }
        }
        ladder_sum<NG>(s1, LPP);
        ladder_sum<NG>(s2, LPP);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a1, a2;
          if (NG == 1) { a1 = s1[0]; a2 = s2[0]; }
          else {
            a1 = 0.f; a2 = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) { a1 = fmaf(gm[g][j], s1[g], a1); a2 = fmaf(gm[g][j], s2[g], a2); }
          }
          o[j] = okf[j] * rstd_j[j] * (dxh[j] - a1 * inv_cg - xh[j] * a2 * inv_cg);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = dxh[j] * rs_c[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (!RES) dbi[j] += o[j];   // sum of dx = gradient of the producing conv's bias
      uint4 packed = pack8(o);
      if (RES && valid) {     // + the residual branch, on the bf16-rounded value: the bits of this kernel followed by an accumulating copy
        float a8[8], r8[8];
        unpack8(packed, a8);
        unpack8(*reinterpret_cast<const uint4*>(p.dres + m * p.lddres + c0), r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] += r8[j];
        packed = pack8(a8);
        unpack8(packed, a8);  // column sums of the STORED gradient: the bias gradient of the Dense layer whose output (+ a residual) it is the gradient of
#pragma unroll
        for (int j = 0; j < 8; ++j) dbi[j] += a8[j];
      }
      if (valid) *reinterpret_cast<uint4*>(p.dx + m * p.lddx + c0) = packed;
    }
  }
  if (FUSE && !BWD) {
    block_chunk_partial(dga, LPP, chunk, chunk_ok, p.gap_ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * p.Cphys, p.Cphys, s_red);
  }
  if (BWD) {
    float* row = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 * p.Cphys;
    block_chunk_partial(dga, LPP, chunk, chunk_ok, row, p.Cphys, s_red);
    block_chunk_partial(dbe, LPP, chunk, chunk_ok, row + p.Cphys, p.Cphys, s_red);
    block_chunk_partial(dbi, LPP, chunk, chunk_ok, row + 2 * p.Cphys, p.Cphys, s_red);
  }
}

template <bool BWD, bool FUSE = false>
static void norm_launch(const NormParams& p, dim3 grid, hipStream_t s) {
  if (p.mode == 2) hipLaunchKernelGGL((norm_act_kernel<true, 2, 1>), grid, dim3(256), 0, s, p);
  else if (p.mode == 1) hipLaunchKernelGGL((norm_act_kernel<BWD, 1, 1, FUSE>), grid, dim3(256), 0, s, p);
  else if (p.G == 1) hipLaunchKernelGGL((norm_act_kernel<BWD, 0, 1, FUSE>), grid, dim3(256), 0, s, p);
  else if (p.G <= 3) hipLaunchKernelGGL((norm_act_kernel<BWD, 0, 3, FUSE>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((norm_act_kernel<BWD, 0, 4, FUSE>), grid, dim3(256), 0, s, p);
}

static int norm_common(const UssegNormDesc* d, NormParams& p) {
  USSEG_CHECK_ARG(d, "null descriptor");
  USSEG_CHECK_ARG(d->C > 0 && d->Cphys % 8 == 0 && d->Cphys >= d->C && d->Cphys <= 512, "norm: C/Cphys out of range (Cphys <= 512, multiple of 8)");
  USSEG_CHECK_ARG(d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->ldx >= d->Cphys && d->ldy >= d->Cphys, "norm: bad strides");
  USSEG_CHECK_ARG(d->mode == 0 || d->mode == 1 || d->mode == 2, "norm: mode must be 0 (LN), 1 (affine) or 2 (affine backward from the activated output)");
  int G = d->mode == 0 ? d->G : 1;
  USSEG_CHECK_ARG(G >= 1 && G <= 4 && d->C % G == 0, "norm: 1 <= G <= 4 and C % G == 0");
  p.M = d->M; p.C = d->C; p.Cphys = d->Cphys; p.G = G; p.Cg = d->mode == 0 ? d->C / G : d->C;
  p.mode = d->mode; p.act = d->act; p.eps = d->eps; p.alpha = d->alpha;
  p.LPP = lanes_per_pixel(d->Cphys / 8);
  return USSEG_OK;
}

extern "C" int usseg_norm_act_fwd(const UssegNormDesc* d, const void* x, const float* gamma, const float* beta,
                                  const float* mean, const float* var, const void* mask, int32_t ldm, void* y,
                                  usseg_stream_t stream) {
  NormParams p = {};
  int rc = norm_common(d, p);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && y && gamma && beta && (d->mode == 0 || (mean && var)), "norm fwd: null pointer");
  USSEG_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)var)) & 15) == 0,
                  "norm: gamma/beta/mean/var must be 16-byte aligned (and readable up to Cphys floats)");
  USSEG_CHECK_ARG(d->mode != 2, "norm fwd: mode 2 is a backward-only mode");
  p.x = (const bf16_t*)x; p.y = (bf16_t*)y; p.gamma = gamma; p.beta = beta; p.mean = mean; p.var = var;
  p.mask = (const bf16_t*)mask; p.ldm = ldm;
  p.ldx = d->ldx; p.ldy = d->ldy;
  if (p.M <= 0) return USSEG_OK;
  int ppb = 4 * (64 / p.LPP);
  norm_launch<false>(p, dim3(grid_for(p.M, ppb * 4)), (hipStream_t)stream);
  return usseg_check_launch("norm_act_fwd");
}

extern "C" int usseg_norm_act_bwd(const UssegNormDesc* d, const void* x, const void* dy, const float* gamma,
                                  const float* beta, const float* mean, const float* var, const void* mask, int32_t ldm,
                                  void* dx, float* dgamma, float* dbeta, float* dbias, float* ws, usseg_stream_t stream) {
  NormParams p = {};
  int rc = norm_common(d, p);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && dy && dx && gamma && beta && dgamma && dbeta && ws && (d->mode == 0 || (mean && var)), "norm bwd: null pointer");
  USSEG_CHECK_ARG(d->mode != 2 || !mask, "norm bwd mode 2 does not take a dropout mask");
  USSEG_CHECK_ARG(d->mode != 2 || d->act == USSEG_ACT_LRELU || d->act == USSEG_ACT_RELU || d->act == USSEG_ACT_NONE,
                  "norm bwd mode 2 (from the activated output) supports LeakyReLU / ReLU / no activation");
  USSEG_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)var)) & 15) == 0,
                  "norm: gamma/beta/mean/var must be 16-byte aligned (and readable up to Cphys floats)");
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dx = (bf16_t*)dx; p.gamma = gamma; p.beta = beta; p.mean = mean; p.var = var;
  p.mask = (const bf16_t*)mask; p.ldm = ldm;
  p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias; p.ws = ws;
  p.ldx = d->ldx; p.lddy = d->ldy; p.lddx = d->lddx > 0 ? d->lddx : d->ldx;
  USSEG_CHECK_ARG(p.lddx % 8 == 0 && p.lddx >= d->Cphys, "norm bwd: bad dx stride");
  if (p.M <= 0) return USSEG_OK;
  if (d->mode == 0 && !mask && d->act == USSEG_ACT_LRELU) {      // the cardinal / shortcut LayerNorms: lane-per-pixel tile kernel (cardinal.hip)
    LnTileArgs t = {};
    t.x = p.x; t.dy = p.dy; t.dx = p.dx; t.gamma = gamma; t.beta = beta; t.M = p.M; t.HW = p.M; t.C = p.C; t.Cphys = p.Cphys; t.G = p.G;
    t.ldx = p.ldx; t.lddy = p.lddy; t.lddx = p.lddx; t.eps = p.eps; t.alpha = p.alpha;
    if (usseg_try_ln_bwd_tile(t, dgamma, dbeta, dbias, ws, (hipStream_t)stream)) return usseg_check_launch("norm_act_bwd (tile)");
  }
  int ppb = 4 * (64 / p.LPP);
  // small tensors are latency bound (each loop trip is a dependent load -> store): spread them over many workgroups
  unsigned grid = grid_for(p.M, ppb * 2, USSEG_REDUCE_MAX_BLOCKS);
  p.ws = ws = usseg_defer_reduce_ws((hipStream_t)stream, ws, (int64_t)grid * 3 * p.Cphys);
  norm_launch<true>(p, dim3(grid), (hipStream_t)stream);
  usseg_launch_reduce_finish(ws, 1, (int)grid, 3, p.Cphys, p.C, 1.f, dgamma, dbeta, dbias, (hipStream_t)stream);
  return usseg_check_launch("norm_act_bwd");
}

// LayerNormalization backward + the residual branch around a pre-norm transformer block in one pass
// (VisionTransformer.py:137-146, SwinTransformer.py:224-257: x = x + f(norm(x))  =>  dx = LN'(f'(dy)) + dy): dx = bf16(bf16(LN backward) + dres).
extern "C" int usseg_norm_act_bwd_res(const UssegNormDesc* d, const void* x, const void* dy, const float* gamma, const float* beta,
                                      const void* dres, int32_t lddres, void* dx, float* dgamma, float* dbeta, float* dbias, float* ws,
                                      usseg_stream_t stream) {
  NormParams p = {};
  int rc = norm_common(d, p);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && dy && dx && dres && gamma && beta && dgamma && dbeta && ws, "norm bwd res: null pointer");
  USSEG_CHECK_ARG(d->mode == 0 && p.G == 1, "norm bwd res: plain LayerNormalization only (mode 0, one group)");
  USSEG_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "norm: gamma/beta must be 16-byte aligned (and readable up to Cphys floats)");
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dx = (bf16_t*)dx; p.gamma = gamma; p.beta = beta;
  p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias; p.ws = ws;
  p.dres = (const bf16_t*)dres; p.lddres = lddres;
  p.ldx = d->ldx; p.lddy = d->ldy; p.lddx = d->lddx > 0 ? d->lddx : d->ldx;
  USSEG_CHECK_ARG(p.lddx % 8 == 0 && p.lddx >= d->Cphys && lddres % 8 == 0 && lddres >= d->Cphys, "norm bwd res: bad dx / residual stride");
  if (p.M <= 0) return USSEG_OK;
  int ppb = 4 * (64 / p.LPP);
  unsigned grid = grid_for(p.M, ppb * 2, USSEG_REDUCE_MAX_BLOCKS);
  p.ws = ws = usseg_defer_reduce_ws((hipStream_t)stream, ws, (int64_t)grid * 3 * p.Cphys);
  hipLaunchKernelGGL((norm_act_kernel<true, 0, 1, false, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  usseg_launch_reduce_finish(ws, 1, (int)grid, 3, p.Cphys, p.C, 1.f, dgamma, dbeta, dbias, (hipStream_t)stream);
  return usseg_check_launch("norm_act_bwd_res");
}

// ---- split-attention fusions (see NormParams) ---------------------------------------------------------------------------------
extern "C" int usseg_norm_act_fwd_gap(const UssegNormDesc* d, const void* x, const float* gamma, const float* beta, const float* mean,
                                      const float* var, void* y, int32_t B, int32_t nb, float* gap_rows, usseg_stream_t stream) {
  NormParams p = {};
  int rc = norm_common(d, p);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && y && gamma && beta && gap_rows && (d->mode == 0 || (mean && var)), "norm fwd gap: null pointer");
  USSEG_CHECK_ARG(d->mode != 2 && B > 0 && nb > 0 && nb <= 1024 && d->M % B == 0, "norm fwd gap: bad arguments");
  USSEG_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)var)) & 15) == 0, "norm: unaligned channel vectors");
  p.x = (const bf16_t*)x; p.y = (bf16_t*)y; p.gamma = gamma; p.beta = beta; p.mean = mean; p.var = var;
  p.ldx = d->ldx; p.ldy = d->ldy; p.HW = d->M / B; p.gap_ws = gap_rows;
  norm_launch<false, true>(p, dim3(nb, B), (hipStream_t)stream);
  return usseg_check_launch("norm_act_fwd_gap");
}

extern "C" int usseg_norm_act_bwd_sa(const UssegNormDesc* d, const void* x, const void* dout, const float* gamma, const float* beta,
                                     const float* mean, const float* var, int32_t B, const float* sa_s, const float* sa_dg, float sa_mult,
                                     void* dx, float* dgamma, float* dbeta, float* dbias, float* ws, usseg_stream_t stream) {
  NormParams p = {};
  int rc = norm_common(d, p);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && dout && dx && gamma && beta && dgamma && dbeta && ws && sa_s && sa_dg && (d->mode == 0 || (mean && var)), "norm bwd sa: null pointer");
  USSEG_CHECK_ARG(d->mode != 2 && B > 0 && d->M % B == 0, "norm bwd sa: bad arguments");
  USSEG_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)var)) & 15) == 0, "norm: unaligned channel vectors");
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dout; p.dx = (bf16_t*)dx; p.gamma = gamma; p.beta = beta; p.mean = mean; p.var = var;
  p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
  p.ldx = d->ldx; p.lddy = d->ldy; p.lddx = d->lddx > 0 ? d->lddx : d->ldx;
  p.HW = d->M / B; p.sa_s = sa_s; p.sa_dg = sa_dg; p.sa_cy = d->C; p.sa_mult = sa_mult;
  USSEG_CHECK_ARG(p.lddx % 8 == 0 && p.lddx >= d->Cphys, "norm bwd: bad dx stride");
  if (d->mode == 0 && d->act == USSEG_ACT_LRELU && p.M > 0) {
    LnTileArgs t = {};
    t.x = p.x; t.dy = p.dy; t.dx = p.dx; t.gamma = gamma; t.beta = beta; t.M = p.M; t.HW = p.HW; t.C = p.C; t.Cphys = p.Cphys; t.G = p.G;
    t.ldx = p.ldx; t.lddy = p.lddy; t.lddx = p.lddx; t.eps = p.eps; t.alpha = p.alpha;
    t.sa_s = sa_s; t.sa_dg = sa_dg; t.sa_mult = sa_mult; t.sa_cy = p.sa_cy;
    if (usseg_try_ln_bwd_tile(t, dgamma, dbeta, dbias, ws, (hipStream_t)stream)) return usseg_check_launch("norm_act_bwd_sa (tile)");
  }
  const int ppb = 4 * (64 / p.LPP);
  int nb = (int)grid_for(p.HW, ppb * 2, USSEG_REDUCE_MAX_BLOCKS / B > 0 ? USSEG_REDUCE_MAX_BLOCKS / B : 1);
  const int64_t grid = (int64_t)nb * B;
  p.ws = ws = usseg_defer_reduce_ws((hipStream_t)stream, ws, grid * 3 * p.Cphys);
  norm_launch<true, true>(p, dim3(nb, B), (hipStream_t)stream);
  usseg_launch_reduce_finish(ws, 1, (int)grid, 3, p.Cphys, p.C, 1.f, dgamma, dbeta, dbias, (hipStream_t)stream);
  return usseg_check_launch("norm_act_bwd_sa");
}

// Two independent LayerNormalization + LeakyReLU backward passes as ONE launch: A plain (the shortcut norm of a residual_S stage,
// ResNest.py:100-101), B with the split-attention re-weighting's backward folded in (conv2_bn, :143-144 behind :194-197).  Returns
// USSEG_ERR_UNSUPPORTED without launching anything when the pair has no instantiation (the caller then issues the two calls).
extern "C" int usseg_norm_act_bwd_pair(const UssegNormDesc* da, const void* xa, const void* dya, const float* gamma_a, const float* beta_a, void* dxa,
                                       float* dgamma_a, float* dbeta_a, float* dbias_a, const UssegNormDesc* db, const void* xb, const void* doutb,
                                       const float* gamma_b, const float* beta_b, int32_t B, const float* sa_s, const float* sa_dg, float sa_mult,
                                       void* dxb, float* dgamma_b, float* dbeta_b, float* dbias_b, float* ws, usseg_stream_t stream) {
  USSEG_CHECK_ARG(da && db && xa && dya && gamma_a && beta_a && dxa && dgamma_a && dbeta_a && dbias_a && xb && doutb && gamma_b && beta_b && sa_s && sa_dg &&
                      dxb && dgamma_b && dbeta_b && dbias_b && ws, "norm bwd pair: null pointer");
  USSEG_CHECK_ARG(da->mode == 0 && db->mode == 0 && da->act == USSEG_ACT_LRELU && db->act == USSEG_ACT_LRELU && B > 0 && db->M % B == 0 && da->M > 0 && db->M > 0,
                  "norm bwd pair: LayerNormalization + LeakyReLU only");
  auto fill = [](const UssegNormDesc* d, const void* x, const void* dy, void* dx, const float* g, const float* b_, LnTileArgs& t) {
    t = {};
    t.x = (const bf16_t*)x; t.dy = (const bf16_t*)dy; t.dx = (bf16_t*)dx; t.gamma = g; t.beta = b_;
    t.M = d->M; t.HW = d->M; t.C = d->C; t.Cphys = d->Cphys; t.G = d->G > 0 ? d->G : 1;
    t.ldx = d->ldx; t.lddy = d->ldy; t.lddx = d->lddx > 0 ? d->lddx : d->ldx; t.eps = d->eps; t.alpha = d->alpha;
  };
  LnTileArgs ta, tb;
  fill(da, xa, dya, dxa, gamma_a, beta_a, ta);
  fill(db, xb, doutb, dxb, gamma_b, beta_b, tb);
  USSEG_CHECK_ARG(ta.ldx % 8 == 0 && ta.lddy % 8 == 0 && ta.lddx % 8 == 0 && tb.ldx % 8 == 0 && tb.lddy % 8 == 0 && tb.lddx % 8 == 0 && ta.lddx >= ta.Cphys &&
                      tb.lddx >= tb.Cphys, "norm bwd pair: bad strides");
  tb.HW = db->M / B; tb.sa_s = sa_s; tb.sa_dg = sa_dg; tb.sa_mult = sa_mult; tb.sa_cy = db->C;
  float* ga[3] = {dgamma_a, dbeta_a, dbias_a};
  float* gb[3] = {dgamma_b, dbeta_b, dbias_b};
  if (!usseg_try_ln_bwd_pair(ta, ga, tb, gb, ws, (hipStream_t)stream)) {
    usseg_set_error("norm bwd pair: no fused instantiation for this pair of norms");
    return USSEG_ERR_UNSUPPORTED;
  }
  return usseg_check_launch("norm_act_bwd_pair");
}

// ---- inference BatchNorm + activation + 2x2 average pool in one pass (the stem's convtmp_2bn -> LeakyReLU -> conv1_pool,
// ResNest.py:45-47; TBI_ResNest.py:90-92): the activated full-resolution tensor is consumed by the pool only, so it is never
// written (forward) and the pool's backward never materialises the upsampled gradient (backward).  Values are rounded to bf16
// where the two-kernel form stored them, so the results are bit-identical to norm_act_kernel + avgpool2_*_kernel.
template <bool BWD>
__global__ __launch_bounds__(256, 4) void bn_act_pool_kernel(const NormParams p, int Ho, int Wo) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int LPP = p.LPP, ppw = 64 / LPP;
  const int chunk = lane & (LPP - 1), slot = lane / LPP;
  const int CH = p.Cphys >> 3;
  const bool chunk_ok = chunk < CH;
  const int c0 = chunk * 8;
  float ga[8], be[8], mu_c[8], rs_c[8], okf[8];
  {
    const int cl = chunk_ok ? c0 : 0;
    float gv[8], bv[8], mv[8], vv[8];
    *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(p.gamma + cl);
    *reinterpret_cast<float4*>(gv + 4) = *reinterpret_cast<const float4*>(p.gamma + cl + 4);
    *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(p.beta + cl);
    *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(p.beta + cl + 4);
    *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(p.mean + cl);
    *reinterpret_cast<float4*>(mv + 4) = *reinterpret_cast<const float4*>(p.mean + cl + 4);
    *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(p.var + cl);
    *reinterpret_cast<float4*>(vv + 4) = *reinterpret_cast<const float4*>(p.var + cl + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = chunk_ok && c0 + j < p.C;
      okf[j] = ok ? 1.f : 0.f;
      ga[j] = ok ? gv[j] : 0.f;
      be[j] = ok ? bv[j] : 0.f;
      mu_c[j] = ok ? mv[j] : 0.f;
      rs_c[j] = ok ? rsqrtf(vv[j] + p.eps) : 0.f;
    }
  }
  __shared__ float s_red[BWD ? 4 * 512 : 1];
  float dga[8], dbe[8], dbi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { dga[j] = 0.f; dbe[j] = 0.f; dbi[j] = 0.f; }
  const int Wi = 2 * Wo;
  const int64_t ppb = 4 * ppw;
  for (int64_t base = (int64_t)blockIdx.x * ppb; base < p.M; base += (int64_t)gridDim.x * ppb) {
    const int64_t m = base + wv * ppw + slot;          // pooled pixel
    const bool valid = chunk_ok && m < p.M;
    const uint32_t mc = valid ? (uint32_t)m : 0u;      // 32-bit decode (the launcher checks M < 2^31): 64-bit div / mod cost ~100 cycles each
    const uint32_t t = mc / (uint32_t)Wo;
    const int ox = (int)(mc - t * (uint32_t)Wo);
    const uint32_t b = t / (uint32_t)Ho;
    const int oy = (int)(t - b * (uint32_t)Ho);
    const int64_t pi = (((int64_t)b * 2 * Ho + 2 * oy) * Wi + 2 * ox);
    const int64_t sub[4] = {pi, pi + 1, pi + Wi, pi + Wi + 1};
    uint4 raw[4];
    if (!BWD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[q] = valid ? *reinterpret_cast<const uint4*>(p.x + sub[q] * p.ldx + c0) : make_uint4(0, 0, 0, 0);
    }
    if (!BWD) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float xv[8], o[8], r8[8];
        unpack8(raw[q], xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = okf[j] * apply_act(ga[j] * ((xv[j] - mu_c[j]) * rs_c[j]) + be[j], p.act, p.alpha);
        unpack8(pack8(o), r8);                         // the value the unfused form stored
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += r8[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= 0.25f;
      if (valid) *reinterpret_cast<uint4*>(p.y + m * p.ldy + c0) = pack8(acc);
    } else {
      float dyv[8];
      unpack8(valid ? *reinterpret_cast<const uint4*>(p.dy + m * p.lddy + c0) : make_uint4(0, 0, 0, 0), dyv);
#pragma unroll
      for (int j = 0; j < 8; ++j) dyv[j] *= 0.25f;     // exact in bf16: the upsampled gradient the pool backward would have stored
      // one uniform branch on the activation for the whole item (act_grad's per-element switch kept every variant's temporaries
      // alive: 204 VGPRs, two waves per SIMD)
      auto body = [&](auto slope) {
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {      // rolled: one sub-pixel's temporaries at a time (four waves per SIMD instead of two)
          float xv[8], o[8];
          const int64_t sp = pi + (q & 1) + (q >> 1) * Wi;
          unpack8(valid ? *reinterpret_cast<const uint4*>(p.x + sp * p.ldx + c0) : make_uint4(0, 0, 0, 0), xv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mu_c[j]) * rs_c[j];
            const float dh = dyv[j] * slope(ga[j] * xh + be[j]);      // dyv is zero in pad channels / invalid lanes
            dga[j] = fmaf(dh, xh, dga[j]);
            dbe[j] += dh;
            o[j] = dh * ga[j] * rs_c[j];
            dbi[j] += o[j];
          }
          if (valid) *reinterpret_cast<uint4*>(p.dx + sp * p.lddx + c0) = pack8(o);
        }
      };
#pragma unroll
      for (int j = 0; j < 8; ++j) dyv[j] *= okf[j];
      const float alpha = p.alpha;
      if (p.act == USSEG_ACT_LRELU) body([alpha](float v) { return v >= 0.f ? 1.f : alpha; });
      else if (p.act == USSEG_ACT_ELU) body([alpha](float v) { return v > 0.f ? 1.f : alpha * __expf(v); });
      else if (p.act == USSEG_ACT_RELU) body([](float v) { return v > 0.f ? 1.f : 0.f; });
      else body([](float) { return 1.f; });
    }
  }
  if (BWD) {
    float* row = p.ws + (int64_t)blockIdx.x * 3 * p.Cphys;
    block_chunk_partial(dga, LPP, chunk, chunk_ok, row, p.Cphys, s_red);
    block_chunk_partial(dbe, LPP, chunk, chunk_ok, row + p.Cphys, p.Cphys, s_red);
    block_chunk_partial(dbi, LPP, chunk, chunk_ok, row + 2 * p.Cphys, p.Cphys, s_red);
  }
}

static int bn_pool_common(NormParams& p, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cphys, const float* gamma, const float* beta,
                          const float* mean, const float* var, float eps, int32_t act, float alpha) {
  USSEG_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && Cphys % 8 == 0 && Cphys >= C && Cphys <= 512, "bn_act_pool: bad geometry");
  USSEG_CHECK_ARG((int64_t)B * H * W < (1ll << 31), "bn_act_pool: more than 2^31 pixels");
  USSEG_CHECK_ARG(gamma && beta && mean && var && ((((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)var)) & 15) == 0,
                  "bn_act_pool: channel vectors must be 16-byte aligned");
  p.M = (int64_t)B * (H / 2) * (W / 2); p.C = C; p.Cphys = Cphys; p.G = 1; p.Cg = C; p.mode = 1; p.act = act; p.eps = eps; p.alpha = alpha;
  p.gamma = gamma; p.beta = beta; p.mean = mean; p.var = var;
  p.LPP = lanes_per_pixel(Cphys / 8);
  return USSEG_OK;
}
extern "C" int usseg_bn_act_pool_fwd(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cphys, int32_t ldx, const float* gamma,
                                     const float* beta, const float* mean, const float* var, float eps, int32_t act, float alpha, void* y,
                                     int32_t ldy, usseg_stream_t stream) {
  NormParams p = {};
  int rc = bn_pool_common(p, B, H, W, C, Cphys, gamma, beta, mean, var, eps, act, alpha);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && y && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= Cphys && ldy >= Cphys, "bn_act_pool_fwd: bad pointers / strides");
  p.x = (const bf16_t*)x; p.y = (bf16_t*)y; p.ldx = ldx; p.ldy = ldy;
  const int ppb = 4 * (64 / p.LPP);
  hipLaunchKernelGGL(bn_act_pool_kernel<false>, dim3(grid_for(p.M, ppb * 2)), dim3(256), 0, (hipStream_t)stream, p, H / 2, W / 2);
  return usseg_check_launch("bn_act_pool_fwd");
}
extern "C" int usseg_bn_act_pool_bwd(const void* x, const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cphys, int32_t ldx,
                                     int32_t lddy, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                     int32_t act, float alpha, void* dx, int32_t lddx, float* dgamma, float* dbeta, float* dbias, float* ws,
                                     usseg_stream_t stream) {
  NormParams p = {};
  int rc = bn_pool_common(p, B, H, W, C, Cphys, gamma, beta, mean, var, eps, act, alpha);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && dy && dx && dgamma && dbeta && ws && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldx >= Cphys && lddy >= Cphys && lddx >= Cphys,
                  "bn_act_pool_bwd: bad pointers / strides");
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dx = (bf16_t*)dx; p.ldx = ldx; p.lddy = lddy; p.lddx = lddx;
  p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
  // backward = the per-channel-affine norm backward over the FULL-resolution pixels (coalesced x / dx streams) whose incoming
  // gradient is read through the pool's index map: a thread per pooled pixel (bn_act_pool_kernel<true>) touches four strided
  // pixels per item and reached 3.0 TB/s, this form runs at the norm backward's ~5 TB/s
  p.M = (int64_t)B * H * W; p.pool_w = W; p.pool_h = H;
  p.pool_ws = p.pool_hs = -1;
  if ((W & (W - 1)) == 0 && (H & (H - 1)) == 0) { p.pool_ws = __builtin_ctz(W); p.pool_hs = __builtin_ctz(H); }
  const int ppb = 4 * (64 / p.LPP);
  unsigned grid = grid_for(p.M, ppb * 2, USSEG_REDUCE_MAX_BLOCKS);
  p.ws = ws = usseg_defer_reduce_ws((hipStream_t)stream, ws, (int64_t)grid * 3 * p.Cphys);
  hipLaunchKernelGGL((norm_act_kernel<true, 1, 1, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  usseg_launch_reduce_finish(ws, 1, (int)grid, 3, p.Cphys, p.C, 1.f, dgamma, dbeta, dbias, (hipStream_t)stream);
  return usseg_check_launch("bn_act_pool_bwd");
}

// ------------------------------------------------------------------------------------------ column sums
// MODE 0: out[c] += sum_m a[m][c]; MODE 1: out[c] += sum a, out2[c] += sum a^2
// Rows wider than 512 channels: grid.y walks 512-channel slabs (partial rows of slab y at ws + y * gridDim.x * rows-per-block * 512).
template <int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* a, int64_t M, int Ctot, int ld, int LPP, float* ws) {
  __shared__ float s_red[4 * 512];
  const int c0 = blockIdx.y * 512, C = Ctot - c0 < 512 ? Ctot - c0 : 512;
  a += c0;
  ws += (int64_t)blockIdx.y * gridDim.x * (MODE + 1) * 512;
  const int Cp = gridDim.y > 1 ? 512 : (C + 7) & ~7;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ppw = 64 / LPP, chunk = lane & (LPP - 1), slot = lane / LPP;
  const bool chunk_ok = chunk * 8 < C;
  float s[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; s2[j] = 0.f; }
  const int64_t ppb = 4 * ppw, stride = (int64_t)gridDim.x * ppb;
  // four rows per lane and trip, loaded at clamped addresses before the first use (one load per trip was a chain of 8 dependent HBM latencies)
  for (int64_t base = (int64_t)blockIdx.x * ppb; base < M; base += 4 * stride) {
    uint4 raw[4];
    float okf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t m = base + u * stride + wv * ppw + slot;
      const bool ok = chunk_ok && m < M;
      okf[u] = ok ? 1.f : 0.f;
      raw[u] = *reinterpret_cast<const uint4*>(a + (ok ? m : 0) * ld + (chunk_ok ? chunk * 8 : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      unpack8(raw[u], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += okf[u] * v[j]; if (MODE == 1) s2[j] += okf[u] * v[j] * v[j]; }
    }
  }
  float* row = ws + (int64_t)blockIdx.x * (MODE + 1) * Cp;
  block_chunk_partial(s, LPP, chunk, chunk_ok, row, Cp, s_red);
  if (MODE == 1) block_chunk_partial(s2, LPP, chunk, chunk_ok, row + Cp, Cp, s_red);
}

extern "C" int usseg_colsum(const void* dy, int64_t M, int32_t C, int32_t ld, float* db, float* ws, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dy && db && ws && C > 0 && ld % 8 == 0 && C <= 16384, "colsum: bad args");
  if (M <= 0) return USSEG_OK;
  // wide rows: 512-channel slabs along grid.y of ONE launch
  const int nslab = (C + 511) / 512;
  const int cwp = nslab > 1 ? 512 : roundup(C, 8);
  const int LPP = lanes_per_pixel(cwp / 8);
  const int ppb = 4 * (64 / LPP);
  unsigned grid = grid_for(M, ppb * 8, USSEG_REDUCE_MAX_BLOCKS * 3 / nslab < USSEG_REDUCE_MAX_BLOCKS ? USSEG_REDUCE_MAX_BLOCKS * 3 / nslab : USSEG_REDUCE_MAX_BLOCKS);
  float* wsr = usseg_defer_reduce_ws((hipStream_t)stream, ws, (int64_t)grid * nslab * cwp);
  hipLaunchKernelGGL(colsum_kernel<0>, dim3(grid, nslab), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, M, C, ld, LPP, wsr);
  for (int sl = 0; sl < nslab; ++sl) {
    const int cw = C - sl * 512 < 512 ? C - sl * 512 : 512;
    usseg_launch_reduce_finish(wsr + (int64_t)sl * grid * cwp, 1, (int)grid, 1, cwp, cw, 1.f, db + sl * 512, nullptr, nullptr, (hipStream_t)stream);
  }
  return usseg_check_launch("colsum");
}

extern "C" int usseg_channel_stats(const void* x, int64_t M, int32_t C, int32_t ldx, float* sum, float* sumsq, float* ws,
                                   usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && sum && sumsq && ws && C > 0 && C <= 512 && ldx % 8 == 0, "channel_stats: bad args");
  if (M <= 0) return USSEG_OK;
  int LPP = lanes_per_pixel(roundup(C, 8) / 8);
  int ppb = 4 * (64 / LPP);
  unsigned grid = grid_for(M, ppb * 8, USSEG_REDUCE_MAX_BLOCKS);
  hipLaunchKernelGGL(colsum_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, M, C, ldx, LPP, ws);
  usseg_launch_reduce_finish(ws, 1, (int)grid, 2, roundup(C, 8), C, 1.f, sum, sumsq, nullptr, (hipStream_t)stream);
  return usseg_check_launch("channel_stats");
}

// ------------------------------------------------------------------------------------------ elementwise (chunk per thread)
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16_t* x, int64_t M, int CH, int ldx, int ldy, int act, float alpha, bf16_t* y) {
  const int64_t total = M * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / CH;
    int c0 = (int)(i - m * CH) * 8;
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(x + m * ldx + c0), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], act, alpha);
    *reinterpret_cast<uint4*>(y + m * ldy + c0) = pack8(v);
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* x, const bf16_t* dy, int64_t M, int CH, int ldx, int lddy, int lddx,
                                                       int act, float alpha, bf16_t* dx) {
  const int64_t total = M * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / CH;
    int c0 = (int)(i - m * CH) * 8;
    float v[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(x + m * ldx + c0), v);
    unpack8(*reinterpret_cast<const uint4*>(dy + m * lddy + c0), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= act_grad(v[j], act, alpha);
    *reinterpret_cast<uint4*>(dx + m * lddx + c0) = pack8(g);
  }
}
// activation backward that also yields the column sums of its output (the bias gradient of the conv whose fused activation this
// undoes - ResNest.py:39-40 conv1 + LeakyReLU): one pass instead of act_bwd + colsum.  Sums the STORED (bf16) values, as colsum would.
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const bf16_t* x, const bf16_t* dy, int64_t M, int Ctot, int ldx, int lddy, int lddx, int LPP,
                                                              int act, float alpha, bf16_t* dx, float* ws) {
  __shared__ float s_red[4 * 512];
  const int c0 = blockIdx.y * 512, C = Ctot - c0 < 512 ? Ctot - c0 : 512;      // 512-channel slabs along grid.y
  x += c0; dy += c0; dx += c0;
  ws += (int64_t)blockIdx.y * gridDim.x * 512;
  const int Cp = gridDim.y > 1 ? 512 : (C + 7) & ~7;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ppw = 64 / LPP, chunk = lane & (LPP - 1), slot = lane / LPP;
  const bool chunk_ok = chunk * 8 < C;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int64_t ppb = 4 * ppw;
  for (int64_t base = (int64_t)blockIdx.x * ppb; base < M; base += (int64_t)gridDim.x * ppb) {
    const int64_t m = base + wv * ppw + slot;
    if (chunk_ok && m < M) {
      float v[8], g[8], r8[8];
      unpack8(*reinterpret_cast<const uint4*>(x + m * ldx + chunk * 8), v);
      unpack8(*reinterpret_cast<const uint4*>(dy + m * lddy + chunk * 8), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] *= act_grad(v[j], act, alpha);
      const uint4 o = pack8(g);
      *reinterpret_cast<uint4*>(dx + m * lddx + chunk * 8) = o;
      unpack8(o, r8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += r8[j];
    }
  }
  block_chunk_partial(s, LPP, chunk, chunk_ok, ws + (int64_t)blockIdx.x * Cp, Cp, s_red);
}
extern "C" int usseg_act_bwd_colsum(const void* x, const void* dy, int64_t M, int32_t C, int32_t ldx, int32_t lddy, int32_t lddx, int32_t act,
                                    float alpha, void* dx, float* db, float* ws, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && dy && dx && db && ws && C > 0 && C <= 16384 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "act_bwd_colsum: bad args");
  if (M <= 0) return USSEG_OK;
  const int nslab = (C + 511) / 512;
  const int cwp = nslab > 1 ? 512 : roundup(C, 8);
  const int LPP = lanes_per_pixel(cwp / 8);
  const int ppb = 4 * (64 / LPP);
  unsigned grid = grid_for(M, ppb * 8, USSEG_REDUCE_MAX_BLOCKS * 3 / nslab < USSEG_REDUCE_MAX_BLOCKS ? USSEG_REDUCE_MAX_BLOCKS * 3 / nslab : USSEG_REDUCE_MAX_BLOCKS);
  float* wsr = usseg_defer_reduce_ws((hipStream_t)stream, ws, (int64_t)grid * nslab * cwp);
  hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3(grid, nslab), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, M, C, ldx, lddy,
                     lddx, LPP, act, alpha, (bf16_t*)dx, wsr);
  for (int sl = 0; sl < nslab; ++sl) {
    const int cw = C - sl * 512 < 512 ? C - sl * 512 : 512;
    usseg_launch_reduce_finish(wsr + (int64_t)sl * grid * cwp, 1, (int)grid, 1, cwp, cw, 1.f, db + sl * 512, nullptr, nullptr, (hipStream_t)stream);
  }
  return usseg_check_launch("act_bwd_colsum");
}

extern "C" int usseg_act_fwd(const void* x, int64_t M, int32_t C, int32_t ldx, int32_t ldy, int32_t act, float alpha, void* y,
                             usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "act_fwd: bad args");
  if (M <= 0) return USSEG_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(M * (C / 8), 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, M, C / 8, ldx, ldy, act, alpha, (bf16_t*)y);
  return usseg_check_launch("act_fwd");
}
extern "C" int usseg_act_bwd(const void* x, const void* dy, int64_t M, int32_t C, int32_t ldx, int32_t lddy, int32_t lddx,
                             int32_t act, float alpha, void* dx, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && dy && dx && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "act_bwd: bad args");
  if (M <= 0) return USSEG_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(M * (C / 8), 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)dy, M, C / 8, ldx, lddy, lddx, act, alpha, (bf16_t*)dx);
  return usseg_check_launch("act_bwd");
}

__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16_t* x, int B, int Ho, int Wo, int CH, int ldx, int ldy, bf16_t* y) {
  const int64_t total = (int64_t)B * Ho * Wo * CH;
  const int Wi = 2 * Wo, Hi = 2 * Ho;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t pix = i / CH;
    int c0 = (int)(i - pix * CH) * 8;
    int ox = (int)(pix % Wo);
    int64_t t = pix / Wo;
    int oy = (int)(t % Ho);
    int b = (int)(t / Ho);
    const bf16_t* s = x + (((int64_t)b * Hi + 2 * oy) * Wi + 2 * ox) * ldx + c0;
    float a[8], v[8];
    unpack8(*reinterpret_cast<const uint4*>(s), a);
    unpack8(*reinterpret_cast<const uint4*>(s + ldx), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += v[j];
    unpack8(*reinterpret_cast<const uint4*>(s + (int64_t)Wi * ldx), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += v[j];
    unpack8(*reinterpret_cast<const uint4*>(s + (int64_t)Wi * ldx + ldx), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (a[j] + v[j]) * 0.25f;
    *reinterpret_cast<uint4*>(y + pix * ldy + c0) = pack8(a);
  }
}
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16_t* dy, int B, int H, int W, int CH, int lddy, int lddx,
                                                            const bf16_t* add, int ldadd, bf16_t* dx) {
  const int64_t total = (int64_t)B * H * W * CH;
  const int Ho = H / 2, Wo = W / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t pix = i / CH;
    int c0 = (int)(i - pix * CH) * 8;
    int x_ = (int)(pix % W);
    int64_t t = pix / W;
    int y_ = (int)(t % H);
    int b = (int)(t / H);
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + (((int64_t)b * Ho + (y_ >> 1)) * Wo + (x_ >> 1)) * lddy + c0), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= 0.25f;
    if (add) {
      float a[8];
      unpack8(*reinterpret_cast<const uint4*>(add + pix * ldadd + c0), a);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += a[j];
    }
    *reinterpret_cast<uint4*>(dx + pix * lddx + c0) = pack8(v);
  }
}
// avgpool2 backward (+ skip-connection gradient `add`) that also yields the column sums of its output: dx of the pool behind a
// residual_S stage is the gradient w.r.t. that stage's concats_2 output, whose sum over the pixels is concats_2's bias gradient
// (ResNest.py:98,49-54) - one pass instead of avgpool2_bwd + colsum.  Sums the STORED (bf16) values, as colsum would.
__global__ __launch_bounds__(256) void avgpool2_bwd_colsum_kernel(const bf16_t* dy, int B, int H, int W, int C, int lddy, int lddx, const bf16_t* add,
                                                                   int ldadd, int LPP, bf16_t* dx, float* ws) {
  __shared__ float s_red[4 * 512];
  const int Cp = (C + 7) & ~7;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ppw = 64 / LPP, chunk = lane & (LPP - 1), slot = lane / LPP;
  const bool chunk_ok = chunk * 8 < C;
  const int Ho = H / 2, Wo = W / 2;
  const int64_t M = (int64_t)B * H * W;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int64_t ppb = 4 * ppw;
  for (int64_t base = (int64_t)blockIdx.x * ppb; base < M; base += (int64_t)gridDim.x * ppb) {
    const int64_t pix = base + wv * ppw + slot;
    if (chunk_ok && pix < M) {
      const int x_ = (int)(pix % W);
      const int64_t t = pix / W;
      const int y_ = (int)(t % H);
      const int64_t b = t / H;
      float v[8], r8[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + ((b * Ho + (y_ >> 1)) * Wo + (x_ >> 1)) * lddy + chunk * 8), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= 0.25f;
      if (add) {
        float a[8];
        unpack8(*reinterpret_cast<const uint4*>(add + pix * ldadd + chunk * 8), a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += a[j];
      }
      const uint4 o = pack8(v);
      *reinterpret_cast<uint4*>(dx + pix * lddx + chunk * 8) = o;
      unpack8(o, r8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += r8[j];
    }
  }
  block_chunk_partial(s, LPP, chunk, chunk_ok, ws + (int64_t)blockIdx.x * Cp, Cp, s_red);
}
extern "C" int usseg_avgpool2_bwd_colsum(const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t lddy, int32_t lddx, const void* add,
                                         int32_t ldadd, void* dx, float* db, float* ws, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dy && dx && db && ws && H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && C <= 512 && lddy % 8 == 0 && lddx % 8 == 0, "avgpool2_bwd_colsum: bad args");
  const int64_t M = (int64_t)B * H * W;
  if (M <= 0) return USSEG_OK;
  const int LPP = lanes_per_pixel(C / 8);
  const int ppb = 4 * (64 / LPP);
  unsigned grid = grid_for(M, ppb * 4, USSEG_REDUCE_MAX_BLOCKS);
  float* wsr = usseg_defer_reduce_ws((hipStream_t)stream, ws, (int64_t)grid * C);
  hipLaunchKernelGGL(avgpool2_bwd_colsum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, B, H, W, C, lddy, lddx,
                     (const bf16_t*)add, ldadd, LPP, (bf16_t*)dx, wsr);
  usseg_launch_reduce_finish(wsr, 1, (int)grid, 1, C, C, 1.f, db, nullptr, nullptr, (hipStream_t)stream);
  return usseg_check_launch("avgpool2_bwd_colsum");
}

extern "C" int usseg_avgpool2_fwd(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t ldy, void* y,
                                  usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && y && H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "avgpool2_fwd: bad args");
  int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  if (total <= 0) return USSEG_OK;
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(grid_for(total, 256 * 2, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, B, H / 2, W / 2, C / 8, ldx, ldy, (bf16_t*)y);
  return usseg_check_launch("avgpool2_fwd");
}
extern "C" int usseg_avgpool2_bwd(const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t lddy, int32_t lddx,
                                  const void* add, int32_t ldadd, void* dx, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dy && dx && H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "avgpool2_bwd: bad args");
  int64_t total = (int64_t)B * H * W * (C / 8);
  if (total <= 0) return USSEG_OK;
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(total, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, B, H, W, C / 8, lddy, lddx, (const bf16_t*)add, ldadd, (bf16_t*)dx);
  return usseg_check_launch("avgpool2_bwd");
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const bf16_t* src, int64_t M, int CH, int lds_, bf16_t* dst, int ldd, int accumulate) {
  const int64_t total = M * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / CH;
    int c0 = (int)(i - m * CH) * 8;
    uint4 v = *reinterpret_cast<const uint4*>(src + m * lds_ + c0);
    if (accumulate) {
      float a[8], b[8];
      unpack8(v, a);
      unpack8(*reinterpret_cast<const uint4*>(dst + m * ldd + c0), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
      v = pack8(a);
    }
    *reinterpret_cast<uint4*>(dst + m * ldd + c0) = v;
  }
}
extern "C" int usseg_copy_channels(const void* src, int64_t M, int32_t C, int32_t lds_, void* dst, int32_t ldd, int32_t accumulate,
                                   usseg_stream_t stream) {
  USSEG_CHECK_ARG(src && dst && C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "copy_channels: bad args");
  if (M <= 0) return USSEG_OK;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(M * (C / 8), 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, M, C / 8, lds_, (bf16_t*)dst, ldd, accumulate);
  return usseg_check_launch("copy_channels");
}

// The decoder re-injects the hidden state at every scale through a RAW row-major reshape (Decoder.py:140-141):
// hidden [B][N][hidden] seen as [B][gh*s][gw*s][c0] is concatenated behind the block output, c0 = hidden / 4^(i+1).  All three
// scales in ONE launch: forward scatters each 8-element vector of the hidden state into its three destinations, backward
// gathers the three gradient slices, adds them in fp32 and writes d_hidden once (three copy launches each way before).
struct ReinjectParams {
  bf16_t* buf[4];
  int32_t c0[4], ld[4];
  int32_t n;
};
__global__ __launch_bounds__(256) void reinject_hidden_kernel(bf16_t* hidden, int64_t nvec, const ReinjectParams q, int backward) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    const int64_t e = v * 8;
    if (!backward) {
      const uint4 val = *reinterpret_cast<const uint4*>(hidden + e);
      for (int i = 0; i < q.n; ++i) {
        const int64_t pix = e / q.c0[i];
        *reinterpret_cast<uint4*>(q.buf[i] + pix * q.ld[i] + (int)(e - pix * q.c0[i])) = val;
      }
    } else {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, t[8];
      for (int i = 0; i < q.n; ++i) {
        const int64_t pix = e / q.c0[i];
        unpack8(*reinterpret_cast<const uint4*>(q.buf[i] + pix * q.ld[i] + (int)(e - pix * q.c0[i])), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += t[j];
      }
      *reinterpret_cast<uint4*>(hidden + e) = pack8(acc);
    }
  }
}
extern "C" int usseg_reinject_hidden(void* hidden, int64_t numel, int32_t n, void* const* bufs, const int32_t* c0, const int32_t* ld,
                                     int32_t backward, usseg_stream_t stream) {
  USSEG_CHECK_ARG(hidden && bufs && c0 && ld && n >= 1 && n <= 4 && numel % 8 == 0, "reinject_hidden: bad args");
  ReinjectParams q = {};
  q.n = n;
  for (int i = 0; i < n; ++i) {
    USSEG_CHECK_ARG(bufs[i] && c0[i] >= 8 && c0[i] % 8 == 0 && ld[i] % 8 == 0 && ld[i] >= c0[i] && numel % c0[i] == 0, "reinject_hidden: bad view");
    q.buf[i] = (bf16_t*)bufs[i]; q.c0[i] = c0[i]; q.ld[i] = ld[i];
  }
  if (numel <= 0) return USSEG_OK;
  hipLaunchKernelGGL(reinject_hidden_kernel, dim3(grid_for(numel / 8, 256 * 2, 4096)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)hidden, numel / 8, q,
                     backward);
  return usseg_check_launch("reinject_hidden");
}

template <typename T>
__global__ __launch_bounds__(256) void cast_input_kernel(const T* src, int64_t M, int C, bf16_t* dst, int Cphys) {
  const int CH = Cphys / 8;
  const int64_t total = M * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / CH;
    int c0 = (int)(i - m * CH) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = (c0 + j < C) ? (float)src[m * C + c0 + j] : 0.f;
      // fp64 input: the float32 value (Keras' layer autocast) must exist before the bf16 rounding - without this fence the
      // compiler fuses double -> float -> bf16 into one rounding, which differs on exact float ties (seen 4 in 600k)
      asm volatile("" : "+v"(v[j]));
    }
    *reinterpret_cast<uint4*>(dst + m * Cphys + c0) = pack8(v);
  }
}
extern "C" int usseg_cast_input(const void* src, int32_t src_is_f64, int64_t M, int32_t C, void* dst, int32_t Cphys,
                                usseg_stream_t stream) {
  USSEG_CHECK_ARG(src && dst && C > 0 && Cphys % 8 == 0 && Cphys >= C, "cast_input: bad args");
  if (M <= 0) return USSEG_OK;
  dim3 grid(grid_for(M * (Cphys / 8), 256 * 2, 4096));
  if (src_is_f64)
    hipLaunchKernelGGL(cast_input_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)src, M, C, (bf16_t*)dst, Cphys);
  else
    hipLaunchKernelGGL(cast_input_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, M, C, (bf16_t*)dst, Cphys);
  return usseg_check_launch("cast_input");
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* src, int64_t M, int C, int lds_, float* dst) {
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / C;
    int c = (int)(i - m * C);
    dst[i] = bf2f(src[m * lds_ + c]);
  }
}
extern "C" int usseg_cast_bf16_to_f32(const void* src, int64_t M, int32_t C, int32_t lds_, float* dst, usseg_stream_t stream) {
  USSEG_CHECK_ARG(src && dst && C > 0, "cast_bf16_to_f32: bad args");
  if (M <= 0) return USSEG_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(M * C, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, M, C, lds_, dst);
  return usseg_check_launch("cast_bf16_to_f32");
}

// ------------------------------------------------------------------------------------------ operand packing
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* src, int64_t sT, int64_t sN, int64_t sK, int T, int Nn, int Kk,
                                                           bf16_t* dst, int Kw, int tap_stride, int n_off, int k_off) {
  const int64_t total = (int64_t)T * Nn * Kk;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int k = (int)(i % Kk);
    int64_t r = i / Kk;
    int n = (int)(r % Nn);
    int t = (int)(r / Nn);
    dst[(int64_t)(n_off + n) * Kw + (int64_t)t * tap_stride + k_off + k] = f2bf(src[t * sT + n * sN + k * sK]);
  }
}
extern "C" int usseg_pack_weight(const float* src, int64_t sT, int64_t sN, int64_t sK, int32_t T, int32_t Nn, int32_t Kk, void* dst,
                                 int32_t Kw, int32_t tap_stride, int32_t n_off, int32_t k_off, usseg_stream_t stream) {
  USSEG_CHECK_ARG(src && dst && T > 0 && Nn > 0 && Kk > 0 && k_off + Kk <= tap_stride && T * tap_stride <= Kw, "pack_weight: bad args");
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for((int64_t)T * Nn * Kk, 256, 1024)), dim3(256), 0, (hipStream_t)stream, src, sT,
                     sN, sK, T, Nn, Kk, (bf16_t*)dst, Kw, tap_stride, n_off, k_off);
  return usseg_check_launch("pack_weight");
}

// scratch is [T][Mrows][Ncols] with m = input-channel (K side of the forward operand), n = output channel
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* scratch, int Mrows, int Ncols, int T, int Nn, int Kk, int n_off,
                                                            int k_off, float* dst, int64_t sT, int64_t sN, int64_t sK, float scale,
                                                            int accumulate) {
  const int64_t total = (int64_t)T * Nn * Kk;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int n = (int)(i % Nn);
    int64_t r = i / Nn;
    int k = (int)(r % Kk);
    int t = (int)(r / Kk);
    float v = scale * scratch[((int64_t)t * Mrows + k_off + k) * Ncols + n_off + n];
    float* d = dst + t * sT + n * sN + k * sK;
    *d = accumulate ? *d + v : v;
  }
}
extern "C" int usseg_unpack_wgrad(const float* scratch, int32_t Mrows, int32_t Ncols, int32_t T, int32_t Nn, int32_t Kk, int32_t n_off,
                                  int32_t k_off, float* dst, int64_t sT, int64_t sN, int64_t sK, float scale, int32_t accumulate,
                                  usseg_stream_t stream) {
  USSEG_CHECK_ARG(scratch && dst && T > 0 && Nn > 0 && Kk > 0 && n_off + Nn <= Ncols && k_off + Kk <= Mrows, "unpack_wgrad: bad args");
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for((int64_t)T * Nn * Kk, 256, 1024)), dim3(256), 0, (hipStream_t)stream, scratch,
                     Mrows, Ncols, T, Nn, Kk, n_off, k_off, dst, sT, sN, sK, scale, accumulate);
  return usseg_check_launch("unpack_wgrad");
}

// several unpacks (the per-branch diagonal blocks of a grouped weight gradient) in ONE launch: blockIdx.y = job
__global__ __launch_bounds__(256) void unpack_wgrad_batched_kernel(const UssegUnpackJob* jobs) {
  const UssegUnpackJob j = jobs[blockIdx.y];
  const int total = j.T * j.Nn * j.Kk;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int n = i % j.Nn;
    int r = i / j.Nn;
    int k = r % j.Kk;
    int t = r / j.Kk;
    float v = j.scale * j.scratch[((int64_t)t * j.Mrows + j.k_off + k) * j.Ncols + j.n_off + n];
    float* d = j.dst + t * j.sT + n * j.sN + k * j.sK;
    *d = j.accumulate ? *d + v : v;
  }
}
extern "C" int usseg_unpack_wgrad_batched(const UssegUnpackJob* jobs_dev, int32_t njobs, int32_t max_elems, usseg_stream_t stream) {
  USSEG_CHECK_ARG(jobs_dev && njobs > 0 && njobs < 65536 && max_elems > 0, "unpack_wgrad_batched: bad args");
  int gx = (max_elems + 255) / 256;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(unpack_wgrad_batched_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return usseg_check_launch("unpack_wgrad_batched");
}

// ---- all operand packs of a model in ONE launch: blockIdx.y = job ---------------------------------------------
// FLAT: blockIdx.x indexes a host-built (job, first tile, tile count) list covering every 32x32 tile of every job exactly once - no
// empty workgroups (the 512 x njobs grid of the other form launches 87 000 workgroups for 12 000 tiles on Arch B).
template <bool FLAT>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const UssegPackJob* jobs, const int32_t* tilemap) {
  // A pack is a (strided) transpose: 32x32 tiles through LDS, read along whichever of (n, k) is the faster source axis,
  // written along k (the destination's contiguous axis).  The element-per-thread version read the Keras [k,k,Cin,Cout]
  // variable with a stride of Cout floats per lane for one of the two operand layouts (0.8 TB/s on Arch A's 31 M parameters).
  const UssegPackJob j = jobs[FLAT ? tilemap[3 * blockIdx.x] : blockIdx.y];
  const int tilesN = (j.Nn + 31) >> 5, tilesK = (j.Kk + 31) >> 5;
  const int id_begin = FLAT ? tilemap[3 * blockIdx.x + 1] : (int)blockIdx.x;
  const int ntile = FLAT ? id_begin + tilemap[3 * blockIdx.x + 2] : j.T * tilesN * tilesK;
  const int id_step = FLAT ? 1 : (int)gridDim.x;
  if (id_begin >= ntile) return;          // (grid form: sized for the largest job)
  __shared__ float tile[32][33];
  bf16_t* const dst = reinterpret_cast<bf16_t*>(j.dst);
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const bool k_fast = j.sK <= j.sN;
  for (int id = id_begin; id < ntile; id += id_step) {
    const int t = id / (tilesN * tilesK), r = id - t * (tilesN * tilesK);
    const int n0 = (r / tilesK) << 5, k0 = (r % tilesK) << 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nn = k_fast ? ly + 8 * i : lx, kk = k_fast ? lx : ly + 8 * i;
      float v = 0.f;
      if (n0 + nn < j.Nn && k0 + kk < j.Kk) {
        v = j.src[(int64_t)t * j.sT + (int64_t)(n0 + nn) * j.sN + (int64_t)(k0 + kk) * j.sK];
        if (j.nscale) v *= j.nscale[n0 + nn];      // folded inference BatchNorm (forward operand only)
      }
      tile[nn][kk] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nn = ly + 8 * i;
      if (n0 + nn < j.Nn && k0 + lx < j.Kk)
        dst[(int64_t)(j.n_off + n0 + nn) * j.Kw + (int64_t)t * j.tap_stride + j.k_off + k0 + lx] = f2bf(tile[nn][lx]);
    }
    __syncthreads();
  }
}
extern "C" int usseg_pack_weights_batched(const UssegPackJob* jobs_dev, int32_t njobs, usseg_stream_t stream) {
  USSEG_CHECK_ARG(jobs_dev && njobs > 0 && njobs < 65536, "pack_weights_batched: bad args");
  // 512 x njobs workgroups; a workgroup past its job's last 32x32 tile exits at once, the others stride over the tiles
  hipLaunchKernelGGL(pack_weights_batched_kernel<false>, dim3(512, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, nullptr);
  return usseg_check_launch("pack_weights_batched");
}
extern "C" int usseg_pack_weights_flat(const UssegPackJob* jobs_dev, const int32_t* tilemap_dev, int32_t nblocks, usseg_stream_t stream) {
  USSEG_CHECK_ARG(jobs_dev && tilemap_dev && nblocks > 0, "pack_weights_flat: bad args");
  hipLaunchKernelGGL(pack_weights_batched_kernel<true>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, tilemap_dev);
  return usseg_check_launch("pack_weights_flat");
}

// ---- the stride-2 3x3 head as a 2x2-tap convolution producing its four output parities as 16 channels ("quad" form):
// out[2i+a, 2j+b, n] = sum over (di,dj) in {0,-1}^2 of x[i+di, j+dj, :] . W[a-2di][b-2dj][n][:]   (terms with a tap index > 2 vanish)
// The conv kernels run it as an ordinary 3x3 conv with Cout = 16 = (parity class)*4 + n; these three helpers move the
// bias and the gradients between the Keras variables (kernel [3,3,Cout,Cin], bias [Cout]) and that form.
__global__ void quad_bias_expand_kernel(const float* bias, int C, int Np, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * Np; i += gridDim.x * blockDim.x) out[i] = (i % Np) < C ? bias[i % Np] : 0.f;
}
__global__ void quad_bias_fold_kernel(const float* d16, int C, float* dbias) {
  int n = threadIdx.x;
  if (n < C) dbias[n] += d16[n] + d16[4 + n] + d16[8 + n] + d16[12 + n];
}
// grad[kh][kw][n][c] += dq[tap(kh,kw)][c][class(kh,kw)*4 + n],  dq = [9][Cin_phys][16] (the 3x3-conv weight gradient of the quad form)
__global__ __launch_bounds__(256) void tconv_quad_unpack_kernel(const float* dq, int Cin_phys, int Cin, int Cout, float* grad, int k, int Np) {
  const int total = k * k * Cout * Cin, pad = k == 4 ? 1 : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int c = i % Cin, r = i / Cin;
    int n = r % Cout, kk = r / Cout;
    int kh = kk / k, kw = kk - k * kh;
    // output parity a = (kh+pad)&1 and source offset di = (a+pad-kh)/2 of kernel tap kh (k=3: pad 0; k=4: pad 1)
    int a = (kh + pad) & 1, b = (kw + pad) & 1;
    int di = (a + pad - kh) / 2, dj = (b + pad - kw) / 2;
    int t = (di + 1) * 3 + (dj + 1);
    grad[i] += dq[((int64_t)t * Cin_phys + c) * (4 * Np) + (a * 2 + b) * Np + n];
  }
}
extern "C" int usseg_quad_bias_expand(const float* bias, int32_t C, int32_t Np, float* out, usseg_stream_t stream) {
  USSEG_CHECK_ARG(bias && out && C >= 1 && Np >= C && Np % 4 == 0, "quad_bias_expand: bad args");
  hipLaunchKernelGGL(quad_bias_expand_kernel, dim3((4 * Np + 255) / 256), dim3(256), 0, (hipStream_t)stream, bias, C, Np, out);
  return usseg_check_launch("quad_bias_expand");
}
extern "C" int usseg_quad_bias_fold(const float* d16, int32_t C, float* dbias, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d16 && dbias && C >= 1 && C <= 4, "quad_bias_fold: bad args");
  hipLaunchKernelGGL(quad_bias_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d16, C, dbias);
  return usseg_check_launch("quad_bias_fold");
}
// the head's two folds in one launch: kernel gradient (every workgroup) + bias gradient (workgroup 0)
__global__ __launch_bounds__(256) void quad_head_fold_kernel(const float* dq, int Cin_phys, int Cin, int Cout, float* grad, int k, int Np, const float* d16,
                                                             float* dbias) {
  const int total = k * k * Cout * Cin, pad = k == 4 ? 1 : 0;
  if (blockIdx.x == 0 && threadIdx.x < Cout) {
    const int n = threadIdx.x;
    dbias[n] += d16[n] + d16[Np + n] + d16[2 * Np + n] + d16[3 * Np + n];
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int c = i % Cin, r = i / Cin;
    int n = r % Cout, kk = r / Cout;
    int kh = kk / k, kw = kk - k * kh;
    int a = (kh + pad) & 1, b = (kw + pad) & 1;
    int di = (a + pad - kh) / 2, dj = (b + pad - kw) / 2;
    int t = (di + 1) * 3 + (dj + 1);
    grad[i] += dq[((int64_t)t * Cin_phys + c) * (4 * Np) + (a * 2 + b) * Np + n];
  }
}
extern "C" int usseg_quad_head_fold(const float* dq, int32_t Cin_phys, int32_t Cin, int32_t Cout, int32_t Np, int32_t ksize, float* grad,
                                    const float* d16, float* dbias, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dq && grad && d16 && dbias && Cin >= 1 && Cin <= Cin_phys && Cout >= 1 && Cout <= Np && Cout <= 256 && (ksize == 3 || ksize == 4),
                  "quad_head_fold: bad args");
  int g = (ksize * ksize * Cout * Cin + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(quad_head_fold_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dq, Cin_phys, Cin, Cout, grad, ksize, Np, d16, dbias);
  return usseg_check_launch("quad_head_fold");
}
extern "C" int usseg_tconv_quad_unpack(const float* dq, int32_t Cin_phys, int32_t Cin, int32_t Cout, int32_t Np, int32_t ksize, float* grad,
                                       usseg_stream_t stream) {
  USSEG_CHECK_ARG(dq && grad && Cin >= 1 && Cin <= Cin_phys && Cout >= 1 && Cout <= Np && (ksize == 3 || ksize == 4), "tconv_quad_unpack: bad args");
  int g = (ksize * ksize * Cout * Cin + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(tconv_quad_unpack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dq, Cin_phys, Cin, Cout, grad, ksize, Np);
  return usseg_check_launch("tconv_quad_unpack");
}

// ---- folded inference BatchNorm: scale = gamma*rsqrt(var+eps), shift = beta - mean*scale + scale*conv_bias, all layers in ONE launch
__global__ __launch_bounds__(256) void bn_fold_batched_kernel(const UssegBnFoldJob* jobs) {
  const UssegBnFoldJob j = jobs[blockIdx.x];
  for (int c = threadIdx.x; c < j.Cp; c += 256) {
    float sc = 0.f, sh = 0.f;
    if (c < j.C) {
      sc = j.gamma[c] * rsqrtf(j.var[c] + j.eps);
      sh = j.beta[c] - j.mean[c] * sc + (j.bias ? sc * j.bias[c] : 0.f);
    }
    j.scale[c] = sc;
    j.shift[c] = sh;
  }
}
extern "C" int usseg_bn_fold_batched(const UssegBnFoldJob* jobs_dev, int32_t njobs, usseg_stream_t stream) {
  USSEG_CHECK_ARG(jobs_dev && njobs > 0 && njobs < 65536, "bn_fold_batched: bad args");
  hipLaunchKernelGGL(bn_fold_batched_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return usseg_check_launch("bn_fold_batched");
}

// space-to-depth / depth-to-space (2x2): full[b, 2i+a, 2j+b', n] <-> quad[b, i, j, (2a+b')*Np + n], 16 bytes per thread
template <bool TO_QUAD>
__global__ __launch_bounds__(256) void s2d_kernel(const bf16_t* src, int B, int H, int W, int CH, int ld_full, int Np, bf16_t* dst, int ld_quad) {
  const int64_t total = (int64_t)B * 4 * H * W * CH;           // (b, y, x of the full-resolution map, 8-channel chunk)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(i % CH) * 8;
    int64_t r = i / CH;
    const int x = (int)(r % (2 * W)); r /= 2 * W;
    const int y = (int)(r % (2 * H));
    const int b = (int)(r / (2 * H));
    const int64_t full = (((int64_t)b * 2 * H + y) * 2 * W + x) * ld_full + c0;
    const int64_t quad = (((int64_t)b * H + (y >> 1)) * W + (x >> 1)) * ld_quad + ((y & 1) * 2 + (x & 1)) * Np + c0;
    if (TO_QUAD) *reinterpret_cast<uint4*>(dst + quad) = *reinterpret_cast<const uint4*>(src + full);
    else *reinterpret_cast<uint4*>(dst + full) = *reinterpret_cast<const uint4*>(src + quad);
  }
}
extern "C" int usseg_space_to_depth2(const void* full, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ld_full, void* quad, int32_t Np,
                                     int32_t ld_quad, int32_t to_quad, usseg_stream_t stream) {
  USSEG_CHECK_ARG(full && quad && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && Np >= C && Np % 8 == 0 && ld_full % 8 == 0 && ld_full >= C &&
                      ld_quad % 8 == 0 && ld_quad >= 4 * Np, "space_to_depth2: bad args");
  const int64_t total = (int64_t)B * 4 * H * W * (C / 8);
  dim3 grid(grid_for(total, 256 * 4, 4096));
  if (to_quad)
    hipLaunchKernelGGL(s2d_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)full, B, H, W, C / 8, ld_full, Np, (bf16_t*)quad, ld_quad);
  else
    hipLaunchKernelGGL(s2d_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)quad, B, H, W, C / 8, ld_full, Np, (bf16_t*)full, ld_quad);
  return usseg_check_launch("space_to_depth2");
}

// Quad-form transposed conv (stride 2, 'same', k = 3 or 4) as tap-masked 3x3 convs on space-to-depth tensors: sets the per-class
// stencil masks for one call of the conv entry point (kernels that do not honour the mask still compute the right result: the
// masked taps carry zero weights / unused gradient entries).
static void quad_masks(int k, bool reversed) {
  const int pad = k == 4 ? 1 : 0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      unsigned m = 0;
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw)
          if (((kh + pad) & 1) == a && ((kw + pad) & 1) == b) {
            int t = ((a + pad - kh) / 2 + 1) * 3 + ((b + pad - kw) / 2 + 1);
            m |= 1u << (reversed ? 8 - t : t);
          }
      usseg_tap_mask.mask[a * 2 + b] = (uint16_t)m;
    }
}
struct TapMaskGuard {
  ~TapMaskGuard() { usseg_tap_mask.group_ch = 0; }
};
extern "C" int usseg_tconv_quad_fwd(const UssegConvDesc* d, int32_t ksize, int32_t Np, const void* x, const void* wq_f, const float* bias_q,
                                    void* y4, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && (ksize == 3 || ksize == 4) && Np > 0 && d->Cout == 4 * Np && d->ksize == 3 && d->dilation == 1, "tconv_quad_fwd: bad args");
  TapMaskGuard guard;
  quad_masks(ksize, false);
  usseg_tap_mask.group_ch = Np;
  return usseg_conv2d_fwd(d, x, wq_f, bias_q, nullptr, 0, y4, stream);
}
extern "C" int usseg_tconv_quad_dgrad(const UssegConvDesc* d, int32_t ksize, int32_t Np, const void* dy4, const void* wq_d, void* dx,
                                      usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && (ksize == 3 || ksize == 4) && Np > 0 && d->Cout == 4 * Np && d->ksize == 3 && d->dilation == 1, "tconv_quad_dgrad: bad args");
  TapMaskGuard guard;
  quad_masks(ksize, true);      // the backward-data loop uses weight tap 8 - t at loop tap t
  usseg_tap_mask.group_ch = Np;
  return usseg_conv2d_dgrad(d, dy4, wq_d, nullptr, 0, dx, stream);
}
extern "C" int usseg_tconv_quad_wgrad(const UssegConvDesc* d, int32_t ksize, int32_t Np, const void* x, const void* dy4, float* dq, float* ws,
                                      int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && (ksize == 3 || ksize == 4) && Np > 0 && d->Cout == 4 * Np && d->ksize == 3 && d->dilation == 1, "tconv_quad_wgrad: bad args");
  TapMaskGuard guard;
  quad_masks(ksize, false);
  usseg_tap_mask.group_ch = Np;
  return usseg_conv2d_wgrad(d, x, dy4, dq, ws, ws_floats, stream);
}

// ---- dropout mask (tf.nn.dropout(x, rate), TBI_ResNest.py:216): mask[m][c] = keep ? 1/(1-rate) : 0, counter-based hash RNG
__device__ __forceinline__ uint32_t hash32(uint64_t v) {
  v ^= v >> 33; v *= 0xff51afd7ed558ccdULL; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL; v ^= v >> 33;
  return (uint32_t)v;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(bf16_t* mask, int64_t M, int CH, int ld, uint64_t seed, float rate,
                                                           const int32_t* step_dev) {
  // step_dev (optional): a device-resident step counter mixed into the seed, so that a captured HIP graph draws a fresh mask
  // at every replay (the host-side seed of a captured launch is frozen)
  if (step_dev) seed += (uint64_t)(uint32_t)(*step_dev) * 0x2545F4914F6CDD1DULL;
  const int64_t total = M * CH;
  const float keep_scale = 1.f / (1.f - rate);
  const uint32_t thr = (uint32_t)(rate * 4294967296.0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / CH;
    int c0 = (int)(i - m * CH) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = hash32(seed * 0x9e3779b97f4a7c15ULL + (uint64_t)(i * 8 + j)) >= thr ? keep_scale : 0.f;
    *reinterpret_cast<uint4*>(mask + m * ld + c0) = pack8(v);
  }
}
extern "C" int usseg_dropout_mask(void* mask, int64_t M, int32_t C, int32_t ld, uint64_t seed, float rate, usseg_stream_t stream) {
  USSEG_CHECK_ARG(mask && C % 8 == 0 && ld % 8 == 0 && rate >= 0.f && rate < 1.f, "dropout_mask: bad args");
  if (M <= 0) return USSEG_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(M * (C / 8), 256 * 2, 4096)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)mask, M,
                     C / 8, ld, seed, rate, (const int32_t*)nullptr);
  return usseg_check_launch("dropout_mask");
}
extern "C" int usseg_dropout_mask_step(void* mask, int64_t M, int32_t C, int32_t ld, uint64_t seed, const int32_t* step_dev, float rate,
                                       usseg_stream_t stream) {
  USSEG_CHECK_ARG(mask && step_dev && C % 8 == 0 && ld % 8 == 0 && rate >= 0.f && rate < 1.f, "dropout_mask_step: bad args");
  if (M <= 0) return USSEG_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(M * (C / 8), 256 * 2, 4096)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)mask, M,
                     C / 8, ld, seed, rate, step_dev);
  return usseg_check_launch("dropout_mask_step");
}

// ---- BatchNormalization TRAINING mode (Keras: batch mean / biased variance, moving statistics updated with momentum) -------
// forward : usseg_channel_stats -> usseg_bn_finalize_stats -> usseg_norm_act_fwd(mode 1, batch statistics)
// backward: usseg_norm_act_bwd(mode 1) gives dx0 = dyh*gamma*rstd and this launch's sums tg = sum dyh*xhat, tb = sum dyh;
//           usseg_bn_train_bwd_fix subtracts the batch-statistics terms: dx = dx0 - gamma*rstd*(tb + xhat*tg)/M.
__global__ void bn_finalize_stats_kernel(const float* sum, const float* sumsq, float inv_m, int C, float momentum, float* mean, float* var,
                                         float* moving_mean, float* moving_var) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu = sum[c] * inv_m;
  float v = fmaxf(sumsq[c] * inv_m - mu * mu, 0.f);
  mean[c] = mu;
  var[c] = v;
  if (moving_mean) {
    moving_mean[c] = moving_mean[c] * momentum + mu * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + v * (1.f - momentum);
  }
}
extern "C" int usseg_bn_finalize_stats(const float* sum, const float* sumsq, int64_t M, int32_t C, float momentum, float* mean, float* var,
                                       float* moving_mean, float* moving_var, usseg_stream_t stream) {
  USSEG_CHECK_ARG(sum && sumsq && mean && var && M > 0 && C > 0 && (!moving_mean == !moving_var), "bn_finalize_stats: bad args");
  hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum, sumsq, 1.f / (float)M, C, momentum,
                     mean, var, moving_mean, moving_var);
  return usseg_check_launch("bn_finalize_stats");
}

__global__ __launch_bounds__(256) void bn_train_bwd_fix_kernel(const bf16_t* x, bf16_t* dx, int64_t M, int C, int CH, int ldx, int lddx, const float* gamma,
                                                                const float* mean, const float* var, float eps, const float* tg, const float* tb,
                                                                float inv_m) {
  const int64_t total = M * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t m = i / CH;
    int c0 = (int)(i - m * CH) * 8;
    float xv[8], dv[8];
    unpack8(*reinterpret_cast<const uint4*>(x + m * ldx + c0), xv);
    unpack8(*reinterpret_cast<const uint4*>(dx + m * lddx + c0), dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = c0 + j;
      if (c < C) {
        float rs = rsqrtf(var[c] + eps);
        float xh = (xv[j] - mean[c]) * rs;
        dv[j] -= gamma[c] * rs * (tb[c] + xh * tg[c]) * inv_m;
      }
    }
    *reinterpret_cast<uint4*>(dx + m * lddx + c0) = pack8(dv);
  }
}
extern "C" int usseg_bn_train_bwd_fix(const void* x, void* dx, int64_t M, int32_t C, int32_t Cphys, int32_t ldx, int32_t lddx, const float* gamma,
                                      const float* mean, const float* var, float eps, const float* tg, const float* tb, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && dx && gamma && mean && var && tg && tb && Cphys % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && M > 0, "bn_train_bwd_fix: bad args");
  hipLaunchKernelGGL(bn_train_bwd_fix_kernel, dim3(grid_for(M * (Cphys / 8), 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)dx, M, C, Cphys / 8, ldx, lddx, gamma, mean, var, eps, tg, tb, 1.f / (float)M);
  return usseg_check_launch("bn_train_bwd_fix");
}
