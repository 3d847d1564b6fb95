// Internal helpers shared by the gfx950 kernels of libusseg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/usseg.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;  // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// Exact floor(n / d) for 0 <= n < 2^20, d >= 1, with r = fdiv_rcp(d): cvt + fma + cvt (+ a multiply-subtract for the remainder) instead
// of the ~20-instruction sequence (three of them quarter-rate 32-bit multiplies) the compiler emits for a runtime integer division.
// The tile / patch decodes of the kernel prologues are chains of 6-7 such divisions per staged item: ~1700 instructions = 3 us of
// pure VALU time in front of the first load of every conv_big workgroup before this.  ((n + 0.5) / d is at least 0.5 / d away from any
// integer, more than the error of the 1-ulp reciprocal and the one rounded multiply for n < 2^20.)
// Written as cvt(2n + 1) * (0.5 / d) with the multiply in inline assembly: the SLP vectoriser would otherwise turn two neighbouring
// decodes into a v_pk_mul_f32 with a broadcast operand - the one packed-fp32 producer form this toolchain leaves unpadded in front of a
// dependent instruction (Makefile: NOPK; the build's isa_check found exactly that pair in the first version of this helper).
__device__ __forceinline__ float fdiv_rcp(int d) { return 0.5f * __builtin_amdgcn_rcpf((float)d); }   // v_rcp_f32: 1 ulp
__device__ __forceinline__ int fdiv(int n, float r) {
  float x;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(x) : "v"((float)(2 * n + 1)), "v"(r));
  return (int)x;
}
__device__ __forceinline__ void fdivmod(int n, int d, float r, int& q, int& rem) {
  q = fdiv(n, r);
  rem = n - __mul24(q, d);      // (n < 2^20: both factors are far inside 24 bits)
}

// Pixel-index variant: exact for 0 <= n < 2^23 and any d >= 1 with r1 = fdiv_rcp1(d) - the float quotient is within one of the true one
// (n / d * 1.5 * 2^-23 < 1 for d >= 3, exact reciprocals for d = 1, 2) and is corrected by one step either way: ~9 instructions.
__device__ __forceinline__ float fdiv_rcp1(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ void fdivmod_px(int n, int d, float r1, int& q, int& rem) {
  float x;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(x) : "v"((float)n), "v"(r1));
  q = (int)x;
  rem = n - __mul24(q, d);
  if (rem < 0) { --q; rem += d; }
  if (rem >= d) { ++q; rem -= d; }
}

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  switch (act) {
    case USSEG_ACT_LRELU: return v >= 0.f ? v : alpha * v;
    case USSEG_ACT_RELU: return v > 0.f ? v : 0.f;
    case USSEG_ACT_ELU: return v > 0.f ? v : alpha * (__expf(v) - 1.f);
    case USSEG_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // exact (erf) GELU, VisionTransformer.py:71
    default: return v;
  }
}
// derivative of act at pre-activation value v
__device__ __forceinline__ float act_grad(float v, int act, float alpha) {
  switch (act) {
    case USSEG_ACT_LRELU: return v >= 0.f ? 1.f : alpha;
    case USSEG_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case USSEG_ACT_ELU: return v > 0.f ? 1.f : alpha * __expf(v);
    case USSEG_ACT_GELU: return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
    default: return 1.f;
  }
}

// ---- bitwise reproducible grid-wide sum (no float atomics: their arrival order changes the rounding from run to run) ----------------
// acc[0] receives the result (overwritten), acc[1] is a ticket counter (zero before and after the call), acc[2 + b] holds workgroup b's
// total.  Every workgroup (256 threads) publishes its total; the LAST one to arrive adds all of them in workgroup order with a
// fixed tree.  No __threadfence(): an agent-scope release writes back the XCD's dirty L2 lines, which costs a kernel that also
// streams tens of MB of stores (the fused softmax + loss kernel went from 36 to 74 us with it).  Ordering comes from data flow
// instead: the partial is published with a RETURNING agent-scope atomic exchange (executed at the memory side), the ticket
// increment consumes that return value, and the last workgroup reads the partials with agent-scope atomic loads.
// All three atomics are RELAXED: under the HIP / LLVM memory model alone this is a data race.  What orders it is the ISA sequence
// (checked on every build by tools/check_ordered_sum.py, csrc/Makefile isa_check): a returning `global_atomic_swap sc0` executes at the
// memory side and only then returns; the wave waits for that value (`s_waitcnt vmcnt(0)`) before it issues the ticket's
// `global_atomic_add`; the last workgroup reads the slots with `global_load_dword sc1` (agent scope: not served from a non-coherent
// cache level).  A release on the ticket / acquire in the last workgroup would be the portable form; an agent-scope release writes the
// XCD's L2 back, which is the cost this construction avoids.
/* USSEG_ACC_FLOATS (usseg.h) = 2 + the largest grid of the kernels that use it */
__device__ __forceinline__ bool grid_ordered_sum(float block_total /* thread 0 */, float* acc, int nblocks, int bid = -1) {
  __shared__ int s_last;
  __shared__ float s_w[4];
  if (bid < 0) bid = blockIdx.x;        // (a 2-D grid passes its linear workgroup index)
  if (threadIdx.x == 0) {
    const float old = __hip_atomic_exchange(acc + 2 + bid, block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned zero = 0u;
    asm volatile("; ticket after the partial is at the memory side" : "+v"(zero) : "v"(old));
    const unsigned t = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(acc + 1), 1u + zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned)nblocks - 1u);
  }
  __syncthreads();
  if (!s_last) return false;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += __hip_atomic_load(acc + 2 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    acc[0] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);     // OVERWRITES: no zero-fill launch in front of the kernel
    __hip_atomic_store(reinterpret_cast<unsigned*>(acc + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}
// (returns true in the workgroup that performed the final addition: for work that must follow the sum)

// ---- shared epilogue of the conv kernels (16x16 MFMA layout: a lane holds Y[pixel = lane&15][n = nbase + 16*bt + j]) ----------------
// vmcnt retires in issue order and counts stores, so a load issued after a store waits for that store's write latency
// (about 1k cycles): the per-tile "load bias, add, store" loop these kernels started with serialized 2*NT store latencies
// per workgroup.  Every load of the epilogue (scale, bias, residual, old output) is therefore issued BEFORE its first store.
struct EpiArgs {
  const float* scale;
  const float* bias;
  const bf16_t* res;
  void* y;
  int32_t ldy, ldr, Nout, act;
  float alpha;
  int32_t out_f32, accumulate;
};
template <int NT>
struct EpiConst {
  float4 bb[NT];   // this lane's bias values (zero without a bias)
};
template <int NT>
__device__ __forceinline__ void epi_const_load(EpiConst<NT>& c, const float* bias, int nbase, int Nout) {
#pragma unroll
  for (int bt = 0; bt < NT; ++bt) {
    const int n = nbase + bt * 16;
    c.bb[bt] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) c.bb[bt] = *reinterpret_cast<const float4*>(bias + (n < Nout ? n : 0));   // clamped: unconditional per lane
  }
}
// The bias comes preloaded (EpiConst).  The rarer operands (scale of a folded BatchNorm, residual, old output when accumulating)
// are loaded per pixel strip `a` for all its channel tiles before that strip's first store: at most NA exposed store latencies.
// Straight-line epilogue body for bf16 outputs, specialised at compile time on (residual, accumulate): the loads of a pixel
// strip precede its stores and there is no untaken load path, so the compiler's vmcnt bookkeeping is exact.  (With the
// optional operands behind runtime branches it put a vmcnt(0) in front of every store: 16 serialized write latencies, half
// the lifetime of a short-K workgroup.)
template <int NA, int NT, bool RES, bool ACC, typename ActF>
__device__ __forceinline__ void epi_fast(const EpiArgs& e, const EpiConst<NT>& c, const f32x4_t (&acc)[NA][NT], const int64_t (&opix)[NA],
                                         const bool (&ovalid)[NA], int nbase, int64_t ybatch, ActF actf) {
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (!ovalid[a]) continue;
    bf16_t* const yrow = reinterpret_cast<bf16_t*>(e.y) + ybatch + opix[a] * e.ldy;
    uint2 rr[NT], oo[NT];
    if constexpr (RES) {
#pragma unroll
      for (int bt = 0; bt < NT; ++bt) {
        const int n = nbase + bt * 16;
        rr[bt] = *reinterpret_cast<const uint2*>(e.res + opix[a] * e.ldr + (n < e.Nout ? n : 0));
      }
    }
    if constexpr (ACC) {
#pragma unroll
      for (int bt = 0; bt < NT; ++bt) {
        const int n = nbase + bt * 16;
        oo[bt] = *reinterpret_cast<const uint2*>(yrow + (n < e.Nout ? n : 0));
      }
    }
#pragma unroll
    for (int bt = 0; bt < NT; ++bt) {
      const int n = nbase + bt * 16;
      if (n >= e.Nout) continue;
      float v0 = actf(acc[a][bt][0] + c.bb[bt].x), v1 = actf(acc[a][bt][1] + c.bb[bt].y);
      float v2 = actf(acc[a][bt][2] + c.bb[bt].z), v3 = actf(acc[a][bt][3] + c.bb[bt].w);
      if constexpr (RES) {
        v0 += __uint_as_float(rr[bt].x << 16); v1 += __uint_as_float(rr[bt].x & 0xffff0000u);
        v2 += __uint_as_float(rr[bt].y << 16); v3 += __uint_as_float(rr[bt].y & 0xffff0000u);
      }
      if constexpr (ACC) {
        v0 += __uint_as_float(oo[bt].x << 16); v1 += __uint_as_float(oo[bt].x & 0xffff0000u);
        v2 += __uint_as_float(oo[bt].y << 16); v3 += __uint_as_float(oo[bt].y & 0xffff0000u);
      }
      uint2 o;
      o.x = pack2bf(v0, v1);
      o.y = pack2bf(v2, v3);
      *reinterpret_cast<uint2*>(yrow + n) = o;
    }
  }
}

// Plain bf16 epilogue through LDS: the MFMA layout gives a lane 4 channels (8 B) of one pixel, so a direct store touches
// 16 pixel rows x 32 B per instruction - sixteen partial cache lines, and the write path serialises on them (measured ~450
// cycles per store instruction on a 128-channel output).  Here each wave transposes its [16*NA pixels][16*NT channels] tile
// through a private LDS slice and stores 16 B per lane with a pixel's channels in consecutive lanes: whole 128-byte lines,
// half the instructions.  `wl`: this wave's slice, epi_lds_bytes<NA,NT>() bytes, free of other users.
template <int NA, int NT>
__host__ __device__ constexpr int epi_lds_bytes() { return 16 * NA * (16 * NT * 2 + 16) + 16 * NA * 4; }
template <int NA, int NT, typename ActF>
__device__ __forceinline__ void epi_lds_body(char* wl, const EpiArgs& e, const EpiConst<NT>& c, const f32x4_t (&acc)[NA][NT], const int64_t (&opix)[NA],
                                             const bool (&ovalid)[NA], int n0, int lane, int64_t ybatch, ActF actf) {
  constexpr int BN = 16 * NT, RB = BN * 2 + 16, NP = 16 * NA, CH = BN / 8, PPI = 64 / CH;
  int* const pixtab = reinterpret_cast<int*>(wl + NP * RB);
  const int frow = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
#pragma unroll
    for (int bt = 0; bt < NT; ++bt) {
      uint2 o;
      o.x = pack2bf(actf(acc[a][bt][0] + c.bb[bt].x), actf(acc[a][bt][1] + c.bb[bt].y));
      o.y = pack2bf(actf(acc[a][bt][2] + c.bb[bt].z), actf(acc[a][bt][3] + c.bb[bt].w));
      *reinterpret_cast<uint2*>(wl + (a * 16 + frow) * RB + (bt * 16 + 4 * g) * 2) = o;
    }
    if (g == 0) pixtab[a * 16 + frow] = ovalid[a] ? (int)opix[a] : -1;
  }
  // same wave, in-order LDS queue: the reads below see the writes above
  const int ch = lane % CH, pl = lane / CH;
  const int n = n0 + ch * 8;
  bf16_t* const ybase = reinterpret_cast<bf16_t*>(e.y) + ybatch + n;
#pragma unroll
  for (int i = 0; i < NP / PPI; ++i) {
    const int P = i * PPI + pl;
    const int pix = pixtab[P];
    const uint4 v = *reinterpret_cast<const uint4*>(wl + P * RB + ch * 16);
    if (pix >= 0 && n < e.Nout) *reinterpret_cast<uint4*>(ybase + (int64_t)pix * e.ldy) = v;
  }
}
// true if it took the tile (plain bf16 output, 16-byte aligned rows); false: the caller runs conv_epilogue
template <int NA, int NT>
__device__ __forceinline__ bool conv_epilogue_lds(char* wl, const EpiArgs& e, const EpiConst<NT>& c, const f32x4_t (&acc)[NA][NT],
                                                  const int64_t (&opix)[NA], const bool (&ovalid)[NA], int n0, int lane, int64_t ybatch = 0) {
  if (e.scale || e.out_f32 || e.res || e.accumulate || (e.ldy & 7) || (ybatch & 7)) return false;
  const float alpha = e.alpha;
  const int act = e.act;
  if (act == USSEG_ACT_NONE) epi_lds_body<NA, NT>(wl, e, c, acc, opix, ovalid, n0, lane, ybatch, [](float v) { return v; });
  else if (act == USSEG_ACT_LRELU) epi_lds_body<NA, NT>(wl, e, c, acc, opix, ovalid, n0, lane, ybatch, [alpha](float v) { return v >= 0.f ? v : alpha * v; });
  else if (act == USSEG_ACT_RELU) epi_lds_body<NA, NT>(wl, e, c, acc, opix, ovalid, n0, lane, ybatch, [](float v) { return v > 0.f ? v : 0.f; });
  else epi_lds_body<NA, NT>(wl, e, c, acc, opix, ovalid, n0, lane, ybatch, [act, alpha](float v) { return apply_act(v, act, alpha); });
  return true;
}

// SIMPLE: the caller guarantees no scale / residual / accumulate / fp32 output (the streaming kernel's per-step epilogue must
// not contain even untaken load paths: the compiler's waits for them would drain the prefetch DMA).
template <int NA, int NT, bool SIMPLE = false>
__device__ __forceinline__ void conv_epilogue(const EpiArgs& e, const EpiConst<NT>& c, const f32x4_t (&acc)[NA][NT], const int64_t (&opix)[NA],
                                              const bool (&ovalid)[NA], int nbase, int64_t ybatch = 0) {
  const float alpha = e.alpha;
  const int act = e.act;
  auto a_none = [](float v) { return v; };
  auto a_lrelu = [alpha](float v) { return v >= 0.f ? v : alpha * v; };
  auto a_relu = [](float v) { return v > 0.f ? v : 0.f; };
  auto a_any = [act, alpha](float v) { return apply_act(v, act, alpha); };
  const bool plain = SIMPLE || (!e.scale && !e.out_f32 && !e.res && !e.accumulate);
  if (plain) {   // one uniform branch on the activation for the whole tile (apply_act's per-element switch costs ~10 scalar branches each)
    if (act == USSEG_ACT_NONE) epi_fast<NA, NT, false, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_none);
    else if (act == USSEG_ACT_LRELU) epi_fast<NA, NT, false, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_lrelu);
    else if (act == USSEG_ACT_RELU) epi_fast<NA, NT, false, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_relu);
    else epi_fast<NA, NT, false, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_any);
    return;
  }
  if constexpr (!SIMPLE) {
    if (!e.scale && !e.out_f32 && e.ldr % 4 == 0) {
      const bool ac = e.accumulate != 0;
      if (e.res && !ac) {
        if (act == USSEG_ACT_NONE) epi_fast<NA, NT, true, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_none);
        else if (act == USSEG_ACT_LRELU) epi_fast<NA, NT, true, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_lrelu);
        else epi_fast<NA, NT, true, false>(e, c, acc, opix, ovalid, nbase, ybatch, a_any);
        return;
      }
      if (!e.res && ac) {
        if (act == USSEG_ACT_NONE) epi_fast<NA, NT, false, true>(e, c, acc, opix, ovalid, nbase, ybatch, a_none);
        else epi_fast<NA, NT, false, true>(e, c, acc, opix, ovalid, nbase, ybatch, a_any);
        return;
      }
    }
  }
  const bool acc_bf = e.accumulate && !e.out_f32;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (!ovalid[a]) continue;
    uint2 rr[NT], oo[NT];
    float4 sc[NT];
    if (e.res) {
#pragma unroll
      for (int bt = 0; bt < NT; ++bt) {
        const int n = nbase + bt * 16;
        rr[bt] = make_uint2(0, 0);
        if (n < e.Nout) rr[bt] = *reinterpret_cast<const uint2*>(e.res + opix[a] * e.ldr + n);
      }
    }
    if (acc_bf) {
#pragma unroll
      for (int bt = 0; bt < NT; ++bt) {
        const int n = nbase + bt * 16;
        oo[bt] = make_uint2(0, 0);
        if (n < e.Nout) oo[bt] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(e.y) + ybatch + opix[a] * e.ldy + n);
      }
    }
    if (e.scale) {
#pragma unroll
      for (int bt = 0; bt < NT; ++bt) {
        const int n = nbase + bt * 16;
        sc[bt] = *reinterpret_cast<const float4*>(e.scale + (n < e.Nout ? n : 0));
      }
    }
#pragma unroll
    for (int bt = 0; bt < NT; ++bt) {
      const int n = nbase + bt * 16;
      if (n >= e.Nout) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][bt][j];
      if (e.scale) { v[0] *= sc[bt].x; v[1] *= sc[bt].y; v[2] *= sc[bt].z; v[3] *= sc[bt].w; }
      v[0] += c.bb[bt].x; v[1] += c.bb[bt].y; v[2] += c.bb[bt].z; v[3] += c.bb[bt].w;
      if (e.act != USSEG_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], e.act, e.alpha);
      }
      if (e.res) {
        v[0] += __uint_as_float(rr[bt].x << 16); v[1] += __uint_as_float(rr[bt].x & 0xffff0000u);
        v[2] += __uint_as_float(rr[bt].y << 16); v[3] += __uint_as_float(rr[bt].y & 0xffff0000u);
      }
      if (e.out_f32) {   // the <=4-class heads only: tiny outputs, element-wise tail handling
        float* dst = reinterpret_cast<float*>(e.y) + ybatch + opix[a] * e.ldy + n;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < e.Nout) dst[j] = e.accumulate ? dst[j] + v[j] : v[j];
      } else {
        if (acc_bf) {
          v[0] += __uint_as_float(oo[bt].x << 16); v[1] += __uint_as_float(oo[bt].x & 0xffff0000u);
          v[2] += __uint_as_float(oo[bt].y << 16); v[3] += __uint_as_float(oo[bt].y & 0xffff0000u);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(e.y) + ybatch + opix[a] * e.ldy + n) = o;
      }
    }
  }
}

// destination map of a weight gradient (UssegWgradDst by value in the kernel parameters); nblocks == 0: identity.
// Internally a block carries two tap strides - tap t = 3*th + tw lands at th*sTr + tw*sT (plain maps: sTr = 3*sT, i.e. t*sT) -
// and the map a tap mask: the parity classes of a transposed conv are 3x3 weight gradients whose valid taps are a 2x2 (or
// smaller) subset sitting at stride 2 in the Keras [k,k,...] variable (wgrad_halo.hip, tconv form).
struct WgBlock {
  float* dst;
  int64_t sT, sTr, sI, sO;
  int32_t i_off, o_off, ni, no;
};
struct WgMap {
  int32_t nblocks;
  uint32_t tapmask;   // taps that exist in the destination (0x1ff for a plain map; identity maps ignore it)
  WgBlock blk[4];
};
__device__ __forceinline__ float* wg_map_dst(const WgMap& m, float* ident, int64_t ident_idx, int t, int mi, int n) {
  if (m.nblocks == 0) return ident + ident_idx;
  if (t < 9 && !((m.tapmask >> t) & 1u)) return nullptr;   // (per-tap kernels pass up to 16 taps with a full mask)
  const int th = t / 3, tw = t - 3 * th;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b < m.nblocks) {
      const WgBlock& k = m.blk[b];
      if ((unsigned)(mi - k.i_off) < (unsigned)k.ni && (unsigned)(n - k.o_off) < (unsigned)k.no)
        return k.dst + th * k.sTr + tw * k.sT + (mi - k.i_off) * k.sI + (n - k.o_off) * k.sO;
    }
  }
  return nullptr;
}
static inline int wg_map_fill(WgMap& m, const UssegWgradDst* d) {
  m = {};
  m.tapmask = 0xffffffffu;
  if (!d) return 1;
  if (d->nblocks < 1 || d->nblocks > 4) return 0;
  m.nblocks = d->nblocks;
  for (int b = 0; b < d->nblocks; ++b) {
    const UssegWgradBlock& u = d->blk[b];
    if (!u.dst) return 0;
    m.blk[b] = {u.dst, u.sT, 3 * u.sT, u.sI, u.sO, u.i_off, u.o_off, u.ni, u.no};
  }
  return 1;
}

// per-output-channel multiplier applied before the bias in the conv epilogues (folded inference BatchNorm): set by the
// *_affine / multi entry points for the duration of one call, read by the launchers (slot = job index)
extern thread_local const float* usseg_epi_scale[4];

// per-channel-class tap masks of the quad-form transposed conv (set by the usseg_tconv_quad_* entry points for one call):
// output channels (fwd, wgrad) or input channels (dgrad) [cls*group_ch, (cls+1)*group_ch) only use the stencil taps in mask[cls]
struct UssegTapMask {
  int32_t group_ch;      // 0 = no masking
  uint16_t mask[4];
};
extern thread_local UssegTapMask usseg_tap_mask;

void usseg_set_error(const char* fmt, ...);
#define USSEG_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      usseg_set_error(__VA_ARGS__);                \
      return USSEG_ERR_BAD_ARG;                    \
    }                                              \
  } while (0)
int usseg_check_launch(const char* what);
// opt-in per-launch timing: returns an event slot (>=0) after recording its start event, or -1 when disabled
// LayerNorm + LeakyReLU backward on the lane-per-pixel tile kernel (cardinal.hip); returns 0 if it has no instantiation for the geometry
struct LnTileArgs {
  const bf16_t *x, *dy;
  bf16_t* dx;
  const float *gamma, *beta;
  const float *sa_s, *sa_dg;     // split-attention re-weighting backward folded in: dy_eff = sa_mult * sa_s[b][c] * dy + sa_dg[b][c]; NULL = plain
  float sa_mult;
  int32_t sa_cy;
  float* ws;                     // per-workgroup partial rows [grid][3][Cphys]
  int64_t M, HW;
  int32_t C, Cphys, G, ldx, lddy, lddx;
  float eps, alpha;
};
int usseg_try_ln_bwd_tile(const LnTileArgs& a, float* dgamma, float* dbeta, float* dbias, float* caller_ws, hipStream_t s);
// two independent LayerNorm + LeakyReLU backward passes in one launch (cardinal.hip); g* = {dgamma, dbeta, dbias}; 0 if no instantiation
int usseg_try_ln_bwd_pair(const LnTileArgs& a, float* const* ga, const LnTileArgs& b, float* const* gb, float* caller_ws, hipStream_t s);
int usseg_prof_start(int kind, hipStream_t s);
void usseg_prof_stop(int kind, int slot, hipStream_t s);

#define USSEG_REDUCE_MAX_BLOCKS 1024
// adds the per-workgroup partial rows written by a reduction kernel to up to three destinations (defer.hip)
void usseg_launch_reduce_finish(const float* ws, int groups, int nb, int K, int Cp, int C, float scale, float* d0, float* d1, float* d2,
                                hipStream_t s, int overwrite = 0);

// 3x3 conv with an LDS halo tile (conv_halo.hip): returns 1 if it took the launch, 0 if the geometry does not fit
int usseg_try_launch_conv_halo(const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W, int d,
                               int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32,
                               int accumulate, int flip, hipStream_t s);

// 3x3 conv, 256-pixel tiles with double-buffered LDS-DMA stages (conv_big.hip): 1 if it took the launch, 0 otherwise
int usseg_try_launch_conv_big(const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W, int d,
                              int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32,
                              int accumulate, int flip, hipStream_t s);

int usseg_try_launch_conv_halo_multi(int njobs, const UssegConvJob* jobs, int flip, hipStream_t s);
int usseg_try_launch_conv_big_multi(int njobs, const UssegConvJob* jobs, int flip, hipStream_t s);

// 3x3 weight gradient with an LDS halo tile (wgrad_halo.hip): 1 if it took the launch, 0 otherwise
int usseg_try_launch_wgrad_halo(const bf16_t* x, const bf16_t* dy, float* out, const WgMap& map, int B, int H, int W, int d, int Ma, int Nb,
                                int ldx, int lddy, float* ws, int64_t ws_floats, hipStream_t s);

// deferral of the finishing reductions (defer.hip): workspace regions for producers between usseg_defer_begin and _flush
float* usseg_defer_reduce_ws(hipStream_t s, float* caller_ws, int64_t need);
float* usseg_defer_wgrad_ws(hipStream_t s, float* caller_ws, int64_t caller_floats, int64_t* avail);
// out (identity or mapped) += sum over the split slabs ws[split][slab_floats] (wgrad_halo.hip)
void usseg_launch_wgrad_finish(const float* ws, int splits, int64_t slab_floats, float* out, const WgMap& map, int Ma, int Nb, hipStream_t s);

int usseg_try_launch_wgrad_halo_multi(int njobs, const UssegWgradJob* jobs, float* ws, int64_t ws_floats, hipStream_t s);
// stride-2 transposed-conv weight gradient as four tap-masked parity-class jobs of the halo-tile kernel (wgrad_halo.hip): 1 if taken
int usseg_try_launch_tconv_wgrad_halo(const bf16_t* x, const bf16_t* dy, const WgMap& map, int B, int H, int W, int Cin, int Cout, int ldx,
                                      int lddy, int k, float* ws, int64_t ws_floats, hipStream_t s);

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }
