// The encoder stem as ONE launch (SURVEY.md section 2.2 K1 + K6 + K8; ResNest.py:39-47):
//
//   x [B,H,W,1 (8 physical)] -> conv1 3x3 (1 -> 16) + LeakyReLU -> convtmp_1 3x3 (16 -> 32) + inference BatchNorm (scale folded into the packed
//   operand, shift as the bias) + LeakyReLU -> convtmp_2 3x3 (32 -> 32) -> BatchNorm + LeakyReLU -> AveragePooling2D(2,2)
//
// A workgroup owns a 16x16-pixel tile of the convtmp_2 output (80 us per 16 images against 98 us for the four launches; bit-identical to them): it stages the x tile with a 3-pixel halo, recomputes conv1 on the 20x20 and
// convtmp_1 on the 18x18 pixels its 16x16 outputs need (halo recompute: +56 % / +27 % of two cheap convs instead of two round trips of
// 16- and 32-channel full-resolution tensors through HBM), and writes what the backward pass needs - y1, t1 (activated), the pre-norm
// convtmp_2 output - plus the pooled tensor, once, in whole rows.  HBM traffic: 1 read of x + 4 writes, where the four unfused launches
// move every intermediate twice.  Zero padding applies to each conv's INPUT, so tile pixels outside the image are forced to zero after
// each activation.  Values are rounded to bf16 exactly where the unfused launches round them.
// MFMA orientation as in the conv kernels: A = weight rows (output channel), B = pixels from the LDS tile; every K step's weight fragments
// of a conv sit in registers (<= 18), loaded straight from L2 one phase ahead.
#include "common.h"

namespace {

struct StemFwd {
  const bf16_t* x;
  const bf16_t *w1, *w2, *w3;
  const float *b1, *b2, *b3, *gamma, *beta, *mean, *var;
  bf16_t *y1, *t1, *c2, *pooled;
  int32_t B, H, W, ldx, tiles_x, ntiles;
  float alpha, eps;
};

#ifndef STEM_TH
#define STEM_TH 16
#endif
constexpr int T = 16, TH = STEM_TH;                    // output tile: T pixels wide, TH high.  (16x8 was measured: 113 us against 80 us for 16x16 -
                                                       // 168 VGPRs hold three workgroups per CU either way and the halo recompute grows)
constexpr int XW = T + 6, Y1W = T + 4, T1W = T + 2;    // grid widths of the staged x / y1 / t1 tiles
constexpr int XH = TH + 6, Y1H = TH + 4, T1H = TH + 2;
constexpr int XS = 8, Y1S = 16 + 8, T1S = 32 + 8, C2S = 32 + 8;   // LDS row strides (elements)
constexpr int XS_B = XW * XH * XS * 2, Y1_B = Y1W * Y1H * Y1S * 2, T1_B = T1W * T1H * T1S * 2;
constexpr int LDS_B = XS_B + Y1_B + T1_B;              // the convtmp_2 staging tile reuses the x + y1 region
static_assert(T * TH * C2S * 2 <= XS_B + Y1_B, "staging tile must fit the dead x / y1 region");
static_assert(XW * XH <= 512 && (T / 2) * (TH / 2) * 4 <= 256, "thread mappings");

__device__ __forceinline__ bf16x8_t zfrag() {
  bf16x8_t z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
  return z;
}

// weight fragments of a conv: K = 9 taps x CPT 8-channel chunks, NT 16-row channel tiles; rows of `w` are 9*CPT*8 elements
template <int CPT, int NT>
struct WFrag {
  static constexpr int NCH = 9 * CPT, KS = (NCH + 3) / 4;
  bf16x8_t a[KS][NT];
  __device__ __forceinline__ void load(const bf16_t* w, int r, int q) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int chunk = 4 * ks + q;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(w + (int64_t)(nt * 16 + r) * (NCH * 8) + (chunk < NCH ? chunk : 0) * 8);
        a[ks][nt] = chunk < NCH ? v : zfrag();
      }
  }
};

// one 3x3 conv from an LDS tile (grid IW = OW + 2 wide, row stride SRC_S) to an LDS tile (OW x OH pixels, row stride DST_S):
// out = act(conv + bias), forced to zero outside the image when ZERO_OUT (the next conv's zero padding); wave w takes pixel tiles w, w+4, ...
template <int CPT, int NT, int OW, int OH, int SRC_S, int DST_S, bool LRELU, bool ZERO_OUT>
__device__ __forceinline__ void conv_phase(const bf16_t* src, bf16_t* dst, const WFrag<CPT, NT>& wf, const float* bias, float alpha, int gy0, int gx0,
                                           int H, int W, int wv, int r, int q) {
  constexpr int IW = OW + 2, NPIX = OW * OH, MT = (NPIX + 15) / 16, NCH = 9 * CPT, KS = (NCH + 3) / 4;
  float4 bb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bb[nt] = *reinterpret_cast<const float4*>(bias + nt * 16 + 4 * q);
  // this lane's K-step offsets into the source tile (tap shift + channel chunk): once per phase, not per pixel tile
  int toff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int chunk = 4 * ks + q;
    const int cc = chunk < NCH ? chunk : 0;
    const int tap = cc / CPT, choff = cc - tap * CPT;
    const int dy = tap / 3, dx = tap - dy * 3;
    toff[ks] = (dy * IW + dx) * SRC_S + choff * 8;
  }
  for (int mt = wv; mt < MT; mt += 4) {
    const int pp = mt * 16 + r;
    const int pc = pp < NPIX ? pp : NPIX - 1;
    const int py = pc / OW, px = pc - py * OW;
    const int base = (py * IW + px) * SRC_S;
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // (a chunk past the last tap reads tap 0 again: its weight fragment is zero)
      const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(src + base + toff[ks]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.a[ks][nt], b, acc[nt], 0, 0, 0);
    }
    if (pp < NPIX) {
      const int gy = gy0 + py, gx = gx0 + px;
      const bool in = !ZERO_OUT || (gy >= 0 && gy < H && gx >= 0 && gx < W);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v0 = acc[nt][0] + bb[nt].x, v1 = acc[nt][1] + bb[nt].y, v2 = acc[nt][2] + bb[nt].z, v3 = acc[nt][3] + bb[nt].w;
        if (LRELU) {
          v0 = v0 >= 0.f ? v0 : alpha * v0; v1 = v1 >= 0.f ? v1 : alpha * v1;
          v2 = v2 >= 0.f ? v2 : alpha * v2; v3 = v3 >= 0.f ? v3 : alpha * v3;
        }
        uint2 o;
        o.x = in ? pack2bf(v0, v1) : 0u;
        o.y = in ? pack2bf(v2, v3) : 0u;
        *reinterpret_cast<uint2*>(dst + pp * DST_S + nt * 16 + 4 * q) = o;
      }
    }
  }
}

// interior (tile-owned) pixels of an LDS tile -> HBM rows; OFF = the tile's halo width, NCHK = 8-channel chunks per pixel
template <int GW, int OFF, int NCHK, int S>
__device__ __forceinline__ void store_interior(const bf16_t* tile, bf16_t* out, int64_t img, int ld, int ty0, int tx0, int H, int W, int tid) {
  for (int it = tid; it < T * TH * NCHK; it += 256) {
    const int pp = it / NCHK, c = it - pp * NCHK;
    const int iy = pp / T, ix = pp - iy * T;
    const int gy = ty0 + iy, gx = tx0 + ix;
    if (gy < H && gx < W)
      *reinterpret_cast<uint4*>(out + (img + (int64_t)gy * W + gx) * ld + c * 8) =
          *reinterpret_cast<const uint4*>(tile + ((iy + OFF) * GW + ix + OFF) * S + c * 8);
  }
}

__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemFwd p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  bf16_t* const XT = reinterpret_cast<bf16_t*>(lds);
  bf16_t* const Y1T = reinterpret_cast<bf16_t*>(lds + XS_B);
  bf16_t* const T1T = reinterpret_cast<bf16_t*>(lds + XS_B + Y1_B);
  bf16_t* const C2T = reinterpret_cast<bf16_t*>(lds);          // reuses the x + y1 region once convtmp_1 has read them
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  // (a persistent form that keeps all 31 weight fragments in registers across tiles was measured: 264 VGPRs = one workgroup per CU, 133 us
  // against 80 us - the tile is a chain of five barrier-separated phases and needs its three co-resident workgroups more than it needs
  // the 125 KB of fragment reloads per tile removed)
  WFrag<1, 1> wf1;
  WFrag<2, 2> wf2;
  wf1.load(p.w1, r, q);
  wf2.load(p.w2, r, q);
  {
  const int b = blockIdx.y, tile = blockIdx.x;
  const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * T;
  const int64_t img = (int64_t)b * p.H * p.W;
  // ---- x tile + 3-pixel halo (zeros outside the image): both loads of a thread before its LDS stores
  {
    uint4 v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hp = tid + 256 * i;
      const int hy = hp / XW, hx = hp - hy * XW;
      const int gy = ty0 - 3 + hy, gx = tx0 - 3 + hx;
      v[i] = (hp < XW * XH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                 ? *reinterpret_cast<const uint4*>(p.x + (img + (int64_t)gy * p.W + gx) * p.ldx) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hp = tid + 256 * i;
      if (hp < XW * XH) *reinterpret_cast<uint4*>(XT + hp * XS) = v[i];
    }
  }
  __syncthreads();
  // ---- conv1 + LeakyReLU on the 20x20 pixels convtmp_1 needs (ResNest.py:39-40)
  conv_phase<1, 1, Y1W, Y1H, XS, Y1S, true, true>(XT, Y1T, wf1, p.b1, p.alpha, ty0 - 2, tx0 - 2, p.H, p.W, wv, r, q);
  WFrag<4, 2> wf3;
  wf3.load(p.w3, r, q);                   // in flight during convtmp_1
  __syncthreads();
  store_interior<Y1W, 2, 2, Y1S>(Y1T, p.y1, img, 16, ty0, tx0, p.H, p.W, tid);
  // ---- convtmp_1 + folded BatchNorm + LeakyReLU on the 18x18 pixels convtmp_2 needs (:41-43)
  conv_phase<2, 2, T1W, T1H, Y1S, T1S, true, true>(Y1T, T1T, wf2, p.b2, p.alpha, ty0 - 1, tx0 - 1, p.H, p.W, wv, r, q);
  __syncthreads();
  store_interior<T1W, 1, 4, T1S>(T1T, p.t1, img, 32, ty0, tx0, p.H, p.W, tid);
  // ---- convtmp_2 (:44): the pre-norm output, kept for the BatchNorm backward
  conv_phase<4, 2, T, TH, T1S, C2S, false, false>(T1T, C2T, wf3, p.b3, 0.f, ty0, tx0, p.H, p.W, wv, r, q);
  __syncthreads();
  store_interior<T, 0, 4, C2S>(C2T, p.c2, img, 32, ty0, tx0, p.H, p.W, tid);
  // ---- BatchNorm + LeakyReLU + 2x2 average pool (:45-47): one (pooled pixel, 8-channel chunk) per thread; the activation is rounded to
  //      bf16 before the pool, as the two-launch form stores it
  if (tid < (T / 2) * (TH / 2) * 4) {
    const int pp = tid >> 2, c = tid & 3;
    const int oy = pp >> 3, ox = pp & 7;
    const int gy = (ty0 >> 1) + oy, gx = (tx0 >> 1) + ox;
    float ga[8], be[8], mu[8], rs[8];
    *reinterpret_cast<float4*>(ga) = *reinterpret_cast<const float4*>(p.gamma + c * 8);
    *reinterpret_cast<float4*>(ga + 4) = *reinterpret_cast<const float4*>(p.gamma + c * 8 + 4);
    *reinterpret_cast<float4*>(be) = *reinterpret_cast<const float4*>(p.beta + c * 8);
    *reinterpret_cast<float4*>(be + 4) = *reinterpret_cast<const float4*>(p.beta + c * 8 + 4);
    *reinterpret_cast<float4*>(mu) = *reinterpret_cast<const float4*>(p.mean + c * 8);
    *reinterpret_cast<float4*>(mu + 4) = *reinterpret_cast<const float4*>(p.mean + c * 8 + 4);
    *reinterpret_cast<float4*>(rs) = *reinterpret_cast<const float4*>(p.var + c * 8);
    *reinterpret_cast<float4*>(rs + 4) = *reinterpret_cast<const float4*>(p.var + c * 8 + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) rs[j] = rsqrtf(rs[j] + p.eps);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int iy = 2 * oy + (s >> 1), ix = 2 * ox + (s & 1);
      float xv[8], o[8], r8[8];
      unpack8(*reinterpret_cast<const uint4*>(C2T + (iy * T + ix) * C2S + c * 8), xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = ga[j] * ((xv[j] - mu[j]) * rs[j]) + be[j];
        o[j] = v >= 0.f ? v : p.alpha * v;
      }
      unpack8(pack8(o), r8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += r8[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= 0.25f;
    if (gy < (p.H >> 1) && gx < (p.W >> 1))
      *reinterpret_cast<uint4*>(p.pooled + ((int64_t)b * (p.H >> 1) * (p.W >> 1) + (int64_t)gy * (p.W >> 1) + gx) * 32 + c * 8) = pack8(acc);
  }
  }
#endif
}

// =====================================================================================================================================
// Backward-data pass of a stem conv + the backward of the BatchNorm / activation in front of it, in one launch.
//   dy [B,H,W,32] (gradient w.r.t. the conv's output) -> 3x3 backward-data (flipped taps, the packed operand of usseg_conv2d_dgrad) -> fp32
//   gradient w.r.t. the conv's input -> MODE 2: the folded inference BatchNorm + LeakyReLU backward from the ACTIVATED tensor the forward
//   stored (usseg_norm_act_bwd mode 2: dx, dgamma, dbeta, sum dx);  MODE 0: LeakyReLU backward + column sums (usseg_act_bwd_colsum).
// Unfused that was conv_stream (writes the 32- / 16-channel gradient at full resolution) -> norm / act kernel (reads it back with the stored
// activation, writes dx): 28 + 40 us and 20 + 21 us of the Arch B step; here a workgroup stages an 18x18-pixel tile of dy in LDS, keeps the
// nine taps' weight fragments in registers and a lane ends up with 4 consecutive channels of one pixel, finishes the elementwise backward
// in registers (the stored activation prefetched before the MFMAs), writes dx once and reduces its per-channel sums over the tile's pixels
// (in-lane over its pixels, a 16-lane shuffle ladder, four waves through LDS): one partial row per workgroup, ordered finish.
struct DgAct {
  const bf16_t *dy, *wd, *yact;
  bf16_t* dx;
  const float *gamma, *beta, *var;
  float* ws;
  int32_t B, H, W, lddy, ldy, lddx, tiles_x, ntiles;
  float eps, alpha;
};
[[maybe_unused]] constexpr int DG_S = 40;      // LDS row stride of the dy tile (elements): 80 B, the 16 pixel rows of a fragment read hit distinct banks

template <int CO, int MODE>
__global__ __launch_bounds__(256, 2) void dgrad_actbwd_kernel(const DgAct p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TW = 16, HWD = TW + 2, NT = CO / 16, NIT = (HWD * HWD * 4 + 255) / 256, CL = 4 * NT;
  __shared__ __attribute__((aligned(16))) bf16_t DT[2][HWD * HWD * DG_S];
  __shared__ float RED[4][4][NT * 4 * 3];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  // A lane ends up with the CL = 4 NT consecutive channels CL q .. CL q + CL - 1 of its pixels (one 8- / 16-byte load of the stored
  // activation, one store of dx): output row 4 q' + j' of fragment nt is channel CL q' + 4 nt + j', a permutation applied where the weight
  // rows are fetched.  The nine taps' fragments stay in registers: A[row][k = tap*32 + q*8 ..] of the packed backward-data operand, tap t of
  // the stencil (reading dy at offset (t/3, t%3) of the halo tile) meets the weights of tap 8 - t.
  // NT == 1: in registers.  NT == 2: 72 registers of fragments + the prefetched tile + the accumulators do not fit the 256 registers that two
  // waves per SIMD leave a lane (measured: 357 wanted, 67 spilled when capped), so the fragments live in LDS, lane-major (conflict-free
  // 16-byte reads, 2 per tap beside the 4 pixel reads), and so do the per-channel constants.
  constexpr bool WREG = NT == 1;
  __shared__ __attribute__((aligned(16))) bf16_t WL[WREG ? 8 : 9 * NT * 64 * 8];
  __shared__ __attribute__((aligned(16))) float CST[3][CO];
  auto wrow = [&](int t, int nt) { return p.wd + (int64_t)(CL * (r >> 2) + 4 * nt + (r & 3)) * 288 + (8 - t) * 32 + q * 8; };
  bf16x8_t a[WREG ? 9 : 1][NT];
  if constexpr (WREG) {
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t][0] = *reinterpret_cast<const bf16x8_t*>(wrow(t, 0));
  } else if (wv == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<bf16x8_t*>(WL + ((t * NT + nt) * 64 + lane) * 8) = *reinterpret_cast<const bf16x8_t*>(wrow(t, nt));
  }
  // per-channel constants: gamma * rstd, beta, 1 / gamma (MODE 2)
  if (MODE == 2 && tid < CO) {
    const float g = p.gamma[tid];
    CST[0][tid] = g * rsqrtf(p.var[tid] + p.eps);
    CST[1][tid] = p.beta[tid];
    CST[2][tid] = fabsf(g) > 1e-20f ? 1.f / g : 0.f;
  }
  const float inv_neg = p.alpha != 0.f ? 1.f / p.alpha : 1.f;
  float sga[NT][4], sbe[NT][4], sbi[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) sga[nt][j] = sbe[nt][j] = sbi[nt][j] = 0.f;
  const int tiles_img = p.tiles_x * ((p.H + TW - 1) / TW);
  // dy tile + halo of tile t into registers: clamped addresses, every load unconditional (a select on the loaded value or a branch around the
  // load would pin the wait right behind it); bit i of the returned mask says piece i lies inside the image, the zeros of the padding are
  // put in where the pieces are stored to LDS
  uint4 dv[NIT];
  auto load_dy = [&](int t) -> uint32_t {
    const int b = t / tiles_img, tile = t - b * tiles_img;
    const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
    const int64_t img = (int64_t)b * p.H * p.W;
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = tid + i * 256;
      const int hp = min(it >> 2, HWD * HWD - 1), c = it & 3;
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int gy = tyi * TW - 1 + hy, gx = txi * TW - 1 + hx;
      m |= (uint32_t)(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) << i;
      const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
      dv[i] = *reinterpret_cast<const uint4*>(p.dy + (img + (int64_t)cy * p.W + cx) * p.lddy + c * 8);
    }
    return m;
  };
  auto stage = [&](int buf, uint32_t m) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = tid + i * 256;
      const uint4 v = (m >> i) & 1u ? dv[i] : make_uint4(0, 0, 0, 0);
      if (it < HWD * HWD * 4) *reinterpret_cast<uint4*>(DT[buf] + (it >> 2) * DG_S + (it & 3) * 8) = v;
    }
  };
  int t = blockIdx.x, buf = 0;
  if (t < p.ntiles) {
    const uint32_t m = load_dy(t);
    stage(0, m);
  }
  __syncthreads();
  for (; t < p.ntiles; t += gridDim.x) {
    const int b = t / tiles_img, tile = t - b * tiles_img;
    const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
    const int ty0 = tyi * TW, tx0 = txi * TW;
    const int64_t img = (int64_t)b * p.H * p.W;
    // the stored activation of this lane's pixels (rows wv, wv+4, .., column r), its channels, then the NEXT tile's dy (this one's again on the
    // last trip): both in flight under the MFMAs, the activation waited for first
    uint32_t yv[4][CL / 2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = min(ty0 + wv + 4 * i, p.H - 1), gx = min(tx0 + r, p.W - 1);
      const bf16_t* yp = p.yact + (img + (int64_t)gy * p.W + gx) * p.ldy + CL * q;
      if constexpr (NT == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(yp);
        yv[i][0] = v.x; yv[i][1] = v.y; yv[i][CL / 2 - 2] = v.z; yv[i][CL / 2 - 1] = v.w;
      } else {
        const uint2 v = *reinterpret_cast<const uint2*>(yp);
        yv[i][0] = v.x; yv[i][1] = v.y;
      }
    }
    const int tn = t + (int)gridDim.x;
    const uint32_t mnext = load_dy(tn < p.ntiles ? tn : t);
    f32x4_t acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int ty = tp / 3, tx = tp - ty * 3;
      bf16x8_t at[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (WREG) at[nt] = a[tp][nt];
        else at[nt] = *reinterpret_cast<const bf16x8_t*>(WL + ((tp * NT + nt) * 64 + lane) * 8);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t bx = *reinterpret_cast<const bf16x8_t*>(DT[buf] + ((wv + 4 * i + ty) * HWD + (r + tx)) * DG_S + q * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[nt], bx, acc[i][nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = ty0 + wv + 4 * i, gx = tx0 + r;
      const bool ok = gy < p.H && gx < p.W;
      uint32_t o2[CL / 2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint32_t y01 = yv[i][2 * nt], y23 = yv[i][2 * nt + 1];
        const float y[4] = {__uint_as_float(y01 << 16), __uint_as_float(y01 & 0xffff0000u), __uint_as_float(y23 << 16), __uint_as_float(y23 & 0xffff0000u)};
        float o[4];
        float4 gr = make_float4(1.f, 1.f, 1.f, 1.f), be = gr, ig = gr;
        if (MODE == 2) {
          gr = *reinterpret_cast<const float4*>(&CST[0][CL * q + 4 * nt]);
          be = *reinterpret_cast<const float4*>(&CST[1][CL * q + 4 * nt]);
          ig = *reinterpret_cast<const float4*>(&CST[2][CL * q + 4 * nt]);
        }
        const float grj[4] = {gr.x, gr.y, gr.z, gr.w}, bej[4] = {be.x, be.y, be.z, be.w}, igj[4] = {ig.x, ig.y, ig.z, ig.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = ok ? acc[i][nt][j] : 0.f;
          if (MODE == 2) {      // usseg_norm_act_bwd mode 2, LeakyReLU: everything from the activated value
            const float xh = ((y[j] >= 0.f ? y[j] : y[j] * inv_neg) - bej[j]) * igj[j];
            const float dh = d * (y[j] > 0.f ? 1.f : p.alpha);
            sga[nt][j] = fmaf(dh, xh, sga[nt][j]);
            sbe[nt][j] += dh;
            o[j] = dh * grj[j];
            sbi[nt][j] += o[j];
          } else {              // usseg_act_bwd_colsum: the sums are those of the STORED (bf16) values
            o[j] = d * (y[j] >= 0.f ? 1.f : p.alpha);
          }
        }
        o2[2 * nt] = pack2bf(o[0], o[1]); o2[2 * nt + 1] = pack2bf(o[2], o[3]);
        if (MODE == 0) {
          sbi[nt][0] += __uint_as_float(o2[2 * nt] << 16); sbi[nt][1] += __uint_as_float(o2[2 * nt] & 0xffff0000u);
          sbi[nt][2] += __uint_as_float(o2[2 * nt + 1] << 16); sbi[nt][3] += __uint_as_float(o2[2 * nt + 1] & 0xffff0000u);
        }
      }
      if (ok) {
        bf16_t* xp = p.dx + (img + (int64_t)gy * p.W + gx) * p.lddx + CL * q;
        if constexpr (NT == 2) *reinterpret_cast<uint4*>(xp) = make_uint4(o2[0], o2[1], o2[CL / 2 - 2], o2[CL / 2 - 1]);
        else *reinterpret_cast<uint2*>(xp) = make_uint2(o2[0], o2[1]);
      }
    }
    stage(buf ^ 1, mnext);
    __syncthreads();      // the next tile is staged; every wave is done reading this one before anyone overwrites it a trip later
    buf ^= 1;
  }
  // per-channel sums over this workgroup's pixels: 16 lanes share a channel group (q), then the four waves through LDS
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v0 = sga[nt][j], v1 = sbe[nt][j], v2 = sbi[nt][j];
#pragma unroll
      for (int msk = 1; msk < 16; msk <<= 1) {
        v0 += __shfl_xor(v0, msk, 64); v1 += __shfl_xor(v1, msk, 64); v2 += __shfl_xor(v2, msk, 64);
      }
      if (r == 0) { RED[wv][q][(nt * 4 + j) * 3 + 0] = v0; RED[wv][q][(nt * 4 + j) * 3 + 1] = v1; RED[wv][q][(nt * 4 + j) * 3 + 2] = v2; }
    }
  __syncthreads();
  if (tid < CO * 3) {
    const int k = tid / CO, c = tid - k * CO;                 // sum k of channel c = CL q + 4 nt + j
    const int qq = c / CL, nt = (c % CL) >> 2, j = c & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += RED[w][qq][(nt * 4 + j) * 3 + k];
    p.ws[((int64_t)blockIdx.x * 3 + k) * CO + c] = s;
  }
#endif
}

}  // namespace

extern "C" int usseg_stem_fwd(int32_t B, int32_t H, int32_t W, const void* x, int32_t ldx, const void* w1, const float* b1, const void* w2,
                              const float* b2, const void* w3, const float* b3, const float* gamma, const float* beta, const float* mean,
                              const float* var, float eps, float alpha, void* y1, void* t1, void* c2, void* pooled, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && gamma && beta && mean && var && y1 && t1 && c2 && pooled, "stem_fwd: null pointer");
  USSEG_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ldx >= 8 && ldx % 8 == 0, "stem_fwd: bad geometry (H, W even; 8 physical input channels)");
  USSEG_CHECK_ARG(((((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)b3) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)var)) & 15) == 0,
                  "stem_fwd: per-channel vectors must be 16-byte aligned");
  StemFwd p = {};
  p.x = (const bf16_t*)x; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.w3 = (const bf16_t*)w3;
  p.b1 = b1; p.b2 = b2; p.b3 = b3; p.gamma = gamma; p.beta = beta; p.mean = mean; p.var = var;
  p.y1 = (bf16_t*)y1; p.t1 = (bf16_t*)t1; p.c2 = (bf16_t*)c2; p.pooled = (bf16_t*)pooled;
  p.B = B; p.H = H; p.W = W; p.ldx = ldx; p.alpha = alpha; p.eps = eps;
  p.tiles_x = (W + T - 1) / T;
  p.ntiles = p.tiles_x * ((H + TH - 1) / TH);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)stem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
    attr_done = true;
  }
  hipStream_t s = (hipStream_t)stream;
  const int slot = usseg_prof_start(4, s);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(p.ntiles, B), dim3(256), LDS_B, s, p);
  usseg_prof_stop(4, slot, s);
  return usseg_check_launch("stem_fwd");
}

extern "C" int usseg_conv3_dgrad_actbwd(int32_t B, int32_t H, int32_t W, const void* dy, int32_t lddy, const void* wd, int32_t Co, const void* yact,
                                        int32_t ldy, int32_t mode, const float* gamma, const float* beta, const float* var, float eps, float alpha,
                                        void* dx, int32_t lddx, float* dgamma, float* dbeta, float* dbias, float* ws, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dy && wd && yact && dx && dbias && ws && B > 0 && H > 0 && W > 0, "conv3_dgrad_actbwd: null pointer / bad geometry");
  USSEG_CHECK_ARG(lddy >= 32 && lddy % 8 == 0 && ldy >= Co && ldy % 4 == 0 && lddx >= Co && lddx % 4 == 0, "conv3_dgrad_actbwd: bad strides");
  USSEG_CHECK_ARG(mode == 0 || (mode == 2 && gamma && beta && var && dgamma && dbeta), "conv3_dgrad_actbwd: mode 0 (activation) or 2 (folded BatchNorm)");
  if (!((Co == 32 && mode == 2) || (Co == 16 && mode == 0))) {      // the two pairs of the stem (ResNest.py:39-44); anything else: the two launches
    usseg_set_error("conv3_dgrad_actbwd: no fused kernel for %d channels / mode %d", Co, mode);
    return USSEG_ERR_UNSUPPORTED;
  }
  DgAct p = {};
  p.dy = (const bf16_t*)dy; p.wd = (const bf16_t*)wd; p.yact = (const bf16_t*)yact; p.dx = (bf16_t*)dx;
  p.gamma = gamma; p.beta = beta; p.var = var;
  p.B = B; p.H = H; p.W = W; p.lddy = lddy; p.ldy = ldy; p.lddx = lddx; p.eps = eps; p.alpha = alpha;
  p.tiles_x = (W + 15) / 16;
  const int64_t total = (int64_t)p.tiles_x * ((H + 15) / 16) * B;
  USSEG_CHECK_ARG(total < (1ll << 31), "conv3_dgrad_actbwd: too many tiles");
  p.ntiles = (int)total;
  // persistent: two workgroups per CU (registers; 2 x 26 KB of LDS each), every one pipelines its tiles (the next tile's loads fly under this one's MFMAs)
  constexpr int DG_GRID = 512;
  static_assert(DG_GRID <= USSEG_REDUCE_MAX_BLOCKS, "one partial row per workgroup");
  const int grid = p.ntiles < DG_GRID ? p.ntiles : DG_GRID;
  hipStream_t s = (hipStream_t)stream;
  p.ws = usseg_defer_reduce_ws(s, ws, (int64_t)grid * 3 * Co);
  const int slot = usseg_prof_start(4, s);
  if (mode == 2) hipLaunchKernelGGL((dgrad_actbwd_kernel<32, 2>), dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((dgrad_actbwd_kernel<16, 0>), dim3(grid), dim3(256), 0, s, p);
  usseg_prof_stop(4, slot, s);
  if (mode == 2) usseg_launch_reduce_finish(p.ws, 1, grid, 3, Co, Co, 1.f, dgamma, dbeta, dbias, s, 0);
  else usseg_launch_reduce_finish(p.ws + 2 * Co, 1, grid, 1, 3 * Co, Co, 1.f, dbias, nullptr, nullptr, s, 0);
  return usseg_check_launch("conv3_dgrad_actbwd");
}
