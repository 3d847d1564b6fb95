// Fused cardinal-group kernels of a residual_S stage: ONE workgroup owns an 8x8-pixel tile of one image in LDS
// and runs the whole chain that the reference writes as ten layers (ResNest.py:136-147 per path, :99-101 shortcut):
//
//   forward:  x tile (+1 halo) -> grouped 1x1 conv (all paths = one GEMM) -> per-path LayerNorm + LeakyReLU -> grouped 3x3 conv (block-diagonal
//             implicit GEMM, zero padding applied to the NORMALISED tensor, so out-of-image halo pixels are zero) -> per-path LayerNorm +
//             LeakyReLU + the partial rows of the global average pool (ResNest.py:179), and from the same x tile the shortcut
//             1x1 conv -> LayerNorm -> LeakyReLU.
//
// Every tensor a backward pass needs is still written (u_raw, u, v_raw, y, sc_raw, sc), bf16-rounded at exactly the points where the
// unfused launches round, and the LayerNorms read those ROUNDED values from LDS as the unfused norm kernel reads them from HBM: the fused
// path has the arithmetic of igemm -> norm_act -> conv -> norm_act(+gap) | igemm -> norm_act, minus five launches and their cold starts.
// The work is tiny and latency-bound (AI 5-70 F/B): the halo recompute of the 1x1 (100 / 64 pixels) is free, the MFMA
// operands of the three GEMMs come from LDS (pixels) and straight from L2 (weights, register-prefetched four K steps ahead).
// MFMA orientation as in the conv kernels: A = weight rows (output channel), B = pixels, so a lane ends up with 4 consecutive channels
// of one pixel and stages them with 8-byte LDS writes.
#include "common.h"

#ifdef CARD_TIMING   /* diagnostic build (tools/time_cardinal.py --phases): workgroup (0,0,z) stamps the shader clock at its phase boundaries */
__device__ unsigned long long card_dbg[2][16];
#define CARD_STAMP(i) do { if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) card_dbg[blockIdx.z][i] = wall_clock64(); } while (0)
extern "C" int usseg_cardinal_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(card_dbg), sizeof(card_dbg)); }
#else
#define CARD_STAMP(i)
#endif

namespace {

struct CardFwd {
  const bf16_t* x;
  const bf16_t *w1, *w2, *wsc;
  const float *b1, *g1, *be1, *b2, *g2, *be2, *bsc, *gsc, *besc;
  bf16_t *u_raw, *u, *v_raw, *y, *sc_raw, *sc;
  float* gap;
  int32_t B, H, W, ldx, ldu, ldv, ldsc, tiles_x, ntiles;
  float eps, alpha;
};

constexpr int TILE = 8, LW = TILE + 2, NHALO = LW * LW, NPIX = TILE * TILE;   // 100 halo pixels, 64 interior pixels
constexpr int PF = 4;                                                         // K steps of weight fragments in flight
constexpr int STAT_B = 2 * 4 * 4 * 64 * 4;                                    // [sum | sum of squares][wave][group][lane] partials

__device__ __forceinline__ bf16x8_t ldg_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ bf16x8_t zero_frag() {
  bf16x8_t z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
  return z;
}

// LayerNormalization (+ LeakyReLU) of staged pixel rows SS[row][RS] (bf16, the values the unfused norm kernel would read from HBM): NG groups
// of CG logical channels, biased variance, two passes (the arithmetic of norm_act_kernel MODE 0).  A LANE owns a ROW (pixel) and a wave walks
// 8-channel chunks whose indices are COMPILE-TIME constants (one instantiation per chunk subset CSI, selected by a uniform switch on the wave
// id), so a channel's group, its gamma / beta address and every bound fold away: ~3 VALU instructions per element and pass.  (History: 4 lanes
// per pixel with per-element compare / select chains + shuffles: 19 us of a 65 us workgroup; lane = row with RUNTIME-uniform chunk indices:
// 14 us - a scalar division, compare chain and taken branch per element.)  Wave w owns row block w % RB (rows 64*(w % RB) + lane) and chunk
// subset w / RB; the per-row partial sums of the 4 / RB waves that share a row block meet in STAT.  All four waves must call it (two barriers
// inside).  Output: OUT[row][ORS], zeros in pad channels and - `live` false - in whole rows.
template <int CSI, int RB, int NCHK, int CG, int NG>
__device__ __forceinline__ void ln_body(const bf16_t* src, float* STAT, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                        float alpha, int wvu, int rb, int lane, bf16_t* dst, bool rv, bool live) {
  constexpr int CS = 4 / RB, CPS = (NCHK + CS - 1) / CS, C = CG * NG, C0 = CSI * CPS;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < CPS; ++j) {
    if (C0 + j < NCHK) {
      float x[8];
      unpack8(*reinterpret_cast<const uint4*>(src + (C0 + j) * 8), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (C0 + j) * 8 + e, g = ch / CG;
        if (ch < C) { if (g == 0) s0 += x[e]; else if (g == 1) s1 += x[e]; else s2 += x[e]; }
      }
    }
  }
  float* st = STAT + wvu * 256 + lane;
  st[0] = s0;
  if (NG > 1) { st[64] = s1; st[128] = s2; }
  __syncthreads();
  const float inv = 1.f / (float)CG;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int k = 0; k < CS; ++k) {
    const float* o = STAT + (k * RB + rb) * 256 + lane;
    m0 += o[0];
    if (NG > 1) { m1 += o[64]; m2 += o[128]; }
  }
  m0 *= inv; m1 *= inv; m2 *= inv;
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int j = 0; j < CPS; ++j) {
    if (C0 + j < NCHK) {
      float x[8];
      unpack8(*reinterpret_cast<const uint4*>(src + (C0 + j) * 8), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (C0 + j) * 8 + e, g = ch / CG;
        if (ch < C) {
          if (g == 0) { const float d = x[e] - m0; q0 = fmaf(d, d, q0); }
          else if (g == 1) { const float d = x[e] - m1; q1 = fmaf(d, d, q1); }
          else { const float d = x[e] - m2; q2 = fmaf(d, d, q2); }
        }
      }
    }
  }
  st = STAT + 1024 + wvu * 256 + lane;
  st[0] = q0;
  if (NG > 1) { st[64] = q1; st[128] = q2; }
  __syncthreads();
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
  for (int k = 0; k < CS; ++k) {
    const float* o = STAT + 1024 + (k * RB + rb) * 256 + lane;
    r0 += o[0];
    if (NG > 1) { r1 += o[64]; r2 += o[128]; }
  }
  r0 = rsqrtf(r0 * inv + eps); r1 = rsqrtf(r1 * inv + eps); r2 = rsqrtf(r2 * inv + eps);
#pragma unroll
  for (int j = 0; j < CPS; ++j) {
    if (C0 + j < NCHK) {
      float x[8];
      unpack8(*reinterpret_cast<const uint4*>(src + (C0 + j) * 8), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (C0 + j) * 8 + e, g = ch / CG;
        float t = 0.f;
        if (ch < C) {
          const float mean = g == 0 ? m0 : (g == 1 ? m1 : m2), rstd = g == 0 ? r0 : (g == 1 ? r1 : r2);
          t = gamma[ch] * ((x[e] - mean) * rstd) + beta[ch];
          t = t >= 0.f ? t : alpha * t;
        }
        x[e] = live ? t : 0.f;
      }
      if (rv) *reinterpret_cast<uint4*>(dst + (C0 + j) * 8) = pack8(x);
    }
  }
}

template <int RB, int NCHK, int CG, int NG>
__device__ __forceinline__ void tile_layernorm(const bf16_t* SS, int RS, int nrows, float* STAT, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps, float alpha, int wvu, int lane, bf16_t* OUT, int ORS,
                                               bool live) {
  constexpr int CS = 4 / RB;
  const int rb = wvu % RB, cs = wvu / RB;
  const int row = rb * 64 + lane;
  const bool rv = row < nrows;
  const bf16_t* src = SS + (rv ? row : 0) * RS;
  bf16_t* dst = OUT + (rv ? row : 0) * ORS;
  if (cs == 0) ln_body<0, RB, NCHK, CG, NG>(src, STAT, gamma, beta, eps, alpha, wvu, rb, lane, dst, rv, live);
  else if (cs == 1) ln_body<1, RB, NCHK, CG, NG>(src, STAT, gamma, beta, eps, alpha, wvu, rb, lane, dst, rv, live);
  else if (CS > 2 && cs == 2) ln_body<(CS > 2 ? 2 : 0), RB, NCHK, CG, NG>(src, STAT, gamma, beta, eps, alpha, wvu, rb, lane, dst, rv, live);
  else if (CS > 2) ln_body<(CS > 2 ? 3 : 0), RB, NCHK, CG, NG>(src, STAT, gamma, beta, eps, alpha, wvu, rb, lane, dst, rv, live);
}

template <int CIN, int CV11, int CVKK, int OC>
struct CardCfg {
  static constexpr int U = 3 * CV11, V = 3 * CVKK, UP = (U + 7) / 8 * 8, VP = (V + 7) / 8 * 8;
  static constexpr int CX = CIN + 8, CU = UP + 8, S1 = UP + 8, S2 = VP + 8, S3 = OC + 8;
  static constexpr int XS_B = NHALO * CX * 2, US_B = NHALO * CU * 2;
  static constexpr int SSA_E = 112 * S1 > NPIX * S2 ? 112 * S1 : NPIX * S2;        // cardinal role: u_raw staging, then v_raw staging
  static constexpr int LDS_CARD = XS_B + US_B + SSA_E * 2 + STAT_B;
  static constexpr int LDS_SC = NPIX * CX * 2 + NPIX * S3 * 2 + STAT_B;             // shortcut role: interior x tile + sc_raw staging
  static constexpr int LDS_B = LDS_CARD > LDS_SC ? LDS_CARD : LDS_SC;
};

// global -> LDS copy of pixel rows: every load of a thread is issued before its first LDS store (a load -> store loop waits one memory
// latency per trip)
template <int NROWS, int CIN, int CX, typename RowFn>
__device__ __forceinline__ void load_rows(bf16_t* XS, int tid, RowFn row_src /* row -> pointer or nullptr */) {
  constexpr int CH = CIN / 8, NIT = (NROWS * CH + 255) / 256;
  uint4 v[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = tid + i * 256;
    const int row = it / CH, c = it - row * CH;
    const bf16_t* src = it < NROWS * CH ? row_src(row) : nullptr;
    v[i] = src ? *reinterpret_cast<const uint4*>(src + c * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = tid + i * 256;
    const int row = it / CH, c = it - row * CH;
    if (it < NROWS * CH) *reinterpret_cast<uint4*>(XS + row * CX + c * 8) = v[i];
  }
}

// grid (tiles, B, 2): z = 0 the cardinal chain of a tile, z = 1 its shortcut (independent work: two short dependency chains side by side
// instead of one long one - at 16x16 a launch is 64 + 64 workgroups whose latency IS the launch time)
template <int CIN, int CV11, int CVKK, int OC>
__global__ __launch_bounds__(256) void cardinal_fwd_kernel(const CardFwd p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = CardCfg<CIN, CV11, CVKK, OC>;
  constexpr int UP = Cfg::UP, VP = Cfg::VP;
  constexpr int CX = Cfg::CX, CU = Cfg::CU, S1 = Cfg::S1, S2 = Cfg::S2, S3 = Cfg::S3;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, tile = blockIdx.x;
  const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
  const int ty0 = tyi * TILE, tx0 = txi * TILE;
  const int64_t img = (int64_t)b * p.H * p.W;
  constexpr int KS1 = CIN / 32;

  if (blockIdx.z == 1) {
    // ================================================================ shortcut: sc_raw = x . Wsc^T + bsc -> LayerNorm -> LeakyReLU (ResNest.py:99-101)
    bf16_t* const XS = reinterpret_cast<bf16_t*>(lds);
    bf16_t* const SS = reinterpret_cast<bf16_t*>(lds + NPIX * CX * 2);
    float* const STAT = reinterpret_cast<float*>(lds + NPIX * CX * 2 + NPIX * S3 * 2);
    constexpr int NTS = OC / 16, NPASS = NTS > 16 ? NTS / 16 : 1, NTWS = (NTS > 16 ? 16 : NTS) / 4;
    CARD_STAMP(0);
    bf16x8_t a3[PF][NTWS];
    auto wsc_frag = [&](int pass, int ks, int j) -> bf16x8_t {
      const int nt = pass * 16 + wv + 4 * j;
      return ldg_frag(p.wsc + (int64_t)(nt * 16 + r) * CIN + ks * 32 + q * 8);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
      for (int j = 0; j < NTWS; ++j) a3[s][j] = s < KS1 ? wsc_frag(0, s, j) : zero_frag();
    load_rows<NPIX, CIN, CX>(XS, tid, [&](int pp) -> const bf16_t* {
      const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
      return (gy < p.H && gx < p.W) ? p.x + (img + (int64_t)gy * p.W + gx) * p.ldx : nullptr;
    });
    __syncthreads();
    CARD_STAMP(1);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      f32x4_t acc[4][NTWS];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTWS; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (pass > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
          for (int j = 0; j < NTWS; ++j) a3[s][j] = s < KS1 ? wsc_frag(pass, s, j) : zero_frag();
      }
#pragma unroll
      for (int ks0 = 0; ks0 < KS1; ks0 += PF)
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          const int ks = ks0 + s;
          if (ks < KS1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bf16x8_t bx = *reinterpret_cast<const bf16x8_t*>(XS + (i * 16 + r) * CX + ks * 32 + q * 8);
#pragma unroll
              for (int j = 0; j < NTWS; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[s][j], bx, acc[i][j], 0, 0, 0);
            }
            if (ks + PF < KS1) {
#pragma unroll
              for (int j = 0; j < NTWS; ++j) a3[s][j] = wsc_frag(pass, ks + PF, j);
            }
          }
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = i * 16 + r;
#pragma unroll
        for (int j = 0; j < NTWS; ++j) {
          const int ch = (pass * 16 + wv + 4 * j) * 16 + 4 * q;
          const float4 bb = *reinterpret_cast<const float4*>(p.bsc + ch);
          uint2 o;
          o.x = pack2bf(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y);
          o.y = pack2bf(acc[i][j][2] + bb.z, acc[i][j][3] + bb.w);
          *reinterpret_cast<uint2*>(SS + pp * S3 + ch) = o;
        }
      }
    }
    __syncthreads();
    CARD_STAMP(2);
    constexpr int NCH3 = OC / 8;
    auto rowptr = [&](bf16_t* base, int pp) -> bf16_t* {
      const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
      return (gy < p.H && gx < p.W) ? base + (img + (int64_t)gy * p.W + gx) * p.ldsc : nullptr;
    };
    for (int it = tid; it < NPIX * NCH3; it += 256) {       // sc_raw -> HBM, whole rows, before the norm overwrites the staging tile
      const int pp = it / NCH3, c = it - pp * NCH3;
      bf16_t* dst = rowptr(p.sc_raw, pp);
      if (dst) *reinterpret_cast<uint4*>(dst + c * 8) = *reinterpret_cast<const uint4*>(SS + pp * S3 + c * 8);
    }
    CARD_STAMP(3);
    tile_layernorm<1, NCH3, OC, 1>(SS, S3, NPIX, STAT, p.gsc, p.besc, p.eps, p.alpha, wv, lane, SS, S3, true);
    __syncthreads();
    CARD_STAMP(4);
    for (int it = tid; it < NPIX * NCH3; it += 256) {
      const int pp = it / NCH3, c = it - pp * NCH3;
      bf16_t* dst = rowptr(p.sc, pp);
      if (dst) *reinterpret_cast<uint4*>(dst + c * 8) = *reinterpret_cast<const uint4*>(SS + pp * S3 + c * 8);
    }
    CARD_STAMP(5);
    return;
  }

  // ================================================================== cardinal chain (ResNest.py:136-147, all paths)
  bf16_t* const XS = reinterpret_cast<bf16_t*>(lds);
  bf16_t* const US = reinterpret_cast<bf16_t*>(lds + Cfg::XS_B);
  bf16_t* const SS = reinterpret_cast<bf16_t*>(lds + Cfg::XS_B + Cfg::US_B);
  float* const STAT = reinterpret_cast<float*>(lds + Cfg::XS_B + Cfg::US_B + Cfg::SSA_E * 2);

  CARD_STAMP(0);
  // ---- GEMM 1 weight fragments, ALL K steps, issued before anything else (their latency hides under the tile load).  Up to three channel
  //      tiles: wave w takes pixel tiles w and w+4 and every channel tile; six (the 256-channel stage): wave w takes channel tiles w and w+4
  //      and every pixel tile, so that its K-step fragments (2 x 8) fit in registers up front
  constexpr int NT1 = (UP + 15) / 16;
  constexpr bool NSPLIT = NT1 >= 4;
  constexpr int NA1 = NSPLIT ? 2 : NT1, MA1 = NSPLIT ? 7 : 2;
  static_assert(KS1 <= 8 && (NSPLIT || KS1 <= 4), "GEMM 1 keeps every K step's weight fragments in registers");
  bf16x8_t a1[KS1][NA1];
#pragma unroll
  for (int s = 0; s < KS1; ++s)
#pragma unroll
    for (int j = 0; j < NA1; ++j) {
      const int nt = NSPLIT ? wv + 4 * j : j;
      a1[s][j] = nt < NT1 ? ldg_frag(p.w1 + (int64_t)(nt * 16 + r) * CIN + s * 32 + q * 8) : zero_frag();
    }

  // ---- A: x tile + one-pixel halo -> LDS (zeros outside the image)
  load_rows<NHALO, CIN, CX>(XS, tid, [&](int hp) -> const bf16_t* {
    const int hy = hp / LW, hx = hp - hy * LW;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    return (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? p.x + (img + (int64_t)gy * p.W + gx) * p.ldx : nullptr;
  });
  __syncthreads();
  CARD_STAMP(1);

  // ---- B: u_raw[halo pixel][Up] = x . W1^T + b1 for the 100 halo pixels (7 pixel tiles of 16)
  {
    f32x4_t acc[MA1][NA1];
#pragma unroll
    for (int i = 0; i < MA1; ++i)
#pragma unroll
      for (int j = 0; j < NA1; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int prow[MA1];
#pragma unroll
    for (int i = 0; i < MA1; ++i) {
      const int pp = (NSPLIT ? i : wv + 4 * i) * 16 + r;
      prow[i] = (pp < NHALO ? pp : NHALO - 1) * CX;
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int i = 0; i < MA1; ++i)
        if (NSPLIT || wv + 4 * i < 7) {
          const bf16x8_t bx = *reinterpret_cast<const bf16x8_t*>(XS + prow[i] + ks * 32 + q * 8);
#pragma unroll
          for (int j = 0; j < NA1; ++j)
            if (!NSPLIT || wv + 4 * j < NT1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks][j], bx, acc[i][j], 0, 0, 0);
        }
#pragma unroll
    for (int i = 0; i < MA1; ++i) {
      const int mt = NSPLIT ? i : wv + 4 * i;
      const int pp = mt * 16 + r;
      if (mt < 7 && pp < NHALO) {
#pragma unroll
        for (int j = 0; j < NA1; ++j) {
          const int ch = (NSPLIT ? wv + 4 * j : j) * 16 + 4 * q;
          if (ch < UP) {
            const float4 bb = *reinterpret_cast<const float4*>(p.b1 + ch);
            uint2 o;
            o.x = pack2bf(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y);
            o.y = pack2bf(acc[i][j][2] + bb.z, acc[i][j][3] + bb.w);
            *reinterpret_cast<uint2*>(SS + pp * S1 + ch) = o;
          }
        }
      }
    }
  }
  // GEMM 2 = the grouped 3x3: block diagonal, so wave g < 3 computes PATH g only - its CVKK output channels from its CV11 input channels
  // (the 8-channel chunks that cover them; the foreign channels inside those chunks meet zero weights), all four pixel tiles: 1 / 2.4 of the
  // dense K x N work and weight bytes, and each weight fragment is pulled through this CU's L1 exactly once (the workgroup is bound by its
  // ~20 B/clk of L2 -> L1 traffic, not by MFMA).  Its first K steps' weight fragments are in flight during the row pass.
  constexpr int TPP = (CVKK + 15) / 16, PF2 = TPP > 4 ? 3 : PF;
  const int pcb = (wv * CV11) / 8, pnck = ((wv + 1) * CV11 + 7) / 8 - pcb, pn0 = wv * CVKK;     // chunk base / chunks per tap / first channel
  const int pnch = 9 * pnck, KSg = (pnch + 3) / 4;
  auto w2_frag = [&](int ks, int j) -> bf16x8_t {
    const int chunk = 4 * ks + q, row = 16 * j + r;
    const bool ok = chunk < pnch && row < CVKK;
    const int tap = ok ? chunk / pnck : 0, choff = ok ? chunk - tap * pnck : 0;
    const bf16x8_t v = ldg_frag(p.w2 + (int64_t)(pn0 + (ok ? row : 0)) * (9 * UP) + tap * UP + (pcb + choff) * 8);
    return ok ? v : zero_frag();
  };
  bf16x8_t a2[PF2][TPP];
  if (wv < 3) {
#pragma unroll
    for (int s = 0; s < PF2; ++s)
#pragma unroll
      for (int j = 0; j < TPP; ++j) a2[s][j] = s < KSg ? w2_frag(s, j) : zero_frag();
  }
  __syncthreads();
  CARD_STAMP(2);

  // ---- C: per-path LayerNorm + LeakyReLU of the 100 halo pixels: u -> LDS (ZERO outside the image: the 3x3 conv pads the normalised
  //         tensor); u_raw / u of the interior pixels -> HBM in whole rows
  {
    const int hp = (wv & 1) * 64 + lane;
    const int hy = hp / LW, hx = hp - hy * LW;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inimg = hp < NHALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    tile_layernorm<2, UP / 8, CV11, 3>(SS, S1, NHALO, STAT, p.g1, p.be1, p.eps, p.alpha, wv, lane, US, CU, inimg);
  }
  __syncthreads();
  CARD_STAMP(3);
  constexpr int CPT = UP / 8;
  for (int it = tid; it < NPIX * CPT; it += 256) {
    const int pp = it / CPT, c = it - pp * CPT;
    const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
    if (gy < p.H && gx < p.W) {
      const int hp = ((pp >> 3) + 1) * LW + (pp & 7) + 1;
      const int64_t gp = (img + (int64_t)gy * p.W + gx) * p.ldu + c * 8;
      *reinterpret_cast<uint4*>(p.u_raw + gp) = *reinterpret_cast<const uint4*>(SS + hp * S1 + c * 8);
      *reinterpret_cast<uint4*>(p.u + gp) = *reinterpret_cast<const uint4*>(US + hp * CU + c * 8);
    }
  }

  CARD_STAMP(4);
  // ---- D: v_raw[interior pixel][path channels] = conv3x3(u) + b2: implicit GEMM over K = (tap, 8-channel chunk of the path) from the LDS u tile
  {
    f32x4_t acc[4][TPP];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TPP; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (wv < 3) {
      int pbase[4];   // halo index of the (0,0) tap of this lane's pixel in each pixel tile
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = i * 16 + r;
        pbase[i] = (pp >> 3) * LW + (pp & 7);
      }
#pragma unroll 1
      for (int ks0 = 0; ks0 < KSg; ks0 += PF2) {
#pragma unroll
        for (int s = 0; s < PF2; ++s) {
          const int ks = ks0 + s;
          if (ks < KSg) {
            const int chunk = 4 * ks + q;
            const bool kv = chunk < pnch;
            const int tap = kv ? chunk / pnck : 0, choff = kv ? chunk - tap * pnck : 0;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int toff = (dy * LW + dx) * CU + (pcb + choff) * 8;
            bf16x8_t bu[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              bu[i] = *reinterpret_cast<const bf16x8_t*>(US + pbase[i] * CU + toff);
              if (!kv) bu[i] = zero_frag();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < TPP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s][j], bu[i], acc[i][j], 0, 0, 0);
            if (ks + PF2 < KSg) {
#pragma unroll
              for (int j = 0; j < TPP; ++j) a2[s][j] = w2_frag(ks + PF2, j);
            }
          }
        }
      }
    }
    CARD_STAMP(5);
    __syncthreads();      // every thread is done copying u_raw out of the staging tile
    if (wv < 3) {
      // a path starts at any channel (85, 170, ...): 2-byte LDS stores
#pragma unroll
      for (int j = 0; j < TPP; ++j) {
        float bb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = 16 * j + 4 * q + e;
          bb[e] = p.b2[pn0 + (cl < CVKK ? cl : 0)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int pp = i * 16 + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cl = 16 * j + 4 * q + e;
            if (cl < CVKK) SS[pp * S2 + pn0 + cl] = f2bf(acc[i][j][e] + bb[e]);
          }
        }
      }
    } else if (VP > 3 * CVKK) {     // the pad channels of the staged rows
      for (int c = 3 * CVKK; c < VP; ++c) SS[lane * S2 + c] = 0;
    }
  }
  __syncthreads();
  CARD_STAMP(6);

  // ---- E: v_raw -> HBM; per-path LayerNorm + LeakyReLU of the 64 interior pixels in place; y -> HBM; pooled partial row of this tile
  constexpr int NCV = VP / 8;
  for (int it = tid; it < NPIX * NCV; it += 256) {
    const int pp = it / NCV, c = it - pp * NCV;
    const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
    if (gy < p.H && gx < p.W)
      *reinterpret_cast<uint4*>(p.v_raw + (img + (int64_t)gy * p.W + gx) * p.ldv + c * 8) = *reinterpret_cast<const uint4*>(SS + pp * S2 + c * 8);
  }
  CARD_STAMP(7);
  tile_layernorm<1, NCV, CVKK, 3>(SS, S2, NPIX, STAT, p.g2, p.be2, p.eps, p.alpha, wv, lane, SS, S2, true);
  __syncthreads();
  CARD_STAMP(8);
  for (int it = tid; it < NPIX * NCV; it += 256) {
    const int pp = it / NCV, c = it - pp * NCV;
    const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
    if (gy < p.H && gx < p.W)
      *reinterpret_cast<uint4*>(p.y + (img + (int64_t)gy * p.W + gx) * p.ldv + c * 8) = *reinterpret_cast<const uint4*>(SS + pp * S2 + c * 8);
  }
  if (tid < VP) {       // the pool sums the STORED (bf16) values in pixel order, as a pass over y would
    float a = 0.f;
#pragma unroll 8
    for (int pp = 0; pp < NPIX; ++pp) {
      const bool in = ty0 + (pp >> 3) < p.H && tx0 + (pp & 7) < p.W;
      const float t = bf2f(SS[pp * S2 + tid]);
      a += in ? t : 0.f;
    }
    p.gap[((int64_t)b * p.ntiles + tile) * VP + tid] = a;
  }
  CARD_STAMP(9);
#endif
}

template <int CIN, int CV11, int CVKK, int OC>
int launch_fwd(const CardFwd& p, hipStream_t s) {
  using Cfg = CardCfg<CIN, CV11, CVKK, OC>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)cardinal_fwd_kernel<CIN, CV11, CVKK, OC>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_B);
    attr_done = true;
  }
  const int slot = usseg_prof_start(1, s);       // counted with the conv family (it replaces three of its launches per stage)
  hipLaunchKernelGGL((cardinal_fwd_kernel<CIN, CV11, CVKK, OC>), dim3(p.ntiles, p.B, 2), dim3(256), Cfg::LDS_B, s, p);
  usseg_prof_stop(1, slot, s);
  return usseg_check_launch("cardinal_fwd");
}

}  // namespace

// =====================================================================================================================================
// LayerNormalization + LeakyReLU BACKWARD of the cardinal group's norms (3 groups of 3..85 channels) and of the shortcut norm (one group of
// 64..512), on the same lane = pixel scheme.  The general kernel (pointwise.hip norm_act_kernel<true, 0, 3>) keeps a pixel on several lanes
// with per-lane 0/1 group masks folded into FMAs: ~30 FMAs per element, one wave per SIMD, 16-41 us per launch for 12-50 MB (7-10x its HBM
// time) - 180 us of the Arch B step.  Here a workgroup stages a 64-pixel tile of x and dy in LDS; row passes (lane = pixel, wave-uniform
// compile-time channels: no masks) produce the per-pixel statistics mean, 1/sigma, sum(dxh)/Cg, sum(dxh*xh)/Cg; a column pass (thread =
// channel) forms dx and the three per-channel sums (dgamma, dbeta, sum dx = the producing conv's bias gradient) in pixel order - no
// cross-lane reduction, bitwise reproducible; the tile leaves in whole rows.  Workgroups walk tiles (persistent), one partial row each.
namespace {

struct LnTile {
  LnTileArgs a;
  int32_t ntiles;
};

template <int CSI, int CP, int CG, int NG, bool FUSE>
__device__ __forceinline__ void lnb_rows(const bf16_t* xrow, const bf16_t* dyrow, float* STAT, float4* pst, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, const float* SA, float eps, float alpha, int wvu, int lane) {
  constexpr int CPS = (CP + 3) / 4, C = CG * NG, C0 = CSI * CPS, CPH = CP * 8;
  const float inv = 1.f / (float)CG;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < CPS; ++j)
    if (C0 + j < CP) {
      float x[8];
      unpack8(*reinterpret_cast<const uint4*>(xrow + (C0 + j) * 8), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (C0 + j) * 8 + e, g = ch / CG;
        if (ch < C) { if (g == 0) s0 += x[e]; else if (g == 1) s1 += x[e]; else s2 += x[e]; }
      }
    }
  float* st = STAT + wvu * 256 + lane;
  st[0] = s0;
  if (NG > 1) { st[64] = s1; st[128] = s2; }
  __syncthreads();
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* o = STAT + k * 256 + lane;
    m0 += o[0];
    if (NG > 1) { m1 += o[64]; m2 += o[128]; }
  }
  m0 *= inv; m1 *= inv; m2 *= inv;
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int j = 0; j < CPS; ++j)
    if (C0 + j < CP) {
      float x[8];
      unpack8(*reinterpret_cast<const uint4*>(xrow + (C0 + j) * 8), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (C0 + j) * 8 + e, g = ch / CG;
        if (ch < C) {
          const float d = x[e] - (g == 0 ? m0 : (g == 1 ? m1 : m2));
          if (g == 0) q0 = fmaf(d, d, q0); else if (g == 1) q1 = fmaf(d, d, q1); else q2 = fmaf(d, d, q2);
        }
      }
    }
  st = STAT + 1024 + wvu * 256 + lane;
  st[0] = q0;
  if (NG > 1) { st[64] = q1; st[128] = q2; }
  __syncthreads();
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* o = STAT + 1024 + k * 256 + lane;
    r0 += o[0];
    if (NG > 1) { r1 += o[64]; r2 += o[128]; }
  }
  r0 = rsqrtf(r0 * inv + eps); r1 = rsqrtf(r1 * inv + eps); r2 = rsqrtf(r2 * inv + eps);
  // sums of dxh and dxh * xh over each group (dxh = dy * act'(gamma * xh + beta) * gamma)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
  for (int j = 0; j < CPS; ++j)
    if (C0 + j < CP) {
      float x[8], dy[8];
      unpack8(*reinterpret_cast<const uint4*>(xrow + (C0 + j) * 8), x);
      unpack8(*reinterpret_cast<const uint4*>(dyrow + (C0 + j) * 8), dy);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (C0 + j) * 8 + e, g = ch / CG;
        if (ch < C) {
          const float mean = g == 0 ? m0 : (g == 1 ? m1 : m2), rstd = g == 0 ? r0 : (g == 1 ? r1 : r2);
          const float xh = (x[e] - mean) * rstd;
          const float pre = gamma[ch] * xh + beta[ch];
          const float dye = FUSE ? fmaf(dy[e], SA[ch], SA[CPH + ch]) : dy[e];       // wave-uniform LDS reads
          const float dxh = dye * (pre >= 0.f ? 1.f : alpha) * gamma[ch];
          if (g == 0) { a0 += dxh; b0 = fmaf(dxh, xh, b0); }
          else if (g == 1) { a1 += dxh; b1 = fmaf(dxh, xh, b1); }
          else { a2 += dxh; b2 = fmaf(dxh, xh, b2); }
        }
      }
    }
  st = STAT + 2048 + wvu * 256 + lane;
  st[0] = a0;
  if (NG > 1) { st[64] = a1; st[128] = a2; }
  st = STAT + 3072 + wvu * 256 + lane;
  st[0] = b0;
  if (NG > 1) { st[64] = b1; st[128] = b2; }
  __syncthreads();
  if (CSI == 0) {
    float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        sa[g] += STAT[2048 + k * 256 + g * 64 + lane];
        sb[g] += STAT[3072 + k * 256 + g * 64 + lane];
      }
    pst[lane * NG + 0] = make_float4(m0, r0, sa[0] * inv, sb[0] * inv);
    if (NG > 1) {
      pst[lane * NG + 1] = make_float4(m1, r1, sa[1] * inv, sb[1] * inv);
      pst[lane * NG + 2] = make_float4(m2, r2, sa[2] * inv, sb[2] * inv);
    }
  }
}

template <int CP, int CG, int NG, bool FUSE>
struct LnbCfg {
  static constexpr int CPH = CP * 8, XS = CPH + 8;
  static constexpr int XT_B = 64 * XS * 2, PST_B = 64 * NG * 16, STAT_B4 = 4 * 1024 * 4, SA_B = FUSE ? 2 * CPH * 4 : 0;
  static constexpr int PG = CPH >= 256 ? 1 : 256 / CPH, NR = CPH > 256 ? CPH / 256 : 1;
  static constexpr int RED_B = PG * 3 * CPH * 4;
  static constexpr int LDS_B = 2 * XT_B + PST_B + STAT_B4 + RED_B + SA_B;
};

// FUSE (the split-attention re-weighting's backward dy_eff = sa_mult*s[b][c]*dy + dg[b][c], formed in fp32 and never stored): a tile lies
// inside ONE image (the launcher checks HW % 64 == 0), so its (s, dg) row is staged in LDS once per tile.
template <int CP, int CG, int NG, bool FUSE>
__global__ __launch_bounds__(256) void ln_bwd_tile_kernel(const LnTile P) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = LnbCfg<CP, CG, NG, FUSE>;
  constexpr int CPH = Cfg::CPH, XS = Cfg::XS, PG = Cfg::PG, NR = Cfg::NR, C = CG * NG;
  const LnTileArgs& a = P.a;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  bf16_t* const XT = reinterpret_cast<bf16_t*>(lds);
  bf16_t* const DT = reinterpret_cast<bf16_t*>(lds + Cfg::XT_B);
  float4* const PST = reinterpret_cast<float4*>(lds + 2 * Cfg::XT_B);
  float* const STAT = reinterpret_cast<float*>(lds + 2 * Cfg::XT_B + Cfg::PST_B);
  float* const RED = reinterpret_cast<float*>(lds + 2 * Cfg::XT_B + Cfg::PST_B + Cfg::STAT_B4);
  float* const SA = reinterpret_cast<float*>(lds + 2 * Cfg::XT_B + Cfg::PST_B + Cfg::STAT_B4 + Cfg::RED_B);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // column role: channel ct (+ 256 in the second round of a 512-channel norm), pixel group cpg
  const int ct = CPH >= 256 ? tid : tid % CPH, cpg = CPH >= 256 ? 0 : tid / CPH;
  const bool col_on = CPH >= 256 || tid < PG * CPH;
  float acc[NR][3];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.f;
  float cga[NR], cbe[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int ch = ct + 256 * r;
    cga[r] = ch < C ? a.gamma[ch] : 0.f;
    cbe[r] = ch < C ? a.beta[ch] : 0.f;
  }
  constexpr int NIT = (64 * CP + 255) / 256;
  uint4 vx[NIT], vd[NIT];
  float sav = 0.f, sag = 0.f;
  auto issue = [&](int tile) {      // the tile's global loads into registers (in flight while the previous tile is processed)
    const int64_t m0 = (int64_t)tile * 64;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = tid + i * 256;
      const int row = it / CP, c = it - row * CP;
      const bool ok = it < 64 * CP && m0 + row < a.M;
      vx[i] = ok ? *reinterpret_cast<const uint4*>(a.x + (m0 + row) * a.ldx + c * 8) : make_uint4(0, 0, 0, 0);
      vd[i] = ok ? *reinterpret_cast<const uint4*>(a.dy + (m0 + row) * a.lddy + c * 8) : make_uint4(0, 0, 0, 0);
    }
    if (FUSE) {
      const int64_t bimg = m0 / a.HW;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int ch = tid + 256 * r;
        if (r == 0 && ch < CPH) {
          const int cc = ch < a.sa_cy ? ch : a.sa_cy - 1;
          sav = ch < C ? a.sa_mult * a.sa_s[bimg * a.sa_cy + cc] : 0.f;
          sag = ch < C ? a.sa_dg[bimg * a.sa_cy + cc] : 0.f;
        }
      }
    }
  };
  if ((int)blockIdx.x < P.ntiles) issue(blockIdx.x);

  for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    const int64_t m0 = (int64_t)tile * 64;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = tid + i * 256;
      const int row = it / CP, c = it - row * CP;
      if (it < 64 * CP) {
        *reinterpret_cast<uint4*>(XT + row * XS + c * 8) = vx[i];
        *reinterpret_cast<uint4*>(DT + row * XS + c * 8) = vd[i];
      }
    }
    if (FUSE && tid < CPH) { SA[tid] = sav; SA[CPH + tid] = sag; }
    __syncthreads();
    if (tile + (int)gridDim.x < P.ntiles) issue(tile + gridDim.x);
    // ---- row passes: per-pixel statistics
    {
      const bf16_t* xrow = XT + lane * XS;
      const bf16_t* dyrow = DT + lane * XS;
      if (wv == 0) lnb_rows<0, CP, CG, NG, FUSE>(xrow, dyrow, STAT, PST, a.gamma, a.beta, SA, a.eps, a.alpha, wv, lane);
      else if (wv == 1) lnb_rows<1, CP, CG, NG, FUSE>(xrow, dyrow, STAT, PST, a.gamma, a.beta, SA, a.eps, a.alpha, wv, lane);
      else if (wv == 2) lnb_rows<2, CP, CG, NG, FUSE>(xrow, dyrow, STAT, PST, a.gamma, a.beta, SA, a.eps, a.alpha, wv, lane);
      else lnb_rows<3, CP, CG, NG, FUSE>(xrow, dyrow, STAT, PST, a.gamma, a.beta, SA, a.eps, a.alpha, wv, lane);
    }
    __syncthreads();
    // ---- column pass: dx (in place of x) and the per-channel sums, pixels in order
    if (col_on) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int ch = ct + 256 * r;
        const int g = NG == 1 ? 0 : (ch >= CG) + (ch >= 2 * CG);
        const float sv = FUSE ? SA[ch] : 1.f, sg = FUSE ? SA[CPH + ch] : 0.f;
        float sga = 0.f, sbe = 0.f, sbi = 0.f;
#pragma unroll 4
        for (int px = cpg; px < 64; px += PG) {
          float o = 0.f;
          if (ch < C) {
            const float4 st = PST[px * NG + (g < NG ? g : 0)];
            const float x = bf2f(XT[px * XS + ch]);
            float dy = bf2f(DT[px * XS + ch]);
            if (FUSE) dy = fmaf(dy, sv, sg);
            const float xh = (x - st.x) * st.y;
            const float pre = cga[r] * xh + cbe[r];
            const float dh = dy * (pre >= 0.f ? 1.f : a.alpha);
            const float dxh = dh * cga[r];
            o = st.y * (dxh - st.z - xh * st.w);
            if (m0 + px < a.M) { sga = fmaf(dh, xh, sga); sbe += dh; sbi += o; }
          }
          XT[px * XS + ch] = f2bf(o);
        }
        if (PG == 1) { acc[r][0] += sga; acc[r][1] += sbe; acc[r][2] += sbi; }
        else { RED[(cpg * 3 + 0) * CPH + ch] = sga; RED[(cpg * 3 + 1) * CPH + ch] = sbe; RED[(cpg * 3 + 2) * CPH + ch] = sbi; }
      }
    }
    __syncthreads();
    if (PG > 1 && tid < CPH) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float t = 0.f;
        for (int gq = 0; gq < PG; ++gq) t += RED[(gq * 3 + k) * CPH + tid];
        acc[0][k] += t;
      }
    }
    // ---- dx leaves in whole rows
    for (int it = tid; it < 64 * CP; it += 256) {
      const int row = it / CP, c = it - row * CP;
      if (m0 + row < a.M) *reinterpret_cast<uint4*>(a.dx + (m0 + row) * a.lddx + c * 8) = *reinterpret_cast<const uint4*>(XT + row * XS + c * 8);
    }
    __syncthreads();
  }
  if (CPH >= 256 || tid < CPH) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) a.ws[((int64_t)blockIdx.x * 3 + k) * CPH + ct + 256 * r] = acc[r][k];
  }
#endif
}

template <int CP, int CG, int NG, bool FUSE>
int launch_lnb(const LnTileArgs& a, float* dgamma, float* dbeta, float* dbias, float* caller_ws, hipStream_t s) {
  using Cfg = LnbCfg<CP, CG, NG, FUSE>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)ln_bwd_tile_kernel<CP, CG, NG, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_B);
    attr_done = true;
  }
  LnTile P;
  P.a = a;
  P.ntiles = (int)((a.M + 63) / 64);
  const int per_cu = Cfg::LDS_B > 80 * 1024 ? 1 : (Cfg::LDS_B > 40 * 1024 ? 2 : 4);
  int grid = P.ntiles < 256 * per_cu ? P.ntiles : 256 * per_cu;
  if (grid > USSEG_REDUCE_MAX_BLOCKS) grid = USSEG_REDUCE_MAX_BLOCKS;
  P.a.ws = usseg_defer_reduce_ws(s, caller_ws, (int64_t)grid * 3 * Cfg::CPH);
  hipLaunchKernelGGL((ln_bwd_tile_kernel<CP, CG, NG, FUSE>), dim3(grid), dim3(256), Cfg::LDS_B, s, P);
  usseg_launch_reduce_finish(P.a.ws, 1, grid, 3, Cfg::CPH, a.C, 1.f, dgamma, dbeta, dbias, s, 0);
  return 1;
}

}  // namespace

int usseg_try_ln_bwd_tile(const LnTileArgs& a, float* dgamma, float* dbeta, float* dbias, float* caller_ws, hipStream_t s) {
  static const int off = getenv("USSEG_LN_TILE") && atoi(getenv("USSEG_LN_TILE")) == 0;
  if (off || a.M <= 0 || a.M >= (1ll << 31)) return 0;
  const bool fuse = a.sa_s != nullptr;
  // a 64-pixel tile is one long dependency chain per workgroup: it beats the streaming kernel (one pixel on 4-16 lanes, mask FMAs) once the
  // tensor has several tiles per CU to overlap (measured: 2x on 128x128 / 64x64 x 16 images, slower below ~32k pixels)
  static const int64_t min_px = getenv("USSEG_LN_TILE_MIN") ? atoll(getenv("USSEG_LN_TILE_MIN")) : 32768;
  if (a.M < min_px || (fuse && a.HW % 64 != 0)) return 0;
  const int cg = a.G > 0 ? a.C / a.G : 0;
#define LNB(CPv, CGv, NGv)                                                                                                  \
  if (a.Cphys == CPv * 8 && a.G == NGv && cg == CGv)                                                                        \
    return fuse ? launch_lnb<CPv, CGv, NGv, true>(a, dgamma, dbeta, dbias, caller_ws, s) : launch_lnb<CPv, CGv, NGv, false>(a, dgamma, dbeta, dbias, caller_ws, s);
  LNB(2, 3, 3) LNB(3, 7, 3) LNB(6, 14, 3) LNB(11, 28, 3)          // conv1_bn of the four stages (ResNest.py:140)
  LNB(4, 10, 3) LNB(8, 21, 3) LNB(16, 42, 3) LNB(32, 85, 3)       // conv2_bn (:143), also with the split-attention re-weighting folded in
  if (!fuse) { LNB(8, 64, 1) LNB(16, 128, 1) LNB(32, 256, 1) LNB(64, 512, 1) }   // convtmp_scbn (:100), DecoderCup.bn1 (Decoder.py:112)
#undef LNB
  return 0;
}

// The channel configurations of a ResNest.py stage with radix 3 / kpaths 3 (ResNest.py:120-121: cv11 = 3/7/14/28, cvkk = 10/21/42/85)
static int card_config(const UssegCardinalDesc* d) {
  if (!d || d->P != 3) return -1;
  const int U = d->P * d->cv11, V = d->P * d->cvkk;
  if (d->Up != (U + 7) / 8 * 8 || d->Vp != (V + 7) / 8 * 8) return -1;
  if (d->Cin == 32 && d->cv11 == 3 && d->cvkk == 10 && d->Oc == 64) return 0;
  if (d->Cin == 64 && d->cv11 == 7 && d->cvkk == 21 && d->Oc == 128) return 1;
  if (d->Cin == 128 && d->cv11 == 14 && d->cvkk == 42 && d->Oc == 256) return 2;
  if (d->Cin == 256 && d->cv11 == 28 && d->cvkk == 85 && d->Oc == 512) return 3;
  return -1;
}

extern "C" int32_t usseg_cardinal_supported(const UssegCardinalDesc* d) { return card_config(d) >= 0 ? 1 : 0; }

extern "C" int usseg_cardinal_fwd(const UssegCardinalDesc* d, const void* x, const void* w1, const float* b1, const float* g1, const float* be1,
                                  const void* w2, const float* b2, const float* g2, const float* be2, const void* wsc, const float* bsc,
                                  const float* gsc, const float* besc, void* u_raw, void* u, void* v_raw, void* y, float* gap_rows,
                                  void* sc_raw, void* sc, usseg_stream_t stream) {
  const int cfg = card_config(d);
  USSEG_CHECK_ARG(cfg >= 0, "cardinal_fwd: no fused kernel for this channel configuration (usseg_cardinal_supported)");
  USSEG_CHECK_ARG(x && w1 && b1 && g1 && be1 && w2 && b2 && g2 && be2 && wsc && bsc && gsc && besc && u_raw && u && v_raw && y && gap_rows && sc_raw && sc,
                  "cardinal_fwd: null pointer");
  USSEG_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->ldx >= d->Cin && d->ldu >= d->Up && d->ldv >= d->Vp && d->ldsc >= d->Oc &&
                      d->ldx % 8 == 0 && d->ldu % 8 == 0 && d->ldv % 8 == 0 && d->ldsc % 8 == 0,
                  "cardinal_fwd: bad geometry / strides");
  USSEG_CHECK_ARG(((((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)bsc)) & 15) == 0, "cardinal_fwd: bias vectors must be 16-byte aligned");
  CardFwd p = {};
  p.x = (const bf16_t*)x; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.wsc = (const bf16_t*)wsc;
  p.b1 = b1; p.g1 = g1; p.be1 = be1; p.b2 = b2; p.g2 = g2; p.be2 = be2; p.bsc = bsc; p.gsc = gsc; p.besc = besc;
  p.u_raw = (bf16_t*)u_raw; p.u = (bf16_t*)u; p.v_raw = (bf16_t*)v_raw; p.y = (bf16_t*)y; p.sc_raw = (bf16_t*)sc_raw; p.sc = (bf16_t*)sc;
  p.gap = gap_rows;
  p.B = d->B; p.H = d->H; p.W = d->W; p.ldx = d->ldx; p.ldu = d->ldu; p.ldv = d->ldv; p.ldsc = d->ldsc;
  p.tiles_x = (d->W + TILE - 1) / TILE;
  p.ntiles = p.tiles_x * ((d->H + TILE - 1) / TILE);
  p.eps = d->eps; p.alpha = d->alpha;
  hipStream_t s = (hipStream_t)stream;
  switch (cfg) {
    case 0: return launch_fwd<32, 3, 10, 64>(p, s);
    case 1: return launch_fwd<64, 7, 21, 128>(p, s);
    case 2: return launch_fwd<128, 14, 42, 256>(p, s);
    default: return launch_fwd<256, 28, 85, 512>(p, s);
  }
}
