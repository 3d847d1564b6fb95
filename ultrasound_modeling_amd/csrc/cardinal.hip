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
  // No select on the loaded value (`ok ? v : 0` made the compiler wait for every fragment right behind its load - vmcnt(0) - so nothing was
  // ever in flight): a K chunk past the path's last one re-reads the last chunk and meets a ZEROED pixel fragment below, a row past the
  // path's last channel re-reads the last row and lands in accumulator rows that are never stored.
  auto w2_frag = [&](int ks, int j) -> bf16x8_t {
    int chunk = 4 * ks + q, row = 16 * j + r;
    chunk = chunk < pnch ? chunk : pnch - 1;
    row = row < CVKK ? row : CVKK - 1;
    const int tap = chunk / pnck, choff = chunk - tap * pnck;
    return ldg_frag(p.w2 + (int64_t)(pn0 + row) * (9 * UP) + tap * UP + (pcb + choff) * 8);
  };
  bf16x8_t a2[PF2][TPP];
  if (wv < 3) {
#pragma unroll
    for (int s = 0; s < PF2; ++s)
#pragma unroll
      for (int j = 0; j < TPP; ++j) a2[s][j] = w2_frag(s, j);
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
          {   // (no `ks < KSg` branch: a step past the end multiplies clamped weights with zeroed pixel fragments; a branch here makes the
              //  compiler drain every weight load in flight - vmcnt(0) - at its join)
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
#pragma unroll
            for (int j = 0; j < TPP; ++j) a2[s][j] = w2_frag(ks + PF2, j);      // (unconditional, clamped: the compiler's vmcnt count stays exact)
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
  const int slot = usseg_prof_start(4, s);       // timed as a fused tile kernel (usseg_prof kind 4): convs + norms in one launch
  hipLaunchKernelGGL((cardinal_fwd_kernel<CIN, CV11, CVKK, OC>), dim3(p.ntiles, p.B, 2), dim3(256), Cfg::LDS_B, s, p);
  usseg_prof_stop(4, slot, s);
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

// Waves = RB row blocks x 4 / RB chunk subsets (wave wvu: row block wvu % RB = rb, subset CSI = wvu / RB); `pst` points at this row block's
// first entry.
template <int CSI, int CP, int CG, int NG, bool FUSE, int RB = 1>
__device__ __forceinline__ void lnb_rows(const bf16_t* xrow, const bf16_t* dyrow, float* STAT, float4* pst, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, const float* SA, float eps, float alpha, int wvu, int lane, int rb = 0) {
  constexpr int CS = 4 / RB, CPS = (CP + CS - 1) / CS, C = CG * NG, C0 = CSI * CPS, CPH = CP * 8;
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
  for (int k = 0; k < CS; ++k) {
    const float* o = STAT + (k * RB + rb) * 256 + lane;
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
  for (int k = 0; k < CS; ++k) {
    const float* o = STAT + 1024 + (k * RB + rb) * 256 + lane;
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
    for (int k = 0; k < CS; ++k)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        sa[g] += STAT[2048 + (k * RB + rb) * 256 + g * 64 + lane];
        sb[g] += STAT[3072 + (k * RB + rb) * 256 + g * 64 + lane];
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
// workgroup `blk` of `nblk` walks the tiles blk, blk + nblk, ... and leaves ONE partial row a.ws[blk][3][CPH]
template <int CP, int CG, int NG, bool FUSE>
__device__ __forceinline__ void lnb_tile_loop(const LnTileArgs& a, const int ntiles, char* lds, const int blk, const int nblk) {
  using Cfg = LnbCfg<CP, CG, NG, FUSE>;
  constexpr int CPH = Cfg::CPH, XS = Cfg::XS, PG = Cfg::PG, NR = Cfg::NR, C = CG * NG;
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
  if (blk < ntiles) issue(blk);

  for (int tile = blk; tile < ntiles; tile += nblk) {
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
    if (tile + nblk < ntiles) issue(tile + nblk);
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
      for (int k = 0; k < 3; ++k) a.ws[((int64_t)blk * 3 + k) * CPH + ct + 256 * r] = acc[r][k];
  }
}

template <int CP, int CG, int NG, bool FUSE>
__global__ __launch_bounds__(256) void ln_bwd_tile_kernel(const LnTile P) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  lnb_tile_loop<CP, CG, NG, FUSE>(P.a, P.ntiles, lds, blockIdx.x, gridDim.x);
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

// =====================================================================================================================================
// K3 backward: one launch for the backward pass of a stage's cardinal chain AND of its shortcut norm.
//
//   role 0 (even workgroups), one 8x8-pixel tile (+1 halo) at a time:
//     dout, v_raw on the 100 halo pixels -> [split-attention re-weighting backward + conv2_bn LayerNorm/LeakyReLU backward] = dv on the halo
//     (zero outside the image: the backward-data pass of a zero-padded conv pads dv) -> dv of the interior -> HBM (the 3x3 weight gradient
//     reads it) -> grouped 3x3 BACKWARD-DATA as a block-diagonal implicit GEMM over (flipped tap, 8-channel chunk of the path) from the LDS
//     dv tile -> du (bf16, as the unfused launch stores it) -> conv1_bn LayerNorm/LeakyReLU backward with the u_raw tile -> du_raw -> dcat[:, :Up];
//   role 1 (odd workgroups): the shortcut norm's backward (convtmp_scbn, ResNest.py:100-101) on 64-pixel tiles -> dcat[:, Up:].
//
// Replaces norm_act_bwd_sa -> conv2d_dgrad (3x3) -> norm_act_bwd and the shortcut's norm_act_bwd; the halo rows of the first norm are
// recomputed (100 / 64).  Per-channel sums (dgamma, dbeta, the producing conv's bias gradient) are taken over INTERIOR pixels only, in
// pixel order per workgroup, one partial row per workgroup and norm: bitwise reproducible.
struct CardBwd {
  const bf16_t *dout, *v_raw, *u_raw, *w2d;
  const float *g2, *be2, *g1, *be1, *sa_s, *sa_dg;
  bf16_t *dv, *dcat;
  float *ws2, *ws1;                 // partial rows [G][3][Vp], [G][3][Up]
  int32_t B, H, W, ldo, ldv, ldu, lddv, ldc, tiles_x, tiles_img, ntiles, G;
  float eps, alpha, sa_mult;
  LnTileArgs sc;                    // the shortcut norm (role 1)
  int32_t sc_tiles;
};

// column pass: a thread owns four channels of every PG-th row
template <int CPH, int CG>
struct LnColCfg {
  static constexpr int NQ = CPH / 4, PG = 256 / NQ > 32 ? 32 : 256 / NQ;
};

template <int CV11, int CVKK, int OC>
struct CardBwdCfg {
  static constexpr int U = 3 * CV11, V = 3 * CVKK, UP = (U + 7) / 8 * 8, VP = (V + 7) / 8 * 8;
  static constexpr int S1 = UP + 8, S2 = VP + 8;
  static constexpr int VT_B = NHALO * S2 * 2, UT_B = NPIX * S1 * 2;
  static constexpr int PST_B = 128 * 3 * 16, STAT_B4 = 4 * 1024 * 4, SA_B = 2 * VP * 4, FLG_B = 128;
  static constexpr int RED2 = LnColCfg<VP, CVKK>::PG * 3 * VP, RED1 = LnColCfg<UP, CV11>::PG * 3 * UP, RED_B = (RED2 > RED1 ? RED2 : RED1) * 4;
  static constexpr int OFF_DT = VT_B, OFF_PST = 2 * VT_B, OFF_STAT = OFF_PST + PST_B, OFF_RED = OFF_STAT + STAT_B4, OFF_SA = OFF_RED + RED_B,
                       OFF_FLG = OFF_SA + SA_B, OFF_UT = OFF_FLG + FLG_B;
  static constexpr int LDS_CARD = OFF_UT + UT_B;
  static constexpr int LDS_SC = LnbCfg<OC / 8, OC, 1, false>::LDS_B;
  static constexpr int LDS_B = LDS_CARD > LDS_SC ? LDS_CARD : LDS_SC;
  static_assert(UT_B <= VT_B, "the du tile lives in the dout tile's LDS once the first norm is done");
};

// column pass of a LayerNorm backward over NPX staged rows: a thread owns FOUR consecutive channels (8-byte LDS accesses; 2-byte ones made
// this pass LDS-instruction bound: 12 us of a 53 us workgroup at 256 channels) of every PG-th row; dx replaces x in XT; the three
// per-channel sums of the rows whose flag has bit 1 set go to RED[pixel group][3][CPH], added up by lnb_cols_finish
template <int CPH, int CG, int NPX, bool FUSE>
__device__ __forceinline__ void lnb_cols(bf16_t* XT, const bf16_t* DT, int XS, const float4* PST, const float* SA, const unsigned char* FLG,
                                         bool halo, float* RED, const float* __restrict__ gamma, const float* __restrict__ beta, float alpha, int tid) {
  constexpr int C = 3 * CG, NQ = LnColCfg<CPH, CG>::NQ, PG = LnColCfg<CPH, CG>::PG;
  const int cq = tid % NQ, cpg = tid / NQ;
  if (cpg >= PG) return;
  const int c0 = 4 * cq;
  float cga[4], cbe[4], sv[4], sg[4], sga[4], sbe[4], sbi[4];
  int gi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = c0 + e;
    const bool on = ch < C;
    cga[e] = on ? gamma[ch] : 0.f;
    cbe[e] = on ? beta[ch] : 0.f;
    sv[e] = FUSE ? SA[ch] : 1.f;
    sg[e] = FUSE ? SA[CPH + ch] : 0.f;
    gi[e] = on ? (ch >= CG) + (ch >= 2 * CG) : 2;
    sga[e] = sbe[e] = sbi[e] = 0.f;
  }
#pragma unroll 2
  for (int px = cpg; px < NPX; px += PG) {
    const int hp = halo ? px : ((px >> 3) + 1) * LW + (px & 7) + 1;
    const unsigned f = FLG[hp];
    const float4 stA = PST[px * 3 + gi[0]], stB = PST[px * 3 + gi[3]];
    float4 stM = stA;
    if (CG < 4) stM = PST[px * 3 + gi[1]];          // (three channels per group: a quad can touch three groups)
    const uint2 xv = *reinterpret_cast<const uint2*>(XT + px * XS + c0);
    const uint2 dv = *reinterpret_cast<const uint2*>(DT + px * XS + c0);
    const float x[4] = {__uint_as_float(xv.x << 16), __uint_as_float(xv.x & 0xffff0000u), __uint_as_float(xv.y << 16), __uint_as_float(xv.y & 0xffff0000u)};
    const float d[4] = {__uint_as_float(dv.x << 16), __uint_as_float(dv.x & 0xffff0000u), __uint_as_float(dv.y << 16), __uint_as_float(dv.y & 0xffff0000u)};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 st = gi[e] == gi[0] ? stA : (gi[e] == gi[3] ? stB : stM);
      const float dy = FUSE ? fmaf(d[e], sv[e], sg[e]) : d[e];
      const float xh = (x[e] - st.x) * st.y;
      const float pre = cga[e] * xh + cbe[e];
      const float dh = dy * (pre >= 0.f ? 1.f : alpha);
      const float dxh = dh * cga[e];
      const float t = st.y * (dxh - st.z - xh * st.w);
      const bool live = (f & 1u) && c0 + e < C;
      o[e] = live ? t : 0.f;
      if ((f & 2u) && c0 + e < C) { sga[e] = fmaf(dh, xh, sga[e]); sbe[e] += dh; sbi[e] += t; }
    }
    uint2 ov;
    ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
    *reinterpret_cast<uint2*>(XT + px * XS + c0) = ov;
  }
  *reinterpret_cast<float4*>(RED + (cpg * 3 + 0) * CPH + c0) = make_float4(sga[0], sga[1], sga[2], sga[3]);
  *reinterpret_cast<float4*>(RED + (cpg * 3 + 1) * CPH + c0) = make_float4(sbe[0], sbe[1], sbe[2], sbe[3]);
  *reinterpret_cast<float4*>(RED + (cpg * 3 + 2) * CPH + c0) = make_float4(sbi[0], sbi[1], sbi[2], sbi[3]);
}
// (after a barrier) thread tid < CPH adds its channel's partial sums in pixel-group order
template <int CPH, int CG>
__device__ __forceinline__ void lnb_cols_finish(const float* RED, int tid, float* sums) {
  constexpr int PG = LnColCfg<CPH, CG>::PG;
  if (tid < CPH) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float s_ = 0.f;
      for (int gq = 0; gq < PG; ++gq) s_ += RED[(gq * 3 + k) * CPH + tid];
      sums[k] += s_;
    }
  }
}

template <int CIN, int CV11, int CVKK, int OC>
__global__ __launch_bounds__(256) void cardinal_bwd_kernel(const CardBwd p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = CardBwdCfg<CV11, CVKK, OC>;
  constexpr int U = Cfg::U, V = Cfg::V, UP = Cfg::UP, VP = Cfg::VP, S1 = Cfg::S1, S2 = Cfg::S2;
  constexpr int CP1 = UP / 8, CP2 = VP / 8;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int blk = blockIdx.x >> 1;
  if (blockIdx.x & 1) {      // ============================================= shortcut norm backward (ResNest.py:100-101)
    lnb_tile_loop<OC / 8, OC, 1, false>(p.sc, p.sc_tiles, lds, blk, p.G);
    return;
  }
  bf16_t* const VT = reinterpret_cast<bf16_t*>(lds);
  bf16_t* const DT = reinterpret_cast<bf16_t*>(lds + Cfg::OFF_DT);
  bf16_t* const DU = DT;      // (the dout tile is dead once dv exists)
  float4* const PST = reinterpret_cast<float4*>(lds + Cfg::OFF_PST);
  float* const STAT = reinterpret_cast<float*>(lds + Cfg::OFF_STAT);
  float* const RED = reinterpret_cast<float*>(lds + Cfg::OFF_RED);
  float* const SA = reinterpret_cast<float*>(lds + Cfg::OFF_SA);
  unsigned char* const FLG = reinterpret_cast<unsigned char*>(lds + Cfg::OFF_FLG);
  bf16_t* const UT = reinterpret_cast<bf16_t*>(lds + Cfg::OFF_UT);

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  float acc2[3] = {0.f, 0.f, 0.f}, acc1[3] = {0.f, 0.f, 0.f};

  // the grouped 3x3's backward-data operand, block diagonal: wave g < 3 computes path g's CV11 input-channel gradients from its CVKK dv channels;
  // PFB K steps of weight fragments in flight (the loop is bound by the L2 latency of its weight rows: 27 K steps at 28 -> 85 channels)
  constexpr int TPP = (CV11 + 15) / 16, PFB = 8;
  const int pcb = (wv * CVKK) / 8, pnck = ((wv + 1) * CVKK + 7) / 8 - pcb, pn0 = wv * CV11;
  const int pnch = 9 * pnck, KSg = (pnch + 3) / 4;
  auto w2_frag = [&](int ks, int j) -> bf16x8_t {      // (clamped, never selected: see the forward kernel's w2_frag)
    int chunk = 4 * ks + q, row = 16 * j + r;
    chunk = chunk < pnch ? chunk : pnch - 1;
    row = row < CV11 ? row : CV11 - 1;
    const int tap = chunk / pnck, choff = chunk - tap * pnck;
    return ldg_frag(p.w2d + (int64_t)(pn0 + row) * (9 * VP) + tap * VP + (pcb + choff) * 8);
  };

  for (int t = blk; t < p.ntiles; t += p.G) {
    const int b = t / p.tiles_img, tile = t - b * p.tiles_img;
    const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
    const int ty0 = tyi * TILE, tx0 = txi * TILE;
    const int64_t img = (int64_t)b * p.H * p.W;
    CARD_STAMP(0);
    const float *g2 = p.g2, *be2 = p.be2, *g1 = p.g1, *be1 = p.be1;
    bf16x8_t a2[PFB][TPP];
    if (wv < 3) {
#pragma unroll
      for (int s = 0; s < PFB; ++s)
#pragma unroll
        for (int j = 0; j < TPP; ++j) a2[s][j] = w2_frag(s, j);
    }
    // ---- A: v_raw and dout on the halo, u_raw on the interior -> LDS; per-pixel flags (bit 0: inside the image, bit 1: interior too)
    if (tid < 128) {
      const int hy = tid / LW, hx = tid - hy * LW;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      const bool in = tid < NHALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const bool inner = hy >= 1 && hy <= TILE && hx >= 1 && hx <= TILE;
      FLG[tid] = (unsigned char)((in ? 1 : 0) | (in && inner ? 2 : 0));
    }
    auto halo_src = [&](const bf16_t* base, int ld, int hp) -> const bf16_t* {
      const int hy = hp / LW, hx = hp - hy * LW;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      return (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? base + (img + (int64_t)gy * p.W + gx) * ld : nullptr;
    };
    {   // every load of the three tiles is issued before the first LDS store (three load -> store rounds cost three memory latencies)
      constexpr int NV = (NHALO * CP2 + 255) / 256, NU = (NPIX * CP1 + 255) / 256;
      uint4 rv[NV], rd[NV], ru[NU];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int it = tid + i * 256;
        const int hp = it / CP2, c = it - hp * CP2;
        const bf16_t* sv_ = it < NHALO * CP2 ? halo_src(p.v_raw, p.ldv, hp) : nullptr;
        const bf16_t* sd_ = it < NHALO * CP2 ? halo_src(p.dout, p.ldo, hp) : nullptr;
        rv[i] = sv_ ? *reinterpret_cast<const uint4*>(sv_ + c * 8) : make_uint4(0, 0, 0, 0);
        rd[i] = sd_ ? *reinterpret_cast<const uint4*>(sd_ + c * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int it = tid + i * 256;
        const int pp = it / CP1, c = it - pp * CP1;
        const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
        const bool ok = it < NPIX * CP1 && gy < p.H && gx < p.W;
        ru[i] = ok ? *reinterpret_cast<const uint4*>(p.u_raw + (img + (int64_t)gy * p.W + gx) * p.ldu + c * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int it = tid + i * 256;
        const int hp = it / CP2, c = it - hp * CP2;
        if (it < NHALO * CP2) {
          *reinterpret_cast<uint4*>(VT + hp * S2 + c * 8) = rv[i];
          *reinterpret_cast<uint4*>(DT + hp * S2 + c * 8) = rd[i];
        }
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int it = tid + i * 256;
        const int pp = it / CP1, c = it - pp * CP1;
        if (it < NPIX * CP1) *reinterpret_cast<uint4*>(UT + pp * S1 + c * 8) = ru[i];
      }
    }
    if (tid < VP) {
      SA[tid] = tid < V ? p.sa_mult * p.sa_s[(int64_t)b * V + tid] : 0.f;
      SA[VP + tid] = tid < V ? p.sa_dg[(int64_t)b * V + tid] : 0.f;
    }
    __syncthreads();
    CARD_STAMP(1);
    // ---- B: conv2_bn backward, row passes on the 100 halo pixels: rows 0-63, then 64-99 (the four waves split the channels; one code copy
    //         with a quarter of the channels per wave keeps the hoisted gamma / beta / SA constants - and the kernel's VGPR count - small)
#pragma unroll 1
    for (int rb = 0; rb < 2; ++rb) {
      const int row = rb * 64 + lane < NHALO ? rb * 64 + lane : NHALO - 1;
      const bf16_t* xrow = VT + row * S2;
      const bf16_t* dyrow = DT + row * S2;
      float4* pst = PST + rb * 64 * 3;
      if (wv == 0) lnb_rows<0, CP2, CVKK, 3, true>(xrow, dyrow, STAT, pst, g2, be2, SA, p.eps, p.alpha, wv, lane);
      else if (wv == 1) lnb_rows<1, CP2, CVKK, 3, true>(xrow, dyrow, STAT, pst, g2, be2, SA, p.eps, p.alpha, wv, lane);
      else if (wv == 2) lnb_rows<2, CP2, CVKK, 3, true>(xrow, dyrow, STAT, pst, g2, be2, SA, p.eps, p.alpha, wv, lane);
      else lnb_rows<3, CP2, CVKK, 3, true>(xrow, dyrow, STAT, pst, g2, be2, SA, p.eps, p.alpha, wv, lane);
      __syncthreads();      // (STAT is reused by the second pass)
    }
    CARD_STAMP(2);
    // ---- C: column pass: dv (zero outside the image) replaces v_raw in LDS; sums over the interior pixels
    lnb_cols<VP, CVKK, NHALO, true>(VT, DT, S2, PST, SA, FLG, true, RED, g2, be2, p.alpha, tid);
    __syncthreads();
    lnb_cols_finish<VP, CVKK>(RED, tid, acc2);
    CARD_STAMP(3);
    // dv of the interior pixels -> HBM in whole rows (the grouped 3x3's weight gradient reads it)
    for (int it = tid; it < NPIX * CP2; it += 256) {
      const int pp = it / CP2, c = it - pp * CP2;
      const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
      if (gy < p.H && gx < p.W) {
        const int hp = ((pp >> 3) + 1) * LW + (pp & 7) + 1;
        *reinterpret_cast<uint4*>(p.dv + (img + (int64_t)gy * p.W + gx) * p.lddv + c * 8) = *reinterpret_cast<const uint4*>(VT + hp * S2 + c * 8);
      }
    }
    CARD_STAMP(4);
    // ---- D: du[interior pixel][path channels] = sum over taps of dv[pixel - (tap - 1)] . W2[tap]^T (block diagonal)
    {
      f32x4_t acc[4][TPP];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TPP; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (wv < 3) {
        int pbase[4];   // halo index of this lane's pixel shifted by (+1, +1): tap (ty, tx) reads pbase - ty * LW - tx
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int pp = i * 16 + r;
          pbase[i] = ((pp >> 3) + 2) * LW + (pp & 7) + 2;
        }
#pragma unroll 1
        for (int ks0 = 0; ks0 < KSg; ks0 += PFB) {
#pragma unroll
          for (int s = 0; s < PFB; ++s) {
            const int ks = ks0 + s;
            {   // (no `ks < KSg` branch, as in the forward kernel)
              const int chunk = 4 * ks + q;
              const bool kv = chunk < pnch;
              const int tap = kv ? chunk / pnck : 0, choff = kv ? chunk - tap * pnck : 0;
              const int ty = tap / 3, tx = tap - ty * 3;
              const int toff = -(ty * LW + tx) * S2 + (pcb + choff) * 8;
              bf16x8_t bu[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                bu[i] = *reinterpret_cast<const bf16x8_t*>(VT + pbase[i] * S2 + toff);
                if (!kv) bu[i] = zero_frag();
              }
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TPP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s][j], bu[i], acc[i][j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TPP; ++j) a2[s][j] = w2_frag(ks + PFB, j);      // (unconditional, clamped: the vmcnt count stays exact)
            }
          }
        }
        // a path starts at any channel: 2-byte LDS stores (the dout tile's LDS: every thread passed the barrier after the column pass)
#pragma unroll
        for (int j = 0; j < TPP; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int pp = i * 16 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int cl = 16 * j + 4 * q + e;
              if (cl < CV11) DU[pp * S1 + pn0 + cl] = f2bf(acc[i][j][e]);
            }
          }
      } else if (UP > U) {
        for (int c = U; c < UP; ++c) DU[lane * S1 + c] = 0;
      }
    }
    __syncthreads();
    CARD_STAMP(5);
    // ---- E: conv1_bn backward on the 64 interior pixels: row pass, column pass (du_raw replaces u_raw in LDS), out in whole rows
    {
      const bf16_t* xrow = UT + lane * S1;
      const bf16_t* dyrow = DU + lane * S1;
      if (wv == 0) lnb_rows<0, CP1, CV11, 3, false>(xrow, dyrow, STAT, PST, g1, be1, SA, p.eps, p.alpha, wv, lane);
      else if (wv == 1) lnb_rows<1, CP1, CV11, 3, false>(xrow, dyrow, STAT, PST, g1, be1, SA, p.eps, p.alpha, wv, lane);
      else if (wv == 2) lnb_rows<2, CP1, CV11, 3, false>(xrow, dyrow, STAT, PST, g1, be1, SA, p.eps, p.alpha, wv, lane);
      else lnb_rows<3, CP1, CV11, 3, false>(xrow, dyrow, STAT, PST, g1, be1, SA, p.eps, p.alpha, wv, lane);
    }
    __syncthreads();
    CARD_STAMP(6);
    lnb_cols<UP, CV11, NPIX, false>(UT, DU, S1, PST, SA, FLG, false, RED, g1, be1, p.alpha, tid);
    __syncthreads();
    lnb_cols_finish<UP, CV11>(RED, tid, acc1);
    CARD_STAMP(7);
    for (int it = tid; it < NPIX * CP1; it += 256) {
      const int pp = it / CP1, c = it - pp * CP1;
      const int gy = ty0 + (pp >> 3), gx = tx0 + (pp & 7);
      if (gy < p.H && gx < p.W)
        *reinterpret_cast<uint4*>(p.dcat + (img + (int64_t)gy * p.W + gx) * p.ldc + c * 8) = *reinterpret_cast<const uint4*>(UT + pp * S1 + c * 8);
    }
    __syncthreads();      // the next tile's loads overwrite the staged tiles
    CARD_STAMP(8);
  }
  if (tid < VP) {
#pragma unroll
    for (int k = 0; k < 3; ++k) p.ws2[((int64_t)blk * 3 + k) * VP + tid] = acc2[k];
  }
  if (tid < UP) {
#pragma unroll
    for (int k = 0; k < 3; ++k) p.ws1[((int64_t)blk * 3 + k) * UP + tid] = acc1[k];
  }
#endif
}

template <int CIN, int CV11, int CVKK, int OC>
int launch_bwd(CardBwd& p, float* dg2, float* dbe2, float* db2, float* dg1, float* dbe1, float* db1, float* dgsc, float* dbesc, float* dbsc,
               float* caller_ws, hipStream_t s) {
  using Cfg = CardBwdCfg<CV11, CVKK, OC>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)cardinal_bwd_kernel<CIN, CV11, CVKK, OC>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_B);
    attr_done = true;
  }
  // workgroup pairs (cardinal tile walker, shortcut tile walker): as many as are resident at once, at most one per tile
  static const int g_env = getenv("USSEG_CARD_BWD_G") ? atoi(getenv("USSEG_CARD_BWD_G")) : 0;
  const int per_cu = 160 * 1024 / (2 * Cfg::LDS_B) > 0 ? 160 * 1024 / (2 * Cfg::LDS_B) : 1;
  const int most = p.ntiles > p.sc_tiles ? p.ntiles : p.sc_tiles;
  int G = g_env > 0 ? g_env : 256 * per_cu;
  if (G > most) G = most;
  if (G > USSEG_REDUCE_MAX_BLOCKS) G = USSEG_REDUCE_MAX_BLOCKS;
  constexpr int VP = Cfg::VP, UP = Cfg::UP;
  while (G > 1 && (int64_t)G * 3 * (VP + UP + OC) > (int64_t)USSEG_REDUCE_MAX_BLOCKS * 3 * 512) G >>= 1;   // the three partial-row regions fit usseg_reduce_ws_floats()
  p.G = G;
  // three partial-row regions (private ones while the finishing reductions are deferred); caller_ws holds all three otherwise
  const int64_t n2 = (int64_t)G * 3 * VP, n1 = (int64_t)G * 3 * UP, nsc = (int64_t)G * 3 * OC;
  p.ws2 = usseg_defer_reduce_ws(s, caller_ws, n2);
  p.ws1 = usseg_defer_reduce_ws(s, caller_ws + n2, n1);
  p.sc.ws = usseg_defer_reduce_ws(s, caller_ws + n2 + n1, nsc);
  const int slot = usseg_prof_start(4, s);       // timed as a fused tile kernel (kind 4)
  hipLaunchKernelGGL((cardinal_bwd_kernel<CIN, CV11, CVKK, OC>), dim3(2 * G), dim3(256), Cfg::LDS_B, s, p);
  usseg_prof_stop(4, slot, s);
  usseg_launch_reduce_finish(p.ws2, 1, G, 3, VP, Cfg::V, 1.f, dg2, dbe2, db2, s, 0);
  usseg_launch_reduce_finish(p.ws1, 1, G, 3, UP, Cfg::U, 1.f, dg1, dbe1, db1, s, 0);
  usseg_launch_reduce_finish(p.sc.ws, 1, G, 3, OC, OC, 1.f, dgsc, dbesc, dbsc, s, 0);
  return usseg_check_launch("cardinal_bwd");
}

// Two INDEPENDENT LayerNorm backward passes in one launch (even workgroups: A, odd: B): the shortcut norm's and the conv2_bn (+ split-attention
// re-weighting) backward of a residual_S stage read different tensors and both stream at 3.5-4 TB/s alone - side by side they share one
// dispatch (every dispatch that consumes its predecessor's output costs ~5-9 us at batch 16) and fill the memory system better.
struct LnPair {
  LnTileArgs a, b;
  int32_t ntiles_a, ntiles_b, G;
};
template <int CPA, int CGA, int NGA, bool FA, int CPB, int CGB, int NGB, bool FB>
__global__ __launch_bounds__(256) void ln_bwd_pair_kernel(const LnPair P) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int blk = blockIdx.x >> 1;
  if (blockIdx.x & 1) lnb_tile_loop<CPB, CGB, NGB, FB>(P.b, P.ntiles_b, lds, blk, P.G);
  else lnb_tile_loop<CPA, CGA, NGA, FA>(P.a, P.ntiles_a, lds, blk, P.G);
#endif
}
template <int CPA, int CGA, int NGA, bool FA, int CPB, int CGB, int NGB, bool FB>
int launch_ln_pair(LnPair& P, float* const* ga, float* const* gb, float* caller_ws, hipStream_t s) {
  using CA = LnbCfg<CPA, CGA, NGA, FA>;
  using CB = LnbCfg<CPB, CGB, NGB, FB>;
  constexpr int LDS = CA::LDS_B > CB::LDS_B ? CA::LDS_B : CB::LDS_B;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)ln_bwd_pair_kernel<CPA, CGA, NGA, FA, CPB, CGB, NGB, FB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int per_cu = LDS > 80 * 1024 ? 1 : (LDS > 40 * 1024 ? 2 : 4);      // workgroups of EACH role per CU
  const int most = P.ntiles_a > P.ntiles_b ? P.ntiles_a : P.ntiles_b;
  int G = 128 * per_cu;
  if (G > most) G = most;
  if (G > USSEG_REDUCE_MAX_BLOCKS) G = USSEG_REDUCE_MAX_BLOCKS;
  P.G = G;
  const int64_t na = (int64_t)G * 3 * CA::CPH, nb = (int64_t)G * 3 * CB::CPH;
  P.a.ws = usseg_defer_reduce_ws(s, caller_ws, na);
  P.b.ws = usseg_defer_reduce_ws(s, caller_ws + na, nb);
  hipLaunchKernelGGL((ln_bwd_pair_kernel<CPA, CGA, NGA, FA, CPB, CGB, NGB, FB>), dim3(2 * G), dim3(256), LDS, s, P);
  usseg_launch_reduce_finish(P.a.ws, 1, G, 3, CA::CPH, P.a.C, 1.f, ga[0], ga[1], ga[2], s, 0);
  usseg_launch_reduce_finish(P.b.ws, 1, G, 3, CB::CPH, P.b.C, 1.f, gb[0], gb[1], gb[2], s, 0);
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

// A = the shortcut norm (one group of Oc channels), B = conv2_bn with the re-weighting's backward folded in (3 groups of cvkk): the two
// stages whose tensors are large enough for the tile kernels (ResNest.py stages 1 and 2 at 256x256).  0: no instantiation, nothing launched.
int usseg_try_ln_bwd_pair(const LnTileArgs& a, float* const* ga, const LnTileArgs& b, float* const* gb, float* caller_ws, hipStream_t s) {
  static const int off = getenv("USSEG_LN_PAIR") && atoi(getenv("USSEG_LN_PAIR")) == 0;
  static const int64_t min_px = getenv("USSEG_LN_TILE_MIN") ? atoll(getenv("USSEG_LN_TILE_MIN")) : 32768;
  if (off || a.M < min_px || b.M < min_px || a.M >= (1ll << 31) || b.M >= (1ll << 31)) return 0;
  if (a.sa_s || !b.sa_s || b.HW % 64 != 0 || a.G != 1 || b.G != 3) return 0;
  LnPair P;
  P.a = a; P.b = b;
  P.ntiles_a = (int)((a.M + 63) / 64); P.ntiles_b = (int)((b.M + 63) / 64);
  const int cgb = b.C / 3;
  if (a.Cphys == 64 && a.C == 64 && b.Cphys == 32 && cgb == 10 && b.C == 30) return launch_ln_pair<8, 64, 1, false, 4, 10, 3, true>(P, ga, gb, caller_ws, s);
  if (a.Cphys == 128 && a.C == 128 && b.Cphys == 64 && cgb == 21 && b.C == 63) return launch_ln_pair<16, 128, 1, false, 8, 21, 3, true>(P, ga, gb, caller_ws, s);
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

extern "C" int usseg_cardinal_bwd(const UssegCardinalDesc* d, const void* dout, int32_t ldo, const void* dsc, int32_t lddsc, const void* v_raw,
                                  const void* u_raw, const void* sc_raw, const void* w2d, const float* g2, const float* be2, const float* g1,
                                  const float* be1, const float* gsc, const float* besc, const float* sa_s, const float* sa_dg, float sa_mult,
                                  void* dv, void* dcat, int32_t ldc, float* dg2, float* dbe2, float* db2, float* dg1, float* dbe1, float* db1,
                                  float* dgsc, float* dbesc, float* dbsc, float* ws, usseg_stream_t stream) {
  const int cfg = card_config(d);
  USSEG_CHECK_ARG(cfg >= 0, "cardinal_bwd: no fused kernel for this channel configuration (usseg_cardinal_supported)");
  USSEG_CHECK_ARG(dout && dsc && v_raw && u_raw && sc_raw && w2d && g2 && be2 && g1 && be1 && gsc && besc && sa_s && sa_dg && dv && dcat && dg2 && dbe2 &&
                      db2 && dg1 && dbe1 && db1 && dgsc && dbesc && dbsc && ws,
                  "cardinal_bwd: null pointer");
  USSEG_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && ldo >= d->Vp && lddsc >= d->Oc && d->ldu >= d->Up && d->ldv >= d->Vp && d->ldsc >= d->Oc &&
                      ldc >= d->Up + d->Oc && ldo % 8 == 0 && lddsc % 8 == 0 && d->ldu % 8 == 0 && d->ldv % 8 == 0 && d->ldsc % 8 == 0 && ldc % 8 == 0,
                  "cardinal_bwd: bad geometry / strides");
  USSEG_CHECK_ARG((int64_t)d->B * d->H * d->W < (1ll << 31), "cardinal_bwd: too many pixels");
  CardBwd p = {};
  p.dout = (const bf16_t*)dout; p.v_raw = (const bf16_t*)v_raw; p.u_raw = (const bf16_t*)u_raw; p.w2d = (const bf16_t*)w2d;
  p.g2 = g2; p.be2 = be2; p.g1 = g1; p.be1 = be1; p.sa_s = sa_s; p.sa_dg = sa_dg; p.sa_mult = sa_mult;
  p.dv = (bf16_t*)dv; p.dcat = (bf16_t*)dcat;
  p.B = d->B; p.H = d->H; p.W = d->W; p.ldo = ldo; p.ldv = d->ldv; p.ldu = d->ldu; p.lddv = d->ldv; p.ldc = ldc;
  p.tiles_x = (d->W + TILE - 1) / TILE;
  p.tiles_img = p.tiles_x * ((d->H + TILE - 1) / TILE);
  p.ntiles = p.tiles_img * d->B;
  p.eps = d->eps; p.alpha = d->alpha;
  LnTileArgs& t = p.sc;
  t.x = (const bf16_t*)sc_raw; t.dy = (const bf16_t*)dsc; t.dx = (bf16_t*)dcat + d->Up; t.gamma = gsc; t.beta = besc;
  t.M = (int64_t)d->B * d->H * d->W; t.HW = t.M; t.C = d->Oc; t.Cphys = d->Oc; t.G = 1;
  t.ldx = d->ldsc; t.lddy = lddsc; t.lddx = ldc; t.eps = d->eps; t.alpha = d->alpha;
  p.sc_tiles = (int)((t.M + 63) / 64);
  hipStream_t s = (hipStream_t)stream;
  switch (cfg) {
    case 0: return launch_bwd<32, 3, 10, 64>(p, dg2, dbe2, db2, dg1, dbe1, db1, dgsc, dbesc, dbsc, ws, s);
    case 1: return launch_bwd<64, 7, 21, 128>(p, dg2, dbe2, db2, dg1, dbe1, db1, dgsc, dbesc, dbsc, ws, s);
    case 2: return launch_bwd<128, 14, 42, 256>(p, dg2, dbe2, db2, dg1, dbe1, db1, dgsc, dbesc, dbsc, ws, s);
    default: return launch_bwd<256, 28, 85, 512>(p, dg2, dbe2, db2, dg1, dbe1, db1, dgsc, dbesc, dbsc, ws, s);
  }
}
