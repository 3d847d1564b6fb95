// Kernels of the windowed-attention encoder (the reference's SwinTransformer.py, BASELINE configs[4]) that the conv / norm kernels
// of the ResNeSt path do not cover.  Tokens live as NHWC bf16 tensors [B][H][W][C] (the reference's [B, L = H*W, C] row-major).
//
//   usseg_patchify          PatchEmbed's Conv2D(kernel = stride = patch) (SwinTransformer.py:352-353) = space-to-depth of the input
//                           image + a 1x1 GEMM: this is the space-to-depth (+ cast to bf16), channel = (ph*patch + pw)*C + c, which is
//                           exactly the row order of the Keras kernel [patch][patch][C][E] seen as [patch*patch*C][E].
//   usseg_patch_merge       PatchMerging's strided gather + concat (:280-284) and its backward scatter.
//   usseg_ln_wide_fwd/bwd   LayerNormalization over up to 4096 channels (the 4C = 1536 / 3072-wide norms of PatchMerging; the per-pixel
//                           norm kernel of pointwise.hip stops at 512): one wave per token, the token's channels in registers.
//   usseg_window_attn_fwd/bwd  W-MSA / SW-MSA (:101-141,219-261) for one (image, window, head) per wave: the cyclic shift, the window
//                           partition and their inverses are INDEX ARITHMETIC on the token tensor (no rolled / partitioned copies), the
//                           relative-position bias and the shifted-window mask are added on the fly, softmax in registers.  Windows are
//                           16 or 64 tokens of 32-wide heads: a 64x64x32 product per wave is far below an MFMA tile pipeline's
//                           break-even, so the products run as fp32 FMAs from LDS (HBM-bound: each token's q, k, v is read once).
//   usseg_token_mean_fwd/bwd   GlobalAveragePooling1D over the tokens (:451).
#include "common.h"

static inline unsigned sw_grid(int64_t work, int per_block, int cap = 4096) {
  int64_t g = (work + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ------------------------------------------------------------------------------------------ patchify
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* x, int B, int H, int W, int C, int ps, bf16_t* out, int Cp) {
  const int Ho = H / ps, Wo = W / ps, CH = Cp / 8, K = ps * ps * C;
  const int64_t total = (int64_t)B * Ho * Wo * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i / CH;
    const int k0 = (int)(i - pix * CH) * 8;
    const int ox = (int)(pix % Wo);
    const int64_t t = pix / Wo;
    const int oy = (int)(t % Ho);
    const int64_t b = t / Ho;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float f = 0.f;
      if (k < K) {
        const int c = k % C, pq = k / C, pw = pq % ps, ph = pq / ps;
        f = (float)x[((b * H + oy * ps + ph) * W + ox * ps + pw) * C + c];
      }
      v[j] = f;
      asm volatile("" : "+v"(v[j]));     // the float32 value must exist before the bf16 rounding (see cast_input_kernel)
    }
    *reinterpret_cast<uint4*>(out + pix * Cp + k0) = pack8(v);
  }
}
extern "C" int usseg_patchify(const void* x, int32_t x_is_f64, int32_t B, int32_t H, int32_t W, int32_t C, int32_t patch, void* out, int32_t Cp,
                              usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && out && B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0 && Cp % 8 == 0 && Cp >= patch * patch * C,
                  "patchify: bad geometry");
  const int64_t total = (int64_t)B * (H / patch) * (W / patch) * (Cp / 8);
  if (x_is_f64)
    hipLaunchKernelGGL(patchify_kernel<double>, dim3(sw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)x, B, H, W, C, patch,
                       (bf16_t*)out, Cp);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(sw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, B, H, W, C, patch,
                       (bf16_t*)out, Cp);
  return usseg_check_launch("patchify");
}

// ------------------------------------------------------------------------------------------ patch merging
// forward: out[b][i][j][(a + 2*bb)*C + c] = x[b][2i + a][2j + bb][c]   (x0 = (0,0), x1 = (1,0), x2 = (0,1), x3 = (1,1), :280-284)
// backward: the same index map with the roles of source and destination exchanged.
__global__ __launch_bounds__(256) void patch_merge_kernel(bf16_t* full, int B, int Ho, int Wo, int C, int ldf, bf16_t* merged, int ldm, int backward) {
  const int CH = C / 8;
  const int64_t total = (int64_t)B * Ho * Wo * 4 * CH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % CH);
    int64_t t = i / CH;
    const int blk = (int)(t % 4);
    t /= 4;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int64_t b = t / Ho;
    const int a = blk & 1, bb = blk >> 1;
    bf16_t* pf = full + ((b * 2 * Ho + 2 * oy + a) * (2 * Wo) + 2 * ox + bb) * (int64_t)ldf + ch * 8;
    bf16_t* pm = merged + ((b * Ho + oy) * (int64_t)Wo + ox) * ldm + blk * C + ch * 8;
    if (backward) *reinterpret_cast<uint4*>(pf) = *reinterpret_cast<const uint4*>(pm);
    else *reinterpret_cast<uint4*>(pm) = *reinterpret_cast<const uint4*>(pf);
  }
}
extern "C" int usseg_patch_merge(void* full, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldf, void* merged, int32_t ldm, int32_t backward,
                                 usseg_stream_t stream) {
  USSEG_CHECK_ARG(full && merged && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && ldf % 8 == 0 && ldm % 8 == 0 && ldf >= C && ldm >= 4 * C,
                  "patch_merge: bad geometry");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * 4 * (C / 8);
  hipLaunchKernelGGL(patch_merge_kernel, dim3(sw_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)full, B, H / 2, W / 2, C, ldf,
                     (bf16_t*)merged, ldm, backward);
  return usseg_check_launch("patch_merge");
}

// ------------------------------------------------------------------------------------------ wide LayerNorm
// One wave per token; lane l holds the 8-channel chunks l, l + 64, ... (NK of them: C <= 512 * NK).
struct LnWide {
  const bf16_t* x; const bf16_t* dy; bf16_t* y; bf16_t* dx;
  const float *gamma, *beta;
  float* ws;           // backward: per-workgroup partial rows [grid][2][C]
  int64_t M;
  int32_t C, ldx, ldy, lddy, lddx;
  float eps;
};
template <int NK, bool BWD>
__global__ __launch_bounds__(256) void ln_wide_kernel(const LnWide p) {
  extern __shared__ float s_ln[];        // backward: [4 waves][C]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int CH = p.C >> 3;
  float ga[NK][8], be[NK][8], dga[NK][8], dbe[NK][8];
  bool ok[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    ok[k] = lane + 64 * k < CH;
    const int c0 = ok[k] ? (lane + 64 * k) * 8 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ga[k][j] = ok[k] ? p.gamma[c0 + j] : 0.f;
      be[k][j] = (!BWD && ok[k]) ? p.beta[c0 + j] : 0.f;
      dga[k][j] = 0.f; dbe[k][j] = 0.f;
    }
  }
  const float inv_c = 1.f / (float)p.C;
  for (int64_t m = (int64_t)blockIdx.x * 4 + wv; m < p.M; m += (int64_t)gridDim.x * 4) {
    float xv[NK][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (ok[k]) raw = *reinterpret_cast<const uint4*>(p.x + m * p.ldx + (lane + 64 * k) * 8);
      unpack8(raw, xv[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[k][j];
    }
    for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 64);
    const float mean = s * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = ok[k] ? xv[k][j] - mean : 0.f;
        xv[k][j] = d;
        ss = fmaf(d, d, ss);
      }
    for (int msk = 32; msk >= 1; msk >>= 1) ss += __shfl_xor(ss, msk, 64);
    const float rstd = rsqrtf(ss * inv_c + p.eps);
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(xv[k][j] * rstd, ga[k][j], be[k][j]);
        if (ok[k]) *reinterpret_cast<uint4*>(p.y + m * p.ldy + (lane + 64 * k) * 8) = pack8(o);
      }
    } else {
      float s1 = 0.f, s2 = 0.f;
      float g[NK][8];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float dyv[8];
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (ok[k]) raw = *reinterpret_cast<const uint4*>(p.dy + m * p.lddy + (lane + 64 * k) * 8);
        unpack8(raw, dyv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = xv[k][j] * rstd;
          xv[k][j] = xh;
          dga[k][j] = fmaf(dyv[j], xh, dga[k][j]);
          dbe[k][j] += dyv[j];
          g[k][j] = dyv[j] * ga[k][j];
          s1 += g[k][j];
          s2 = fmaf(g[k][j], xh, s2);
        }
      }
      for (int msk = 32; msk >= 1; msk >>= 1) { s1 += __shfl_xor(s1, msk, 64); s2 += __shfl_xor(s2, msk, 64); }
      s1 *= inv_c; s2 *= inv_c;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[k][j] - s1 - xv[k][j] * s2);
        if (ok[k]) *reinterpret_cast<uint4*>(p.dx + m * p.lddx + (lane + 64 * k) * 8) = pack8(o);
      }
    }
  }
  if (BWD) {
    // partial rows of dgamma / dbeta: the four waves' register sums meet in LDS, one row per workgroup (finished by reduce_finish)
    float* row = p.ws + (int64_t)blockIdx.x * 2 * p.C;
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (ok[k])
#pragma unroll
          for (int j = 0; j < 8; ++j) s_ln[wv * p.C + (lane + 64 * k) * 8 + j] = pass == 0 ? dga[k][j] : dbe[k][j];
      __syncthreads();
      for (int c = threadIdx.x; c < p.C; c += 256) row[pass * p.C + c] = (s_ln[c] + s_ln[p.C + c]) + (s_ln[2 * p.C + c] + s_ln[3 * p.C + c]);
    }
  }
}
template <bool BWD>
static void ln_wide_launch(const LnWide& p, unsigned grid, hipStream_t s) {
  const int nk = (p.C / 8 + 63) / 64;
  const size_t dyn = BWD ? (size_t)4 * p.C * sizeof(float) : 0;
#define LNW(NK_)                                                                                                             \
  {                                                                                                                          \
    if (BWD && dyn > 48 * 1024) (void)hipFuncSetAttribute((const void*)ln_wide_kernel<NK_, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
    hipLaunchKernelGGL((ln_wide_kernel<NK_, BWD>), dim3(grid), dim3(256), dyn, s, p);                                        \
  }
  if (nk <= 1) LNW(1) else if (nk <= 2) LNW(2) else if (nk <= 4) LNW(4) else LNW(8)
#undef LNW
}
extern "C" int usseg_ln_wide_fwd(const void* x, int64_t M, int32_t C, int32_t ldx, const float* gamma, const float* beta, float eps, void* y,
                                 int32_t ldy, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && y && gamma && beta && C > 0 && C % 8 == 0 && C <= 4096 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C,
                  "ln_wide_fwd: C must be a multiple of 8, <= 4096");
  if (M <= 0) return USSEG_OK;
  LnWide p = {};
  p.x = (const bf16_t*)x; p.y = (bf16_t*)y; p.gamma = gamma; p.beta = beta; p.M = M; p.C = C; p.ldx = ldx; p.ldy = ldy; p.eps = eps;
  ln_wide_launch<false>(p, sw_grid(M, 4 * 2, 4096), (hipStream_t)stream);
  return usseg_check_launch("ln_wide_fwd");
}
extern "C" int usseg_ln_wide_bwd(const void* x, const void* dy, int64_t M, int32_t C, int32_t ldx, int32_t lddy, const float* gamma, float eps,
                                 void* dx, int32_t lddx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && dy && dx && gamma && dgamma && dbeta && ws && C > 0 && C % 8 == 0 && C <= 4096 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0,
                  "ln_wide_bwd: bad arguments");
  if (M <= 0) return USSEG_OK;
  int64_t grid = sw_grid(M, 4 * 4, 512);
  while (grid > 1 && grid * 2 * C > ws_floats) grid >>= 1;
  USSEG_CHECK_ARG(grid * 2 * C <= ws_floats, "ln_wide_bwd: workspace too small (2*C floats per workgroup)");
  LnWide p = {};
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dx = (bf16_t*)dx; p.gamma = gamma; p.M = M; p.C = C; p.ldx = ldx; p.lddy = lddy; p.lddx = lddx;
  p.eps = eps;
  p.ws = usseg_defer_reduce_ws((hipStream_t)stream, ws, grid * 2 * C);
  ln_wide_launch<true>(p, (unsigned)grid, (hipStream_t)stream);
  usseg_launch_reduce_finish(p.ws, 1, (int)grid, 2, C, C, 1.f, dgamma, dbeta, nullptr, (hipStream_t)stream);
  return usseg_check_launch("ln_wide_bwd");
}

// ------------------------------------------------------------------------------------------ window attention
struct WinAttn {
  const bf16_t* qkv;      // [B][H][W][3*C]: q | k | v, each [heads][d]
  const bf16_t* dout;     // backward: [B][H][W][C]
  bf16_t* out;            // forward: [B][H][W][C];  backward: dqkv [B][H][W][3*C]
  const float* table;     // [(2*ws-1)^2][heads]
  float* dtab_ws;         // backward: partial rows [B*nW][(2*ws-1)^2 * heads]
  int32_t B, H, W, C, heads, d, ws, shift, ldq, ldo;
  float scale;
};
#define WA_MAXN 64
#define WA_MAXD 64
// token t = (r, c) of window (wy, wx) sits at the SOURCE position ((wy*ws + r + shift) mod H, (wx*ws + c + shift) mod W):
// torch.roll(x, -shift) followed by window_partition; the same map scatters the output back (window_reverse + roll(+shift)).
__device__ __forceinline__ int64_t wa_token(const WinAttn& p, int b, int wy, int wx, int t) {
  const int r = t / p.ws, c = t - r * p.ws;
  int y = wy * p.ws + r + p.shift, x = wx * p.ws + c + p.shift;
  if (y >= p.H) y -= p.H;
  if (x >= p.W) x -= p.W;
  return ((int64_t)b * p.H + y) * p.W + x;
}
// region id of a token in the SHIFTED image (SwinTransformer.py:195-205): 3 bands per axis
__device__ __forceinline__ int wa_region(const WinAttn& p, int wy, int wx, int t) {
  const int r = t / p.ws, c = t - r * p.ws;
  const int y = wy * p.ws + r, x = wx * p.ws + c;
  const int hy = y < p.H - p.ws ? 0 : (y < p.H - p.shift ? 1 : 2);
  const int hx = x < p.W - p.ws ? 0 : (x < p.W - p.shift ? 1 : 2);
  return hy * 3 + hx;
}

// WS = window side (2, 4 or 8 -> N = 4, 16 or 64 tokens): compile-time so that the score row lives in registers.
template <bool BWD, int WS>
__global__ __launch_bounds__(64) void window_attn_kernel(const WinAttn p) {
  constexpr int N = WS * WS, T = 2 * WS - 1;
  extern __shared__ float s_wa[];
  const int d = p.d, DS = d + 1;
  float* s_q = s_wa;                       // [N][DS]  (q pre-scaled)
  float* s_k = s_q + N * DS;
  float* s_v = s_k + N * DS;
  float* s_do = s_v + N * DS;              // backward only
  float* s_p = s_do + (BWD ? N * DS : 0);  // [N][N+1]
  float* s_ds = s_p + (BWD ? N * (N + 1) : 0);
  __shared__ int s_reg[N];
  const int nwx = p.W / WS, nW = (p.H / WS) * nwx;
  const int h = blockIdx.y, b = blockIdx.z;
  const int w = blockIdx.x, wy = w / nwx, wx = w - wy * nwx;
  const int tid = threadIdx.x;
  // ---- stage q (pre-scaled), k, v (and dout) of the window's tokens: thread = (token, 8-channel chunk)
  const int CHd = d / 8;
  for (int i = tid; i < N * CHd; i += 64) {
    const int t = i / CHd, ch = i - t * CHd;
    const int64_t tok = wa_token(p, b, wy, wx, t);
    const bf16_t* src = p.qkv + tok * p.ldq + h * d + ch * 8;
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(src), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s_q[t * DS + ch * 8 + j] = f[j] * p.scale;
    unpack8(*reinterpret_cast<const uint4*>(src + p.C), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s_k[t * DS + ch * 8 + j] = f[j];
    unpack8(*reinterpret_cast<const uint4*>(src + 2 * p.C), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s_v[t * DS + ch * 8 + j] = f[j];
    if (BWD) {
      unpack8(*reinterpret_cast<const uint4*>(p.dout + tok * p.ldo + h * d + ch * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s_do[t * DS + ch * 8 + j] = f[j];
    }
  }
  if (tid < N) s_reg[tid] = p.shift > 0 ? wa_region(p, wy, wx, tid) : 0;
  __syncthreads();
  const int i = tid;                  // this thread's query row
  float prow[N];
  if (i < N) {
    const int ri = i / WS, ci = i - ri * WS, regi = s_reg[i];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float s = 0.f;
      for (int dd = 0; dd < d; ++dd) s = fmaf(s_q[i * DS + dd], s_k[j * DS + dd], s);
      const int rj = j / WS, cj = j - rj * WS;
      s += p.table[((ri - rj + WS - 1) * T + (ci - cj + WS - 1)) * p.heads + h];           // :94-104
      if (s_reg[j] != regi) s += -100.f;                                                     // :106-110
      prow[j] = s;
      mx = fmaxf(mx, s);
    }
    float rsum = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) { prow[j] = __expf(prow[j] - mx); rsum += prow[j]; }
    const float inv = 1.f / rsum;
#pragma unroll
    for (int j = 0; j < N; ++j) prow[j] *= inv;
  }
  if (!BWD) {
    if (i < N) {
      const int64_t tok = wa_token(p, b, wy, wx, i);
      bf16_t* dst = p.out + tok * p.ldo + h * d;
      for (int c8 = 0; c8 < d; c8 += 8) {
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float pj = prow[j];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, s_v[j * DS + c8 + e], o[e]);
        }
        *reinterpret_cast<uint4*>(dst + c8) = pack8(o);
      }
    }
    return;
  }
  // ---- backward
  if (i < N) {
    float delta = 0.f;
    float dp[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float s = 0.f;
      for (int dd = 0; dd < d; ++dd) s = fmaf(s_do[i * DS + dd], s_v[j * DS + dd], s);
      dp[j] = s;
      delta = fmaf(prow[j], s, delta);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      dp[j] = prow[j] * (dp[j] - delta);          // dS_ij
      s_p[i * (N + 1) + j] = prow[j];
      s_ds[i * (N + 1) + j] = dp[j];
    }
    // dq_i = scale * sum_j dS_ij k_j
    const int64_t tok = wa_token(p, b, wy, wx, i);
    bf16_t* dst = p.out + tok * p.ldq + h * d;
    for (int c8 = 0; c8 < d; c8 += 8) {
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(dp[j], s_k[j * DS + c8 + e], o[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= p.scale;
      *reinterpret_cast<uint4*>(dst + c8) = pack8(o);
    }
  }
  __syncthreads();
  if (tid < N) {   // thread = key / value row j: dk_j = sum_i dS_ij (scale*q_i) ; dv_j = sum_i P_ij dO_i
    const int j = tid;
    const int64_t tok = wa_token(p, b, wy, wx, j);
    bf16_t* dst = p.out + tok * p.ldq + h * d;
    for (int c8 = 0; c8 < d; c8 += 8) {
      float ok_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ov[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < N; ++r) {
        const float ds = s_ds[r * (N + 1) + j], pr = s_p[r * (N + 1) + j];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ok_[e] = fmaf(ds, s_q[r * DS + c8 + e], ok_[e]); ov[e] = fmaf(pr, s_do[r * DS + c8 + e], ov[e]); }
      }
      *reinterpret_cast<uint4*>(dst + p.C + c8) = pack8(ok_);
      *reinterpret_cast<uint4*>(dst + 2 * p.C + c8) = pack8(ov);
    }
  }
  // relative-position bias gradient: bin (dr, dc) collects dS_ij over the pairs with r_i - r_j = dr, c_i - c_j = dc
  float* row = p.dtab_ws + ((int64_t)b * nW + w) * (T * T) * p.heads;
  for (int bin = tid; bin < T * T; bin += 64) {
    const int dr = bin / T - (WS - 1), dc = bin % T - (WS - 1);
    float s = 0.f;
    for (int ii = 0; ii < N; ++ii) {
      const int ri = ii / WS, ci = ii - ri * WS;
      const int rj = ri - dr, cj = ci - dc;
      if ((unsigned)rj < (unsigned)WS && (unsigned)cj < (unsigned)WS) s += s_ds[ii * (N + 1) + rj * WS + cj];
    }
    row[bin * p.heads + h] = s;
  }
}
template <bool BWD>
static int wa_launch(const WinAttn& p, hipStream_t s) {
  const int N = p.ws * p.ws;
  const size_t dyn = sizeof(float) * ((size_t)(BWD ? 4 : 3) * N * (p.d + 1) + (BWD ? 2 * (size_t)N * (N + 1) : 0));
  const dim3 grid((p.H / p.ws) * (p.W / p.ws), p.heads, p.B);
#define WAL(WS_)                                                                                                                     {                                                                                                                                    if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void*)window_attn_kernel<BWD, WS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);     hipLaunchKernelGGL((window_attn_kernel<BWD, WS_>), grid, dim3(64), dyn, s, p);                                                    }
  if (p.ws == 2) WAL(2) else if (p.ws == 4) WAL(4) else if (p.ws == 8) WAL(8) else return 0;
#undef WAL
  return 1;
}

static int wa_common(WinAttn& p, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift) {
  USSEG_CHECK_ARG(B > 0 && heads > 0 && C % heads == 0 && ws > 0 && H % ws == 0 && W % ws == 0, "window_attn: bad geometry");
  USSEG_CHECK_ARG((ws == 2 || ws == 4 || ws == 8) && (C / heads) % 8 == 0 && C / heads <= WA_MAXD, "window_attn: window side 2, 4 or 8; head dim a multiple of 8 and <= 64");
  USSEG_CHECK_ARG(shift >= 0 && shift < ws, "window_attn: 0 <= shift < window");
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.d = C / heads; p.ws = ws; p.shift = shift;
  p.scale = 1.f / sqrtf((float)(C / heads));         // head_dim ** -0.5 (:68)
  return USSEG_OK;
}
extern "C" int usseg_window_attn_fwd(const void* qkv, int32_t ldq, const float* table, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                                     int32_t ws, int32_t shift, void* out, int32_t ldo, usseg_stream_t stream) {
  WinAttn p = {};
  int rc = wa_common(p, B, H, W, C, heads, ws, shift);
  if (rc) return rc;
  USSEG_CHECK_ARG(qkv && table && out && ldq % 8 == 0 && ldo % 8 == 0 && ldq >= 3 * C && ldo >= C, "window_attn_fwd: bad pointers / strides");
  p.qkv = (const bf16_t*)qkv; p.table = table; p.out = (bf16_t*)out; p.ldq = ldq; p.ldo = ldo;
  (void)wa_launch<false>(p, (hipStream_t)stream);
  return usseg_check_launch("window_attn_fwd");
}
extern "C" int64_t usseg_window_attn_bwd_ws_floats(int32_t B, int32_t H, int32_t W, int32_t heads, int32_t ws) {
  return (int64_t)B * (H / ws) * (W / ws) * (2 * ws - 1) * (2 * ws - 1) * heads;
}
extern "C" int usseg_window_attn_bwd(const void* qkv, int32_t ldq, const void* dout, int32_t ldo, const float* table, int32_t B, int32_t H, int32_t W,
                                     int32_t C, int32_t heads, int32_t ws, int32_t shift, void* dqkv, float* dtable, float* ws_rows,
                                     usseg_stream_t stream) {
  WinAttn p = {};
  int rc = wa_common(p, B, H, W, C, heads, ws, shift);
  if (rc) return rc;
  USSEG_CHECK_ARG(qkv && dout && table && dqkv && dtable && ws_rows && ldq % 8 == 0 && ldo % 8 == 0, "window_attn_bwd: bad pointers / strides");
  p.qkv = (const bf16_t*)qkv; p.dout = (const bf16_t*)dout; p.table = table; p.out = (bf16_t*)dqkv; p.ldq = ldq; p.ldo = ldo;
  const int nW = (H / ws) * (W / ws), bins = (2 * ws - 1) * (2 * ws - 1) * heads;
  p.dtab_ws = usseg_defer_reduce_ws((hipStream_t)stream, ws_rows, (int64_t)B * nW * bins);
  (void)wa_launch<true>(p, (hipStream_t)stream);
  usseg_launch_reduce_finish(p.dtab_ws, 1, B * nW, 1, bins, bins, 1.f, dtable, nullptr, nullptr, (hipStream_t)stream);
  return usseg_check_launch("window_attn_bwd");
}

// ------------------------------------------------------------------------------------------ token mean (GlobalAveragePooling1D)
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const bf16_t* x, int L, int C, int ld, float* out) {
  const int b = blockIdx.y;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    float s0 = 0.f, s1 = 0.f;
    const bf16_t* col = x + (int64_t)b * L * ld + c;
    int t = 0;
    for (; t + 1 < L; t += 2) { s0 += bf2f(col[(int64_t)t * ld]); s1 += bf2f(col[(int64_t)(t + 1) * ld]); }
    if (t < L) s0 += bf2f(col[(int64_t)t * ld]);
    out[(int64_t)b * C + c] = (s0 + s1) / (float)L;
  }
}
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* dy, int B, int L, int C, int ld, bf16_t* dx) {
  const int CH = C / 8;
  const int64_t total = (int64_t)B * L * CH;
  const float inv = 1.f / (float)L;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % CH);
    const int64_t tok = i / CH;
    const int64_t b = tok / L;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = dy[b * C + ch * 8 + j] * inv;
    *reinterpret_cast<uint4*>(dx + tok * ld + ch * 8) = pack8(o);
  }
}
extern "C" int usseg_token_mean_fwd(const void* x, int32_t B, int32_t L, int32_t C, int32_t ld, float* out, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && out && B > 0 && L > 0 && C > 0 && ld >= C, "token_mean_fwd: bad arguments");
  hipLaunchKernelGGL(token_mean_fwd_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, L, C, ld, out);
  return usseg_check_launch("token_mean_fwd");
}
extern "C" int usseg_token_mean_bwd(const float* dy, int32_t B, int32_t L, int32_t C, int32_t ld, void* dx, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dy && dx && B > 0 && L > 0 && C % 8 == 0 && ld % 8 == 0 && ld >= C, "token_mean_bwd: bad arguments");
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(sw_grid((int64_t)B * L * (C / 8), 256 * 2)), dim3(256), 0, (hipStream_t)stream, dy, B, L, C, ld,
                     (bf16_t*)dx);
  return usseg_check_launch("token_mean_bwd");
}
