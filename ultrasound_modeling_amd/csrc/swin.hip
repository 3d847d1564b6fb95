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
// ---- MFMA form for 8x8 windows with head size 32 (every stage of the swin_* configurations): one wave per (image, window, head).
// q, k, v (and dO) of the window's 64 tokens are staged once into LDS as [64][32] bf16 tiles (row stride 48 elements: the row reads
// ds_read_b128 and the transposed reads ds_read_b64_tr_b16 are both conflict-free).  The score tile is computed TRANSPOSED
// (S^T = K Q^T: rows keys, columns queries) so that its accumulator layout is the B-operand layout of the second GEMM; MFMA row i of
// key tile t is key 32*(t>>1) + 8*(i>>2) + 4*(t&1) + (i&3), which makes two tiles fill the 32 contraction slots in natural order.
// Relative-position bias, shifted-window mask and the softmax run on the accumulators (a query column lives in 4 lanes: two
// shuffles per reduction).  Probabilities and dS enter the second GEMMs as hi + lo bf16 pairs (relative error 2^-17), so the
// results match the fp32 scalar kernel to rounding.  Backward: P^T, then dS^T, are parked in ONE fp32 [key][query] LDS matrix from
// which the dV / dK GEMMs read their B operands (8 consecutive queries per lane); dS^T summed over the block's windows stays in
// registers and goes through the same matrix once at the end, where the bias-table bins are summed in a fixed order (42 KB of LDS).
#define WM_VS 48
#define WM_PS 68
__device__ __forceinline__ bf16x8_t wm_row(const bf16_t* tile, int row, int g) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * WM_VS + 8 * g);
}
// A operand [free = 16*dt + li][slot 8g + j] = tile[32c + 8g + j][16*dt + li]
__device__ __forceinline__ bf16x8_t wm_tr(const bf16_t* tile, int c, int dt, int g, int tq, int tp) {
  const bf16_t* a0 = tile + (32 * c + 8 * g + tq) * WM_VS + 16 * dt + 4 * tp;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * WM_VS));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void wm_split(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e] - bf2f(f2bf(v[e]));
  u32x4_t h = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  u32x4_t l = {pack2bf(r[0], r[1]), pack2bf(r[2], r[3]), pack2bf(r[4], r[5]), pack2bf(r[6], r[7])};
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, l);
}
__device__ __forceinline__ int wm_tile_row(int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }

// the window's rows as 16 named registers (an aggregate carried round the window loop is kept in scratch memory by the compiler)
#define WM_FETCH(w_)                                                                                   \
  {                                                                                                    \
    const int fy_ = (w_) / nwx, fx_ = (w_) - fy_ * nwx;                                                \
    const int64_t ft_ = wa_token(p, b, fy_, fx_, lane);                                                \
    const bf16_t* fs_ = p.qkv + ft_ * p.ldq + h * 32;                                                  \
    rq0 = *reinterpret_cast<const uint4*>(fs_); rq1 = *reinterpret_cast<const uint4*>(fs_ + 8);        \
    rq2 = *reinterpret_cast<const uint4*>(fs_ + 16); rq3 = *reinterpret_cast<const uint4*>(fs_ + 24);  \
    fs_ += p.C;                                                                                        \
    rk0 = *reinterpret_cast<const uint4*>(fs_); rk1 = *reinterpret_cast<const uint4*>(fs_ + 8);        \
    rk2 = *reinterpret_cast<const uint4*>(fs_ + 16); rk3 = *reinterpret_cast<const uint4*>(fs_ + 24);  \
    fs_ += p.C;                                                                                        \
    rv0 = *reinterpret_cast<const uint4*>(fs_); rv1 = *reinterpret_cast<const uint4*>(fs_ + 8);        \
    rv2 = *reinterpret_cast<const uint4*>(fs_ + 16); rv3 = *reinterpret_cast<const uint4*>(fs_ + 24);  \
    if (BWD) {                                                                                         \
      const bf16_t* fd_ = p.dout + ft_ * p.ldo + h * 32;                                               \
      rd0 = *reinterpret_cast<const uint4*>(fd_); rd1 = *reinterpret_cast<const uint4*>(fd_ + 8);      \
      rd2 = *reinterpret_cast<const uint4*>(fd_ + 16); rd3 = *reinterpret_cast<const uint4*>(fd_ + 24);\
    }                                                                                                  \
  }
#define WM_PUT(tile_, r0_, r1_, r2_, r3_)                                  \
  {                                                                        \
    uint4* d_ = reinterpret_cast<uint4*>(&tile_[lane * WM_VS]);            \
    d_[0] = r0_; d_[1] = r1_; d_[2] = r2_; d_[3] = r3_;                    \
  }

template <bool BWD, bool ONE>
__global__ __launch_bounds__(64) void window_attn_mfma_kernel(const WinAttn p, const int wpb_) {
  const int wpb = ONE ? 1 : wpb_;
  constexpr int WS = 8, N = 64, T = 15;
  __shared__ __attribute__((aligned(16))) bf16_t s_q[N * WM_VS], s_k[N * WM_VS], s_v[N * WM_VS];
  __shared__ __attribute__((aligned(16))) bf16_t s_do[BWD ? N * WM_VS : 8];
  __shared__ __attribute__((aligned(16))) float s_mT[BWD ? N * WM_PS : 4];   // [key][query]: P^T, then dS^T, at the end the summed dS^T
  __shared__ float s_tab[T * T];
  __shared__ int s_reg[N];
  const int nwx = p.W / WS, nW = (p.H / WS) * nwx;
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  for (int i = lane; i < T * T; i += 64) s_tab[i] = p.table[i * p.heads + h];
  f32x4_t accd[BWD ? 4 : 1][4];          // backward: dS^T summed over the block's windows, in the accumulator layout (registers: LDS is what limits occupancy)
#pragma unroll
  for (int kt = 0; kt < (BWD ? 4 : 1); ++kt)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) accd[kt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // a block walks wpb consecutive windows of one (image, head): the next window's rows are loaded while this one computes
  uint4 rq0, rq1, rq2, rq3, rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3, rd0, rd1, rd2, rd3;
  rd0 = rd1 = rd2 = rd3 = make_uint4(0, 0, 0, 0);
  const int w0 = blockIdx.x * wpb;
  WM_FETCH(w0);
  for (int wi = 0; wi < wpb; ++wi) {
  const int w = w0 + wi, wy = w / nwx, wx = w - wy * nwx;
  WM_PUT(s_q, rq0, rq1, rq2, rq3);
  WM_PUT(s_k, rk0, rk1, rk2, rk3);
  WM_PUT(s_v, rv0, rv1, rv2, rv3);
  if (BWD) WM_PUT(s_do, rd0, rd1, rd2, rd3);
  s_reg[lane] = p.shift > 0 ? wa_region(p, wy, wx, lane) : 0;
  __syncthreads();
  if (!ONE && wi + 1 < wpb) WM_FETCH(w + 1);

  // ---- transposed scores: lane holds key kj = 32*(kt>>1) + 8g + 4*(kt&1) + r (row rj = 4*(kt>>1) + g, column cj = 4*(kt&1) + r)
  //      of query qi = qt*16 + li
  f32x4_t st[4][4], dpt[4][4];
  {
    bf16x8_t kf[4], qf[4], vf[4], df[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kf[t] = wm_row(s_k, wm_tile_row(t, li), g);
      qf[t] = wm_row(s_q, t * 16 + li, g);
      if (BWD) { vf[t] = wm_row(s_v, wm_tile_row(t, li), g); df[t] = wm_row(s_do, t * 16 + li, g); }
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (BWD) dpt[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt], df[qt], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
  }
  int kreg[4][4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) kreg[kt][r] = s_reg[32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int qi = qt * 16 + li, ri = qi >> 3, ci = qi & 7, regq = s_reg[qi];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rj = 4 * (kt >> 1) + g, cj = 4 * (kt & 1) + r;
        float s = st[kt][qt][r] * p.scale + s_tab[(ri - rj + WS - 1) * T + (ci - cj + WS - 1)];      // :94-104
        if (kreg[kt][r] != regq) s += -100.f;                                                            // :106-110
        st[kt][qt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float e = __expf(st[kt][qt][r] - mx); st[kt][qt][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) st[kt][qt] *= inv;
  }

  if (!BWD) {
    // O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]
    f32x4_t o[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) o[dt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8_t vt[2] = {wm_tr(s_v, c, 0, g, tq, tp), wm_tr(s_v, c, 1, g, tq, tp)};
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const float pv[8] = {st[2 * c][qt][0], st[2 * c][qt][1], st[2 * c][qt][2], st[2 * c][qt][3],
                             st[2 * c + 1][qt][0], st[2 * c + 1][qt][1], st[2 * c + 1][qt][2], st[2 * c + 1][qt][3]};
        bf16x8_t ph, pl;
        wm_split(pv, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[dt], ph, o[dt][qt], 0, 0, 0);
          o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[dt], pl, o[dt][qt], 0, 0, 0);
        }
      }
    }
    // lane holds O[q = qt*16 + li][d = 16*dt + 4g + r]
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      bf16_t* dst = p.out + wa_token(p, b, wy, wx, qt * 16 + li) * p.ldo + h * 32 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        uint2 wv;
        wv.x = pack2bf(o[dt][qt][0], o[dt][qt][1]);
        wv.y = pack2bf(o[dt][qt][2], o[dt][qt][3]);
        *reinterpret_cast<uint2*>(dst + 16 * dt) = wv;
      }
    }
    __syncthreads();
    continue;
  }

  // ---- backward: dS^T = P^T (dP^T - delta_q), dQ^T[d][q] = scale * sum_keys K^T[d][key] dS^T[key][q]
  f32x4_t dq[2][4];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) dq[dt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) delta = fmaf(st[kt][qt][r], dpt[kt][qt][r], delta);
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
    const int qi = qt * 16 + li;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r;
        const float pr = st[kt][qt][r], ds = pr * (dpt[kt][qt][r] - delta);
        dpt[kt][qt][r] = ds;
        accd[kt][qt][r] += ds;
        s_mT[kj * WM_PS + qi] = pr;
      }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    bf16x8_t kt_[2] = {wm_tr(s_k, c, 0, g, tq, tp), wm_tr(s_k, c, 1, g, tq, tp)};
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const float dv[8] = {dpt[2 * c][qt][0], dpt[2 * c][qt][1], dpt[2 * c][qt][2], dpt[2 * c][qt][3],
                           dpt[2 * c + 1][qt][0], dpt[2 * c + 1][qt][1], dpt[2 * c + 1][qt][2], dpt[2 * c + 1][qt][3]};
      bf16x8_t dh, dl;
      wm_split(dv, dh, dl);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dq[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt], dh, dq[dt][qt], 0, 0, 0);
        dq[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt], dl, dq[dt][qt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    bf16_t* dst = p.out + wa_token(p, b, wy, wx, qt * 16 + li) * p.ldq + h * 32 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      uint2 wv;
      wv.x = pack2bf(dq[dt][qt][0] * p.scale, dq[dt][qt][1] * p.scale);
      wv.y = pack2bf(dq[dt][qt][2] * p.scale, dq[dt][qt][3] * p.scale);
      *reinterpret_cast<uint2*>(dst + 16 * dt) = wv;
    }
  }
  __syncthreads();
  // ---- dV^T[d][key] = sum_q dO^T[d][q] P[q][key],  then (same LDS matrix rewritten with dS^T) dK^T[d][key] = scale * sum_q Q^T[d][q] dS[q][key]
  //      B operand [slot q = 32c + 8g + j][col key = kt*16 + li]: 8 consecutive floats of row `key` of the [key][query] matrix
  f32x4_t dvv[2][4], dkk[2][4];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) dvv[dt][kt] = dkk[dt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      __syncthreads();
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s_mT[(32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r) * WM_PS + qt * 16 + li] = dpt[kt][qt][r];
      __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16_t* at = pass == 0 ? s_do : s_q;
      bf16x8_t atr[2] = {wm_tr(at, c, 0, g, tq, tp), wm_tr(at, c, 1, g, tq, tp)};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const float* pr = &s_mT[(kt * 16 + li) * WM_PS + 32 * c + 8 * g];
        const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(pr), p1 = *reinterpret_cast<const f32x4_t*>(pr + 4);
        const float pv[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
        bf16x8_t ph, pl;
        wm_split(pv, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          if (pass == 0) {
            dvv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr[dt], ph, dvv[dt][kt], 0, 0, 0);
            dvv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr[dt], pl, dvv[dt][kt], 0, 0, 0);
          } else {
            dkk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr[dt], ph, dkk[dt][kt], 0, 0, 0);
            dkk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr[dt], pl, dkk[dt][kt], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    bf16_t* dst = p.out + wa_token(p, b, wy, wx, kt * 16 + li) * p.ldq + h * 32 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      uint2 wv;
      wv.x = pack2bf(dkk[dt][kt][0] * p.scale, dkk[dt][kt][1] * p.scale);
      wv.y = pack2bf(dkk[dt][kt][2] * p.scale, dkk[dt][kt][3] * p.scale);
      *reinterpret_cast<uint2*>(dst + p.C + 16 * dt) = wv;
      wv.x = pack2bf(dvv[dt][kt][0], dvv[dt][kt][1]);
      wv.y = pack2bf(dvv[dt][kt][2], dvv[dt][kt][3]);
      *reinterpret_cast<uint2*>(dst + 2 * p.C + 16 * dt) = wv;
    }
  }
  __syncthreads();
  }   // windows
  if (!BWD) return;
#pragma unroll
  for (int qt = 0; qt < 4; ++qt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_mT[(32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r) * WM_PS + qt * 16 + li] = accd[kt][qt][r];
  __syncthreads();
  // relative-position bias gradient: bin (dr, dc) collects dS_ij over the pairs with r_i - r_j = dr, c_i - c_j = dc (fixed order),
  // once per block from the sum over its windows
  float* row = p.dtab_ws + ((int64_t)b * (nW / wpb) + blockIdx.x) * (T * T) * p.heads;
  for (int bin = lane; bin < T * T; bin += 64) {
    const int dr = bin / T - (WS - 1), dc = bin % T - (WS - 1);
    float s = 0.f;
    for (int ii = 0; ii < N; ++ii) {
      const int ri = ii >> 3, ci = ii & 7;
      const int rj = ri - dr, cj = ci - dc;
      if ((unsigned)rj < (unsigned)WS && (unsigned)cj < (unsigned)WS) s += s_mT[(rj * WS + cj) * WM_PS + ii];
    }
    row[bin * p.heads + h] = s;
  }
}

// returns the number of partial bias-table rows the backward wrote (0 = unsupported window side)
template <bool BWD>
static int wa_launch(const WinAttn& p, hipStream_t s) {
  const int N = p.ws * p.ws, nW = (p.H / p.ws) * (p.W / p.ws);
  const size_t dyn = sizeof(float) * ((size_t)(BWD ? 4 : 3) * N * (p.d + 1) + (BWD ? 2 * (size_t)N * (N + 1) : 0));
  static const int no_mfma = getenv("USSEG_WINATTN_MFMA") && atoi(getenv("USSEG_WINATTN_MFMA")) == 0;
  if (p.ws == 8 && p.d == 32 && !no_mfma) {
    // windows per block: a power of two dividing nW that still fills the chip (the forward holds 8 one-wave blocks per CU, the
    // backward 2: 78 KB of LDS); the backward gains most - one bias-table reduction per block instead of one per window
    const int64_t min_blocks = BWD ? 1024 : 8192;
    int wpb = BWD ? 16 : 4;
    while (wpb > 1 && (nW % wpb != 0 || (int64_t)(nW / wpb) * p.heads * p.B < min_blocks)) wpb >>= 1;
    if (wpb == 1) hipLaunchKernelGGL((window_attn_mfma_kernel<BWD, true>), dim3(nW, p.heads, p.B), dim3(64), 0, s, p, 1);
    else hipLaunchKernelGGL((window_attn_mfma_kernel<BWD, false>), dim3(nW / wpb, p.heads, p.B), dim3(64), 0, s, p, wpb);
    return p.B * (nW / wpb);
  }
  const dim3 grid(nW, p.heads, p.B);
#define WAL(WS_)                                                                                                                     \
  {                                                                                                                                  \
    if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void*)window_attn_kernel<BWD, WS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);   \
    hipLaunchKernelGGL((window_attn_kernel<BWD, WS_>), grid, dim3(64), dyn, s, p);                                                  \
  }
  if (p.ws == 2) WAL(2) else if (p.ws == 4) WAL(4) else if (p.ws == 8) WAL(8) else return 0;
#undef WAL
  return p.B * nW;
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
  const int rows = wa_launch<true>(p, (hipStream_t)stream);
  usseg_launch_reduce_finish(p.dtab_ws, 1, rows, 1, bins, bins, 1.f, dtable, nullptr, nullptr, (hipStream_t)stream);
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
