// 3x3 convolution (stride 1, SAME, any dilation) with an LDS-resident input halo tile, bf16 MFMA, for gfx950.
//
// Why a second conv kernel: the gather implicit GEMM (conv_igemm.hip) re-stages the 128-pixel activation tile for
// every tap, i.e. 9 x per channel chunk.  Here a workgroup stages its pixel patch PLUS a one-pixel halo ONCE per
// 32-channel chunk and all 9 taps read their shifted A fragments straight from that LDS image (per-lane addresses),
// so global->LDS traffic per MFMA drops ~5x.
//
// Dilation d is handled by decomposition: a dilated 3x3 conv is d*d independent DENSE 3x3 convs on the sub-lattices
// {(a + d*i, b + d*j)}.  Every (image, a, b) is a "virtual image" of (H/d) x (W/d) pixels read with pixel stride d,
// so the halo is always one lattice pixel wide, whatever the dilation.
//
// Workgroup = 128 output pixels (4 waves x 2 MFMA pixel-strips of 16) x 16*NT output channels.  The 128 pixels are
// NV patches of PH x PW lattice pixels (PW = min(16, W/d)); small virtual images (8x8, 4x4) are packed several to a
// workgroup.  Operands: x NHWC bf16, packed weights Wp[N][9*Cin] as for the gather kernel.  The same kernel runs the
// backward-data pass (flip = 1: tap position (kh,kw) uses weight tap 8 - (3*kh+kw), operand packed with in/out swapped).
// Pipeline: single LDS image, next chunk prefetched into registers during the MFMAs (2 workgroups per CU overlap).
#include "common.h"

struct HaloParams {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const float* bias;
  const float* scale;   // optional per-channel multiplier applied before the bias (folded inference BatchNorm)
  const bf16_t* res;
  int32_t B, H, W, d;
  int32_t Hl, Wl;        // lattice size = H/d, W/d
  int32_t PH, PW, NV;    // patch size, patches per workgroup
  int32_t tiles_x, tiles_per_v, npatches;
  int32_t ldx, ldy, ldr;
  int32_t Cin, nchunks;  // physical input channels (multiple of 8), ceil(Cin/32)
  int32_t Nw, Kw, Nout;
  int32_t act;
  float alpha;
  int32_t out_f32, accumulate, flip;
};

// Up to 4 independent convolutions of identical grid shape (the DecoderBlock's parallel dilation branches) share one
// launch, blockIdx.z = job: three times the workgroups of one branch, so 2-3 of them are resident per CU and hide each
// other's staging latency (one branch alone is a single wave of 256 one-per-CU workgroups at batch 16).
struct HaloMulti {
  HaloParams job[4];
};

template <int NT>
__global__ __launch_bounds__(256) void conv_halo_kernel(const HaloMulti P) {
  const HaloParams& p = P.job[blockIdx.z];
  constexpr int BN = 16 * NT, LDSS = 40;
  constexpr int MAXHP = 288;                    // halo pixels per workgroup (8 patches of 6x6 is the maximum)
  constexpr int A_IT = (MAXHP * 4 + 255) / 256;  // 5
  constexpr int W_IT = (9 * BN * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t lds_a[MAXHP * LDSS];
  __shared__ __attribute__((aligned(16))) bf16_t lds_w[9 * BN * LDSS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = blockIdx.y * BN;
  const int HW2 = p.PW + 2, HPP = (p.PH + 2) * HW2;  // halo row length, halo pixels per patch
  const int NHP = p.NV * HPP;

  // ---- staging assignments (fixed for the whole kernel): A items = (halo pixel, 16-byte chunk q)
  int a_off[A_IT];  // element offset of the pixel in x, or -1
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int idx = tid + 256 * it;
    int hp = idx >> 2;
    int off = -1;
    if (hp < NHP) {
      int pi = hp / HPP, rem = hp - pi * HPP;
      int hy = rem / HW2, hx = rem - hy * HW2;
      int gp = blockIdx.x * p.NV + pi;
      if (gp < p.npatches) {
        int v = gp / p.tiles_per_v, tt = gp - v * p.tiles_per_v;
        int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        int dd = p.d * p.d;
        int b = v / dd, ab = v - b * dd;
        int la = ab / p.d, lb = ab - la * p.d;
        int ly = ty * p.PH + hy - 1, lx = tx * p.PW + hx - 1;
        if ((unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl)
          off = ((b * p.H + la + p.d * ly) * p.W + lb + p.d * lx) * p.ldx;
      }
    }
    a_off[it] = off;
  }
  const int q = tid & 3;

  uint4 ra[A_IT], rw[W_IT];
  auto load_chunk = [&](int ck) {
    const int c = ck * 32 + q * 8;
    const bool cok = c < p.Cin;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (cok && a_off[it] >= 0) v = *reinterpret_cast<const uint4*>(p.x + a_off[it] + c);
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      int idx = tid + 256 * it;
      int row = idx >> 2;  // t*BN + n
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < 9 * BN) {
        int t = row / BN, n = row - t * BN;
        int tw = p.flip ? 8 - t : t;
        if (cok && (n0 + n) < p.Nw) v = *reinterpret_cast<const uint4*>(p.w + (int64_t)(n0 + n) * p.Kw + tw * p.Cin + c);
      }
      rw[it] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int idx = tid + 256 * it;
      if ((idx >> 2) < NHP) *reinterpret_cast<uint4*>(&lds_a[(idx >> 2) * LDSS + q * 8]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      int idx = tid + 256 * it;
      if ((idx >> 2) < 9 * BN) *reinterpret_cast<uint4*>(&lds_w[(idx >> 2) * LDSS + q * 8]) = rw[it];
    }
  };

  // ---- this lane's two output pixels (one per strip): LDS halo index of the top-left tap, and the image pixel
  const int pl = lane & 15, fk = (lane >> 4) * 8;
  const int rows_per_strip = 16 / p.PW;  // PW in {4, 8, 16}
  const int spp = (p.PH * p.PW) >> 4;    // strips per patch
  int a_base[2];
  int64_t opix[2];
  bool ovalid[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    int s = wv * 2 + s2;
    int pi = s / spp, sl = s - pi * spp;
    int r = pl / p.PW, c = pl - r * p.PW;
    int row = sl * rows_per_strip + r;
    a_base[s2] = (pi * HPP + row * HW2 + c) * LDSS + fk;
    int gp = blockIdx.x * p.NV + pi;
    ovalid[s2] = gp < p.npatches;
    int gpc = ovalid[s2] ? gp : 0;
    int v = gpc / p.tiles_per_v, tt = gpc - v * p.tiles_per_v;
    int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    int dd = p.d * p.d;
    int b = v / dd, ab = v - b * dd;
    int la = ab / p.d, lb = ab - la * p.d;
    int iy = la + p.d * (ty * p.PH + row), ix = lb + p.d * (tx * p.PW + c);
    opix[s2] = ((int64_t)b * p.H + iy) * p.W + ix;
  }
  const int w_base = (lane & 15) * LDSS + fk;

  f32x4_t acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_chunk(0);
  for (int ck = 0; ck < p.nchunks; ++ck) {
    store_chunk();
    __syncthreads();
    if (ck + 1 < p.nchunks) load_chunk(ck + 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t - 3 * kh;
      const int toff = (kh * HW2 + kw) * LDSS;
      bf16x8_t xf[2], wf[NT];
#pragma unroll
      for (int a = 0; a < 2; ++a) xf[a] = *reinterpret_cast<const bf16x8_t*>(&lds_a[a_base[a] + toff]);
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(&lds_w[(t * BN + b * 16) * LDSS + w_base]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue (same contract as the gather kernel): lane holds Y[its pixel][n = 4*(lane>>4) + j] per n-tile
  const EpiArgs e = {p.scale, p.bias, p.res, p.y, p.ldy, p.ldr, p.Nout, p.act, p.alpha, p.out_f32, p.accumulate};
  const int nbase = n0 + (lane >> 4) * 4;
  EpiConst<NT> ec;
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);
  conv_epilogue<2, NT>(e, ec, acc, opix, ovalid, nbase);
}

// ---- persistent variant for the HBM-bound layers (few input channels, many pixels) ------------------------------
// When the whole weight operand of a workgroup's channel tile (nchunks x 9 x BN x 32) fits in LDS it is staged ONCE and
// the workgroup then walks over many pixel groups: per group it only stages the activation halo tile.  (In the plain
// kernel a 128-pixel workgroup of the 32->32 stem conv moves 18 KB of weights for 11 KB of activations.)
// `gpb` consecutive pixel groups per workgroup; their patch geometry is decoded once into LDS.
template <int NT>
__global__ __launch_bounds__(256, NT <= 1 ? 4 : (NT == 2 ? 3 : 2)) void conv_halo_persist_kernel(const HaloMulti P, const int gpb) {
  const HaloParams& p = P.job[blockIdx.z];
  constexpr int BN = 16 * NT, LDSS = 40, MAXHP = 288, MAXG = 16;
  constexpr int A_IT = (MAXHP * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_dyn[];   // [nchunks][9*BN][LDSS] weights, then the halo tile
  __shared__ int s_patch[MAXG * 8][6];                                // b, la, lb, ly0, lx0, valid
  bf16_t* lds_w = lds_dyn;
  bf16_t* lds_a = lds_dyn + (size_t)p.nchunks * 9 * BN * LDSS;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = blockIdx.y * BN;
  const int HW2 = p.PW + 2, HPP = (p.PH + 2) * HW2, NHP = p.NV * HPP;
  const int ngroups = (p.npatches + p.NV - 1) / p.NV;
  // XCD-aware walk (workgroup i runs on XCD i % 8): each XCD owns a contiguous eighth of the pixel groups and its
  // workgroups take ADJACENT groups at the same time, so concurrent halo tiles are neighbours in memory (spread over all
  // L2 channels, shared halos hit in that XCD's L2) instead of sitting a power-of-two stride apart.
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, nj = gridDim.x >> 3;   // gridDim.x is a multiple of 8
  const int gpx = (ngroups + 7) >> 3;
  const int g_lo = xcd * gpx + wj;
  int g_hi = (xcd + 1) * gpx;
  if (g_hi > ngroups) g_hi = ngroups;
  int ng = g_lo < g_hi ? (g_hi - g_lo + nj - 1) / nj : 0;
  if (ng > gpb) ng = gpb;   // host sizes gpb so that this never truncates
  const int q = tid & 3;

  // weights: all chunks, once
  for (int idx = tid; idx < p.nchunks * 9 * BN * 4; idx += 256) {
    int row = idx >> 2;                  // (ck*9 + t)*BN + n
    int ck = row / (9 * BN), rem = row - ck * 9 * BN;
    int t = rem / BN, n = rem - t * BN;
    int c = ck * 32 + (idx & 3) * 8;
    int tw = p.flip ? 8 - t : t;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < p.Cin && (n0 + n) < p.Nw) v = *reinterpret_cast<const uint4*>(p.w + (int64_t)(n0 + n) * p.Kw + tw * p.Cin + c);
    *reinterpret_cast<uint4*>(&lds_w[row * LDSS + (idx & 3) * 8]) = v;
  }
  // patch table of every group this workgroup owns
  for (int i = tid; i < ng * p.NV; i += 256) {
    int gp = (g_lo + (i / p.NV) * nj) * p.NV + (i % p.NV);
    int valid = gp < p.npatches;
    int gpc = valid ? gp : 0;
    int v = gpc / p.tiles_per_v, tt = gpc - v * p.tiles_per_v;
    int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    int dd = p.d * p.d;
    int b = v / dd, ab = v - b * dd;
    int* e = s_patch[(i / p.NV) * 8 + (i % p.NV)];
    e[0] = b; e[1] = ab / p.d; e[2] = ab - (ab / p.d) * p.d; e[3] = ty * p.PH; e[4] = tx * p.PW; e[5] = valid;
  }
  // fixed staging geometry: (patch, halo row, halo col) of each A item of this thread
  int a_geo[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int hp = (tid + 256 * it) >> 2;
    a_geo[it] = -1;
    if (hp < NHP) {
      int pi = hp / HPP, rem = hp - pi * HPP;
      int hy = rem / HW2;
      a_geo[it] = (pi << 16) | (hy << 8) | (rem - hy * HW2);
    }
  }
  // this lane's two output pixels: patch, row, col (fixed), LDS base of the top-left tap
  const int pl = lane & 15, fk = (lane >> 4) * 8;
  const int rows_per_strip = 16 / p.PW, spp = (p.PH * p.PW) >> 4;
  int a_base[2], o_pi[2], o_row[2], o_col[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    int s = wv * 2 + s2;
    int pi = s / spp, sl = s - pi * spp;
    int r = pl / p.PW, c = pl - r * p.PW;
    o_pi[s2] = pi; o_row[s2] = sl * rows_per_strip + r; o_col[s2] = c;
    a_base[s2] = (pi * HPP + o_row[s2] * HW2 + c) * LDSS + fk;
  }
  const int w_base = (lane & 15) * LDSS + fk;
  const EpiArgs e = {p.scale, p.bias, p.res, p.y, p.ldy, p.ldr, p.Nout, p.act, p.alpha, p.out_f32, p.accumulate};
  const int nbase = n0 + (lane >> 4) * 4;
  EpiConst<NT> ec;                               // per-lane bias: loaded once for all pixel groups
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);
  __syncthreads();

  uint4 ra[A_IT];
  auto load_a = [&](int gl, int ck) {
    const int c = ck * 32 + q * 8;
    const bool cok = c < p.Cin;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_geo[it] >= 0 && cok) {
        const int* pt = s_patch[gl * 8 + (a_geo[it] >> 16)];
        int ly = pt[3] + ((a_geo[it] >> 8) & 255) - 1, lx = pt[4] + (a_geo[it] & 255) - 1;
        if (pt[5] && (unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl)
          v = *reinterpret_cast<const uint4*>(p.x + ((int64_t)(pt[0] * p.H + pt[1] + p.d * ly) * p.W + pt[2] + p.d * lx) * p.ldx + c);
      }
      ra[it] = v;
    }
  };

  f32x4_t acc[2][NT];
  const int total = ng * p.nchunks;
  if (total > 0) load_a(0, 0);
  for (int it = 0; it < total; ++it) {
    const int gl = it / p.nchunks, ck = it - gl * p.nchunks;
    if (ck == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < A_IT; ++k) {
      int idx = tid + 256 * k;
      if ((idx >> 2) < NHP) *reinterpret_cast<uint4*>(&lds_a[(idx >> 2) * LDSS + q * 8]) = ra[k];
    }
    __syncthreads();
    if (it + 1 < total) load_a((it + 1) / p.nchunks, (it + 1) % p.nchunks);
    const bf16_t* wck = lds_w + (size_t)ck * 9 * BN * LDSS;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int toff = ((t / 3) * HW2 + (t % 3)) * LDSS;
      bf16x8_t xf[2], wf[NT];
#pragma unroll
      for (int a = 0; a < 2; ++a) xf[a] = *reinterpret_cast<const bf16x8_t*>(&lds_a[a_base[a] + toff]);
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(&wck[(t * BN + b * 16) * LDSS + w_base]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    if (ck != p.nchunks - 1) continue;
    // ---- epilogue of pixel group gl
    int64_t opix[2];
    bool ovalid[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int* pt = s_patch[gl * 8 + o_pi[a]];
      ovalid[a] = pt[5] != 0;
      opix[a] = ((int64_t)pt[0] * p.H + pt[1] + p.d * (pt[3] + o_row[a])) * p.W + pt[2] + p.d * (pt[4] + o_col[a]);
    }
    conv_epilogue<2, NT>(e, ec, acc, opix, ovalid, nbase);
  }
}

// Fills p from a conv geometry; returns 0 if the geometry does not fit the halo tiling.
static int halo_fill(HaloParams& p, const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W, int d,
                     int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32, int accumulate,
                     int flip) {
  if (d < 1 || H % d || W % d) return 0;
  const int Hl = H / d, Wl = W / d;
  int PW;
  if (Wl % 16 == 0) PW = 16;
  else if (Wl == 8 || Wl == 4) PW = Wl;
  else return 0;
  int PH = 128 / PW;
  if (PH > Hl) PH = Hl;
  if (Hl % PH || (PH * PW) % 16) return 0;
  const int NV = 128 / (PH * PW);
  if (NV * (PH + 2) * (PW + 2) > 288) return 0;
  if ((int64_t)B * H * W * ldx >= (1ll << 31)) return 0;
  p = {};
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res;
  p.B = B; p.H = H; p.W = W; p.d = d; p.Hl = Hl; p.Wl = Wl; p.PH = PH; p.PW = PW; p.NV = NV;
  p.tiles_x = Wl / PW;
  p.tiles_per_v = (Hl / PH) * p.tiles_x;
  p.npatches = B * d * d * p.tiles_per_v;
  p.ldx = ldx; p.ldy = ldy; p.ldr = ldr; p.Cin = Cin; p.nchunks = (Cin + 31) / 32;
  p.Nw = Nw; p.Kw = Kw; p.Nout = Nout; p.act = act; p.alpha = alpha; p.out_f32 = out_f32; p.accumulate = accumulate; p.flip = flip;
  return 1;
}

// Launches njobs jobs of identical (npatches/NV, Nout, nchunks) as one grid; returns 0 if they do not match.
static int halo_launch(HaloMulti& P, int njobs, hipStream_t s) {
  const HaloParams& p0 = P.job[0];
  const int gx = (p0.npatches + p0.NV - 1) / p0.NV;
  int maxhp = 0;
  for (int j = 0; j < njobs; ++j) {
    const HaloParams& p = P.job[j];
    if ((p.npatches + p.NV - 1) / p.NV != gx || p.Nout != p0.Nout || p.nchunks != p0.nchunks) return 0;
    int hp = p.NV * (p.PH + 2) * (p.PW + 2);
    if (hp > maxhp) maxhp = hp;
  }
  const int Nout = p0.Nout;
  // pick the channel tile: the widest that still gives the chip a few hundred workgroups
  int nt = Nout <= 16 ? 1 : (Nout <= 32 ? 2 : 4);
  while (nt > 1 && (int64_t)gx * njobs * ((Nout + 16 * nt - 1) / (16 * nt)) < 256 && Nout > 16 * (nt / 2)) nt >>= 1;
  const int gy = (Nout + 16 * nt - 1) / (16 * nt);
  const int slot = usseg_prof_start(1, s);
  // persistent variant: the tile's whole weight operand fits in 46 KB of LDS and there are pixel groups to amortise it over
  static const int no_persist = getenv("USSEG_NO_PERSIST") != nullptr;
  const size_t wbytes = (size_t)p0.nchunks * 9 * 16 * nt * 40 * sizeof(bf16_t);
  if (!no_persist && wbytes <= 46080 && gx >= 1024) {
    const size_t dyn = wbytes + (size_t)maxhp * 40 * sizeof(bf16_t);   // weights + the largest halo tile of the jobs
    // One resident wave of workgroups: `occ` per CU by LDS (160 KB) and by the launch bounds.  A second, partly filled wave
    // would cost a whole extra pass (3 resident + 1024 workgroups used to run as 768 + 256).
    static const int occ_env = getenv("USSEG_PERSIST_OCC") ? atoi(getenv("USSEG_PERSIST_OCC")) : 0;
    int occ = (int)((160 * 1024) / (dyn + 3072 + 256));
    const int occ_max = nt <= 2 ? 4 : 2;
    if (occ > occ_max) occ = occ_max;
    if (occ < 1) occ = 1;
    if (occ_env > 0) occ = occ_env;
    const int64_t total_groups = (int64_t)gx * njobs;
    int waves = 1;
    while ((total_groups + 256 * occ * waves - 1) / (256 * occ * waves) > 16) ++waves;   // at most 16 groups per workgroup
    int gpb = (int)((total_groups + 256 * occ * waves - 1) / (256 * occ * waves));
    if (gpb < 1) gpb = 1;
    int pgx = ((gx + gpb - 1) / gpb + 7) & ~7;       // a multiple of 8: workgroups per XCD = pgx / 8
    gpb = (((gx + 7) >> 3) + (pgx >> 3) - 1) / (pgx >> 3);   // groups per workgroup under the per-XCD split
    if (gpb > 16) { pgx = ((((gx + 7) >> 3) + 15) / 16) * 8; gpb = 16; }
    static bool attr_done = false;   // > 64 KB of LDS per workgroup needs the opt-in (one-time host call, never a stream op)
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)conv_halo_persist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_halo_persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_halo_persist_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_done = true;
    }
    if (nt == 1) hipLaunchKernelGGL(conv_halo_persist_kernel<1>, dim3(pgx, gy, njobs), dim3(256), dyn, s, P, gpb);
    else if (nt == 2) hipLaunchKernelGGL(conv_halo_persist_kernel<2>, dim3(pgx, gy, njobs), dim3(256), dyn, s, P, gpb);
    else hipLaunchKernelGGL(conv_halo_persist_kernel<4>, dim3(pgx, gy, njobs), dim3(256), dyn, s, P, gpb);
    usseg_prof_stop(1, slot, s);
    return 1;
  }
  if (nt == 1) hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(gx, gy, njobs), dim3(256), 0, s, P);
  else if (nt == 2) hipLaunchKernelGGL(conv_halo_kernel<2>, dim3(gx, gy, njobs), dim3(256), 0, s, P);
  else hipLaunchKernelGGL(conv_halo_kernel<4>, dim3(gx, gy, njobs), dim3(256), 0, s, P);
  usseg_prof_stop(1, slot, s);
  return 1;
}

// Returns 1 and launches if the geometry fits the halo kernel, 0 if the caller must use the gather kernel.
int usseg_try_launch_conv_halo(const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W, int d,
                               int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32,
                               int accumulate, int flip, hipStream_t s) {
  static const int disabled = getenv("USSEG_NO_HALO") != nullptr;
  if (disabled) return 0;
  HaloMulti P;
  if (!halo_fill(P.job[0], x, w, y, bias, res, B, H, W, d, Cin, ldx, Nout, ldy, ldr, Nw, Kw, act, alpha, out_f32, accumulate, flip)) return 0;
  P.job[0].scale = flip ? nullptr : usseg_epi_scale[0];
  return halo_launch(P, 1, s);
}

// Multi-job form (usseg_conv2d_fwd_multi / _dgrad_multi): every job must fit and share the grid shape.
int usseg_try_launch_conv_halo_multi(int njobs, const UssegConvJob* jobs, int flip, hipStream_t s) {
  static const int disabled = getenv("USSEG_NO_HALO") != nullptr;
  if (disabled || njobs < 1 || njobs > 4) return 0;
  HaloMulti P;
  for (int j = 0; j < njobs; ++j) {
    const UssegConvJob& q = jobs[j];
    const UssegConvDesc& d = q.desc;
    const int out_f32 = (d.flags & USSEG_OUT_F32) ? 1 : 0, acc = (d.flags & USSEG_ACCUMULATE) ? 1 : 0;
    int ok = flip ? halo_fill(P.job[j], (const bf16_t*)q.x, (const bf16_t*)q.wp, q.y, nullptr, (const bf16_t*)q.residual, d.B, d.H, d.W,
                              d.dilation, d.Cout, d.ldy, d.Cin, d.ldx, q.ldr, roundup(d.Cin, 16), 9 * d.Cout, USSEG_ACT_NONE, 0.f, 0, acc, 1)
                  : halo_fill(P.job[j], (const bf16_t*)q.x, (const bf16_t*)q.wp, q.y, q.bias, (const bf16_t*)q.residual, d.B, d.H, d.W,
                              d.dilation, d.Cin, d.ldx, d.Cout, d.ldy, q.ldr, roundup(d.Cout, 16), 9 * d.Cin, d.act, d.alpha, out_f32, acc, 0);
    if (!ok) return 0;
    P.job[j].scale = flip ? nullptr : usseg_epi_scale[j];
  }
  return halo_launch(P, njobs, s);
}
