// Weight gradient of a 3x3 convolution (stride 1, SAME, any dilation) with an LDS-resident input halo tile, gfx950.
//
//   dW[kh,kw][ci][co] += sum_p x[p + (kh-1,kw-1)*d][ci] * dy[p][co]
//
// The per-tap kernel (wgrad.hip) re-reads x and dy once per tap (9x).  Here a workgroup stages, per 128-pixel
// chunk, the x patch + one-pixel halo and the dy patch ONCE and runs all 9 taps out of LDS: the shifted A^T
// fragments are transposed reads (ds_read_b64_tr_b16) whose per-lane row addresses carry the tap offset.
// Dilation uses the sub-lattice decomposition of conv_halo.hip (same patch geometry).
// Workgroup tile: (32*WM input channels) x (32*WN output channels) x 9 taps, 4 waves as 2x2; split-K over pixel
// chunks across blockIdx.x, fp32 atomics into the [9][Ma][Nb] scratch (only the valid m<Ma, n<Nb entries).
#include "common.h"

struct WgHaloParams {
  const bf16_t* x;
  const bf16_t* dy;
  float* out;
  float* ws;  // partial slabs [splits][9][Ma][Nb], or NULL (atomics into out)
  int32_t B, H, W, d, Hl, Wl, PH, PW, NV;
  int32_t tiles_x, tiles_per_v, npatches, ngroups, groups_per_block;
  int32_t ldx, lddy, Ma, Nb, ntiles_n;
  int32_t mask_ch;        // quad-form transposed conv: output channels per parity class (0 = all taps)
  uint16_t tapmask[4];
  // operand views: lattice pixel (ly, lx) of a patch with lattice offset (la, lb) is image pixel
  // (va + la + vd*ly, vb + lb + vd*lx) of a vH x vW image.  Conv: both (d, H, W, 0, 0).  Transposed conv, parity class (a, b):
  // the halo operand is dy seen through (2, 2H, 2W, a, b), the centre operand x through (1, H, W, 0, 0).
  int32_t xvd, xvH, xvW, xva, xvb;
  int32_t yvd, yvH, yvW, yva, yvb;
};

// WVM = waves along the input-channel axis (2 or 4; the other 4/WVM waves split the output channels),
// WM x WN = 16x16 MFMA tiles per wave: workgroup tile = (16*WM*WVM) x (16*WN*(4/WVM)).
// Up to 4 independent weight gradients of identical shape (the DecoderBlock's dilation branches) share one launch,
// blockIdx.z = job: the same number of workgroups with a third of the split-K slabs per job.
struct WgHaloMulti {
  WgHaloParams job[4];
  // workgroups go to the 8 XCDs round-robin in dispatch order (x = split fastest): the channel tiles / jobs that stage the SAME pixel
  // groups would run in different XCDs at different times.  Remapped, XCD c owns a contiguous run of the (split, job, tile) items in
  // tile-fastest order, so the items of one pixel range share that XCD's L2 (see WgradParams::xcd_remap).
  int32_t xcd_remap;
};

// PF: the next pixel group's operands are loaded into registers while the current group computes (the small tile shapes
// have the registers for it; without it a group is a serial load -> LDS -> barrier -> MFMA chain of ~4 us).
// NTG: tap groups.  NTG = 2 runs 8 waves: waves 0-3 accumulate taps 0-4, waves 4-7 taps 5-8 of the SAME LDS tiles, which
// halves the accumulator registers per wave (144 -> 80 for the 64x64 tile) and makes room for the prefetch registers.
template <int WVM, int WM, int WN, bool PF, int NTG, bool MASKED>
__global__ __launch_bounds__(256 * NTG, NTG == 1 ? 2 : 1) void wgrad_halo_kernel(const WgHaloMulti P) {
  constexpr int NTHR = 256 * NTG, TPG = NTG == 1 ? 9 : 5;   // threads, taps per group
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;    // (split, tile, job)
  if (P.xcd_remap) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    const int L = bx + gx * (by + gy * bz);
    const int c = L & 7, q = total >> 3, r = total & 7;
    const int item = c * q + (c < r ? c : r) + (L >> 3);
    const int u = __builtin_amdgcn_readfirstlane(fdiv(item, fdiv_rcp(gy)));     // (grids are far below 2^20 workgroups)
    by = item - u * gy;
    bx = __builtin_amdgcn_readfirstlane(fdiv(u, fdiv_rcp(gz)));
    bz = u - bx * gz;
  }
  const WgHaloParams& p = P.job[bz];
  constexpr int MAXHP = 288;
  constexpr int WVN = 4 / WVM;
  constexpr int BMc = 16 * WM * WVM, BNc = 16 * WN * WVN;      // channels per workgroup tile
  // LDS row strides (elements).  A transposed fragment read covers 16 consecutive pixel rows x 32 B, 8 rows per 32-lane
  // pass: with a stride of 16 (mod 32) elements the 8 rows start at 8 distinct multiples of 8 banks - conflict-free
  // (the +8 padding this started with was 2-way conflicted for every row mapping).
  constexpr int XS = BMc + 16, YS = BNc == 16 ? 48 : BNc + 16;
  constexpr int XCH = BMc / 8, YCH = BNc / 8;      // 16-byte chunks per pixel row
  constexpr int X_IT = (MAXHP * XCH + NTHR - 1) / NTHR, Y_IT = (128 * YCH + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) bf16_t lds_x[MAXHP * XS];
  __shared__ __attribute__((aligned(16))) bf16_t lds_y[128 * YS];
  // per (pixel group of the current batch, patch): x origin, dy origin, ly0 - 1, lx0 - 1, valid.  One thread per entry fills the
  // table for TB groups at a time: computed per group by the first NV lanes of wave 0 (130 VALU instructions, 30 of them
  // quarter-rate integer multiplies of four divisions, every group, with the other waves waiting at the barrier) it cost that
  // wave more issue time than the group's MFMAs
  constexpr int TB = NTHR / 8;
  __shared__ __attribute__((aligned(16))) int s_tab[TB][8][8];

  const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3, tg = tid >> 8;   // tile wave, tap group
  const int t0 = tg * TPG;
  const int HW2 = p.PW + 2, HPP = (p.PH + 2) * HW2, NHP = p.NV * HPP;
  const int rps = 16 / p.PW, spp = (p.PH * p.PW) >> 4;
  // tile / patch decodes with fdiv() (common.h): every index here is a tile, patch or halo-pixel number far below 2^20
  const float r_hpp = fdiv_rcp(HPP), r_hw2 = fdiv_rcp(HW2), r_spp = fdiv_rcp(spp), r_pw = fdiv_rcp(p.PW), r_tpv = fdiv_rcp(p.tiles_per_v),
              r_tx = fdiv_rcp(p.tiles_x), r_dd = fdiv_rcp(p.d * p.d), r_d = fdiv_rcp(p.d);
  int mt, nt;
  fdivmod(by, p.ntiles_n, fdiv_rcp(p.ntiles_n), mt, nt);
  const int m0 = mt * BMc, n0 = nt * BNc;

  // ---- fixed per-thread staging geometry
  int x_geo[X_IT];  // (patch << 16) | (halo row << 8) | halo col, or -1
#pragma unroll
  for (int it = 0; it < X_IT; ++it) {
    int idx = tid + NTHR * it;
    int hp = idx / XCH;
    x_geo[it] = -1;
    if (hp < NHP) {
      int pi, rem, hy, hx;
      fdivmod(hp, HPP, r_hpp, pi, rem);
      fdivmod(rem, HW2, r_hw2, hy, hx);
      x_geo[it] = (pi << 16) | (hy << 8) | hx;
    }
  }
  int y_geo[Y_IT];  // (patch << 16) | (row << 8) | col
#pragma unroll
  for (int it = 0; it < Y_IT; ++it) {
    int idx = tid + NTHR * it;
    int pk = (idx / YCH) & 127;  // 0..127 (threads past the tile are masked at the load)
    int s = pk >> 4, pl = pk & 15;
    int pi, sl, r, c;
    fdivmod(s, spp, r_spp, pi, sl);
    fdivmod(pl, p.PW, r_pw, r, c);
    y_geo[it] = (pi << 16) | ((sl * rps + r) << 8) | c;
  }
  const int xq = tid % XCH, yq = tid % YCH;  // NTHR % XCH == 0 and NTHR % YCH == 0, so the chunk column is fixed
  const bool x_cok = (m0 + xq * 8) < p.Ma, y_cok = (n0 + yq * 8) < p.Nb;
  // element offsets of each item relative to its patch origin (the per-group part comes from the patch table): the loads
  // are then branch-free - one 16-byte table read, an add, a clamped unconditional load and a mask.  (The version that
  // decoded the patch and tested the bounds under per-item branches spent ~430 cycles per item, more than the MFMAs.)
  int x_rel[X_IT], y_rel[Y_IT];
#pragma unroll
  for (int it = 0; it < X_IT; ++it) {
    const int hy = (x_geo[it] >> 8) & 255, hx = x_geo[it] & 255;
    x_rel[it] = __mul24(__mul24(__mul24(p.xvd, hy), p.xvW) + __mul24(p.xvd, hx), p.ldx) + m0 + xq * 8;
  }
#pragma unroll
  for (int it = 0; it < Y_IT; ++it) {
    const int row = (y_geo[it] >> 8) & 255, col = y_geo[it] & 255;
    y_rel[it] = __mul24(__mul24(__mul24(p.yvd, row), p.yvW) + __mul24(p.yvd, col), p.lddy) + n0 + yq * 8;
  }

  // ---- fragment addressing (fixed): K index 8g + 4h + tq of a 32-pixel step is pixel 16h + 4g + tq (same permutation for x and dy)
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int wm = wv / WVN, wn = wv % WVN;
  int xb[4][2], yb[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int pk = 32 * ks + 16 * h + 4 * g + tq;   // one read (fixed h) = one 16-pixel strip
      int s = pk >> 4, pl = pk & 15;
      int pi, sl, r, c;
      fdivmod(s, spp, r_spp, pi, sl);
      fdivmod(pl, p.PW, r_pw, r, c);
      xb[ks][h] = (pi * HPP + (sl * rps + r) * HW2 + c) * XS + wm * 16 * WM + 4 * tp;
      yb[ks][h] = pk * YS + wn * 16 * WN + 4 * tp;
    }

  f32x4_t acc[TPG][WM][WN];
#pragma unroll
  for (int t = 0; t < TPG; ++t)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto tr8 = [&](const bf16_t* base, int a0, int a1) -> bf16x8_t {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(base + a0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(base + a1));
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  const int g_begin = bx * p.groups_per_block;
  int g_end = g_begin + p.groups_per_block;
  if (g_end > p.ngroups) g_end = p.ngroups;

  auto fill_batch = [&](int grp0) {     // entries of groups grp0 .. grp0 + TB - 1 (past g_end: marked invalid, never read)
    const int gi = tid >> 3, pi = tid & 7;
    int (*tab)[8] = s_tab[gi];
    if (pi < p.NV) {
      const int tid = pi;               // (the body below is the per-patch code: "tid" is the patch slot)
      int gp = (grp0 + gi) * p.NV + pi;
      int valid = gp < p.npatches && (grp0 + gi) < g_end;
      int gpc = valid ? gp : 0;
      int v, tt, ty, tx, b, ab, la, lb;
      fdivmod(gpc, p.tiles_per_v, r_tpv, v, tt);
      fdivmod(tt, p.tiles_x, r_tx, ty, tx);
      fdivmod(v, p.d * p.d, r_dd, b, ab);
      fdivmod(ab, p.d, r_d, la, lb);
      const int ly0 = __mul24(ty, p.PH), lx0 = __mul24(tx, p.PW);
      // [0] x offset of halo pixel (0,0) (may be "negative": only in-image items use it), [1] dy offset of patch pixel (0,0),
      // [2] ly0 - 1 (far out of range for an invalid patch: every bounds test then fails), [3] lx0 - 1, [4] valid
      // (24-bit multiplies: pixel indices are below 2^24 and element offsets below 2^31 - the launcher's checks - so the low 32 bits are exact;
      //  a "negative" halo origin wraps exactly as the 64-bit form truncated to int did)
      tab[tid][0] = __mul24(__mul24(__mul24(b, p.xvH) + p.xva + la + __mul24(p.xvd, ly0 - 1), p.xvW) + p.xvb + lb + __mul24(p.xvd, lx0 - 1), p.ldx);
      tab[tid][1] = __mul24(__mul24(__mul24(b, p.yvH) + p.yva + la + __mul24(p.yvd, ly0), p.yvW) + p.yvb + lb + __mul24(p.yvd, lx0), p.lddy);
      tab[tid][2] = valid ? ly0 - 1 : -(1 << 24);
      tab[tid][3] = lx0 - 1;
      tab[tid][4] = valid;
    }
  };
  const bool nv1 = p.NV == 1;
  // two passes so that the table reads of every item are in flight together, then the global loads back to back
  // The loads are clamped, NOT masked: a select on a loaded value (`if (!ok) v = 0`) makes the compiler wait for every prefetch load right
  // behind its issue (s_waitcnt vmcnt(N) + v_cndmask per item, in front of the MFMA phase the loads were meant to fly under).  The validity
  // bits travel in mx / my and the zeroing happens where the registers are consumed: in front of the LDS stores, after the MFMAs.
  auto load_xy = [&](int (*tab)[8], uint4 (&rx)[X_IT], uint4 (&ry)[Y_IT], uint32_t& mx, uint32_t& my) {
    mx = my = 0u;
    int4 ex[X_IT];
    int yorg[Y_IT], yval[Y_IT];
    if (nv1) {      // one patch per group (every image at least 16 pixels wide and 8 high): one entry for all items
      const int4 e = *reinterpret_cast<const int4*>(tab[0]);
      const int v = tab[0][4];
#pragma unroll
      for (int it = 0; it < X_IT; ++it) ex[it] = e;
#pragma unroll
      for (int it = 0; it < Y_IT; ++it) { yorg[it] = e.y; yval[it] = v; }
    } else {
#pragma unroll
      for (int it = 0; it < X_IT; ++it) {
        const int geo = x_geo[it] < 0 ? 0 : x_geo[it];
        ex[it] = *reinterpret_cast<const int4*>(tab[geo >> 16]);
      }
#pragma unroll
      for (int it = 0; it < Y_IT; ++it) {
        const int* pt = tab[y_geo[it] >> 16];
        yorg[it] = pt[1];
        yval[it] = pt[4];
      }
    }
#pragma unroll
    for (int it = 0; it < X_IT; ++it) asm volatile("" : "+v"(ex[it].x), "+v"(ex[it].z), "+v"(ex[it].w));   // pin the reads above the loads
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) asm volatile("" : "+v"(yorg[it]), "+v"(yval[it]));
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int geo = x_geo[it] < 0 ? 0 : x_geo[it];
      const int ly = ex[it].z + ((geo >> 8) & 255), lx = ex[it].w + (geo & 255);
      // bitwise &: with && the compiler evaluates lazily and puts a branch per condition back
      const bool ok = (x_geo[it] >= 0) & x_cok & ((unsigned)ly < (unsigned)p.Hl) & ((unsigned)lx < (unsigned)p.Wl);
      rx[it] = *reinterpret_cast<const uint4*>(p.x + (ok ? ex[it].x + x_rel[it] : 0));
      mx |= (ok ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const bool ok = (tid + NTHR * it < 128 * YCH) & y_cok & (yval[it] != 0);
      ry[it] = *reinterpret_cast<const uint4*>(p.dy + (ok ? yorg[it] + y_rel[it] : 0));
      my |= (ok ? 1u : 0u) << it;
    }
  };
  const uint32_t tmask = (MASKED && p.mask_ch) ? p.tapmask[n0 / p.mask_ch] : 0x1ffu;   // workgroup-uniform
  // LDS offset of each of this wave's taps (uniform values): no tap arithmetic, and - without the mask, which is its own
  // instantiation - no branch inside the tap loop except in front of the second tap group's absent fifth tap, so the
  // compiler can hoist a tap's fragment reads above the previous tap's MFMAs (with a branch per tap every tap paid the
  // LDS latency: 4 reads -> wait -> 4 MFMAs, 3.4x the MFMA time).
  int toff_t[TPG];
#pragma unroll
  for (int tt = 0; tt < TPG; ++tt) {
    const int t = (t0 + tt) < 9 ? (t0 + tt) : 8;
    toff_t[tt] = __builtin_amdgcn_readfirstlane(((t / 3) * HW2 + (t % 3)) * XS);
  }
  const bool last_ok = __builtin_amdgcn_readfirstlane(t0 + TPG - 1) < 9;
  // Software-pipelined by hand over the 4 x TPG (K step, tap) stages: the fragment reads of stage s+1 are issued before the
  // MFMAs of stage s and pinned there with scheduling barriers (left to itself the compiler emits, per tap, 4 reads -> wait
  // -> 4 MFMAs, and every tap pays the LDS latency).
  auto compute = [&]() {
    bf16x8_t af[2][WM], bfr[2][WN];
    auto load_b = [&](int ks, bf16x8_t (&b)[WN]) {
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = tr8(lds_y, yb[ks][0] + 16 * j, yb[ks][1] + 16 * j);
    };
    auto load_a = [&](int ks, int tt, bf16x8_t (&a)[WM]) {
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = tr8(lds_x, xb[ks][0] + toff_t[tt] + 16 * i, xb[ks][1] + toff_t[tt] + 16 * i);
    };
    load_b(0, bfr[0]);
    load_a(0, 0, af[0]);
#pragma unroll
    for (int st = 0; st < 4 * TPG; ++st) {
      const int ks = st / TPG, tt = st - ks * TPG;
      if (st + 1 < 4 * TPG) {
        const int ks1 = (st + 1) / TPG, tt1 = (st + 1) - ks1 * TPG;
        if (tt1 == 0) load_b(ks1, bfr[ks1 & 1]);
        load_a(ks1, tt1, af[(st + 1) & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool skip = (NTG > 1 && tt == TPG - 1 && !last_ok) ||      // the second tap group has four taps
                        (MASKED && !((tmask >> (t0 + tt)) & 1u));        // masked taps keep a zero accumulator
      if (!skip) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[tt][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks & 1][j], af[st & 1][i], acc[tt][i][j], 0, 0, 0);   // D^T: lane holds 4 consecutive n
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (!PF) {
    for (int grp = g_begin; grp < g_end; ++grp) {
      const int rel = grp - g_begin;
      if ((rel & (TB - 1)) == 0) fill_batch(grp);
      __syncthreads();  // patch table ready; also: previous iteration's LDS reads are done
      uint4 rx[X_IT], ry[Y_IT];
      uint32_t mx, my;
      load_xy(s_tab[rel & (TB - 1)], rx, ry, mx, my);
#pragma unroll
      for (int it = 0; it < X_IT; ++it)
        if (x_geo[it] >= 0)
          *reinterpret_cast<uint4*>(&lds_x[((tid + NTHR * it) / XCH) * XS + xq * 8]) = (mx >> it) & 1u ? rx[it] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int it = 0; it < Y_IT; ++it)
        if (tid + NTHR * it < 128 * YCH)
          *reinterpret_cast<uint4*>(&lds_y[((tid + NTHR * it) / YCH) * YS + yq * 8]) = (my >> it) & 1u ? ry[it] : make_uint4(0, 0, 0, 0);
      __syncthreads();
      compute();
    }
  } else {
    uint4 rx[X_IT], ry[Y_IT];
    uint32_t mx = 0u, my = 0u;
    if (g_begin < g_end) {
      fill_batch(g_begin);
      __syncthreads();
      load_xy(s_tab[0], rx, ry, mx, my);
    }
    for (int grp = g_begin; grp < g_end; ++grp) {
      const int rel1 = grp + 1 - g_begin;                 // the next group's slot; a new batch overwrites entries whose last
      int (*nxt)[8] = s_tab[rel1 & (TB - 1)];             // readers finished before the barrier that ended the previous group
#pragma unroll
      for (int it = 0; it < X_IT; ++it)
        if (x_geo[it] >= 0)
          *reinterpret_cast<uint4*>(&lds_x[((tid + NTHR * it) / XCH) * XS + xq * 8]) = (mx >> it) & 1u ? rx[it] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int it = 0; it < Y_IT; ++it)
        if (tid + NTHR * it < 128 * YCH)
          *reinterpret_cast<uint4*>(&lds_y[((tid + NTHR * it) / YCH) * YS + yq * 8]) = (my >> it) & 1u ? ry[it] : make_uint4(0, 0, 0, 0);
      const bool more = grp + 1 < g_end;
      if (more && (rel1 & (TB - 1)) == 0) fill_batch(grp + 1);
      __syncthreads();      // this group's tile and the next group's patch table are visible
      if (more) {
        load_xy(nxt, rx, ry, mx, my);
      }
      compute();
      __syncthreads();      // everyone is done reading the tile before the next stores
    }
  }

  // The MFMA runs with dy as its row operand, so a lane holds D[n_local = 4g + r][m_local = li]: four consecutive output
  // channels of one input channel = one 16-byte store into the [9][Ma][Nb] slab (80 -> 20 store instructions per lane for
  // the 64x64 tile).  With a partials workspace every split stores its own slab with plain stores (summed by
  // wgrad_finish_kernel); without one it falls back to atomics.
  float* dst = p.ws ? p.ws + (int64_t)bx * 9 * p.Ma * p.Nb : p.out;
  const bool vec_ok = (p.Nb & 3) == 0;
#pragma unroll
  for (int tt = 0; tt < TPG; ++tt)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + tt;
        if (NTG > 1 && t >= 9) continue;
        const int m = m0 + wm * 16 * WM + i * 16 + li;
        const int n = n0 + wn * 16 * WN + j * 16 + g * 4;
        if (m >= p.Ma || n >= p.Nb) continue;
        float* q = dst + ((int64_t)t * p.Ma + m) * p.Nb + n;
        if (p.ws && vec_ok && n + 3 < p.Nb) {
          *reinterpret_cast<float4*>(q) = make_float4(acc[tt][i][j][0], acc[tt][i][j][1], acc[tt][i][j][2], acc[tt][i][j][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.Nb) {
              if (p.ws) q[r] = acc[tt][i][j][r];
              else atomicAdd(q + r, acc[tt][i][j][r]);
            }
        }
      }
}

// Returns 1 and launches if the geometry fits, 0 if the caller must use the per-tap kernel.
struct WgHaloGeom {
  const bf16_t* x; const bf16_t* dy; float* out; WgMap map;
  int B, H, W, d, Ma, Nb, ldx, lddy;
  int tconv_cls = -1;      // >= 0: parity class 2a+b of a stride-2 transposed conv (x = dy tensor at 2H x 2W, dy = x tensor; see the views)
  uint32_t tapmask = 0x1ff;
};

// Launches njobs (1..4) weight gradients of identical (B,H,W) - dilations may differ as long as the group counts match, channel counts may
// differ as long as the jobs take the same tile shape and the same number of tiles (the stem's three convs: 32x32, 16x32 and 8x16 channels are
// one 32x32 tile each) - as one grid; returns 0 if the geometry does not fit (the caller then launches the jobs one at a time).
static int wgrad_halo_launch(const WgHaloGeom* gm, int njobs, float* ws, int64_t ws_floats, hipStream_t s) {
  static const int disabled = getenv("USSEG_NO_HALO") != nullptr;
  if (disabled || njobs < 1 || njobs > 4) return 0;
  WgHaloMulti P;
  const int Ma = gm[0].Ma, Nb = gm[0].Nb;     // job 0 decides the tile shape; every job must land in the same shape class with the same tile count
  auto shape_of = [](int ma, int nb) { return (ma <= 32 && nb <= 32) ? 1 : (nb <= 16 ? 2 : (nb <= 32 ? 3 : 0)); };
  auto tiles_of = [&](int ma, int nb) {
    const int sh = shape_of(ma, nb), bm = sh == 1 ? 32 : 64, bn = sh == 0 ? 64 : (sh == 2 ? 16 : 32);
    return ((ma + bm - 1) / bm) * ((nb + bn - 1) / bn);
  };
  int64_t slab_max = 0, slab_sum = 0;
  for (int j = 0; j < njobs; ++j) {
    const WgHaloGeom& q = gm[j];
    if (q.B != gm[0].B || q.H != gm[0].H || q.W != gm[0].W) return 0;
    if (shape_of(q.Ma, q.Nb) != shape_of(Ma, Nb) || tiles_of(q.Ma, q.Nb) != tiles_of(Ma, Nb)) return 0;
    if ((q.Ma != Ma || q.Nb != Nb) && (usseg_tap_mask.group_ch || q.tconv_cls >= 0)) return 0;
    slab_sum += (int64_t)9 * q.Ma * q.Nb;
    if ((int64_t)9 * q.Ma * q.Nb > slab_max) slab_max = (int64_t)9 * q.Ma * q.Nb;
    const int d = q.d;
    if (d < 1 || q.H % d || q.W % d) return 0;
    const int Hl = q.H / d, Wl = q.W / d;
    int PW;
    if (Wl % 16 == 0) PW = 16;
    else if (Wl == 8 || Wl == 4) PW = Wl;
    else return 0;
    int PH = 128 / PW;
    if (PH > Hl) PH = Hl;
    if (Hl % PH || (PH * PW) % 16) return 0;
    const int NV = 128 / (PH * PW);
    if (NV * (PH + 2) * (PW + 2) > 288) return 0;
    WgHaloParams& p = P.job[j];
    p = {};
    p.x = q.x; p.dy = q.dy; p.out = q.out;
    p.B = q.B; p.H = q.H; p.W = q.W; p.d = d; p.Hl = Hl; p.Wl = Wl; p.PH = PH; p.PW = PW; p.NV = NV;
    p.tiles_x = Wl / PW;
    p.tiles_per_v = (Hl / PH) * p.tiles_x;
    p.npatches = q.B * d * d * p.tiles_per_v;
    if (q.tconv_cls >= 0) {
      p.xvd = 2; p.xvH = 2 * q.H; p.xvW = 2 * q.W; p.xva = q.tconv_cls >> 1; p.xvb = q.tconv_cls & 1;
      p.yvd = 1; p.yvH = q.H; p.yvW = q.W; p.yva = p.yvb = 0;
      p.mask_ch = 1 << 30;                       // one class for every channel tile: tapmask[0]
      p.tapmask[0] = (uint16_t)q.tapmask;
    } else {
      p.xvd = p.yvd = d; p.xvH = p.yvH = q.H; p.xvW = p.yvW = q.W; p.xva = p.xvb = p.yva = p.yvb = 0;
    }
    p.ngroups = (p.npatches + NV - 1) / NV;
    p.ldx = q.ldx; p.lddy = q.lddy; p.Ma = q.Ma; p.Nb = q.Nb;
    if (p.ngroups != P.job[0].ngroups) return 0;
    if ((int64_t)q.B * p.xvH * p.xvW * q.ldx >= (1ll << 31) || (int64_t)q.B * p.yvH * p.yvW * q.lddy >= (1ll << 31)) return 0;   // 32-bit element offsets
    if ((int64_t)q.B * p.xvH * p.xvW >= (1ll << 23) || (int64_t)q.B * p.yvH * p.yvW >= (1ll << 23) || p.npatches >= (1 << 20)) return 0;   // signed 24-bit multiplies / fdiv() in the kernel's decodes
  }
  if (ws) ws = usseg_defer_wgrad_ws(s, ws, ws_floats, &ws_floats);   // deferred finishing: a private region of the step's workspace
  // tile shape by channel counts: 64x64, 32x32 (both small), 64x16 / 64x32 (few output channels, e.g. the decoder branches)
  int shape = 0;
  if (Ma <= 32 && Nb <= 32) shape = 1;
  else if (Nb <= 16) shape = 2;
  else if (Nb <= 32) shape = 3;
  const int bm = shape == 1 ? 32 : 64, bn = shape == 0 ? 64 : (shape == 2 ? 16 : 32);
  const int tm = (Ma + bm - 1) / bm;
  const int ntn = (Nb + bn - 1) / bn;
  const int tiles = tm * ntn;
  const int ngroups = P.job[0].ngroups;
  // split-K over pixel chunks: aim for ~256 workgroups (over all jobs; 512 until round 3) with >= 2 chunks each.  With a workspace every
  // split writes its own partial slab (plain stores) and the finishing kernel sums them; fp32 atomics into a 9*64*64 tile
  // from hundreds of workgroups run at the ~1.3 TB/s atomic rate and cost more than the MFMAs (no-workspace fallback only).
  const int64_t slab = slab_max;             // per-job slabs may be smaller: the caps below are sized by the largest
  // traffic guard (rocprofv3 FETCH/WRITE_SIZE showed the partial slabs costing ~2 GB/step): keep the slabs written and
  // re-read within 2x the bytes of the operands themselves (swept 1/2/3/4/8/16 with the multi-job launches and the
  // register prefetch in place: 2x is the minimum, 4.95 vs 5.05 ms/step at 8x; sending the deep layers back to the
  // per-tap kernel was slower).
  int64_t in_bytes = 0;
  for (int j = 0; j < njobs; ++j) in_bytes += (int64_t)gm[j].B * gm[j].H * gm[j].W * (gm[j].Ma + gm[j].Nb) * 2;
  in_bytes /= njobs;
  const int64_t slab_bytes = slab * 4;
  static const int wg_target = getenv("USSEG_WG_TARGET") ? atoi(getenv("USSEG_WG_TARGET")) : 256;   // round 3 sweep (128 / 192 / 256 / 384 / 512 / 768): 256 is -25 us on Arch B (half the slabs for wgrad_finish), neutral on A / T
  int splits = (wg_target + tiles * njobs - 1) / (tiles * njobs);
  int max_splits = (ngroups + 1) / 2;
  static const int slab_cap = getenv("USSEG_SLAB_CAP") ? atoi(getenv("USSEG_SLAB_CAP")) : 2;
  const int64_t traffic_cap = slab_cap * in_bytes / slab_bytes < 4 ? 4 : slab_cap * in_bytes / slab_bytes;
  if (max_splits > traffic_cap) max_splits = (int)traffic_cap;
  if (!ws) max_splits = (ngroups + 15) / 16;
  else if ((int64_t)max_splits * slab_sum > ws_floats) max_splits = (int)(ws_floats / slab_sum);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  {
    // Wave quantisation: the 64x64 shape runs one workgroup per CU (8 waves), the others two, so a launch of 528 workgroups
    // is three rounds of 256 where 504 would be two.  Around the target, pick the split count with the cheapest
    // rounds x (groups per workgroup + fixed cost) estimate (fixed cost ~ 3 group times: prologue + slab epilogue).
    static const int quant = getenv("USSEG_WG_QUANT") ? atoi(getenv("USSEG_WG_QUANT")) : 1;
    if (quant && max_splits > 1) {
      const int slots = 256 * (shape == 0 ? 1 : 2);
      const int64_t tj = (int64_t)tiles * njobs;
      int best = splits;
      double best_cost = 1e30;
      const int lo = splits / 2 > 1 ? splits / 2 : 1, hi = splits * 2 < max_splits ? splits * 2 : max_splits;
      for (int sp = lo; sp <= hi; ++sp) {
        const int g = (ngroups + sp - 1) / sp;
        const int real = (ngroups + g - 1) / g;
        const double rounds = (double)((tj * real + slots - 1) / slots);
        const double cost = rounds * (g + 3.0) + 0.02 * real;   // mild preference for fewer slabs at equal time
        if (cost < best_cost) { best_cost = cost; best = sp; }
      }
      splits = best;
    }
  }
  bool mapped = false;
  for (int j = 0; j < njobs; ++j) mapped = mapped || gm[j].map.nblocks;
  // a mapped destination is scattered by the finishing kernel, so it always goes through a slab
  const bool use_ws = ws && (splits > 1 || mapped) && (int64_t)splits * slab_sum <= ws_floats;
  if (mapped && !use_ws) return 0;
  const int gpb = (ngroups + splits - 1) / splits;
  splits = (ngroups + gpb - 1) / gpb;
  {
    int64_t off = 0;
    for (int j = 0; j < njobs; ++j) {
      P.job[j].ntiles_n = (gm[j].Nb + bn - 1) / bn;
      P.job[j].groups_per_block = gpb;
      P.job[j].ws = use_ws ? ws + off : nullptr;
      off += (int64_t)splits * 9 * gm[j].Ma * gm[j].Nb;
    }
  }
  const bool masked = usseg_tap_mask.group_ch != 0 || gm[0].tconv_cls >= 0;
  if (usseg_tap_mask.group_ch) {
    if (njobs != 1 || usseg_tap_mask.group_ch % bn) return 0;      // an output-channel tile must lie inside one class
    P.job[0].mask_ch = usseg_tap_mask.group_ch;
    for (int c = 0; c < 4; ++c) P.job[0].tapmask[c] = usseg_tap_mask.mask[c];
  }
  const dim3 grid(splits, tiles, njobs);
  static const int xcd_env = getenv("USSEG_WGHALO_XCD") ? atoi(getenv("USSEG_WGHALO_XCD")) : 1;
  P.xcd_remap = xcd_env && tiles * njobs > 1 && (int64_t)splits * tiles * njobs < (1ll << 20);     // (fdiv() in the remap)
  static const int dbg = getenv("USSEG_WGRAD_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[wgrad_halo] B %d H %d W %d d %d Ma %d Nb %d jobs %d shape %d tiles %d ngroups %d splits %d gpb %d max_splits %d traffic_cap %lld slab_MB %.2f in_MB %.2f\n",
                   gm[0].B, gm[0].H, gm[0].W, gm[0].d, Ma, Nb, njobs, shape, tiles, ngroups, splits, gpb, max_splits, (long long)traffic_cap, slab_bytes / 1e6, in_bytes / 1e6);
  const int slot = usseg_prof_start(2, s);
  static const int pf = getenv("USSEG_WGRAD_PF") ? atoi(getenv("USSEG_WGRAD_PF")) : 1;
  if (shape == 1) {
    if (pf) { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<2, 1, 1, true, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<2, 1, 1, true, 1, false>), grid, dim3(256), 0, s, P); }
    else { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<2, 1, 1, false, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<2, 1, 1, false, 1, false>), grid, dim3(256), 0, s, P); }
  } else if (shape == 2) {
    if (pf) { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 1, true, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 1, true, 1, false>), grid, dim3(256), 0, s, P); }
    else { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 1, false, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 1, false, 1, false>), grid, dim3(256), 0, s, P); }
  } else if (shape == 3) {
    if (pf) { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 2, true, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 2, true, 1, false>), grid, dim3(256), 0, s, P); }
    else { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 2, false, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<4, 1, 2, false, 1, false>), grid, dim3(256), 0, s, P); }
  } else {
    static const int tg8 = getenv("USSEG_WGRAD_8W") ? atoi(getenv("USSEG_WGRAD_8W")) : 1;
    if (tg8) { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<2, 2, 2, true, 2, true>), grid, dim3(512), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<2, 2, 2, true, 2, false>), grid, dim3(512), 0, s, P); }
    else { if (masked) hipLaunchKernelGGL((wgrad_halo_kernel<2, 2, 2, false, 1, true>), grid, dim3(256), 0, s, P); else hipLaunchKernelGGL((wgrad_halo_kernel<2, 2, 2, false, 1, false>), grid, dim3(256), 0, s, P); }
  }
  if (use_ws)
    for (int j = 0; j < njobs; ++j) usseg_launch_wgrad_finish(P.job[j].ws, splits, (int64_t)9 * gm[j].Ma * gm[j].Nb, gm[j].out, gm[j].map, gm[j].Ma, gm[j].Nb, s);
  usseg_prof_stop(2, slot, s);
  return 1;
}

// Returns 1 and launches if the geometry fits, 0 if the caller must use the per-tap kernel.
int usseg_try_launch_wgrad_halo(const bf16_t* x, const bf16_t* dy, float* out, const WgMap& map, int B, int H, int W, int d, int Ma, int Nb,
                                int ldx, int lddy, float* ws, int64_t ws_floats, hipStream_t s) {
  WgHaloGeom g = {x, dy, out, map, B, H, W, d, Ma, Nb, ldx, lddy};
  return wgrad_halo_launch(&g, 1, ws, ws_floats, s);
}

int usseg_try_launch_wgrad_halo_multi(int njobs, const UssegWgradJob* jobs, float* ws, int64_t ws_floats, hipStream_t s) {
  if (njobs < 1 || njobs > 4) return 0;
  WgHaloGeom g[4];
  for (int j = 0; j < njobs; ++j) {
    const UssegWgradJob& q = jobs[j];
    const UssegConvDesc& d = q.desc;
    if (d.ksize != 3) return 0;
    g[j] = {(const bf16_t*)q.x, (const bf16_t*)q.dy, q.dw, WgMap(), d.B, d.H, d.W, d.dilation, d.Cin, d.Cout, d.ldx, d.ldy};
    if (!wg_map_fill(g[j].map, q.dst)) return 0;
  }
  return wgrad_halo_launch(g, njobs, ws, ws_floats, s);
}

// Weight gradient of a stride-2 'same' Conv2DTranspose (k = 3 or 4) on the halo-tile kernel: its four output-parity classes are
// four 3x3 stride-1 weight gradients between x and dy seen through a stride-2 view,
//   dW[kh][kw][co][ci] = sum_g x[g][ci] * dy[2g + (kh-pad, kw-pad)][co] = sum_g x[g][ci] * dy_(a,b)[g + (th-1, tw-1)][co],
//   kh = a + pad - 2 + 2*th (likewise kw), so class (a, b) owns the taps whose (kh, kw) fall inside the k x k kernel: a 2x2
// (or smaller) subset of the 3x3 - the kernel's tap mask - that sits at stride 2 in the Keras [k,k,Cout,Cin] variable - the map's
// two tap strides.  One launch, blockIdx.z = class; both operands staged once per 128 pixels for all taps instead of once per tap
// (the per-tap kernel moves 16x the operand bytes for k = 4).  `map`: kernel axes (m, n) = (output channel, input channel).
int usseg_try_launch_tconv_wgrad_halo(const bf16_t* x, const bf16_t* dy, const WgMap& map, int B, int H, int W, int Cin, int Cout, int ldx,
                                      int lddy, int k, float* ws, int64_t ws_floats, hipStream_t s) {
  // Built and parity-tested, off by default: with 4 of 9 taps live per class the 64x64 halo tile is staging-bound and loses to the
  // per-tap LDS-DMA kernel with 128x128 tiles (Arch A 9.20 vs 8.78 ms, Arch B 4.32 vs 4.22 ms per step).
  const char* en = getenv("USSEG_TCONV_HALO");   // read per call: the parity test flips it inside one process
  const int enabled = en ? atoi(en) : 0;
  if (!enabled || map.nblocks < 1 || (k != 3 && k != 4)) return 0;
  const int pad = k == 4 ? 1 : 0;
  WgHaloGeom g[4];
  for (int c = 0; c < 4; ++c) {
    const int a = c >> 1, b = c & 1;
    uint32_t mask = 0;
    for (int th = 0; th < 3; ++th)
      for (int tw = 0; tw < 3; ++tw) {
        const int kh = a + pad - 2 + 2 * th, kw = b + pad - 2 + 2 * tw;
        if (kh >= 0 && kh < k && kw >= 0 && kw < k) mask |= 1u << (3 * th + tw);
      }
    g[c] = {dy, x, nullptr, map, B, H, W, 1, Cout, Cin, lddy, ldx};
    g[c].tconv_cls = c;
    g[c].tapmask = mask;
    g[c].map.tapmask = mask;
    for (int i = 0; i < map.nblocks; ++i) {
      WgBlock& u = g[c].map.blk[i];
      const int64_t sk = map.blk[i].sT;           // stride of one Keras tap (kh*k + kw)
      u.dst = map.blk[i].dst + ((int64_t)(a + pad - 2) * k + (b + pad - 2)) * sk;
      u.sTr = 2 * k * sk;
      u.sT = 2 * sk;
    }
  }
  return wgrad_halo_launch(g, 4, ws, ws_floats, s);
}
