// Finishing reductions of the two-pass (no-atomics) kernels, and their DEFERRED, batched form.
//
// Every norm backward / bias column sum / weight gradient ends in a small "sum the per-workgroup partial rows (or the
// split-K slabs) into the variable's gradient" kernel.  Alone each is a 4-14 us launch (one dependent L2 round trip plus
// launch overhead) and a training step has ~90 of them: 0.6 ms of a 6.2 ms step.  Their results are only read by the
// optimiser, so between usseg_defer_begin() and usseg_defer_flush() on a stream the library queues them instead, hands
// every producer its own region of the caller's workspace (bump allocation, so partials stay intact) and runs the whole
// queue as a few batched launches (job table in the kernel arguments, blockIdx.y / .z = job).
#include <vector>
#include "common.h"

// ------------------------------------------------------------------------------------------ per-channel partial rows
// dst_k[g*C + c] += scale * sum_{j<nb} ws[((g*nb + j)*K + k)*Cp + c]   for k < K (dst_k may be NULL)
struct RJob {
  const float* ws;
  float *d0, *d1, *d2;
  int32_t groups, nb, K, Cp, C, nblocks;
  float scale;
  int32_t overwrite;   // 1: dst = scale*sum (no pre-zeroed destination needed), 0: dst += scale*sum
};
struct RBatch {
  int32_t njobs, pad;
  int32_t start[60];   // first flat workgroup id of each job (start[njobs] = total): no empty workgroups in the grid (as WBatch)
  RJob job[59];        // 59 x 64 B + 248 B: inside the 4 KB of kernel arguments
};
static_assert(sizeof(RBatch) <= 4096, "RBatch must fit the kernel-argument segment");

// workgroup = 16 outputs x 16 row slices: the nb partial rows of an output are summed by 16 threads in parallel
// (a serial walk over 512 rows is a 512-deep chain of dependent loads: 45 us; this form is ~3 us)
__device__ __forceinline__ void reduce_finish_body(const RJob& q, int bx, int nbx) {
  __shared__ float s_part[16][17];
  const int total = q.groups * q.K * q.C;
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (int base = bx * 16; base < total; base += nbx * 16) {
    const int i = base + o;
    float s = 0.f;
    int c = 0, k = 0, g = 0;
    if (i < total) {
      c = i % q.C;
      k = (i / q.C) % q.K;
      g = i / (q.C * q.K);
      const float* src = q.ws + ((int64_t)g * q.nb * q.K + k) * q.Cp + c;
      const int64_t rs = (int64_t)q.K * q.Cp;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;    // independent chains: the loads of a slice are in flight together (eight per trip while the rows last:
      int j = sl;                            //  the 1024 partial rows of a full-chip norm backward were 16 dependent trips of four)
      for (; j + 112 < q.nb; j += 128) {
        const float a0 = src[j * rs], a1 = src[(j + 16) * rs], a2 = src[(j + 32) * rs], a3 = src[(j + 48) * rs];
        const float a4 = src[(j + 64) * rs], a5 = src[(j + 80) * rs], a6 = src[(j + 96) * rs], a7 = src[(j + 112) * rs];
        s += a0; s1 += a1; s2 += a2; s3 += a3; s += a4; s1 += a5; s2 += a6; s3 += a7;
      }
      for (; j + 48 < q.nb; j += 64) {
        s += src[j * rs]; s1 += src[(j + 16) * rs]; s2 += src[(j + 32) * rs]; s3 += src[(j + 48) * rs];
      }
      for (; j < q.nb; j += 16) s += src[j * rs];
      s += s1 + s2 + s3;
    }
    s_part[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += s_part[r][o];
      float* dst = k == 0 ? q.d0 : (k == 1 ? q.d1 : q.d2);
      if (dst) dst[g * q.C + c] = q.overwrite ? q.scale * t : dst[g * q.C + c] + q.scale * t;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void reduce_finish_kernel(const RJob q) { reduce_finish_body(q, blockIdx.x, gridDim.x); }
// flat grid: a (max blocks x jobs) grid of the step's ~50 jobs was 36 850 workgroups of which ~4 000 had work (one job has 670 blocks, most
// have under 100) - dispatching the empty ones was most of the launch's 21 us
__global__ __launch_bounds__(256) void reduce_finish_batched_kernel(const RBatch b) {
  // this workgroup's job: lane i holds start[i + 1]; the jobs whose range ends at or before this workgroup form a prefix (one vector load and a
  // ballot instead of a walk of up to 58 dependent scalar loads - 2-3 us for the workgroups of the last jobs)
  const int lane = threadIdx.x & 63;
  const int nxt = lane + 1 < b.njobs ? b.start[lane + 1] : 0x7fffffff;
  const int j = __builtin_amdgcn_readfirstlane(__popcll(__ballot((int)blockIdx.x >= nxt)));
  const RJob& q = b.job[j];
  reduce_finish_body(q, blockIdx.x - b.start[j], q.nblocks);
}

// ------------------------------------------------------------------------------------------ split-K slabs of a weight gradient
// out (identity) or the mapped variables += sum_s ws[s][i]  (n floats per slab, n % 4 == 0)
// Workgroup = (256 / SL) float4 outputs x SL slab slices, SL in {4, 16, 64}: a small output with many slabs gets more
// slices per output instead of more workgroups per output, so every output element is finished by ONE workgroup in a fixed
// order - no atomics, bitwise reproducible (the earlier form split the slab axis over workgroups that met through float
// atomics: the stem's weight gradients changed in the last bits from run to run).
struct WJob {
  const float* ws;
  float* out;
  int64_t n4;
  int32_t splits, Ma, Nb, gx, sl_log2, pad;
  WgMap map;
};
struct WBatch {
  int32_t njobs, pad;
  int32_t start[16];   // first flat workgroup id of each job (start[njobs] = total): no empty workgroups in the grid
  WJob job[14];
};

__device__ __forceinline__ void wgrad_finish_body(const WJob& q, int bx) {
  __shared__ float4 s_part[256];
  const float4* w4 = reinterpret_cast<const float4*>(q.ws);
  const int SL = 1 << q.sl_log2, O = 256 >> q.sl_log2;
  const int o = threadIdx.x & (O - 1), sl = threadIdx.x >> (8 - q.sl_log2);
  for (int64_t base = (int64_t)bx * O; base < q.n4; base += (int64_t)q.gx * O) {
    const int64_t i = base + o;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < q.n4) {
#pragma unroll 4
      for (int s = sl; s < q.splits; s += SL) {
        float4 v = w4[(int64_t)s * q.n4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    s_part[threadIdx.x] = a;
    __syncthreads();
    for (int h = SL >> 1; h >= 1; h >>= 1) {          // fixed tree over the slices
      if (sl < h) {
        float4 u = s_part[threadIdx.x], w = s_part[threadIdx.x + h * O];
        u.x += w.x; u.y += w.y; u.z += w.z; u.w += w.w;
        s_part[threadIdx.x] = u;
      }
      __syncthreads();
    }
    if (sl == 0 && i < q.n4) {
      const float4 t = s_part[o];
      if (q.map.nblocks == 0) {
        float* d = q.out + i * 4;
        d[0] += t.x; d[1] += t.y; d[2] += t.z; d[3] += t.w;
      } else {   // scatter into the framework's variables (logical channels, Keras strides); the four elements share (tap, ci)
        const int64_t e = i * 4;
        const int tap = (int)(e / ((int64_t)q.Ma * q.Nb));
        const int rem = (int)(e - (int64_t)tap * q.Ma * q.Nb);
        const int mi = rem / q.Nb, n = rem - mi * q.Nb;
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* d = wg_map_dst(q.map, nullptr, 0, tap, mi, n + r);
          if (d) *d += v[r];
        }
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const WJob q) { wgrad_finish_body(q, blockIdx.x); }
__global__ __launch_bounds__(256) void wgrad_finish_batched_kernel(const WBatch b) {
  const int lane = threadIdx.x & 63;      // (job lookup as in reduce_finish_batched_kernel)
  const int nxt = lane + 1 < b.njobs ? b.start[lane + 1 < 16 ? lane + 1 : 15] : 0x7fffffff;
  const int j = __builtin_amdgcn_readfirstlane(__popcll(__ballot((int)blockIdx.x >= nxt)));
  wgrad_finish_body(b.job[j], blockIdx.x - b.start[j]);
}

// ------------------------------------------------------------------------------------------ deferral context (host)
struct DeferCtx {
  hipStream_t s;
  float* rws; int64_t rcap, rused;
  float* wws; int64_t wcap, wused;
  std::vector<RJob> rj;
  std::vector<WJob> wj;
};
static thread_local std::vector<DeferCtx> g_ctx;   // per host thread: one entry per stream between begin and end (a handful at most)

static DeferCtx* find_ctx(hipStream_t s) {
  for (auto& c : g_ctx)
    if (c.s == s) return &c;
  return nullptr;
}

// The batched kernels read-modify-write their destinations without atomics and the jobs of one launch run concurrently, so
// two jobs that share a destination (a gradient variable produced twice before the flush) must not share a launch: a
// batch is closed in front of the first job that repeats a destination and the next launch is stream-ordered behind it.
static bool rjob_shares_dst(const RBatch& b, const RJob& q) {
  for (int j = 0; j < b.njobs; ++j) {
    const float* have[3] = {b.job[j].d0, b.job[j].d1, b.job[j].d2};
    const float* want[3] = {q.d0, q.d1, q.d2};
    for (const float* h : have)
      for (const float* w : want)
        if (h && h == w) return true;
  }
  return false;
}
static const float* wjob_dst(const WJob& q, int b) { return q.map.nblocks == 0 ? (b == 0 ? q.out : nullptr) : (b < q.map.nblocks ? q.map.blk[b].dst : nullptr); }
static bool wjob_shares_dst(const WBatch& b, const WJob& q) {
  for (int j = 0; j < b.njobs; ++j)
    for (int x = 0; x < 4; ++x)
      for (int y = 0; y < 4; ++y) {
        const float* h = wjob_dst(b.job[j], x);
        if (h && h == wjob_dst(q, y)) return true;
      }
  return false;
}

static void flush_reduce(DeferCtx& c) {
  size_t i = 0;
  while (i < c.rj.size()) {
    RBatch b = {};
    int total = 0;
    while (i < c.rj.size() && b.njobs < 59 && !rjob_shares_dst(b, c.rj[i])) {
      b.job[b.njobs] = c.rj[i++];
      b.start[b.njobs] = total;
      total += b.job[b.njobs].nblocks;
      ++b.njobs;
    }
    b.start[b.njobs] = total;
    hipLaunchKernelGGL(reduce_finish_batched_kernel, dim3(total), dim3(256), 0, c.s, b);
  }
  c.rj.clear();
  c.rused = 0;
}
static void flush_wgrad(DeferCtx& c) {
  size_t i = 0;
  while (i < c.wj.size()) {
    WBatch b = {};
    int total = 0;
    while (i < c.wj.size() && b.njobs < 14 && !wjob_shares_dst(b, c.wj[i])) {
      b.job[b.njobs] = c.wj[i++];
      b.start[b.njobs] = total;
      total += b.job[b.njobs].gx;
      ++b.njobs;
    }
    b.start[b.njobs] = total;
    const int slot = usseg_prof_start(2, c.s);     // counted with the weight-gradient kernels it finishes
    hipLaunchKernelGGL(wgrad_finish_batched_kernel, dim3(total), dim3(256), 0, c.s, b);
    usseg_prof_stop(2, slot, c.s);
  }
  c.wj.clear();
  c.wused = 0;
}

// Workspace for `need` floats of per-workgroup partial rows: the caller's own ws, or (deferring) a private region.
float* usseg_defer_reduce_ws(hipStream_t s, float* caller_ws, int64_t need) {
  DeferCtx* c = find_ctx(s);
  if (!c || !c->rws || need > c->rcap) return caller_ws;
  if (c->rused + need > c->rcap) flush_reduce(*c);
  float* r = c->rws + c->rused;
  c->rused += (need + 63) & ~(int64_t)63;
  return r;
}

void usseg_launch_reduce_finish(const float* ws, int groups, int nb, int K, int Cp, int C, float scale, float* d0, float* d1, float* d2,
                                hipStream_t s, int overwrite) {
  RJob q = {};
  q.ws = ws; q.d0 = d0; q.d1 = d1; q.d2 = d2; q.groups = groups; q.nb = nb; q.K = K; q.Cp = Cp; q.C = C; q.scale = scale; q.overwrite = overwrite;
  int total = groups * K * C;
  int grid = (total + 15) / 16;
  if (grid > 1024) grid = 1024;
  q.nblocks = grid;
  DeferCtx* c = find_ctx(s);
  if (c && c->rws && ws >= c->rws && ws < c->rws + c->rcap) {   // partials live in the deferred region: queue
    c->rj.push_back(q);
    return;
  }
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(grid), dim3(256), 0, s, q);
}

// Slab workspace for a weight gradient: base pointer and capacity (floats) the launcher may use right now.
float* usseg_defer_wgrad_ws(hipStream_t s, float* caller_ws, int64_t caller_floats, int64_t* avail) {
  DeferCtx* c = find_ctx(s);
  if (!c || !c->wws) { *avail = caller_floats; return caller_ws; }
  if (c->wcap - c->wused < c->wcap / 4 && !c->wj.empty()) flush_wgrad(*c);   // keep at least a quarter free for the next producer
  *avail = c->wcap - c->wused;
  return c->wws + c->wused;
}

void usseg_launch_wgrad_finish(const float* ws, int splits, int64_t slab_floats, float* out, const WgMap& map, int Ma, int Nb, hipStream_t s) {
  WJob q = {};
  q.ws = ws; q.out = out; q.n4 = slab_floats / 4; q.splits = splits; q.Ma = Ma; q.Nb = Nb; q.map = map;
  // slices per output: 4, or 16 / 64 while that still leaves fewer than 256 workgroups and >= 2 slabs per slice
  int sl_log2 = 2;
  while (sl_log2 < 6 && (q.n4 + (256 >> sl_log2) - 1) / (256 >> sl_log2) < 256 && splits >= (2 << (sl_log2 + 2))) sl_log2 += 2;
  const int O = 256 >> sl_log2;
  int gx = (int)((q.n4 + O - 1) / O);
  if (gx > 2048) gx = 2048;
  q.gx = gx; q.sl_log2 = sl_log2;
  DeferCtx* c = find_ctx(s);
  if (c && c->wws && ws >= c->wws && ws < c->wws + c->wcap) {
    c->wused = (ws - c->wws) + (((int64_t)splits * slab_floats + 63) & ~(int64_t)63);   // commit the slabs this producer wrote
    c->wj.push_back(q);
    return;
  }
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(gx), dim3(256), 0, s, q);
}

extern "C" int usseg_defer_begin(usseg_stream_t stream, float* reduce_ws, int64_t reduce_floats, float* wgrad_ws, int64_t wgrad_floats) {
  USSEG_CHECK_ARG((!reduce_ws || reduce_floats > 0) && (!wgrad_ws || wgrad_floats > 0), "defer_begin: bad workspace sizes");
  hipStream_t s = (hipStream_t)stream;
  DeferCtx* c = find_ctx(s);
  if (c) { flush_reduce(*c); flush_wgrad(*c); }
  else { g_ctx.push_back(DeferCtx()); c = &g_ctx.back(); c->s = s; }
  c->rws = reduce_ws; c->rcap = reduce_ws ? reduce_floats : 0; c->rused = 0;
  c->wws = wgrad_ws; c->wcap = wgrad_ws ? wgrad_floats : 0; c->wused = 0;
  return USSEG_OK;
}
extern "C" int usseg_defer_flush(usseg_stream_t stream) {
  DeferCtx* c = find_ctx((hipStream_t)stream);
  if (c) { flush_reduce(*c); flush_wgrad(*c); }
  return usseg_check_launch("defer_flush");
}
extern "C" int usseg_defer_end(usseg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  for (size_t i = 0; i < g_ctx.size(); ++i)
    if (g_ctx[i].s == s) {
      flush_reduce(g_ctx[i]);
      flush_wgrad(g_ctx[i]);
      g_ctx.erase(g_ctx.begin() + i);
      break;
    }
  return usseg_check_launch("defer_end");
}
