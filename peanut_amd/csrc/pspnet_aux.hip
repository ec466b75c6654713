// HBM-bound helper kernels of the map-prediction forward: layout change, max-pool, pyramid pooling,
// pyramid upsample+concat and the final logits resize.  All are pure streaming kernels: wide
// (16-byte) coalesced accesses, grid-stride loops capped at 256 CUs x 8 workgroups, no LDS.
//
// Reference operators replaced (paths relative to /root/reference/prediction/mmseg):
//   models/backbones/resnet.py:638            nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
//   models/decode_heads/psp_head.py:38        nn.AdaptiveAvgPool2d(pool_scale)
//   models/decode_heads/psp_head.py:53-57     resize(..., mode='bilinear', align_corners=False)
//   models/decode_heads/psp_head.py:109       torch.cat(psp_outs, dim=1)
//   models/segmentors/encoder_decoder.py:75-79 resize(out, size=img.shape[2:], bilinear)
//   nav/agent/prediction.py:158               scipy.special.expit (optional fused sigmoid)
#include <algorithm>
#include <vector>

#include "common.h"
#include "options.h"

namespace peanut {

static inline unsigned grid_for(long long work_items, int block = 256) {
  long long g = (work_items + block - 1) / block;
  const long long cap = 256LL * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---- NCHW [B,C,H,W] -> NHWC [B,H,W,Cpad] with zero channel padding ----
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int C, long long HW, long long npix, int Cpad) {
  // a workgroup transposes 256 consecutive pixels x 16 channels through LDS: every channel-plane read is a fully
  // coalesced 256-byte run per wave, every NHWC store a fully contiguous 1 KiB run per wave.  (Measured alternatives,
  // profiles/r2h: one lane per pixel with strided 16-byte stores 0.235 ms, four pixels per lane with 16-byte plane
  // loads and 256-byte-strided stores 0.45 ms; this form 0.186 ms.)
  __shared__ float t[256][17];
  const long long p0 = (long long)blockIdx.x * 256;
  for (int c0 = 0; c0 < Cpad; c0 += 16) {
    const long long pix = p0 + threadIdx.x;
    if (pix < npix) {
      const long long b = pix / HW, hw = pix - b * HW;
      const float* src = x + (b * C) * HW + hw;
#pragma unroll
      for (int j = 0; j < 16; ++j) t[threadIdx.x][j] = (c0 + j < C) ? src[(long long)(c0 + j) * HW] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = k * 256 + threadIdx.x;          // float4 index inside the 256 x 16 tile
      const int px = q >> 2, c4 = (q & 3) * 4;
      if (p0 + px < npix && c0 + c4 < Cpad)
        *reinterpret_cast<float4*>(y + (p0 + px) * Cpad + c0 + c4) = make_float4(t[px][c4], t[px][c4 + 1], t[px][c4 + 2], t[px][c4 + 3]);
    }
    __syncthreads();
  }
}

int launch_nchw_to_nhwc_pad(const float* x, float* y, int B, int C, int H, int W, int Cpad, hipStream_t s) {
  const long long HW = (long long)H * W, npix = HW * B;
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, y, C, HW, npix, Cpad);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("nchw_to_nhwc: ") + hipGetErrorString(e));
}

// ---- MaxPool 3x3 stride 2 pad 1 (NHWC) ----
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                           int W, int C, int Ho, int Wo, long long total) {
  const int groups = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    long long pix = i / groups;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const long long b = pix / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + iy) * W + ix) * C + g * 4);
        // NaN goes through, like torch's max_pool2d (fmaxf would drop it)
        m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x; m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
        m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z; m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
      }
    }
    *reinterpret_cast<float4*>(y + i * 4) = m;
  }
}

int launch_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
  if (C % 4) return fail(-2, "maxpool: C must be a multiple of 4");
  const long long total = (long long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, y, H, W, C, Ho, Wo, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("maxpool: ") + hipGetErrorString(e));
}

// ---- adaptive average pooling into every pyramid bin ----
// out is SCALE-MAJOR: [scale s][B][k_s*k_s][C] (row0_s = B * sum_{j<s} k_j^2), so that the 1x1 conv of
// scale s sees one contiguous [B*k_s^2, C] matrix.  One workgroup per (b, bin, 1024-channel slab): thread t owns 4 channels, loops the bin's pixels (coalesced 16-B
// reads across the wave: 64 lanes x 16 B = one 1-KiB row segment per pixel).
// Bin edges follow ATen's adaptive pooling: start = floor(i*n/k), end = ceil((i+1)*n/k).
// rows: row stride between the scales of the pooled / table / Q tensors (0: packed, scale s starts at row B * sum_{j<s} k_j^2)
struct PpmScales { int s[8]; int n; int rows; };
__device__ __forceinline__ PEANUT_NO_PK_F32 size_t  ppm_row0(const PpmScales& sc, int s, int cells_before, int B) {
  return sc.rows ? (size_t)s * sc.rows : (size_t)B * cells_before;
}

__global__ __launch_bounds__(256) void ppm_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                                       int W, int C, PpmScales sc, int nbins) {
  const int bin = blockIdx.x, b = blockIdx.y;
  const int c = (blockIdx.z * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  int k = 0, local = bin, before = 0, si = 0;
  for (int i = 0; i < sc.n; ++i) {
    const int kk = sc.s[i];
    if (local < kk * kk) { k = kk; si = i; break; }
    local -= kk * kk;
    before += kk * kk;
  }
  const int by = local / k, bx = local - by * k;
  const int y0 = (by * H) / k, y1 = ((by + 1) * H + k - 1) / k;
  const int x0 = (bx * W) / k, x1 = ((bx + 1) * W + k - 1) / k;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yy) * W + xx) * C + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  const float cnt = (float)((y1 - y0) * (x1 - x0));   // ATen: sum / count
  acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt;
  const size_t row = ppm_row0(sc, si, before, gridDim.y) + (size_t)b * k * k + local;
  *reinterpret_cast<float4*>(out + row * C + c) = acc;
}

int launch_ppm_pool(const float* x, float* out, int B, int H, int W, int C, const int* scales, int nscales,
                    hipStream_t s, int scale_rows) {
  if (nscales > 8 || C % 4) return fail(-2, "ppm_pool: unsupported configuration");
  PpmScales sc;
  sc.n = nscales;
  sc.rows = scale_rows;
  int nbins = 0;
  for (int i = 0; i < nscales; ++i) { sc.s[i] = scales[i]; nbins += scales[i] * scales[i]; }
  const dim3 grid(nbins, B, (C / 4 + 255) / 256);
  hipLaunchKernelGGL(ppm_pool_kernel, grid, dim3(256), 0, s, x, out, H, W, C, sc, nbins);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("ppm_pool: ") + hipGetErrorString(e));
}

// ---- pyramid pooling, two-pass form: the feature map is read ONCE ----
// Pass 1 (one workgroup per (row y, image b, 1024-channel slab)): for every pyramid scale and every
// bin column, the sum over that bin's x-range of row y  -> rowsum[b][y][slot][C]  (slot = scale-major
// bin-column index, at most PPM_MAX_SLOTS = sum of the scales).  Pass 2 adds a bin's rows and divides.
// The single-pass kernel above stays as the fallback for exotic scale sets.
#define PPM_MAX_SLOTS 16
struct PpmSlots { short x0[PPM_MAX_SLOTS], x1[PPM_MAX_SLOTS]; int n; };

// Row pass.  The bins of all scales cut a row into at most 2 * PPM_MAX_SLOTS segments (the union of their edges);
// a segment lies inside at most one bin-column per scale.  Per segment: a branch-free run of 16-byte loads summed in
// registers, then ONE round of (wave-uniform) slot tests -- per pixel the loop is a load and four adds.
struct PpmSegs { short x0[2 * PPM_MAX_SLOTS], x1[2 * PPM_MAX_SLOTS]; unsigned mask[2 * PPM_MAX_SLOTS]; int n; };
// Row groups (round 4): runs of at most PPM_GROUP_ROWS consecutive rows that lie inside ONE bin-row of every scale (between two
// consecutive y edges of the pyramid).  A workgroup of the row pass adds up a whole group, so the intermediate holds one
// partial per group instead of one per row (60 rows of a 480 x 480 map's feature map: 12 groups of 5) -- the second pass reads
// a fifth of the bytes, and the row pass writes a fifth.
#define PPM_MAX_GROUPS 128
#define PPM_GROUP_ROWS 5
struct PpmGroups { short y0[PPM_MAX_GROUPS], y1[PPM_MAX_GROUPS]; int n; };

__global__ __launch_bounds__(256) void ppm_rowsum_kernel(const float* __restrict__ x, float* __restrict__ rowsum, int H,
                                                         int W, int C, PpmSlots sl, PpmSegs sg, PpmGroups gr) {
  const int grp = blockIdx.x, b = blockIdx.y;
  const int c = (blockIdx.z * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  float4 acc[PPM_MAX_SLOTS];
#pragma unroll
  for (int s = 0; s < PPM_MAX_SLOTS; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int y = gr.y0[grp]; y < gr.y1[grp]; ++y) {
  const float* row = x + (((size_t)b * H + y) * W) * C + c;
  for (int g = 0; g < sg.n; ++g) {
    const int x0 = sg.x0[g], x1 = sg.x1[g];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a;
    int xx = x0;
    constexpr int U = 10;      // ten independent 16-byte loads in flight per lane
    for (; xx + U <= x1; xx += U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(row + (size_t)(xx + u) * C);
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
        a2.x += v[u + 1].x; a2.y += v[u + 1].y; a2.z += v[u + 1].z; a2.w += v[u + 1].w;
      }
    }
    {   // the segment's last < U pixels: requested together (a clamped address for the slots past the end), added one by one in
        // order as before -- a loop of single loads had every load wait for the one before it (round 6: one 720 x 720 map's row pass
        // 42 -> 15 us; the sums are the same bits)
      float4 v[U - 1];
      const int rem = x1 - xx;
      if (rem > 0) {
#pragma unroll
      for (int u = 0; u < U - 1; ++u) v[u] = *reinterpret_cast<const float4*>(row + (size_t)(u < rem ? xx + u : x1 - 1) * C);
#pragma unroll
      for (int u = 0; u < U - 1; ++u)
        if (u < rem) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
      }
    }
    a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
    const unsigned m = sg.mask[g];
#pragma unroll
    for (int s = 0; s < PPM_MAX_SLOTS; ++s)
      if ((m >> s) & 1u) { acc[s].x += a.x; acc[s].y += a.y; acc[s].z += a.z; acc[s].w += a.w; }
  }
  }
  float* out = rowsum + (((size_t)b * gr.n + grp) * sl.n) * C + c;
#pragma unroll
  for (int s = 0; s < PPM_MAX_SLOTS; ++s)
    if (s < sl.n) *reinterpret_cast<float4*>(out + (size_t)s * C) = acc[s];
}

__global__ __launch_bounds__(256) void ppm_binsum_kernel(const float* __restrict__ rowsum, float* __restrict__ out, int H,
                                                         int W, int C, PpmScales sc, int nslots, PpmGroups gr) {
  const int bin = blockIdx.x, b = blockIdx.y;
  const int c = (blockIdx.z * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  int k = 0, local = bin, before = 0, slot0 = 0, si = 0;
  for (int i = 0; i < sc.n; ++i) {
    const int kk = sc.s[i];
    if (local < kk * kk) { k = kk; si = i; break; }
    local -= kk * kk;
    before += kk * kk;
    slot0 += kk;
  }
  const int by = local / k, bx = local - by * k;
  const int y0 = (by * H) / k, y1 = ((by + 1) * H + k - 1) / k;
  const int x0 = (bx * W) / k, x1 = ((bx + 1) * W + k - 1) / k;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // the groups inside this bin's rows: a contiguous run (groups are sorted and never straddle a bin edge); the run is found
  // first so that the loads below form a plain loop the compiler can keep several of in flight
  int g0 = 0, g1 = 0;
  for (int g = 0; g < gr.n; ++g) {
    if (gr.y1[g] <= y0) g0 = g + 1;
    if (gr.y0[g] < y1) g1 = g + 1;
  }
  // sixteen groups' loads in flight, added in group order (round 6: the bin of scale 1 adds up to one group per row of a batch-1 map)
  for (int gb = g0; gb < g1; gb += 16) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int g = gb + u < g1 ? gb + u : g1 - 1;
      v[u] = *reinterpret_cast<const float4*>(rowsum + (((size_t)b * gr.n + g) * nslots + slot0 + bx) * C + c);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (gb + u < g1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  const float cnt = (float)((y1 - y0) * (x1 - x0));
  acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt;
  const size_t row = ppm_row0(sc, si, before, gridDim.y) + (size_t)b * k * k + local;
  *reinterpret_cast<float4*>(out + row * C + c) = acc;
}

size_t ppm_pool_scratch_floats(int B, int H, int C, const int* scales, int nscales) {
  int slots = 0;
  for (int i = 0; i < nscales; ++i) slots += scales[i];
  if (slots > PPM_MAX_SLOTS) return 0;      // fallback kernel needs no scratch
  return (size_t)B * H * slots * C;
}

int launch_ppm_pool2(const float* x, float* scratch, float* out, int B, int H, int W, int C, const int* scales,
                     int nscales, hipStream_t s, int scale_rows) {
  if (nscales > 8 || C % 4) return fail(-2, "ppm_pool: unsupported configuration");
  PpmScales sc;
  PpmSlots sl;
  sc.n = nscales;
  sc.rows = scale_rows;
  sl.n = 0;
  int nbins = 0;
  for (int i = 0; i < nscales; ++i) {
    const int k = scales[i];
    sc.s[i] = k;
    nbins += k * k;
    for (int bx = 0; bx < k; ++bx) {
      if (sl.n >= PPM_MAX_SLOTS) return launch_ppm_pool(x, out, B, H, W, C, scales, nscales, s, scale_rows);
      sl.x0[sl.n] = (short)((bx * W) / k);
      sl.x1[sl.n] = (short)(((bx + 1) * W + k - 1) / k);
      ++sl.n;
    }
  }
  if (!scratch) return launch_ppm_pool(x, out, B, H, W, C, scales, nscales, s, scale_rows);
  // segments between consecutive bin edges, each with the set of bin-columns (slots) that contain it
  PpmSegs sg;
  sg.n = 0;
  {
    short edges[4 * PPM_MAX_SLOTS];
    int ne = 0;
    for (int i = 0; i < sl.n; ++i) { edges[ne++] = sl.x0[i]; edges[ne++] = sl.x1[i]; }
    std::sort(edges, edges + ne);
    ne = (int)(std::unique(edges, edges + ne) - edges);
    for (int i = 0; i + 1 < ne; ++i) {
      unsigned m = 0;
      for (int q = 0; q < sl.n; ++q)
        if (edges[i] >= sl.x0[q] && edges[i + 1] <= sl.x1[q]) m |= 1u << q;
      if (!m) continue;
      if (sg.n >= 2 * PPM_MAX_SLOTS) return launch_ppm_pool(x, out, B, H, W, C, scales, nscales, s, scale_rows);
      sg.x0[sg.n] = edges[i]; sg.x1[sg.n] = edges[i + 1]; sg.mask[sg.n] = m;
      ++sg.n;
    }
  }
  // row groups: cut [0, H) at every bin-row edge of every scale, then into runs of at most PPM_GROUP_ROWS rows
  PpmGroups gr;
  gr.n = 0;
  const int slabs = (C / 4 + 255) / 256;
  // rows per group: as many as still leave ~768 workgroups for the row pass (small batches keep one row per workgroup: their
  // pass is latency-bound, not traffic-bound)
  const int forced_rows = (int)opt(OPT_PPM_GROUP_ROWS);
  const int group_rows = forced_rows > 0 ? std::min(forced_rows, PPM_GROUP_ROWS)
                                         : std::max(1, std::min(PPM_GROUP_ROWS, (int)((long long)H * B * slabs / 768)));
  {
    std::vector<int> edges;
    for (int i = 0; i < nscales; ++i)
      for (int by = 0; by < scales[i]; ++by) {
        edges.push_back((by * H) / scales[i]);
        edges.push_back(((by + 1) * H + scales[i] - 1) / scales[i]);
      }
    edges.push_back(0);
    edges.push_back(H);
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    bool fits = true;
    for (size_t i = 0; i + 1 < edges.size() && fits; ++i)
      for (int y = edges[i]; y < edges[i + 1]; y += group_rows) {
        if (gr.n >= PPM_MAX_GROUPS) { fits = false; break; }
        gr.y0[gr.n] = (short)y;
        gr.y1[gr.n] = (short)std::min(y + group_rows, edges[i + 1]);
        ++gr.n;
      }
    if (!fits) {                      // very tall maps: one row per group (the scratch is sized for H rows)
      if (H > PPM_MAX_GROUPS) return launch_ppm_pool(x, out, B, H, W, C, scales, nscales, s, scale_rows);
      gr.n = H;
      for (int y = 0; y < H; ++y) { gr.y0[y] = (short)y; gr.y1[y] = (short)(y + 1); }
    }
  }
  hipLaunchKernelGGL(ppm_rowsum_kernel, dim3(gr.n, B, slabs), dim3(256), 0, s, x, scratch, H, W, C, sl, sg, gr);
  hipLaunchKernelGGL(ppm_binsum_kernel, dim3(nbins, B, slabs), dim3(256), 0, s, scratch, out, H, W, C, sc, sl.n, gr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("ppm_pool2: ") + hipGetErrorString(e));
}

// ---- bilinear source index/weight, ATen upsample_bilinear2d semantics ----
__device__ __forceinline__ PEANUT_NO_PK_F32 void bilinear_src(int dst, int in_size, int out_size, int align_corners, int* i0,
                                             int* i1, float* l1) {
  float src;
  if (align_corners) {
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    src = scale * (float)dst;
  } else {
    const float scale = (float)in_size / (float)out_size;
    src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  int a = (int)src;
  if (a > in_size - 1) a = in_size - 1;
  *i0 = a;
  *i1 = a + (a < in_size - 1 ? 1 : 0);
  *l1 = src - (float)a;
}

// table (scale-major, see ppm_pool_kernel; post conv/BN/ReLU) -> out [B,H,W,nscales*Cp]: channel block s holds scale s
// upsampled to HxW.  One thread per (pixel, scale, 4 channels).
__global__ __launch_bounds__(256) void ppm_upsample_concat_kernel(const float* __restrict__ table,
                                                                  float* __restrict__ out, int H, int W, int Cp,
                                                                  PpmScales sc, int B, int align_corners,
                                                                  long long total) {
  const int groups = Cp >> 2;
  const int per_pix = sc.n * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i % per_pix);
    long long pix = i / per_pix;
    const int s = r / groups, g = r - s * groups;
    const int xx = (int)(pix % W);
    const long long t = pix / W;
    const int yy = (int)(t % H);
    const long long b = t / H;
    int base = 0;
    for (int j = 0; j < s; ++j) base += sc.s[j] * sc.s[j];
    const int k = sc.s[s];
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(yy, k, H, align_corners, &y0, &y1, &ly);
    bilinear_src(xx, k, W, align_corners, &x0, &x1, &lx);
    const float* tb = table + (ppm_row0(sc, s, base, B) + (size_t)b * k * k) * Cp + g * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(tb + (size_t)(y0 * k + x0) * Cp);
    const float4 v01 = *reinterpret_cast<const float4*>(tb + (size_t)(y0 * k + x1) * Cp);
    const float4 v10 = *reinterpret_cast<const float4*>(tb + (size_t)(y1 * k + x0) * Cp);
    const float4 v11 = *reinterpret_cast<const float4*>(tb + (size_t)(y1 * k + x1) * Cp);
    const float hy = 1.f - ly, hx = 1.f - lx;
    float4 o;
    // ATen order: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *reinterpret_cast<float4*>(out + (size_t)pix * (sc.n * Cp) + s * Cp + g * 4) = o;
  }
}

int launch_ppm_upsample_concat(const float* table, float* out, int B, int H, int W, int Cp, const int* scales,
                               int nscales, int align_corners, hipStream_t s) {
  if (nscales > 8 || Cp % 4) return fail(-2, "ppm_upsample: unsupported configuration");
  PpmScales sc;
  sc.n = nscales;
  sc.rows = 0;
  for (int i = 0; i < nscales; ++i) sc.s[i] = scales[i];
  const long long total = (long long)B * H * W * nscales * (Cp / 4);
  hipLaunchKernelGGL(ppm_upsample_concat_kernel, dim3(grid_for(total)), dim3(256), 0, s, table, out, H, W, Cp, sc,
                     B, align_corners, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("ppm_upsample: ") + hipGetErrorString(e));
}

int launch_ppm_conv_term_l2(const float* Q, float* R, int B, int H, int W, int C, const int* scales, int nscales,
                            int align_corners, hipStream_t s, int scale_rows);

// ---- pyramid half of the PSP bottleneck conv, folded through linearity ----
// The reference convolves cat([x, up(p_1), up(p_2), up(p_3), up(p_6)]) with a 3x3 kernel
// (psp_head.py:107-110).  The pyramid half of that sum is low-rank: up(p_s) is a bilinear
// interpolation of only k_s^2 vectors, and the conv is linear, so
//     sum_tap sum_c W[n][c][tap] * up(p_s)[c](pix+tap)  =  sum_tap bilinear_s(Q_s[.][tap][n])(pix+tap)
// with Q_s[g][tap][n] = sum_c W[n][c][tap] * p_s[g][c]  (a tiny GEMM over the 50 pooled vectors, done by
// the conv kernel).  This kernel evaluates the right-hand side for every output pixel: 9 taps (zero
// outside the image, like the conv's zero padding) x nscales x 4 bilinear neighbours.  Q already
// carries the BatchNorm scale, so the result is added as the bottleneck conv's "residual".
// Q rows are scale-major like the pooled table: row = B*base_s + b*k_s^2 + g, each row [9][C].
__global__ __launch_bounds__(256) PEANUT_NO_PK_F32 void ppm_conv_term_kernel(const float* __restrict__ Q, float* __restrict__ R, int H,
                                                            int W, int C, PpmScales sc, int B, int align_corners,
                                                            long long total) {
  const int groups = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    long long pix = i / groups;
    const int xx = (int)(pix % W);
    const long long t = pix / W;
    const int yy = (int)(t % H);
    const int b = (int)(t / H);
    float4 acc;      // (not make_float4: HIP's helper would stay a call from a kernel compiled without packed fp32, common.h)
    acc.x = 0.f; acc.y = 0.f; acc.z = 0.f; acc.w = 0.f;
    int base = 0;
    for (int s = 0; s < sc.n; ++s) {
      const int k = sc.s[s];
      const float* qs = Q + (ppm_row0(sc, s, base, B) + (size_t)b * k * k) * (9 * C) + g * 4;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int py = yy + dy - 1;
        if ((unsigned)py >= (unsigned)H) continue;
        int y0, y1;
        float ly;
        bilinear_src(py, k, H, align_corners, &y0, &y1, &ly);
        const float hy = 1.f - ly;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int px = xx + dx - 1;
          if ((unsigned)px >= (unsigned)W) continue;
          int x0, x1;
          float lx;
          bilinear_src(px, k, W, align_corners, &x0, &x1, &lx);
          const float hx = 1.f - lx;
          const float* qt = qs + (dy * 3 + dx) * C;
          const float4 v00 = *reinterpret_cast<const float4*>(qt + (size_t)(y0 * k + x0) * (9 * C));
          const float4 v01 = *reinterpret_cast<const float4*>(qt + (size_t)(y0 * k + x1) * (9 * C));
          const float4 v10 = *reinterpret_cast<const float4*>(qt + (size_t)(y1 * k + x0) * (9 * C));
          const float4 v11 = *reinterpret_cast<const float4*>(qt + (size_t)(y1 * k + x1) * (9 * C));
          acc.x += hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
          acc.y += hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
          acc.z += hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
          acc.w += hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        }
      }
      base += k * k;
    }
    *reinterpret_cast<float4*>(R + (size_t)pix * C + g * 4) = acc;
  }
}

// LDS-staged form of the same evaluation: one workgroup per (image, 32-channel chunk) keeps that
// image's folded tables Q[all bins][9 taps][32 ch] (57.6 KB for the (1,2,3,6) pyramid) and the
// per-coordinate bilinear taps in LDS and sweeps the image row by row with a separable evaluation
// (24 LDS reads per output instead of 144 L2 round trips).
struct BlTap { short i0, i1; float l; };

__global__ __launch_bounds__(256) void ppm_conv_term_lds_kernel(const float* __restrict__ Q, float* __restrict__ R,
                                                                int H, int W, int C, PpmScales sc, int B, int nbins,
                                                                int align_corners) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int CH = 32;                                   // channels per workgroup
  float* q = reinterpret_cast<float*>(lds_raw);            // [nbins][9][CH]
  BlTap* taps = reinterpret_cast<BlTap*>(lds_raw + (size_t)nbins * 9 * CH * sizeof(float));   // [n][2][L+2]
  const int L = (H > W ? H : W) + 2;
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  // stage Q (rows are scale-major: row(s, b, g) = B*base_s + b*k*k + g)
  {
    int base = 0;
    for (int s = 0; s < sc.n; ++s) {
      const int k = sc.s[s], cells = k * k;
      const float* src = Q + (ppm_row0(sc, s, base, B) + (size_t)b * cells) * (9 * C) + c0;
      for (int i = threadIdx.x; i < cells * 9 * (CH / 4); i += blockDim.x) {
        const int v4 = i % (CH / 4), rt = i / (CH / 4);        // rt = cell*9 + tap
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)rt * C + v4 * 4);
        *reinterpret_cast<float4*>(q + ((size_t)(base * 9 + rt)) * CH + v4 * 4) = v;
      }
      base += cells;
    }
  }
  for (int i = threadIdx.x; i < sc.n * 2 * L; i += blockDim.x) {
    const int s = i / (2 * L), r = i - s * 2 * L, dim = r / L, t = r - dim * L;   // coordinate = t - 1
    const int n = dim == 0 ? H : W, coord = t - 1;
    BlTap tp;
    tp.i0 = -1; tp.i1 = -1; tp.l = 0.f;
    if ((unsigned)coord < (unsigned)n) {
      int i0, i1;
      float l;
      bilinear_src(coord, sc.s[s], n, align_corners, &i0, &i1, &l);
      tp.i0 = (short)i0; tp.i1 = (short)i1; tp.l = l;
    }
    taps[i] = tp;
  }
  __syncthreads();
  // Separable evaluation, one output row at a time.  The x-interpolation weights do not depend on the
  // vertical tap, so the three vertical taps are folded first:
  //     S[s][gx][dx][n] = sum_dy inb(y+dy-1) * ( hy*Q_s[y0][gx][dy*3+dx][n] + ly*Q_s[y1][gx][dy*3+dx][n] )
  //     R[y][x][n]      = sum_s sum_dx inb(x+dx-1) * ( hx*S[s][x0][dx][n] + lx*S[s][x1][dx][n] )
  // -> 24 LDS reads per output instead of 144.
  float* srow = reinterpret_cast<float*>(taps + (size_t)sc.n * 2 * L);       // [slots][3][CH], slots = sum_s k_s
  int slots = 0;
  for (int s = 0; s < sc.n; ++s) slots += sc.s[s];
  // rows are split over gridDim.z workgroups (small batches: (C/32) x B workgroups alone would leave the chip idle)
  const int rows_per = (H + gridDim.z - 1) / gridDim.z;
  const int y_begin = blockIdx.z * rows_per, y_end = min(H, y_begin + rows_per);
  for (int yy = y_begin; yy < y_end; ++yy) {
    for (int i = threadIdx.x; i < slots * 3 * (CH / 4); i += blockDim.x) {
      const int g = i % (CH / 4), r = i / (CH / 4), dx = r % 3, slot = r / 3;
      int s = 0, gx = slot, base = 0;
      while (gx >= sc.s[s]) { gx -= sc.s[s]; base += sc.s[s] * sc.s[s]; ++s; }
      const int k = sc.s[s];
      const BlTap* ty = taps + (s * 2 + 0) * L + yy;
      const float* qs = q + (size_t)base * 9 * CH + g * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const BlTap t = ty[dy];
        if (t.i0 < 0) continue;
        const float hy = 1.f - t.l;
        const float4 v0 = *reinterpret_cast<const float4*>(qs + ((t.i0 * k + gx) * 9 + dy * 3 + dx) * CH);
        const float4 v1 = *reinterpret_cast<const float4*>(qs + ((t.i1 * k + gx) * 9 + dy * 3 + dx) * CH);
        acc.x += hy * v0.x + t.l * v1.x; acc.y += hy * v0.y + t.l * v1.y;
        acc.z += hy * v0.z + t.l * v1.z; acc.w += hy * v0.w + t.l * v1.w;
      }
      *reinterpret_cast<float4*>(srow + (size_t)r * CH + g * 4) = acc;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < W * (CH / 4); it += blockDim.x) {
      const int g = it % (CH / 4), xx = it / (CH / 4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int slot0 = 0;
      for (int s = 0; s < sc.n; ++s) {
        const BlTap* tx = taps + (s * 2 + 1) * L + xx;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const BlTap t = tx[dx];
          if (t.i0 < 0) continue;
          const float hx = 1.f - t.l;
          const float4 v0 = *reinterpret_cast<const float4*>(srow + ((size_t)(slot0 + t.i0) * 3 + dx) * CH + g * 4);
          const float4 v1 = *reinterpret_cast<const float4*>(srow + ((size_t)(slot0 + t.i1) * 3 + dx) * CH + g * 4);
          acc.x += hx * v0.x + t.l * v1.x; acc.y += hx * v0.y + t.l * v1.y;
          acc.z += hx * v0.z + t.l * v1.z; acc.w += hx * v0.w + t.l * v1.w;
        }
        slot0 += sc.s[s];
      }
      *reinterpret_cast<float4*>(R + (((size_t)b * H + yy) * W + xx) * C + c0 + g * 4) = acc;
    }
    __syncthreads();
  }
}

// The same evaluation with ONE WAVE PER OUTPUT ROW and no workgroup barrier inside the row loop (round 4).  The kernel above
// runs its two phases workgroup-wide: two __syncthreads per row (120 per workgroup), 288 and 480 work items on 256 threads
// (second passes 12 % / 87 % full) -- 0.28 ms for the headline's 236 MB of output, 0.12 of the HBM rate.  Here each of the 16
// waves owns whole rows: it folds the vertical taps of its row into a wave-private S buffer (288 items over 64 lanes) and
// evaluates the row from it (480 items), lanes in lock step, LDS operations of a wave executing in issue order -- the waves
// drift freely, LDS reads of one overlap the arithmetic of another.  Same arithmetic, operation by operation: bit-identical.
template <int NW>
__global__ __launch_bounds__(64 * NW) void ppm_conv_term_rows_kernel(const float* __restrict__ Q, float* __restrict__ R,
                                                                     int H, int W, int C, PpmScales sc, int B, int nbins,
                                                                     int align_corners) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int CH = 32;
  float* q = reinterpret_cast<float*>(lds_raw);            // [nbins][9][CH]
  BlTap* taps = reinterpret_cast<BlTap*>(lds_raw + (size_t)nbins * 9 * CH * sizeof(float));   // [n][2][L+2]
  const int L = (H > W ? H : W) + 2;
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  {
    int base = 0;
    for (int s = 0; s < sc.n; ++s) {
      const int k = sc.s[s], cells = k * k;
      const float* src = Q + (ppm_row0(sc, s, base, B) + (size_t)b * cells) * (9 * C) + c0;
      for (int i = threadIdx.x; i < cells * 9 * (CH / 4); i += blockDim.x) {
        const int v4 = i % (CH / 4), rt = i / (CH / 4);
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)rt * C + v4 * 4);
        *reinterpret_cast<float4*>(q + ((size_t)(base * 9 + rt)) * CH + v4 * 4) = v;
      }
      base += cells;
    }
  }
  for (int i = threadIdx.x; i < sc.n * 2 * L; i += blockDim.x) {
    const int s = i / (2 * L), r = i - s * 2 * L, dim = r / L, t = r - dim * L;
    const int n = dim == 0 ? H : W, coord = t - 1;
    BlTap tp;
    tp.i0 = -1; tp.i1 = -1; tp.l = 0.f;
    if ((unsigned)coord < (unsigned)n) {
      int i0, i1;
      float l;
      bilinear_src(coord, sc.s[s], n, align_corners, &i0, &i1, &l);
      tp.i0 = (short)i0; tp.i1 = (short)i1; tp.l = l;
    }
    taps[i] = tp;
  }
  __syncthreads();
  int slots = 0;
  for (int s = 0; s < sc.n; ++s) slots += sc.s[s];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* srow = reinterpret_cast<float*>(taps + (size_t)sc.n * 2 * L) + (size_t)wave * slots * 3 * CH;    // this wave's [slots][3][CH]
  const int rows_per = (H + gridDim.z - 1) / gridDim.z;
  const int y_begin = blockIdx.z * rows_per, y_end = min(H, y_begin + rows_per);
  for (int yy = y_begin + wave; yy < y_end; yy += NW) {
    for (int i = lane; i < slots * 3 * (CH / 4); i += 64) {
      const int g = i % (CH / 4), r = i / (CH / 4), dx = r % 3, slot = r / 3;
      int s = 0, gx = slot, base = 0;
      while (gx >= sc.s[s]) { gx -= sc.s[s]; base += sc.s[s] * sc.s[s]; ++s; }
      const int k = sc.s[s];
      const BlTap* ty = taps + (s * 2 + 0) * L + yy;
      const float* qs = q + (size_t)base * 9 * CH + g * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const BlTap t = ty[dy];
        if (t.i0 < 0) continue;
        const float hy = 1.f - t.l;
        const float4 v0 = *reinterpret_cast<const float4*>(qs + ((t.i0 * k + gx) * 9 + dy * 3 + dx) * CH);
        const float4 v1 = *reinterpret_cast<const float4*>(qs + ((t.i1 * k + gx) * 9 + dy * 3 + dx) * CH);
        acc.x += hy * v0.x + t.l * v1.x; acc.y += hy * v0.y + t.l * v1.y;
        acc.z += hy * v0.z + t.l * v1.z; acc.w += hy * v0.w + t.l * v1.w;
      }
      *reinterpret_cast<float4*>(srow + (size_t)r * CH + g * 4) = acc;
    }
    // the wave's own S row: written above, read below by other lanes of the SAME wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int it = lane; it < W * (CH / 4); it += 64) {
      const int g = it % (CH / 4), xx = it / (CH / 4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int slot0 = 0;
      for (int s = 0; s < sc.n; ++s) {
        const BlTap* tx = taps + (s * 2 + 1) * L + xx;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const BlTap t = tx[dx];
          if (t.i0 < 0) continue;
          const float hx = 1.f - t.l;
          const float4 v0 = *reinterpret_cast<const float4*>(srow + ((size_t)(slot0 + t.i0) * 3 + dx) * CH + g * 4);
          const float4 v1 = *reinterpret_cast<const float4*>(srow + ((size_t)(slot0 + t.i1) * 3 + dx) * CH + g * 4);
          acc.x += hx * v0.x + t.l * v1.x; acc.y += hx * v0.y + t.l * v1.y;
          acc.z += hx * v0.z + t.l * v1.z; acc.w += hx * v0.w + t.l * v1.w;
        }
        slot0 += sc.s[s];
      }
      *reinterpret_cast<float4*>(R + (((size_t)b * H + yy) * W + xx) * C + c0 + g * 4) = acc;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the row's S reads, before the next row overwrites the buffer
    __builtin_amdgcn_wave_barrier();
  }
}

int launch_ppm_conv_term(const float* Q, float* R, int B, int H, int W, int C, const int* scales, int nscales,
                         int align_corners, hipStream_t s, int scale_rows) {
  if (nscales <= 8 && C % 32 == 0) {
    PpmScales sc;
    sc.n = nscales;
    sc.rows = scale_rows;
    int nbins = 0;
    for (int i = 0; i < nscales; ++i) { sc.s[i] = scales[i]; nbins += scales[i] * scales[i]; }
    const int L = (H > W ? H : W) + 2;
    int slots = 0;
    for (int i = 0; i < nscales; ++i) slots += scales[i];
    const size_t lds = (size_t)nbins * 9 * 32 * sizeof(float) + (size_t)nscales * 2 * L * sizeof(BlTap) +
                       (size_t)slots * 3 * 32 * sizeof(float);
    // one wave per row (16 waves, each with its own S buffer), when that fits LDS and the option allows
    constexpr int ROWS_NW = 16;
    const size_t lds_rows = lds + (size_t)(ROWS_NW - 1) * slots * 3 * 32 * sizeof(float);
    // (small batches keep the kernel below: with fewer than ~256 (image, channel chunk) pairs the rows of an image have to be split
    // over several workgroups, and every one of these 1024-thread workgroups stages the image's whole 58 KB table first)
    if (lds_rows <= 150 * 1024 && opt(OPT_PPM_TERM_ROWS) != 0 && (long long)(C / 32) * B >= 128) {
      static bool raised_rows = false;
      if (!raised_rows) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppm_conv_term_rows_kernel<ROWS_NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return fail(-3, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        raised_rows = true;
      }
      int row_splits = (256 + (C / 32) * B - 1) / ((C / 32) * B);      // one workgroup per CU: aim at >= 256 workgroups
      row_splits = std::max(1, std::min(row_splits, H / ROWS_NW > 0 ? H / ROWS_NW : 1));
      hipLaunchKernelGGL(ppm_conv_term_rows_kernel<ROWS_NW>, dim3(C / 32, B, row_splits), dim3(64 * ROWS_NW), lds_rows, s, Q, R, H, W, C,
                         sc, B, nbins, align_corners);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : fail(-3, std::string("ppm_conv_term_rows: ") + hipGetErrorString(e));
    }
    if (lds <= 150 * 1024) {    // the (1,2,3,6) pyramid needs 66 KB at 60x60: above the 64 KB default limit
      static bool raised = false;
      if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppm_conv_term_lds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return fail(-3, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        raised = true;
      }
      int row_splits = (512 + (C / 32) * B - 1) / ((C / 32) * B);      // aim at >= 512 workgroups
      row_splits = std::max(1, std::min(row_splits, H / 4 > 0 ? H / 4 : 1));
      hipLaunchKernelGGL(ppm_conv_term_lds_kernel, dim3(C / 32, B, row_splits), dim3(256), lds, s, Q, R, H, W, C, sc, B, nbins,
                         align_corners);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : fail(-3, std::string("ppm_conv_term_lds: ") + hipGetErrorString(e));
    }
  }
  return launch_ppm_conv_term_l2(Q, R, B, H, W, C, scales, nscales, align_corners, s, scale_rows);
}

int launch_ppm_conv_term_l2(const float* Q, float* R, int B, int H, int W, int C, const int* scales, int nscales,
                            int align_corners, hipStream_t s, int scale_rows) {
  if (nscales > 8 || C % 4) return fail(-2, "ppm_conv_term: unsupported configuration");
  PpmScales sc;
  sc.n = nscales;
  sc.rows = scale_rows;
  for (int i = 0; i < nscales; ++i) sc.s[i] = scales[i];
  const long long total = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(ppm_conv_term_kernel, dim3(grid_for(total)), dim3(256), 0, s, Q, R, H, W, C, sc, B,
                     align_corners, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("ppm_conv_term: ") + hipGetErrorString(e));
}

// ---- final resize: NHWC logits [B,h,w,K] -> NCHW [B,K,H,W] (+ optional sigmoid) ----
// One thread per output pixel, all K classes: consecutive lanes = consecutive x -> every class
// plane is written in coalesced 256-byte wave segments; the low-res source stays in L1/L2.
template <int KMAX>
__global__ __launch_bounds__(256) PEANUT_NO_PK_F32 void upsample_logits_kernel(const float* __restrict__ lo, float* __restrict__ out,
                                                              int h, int w, int K, int H, int W, int align_corners,
                                                              int sigmoid, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const long long t = i / W;
    const int yy = (int)(t % H);
    const long long b = t / H;
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(yy, h, H, align_corners, &y0, &y1, &ly);
    bilinear_src(xx, w, W, align_corners, &x0, &x1, &lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = lo + (((size_t)b * h + y0) * w + x0) * K;
    const float* p01 = lo + (((size_t)b * h + y0) * w + x1) * K;
    const float* p10 = lo + (((size_t)b * h + y1) * w + x0) * K;
    const float* p11 = lo + (((size_t)b * h + y1) * w + x1) * K;
    float* o = out + ((size_t)b * K) * H * W + (size_t)yy * W + xx;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        float v = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
        if (sigmoid) v = 1.f / (1.f + expf(-v));
        o[(size_t)k * H * W] = v;
      }
    }
  }
}

int launch_upsample_logits(const float* lo, float* out, int B, int h, int w, int K, int H, int W,
                           int align_corners, int sigmoid, hipStream_t s) {
  const long long total = (long long)B * H * W;
  if (K <= 8)
    hipLaunchKernelGGL(upsample_logits_kernel<8>, dim3(grid_for(total)), dim3(256), 0, s, lo, out, h, w, K, H, W,
                       align_corners, sigmoid, total);
  else if (K <= 32)
    hipLaunchKernelGGL(upsample_logits_kernel<32>, dim3(grid_for(total)), dim3(256), 0, s, lo, out, h, w, K, H, W,
                       align_corners, sigmoid, total);
  else
    return fail(-2, "upsample_logits: more than 32 classes is not supported");
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("upsample_logits: ") + hipGetErrorString(e));
}

}  // namespace peanut
