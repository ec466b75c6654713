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
#include "common.h"

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
  // one thread per (pixel, 4-channel group): reads are coalesced per channel plane across the
  // 64/ (Cpad/4) pixels of a wave, writes are contiguous 16-byte pieces of the NHWC rows.
  const int groups = Cpad >> 2;
  const long long total = npix * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i / groups;
    const int g = (int)(i - pix * groups);
    const long long b = pix / HW, hw = pix - b * HW;
    const float* src = x + (b * C) * HW + hw;
    float4 v;
    const int c0 = g * 4;
    v.x = c0 + 0 < C ? src[(long long)(c0 + 0) * HW] : 0.f;
    v.y = c0 + 1 < C ? src[(long long)(c0 + 1) * HW] : 0.f;
    v.z = c0 + 2 < C ? src[(long long)(c0 + 2) * HW] : 0.f;
    v.w = c0 + 3 < C ? src[(long long)(c0 + 3) * HW] : 0.f;
    *reinterpret_cast<float4*>(y + pix * Cpad + c0) = v;
  }
}

int launch_nchw_to_nhwc_pad(const float* x, float* y, int B, int C, int H, int W, int Cpad, hipStream_t s) {
  const long long HW = (long long)H * W, npix = HW * B;
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for(npix * (Cpad / 4))), dim3(256), 0, s, x, y, C, HW, npix,
                     Cpad);
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
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
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
struct PpmScales { int s[8]; int n; };

__global__ __launch_bounds__(256) void ppm_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                                       int W, int C, PpmScales sc, int nbins) {
  const int bin = blockIdx.x, b = blockIdx.y;
  const int c = (blockIdx.z * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  int k = 0, local = bin, before = 0;
  for (int i = 0; i < sc.n; ++i) {
    const int kk = sc.s[i];
    if (local < kk * kk) { k = kk; break; }
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
  const size_t row = (size_t)gridDim.y * before + (size_t)b * k * k + local;
  *reinterpret_cast<float4*>(out + row * C + c) = acc;
}

int launch_ppm_pool(const float* x, float* out, int B, int H, int W, int C, const int* scales, int nscales,
                    hipStream_t s) {
  if (nscales > 8 || C % 4) return fail(-2, "ppm_pool: unsupported configuration");
  PpmScales sc;
  sc.n = nscales;
  int nbins = 0;
  for (int i = 0; i < nscales; ++i) { sc.s[i] = scales[i]; nbins += scales[i] * scales[i]; }
  const dim3 grid(nbins, B, (C / 4 + 255) / 256);
  hipLaunchKernelGGL(ppm_pool_kernel, grid, dim3(256), 0, s, x, out, H, W, C, sc, nbins);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("ppm_pool: ") + hipGetErrorString(e));
}

// ---- bilinear source index/weight, ATen upsample_bilinear2d semantics ----
__device__ __forceinline__ void bilinear_src(int dst, int in_size, int out_size, int align_corners, int* i0,
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
    const float* tb = table + ((size_t)B * base + (size_t)b * k * k) * Cp + g * 4;
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
  for (int i = 0; i < nscales; ++i) sc.s[i] = scales[i];
  const long long total = (long long)B * H * W * nscales * (Cp / 4);
  hipLaunchKernelGGL(ppm_upsample_concat_kernel, dim3(grid_for(total)), dim3(256), 0, s, table, out, H, W, Cp, sc,
                     B, align_corners, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("ppm_upsample: ") + hipGetErrorString(e));
}

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
__global__ __launch_bounds__(256) void ppm_conv_term_kernel(const float* __restrict__ Q, float* __restrict__ R, int H,
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
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int base = 0;
    for (int s = 0; s < sc.n; ++s) {
      const int k = sc.s[s];
      const float* qs = Q + ((size_t)B * base + (size_t)b * k * k) * (9 * C) + g * 4;
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

int launch_ppm_conv_term(const float* Q, float* R, int B, int H, int W, int C, const int* scales, int nscales,
                         int align_corners, hipStream_t s) {
  if (nscales > 8 || C % 4) return fail(-2, "ppm_conv_term: unsupported configuration");
  PpmScales sc;
  sc.n = nscales;
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
__global__ __launch_bounds__(256) void upsample_logits_kernel(const float* __restrict__ lo, float* __restrict__ out,
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
