// Shared declarations for the peanut_hip library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace peanut {

// ---- error plumbing (thread-local message, negative codes; include/peanut_hip.h) ----
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
// the kernel family the calling thread's last conv launch picked (string with static storage; peanut_last_conv_kernel)
void note_kernel(const char* family);
const char* noted_kernel();

#define PEANUT_HIP_CHECK(expr)                                                            \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return ::peanut::fail(-3, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

// ---- fused conv (implicit GEMM on fp32 MFMA) ----
// Activations are NHWC fp32 with the channel count padded to a multiple of 16.
// Weights are pre-packed per (n-tile, k-tile) into contiguous [BN][BK] blocks, k-tiles ordered
// channel-chunk outer / filter-tap inner (see pack_conv_weights).
struct ConvDesc {
  int cin;        // padded input channels of the (possibly concatenated) input
  int cout;       // real output channels
  int kh, kw, stride, pad, dil;
  int relu;
  int bn_tile;    // BN used for packing (32/64/128)
  int bk;         // BK used for packing (16/32)
  int cout_pad;   // cout rounded up to bn_tile
  const float* w_packed;   // device: fp32 weights packed for the fp32 MFMA kernels
  const float* scale;      // device [cout_pad]
  const float* shift;      // device [cout_pad]
  // emulated-fp32 modes (bf16x6 / fp16x3 / bf16x3): fp32 activations split into 16-bit pieces in registers, the weights'
  // pieces pre-split in w_s -- pointwise layers on gemm_rs.hip (rs = 1), every other conv on conv_rs.hip (rs = 2)
  int rs;                  // 1: pointwise layer on gemm_rs.hip; 2: any other conv on conv_rs.hip (k-tile = 16 channels of one tap)
  const void* w_s;         // device: pre-split weights [n-tile][k-tile of 16][plane][bn_tile][16 bf16] (pack_weights_sx), or null
  int s_planes;            // emulation kind (rs_common.h): 2 = bf16x3 (two bf16 pieces, three products), 3 = bf16x6 (three
                           // pieces, six products), 4 = fp16x3 (two fp16 pieces, three products)
  float s_alpha;           // 1 / the power of two the weights were scaled by before the split (fp16 pieces), else 1
  int flush_ch;            // fp32 pointwise kernels (conv_pw.hip): channels per partial sum of the two-level fp32 accumulation (conv_common.h:
                           // PEANUT_FLUSH_*), 0 = one running sum over all of K
};

// A pointwise conv whose tail split-K partial tiles are summed by its CONSUMER (round 6): when every 128 x 128 tile of a launch is cut
// into k-ranges (a batch-1 layer: fewer tiles than CUs), the launcher may leave the partial tiles unsummed -- no reduce launch -- and
// describe them here; the small-problem Winograd input transform of the next layer then reads a pixel as
// relu(scale * alpha * (p0 + p1 + ...) + shift), the reduce kernel's own expression in its own order (bit-identical), and the conv's
// output tensor is never written.  Saves one launch (~5.2 us: the floor of a small dependent kernel) per Bottleneck at batch 1.
// The caller offers the struct (ConvArgs::defer) only when the consumer has said it can read it (wino_input_accepts_deferred).
struct DeferredSplit {
  bool valid;
  const float* partial;      // [tile = mt * ntiles + nt][split_p][128 * 128] raw accumulator tiles
  int split_p, ntiles, M, cout, relu;
  const float* scale;
  const float* shift;
  float alpha;
};

struct ConvArgs {
  const float* x;     // [B,H,W,c1]
  const float* x2;    // optional second source, channels [c1, c1+c2) of the logical input
  const float* res;   // optional residual [B,Ho,Wo,cout]
  float* y;           // [B,Ho,Wo,cout]
  int B, H, W;
  int c1, c2;         // c1 + c2 == desc.cin
  int Ho, Wo;
  float* ws;          // optional scratch for tail split-K partial tiles (see conv_common.h)
  size_t ws_floats;
  int mt_per_group;        // grouped GEMM: 128-row m-tile mt reads weight block mt / mt_per_group (0 = single block)
  size_t w_group_stride;   // floats between weight blocks (bytes for the pre-split weights of a register-split layer)
  int ss_group_stride;     // floats between the scale (and shift) blocks of the weight groups (0: one block for all)
  int group_valid_rows;    // grouped GEMM: rows of each group that hold data, the rest up to the group's whole tiles being zero padding
                           // (Winograd positions: tiles before padding); 0 = unknown / all.  A kernel may skip work on the padding.
  DeferredSplit* defer;    // optional: the launcher may skip its split-K reduce and describe the partial tiles here (see DeferredSplit)
  const int* group_rows;   // optional (host): grouped GEMM, data rows of EACH group (<= group_valid_rows) -- the skinny kernel skips the rest
};

constexpr size_t kSplitKScratchFloats = (size_t)48 << 20;   // 192 MiB: 768 partial 256x256 tiles (the stream-K tail of the persistent 256 x 256 kernel: up to 3 fragments for each of < 256 tail tiles)
constexpr size_t kSplitKSideFloats = (size_t)8 << 20;       // 32 MiB: split-K scratch of the ops a plan runs on a side stream (small layers)

// PEANUT_NO_PK_F32 on a kernel: no packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in its code.
//
// Why (round 6, profiles/r9r): on gfx950 a packed fp32 instruction whose LOW half takes an operand from the HIGH register of a pair
// -- `op_sel:[..1..]`, which hipcc emits freely when it packs scalar code -- returns wrong low halves (lanes 48-63) while a wave
// on the same SIMD issues fp16 / bf16 MFMAs (the emulated modes' gemm_rs / conv_rs kernels).  Found in gemm_skinny.hip next to the
// bottleneck GEMM of the two-stream head; reproduced with a hand-written `v_pk_fma_f32 ... op_sel:[0,1,0]`: 20 different results in 20
// forwards, while the same sums through op_sel_hi only, through no modifier, or unpacked are exact, as is every form next to fp32
// MFMAs or alone.  The kernels in which hipcc had produced the form carry this attribute (gemm_skinny_kernel, upsample_logits_kernel, the
// fallback ppm_conv_term_kernel, map_finish_kernel, box_post_kernel); tests/test_abi.py disassembles the library and refuses the form
// anywhere but in the test hook that demonstrates it (pkfma_canary_kernel).  A function with the attribute does not inline callees
// compiled without it (they would become real calls, the same test refuses those too): helpers such a kernel uses carry the attribute
// as well (a callee WITH it inlines anywhere), and HIP's header wrappers (__syncthreads, atomicAdd, make_float4) are spelled out.
#if defined(__HIP_DEVICE_COMPILE__)
#define PEANUT_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define PEANUT_NO_PK_F32
#endif

// paste_masks_in_image for one (output pixel, instance): the value of the M x M mask probabilities `m` at pixel (x, y) of the image
// under box `bx` (grid_sample, bilinear, zero padding, align_corners = False).  Shared by paste_masks_kernel (rcnn_ops.hip) and the
// fused paste + accumulate kernel (rcnn_post.hip) with contraction OFF: the two kernels must put the same pixels on the same side of
// the mask threshold, and with -ffp-contract=fast hipcc decided per kernel which products to fuse into which adds -- in the source
// coordinate (`(gx + 1) * M - 1`) and in the blend (round 6: one borderline pixel of a 16-frame batch differed between the entry points).
__device__ __forceinline__ float paste_value(const float* __restrict__ m, int M, const float* __restrict__ bx, int x, int y) {
#pragma clang fp contract(off)
  const float gx = ((float)x + 0.5f - bx[0]) / (bx[2] - bx[0]) * 2.f - 1.f;
  const float gy = ((float)y + 0.5f - bx[1]) / (bx[3] - bx[1]) * 2.f - 1.f;
  const float ix = ((gx + 1.f) * (float)M - 1.f) / 2.f, iy = ((gy + 1.f) * (float)M - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)M + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)M + 1.f);
  if (x0 < -1 || x0 >= M || y0 < -1 || y0 >= M) return 0.f;   // all four taps read zero padding
  const float lx = ix - fx, ly = iy - fy;
  const float wx = 1.f - lx, wy = 1.f - ly;
  auto at = [&](int yy, int xx) -> float {
    return ((unsigned)yy < (unsigned)M && (unsigned)xx < (unsigned)M) ? m[yy * M + xx] : 0.f;
  };
  return at(y0, x0) * wy * wx + at(y0, x0 + 1) * wy * lx + at(y0 + 1, x0) * ly * wx + at(y0 + 1, x0 + 1) * ly * lx;
}

inline int conv_out_dim(int n, int k, int s, int p, int d) { return (n + 2 * p - d * (k - 1) - 1) / s + 1; }

// choose (bn_tile, bk) for a layer
void conv_pick_tiles(int cin_pad, int cout, int* bn_tile, int* bk, bool pointwise = false);
// host-side packing: w is OIHW [cout][cin_real][kh][kw]; returns floats written
size_t conv_packed_floats(int cin_pad, int cout, int kh, int kw, int bn_tile);
void pack_conv_weights(const float* w_oihw, int cout, int cin_real, int cin_pad, int kh, int kw,
                       int bn_tile, int bk, float* out);
int launch_conv(const ConvDesc& d, const ConvArgs& a, hipStream_t stream);
// true unless PEANUT_PW_GLDS=0: fp32 1x1 convs / grouped GEMMs run on the LDS-DMA kernel of conv_pw.hip
bool conv_pw_enabled();
// conv_pw.hip: whether a pointwise layer / grouped GEMM runs on the 256 x 128 three-stage kernel
bool conv_pw_uses_256(int cout, long long M, int mt_per_group, int bn_tile, int cin);
int conv_pw_256_min_k();
// conv_pw.hip: ... on the 256 x 256 two-stage kernel (flush_ktiles: k-tiles per partial sum of the two-level accumulation, 0 = none)
bool conv_pw_uses_256w(int cout, long long M, int mt_per_group, int bn_tile, int cin, int flush_ktiles);
// conv_pw.hip: a 128-wide-packed short-K layer on 128 x 64 tiles (few tiles: batch-1 shapes)
bool conv_pw_narrow_tiles(int cin, int cout, long long M, int bn_tile, int mt_per_group);
// conv_pw256p.hip: ... on the persistent 256 x 128 kernel (epilogue of the previous tile inside the next tile's k-loop)
bool conv_pw_uses_256p(int cout, long long M, int mt_per_group, int bn_tile, int cin, int flush_ktiles, long long in_pixels = 0);
// gemm_skinny.hip (round 6): a grouped pointwise launch with a handful of data rows per group (the PSP pyramid's per-scale convs at batch 1)
struct ConvKParams;
bool gemm_skinny_takes(const ConvKParams& p, int bn_tile, size_t ws_floats);
int launch_gemm_skinny(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream);
// conv_pw256wp.hip: ... on the persistent 256 x 256 kernel (in-place epilogue inside the next tile's first iteration); stride 1
// only, c1 / c2 the channels of the two sources
bool conv_pw_uses_256wp(int cout, long long M, int stride, int mt_per_group, int bn_tile, int c1, int c2, int flush_ktiles);
// conv_pw_ares.hip: ... on the persistent A-resident kernel (K = 128 / 256)
bool conv_pw_uses_ares(int cin, int cout, long long M, int stride, bool two_source, int flush_ktiles, int bn_tile);

// conv_patch.hip: 3x3 convs with 16 / 32 input channels (the deep stem) on the persistent LDS-patch kernel
bool conv_patch_eligible(const ConvDesc& d, const ConvArgs& a);
const char* conv_patch_kernel_name(const ConvDesc& d, bool nchw);
// ... and its NCHW variant: the network's first conv (16 padded channels, stride 2) straight from the caller's [B][creal][H][W] input
bool conv_patch_nchw_eligible(const ConvDesc& d, int B, int Ho, int Wo, int creal);
int launch_conv_patch_nchw(const ConvDesc& d, const float* x_nchw, int creal, float* y, int B, int H, int W, int Ho, int Wo, hipStream_t stream);

// ---- Winograd F(4x4,3x3) transforms around the GEMM kernel (winograd.hip) ----
// tiles per sub-grid (th x tw), tile count and its padding to whole `gran`-row GEMM tiles (128; 256 when the position
// GEMMs run on a 256-row kernel)
void wino_geometry(int B, int H, int W, int dil, int* th, int* tw, long long* n_tiles, long long* m_pad, int gran = 128, int m = 4);
// host: OIHW 3x3 weights -> U [36][cout][cin]
void wino_transform_weights(const float* w_oihw, int cout, int cin, float* out, int m = 4);   // m = 4: U [36]..., m = 6: U [64][cout][cin]
// x [B,H,W,C] -> V [36][m_pad][C] fp32
// m_pad_total > 0: the rows of a position are shared by several tensors (the five RPN levels as one grouped GEMM): V / Mb point at this
// tensor's first tile, positions are m_pad_total rows apart
// df (optional, valid): x was not written -- the transform sums the producer's split-K partial tiles itself (DeferredSplit)
int launch_wino_input(const float* x, float* V, int B, int H, int W, int C, int dil, hipStream_t s, int gran = 128, int m = 4, long long m_pad_total = 0,
                      const DeferredSplit* df = nullptr);
// whether that transform of an input [B,H,W,C] can read deferred split-K partials (the small-problem variants only)
bool wino_input_accepts_deferred(int B, int H, int W, int C, int dil, int gran, int m);
long long wino_deferred_count();      // transforms launched with deferred input so far (process-wide; test hook)
// Mb [36][m_pad][C] -> y [B,H,W,C] = relu(scale * (A^T M A) + shift + res)
int launch_wino_output(const float* Mb, const float* scale, const float* shift, const float* res, float* y,
                       int B, int H, int W, int C, int dil, int relu, hipStream_t s, int gran = 128, int m = 4, long long m_pad_total = 0);

// ---- emulated-fp32 GEMM on the bf16 matrix cores, fp32 activations split in registers (gemm_rs.hip) ----
// whether the 256 x 256 kernel runs a [M x cout] output (mt_per_group: 128-row tiles per Winograd position, 0 = plain)
bool gemm_rs_uses_256(int cout, long long M, int mt_per_group, int bn_tile, int cin);
const char* gemm_rs_kernel_name(int cout, long long M, int mt_per_group, int bn_tile, int cin, int planes);
// the weights' pieces (planes = emulation kind), packed per (n-tile of bn_tile rows, k-tile of 16 channels, plane)
size_t sx_packed_bytes(int cin_pad, int cout, int bn_tile, int planes);
void pack_weights_sx(const float* w, int cout, int cin_real, int cin_pad, int bn_tile, int planes, float wscale, void* out);
// the power of two a layer's weights are scaled by before the split (fp16 pieces: largest |w| to 2^13..2^14; else 1)
float sx_pack_scale(const float* w, size_t n, int planes);
// conv_rs.hip: the same for a kh x kw layer, k-tiles ordered channel chunk outer / filter tap inner
size_t sx_conv_packed_bytes(int cin_pad, int cout, int kh, int kw, int bn_tile, int planes);
void pack_weights_sx_conv(const float* w_oihw, int cout, int cin_real, int cin_pad, int kh, int kw, int bn_tile, int planes,
                          float wscale, void* out);

// ---- auxiliary (HBM-bound) kernels: layout, pooling, resampling ----
int launch_nchw_to_nhwc_pad(const float* x, float* y, int B, int C, int H, int W, int Cpad, hipStream_t s);
int launch_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s);
// adaptive average pooling of [B,H,W,C] into all pyramid bins: out [B,nbins_total,C]
int launch_ppm_pool(const float* x, float* out, int B, int H, int W, int C, const int* scales, int nscales,
                    hipStream_t s, int scale_rows = 0);
// two-pass form (reads x once); scratch = ppm_pool_scratch_floats(...) floats (0 -> falls back to the above)
size_t ppm_pool_scratch_floats(int B, int H, int C, const int* scales, int nscales);
// scale_rows: row stride between the scales of `out` (0 = packed; a uniform stride lets ONE grouped GEMM take all scales)
int launch_ppm_pool2(const float* x, float* scratch, float* out, int B, int H, int W, int C, const int* scales,
                     int nscales, hipStream_t s, int scale_rows = 0);
// bilinear (align_corners=False) upsample of the pooled pyramid + channel concat: out [B,H,W,nscales*Cp]
int launch_ppm_upsample_concat(const float* table, float* out, int B, int H, int W, int Cp, const int* scales,
                               int nscales, int align_corners, hipStream_t s);
// pyramid half of the 3x3 bottleneck conv evaluated from the folded tables Q (see pspnet_aux.hip)
int launch_ppm_conv_term(const float* Q, float* R, int B, int H, int W, int C, const int* scales, int nscales,
                         int align_corners, hipStream_t s, int scale_rows = 0);
// bilinear resize of NHWC logits [B,h,w,K] to NCHW [B,K,H,W], optional sigmoid
int launch_upsample_logits(const float* lo, float* out, int B, int h, int w, int K, int H, int W,
                           int align_corners, int sigmoid, hipStream_t s);

}  // namespace peanut
