// C ABI of the window convolutions over pre-split pixel images (evae_conv_win.h): the gated convolution layers of the
// convolutional encoders (reference utils/nn.py:72-97, models/convHVAE_2level.py:21-46) between two layers of a stack, where
// activations never exist as fp32 tensors unless a caller asks for a copy.
#include "evae_conv_win.h"

namespace evae {

// fp32 channels-last [N][H][W][C] -> pixel image (rows natural or parity-planar); one thread per (pixel, 8 channels)
__global__ __launch_bounds__(256) void cw_pack_image_kernel(const float* __restrict__ x, int N, int H, int W, int C, int planar, int elu,
                                                            unsigned char* __restrict__ img) {
  const int c8n = C >> 3, nks = C >> 4;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (no early exit on the thread index: inside a 16-pixel group the threads run channel-group-major, so the last, partial group's
  // pixels sit behind thread N H W (C / 8) -- the pixel test below is the only one)
  // thread -> pixel fastest inside 16, then the 8-channel group, then the 16-pixel groups (whole 512-byte chunks per 32 lanes)
  const size_t grp = t / ((size_t)16 * c8n);
  const int r = (int)(t - grp * 16 * c8n), c8 = r >> 4, p16 = r & 15;
  const size_t m = grp * 16 + p16;
  if (m >= (size_t)N * H * W) return;
  float4 a = *reinterpret_cast<const float4*>(x + m * C + c8 * 8), b = *reinterpret_cast<const float4*>(x + m * C + c8 * 8 + 4);
  if (elu) {
    a.x = a.x > 0.f ? a.x : expm1f(a.x); a.y = a.y > 0.f ? a.y : expm1f(a.y); a.z = a.z > 0.f ? a.z : expm1f(a.z); a.w = a.w > 0.f ? a.w : expm1f(a.w);
    b.x = b.x > 0.f ? b.x : expm1f(b.x); b.y = b.y > 0.f ? b.y : expm1f(b.y); b.z = b.z > 0.f ? b.z : expm1f(b.z); b.w = b.w > 0.f ? b.w : expm1f(b.w);
  }
  size_t row = m;
  if (planar) {
    const size_t n = m / ((size_t)H * W);
    const int rem = (int)(m - n * H * W), y = rem / W, xx = rem - y * W;
    row = n * H * W + cw_planar(y, xx, H, W);
  }
  unsigned t0[4], t1[4], t2[4];
  p6_split2(a.x, a.y, t0[0], t1[0], t2[0]); p6_split2(a.z, a.w, t0[1], t1[1], t2[1]);
  p6_split2(b.x, b.y, t0[2], t1[2], t2[2]); p6_split2(b.z, b.w, t0[3], t1[3], t2[3]);
  unsigned char* o = img + p6_off64(row, c8 * 8, nks);
  *reinterpret_cast<uint4*>(o) = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  *reinterpret_cast<uint4*>(o + P6_CHUNK) = make_uint4(t1[0], t1[1], t1[2], t1[3]);
  *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = make_uint4(t2[0], t2[1], t2[2], t2[3]);
}

// The general form: x has Cx <= C real channels (the image's other channels are zero) in rows of ldx floats (channels-last) or as
// contiguous NCHW planes (nchw); optionally times ELU'(pre) with aux = ELU(pre) (fp32, same layout as x has channels-last, row
// stride lda): the gradient entering a convolution that an ELU follows.
__global__ __launch_bounds__(256) void cw_pack_image_ex_kernel(const float* __restrict__ x, long long ldx, int Cx, int nchw,
                                                               const float* __restrict__ aux, long long lda, int N, int H, int W, int C,
                                                               int planar, int elu, int up, unsigned char* __restrict__ img) {
  const int c8n = C >> 3, nks = C >> 4;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (no early exit on the thread index: inside a 16-pixel group the threads run channel-group-major, so the last, partial group's
  // pixels sit behind thread N H W (C / 8) -- the pixel test below is the only one)
  const size_t grp = t / ((size_t)16 * c8n);
  const int r = (int)(t - grp * 16 * c8n), c8 = r >> 4, p16 = r & 15;
  const size_t m = grp * 16 + p16;
  const size_t HW = (size_t)H * W;
  if (m >= (size_t)N * HW) return;
  const size_t n = m / HW;
  const int rem = (int)(m - n * HW);
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c8 * 8 + k;
    float a = 0.f;
    if (c < Cx) {
      if (up) {                                  // x is the half-resolution tensor: nearest-neighbour upsampling by 2 (nn.Upsample(scale_factor=2))
        const int y = rem / W, xx = rem - y * W, h2 = H >> 1, w2 = W >> 1;
        const size_t ms = (n * h2 + (y >> 1)) * w2 + (xx >> 1);
        a = nchw ? x[(n * Cx + c) * ((size_t)h2 * w2) + (size_t)(y >> 1) * w2 + (xx >> 1)] : x[ms * ldx + c];
      } else {
        a = nchw ? x[(n * Cx + c) * HW + rem] : x[m * ldx + c];
      }
      if (aux) { const float e = aux[m * lda + c]; a *= e > 0.f ? 1.0f : e + 1.0f; }
      if (elu) a = a > 0.f ? a : expm1f(a);
    }
    v[k] = a;
  }
  size_t row = m;
  if (planar) {
    const int y = rem / W, xx = rem - y * W;
    row = n * HW + cw_planar(y, xx, H, W);
  }
  unsigned t0[4], t1[4], t2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) p6_split2(v[2 * i], v[2 * i + 1], t0[i], t1[i], t2[i]);
  unsigned char* o = img + p6_off64(row, c8 * 8, nks);
  *reinterpret_cast<uint4*>(o) = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  *reinterpret_cast<uint4*>(o + P6_CHUNK) = make_uint4(t1[0], t1[1], t1[2], t1[3]);
  *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = make_uint4(t2[0], t2[1], t2[2], t2[3]);
}

// [dh | dg] = [v s | v out (1 - s)] for an upstream gradient v that exists as an fp32 tensor (the layer above ran outside the
// stack): v, s fp32 natural [rows][C]; out from its pixel image (rows natural or planar); result -> pixel image in the image's row
// order (2 C channels) and / or fp32 natural [rows][2 C]
__global__ __launch_bounds__(256) void cw_gate_bwd_image_kernel(const float* __restrict__ v, const unsigned char* __restrict__ eimg, int planar,
                                                                const float* __restrict__ s, int N, int H, int W, int C,
                                                                unsigned char* __restrict__ oimg, float* __restrict__ out_f) {
  const int c8n = C >> 3, nks_e = C >> 4, nks_o = C >> 3;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t grp = t / ((size_t)16 * c8n);
  const int r = (int)(t - grp * 16 * c8n), c8 = r >> 4, p16 = r & 15;
  const size_t m = grp * 16 + p16;
  if (m >= (size_t)N * H * W) return;
  size_t row = m;
  if (planar) {
    const size_t n = m / ((size_t)H * W);
    const int rem = (int)(m - n * H * W), y = rem / W, xx = rem - y * W;
    row = n * H * W + cw_planar(y, xx, H, W);
  }
  const int ch = c8 * 8;
  const float4 v0 = *reinterpret_cast<const float4*>(v + m * C + ch), v1 = *reinterpret_cast<const float4*>(v + m * C + ch + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(s + m * C + ch), s1 = *reinterpret_cast<const float4*>(s + m * C + ch + 4);
  const unsigned char* e = eimg + p6_off64(row, ch, nks_e);
  const uint4 e0 = *reinterpret_cast<const uint4*>(e), e1 = *reinterpret_cast<const uint4*>(e + P6_CHUNK), e2 = *reinterpret_cast<const uint4*>(e + 2 * P6_CHUNK);
  const unsigned w0[4] = {e0.x, e0.y, e0.z, e0.w}, w1[4] = {e1.x, e1.y, e1.z, e1.w}, w2[4] = {e2.x, e2.y, e2.z, e2.w};
  const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  float dh[8], dg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int sh = 16 * (k & 1);
    const float ov = (__uint_as_float(((w2[k >> 1] >> sh) & 0xFFFFu) << 16) + __uint_as_float(((w1[k >> 1] >> sh) & 0xFFFFu) << 16)) +
                     __uint_as_float(((w0[k >> 1] >> sh) & 0xFFFFu) << 16);
    dh[k] = vv[k] * sv[k];
    dg[k] = vv[k] * ov * (1.0f - sv[k]);
  }
  auto put_img = [&](int chan, const float* a) {
    unsigned t0[4], t1[4], t2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p6_split2(a[2 * i], a[2 * i + 1], t0[i], t1[i], t2[i]);
    unsigned char* o = oimg + p6_off64(row, chan, nks_o);
    *reinterpret_cast<uint4*>(o) = make_uint4(t0[0], t0[1], t0[2], t0[3]);
    *reinterpret_cast<uint4*>(o + P6_CHUNK) = make_uint4(t1[0], t1[1], t1[2], t1[3]);
    *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = make_uint4(t2[0], t2[1], t2[2], t2[3]);
  };
  if (oimg) { put_img(ch, dh); put_img(C + ch, dg); }
  if (out_f) {
    float* op = out_f + m * 2 * C + ch;
    *reinterpret_cast<float4*>(op) = make_float4(dh[0], dh[1], dh[2], dh[3]); *reinterpret_cast<float4*>(op + 4) = make_float4(dh[4], dh[5], dh[6], dh[7]);
    *reinterpret_cast<float4*>(op + C) = make_float4(dg[0], dg[1], dg[2], dg[3]); *reinterpret_cast<float4*>(op + C + 4) = make_float4(dg[4], dg[5], dg[6], dg[7]);
  }
}

static bool cw_geometry_ok(const evae_conv_desc_t* d) {
  if (!d || d->N <= 0 || d->KH != d->KW || (d->KH & 1) == 0 || d->pad * 2 + 1 != d->KH) return false;
  if (d->stride != 1 && d->stride != 2) return false;
  if (d->stride == 2 && ((d->H | d->W) & 1)) return false;
  if (d->H != d->W) return false;                                  // (square grids: the only ones the encoders have)
  if ((long long)d->N * d->H * d->W >= (1ll << 31)) return false;
  return true;
}
// forward tile shape: 0 = none, 1 = 128 pixels x 64 gated outputs (Co % 64 == 0), 2 = 256 pixels x 32 gated outputs
static int cw_fwd_shape(const evae_conv_desc_t* d) {
  if (!cw_geometry_ok(d) || d->C % 16 != 0 || d->Co % 32 != 0) return 0;
  int plo, phi;
  (void)cw_taps_fwd(d->KH, d->stride, d->pad, &plo, &phi);
  const int OH = d->H / d->stride;
  if (d->Co % 64 == 0 && cw_window_slots(OH, OH, plo, phi, 128) <= 320) return 1;
  if (cw_window_slots(OH, OH, plo, phi, 256) <= 576) return 2;
  return 0;
}
// data gradient (into C channels): 1 = 32-column tiles (C == 32), 2 = 64-column tiles (C % 64 == 0)
static int cw_dgrad_shape(const evae_conv_desc_t* d) {
  if (!cw_geometry_ok(d) || d->Co % 8 != 0) return 0;
  const int shape = d->C == 32 ? 1 : (d->C % 16 == 0 ? 2 : 0);
  if (!shape) return 0;
  const int OH = d->H / d->stride;
  for (int py = 0; py < d->stride; ++py)
    for (int px = 0; px < d->stride; ++px) {
      int plo, phi;
      (void)cw_taps_dgrad(d->KH, d->stride, d->pad, py, px, &plo, &phi);
      if (cw_window_slots(OH, OH, plo, phi, 256) > 576) return 0;
    }
  return shape;
}
// weight gradient: 32 input channels per launch, every tap of the filter as a column tile of one launch (5 x 5: 13 + 12).  Variant:
// 1 = 5 x 5 stride 1 (window 192 slots, 128 merged channels), 2 = 3 x 3 stride 1, 3 = 3 x 3 stride 2 with <= 64 merged channels
// (window 320), 4 = 3 x 3 stride 2 with <= 128 (window 256)
static int cw_wgrad_ok(const evae_conv_desc_t* d, int gated = 1) {
  const int CCq = gated ? 2 * d->Co : d->Co;
  if (!cw_geometry_ok(d) || d->C % 16 != 0 || CCq % 16 != 0 || CCq > 128) return 0;
  const int OH = d->H / d->stride;
  const int need = cw_wgrad_window_slots(OH, OH, d->KH, d->pad, d->stride, 32);
  if (d->stride == 1 && d->KH == 5) return need <= 192 ? 1 : 0;
  if (d->stride == 1 && d->KH == 3) {
    // a window for several stages (variants 5 / 6): four stages in 288 slots with <= 64 merged channels (aligned runs where the grid
    // is wide), two stages in 224 slots with <= 128
    if (CCq <= 64 && cw_wgrad_window_slots(OH, OH, 3, d->pad, 1, 128, 128) <= 288) return 5;
    if (CCq > 64 && cw_wgrad_window_slots(OH, OH, 3, d->pad, 1, 64) <= 224) return 6;
    return need <= 192 ? 2 : ((CCq <= 64 && need <= 320) ? 3 : 0);    // (wide grids: the 320-slot window of variant 3)
  }
  if (d->stride == 2 && d->KH == 3) {
    if (CCq <= 64 && need <= 320) return 3;
    if (need <= 256) return 4;
    // wider grids (fully_conv's 64 -> 32 and 32 -> 16 layers): the window on the input grid fits LDS for ONE channel group at a time
    // -- every group as a two-taps-per-tile launch (variants 7 / 8)
    if (CCq <= 64 && need <= 416) return 7;
    return need <= 288 ? 8 : 0;
  }
  return 0;
}
constexpr int CW_WGRAD_BLOCKS = 256;
constexpr int CW_FIRST_SLOTS = 768, CW_FIRST_WGRAD_BLOCKS = 512;
// first layer (one input channel): <= 50 taps, stride 1, 'same' padding, Co a multiple of 32; weight gradient: 2 Co <= 64
static int cw_first_ok(const evae_conv_desc_t* d, int wgrad) {
  if (!cw_geometry_ok(d) || d->C != 1 || d->stride != 1 || d->KH * d->KW > 49 || d->Co % 32 != 0) return 0;
  if (wgrad && 2 * d->Co > 64) return 0;
  return cw_window_slots(d->H, d->W, d->pad, d->pad, 256) <= CW_FIRST_SLOTS;
}

}  // namespace evae

using namespace evae;

extern "C" size_t evae_cw_image_bytes(long long rows, int channels) {
  if (rows <= 0 || channels <= 0 || channels % 16 != 0) return 0;
  return p6_image_bytes((int)rows, channels / 16);
}

extern "C" int evae_cw_supported(const evae_conv_desc_t* d, int what) {
  if (what == 0) return cw_fwd_shape(d) != 0;
  if (what == 1) return cw_dgrad_shape(d) != 0;
  if (what == 2) return cw_wgrad_ok(d);
  if (what == 3) return cw_first_ok(d, 0);
  if (what == 4) return cw_first_ok(d, 1);
  return 0;
}

extern "C" size_t evae_cw_workspace_bytes(const evae_conv_desc_t* d, int what) {
  if (!d) return 256;
  const int taps = d->KH * d->KW;
  if (what == 0) {
    const int bn = cw_fwd_shape(d) == 1 ? 128 : 64, tiles_n = cdiv(2 * d->Co, bn);
    return p6_image_bytes(tiles_n * bn, (d->C / 16) * taps) + 8192;
  }
  if (what == 1) {
    const int bn = d->C == 32 ? 32 : 64, tiles_n = cdiv(d->C, bn);
    return (size_t)d->stride * d->stride * (p6_image_bytes(tiles_n * bn, (2 * d->Co / 16) * taps) + 8192);   // one filter image per parity class (upper bound)
  }
  if (what == 5 || what == 6) {      // residual block: a plain filter image, 64-column tiles
    const int tiles_n = cdiv(d->Co, 64), wrows = (tiles_n * 64 + 127) / 128 * 128;
    return p6_image_bytes(wrows, (d->C / 16) * taps) + 8192;
  }
  const int CCw = what == 7 ? d->Co : 2 * d->Co;
  return align_up((size_t)CW_WGRAD_BLOCKS * CCw * taps * d->C * sizeof(float), 256) + align_up((size_t)CW_WGRAD_BLOCKS * CCw * sizeof(float), 256) + 256;
}

extern "C" int evae_cw_pack_image(const float* x, int N, int H, int W, int C, int planar, void* img, evae_stream_t stream_) {
  // planar: bit 0 = parity-planar rows, bit 1 = the image of ELU(x) instead of x
  EVAE_REQUIRE(x && img && N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "cw_pack_image: bad arguments (channels a multiple of 16)");
  EVAE_REQUIRE(!(planar & 1) || ((H | W) & 1) == 0, "cw_pack_image: parity-planar rows need even H and W");
  const size_t rows = (size_t)N * H * W, rows16 = (rows + 15) / 16 * 16;
  const size_t total = rows16 * (size_t)(C / 8);
  cw_pack_image_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(x, N, H, W, C, planar & 1, (planar >> 1) & 1, (unsigned char*)img);
  return check_launch("cw_pack_image_kernel");
}

extern "C" int evae_cw_pack_image_ex(const float* x, long long ldx, int Cx, int nchw, const float* aux, long long lda, int N, int H, int W,
                                     int C, int flags, void* img, evae_stream_t stream_) {
  // flags: bit 0 = parity-planar rows, bit 1 = the image of ELU(x), bit 2 = x is [N][H / 2][W / 2]: nearest upsampling by 2 on the way
  EVAE_REQUIRE(x && img && N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0 && Cx > 0 && Cx <= C, "cw_pack_image_ex: bad arguments");
  EVAE_REQUIRE(!(flags & 4) || (((H | W) & 1) == 0 && !aux), "cw_pack_image_ex: upsampling needs even H and W (and takes no aux)");
  EVAE_REQUIRE(nchw || ldx >= Cx, "cw_pack_image_ex: row stride smaller than the channel count");
  EVAE_REQUIRE(!aux || lda >= Cx, "cw_pack_image_ex: aux row stride smaller than the channel count");
  EVAE_REQUIRE(!(flags & 1) || ((H | W) & 1) == 0, "cw_pack_image_ex: parity-planar rows need even H and W");
  const size_t rows = (size_t)N * H * W, rows16 = (rows + 15) / 16 * 16;
  const size_t total = rows16 * (size_t)(C / 8);
  cw_pack_image_ex_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(x, ldx, Cx, nchw, aux, lda, N, H, W, C, flags & 1,
                                                                                     (flags >> 1) & 1, (flags >> 2) & 1, (unsigned char*)img);
  return check_launch("cw_pack_image_ex_kernel");
}

// gradient of nearest upsampling by 2: dx [N][H / 2][W / 2][C] = the sum of the four dy [N][H][W][ld] pixels it was copied to
__global__ __launch_bounds__(256) void cw_upsample2_bwd_kernel(const float* __restrict__ dy, long long ld, int N, int H, int W, int C,
                                                               float* __restrict__ dx) {
  const int c4n = C >> 2, h2 = H >> 1, w2 = W >> 1;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * h2 * w2 * c4n) return;
  const int c4 = (int)(t % c4n);
  const size_t p = t / c4n;
  const int x = (int)(p % w2), y = (int)((p / w2) % h2);
  const size_t n = p / ((size_t)w2 * h2);
  const float* s = dy + ((n * H + 2 * y) * W + 2 * x) * ld + c4 * 4;
  const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + ld);
  const float4 c = *reinterpret_cast<const float4*>(s + (size_t)W * ld), d = *reinterpret_cast<const float4*>(s + (size_t)W * ld + ld);
  *reinterpret_cast<float4*>(dx + p * C + c4 * 4) = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
}

extern "C" int evae_cw_upsample2_bwd(const float* dy, long long ld, int N, int H, int W, int C, float* dx, evae_stream_t stream_) {
  EVAE_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && ((H | W) & 1) == 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0, "cw_upsample2_bwd: bad arguments");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  cw_upsample2_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(dy, ld, N, H, W, C, dx);
  return check_launch("cw_upsample2_bwd_kernel");
}

extern "C" int evae_cw_gate_bwd_image(const float* v, const void* eimg, int planar, const float* s, int N, int H, int W, int C, void* oimg,
                                      float* out_f, evae_stream_t stream_) {
  EVAE_REQUIRE(v && eimg && s && (oimg || out_f) && N > 0 && C > 0 && C % 16 == 0, "cw_gate_bwd_image: bad arguments");
  EVAE_REQUIRE(!planar || ((H | W) & 1) == 0, "cw_gate_bwd_image: parity-planar rows need even H and W");
  const size_t rows = (size_t)N * H * W, rows16 = (rows + 15) / 16 * 16;
  const size_t total = rows16 * (size_t)(C / 8);
  cw_gate_bwd_image_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(v, (const unsigned char*)eimg, planar, s, N, H, W, C,
                                                                                      (unsigned char*)oimg, out_f);
  return check_launch("cw_gate_bwd_image_kernel");
}

// Forward of a gated layer: out = (conv(x, wh) + bh) * sigmoid(conv(x, wg) + bg).  ximg: the input's pixel image (rows natural for
// a stride-1 layer, parity-planar for a stride-2 one); oimg: the output's image (rows parity-planar when out_planar: the consumer has
// stride 2); out_s: the gate, fp32 natural [N OH OW][Co]; out_f: fp32 copy of the output or NULL.
extern "C" int evae_cw_fwd_gated(const void* ximg, const evae_conv_desc_t* d, const float* wh, const float* bh, const float* wg,
                                 const float* bg, void* oimg, int out_planar, float* out_s, float* out_f, void* ws, size_t ws_bytes,
                                 evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int shape = cw_fwd_shape(d);
  EVAE_REQUIRE(shape != 0, "cw_fwd_gated: unsupported geometry");
  EVAE_REQUIRE(ximg && wh && wg && (oimg || out_f) && ws, "cw_fwd_gated: null pointer");
  EVAE_REQUIRE(ws_bytes >= evae_cw_workspace_bytes(d, 0), "cw_fwd_gated: workspace too small");
  const int st = d->stride, OH = d->H / st, K = d->KH;
  EVAE_REQUIRE(!out_planar || (OH & 1) == 0, "cw_fwd_gated: parity-planar output rows need an even output grid");
  int plo, phi;
  const CwTaps tp = cw_taps_fwd(K, st, d->pad, &plo, &phi);
  const int ncg = d->C / 16, nks_w = cw_ksteps(tp, ncg);
  const int bn = shape == 1 ? 128 : 64, tiles_n = cdiv(2 * d->Co, bn), wrows = (tiles_n * bn + 127) / 128 * 128;
  unsigned char* iw = (unsigned char*)ws;
  cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256, 0, stream>>>(wh, wg, d->Co, d->C, K, K, tp, ncg, 0, bn, wrows, nks_w, iw);
  int rc = check_launch("cw_pack_filter_kernel");
  if (rc) return rc;
  ConvWinArgs g = {};
  g.xin = (const unsigned char*)ximg; g.nks_in = ncg; g.ncg = ncg; g.N = d->N; g.H = OH; g.W = OH; g.plo = plo; g.phi = phi; g.taps = tp;
  g.istride = st * st * OH * OH;
  for (int s2 = 0; s2 < st * st; ++s2) g.ioff[s2] = s2 * OH * OH;
  g.wimg = iw; g.nks_w = nks_w; g.Co = d->Co; g.tiles_n = tiles_n; g.bias0 = bh; g.bias1 = bg;
  g.out_planar = out_planar;
  g.oimg = (unsigned char*)oimg; g.nks_o = d->Co / 16; g.out_s = out_s; g.out_f = out_f; g.ldo = d->Co;
  if (shape == 1) {
    g.nsp = (cw_window_slots(OH, OH, plo, phi, 128) + 31) / 32;
    return launch_conv_win<CW_FWD_GATED, 2, 2, 320>(g, stream, "cw_fwd_gated");
  }
  g.nsp = (cw_window_slots(OH, OH, plo, phi, 256) + 31) / 32;
  return launch_conv_win<CW_FWD_GATED, 4, 2, 576>(g, stream, "cw_fwd_gated");
}

// Data gradient of a gated layer (C -> Co) with the gate derivative of the layer below in the epilogue.  dyimg: the merged [dh | dg]
// image of THIS layer's output (rows planar when dy_planar); eimg / e_s: forward output (image) and gate (fp32 natural) of the
// layer below = this layer's input, image rows planar when this layer has stride 2; result: [dh | dg] of the layer below as an
// image in eimg's row order (oimg, 2 C channels) and / or fp32 natural [N H W][2 C] (out_f).
extern "C" int evae_cw_bwd_data_gate(const void* dyimg, int dy_planar, const evae_conv_desc_t* d, const float* wh, const float* wg,
                                     const void* eimg, const float* e_s, void* oimg, float* out_f, void* ws, size_t ws_bytes,
                                     evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int shape = cw_dgrad_shape(d);
  EVAE_REQUIRE(shape != 0, "cw_bwd_data_gate: unsupported geometry");
  EVAE_REQUIRE(dyimg && wh && wg && eimg && e_s && (oimg || out_f) && ws, "cw_bwd_data_gate: null pointer");
  EVAE_REQUIRE(ws_bytes >= evae_cw_workspace_bytes(d, 1), "cw_bwd_data_gate: workspace too small");
  const int st = d->stride, OH = d->H / st, K = d->KH, C = d->C, Co = d->Co;
  const int ncg = 2 * Co / 16, bn = shape == 1 ? 32 : 64, tiles_n = cdiv(C, bn), wrows = (tiles_n * bn + 127) / 128 * 128;
  size_t used = 0;
  for (int py = 0; py < st; ++py)
    for (int px = 0; px < st; ++px) {
      int plo, phi;
      const CwTaps tp = cw_taps_dgrad(K, st, d->pad, py, px, &plo, &phi);
      const int nks_w = cw_ksteps(tp, ncg);
      ConvWinArgs g = {};
      g.N = d->N; g.H = OH; g.W = OH; g.plo = plo; g.phi = phi; g.taps = tp;
      g.ostride = st * st * OH * OH; g.ooff = (py * st + px) * OH * OH;
      g.nat_h = d->H; g.nat_w = d->W; g.nat_s = st; g.nat_y = py; g.nat_x = px;
      g.Co = C; g.tiles_n = tiles_n;
      g.oimg = (unsigned char*)oimg; g.nks_o = 2 * C / 16; g.out_f = out_f; g.ldo = 2 * C;
      g.eimg = (const unsigned char*)eimg; g.nks_e = C / 16; g.e_s = e_s;
      if (nks_w == 0) {
        // no tap reaches this class (stride larger than the filter): its gradient is zero -- not a case the encoders have
        set_error("cw_bwd_data_gate: a parity class without taps (filter smaller than the stride)");
        return EVAE_EINVAL;
      }
      unsigned char* iw = (unsigned char*)ws + used;
      used += p6_image_bytes(wrows, nks_w) + 8192;
      cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256, 0, stream>>>(wh, wg, Co, C, K, K, tp, ncg, 1, bn, wrows, nks_w, iw);
      int rc = check_launch("cw_pack_filter_kernel");
      if (rc) return rc;
      g.xin = (const unsigned char*)dyimg; g.nks_in = ncg; g.ncg = ncg; g.istride = OH * OH; g.in_planar = dy_planar;
      g.wimg = iw; g.nks_w = nks_w;
      g.nsp = (cw_window_slots(OH, OH, plo, phi, 256) + 31) / 32;
      if (shape == 1) rc = launch_conv_win<CW_DGRAD_GATE, 4, 1, 576>(g, stream, "cw_bwd_data_gate");
      else rc = launch_conv_win<CW_DGRAD_GATE, 4, 2, 576>(g, stream, "cw_bwd_data_gate");
      if (rc) return rc;
    }
  return EVAE_OK;
}

// Weight gradient of a gated layer from the merged-gradient image (rows planar when dy_planar) and the input image (rows natural
// for a stride-1 layer, parity-planar for a stride-2 one): dw [2 Co][C][K][K] (h rows then g rows: nn.Conv2d layout), db [2 Co].
static int cw_bwd_weight_impl(const void* dyimg, int dy_planar, const void* ximg, const evae_conv_desc_t* d, int gated, float* dw, float* db,
                              void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int variant = cw_wgrad_ok(d, gated);
  EVAE_REQUIRE(variant != 0, "cw_bwd_weight: unsupported geometry");
  EVAE_REQUIRE(dyimg && ximg && dw && ws, "cw_bwd_weight: null pointer");
  EVAE_REQUIRE(ws_bytes >= evae_cw_workspace_bytes(d, gated ? 2 : 7), "cw_bwd_weight: workspace too small");
  const int K = d->KH, taps = K * K, CC = gated ? 2 * d->Co : d->Co, C = d->C, st = d->stride, OH = d->H / st;
  EVAE_REQUIRE(!dy_planar || (OH & 1) == 0, "cw_bwd_weight: parity-planar rows need an even grid");
  float* part = (float*)ws;
  float* dbp = (float*)((char*)ws + align_up((size_t)CW_WGRAD_BLOCKS * CC * taps * C * sizeof(float), 256));
  const int PW = OH * st + 2 * d->pad;
  CwWgradArgs g = {};
  g.dyimg = (const unsigned char*)dyimg; g.nks_dy = CC / 16; g.dy_planar = dy_planar;
  g.ximg = (const unsigned char*)ximg; g.nks_x = C / 16; g.nseg = 1;
  g.N = d->N; g.H = OH; g.W = OH; g.plo = d->pad; g.phi = d->pad; g.xs = st; g.x_planar = st == 2;
  g.ntap_f = taps; g.Cin = C; g.CC = CC; g.part = part;
  int nblk = 0;
  bool first = true;
  const bool single = variant == 7 || variant == 8;       // one channel group per launch
  for (int cp = 0; cp < (single ? C / 16 : (C / 16 + 1) / 2); ++cp) {  // channel-group pairs of the input (an odd last group: half a pair)
    g.xcg0 = single ? cp : 2 * cp;
    for (int t0 = 0; t0 < taps; ) {
      const int nt = taps == 25 ? (t0 == 0 ? 13 : 12) : taps;
      for (int t = 0; t < nt; ++t) { const int tt = t0 + t; g.tile_seg[t] = 0; g.tile_to[t] = (tt / K) * PW + tt % K; g.tile_tap[t] = tt; }
      g.dbpart = first ? dbp : nullptr;
      int rc;
      if (variant == 1) rc = nt == 13 ? launch_conv_wgrad_win<13, 192, 8>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight")
                                      : launch_conv_wgrad_win<12, 192, 8>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      else if (single) {
        for (int t = 0; t < 10; ++t) { g.tile_to[t] = t < 9 ? (t / K) * PW + t % K : 0; g.tile_tap[t] = t < 9 ? t : -1; }
        rc = variant == 7 ? launch_conv_wgrad_win<5, 416, 4, 1, true>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight")
                          : launch_conv_wgrad_win<5, 288, 8, 1, true>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      }
      else if ((variant == 5 || variant == 6) && 2 * cp + 1 == C / 16) {
        // the odd last channel group: two taps per column tile (five tiles for the nine taps)
        for (int t = 0; t < 10; ++t) { g.tile_to[t] = t < 9 ? (t / K) * PW + t % K : 0; g.tile_tap[t] = t < 9 ? t : -1; }
        rc = variant == 5 ? launch_conv_wgrad_win<5, 288, 4, 4, true>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight")
                          : launch_conv_wgrad_win<5, 224, 8, 2, true>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      }
      else if (variant == 5) rc = launch_conv_wgrad_win<9, 288, 4, 4>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      else if (variant == 6) rc = launch_conv_wgrad_win<9, 224, 8, 2>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      else if (variant == 2 && CC <= 64) rc = launch_conv_wgrad_win<9, 192, 4>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      else if (variant == 2) rc = launch_conv_wgrad_win<9, 192, 8>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      else if (variant == 3) rc = launch_conv_wgrad_win<9, 320, 4>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      else rc = launch_conv_wgrad_win<9, 256, 8>(g, CW_WGRAD_BLOCKS, stream, "cw_bwd_weight");
      if (rc) return rc;
      nblk = cdiv(g.nchunk, g.cper);
      first = false;
      t0 += nt;
    }
  }
  cw_wgrad_finish_kernel<<<cw_wgrad_finish_blocks(CC, taps, C), 256, 0, stream>>>(part, dbp, nblk, CC, taps, C, dw, db);
  return check_launch("cw_wgrad_finish_kernel");
}

extern "C" int evae_cw_bwd_weight(const void* dyimg, int dy_planar, const void* ximg, const evae_conv_desc_t* d, float* dw, float* db,
                                  void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return cw_bwd_weight_impl(dyimg, dy_planar, ximg, d, 1, dw, db, ws, ws_bytes, stream_);
}

// ---- residual blocks y = x + conv(ELU(x), w) + b (reference models/fully_conv.py:13-23; C == Co, stride 1, 'same' padding) on pixel images ----
static int cw_res_ok(const evae_conv_desc_t* d) {
  if (!cw_geometry_ok(d) || d->stride != 1 || d->C != d->Co || d->C % 16 != 0 || d->Co > 128) return 0;
  if (cw_window_slots(d->H, d->W, d->pad, d->pad, 256) > 576) return 0;
  return cw_wgrad_ok(d, 0) != 0;
}
// few pixels (256-pixel row tiles would leave CUs without a block, or with one and nobody to cover its waits): 128-pixel row tiles,
// same 64 columns (2 x 2 waves of 64 x 32)
static bool cw_res_small(const evae_conv_desc_t* d, int plo, int phi) {
  const long long M = (long long)d->N * d->H * d->W;
  return cdiv((int)((M + 255) / 256), 1) * cdiv(d->C, 64) <= 320 && cw_window_slots(d->H, d->W, plo, phi, 128) <= 320;
}
extern "C" int evae_cw_res_supported(const evae_conv_desc_t* d) { return cw_res_ok(d); }

// The filter images of a run of n <= 16 same-shaped blocks in ONE launch: forward images behind fwd_imgs, data-gradient images behind
// bwd_imgs (either may be NULL), evae_cw_workspace_bytes(d, 5) apart; evae_cw_res_fwd / _bwd_data take one as `ws` with w == NULL.
extern "C" int evae_cw_res_pack_filters(const evae_conv_desc_t* d, int n, const void* const* w, void* fwd_imgs, void* bwd_imgs,
                                        evae_stream_t stream_) {
  EVAE_REQUIRE(cw_res_ok(d), "cw_res_pack_filters: unsupported geometry");
  EVAE_REQUIRE(n >= 1 && n <= 16 && w && (fwd_imgs || bwd_imgs), "cw_res_pack_filters: 1 .. 16 filters, at least one destination");
  CwPackSet s = {};
  for (int i = 0; i < n; ++i) { EVAE_REQUIRE(w[i], "cw_res_pack_filters: null filter"); s.w[i] = (const float*)w[i]; }
  const int K = d->KH, C = d->C, tiles_n = cdiv(C, 64);
  int plo, phi;
  s.tpf = cw_taps_fwd(K, 1, d->pad, &plo, &phi);
  s.tpb = cw_taps_dgrad(K, 1, d->pad, 0, 0, &plo, &phi);
  s.C = C; s.K = K; s.ncg = C / 16; s.rows_img = (tiles_n * 64 + 127) / 128 * 128; s.nks = cw_ksteps(s.tpf, s.ncg);
  EVAE_REQUIRE(cw_ksteps(s.tpb, s.ncg) == s.nks, "cw_res_pack_filters: forward and data-gradient contractions differ");
  s.fwd = (unsigned char*)fwd_imgs; s.bwd = (unsigned char*)bwd_imgs; s.stride = evae_cw_workspace_bytes(d, 5);
  cw_pack_filter_set_kernel<<<dim3((unsigned)(((size_t)s.rows_img * s.nks * 2 + 255) / 256), n, 2), 256, 0, (hipStream_t)stream_>>>(s);
  return check_launch("cw_pack_filter_set_kernel");
}

// aimg: the image of ELU(x) (natural rows); x: fp32 [N H W][C]; y = x + conv(ELU(x)) + b -> out_f (fp32) and / or oimg = the image of ELU(y)
extern "C" int evae_cw_res_fwd(const void* aimg, const evae_conv_desc_t* d, const float* w, const float* b, const float* x, float* out_f,
                               void* oimg, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(cw_res_ok(d), "cw_res_fwd: unsupported geometry");
  EVAE_REQUIRE(aimg && x && (out_f || oimg) && ws && ws_bytes >= evae_cw_workspace_bytes(d, 5), "cw_res_fwd: null pointer / workspace too small");
  const int K = d->KH, C = d->C, ncg = C / 16, tiles_n = cdiv(C, 64), wrows = (tiles_n * 64 + 127) / 128 * 128;
  int plo, phi;
  const CwTaps tp = cw_taps_fwd(K, 1, d->pad, &plo, &phi);
  const int nks_w = cw_ksteps(tp, ncg);
  unsigned char* iw = (unsigned char*)ws;
  if (w) {                                       // (w == NULL: ws holds the filter image evae_cw_res_pack_filters wrote)
    cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256, 0, stream>>>(w, nullptr, C, C, K, K, tp, ncg, 2, 64, wrows, nks_w, iw);
    int rc = check_launch("cw_pack_filter_kernel");
    if (rc) return rc;
  }
  ConvWinArgs g = {};
  g.xin = (const unsigned char*)aimg; g.nks_in = ncg; g.ncg = ncg; g.N = d->N; g.H = d->H; g.W = d->W; g.plo = plo; g.phi = phi; g.taps = tp;
  g.wimg = iw; g.nks_w = nks_w; g.Co = C; g.tiles_n = tiles_n; g.bias0 = b;
  g.oimg = (unsigned char*)oimg; g.nks_o = ncg; g.out_f = out_f; g.ldo = C; g.e_s = x;
  if (cw_res_small(d, plo, phi)) {
    g.nsp = (cw_window_slots(d->H, d->W, plo, phi, 128) + 31) / 32;
    return launch_conv_win<CW_RES_FWD, 2, 1, 320>(g, stream, "cw_res_fwd");
  }
  g.nsp = (cw_window_slots(d->H, d->W, plo, phi, 256) + 31) / 32;
  return launch_conv_win<CW_RES_FWD, 4, 2, 576>(g, stream, "cw_res_fwd");
}

// dx = dy + ELU'(x) * conv_transpose(dy, w): dyimg = the image of dy, dy_f the same gradient in fp32, aimg = the image of ELU(x);
// -> dx_f (fp32) and / or dximg (the image of dx: the operand of the block below's gradients)
extern "C" int evae_cw_res_bwd_data(const void* dyimg, const evae_conv_desc_t* d, const float* w, const void* aimg, const float* dy_f,
                                    float* dx_f, void* dximg, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(cw_res_ok(d), "cw_res_bwd_data: unsupported geometry");
  EVAE_REQUIRE(dyimg && aimg && dy_f && (dx_f || dximg) && ws && ws_bytes >= evae_cw_workspace_bytes(d, 5),
               "cw_res_bwd_data: null pointer / workspace too small");
  const int K = d->KH, C = d->C, ncg = C / 16, tiles_n = cdiv(C, 64), wrows = (tiles_n * 64 + 127) / 128 * 128;
  int plo, phi;
  const CwTaps tp = cw_taps_dgrad(K, 1, d->pad, 0, 0, &plo, &phi);
  const int nks_w = cw_ksteps(tp, ncg);
  unsigned char* iw = (unsigned char*)ws;
  if (w) {                                       // (w == NULL: ws holds the data-gradient filter image evae_cw_res_pack_filters wrote)
    cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256, 0, stream>>>(w, nullptr, C, C, K, K, tp, ncg, 1, 64, wrows, nks_w, iw);
    int rc = check_launch("cw_pack_filter_kernel");
    if (rc) return rc;
  }
  ConvWinArgs g = {};
  g.xin = (const unsigned char*)dyimg; g.nks_in = ncg; g.ncg = ncg; g.N = d->N; g.H = d->H; g.W = d->W; g.plo = plo; g.phi = phi; g.taps = tp;
  g.wimg = iw; g.nks_w = nks_w; g.Co = C; g.tiles_n = tiles_n;
  g.oimg = (unsigned char*)dximg; g.nks_o = ncg; g.out_f = dx_f; g.ldo = C; g.e_s = dy_f;
  g.eimg = (const unsigned char*)aimg; g.nks_e = ncg;
  if (cw_res_small(d, plo, phi)) {
    g.nsp = (cw_window_slots(d->H, d->W, plo, phi, 128) + 31) / 32;
    return launch_conv_win<CW_RES_BWD, 2, 1, 320>(g, stream, "cw_res_bwd_data");
  }
  g.nsp = (cw_window_slots(d->H, d->W, plo, phi, 256) + 31) / 32;
  return launch_conv_win<CW_RES_BWD, 4, 2, 576>(g, stream, "cw_res_bwd_data");
}

// weight gradient of a plain (un-gated) stride-1 layer: dw [Co][C][K][K], db [Co] from dy's image and the input's image
extern "C" int evae_cw_bwd_weight_plain(const void* dyimg, const void* ximg, const evae_conv_desc_t* d, float* dw, float* db, void* ws,
                                        size_t ws_bytes, evae_stream_t stream_) {
  return cw_bwd_weight_impl(dyimg, 0, ximg, d, 0, dw, db, ws, ws_bytes, stream_);
}

// ---- plain convolutions (3 x 3, stride 1 or 2, 'same' padding) on pixel images: fully_conv's weight-normed convolutions outside its
// residual runs (models/fully_conv.py:41-58).  d->C / d->Co are the REAL channel counts; an image carries them rounded up to 16
// (zeros above), a filter tile zero rows / columns for them.
static int cup16(int c) { return (c + 15) / 16 * 16; }
// what: 0 forward, 1 data gradient, 2 weight gradient (evae_cw_bwd_weight_plain on a descriptor with the rounded channel counts)
static int cw_plain_ok(const evae_conv_desc_t* d, int what) {
  if (!cw_geometry_ok(d) || d->KH != 3 || d->C <= 0 || d->Co <= 0) return 0;
  const int OH = d->H / d->stride;
  if (what == 0) {
    if (d->Co > 128) return 0;
    int plo, phi;
    (void)cw_taps_fwd(3, d->stride, d->pad, &plo, &phi);
    return cw_window_slots(OH, OH, plo, phi, 256) <= 576;
  }
  if (what == 1) {
    if (d->C > 128) return 0;
    for (int py = 0; py < d->stride; ++py)
      for (int px = 0; px < d->stride; ++px) {
        int plo, phi;
        (void)cw_taps_dgrad(3, d->stride, d->pad, py, px, &plo, &phi);
        if (cw_window_slots(OH, OH, plo, phi, 256) > 576) return 0;
      }
    return 1;
  }
  evae_conv_desc_t r = *d;
  r.C = cup16(d->C); r.Co = cup16(d->Co);
  return cw_wgrad_ok(&r, 0) != 0;
}
extern "C" int evae_cw_plain_supported(const evae_conv_desc_t* d, int what) { return what >= 0 && what <= 2 ? cw_plain_ok(d, what) : 0; }

extern "C" size_t evae_cw_plain_workspace_bytes(const evae_conv_desc_t* d, int what) {
  if (!d) return 256;
  if (what == 2) { evae_conv_desc_t r = *d; r.C = cup16(d->C); r.Co = cup16(d->Co); return evae_cw_workspace_bytes(&r, 7); }
  const int rows = what == 0 ? d->Co : d->C, kch = what == 0 ? d->C : d->Co;
  const int wrows = (cdiv(rows, 64) * 64 + 127) / 128 * 128;
  return (size_t)d->stride * d->stride * (p6_image_bytes(wrows, (cup16(kch) / 16) * 9) + 8192);
}

// ximg: the input's image (cup16(C) channels; rows natural for stride 1, parity-planar for stride 2).  y = conv(x, w) + b, act bit 0:
// y = ELU(y); -> out_f fp32 [N OH OW][ldo] (ldo >= Co rounded up to 8: whole 8-channel pieces are written, zeros above Co) and / or
// oimg (cup16(Co) channels... rows planar when out_planar), holding ELU(y) when act bit 1.
extern "C" int evae_cw_plain_fwd(const void* ximg, const evae_conv_desc_t* d, const float* w, const float* b, int act, float* out_f, int ldo,
                                 void* oimg, int out_planar, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(cw_plain_ok(d, 0), "cw_plain_fwd: unsupported geometry");
  EVAE_REQUIRE(ximg && w && (out_f || oimg) && ws && ws_bytes >= evae_cw_plain_workspace_bytes(d, 0), "cw_plain_fwd: null pointer / workspace too small");
  EVAE_REQUIRE(!out_f || ldo >= (d->Co + 7) / 8 * 8, "cw_plain_fwd: ldo smaller than the output channels rounded up to 8");
  const int st = d->stride, OH = d->H / st, Co = d->Co, ncg = cup16(d->C) / 16;
  EVAE_REQUIRE(!out_planar || (OH & 1) == 0, "cw_plain_fwd: parity-planar output rows need an even output grid");
  int plo, phi;
  const CwTaps tp = cw_taps_fwd(3, st, d->pad, &plo, &phi);
  const int nks_w = cw_ksteps(tp, ncg);
  const int bn = Co <= 32 ? 32 : 64, tiles_n = cdiv(Co, bn), wrows = (tiles_n * bn + 127) / 128 * 128;
  unsigned char* iw = (unsigned char*)ws;
  cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256, 0, stream>>>(w, nullptr, Co, d->C, 3, 3, tp, ncg, 2, bn, wrows, nks_w, iw);
  int rc = check_launch("cw_pack_filter_kernel");
  if (rc) return rc;
  ConvWinArgs g = {};
  g.xin = (const unsigned char*)ximg; g.nks_in = ncg; g.ncg = ncg; g.N = d->N; g.H = OH; g.W = OH; g.plo = plo; g.phi = phi; g.taps = tp;
  g.istride = st * st * OH * OH;
  for (int s2 = 0; s2 < st * st; ++s2) g.ioff[s2] = s2 * OH * OH;
  g.wimg = iw; g.nks_w = nks_w; g.Co = Co; g.tiles_n = tiles_n; g.bias0 = b; g.act = act;
  g.out_planar = out_planar;
  g.oimg = (unsigned char*)oimg; g.nks_o = cup16(Co) / 16; g.out_f = out_f; g.ldo = ldo;
  g.nsp = (cw_window_slots(OH, OH, plo, phi, 256) + 31) / 32;
  if (bn == 32) return launch_conv_win<CW_PLAIN, 4, 1, 576>(g, stream, "cw_plain_fwd");
  return launch_conv_win<CW_PLAIN, 4, 2, 576>(g, stream, "cw_plain_fwd");
}

// dyimg: the image of the gradient wrt the convolution's result (cup16(Co) channels, rows planar when dy_planar) -> the gradient wrt
// its input: dx_f fp32 natural [N H W][ldx] (ldx >= C, C % 8 == 0) and / or dximg (C % 16 == 0; rows planar for a stride-2 layer, the
// order its input image has)
extern "C" int evae_cw_plain_bwd_data(const void* dyimg, int dy_planar, const evae_conv_desc_t* d, const float* w, float* dx_f, int ldx,
                                      void* dximg, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(cw_plain_ok(d, 1), "cw_plain_bwd_data: unsupported geometry");
  EVAE_REQUIRE(dyimg && w && (dx_f || dximg) && ws && ws_bytes >= evae_cw_plain_workspace_bytes(d, 1), "cw_plain_bwd_data: null pointer / workspace too small");
  EVAE_REQUIRE(!dx_f || ldx >= (d->C + 7) / 8 * 8, "cw_plain_bwd_data: ldx smaller than the channel count rounded up to 8");
  EVAE_REQUIRE(!dximg || d->C % 16 == 0, "cw_plain_bwd_data: an image needs a multiple of 16 channels");
  const int st = d->stride, OH = d->H / st, C = d->C, Co = d->Co;
  const int ncg = cup16(Co) / 16, bn = C <= 32 ? 32 : 64, tiles_n = cdiv(C, bn), wrows = (tiles_n * bn + 127) / 128 * 128;
  size_t used = 0;
  for (int py = 0; py < st; ++py)
    for (int px = 0; px < st; ++px) {
      int plo, phi;
      const CwTaps tp = cw_taps_dgrad(3, st, d->pad, py, px, &plo, &phi);
      const int nks_w = cw_ksteps(tp, ncg);
      EVAE_REQUIRE(nks_w > 0, "cw_plain_bwd_data: a parity class without taps");
      ConvWinArgs g = {};
      g.N = d->N; g.H = OH; g.W = OH; g.plo = plo; g.phi = phi; g.taps = tp;
      g.ostride = st * st * OH * OH; g.ooff = (py * st + px) * OH * OH;
      g.nat_h = d->H; g.nat_w = d->W; g.nat_s = st; g.nat_y = py; g.nat_x = px;
      g.Co = C; g.tiles_n = tiles_n;
      g.oimg = (unsigned char*)dximg; g.nks_o = C / 16; g.out_f = dx_f; g.ldo = ldx;
      unsigned char* iw = (unsigned char*)ws + used;
      used += p6_image_bytes(wrows, nks_w) + 8192;
      cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256, 0, stream>>>(w, nullptr, Co, C, 3, 3, tp, ncg, 1, bn, wrows, nks_w, iw);
      int rc = check_launch("cw_pack_filter_kernel");
      if (rc) return rc;
      g.xin = (const unsigned char*)dyimg; g.nks_in = ncg; g.ncg = ncg; g.istride = OH * OH; g.in_planar = dy_planar;
      g.wimg = iw; g.nks_w = nks_w;
      g.nsp = (cw_window_slots(OH, OH, plo, phi, 256) + 31) / 32;
      if (bn == 32) rc = launch_conv_win<CW_PLAIN, 4, 1, 576>(g, stream, "cw_plain_bwd_data");
      else rc = launch_conv_win<CW_PLAIN, 4, 2, 576>(g, stream, "cw_plain_bwd_data");
      if (rc) return rc;
    }
  return EVAE_OK;
}

// A whole run of n <= 16 residual blocks in ONE call (the launches are the same; what goes is n - 1 trips through the caller's
// language per direction -- an eager fully_conv step is ~500 launches and waits for its host).
//   forward: block k reads aimg_k (k == 0: aimg0, else oimg[k - 1]) and x_k (x0 / out_f[k - 1]), writes out_f[k] and oimg[k]
//   (oimg[n - 1] may be NULL); fimgs: the forward filter images of evae_cw_res_pack_filters.
extern "C" int evae_cw_res_run_fwd(const evae_conv_desc_t* d, int n, const void* fimgs, const void* const* bias, const void* aimg0, const float* x0,
                                   void* const* out_f, void* const* oimg, evae_stream_t stream) {
  EVAE_REQUIRE(n >= 1 && n <= 16 && fimgs && bias && aimg0 && x0 && out_f && oimg, "cw_res_run_fwd: null pointer / 1 .. 16 blocks");
  const size_t fb = evae_cw_workspace_bytes(d, 5);
  for (int k = 0; k < n; ++k) {
    const int rc = evae_cw_res_fwd(k ? oimg[k - 1] : aimg0, d, nullptr, (const float*)bias[k], k ? (const float*)out_f[k - 1] : x0, (float*)out_f[k], oimg[k],
                                   (char*)fimgs + (size_t)k * fb, fb, stream);
    if (rc) return rc;
  }
  return EVAE_OK;
}
//   backward, blocks n - 1 .. 0: weight gradient (dw[k], db[k]; db[k] may be NULL) from the image of the gradient entering block k
//   (dyimg_top for the last one, else dximg[k + 1]) and aimgs[k] = the image of ELU(x_k); data gradient -> dx_f[k], dximg[k]
//   (dximg[0] may be NULL); bimgs: the data-gradient filter images; ws: evae_cw_workspace_bytes(d, 7) bytes.
extern "C" int evae_cw_res_run_bwd(const evae_conv_desc_t* d, int n, const void* bimgs, const void* const* aimgs, const void* dyimg_top,
                                   const float* dy_top, void* const* dx_f, void* const* dximg, void* const* dw, void* const* db, void* ws,
                                   size_t ws_bytes, evae_stream_t stream) {
  EVAE_REQUIRE(n >= 1 && n <= 16 && bimgs && aimgs && dyimg_top && dy_top && dx_f && dximg && dw && db && ws, "cw_res_run_bwd: null pointer / 1 .. 16 blocks");
  const size_t fb = evae_cw_workspace_bytes(d, 5);
  for (int k = n - 1; k >= 0; --k) {
    const void* dyi = k == n - 1 ? dyimg_top : dximg[k + 1];
    const float* dyf = k == n - 1 ? dy_top : (const float*)dx_f[k + 1];
    int rc = evae_cw_bwd_weight_plain(dyi, aimgs[k], d, (float*)dw[k], (float*)db[k], ws, ws_bytes, stream);
    if (rc) return rc;
    rc = evae_cw_res_bwd_data(dyi, d, nullptr, aimgs[k], dyf, (float*)dx_f[k], dximg[k], (char*)bimgs + (size_t)k * fb, fb, stream);
    if (rc) return rc;
  }
  return EVAE_OK;
}

// First layer of a stack (C == 1): x fp32 [N][H][W] -> output image (rows planar when out_planar) + gate (+ fp32 copy); exact fp32
// arithmetic (fp32 matrix instruction).  what: 3 = forward, 4 = weight gradient in evae_cw_supported.
extern "C" int evae_cw_first_fwd(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh, const float* wg, const float* bg,
                                 void* oimg, int out_planar, float* out_s, float* out_f, evae_stream_t stream_) {
  EVAE_REQUIRE(cw_first_ok(d, 0), "cw_first_fwd: unsupported geometry");
  EVAE_REQUIRE(x && wh && wg && (oimg || out_f), "cw_first_fwd: null pointer");
  EVAE_REQUIRE(!out_planar || ((d->H | d->W) & 1) == 0, "cw_first_fwd: parity-planar output rows need an even grid");
  ConvFirstArgs a = {};
  ConvWinArgs& g = a.e;
  g.N = d->N; g.H = d->H; g.W = d->W; g.plo = g.phi = d->pad; g.Co = d->Co; g.tiles_n = d->Co / 32; g.bias0 = bh; g.bias1 = bg;
  g.out_planar = out_planar; g.oimg = (unsigned char*)oimg; g.nks_o = d->Co / 16; g.out_s = out_s; g.out_f = out_f; g.ldo = d->Co;
  a.x = x; a.w0 = wh; a.w1 = wg; a.K = d->KH;
  return launch_conv_first<CW_FIRST_SLOTS>(a, (hipStream_t)stream_, "cw_first_fwd");
}

extern "C" size_t evae_cw_first_workspace_bytes(void) { return (size_t)CW_FIRST_WGRAD_BLOCKS * 4096 * sizeof(float) + 256; }

// dy: merged fp32 gradient [N H W][2 Co] (what evae_cw_bwd_data_gate writes as out_f); dw [2 Co][K K], db [2 Co]
extern "C" int evae_cw_first_bwd_weight(const float* dy, const float* x, const evae_conv_desc_t* d, float* dw, float* db, void* ws,
                                        size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(cw_first_ok(d, 1), "cw_first_bwd_weight: unsupported geometry");
  EVAE_REQUIRE(dy && x && dw && ws && ws_bytes >= evae_cw_first_workspace_bytes(), "cw_first_bwd_weight: null pointer / workspace too small");
  ConvFirstWgradArgs a = {};
  ConvWinArgs& g = a.e;
  g.N = d->N; g.H = d->H; g.W = d->W; g.plo = g.phi = d->pad;
  g.PW = g.W + 2 * d->pad; g.SP = (g.H + 2 * d->pad) * g.PW;
  g.div_w = make_fastdiv((unsigned)g.W); g.div_hw = make_fastdiv((unsigned)(g.H * g.W));
  g.div_pw = make_fastdiv((unsigned)g.PW); g.div_sp = make_fastdiv((unsigned)g.SP);
  g.M = g.N * g.H * g.W;
  a.x = x; a.dy = dy; a.CC = 2 * d->Co; a.K = d->KH; a.part = (float*)ws;
  a.nstage = cdiv(g.M, 256);
  a.cper = cdiv(a.nstage, CW_FIRST_WGRAD_BLOCKS);
  const int nblk = cdiv(a.nstage, a.cper);
  conv_first_wgrad_kernel<CW_FIRST_SLOTS><<<nblk, 256, 0, stream>>>(a);
  int rc = check_launch("conv_first_wgrad_kernel");
  if (rc) return rc;
  cw_first_wgrad_finish_kernel<<<(a.CC * 64 + 255) / 256, 256, 0, stream>>>(a.part, nblk, a.CC, d->KH * d->KW, dw, db);
  return check_launch("cw_first_wgrad_finish_kernel");
}
