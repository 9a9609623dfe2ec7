// Dense-path helper kernels that are HBM-bound rather than MFMA-bound (gfx950 only): the depthwise half of the
// separable 5x5 convolutions of the decoder and the heads.  The GEMM-shaped convolutions stay with MIOpen.
#include "emp_common.h"

#ifndef EMP_DW_TX5
#define EMP_DW_TX5 4      // pixels per thread of the 5x5 depthwise kernel
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

struct f4 {
    v2f lo, hi;
};

__device__ __forceinline__ f4 ld4(const float4 *p)
{
    float4 v = *p;
    f4 r;
    r.lo = (v2f){v.x, v.y};
    r.hi = (v2f){v.z, v.w};
    return r;
}

__device__ __forceinline__ f4 zero4()
{
    f4 r;
    r.lo = (v2f){0.f, 0.f};
    r.hi = (v2f){0.f, 0.f};
    return r;
}

// ------------------------------------------------------------------------------------------
// D2: depthwise K x K convolution, NHWC fp32, stride 1, zero "same" padding.
// A row of an NHWC image is W*C contiguous floats.  A thread owns 4 channels (one float4) of TX adjacent pixels;
// lanes run over the channel groups first, so every load of a wave is a contiguous segment (1 KiB when C >= 256).
// Per input row a thread loads the TX + K - 1 pixels its outputs touch (instead of K per output: the horizontal
// re-read through L2 drops from K x to (TX + K - 1) / TX x -- with TX = 1 the kernel was L2-bandwidth bound at
// 3.2 TB/s of algorithmic traffic).  Vertically a thread marches down RY output rows keeping K partial output
// rows per pixel in registers, so each input row is loaded once per thread.  The K*K weights of the thread's 4
// channels stay in registers.  Output element = bias + fma chain over the taps in raster order (i, then j) from +0.
// HBM traffic: 4 B read + 4 B written per element (+ halo rows / columns, L2 hits thanks to the XCD-aware tile order).
template <int K, int TX>
__global__ __launch_bounds__(256) void dwconv_nhwc_kernel(const float4 *__restrict__ x, const float4 *__restrict__ w,
                                                          const float4 *__restrict__ bias, int H, int W, int C4,
                                                          int RY, int gx, int gy, int total, float4 *__restrict__ y)
{
    constexpr int P = K / 2;
    constexpr int NL = TX + K - 1;                     // pixels loaded per input row
    // Consecutive hardware block ids go to different XCDs (own L2 each).  Blocks that are neighbours along a row
    // share halo columns, neighbours in y share K-1 halo rows: renumber so that every XCD works on a contiguous
    // range of (n, y-strip, x-block) tiles, x fastest, and the halos are L2 hits instead of second HBM fetches
    // (PMC: 2.2x the input bytes fetched without this).
    const int chunk = (total + 7) >> 3;
    const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile >= total) return;
    const int bx = tile % gx, by = (tile / gx) % gy;
    const int t = bx * 256 + threadIdx.x;
    const int wg = (W + TX - 1) / TX;                  // pixel groups per row
    if (t >= wg * C4) return;
    const int c4 = t % C4, x0 = (t / C4) * TX;
    const int n = tile / (gx * gy), y0 = by * RY;
    const int y1 = min(y0 + RY, H);
    const int rowlen = W * C4;

    f4 wt[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) wt[i][j] = ld4(w + (i * K + j) * C4 + c4);
    bool xin[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) xin[j] = (x0 + j - P >= 0) && (x0 + j - P < W);
    f4 b = zero4();
    if (bias) b = ld4(bias + c4);

    f4 acc[K][TX];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int q = 0; q < TX; ++q) acc[s][q] = zero4();

    const float4 *xin_n = x + (int64_t)n * H * rowlen + (int64_t)x0 * C4 + c4;
    float4 *yout_n = y + (int64_t)n * H * rowlen + (int64_t)x0 * C4 + c4;
    const int rows = (y1 - y0) + 2 * P;
    for (int rr = 0; rr < rows; rr += K) {
#pragma unroll
        for (int ph = 0; ph < K; ++ph) {
            const int r = y0 - P + rr + ph;                 // input row
            if (rr + ph < rows) {
                f4 in[NL];
                const bool rin = (r >= 0) && (r < H);
                const float4 *xr = xin_n + (int64_t)r * rowlen;
#pragma unroll
                for (int j = 0; j < NL; ++j) in[j] = (rin && xin[j]) ? ld4(xr + (j - P) * C4) : zero4();
                // input row r is tap row i of output row r + P - i, which lives in slot (ph - i) mod K
#pragma unroll
                for (int i = 0; i < K; ++i)
#pragma unroll
                    for (int q = 0; q < TX; ++q) {
                        f4 &a = acc[(ph - i + K) % K][q];
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            a.lo = __builtin_elementwise_fma(in[q + j].lo, wt[i][j].lo, a.lo);
                            a.hi = __builtin_elementwise_fma(in[q + j].hi, wt[i][j].hi, a.hi);
                        }
                    }
                // output row r - P has now seen its last tap row (i = K - 1)
                const int o = r - P;
#pragma unroll
                for (int q = 0; q < TX; ++q) {
                    f4 &done = acc[(ph + 1) % K][q];
                    if (o >= y0 && o < y1 && x0 + q < W) {
                        float4 v;
                        v.x = done.lo.x + b.lo.x; v.y = done.lo.y + b.lo.y;
                        v.z = done.hi.x + b.hi.x; v.w = done.hi.y + b.hi.y;
                        yout_n[(int64_t)o * rowlen + q * C4] = v;
                    }
                    done = zero4();
                }
            }
        }
    }
}

extern "C" int emp_dwconv_nhwc(const float *x, const float *w_kkc, const float *bias, int N, int H, int W, int C,
                               int k, float *y, void *stream)
{
    EMP_REQUIRE(x && w_kkc && y, "dwconv: null pointer");
    EMP_REQUIRE(x != y, "dwconv: in-place operation is not supported");
    EMP_REQUIRE(k == 3 || k == 5, "dwconv: kernel size %d not in {3, 5}", k);
    EMP_REQUIRE(C > 0 && C % 4 == 0, "dwconv: channel count %d must be a multiple of 4", C);
    EMP_REQUIRE(N >= 0 && N <= 65535 && H > 0 && W > 0, "dwconv: bad shape");
    EMP_REQUIRE((int64_t)W * (C / 4) < (1LL << 30), "dwconv: row too long");
    EMP_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_kkc) |
                  reinterpret_cast<uintptr_t>(bias)) & 15) == 0, "dwconv: pointers must be 16-byte aligned");
    if (N == 0) return EMP_OK;
    const int C4 = C / 4;
    const int TX = k == 5 ? EMP_DW_TX5 : 1;
    const int gx = (int)emp_cdiv(emp_cdiv(W, TX) * C4, 256);
    // rows per block: long strips amortise the K-1 halo rows; short ones keep >= ~2k blocks in flight
    int RY = 32;
    while (RY > 8 && (int64_t)gx * emp_cdiv(H, RY) * N < 2048) RY >>= 1;
    const int gy = (int)emp_cdiv(H, RY);
    EMP_REQUIRE(gy <= 65535, "dwconv: image too tall");
    hipStream_t st = emp_stream(stream);
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *w4 = reinterpret_cast<const float4 *>(w_kkc);
    const float4 *b4 = reinterpret_cast<const float4 *>(bias);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    EMP_REQUIRE((int64_t)gx * gy * N < (1LL << 30), "dwconv: too many blocks");
    const int total = gx * gy * N;
    dim3 grid(8 * ((total + 7) / 8));
    if (k == 3) hipLaunchKernelGGL((dwconv_nhwc_kernel<3, 1>), grid, dim3(256), 0, st, x4, w4, b4, H, W, C4, RY, gx, gy, total, y4);
    else hipLaunchKernelGGL((dwconv_nhwc_kernel<5, EMP_DW_TX5>), grid, dim3(256), 0, st, x4, w4, b4, H, W, C4, RY, gx, gy, total, y4);
    EMP_CHECK_LAUNCH("emp_dwconv_nhwc");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// D3: bilinear up-sampling, align_corners = True (the decoder's F.interpolate calls and the x4 up-sampling of the
// heads).  Source coordinate = dst * (in - 1) / (out - 1) in fp32; value =
//   (1-ly) * ((1-lx) * v00 + lx * v01) + ly * ((1-lx) * v10 + lx * v11), unfused fp32.
// Generic element strides on both sides: the destination may be a channel slice of a wider NHWC buffer (which
// is how the decoder's torch.cat disappears), the source may be NCHW or NHWC.
struct UpGeom {
    int N, C, h, w, H, W;
    int64_t xs_n, xs_c, xs_h, xs_w, ys_n, ys_c, ys_h, ys_w;
    float ry, rx;
};

__device__ __forceinline__ void up_src(float r, int dst, int in, int &i0, int &i1, float &l1)
{
    const float s = __fmul_rn(r, (float)dst);
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = __fsub_rn(s, (float)i0);
}

__device__ __forceinline__ float up_mix(float v00, float v01, float v10, float v11, float lx, float ly)
{
    const float lx0 = __fsub_rn(1.0f, lx), ly0 = __fsub_rn(1.0f, ly);
    const float top = __fadd_rn(__fmul_rn(lx0, v00), __fmul_rn(lx, v01));
    const float bot = __fadd_rn(__fmul_rn(lx0, v10), __fmul_rn(lx, v11));
    return __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly, bot));
}

// channels innermost on both sides (xs_c == ys_c == 1, C % 4 == 0).  A lane owns one float4 of channels and walks UP_SEG
// consecutive output columns of one output row: the source column advances by 0 or 1 per step, so the two source texels of
// a row are kept in registers and only a NEW column is fetched (2x up-sampling: ~9 fetches of two rows for 16 outputs,
// where the one-output-per-thread form made 64 -- 8.6 GB through L1 / L2 for a 2.7 GB call, and four 64-bit divisions per
// output).  Lanes are adjacent in the channel index: every load and store of a wave is one contiguous segment.
#define UP_SEG 16
__global__ __launch_bounds__(256) void upsample_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, UpGeom g,
                                                            uint32_t nseg, uint32_t total)
{
    const uint32_t C4 = (uint32_t)g.C >> 2;
    for (uint32_t item = blockIdx.x * 256u + threadIdx.x; item < total; item += gridDim.x * 256u) {
        const uint32_t c4 = item % C4, t = item / C4;
        const uint32_t seg = t % nseg, row = t / nseg;
        const int n = (int)(row / (uint32_t)g.H), Y = (int)(row % (uint32_t)g.H);
        int y0, y1;
        float ly;
        up_src(g.ry, Y, g.h, y0, y1, ly);
        const float *b0 = x + n * g.xs_n + y0 * g.xs_h + 4 * c4;
        const float *b1 = x + n * g.xs_n + y1 * g.xs_h + 4 * c4;
        float *out = y + n * g.ys_n + Y * g.ys_h + 4 * c4;
        const int X0 = (int)(seg * UP_SEG), Xe = min(X0 + UP_SEG, g.W);
        int cx0 = -1, cx1 = -1;                                     // source columns held in (a0, c0) and (a1, c1)
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, c0 = a0, c1 = a0;
        for (int X = X0; X < Xe; ++X) {
            int x0, x1;
            float lx;
            up_src(g.rx, X, g.w, x0, x1, lx);
            if (x0 != cx0) {
                if (x0 == cx1) { a0 = a1; c0 = c1; }
                else {
                    a0 = *reinterpret_cast<const float4 *>(b0 + x0 * g.xs_w);
                    c0 = *reinterpret_cast<const float4 *>(b1 + x0 * g.xs_w);
                }
                cx0 = x0;
            }
            if (x1 != cx1) {
                if (x1 == cx0) { a1 = a0; c1 = c0; }
                else {
                    a1 = *reinterpret_cast<const float4 *>(b0 + x1 * g.xs_w);
                    c1 = *reinterpret_cast<const float4 *>(b1 + x1 * g.xs_w);
                }
                cx1 = x1;
            }
            float4 o;
            o.x = up_mix(a0.x, a1.x, c0.x, c1.x, lx, ly);
            o.y = up_mix(a0.y, a1.y, c0.y, c1.y, lx, ly);
            o.z = up_mix(a0.z, a1.z, c0.z, c1.z, lx, ly);
            o.w = up_mix(a0.w, a1.w, c0.w, c1.w, lx, ly);
            *reinterpret_cast<float4 *>(out + X * g.ys_w) = o;
        }
    }
}

// any strides: one thread per 4 consecutive output columns of one (n, c, Y) row
__global__ __launch_bounds__(256) void upsample_planar_kernel(const float *__restrict__ x, float *__restrict__ y, UpGeom g)
{
    const int W4 = (g.W + 3) >> 2;
    const int64_t total = (int64_t)g.N * g.C * g.H * W4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X0 = (int)(i % W4) * 4;
        int64_t p = i / W4;
        const int Y = (int)(p % g.H);
        p /= g.H;
        const int c = (int)(p % g.C);
        const int n = (int)(p / g.C);
        int y0, y1;
        float ly;
        up_src(g.ry, Y, g.h, y0, y1, ly);
        const float *r0 = x + n * g.xs_n + c * g.xs_c + y0 * g.xs_h;
        const float *r1 = x + n * g.xs_n + c * g.xs_c + y1 * g.xs_h;
        float *out = y + n * g.ys_n + c * g.ys_c + Y * g.ys_h;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int X = X0 + j;
            o[j] = 0.f;
            if (X < g.W) {
                int x0, x1;
                float lx;
                up_src(g.rx, X, g.w, x0, x1, lx);
                o[j] = up_mix(r0[x0 * g.xs_w], r0[x1 * g.xs_w], r1[x0 * g.xs_w], r1[x1 * g.xs_w], lx, ly);
            }
        }
        if (g.ys_w == 1 && X0 + 3 < g.W && ((reinterpret_cast<uintptr_t>(out + X0) & 15) == 0)) {
            *reinterpret_cast<float4 *>(out + X0) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (X0 + j < g.W) out[(X0 + j) * g.ys_w] = o[j];
        }
    }
}

extern "C" int emp_upsample_bilinear(const float *x, int N, int C, int h, int w, const int64_t *x_strides,
                                     float *y, int H, int W, const int64_t *y_strides, void *stream)
{
    EMP_REQUIRE(x && y && x_strides && y_strides, "upsample: null pointer");
    EMP_REQUIRE(N >= 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample: bad shape");
    EMP_REQUIRE(h < (1 << 24) && w < (1 << 24) && H < (1 << 24) && W < (1 << 24), "upsample: sizes exceed fp32 integers");
    if (N == 0) return EMP_OK;
    UpGeom g;
    g.N = N; g.C = C; g.h = h; g.w = w; g.H = H; g.W = W;
    g.xs_n = x_strides[0]; g.xs_c = x_strides[1]; g.xs_h = x_strides[2]; g.xs_w = x_strides[3];
    g.ys_n = y_strides[0]; g.ys_c = y_strides[1]; g.ys_h = y_strides[2]; g.ys_w = y_strides[3];
    g.ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    g.rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipStream_t st = emp_stream(stream);
    const bool vec = g.xs_c == 1 && g.ys_c == 1 && C % 4 == 0 &&
                     ((g.xs_n | g.xs_h | g.xs_w | g.ys_n | g.ys_h | g.ys_w) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (vec) {
        const int64_t nseg = emp_cdiv(W, UP_SEG), total = (int64_t)N * H * nseg * (C / 4);
        EMP_REQUIRE(total < (1LL << 31), "upsample: too many items for one launch");
        hipLaunchKernelGGL(upsample_nhwc_kernel, dim3(emp_grid(total, 256, 65536)), dim3(256), 0, st, x, y, g,
                           (uint32_t)nseg, (uint32_t)total);
    } else {
        const int64_t total = (int64_t)N * C * H * ((W + 3) / 4);
        hipLaunchKernelGGL(upsample_planar_kernel, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, st, x, y, g);
    }
    EMP_CHECK_LAUNCH("emp_upsample_bilinear");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// D0: slice feeder.  A resident uint8 volume is cut along any axis by element strides (xy slices are contiguous,
// xz / yz slices are strided views: no host transposes, no extra copies), normalised like the reference's
// albumentations Normalize (fp32: (x - mean*255) * (1 / (std*255))) and zero-padded to (hp, wp) in one pass.
__global__ __launch_bounds__(256) void slices_to_input_kernel(const uint8_t *__restrict__ vol, int64_t s_slice,
                                                             int64_t s_row, int64_t s_col, int n, int h, int w, int hp,
                                                             int wp, float mean255, float inv_std255,
                                                             float *__restrict__ out)
{
    const int wq = (wp + 3) >> 2;
    const int64_t total = (int64_t)n * hp * wq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % wq) * 4;
        const int64_t q = i / wq;
        const int r = (int)(q % hp);
        const int64_t s = q / hp;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < h) {
            const uint8_t *src = vol + s * s_slice + (int64_t)r * s_row;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < w) v[j] = __fmul_rn(__fsub_rn((float)src[(int64_t)(c0 + j) * s_col], mean255), inv_std255);
        }
        float *dst = out + (s * hp + r) * (int64_t)wp + c0;
        if (c0 + 3 < wp && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < wp) dst[j] = v[j];
        }
    }
}

extern "C" int emp_slices_to_input(const uint8_t *vol, int64_t stride_slice, int64_t stride_row, int64_t stride_col,
                                   int n_slices, int h, int w, int hp, int wp, float mean255, float inv_std255,
                                   float *out, void *stream)
{
    EMP_REQUIRE(vol && out, "slices_to_input: null pointer");
    EMP_REQUIRE(n_slices >= 0 && h > 0 && w > 0 && hp >= h && wp >= w, "slices_to_input: bad shape");
    if (n_slices == 0) return EMP_OK;
    const int64_t total = (int64_t)n_slices * hp * ((wp + 3) / 4);
    hipLaunchKernelGGL(slices_to_input_kernel, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, emp_stream(stream), vol,
                       stride_slice, stride_row, stride_col, n_slices, h, w, hp, wp, mean255, inv_std255, out);
    EMP_CHECK_LAUNCH("emp_slices_to_input");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// D6: 1x1 convolution to a few output channels (the last layer of every head: 256 -> 1 or 2, bias).  Pure streaming:
// one wave per pixel, lane l holds the channel groups l, l + 64, ... (float4 each) and the matching weights in
// registers; per pixel and output channel an fma chain per lane, then a butterfly sum over the wave
// (xor 32, 16, 8, 4, 2, 1 -- every lane ends with the same value).  64 pixels per wave iteration so that the
// results are written as one coalesced row.  Output is planar (N, Cout, H*W): what the up-sampling and the
// post-processing kernels read.
template <int CO, int G>   // CO output channels, G float4 groups per lane
__global__ __launch_bounds__(256) void pointwise_out_kernel(const float4 *__restrict__ x, const float4 *__restrict__ w,
                                                            const float *__restrict__ bias, int64_t P, int64_t HW,
                                                            int C4, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float4 wt[CO][G];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c4 = lane + 64 * g;
            wt[co][g] = c4 < C4 ? w[co * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float b[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) b[co] = bias ? bias[co] : 0.f;
    for (int64_t p0 = wave * 64; p0 < P; p0 += n_waves * 64) {
        float res[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) res[co] = 0.f;
        const int np = (int)min((int64_t)64, P - p0);
        for (int j0 = 0; j0 < np; j0 += 4) {
            float4 v[4][G];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int c4 = lane + 64 * g;
                    v[q][g] = (j0 + q < np && c4 < C4) ? x[(p0 + j0 + q) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    float s = 0.f;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        s = __fmaf_rn(v[q][g].x, wt[co][g].x, s);
                        s = __fmaf_rn(v[q][g].y, wt[co][g].y, s);
                        s = __fmaf_rn(v[q][g].z, wt[co][g].z, s);
                        s = __fmaf_rn(v[q][g].w, wt[co][g].w, s);
                    }
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) s = __fadd_rn(s, __shfl_xor(s, o));
                    if (lane == j0 + q) res[co] = __fadd_rn(s, b[co]);
                }
        }
        if (lane < np) {
            const int64_t p = p0 + lane;
            const int64_t n = p / HW, r = p - n * HW;
#pragma unroll
            for (int co = 0; co < CO; ++co) out[(n * CO + co) * HW + r] = res[co];
        }
    }
}

extern "C" int emp_pointwise_out_nhwc(const float *x, const float *w, const float *bias, int64_t n_pixels,
                                      int64_t pixels_per_image, int C, int Cout, float *out, void *stream)
{
    EMP_REQUIRE(x && w && out, "pointwise_out: null pointer");
    EMP_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024, "pointwise_out: C %d must be a multiple of 4, at most 1024", C);
    EMP_REQUIRE(Cout >= 1 && Cout <= 4, "pointwise_out: Cout %d not in 1..4", Cout);
    EMP_REQUIRE(n_pixels >= 0 && pixels_per_image > 0 && n_pixels % pixels_per_image == 0, "pointwise_out: bad sizes");
    EMP_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0, "pointwise_out: alignment");
    if (n_pixels == 0) return EMP_OK;
    const int C4 = C / 4, G = (C4 + 63) / 64;
    const int grid = emp_grid(emp_cdiv(n_pixels, 64) * 64, 256, 8192);
    hipStream_t st = emp_stream(stream);
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *w4 = reinterpret_cast<const float4 *>(w);
#define EMP_PW(CO, GG) hipLaunchKernelGGL((pointwise_out_kernel<CO, GG>), dim3(grid), dim3(256), 0, st, x4, w4, bias, n_pixels, pixels_per_image, C4, out)
#define EMP_PW_G(CO) do { if (G == 1) EMP_PW(CO, 1); else if (G == 2) EMP_PW(CO, 2); else if (G == 3) EMP_PW(CO, 3); else EMP_PW(CO, 4); } while (0)
    if (Cout == 1) EMP_PW_G(1);
    else if (Cout == 2) EMP_PW_G(2);
    else if (Cout == 3) EMP_PW_G(3);
    else EMP_PW_G(4);
#undef EMP_PW_G
#undef EMP_PW
    EMP_CHECK_LAUNCH("emp_pointwise_out_nhwc");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// D7: stem epilogue: eval-BatchNorm + ReLU + MaxPool2d(3, stride 2, padding 1) in one pass over the NHWC output of
// the first convolution (4x more pixels than anything downstream): reads it once, writes the pooled quarter.
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float4 *__restrict__ x, const float4 *__restrict__ scale,
                                                              const float4 *__restrict__ shift, int N, int H, int W,
                                                              int C4, int OH, int OW, float4 *__restrict__ y)
{
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int ox = (int)(p % OW);
        p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        const float4 a = scale[c4], b = shift[c4];
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = 2 * oy + dy, xx = 2 * ox + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float4 v = x[(((int64_t)n * H + yy) * W + xx) * C4 + c4];
                    m.x = fmaxf(m.x, fmaxf(__fadd_rn(__fmul_rn(v.x, a.x), b.x), 0.f));
                    m.y = fmaxf(m.y, fmaxf(__fadd_rn(__fmul_rn(v.y, a.y), b.y), 0.f));
                    m.z = fmaxf(m.z, fmaxf(__fadd_rn(__fmul_rn(v.z, a.z), b.z), 0.f));
                    m.w = fmaxf(m.w, fmaxf(__fadd_rn(__fmul_rn(v.w, a.w), b.w), 0.f));
                }
            }
        y[i] = m;
    }
}

extern "C" int emp_bn_relu_maxpool_nhwc(const float *x, const float *scale, const float *shift, int N, int H, int W,
                                        int C, float *y, void *stream)
{
    EMP_REQUIRE(x && scale && shift && y, "bn_relu_maxpool: null pointer");
    EMP_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bn_relu_maxpool: bad shape");
    EMP_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) |
                  reinterpret_cast<uintptr_t>(shift)) & 15) == 0, "bn_relu_maxpool: alignment");
    if (N == 0) return EMP_OK;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, emp_stream(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(scale),
                       reinterpret_cast<const float4 *>(shift), N, H, W, C / 4, OH, OW, reinterpret_cast<float4 *>(y));
    EMP_CHECK_LAUNCH("emp_bn_relu_maxpool_nhwc");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// D2: logits_to_prob (engines.py:22-30): sigmoid for one channel, softmax over the channel axis otherwise, planar
// (N, C, HW) fp32.  One streaming pass; every lane owns 4 consecutive pixels (float4 per channel plane).
// y may be x itself (in place): no __restrict__ here -- a lane reads every channel of its pixels in the first two
// loops and, in the third, each channel again right before it stores that same address.
template <int VEC>
__global__ __launch_bounds__(256) void logits_to_prob_kernel(const float *x, int C, int64_t HW, int64_t total, float *y)
{
    // total = N * HW / VEC work items; item i covers pixels [VEC * (i % (HW / VEC)), +VEC) of image i / (HW / VEC)
    const int64_t per = HW / VEC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / per, p = (i - n * per) * VEC;
        const float *xp = x + n * C * HW + p;
        float *yp = y + n * C * HW + p;
        if (C == 1) {
            float v[VEC];
            if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4 *>(xp); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
            else v[0] = xp[0];
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v[e])));
            if constexpr (VEC == 4) *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            else yp[0] = v[0];
        } else {
            float m[VEC], sum[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { m[e] = -INFINITY; sum[e] = 0.f; }
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], xp[c * HW + e]);
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e) sum[e] = __fadd_rn(sum[e], expf(__fsub_rn(xp[c * HW + e], m[e])));
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e) yp[c * HW + e] = __fdiv_rn(expf(__fsub_rn(xp[c * HW + e], m[e])), sum[e]);
        }
    }
}

// softmax over 2..L2P_CMAX channels, 4 pixels per lane: every channel's float4 is loaded ONCE into registers (the form
// above reads each logit three times with 4-byte loads and evaluates every exponential twice: 1.0 TB/s on C = 5);
// max, exp(x - max) (kept), ascending sum, divide -- the same operations in the same order, so the same bits.
#define L2P_CMAX 8
__global__ __launch_bounds__(256) void softmax4_kernel(const float *x, int C, int64_t HW, int64_t total, float *y)
{
    const int64_t per = HW / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / per, p = (i - n * per) * 4;
        const float *xp = x + n * C * HW + p;
        float *yp = y + n * C * HW + p;
        float v[L2P_CMAX][4];
#pragma unroll
        for (int c = 0; c < L2P_CMAX; ++c)
            if (c < C) {
                const float4 t = *reinterpret_cast<const float4 *>(xp + c * HW);
                v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
            }
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < L2P_CMAX; ++c)
            if (c < C)
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[c][e]);
#pragma unroll
        for (int c = 0; c < L2P_CMAX; ++c)
            if (c < C)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[c][e] = expf(__fsub_rn(v[c][e], m[e]));
                    sum[e] = __fadd_rn(sum[e], v[c][e]);
                }
#pragma unroll
        for (int c = 0; c < L2P_CMAX; ++c)
            if (c < C)
                *reinterpret_cast<float4 *>(yp + c * HW) =
                    make_float4(__fdiv_rn(v[c][0], sum[0]), __fdiv_rn(v[c][1], sum[1]), __fdiv_rn(v[c][2], sum[2]),
                                __fdiv_rn(v[c][3], sum[3]));
    }
}

extern "C" int emp_logits_to_prob(const float *logits, int N, int C, int64_t HW, float *prob, void *stream)
{
    EMP_REQUIRE(logits && prob, "logits_to_prob: null pointer");
    EMP_REQUIRE(N >= 0 && C >= 1 && C <= 64 && HW > 0, "logits_to_prob: bad shape");
    if (N == 0) return EMP_OK;
    const bool vec = HW % 4 == 0 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(prob)) & 15) == 0;
    if (vec && C == 1) {
        const int64_t total = (int64_t)N * (HW / 4);
        hipLaunchKernelGGL(logits_to_prob_kernel<4>, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, emp_stream(stream),
                           logits, C, HW, total, prob);
    } else if (vec && C <= L2P_CMAX) {
        const int64_t total = (int64_t)N * (HW / 4);
        hipLaunchKernelGGL(softmax4_kernel, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, emp_stream(stream), logits, C,
                           HW, total, prob);
    } else {
        const int64_t total = (int64_t)N * HW;
        hipLaunchKernelGGL(logits_to_prob_kernel<1>, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, emp_stream(stream),
                           logits, C, HW, total, prob);
    }
    EMP_CHECK_LAUNCH("emp_logits_to_prob");
    return EMP_OK;
}
