// D9: the ResNet stem in one kernel: Conv2d(1, 64, 7, stride 2, padding 3) -> BatchNorm(eval) -> ReLU ->
// MaxPool2d(3, stride 2, padding 1) on a one-channel fp32 image, output NHWC at 1/4 resolution.  gfx950 only.
//
//   The convolution has K = 49 (one input channel): far too short for the matrix cores' tiles to pay for an im2col
//   through LDS, and its 64-channel output at 1/2 resolution (16 B per input byte) is only needed by the pooling that
//   follows.  So it runs on the vector ALUs (the fp32 vector rate equals the fp32 matrix rate on this chip) and the
//   half-resolution activation never leaves the CU:
//     block = pooled tile of 4 x 16 pixels; it needs the 9 x 33 convolution outputs around it and a 23 x 71 input patch;
//     lane = one convolution pixel: it reads its 7 x 7 window from the patch one filter row at a time, the filter values are wave-uniform and
//     arrive through the scalar cache (SGPR operands), the 64 channels accumulate in registers as fma chains over the
//     taps in raster order from +0; BN and ReLU are applied (separate roundings, as in D7) and 32 channels at a time go to
//     an LDS tile [channel][conv pixel] (row length 297 is odd: conflict-free for the pooling reads);
//     then 64 x 32 pooled values are maxima over 3 x 3 conv pixels (outside the image: skipped) and leave as
//     128-byte rows of the NHWC output.
//   Measured alternatives (same outputs, bit for bit): the convolution as an MFMA GEMM with the im2col gathered per lane
//   from the patch (M = conv pixels, K = 49 + 1 taps, filter table in registers or LDS; 5-wave blocks, 4-wave blocks of
//   3 x 16 pooled pixels, persistent blocks): 1.38-1.89 ms per call at the bench's shape against 1.21 ms for this
//   version -- with K = 50 a tile is 6 400 matrix cycles and the patch / pooling phases around it dominate.
#include "emp_common.h"

#define ST_THREADS 320
#define ST_PH 4
#define ST_PW 16
#define ST_CH (2 * ST_PH + 1)      // 9 conv rows
#define ST_CW (2 * ST_PW + 1)      // 33 conv cols
#define ST_NPX (ST_CH * ST_CW)     // 297 conv pixels
#define ST_IH (2 * ST_CH + 5)      // 23 input rows
#define ST_IW (2 * ST_CW + 5)      // 71 input cols
#define ST_IS 72                   // patch row stride

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(ST_THREADS, 3) void stem7_kernel(const float *__restrict__ x, const float *__restrict__ w_tc,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             int N, int H, int W, int OH, int OW, int PH, int PW,
                                                             int tiles_y, int tiles_x, float *__restrict__ y)
{
    __shared__ float patch[ST_IH * ST_IS];
    __shared__ float tile[32 * ST_NPX];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int py0 = ty * ST_PH, px0 = tx * ST_PW;         // pooled origin
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;       // conv origin (may be -1)
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;       // input origin

    const float *xin = x + (int64_t)n * H * W;
    for (int i = tid; i < ST_IH * ST_IW; i += ST_THREADS) {
        const int r = i / ST_IW, c = i - r * ST_IW;
        const int yy = iy0 + r, xx = ix0 + c;
        patch[r * ST_IS + c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xin[(int64_t)yy * W + xx] : 0.f;
    }
    __syncthreads();

    const bool own = tid < ST_NPX;
    const int lcy = own ? tid / ST_CW : 0, lcx = own ? tid - (tid / ST_CW) * ST_CW : 0;
    const int gcy = cy0 + lcy, gcx = cx0 + lcx;
    const bool cvalid = own && gcy >= 0 && gcy < OH && gcx >= 0 && gcx < OW;
    const float *pwin = &patch[(2 * lcy) * ST_IS + 2 * lcx];    // the lane's 7 x 7 window, re-read per filter row

    // All 64 channels of the lane's pixel accumulate in registers (32 packed pairs).  One tap per iteration, NOT
    // unrolled: a tap's 64 filter values are four s_load_dwordx16; unrolled, the scheduler hoists every load of the
    // loop and spills 1 500 SGPRs.  The scalar-load latency hides behind the other waves of the SIMD.
    f32x2 acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = (f32x2){0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll 1
        for (int kx = 0; kx < 7; ++kx) {
            const float xk = pwin[ky * ST_IS + kx];
            const f32x2 xs = (f32x2){xk, xk};
            const float *wt = w_tc + (ky * 7 + kx) * 64;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const f32x2 wv = (f32x2){wt[2 * c], wt[2 * c + 1]};                        // wave-uniform: scalar loads
                acc[c] = __builtin_elementwise_fma(xs, wv, acc[c]);
            }
        }
    }
    // BN + ReLU, then pooling through the LDS tile, 32 channels at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half > 0) __syncthreads();                      // the previous half's pooling reads are done
        if (own) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ch = half * 32 + 2 * c + e;
                    const float v = fmaxf(__fadd_rn(__fmul_rn(acc[half * 16 + c][e], scale[ch]), shift[ch]), 0.f);
                    tile[(2 * c + e) * ST_NPX + tid] = cvalid ? v : -INFINITY;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < ST_PH * ST_PW * 32; i += ST_THREADS) {
            const int c = i & 31, pp = i >> 5;
            const int ly = pp / ST_PW, lx = pp - ly * ST_PW;
            const int py = py0 + ly, px = px0 + lx;
            if (py >= PH || px >= PW) continue;
            const float *tc = &tile[c * ST_NPX + (2 * ly) * ST_CW + 2 * lx];
            float m = -INFINITY;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, tc[dy * ST_CW + dx]);
            y[(((int64_t)n * PH + py) * PW + px) * 64 + half * 32 + c] = m;
        }
    }
}

extern "C" int emp_stem_conv7_bn_relu_maxpool(const float *x, const float *w_tc, const float *scale, const float *shift,
                                              int N, int H, int W, float *y, void *stream)
{
    EMP_REQUIRE(x && w_tc && scale && shift && y, "stem: null pointer");
    EMP_REQUIRE(N >= 0 && H > 0 && W > 0, "stem: bad shape");
    EMP_REQUIRE((reinterpret_cast<uintptr_t>(w_tc) & 15) == 0, "stem: alignment");
    if (N == 0) return EMP_OK;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;         // 7x7, stride 2, padding 3
    const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;       // 3x3, stride 2, padding 1
    const int tiles_y = (int)emp_cdiv(PH, ST_PH), tiles_x = (int)emp_cdiv(PW, ST_PW);
    const int64_t blocks = (int64_t)N * tiles_y * tiles_x;
    EMP_REQUIRE(blocks < (1LL << 31), "stem: too many tiles");
    hipLaunchKernelGGL(stem7_kernel, dim3((unsigned)blocks), dim3(ST_THREADS), 0, emp_stream(stream), x, w_tc, scale, shift,
                       N, H, W, OH, OW, PH, PW, tiles_y, tiles_x, y);
    EMP_CHECK_LAUNCH("emp_stem_conv7_bn_relu_maxpool");
    return EMP_OK;
}
