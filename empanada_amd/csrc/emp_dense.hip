// Dense-path helper kernels that are HBM-bound rather than MFMA-bound (gfx950 only): the depthwise half of the
// separable 5x5 convolutions of the decoder and the heads.  The GEMM-shaped convolutions stay with MIOpen.
#include "emp_common.h"

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
// A row of an NHWC image is W*C contiguous floats, so thread t of a row owns the float4 at index t of that row
// (pixel t / C4, channels 4*(t % C4) ..) and its K horizontal taps are the float4s at t + (j - P) * C4: every
// load of a wave is one contiguous 1 KiB segment.  Vertically a thread marches down RY output rows keeping K
// partial output rows in registers, so each input row is loaded once per thread (K times per pixel over the
// neighbouring lanes, served by L1/L2).  The K*K weights of the thread's 4 channels stay in registers.
// Output element = bias + fma chain over the taps in raster order (i, then j), starting from +0.
// HBM traffic: 4 B read + 4 B written per element (+ (K-1)/RY of halo rows, mostly L2 hits).
template <int K>
__global__ __launch_bounds__(256) void dwconv_nhwc_kernel(const float4 *__restrict__ x, const float4 *__restrict__ w,
                                                          const float4 *__restrict__ bias, int H, int W, int C4,
                                                          int RY, float4 *__restrict__ y)
{
    constexpr int P = K / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int rowlen = W * C4;
    if (t >= rowlen) return;
    const int c4 = t % C4, xcol = t / C4;
    const int n = blockIdx.z, y0 = blockIdx.y * RY;
    const int y1 = min(y0 + RY, H);

    f4 wt[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) wt[i][j] = ld4(w + (i * K + j) * C4 + c4);
    bool xin[K];
#pragma unroll
    for (int j = 0; j < K; ++j) xin[j] = (xcol + j - P >= 0) && (xcol + j - P < W);
    f4 b = zero4();
    if (bias) b = ld4(bias + c4);

    f4 acc[K];
#pragma unroll
    for (int s = 0; s < K; ++s) acc[s] = zero4();

    const float4 *xin_n = x + (int64_t)n * H * rowlen + t;
    float4 *yout_n = y + (int64_t)n * H * rowlen + t;
    const int rows = (y1 - y0) + 2 * P;
    for (int rr = 0; rr < rows; rr += K) {
#pragma unroll
        for (int ph = 0; ph < K; ++ph) {
            const int r = y0 - P + rr + ph;                 // input row
            if (rr + ph < rows) {
                f4 in[K];
                const bool rin = (r >= 0) && (r < H);
                const float4 *xr = xin_n + (int64_t)r * rowlen;
#pragma unroll
                for (int j = 0; j < K; ++j) in[j] = (rin && xin[j]) ? ld4(xr + (j - P) * C4) : zero4();
                // input row r is tap row i of output row r + P - i, which lives in slot (ph - i) mod K
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    f4 &a = acc[(ph - i + K) % K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        a.lo = __builtin_elementwise_fma(in[j].lo, wt[i][j].lo, a.lo);
                        a.hi = __builtin_elementwise_fma(in[j].hi, wt[i][j].hi, a.hi);
                    }
                }
                // output row r - P has now seen its last tap row (i = K - 1)
                f4 &done = acc[(ph + 1) % K];
                const int o = r - P;
                if (o >= y0 && o < y1) {
                    float4 v;
                    v.x = done.lo.x + b.lo.x; v.y = done.lo.y + b.lo.y;
                    v.z = done.hi.x + b.hi.x; v.w = done.hi.y + b.hi.y;
                    yout_n[(int64_t)o * rowlen] = v;
                }
                done = zero4();
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
    const int gx = (int)emp_cdiv((int64_t)W * C4, 256);
    // rows per block: long strips amortise the K-1 halo rows; short ones keep >= ~2k blocks in flight
    int RY = 32;
    while (RY > 8 && (int64_t)gx * emp_cdiv(H, RY) * N < 2048) RY >>= 1;
    const int gy = (int)emp_cdiv(H, RY);
    EMP_REQUIRE(gy <= 65535, "dwconv: image too tall");
    hipStream_t st = emp_stream(stream);
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *w4 = reinterpret_cast<const float4 *>(w_kkc);
    const float4 *b4 = reinterpret_cast<const float4 *>(bias);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    dim3 grid(gx, gy, N);
    if (k == 3) hipLaunchKernelGGL((dwconv_nhwc_kernel<3>), grid, dim3(256), 0, st, x4, w4, b4, H, W, C4, RY, y4);
    else hipLaunchKernelGGL((dwconv_nhwc_kernel<5>), grid, dim3(256), 0, st, x4, w4, b4, H, W, C4, RY, y4);
    EMP_CHECK_LAUNCH("emp_dwconv_nhwc");
    return EMP_OK;
}
