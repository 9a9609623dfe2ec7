// D4b: weight-stationary 1x1 convolution + BatchNorm (+ residual) (+ ReLU) for the SHORT-K pointwise layers of the
// ResNet bottleneck (layer1 / layer2: conv3 64 -> 256 and 128 -> 512 with the identity added, the 64 -> 256 shortcut
// projection).  gfx950 only.
//
// Why a second kernel.  These layers move 2.3-4.6 KB per pixel for 33-131 kFLOP: 14-28 flop per byte, at or below the
// ridge of the fp32 matrix pipe (157 TF/s / 6.3 TB/s = 25).  The tiled implicit-GEMM kernel (emp_conv.hip) spends a
// block's life in prologue and epilogue there -- 2 or 4 K-slabs between staging 64 KB of operands and draining 64 KB
// of output through LDS -- and reaches 3.6-3.9 TB/s on layer1, 2.7 TB/s on layer2 (profiles/r2_res1x1_variants).
// Here nothing but the weights touches LDS, and they are staged ONCE per block:
//   * grid = one 256-thread block per CU, persistent; a block owns 128 couts (blockIdx -> (pixel block, cout group),
//     the cout groups of one pixel block on the same XCD so that the activations they share come out of that XCD's
//     L2) and keeps their 128 x Cin weights in LDS for its whole life (34-68 KB);
//   * a wave owns a 32-pixel x 128-cout tile per step: the A operand (32 pixels x Cin) goes from global memory
//     STRAIGHT INTO THE MFMA SOURCE REGISTERS -- with the K order "k-step j consumes channels j (lanes 0-31) and
//     32 + j (lanes 32-63) of a 64-channel slab" lane (r, h) needs exactly the contiguous 128 bytes
//     x[pixel r][64 s + 32 h .. + 32) -- no LDS staging, no barrier anywhere in the loop;
//   * the four 32 x 32 accumulator tiles of a wave cover couts n0 + 4 c + j (tile j, column c): the weights are
//     staged in that permuted row order, so that in the epilogue lane c holds FOUR CONSECUTIVE couts of a pixel in the
//     same register index of its four tiles -- residual loads and output stores are float4 per lane, 512 contiguous
//     bytes per half-wave, straight from / to the accumulators (no LDS transpose);
//   * ONE wave per SIMD with the whole 512-entry register file (two waves of 256 spill), software-pipelined over two
//     register sets: a tile's residual and the NEXT tile's activations are requested a whole matrix phase (128-256
//     MFMAs, 3-7 us) before they are consumed (~100 KB in flight per CU), and the epilogue of tile i - 1 is issued
//     in the shadow of tile i's MFMAs (two accumulator sets).
// Summation order per output: one fmaf chain from +0 over 64-channel slabs ascending, inside a slab j = 0..31:
// channel j, then channel 32 + j -- the order of emp_conv_bn_act_nhwc with a K-slab of 64 (emp_conv_k_slab_geom);
// oracle/dense.py::conv_bn_act_nhwc(slab=64) reproduces it bit for bit.
#include "emp_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PW_THREADS 256
#define PW_BN 128                  // couts per block
#define PW_ROWS 32                 // pixels per wave tile

struct PwGeom {
    const float *x, *w, *scale, *shift, *res;
    float *out;
    int64_t M, out_ps, res_ps;
    int Cin, Cout, relu, groups, pix_blocks;
};

template <int KS, bool RES, bool RELU>
__global__ __launch_bounds__(PW_THREADS, 1) void conv1x1_ws_kernel(PwGeom g)
{
    constexpr int CIN = 64 * KS;
    constexpr int LD = CIN + 4;                        // LDS row (floats): 16 lanes x 16 B cover all 64 banks once
    __shared__ __attribute__((aligned(16))) float Bs[PW_BN * LD];

    // blockIdx -> (pixel block pb, cout group gi): consecutive hardware block ids go to different XCDs; the cout
    // groups of one pixel block take consecutive slots of ONE XCD
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int per_xcd = g.pix_blocks >> 3;             // pixel blocks per XCD (pix_blocks is a multiple of 8)
    const int gi = slot % g.groups;
    const int pb = xcd * per_xcd + slot / g.groups;
    const int n0 = gi * PW_BN;
    const int tid = threadIdx.x;

    // weights of couts n0 + 4 c + j  ->  LDS row j * 32 + c
    for (int idx = tid; idx < PW_BN * (CIN / 4); idx += PW_THREADS) {
        const int rr = idx / (CIN / 4), k4 = idx - rr * (CIN / 4);
        const int co = n0 + 4 * (rr & 31) + (rr >> 5);
        *reinterpret_cast<float4 *>(&Bs[rr * LD + 4 * k4]) = *reinterpret_cast<const float4 *>(g.w + (int64_t)co * CIN + 4 * k4);
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int co = n0 + 4 * r;                         // the lane's four consecutive couts
    float sc[4], sh[4];                                // scale and shift are both present (eligibility)
    { const float4 t = *reinterpret_cast<const float4 *>(g.scale + co); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
    { const float4 t = *reinterpret_cast<const float4 *>(g.shift + co); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }

    const int64_t n_tiles = (g.M + PW_ROWS - 1) / PW_ROWS;
    const int64_t stride_t = (int64_t)g.pix_blocks * (PW_THREADS / 64);
    const float *Bl = &Bs[r * LD + hh * 32];           // the lane's row of tile 0, its half of a slab

    // ---- the pieces of a tile's life --------------------------------------------------------------------------
    // A = 128 contiguous bytes per lane and slab
    auto load_a = [&](int64_t t, float4 (&a)[KS][8]) {
        const int64_t pa = t * PW_ROWS + r;            // (only full tiles come through here)
        const float *ap = g.x + pa * CIN + hh * 32;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) a[s][q] = *reinterpret_cast<const float4 *>(ap + 64 * s + 4 * q);
    };
    // residual of the lane's 16 (pixel, 4 couts) outputs
    auto load_rs = [&](int64_t t, float4 (&rs)[RES ? 16 : 1]) {
        if constexpr (RES) {
            const float *rp = g.res + (t * PW_ROWS + 4 * hh) * g.res_ps + co;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                rs[q] = *reinterpret_cast<const float4 *>(rp + (int64_t)((q & 3) + 8 * (q >> 2)) * g.res_ps);
        }
    };
    auto mma = [&](f32x16 (&acc)[4], const float4 (&a)[KS][8]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        // the weight fragments do not depend on the tile: without this the compiler hoists all 32 * KS float4 LDS reads
        // of a lane out of the tile loop (512+ registers) -- the address is made opaque once per tile instead
        int opaque = 0;
        asm volatile("" : "+v"(opaque));               // (an offset, not the pointer: the LDS address space must survive)
        const float *Bt = Bl + opaque;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float4 b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float4 *>(Bt + j * 32 * LD + 64 * s + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = e == 0 ? a[s][q].x : e == 1 ? a[s][q].y : e == 2 ? a[s][q].z : a[s][q].w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bv = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                    }
                }
            }
    };
    // epilogue straight from the accumulators: register q of tile j = pixel row(q, hh), cout co + j
    auto outv = [&](const f32x16 (&acc)[4], const float4 (&rs)[RES ? 16 : 1], int q) {
        float v[4] = {acc[0][q], acc[1][q], acc[2][q], acc[3][q]};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(__fmul_rn(v[e], sc[e]), sh[e]);
        if constexpr (RES) {
            v[0] = __fadd_rn(v[0], rs[q].x); v[1] = __fadd_rn(v[1], rs[q].y);
            v[2] = __fadd_rn(v[2], rs[q].z); v[3] = __fadd_rn(v[3], rs[q].w);
        }
        if constexpr (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto epi = [&](const f32x16 (&acc)[4], const float4 (&rs)[RES ? 16 : 1], int64_t t) {       // full tiles
        float *op = g.out + (t * PW_ROWS + 4 * hh) * g.out_ps + co;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            *reinterpret_cast<float4 *>(op + (int64_t)((q & 3) + 8 * (q >> 2)) * g.out_ps) = outv(acc, rs, q);
    };

    // ---- full tiles, software-pipelined over two register sets: in phase i the residual of tile i and the activations
    // of tile i + 1 are requested, then the MFMAs of tile i (into acc[i % 2]) and the epilogue of tile i - 1 (out of
    // acc[(i - 1) % 2]) sit in ONE basic block, so that the scheduler can slot the epilogue's vector ALU work and
    // stores between the matrix instructions -- with one wave per SIMD nothing else would fill the matrix pipe's
    // shadow.  Loads are consumed one phase (128-256 MFMAs, 3-7 us) after they were issued.
    const int64_t n_full = g.M / PW_ROWS;
    const int64_t t_first = (int64_t)pb * (PW_THREADS / 64) + wave;
    if (t_first < n_full) {
        const int64_t m = (n_full - t_first + stride_t - 1) / stride_t;       // tiles of this wave
        auto tile = [&](int64_t i) { return t_first + (i < m ? i : m - 1) * stride_t; };   // (past the end: re-request the last)
        float4 a0[KS][8], a1[KS][8], rs0[RES ? 16 : 1], rs1[RES ? 16 : 1];
        f32x16 acc0[4], acc1[4];
        load_a(tile(0), a0);
        load_rs(tile(0), rs0);
        load_a(tile(1), a1);
        mma(acc0, a0);
        int64_t i = 1;
        for (; i + 1 < m; i += 2) {
            load_rs(tile(i), rs1);
            load_a(tile(i + 1), a0);
            mma(acc1, a1);
            epi(acc0, rs0, tile(i - 1));
            load_rs(tile(i + 1), rs0);
            load_a(tile(i + 2), a1);
            mma(acc0, a0);
            epi(acc1, rs1, tile(i));
        }
        if (i < m) {                                   // one more tile (odd index), then its epilogue
            load_rs(tile(i), rs1);
            mma(acc1, a1);
            epi(acc0, rs0, tile(i - 1));
            epi(acc1, rs1, tile(i));
        } else {
            epi(acc0, rs0, tile(i - 1));
        }
    }
    // ---- the launch's one partial tile (M % 32 rows), by the wave whose turn it would be: clamped loads, masked stores
    if (n_full < n_tiles && (n_full - t_first) % stride_t == 0 && n_full >= t_first) {
        const int64_t p0 = n_full * PW_ROWS;
        const int64_t pa = (p0 + r < g.M) ? p0 + r : g.M - 1;
        const float *ap = g.x + pa * CIN + hh * 32;
        float4 a[KS][8], rs[RES ? 16 : 1];
        f32x16 acc[4];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) a[s][q] = *reinterpret_cast<const float4 *>(ap + 64 * s + 4 * q);
        if constexpr (RES) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int64_t p = p0 + (q & 3) + 8 * (q >> 2) + 4 * hh;
                p = p < g.M ? p : g.M - 1;
                rs[q] = *reinterpret_cast<const float4 *>(g.res + p * g.res_ps + co);
            }
        }
        mma(acc, a);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t p = p0 + (q & 3) + 8 * (q >> 2) + 4 * hh;
            if (p < g.M) *reinterpret_cast<float4 *>(g.out + p * g.out_ps + co) = outv(acc, rs, q);
        }
    }
}

// Shapes the weight-stationary kernel takes (everything else stays on conv_igemm_f32_kernel): 1x1, stride 1, no
// padding, Cin 64 or 128, whole 128-cout groups, and enough pixels to give every CU's eight waves several tiles.
// (the launcher additionally wants scale AND shift -- every call site on the path is conv + BatchNorm)
extern "C" int emp_conv1x1_ws_eligible(int64_t M, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu)
{
    static const char *off = getenv("EMP_CONV_NO_WS");         // experiments only (A/B against the tiled kernel)
    if (off && off[0] == '1') return 0;
    return KH == 1 && KW == 1 && stride == 1 && pad == 0 && (Cin == 64 || Cin == 128) && Cout % PW_BN == 0 &&
           Cout / PW_BN <= 8 && relu != 2 && M >= 65536;
}

// called by emp_conv_bn_act_nhwc for eligible shapes (pointers and strides already checked for 16-byte alignment)
extern "C" __attribute__((visibility("hidden"))) int emp_conv1x1_ws_launch(const float *x, const float *w, const float *scale, const float *shift,
                                     const float *res, int64_t res_ps, int relu, int64_t M, int Cin, int Cout,
                                     float *out, int64_t out_ps, void *stream)
{
    PwGeom g;
    g.x = x; g.w = w; g.scale = scale; g.shift = shift; g.res = res; g.out = out;
    g.M = M; g.out_ps = out_ps; g.res_ps = res_ps; g.Cin = Cin; g.Cout = Cout; g.relu = relu;
    g.groups = Cout / PW_BN;
    // one block per CU (256 CUs): pix_blocks x groups blocks, pix_blocks a multiple of 8 (XCDs)
    int pix = 256 / g.groups;
    pix = pix / 8 * 8;
    if (pix < 8) pix = 8;
    g.pix_blocks = pix;
    const dim3 grid(pix * g.groups), block(PW_THREADS);
    hipStream_t st = emp_stream(stream);
#define PW_GO(KS_, RES_, RELU_) hipLaunchKernelGGL((conv1x1_ws_kernel<KS_, RES_, RELU_>), grid, block, 0, st, g)
#define PW_GO2(KS_, RES_) do { if (relu) PW_GO(KS_, RES_, true); else PW_GO(KS_, RES_, false); } while (0)
    if (Cin == 64) { if (res) PW_GO2(1, true); else PW_GO2(1, false); }
    else { if (res) PW_GO2(2, true); else PW_GO2(2, false); }
#undef PW_GO2
#undef PW_GO
    EMP_CHECK_LAUNCH("emp_conv_bn_act_nhwc(1x1 weight-stationary)");
    return EMP_OK;
}
