// D8: grouped 3x3 convolution (+ BatchNorm + ReLU epilogue) on the fp32 matrix cores, NHWC.  gfx950 only.
//
//   RegNet bottlenecks (empanada/models/encoders/regnet.py:59-71) run their 3x3 convolution in groups of `group_w`
//   channels (72 for RegNetY-6.4GF: 2 / 4 / 8 / 18 groups): a block-diagonal GEMM.  A dense implicit-GEMM kernel
//   would spend its 32x32 / 128-wide tiles on the zero blocks, so the group is a grid dimension here and the tile
//   is cut to the group: block = one group x 128 output pixels, GEMM tile 128 x GW (padded to 16s) x (9 * GW), on
//   v_mfma_f32_16x16x4_f32 (16-wide tiles waste 80 / 72 on the couts and nothing on K, where 32x32x2 tiles would waste
//   96 / 72 and a 16- or 32-channel K granule another 80 / 72 or 96 / 72).
//   256 threads = 4 waves, each wave 32 pixels (2 row tiles) x all NTL cout tiles: 2 * NTL accumulators of 4 VGPRs.
//   K loop: 9 taps x GW / CK chunks of CK channels (24, 16 or 8: the largest that divides GW), one barrier per chunk,
//   global -> registers -> LDS with two LDS buffers: chunk s is consumed while chunk s + 1 is written to the other
//   buffer and the loads of chunk s + 2 are in flight.  LDS rows are CK + 4 floats: (CK + 4) / 4 is odd, so the 16
//   rows of a fragment read start in 16 different 4-bank groups and the 8-byte reads of two k-quarters fill a group.
//   K order inside a chunk (the MFMA is bit-for-bit an fmaf chain over its 4 k's, lane quarter kq = k): for each
//   8-channel slab j, for e in {0, 1}: channels 8j + 2kq + e, kq = 0..3 -- see emp_hip.h, the oracle mirrors it.
//   Tile numbering is XCD-aware (as in emp_conv.hip) with the group fastest: the G blocks that read the same pixels
//   (different channel ranges of the same cache lines) are neighbours on one XCD's L2.
#include "emp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) const float lds_cf;

#define GC_BM 128
#define GC_THREADS 256

struct GConvGeom {
    const float *x, *w, *scale, *shift;
    float *out;
    int N, H, W, OH, OW, G, GW, stride, relu;
    int64_t M, x_ps, out_ps;
    int tiles_m;
};

template <int CK, int NTL>
__global__ __launch_bounds__(GC_THREADS, (NTL >= 6 ? 2 : 3)) void gconv3x3_f32_kernel(GConvGeom g)
{
    constexpr int RS = CK + 4;                          // LDS row (floats)
    constexpr int F4R = CK / 4;                         // float4 per staged row
    constexpr int BN = NTL * 16;                        // padded couts of the group
    constexpr int A_F4 = GC_BM * F4R / GC_THREADS;      // A float4 per thread and chunk (CK / 8)
    constexpr int B_TOT = BN * F4R;                     // B float4 per chunk
    constexpr int B_F4 = (B_TOT + GC_THREADS - 1) / GC_THREADS;
    constexpr int CLD = BN + 4;                         // epilogue staging row
    constexpr int AB_ELEMS = 2 * (GC_BM + BN) * RS;
    constexpr int C_ELEMS = GC_BM * CLD;
    constexpr int SMEM = AB_ELEMS > C_ELEMS ? AB_ELEMS : C_ELEMS;
    static_assert(CK % 8 == 0 && (RS / 4) % 2 == 1, "chunk must be 8, 16 or 24 channels");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float *As = smem, *Bs = smem + 2 * GC_BM * RS;

    const int T = g.tiles_m * g.G;
    const int chunk = (T + 7) >> 3;
    const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile >= T) return;
    const int tm = tile / g.G, grp = tile - tm * g.G;
    const int64_t m0 = (int64_t)tm * GC_BM;
    const int tid = threadIdx.x;
    const int GW = g.GW;
    const float *xg = g.x + (int64_t)grp * GW;
    const float *wg = g.w + (int64_t)grp * GW * 9 * GW;

    // rows this thread stages
    int a_row[A_F4], a_c[A_F4], a_n[A_F4], a_iy[A_F4], a_ix[A_F4];
    bool a_ok[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int q = tid + GC_THREADS * i;
        a_row[i] = q / F4R;
        a_c[i] = (q - a_row[i] * F4R) * 4;
        const int64_t p = m0 + a_row[i];
        a_ok[i] = p < g.M;
        const int64_t pp = a_ok[i] ? p : 0;
        const int ox = (int)(pp % g.OW);
        const int oy = (int)((pp / g.OW) % g.OH);
        a_n[i] = (int)(pp / ((int64_t)g.OW * g.OH));
        a_iy[i] = oy * g.stride - 1;
        a_ix[i] = ox * g.stride - 1;
    }
    int b_row[B_F4], b_c[B_F4];
    bool b_ok[B_F4], b_st[B_F4];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int q = tid + GC_THREADS * i;
        b_st[i] = q < B_TOT;                               // this thread stages a float4 of B in pass i
        b_row[i] = b_st[i] ? q / F4R : 0;
        b_c[i] = (q - (q / F4R) * F4R) * 4;
        b_ok[i] = b_st[i] && b_row[i] < GW;                // rows past the group's couts are zeros
    }

    const int cpt = GW / CK;                               // chunks per tap
    const int S = 9 * cpt;
    float4 ra[A_F4], rb[B_F4];
    bool r_in[A_F4];
    int ld_tap = 0, ld_c0 = 0;
    const float *a_ptr[A_F4];
    bool a_in[A_F4];
    auto set_tap = [&](int tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            a_in[i] = a_ok[i] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            a_ptr[i] = a_in[i] ? xg + (((int64_t)a_n[i] * g.H + iy) * g.W + ix) * g.x_ps + a_c[i] : xg + a_c[i];
        }
    };
    set_tap(0);
    auto load_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            r_in[i] = a_in[i];
            ra[i] = *reinterpret_cast<const float4 *>(a_ptr[i] + ld_c0);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i)
            rb[i] = *reinterpret_cast<const float4 *>(wg + ((int64_t)(b_ok[i] ? b_row[i] : 0) * 9 + ld_tap) * GW + ld_c0 + b_c[i]);
        ld_c0 += CK;
        if (ld_c0 == GW) {                                  // block-uniform
            ld_c0 = 0;
            ++ld_tap;
            if (ld_tap < 9) set_tap(ld_tap);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            float4 v = ra[i];
            if (!r_in[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&As[(buf * GC_BM + a_row[i]) * RS + a_c[i]]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i)
            if (b_st[i]) {
                float4 v = rb[i];
                if (!b_ok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(&Bs[(buf * BN + b_row[i]) * RS + b_c[i]]) = v;
            }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, kq = lane >> 4;
    f32x4 acc[2][NTL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk();
    store_chunk(0);
    if (S > 1) load_chunk();
    __syncthreads();
    // Fragment reads are volatile 8-byte loads: the compiler would otherwise pair them into ds_read2_b64, which is
    // serviced in 16-lane groups on 32 banks (the rows r and r + 8 of a tile collide with this row stride) at half the
    // rate; a plain ds_read_b64 is serviced in 32-lane groups on 64 banks, for which the layout is conflict-free.
    // The fragments of slab j + 1 are requested before the MFMAs of slab j.
    constexpr int NJ = CK / 8;
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        const float *Ab = &As[(buf * GC_BM + wave * 32 + r16) * RS + kq * 2];
        const float *Bb = &Bs[(buf * BN + r16) * RS + kq * 2];
        f32x2 fa[2][2], fb[2][NTL];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = *(const volatile lds_f32x2 *)(lds_cf *)(Ab + i * 16 * RS);
#pragma unroll
        for (int n = 0; n < NTL; ++n) fb[0][n] = *(const volatile lds_f32x2 *)(lds_cf *)(Bb + n * 16 * RS);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cur = j & 1;
            if (j + 1 < NJ) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[cur ^ 1][i] = *(const volatile lds_f32x2 *)(lds_cf *)(Ab + i * 16 * RS + (j + 1) * 8);
#pragma unroll
                for (int n = 0; n < NTL; ++n)
                    fb[cur ^ 1][n] = *(const volatile lds_f32x2 *)(lds_cf *)(Bb + n * 16 * RS + (j + 1) * 8);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < NTL; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][e], fb[cur][n][e], acc[i][n], 0, 0, 0);
            if (j == 0) {
                if (s + 1 < S) store_chunk(buf ^ 1);        // chunk s+1: registers -> the buffer read in iteration s-1
                if (s + 2 < S) load_chunk();                // chunk s+2: global -> registers
            }
        }
        __syncthreads();
    }

    // epilogue: accumulators -> LDS tile [128][BN] (C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg), then
    // y = relu?(acc * scale + shift) as float4 along the group's couts
    float *Cs = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NTL; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) Cs[(wave * 32 + i * 16 + kq * 4 + q) * CLD + n * 16 + r16] = acc[i][n][q];
    __syncthreads();
    const int c4n = GW >> 2;                                // float4 columns of the group
    const int co0 = grp * GW;
    for (int idx = tid; idx < GC_BM * c4n; idx += GC_THREADS) {
        const int row = idx / c4n, c = (idx - row * c4n) * 4;
        const int64_t p = m0 + row;
        if (p >= g.M) continue;
        const float4 a4 = *reinterpret_cast<const float4 *>(&Cs[row * CLD + c]);
        float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (g.scale) v[e] = __fmul_rn(v[e], g.scale[co0 + c + e]);
            if (g.shift) v[e] = __fadd_rn(v[e], g.shift[co0 + c + e]);
            if (g.relu) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<float4 *>(g.out + p * g.out_ps + co0 + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int emp_gconv_chunk(int group_w) { return group_w % 24 == 0 ? 24 : group_w % 16 == 0 ? 16 : 8; }

extern "C" int emp_gconv3x3_bn_act_nhwc(const float *x, int64_t x_pixel_stride, const float *w_okkc, const float *scale,
                                        const float *shift, int relu, int N, int H, int W, int groups, int group_w,
                                        int stride, float *out, int64_t out_pixel_stride, void *stream)
{
    EMP_REQUIRE(x && w_okkc && out, "gconv: null pointer");
    EMP_REQUIRE(N >= 0 && H > 0 && W > 0 && groups >= 1, "gconv: bad shape");
    EMP_REQUIRE(group_w >= 8 && group_w % 8 == 0 && group_w <= 128, "gconv: group width %d must be a multiple of 8 in 8..128", group_w);
    EMP_REQUIRE(stride == 1 || stride == 2, "gconv: stride %d not 1 or 2", stride);
    const int64_t C = (int64_t)groups * group_w;
    if (x_pixel_stride == 0) x_pixel_stride = C;
    if (out_pixel_stride == 0) out_pixel_stride = C;
    EMP_REQUIRE(x_pixel_stride >= C && out_pixel_stride >= C && (x_pixel_stride & 3) == 0 && (out_pixel_stride & 3) == 0,
                "gconv: bad pixel stride");
    EMP_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_okkc) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                "gconv: x, w and out must be 16-byte aligned");
    EMP_REQUIRE(out != x, "gconv: output cannot alias the input");
    if (N == 0) return EMP_OK;
    GConvGeom g;
    g.x = x; g.w = w_okkc; g.scale = scale; g.shift = shift; g.out = out;
    g.N = N; g.H = H; g.W = W; g.G = groups; g.GW = group_w; g.stride = stride; g.relu = relu;
    g.OH = (H + 2 - 3) / stride + 1;
    g.OW = (W + 2 - 3) / stride + 1;
    g.M = (int64_t)N * g.OH * g.OW;
    g.x_ps = x_pixel_stride; g.out_ps = out_pixel_stride;
    const int64_t tiles_m = emp_cdiv(g.M, GC_BM);
    EMP_REQUIRE(tiles_m * groups < (1LL << 28), "gconv: too many tiles");
    g.tiles_m = (int)tiles_m;
    const int T = g.tiles_m * groups;
    const int grid = 8 * ((T + 7) / 8);
    const int ck = emp_gconv_chunk(group_w), ntl = (group_w + 15) / 16;
#define GC_GO(CK_, NTL_) hipLaunchKernelGGL((gconv3x3_f32_kernel<CK_, NTL_>), dim3(grid), dim3(GC_THREADS), 0, emp_stream(stream), g)
    // chunk / tile combinations that exist: CK 24 -> GW 24, 48, 72, 96, 120; CK 16 -> 16, 32, 64, 80, 112, 128;
    // CK 8 -> 8, 40, 56, 88, 104
    if (ck == 24) {
        switch (ntl) { case 2: GC_GO(24, 2); break; case 3: GC_GO(24, 3); break; case 5: GC_GO(24, 5); break;
                       case 6: GC_GO(24, 6); break; default: GC_GO(24, 8); break; }
    } else if (ck == 16) {
        switch (ntl) { case 1: GC_GO(16, 1); break; case 2: GC_GO(16, 2); break; case 4: GC_GO(16, 4); break;
                       case 5: GC_GO(16, 5); break; case 7: GC_GO(16, 7); break; default: GC_GO(16, 8); break; }
    } else {
        switch (ntl) { case 1: GC_GO(8, 1); break; case 3: GC_GO(8, 3); break; case 4: GC_GO(8, 4); break;
                       case 6: GC_GO(8, 6); break; default: GC_GO(8, 7); break; }
    }
#undef GC_GO
    EMP_CHECK_LAUNCH("emp_gconv3x3_bn_act_nhwc");
    return EMP_OK;
}
