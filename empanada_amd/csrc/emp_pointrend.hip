// D10: the PointRend subdivision step of the exported MitoNet models (empanada/models/point_rend.py:241-269,
// eval branch) as hand-written kernels.  gfx950 only.
//
// One step = upsample the semantic logits x2 (bilinear, align_corners = False), take the `k` most uncertain points of
// every image (uncertainty = -|logit| for one class, second-largest minus largest logit otherwise,
// point_rend.py:62-79), sample the decoder features and the coarse logits at those points (grid_sample, bilinear,
// align_corners = False, zero padding, point_rend.py:35-60), run the point MLP on [features, coarse] (three
// Conv1d + ReLU that re-read the coarse logits, then the predictor, point_rend.py:138-190) and write the predictions
// over the upsampled logits at those points.  The reference does this with five library calls per step (interpolate,
// topk, 2 x grid_sample, Conv1d x 4, scatter_) on NCHW tensors; here
//   emp_pr_upsample2x   one pass: upsampled logits + uncertainty map;
//   emp_pr_topk         exact top-k per image by radix select over an order-preserving integer key (4 passes of 8 bits,
//                       block-private LDS histograms with wave-aggregated increments, merged by atomics; one small
//                       "pick" launch per pass), then an ordered two-level compaction: the result is the SET of the k
//                       largest, ties at the k-th value going to the lowest pixel indices, emitted in a deterministic
//                       order (all strictly larger ones in pixel order, then the ties);
//   emp_pr_point_sample one wave per point: the four neighbours of the NHWC feature map are four contiguous 1 KiB
//                       reads (no NCHW copy of the 256-channel decoder output -- 2 GB per model call at the bench's
//                       shape), written as the rows of the MLP's input matrix with the coarse logits in channel CF;
//   the MLP             emp_conv_bn_act_nhwc on that matrix (a 1 x 1 convolution over P "pixels", bias in `shift`, the
//                       layer's output a channel slice of the next layer's input matrix, whose coarse channels are
//                       filled by emp_pr_point_sample);
//   emp_pr_scatter      predictions -> logits[n, c, idx].
// Every kernel is plain launches on the caller's stream: capturable in a HIP graph.
#include "emp_common.h"

#define PR_BLOCK 1024
#define PR_PER_THREAD 16
#define PR_CHUNK (PR_BLOCK * PR_PER_THREAD)        // elements of one image one block owns in the top-k kernels

// ------------------------------------------------------------------------------------------ upsample x2 + uncertainty
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False): src = (dst + 0.5) * 0.5 - 0.5 clamped at 0,
// i0 = floor(src), i1 = min(i0 + 1, in - 1), l1 = src - i0; value = l0y * (l0x v00 + l1x v01) + l1y * (l0x v10 + l1x v11)
__device__ __forceinline__ void pr_src(int dst, int in, int &i0, int &i1, float &l1)
{
    float s = __fsub_rn(__fmul_rn(__fadd_rn((float)dst, 0.5f), 0.5f), 0.5f);
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = __fsub_rn(s, (float)i0);
}

__global__ __launch_bounds__(256) void pr_upsample2x_kernel(const float *__restrict__ x, int N, int C, int h, int w,
                                                            float *__restrict__ y, float *__restrict__ unc)
{
    const int H = 2 * h, W = 2 * w;
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % W);
        const int Y = (int)((i / W) % H);
        const int n = (int)(i / ((int64_t)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        pr_src(Y, h, y0, y1, ly);
        pr_src(X, w, x0, x1, lx);
        const float ly0 = __fsub_rn(1.f, ly), lx0 = __fsub_rn(1.f, lx);
        float best = -INFINITY, second = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float *p = x + ((int64_t)n * C + c) * h * w;
            const float v00 = p[(int64_t)y0 * w + x0], v01 = p[(int64_t)y0 * w + x1];
            const float v10 = p[(int64_t)y1 * w + x0], v11 = p[(int64_t)y1 * w + x1];
            const float top = __fadd_rn(__fmul_rn(lx0, v00), __fmul_rn(lx, v01));
            const float bot = __fadd_rn(__fmul_rn(lx0, v10), __fmul_rn(lx, v11));
            const float v = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly, bot));
            y[((int64_t)n * C + c) * H * W + (int64_t)Y * W + X] = v;
            if (v > best) { second = best; best = v; } else if (v > second) second = v;
        }
        unc[i] = (C == 1) ? -fabsf(best) : __fsub_rn(second, best);
    }
}

extern "C" int emp_pr_upsample2x(const float *logits, int N, int C, int h, int w, float *out, float *uncertainty,
                                 void *stream)
{
    EMP_REQUIRE(logits && out && uncertainty, "pr_upsample2x: null pointer");
    EMP_REQUIRE(N >= 0 && C >= 1 && C <= 64 && h > 0 && w > 0 && (int64_t)h * w < (1LL << 28), "pr_upsample2x: bad shape");
    if (N == 0) return EMP_OK;
    const int64_t total = (int64_t)N * 4 * h * w;
    hipLaunchKernelGGL(pr_upsample2x_kernel, dim3(emp_grid(total, 256, 16384)), dim3(256), 0, emp_stream(stream), logits, N,
                       C, h, w, out, uncertainty);
    EMP_CHECK_LAUNCH("emp_pr_upsample2x");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ top-k (radix select)
// order-preserving key: larger float <-> larger unsigned (negative floats flipped, positive ones get the top bit)
__device__ __forceinline__ uint32_t pr_key(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// state = (prefix 0, mask 0, remaining k, 0) per image, histograms = 0.  A kernel, not hipMemsetAsync: a memset NODE of
// a captured HIP graph went wrong on replay once other launches had run in between -- this step
// sits inside the captured forward.
__global__ void pr_init_kernel(uint32_t *state, int N, uint32_t k)
{
    const int64_t total = (int64_t)N * (4 + 256);      // state (N, 4) followed by hist (N, 256)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        state[i] = (i < (int64_t)N * 4 && (i & 3) == 2) ? k : 0u;
}

// per-image selection state: [0] prefix, [1] mask, [2] how many of the k are still to be found among the elements
// matching the prefix, [3] unused; hist: (N, 256) zeroed by the launcher and by the pick kernel after use
__global__ __launch_bounds__(PR_BLOCK) void pr_hist_kernel(const float *__restrict__ unc, int64_t HW, int shift,
                                                           const uint32_t *__restrict__ state, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    const int n = blockIdx.y;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = state[4 * n], mask = state[4 * n + 1];
    const float *u = unc + (int64_t)n * HW;
    const int64_t base = (int64_t)blockIdx.x * PR_CHUNK;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < PR_PER_THREAD; ++it) {
        const int64_t i = base + (int64_t)it * PR_BLOCK + threadIdx.x;
        const bool in = i < HW;
        const uint32_t key = in ? pr_key(u[i]) : 0u;
        bool live = in && ((key & mask) == prefix);
        const uint32_t digit = (key >> shift) & 255u;
        // wave-aggregated increments: one LDS atomic per distinct digit in the wave (the top bytes of a float map take
        // a handful of values; plain per-lane atomics would serialise 64-fold on them)
        uint64_t todo = __ballot(live);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t d0 = __shfl(digit, leader);
            const uint64_t same = __ballot(live && digit == d0);
            if (lane == leader) atomicAdd(&h[d0], (uint32_t)__popcll(same));
            if (live && digit == d0) live = false;
            todo &= ~same;
        }
    }
    __syncthreads();
    if (threadIdx.x < 256 && h[threadIdx.x]) atomicAdd(&hist[n * 256 + threadIdx.x], h[threadIdx.x]);
}

// one wave per image: the bin of the k-th largest among the matching elements, scanning from the top bin down
__global__ __launch_bounds__(64) void pr_pick_kernel(uint32_t *__restrict__ state, uint32_t *__restrict__ hist, int shift)
{
    const int n = blockIdx.x, lane = threadIdx.x;
    uint32_t *h = hist + n * 256;
    // lane l owns bins 255 - 4 l .. 252 - 4 l (descending), in that order
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = h[255 - (4 * lane + j)]; sum += c[j]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    uint32_t before = incl - sum;                    // elements in bins above this lane's
    const uint32_t want = state[4 * n + 2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (before < want && want <= before + c[j]) {
            state[4 * n] |= (uint32_t)(255 - (4 * lane + j)) << shift;
            state[4 * n + 1] |= 255u << shift;
            state[4 * n + 2] = want - before;        // still to be found inside this bin
        }
        before += c[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) h[4 * lane + j] = 0;   // ready for the next pass
}

// counts of keys above / equal to the threshold in every block's chunk
__global__ __launch_bounds__(PR_BLOCK) void pr_count_kernel(const float *__restrict__ unc, int64_t HW,
                                                            const uint32_t *__restrict__ state, uint32_t *__restrict__ counts,
                                                            int G)
{
    __shared__ uint32_t sg, se;
    const int n = blockIdx.y;
    if (threadIdx.x == 0) { sg = 0; se = 0; }
    __syncthreads();
    const uint32_t T = state[4 * n];
    const float *u = unc + (int64_t)n * HW;
    const int64_t base = (int64_t)blockIdx.x * PR_CHUNK;
    uint32_t g = 0, e = 0;
    for (int it = 0; it < PR_PER_THREAD; ++it) {
        const int64_t i = base + (int64_t)it * PR_BLOCK + threadIdx.x;
        if (i < HW) {
            const uint32_t key = pr_key(u[i]);
            g += key > T;
            e += key == T;
        }
    }
    for (int o = 32; o; o >>= 1) { g += __shfl_xor(g, o); e += __shfl_xor(e, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sg, g); atomicAdd(&se, e); }
    __syncthreads();
    if (threadIdx.x == 0) { counts[((int64_t)n * G + blockIdx.x) * 2] = sg; counts[((int64_t)n * G + blockIdx.x) * 2 + 1] = se; }
}

// ordered emission: indices of keys > T in pixel order, then the first `ties` keys == T in pixel order
__global__ __launch_bounds__(PR_BLOCK) void pr_emit_kernel(const float *__restrict__ unc, int64_t HW, int k,
                                                           const uint32_t *__restrict__ state,
                                                           const uint32_t *__restrict__ counts, int G, int32_t *__restrict__ idx)
{
    __shared__ uint32_t wg[PR_BLOCK / 64], we[PR_BLOCK / 64], run[2];
    const int n = blockIdx.y;
    const uint32_t T = state[4 * n], ties = state[4 * n + 2];
    const uint32_t n_gt = (uint32_t)k - ties;
    if (threadIdx.x == 0) {
        uint32_t g = 0, e = 0;
        for (int b = 0; b < (int)blockIdx.x; ++b) { g += counts[((int64_t)n * G + b) * 2]; e += counts[((int64_t)n * G + b) * 2 + 1]; }
        run[0] = g;
        run[1] = e;
    }
    __syncthreads();
    const float *u = unc + (int64_t)n * HW;
    int32_t *out = idx + (int64_t)n * k;
    const int64_t base = (int64_t)blockIdx.x * PR_CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < PR_PER_THREAD; ++it) {
        const int64_t i = base + (int64_t)it * PR_BLOCK + threadIdx.x;
        uint32_t key = 0;
        const bool in = i < HW;
        if (in) key = pr_key(u[i]);
        const bool fg = in && key > T, fe = in && key == T;
        const uint64_t mg = __ballot(fg), me = __ballot(fe);
        const uint64_t below = (1ull << lane) - 1ull;
        if (lane == 0) { wg[wave] = (uint32_t)__popcll(mg); we[wave] = (uint32_t)__popcll(me); }
        __syncthreads();
        uint32_t og = run[0], oe = run[1];
        for (int v = 0; v < wave; ++v) { og += wg[v]; oe += we[v]; }
        if (fg) out[og + (uint32_t)__popcll(mg & below)] = (int32_t)i;
        if (fe) {
            const uint32_t pos = oe + (uint32_t)__popcll(me & below);
            if (pos < ties) out[n_gt + pos] = (int32_t)i;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t g = 0, e = 0;
            for (int v = 0; v < PR_BLOCK / 64; ++v) { g += wg[v]; e += we[v]; }
            run[0] += g;
            run[1] += e;
        }
        __syncthreads();
    }
}

extern "C" int64_t emp_pr_topk_work_bytes(int N, int64_t HW)
{
    const int64_t G = emp_cdiv(HW > 0 ? HW : 1, PR_CHUNK);
    return 4 * ((int64_t)N * 4 + (int64_t)N * 256 + (int64_t)N * G * 2);
}

extern "C" int emp_pr_topk(const float *uncertainty, int N, int64_t HW, int k, void *work, int64_t work_bytes,
                           int32_t *idx, void *stream)
{
    EMP_REQUIRE(uncertainty && work && idx, "pr_topk: null pointer");
    EMP_REQUIRE(N >= 0 && N <= 65535 && HW > 0 && HW < (1LL << 31) && k >= 1 && k <= HW, "pr_topk: bad sizes (k %d of %lld)", k,
                (long long)HW);
    EMP_REQUIRE(work_bytes >= emp_pr_topk_work_bytes(N, HW), "pr_topk: work buffer too small");
    if (N == 0) return EMP_OK;
    const int G = (int)emp_cdiv(HW, PR_CHUNK);
    hipStream_t st = emp_stream(stream);
    uint32_t *state = reinterpret_cast<uint32_t *>(work), *hist = state + (int64_t)N * 4, *counts = hist + (int64_t)N * 256;
    hipLaunchKernelGGL(pr_init_kernel, dim3(emp_grid((int64_t)N * 260, 256, 1024)), dim3(256), 0, st, state, N, (uint32_t)k);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hipLaunchKernelGGL(pr_hist_kernel, dim3(G, N), dim3(PR_BLOCK), 0, st, uncertainty, HW, shift, state, hist);
        hipLaunchKernelGGL(pr_pick_kernel, dim3(N), dim3(64), 0, st, state, hist, shift);
    }
    hipLaunchKernelGGL(pr_count_kernel, dim3(G, N), dim3(PR_BLOCK), 0, st, uncertainty, HW, state, counts, G);
    hipLaunchKernelGGL(pr_emit_kernel, dim3(G, N), dim3(PR_BLOCK), 0, st, uncertainty, HW, k, state, counts, G, idx);
    EMP_CHECK_LAUNCH("emp_pr_topk");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ point sample
// grid_sample(bilinear, align_corners=False, zeros) at the centre of grid point idx of the (H, W) logits grid:
// cx = 0.5 / W + x / W (point_rend.py:118-121), grid = 2 c - 1, ix = ((grid + 1) * Wf - 1) / 2, neighbours
// nw = floor, weights nw = (ix_se - ix)(iy_se - iy), ne = (ix - ix_sw)(iy_sw - iy), sw = (ix_ne - ix)(iy - iy_ne),
// se = (ix - ix_nw)(iy - iy_nw); out = nw v_nw + ne v_ne + sw v_sw + se v_se over the neighbours inside the map, in
// that order (unfused fp32).  One wave per point; lane l owns feature channels 4 l .. 4 l + 3.
struct PrSample {
    const float *feat, *coarse;
    const int32_t *idx;
    float *X0, *X1;
    int N, Hf, Wf, CF, C, k, H, W, ld;
    int64_t feat_ps;
};

__global__ __launch_bounds__(256) void pr_sample_kernel(PrSample g)
{
    const int lane = threadIdx.x & 63;
    const int64_t P = (int64_t)g.N * g.k;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = wave0; p < P; p += n_waves) {
        const int n = (int)(p / g.k);
        const int id = g.idx[p];
        const int px = id % g.W, py = id / g.W;
        const float cx = __fadd_rn(__fdiv_rn(0.5f, (float)g.W), __fdiv_rn((float)px, (float)g.W));
        const float cy = __fadd_rn(__fdiv_rn(0.5f, (float)g.H), __fdiv_rn((float)py, (float)g.H));
        const float gx = __fsub_rn(__fmul_rn(2.0f, cx), 1.0f), gy = __fsub_rn(__fmul_rn(2.0f, cy), 1.0f);
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)g.Wf), 1.f), 2.f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)g.Hf), 1.f), 2.f);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float w_nw = __fmul_rn(__fsub_rn((float)x1, ix), __fsub_rn((float)y1, iy));
        const float w_ne = __fmul_rn(__fsub_rn(ix, (float)x0), __fsub_rn((float)y1, iy));
        const float w_sw = __fmul_rn(__fsub_rn((float)x1, ix), __fsub_rn(iy, (float)y0));
        const float w_se = __fmul_rn(__fsub_rn(ix, (float)x0), __fsub_rn(iy, (float)y0));
        const bool in_x0 = x0 >= 0 && x0 < g.Wf, in_x1 = x1 >= 0 && x1 < g.Wf;
        const bool in_y0 = y0 >= 0 && y0 < g.Hf, in_y1 = y1 >= 0 && y1 < g.Hf;
        float *row0 = g.X0 + p * g.ld, *row1 = g.X1 + p * g.ld;
        // features: NHWC, 16 bytes per lane and neighbour
        for (int c = 4 * lane; c < g.CF; c += 256) {
            const float *f = g.feat + (int64_t)n * g.Hf * g.Wf * g.feat_ps + c;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define PR_ACC(ok, yy, xx, wt)                                                                          \
            if (ok) {                                                                                       \
                const float4 v = *reinterpret_cast<const float4 *>(f + ((int64_t)(yy) * g.Wf + (xx)) * g.feat_ps); \
                acc.x = __fadd_rn(acc.x, __fmul_rn(v.x, wt)); acc.y = __fadd_rn(acc.y, __fmul_rn(v.y, wt));     \
                acc.z = __fadd_rn(acc.z, __fmul_rn(v.z, wt)); acc.w = __fadd_rn(acc.w, __fmul_rn(v.w, wt));     \
            }
            PR_ACC(in_y0 && in_x0, y0, x0, w_nw)
            PR_ACC(in_y0 && in_x1, y0, x1, w_ne)
            PR_ACC(in_y1 && in_x0, y1, x0, w_sw)
            PR_ACC(in_y1 && in_x1, y1, x1, w_se)
#undef PR_ACC
            *reinterpret_cast<float4 *>(row0 + c) = acc;
        }
        // coarse logits (planar): channel CF + c of BOTH layer-input matrices; the padding channels are zeroed
        for (int c = lane; c < g.ld - g.CF; c += 64) {
            float v = 0.f;
            if (c < g.C) {
                const float *q = g.coarse + ((int64_t)n * g.C + c) * g.Hf * g.Wf;
                if (in_y0 && in_x0) v = __fadd_rn(v, __fmul_rn(q[(int64_t)y0 * g.Wf + x0], w_nw));
                if (in_y0 && in_x1) v = __fadd_rn(v, __fmul_rn(q[(int64_t)y0 * g.Wf + x1], w_ne));
                if (in_y1 && in_x0) v = __fadd_rn(v, __fmul_rn(q[(int64_t)y1 * g.Wf + x0], w_sw));
                if (in_y1 && in_x1) v = __fadd_rn(v, __fmul_rn(q[(int64_t)y1 * g.Wf + x1], w_se));
            }
            row0[g.CF + c] = v;
            row1[g.CF + c] = v;
        }
    }
}

extern "C" int emp_pr_point_sample(const float *feat_nhwc, int64_t feat_pixel_stride, const float *coarse, int N, int Hf,
                                   int Wf, int CF, int C, const int32_t *idx, int k, int H, int W, float *X0, float *X1,
                                   int ld, void *stream)
{
    EMP_REQUIRE(feat_nhwc && coarse && idx && X0 && X1, "pr_point_sample: null pointer");
    EMP_REQUIRE(N >= 0 && Hf > 0 && Wf > 0 && CF > 0 && CF % 4 == 0 && C >= 1 && k >= 1 && H > 0 && W > 0,
                "pr_point_sample: bad shape");
    EMP_REQUIRE(ld >= CF + C && ld % 4 == 0 && feat_pixel_stride >= CF && feat_pixel_stride % 4 == 0,
                "pr_point_sample: bad strides");
    EMP_REQUIRE(((reinterpret_cast<uintptr_t>(feat_nhwc) | reinterpret_cast<uintptr_t>(X0) | reinterpret_cast<uintptr_t>(X1)) & 15) == 0,
                "pr_point_sample: pointers must be 16-byte aligned");
    if (N == 0) return EMP_OK;
    PrSample g;
    g.feat = feat_nhwc; g.coarse = coarse; g.idx = idx; g.X0 = X0; g.X1 = X1;
    g.N = N; g.Hf = Hf; g.Wf = Wf; g.CF = CF; g.C = C; g.k = k; g.H = H; g.W = W; g.ld = ld; g.feat_ps = feat_pixel_stride;
    const int64_t P = (int64_t)N * k;
    hipLaunchKernelGGL(pr_sample_kernel, dim3(emp_grid(P * 64, 256, 16384)), dim3(256), 0, emp_stream(stream), g);
    EMP_CHECK_LAUNCH("emp_pr_point_sample");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ scatter
__global__ __launch_bounds__(256) void pr_scatter_kernel(const float *__restrict__ pts, int ldp, const int32_t *__restrict__ idx,
                                                         int N, int C, int k, int64_t HW, float *__restrict__ logits)
{
    const int64_t total = (int64_t)N * k * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int n = (int)(p / k);
        const int64_t id = idx[p];
        if (id >= 0 && id < HW) logits[((int64_t)n * C + c) * HW + id] = pts[p * ldp + c];    // (a bad index writes nothing)
    }
}

extern "C" int emp_pr_scatter(const float *points, int ld_points, const int32_t *idx, int N, int C, int k, int64_t HW,
                              float *logits, void *stream)
{
    EMP_REQUIRE(points && idx && logits, "pr_scatter: null pointer");
    EMP_REQUIRE(N >= 0 && C >= 1 && k >= 1 && HW >= k && ld_points >= C, "pr_scatter: bad shape");
    if (N == 0) return EMP_OK;
    const int64_t total = (int64_t)N * k * C;
    hipLaunchKernelGGL(pr_scatter_kernel, dim3(emp_grid(total, 256, 4096)), dim3(256), 0, emp_stream(stream), points,
                       ld_points, idx, N, C, k, HW, logits);
    EMP_CHECK_LAUNCH("emp_pr_scatter");
    return EMP_OK;
}
