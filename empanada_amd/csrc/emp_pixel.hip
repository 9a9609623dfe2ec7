// Per-pixel kernels of the panoptic post-processing: recursive median + harden (P1/P2),
// centre NMS (P3), nearest-centre grouping (P4), semantic/instance fusion (P5).
// gfx950 only; wave = 64.  All HBM-bound except group_pixels at large K (VALU-bound).
#include "emp_common.h"

thread_local char emp_err_buf[512] = "";

extern "C" int emp_version(void) { return 100; }
extern "C" const char *emp_last_error(void) { return emp_err_buf; }
extern "C" int emp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// median of KS registers (odd KS): partial bubble, m+1 passes leave the median at v[m].
template <int KS>
__device__ __forceinline__ float median_regs(const float (&in)[KS])
{
    float v[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) v[i] = in[i];
    constexpr int M = KS / 2;
#pragma unroll
    for (int pass = 0; pass <= M; ++pass) {
#pragma unroll
        for (int i = 0; i < KS - 1 - pass; ++i) {
            float lo = fminf(v[i], v[i + 1]);
            float hi = fmaxf(v[i], v[i + 1]);
            v[i] = lo;
            v[i + 1] = hi;
        }
    }
    return v[M];
}

// ------------------------------------------------------------------------------------------
// P1+P2, C == 1: one thread per pixel walks the stack in z with the filter window in registers.
// Loads of the next PF slices are issued before the current PF medians are computed so that
// every lane keeps PF dword loads in flight (the recursion itself is serial in z).
// Algorithmic traffic: 4 B read + 1 B write per voxel (+4 B if out_prob).
template <int KS>
__global__ __launch_bounds__(256) void median_harden_c1_kernel(const float *__restrict__ prob, int D,
                                                               int64_t HW, float thr,
                                                               uint8_t *__restrict__ out_sem,
                                                               float *__restrict__ out_prob)
{
    constexpr int M = KS / 2;
    constexpr int PF = 8;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < HW;
         p += (int64_t)gridDim.x * blockDim.x) {
        const float *src = prob + p;
        float win[KS];
        // slices 0..KS-2 enter the window; the first M of them pass through raw
#pragma unroll
        for (int i = 0; i < KS - 1; ++i) win[i] = src[(int64_t)i * HW];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            out_sem[(int64_t)i * HW + p] = win[i] >= thr ? 1 : 0;
            if (out_prob) out_prob[(int64_t)i * HW + p] = win[i];
        }
        float cur[PF], nxt[PF];
        const int s_end = D - M;  // filtered slices are [M, s_end)
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            int z = M + u + M;
            cur[u] = (z < D) ? src[(int64_t)z * HW] : 0.f;
        }
        for (int s0 = M; s0 < s_end; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                int z = s0 + PF + u + M;
                nxt[u] = (z < D) ? src[(int64_t)z * HW] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                int s = s0 + u;
                if (s < s_end) {
                    win[KS - 1] = cur[u];
                    float med = median_regs<KS>(win);
                    out_sem[(int64_t)s * HW + p] = med >= thr ? 1 : 0;
                    if (out_prob) out_prob[(int64_t)s * HW + p] = med;
                    // slide: the filtered value replaces the raw one (recursive filter)
                    win[M] = med;
#pragma unroll
                    for (int i = 0; i < KS - 1; ++i) win[i] = win[i + 1];
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) cur[u] = nxt[u];
        }
        // tail: win[M .. KS-2] hold the raw slices D-M .. D-1
#pragma unroll
        for (int i = 0; i < M; ++i) {
            int s = D - M + i;
            float v = win[M + i];
            out_sem[(int64_t)s * HW + p] = v >= thr ? 1 : 0;
            if (out_prob) out_prob[(int64_t)s * HW + p] = v;
        }
    }
}

// P1+P2, C > 1: same scan, one filter window per channel kept in LDS ([c][k][tid] -> conflict
// free), argmax over the filtered channels (first maximum wins, like torch.argmax).
template <int KS>
__global__ __launch_bounds__(256) void median_harden_mc_kernel(const float *__restrict__ prob, int D,
                                                               int C, int64_t HW,
                                                               uint8_t *__restrict__ out_sem,
                                                               float *__restrict__ out_prob)
{
    extern __shared__ float lds[];  // C * KS * blockDim.x
    constexpr int M = KS / 2;
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    for (int64_t p0 = (int64_t)blockIdx.x * nt; p0 < HW; p0 += (int64_t)gridDim.x * nt) {
        int64_t p = p0 + tid;
        bool live = p < HW;
        if (live) {
            for (int c = 0; c < C; ++c)
                for (int i = 0; i < KS - 1; ++i)
                    lds[(c * KS + i) * nt + tid] = prob[((int64_t)i * C + c) * HW + p];
            for (int s = 0; s < D; ++s) {
                bool filt = (s >= M) && (s < D - M);
                float best = -INFINITY;
                int arg = 0;
                for (int c = 0; c < C; ++c) {
                    float v;
                    if (filt) {
                        float w[KS];
                        // slot of slice z is z % KS; the incoming slice s+M overwrites s-M-1
                        lds[(c * KS + (s + M) % KS) * nt + tid] = prob[((int64_t)(s + M) * C + c) * HW + p];
#pragma unroll
                        for (int i = 0; i < KS; ++i) w[i] = lds[(c * KS + i) * nt + tid];
                        v = median_regs<KS>(w);
                        lds[(c * KS + s % KS) * nt + tid] = v;
                    } else {
                        v = (KS == 1) ? prob[((int64_t)s * C + c) * HW + p]
                                      : lds[(c * KS + s % KS) * nt + tid];
                    }
                    if (out_prob) out_prob[((int64_t)s * C + c) * HW + p] = v;
                    if (v > best) { best = v; arg = c; }
                }
                out_sem[(int64_t)s * HW + p] = (uint8_t)arg;
            }
        }
        __syncthreads();
    }
}

// The same for C <= MC_CMAX channels with the incoming slices PREFETCHED: the loads of the next MC_PF slices (all
// channels: MC_PF x C dword loads per lane) are in flight while the current slice is filtered -- the form above issues
// every load right before its use and ran at 0.6 TB/s on C = 5 (one HBM latency per slice and channel, 16 waves per CU).
// Filtered steps t = 0 .. D - 2M - 1 (slice s = M + t, incoming slice s + M) are unrolled by MC_PF so that the prefetch
// registers are indexed statically.
#define MC_PF 4
#define MC_CMAX 8
template <int KS>
__global__ __launch_bounds__(256) void median_harden_mc8_kernel(const float *__restrict__ prob, int D, int C,
                                                                int64_t HW, uint8_t *__restrict__ out_sem,
                                                                float *__restrict__ out_prob)
{
    extern __shared__ float lds[];  // C * KS * blockDim.x, [c][slot][tid]; a lane only touches its own column
    constexpr int M = KS / 2;
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * nt + tid; p < HW; p += (int64_t)gridDim.x * nt) {
        const float *src = prob + p;
        const int64_t cs = HW;                                    // channel stride; slice stride = C * HW
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int i = 0; i < KS - 1; ++i) lds[(c * KS + i) * nt + tid] = src[((int64_t)i * C + c) * cs];
        float nx[MC_PF][MC_CMAX];
#pragma unroll
        for (int j = 0; j < MC_PF; ++j)
#pragma unroll
            for (int c = 0; c < MC_CMAX; ++c)
                nx[j][c] = (c < C && KS - 1 + j < D) ? src[((int64_t)(KS - 1 + j) * C + c) * cs] : 0.f;
        // the first M slices pass through raw
        for (int s = 0; s < M; ++s) {
            float best = -INFINITY;
            int arg = 0;
            for (int c = 0; c < C; ++c) {
                const float v = lds[(c * KS + s) * nt + tid];
                if (out_prob) out_prob[((int64_t)s * C + c) * cs + p] = v;
                if (v > best) { best = v; arg = c; }
            }
            out_sem[(int64_t)s * HW + p] = (uint8_t)arg;
        }
        const int n_f = D - 2 * M;                                // filtered slices (D >= KS: at least one)
        int slot_in = (KS - 1) % KS, slot_s = M % KS;             // ring slots of the incoming slice and of slice s
        for (int t0 = 0; t0 < n_f; t0 += MC_PF) {
#pragma unroll
            for (int j = 0; j < MC_PF; ++j) {
                const int t = t0 + j;
                if (t < n_f) {                                     // block-uniform
                    const int s = M + t, zin = s + M;
                    float best = -INFINITY;
                    int arg = 0;
#pragma unroll
                    for (int c = 0; c < MC_CMAX; ++c) {
                        if (c < C) {
                            float w[KS];
                            lds[(c * KS + slot_in) * nt + tid] = nx[j][c];
                            nx[j][c] = (zin + MC_PF < D) ? src[((int64_t)(zin + MC_PF) * C + c) * cs] : 0.f;
#pragma unroll
                            for (int i = 0; i < KS; ++i) w[i] = lds[(c * KS + i) * nt + tid];
                            const float v = median_regs<KS>(w);
                            lds[(c * KS + slot_s) * nt + tid] = v;        // recursive: later windows see the filtered value
                            if (out_prob) out_prob[((int64_t)s * C + c) * cs + p] = v;
                            if (v > best) { best = v; arg = c; }
                        }
                    }
                    out_sem[(int64_t)s * HW + p] = (uint8_t)arg;
                    slot_in = slot_in + 1 == KS ? 0 : slot_in + 1;
                    slot_s = slot_s + 1 == KS ? 0 : slot_s + 1;
                }
            }
        }
        // the last M slices pass through raw
        for (int s = D - M; s < D; ++s) {
            float best = -INFINITY;
            int arg = 0;
            for (int c = 0; c < C; ++c) {
                const float v = lds[(c * KS + s % KS) * nt + tid];
                if (out_prob) out_prob[((int64_t)s * C + c) * cs + p] = v;
                if (v > best) { best = v; arg = c; }
            }
            out_sem[(int64_t)s * HW + p] = (uint8_t)arg;
        }
    }
}

template <int KS>
static int launch_median(const float *prob, int D, int C, int64_t HW, float thr, uint8_t *out_sem,
                         float *out_prob, hipStream_t st)
{
    const int block = 256;
    int grid = emp_grid(HW, block, 8192);
    if (C == 1) {
        hipLaunchKernelGGL(median_harden_c1_kernel<KS>, dim3(grid), dim3(block), 0, st, prob, D, HW, thr,
                           out_sem, out_prob);
    } else {
        size_t lds = (size_t)C * KS * block * sizeof(float);
        if (lds > 160 * 1024) EMP_FAIL(EMP_EINVAL, "median: C*ks too large for LDS (%d x %d)", C, KS);
        if (KS > 1 && C <= MC_CMAX) {
            if (lds > 64 * 1024)
                hipFuncSetAttribute((const void *)median_harden_mc8_kernel<KS>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(median_harden_mc8_kernel<KS>, dim3(grid), dim3(block), lds, st, prob, D, C, HW,
                               out_sem, out_prob);
        } else {
            if (lds > 64 * 1024)
                hipFuncSetAttribute((const void *)median_harden_mc_kernel<KS>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(median_harden_mc_kernel<KS>, dim3(grid), dim3(block), lds, st, prob, D, C, HW,
                               out_sem, out_prob);
        }
    }
    EMP_CHECK_LAUNCH("emp_median_harden_stack");
    return EMP_OK;
}

extern "C" int emp_median_harden_stack(const float *prob, int D, int C, int64_t HW, int ks, float thr,
                                       uint8_t *out_sem, float *out_prob, void *stream)
{
    EMP_REQUIRE(prob && out_sem, "median: null pointer");
    EMP_REQUIRE(D >= 0 && HW >= 0, "median: negative size");
    EMP_REQUIRE(C >= 1 && C <= EMP_MAX_CLASSES, "median: C=%d out of range", C);
    EMP_REQUIRE(ks >= 1 && ks <= EMP_MAX_KS && (ks & 1), "median: ks=%d must be odd in 1..%d", ks, EMP_MAX_KS);
    EMP_REQUIRE(ks == 1 || D >= ks, "median: stack shorter than ks (D=%d, ks=%d): host must degrade", D, ks);
    if (D == 0 || HW == 0) return EMP_OK;
    hipStream_t st = emp_stream(stream);
    switch (ks) {
        case 1: return launch_median<1>(prob, D, C, HW, thr, out_sem, out_prob, st);
        case 3: return launch_median<3>(prob, D, C, HW, thr, out_sem, out_prob, st);
        case 5: return launch_median<5>(prob, D, C, HW, thr, out_sem, out_prob, st);
        case 7: return launch_median<7>(prob, D, C, HW, thr, out_sem, out_prob, st);
        case 9: return launch_median<9>(prob, D, C, HW, thr, out_sem, out_prob, st);
        default: return launch_median<11>(prob, D, C, HW, thr, out_sem, out_prob, st);
    }
}

extern "C" int emp_harden(const float *prob, int D, int C, int64_t HW, float thr, uint8_t *out_sem,
                          void *stream)
{
    return emp_median_harden_stack(prob, D, C, HW, 1, thr, out_sem, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// streaming median step: ks slice pointers by value
struct SlicePtrs { const float *p[EMP_MAX_KS]; };

template <int KS>
__global__ __launch_bounds__(256) void median_step_kernel(SlicePtrs sp, int64_t n, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float w[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) w[k] = sp.p[k][i];
        out[i] = median_regs<KS>(w);
    }
}

extern "C" int emp_median_step(const float *const *slices_host, int ks, int64_t n, float *out, void *stream)
{
    EMP_REQUIRE(slices_host && out, "median_step: null pointer");
    EMP_REQUIRE(ks >= 1 && ks <= EMP_MAX_KS && (ks & 1), "median_step: ks=%d must be odd in 1..%d", ks, EMP_MAX_KS);
    if (n <= 0) return EMP_OK;
    SlicePtrs sp;
    for (int k = 0; k < EMP_MAX_KS; ++k) sp.p[k] = slices_host[k < ks ? k : 0];
    hipStream_t st = emp_stream(stream);
    int grid = emp_grid(n, 256, 4096);
#define EMP_MS(K) hipLaunchKernelGGL(median_step_kernel<K>, dim3(grid), dim3(256), 0, st, sp, n, out)
    switch (ks) {
        case 1: EMP_MS(1); break;
        case 3: EMP_MS(3); break;
        case 5: EMP_MS(5); break;
        case 7: EMP_MS(7); break;
        case 9: EMP_MS(9); break;
        default: EMP_MS(11); break;
    }
#undef EMP_MS
    EMP_CHECK_LAUNCH("emp_median_step");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// P3: centre NMS, streaming form.  A centre must exceed the threshold and be a maximum of its k x k
// window, so it is in particular >= its in-row neighbours.  Each lane streams 4 pixels (one 16-byte
// load, 1 KiB per wave instruction), tests that cheap necessary condition in registers, and only the
// few survivors (a handful per object) check the full window straight from L2/L1.  Traffic is the
// algorithmic 4 B/pixel; the windowed maximum is never materialised.
#define CT_MAXK 15

__device__ __forceinline__ float thresholded(float v, float thr) { return (v > thr) ? v : -1.0f; }

// full k x k test for the rare survivors: no early exit, so the loads of a row are independent and overlap
__device__ __forceinline__ bool is_window_max(const float *__restrict__ img, int h, int w, int y, int x, float v,
                                              float thr, int k)
{
    const int pad = k / 2;
    float m = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
        int yy = y - pad + dy;
        if (yy < 0 || yy >= h) continue;
        const float *row = img + (int64_t)yy * w;
        for (int dx = 0; dx < k; ++dx) {
            int xx = x - pad + dx;
            if (xx >= 0 && xx < w) m = fmaxf(m, thresholded(row[xx], thr));
        }
    }
    return !(m > v);
}

template <int VEC>
__global__ __launch_bounds__(256) void find_centers_kernel(const float *__restrict__ hmp, int h, int w, float thr,
                                                           int k, int cap, int32_t *__restrict__ out_idx,
                                                           int32_t *__restrict__ out_count)
{
    const int d = blockIdx.y;
    const float *img = hmp + (int64_t)d * h * w;
    const uint32_t hw = (uint32_t)h * (uint32_t)w;                  // < 2^31 (checked by the entry point)
    const int lane = threadIdx.x & 63;
    const bool need_left = (k / 2) >= 1, need_right = (k - 1 - k / 2) >= 1;
    // wave-uniform loop bounds: all 64 lanes take part in the shuffles of every iteration.  Each lane issues
    // FC_U independent 16-byte loads before touching any of them (4 KiB in flight per wave); the two values a wave
    // cannot get from a neighbouring lane (left of lane 0, right of lane 63) are requested in the same phase, so that
    // no load sits between the streaming loads and their use.
    constexpr int FC_U = 4;
    const uint32_t stride = gridDim.x * blockDim.x * VEC;
    const uint32_t wave_q0 = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * VEC;
    // pixel coordinates advance by a fixed step from load to load: one division per lane for the whole kernel (the
    // kernel was bound by its vector instructions -- ~37 per pixel, a wave64 instruction takes 4 cycles --, not by HBM)
    const int step_y = (int)(stride / (uint32_t)w), step_x = (int)(stride % (uint32_t)w);
    int y_run = (int)((wave_q0 + (uint32_t)lane * VEC) / (uint32_t)w);
    int x_run = (int)((wave_q0 + (uint32_t)lane * VEC) % (uint32_t)w);
    const float nl = need_left ? 0.f : -INFINITY, nr = need_right ? 0.f : -INFINITY;   // + 0 keeps, + -inf drops a side
    for (uint64_t base0 = wave_q0; base0 < hw; base0 += (uint64_t)stride * FC_U) {
        float vv[FC_U][VEC], edge[FC_U];
        int yy[FC_U], xx[FC_U];
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const uint64_t q64 = base0 + (uint64_t)u * stride + (uint32_t)lane * VEC;
            const bool live = q64 < hw;
            const uint32_t q = live ? (uint32_t)q64 : 0u;
            if (VEC == 4) {
                float4 f = live ? *reinterpret_cast<const float4 *>(img + q) : make_float4(-1.f, -1.f, -1.f, -1.f);
                vv[u][0] = f.x; vv[u][1] = f.y; vv[u][2] = f.z; vv[u][3] = f.w;
            } else {
                vv[u][0] = live ? img[q] : -1.f;
            }
            const int y = y_run, x = x_run;
            yy[u] = y; xx[u] = x;
            x_run += step_x; y_run += step_y;
            if (x_run >= w) { x_run -= w; ++y_run; }
            edge[u] = -INFINITY;
            if (lane == 0 && live && x > 0) edge[u] = img[q - 1];
            if (lane == 63 && live && x + VEC < w) edge[u] = img[q + VEC];
        }
        // phase 1, registers only: pixels above the threshold that are >= their in-row neighbours -- the ridge of every
        // blob, one pixel per blob row.  For a pixel c > thr, c > 0 the reference's comparison with a neighbour n,
        // thresholded(n) > c, is n > c (n > c > thr passes the threshold; n <= thr compares as -1, never above c > 0).
        unsigned ridge = 0;
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const uint64_t base = base0 + (uint64_t)u * stride;
            if (base >= hw) break;                                   // wave-uniform; dead lanes hold -1: never a ridge
            const int x = xx[u];
            float v[VEC + 2];
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j + 1] = vv[u][j];
            float left = __shfl_up(v[VEC], 1), right = __shfl_down(v[1], 1);
            // neighbours across the wave edge come from memory, neighbours across the row edge do not exist
            if (lane == 0) left = edge[u];
            if (lane == 63) right = edge[u];
            if (x == 0) left = -INFINITY;
            if (x + VEC >= w) right = -INFINITY;
            v[0] = left;
            v[VEC + 1] = right;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float c = v[j + 1];
                if (c > thr && c > 0.0f && !(v[j] + nl > c) && !(v[j + 2] + nr > c)) ridge |= 1u << (u * VEC + j);
            }
        }
        if (!__ballot(ridge != 0)) continue;
        // phase 2: the pixels straight above and below (in the window under the same conditions as left / right) leave
        // the summit of a blob only.  All loads of the phase are issued before the first comparison (rows other waves
        // are streaming: L2 hits)
        float up_v[FC_U * VEC], dn_v[FC_U * VEC];
#pragma unroll
        for (int u = 0; u < FC_U; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int b = u * VEC + j;
                up_v[b] = dn_v[b] = -INFINITY;
                if ((ridge >> b) & 1u) {
                    const uint32_t q = (uint32_t)yy[u] * (uint32_t)w + (uint32_t)(xx[u] + j);
                    if (need_left && yy[u] > 0) up_v[b] = img[q - w];
                    if (need_right && yy[u] + 1 < h) dn_v[b] = img[q + w];
                }
            }
#pragma unroll
        for (int u = 0; u < FC_U; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int b = u * VEC + j;
                if (((ridge >> b) & 1u) && (up_v[b] > vv[u][j] || dn_v[b] > vv[u][j])) ridge &= ~(1u << b);
            }
        // phase 3: the survivors are settled by the whole wave: the k x k window is fetched by k*k lanes at once and
        // reduced with cross-lane maxima
#pragma unroll
        for (int u = 0; u < FC_U; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int b = u * VEC + j;
                unsigned long long todo = __ballot((ridge >> b) & 1u);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int cy = __shfl(yy[u], src), cx = __shfl(xx[u], src) + j;
                    const float cv = __shfl(vv[u][j], src);
                    float m = -INFINITY;
                    for (int t = lane; t < k * k; t += 64) {
                        int y2 = cy - k / 2 + t / k, x2 = cx - k / 2 + t % k;
                        if (y2 >= 0 && y2 < h && x2 >= 0 && x2 < w) m = fmaxf(m, thresholded(img[(int64_t)y2 * w + x2], thr));
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                    if (lane == src && !(m > cv)) {
                        int slot = atomicAdd(&out_count[d], 1);
                        if (slot < cap) out_idx[(int64_t)d * cap + slot] = cy * w + cx;
                    }
                }
            }
    }
}

// per-slice bitonic sort of the (few) centre indices -> raster order
__global__ __launch_bounds__(256) void sort_centers_kernel(int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ count, int cap)
{
    __shared__ int32_t buf[EMP_MAX_CENTERS];
    const int d = blockIdx.x;
    int n = count[d];
    if (n > cap) n = cap;
    if (n <= 1) return;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    int32_t *g = idx + (int64_t)d * cap;
    for (int i = threadIdx.x; i < np2; i += blockDim.x) buf[i] = (i < n) ? g[i] : 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < np2 / 2; i += blockDim.x) {
                int lo = (i / stride) * 2 * stride + (i % stride);
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                int32_t a = buf[lo], b = buf[hi];
                if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) g[i] = buf[i];
}

extern "C" int emp_find_centers(const float *hmp, int D, int h, int w, float thr, int k, int cap,
                                int32_t *out_idx, int32_t *out_count, void *stream)
{
    EMP_REQUIRE(hmp && out_idx && out_count, "find_centers: null pointer");
    EMP_REQUIRE(k >= 1 && k <= CT_MAXK, "find_centers: nms kernel %d not in 1..%d", k, CT_MAXK);
    EMP_REQUIRE(cap >= 1 && cap <= EMP_MAX_CENTERS, "find_centers: cap %d not in 1..%d", cap, EMP_MAX_CENTERS);
    EMP_REQUIRE(D >= 0 && D <= 65535 && h > 0 && w > 0, "find_centers: bad shape D=%d h=%d w=%d", D, h, w);
    EMP_REQUIRE((int64_t)h * w < (1LL << 31), "find_centers: slice too large");
    if (D == 0) return EMP_OK;
    hipStream_t st = emp_stream(stream);
    if (hipMemsetAsync(out_count, 0, sizeof(int32_t) * D, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "find_centers: memset failed");
    const bool vec4 = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(hmp) & 15) == 0);
    const int64_t hw = (int64_t)h * w;
    if (vec4) {
        // 64 float4 per lane = 16 trips of the 4-load loop: few, long-lived blocks (one trip per lane: 1.52 ms per
        // 1024^3 plane, 16 trips: 1.27 ms)
        int gx = emp_grid(emp_cdiv(hw / 4, 64), 256, 512);
        hipLaunchKernelGGL(find_centers_kernel<4>, dim3(gx, D), dim3(256), 0, st, hmp, h, w, thr, k, cap, out_idx,
                           out_count);
    } else {
        int gx = emp_grid(emp_cdiv(hw, 4), 256, 1024);
        hipLaunchKernelGGL(find_centers_kernel<1>, dim3(gx, D), dim3(256), 0, st, hmp, h, w, thr, k, cap, out_idx,
                           out_count);
    }
    EMP_CHECK_LAUNCH("emp_find_centers");
    hipLaunchKernelGGL(sort_centers_kernel, dim3(D), dim3(256), 0, st, out_idx, out_count, cap);
    EMP_CHECK_LAUNCH("emp_find_centers(sort)");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// P4: nearest-centre vote.  Block = GP_THREADS threads x GP_PER_THREAD pixels of one slice; the slice's centres
// (pre-scaled float2, padded with +inf) are read with wave-uniform indices, i.e. as scalar loads.  Exact rounding
// contract (see emp_hip.h):
//   s = fmaf(dx, dx, fl(dy*dy)); d = sqrt_rn(s); first strictly smaller d wins.
// sqrt is monotone, so the comparison runs on s and d is only evaluated for near ties.
// Per WAVE (64 consecutive voted pixels = one or two objects) the centre list is pruned by the bounding box of the voted
// locations before the exact vote (see phase 2): ~200 candidates per pixel at 1024^2 become a handful.
// Work compaction: a block owns GP_TILE consecutive pixels.  Pixels that need a vote (all of them without a
// semantic map, only thing pixels with one -- ~10 % of an EM slice) are compacted into an LDS list so that the
// K-centre loop runs with full lanes; ids are staged in LDS and written back as one coalesced 16-byte store
// per lane.  Traffic: 1 B (class) + 8 B (offsets, voted pixels only) read, 2 B written per pixel.
#define GP_BATCH 8
#define GP_CAND 248                // per-wave candidate list (entries); more survivors -> full scan
#define GP_PRUNE_MIN 16            // slices with at most this many centres are scanned in full
#define GP_THREADS 128
#define GP_TILE 1024
#define GP_TW 64                   // a block's patch: GP_TH rows x GP_TW columns = GP_TILE pixels
#define GP_TH (GP_TILE / GP_TW)
#define GP_PER_THREAD (GP_TILE / GP_THREADS)

// correctly rounded fp32 square root: the double-precision root of a float rounds to the correctly rounded float
// root (53 >= 2*24 + 2), independent of how v_sqrt_f32 is refined by the compiler flags in use
__device__ __forceinline__ float sqrt_rn_exact(float s) { return (float)__dsqrt_rn((double)s); }

// centres of every slice as fp32 (step*y, step*x): step * ctr is int64 * python float -> fp32 (postprocess.py:151)
__global__ void group_centers_kernel(const int32_t *__restrict__ ctr_idx, const int32_t *__restrict__ ctr_count,
                                     int cap, int w, int step, int D, float2 *__restrict__ ctr_f)
{
    const float fstep = (float)step;
    // slice d owns ctr_f[d*stride .. d*stride + cap + 2*GP_BATCH); entries past its K are +inf (never the nearest)
    const int stride = cap + 2 * GP_BATCH;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)D * stride;
         i += (int64_t)gridDim.x * blockDim.x) {
        int d = (int)(i / stride), k = (int)(i % stride);
        int n = ctr_count[d] < cap ? ctr_count[d] : cap;
        float2 v = make_float2(INFINITY, INFINITY);
        if (k < n) {
            int f = ctr_idx[(int64_t)d * cap + k];
            v = make_float2(__fmul_rn(fstep, (float)(f / w)), __fmul_rn(fstep, (float)(f % w)));
        }
        ctr_f[i] = v;
    }
}

__global__ __launch_bounds__(GP_THREADS) void group_pixels_kernel(const float2 *__restrict__ ctr_f,
                                                           const int32_t *__restrict__ ctr_count, int cap,
                                                           const float *__restrict__ offsets, int h, int w,
                                                           int step, const uint8_t *__restrict__ sem,
                                                           uint32_t thing_mask, uint16_t *__restrict__ out_ids)
{
    __shared__ uint16_t todo[GP_TILE];
    __shared__ __attribute__((aligned(16))) uint16_t ids_tile[GP_TILE];
    __shared__ int n_todo;
    const int d = blockIdx.y;
    int K = ctr_count[d];
    if (K > cap) K = cap;
    const int64_t hw = (int64_t)h * w;
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    // the slice's centres are read with a wave-uniform index: scalar loads, operands straight from SGPRs
    const float2 *__restrict__ ctr = ctr_f + (int64_t)d * (cap + 2 * GP_BATCH);
    const float *offy = offsets + (int64_t)d * 2 * hw;
    const float *offx = offy + hw;
    uint16_t *out = out_ids + (int64_t)d * hw;
    const uint8_t *sm = sem ? sem + (int64_t)d * hw : nullptr;
    // a block owns a GP_TH x GP_TW patch of the slice (not GP_TILE consecutive pixels of a row): its voted pixels belong
    // to one or two objects, which is what makes the per-wave pruning of phase 2 bite
    const int tiles_x = (w + GP_TW - 1) / GP_TW;
    const int ty0 = (int)(blockIdx.x / tiles_x) * GP_TH, tx0 = (int)(blockIdx.x % tiles_x) * GP_TW;
    const float sinit = (K > 20) ? 1e10f : INFINITY;   // sqrt_rn(1e10f) == 1e5f exactly
    const int idinit = (K > 20 || K == 0) ? 0 : 1;

    // phase 1: which pixels of the tile are voted on
    const int l0 = threadIdx.x * GP_PER_THREAD;        // local index l = row * GP_TW + column inside the patch
    const int y_t = ty0 + l0 / GP_TW, x_t = tx0 + l0 % GP_TW;          // the thread's GP_PER_THREAD pixels: one row
    const int64_t p0 = (int64_t)y_t * w + x_t;
    const bool row_ok = y_t < h;
    unsigned wantbits = 0;
    {
        if (sm && row_ok && x_t + GP_PER_THREAD <= w && ((reinterpret_cast<uintptr_t>(sm + p0) & 7) == 0)) {
            uint2 c8 = *reinterpret_cast<const uint2 *>(sm + p0);
            unsigned long long bytes = ((unsigned long long)c8.y << 32) | c8.x;
#pragma unroll
            for (int j = 0; j < GP_PER_THREAD; ++j)
                if ((thing_mask >> ((bytes >> (8 * j)) & 0xff)) & 1u) wantbits |= 1u << j;
        } else {
#pragma unroll
            for (int j = 0; j < GP_PER_THREAD; ++j)
                if (row_ok && x_t + j < w && (!sm || ((thing_mask >> sm[p0 + j]) & 1u))) wantbits |= 1u << j;
        }
    }
    if (K == 0) wantbits = 0;
#pragma unroll
    for (int j = 0; j < GP_PER_THREAD; ++j) ids_tile[l0 + j] = 0;
    // compaction slots: wave prefix sum of the per-lane counts, ONE LDS atomic per wave (a returning atomic per
    // lane on one address serialises the whole block)
    int mine = __popc(wantbits);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if ((threadIdx.x & 63) >= o) incl += t;
    }
    int wave_total = __shfl(incl, 63);
    int wave_base = 0;
    if ((threadIdx.x & 63) == 63 && wave_total) wave_base = atomicAdd(&n_todo, wave_total);
    wave_base = __shfl(wave_base, 63);
    int slot = wave_base + incl - mine;
#pragma unroll
    for (int j = 0; j < GP_PER_THREAD; ++j)
        if ((wantbits >> j) & 1u) todo[slot++] = (uint16_t)(l0 + j);
    __syncthreads();

    // phase 2: nearest-centre vote over the compacted list, one wave = 64 consecutive voted pixels at a time.
    // Those 64 pixels belong to one or two objects of one or two image rows, so the locations they vote for
    // (pixel + offset) fall into a small box around one or two centres.  Per wave (K > GP_PRUNE_MIN): the box of the
    // wave's locations; U = the smallest over all centres of the LARGEST distance to the box (that centre is within U
    // of every location); a centre whose SMALLEST distance to the box exceeds U -- by a relative 2e-4 in the squared
    // distance, three orders of magnitude more than any rounding or root tie -- can be the nearest of no pixel of the
    // wave and cannot tie with one.  The survivors (typically 1-4 of ~200 at 1024^2) keep their index order in a
    // per-wave LDS list, and the exact vote below runs over them only.  Locations with NaN / inf widen the box to
    // "keep everything"; a list that overflows falls back to the full scan.  (Round 2 tried this per BLOCK -- two
    // image rows, a box spanning the slice -- and pruned nothing.)
    __shared__ uint16_t cand[GP_THREADS / 64][GP_CAND + GP_BATCH];
    const int n = n_todo;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int base = wave * 64; base < n; base += GP_THREADS) {           // wave-uniform trip count
        const int i = base + lane;
        const bool act = i < n;
        const int l = act ? todo[i] : 0;
        const int y = ty0 + l / GP_TW, x = tx0 + l % GP_TW;
        const int64_t p = (int64_t)y * w + x;
        const float ly = act ? __fadd_rn((float)(y * step), offy[p]) : 0.f;  // coord + offsets
        const float lx = act ? __fadd_rn((float)(x * step), offx[p]) : 0.f;
        // candidate list of this wave-step: all centres 0..K-1, or the survivors of the box test
        int n_cand = -1;                                                  // -1: scan all K centres
        if (K > GP_PRUNE_MIN) {
            float ymin = act ? ly : INFINITY, ymax = act ? ly : -INFINITY;
            float xmin = act ? lx : INFINITY, xmax = act ? lx : -INFINITY;
            bool wild = act && !(fabsf(ly) < 3e38f && fabsf(lx) < 3e38f);  // NaN / inf location in this lane
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
                xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
            }
            if (!__ballot(wild)) {
                float u2 = INFINITY;
                for (int k = lane; k < K; k += 64) {
                    const float2 c = ctr[k];
                    const float dyh = fmaxf(fabsf(c.x - ymin), fabsf(c.x - ymax));
                    const float dxh = fmaxf(fabsf(c.y - xmin), fabsf(c.y - xmax));
                    u2 = fminf(u2, dyh * dyh + dxh * dxh);
                }
#pragma unroll
                for (int o = 32; o; o >>= 1) u2 = fminf(u2, __shfl_xor(u2, o));
                u2 *= 1.0001f;
                int cnt = 0;
                for (int k0 = 0; k0 < K; k0 += 64) {
                    const int k = k0 + lane;
                    bool keep = false;
                    if (k < K) {
                        const float2 c = ctr[k];
                        const float dyl = fmaxf(fmaxf(ymin - c.x, c.x - ymax), 0.f);
                        const float dxl = fmaxf(fmaxf(xmin - c.y, c.y - xmax), 0.f);
                        keep = (dyl * dyl + dxl * dxl) * 0.9999f <= u2;
                    }
                    const uint64_t m = __ballot(keep);
                    const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (keep && pos < GP_CAND) cand[wave][pos] = (uint16_t)k;
                    cnt += __popcll(m);
                }
                if (cnt <= GP_CAND) {
                    if (lane < GP_BATCH) cand[wave][cnt + lane] = (uint16_t)K;      // padding: centre K is +inf
                    n_cand = cnt;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // First index with the strictly smallest d = sqrt_rn(s).  sqrt_rn is monotone, and two floats whose ratio
        // exceeds 1 + 2^-21 cannot round to the same square root, so the comparison of d reduces to a comparison
        // of s except in a narrow near-tie band, where both roots are evaluated exactly.  sbest is the s of the
        // current winner (a virtual candidate with d = 1e5, i.e. s = 1e10 exactly, when K > 20).
        float sbest = sinit;
        // sbest * (1 - 2^-20): at or below it, s is smaller beyond any tie of the rounded roots
        float sclear = __fmul_rn(sinit, 0.99999904632568359375f);
        int id = idinit;
        // Centres past K are padded with +inf (s = inf never wins).  Each batch is first run branch-free under the
        // assumption "a near tie does not win"; if any lane of the wave met a near tie (rare), the batch is redone
        // from the saved state with the exact comparison of the correctly rounded roots.
        const int n_scan = n_cand < 0 ? K : n_cand;
        for (int k0 = 0; k0 < n_scan; k0 += GP_BATCH) {
            float2 c[GP_BATCH];
            int cid[GP_BATCH];
#pragma unroll
            for (int j = 0; j < GP_BATCH; ++j) {
                cid[j] = n_cand < 0 ? k0 + j : (int)((volatile uint16_t *)cand[wave])[k0 + j];
                c[j] = ctr[cid[j]];
            }
            const float sbest0 = sbest, sclear0 = sclear;
            const int id0 = id;
            bool any_near = false;
#pragma unroll
            for (int j = 0; j < GP_BATCH; ++j) {
                float dy = __fsub_rn(c[j].x, ly);
                float dx = __fsub_rn(c[j].y, lx);
                float s2 = __fmaf_rn(dx, dx, __fmul_rn(dy, dy));
                bool lt = s2 < sbest;
                bool better = lt && s2 <= sclear;      // "lt" matters when sclear == sbest (zero / denormal sbest)
                any_near = any_near || (lt && !better);
                sbest = better ? s2 : sbest;
                sclear = better ? __fmul_rn(s2, 0.99999904632568359375f) : sclear;
                id = better ? cid[j] + 1 : id;
            }
            if (__ballot(any_near)) {
                if (any_near) {
                    sbest = sbest0; sclear = sclear0; id = id0;
#pragma unroll
                    for (int j = 0; j < GP_BATCH; ++j) {
                        float dy = __fsub_rn(c[j].x, ly);
                        float dx = __fsub_rn(c[j].y, lx);
                        float s2 = __fmaf_rn(dx, dx, __fmul_rn(dy, dy));
                        bool lt = s2 < sbest;
                        bool better = lt && (s2 <= sclear || sqrt_rn_exact(s2) < sqrt_rn_exact(sbest));
                        sbest = better ? s2 : sbest;
                        sclear = better ? __fmul_rn(s2, 0.99999904632568359375f) : sclear;
                        id = better ? cid[j] + 1 : id;
                    }
                }
            }
        }
        if (act) ids_tile[l] = (uint16_t)id;
        __builtin_amdgcn_wave_barrier();                                  // the next step rewrites this wave's list
    }
    __syncthreads();

    // phase 3: write-back (8 ids = 16 bytes per lane, 128-byte row segments)
    if (row_ok) {
        if (x_t + GP_PER_THREAD <= w && ((reinterpret_cast<uintptr_t>(out + p0) & 15) == 0)) {
            *reinterpret_cast<uint4 *>(out + p0) = *reinterpret_cast<const uint4 *>(&ids_tile[l0]);
        } else {
#pragma unroll
            for (int j = 0; j < GP_PER_THREAD; ++j)
                if (x_t + j < w) out[p0 + j] = ids_tile[l0 + j];
        }
    }
}

extern "C" int64_t emp_group_work_elems(int D, int cap) { return 2 * (int64_t)(D > 0 ? D : 1) * (cap + 2 * GP_BATCH); }

extern "C" int emp_group_pixels(const int32_t *ctr_idx, const int32_t *ctr_count, int cap,
                                const float *offsets, int D, int h, int w, int step, const uint8_t *sem,
                                uint32_t thing_mask, float *work, uint16_t *out_ids, void *stream)
{
    EMP_REQUIRE(ctr_idx && ctr_count && offsets && out_ids && work, "group_pixels: null pointer");
    EMP_REQUIRE((reinterpret_cast<uintptr_t>(work) & 7) == 0, "group_pixels: work must be 8-byte aligned");
    EMP_REQUIRE(cap >= 1 && cap <= EMP_MAX_CENTERS, "group_pixels: cap %d not in 1..%d", cap, EMP_MAX_CENTERS);
    EMP_REQUIRE(step == 1 || step == 4, "group_pixels: step must be 1 or 4");
    EMP_REQUIRE(D >= 0 && D <= 65535 && h > 0 && w > 0, "group_pixels: bad shape");
    EMP_REQUIRE((int64_t)h * step < (1 << 24) && (int64_t)w * step < (1 << 24), "group_pixels: coords exceed fp32 integers");
    if (D == 0) return EMP_OK;
    int gx = (int)(emp_cdiv(h, GP_TH) * emp_cdiv(w, GP_TW));
    float2 *ctr_f = reinterpret_cast<float2 *>(work);
    hipLaunchKernelGGL(group_centers_kernel, dim3(emp_grid((int64_t)D * (cap + 2 * GP_BATCH), 256, 1024)), dim3(256), 0,
                       emp_stream(stream), ctr_idx, ctr_count, cap, w, step, D, ctr_f);
    hipLaunchKernelGGL(group_pixels_kernel, dim3(gx, D), dim3(GP_THREADS), 0, emp_stream(stream), ctr_f, ctr_count,
                       cap, offsets, h, w, step, sem, thing_mask, out_ids);
    EMP_CHECK_LAUNCH("emp_group_pixels");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// P4b + P5: fuse semantic classes with instance cells.
// work layout (int32): hist[D][cap+1][NC] | stuff[D][NC] | lut[D][cap+1] (pan value as u32) |
//                      stuff_ok[D][NC]
struct FuseLayout {
    int64_t hist, stuff, lut, ok, total;
};
static inline FuseLayout fuse_layout(int D, int cap, int nc)
{
    FuseLayout L;
    L.hist = 0;
    L.stuff = L.hist + (int64_t)D * (cap + 1) * nc;
    L.lut = L.stuff + (int64_t)D * nc;
    L.ok = L.lut + (int64_t)D * (cap + 1);
    L.total = L.ok + (int64_t)D * nc;
    return L;
}
extern "C" int64_t emp_fuse_work_elems(int D, int cap, int n_classes)
{
    return fuse_layout(D, cap, n_classes).total;
}

// wave-aggregated histogram increment: lanes holding the same key as their left neighbour are
// folded into the run head, which adds the run length with one atomic.
__device__ __forceinline__ void wave_hist_add(int32_t *base, int64_t key, bool valid)
{
    const int lane = threadIdx.x & 63;
    int64_t prev = __shfl_up(key, 1);
    bool pvalid = __shfl_up((int)valid, 1) != 0;
    bool head = valid && (lane == 0 || !pvalid || prev != key);
    unsigned long long heads = __ballot(head);
    unsigned long long valids = __ballot(valid);
    if (head) {
        // run ends at the next head or the first invalid lane after this one
        unsigned long long stop = (heads | ~valids) & ~((2ULL << lane) - 1ULL);
        int end = stop ? __ffsll((long long)stop) - 1 : 64;
        atomicAdd(base + key, end - lane);
    }
}

// Histogram pass.  Each block owns a contiguous span of one slice and accumulates into an LDS
// histogram ((cap+1)*nc instance/class bins + nc stuff bins) that is flushed once with global atomics
// (non-zero bins only); without LDS (huge cap) it adds to the global bins directly.
__global__ __launch_bounds__(256) void fuse_hist_kernel(const uint8_t *__restrict__ sem,
                                                        const uint16_t *__restrict__ ids, int H, int W, int up,
                                                        int cap, int nc, uint32_t thing_mask,
                                                        int32_t *__restrict__ hist, int32_t *__restrict__ stuff,
                                                        int use_lds)
{
    extern __shared__ int32_t lh[];
    const int d = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int w = W / up;
    const uint8_t *s = sem + (int64_t)d * HW;
    const uint16_t *g = ids + (int64_t)d * (H / up) * w;
    int32_t *hh = hist + (int64_t)d * (cap + 1) * nc;
    int32_t *ss = stuff + (int64_t)d * nc;
    const int n_hist = (cap + 1) * nc;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_hist + nc; i += blockDim.x) lh[i] = 0;
        __syncthreads();
    }
    int32_t *bh = use_lds ? lh : hh;
    int32_t *bs = use_lds ? lh + n_hist : ss;
    // contiguous span per block; whole waves iterate together so that the ballots see uniform control flow
    const int64_t span = (HW + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * span;
    const int64_t p1 = (p0 + span < HW) ? p0 + span : HW;
    for (int64_t q = p0; q < p1; q += blockDim.x) {
        int64_t p = q + threadIdx.x;
        bool live = p < p1;
        int c = 0, ins = 0;
        if (live) {
            c = s[p];
            if (c >= nc) c = nc - 1;
            int y = (int)(p / W), x = (int)(p % W);
            int id = g[(int64_t)(y / up) * w + x / up];
            ins = ((thing_mask >> c) & 1u) ? id : 0;
        }
        wave_hist_add(bh, (int64_t)ins * nc + c, live && ins > 0);
        wave_hist_add(bs, (int64_t)c, live && ins == 0);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_hist; i += blockDim.x) {
            int v = lh[i];
            if (v) atomicAdd(hh + i, v);
        }
        if (threadIdx.x < nc) {
            int v = lh[n_hist + threadIdx.x];
            if (v) atomicAdd(ss + threadIdx.x, v);
        }
    }
}

// one block per slice: majority class per instance, per-class renumbering (block scan), stuff areas
__global__ __launch_bounds__(256) void fuse_lut_kernel(int cap, int nc, uint32_t thing_mask, int64_t div,
                                                       int64_t stuff_area, const int32_t *__restrict__ hist,
                                                       const int32_t *__restrict__ stuff,
                                                       int32_t *__restrict__ lut, int32_t *__restrict__ ok)
{
    __shared__ int32_t scan[256];
    __shared__ int32_t carry[EMP_MAX_CLASSES];
    const int d = blockIdx.x;
    const int32_t *hh = hist + (int64_t)d * (cap + 1) * nc;
    uint32_t *ll = reinterpret_cast<uint32_t *>(lut) + (int64_t)d * (cap + 1);
    if (threadIdx.x < EMP_MAX_CLASSES) carry[threadIdx.x] = 0;
    if (threadIdx.x < nc) {
        bool thing = (thing_mask >> threadIdx.x) & 1u;
        ok[(int64_t)d * nc + threadIdx.x] = (!thing && stuff[(int64_t)d * nc + threadIdx.x] >= stuff_area) ? 1 : 0;
    }
    __syncthreads();
    for (int base = 1; base <= cap; base += blockDim.x) {
        int id = base + threadIdx.x;
        int mode = -1;
        if (id <= cap) {
            int best = 0;
            for (int c = 0; c < nc; ++c) {
                int n = hh[(int64_t)id * nc + c];
                if (n > best) { best = n; mode = c; }  // ties -> smallest class (torch.mode)
            }
        }
        for (int c = 0; c < nc; ++c) {
            if (!((thing_mask >> c) & 1u)) continue;
            int flag = (mode == c) ? 1 : 0;
            scan[threadIdx.x] = flag;
            __syncthreads();
            for (int off = 1; off < blockDim.x; off <<= 1) {
                int v = (threadIdx.x >= off) ? scan[threadIdx.x - off] : 0;
                __syncthreads();
                scan[threadIdx.x] += v;
                __syncthreads();
            }
            int incl = scan[threadIdx.x];
            int total = scan[blockDim.x - 1];
            if (flag) ll[id] = (uint32_t)((int64_t)c * div + carry[c] + incl);
            __syncthreads();
            if (threadIdx.x == 0) carry[c] += total;
            __syncthreads();
        }
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void fuse_apply_kernel(const uint8_t *__restrict__ sem,
                                                         const uint16_t *__restrict__ ids, int H, int W, int up,
                                                         int cap, int nc, uint32_t thing_mask, int64_t div,
                                                         int64_t void_label, const int32_t *__restrict__ lut,
                                                         const int32_t *__restrict__ ok, OutT *__restrict__ out)
{
    const int d = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int w = W / up;
    const uint8_t *s = sem + (int64_t)d * HW;
    const uint16_t *g = ids + (int64_t)d * (H / up) * w;
    const uint32_t *ll = reinterpret_cast<const uint32_t *>(lut) + (int64_t)d * (cap + 1);
    const int32_t *oo = ok + (int64_t)d * nc;
    OutT *o = out + (int64_t)d * HW;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < HW;
         p += (int64_t)gridDim.x * blockDim.x) {
        int c = s[p];
        if (c >= nc) c = nc - 1;
        int y = (int)(p / W), x = (int)(p % W);
        int id = g[(int64_t)(y / up) * w + x / up];
        int ins = ((thing_mask >> c) & 1u) ? id : 0;
        int64_t v;
        if (ins > 0) v = (int64_t)ll[ins];
        else v = oo[c] ? (int64_t)c * div : void_label;
        o[p] = (OutT)v;
    }
}

// 4 pixels per lane: one 4-byte class load, one 8-byte (or 2-byte when up is a multiple of 4) id load,
// one 16-byte label store -- 1 KiB of output per wave instruction.  Requires W % 4 == 0 and up in {1, 4k}.
template <typename OutT>
__global__ __launch_bounds__(256) void fuse_apply_vec4_kernel(const uint8_t *__restrict__ sem,
                                                              const uint16_t *__restrict__ ids, int H, int W, int up,
                                                              int cap, int nc, uint32_t thing_mask, int64_t div,
                                                              int64_t void_label, const int32_t *__restrict__ lut,
                                                              const int32_t *__restrict__ ok, OutT *__restrict__ out)
{
    const int d = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int w = W / up;
    const uint8_t *s = sem + (int64_t)d * HW;
    const uint16_t *g = ids + (int64_t)d * (H / up) * w;
    const uint32_t *ll = reinterpret_cast<const uint32_t *>(lut) + (int64_t)d * (cap + 1);
    const int32_t *oo = ok + (int64_t)d * nc;
    OutT *o = out + (int64_t)d * HW;
    const int64_t n4 = HW / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = q * 4;
        uchar4 c4 = *reinterpret_cast<const uchar4 *>(s + p);
        int cls[4] = {c4.x, c4.y, c4.z, c4.w};
        int id[4];
        const int y = (int)(p / W), x = (int)(p % W);
        if (up == 1) {
            ushort4 i4 = *reinterpret_cast<const ushort4 *>(g + p);
            id[0] = i4.x; id[1] = i4.y; id[2] = i4.z; id[3] = i4.w;
        } else {
            int v = g[(int64_t)(y / up) * w + x / up];
            id[0] = id[1] = id[2] = id[3] = v;
        }
        OutT r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = cls[j] < nc ? cls[j] : nc - 1;
            int ins = ((thing_mask >> c) & 1u) ? id[j] : 0;
            int64_t v = ins > 0 ? (int64_t)ll[ins] : (oo[c] ? (int64_t)c * div : void_label);
            r[j] = (OutT)v;
        }
        if (sizeof(OutT) == 4) {
            *reinterpret_cast<uint4 *>(o + p) = make_uint4((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[p + j] = r[j];
        }
    }
}

// uint32 labels, FA_U independent 4-pixel items per lane: every wave instruction still covers one contiguous segment
// (256 B of classes, 512 B of ids, 1 KiB of labels -- a lane that owned 16 consecutive pixels would store 16-byte pieces
// 64 bytes apart, which measured 25-45 % SLOWER), but the loads of all items are issued before the first label is formed,
// so a lane has FA_U x 28 bytes in flight.  The per-class "keep this stuff class" flags become a bit mask in a scalar
// register; the label table is only touched by thing pixels.
#define FA_U 4
__global__ __launch_bounds__(256) void fuse_apply_multi_kernel(const uint8_t *__restrict__ sem,
                                                               const uint16_t *__restrict__ ids, int H, int W, int up,
                                                               int cap, int nc, uint32_t thing_mask, int64_t div,
                                                               int64_t void_label, const int32_t *__restrict__ lut,
                                                               const int32_t *__restrict__ ok,
                                                               uint32_t *__restrict__ out)
{
    const int d = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int w = W / up;
    const uint8_t *s = sem + (int64_t)d * HW;
    const uint16_t *g = ids + (int64_t)d * (H / up) * w;
    const uint32_t *ll = reinterpret_cast<const uint32_t *>(lut) + (int64_t)d * (cap + 1);
    const int32_t *oo = ok + (int64_t)d * nc;
    uint32_t *o = out + (int64_t)d * HW;
    uint32_t okmask = 0;
    for (int c = 0; c < nc; ++c) okmask |= oo[c] ? (1u << c) : 0u;          // wave-uniform: scalar loads
    const uint32_t vlab = (uint32_t)void_label, udiv = (uint32_t)div;
    const int64_t n4 = HW / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < n4; q0 += stride * FA_U) {
        uint32_t cw[FA_U];
        uint2 iw[FA_U];
#pragma unroll
        for (int u = 0; u < FA_U; ++u) {
            const int64_t q = q0 + u * stride;
            cw[u] = 0; iw[u] = make_uint2(0, 0);
            if (q < n4) {
                const int64_t p = q * 4;
                cw[u] = *reinterpret_cast<const uint32_t *>(s + p);
                if (up == 1) {
                    iw[u] = *reinterpret_cast<const uint2 *>(g + p);
                } else {
                    const uint32_t p32 = (uint32_t)p, W32 = (uint32_t)W;      // a slice has < 2^31 pixels here
                    const uint32_t v = g[(int64_t)((p32 / W32) / (uint32_t)up) * w + (p32 % W32) / (uint32_t)up];
                    iw[u] = make_uint2(v | (v << 16), v | (v << 16));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FA_U; ++u) {
            const int64_t q = q0 + u * stride;
            if (q < n4) {
                uint32_t r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t c = (cw[u] >> (8 * j)) & 0xffu;
                    c = c < (uint32_t)nc ? c : (uint32_t)(nc - 1);
                    const uint32_t id = ((j < 2 ? iw[u].x : iw[u].y) >> (16 * (j & 1))) & 0xffffu;
                    uint32_t v = ((okmask >> c) & 1u) ? c * udiv : vlab;
                    if (((thing_mask >> c) & 1u) && id > 0) v = ll[id];
                    r[j] = v;
                }
                *reinterpret_cast<uint4 *>(o + q * 4) = make_uint4(r[0], r[1], r[2], r[3]);
            }
        }
    }
}

// 4 pixels per lane for the histogram pass (same loads as above); sub-position j of all lanes is
// aggregated across the wave before touching the block-local LDS bins.
__global__ __launch_bounds__(256) void fuse_hist_vec4_kernel(const uint8_t *__restrict__ sem,
                                                             const uint16_t *__restrict__ ids, int H, int W, int up,
                                                             int cap, int nc, uint32_t thing_mask,
                                                             int32_t *__restrict__ hist, int32_t *__restrict__ stuff)
{
    extern __shared__ int32_t lh[];
    const int d = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int w = W / up;
    const uint8_t *s = sem + (int64_t)d * HW;
    const uint16_t *g = ids + (int64_t)d * (H / up) * w;
    int32_t *hh = hist + (int64_t)d * (cap + 1) * nc;
    int32_t *ss = stuff + (int64_t)d * nc;
    const int n_hist = (cap + 1) * nc;
    for (int i = threadIdx.x; i < n_hist + nc; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const int64_t n4 = HW / 4;
    const int64_t span = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = (int64_t)blockIdx.x * span;
    const int64_t q1 = (q0 + span < n4) ? q0 + span : n4;
    // the loads of the NEXT two trips are in flight while a trip is binned (one trip at a time ran at 2.2 TB/s)
    uint32_t cw[3] = {0, 0, 0};
    uint2 iw[3] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)};
    auto fetch = [&](int64_t q, uint32_t &c, uint2 &i) {
        c = 0; i = make_uint2(0, 0);
        if (q < q1) {
            const int64_t p = q * 4;
            c = *reinterpret_cast<const uint32_t *>(s + p);
            if (up == 1) {
                i = *reinterpret_cast<const uint2 *>(g + p);
            } else {
                const uint32_t p32 = (uint32_t)p, W32 = (uint32_t)W;       // HW < 2^31 (checked by the launcher)
                const uint32_t v = g[(int64_t)((p32 / W32) / (uint32_t)up) * w + (p32 % W32) / (uint32_t)up];
                i = make_uint2(v | (v << 16), v | (v << 16));
            }
        }
    };
    fetch(q0 + threadIdx.x, cw[0], iw[0]);
    fetch(q0 + blockDim.x + threadIdx.x, cw[1], iw[1]);
    for (int64_t qq = q0; qq < q1; qq += blockDim.x) {
        int64_t q = qq + threadIdx.x;
        bool live = q < q1;
        fetch(q + 2 * (int64_t)blockDim.x, cw[2], iw[2]);
        int cls[4], id[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cls[j] = (int)((cw[0] >> (8 * j)) & 0xffu);
            id[j] = (int)(((j < 2 ? iw[0].x : iw[0].y) >> (16 * (j & 1))) & 0xffffu);
        }
        cw[0] = cw[1]; iw[0] = iw[1];
        cw[1] = cw[2]; iw[1] = iw[2];
        // fold the lane's own 4 pixels first when they agree (the common case), else add them one by one
        int c0 = cls[0] < nc ? cls[0] : nc - 1;
        int i0 = ((thing_mask >> c0) & 1u) ? id[0] : 0;
        bool same = true;
#pragma unroll
        for (int j = 1; j < 4; ++j) same = same && (cls[j] == cls[0]) && (id[j] == id[0]);
        if (__all(same || !live)) {
            const int lane = threadIdx.x & 63;
            int64_t key = (i0 > 0) ? (int64_t)i0 * nc + c0 : (int64_t)n_hist + c0;
            int64_t prev = __shfl_up(key, 1);
            bool pvalid = __shfl_up((int)live, 1) != 0;
            bool head = live && (lane == 0 || !pvalid || prev != key);
            unsigned long long heads = __ballot(head), valids = __ballot(live);
            if (head) {
                unsigned long long stop = (heads | ~valids) & ~((2ULL << lane) - 1ULL);
                int end = stop ? __ffsll((long long)stop) - 1 : 64;
                atomicAdd(&lh[key], 4 * (end - lane));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int c = cls[j] < nc ? cls[j] : nc - 1;
                int ins = ((thing_mask >> c) & 1u) ? id[j] : 0;
                wave_hist_add(lh, (int64_t)ins * nc + c, live && ins > 0);
                wave_hist_add(lh + n_hist, (int64_t)c, live && ins == 0);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_hist; i += blockDim.x) {
        int v = lh[i];
        if (v) atomicAdd(hh + i, v);
    }
    if (threadIdx.x < nc) {
        int v = lh[n_hist + threadIdx.x];
        if (v) atomicAdd(ss + threadIdx.x, v);
    }
}

static bool fuse_vec4_ok(const uint8_t *sem, const uint16_t *ids, const void *out, int W, int up)
{
    // 4-pixel-per-lane kernels need aligned rows and ids that are either per pixel or shared by the 4 pixels
    return (W % 4 == 0) && (up == 1 || up % 4 == 0) && ((reinterpret_cast<uintptr_t>(sem) & 3) == 0) &&
           ((reinterpret_cast<uintptr_t>(ids) & 7) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
}

static int fuse_check(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up, int cap, int n_classes,
                      int64_t label_divisor, const int32_t *work)
{
    EMP_REQUIRE(sem && ids && work, "fuse: null pointer");
    EMP_REQUIRE(up >= 1 && H % up == 0 && W % up == 0, "fuse: H,W must be multiples of up=%d", up);
    EMP_REQUIRE(n_classes >= 1 && n_classes <= EMP_MAX_CLASSES, "fuse: n_classes out of range");
    EMP_REQUIRE(cap >= 1 && cap <= 65535, "fuse: cap out of range");
    EMP_REQUIRE(D >= 0 && D <= 65535 && H > 0 && W > 0, "fuse: bad shape");
    EMP_REQUIRE(label_divisor > 0 && (n_classes * label_divisor) < (1LL << 32), "fuse: labels exceed 32 bits");
    return EMP_OK;
}

extern "C" int emp_fuse_lut(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up, int cap,
                            int n_classes, uint32_t thing_mask, int64_t label_divisor, int64_t stuff_area,
                            int32_t *work, void *stream)
{
    int rc = fuse_check(sem, ids, D, H, W, up, cap, n_classes, label_divisor, work);
    if (rc != EMP_OK) return rc;
    if (D == 0) return EMP_OK;
    hipStream_t st = emp_stream(stream);
    FuseLayout L = fuse_layout(D, cap, n_classes);
    if (hipMemsetAsync(work, 0, sizeof(int32_t) * L.total, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "fuse: memset failed");
    int64_t HW = (int64_t)H * W;
    int gx = emp_grid(HW, 256, 1024);
    const bool vec4 = fuse_vec4_ok(sem, ids, nullptr, W, up) && HW < (1LL << 31);
    // ~8K pixels per block keeps the flush cheap; D * gh blocks fill the chip
    int gh = (int)emp_cdiv(HW, 8192);
    if ((int64_t)gh * D < 1024) gh = (int)emp_cdiv(1024, D);
    if (gh > gx) gh = gx;
    size_t lds = ((size_t)(cap + 1) * n_classes + n_classes) * sizeof(int32_t);
    int use_lds = lds <= 48 * 1024;
    if (vec4 && use_lds)
        hipLaunchKernelGGL(fuse_hist_vec4_kernel, dim3(gh, D), dim3(256), lds, st, sem, ids, H, W, up, cap, n_classes,
                           thing_mask, work + L.hist, work + L.stuff);
    else
        hipLaunchKernelGGL(fuse_hist_kernel, dim3(gh, D), dim3(256), use_lds ? lds : 0, st, sem, ids, H, W, up, cap,
                           n_classes, thing_mask, work + L.hist, work + L.stuff, use_lds);
    EMP_CHECK_LAUNCH("emp_fuse_lut(hist)");
    hipLaunchKernelGGL(fuse_lut_kernel, dim3(D), dim3(256), 0, st, cap, n_classes, thing_mask, label_divisor,
                       stuff_area, work + L.hist, work + L.stuff, work + L.lut, work + L.ok);
    EMP_CHECK_LAUNCH("emp_fuse_lut(lut)");
    return EMP_OK;
}

extern "C" int emp_fuse_apply(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up, int cap,
                              int n_classes, uint32_t thing_mask, int64_t label_divisor, int64_t void_label,
                              const int32_t *work, uint32_t *out_pan_u32, int64_t *out_pan_i64, void *stream)
{
    int rc = fuse_check(sem, ids, D, H, W, up, cap, n_classes, label_divisor, work);
    if (rc != EMP_OK) return rc;
    EMP_REQUIRE((out_pan_u32 != nullptr) != (out_pan_i64 != nullptr), "fuse: exactly one output must be given");
    if (D == 0) return EMP_OK;
    hipStream_t st = emp_stream(stream);
    FuseLayout L = fuse_layout(D, cap, n_classes);
    int64_t HW = (int64_t)H * W;
    int gx = emp_grid(HW, 256, 1024);
    const bool vec4 = fuse_vec4_ok(sem, ids, out_pan_u32 ? (void *)out_pan_u32 : (void *)out_pan_i64, W, up);
    if (vec4 && out_pan_u32 && HW < (1LL << 31)) {
        int gv = emp_grid(emp_cdiv(HW / 4, FA_U), 256, 1024);           // one trip of FA_U items per lane
        hipLaunchKernelGGL(fuse_apply_multi_kernel, dim3(gv, D), dim3(256), 0, st, sem, ids, H, W, up, cap, n_classes,
                           thing_mask, label_divisor, void_label, work + L.lut, work + L.ok, out_pan_u32);
    } else if (vec4) {
        int gv = emp_grid(HW / 4, 256, 1024);
        if (out_pan_u32)
            hipLaunchKernelGGL(fuse_apply_vec4_kernel<uint32_t>, dim3(gv, D), dim3(256), 0, st, sem, ids, H, W, up,
                               cap, n_classes, thing_mask, label_divisor, void_label, work + L.lut, work + L.ok,
                               out_pan_u32);
        else
            hipLaunchKernelGGL(fuse_apply_vec4_kernel<int64_t>, dim3(gv, D), dim3(256), 0, st, sem, ids, H, W, up,
                               cap, n_classes, thing_mask, label_divisor, void_label, work + L.lut, work + L.ok,
                               out_pan_i64);
    } else if (out_pan_u32)
        hipLaunchKernelGGL(fuse_apply_kernel<uint32_t>, dim3(gx, D), dim3(256), 0, st, sem, ids, H, W, up, cap,
                           n_classes, thing_mask, label_divisor, void_label, work + L.lut, work + L.ok,
                           out_pan_u32);
    else
        hipLaunchKernelGGL(fuse_apply_kernel<int64_t>, dim3(gx, D), dim3(256), 0, st, sem, ids, H, W, up, cap,
                           n_classes, thing_mask, label_divisor, void_label, work + L.lut, work + L.ok,
                           out_pan_i64);
    EMP_CHECK_LAUNCH("emp_fuse_apply");
    return EMP_OK;
}

extern "C" int emp_fuse_panoptic(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up, int cap,
                                 int n_classes, uint32_t thing_mask, int64_t label_divisor, int64_t stuff_area,
                                 int64_t void_label, int32_t *work, uint32_t *out_pan_u32,
                                 int64_t *out_pan_i64, void *stream)
{
    EMP_REQUIRE((out_pan_u32 != nullptr) != (out_pan_i64 != nullptr), "fuse: exactly one output must be given");
    int rc = emp_fuse_lut(sem, ids, D, H, W, up, cap, n_classes, thing_mask, label_divisor, stuff_area, work, stream);
    if (rc != EMP_OK) return rc;
    return emp_fuse_apply(sem, ids, D, H, W, up, cap, n_classes, thing_mask, label_divisor, void_label, work,
                          out_pan_u32, out_pan_i64, stream);
}

// ------------------------------------------------------------------------------------------
// D1 epilogue: y = relu?(x * scale[c] + shift[c] (+ residual)) on NHWC fp32 activations, in one pass.
// Replaces the separate BatchNorm(eval) / residual add / ReLU elementwise kernels that follow every
// convolution of the encoder-decoder (models/encoders/resnet.py:110-128, blocks.py:121-171).
// Pure streaming: 4 channels per lane (float4), scale/shift stay in L1.  Traffic 4 B read (+4 B residual)
// + 4 B written per element.
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_nhwc_kernel(const float4 *__restrict__ x,
                                                          const float4 *__restrict__ scale,
                                                          const float4 *__restrict__ shift,
                                                          const float4 *__restrict__ res, int64_t n4, int c4,
                                                          int64_t out_stride4, float4 *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        float4 v = x[i], a = scale[c], b = shift[c];
        float4 y;
        y.x = __fadd_rn(__fmul_rn(v.x, a.x), b.x);
        y.y = __fadd_rn(__fmul_rn(v.y, a.y), b.y);
        y.z = __fadd_rn(__fmul_rn(v.z, a.z), b.z);
        y.w = __fadd_rn(__fmul_rn(v.w, a.w), b.w);
        if (RES) {
            float4 r = res[i];
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        if (RELU) {
            y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
        }
        // out_stride4 == c4: dense output (index i); otherwise a channel slice of a wider NHWC buffer
        out[out_stride4 == c4 ? i : (i / c4) * out_stride4 + c] = y;
    }
}

extern "C" int emp_bn_act_nhwc(const float *x, const float *scale, const float *shift, const float *residual,
                               int relu, int64_t n_pixels, int C, float *out, int64_t out_pixel_stride, void *stream)
{
    EMP_REQUIRE(x && scale && shift && out, "bn_act: null pointer");
    EMP_REQUIRE(C > 0 && C % 4 == 0, "bn_act: channel count %d must be a multiple of 4", C);
    EMP_REQUIRE(n_pixels >= 0, "bn_act: bad size");
    if (out_pixel_stride == 0) out_pixel_stride = C;
    EMP_REQUIRE(out_pixel_stride >= C && out_pixel_stride % 4 == 0, "bn_act: bad output pixel stride");
    EMP_REQUIRE(out_pixel_stride == C || out != x, "bn_act: a strided output cannot alias the input");
    EMP_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(scale) |
                  reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0,
                "bn_act: pointers must be 16-byte aligned");
    if (n_pixels == 0) return EMP_OK;
    const int64_t n4 = n_pixels * (C / 4);
    const int grid = emp_grid(n4, 256, 16384);
    hipStream_t st = emp_stream(stream);
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *s4 = reinterpret_cast<const float4 *>(scale);
    const float4 *b4 = reinterpret_cast<const float4 *>(shift), *r4 = reinterpret_cast<const float4 *>(residual);
    float4 *o4 = reinterpret_cast<float4 *>(out);
#define EMP_BN(R, A) hipLaunchKernelGGL((bn_act_nhwc_kernel<R, A>), dim3(grid), dim3(256), 0, st, x4, s4, b4, r4, n4, C / 4, out_pixel_stride / 4, o4)
    if (residual) { if (relu) EMP_BN(true, true); else EMP_BN(true, false); }
    else { if (relu) EMP_BN(false, true); else EMP_BN(false, false); }
#undef EMP_BN
    EMP_CHECK_LAUNCH("emp_bn_act_nhwc");
    return EMP_OK;
}
