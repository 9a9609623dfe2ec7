/*
 * oracle/c/oracle_kernels.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the per-pixel arithmetic of empanada's panoptic
 * post-processing, used as the CPU checker ("oracle") for the HIP kernels in
 * empanada_amd/csrc.  Nothing in the product package may link or call this.
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Build: gcc -O2 -ffp-contract=off -shared -fPIC.
 * -ffp-contract=off matters: the only fused multiply-add allowed is the
 * explicit fmaf() in emp_oracle_group_pixels (see the note there).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* P3  find_instance_center           empanada/inference/postprocess.py:38-76 */
/*
 * F.threshold(h, thr, -1): keep v where v > thr (strict), else -1.
 * max_pool2d(k, stride 1, pad k//2) pads with -inf; an even k crops the last
 * row/col (:63-65) so pixel (y,x) sees rows y-k/2 .. y-k/2+k-1.  A pixel stays
 * a centre when it equals the pooled value and is > 0.  Centres come out in
 * raster order (torch.nonzero).  Returns the number of centres; writes at most
 * cap (y,x) pairs.
 */
int64_t emp_oracle_find_centers(const float *hmp, int h, int w, float thr,
                                int k, int64_t *out_yx, int64_t cap)
{
    int64_t n = 0;
    int pad = k / 2;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            float v = hmp[(size_t)y * w + x];
            v = (v > thr) ? v : -1.0f;
            if (!(v > 0.0f)) continue;
            float m = -INFINITY;
            for (int dy = 0; dy < k; ++dy) {
                int yy = y - pad + dy;
                if (yy < 0 || yy >= h) continue;
                for (int dx = 0; dx < k; ++dx) {
                    int xx = x - pad + dx;
                    if (xx < 0 || xx >= w) continue;
                    float u = hmp[(size_t)yy * w + xx];
                    u = (u > thr) ? u : -1.0f;
                    if (u > m) m = u;
                }
            }
            if (v == m) {
                if (n < cap) { out_yx[2 * n] = y; out_yx[2 * n + 1] = x; }
                ++n;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------ */
/* P4  group_pixels / chunked_pixel_grouping  postprocess.py:78-169           */
/*
 * coord = arange(0, h*step, step) (fp32), ctr_loc = coord + offsets (fp32 add),
 * ctr = step * ctr (exact in fp32), distance = torch.norm(ctr - ctr_loc, dim=-1).
 *
 * Rounding of torch.norm over a last dim of size 2 on the CPU build of torch
 * 2.10 (measured in this container on 1.4e6 random pairs, 0 mismatches):
 *     s = fmaf(dx, dx, fl(dy*dy));  d = sqrtf(s)   (sqrt correctly rounded)
 * i.e. the dy term is squared and rounded first, the dx term is fused.
 *
 * K <= 20: id = 1 + argmin_k d (first minimum).  K > 20: chunks of 20, running
 * strict '<' against nearest (initialised to 1e5), ids offset by the chunk
 * base; a pixel farther than 1e5 from every centre keeps id 0 (:97-111).  Both
 * reduce to "first k with the strictly smallest d", with the 1e5 ceiling when
 * K > 20.
 */
void emp_oracle_group_pixels(const int64_t *ctr_yx, int64_t K,
                             const float *offsets /* (2,h,w) */, int h, int w,
                             int step, int64_t *out_ids /* (h,w) */)
{
    const float *offy = offsets;
    const float *offx = offsets + (size_t)h * w;
    for (int y = 0; y < h; ++y) {
        float cy = (float)(y * step);
        for (int x = 0; x < w; ++x) {
            float cx = (float)(x * step);
            size_t p = (size_t)y * w + x;
            float ly = cy + offy[p];
            float lx = cx + offx[p];
            float best = (K > 20) ? 1e5f : INFINITY;
            int64_t id = 0;
            int first = (K <= 20);
            for (int64_t k = 0; k < K; ++k) {
                float ky = (float)step * (float)ctr_yx[2 * k];
                float kx = (float)step * (float)ctr_yx[2 * k + 1];
                float dy = ky - ly;
                float dx = kx - lx;
                float s = dy * dy;
                s = fmaf(dx, dx, s);
                float d = sqrtf(s);
                if (first) { best = d; id = 1; first = 0; }
                else if (d < best) { best = d; id = k + 1; }
            }
            out_ids[p] = id;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* R1  connected_components            empanada/inference/rle.py:18-24        */
/*
 * cc3d.connected_components(connectivity=8) / skimage.measure.label(): multi-
 * value labelling, 8-connectivity, background 0, output ids 1..n numbered in
 * raster order of each component's first pixel.  (Third-party algorithm;
 * neither library is in the image, this restates the published contract.)
 * Two-pass union-find, then renumber by first appearance in raster order.
 */
static int64_t uf_find(int64_t *parent, int64_t a)
{
    while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; }
    return a;
}
static void uf_union(int64_t *parent, int64_t a, int64_t b)
{
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) parent[b] = a; else parent[a] = b;
}
int64_t emp_oracle_cc8(const int64_t *seg, int h, int w, int64_t *out)
{
    size_t n = (size_t)h * w;
    int64_t *parent = (int64_t *)malloc(n * sizeof(int64_t));
    if (!parent) return -1;
    for (size_t i = 0; i < n; ++i) parent[i] = (int64_t)i;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            size_t p = (size_t)y * w + x;
            int64_t v = seg[p];
            if (v == 0) continue;
            if (x > 0 && seg[p - 1] == v) uf_union(parent, p, p - 1);
            if (y > 0) {
                if (seg[p - w] == v) uf_union(parent, p, p - w);
                if (x > 0 && seg[p - w - 1] == v) uf_union(parent, p, p - w - 1);
                if (x + 1 < w && seg[p - w + 1] == v) uf_union(parent, p, p - w + 1);
            }
        }
    }
    int64_t next = 0;
    /* roots are minimal flat indices, so a root is met before its members */
    for (size_t p = 0; p < n; ++p) {
        if (seg[p] == 0) { out[p] = 0; continue; }
        int64_t r = uf_find(parent, (int64_t)p);
        if ((size_t)r == p) out[p] = ++next;
        else out[p] = out[r];
    }
    free(parent);
    return next;
}

/* ------------------------------------------------------------------------ */
/* P1  _MedianQueue.get_median         empanada/inference/engines.py:59-66    */
/* median over ks stacked slices, per pixel (torch.median, odd count).        */
static int cmp_f32(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}
void emp_oracle_median(const float *const *slices, int ks, int64_t n, float *out)
{
    float buf[32];
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < ks; ++k) buf[k] = slices[k][i];
        qsort(buf, ks, sizeof(float), cmp_f32);
        out[i] = buf[(ks - 1) / 2];
    }
}

/* ------------------------------------------------------------------------ */
/* R4/Z1  numpy_fill_instances / fill_func  array_utils.py:725-737,            */
/*                                          zarr_utils.py:49-58                */
void emp_oracle_fill_u32(uint32_t *vol, int64_t nvox, const int64_t *starts,
                         const int64_t *runs, int64_t nruns, uint32_t value)
{
    for (int64_t i = 0; i < nruns; ++i) {
        int64_t s = starts[i], e = s + runs[i];
        if (s < 0) s = 0;
        if (e > nvox) e = nvox;
        for (int64_t j = s; j < e; ++j) vol[j] = value;
    }
}

/* Depthwise k x k convolution, NHWC fp32, stride 1, zero "same" padding: the depthwise half of
 * SeparableConv2d (empanada/models/blocks.py:15-33; torch Conv2d with groups == channels).
 * The reference leaves the summation order to the backend; this restatement fixes it to the order the
 * HIP kernel documents (include/emp_hip.h, D2): one fmaf chain over the taps in raster order from +0,
 * out-of-image taps entering as 0, bias (or +0) added last.  Agreement with torch's own conv2d is a
 * tolerance test (tests/test_hip_kernels.py). */
void emp_oracle_dwconv_nhwc(const float *x, const float *w_kkc, const float *bias, int N, int H, int W,
                            int C, int k, float *y)
{
    const int P = k / 2;
    for (int n = 0; n < N; ++n)
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c)
                for (int ch = 0; ch < C; ++ch) {
                    float acc = 0.0f;
                    for (int i = 0; i < k; ++i)
                        for (int j = 0; j < k; ++j) {
                            int rr = r + i - P, cc = c + j - P;
                            float v = 0.0f;
                            if (rr >= 0 && rr < H && cc >= 0 && cc < W)
                                v = x[(((int64_t)n * H + rr) * W + cc) * C + ch];
                            acc = fmaf(v, w_kkc[(int64_t)(i * k + j) * C + ch], acc);
                        }
                    y[(((int64_t)n * H + r) * W + c) * C + ch] = acc + (bias ? bias[ch] : 0.0f);
                }
}

/* Convolution + eval-BatchNorm affine + residual + ReLU on NHWC fp32 (Conv2d -> BatchNorm2d -> (+identity) -> ReLU,
 * empanada/models/encoders/resnet.py:66-82,110-128; blocks.py:121-171).  The reference leaves the summation order
 * to the backend; this restatement fixes it to the order include/emp_hip.h (D4) documents for the MFMA kernel:
 * one fmaf chain from +0 over filter taps in raster order, per tap over slabs of 32 input channels, per slab in
 * the order c, c + slab/2 for c = 0..slab/2-1 (slab = 32 or 16: emp_conv_k_slab); taps outside the image enter
 * as 0.  w: (Cout, KH, KW, Cin). */
void emp_oracle_conv_bn_act_nhwc(const float *x, const float *w, const float *scale, const float *shift,
                                 const float *res, int relu, int N, int H, int W, int Cin, int Cout, int KH,
                                 int KW, int stride, int pad, int dil, int slab, float *out)
{
    const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    const int OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                const int64_t p = ((int64_t)n * OH + oy) * OW + ox;
                for (int co = 0; co < Cout; ++co) {
                    float acc = 0.0f;
                    for (int ky = 0; ky < KH; ++ky)
                        for (int kx = 0; kx < KW; ++kx) {
                            const int iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
                            const int in = iy >= 0 && iy < H && ix >= 0 && ix < W;
                            const float *xp = in ? x + (((int64_t)n * H + iy) * W + ix) * Cin : 0;
                            const float *wp = w + (((int64_t)co * KH + ky) * KW + kx) * Cin;
                            for (int c0 = 0; c0 < Cin; c0 += slab)
                                for (int j = 0; j < slab / 2; ++j) {
                                    acc = fmaf(in ? xp[c0 + j] : 0.0f, wp[c0 + j], acc);
                                    acc = fmaf(in ? xp[c0 + slab / 2 + j] : 0.0f, wp[c0 + slab / 2 + j], acc);
                                }
                        }
                    float v = acc;
                    if (scale) v = v * scale[co];
                    if (shift) v = v + shift[co];
                    if (relu == 2) {                 /* squeeze-excite gate (blocks.py:35-50): x * sigmoid(conv(s) + b) */
                        v = res[p * Cout + co] * (1.0f / (1.0f + expf(-v)));
                    } else {
                        if (res) v = v + res[p * Cout + co];
                        if (relu) v = v > 0.0f ? v : 0.0f;
                    }
                    out[p * Cout + co] = v;
                }
            }
}

/* D4c (emp_conv_splitk_bn_act_nhwc): the same convolution with the reduction cut into k_splits ranges of whole
 * 32-channel slabs.  Slab index s = tap * (Cin / 32) + c0 / 32 runs over S = KH KW Cin / 32 slabs; range z covers
 * [S z / k, S (z + 1) / k); partial p_z = the fmaf chain of the function above over its slabs, from +0; the result is
 * ((p_0 + p_1) + p_2) + ... then * scale, + shift, + residual, relu -- every step a separate fp32 rounding. */
void emp_oracle_conv_splitk_bn_act_nhwc(const float *x, const float *w, const float *scale, const float *shift,
                                        const float *res, int relu, int N, int H, int W, int Cin, int Cout, int KH,
                                        int KW, int stride, int pad, int dil, int k_splits, float *out)
{
    const int slab = 32;
    const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    const int OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    const int cslabs = Cin / slab, S = KH * KW * cslabs;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                const int64_t p = ((int64_t)n * OH + oy) * OW + ox;
                for (int co = 0; co < Cout; ++co) {
                    float total = 0.0f;
                    for (int z = 0; z < k_splits; ++z) {
                        const int s_lo = (int)((int64_t)S * z / k_splits), s_hi = (int)((int64_t)S * (z + 1) / k_splits);
                        float acc = 0.0f;
                        for (int s = s_lo; s < s_hi; ++s) {
                            const int tap = s / cslabs, c0 = (s % cslabs) * slab;
                            const int ky = tap / KW, kx = tap % KW;
                            const int iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
                            const int in = iy >= 0 && iy < H && ix >= 0 && ix < W;
                            const float *xp = in ? x + (((int64_t)n * H + iy) * W + ix) * Cin : 0;
                            const float *wp = w + (((int64_t)co * KH + ky) * KW + kx) * Cin;
                            for (int j = 0; j < slab / 2; ++j) {
                                acc = fmaf(in ? xp[c0 + j] : 0.0f, wp[c0 + j], acc);
                                acc = fmaf(in ? xp[c0 + slab / 2 + j] : 0.0f, wp[c0 + slab / 2 + j], acc);
                            }
                        }
                        total = z == 0 ? acc : total + acc;
                    }
                    float v = total;
                    v = v * (scale ? scale[co] : 1.0f);
                    v = v + (shift ? shift[co] : 0.0f);
                    if (res) v = v + res[p * Cout + co];
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    out[p * Cout + co] = v;
                }
            }
}

/* D8 (emp_gconv3x3_bn_act_nhwc): grouped 3x3 convolution, padding 1, stride 1 or 2, NHWC, G groups of GW channels
 * (Conv2d(w, w, 3, stride, 1, groups=G) of the RegNet bottleneck, empanada/models/encoders/regnet.py:59-71) + affine +
 * ReLU.  w: (G*GW, 3, 3, GW).  One fmaf chain from +0 per output: taps in raster order; per tap chunks of CK channels
 * ascending (CK = emp_gconv_chunk(GW): 24, 16 or 8); per chunk 8-channel slabs j ascending; per slab e = 0, 1; per e
 * the channels 8j + 2kq + e for kq = 0..3.  Taps outside the image enter as x = 0. */
void emp_oracle_gconv3x3_bn_act_nhwc(const float *x, const float *w, const float *scale, const float *shift, int relu,
                                     int N, int H, int W, int G, int GW, int stride, int CK, float *out)
{
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    const int C = G * GW;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                const int64_t p = ((int64_t)n * OH + oy) * OW + ox;
                for (int co = 0; co < C; ++co) {
                    const int grp = co / GW;
                    float acc = 0.0f;
                    for (int tap = 0; tap < 9; ++tap) {
                        const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
                        const int in = iy >= 0 && iy < H && ix >= 0 && ix < W;
                        const float *xp = in ? x + (((int64_t)n * H + iy) * W + ix) * C + grp * GW : 0;
                        const float *wp = w + ((int64_t)co * 9 + tap) * GW;
                        for (int c0 = 0; c0 < GW; c0 += CK)
                            for (int j = 0; j < CK / 8; ++j)
                                for (int e = 0; e < 2; ++e)
                                    for (int kq = 0; kq < 4; ++kq) {
                                        const int c = c0 + 8 * j + 2 * kq + e;
                                        acc = fmaf(in ? xp[c] : 0.0f, wp[c], acc);
                                    }
                    }
                    float v = acc;
                    if (scale) v = v * scale[co];
                    if (shift) v = v + shift[co];
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    out[p * C + co] = v;
                }
            }
}

/* D9 (emp_stem_conv7_bn_relu_maxpool), convolution part: Conv2d(1, Cout, 7, stride 2, padding 3) of the ResNet stem
 * (empanada/models/encoders/resnet.py:186-188,217-222) on a one-channel image: one fmaf chain from +0 per output over
 * the 49 taps in raster order, taps outside the image enter as x = 0.  w_tc: (49, Cout).  out: (N, OH, OW, Cout). */
void emp_oracle_conv7s2_c1(const float *x, const float *w_tc, int N, int H, int W, int Cout, float *out)
{
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox)
                for (int co = 0; co < Cout; ++co) {
                    float acc = 0.0f;
                    for (int ky = 0; ky < 7; ++ky)
                        for (int kx = 0; kx < 7; ++kx) {
                            const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
                            const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((int64_t)n * H + iy) * W + ix] : 0.0f;
                            acc = fmaf(xv, w_tc[(ky * 7 + kx) * Cout + co], acc);
                        }
                    out[(((int64_t)n * OH + oy) * OW + ox) * Cout + co] = acc;
                }
}
