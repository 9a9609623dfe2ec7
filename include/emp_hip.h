/*
 * emp_hip.h -- C ABI of libemp_hip.so, the MI355X (gfx950) implementation of empanada's
 * orthoplane-inference post-processing hot path.
 *
 * The reference (volume-em/empanada) is 100 % Python: it has no FFI.  This ABI is therefore
 * NEW surface; each entry point names the reference function(s) (path:line relative to the
 * reference root) whose arithmetic it replaces.  The Python package `empanada_amd` binds these
 * with ctypes (empanada_amd/_hip.py) behind the reference's own engine / matcher / tracker
 * names; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and never allocates, frees or synchronises, so callers may capture it in a
 *     hipGraph.  Workspaces are caller-provided.
 *   - return value: 0 on success, negative EMP_E* on error; emp_last_error() returns a
 *     per-thread message.  No exceptions cross the boundary.
 *   - images are row-major; a "stack" is D slices of (H, W); "HW" = H*W.
 *   - labels: pan = class_id * label_divisor + instance_id (reference convention).
 */
#ifndef EMP_HIP_H
#define EMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMP_OK 0
#define EMP_EINVAL (-1)   /* bad argument (shape, size, unsupported parameter) */
#define EMP_ELAUNCH (-2)  /* HIP launch / runtime error */
#define EMP_ENODEV (-3)   /* no usable gfx950 device */

#define EMP_MAX_KS 11       /* median kernel sizes 1,3,...,11 (reference: scripts/pdl_inference3d.py:28) */
#define EMP_MAX_CLASSES 16  /* semantic channels C, and class ids < 16 */
#define EMP_MAX_CENTERS 4096 /* per-slice centre capacity the in-LDS sort supports */

int emp_version(void);
const char *emp_last_error(void);
/* Number of visible HIP devices (does not create a context). */
int emp_device_count(void);

/* ---- D0: slice feeder: uint8 volume -> normalised, zero-padded fp32 model input ---------------------
 * replaces VolumeDataset.__getitem__ + albumentations Normalize + factor_pad for a device-resident volume
 *          empanada/data/volume_dataset.py:7-53, inference/postprocess.py:25-36 (factor_pad)
 * Slice s, row r, column c of the chosen plane is vol[s*stride_slice + r*stride_row + c*stride_col] (element
 * strides; xy: (H*W, W, 1), xz: (W, H*W, 1), yz: (1, H*W, W) for a (D,H,W) volume), so no transposed copy of the
 * volume is ever made.  out (n_slices, 1, hp, wp) fp32 = (x - mean255) * inv_std255 (two fp32 roundings, the
 * arithmetic of albumentations' Normalize with max_pixel_value = 255), rows >= h and columns >= w zero.       */
int emp_slices_to_input(const uint8_t *vol, int64_t stride_slice, int64_t stride_row, int64_t stride_col,
                        int n_slices, int h, int w, int hp, int wp, float mean255, float inv_std255,
                        float *out, void *stream);

/* ---- D1 epilogue: fused BatchNorm(eval) + residual + ReLU on NHWC fp32 activations -----------------
 * replaces the elementwise tail of every conv block of the dense path:
 *          Bottleneck / BasicBlock forward         empanada/models/encoders/resnet.py:66-82,110-128
 *          conv_bn_act / separable_conv_bn_act     empanada/models/blocks.py:121-171
 * out[p, c] = act(x[p, c] * scale[c] + shift[c] (+ residual[p, c])), act = ReLU if relu != 0.
 * scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (precomputed by the host).
 * x, residual: (n_pixels, C) fp32 (NHWC memory), C % 4 == 0, 16-byte aligned.  out: pixel p starts at
 * out + p * out_pixel_stride (floats; 0 means C): either dense (may alias x) or a channel slice of a wider NHWC
 * buffer, which is how the torch.cat of ASPP / decoder branches (aspp.py:96-101,
 * decoders/panoptic_deeplab.py:76-77) is written in place.                                                  */
int emp_bn_act_nhwc(const float *x, const float *scale, const float *shift, const float *residual,
                    int relu, int64_t n_pixels, int C, float *out, int64_t out_pixel_stride, void *stream);

/* ---- D2: depthwise k x k convolution on NHWC fp32 activations (stride 1, zero "same" padding) -------
 * replaces the depthwise half of SeparableConv2d   empanada/models/blocks.py:15-33
 *          (decoder fuse stages                     empanada/models/decoders/panoptic_deeplab.py:52-60,
 *           head trunks                             empanada/models/heads.py:9-19)
 * y[n, r, c, ch] = bias[ch] + sum_{i, j} x[n, r + i - k/2, c + j - k/2, ch] * w_kkc[(i * k + j) * C + ch]
 * evaluated as one fp32 FMA chain over the taps in raster order (i, then j) starting from +0; taps outside the
 * image contribute fma(0, w, acc).  x, y: (N, H, W, C) fp32; w_kkc: (k*k, C) (the (C, 1, k, k) Conv2d weight
 * transposed); bias: (C) or NULL; k in {3, 5}; C % 4 == 0; all pointers 16-byte aligned; y must not alias x. */
int emp_dwconv_nhwc(const float *x, const float *w_kkc, const float *bias, int N, int H, int W, int C,
                    int k, float *y, void *stream);

/* ---- D3: bilinear up-sampling with align_corners = True on fp32 activations ------------------------
 * replaces F.interpolate(..., mode='bilinear', align_corners=True) in
 *          PanopticDeepLab.forward (x4 heads)       empanada/models/panoptic_deeplab.py:100-113
 *          PanopticDeepLabDecoder.forward           empanada/models/decoders/panoptic_deeplab.py:70-78
 *          ASPPPooling.forward                      empanada/models/decoders/aspp.py:48-52
 * x: logical (N, C, h, w), y: logical (N, C, H, W), both addressed through 4 element strides (n, c, h, w), so
 * either side may be NCHW, NHWC or a channel slice of a wider NHWC buffer (the caller's torch.cat target).
 * src = dst * (in - 1) / (out - 1) in fp32 (0 when out == 1); i0 = floor(src), l = src - i0, i1 = min(i0+1, in-1);
 * y = (1-ly) * ((1-lx) * v00 + lx * v01) + ly * ((1-lx) * v10 + lx * v11), every operation a separate fp32
 * rounding.  x_strides / y_strides: HOST arrays of 4 int64.                                                 */
int emp_upsample_bilinear(const float *x, int N, int C, int h, int w, const int64_t *x_strides,
                          float *y, int H, int W, const int64_t *y_strides, void *stream);

/* ---- D4: convolution + BatchNorm(eval) + residual + ReLU in one kernel, NHWC fp32, fp32 matrix cores -----
 * replaces Conv2d -> BatchNorm2d -> (+ identity) -> ReLU chains of the dense path:
 *          Bottleneck / BasicBlock forward         empanada/models/encoders/resnet.py:66-82,110-128
 *          conv_bn_act, ASPP branches, projections empanada/models/blocks.py:121-171, decoders/aspp.py:22-102
 * out[p, co] = act(acc[p, co] * scale[co] + shift[co] + residual[p, co]); scale / shift / residual optional (NULL),
 * act = ReLU if relu != 0; acc = sum over taps (ky, kx) and input channels of x[n, oy*stride - pad + ky*dil,
 * ox*stride - pad + kx*dil, c] * w_okkc[co, ky, kx, c], evaluated as ONE fp32 fma chain from +0 (the MFMA is
 * bit-for-bit an fmaf chain): taps in raster order; per tap, slabs of S channels ascending; per slab the order
 * c, c + S/2 for c = 0..S/2-1; taps outside the image enter as x = 0.  S = emp_conv_k_slab(M = N*OH*OW,
 * Cout, 1, residual != NULL): 16 (three resident blocks per CU) when the launch has more than 512 blocks and does
 * not use the residual-prefetch variant, else 32.  The epilogue's multiply and adds are separate
 * fp32 roundings, as in emp_bn_act_nhwc.
 * x: (N, H, W, Cin), Cin % 16 == 0; w_okkc: (Cout, KH, KW, Cin) (the Conv2d weight permuted); residual and out
 * are addressed as base + pixel * pixel_stride + co (stride 0 means Cout), so out may be a channel slice of a
 * wider NHWC concat buffer.  x and w 16-byte aligned; out must not alias x.                                    */
int emp_conv_k_slab(int64_t M, int Cout, int batch, int has_residual);
/* the same for a given Cin: Cin % 32 != 0 (RegNet widths 144, 1296: multiples of 16 only) always takes S = 16 */
int emp_conv_k_slab_cin(int64_t M, int Cout, int batch, int has_residual, int Cin);
/* D4b: short-K pointwise layers (1x1, stride 1, no padding, Cin 64 or 128, Cout a multiple of 128, at least 65 536
 * output pixels: conv3 + identity and the shortcut projection of the ResNet bottleneck's layer1 / layer2,
 * empanada/models/encoders/resnet.py:98-118) are computed by a weight-stationary kernel inside emp_conv_bn_act_nhwc
 * (emp_conv1x1.hip: weights resident in LDS, activations straight into the MFMA operands, epilogue from the
 * accumulators).  Its summation order is the D4 order with a K-slab of 64: slabs of 64 channels ascending, inside a
 * slab j = 0..31: channel j, then channel 32 + j.  emp_conv_k_slab_geom returns the slab emp_conv_bn_act_nhwc will use
 * for a given geometry (64 for these layers, else emp_conv_k_slab_cin's answer); relu as in emp_conv_bn_act_nhwc. */
int emp_conv_k_slab_geom(int64_t M, int Cout, int has_residual, int Cin, int KH, int KW, int stride, int pad, int relu);
int emp_conv1x1_ws_eligible(int64_t M, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu);
/* relu == 2 selects the squeeze-excite gate epilogue (SqueezeExcite.forward, empanada/models/blocks.py:35-50:
 * x * sigmoid(conv(s) + bias)): out = residual * (1 / (1 + expf(-(acc * scale + shift)))), residual = the gated
 * tensor x (required); the division and the product are separate fp32 roundings, expf is the device library's;
 * its K-slab is emp_conv_k_slab_cin(M, Cout, 1, 0, Cin) (the residual-prefetch variant is never used for it).
 * Cin % 16 == 0 (round 2; was 32).                                                                            */
int emp_conv_bn_act_nhwc(const float *x, const float *w_okkc, const float *scale, const float *shift,
                         const float *residual, int64_t res_pixel_stride, int relu, int N, int H, int W,
                         int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, float *out,
                         int64_t out_pixel_stride, void *stream);
/* D4c: the same convolution for SMALL launches (one 512^2 tile at batch 1: layer3 / layer4 / ASPP have 8-64 tiles of 128
 * pixels for 256 CUs) -- the per-slice protocol's forward, empanada/inference/engines.py:141-159 one image per call.
 * The reduction over the S = KH KW Cin / 32 slabs of 32 channels is cut into k_splits ranges [S z / k, S (z + 1) / k);
 * every range is a block of its own and writes its partial sums (the fmaf chain of emp_conv_bn_act_nhwc with K-slab 32,
 * from +0) to plane z of `work` (k_splits x N*OH*OW x Cout floats); a second pass adds the planes in ascending order
 * and applies the epilogue, out = relu?(sum * scale + shift (+ residual)), every operation a separate fp32 rounding.
 * Deterministic.  Cin % 32 == 0, Cout % 4 == 0, relu 0 / 1; all pointers 16-byte aligned, pixel strides % 4 == 0.
 * emp_conv_splitk_plan: the number of ranges worth using for a geometry (1 = the launch fills the chip as it is). */
int emp_conv_splitk_plan(int64_t M, int Cout, int Cin, int KH, int KW);
int emp_conv_splitk_bn_act_nhwc(const float *x, const float *w_okkc, const float *scale, const float *shift,
                                const float *residual, int64_t res_pixel_stride, int relu, int N, int H, int W,
                                int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int k_splits,
                                float *work, float *out, int64_t out_pixel_stride, void *stream);

/* D4 + D6 in one launch: the convolution's epilogue also evaluates a following 1x1 convolution to proj_n <= 4
 * channels (head = separable conv -> BN -> ReLU -> Conv2d(256, n, 1): heads.py:9-19), so the 256-channel activation
 * is neither written nor re-read (out may be NULL).  proj_out (N, proj_n, OH*OW) planar, ZEROED by the caller;
 * bias is the caller's to add.  Per output pixel and q: for each cout tile (128 wide) the 32 lanes of the row hold
 * ((v0 w0 + v1 w1) + v2 w2) + v3 w3 over their 4 couts (products and sums separate fp32 roundings), summed over
 * the lanes in ascending cout order (left fold), and the tile sums are added to proj_out with one atomicAdd each
 * (Cout is 128 or 256: with at most two tiles the result is independent of their order).                      */
int emp_conv_bn_act_proj_nhwc(const float *x, const float *w_okkc, const float *scale, const float *shift,
                              int relu, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                              int pad, int dil, const float *proj_w, int proj_n, float *proj_out,
                              float *out, int64_t out_pixel_stride, void *stream);

/* ---- D8: grouped 3x3 convolution + BatchNorm(eval) + ReLU, NHWC fp32, fp32 matrix cores (16x16x4 tiles) -----
 * replaces Conv2d(w, w, 3, stride, padding=1, groups=w/group_w) -> BatchNorm2d -> ReLU of the RegNet bottleneck
 *          Bottleneck.__init__ / forward              empanada/models/encoders/regnet.py:59-71,88-104
 *          conv_bn_act(groups=...)                    empanada/models/blocks.py:121-171
 * out[p, co] = act(acc * scale[co] + shift[co]), co = grp * group_w + j; acc = sum over the 9 taps and the group's
 * group_w input channels of x[n, oy*stride - 1 + ky, ox*stride - 1 + kx, grp*group_w + c] * w_okkc[co, ky, kx, c],
 * evaluated as ONE fp32 fma chain from +0: taps in raster order; per tap chunks of CK = emp_gconv_chunk(group_w)
 * channels ascending (24, 16 or 8: the largest that divides group_w); per chunk 8-channel slabs j ascending; per
 * slab e = 0, 1; per e the channels 8j + 2kq + e, kq = 0..3 (one v_mfma_f32_16x16x4_f32 = an fmaf chain over its four
 * k's).  Taps outside the image enter as x = 0.  group_w: a multiple of 8 in 8..128; stride 1 or 2.
 * x: base + pixel * x_pixel_stride + channel (0 means groups * group_w), w_okkc: (groups * group_w, 3, 3, group_w) =
 * the Conv2d weight permuted; out likewise with out_pixel_stride.  x, w, out 16-byte aligned, strides % 4 == 0. */
int emp_gconv_chunk(int group_w);
int emp_gconv3x3_bn_act_nhwc(const float *x, int64_t x_pixel_stride, const float *w_okkc, const float *scale,
                             const float *shift, int relu, int N, int H, int W, int groups, int group_w,
                             int stride, float *out, int64_t out_pixel_stride, void *stream);

/* ---- D5: Winograd F(2x2, 3x3) convolution in three calls (3x3, stride 1, padding == dilation) ---------
 * replaces the dilated 3x3 Conv2d -> BatchNorm2d -> ReLU of ASPP and of the dilated ResNet stage
 *          (empanada/models/decoders/aspp.py:22-46, encoders/resnet.py:110-128) where Cin is large.
 * A convolution with dilation d is d*d plain 3x3 convolutions on the sub-grids (y mod d, x mod d); a tile is a
 * 2x2 block of outputs of one sub-grid and reads a 4x4 patch with pixel spacing d.
 * tiles (DEVICE, (T, 3) int32): image index n and (y, x) of the patch's top-left pixel (may be negative: padding);
 *   its outputs are the pixels (y + d + a*d, x + d + b*d), a, b in {0, 1}, that lie inside the image.
 * 1. emp_wino_input_transform: V[p = 4*u + v, t, c] = (B^T patch B)[u, v], B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0;
 *    0 1 0 -1], evaluated as: columns first (t_a0 = d_a0 - d_a2, t_a1 = d_a1 + d_a2, t_a2 = d_a2 - d_a1,
 *    t_a3 = d_a1 - d_a3), then the same combination over rows.  x (N,H,W,C) fp32, V (16, T, C).
 * 2. emp_gemm_nt_batched: C[b] (M, N) = A[b] (M, K) * B[b] (N, K)^T for b < batch, on the fp32 matrix cores with
 *    the fma-chain order of emp_conv_bn_act_nhwc (slabs of S = emp_conv_k_slab(M, N, batch, 0)).  Here M = T, K = Cin, N = Cout,
 *    batch = 16, B = the transformed filters U[p, co, c] = (G g G^T)[u, v] prepared by the host.
 * 3. emp_wino_output_transform: Y = A^T M A per tile, A^T = [1 1 1 0; 0 1 -1 -1], rows first
 *    (s_0 = (m_0 + m_1) + m_2, s_1 = (m_1 - m_2) - m_3), then columns likewise; then the epilogue of
 *    emp_bn_act_nhwc (scale, shift optional) and the scatter to out (N,H,W,Cout), pixel stride out_pixel_stride
 *    (0 means Cout).                                                                                          */
int emp_wino_input_transform(const float *x, int N, int H, int W, int C, int dil, const int32_t *tiles,
                             int64_t T, float *V, void *stream);
int emp_gemm_nt_batched(const float *A, const float *B, int batch, int64_t M, int N, int K, float *C,
                        void *stream);
/* Steps 1 + 2 in one launch: the GEMM's loader fetches, per tile and position, the four patch pixels B^T d B
 * combines and applies the transform on the way into LDS (same roundings as emp_wino_input_transform), so V is
 * never written.  Mw (16, T, Cout).  N*H*W*Cin < 2^31.                                                       */
int emp_wino_gemm_fused(const float *x, int N, int H, int W, int Cin, int dil, const int32_t *tiles,
                        int64_t T, const float *U, int Cout, float *Mw, void *stream);
int emp_wino_output_transform(const float *Mw, const int32_t *tiles, int64_t T, int N, int H, int W,
                              int Cout, int dil, const float *scale, const float *shift, int relu,
                              float *out, int64_t out_pixel_stride, void *stream);

/* ---- D5b: Winograd F(4x4, 3x3): the same three steps with 6x6 patches, 36 positions, 4x4 outputs per tile ----
 * tiles: (T, 3) int32 (n, y, x) of the patch origin; outputs are (y + d + a*d, x + d + b*d), a, b in 0..3.
 * Input transform  (B^T d B, columns first then rows), per 6-vector d:
 *   r0 = (4 d0 - 5 d2) + d4;  r1 = (d3 + d4) - 4 (d1 + d2);  r2 = (d4 - d3) + 4 (d1 - d2);
 *   r3 = (d4 - d2) + 2 (d3 - d1);  r4 = (d4 - d2) + 2 (d1 - d3);  r5 = (4 d1 - 5 d3) + d5
 * Output transform (A^T m A, rows first then columns), per 6-vector m:
 *   s0 = ((m0 + m1) + m2) + (m3 + m4);  s1 = (m1 - m2) + 2 (m3 - m4);  s2 = (m1 + m2) + 4 (m3 + m4);
 *   s3 = ((m1 - m2) + 8 (m3 - m4)) + m5
 * every operation one fp32 rounding.  V (36, T, C), position p = 6u + v; GEMMs: emp_gemm_nt_batched with batch 36
 * and B = U (36, Cout, Cin) = fp32(G g G^T evaluated in fp64) prepared by the host.  Rounding error is about 10x
 * that of the direct form (1.3e-6 * sum|x||w| measured at K = 18432): stated test tolerance 2e-5 * sum|x||w|.   */
int emp_wino4_input_transform(const float *x, int N, int H, int W, int C, int dil, const int32_t *tiles,
                              int64_t T, float *V, void *stream);
int emp_wino4_output_transform(const float *Mw, const int32_t *tiles, int64_t T, int N, int H, int W,
                               int Cout, int dil, const float *scale, const float *shift, int relu,
                               float *out, int64_t out_pixel_stride, void *stream);

/* ---- D5c: Winograd F(3x3, 3x3) (points 0, 1, -1, 2, inf): 5x5 patches, 25 positions, 3x3 outputs per tile -----
 * tiles as in D5 with outputs (y + d + a*d, x + d + b*d), a, b in 0..2.  Transforms: r[u] = left fold, over the
 * non-zero entries c of row u in ascending index, of fl(c * d[a]) with fp32 adds (columns first, then rows for
 * the input; rows first, then columns for the output):
 *   B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1],  A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1].
 * V (25, T, C), p = 5u + v; GEMMs: emp_gemm_nt_batched, batch 25, U (25, Cout, Cin) = fp32(G g G^T in fp64),
 * G = [1/2 0 0; -1/2 -1/2 -1/2; -1/6 1/6 -1/6; 1/6 1/3 2/3; 0 0 1].  Stated tolerance as D5b.                    */
int emp_wino3_input_transform(const float *x, int N, int H, int W, int C, int dil, const int32_t *tiles,
                              int64_t T, float *V, void *stream);
int emp_wino3_output_transform(const float *Mw, const int32_t *tiles, int64_t T, int N, int H, int W,
                               int Cout, int dil, const float *scale, const float *shift, int relu,
                               float *out, int64_t out_pixel_stride, void *stream);

/* ---- D6: 1x1 convolution to 1..4 output channels on NHWC fp32 activations (last layer of every head) ----
 * replaces the final nn.Conv2d(nin, n_classes, 1, bias=True) of PanopticDeepLabHead   empanada/models/heads.py:9-19
 * out[n, co, r] = bias[co] + sum_c x[n*HW + r, c] * w[co, c]; evaluated per wave lane l as an fp32 fma chain from +0
 * over its channel groups (channels 4g..4g+3 of groups g = l, l+64, ...), then a butterfly sum over the 64 lanes
 * (pairs at lane distance 32, 16, 8, 4, 2, 1; each step one fp32 add), then + bias.  x (n_pixels, C) NHWC,
 * w (Cout, C), out planar (N, Cout, pixels_per_image).  C % 4 == 0, C <= 1024, Cout <= 4.                    */
int emp_pointwise_out_nhwc(const float *x, const float *w, const float *bias, int64_t n_pixels,
                           int64_t pixels_per_image, int C, int Cout, float *out, void *stream);

/* ---- D7: BatchNorm(eval) + ReLU + MaxPool2d(3, stride 2, padding 1) on NHWC fp32 (ResNet stem) ------------
 * replaces bn1 -> relu -> maxpool of ResNet.forward             empanada/models/encoders/resnet.py:217-222
 * y[n, oy, ox, c] = max over the 3x3 window (taps outside the image skipped) of max(x*scale[c] + shift[c], 0),
 * multiply and add separate fp32 roundings.  x (N,H,W,C), y (N, (H-1)/2+1, (W-1)/2+1, C), C % 4 == 0.         */
int emp_bn_relu_maxpool_nhwc(const float *x, const float *scale, const float *shift, int N, int H, int W,
                             int C, float *y, void *stream);

/* ---- D9: the whole ResNet stem on a one-channel image: conv 7x7 / 2 + BatchNorm(eval) + ReLU + MaxPool 3x3 / 2 ----
 * replaces conv1 -> bn1 -> relu -> maxpool of ResNet.forward     empanada/models/encoders/resnet.py:186-188,217-222
 * conv[n, oy, ox, co] = ONE fp32 fma chain from +0 over the 49 taps in raster order of x[n, 2oy-3+ky, 2ox-3+kx] *
 * w_tc[7ky+kx, co] (taps outside the image enter as 0; evaluated on the vector ALUs, two channels per v_pk_fma_f32);
 * y = D7 of conv: max over the 3x3 window (outside the image: skipped) of max(conv*scale[co] + shift[co], 0).
 * x (N, H, W) fp32, w_tc (49, 64) = the Conv2d(1, 64, 7) weight as [tap][cout], 16-byte aligned,
 * y (N, PH, PW, 64) NHWC with OH = (H-1)/2+1, PH = (OH-1)/2+1 (likewise W).  The half-resolution activation is never
 * written.                                                                                                        */
int emp_stem_conv7_bn_relu_maxpool(const float *x, const float *w_tc, const float *scale, const float *shift,
                                   int N, int H, int W, float *y, void *stream);

/* ---- D2: logits -> probabilities ---------------------------------------------------------------------------
 * replaces logits_to_prob                                         empanada/inference/engines.py:22-30
 * C == 1: prob = 1 / (1 + expf(-x)) (negation exact, add and divide separate fp32 roundings);
 * C  > 1: softmax over the channel axis: m = max_c x_c, e_c = expf(x_c - m), s = sum of e_c in ascending c from +0,
 *         prob_c = e_c / s.  expf is the device library's (torch's GPU and CPU kernels use their own: results agree within
 *         2 ulp, the hardening decision `p >= thr` on logits within 1e-3 of logit(thr) in all but a few of 2 * 10^6
 *         cases -- tests/test_pipeline_gpu.py).  logits, prob: planar (N, C, HW) fp32; may alias.               */
int emp_logits_to_prob(const float *logits, int N, int C, int64_t HW, float *prob, void *stream);

/* ---- P1 + P2: recursive median over a resident stack, fused with hardening ----------------
 * replaces _MedianQueue.get_next/get_median/end   empanada/inference/engines.py:47-90
 *          _harden_seg / harden_seg               engines.py:114-121, inference/patterns.py:242-251
 * prob  (D, C, HW) fp32 probabilities.  For slice s in [m, D-m) (m = ks/2) the output is the
 * median of {out[s-m..s-1], prob[s..s+m]} (the reference writes the median back into its queue,
 * which makes the filter recursive); the first and last m slices pass through.  Requires
 * D >= ks or ks == 1 (the reference loses slices for shorter stacks; the host mirrors that).
 * out_sem (D, HW) u8: C == 1 -> (p >= thr), C > 1 -> argmax over the filtered channels.
 * out_prob (D, C, HW) fp32 or NULL: the filtered probabilities.                                */
int emp_median_harden_stack(const float *prob, int D, int C, int64_t HW, int ks, float thr,
                            uint8_t *out_sem, float *out_prob, void *stream);

/* One streaming step of the same filter: median over ks slices of n floats each.
 * slices_host: host array of ks device pointers; out may alias any of them.
 * replaces _MedianQueue.get_median                 engines.py:59-66                             */
int emp_median_step(const float *const *slices_host, int ks, int64_t n, float *out, void *stream);

/* Harden only (no median).  prob (D, C, HW) -> out_sem (D, HW) u8.                              */
int emp_harden(const float *prob, int D, int C, int64_t HW, float thr, uint8_t *out_sem,
               void *stream);

/* ---- P3: centre detection ------------------------------------------------------------------
 * replaces find_instance_center                    empanada/inference/postprocess.py:38-76
 * hmp (D, h, w) fp32.  A pixel is a centre when v > thr, v > 0 and v equals the maximum of
 * threshold(v) over the k x k window rows y-k/2 .. y-k/2+k-1 (same for x; -inf padding).
 * out_idx (D, cap) int32 flat indices y*w+x, ascending (raster order) per slice;
 * out_count (D) int32 = number found (may exceed cap: then only `cap` are stored, unordered
 * selection -- the caller must treat count > cap as an error).  cap <= EMP_MAX_CENTERS.         */
int emp_find_centers(const float *hmp, int D, int h, int w, float thr, int k, int cap,
                     int32_t *out_idx, int32_t *out_count, void *stream);

/* ---- P4: nearest-centre pixel grouping -----------------------------------------------------
 * replaces group_pixels / chunked_pixel_grouping   postprocess.py:78-169
 * offsets (D, 2, h, w) fp32 (dy, dx) in full-resolution pixel units; step = 1 or 4.
 * id = 1 + index of the first centre with the strictly smallest
 *      d = sqrtf(fmaf(dx, dx, dy*dy)),  dy = step*cy - (step*y + off_y), dx likewise
 * (the rounding torch.norm performs on the reference's CPU path); when K > 20 a pixel whose
 * every d >= 1e5 keeps id 0 (chunked path, :97-111).  K = min(count, cap); K == 0 -> ids 0.
 * sem (D, h, w) u8 or NULL: when given (same resolution as the offsets), pixels whose class bit is
 * clear in thing_mask are not voted on and get id 0 -- every consumer multiplies the ids by the
 * thing mask anyway (postprocess.py:221, engines.py:280-285), so results downstream are unchanged.
 * work: emp_group_work_elems(D, cap) floats (the centres as fp32 coordinates, read with scalar loads).
 * out_ids (D, h, w) uint16.                                                                     */
int64_t emp_group_work_elems(int D, int cap);
int emp_group_pixels(const int32_t *ctr_idx, const int32_t *ctr_count, int cap,
                     const float *offsets, int D, int h, int w, int step, const uint8_t *sem,
                     uint32_t thing_mask, float *work, uint16_t *out_ids, void *stream);

/* ---- P4b + P5: instance cells -> panoptic labels -------------------------------------------
 * replaces get_instance_cells (nearest upsample)   engines.py:257-275
 *          get_panoptic_seg                        engines.py:277-292, patterns.py:253-277
 *          merge_semantic_and_instance             postprocess.py:223-296
 * sem (D, H, W) u8 class ids (< n_classes <= EMP_MAX_CLASSES); ids (D, H/up, W/up) u16 in 0..cap
 * (nearest-upsampled by `up` on the fly); thing_mask bit c set = class c is a thing.
 * Per instance id (ascending) with >= 1 thing pixel: class = most frequent class among its thing
 * pixels (ties -> smallest), new id = per-class counter from 1, pan = class*div + new id.
 * Non-thing classes: pixels with instance 0 get class*div when their count >= stuff_area.
 * Everything else = void_label.
 * work: int32 workspace of emp_fuse_work_elems(D, cap, n_classes) elements (zeroed by the call).
 * Exactly one of out_pan_u32 / out_pan_i64 (D, H, W) must be non-NULL.                          */
int64_t emp_fuse_work_elems(int D, int cap, int n_classes);
int emp_fuse_panoptic(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up,
                      int cap, int n_classes, uint32_t thing_mask, int64_t label_divisor,
                      int64_t stuff_area, int64_t void_label, int32_t *work,
                      uint32_t *out_pan_u32, int64_t *out_pan_i64, void *stream);
/* The two halves of emp_fuse_panoptic, for callers that keep the per-slice tables (and so that each pass can be
 * timed on its own): emp_fuse_lut = histogram + per-slice label table into `work`; emp_fuse_apply = label pass.   */
int emp_fuse_lut(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up, int cap,
                 int n_classes, uint32_t thing_mask, int64_t label_divisor, int64_t stuff_area,
                 int32_t *work, void *stream);
int emp_fuse_apply(const uint8_t *sem, const uint16_t *ids, int D, int H, int W, int up, int cap,
                   int n_classes, uint32_t thing_mask, int64_t label_divisor, int64_t void_label,
                   const int32_t *work, uint32_t *out_pan_u32, int64_t *out_pan_i64, void *stream);

/* ---- R1-R3: run extraction + 8-connected components on runs --------------------------------
 * replaces connected_components                    empanada/inference/rle.py:18-24
 *          pan_seg_to_rle_seg                      rle.py:26-86
 *          rle_encode                              empanada/array_utils.py:209-235
 * Step 1  emp_runs_count: row_counts[d*H+y] = number of maximal constant non-zero spans in row y
 *         of slice d of pan (D, H, W) u32.
 * Step 2  emp_exclusive_scan_i32: out[0..n] (n+1 entries) exclusive prefix sum; tmp holds
 *         emp_scan_tmp_elems(n) int32.
 * Step 3  emp_runs_extract: SoA run table in raster order, run i of row r sits at
 *         row_offsets[r] + i: r_start (flat y*W+x0 inside the slice), r_len, r_val.
 * Step 4  emp_runs_label: union-find over runs.  Classes whose bit is set in cc_mask are split
 *         into 8-connected components of equal value and renumbered class*div + k, k = 1.. in
 *         raster order of the component's first pixel, per slice; runs of other classes keep
 *         their value as label (one instance per distinct value).  Outputs, per run:
 *         r_comp (int32) = dense component index over the whole stack (ordered by first run),
 *         and per component: c_slice, c_label (int64), c_area (int64), c_box (4 x int32:
 *         y0, x0, y1, x1 half-open), c_first (first run).  n_comp_out (device int32[1]).
 *         work: int32 workspace of emp_runs_label_work_elems(n_runs) elements.               */
int emp_runs_count(const uint32_t *pan, int D, int H, int W, int32_t *row_counts, void *stream);
int64_t emp_scan_tmp_elems(int64_t n);
int emp_exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int32_t *tmp, void *stream);
int emp_runs_extract(const uint32_t *pan, int D, int H, int W, const int32_t *row_offsets,
                     int32_t *r_start, int32_t *r_len, uint32_t *r_val, void *stream);
int64_t emp_runs_label_work_elems(int64_t n_runs);
int emp_runs_label(const int32_t *r_start, const int32_t *r_len, const uint32_t *r_val,
                   const int32_t *row_offsets, int64_t n_runs, int D, int H, int W,
                   int64_t label_divisor, uint32_t cc_mask, int32_t *work, int32_t *r_comp,
                   int32_t *c_slice, int64_t *c_label, int64_t *c_area, int32_t *c_box,
                   int32_t *c_first, int32_t *n_comp_out, void *stream);

/* ---- M2: overlaps between components of consecutive slices ---------------------------------
 * replaces rle_intersection / intersection_from_ranges  array_utils.py:340-403 for the pairs the
 *          matcher asks for (rle_matcher                 inference/matcher.py:198-210)
 * For every run of slice d and every run of slice d+1 in the same row whose x-spans overlap and
 * whose values r_val are of the same class (value / label_divisor), appends (comp_a, comp_b, overlap) to out_triplets
 * (cap_triplets x 3 int32, unordered; duplicates of a pair must be summed by the caller).
 * n_out (device int32[1]) counts all triplets found (may exceed the capacity -> error).        */
int emp_runs_overlap_next(const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp,
                          const uint32_t *r_val, const int32_t *row_offsets, int64_t n_runs, int D,
                          int H, int W, int64_t label_divisor, int32_t *out_triplets,
                          int64_t cap_triplets, int32_t *n_out, void *stream);

/* ---- R3: run-length encoding of index lists ----------------------------------------------------
 * replaces rle_encode / rle_decode                empanada/array_utils.py:209-252
 * emp_rle_decode: offsets = exclusive prefix sum of runs (n_runs entries); out_indices receives
 *   starts[i] .. starts[i]+runs[i]-1 at offsets[i].
 * emp_rle_encode: indices (n, ascending) -> maximal runs of consecutive values; out_starts / out_runs hold
 *   up to n entries, n_runs_out (device int32[1]) the count; work: emp_rle_encode_work_elems(n) int32.        */
int emp_rle_decode(const int64_t *starts, const int64_t *runs, const int64_t *offsets, int64_t n_runs,
                   int64_t *out_indices, void *stream);
int64_t emp_rle_encode_work_elems(int64_t n);
int emp_rle_encode(const int64_t *indices, int64_t n, int32_t *work, int64_t *out_starts,
                   int64_t *out_runs, int32_t *n_runs_out, void *stream);

/* ---- M1: box screening ------------------------------------------------------------------------
 * replaces _box_iou / box_iou                     empanada/array_utils.py:144-207 (which pairs exist)
 *          bounding_box_screening                  empanada/consensus.py:197-231
 * boxes (n, 2*ndim) int32, half-open (lo..., hi...); ndim 2 or 3.  Emits every (a, b) whose
 * intersection is strictly positive along every axis; pairs with src_a[a] == src_b[b] are skipped
 * when both src arrays are given; upper_only != 0 keeps only b > a (self-screening without
 * duplicates).  out_pairs (cap, 2) int32 in arbitrary order; n_out (device int32[1]) counts all
 * pairs found (may exceed cap -> caller retries).                                                */
int emp_box_pairs(const int32_t *boxes_a, int64_t na, const int32_t *boxes_b, int64_t nb, int ndim,
                  const int32_t *src_a, const int32_t *src_b, int upper_only, int32_t *out_pairs,
                  int64_t cap, int32_t *n_out, void *stream);

/* ---- M2/C1: intersection of arbitrary RLE pairs --------------------------------------------
 * replaces rle_intersection                        array_utils.py:371-403 (the exact sweep of
 *          intersection_from_ranges :340-369, also for malformed / overlapping runs), as used by
 *          rle_iou :405-429 in object_iou_graph    empanada/consensus.py:276-285
 * Instance i owns runs [inst_off[i], inst_off[i+1]) of (starts, lens) int64; each instance's runs
 * must already be STABLY sorted by start (emp_sort_u64_i32 with key = instance<<40 | start does
 * it).  pairs (n_pairs, 2) int32 instance indices (a, b); out_inter (n_pairs) int64.            */
int emp_rle_pair_intersections(const int64_t *starts, const int64_t *lens, const int64_t *inst_off,
                               const int32_t *pairs, int64_t n_pairs, int64_t *out_inter,
                               void *stream);

/* ---- stable key/value radix sort (rocPRIM via hipCUB), used to order run tables --------------
 * keys 64-bit unsigned, values int32; bits [begin_bit, end_bit) are compared.
 * work: emp_sort_work_bytes(n) bytes.                                                          */
int64_t emp_sort_work_bytes(int64_t n);
int emp_sort_u64_i32(const uint64_t *keys_in, uint64_t *keys_out, const int32_t *vals_in,
                     int32_t *vals_out, int64_t n, int begin_bit, int end_bit, void *work,
                     int64_t work_bytes, void *stream);

/* ---- C3: range voting / joining -------------------------------------------------------------
 * replaces vote_by_ranges / rle_voting / split_range_by_votes / extend_range / join_ranges
 *          empanada/array_utils.py:457-671
 * n ranges [starts[i], ends[i]) int64 (< 2^40), each tagged with a group id grp[i] in
 * [0, n_groups), in any order.  For each group emits, ascending, the maximal ranges covered by
 * at least vote_thr of its input ranges (vote_thr == 1: the union; touching ranges merge -- the
 * reference's sweeps reduce to exactly this coverage count).
 * out_ranges (n, 2) int64 receives all groups back to back; group g owns rows
 * [out_off[g], out_off[g+1]) (out_off: n_groups+1 int32).
 * work: emp_vote_work_bytes(n) bytes.                                                          */
int64_t emp_vote_work_bytes(int64_t n);
int emp_vote_ranges(const int64_t *starts, const int64_t *ends, const int32_t *grp, int64_t n,
                    int n_groups, int vote_thr, void *work, int64_t work_bytes,
                    int64_t *out_ranges, int32_t *out_off, void *stream);

/* ---- R4 / Z1: paint runs into a label volume ------------------------------------------------
 * replaces numpy_fill_instances                    array_utils.py:725-737
 *          fill_func / zarr_fill_instances         empanada/zarr_utils.py:49-58,88-175
 * Runs (starts, lens int64 into the flat volume) carry an order index (position of their
 * instance in dict order) and ids[order] is painted; where runs of different instances overlap
 * the later instance wins, as in the reference's sequential fill.  Ids must be < 2^31; runs whose
 * id is 0 are skipped (an instance deleted by a filter neither paints nor shadows).              */
int emp_fill_runs_u32(uint32_t *vol, int64_t n_vox, const int64_t *starts, const int64_t *lens,
                      const int32_t *order, int64_t n_runs, const uint32_t *ids, void *stream);
int emp_fill_runs_u8(uint8_t *vol, int64_t n_vox, const int64_t *starts, const int64_t *lens,
                     int64_t n_runs, uint8_t value, void *stream);
/* Fill a slab (n_slices, HW) of an xy-stack volume straight from the run table of emp_runs_extract /
 * emp_runs_label: run i paints value[r_comp[i]] at slice c_slice[r_comp[i]] - slice0 (runs whose value
 * is 0 or whose slice falls outside the slab are skipped).  Runs of one stack never overlap.
 * replaces update_trackers + numpy_fill_instances for stack mode (scripts/pdl_inference3d.py:190-233). */
int emp_fill_table_u32(uint32_t *vol, int64_t HW, int n_slices, int slice0, const int32_t *r_start,
                       const int32_t *r_len, const int32_t *r_comp, const int32_t *c_slice,
                       const uint32_t *value, int64_t n_runs, void *stream);

/* ---- T1 (yz plane): runs of a yz stack -> dense (Z, Y, X) label volume -----------------------------
 * replaces the per-pixel decode + sort + re-encode of InstanceTracker.update/finish for axis 'yz'
 *          empanada/inference/tracker.py:83-88,110-113
 * The stack has X slices of (Z, Y) pixels; run i covers pixels r_start[i] .. +r_len[i] (flat z*Y+y) of slice
 * c_slice[r_comp[i]]; voxel (z, y, slice) receives value[r_comp[i]] (0 = skip).  vol must be zeroed.
 * The 3D run-length encoding along x is then obtained with emp_runs_count / emp_runs_extract on vol.   */
int emp_scatter_yz_u32(uint32_t *vol, int Z, int Y, int X, const int32_t *r_start, const int32_t *r_len,
                       const int32_t *r_comp, const int32_t *c_slice, const uint32_t *value,
                       int64_t n_runs, void *stream);

/* ---- M2 (cont.): one triplet per COMPONENT pair -------------------------------------------------------------
 * emp_runs_overlap_next emits one (comp_a, comp_b, pixels) triplet per overlapping run pair; the matcher needs the
 * intersection of two instances (rle_intersection, array_utils.py:371-403), i.e. the sum over all run pairs of the two
 * components.  emp_triplets_reduce sorts the triplets by (a, b) and sums equal pairs on the device: `out` (<= n
 * triplets, ascending (a, b)), n_out device int32.  work: emp_triplets_reduce_work_bytes(n) bytes.               */
int64_t emp_triplets_reduce_work_bytes(int64_t n);
int emp_triplets_reduce(const int32_t *triplets, int64_t n, void *work, int64_t work_bytes, int32_t *out,
                        int32_t *n_out, void *stream);

/* ---- T1 on the device: run table of a plane's slices -> per-instance 3D runs sorted by (instance, start) ------
 * replaces InstanceTracker.update / finish        empanada/inference/tracker.py:61-123
 *          to_coords3d                             empanada/inference/tracker.py:25-38
 * and feeds merge_objects_from_trackers (empanada/consensus.py:348-469) and the fill without a host round trip.
 * A 3D run is (key, len): key = instance << 40 | flat (z, y, x) start of the (Z, Y, X) volume; len in voxels.
 * comp_inst[c] = instance index of component c of the run table (its position in the tracker's dict order), -1 =
 * skip (halo slice, removed).  inst_base is added to every instance index (planes concatenated for the consensus).
 *
 * emp_track_lift  axis 0 (xy: plane (H, W) = (Y, X), slices along z): start3d = start + slice * Y * X;
 *                 axis 1 (xz: plane (H, W) = (Z, X), slices along y): start3d = (start / W) * Y * X + slice * X +
 *                 start % W -- only the START is mapped and the length kept, reproducing the row wrap of
 *                 tracker.py:78-82.  Runs of one instance that are contiguous inside a slice are merged first (what
 *                 rle_encode over the instance's pixels yields, array_utils.py:209-235).  slice = c_slice + slice0.
 *                 Outputs hold at most n_runs entries; n_out is a device int32.  work: emp_track_work_elems(n_runs)
 *                 int32.
 * emp_track_lift_yz  for the yz plane the tracker is the run-length encoding along x of the dense labelling
 *                 (tracker.py:83-88,110-113): scatter instance + 1 with emp_scatter_yz_u32 into a (Z, Y, Xl) volume,
 *                 extract its row runs (emp_runs_count / _extract) and pass them here: start3d = row * X + x0 + x.
 * emp_track_sort  stable radix sort by key; with merge_touching, runs of one instance whose previous end equals their
 *                 start are joined (np.sort + rle_encode in tracker.finish).  Any of out_key / out_st (= key's low 40
 *                 bits) may be NULL.  work: emp_track_sort_work_bytes(n) bytes.
 * emp_track_offsets  CSR offsets of the instances in a sorted key array: out_off[k] = first run with instance >= k,
 *                 k = 0 .. n_inst.
 * emp_track_expand  out[i] = obj_val[instance that owns run i] (vote groups, fill order) from CSR offsets.
 * emp_track_clip  the part of every run inside the flat interval [lo, hi): a rank's z-slab of the output volume
 *                 (chunk_ranges, empanada/zarr_utils.py:11-47, splits runs at chunk borders the same way).
 *                 work: emp_track_work_elems(n) int32.                                                        */
int64_t emp_track_work_elems(int64_t n_runs);
int emp_track_lift(int axis, const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp,
                   const int32_t *c_slice, const int32_t *comp_inst, int64_t n_runs, int H, int W, int Y, int X,
                   int slice0, int64_t inst_base, int32_t *work, uint64_t *out_key, int64_t *out_len,
                   int32_t *n_out, void *stream);
/* emp_tile_lift  C5, Tiler.translate_rle_seg (empanada/inference/tile.py:122-168) for all slices of one tile: runs of
 *                 every object merged as rle_encode of its flat TILE indices merges them, start mapped into the image
 *                 frame ((start / tw + y0) * X + start % tw + x0), length kept; 2D positions, the slice stays with the
 *                 object: key = (inst_base + comp_inst[comp]) << 40 | start2d.  work: emp_track_work_elems(n_runs). */
int emp_tile_lift(const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp, const int32_t *c_slice,
                  const int32_t *comp_inst, int64_t n_runs, int tw, int X, int y0, int x0, int64_t inst_base,
                  int32_t *work, uint64_t *out_key, int64_t *out_len, int32_t *n_out, void *stream);
int emp_track_lift_yz(const int32_t *row_offsets, const int32_t *r_start, const int32_t *r_len,
                      const uint32_t *r_val, int64_t n_rows, int64_t n_runs, int Xl, int X, int x0,
                      int64_t inst_base, uint64_t *out_key, int64_t *out_len, void *stream);
int64_t emp_track_sort_work_bytes(int64_t n);
int emp_track_sort(const uint64_t *key_in, const int64_t *len_in, int64_t n, int merge_touching, void *work,
                   int64_t work_bytes, uint64_t *out_key, int64_t *out_st, int64_t *out_len, int32_t *n_out,
                   void *stream);
int emp_track_offsets(const uint64_t *keys_sorted, int64_t n, int64_t n_inst, int64_t *out_off, void *stream);
int emp_track_expand(const int64_t *off, const int32_t *obj_val, int64_t n_obj, int64_t n_runs, int32_t *out,
                     void *stream);
int emp_track_clip(const uint64_t *key, const int64_t *len, int64_t n, int64_t lo, int64_t hi, int32_t *work,
                   uint64_t *out_key, int64_t *out_len, int32_t *n_out, void *stream);

/* ---- M4/M5 (host): slice-to-slice label propagation over component tables, plain C++ on the CPU ---------
 * replaces RLEMatcher.__call__ driven by forward_matching / backward_matching
 *          empanada/inference/matcher.py:262-323, inference/patterns.py:60-121
 * for the whole-stack path, where the GPU has already reduced every slice to connected components
 * (emp_runs_label) and their overlaps with the next slice (emp_runs_overlap_next).  One call per class.
 * Components sorted by (slice, cc label): slice t owns [bounds[t], bounds[t+1]); pa / pb are positions inside
 * slice t / t+1 of the overlap triplets [tb_bounds[t], tb_bounds[t+1]).  IoU fp64, IoA fp32, thresholds and merge
 * rule as in the reference; the Hungarian step runs only when an overlap matrix has a row or column with two
 * non-zeros and is delegated to `lsap` (the caller passes scipy.optimize.linear_sum_assignment(maximize=True),
 * the routine the reference calls, so ties break identically): lsap(iou row-major, n_rows, n_cols, rows_out,
 * cols_out) -> number of pairs or -1.
 * comp_final[n]: final label per component (sorted order); seen_labels / n_seen: labels in order of first update
 * when slices are visited last to first (the tracker's dict order).
 * Returns 0; 1 where the reference raises ValueError (ioa_thr <= 0 and an empty target slice); 2 lsap failed.  */
typedef int64_t (*emp_lsap_fn)(const double *iou, int64_t n_rows, int64_t n_cols, int64_t *rows_out,
                               int64_t *cols_out);
/* lsap == NULL: the library's own restatement of that routine (emp_lsap_maximize: shortest augmenting paths after
 * Crouse 2016 with scipy 1.15's column order and tie rule, checked against scipy itself by
 * tests/test_sharded_gloo.py::test_native_lsap_equals_scipy) -- no Python in the loop. */
int64_t emp_lsap_maximize(const double *cost, int64_t n_rows, int64_t n_cols, int64_t *rows_out, int64_t *cols_out);
int emp_chain_class(int64_t D, const int64_t *bounds, const int64_t *comp_label, const int64_t *comp_area,
                    int is_thing, const int64_t *tb_bounds, const int64_t *pa, const int64_t *pb,
                    const int64_t *tv, int64_t class_id, int64_t label_divisor, double iou_thr, double ioa_thr,
                    emp_lsap_fn lsap, int64_t *comp_final, int64_t *seen_labels, int64_t *n_seen);

/* ---- D10: PointRend subdivision step (eval branch) ---------------------------------------------------------------
 * replaces, per render step of the exported models' `forward(x, render_steps, interpolate_ins)`:
 *   F.interpolate(x2, bilinear, align_corners=False)       empanada/models/point_rend.py:244-245
 *   calculate_uncertainty + get_uncertain_point_coords_on_grid (torch.topk)   point_rend.py:62-79,97-125
 *   point_sample (F.grid_sample, bilinear, align_corners=False, zeros) of features and coarse logits   :35-60,253-254
 *   StandardPointHead (Conv1d + ReLU x num_fc, predictor; coarse logits re-fed to every layer)          :138-190
 *   scatter_ of the point predictions into the upsampled logits                                          :258-265
 * emp_pr_upsample2x  logits (N, C, h, w) planar -> out (N, C, 2h, 2w) and uncertainty (N, 4hw): -|l| for C == 1,
 *                    second-largest minus largest logit otherwise.
 * emp_pr_topk        idx (N, k) int32 = pixel indices of the k largest uncertainties of every image: exact (radix
 *                    select); ties at the k-th value go to the lowest indices; order: the strictly larger ones in pixel
 *                    order, then the ties in pixel order.  work: emp_pr_topk_work_bytes(N, HW) bytes.
 * emp_pr_point_sample  features NHWC (N, Hf, Wf, CF), pixel stride feat_pixel_stride; coarse (N, C, Hf, Wf) planar;
 *                    point p = n * k + j sits at the centre of pixel idx[p] of the (H, W) grid.  Writes row p of the
 *                    MLP input matrix X0 (P, ld): channels [0, CF) sampled features, [CF, CF + C) sampled coarse logits,
 *                    [CF + C, ld) zeros; and the channels [CF, ld) of X1 (the next layer's input, whose first CF
 *                    channels the MLP layer writes).  ld % 16 == 0 so that emp_conv_bn_act_nhwc can consume it.
 * the MLP            emp_conv_bn_act_nhwc(x = X as (1, P, 1, ld), w = (Cout, 1, 1, ld) zero-padded, shift = bias,
 *                    relu, out = channel slice [0, Cout) of the other matrix): summation order of D4.
 * emp_pr_scatter     logits[n, c, idx[p]] = points[p * ld_points + c].                                            */
int emp_pr_upsample2x(const float *logits, int N, int C, int h, int w, float *out, float *uncertainty, void *stream);
int64_t emp_pr_topk_work_bytes(int N, int64_t HW);
int emp_pr_topk(const float *uncertainty, int N, int64_t HW, int k, void *work, int64_t work_bytes, int32_t *idx,
                void *stream);
int emp_pr_point_sample(const float *feat_nhwc, int64_t feat_pixel_stride, const float *coarse, int N, int Hf, int Wf,
                        int CF, int C, const int32_t *idx, int k, int H, int W, float *X0, float *X1, int ld,
                        void *stream);
int emp_pr_scatter(const float *points, int ld_points, const int32_t *idx, int N, int C, int k, int64_t HW,
                   float *logits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EMP_HIP_H */
