/* libfgn_hip.so -- C-ABI of the MI355X (gfx950) FGN inference kernels.
 *
 * The reference (tooHotSpot/FGN) has no native/FFI layer: its hot path is Python
 * (subprojects/sp02_omniiseg_fgn_mmdet/fgn.py:187-303) calling CUDA kernels inside the
 * mmdet / mmcv / torchvision / torch wheels.  Each entry point below replaces one of
 * those third-party kernel families at the call site cited; a maintainer binds them
 * with ctypes or any other FFI: plain pointers and sizes only, no
 * torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; tensors are fp32, NHWC
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing here
 *     allocates, frees or synchronises: the caller owns workspaces, calls are
 *     asynchronous and safe to capture in a hipGraph
 *   - `n_*_dev` arguments are optional device int32 counters: when non-NULL the kernel
 *     processes min(*n_dev, n) items, so data-dependent counts (proposals, detections)
 *     never round-trip through the host
 *   - return value: 0 on success, <0 for FGN_ERR_* (bad argument / unsupported shape),
 *     >0 for a hipError_t from the launch
 */
#ifndef FGN_HIP_H
#define FGN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGN_OK 0
#define FGN_ERR_SHAPE (-1)
#define FGN_ERR_ARG (-2)

int fgn_abi_version(void);

/* Profiler hook (no reference counterpart): arm the calling thread so that the NEXT convolution-family kernel it
 * launches stamps the two HIP events (hipEvent_t, created by the caller) with that kernel's own start and end -
 * the duration a rocprofv3 kernel trace reports.  NULL, NULL disarms. */
int fgn_profile_next_launch(void* start_event, void* stop_event);
/* Launch records that work inside a replayed hipGraph (no reference counterpart): while the calling thread is armed,
 * every launch of the dominant kernel (conv_pw_persist_kernel) is handed the next record of `records` (device memory,
 * `capacity` records of fgn_profile_stamp_words() x uint64, all zero except word 4 = ~0); each execution of such a
 * launch - every replay of a graph it was captured into - adds its span (first workgroup start -> last workgroup end,
 * 10 ns ticks of the constant 100 MHz clock) to the record: [1] sum, [3] executions, [4] shortest, [5] longest.  A
 * recorded launch pays one returning atomic per workgroup at its exit (~1 us at the end of the launch).  NULL disarms.
 * Returns the number of records handed out since the previous call (= launches recorded, in launch order). */
int fgn_profile_stamp_words(void);
int fgn_profile_stamps(void* records, int capacity);
/* Phase marks between streams that replay captured graphs (no reference counterpart; the pipelined serving loop of
 * INTEGRATION.md): fgn_phase_signal bumps *counter (device memory, int32) from inside an episode - captured into its
 * graph like any kernel; fgn_phase_wait holds `stream` until *counter - target >= 0 or timeout_us (<= 1 s) have passed. */
int fgn_phase_signal(int32_t* counter, void* stream);
int fgn_phase_wait(const int32_t* counter, int32_t target, int timeout_us, void* stream);

/* Implicit-GEMM convolution on the fp32 MFMA pipe with fused epilogue
 *   y = conv(x * in_scale?, w) * scale[c] + shift[c] (+ residual) (ReLU)
 * Replaces cuDNN conv + BN(eval) + ReLU of mmdet ResNet (fgn.py:212,215), RPNHead
 * (fgn_ag_rpn_head.py:48), shared_head ResLayer (fgn_roi_head.py:236,369,436), the two
 * halves of cls_reg_shared_conv (fgn_roi_head.py:270), FCNMaskHead convs and the 2x2
 * deconv (fgn_roi_head.py:380); in_scale fuses the guidance multiplies at
 * fgn_ag_rpn_head.py:44 and fgn_roi_head.py:379.
 *   x [n_img/a_img_div, H, W, Cin]   w_packed [cout_pad, KH, KW, Cin] (K padded to x32,
 *   cout_pad multiple of 128, zero rows)   y [n_img, Ho, Wo, Cout]
 *   scale/shift [Cout] or NULL; residual like y or NULL; in_scale [n_img, Cin] or NULL
 *   Cin must be a multiple of 32, or exactly 4 = the stem's NHWC4 image: three channels and a ZERO fourth one, in x
 *   (fgn_nchw3_to_nhwc4_f32) and in w_packed alike - the kernel does not multiply the fourth channel
 *   tile_hint 0 = auto, 1..4 = force a tile configuration, negative = no split-K,
 *   +100 = register-staged loader instead of LDS-DMA (tests)
 *   splitk_ws: optional workspace of fgn_conv2d_workspace_bytes() bytes; when given and the plain
 *   grid would under-fill the GPU, K is split over blockIdx.y into partial-tile slabs that are
 *   summed in slab order (bit-reproducible) before the epilogue.  NULL = never split.
 *   (The kernels behind this entry point that were measured and not adopted - Stream-K, a tile scheduler, other
 *   tile shapes - are not part of this library: tools/micro/conv_pw_experiments.inc.) */
size_t fgn_conv2d_workspace_bytes(int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                  int pad, int tile_hint);
/* Which kernel the dispatcher launches for a layer (tile*10 + mode; 41 = conv_igemm_dma_kernel<64,64,32,32,2,4,1>):
 * lets a profiler attribute a launch to the kernel name rocprofv3 reports.  No reference counterpart. */
int fgn_conv2d_kernel_id(int n_img, int H, int W, int Cin, int Cout, int cout_pad, int KH, int KW, int stride,
                         int pad, int a_img_div, int has_in_scale, int has_residual, int tile_hint);
int fgn_conv2d_nhwc_f32(const float* x, const float* w_packed, float* y, const float* scale,
                        const float* shift, const float* residual, const float* in_scale,
                        const int32_t* n_img_dev, int n_img, int H, int W, int Cin, int Cout,
                        int cout_pad, int KH, int KW, int stride, int pad, int a_img_div, int relu,
                        int tile_hint, float* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* Two 1x1 / stride 1 convolutions on the same rows summed in one K loop: y = x W1^T + x2 W2^T + shift (ReLU), w_packed
 * [cout_pad][Cin1 + Cin2] with both BatchNorm scales folded into its rows.  Replaces conv3 + bn3 and the shortcut
 * conv + bn + add + ReLU of the first Bottleneck of a stride-1 stage (mmdet ResNet layer1.0 under fgn.py:212,215):
 * no launch and no [rows, Cout] round trip for the shortcut.  x [rows,Cin1], y [rows,Cout]; x2 [x2_total_rows,Cin2]:
 * output row m reads row x2_rows[m] of it (device int32: the 1x1 / stride 2 shortcut of layer2.0 / layer3.0), or row m
 * when x2_rows is NULL (x2_total_rows = rows).  Cin1, Cin2 multiples of 32, Cout % 4 == 0. */
int fgn_conv1x1_dual_nhwc_f32(const float* x, const float* x2, const int32_t* x2_rows, int x2_total_rows,
                              const float* w_packed, float* y, const float* shift, int rows, int Cin1, int Cin2, int Cout,
                              int cout_pad, int relu, void* stream);

/* The GEMM-shaped launches of the path with their PRODUCTS on the bf16 matrix pipe (conv_pw_x3_kernel, csrc/conv_pw_x3.h;
 * gfx950's f32-input MFMA runs at 1/16 of the bf16 MFMA's rate and has no xf32 / TF32 form).  An f32 value is the exact
 * sum of three bf16 values (8 significant bits each); a product of two such sums is nine exact bf16 products, of which
 * the six that reach 2^-23 of |a b| are issued as bf16 MFMAs and accumulated in f32 - the operands, the accumulation,
 * the epilogue and the results are f32, and the error against fp64 is that of the f32 MFMA kernels
 * (tests/test_hip_conv.py::test_x3_*).  The same call sites as their f32 counterparts (fgn_conv2d_nhwc_f32 with a 1x1 /
 * stride 1 / unpadded kernel, fgn_conv1x1_dual_nhwc_f32, fgn_winograd_gemm_f32), same arguments, except that the weights
 * are given as their bf16-plane IMAGE (host-packed once: fgn_amd/ops.py::pack_x3):
 *   w_x3 [groups][K / 32][3 planes][cout_pad][32] bf16, plane 0 = w with the low 16 bits cleared, plane 1 the same of
 *   w - plane 0, plane 2 = w - plane 0 - plane 1; inside a K-tile chunk g (8 values) holds k = 4g..4g+3, 16+4g..16+4g+3 -
 *   the operand order of v_mfma_f32_16x16x32_bf16 under which ds_read_b128's lane groups read both operands without bank
 *   conflicts -, and the four 16-byte chunks of a 64-byte row are XOR-ed with tau[(n >> 2) & 3], tau = (0, 3, 2, 1).
 * fgn_x3_image_bytes = its size.  K >= 64, K % 32 == 0, Cout % 4 == 0, cout_pad % 128 == 0.
 * fgn_gemm_x3_f32: y[rows, Cout] = relu?(x[rows, K] W^T + shift + residual) directly (grouped: rows = n_groups *
 * grp_rows, image g for group g, first grp_valid rows of a group computed); bm 0 (= fgn_x3_row_tile) / 64 / 128 and
 * nterms 6 / 9 choose the kernel instance (tests, tools/x3_probe.py). */
size_t fgn_x3_image_bytes(int K, int npad, int n_groups);
int fgn_x3_row_tile(long long M, int Cout, int K, int grp_rows, int grp_valid);   /* 64 / 128: the kernel instance a launch of this shape runs on; 0: use the f32 entry point */
int fgn_gemm_x3_f32(const float* x, const void* w_x3, float* y, const float* shift, const float* residual, int rows, int K,
                    int Cout, int npad, int relu, int grp_rows, int grp_valid, int n_groups, int bm, int nterms,
                    void* stream);
int fgn_conv1x1_x3_nhwc_f32(const float* x, const void* w_x3, float* y, const float* scale, const float* shift,
                            const float* residual, const int32_t* n_img_dev, int n_img, int H, int W, int Cin, int Cout,
                            int cout_pad, int relu, void* stream);
int fgn_conv1x1_dual_x3_nhwc_f32(const float* x, const float* x2, const int32_t* x2_rows, int x2_total_rows,
                                 const void* w_x3, float* y, const float* shift, int rows, int Cin1, int Cin2, int Cout,
                                 int cout_pad, int relu, void* stream);
int fgn_winograd_gemm_x3_f32(const float* V, const void* U_x3, float* Mo, const int32_t* n_img_dev, int n_img,
                             int tiles_per_img, int t_pad, int Cin, int Cout, int cout_pad, int n_groups, void* stream);
/* The same GEMMs with THREE f16 MFMA products per f32 product (conv_pw_h2_kernel, csrc/conv_pw_h2.h; the default of the
 * host wrappers): an f32 value scaled by a power of two into the f16 range is h + l to within 2^-23 of itself (h = f16(x),
 * l = f16(x - h), round to nearest), products of f16 values are exact in f32, and h_a h_b + h_a l_b + l_a h_b leaves out
 * only l_a l_b <= 2^-22 |a b| (per product ~2^-24 |a b| rms: the order of an f32 FMA chain's own rounding); f32 accumulation, f32 in / out, the f32 kernels' error against fp64
 * (tests/test_hip_conv.py::test_h2_*).  The scales are powers of two (exact) and taken out again inside the kernel: the
 * weights' per output column at pack time (fgn_amd/ops.py::pack_h2), the activations' found by the kernel itself per wave
 * and output tile (the first non-zero K-tile of a tile sets it; a later K-tile that would leave the f16 range picks a
 * new one and the accumulators follow by the exact ratio).  Same call sites and arguments as the x3 entry points, with
 *   w_h2: [groups][K / 32][2 planes][cout_pad][32] f16 of the column-scaled weights (k order and chunk swizzle of w_x3),
 *         then [groups][cout_pad] f32 inverse column scales.  fgn_h2_image_bytes = the size of both.
 * Shapes: as the x3 entry points, and Cout 48..64 on a 64-column tile (fgn_h2_row_tile decides).  fgn_gemm_h2_f32: the direct entry (tests, tools);
 * bm 0 (= fgn_h2_row_tile) / 64 / 128 / 264 force the tile. */
size_t fgn_h2_image_bytes(int K, int npad, int n_groups);
int fgn_h2_row_tile(long long M, int Cout, int K, int grp_rows, int grp_valid);   /* 64 / 128: rows of a 128-column tile; 264: 128 rows x 64 columns (Cout 48..64); 0: use the f32 entry point */
/* A 3x3 or 1x1 convolution of any stride / padding with folded scale / shift / ReLU on ONE or TWO NHWC tensors that share
 * the weights (x1 == NULL: one; two: the query map and the support maps of a backbone layer, fgn_conv2d_pair_nhwc_f32's
 * call sites) as an implicit GEMM on conv_pw_h2_kernel.  w_h2 = the f16-plane image of the packed weights
 * [cout_pad][KH KW Cin] (K order tap, channel: fgn_amd/ops.py::pack_conv, pack_h2); Cin / 32 a power of two; the two
 * inputs within 2 GiB of each other. */
int fgn_conv2d_pair_h2_nhwc_f32(const float* x0, int n_img0, int H0, int W0, const float* x1, int n_img1, int H1, int W1,
                                const void* w_h2, float* y0, float* y1, const float* scale, const float* shift, int Cin,
                                int Cout, int cout_pad, int KH, int KW, int stride, int pad, int relu, void* stream);
int fgn_gemm_h2_f32(const float* x, const void* w_h2, float* y, const float* shift, const float* residual, int rows, int K,
                    int Cout, int npad, int relu, int grp_rows, int grp_valid, int n_groups, int bm, void* stream);
int fgn_conv1x1_h2_nhwc_f32(const float* x, const void* w_h2, float* y, const float* scale, const float* shift,
                            const float* residual, const int32_t* n_img_dev, int n_img, int H, int W, int Cin, int Cout,
                            int cout_pad, int relu, void* stream);
int fgn_conv1x1_dual_h2_nhwc_f32(const float* x, const float* x2, const int32_t* x2_rows, int x2_total_rows,
                                 const void* w_h2, float* y, const float* shift, int rows, int Cin1, int Cin2, int Cout,
                                 int cout_pad, int relu, void* stream);
int fgn_winograd_gemm_h2_f32(const float* V, const void* U_h2, float* Mo, const int32_t* n_img_dev, int n_img,
                             int tiles_per_img, int t_pad, int Cin, int Cout, int cout_pad, int n_groups, void* stream);

/* The same convolution (w_packed, scale, shift, relu as in fgn_conv2d_nhwc_f32) on TWO NHWC tensors of different
 * geometry in one launch: x0 [n_img0,H0,W0,Cin] -> y0, x1 [n_img1,H1,W1,Cin] -> y1.  For the backbone layers that
 * stride over the spatial structure when the query image and the support crops go through the backbone together
 * (fgn.py:212,215: the 3x3 / stride 2 conv2 and the 1x1 / stride 2 shortcut of a stage's first block, the stem) -
 * the other layers of the two passes already share launches by sharing rows.  Per tensor the arithmetic of
 * fgn_conv2d_nhwc_f32 without split-K; no residual / in_scale / device-side count; Cout % 4 == 0. */
int fgn_conv2d_pair_nhwc_f32(const float* x0, float* y0, int n_img0, int H0, int W0, const float* x1, float* y1,
                             int n_img1, int H1, int W1, const float* w_packed, const float* scale, const float* shift,
                             int Cin, int Cout, int cout_pad, int KH, int KW, int stride, int pad, int relu,
                             void* stream);

/* Winograd F(2x2,3x3) form of a 3x3 / stride 1 / pad 1 convolution (same call sites as
 * fgn_conv2d_nhwc_f32: fgn_ag_rpn_head.py:48 rpn_conv, fgn_roi_head.py:236 shared_head conv2):
 *   fgn_winograd_input_f32   V[16][t_pad][C]   = B^T (x * in_scale?) B per 4x4 input tile
 *   fgn_winograd_gemm_f32    Mo[g] = V[g] U[g]^T for the 16 tile positions, one MFMA launch (64x64 kernel,
 *                            per-group weights); U [16][cout_pad][Cin] = G w G^T (host-packed, BN scale folded)
 *   fgn_winograd_output_f32  y = A^T Mo A + shift (ReLU), y [n_img,H,W,Cout]
 * tiles per image = ceil(H/2)*ceil(W/2); t_pad = fgn_winograd_t_pad(n_img*tiles) (a multiple of 128 rows: whole GEMM tiles per group).
 * Image i reads x[i / a_img_div]; in_scale [n_img][C] optional (AG-RPN guidance, fgn_ag_rpn_head.py:44; mask
 * guidance, fgn_roi_head.py:379). */
int fgn_winograd_input_f32(const float* x, const float* in_scale, float* V, const int32_t* n_img_dev, int n_img,
                           int a_img_div, int H, int W, int C, int t_pad, void* stream);
int fgn_winograd_t_pad(int tiles_total);
int fgn_winograd_gemm_f32(const float* V, const float* U, float* Mo, const int32_t* n_img_dev, int n_img,
                          int tiles_per_img, int t_pad, int Cin, int Cout, int cout_pad, int n_groups, void* stream);
int fgn_winograd_output_f32(const float* Mo, float* y, const float* shift, const int32_t* n_img_dev, int n_img,
                            int H, int W, int C, int t_pad, int relu, void* stream);
/* F(4x4,3x3) form of the same convolutions (the default): 36 tile positions (n_groups = 36 in
 * fgn_winograd_gemm_f32), V / Mo [36][t_pad][C], tiles per image = ceil(H/4)*ceil(W/4); interpolation points
 * {0, 1, -1, 1/2, -2, inf} (winograd.hip).  4x fewer multiply-adds than the direct form, 1.78x fewer than F(2x2). */
int fgn_winograd4_input_f32(const float* x, const float* in_scale, float* V, const int32_t* n_img_dev, int n_img,
                            int a_img_div, int H, int W, int C, int t_pad, void* stream);
int fgn_winograd4_output_f32(const float* Mo, float* y, const float* shift, const int32_t* n_img_dev, int n_img,
                             int H, int W, int C, int t_pad, int relu, void* stream);
/* The F(4x4) transforms of TWO tensors of one layer (the query map and the support maps of a backbone layer, each with
 * its own image count and size) in one launch: tensor 0 fills tiles [0, n0*tiles(H0,W0)) of V / reads them from Mo,
 * tensor 1 the tiles behind them; no input scale, no device-side count.  t_pad >= all tiles. */
int fgn_winograd4_input2_f32(const float* x0, int n_img0, int H0, int W0, const float* x1, int n_img1, int H1, int W1,
                             float* V, int C, int t_pad, void* stream);
int fgn_winograd4_output2_f32(const float* Mo, const float* shift, float* y0, int n_img0, int H0, int W0, float* y1,
                              int n_img1, int H1, int W1, int C, int t_pad, int relu, void* stream);
/* Which instance of the F(4x4) transform kernels a layer with `tiles_total` tiles and C channels launches: the
 * channel vector width of a thread (1, 2 or 4 floats) * 10 + 1 when the input transform issues all 36 loads up front.
 * Informational (lets a profiler name the kernel of a launch); the transforms choose it themselves. */
int fgn_winograd4_variant(int tiles_total, int C, int is_output);
/* Weight side of both Winograd forms on the device: U[(a*R+b)][co][ci] = sum_ij G[a][i] w[co][ci][i][j] G[b][j] (R = m + 2,
 * fp64 arithmetic, one rounding), written IN PLACE into U [R*R][cout_pad][Cin] (rows past cout untouched).  w is the
 * torch-layout weight [Cout][Cin][3][3], G the [R][3] fp64 table of the interpolation points.  What mmcv's optimizer
 * step + the next forward need between two training steps (no reference counterpart: cuDNN picks its own algorithm). */
int fgn_winograd_pack_weights_f32(const float* w, const double* G, float* U, int cout, int cin, int cout_pad, int m,
                                  void* stream);

/* NCHW [n,3,H,W] -> NHWC4 [n,H,W,4] (input side of fgn.py:212,215) */
int fgn_nchw3_to_nhwc4_f32(const float* x, float* y, int n_img, int H, int W, void* stream);

/* 3x3/2 pad 1 max-pool of the ResNet stem */
int fgn_maxpool3x3s2_nhwc_f32(const float* x, float* y, int n_img, int H, int W, int C, void* stream);

/* GroupNorm over NHWC (+ residual) (+ ReLU): the norm layers of the from-scratch backbone variant
 * (fgn_r50_c4_scratch.py:16-23, norm_cfg GN(32); torch.nn.GroupNorm inside mmdet ResNet at
 * fgn.py:71-73).  y = (x - mean_g) * rstd_g * gamma[c] + beta[c] per (image, group);
 * ws from fgn_group_norm_workspace_bytes.  x == y (in place) is allowed. */
size_t fgn_group_norm_workspace_bytes(int n_img, int HW, int C, int groups);
int fgn_group_norm_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta,
                            const float* residual, void* ws, size_t ws_bytes, int n_img, int HW, int C,
                            int groups, float eps, int relu, void* stream);

/* AvgPool2d(2, stride 2, ceil_mode, count_include_pad=False): shortcut of a strided bottleneck
 * under avg_down (fgn_r50_c4_scratch.py:18-19); out [n, ceil(H/2), ceil(W/2), C] */
int fgn_avgpool2x2_nhwc_f32(const float* x, float* y, int n_img, int H, int W, int C, void* stream);

/* RoIAlign (avg), rois [R,5] = (batch_idx,x1,y1,x2,y2), out [R,P,P,C].
 * aligned=1,sampling_ratio=0 : mmcv.ops.RoIAlign (fgn_roi_head.py:331,366)
 * aligned=0,sampling_ratio=-1: torchvision.ops.roi_align (fgn_roi_head.py:429,432)
 * post_shift [C] (optional) is added and ReLU applied after the average: RoIAlign is linear per channel, so
 * the first 1x1 conv of the shared_head (fgn_roi_head.py:236) is applied to the feature map once and its
 * BN shift + ReLU here, instead of a conv over every RoI's 49 pixels. */
int fgn_roi_align_nhwc_f32(const float* fmap, const float* rois, float* out, const int32_t* n_rois_dev,
                           int n_rois, int n_img, int H, int W, int C, int out_size, float spatial_scale,
                           int sampling_ratio, int aligned, const float* post_shift, int relu, void* stream);
/* The same pooling of TWO maps of one spatial size at the same RoIs in one launch: out [R,P,P,C] from fmap, out2
 * [R,P,P,C2] from fmap2 (+ post_shift2 [C2], ReLU) - the C4 map and the map of the shared head's first 1x1 conv taken
 * in front of the pooling (fgn_roi_head.py:331-336, 366-369).  Identical bytes to two single-map calls. */
int fgn_roi_align2_nhwc_f32(const float* fmap, const float* fmap2, const float* rois, float* out, float* out2,
                            const int32_t* n_rois_dev, int n_rois, int n_img, int H, int W, int C, int C2,
                            int out_size, float spatial_scale, int sampling_ratio, int aligned,
                            const float* post_shift2, int relu2, void* stream);
/* same for a single-channel uint8/bool map [n_img,H,W] -> [R,P,P] (support masks,
 * fgn_roi_head.py:429) */
int fgn_roi_align_mask_u8(const uint8_t* mask, const float* rois, float* out, int n_rois, int n_img, int H,
                          int W, int out_size, float spatial_scale, int sampling_ratio, int aligned,
                          void* stream);

/* out[g][c] = mean_{k<K,p<P} x[g*K+k][p][c] * (weights ? weights[g*K+k][p] : 1)
 * (fgn_ag_rpn_head.py:38-41 with weights=NULL; fgn_roi_head.py:444-447 with the pooled masks) */
int fgn_support_class_vectors_f32(const float* x, const float* weights, float* out, int n_groups, int K,
                                  int P, int C, void* stream);
/* out[g][p][c] = mean_k x[g*K+k][p][c]   (fgn_roi_head.py:439-442) */
int fgn_support_kmean_f32(const float* x, float* out, int n_groups, int K, int P, int C, void* stream);
/* out[i][:] = table[labels[i] + n_ways*img(i)][:]  (fgn_roi_head.py:707-714); rois may be NULL (img 0) */
int fgn_gather_support_vectors_f32(const float* table, const int64_t* labels, const float* rois, float* out,
                                   const int32_t* n_dev, int n, int n_ways, int C, void* stream);

/* out[n][p][c] = x[n / div][p][c] * v[n][c]  (guidance multiply, fgn_ag_rpn_head.py:44-46, materialised
 * for the LDS-DMA conv kernel where a layer does not take the Winograd form); x [n_out/div, P, C], v [n_out, C], out [n_out, P, C] */
int fgn_scale_channels_f32(const float* x, const float* v, float* out, int n_out, int div, int P, int C,
                           void* stream);

/* Fused tail of count_one_roi_by_n_spp + BBoxHead.forward (fgn_roi_head.py:253-279,338):
 * x = Q[r] + S[img*N+n]; GroupNorm(32)+ReLU; 7x7 avg-pool; fc_cls/fc_reg.
 * Q [R,49,C], S [B*N,49,C] (bias included), fc_weight [6,C] (2 cls rows then 4 reg rows) */
int fgn_relation_gn_head_f32(const float* Q, const float* S, const float* rois, const float* gn_weight,
                             const float* gn_bias, const float* fc_weight, const float* fc_bias, float* cls_out,
                             float* reg_out, const int32_t* n_rois_dev, int n_rois, int n_ways, int C,
                             int gn_groups, int roi_size, float eps, float* rel_out_debug, float* scratch, void* stream);
/* scratch of fgn_relation_gn_head_f32: per (RoI, 256-channel chunk, class) partial fc products, summed in chunk order */
size_t fgn_relation_gn_head_scratch_bytes(int n_rois, int n_ways, int C);

/* AG-RPN merge: per-anchor arg-max over the N guided passes (fgn_ag_rpn_head.py:81-113) + sigmoid.
 * head [B*N,HW,head_channels]: channels [0,A) objectness, [A,5A) deltas. Outputs in (y,x,a) order. */
int fgn_rpn_merge_f32(const float* head, float* logits, float* scores, float* deltas, int batch, int n_ways,
                      int HW, int n_anchors, int head_channels, void* stream);

/* Proposals (mmdet RPNHead._get_bboxes_single/_bbox_post_process via fgn.py:229-235):
 * top nms_pre -> delta2bbox -> min size -> NMS -> max_per_img.  proposals [B,max_per_img,5],
 * n_props [B]. scratch: fgn_rpn_proposals_scratch_bytes(). dbg_topk_idx optional [B,cap]. */
size_t fgn_rpn_proposals_scratch_bytes(int batch, int n_total, int nms_pre);
/* rois_out (optional) [B*max_per_img,5]: the same boxes as (image index, x1, y1, x2, y2) = bbox2roi (fgn_roi_head.py:556) */
/* pre_zeroed (optional): fgn_rpn_proposals_zeroed_bytes(batch) bytes of ZERO-FILLED device memory (fresh per call);
 * when given, the top-k selection of the ~63 000 scores runs as multi-workgroup kernels in front of the proposal
 * kernel (same result, bit for bit) */
size_t fgn_rpn_proposals_zeroed_bytes(int batch);
int fgn_rpn_proposals_f32(const float* scores, const float* deltas, const float* base_anchors, void* scratch,
                          void* pre_zeroed, float* proposals, float* rois_out, int32_t* n_props, int32_t* dbg_topk_idx, int batch, int feat_h,
                          int feat_w, int n_anchors, int stride, float img_h, float img_w,
                          const float* host_means4, const float* host_stds4, float max_ratio, int nms_pre,
                          float min_bbox_size, float iou_thr, int max_per_img, void* stream);

/* count_modified_cls_bbox + BBoxHead.get_bboxes + multiclass_nms, one workgroup per image
 * (fgn_roi_head.py:302-326, 606-613).  `batch` images with n_rois RoIs each, stacked: rois [batch*n_rois,5],
 * cls_raw [batch*n_rois*n_ways,2], reg_raw [batch*n_rois*n_ways,4], n_rois_dev (optional) [batch];
 * det_bboxes [batch*max_per_img,5], det_labels int64 [batch*max_per_img], n_dets [batch]; image i carries
 * image index img_index + i in mask_rois_out.  scratch: batch * fgn_det_post_scratch_bytes() bytes. */
size_t fgn_det_post_scratch_bytes(int max_rois, int n_ways);   /* per image */
int fgn_det_post_f32(const float* rois, const float* cls_raw, const float* reg_raw, const int32_t* n_rois_dev,
                     void* scratch, float* det_bboxes, float* mask_rois_out /* optional [max_per_img,5]: (img_index, box),
                     the mask branch's bbox2roi, fgn_roi_head.py:654 */, int img_index, int64_t* det_labels, int32_t* n_dets,
                     float* dbg_scores /* optional, tests / diagnostics: n_rois*(n_ways+1) softmax scores of the first image
                     + 16 words of phase stamps */,
                     int batch, int n_rois, int n_ways, float img_h, float img_w, const float* host_means4,
                     const float* host_stds4, float max_ratio, float score_thr, float iou_thr, int max_per_img,
                     void* stream);

/* conv_logits (1x1, one class) + sigmoid on the un-shuffled deconv output [D,P*P,4,C];
 * logits/prob [D,2P,2P] (FCNMaskHead, fgn_roi_head.py:380; sigmoid of get_seg_masks).  bias_dev (optional): the bias
 * as one float in device memory, read by the kernel instead of `bias` (a training loop updates it on the device). */
int fgn_mask_logits_f32(const float* x, const float* w, float bias, const float* bias_dev, float* logits, float* prob,
                        const int32_t* n_dev, int n_det, int roi_size, int C, void* stream);

/* _do_paste_mask + threshold (fgn_roi_head.py:668-671): out uint8 [D,H,W].
 * skip_empty = 1: mmdet's CPU path (paste inside the integer-expanded box only - the CPU reference of north_star);
 * skip_empty = 0: its CUDA path (the grid spans the whole image; the reference runs on cuda:0, main.py:365).  The two
 * agree for thr >= 0.5 (the configured 0.5, fgn_r50_c4_densecl.py:186) on boxes of positive width and height, and
 * differ below it (and on degenerate boxes, whose grid coordinate mmdet sets to 0: the mask's centre line everywhere). */
int fgn_mask_paste_u8(const float* prob, const float* boxes, int box_stride, uint8_t* out, const int32_t* n_dev,
                      int n_det, int img_h, int img_w, int mask_size, float thr, int skip_empty, void* stream);

/* Fused paste + threshold + COCO RLE (replaces get_seg_masks -> .cpu() -> pycocotools encode,
 * fgn_roi_head.py:668-671 + fgn.py:267,281): out_bytes [D,byte_cap] holds the COCO "counts"
 * string of detection d in its first out_len[d] bytes.  trans_scratch uint32 [D,trans_cap].
 * overflow[d] != 0: a cap was too small, use fgn_mask_paste_u8 + host RLE for that detection. */
int fgn_mask_rle(const float* prob, const float* boxes, int box_stride, uint32_t* trans_scratch,
                 uint8_t* out_bytes, int32_t* out_len, int32_t* overflow, const int32_t* n_dev, int n_det,
                 int img_h, int img_w, int mask_size, float thr, int trans_cap, int byte_cap, int skip_empty,
                 void* stream);

/* COCO RLE of dense binary masks on the device: the query's ground-truth masks, which the reference copies to
 * the GPU with the batch (fgn.py:92-99) and encodes on the host with pycocotools (fgn.py:298, `qry_isegmaps_rle`).
 * masks [n][H][W] bytes (non-zero = set); out_bytes [n][byte_cap], out_len [n], overflow [n] (a cap was exceeded:
 * the caller encodes that mask on the host).  scratch: fgn_dense_rle_scratch_bytes(). */
size_t fgn_dense_rle_scratch_bytes(int n_masks, int img_h, int img_w, int trans_cap);
int fgn_dense_mask_rle(const uint8_t* masks, void* scratch, size_t scratch_bytes, uint8_t* out_bytes,
                       int32_t* out_len, int32_t* overflow, int n_masks, int img_h, int img_w, int trans_cap,
                       int byte_cap, void* stream);

/* ---- forward_train (fgn.py:125-185) ----------------------------------------------------- */

/* The proposal stage at training sizes (train_cfg.rpn_proposal, fgn_r50_c4_densecl.py:153-157: nms_pre 12000,
 * max_per_img 2000 - any nms_pre, max_per_img <= 4096): global bitonic sort of all anchor keys, then one workgroup
 * per image decodes the best nms_pre and runs the greedy NMS.  Same outputs as fgn_rpn_proposals_f32. */
size_t fgn_rpn_proposals_large_scratch_bytes(int batch, int n_total, int nms_pre);
int fgn_rpn_proposals_large_f32(const float* scores, const float* deltas, const float* base_anchors, void* scratch,
                                float* proposals, float* rois_out, int32_t* n_props, int batch, int feat_h, int feat_w,
                                int n_anchors, int stride, float img_h, float img_w, const float* host_means4,
                                const float* host_stds4, float max_ratio, int nms_pre, float min_bbox_size,
                                float iou_thr, int max_per_img, void* stream);

/* MaxIoUAssigner.assign (my_max_iou_assigner.py:60-213 on mmdet's bbox_overlaps): boxes [n, box_stride >= 4],
 * inside (optional) [n] bytes: 0 = not a candidate (anchor outside the image, my_anchor_head.py:233-239),
 * gts [k,4], k <= 256.  gt_inds [n] int32: -2 not a candidate, -1 ignored, 0 negative, i+1 = positive of GT i;
 * max_overlaps (optional) [n].  scratch: fgn_box_assign_scratch_bytes(n, k), uninitialised. */
size_t fgn_box_assign_scratch_bytes(int n, int k);
int fgn_box_assign_f32(const float* boxes, int box_stride, const uint8_t* inside, const float* gts, int n, int k,
                       float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality, void* scratch,
                       int32_t* gt_inds, float* max_overlaps, void* stream);

/* DeltaXYWHBBoxCoder.encode (bbox_coder.encode at my_anchor_head.py:256, fgn_roi_head.py:141): [n,4] x [n,4] -> [n,4] */
int fgn_bbox2delta_f32(const float* proposals, const float* gts, float* out, int n, const float* host_means4,
                       const float* host_stds4, void* stream);

/* Weighted loss sums, out[0] = sum_i w_i * loss_i / avg_factor (mmdet weight_reduce_loss; w optional):
 * sigmoid CE (F.binary_cross_entropy_with_logits; y >= y_threshold is the target when y_threshold >= 0 - the mask
 * targets of mask_target_single), smooth L1, softmax CE (labels int64, outside [0, n_classes) ignored). */
int fgn_bce_logits_sum_f32(const float* x, const float* y, const float* w, long long n, float y_threshold,
                           double avg_factor, float* out, void* stream);
int fgn_smooth_l1_sum_f32(const float* pred, const float* target, const float* w, long long n, float beta,
                          double avg_factor, float* out, void* stream);
int fgn_softmax_ce_sum_f32(const float* logits, const int64_t* labels, const float* w, int n, int n_classes,
                           double avg_factor, float* out, void* stream);

/* BatchNorm2d in training mode (the shared head's BN layers during forward_train, fgn_roi_head.py:202-238) on
 * NHWC rows x [P,C]: batch mean / biased variance -> mean, var; y = (x-mean)/sqrt(var+eps)*gamma+beta (+residual)
 * (ReLU) -> out (may alias x); running_mean / running_var (optional) updated with `momentum` (unbiased variance). */
size_t fgn_bn_train_scratch_bytes(int C);
int fgn_bn_train_f32(const float* x, int P, int C, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, const float* residual, int relu, void* scratch,
                     float* mean, float* var, float* out, void* stream);

/* ---- backward of the trainable heads (the reference: torch.autograd under fgn.py:125-185) ---------------- */

/* d(sum_i w_i loss_i / avg_factor)/d(prediction) * scale for the three loss forms above (scale carries 1/avg_factor
 * and any upstream factor such as the 1/N balancer of fgn_ag_rpn_head.py:77-78) */
int fgn_bce_logits_grad_f32(const float* x, const float* y, const float* w, long long n, float y_threshold, float scale,
                            float* dx, void* stream);
int fgn_smooth_l1_grad_f32(const float* pred, const float* target, const float* w, long long n, float beta, float scale,
                           float* dpred, void* stream);
int fgn_softmax_ce_grad_f32(const float* logits, const int64_t* labels, const float* w, int n, int n_classes,
                            float scale, float* dlogits, void* stream);

/* ReLU backward: out = dy * [y > 0], y = the ReLU's output; n % 4 == 0 (mask convs, rpn_conv) */
int fgn_relu_backward_f32(const float* dy, const float* y, float* out, long long n, void* stream);

/* out[c] (+)= sum_r x[r][c]: bias gradients and reductions over RoIs; fp64 partials in a fixed order */
size_t fgn_colsum_scratch_bytes(int C);
int fgn_colsum_f32(const float* x, long long R, int C, void* scratch, float* out, int accumulate, void* stream);

/* BatchNorm2d(train) backward on NHWC rows [P,C]: g = dy * [y_post > 0] (y_post optional: the ReLU after the norm);
 * dx, dgamma, dbeta; g_out (optional) = g, the gradient of the identity branch of relu(bn3(conv3) + identity) */
size_t fgn_bn_train_backward_scratch_bytes(int C);
int fgn_bn_train_backward_f32(const float* x_pre, const float* y_post, const float* dy, const float* mean,
                              const float* var, const float* gamma, float eps, int P, int C, void* scratch, float* dx,
                              float* g_out, float* dgamma, float* dbeta, void* stream);

/* Backward of fgn_relation_gn_head_f32 (fc_cls|fc_reg -> avg-pool -> ReLU -> GroupNorm -> Q + S): d_out6 [R*N,6] =
 * (d cls_raw | d reg_raw); dQ [R,49,C]; dZ [R*N,49,C] (summed over the RoIs of an image it is dS); pooled [R*N,C]
 * (forward value, for the fc weight gradient); dgamma_part / dbeta_part [R,C] (column-summed by the caller) */
int fgn_relation_gn_head_backward_f32(const float* Q, const float* S, const float* rois, const float* gn_weight,
                                      const float* gn_bias, const float* fc_weight, const float* d_out6, float* dQ,
                                      float* dZ, float* pooled, float* dgamma_part, float* dbeta_part, int n_rois,
                                      int n_ways, int C, int gn_groups, int roi_size, float eps, void* stream);

/* Backward of fgn_mask_logits_f32 (ReLU(deconv) -> conv_logits): up [D,P*P,4,C], dlogit [D,2P,2P] ->
 * d_up (through the ReLU) and dw_part [D,C] (column-summed by the caller) */
int fgn_mask_logits_backward_f32(const float* up, const float* dlogit, const float* w, float* d_up, float* dw_part,
                                 int n_det, int roi_size, int C, void* stream);

/* im2col of a 3x3 / stride 1 / pad 1 input, NHWC: out [n*H*W, 9*C], column (ky*3+kx)*C + ci (weight gradients) */
int fgn_im2col3x3_f32(const float* x, float* out, int n, int H, int W, int C, void* stream);

/* Weight-gradient GEMM C [M,N] = A [R,M]^T . B [R,N] (A = dY, B = X or im2col(X); reduction over the rows) on fp32
 * MFMA; M % 4 == 0, N % 4 == 0; slabs of rows reduced in a fixed order.  workspace: fgn_gemm_tn_workspace_bytes() */
size_t fgn_gemm_tn_workspace_bytes(int R, int M, int N);
int fgn_gemm_tn_f32(const float* A, const float* B, float* C, int R, int M, int N, void* workspace, void* stream);

/* Products too narrow for the MFMA kernels (a dimension that is not a multiple of 4 / 32): C[M,N] = A[M,K] B[K,N]
 * (trans_a = 0) or A[K,M]^T B[K,N] (trans_a = 1), one thread per output element, reduction in index order.  Training
 * only: the 6-row fc weight / data gradients and the 75-channel AG-RPN head (torch.matmul in round 2). */
int fgn_gemm_small_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int trans_a, void* stream);

/* torch.optim.Adagrad step (fgn_train_schedule.py:5-13): g += wd*p; state += g*g; p -= lr*g/(sqrt(state)+eps) */
/* the same update for `count` parameter tensors (each with its own learning rate) in one launch; element-wise, so
 * bit-identical to `count` calls of fgn_adagrad_step_f32 (Trainer.step: ~40 tensors of the heads) */
int fgn_adagrad_multi_f32(float* const* params, const float* const* grads, float* const* state_sums, const long long* n,
                          const float* lr, int count, float weight_decay, float eps, void* stream);
int fgn_adagrad_step_f32(float* param, const float* grad, float* state_sum, long long n, float lr, float weight_decay,
                         float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
