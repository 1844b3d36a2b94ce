/* C ABI of libu2seg_hip.so — the MI355X (gfx950) kernels behind the U2Seg Panoptic-FPN hot path.
 *
 * Every entry point takes raw device pointers, sizes and a hipStream_t passed as void*; nothing is
 * allocated inside (workspaces are caller-provided); the return value is 0 on success, a positive
 * hipError_t on a launch failure, a negative number on an argument the kernel cannot serve.
 * The caller is a torch.autograd.Function in u2seg_amd/layers (the reference binds its native ops
 * the same way: detectron2/layers/roi_align_rotated.py:9-46 + detectron2/layers/csrc/vision.cpp:111-116).
 *
 * Activations are NHWC bfloat16 (raw uint16 bits), statistics / losses / box arithmetic are fp32.
 * Each prototype cites the reference call site(s) it replaces (paths relative to the reference root).
 */
#ifndef U2SEG_HIP_H
#define U2SEG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- convolution / linear family (conv_igemm.hip) -------------------------------------------
 * Replaces F.conv2d in detectron2/layers/wrappers.py:127-134 (all Conv2d of backbone/resnet.py:194-210,
 * 355-358, backbone/fpn.py:141-158, meta_arch/semantic_seg.py:196-205, proposal_generator/rpn.py:127-134,
 * roi_heads/mask_head.py:242-251), nn.Linear in roi_heads/box_head.py:66-74 and roi_heads/fast_rcnn.py:236-239,
 * and (through mul/div + a flipped, transposed filter) their data gradients.
 *   out[m][n] = sum_{kh,kw,c} in[src(m,kh,kw)][c] * wt[n][kh][kw][c]   (+bias)(+=old)(relu)
 *   src: sy = oy*mul - pad_h + kh; if div > 1 the tap contributes only when sy % div == 0 and then sy /= div.
 *   stats (optional, [2][N] fp32, pre-zeroed): column sum and sum of squares of the stored bf16 outputs.
 * variant bit0: 1 = register-staged operands instead of global_load_lds; bit2 BK 32; bit3 256x128 tile; bits 4-5 LDS ring depth;
 * bit8 / bit9 force / forbid conv_igemm256; bits 12-15: persistent tile configuration of conv_tile.hip (0 automatic,
 * 1-7 see there, 15 never); bit16: 8 work-groups only (tests). */
int u2_conv_igemm(const void* in, const void* wt, void* out, const float* bias, float* stats,
                  int B, int Hin, int Win, int C, int in_ld, int Hout, int Wout, int N, int out_ld,
                  int KH, int KW, int pad_h, int pad_w, int mul, int div, int relu, int accumulate,
                  int variant, void* stream);

/* Weight gradient: dw[n][kh][kw][c] += sum_m dy[m][n] * x[src(m,kh,kw)][c]; fp32 atomics, dw pre-zeroed.
 * variant bit0: register staging; bit1: scalar LDS gathers instead of ds_read_b64_tr_b16; bits 4-5: fewer pixel splits;
 * bit6 / bit7: force / forbid the XCD-grouped launch (default: on for multi-tap filters over >= 200k pixels); bit8 / bit11:
 * force / forbid the 256 x 256 tile kernel (default: 1x1 layers over >= 200k pixels and the 7x7 fc1); bits 9-10: LDS ring depth. */
int u2_conv_wgrad(const void* x, const void* dy, float* dw, int B, int Hin, int Win, int C, int x_ld,
                  int Hout, int Wout, int N, int dy_ld, int KH, int KW, int pad_h, int pad_w,
                  int stride, int variant, void* stream);
/* Same reduction accumulated into a caller-laid-out destination: dw[n * dw_stride_n + tap * dw_stride_tap + c * dw_stride_c]
 * for n < n_valid, c < c_valid (tap = kh * KW + kw).  With (Cin*KH*KW, 1, KH*KW) the destination is the reference's
 * own [N][Cin][KH][KW] parameter gradient (torch.nn.Conv2d.weight.grad, layers/wrappers.py:127-134), so the gradient
 * lands in the optimizer's flat arena without a temporary, a permute or an accumulate pass. */
int u2_conv_wgrad_into(const void* x, const void* dy, float* dw, int B, int Hin, int Win, int C, int x_ld,
                       int Hout, int Wout, int N, int dy_ld, int KH, int KW, int pad_h, int pad_w,
                       int stride, int n_valid, int c_valid, long long dw_stride_n, int dw_stride_tap,
                       int dw_stride_c, int variant, void* stream);

/* Both gradients of a 1x1 / stride-1 convolution in ONE pass over dy (autograd's conv backward behind layers/wrappers.py:127-134
 * for the expanding 1x1 layers of the bottlenecks, backbone/resnet.py:194-203): dx[m][c] = sum_n dy[m][n] wt[c][n] (bf16, all dx_ld
 * physical channels written) and dw[n * dw_stride_n + c * dw_stride_c] += sum_m dy[m][n] x[m][c] (fp32, n < n_valid, c < c_valid).
 * wt is the data-gradient layout of the filter, [C][wt_ld] with n contiguous (u2_weight_layout mode 1).  Returns 0 when launched,
 * 1 when the shape is not served (blocks of N <= 256 by C <= 64 or N <= 512 by C <= 128 over >= 200 000 pixels; variant bit 0
 * lifts the size rule, bit 1: 8 pixel ranges, bit 2: never) - the caller then issues u2_conv_igemm + u2_conv_wgrad_into. */
int u2_conv1x1_bwd_fused(const void* x, const void* dy, const void* wt, void* dx, float* dw, int M, int C, int x_ld, int N, int dy_ld,
                         int wt_ld, int dx_ld, int n_valid, int c_valid, long long dw_stride_n, int dw_stride_c, int variant,
                         void* stream);

/* The same pass for a layer whose output y feeds a batch normalisation (layers/batch_norm.py:169-197 behind layers/wrappers.py:
 * 127-134: conv3 + norm of a bottleneck, the projection shortcut): the normalisation's backward apply step
 * dy[m][n] = k1[n] dz[m][n] + k2[n] y[m][n] + k3[n] (u2_norm_bwd_apply with relu = 0, coefficients of u2_bn_finalize_bwd) is
 * evaluated on the staged rows instead of being written and read back; dx and dw are those of u2_conv1x1_bwd_fused on that dy.
 * Served: N <= 256 by C <= 64 (same size rule and variant bits); returns 1 otherwise - the caller then runs u2_norm_bwd_apply
 * followed by u2_conv1x1_bwd_fused (or the two separate launches). */
int u2_conv1x1_bwd_fused_bn(const void* x, const void* dz, const void* y, const float* k1, const float* k2, const float* k3,
                            const void* wt, void* dx, float* dw, int M, int C, int x_ld, int N, int dy_ld, int wt_ld, int dx_ld,
                            int n_valid, int c_valid, long long dw_stride_n, int dw_stride_c, int variant, void* stream);

/* Test / debugging aid: a code for the kernel (family, tile configuration) the most recent u2_conv_igemm or u2_conv_wgrad
 * call on this thread selected (encoding in csrc/conv_args.h).  With variant 0 the selection can be steered through the
 * environment variables U2_CONV_VARIANT / U2_WGRAD_VARIANT (same bits as the variant argument). */
int u2_conv_last_kernel(void);

/* The only device memory the library owns: the conv kernels' scratch (stream-K hand-over slots of u2_conv_igemm, partial tiles of
 * u2_conv_wgrad), one block per (device, stream), allocated on first need at the size of that launch and grown on demand.  This
 * call drains the streams concerned and frees every block (the next launch that needs one allocates again); returns the number
 * of blocks freed.  The reference's ops (ATen conv behind layers/wrappers.py:127-134) take such workspaces from torch's caching
 * allocator, where torch.cuda.empty_cache() releases them - this is the equivalent for a plain C ABI. */
int u2_release_scratch(void);

/* fp32 master weights [N][Cin][T] -> bf16 kernel layouts: mode 0 [N][T][Cp] (forward / wgrad), mode 1 [Cp][T][Npad] with the
 * taps reversed (data gradient), mode 2 [T][Cp][Npad] (data gradient of a fully connected conv). Zero padded. */
int u2_weight_layout(const float* w, void* out, int N, int Cin, int T, int Cp, int Npad, int mode, void* stream);
/* The same for a whole table of (parameter, layout) pairs in one launch - run once after the optimizer step on the flat
 * parameter arena (base + src_offset floats), so that no per-layer layout launch is left in the forward/backward pass. */
typedef struct U2LayoutDesc {
  long long src_offset; /* floats from `base` to the [N][Cin][T] fp32 parameter */
  void* dst;            /* device pointer of the bf16 layout */
  int N, Cin, T, Cp, Npad, mode;
  int block_begin;      /* first work-group of this entry: prefix sum of blocks(entry) = mode 0: N * ceil(Cp/64);
                           modes 1, 2 (need Cp == Cin): ceil(Npad/64) * ceil(Cin*T/64); mode 3 (this table only): dst is an
                           fp32 vector of N elements = the source rounded through bf16 (a conv bias under autocast),
                           ceil(N/4096) blocks; mode 0 with T == 1 and Cp == Cin: ceil(N*Cp/4096) */
  int reserved;
} U2LayoutDesc;
int u2_weight_layout_batched(const float* base, const U2LayoutDesc* table, int n_entries, int total_blocks, void* stream);

/* ---- normalisation / activation (norm.hip) ----------------------------------------------------
 * Replaces nn.SyncBatchNorm / nn.GroupNorm / relu_ chosen by detectron2/layers/batch_norm.py:169-197. */
int u2_colstats(const void* x, float* out /*[slots][2][C]*/, int slots, int rows_per_slot, int C, int ld, void* stream);
/* dst[c] += sum over rows of x[row][c] for c < n_valid (x: [rows][ld] bf16, C physical channels): the bias gradient of a conv /
 * linear layer (autograd's sum over the output gradient behind layers/wrappers.py:127-134) accumulated straight into the
 * parameter's fp32 gradient storage - no temporary, no AccumulateGrad add. */
int u2_colsum_add(const void* x, float* dst, int rows, int C, int ld, int n_valid, void* stream);
/* nn.GroupNorm finalize (layers/batch_norm.py:189 "GN", semantic_seg.py:196-205). fwd: stats [B][2][C] (u2_colstats per image)
 * -> mean / invstd / scale / shift [B][C], n = H*W*(C/groups) elements per group. bwd: sums [B][2][C] (u2_norm_bwd_reduce)
 * -> k1/k2/k3 [B][C] for u2_norm_bwd_apply and dgamma / dbeta [C] summed over the images in order; accumulate != 0: added to
 * what dgamma / dbeta hold (the optimizer's gradient slices: no temporary, no AccumulateGrad add). */
int u2_gn_finalize_fwd(const float* stats, const float* gamma, const float* beta, float n, float eps, int B, int C, int groups,
                       float* mean, float* invstd, float* scale, float* shift, void* stream);
int u2_gn_finalize_bwd(const float* sums, const float* gamma, const float* mean, const float* invstd, float n, int B, int C,
                       int groups, float* k1, float* k2, float* k3, float* dgamma, float* dbeta, int accumulate, void* stream);
/* count: elements per channel behind `sums`; count_dev (optional, device, 1 float) overrides it - SyncBN all-reduces the
 * per-rank counts together with the sums, because ranks pad their batches to different sizes (nn.SyncBatchNorm does). */
int u2_bn_finalize_fwd(const float* sums, float count, const float* count_dev, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                       float* shift, int C, void* stream);
/* relu_bits (optional, uint8 [slots * rows_per_slot][C / 8], needs ld == C and C / 8 dividing 256): bit e of byte c is
 * out[row][8 c + e] > 0 - what the backward pass of a residual block's tail needs of the activation, in 1/16 of its bytes. */
int u2_affine_act(const void* x, const float* scale, const float* shift, const void* resid, void* out, int slots,
                  int rows_per_slot, int C, int ld, int relu, void* relu_bits, void* stream);
/* backbone/fpn.py:141-158 in one pass: out = bf16(x * scale + shift) + nearest_x2(top), top [B][H/2][W/2][C]: the lateral conv's
 * BatchNorm apply fused with the top-down upsample-add (bit-identical to u2_affine_act followed by u2_fpn_upsample_add_fwd). */
int u2_affine_upadd(const void* x, const float* scale, const float* shift, const void* top, void* out, int B, int H, int W,
                    int C, int relu, void* stream);
int u2_norm_bwd_reduce(const void* dout, const void* mask, const void* x, const float* mean, const float* invstd,
                       float* out /*[slots][2][C]*/, int slots, int rows_per_slot, int C, int ld, int relu,
                       const float* mask_scale, const float* mask_shift, const void* dout2, void* dz_out, const void* dout3,
                       int mask_is_bits, void* stream);
/* mask_scale/mask_shift [slots][C] (optional, both or neither): recompute the ReLU mask as x*scale+shift > 0 - the
 * expression u2_affine_act evaluated in the forward pass - instead of reading the activation `mask` (which may be NULL).
 * u2_norm_bwd_reduce only: dz_out (optional) receives the masked gradient dz = (dout [+ dout2]) * mask, so that
 * u2_norm_bwd_apply can run on (dz, x) with relu = 0 and dz doubles as the residual branch's gradient; dout2 (optional,
 * needs dz_out) is a second incoming gradient summed on the fly (resnet.py:204-210: the block output feeds the next
 * block's conv1 and its identity shortcut); dout3 (optional, needs dout2) a third one (the last block of a stage also feeds
 * the FPN lateral conv, backbone/fpn.py:141-146); mask_is_bits: `mask` is the relu_bits array of u2_affine_act (needs dz_out). */
int u2_bn_finalize_bwd(const float* sums, float count, const float* count_dev, const float* gamma, const float* mean, const float* invstd,
                       const float* local_sums, float* dgamma, float* dbeta, float* k1, float* k2, float* k3, int C,
                       int accumulate /* dgamma/dbeta += instead of = (parameter gradient arena) */, void* stream);
int u2_norm_bwd_apply(const void* dout, const void* mask, const void* x, const float* k1, const float* k2,
                      const float* k3, void* dx, void* dres, int slots, int rows_per_slot, int C, int ld, int relu,
                      const float* mask_scale, const float* mask_shift, void* stream);
/* Round 4: nn.SyncBatchNorm's finalize and apply steps (layers/batch_norm.py:169-197 + the residual add / relu_ of
 * backbone/resnet.py:204-210) as ONE launch each way - u2_bn_finalize_fwd + u2_affine_act (slots = 1), resp. u2_bn_finalize_bwd +
 * u2_norm_bwd_apply, with identical results: every thread derives the coefficients of its channels from the column sums, one
 * work-group writes mean / invstd / scale / shift and the running statistics (forward) or adds dgamma / dbeta (backward).
 * k123: [3][C] scratch, only written when C is not served by the one-launch form (C / 8 must divide 256). */
int u2_bn_act_fused(const void* x, const float* sums, float count, const float* count_dev, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                    float* shift, const void* resid, void* out, int rows, int C, int ld, int relu, void* relu_bits, void* stream);
int u2_bn_bwd_apply_fused(const float* sums, float count, const float* count_dev, const float* gamma, const float* mean,
                          const float* invstd, const float* local_sums, float* dgamma, float* dbeta, float* k123, int accumulate,
                          const void* dout, const void* mask, const void* x, void* dx, void* dres, int rows, int C, int ld,
                          int relu, const float* mask_scale, const float* mask_shift, void* stream);
int u2_relu_bwd(const void* dout, const void* out, void* dz, long long numel, void* stream);
/* The same with the bias gradient of the layer in the same pass: dst[c] += sum over rows of dz[row][c] for c < n_valid (dz: [rows][ld],
 * C physical channels; zeros: C floats of 0).  Replaces u2_relu_bwd + u2_colsum_add for a conv with bias and ReLU. */
int u2_relu_bwd_colsum(const void* dout, const void* out, void* dz, float* dst, const float* zeros, int rows, int C, int ld, int n_valid,
                       void* stream);
/* out = a + b (+ c (+ d)) on bf16 tensors of numel elements (numel % 8 == 0), summed in fp32 and rounded once; out may alias an
 * input. The gradient sum autograd would otherwise make with k - 1 separate adds where a tensor has k consumers (the FPN
 * outputs feed the RPN head, the ROI poolers and the semantic head: meta_arch/panoptic_fpn.py:105-131). */
int u2_add_n(const void* a, const void* b, const void* c, const void* d, void* out, long long numel, void* stream);

/* The kernel-layout weight gradient of a multi-tap convolution, scratch fp32 [Npad][T][Cp] (u2_conv_wgrad), accumulated into the
 * parameter's own gradient storage fp32 [N][Cin][T] (= [Cout, Cin, kh, kw], the optimizer arena): grad[n][c][t] += scratch[n][t][c]
 * for n < N, c < Cin.  Replaces the strided `grad.add_(scratch.view(...).permute(0, 3, 1, 2))` of autograd's AccumulateGrad
 * (engine/train_loop.py:479-521 leaves that accumulation to torch). */
int u2_wgrad_permute_add(const float* scratch, float* grad, int N, int Cin, int T, int Cp, void* stream);

/* ---- pooling / resampling (pool_resize.hip) ----------------------------------------------------
 * backbone/resnet.py:358 (max_pool2d 3x3 s2 p1), backbone/fpn.py:153-155 (nearest x2 + add),
 * meta_arch/semantic_seg.py:206-211 (bilinear x2), meta_arch/rcnn.py:223-234 (normalise + pad) feeding the stem. */
int u2_maxpool3x3s2_fwd(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream);
int u2_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int B, int H, int W, int C, void* stream);
/* Round 6: the stem's tail - norm -> relu_ -> max_pool2d (backbone/resnet.py:355-359, layers/batch_norm.py:169-197) - without the
 * 550 MB activation between the normalisation and the pool.  Forward: y / idx of u2_maxpool3x3s2_fwd applied to
 * u2_affine_act(x, scale, shift, relu = 1), bit for bit (the nine taps are normalised, rounded to bf16 and compared as the two
 * launches do).  Backward: u2_norm_bwd_reduce / u2_norm_bwd_apply (mask recomputed from x * mask_scale + mask_shift > 0) on the
 * gradient u2_maxpool3x3s2_bwd would have stored, which is rebuilt per input pixel from dy / idx instead:
 * sums[0][c] += sum dz, sums[1][c] += sum dz * (x - mean) * invstd (fp32 [2][C], zeroed by the caller); dx = k1 dz + k2 x + k3 with
 * the coefficients of u2_bn_finalize_bwd.  x: [B][H][W][C] bf16 conv output, y / idx / dy: [B][Ho][Wo][C]; C / 8 must divide 256. */
int u2_affine_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, void* idx, int B, int H, int W,
                               int C, void* stream);
int u2_affine_relu_maxpool_bwd_reduce(const void* dy, const void* idx, const void* x, const float* mean, const float* invstd,
                                      const float* mask_scale, const float* mask_shift, float* sums, int B, int H, int W, int C,
                                      void* stream);
int u2_affine_relu_maxpool_bwd_apply(const void* dy, const void* idx, const void* x, const float* k1, const float* k2,
                                     const float* k3, const float* mask_scale, const float* mask_shift, void* dx, int B, int H, int W,
                                     int C, void* stream);
int u2_fpn_upsample_add_fwd(const void* lateral, const void* top, void* out, int B, int H, int W, int C, void* stream);
int u2_fpn_upsample_add_bwd(const void* dout, void* dtop, int B, int H, int W, int C, void* stream);
int u2_bilinear_up2_fwd(const void* x, const void* addend /*nullable, out = up2(x) + addend*/, void* out, int B, int H,
                        int W, int C, void* stream);
int u2_bilinear_up2_bwd(const void* dout, void* dx, int B, int H, int W, int C, void* stream);
int u2_stem_im2col(const void* img, int is_uint8, const float* mean, const float* stdv, void* col, int b, int h, int w,
                   int Hpad, int Wpad, int KP, void* stream);
/* Gradient of p6 = p5[:, ::2, ::2, :] (LastLevelMaxPool, detectron2/modeling/backbone/fpn.py:188-200): g bf16 [B][ceil(H/2)][ceil(W/2)][C]
 * -> dx bf16 [B][H][W][C] = g at even (y, x), zero elsewhere.  C % 8 == 0. */
int u2_subsample2_bwd(const void* g, void* dx, int B, int H, int W, int C, void* stream);
/* ImageList.from_tensors(gt_sem_seg, size_divisibility, ignore_value) (detectron2/structures/image_list.py:70-122 as called by
 * modeling/meta_arch/panoptic_fpn.py:118-126) for the label maps of a batch in one launch: out uint8 [n_imgs][Hpad][Wpad]
 * (16-byte aligned, Wpad % 16 == 0) = the image's labels (int64 if is_int64, else uint8; host arrays of device pointers / sizes,
 * h <= Hpad, w <= Wpad) in the top-left corner, `pad` elsewhere. */
int u2_label_pad_batch(const void* const* imgs, const int* hs, const int* ws, int n_imgs, int is_int64, void* out, int Hpad,
                       int Wpad, int pad, void* stream);
/* All images of the batch (host arrays of device pointers / sizes; images may differ in size) in one launch. */
int u2_stem_im2col_batch(const void* const* imgs, const int* hs, const int* ws, int n_imgs, int is_uint8, const float* mean,
                         const float* stdv, void* col, int Hpad, int Wpad, int KP, void* stream);

/* ---- losses (losses.hip) -----------------------------------------------------------------------
 * meta_arch/semantic_seg.py:255-267, roi_heads/fast_rcnn.py:307-347,424-463, roi_heads/mask_head.py:33-112,
 * proposal_generator/rpn.py:366-429, modeling/box_regression.py:43-76. */
/* grad_acc [B][h][w][LP] fp32 is overwritten with the un-normalised logit gradient (no pre-zeroing needed);
 * loss_sum / valid_cnt are accumulated (pre-zeroed by the caller). */
int u2_semseg_upsample_ce(const void* logits, const void* target, float* grad_acc, float* loss_sum, float* valid_cnt,
                          int B, int h, int w, int LP, int NC, int ignore, void* stream);
int u2_scale_to_bf16(const float* acc, const float* num, const float* den, float mult, void* out, long long n,
                     void* stream);
int u2_softmax_ce(const void* logits, const void* labels, void* dlogits, float* loss_sum, int R, int NC, int LP,
                  float gscale, void* stream);
/* (u2_mask_predict_bce: phased_side = 0: x / dx are [N][P][C] in pixel order; phased_side = S2 (P = S2 * S2): they are the
 * 2x2 / stride-2 deconvolution's unshuffled GEMM output [N][S2/2][S2/2][2][2][C], target / logit_out stay in pixel order.
 * dx, dWp, dbp, loss_sum may each be NULL (not produced): the forward pass asks for the loss alone, the backward pass for the
 * gradients with gmul = the loss's upstream gradient, a device scalar the gradient scale gscale is multiplied with.) */
int u2_mask_predict_bce(const void* x, const float* Wp, const float* bp, const void* cls, const void* target, void* dx,
                        float* dWp, float* dbp, float* loss_sum, void* logit_out, int N, int P, int C, float gscale,
                        int phased_side, const float* gmul, void* stream);
/* proposal_generator/rpn.py:366-429, one feature level.  dlt == NULL (and ddlt == NULL, A == 3): `obj` is the output of the
 * objectness and anchor-delta 1x1 convs run as ONE conv (columns 0-2 objectness, 3-14 deltas of an LPo-wide NHWC map) and dobj
 * receives both gradients in the same columns - the map their common input's gradient is then formed from by one data-gradient
 * conv instead of two and a sum. */
int u2_rpn_loss_level(const void* obj, const void* dlt, const void* labels, const int* match, const float* gt,
                      const float* anchors, void* dobj, void* ddlt, float* loss, int B, int HW, int A, int LPo, int LPd,
                      int Atot, int lvl_off, int G, float gscale, void* stream);
int u2_box_reg_l1(const void* pred, const float* prop, const float* gtb, const void* labels, void* dpred, float* loss,
                  int R, int LP, int bg_label, float wx, float wy, float ww, float wh, float gscale, void* stream);

/* ---- ROI bookkeeping (roi.hip) -----------------------------------------------------------------
 * layers/roi_align.py:49-65 via modeling/poolers.py:206-263; structures/masks.py:191-218; modeling/poolers.py:23-59;
 * structures/boxes.py:312-358 + modeling/matcher.py:62-127; modeling/box_regression.py:78-116; layers/nms.py:9-20. */
/* `order` (may be NULL): a permutation of 0 .. R-1, the order the ROIs are PROCESSED in (results stay at their own rows of `out`):
 * with ROIs sorted by (level, image, row) the work-groups running at any one time read neighbouring feature rows, and each XCD
 * takes a contiguous part of the list, so its L2 keeps the band it is working on. */
int u2_roi_align_fwd(const void* const* feats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                     const float* rois, const int* level, const int* order, void* out, int R, int C, int PH, int PW,
                     void* stream);
int u2_roi_align_bwd(float* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                     const float* rois, const int* level, const void* dout, int R, int C, int PH, int PW, float gscale,
                     void* stream);
/* atomics-free backward: `order` lists ROI ids grouped by (image, level), seg[b*nlevels + l] .. seg[.. + 1] is the
 * group's range; gfeats[l] are bf16 NHWC gradient maps written in full (no pre-zeroing needed). */
int u2_roi_align_bwd_gather(void* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                            const float* rois, const int* order, const int* seg, const void* dout, int B, int C, int PH,
                            int PW, float gscale, void* stream);
/* ROIs grouped by (image, level) for u2_roi_align_bwd_gather(_multi): order int32 [R] = ROI indices sorted by
 * key = image * nlevels + level with equal keys in index order (stable, = torch.argsort(key, stable=True)); seg int32
 * [num_images * nlevels + 1] = first position of every key, seg[last] = R.  rois [R][5] (image index first), level int32 [R].
 * num_images * nlevels <= 256, R <= 32768 and 512 * (num_images * nlevels + 2) + 2 * R <= 156 KB (the sort runs in one
 * work-group's LDS); -1 otherwise. */
int u2_roi_group(const float* rois, const int* level, int* order, int* seg, int R, int num_images, int nlevels, void* stream);
/* The same gather over up to four ROI sets at once (the cascade's three box poolers and the mask pooler share the FPN
 * maps): set i has its own rois / order / seg / dout, pooled size P[i] x P[i] and gradient scale.  Each level's gradient
 * map is written once; the per-set gradient maps and their sums (autograd's accumulation) are never formed. */
int u2_roi_align_bwd_gather_multi(void* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                                  int nsets, const void* const* rois, const void* const* order, const void* const* seg,
                                  const void* const* dout, const int* P, const float* gscale, int B, int C, void* stream);
/* The same launch for the levels of `level_mask` only (bit l; gfeats[l] of the others may be NULL), with up to two more gradient
 * maps per level added in fp32 before the one bf16 rounding: add0[l] / add1[l] (NULL arrays or NULL entries: none) are bf16 maps of
 * gfeats[l]'s shape - the gradients the map's OTHER readers produced (RPN head, semantic head: meta_arch/panoptic_fpn.py:105-131),
 * i.e. autograd's accumulation over a tensor with several consumers happens where the last addend is formed. */
int u2_roi_align_bwd_gather_sum(void* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                                int level_mask, int nsets, const void* const* rois, const void* const* order,
                                const void* const* seg, const void* const* dout, const int* P, const float* gscale,
                                const void* const* add0, const void* const* add1, int B, int C, void* stream);
int u2_mask_crop(const void* masks, const float* rois, void* out, int R, int H, int W, int P, void* stream);
/* The crops of a whole batch in one launch: image i's bitmaps are mask_bases[i] ([K_i][Hs[i]][Ws[i]] uint8, host arrays of
 * device pointers / sizes), roi_image[r] (device) is the image of ROI r, rois[r] = (bitmap row in that image, x0, y0, x1, y1). */
int u2_mask_crop_batch(const void* const* mask_bases, const int* Hs, const int* Ws, int n_images, const float* rois,
                       const int* roi_image, void* out, int R, int P, void* stream);
int u2_assign_levels(const float* boxes, int* level, int n, int min_level, int max_level, float canonical_size,
                     int canonical_level, void* stream);
int u2_iou_match(const float* boxes, int per_image_boxes, const float* gt, const int* ngt, int* match, float* mval,
                 unsigned int* gt_max, signed char* labels, int B, int n, int G, float lo, float hi,
                 int allow_low_quality, void* stream);
int u2_apply_deltas(const float* src, const float* deltas, const int* img, const float* sizes, float* out, int n,
                    float wx, float wy, float ww, float wh, float clamp, int do_clip, void* stream);
long long u2_nms_workspace_bytes(int B, int n);
int u2_batched_nms(const float* boxes, const int* group, const int* cnt, void* workspace, int* keep, int* nkeep, int B,
                   int n, float thr, int max_keep, void* stream);

/* ---- selection with a total order (select.hip) -----------------------------------------------
 * proposal_generator/proposal_utils.py:79-96 (per-level pre-NMS topk on the objectness logits, the score sort in front of
 * batched_nms) and modeling/sampling.py:38-54 in its "k smallest random keys" form.  Every row is ranked by
 * (value descending if largest else ascending, index ascending) - torch.topk leaves the order of ties open.
 *   vals: fp32 (dtype 0) or bf16 (dtype 1); element i of row r is vals[r * row_stride + (i / group) * pitch + i % group]
 *         (group = pitch = 1: contiguous rows; group = A, pitch = 32: the A valid columns of a 32-wide NHWC logit map);
 *   mask (optional, int8 [rows][n]): only elements with mask == mask_value take part;
 *   out_vals (optional) / out_idx [rows][k]: the k best in rank order; entries beyond out_cnt[r] = min(k, participants)
 *   are (-/+inf, 0).  n < 2^24, k <= 16384;
 *   idx_in (optional, int32 [rows][n], values in [0, 2^24)): the index element i stands for - ties are broken on it and it
 *   is what out_idx reports - so that the survivors of a first selection over segments of a long row can be merged by a
 *   second call with the same total order. */
int u2_topk_rows(const void* vals, int dtype, int rows, int n, long long row_stride, int group, int pitch,
                 const signed char* mask, int mask_value, int k, int largest, float* out_vals, int* out_idx,
                 int* out_cnt, const int* idx_in, void* stream);

/* Proposal decoding of all feature levels in one launch (proposal_generator/rpn.py:482-533, proposal_utils.py:56-91, the part
 * between the per-level top-k and the NMS): for level l and image b the k_l candidates idx[b][j] (anchor index = pixel * A + anchor,
 * ranked) with logits scores[b][j] are decoded from the NHWC bf16 deltas map (`deltas` points at channel 0 of the 4 A delta channels
 * of pixel 0, `pitch` channels per pixel) against anchors [hwa][4], clipped to sizes[b] = (h, w), and written as rows of kmax >= k_l:
 * boxes / scores [B * L][kmax] with row = b * L + l (padding: zero box, score -3e38), keep[row][j] = finite and wider / taller than
 * min_size; *nonfinite (pre-zeroed) counts candidates with a non-finite box or logit. */
typedef struct U2RpnLevel {
  const void* deltas;
  const float* anchors;
  const int* idx;
  const float* scores;
  int hwa, k, pitch, reserved;
} U2RpnLevel;
int u2_rpn_decode(const U2RpnLevel* levels, int L, int A, int B, int kmax, const float* sizes, float wx, float wy, float ww, float wh,
                  float clamp, float min_size, float* boxes, float* scores, signed char* keep, int* nonfinite, void* stream);

/* Several selections in one launch (up to 8 segments; a work-group per row of every segment): the per-level pre-NMS top-k of
 * proposal_utils.py:79-96 over all feature levels at once, or the positive and the negative draw of sampling.py:38-54 over the
 * same keys.  Fields as the arguments of u2_topk_rows, plus:
 *   cnt_in / cnt_group: element i takes part iff i % cnt_group < cnt_in[row * (n / cnt_group) + i / cnt_group] - the rows of a
 *     first selection (k entries each, cnt real ones) concatenated as the input of a merging second selection;
 *   idx_mod / idx_mul: the reported index is the element's index + (row % idx_mod) * idx_mul - rows that are equal segments of a
 *     longer row report positions in that row (idx_mod = 1, idx_mul = 0: none);
 *   dtype 2: fp32 storage of bf16-representable values (the fp32 out_vals of a first selection over a bf16 map): ranked on the
 *     upper 16 bits only. */
typedef struct U2TopkSeg {
  const void* vals;
  const signed char* mask;
  const int* idx_in;
  const int* cnt_in;
  float* out_vals;
  int* out_idx;
  int* out_cnt;
  long long row_stride;
  int dtype, rows, n, group, pitch, mask_value, k, largest;
  int cnt_group, idx_mod, idx_mul, reserved;
} U2TopkSeg;
int u2_topk_rows_multi(const U2TopkSeg* segs, int nseg, void* stream);

/* ---- inference tails (postprocess.hip) -------------------------------------------------------
 * layers/mask_ops.py:17-147 (paste_masks_in_image, GPU branch), meta_arch/panoptic_fpn.py:184-269. */
/* meta_arch/semantic_seg.py:240-244 + panoptic_fpn.py:173 at inference: logits [B][H][W][Cp] NHWC bf16 (K valid channels) ->
 * out (optional) fp32 NCHW [B][K][H*S][W*S] = F.interpolate(logits.float(), scale_factor=S, mode="bilinear",
 * align_corners=False) and argmax (optional) int64 [B][H*S][W*S] = out.argmax(1) (first maximum), one pass. */
int u2_semseg_upsample(const void* logits, float* out, long long* argmax, int B, int H, int W, int Cp, int K, int S,
                       void* stream);
/* modeling/postprocessing.py:77-100 (sem_seg_postprocess): the [C][Hin][Win] fp32 window of a larger map (channel stride in_cs,
 * row stride in_rs, in elements) -> out fp32 [C][Hout][Wout] = F.interpolate(window, size=(Hout, Wout), mode="bilinear",
 * align_corners=False): source index scale * (dst + 0.5) - 0.5 clamped at 0 with scale = in / out in fp32, like ATen. */
int u2_bilinear_resize_f32(const float* in, float* out, int C, int Hin, int Win, long long in_cs, long long in_rs, int Hout,
                           int Wout, void* stream);
/* out[k][y][x] (uint8 0/1) = bilinear sample of probs[k] (P x P fp32) on F.grid_sample(align_corners=False)'s grid over
 * boxes[k] = (x0, y0, x1, y1), zero outside the map, >= threshold.  `out` is 16-byte aligned. */
int u2_paste_masks(const float* probs, const float* boxes, void* out, int n, int P, int H, int W, float threshold,
                   void* stream);
/* mask_rcnn_inference (detectron2/modeling/roi_heads/mask_head.py:115-158) with the 1x1 predictor folded in: only the predicted
 * class's channel is formed.  x: bf16 [N][S2][S2][C] trunk output (C % 8 == 0, C <= 1024), or with phased = 1 the 2x2 / stride-2 deconvolution's
 * GEMM output before its pixel shuffle, [N][S2/2][S2/2][2][2][C]; Wp [K][C], bp [K] fp32 master parameters (rounded to bf16 like the
 * conv operands); cls int64 [N]; prob fp32 [N][S2][S2] = sigmoid(bf16(x . Wp[cls] + bp[cls])). */
int u2_mask_predict_prob(const void* x, const float* Wp, const float* bp, const void* cls, float* prob, int N, int S2, int C,
                         int phased, void* stream);
/* The same for the masks of a batch of images in one launch (detectron2/modeling/postprocessing.py:9-74 pastes per image):
 * image i owns rows [first, first + n) of probs / boxes and writes its n canvases of H x W bytes at byte out_offset (a multiple
 * of 16) of the 16-byte aligned `out` (the padding bytes between two images' canvases may be zeroed); `images` is a HOST array. */
typedef struct U2PasteImage { int first, n, H, W; long long out_offset; } U2PasteImage;
int u2_paste_masks_batch(const float* probs, const float* boxes, void* out, const U2PasteImage* images, int num_images, int P,
                         float threshold, void* stream);
/* Panoptic merge of a batch of images (one work-group each); `images` is a HOST array of descriptors holding device
 * pointers.  inst_segment[rank] = segment id given to the rank-th highest scoring instance (0 = rejected); stuff_segment /
 * stuff_area [num_sem] = id (0 = rejected) and free-pixel area per semantic label.  boxes (optional) + mask_res bound the
 * pixels a pasted mask can occupy (half a source texel past the box); boxes == NULL scans the whole image. */
typedef struct U2PanopticImage {
  const unsigned char* masks; /* [K][H][W] 0/1 */
  const int* order;           /* [K] instance index by descending score */
  const float* scores_sorted; /* [K] */
  const float* boxes;         /* [K][4] or NULL */
  const long long* semantic;  /* [H][W] argmax of the semantic head, rows sem_stride elements apart */
  int* panoptic;              /* [H][W] out */
  int* inst_segment;          /* [K] out */
  int* stuff_segment;         /* [num_sem] out */
  int* stuff_area;            /* [num_sem] out */
  int K, H, W, num_sem;
  int sem_stride;             /* >= W: the label map may be a window of a wider (padded) map */
} U2PanopticImage;
int u2_panoptic_merge(const U2PanopticImage* images, int n_images, float overlap_thr, int stuff_area_thr, float score_thr,
                      int mask_res, void* stream);

/* ---- optimizer (optim.hip): solver/build.py:36-37,63-73,119-139 ------------------------------- */
int u2_sgd_clip_step(float* params, const float* grads, float* momentum_buf, const int* chunk_tensor,
                     const long long* chunk_begin, const int* chunk_len, int n_chunks, float* partial /*[n_chunks]*/,
                     const int* tensor_first_chunk /*[n_tensors + 1]*/, const float* wd_per_tensor, float lr, float momentum,
                     float clip, float grad_scale, void* stream);

/* ---- k-means over DINO embeddings (kmeans.hip): u2seg/Instance_Clustering/shared/utils/nn_utils.py:304-379 ---- */
/* labels[i] = argmin_j |x_i - c_j|^2 (first minimum).  workspace: u2_kmeans_assign_workspace_floats(N, D, K) floats (contents
 * need no initialisation; keep the same buffer between the iterations of a run: it also carries the screening state).  The
 * distances are screened on bf16 MFMA - first with the leading bf16 piece of x and c alone, then, for the points that pass
 * cannot decide, with the split products hi.hi + hi.lo + lo.hi - and every point whose two best candidates are closer than
 * the screening error bound is labelled again with exact fp32 products, so the result is that of the exact kernel
 * (exact_only = 1 runs only that one; it is also what D % 32 != 0 or K > 320 fall back to).  After the call
 * ((int*)workspace)[((K + 3) & ~3) + 1] holds the number of points re-checked exactly, [... + 2] the number the first pass
 * left undecided. */
long long u2_kmeans_assign_workspace_floats(int N, int D, int K);
int u2_kmeans_assign(const float* x, const float* c, float* workspace, long long* labels, int N, int D, int K, int exact_only,
                     void* stream);
/* The same with a shadow of x: x does not change between the Lloyd iterations of a run (nn_utils.py:352 builds its x_i once,
 * outside the loop), so u2_kmeans_prepare writes mu = the column mean of x, fp16(S (x - mu)) with a power-of-two scale S (argmin_j
 * |x - c_j|^2 does not change under a common translation, and the screening margin is proportional to the norms of what is
 * multiplied; fp16 has the MFMA rate of bf16 and three more bits), the norm of x - mu, the norm of what the rounding dropped and |x_p|
 * once -
 * u2_kmeans_shadow_floats(N, D) floats, 0 when D % 32 != 0 - and the first screening pass of every later E step streams 2 instead
 * of 4 bytes per element, with a margin per point from those norms instead of a worst case.  Labels are the exact kernel's either way;
 * shadow = NULL: u2_kmeans_assign (first pass on bf16(x), rounded on the fly).  With a shadow the
 * screening also serves 320 < K <= 1280 (one first pass per block of 320 centroids, then the exact kernel for the undecided points;
 * [... + 1] counts those, [... + 2] is not written); without one such K run the exact kernel alone.
 * The shadow must be re-made when x changes. */
long long u2_kmeans_shadow_floats(int N, int D);
int u2_kmeans_prepare(const float* x, float* shadow, int N, int D, void* stream);
int u2_kmeans_assign_shadow(const float* x, const float* shadow, const float* c, float* workspace, long long* labels, int N, int D,
                            int K, int exact_only, void* stream);
/* csum [K][D] += sum of the rows of x by label, counts [K] += label histogram (both pre-zeroed by the caller, or holding
 * another shard's partial sums).  workspace (optional, u2_kmeans_update_workspace_floats(N, D, K) floats): the points are
 * bucketed by label there and the sums become a segmented reduction that reads every row of x once, whole (0.75 -> ~4 TB/s);
 * without it a privatised-LDS kernel is used. */
long long u2_kmeans_update_workspace_floats(int N, int D, int K);
int u2_kmeans_update(const float* x, const long long* labels, float* csum, float* counts, int N, int D, int K,
                     float* workspace, void* stream);
int u2_kmeans_finalize(const float* csum, const float* counts, float* c, int D, int K, void* stream);

/* ---- k nearest neighbours over the same embeddings (knn.hip): nn_utils.py:204-299 (kNN, partitioned_kNN) ----
 * d_knn[i][0..K) = the K smallest sum_d (x_query[i][d] - x_train[j][d])^2 in ascending order (ties: smaller j first),
 * ind_knn[i][k] = the train row j it belongs to; what pykeops' Kmin_argKmin(K, dim=1) returns at nn_utils.py:216 and what
 * partitioned_kNN's merge over 130 000-row partitions (:226-266) reduces to.  D % 16 == 0, 1 <= K <= 28, Nt >= K (else -2).
 * workspace: u2_knn_workspace_ints() 4-byte words of device memory. */
int u2_knn_workspace_ints(int Nq, int Nt, int D, int K, long long* n_ints);
int u2_knn(const float* x_query, const float* x_train, void* workspace, float* d_knn /*[Nq][K]*/,
           long long* ind_knn /*[Nq][K]*/, int Nq, int Nt, int D, int K, void* stream);

/* library self-description */
int u2_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
