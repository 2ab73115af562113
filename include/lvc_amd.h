/*
 * lvc_amd.h -- C ABI of liblvc_amd.so: the MI355X (gfx950) kernels of the lvc hot path.
 *
 * The reference (prannaykaul/lvc) has no C ABI for this path: it sits behind Python registries and
 * nn.Module call signatures, with exactly one native op of its own
 * (detectron2/layers/csrc/vision.cpp:96-97, roi_align_forward/backward) and torchvision's nms.  This header
 * is the drop-in boundary a maintainer binds instead (INTEGRATION.md shows the ctypes / torch.library
 * stubs).  Conventions:
 *   - extern "C", plain pointers and sizes; no torch types.  All data pointers are DEVICE pointers unless
 *     marked [host]; tensors are dense fp32 / int32 / int64 as stated.
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     returns immediately; nothing here synchronises the device (the reference's ROIAlign does:
 *     ROIAlign_cuda.cu:364, a defect we do not reproduce).
 *   - return value: 0 = LVC_OK, 1 = LVC_ERR_INVALID (bad argument), 2 = LVC_ERR_HIP (launch failed);
 *     lvc_last_error() gives the message for the calling thread.  Data-dependent conditions that the
 *     reference asserts on the host (negative RoI size, ROIAlign_cpu.cpp:149-152) are reported
 *     asynchronously through a device status word (`d_status`, bit 0 = negative RoI size, bit 1 =
 *     detection candidate overflow) which the host reads together with the results.
 *   - ownership: the caller owns every buffer, including workspaces (sizes from the *_workspace_bytes
 *     queries); the library allocates nothing and keeps no state between calls.
 *   - threading: one host thread per stream; functions are re-entrant.
 */
#ifndef LVC_AMD_H
#define LVC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define LVC_OK 0
#define LVC_ERR_INVALID 1
#define LVC_ERR_HIP 2

int lvc_abi_version(void);
const char* lvc_last_error(void);
void lvc_set_error(const char* fmt, ...);

/* ---------------------------------------------------------------------------------------------------
 * Convolution / linear layers: NHWC fp32 implicit GEMM with the per-channel affine (FrozenBN and/or bias),
 * optional residual and ReLU fused into the epilogue.  This entry point is the exact-fp32 engine
 * (v_mfma_f32_32x32x2_f32); the default engines of the detector are the fp32-accurate operand-split forms
 * further down (lvc_conv*_f16x2: two-way fp16 split, 3 fp16 MFMAs per block; lvc_conv*_bf16x3: three-way
 * bf16 split, 6 bf16 MFMAs per block), which take the same tensors and the same epilogue arguments.
 * Replaces: detectron2/layers/wrappers.py:41-99 (Conv2d+norm+activation), batch_norm.py:45-65 (FrozenBN),
 *   backbone/resnet.py:195-211 (residual add + ReLU), backbone/fpn.py:131-133 (nearest x2 upsample + add),
 *   lvc/modeling/roi_heads/box_head.py:82-91 and fast_rcnn.py:583-598 (Linear = 1x1 conv on [M,1,1,K]).
 *   x        [N,H,W,C]        C = physical channels (mode 0: C % 32 == 0; mode 1: C == 4, the RGB0 stem)
 *   w_packed [Kpad,Kg]        rows = out channel, zero rows up to a multiple of 128;
 *                             mode 0: k = (c/32, r, s, c%32), Kg = R*S*C
 *                             mode 1: k = (r, 8 pixels x 4 ch), Kg = R*32 (7x7 stem: pixel 7 / ch 3 zero)
 *   scale, shift [K] or NULL  y = acc*scale + shift
 *   residual                  res_mode 0 none | 1: [M,ldr] same rows as y | 2: [N,Ho/2,Wo/2,ldr], upsampled x2
 *   y        [N*Ho*Wo, ldy]   ldy/ldr = row strides in floats (<=0: K)
 *   workspace                 lvc_conv_workspace_bytes() bytes, zero-initialised ONCE by the caller and then
 *                             reused by every launch on the same stream (split-tile partials + flags of the
 *                             stream-K decomposition); launches on different streams need different workspaces
 */
long long lvc_conv_workspace_bytes(void);
/* Per-layer range words.  The workspace ends in lvc_range_slots() int32 words behind the worker flags (byte offset
 * 1024*256*128*4 + 1024*4).  Word 0 is the shared error word (bit 0: a stream-K worker timed out; bit 1: an operand left a
 * two-way fp16 split kernel's range).  lvc_set_range_slot(s), 0 < s < lvc_range_slots(), makes the fp16-split conv/GEMM
 * launches that follow ON THE CALLING THREAD raise their bits in word s instead, so the host can re-route exactly the layer
 * that overflowed (lvc_amd.kernels.check_conv_error_word); 0 restores the shared word. */
void lvc_set_range_slot(int slot);
int lvc_range_slot(void);
int lvc_range_slots(void);
int lvc_conv2d_nhwc_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                        const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S,
                        int stride, int pad, int Kg, int relu, int res_mode, int ldy, int ldr, int mode,
                        void* workspace, void* stream);

/* Same layer contract on the bf16 matrix cores with fp32-level accuracy (csrc/conv_bf16x3.hip): every fp32 operand is
 * split exactly into three bf16 values and the product assembled from six v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
 * w_split: [3][Kpad][Kg] bf16 planes (hi, mid, lo) of the packed weights of lvc_conv2d_nhwc_f32 mode 0.
 * Requires C % 32 == 0, K % 4 == 0, ldy % 4 == 0, ldr % 4 == 0.  Same workspace as lvc_conv2d_nhwc_f32. */
int lvc_conv2d_nhwc_bf16x3(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                           const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S,
                           int stride, int pad, int Kg, int relu, int res_mode, int ldy, int ldr, void* workspace,
                           void* stream);

/* 3x3 / stride 1 / pad 1 specialisation of lvc_conv2d_nhwc_bf16x3 (csrc/conv3x3_halo.hip): identical arguments minus
 * (R, S, stride, pad), identical packed weights, workspace and result contract -- the layers it serves are conv2 of
 * every BottleneckBlock (detectron2/modeling/backbone/resnet.py:178-187 with STRIDE_IN_1X1), the FPN output convs
 * (fpn.py:88-96) and the RPN head conv (rpn.py:83).  The output is tiled as 2-D pixel patches x 128 channels and the
 * (PH+2) x (PW+2) activation window of a 32-channel chunk is staged once in LDS for all nine taps. */
int lvc_conv3x3_nhwc_bf16x3(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                            const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu,
                            int res_mode, int ldy, int ldr, void* workspace, void* stream);

/* Two-way fp16 operand split (Ootomo & Yokota 2022) form of lvc_conv3x3_nhwc_bf16x3 (csrc/conv3x3_halo_h2.hip):
 * a = a1 + 2^-11 a2 with a1 = fp16(a), a2 = fp16((a - a1) * 2048); products a1 b1 (main accumulator) and a1 b2 + a2 b1
 * (cross accumulator, folded in as cross / 2048): three fp16 MFMAs per block instead of six bf16 ones, two operand planes
 * instead of three, fp32-level accuracy.  w_split: [2][Kpad][Kg] fp16 planes (w1, (w - w1) * 2048), same k order and
 * padding as the bf16 planes.  Operands beyond fp16's range (|a| > 65504, NaN) set bit 1 of the workspace error word
 * (the word after the LVC worker flags; lvc_amd.kernels.conv_error_word reads it).  No small-map fallback. */
int lvc_conv3x3_nhwc_f16x2(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                           const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu,
                           int res_mode, int ldy, int ldr, void* workspace, void* stream);
/* Single-accumulator, software-pipelined form of lvc_conv3x3_nhwc_f16x2 (csrc/conv3x3_halo_s1.hip; the forward 3x3 layers of
 * detectron2/modeling/backbone/resnet.py:195-211, fpn.py:101-113 and proposal_generator/rpn.py:108-127).  Same arguments and
 * results except the weights: w_split = the [2][Kpad][Kg] fp16 planes of lvc_split_weights_rowscaled (row k times 2^e_k, the
 * residual plane unscaled), and `scale` (never NULL) = (the layer's per-channel scale or 1) x row_factor[k] of that call.
 * Activations are multiplied by 2^4 before their split: |a| > 4094 (or NaN) sets bit 1 of the workspace error word. */
int lvc_conv3x3_nhwc_f16s1(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                           const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu,
                           int res_mode, int ldy, int ldr, void* workspace, void* stream);
/* lvc_conv3x3_nhwc_f16x2 -- same arguments, weight planes, numerics (main + cross accumulators, |a| <= 65504) and results up to
 * the fp32 summation order -- on the software-pipelined tap loop of csrc/conv3x3_halo_s1.hip (LDS-DMA weight ring, rotating
 * fragment registers, barrier in the middle of a tap, double-buffered halo window). */
int lvc_conv3x3_nhwc_f16x2_pipe(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu,
                                int res_mode, int ldy, int ldr, void* workspace, void* stream);
/* L <= 6 maps [N,Hs[l],Ws[l],C] through the SAME 3x3 / stride 1 / pad 1 layer in ONE launch of that kernel (ys[l] [N,Hs[l],Ws[l],K]; xs / ys /
 * Hs / Ws are [host] arrays): one stream-K split over the row tiles of all maps -- StandardRPNHead.conv over the pyramid levels
 * (detectron2/modeling/proposal_generator/rpn.py:112-120 calls it once per level).  oneacc: 1 = the arguments of lvc_conv3x3_nhwc_f16s1,
 * 0 = those of lvc_conv3x3_nhwc_f16x2_pipe.  No residual.  Per output pixel the arithmetic is the single-map launch's. */
int lvc_conv3x3_nhwc_f16_levels(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                                const unsigned short* w_split, const float* scale, const float* shift, int N, int C, int K, int Kg,
                                int relu, void* workspace, void* stream);
/* ... with a POINTWISE layer on top of act(conv), the hidden maps never written: StandardRPNHead.forward of the reference
 * (detectron2/modeling/proposal_generator/rpn.py:112-120: t = relu(conv(x)); objectness_logits(t), anchor_deltas(t)) as one launch over
 * the levels.  ys[l] [N,Hs[l],Ws[l],pred_ld] are the POINTWISE layer's outputs and must be ZEROED by the caller (every workgroup adds
 * its 128-channel slice of the contraction atomically: K = 128 or 256 hidden channels -> at most two addends per element, the sum does
 * not depend on their order).  oneacc must be 0 (w_split / scale / shift as for lvc_conv3x3_nhwc_f16x2_pipe).  pred_w: the
 * [2][pred_rows >= 32][K] fp16 planes lvc_split_weights writes for the pointwise weights [pred_rows][K] (rows >= pred_K zero), pred_scale /
 * pred_shift [pred_K] or NULL, 1 <= pred_K <= 32 <= pred_rows, pred_ld >= pred_K.  pred_slot: the pointwise layer's range word (a hidden
 * value beyond 65504 raises it; the 3x3 layer's own word is the slot set by lvc_set_range_slot). */
int lvc_conv3x3_nhwc_f16_levels_pred(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                                     const unsigned short* w_split, const float* scale, const float* shift, int N, int C, int K, int Kg,
                                     int relu, const unsigned short* pred_w, const float* pred_scale, const float* pred_shift, int pred_K,
                                     int pred_rows, int pred_ld, int pred_slot, void* workspace, void* stream);
/* The same with a LAYER per map (L layers of one shape and form: the FPN output convs, detectron2/modeling/backbone/fpn.py:128-141): ws /
 * scales / shifts are [host] arrays of L device pointers (shift entries may be NULL). */
int lvc_conv3x3_nhwc_f16_layers(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                                const unsigned short* const* ws, const float* const* scales, const float* const* shifts, int N, int C,
                                int K, int Kg, int relu, void* workspace, void* stream);
/* Work distribution of the two entries below: 0 (default) = one workgroup per tile; 1 = stream-K (one resident workgroup per CU takes an
 * equal share of the (tile, 16-channel chunk) list; split tiles are completed through `workspace` as in the other conv kernels, in worker
 * order: deterministic); 2 = persistent workgroups on whole tiles.  Measured: neither beats 0 on the detector's layers (conv3x3_wino.hip). */
void lvc_set_wino_streamk(int mode);
/* Process-wide A/B and test switches (no environment lookups inside the library): the NMS ordered reduce of short lists from global
 * memory instead of LDS (same keep lists); the bf16x3 3x3 halo kernel with a fixed patch shape / without its small-map fallback. */
void lvc_set_nms_reduce_global(int on);
void lvc_set_halo_test_hooks(int ph, int pw, int force);
/* 3x3 / stride 1 / pad 1 as Winograd F(2,3) along x (round 5, csrc/conv3x3_wino.hip; the 256-channel layers of
 * detectron2/modeling/backbone/fpn.py:141-144 and proposal_generator/rpn.py:92-94 on the large maps): two thirds of the MFMAs of
 * lvc_conv3x3_nhwc_f16s1 at the same operand precision and the direct evaluation's fp32 error.  u = transformed row-scaled weight planes
 * [3][C/16][4][2][Kpad][16] fp16 (Kpad % 128 == 0), scale [K] = their row factors x the layer's scale (lvc_amd.kernels.pack_wino);
 * C % 16 == 0; |window value| <= 4094 or the layer's range word in `workspace` is raised. */
int lvc_conv3x3_nhwc_wino(const float* x, const unsigned short* u, const float* scale, const float* shift, float* y, int N, int H, int W,
                          int C, int K, int Kpad, int relu, int ldy, void* workspace, void* stream);
/* ... with a pointwise layer (<= 32 outputs; the RPN predictor, rpn.py:95-106) on top of act(conv): y [N,H,W,ldy] = its outputs, ZEROED
 * by the caller, the hidden map is never written; K in {128, 256} (at most two atomically added slices per element: order-free).  pred_*
 * as for lvc_conv3x3_nhwc_f16_levels_pred. */
int lvc_conv3x3_nhwc_wino_pred(const float* x, const unsigned short* u, const float* scale, const float* shift, float* y, int N, int H,
                               int W, int C, int K, int Kpad, int relu, int ldy, const unsigned short* pred_w, const float* pred_scale,
                               const float* pred_shift, int pred_K, int pred_rows, int pred_slot, void* workspace, void* stream);
/* Pointwise (R = S = 1, pad 0) layers with a long contraction on the pipelined loop of the 3x3 kernel (csrc/conv_pw_s1.hip; the conv1
 * / FC layers of detectron2/modeling/backbone/resnet.py:195-211, roi_heads/box_head.py:80-93 and the ViT linears): y = act(conv(x,
 * w) * scale + shift (+ residual)), x [N,H,W,C] fp32 NHWC with C % 32 == 0, stride >= 1, relu: 0 none / 1 ReLU / 2 exact GELU,
 * res_mode / ldy / ldr / workspace as lvc_conv2d_nhwc_f16x2.  _f16x2_pipe: the numerics and weight planes of
 * lvc_conv2d_nhwc_f16x2 (|a| <= 65504).  _f16s1: the single-accumulator form, w_split / scale as for lvc_conv3x3_nhwc_f16s1
 * (|a| <= 4094). */
int lvc_conv1x1_nhwc_f16x2_pipe(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                const float* residual, float* y, int N, int H, int W, int C, int K, int stride, int relu,
                                int res_mode, int ldy, int ldr, void* workspace, void* stream);
int lvc_conv1x1_nhwc_f16s1(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                           const float* residual, float* y, int N, int H, int W, int C, int K, int stride, int relu,
                           int res_mode, int ldy, int ldr, void* workspace, void* stream);
/* A bottleneck's conv2 -> conv3 hand-over without the consumer's operand split (round 6; reference resnet.py:200-212).
 * lvc_conv3x3_nhwc_f16s1_presplit = lvc_conv3x3_nhwc_f16s1 with ReLU and no residual whose output y [N,H,W,K] (K % 32 == 0; same bytes as
 * the fp32 tensor) holds, per pixel and 32-channel chunk, the 32 hi halves then the 32 lo halves of the two-way fp16 split of the result
 * x 2^4 -- the planes lvc_conv1x1_nhwc_f16s1's own split would form; a value beyond |a| <= 4094 raises range word `next_slot` (the
 * consumer's).  lvc_conv1x1_nhwc_f16s1_presplit = lvc_conv1x1_nhwc_f16s1 (stride 1, K > 64) reading such planes: the same products in
 * the same order, results bit-identical to the fp32 hand-over. */
int lvc_conv3x3_nhwc_f16s1_presplit(const float* x, const unsigned short* w_split, const float* scale, const float* shift, void* y,
                                    int N, int H, int W, int C, int K, int Kg, int next_slot, void* workspace, void* stream);
int lvc_conv1x1_nhwc_f16s1_presplit(const void* x, const unsigned short* w_split, const float* scale, const float* shift,
                                    const float* residual, float* y, int N, int H, int W, int C, int K, int relu,
                                    int res_mode, int ldy, int ldr, void* workspace, void* stream);
/* _f16s1_w2 (round 5, csrc/conv_pw_w2.hip): the single-accumulator form on a 256-row x 256-channel workgroup tile (wave tile 64 x
 * 128, 16-deep stages) for layers with >= 256 input and output channels -- res4 / res5 conv1 and conv3, the FPN laterals
 * (detectron2/modeling/backbone/fpn.py:128-140), box-head fc1 / fc2: 0.67 x the operand bytes per MFMA of the 256 x 128 tile.  Same
 * operands and results as lvc_conv1x1_nhwc_f16s1 up to the fp32 summation order; C % 16 == 0; Kpad = rows of a weight plane
 * (>= K rounded up to 256). */
int lvc_conv1x1_nhwc_f16s1_w2(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                              const float* residual, float* y, int N, int H, int W, int C, int K, int Kpad, int stride, int relu,
                              int res_mode, int ldy, int ldr, void* workspace, void* stream);
/* Two chained pointwise layers in one launch, the second fed from the first one's accumulators (csrc/conv_pw_chain.hip):
 *   y1 = act1(x Wa^T * sa + ta (+ residual))   -- a bottleneck's conv3 + FrozenBN + shortcut add + ReLU,
 *   y2 = act2(y1 Wb^T * sb + tb)               -- the next bottleneck's conv1 + FrozenBN + ReLU,
 * detectron2/modeling/backbone/resnet.py:205-211 of block i followed by :195-197 of block i+1; both results are stored, y1 is never
 * read back.  x [M][ldx] (K1 channels), residual [M][ldr] or NULL, y1 [M][ldy1] (N1), y2 [M][ldy2] (N2).  wa [2][wa_rows][K1] /
 * wb [2][wb_rows][N1]: fp16 planes of lvc_split_weights_rowscaled over weights whose contraction index is permuted within every
 * 16 entries (0-3, 8-11, 4-7, 12-15); sa / sb = (per-channel scale or 1) x that call's row factors (never NULL); ta / tb shifts
 * or NULL; relu1 / relu2: 0 none, 1 ReLU.  (K1, N1, N2) in {(64,256,64), (128,256,64), (128,512,128)}.  |x| or |y1| > 4094 (or
 * NaN) sets bit 1 of the workspace error word. */
int lvc_conv1x1_chain_nhwc_f16s1(const float* x, int ldx, const unsigned short* wa, int wa_rows, const float* sa, const float* ta,
                                 const float* residual, int ldr, float* y1, int ldy1, int relu1, const unsigned short* wb,
                                 int wb_rows, const float* sb, const float* tb, float* y2, int ldy2, int relu2, int M, int K1,
                                 int N1, int N2, void* workspace, void* stream);
/* A whole bottleneck block in one launch (round 6, csrc/conv_bneck.hip), replacing the three conv + FrozenBN launches and the shortcut
 * add of detectron2/modeling/backbone/resnet.py:195-211 for the 64-mid / 256-output-channel blocks of res2 (stride 1):
 *   y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + shortcut(x)).
 * conv1's output lives in LDS (output tile + one-pixel halo, fp16 planes), conv2's in registers; x is read once, y written once.
 * x [N][H][W][ldx] (cin channels used), y [N][H][W][ldy].  proj = 0: cin = 256, shortcut = x.  proj = 1: cin = 64, the projection
 * shortcut's weights are the last 64 contraction columns of the third layer.  w: stage images built by
 * lvc_amd.kernels.pack_bottleneck (row-scaled fp16 planes of lvc_split_weights_rowscaled in fragment order: (cin/32) x 8 KB,
 * 18 x 8 KB, 8 or 16 x 8 KB: 34 or 36 stages of eight 1 KB fragments); s1/t1, s2/t2 (64 entries), s3/t3 (256): epilogue scales (x row factors) and shifts, never NULL.
 * |x|, |conv1 out| or |conv2 out| > 4094 (or non-finite) sets bit 1 / 2 of the launch's range word. */
int lvc_bottleneck_nhwc_f16s1(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int cin, int proj,
                              const unsigned short* w, const float* s1, const float* t1, const float* s2, const float* t2,
                              const float* s3, const float* t3, void* workspace, void* stream);
/* wp [rows][Kg] fp32 (lvc_pack_conv_weights) -> planes_out [2][rows][Kg] fp16: w1 = fp16(wp 2^e), w2 = fp16(wp 2^e - w1) with
 * e = 13 - floor(log2(max |wp[row][:]|)) per row (0 for an all-zero row); row_factor[row] = 2^-(e + 4). */
int lvc_split_weights_rowscaled(const float* wp, int rows, int Kg, void* planes_out, float* row_factor, void* stream);

/* Two-way fp16 split form of the POINTWISE shapes of lvc_conv2d_nhwc_bf16x3 (csrc/conv_f16x2.hip): R = S = 1, pad 0
 * and (C <= 512 or N*Ho*Wo >= 2048); anything else returns LVC_ERR_INVALID.  w_split as lvc_conv3x3_nhwc_f16x2. */
int lvc_conv2d_nhwc_f16x2(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                          const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int Kg, int relu, int res_mode, int ldy, int ldr, void* workspace,
                          void* stream);
/* The same contract for pointwise layers (R = S = 1, pad 0, C % 32 == 0; y and residual below 2 GiB), operands streamed by
 * LDS-DMA (csrc/conv_pw_dma.hip: raw fp32 activation rows and the fp16 weight planes go HBM/L2 -> LDS by
 * global_load_lds_dwordx4 in a three-stage ring that runs across tile boundaries, the fp16 split happens in registers at
 * fragment time, the epilogue stores from the accumulators).  Results equal lvc_conv2d_nhwc_f16x2's to fp32 rounding
 * (same products, same accumulation order per tile).  LVC_ERR_INVALID for non-pointwise shapes. */
int lvc_conv2d_nhwc_f16x2_dma(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                          const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int Kg, int relu, int res_mode, int ldy, int ldr, void* workspace,
                          void* stream);

/* BasicStem in one launch (detectron2/modeling/backbone/resnet.py:588-592): conv 7x7 s2 p3 (3 -> 64) -> FrozenBN fold
 * (scale/shift, NULL = identity) -> ReLU -> max_pool2d 3x3 s2 p1, with the two-way fp16 operand split (csrc/stem_pool_h2.hip).
 * x [N,H,W,4] NHWC4 (the layout lvc_preprocess_nhwc4 writes), w_split = the [2][Kpad][224] fp16 planes of the mode-1 packed
 * stem weights of lvc_conv2d_nhwc_f32 (k = r*32 + s*4 + c), Kpad >= 64 rows per plane; y [N,Hp,Wp,64] with Ho = (H-1)/2+1,
 * Hp = (Ho-1)/2+1 (same for W).  d_error_word: device int whose bit 1 is set when an input beyond fp16's range is met (may be
 * NULL).  (The three-way bf16 form of this kernel was removed in round 4: under the range-free split the stem runs as conv +
 * max-pool, two launches.) */
int lvc_stem_conv_pool_nhwc4_f16x2(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                   float* y, int N, int H, int W, int Kpad, int relu, int* d_error_word,
                                   float* y2 /* optional second copy of y, rows of ldy2 floats */, int ldy2, void* stream);

/* GeneralizedRCNN.preprocess_image (lvc/modeling/meta_arch/rcnn.py:324-333) + ImageList.from_tensors
 * padding (detectron2/structures/image_list.py:95-119): out[y,x,:] = ((img[:,y,x]-mean)/std, 0), zero
 * outside h x w.  image: CHW, dtype 0 = fp32, 1 = uint8.  mean3/std3 are [host] arrays of 3 floats. */
int lvc_preprocess_nhwc4(const void* image, int dtype, int h, int w, const float* mean3, const float* std3,
                         float* out, int Hp, int Wp, void* stream);
/* The same for B images in one launch (16 per launch): images[i] -> out[i] ([B,Hp,Wp,4]); all float32 or all uint8. */
int lvc_preprocess_batch_nhwc4(const void* const* images, int dtype, const int* hs, const int* ws, int B, const float* mean3,
                               const float* std3, float* out, int Hp, int Wp, void* stream);

/* Test-time input pipeline: ResizeShortestEdge's Pillow bilinear resize of a uint8 HWC image
 * (detectron2/data/transforms/transform.py:101-109), bit-exact with Pillow's ImagingResample (22-bit fixed point,
 * horizontal then vertical pass, uint8 intermediate), optionally fused with preprocess_image's normalise + zero-pad
 * (lvc/modeling/meta_arch/rcnn.py:324-333) into the batch's NHWC4 slot.
 *   image [H,W,3] u8; xb [new_w,2] / xk [new_w,kxs] int32 = (first source column, count) and coefficients of each
 *   output column (NULL when new_w == W; host: lvc_amd/data/transforms.py resample_coeffs = Pillow's
 *   precompute_coeffs + normalize_coeffs_8bpc); yb / yk / kys likewise for rows; tmp: H*new_w*3 bytes of scratch;
 *   out_u8 [new_h,new_w,3] and out_nhwc4 [Hp,Wp,4] fp32 are each optional; mean3 / std3: host float[3]. */
int lvc_resize_bilinear_u8(const unsigned char* image, int H, int W, int new_h, int new_w, const int* xb, const int* xk,
                           int kxs, const int* yb, const int* yk, int kys, unsigned char* tmp, unsigned char* out_u8,
                           float* out_nhwc4, int Hp, int Wp, const float* mean3, const float* std3, void* stream);
/* F.max_pool2d on NHWC (BasicStem resnet.py:591: k3 s2 p1; LastLevelMaxPool fpn.py:176: k1 s2 p0). */
int lvc_maxpool2d_nhwc(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad,
                       void* stream);

/* Descriptor crops (lvc/data/utils.py:485-519 get_crops_qe): zero-pad the box window to a square and resize to
 * out_size x out_size with F.interpolate(mode='nearest').  image [C,H,W]; d_windows [K,8] int32 =
 * (x1,y1,x2,y2 inclusive, l_pad, t_pad, side_w, side_h) (integer bookkeeping done by the host); out [K,C,out,out]. */
int lvc_crop_resize_nearest(const float* image_chw, int C, int H, int W, const int* d_windows, int K, int out_size,
                            float* out, void* stream);

/* Row-wise (x - mu) / den; den = |x-mu| + eps (mode 0, CosineSimOutputLayers fast_rcnn.py:822-833) or
 * max(|x-mu|, eps) (mode 1, F.cosine_similarity as used by tools/run_nearest_neighbours.py:150-153). */
int lvc_rownorm(const float* x, const float* mu, float* y, int M, int D, int ldx, int ldy, float eps, int mode,
                void* stream);
/* Backward of lvc_rownorm mode 0 without mu (CosineSimOutputLayers training, lvc/modeling/roi_heads/fast_rcnn.py:
 * 823-825): dx = dy/(n+eps) - x (dy.x)/((n+eps)^2 n), n = |x| per row; accumulate != 0 adds to dx. */
int lvc_rownorm_backward(const float* x, const float* dy, float* dx, int M, int D, float eps, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * ROIAlign forward.  Same arithmetic as ROIAlign_cpu.cpp:20-218 / ROIAlign_cuda.cu:65-139 (and the same operation order in the NCHW form).
 * lvc_roi_align_forward_nchw has the reference op's shape contract (csrc/vision.cpp:96):
 *   input [B,C,H,W], rois [K,5] = (batch index, x1, y1, x2, y2), output [K,C,pooled_h,pooled_w].
 * lvc_roi_align_fpn_nhwc is the engine form: L pyramid levels [B,H_l,W_l,C] (feats/Hs/Ws/scales are [host]
 * arrays of L entries), per-RoI level ids [K] int32 (NULL if L == 1), output [K,pooled_h,pooled_w,C];
 * replaces the per-level gather/ROIAlign/scatter loop of detectron2/modeling/poolers.py:236-246.
 *   The engine form sums a bin in the separable order (weights of a row and of a column pre-summed, every window pixel read once:
 *   csrc/roi_align.hip): the reference's value to fp32 rounding (1e-6 of the output scale in the tests), not its bits; the NCHW
 *   form keeps the reference's order.
 * d_num_valid (device int, may be NULL): rows >= *d_num_valid are zero-filled.
 * lvc_roi_work_order writes d_order [K], a permutation of the RoIs with the largest windows first; lvc_roi_align_fpn_nhwc_ordered
 *   lets workgroup i pool RoI d_order[i] (output row k is RoI k either way; NULL = RoI order): the launch no longer ends on a few
 *   large RoIs that started last.
 */
int lvc_roi_align_forward_nchw(const float* input, const float* rois, float* output, int B, int C, int H, int W,
                               int K, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                               int aligned, int* d_status, void* stream);
int lvc_roi_align_fpn_nhwc(const float* const* feats, const int* Hs, const int* Ws, const float* scales, int L,
                           int B, int C, const float* rois, const int* levels, const int* d_num_valid, int K,
                           int pooled_h, int pooled_w, int sampling_ratio, int aligned, float* output,
                           int* d_status, void* stream);
int lvc_roi_work_order(const float* rois, const int* levels, const float* scales, int L, int K, int pooled_h, int* d_order,
                       void* stream);
/* XCD-local order for B <= 16 images: position i holds a RoI of image i % B (workgroup i runs on XCD i % 8), inside an image by
 * (level, 16-pixel band of the box centre's row) -- the windows an XCD reads next to each other in time share its L2. */
int lvc_roi_work_order_xcd(const float* rois, const int* levels, int K, int B, int* d_order, void* stream);
int lvc_roi_align_fpn_nhwc_ordered(const float* const* feats, const int* Hs, const int* Ws, const float* scales, int L,
                                   int B, int C, const float* rois, const int* levels, const int* d_num_valid, int K,
                                   int pooled_h, int pooled_w, int sampling_ratio, int aligned, float* output,
                                   int* d_status, const int* d_order, void* stream);

/* ROIAlign backward (csrc/vision.cpp:97 roi_align_backward; ROIAlign_cuda.cu:142-306 / ROIAlign_cpu.cpp:219-406):
 *   grad [K,C,pooled_h,pooled_w] contiguous -> grad_input [B,C,H,W], zeroed by the call (the reference returns a
 *   fresh at::zeros tensor).  Scatter by fp32 atomic adds like the reference CUDA kernel: equal to the reference CPU
 *   kernel up to summation order.  The _fpn_ form is the gradient of lvc_roi_align_fpn_nhwc: grad
 *   [K,pooled_h,pooled_w,C] -> grad_feats[l] [B,H_l,W_l,C] (grad_feats/Hs/Ws/scales are [host] arrays of L entries). */
int lvc_roi_align_backward_nchw(const float* grad, const float* rois, float* grad_input, int B, int C, int H, int W,
                                int K, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                int aligned, int* d_status, void* stream);
int lvc_roi_align_fpn_backward_nhwc(const float* grad, float* const* grad_feats, const int* Hs, const int* Ws,
                                    const float* scales, int L, int B, int C, const float* rois, const int* levels,
                                    const int* d_num_valid, int K, int pooled_h, int pooled_w, int sampling_ratio,
                                    int aligned, int* d_status, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Batched NMS, B images per call.  Replaces torchvision.ops.boxes.batched_nms / nms as called from
 * detectron2/layers/nms.py:10-29 (consumers proposal_utils.py:104, lvc fast_rcnn.py:128).  Keep indices are
 * bit-exact w.r.t. the CPU algorithm (offset trick in fp32, IoU in fp32 compared against the double
 * threshold, score ties -> lower index).
 *   boxes [B,Nmax,4], scores [B,Nmax], idxs [B,Nmax] int32 or NULL, d_counts [B] int32 or NULL (= Nmax)
 *   keep [B,Nmax] int32 (indices into the image's rows, score-descending), d_num_keep [B] int32
 *   max_keep <= 0: unlimited.  Any Nmax: up to 16384 rows per image run as one LDS sort + one mask + one reduce launch;
 *   beyond that (the reference switches to a per-class loop at 40000 boxes, nms.py:22-29, same result) a global bitonic
 *   sort and the greedy pass in blocks of 16384 sorted rows.
 */
long long lvc_batched_nms_workspace_bytes(int B, int Nmax);
int lvc_batched_nms(const float* boxes, const float* scores, const int* idxs, const int* d_counts, int B,
                    int Nmax, double iou_threshold, int max_keep, int* keep, int* d_num_keep, void* workspace,
                    long long workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * RPN.predict_proposals + find_top_rpn_proposals (detectron2/modeling/proposal_generator/rpn.py:455-508,
 * proposal_utils.py:13-118, anchor_generator.py:157-178, box_regression.py:73-110), inference branch.
 *   per level l (arrays of L [host] entries):
 *     logits[l]  device ptr: objectness of anchor a at pixel p of image b = logits[l][(b*H*W + p)*ld_logit[l] + a]
 *     deltas[l]  device ptr: delta c of anchor a                          = deltas[l][(b*H*W + p)*ld_delta[l] + a*4 + c]
 *     cell_anchors[l] device ptr [A,4]; Hs, Ws, strides
 *   d_image_sizes [B,2] int32 (h, w).  Outputs: out_boxes [B,post,4], out_logits [B,post] (zero rows past
 *   d_out_count[b]).  pre_nms_topk <= 2048.
 */
long long lvc_rpn_proposals_workspace_bytes(int B, int L, int A, const int* Hs, const int* Ws, int pre_nms_topk);
int lvc_rpn_proposals(const float* const* logits, const int* ld_logit, const float* const* deltas,
                      const int* ld_delta, const float* const* cell_anchors, const int* Hs, const int* Ws,
                      const int* strides, int L, int A, int B, const int* d_image_sizes, int pre_nms_topk,
                      int post_nms_topk, double nms_thresh, float min_box_size, float scale_clamp,
                      float* out_boxes, float* out_logits, int* d_out_count, void* workspace,
                      long long workspace_bytes, void* stream);

/* assign_boxes_to_levels + convert_boxes_to_pooler_format (detectron2/modeling/poolers.py:23-59, 69-96).
 * boxes [B,R,4] -> levels [B*R] int32 (offset from min_level), rois [B*R,5] (may be NULL). */
int lvc_assign_levels_rois(const float* boxes, int B, int R, int min_level, int max_level, int canonical_box_size,
                           int canonical_level, int* levels, float* rois, void* stream);

/* FastRCNNOutputs.predict_boxes / predict_probs + fast_rcnn_inference (lvc/modeling/roi_heads/fast_rcnn.py:95-137,
 * 440-468) + optional detector_postprocess (detectron2/modeling/postprocessing.py:10-79).
 *   cls_logits [B*R, ld_cls] (K+1 used), deltas [B*R, ld_delta] (4K, or 4 when cls_agnostic), proposals [B,R,4],
 *   d_prop_count [B] or NULL, d_image_sizes [B,2] (h,w), (wx,wy,ww,wh) = ROI_BOX_HEAD.BBOX_REG_WEIGHTS,
 *   d_post [B,4] = (scale_x, scale_y, out_h, out_w) or NULL.
 *   Outputs [B,topk,*] + d_out_count [B]; out_rows = index of the proposal each detection came from.
 *   max_candidates: capacity of the (roi,class) candidate list per image (pairs above score_thresh); R*K can never
 *   overflow; a smaller capacity that does overflow sets d_status bit 1 (value 2) and the host re-runs with R*K.
 */
long long lvc_fast_rcnn_inference_workspace_bytes(int B, int max_candidates);
int lvc_fast_rcnn_inference(const float* cls_logits, int ld_cls, const float* deltas, int ld_delta, int K,
                            int cls_agnostic, const float* proposals, const int* d_prop_count, int B, int R,
                            const int* d_image_sizes, float wx, float wy, float ww, float wh, float scale_clamp,
                            float score_thresh, double nms_thresh, int topk, int max_candidates,
                            const float* d_post, float* out_boxes, float* out_scores, int* out_classes,
                            int* out_rows, int* d_out_count, int* d_status, void* workspace,
                            long long workspace_bytes, void* stream);

/* One class-agnostic cascade stage of the box corrector: BoxOnlyLayersCascade.predict_boxes
 * (lvc/modeling/roi_heads/roi_heads_cascade.py:197-211) + the clip of _create_proposals_from_boxes
 * (cascade_rcnn.py:348-369).  deltas [M, ld] (4 used), boxes [M,4] = B images x R rows, d_image_sizes [B,2] or NULL. */
int lvc_decode_boxes(const float* deltas, int ld, const float* boxes, int M, int R, const int* d_image_sizes,
                     float wx, float wy, float ww, float wh, float scale_clamp, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Training-time kernels of the fine-tune step (BASELINE config 3; only the box predictor trains).
 * lvc_match_boxes: pairwise_iou (detectron2/structures/boxes.py:315-347) + Matcher (modeling/matcher.py:61-126)
 *   without materialising the G x N matrix.  gt [G,4] (1..512), boxes [N,4]; thresholds t0 (,t1), labels l0,l1(,l2);
 *   outputs matches [N] int64 (first arg-max), labels [N] int8, matched_vals [N]; d_gt_best [G] uint32 scratch.
 * lvc_fast_rcnn_losses: FastRCNNOutputs.losses (lvc/modeling/roi_heads/fast_rcnn.py:267-279, 296-359): mean softmax CE
 *   and smooth-L1(sum)/R, plus d(loss)/d(logits) [R,K+1] and d(loss)/d(deltas) [R,4K|4].  gt_classes int64, K = bg.
 *   One wave per row; the per-row terms are added in a fixed order from the fp64 scratch row_terms [2R].
 * lvc_rpn_losses: RPN.losses (proposal_generator/rpn.py:328-400) over S sampled anchors, forward only:
 *   out = (BCE-with-logits sum, smooth-L1 sum over positives) / normalizer.
 */
int lvc_match_boxes(const float* gt, int G, const float* boxes, int N, float t0, float t1, int nthr, int l0, int l1,
                    int l2, int allow_low_quality, long long* matches, signed char* labels, float* matched_vals,
                    unsigned int* d_gt_best, void* stream);
/* The label-and-sample steps of a training forward for the whole batch (round 5, csrc/train_targets.hip; RPN.label_and_sample_anchors,
 * detectron2/modeling/proposal_generator/rpn.py:269-325, and ROIHeads.label_and_sample_proposals, lvc/modeling/roi_heads/roi_heads.py:
 * 173-278, without their per-image loops and device->host reads).
 * lvc_match_boxes_batched: lvc_match_boxes for B images in two launches.  gt [Gtot,4] = the images' boxes one after the other, gt_off
 *   [B+1] int32 DEVICE prefix of the per-image counts (0..512 each; an image without gt gets label l0, match 0 everywhere);
 *   box_img_stride 0: boxes [N,4] shared (anchors), else [B][N][4] with nbox [B] rows in use (NULL: N; rows behind: label -1).
 * lvc_subsample_batched: subsample_labels (sampling.py:10-54) per row of labels int8 [B,N] (1 positive, 0 negative, else ignored): the
 *   min(#pos, cap_pos) positives and min(#neg, bs - num_pos) negatives with the smallest keys (int64 [B,N], distinct, < 2^nbits; or NULL:
 *   keys = a pseudo-random bijection of b N + i generated from `seed` by a 4-round Feistel network, nothing to sort or read) ->
 *   sel int32 [B,bs] (positives first, each group by increasing key, -1 padded), counts int32 [B,2].  bs <= 1024.
 * lvc_rpn_gather_sampled: for sel / counts over the R = sum_l H_l W_l A anchors: logits [B bs], deltas / anchors / matched gt boxes
 *   [B bs,4], labels int8 (1 / 0 / -1 padding) read from the head's per-level outputs fused[l] [B,H_l,W_l,ld_l] (channel a objectness,
 *   A + 4a + c delta c); grid anchors = shift + cell anchor (anchor_generator.py:161-185).  Feeds lvc_rpn_losses.
 * lvc_roi_build_table: add_ground_truth_to_proposals (proposal_utils.py:121-162) into a padded table boxes [B,Wt,4], logits [B,Wt],
 *   nrow [B]; lvc_roi_gather_sampled: its sampled rows with classes (gt class of the match / K for background and padding rows). */
int lvc_match_boxes_batched(const float* gt, const int* gt_off, int Gtot, int B, const float* boxes, long long box_img_stride,
                            const int* nbox, int N, float t0, float t1, int nthr, int l0, int l1, int l2, int allow_low_quality,
                            int* matches, signed char* labels, float* vals, unsigned int* gt_best, void* stream);
long long lvc_subsample_workspace_bytes(int B);   /* zeroed by the caller before the first use; the launches leave it zeroed */
int lvc_subsample_batched(const signed char* labels, const long long* keys /* NULL: generated from seed */, unsigned long long seed,
                          int B, int N, int nbits, int cap_pos, int bs, int* sel, int* counts, void* workspace, void* stream);
int lvc_rpn_gather_sampled(const void* const* fused, const int* ld, const void* const* cell_anchors, const int* H, const int* W,
                           const int* strides, int L, int A, int B, int bs, const int* sel, const int* counts, const int* matches,
                           const float* gt, const int* gt_off, float* logits, float* deltas, float* anchors, float* gt_boxes,
                           signed char* labels, void* stream);
int lvc_roi_build_table(const float* pboxes, const float* plogits, const int* pcount, int B, int P, const float* gt, const int* gt_off,
                        float gt_logit, int Wt, float* boxes, float* logits, int* nrow, void* stream);
int lvc_roi_gather_sampled(const float* boxes, const float* logits, const int* matches, const int* sel, const int* counts,
                           const long long* gt_classes, const int* gt_off, int B, int Wt, int bs, int K, float* s_boxes, float* s_logits,
                           long long* s_cls, long long* s_match, void* stream);
int lvc_fast_rcnn_losses(const float* logits, int ld_cls, const float* deltas, int ld_delta, int K, int cls_agnostic,
                         const float* proposals, const float* gt_boxes, const long long* gt_classes, int R, float wx,
                         float wy, float ww, float wh, float smooth_l1_beta, float* out_losses, float* dlogits,
                         float* ddeltas, double* row_terms /* [2R] scratch */, void* stream);
int lvc_rpn_losses(const float* logits, const float* deltas, const float* anchors, const float* gt_boxes,
                   const signed char* labels, int S, float smooth_l1_beta, float normalizer, float* out_losses,
                   void* stream);
/* lvc_rpn_losses plus the gradients the RPN head trains on (faster_rcnn_R_50_FPN_base.yaml, ft_all yaml):
 * dlogits [S] = (sigmoid(x) - label) / normalizer, ddeltas [S,4] = smooth-L1' / normalizer on positive rows, else 0. */
int lvc_rpn_losses_grad(const float* logits, const float* deltas, const float* anchors, const float* gt_boxes,
                        const signed char* labels, int S, float smooth_l1_beta, float normalizer, float* out_losses,
                        float* dlogits, float* ddeltas, void* stream);

/* Box-corrector training (BASELINE config 5; the shipped fine-tune yaml freezes the backbone).
 * lvc_giou_box_loss: BoxOnlyLayersCascade.box_reg_loss / BoxOnlyLayers.box_reg_loss
 *   (lvc/modeling/roi_heads/roi_heads_cascade.py:165-195): apply_deltas (box_regression.py:73-110, weights wx..wh,
 *   clamp) on the foreground rows (gt_classes in [0,K)) -> GIoU loss (fvcore.nn.giou_loss, eps 1e-7) -> mean; with
 *   iterate != 0: mean(max(loss_after - lambda * loss_before, 0)).  out_loss [1] (NaN when no row is foreground, as the
 *   reference's mean of an empty tensor), ddeltas [R,4] = d(loss)/d(deltas).
 * lvc_relu_backward: out = y > 0 ? dy : 0 (backward of the fused Linear+ReLU of FastRCNNConvFCHead, box_head.py:82-91).
 * lvc_colsum: out[n] = sum_m x[m*ldx + n], rows added in order (bias gradients). */
int lvc_giou_box_loss(const float* deltas, int ld_delta, const float* proposals, const float* gt_boxes,
                      const long long* gt_classes, int R, int K, float wx, float wy, float ww, float wh,
                      float scale_clamp, int iterate, float lambda, float* out_loss, float* ddeltas, void* stream);
int lvc_relu_backward(const float* dy, const float* y, long long n, float* out, void* stream);
int lvc_colsum(const float* x, int M, int N, int ldx, float* out, void* stream);

/* Backward into the trunk (BASELINE config 5 with `cascade_ubbr_R_50_FPN_base.yaml`: `BACKBONE.FREEZE_AT 2`, and the
 * base / ft_all detector yamls): the reference trains through ATen's conv2d backward for BottleneckBlock
 * (detectron2/modeling/backbone/resnet.py:195-211), FPN (backbone/fpn.py:109-144) and StandardRPNHead
 * (proposal_generator/rpn.py:120-139).
 * lvc_conv_wgrad_nhwc: dw[k][r][s][c] = scale[k] * sum_{n,oy,ox} dy[n,oy,ox,k] * x[n, oy*stride+r-pad, ox*stride+s-pad, c]
 *   (x zero outside the map).  x [N,H,W,C], dy [N,Ho,Wo,*] with row pitch lddy floats, scale [K] or NULL (the
 *   FrozenBatchNorm2d scale that follows the conv, batch_norm.py:45-65), dw [K,R,S,C] zeroed by the call.  Exact fp32
 *   MFMA; partial sums over pixel ranges are combined with fp32 atomics (not run-to-run deterministic).
 *   The data gradient needs no entry point of its own: it is lvc_conv2d_nhwc_* / lvc_conv3x3_nhwc_bf16x3 on dy with the
 *   weights flipped and transposed (host: kernels.pack_conv_dgrad), followed for a stride-2 1x1 by lvc_scatter_stride2_nhwc.
 * lvc_scatter_stride2_nhwc: y[n,2i,2j,:] = x[n,i,j,:], 0 elsewhere; x [N,(H-1)/2+1,(W-1)/2+1,C] -> y [N,H,W,C]
 *   (input gradient of a stride-2 1x1 conv and of LastLevelMaxPool, fpn.py:165-177).
 * lvc_downsum2x2_nhwc: y[n,i,j,:] = sum of x[n,2i..2i+1,2j..2j+1,:]; x [N,2Hs,2Ws,C] (backward of the nearest x2
 *   upsample of the FPN top-down path, fpn.py:131-133).
 * lvc_colsum_atomic: lvc_colsum for 10^5-row operands (conv bias gradients): row slabs combined with fp32 atomics. */
/* Device-side packing of the reference's OIHW parameters (detectron2/layers/wrappers.py:41-99 keeps `weight` as
 * [K,C,R,S]) into the conv kernels' operand: wp [rows_pad][R*S*cin_pad] fp32, k = (c/32, r, s, c%32).
 *   mode 0 (forward): rows = K output channels, cin = C.   mode 1 (data gradient): rows = C, contraction over the K
 *   output channels, taps flipped, times scale[k] (FrozenBatchNorm2d scale or NULL).  Padding rows/channels are zero.
 * lvc_split_weights: planes == 3 -> out = 3 x n bf16 (hi, mid, lo), planes == 2 -> 2 x n fp16 (w1, (w - w1) * 2048);
 *   |w| > 65504 in the fp16 split raises bit 1 of *err_word (the conv error word of lvc_conv_workspace). */
int lvc_pack_conv_weights(const float* w, const float* scale, float* wp, int K, int C, int R, int S, int rows_pad,
                          int cin_pad, int mode, void* stream);
int lvc_split_weights(const float* wp, long long n, int planes, void* out, int* err_word, void* stream);
/* lvc_pack_conv_weights and lvc_split_weights in one launch: planes_out [planes][rows_pad][R*S*cin_pad], planes = 2
 * (fp16) or 3 (bf16); err_word as lvc_split_weights. */
int lvc_pack_split_conv_weights(const float* w, const float* scale, float* wp, void* planes_out, int planes, int* err_word,
                                int K, int C, int R, int S, int rows_pad, int cin_pad, int mode, void* stream);
/* lvc_pack_split_conv_weights (+ lvc_split_weights_rowscaled) for `njobs` layers in one launch per 24 jobs (round 5: an optimizer
 * step invalidates the packed operands of every trainable layer -- the reference has no such step, ATen reads OIHW directly).
 * ptrs: host array, 6 device pointers per job = w (OIHW), scale (mode 1: folded into the operand, or NULL), fac_scale (fmt 4:
 * multiplied into the row factors, or NULL), wp ([rows_pad][R*S*cin_pad] fp32), planes, fac ([rows_pad] fp32, fmt 4);
 * shapes: 8 ints per job = K, C, R, S, rows_pad, cin_pad, mode (as lvc_pack_conv_weights), fmt -- 0: wp only, 2: fp16 planes
 * (lvc_split_weights), 3: bf16 planes, 4: row-scaled fp16 planes + fac[row] = 2^-e / 16 (* fac_scale[row]) (lvc_split_weights_rowscaled). */
int lvc_pack_group(int njobs, const void* const* ptrs, const int* shapes, int* err_word, void* stream);
int lvc_conv_wgrad_nhwc(const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W, int C,
                        int K, int R, int S, int stride, int pad, int lddy, void* stream);
/* lvc_conv_wgrad_nhwc on the three-way bf16 split MFMA path (six bf16 MFMAs per fp32-accurate product, one fp32
 * accumulator; gfx950 LDS transpose reads for the pixel-major operands).  No range restriction: the default. */
int lvc_conv_wgrad_nhwc_bf16x3(const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W, int C,
                               int K, int R, int S, int stride, int pad, int lddy, void* stream);
/* lvc_conv_wgrad_nhwc on the two-way fp16 split MFMA path (gfx950 LDS transpose reads for the pixel-major operands).
 * dy and x must lie inside fp16's range -- gradients scaled by a power of two (lvc_amd.solver.LossScaler); a value beyond
 * 65504 raises bit 1 (value 2) of *err_word (the conv error word of lvc_conv_workspace; may be NULL). */
int lvc_conv_wgrad_nhwc_f16x2(const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W, int C,
                              int K, int R, int S, int stride, int pad, int lddy, int* err_word, void* stream);
/* The weight gradients of `njobs` layers in one launch of the bf16x3 kernel (round 5): ATen's conv2d backward runs one wgrad per
 * layer (resnet.py:195-211, fpn.py:109-144 under autograd); at 2 images per GPU (cascade_ubbr_R_101_FPN_base.yaml) a res4 layer's
 * 16-36 tiles cannot fill the chip, and the weight gradient is off the backward's critical path, so the host queues the (x, dy)
 * pairs (lvc_amd.kernels.defer_wgrad) and launches them together.  x / dy / scale / dw: host arrays of njobs device pointers
 * (scale[j] may be NULL); shapes: host array, 10 ints per job = N, H, W, C, K, R, S, stride, pad, lddy (as lvc_conv_wgrad_nhwc).
 * dw[j] ([K][R][S][C]) must be ZERO on entry: partial sums over pixel slices are added with fp32 atomics. */
int lvc_conv_wgrad_group_bf16x3(int njobs, const float* const* x, const float* const* dy, const float* const* scale,
                                float* const* dw, const int* shapes, void* stream);
/* dst[j] (the parameter's OIHW layout [K][C][R*S]) = (beta ? dst[j] : 0) + src[j] ([K][R*S][C], what the wgrad kernels write), all
 * jobs in one launch; shapes: 4 ints per job = K, C, R*S, beta.  (AccumulateGrad's `grad += dw` / first assignment.) */
int lvc_wgrad_finalize_group(int njobs, const float* const* src, float* const* dst, const int* shapes, void* stream);
int lvc_scatter_stride2_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* y[n, i, j, 0..C) = x[n, 2i, 2j, :], rows of ldy floats in y (0 = C): the sampling of a stride-2 1x1 convolution as a copy (the
 * block input of res3.0 / res4.0 / res5.0 next to conv2's output: conv3 + projection shortcut as one GEMM, resnet.py:117-160). */
int lvc_subsample2_nhwc(const float* x, float* y, int N, int H, int W, int C, int ldy, void* stream);
int lvc_downsum2x2_nhwc(const float* x, float* y, int N, int Hs, int Ws, int C, void* stream);
int lvc_colsum_atomic(const float* x, int M, int N, int ldx, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Label-verification kNN (tools/run_nearest_neighbours.py:142-162, 214-227).
 * lvc_colmean: mu[d] = mean_m x[m,d], summed in a fixed order (run-to-run deterministic).  lvc_knn_topk_vote: per query row, class ids of the 10 most similar
 * shots (ties -> lower shot index) and keep = (mode of the first kvote ids, ties -> smallest id, == detector class).
 *   sims [Q,ld] fp32 (S columns used, S <= 4096), shot_classes [S] int64, det_classes [Q] int64 or NULL,
 *   top_classes [Q,10] int64, keep [Q] int64 or NULL.
 */
int lvc_colmean(const float* x, float* mu, int M, int D, int ld, void* stream);
/* Shot sets of any size (tools/run_nearest_neighbours.py:146-160 ranks any S): lvc_knn_topk_candidates writes, for ONE block of
 * S <= 4096 columns of sims [Q,ld], the ten best (similarity, idx_base + column) pairs per row (cand_val / cand_idx [Q,10];
 * value descending, ties -> lower index, padded with (-inf, INT_MAX)); lvc_knn_merge_vote ranks nlists <= 64 such lists
 * ([nlists][Q][10]) against each other, gathers the classes and votes: outputs as lvc_knn_topk_vote.
 * lvc_max_f32: out[0] = max of x[0..n). */
int lvc_knn_topk_candidates(const float* sims, int ld, int Q, int S, int idx_base, float* cand_val, int* cand_idx, void* stream);
int lvc_knn_merge_vote(const float* cand_val, const int* cand_idx, int nlists, int Q, const long long* shot_classes,
                       const long long* det_classes, int kvote, long long* top_classes, long long* keep, void* stream);
int lvc_max_f32(const float* x, long long n, float* out, void* stream);
int lvc_knn_topk_vote(const float* sims, int ld, int Q, int S, const long long* shot_classes,
                      const long long* det_classes, int kvote, long long* top_classes, long long* keep,
                      void* stream);
/* Two-stage form of the same sweep (cosine branch; the similarities of run_nearest_neighbours.py:146-151 are never
 * materialised in full precision).
 * lvc_rownorm_h: lvc_rownorm that writes the rows rounded to fp16 (yh [M,D]), the denominators (den [M] or NULL), the
 *   2-norm of each row's rounding residual row - fp16(row) (resid [M] or NULL) and, if y is not NULL, the fp32 rows
 *   (contiguous, bit-identical to lvc_rownorm).
 * lvc_gemm_f16: y [M,ldy] fp32 = a [M,C] . b [N,ldb]^T on fp16 operands (csrc/gemm_h.hip; C % 32 == 0, ldb = elements
 *   between rows of b, 0 = C; y below 2 GiB) -- for unit-norm rows |y - exact| < 2^-10.
 * lvc_knn_verify_topk_vote: per query row, the shots whose approximate similarity is within `margin` (>= 2 x that bound)
 *   of the 10th largest approximate value provably contain the exact ten best; those of them that have a shot of another
 *   class within margin are re-evaluated in fp32 from q [Q,ldq] (raw descriptors; (q - mu) / den[row] is redone exactly as
 *   lvc_rownorm_h did it; mu / den may be NULL: q then already holds the rows) and sn [S,D] (normalised shots), and the
 *   candidates are ranked so that the class sequence equals the exact ranking's (ties -> lower shot index; csrc/knn.hip
 *   states the argument).  margins [Q] (or NULL): a margin per row that replaces `margin` -- 2 x the row's own error bound
 *   |resid_q| max|s_h| + |q| max resid_s + D 2^-24 (Cauchy-Schwarz on the two rounding residuals + fp32 accumulation), about
 *   half the worst case.  D % 4 == 0, D <= 2048.  Outputs as lvc_knn_topk_vote. */
int lvc_rownorm_h(const float* x, const float* mu, float* y, unsigned short* yh, float* den, float* resid, int M, int D, int ldx,
                  float eps, int mode, void* stream);
int lvc_gemm_f16(const unsigned short* a, const unsigned short* b, int ldb, float* y, int M, int N, int C, int ldy,
                 void* stream);
int lvc_knn_verify_topk_vote(const float* approx, int ld, int Q, int S, const float* q, int ldq, const float* mu,
                             const float* den, const float* sn, int D, float margin, const float* margins,
                             const long long* shot_classes, const long long* det_classes, int kvote, long long* top_classes,
                             long long* keep, void* stream);
/* 16-bit fixed-point form of the two-stage sweep's similarity matrix (tools/run_nearest_neighbours.py:146-151 keeps it in fp32; here
 * it only pre-filters): lvc_gemm_f16_q15 writes y[m][n] = rint(32766 * dot) clamped to +-32766 as int16 (NaN -> 32767; ldy even), and
 * lvc_knn_verify_topk_vote_q15 reads it (ld % 8 == 0, rows 16-byte aligned) -- half the bytes of the matrix round trip; the margin
 * handed in must include 2 x 1.6e-5 for the quantisation. */
/* The sweep's per-row margins in one launch: margins[i] = 2 (1 + 1e-4) (qres[i] (1 + 1e-6 + *d_sres_max) + *d_sres_max (1 + 1e-5) + acc) + extra
 * (qres: the queries' rounding-residual norms from lvc_rownorm_h, d_sres_max: device scalar = lvc_max_f32 over the shots' residual
 * norms, acc: the accumulation term, extra: the 16-bit matrix's quantisation margin or 0). */
int lvc_knn_margins(const float* qres, const float* d_sres_max, float acc, float extra, int Q, float* margins, void* stream);
int lvc_gemm_f16_q15(const unsigned short* a, const unsigned short* b, int ldb, short* y, int M, int N, int C, int ldy, void* stream);
int lvc_knn_verify_topk_vote_q15(const short* approx, int ld, int Q, int S, const float* q, int ldq, const float* mu,
                             const float* den, const float* sn, int D, float margin, const float* margins,
                             const long long* shot_classes, const long long* det_classes, int kvote, long long* top_classes,
                             long long* keep, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Descriptor network of the label-verification step: DINO ViT-S/8 as loaded by
 * tools/run_nearest_neighbours.py:292-293 (torch.hub 'facebookresearch/dino', a third-party model: published
 * architecture, restated for the tests in oracle/vit.py).  The linear layers run on the conv/GEMM entry points
 * above; these are the remaining pieces.  All tensors fp32, row-major.
 *   lvc_vit_patchify : img [B,C,H,W] -> [B*(H/ps)*(W/ps), C*ps*ps], column = c*ps*ps + r*ps + s (so that the ps x ps
 *                      stride-ps convolution of patch_embed is one GEMM with proj.weight.reshape(D, C*ps*ps))
 *   lvc_vit_tokens   : out[b][0] = cls + pos[0], out[b][1+p] = emb[b*P+p] + pos[1+p]   (out [B*(P+1), D], D % 4 == 0)
 *   lvc_layernorm    : torch.nn.LayerNorm over rows of D <= 2048 elements (biased variance, eps inside the sqrt)
 *   lvc_gelu         : torch.nn.GELU() exact (erf) form, n % 4 == 0
 *   lvc_mha          : qkv [B*N, 3*H*64] (column = which*H*64 + h*64 + d) -> out [B*N, H*64] =
 *                      softmax(q k^T * scale) v per (image, head); head_dim must be 64 (one thread per query, fp32 VALU)
 *   lvc_mha_mfma     : the same result on the matrix cores (csrc/vit.hip: both products as fp32-accurate two-way fp16 splits, the
 *                      probabilities stay in registers between them); workspace = lvc_mha_workspace_bytes(B, N, H) bytes,
 *                      16-byte aligned (fp16 operand planes of q, k, v); d_error_word (may be NULL): device int whose bit 1 is
 *                      set when q * scale * log2(e), k or v leaves fp16's range (|x| > 65504) or is NaN -- the result is
 *                      then invalid and the caller falls back to lvc_mha */
int lvc_vit_patchify(const float* img, float* out, int B, int C, int H, int W, int ps, void* stream);
/* lvc_vit_patchify of (img - mean[c]) / std[c] (tools/run_nearest_neighbours.py:95-99 preprocess_crops fused into the gather);
 * mean / std: HOST arrays of C <= 8 floats. */
int lvc_vit_patchify_norm(const float* img, const float* mean, const float* std, float* out, int B, int C, int H, int W, int ps,
                          void* stream);
int lvc_vit_tokens(const float* emb, const float* cls, const float* pos, float* out, int B, int P, int D, void* stream);
int lvc_layernorm(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int D, float eps,
                  void* stream);
int lvc_gelu(const float* x, float* y, long long n, void* stream);
int lvc_mha(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale, void* stream);
/* The attention output of token 0 (the class token) of every image only: qkv [B*N, 3*H*64] -> out [B, H*64] (N <= 1024).  The last block
 * of the descriptor network: its output is read at the class rows (tools/run_nearest_neighbours.py:102-128). */
int lvc_mha_cls(const float* qkv, float* out, int B, int N, int H, float scale, void* stream);
long long lvc_mha_workspace_bytes(int B, int N, int H);
int lvc_mha_mfma(const float* qkv, float* out, void* workspace, int B, int N, int H, float scale, int* d_error_word, void* stream);
/* The qkv layer of a ViT block written straight into lvc_mha_mfma's operand planes (round 5; the DINO ViT's `attention.qkv` + the first
 * pass of the attention, tools/run_nearest_neighbours.py:102-128): lvc_conv1x1_nhwc_f16s1 on x [B*N][C] (w_split / scale / shift as there,
 * K = 3 * H * 64 columns = (q | k | v, head, d)); planes [6][B*H][Npad][64] fp16 (lvc_mha_workspace_bytes), q multiplied by softmax_scale *
 * log2(e) (as lvc_mha_mfma does); rows N..Npad-1 are not written (the caller keeps them zero).  Bit-identical to the two launches it replaces.
 * An operand beyond fp16's range raises bit 1 (value 2) of *err_word.  lvc_mha_mfma_planes: the attention on such planes. */
int lvc_conv1x1_qkv_planes_f16s1(const float* x, const unsigned short* w_split, const float* scale, const float* shift, void* planes,
                                 int B, int N, int C, int H, float softmax_scale, int* err_word, void* workspace, void* stream);
int lvc_mha_mfma_planes(const void* planes, float* out, int B, int N, int H, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LVC_AMD_H */
