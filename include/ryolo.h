/*
 * ryolo.h — C ABI of libryolo_hip.so (MI355X / gfx950 hot path of yingkunwu/R-YOLOv4).
 *
 * The reference has no C ABI of its own: its boundary is three Python call surfaces plus one torch-op schema
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces (paths in /root/reference).
 *
 * Conventions (all entry points):
 *   - plain C, raw DEVICE pointers + explicit sizes, a hipStream_t; no torch types;
 *   - returns 0 on success, RY_ERR_* otherwise; never throws, never allocates, never synchronises:
 *     all outputs and workspaces are caller-allocated (query *_workspace_bytes first), so every call is
 *     hipGraph-capturable;
 *   - re-entrant across streams; the library keeps no mutable global state.
 */
#ifndef RYOLO_H
#define RYOLO_H
#include <stddef.h>
#include <stdint.h>
#include "ryolo_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ryolo_stream_t;   /* == hipStream_t */

#define RYOLO_OK 0
#define RYOLO_ERR_ARG 1
#define RYOLO_ERR_WORKSPACE 2
#define RYOLO_ERR_LAUNCH 3
#define RYOLO_ERR_UNSUPPORTED 4

/* ------------------------------------------------------------------------------------------------------------
 * Rotated NMS / SkewIoU — replaces torch.ops.detectron2.nms_rotated / box_iou_rotated
 * (detectron2 is un-vendored; call sites lib/general.py:177, test.py:135, dead code lib/loss.py:241).
 * Boxes are (xc, yc, w, h, angle_degrees), float32.
 * ------------------------------------------------------------------------------------------------------------ */
int ryolo_nms_workspace_bytes(int batch, int64_t nmax, size_t* bytes);

/* boxes [batch, nmax, 5] already sorted by score descending (lib/general.py:166-168);
 * counts [batch] (device, may be NULL = all nmax valid); gt_only=1: suppress iff IoU > thr (detectron2 CUDA kernel),
 * 0: IoU >= thr (detectron2 CPU kernel); keep [batch, keep_stride] receives POSITIONS in the sorted order, ascending;
 * at most max_keep (<=0: no cap) are produced (lib/general.py:178-179 keeps max_det=1500); num_keep [batch] (device). */
int ryolo_nms_rotated_batched(const float* boxes, const int32_t* counts, int batch, int64_t nmax, float iou_thr,
                              int gt_only, int64_t max_keep, void* workspace, size_t workspace_bytes, int64_t* keep,
                              int64_t keep_stride, int32_t* num_keep, ryolo_stream_t stream);

/* IoU[n, m] row-major = pairwise_iou_rotated(b1[n,5], b2[m,5]) (test.py:135).
 * workspace >= align256(n*48) + m*48 bytes. */
int ryolo_box_iou_rotated(const float* b1, int n, const float* b2, int m, void* workspace, size_t workspace_bytes,
                          float* out, ryolo_stream_t stream);

/* out[i] = IoU(b1[i], b2[i]) — the element-wise SkewIoU score of the commented-out block lib/loss.py:233-245. */
int ryolo_diag_iou_rotated(const float* b1, const float* b2, int n, float* out, ryolo_stream_t stream);

/* mAP evaluation (test.py:102-149 `get_batch_statistics`): true-positive matrix for a whole batch in one launch.
 * preds [npred,7] = (x, y, w, h, theta_rad, score, cls), images concatenated, each image score-descending (post_process order);
 * pred_off / tgt_off [batch+1] row offsets (int64, device); targets [ntgt,7] = (img, cls, x, y, w, h, theta_rad) grouped by image in
 * the caller's order; iouv [niou] ascending thresholds (device); tp [npred,niou] bytes out.  Class ids must be integers in
 * [0, num_classes), num_classes <= 256.  Side effect of the reference kept: theta of preds becomes degrees in place for images
 * that have both predictions and labels (test.py:126). */
int ryolo_map_match_workspace_bytes(int64_t npred, int64_t ntgt, size_t* bytes);
int ryolo_map_match(float* preds, const int64_t* pred_off, const float* targets, const int64_t* tgt_off, int batch, int64_t npred,
                    int64_t ntgt, const float* iouv, int niou, int num_classes, unsigned char* tp, void* workspace, size_t workspace_bytes,
                    ryolo_stream_t stream);

/* AP from the concatenated statistics (test.py:16-99 `ap_per_class` + `compute_ap`), on the device: confidence sort (score desc, index
 * asc), per-class cumulative hits, recall = hits / (n_labels + 1e-16), precision = hits / rank, precision envelope, trapezoid over the
 * 101-point recall grid for every IoU threshold, and precision / recall at the first threshold interpolated over `conf_grid` (the
 * reference's 1000-point px) — numpy's float64 expressions restated (np.interp, pairwise-summed np.trapz).
 * tp [n][niou] bytes, conf / pred_cls [n] fp32, target_cls [nl] fp32 (class ids: integers in [0, nc), nc <= 256), recall_grid [101] and
 * conf_grid [nconf] float64 on the device (np.linspace values from the host).  Out, all classes 0 .. nc-1 (the caller keeps those with
 * n_labels > 0, like np.unique(target_cls)): ap [nc][niou], prec_at / rec_at [nc][nconf] (zero rows for classes without labels or
 * predictions), n_labels [nc], n_pred [nc].  No host synchronisation. */
int ryolo_ap_workspace_bytes(int64_t n, int niou, int nc, size_t* bytes);
int ryolo_ap_per_class(const unsigned char* tp, const float* conf, const float* pred_cls, int64_t n, const float* target_cls, int64_t nl,
                       int nc, int niou, const double* recall_grid, const double* conf_grid, int nconf, void* workspace, size_t workspace_bytes,
                       double* ap, double* prec_at, double* rec_at, int64_t* n_labels, int64_t* n_pred, ryolo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * YoloLayer — replaces model/yololayer.py:15-56 (YoloCSLLayer.forward) and :66-105 (YoloKFIoULayer.forward).
 * ------------------------------------------------------------------------------------------------------------ */
/* in [batch, na*attrs, gs, gs] (NCHW conv output) -> out [batch, na, gs, gs, attrs]
 * == out[i].view(bs,na,attrs,gs,gs).permute(0,1,3,4,2).contiguous()  (model/yololayer.py:25, :76).
 * (The conv stack of this library emits that layout directly from the head-conv epilogue.) */
int ryolo_head_permute(const float* in, float* out, int batch, int na, int attrs, int gs, ryolo_stream_t stream);

/* Eval-time decode of ONE scale into its row slice of infer_out [batch, rows_per_image, nc+6]:
 * mode 0 = csl  (attrs = nc+185: x,y,w,h,obj,cls[nc],theta[180]; first-max argmax, model/yololayer.py:28-54),
 * mode 1 = kfiou(attrs = nc+6  : x,y,w,h,a,obj,cls[nc]; angle scale 0.5236, no norm_angle, :79-103).
 * head [batch, na, gs, gs, attrs]; anchors_host = HOST pointer to na*3 floats (w, h, angle_rad; csl ignores angle),
 * grid units (model/yolo.py:54-72); rows of this scale start at row_offset (scale order 8 -> 16 -> 32). */
int ryolo_decode(int mode, const float* head, float* infer_out, int batch, int na, int gs, int nc, float stride,
                 const float* anchors_host, int64_t row_offset, int64_t rows_per_image, ryolo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * post_process stages — replace the per-image loop of lib/general.py:153-181.
 * ------------------------------------------------------------------------------------------------------------ */
/* pred [batch, M, nc+6] is MUTATED: cls *= obj (lib/general.py:155).  key[b,i] = max_k cls (first max) if it is
 * > conf_thres (strict, :161) else -inf; cls[b,i] = argmax as float; count[b] = number of passing candidates. */
int ryolo_pp_score(float* pred, int batch, int64_t M, int nc, float conf_thres, float* key, float* cls,
                   int32_t* count, ryolo_stream_t stream);

/* Score ordering on the device (csrc/topk.hip) — the `argsort(descending=True)[:max_nms]` of lib/general.py:166-168 and the score
 * sort inside detectron2's nms_rotated, ties broken by ascending index.  Workspace for both: ryolo_sort_workspace_bytes
 * (rows, n) with n = K resp. N.
 * ryolo_topk_desc: per row of key [batch, M] the K <= 16384 largest entries in (key desc, index asc) order -> skey [batch, K]
 * (-inf padded), order [batch, K] (-1 padded), nsel [batch] = entries selected (null ok); key == -inf is never selected.
 * ryolo_argsort_desc: order [N] = stable descending argsort of scores [N]. */
int ryolo_sort_workspace_bytes(int rows, int64_t n_sorted, size_t* bytes);
int ryolo_topk_desc(const float* key, int batch, int64_t M, int K, float* skey, int64_t* order, int32_t* nsel, void* workspace,
                    size_t workspace_bytes, ryolo_stream_t stream);
int ryolo_argsort_desc(const float* scores, int64_t N, int64_t* order, void* workspace, size_t workspace_bytes, ryolo_stream_t stream);

/* sorted_key/order (row stride `stride` >= K) = the top-K of key along M in (key desc, index asc) order.  Writes dets [batch,K,7] =
 * (x,y,w,h,theta_rad,score,cls) and rboxes [batch,K,5] = NMS boxes with class offset cls*max_wh on x,y and theta in
 * degrees (lib/general.py:166-174); count[b] is clamped to K. */
int ryolo_pp_gather(const float* pred, const float* sorted_key, const int64_t* order, const float* cls, int batch,
                    int64_t M, int nc, int64_t K, int64_t stride, float max_wh, float* dets, float* rboxes, int32_t* count,
                    ryolo_stream_t stream);

/* out [batch, keep_stride, 7]: out[b,j] = dets[b, keep[b,j]] for j < num_keep[b], zeros after (lib/general.py:181). */
int ryolo_pp_emit(const float* dets, const int64_t* keep, const int32_t* num_keep, int batch, int64_t K,
                  int64_t keep_stride, float* out, ryolo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Conv stack — replaces every nn.Conv2d / nn.BatchNorm2d / activation / MaxPool2d / Upsample / torch.cat of
 * model/utils.py:6-282, model/backbone.py, model/neck.py (cuDNN/ATen under the reference).  Activations are NHWC
 * bf16 with an explicit channel stride; parameter blocks are defined in ryolo_params.h.
 * ------------------------------------------------------------------------------------------------------------ */
/* implicit-GEMM convolution: forward (taps = kernel window), data gradient (mirrored taps, stride-2 as 4 parity
 * classes on grid.z).  Epilogues (p->epi): 0 raw bf16, 1 raw + per-tile BatchNorm partial sums, 2 folded-BN + activation,
 * 3 fp32 + bias (detection heads), 4 bf16 accumulate (tensor with several consumers). */
int ryolo_conv_gemm(const ConvGemmParams* p, ryolo_stream_t stream);
/* UPPER BOUND of the [2][Nout] partial-statistics rows epilogue 1 writes for an M x Nout problem (sizing only: the exact count, which
 * the reduction must use, depends on the tile ryolo_conv_gemm picks and is ryolo_conv_gemm_plan's stats_rows) */
int ryolo_conv_gemm_stats_rows(int64_t M, int Nout, int pipe, int* rows);
/* which kernel ryolo_conv_gemm runs for *p and the number of partial-statistics rows its epilogue 1 writes; `kernel` may be null.
 * *kernel & 0xff: 0 generic implicit GEMM, 1 the 3x3 stride-1 halo-patch kernel (pipe bit 0x200, eligible layers), 2 the weight-stationary
 * persistent 1x1 kernel (RYOLO_GEMM_WS and its tapped / pool-gradient instantiations), 3 the persistent weight-stationary 3x3 kernel for 64 -> <= 64
 * channels (conv3x3_ws.hip), 4 the 256-wide 8-wave pointwise kernel (gemm256.hip; bits 16-19 = tile columns / 32); for 0 also bit 0x100 = the 1x1
 * instantiation, bits 12-15 = tile rows / 64, bits 16-19 = tile columns / 32 */
int ryolo_conv_gemm_plan(const ConvGemmParams* p, int* stats_rows, int* kernel);
/* weight gradient: split-K over output pixels into p->partial ([splitk][Cout][taps*Cin] fp32, size from _plan), then a
 * deterministic reduction that accumulates into the torch-layout .grad [Cout][Cin][kh*kw] (no float atomics). */
int ryolo_conv_wgrad_plan(const WgradParams* p, int* splitk, size_t* workspace_bytes);
int ryolo_conv_wgrad(const WgradParams* p, ryolo_stream_t stream);
/* which kernel ryolo_conv_wgrad launches for *p: 0 generic split-K (register-staged, or the LDS-DMA pointwise form), 1 the 3x3 stride-1 halo-ring
 * kernel (needs p->zeros), 2 the tapped LDS-DMA kernel (tapped / strided layers with > 64 output channels; needs p->zeros), 3 the 8-wave
 * 256 x 256-tile pointwise kernel (stride-1 1x1 layers with Cin >= 256 and Cout > 128; needs p->zeros), 4 the 8-wave parity-plane ring kernel
 * (3x3 stride-2 pad-1 layers with more than 64 output channels; needs p->zeros) */
int ryolo_conv_wgrad_kernel(const WgradParams* p, int* kernel);
/* launch shape of that kernel: workgroups, waves per workgroup (8: a workgroup holds its CU exclusively and the grid is sized to part of the chip) */
int ryolo_conv_wgrad_grid(const WgradParams* p, int* workgroups, int* waves);
/* weights of a stride-2 3x3 (pad 1) data gradient in its space-to-depth form (ConvGemmParams.s2d_cin): w fp32 [Cout][Cin][3][3] ->
 * out bf16 [4 * Cin][4][round_up(Cout, 32)] */
int ryolo_pack_s2d(const float* w, int Cout, int Cin, bf16_t* out, ryolo_stream_t stream);

/* first layer, 3x3 stride 1 pad 1 on the fp32 NCHW image (Cin = 3, Cout <= 32), without the im2col round trip (csrc/stem.hip);
 * _plan: partial-statistics rows of epilogue 1 and the weight-gradient workspace; _wgrad needs Cout == 32 and W % 16 == 0 and
 * leaves dW in the [Cout][32] GEMM layout in p->scratch (ryolo_unpack_wgrad adds it to the torch-layout .grad). */
int ryolo_stem3x3_plan(int NB, int H, int W, int Cout, int* stats_rows, size_t* wgrad_workspace_bytes);
int ryolo_stem3x3_fwd(const StemParams* p, ryolo_stream_t stream);
int ryolo_stem3x3_wgrad(const StemWgradParams* p, ryolo_stream_t stream);
/* the whole backward of that layer (BatchNorm + activation backward, BatchNorm parameter gradients, weight gradient) in one pass over
 * dz with the conv output recomputed from the image: Cout == 32, W % 32 == 0 (ryolo_stem3x3_bwd_plan returns UNSUPPORTED otherwise) */
int ryolo_stem3x3_bwd_plan(int NB, int H, int W, int Cout, size_t* workspace_bytes);
int ryolo_stem3x3_bwd(const StemBwdParams* p, ryolo_stream_t stream);

/* training BatchNorm2d (eps, momentum of nn.BatchNorm2d; model/utils.py:17): partial [rows][2][C] (the buffer must have
 * room for 64 more rows: fold scratch for big layers) -> coeffs [4][C] =
 * mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; running_mean/var updated in place (unbiased var). */
int ryolo_bn_finalize(const float* partial, int rows, int C, double count, float eps, float momentum, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float* coeffs, ryolo_stream_t stream);
/* the same for channels [c0, c0 + C) of partial rows that are [2][ld] wide: sibling convolutions of a block that read the same input
 * (ELAN / CSP / C3 / SPPCSPC cv1 + cv2, model/utils.py:49-143,264-282) are emitted as ONE GEMM with concatenated output channels;
 * each BatchNorm finalizes its own slice into its own coeffs [4][C] */
int ryolo_bn_finalize_slice(const float* partial, int rows, int ld, int c0, int C, double count, float eps, float momentum,
                            const float* gamma, const float* beta, float* running_mean, float* running_var, float* coeffs,
                            ryolo_stream_t stream);
/* eval BatchNorm2d: coeffs from the running statistics */
int ryolo_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                         int C, float* coeffs, ryolo_stream_t stream);
/* ... into channels [c0, c0 + C) of a shared coeffs [4][ld] (folded-BN epilogue of a shared GEMM launch) */
int ryolo_bn_eval_coeffs_slice(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                               int C, float* coeffs, int ld, int c0, ryolo_stream_t stream);
/* z = act(bn1(y1) [+ bn2(y2)]) [+ residual]   (Conv / RepConv / Bottleneck of model/utils.py) */
int ryolo_bn_act_fwd(const BnActParams* p, ryolo_stream_t stream);
int ryolo_bn_act_bwd_blocks(int64_t M, int C, int* nblk, int* rows_per_block);
/* backward of the above: dy1 [, dy2] [, dres], dgamma/dbeta accumulated; p->partial needs (nblk+64)*K*C floats, bco 3*C.
 * frozen=1: the coefficients came from ryolo_bn_eval_coeffs (fixed affine map, no batch-statistics coupling).
 * p->dy1 == NULL (single branch, no residual): statistics only — bco [2][C] = mean g, mean g*xhat and dgamma/dbeta are produced and
 * the apply pass is left to the consumer (ryolo_stem3x3_wgrad with StemWgradParams.y set). */
int ryolo_bn_act_bwd(const BnActParams* p, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* bco,
                     int frozen, ryolo_stream_t stream);

int ryolo_maxpool_fwd(const PoolParams* p, ryolo_stream_t stream);      /* nn.MaxPool2d k2 s2 / k5,9,13 s1 (utils.py:152,231-233) */
int ryolo_maxpool_bwd(const PoolParams* p, ryolo_stream_t stream);
int ryolo_upsample2x_fwd(const UpParams* p, ryolo_stream_t stream);     /* nn.Upsample(scale_factor=2) nearest (neck.py) */
int ryolo_upsample2x_bwd(const UpParams* p, ryolo_stream_t stream);
/* first layer (Cin = 3): fp32 NCHW image -> bf16 [NB*OH*OW][Kpad] patches, k = (r*kw + s)*Cin + c */
int ryolo_im2col(const float* img, int NB, int Cin, int H, int W, int kh, int kw, int stride, int pad, int OH, int OW, int Kpad,
                 bf16_t* col, ryolo_stream_t stream);
/* detection head tail: pre [M][ldp] fp32 (conv + bias) [* ImplicitM] -> [B, na, gs, gs, attrs] (yololayer.py:25 fused) */
int ryolo_head_finish_fwd(const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs, float* out,
                          ryolo_stream_t stream);
/* backward of the head tail: dpre (bf16 GEMM operand), and ACCUMULATED into dbias (conv bias gradient = column sums of dpre, may be
 * null) and dmul (ImplicitM gradient, with mul).  scratch >= (B*ceil(gs*gs/128) + 64) * 2*na*attrs floats; dpre pad columns
 * (>= na*attrs) must be pre-zeroed by the caller */
int ryolo_head_finish_bwd(const float* dout, const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs,
                          bf16_t* dpre, int ldd, float* dbias, float* dmul, float* scratch, ryolo_stream_t stream);
/* the same over the sparse description of dout that ryolo_loss leaves (LossParams.objgrad, ryolo_loss_owner_grids): only the dense rows of matched cells
 * (owner[cell] >= 0) are read from dout, every other cell is zero except its objectness element objgrad[cell] at index och.  preobj (may be null) =
 * the compact objectness column of pre written by ryolo_head_finish_fwd_obj: with it, pre is read at matched cells only.  Bit-identical results.
 * (the reference has no counterpart: autograd walks the dense maps, lib/loss.py:282-331 -> model/yololayer.py:25) */
int ryolo_head_finish_bwd_sparse(const float* dout, const float* objgrad, const int* owner, int och, const float* preobj, const float* pre, int ldp,
                                 const float* mul, int B, int gs, int na, int attrs, bf16_t* dpre, int ldd, float* dbias, float* dmul,
                                 float* scratch, ryolo_stream_t stream);
/* ryolo_head_finish_fwd + preobj [B, na, gs, gs] fp32 = pre[.., a*attrs + och] (och: 4 csl, 5 kfiou) and, if xobj is given, the same column of
 * out (after ImplicitM) — LossParams.headobj */
int ryolo_head_finish_fwd_obj(const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs, float* out, int och, float* preobj,
                              float* xobj, ryolo_stream_t stream);
/* parameter gradients of a detection head whose ImplicitM is applied by the GEMM epilogue (ConvGemmParams.head_attrs), out = (W (x + a) + b) m:
 * G [Cout][K] = weight gradient of the UNSCALED head gradient against x, s [Cout] = its column sums, a = ImplicitA [K] or null;
 * Ge = G + s (x) a; dW += m Ge, db += m s, dm += rowdot(W, Ge) + b s, da += W^T (m s); G and s are cleared.
 * (model/neck.py:173-186 ImplicitA / ImplicitM; autograd's chain through the add and the multiply) */
int ryolo_head_wgrad_finish(float* G, float* s, const float* W, const float* b, const float* m, const float* a, int Cout, int K, float* dW, float* db,
                            float* dm, float* da, ryolo_stream_t stream);
/* out[c] = b[c] + sum_k W[c][k] a[k]: the bias of a head with ImplicitA folded in (W (x + a) + b = W x + out) */
int ryolo_head_bias_fold(const float* W, const float* b, const float* a, int Cout, int K, float* out, ryolo_stream_t stream);
int ryolo_chan_add(const bf16_t* x, int ldx, const float* a, int64_t M, int C, bf16_t* z, int ldz, ryolo_stream_t stream);  /* ImplicitA */
/* out[c] += sum_m x[m][c] for c < Cvalid; C = readable (padded, multiple of 8) width; scratch >= (ceil(M/256) + 64)*C floats */
int ryolo_colsum_bf16(const bf16_t* x, int ldx, int64_t M, int C, int Cvalid, float* out, float* scratch, ryolo_stream_t stream);

/* inference re-parameterisation of RepConv (model/utils.py:189-215 leaves the 3 branches un-fused): w3 fp32 [Cout][Cin][3][3],
 * w1 fp32 [Cout][Cin], coa / cob = ryolo_bn_eval_coeffs of the two BatchNorms -> wf bf16 [Cout][9][Cin] (GEMM image of the merged
 * 3x3 kernel) and co_out [4][Cout] (scale 1, shift = shift3 + shift1) for the EPI_AFFINE_ACT epilogue */
int ryolo_repconv_fold(const float* w3, const float* w1, const float* coa, const float* cob, int Cout, int Cin, bf16_t* wf, float* co_out,
                       ryolo_stream_t stream);
/* fp32 master weights (torch layout) -> bf16 GEMM images, all convolutions in one launch */
int ryolo_pack_weights(const PackEntry* table_dev, int n, int64_t total, ryolo_stream_t stream);
int ryolo_unpack_wgrad(const float* scratch, int Cout, int Cin, int taps, int CinP, float* grad, ryolo_stream_t stream);
/* torch.optim.SGD(momentum, nesterov=True) of train.py:156 over flat buffers: buf = mu*buf + g; p -= lr*(g + mu*buf) */
int ryolo_sgd_nesterov(float* p, float* g, float* buf, int64_t n, float lr, float mu, float gscale, int zero_grad,
                       ryolo_stream_t stream);   /* g is scaled by gscale on read; zero_grad=1 clears it (optimizer.zero_grad fused) */
/* torch.optim.Adam(lr) of train.py:153-154 (betas / eps from the caller, no weight decay, no amsgrad) over flat buffers, `step` = 1, 2, ...
 * (the bias corrections 1 - beta^step are formed in double on the host, as torch does): m = m + (1 - b1)(g - m); v = b2 v + (1 - b2) g^2;
 * p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).  gscale / zero_grad as in ryolo_sgd_nesterov. */
int ryolo_adam(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, int64_t step, float gscale,
               int zero_grad, ryolo_stream_t stream);
/* ---- image-side augmentations of the loader on uint8 HWC (BGR) images in HBM (csrc/augment.hip; datasets/base_dataset.py:224-330,
 * lib/augmentations.py:8-74).  paste: `rects` = device array of nrect records {int64 src_off; int src_w, sx, sy, dx, dy, w, h, canvas}
 * (ryolo_paste_rect_bytes = sizeof), canvases [ncanvas, CH, CW, 3] filled with `fill` first, rectangles applied in order (later wins).
 * warp: dst[b] = cv2.warpPerspective(src[b], M[b], (DW, DH), borderValue = border) with Minv[b] = inverse(M[b]) as 9 doubles.
 * hsv: BGR->HSV, lut [3][256] (hue, sat, val), HSV->BGR, in place.  mixup: out = uint8(a*r + b*(1-r)). */
int ryolo_paste_rects(const uint8_t* pool, const void* rects_dev, int nrect, uint8_t* canvas, int ncanvas, int CH, int CW, int fill,
                      ryolo_stream_t stream);
/* the same with the rectangles grouped by canvas (first_dev [ncanvas + 1] ints: canvas b owns rects[first[b] .. first[b + 1])): a canvas pixel looks at
 * its own canvas's <= 9 rectangles instead of the whole batch's table (what BaseDataset.assemble_batch uses: load_mosaic / load_mosaic9 of every sample) */
int ryolo_paste_rects_grouped(const uint8_t* pool, const void* rects_dev, int nrect, const int* first_dev, uint8_t* canvas, int ncanvas, int CH, int CW,
                              int fill, ryolo_stream_t stream);
int ryolo_paste_rect_bytes(int* bytes);
int ryolo_warp_perspective_u8(const uint8_t* src, int batch, int SH, int SW, const double* Minv, uint8_t* dst, int DH, int DW, int border,
                              ryolo_stream_t stream);
int ryolo_hsv_gain_u8(uint8_t* img, int64_t npix, const uint8_t* lut, ryolo_stream_t stream);
int ryolo_mixup_u8(const uint8_t* a, const uint8_t* b, double r, int64_t n, uint8_t* out, ryolo_stream_t stream);
/* pad_to_square (datasets/base_dataset.py:33-56): src [SH, SW, 3] resized (cv2.resize INTER_LINEAR semantics) to NH x NW and placed at
 * (top, left) of the OH x OW canvas filled with `fill` */
int ryolo_letterbox_u8(const uint8_t* src, int SH, int SW, int NH, int NW, int top, int left, uint8_t* dst, int OH, int OW, int fill,
                       ryolo_stream_t stream);
/* load_image for a whole batch (datasets/base_dataset.py:170-186): `items` = device array of nitems records {int64 src_off, dst_off; int SH,
 * SW, NH, NW, interp, lut} (ryolo_resize_item_bytes = sizeof): image at pool + src_off resized to NH x NW into stage + dst_off; interp
 * 0 = cv2.INTER_LINEAR, 1 = cv2.INTER_AREA (downscaling, generic float form), 2 = copy; lut >= 0: hsv tables luts[lut][3][256] applied
 * to the resized pixel (lib/augmentations.py:8-21).  max_pixels = largest NH * NW of the batch (grid size). */
int ryolo_resize_item_bytes(int* bytes);
int ryolo_resize_hsv_batch(const uint8_t* pool, const void* items_dev, int nitems, int64_t max_pixels, const uint8_t* luts, uint8_t* stage,
                           ryolo_stream_t stream);
/* label side of the sample composition for every label row of a batch, element-wise (load_target datasets/base_dataset.py:188-222, the
 * mosaic-9 crop :318-330, the vertex warp lib/augmentations.py:67-74): `rows` = device array of LabelRow records (csrc/augment.hip;
 * ryolo_label_row_bytes = sizeof), mats = 3x3 double matrices (row-major) indexed by LabelRow.mat; out [nrows][10] = (slot, class, 8
 * vertex coordinates), NaN vertices for rows a filter dropped (ryolo_encode_labels then removes them, keeping the order). */
int ryolo_label_row_bytes(int* bytes);
int ryolo_label_stage(const void* rows_dev, int64_t nrows, const double* mats, float* out, ryolo_stream_t stream);

int ryolo_struct_sizes(int* sizes /* [11] */);

/* ------------------------------------------------------------------------------------------------------------
 * Loss — replaces ComputeCSLLoss.__call__/build_targets (lib/loss.py:191-331) and ComputeKFIoULoss (:368-492),
 * bbox_ciou (:36-78), KFLoss (:100-150).  Forward + gradient w.r.t. the head maps in one call; items[6] =
 * reg, conf, cls, theta, total (already scaled by the hyp gains), and the number of target rows whose image index lies outside
 * [0, batch) — the reference raises IndexError on those (lib/loss.py:209,385), this library skips them and reports the count.
 * ------------------------------------------------------------------------------------------------------------ */
int ryolo_loss_workspace_bytes(const LossParams* p, size_t* bytes);
int ryolo_loss(const LossParams* p, ryolo_stream_t stream);
/* owner grids of the last ryolo_loss call with THESE params on this workspace: owner[i][cell] >= 0 iff cell of scale i was matched to a target (valid until
 * the next ryolo_loss on the workspace) */
int ryolo_loss_owner_grids(const LossParams* p, const int** owner);
/* autograd chain rule of the loss node (lib/loss.py:256,414 return a [1] tensor; `loss.backward()` hands back d(out)/d(loss) as a
 * device scalar): grad[0..n) *= *scale, skipped ON THE DEVICE when *scale == 1.0f (the usual case) — no host read, capturable */
int ryolo_loss_grad_scale(float* grad, int64_t n, const float* scale, ryolo_stream_t stream);
/* the same for count <= 8 arrays (host arrays of device pointers / lengths) in ONE launch: the three maps and their compact objectness copies */
int ryolo_loss_grad_scale_multi(float* const* grads, const int64_t* n, int count, const float* scale, ryolo_stream_t stream);
/* match counters / records of the last ryolo_loss call with THESE params: count[i] -> one int, rec[i] -> [count][8] ints (b, a, gj, gi, cls, target
 * row, cell, pad) in the reference's enumeration order (lib/loss.py:275-310, :432-471); for tests of the target assignment */
int ryolo_loss_match_records(const LossParams* p, const int** count, const int** rec);

/* ------------------------------------------------------------------------------------------------------------
 * Data side (SURVEY.md §8(f) N2 slice, N4): the batch-finalisation end of BaseDataset.__getitem__ + collate_fn and the box
 * geometry of the detect path.  uint8 images and polygon targets come from whatever produced them (the reference's cv2 pipeline, or
 * a decoded cache resident in HBM); everything after them runs on the device.
 * ------------------------------------------------------------------------------------------------------------ */
/* imgs uint8 [B][H][W][3] BGR -> dst fp32 [B][3][H][W] RGB / 255; flags[b] bit0 = fliplr, bit1 = flipud (null: none).
 * Replaces np.fliplr / np.flipud (lib/augmentations.py:33-42), transpose + [::-1] + float() / 255 (datasets/base_dataset.py:155-157),
 * torch.stack (:166).  Bit-exact. */
int ryolo_to_tensor(const uint8_t* imgs, int B, int H, int W, const uint8_t* flags, float* dst, ryolo_stream_t stream);
/* targets fp32 [nt][10] = (image slot, class, x1, y1, ..., x4, y4) in pixels of the H x W network input, clockwise vertices ->
 * out [count][7 | 187] = (sample, class, x, y, w, h, theta [, csl x180]) in the reference's row order, *count on the device.
 * Replaces BaseDataset.filtering (:340-352, border (0, W, 0, H)), normalize (:354-361), the label half of horizontal_flip /
 * vertical_flip (flags[slot], as above), xyxyxyxy2xywha (lib/general.py:70-104), gaussian_label (base_dataset.py:13-31,143-149)
 * and collate_fn's sample index (:161-164; sample_of_img[slot], null = identity).  workspace: nt ints (csl only). */
int ryolo_encode_labels(const float* targets, int64_t nt, int H, int W, const uint8_t* flags, const int* sample_of_img, int csl,
                        float* out, int* count, int* workspace, ryolo_stream_t stream);
/* xyxyxyxy2xywha (lib/general.py:70-104) alone: clockwise polygons fp32 [n][8] -> (x, y, w, h, theta) fp32 [n][5] */
int ryolo_polys_to_xywha(const float* polys, int64_t n, float* out, ryolo_stream_t stream);
/* dets fp32 [n][7] = (x, y, w, h, theta, conf, cls) as post_process returns them; rescale != 0: rescale_boxes (lib/plot.py:9-31) in
 * place on columns 0-3 with shapes[img_of_det[i]] = original (h, w) and the padded network size current_dim; then
 * xywha2xyxyxyxy (lib/general.py:41-67, cv2.getRotationMatrix2D restated) -> polys fp32 [n][4][2]. */
int ryolo_dets_to_polys(float* dets, const int* img_of_det, const int* shapes, int current_dim, int rescale, int64_t n, float* polys,
                        ryolo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RYOLO_H */
