/* ryolo_params.h — POD parameter blocks of the libryolo_hip.so C ABI (plain C; mirrored with ctypes in
 * r-yolov4_amd/engine/structs.py and size-checked at load time through ryolo_struct_sizes()). */
#ifndef RYOLO_PARAMS_H
#define RYOLO_PARAMS_H
#include <stddef.h>
#include <stdint.h>

typedef unsigned short bf16_t;          /* raw bfloat16 bits */
#define RY_MAX_TAPS 9
#define LOSS_MAX_NA 18


typedef struct TapClass {
    int ntaps;
    int oh_add, ow_add;                      // full-grid output pixel = (oh*oh_mul + oh_add, ow*ow_mul + ow_add)
    signed char dh[RY_MAX_TAPS], dw[RY_MAX_TAPS], widx[RY_MAX_TAPS];
} TapClass;

typedef struct ConvGemmParams {
    const bf16_t* A; int NB, IH, IW, Cin, ldA;      // gathered operand: [NB, IH, IW, Cin] with channel stride ldA
    const bf16_t* W; int Nout, wtaps;               // packed weights [Nout][wtaps][Cin]
    int OH, OW;                                     // iteration grid per image (per class); M = NB*OH*OW
    int sh, sw;                                     // input coordinate = o*s + d[tap]
    int oh_mul, ow_mul, OHf, OWf;                   // full output grid (differs from OH/OW only for strided dgrad)
    int nclasses;
    TapClass cls[4];
    int epi;                                        // see EPI_*
    void* out; int ldC;
    float* stats;                                   // EPI_STATS: [gridM][2][Nout]
    const float* scale; const float* shift; int act; // EPI_AFFINE_ACT
    const float* bias;                              // EPI_F32_BIAS (may be null)
    const bf16_t* zeros;                            // >= 64 zero bytes in device memory: source of padded / out-of-range rows (LDS-DMA path)
    int pipe;                                       // bits 0-7 mainloop of the generic kernel: 0 register-staged double buffer, 1 LDS-DMA ring;
                                                    // 0x100 force 32-channel stages; 0x200 3x3 stride-1 layers with >= 512 tiles run the halo-patch kernel (conv3x3.hip), 0x400 also smaller ones
    unsigned a_bytes, w_bytes;                      // byte extents of A / W (informational; reserved for buffer-descriptor addressing)
    const unsigned char* pool_idx; const bf16_t* pool_dz; int pool_ldi, pool_ld;
                                                    // pool_idx != null (bf16 epilogues of the generic kernel, identity output grid): the gradient of a
                                                    // MaxPool2d(2, 2) of the SAME tensor is added in the store — out[(h, w), n] += pool_dz[(h/2, w/2), n]
                                                    // where pool_idx[(h/2, w/2), n] == (h & 1) * 2 + (w & 1) (argmax window offsets of ryolo_maxpool_fwd;
                                                    // row strides pool_ldi / pool_ld elements).  Replaces a read-modify-write pass of ryolo_maxpool_bwd
                                                    // over the full-resolution gradient (MaxConv, model/utils.py:146-160); bits equal the two-pass result
    int s2d_cin;                                    // > 0: depth-to-space store of a stride-2 data gradient computed as ONE stride-1 GEMM over the dY
                                                    // grid with Nout = 4 * s2d_cin columns (output parity ph, pw, then channel) and 2x2 taps (weights from
                                                    // ryolo_pack_s2d): column n of row (img, a, b) goes to pixel (2a + ph, 2b + pw), channel n % s2d_cin —
                                                    // every store instruction writes whole 128-byte pixel pairs (oh_mul = ow_mul = 2, bf16 epilogues only)
    int head_attrs, head_och;                       // head_attrs > 0 (EPI_F32_BIAS, 1x1 stride 1, LDS-DMA mainloop; else RY_ERR_UNSUPPORTED): the output is a
                                                    // detection head written in its FINAL layout — column n = anchor n / head_attrs, attribute n % head_attrs
                                                    // of GEMM row (image b, cell) goes to out[b][anchor][cell][attribute] (fp32 [NB, na, OH, OW, attrs]:
                                                    // model/yololayer.py:25) after the bias and, if `scale` is set, the per-column ImplicitM factor
                                                    // (model/neck.py:186); `stats`, if set, receives the objectness logit (attribute head_och) of every
                                                    // (anchor, cell) as a compact [NB, na, OH, OW] array (LossParams.headobj).  ldC is ignored.
} ConvGemmParams;

typedef struct WgradParams {
    const bf16_t* dY; int ldY, Cout, CoutPad;        // [M][CoutPad] readable (zero padded), Cout rows stored
    const bf16_t* X; int NB, IH, IW, Cin, ldX;        // gathered by tap
    int OH, OW, sh, sw;
    int ntaps; signed char dh[RY_MAX_TAPS], dw[RY_MAX_TAPS];
    float* dW;                                       // torch layout [Cout][Cin][ntaps] fp32, atomically accumulated
    int splitk; int64_t kchunk;                      // pixels per split (multiple of BK) — filled by the library
    float* partial;                                  // split-K workspace [splitk][Cout][ntaps*Cin] fp32 (ryolo_conv_wgrad_plan)
    const bf16_t* zeros;                             // >= 64 zero bytes in device memory (LDS-DMA source of padding rows); null: generic kernel only
    float* dW2; int Cout1;                           // dW2 != null: output channels >= Cout1 belong to a second parameter tensor (sibling
                                                     // convolutions sharing one launch): row co of the GEMM goes to dW2[co - Cout1]
} WgradParams;

/* first layer (3x3 stride 1, Cin = 3) computed directly from the fp32 NCHW image: csrc/stem.hip */
typedef struct StemParams {
    const float* img; int NB, H, W;                  // [NB, 3, H, W] fp32 (what train.py:186 hands over)
    const bf16_t* wf; int Cout;                      // packed weights [Cout][32], k = (r*3 + s)*3 + c, zero padded 27 -> 32; Cout <= 32
    int epi;                                         // 0 raw, 1 BatchNorm statistics, 2 folded BN + activation (fp32 accumulator),
                                                     // 5 the same on the bf16-ROUNDED conv output (training: bit-identical to storing the
                                                     // raw output and running ryolo_bn_act_fwd over it)
    bf16_t* out; int ldC;                            // out == null with epi 1: statistics only, nothing is stored
    float* stats;                                    // epi 1: [rows][2][Cout], rows from ryolo_stem3x3_plan
    const float* scale; const float* shift; int act; // epi 2
} StemParams;

typedef struct StemWgradParams {
    const float* img; int NB, H, W;
    const bf16_t* dY; int ldY, Cout;                 // [NB*H*W][ldY] bf16, Cout == 32
    float* scratch;                                  // out: dW in the GEMM layout [Cout][32] fp32 (overwritten; ryolo_unpack_wgrad adds it to .grad)
    float* workspace;                                // ryolo_stem3x3_plan bytes
    // fused BatchNorm + activation backward (y != null): dY then holds dz = dL/d act(bn(y)) and the kernel forms the gradient of the
    // raw conv output on the fly, dy = sc*g + A*y + B with g = dz*act'(sc*y + sh) (the algebra of ryolo_bn_act_bwd's apply pass,
    // which is then skipped: its 3 tensor passes over the largest activation of the network go away)
    const bf16_t* y; int ldy; int act;               // raw conv output [NB*H*W][ldy]
    const float* co;                                 // [4][Cout]: mean, invstd, scale, shift (ryolo_bn_finalize / ryolo_bn_eval_coeffs)
    const float* bco;                                // [2][Cout]: mean g, mean g*xhat (ryolo_bn_act_bwd with dy1 == null)
} StemWgradParams;

/* Whole backward of the first layer in ONE pass over dz (csrc/stem.hip): the raw conv output is RECOMPUTED from the image (K = 27),
 * never stored; BatchNorm-backward sums, the BatchNorm parameter gradients and the weight gradient come out of the same pass:
 * dW = sc*G + A*(W.XX) + B*X1 with G = sum g (x) patch, XX = sum patch (x) patch, g = dz*act'(sc*y + sh) (linear in g, so no second
 * pass with the finished statistics is needed). */
typedef struct StemBwdParams {
    const float* img; int NB, H, W;                  // [NB, 3, H, W] fp32, W % 32 == 0
    const bf16_t* dz; int lddz;                      // gradient of the layer's activation output [NB*H*W][lddz], 32 channels
    const bf16_t* wf;                                // packed forward weights [32][32] (k = (r*3 + s)*3 + c, zero padded)
    const float* co;                                 // [4][32]: mean, invstd, scale, shift
    int act, frozen;                                 // frozen: fixed affine map (no coupling through the batch statistics)
    float* workspace;                                // ryolo_stem3x3_bwd_plan bytes
    float* dW;                                       // fp32 [32][3][3][3] (torch layout), accumulated
    float* dgamma; float* dbeta;                     // accumulated; may be null
} StemBwdParams;

typedef struct BnActParams {
    const bf16_t* y1; int ld1; const float* co1;      // co = [4][C]: mean, invstd, scale, shift
    const bf16_t* y2; int ld2; const float* co2;      // optional second branch (RepConv rbr_1x1), summed before the activation
    const bf16_t* res; int ldr;                       // optional residual added AFTER the activation (Bottleneck)
    bf16_t* z; int ldz;
    int64_t M; int C; int act;
    // backward only
    const bf16_t* dz; int lddz;
    bf16_t* dy1; int lddy1; bf16_t* dy2; int lddy2;
    bf16_t* dres; int lddres; int dres_accum;
    float* partial;                                   // [nblk][K][C], K = 2 (one branch) or 3
    const float* bco;                                 // backward coefficients [K][C]: mean_g, mean_gx1, mean_gx2
    int rows_per_block;
} BnActParams;

typedef struct PoolParams {
    const bf16_t* x; int ldx; bf16_t* z; int ldz;
    int NB, H, W, C, k, stride, pad, OH, OW;
    unsigned char* idx;            // [NB,OH,OW,C] argmax window offset (k>2 only)
    const bf16_t* dz; int lddz; bf16_t* dx; int lddx; int accum;
    /* optional, stride-1 windows (SPP / SPPF / SPPCSPC, k = 5, 9, 13): with rowmax set the pool runs as a row pass + a column
     * pass (2k reads per output instead of k*k) with the SAME first-maximum argmax as the direct form */
    bf16_t* rowmax;                /* [NB,H,W,C] row-window maxima (forward scratch) */
    unsigned char* rowidx;         /* [NB,H,W,C] their first-max column offset (kept for backward; null in eval plans) */
    float* growws;                 /* [NB,H,W,C] fp32 scratch of the backward column pass */
} PoolParams;

typedef struct UpParams { const bf16_t* x; int ldx; bf16_t* z; int ldz; int NB, H, W, C; int accum; } UpParams;

/* ldWd: row length of the [Cin][taps][...] data-gradient image (0 = CoutP); larger when sibling convolutions share one image and
 * this entry owns a column range of it (wd then points at its first column) */
typedef struct PackEntry { const float* src; bf16_t* wf; bf16_t* wd; int Cout, Cin, taps, CinP, CoutP, ldWd; int64_t start;
                           const float* wd_scale;   /* optional [Cout]: the data-gradient image holds W[co] * wd_scale[co] (detection heads with ImplicitM:
                                                       dx = dout . (W * m), model/neck.py:186) */
} PackEntry;

typedef struct LossParams {
    int mode;                 // 0 csl, 1 kfiou
    int nc, na, batch, nt, tcols;
    const float* targets;     // [nt, tcols]  (img, cls, x, y, w, h, theta[, csl x 180])
    const float* head[3];     // [B, na, gs, gs, attrs]
    float* grad[3];           // same shape, d(total_loss)/d(logit); may be null when compute_grad == 0
    int gs[3];
    float anchors[3][LOSS_MAX_NA][3];
    float box, obj, cls, theta_gain, obj_pw, cls_pw;
    void* ws; size_t ws_bytes;
    float* items;             // [6] device: reg, conf, cls, theta, total, number of target rows dropped because their image index is outside [0, batch)
    int compute_grad;
    float fl_gamma, fl_alpha; // FocalLoss (lib/loss.py:10-33) around every BCE term when fl_gamma > 0 (hyp['fl_gamma']); alpha = 0.25 in the reference
    float* objgrad[3];        // optional (compute_grad): the objectness gradient of every cell ALSO as a compact [B, na, gs, gs] array — with the owner
                              // grids (ryolo_loss_owner_grids) the sparse description of grad[]: unmatched cells are zero except their objectness
                              // element.  ryolo_head_finish_bwd_sparse reads that instead of the dense 88-byte rows (r05)
    const float* headobj[3];  // optional: the objectness logit of every cell (head[i][cell * attrs + och]) as a compact [B, na, gs, gs] array, as
                              // ryolo_head_finish_fwd_obj leaves it; the objectness pass then reads 4 bytes per cell instead of one strided element
                              // of every row (= the whole map through the cache).  Must hold exactly the head maps' values.
} LossParams;

#endif /* RYOLO_PARAMS_H */
