"""ctypes mirrors of the POD parameter blocks of csrc/conv.hip, elementwise.hip and loss.hip (checked against the
library's own sizeof at load time) and the registration of their entry points with hip.py."""
import ctypes as C

from .. import hip

P, I, L, F, D, Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
MAX_TAPS = 9
EPI_RAW, EPI_STATS, EPI_AFFINE_ACT, EPI_F32_BIAS, EPI_ACCUM, EPI_AFFINE_ACT_R = range(6)
ACT = {"linear": 0, "mish": 1, "leaky": 2, "swish": 3}


class TapClass(C.Structure):
    _fields_ = [("ntaps", I), ("oh_add", I), ("ow_add", I), ("dh", C.c_byte * MAX_TAPS), ("dw", C.c_byte * MAX_TAPS),
                ("widx", C.c_byte * MAX_TAPS)]


class ConvGemmParams(C.Structure):
    _fields_ = [("A", P), ("NB", I), ("IH", I), ("IW", I), ("Cin", I), ("ldA", I),
                ("W", P), ("Nout", I), ("wtaps", I), ("OH", I), ("OW", I), ("sh", I), ("sw", I),
                ("oh_mul", I), ("ow_mul", I), ("OHf", I), ("OWf", I), ("nclasses", I), ("cls", TapClass * 4),
                ("epi", I), ("out", P), ("ldC", I), ("stats", P), ("scale", P), ("shift", P), ("act", I), ("bias", P), ("zeros", P), ("pipe", I), ("a_bytes", C.c_uint), ("w_bytes", C.c_uint),
                ("pool_idx", P), ("pool_dz", P), ("pool_ldi", I), ("pool_ld", I), ("s2d_cin", I), ("head_attrs", I), ("head_och", I)]


class WgradParams(C.Structure):
    _fields_ = [("dY", P), ("ldY", I), ("Cout", I), ("CoutPad", I),
                ("X", P), ("NB", I), ("IH", I), ("IW", I), ("Cin", I), ("ldX", I),
                ("OH", I), ("OW", I), ("sh", I), ("sw", I), ("ntaps", I), ("dh", C.c_byte * MAX_TAPS), ("dw", C.c_byte * MAX_TAPS),
                ("dW", P), ("splitk", I), ("kchunk", L), ("partial", P), ("zeros", P), ("dW2", P), ("Cout1", I)]


class StemParams(C.Structure):
    _fields_ = [("img", P), ("NB", I), ("H", I), ("W", I), ("wf", P), ("Cout", I), ("epi", I), ("out", P), ("ldC", I), ("stats", P),
                ("scale", P), ("shift", P), ("act", I)]


class StemWgradParams(C.Structure):
    _fields_ = [("img", P), ("NB", I), ("H", I), ("W", I), ("dY", P), ("ldY", I), ("Cout", I), ("scratch", P), ("workspace", P),
                ("y", P), ("ldy", I), ("act", I), ("co", P), ("bco", P)]


class StemBwdParams(C.Structure):
    _fields_ = [("img", P), ("NB", I), ("H", I), ("W", I), ("dz", P), ("lddz", I), ("wf", P), ("co", P), ("act", I), ("frozen", I),
                ("workspace", P), ("dW", P), ("dgamma", P), ("dbeta", P)]


class BnActParams(C.Structure):
    _fields_ = [("y1", P), ("ld1", I), ("co1", P), ("y2", P), ("ld2", I), ("co2", P), ("res", P), ("ldr", I),
                ("z", P), ("ldz", I), ("M", L), ("C", I), ("act", I),
                ("dz", P), ("lddz", I), ("dy1", P), ("lddy1", I), ("dy2", P), ("lddy2", I),
                ("dres", P), ("lddres", I), ("dres_accum", I), ("partial", P), ("bco", P), ("rows_per_block", I)]


class PoolParams(C.Structure):
    _fields_ = [("x", P), ("ldx", I), ("z", P), ("ldz", I), ("NB", I), ("H", I), ("W", I), ("C", I), ("k", I), ("stride", I),
                ("pad", I), ("OH", I), ("OW", I), ("idx", P), ("dz", P), ("lddz", I), ("dx", P), ("lddx", I), ("accum", I),
                ("rowmax", P), ("rowidx", P), ("growws", P)]


class UpParams(C.Structure):
    _fields_ = [("x", P), ("ldx", I), ("z", P), ("ldz", I), ("NB", I), ("H", I), ("W", I), ("C", I), ("accum", I)]


class PackEntry(C.Structure):
    _fields_ = [("src", P), ("wf", P), ("wd", P), ("Cout", I), ("Cin", I), ("taps", I), ("CinP", I), ("CoutP", I), ("ldWd", I),
                ("start", L), ("wd_scale", P)]


class LossParams(C.Structure):
    _fields_ = [("mode", I), ("nc", I), ("na", I), ("batch", I), ("nt", I), ("tcols", I), ("targets", P),
                ("head", P * 3), ("grad", P * 3), ("gs", I * 3), ("anchors", (F * 3 * 18) * 3),
                ("box", F), ("obj", F), ("cls", F), ("theta_gain", F), ("obj_pw", F), ("cls_pw", F),
                ("ws", P), ("ws_bytes", Z), ("items", P), ("compute_grad", I), ("fl_gamma", F), ("fl_alpha", F), ("objgrad", P * 3), ("headobj", P * 3)]


_PTR = C.POINTER
for _name, _sig in {
    "ryolo_conv_gemm": [_PTR(ConvGemmParams), P],
    "ryolo_conv_gemm_stats_rows": [L, I, I, _PTR(I)],
    "ryolo_conv_gemm_plan": [_PTR(ConvGemmParams), _PTR(I), _PTR(I)],
    "ryolo_conv_wgrad": [_PTR(WgradParams), P],
    "ryolo_conv_wgrad_plan": [_PTR(WgradParams), _PTR(I), _PTR(Z)],
    "ryolo_conv_wgrad_kernel": [_PTR(WgradParams), _PTR(I)],
    "ryolo_conv_wgrad_grid": [_PTR(WgradParams), _PTR(I), _PTR(I)],
    "ryolo_stem3x3_plan": [I, I, I, I, _PTR(I), _PTR(Z)],
    "ryolo_stem3x3_fwd": [_PTR(StemParams), P],
    "ryolo_stem3x3_wgrad": [_PTR(StemWgradParams), P],
    "ryolo_stem3x3_bwd_plan": [I, I, I, I, _PTR(Z)],
    "ryolo_stem3x3_bwd": [_PTR(StemBwdParams), P],
    "ryolo_bn_finalize": [P, I, I, D, F, F, P, P, P, P, P, P],
    "ryolo_bn_eval_coeffs": [P, P, P, P, F, I, P, P],
    "ryolo_bn_finalize_slice": [P, I, I, I, I, D, F, F, P, P, P, P, P, P],
    "ryolo_bn_eval_coeffs_slice": [P, P, P, P, F, I, P, I, I, P],
    "ryolo_bn_act_fwd": [_PTR(BnActParams), P],
    "ryolo_bn_act_bwd_blocks": [L, I, _PTR(I), _PTR(I)],
    "ryolo_bn_act_bwd": [_PTR(BnActParams), P, P, P, P, P, I, P],
    "ryolo_maxpool_fwd": [_PTR(PoolParams), P],
    "ryolo_maxpool_bwd": [_PTR(PoolParams), P],
    "ryolo_upsample2x_fwd": [_PTR(UpParams), P],
    "ryolo_upsample2x_bwd": [_PTR(UpParams), P],
    "ryolo_im2col": [P, I, I, I, I, I, I, I, I, I, I, I, P, P],
    "ryolo_head_finish_fwd": [P, I, P, I, I, I, I, P, P],
    "ryolo_head_finish_bwd": [P, P, I, P, I, I, I, I, P, I, P, P, P, P],
    "ryolo_head_finish_bwd_sparse": [P, P, P, I, P, P, I, P, I, I, I, I, P, I, P, P, P, P],
    "ryolo_head_finish_fwd_obj": [P, I, P, I, I, I, I, P, I, P, P, P],
    "ryolo_head_wgrad_finish": [P, P, P, P, P, P, I, I, P, P, P, P, P],
    "ryolo_head_bias_fold": [P, P, P, I, I, P, P],
    "ryolo_chan_add": [P, I, P, L, I, P, I, P],
    "ryolo_colsum_bf16": [P, I, L, I, I, P, P, P],
    "ryolo_pack_weights": [P, I, L, P],
    "ryolo_pack_s2d": [P, I, I, P, P],
    "ryolo_unpack_wgrad": [P, I, I, I, I, P, P],
    "ryolo_repconv_fold": [P, P, P, P, I, I, P, P, P],
    "ryolo_sgd_nesterov": [P, P, P, L, F, F, F, I, P],
    "ryolo_adam": [P, P, P, P, L, D, D, D, D, L, F, I, P],
    "ryolo_struct_sizes": [_PTR(I)],
    "ryolo_loss_workspace_bytes": [_PTR(LossParams), _PTR(Z)],
    "ryolo_loss": [_PTR(LossParams), P],
    "ryolo_loss_owner_grids": [_PTR(LossParams), _PTR(P * 3)],
    "ryolo_loss_grad_scale": [P, L, P, P],
    "ryolo_loss_grad_scale_multi": [_PTR(P * 8), _PTR(L * 8), I, P, P],
    "ryolo_loss_match_records": [_PTR(LossParams), _PTR(P * 3), _PTR(P * 3)],
}.items():
    hip.register(_name, _sig)

_checked = False


def check_layouts():
    """Fail loudly if a ctypes mirror drifted from the C struct it shadows."""
    global _checked
    if _checked:
        return
    sizes = (I * 11)()
    hip.call("ryolo_struct_sizes", sizes)
    want = [BnActParams, PoolParams, UpParams, PackEntry, ConvGemmParams, WgradParams, LossParams, TapClass, StemParams, StemWgradParams, StemBwdParams]
    for k, t in enumerate(want):
        if sizes[k] != C.sizeof(t):
            raise RuntimeError(f"ryolov4_amd: struct layout mismatch for {t.__name__}: C {sizes[k]} vs ctypes {C.sizeof(t)}")
    _checked = True
