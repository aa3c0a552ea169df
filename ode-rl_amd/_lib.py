"""ctypes binding of libodecgru_hip.so (the C ABI declared in include/odecgru_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes through
the C ABI with raw pointers.  There is NO fallback: if the library is missing the import of any
compute entry point raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODEHIP_LIB") or os.path.join(_HERE, "lib", "libodecgru_hip.so")  # env override: A/B builds

ABI_VERSION = 12   # == odehip_version() of the library these ctypes structs were written against (ODEHIP_ABI_VERSION in the header)
MAX_LAYERS = 8
MAX_STAGES = 7
EULER, MIDPOINT, RK4, DOPRI5 = 0, 1, 2, 3
METHODS = {"euler": EULER, "midpoint": MIDPOINT, "rk4": RK4, "dopri5": DOPRI5}

c_float_p = ctypes.POINTER(ctypes.c_float)


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("src1", ctypes.c_void_p), ("src2", ctypes.c_void_p),
        ("cin1", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("ks", ctypes.c_int),
        ("batch", ctypes.c_int),
        ("w_packed", ctypes.c_void_p), ("w_wino", ctypes.c_void_p), ("w_bf16", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("dst", ctypes.c_void_p), ("relu", ctypes.c_int),
    ]


MAX_PACK_JOBS = 32


class PackJob(ctypes.Structure):   # odehip_pack_job
    _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("cout", ctypes.c_int), ("cin", ctypes.c_int), ("ks", ctypes.c_int),
                ("kind", ctypes.c_int), ("transpose_flip", ctypes.c_int)]


class ConvStack(ctypes.Structure):
    _fields_ = [
        ("n_convs", ctypes.c_int), ("ks", ctypes.c_int),
        ("channels", ctypes.c_int * (MAX_LAYERS + 1)),
        ("w_packed", ctypes.c_void_p * MAX_LAYERS),
        ("w_wino", ctypes.c_void_p * MAX_LAYERS),
        ("w_bf16", ctypes.c_void_p * MAX_LAYERS),
        ("w_fused", ctypes.c_void_p),
        ("bias", ctypes.c_void_p * MAX_LAYERS),
        ("final_tanh", ctypes.c_int),
    ]


class ConvGRUCellDesc(ctypes.Structure):
    _fields_ = [("input", ctypes.c_int), ("hidden", ctypes.c_int), ("ks", ctypes.c_int),
                ("w_gates", ctypes.c_void_p), ("b_gates", ctypes.c_void_p), ("gn_gates_w", ctypes.c_void_p),
                ("gn_gates_b", ctypes.c_void_p), ("w_can", ctypes.c_void_p), ("b_can", ctypes.c_void_p),
                ("gn_can_w", ctypes.c_void_p), ("gn_can_b", ctypes.c_void_p),
                ("w_gates_bf16", ctypes.c_void_p), ("w_can_bf16", ctypes.c_void_p),
                ("w_gates_wino", ctypes.c_void_p), ("w_can_wino", ctypes.c_void_p)]


class EncoderDesc(ctypes.Structure):
    _fields_ = [("f_enc", ConvStack), ("cell", ConvGRUCellDesc), ("head_hidden", ctypes.c_int), ("out_ch", ctypes.c_int),
                ("w_head0", ctypes.c_void_p), ("b_head0", ctypes.c_void_p), ("w_head1", ctypes.c_void_p),
                ("b_head1", ctypes.c_void_p)]


class ConvGRUCellBwd(ctypes.Structure):
    _fields_ = [("w_gates_dx", ctypes.c_void_p), ("w_gates_dh", ctypes.c_void_p), ("w_can_dx", ctypes.c_void_p),
                ("w_can_dh", ctypes.c_void_p), ("bf16", ctypes.c_void_p * 4), ("wino", ctypes.c_void_p * 4)]


class ConvGRUCellGrads(ctypes.Structure):
    _fields_ = [("w_gates", ctypes.c_void_p), ("b_gates", ctypes.c_void_p), ("gn_gates_w", ctypes.c_void_p),
                ("gn_gates_b", ctypes.c_void_p), ("w_can", ctypes.c_void_p), ("b_can", ctypes.c_void_p),
                ("gn_can_w", ctypes.c_void_p), ("gn_can_b", ctypes.c_void_p)]


class EncoderBwd(ctypes.Structure):
    _fields_ = [("f_dgrad", ConvStack), ("w_gates_dx", ctypes.c_void_p), ("w_gates_dh", ctypes.c_void_p),
                ("w_can_dx", ctypes.c_void_p), ("w_can_dh", ctypes.c_void_p), ("w_head0_t", ctypes.c_void_p),
                ("w_head1_t", ctypes.c_void_p), ("bf16", ctypes.c_void_p * 4), ("wino", ctypes.c_void_p * 4)]


class EncoderGrads(ctypes.Structure):
    _fields_ = [("f_w", ctypes.c_void_p * MAX_LAYERS), ("f_b", ctypes.c_void_p * MAX_LAYERS),
                ("w_gates", ctypes.c_void_p), ("b_gates", ctypes.c_void_p), ("gn_gates_w", ctypes.c_void_p),
                ("gn_gates_b", ctypes.c_void_p), ("w_can", ctypes.c_void_p), ("b_can", ctypes.c_void_p),
                ("gn_can_w", ctypes.c_void_p), ("gn_can_b", ctypes.c_void_p), ("w_head0", ctypes.c_void_p),
                ("b_head0", ctypes.c_void_p), ("w_head1", ctypes.c_void_p), ("b_head1", ctypes.c_void_p)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)


class OdeHipError(RuntimeError):
    pass


class AsyncSolveTruncated(OdeHipError):
    """An asynchronous dopri5 solve needed more attempted steps than were enqueued up front (ODEHIP_ETRUNC): the frames it had not
    reached are NaN and everything computed from them since is invalid.  The next asynchronous solve enqueues more; a training step
    can simply be repeated (nothing has been applied to the parameters)."""


_lib = None

# name -> (restype, argtypes); the parity of this table with include/odecgru_hip.h is tested on CPU
SIGNATURES = {
    "odehip_last_error": (ctypes.c_char_p, []),
    "odehip_version": (ctypes.c_int, []),
    "odehip_bf16_weight_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "odehip_pack_conv_weight_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_void_p]),
    "odehip_adam_step": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]),
    "odehip_set_persistent_trajectory": (ctypes.c_int, [ctypes.c_int]),
    "odehip_persistent_trajectory_launches": (ctypes.c_longlong, []),
    "odehip_persistent_error": (ctypes.c_int, [ctypes.c_int]),
    "odehip_upsample2x_bilinear": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "odehip_upsample2x_bilinear_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                                            ctypes.c_void_p]),
    "odehip_bn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "odehip_bn_relu_up2x_forward": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_bn_relu_up2x_backward": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 +
                                     [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_warp_composite": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4),
    "odehip_warp_composite_backward": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3),
    "odehip_winograd5_weight_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "odehip_pack_conv_weight_winograd5": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "odehip_frame_encoder_pack_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "odehip_frame_decoder_pack_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "odehip_pack_frame_encoder": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2),
    "odehip_pack_frame_decoder": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2),
    "odehip_frame_encode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "odehip_frame_decode_backward_workspace_floats": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "odehip_frame_decode_train": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "odehip_frame_decode_backward": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] +
                                     [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_frame_encode_backward_workspace_floats": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "odehip_frame_encode_backward": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_float] +
                                     [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_frame_decode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "odehip_mmnist_render": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "odehip_fused_bf16_weight_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "odehip_pack_convstack_fused_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "odehip_pack_conv_weight_bf16_ks": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.c_void_p]),
    "odehip_debug_repeat_f": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p]),
    "odehip_set_debug_flags": (None, [ctypes.c_int]),
    "odehip_set_norm_allreduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "odehip_set_debug_buffer": (None, [ctypes.c_void_p]),
    "odehip_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "odehip_pack_conv_weight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "odehip_winograd_weight_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "odehip_pack_conv_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "odehip_pack_conv_weight_winograd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_void_p]),
    "odehip_nchw_to_q4": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "odehip_q4_to_nchw": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "odehip_conv_q4": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "odehip_debug_repeat_conv": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.c_void_p]),
    "odehip_convstack_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvStack), ctypes.c_int]),
    "odehip_convstack_forward": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_void_p]),
    "odehip_odeint_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvStack), ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int]),
    "odehip_odeint_fixed": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.c_int, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "odehip_odeint_fixed_backward": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.POINTER(ConvStack), ctypes.c_int,
                                                    ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                                    ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                    ctypes.c_void_p]),
    "odehip_odeint_adjoint_backward": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.POINTER(ConvStack), ctypes.c_int,
                                                      ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_adjoint_dopri5_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvStack), ctypes.c_int, ctypes.c_int,
                                                                ctypes.c_int]),
    "odehip_odeint_adjoint_dopri5_backward": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.POINTER(ConvStack),
                                                             ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                                             ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                                             ctypes.c_void_p, ctypes.c_void_p,
                                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                                             ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
                                                             ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_convgru_cell_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvGRUCellDesc), ctypes.c_int]),
    "odehip_convgru_cell_forward": (ctypes.c_int, [ctypes.POINTER(ConvGRUCellDesc), ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                   ctypes.c_void_p]),
    "odehip_encoder_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(EncoderDesc), ctypes.c_int, ctypes.c_int]),
    "odehip_odeconvgru_encode": (ctypes.c_int, [ctypes.POINTER(EncoderDesc), ctypes.c_void_p,
                                                ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_dopri5_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvStack), ctypes.c_int, ctypes.c_int]),
    "odehip_convgru_cell_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvGRUCellDesc), ctypes.c_int]),
    "odehip_convgru_cell_backward": (ctypes.c_int, [ctypes.POINTER(ConvGRUCellDesc), ctypes.POINTER(ConvGRUCellBwd),
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.POINTER(ConvGRUCellGrads), ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_encoder_train_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(EncoderDesc), ctypes.c_int, ctypes.c_int]),
    "odehip_odeconvgru_encode_train": (ctypes.c_int, [ctypes.POINTER(EncoderDesc), ctypes.c_void_p,
                                                      ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_odeconvgru_encode_backward": (ctypes.c_int, [ctypes.POINTER(EncoderDesc), ctypes.POINTER(EncoderBwd),
                                                         ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                         ctypes.POINTER(EncoderGrads), ctypes.c_void_p, ctypes.c_size_t,
                                                         ctypes.c_void_p]),
    "odehip_odeint_dopri5": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                            ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_dopri5_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvStack), ctypes.c_int, ctypes.c_int,
                                                                 ctypes.c_int]),
    "odehip_dopri5_saving_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvStack), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "odehip_odeint_dopri5_saving": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                                   ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
                                                   ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
                                                   ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_odeint_dopri5_start": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
                                                  ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_odeint_dopri5_collect": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                                    ctypes.POINTER(ctypes.c_int)]),
    "odehip_odeint_dopri5_backward_saved": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.POINTER(ConvStack),
                                                           ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                                           ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "odehip_odeint_dopri5_backward": (ctypes.c_int, [ctypes.POINTER(ConvStack), ctypes.POINTER(ConvStack),
                                                     ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                                     ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                                     ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t,
                                                     ctypes.c_void_p]),
}


def load():
    """Load the shared library once; raise if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OdeHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ode-rl_amd/csrc`.  The HIP path has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    got = lib.odehip_version()
    if got != ABI_VERSION:   # the struct layouts above belong to exactly one ABI: an older / newer .so would read them wrongly
        raise OdeHipError(f"{LIB_PATH} reports ABI version {got}, this binding needs {ABI_VERSION}: rebuild the library (make -C ode-rl_amd/csrc)")
    _lib = lib
    return lib


def check(rc):
    """Map the C status code to the exception the reference's Python path would raise.  Also looks at the sticky error word of
    the persistent launches (a capped in-kernel wait that gave up): whatever launch set it has produced invalid results."""
    if rc == 0:
        code = _lib.odehip_persistent_error(1) if _lib is not None else 0
        if code:
            raise OdeHipError(f"a persistent launch of THIS OR AN EARLIER call gave up waiting for a partner workgroup (code {code}): the word is "
                              "read without synchronising, so the kernel that set it may belong to any call enqueued since the last check.  "
                              "Results since then are invalid (whole-trajectory launches NaN-fill their outputs; single-evaluation launches "
                              "-- ODEHIP_PERSISTENT_SMALL=1 -- do not); persistent launches are now disabled for this process")
        return
    msg = load().odehip_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc in (-3, -4):
        raise AssertionError(msg)
    if rc == -5:
        raise AsyncSolveTruncated(msg)
    raise OdeHipError(f"odehip status {rc}: {msg}")
