"""ctypes binding of libevae_hip.so (C ABI: include/evae_hip.h).

There is NO CPU fallback: `load()` raises if the shared library is missing, and every op in
`evae.ops` raises on non-CUDA tensors.  The library is built in-tree by `__graft_entry__.build()`
(or `make -C exemplar-vae_amd/csrc`)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVAE_LIB_PATH: another build of the same library (ablation builds of tools/, never a fallback: the file must exist)
LIB_PATH = os.environ.get("EVAE_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libevae_hip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_z = C.c_size_t
_l = C.c_int64
_u = C.c_uint


class ConvDesc(C.Structure):
    """evae_conv_desc_t"""
    _fields_ = [(n, C.c_int) for n in ("N", "C", "H", "W", "Co", "KH", "KW", "stride", "pad")]


class WtJob(C.Structure):
    """evae_wt_job_t (include/evae_hip.h)"""
    _fields_ = [("w1", C.c_void_p), ("w2", C.c_void_p), ("dst", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("ldt", C.c_int)]


class CtlJob(C.Structure):
    """evae_ctl_job_t"""
    _fields_ = [("stage0", C.c_void_p), ("stage1", C.c_void_p), ("ctl", C.c_void_p), ("bytes", C.c_size_t), ("state", C.c_void_p),
                ("idx_word", C.c_size_t), ("seed_word", C.c_size_t)]


class P6PackJob(C.Structure):
    """evae_p6_pack_job_t"""
    _fields_ = [("x", C.c_void_p), ("x2", C.c_void_p), ("img", C.c_void_p), ("img_bytes", C.c_size_t), ("ld", C.c_longlong),
                ("cols", C.c_int), ("R", C.c_int), ("K", C.c_int), ("flag", C.c_int), ("nks", C.c_int)]


class WgradJob(C.Structure):
    """evae_wgrad_job_t"""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("M", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("ldy", C.c_int), ("ldx", C.c_int), ("accumulate", C.c_int)]


class WgradFinishJob(C.Structure):
    """evae_wgrad_finish_job_t"""
    _fields_ = [("byte_rows", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ldy", C.c_int), ("ldx", C.c_int),
                ("x_scale", C.c_float),
                ("dw", C.c_void_p), ("db", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


class AdamTensor(C.Structure):
    """evae_adam_tensor_t"""
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("numel", _l)]


# name -> (restype, argtypes); mirrors include/evae_hip.h one to one
SIGNATURES = {
    "evae_version": (_i, []),
    "evae_last_error": (C.c_char_p, []),
    "evae_ctl_upload": (_i, [_p, _p, _p, _z, _p, _p, _p, _p]),
    "evae_host_dedup": (_i, [_p, _i, C.c_int64, _i, _p, _p, _p, _p]),
    "evae_prior_set_norm_limit": (_i, [_f]),
    "evae_prior_lse_fwd_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_prior_lse_fwd": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_prior_merge": (_i, [_p, _p, _p, _i, _i, _f, _p, _p, _p]),
    "evae_prior_lse_fwd_splits": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p, _z, C.POINTER(C.c_int), C.POINTER(C.c_int), _p]),
    "evae_prior_elbo_fwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p]),
    "evae_prior_elbo_fwd_coef": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "evae_prior_merge_coef": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _f, _p, _p, _p, _p, _p, _p]),
    "evae_elbo_assemble": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p, _p]),
    "evae_prior_train_applies": (_i, [_i, _i, _i]),
    "evae_prior_train_gave_up": (_i, [_p, _p]),
    "evae_prior_train_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_prior_train_step": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _f, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _p]),
    "evae_prior_train_step_rows": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _p, _p, _p, _f, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_prior_lse_bwd_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_prior_lse_bwd": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_prior_lse_bwd_phased": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _p]),
    "evae_pairdist_topk_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "evae_pairdist_topk": (_i, [_p, _i, _p, _i, _i, _i, _u, _l, _p, _p, _p, _z, _p]),
    "evae_pairwise_distance": (_i, [_p, _i, _p, _i, _i, _p, _p]),
    "evae_topk_merge": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "evae_select_exemplars": (_i, [_p, _i, _p, _i, _p, _p, _p, _p]),
    "evae_dense_fwd_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "evae_gated_dense_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _z, _p]),
    "evae_linear_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _f, _f, _p, _p, _p, _z, _p]),
    "evae_heads_reparam_fwd_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_heads_reparam_fwd": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_heads_reparam_fwd_bcast_applies": (_i, [_i, _i, _i, _i]),
    "evae_heads_reparam_fwd_bcast": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "evae_heads_density_fwd": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _f, _f, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_log_normal_diag_bwd_hardtanh": (_i, [_p, _p, _p, _p, _f, _f, _p, _i, _i, _p, _p, _p, _p]),
    "evae_dense_bwd_data_wt_bytes": (_z, [_i, _i, _i]),
    "evae_dense_bwd_data_wt_ld": (_i, [_i]),
    "evae_dense_bwd_data_wt": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _z, _p]),
    "evae_dense_bwd_data_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "evae_dense_bwd_data": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _z, _p]),
    "evae_dense_bwd_weight_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_dense_bwd_weight": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _p, _z, _p]),
    "evae_dense_bwd_weight_phased": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _p, _z, _i, _p]),
    "evae_dense_u8_supported": (_i, [_i, C.c_longlong]),
    "evae_dense_u8_prepared_bytes": (_z, [_i, _i]),
    "evae_dense_u8_prepare": (_i, [_p, _p, _i, _i, _p, _z, _p]),
    "evae_gated_dense_fwd_u8": (_i, [_p, _p, _i, _i, C.c_longlong, _f, _p, _p, _p, _i, _p, _p, _p]),
    "evae_dense_bwd_weight_u8_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_dense_bwd_weight_u8": (_i, [_p, _i, _i, C.c_longlong, _p, _p, _i, C.c_longlong, _f, _p, _p, _p, _z, _p]),
    "evae_dense_bwd_weight_u8_phased": (_i, [_p, _i, _i, C.c_longlong, _p, _p, _i, C.c_longlong, _f, _p, _p, _p, _z, _i, _p]),
    "evae_dense_bwd_weight_u8_images": (_i, [_i, _i, _i, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "evae_broadcast_scalar": (_i, [_p, _p, _i, _p]),
    "evae_sum_small": (_i, [_p, _i, _p, _p]),
    "evae_thin_configure": (_i, [_i]),
    "evae_p6_nks": (_i, [_i]),
    "evae_p6_nks_rows": (_i, [_i]),
    "evae_p6_image_bytes": (_z, [_i, _i]),
    "evae_gemm_p6_applies": (_i, [_i, _i, _i]),
    "evae_p6_pack_rows": (_i, [_p, _p, _i, _i, C.c_longlong, _i, _p, _z, _p]),
    "evae_p6_pack_cols": (_i, [_p, _p, _i, _i, C.c_longlong, _i, _i, _p, _z, _p]),
    "evae_p6_fill_row": (_i, [_p, _i, _i, _f, _i, _i, _p]),
    "evae_gated_dense_fwd_timg": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "evae_gated_dense_fwd_u8_timg": (_i, [_p, _p, _i, _i, C.c_longlong, _f, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p]),
    "evae_dense_bwd_data_timg": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _z, _p]),
    "evae_gated_dense_fwd_p6t": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p]),
    "evae_dense_bwd_data_p6t": (_i, [_p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _p]),
    "evae_dense_bwd_weight_p6_workspace_bytes": (_z, [_i, _i, _i]),
    "evae_dense_bwd_weight_p6": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _z, _p]),
    "evae_dense_bwd_data_img": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p, _z, _p]),
    "evae_dense_bwd_weight_group": (_i, [_p, _i, _p]),
    "evae_dense_bwd_weight_finish_group": (_i, [_p, _i, _p]),
    "evae_gated_dense_bwd_input": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "evae_gated_dense_bwd_input_ld": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _i, _p]),
    "evae_gather_rows": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "evae_gated_dense_bwd": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _z, _p]),
    "evae_act_bwd": (_i, [_p, _p, _z, _i, _f, _f, _p, _p]),
    "evae_conv2d_workspace_bytes": (_z, [_p, _i, _i]),
    "evae_conv2d_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _f, _f, _p, _p, _p, _p, _z, _p]),
    "evae_conv2d_bwd_data": (_i, [_p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_conv2d_bwd_weight": (_i, [_p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_conv2d_cl_supported": (_i, [_p, _i, _i]),
    "evae_conv2d_cl_dy_stride": (_i, [_i]),
    "evae_conv2d_cl_workspace_bytes": (_z, [_p, _i, _i]),
    "evae_conv2d_cl_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _f, _f, _p, _p, _p, _p, _z, _p]),
    "evae_conv2d_cl_bwd_data": (_i, [_p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_conv2d_cl_bwd_weight": (_i, [_p, _p, _p, _i, _p, _p, _p, _z, _p]),
    "evae_gemm_x6_configure": (_i, [_i, _i]),
    "evae_gemm_x6_applies": (_i, [_i, _i, _i]),
    "evae_elu_fwd": (_i, [_p, _z, _p, _p]),
    "evae_conv2d_cl_res_supported": (_i, [_p]),
    "evae_conv2d_cl_fwd_res": (_i, [_p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_conv2d_cl_bwd_data_res": (_i, [_p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_image_bytes": (_z, [C.c_longlong, _i]),
    "evae_cw_supported": (_i, [_p, _i]),
    "evae_cw_workspace_bytes": (_z, [_p, _i]),
    "evae_cw_pack_image": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "evae_cw_fwd_gated": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _z, _p]),
    "evae_cw_bwd_data_gate": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_gate_bwd_image": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _p, _p, _p]),
    "evae_cw_bwd_weight": (_i, [_p, _i, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_res_supported": (_i, [_p]),
    "evae_cw_res_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_res_bwd_data": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_bwd_weight_plain": (_i, [_p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_first_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "evae_cw_first_workspace_bytes": (_z, []),
    "evae_cw_first_bwd_weight": (_i, [_p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_res_run_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "evae_cw_res_run_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "evae_cw_res_pack_filters": (_i, [_p, _i, _p, _p, _p, _p]),
    "evae_cw_plain_supported": (_i, [_p, _i]),
    "evae_cw_plain_workspace_bytes": (_z, [_p, _i]),
    "evae_cw_pack_image_ex": (_i, [_p, C.c_longlong, _i, _i, _p, C.c_longlong, _i, _i, _i, _i, _i, _p, _p]),
    "evae_cw_upsample2_bwd": (_i, [_p, C.c_longlong, _i, _i, _i, _i, _p, _p]),
    "evae_cw_plain_fwd": (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _z, _p]),
    "evae_cw_plain_bwd_data": (_i, [_p, _i, _p, _p, _p, _i, _p, _p, _z, _p]),
    "evae_weight_norm_set_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "evae_weight_norm_set_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "evae_reparam_logq_fwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _p]),
    "evae_reparam_logq_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "evae_reparam_logq_bwd_hardtanh": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p]),
    "evae_reparam_logq_bwd_hardtanh_tail": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p,
                                                 _p, _i, _p, _p]),
    "evae_log_normal_diag_fwd": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "evae_log_normal_diag_bwd": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "evae_elbo_fwd": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p, _p]),
    "evae_elbo2_fwd": (_i, [_p, _p, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p]),
    "evae_elbo_bwd": (_i, [_p, _i, _p, _i, _p, _i, _p, _f, _i, _p, _p, _p, _p]),
    "evae_step_stats_add": (_i, [_p, _p, _p, _p, _p, _p]),
    "evae_bernoulli_ll_fwd": (_i, [_p, _p, _i, _i, _p, _p]),
    "evae_batch_prologue": (_i, [_p, _l, _p, _i, _i, _i, _p, _p, _l, _p, _i, _p]),
    "evae_batch_prologue_u8": (_i, [_p, _l, _p, _i, _i, _i, _p, _f, _p, _l, _p, _l, _p, _i, _p]),
    "evae_batch_prologue_u8_prepare": (_i, [_p, _l, _p, _i, _i, _i, _p, _f, _p, _l, _p, _l, _p, _i, _p, _p, _i, _i, _p, _z, _p, _i,
                                            _p]),
    "evae_batch_prologue_u8_step": (_i, [_p, _l, _i, _i, _i, _f, _p, _l, _p, _l, _p, _i, _p, _p, _i, _i, _p, _z, _p, _i, _p, _p, _i, _p]),
    "evae_bernoulli_ll_bwd": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "evae_bernoulli_sigmoid_bwd": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "evae_bernoulli_unit_step": (_i, [_p, _p, _i, _i, _p, _f, _p, _p, _p, _p, _p, _p]),
    "evae_log_logistic256_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "evae_log_logistic256_bwd": (_i, [_p, _p, _p, _i, _p, _i, _i, _p, _p, _p, _p]),
    "evae_adam_normgrad_workspace_bytes": (_z, [_i]),
    "evae_adam_normgrad_step": (_i, [_p, _i, _l, _i, _d, _d, _d, _d, _d, _p, _p, _z, _p]),
    "evae_adam_normgrad_step_stats": (_i, [_p, _i, _l, _i, _d, _d, _d, _d, _d, _p, _p, _z, _p, _p, _p, _p, _p, _p, _p]),
}

_lib = None


class EvaeError(RuntimeError):
    pass


def load():
    """Load libevae_hip.so and bind every entry point.  Fails loudly when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvaeError(
            "libevae_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C exemplar-vae_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # torch first: it ships its own libamdhip64, and the library must bind to THAT runtime (the one that owns the tensors'
    # memory and streams).  Loaded the other way round -- build() and smoke() in one process did it -- the process ends up with
    # two HIP runtimes and every launch of this library fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch with include/evae_hip.h
        fn.restype = res
        fn.argtypes = args
    if lib.evae_version() != 1:
        raise EvaeError("libevae_hip.so ABI version %d, expected 1" % lib.evae_version())
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().evae_last_error().decode("utf-8", "replace")
        raise EvaeError("%s failed (%d): %s" % (what, rc, msg))


class count_calls:
    """`with count_calls("evae_cw_") as n:` -- counts this process's calls of the entry points whose name starts with the prefix
    (n: name -> calls) while the block runs.  For tests that must know WHICH kernels family served a model (e.g. that the
    pixel-image convolution operators, not the layer-by-layer ones, ran); the library itself keeps no counters."""

    def __init__(self, prefix):
        self.prefix, self.counts, self._saved = prefix, {}, {}

    def __enter__(self):
        lib = load()
        for name in SIGNATURES:
            if name.startswith(self.prefix):
                fn = getattr(lib, name)
                self._saved[name] = fn

                def proxy(*a, _fn=fn, _name=name):
                    self.counts[_name] = self.counts.get(_name, 0) + 1
                    return _fn(*a)
                setattr(lib, name, proxy)
        return self.counts

    def __exit__(self, *exc):
        lib = load()
        for name, fn in self._saved.items():
            setattr(lib, name, fn)
        return False
