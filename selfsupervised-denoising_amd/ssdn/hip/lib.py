"""ctypes binding of libssdn_hip.so (C-ABI declared in include/ssdn_hip.h -- the struct mirrors below follow that
header field by field; tests/test_abi.py checks sizes and that every declared symbol is exported).

The product path has NO CPU fallback: if the shared library is missing, `load()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSDN_HIP_LIB: another build of the SAME library (tools/build_tuning.sh: -DSSDN_TUNING with ablation bits and stamps); never a fallback
LIB_PATH = os.environ.get("SSDN_HIP_LIB") or os.path.join(_HERE, "libssdn_hip.so")
MAX_TAPS = 9

# op type codes (enum ssdn_op_type)
OP = dict(pack_input=1, conv=2, pool_fwd=3, pool_bwd=4, upsum_bwd=5, unrot_fwd=6, unrot_bwd=7, wgrad=8, wreduce=9,
          wpack=10, grad_pack=11, head_ssdn=12, head_final=13, spatial_mean=14, mse=15, mask_mse=16, adam=17,
          metrics=18, zero=19, event_record=20, noise=21)

i32, f32, vp = C.c_int32, C.c_float, C.c_void_p


class View(C.Structure):
    _fields_ = [("p", vp), ("cs", i32), ("co", i32)]


class OpRec(C.Structure):
    _fields_ = [("type", i32), ("lane", i32), ("args", vp)]


class PackInputArgs(C.Structure):
    _fields_ = [("src", vp), ("dst", View), ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("R", i32), ("cpad", i32)]


class ConvArgs(C.Structure):
    _fields_ = [("src0", View), ("src1", View), ("c0", i32), ("c1", i32), ("up0", i32), ("N", i32), ("H", i32), ("W", i32),
                ("ntaps", i32), ("dy", i32 * MAX_TAPS), ("dx", i32 * MAX_TAPS), ("w", vp), ("M", i32), ("Mpad", i32),
                ("Ktot", i32), ("bias", vp), ("act", i32), ("mask", View), ("add", View), ("dst", View), ("dst32", vp),
                ("ltw", i32), ("lth", i32), ("ltn", i32), ("kc", i32), ("bf16", i32), ("wc", vp), ("kreal", i32),
                ("pool", View), ("pool_shifted", i32), ("upsum", View), ("upsum_mask", View), ("upsum_c", i32),
                ("unrot", View), ("unrot_mask", View), ("unrot_smask", vp),
                ("urot", View), ("urot_smask", vp), ("sign_out", vp), ("mask_sign", vp), ("upsum_mask_sign", vp)]


class PoolArgs(C.Structure):
    _fields_ = [("act", View), ("pooled", View), ("dpool", View), ("dz", View), ("N", i32), ("H", i32), ("W", i32),
                ("C", i32), ("shifted", i32), ("route", vp)]


class UpsumArgs(C.Structure):
    _fields_ = [("src", View), ("mask", View), ("dst", View), ("N", i32), ("H", i32), ("W", i32), ("C", i32)]


class UnrotArgs(C.Structure):
    _fields_ = [("src", View), ("dst", View), ("mask", View), ("B", i32), ("P", i32), ("C", i32), ("smask", vp)]


class WgradArgs(C.Structure):
    _fields_ = [("dz", View), ("src0", View), ("src1", View), ("c0", i32), ("c1", i32), ("up0", i32), ("N", i32),
                ("H", i32), ("W", i32), ("ntaps", i32), ("dy", i32 * MAX_TAPS), ("dx", i32 * MAX_TAPS), ("coff", i32 * MAX_TAPS),
                ("M", i32), ("Mpad", i32), ("Ktot", i32), ("Kpad", i32), ("slab", vp), ("bslab", vp), ("nslabs", i32), ("ltw", i32),
                ("lth", i32), ("ltn", i32), ("csplit", i32), ("mblocks", i32), ("kreal", i32), ("mega", i32), ("cost", C.c_float)]


class WreduceArgs(C.Structure):
    _fields_ = [("slab", vp), ("bslab", vp), ("nslabs", i32), ("ntaps", i32), ("M", i32), ("Mpad", i32), ("Kpad", i32),
                ("cin", i32), ("cin_full", i32), ("m_off", i32), ("c_off", i32), ("tapblock", i32), ("gw", vp), ("gb", vp),
                ("inv_scale", vp)]


class WpackArgs(C.Structure):
    _fields_ = [("w", vp), ("wf", vp), ("wd", vp), ("M", i32), ("cin", i32), ("ntaps", i32), ("c0", i32), ("c1_real", i32),
                ("Mpad_f", i32), ("Ktot", i32), ("Mpad_d", i32), ("Kd", i32), ("wfc", vp), ("wdc", vp)]


class GradPackArgs(C.Structure):
    _fields_ = [("g", vp), ("dst", View), ("N", i32), ("C", i32), ("H", i32), ("W", i32), ("cpad", i32), ("gmax", vp),
                ("scale_out", vp)]


class HeadArgs(C.Structure):
    _fields_ = [("net_out", vp), ("noisy", vp), ("noise_param", vp), ("est_raw", vp), ("B", i32), ("C", i32), ("H", i32),
                ("W", i32), ("style", i32), ("mode", i32), ("want_grad", i32), ("mu", vp), ("pme", vp), ("model_std", vp),
                ("noise_std", vp), ("g_net_out", vp), ("partial", vp), ("nchunks", i32), ("gmax", vp)]


class HeadFinalArgs(C.Structure):
    _fields_ = [("partial", vp), ("B", i32), ("nchunks", i32), ("H", i32), ("W", i32), ("mode", i32), ("loss", vp),
                ("g_est", vp), ("g_sigma_out", vp), ("gmax2", vp)]


class SpatialMeanArgs(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("B", i32), ("HW", i32)]


class MseArgs(C.Structure):
    _fields_ = [("out", vp), ("ref", vp), ("coords", vp), ("ncoords", i32), ("B", i32), ("C", i32), ("H", i32), ("W", i32),
                ("loss", vp), ("g", vp), ("gmax", vp)]


class AdamArgs(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", C.c_int64), ("lr", f32), ("b1", f32), ("b2", f32),
                ("eps", f32), ("bc1", f32), ("bc2", f32), ("gscale", f32)]


class MetricsArgs(C.Structure):
    _fields_ = [("out", vp), ("mu", vp), ("clean", vp), ("loss", vp), ("model_std", vp), ("noise_std", vp), ("ext", vp),
                ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("noise_n", i32), ("per", vp), ("acc", vp)]


class ZeroArgs(C.Structure):
    _fields_ = [("p", vp), ("bytes", C.c_int64)]


class EventArgs(C.Structure):
    _fields_ = [("event", vp)]


class NoiseArgs(C.Structure):
    _fields_ = [("clean_u8", vp), ("clean32", vp), ("noisy32", vp), ("ref32", vp), ("param", vp), ("param_ref", vp), ("coords", vp),
                ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("style", i32), ("clip", i32), ("p_lo", f32), ("p_hi", f32),
                ("n2v_box", i32), ("n2v_radius", i32), ("seed", C.c_uint64), ("offset", C.c_uint64)]


ARG_TYPES = dict(pack_input=PackInputArgs, conv=ConvArgs, pool_fwd=PoolArgs, pool_bwd=PoolArgs, upsum_bwd=UpsumArgs,
                 unrot_fwd=UnrotArgs, unrot_bwd=UnrotArgs, wgrad=WgradArgs, wreduce=WreduceArgs, wpack=WpackArgs,
                 grad_pack=GradPackArgs, head_ssdn=HeadArgs, head_final=HeadFinalArgs, spatial_mean=SpatialMeanArgs,
                 mse=MseArgs, mask_mse=MseArgs, adam=AdamArgs, metrics=MetricsArgs, zero=ZeroArgs, event_record=EventArgs, noise=NoiseArgs)

# every symbol include/ssdn_hip.h declares
ABI_VERSION = 14      # SSDN_ABI_VERSION of include/ssdn_hip.h this binding mirrors

SYMBOLS = ["ssdn_run_ops", "ssdn_stream_order", "ssdn_conv_lds_bytes", "ssdn_wgrad_lds_bytes", "ssdn_abi_version", "ssdn_last_error",
           "ssdn_device_cus", "ssdn_probe_mfma", "ssdn_probe_tr16", "ssdn_struct_size", "ssdn_profile_enable",
           "ssdn_profile_read", "ssdn_profile_set_stride", "ssdn_debug_set_trace", "ssdn_debug_get_trace", "ssdn_conv_set_mode",
           "ssdn_wgrad_mergeable", "ssdn_conv_fuses_pool", "ssdn_conv_fuses_upsum", "ssdn_conv_fuses_unrot", "ssdn_conv_fuses_urot", "ssdn_conv_signs", "ssdn_chain_len",
           "ssdn_conv_set_chain", "ssdn_wgrad_mega_ok", "ssdn_wgrad_variant",
           "ssdn_plan_load", "ssdn_plan_destroy", "ssdn_plan_arena_bytes", "ssdn_plan_meta", "ssdn_plan_bind", "ssdn_plan_tensor", "ssdn_plan_run",
           "ssdn_plan_set_lr", "ssdn_net_forward", "ssdn_train_step"]
PLAN_PHASES = dict(repack=0, forward=1, backward=2, optimiser=3)
PROF = dict(conv_mt3=0, conv_mt2=1, conv_mt1=2, wgrad=3, gemm=4, cdma_mt3=5, cdma_mt21=6, wgrad_side=7)

def pointer_fields(st, base: int = 0):
    """byte offsets of every pointer inside a (nested) argument struct: c_void_p fields and the `p` of ssdn_view"""
    out = []
    for name, ty in st._fields_:
        off = base + getattr(st, name).offset
        if ty is vp:
            out.append(off)
        elif isinstance(ty, type) and issubclass(ty, C.Structure):
            out += pointer_fields(ty, off)
    return out


_lib = None


class SsdnHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libssdn_hip.so or fail loudly -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsdnHipError("libssdn_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the ssdn hot path has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.ssdn_run_ops.argtypes = [C.POINTER(OpRec), C.c_int, vp]
    lib.ssdn_run_ops.restype = C.c_int
    if hasattr(lib, "ssdn_stream_order"):        # (tools/ab_libs.sh loads older builds of the library through SSDN_HIP_LIB)
        lib.ssdn_stream_order.argtypes = [vp, vp]
        lib.ssdn_stream_order.restype = C.c_int
    lib.ssdn_conv_lds_bytes.argtypes = [C.POINTER(ConvArgs)]
    lib.ssdn_conv_lds_bytes.restype = C.c_int
    lib.ssdn_wgrad_lds_bytes.argtypes = [C.POINTER(WgradArgs)]
    lib.ssdn_wgrad_lds_bytes.restype = C.c_int
    lib.ssdn_abi_version.restype = C.c_int
    lib.ssdn_last_error.restype = C.c_char_p
    lib.ssdn_device_cus.restype = C.c_int
    lib.ssdn_probe_mfma.argtypes = [vp, vp, vp, vp]
    lib.ssdn_probe_mfma.restype = C.c_int
    lib.ssdn_probe_tr16.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.ssdn_probe_tr16.restype = C.c_int
    lib.ssdn_profile_enable.argtypes = [C.c_int, C.c_int]
    lib.ssdn_profile_enable.restype = C.c_int
    lib.ssdn_profile_set_stride.argtypes = [C.c_int, C.c_int]
    lib.ssdn_profile_set_stride.restype = C.c_int
    lib.ssdn_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ssdn_profile_read.restype = C.c_int
    lib.ssdn_conv_set_mode.argtypes = [C.c_int]
    lib.ssdn_conv_set_mode.restype = C.c_int
    lib.ssdn_conv_set_chain.argtypes = [C.c_int]
    lib.ssdn_conv_set_chain.restype = C.c_int
    lib.ssdn_chain_len.argtypes = [C.c_void_p, C.c_int]
    lib.ssdn_chain_len.restype = C.c_int
    lib.ssdn_conv_fuses_unrot.argtypes = [C.c_void_p]
    lib.ssdn_conv_fuses_unrot.restype = C.c_int
    lib.ssdn_conv_fuses_urot.argtypes = [C.c_void_p]
    lib.ssdn_conv_fuses_urot.restype = C.c_int
    lib.ssdn_conv_signs.argtypes = [C.c_void_p]
    lib.ssdn_conv_signs.restype = C.c_int
    lib.ssdn_conv_fuses_upsum.argtypes = [C.c_void_p]
    lib.ssdn_conv_fuses_upsum.restype = C.c_int
    lib.ssdn_conv_fuses_pool.argtypes = [C.c_void_p]
    lib.ssdn_conv_fuses_pool.restype = C.c_int
    lib.ssdn_wgrad_mergeable.argtypes = [C.c_void_p]
    lib.ssdn_wgrad_mergeable.restype = C.c_int
    lib.ssdn_wgrad_mega_ok.argtypes = [C.c_void_p]
    lib.ssdn_wgrad_mega_ok.restype = C.c_int
    lib.ssdn_wgrad_variant.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.ssdn_wgrad_variant.restype = C.c_int
    lib.ssdn_struct_size.argtypes = [C.c_int]
    lib.ssdn_struct_size.restype = C.c_int
    lib.ssdn_plan_load.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.ssdn_plan_load.restype = C.c_int
    lib.ssdn_plan_destroy.argtypes = [C.c_void_p]
    lib.ssdn_plan_destroy.restype = None
    lib.ssdn_plan_arena_bytes.argtypes = [C.c_void_p]
    lib.ssdn_plan_arena_bytes.restype = C.c_int64
    lib.ssdn_plan_meta.argtypes = [C.c_void_p]
    lib.ssdn_plan_meta.restype = C.c_char_p
    lib.ssdn_plan_bind.argtypes = [C.c_void_p, C.c_void_p]
    lib.ssdn_plan_bind.restype = C.c_int
    lib.ssdn_plan_tensor.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.ssdn_plan_tensor.restype = C.c_int
    lib.ssdn_plan_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ssdn_plan_run.restype = C.c_int
    lib.ssdn_plan_set_lr.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float]
    lib.ssdn_plan_set_lr.restype = C.c_int
    lib.ssdn_net_forward.argtypes = [C.c_void_p, C.c_void_p]
    lib.ssdn_net_forward.restype = C.c_int
    lib.ssdn_train_step.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    lib.ssdn_train_step.restype = C.c_int
    if lib.ssdn_abi_version() != ABI_VERSION:
        raise SsdnHipError("libssdn_hip.so ABI version mismatch")
    if os.environ.get("SSDN_CHAIN_MODE"):        # A/B aid (tools/): ssdn_conv_set_chain before any plan is made -- 0 off, 1 default, 2 see conv_chain.hip
        lib.ssdn_conv_set_chain(int(os.environ["SSDN_CHAIN_MODE"]))
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise SsdnHipError(load().ssdn_last_error().decode())
