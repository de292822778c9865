"""ctypes loader for libbpmi.so (include/bpmi.h).  There is no CPU fallback: if the
library or a gfx950 device is missing, everything that needs it raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPMI_LIB") or os.path.join(HERE, "libbpmi.so")      # BPMI_LIB: A/B experiments with two builds in one run
NSTAGES = 15

# name -> (restype, argtypes); this table is checked against include/bpmi.h by tests
_vp, _cp, _u64, _i, _sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_size_t
SIGNATURES = {
    "bpmi_version": (_i, []),
    "bpmi_device_count": (_i, []),
    "bpmi_ctx_create": (_vp, [_i, _vp]),
    "bpmi_ctx_destroy": (None, [_vp]),
    "bpmi_last_error": (_cp, [_vp]),
    "bpmi_sync": (_i, [_vp]),
    "bpmi_set_option": (_i, [_vp, _cp, ctypes.c_int64]),
    "bpmi_malloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "bpmi_free": (_i, [_vp, _vp]),
    "bpmi_upload": (_i, [_vp, _vp, _cp, _sz]),
    "bpmi_download": (_i, [_vp, _vp, _vp, _sz]),
    "bpmi_msm": (_i, [_vp, _cp, _cp, _u64, _cp]),
    "bpmi_msm2": (_i, [_vp, _cp, _cp, _u64, _cp, _cp, _cp, _u64, _cp]),
    "bpmi_msm_dev": (_i, [_vp, _vp, _vp, _u64, _cp]),
    "bpmi_msm_dev_enqueue": (_i, [_vp, _i, _vp, _vp, _u64]),
    "bpmi_msm_finish": (_i, [_vp, _i, _cp]),
    "bpmi_msm_geometry": (_i, [_vp, _u64, _i, ctypes.POINTER(ctypes.c_uint32)]),
    "bpmi_ec_mul_batch": (_i, [_vp, _cp, _cp, _u64, _cp]),
    "bpmi_ec_mul_batch_dev": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "bpmi_ec_lincomb2_batch": (_i, [_vp, _cp, _cp, _cp, _cp, _u64, _cp]),
    "bpmi_ec_lincomb2_batch_dev": (_i, [_vp, _vp, _vp, _cp, _cp, _u64, _vp]),
    "bpmi_ec_sum": (_i, [_vp, _cp, _u64, _cp]),
    "bpmi_ec_sum_dev": (_i, [_vp, _vp, _u64, _cp]),
    "bpmi_ec_sum_dev_enqueue": (_i, [_vp, _vp, _u64, _vp]),
    "bpmi_ec_decompress_batch": (_i, [_vp, _cp, _u64, _cp, _cp]),
    "bpmi_ec_decompress_batch_dev": (_i, [_vp, _vp, _u64, _vp, _vp]),
    "bpmi_memcpy_dev": (_i, [_vp, _vp, _vp, _sz]),
    "bpmi_msm_segs_dev": (_i, [_vp, ctypes.c_uint32, _vp, _vp, _vp, _cp]),
    "bpmi_sc_dot": (_i, [_vp, _cp, _cp, _u64, _cp]),
    "bpmi_sc_dot_dev": (_i, [_vp, _vp, _vp, _u64, _cp]),
    "bpmi_sc_fold": (_i, [_vp, _cp, _cp, _cp, _cp, _u64, _cp]),
    "bpmi_sc_fold_dev": (_i, [_vp, _vp, _vp, _cp, _cp, _u64, _vp]),
    "bpmi_sc_svector": (_i, [_vp, _cp, _cp, ctypes.c_uint32, _cp, _cp, _cp, _cp, _cp]),
    "bpmi_ipa_verify_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _cp, _cp, ctypes.c_uint32, _cp, _cp, _cp, _cp, _u64, _cp]),
    "bpmi_ipa_create": (_i, [_vp, _cp, _cp, _cp, _cp, _u64, _cp, ctypes.POINTER(_vp)]),
    "bpmi_ipa_create_scaled": (_i, [_vp, _cp, _cp, _cp, _cp, _u64, _cp, _cp, ctypes.POINTER(_vp)]),
    "bpmi_ipa_create_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _u64, _cp, ctypes.POINTER(_vp)]),
    "bpmi_ipa_len": (_u64, [_vp]),
    "bpmi_ipa_round_LR": (_i, [_vp, _cp, _cp]),
    "bpmi_ipa_fold": (_i, [_vp, _cp, _cp]),
    "bpmi_ipa_finish": (_i, [_vp, _cp, _cp]),
    "bpmi_ipa_export": (_i, [_vp, _cp, _cp, _cp, _cp]),
    "bpmi_rp_batch_prepare": (_i, [ctypes.c_uint32, ctypes.c_uint32, _u64, _vp, _u64, ctypes.c_void_p, _cp, _cp, _i, _cp, _cp, _cp, _cp, ctypes.c_void_p]),
    "bpmi_mod_hash_range": (_i, [_cp, _u64, _u64, _u64, _i, _cp]),
    "bpmi_rp_poly_coeffs": (_i, [ctypes.c_uint32, ctypes.c_uint32, _i, _cp, _cp, _cp, _cp, _cp, _i, _cp, _cp]),
    "bpmi_rp_final_vectors": (_i, [ctypes.c_uint32, ctypes.c_uint32, _i, _cp, _cp, _cp, _cp, _cp, _cp, _i, _cp, _cp, _cp, _cp, _cp]),
    "bpmi_rp_verifier_vectors": (_i, [ctypes.c_uint32, ctypes.c_uint32, _i, _cp, _cp, _i, _cp, _cp, _cp]),
    "bpmi_rp_wire_v2_to_v1": (_i, [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp]),
    "bpmi_rp_batch_prepare_dev": (_i, [_vp, ctypes.c_uint32, ctypes.c_uint32, _u64, _vp, _u64, _vp, _cp, _cp, _vp, _vp, _vp, _cp, _vp]),
    "bpmi_rp_batch_verify_dev": (_i, [_vp, ctypes.c_uint32, ctypes.c_uint32, _u64, _vp, _u64, _vp, _cp, _cp, _vp, _vp, _vp, _vp, _cp, _vp]),      # (v_points: bytes or a page-locked address)
    "bpmi_rp_prover_create": (_i, [_vp, ctypes.c_uint32, _cp, _cp, _cp, _cp, _cp, ctypes.POINTER(_vp)]),
    "bpmi_rp_prover_create_aggregated": (_i, [_vp, ctypes.c_uint32, ctypes.c_uint32, _cp, _cp, _cp, _cp, _cp, ctypes.POINTER(_vp)]),
    "bpmi_rp_prover_destroy": (None, [_vp]),
    "bpmi_rp_prove_batch_proof_bytes": (_u64, [_vp, _u64]),
    "bpmi_rp_prove_batch": (_i, [_vp, _u64, _cp, _cp, _cp, _vp, _vp, _u64, _vp]),
    "bpmi_rp_prover_last_ms": (_i, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "bpmi_host_alloc": (_i, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "bpmi_host_free": (_i, [_vp, _vp]),
    "bpmi_ipa_destroy": (None, [_vp]),
    "bpmi_ipa_prove_rounds": (_i, [_vp, _cp, _u64, _vp, _u64, _vp, _vp, _vp, _vp, ctypes.c_uint32, _vp]),
    "bpmi_debug_fe_op": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _u64, _vp]),
    "bpmi_debug_quad_add": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "bpmi_profile": (_i, [_vp, _i]),
    "bpmi_profile_reset": (_i, [_vp]),
    "bpmi_profile_read": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_u64)]),
    "bpmi_profile_stage_name": (_cp, [_i]),
}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load libbpmi.so and declare every entry point; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "libbpmi.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for the MSM / IPA path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
