"""ctypes binding of libzklc_mi355.so (the C ABI of include/zklc.h).

There is NO fallback: if the shared library is missing or a symbol is absent
this module raises, so a GPU box can never silently run something else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzklc_mi355.so")

_u8p = ctypes.c_void_p
_SIGS = {
    "zklc_init": (ctypes.c_int32, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32]),
    "zklc_init_priority": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]),
    "zklc_destroy": (None, [ctypes.c_void_p]),
    "zklc_strerror": (ctypes.c_char_p, [ctypes.c_int32]),
    "zklc_last_hip_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "zklc_synchronize": (ctypes.c_int32, [ctypes.c_void_p]),
    "zklc_abi_version": (ctypes.c_uint32, []),
    "zklc_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "zklc_device_alloc": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]),
    "zklc_device_free": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "zklc_device_copy": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32]),
    "zklc_ed25519_verify_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, _u8p, ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.c_uint32, _u8p]),
    "zklc_ed25519_verify_batch_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, _u8p, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_uint32, _u8p]),
    "zklc_sha512_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _u8p]),
    "zklc_sha512_batch_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.c_uint32, _u8p]),
    "zklc_gl_ntt": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]),
    "zklc_gl_ntt_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint64]),
    "zklc_gl_lde": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, _u8p,
                                     ctypes.c_uint32]),
    "zklc_gl_lde_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint64, _u8p, ctypes.c_uint32]),
    "zklc_poseidon_gl_permute": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32]),
    "zklc_poseidon_gl_permute_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32]),
    "zklc_gl_merkle_tree_words": (ctypes.c_uint64, [ctypes.c_uint32, ctypes.c_uint32]),
    "zklc_gl_merkle_commit": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.c_uint32, _u8p]),
    "zklc_gl_merkle_commit_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint64, ctypes.c_uint32,
                                                   ctypes.c_uint32, ctypes.c_uint32, _u8p]),
    "zklc_poseidon_bn254_permute": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32]),
    "zklc_poseidon_bn254_permute_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32]),
    "zklc_bn254_merkle_commit": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                                  ctypes.c_uint32, _u8p]),
    "zklc_bn254_merkle_commit_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint64, ctypes.c_uint32,
                                                      ctypes.c_uint32, ctypes.c_uint32, _u8p]),
    "zklc_bn254_g1_msm": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p]),
    "zklc_bn254_g1_msm_workspace_bytes": (ctypes.c_uint64, [ctypes.c_uint64]),
    "zklc_bn254_g1_msm_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p, _u8p,
                                               ctypes.c_uint64]),
    "zklc_bn254_g2_msm": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p]),
    "zklc_bn254_g2_msm_workspace_bytes": (ctypes.c_uint64, [ctypes.c_uint64]),
    "zklc_bn254_g2_msm_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p, _u8p,
                                               ctypes.c_uint64]),
    "zklc_bn254_g1_msm_fixed_table_bytes": (ctypes.c_uint64, [ctypes.c_uint64]),
    "zklc_bn254_g1_msm_fixed_table_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint64, _u8p, ctypes.c_uint64]),
    "zklc_bn254_g1_msm_fixed_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p, _u8p,
                                                     ctypes.c_uint64]),
    "zklc_bn254_g2_msm_fixed_table_bytes": (ctypes.c_uint64, [ctypes.c_uint64]),
    "zklc_bn254_g2_msm_fixed_table_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint64, _u8p, ctypes.c_uint64]),
    "zklc_bn254_g2_msm_fixed_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p, _u8p,
                                                     ctypes.c_uint64]),
    "zklc_bn254_pairing_check": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint32, ctypes.c_uint32, _u8p, _u8p]),
    "zklc_bn254_pairing_check_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint32, ctypes.c_uint32,
                                                      _u8p, _u8p]),
    "zklc_bn254_fr_ntt": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "zklc_bn254_fr_ntt_workspace_bytes": (ctypes.c_uint64, [ctypes.c_uint32]),
    "zklc_bn254_fr_ntt_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                               _u8p, ctypes.c_uint64]),
    "zklc_bn254_fr_mul_sub_scale_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                         _u8p, ctypes.c_uint64]),
    "zklc_plonky2_circuit_create": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32, _u8p,
                                                     _u8p, _u8p, ctypes.POINTER(ctypes.c_void_p)]),
    "zklc_plonky2_circuit_destroy": (None, [ctypes.c_void_p]),
    "zklc_plonky2_verifier_data": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p]),
    "zklc_plonky2_proof_bytes": (ctypes.c_uint64, [ctypes.c_void_p]),
    "zklc_plonky2_prove": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, _u8p, ctypes.c_uint64,
                                            ctypes.POINTER(ctypes.c_uint64)]),
    "zklc_plonky2_prove_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _u8p, _u8p, _u8p, ctypes.c_uint64,
                                                ctypes.POINTER(ctypes.c_uint64)]),
    "zklc_plonky2_last_challenges": (ctypes.c_uint32, [ctypes.c_void_p, _u8p, ctypes.c_uint32]),
    "zklc_plonky2_last_timings": (ctypes.c_uint32, [ctypes.c_void_p, _u8p, ctypes.c_uint32]),
    "zklc_poseidon_gl_gate_rows": (ctypes.c_int32, [_u8p, _u8p, ctypes.c_uint32, _u8p]),
    "zklc_plonky2_witness_release": (None, []),
    "zklc_poseidon_gl_constants": (None, [_u8p, _u8p, _u8p, _u8p, _u8p]),
    "zklc_gl_mul_vec": (None, [_u8p, _u8p, _u8p, ctypes.c_uint64]),
    "zklc_host_copy_classes": (ctypes.c_int32, [_u8p, _u8p, ctypes.c_uint64, ctypes.c_uint64, _u8p]),
    "zklc_plonky2_witness_program_create": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, _u8p, ctypes.c_uint64, ctypes.c_uint32,
                                                             _u8p, ctypes.c_uint32, _u8p, _u8p, ctypes.c_uint64, ctypes.c_uint32,
                                                             ctypes.c_uint32, _u8p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    "zklc_plonky2_witness_program_destroy": (None, [ctypes.c_void_p]),
    "zklc_plonky2_witness_program_info": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64),
                                                           ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "zklc_plonky2_witness_run_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32,
                                                      ctypes.c_void_p, _u8p, _u8p, ctypes.c_char_p]),
    "zklc_plonky2_witness_run": (ctypes.c_int32, [_u8p, ctypes.c_uint64, _u8p, ctypes.c_uint32, _u8p, ctypes.c_uint32, _u8p,
                                                  ctypes.c_uint32, _u8p, _u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _u8p,
                                                  _u8p, ctypes.c_uint32, _u8p, _u8p, ctypes.c_char_p, ctypes.c_uint32]),
    # ---- circuit container (csrc/container.cpp): host functions, no GPU
    "zklc_container_write": (ctypes.c_int32, [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32]),
    "zklc_plonky2_container_write": (ctypes.c_int32, [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_uint32, _u8p, _u8p,
                                                      _u8p, ctypes.c_void_p, _u8p, _u8p, _u8p, _u8p, _u8p, _u8p, ctypes.c_void_p,
                                                      ctypes.c_uint32]),
    "zklc_container_open": (ctypes.c_int32, [ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    "zklc_container_close": (None, [ctypes.c_void_p]),
    "zklc_container_count": (ctypes.c_uint32, [ctypes.c_void_p]),
    "zklc_container_entry_at": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "zklc_container_find": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "zklc_container_release_pages": (None, [ctypes.c_void_p]),
    "zklc_plonky2_container_params": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "zklc_plonky2_circuit_create_from_container": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                                    ctypes.POINTER(ctypes.c_void_p)]),
    "zklc_plonky2_witness_program_create_from_container": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p,
                                                                            ctypes.POINTER(ctypes.c_void_p)]),
    "zklc_plonky2_witness_run_from_container": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint32, _u8p, _u8p, _u8p, ctypes.c_char_p,
                                                                 ctypes.c_uint32]),
}

NTT_INVERSE, NTT_IN_BITREV, NTT_OUT_BITREV = 1, 2, 4

_lib = None


class ZklcError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("zklc status %d: %s%s" % (code, _strerror(code), (" [" + detail + "]") if detail else ""))


def _strerror(code):
    try:
        return load().zklc_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "?"


def load():
    """Load the shared library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libzklc_mi355.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; importing
    # torch first makes the dynamic loader resolve our NEEDED entry to that copy, so
    # torch tensors, streams and RCCL share a runtime with the zklc kernels.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch is plumbing, not a dependency of the ABI
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols():
    return sorted(_SIGS)
