"""ctypes binding of liboracle_hip.so (include/oracle_hip.h). Fails loudly when the library is missing:
there is no CPU or eager-PyTorch fallback anywhere in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'liboracle_hip.so')

_lib = None


class OrcConfig(C.Structure):
    _fields_ = [
        ('vocab', C.c_int32), ('d_model', C.c_int32), ('n_layers', C.c_int32), ('n_heads', C.c_int32), ('d_ff', C.c_int32),
        ('rms_eps', C.c_float), ('rope_theta', C.c_float),
        ('clip_layers', C.c_int32), ('clip_d', C.c_int32), ('clip_heads', C.c_int32), ('clip_ff', C.c_int32),
        ('clip_image', C.c_int32), ('clip_patch', C.c_int32), ('clip_eps', C.c_float),
        ('pool_layers', C.c_int32), ('pool_heads', C.c_int32), ('pool_ff', C.c_int32), ('pool_maxpos', C.c_int32),
        ('pool_eps', C.c_float),
        ('max_batch', C.c_int32), ('max_seq', C.c_int32), ('max_images', C.c_int32), ('max_new_tokens', C.c_int32),
        ('pad_token_id', C.c_int32), ('eos_token_id', C.c_int32),
    ]


# every symbol include/oracle_hip.h declares: (restype, argtypes)
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    'orc_last_error': (C.c_char_p, []),
    'orc_abi_version': (_I, []),
    'orc_engine_create': (_I, [C.POINTER(OrcConfig), C.POINTER(_P)]),
    'orc_engine_destroy': (None, [_P]),
    'orc_engine_load_weight': (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_L), _I]),
    'orc_engine_finalize': (_I, [_P]),
    'orc_vision_encode': (_I, [_P, _P, _I, _P, _P]),
    'orc_vision_global': (_I, [_P, _P, _I, _P, _P]),
    'orc_pool_views': (_I, [_P, _P, C.POINTER(C.c_int32), _I, _P, _P]),
    'orc_project': (_I, [_P, _P, _I, _P, _P]),
    'orc_view_reduce': (_I, [_P, _P, _I, _I, _L, _I, _P, _P]),
    'orc_pool2x2': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'orc_preprocess_images': (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, C.POINTER(C.c_int32), _P, _P, _P]),
    'orc_preprocess_images_noalloc': (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, C.POINTER(C.c_int32), _P, _P]),
    'orc_preprocess_noalloc_capacity': (_L, []),
    'orc_splice': (_I, [_P, _P, _I, _P, _P, _P, _P]),
    'orc_prefill': (_I, [_P, _P, _P, C.POINTER(C.c_int32), _I, _I, _P, _P]),
    'orc_decode_step': (_I, [_P, _P, _P, _P]),
    'orc_decode_steps': (_I, [_P, _I, _P]),
    'orc_read_state': (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    'orc_set_row_limits': (_I, [_P, C.POINTER(C.c_int32), _I, _P]),
    'orc_stream_begin': (_I, [_P, _I, _P]),
    'orc_prefill_slots': (_I, [_P, C.POINTER(C.c_int32), _I, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _P, _P]),
    'orc_stream_attach': (_I, [_P, _I, _I]),
    'orc_read_slots': (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    'orc_test_gemm': (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _I, _P, _P]),
    'orc_test_rmsnorm': (_I, [_P, _P, _I, _I, _F, _P, _P]),
    'orc_test_folded_norm': (_I, [_P, _I, _I, _P, _P, _I, _F, _P, _P, _P]),
    'orc_test_layernorm': (_I, [_P, _P, _P, _I, _I, _F, _P, _P]),
    'orc_test_attention': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _F, _P, _P]),
    'orc_test_decode_attention': (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    'orc_profile_enable': (_I, [_P, _I]),
    'orc_profile_read': (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(_L)]),
    'orc_decode_weight_bytes': (_L, [_P]),
    'orc_kv_bytes_per_key': (_L, [_P]),
    'orc_decode_stream_bytes': (_L, [_P, _I]),
    'orc_decode_weight_stats': (_I, [_P, C.POINTER(_L)]),
}


class OracleHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OracleHipError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950). There is no fallback path.')
        # torch ships its own HIP runtime: load it first so this library binds to the same libamdhip64 the
        # process uses for device memory and streams (a second runtime would not see torch's allocations)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().orc_last_error()
        raise OracleHipError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')
