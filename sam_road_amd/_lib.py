"""ctypes binding of libsamroad_hip.so (C ABI: include/samroad_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an
exception is raised.  Nothing here imports ``oracle/``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRH_LIB_PATH: load another build of the same library (A/B measurements of kernel changes); still no CPU fallback
LIB_PATH = os.environ.get("SRH_LIB_PATH") or os.path.join(_HERE, "libsamroad_hip.so")

SRH_F32, SRH_F16, SRH_U8, SRH_I32, SRH_I64 = 0, 1, 2, 3, 4
ABI_VERSION = 8
SRH_GEMM_A_BLOCKED16, SRH_GEMM_OUT_BLOCKED16 = 1, 2       # srh_op_gemm_ex flags (include/samroad_hip.h)


class SrhError(RuntimeError):
    pass


class ModelCfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32),
                ("patch_size", C.c_int32), ("n_global", C.c_int32), ("global_attn_indexes", C.c_int32 * 8),
                ("window_size", C.c_int32), ("toponet_version", C.c_int32), ("use_sam_decoder", C.c_int32)]


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("on_device", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class ProfileRow(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/samroad_hip.h declares: (restype, argtypes)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SYMBOLS = {
    "srh_abi_version": (_I, []),
    "srh_build_id": (C.c_char_p, []),
    "srh_ctx_create": (_I, [_I, C.POINTER(_P)]),
    "srh_ctx_destroy": (None, [_P]),
    "srh_last_error": (C.c_char_p, [_P]),
    "srh_weights_pack": (_I, [_P, C.POINTER(ModelCfg), C.POINTER(NamedTensor), _I, C.POINTER(_P)]),
    "srh_weights_free": (None, [_P]),
    "srh_weights_export": (_I, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "srh_weights_import": (_I, [_P, C.POINTER(ModelCfg), _P, C.c_size_t, C.POINTER(_P)]),
    "srh_encode_decode": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "srh_toponet": (_I, [_P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P]),
    "srh_scene_pass1": (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P]),
    "srh_scene_normalise": (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "srh_op_gemm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "srh_op_gemm_ex": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "srh_ctx_device_bytes": (C.c_size_t, [_P]),
    "srh_op_conv3x3": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "srh_op_layernorm": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P]),
    "srh_op_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "srh_op_attention_hd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "srh_nms_points_host": (_I, [_P, _P, C.c_int64, C.c_int32, _P]),
    "srh_pass2_count": (_I, [_P, C.c_int64, _P, C.c_int32, _P]),
    "srh_pass2_fill": (_I, [_P, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, _P, _P, C.c_int32]),
    "srh_pass2_votes": (_I, [_P, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, C.c_int64, _P, _P, C.c_int64, _P]),
    "srh_pass2_vote_sums": (_I, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, C.c_int64, _P, _P, _P, _P, C.c_int64, _P, C.c_int32]),
    "srh_votes_to_edges": (_I, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, _P, _P]),
    "srh_pass2_pack": (_I, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P]),
    "srh_pass2_pack_ragged": (_I, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "srh_toponet_ragged": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, C.c_int64, _I, _P, _P, _P]),
    "srh_ctx_check": (_I, [_P, _P, _I]),
    "srh_kdtree_knn_host": (_I, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, C.c_double, _P, _P]),
    "srh_mask_candidates": (_I, [_P, C.c_int32, C.c_int32, C.c_float, _P, _P, C.c_int64, _P, C.c_int32]),
    "srh_nms_merge_points": (_I, [_P, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int32, _P, _P]),
    "srh_edge_vote_accumulate": (_I, [_P, _P, C.c_int64, _P, _P, _P, _P, _P]),
    "srh_edge_vote_accumulate_mt": (_I, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int32]),
    "srh_profile_enable": (_I, [_P, _I]),
    "srh_profile_read": (_I, [_P, C.POINTER(ProfileRow), _I, C.POINTER(_I)]),
    "srh_profile_overhead": (_I, [_P, _P, C.POINTER(C.c_double)]),
}

_lib = None


def build_id():
    """Build id of the loaded library (sha256 of its sources, 16 hex digits; sam_road_amd/build.py source_id)."""
    return load().srh_build_id().decode()


def load():
    """Load the shared library (raises if it has not been built — no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrhError(f"{LIB_PATH} not found: build it with `python -m sam_road_amd.build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.srh_abi_version() != ABI_VERSION:
        raise SrhError("libsamroad_hip.so ABI version mismatch")
    _lib = lib
    return lib


class Context:
    """One per (process, device).  A context is SINGLE-STREAM: its workspaces (activations, split-K partials, TopoNet
    scratch) are shared by every call, so calls on the same context must be issued on one HIP stream at a time (the
    reference is single-threaded on the default stream too, inferencer.py:93,200); use one process per GPU for more."""
    _by_device = {}

    def __init__(self, device_index):
        self.lib = load()
        h = _P()
        rc = self.lib.srh_ctx_create(int(device_index), C.byref(h))
        if rc != 0:
            raise SrhError(f"srh_ctx_create(device={device_index}) failed with status {rc} "
                           "(no MI355X visible? there is no CPU fallback)")
        self.handle = h
        self.device_index = int(device_index)

    @classmethod
    def get(cls, device_index):
        ctx = cls._by_device.get(device_index)
        if ctx is None:
            ctx = cls._by_device[device_index] = Context(device_index)
        return ctx

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.srh_last_error(self.handle)
            raise SrhError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")

    def profile_enable(self, on=True):
        self.check(self.lib.srh_profile_enable(self.handle, 1 if on else 0), "srh_profile_enable")

    def profile_overhead(self, stream=None):
        """ms an event pair adds around one launch (srh_profile_overhead)."""
        v = C.c_double(0.0)
        self.check(self.lib.srh_profile_overhead(self.handle, stream, C.byref(v)), "srh_profile_overhead")
        return v.value

    def profile_read(self):
        rows = (ProfileRow * 64)()
        n = _I(0)
        self.check(self.lib.srh_profile_read(self.handle, rows, 64, C.byref(n)), "srh_profile_read")
        return [dict(name=rows[i].name.decode(), launches=rows[i].launches, ms=rows[i].ms,
                     flops=rows[i].flops, bytes=rows[i].bytes) for i in range(n.value)]
