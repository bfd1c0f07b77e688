"""ctypes binding of libsml_hip.so (C ABI declared in include/sml_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared
object is missing or cannot be loaded this module raises, and so does every
caller (sml_amd.engine, model.*, evalution.*).
"""
import ctypes
import os

import torch  # noqa: F401  -- must come first: it loads the HIP runtime (libamdhip64.so.7) this library binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsml_hip.so")

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_void = ctypes.c_void_p

GRAD_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64)
MF_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64)

LOSS_BCE, LOSS_BPR, LOSS_BPR_NORM, LOSS_BPR_UNIT = 0, 1, 2, 3


class MFTables(ctypes.Structure):
    _fields_ = [("w_user", c_void), ("w_item", c_void), ("last_user", c_void), ("last_item", c_void),
                ("m_user", c_void), ("v_user", c_void), ("m_item", c_void), ("v_item", c_void),
                ("step_user", c_void), ("step_item", c_void), ("n_user", ctypes.c_int64), ("n_item", ctypes.c_int64)]


class MFExchange(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int), ("key_items", c_void), ("val_items", c_void), ("dx_local", c_void),
                ("dx_items_all", c_void), ("hook", MF_HOOK), ("hook_user", c_void), ("loss_scale", ctypes.c_float),
                ("slot_stride", ctypes.c_int64), ("item_off", c_void), ("push_rows", ctypes.c_int64), ("lists_unsorted", ctypes.c_int)]


class BareExchange(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int), ("items_all", c_void), ("dx_local", c_void), ("dx_items_all", c_void), ("hook", MF_HOOK),
                ("hook_user", c_void), ("loss_scale", ctypes.c_float)]


class BareShard(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int), ("rank", ctypes.c_int), ("head_rows", ctypes.c_int64), ("shard_rows", ctypes.c_int64),
                ("item_shard", ctypes.POINTER(c_void)), ("w_item_head", c_void), ("items_all", c_void), ("loss_scale", ctypes.c_float)]


class BatchPlan(ctypes.Structure):
    _fields_ = [("n_batches", ctypes.c_int64), ("batch_off", c_void), ("batch_off_dev", c_void), ("loss_scale", c_void)]


class TRTables(ctypes.Structure):
    _fields_ = [("last_user", c_void), ("last_item", c_void), ("hat_user", c_void), ("hat_item", c_void),
                ("n_user", ctypes.c_int64), ("n_item", ctypes.c_int64)]


# every exported symbol of include/sml_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    "sml_last_error": (ctypes.c_char_p, []),
    "sml_version": (ctypes.c_int, []),
    "sml_ctx_create": (ctypes.c_int, [ctypes.POINTER(c_void), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sml_ctx_set_variant": (ctypes.c_int, [c_void, ctypes.c_int]),
    "sml_ctx_set_grad_clip": (ctypes.c_int, [c_void, ctypes.c_float]),
    "sml_ctx_set_adaptive": (ctypes.c_int, [c_void, ctypes.c_float]),
    "sml_ctx_destroy": (ctypes.c_int, [c_void]),
    "sml_theta_net_size": (ctypes.c_int64, [ctypes.c_int]),
    "sml_theta_offset": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "sml_theta_pack": (ctypes.c_int, [c_void, c_void, c_void]),
    "sml_transfer_forward": (ctypes.c_int, [c_void, c_void, ctypes.c_int, c_void, c_void, c_void, ctypes.c_int64, c_void]),
    "sml_mf_stage_epoch": (ctypes.c_int, [c_void, c_void, ctypes.POINTER(MFTables), c_void, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                          c_void, ctypes.POINTER(MFExchange), ctypes.POINTER(BatchPlan), c_void]),
    "sml_mf_adam_flush": (ctypes.c_int, [c_void, ctypes.POINTER(MFTables), ctypes.c_float, ctypes.c_int64, c_void]),
    "sml_tr_stage_epoch": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, ctypes.POINTER(TRTables), c_void,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                          ctypes.c_float, ctypes.POINTER(ctypes.c_int64), c_void, GRAD_HOOK, c_void,
                                          ctypes.POINTER(BatchPlan), c_void]),
    "sml_run_mf_grad": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void,
                                       c_void, c_void]),
    "sml_embed_loss_sgd_epoch": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                                c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_int, c_void, ctypes.c_int, ctypes.POINTER(BareExchange), c_void]),
    "sml_embed_loss_sgd_epoch_sharded": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, c_void, ctypes.c_int64,
                                                        ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_void,
                                                        ctypes.POINTER(BareShard), c_void]),
    "sml_embed_loss_adam_epoch": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_int, c_void, c_void, c_void]),
    "sml_index_lists_read": (ctypes.c_int64, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int64]),
    "sml_embed_loss_sgd_prepare": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_int, ctypes.POINTER(BareExchange), c_void]),
    "sml_mf_forward": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int, c_void,
                                      c_void, c_void, c_void]),
    "sml_eval_ranks": (ctypes.c_int, [c_void, c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int, c_void, c_void]),
    "sml_eval_metrics": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, ctypes.c_int, c_void, c_void]),
    "sml_eval_prepare": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, c_void, c_void, c_void]),
    "sml_eval_ranks_blocked": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int, c_void, ctypes.c_int,
                                              c_void]),
    "sml_eval_sliced_slices": (ctypes.c_int, [c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]),
    "sml_eval_sliced_entries": (ctypes.c_int64, [c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]),
    "sml_eval_sliced_work_ints": (ctypes.c_int64, [c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]),
    "sml_eval_sliced_scratch_bytes": (ctypes.c_int64, [c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]),
    "sml_eval_prepare_sliced": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, c_void, c_void, c_void, c_void]),
    "sml_eval_ranks_sliced": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                             c_void, c_void, ctypes.c_int, c_void]),
    "sml_stream_create_cu_range": (ctypes.c_int, [ctypes.POINTER(c_void), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sml_stream_destroy": (ctypes.c_int, [c_void]),
    "sml_stream_wait_stream": (ctypes.c_int, [c_void, c_void]),
    "sml_copy_tables": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_void), ctypes.POINTER(c_void), ctypes.POINTER(ctypes.c_int64), c_void]),
    "sml_flag_set": (ctypes.c_int, [c_void, ctypes.c_int, c_void]),
    "sml_flag_wait": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_double, c_void]),
    "sml_comm_load": (ctypes.c_int, [ctypes.c_char_p]),
    "sml_comm_unique_id": (ctypes.c_int, [c_void]),
    "sml_comm_init": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void]),
    "sml_comm_destroy": (ctypes.c_int, [c_void]),
    "sml_comm_allreduce": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, c_void]),
    "sml_comm_allgather": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int64, c_void]),
    "sml_peer_region_bytes": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "sml_peer_alloc": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.POINTER(c_void)]),
    "sml_peer_free": (ctypes.c_int, [ctypes.c_int, c_void]),
    "sml_device_epoch": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_uint64, c_void, c_void]),
    "sml_peer_mem_kind": (ctypes.c_int, [c_void]),
    "sml_peer_read": (ctypes.c_int, [ctypes.c_int, c_void, c_void, ctypes.c_int64, c_void]),
    "sml_peer_export": (ctypes.c_int, [c_void, c_void]),
    "sml_peer_open": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.POINTER(c_void)]),
    "sml_peer_close": (ctypes.c_int, [ctypes.c_int, c_void]),
    "sml_peer_attach": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_void), ctypes.POINTER(c_void), ctypes.c_int64,
                                       ctypes.c_double]),
    "sml_peer_detach": (ctypes.c_int, [c_void]),
    "sml_peer_status": (ctypes.c_int, [c_void, ctypes.POINTER(ctypes.c_int)]),
    "sml_peer_allreduce_check": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int64, ctypes.c_double, c_void]),
    "sml_prof_enable": (ctypes.c_int, [c_void, ctypes.c_int]),
    "sml_debug_timeline": (ctypes.c_int, [c_void]),
    "sml_prof_reset": (ctypes.c_int, [c_void]),
    "sml_prof_pair_overhead": (ctypes.c_int, [c_void, ctypes.c_int, c_void, ctypes.POINTER(ctypes.c_double)]),
    "sml_prof_classes": (ctypes.c_int, []),
    "sml_prof_name": (ctypes.c_char_p, [ctypes.c_int]),
    "sml_prof_get": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]),
    "sml_sample_negatives": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, c_void, ctypes.c_int64, c_void, ctypes.c_int64, c_void,
                                            ctypes.c_uint64, c_void, c_void, c_void]),
    "sml_host_resolve_negatives": (ctypes.c_int, [c_void, ctypes.c_int64, c_void, ctypes.c_int64, c_void, ctypes.c_int64,
                                                  ctypes.c_int64, c_void, ctypes.POINTER(ctypes.c_int64),
                                                  ctypes.POINTER(ctypes.c_int64)]),
    "sml_host_gather_pairs": (ctypes.c_int, [c_void, ctypes.c_int64, c_void, ctypes.c_int64, c_void]),
    "sml_host_gather_column": (ctypes.c_int, [c_void, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, c_void, ctypes.c_int64,
                               c_void, ctypes.c_int]),
    "sml_host_resolve_negatives_csr": (ctypes.c_int, [c_void, ctypes.c_int64, c_void, ctypes.c_int64, c_void, ctypes.c_int64, c_void,
                                                      c_void, c_void, c_void]),
    "sml_selftest": (ctypes.c_int, [ctypes.c_int]),
}

_lib = None


def load():
    """Load libsml_hip.so (once). Raises RuntimeError if it is missing or unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsml_hip.so not found at %s: build it with `python -m sml_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("cannot load %s: %s" % (LIB_PATH, e))
    _bind(lib)
    _lib = lib
    return lib


def _bind(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


def load_other(path):
    """Another build of the same C ABI (tests: the library compiled WITH the library-sort reference of the index
    preparation, tests/build_reference.py).  Never used by the product: HipEngine(lib=...) is a test hook."""
    return _bind(ctypes.CDLL(path))


class SmlError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().sml_last_error()
        raise SmlError("%s failed (%d): %s" % (what, rc, msg.decode(errors="replace") if msg else ""))
