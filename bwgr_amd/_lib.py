"""ctypes binding of libbwgr_hip.so (include/bwgr.h).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os
import sys

from . import build as _build

_lib = None

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)


class BwgrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bwgr status %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        raise ImportError("libbwgr_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `python -m bwgr_amd.build`); bwgr_amd has no CPU fallback")
    # PyTorch bundles its own HIP runtime; if this library loads the system one first, a LATER `import torch` in the same process
    # finds "No HIP GPUs".  A caller that uses both imports torch first (bench.py, bwgr_amd.dist and bwgr_amd.synth do), or sets
    # BWGR_PRELOAD_TORCH=1 to have it imported here.  The library itself does not need torch.
    if "torch" not in sys.modules and os.environ.get("BWGR_PRELOAD_TORCH", "0") == "1":
        try:
            import torch  # noqa: F401
        except ImportError:   # a box without torch: the library does not need it
            pass
    L = C.CDLL(path)
    L.bwgr_last_error.restype = C.c_char_p
    vp, i64, u64, u32, i32, f32, f64 = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_double
    L.bwgr_panel_create.argtypes = [C.POINTER(vp), vp, i32, i32, i64, i64, i64, i32, i32, i32]
    L.bwgr_panel_destroy.argtypes = [vp]
    L.bwgr_panel_set_stream.argtypes = [vp, vp]
    L.bwgr_panel_info.argtypes = [vp, C.POINTER(i64)]
    L.bwgr_em.argtypes = [vp, i32, c_f, f32, f32, f32, c_f, i32, C.POINTER(f32), c_f, c_f, c_f, c_f, c_f, C.POINTER(i32)]
    L.bwgr_em_order.argtypes = [i64, i32, C.POINTER(C.c_int32)]
    L.bwgr_panel_clone.argtypes = [C.POINTER(vp), vp]
    L.bwgr_panel_max_concurrent.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.bwgr_panel_max_pairs.argtypes = [vp, C.POINTER(C.c_int)]
    L.bwgr_debug_occupancy_fits.argtypes = [i32, i32, i32, i32, C.POINTER(i32)]
    L.bwgr_debug_stream3_dma.argtypes = [i64, i64]
    L.bwgr_panel_pipeline.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.bwgr_panel_stats.argtypes = [vp, c_f, c_f, c_f]
    L.bwgr_kmup.argtypes = [vp, c_f, c_f, c_f, c_f, c_f, f32, f32, u64, u32, i32]
    L.bwgr_kmup2.argtypes = [vp, C.POINTER(i32), i64, c_f, c_f, c_f, c_f, c_f, c_f, f32, f32, u64, u32, i32]
    L.bwgr_chain_create.argtypes = [C.POINTER(vp), vp, i32, vp, i32, f32, f32, f32, f32, f32, u64, i32]
    L.bwgr_chain_create_sharded.argtypes = [C.POINTER(vp), vp, i32, vp, i32, f32, f32, f32, f32, f32, u64, i32, i64, i64, f32, vp]
    L.bwgr_chain_sweep_blocks.argtypes = [vp, i32, i32]
    L.bwgr_chain_round_sweep.argtypes = [vp, i32, i32, vp]
    L.bwgr_chain_round_apply.argtypes = [vp, vp]
    L.bwgr_chain_get_sums_dev.argtypes = [vp, vp]
    L.bwgr_chain_end_iteration_dev.argtypes = [vp, vp]
    L.bwgr_chain_get_sums.argtypes = [vp, c_d]
    L.bwgr_chain_end_iteration.argtypes = [vp, c_d]
    L.bwgr_chain_destroy.argtypes = [vp]
    L.bwgr_chain_run.argtypes = [vp, i32]
    L.bwgr_chain_run_pair.argtypes = [vp, vp, i32]
    L.bwgr_chain_sync.argtypes = [vp]
    L.bwgr_chain_iterations.argtypes = [vp, C.POINTER(i32)]
    L.bwgr_chain_result.argtypes = [vp] + [c_f] * 10
    L.bwgr_chain_state.argtypes = [vp] + [c_f] * 5
    L.bwgr_chain_sweep_ms.argtypes = [vp, c_f, C.POINTER(i32)]
    L.bwgr_chain_redo_count.argtypes = [vp, C.POINTER(i32)]
    L.bwgr_group_sound.argtypes = [vp, C.POINTER(i32)]
    L.bwgr_panel_centred.argtypes = [vp, C.POINTER(i32)]
    L.bwgr_panel_set_centred.argtypes = [vp, i32]
    L.bwgr_bayes.argtypes = [vp, i32, c_f, f32, f32, f32, f32, f32, u64, i32] + [c_f] * 10
    L.bwgr_bayes2.argtypes = [vp, vp, i32, c_f, f32, f32, f32, f32, f32, u64, i32] + [c_f] * 10
    L.bwgr_wgr.argtypes = [vp, c_d, i32, i32, i32, i32, i32, f64, f64, f64, u64, i32] + [c_d] * 7
    L.bwgr_wgr_ex.argtypes = [vp, c_d, i32, i32, i32, i32, i32, f64, f64, f64, u64, i32, c_d, c_d, i64, f64, i32] + [c_d] * 9
    L.bwgr_synth_genotypes.argtypes = [vp, i64, i64, i64, i64, u64, vp, i32, vp]
    L.bwgr_debug_variates.argtypes = [i32, u64, i32, f64, u32, u32, u32, i32, c_d]
    L.bwgr_debug_withhold.argtypes = [vp, i32]
    L.bwgr_group_create.argtypes = [C.POINTER(vp), i32, C.POINTER(i32), vp, i32, i64, i64, i64, i32, c_f, i32, f32, f32, f32, f32, f32, u64, i32, i64]
    L.bwgr_group_create_centred.argtypes = list(L.bwgr_group_create.argtypes) + [i32]
    L.bwgr_group_run.argtypes = [vp, i32]
    L.bwgr_group_sync.argtypes = [vp]
    L.bwgr_group_info.argtypes = [vp, C.POINTER(i64)]
    L.bwgr_group_result.argtypes = [vp] + [c_f] * 10
    L.bwgr_group_destroy.argtypes = [vp]
    L.bwgr_device_count.argtypes = [C.POINTER(i32)]
    L.bwgr_sample_rows.argtypes = [u64, u32, i64, i64, i32, C.POINTER(i32)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise BwgrError(rc, lib().bwgr_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    lib().bwgr_device_count(C.byref(n))
    return n.value


EXPORTS = ["bwgr_abi_version", "bwgr_last_error", "bwgr_device_count", "bwgr_panel_create", "bwgr_panel_destroy",
           "bwgr_panel_set_stream", "bwgr_panel_info", "bwgr_panel_pipeline", "bwgr_panel_clone", "bwgr_em", "bwgr_em_order", "bwgr_panel_max_concurrent", "bwgr_panel_max_pairs", "bwgr_debug_occupancy_fits", "bwgr_debug_stream3_dma", "bwgr_panel_stats", "bwgr_kmup", "bwgr_kmup2", "bwgr_chain_create",
           "bwgr_chain_create_sharded", "bwgr_chain_sweep_blocks", "bwgr_chain_round_sweep", "bwgr_chain_round_apply", "bwgr_chain_get_sums_dev", "bwgr_chain_end_iteration_dev", "bwgr_chain_get_sums", "bwgr_chain_end_iteration",
           "bwgr_chain_destroy", "bwgr_chain_run", "bwgr_chain_run_pair", "bwgr_chain_sync", "bwgr_chain_iterations", "bwgr_chain_result",
           "bwgr_chain_state", "bwgr_chain_sweep_ms", "bwgr_chain_redo_count", "bwgr_group_sound", "bwgr_panel_centred", "bwgr_panel_set_centred", "bwgr_bayes", "bwgr_bayes2", "bwgr_wgr", "bwgr_wgr_ex", "bwgr_synth_genotypes",
           "bwgr_debug_variates", "bwgr_debug_withhold", "bwgr_sample_rows", "bwgr_group_create", "bwgr_group_create_centred", "bwgr_group_run", "bwgr_group_sync",
           "bwgr_group_info", "bwgr_group_result", "bwgr_group_destroy"]
