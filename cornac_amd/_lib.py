"""ctypes binding of libcornac_hip.so (the C ABI declared in include/cornac_hip.h).

There is deliberately NO fallback: if the shared library is missing, or no
gfx950 device is visible when a kernel is requested, the product path raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CORNAC_HIP_PROFILE=1 loads the -DCORNAC_PROFILE build of the same sources (A/B switches and ablation bits compiled
# in; `make -C cornac_amd/csrc PROFILE=1`): tools/ only — the tests, bench.py and smoke() use the shipped library
PROFILE = os.environ.get("CORNAC_HIP_PROFILE", "") == "1"
LIB_PATH = os.path.join(_HERE, "lib", "libcornac_hip_profile.so" if PROFILE else "libcornac_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MODE_DETERMINISTIC = 0
MODE_HOGWILD = 1
VEBPR_NO_OWNERSHIP = 0x100
# hogwild_flags bits 16..19: the form of a whole-epoch hogwild call (include/cornac_hip.h)
FORM_AUTO, FORM_FUSED, FORM_STRATA, FORM_LDSBIN = 0, 1 << 16, 2 << 16, 3 << 16
NEG_UNIFORM = 0
NEG_POPULARITY = 1

# every symbol include/cornac_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cornac_hip_last_error", "cornac_hip_version", "cornac_hip_device_count", "cornac_hip_device_info",
    "cornac_hip_device_probe",
    "cornac_hip_bpr_create", "cornac_hip_bpr_destroy", "cornac_hip_bpr_set_factors", "cornac_hip_bpr_get_factors",
    "cornac_hip_bpr_bind_device", "cornac_hip_bpr_rebind_items", "cornac_hip_bpr_conveyor_setup", "cornac_hip_bpr_conveyor_layout", "cornac_hip_bpr_conveyor_enqueue", "cornac_hip_bpr_set_negative_population", "cornac_hip_bpr_device_ptrs", "cornac_hip_bpr_set_stream",
    "cornac_hip_bpr_seed_mt19937", "cornac_hip_bpr_seed_hogwild", "cornac_hip_bpr_fit_epochs",
    "cornac_hip_bpr_set_factors_f64", "cornac_hip_bpr_get_factors_f64", "cornac_hip_bpr_fit_epochs_f64",
    "cornac_hip_bpr_hogwild_enqueue", "cornac_hip_bpr_sync", "cornac_hip_bpr_debug_draw",
    "cornac_hip_bpr_last_timing", "cornac_hip_bpr_kernel_timing", "cornac_hip_mf_kernel_timing",
    "cornac_hip_bpr_debug_ownership", "cornac_hip_bpr_set_views", "cornac_hip_bpr_seed_view_stream",
    "cornac_hip_vebpr_fit_epochs", "cornac_hip_vebpr_fit_epochs_f64", "cornac_hip_vebpr_hogwild_form",
    "cornac_hip_bpr_strata_config", "cornac_hip_bpr_chunk_records", "cornac_hip_bpr_strata_stats", "cornac_hip_bpr_debug_strata",
    "cornac_hip_bpr_ldsbin_config", "cornac_hip_bpr_ldsbin_pass_config", "cornac_hip_bpr_ldsbin_stats",
    "cornac_hip_bpr_ldsbin_deal_config", "cornac_hip_bpr_debug_ldsbin_deal",
    "cornac_hip_bpr_sample_triplets", "cornac_hip_bpr_apply_triplets", "cornac_hip_bpr_gather_rows",
    "cornac_hip_bpr_staged_slots", "cornac_hip_bpr_emit_triplets", "cornac_hip_bpr_apply_staged",
    "cornac_hip_bpr_shard_mark", "cornac_hip_bpr_shard_slots", "cornac_hip_bpr_shard_uniq",
    "cornac_hip_bpr_scatter_diff_rows", "cornac_hip_bpr_switch_stream",
    "cornac_hip_bpr_scatter_add_rows", "cornac_hip_bpr_table_delta_begin", "cornac_hip_bpr_table_delta_finish",
    "cornac_hip_bpr_table_delta_step", "cornac_hip_table_delta", "cornac_hip_bpr_table_delta",
    "cornac_hip_bpr_resident_exchange_bins", "cornac_hip_bpr_epoch_resident_enqueue", "cornac_hip_bpr_resident_flush",
    "cornac_hip_stream_wait_counter", "cornac_hip_stream_set_flag", "cornac_hip_stream_ring_standin",
    "cornac_hip_vbpr_create", "cornac_hip_vbpr_destroy", "cornac_hip_vbpr_set_params", "cornac_hip_vbpr_get_params",
    "cornac_hip_vbpr_fit_batches", "cornac_hip_vbpr_item_tables",
    "cornac_hip_wmf_create", "cornac_hip_wmf_destroy", "cornac_hip_wmf_set_factors", "cornac_hip_wmf_get_factors",
    "cornac_hip_wmf_fit_batches", "cornac_hip_wmf_kernel_timing", "cornac_hip_wmf_last_timing",
    "cornac_hip_mf_create", "cornac_hip_mf_destroy", "cornac_hip_mf_set_factors", "cornac_hip_mf_get_factors",
    "cornac_hip_mf_fit", "cornac_hip_mf_bind_items", "cornac_hip_mf_bind_users", "cornac_hip_mf_set_stream", "cornac_hip_mf_epoch_enqueue",
    "cornac_hip_mf_sync", "cornac_hip_mf_fit_sgd", "cornac_hip_mf_last_timing",
    "cornac_hip_mf_hogwild_form", "cornac_hip_mf_hogwild_stats",
    "cornac_hip_mf_fit_minibatch", "cornac_hip_mf_fit_minibatch_dropout", "cornac_hip_mf_reset_optimizer",
    "cornac_hip_scorer_create", "cornac_hip_scorer_destroy", "cornac_hip_scorer_set", "cornac_hip_score_user",
    "cornac_hip_scorer_set_f64", "cornac_hip_score_user_f64",
    "cornac_hip_score_block", "cornac_hip_rank_topk", "cornac_hip_rank_topk_device", "cornac_hip_score_pairs",
    "cornac_hip_scorer_set_exclusions", "cornac_hip_rank_topk_resident", "cornac_hip_scorer_host_buffer",
    "cornac_hip_rank_positions",
]


class HipError(RuntimeError):
    """A libcornac_hip call returned a non-zero status."""

    def __init__(self, code, msg):
        super().__init__("libcornac_hip error %d: %s" % (code, msg))
        self.code = code


def build(force=False, verbose=False):
    """Compile libcornac_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", "4"] + (["-B"] if force else []) + (["PROFILE=1"] if PROFILE else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building libcornac_hip.so failed")
    return LIB_PATH


_lib = None

_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_vp = C.c_void_p


def _preload_torch_hip_runtime():
    """One process must use ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so; if
    torch is imported after this library has pulled in /opt/rocm's copy, torch finds no GPUs (and the other
    way round works).  So when a torch wheel with a bundled runtime is installed, load that copy first
    (by path, without importing torch); the multi-GPU driver (torch.distributed + this library in one
    process) then shares it."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    except Exception:
        return
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load the shared library (raises if it has not been built — no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libcornac_hip.so not found at %s. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C cornac_amd/csrc`. cornac_amd has no CPU fallback." % LIB_PATH)
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.cornac_hip_last_error.restype = C.c_char_p
        L.cornac_hip_version.restype = C.c_char_p
        L.cornac_hip_device_count.argtypes = [C.POINTER(C.c_int)]
        L.cornac_hip_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.cornac_hip_device_probe.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_double)]
        L.cornac_hip_bpr_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                            C.c_int, _i32, _i32, C.c_int64]
        L.cornac_hip_bpr_destroy.argtypes = [_vp]
        L.cornac_hip_bpr_set_factors.argtypes = [_vp, _vp, _vp, _vp]
        L.cornac_hip_bpr_get_factors.argtypes = [_vp, _vp, _vp, _vp]
        L.cornac_hip_bpr_bind_device.argtypes = [_vp, _vp, _vp, _vp]
        L.cornac_hip_bpr_rebind_items.argtypes = [_vp, _vp, _vp]
        L.cornac_hip_bpr_conveyor_setup.argtypes = [_vp, C.c_int, _vp, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                    C.POINTER(C.c_int)]
        L.cornac_hip_bpr_conveyor_layout.argtypes = [_vp, C.c_uint32, _vp, _vp]
        L.cornac_hip_bpr_conveyor_enqueue.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int32), C.POINTER(_vp),
                                                      C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        L.cornac_hip_bpr_set_negative_population.argtypes = [_vp, _vp, C.c_int64]
        L.cornac_hip_bpr_device_ptrs.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]
        L.cornac_hip_bpr_set_stream.argtypes = [_vp, _vp]
        L.cornac_hip_bpr_switch_stream.argtypes = [_vp, _vp]
        L.cornac_hip_bpr_seed_mt19937.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_int]
        L.cornac_hip_bpr_seed_hogwild.argtypes = [_vp, C.c_uint64]
        L.cornac_hip_bpr_fit_epochs.argtypes = [_vp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_set_factors_f64.argtypes = [_vp, _vp, _vp, _vp]
        L.cornac_hip_bpr_get_factors_f64.argtypes = [_vp, _vp, _vp, _vp]
        L.cornac_hip_bpr_fit_epochs_f64.argtypes = [_vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_hogwild_enqueue.argtypes = [_vp, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        L.cornac_hip_bpr_sync.argtypes = [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_debug_draw.argtypes = [_vp, C.c_int, C.c_uint64, C.c_int64, _i64]
        L.cornac_hip_bpr_last_timing.argtypes = [_vp, C.POINTER(C.c_double)]
        L.cornac_hip_bpr_set_views.argtypes = [_vp, _i32, _i32, C.c_int64]
        L.cornac_hip_bpr_seed_view_stream.argtypes = [_vp, C.c_uint32]
        L.cornac_hip_vebpr_hogwild_form.argtypes = [_vp, C.POINTER(C.c_int)]
        L.cornac_hip_vebpr_fit_epochs.argtypes = [_vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cornac_hip_vebpr_fit_epochs_f64.argtypes = [_vp, C.c_int, C.c_double, C.c_double, C.c_double,
                                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_sample_triplets.argtypes = [_vp, C.c_int64, C.c_int, _vp, _vp, _vp]
        L.cornac_hip_bpr_apply_triplets.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int, C.c_float,
                                                    C.c_float, C.c_int]
        L.cornac_hip_bpr_gather_rows.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, _vp]
        L.cornac_hip_bpr_staged_slots.argtypes = [_vp, C.c_int64, C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_emit_triplets.argtypes = [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_int64, C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_apply_staged.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int, C.c_float, C.c_float,
                                                  C.c_int]
        L.cornac_hip_bpr_shard_mark.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp]
        L.cornac_hip_bpr_shard_slots.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp, _vp, _vp]
        L.cornac_hip_bpr_shard_uniq.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int64, _vp]
        L.cornac_hip_bpr_scatter_diff_rows.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp]
        L.cornac_hip_bpr_scatter_add_rows.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, _vp]
        L.cornac_hip_bpr_table_delta_begin.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp]
        L.cornac_hip_bpr_table_delta_finish.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int]
        L.cornac_hip_bpr_table_delta_step.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp]
        L.cornac_hip_table_delta.argtypes = [C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp]
        L.cornac_hip_bpr_table_delta.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp]
        L.cornac_hip_bpr_resident_exchange_bins.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.cornac_hip_bpr_epoch_resident_enqueue.argtypes = [_vp, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                                            C.c_int, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, _vp,
                                                            C.POINTER(C.c_int)]
        L.cornac_hip_bpr_resident_flush.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp]
        L.cornac_hip_stream_wait_counter.argtypes = [C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_int]
        L.cornac_hip_stream_set_flag.argtypes = [C.c_int, _vp, _vp, C.c_uint32, _vp]
        L.cornac_hip_stream_ring_standin.argtypes = [C.c_int, _vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int]
        L.cornac_hip_bpr_debug_ownership.argtypes = [_vp, C.POINTER(C.c_int64), _vp, _vp, _vp]
        L.cornac_hip_bpr_strata_config.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
        L.cornac_hip_bpr_chunk_records.argtypes = [_vp, C.c_int]
        L.cornac_hip_bpr_strata_stats.argtypes = [_vp, C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_debug_strata.argtypes = [_vp, C.c_uint32, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32)]
        L.cornac_hip_bpr_ldsbin_config.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
        L.cornac_hip_bpr_ldsbin_pass_config.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.cornac_hip_bpr_ldsbin_stats.argtypes = [_vp, C.POINTER(C.c_int64)]
        L.cornac_hip_bpr_ldsbin_deal_config.argtypes = [_vp, C.c_int, C.c_int]
        L.cornac_hip_bpr_debug_ldsbin_deal.argtypes = [_vp, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp]
        L.cornac_hip_bpr_kernel_timing.argtypes = [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.cornac_hip_mf_kernel_timing.argtypes = [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.cornac_hip_vbpr_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _f32]
        L.cornac_hip_vbpr_destroy.argtypes = [_vp]
        L.cornac_hip_vbpr_set_params.argtypes = [_vp] + [_vp] * 6
        L.cornac_hip_vbpr_get_params.argtypes = [_vp] + [_vp] * 6
        L.cornac_hip_vbpr_fit_batches.argtypes = [_vp, _i32, _i32, _i32, C.c_int64, C.c_int, C.c_float, C.c_float,
                                                  C.c_float, C.c_float, C.POINTER(C.c_double)]
        L.cornac_hip_vbpr_item_tables.argtypes = [_vp, _f32, _f32]
        L.cornac_hip_mf_fit_minibatch.argtypes = [_vp, _i64, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float,
                                                  C.c_float, C.c_int, C.POINTER(C.c_double)]
        L.cornac_hip_mf_fit_minibatch_dropout.argtypes = [_vp, _i64, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float,
                                                          C.c_float, C.c_int, _vp, _vp, C.c_float, C.POINTER(C.c_double)]
        L.cornac_hip_mf_reset_optimizer.argtypes = [_vp]
        L.cornac_hip_wmf_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int64, C.c_int64, C.c_int, _i64, _i32, _f32,
                                            C.c_int64]
        L.cornac_hip_wmf_destroy.argtypes = [_vp]
        L.cornac_hip_wmf_destroy.restype = None
        L.cornac_hip_wmf_set_factors.argtypes = [_vp, _f32, _f32]
        L.cornac_hip_wmf_get_factors.argtypes = [_vp, _f32, _f32]
        L.cornac_hip_wmf_fit_batches.argtypes = [_vp, _i32, _i64, C.c_int64, C.c_float, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, _vp]
        L.cornac_hip_wmf_kernel_timing.argtypes = [_vp, C.c_int]
        L.cornac_hip_wmf_last_timing.argtypes = [_vp, C.POINTER(C.c_double)]
        L.cornac_hip_mf_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int64, C.c_int64, C.c_int, _i64, _i64, _f32,
                                           C.c_int64]
        L.cornac_hip_mf_destroy.argtypes = [_vp]
        L.cornac_hip_mf_set_factors.argtypes = [_vp, _vp, _vp, _vp, _vp]
        L.cornac_hip_mf_get_factors.argtypes = [_vp, _vp, _vp, _vp, _vp]
        L.cornac_hip_mf_bind_items.argtypes = [_vp, _vp, _vp]
        L.cornac_hip_mf_bind_users.argtypes = [_vp, _vp, _vp]
        L.cornac_hip_mf_set_stream.argtypes = [_vp, _vp]
        L.cornac_hip_mf_epoch_enqueue.argtypes = [_vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
        L.cornac_hip_mf_sync.argtypes = [_vp, C.POINTER(C.c_double)]
        L.cornac_hip_mf_fit.argtypes = [_vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _vp,
                                        C.POINTER(C.c_int)]
        L.cornac_hip_mf_fit_sgd.argtypes = [C.c_int, _i64, _i64, _f32, C.c_int64, _f32, _f32, _f32, _f32, C.c_int64,
                                            C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                            C.c_int, C.c_int, _vp, C.POINTER(C.c_int)]
        L.cornac_hip_mf_last_timing.argtypes = [_vp, C.POINTER(C.c_double)]
        L.cornac_hip_mf_hogwild_form.argtypes = [_vp, C.c_int]
        L.cornac_hip_mf_hogwild_stats.argtypes = [_vp, C.POINTER(C.c_int64)]
        L.cornac_hip_scorer_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int64, C.c_int64, C.c_int]
        L.cornac_hip_scorer_destroy.argtypes = [_vp]
        L.cornac_hip_scorer_set.argtypes = [_vp, _f32, _f32, _vp, _vp]
        L.cornac_hip_score_user.argtypes = [_vp, C.c_int64, _f32]
        L.cornac_hip_scorer_set_f64.argtypes = [_vp, _vp, _vp, _vp, _vp]
        L.cornac_hip_score_user_f64.argtypes = [_vp, C.c_int64, _vp]
        L.cornac_hip_score_block.argtypes = [_vp, _i32, C.c_int64, _f32]
        L.cornac_hip_score_pairs.argtypes = [_vp, _i32, _i32, C.c_int64, C.c_int, C.c_float, C.c_float, _f32]
        L.cornac_hip_rank_topk.argtypes = [_vp, _i32, C.c_int64, C.c_int, _vp, _vp, _i32, _f32]
        L.cornac_hip_rank_positions.argtypes = [_vp, _i32, C.c_int64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32]
        L.cornac_hip_rank_topk_device.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.cornac_hip_scorer_set_exclusions.argtypes = [_vp, _vp, _vp]
        L.cornac_hip_scorer_host_buffer.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
        L.cornac_hip_rank_topk_resident.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp, _vp]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise HipError(rc, lib().cornac_hip_last_error().decode("utf-8", "replace"))


def stream_wait_counter(device, stream, d_counter, target, d_error, timeout_ms=5000):
    """enqueue on `stream` (a hipStream_t as int): wait until *d_counter >= target; after timeout_ms *d_error = 1"""
    check(lib().cornac_hip_stream_wait_counter(int(device), stream, d_counter, int(target), d_error, int(timeout_ms)))


def stream_ring_standin(device, stream, d_src, src_floats, d_dst, dst_floats, n_floats, n_workgroups=16):
    """what a ring all-reduce would move through this rank, as n_workgroups workgroups on `stream` (measurement aid)"""
    check(lib().cornac_hip_stream_ring_standin(int(device), stream, d_src, int(src_floats), d_dst, int(dst_floats),
                                               int(n_floats), int(n_workgroups)))


def stream_set_flag(device, stream, d_flag, value=1, d_unless=None):
    """*d_flag = value on `stream` — unless d_unless is given and *d_unless != 0 (an earlier stream_wait_counter gave up)"""
    check(lib().cornac_hip_stream_set_flag(int(device), stream, d_flag, int(value), d_unless))


def device_count():
    n = C.c_int(0)
    check(lib().cornac_hip_device_count(C.byref(n)))
    return n.value


def device_probe(device=0, nbytes=6 << 30):
    """memory-system calibration of this box: GB/s of a device-to-device copy, a streaming read and random 512-byte
    row gathers over buffers of nbytes"""
    o = (C.c_double * 3)()
    check(lib().cornac_hip_device_probe(device, int(nbytes), o))
    return {"d2d_copy_GBps": o[0], "stream_read_GBps": o[1], "row_gather_512B_GBps": o[2], "buffer_bytes": int(nbytes)}


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus = C.c_int()
    mem = C.c_int64()
    check(lib().cornac_hip_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f32c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class BprTrainer:
    """Thin owner of a cornac_hip_bpr_t handle."""

    def __init__(self, indptr, indices, n_users, n_items, total_users, total_items, k, device=0):
        self.indptr = np.ascontiguousarray(indptr, np.int32)
        self.indices = np.ascontiguousarray(indices, np.int32)
        self.shape = (int(total_users), int(total_items), int(k))
        self.nnz = len(self.indices)
        self.n_items = int(n_items)
        self.device, self._stream = int(device), None
        self.h = _vp()
        check(lib().cornac_hip_bpr_create(C.byref(self.h), device, n_users, n_items, total_users, total_items, k,
                                          self.indptr, self.indices, self.nnz))

    def close(self):
        if getattr(self, "h", None) is not None and self.h and lib is not None:
            lib().cornac_hip_bpr_destroy(self.h)
            self.h = None

    __del__ = close

    def set_factors(self, U=None, V=None, B=None):
        U, V, B = _f32c(U), _f32c(V), _f32c(B)
        tu, ti, k = self.shape
        assert U is None or U.shape == (tu, k)
        assert V is None or V.shape == (ti, k)
        assert B is None or B.shape == (ti,)
        check(lib().cornac_hip_bpr_set_factors(self.h, _ptr(U), _ptr(V), _ptr(B)))

    def get_factors(self):
        tu, ti, k = self.shape
        U, V, B = np.empty((tu, k), np.float32), np.empty((ti, k), np.float32), np.empty(ti, np.float32)
        check(lib().cornac_hip_bpr_get_factors(self.h, U.ctypes.data, V.ctypes.data, B.ctypes.data))
        return U, V, B

    def set_factors_f64(self, U, V, B):
        """float64 tables (recom_bpr.pyx:211-214 is a fused-type function): the handle then trains in double"""
        tu, ti, k = self.shape
        U, V, B = (np.ascontiguousarray(a, np.float64) for a in (U, V, B))
        assert U.shape == (tu, k) and V.shape == (ti, k) and B.shape == (ti,)
        check(lib().cornac_hip_bpr_set_factors_f64(self.h, U.ctypes.data, V.ctypes.data, B.ctypes.data))

    def get_factors_f64(self):
        tu, ti, k = self.shape
        U, V, B = np.empty((tu, k), np.float64), np.empty((ti, k), np.float64), np.empty(ti, np.float64)
        check(lib().cornac_hip_bpr_get_factors_f64(self.h, U.ctypes.data, V.ctypes.data, B.ctypes.data))
        return U, V, B

    def fit_epochs_f64(self, n_epochs, lr, reg, use_bias=True, neg_population=NEG_UNIFORM):
        c, s = C.c_int64(), C.c_int64()
        check(lib().cornac_hip_bpr_fit_epochs_f64(self.h, n_epochs, float(lr), float(reg), int(use_bias), neg_population,
                                                  C.byref(c), C.byref(s)))
        return c.value, s.value

    def get_user_factors(self):
        """U only (the item tables may live elsewhere: row-sharded multi-GPU mode binds a local shard)"""
        tu, _, k = self.shape
        U = np.empty((tu, k), np.float32)
        check(lib().cornac_hip_bpr_get_factors(self.h, U.ctypes.data, None, None))
        return U

    def get_item_factors(self):
        """(V, B) only"""
        _, ti, k = self.shape
        V, B = np.empty((ti, k), np.float32), np.empty(ti, np.float32)
        check(lib().cornac_hip_bpr_get_factors(self.h, None, V.ctypes.data, B.ctypes.data))
        return V, B

    def seed_mt19937(self, seed_pos, seed_neg, shared_stream=False):
        check(lib().cornac_hip_bpr_seed_mt19937(self.h, seed_pos, seed_neg, int(shared_stream)))

    def seed_hogwild(self, seed):
        check(lib().cornac_hip_bpr_seed_hogwild(self.h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def fit_epochs(self, n_epochs, lr, reg, use_bias=True, neg_population=NEG_UNIFORM, mode=MODE_HOGWILD, flags=0):
        c, s = C.c_int64(), C.c_int64()
        check(lib().cornac_hip_bpr_fit_epochs(self.h, n_epochs, lr, reg, int(use_bias), neg_population, mode, flags,
                                              C.byref(c), C.byref(s)))
        return c.value, s.value

    def hogwild_enqueue(self, n_samples, lr, reg, use_bias=True, neg_population=NEG_UNIFORM, flags=0):
        check(lib().cornac_hip_bpr_hogwild_enqueue(self.h, n_samples, lr, reg, int(use_bias), neg_population, flags))

    def sync(self):
        c, s = C.c_int64(), C.c_int64()
        check(lib().cornac_hip_bpr_sync(self.h, C.byref(c), C.byref(s)))
        return c.value, s.value

    def bind_device(self, dU=None, dV=None, dB=None):
        check(lib().cornac_hip_bpr_bind_device(self.h, dU, dV, dB))

    def set_negative_population(self, items):
        """population of the popularity-weighted negative draw (WBPR): a multiset of train item ids, e.g. the GLOBAL item
        popularity when this handle holds a user slice; None / empty = the handle's own interactions (recom_wbpr.pyx:135)"""
        items = np.ascontiguousarray(items if items is not None else [], np.int32)
        check(lib().cornac_hip_bpr_set_negative_population(self.h, items.ctypes.data if len(items) else None, len(items)))

    def rebind_items(self, dV, dB):
        """swap the bound (caller-owned) item tables without synchronising the handle's stream"""
        check(lib().cornac_hip_bpr_rebind_items(self.h, dV, dB))

    # ---- conveyor layout (multi-GPU regime 2; cornac_amd/dist.py BinConveyorBprTrainer) ----------------------
    def conveyor_setup(self, n_blocks, rank_item=None, deal_seed=0, release_item_tables=True):
        """-> (n_bins, bins_per_block, cap): see include/cornac_hip.h"""
        order = None if rank_item is None else np.ascontiguousarray(rank_item, np.int32)
        nb, bpb, cap = C.c_int(), C.c_int(), C.c_int()
        check(lib().cornac_hip_bpr_conveyor_setup(self.h, int(n_blocks), None if order is None else order.ctypes.data,
                                                  int(deal_seed) & 0xFFFFFFFFFFFFFFFF, int(bool(release_item_tables)),
                                                  C.byref(nb), C.byref(bpb), C.byref(cap)))
        return nb.value, bpb.value, cap.value

    def conveyor_layout(self, layout_epoch, d_slot_item, d_item_slot):
        check(lib().cornac_hip_bpr_conveyor_layout(self.h, int(layout_epoch), d_slot_item, d_item_slot))

    def conveyor_enqueue(self, epoch, layout_epoch, first_blocks, d_rows, lr, reg, use_bias=True, neg_population=NEG_UNIFORM,
                         flags=0):
        n = len(first_blocks)
        fb = (C.c_int32 * n)(*[int(b) for b in first_blocks])
        rows = (_vp * n)(*[int(p) for p in d_rows])
        check(lib().cornac_hip_bpr_conveyor_enqueue(self.h, int(epoch), int(layout_epoch), n, fb, rows, lr, reg, int(use_bias),
                                                    neg_population, int(flags)))

    def device_ptrs(self):
        u, v, b = _vp(), _vp(), _vp()
        check(lib().cornac_hip_bpr_device_ptrs(self.h, C.byref(u), C.byref(v), C.byref(b)))
        return u.value, v.value, b.value

    def set_stream(self, stream_ptr):
        check(lib().cornac_hip_bpr_set_stream(self.h, stream_ptr))
        self._stream = stream_ptr

    def switch_stream(self, stream_ptr):
        """set_stream without waiting for the previous stream's queued work (the caller orders the streams)"""
        check(lib().cornac_hip_bpr_switch_stream(self.h, stream_ptr))
        self._stream = stream_ptr

    # ---- row-sharded item table building blocks (device pointers; see cornac_amd/dist.py) -----------
    def sample_triplets(self, n_draws, d_u, d_i, d_j, neg_population=NEG_UNIFORM):
        check(lib().cornac_hip_bpr_sample_triplets(self.h, int(n_draws), neg_population, d_u, d_i, d_j))

    def apply_triplets(self, d_u, d_slot_i, d_slot_j, n, d_rows, d_bias, bias_stride, lr, reg, use_bias=True):
        check(lib().cornac_hip_bpr_apply_triplets(self.h, d_u, d_slot_i, d_slot_j, int(n), d_rows, d_bias,
                                                  int(bias_stride), lr, reg, int(use_bias)))

    def staged_slots(self, n_draws):
        """array length emit_triplets needs for n_draws (0: this shape has no owned kernel)"""
        n = C.c_int64()
        check(lib().cornac_hip_bpr_staged_slots(self.h, int(n_draws), C.byref(n)))
        return n.value

    def emit_triplets(self, n_draws, d_u, d_i, d_j, slots_cap, neg_population=NEG_UNIFORM):
        n = C.c_int64()
        check(lib().cornac_hip_bpr_emit_triplets(self.h, int(n_draws), neg_population, d_u, d_i, d_j, int(slots_cap),
                                                 C.byref(n)))
        return n.value

    def apply_staged(self, d_u, d_slot_i, d_slot_j, n_slots, d_rows, d_bias, bias_stride, lr, reg, use_bias=True):
        check(lib().cornac_hip_bpr_apply_staged(self.h, d_u, d_slot_i, d_slot_j, int(n_slots), d_rows, d_bias,
                                                int(bias_stride), lr, reg, int(use_bias)))

    def shard_mark(self, d_i, d_j, n, world, rows_per_rank, d_mark):
        check(lib().cornac_hip_bpr_shard_mark(self.h, d_i, d_j, int(n), int(world), int(rows_per_rank), d_mark))

    def shard_slots(self, d_i, d_j, n, world, rows_per_rank, d_scan, d_slot_i, d_slot_j):
        check(lib().cornac_hip_bpr_shard_slots(self.h, d_i, d_j, int(n), int(world), int(rows_per_rank), d_scan, d_slot_i,
                                               d_slot_j))

    def shard_uniq(self, d_mark, d_scan, n_rows, rows_per_rank, d_uniq_local):
        check(lib().cornac_hip_bpr_shard_uniq(self.h, d_mark, d_scan, int(n_rows), int(rows_per_rank), d_uniq_local))

    def scatter_diff_rows(self, d_table, d_ids, n, width, d_now, d_before, d_scale=None):
        check(lib().cornac_hip_bpr_scatter_diff_rows(self.h, d_table, d_ids, int(n), int(width), d_now, d_before, d_scale))

    # the replicated table's elementwise passes (ItemTableReplica) on the handle's stream (and on its packed records
    # where it keeps them); rule 0 = sqrt through the first entry points, any rule through cornac_hip_bpr_table_delta
    def _delta(self, op, rule, *args):
        check(lib().cornac_hip_bpr_table_delta(self.h, op, int(rule), *args))

    def table_delta_begin(self, d_flat, d_base, n_items, k, d_bucket, d_local, rule=0):
        if rule:
            return self._delta(0, rule, d_flat, d_base, None, None, int(n_items), int(k), d_bucket, d_local)
        check(lib().cornac_hip_bpr_table_delta_begin(self.h, d_flat, d_base, int(n_items), int(k), d_bucket, d_local))

    def table_delta_finish(self, d_flat, d_base, d_bucket, d_local, n_items, k, rule=0):
        if rule:
            return self._delta(1, rule, d_flat, d_base, d_bucket, d_local, int(n_items), int(k), None, None)
        check(lib().cornac_hip_bpr_table_delta_finish(self.h, d_flat, d_base, d_bucket, d_local, int(n_items), int(k)))

    def table_delta_step(self, d_flat, d_base, d_bucket_prev, d_local_prev, n_items, k, d_bucket, d_local, rule=0):
        if rule:
            return self._delta(2, rule, d_flat, d_base, d_bucket_prev, d_local_prev, int(n_items), int(k), d_bucket, d_local)
        check(lib().cornac_hip_bpr_table_delta_step(self.h, d_flat, d_base, d_bucket_prev, d_local_prev, int(n_items),
                                                    int(k), d_bucket, d_local))

    # ---- resident exchange (multi-GPU regime 1 inside one launch per epoch; include/cornac_hip.h) ----
    def resident_exchange_bins(self, neg_population=NEG_UNIFORM, flags=0):
        """workgroups of the resident-exchange launch (= arrivals per exchange); 0: this shape / these flags do not take
        the LDS-bin form and the driver has to cut the epoch into chunk launches"""
        n = C.c_int()
        check(lib().cornac_hip_bpr_resident_exchange_bins(self.h, neg_population, flags, C.byref(n)))
        return n.value

    def epoch_resident_enqueue(self, lr, reg, use_bias, neg_population, flags, n_exchanges, rule, d_base, d_buckets,
                               bucket_stride, d_keeps, keep_stride, d_arrive, d_landed, d_applied):
        n = C.c_int()
        check(lib().cornac_hip_bpr_epoch_resident_enqueue(self.h, lr, reg, int(use_bias), neg_population, flags,
                                                          int(n_exchanges), int(rule), d_base, d_buckets, int(bucket_stride),
                                                          d_keeps, int(keep_stride), d_arrive, d_landed, d_applied, C.byref(n)))
        return n.value

    def resident_flush(self, n_exchanges, rule, d_base, d_buckets, bucket_stride, d_keeps, keep_stride, d_applied, d_landed=None):
        check(lib().cornac_hip_bpr_resident_flush(self.h, int(n_exchanges), int(rule), d_base, d_buckets, int(bucket_stride),
                                                  d_keeps, int(keep_stride), d_applied, d_landed))

    def gather_rows(self, d_table, d_ids, n, width, d_out):
        check(lib().cornac_hip_bpr_gather_rows(self.h, d_table, d_ids, int(n), int(width), d_out))

    def scatter_add_rows(self, d_table, d_ids, n, width, d_delta):
        check(lib().cornac_hip_bpr_scatter_add_rows(self.h, d_table, d_ids, int(n), int(width), d_delta))

    def debug_draw(self, stream, hi, n):
        out = np.empty(n, np.int64)
        check(lib().cornac_hip_bpr_debug_draw(self.h, stream, hi, n, out))
        return out

    def set_views(self, view_indptr, view_indices):
        self.v_indptr = np.ascontiguousarray(view_indptr, np.int32)
        self.v_indices = np.ascontiguousarray(view_indices, np.int32)
        if len(self.v_indices) == 0:
            self.v_indices = np.zeros(1, np.int32)
        check(lib().cornac_hip_bpr_set_views(self.h, self.v_indptr, self.v_indices, int(self.v_indptr[-1])))

    def seed_view_stream(self, seed_view):
        check(lib().cornac_hip_bpr_seed_view_stream(self.h, seed_view))

    def fit_epochs_vebpr(self, n_epochs, lr, reg, alpha, mode=MODE_HOGWILD, ownership=True):
        c, s = C.c_int64(), C.c_int64()
        check(lib().cornac_hip_vebpr_fit_epochs(self.h, n_epochs, lr, reg, alpha, mode | (0 if ownership else VEBPR_NO_OWNERSHIP),
                                                C.byref(c), C.byref(s)))
        return c.value, s.value

    def fit_epochs_vebpr_f64(self, n_epochs, lr, reg, alpha):
        """float64 tables (set_factors_f64): sequential semantics, everything in double (recom_vebpr.pyx:219)"""
        c, s = C.c_int64(), C.c_int64()
        check(lib().cornac_hip_vebpr_fit_epochs_f64(self.h, n_epochs, float(lr), float(reg), float(alpha), C.byref(c),
                                                    C.byref(s)))
        return c.value, s.value

    def vebpr_hogwild_owned(self):
        """True if the last VEBPR hogwild epoch ran with user-row ownership (k > 32, enough interactions per wave)"""
        o = C.c_int()
        check(lib().cornac_hip_vebpr_hogwild_form(self.h, C.byref(o)))
        return bool(o.value)

    def debug_ownership(self):
        """(wave_ptr, own_u, own_i) of the hogwild sampler's user-row ownership, or None if unused"""
        w = C.c_int64()
        check(lib().cornac_hip_bpr_debug_ownership(self.h, C.byref(w), None, None, None))
        if w.value == 0:
            return None
        wp = np.empty(w.value + 1, np.int64)
        ou, oi = np.empty(self.nnz, np.int32), np.empty(self.nnz, np.int32)
        check(lib().cornac_hip_bpr_debug_ownership(self.h, C.byref(w), wp.ctypes.data, ou.ctypes.data,
                                                   oi.ctypes.data))
        return wp, ou, oi

    def ldsbin_config(self, hot_x1000=75, min_candidates=48, max_rounds=4):
        check(lib().cornac_hip_bpr_ldsbin_config(self.h, int(hot_x1000), int(min_candidates), int(max_rounds)))

    def ldsbin_deal_config(self, strata_groups=16, hot_cost_x16=32):
        check(lib().cornac_hip_bpr_ldsbin_deal_config(self.h, int(strata_groups), int(hot_cost_x16)))

    def debug_ldsbin_deal(self, seed, epoch):
        """(bin_of_item, cold_mass, hot_off, hot_u, hot_i) of the deal of `epoch` (test hook)."""
        st = self.ldsbin_stats()
        bins, nh = st["bins"], st["hot_interactions"]
        bin_of = np.empty(self.n_items, np.int32)
        cold, off = np.empty(bins, np.uint32), np.empty(bins + 1, np.uint32)
        hu, hi = np.empty(max(nh, 1), np.int32), np.empty(max(nh, 1), np.int32)
        check(lib().cornac_hip_bpr_debug_ldsbin_deal(self.h, int(seed), int(epoch), bin_of.ctypes.data, cold.ctypes.data,
                                                     off.ctypes.data, hu.ctypes.data, hi.ctypes.data))
        return bin_of, cold, off, hu[:nh], hi[:nh]

    def ldsbin_pass_config(self, enable=True, waves=8, lds_kb=64, min_draws_x100=200):
        """the "passing bins" regime of the LDS-bin form (item tables that need more than max_rounds rounds): see
        include/cornac_hip.h"""
        check(lib().cornac_hip_bpr_ldsbin_pass_config(self.h, int(bool(enable)), int(waves), int(lds_kb), int(min_draws_x100)))

    def ldsbin_stats(self):
        o = (C.c_int64 * 8)()
        check(lib().cornac_hip_bpr_ldsbin_stats(self.h, o))
        return {"bins": o[0], "rows_per_bin": o[1], "n_hot": o[2], "hot_interactions": o[3], "bitmap_words": o[4],
                "lds_bytes": o[5], "lock_timeouts": o[6], "block_threads": o[7]}

    def strata_config(self, hot_permille=120, hot_min_mult_x100=200, rehash_period=1):
        check(lib().cornac_hip_bpr_strata_config(self.h, int(hot_permille), int(hot_min_mult_x100), int(rehash_period)))

    def chunk_records(self, enable=True):
        """let hogwild_enqueue keep the strata form's packed item records from call to call (see include/cornac_hip.h:
        the caller touches the bound dense table only through table_delta_* until sync())"""
        check(lib().cornac_hip_bpr_chunk_records(self.h, int(bool(enable))))

    def strata_stats(self):
        o = (C.c_int64 * 4)()
        check(lib().cornac_hip_bpr_strata_stats(self.h, o))
        return {"n_hot": o[0], "misplaced_workgroups": o[1], "bucket_builds": o[2], "waves": o[3]}

    def debug_strata(self, epoch):
        """(sptr, rec_u, rec_i, rank_item, key): the partition buckets the strata form uses in `epoch`"""
        wp = self.debug_ownership()[0]
        W = len(wp) - 1
        sptr = np.empty(8 * W + 1, np.int64)
        ru, ri = np.empty(self.nnz, np.int32), np.empty(self.nnz, np.int32)
        rank_item = np.empty(self.n_items, np.int32)
        key = C.c_uint32()
        check(lib().cornac_hip_bpr_debug_strata(self.h, int(epoch), sptr.ctypes.data, ru.ctypes.data, ri.ctypes.data,
                                                rank_item.ctypes.data, C.byref(key)))
        return sptr, ru, ri, rank_item, key.value

    def kernel_timing(self, enable=True):
        """(total_ms, launches) of the hogwild kernel launches recorded since the last call (HIP events)."""
        ms, n = C.c_double(), C.c_int64()
        check(lib().cornac_hip_bpr_kernel_timing(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_timing(self):
        t = (C.c_double * 4)()
        check(lib().cornac_hip_bpr_last_timing(self.h, t))
        return {"sampler_ms": t[0], "schedule_ms": t[1], "sgd_ms": t[2], "total_ms": t[3]}


class MfTrainer:
    def __init__(self, rid, cid, val, n_users, n_items, k, device=0):
        self.rid = np.ascontiguousarray(rid, np.int64)
        self.cid = np.ascontiguousarray(cid, np.int64)
        self.val = np.ascontiguousarray(val, np.float32)
        self.shape = (int(n_users), int(n_items), int(k))
        self.h = _vp()
        check(lib().cornac_hip_mf_create(C.byref(self.h), device, n_users, n_items, k, self.rid, self.cid, self.val,
                                         len(self.val)))
        self.device, self._stream = int(device), None

    def close(self):
        if getattr(self, "h", None) is not None and self.h and lib is not None:
            lib().cornac_hip_mf_destroy(self.h)
            self.h = None

    __del__ = close

    def set_factors(self, U=None, V=None, Bu=None, Bi=None):
        U, V, Bu, Bi = _f32c(U), _f32c(V), _f32c(Bu), _f32c(Bi)
        check(lib().cornac_hip_mf_set_factors(self.h, _ptr(U), _ptr(V), _ptr(Bu), _ptr(Bi)))

    def get_factors(self):
        nu, ni, k = self.shape
        U, V = np.empty((nu, k), np.float32), np.empty((ni, k), np.float32)
        Bu, Bi = np.empty(nu, np.float32), np.empty(ni, np.float32)
        check(lib().cornac_hip_mf_get_factors(self.h, U.ctypes.data, V.ctypes.data, Bu.ctypes.data, Bi.ctypes.data))
        return U, V, Bu, Bi

    def fit(self, max_iter, lr, reg, mu, use_bias=True, early_stop=False, mode=MODE_HOGWILD):
        loss = np.zeros(max(max_iter, 1), np.float32)
        n = C.c_int()
        check(lib().cornac_hip_mf_fit(self.h, max_iter, lr, reg, mu, int(use_bias), int(early_stop), mode,
                                      loss.ctypes.data, C.byref(n)))
        return loss[:n.value], n.value

    # ---- multi-GPU driver surface (cornac_amd/dist.py ShardedMfTrainer) ----------------------------------------
    def bind_items(self, d_V, d_Bi):
        """train into caller-owned device buffers for the item side (the replicated [V | Bi] table)"""
        check(lib().cornac_hip_mf_bind_items(self.h, d_V, d_Bi))

    def bind_users(self, d_U, d_Bu):
        """train into caller-owned device buffers for the user side (dist.MfBlockRotationTrainer: the handles of a rank's item
        blocks share it)"""
        check(lib().cornac_hip_mf_bind_users(self.h, d_U, d_Bu))

    def set_stream(self, hip_stream):
        check(lib().cornac_hip_mf_set_stream(self.h, hip_stream))
        self._stream = hip_stream

    def epoch_enqueue(self, part, n_parts, lr, reg, mu, use_bias=True):
        """ratings [nnz*part/n_parts, nnz*(part+1)/n_parts) of the stored order, no host synchronisation"""
        check(lib().cornac_hip_mf_epoch_enqueue(self.h, int(part), int(n_parts), lr, reg, mu, int(use_bias)))

    def sync(self):
        """wait for the enqueued slices; the sum of squared errors they saw"""
        l = C.c_double()
        check(lib().cornac_hip_mf_sync(self.h, C.byref(l)))
        return l.value

    # the replicated table's elementwise passes (ItemTableReplica), on the stream of set_stream; rule: dist.RULES
    def table_delta_begin(self, d_flat, d_base, n_items, k, d_bucket, d_local, rule=0):
        check(lib().cornac_hip_table_delta(0 | (int(rule) << 4), self.device, self._stream, d_flat, d_base, None, None,
                                           int(n_items), int(k), d_bucket, d_local))

    def table_delta_finish(self, d_flat, d_base, d_bucket, d_local, n_items, k, rule=0):
        check(lib().cornac_hip_table_delta(1 | (int(rule) << 4), self.device, self._stream, d_flat, d_base, d_bucket, d_local,
                                           int(n_items), int(k), None, None))

    def table_delta_step(self, d_flat, d_base, d_bucket_prev, d_local_prev, n_items, k, d_bucket, d_local, rule=0):
        check(lib().cornac_hip_table_delta(2 | (int(rule) << 4), self.device, self._stream, d_flat, d_base, d_bucket_prev,
                                           d_local_prev, int(n_items), int(k), d_bucket, d_local))

    OPTIMIZERS = {"sgd": 0, "adam": 1, "rmsprop": 2, "adagrad": 3}

    def fit_minibatch(self, order, batch_size, optimizer, lr, reg, mu, use_bias=True, keep_u=None, keep_i=None,
                      keep_scale=1.0):
        """one optimiser step per consecutive slice of `batch_size` entries of `order` (indices into the
        rating arrays); returns the summed squared error over all visited ratings.  keep_u / keep_i: the dropout keep
        masks of the gathered user / item rows, uint8 [len(order), k] (row b belongs to order[b]; kept factors are scaled
        by keep_scale); None = no dropout"""
        order = np.ascontiguousarray(order, np.int64)
        loss = C.c_double()
        if keep_u is not None:
            keep_u, keep_i = np.ascontiguousarray(keep_u, np.uint8), np.ascontiguousarray(keep_i, np.uint8)
            if keep_u.shape != (len(order), self.shape[2]) or keep_i.shape != keep_u.shape:
                raise ValueError("keep masks must be [len(order), k] = %r, got %r / %r"
                                 % ((len(order), self.shape[2]), keep_u.shape, keep_i.shape))
            check(lib().cornac_hip_mf_fit_minibatch_dropout(self.h, order, len(order), int(batch_size),
                                                            self.OPTIMIZERS[optimizer], lr, reg, mu, int(use_bias),
                                                            keep_u.ctypes.data, keep_i.ctypes.data, float(keep_scale),
                                                            C.byref(loss)))
            return loss.value
        check(lib().cornac_hip_mf_fit_minibatch(self.h, order, len(order), int(batch_size), self.OPTIMIZERS[optimizer],
                                                lr, reg, mu, int(use_bias), C.byref(loss)))
        return loss.value

    def reset_optimizer(self):
        check(lib().cornac_hip_mf_reset_optimizer(self.h))

    def kernel_timing(self, enable=True):
        ms, n = C.c_double(), C.c_int64()
        check(lib().cornac_hip_mf_kernel_timing(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def hogwild_form(self, form):
        """0 automatic, 1 fused atomic kernel, 2 block rotation (include/cornac_hip.h)"""
        check(lib().cornac_hip_mf_hogwild_form(self.h, int(form)))

    def hogwild_stats(self):
        o = (C.c_int64 * 4)()
        check(lib().cornac_hip_mf_hogwild_stats(self.h, o))
        return {"form_used": o[0], "tiles": o[1], "rows_per_bin": o[2], "gave_up": bool(o[3])}

    def last_timing(self):
        t = (C.c_double * 4)()
        check(lib().cornac_hip_mf_last_timing(self.h, t))
        return {"schedule_ms": t[1], "sgd_ms": t[2], "total_ms": t[3]}


def mf_fit_sgd(rid, cid, val, U, V, Bu, Bi, lr, reg, mu, max_iter, use_bias, early_stop, mode, device=0):
    """One-shot in-place call with the reference's argument list (backend_cpu.pyx:35-40)."""
    loss = np.zeros(max(max_iter, 1), np.float32)
    n = C.c_int()
    check(lib().cornac_hip_mf_fit_sgd(device, np.ascontiguousarray(rid, np.int64), np.ascontiguousarray(cid, np.int64),
                                      np.ascontiguousarray(val, np.float32), len(val), U, V, Bu, Bi, U.shape[0],
                                      V.shape[0], U.shape[1], lr, reg, mu, max_iter, int(use_bias), int(early_stop),
                                      mode, loss.ctypes.data, C.byref(n)))
    return loss[:n.value]


class Scorer:
    def __init__(self, U, V, item_base=None, user_base=None, device=0):
        U, V = _f32c(U), _f32c(V)
        self.n_users, self.k = U.shape
        self.n_items = V.shape[0]
        self.h = _vp()
        check(lib().cornac_hip_scorer_create(C.byref(self.h), device, self.n_users, self.n_items, self.k))
        self.set(U, V, item_base, user_base)

    def set(self, U, V, item_base=None, user_base=None):
        ib, ub = _f32c(item_base), _f32c(user_base)
        check(lib().cornac_hip_scorer_set(self.h, _f32c(U), _f32c(V), _ptr(ib), _ptr(ub)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h and lib is not None:
            lib().cornac_hip_scorer_destroy(self.h)
            self.h = None

    __del__ = close

    def score_user(self, user):
        out = np.empty(self.n_items, np.float32)
        check(lib().cornac_hip_score_user(self.h, int(user), out))
        return out

    def set_f64(self, U, V, item_base=None, user_base=None):
        """the float64 tables of a model trained in double (fast_dot's ddot variant, fast_dot.pyx:25-43)"""
        U, V = np.ascontiguousarray(U, np.float64), np.ascontiguousarray(V, np.float64)
        assert U.shape == (self.n_users, self.k) and V.shape == (self.n_items, self.k)
        ib = None if item_base is None else np.ascontiguousarray(item_base, np.float64)
        ub = None if user_base is None else np.ascontiguousarray(user_base, np.float64)
        check(lib().cornac_hip_scorer_set_f64(self.h, U.ctypes.data, V.ctypes.data, _ptr(ib), _ptr(ub)))

    def score_user_f64(self, user):
        out = np.empty(self.n_items, np.float64)
        check(lib().cornac_hip_score_user_f64(self.h, int(user), out.ctypes.data))
        return out

    def score_block(self, users):
        users = np.ascontiguousarray(users, np.int32)
        out = np.empty((len(users), self.n_items), np.float32)
        check(lib().cornac_hip_score_block(self.h, users, len(users), out))
        return out

    def score_pairs(self, users, items, clip=None):
        users = np.ascontiguousarray(users, np.int32)
        items = np.ascontiguousarray(items, np.int32)
        out = np.empty(len(users), np.float32)
        lo, hi = (0.0, 0.0) if clip is None else (float(clip[0]), float(clip[1]))
        check(lib().cornac_hip_score_pairs(self.h, users, items, len(users), int(clip is not None), lo, hi, out))
        return out

    def rank_topk(self, users, topk, exclude=None):
        """exclude: optional (indptr int64[n+1], indices int32) CSR of items to drop per listed user."""
        users = np.ascontiguousarray(users, np.int32)
        items = np.empty((len(users), topk), np.int32)
        scores = np.empty((len(users), topk), np.float32)
        ip = ix = None
        if exclude is not None:
            ip = np.ascontiguousarray(exclude[0], np.int64)
            ix = np.ascontiguousarray(exclude[1], np.int32)
        check(lib().cornac_hip_rank_topk(self.h, users, len(users), topk, _ptr(ip), _ptr(ix), items, scores))
        return items, scores

    def rank_positions(self, users, targets, exclude=None):
        """targets / exclude: CSR `(indptr int64[n+1], indices int32)` per listed user.  Returns per target
        `(greater, pos, ge, score)`: candidates scored strictly higher, its position in the ranked order, candidates
        scored at least as high (itself included), its score — cornac_hip_rank_positions."""
        users = np.ascontiguousarray(users, np.int32)
        tp = np.ascontiguousarray(targets[0], np.int64)
        tx = np.ascontiguousarray(targets[1], np.int32)
        nt = int(tp[-1])
        greater, pos, ge = (np.empty(nt, np.int32) for _ in range(3))
        scores = np.empty(nt, np.float32)
        ip = ix = None
        if exclude is not None:
            ip = np.ascontiguousarray(exclude[0], np.int64)
            ix = np.ascontiguousarray(exclude[1], np.int32)
        check(lib().cornac_hip_rank_positions(self.h, users, len(users), _ptr(ip), _ptr(ix), _ptr(tp), _ptr(tx), greater,
                                              pos, ge, scores))
        return greater, pos, ge, scores

    def set_exclusions(self, indptr=None, indices=None):
        """Keep per-USER exclusion lists on the device: CSR (indptr int64[n_users + 1], indices int32); None drops them."""
        if indptr is None:
            check(lib().cornac_hip_scorer_set_exclusions(self.h, None, None))
            return
        ip = np.ascontiguousarray(indptr, np.int64)
        ix = np.ascontiguousarray(indices, np.int32)
        assert len(ip) == self.n_users + 1
        check(lib().cornac_hip_scorer_set_exclusions(self.h, ip.ctypes.data, _ptr(ix) if len(ix) else None))

    def rank_topk_resident(self, users, topk, fetch=True, timed=False, pinned=False):
        """top-k with the resident exclusion lists.  users: array of user ids, or (u0, n) for a contiguous range.
        fetch=False leaves the results on the device, fetch="items" copies only the item ids back (what the @k metrics
        read); timed=True also returns the HIP-event milliseconds; pinned=True returns views of page-locked memory owned
        by the scorer (faster device-to-host copy, overwritten by the next pinned call)."""
        if isinstance(users, tuple):
            up, u0, n = None, int(users[0]), int(users[1])
        else:
            ua = np.ascontiguousarray(users, np.int32)
            up, u0, n = ua.ctypes.data, 0, len(ua)
        if fetch and pinned:
            # results land in page-locked memory owned by the scorer: the arrays returned are VIEWS of it, overwritten by
            # the next pinned call and valid until close()
            want_scores = fetch != "items"
            nbytes = n * topk * 4 * (2 if want_scores else 1)
            buf = _vp()
            check(lib().cornac_hip_scorer_host_buffer(self.h, nbytes, C.byref(buf)))
            raw = (C.c_char * nbytes).from_address(buf.value)
            items = np.frombuffer(raw, np.int32, n * topk).reshape(n, topk)
            scores = np.frombuffer(raw, np.float32, n * topk, offset=n * topk * 4).reshape(n, topk) if want_scores else None
        else:
            items = np.empty((n, topk), np.int32) if fetch else None
            scores = np.empty((n, topk), np.float32) if fetch and fetch != "items" else None
        ms = C.c_double()
        check(lib().cornac_hip_rank_topk_resident(self.h, up, u0, n, topk, _ptr(items), _ptr(scores),
                                                  C.cast(C.byref(ms), _vp) if timed else None))
        return (items, scores, ms.value) if timed else (items, scores)

    def rank_topk_device_ms(self, u0, n, topk, repeats=1):
        ms = C.c_double()
        check(lib().cornac_hip_rank_topk_device(self.h, u0, n, topk, repeats, C.byref(ms)))
        return ms.value


class VbprTrainer:
    NAMES = ("Bi", "Gu", "Gi", "Tu", "E", "Bp")

    def __init__(self, features, n_users, n_items, k, k2, device=0):
        self.F = np.ascontiguousarray(features, np.float32)
        assert self.F.shape[0] == n_items
        self.dims = dict(n_users=int(n_users), n_items=int(n_items), k=int(k), k2=int(k2), n_feat=self.F.shape[1])
        self.h = _vp()
        check(lib().cornac_hip_vbpr_create(C.byref(self.h), device, n_users, n_items, k, k2, self.F.shape[1], self.F))

    def close(self):
        if getattr(self, "h", None) is not None and self.h and lib is not None:
            lib().cornac_hip_vbpr_destroy(self.h)
            self.h = None

    __del__ = close

    def _shapes(self):
        d = self.dims
        return {"Bi": (d["n_items"],), "Gu": (d["n_users"], d["k"]), "Gi": (d["n_items"], d["k"]),
                "Tu": (d["n_users"], d["k2"]), "E": (d["n_feat"], d["k2"]), "Bp": (d["n_feat"],)}

    def set_params(self, **params):
        arrs = []
        for n in self.NAMES:
            a = params.get(n)
            if a is not None:
                a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(self._shapes()[n]))
            arrs.append(a)
        check(lib().cornac_hip_vbpr_set_params(self.h, *[_ptr(a) for a in arrs]))

    def get_params(self):
        out = {n: np.empty(s, np.float32) for n, s in self._shapes().items()}
        check(lib().cornac_hip_vbpr_get_params(self.h, *[out[n].ctypes.data for n in self.NAMES]))
        return out

    def fit_batches(self, u, i, j, batch_size, lr, lambda_w, lambda_b, lambda_e):
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        nll = C.c_double()
        check(lib().cornac_hip_vbpr_fit_batches(self.h, u, i, j, len(u), batch_size, lr, lambda_w, lambda_b, lambda_e,
                                                C.byref(nll)))
        return nll.value

    def item_tables(self):
        d = self.dims
        th, vb = np.empty((d["n_items"], d["k2"]), np.float32), np.empty(d["n_items"], np.float32)
        check(lib().cornac_hip_vbpr_item_tables(self.h, th, vb))
        return th, vb


class WmfTrainer:
    """Resident WMF trainer: CSC rating matrix, U/V and the Adam state live on the device."""

    def __init__(self, csc, k, device=0):
        csc = csc.tocsc()
        self.n_users, self.n_items = (int(x) for x in csc.shape)
        self.k = int(k)
        self._indptr = np.ascontiguousarray(csc.indptr, np.int64)
        self._rows = np.ascontiguousarray(csc.indices, np.int32)
        self._vals = np.ascontiguousarray(csc.data, np.float32)
        self.h = _vp()
        check(lib().cornac_hip_wmf_create(C.byref(self.h), device, self.n_users, self.n_items, self.k, self._indptr,
                                          self._rows if len(self._rows) else np.zeros(1, np.int32),
                                          self._vals if len(self._vals) else np.zeros(1, np.float32), len(self._vals)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h and lib is not None:
            lib().cornac_hip_wmf_destroy(self.h)
            self.h = None

    __del__ = close

    def set_factors(self, U, V):
        U = np.ascontiguousarray(np.asarray(U, np.float32).reshape(self.n_users, self.k))
        V = np.ascontiguousarray(np.asarray(V, np.float32).reshape(self.n_items, self.k))
        check(lib().cornac_hip_wmf_set_factors(self.h, U, V))

    def get_factors(self):
        U, V = np.empty((self.n_users, self.k), np.float32), np.empty((self.n_items, self.k), np.float32)
        check(lib().cornac_hip_wmf_get_factors(self.h, U, V))
        return U, V

    def fit_batches(self, batches, lambda_u, lambda_v, a, b, lr):
        """batches: sequence of item-id arrays (1..128 distinct ids each); returns the per-step losses"""
        batches = [np.asarray(x, np.int32).ravel() for x in batches]
        if not batches:
            return np.zeros(0)
        ptr = np.zeros(len(batches) + 1, np.int64)
        np.cumsum([len(x) for x in batches], out=ptr[1:])
        ids = np.ascontiguousarray(np.concatenate(batches), np.int32)
        loss = np.zeros(len(batches), np.float64)
        check(lib().cornac_hip_wmf_fit_batches(self.h, ids, ptr, len(batches), lambda_u, lambda_v, a, b, lr,
                                               loss.ctypes.data))
        return loss

    def kernel_timing(self, enabled=True):
        check(lib().cornac_hip_wmf_kernel_timing(self.h, int(enabled)))

    def last_device_ms(self):
        ms = C.c_double()
        check(lib().cornac_hip_wmf_last_timing(self.h, C.byref(ms)))
        return ms.value
