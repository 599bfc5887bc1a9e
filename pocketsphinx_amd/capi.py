"""ctypes binding of libpsgpu.so (include/psgpu.h)."""
import ctypes as C
import glob
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("PSGPU_LIB_PATH") or os.path.join(PKG_DIR, "libpsgpu.so")     # (PSGPU_LIB_PATH: a measuring build, tools/build_prof_lib.py)
CSRC = os.path.join(PKG_DIR, "csrc")
SOURCES = ["psgpu_core.hip", "psgpu_ptm.hip", "psgpu_ptm_frame.hip", "psgpu_hmm.hip", "psgpu_semi.hip", "psgpu_ms.hip", "psgpu_feat.hip", "psgpu_fe.hip", "psgpu_search.hip", "psgpu_lm.hip", "psgpu_flat.hip", "psgpu_decode.hip"]

# per-source flags (after the common ones).  The tree search is ONE kernel of ~9,000 instructions whose frame loop is a chain of
# dependent steps: its speed follows the number of scalar values the compiler keeps alive (what does not fit the scalar registers is
# spilled to vector-register lanes and read back with v_readlane at every use) and the size of the loop.  -O3's unrolling of its
# `for (i = tid; i < n; i += NT)` loops -- nearly all of which run once -- buys nothing and costs both: measured on the 512 x 279-frame
# search (round 4), -O3 6.04 ms, -O3 -fno-unroll-loops 5.47, -O1 5.42, -Os 5.20, -Oz 6.65.
FILE_FLAGS = {"psgpu_search.hip": ["-Os"]}

# every symbol include/psgpu.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "psgpu_version", "psgpu_last_error", "psgpu_device_count", "psgpu_set_device",
    "psgpu_get_device", "psgpu_malloc", "psgpu_free", "psgpu_host_alloc", "psgpu_host_free", "psgpu_memcpy_h2d", "psgpu_memcpy_d2h", "psgpu_stream_sync",
    "psgpu_ptm_model_create", "psgpu_ptm_model_free", "psgpu_ptm_n_sen", "psgpu_ptm_n_chain",
    "psgpu_ptm_veclen", "psgpu_ptm_topn", "psgpu_ptm_score_batch_dev", "psgpu_ptm_score_batch",
    "psgpu_event_create", "psgpu_event_destroy", "psgpu_event_record", "psgpu_event_elapsed_ms",
    "psgpu_ptm_topn_dev", "psgpu_ptm_senone_dev", "psgpu_ptm_kernel_timing", "psgpu_ptm_last_kernel_ms",
    "psgpu_ptm_state_create", "psgpu_ptm_state_free", "psgpu_ptm_state_reset",
    "psgpu_ptm_frame_eval", "psgpu_ptm_state_get_topn", "psgpu_ptm_state_set_topn", "psgpu_ptm_state_lookahead", "psgpu_ptm_state_lookahead_stats", "psgpu_ptm_state_lookahead_rows", "psgpu_ptm_state_mark_fresh",
    "psgpu_semi_model_create", "psgpu_semi_model_free", "psgpu_semi_state_create",
    "psgpu_semi_state_free", "psgpu_semi_state_reset", "psgpu_semi_frame_eval",
    "psgpu_semi_score_batch_dev", "psgpu_semi_score_batch",
    "psgpu_semi_state_get_topn", "psgpu_semi_state_set_topn",
    "psgpu_ms_model_create", "psgpu_ms_model_free", "psgpu_ms_n_sen", "psgpu_ms_veclen",
    "psgpu_ms_frame_eval", "psgpu_ms_lookahead", "psgpu_ms_lookahead_covers", "psgpu_ms_lookahead_stats",
    "psgpu_ms_frame_eval_at", "psgpu_ms_score_batch_dev", "psgpu_ms_batch_check", "psgpu_ms_score_batch",
    "psgpu_feat_1s_c_d_dd_dev", "psgpu_feat_1s_c_d_dd",
    "psgpu_fe_create", "psgpu_fe_free", "psgpu_fe_out_dim", "psgpu_fe_n_frames", "psgpu_fe_log_dev",
    "psgpu_fe_process_utts_dev", "psgpu_fe_process_utts",
    "psgpu_hmm_ctx_create", "psgpu_hmm_ctx_free", "psgpu_hmm_n_emit_state",
    "psgpu_hmm_vit_eval_dev", "psgpu_hmm_vit_eval", "psgpu_phone_loop_run_dev", "psgpu_phone_loop_run_lists_dev", "psgpu_hmm_ctx_stream",
    "psgpu_fwdtree_create", "psgpu_fwdtree_free", "psgpu_fwdtree_search_dev", "psgpu_fwdtree_search_session_dev", "psgpu_fwdtree_search_lists_dev", "psgpu_fwdtree_can_score_lists", "psgpu_fwdtree_n_mpx_channels", "psgpu_fwdtree_set_lm", "psgpu_fwdtree_backtrace_dev", "psgpu_fwdtree_layout", "psgpu_fwdtree_use_slab_layout", "psgpu_fwdtree_grow", "psgpu_fwdtree_full_capacity", "psgpu_fwdtree_n_single_phone_words", "psgpu_abi_version", "psgpu_capabilities", "psgpu_fe_stream_step_dev", "psgpu_fe_frame_size", "psgpu_fe_frame_shift", "psgpu_feat_live_state_words", "psgpu_feat_live_state_init", "psgpu_feat_live_step_dev", "psgpu_decode_streams_pcm_begin", "psgpu_decode_streams_step_pcm", "psgpu_live_pieces", "psgpu_feat_create", "psgpu_feat_free", "psgpu_feat_out_dim", "psgpu_feat_cepsize", "psgpu_feat_window", "psgpu_feat_compute_dev", "psgpu_feat_compute", "psgpu_decode_set_feat",
    "psgpu_decode_create", "psgpu_decode_free", "psgpu_decode_set_model", "psgpu_decode_score_mode", "psgpu_decode_session", "psgpu_decode_session_set", "psgpu_decode_session_get", "psgpu_decode_first_pass_dev", "psgpu_decode_front_end_ahead", "psgpu_decode_first_pass", "psgpu_decode_first_pass_feat",
    "psgpu_decode_view", "psgpu_decode_fetch_hyps", "psgpu_decode_fetch_tables", "psgpu_decode_fetch_tables_range", "psgpu_decode_stage_timing", "psgpu_decode_last_stage_ms", "psgpu_decode_search_after", "psgpu_decode_wait_scored", "psgpu_stream_create_dedicated", "psgpu_stream_destroy", "psgpu_fwdtree_hyp_out", "psgpu_decode_table_capacity", "psgpu_decode_second_pass", "psgpu_decode_tables_grown", "psgpu_decode_search_lag", "psgpu_fwdtree_search_lag", "psgpu_fwdtree_search_resume", "psgpu_fwdtree_search_streams", "psgpu_fwdtree_search_restart", "psgpu_phone_loop_carry_restart", "psgpu_decode_live_begin", "psgpu_decode_live_restart", "psgpu_decode_streams_begin", "psgpu_decode_streams_step", "psgpu_decode_streams_restart", "psgpu_decode_streams_next_utt", "psgpu_decode_live_step", "psgpu_decode_live_frames_searched", "psgpu_phone_loop_run_carry_dev", "psgpu_phone_loop_carry_words", "psgpu_decode_set_scorer", "psgpu_decode_compallsen", "psgpu_ms_score_batch_raw_dev", "psgpu_ms_batch_needs_lists", "psgpu_ms_list_entries_per_frame", "psgpu_semi_n_sen", "psgpu_semi_score_batch_carry_dev", "psgpu_semi_n_feat", "psgpu_semi_topn", "psgpu_semi_veclen",
    "psgpu_fwdflat_create", "psgpu_fwdflat_free", "psgpu_fwdflat_set_lm", "psgpu_fwdflat_search_dev", "psgpu_fwdflat_search_feats_dev", "psgpu_fwdflat_search_feats_lists_dev", "psgpu_ptm_batch_open_flags", "psgpu_ptm_model_view",
    "psgpu_lm_create", "psgpu_lm_create_interp", "psgpu_lm_free", "psgpu_lm_tg_score_dev",
]


class PsgpuError(RuntimeError):
    pass


def build_library(force=False, extra_flags=(), lib_path=None, build_dir=None):
    """Compile the HIP sources for gfx950 into pocketsphinx_amd/libpsgpu.so
    (in-tree, so it travels to the GPU box).  hipcc cross-compiles without a GPU.
    One object per source under pocketsphinx_amd/_build/ (recompiled when the source or a header is newer),
    compiled in parallel, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    lib_path = lib_path or os.path.join(PKG_DIR, "libpsgpu.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    # every header under csrc/ (device-side headers are included by several sources) + the public one
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "psgpu.h")]
    if (not force) and os.path.exists(lib_path) and \
            all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in srcs + hdrs):
        return lib_path
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
             # packed fp32 (v_pk_*) runs at half rate on gfx950 and the SLP pass
             # doubles register pressure here: keep the distance chain scalar
             "-fno-slp-vectorize", "-Wno-unused-value", "-Wno-unused-result", "-fPIC", "-I" + os.path.join(ROOT, "include")] + list(extra_flags)
    bdir = build_dir or os.path.join(PKG_DIR, "_build")
    os.makedirs(bdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(src):
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            subprocess.check_call([hipcc] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


_lib = None


def lib():
    """Load libpsgpu.so; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64; load it FIRST so libpsgpu binds to
        # the same runtime (two HIP runtimes in one process lose the device).
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise PsgpuError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
    L.psgpu_version.restype = C.c_char_p
    L.psgpu_last_error.restype = C.c_char_p
    L.psgpu_set_device.argtypes = [C.c_int]
    L.psgpu_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.psgpu_free.argtypes = [vp]
    L.psgpu_memcpy_h2d.argtypes = [vp, vp, C.c_size_t, vp]
    L.psgpu_memcpy_d2h.argtypes = [vp, vp, C.c_size_t, vp]
    L.psgpu_stream_sync.argtypes = [vp]
    L.psgpu_ptm_model_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp, i32, i32, i32,
                                         vp, vp, vp, vp, vp, vp, i32]
    L.psgpu_ptm_model_free.argtypes = [vp]
    L.psgpu_ptm_model_free.restype = None
    for f in ("psgpu_ptm_n_sen", "psgpu_ptm_n_chain", "psgpu_ptm_veclen", "psgpu_ptm_topn"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = i32
    L.psgpu_ptm_score_batch_dev.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, u32, vp]
    L.psgpu_ptm_score_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, u32]
    L.psgpu_ptm_topn_dev.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.psgpu_ptm_senone_dev.argtypes = [vp, i32, vp, vp, vp, vp, u32, vp]
    L.psgpu_ptm_state_create.argtypes = [C.POINTER(vp), vp, i32]
    L.psgpu_ptm_state_free.argtypes = [vp]
    L.psgpu_ptm_state_free.restype = None
    L.psgpu_ptm_state_reset.argtypes = [vp]
    L.psgpu_ptm_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32]
    L.psgpu_ptm_state_get_topn.argtypes = [vp, i32, vp, vp, vp]
    L.psgpu_ptm_state_set_topn.argtypes = [vp, i32, vp, vp, vp]
    L.psgpu_ptm_state_lookahead.argtypes = [vp, vp, i32, i32]
    L.psgpu_ptm_state_lookahead_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.psgpu_semi_model_create.argtypes = [C.POINTER(vp), i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32]
    L.psgpu_semi_model_free.argtypes = [vp]
    L.psgpu_semi_model_free.restype = None
    L.psgpu_semi_state_create.argtypes = [C.POINTER(vp), vp, i32]
    L.psgpu_semi_state_free.argtypes = [vp]
    L.psgpu_semi_state_free.restype = None
    L.psgpu_semi_state_reset.argtypes = [vp]
    L.psgpu_semi_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32]
    L.psgpu_semi_state_get_topn.argtypes = [vp, i32, vp, vp, vp]
    L.psgpu_semi_state_set_topn.argtypes = [vp, i32, vp, vp, vp]
    L.psgpu_ms_model_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp,
                                        vp, i32, i32, i32]
    L.psgpu_ms_model_free.argtypes = [vp]
    L.psgpu_ms_model_free.restype = None
    L.psgpu_ms_n_sen.argtypes = [vp]
    L.psgpu_ms_veclen.argtypes = [vp]
    L.psgpu_ms_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32]
    L.psgpu_ms_score_batch_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.psgpu_ms_batch_check.argtypes = [vp, vp]
    L.psgpu_ms_score_batch.argtypes = [vp, vp, i32, vp]
    L.psgpu_feat_1s_c_d_dd_dev.argtypes = [vp, vp, i32, i32, vp, vp]
    L.psgpu_feat_1s_c_d_dd.argtypes = [vp, vp, i32, i32, vp]
    L.psgpu_hmm_ctx_create.argtypes = [C.POINTER(vp), i32, i32, vp, i32, vp, i32]
    L.psgpu_hmm_ctx_free.argtypes = [vp]
    L.psgpu_hmm_ctx_free.restype = None
    L.psgpu_hmm_n_emit_state.argtypes = [vp]
    L.psgpu_hmm_vit_eval_dev.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp]
    L.psgpu_hmm_vit_eval.argtypes = [vp, vp, i32, vp, C.POINTER(i32)]
    L.psgpu_ptm_kernel_timing.argtypes = [vp, i32]
    L.psgpu_ptm_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.psgpu_event_create.argtypes = [C.POINTER(vp)]
    L.psgpu_event_destroy.argtypes = [vp]
    L.psgpu_event_record.argtypes = [vp, vp]
    L.psgpu_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    _lib = L
    return L


def check(rc, what="psgpu call"):
    if rc != 0:
        raise PsgpuError("%s failed (%d): %s" % (what, rc, lib().psgpu_last_error().decode()))
