"""ctypes binding of libsvx.so (include/svx.h).  No CPU fallback: if the library or a GPU is
missing, importing/creating a context raises."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVX_LIB") or os.path.join(_HERE, "libsvx.so")   # (SVX_LIB: A/B builds, profiles/build_variant.sh)

SVX_F32, SVX_F16, SVX_BF16 = 0, 1, 2
SVX_MAX_TYPES = 128
SVX_MAX_LEVELS = 16
SVX_MARGIN_RATIO, SVX_MARGIN_DISTANCE = 0, 1
SVX_SEARCH_COARSE_TO_FINE, SVX_SEARCH_STRAIGHT = 0, 1

SVX_OK, SVX_ERR_ARG, SVX_ERR_OVERLAPS, SVX_ERR_HIP, SVX_ERR_TRACEBACK = 0, 1, 2, 3, 4
SVX_ERR_NOMEM, SVX_ERR_EXTEND, SVX_ERR_PATH, SVX_ERR_BP = 5, 6, 7, 8

# messages of the reference for the device-side failure codes (dp_utils.py:124,167,243)
DEVICE_ERRORS = {
    SVX_ERR_TRACEBACK: "traceback bug",
    SVX_ERR_BP: "got unknown value",
    SVX_ERR_EXTEND: "asked to extend alignments but already bigger than requested",
    SVX_ERR_PATH: "search path is not a unit-step lattice path",
}

c_int, c_i64, c_f32, c_f64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class AlignParams(ctypes.Structure):
    _fields_ = [
        ("dtype", ctypes.c_int32),
        ("d", ctypes.c_int32),
        ("n_types", ctypes.c_int32),
        ("types", ctypes.c_int32 * (2 * SVX_MAX_TYPES)),
        ("width_over2", ctypes.c_int32),
        ("max_size_full_dp", ctypes.c_int32),
        ("costs_sample_size", ctypes.c_int32),
        ("num_samps_for_norm", ctypes.c_int32),
        ("del_percentile_frac", ctypes.c_double),
        ("search_mode", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
    ]


class LevelView(ctypes.Structure):
    _fields_ = [
        ("size0", ctypes.c_int32), ("size1", ctypes.c_int32), ("k0", ctypes.c_int32), ("k1", ctypes.c_int32),
        ("n_types", ctypes.c_int32), ("band", ctypes.c_int32), ("path_len", ctypes.c_int32), ("n_align", ctypes.c_int32),
        ("n0", c_vp), ("n1", c_vp), ("del_penalty", c_vp), ("searchpath", c_vp), ("a_b_costs", c_vp), ("b_offset", c_vp),
        ("a_b_csum", c_vp), ("a_b_bp", c_vp), ("a_b_xp", c_vp), ("a_b_yp", c_vp), ("new_b_offset", c_vp),
        ("alignments", c_vp), ("alignment_scores", c_vp),
        ("costs_1to1", c_vp), ("x_y_tb_diag", c_vp), ("v0_l0", c_vp), ("v1_l0", c_vp),
        ("knob_scores", c_vp), ("n_knob", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class Pair(ctypes.Structure):
    _fields_ = [
        ("vecs0", c_vp), ("vecs1", c_vp),
        ("n", ctypes.c_int32), ("m", ctypes.c_int32), ("k0", ctypes.c_int32), ("k1", ctypes.c_int32),
        ("norm_idx", c_vp), ("knob_idx", c_vp),
        ("norms0", c_vp), ("norms1", c_vp),
        ("align", c_vp), ("scores", c_vp), ("info", c_vp), ("del_pen", c_vp),
    ]


_SIGS = {
    "svx_create": (c_int, [c_int, ctypes.POINTER(c_vp)]),
    "svx_destroy": (c_int, [c_vp]),
    "svx_set_stream": (c_int, [c_vp, c_vp]),
    "svx_synchronize": (c_int, [c_vp]),
    "svx_last_error": (ctypes.c_char_p, [c_vp]),
    "svx_version": (ctypes.c_char_p, []),
    "svx_scratch_bytes": (c_i64, [c_vp]),
    "svx_dense_costs": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp]),
    "svx_dense_dp": (c_int, [c_vp, c_vp, c_int, c_int, c_f32, c_vp, c_vp]),
    "svx_score_path": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp]),
    "svx_sparse_costs": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int,
                                 ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_vp, c_vp]),
    "svx_sparse_dp": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(ctypes.c_int32), c_int, c_f64, c_int, c_int,
                              c_vp, c_vp, c_vp, c_vp]),
    "svx_make_norm1": (c_int, [c_vp, c_vp, c_i64, c_int]),
    "svx_downsample": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "svx_compute_norms": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "svx_del_penalty": (c_int, [c_vp, c_vp, c_i64, c_f64, c_vp]),
    "svx_dense_traceback": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "svx_sparse_traceback": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "svx_search_path": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "svx_gather_rows": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_i64, c_vp]),
    "svx_unit_rows": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_int]),
    "svx_knn_mean_sim": (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_vp]),
    "svx_knn_topk_merge": (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_vp, c_int, c_vp]),
    "svx_margin_scores": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_int, c_vp]),
    "svx_mt19937_choice": (c_int, [c_vp, ctypes.POINTER(ctypes.c_int32), c_i64, c_i64, c_vp]),
    "svx_norm_index_count": (c_i64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "svx_knob_index_count": (c_i64, [c_int, c_int, c_int, c_int]),
    "svx_draw_indices": (c_int, [c_vp, ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_vp, c_vp]),
    "svx_candidate_table": (c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, c_int, c_vp, c_int,
                                    ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_i64), ctypes.c_char_p, c_int]),
    "svx_format_alignments": (c_i64, [c_vp, c_vp, c_i64, c_vp, c_i64]),
    "svx_num_levels": (c_int, [c_int, c_int, c_int]),
    "svx_knob_count": (c_i64, [c_int, c_int, c_int]),
    "svx_align_batch": (c_int, [c_vp, ctypes.POINTER(AlignParams), ctypes.POINTER(Pair), c_int]),
    "svx_debug_level": (c_int, [c_vp, c_int, c_int, ctypes.POINTER(LevelView)]),
    "svx_copy_to_host": (c_int, [c_vp, c_vp, c_vp, c_i64]),
    "svx_set_profiling": (c_int, [c_vp, c_int]),
    "svx_set_pipeline": (c_int, [c_vp, c_int]),
    "svx_flush": (c_int, [c_vp]),
    "svx_stage_ms": (c_f64, [c_vp, ctypes.c_char_p]),
    "svx_stage_launches": (c_int, [c_vp, ctypes.c_char_p]),
}

EXPORTS = tuple(_SIGS)

_lib = None
_lock = threading.Lock()


def load():
    """Load libsvx.so (built by __graft_entry__.build() / `make -C speech-vecalign_amd/csrc`)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    "libsvx.so not found at %s: build it with `make -C speech-vecalign_amd/csrc` "
                    "(there is no CPU fallback)" % LIB_PATH)
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


class SvxError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class Context:
    """One svx_ctx per (process, device)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("svx needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = load()
        self.device = int(device)
        h = c_vp()
        rc = self.lib.svx_create(self.device, ctypes.byref(h))
        if rc != 0:
            raise SvxError(rc, self.lib.svx_last_error(None).decode())
        self.h = h
        self.torch = torch
        self.tdev = torch.device("cuda", self.device)
        self.pipeline = False
        self._held = []   # batches whose device work the pipeline may still hold back: kept alive until the next flush
        if os.environ.get("SVX_PIPELINE") == "1":   # default for this process (the GPU suite is also run once this way)
            self.set_pipeline(True)

    def use_current_stream(self):
        self.lib.svx_set_stream(self.h, c_vp(self.torch.cuda.current_stream(self.tdev).cuda_stream))

    def check(self, rc):
        if rc != 0:
            msg = self.lib.svx_last_error(self.h).decode()
            if rc == SVX_ERR_OVERLAPS:
                raise Exception(msg)  # the reference raises a plain Exception with this text
            raise SvxError(rc, msg)

    def sync(self):
        self.check(self.lib.svx_synchronize(self.h))
        del self._held[:]

    def set_pipeline(self, on):
        """Software pipeline over consecutive svx_align_batch calls (include/svx.h: svx_set_pipeline); while it is on,
        outputs are complete only after flush() / sync()."""
        self.check(self.lib.svx_set_pipeline(self.h, 1 if on else 0))
        self.pipeline = bool(on)
        del self._held[:]   # (switching flushes)

    def flush(self):
        self.check(self.lib.svx_flush(self.h))
        # everything the pipeline held back is now ordered in front of whatever follows on the current stream, so memory
        # released from here on is reused behind it
        del self._held[:]

    def hold(self, batch):
        """The pipeline reads a batch's tensors on internal streams the tensor library knows nothing about, possibly
        during the NEXT call: keep the last few batches alive until a flush."""
        if self.pipeline and not any(b is batch for b in self._held):   # (a batch that is run again and again counts once)
            if len(self._held) >= 4:
                self.flush()
            self._held.append(batch)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.svx_destroy(self.h)
                self.h = None
        except Exception:
            pass


_ctxs = {}


def context(device=None):
    """The svx context of `device` (default: torch's current device, so that a process that called
    torch.cuda.set_device(LOCAL_RANK) works on its own GPU without passing the index around)."""
    if device is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("svx needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        device = torch.cuda.current_device()
    device = int(device)
    if device not in _ctxs:
        _ctxs[device] = Context(device)
    c = _ctxs[device]
    c.use_current_stream()
    return c
