"""ctypes binding of libkpb.so (include/kpb.h).  There is no CPU fallback: if the HIP library is
missing or no gfx950 device is visible, every entry point raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPB_LIB_PATH: another build of the library (the measurement-only knock-out builds of scripts/hack_build.py, an A/B partner) WITHOUT copying it over
# the tree's own -- an interrupted script then cannot leave a hack build where the evidence runs look for the product
SO_PATH = os.environ.get("KPB_LIB_PATH") or os.path.join(_HERE, "libkpb.so")

c_int, c_float, c_double, c_void_p, c_int64, c_size_t = (ctypes.c_int, ctypes.c_float, ctypes.c_double,
                                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t)


class DetectParams(ctypes.Structure):
    _fields_ = [("nms_dist", ctypes.c_int32), ("threshold", c_float), ("border_dist", ctypes.c_int32),
                ("top_k", ctypes.c_int32), ("min_score", c_float)]


class LgParams(ctypes.Structure):
    _fields_ = [("depth_confidence", c_float), ("width_confidence", c_float), ("filter_threshold", c_float),
                ("prune_min_kpts", ctypes.c_int32)]


class LkParams(ctypes.Structure):           # kpb_lk_params
    _fields_ = [("distance", ctypes.c_float), ("win_size", ctypes.c_int32), ("levels", ctypes.c_int32), ("iterations", ctypes.c_int32)]


class RansacParams(ctypes.Structure):       # kpb_ransac_params
    _fields_ = [("threshold", ctypes.c_double), ("confidence", ctypes.c_double), ("max_iters", ctypes.c_int32), ("refine", ctypes.c_int32)]


class MatchParams(ctypes.Structure):
    _fields_ = [("max_distance", c_double), ("cross_check", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/kpb.h one to one (tests check the export list)
SIGNATURES = {
    "kpb_version": (c_int, []),
    "kpb_last_error": (ctypes.c_char_p, [c_void_p]),
    "kpb_ctx_create": (c_int, [c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "kpb_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "kpb_ctx_destroy": (None, [c_void_p]),
    "kpb_sync": (c_int, [c_void_p]),
    "kpb_ctx_set_option": (c_int, [c_void_p, c_int, c_int64]),
    "kpb_prof_enable": (c_int, [c_void_p, c_int]),
    "kpb_prof_report": (c_int, [c_void_p, ctypes.c_char_p, c_size_t]),
    "kpb_fast_nms": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "kpb_detect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(DetectParams), c_void_p, c_void_p,
                           c_void_p, c_int]),
    "kpb_detect_check": (c_int, [c_void_p]),
    "kpb_detect_counts": (c_int, [c_void_p, c_void_p, c_int]),
    "kpb_match_counts": (c_int, [c_void_p, c_void_p, c_int]),
    "kpb_sample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                           c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "kpb_match": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                          ctypes.POINTER(MatchParams), c_void_p, c_void_p, c_void_p]),
    "kpb_gather_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                c_void_p]),
    "kpb_warp_homography": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "kpb_warp_se3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "kpb_val_keypoints": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "kpb_lk_track": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                             c_void_p, c_void_p, c_void_p]),
    "kpb_epipolar_error": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_float, c_void_p, c_void_p]),
    "kpb_find_homography": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_uint32,
                                    ctypes.POINTER(RansacParams), c_void_p, c_void_p, c_void_p]),
    "kpb_find_fundamental": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_uint32,
                                     ctypes.POINTER(RansacParams), c_void_p, c_void_p, c_void_p]),
    "kpb_find_essential": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   ctypes.c_uint32, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "kpb_recover_pose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p,
                                 c_void_p]),
    "kpb_preprocess": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "kpb_net_create": (c_int, [c_void_p, c_int, c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    "kpb_net_destroy": (None, [c_void_p]),
    "kpb_net_desc_dim": (c_int, [c_void_p]),
    "kpb_net_desc_div": (c_int, [c_void_p]),
    "kpb_net_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "kpb_net_desc_at": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "kpb_lg_create": (c_int, [c_void_p, c_void_p, c_size_t, c_float, ctypes.POINTER(c_void_p)]),
    "kpb_lg_destroy": (None, [c_void_p]),
    "kpb_lg_input_dim": (c_int, [c_void_p]),
    "kpb_lg_set_attention": (c_int, [c_void_p, c_int]),
    "kpb_lg_match": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int64, c_int64, c_int64, c_int64, c_int, c_int, ctypes.POINTER(LgParams), c_void_p, c_void_p,
                             c_void_p, c_void_p]),
}

_lib = None


class KpbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkpb error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libkpb.so and declare every prototype.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        # torch first: libkpb.so must bind to the HIP runtime torch already loaded (tensors are the
        # containers for every device buffer); a second runtime loaded ahead of torch's sees no device.
        import torch  # noqa: F401
        if not os.path.exists(SO_PATH):
            raise RuntimeError("%s is missing: run `python -m keypoint_bench_amd.build` (hipcc, gfx950). "
                               "keypoint_bench_amd has no CPU fallback." % SO_PATH)
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Context:
    """One per process / GPU (kpb_ctx).  Enqueues on torch's current stream of `device`."""
    _instances = {}

    def __init__(self, device_index: int):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("keypoint_bench_amd needs an MI355X (gfx950) GPU; none is visible and there is no "
                               "CPU fallback")
        self.lib = load()
        self.device_index = device_index
        stream = torch.cuda.current_stream(device_index).cuda_stream
        self.stream = stream
        h = c_void_p()
        rc = self.lib.kpb_ctx_create(device_index, c_void_p(stream), ctypes.byref(h))
        if rc != 0:
            raise KpbError(rc, self.lib.kpb_last_error(None).decode())
        self.handle = h

    @classmethod
    def get(cls, device):
        import torch
        idx = device.index if isinstance(device, torch.device) else int(device)
        if idx is None:
            idx = torch.cuda.current_device()
        if idx not in cls._instances:
            cls._instances[idx] = cls(idx)
        inst = cls._instances[idx]
        stream = torch.cuda.current_stream(idx).cuda_stream      # follow torch's current stream (with torch.cuda.stream(...))
        if stream != inst.stream:
            inst.check(inst.lib.kpb_ctx_set_stream(inst.handle, c_void_p(stream)))
            inst.stream = stream
        return inst

    def check(self, rc):
        if rc != 0:
            raise KpbError(rc, self.lib.kpb_last_error(self.handle).decode())

    def sync(self):
        self.check(self.lib.kpb_sync(self.handle))

    OPT_COVIS_STORE_BYTES = 1       # KPB_OPT_COVIS_STORE_BYTES

    def set_option(self, option: int, value: int):
        """kpb_ctx_set_option (include/kpb.h): a limit of this context, e.g. OPT_COVIS_STORE_BYTES."""
        self.check(self.lib.kpb_ctx_set_option(self.handle, int(option), int(value)))

    def prof_enable(self, on: bool):
        self.check(self.lib.kpb_prof_enable(self.handle, 1 if on else 0))

    def prof_report(self):
        """{kernel name: (launches, total_ms)} since the last report (HIP events on the launch stream)."""
        buf = ctypes.create_string_buffer(1 << 16)
        self.check(self.lib.kpb_prof_report(self.handle, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = (int(calls), float(ms))
        return out


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
