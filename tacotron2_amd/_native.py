"""ctypes binding of libmelvoc.so (include/melvoc.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception, never a silent PyTorch path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MELVOC_LIB") or os.path.join(_HERE, "libmelvoc.so")  # MELVOC_LIB: diagnostic builds

PREC_FP32, PREC_BF16, PREC_BF16X3, PREC_F16 = 0, 1, 2, 3
OUT_F32, OUT_PCM16 = 0, 1
PRECISIONS = {"fp32": PREC_FP32, "float32": PREC_FP32, "bf16": PREC_BF16, "bfloat16": PREC_BF16,
              "bf16x3": PREC_BF16X3,  # bf16x3: fp32-grade results from split bf16 operands (melvoc.h)
              "f16": PREC_F16, "fp16": PREC_F16, "float16": PREC_F16}  # float16 operands at the bf16 rate (melvoc.h)

EXPORTS = (
    "mv_backend", "mv_last_error", "mv_hifigan_create", "mv_hifigan_forward", "mv_waveglow_create",
    "mv_waveglow_infer", "mv_workspace_bytes", "mv_hop", "mv_macs_per_frame", "mv_destroy", "mv_pcm16",
    "mv_test_conv1d", "mv_test_conv_transpose1d", "mv_profile_begin", "mv_profile_end",
    "mv_postnet_create", "mv_postnet_forward", "mv_default_options", "mv_hifigan_forward_static",
)
STATIC_SCRATCH_BYTES = 64 << 20  # MV_STATIC_SCRATCH_BYTES


class MelvocError(RuntimeError):
    pass


class mv_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class mv_hifigan_cfg(C.Structure):
    _fields_ = [
        ("num_mels", C.c_int),
        ("upsample_initial_channel", C.c_int),
        ("num_upsamples", C.c_int),
        ("upsample_rates", C.c_int * 8),
        ("upsample_kernel_sizes", C.c_int * 8),
        ("num_kernels", C.c_int),
        ("resblock_kernel_sizes", C.c_int * 8),
        ("resblock_dilations", (C.c_int * 3) * 8),
    ]


class mv_waveglow_cfg(C.Structure):
    _fields_ = [
        ("n_mel_channels", C.c_int),
        ("n_flows", C.c_int),
        ("n_group", C.c_int),
        ("n_early_every", C.c_int),
        ("n_early_size", C.c_int),
        ("n_layers", C.c_int),
        ("n_channels", C.c_int),
        ("kernel_size", C.c_int),
    ]


class mv_postnet_cfg(C.Structure):
    _fields_ = [("dim_data", C.c_int), ("embedding_dim", C.c_int), ("kernel_size", C.c_int),
                ("n_convolutions", C.c_int), ("bn_eps", C.c_float)]


class mv_options(C.Structure):
    _fields_ = [("fuse", C.c_int), ("io16", C.c_int), ("wg_fold", C.c_int), ("wg_gemm", C.c_int)]


def make_options(options):
    """dict of mv_options fields (missing ones keep the library's defaults) -> mv_options, or None for NULL."""
    if not options:
        return None
    o = mv_options()
    lib().mv_default_options(C.byref(o))
    for k, v in options.items():
        if k not in ("fuse", "io16", "wg_fold", "wg_gemm"):
            raise MelvocError(f"unknown option {k!r} (mv_options has fuse, io16, wg_fold, wg_gemm)")
        setattr(o, k, int(v))
    return o


class mv_profile_row(C.Structure):
    _fields_ = [("name", C.c_char * 192), ("detail", C.c_char * 32), ("launches", C.c_int), ("ms", C.c_double),
                ("macs_per_frame", C.c_double)]


_lib = None


def build(force: bool = False) -> str:
    """Compile libmelvoc.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MelvocError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C tacotron2_amd/csrc`). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load it first so that libmelvoc.so binds to the
    # SAME HIP runtime as the tensors and streams it is handed (two runtimes in one process cannot
    # share device pointers).
    import torch  # noqa: F401

    l = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    l.mv_backend.restype = i32
    l.mv_last_error.restype = C.c_char_p
    optp = C.POINTER(mv_options)
    l.mv_default_options.argtypes = [optp]
    l.mv_default_options.restype = None
    l.mv_hifigan_create.argtypes = [C.POINTER(mv_hifigan_cfg), C.POINTER(mv_tensor), i32, i32, i32, optp, C.POINTER(vp)]
    l.mv_hifigan_forward.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, C.c_size_t, vp]
    l.mv_hifigan_forward_static.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp]
    l.mv_waveglow_create.argtypes = [C.POINTER(mv_waveglow_cfg), C.POINTER(mv_tensor), i32, i32, i32, optp, C.POINTER(vp)]
    l.mv_waveglow_infer.argtypes = [vp, vp, vp, i32, i32, C.POINTER(vp), i32, C.c_float, vp, i32, vp, C.c_size_t, vp]
    l.mv_postnet_create.argtypes = [C.POINTER(mv_postnet_cfg), C.POINTER(mv_tensor), i32, i32, i32, optp, C.POINTER(vp)]
    l.mv_postnet_forward.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, C.c_size_t, vp]
    l.mv_workspace_bytes.argtypes = [vp, i32, i32]
    l.mv_workspace_bytes.restype = C.c_size_t
    l.mv_hop.argtypes = [vp]
    l.mv_macs_per_frame.argtypes = [vp]
    l.mv_macs_per_frame.restype = C.c_double
    l.mv_destroy.argtypes = [vp]
    l.mv_destroy.restype = None
    l.mv_pcm16.argtypes = [vp, C.c_size_t, vp, vp]
    l.mv_test_conv1d.argtypes = [vp, vp, i32, i32, i32, f32p, f32p, i32, i32, i32, C.c_float, i32, vp, vp]
    l.mv_test_conv_transpose1d.argtypes = [vp, vp, i32, i32, i32, f32p, f32p, i32, i32, i32, i32, C.c_float, i32, vp, vp]
    l.mv_profile_begin.argtypes = [vp]
    l.mv_profile_end.argtypes = [vp, C.POINTER(mv_profile_row), i32, C.POINTER(i32)]
    _lib = l
    return l


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().mv_last_error().decode(errors="replace")
        raise MelvocError(f"{what} failed (code {rc}): {msg}")


def make_tensors(state):
    """dict name -> float32 numpy array  =>  (ctypes array of mv_tensor, keep-alive list)."""
    keep = []
    arr = (mv_tensor * len(state))()
    for i, (k, v) in enumerate(state.items()):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        if a.ndim < 1 or a.ndim > 4:
            raise MelvocError(f"tensor {k}: unsupported rank {a.ndim}")
        name = k.encode()
        keep.append((a, name))
        arr[i].name = name
        arr[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].ndim = a.ndim
        for d in range(a.ndim):
            arr[i].shape[d] = a.shape[d]
    return arr, keep


def hifigan_cfg(cfg) -> mv_hifigan_cfg:
    c = mv_hifigan_cfg()
    c.num_mels = int(cfg.get("num_mels", 80))
    c.upsample_initial_channel = int(cfg["upsample_initial_channel"])
    rates, ks = list(cfg["upsample_rates"]), list(cfg["upsample_kernel_sizes"])
    rk, rd = list(cfg["resblock_kernel_sizes"]), list(cfg["resblock_dilation_sizes"])
    if len(rates) != len(ks) or len(rates) > 8 or len(rk) != len(rd) or len(rk) > 8:
        raise MelvocError("hifigan config: inconsistent upsample / resblock lists")
    c.num_upsamples = len(rates)
    for i, (u, k) in enumerate(zip(rates, ks)):
        c.upsample_rates[i] = int(u)
        c.upsample_kernel_sizes[i] = int(k)
    c.num_kernels = len(rk)
    for j, (k, d) in enumerate(zip(rk, rd)):
        if len(d) != 3:
            raise MelvocError("hifigan config: ResBlock1 takes three dilations per kernel size")
        c.resblock_kernel_sizes[j] = int(k)
        for m in range(3):
            c.resblock_dilations[j][m] = int(d[m])
    return c


def waveglow_cfg(cfg) -> mv_waveglow_cfg:
    wn = cfg["WN_config"]
    return mv_waveglow_cfg(int(cfg["n_mel_channels"]), int(cfg["n_flows"]), int(cfg["n_group"]),
                           int(cfg["n_early_every"]), int(cfg["n_early_size"]), int(wn["n_layers"]),
                           int(wn["n_channels"]), int(wn["kernel_size"]))


class Handle:
    """Owns an mv_handle; destroyed with the Python object."""

    def __init__(self, ptr, kind):
        self.ptr = C.c_void_p(ptr)
        self.kind = kind

    def __del__(self):
        try:
            if self.ptr:
                lib().mv_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    @property
    def hop(self):
        return lib().mv_hop(self.ptr)

    @property
    def macs_per_frame(self):
        return lib().mv_macs_per_frame(self.ptr)

    def workspace_bytes(self, B, Tmax):
        return lib().mv_workspace_bytes(self.ptr, int(B), int(Tmax))

    def profile_begin(self):
        check(lib().mv_profile_begin(self.ptr), "mv_profile_begin")

    def profile_end(self):
        """-> list of dicts {name, detail, launches, ms, macs_per_frame}: `name` is the kernel's symbol as rocprofv3
        prints it, `detail` tells the kinds of launches of one symbol apart."""
        rows = (mv_profile_row * 64)()
        n = C.c_int(0)
        check(lib().mv_profile_end(self.ptr, rows, 64, C.byref(n)), "mv_profile_end")
        return [dict(name=rows[i].name.decode(), detail=rows[i].detail.decode(), launches=rows[i].launches,
                     ms=rows[i].ms, macs_per_frame=rows[i].macs_per_frame) for i in range(n.value)]


def _optref(options):
    o = make_options(options)
    return C.byref(o) if o is not None else None


def create_hifigan(cfg, state, device_index: int, precision: int, options=None) -> Handle:
    arr, keep = make_tensors(state)
    c = hifigan_cfg(cfg)
    out = C.c_void_p()
    check(lib().mv_hifigan_create(C.byref(c), arr, len(keep), int(device_index), int(precision), _optref(options),
                                  C.byref(out)),
          "mv_hifigan_create")
    return Handle(out.value, "hifigan")


def create_waveglow(cfg, state, device_index: int, precision: int, options=None) -> Handle:
    arr, keep = make_tensors(state)
    c = waveglow_cfg(cfg)
    out = C.c_void_p()
    check(lib().mv_waveglow_create(C.byref(c), arr, len(keep), int(device_index), int(precision), _optref(options),
                                   C.byref(out)),
          "mv_waveglow_create")
    return Handle(out.value, "waveglow")


def create_postnet(cfg: mv_postnet_cfg, state, device_index: int, precision: int, options=None) -> Handle:
    arr, keep = make_tensors(state)
    out = C.c_void_p()
    check(lib().mv_postnet_create(C.byref(cfg), arr, len(keep), int(device_index), int(precision), _optref(options),
                                  C.byref(out)),
          "mv_postnet_create")
    return Handle(out.value, "postnet")


WORKSPACE_CACHE_SLOTS = 4  # scratch tensors kept per model: least recently used ones beyond that are dropped


def workspace(cache: dict, dev, nbytes: int):
    """Scratch tensor for one forward/infer call, cached per (device, current stream): two calls in flight on
    different streams never share a buffer, and one stream's calls are ordered anyway.  (A handle itself is
    read-only in forward/infer; only mv_profile_* mutate it.)  The cache keeps the WORKSPACE_CACHE_SLOTS most recently
    used entries: a caller that makes a stream per request does not pile up one workspace per dead stream (a dropped
    tensor goes back to torch's caching allocator, which keeps it alive for work still queued on its stream)."""
    import torch

    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = cache.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
    cache[key] = ws  # (re-inserted last: dicts keep insertion order, so the first key is the least recently used)
    while len(cache) > WORKSPACE_CACHE_SLOTS:
        cache.pop(next(iter(cache)))
    return ws


def device_lengths(lengths, B, T, dev):
    """int32 [B] frame counts on `dev`.  Host-side lengths are checked here; lengths that already live on the device
    go through as they are (a check would cost a device->host sync in every call): every kernel of the library
    clamps the entries it reads into [0, T]."""
    import torch

    lengths = torch.as_tensor(lengths)
    if lengths.numel() != B:
        raise RuntimeError("lengths must hold B frame counts in [0, T]")
    if lengths.is_cuda:
        return lengths.to(device=dev, dtype=torch.int32).contiguous()
    lengths = lengths.to(torch.int32)
    if B and (int(lengths.max()) > T or int(lengths.min()) < 0):
        raise RuntimeError("lengths must hold B frame counts in [0, T]")
    return lengths.to(dev, non_blocking=True).contiguous()
