"""ctypes binding of libfa_hip.so (include/fa_hip.h).  No torch imports here.

This is the stub INTEGRATION.md shows for the reference side: the reference's
pybind entry (src/flash_attention.cu:137-140) reduces to filling ``fa_fwd_args``
and calling ``fa_fwd_launch``.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FA_HIP_LIB: another build of the same library (tests/test_gpu_parity.py runs the timing-perturbed FA_JITTER build,
# lib/libfa_hip_jitter.so, through the same binding).  Not a fallback: the named file must exist.
LIB_PATH = os.environ.get("FA_HIP_LIB") or os.path.join(_HERE, "lib", "libfa_hip.so")

FA_FP16, FA_BF16 = 5, 15

CONFIG_FIELDS = (
    "dtype", "d_head", "B_r", "B_c", "n_warps", "async_copy", "eager_load_blocks",
    "swizzled", "Q_mma_load_K_tiles", "K_mma_load_K_tiles", "V_mma_load_K_tiles",
    "mma_double_buffer_loads", "optimized_softmax",
)

# every symbol include/fa_hip.h declares (tests check the library exports them all)
EXPORTED_SYMBOLS = (
    "fa_init", "fa_fwd_supported", "fa_fwd_lds_bytes", "fa_fwd_launch",
    "fa_fwd_launch_timed", "fa_num_kernels", "fa_get_kernel", "fa_last_error", "fa_version",
    "fa_fwd_masked_supported", "fa_fwd_launch_masked", "fa_device_state",
    "fa_fwd_ex_supported", "fa_fwd_launch_ex", "fa_fwd_query",
    "fa_adaptive_state", "fa_adaptive_state_for", "fa_adaptive_reset", "fa_adaptive_simulate", "fa_get_kernel_sized", "fa_fwd_query_sized", "fa_abi_version",
)
FA_SPECULATIVE_OFF, FA_SPECULATIVE_ALWAYS, FA_SPECULATIVE_ADAPTIVE = 0, 1, 2  # fa_speculative_mode
FA_ABI_VERSION = 6
SOFTMAX_MODES = ("eager", "first_block_skip", "lazy", "speculative")  # fa_softmax_mode


class FaFwdConfig(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in CONFIG_FIELDS]


class FaFwdArgs(ctypes.Structure):
    _fields_ = [
        ("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p),
        ("o", ctypes.c_void_p),
        ("batch", ctypes.c_int64), ("seq_len", ctypes.c_int64), ("n_heads", ctypes.c_int64),
        ("d_head", ctypes.c_int64),
        ("batch_stride", ctypes.c_int64), ("seq_stride", ctypes.c_int64),
        ("head_stride", ctypes.c_int64),
        ("cfg", FaFwdConfig),
    ]


class FaKernelInfo(ctypes.Structure):
    _fields_ = [
        ("cfg", FaFwdConfig), ("threads", ctypes.c_int32), ("lds_bytes", ctypes.c_int32),
        ("num_regs", ctypes.c_int32), ("scratch_bytes", ctypes.c_int32),
        ("rows_per_wave", ctypes.c_int32), ("masked", ctypes.c_int32),
        ("softmax_mode", ctypes.c_int32), ("prescaled_q", ctypes.c_int32),
        # ABI 5: the ring form of a 32-rows-per-wave configuration (include/fa_hip.h)
        ("ring_form", ctypes.c_int32), ("ring_softmax_mode", ctypes.c_int32),
        ("ring_num_regs", ctypes.c_int32), ("ring_scratch_bytes", ctypes.c_int32),
        ("ring_lds_bytes", ctypes.c_int32), ("persistent", ctypes.c_int32), ("alt_form", ctypes.c_int32),
        ("ring_threads", ctypes.c_int32),
    ]


class FaFwdStats(ctypes.Structure):
    """fa_fwd_stats: two uint32 counters in DEVICE memory the kernel adds to."""

    _fields_ = [("items", ctypes.c_uint32), ("items_redone", ctypes.c_uint32)]


class FaFwdOpts(ctypes.Structure):
    """fa_fwd_opts: the extensions of fa_fwd_launch_ex (zero = the reference's launch)."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("causal", ctypes.c_int32), ("allow_ragged", ctypes.c_int32),
        ("speculative", ctypes.c_int32), ("prescaled_q", ctypes.c_int32),
        ("ms", ctypes.POINTER(ctypes.c_float)), ("stats", ctypes.c_void_p),
    ]


class FaAdaptiveInfo(ctypes.Structure):
    """fa_adaptive_info: the adaptive speculative mode's record on one device."""

    _fields_ = [(name, ctypes.c_uint32) for name in
                ("available", "launches", "demoted", "reports", "hold", "mode", "remaining", "last_report")]


def make_opts(causal=False, allow_ragged=False, speculative=False, prescaled_q=False, ms=None, stats_ptr=None):
    """`speculative`: False / True, or "adaptive" (= 2, FA_SPECULATIVE_ADAPTIVE)."""
    o = FaFwdOpts()
    o.struct_size = ctypes.sizeof(FaFwdOpts)
    o.causal, o.allow_ragged = int(bool(causal)), int(bool(allow_ragged))
    adaptive = speculative == "adaptive" or (not isinstance(speculative, bool) and speculative == FA_SPECULATIVE_ADAPTIVE)
    o.speculative = FA_SPECULATIVE_ADAPTIVE if adaptive else int(bool(speculative))
    o.prescaled_q = int(bool(prescaled_q))
    if ms is not None:
        o.ms = ctypes.pointer(ms)
    o.stats = stats_ptr
    return o


class FaError(RuntimeError):
    """Non-zero fa_status; message from fa_last_error() (RuntimeError like TORCH_CHECK)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


_lib = None


def load():
    """Load libfa_hip.so or fail loudly -- there is no CPU / eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C flash_attention_from_scratch_amd/csrc`. "
            "flash_attention has no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    cfg_p, args_p = ctypes.POINTER(FaFwdConfig), ctypes.POINTER(FaFwdArgs)
    lib.fa_init.restype = ctypes.c_int
    lib.fa_init.argtypes = []
    lib.fa_device_state.restype = ctypes.c_int
    lib.fa_device_state.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.fa_fwd_supported.restype = ctypes.c_int
    lib.fa_fwd_supported.argtypes = [cfg_p]
    lib.fa_fwd_lds_bytes.restype = ctypes.c_int
    lib.fa_fwd_lds_bytes.argtypes = [cfg_p]
    lib.fa_fwd_launch.restype = ctypes.c_int
    lib.fa_fwd_launch.argtypes = [args_p, ctypes.c_void_p]
    lib.fa_fwd_launch_timed.restype = ctypes.c_int
    lib.fa_fwd_launch_timed.argtypes = [args_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.fa_fwd_masked_supported.restype = ctypes.c_int
    lib.fa_fwd_masked_supported.argtypes = [cfg_p]
    lib.fa_fwd_launch_masked.restype = ctypes.c_int
    lib.fa_fwd_launch_masked.argtypes = [args_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.fa_fwd_ex_supported.restype = ctypes.c_int
    lib.fa_fwd_ex_supported.argtypes = [cfg_p, ctypes.POINTER(FaFwdOpts)]
    lib.fa_fwd_query.restype = ctypes.c_int
    lib.fa_fwd_query.argtypes = [cfg_p, ctypes.POINTER(FaFwdOpts), ctypes.POINTER(FaKernelInfo)]
    lib.fa_fwd_launch_ex.restype = ctypes.c_int
    lib.fa_fwd_launch_ex.argtypes = [args_p, ctypes.POINTER(FaFwdOpts), ctypes.c_void_p]
    lib.fa_num_kernels.restype = ctypes.c_int
    lib.fa_num_kernels.argtypes = []
    lib.fa_get_kernel.restype = ctypes.c_int
    lib.fa_get_kernel.argtypes = [ctypes.c_int, ctypes.POINTER(FaKernelInfo)]
    lib.fa_adaptive_state.restype = ctypes.c_int
    lib.fa_adaptive_state.argtypes = [ctypes.c_int, ctypes.POINTER(FaAdaptiveInfo)]
    lib.fa_adaptive_state_for.restype = ctypes.c_int
    lib.fa_adaptive_state_for.argtypes = [ctypes.c_int, ctypes.POINTER(FaFwdConfig), ctypes.POINTER(FaFwdOpts), ctypes.POINTER(FaAdaptiveInfo)]
    lib.fa_adaptive_reset.restype = ctypes.c_int
    lib.fa_adaptive_reset.argtypes = [ctypes.c_int]
    lib.fa_adaptive_simulate.restype = ctypes.c_int
    lib.fa_adaptive_simulate.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32),
                                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(FaAdaptiveInfo)]
    lib.fa_get_kernel_sized.restype = ctypes.c_int
    lib.fa_get_kernel_sized.argtypes = [ctypes.c_int, ctypes.POINTER(FaKernelInfo), ctypes.c_uint32]
    lib.fa_fwd_query_sized.restype = ctypes.c_int
    lib.fa_fwd_query_sized.argtypes = [cfg_p, ctypes.POINTER(FaFwdOpts), ctypes.POINTER(FaKernelInfo), ctypes.c_uint32]
    lib.fa_abi_version.restype = ctypes.c_int
    lib.fa_abi_version.argtypes = []
    lib.fa_last_error.restype = ctypes.c_char_p
    lib.fa_last_error.argtypes = []
    lib.fa_version.restype = ctypes.c_char_p
    lib.fa_version.argtypes = []
    _lib = lib
    return lib


def last_error():
    return load().fa_last_error().decode("utf-8", "replace")


def check(status):
    if status < 0:
        raise FaError(status, last_error())
    return status


def make_config(kernel_cfg) -> FaFwdConfig:
    """Read the 13 attributes by name, as the reference does (flash_attention.cu:16-32).
    `dtype` may be a DType (IntEnum) or anything int() accepts."""
    values = {}
    for name in CONFIG_FIELDS:
        values[name] = int(getattr(kernel_cfg, name))
    return FaFwdConfig(**values)


def supported(kernel_cfg) -> bool:
    cfg = make_config(kernel_cfg)
    return bool(load().fa_fwd_supported(ctypes.byref(cfg)))


def masked_supported(kernel_cfg) -> bool:
    cfg = make_config(kernel_cfg)
    return bool(load().fa_fwd_masked_supported(ctypes.byref(cfg)))


def ex_supported(kernel_cfg, **opts) -> bool:
    """fa_fwd_ex_supported: is there a device variant for the config with these fa_fwd_opts
    (causal, allow_ragged, speculative, prescaled_q)?"""
    cfg = make_config(kernel_cfg)
    o = make_opts(**opts)
    return bool(load().fa_fwd_ex_supported(ctypes.byref(cfg), ctypes.byref(o)))


def query(kernel_cfg, **opts) -> FaKernelInfo:
    """fa_fwd_query: the device variant that serves the config with these fa_fwd_opts (raises FaError if none)."""
    cfg = make_config(kernel_cfg)
    o = make_opts(**opts)
    info = FaKernelInfo()
    check(load().fa_fwd_query(ctypes.byref(cfg), ctypes.byref(o), ctypes.byref(info)))
    return info


def lds_bytes(kernel_cfg) -> int:
    cfg = make_config(kernel_cfg)
    return check(load().fa_fwd_lds_bytes(ctypes.byref(cfg)))


def kernels():
    """List of FaKernelInfo for every device variant in the library."""
    lib = load()
    out = []
    for i in range(lib.fa_num_kernels()):
        info = FaKernelInfo()
        check(lib.fa_get_kernel(i, ctypes.byref(info)))
        out.append(info)
    return out


def version() -> str:
    return load().fa_version().decode()


def device_state(device: int):
    """(inited, status, num_cus) of libfa_hip.so's per-device setup for `device` (fa_device_state)."""
    a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(load().fa_device_state(int(device), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return bool(a.value), b.value, c.value


def adaptive_state(device: int, kernel_cfg=None, **opts) -> dict:
    """fa_adaptive_state: the adaptive speculative mode's records on `device` taken together, as a dict; with
    `kernel_cfg` (and causal= / allow_ragged= / prescaled_q=) fa_adaptive_state_for: the record of the one device variant
    that serves that configuration (records are per variant since ABI 5)."""
    info = FaAdaptiveInfo()
    if kernel_cfg is None:
        check(load().fa_adaptive_state(int(device), ctypes.byref(info)))
    else:
        cfg = make_config(kernel_cfg)
        o = make_opts(**opts)
        check(load().fa_adaptive_state_for(int(device), ctypes.byref(cfg), ctypes.byref(o), ctypes.byref(info)))
    return {name: int(getattr(info, name)) for name, _ in FaAdaptiveInfo._fields_}


def adaptive_reset(device: int) -> None:
    check(load().fa_adaptive_reset(int(device)))


def adaptive_simulate(report_word, probe_state):
    """fa_adaptive_simulate: the adaptive policy on a fresh state, scripted -- launch i sees report_word[i] and
    probe_state[i] (-1 none, 0 pending, 1 complete, 2 error).  -> ([run_i], final state dict); run: 0 speculative,
    1 demoted, 2 probe."""
    n = len(report_word)
    assert len(probe_state) == n
    rep = (ctypes.c_uint32 * n)(*report_word)
    prb = (ctypes.c_int32 * n)(*probe_state)
    run = (ctypes.c_int32 * n)()
    info = FaAdaptiveInfo()
    check(load().fa_adaptive_simulate(n, rep, prb, run, ctypes.byref(info)))
    return list(run), {name: int(getattr(info, name)) for name, _ in FaAdaptiveInfo._fields_}
