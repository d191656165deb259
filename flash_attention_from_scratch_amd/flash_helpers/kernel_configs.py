"""Kernel-config surface of the MI355X build (torch-free, like the reference's).

Mirrors the public names of the reference's ``py/flash_helpers/kernel_configs.py``
(reference file:line cited per item) so tests, autotune and bench scripts written
against the reference keep working.  The 13-field key is unchanged; what the
fields *select* on CDNA4 is documented in DESIGN.md ("config -> device variant"):

* ``B_r`` / ``B_c``      Q rows per workgroup / keys per LDS tile.
* ``n_warps``            wave64 wavefronts per workgroup (64 lanes, not 32).
* ``async_copy``         K/V tiles by direct global->LDS DMA (the cp.async analogue, no
  registers) vs ``False``: through registers (coalesced loads a visit ahead, ds_write later).
* ``eager_load_blocks``  next K/V tile prefetched into the second LDS buffer.
* ``swizzled``           XOR-swizzled K image (conflict-free ds_read_b128).
* ``*_mma_load_K_tiles`` / ``mma_double_buffer_loads``  operand-fetch schedule
  hints; validated exactly like the reference, but the CDNA4 compiler schedules
  LDS->MFMA operand reads itself, so they map onto the same device code.
* ``optimized_softmax``  first K/V block skips the rescale of (l, O) -- the reference's meaning and
  nothing else.  The result is the same with or without it, so device variants whose schedule
  has no first-block rescale to skip ignore it (``softmax_mode(cfg)`` says what a config runs).

Extensions of this build live OUTSIDE the 13 fields, in ``NativeKernelConfig`` (a subclass):
``speculative_softmax`` and ``prescaled_q``; ``best_config()``
returns one.  A plain 13-field config never selects them (unless FA_ALLOW_SPECULATIVE=1 is set
in the environment, which maps ``optimized_softmax`` to the speculative softmax where it is built
-- the round-2 behaviour, kept for sweeps).
"""

import itertools
import os
import re
from dataclasses import dataclass, fields, replace
from enum import IntEnum

ELEM_SIZE = 2  # bytes per fp16/bf16 element


class DType(IntEnum):
    """torch ScalarType codes without importing torch (ref kernel_configs.py:9-13)."""

    FP16 = 5
    BF16 = 15

    def to_cpp_str(self) -> str:
        # ref kernel_configs.py:15-21 (kept for name-parsing round trips)
        return {DType.FP16: "torch::kFloat16", DType.BF16: "torch::kBFloat16"}[self]

    def to_torch_dtype(self):
        import torch

        return {DType.FP16: torch.float16, DType.BF16: torch.bfloat16}[self]

    def to_c_abi(self) -> int:
        """Value of ``fa_dtype`` in include/fa_hip.h (same integers)."""
        return int(self)

    @classmethod
    def from_string(cls, dtype_str: str) -> "DType":
        """'fp16' / 'BF16' / '5' / '15' -> DType (ref kernel_configs.py:34-55)."""
        text = dtype_str.strip()
        if re.fullmatch(r"[+-]?\d+", text):
            return cls(int(text))
        try:
            return cls[text.upper()]
        except KeyError:
            options = [f"{m.name} ({m.value})" for m in cls]
            raise ValueError(
                f"Invalid dtype string '{text.upper()}'. Valid options: {options}"
            ) from None


# ---------------------------------------------------------------------------
# FLOP / byte model (ref kernel_configs.py:61-103).  calc_self_attn_flop is the
# reference's headline convention; 4*B*H*S^2*d is the roofline figure.
# ---------------------------------------------------------------------------
def tile_softmax_flop(B_r, B_c, d_head) -> int:
    return B_r * (4 * B_c + d_head + 4)


def kv_tile_flop(B_r, B_c, d_head) -> int:
    matmuls = 2 * (2 * B_r * B_c * d_head)  # S = Q K^T and O += P V
    return matmuls + tile_softmax_flop(B_r, B_c, d_head)


def gmem_transfer_size(B_r, B_c, d_head) -> int:
    return 2 * d_head * (B_r + B_c) * ELEM_SIZE


def arithmetic_intensity(B_r, B_c, kv_seq_len, d_head) -> float:
    flop = kv_tile_flop(B_r, B_c, d_head) * (kv_seq_len // B_c)
    return flop / gmem_transfer_size(B_r, kv_seq_len, d_head)


def calc_total_flop(n_samples, n_heads, seq_len, B_r, B_c, d_head):
    if seq_len % B_r or seq_len % B_c:
        raise AssertionError("seq_len must be a multiple of B_r and B_c")
    q_blocks, kv_blocks = seq_len // B_r, seq_len // B_c
    per_q_block = kv_blocks * kv_tile_flop(B_r, B_c, d_head) + B_r * d_head
    return n_samples * n_heads * q_blocks * per_q_block


def calc_self_attn_flop(n_samples, n_heads, seq_len, d_head):
    return n_samples * n_heads * (4 * seq_len**2 * d_head + 6 * seq_len**2)


def calc_mfma_flop(n_samples, n_heads, seq_len, d_head):
    """Algorithmic matmul FLOPs 4*B*H*S^2*d -- the roofline numerator."""
    return 4 * n_samples * n_heads * seq_len**2 * d_head


_FLAG_WORDS = (
    ("async_copy", "async"),
    ("eager_load_blocks", "eager"),
    ("swizzled", "swizzled"),
)
_TAIL_WORDS = (
    ("mma_double_buffer_loads", "buffer"),
    ("optimized_softmax", "opt_softmax"),
)
# extensions of this build (NativeKernelConfig), spelled behind the reference's words
_NATIVE_WORDS = (
    ("speculative_softmax", "spec_softmax"),
    ("adaptive_softmax", "adaptive"),
    ("prescaled_q", "prescaled_q"),
)


@dataclass(frozen=True, order=True)
class FlashForwardKernelConfig:
    """13-field kernel key (ref kernel_configs.py:106-120, flash_attention.cuh:34-52)."""

    dtype: DType
    d_head: int
    B_r: int
    B_c: int
    n_warps: int
    async_copy: bool
    eager_load_blocks: bool
    swizzled: bool
    Q_mma_load_K_tiles: int
    K_mma_load_K_tiles: int
    V_mma_load_K_tiles: int
    mma_double_buffer_loads: bool
    optimized_softmax: bool

    def __str__(self):
        return self.short_form()

    def short_form(self, include_d_head=True, include_tup=True):
        """'(FP16, 128, 64, 64, 4): async+eager+...' (ref kernel_configs.py:125-147)."""
        words = [word for attr, word in _FLAG_WORDS if getattr(self, attr)]
        words.append(
            "load_%d_%d_%d_tiles"
            % (self.Q_mma_load_K_tiles, self.K_mma_load_K_tiles, self.V_mma_load_K_tiles)
        )
        words += [word for attr, word in _TAIL_WORDS if getattr(self, attr)]
        words += [word for attr, word in _NATIVE_WORDS if getattr(self, attr, False)]
        features = "+".join(words)
        if not include_tup:
            return features
        dims = [self.dtype.name]
        if include_d_head:
            dims.append(str(self.d_head))
        dims += [str(self.B_r), str(self.B_c), str(self.n_warps)]
        return f"({', '.join(dims)}): {features}"

    def to_cpp_struct(self) -> str:
        """Brace initialiser as the reference's generator writes it (ref :149-165)."""

        def lit(value):
            if isinstance(value, DType):
                return value.to_cpp_str()
            if isinstance(value, bool):
                return "true" if value else "false"
            return str(int(value))

        return "FlashForwardKernelConfig{%s}" % ", ".join(
            lit(getattr(self, f.name)) for f in fields(FlashForwardKernelConfig)
        )

    def to_c_abi_tuple(self):
        """The 13 ints of ``fa_fwd_config`` (include/fa_hip.h), in field order."""
        return tuple(int(getattr(self, f.name)) for f in fields(FlashForwardKernelConfig))

    def kernel_name(self) -> str:
        return "flash_forward_kernel"

    def smem_bytes(self, elem_size=ELEM_SIZE) -> int:
        """Reference formula (flash_attention.cuh:54-56); LDS use of the HIP kernel
        is reported by fa_fwd_lds_bytes() and differs (double-buffered K/V, Q in VGPRs)."""
        return (self.B_r + 2 * self.B_c) * self.d_head * elem_size

    def total_flop(self, n_samples: int, n_heads: int, seq_len: int) -> int:
        return calc_total_flop(n_samples, n_heads, seq_len, self.B_r, self.B_c, self.d_head)

    def attn_flop(self, n_samples: int, n_heads: int, seq_len: int) -> int:
        return calc_self_attn_flop(n_samples, n_heads, seq_len, self.d_head)

    def mfma_flop(self, n_samples: int, n_heads: int, seq_len: int) -> int:
        return calc_mfma_flop(n_samples, n_heads, seq_len, self.d_head)


@dataclass(frozen=True, order=True)
class NativeKernelConfig(FlashForwardKernelConfig):
    """The 13-field key plus this build's extensions, which the reference has no field for and which
    therefore travel beside ``fa_fwd_config`` in ``fa_fwd_opts`` (include/fa_hip.h):

    * ``speculative_softmax``  the item's first visited tile gives the reference max for all its tiles
      (no per-tile row max, no rescale); the row sums are checked at the end and a failed item is
      computed again with the running max.  Same real result; the cost of a failed
      item is 2x (``fa_fwd_stats.items_redone`` counts them).
    * ``prescaled_q``  logits from a 16-bit ``Q * log2(e)/sqrt(d)`` instead of an fp32 multiply per
      logit: not the reference's arithmetic, inside its tolerance rule.
    """

    speculative_softmax: bool = False
    prescaled_q: bool = False
    # with speculative_softmax: fa_speculative_mode ADAPTIVE (include/fa_hip.h) -- after a speculative launch on the device
    # has reported items it computed twice, the library serves the next launches of this config with its non-speculative
    # sibling for a while, then probes again.  Same tolerance either way; which of the two roundings a launch gets depends on
    # the device's recent record, so bit-reproducible callers leave it off.
    adaptive_softmax: bool = False

    def __post_init__(self):
        # adaptive_softmax qualifies speculative_softmax: without it there is nothing to adapt, and a construction that
        # asks for it alone is refused (rounds 4-5 cleared the flag silently, because best_config() carried it
        # and replace(best_config(...), speculative_softmax=False) had to work; the default no longer carries it)
        if self.adaptive_softmax and not self.speculative_softmax:
            raise ValueError("adaptive_softmax=True qualifies speculative_softmax=True (fa_speculative_mode ADAPTIVE); "
                             "it cannot be asked for on its own")

    def base(self) -> FlashForwardKernelConfig:
        """The plain 13-field config (what a reference user would pass)."""
        return FlashForwardKernelConfig(*(getattr(self, f.name) for f in fields(FlashForwardKernelConfig)))


def as_native(cfg, speculative_softmax=None, prescaled_q=None, adaptive_softmax=None) -> NativeKernelConfig:
    """`cfg` (plain or native) as a NativeKernelConfig with the given extensions (None = keep / off)."""
    base = [getattr(cfg, f.name) for f in fields(FlashForwardKernelConfig)]
    spec = getattr(cfg, "speculative_softmax", False) if speculative_softmax is None else speculative_softmax
    psq = getattr(cfg, "prescaled_q", False) if prescaled_q is None else prescaled_q
    ada = getattr(cfg, "adaptive_softmax", False) if adaptive_softmax is None else adaptive_softmax
    return NativeKernelConfig(*base, speculative_softmax=bool(spec), prescaled_q=bool(psq), adaptive_softmax=bool(ada) and bool(spec))


def config_sort_key(cfg):
    """Total order over plain and native configs (dataclass ordering refuses mixed classes)."""
    return cfg.to_c_abi_tuple() + (bool(getattr(cfg, "speculative_softmax", False)), bool(getattr(cfg, "prescaled_q", False)),
                                   bool(getattr(cfg, "adaptive_softmax", False)))


# ---------------------------------------------------------------------------
# Name parsers (ref kernel_configs.py:178-330): three spellings of a config.
# ---------------------------------------------------------------------------
_BRACES = re.compile(r"FlashForwardKernelConfig\{([^}]*)\}")
_N_FIELDS = len(fields(FlashForwardKernelConfig))


def _build(values):
    if len(values) != _N_FIELDS:
        raise ValueError("Incorrect number of parameters parsed")
    kw = {}
    for f, v in zip(fields(FlashForwardKernelConfig), values):
        if f.name == "dtype":
            kw[f.name] = v if isinstance(v, DType) else DType(int(v))
        elif f.type is bool or f.type == "bool":
            kw[f.name] = bool(int(v))
        else:
            kw[f.name] = int(v)
    return FlashForwardKernelConfig(**kw)


def _parse_flash_forward_demanged_name(line) -> FlashForwardKernelConfig:
    """'...FlashForwardKernelConfig{5, 128, 64, 64, 4, 1, 1, 1, 0, 2, 0, 1, 1}...'"""
    found = _BRACES.search(line)
    if not found:
        raise ValueError("Invalid line format: FlashForwardKernelConfig not found")
    words = {"true": 1, "false": 0}
    values = []
    for token in found.group(1).split(","):
        token = token.strip()
        values.append(words[token] if token in words else int(token))
    return _build(values)


def _parse_flash_forward_demanged_name_with_types(line: str) -> FlashForwardKernelConfig:
    """'...{(c10::ScalarType)5, (int)128, ..., (bool)1}...' (typed demangling)."""
    found = _BRACES.search(line)
    if not found:
        raise ValueError("Invalid line format: FlashForwardKernelConfig block not found")
    values = []
    for token in found.group(1).split(","):
        cast = re.fullmatch(r"\s*\((c10::ScalarType|int|bool)\)\s*(\S+)\s*", token)
        if not cast:
            raise ValueError(f"Unexpected parameter format: {token.strip()}")
        kind, text = cast.groups()
        values.append(DType.from_string(text) if kind == "c10::ScalarType" else int(text))
    return _build(values)


_SHORT = re.compile(r"\(([^)]*)\):\s*([^|\s]+)")


def _parse_short_form_flash_forward_kernel_config(line: str) -> FlashForwardKernelConfig:
    """'(FP16, 128, 64, 64, 4): async+eager+swizzled+load_0_2_2_tiles+opt_softmax'
    optionally embedded in a '|'-separated table row."""
    cells = [c.strip() for c in line.split("|") if c.strip()]
    if not cells:
        raise ValueError(f"Cannot parse line (empty after splitting): {line}")
    found = _SHORT.match(cells[0])
    if not found:
        raise ValueError(f"Cannot parse config string (no matching pattern): {cells[0]}")
    dims = [d.strip() for d in found.group(1).split(",")]
    if len(dims) != 5:
        raise ValueError(f"Cannot parse config tuple: {found.group(1)}")
    words = found.group(2).split("+")
    loads = [w for w in words if w.startswith("load_") and w.endswith("_tiles")]
    if not loads:
        raise ValueError(f"Cannot find load segment in features: {found.group(2)}")
    q_t, k_t, v_t = (int(x) for x in loads[0][len("load_"):-len("_tiles")].split("_"))
    flags = {attr: (word in words) for attr, word in _FLAG_WORDS + _TAIL_WORDS}
    native = {attr: (word in words) for attr, word in _NATIVE_WORDS}
    cls = NativeKernelConfig if any(native.values()) else FlashForwardKernelConfig
    if cls is NativeKernelConfig:
        flags.update(native)
    return cls(
        dtype=DType.from_string(dims[0]),
        d_head=int(dims[1]),
        B_r=int(dims[2]),
        B_c=int(dims[3]),
        n_warps=int(dims[4]),
        Q_mma_load_K_tiles=q_t,
        K_mma_load_K_tiles=k_t,
        V_mma_load_K_tiles=v_t,
        **flags,
    )


def parse_kernel_name_into_config(kernel_name: str) -> FlashForwardKernelConfig:
    for parser in (
        _parse_flash_forward_demanged_name,
        _parse_flash_forward_demanged_name_with_types,
        _parse_short_form_flash_forward_kernel_config,
    ):
        try:
            return parser(kernel_name)
        except (ValueError, KeyError):
            continue
    raise ValueError(f"Invalid kernel name: {kernel_name}")


# Names a rocprofv3 kernel trace shows for the comparators this build benches
# against (the reference maps Dao-AILab FA2/FA3 CUDA symbols here, :333-341;
# those wheels do not exist on ROCm -- torch SDPA is the comparator).
REF_KERNEL_NAME_MAP = {
    "torch.sdpa": "Reference",
}


def transform_kernel_name_to_short_form(kernel_name: str) -> str:
    if kernel_name in REF_KERNEL_NAME_MAP:
        return REF_KERNEL_NAME_MAP[kernel_name]
    return parse_kernel_name_into_config(kernel_name).short_form()


def transform_kernel_name(kernel_name: str) -> str:
    try:
        return parse_kernel_name_into_config(kernel_name).short_form()
    except ValueError:
        return kernel_name


# ---------------------------------------------------------------------------
# Enumerations (ref kernel_configs.py:364-485)
# ---------------------------------------------------------------------------
def should_autotune_config(cfg: FlashForwardKernelConfig) -> bool:
    """Same pruning rule as the reference (:364-386) so the drop-in list is identical."""
    if cfg.eager_load_blocks and not cfg.async_copy:
        return False
    q_t, k_t = cfg.Q_mma_load_K_tiles, cfg.K_mma_load_K_tiles
    if q_t not in (0, k_t):
        return False
    if cfg.B_r == 64:
        if cfg.n_warps == 8:
            return False
        if cfg.B_c == 32 and q_t == 0:
            return False
        if cfg.B_c == 64 and q_t != 0:
            return False
    elif cfg.B_r == 128 and q_t == 0:
        return False
    return True


def get_autotuning_kernel_configs(dtypes=(DType.BF16, DType.FP16)):
    """The reference's 80-config sweep (40 per dtype), same order (:389-423)."""
    axes = [
        list(dtypes),
        [128],  # d_head
        [64, 128],  # B_r
        [32, 64],  # B_c
        [4],  # n_warps
        [True],  # async_copy
        [True],  # eager_load_blocks
        [True],  # swizzled
        [0, 2],  # Q_mma_load_K_tiles
        [0, 2],  # K_mma_load_K_tiles
        [0, 2],  # V_mma_load_K_tiles
        [False, True],  # mma_double_buffer_loads
        [False, True],  # optimized_softmax
    ]
    candidates = (FlashForwardKernelConfig(*point) for point in itertools.product(*axes))
    return [cfg for cfg in candidates if should_autotune_config(cfg)]


_PROGRESSION = (
    "async+load_0_0_0_tiles",
    "async+swizzled+load_0_0_0_tiles",
    "async+eager+swizzled+load_0_0_0_tiles",
    "async+eager+swizzled+load_2_2_2_tiles",
    "async+eager+swizzled+load_2_2_2_tiles+buffer",
    "async+eager+swizzled+load_2_2_2_tiles+buffer+opt_softmax",
    "async+eager+swizzled+load_0_2_2_tiles+opt_softmax",
)


def get_kernel_progression_configs(all_block_sizes=False):
    """The reference's 7-step story at (FP16, 128, 64, 64, 4) (:426-455)."""
    steps = [
        _parse_short_form_flash_forward_kernel_config(f"(FP16, 128, 64, 64, 4): {s}")
        for s in _PROGRESSION
    ]
    if not all_block_sizes:
        return steps
    out = []
    for step in steps:
        for B_r, B_c, n_warps in itertools.product([128, 64], [128, 64, 32], [4, 8]):
            if (B_r < 128 and n_warps == 8) or B_r < B_c:
                continue
            out.append(replace(step, B_r=B_r, B_c=B_c, n_warps=n_warps))
    return out


def get_native_kernel_configs(dtypes=(DType.BF16, DType.FP16)):
    """CDNA4-native tile shapes beyond the reference's list: wave64 workgroups of
    4 or 8 waves, 32 or 64 Q rows per wave, 64/128-key LDS tiles.  These are the
    shapes the MI355X autotune sweeps (KERNELS=native / KERNELS=tune)."""
    shapes = [
        # (B_r, B_c, n_waves)
        (128, 64, 4),
        (128, 128, 4),
        (256, 64, 8),
        (256, 128, 8),
        (256, 32, 8),
        (256, 64, 4),
    ]
    # async_copy: True = global->LDS DMA, False = through registers (32 rows/wave shapes)
    reg_staged = {(128, 64, 4), (256, 64, 8), (128, 128, 4), (256, 128, 8)}
    out = []
    for dtype in dtypes:
        for B_r, B_c, n_waves in shapes:
            for dma in (True, False):
                if not dma and (B_r, B_c, n_waves) not in reg_staged:
                    continue
                for pipelined in (False, True):
                    if pipelined and not dma and n_waves == 4:
                        continue  # 4 landing registers per tile kind: the pipelined loop would spill
                    for second in (False, True):
                        # LDS-DMA shapes: with / without the speculative softmax (a native extension);
                        # register-staged shapes: with / without the reference's first-block skip
                        base = (dtype, 128, B_r, B_c, n_waves, dma, True, True, 0, 0, 0, pipelined)
                        if dma and second:
                            out.append(NativeKernelConfig(*base, False, speculative_softmax=True))
                        else:
                            out.append(FlashForwardKernelConfig(*base, second))
                        if (B_r, B_c, n_waves, dma, pipelined) == (256, 64, 4, True, True):
                            # the persistent kernel with the pre-scaled Q, with / without the speculative softmax
                            out.append(NativeKernelConfig(*base, False, speculative_softmax=second, prescaled_q=True))
    return out


def get_d64_kernel_configs(dtypes=(DType.BF16, DType.FP16)):
    """d_head = 64 (a scope widener; the reference's config comments allow [64, 128] but only
    128 was ever built): the 32-rows-per-wave kernel at three tile shapes."""
    out = []
    for dtype in dtypes:
        for B_r, B_c, n_waves, pipes in ((128, 64, 4, (False, True)), (256, 64, 8, (False, True)),
                                         (256, 128, 8, (False,))):
            for pipelined in pipes:
                base = (dtype, 64, B_r, B_c, n_waves, True, True, True, 0, 0, 0, pipelined, False)
                out.append(FlashForwardKernelConfig(*base))
                out.append(NativeKernelConfig(*base, speculative_softmax=True))
    return out


def get_speculative_reference_shape_configs(dtypes=(DType.BF16, DType.FP16)):
    """The speculative softmax on the REFERENCE's tile shapes ((64,32), (64,64), (128,32), (128,64), 4 waves):
    native configs (`speculative_softmax`), one per device variant -- with and without the pipelined loop."""
    out = set()
    for cfg in get_autotuning_kernel_configs(dtypes):
        if cfg.optimized_softmax and has_speculative_variant(cfg):
            out.add(as_native(replace(cfg, optimized_softmax=False, Q_mma_load_K_tiles=0, K_mma_load_K_tiles=0,
                                      V_mma_load_K_tiles=0), speculative_softmax=True))
    return sorted(out, key=config_sort_key)


def get_kernels_to_build():
    """The drop-in list: exactly the reference's built set (:458-463)."""
    return sorted(set(get_autotuning_kernel_configs()), key=config_sort_key)


def get_all_supported_configs():
    """Everything libfa_hip.so accepts that these helpers can enumerate."""
    cfgs = set(get_autotuning_kernel_configs())
    cfgs.update(get_native_kernel_configs())
    cfgs.update(get_d64_kernel_configs())
    cfgs.update(get_kernel_progression_configs())
    cfgs.update(get_speculative_reference_shape_configs())
    return sorted(cfgs, key=config_sort_key)


def get_kernel_configs(kernels_key=""):
    """KERNELS env selector (ref :466-485) plus 'native' for the CDNA4 shapes."""
    if kernels_key == "":
        kernels_key = os.environ.get("KERNELS", "")
    if kernels_key.startswith("prog"):
        return get_kernel_progression_configs(all_block_sizes="all" in kernels_key)
    if kernels_key == "all":
        return get_kernels_to_build()
    if kernels_key == "tune":
        return get_autotuning_kernel_configs()
    if kernels_key == "native":
        return get_native_kernel_configs()
    if kernels_key == "d64":
        return get_d64_kernel_configs()
    if kernels_key == "best":
        return [best_config(DType.BF16), best_config(DType.FP16)]
    if "," in kernels_key:
        B_r, B_c = (int(x) for x in kernels_key.split(","))
        pool = get_autotuning_kernel_configs() + get_native_kernel_configs()
        return [cfg for cfg in pool if (cfg.B_r, cfg.B_c) == (B_r, B_c)]
    raise ValueError(f"Invalid kernels env key: {kernels_key}")


def is_persistent_shape(cfg) -> bool:
    """The config served by the persistent 64-rows-per-wave kernel."""
    return (cfg.d_head, cfg.B_r, cfg.B_c, cfg.n_warps, bool(cfg.mma_double_buffer_loads)) == (128, 256, 64, 4, True) \
        and bool(cfg.async_copy and cfg.eager_load_blocks and cfg.swizzled)


def uses_lazy_rescale(cfg, seq_len=None) -> bool:
    """True for the config served by the hand-placed persistent device schedule, whose softmax moves
    its reference max lazily (CPU restatement: oracle blockwise_forward_lazy): the 64-rows-per-wave
    shape, and -- given a ``seq_len`` that is a multiple of 256 -- the ring form of (128, 64, 4) + buffer
    (``has_ring_form``; consistent with ``softmax_mode(cfg, seq_len=...)``)."""
    if (cfg.d_head, cfg.B_r, cfg.B_c, cfg.n_warps, bool(cfg.mma_double_buffer_loads)) == (128, 256, 64, 4, True):
        return True
    return seq_len is not None and seq_len % 256 == 0 and has_ring_form(cfg)


def has_speculative_variant(cfg, masked=False) -> bool:
    """Mirror of the device-side predicates (fa_registry.hpp softmax_mode_of): the speculative softmax is
    built on the persistent kernel (plain, causal and ragged forms) and on every double-buffered LDS-DMA
    variant of the other kernels WITHOUT a mask (their masked forms keep the running max)."""
    if is_persistent_shape(cfg):
        return True
    return bool(cfg.eager_load_blocks and cfg.async_copy and not masked)


def wants_speculative(cfg) -> bool:
    """Does this config ASK for the speculative softmax?  Only a NativeKernelConfig can
    (``speculative_softmax``); a plain 13-field config asks for the reference's arithmetic -- unless
    FA_ALLOW_SPECULATIVE=1 maps its ``optimized_softmax`` to it (round 2's behaviour, for sweeps)."""
    if getattr(cfg, "speculative_softmax", False):
        return True
    return bool(cfg.optimized_softmax) and os.environ.get("FA_ALLOW_SPECULATIVE", "") == "1"


def has_ring_form(cfg, masked=False) -> bool:
    """(B_r 128, B_c 64, 4 warps) + buffer, the reference's own winning tile shape (kernel_configs.py:389-423 there,
    kernel_sass/16_A100.asm:5): launches with ``seq_len % 256 == 0`` run the hand-placed persistent kernel with one 32-row
    Q tile per wave (``fa_kernel_info.ring_form``), the other multiples of 128 the compiler-scheduled body.
    The plain configuration runs that kernel's lazy rescale, the speculative one its speculative schedule
    (``fa_kernel_info.ring_softmax_mode``).  Not the masked forms.  Round 6: the ring form runs EIGHT waves -- two of the
    shape's 128-row Q blocks fused into one 256-row item whose waves share one set of K / V rings (two waves per SIMD); the
    bits are the four-wave form's, and ``fa_fwd_stats`` still counts 128-row Q blocks (a redone item counts as two)."""
    return (not masked and cfg.d_head == 128 and (cfg.B_r, cfg.B_c, cfg.n_warps) == (128, 64, 4) and bool(cfg.async_copy)
            and bool(cfg.eager_load_blocks) and bool(cfg.swizzled) and bool(cfg.mma_double_buffer_loads))


def softmax_mode(cfg, masked=False, seq_len=None) -> str:
    """What the device variant behind ``cfg`` does to keep exp2 in range -- the Python mirror of
    ``fa_kernel_info.softmax_mode`` (include/fa_hip.h): 'eager' (the reference, softmax.cuh:85-105),
    'first_block_skip' (the reference's optimized_softmax), 'lazy' (persistent kernel: the reference max
    moves only when a row max rose by more than 8 binades), 'speculative' (DESIGN.md 3.6).  ``seq_len``: where the
    config has a ring form (``has_ring_form``) and the length is a multiple of 256, that form's mode ('lazy', or
    'speculative' where the configuration asks for it: ``fa_kernel_info.ring_softmax_mode``)."""
    if wants_speculative(cfg) and has_speculative_variant(cfg, masked):
        return "speculative"
    if is_persistent_shape(cfg):
        return "lazy"
    if seq_len is not None and seq_len % 256 == 0 and has_ring_form(cfg, masked):
        return "lazy"
    if cfg.optimized_softmax and not has_speculative_variant(cfg, masked):
        return "first_block_skip"  # (where OPT builds the speculative schedule there is no first-block variant: ignored)
    return "eager"


SOFTMAX_MODES = ("eager", "first_block_skip", "lazy", "speculative")  # index = fa_softmax_mode


def walks_kv_forward(cfg, masked=False, seq_len=None) -> bool:
    """Round 6: the speculative FIRST PASS of the hand-placed persistent kernel (the 64-rows-per-wave shape, and the ring
    form of (128, 64, 4) + buffer at ``seq_len % 256 == 0``) visits an item's K / V tiles first-to-last -- its reference is
    the row max of the first tile it visits, and attention sinks sit at the first keys.  Every other
    variant, the masked forms and the second pass keep the reference's last-to-first order (forward_kernel.cuh:142)."""
    if masked or not (wants_speculative(cfg) and has_speculative_variant(cfg, masked)):
        return False
    if is_persistent_shape(cfg):
        return True
    return seq_len is not None and seq_len % 256 == 0 and has_ring_form(cfg, masked)


def kv_walk_alternates(cfg, n_bh, seq_len, masked=False, num_cus=256) -> int:
    """Round 6, long sequences: the 64-rows-per-wave speculative plain kernel has a form whose first pass walks every second
    round of a head's Q blocks as [tile 0, then last-to-second] (fa_fwd_kernel64<..., ALT>, KernelEntry::fn_alt; DESIGN.md
    3.5), so that the K / V tail the round before left in the XCD's L2 is read again first.  The launcher takes it when
    batch * heads is a multiple of 8 and a head's Q blocks fill an even number of rounds of an XCD's workgroups.
    -> G, the workgroups per XCD (Q block qb walks that way when (qb // G) is odd), or 0 when every item walks first-to-last.
    A function of seq_len and the device's CU count alone: an item's bits never depend on the batch it sits in."""
    if masked or getattr(cfg, "prescaled_q", False) or not is_persistent_shape(cfg) or not walks_kv_forward(cfg, masked, seq_len):
        return 0
    n_q = seq_len // 256
    n_wg = min(n_bh * n_q, (num_cus & ~7) or 8)
    g = n_wg >> 3
    return g if (n_bh % 8 == 0 and n_wg >= 16 and seq_len % 256 == 0 and n_q % (2 * g) == 0) else 0


def uses_speculative_softmax(cfg, masked=False) -> bool:
    """True where the config runs the speculative softmax: an item is first run against
    the row max of its first K/V tile only (no per-tile row max, no rescale), its row sums are checked
    against an overflow limit at the end, and an item that fails is run again with the running max -- by
    the lazy-rescale schedule on the persistent 64-rows-per-wave kernel, by the same workgroup starting
    over on the other kernels.  CPU restatement: blockwise_forward_lazy with an infinite threshold.
    ``masked``: the causal / ragged form of the config (forward_ex) -- only the persistent kernel's
    masked forms are speculative."""
    return softmax_mode(cfg, masked) == "speculative"


def best_config(dtype=DType.BF16, seq_len=4096, masked=False) -> FlashForwardKernelConfig:
    """Autotune winner on MI355X (profiles/): the persistent hand-placed
    64-rows-per-wave kernel (4 waves x 64 rows, 64-key tiles, one wave per SIMD: each LDS operand
    feeds two MFMAs; one workgroup per CU walks the (batch*head, Q block) items) whenever seq_len is a
    multiple of its 256-row Q block; otherwise the pipelined 4-wave x 32-row kernel.
    masked=True: the best config that has a causal / ragged-length variant (forward_ex): the same
    persistent kernel when seq_len is a multiple of 256 (causal form) or when rounding seq_len up to
    one costs at most an eighth more rows (its ragged form works on whole 256-row Q blocks and whole
    rounds of four 64-key tiles; measured ahead of the 32-rows-per-wave kernels from seq_len ~1000 up,
    profiles/r01/ragged_persistent.txt), otherwise 4 waves x 32 rows.

    Softmax: the speculative softmax (``speculative_softmax``; fa_fwd_opts.speculative = 1), for both dtypes, and nothing
    else: since round 6 the default keeps NO host-side state -- same inputs, same bits, from any process, thread, stream
    or launch history, like the reference's launcher (src/flash_attention.cu:42,118,126-131).  What made that possible is on
    the device: the first pass of the persistent kernel walks an item's keys first-to-last, so
    attention-sink keys at the start of a sequence ARE its reference instead of arriving last, 17 binades above it (fp16:
    every item used to be computed twice); it re-centres rising rows every four visits, so only a JUMP of ~83 nats (bf16) /
    ~1.4-10 nats (fp16) inside 256 keys sends an item to the second pass.  The adaptive mode of rounds 4-5
    (``adaptive_softmax=True``: the library watches failure reports and serves the running-max variant for a while) is
    still there, opt-in; ``speculative_softmax=False`` is the running-max (lazy) kernel, the reference's arithmetic
    family."""
    dtype = DType(dtype)
    pad = (-seq_len) % 256
    if pad == 0 or (masked and seq_len >= 64 and pad * 8 <= seq_len):
        return NativeKernelConfig(
            dtype, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False, speculative_softmax=True
        )
    return NativeKernelConfig(
        dtype, 128, 128, 64, 4, True, True, True, 0, 0, 0, True, False, speculative_softmax=not masked
    )
