"""CPU tier: host logic, config surface and the C-ABI library's exports (no compute)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from flash_attention_from_scratch_amd import _capi
from dataclasses import fields, replace  # noqa: F401
from flash_helpers import kernel_configs as kc
from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fa_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fa_[a-z_0-9]+)\s*\(", body))
    assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = set(re.findall(r" T (fa_[a-z_0-9]+)", nm.stdout))
        assert declared <= exported
    assert "gfx950" in _capi.version()


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_capi.FaFwdConfig) == 13 * 4
    assert ctypes.sizeof(_capi.FaFwdArgs) == 4 * 8 + 7 * 8 + 13 * 4 + 4  # tail padding to 8
    assert ctypes.sizeof(_capi.FaKernelInfo) == 13 * 4 + 8 * 4 + 4 * 4 + 4 * 4   # (+ the four ring_* fields of ABI 5, + ring_lds_bytes, persistent, alt_form, ring_threads of ABI 6)
    assert ctypes.sizeof(_capi.FaFwdStats) == 8
    assert ctypes.sizeof(_capi.FaFwdOpts) == 5 * 4 + 4 + 2 * 8   # five 32-bit fields, padding, two pointers
    # the header's own view, compiled: sizes and offsets of the structs ctypes mirrors
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "fa_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(fa_fwd_config), sizeof(fa_fwd_args), sizeof(fa_kernel_info),
           sizeof(fa_fwd_stats), sizeof(fa_fwd_opts), offsetof(fa_fwd_opts, ms), offsetof(fa_fwd_opts, stats),
           offsetof(fa_kernel_info, softmax_mode), offsetof(fa_fwd_opts, prescaled_q));
    return 0;
}'''
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(tmp, "t.c"), "-o", os.path.join(tmp, "t")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(tmp, "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [ctypes.sizeof(_capi.FaFwdConfig), ctypes.sizeof(_capi.FaFwdArgs), ctypes.sizeof(_capi.FaKernelInfo),
                   ctypes.sizeof(_capi.FaFwdStats), ctypes.sizeof(_capi.FaFwdOpts), _capi.FaFwdOpts.ms.offset,
                   _capi.FaFwdOpts.stats.offset, _capi.FaKernelInfo.softmax_mode.offset, _capi.FaFwdOpts.prescaled_q.offset]


def test_every_enumerated_config_has_a_device_kernel():
    for cfg in kc.get_all_supported_configs():
        assert _capi.supported(cfg), cfg
        lds = _capi.lds_bytes(cfg)
        stages = 2 if cfg.eager_load_blocks else 1
        if kc.uses_lazy_rescale(cfg):
            # the persistent 64-rows-per-wave schedule: 4-stage K and V rings plus 8 KB of O staging
            # per wave beside them = the whole 160 KB
            assert lds == 2 * 4 * cfg.B_c * cfg.d_head * 2 + cfg.n_warps * 32 * cfg.d_head * 2 == 160 * 1024
            continue
        # K/V stages, or the O tile staged through LDS in the epilogue, whichever is larger
        assert lds == max(2 * stages * cfg.B_c * cfg.d_head * 2, cfg.B_r * cfg.d_head * 2)
        assert lds <= 160 * 1024


def test_registry_enumeration_is_consistent():
    infos = _capi.kernels()
    assert len(infos) >= 40
    seen = set()
    for info in infos:
        key = tuple(getattr(info.cfg, f) for f in _capi.CONFIG_FIELDS)
        ident = (key, info.masked, info.softmax_mode, info.prescaled_q)
        assert ident not in seen
        seen.add(ident)
        assert 0 <= info.softmax_mode < 4
        # the canonical config's optimized_softmax has the reference's meaning only
        assert bool(info.cfg.optimized_softmax) == (_capi.SOFTMAX_MODES[info.softmax_mode] == "first_block_skip")
        assert info.threads == 64 * info.cfg.n_warps
        assert info.rows_per_wave * info.cfg.n_warps == info.cfg.B_r
        assert info.rows_per_wave in (16, 32, 64)
        assert info.cfg.d_head in (64, 128)
        cfg = kc.FlashForwardKernelConfig(kc.DType(info.cfg.dtype), *key[1:5], *map(bool, key[5:8]),
                                          *key[8:11], *map(bool, key[11:13]))
        assert _capi.supported(cfg)


def test_one_flag_one_meaning_softmax_mode_of_every_config():
    """`optimized_softmax` keeps the reference's meaning (softmax.cuh:85-105, forward_kernel.cuh:158-161) through the C
    ABI: a plain 13-field config NEVER reaches a speculative variant; the speculative softmax and the pre-scaled Q are
    asked for in fa_fwd_opts (a NativeKernelConfig on the Python side).  kc.softmax_mode() is the Python mirror of
    fa_kernel_info.softmax_mode: checked here for every enumerable config, plain and masked, against the registry's
    own lookup (fa_fwd_ex_supported) and the enumeration (fa_get_kernel)."""
    from dataclasses import replace

    n_spec = 0
    for cfg in kc.get_all_supported_configs():
        for masked in (False, True):
            if not _capi.ex_supported(cfg, allow_ragged=masked):
                assert masked
                continue
            wants = kc.wants_speculative(cfg)
            has = _capi.ex_supported(cfg, allow_ragged=masked, speculative=True)
            assert has == kc.has_speculative_variant(cfg, masked) or not has, (str(cfg), masked)
            if wants and not has:
                assert masked  # every enumerated native config has its plain speculative variant
                continue
            info = _capi.query(cfg, allow_ragged=masked, speculative=wants)
            mode = kc.softmax_mode(cfg, masked)
            assert mode == _capi.SOFTMAX_MODES[info.softmax_mode], (str(cfg), masked, mode, info.softmax_mode)
            assert (mode == "speculative") == wants
            n_spec += wants
            if not isinstance(cfg, kc.NativeKernelConfig):
                assert mode != "speculative"
                # the flag alone never changes which arithmetic family runs
                other = kc.softmax_mode(replace(cfg, optimized_softmax=not cfg.optimized_softmax), masked)
                assert {mode, other} <= {"eager", "first_block_skip"} or mode == other == "lazy"
    assert n_spec >= 30
    os.environ["FA_ALLOW_SPECULATIVE"] = "1"
    try:
        cfg = kc.get_kernel_configs("tune")[-1]
        assert cfg.optimized_softmax and kc.softmax_mode(cfg) == "speculative"   # the round-2 mapping, opt-in
    finally:
        del os.environ["FA_ALLOW_SPECULATIVE"]
    assert kc.softmax_mode(cfg) == "eager"
    # best_config: the speculative softmax for both dtypes, STATELESS since round 6 (fa_fwd_opts.speculative = 1; the
    # adaptive mode of rounds 4-5 is opt-in: DESIGN.md 3.6)
    for dt in (kc.DType.BF16, kc.DType.FP16):
        best = kc.best_config(dt)
        assert kc.softmax_mode(best) == "speculative" and best.speculative_softmax and not best.adaptive_softmax
        assert kc.parse_kernel_name_into_config(best.short_form()) == best and "+adaptive" not in best.short_form()
        ada = replace(best, adaptive_softmax=True)
        assert ada.short_form() == best.short_form() + "+adaptive" and kc.parse_kernel_name_into_config(ada.short_form()) == ada
        assert kc.walks_kv_forward(best) and not kc.walks_kv_forward(best, masked=True)
    # "never speculative" from the default in ONE step; adaptive on its own is refused
    never = replace(best, speculative_softmax=False)
    assert not never.adaptive_softmax and kc.softmax_mode(never) == "lazy" and not kc.walks_kv_forward(never)
    with pytest.raises(ValueError):
        kc.NativeKernelConfig(*(getattr(best, f.name) for f in fields(kc.FlashForwardKernelConfig)), adaptive_softmax=True)
    assert kc.softmax_mode(kc.best_config(kc.DType.BF16, 1000, masked=True), masked=True) == "speculative"
    # the ring form (round 5): (B_r 128, B_c 64, 4 warps) + buffer, plain -- the mirror of fa_kernel_info.ring_form
    ring = [c for c in kc.get_kernels_to_build() if kc.has_ring_form(c)]
    assert len(ring) == 8 and all((c.B_r, c.B_c, c.n_warps) == (128, 64, 4) and c.mma_double_buffer_loads for c in ring)   # four per dtype
    infos = [_capi.query(c) for c in ring]
    assert all(i.ring_form == 1 and i.ring_softmax_mode == 2 and i.softmax_mode in (0, 1) for i in infos)
    assert all(i.ring_threads == 512 and i.ring_lds_bytes == 163840 and i.threads == 256 for i in infos)   # (round 6: eight waves)
    others = [c for c in kc.get_kernels_to_build() if not kc.has_ring_form(c)]
    assert all(_capi.query(c).ring_form == 0 for c in others)
    assert kc.softmax_mode(ring[0], seq_len=1024) == "lazy" and kc.softmax_mode(ring[0], seq_len=640) != "lazy" and kc.softmax_mode(ring[0]) != "lazy"
    assert not kc.has_ring_form(ring[0], masked=True)
    # ... and its speculative sibling (asked for through fa_fwd_opts): the same kernel's speculative schedule
    spec = kc.as_native(ring[0], speculative_softmax=True)
    assert kc.has_ring_form(spec) and kc.softmax_mode(spec, seq_len=1024) == kc.softmax_mode(spec, seq_len=640) == "speculative"
    i = _capi.query(ring[0], speculative=1)
    assert i.ring_form == 1 and i.ring_softmax_mode == 3 and i.softmax_mode == 3
    assert kc.softmax_mode(kc.best_config(kc.DType.BF16, 100, masked=True), masked=True) == "eager"


def test_opts_struct_size_is_checked():
    lib = _capi.load()
    cfg = kc.best_config(kc.DType.BF16)
    args = _args(cfg)
    bad = _capi.FaFwdOpts()  # struct_size left 0
    assert lib.fa_fwd_launch_ex(ctypes.byref(args), ctypes.byref(bad), None) == -4 and "struct_size" in _capi.last_error()
    o = _capi.make_opts(prescaled_q=True)
    o_lazy = _capi.make_opts()
    c = _capi.make_config(cfg)
    assert lib.fa_fwd_ex_supported(ctypes.byref(c), ctypes.byref(o_lazy)) == 1
    assert lib.fa_fwd_ex_supported(ctypes.byref(c), None) == 1
    short = _capi.make_opts(speculative=True)
    short.struct_size = 16   # an older caller whose header ended behind `speculative`
    assert lib.fa_fwd_ex_supported(ctypes.byref(c), ctypes.byref(short)) == 1


def test_launch_ex_validation_needs_no_gpu():
    """fa_fwd_launch_ex validates like fa_fwd_launch (before any HIP call) and names what is missing when a native option
    has no device variant: the pre-scaled Q exists on the persistent kernel's plain form only, the speculative softmax
    not on the masked 32-rows-per-wave variants."""
    lib = _capi.load()
    persistent = kc.best_config(kc.DType.BF16)
    small = kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 128, 64, 4, True, True, True, 2, 2, 0, True, False)

    def status(cfg, seq=256, **opts):
        args = _args(cfg, seq=seq)
        o = _capi.make_opts(**opts)
        return lib.fa_fwd_launch_ex(ctypes.byref(args), ctypes.byref(o), None), _capi.last_error()

    rc, msg = status(persistent, prescaled_q=True, causal=True)
    assert rc == -3 and "requested options" in msg
    rc, msg = status(small, prescaled_q=True)
    assert rc == -3 and "requested options" in msg
    rc, msg = status(small, speculative=True, allow_ragged=True)
    assert rc == -3
    rc, msg = status(small, seq=200)                       # not a multiple of the tiles, and no allow_ragged
    assert rc == -4 and msg == "Only multiples of B_r are supported for seq_len Q currently"
    rc, msg = status(small, seq=200, causal=True)          # causal alone does not buy ragged lengths either
    assert rc == -4 and msg == "Only multiples of B_r are supported for seq_len Q currently"
    rc, msg = status(small, seq=192, causal=True)          # (a multiple of B_c, not of B_r)
    assert rc == -4 and "B_r" in msg
    args = _args(persistent)
    o = _capi.make_opts(speculative=True, stats_ptr=4098)  # a misaligned device pointer for the counters
    assert lib.fa_fwd_launch_ex(ctypes.byref(args), ctypes.byref(o), None) == -5 and "stats" in _capi.last_error()
    # everything valid: the next step is the device, which this box does not have
    if not torch.cuda.is_available():
        for cfg, opts in ((persistent, dict(speculative=True)), (persistent, dict(speculative=True, prescaled_q=True)),
                          (small, dict(allow_ragged=True, causal=True))):
            rc, msg = status(cfg, **opts)
            assert rc == -7 and "no HIP device" in msg, (rc, msg)
    for cfg, opts, mode in ((persistent, dict(speculative=True), "speculative"), (persistent, {}, "lazy"),
                            (small, {}, "eager"), (small, dict(speculative=True), "speculative"),
                            (small, dict(allow_ragged=True), "eager")):
        assert _capi.SOFTMAX_MODES[_capi.query(cfg, **opts).softmax_mode] == mode
    assert _capi.query(persistent, speculative=True, prescaled_q=True).prescaled_q == 1


def test_unsupported_configs_are_rejected():
    base = kc.get_kernels_to_build()[0]
    from dataclasses import replace

    assert not _capi.supported(replace(base, d_head=96))
    assert not _capi.supported(replace(base, d_head=64))  # no 16-rows-per-wave kernel at d_head 64
    assert not _capi.supported(replace(base, B_c=48))
    assert not _capi.supported(replace(base, n_warps=3))
    assert not _capi.supported(replace(base, Q_mma_load_K_tiles=2, K_mma_load_K_tiles=0))
    assert not _capi.supported(replace(base, K_mma_load_K_tiles=3))
    with pytest.raises(_capi.FaError) as e:
        _capi.lds_bytes(replace(base, d_head=96))
    assert "d_head" in str(e.value)


def _args(cfg, seq=256, **over):
    a = dict(q=4096, k=8192, v=12288, o=16384, batch=1, seq_len=seq, n_heads=1, d_head=128,
             batch_stride=seq * 128, seq_stride=128, head_stride=128, cfg=_capi.make_config(cfg))
    a.update(over)
    return _capi.FaFwdArgs(**a)


def test_launch_argument_validation_needs_no_gpu():
    """Validation runs before any HIP call: status codes + the reference's messages."""
    lib = _capi.load()
    cfg = kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 128, 64, 4, True, True, True, 2, 2, 0, False, False)

    def status(args):
        return lib.fa_fwd_launch(ctypes.byref(args), None), _capi.last_error()

    rc, msg = status(_args(cfg, q=0))
    assert rc == -1 and "null" in msg
    rc, msg = status(_args(cfg, seq=192, batch_stride=192 * 128))
    assert rc == -4 and msg == "Only multiples of B_r are supported for seq_len Q currently"
    rc, msg = status(_args(kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 64, 64, 4, True, True, True, 0, 0, 0, False, False), seq=96))
    assert rc == -4 and "B_r" in msg
    rc, msg = status(_args(cfg, d_head=64))
    assert rc == -4 and "d_head" in msg
    rc, msg = status(_args(cfg, seq_stride=132))
    assert rc == -5
    rc, msg = status(_args(cfg, q=4100))
    assert rc == -5
    # strides: sign, zero with more than one entry, and the 32-bit range of row * seq_stride * 2 bytes
    for over in (dict(seq_stride=-128), dict(seq_stride=0), dict(batch_stride=-8), dict(head_stride=-8),
                 dict(batch=2, batch_stride=0), dict(n_heads=2, head_stride=0)):
        rc, msg = status(_args(cfg, **over))
        assert rc == -4 and "strides must be positive" in msg, over
    limit = 0xFFFFFFFF // (2 * 128)  # B_r = 128 rows
    rc, msg = status(_args(cfg, seq_stride=(limit // 8 + 1) * 8))
    assert rc == -4 and "too large" in msg
    if not torch.cuda.is_available():  # (with a GPU this would launch on the fake pointers)
        rc, msg = status(_args(cfg, seq_stride=(limit // 8) * 8, batch_stride=1 << 40))
        assert rc == -7  # valid arguments: the first HIP call (no device here) is what fails
    bad = _args(cfg)
    bad.cfg.dtype = 6
    rc, msg = status(bad)
    assert rc == -2 and msg == "Only fp16 and bf16 are supported"
    bad = _args(cfg)
    bad.cfg.B_c = 48
    rc, msg = status(bad)
    assert rc == -3 and "not found" in msg
    if not torch.cuda.is_available():
        # a valid call on a box without a gfx950 device fails loudly, never falls back
        rc, msg = status(_args(cfg))
        assert rc in (-7, -6) and msg


def test_python_shim_checks_follow_the_reference():
    import flash_attention

    cfg = kc.get_kernels_to_build()[0]
    q = torch.zeros((1, 256, 2, 128), dtype=cfg.dtype.to_torch_dtype())
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        flash_attention.forward(cfg, q, q, q)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _capi.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "flash_attention_from_scratch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "fa_oracle" not in text, f


def test_short_form_round_trips_and_env_selector(monkeypatch):
    for cfg in kc.get_all_supported_configs():
        assert kc.parse_kernel_name_into_config(cfg.short_form()) == cfg
        assert kc.parse_kernel_name_into_config(f"| {cfg.short_form()} | 1.0 |") == cfg
    monkeypatch.setenv("KERNELS", "128,64")
    sel = kc.get_kernel_configs()
    assert sel and all((c.B_r, c.B_c) == (128, 64) for c in sel)
    monkeypatch.setenv("KERNELS", "tune")
    assert len(kc.get_kernel_configs()) == 80
    monkeypatch.setenv("KERNELS", "prog")
    assert len(kc.get_kernel_configs()) == 7
    monkeypatch.setenv("KERNELS", "bogus")
    with pytest.raises(ValueError):
        kc.get_kernel_configs()
    assert kc.DType.from_string("bf16") is kc.DType.BF16 and kc.DType.from_string("5") is kc.DType.FP16
    assert kc.transform_kernel_name("not a kernel") == "not a kernel"


def test_config_name_round_trips_property():
    """Any field combination survives short_form -> parse and to_cpp_struct -> parse (hypothesis)."""
    from hypothesis import given, settings, strategies as st

    cfgs = st.builds(
        kc.FlashForwardKernelConfig,
        dtype=st.sampled_from(list(kc.DType)), d_head=st.sampled_from([64, 128]),
        B_r=st.sampled_from([64, 128, 256]), B_c=st.sampled_from([32, 64, 128]), n_warps=st.sampled_from([4, 8]),
        async_copy=st.booleans(), eager_load_blocks=st.booleans(), swizzled=st.booleans(),
        Q_mma_load_K_tiles=st.sampled_from([0, 2, 4]), K_mma_load_K_tiles=st.sampled_from([0, 2, 4]),
        V_mma_load_K_tiles=st.sampled_from([0, 2, 4]), mma_double_buffer_loads=st.booleans(),
        optimized_softmax=st.booleans(),
    )

    @settings(max_examples=200, deadline=None)
    @given(cfgs)
    def check(cfg):
        assert kc.parse_kernel_name_into_config(cfg.short_form()) == cfg
        demangled = "void flash_forward_kernel<" + cfg.to_cpp_struct().replace("torch::kFloat16", "5").replace(
            "torch::kBFloat16", "15") + ">(args)"
        assert kc.parse_kernel_name_into_config(demangled) == cfg
        assert _capi.make_config(cfg).B_r == cfg.B_r and tuple(cfg.to_c_abi_tuple())[0] == int(cfg.dtype)
        assert cfg.total_flop(2, 3, 256) == kc.calc_total_flop(2, 3, 256, cfg.B_r, cfg.B_c, cfg.d_head)

    check()

    native = st.builds(
        kc.NativeKernelConfig,
        dtype=st.sampled_from(list(kc.DType)), d_head=st.sampled_from([64, 128]),
        B_r=st.sampled_from([64, 128, 256]), B_c=st.sampled_from([32, 64, 128]), n_warps=st.sampled_from([4, 8]),
        async_copy=st.booleans(), eager_load_blocks=st.booleans(), swizzled=st.booleans(),
        Q_mma_load_K_tiles=st.sampled_from([0, 2]), K_mma_load_K_tiles=st.sampled_from([0, 2]),
        V_mma_load_K_tiles=st.sampled_from([0, 2]), mma_double_buffer_loads=st.booleans(),
        optimized_softmax=st.booleans(), speculative_softmax=st.booleans(), prescaled_q=st.booleans(),
    )

    @settings(max_examples=200, deadline=None)
    @given(native)
    def check_native(cfg):
        """The native extensions ride behind the reference's words in the short form and never enter the 13-field key."""
        back = kc.parse_kernel_name_into_config(cfg.short_form())
        assert kc.config_sort_key(back) == kc.config_sort_key(cfg)
        assert isinstance(back, kc.NativeKernelConfig) == (cfg.speculative_softmax or cfg.prescaled_q)
        assert cfg.to_c_abi_tuple() == cfg.base().to_c_abi_tuple() and len(cfg.to_c_abi_tuple()) == 13
        assert cfg.base().short_form() == cfg.short_form().replace("+spec_softmax", "").replace("+prescaled_q", "")
        assert kc.parse_kernel_name_into_config("void k<" + cfg.to_cpp_struct().replace("torch::kFloat16", "5").replace(
            "torch::kBFloat16", "15") + ">(a)") == cfg.base()
        assert kc.wants_speculative(cfg) == cfg.speculative_softmax

    check_native()


def test_best_config_selection_rules():
    """The default: the persistent 64-rows-per-wave kernel whenever seq_len is a multiple of its
    256-row Q block; for forward_ex also when rounding seq_len up to one costs at most an eighth more
    rows (its ragged form); otherwise the 32-rows-per-wave kernel, whose masked form takes any length."""
    persistent = dict(B_r=256, B_c=64, n_warps=4, mma_double_buffer_loads=True)
    for dtype in (kc.DType.BF16, kc.DType.FP16):
        for S in (256, 512, 4096, 16384, 32768):
            for masked in (False, True):
                cfg = kc.best_config(dtype, S, masked=masked)
                assert kc.uses_lazy_rescale(cfg) and cfg.dtype == dtype
                assert all(getattr(cfg, k) == v for k, v in persistent.items())
                assert _capi.supported(cfg) and _capi.masked_supported(cfg)
        for S in (1000, 2500, 4000, 5000, 8191):
            assert kc.uses_lazy_rescale(kc.best_config(dtype, S, masked=True))
            assert not kc.uses_lazy_rescale(kc.best_config(dtype, S))
        for S in (1, 63, 100, 320, 769, 1100):
            cfg = kc.best_config(dtype, S, masked=True)
            assert not kc.uses_lazy_rescale(cfg) and (cfg.B_r, cfg.B_c, cfg.n_warps) == (128, 64, 4)
            assert _capi.masked_supported(cfg)


def test_ragged_window_rule_covers_every_key_exactly_once():
    """The persistent kernel's ragged form (csrc/fa_fwd_kernel64.hpp, RAG): the host rounds the Q blocks up
    and runs 4 tiles of 64 keys per Q block; tile t is fetched from row r0 = min(64 t, S - 64) and the first
    delta = 64 t - r0 keys of that window are masked.  Restated here: for every S >= 64 the unmasked keys of
    all tiles are exactly 0 .. S-1, each once, and no window leaves the tensor."""
    for S in list(range(64, 700)) + [1000, 4000, 8191, 16385]:
        n_tiles = 4 * ((S + 255) // 256)
        seen = [0] * S
        for t in range(n_tiles):
            r0 = min(64 * t, S - 64)
            delta = 64 * t - r0
            assert 0 <= r0 and r0 + 64 <= S
            for key_in_window in range(max(delta, 0), 64):
                seen[r0 + key_in_window] += 1
        assert seen == [1] * S, S



def test_per_device_init_bookkeeping_without_a_gpu():
    """fa_init keeps its state per device ordinal (a host may drive several GPUs from one process) and a
    box with no device gets FA_ERR_DEVICE on EVERY call -- nothing is cached for a device that could not
    even be named.  (The multi-GPU side is a -m gpu test that needs two devices.)"""
    lib = _capi.load()
    if torch.cuda.is_available():
        pytest.skip("CPU-tier bookkeeping test")
    for _ in range(2):
        assert lib.fa_init() == -7  # FA_ERR_DEVICE
        assert b"no HIP device" in lib.fa_last_error()
    for dev in (0, 1, 63):
        inited, status, num_cus = _capi.device_state(dev)
        assert (inited, status, num_cus) == (False, 0, 0)   # (nothing is published before a device's init has finished)
    with pytest.raises(RuntimeError):
        _capi.device_state(64)
    with pytest.raises(RuntimeError):
        _capi.device_state(-1)


def test_the_two_distributions_build_and_import_outside_the_repo(tmp_path):
    """Packaging (the reference's setup.py:45-75 + py/setup.py:6-9): `flash_attention` (+ the
    `flash_attention_kernels` module and libfa_hip.so as package data) and `flash_helpers` build as
    wheels, and the reference's import names resolve from the unpacked wheels alone -- with the
    repository NOT on sys.path -- down to the loaded C-ABI library."""
    import shutil
    import sys
    import zipfile

    src = tmp_path / "src"
    shutil.copytree(ROOT, src, ignore=shutil.ignore_patterns(".git", "gpurun_out", "__pycache__", "build", "*.o",
                                                             ".pytest_cache", ".hypothesis", "profiles", "golden"))
    out = tmp_path / "whl"
    env = dict(os.environ, FA_SKIP_NATIVE_BUILD="1")  # the library is already built in-tree (build())
    for where in (src, src / "py"):
        res = subprocess.run([sys.executable, "-m", "pip", "wheel", ".", "--no-build-isolation", "--no-deps", "-q",
                              "-w", str(out)], cwd=where, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
    site = tmp_path / "site"
    names = {}
    for whl in sorted(out.glob("*.whl")):
        with zipfile.ZipFile(whl) as z:
            names[whl.name.split("-")[0]] = set(z.namelist())
            z.extractall(site)
    assert set(names) == {"flash_attention", "flash_helpers"}
    fa = names["flash_attention"]
    assert {"flash_attention/__init__.py", "flash_attention_kernels.py",
            "flash_attention_from_scratch_amd/lib/libfa_hip.so", "flash_attention_from_scratch_amd/_capi.py"} <= fa
    assert {"flash_helpers/kernel_configs.py", "flash_helpers/test/utils.py", "flash_helpers/test/test.py"} <= names["flash_helpers"]
    probe = ("import sys; assert not any(p.rstrip('/') == %r for p in sys.path)\n"
             "import flash_attention, flash_attention_kernels\n"
             "from flash_helpers.kernel_configs import get_kernels_to_build, best_config\n"
             "from flash_helpers.test.utils import BATCH_SIZE_FOR_SEQ_LEN\n"
             "import flash_helpers.test.test as t\n"
             "from flash_attention_from_scratch_amd import _capi\n"
             "assert _capi.LIB_PATH.startswith(%r), _capi.LIB_PATH\n"
             "assert _capi.load().fa_num_kernels() > 40 and all(_capi.supported(c) for c in get_kernels_to_build())\n"
             "assert hasattr(flash_attention, 'forward') and hasattr(flash_attention, 'forward_timed')\n"
             "print(len(get_kernels_to_build()), flash_attention.__file__)\n") % (ROOT, str(site))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = str(site)
    res = subprocess.run([sys.executable, "-c", probe], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    assert res.stdout.split()[0] == "80" and str(site) in res.stdout


def test_py_distribution_ships_its_own_alias_package():
    """py/ (the `flash_helpers` distribution, reference py/setup.py:6-9) carries the alias package inside its own project
    root -- an sdist or an isolated build sees nothing above it.  The repository root's flash_helpers/ is only a pointer
    to it (one `__init__.py` that sets `__path__`): ONE copy of the package in the tree (round 3 kept two in step by a test)."""
    import flash_helpers
    import flash_helpers.kernel_configs as kc_alias
    import flash_helpers.test.utils as ut_alias

    inner = os.path.join(ROOT, "py", "flash_helpers")
    assert [os.path.realpath(p) for p in flash_helpers.__path__] == [os.path.realpath(inner)]
    assert os.path.realpath(kc_alias.__file__) == os.path.realpath(os.path.join(inner, "kernel_configs.py"))
    assert os.path.realpath(ut_alias.__file__) == os.path.realpath(os.path.join(inner, "test", "utils.py"))
    assert [n for n in sorted(os.listdir(os.path.join(ROOT, "flash_helpers"))) if n != "__pycache__"] == ["__init__.py"]
    assert kc_alias.best_config is kc.best_config
    setup_py = open(os.path.join(ROOT, "py", "setup.py")).read()
    assert "os.pardir" not in setup_py and "package_dir" not in setup_py


def build_c_client(out_dir):
    """examples/c_client.c with gcc -std=c99 (no C++, no torch): the boundary is a C ABI."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(out_dir), "c_client")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"),
           "-I/opt/rocm/include", os.path.join(root, "examples", "c_client.c"),
           "-L" + os.path.dirname(_capi.LIB_PATH), "-lfa_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-o", exe]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    return exe


def test_plain_c_client_compiles_and_links(tmp_path):
    """include/fa_hip.h is C: a C99 translation unit that fills fa_fwd_args / fa_fwd_opts, calls fa_fwd_query and
    fa_fwd_launch_ex and reads fa_fwd_stats builds with gcc and links against libfa_hip.so (the GPU tier runs it)."""
    _capi.load()
    exe = build_c_client(tmp_path)
    assert os.path.getsize(exe) > 0


def build_torch_binding(out_dir):
    """examples/torch_binding.cpp -- the pybind function a maintainer of the reference would compile instead of
    src/flash_attention.cu -- JIT-built by torch.utils.cpp_extension with g++ (no device code)."""
    from torch.utils.cpp_extension import load

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(_capi.LIB_PATH)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    return load("fa_ref_binding", [os.path.join(root, "examples", "torch_binding.cpp")],
                extra_include_paths=[os.path.join(root, "include"), "/opt/rocm/include"],
                extra_cflags=["-D__HIP_PLATFORM_AMD__", "-O2"],
                extra_ldflags=["-L" + lib_dir, "-lfa_hip", "-Wl,-rpath," + lib_dir, "-L" + torch_lib, "-lc10_hip"],
                build_directory=str(out_dir), verbose=False)


def test_reference_side_binding_builds(tmp_path):
    """The binding of INTEGRATION.md 2 is a file that compiles: a torch extension with the reference module's signature over
    the C ABI.  Without a GPU its tensor checks still answer (the reference's CHECK_INPUT); the GPU tier runs it."""
    _capi.load()
    ext = build_torch_binding(tmp_path)
    assert ext.version() == _capi.load().fa_version().decode()
    cfg = kc.get_kernels_to_build()[0]
    x = torch.zeros(1, 128, 1, 128, dtype=cfg.dtype.to_torch_dtype())
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ext.forward(cfg, x, x, x, None)


def test_jitter_library_is_built_and_exports_the_same_abi():
    """lib/libfa_hip_jitter.so (csrc/Makefile `jitter`, built by __graft_entry__.build()): the timing-perturbed twin the GPU
    tier runs through FA_HIP_LIB.  Same entry points, same registry; its 64-rows-per-wave slices carry the sleeps."""
    path = os.path.join(ROOT, "flash_attention_from_scratch_amd", "lib", "libfa_hip_jitter.so")
    assert os.path.exists(path), "run `make -C flash_attention_from_scratch_amd/csrc jitter`"
    lib = ctypes.CDLL(path)
    for sym in _capi.EXPORTED_SYMBOLS:
        assert hasattr(lib, sym), sym
    lib.fa_num_kernels.restype = ctypes.c_int
    assert lib.fa_num_kernels() == _capi.load().fa_num_kernels()
    src = open(os.path.join(ROOT, "flash_attention_from_scratch_amd", "csrc", "fa_fwd_kernel64.hpp")).read()
    assert src.count("FA_JIT(") >= 8 and "#ifdef FA_JITTER" in src
    # the product path never names it: FA_HIP_LIB is the only way in
    for rel in ("flash_attention_from_scratch_amd/_capi.py", "flash_attention_from_scratch_amd/flash_attention_kernels.py", "bench.py"):
        code = [ln for ln in open(os.path.join(ROOT, rel)).read().splitlines() if not ln.lstrip().startswith("#")]
        assert not any("libfa_hip_jitter" in ln for ln in code), rel


def test_adaptive_mode_abi():
    """fa_speculative_mode / fa_adaptive_info (include/fa_hip.h): struct layout of the ctypes mirror, the version numbers."""
    assert ctypes.sizeof(_capi.FaAdaptiveInfo) == 8 * 4
    lib = _capi.load()
    assert lib.fa_abi_version() == _capi.FA_ABI_VERSION == 6
    header = open(os.path.join(ROOT, "include", "fa_hip.h")).read()
    assert "#define FA_ABI_VERSION 6" in header and "FA_SPECULATIVE_ADAPTIVE = 2" in header
    # the boundary row (SURVEY 8b threading / streams; the reference's launcher is stateless, src/flash_attention.cu:42,118,
    # 126-131): the header says which entry points keep state, and no longer claims that none does
    assert "no global mutable state" not in header and "STATELESS" in header and "per device variant" in header
    # an out-of-range `speculative` is refused by supported / query / launch alike, without a device
    cfg_c = _capi.make_config(kc.best_config(kc.DType.BF16))
    bad = _capi.make_opts(speculative=True)
    bad.speculative = 3
    assert lib.fa_fwd_ex_supported(ctypes.byref(cfg_c), ctypes.byref(bad)) == 0
    assert lib.fa_fwd_query(ctypes.byref(cfg_c), ctypes.byref(bad), ctypes.byref(_capi.FaKernelInfo())) == -4
    assert "must be 0, 1 (always) or 2 (adaptive)" in _capi.last_error()
    ok = _capi.make_opts(speculative="adaptive")
    assert lib.fa_fwd_ex_supported(ctypes.byref(cfg_c), ctypes.byref(ok)) == 1
    # the per-variant record answers without a device too (nothing initialised: zeros, hold 32)
    st = _capi.adaptive_state(0, kc.best_config(kc.DType.FP16))
    assert st["launches"] == 0 and st["mode"] == 0 and st["hold"] == 32
    assert _capi.make_opts(speculative="adaptive").speculative == 2 and _capi.make_opts(speculative=True).speculative == 1
    assert _capi.make_opts(speculative=False).speculative == 0
    info = _capi.FaKernelInfo()
    small = ctypes.sizeof(_capi.FaKernelInfo) - 8 - 16     # a client built against the 0.2 header (no softmax_mode / prescaled_q, no ring_* fields)
    buf = (ctypes.c_char * ctypes.sizeof(_capi.FaKernelInfo))()
    ctypes.memset(buf, 0x5A, ctypes.sizeof(buf))
    assert lib.fa_get_kernel_sized(0, ctypes.cast(buf, ctypes.POINTER(_capi.FaKernelInfo)), small) == 0
    assert bytes(buf)[small:] == b"\x5a" * 24         # nothing written beyond the caller's size
    assert lib.fa_get_kernel(0, ctypes.byref(info)) == 0 and bytes(buf)[:small] == bytes(info)[:small]


def test_adaptive_policy_scripted():
    """The adaptive speculative mode's policy (csrc/fa_capi.hip AdaptivePolicy; include/fa_hip.h fa_speculative_mode), run
    without a device through fa_adaptive_simulate: NORMAL -> a report -> DEMOTED for `hold` launches -> one PROBE -> demoted
    while the probe is pending -> a failed probe doubles the hold (capped at 4096), a clean one returns to NORMAL / hold 32."""
    SPEC, DEM, PROBE = 0, 1, 2
    # benign: nothing ever reported
    run, st = _capi.adaptive_simulate([0] * 50, [-1] * 50)
    assert run == [SPEC] * 50 and st["mode"] == 0 and st["demoted"] == 0 and st["launches"] == 50
    # launch 3 (sequence number 3) fails; its report lands while launches 4 .. 6 are enqueued; seen by launch 7
    rep = [0, 0, 0, 0, 0, 0] + [3] * 32 + [3]
    prb = [-1] * 38 + [-1]
    run, st = _capi.adaptive_simulate(rep, prb)
    assert run[:6] == [SPEC] * 6 and run[6:38] == [DEM] * 32 and run[38] == PROBE
    assert st["mode"] == 2 and st["reports"] == 1 and st["hold"] == 32 and st["demoted"] == 32
    # the probe (sequence number 39) is pending for three launches, then complete WITH a report: hold doubles
    rep2 = rep + [3, 3, 3, 39]
    prb2 = prb + [0, 0, 0, 1]
    run, st = _capi.adaptive_simulate(rep2, prb2)
    assert run[39:] == [DEM] * 4 and st["mode"] == 1 and st["hold"] == 64 and st["remaining"] == 63 and st["reports"] == 2
    # ... 63 more demoted, the next probe (sequence 107) runs clean: NORMAL, hold back to 32
    rep3 = rep2 + [39] * 63 + [39, 39, 39]
    prb3 = prb2 + [-1] * 63 + [-1, 0, 1]
    run, st = _capi.adaptive_simulate(rep3, prb3)
    assert run[43:106] == [DEM] * 63 and run[106] == PROBE and run[107] == DEM and run[108] == SPEC
    assert st["mode"] == 0 and st["hold"] == 32 and st["reports"] == 2
    # steady failure: every probe fails -> holds 32, 64, ... 4096, 4096: ONE speculative launch per hold
    rep, prb, want = [0], [-1], [SPEC]          # launch 1: speculative, fails (its report word: 1)
    word, start, hold, holds = 1, 2, 32, []
    for _ in range(9):
        holds.append(hold)
        # the launch at `start` sees the report (and, behind a probe, that the probe's event has completed)
        rep += [word] * hold
        prb += [1 if len(holds) > 1 else -1] + [-1] * (hold - 1)
        want += [DEM] * hold
        rep.append(word); prb.append(-1); want.append(PROBE)    # the hold has run out: the probe, which fails too
        word = start + hold
        start = word + 1
        hold = min(2 * hold, 4096)
    run, st = _capi.adaptive_simulate(rep, prb)
    assert holds == [32, 64, 128, 256, 512, 1024, 2048, 4096, 4096]
    assert run == want and run.count(PROBE) == 9 and run.count(SPEC) == 1
    assert st["mode"] == 2 and st["hold"] == 4096 and st["reports"] == 9 and st["demoted"] == sum(holds)
