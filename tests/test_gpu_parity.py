"""GPU tier (-m gpu): the HIP path, called through the C ABI (flash_attention ->
flash_attention_kernels -> libfa_hip.so), against
  * the committed golden fixtures (reference's eager oracle outputs),
  * the C oracle (oracle/fa_oracle.c, the device algorithm restated on CPU),
  * the reference's own tolerance rule on its own test fixture (test.py:18-61),
  * torch SDPA,
and, at BASELINE.json's full sizes, through size-independent properties.

Tolerances (floating point; stated here as the task requires):
  vs fp32 eager cast to 16 bit : <= 2 ulp of the 16-bit type at |o| < 1  (bf16 2^-7, fp16 2^-10)
  reference rule               : max|out - eager_b16| <= 2 * max|eager_b16 - eager_f32|
  vs C oracle (same algorithm) : <= 2 ulp (v_exp_f32 vs libm exp2f, MFMA vs serial fp32 sums)
"""
import os
from dataclasses import replace

import pytest
import torch

pytestmark = pytest.mark.gpu

import flash_attention  # noqa: E402
import flash_attention_kernels  # noqa: E402
from flash_attention_from_scratch_amd import _capi  # noqa: E402
from flash_helpers import kernel_configs as kc  # noqa: E402
from flash_helpers.test import utils as ut  # noqa: E402
from oracle import fa_oracle as fo  # noqa: E402
from tests.conftest import load_eager_golden, load_seam_golden  # noqa: E402

DEV = "cuda:0"
TOL = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}
TAG = {kc.DType.BF16: "bf16", kc.DType.FP16: "fp16"}


def _unique_variants(cfgs):
    """One config per device variant (the operand-fetch hints share device code)."""
    seen, out = set(), []
    for c in cfgs:
        key = (c.dtype, c.d_head, c.B_r, c.B_c, c.n_warps, c.async_copy, c.eager_load_blocks, c.swizzled,
               c.optimized_softmax, kc.wants_speculative(c), getattr(c, "prescaled_q", False), c.mma_double_buffer_loads and ((c.B_r // c.n_warps == 32 and c.B_c <= 64) or (c.B_r // c.n_warps == 64 and c.B_c == 64)))
        if key not in seen:
            seen.add(key)
            out.append(c)
    return out


ALL = [c for c in kc.get_all_supported_configs() if c.d_head == 128]   # the reference's scope
VARIANTS = _unique_variants(ALL)
D64 = _unique_variants(kc.get_d64_kernel_configs())                      # scope widener


def test_library_loaded_and_device_is_gfx950():
    assert os.path.exists(_capi.LIB_PATH)
    assert _capi.check(_capi.load().fa_init()) == 0
    assert ut.is_mi355x()
    n_ring = 0
    for info in _capi.kernels():
        assert info.scratch_bytes == 0, "register spill in a device variant"
        assert 0 < info.num_regs <= 512
        if info.ring_form:   # the hand-placed ring form of (B_r 128, B_c 64, 4 warps) + buffer (round 5)
            n_ring += 1
            assert info.ring_scratch_bytes == 0 and 0 < info.ring_num_regs <= 512
            assert info.ring_softmax_mode == (3 if info.softmax_mode == 3 else 2)   # speculative variant: speculative ring
            assert (info.cfg.B_r, info.cfg.B_c, info.cfg.n_warps, info.masked, info.rows_per_wave) == (128, 64, 4, 0, 32)
    assert n_ring == 4   # plain + speculative, per dtype


def test_c_abi_per_device_state():
    """libfa_hip.so keeps its one-time setup per device ordinal (arch check, CU count, the > 48 KB LDS
    opt-in of every kernel function): the current device is initialised by the first call made on it."""
    inited, status, num_cus = _capi.device_state(0)
    assert inited and status == 0 and num_cus == torch.cuda.get_device_properties(0).multi_processor_count
    if torch.cuda.device_count() < 2:
        pytest.skip("the second half needs two GPUs: a launch on device 1 after device 0 was initialised")
    cfg = kc.best_config(kc.DType.BF16, 512)
    outs = []
    for dev in (0, 1):
        gen = torch.Generator(device=f"cuda:{dev}").manual_seed(3)
        q, k, v = (torch.randn((2, 512, 4, 128), dtype=torch.bfloat16, device=f"cuda:{dev}", generator=gen)
                   for _ in range(3))
        outs.append(flash_attention.forward(cfg, q, k, v).cpu())   # (the shim launches under q's device)
        assert _capi.device_state(dev)[:2] == (True, 0)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cfg", VARIANTS, ids=str)
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_golden_fixtures(cfg, case):
    g = load_eager_golden(TAG[cfg.dtype], case)
    S = g["q"].shape[1]
    if S % cfg.B_r or S % cfg.B_c:
        pytest.skip("fixture seq_len not a multiple of the tile")
    q, k, v = (g[n].to(DEV) for n in ("q", "k", "v"))
    out = flash_attention.forward(cfg, q, k, v).cpu()
    assert torch.isfinite(out.float()).all()
    err = (out.float() - g["o_f32"].float()).abs().max().item()
    assert err <= TOL[g["dtype"]], err
    lhs, rhs = fo.tolerance_rule(out, g["o_b16"], g["o_f32"])
    assert lhs <= rhs, (lhs, rhs)
    ref = fo.blockwise_for_config(cfg, g["q"], g["k"], g["v"])
    assert (out.float() - ref.float()).abs().max().item() <= TOL[g["dtype"]]


@pytest.mark.parametrize("tag", ["bf16", "fp16"])
def test_seam_golden_with_spikes_and_the_redo_counter(tag):
    """The reference-generated fixture that reaches what the small ones cannot (tests/golden/seam_*.npz, rebuilt from its
    recipe): 288 items of 256 rows -- more than the chip has CUs, so the persistent walk crosses item seams -- with logit
    spikes in a FIRST item (item 5) and a SECOND item (item 260) of a workgroup and a mild one (20 binades: bf16 stays
    in the first pass, fp16 does not).  Every form of the persistent kernel and the reference's winning shape against
    the stored rows of the reference's eager outputs; and fa_fwd_stats counts exactly the items that ran twice."""
    g = load_seam_golden(tag)
    dtype, name = g["dtype"], {"bf16": kc.DType.BF16, "fp16": kc.DType.FP16}[tag]
    q, k, v = (g[n].to(DEV) for n in ("q", "k", "v"))
    b, r, h = g["b"], g["r"], g["h"]
    tol = TOL[dtype] * (1 + g["o_f32"].float().abs())
    sane = torch.isfinite(g["o_b16"].float()).all(dim=-1)   # (fp16: the reference's 16-bit eager overflows on the 30-sigma rows)
    n_items = {256: 3 * 24 * 4, 128: 3 * 24 * 8}

    def predicted_redone(B_r, ring=True, fwd=True):
        """Items whose first pass must fail, from the fp32 logits: a row fails when l = sum_k 2^((s_k - m_first) c) reaches
        the limit (bf16: spec_limit 2^64 on the compiler-scheduled 32-rows-per-wave kernel, spec_limit64 2^120 on the
        persistent one -- both its forms, ring = True -- whose guard looks at O itself; fp16 2^15), m_first = the row's max over the
        64 keys visited first (the FIRST 64 for the persistent kernel's speculative pass, fwd; the LAST 64 for the compiler-
        scheduled bodies, which keep the reference's order); an item fails
        when one of its rows does.  (One spiked key per row and N(0, 1) elsewhere: no reference has moved before the spike
        arrives, so the guarded kernel's criterion is the same comparison.)  A 30-sigma K row gives EVERY query of its head
        logits of ~+-43 binades, so (nearly) all Q blocks of the two spiked heads fail; the mild spike fails in fp16 only,
        and only in its own Q block."""
        c = 1.4426950408889634 / 128 ** 0.5
        limit = (2.0 ** 120 if ring else 2.0 ** 64) if tag == "bf16" else 2.0 ** 15
        n = 0
        for bb in range(q.shape[0]):
            sc = torch.einsum("qhd,khd->hqk", q[bb].float(), k[bb].float())
            m_first = (sc[:, :, :64] if fwd else sc[:, :, -64:]).amax(dim=-1, keepdim=True)
            l = torch.exp2((sc - m_first).double() * c).sum(dim=-1)          # (heads, rows)
            bad = ~(l < limit)
            n += int(bad.view(bad.shape[0], -1, B_r).any(dim=-1).sum())
        return n

    # (seq_len % 256 == 0 here: the (128, 64, 4)+buffer configurations run their ring form, the persistent kernel with
    # one Q tile per wave -- its limit; since round 6 EIGHT waves: two of the shape's 128-row Q blocks make one 256-row item, and
    # an item that fails is counted as the two Q blocks that ran twice)
    expect = {256: predicted_redone(256), 128: 2 * predicted_redone(256)}
    # (the spike at key 3 sits in the tile the forward walk visits FIRST: it is the reference there and fails nothing; rounds
    # 2-5, walking last-to-first, counted 7 / 8 (bf16) and 9 / 17 (fp16) -- predicted_redone(.., fwd=False) still does)
    assert expect[256] == (4 if tag == "bf16" else 5) and expect[128] == (8 if tag == "bf16" else 10), expect
    assert predicted_redone(256, fwd=False) == (7 if tag == "bf16" else 9)
    assert predicted_redone(128, ring=False, fwd=False) == (16 if tag == "bf16" else 17)   # (the compiler-scheduled body's limit and order)
    if tag == "fp16":
        # the mild spike: 20 binades at once fail fp16 -- unless the guard has moved that row's reference up by then
        expect[256], expect[128] = (4, 5), (8, 10)
    for cfg, redone in ((_persistent_cfg(name, True), expect[256]), (_persistent_cfg(name, False), 0),
                        (_native(name, 128, 64, 4, True, True), expect[128]), (_native(name, 128, 64, 4, True, False), 0),
                        (kc.FlashForwardKernelConfig(name, 128, 128, 64, 4, True, True, True, 2, 2, 0, True, True), 0)):
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        out, _ = flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
        got = out.cpu()[b, r, h]
        assert torch.isfinite(out.float()).all(), str(cfg)
        assert ((got.float() - g["o_f32"].float()).abs() <= tol).all(), str(cfg)
        lhs, rhs = fo.tolerance_rule(got[sane], g["o_b16"][sane], g["o_f32"][sane])
        assert lhs <= rhs, (str(cfg), lhs, rhs)
        first = stats.tolist()
        assert first[0] == n_items[cfg.B_r] and first[1] in (redone if isinstance(redone, tuple) else (redone,)), (str(cfg), first)
        # the counters ADD: a second launch doubles them; and without the pointer nothing changes
        flash_attention_kernels.forward(cfg, q, k, v, out, stats=stats)
        assert stats.tolist() == [2 * first[0], 2 * first[1]]
        assert torch.equal(flash_attention.forward(cfg, q, k, v), out)


def test_optimized_softmax_keeps_the_reference_meaning(monkeypatch):
    """One flag, one meaning: a plain 13-field config with optimized_softmax (what a reference user passes, e.g. the
    A100 winner of kernel_sass/16_A100.asm:5 plus the RTX 3090 winner's flag) computes the reference's arithmetic -- the
    first-block skip changes nothing, so the result is BIT-identical to the same config without the flag -- and never
    the speculative softmax: a spike that would send that one through its second pass is not counted as redone.  Under
    FA_ALLOW_SPECULATIVE=1 the round-2 mapping is back (the counter sees it)."""
    from flash_helpers.kernel_configs import parse_kernel_name_into_config as parse
    a100 = parse("(FP16, 128, 128, 64, 4): async+eager+swizzled+load_2_2_0_tiles+buffer+opt_softmax")
    for cfg in (a100, replace(a100, dtype=kc.DType.BF16), replace(a100, B_r=64, B_c=32, Q_mma_load_K_tiles=2, K_mma_load_K_tiles=2)):
        monkeypatch.delenv("FA_ALLOW_SPECULATIVE", raising=False)
        dtype = cfg.dtype.to_torch_dtype()
        qc = ut.QKVConfig(n_heads=3, d_head=128, batch_size=2, seq_len=1024, dtype=dtype, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=23)
        u = _sign_vector(5).to(dtype)
        # (in the tile the speculative form of the config visits LAST: the ring form walks first-to-last)
        k[1, 1024 - 1 - 3 if kc.has_ring_form(cfg) else 3, 2] = 30.0 * u
        q[1, 600:604, 2] = 30.0 * u
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        out, _ = flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
        assert kc.softmax_mode(cfg) == "eager" and stats[1].item() == 0
        assert torch.equal(out, flash_attention.forward(replace(cfg, optimized_softmax=False), q, k, v))
        monkeypatch.setenv("FA_ALLOW_SPECULATIVE", "1")
        stats.zero_()
        out2, _ = flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
        # (a 30-sigma K row sends EVERY query of its head ~+-43 binades off: all S / B_r workgroups of that head start over)
        # (... of the compiler-scheduled body and of fp16; the bf16 ring form of (128, 64, 4)+buffer -- seq_len % 256 == 0 --
        # has the persistent kernel's 2^120 limit, which +-43 binades stay under: only the spiked queries' items are redone)
        assert kc.softmax_mode(cfg) == "speculative"
        if kc.has_ring_form(cfg) and cfg.dtype == kc.DType.BF16:
            assert 1 <= stats[1].item() < 1024 // cfg.B_r, stats.tolist()
        else:
            assert stats[1].item() == 1024 // cfg.B_r, stats.tolist()
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        for o in (out, out2):
            assert ((o.float() - ref).abs() <= TOL[dtype] * (1 + ref.abs())).all()


@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def reference_fixture(request):
    """The reference suite's fixture: (16, 2048, 16, 128) N(0,1) (test.py:20-49), seeded."""
    dtype = request.param
    cfg = ut.QKVConfig(n_heads=ut.BENCHMARK_N_HEADS, d_head=128,
                       batch_size=ut.BATCH_SIZE_FOR_SEQ_LEN[2048], seq_len=2048,
                       dtype=dtype, device=torch.device(DEV))
    q, k, v = ut.generate_qkv(cfg, seed=1234)
    ref_b16 = ut.py_flash_attention(q, k, v, upcast=False)
    ref_f32 = ut.py_flash_attention(q, k, v, upcast=True)
    sdpa = ut.sdpa_attention(q, k, v)
    return dtype, q, k, v, ref_b16, ref_f32, sdpa


def test_reference_suite_every_config(reference_fixture):
    """test.py:51-61 for every config of get_kernels_to_build() + the native shapes."""
    dtype, q, k, v, ref_b16, ref_f32, sdpa = reference_fixture
    rhs = 2 * (ref_b16 - ref_f32).abs().max().item()
    failures = []
    n = 0
    for cfg in ALL:
        if cfg.dtype.to_torch_dtype() != dtype:
            continue
        n += 1
        out = flash_attention.forward(cfg, q, k, v)
        lhs = (out - ref_b16).abs().max().item()
        err32 = (out.float() - ref_f32.float()).abs().max().item()
        err_sdpa = (out.float() - sdpa.float()).abs().max().item()
        if not (lhs <= rhs and err32 <= TOL[dtype] and err_sdpa <= 2 * TOL[dtype]):
            failures.append((str(cfg), lhs, rhs, err32, err_sdpa))
    assert n >= 50
    assert not failures, failures[:5]


def test_output_buffer_ownership_and_determinism():
    cfg = kc.best_config(kc.DType.BF16)
    qc = ut.QKVConfig(n_heads=3, d_head=128, batch_size=2, seq_len=1024, dtype=torch.bfloat16,
                      device=torch.device(DEV))
    q, k, v, o = ut.generate_qkvo(qc, seed=5)
    out = flash_attention.forward(cfg, q, k, v, o)
    assert out.data_ptr() == o.data_ptr()
    first = out.clone()
    for _ in range(5):  # bitwise run-to-run determinism (racecheck substitute)
        again = flash_attention.forward(cfg, q, k, v)
        assert torch.equal(again, first)
    out2, ms = flash_attention.forward_timed(cfg, q, k, v)
    assert torch.equal(out2, first) and ms > 0


@pytest.mark.parametrize("S", [64, 128, 192, 256, 512])
def test_shortest_sequences_every_variant(S):
    """One to eight K/V tiles per item: prologue-only and single-visit loops, the key-split merge with one
    tile per key group, the speculative check right behind the first tile, the persistent walk with four
    tiles per item -- every device variant whose tiles divide S, against fp32 eager."""
    n = 0
    for cfg in VARIANTS:
        if S % cfg.B_r or S % cfg.B_c:
            continue
        n += 1
        dtype = cfg.dtype.to_torch_dtype()
        gen = torch.Generator(device=DEV).manual_seed(S)
        q, k, v = (torch.randn((3, S, 5, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        out = flash_attention.forward(cfg, q, k, v)
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        assert torch.isfinite(out.float()).all(), str(cfg)
        assert ((out.float() - ref).abs() <= TOL[dtype] * (1 + ref.abs())).all(), str(cfg)
        assert torch.equal(flash_attention.forward(cfg, q, k, v), out), str(cfg)
    assert n >= (10 if S < 256 else 40)


def test_launch_is_capturable_into_a_hip_graph():
    """fa_fwd_launch is one plain asynchronous kernel launch on the caller's stream (persistent variants
    included: no cooperative launch, no host round trip), so a forward can be captured into a hipGraph and
    replayed on new data in the same buffers -- how a serving loop would issue it."""
    for cfg in (kc.best_config(kc.DType.BF16, 1024), kc.best_config(kc.DType.FP16, 320)):
        dtype = cfg.dtype.to_torch_dtype()
        S = 1024 if cfg.B_r == 256 else 384
        gen = torch.Generator(device=DEV).manual_seed(9)
        q, k, v = (torch.randn((2, S, 4, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        o = torch.empty_like(q)
        expect = flash_attention.forward(cfg, q, k, v).clone()   # (also the per-device setup, outside the capture)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            flash_attention.forward(cfg, q, k, v, o)
        o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, expect), str(cfg)
        q2, k2, v2 = (torch.randn((2, S, 4, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        expect2 = flash_attention.forward(cfg, q2, k2, v2).clone()
        q.copy_(q2), k.copy_(k2), v.copy_(v2)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, expect2), str(cfg)


def test_every_variant_is_bitwise_deterministic_under_load():
    """LDS stage-recycling protocol check (no racecheck tool on ROCm): every device variant,
    many workgroups in flight (uneven progress), three runs, identical bits; and variants that
    differ only in schedule (plain / pipelined, DMA / register-staged, first-block shortcut)
    but share tile shapes agree within 2 ulp."""
    for dtype in (torch.bfloat16, torch.float16):
        qc = ut.QKVConfig(n_heads=16, d_head=128, batch_size=8, seq_len=1024, dtype=dtype,
                          device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=9)
        by_shape = {}
        for cfg in VARIANTS:
            if cfg.dtype.to_torch_dtype() != dtype:
                continue
            runs = [flash_attention.forward(cfg, q, k, v) for _ in range(3)]
            assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), str(cfg)
            by_shape.setdefault((cfg.B_r // cfg.n_warps, cfg.B_c), []).append(runs[0])
        for outs in by_shape.values():
            for other in outs[1:]:
                assert (other.float() - outs[0].float()).abs().max().item() <= TOL[dtype]


def test_heads_not_16_and_batch_strides():
    """gotcha G1: strides are runtime values; heads 1, 3, 8, 32."""
    for heads in (1, 3, 8, 32):
        for cfg in (kc.best_config(kc.DType.FP16),
                    kc.FlashForwardKernelConfig(kc.DType.FP16, 128, 64, 32, 4, True, True, True, 2, 2, 0, False, False)):
            qc = ut.QKVConfig(n_heads=heads, d_head=128, batch_size=2, seq_len=512,
                              dtype=torch.float16, device=torch.device(DEV))
            q, k, v = ut.generate_qkv(qc, seed=heads)
            out = flash_attention.forward(cfg, q, k, v)
            ref = ut.py_flash_attention(q, k, v, upcast=True)
            assert (out.float() - ref.float()).abs().max().item() <= TOL[torch.float16]


def test_online_softmax_rescale_is_exercised():
    """A key spike in the LAST-visited block (block 0) forces a large rescale of the
    accumulated O and l (guide rule 26): one Q row against one K row."""
    for dtype, cfg in ((torch.bfloat16, kc.best_config(kc.DType.BF16)),
                       (torch.float16, _native(kc.DType.FP16, 64, 64, 4, False, True))):
        qc = ut.QKVConfig(n_heads=2, d_head=128, batch_size=1, seq_len=1024, dtype=dtype,
                          device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=77)
        k[0, 5, 0] = q[0, 700, 0] * 3.0       # huge logit for (row 700, key 5) in block 0
        k[0, 1000, 1] = q[0, 3, 1] * 3.0      # and one in the FIRST-visited block
        out = flash_attention.forward(cfg, q, k, v)
        ref = ut.py_flash_attention(q, k, v, upcast=True)
        assert torch.isfinite(out.float()).all()
        assert (out.float() - ref.float()).abs().max().item() <= 2 * TOL[dtype]


def test_64_row_variant_against_its_lazy_rescale_restatement():
    """(B_r 256, B_c 64, 4 waves, buffer) keeps O and l relative to a reference max that moves
    only past a threshold.  Checked against the CPU restatement of exactly that
    arithmetic and against fp32 eager, on data whose row maxima keep rising along the visit order
    (keys near the start of the sequence are visited last)."""
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        cfg = _native(name, 256, 64, 4, True, False)
        qc = ut.QKVConfig(n_heads=2, d_head=128, batch_size=1, seq_len=1024, dtype=dtype,
                          device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=31)
        stair = (k.float() * torch.linspace(8, 1, 1024, device=DEV).view(1, -1, 1, 1)).to(dtype)
        spike = k.clone()
        spike[0, 5, 0] = q[0, 700, 0] * 3.0
        for kk in (k, stair, spike):
            out = flash_attention.forward(cfg, q, kk, v)
            ref = ut.py_flash_attention(q, kk, v, upcast=True).float()
            oracle = fo.blockwise_forward_lazy(q.cpu(), kk.cpu(), v.cpu(), 256, 64).float()
            assert torch.isfinite(out.float()).all()
            tol = TOL[dtype] * (1 + ref.abs())
            assert ((out.float() - ref).abs() <= tol).all()
            assert ((out.float().cpu() - oracle).abs() <= tol.cpu()).all()


@pytest.mark.parametrize("step", [7.9, 8.1, 3.95])
def test_row_max_rising_by_about_the_threshold_every_tile(step):
    """Adversarial for the lazy rescale (TAU = 8): every visited tile lifts the row max by `step`
    binades -- just under the threshold (P reaches ~2^7.9 before the reference max moves, every
    second tile: the worst-case P magnitude, in fp16 too), just over it (a rescale at every tile), and
    half of it.  The logits are exact by construction (q = a u, k_j = b_j u with u in {-1, +1}^128),
    the other keys of a tile are random, and V is N(0,1).  Lazy and speculative builds, both dtypes
    (16 tiles x 7.9 binades also overflows the speculative first pass in both: second pass), against fp32
    eager and the lazy restatement."""
    S, H, B = 1024, 2, 1
    c = 1.4426950408889634 / 128 ** 0.5
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        gen = torch.Generator(device=DEV).manual_seed(int(step * 100))
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        u = _sign_vector(11)
        q[0, 100:132, 0] = (0.25 * u).to(dtype)           # rows 100..131 of head 0: q.k_j = 32 b_j exactly
        for t in range(S // 64):                            # tile t is visited at position 15 - t
            rise = step * (S // 64 - 1 - t)                 # binades above the first-visited tile's planted key
            k[0, 64 * t + 5, 0] = ((rise / c) / 32.0 * u).to(dtype)
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        for opt in (False, True):
            cfg = _persistent_cfg(name, opt)
            out = flash_attention.forward(cfg, q, k, v)
            assert torch.isfinite(out.float()).all(), (str(dtype), opt)
            tol = TOL[dtype] * (1 + ref.abs())
            assert ((out.float() - ref).abs() <= tol).all(), (str(dtype), opt, step)
        lazy = fo.blockwise_forward_lazy(q.cpu(), k.cpu(), v.cpu(), 256, 64).float()
        out = flash_attention.forward(_persistent_cfg(name, False), q, k, v)
        assert ((out.float().cpu() - lazy).abs() <= (TOL[dtype] * (1 + ref.abs())).cpu()).all()


@pytest.mark.parametrize("shape", [(8, 16, 1024), (32, 16, 256), (40, 16, 256), (5, 7, 512), (3, 16, 2048)],
                         ids=lambda s: "B%d_H%d_S%d" % s)
def test_persistent_walk_seams(shape):
    """The 64-rows-per-wave kernel is persistent: 256 workgroups walk B*H*S/256 items, keep the K/V
    tile stream running across item seams and form the next item's S(0) in the last visit of the
    current one.  More items than workgroups (uneven last round included), the 4-tile minimum
    (S = 256: every request already belongs to the next item), batch*heads not a multiple of 8 (no
    XCD-aware mapping): compare with the 32-rows-per-wave kernel and with fp32 eager on a slice, and
    require bitwise run-to-run determinism (a cold first run included: the operand-register hazard
    the schedule guards against showed up exactly there)."""
    B, H, S = shape
    for dtype, name, opt in ((torch.bfloat16, kc.DType.BF16, False), (torch.float16, kc.DType.FP16, False),
                             (torch.bfloat16, kc.DType.BF16, True), (torch.float16, kc.DType.FP16, True)):
        cfg = _native(name, 256, 64, 4, True, opt)
        other = _native(name, 128, 64, 4, True, False)
        qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=B + S)
        torch.cuda.synchronize()
        runs = [flash_attention.forward(cfg, q, k, v) for _ in range(4)]
        assert all(torch.equal(runs[0], r) for r in runs[1:]), (str(cfg), shape)
        ref = flash_attention.forward(other, q, k, v)
        assert (runs[0].float() - ref.float()).abs().max().item() <= 2 * TOL[dtype]
        sl = slice(B - 1, B)  # the items served last
        eager = ut.py_flash_attention(q[sl], k[sl], v[sl], upcast=True)
        assert (runs[0][sl].float() - eager.float()).abs().max().item() <= TOL[dtype]


def _native(name, B_r, B_c, n_waves, buffer, speculative):
    """(B_r, B_c, n_waves) LDS-DMA config with / without the speculative softmax (a NativeKernelConfig: the
    13-field key's optimized_softmax keeps the reference's meaning and never selects it)."""
    return kc.NativeKernelConfig(name, 128, B_r, B_c, n_waves, True, True, True, 0, 0, 0, buffer, False,
                                 speculative_softmax=speculative)


def _launch_ex(lib, args, cfg, causal=False, allow_ragged=False, stream=None):
    """fa_fwd_launch_ex straight through the C ABI with the config's native options in fa_fwd_opts (the 13-field key
    alone never selects them)."""
    import ctypes
    opts = _capi.make_opts(causal=causal, allow_ragged=allow_ragged, speculative=kc.wants_speculative(cfg),
                           prescaled_q=bool(getattr(cfg, "prescaled_q", False)))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if stream is None else stream
    return _capi.check(lib.fa_fwd_launch_ex(ctypes.byref(args), ctypes.byref(opts), stream))


def _persistent_cfg(name, speculative):
    return _native(name, 256, 64, 4, True, speculative)


def _late_key(cfg, S, key):
    """Sequence position of the key that is visited `key`-th FROM THE END of an item's walk (key < 64: in the LAST visited
    tile -- the place for a spike that has to fail the speculative first pass, whose reference is the row max of the FIRST
    visited tile).  The reference's order, last-to-first (forward_kernel.cuh:142): position `key`; the speculative first
    pass of the hand-placed persistent kernel walks first-to-last since round 6 (kc.walks_kv_forward): S - 1 - key."""
    return S - 1 - key if kc.walks_kv_forward(cfg, seq_len=S) else key


def _sign_vector(seed):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randint(0, 2, (128,), generator=gen).float() * 2 - 1).to(DEV)


def test_speculative_softmax_against_its_restatement():
    """speculative_softmax on the persistent kernel: the first
    pass keeps the row max of an item's first tile as the reference for the whole item.  On inputs
    that do not trip the overflow check the result is the lazy restatement with an infinite
    threshold; rising logits (P up to ~2^40 in bf16) lose nothing."""
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        cfg = _persistent_cfg(name, True)
        assert kc.uses_speculative_softmax(cfg)
        qc = ut.QKVConfig(n_heads=3, d_head=128, batch_size=2, seq_len=1024, dtype=dtype, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=91)
        kk_list = [k]
        if dtype == torch.bfloat16:  # fp16 takes the second pass on such data (next test)
            # (rising along the walk: first-to-last for this kernel's speculative pass)
            kk_list.append((k.float() * torch.linspace(1, 6, 1024, device=DEV).view(1, -1, 1, 1)).to(dtype))
        for i_k, kk in enumerate(kk_list):
            out = flash_attention.forward(cfg, q, kk, v)
            ref = ut.py_flash_attention(q, kk, v, upcast=True).float()
            assert kc.walks_kv_forward(cfg)
            oracle = fo.blockwise_forward_spec(q.cpu(), kk.cpu(), v.cpu(), 256, 64, kv_forward=True).float()
            assert torch.isfinite(out.float()).all()
            tol = TOL[dtype] * (1 + ref.abs())
            assert ((out.float() - ref).abs() <= tol).all()
            # (the rising keys put a row's weight on its last few keys: P's 16-bit rounding no longer averages out, the
            # restatement itself sits at 0.67 of the bar against fp32 eager -- two roundings of one function: twice the bar)
            worst = ((out.float().cpu() - oracle).abs() / tol.cpu()).max().item()
            assert worst <= (2.0 if i_k else 1.0), (str(dtype), i_k, worst)


def test_long_sequences_alternate_the_kv_direction_by_rounds():
    """Round 6 (VERDICT r05 task 5): at seq_len 16384 a head's 64 Q blocks take two rounds of an XCD's 32
    workgroups, and its 8 MiB of K / V do not fit the XCD's 4 MiB L2: the launcher takes the form of the speculative plain
    kernel whose second round walks [tile 0, then last-to-second] (kc.kv_walk_alternates).  Checked: against the CPU
    restatement with the same order and against fp32 eager; an attention sink at the first keys sends NO item to the second
    pass in fp16 (tile 0 stays the first tile of both directions); a spiked key in the tile a reversed item visits last fails
    exactly that item, redone to the lazy variant's bits; and a sample's bits do not depend on the batch it sits in."""
    S = 16384
    for dtype, name in ((torch.float16, kc.DType.FP16), (torch.bfloat16, kc.DType.BF16)):
        cfg, lazy = _persistent_cfg(name, True), _persistent_cfg(name, False)
        G = kc.kv_walk_alternates(cfg, 8, S)
        assert G == 32 and kc.kv_walk_alternates(cfg, 8, 8192) == 0 and kc.kv_walk_alternates(cfg, 4, S) == 0
        gen = torch.Generator(device=DEV).manual_seed(77)
        q, k, v = (torch.randn((1, S, 8, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        a = (12.0 * 128 ** 0.5) ** 0.5           # an attention sink worth +12 nats at the first four keys (bench.py --data sink)
        q[..., 0] = a
        k[..., 0] = 0
        k[:, :4, :, 0] = a
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        out, _ = flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
        assert stats.tolist() == [8 * 64, 0], (str(dtype), stats.tolist())
        out_lazy = flash_attention.forward(lazy, q, k, v)
        # (the four sink keys carry a row: outputs of magnitude 2 .. 4 -- the bar is relative)
        assert ((out.float() - out_lazy.float()).abs() <= TOL[dtype] * (1 + out_lazy.float().abs())).all()
        # two heads against the restatement with the launch's own order (the direction is a function of the Q block), and
        # against fp32 eager
        hs = [0, 5]
        qs, ks, vs = (t[:, :, hs].contiguous() for t in (q, k, v))
        want = fo.blockwise_forward_spec(qs.cpu(), ks.cpu(), vs.cpu(), 256, 64, kv_forward=True, alt_group=G).float()
        for i, h_ in enumerate(hs):
            e32 = ut.py_flash_attention(*(t[:, :, i:i + 1].contiguous() for t in (qs, ks, vs)), upcast=True).float()
            tol = TOL[dtype] * (1 + e32.abs())
            assert ((out[:, :, h_:h_ + 1].float() - e32).abs() <= tol).all(), (str(dtype), h_)
            assert ((out[:, :, h_:h_ + 1].float().cpu() - want[:, :, i:i + 1]).abs() <= tol.cpu()).all(), (str(dtype), h_)
            del e32, tol
        # the same sample inside a batch of two: the same bits (the direction never depends on the batch)
        q2, k2, v2 = (torch.cat([t, t], dim=0) for t in (q, k, v))
        o2 = flash_attention.forward(cfg, q2, k2, v2)
        assert torch.equal(o2[0], out[0]) and torch.equal(o2[1], out[0])
        del q2, k2, v2, o2
        # a reversed item (Q block 40: (40 // 32) is odd) that fails: its half is redone to the lazy variant's bits
        u = _sign_vector(5).to(dtype)
        k[0, 64 + 9, 3] = 30.0 * u
        q[0, 40 * 256 + 7, 3] = 30.0 * u
        stats.zero_()
        out, _ = flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
        # (the spiked key's logits are N(0, 43 binades) over the rows of its head: beyond fp16's 15 in every item of the head, and
        # beyond bf16's 120 in most -- some row of 256 reaches +2.8 sigma; the item of the spiked ROW fails for certain)
        assert (stats.tolist()[1] == 64) if dtype == torch.float16 else (1 <= stats.tolist()[1] <= 64), (str(dtype), stats.tolist())
        out_lazy = flash_attention.forward(lazy, q, k, v)
        assert torch.isfinite(out.float()).all()
        assert torch.equal(out[0, 40 * 256:40 * 256 + 128, 3], out_lazy[0, 40 * 256:40 * 256 + 128, 3])
        assert ((out.float() - out_lazy.float()).abs() <= TOL[dtype] * (1 + out_lazy.float().abs())).all()


@pytest.mark.parametrize("psq", [False, True], ids=["exact_c", "prescaled_q"])
@pytest.mark.parametrize("rise", ["overflow", "moderate"])
def test_speculative_softmax_second_pass(rise, psq):
    """An item whose logits rise far above its first tile's row max fails the epilogue's check and is
    run again by the lazy-rescale schedule after the walk: its 256 rows must then be BIT-identical to
    what the lazy-rescale build (speculative_softmax = False) computes, every other item keeps the
    first pass's result, and everything stays within tolerance of fp32 eager.  `overflow`: q.k c of
    ~14 000 binades (exp2 overflows fp32: inf / NaN in the first pass) -- both dtypes fail.
    `moderate`: ~20 binades -- fp16 fails (P would pass 65504), bf16 does not."""
    a = 30.0 if rise == "overflow" else 1.107
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        spec, safe = (replace(_persistent_cfg(name, sp), prescaled_q=psq) for sp in (True, False))
        # 2 * 3 * 4 = 24 items on 24 workgroups, and 40 * 16 * 1 = 640 items on 256 (ordinals 0..2)
        for (B, H, S, b_, h_, rows, key) in ((2, 3, 1024, 1, 2, slice(300, 310), 0), (40, 16, 256, 33, 5, slice(17, 19), 70)):
            qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype, device=torch.device(DEV))
            q, k, v = ut.generate_qkv(qc, seed=7 + B)
            u = _sign_vector(B).to(dtype)
            k[b_, _late_key(spec, S, key), h_] = a * u           # (0: in the LAST visited tile, 70: in the last but one)
            q[b_, rows, h_] = a * u
            stats = torch.zeros(2, dtype=torch.int32, device=DEV)
            out, _ = flash_attention_kernels.forward(spec, q, k, v, None, stats=stats)
            out_safe = flash_attention.forward(safe, q, k, v)
            assert torch.isfinite(out.float()).all()
            qb = rows.start // 256
            blk = (b_, slice(256 * qb, 256 * qb + 256), h_)
            # (`moderate` in fp16: 20 binades at once are beyond its 15 -- unless the guard has already moved the row's
            # reference up by then, which it does once l has passed 2^7: the counter says which it was)
            takes_second_pass = stats[1].item() > 0
            assert takes_second_pass or rise == "moderate", (str(dtype), rise, B)
            assert not (takes_second_pass and rise == "moderate" and dtype == torch.bfloat16)
            if takes_second_pass:
                # (round 6: the plain 64-row form redoes a failed item as its failed 128-row HALVES, one 32-row tile per wave
                # -- the same arithmetic per row; the pre-scaled-Q build walks the whole item again)
                if not psq:
                    h0 = 256 * qb + 128 * ((rows.start % 256) // 128)
                    blk = (b_, slice(h0, h0 + 128), h_)
                assert torch.equal(out[blk], out_safe[blk]), (str(dtype), rise, B)
            ref = ut.py_flash_attention(q, k, v, upcast=True).float()
            # (the pre-scaled Q moves a logit by ~|k| |q c| 2^-9: with these 30-sigma keys its bar is wider)
            tol = TOL[dtype] * (1 + ref.abs()) * (16 if psq else 1)
            assert ((out.float() - ref).abs() <= tol).all()
            assert ((out_safe.float() - ref).abs() <= tol).all()
            for _ in range(3):  # both passes are deterministic
                assert torch.equal(flash_attention.forward(spec, q, k, v), out)


def test_speculative_second_pass_redoes_only_the_failed_halves():
    """Round 6: the 64-row speculative plain kernel redoes a failed item as its failed 128-row halves -- a walk over half
    items, one 32-row tile per wave (the one-Q-tile-per-wave machinery inside the 64-row kernel), half the time of a walk
    over whole items.  fp16, a moderate rise (20 binades) planted in a few rows: only those rows' item fails, in the half
    that holds them.  That half comes out BIT-identical to the lazy variant's rows (per 32-row tile the same arithmetic),
    the counter counts the ITEM once -- also when both of its halves fail --, and everything is inside the tolerance."""
    dtype, name = torch.float16, kc.DType.FP16
    spec, safe = _persistent_cfg(name, True), _persistent_cfg(name, False)
    B, H, S = 2, 3, 1024
    for rows_list in ([slice(700, 705)], [slice(520, 523)], [slice(530, 533), slice(760, 764)], [slice(10, 12), slice(900, 903)]):
        qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=23)
        u = _sign_vector(9).to(dtype)
        k[1, _late_key(spec, S, 5), 2] = 1.107 * u
        for rows in rows_list:
            q[1, rows, 2] = 1.107 * u
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        out, _ = flash_attention_kernels.forward(spec, q, k, v, None, stats=stats)
        lazy = flash_attention.forward(safe, q, k, v)
        items = {r.start // 256 for r in rows_list}
        assert stats.tolist() == [B * H * (S // 256), len(items)], (rows_list, stats.tolist())
        for r in rows_list:
            h0 = 128 * (r.start // 128)
            assert torch.equal(out[1, h0:h0 + 128, 2], lazy[1, h0:h0 + 128, 2]), rows_list
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        assert ((out.float() - ref).abs() <= TOL[dtype] * (1 + ref.abs())).all()
        assert torch.equal(flash_attention.forward(spec, q, k, v), out)


@pytest.mark.parametrize("family", ["persistent", "ring", "32-row", "16-row"])
def test_speculative_softmax_limit_and_large_values(family):
    """The bf16 limit of the first pass.  32- and 16-rows-per-wave kernels: l < 2^64 (spec_limit) -- a rise of ~50 binades
    above the first visited tile stays in the first pass, ~75 binades takes the second; there the rows equal the
    speculative_softmax = False build bit for bit.  The persistent kernel guards its first pass (spec_guard: above l = 2^32 it
    rescales O and l by an exact power of two and looks at O itself), so its limit only has to keep P finite: 2^120 --
    50 and 75 binades stay in the first pass (with V scaled by 2^40: |O| <= 2^115 finite), ~135 binades takes the second.
    `ring`: (128, 64, 4)+buffer at seq_len % 256 == 0 IS the persistent kernel (one Q tile per wave), with its guard and its
    limit; `32-row`: the same configuration at another multiple of 128 is the compiler-scheduled body with the 2^64 limit.
    Everything stays finite and within relative tolerance of fp32 eager; the counter says which pass ran."""
    cfg_of = {"persistent": lambda o: _persistent_cfg(kc.DType.BF16, o),
              "ring": lambda o: _native(kc.DType.BF16, 128, 64, 4, True, o),
              "32-row": lambda o: _native(kc.DType.BF16, 128, 64, 4, True, o),
              "16-row": lambda o: _native(kc.DType.BF16, 64, 32, 4, False, o)}[family]
    spec, safe = cfg_of(True), cfg_of(False)
    B, H, S, b_, h_ = 2, 3, (896 if family == "32-row" else 1024), 1, 2
    guarded = family in ("persistent", "ring")
    assert kc.has_ring_form(spec) == (family in ("ring", "32-row"))
    vscale = 2.0 ** 40 if guarded else 2.0 ** 50
    cases = ((50.0, False), (75.0, False), (135.0, True)) if guarded else ((50.0, False), (75.0, True))
    for binades, second in cases:
        a = (binades / (128 * 1.4426950408889634 / 128 ** 0.5)) ** 0.5   # q.k c = a^2 128 c binades above an N(0, 1) tile
        qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=torch.bfloat16, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=11)
        u = _sign_vector(5).to(torch.bfloat16)
        k[b_, _late_key(spec, S, 3), h_] = a * u                 # in the LAST visited tile
        q[b_, 256:512, h_] = a * u           # one whole Q block (two / four workgroups of the smaller tilings)
        v = (v.float() * vscale).to(torch.bfloat16)
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        out, _ = flash_attention_kernels.forward(spec, q, k, v, None, stats=stats)
        out_safe = flash_attention.forward(safe, q, k, v)
        assert torch.isfinite(out.float()).all() and torch.isfinite(out_safe.float()).all()
        blk = (b_, slice(256, 512), h_)
        # the spiked key sends EVERY query of its head off (+-a * sqrt(128) sigma): the other Q blocks of the head may fail too,
        # the spiked block must
        assert (stats[1].item() >= 256 // spec.B_r) == second, (family, binades, stats.tolist())
        if second:
            assert torch.equal(out[blk], out_safe[blk]), (family, binades)
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        tol = TOL[torch.bfloat16] * (vscale + ref.abs())
        assert ((out.float() - ref).abs() <= tol).all(), (family, binades)
        assert ((out_safe.float() - ref).abs() <= tol).all(), (family, binades)


def test_speculative_guard_rescues_rising_logits():
    """The guard of the persistent kernel's first pass (spec_guard): logits that keep rising along the visit order -- here
    a staircase of +6 binades per 64-key tile over 64 tiles, ~380 binades in all, far beyond any fixed limit -- never fail:
    every four visits the wave brings O, l and its reference down by an exact power of two.  No item is redone (the
    counter says so), the result is within tolerance of fp32 eager, and it equals the lazy-rescale build's to 2 ulp.  fp16
    has 15 binades in all (threshold 2^13 since round 4 -- 2^11 made every wave take the rescue near the end of a
    16384-key item of plain N(0, 1) data -- limit 2^15: two binades of rise per four visits): a staircase of +0.2 binades
    per tile passes (+0.3 too, +0.4 for most items, +0.5 for none), +6 does not."""
    for dtype, name, step, redo in ((torch.bfloat16, kc.DType.BF16, 6.0, False), (torch.float16, kc.DType.FP16, 0.2, False),
                                    (torch.float16, kc.DType.FP16, 6.0, True)):
        B, H, S = 2, 4, 4096
        gen = torch.Generator(device=DEV).manual_seed(5)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        # one head dimension carries the staircase: q[..., 0] = a, k[key, ..., 0] = a * (tiles along the walk)
        a = (step / 0.12751743) ** 0.5
        # (both kernels below walk first-to-last in their speculative pass: the staircase rises with the position)
        tile_along_walk = torch.arange(S, device=DEV) // 64
        q[..., 0] = a
        k[..., 0] = (a * tile_along_walk.float()).view(1, S, 1).to(dtype)
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        tol = TOL[dtype] * (1 + ref.abs())
        # the (256, 64, 4) kernel, and its one-Q-tile-per-wave form behind (128, 64, 4)+buffer (seq_len % 256 == 0)
        for spec, safe in ((_persistent_cfg(name, True), _persistent_cfg(name, False)),
                           (_native(name, 128, 64, 4, True, True), _native(name, 128, 64, 4, True, False))):
            stats = torch.zeros(2, dtype=torch.int32, device=DEV)
            out, _ = flash_attention_kernels.forward(spec, q, k, v, None, stats=stats)
            assert torch.isfinite(out.float()).all()
            assert (stats[1].item() > 0) == redo, (str(spec), step, stats.tolist())
            assert ((out.float() - ref).abs() <= tol).all(), (str(spec), step)
            out_safe = flash_attention.forward(safe, q, k, v)
            assert ((out.float() - out_safe.float()).abs() <= 2 * tol).all()
            assert torch.equal(flash_attention.forward(spec, q, k, v), out)


def test_speculative_softmax_on_the_32_row_kernels_starts_over():
    """speculative_softmax on the double-buffered 32-rows-per-wave variants (the key-split (64, 64, 4) form
    and the 16-rows-per-wave (64, 32, 4) kernel too): a workgroup whose check fails runs its item again with the
    running max.  That second attempt is the arithmetic of the same variant without the flag, so the
    rows of the failed workgroup must equal the speculative_softmax = False build bit for bit; all rows
    stay within tolerance of fp32 eager."""
    shapes = [(128, 64, 4, True), (128, 64, 4, False), (128, 32, 4, True), (256, 128, 8, False), (64, 64, 4, True),
              (64, 64, 4, False), (256, 64, 8, True), (64, 32, 4, False)]
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        data = {}
        for S in (1024, 896):   # 896: a multiple of 128 that is not one of 256 (see below)
            qc = ut.QKVConfig(n_heads=3, d_head=128, batch_size=2, seq_len=S, dtype=dtype, device=torch.device(DEV))
            q, k, v = ut.generate_qkv(qc, seed=19)
            u = _sign_vector(5).to(dtype)
            k[1, 3, 2] = 30.0 * u            # key 3 lies in the LAST visited tile
            q[1, 600:604, 2] = 30.0 * u      # rows 600..603 of (batch 1, head 2)
            ref = ut.py_flash_attention(q, k, v, upcast=True).float()
            data[S] = (q, k, v, ref, TOL[dtype] * (1 + ref.abs()))
        for B_r, B_c, nw, buf in shapes:
            spec = _native(name, B_r, B_c, nw, buf, True)
            plain = replace(spec, speculative_softmax=False)
            assert kc.uses_speculative_softmax(spec) and not kc.uses_speculative_softmax(plain)
            # (128, 64, 4) + buffer without the flag is served by its hand-placed RING FORM (lazy rescale) where seq_len is
            # a multiple of 256: "the same variant without the flag" is the 32-rows-per-wave body only off those lengths
            q, k, v, ref, tol = data[896 if kc.has_ring_form(plain) else 1024]
            out, out_plain = flash_attention.forward(spec, q, k, v), flash_attention.forward(plain, q, k, v)
            assert torch.isfinite(out.float()).all(), str(spec)
            blk = slice((600 // B_r) * B_r, (600 // B_r) * B_r + B_r)   # the workgroup that holds rows 600..603
            assert torch.equal(out[1, blk, 2], out_plain[1, blk, 2]), str(spec)
            assert ((out.float() - ref).abs() <= tol).all(), str(spec)
            assert torch.equal(flash_attention.forward(spec, q, k, v), out)


def test_speculative_softmax_causal_second_pass():
    """The causal form of the persistent kernel under speculative_softmax: a wave's reference is the row max
    of its diagonal tile.  A key far below the diagonal with a huge logit (visited later) overflows the first
    pass; the item is redone by the lazy-rescale schedule: bit-identical to the build without the flag for
    that item, within tolerance of the masked fp32 eager result everywhere."""
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        spec, safe = _persistent_cfg(name, True), _persistent_cfg(name, False)
        for a in (30.0, 1.107):
            gen = torch.Generator(device=DEV).manual_seed(41)
            q, k, v = (torch.randn((2, 1024, 3, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
            u = _sign_vector(8).to(dtype)
            k[1, 10, 1] = a * u                 # key 10: below the diagonal of every later row
            q[1, 700:708, 1] = a * u            # rows 700..707 (Q block 2, wave 2)
            stats = torch.zeros(2, dtype=torch.int32, device=DEV)
            out = flash_attention.forward_ex(spec, q, k, v, causal=True, stats=stats)
            out_safe = flash_attention.forward_ex(safe, q, k, v, causal=True)
            assert torch.isfinite(out.float()).all()
            assert stats[1].item() > 0 or a < 2     # (a = 1.107 in fp16: redone unless the guard had moved the row's reference up)
            if stats[1].item() > 0:
                assert torch.equal(out[1, 512:768, 1], out_safe[1, 512:768, 1]), (str(dtype), a)
            for b_, h_ in ((1, 1), (0, 0)):
                qs, ks, vs = (t[b_:b_ + 1, :, h_:h_ + 1].contiguous() for t in (q, k, v))
                eager = fo.eager_attention_masked(qs.cpu(), ks.cpu(), vs.cpu(), True)
                assert _rel_ok(out[b_:b_ + 1, :, h_:h_ + 1].cpu(), eager, dtype), (str(dtype), a, b_, h_)
            assert torch.equal(flash_attention.forward_ex(spec, q, k, v, causal=True), out)


def test_speculative_softmax_causal_second_pass_starting_at_an_odd_round():
    """More items than workgroups, causal: odd rounds of the walk run their window of Q blocks in reverse, and the
    second pass starts at whatever round the first failed item sits in.  (tools/soak.py found the first item of a
    walk taking its Q block un-reversed: the second pass then redid the mirror block and left the failed one as the
    first pass had stored it -- inf / NaN rows.)  Failed items in odd and even rounds, every (batch, head) checked."""
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        spec, safe = _persistent_cfg(name, True), _persistent_cfg(name, False)
        B, H, S = 4, 16, 4096                       # 1024 items on 256 workgroups: rounds 0 .. 3
        gen = torch.Generator(device=DEV).manual_seed(9)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        u = _sign_vector(2).to(dtype)
        for b_, h_, key, row in ((1, 3, 3088, 3500), (0, 0, 5, 300), (2, 9, 1000, 1469), (3, 15, 2047, 4095), (1, 12, 700, 2100)):
            k[b_, key, h_] = 30.0 * u
            q[b_, row, h_] = 30.0 * u
        out = flash_attention.forward_ex(spec, q, k, v, causal=True)
        assert torch.isfinite(out.float()).all(), str(dtype)
        out_safe = flash_attention.forward_ex(safe, q, k, v, causal=True)
        ulp = TOL[dtype]
        for b_ in range(B):
            qf, kf, vf = (t[b_].float().permute(1, 0, 2) for t in (q, k, v))          # fp32 masked attention on the device
            sc = (qf @ kf.transpose(-1, -2)) / 128 ** 0.5
            sc = sc.masked_fill(torch.ones(S, S, dtype=torch.bool, device=DEV).triu(1), float("-inf"))
            ref = (torch.softmax(sc, dim=-1) @ vf).permute(1, 0, 2)
            for o_ in (out, out_safe):
                assert ((o_[b_].float() - ref).abs() <= ulp * (1 + ref.abs())).all(), (str(dtype), b_)
        assert torch.equal(flash_attention.forward_ex(spec, q, k, v, causal=True), out)


def test_two_streams_launching_at_once():
    """Launches of different shapes and variants queued on two streams at the same time (the persistent kernel sizes
    its grid to the whole chip and owns a CU's LDS: workgroups of the other launch simply wait): every result must
    equal the one computed alone."""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    jobs = []
    for i, (cfg, shape) in enumerate(((_persistent_cfg(kc.DType.BF16, True), (4, 2048, 16, 128)),
                                      (_native(kc.DType.FP16, 128, 64, 4, True, True), (3, 1024, 8, 128)),
                                      (_persistent_cfg(kc.DType.FP16, False), (2, 4096, 8, 128)),
                                      (_native(kc.DType.BF16, 64, 64, 4, True, True), (5, 512, 7, 128)))):
        gen = torch.Generator(device=DEV).manual_seed(50 + i)
        q, k, v = (torch.randn(shape, dtype=cfg.dtype.to_torch_dtype(), device=DEV, generator=gen) for _ in range(3))
        jobs.append((cfg, q, k, v, flash_attention.forward(cfg, q, k, v)))
    torch.cuda.synchronize()
    outs = []
    for rep in range(20):
        for j, (cfg, q, k, v, _) in enumerate(jobs):
            with torch.cuda.stream(s1 if (j + rep) % 2 else s2):
                outs.append((j, flash_attention.forward(cfg, q, k, v)))
    torch.cuda.synchronize()
    for j, o in outs:
        assert torch.equal(o, jobs[j][4]), j


def test_soak_of_the_persistent_kernel_short():
    """tools/soak.py for a few seconds with a fixed seed: random shapes, dtypes, causal / ragged / plain, speculative or
    not, spikes, a second stream disturbing memory; every launch checked against fp32 attention and repeated."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "12", "7"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_jitter_build_matches_the_product_bit_for_bit():
    """SURVEY.md 5's racecheck substitute (reference: tools/debug/check_race.sh:3-4, the FA_DEBUG build of setup.py:15,37-38):
    lib/libfa_hip_jitter.so is the product's source with -DFA_JITTER -- every wave of the persistent kernel sleeps a
    pseudo-random 0 .. 7 x 64 cycles in front of every DMA piece, sync point and (one time in eight) operand wait, so the
    four waves of a workgroup drift apart by up to a visit and meet every step of the ring protocol (counted vmcnt waits,
    the barrier two MFMAs into a visit, tiles requested three visits ahead) in another order at every launch.  A missing
    wait or a stage overwritten too early shows as different bits.  tools/jitter_check.py: twelve seeded cases (item seams, the
    speculative second pass, lazy rescales, causal, ragged, both dtypes), three launches each; the jitter build's hashes
    must equal the product library's and repeat.  Then the soak (random shapes against fp32 attention) under it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jit = os.path.join(root, "flash_attention_from_scratch_amd", "lib", "libfa_hip_jitter.so")
    assert os.path.exists(jit), "lib/libfa_hip_jitter.so is not built (make -C flash_attention_from_scratch_amd/csrc jitter)"
    env = {k: v for k, v in os.environ.items() if k != "FA_HIP_LIB"}

    def lines(extra):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "jitter_check.py")], cwd=root, capture_output=True,
                           text=True, timeout=900, env={**env, **extra})
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got = [ln for ln in r.stdout.splitlines() if ln.startswith("case ")]
        assert len(got) == 12 and all(" finite=1 repeat=1 " in ln for ln in got), r.stdout[-3000:]
        return got, r.stdout.splitlines()[0]
    product, which_p = lines({})
    jittered, which_j = lines({"FA_HIP_LIB": jit})
    assert "libfa_hip_jitter.so" in which_j and "libfa_hip_jitter.so" not in which_p
    assert product == jittered, "\n".join(f"{a}\n{b}" for a, b in zip(product, jittered) if a != b)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "15", "11"], cwd=root, capture_output=True,
                       text=True, timeout=900, env={**env, "FA_HIP_LIB": jit})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_adaptive_speculative_mode_demotes_after_a_reported_redo():
    """fa_speculative_mode ADAPTIVE (include/fa_hip.h; opt-in since round 6 -- best_config() is the stateless always-
    speculative form): a speculative launch that had to
    compute items twice stores its sequence number into the pinned report word of ITS device variant (ABI 5); the adaptive launches enqueued
    after the library has seen that report take the non-speculative variant for `hold` launches, then the speculative one
    is probed again.  Outputs are inside the tolerance whichever variant served."""
    from flash_attention_from_scratch_amd import _capi
    dev = torch.cuda.current_device()
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        cfg = replace(kc.best_config(name, 1024), adaptive_softmax=True)
        assert cfg.adaptive_softmax and cfg.speculative_softmax and not kc.best_config(name, 1024).adaptive_softmax
        lazy = replace(cfg, speculative_softmax=False, adaptive_softmax=False)
        spec = replace(cfg, adaptive_softmax=False)
        gen = torch.Generator(device=DEV).manual_seed(77)
        q, k, v = (torch.randn((2, 1024, 8, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        # 1. benign data: never demoted, bit-identical to the always-speculative variant
        flash_attention.forward(cfg, q, k, v)
        torch.cuda.synchronize()
        _capi.adaptive_reset(dev)
        st0 = _capi.adaptive_state(dev, cfg)
        assert st0["available"] == 1
        outs = [flash_attention.forward(cfg, q, k, v) for _ in range(6)]
        torch.cuda.synchronize()
        st1 = _capi.adaptive_state(dev, cfg)
        assert st1["launches"] == st0["launches"] + 6 and st1["demoted"] == 0 and st1["reports"] == 0
        want = flash_attention.forward(spec, q, k, v)
        assert all(torch.equal(o, want) for o in outs)
        # 2. a spike (one 30-sigma key against a few queries): the speculative pass fails -> report -> demotion
        ks, qs = k.clone(), q.clone()
        u = _sign_vector(5).to(dtype)
        ks[1, _late_key(spec, 1024, 3), 2] = 30.0 * u
        qs[1, 600:604, 2] = 30.0 * u
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        vs = v
        flash_attention_kernels.forward(spec, qs, ks, vs, None, stats=stats)
        assert stats[1].item() > 0          # (the always-speculative variant does redo items on this input)
        first = flash_attention.forward(cfg, qs, ks, vs)      # adaptive, still speculative: fails, reports
        torch.cuda.synchronize()
        st2 = _capi.adaptive_state(dev, cfg)
        assert st2["last_report"] == st2["launches"] and st2["demoted"] == 0
        demoted = [flash_attention.forward(cfg, qs, ks, vs) for _ in range(5)]
        torch.cuda.synchronize()
        st3 = _capi.adaptive_state(dev, cfg)
        assert st3["reports"] == 1 and st3["demoted"] == 5 and st3["mode"] == 1 and st3["remaining"] == st3["hold"] - 5
        want_lazy = flash_attention.forward(lazy, qs, ks, vs)
        assert all(torch.equal(o, want_lazy) for o in demoted)      # the demoted launches ARE the lazy variant
        # (a failed item's second pass is the lazy schedule: the speculative launch agrees with it on those items, and on
        # the others within the tolerance)
        eager = ut.py_flash_attention(qs, ks, vs, upcast=True).float()
        for o in [first] + demoted:
            assert ((o.float() - eager).abs() <= TOL[dtype] * (1 + eager.abs())).all()
        # 3. the hold runs out: the next launch probes the speculative variant again (fails again here: hold doubles)
        hold = st3["hold"]
        for _ in range(hold - 5):
            flash_attention.forward(cfg, qs, ks, vs)
        torch.cuda.synchronize()
        assert _capi.adaptive_state(dev, cfg)["demoted"] == hold
        probe = flash_attention.forward(cfg, qs, ks, vs)      # the probe: speculative again, an event recorded behind it
        assert _capi.adaptive_state(dev, cfg)["mode"] == 2 and torch.equal(probe, first)
        torch.cuda.synchronize()
        flash_attention.forward(cfg, qs, ks, vs)      # sees the probe's report: a longer hold
        st4 = _capi.adaptive_state(dev, cfg)
        assert st4["reports"] == 2 and st4["hold"] == 2 * hold and st4["demoted"] == hold + 1 and st4["mode"] == 1
        # 4. the data turns benign: the probe behind the (longer) hold succeeds and the device is back to NORMAL
        for _ in range(st4["remaining"]):
            flash_attention.forward(cfg, q, k, v)
        ok_probe = flash_attention.forward(cfg, q, k, v)
        assert _capi.adaptive_state(dev, cfg)["mode"] == 2 and torch.equal(ok_probe, want)
        torch.cuda.synchronize()
        flash_attention.forward(cfg, q, k, v)
        st5 = _capi.adaptive_state(dev, cfg)
        assert st5["mode"] == 0 and st5["hold"] == 32 and st5["reports"] == 2
        _capi.adaptive_reset(dev)


def test_speculative_request_is_dropped_where_the_masked_form_has_no_speculative_build():
    """best_config(bf16, seq_len not a multiple of 256) is the pipelined (128, 64, 4) kernel with the speculative
    softmax; its MASKED forms keep the running max (no speculative build), and kc.softmax_mode(cfg, masked=True) says
    'eager'.  forward_ex(cfg, ..., causal=True) used to raise; it now runs the masked variant the mirror names."""
    for name, dtype in ((kc.DType.BF16, torch.bfloat16), (kc.DType.FP16, torch.float16)):
        cfg = kc.best_config(name, 1000)
        assert (cfg.B_r, cfg.n_warps) == (128, 4) and cfg.speculative_softmax and kc.softmax_mode(cfg, masked=True) != "speculative"
        gen = torch.Generator(device=DEV).manual_seed(9)
        q, k, v = (torch.randn((2, 1024, 4, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        out = flash_attention.forward_ex(cfg, q, k, v, causal=True)
        plain = replace(cfg, speculative_softmax=False, adaptive_softmax=False)
        assert torch.equal(out, flash_attention.forward_ex(plain, q, k, v, causal=True))
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        flash_attention_kernels.forward(cfg, q, k, v, None, causal=True, stats=stats)
        assert stats[1].item() == 0 and stats[0].item() > 0


def test_adaptive_mode_from_two_threads_on_two_streams():
    """The adaptive mode's record of a device variant is shared by every caller of that configuration on the device: two host threads, each on its own
    stream -- one feeding benign data, one data that makes the speculative pass fail -- launch concurrently.  Every output is
    inside the tolerance (whichever variant served it), the counters add up, nothing deadlocks (a probe holds the policy
    lock across its launch and event record)."""
    import threading
    from flash_attention_from_scratch_amd import _capi
    dev = torch.cuda.current_device()
    dtype = torch.bfloat16
    cfg = replace(kc.best_config(kc.DType.BF16, 1024), adaptive_softmax=True)
    gen = torch.Generator(device=DEV).manual_seed(123)
    q, k, v = (torch.randn((2, 1024, 8, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
    ks, qs = k.clone(), q.clone()
    u = _sign_vector(5).to(dtype)
    ks[1, _late_key(cfg, 1024, 3), 2] = 30.0 * u
    qs[1, 600:604, 2] = 30.0 * u
    refs = {"benign": ut.py_flash_attention(q, k, v, upcast=True).float(), "spiky": ut.py_flash_attention(qs, ks, v, upcast=True).float()}
    torch.cuda.synchronize()
    _capi.adaptive_reset(dev)
    before = _capi.adaptive_state(dev, cfg)["launches"]
    outs, errors = {"benign": [], "spiky": []}, []
    n_each = 60

    def worker(name, args):
        try:
            st = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st):
                for i in range(n_each):
                    outs[name].append(flash_attention.forward(cfg, *args))
                    if i % 16 == 15:
                        st.synchronize()      # (lets reports land between batches, as a real caller's host work would)
            st.synchronize()
        except Exception as exc:  # noqa: BLE001
            errors.append((name, repr(exc)))
    threads = [threading.Thread(target=worker, args=("benign", (q, k, v))), threading.Thread(target=worker, args=("spiky", (qs, ks, v)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads) and not errors, errors
    torch.cuda.synchronize()
    after = _capi.adaptive_state(dev, cfg)
    assert after["launches"] == before + 2 * n_each and after["reports"] >= 1 and 0 < after["demoted"] <= 2 * n_each
    for name, ref in refs.items():
        tol = TOL[dtype] * (1 + ref.abs())
        assert len(outs[name]) == n_each
        for o in outs[name]:
            assert ((o.float() - ref).abs() <= tol).all(), name
    _capi.adaptive_reset(dev)


@pytest.mark.parametrize("S", [1000, 2500])
def test_speculative_softmax_ragged_second_pass(S):
    """The ragged form under speculative_softmax: the rounded-up tiles beyond the sequence are masked whole,
    a wave's reference is the row max of the last tile that holds keys (causal: or its diagonal tile, the
    earlier of the two).  A huge logit at a key visited later fails the check; the item is redone."""
    for dtype, name in ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)):
        spec, safe = _persistent_cfg(name, True), _persistent_cfg(name, False)
        for causal in (False, True):
            gen = torch.Generator(device=DEV).manual_seed(S + causal)
            q, k, v = (torch.randn((2, S, 3, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
            u = _sign_vector(S).to(dtype)
            k[1, 7, 2] = 30.0 * u
            q[1, S - 20:S - 12, 2] = 30.0 * u
            out = flash_attention.forward_ex(spec, q, k, v, causal=causal)
            out_safe = flash_attention.forward_ex(safe, q, k, v, causal=causal)
            assert torch.isfinite(out.float()).all()
            blk = slice(((S - 20) // 256) * 256, min(S, ((S - 20) // 256) * 256 + 256))
            assert torch.equal(out[1, blk, 2], out_safe[1, blk, 2]), (str(dtype), S, causal)
            for b_, h_ in ((1, 2), (0, 1)):
                qs, ks, vs = (t[b_:b_ + 1, :, h_:h_ + 1].contiguous() for t in (q, k, v))
                eager = fo.eager_attention_masked(qs.cpu(), ks.cpu(), vs.cpu(), causal)
                assert _rel_ok(out[b_:b_ + 1, :, h_:h_ + 1].cpu(), eager, dtype), (str(dtype), S, causal, b_, h_)
            assert torch.equal(flash_attention.forward_ex(spec, q, k, v, causal=causal), out)


def test_speculative_softmax_fuzz():
    """Seeded fuzz over shapes, spike positions and magnitudes: logits that rise by 0 ... thousands of
    binades anywhere along the visit order, in any number of rows and heads, on the persistent kernel and
    on the one-item kernels (speculative builds): always finite, always within tolerance of fp32 eager,
    always bitwise repeatable."""
    import random
    rng = random.Random(2024)
    shapes = [(128, 64, 4, True), (64, 64, 4, True), (64, 32, 4, False), (256, 64, 4, True), (256, 64, 4, True),
              (256, 64, 4, True)]
    for trial in range(36):
        B_r, B_c, nw, buf = shapes[trial % len(shapes)]
        dtype, name = ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16))[trial & 1]
        cfg = _native(name, B_r, B_c, nw, buf, True)
        B, H = rng.choice([1, 2, 5]), rng.choice([1, 3, 8])
        S = B_r * rng.choice([1, 2, 3, 4, 8] if B_r == 256 else [1, 2, 4, 9, 16])
        gen = torch.Generator(device=DEV).manual_seed(trial)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        for _ in range(rng.choice([0, 1, 1, 2, 4])):
            b_, h_ = rng.randrange(B), rng.randrange(H)
            key, row, n_rows = rng.randrange(S), rng.randrange(S), rng.choice([1, 3, 40])
            a = rng.choice([0.6, 1.1, 2.0, 6.0, 30.0])       # q.k c = a^2 * 128 c: 6 ... 16 000 binades
            u = _sign_vector(rng.randrange(1000)).to(dtype)
            k[b_, key, h_] = a * u
            q[b_, row:row + n_rows, h_] = a * u
        out = flash_attention.forward(cfg, q, k, v)
        ref = ut.py_flash_attention(q, k, v, upcast=True).float()
        assert torch.isfinite(out.float()).all(), (trial, str(cfg))
        assert ((out.float() - ref).abs() <= TOL[dtype] * (1 + ref.abs())).all(), (trial, str(cfg), B, H, S)
        assert torch.equal(flash_attention.forward(cfg, q, k, v), out), (trial, str(cfg))


def test_speculative_softmax_masked_fuzz():
    """The same fuzz through forward_ex on the persistent kernel's masked forms: any seq_len >= 64, causal or
    not, spikes anywhere (also at keys a causal row never sees, and in rows of the rounded-up tail)."""
    import random
    rng = random.Random(77)
    for trial in range(28):
        dtype, name = ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16))[trial & 1]
        cfg = _persistent_cfg(name, True)
        B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 5])
        S = rng.choice([64, 100, 256, 300, 511, 512, 777, 1024, 1500, 2048, 2300])
        causal = bool(rng.getrandbits(1))
        gen = torch.Generator(device=DEV).manual_seed(1000 + trial)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        for _ in range(rng.choice([0, 1, 2, 3])):
            b_, h_ = rng.randrange(B), rng.randrange(H)
            key, row, n_rows = rng.randrange(S), rng.randrange(S), rng.choice([1, 5, 70])
            a = rng.choice([0.8, 1.3, 3.0, 30.0])
            u = _sign_vector(rng.randrange(1000)).to(dtype)
            k[b_, key, h_] = a * u
            q[b_, row:row + n_rows, h_] = a * u
        out = flash_attention.forward_ex(cfg, q, k, v, causal=causal)
        assert torch.isfinite(out.float()).all(), (trial, S, causal)
        eager = fo.eager_attention_masked(q.cpu(), k.cpu(), v.cpu(), causal)
        assert _rel_ok(out.cpu(), eager, dtype), (trial, str(dtype), B, H, S, causal)
        assert torch.equal(flash_attention.forward_ex(cfg, q, k, v, causal=causal), out), (trial, S, causal)


def test_speculative_softmax_second_pass_beyond_ordinal_63():
    """A workgroup records failed items in a 64-bit mask of walk ordinals; ordinals >= 63 share the
    last bit (the second pass then redoes all of them).  130 * 128 items of one Q block each on 256
    workgroups = 65 rounds: spikes in rounds 2, 63 and 64."""
    B, H, S = 130, 128, 256
    dtype, name = torch.bfloat16, kc.DType.BF16
    spec, safe = _persistent_cfg(name, True), _persistent_cfg(name, False)
    gen = torch.Generator(device=DEV).manual_seed(12)
    q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
    u = _sign_vector(3).to(dtype)
    # XCD-aware item numbering (one Q block per head): item -> bh = (item >> 3) * 8 + (item & 7) = item
    spiked = []
    for item in (2 * 256 + 5, 63 * 256 + 77, 64 * 256 + 255, 64 * 256 + 3):
        b_, h_ = divmod(item, H)
        assert b_ < B
        k[b_, 100, h_] = 30.0 * u
        q[b_, 40:44, h_] = 30.0 * u
        spiked.append((b_, h_))
    out = flash_attention.forward(spec, q, k, v)
    out_safe = flash_attention.forward(safe, q, k, v)
    assert torch.isfinite(out.float()).all()
    for b_, h_ in spiked:
        # (the half that holds the spiked rows is redone for certain -- bit-identical to the lazy variant's rows; the other half
        # only if one of its own rows met the spiked key badly enough: round 6 redoes a failed item by its failed halves)
        assert torch.equal(out[b_, :128, h_], out_safe[b_, :128, h_])
    # (rows of a spiked item's OTHER half that met the spiked key keep the first pass's result: outputs of magnitude 2 .. 4
    # there -- the spiked key's V row -- where one bf16 ulp is 2^-6: the bar is relative)
    assert ((out.float() - out_safe.float()).abs() <= TOL[dtype] * (1 + out_safe.float().abs())).all()
    for b_, h_ in spiked + [(0, 0), (B - 1, H - 1), (64, 64)]:
        sl = (slice(b_, b_ + 1), slice(None), slice(h_, h_ + 1))
        ref = ut.py_flash_attention(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), upcast=True)
        assert ((out[sl].float() - ref.float()).abs() <= TOL[dtype] * (1 + ref.float().abs())).all()


def test_error_behaviour_matches_reference():
    cfg = kc.best_config(kc.DType.BF16)
    q = torch.zeros((1, 512, 2, 128), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="same data type"):
        flash_attention.forward(cfg, q, q.half(), q)
    with pytest.raises(RuntimeError, match="dtype does not match"):
        flash_attention.forward(cfg, q.half(), q.half(), q.half())
    with pytest.raises(RuntimeError, match="Only fp16 and bf16"):
        flash_attention.forward(cfg, q.float(), q.float(), q.float())
    with pytest.raises(RuntimeError, match="contiguous"):
        flash_attention.forward(cfg, q.transpose(1, 2), q, q)
    with pytest.raises(RuntimeError, match="multiples of B_r"):
        flash_attention.forward(cfg, q[:, :320].contiguous(), q[:, :320].contiguous(), q[:, :320].contiguous())
    with pytest.raises(RuntimeError, match="not found"):
        flash_attention.forward(replace(cfg, B_c=48), q, q, q)
    with pytest.raises(RuntimeError, match="same shape"):
        flash_attention.forward(cfg, q, q[:, :256].contiguous(), q)
    with pytest.raises(RuntimeError, match="same dtype"):
        flash_attention.forward(cfg, q, q, q, torch.empty_like(q, dtype=torch.float16))


def test_every_variant_is_bit_identical_run_to_run():
    """The MFMAs, the DMA pieces and the epilogue stores are inline asm that hipcc neither pads nor looks into; a
    hazard it cannot see shows as rare run-to-run differences (one was found that way in round 1).  Every device
    variant, several launches with the caches disturbed in between, must give the same bits
    (tools/determinism_probe.py runs the same check for hundreds of launches)."""
    scratch = torch.empty(256 << 20, dtype=torch.int8, device=DEV)
    for cfg in VARIANTS + D64:
        dtype = cfg.dtype.to_torch_dtype()
        gen = torch.Generator(device=DEV).manual_seed(3)
        q, k, v = (torch.randn((3, 1024, 5, cfg.d_head), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        first = flash_attention.forward(cfg, q, k, v)
        for rep in range(6):
            if rep % 2:
                scratch.zero_()
            assert torch.equal(flash_attention.forward(cfg, q, k, v), first), (str(cfg), rep)
    for name, dtype in ((kc.DType.BF16, torch.bfloat16), (kc.DType.FP16, torch.float16)):   # the masked and ragged forms
        for spec in (True, False):
            cfg = _persistent_cfg(name, spec)
            for S, causal in ((1024, True), (1000, False), (1000, True)):
                gen = torch.Generator(device=DEV).manual_seed(4)
                q, k, v = (torch.randn((3, S, 5, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
                first = flash_attention.forward_ex(cfg, q, k, v, causal=causal)
                for rep in range(6):
                    if rep % 2:
                        scratch.zero_()
                    assert torch.equal(flash_attention.forward_ex(cfg, q, k, v, causal=causal), first), (str(cfg), S, causal, rep)


# ---- BASELINE.json full sizes: size-independent properties ----------------------
FULL = [
    ("c1", torch.bfloat16, 4, 16, 4096),
    ("c3", torch.float16, 2, 32, 16384),
    ("c4-shard", torch.bfloat16, 8, 32, 8192),
]


@pytest.mark.parametrize("name,dtype,B,H,S", FULL, ids=[f[0] for f in FULL])
def test_full_size_properties(name, dtype, B, H, S):
    cfg = kc.best_config(kc.DType.BF16 if dtype == torch.bfloat16 else kc.DType.FP16)
    qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype,
                      device=torch.device(DEV))
    q, k, v = ut.generate_qkv(qc, seed=42)
    out = flash_attention.forward(cfg, q, k, v)
    assert torch.isfinite(out.float()).all()
    # (1) linearity in V by a power of two is EXACT in floating point
    #     (RNE of 2x == 2 RNE of x) wherever the result stays a normal number; fp16
    #     subnormals (|o| < 2^-14) round on a fixed grid, so those are compared to 1 grid step
    out2 = flash_attention.forward(cfg, q, k, v * 2)
    normal = out.float().abs() > 2.0 ** -14   # strictly: 2^-14 itself may be a rounded-up subnormal
    assert torch.equal(out2[normal], (out * 2)[normal])
    assert (out2.float() - 2 * out.float()).abs().max().item() <= 2.0 ** -23
    # (2) batch-shard independence (the 8-GPU split): a sub-batch gives identical bits
    sub = flash_attention.forward(cfg, q[B // 2:].contiguous(), k[B // 2:].contiguous(), v[B // 2:].contiguous())
    assert torch.equal(sub, out[B // 2:])
    # (3) convex combination: every output lies within [min V, max V] per (batch, head, d)
    vmin = v.float().amin(dim=1, keepdim=True) - 2e-2
    vmax = v.float().amax(dim=1, keepdim=True) + 2e-2
    assert ((out.float() >= vmin) & (out.float() <= vmax)).all()
    # (4) constant V -> output is that constant (row sums of P / l == 1 within rounding)
    ones = torch.ones_like(v)
    oc = flash_attention.forward(cfg, q, k, ones)
    assert (oc.float() - 1).abs().max().item() <= 2.0 ** -7
    # (5) every 8th head of every sample against eager attention, under the reference's own rule (test.py:57-61:
    #     max|out - eager_16| <= 2 max|eager_16 - eager_f32|, eager in the 16-bit type and in fp32 as py_flash_attention
    #     computes them, here on the device one head at a time) and against fp32 eager to 2 ulp -- the default AND the
    #     running-max (lazy) variant, the reference's arithmetic family (rounds 1-5 looked at ONE head)
    lazy = replace(cfg, speculative_softmax=False)
    out_lazy = flash_attention.forward(lazy, q, k, v)
    for b_ in range(B):
        for h_ in range(0, H, 8):
            sl = (slice(b_, b_ + 1), slice(None), slice(h_, h_ + 1))
            qs, ks, vs = (t[sl].contiguous() for t in (q, k, v))
            e32 = ut.py_flash_attention(qs, ks, vs, upcast=True)
            e16 = ut.py_flash_attention(qs, ks, vs, upcast=False)
            for tag, o_ in (("default", out), ("lazy", out_lazy)):
                lhs, rhs = fo.tolerance_rule(o_[sl], e16, e32)
                assert lhs <= rhs, (name, tag, b_, h_, lhs, rhs)
                assert (o_[sl].float() - e32.float()).abs().max().item() <= TOL[dtype], (name, tag, b_, h_)
            del e32, e16
    if name == "c1":
        # ... and C1's sample of heads against the C oracle's restatement of the device arithmetic (oracle/fa_oracle.c,
        # OpenMP over the heads), both variants: <= 2 ulp (v_exp_f32 vs libm exp2f, MFMA vs serial fp32 sums)
        qh, kh, vh = (t[:, :, ::8].contiguous().cpu() for t in (q, k, v))
        for tag, c_, o_ in (("default", cfg, out), ("lazy", lazy, out_lazy)):
            want = fo.blockwise_for_config(c_, qh, kh, vh).float()
            assert (o_[:, :, ::8].float().cpu() - want).abs().max().item() <= TOL[dtype], (name, tag)
    # (6) a different tile shape computes the same function (different summation order)
    other = kc.as_native(replace(cfg, B_r=128, B_c=64, n_warps=4, optimized_softmax=False), speculative_softmax=False)
    oo = flash_attention.forward(other, q, k, v)
    assert (oo.float() - out.float()).abs().max().item() <= TOL[dtype]


def test_eight_launchers_on_one_host_assemble_the_single_process_result():
    """SURVEY 8e before the 8-GPU node exists: `bench.py --gpus 8 --workload c4 --batch-per-rank 1`,
    self-launched -- eight processes, one per rank, rendezvous over gloo on 127.0.0.1, each pinned to its own CPUs, all
    eight sharing THIS box's one GPU (the only thing the path's ranks ever share is the host: src/flash_attention.cu:110-112
    has no exchange step).  Every rank reports the hash of its shard's output; a single process that builds the same eight
    samples (rank r's inputs are seeded 1000 + r) and runs them as ONE batch of eight must get the same bits, sample by
    sample.  The line also carries what the rehearsal is for: per-rank host enqueue time, affinity, barrier latency."""
    import hashlib
    import json
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "c4", "--batch-per-rank", "1", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline", "--no-traffic", "--hermetic-reps", "0", "--no-mfma-roof", "--no-variants",
           "--precondition-ms", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    per = line["per_gpu"]
    assert line["n_gpus"] == 8 and len(per) == 8 and line["scaling"] == "weak"
    assert all(p_["batch_rows"] == [i, i + 1] for i, p_ in enumerate(per))
    assert all(p_["host_us_per_launch"] > 0 and p_["idle_barrier_us_median"] > 0 for p_ in per)
    cpus = [p_["affinity"]["cpus"] for p_ in per if p_.get("affinity") and p_["affinity"].get("pinned")]
    assert len(set(cpus)) == len(cpus)          # (pinned ranks never share a CPU slice)
    _, batch, heads, seq, d = bench.WORKLOADS["c4"]
    cfg = kc.best_config(kc.DType.BF16, seq)
    q, k, v = (torch.empty((8, seq, heads, d), dtype=torch.bfloat16, device=DEV) for _ in range(3))
    for rk in range(8):
        gen = torch.Generator(device=DEV).manual_seed(1000 + rk)
        for dst, src in zip((q, k, v), bench.make_inputs("randn", (1, seq, heads, d), torch.bfloat16, torch.device(DEV), gen)):
            dst[rk:rk + 1].copy_(src)
    out = flash_attention.forward(cfg, q, k, v)
    torch.cuda.synchronize()
    for rk in range(8):
        mine = hashlib.sha256(out[rk:rk + 1].contiguous().cpu().view(torch.int16).numpy().tobytes()).hexdigest()
        assert mine == per[rk]["output_sha256"], rk


@pytest.mark.parametrize("shape", ["persistent", "32-row", "key-split", "16-row"])
def test_whole_c4_job_on_one_device_offsets_beyond_4_gib(shape):
    """BASELINE.json configs[3] unsharded: B=64 H=32 S=8192 d=128 bf16 -- every tensor is 4 GiB, so byte offsets of
    the last samples need more than 32 bits (batch / head offsets are 64-bit in every kernel; only the offset inside
    one sequence is 32-bit, which validate() bounds).  Each sample of the big launch must equal, bit for bit, the
    same sample launched alone -- that is also the 8-GPU shard independence at the job's own size."""
    B, H, S = 64, 32, 8192
    free, _ = torch.cuda.mem_get_info(0)
    if free < 24 << 30:
        pytest.skip("needs 16 GiB of q/k/v/o")
    cfg = {"persistent": kc.best_config(kc.DType.BF16),
           "32-row": _native(kc.DType.BF16, 128, 64, 4, True, True),
           "key-split": _native(kc.DType.BF16, 64, 64, 4, True, True),
           "16-row": _native(kc.DType.BF16, 64, 32, 4, False, False)}[shape]
    gen = torch.Generator(device=DEV).manual_seed(64)
    q, k, v = (torch.empty((B, S, H, 128), dtype=torch.bfloat16, device=DEV) for _ in range(3))
    for t in (q, k, v):
        for b in range(0, B, 8):                      # fill by slices: no 16-GiB fp32 temporary
            t[b:b + 8].normal_(generator=gen)
    assert q.numel() * q.element_size() == 1 << 32
    out = flash_attention.forward(cfg, q, k, v)
    for b in (0, 31, 32, 63):
        alone = flash_attention.forward(cfg, q[b:b + 1], k[b:b + 1], v[b:b + 1])
        assert torch.equal(alone[0], out[b]), (shape, b)
    ref = ut.py_flash_attention(q[B - 1:B, :, H - 1:H].contiguous(), k[B - 1:B, :, H - 1:H].contiguous(),
                                v[B - 1:B, :, H - 1:H].contiguous(), upcast=True)
    assert (out[B - 1:B, :, H - 1:H].float() - ref.float()).abs().max().item() <= TOL[torch.bfloat16]
    del q, k, v, out
    torch.cuda.empty_cache()


# BASELINE.json configs[2]: the bf16 seq_len sweep, batch from the reference's table
# (py/flash_helpers/test/utils.py:9-17), heads 16 -- every shape at full size
C2 = [(S, ut.BATCH_SIZE_FOR_SEQ_LEN[S]) for S in (512, 1024, 2048, 4096, 8192, 16384)]


@pytest.mark.parametrize("S,B", C2, ids=["c2-S%d-B%d" % sb for sb in C2])
def test_c2_sweep_shape_properties(S, B):
    H, dtype = ut.BENCHMARK_N_HEADS, torch.bfloat16
    cfg = kc.best_config(kc.DType.BF16, S)
    qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype, device=torch.device(DEV))
    q, k, v = ut.generate_qkv(qc, seed=S)
    out = flash_attention.forward(cfg, q, k, v)
    assert torch.isfinite(out.float()).all()
    # exact linearity in V under a power of two
    out2 = flash_attention.forward(cfg, q, k, v * 2)
    assert torch.equal(out2, out * 2)
    # batch-shard bit-identity (a sub-batch is the same items on other workgroups)
    sub = flash_attention.forward(cfg, q[B // 2:].contiguous(), k[B // 2:].contiguous(), v[B // 2:].contiguous())
    assert torch.equal(sub, out[B // 2:])
    # convex hull of V, constant V
    vmin = v.float().amin(dim=1, keepdim=True) - 2e-2
    vmax = v.float().amax(dim=1, keepdim=True) + 2e-2
    assert ((out.float() >= vmin) & (out.float() <= vmax)).all()
    oc = flash_attention.forward(cfg, q, k, torch.ones_like(v))
    assert (oc.float() - 1).abs().max().item() <= 2.0 ** -7
    # one head against fp32 eager, and everything against another tile shape
    sl = (slice(B - 1, B), slice(None), slice(3, 4))
    ref = ut.py_flash_attention(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), upcast=True)
    assert (out[sl].float() - ref.float()).abs().max().item() <= TOL[dtype]
    other = _native(kc.DType.BF16, 128, 64, 4, True, False)
    oo = flash_attention.forward(other, q, k, v)
    assert (oo.float() - out.float()).abs().max().item() <= TOL[dtype]
    # bitwise run-to-run determinism at full size
    assert torch.equal(flash_attention.forward(cfg, q, k, v), out)


def test_c4_all_eight_shards_on_one_gpu():
    """BASELINE.json configs[4]: batch 64, heads 32, seq_len 8192 bf16, sharded 8 ways by batch with no
    exchange (src/flash_attention.cu:110-112: the grid is independent over batch x head x Q block).
    All eight shards of bench.py's shard_for_rank are walked on the one GPU here: every shard's output
    must be bit-identical to the same batch entries computed inside a two-shard launch (an item's
    result may not depend on which launch, workgroup or walk position served it), and a head of every
    shard is checked against fp32 eager."""
    import bench

    G, H, S, dtype = 64, 32, 8192, torch.bfloat16
    cfg = kc.best_config(kc.DType.BF16, S)
    gen = torch.Generator(device=DEV).manual_seed(64)
    q, k, v = (torch.randn((G, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
    shards = [bench.shard_for_rank(G, 8, r) for r in range(8)]
    assert shards[0] == (0, 8) and shards[-1] == (56, 64) and all(hi - lo == 8 for lo, hi in shards)
    outs = []
    for lo, hi in shards:
        o = flash_attention.forward(cfg, q[lo:hi], k[lo:hi], v[lo:hi])  # contiguous batch slices, as a rank holds them
        assert torch.isfinite(o.float()).all()
        outs.append(o)
    for r in range(0, 8, 2):  # two-shard recomputation: entries lo .. hi of shards r, r + 1 in one launch
        lo, hi = shards[r][0], shards[r + 1][1]
        both = flash_attention.forward(cfg, q[lo:hi], k[lo:hi], v[lo:hi])
        assert torch.equal(both[:8], outs[r]) and torch.equal(both[8:], outs[r + 1])
    for r, (lo, hi) in enumerate(shards):
        sl = (slice(lo + r, lo + r + 1), slice(None), slice(4 * r, 4 * r + 1))
        ref = ut.py_flash_attention(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), upcast=True)
        assert (outs[r][r:r + 1, :, 4 * r:4 * r + 1].float() - ref.float()).abs().max().item() <= TOL[dtype]


# ---- scope wideners beyond the reference: causal mask, ragged seq_len -----------
# (a native config that asks for the speculative softmax has a masked form on the persistent kernel only)
MASKED = [c for c in VARIANTS if _capi.ex_supported(c, allow_ragged=True, speculative=kc.wants_speculative(c),
                                                   prescaled_q=bool(getattr(c, "prescaled_q", False)))]


def _rel_ok(out, ref, dtype):
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    return bool(((out.float() - ref.float()).abs() <= ulp + ulp * ref.float().abs()).all())


@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
@pytest.mark.parametrize("S", [1, 100, 256, 257, 1000, 2048, 2500, 2560])
def test_masked_variants_against_eager_sdpa_and_oracle(S, causal):
    assert len(MASKED) >= 10
    for dtype in (torch.bfloat16, torch.float16):
        gen = torch.Generator(device=DEV).manual_seed(S + causal)
        q, k, v = (torch.randn((2, S, 3, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        ref = fo.eager_attention_masked(q, k, v, causal)            # fp32 eager statement, on the GPU
        sdpa = torch.nn.functional.scaled_dot_product_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal).transpose(1, 2)
        oracle = None
        for cfg in MASKED:
            if cfg.dtype.to_torch_dtype() != dtype:
                continue
            if kc.uses_lazy_rescale(cfg) and S % cfg.B_r and S < cfg.B_c:
                # the persistent kernel's ragged form fetches a tile that would reach beyond the sequence as
                # the window of its last B_c keys: it needs that many (fa_kernel_info.masked == 2)
                with pytest.raises(RuntimeError, match="needs seq_len >= B_c"):
                    flash_attention.forward_ex(cfg, q, k, v, causal=causal)
                continue
            out = flash_attention.forward_ex(cfg, q, k, v, causal=causal)
            assert torch.isfinite(out.float()).all(), (str(cfg), S, causal)
            assert _rel_ok(out, ref, dtype), (str(cfg), S, causal, (out.float() - ref.float()).abs().max().item())
            assert _rel_ok(out, sdpa, dtype) or (out.float() - sdpa.float()).abs().max().item() <= 2 * TOL[dtype]
            if S <= 1000 and oracle is None:
                oracle = fo.blockwise_forward_masked(q.cpu(), k.cpu(), v.cpu(), cfg.B_r, cfg.B_c, causal,
                                                     optimized_softmax=cfg.optimized_softmax)
                assert _rel_ok(out.cpu(), oracle, dtype)


def test_masked_variant_equals_plain_kernel_when_nothing_is_masked():
    """At a tile-multiple seq_len without causal mask the widened variant must reproduce the
    reference-scope kernel bit for bit.  (The masked forms of the 32-rows-per-wave kernels keep a running max and
    build the reference's first-block skip under optimized_softmax, which the plain LDS-DMA kernels ignore: the
    skip multiplies by exact zeros and ones, so the two still agree bit for bit.)"""
    for cfg in MASKED:
        dtype = cfg.dtype.to_torch_dtype()
        gen = torch.Generator(device=DEV).manual_seed(3)
        # (a configuration with a ring form -- (128, 64, 4) + buffer -- runs another device form, with the lazy rescale, at
        # multiples of 256: its masked variant is compared with the 32-rows-per-wave body, i.e. off those lengths)
        S = 896 if kc.has_ring_form(cfg) else 1024
        q, k, v = (torch.randn((2, S, 4, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        assert kc.uses_speculative_softmax(cfg, masked=True) == kc.uses_speculative_softmax(cfg)  # (by construction of MASKED)
        masked_out, plain_out = flash_attention.forward_ex(cfg, q, k, v), flash_attention.forward(cfg, q, k, v)
        if kc.uses_lazy_rescale(cfg) and kc.uses_speculative_softmax(cfg):
            # round 4: the plain speculative form of the persistent kernel carries the next tile's first four softmax units
            # in a visit's last gaps (rotated plan) and adds their share of a row sum as one side sum; the
            # masked forms keep the unrotated plan.  Same P, same O accumulation -- only the fp32 row sum is associated
            # differently, so l may differ in its last bit and an output element by one ulp of the 16-bit type
            # Round 6: the plain speculative form also walks K / V first-to-last (its reference is the first 64 keys' row max,
            # the masked form's the last 64 keys'): two roundings of the same function -- the tolerance of every parity test
            diff = (masked_out.float() - plain_out.float()).abs()
            assert (diff <= TOL[dtype] * (1 + plain_out.float().abs())).all(), (str(cfg), diff.max().item())
        else:
            assert torch.equal(masked_out, plain_out), str(cfg)


@pytest.mark.parametrize("shape", [(8, 16, 1024), (40, 16, 256), (5, 7, 512), (3, 16, 2048), (2, 16, 4096),
                                   (1, 8, 16384), (2, 4, 6144)], ids=lambda s: "B%d_H%d_S%d" % s)
def test_persistent_walk_causal(shape):
    """Causal mask on the persistent 64-rows-per-wave kernel: items of different length (4 (qb + 1)
    tiles) along one walk, the mask applied to the S tile a visit forms for the NEXT item, waves whose
    first tiles are masked whole (m = -inf until their diagonal tile).  Against fp32 eager with the
    mask, the masked 32-rows-per-wave kernel, and itself (bitwise)."""
    B, H, S = shape
    for dtype, name, opt in ((torch.bfloat16, kc.DType.BF16, False), (torch.float16, kc.DType.FP16, False),
                             (torch.bfloat16, kc.DType.BF16, True), (torch.float16, kc.DType.FP16, True)):
        cfg = _native(name, 256, 64, 4, True, opt)
        other = _native(name, 128, 64, 4, True, False)
        qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=B + S + 1)
        runs = [flash_attention.forward_ex(cfg, q, k, v, causal=True) for _ in range(3)]
        assert torch.isfinite(runs[0].float()).all()
        assert all(torch.equal(runs[0], r) for r in runs[1:]), (str(cfg), shape)
        ref = flash_attention.forward_ex(other, q, k, v, causal=True)
        assert _rel_ok(runs[0], ref, dtype) or (runs[0].float() - ref.float()).abs().max().item() <= 2 * TOL[dtype]
        for b, h in ((0, 0), (B - 1, H - 1)):  # one head each: the fp32 score matrix is S x S
            qs, ks, vs = (t[b:b + 1, :, h:h + 1].contiguous() for t in (q, k, v))
            eager = fo.eager_attention_masked(qs, ks, vs, True)
            assert _rel_ok(runs[0][b:b + 1, :, h:h + 1], eager, dtype), (str(cfg), shape)


@pytest.mark.parametrize("S", [64, 65, 127, 255, 257, 300, 511, 1000, 1025, 2500, 4000, 8191])
def test_persistent_ragged_lengths(S):
    """Any seq_len >= 64 on the persistent kernel (second masked form): Q blocks rounded up, whole ring
    rounds of K / V tiles, a tile that would reach beyond the sequence fetched as the window of its last
    64 keys with the keys in front of the tile's own first key masked, Q rows beyond the sequence fetched
    from the last row and not stored.  Full and causal, both dtypes: against the 32-rows-per-wave masked
    kernel, fp32 eager on two heads, bitwise repeatable, and nothing written outside the S rows (O is a
    window of a larger buffer whose guard rows must keep their fill)."""
    import ctypes
    lib = _capi.load()
    B, H = (3, 5) if S < 2048 else (2, 3)
    for dtype, name, opt in ((torch.bfloat16, kc.DType.BF16, False), (torch.float16, kc.DType.FP16, False),
                             (torch.bfloat16, kc.DType.BF16, True), (torch.float16, kc.DType.FP16, True)):
        cfg = _native(name, 256, 64, 4, True, opt)
        other = _native(name, 128, 64, 4, True, False)
        gen = torch.Generator(device=DEV).manual_seed(S)
        # q, k, v, o: the first S rows of (B, S + 8, H, 128) buffers (through the C ABI: the shim wants contiguous)
        big = [torch.randn((B, S + 8, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3)]
        q, k, v = (t[:, :S] for t in big)
        qc, kc_, vc = (t.contiguous() for t in (q, k, v))
        for causal in (False, True):
            runs = []
            for _ in range(3):
                guard = torch.full((B, S + 8, H, 128), 7.0, dtype=dtype, device=DEV)
                o = guard[:, :S]
                args = _capi.FaFwdArgs(q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=o.data_ptr(), batch=B,
                                       seq_len=S, n_heads=H, d_head=128, batch_stride=q.stride(0),
                                       seq_stride=q.stride(1), head_stride=q.stride(2), cfg=_capi.make_config(cfg))
                _launch_ex(lib, args, cfg, causal=causal, allow_ragged=True)
                torch.cuda.synchronize()
                assert bool((guard[:, S:] == 7.0).all()), (S, causal, "rows beyond seq_len were written")
                runs.append(o.contiguous())
            assert torch.isfinite(runs[0].float()).all(), (S, causal)
            assert all(torch.equal(runs[0], r) for r in runs[1:]), (str(cfg), S, causal)
            assert torch.equal(runs[0], flash_attention.forward_ex(cfg, qc, kc_, vc, causal=causal))
            ref = flash_attention.forward_ex(other, qc, kc_, vc, causal=causal)
            assert _rel_ok(runs[0], ref, dtype) or (runs[0].float() - ref.float()).abs().max().item() <= 2 * TOL[dtype]
            for b, h in ((0, 0), (B - 1, H - 1)):
                qs, ks, vs = (t[b:b + 1, :, h:h + 1].contiguous() for t in (qc, kc_, vc))
                eager = fo.eager_attention_masked(qs, ks, vs, causal)
                assert _rel_ok(runs[0][b:b + 1, :, h:h + 1], eager, dtype), (str(cfg), S, causal, b, h)


@pytest.mark.parametrize("S", [4000, 2080, 1990])
def test_ragged_item_seams_under_load(S):
    """At an item seam the counted DMA waits allow the epilogue's row stores to stay in flight in front
    of the next item's pieces.  A wave whose rows reach beyond the sequence issues FEWER than 16 stores
    (S = 4000: waves 2 and 3 of the last Q block issue 8 and 0; S = 2080: 8 / 0 / 0 / 0 for waves 0..3;
    S = 1990: 16 / 16 / 16 / 2), and a wait that assumed 16 would publish K / V stages whose pieces had not
    landed -- a timing-dependent wrong result.  Many more items than workgroups (every workgroup crosses
    ragged seams, with new-head pages at most of them), cold and warm runs, full and causal: bitwise
    repeatable and equal to the 32-rows-per-wave masked kernel within tolerance on EVERY row."""
    B, H = 6, 24
    n_qb = (S + 255) // 256
    assert B * H * n_qb > 4 * 256
    for dtype, name, opt in ((torch.bfloat16, kc.DType.BF16, False), (torch.float16, kc.DType.FP16, True)):
        cfg = _native(name, 256, 64, 4, True, opt)
        other = _native(name, 128, 64, 4, True, False)
        gen = torch.Generator(device=DEV).manual_seed(S)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        flush = torch.empty(600 * 1024 * 1024, dtype=torch.int8, device=DEV)
        for causal in (False, True):
            ref = flash_attention.forward_ex(other, q, k, v, causal=causal)
            runs = []
            for rep in range(6):
                if rep % 2 == 0:
                    flush.zero_()  # cold caches and TLBs: late landings at the seams
                runs.append(flash_attention.forward_ex(cfg, q, k, v, causal=causal))
            torch.cuda.synchronize()
            assert all(torch.equal(runs[0], r) for r in runs[1:]), (S, str(dtype), causal)
            assert torch.isfinite(runs[0].float()).all()
            assert _rel_ok(runs[0], ref, dtype) or (runs[0].float() - ref.float()).abs().max().item() <= 2 * TOL[dtype]


def test_persistent_walk_random_shapes():
    """Seeded fuzz over (batch, heads, seq_len multiple of 256), both dtypes, full and causal: the
    persistent kernel against the 32-rows-per-wave kernels (different tile shape, eager rescale),
    bitwise repeatable.  Item counts from 1 to a few thousand, heads not a multiple of 8, uneven
    last rounds of the walk."""
    import random
    rng = random.Random(20260927)
    for trial in range(24):
        B, H = rng.choice([1, 2, 3, 5, 9, 17]), rng.choice([1, 2, 3, 4, 8, 12, 16])
        S = 256 * rng.choice([1, 2, 3, 4, 5, 8, 12, 16])
        if B * H * S > 17 * 16 * 2048:
            S = 256 * rng.choice([1, 2, 4])
        causal = bool(trial & 1)
        dtype, name = ((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16))[(trial >> 1) & 1]
        cfg = _native(name, 256, 64, 4, True, bool(trial & 4))
        other = _native(name, 128, 64, 4, True, False)
        gen = torch.Generator(device=DEV).manual_seed(trial)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        a = flash_attention.forward_ex(cfg, q, k, v, causal=causal) if causal else flash_attention.forward(cfg, q, k, v)
        b = flash_attention.forward_ex(cfg, q, k, v, causal=causal) if causal else flash_attention.forward(cfg, q, k, v)
        ref = flash_attention.forward_ex(other, q, k, v, causal=causal)
        assert torch.equal(a, b), (trial, B, H, S, causal)
        assert torch.isfinite(a.float()).all(), (trial, B, H, S, causal)
        assert _rel_ok(a, ref, dtype) or (a.float() - ref.float()).abs().max().item() <= 2 * TOL[dtype], (trial, B, H, S, causal)


def test_c_abi_strided_views_and_long_sequence():
    """Through the C ABI directly (the Python shim, like the reference, insists on contiguous
    tensors): q, k, v, o as the first H heads of (B, S, 2H, 128) buffers -- seq_stride = 2H*128,
    head_stride = 128 -- and one S = 32768 item walk (n_kv = 512 tiles per item), against the same
    data laid out contiguously."""
    import ctypes
    lib = _capi.load()
    for cfg in (kc.best_config(kc.DType.BF16, 4096),
                _native(kc.DType.BF16, 256, 128, 8, False, True)):
        B, S, H = 3, 1024, 5
        gen = torch.Generator(device=DEV).manual_seed(11)
        big = [torch.randn((B, S, 2 * H, 128), dtype=torch.bfloat16, device=DEV, generator=gen) for _ in range(3)]
        obig = torch.zeros((B, S, 2 * H, 128), dtype=torch.bfloat16, device=DEV)
        q, k, v = (t[:, :, :H] for t in big)
        o = obig[:, :, :H]
        args = _capi.FaFwdArgs(q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=o.data_ptr(), batch=B, seq_len=S,
                               n_heads=H, d_head=128, batch_stride=q.stride(0), seq_stride=q.stride(1),
                               head_stride=q.stride(2), cfg=_capi.make_config(cfg))
        _launch_ex(lib, args, cfg)
        torch.cuda.synchronize()
        ref = flash_attention.forward(cfg, q.contiguous(), k.contiguous(), v.contiguous())
        assert torch.equal(o, ref), str(cfg)
        assert torch.count_nonzero(obig[:, :, H:]) == 0   # the other heads of the buffer are untouched
    cfg = kc.best_config(kc.DType.BF16, 32768)
    other = _native(kc.DType.BF16, 128, 64, 4, True, False)
    gen = torch.Generator(device=DEV).manual_seed(12)
    q, k, v = (torch.randn((1, 32768, 2, 128), dtype=torch.bfloat16, device=DEV, generator=gen) for _ in range(3))
    a = flash_attention.forward(cfg, q, k, v)
    assert torch.isfinite(a.float()).all()
    assert (a.float() - flash_attention.forward(other, q, k, v).float()).abs().max().item() <= TOL[torch.bfloat16]


def test_reference_errors_unchanged_without_the_wideners():
    cfg = kc.best_config(kc.DType.BF16, seq_len=320, masked=True)
    q = torch.zeros((1, 320, 2, 128), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of B_r"):
        flash_attention.forward(cfg, q, q, q)
    assert flash_attention.forward_ex(cfg, q, q, q).shape == q.shape
    no_mask = kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 64, 32, 4, True, True, True, 2, 2, 0, False, False)
    with pytest.raises(RuntimeError, match="no causal"):
        flash_attention.forward_ex(no_mask, q, q, q, causal=True)


def test_causal_full_size_timing_sanity():
    """C1 shape, causal: about half the FLOPs -> clearly faster than the full mask."""
    cfg = kc.best_config(kc.DType.BF16, masked=True)
    qc = ut.QKVConfig(n_heads=16, d_head=128, batch_size=4, seq_len=4096, dtype=torch.bfloat16,
                      device=torch.device(DEV))
    q, k, v = ut.generate_qkv(qc, seed=0)
    for _ in range(3):
        flash_attention.forward_ex(cfg, q, k, v, causal=True)
    t_causal = min(flash_attention.forward_ex(cfg, q, k, v, causal=True, timed=True)[1] for _ in range(5))
    t_full = min(flash_attention.forward_timed(cfg, q, k, v)[1] for _ in range(5))
    assert t_causal < 0.75 * t_full, (t_causal, t_full)
    ref = fo.eager_attention_masked(q[:1, :, :2].contiguous(), k[:1, :, :2].contiguous(), v[:1, :, :2].contiguous(), True)
    out = flash_attention.forward_ex(cfg, q[:1, :, :2].contiguous(), k[:1, :, :2].contiguous(), v[:1, :, :2].contiguous(), causal=True)
    assert _rel_ok(out, ref, torch.bfloat16)


def test_extreme_logit_ranges_stay_finite_and_accurate():
    """Edge inputs: huge logits (near one-hot softmax), tiny logits (uniform softmax), identical
    keys (exact ties), and a constant shift of all logits of a row (softmax-invariant)."""
    for dtype in (torch.bfloat16, torch.float16):
        cfgs = [kc.best_config(kc.DType.BF16 if dtype == torch.bfloat16 else kc.DType.FP16, s) for s in (4096, 512)]
        qc = ut.QKVConfig(n_heads=4, d_head=128, batch_size=2, seq_len=1024, dtype=dtype, device=torch.device(DEV))
        q, k, v = ut.generate_qkv(qc, seed=21)
        cases = {
            "huge": (q * 6, k * 6, v),            # |logit| up to ~2000 before scaling
            "tiny": (q * 1e-3, k * 1e-3, v),
            "ties": (q, k[:, :1].expand_as(k).contiguous(), v),
            "zeros": (torch.zeros_like(q), k, v),
        }
        for name, (qq, kk, vv) in cases.items():
            ref = ut.py_flash_attention(qq, kk, vv, upcast=True)
            for cfg in cfgs:
                out = flash_attention.forward(cfg, qq, kk, vv)
                assert torch.isfinite(out.float()).all(), (name, str(cfg))
                err = (out.float() - ref.float()).abs()
                tol = TOL[dtype] * (1 + ref.float().abs())
                assert (err <= tol).all(), (name, str(cfg), err.max().item())



@pytest.mark.parametrize("cfg", D64, ids=str)
def test_d_head_64_widener(cfg):
    """d_head = 64 (the reference's config comments allow it, no kernel was ever built): parity
    against the fp32 eager statement, torch SDPA, the C oracle and the reference's rule."""
    dtype = cfg.dtype.to_torch_dtype()
    for B, S, H in ((2, 512, 3), (1, 2048, 4)):
        gen = torch.Generator(device=DEV).manual_seed(S)
        q, k, v = (torch.randn((B, S, H, 64), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        out = flash_attention.forward(cfg, q, k, v)
        ref_f32 = ut.py_flash_attention(q, k, v, upcast=True)
        ref_b16 = ut.py_flash_attention(q, k, v, upcast=False)
        assert torch.isfinite(out.float()).all()
        assert (out.float() - ref_f32.float()).abs().max().item() <= TOL[dtype]
        lhs, rhs = fo.tolerance_rule(out, ref_b16, ref_f32)
        assert lhs <= rhs
        assert (out.float() - ut.sdpa_attention(q, k, v).float()).abs().max().item() <= 2 * TOL[dtype]
        again = flash_attention.forward(cfg, q, k, v)
        assert torch.equal(out, again)
        if S == 512:
            oracle = fo.blockwise_forward(q.cpu(), k.cpu(), v.cpu(), cfg.B_r, cfg.B_c,
                                          optimized_softmax=cfg.optimized_softmax)
            assert (out.cpu().float() - oracle.float()).abs().max().item() <= TOL[dtype]
    with pytest.raises(RuntimeError, match="d_head"):
        bad = torch.zeros((1, 512, 2, 128), dtype=dtype, device=DEV)
        flash_attention.forward(cfg, bad, bad, bad)


@pytest.mark.parametrize("name", [kc.DType.BF16, kc.DType.FP16], ids=["bf16", "fp16"])
def test_non_finite_inputs_propagate_like_fp32_attention(name):
    """NaN / Inf in the inputs: the reference has no special case and neither has this build -- a NaN reaches exactly the
    outputs it reaches in fp32 eager attention, everything else stays right.  (Speculative variants take the detour: a
    NaN row sum fails the epilogue check, the item is computed again with the running max, and comes out NaN again.)
      * a NaN in one K row  -> every logit of that key: all rows of that (batch, head) are NaN
      * a NaN in one Q row  -> that output row only
      * a NaN in V[j, d]    -> column d of every row of that (batch, head)
      * +Inf in V[j, d]     -> column d is +Inf or NaN (NaN where the key's weight underflowed to 0), never finite-wrong
    """
    dtype = name.to_torch_dtype()
    B, S, H = 2, 1024, 3
    cfgs = [kc.best_config(name), _persistent_cfg(name, True), _persistent_cfg(name, False),
            _native(name, 128, 64, 4, True, False), _native(name, 128, 64, 4, True, True)]
    gen = torch.Generator(device=DEV).manual_seed(77)
    base = [torch.randn(B, S, H, 128, device=DEV, dtype=dtype, generator=gen) for _ in range(3)]
    want = ut.py_flash_attention(*base, upcast=True).to(dtype)

    def run(cfg, q, k, v):
        out = flash_attention.forward(cfg, q, k, v)
        torch.cuda.synchronize()
        return out

    for cfg in cfgs:
        # K: key 700 of (batch 1, head 2)
        q, k, v = (t.clone() for t in base)
        k[1, 700, 2, 5] = float("nan")
        out = run(cfg, q, k, v)
        assert torch.isnan(out[1, :, 2]).all(), str(cfg)
        mask = torch.ones(B, S, H, dtype=torch.bool, device=DEV)
        mask[1, :, 2] = False
        assert torch.isfinite(out[mask]).all(), str(cfg)
        assert (out[mask].float() - want[mask].float()).abs().max().item() <= 2 * TOL[dtype], str(cfg)
        # Q: row 300 of (batch 0, head 1)
        q, k, v = (t.clone() for t in base)
        q[0, 300, 1, 127] = float("nan")
        out = run(cfg, q, k, v)
        assert torch.isnan(out[0, 300, 1]).all(), str(cfg)
        mask = torch.ones(B, S, H, dtype=torch.bool, device=DEV)
        mask[0, 300, 1] = False
        assert torch.isfinite(out[mask]).all(), str(cfg)
        assert (out[mask].float() - want[mask].float()).abs().max().item() <= 2 * TOL[dtype], str(cfg)
        # V: element (key 9, d 64) of (batch 1, head 0): NaN, then +Inf
        for bad in (float("nan"), float("inf")):
            q, k, v = (t.clone() for t in base)
            v[1, 9, 0, 64] = bad
            out = run(cfg, q, k, v)
            col = out[1, :, 0, 64]
            assert not torch.isfinite(col).any(), (str(cfg), bad)
            if bad != bad:
                assert torch.isnan(col).all(), str(cfg)
            else:
                assert (torch.isnan(col) | (col == float("inf"))).all(), str(cfg)
            mask = torch.ones(B, S, H, 128, dtype=torch.bool, device=DEV)
            mask[1, :, 0, 64] = False
            assert torch.isfinite(out[mask]).all(), (str(cfg), bad)
            assert (out[mask].float() - want[mask].float()).abs().max().item() <= 2 * TOL[dtype], (str(cfg), bad)


def test_launch_ex_with_device_counters_is_capturable_too():
    """fa_fwd_launch_ex with the native options and fa_fwd_stats is still one hipLaunchKernel on the caller's stream -- no
    host read-back: the speculative softmax's second pass runs inside the same launch and the counters are device
    atomics -- so several of them capture into one hipGraph (test_launch_is_capturable_into_a_hip_graph has the plain
    launch).  Replays give the bits of the eager launch, also after the inputs changed in place, and count their items."""
    cfgs = [kc.best_config(kc.DType.BF16), kc.best_config(kc.DType.FP16),
            _native(kc.DType.BF16, 128, 64, 4, True, False)]
    for cfg in cfgs:
        dtype = cfg.dtype.to_torch_dtype()
        gen = torch.Generator(device=DEV).manual_seed(5)
        q, k, v = (torch.randn(2, 1024, 4, 128, device=DEV, dtype=dtype, generator=gen) for _ in range(3))
        o = torch.empty_like(q)
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        eager = flash_attention.forward(cfg, q, k, v).clone()   # (also the one-time per-device init, outside the capture)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(3):
                flash_attention_kernels.forward(cfg, q, k, v, o, stats=stats)
        o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, eager), str(cfg)
        n_items = 2 * 4 * (1024 // cfg.B_r)
        assert stats.tolist() == [3 * n_items, 0], (str(cfg), stats.tolist())
        q.copy_(torch.randn(q.shape, device=DEV, dtype=dtype, generator=gen))
        want = flash_attention.forward(cfg, q, k, v).clone()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, want), str(cfg)


def test_speculative_verdict_looks_at_the_accumulators_bf16():
    """bf16 V reaches 2^127, so 'l below the limit' does not bound |O| <= l max|V|: a row whose logits rise
    30 nats (l ~ 2^52 < 2^64: the row-sum check alone passes) with |V| = 2^90 overflows the fp32 accumulators of the
    first pass.  Every speculative variant must notice (inf / NaN in O), run the item again with the running max and
    return the finite values the non-speculative variant returns."""
    B, S, H = 1, 1024, 2
    q = torch.full((B, S, H, 128), 1.0, device=DEV, dtype=torch.bfloat16)
    k_hi = torch.full((B, S, H, 128), 30.0 * (128 ** 0.5) / 128, device=DEV, dtype=torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(3)
    sign = (torch.randint(0, 2, (B, S, H, 128), device=DEV, generator=gen) * 2 - 1).to(torch.bfloat16)
    v = sign * (2.0 ** 90)
    want = None
    for B_r, B_c, n_w in ((256, 64, 4), (128, 64, 4), (64, 32, 4)):
        spec, safe = _native(kc.DType.BF16, B_r, B_c, n_w, True, True), _native(kc.DType.BF16, B_r, B_c, n_w, True, False)
        # the tile visited FIRST: logit 0; every other key: 30 nats (the persistent kernel's speculative pass -- both its forms
        # at this seq_len -- walks first-to-last, the 16-rows-per-wave kernel last-to-first)
        k = k_hi.clone()
        k[:, (slice(0, 64) if kc.walks_kv_forward(spec, seq_len=S) else slice(S - 64, S))] = 0.0
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        out = flash_attention_kernels.forward(spec, q, k, v, None, stats=stats)[0]   # (plain form: no mask, no ragged length)
        ref = flash_attention.forward(safe, q, k, v)
        torch.cuda.synchronize()
        assert torch.isfinite(ref.float()).all() and torch.isfinite(out.float()).all(), (B_r, B_c)
        items = stats.tolist()
        assert items[1] == items[0] > 0, (B_r, B_c, items)    # every item overflowed and was redone
        scale = 2.0 ** -90
        assert ((out.float() * scale) - (ref.float() * scale)).abs().max().item() <= 2.0 ** -6, (B_r, B_c)
        # (the variants that walk the same way see the same k: their running-max results agree)
        fwd = kc.walks_kv_forward(spec, seq_len=S)
        want = want if want is not None else {}
        want.setdefault(fwd, ref)
        assert ((ref.float() * scale) - (want[fwd].float() * scale)).abs().max().item() <= 2.0 ** -6, (B_r, B_c)


def test_plain_c_client_runs(tmp_path):
    """examples/c_client.c: the forward from a C99 program -- hipMalloc, fa_fwd_query, fa_fwd_launch_ex with the device
    counters and event timing -- checked in double precision on the host (exit code 0)."""
    import subprocess
    from tests.test_host_cpu import build_c_client

    exe = build_c_client(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(_capi.LIB_PATH) + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    done = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert done.returncode == 0, done.stdout + done.stderr
    assert "computed twice 0" in done.stdout and "0 outside" in done.stdout, done.stdout


def test_reference_side_binding_runs(tmp_path):
    """examples/torch_binding.cpp (the C++ pybind function of INTEGRATION.md 2, over the C ABI): same bits as this
    repository's ctypes path for reference configs, the caller's `o` written in place, event timing, the reference's
    error texts from behind the ABI."""
    from tests.test_host_cpu import build_torch_binding

    ext = build_torch_binding(tmp_path)
    gen = torch.Generator(device=DEV).manual_seed(21)
    for cfg in (kc.get_kernels_to_build()[0], kc.get_kernels_to_build()[-1],
                kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False)):
        dtype = cfg.dtype.to_torch_dtype()
        q, k, v = (torch.randn(2, 512, 3, 128, device=DEV, dtype=dtype, generator=gen) for _ in range(3))
        want = flash_attention.forward(cfg, q, k, v)
        out, ms = ext.forward(cfg, q, k, v, None)
        assert ms == 0.0 and torch.equal(out, want), str(cfg)
        o = torch.zeros_like(q)
        out2, ms2 = ext.forward(cfg, q, k, v, o, True)
        assert out2.data_ptr() == o.data_ptr() and ms2 > 0.0 and torch.equal(o, want), str(cfg)
        with pytest.raises(RuntimeError, match="Only multiples of B_r are supported for seq_len Q currently"):
            ext.forward(cfg, q[:, :500].contiguous(), k[:, :500].contiguous(), v[:, :500].contiguous(), None)
        with pytest.raises(RuntimeError, match="Kernel configuration dtype does not match input dtype"):
            other = torch.float16 if dtype == torch.bfloat16 else torch.bfloat16
            ext.forward(cfg, q.to(other), k.to(other), v.to(other), None)


_CHILD_DEFAULT = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
import flash_attention
from flash_helpers import kernel_configs as kc
heavy_first = sys.argv[1] == "1"
out = {}
for name, dtype in ((kc.DType.BF16, torch.bfloat16), (kc.DType.FP16, torch.float16)):
    cfg = kc.best_config(name, 1024)
    g = torch.Generator(device="cuda:0").manual_seed(2024)
    q, k, v = (torch.randn((2, 1024, 8, 128), dtype=dtype, device="cuda:0", generator=g) for _ in range(3))
    g2 = torch.Generator().manual_seed(1005)
    u = ((torch.randint(0, 2, (128,), generator=g2).float() * 2 - 1) * 30.0).to(dtype).to("cuda:0")
    qs, ks = q.clone(), k.clone()
    ks[1, 1020, 2] = u          # a 30-sigma key in the tile visited last: the speculative first pass fails there
    qs[1, 600:604, 2] = u
    if heavy_first:
        for i in range(40):     # a launch history full of failed items, with pauses (what the adaptive mode would react to)
            flash_attention.forward(cfg, qs, ks, v)
            if i %% 4 == 3:
                torch.cuda.synchronize()
    for tag, args in (("benign", (q, k, v)), ("spiky", (qs, ks, v))):
        o = flash_attention.forward(cfg, *args)
        torch.cuda.synchronize()
        out[str(name) + tag] = hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
print(json.dumps(out))
"""


def test_default_config_is_stateless_across_processes_and_launch_histories():
    """SURVEY 8b, threading / streams: the reference's launcher keeps no state (src/flash_attention.cu:42,118,126-131) and
    its softmax costs the same on any data (softmax.cuh:85-105).  Since round 6 so does this library's DEFAULT
    (best_config(): speculative = 1, no host-side policy): the bits of a default launch depend on its inputs only.  This
    test resets NOTHING (no fixture does any more): it runs the default on benign and on failing data (a) in this process
    after whatever the tests before it launched, (b) in a fresh child process, (c) in a second child behind a history of 40
    launches whose items failed -- and asks for the same output hashes from all three, both dtypes.  The adaptive mode
    (opt-in) keeps its per-process record and is tested beside this (test_adaptive_*)."""
    import hashlib
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = []
    for heavy_first in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _CHILD_DEFAULT % root, heavy_first], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1], (got[0], got[1])
    mine = {}
    for name, dtype in ((kc.DType.BF16, torch.bfloat16), (kc.DType.FP16, torch.float16)):
        cfg = kc.best_config(name, 1024)
        assert cfg.speculative_softmax and not cfg.adaptive_softmax
        gen = torch.Generator(device=DEV).manual_seed(2024)
        q, k, v = (torch.randn((2, 1024, 8, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        g2 = torch.Generator().manual_seed(1005)
        u = ((torch.randint(0, 2, (128,), generator=g2).float() * 2 - 1) * 30.0).to(dtype).to(DEV)
        qs, ks = q.clone(), k.clone()
        ks[1, 1020, 2] = u
        qs[1, 600:604, 2] = u
        stats = torch.zeros(2, dtype=torch.int32, device=DEV)
        flash_attention_kernels.forward(cfg, qs, ks, v, None, stats=stats)
        assert stats[1].item() > 0        # (the failing data does fail: otherwise nothing to show)
        for tag, args in (("benign", (q, k, v)), ("spiky", (qs, ks, v))):
            o = flash_attention.forward(cfg, *args)
            torch.cuda.synchronize()
            mine[str(name) + tag] = hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
            eager = ut.py_flash_attention(*args, upcast=True).float()
            assert ((o.float() - eager).abs() <= TOL[dtype] * (1 + eager.abs())).all()
    assert mine == got[0], (mine, got[0])
    from flash_attention_from_scratch_amd import _capi
    st = _capi.adaptive_state(torch.cuda.current_device(), replace(kc.best_config(kc.DType.BF16, 1024), adaptive_softmax=True))
    assert st["mode"] == 0      # (no default launch ever touches a policy record)


def test_adaptive_record_is_per_process():
    """Two processes on one device: each has its own libfa_hip.so, its own pinned report words and its own records (the
    header's 'per process').  A child process drives the default configuration into demotion on heavy data; this process'
    record of the same configuration does not move and its next adaptive launch is still the speculative variant.  (The
    adaptive mode is opt-in since round 6; the default keeps no record at all.)"""
    import json
    import subprocess
    import sys
    from flash_attention_from_scratch_amd import _capi
    dev = torch.cuda.current_device()
    cfg = replace(kc.best_config(kc.DType.BF16, 1024), adaptive_softmax=True)
    _capi.adaptive_reset(dev)
    gen = torch.Generator(device=DEV).manual_seed(99)
    q, k, v = (torch.randn((2, 1024, 8, 128), dtype=torch.bfloat16, device=DEV, generator=gen) for _ in range(3))
    before = _capi.adaptive_state(dev, cfg)
    child = r"""
import json, sys, torch
sys.path.insert(0, %r)
import flash_attention
from flash_helpers import kernel_configs as kc
from dataclasses import replace
from flash_attention_from_scratch_amd import _capi
cfg = replace(kc.best_config(kc.DType.BF16, 1024), adaptive_softmax=True)
g = torch.Generator(device="cuda:0").manual_seed(5)
q, k, v = (torch.randn((2, 1024, 8, 128), dtype=torch.bfloat16, device="cuda:0", generator=g) for _ in range(3))
g2 = torch.Generator().manual_seed(1005)
u = ((torch.randint(0, 2, (128,), generator=g2).float() * 2 - 1) * 30.0).to(torch.bfloat16).to("cuda:0")
k[1, 1020, 2] = u
q[1, 600:604, 2] = u
for i in range(12):
    flash_attention.forward(cfg, q, k, v)
    if i %% 3 == 2:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print(json.dumps(_capi.adaptive_state(0, cfg)))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    theirs = json.loads(r.stdout.strip().splitlines()[-1])
    assert theirs["reports"] >= 1 and theirs["demoted"] >= 1 and theirs["launches"] == 12
    mine = _capi.adaptive_state(dev, cfg)
    assert mine["launches"] == before["launches"] and mine["demoted"] == before["demoted"] and mine["mode"] == 0
    out = flash_attention.forward(cfg, q, k, v)
    assert torch.equal(out, flash_attention.forward(replace(cfg, adaptive_softmax=False), q, k, v))


def test_ring_form_of_the_reference_winning_shape():
    """(B_r 128, B_c 64, 4 warps) + buffer -- the reference's own winning tile shape (kernel_sass/16_A100.asm:5,
    kernel_configs.py:389-423 there) -- is served, for seq_len % 256 == 0, by the hand-placed persistent kernel with ONE
    32-row Q tile per wave (fa_fwd_kernel64<..., QTP = 1>, since round 6 with eight waves sharing the rings: 256-row items,
    counted as two Q blocks each in fa_fwd_stats; fa_kernel_info.ring_form), for the other
    multiples of 128 by the compiler-scheduled 32-rows-per-wave body.  The ring form runs the lazy rescale per 32-row tile
    exactly as the (256, 64, 4) kernel's non-speculative form does, so the two agree BIT FOR BIT; both forms are inside
    the reference's tolerance rule against the eager golden; every reference config of the shape reaches it (the
    operand-fetch hints and optimized_softmax do not change the device variant).  The speculative sibling of the
    configuration runs the same kernel's speculative schedule (first-to-last over K / V since round 6: 2 ulp of the lazy
    form); an item it gives up on is redone BY the lazy schedule (its own workgroup walks it again, whole): bit-identical to
    the lazy form's rows."""
    shape_cfgs = [c for c in kc.get_kernels_to_build() if (c.B_r, c.B_c, c.n_warps) == (128, 64, 4) and c.mma_double_buffer_loads]
    assert len(shape_cfgs) >= 8 and all(kc.has_ring_form(c) for c in shape_cfgs)
    for name, dtype in ((kc.DType.BF16, torch.bfloat16), (kc.DType.FP16, torch.float16)):
        cfgs = [c for c in shape_cfgs if c.dtype == name]
        big = kc.as_native(kc.best_config(name), speculative_softmax=False)       # (256, 64, 4)+buffer, lazy
        assert kc.softmax_mode(big) == "lazy" and kc.softmax_mode(cfgs[0], seq_len=512) == "lazy" and kc.softmax_mode(cfgs[0], seq_len=384) == "eager"
        for (B, S, H) in ((2, 512, 3), (1, 1024, 8), (3, 256, 5), (2, 2048, 16)):
            gen = torch.Generator(device=DEV).manual_seed(S + H)
            q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
            # a logit spike, so that the lazy rescale actually moves a reference max somewhere
            u = _sign_vector(7).to(dtype)
            k[0, S // 3, 1] = 12.0 * u
            q[0, S // 2:S // 2 + 3, 1] = 12.0 * u
            want = flash_attention.forward(big, q, k, v)
            eager = ut.py_flash_attention(q, k, v, upcast=True).float()
            for c in cfgs:
                out = flash_attention.forward(c, q, k, v)
                assert torch.equal(out, want), (str(c), B, S, H)
            assert ((want.float() - eager).abs() <= TOL[dtype] * (1 + eager.abs())).all()
            spec = kc.as_native(cfgs[0], speculative_softmax=True)
            assert kc.has_ring_form(spec) and kc.softmax_mode(spec, seq_len=S) == "speculative"
            stats = torch.zeros(2, dtype=torch.int32, device=DEV)
            out, _ = flash_attention_kernels.forward(spec, q, k, v, None, stats=stats)
            assert stats[0].item() == B * H * (S // 128) and 1 <= stats[1].item() <= 2 * (S // 128), stats.tolist()
            # the spiked queries' item was redone by the lazy schedule (in its own workgroup, whole): the lazy form's bits;
            # every other item ran the speculative first pass, which walks K / V first-to-last since round 6 -- the lazy
            # form's arithmetic per tile, its fp32 sums in the other order: 2 ulp of the lazy form
            blk = slice((S // 2) // 128 * 128, (S // 2) // 128 * 128 + 128)
            assert torch.equal(out[0, blk, 1], want[0, blk, 1]), (str(spec), B, S, H)
            assert ((out.float() - want.float()).abs() <= 2 * TOL[dtype] * (1 + want.float().abs())).all(), (str(spec), B, S, H)
            assert ((out.float() - eager).abs() <= TOL[dtype] * (1 + eager.abs())).all()
            # nothing to give up on: nothing redone, the 64-row speculative kernel's neighbourhood
            k2, q2 = k.clone(), q.clone()
            k2[0, S // 3, 1] = k[0, S // 3, 0]
            q2[0, S // 2:S // 2 + 3, 1] = q[0, S // 2:S // 2 + 3, 0]
            stats.zero_()
            out, _ = flash_attention_kernels.forward(spec, q2, k2, v, None, stats=stats)
            want2 = flash_attention.forward(big, q2, k2, v)
            assert stats[1].item() == 0 and ((out.float() - want2.float()).abs() <= 2 * TOL[dtype] * (1 + want2.float().abs())).all()
        # a multiple of 128 that is not one of 256: the compiler-scheduled body, the reference's eager arithmetic
        gen = torch.Generator(device=DEV).manual_seed(384)
        q, k, v = (torch.randn((2, 384, 4, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        out = flash_attention.forward(cfgs[0], q, k, v)
        eager = ut.py_flash_attention(q, k, v, upcast=True).float()
        assert ((out.float() - eager).abs() <= TOL[dtype] * (1 + eager.abs())).all()
        # many items per workgroup (the persistent walk's seams) at a C1-like head count, against the 64-row kernel
        gen = torch.Generator(device=DEV).manual_seed(11)
        q, k, v = (torch.randn((4, 2048, 16, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        assert torch.equal(flash_attention.forward(cfgs[-1], q, k, v), flash_attention.forward(big, q, k, v))
        # determinism
        first = flash_attention.forward(cfgs[0], q, k, v)
        assert all(torch.equal(flash_attention.forward(cfgs[0], q, k, v), first) for _ in range(3))
