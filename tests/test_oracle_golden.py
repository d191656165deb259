"""CPU tier: the oracle (oracle/) against the fixtures generated from the
reference's own Python oracles (oracle/gen_golden.py).  This is what pins parity."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fa_oracle as fo
from tests.conftest import GOLDEN, load_eager_golden, load_seam_golden

CASES = [(tag, case) for tag in ("bf16", "fp16") for case in "abc"]
# one 16-bit ulp at |o| < 0.5 (outputs of N(0,1) attention are O(0.1))
ULP = {"bf16": 2.0 ** -9, "fp16": 2.0 ** -12}


@pytest.mark.parametrize("tag,case", CASES)
def test_torch_restatement_is_bit_identical_to_reference_eager(tag, case):
    g = load_eager_golden(tag, case)
    assert torch.equal(fo.eager_attention(g["q"], g["k"], g["v"], upcast=True), g["o_f32"])
    assert torch.equal(fo.eager_attention(g["q"], g["k"], g["v"], upcast=False), g["o_b16"])


@pytest.mark.parametrize("tag,case", CASES)
@pytest.mark.parametrize("tiles", [(128, 64), (64, 32), (256, 128)])
def test_c_blockwise_matches_reference_fp32_eager_within_one_ulp(tag, case, tiles):
    g = load_eager_golden(tag, case)
    B_r, B_c = tiles
    out = fo.blockwise_forward(g["q"], g["k"], g["v"], B_r, B_c)
    err = (out.float() - g["o_f32"].float()).abs().max().item()
    assert err <= 2 * ULP[tag], err
    # and the reference's own accuracy bar (test.py:57-61)
    lhs, rhs = fo.tolerance_rule(out, g["o_b16"], g["o_f32"])
    assert lhs <= rhs


@pytest.mark.parametrize("tag,case", CASES)
def test_c_blockwise_optimized_softmax_is_same_arithmetic(tag, case):
    g = load_eager_golden(tag, case)
    a = fo.blockwise_forward(g["q"], g["k"], g["v"], 64, 64, optimized_softmax=False)
    b = fo.blockwise_forward(g["q"], g["k"], g["v"], 64, 64, optimized_softmax=True)
    # first block: exp2(-inf)=0 rescale of zeros vs skipping it -> identical bits
    assert torch.equal(a, b)


@pytest.mark.parametrize("tag,case", CASES)
def test_c_lazy_rescale_restatement_meets_the_reference_bars(tag, case):
    """The 64-rows-per-wave device variant moves its reference max lazily.  Its
    CPU restatement must meet the same bars as the reference's arithmetic on the reference's
    fixtures; with a vanishing threshold it rescales whenever a max moves, which IS the
    reference's arithmetic, bit for bit."""
    g = load_eager_golden(tag, case)
    S = g["q"].shape[1]
    B_r = 256 if S % 256 == 0 else 64
    out = fo.blockwise_forward_lazy(g["q"], g["k"], g["v"], B_r, 64, tau=8.0)
    assert (out.float() - g["o_f32"].float()).abs().max().item() <= 2 * ULP[tag]
    lhs, rhs = fo.tolerance_rule(out, g["o_b16"], g["o_f32"])
    assert lhs <= rhs
    # an (almost) zero threshold rescales whenever a max moves: the reference's arithmetic
    eager_rescale = fo.blockwise_forward_lazy(g["q"], g["k"], g["v"], B_r, 64, tau=1e-30)
    assert torch.equal(eager_rescale, fo.blockwise_forward(g["q"], g["k"], g["v"], B_r, 64))


@pytest.mark.parametrize("tag,case", CASES)
def test_c_speculative_restatement_meets_the_reference_bars(tag, case):
    """The speculative softmax keeps the row max of the FIRST visited tile as the
    reference for the whole item -- the lazy restatement with an infinite threshold.  Any reference
    gives the same real result while nothing overflows, so it must meet the reference's bars too."""
    g = load_eager_golden(tag, case)
    S = g["q"].shape[1]
    B_r = 256 if S % 256 == 0 else 64
    out = fo.blockwise_forward_lazy(g["q"], g["k"], g["v"], B_r, 64, tau=fo.SPEC_TAU)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - g["o_f32"].float()).abs().max().item() <= 2 * ULP[tag]
    lhs, rhs = fo.tolerance_rule(out, g["o_b16"], g["o_f32"])
    assert lhs <= rhs


def test_c_speculative_restatement_rising_logits_bf16():
    """bf16 has fp32's exponent range: P far above 1 (row maxima rising ~60 binades along the visit
    order) loses nothing -- relative rounding is scale-free -- so the never-rescale restatement
    still matches fp32 eager as closely as the reference's arithmetic does."""
    torch.manual_seed(5)
    q, k, v = (torch.randn(1, 1024, 2, 128).to(torch.bfloat16) for _ in range(3))
    k = (k.float() * torch.linspace(8, 1, 1024).view(1, -1, 1, 1)).to(torch.bfloat16)
    ref = fo.eager_attention(q, k, v, upcast=True).float()
    spec = fo.blockwise_forward_lazy(q, k, v, 256, 64, tau=fo.SPEC_TAU).float()
    exact = fo.blockwise_forward(q, k, v, 256, 64).float()
    assert torch.isfinite(spec).all()
    tol = 4 * 2.0 ** -9 * (1 + ref.abs())
    assert ((spec - ref).abs() <= tol).all()
    assert ((exact - ref).abs() <= tol).all()


def test_c_speculative_restatement_alternating_direction_keeps_block_0_first():
    """Round 6: the device's long-sequence form walks the Q blocks with (qb // G) odd as [block 0, then last-to-second]
    (fa_fwd_kernel64<..., ALT>; kernel_configs.kv_walk_alternates).  The restatement with alt_group = G: the even groups'
    rows are the first-to-last restatement's bits; the odd groups' rows differ from them in the last bits at most (same
    terms, other order) and stay inside the bars; an fp16 attention sink at the first keys overflows nothing in either
    direction (the reference is block 0's row max both ways: P <= 1 for the sink keys)."""
    from flash_helpers import kernel_configs as kc

    torch.manual_seed(11)
    q, k, v = (torch.randn(1, 1024, 2, 128).to(torch.float16) for _ in range(3))
    fwd = fo.blockwise_forward_spec(q, k, v, 256, 64, kv_forward=True)
    alt = fo.blockwise_forward_spec(q, k, v, 256, 64, kv_forward=True, alt_group=2)   # Q blocks 2, 3 walk [0, 15 .. 1]
    assert torch.equal(alt[:, :512], fwd[:, :512])
    assert not torch.equal(alt[:, 512:], fwd[:, 512:])
    ref = fo.eager_attention(q, k, v, upcast=True).float()
    assert (alt.float() - ref).abs().max().item() <= 2 * ULP["fp16"]
    a = (12.0 * 128 ** 0.5) ** 0.5
    q[..., 0] = a
    k[..., 0] = 0
    k[:, :4, :, 0] = a
    ref = fo.eager_attention(q, k, v, upcast=True).float()
    alt = fo.blockwise_forward_spec(q, k, v, 256, 64, kv_forward=True, alt_group=2)
    # (the four sink keys carry the row: outputs of magnitude 2 .. 4, where one fp16 ulp is 2^-9)
    assert torch.isfinite(alt.float()).all() and ((alt.float() - ref).abs() <= 2.0 ** -10 * (1 + ref.abs())).all()
    # which launches alternate: batch * heads a multiple of 8, a head's Q blocks an even number of rounds of 32 workgroups
    cfg = kc.best_config(kc.DType.FP16, 16384)
    assert kc.kv_walk_alternates(cfg, 64, 16384) == 32 and kc.kv_walk_alternates(cfg, 64, 32768) == 32
    assert kc.kv_walk_alternates(cfg, 64, 8192) == 0 and kc.kv_walk_alternates(cfg, 64, 4096) == 0
    assert kc.kv_walk_alternates(cfg, 12, 16384) == 0 and kc.kv_walk_alternates(cfg, 64, 16384, masked=True) == 0
    assert kc.kv_walk_alternates(cfg, 64, 16384, num_cus=304) == 0   # (38 workgroups per XCD: 64 Q blocks are not whole rounds)
    from dataclasses import replace
    assert kc.kv_walk_alternates(replace(cfg, speculative_softmax=False), 64, 16384) == 0


def test_c_lazy_rescale_staircase_logits():
    """Row maxima that keep rising along the visit order (keys near the start of the sequence are
    larger and are visited last) cross the threshold several times, in fp16 too (P <= 2^8)."""
    torch.manual_seed(3)
    for dtype, ulp in ((torch.bfloat16, 2.0 ** -9), (torch.float16, 2.0 ** -12)):
        q, k, v = (torch.randn(1, 1024, 2, 128).to(dtype) for _ in range(3))
        k = (k.float() * torch.linspace(8, 1, 1024).view(1, -1, 1, 1)).to(dtype)
        ref = fo.eager_attention(q, k, v, upcast=True).float()
        lazy = fo.blockwise_forward_lazy(q, k, v, 256, 64).float()
        exact = fo.blockwise_forward(q, k, v, 256, 64).float()
        assert torch.isfinite(lazy).all()
        tol = 4 * ulp * (1 + ref.abs())
        assert ((lazy - ref).abs() <= tol).all()
        assert ((exact - ref).abs() <= tol).all()


@pytest.mark.parametrize("tag,case", CASES)
def test_c_eager_matches_reference_eager(tag, case):
    g = load_eager_golden(tag, case)
    e32 = fo.eager_forward_c(g["q"], g["k"], g["v"], upcast=True)
    assert (e32.float() - g["o_f32"].float()).abs().max().item() <= ULP[tag]
    e16 = fo.eager_forward_c(g["q"], g["k"], g["v"], upcast=False)
    # 16-bit eager accumulates rounding differently inside torch's matmul: a few ulp
    assert (e16.float() - g["o_b16"].float()).abs().max().item() <= 8 * ULP[tag]


@pytest.mark.parametrize("case,tiles", [("a", (128, 64)), ("c", (64, 32))])
def test_blockwise_trace_matches_reference_block_flash_attention(case, tiles):
    """tools/debug/debug.py:block_flash_attention output (rows of 'warp 2')."""
    z = np.load(os.path.join(GOLDEN, f"block_{case}.npz"))
    g = load_eager_golden("bf16", case)
    B_r, B_c = tiles
    assert (int(z["B_r"]), int(z["B_c"])) == tiles
    rows = slice(int(z["row_start"]), int(z["row_stop"]))
    q2, k2, v2 = (g[n][0, :, 0].float() for n in ("q", "k", "v"))
    mine = fo.blockwise_attention_torch(q2, k2, v2, B_c, rows)
    assert np.abs(mine.numpy() - z["o_final"]).max() < 5e-7
    # C restatement with fp32 P (round_p=False) reproduces it up to the final bf16 rounding
    oc, m, l = fo.blockwise_forward(g["q"], g["k"], g["v"], B_r, B_c, round_p=False, return_stats=True)
    assert np.abs(oc[0, rows, 0].float().numpy() - z["o_final"]).max() <= 2.0 ** -9
    # statistics: m is the raw-logit row max, l the base-2 row sum
    S = q2 @ k2.T
    assert torch.allclose(m[0, 0], S.max(dim=-1).values, rtol=0, atol=1e-4)
    scale = (128 ** -0.5) * 1.4426950408889634
    l_ref = (2 ** ((S - S.max(dim=-1, keepdim=True).values) * scale)).sum(-1)
    assert torch.allclose(l[0, 0], l_ref, rtol=1e-5)


def test_16bit_conversions_match_torch_rne():
    L = fo.lib()
    gen = torch.Generator().manual_seed(3)
    xs = torch.cat([
        torch.randn(4000, generator=gen) * 3,
        torch.randn(500, generator=gen) * 1e-6,
        torch.tensor([0.0, -0.0, 65504.0, 65519.0, 65520.0, 1e-8, 6e-8, 2.0 ** -24, 2.0 ** -25,
                      3 * 2.0 ** -25, 1.0009765625, 1.00048828125, float("inf"), -float("inf")]),
    ])
    for code, dt in ((fo.BF16, torch.bfloat16), (fo.FP16, torch.float16)):
        want = xs.to(dt)
        want_bits = want.view(torch.int16).numpy().view(np.uint16)
        for x, wb, w in zip(xs.tolist(), want_bits.tolist(), want.float().tolist()):
            got = L.fa_oracle_f32_to_b16(x, code)
            assert got == wb, (x, got, wb, dt)
            back = L.fa_oracle_b16_to_f32(wb, code)
            assert back == w or (back != back and w != w)


def test_oracle_rejects_bad_tiling():
    q = torch.zeros((1, 96, 1, 128), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        fo.blockwise_forward(q, q, q, 64, 64)


def test_reference_enumerations_pinned():
    """kernel_configs.py enumerations / FLOP model, as captured from the reference."""
    from flash_helpers import kernel_configs as kc

    with open(os.path.join(GOLDEN, "configs.json")) as f:
        ref = json.load(f)
    assert [c.short_form() for c in kc.get_kernels_to_build()] == ref["kernels_to_build"]
    assert [c.to_cpp_struct() for c in kc.get_kernels_to_build()] == ref["kernels_to_build_cpp"]
    assert [c.short_form() for c in kc.get_autotuning_kernel_configs()] == ref["autotune"]
    assert [c.short_form() for c in kc.get_kernel_progression_configs()] == ref["progression"]
    assert [c.short_form() for c in kc.get_kernel_progression_configs(True)] == ref["progression_all"]
    assert kc.calc_self_attn_flop(4, 16, 4096, 128) == ref["self_attn_flop_4_16_4096_128"]
    assert kc.calc_total_flop(4, 16, 4096, 128, 64, 128) == ref["total_flop_4_16_4096_128_64_128"]
    assert kc.arithmetic_intensity(128, 64, 4096, 128) == ref["arithmetic_intensity_128_64_4096_128"]
    assert kc.get_kernels_to_build()[0].__class__(*kc.get_kernels_to_build()[0].__dict__.values()).smem_bytes() > 0
    cfg = kc.parse_kernel_name_into_config(ref["typed_name_example"]["name"])
    assert cfg.short_form() == ref["typed_name_example"]["short"]
    from flash_helpers.test import utils as ut

    assert {str(k): v for k, v in ut.BATCH_SIZE_FOR_SEQ_LEN.items()} == ref["batch_size_for_seq_len"]
    assert ut.BENCHMARK_N_HEADS == ref["benchmark_n_heads"]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("S", [100, 257])
def test_masked_oracle_matches_eager_statement(dtype, causal, S):
    """Widened modes (causal, ragged seq_len) are not in the reference: the C restatement is
    checked against the fp32 eager statement (and, without mask at a tile multiple, reduces to
    the pinned unmasked path bit for bit)."""
    gen = torch.Generator().manual_seed(S)
    q, k, v = (torch.randn((2, S, 3, 128), generator=gen).to(dtype) for _ in range(3))
    ref = fo.eager_attention_masked(q, k, v, causal).float()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for B_r, B_c in ((128, 64), (256, 128), (64, 32)):
        out = fo.blockwise_forward_masked(q, k, v, B_r, B_c, causal).float()
        assert torch.isfinite(out).all()
        assert ((out - ref).abs() <= ulp + ulp * ref.abs()).all()
    if not causal:
        g = load_eager_golden("bf16", "a")
        a = fo.blockwise_forward(g["q"], g["k"], g["v"], 128, 64)
        b = fo.blockwise_forward_masked(g["q"], g["k"], g["v"], 128, 64, causal=False)
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["bf16", "fp16"])
def test_seam_golden_with_planted_spikes_against_the_restatements(tag):
    """tests/golden/seam_<tag>.npz (oracle/gen_golden.py): the reference's py_flash_attention on a case big enough to
    cross item seams of the persistent walk (288 items of 256 rows) with logit spikes planted in a first and in a second
    item of a workgroup -- the shape the smaller fixtures never reach.  Inputs are rebuilt from the recipe (seed +
    spikes, checksummed); the stored row sample pins the torch restatement bit for bit and the C restatements (the
    reference's arithmetic, and the lazy rescale that serves as the speculative softmax's second pass) to the usual bars."""
    g = load_seam_golden(tag)
    # the torch restatement on the whole tensor (bit identity needs the reference's own shapes: the matmul's summation order)
    assert torch.equal(fo.eager_attention(g["q"], g["k"], g["v"], upcast=True)[g["b"], g["r"], g["h"]], g["o_f32"])
    # (bit patterns: in fp16 the reference's 16-bit eager overflows on the 30-sigma spikes -- q.k = 115 200 > 65 504 --
    # and returns NaN rows there; they are part of the fixture, and excluded from the reference's rule below)
    assert torch.equal(fo.eager_attention(g["q"], g["k"], g["v"], upcast=False)[g["b"], g["r"], g["h"]].view(torch.int16),
                       g["o_b16"].view(torch.int16))
    # the C restatements on the three spiked heads and five others (of 72); the sample rows of the rest are covered on the
    # GPU, where the whole tensor goes through every kernel form
    picked = sorted({(int(s_[0]), int(s_[1])) for s_ in g["spikes"]} | {(0, 0), (0, 11), (1, 23), (2, 3), (2, 19)})
    for (bb, hh) in picked:
        sel = (g["b"] == bb) & (g["h"] == hh)
        rows = g["r"][sel]
        q, k, v = (g[n][bb:bb + 1, :, hh:hh + 1].contiguous() for n in ("q", "k", "v"))
        o_f32, o_b16 = g["o_f32"][sel], g["o_b16"][sel]
        sane = torch.isfinite(o_b16.float()).all(dim=-1)
        tol = 2 * ULP[tag] * (1 + o_f32.float().abs())
        for out in (fo.blockwise_forward(q, k, v, 128, 64), fo.blockwise_forward_lazy(q, k, v, 256, 64, tau=8.0)):
            got = out[0, rows, 0]
            assert ((got.float() - o_f32.float()).abs() <= tol).all()
            lhs, rhs = fo.tolerance_rule(got[sane], o_b16[sane], o_f32[sane])
            assert lhs <= rhs
    all_sane = torch.isfinite(g["o_b16"].float()).all(dim=-1)
    assert all_sane.all() if tag == "bf16" else (~all_sane).sum() == 8 + 4   # the rows of the two 30-sigma spikes
    q, k, v = g["q"], g["k"], g["v"]
    # the spiked rows really are spikes: one key takes (almost) all the weight, so the output row is that key's V row
    for (bb, hh, key, row0, nrows, amp, _s) in g["spikes"]:
        if amp > 2:
            ref_row = fo.eager_attention(q[int(bb):int(bb) + 1, :, int(hh):int(hh) + 1].contiguous(),
                                         k[int(bb):int(bb) + 1, :, int(hh):int(hh) + 1].contiguous(),
                                         v[int(bb):int(bb) + 1, :, int(hh):int(hh) + 1].contiguous(), upcast=True)[0, int(row0), 0]
            assert torch.equal(ref_row, v[int(bb), int(key), int(hh)])
